"""Chat formats and the Hugging Face tokenizer adapter, on a tokenizer.json built in the test
(no files can be downloaded here; the reference never tokenizes, it sends strings to a remote
chat endpoint, /root/reference/src/main.rs:84)."""
import pytest

from llm_consensus_amd._native import core
from llm_consensus_amd.config import config_from_dict
from llm_consensus_amd.consensus.backend import GenParams
from llm_consensus_amd.engine.backend import EngineBackend
from llm_consensus_amd.engine.engine import Engine
from llm_consensus_amd.models.config import get_preset
from llm_consensus_amd.models.transformer import TransformerModel
from llm_consensus_amd.utils.tokenizer import CHAT_FORMATS, HFTokenizer, apply_chat_format


@pytest.fixture(scope="module")
def llama3_like_tokenizer(tmp_path_factory):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers

    tok = Tokenizer(models.BPE(unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    special = ["<unk>", "<|begin_of_text|>", "<|end_of_text|>", "<|eot_id|>", "<|start_header_id|>", "<|end_header_id|>"]
    trainer = trainers.BpeTrainer(vocab_size=300, special_tokens=special,
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet())
    tok.train_from_iterator(["the panel must agree on an answer", "Good NeedsRefinement user assistant system"] * 4, trainer)
    d = tmp_path_factory.mktemp("tok")
    tok.save(str(d / "tokenizer.json"))
    return HFTokenizer(str(d))


def test_formats_wrap_one_user_turn_and_open_the_assistant_turn():
    assert apply_chat_format("none", "hi") == "hi"
    s = apply_chat_format("llama3", "hi", system="be brief")
    assert s.startswith("<|start_header_id|>system<|end_header_id|>\n\nbe brief<|eot_id|>")
    assert s.endswith("<|start_header_id|>assistant<|end_header_id|>\n\n") and "user<|end_header_id|>\n\nhi<|eot_id|>" in s
    assert apply_chat_format("chatml", "hi") == "<|im_start|>user\nhi<|im_end|>\n<|im_start|>assistant\n"
    assert apply_chat_format("mistral", "hi") == "[INST] hi [/INST]"
    assert set(CHAT_FORMATS) == {"none", "llama3", "chatml", "mistral"}
    with pytest.raises(ValueError):
        apply_chat_format("vicuna", "hi")
    with pytest.raises(ValueError):
        config_from_dict({"models": {"default": {"chat_format": "none"}}, "strategy": "nope"})
    assert config_from_dict({"models": {"default": {"chat_format": "chatml"}}}).models["default"].chat_format == "chatml"


def test_hf_adapter_finds_turn_end_tokens_and_keeps_markers_atomic(llama3_like_tokenizer):
    t = llama3_like_tokenizer
    eot, eos = t._t.token_to_id("<|eot_id|>"), t._t.token_to_id("<|end_of_text|>")
    assert t.eos_id == eos and tuple(t.stop_ids) == (eot,) and t.bos_id == t._t.token_to_id("<|begin_of_text|>")
    ids = t.encode(apply_chat_format("llama3", "the answer"))
    assert ids[0] == t.bos_id and ids.count(eot) == 1
    assert ids.count(t._t.token_to_id("<|start_header_id|>")) == 2        # markers are single tokens
    assert t.decode(t.encode("the panel must agree", add_bos=False)) == "the panel must agree"
    assert t.first_token_id("Good") != t.first_token_id("NeedsRefinement")


def test_backend_applies_the_format_and_stops_at_any_turn_end(llama3_like_tokenizer):
    cfg = get_preset("tiny-llama")
    model = TransformerModel.random_init(cfg, "cpu", seed=2, std=0.1)
    eng = Engine(model, num_kv_blocks=64, block_size=16, use_graphs=False)
    plain = EngineBackend(eng, llama3_like_tokenizer, name="plain")
    chat = EngineBackend(eng, llama3_like_tokenizer, name="chat", chat_format="llama3")
    sp = chat._params(GenParams(max_new_tokens=4))
    assert sp.eos_id == llama3_like_tokenizer.eos_id and list(sp.stop_ids) == list(llama3_like_tokenizer.stop_ids)
    h1 = plain.submit_generate("the answer", GenParams(temperature=0.0, max_new_tokens=3))
    h2 = chat.submit_generate("the answer", GenParams(temperature=0.0, max_new_tokens=3))
    plain.run(); chat.run()
    assert h2.result().prompt_tokens > h1.result().prompt_tokens + 8    # turn markers were added
    with pytest.raises(ValueError):
        EngineBackend(eng, llama3_like_tokenizer, chat_format="bogus")


def test_qwen_style_vocabulary_gets_no_bos(tmp_path):
    """Qwen checkpoints have no BOS token: <|im_start|> opens a chat turn and must not be
    prepended to every prompt as if it were one."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers

    tok = Tokenizer(models.BPE(unk_token=None))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=280, special_tokens=["<|endoftext|>", "<|im_start|>", "<|im_end|>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet())
    tok.train_from_iterator(["user assistant system answer"] * 4, trainer)
    tok.save(str(tmp_path / "tokenizer.json"))
    t = HFTokenizer(str(tmp_path))
    assert t.bos_id is None
    im_start = t._t.token_to_id("<|im_start|>")
    assert t.encode("answer")[0] != im_start
    ids = t.encode(apply_chat_format("chatml", "answer"))
    assert ids[0] == im_start and ids.count(im_start) == 2            # only the two the format wrote
    assert t.eos_id == t._t.token_to_id("<|im_end|>") and t._t.token_to_id("<|endoftext|>") in t.stop_ids
