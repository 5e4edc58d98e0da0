"""Tokenizers: the native byte-level fallback and an adapter over Hugging Face
``tokenizers`` files. The reference sends plain strings to a remote API
(/root/reference/src/main.rs:84) and never tokenizes; hosting the model in-process
means text <-> ids happens here. Members with different vocabularies exchange answers
as text and re-tokenize per member.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Sequence

from .._native import core


# Prompt wrappers of the instruction-tuned model families the engine loads. The reference
# delegates this to the remote API's chat endpoint (`client.text_only`,
# /root/reference/src/main.rs:84); a local model sees raw text and needs its own turn markers.
CHAT_FORMATS = {
    "none": "{system}{prompt}",
    "llama3": "{system}<|start_header_id|>user<|end_header_id|>\n\n{prompt}<|eot_id|>"
              "<|start_header_id|>assistant<|end_header_id|>\n\n",
    "chatml": "{system}<|im_start|>user\n{prompt}<|im_end|>\n<|im_start|>assistant\n",
    "mistral": "[INST] {system}{prompt} [/INST]",
}
_SYSTEM_FORMATS = {
    "none": "{system}\n\n",
    "llama3": "<|start_header_id|>system<|end_header_id|>\n\n{system}<|eot_id|>",
    "chatml": "<|im_start|>system\n{system}<|im_end|>\n",
    "mistral": "{system}\n\n",
}


def apply_chat_format(fmt: str, prompt: str, system: Optional[str] = None) -> str:
    """Wraps ``prompt`` as one user turn followed by the opening of the assistant turn."""
    if fmt not in CHAT_FORMATS:
        raise ValueError(f"unknown chat_format {fmt!r}; known: {sorted(CHAT_FORMATS)}")
    sys_part = _SYSTEM_FORMATS[fmt].replace("{system}", system) if system else ""
    return CHAT_FORMATS[fmt].replace("{system}", sys_part).replace("{prompt}", prompt)


class Tokenizer:
    bos_id: Optional[int] = None
    eos_id: Optional[int] = None
    vocab_size: int = 0
    stop_ids: Sequence[int] = ()      # further end-of-turn tokens besides eos_id

    def encode(self, text: str, add_bos: bool = True) -> List[int]:
        raise NotImplementedError

    def decode(self, ids: Sequence[int]) -> str:
        raise NotImplementedError

    def first_token_id(self, text: str) -> int:
        """Id of the first token of ``text`` when it continues a prompt (decision targets)."""
        ids = self.encode(text, add_bos=False)
        if not ids:
            raise ValueError(f"target {text!r} encodes to nothing")
        return ids[0]


class ByteLevelTokenizer(Tokenizer):
    """UTF-8 bytes + BOS/EOS/PAD (ids 256-258). Works with any vocab_size >= 259; ids a
    random-weight model emits beyond the byte range fold into printable bytes on decode."""

    def __init__(self, vocab_size: int = 259) -> None:
        self._t = core.ByteTokenizer()
        self.bos_id, self.eos_id, self.pad_id = self._t.bos_id, self._t.eos_id, self._t.pad_id
        self.vocab_size = max(vocab_size, self._t.vocab_size)

    def encode(self, text: str, add_bos: bool = True) -> List[int]:
        return list(self._t.encode(text, add_bos))

    def decode(self, ids: Sequence[int]) -> str:
        return self._t.decode(list(int(i) for i in ids), True).decode("utf-8", errors="replace")

    def first_token_id(self, text: str) -> int:
        # leading spaces carry no signal for a byte tokenizer: " A" and " B" would collide
        s = text.lstrip(" ") or text
        if not s:
            raise ValueError("an empty target encodes to nothing")
        return self.encode(s, add_bos=False)[0]


class HFTokenizer(Tokenizer):
    """``tokenizer.json`` through the ``tokenizers`` library (no network access needed)."""

    def __init__(self, path: str) -> None:
        from tokenizers import Tokenizer as _Tok

        p = Path(path)
        f = p / "tokenizer.json" if p.is_dir() else p
        self._t = _Tok.from_file(str(f))
        self.vocab_size = self._t.get_vocab_size()
        # Qwen checkpoints have no BOS: <|im_start|> opens a chat turn and is part of the chat
        # format, not a sequence prefix, so it is not a candidate here.
        self.bos_id = self._find("<|begin_of_text|>", "<s>")
        ends = [self._t.token_to_id(n) for n in ("<|end_of_text|>", "<|eot_id|>", "</s>", "<|im_end|>", "<|endoftext|>")]
        ends = [i for i in ends if i is not None]
        self.eos_id = ends[0] if ends else None
        self.stop_ids = tuple(ends[1:])   # e.g. Llama-3 chat turns end with <|eot_id|>, documents with <|end_of_text|>

    def _find(self, *names: str) -> Optional[int]:
        for n in names:
            i = self._t.token_to_id(n)
            if i is not None:
                return i
        return None

    def encode(self, text: str, add_bos: bool = True) -> List[int]:
        ids = self._t.encode(text, add_special_tokens=False).ids
        if add_bos and self.bos_id is not None:
            ids = [self.bos_id] + ids
        return ids

    def decode(self, ids: Sequence[int]) -> str:
        return self._t.decode(list(int(i) for i in ids), skip_special_tokens=True)


def make_tokenizer(path: Optional[str], vocab_size: int) -> Tokenizer:
    if path:
        return HFTokenizer(path)
    return ByteLevelTokenizer(vocab_size)
