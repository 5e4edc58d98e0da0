"""Checkpoint loader: shard-sliced lazy reads and the tensor-parallel / expert-parallel shards
they produce. (The reference has no checkpoints: its model is a remote API,
/root/reference/src/main.rs:82-86.)"""
import numpy as np
import pytest
import torch

from llm_consensus_amd.models.config import get_preset
from llm_consensus_amd.models.loader import LazyTensor, PinnedStager, SafetensorsIndex, load_model, write_random_checkpoint


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    cfg = get_preset("tiny-llama")
    return cfg, write_random_checkpoint(cfg, str(tmp_path_factory.mktemp("ck") / "tiny"), seed=3, std=0.1,
                                        shard_bytes=1 << 16)


def test_lazy_tensor_reads_only_what_is_indexed(ckpt):
    from safetensors import safe_open

    cfg, path = ckpt
    idx = SafetensorsIndex(str(path))
    assert len(set(idx.where.values())) > 1                       # several shard files
    name = "model.layers.1.mlp.down_proj.weight"
    lazy = idx.get(name)
    assert isinstance(lazy, LazyTensor)
    with safe_open(str(path / idx.where[name]), framework="pt", device="cpu") as h:
        full = h.get_tensor(name)
    assert lazy.shape == tuple(full.shape)
    assert torch.equal(lazy[3:9], full[3:9])
    assert torch.equal(lazy[:, 5:11], full[:, 5:11])
    assert torch.equal(lazy[slice(0, 4)], full[0:4])
    assert torch.equal(lazy.to(torch.float32), full.float())
    with pytest.raises(KeyError):
        idx.get("no.such.tensor")


def test_stager_is_a_passthrough_on_cpu():
    st = PinnedStager("cpu")
    t = torch.arange(12).view(3, 4)
    assert st.to_device(t) is t and st.bytes_moved == 0


def test_tp_shards_loaded_lazily_match_slices_of_the_full_model(ckpt):
    cfg, path = ckpt
    full = load_model(str(path), "cpu")
    shards = [load_model(str(path), "cpu", tp_rank=r, tp_size=2) for r in range(2)]
    ids = torch.tensor([5, 9, 2, 7], dtype=torch.int32)
    # vocab-parallel lm_head: the two shards' rows concatenate to the full head
    lo = [s.vocab_lo for s in shards]
    hi = [s.vocab_hi for s in shards]
    assert lo[0] == 0 and hi[0] == lo[1] and hi[1] == cfg.vocab_size
    assert torch.equal(shards[0].embed, full.embed) and torch.equal(shards[1].final_norm, full.final_norm)
    assert shards[0].layers[0].qkv.n + shards[1].layers[0].qkv.n == full.layers[0].qkv.n
    assert shards[0].layers[0].down.k * 2 == full.layers[0].down.k
    assert np.isfinite(full.embed[ids.long()].float().numpy()).all()


@pytest.mark.gpu
def test_pinned_staging_moves_exact_bytes_and_alternates_buffers():
    st = PinnedStager("cuda")
    ts = [torch.randn(257, 33 + i).to(torch.bfloat16) for i in range(5)]
    outs = [st.to_device(t) for t in ts]
    torch.cuda.synchronize()
    for t, o in zip(ts, outs):
        assert o.is_cuda and torch.equal(o.cpu(), t)
    assert st.bytes_moved == sum(t.numel() * 2 for t in ts)
    assert st._bufs[0] is not None and st._bufs[1] is not None and st._bufs[0].nbytes >= 257 * 37 * 2


@pytest.mark.gpu
def test_load_model_on_gpu_equals_cpu_load(ckpt):
    cfg, path = ckpt
    g = load_model(str(path), "cuda")
    c = load_model(str(path), "cpu")
    assert torch.equal(g.embed.cpu(), c.embed)
    assert torch.equal(g.layers[1].ln2.cpu(), c.layers[1].ln2)
    tok = torch.tensor([1, 2, 3], dtype=torch.int32)
    assert g.lm_head.n == c.lm_head.n and g.layers[0].gate_up.k == c.layers[0].gate_up.k


def test_rope_scaling_llama3_matches_closed_form_and_unknown_rules_raise(tmp_path):
    """Llama-3.1-style rope_scaling: wavelengths beyond original_max/low_freq_factor are divided
    by `factor`, those below original_max/high_freq_factor are kept, the band between blends
    linearly in original_max/wavelength. Unknown rules and model types are refused."""
    import json
    import math

    import pytest
    import torch

    from llm_consensus_amd.models.config import from_hf_config, get_preset, to_hf_config
    from llm_consensus_amd.ops.reference import rope_inv_freq, rope_tables

    sc = {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
          "original_max_position_embeddings": 8192}
    theta, Dh = 5e5, 128
    got = rope_inv_freq(Dh, theta, sc)
    want = []
    for i in range(0, Dh, 2):
        f = 1.0 / theta ** (i / Dh)
        wl = 2 * math.pi / f
        if wl < 8192 / 4.0:
            want.append(f)
        elif wl > 8192 / 1.0:
            want.append(f / 8.0)
        else:
            sm = (8192 / wl - 1.0) / (4.0 - 1.0)
            want.append((1 - sm) * f / 8.0 + sm * f)
    assert torch.allclose(got, torch.tensor(want, dtype=torch.float64), rtol=1e-12)
    assert (got[:20] == rope_inv_freq(Dh, theta)[:20]).all() and got[-1] == rope_inv_freq(Dh, theta)[-1] / 8.0
    cos, _ = rope_tables(Dh, 64, theta, scaling=sc)
    assert abs(float(cos[63, -1]) - math.cos(63 * want[-1])) < 1e-6
    assert torch.allclose(rope_inv_freq(Dh, theta, {"type": "linear", "factor": 2.0}), rope_inv_freq(Dh, theta) / 2)

    hf = to_hf_config(get_preset("tiny-llama"))
    hf["rope_scaling"] = sc
    (tmp_path / "config.json").write_text(json.dumps(hf))
    cfg = from_hf_config(str(tmp_path))
    assert cfg.rope_scaling_dict["factor"] == 8.0 and to_hf_config(cfg)["rope_scaling"]["rope_type"] == "llama3"
    hf["rope_scaling"] = {"rope_type": "yarn", "factor": 4.0}
    (tmp_path / "config.json").write_text(json.dumps(hf))
    with pytest.raises(ValueError, match="yarn"):
        from_hf_config(str(tmp_path))
    hf.pop("rope_scaling")
    hf["model_type"] = "gpt_neox"
    (tmp_path / "config.json").write_text(json.dumps(hf))
    with pytest.raises(ValueError, match="gpt_neox"):
        from_hf_config(str(tmp_path))
