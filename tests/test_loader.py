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
    assert st._bufs[0] is not None and st._bufs[1] is not None and st._bufs[0].is_pinned()


@pytest.mark.gpu
def test_load_model_on_gpu_equals_cpu_load(ckpt):
    cfg, path = ckpt
    g = load_model(str(path), "cuda")
    c = load_model(str(path), "cpu")
    assert torch.equal(g.embed.cpu(), c.embed)
    assert torch.equal(g.layers[1].ln2.cpu(), c.layers[1].ln2)
    tok = torch.tensor([1, 2, 3], dtype=torch.int32)
    assert g.lm_head.n == c.lm_head.n and g.layers[0].gate_up.k == c.layers[0].gate_up.k
