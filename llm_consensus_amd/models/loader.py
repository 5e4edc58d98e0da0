"""Checkpoint IO: Hugging Face safetensors -> HBM, and a random-init checkpoint writer.

``load_model`` memory-maps the ``*.safetensors`` shards of a checkpoint directory
(``safetensors.safe_open`` executes nothing from the file). Tensors are handed to the model
as ``LazyTensor``s: indexing one reads only the requested rows / columns from the mapping, so
a tensor- or expert-parallel rank touches only its own shard of the file. On a GPU every
slice goes through a pair of pinned staging buffers (the host copy of slice i+1 overlaps the
DMA of slice i) and arrives as a device tensor; the model then builds the device layout
(fused QKV, interleaved gate/up, MFMA fragment order) on the GPU.

The reference has no checkpoints (its model is a remote API,
/root/reference/src/main.rs:82-86); "checkpoint/resume" for this engine means loading
standard HF weights.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Callable, Dict, Optional

import torch

from .config import ModelConfig, from_hf_config, get_preset, to_hf_config
from .transformer import TransformerModel, hf_param_names, hf_param_shape


class PinnedStager:
    """Host -> device copies through two pinned buffers used alternately. A buffer is reused
    only after the copy that last read it has finished (one event per buffer), so filling
    buffer B on the CPU overlaps the DMA out of buffer A. Buffers, events and copies are the
    device runtime's (runtime.py: pinned arenas, hipMemcpyAsync)."""

    def __init__(self, device) -> None:
        self.device = torch.device(device)
        self._bufs = [None, None]        # numpy uint8 views of pinned arena memory
        self._events = [None, None]
        self._busy = [False, False]
        self._turn = 0
        self.bytes_moved = 0

    def to_device(self, host: torch.Tensor) -> torch.Tensor:
        if self.device.type != "cuda":
            return host
        from .. import runtime as rt

        host = host.contiguous()
        n = host.numel() * host.element_size()
        i = self._turn
        self._turn ^= 1
        if self._busy[i]:
            self._events[i].synchronize()
            self._busy[i] = False
        dev = rt.indexed(self.device)
        with torch.cuda.device(dev):
            if self._bufs[i] is None or self._bufs[i].nbytes < n:
                cap = max(n, 1 << 20)
                self._bufs[i] = rt.Arena(cap + 256, pinned=True).array((cap,), "uint8")
            if self._events[i] is None:
                self._events[i] = rt.Event()
            stage = self._bufs[i]
            # bf16 has no numpy dtype: move raw bytes
            stage[:n] = host.reshape(-1).view(torch.uint8).numpy() if n else stage[:0]
            out = torch.empty(host.shape, dtype=host.dtype, device=dev)
            rt.copy_async(out.data_ptr(), stage.ctypes.data, n, rt.H2D,
                          torch.cuda.current_stream(dev).cuda_stream)
            self._events[i].record(torch.cuda.current_stream(dev).cuda_stream)
            self._busy[i] = True
        self.bytes_moved += n
        return out


class LazyTensor:
    """A tensor of a safetensors file that has not been read yet. ``t[a:b]`` / ``t[:, a:b]``
    read just that part of the mapping; ``t.to(dtype)`` reads all of it. Results are device
    tensors when a stager for a GPU is attached."""

    def __init__(self, handle, name: str, stager: Optional[PinnedStager] = None) -> None:
        self._h, self.name, self._stager = handle, name, stager
        self._slice = handle.get_slice(name)

    @property
    def shape(self):
        return tuple(self._slice.get_shape())

    def _out(self, t: torch.Tensor) -> torch.Tensor:
        return self._stager.to_device(t) if self._stager is not None else t

    def __getitem__(self, idx) -> torch.Tensor:
        return self._out(self._slice[idx])

    def materialize(self) -> torch.Tensor:
        return self._out(self._h.get_tensor(self.name))

    def to(self, *args, **kwargs) -> torch.Tensor:
        return self.materialize().to(*args, **kwargs)


class SafetensorsIndex:
    """name -> shard file lookup over a checkpoint directory, with lazy per-tensor reads."""

    def __init__(self, path: str, stager: Optional[PinnedStager] = None) -> None:
        self.stager = stager
        from safetensors import safe_open

        self.dir = Path(path)
        self._open = safe_open
        self._handles: Dict[str, object] = {}
        idx = self.dir / "model.safetensors.index.json"
        self.where: Dict[str, str] = {}
        if idx.exists():
            self.where = dict(json.loads(idx.read_text())["weight_map"])
        else:
            for f in sorted(self.dir.glob("*.safetensors")):
                with safe_open(str(f), framework="pt", device="cpu") as h:
                    for k in h.keys():
                        self.where[k] = f.name
        if not self.where:
            raise FileNotFoundError(f"no safetensors weights under {self.dir}")

    def get(self, name: str) -> LazyTensor:
        fname = self.where.get(name)
        if fname is None:
            raise KeyError(f"tensor {name!r} not found in {self.dir}")
        h = self._handles.get(fname)
        if h is None:
            h = self._open(str(self.dir / fname), framework="pt", device="cpu")
            self._handles[fname] = h
        return LazyTensor(h, name, self.stager)

    def __contains__(self, name: str) -> bool:
        return name in self.where


def load_model(path: str, device, tp_rank: int = 0, tp_size: int = 1, comm=None,
               cfg: Optional[ModelConfig] = None, ep_size: int = 1, ep_mode: str = "allreduce") -> TransformerModel:
    cfg = cfg or from_hf_config(path)
    model = TransformerModel(cfg, device, tp_rank, tp_size, comm, ep_size, ep_mode)
    index = SafetensorsIndex(path, PinnedStager(model.device))
    model.set_weights_from_hf(index.get)
    return model


def write_random_checkpoint(cfg: ModelConfig, path: str, seed: int = 0, std: float = 0.02,
                            shard_bytes: int = 1 << 30) -> Path:
    """Writes a HF-format checkpoint (config.json + sharded safetensors + index) with
    N(0, std) weights and unit norm gains. Test helper and offline stand-in for real
    checkpoints, which cannot be downloaded in this environment."""
    from safetensors.torch import save_file

    out = Path(path)
    out.mkdir(parents=True, exist_ok=True)
    (out / "config.json").write_text(json.dumps(to_hf_config(cfg), indent=1))
    gen = torch.Generator().manual_seed(seed)
    shards, cur, cur_bytes = [], {}, 0
    for name in hf_param_names(cfg):
        shape = hf_param_shape(cfg, name)
        if "norm" in name:
            t = 1.0 + 0.05 * torch.randn(shape, generator=gen)
        else:
            t = torch.randn(shape, generator=gen) * std
        t = t.to(torch.bfloat16)
        nbytes = t.numel() * 2
        if cur and cur_bytes + nbytes > shard_bytes:
            shards.append(cur)
            cur, cur_bytes = {}, 0
        cur[name] = t
        cur_bytes += nbytes
    if cur:
        shards.append(cur)
    weight_map = {}
    for i, sh in enumerate(shards):
        fname = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, str(out / fname), metadata={"format": "pt"})
        for k in sh:
            weight_map[k] = fname
    (out / "model.safetensors.index.json").write_text(json.dumps({"metadata": {}, "weight_map": weight_map}))
    return out


def build_model(spec, device, tp_rank: int = 0, tp_size: int = 1, comm=None, ep_size: int = 1,
                ep_mode: str = "allreduce") -> TransformerModel:
    """From a config.ModelSpec: a checkpoint path wins; otherwise random-init of a preset. On a
    GPU the rank's share of the weights must be free on the card before anything is allocated
    (all processes counted): see runtime.require_hbm."""
    import torch

    dev = torch.device(device)
    if dev.type == "cuda":
        from .. import runtime
        from .config import from_hf_config

        mc = from_hf_config(spec.path) if getattr(spec, "path", None) else get_preset(spec.preset or "tiny-llama")
        with torch.cuda.device(dev):
            runtime.require_hbm(mc.weight_bytes() // max(1, tp_size, ep_size), f"model {getattr(spec, 'name', '?')}")
    if getattr(spec, "path", None):
        return load_model(spec.path, device, tp_rank, tp_size, comm, ep_size=ep_size, ep_mode=ep_mode)
    cfg = get_preset(spec.preset or "tiny-llama")
    return TransformerModel.random_init(cfg, device, seed=getattr(spec, "seed", 0), tp_rank=tp_rank,
                                        tp_size=tp_size, comm=comm, ep_size=ep_size, ep_mode=ep_mode)
