"""Checkpoint IO: Hugging Face safetensors -> HBM, and a random-init checkpoint writer.

``load_model`` memory-maps the ``*.safetensors`` shards of a checkpoint directory
(``safetensors.safe_open`` executes nothing from the file), slices each tensor to this
rank's tensor-parallel shard on the host, and hands it to the model, which converts it
to device layout (fused QKV, interleaved gate/up, MFMA fragment order) on the GPU. Only
one layer's source tensors are resident on the host at a time.

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


class SafetensorsIndex:
    """name -> shard file lookup over a checkpoint directory, with lazy per-tensor reads."""

    def __init__(self, path: str) -> None:
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

    def get(self, name: str) -> torch.Tensor:
        fname = self.where.get(name)
        if fname is None:
            raise KeyError(f"tensor {name!r} not found in {self.dir}")
        h = self._handles.get(fname)
        if h is None:
            h = self._open(str(self.dir / fname), framework="pt", device="cpu")
            self._handles[fname] = h
        return h.get_tensor(name)

    def __contains__(self, name: str) -> bool:
        return name in self.where


def load_model(path: str, device, tp_rank: int = 0, tp_size: int = 1, comm=None,
               cfg: Optional[ModelConfig] = None, ep_size: int = 1, ep_mode: str = "allreduce") -> TransformerModel:
    cfg = cfg or from_hf_config(path)
    index = SafetensorsIndex(path)
    model = TransformerModel(cfg, device, tp_rank, tp_size, comm, ep_size, ep_mode)
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
    """From a config.ModelSpec: a checkpoint path wins; otherwise random-init of a preset."""
    if getattr(spec, "path", None):
        return load_model(spec.path, device, tp_rank, tp_size, comm, ep_size=ep_size, ep_mode=ep_mode)
    cfg = get_preset(spec.preset or "tiny-llama")
    return TransformerModel.random_init(cfg, device, seed=getattr(spec, "seed", 0), tp_rank=tp_rank,
                                        tp_size=tp_size, comm=comm, ep_size=ep_size, ep_mode=ep_mode)
