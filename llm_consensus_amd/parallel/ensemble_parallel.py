"""Ensemble parallelism: members of one ensemble on different GPUs, whole ensembles
replicated over the rest of the node.

In the reference every persona calls the same remote model, members answer concurrently
only during evaluation (fan-out /root/reference/src/main.rs:250-253) and one at a time for
proposals and refinements (main.rs:229, :272). Here a member is a model resident on a GPU
(or a tensor/expert-parallel group of GPUs), so *where* members live is the first
parallelism axis:

    node GPUs = replica 0 | replica 1 | ...           questions are sharded over replicas
    replica   = model group A | model group B | ...   members run concurrently
    model group = 1 GPU, or tp/ep GPUs in lock step

Execution is SPMD inside a replica: every rank runs the same strategy code on the same
questions, so all ranks submit the same requests in the same order. A ``ShardedBackend``
stands in for each model on every rank of the replica: on ranks that host the model it
forwards to the local engine, and after the compute phase the group's first rank
publishes the results (texts, log-probs, embeddings: a few KB) to the replica with one
host-object broadcast. Because all compute phases start before any exchange
(consensus.backend.run_backends), models on different GPUs generate at the same time.
Strategy-level randomness (proposer choice, refiner choice) is seeded per session, so the
ranks stay in step without exchanging decisions.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import torch

from ..consensus.backend import Backend, BackendError, GenParams, Handle
from ..utils.log import get_logger
from ..utils.trace import span
from .comm import Comm, SingleComm
from .placement import plan_for_config
from .replica import shard_bounds

log = get_logger("consensus.parallel")


class ShardedBackend(Backend):
    """Replica-wide facade of one model. ``local`` is the engine backend on ranks that host
    the model (None elsewhere); ``owner`` is the replica rank that publishes results."""

    def __init__(self, name: str, local: Optional[Backend], owner: int, comm: Comm, embed_dim: int = 0,
                 device: Optional[torch.device] = None) -> None:
        self.name, self.local, self.owner, self.comm = name, local, owner, comm
        self.embed_dim = embed_dim           # width of an embedding row (0: embeddings travel as objects)
        self.device = device or torch.device("cpu")
        self._pending: List[tuple] = []      # (handle, inner handle | None, numeric width | 0)

    def _submit(self, method: str, *args, width: int = 0) -> Handle:
        h = Handle()
        inner = getattr(self.local, method)(*args) if self.local is not None else None
        self._pending.append((h, inner, width))
        return h

    def submit_generate(self, prompt: str, params: GenParams, on_text=None) -> Handle:
        return self._submit("submit_generate", prompt, params)     # results arrive whole, after the exchange

    def submit_score(self, prompt: str, targets: Sequence[str]) -> Handle:
        return self._submit("submit_score", prompt, list(targets), width=len(targets))

    def submit_embed(self, text: str) -> Handle:
        return self._submit("submit_embed", text, width=self.embed_dim)

    def run_local(self) -> None:
        if self.local is not None and self._pending:
            self.local.run()

    def sync(self) -> None:
        """Publishes the results of the queued calls to every rank of the replica. Log-prob
        scores and embeddings are rows of numbers whose widths every rank knows, so they travel
        as ONE float32 tensor broadcast on the communicator's own transport (device memory under
        RCCL): columns [0, width) hold the row, the next three the token usage, the last one a
        status flag. Generated text, and the message of a failed call, go through one host-object
        broadcast; a phase that only scored or embedded sends no pickled payload at all."""
        pend, self._pending = self._pending, []
        if not pend:
            return
        mine = self.comm.rank == self.owner
        numeric = [i for i, (_, _, w) in enumerate(pend) if w > 0]
        objects = [i for i, (_, _, w) in enumerate(pend) if w <= 0]
        results: List[Optional[tuple]] = [None] * len(pend)
        if mine:
            for i, (_, inner, _) in enumerate(pend):
                try:
                    results[i] = ("ok", inner.result(), inner.usage)
                except BackendError as e:
                    results[i] = ("err", str(e), None)
        if self.comm.world > 1:
            with span(f"publish:{self.name}", "parallel", requests=len(pend)):
                if numeric:
                    wmax = max(pend[i][2] for i in numeric)
                    table = torch.zeros(len(numeric), wmax + 4, dtype=torch.float32)
                    if mine:
                        for row, i in enumerate(numeric):
                            status, value, usage = results[i]
                            if status == "ok":
                                table[row, :pend[i][2]] = torch.as_tensor(value, dtype=torch.float32).reshape(-1)
                                table[row, wmax:wmax + 3] = torch.tensor([float(u) for u in (usage or (0, 0, 0))])
                                table[row, wmax + 3] = 1.0
                    table = self.comm.broadcast(table.to(self.device), self.owner).cpu()
                    failed = [i for row, i in enumerate(numeric) if table[row, wmax + 3] == 0]
                    for row, i in enumerate(numeric):
                        if table[row, wmax + 3] != 0:
                            usage = tuple(int(u) for u in table[row, wmax:wmax + 3].tolist())
                            results[i] = ("ok", table[row, :pend[i][2]].numpy().copy(), usage)
                else:
                    failed = []
                rest = objects + failed
                if rest:
                    payload = self.comm.broadcast_object([results[i] for i in rest] if mine else None, self.owner)
                    for i, r in zip(rest, payload):
                        results[i] = r
        for (h, _, _), (status, value, usage) in zip(pend, results):  # type: ignore[misc]
            h.usage = usage
            if status == "ok":
                h.set_result(value)
            else:
                h.set_error(BackendError(value))

    def run(self) -> None:
        self.run_local()
        self.sync()

    def engines(self) -> list:
        return self.local.engines() if self.local is not None else []

    def warm_up(self, **kw) -> Dict[str, float]:
        """Ranks that host the model warm their engine up (together: a sharded model's warm-up
        steps contain the group's collectives); the others have nothing to do."""
        return self.local.warm_up(**kw) if self.local is not None else {}

    def stats(self) -> Dict[str, float]:
        return self.local.stats() if self.local is not None else {}


@dataclass
class RankLayout:
    replica: int                         # -1: this rank is idle under the plan
    num_replicas: int
    replica_comm: Optional[Comm]
    hosted: Dict[str, Comm]              # model name -> communicator of its group (this rank is a member)
    owners: Dict[str, int]               # model name -> replica rank that publishes results
    devices_of: Dict[str, List[int]] = field(default_factory=dict)   # model name -> ranks of its group (this rank's replica)


def build_layout(cfg, comm: Comm, plan=None) -> RankLayout:
    """Creates the replica and model-group communicators for this rank. Collective over
    ``comm``: every rank walks the plan in the same order."""
    plan = plan or plan_for_config(cfg, comm.world)
    if not plan.feasible:
        raise RuntimeError(f"placement failed: {plan.error}")
    model_of = {m.name: cfg.models[m.model].name for m in cfg.members}
    mine = RankLayout(-1, len(plan.replicas), None, {}, {})
    for ri, rep in enumerate(plan.replicas):
        devices = list(rep.devices)
        rc = comm.subgroup(devices)
        groups: Dict[str, List[int]] = {}
        for mp in rep.members:
            groups.setdefault(model_of[mp.name], list(mp.devices))
        here = comm.rank in devices
        if here:
            mine.replica, mine.replica_comm = ri, rc if rc is not None else SingleComm()
        for name in sorted(groups):
            gc = comm.subgroup(groups[name])
            if here:
                mine.devices_of[name] = list(groups[name])
                mine.owners[name] = devices.index(groups[name][0])
                if comm.rank in groups[name]:
                    mine.hosted[name] = gc if gc is not None else SingleComm()
    return mine


class DistributedEnsemble:
    """What ``app.Ensemble`` is for one process, for a whole node: ``ask_many`` takes the
    global question list on every rank and returns the answers to all of them on every rank."""

    def __init__(self, cfg, comm: Comm, device: Optional[torch.device] = None, plan=None,
                 backend_factory=None) -> None:
        from ..app import build_backend, make_members, make_strategy, model_config_of

        self.config, self.comm = cfg, comm
        self.device = device or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                                 else torch.device("cpu"))
        self.layout = build_layout(cfg, comm, plan)
        self.backends: Dict[str, ShardedBackend] = {}
        self.strategy = None
        lay = self.layout
        if lay.replica < 0:
            log.info("rank %d is idle under the placement plan", comm.rank)
            return
        from .p2p import wrap_group_comm

        # A model group of 2, 4 or 8 GPUs talks through peer-to-peer kernels (one launch per
        # collective, capturable); the process-group communicator only bootstraps it.
        factory = backend_factory or (lambda spec, gc: build_backend(spec, self.device, wrap_group_comm(gc, self.device),
                                                                     gc.rank))
        for name in sorted(lay.owners):
            local = factory(cfg.models[name], lay.hosted[name]) if name in lay.hosted else None
            self.backends[name] = ShardedBackend(name, local, lay.owners[name], lay.replica_comm,
                                                 embed_dim=model_config_of(cfg.models[name]).hidden_size,
                                                 device=self.device)
        self.members = make_members(cfg, self.backends)
        judge_gpus = {d for m in cfg.members if m.role == "judge" for d in lay.devices_of.get(cfg.models[m.model].name, [])}
        other_gpus = {d for m in cfg.members if m.role != "judge" for d in lay.devices_of.get(cfg.models[m.model].name, [])}
        self.strategy = make_strategy(cfg, self.members, self.device,
                                      judges_have_own_gpus=bool(judge_gpus) and not (judge_gpus & other_gpus))

    def ask_many(self, questions: Sequence[str]) -> List[Any]:
        lay = self.layout
        local: List[Any] = []
        if lay.replica >= 0:
            lo, hi = shard_bounds(len(questions), lay.replica, lay.num_replicas)
            if hi > lo:
                local = self.strategy.ask_many(list(questions[lo:hi]))
        lead = lay.replica >= 0 and lay.replica_comm.rank == 0
        parts = self.comm.all_gather_object((lay.replica, local) if lead else None)
        by_replica = {p[0]: p[1] for p in parts if p is not None}
        out: List[Any] = []
        for r in range(lay.num_replicas):
            out += by_replica.get(r, [])
        return out

    def ask(self, question: str):
        return self.ask_many([question])[0]

    def stats(self) -> Dict[str, Dict[str, float]]:
        return {n: b.stats() for n, b in self.backends.items()}

    def warm_up(self, **kw) -> Dict[str, Dict[str, float]]:
        """Every rank warms up the engines it hosts, model by model in name order (the ranks of a
        sharded model's group meet in its collectives)."""
        return {n: self.backends[n].warm_up(**kw) for n in sorted(self.backends)}

    def collectives(self) -> Dict[str, str]:
        """Which transport every sharded model's group ended up on (peer-to-peer kernels, or the
        bootstrap communicator after a failed first-contact check, with the reason). Collective."""
        mine = {}
        for name, b in self.backends.items():
            eng = getattr(b.local, "engine", None) if b.local is not None else None
            comm = getattr(getattr(eng, "model", None), "comm", None)
            if comm is not None and comm.world > 1:
                what = getattr(comm, "collectives", type(comm).__name__)
                why = getattr(comm, "fallback_reason", None)
                mine[name] = what + (f" (peer-to-peer self-check failed: {why})" if why else "") + \
                    (", decode steps recorded as hipGraphs" if eng.use_graphs else ", eager decode steps")
        out: Dict[str, str] = {}
        for part in self.comm.all_gather_object(mine):
            out.update(part or {})
        return out


class LeaderEnsemble:
    """Front-end view of a DistributedEnsemble on rank 0: the REPL / HTTP server calls
    ``ask_many`` here, the question batch is broadcast to the follower ranks (which sit in
    ``follow``), and everyone runs the SPMD ``DistributedEnsemble.ask_many``."""

    def __init__(self, dist_ensemble: DistributedEnsemble) -> None:
        self.inner = dist_ensemble
        self.config = dist_ensemble.config
        self.members = getattr(dist_ensemble, "members", [])
        self.backends: Dict[str, Backend] = {}       # member models are not individually addressable here

    def ask_many(self, questions: Sequence[str]) -> List[Any]:
        qs = list(questions)
        self.inner.comm.broadcast_object(("ask", qs), 0)
        return self.inner.ask_many(qs)

    def ask(self, question: str):
        return self.ask_many([question])[0]

    def stats(self) -> Dict[str, Dict[str, float]]:
        return self.inner.stats()

    def warm_up(self, **kw):
        self.inner.comm.broadcast_object(("warm_up", kw), 0)
        return self.inner.warm_up(**kw)

    def close(self) -> None:
        self.inner.comm.broadcast_object(("stop", None), 0)


def follow(dist_ensemble: DistributedEnsemble) -> int:
    """Loop of every rank but 0: take part in each batch the leader announces until it
    closes. Returns the number of batches served."""
    n = 0
    while True:
        op, qs = dist_ensemble.comm.broadcast_object(None, 0)
        if op == "stop":
            return n
        if op == "warm_up":
            dist_ensemble.warm_up(**qs)
            continue
        dist_ensemble.ask_many(qs)
        n += 1


def init_distributed(device_type: Optional[str] = None):
    """Process-group bring-up from torchrun's environment: RCCL ("nccl") with one GPU per
    process when GPUs are present, gloo otherwise. Returns (comm, device)."""
    import os

    import torch.distributed as dist

    from .comm import make_comm

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    use_gpu = (device_type or ("cuda" if torch.cuda.is_available() else "cpu")) == "cuda"
    if use_gpu:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group("gloo")
    return make_comm(dist, device), device
