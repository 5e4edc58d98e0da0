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

import time
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
    the model (None elsewhere); ``owner`` is the replica rank that publishes results.

    Results cross ranks as numbers wherever they are numbers: log-prob scores and embeddings as
    one float32 table, generated answers as one int32 table of token ids (plus one float32 table
    of their log-probs), from which every rank re-creates the text with its own copy of the
    model's tokenizer. Only the message of a failed call travels as a host object. With
    ``tokenizer`` None (backends without token ids: scripted fakes) answers travel as objects.

    Two ways of publishing: ``sync()`` after a whole phase (the wave drivers), and
    ``take_finished`` / ``complete`` request by request for the lock-step continuous driver
    (LockstepDriver), which carries them in its per-round exchange."""

    def __init__(self, name: str, local: Optional[Backend], owner: int, comm: Comm, embed_dim: int = 0,
                 device: Optional[torch.device] = None, tokenizer=None) -> None:
        self.name, self.local, self.owner, self.comm = name, local, owner, comm
        self.embed_dim = embed_dim           # width of an embedding row (0: embeddings travel as objects)
        self.device = device or torch.device("cpu")
        self.tok = tokenizer
        self._pending: List[dict] = []       # submission order; every rank of the replica holds the same list
        self._seq = 0
        self._side = None                    # side stream of the publishing broadcasts (GPU)

    def _submit(self, method: str, *args, width: int = 0, gen: Optional[GenParams] = None) -> Handle:
        h = Handle()
        inner = getattr(self.local, method)(*args) if self.local is not None else None
        self._pending.append({"seq": self._seq, "h": h, "inner": inner, "width": width, "gen": gen, "sent": False})
        self._seq += 1
        return h

    def submit_generate(self, prompt: str, params: GenParams, on_text=None) -> Handle:
        return self._submit("submit_generate", prompt, params, gen=params)   # results arrive whole, after the exchange

    def submit_score(self, prompt: str, targets: Sequence[str]) -> Handle:
        return self._submit("submit_score", prompt, list(targets), width=len(targets))

    def submit_embed(self, text: str) -> Handle:
        return self._submit("submit_embed", text, width=self.embed_dim)

    def token_ids_of(self, text: str):
        return list(self.tok.encode(text, add_bos=False)) if self.tok is not None else None

    def run_local(self) -> None:
        if self.local is not None and self._pending:
            self.local.run()

    # ------------------------------------------------------------------ wire forms
    def _text_of(self, toks: Sequence[int]) -> str:
        eos = self.tok.eos_id
        if eos is not None and len(toks) and toks[-1] == eos:
            toks = toks[:-1]
        return self.tok.decode(list(toks))

    def _gen_tables(self, entries: List[dict], results: List[Optional[tuple]], mine: bool):
        """Int32 [rows, 4 + Tmax] and float32 [rows, 1 + Cmax] for the generation calls in
        ``entries`` (one row per chain; every rank knows the shapes from the calls' parameters).
        Row: status, length, prompt tokens, cached tokens, token ids | cum log-prob, decision log-probs."""
        chains = [max(1, e["gen"].n) for e in entries]
        tmax = max(1, max(e["gen"].max_new_tokens for e in entries))
        cmax = max((len(e["gen"].decision_targets or []) for e in entries), default=0)
        it = torch.zeros(sum(chains), 4 + tmax, dtype=torch.int32)
        ft = torch.zeros(sum(chains), 1 + cmax, dtype=torch.float32)
        if mine:
            row = 0
            for e, res, n in zip(entries, results, chains):
                status, value, _ = res
                ok = status == "ok" and bool(getattr(value, "token_ids", None)) and len(value.token_ids) <= n
                for k in range(n):
                    if ok and k < len(value.token_ids):
                        ids = value.token_ids[k][:tmax]
                        it[row, 0], it[row, 1] = 1, len(ids)
                        it[row, 2], it[row, 3] = value.prompt_tokens, value.cached_tokens
                        it[row, 4:4 + len(ids)] = torch.as_tensor(ids, dtype=torch.int32)
                        ft[row, 0] = value.cum_logprobs[k] if k < len(value.cum_logprobs) else 0.0
                        if value.decision_logprobs:
                            d = value.decision_logprobs[k]
                            ft[row, 1:1 + len(d)] = torch.as_tensor(d, dtype=torch.float32)
                    row += 1
        return it, ft, chains

    def _gen_results(self, entries: List[dict], it: torch.Tensor, ft: torch.Tensor, chains: List[int]):
        """The inverse of _gen_tables on any rank: (status, GenResult | None, usage) per entry."""
        from ..consensus.backend import GenResult

        out, row = [], 0
        for e, n in zip(entries, chains):
            rows = range(row, row + n)
            row += n
            live = [r for r in rows if int(it[r, 0]) == 1]
            if not live:
                out.append(None)                       # failed (or not expressible): the object path carries it
                continue
            ids = [it[r, 4:4 + int(it[r, 1])].tolist() for r in live]
            C = len(e["gen"].decision_targets or [])
            res = GenResult(texts=[self._text_of(t) for t in ids], token_ids=ids,
                            cum_logprobs=[float(ft[r, 0]) for r in live],
                            decision_logprobs=[ft[r, 1:1 + C].tolist() for r in live] if C else None,
                            prompt_tokens=int(it[live[0], 2]), cached_tokens=int(it[live[0], 3]))
            out.append(("ok", res, (res.prompt_tokens, sum(len(t) for t in ids), res.cached_tokens)))
        return out

    def _broadcast_tables(self, tables: List[torch.Tensor]) -> List[torch.Tensor]:
        """Owner -> replica, on the communicator's own transport. On a GPU the copies and the
        collectives are queued on a side stream (never behind an engine's step) and the host
        waits for that stream's event only."""
        if self.device.type != "cuda":
            return [self.comm.broadcast(t, self.owner) for t in tables]
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self._side):
            devs = [self.comm.broadcast(t.to(self.device, non_blocking=True), self.owner) for t in tables]
            hosts = [d.to("cpu", non_blocking=True) for d in devs]
            done = torch.cuda.Event()
            done.record(self._side)
        done.synchronize()
        return hosts

    def sync(self) -> None:
        """Publishes the results of the queued calls to every rank of the replica (see the class
        comment for the wire forms). A phase without failures sends no pickled payload at all."""
        pend, self._pending = self._pending, []
        if not pend:
            return
        mine = self.comm.rank == self.owner
        numeric = [i for i, e in enumerate(pend) if e["width"] > 0]
        gens = [i for i, e in enumerate(pend) if e["gen"] is not None and self.tok is not None]
        results: List[Optional[tuple]] = [None] * len(pend)
        if mine:
            for i, e in enumerate(pend):
                try:
                    results[i] = ("ok", e["inner"].result(), e["inner"].usage)
                except BackendError as err:
                    results[i] = ("err", str(err), None)
        if self.comm.world > 1:
            with span(f"publish:{self.name}", "parallel", requests=len(pend)):
                tables: List[torch.Tensor] = []
                wmax = 0
                if numeric:
                    wmax = max(pend[i]["width"] for i in numeric)
                    table = torch.zeros(len(numeric), wmax + 4, dtype=torch.float32)
                    if mine:
                        for row, i in enumerate(numeric):
                            status, value, usage = results[i]
                            if status == "ok":
                                table[row, :pend[i]["width"]] = torch.as_tensor(value, dtype=torch.float32).reshape(-1)
                                table[row, wmax:wmax + 3] = torch.tensor([float(u) for u in (usage or (0, 0, 0))])
                                table[row, wmax + 3] = 1.0
                    tables.append(table)
                chains: List[int] = []
                if gens:
                    it, ft, chains = self._gen_tables([pend[i] for i in gens], [results[i] for i in gens], mine)
                    tables += [it, ft]
                if tables:
                    tables = self._broadcast_tables(tables)
                settled = set()
                if numeric:
                    table = tables[0]
                    for row, i in enumerate(numeric):
                        if table[row, wmax + 3] != 0:
                            usage = tuple(int(u) for u in table[row, wmax:wmax + 3].tolist())
                            results[i] = ("ok", table[row, :pend[i]["width"]].numpy().copy(), usage)
                            settled.add(i)
                if gens:
                    got = self._gen_results([pend[i] for i in gens], tables[-2], tables[-1], chains)
                    for i, r in zip(gens, got):
                        if r is not None:
                            results[i] = r
                            settled.add(i)
                rest = [i for i in range(len(pend)) if i not in settled]
                if rest:
                    payload = self.comm.broadcast_object([results[i] for i in rest] if mine else None, self.owner)
                    for i, r in zip(rest, payload):
                        results[i] = r
        for e, (status, value, usage) in zip(pend, results):  # type: ignore[misc]
            self._complete(e, status, value, usage)

    @staticmethod
    def _complete(e: dict, status: str, value, usage) -> None:
        h = e["h"]
        h.usage = tuple(usage) if usage else None
        if status == "ok":
            h.set_result(value)
        else:
            h.set_error(BackendError(value))

    def run(self) -> None:
        self.run_local()
        self.sync()

    # ------------------------------------------------------------------ request-by-request publishing
    def take_finished(self) -> List[dict]:
        """Owner rank: wire records of the calls whose local handle has completed since the last
        call (plain numbers, lists and strings; see ``complete``)."""
        out = []
        for e in self._pending:
            if e["sent"] or e["inner"] is None or not e["inner"].done:
                continue
            e["sent"] = True
            rec = {"seq": e["seq"]}
            try:
                v = e["inner"].result()
                rec["usage"] = list(e["inner"].usage or ())
                if e["gen"] is not None and getattr(v, "token_ids", None) and self.tok is not None:
                    rec.update(kind="gen", ids=[list(map(int, t)) for t in v.token_ids],
                               lp=[float(x) for x in v.cum_logprobs],
                               dec=[[float(x) for x in d] for d in v.decision_logprobs] if v.decision_logprobs else None,
                               pt=int(v.prompt_tokens), ct=int(v.cached_tokens), fr=list(v.finish_reasons or ()))
                elif e["gen"] is not None:
                    rec.update(kind="text", texts=list(v.texts),
                               dec=[[float(x) for x in d] for d in v.decision_logprobs] if v.decision_logprobs else None)
                else:
                    rec.update(kind="num", row=[float(x) for x in v])
            except BackendError as err:
                rec.update(kind="err", msg=str(err))
            out.append(rec)
        return out

    def complete(self, rec: dict) -> None:
        """Any rank of the replica: finishes the call a wire record names."""
        from ..consensus.backend import GenResult

        e = next((x for x in self._pending if x["seq"] == rec["seq"]), None)
        if e is None:
            return
        self._pending.remove(e)
        usage = tuple(rec.get("usage") or ()) or None
        kind = rec["kind"]
        if kind == "err":
            return self._complete(e, "err", rec["msg"], None)
        if kind == "gen":
            v = GenResult(texts=[self._text_of(t) for t in rec["ids"]], token_ids=rec["ids"], cum_logprobs=rec["lp"],
                          decision_logprobs=rec["dec"], prompt_tokens=rec["pt"], cached_tokens=rec["ct"],
                          finish_reasons=list(rec.get("fr") or ()))
        elif kind == "text":
            v = GenResult(texts=rec["texts"], decision_logprobs=rec["dec"])
        else:
            import numpy as np

            v = np.asarray(rec["row"], dtype=np.float32)
        self._complete(e, "ok", v, usage)

    @property
    def outstanding(self) -> int:
        return len(self._pending)

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
        from ..utils.tokenizer import make_tokenizer

        for name in sorted(lay.owners):
            local = factory(cfg.models[name], lay.hosted[name]) if name in lay.hosted else None
            mc = model_config_of(cfg.models[name])
            # every rank of the replica holds the model's tokenizer: answers cross ranks as token ids
            tok = make_tokenizer(cfg.models[name].tokenizer, mc.vocab_size) if backend_factory is None else None
            self.backends[name] = ShardedBackend(name, local, lay.owners[name], lay.replica_comm,
                                                 embed_dim=mc.hidden_size, device=self.device, tokenizer=tok)
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


def _pack(obj) -> bytes:
    """Wire form of a round's record: msgpack (numbers, strings, lists, maps; nothing executable)."""
    import msgpack

    return msgpack.packb(obj, use_bin_type=True, default=str)


def _unpack(data: bytes):
    import msgpack

    return msgpack.unpackb(data, raw=False, strict_map_key=False)


class LockstepDriver:
    """Continuous serving for an ensemble that spans GPUs: every rank runs the SAME session pool
    (protocol.SessionPool, strategies.VotePool / PhasedPool) over the replica's ShardedBackends
    and the ranks advance in rounds. One round, on every rank:

      1. step the engines this rank hosts for about ``quantum_s`` (decode steps run ahead of
         their results, so the GPU keeps working through the rest of the round);
      2. ONE exchange over the whole group (``Comm.all_gather_bytes``): rank 0 contributes the
         questions that arrived since the last round (ticket, text) and the stop flag, every rank
         the wire records of the member calls its engines finished (ShardedBackend.take_finished),
         the first rank of every replica the answers its pool produced;
      3. apply what came in, in rank order: complete the handles of the own replica's calls,
         admit the new questions assigned to the own replica (ticket modulo replicas), then let
         the pool advance its sessions (which queues their next member calls on the engines).

    All ranks of a replica see the same completions in the same order, so their pools stay in
    step without exchanging decisions; a question admitted in round r is decoding next to the
    ones already in flight from round r + 1 on, and every finished answer reaches rank 0 one
    round after its last member call ended. The reference consumes its members' replies as
    they arrive (/root/reference/src/main.rs:259-291), one question at a time; this is that
    loop for a node."""

    def __init__(self, ens: "DistributedEnsemble", quantum_s: float = 0.02) -> None:
        from ..consensus.backend import BackendStepper

        self.ens, self.comm, self.quantum_s = ens, ens.comm, quantum_s
        lay = ens.layout
        self.replica, self.num_replicas = lay.replica, lay.num_replicas
        self.lead = lay.replica >= 0 and lay.replica_comm.rank == 0
        self.pool = None
        self._local = None
        if lay.replica >= 0:
            if not callable(getattr(ens.strategy, "pool", None)):
                raise RuntimeError(f"strategy {type(ens.strategy).__name__} has no continuous driver")
            self.pool = ens.strategy.pool()
            self.pool.stepper = _NoStep()                      # the rounds below advance the engines
            hosted = [b.local for b in ens.backends.values() if b.local is not None]
            self._local = BackendStepper(hosted)
            # A rank that holds a shard of a tensor / expert parallel model advances its engines
            # exactly once per round: the ranks of such a group must launch the same steps (their
            # kernels wait for each other), and a rank that ran ahead on its own clock would block
            # on a step its peer only launches after the exchange both of them have to join.
            self._sharded = any(getattr(getattr(e, "model", None), "tp_size", 1) > 1
                                or getattr(getattr(e, "model", None), "ep_size", 1) > 1
                                for b in hosted for e in ([b.engine] if hasattr(b, "engine") else []))
        if self.pool is None:
            self._sharded = False
        self._ticket_of: Dict[int, int] = {}                   # pool ticket -> global ticket
        self._answers: List[tuple] = []                        # (global ticket, ConsensusResult) awaiting the exchange
        self.rounds = 0
        self.stopped = False

    def _step_engines(self) -> None:
        if self._local is None:
            return
        t_end = time.perf_counter() + self.quantum_s
        while any(b.outstanding for b in self.ens.backends.values() if b.local is not None):
            if not self._local.step():
                break                                          # everything in flight is still running
            if self._sharded or time.perf_counter() >= t_end:
                break

    def round(self, new_questions: Sequence[tuple] = (), stop: bool = False, control: Optional[dict] = None) -> List[tuple]:
        """One lock-step round. ``new_questions`` [(ticket, text)], ``stop`` and ``control`` count
        on rank 0 only (``control`` = {"op": "warm_up", "kw": {...}}: every rank runs that
        ensemble-wide operation at the end of the round). Returns the (ticket, result record)
        pairs that reached this rank in the exchange (every rank gets all of them; rank 0 is the
        one that hands them to callers)."""
        with span("lockstep.step", "parallel"):
            self._step_engines()
        mine: Dict[str, object] = {}
        if self.comm.rank == 0:
            mine["new"] = [[int(t), str(q)] for t, q in new_questions]
            mine["stop"] = bool(stop)
            if control:
                mine["ctl"] = control
        if self.replica >= 0:
            done = {name: b.take_finished() for name, b in self.ens.backends.items()
                    if b.local is not None and b.comm.rank == b.owner}
            done = {k: v for k, v in done.items() if v}
            if done:
                mine["done"] = [self.replica, done]
            if self.lead and self._answers:
                mine["answers"] = [[t, _result_record(r)] for t, r in self._answers]
                self._answers = []
        with span("publish:round", "parallel", bytes=0):
            parts = [(_unpack(p) if p else {}) for p in self.comm.all_gather_bytes(_pack(mine))]
        self.rounds += 1
        finished: List[tuple] = []
        for part in parts:                                     # rank order: the same on every rank
            for t, rec in part.get("answers", []):
                finished.append((int(t), rec))
            d = part.get("done")
            if d and int(d[0]) == self.replica:
                for name in sorted(d[1]):
                    for rec in d[1][name]:
                        self.ens.backends[name].complete(rec)
        head = parts[0]
        if self.pool is not None:
            for t, q in head.get("new", []):
                if int(t) % self.num_replicas == self.replica:
                    self._ticket_of[self.pool.submit(q)] = int(t)
            for pt, res in self.pool.pump():
                self._answers.append((self._ticket_of.pop(pt), res))
        ctl = head.get("ctl")
        if ctl and ctl.get("op") == "warm_up":
            self.ens.warm_up(**(ctl.get("kw") or {}))
        if head.get("stop"):
            self.stopped = True
        return finished

    @property
    def busy(self) -> bool:
        """This rank still has sessions in flight or answers to report."""
        return bool(self.pool is not None and (self.pool.active or self._answers))

    def follow(self) -> int:
        """Loop of every rank but 0: rounds until rank 0 says stop. Returns the round count."""
        while not self.stopped:
            self.round()
        return self.rounds


class _NoStep:
    """Stepper of a pool whose engines are advanced by LockstepDriver's rounds."""

    @staticmethod
    def step() -> bool:
        return True


def _result_record(r) -> dict:
    """A ConsensusResult as plain data (what crosses ranks and what the front ends print)."""
    return {"question": r.question, "answer": r.answer, "strategy": r.strategy, "rounds": int(r.rounds),
            "member_calls": int(r.member_calls), "unanimous": bool(r.unanimous), "forced": bool(getattr(r, "forced", False)),
            "agreement": float(getattr(r, "agreement", 0.0) or 0.0), "proposer": getattr(r, "proposer", None),
            "latency_s": float(r.latency_s), "votes": _plain(getattr(r, "votes", None)),
            "usage": _plain(getattr(r, "usage", None)), "extra": _plain(getattr(r, "extra", None))}


def _plain(x):
    import numpy as np

    if isinstance(x, dict):
        return {str(k): _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    if isinstance(x, np.generic):
        return x.item()
    if isinstance(x, np.ndarray):
        return x.tolist()
    return x


def result_from_record(rec: dict):
    from ..consensus.protocol import ConsensusResult

    return ConsensusResult(question=rec["question"], answer=rec["answer"], strategy=rec["strategy"],
                           rounds=rec["rounds"], member_calls=rec["member_calls"], unanimous=rec["unanimous"],
                           forced=rec.get("forced", False), agreement=rec.get("agreement", 0.0),
                           proposer=rec.get("proposer"), latency_s=rec["latency_s"], votes=rec.get("votes") or [],
                           usage=rec.get("usage") or {}, extra=rec.get("extra") or {})


class LockstepEnsemble:
    """What the front ends hold on rank 0 of a distributed ensemble served continuously: the
    ``ask`` / ``ask_many`` / ``warm_up`` / ``stats`` / ``close`` of app.Ensemble, carried out by
    lock-step rounds (the other ranks sit in ``LockstepDriver.follow``). ``submit`` / ``poll``
    are the non-blocking form the HTTP server's driver loop uses."""

    def __init__(self, dist_ensemble: DistributedEnsemble, quantum_s: float = 0.02) -> None:
        self.inner = dist_ensemble
        self.config = dist_ensemble.config
        self.members = getattr(dist_ensemble, "members", [])
        self.backends: Dict[str, Backend] = {}       # member models are not individually addressable here
        self.driver = LockstepDriver(dist_ensemble, quantum_s)
        self._next = 0
        self._new: List[tuple] = []
        self.in_flight = 0

    def submit(self, question: str) -> int:
        t = self._next
        self._next += 1
        self._new.append((t, question))
        self.in_flight += 1
        return t

    def poll(self) -> List[tuple]:
        """One round; returns [(ticket, ConsensusResult)] of the questions answered in it."""
        new, self._new = self._new, []
        out = [(t, result_from_record(rec)) for t, rec in self.driver.round(new)]
        self.in_flight -= len(out)
        return out

    def ask_many(self, questions: Sequence[str]) -> List[Any]:
        tickets = [self.submit(q) for q in questions]
        got: Dict[int, Any] = {}
        while any(t not in got for t in tickets):
            got.update(dict(self.poll()))
        return [got[t] for t in tickets]

    def ask(self, question: str):
        return self.ask_many([question])[0]

    def warm_up(self, **kw):
        self.driver.round(control={"op": "warm_up", "kw": kw})
        return {}

    def stats(self) -> Dict[str, Dict[str, float]]:
        return self.inner.stats()

    def record(self, results) -> None:
        """Ensemble-level counters hook of the server's continuous loop (kept per engine here)."""

    def close(self) -> None:
        while self.in_flight > 0:
            self.poll()
        self.driver.round(stop=True)


def serve_distributed(cfg, comm: Comm, device, continuous: bool = True):
    """Builds the distributed ensemble on this rank. Rank 0 gets the front-end object
    (LockstepEnsemble when the strategy can be served continuously, LeaderEnsemble otherwise);
    every other rank serves until rank 0 closes and gets None."""
    dist_ens = DistributedEnsemble(cfg, comm, device)
    lockstep = continuous and callable(getattr(dist_ens.strategy, "pool", None)) if dist_ens.strategy is not None \
        else continuous
    # ranks that are idle under the plan have no strategy; rank 0 decides for everybody
    lockstep = bool(comm.broadcast_object(lockstep if comm.rank == 0 else None, 0))
    if comm.rank != 0:
        if lockstep:
            LockstepDriver(dist_ens).follow()
        else:
            follow(dist_ens)
        return None
    return LockstepEnsemble(dist_ens) if lockstep else LeaderEnsemble(dist_ens)


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
    """Process-group bring-up from the launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*): RCCL ("nccl") with one GPU per process when GPUs are present, gloo otherwise.
    The collective timeout is a day: a front end's ranks wait in a collective while rank 0 waits
    for the next question. LC_SHARE_GPU=1 is the rehearsal mode of a one-GPU machine: every rank
    uses GPU 0, the process group is gloo with host-staged tensors (RCCL refuses two ranks on one
    device) and model groups still use the peer-to-peer kernels over IPC-mapped memory.
    Returns (comm, device)."""
    import os
    from datetime import timedelta

    import torch.distributed as dist

    from .comm import make_comm

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    use_gpu = (device_type or ("cuda" if torch.cuda.is_available() else "cpu")) == "cuda"
    if use_gpu and os.environ.get("LC_SHARE_GPU", "0") == "1":
        torch.cuda.set_device(0)
        device = torch.device("cuda", 0)
        dist.init_process_group("gloo", timeout=timedelta(hours=24))
    elif use_gpu:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=device, timeout=timedelta(hours=24))
    else:
        device = torch.device("cpu")
        dist.init_process_group("gloo", timeout=timedelta(hours=24))
    return make_comm(dist, device), device
