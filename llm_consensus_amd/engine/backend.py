"""``EngineBackend``: the text-level Backend over an in-process Engine.

This is the drop-in for the reference's ``call_gemini`` (/root/reference/src/main.rs:82-86)
that members talk to: prompts are tokenized, queued on the engine and completed by
``run()``, which steps the engine until every queued request has finished. Several
members (personas) share one EngineBackend when they share a model, so their requests
batch together and their common prompt prefixes are served from the prefix cache.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

from .._native import core
from ..consensus.backend import Backend, BackendError, GenParams, GenResult, Handle
from ..utils.log import get_logger
from ..utils.tokenizer import Tokenizer, apply_chat_format
from .engine import Engine

log = get_logger("consensus.engine.backend")


class EngineBackend(Backend):
    def __init__(self, engine: Engine, tokenizer: Tokenizer, name: Optional[str] = None,
                 stop_at_eos: bool = True, chat_format: str = "none") -> None:
        self.engine, self.tok = engine, tokenizer
        self.name = name or engine.name
        self.stop_at_eos = stop_at_eos
        apply_chat_format(chat_format, "")        # validates the name
        self.chat_format = chat_format
        self._pending: Dict[int, tuple] = {}
        self._streams: Dict[int, list] = {}
        self._stops: Dict[int, List[str]] = {}      # text-level stop sequences of pending requests
        self._discard: set = set()                  # requests finished early here whose engine result is to be dropped
        self._stop_plen: Dict[int, int] = {}        # prompt lengths of requests with stop sequences (for their usage record)

    def _params(self, p: GenParams) -> "core.SamplingParams":
        sp = core.SamplingParams()
        sp.temperature, sp.top_p, sp.top_k = p.temperature, p.top_p, p.top_k
        sp.max_new_tokens, sp.seed = p.max_new_tokens, p.seed
        if self.stop_at_eos and self.tok.eos_id is not None:
            sp.eos_id = self.tok.eos_id
            sp.stop_ids = [int(i) for i in self.tok.stop_ids]
        else:
            sp.ignore_eos = True
        return sp

    def _wrap(self, prompt: str) -> str:
        return prompt if self.chat_format == "none" else apply_chat_format(self.chat_format, prompt)

    def submit_generate(self, prompt: str, params: GenParams, on_text=None) -> Handle:
        h = Handle()
        try:
            ids = self.tok.encode(self._wrap(prompt))
            dec = [self.tok.first_token_id(t) for t in params.decision_targets] if params.decision_targets else None
            rid = self.engine.submit(ids, core.RequestKind.Generate, max(1, params.n), self._params(params), dec)
            self._pending[rid] = ("gen", h)
            h.backend = self
            if params.stop:
                self._stops[rid] = [t for t in params.stop if t]
                self._stop_plen[rid] = len(ids)
            if on_text is not None:
                self._streams[rid] = [on_text, [""] * max(1, params.n)]   # callback, text already sent per chain
        except Exception as e:  # a bad request fails that member only
            h.set_error(e)
        return h

    def token_ids_of(self, text: str):
        return list(self.tok.encode(text, add_bos=False))

    def submit_score(self, prompt: str, targets: Sequence[str]) -> Handle:
        h = Handle()
        try:
            if not targets:
                raise ValueError("no targets to score")
            ids = self.tok.encode(self._wrap(prompt))
            tids = [self.tok.first_token_id(t) for t in targets]
            rid = self.engine.submit(ids, core.RequestKind.Score, score_ids=tids)
            self._pending[rid] = ("score", h)
            h.backend = self
        except Exception as e:
            h.set_error(e)
        return h

    def submit_embed(self, text: str) -> Handle:
        h = Handle()
        try:
            rid = self.engine.submit(self.tok.encode(text), core.RequestKind.Embed)
            self._pending[rid] = ("embed", h)
            h.backend = self
        except Exception as e:
            h.set_error(e)
        return h

    incremental = True
    RESTART_MARKER = "\n[stream restarted]\n"

    def _group_failure(self, e: BaseException) -> None:
        """A failed step of a tensor / expert parallel member cannot be absorbed locally: the
        other ranks of the group are inside (or about to enter) a collective this rank will
        never join. The only state all ranks can agree on again is "the group is down", so
        the error leaves the process (the launcher ends the other ranks)."""
        m = self.engine.model
        if m.tp_size > 1 or m.ep_size > 1:
            raise RuntimeError(f"{self.name}: step failed on rank {m.comm.rank} of a {m.comm.world}-rank "
                               f"model group; the group cannot continue") from e

    def run(self) -> None:
        try:
            self.engine.run_until_done()
        except Exception as e:  # noqa: BLE001  a failed step (HIP / RCCL error, bad shapes) fails its requests only
            self._fail_all(e)
            self._group_failure(e)
            return
        self._harvest(final=True)

    def poll(self) -> bool:
        """One engine step, then the handles of whatever finished in it."""
        try:
            if self.engine.has_work():
                self.engine.step()
        except Exception as e:  # noqa: BLE001
            self._fail_all(e)
            self._group_failure(e)
            return False
        self._harvest(final=not self.engine.has_work())
        return bool(self._pending or self._discard)

    def harvest(self) -> bool:
        """Completes the handles of requests the engine has finished (for drivers that step
        the engine themselves). Returns True while requests remain."""
        self._harvest(final=not self.engine.has_work())
        return bool(self._pending or self._discard)

    def fail_pending(self, err: BaseException) -> None:
        self._fail_all(err)

    def has_leftovers(self) -> bool:
        return bool(self._discard or self.engine._cancel)

    def cancel(self, handle: Handle) -> bool:
        """The engine drops the request at its next step boundary (waiting, prefilling or
        decoding), its KV blocks go back to the pool, and the handle completes with an error.
        Members of a tensor / expert parallel group cannot cancel on one rank alone."""
        m = self.engine.model
        if m.tp_size > 1 or m.ep_size > 1:
            return False
        for rid, (_, h) in self._pending.items():
            if h is handle:
                self._streams.pop(rid, None)
                self._stops.pop(rid, None)
                return self.engine.cancel(rid)
        return False

    def _fail_all(self, e: BaseException) -> None:
        """An engine step raised: every request in flight gets the error (the protocol maps it
        to its failure policy, main.rs has `.expect()` here and dies), the engine drops them
        and is ready for new work. The process keeps serving."""
        log.error("%s: engine step failed, %d request(s) dropped: %s: %s", self.name, len(self._pending),
                  type(e).__name__, e)
        log.debug("%s: traceback of the failed step", self.name, exc_info=e)
        for rid, (_, h) in self._pending.items():
            if not h.done:
                h.set_error(BackendError(f"engine failure: {type(e).__name__}: {e}"))
        self._pending.clear()
        self._streams.clear()
        self._stops.clear()
        self._discard.clear()
        try:
            self.engine.abort_all()
        except Exception as e2:  # noqa: BLE001
            log.error("%s: abort after failure failed too: %s", self.name, e2)

    def _emit(self, rid: int, chains, final: bool) -> None:
        """Streams what is new in ``chains`` (token lists) since the last call. Text is
        decoded from the whole chain every time and only the stable part goes out: a trailing
        replacement character is an unfinished multi-byte sequence, held back until it
        completes (or the chain ends)."""
        cb, sent = self._streams[rid]
        eos = self.tok.eos_id
        for i, toks in enumerate(chains):
            if i >= len(sent):
                break
            if eos is not None and toks and toks[-1] == eos:
                toks = toks[:-1]
            text = self.tok.decode(toks)
            if not final and text.endswith("\ufffd"):
                text = text[:-1]            # an unfinished sequence decodes to ONE replacement character
            stops = self._stops.get(rid)
            if stops:
                # nothing of a stop sequence, or after it, may reach the client: cut at the first one,
                # and while the chain runs hold back a tail that could still grow into one
                cut = self._first_stop(text, stops)
                if cut >= 0:
                    text = text[:cut]
                elif not final:
                    text = text[:max(0, len(text) - (max(len(t) for t in stops) - 1))]
            if len(text) > len(sent[i]) and text.startswith(sent[i]):
                cb(i, text[len(sent[i]):])
                sent[i] = text
            elif sent[i] and not sent[i].startswith(text):
                # The chain no longer extends what the client has seen: the request was preempted
                # and its recomputation took another numerical path (other batch, other kernel
                # split) and parted ways on a near-tie. Say so instead of going silent: the
                # client gets a restart marker followed by the text as it now stands, so what it
                # has shown and the final result agree again.
                cb(i, self.RESTART_MARKER + text)
                sent[i] = text

    @staticmethod
    def _first_stop(text: str, stops: Sequence[str]) -> int:
        at = [i for i in (text.find(t) for t in stops) if i >= 0]
        return min(at) if at else -1

    def _texts(self, token_lists) -> List[str]:
        eos = self.tok.eos_id
        out = []
        for toks in token_lists:
            if eos is not None and toks and toks[-1] == eos:
                toks = toks[:-1]
            out.append(self.tok.decode(toks))
        return out

    def _stop_early(self) -> None:
        """Requests with text-level stop sequences: once EVERY chain of a request has produced one, the
        request is finished here with the truncated texts and the engine is told to drop it (its
        blocks return at the next step; its Aborted result is discarded when it comes)."""
        for rid, stops in list(self._stops.items()):
            ent = self._pending.get(rid)
            if ent is None:
                self._stops.pop(rid, None)
                continue
            chains = self.engine.sched.tokens_so_far(rid)
            if not chains or not all(chains):
                continue
            texts = self._texts(chains)
            # A trailing replacement character may be the first bytes of a multi-byte character that
            # the next tokens complete: it is not text yet and must not match a stop sequence (the
            # finished chain is searched whole in _harvest).
            cuts = [self._first_stop(t[:-1] if t.endswith("\ufffd") else t, stops) for t in texts]
            if any(c < 0 for c in cuts):
                continue
            if rid in self._streams:
                self._emit(rid, chains, final=True)
                del self._streams[rid]
            kind, h = ent
            plen = self._stop_plen.pop(rid, 0)
            h.usage = (plen, sum(len(t) for t in chains), 0)
            h.set_result(GenResult(texts=[t[:c] for t, c in zip(texts, cuts)], token_ids=[list(t) for t in chains],
                                   cum_logprobs=[0.0] * len(chains), prompt_tokens=plen,
                                   finish_reasons=["stop"] * len(chains)))
            del self._pending[rid]
            del self._stops[rid]
            if self.engine.cancel(rid):
                self._discard.add(rid)

    def _harvest(self, final: bool) -> None:
        if self._stops:
            self._stop_early()
        for rid in list(self._discard):
            if self.engine.pop_result(rid) is not None or rid not in self.engine._reqs:
                self._discard.discard(rid)
        for rid in list(self._streams):
            if rid in self._pending:
                chains = self.engine.sched.tokens_so_far(rid)
                if chains:
                    self._emit(rid, chains, final=False)
        for rid, (kind, h) in list(self._pending.items()):
            out = self.engine.pop_result(rid)
            if out is None:
                if final:
                    h.set_error(BackendError("request did not finish"))
                    del self._pending[rid]
                continue
            del self._pending[rid]
            h.usage = (out.prompt_len, sum(len(t) for t in out.token_ids), out.cached_tokens)
            if out.finish_reasons and all(r == "Aborted" for r in out.finish_reasons):
                self._streams.pop(rid, None)
                h.set_error(BackendError("cancelled"))
                continue
            if rid in self._streams:
                self._emit(rid, out.token_ids, final=True)
                del self._streams[rid]
            if kind == "gen":
                texts = self._texts(out.token_ids)
                reasons = ["length" if r == "Length" else "stop" for r in out.finish_reasons]
                stops = self._stops.pop(rid, None)
                self._stop_plen.pop(rid, None)
                if stops:       # a chain that ran on (the others had not stopped yet) still ends at its stop sequence
                    for i, t in enumerate(texts):
                        cut = self._first_stop(t, stops)
                        if cut >= 0:
                            texts[i], reasons[i] = t[:cut], "stop"
                h.set_result(GenResult(texts=texts, token_ids=out.token_ids, cum_logprobs=out.cum_logprobs,
                                       decision_logprobs=out.decision_logprobs, prompt_tokens=out.prompt_len,
                                       cached_tokens=out.cached_tokens, finish_reasons=reasons), at=out.finished_at)
            elif kind == "score":
                h.set_result(out.scores, at=out.finished_at)
            else:
                h.set_result(out.embedding, at=out.finished_at)

    def engines(self) -> list:
        return [self.engine] if (self._pending or self.has_leftovers()) else []

    def warm_up(self, **kw) -> Dict[str, float]:
        """See Engine.warm_up (first-use costs paid before the first request)."""
        return self.engine.warm_up(**kw)

    def stats(self) -> Dict[str, float]:
        return self.engine.stats()
