"""Backend seam: what a member calls to get text, scores or embeddings.

The reference calls a free function ``call_gemini(prompt)`` straight from its actor
handlers (/root/reference/src/main.rs:82-86, :97, :138, :178): no interface, so no way
to substitute a fake, and every failure is a panic. ``Backend`` is that missing seam.
Requests are *submitted* (cheap, returns a handle) and completed by ``run()``, so a
driver can queue work from many sessions and members before the engine batches it.

Implementations: ``EngineBackend`` (llm_consensus_amd/engine/backend.py, the in-process
MI355X engine), ``ScriptedBackend`` (canned replies + fault injection for protocol
tests) and ``CallableBackend`` (wraps any ``str -> str`` function).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Sequence


class BackendError(RuntimeError):
    """A request failed (timeout, engine fault, injected failure)."""


@dataclass
class GenParams:
    temperature: float = 0.7
    top_p: float = 0.95
    top_k: int = 0
    max_new_tokens: int = 256
    seed: int = 0
    n: int = 1                                  # chains forked from the prompt
    decision_targets: Optional[List[str]] = None  # if set, also return log-probs of these continuations
    #                                               at the position after the generated text
    stop: Optional[List[str]] = None            # text-level stop sequences: a chain ends before the first of them


@dataclass
class GenResult:
    texts: List[str]                            # one per chain
    token_ids: List[List[int]] = field(default_factory=list)
    cum_logprobs: List[float] = field(default_factory=list)
    decision_logprobs: Optional[List[List[float]]] = None  # [chain][target]
    prompt_tokens: int = 0
    cached_tokens: int = 0
    finish_reasons: List[str] = field(default_factory=list)   # per chain: "stop" (stop token / sequence) or "length"


class Handle:
    """Result slot for one submitted request."""

    __slots__ = ("_value", "_error", "_done", "submitted_at", "finished_at", "tag", "usage", "backend")

    def __init__(self, tag: Any = None) -> None:
        self.backend: Optional["Backend"] = None     # set by backends that can cancel (see cancel())
        # (prompt tokens, completion tokens, prompt tokens served from the prefix cache) of the
        # call, when the backend counts tokens; summed into ConsensusResult.usage
        self.usage: Optional[tuple] = None
        self._value = None
        self._error: Optional[BaseException] = None
        self._done = False
        self.submitted_at = time.perf_counter()
        self.finished_at: Optional[float] = None
        self.tag = tag

    def set_result(self, value: Any, at: Optional[float] = None) -> None:
        """``at``: when the result became available (perf_counter), if the backend knows better
        than "now" (an engine finishes requests long before a batch driver collects them)."""
        self._value, self._done, self.finished_at = value, True, (at if at is not None else time.perf_counter())

    def set_error(self, err: BaseException) -> None:
        self._error, self._done, self.finished_at = err, True, time.perf_counter()

    @property
    def done(self) -> bool:
        return self._done

    def cancel(self) -> bool:
        """Asks the backend to stop working on this request (its caller has gone away). Returns
        whether the backend took the request; the handle then completes with an error at the
        backend's next step. A finished handle, or one of a backend that cannot cancel, is left
        alone. Call from the thread that drives the backend."""
        if self._done or self.backend is None:
            return False
        return self.backend.cancel(self)

    @property
    def failed(self) -> bool:
        return self._done and self._error is not None

    def result(self) -> Any:
        if not self._done:
            raise BackendError("result() called before the backend ran this request")
        if self._error is not None:
            raise BackendError(str(self._error)) from self._error
        return self._value


class Backend:
    """Text-level LLM service. All submit_* are non-blocking; run() completes them."""

    name = "backend"

    def submit_generate(self, prompt: str, params: GenParams, on_text: Optional[Callable[[int, str], None]] = None) -> Handle:
        """``on_text(chain, new_text)``, where the backend supports it, is called from the
        thread that drives the backend with every piece of text as it is generated (token
        streaming); the handle still completes with the whole result."""
        raise NotImplementedError

    def submit_score(self, prompt: str, targets: Sequence[str]) -> Handle:
        """Log-probability of each target continuation (its first token) after ``prompt``.
        Result: ``List[float]`` aligned with ``targets``."""
        raise NotImplementedError

    def submit_embed(self, text: str) -> Handle:
        """Pooled, L2-normalised embedding of ``text``. Result: ``List[float]`` / ndarray."""
        raise NotImplementedError

    def run(self) -> None:
        """Drive the backend until every submitted handle is done."""
        raise NotImplementedError

    # Backends that can make progress in small increments set this; a driver may then keep
    # admitting new work between increments instead of running whole batches to completion.
    incremental = False

    def poll(self) -> bool:
        """Makes some progress on the submitted requests and completes the handles of those
        that finished. Returns True while requests remain. The default runs everything."""
        self.run()
        return False

    def fail_pending(self, err: BaseException) -> None:
        """Completes every outstanding handle with ``err`` (used by drivers that step this
        backend's engines themselves and caught a failure)."""

    def has_leftovers(self) -> bool:
        """True while the backend still has to let go of work nobody waits for any more (cancelled
        requests that leave at the engine's next step): a driver that is otherwise idle keeps
        stepping until this is False."""
        return False

    def cancel(self, handle: Handle) -> bool:
        """Stops work on one submitted request, where a backend can (the reference cannot: a call
        to its remote model runs to the end whoever still waits for it,
        /root/reference/src/main.rs:82-86). The default cannot."""
        return False

    def run_local(self) -> None:
        """Compute phase of run(): everything that needs no other backend's ranks."""
        self.run()

    def sync(self) -> None:
        """Exchange phase of run(): publish results to ranks that do not host this backend
        (a no-op for backends that live entirely in this process)."""

    def engines(self) -> list:
        """In-process engines this backend would drive in run_local() (empty for remote or
        scripted backends); lets a driver overlap the steps of several engines."""
        return []

    def warm_up(self, **kw) -> Dict[str, float]:
        """Pays first-use costs (kernel loading, graph recording) before the first request, where
        a backend has any; returns what it did. The default has none."""
        return {}

    def token_ids_of(self, text: str) -> Optional[List[int]]:
        """Token ids of ``text`` in this backend's vocabulary (no BOS), or None when the backend
        has no tokenizer of its own (remote, scripted). Strategies that classify answers at the
        token level (``decision_mode="tokens"``) fall back to the text parser on None."""
        return None

    def stats(self) -> Dict[str, float]:
        return {}


def add_usage(total: Dict[str, int], handles) -> Dict[str, int]:
    """Adds the token counts of finished member calls to ``total`` (prompt_tokens,
    completion_tokens, cached_tokens, total_tokens)."""
    for h in handles:
        u = getattr(h, "usage", None)
        if u:
            total["prompt_tokens"] = total.get("prompt_tokens", 0) + int(u[0])
            total["completion_tokens"] = total.get("completion_tokens", 0) + int(u[1])
            total["cached_tokens"] = total.get("cached_tokens", 0) + int(u[2])
    total["total_tokens"] = total.get("prompt_tokens", 0) + total.get("completion_tokens", 0)
    return total


def run_backends(backends: Sequence["Backend"]) -> None:
    """Completes the submitted requests of several backends. Engines that live in this
    process are stepped together with their GPU work overlapped on separate streams; all
    compute phases are started before any exchange phase, so backends hosted on different
    GPUs work concurrently and a rank that hosts none of them waits only once."""
    engines = [e for b in backends for e in b.engines()]
    if len(engines) > 1:
        from ..engine.engine import run_engines

        try:
            run_engines(engines)
        except Exception as e:  # noqa: BLE001
            # A step of one engine failed while several were in flight: none of the interleaved
            # steps can be trusted to have completed, so every engine-backed request of this
            # batch fails (each backend drops its requests and frees their KV).
            for b in backends:
                if b.engines():
                    b.fail_pending(e)
    for b in backends:
        b.run_local()
    for b in backends:
        b.sync()


class BackendStepper:
    """Advances a set of backends a little at a time, for drivers that admit new work between
    increments (SessionPool, VotePool, the HTTP server). Engines that live in this process are
    stepped through one EngineDriver, so their steps overlap on their own streams and prefills
    take turns exactly as in run_backends; other backends get ``poll()``."""

    def __init__(self, backends: Sequence["Backend"]) -> None:
        self.backends = list(backends)
        self._driver = None

    def step(self) -> bool:
        """One increment. Returns False when nothing moved (work in flight is still running)."""
        inc = [b for b in self.backends if b.incremental]
        pairs = [(b, e) for b in inc for e in b.engines()]
        moved = False
        if pairs:
            if self._driver is None:
                from ..engine.engine import EngineDriver

                self._driver = EngineDriver()
            owners = list({id(b): b for b, _ in pairs}.values())
            try:
                moved = self._driver.advance([e for _, e in pairs])
            except Exception as err:  # noqa: BLE001  interleaved steps: none can be trusted after a failure
                self._driver.reset()
                for b in owners:
                    b.fail_pending(err)
                moved = True
            for b in owners:
                b.harvest()
        for b in self.backends:
            if not b.incremental:
                b.poll()
                moved = True
        return moved


class CallableBackend(Backend):
    """Wraps ``fn(prompt) -> str``; generation only."""

    def __init__(self, fn: Callable[[str], str], name: str = "callable") -> None:
        self.fn = fn
        self.name = name
        self._queue: List[tuple] = []

    def submit_generate(self, prompt: str, params: GenParams, on_text=None) -> Handle:
        h = Handle()
        self._queue.append((h, prompt, params, on_text))
        return h

    def run(self) -> None:
        q, self._queue = self._queue, []
        for h, prompt, params, on_text in q:
            try:
                texts = [self.fn(prompt) for _ in range(max(1, params.n))]
                if on_text is not None:
                    for i, t in enumerate(texts):
                        on_text(i, t)
                h.usage = (len(prompt.split()), sum(len(t.split()) for t in texts), 0)
                h.set_result(GenResult(texts=texts))
            except Exception as e:  # surfaced as a member failure, never a crash
                h.set_error(e)


class ScriptedBackend(Backend):
    """Fake backend for protocol tests.

    ``script`` maps a role to a list of replies consumed in order, or to a callable
    ``(prompt) -> reply``. Roles are detected from the rendered prompt: the evaluate
    template contains "Good or NeedsRefinement" style instructions, the refine template
    mentions refinement, everything else is a proposal. Fault injection:

    * ``fail_every``   -- every n-th request raises (a dropped reply);
    * ``garbage_every``-- every n-th evaluate reply is replaced by unparseable text;
    * ``delay_s``      -- sleep per request (a slow member).
    """

    def __init__(
        self,
        script: Optional[Dict[str, Any]] = None,
        default_verdict: str = "Good",
        fail_every: int = 0,
        garbage_every: int = 0,
        delay_s: float = 0.0,
        scores: Optional[Callable[[str, Sequence[str]], List[float]]] = None,
        embed: Optional[Callable[[str], List[float]]] = None,
        name: str = "scripted",
    ) -> None:
        self.script = {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in (script or {}).items()}
        self.default_verdict = default_verdict
        self.fail_every = fail_every
        self.garbage_every = garbage_every
        self.delay_s = delay_s
        self.scores = scores
        self.embed = embed
        self.name = name
        self.calls: List[tuple] = []      # (role, prompt) log for assertions
        self._queue: List[tuple] = []
        self._count = 0
        self._eval_count = 0

    @staticmethod
    def role_of(prompt: str) -> str:
        if "Your Instructions:" in prompt:
            head = prompt.split("Your Instructions:", 1)[1]
            return "refine" if "Examples:" not in head else "evaluate"
        return "propose"

    def _next(self, role: str, prompt: str) -> str:
        src = self.script.get(role)
        if callable(src):
            return src(prompt)
        if src:
            return src.pop(0) if len(src) > 1 else src[0]
        if role == "evaluate":
            return f"{self.default_verdict}\nscripted reasoning"
        if role == "refine":
            return "refined answer"
        return "proposed answer"

    def submit_generate(self, prompt: str, params: GenParams, on_text=None) -> Handle:
        h = Handle()
        self._queue.append(("gen", h, prompt, params))
        return h

    def submit_score(self, prompt: str, targets: Sequence[str]) -> Handle:
        h = Handle()
        self._queue.append(("score", h, prompt, list(targets)))
        return h

    def submit_embed(self, text: str) -> Handle:
        h = Handle()
        self._queue.append(("embed", h, text, None))
        return h

    def run(self) -> None:
        q, self._queue = self._queue, []
        for kind, h, prompt, extra in q:
            self._count += 1
            if self.delay_s:
                time.sleep(self.delay_s)
            if self.fail_every and self._count % self.fail_every == 0:
                h.set_error(BackendError("injected failure"))
                continue
            try:
                if kind == "gen":
                    role = self.role_of(prompt)
                    self.calls.append((role, prompt))
                    n = max(1, extra.n)
                    texts = []
                    for _ in range(n):
                        reply = self._next(role, prompt)
                        if role == "evaluate":
                            self._eval_count += 1
                            if self.garbage_every and self._eval_count % self.garbage_every == 0:
                                reply = "I would rather not say."
                        texts.append(reply)
                    dec = None
                    if extra.decision_targets:
                        fn = self.scores or (lambda p, t: [0.0] + [-1.0] * (len(t) - 1))
                        dec = [list(fn(prompt + x, extra.decision_targets)) for x in texts]
                    h.usage = (len(prompt.split()), sum(len(t.split()) for t in texts), 0)   # words stand in for tokens
                    h.set_result(GenResult(texts=texts, decision_logprobs=dec))
                elif kind == "score":
                    self.calls.append(("score", prompt))
                    fn = self.scores or (lambda p, t: [0.0] + [-1.0] * (len(t) - 1))
                    h.usage = (len(prompt.split()), 0, 0)
                    h.set_result(list(fn(prompt, extra)))
                else:
                    self.calls.append(("embed", prompt))
                    fn = self.embed or (lambda t: [float(len(t) % 7), 1.0, float(t.count("a"))])
                    h.usage = (len(prompt.split()), 0, 0)
                    h.set_result(list(fn(prompt)))
            except Exception as e:
                h.set_error(e)
