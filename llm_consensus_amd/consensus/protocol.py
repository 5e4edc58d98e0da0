"""Reference-protocol driver: propose -> everyone evaluates -> refine until unanimous.

``Coordinator`` plays the role of the reference's Coordinator actor plus its REPL
glue (/root/reference/src/main.rs:187-348, :441-470) for any number of questions at
once. The per-question logic lives in the native ``ConsensusSession`` state machine;
this file only moves text between sessions and members:

    wave := { submit every pending member call of every live session }
            -> run each distinct backend once (the engine batches across sessions,
               members and roles)
            -> feed replies back into the sessions.

Completion is event-driven (an ``Action.Done``), not the reference's 500 ms readiness
poll (main.rs:448-459). Member failures become abstentions (or NeedsRefinement votes)
instead of the reference's task panics that leave the REPL spinning forever.
"""
from __future__ import annotations

import json
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

from .._native import core
from ..utils.log import get_logger
from ..utils.trace import instant, span
from .backend import BackendError, BackendStepper, Handle, add_usage, run_backends
from .members import Member, distinct_backends

log = get_logger("consensus.coordinator")
K = core.Action.Kind


@dataclass
class ConsensusResult:
    question: str
    answer: str
    strategy: str = "reference_protocol"
    rounds: int = 0                 # evaluation rounds used
    member_calls: int = 0
    unanimous: bool = False
    forced: bool = False            # round/deadline budget broke the loop
    proposer: Optional[str] = None
    latency_s: float = 0.0
    agreement: float = 0.0
    votes: List[Dict] = field(default_factory=list)  # last round: [{member, verdict, reasoning}]
    transcript: List[Dict] = field(default_factory=list)
    extra: Dict = field(default_factory=dict)
    usage: Dict[str, int] = field(default_factory=dict)   # tokens over all member calls of the question

    def to_json(self) -> str:
        return json.dumps(self.__dict__, default=str)


class _Live:
    __slots__ = ("session", "action", "t0", "handles", "index", "failed", "usage")

    def __init__(self, session, action, index):
        self.session, self.action, self.index = session, action, index
        self.usage: Dict[str, int] = {}
        self.t0 = time.perf_counter()
        self.handles: List[Handle] = []
        self.failed: Optional[str] = None


class Coordinator:
    """Owns the member registry and runs consensus sessions (main.rs:187-208)."""

    def __init__(self, members: Optional[Sequence[Member]] = None,
                 session_config: Optional["core.SessionConfig"] = None,
                 deadline_s: Optional[float] = None,
                 transcript_path: Optional[str] = None) -> None:
        self.members: List[Member] = []
        self.session_config = session_config or core.SessionConfig()
        self.deadline_s = deadline_s
        self.transcript_path = transcript_path
        self._asked = 0
        for m in members or ():
            self.register(m)

    # main.rs:210-218
    def register(self, member: Member) -> bool:
        if any(m.name == member.name for m in self.members):
            # HashMap::insert semantics: a re-registration replaces the entry
            self.members = [member if m.name == member.name else m for m in self.members]
        else:
            self.members.append(member)
        log.debug("%s registered with Coordinator.", member.name)
        return True

    # ------------------------------------------------------------------
    def pool(self) -> "SessionPool":
        """The driver without batch boundaries (same interface as strategies.VotePool)."""
        return SessionPool(self)

    def ask(self, question: str) -> ConsensusResult:
        return self.ask_many([question])[0]

    def ask_many(self, questions: Sequence[str], continuous: bool = False) -> List[ConsensusResult]:
        """Answers all questions. By default the sessions advance in waves (every session's
        pending calls, then all backends to completion); ``continuous`` lets each session move
        on as soon as its own calls are done (see SessionPool)."""
        if not self.members:
            raise RuntimeError("no members registered")  # the reference panics in choose() (main.rs:229)
        if continuous:
            pool = SessionPool(self)
            tickets = [pool.submit(q) for q in questions]
            done = dict(pool.drain())
            return [done[t] for t in tickets]
        names = [m.name for m in self.members]
        live: List[_Live] = []
        for q in questions:
            s = core.ConsensusSession(names, self.session_config, self._asked)
            self._asked += 1
            log.debug("Received AskQuestion: %s", q)
            live.append(_Live(s, s.ask(q), len(live)))
        results: List[Optional[ConsensusResult]] = [None] * len(live)
        backends = distinct_backends(self.members)

        pending = list(live)
        wave = 0
        while pending:
            for L in pending:
                self._submit(L)
            wave += 1
            with span(f"wave{wave}", "protocol", sessions=len(pending),
                      calls=sum(len(L.handles) for L in pending)):
                run_backends(backends)
            nxt: List[_Live] = []
            for L in pending:
                res = self._collect(L)
                if res is not None:
                    results[L.index] = res
                else:
                    nxt.append(L)
            pending = nxt
        return results  # type: ignore[return-value]

    def _submit(self, L: _Live) -> None:
        """Issues the member calls of the session's pending action (non-blocking)."""
        s, a = L.session, L.action
        L.handles = []
        if a.kind == K.Propose:
            L.handles = [self.members[a.member].submit_propose(s.question)]
        elif a.kind == K.Evaluate:
            log.debug("Asking actors to evaluate answer.")
            L.handles = [m.submit_evaluate(s.question, s.answer()) for m in self.members]
        elif a.kind == K.Refine:
            log.debug("Asking %s to refine the answer.", self.members[a.member].name)
            L.handles = [self.members[a.member].submit_refine(s.question, s.answer(), s.refiner_context())]
        # per-session timeline: one event per protocol transition (the wave spans only exist in
        # the wave driver)
        instant(f"session.{a.kind.name.lower()}", "protocol", session=L.index, round=s.evaluation_count,
                calls=len(L.handles), member=self.members[a.member].name if a.kind != K.Evaluate else "*")

    def _collect(self, L: _Live) -> Optional[ConsensusResult]:
        """Feeds the completed member calls to the session. Returns the result when the
        session is finished, None when it has a next action to submit."""
        add_usage(L.usage, L.handles)
        try:
            L.action = self._advance(L)
        except BackendError as e:
            # A failed propose/refine leaves nothing to vote on: finish with what we have.
            log.error("member call failed: %s", e)
            L.failed = str(e)
            L.action = L.session.abort_with_current_answer()
        if L.action.kind != K.Done and self.deadline_s is not None and \
                time.perf_counter() - L.t0 > self.deadline_s and L.session.has_answer:
            log.debug("Deadline exceeded; returning the current answer.")
            L.action = L.session.abort_with_current_answer()
        return self._finish(L) if L.action.kind == K.Done else None

    # ------------------------------------------------------------------
    def _advance(self, L: _Live):
        s, a = L.session, L.action
        if a.kind == K.Propose:
            answer = Member.finish_text(L.handles[0])
            log.debug("Received answer to current question: %s", answer)
            return s.on_proposal(answer)
        if a.kind == K.Refine:
            answer = Member.finish_text(L.handles[0])
            log.debug("Received new answer to current question: %s", answer)
            act = s.on_refinement(answer)
            if act.kind == K.Done:
                log.debug("Evaluated the maximum number of times. Breaking the loop.")
            return act
        if a.kind == K.Evaluate:
            act = a
            for i, (m, h) in enumerate(zip(self.members, L.handles)):
                try:
                    ev = m.finish_evaluate(h)
                    log.debug("%s evaluated the answer as %s. %s", m.name, ev.verdict.name, ev.reasoning)
                    act = s.on_evaluation(i, ev.verdict, ev.reasoning)
                except BackendError as e:
                    log.error("%s failed to evaluate: %s", m.name, e)
                    act = s.on_member_failure(i)
            return act
        raise RuntimeError(f"unexpected action {a.kind}")

    def _finish(self, L: _Live) -> ConsensusResult:
        s = L.session
        names = s.member_names
        tr = [
            {
                "round": r.round,
                "answer": r.answer,
                "refiner": names[r.refiner] if r.refiner >= 0 else None,
                "votes": [
                    {"member": names[v.member], "verdict": v.verdict.name, "reasoning": v.reasoning} for v in r.votes
                ],
            }
            for r in s.transcript
        ]
        last_votes = tr[-1]["votes"] if tr else []
        good = sum(1 for v in last_votes if v["verdict"] != "NeedsRefinement")
        res = ConsensusResult(
            question=s.question,
            answer=s.answer(),
            rounds=s.evaluation_count,
            member_calls=s.member_calls,
            unanimous=s.ready() and not s.forced,
            forced=s.forced,
            proposer=names[s.proposer] if s.proposer >= 0 else None,
            latency_s=time.perf_counter() - L.t0,
            agreement=(good / len(last_votes)) if last_votes else 0.0,
            votes=last_votes,
            transcript=tr,
            usage=dict(L.usage),
        )
        if L.failed:
            res.extra["error"] = L.failed
        if self.transcript_path:
            with open(self.transcript_path, "a") as f:
                f.write(res.to_json() + "\n")
        return res


class SessionPool:
    """The protocol driver without waves: sessions are admitted at any time and each one moves
    to its next action as soon as ITS member calls have completed, while the backends are
    advanced one engine step at a time in between. A question that arrives while others are
    being answered joins the running batch at the next step instead of waiting for it to
    drain (the reference serves one question at a time and polls it every 500 ms,
    /root/reference/src/main.rs:428-471). Needs backends that live in this process; ranks of a
    distributed ensemble must stay in lock step and use the wave driver."""

    def __init__(self, coordinator: Coordinator) -> None:
        self.c = coordinator
        self.backends = distinct_backends(coordinator.members)
        self.stepper = BackendStepper(self.backends)
        self.live: Dict[int, _Live] = {}
        self._next = 0

    @property
    def active(self) -> int:
        return len(self.live)

    def submit(self, question: str) -> int:
        """Admits a question; returns its ticket."""
        c = self.c
        if not c.members:
            raise RuntimeError("no members registered")
        s = core.ConsensusSession([m.name for m in c.members], c.session_config, c._asked)
        c._asked += 1
        L = _Live(s, s.ask(question), self._next)
        self._next += 1
        c._submit(L)
        self.live[L.index] = L
        return L.index

    def pump(self) -> List[tuple]:
        """One round of progress. Returns [(ticket, ConsensusResult)] of the sessions that
        finished in it."""
        finished = []
        for t, L in list(self.live.items()):
            if not all(h.done for h in L.handles):
                continue
            res = self.c._collect(L)
            if res is not None:
                finished.append((t, res))
                del self.live[t]
            else:
                self.c._submit(L)
        if not self.stepper.step() and not finished:
            time.sleep(20e-6)
        return finished

    def cancel(self, ticket: int) -> bool:
        """Drops a session whose caller has gone away: the member calls it waits for are cancelled
        where the backends can, and it produces no result."""
        L = self.live.pop(ticket, None)
        if L is None:
            return False
        for h in L.handles:
            h.cancel()
        return True

    def drain(self) -> List[tuple]:
        out = []
        while self.live:
            out += self.pump()
        return out
