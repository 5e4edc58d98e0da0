"""Ensemble members: a persona bound to a backend.

Counterpart of the reference's ``LlmActor`` (/root/reference/src/main.rs:71-185):
``submit_propose`` / ``submit_evaluate`` / ``submit_refine`` build the prompt for the
role (main.rs:95, :111-136, :166-175) and queue it on the member's backend. Unlike
the reference, members may sit on *different* backends (different models on
different GPUs), several members may share one backend (personas on one model, the
reference's topology), and an evaluation can be a constrained two-way log-prob
decision instead of free text (``decision_mode="logprob"``).
"""
from __future__ import annotations

from dataclasses import dataclass
import numpy as np
from typing import List, Optional, Tuple

from .._native import core
from ..utils.log import get_logger
from .backend import Backend, BackendError, GenParams, Handle

log = get_logger("consensus.member")

VERDICT_TARGETS = ["Good", "NeedsRefinement"]


@dataclass
class Evaluation:
    verdict: "core.Verdict"
    reasoning: str
    well_formed: bool = True
    logprobs: Optional[Tuple[float, float]] = None  # (Good, NeedsRefinement) when decided on device


class Member:
    def __init__(
        self,
        persona: "core.Persona",
        backend: Backend,
        templates: Optional["core.PromptTemplates"] = None,
        params: Optional[GenParams] = None,
        parse_mode: "core.ParseMode" = core.ParseMode.Strict,
        decision_mode: str = "text",
        role: str = "member",
        weight: float = 1.0,
        dissent_rate: float = 0.0,
        dissent_seed: int = 0,
    ) -> None:
        self.persona = persona
        self.backend = backend
        self.templates = templates or core.PromptTemplates()
        self.params = params or GenParams()
        self.parse_mode = parse_mode
        self.decision_mode = decision_mode
        self.role = role
        self.weight = weight
        # Fault injection for tests and synthetic benchmarks: a Good verdict is turned into
        # NeedsRefinement with this probability (seeded, per evaluation), so the refinement
        # loop runs even when the model behind the member never objects -- random-init weights
        # give every persona the same two-way decision.
        self.dissent_rate = dissent_rate
        self._dissent_rng = np.random.default_rng(dissent_seed) if dissent_rate > 0 else None

    @property
    def name(self) -> str:
        return self.persona.name

    # ---- propose (main.rs:88-104) ----
    def submit_propose(self, question: str, params: Optional[GenParams] = None) -> Handle:
        log.debug("LLM member %s received question: %s", self.name, question)
        return self.backend.submit_generate(core.render_propose(self.templates, question), params or self.params)

    def submit_prompt(self, prompt: str, params: Optional[GenParams] = None) -> Handle:
        """Raw prompt (used by the vote strategies with their own templates)."""
        return self.backend.submit_generate(prompt, params or self.params)

    # ---- evaluate (main.rs:106-159) ----
    def submit_evaluate(self, question: str, answer: str) -> Handle:
        prompt = core.render_evaluate(self.templates, question, answer, self.persona)
        if self.decision_mode == "logprob":
            # One prefill, no decode: compare log P("Good") with log P("NeedsRefinement")
            # at the first reply position. The verdict line is all the protocol consumes;
            # the reference only logs the reasoning at debug level (main.rs:263).
            h = self.backend.submit_score(prompt + "\n", VERDICT_TARGETS)
            h.tag = "logprob"
            return h
        h = self.backend.submit_generate(prompt, self.params)
        h.tag = "text"
        return h

    def finish_evaluate(self, h: Handle) -> Evaluation:
        """Turns a completed evaluate handle into a verdict. Raises BackendError if the
        request failed; the caller maps that to abstain / NeedsRefinement."""
        ev = self._finish_evaluate(h)
        if self._dissent_rng is not None and ev.verdict == core.Verdict.Good and \
                self._dissent_rng.random() < self.dissent_rate:
            return Evaluation(core.Verdict.NeedsRefinement, "injected dissent", ev.well_formed, ev.logprobs)
        return ev

    def _finish_evaluate(self, h: Handle) -> Evaluation:
        res = h.result()
        if h.tag == "logprob":
            good, needs = float(res[0]), float(res[1])
            v = core.Verdict.Good if good >= needs else core.Verdict.NeedsRefinement
            return Evaluation(v, f"logprob decision: Good={good:.4f} NeedsRefinement={needs:.4f}", True, (good, needs))
        text = res.texts[0]
        parsed = core.parse_verdict(text, self.parse_mode)
        if not parsed.well_formed:
            log.error("Unexpected response from EvaluateAnswer: %s", text)  # main.rs:150
        return Evaluation(parsed.verdict, parsed.reasoning, parsed.well_formed)

    # ---- refine (main.rs:161-185) ----
    def submit_refine(self, question: str, answer: str, reasoning: str = "") -> Handle:
        prompt = core.render_refine(self.templates, question, answer, self.persona, reasoning)
        return self.backend.submit_generate(prompt, self.params)

    @staticmethod
    def finish_text(h: Handle) -> str:
        return h.result().texts[0]


def distinct_backends(members: List[Member]) -> List[Backend]:
    seen, out = set(), []
    for m in members:
        if id(m.backend) not in seen:
            seen.add(id(m.backend))
            out.append(m.backend)
    return out


__all__ = ["Member", "Evaluation", "distinct_backends", "BackendError", "VERDICT_TARGETS"]
