"""Span tracing: Chrome trace output, per-phase summaries, and that engine / strategy /
protocol code paths emit spans. The reference only has log lines
(/root/reference/src/main.rs:352); nothing here has a counterpart to compare with."""
import json

import torch

from llm_consensus_amd.app import build_ensemble
from llm_consensus_amd.config import ConsensusConfig, load_config
from llm_consensus_amd.consensus.backend import ScriptedBackend
from llm_consensus_amd.utils import trace
from pathlib import Path

CONFIGS = Path(__file__).resolve().parent.parent / "configs"


def test_disabled_tracer_records_nothing():
    t = trace.Tracer()
    with t.span("x"):
        pass
    assert t.events == []


def test_spans_nest_and_save(tmp_path):
    t = trace.Tracer()
    t.enable(str(tmp_path / "t.json"), roctx=False)
    with t.span("outer", n=1):
        with t.span("inner"):
            pass
    t.instant("mark", k=2)
    path = t.save()
    ev = json.load(open(path))["traceEvents"]
    names = [e["name"] for e in ev]
    assert names == ["inner", "outer", "mark"]
    inner, outer = ev[0], ev[1]
    assert outer["ts"] <= inner["ts"] and outer["ts"] + outer["dur"] >= inner["ts"] + inner["dur"]
    assert outer["args"] == {"n": 1}
    assert t.summary()["outer"]["count"] == 1


def test_engine_strategy_and_protocol_emit_spans():
    trace.TRACER.enable(None, roctx=False)
    trace.TRACER.clear()
    try:
        ens = build_ensemble(load_config(str(CONFIGS / "plumbing_tiny.toml")), torch.device("cpu"))
        ens.ask("Q? A) a B) b C) c D) d")
        names = set(trace.TRACER.summary())
        assert {"m0.prefill", "m0.decode", "generate:m0+m1+m2", "reduce"} <= names, names
        trace.TRACER.clear()
        be = ScriptedBackend({"propose": ["a"], "evaluate": ["Good\nok"] * 8})
        build_ensemble(ConsensusConfig(), torch.device("cpu"), backends={"default": be}).ask("why?")
        assert {"wave1", "wave2"} <= set(trace.TRACER.summary())
    finally:
        trace.TRACER.disable()
        trace.TRACER.clear()
