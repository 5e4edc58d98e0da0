"""The native text functions (csrc/core/verdict.cpp, prompts.cpp, tokenizer) on arbitrary input.
A model's reply is untrusted text: whatever it contains, the parsers must return a value (the
reference panics on a reply without the expected JSON path and treats every unknown first line as
an error string, /root/reference/src/main.rs:86, 139-153) and strict parsing must stay a subset of
tolerant parsing."""
import numpy as np
from hypothesis import given, settings, strategies as st

from llm_consensus_amd._native import core

text = st.text(alphabet=st.characters(blacklist_categories=("Cs",)), max_size=300)
V = core.Verdict


@settings(max_examples=500, deadline=None, derandomize=True)
@given(text)
def test_verdict_parsers_total_and_consistent(reply):
    s = core.parse_verdict(reply, core.ParseMode.Strict)
    t = core.parse_verdict(reply, core.ParseMode.Tolerant)
    assert s.verdict in (V.Good, V.NeedsRefinement, V.Abstain) and t.verdict in (V.Good, V.NeedsRefinement, V.Abstain)
    if s.well_formed:                       # what the strict parser accepts, the tolerant one reads the same way
        assert t.well_formed and t.verdict == s.verdict
    first = reply.split("\n", 1)[0]
    if first == "Good":
        assert s.well_formed and s.verdict == V.Good
    if first == "NeedsRefinement":
        assert s.well_formed and s.verdict == V.NeedsRefinement


@settings(max_examples=300, deadline=None, derandomize=True)
@given(text, st.sampled_from(["Good", "NeedsRefinement"]))
def test_a_verdict_on_the_first_line_survives_any_reasoning(reasoning, word):
    p = core.parse_verdict(word + "\n" + reasoning, core.ParseMode.Strict)
    assert p.well_formed and p.verdict == (V.Good if word == "Good" else V.NeedsRefinement)


@settings(max_examples=500, deadline=None, derandomize=True)
@given(text)
def test_answer_extractors_total(reply):
    c = core.extract_choice(reply, "ABCD")
    assert -1 <= c < 4
    n = core.extract_number(reply)
    assert n is None or isinstance(n, float)
    assert isinstance(core.normalize_answer(reply), str)
    for i, letter in enumerate("ABCD"):      # an explicit answer line wins over whatever precedes it
        assert core.extract_choice(reply.replace("Answer", "answ") + f"\nAnswer: {letter}", "ABCD") == i


@settings(max_examples=300, deadline=None, derandomize=True)
@given(text, text, text)
def test_prompts_render_for_any_text(question, answer, reasoning):
    t = core.PromptTemplates()
    p = core.default_personas()[0]
    for layout in (t, core.PromptTemplates.persona_last(), core.PromptTemplates.persona_first()):
        for out in (core.render_propose(layout, question), core.render_evaluate(layout, question, answer, p),
                    core.render_refine(layout, question, answer, p, reasoning),
                    core.evaluate_shared_prefix(layout, question, answer)):
            assert isinstance(out, str)
        assert question in core.render_propose(layout, question)
        ev = core.render_evaluate(layout, question, answer, p)
        assert ev.startswith(core.evaluate_shared_prefix(layout, question, answer))


@settings(max_examples=500, deadline=None, derandomize=True)
@given(text)
def test_byte_tokenizer_round_trip(s):
    tok = core.ByteTokenizer()
    ids = tok.encode(s, True)
    assert len(ids) == len(s.encode("utf-8")) + 1
    raw = s.encode("utf-8")
    assert bytes(tok.decode(ids)) == raw                 # the native decoder returns the bytes; text is the caller's choice
    assert bytes(tok.decode(tok.encode(s, False))) == raw
