"""Verdict parser parity with /root/reference/src/main.rs:139-153 plus tolerant mode."""
import pytest

from llm_consensus_amd._native import core

V = core.Verdict
S, T = core.ParseMode.Strict, core.ParseMode.Tolerant


@pytest.mark.parametrize(
    "reply,verdict,well_formed,reasoning",
    [
        ("Good\nbecause it is", V.Good, True, "because it is"),
        ("NeedsRefinement\nr1\n\n\nr2", V.NeedsRefinement, True, "r1\n\nr2"),
        # spaces are stripped from the first line only (main.rs:142)
        ("Needs Refinement\nwhy", V.NeedsRefinement, True, "why"),
        ("  G o o d  \nx", V.Good, True, "x"),
        # leading blank lines are dropped before indexing (main.rs:139-141)
        ("\n\nGood\nx", V.Good, True, "x"),
        # imitating the few-shot examples is rejected in strict mode (main.rs:129 vs :145)
        ("Evaluation: Good\nReasoning: fine", V.NeedsRefinement, False, "Reasoning: fine"),
        ("good\nlowercase is not accepted", V.NeedsRefinement, False, "lowercase is not accepted"),
        ("Good.\ntrailing punctuation is not accepted", V.NeedsRefinement, False, "trailing punctuation is not accepted"),
        ("", V.NeedsRefinement, False, ""),           # the reference would panic here
        ("\n\n", V.NeedsRefinement, False, ""),
    ],
)
def test_strict(reply, verdict, well_formed, reasoning):
    p = core.parse_verdict(reply, S)
    assert (p.verdict, p.well_formed, p.reasoning) == (verdict, well_formed, reasoning)


@pytest.mark.parametrize(
    "reply,verdict",
    [
        ("Evaluation: Good\nReasoning: fine", V.Good),
        ("Evaluation: NeedsRefinement\nReasoning: thin", V.NeedsRefinement),
        ("**Needs Refinement.**\nmore", V.NeedsRefinement),
        ("good - it is fine", V.Good),
        ("Verdict: needs refinement because x", V.NeedsRefinement),
    ],
)
def test_tolerant_accepts(reply, verdict):
    p = core.parse_verdict(reply, T)
    assert p.well_formed and p.verdict == verdict


def test_tolerant_same_line_reasoning_is_kept():
    p = core.parse_verdict("Good - it is fine\nand more", T)
    assert p.reasoning.startswith("it is fine") and "and more" in p.reasoning


def test_tolerant_still_rejects_garbage():
    p = core.parse_verdict("I would rather not say.", T)
    assert not p.well_formed and p.verdict == V.NeedsRefinement


@pytest.mark.parametrize(
    "text,idx",
    [
        ("The answer is B.", 1),
        ("Reasoning... Answer: (C)", 2),
        ("answer: a\nthen later Answer: D", 3),
        ("I pick (B) because", 1),
        ("C", 2),
        ("B) since", 1),
        ("Because of reasons", -1),
        ("Answer: E", -1),
        ("", -1),
    ],
)
def test_extract_choice(text, idx):
    assert core.extract_choice(text, "ABCD") == idx


def test_extract_number_and_normalize():
    assert core.extract_number("so 3 + 4 = 7 apples") == 7.0
    assert core.extract_number("total is 1,234.5 dollars") == 1234.5
    assert core.extract_number("first 9 then #### 42") == 42.0
    assert core.extract_number("minus -3") == -3.0
    assert core.extract_number("none here") is None
    assert core.normalize_answer("  The  Answer!  ") == "the answer"
