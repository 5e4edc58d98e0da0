"""Prompt rendering semantics (/root/reference/src/main.rs:95, :111-136, :166-175)."""
from llm_consensus_amd._native import core


def test_default_personas_match_reference_registry():
    ps = core.default_personas()
    assert [(p.name, p.domain) for p in ps] == [
        ("High Society", "Society and Culture"),
        ("The Technician", "Technical Detail"),
        ("Art Boy", "Art and Imagination"),
        ("Programming Nerd", "Computer Science"),
    ]
    for p in ps:
        bullets = [l for l in p.tuning.split("\n") if l.startswith("* ")]
        assert len(bullets) == 10 and p.tuning.startswith("\n")


def test_propose_keeps_quotes_and_question():
    t = core.PromptTemplates()
    s = core.render_propose(t, 'What is "x"?')
    assert s.endswith('\n\nWhat is "x"?') and "language model" in s


def test_evaluate_structure_and_quote_stripping():
    t = core.PromptTemplates()
    p = core.default_personas()[1]
    s = core.render_evaluate(t, 'Why "q"?', 'Because "a".', p)
    assert '"' not in s                                  # main.rs:136
    assert s.index("Question: Why q?") < s.index("Answer: Because a.") < s.index("Your Instructions:")
    assert p.domain in s and "* Cause and effect" in s
    assert "Good" in s and "NeedsRefinement" in s and "Examples:" in s
    assert s.count("Evaluation:") == 2                   # two few-shot examples


def test_evaluate_prefix_is_persona_independent():
    t = core.PromptTemplates()
    ps = core.default_personas()
    prefix = core.evaluate_shared_prefix(t, "Q?", "A.")
    rendered = [core.render_evaluate(t, "Q?", "A.", p) for p in ps]
    assert all(r.startswith(prefix) for r in rendered)
    assert len(prefix) > len("Question: Q?")
    # and personas do differ after it
    assert len({r[len(prefix):] for r in rendered}) == len(ps)


def test_persona_first_layout_puts_everything_question_independent_in_front():
    """The head of a persona's evaluate prompt (instructions, domain, tuning, rules, examples) is
    the same for every question, so a prefix cache serves it after the first one; only the
    question and the answer follow. Same information as the reference layout."""
    t = core.PromptTemplates.persona_first()
    p = core.default_personas()[2]
    a = core.render_evaluate(t, "Q one?", "A one.", p)
    b = core.render_evaluate(t, "A much longer second question?", "Another answer.", p)
    head = a[: a.index("\n---\nQuestion: Q one?")]
    assert b.startswith(head) and len(head) > 0.8 * len(a)
    assert p.domain in head and "* " in head and head.count("Evaluation:") == 2 and "NeedsRefinement" in head
    assert a.endswith("Question: Q one?\n---\nAnswer: A one.")
    other = core.render_evaluate(t, "Q one?", "A one.", core.default_personas()[0])
    assert other != a and other.endswith("Question: Q one?\n---\nAnswer: A one.")
    r = core.render_refine(t, "Q?", "A.", p, "- X: too short")
    assert r.index(p.domain) < r.index("Question: Q?") < r.index("Answer: A.") < r.index("too short")


def test_refine_omits_reasoning_by_default():
    t = core.PromptTemplates()
    p = core.default_personas()[0]
    s = core.render_refine(t, "Q?", "A.", p)
    assert "objections" not in s and p.domain in s and "needed refinement" in s
    s2 = core.render_refine(t, "Q?", "A.", p, "- X: too short")
    assert "too short" in s2


def test_placeholders_in_user_text_are_not_expanded():
    t = core.PromptTemplates()
    p = core.default_personas()[0]
    s = core.render_evaluate(t, "what is {answer}?", "{domain}", p)
    assert "what is {answer}?" in s and "Answer: {domain}" in s


def test_multiple_choice_prompt():
    s = core.render_multiple_choice("2+2?", ["3", "4"], "AB", False)
    assert "A. 3" in s and "B. 4" in s and s.endswith("Answer:")
