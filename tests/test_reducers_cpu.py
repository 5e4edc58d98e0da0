import pytest
import numpy as np
from hypothesis import given, settings, strategies as st

from llm_consensus_amd._native import core
from llm_consensus_amd.consensus import reducers


def test_majority_basic_and_ties():
    r = core.majority_vote(np.array([1, 1, 2, -1], dtype=np.int32), 4)
    assert (r.winner, r.count, r.valid, r.unanimous) == (1, 2, 3, False) and abs(r.agreement - 2 / 3) < 1e-6
    r = core.majority_vote(np.array([2, 1, 2, 1], dtype=np.int32), 4, core.TieBreak.LowestClass)
    assert r.winner == 1
    r = core.majority_vote(np.array([2, 1, 2, 1], dtype=np.int32), 4, core.TieBreak.FirstVoter)
    assert r.winner == 2
    r = core.majority_vote(np.array([-1, -1], dtype=np.int32), 4)
    assert r.winner == -1 and r.valid == 0
    r = core.majority_vote(np.array([0, 3, 3], dtype=np.int32), 4, core.TieBreak.LowestClass,
                           np.array([5.0, 1.0, 1.0], dtype=np.float32))
    assert r.winner == 0


@settings(max_examples=100, deadline=None, derandomize=True)
@given(st.lists(st.lists(st.integers(-1, 3), min_size=5, max_size=5), min_size=1, max_size=8))
def test_majority_batch_matches_numpy(rows):
    v = np.array(rows, dtype=np.int32)
    out = reducers.majority_vote_batch(v, 4)
    for r in range(v.shape[0]):
        cnt = np.bincount(v[r][v[r] >= 0], minlength=4)
        if cnt.sum() == 0:
            assert out.winner[r] == -1
        else:
            assert out.winner[r] == int(np.argmax(cnt)) and out.count[r] == cnt.max()
            assert out.unanimous[r] == (cnt.max() == cnt.sum())


def test_medoid_picks_central_vector():
    rng = np.random.default_rng(0)
    base = rng.normal(size=16).astype(np.float32)
    E = np.stack([base + 0.05 * rng.normal(size=16) for _ in range(5)] + [-base]).astype(np.float32)
    r = reducers.cosine_medoid(E)
    assert r.index != 5
    sims = (E / np.linalg.norm(E, axis=1, keepdims=True)) @ (E / np.linalg.norm(E, axis=1, keepdims=True)).T
    assert r.index == int(np.argmax(sims.sum(1) - 1))
    assert abs(r.score - (sims.sum(1)[r.index] - 1) / 5) < 1e-5


def test_judge_reduce_and_unanimity():
    s = np.array([[[0.1, 0.2], [0.9, 0.8], [0.0, 0.95]]], dtype=np.float32)
    out = reducers.judge_reduce_batch(s)
    assert out.winner[0] == 1 and not out.unanimous[0]
    u = reducers.unanimity(np.array([[0, -1], [-2, -1], [0, -3]], dtype=np.float32))
    assert not u.all_good and u.dissenters == [1]


def test_answer_extract_host():
    from llm_consensus_amd.consensus import reducers

    marker, choices = [10, 11], [65, 66, 67, 68]
    ids = np.array([[1, 10, 11, 66, 0, 0, 0, 0],        # B after the cue
                    [10, 11, 65, 5, 10, 11, 9, 68],     # last cue wins; D is 2 tokens after it
                    [65, 66, 67, 68, 1, 2, 3, 4],       # no cue
                    [10, 11, 1, 2, 3, 4, 5, 66],        # choice outside the window
                    [1, 2, 10, 11, 67, 9, 9, 9]], dtype=np.int32)
    lens = np.array([4, 8, 8, 8, 4], dtype=np.int32)       # row 4: the cue is cut off by its length
    got = reducers.answer_extract(ids, lens, marker, choices, window=3)
    assert got.tolist() == [1, 3, -1, -1, -1]
    assert reducers.answer_extract(ids, lens, [], choices).tolist() == [1, 3, 3, 1, -1]


def test_member_weights_flip_a_two_to_one_vote():
    """MemberConfig.weight reaches the reducer: one heavy member outvotes two light ones; with
    tie_break = highest_weight the weights only decide ties in count."""
    from llm_consensus_amd._native import core
    from llm_consensus_amd.consensus import reducers
    from llm_consensus_amd.consensus.backend import GenParams, ScriptedBackend
    from llm_consensus_amd.consensus.members import Member
    from llm_consensus_amd.consensus.strategies import MajorityVote

    votes = np.array([[1, 1, 2], [0, 3, -1]], dtype=np.int32)
    plain = reducers.majority_vote_batch(votes, 4)
    assert plain.winner.tolist() == [1, 0]
    heavy = reducers.majority_vote_batch(votes, 4, weights=np.array([1.0, 1.0, 2.5], dtype=np.float32))
    assert heavy.winner.tolist() == [2, 0] and heavy.count.tolist() == [1, 1]
    tie = reducers.majority_vote_batch(votes, 4, core.TieBreak.HighestWeight,
                                       weights=np.array([1.0, 3.0, 2.5], dtype=np.float32))
    assert tie.winner.tolist() == [1, 3]          # row 0: count decides; row 1: 1-1 tie goes to the heavier voter
    with pytest.raises(ValueError):
        reducers.majority_vote_batch(votes, 4, weights=np.ones(2, dtype=np.float32))

    personas = core.default_personas()
    answers = ["Answer: B", "Answer: B", "Answer: C"]
    members = [Member(personas[i], ScriptedBackend({"propose": [a]}, name=f"m{i}"), params=GenParams(max_new_tokens=4),
                      weight=w) for i, (a, w) in enumerate(zip(answers, (1.0, 1.0, 2.5)))]
    res = MajorityVote(members).ask("Which? A) a B) b C) c D) d")
    assert res.answer == "C"
    for m in members:
        m.weight = 1.0
    assert MajorityVote(members).ask("Which? A) a B) b C) c D) d").answer == "B"


def test_token_level_decisions_in_the_vote_strategies():
    """decision_mode="tokens": answers are classified from token ids with answer_extract (the first
    choice token after the cue), rows without the cue fall back to the text parser, and a backend
    without a tokenizer is parsed as text throughout."""
    from llm_consensus_amd.consensus.backend import Backend, GenParams, GenResult, Handle
    from llm_consensus_amd.consensus.members import Member
    from llm_consensus_amd.consensus.strategies import MajorityVote, SelfConsistency
    from llm_consensus_amd.utils.tokenizer import ByteLevelTokenizer

    class Canned(Backend):
        def __init__(self, replies, with_tokens=True):
            self.replies, self.tok, self.with_tokens, self.pending = replies, ByteLevelTokenizer(), with_tokens, []

        def token_ids_of(self, text):
            return list(self.tok.encode(text, add_bos=False)) if self.with_tokens else None

        def submit_generate(self, prompt, params, on_text=None):
            h = Handle()
            texts = [self.replies[(len(self.pending) + c) % len(self.replies)] for c in range(params.n)]
            self.pending.append((h, texts))
            return h

        def run(self):
            for h, texts in self.pending:
                ids = [list(self.tok.encode(t, add_bos=False)) for t in texts] if self.with_tokens else []
                h.set_result(GenResult(texts=texts, token_ids=ids))
            self.pending = []

    personas = core.default_personas()
    # "B ... Answer: C": the text parser and the token scan must both pick what follows the cue
    replies = ["I think B at first. Answer: C", "Answer:D", "The answer is (A).", "Answer: zzzz"]
    gp = GenParams(max_new_tokens=8)
    for with_tokens in (True, False):
        members = [Member(personas[i], Canned(replies[i:] + replies[:i], with_tokens), params=gp) for i in range(3)]
        res = MajorityVote(members, decision_mode="tokens").ask_many(["q0"])[0]
        want = [core.extract_choice(replies[i], "ABCD") for i in range(3)]
        assert [v["class"] for v in res.votes] == want == [2, 3, 0]
    sc = SelfConsistency(Member(personas[0], Canned(replies), params=gp), k_chains=4, decision_mode="tokens")
    out = sc.ask_many(["q0", "q1"])
    assert [v["class"] for v in out[0].votes] == [2, 3, 0, -1]
    assert [v["class"] for v in out[1].votes] == [3, 0, -1, 2]
    with pytest.raises(ValueError):
        MajorityVote(members, decision_mode="nope")
    # members whose calls return several chains (GenParams.n > 1): a member's vote is chain 0 of
    # ITS result for THAT question; the rows of the other chains must not shift later questions
    gp3 = GenParams(max_new_tokens=8, n=3)
    members = [Member(personas[i], Canned(["Answer: A", "Answer: B", "Answer: C", "Answer: D"]), params=gp3)
               for i in range(2)]
    res = MajorityVote(members, decision_mode="tokens").ask_many(["q0", "q1", "q2"])
    assert [[v["class"] for v in r.votes] for r in res] == [[0, 0], [1, 1], [2, 2]]
    # the two tokenisation forms of the choices interleave after the cue ("Answer: C" is form
    # " C", "Answer:D" is form "D"): the FIRST choice token wins, whichever form it has
    one = [Member(personas[0], Canned(["Answer:D then C", "Answer: C then:D"]), params=gp)]
    res = MajorityVote(one, decision_mode="tokens", cue_window=12).ask_many(["q0", "q1"])
    assert [r.votes[0]["class"] for r in res] == [3, 2]


@settings(max_examples=150, deadline=None, derandomize=True)
@given(st.integers(2, 9), st.integers(2, 24), st.integers(0, 10_000))
def test_medoid_matches_numpy(n, d, seed):
    rng = np.random.default_rng(seed)
    E = rng.normal(size=(n, d)).astype(np.float32)
    r = reducers.cosine_medoid(E)
    U = E / np.linalg.norm(E, axis=1, keepdims=True)
    s = (U @ U.T).sum(1) - 1.0
    assert 0 <= r.index < n and s[r.index] >= s.max() - 1e-4          # ties may go either way
    assert abs(r.score - s[r.index] / (n - 1)) < 1e-4


@settings(max_examples=150, deadline=None, derandomize=True)
@given(st.integers(1, 5), st.integers(1, 6), st.integers(1, 4), st.integers(0, 10_000))
def test_judge_reduce_matches_numpy(q, c, j, seed):
    rng = np.random.default_rng(seed)
    s = rng.normal(size=(q, c, j)).astype(np.float32)             # [question, candidate, judge]
    out = reducers.judge_reduce_batch(s)
    mean = s.mean(2)
    for i in range(q):
        assert mean[i, out.winner[i]] >= mean[i].max() - 1e-5
        assert abs(out.score[i] - mean[i, out.winner[i]]) < 1e-5
        assert bool(out.unanimous[i]) == bool((s[i].argmax(0) == out.winner[i]).all()) or c == 1


@settings(max_examples=150, deadline=None, derandomize=True)
@given(st.lists(st.tuples(st.floats(-20, 0), st.floats(-20, 0)), min_size=1, max_size=8))
def test_unanimity_matches_its_definition(rows):
    lp = np.array(rows, dtype=np.float32)                          # [member, (Good, NeedsRefinement)]
    u = reducers.unanimity(lp)
    dissent = [i for i in range(len(rows)) if lp[i, 1] > lp[i, 0]]
    assert u.dissenters == dissent and u.all_good == (not dissent)


@settings(max_examples=200, deadline=None, derandomize=True)
@given(st.lists(st.lists(st.integers(0, 9), min_size=1, max_size=24), min_size=1, max_size=6),
       st.lists(st.integers(0, 9), min_size=0, max_size=3), st.integers(1, 6))
def test_answer_extract_matches_a_python_scan(rows, marker, window):
    choices = [3, 7, 8]
    L = max(len(r) for r in rows)
    ids = np.zeros((len(rows), L), dtype=np.int32)
    lens = np.array([len(r) for r in rows], dtype=np.int32)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = r
    got = reducers.answer_extract(ids, lens, marker, choices, window=window).tolist()
    for i, r in enumerate(rows):
        start = 0
        if marker:
            hits = [k for k in range(len(r) - len(marker) + 1) if r[k:k + len(marker)] == marker]
            if not hits:
                assert got[i] == -1
                continue
            start = hits[-1] + len(marker)
            span = r[start:start + window]
            want = next((choices.index(t) for t in span if t in choices), -1)
        else:                                                       # no cue: the LAST choice token anywhere
            want = next((choices.index(t) for t in reversed(r) if t in choices), -1)
        assert got[i] == want, (r, marker, window)
