"""Properties of the placement planner (csrc/core/placement.cpp) over random ensembles. The
reference has nothing to place (one remote model, /root/reference/src/main.rs:82-86); here a wrong
plan is a model loaded onto a GPU that cannot hold it, or two ranks that disagree about who hosts
what, so the invariants are checked for every plan the planner calls feasible."""
from hypothesis import given, settings, strategies as st

from llm_consensus_amd._native import core

GB = 1000 ** 3
HBM = 288 * GB


def spec(name, key, weight_gb, kv_gb, tp, ep):
    m = core.MemberSpec()
    m.name, m.model_key, m.weight_bytes, m.min_kv_bytes, m.tp, m.ep = name, key, int(weight_gb * GB), int(kv_gb * GB), tp, ep
    return m


member = st.tuples(st.integers(0, 3),                      # model key: members may share a model
                   st.sampled_from([1, 8, 16, 47, 93, 141, 200, 400]),
                   st.sampled_from([1, 4, 16]),
                   st.sampled_from([1, 1, 1, 2, 4, 8]),
                   st.sampled_from([1, 1, 1, 2, 4, 8]))


@settings(max_examples=400, deadline=None, derandomize=True)
@given(st.lists(member, min_size=1, max_size=8), st.sampled_from([1, 2, 4, 8]), st.booleans(),
       st.sampled_from([0.0, 0.08, 0.2]))
def test_feasible_plans_keep_every_invariant(members, gpus, spread, reserve):
    # members that share a key share one model: same size and degrees
    by_key = {}
    specs = []
    for i, (key, w, kv, tp, ep) in enumerate(members):
        w, kv, tp, ep = by_key.setdefault(key, (w, kv, tp, ep))
        specs.append(spec(f"m{i}", f"model{key}", w, kv, tp, ep))
    policy = core.PlacementPolicy.Spread if spread else core.PlacementPolicy.Pack
    plan = core.plan_placement(specs, gpus, HBM, reserve, policy)
    if not plan.feasible:
        assert plan.error and not plan.replicas
        return
    cap = HBM * (1.0 - reserve)
    assert 1 <= plan.gpus_per_replica <= gpus and len(plan.replicas) >= 1
    assert len(plan.bytes_per_device) == gpus and all(b <= cap + 1 for b in plan.bytes_per_device)
    seen = set()
    for rep in plan.replicas:
        devs = list(rep.devices)
        assert len(devs) == plan.gpus_per_replica == len(set(devs)) and all(0 <= d < gpus for d in devs)
        assert not (set(devs) & seen), "replicas overlap"
        seen |= set(devs)
        assert [m.name for m in rep.members] == [s.name for s in specs], "every replica hosts every member once, in order"
        home = {}
        for m, s in zip(rep.members, specs):
            group = max(s.tp, s.ep)
            assert len(m.devices) == group == len(set(m.devices)) and set(m.devices) <= set(devs)
            # members of one model live on the same GPUs, and exactly the first of them owns the weights
            if s.model_key in home:
                assert m.shares_weights and list(m.devices) == home[s.model_key]
            else:
                assert not m.shares_weights
                home[s.model_key] = list(m.devices)
    assert set(plan.idle_devices) == set(range(gpus)) - seen
    # what the plan says a device holds is what its members add up to
    want = [0] * gpus
    for rep in plan.replicas:
        for m, s in zip(rep.members, specs):
            group = max(s.tp, s.ep)
            for d in m.devices:                 # a member that shares a model adds its KV room, not the weights again
                want[d] += (0 if m.shares_weights else -(-s.weight_bytes // group)) + -(-s.min_kv_bytes // group)
    for d in range(gpus):
        assert abs(plan.bytes_per_device[d] - want[d]) <= 8 * len(specs), (d, plan.bytes_per_device[d], want[d])


def test_known_shapes():
    # 3 x 16 GB on 8 GPUs: packed -> one GPU per replica, eight replicas; spread -> three GPUs per replica, two replicas
    three = [spec(f"m{i}", f"k{i}", 16, 4, 1, 1) for i in range(3)]
    p = core.plan_placement(three, 8, HBM, 0.08, core.PlacementPolicy.Pack)
    assert p.feasible and p.gpus_per_replica == 1 and len(p.replicas) == 8
    s = core.plan_placement(three, 8, HBM, 0.08, core.PlacementPolicy.Spread)
    assert s.feasible and s.gpus_per_replica == 3 and len(s.replicas) == 2 and sorted(s.idle_devices) == [6, 7]
    # a model that fits nowhere
    big = [spec("big", "k", 400, 4, 1, 1)]
    f = core.plan_placement(big, 8, HBM, 0.08, core.PlacementPolicy.Pack)
    assert not f.feasible and "big" in f.error
