"""Expert-parallel load balancing: lvllm_amd/eplb.py on the CPU.

* the policy against goldens produced by RUNNING the reference's DefaultEplbPolicy / compute_logical_maps
  (tests/golden/make_golden_eplb.py), including the known answers of the reference's own test
  (tests/distributed/test_eplb_algo.py:12-72) -- index for index;
* the reference's remaining policy tests re-stated (test_eplb_algo.py:75-300, 472-560);
* the transfer plan by simulation, and the weight exchange over gloo (world 2 and 4) the way the reference
  tests it (tests/distributed/test_eplb_execute.py: after the call every physical slot holds the weights of
  the logical expert the new map names);
* the EplbState loop (load window -> policy -> exchange -> maps) over gloo;
* the CPU restatement of the id-map / load-recording kernel (oracle.eplb_map_record) against the reference's
  own test of that kernel and against a slot-by-slot Python loop.  The HIP kernel itself is checked against the
  restatement on the GPU (tests/test_zz3_gpu_eplb.py).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lvllm_amd import eplb
from lvllm_amd.eplb import DefaultEplbPolicy as Policy
from oracle import oracle as orc

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "eplb.npz"))


# ------------------------------------------------------------------------------------------ policy
def test_known_answers_of_the_reference_test():
    p2l = Policy.rebalance_experts(torch.from_numpy(GOLD["ka_weight"]), 16, 4, 2, 8)
    assert p2l.dtype == torch.int64 and tuple(p2l.shape) == (2, 16)
    np.testing.assert_array_equal(p2l.numpy(), GOLD["ka_phy2log"])
    l2p, cnt = eplb.compute_logical_maps(p2l, 12)
    np.testing.assert_array_equal(cnt.numpy(), GOLD["ka_logcnt"])
    np.testing.assert_array_equal(l2p.numpy(), GOLD["ka_log2phy"])


@pytest.mark.parametrize("gi", range(int(GOLD["n_geometries"])))
def test_policy_equals_the_reference_run(gi):
    L, E, P, G, N, R = GOLD[f"g{gi}_geom"].tolist()
    w0, w1 = torch.from_numpy(GOLD[f"g{gi}_w0"]), torch.from_numpy(GOLD[f"g{gi}_w1"])
    p0 = Policy.rebalance_experts(w0, P, G, N, R)
    np.testing.assert_array_equal(p0.numpy(), GOLD[f"g{gi}_p0"])
    np.testing.assert_array_equal(Policy.rebalance_experts(w1, P, G, N, R).numpy(), GOLD[f"g{gi}_p1_free"])
    # second round with the previous placement: experts that stay on a GPU keep their slot
    p1 = Policy.rebalance_experts(w1, P, G, N, R, p0)
    np.testing.assert_array_equal(p1.numpy(), GOLD[f"g{gi}_p1"])
    l2p, cnt = eplb.compute_logical_maps(p0, E)
    np.testing.assert_array_equal(l2p.numpy(), GOLD[f"g{gi}_l2p0"])
    np.testing.assert_array_equal(cnt.numpy(), GOLD[f"g{gi}_cnt0"])
    # numpy input, same answer
    np.testing.assert_array_equal(Policy.rebalance_experts(GOLD[f"g{gi}_w0"], P, G, N, R).numpy(), GOLD[f"g{gi}_p0"])


def test_building_blocks_equal_the_reference_run():
    for ci in range(int(GOLD["n_bp"])):
        pack, rank = Policy.balanced_packing(GOLD[f"bp{ci}_w"], int(GOLD[f"bp{ci}_packs"]))
        np.testing.assert_array_equal(pack, GOLD[f"bp{ci}_pack"])
        np.testing.assert_array_equal(rank, GOLD[f"bp{ci}_rank"])
    for ci in range(int(GOLD["n_re"])):
        p2l, cnt = Policy.replicate_experts(GOLD[f"re{ci}_w"], int(GOLD[f"re{ci}_phy"]))
        np.testing.assert_array_equal(p2l, GOLD[f"re{ci}_p2l"])
        np.testing.assert_array_equal(cnt, GOLD[f"re{ci}_cnt"])


def _check_placement(p2l, E, P):
    a = p2l.numpy()
    assert a.shape[1] == P and a.min() >= 0 and a.max() < E
    for row in a:
        assert set(row.tolist()) == set(range(E)), "every logical expert needs at least one replica"


def test_reference_policy_cases_restated():
    # single GPU (test_eplb_algo.py:75-95)
    p = Policy.rebalance_experts(torch.tensor([[10, 20, 30, 40]]), 4, 1, 1, 1)
    assert set(p[0].tolist()) == {0, 1, 2, 3}
    # equal weights (:98-119): every expert once, deterministic
    w = torch.full((1, 8), 50)
    p = Policy.rebalance_experts(w, 8, 4, 2, 4)
    _, cnt = eplb.compute_logical_maps(p, 8)
    assert torch.all(cnt == 1)
    assert torch.equal(p, Policy.rebalance_experts(w, 8, 4, 2, 4))
    # extreme imbalance (:122-142): the hot expert gets the redundant replicas
    w = torch.tensor([[1000, 1, 1, 1, 1, 1, 1, 1]])
    p = Policy.rebalance_experts(w, 12, 4, 2, 4)
    _, cnt = eplb.compute_logical_maps(p, 8)
    assert int(cnt.sum()) == 12 and torch.all(cnt >= 1) and cnt[0, 0] == int(cnt.max()) and cnt[0, 0] >= 2
    # several layers (:145-175)
    w = torch.tensor([[10, 20, 30, 40, 50, 60], [60, 50, 40, 30, 20, 10], [25, 25, 25, 25, 25, 25]])
    p = Policy.rebalance_experts(w, 8, 2, 1, 2)
    _check_placement(p, 6, 8)
    # small hierarchical (:196-223)
    w = torch.tensor([[100, 50, 200, 75, 150, 25, 300, 80]])
    p = Policy.rebalance_experts(w, 12, 4, 2, 4)
    _, cnt = eplb.compute_logical_maps(p, 8)
    assert int(cnt.sum()) == 12 and torch.all(cnt >= 1) and cnt[0, 6] >= 2
    # groups do not divide over the nodes: global policy (:226-245)
    p = Policy.rebalance_experts(torch.tensor([[10, 20, 30, 40, 50, 60]]), 8, 3, 2, 4)
    _, cnt = eplb.compute_logical_maps(p, 6)
    assert tuple(p.shape) == (1, 8) and int(cnt.sum()) == 8
    # additional cases (:270-300)
    w1 = torch.tensor([[50, 100, 75, 120, 90, 60, 80, 110, 40, 70, 95, 85, 65, 55, 45, 35]])
    _, c1 = eplb.compute_logical_maps(Policy.rebalance_experts(w1, 24, 8, 4, 8), 16)
    assert int(c1.sum()) == 24
    w2 = torch.tensor([[200, 150, 100, 50, 25, 12], [12, 25, 50, 100, 150, 200]])
    _, c2 = eplb.compute_logical_maps(Policy.rebalance_experts(w2, 10, 3, 1, 2), 6)
    for layer in range(2):
        assert c2[layer, int(torch.argmax(w2[layer]))] >= 2


def test_parameter_validation():
    # physical experts must divide over the GPUs (test_eplb_algo.py:178-193 expects an error as well)
    with pytest.raises((ValueError, AssertionError)):
        Policy.rebalance_experts(torch.tensor([[10, 20, 30, 40]]), 5, 1, 1, 2)
    with pytest.raises(ValueError):
        Policy.replicate_experts(np.ones((1, 4), np.float32), 3)
    with pytest.raises(ValueError):
        Policy.balanced_packing(np.ones((1, 5), np.float32), 2)


def test_balance_quality_on_skewed_loads():
    """Zipf loads over 256 experts, 288 physical on 8 GPUs: the heaviest GPU carries < 1.15 x the mean
    (linear placement of the same loads: > 1.5 x)."""
    rng = np.random.default_rng(3)
    E, P, R = 256, 288, 8
    w = (1.0 / np.arange(1, E + 1)) ** 1.0
    w = rng.permutation(w)[None, :].astype(np.float32) * 1e6
    p = Policy.rebalance_experts(w, P, 8, 1, R).numpy()[0]
    _, cnt = eplb.compute_logical_maps(torch.from_numpy(p), E)
    per_phys = w[0][p] / cnt.numpy()[p]
    gpu = per_phys.reshape(R, -1).sum(axis=1)
    linear = w[0].reshape(R, -1).sum(axis=1)
    assert gpu.max() / gpu.mean() < 1.15 < 1.5 < linear.max() / linear.mean()


def test_ties_give_a_valid_deterministic_placement():
    rng = np.random.default_rng(5)
    w = rng.integers(0, 4, size=(3, 64)).astype(np.float32)           # many exact ties, zeros included
    a = Policy.rebalance_experts(w, 80, 8, 1, 8)
    b = Policy.rebalance_experts(w.copy(), 80, 8, 1, 8)
    assert torch.equal(a, b)
    _check_placement(a, 64, 80)


@pytest.mark.parametrize("case", [
    # (old, new, ranks, expected): expected = what the reference's preserve_intragpu_slots returns for these inputs
    # (run through tests/golden/make_golden_eplb.py: load_policy); cf. test_eplb_algo.py:472-560
    ([[0, 1, 2, 3]], [[1, 0, 3, 2]], 2, [[0, 1, 2, 3]]),                   # same GPU, other slot: slot kept
    ([[0, 1, 2, 3]], [[2, 3, 0, 1]], 2, [[2, 3, 0, 1]]),                   # everything moves GPU: unchanged
    ([[0, 1, 0, 2]], [[1, 0, 2, 0]], 2, [[0, 1, 0, 2]]),                   # duplicates
    ([[0, 1, 2, 3, 4, 5]], [[2, 6, 0, 5, 7, 3]], 2, [[0, 6, 2, 3, 7, 5]]),  # partial overlap: stayers keep slots
])
def test_preserve_intragpu_slots(case):
    old, new, ranks, expected = case
    got = Policy.preserve_intragpu_slots(np.array(new), ranks, np.array(old))
    np.testing.assert_array_equal(got, np.array(expected))
    per = len(new[0]) // ranks
    for r in range(ranks):                                                 # per GPU the multiset is untouched
        assert sorted(got[0, r * per:(r + 1) * per]) == sorted(new[0][r * per:(r + 1) * per])


def test_logical_maps_and_initial_placement():
    for k in range(3):
        l2p, cnt = eplb.compute_logical_maps(torch.from_numpy(GOLD[f"lm{k}_p2l"]), 4)
        np.testing.assert_array_equal(l2p.numpy(), GOLD[f"lm{k}_l2p"])
        np.testing.assert_array_equal(cnt.numpy(), GOLD[f"lm{k}_cnt"])
    # fixed map width: same replicas, -1 padding
    l2p, cnt = eplb.compute_logical_maps(torch.from_numpy(GOLD["lm1_p2l"]), 4, max_slots=5)
    assert tuple(l2p.shape) == (2, 4, 5)
    R = GOLD["lm1_l2p"].shape[2]
    np.testing.assert_array_equal(l2p.numpy()[:, :, :R], GOLD["lm1_l2p"])
    assert (l2p.numpy()[:, :, R:] == -1).all()
    with pytest.raises(ValueError):
        eplb.compute_logical_maps(torch.from_numpy(GOLD["lm1_p2l"]), 4, max_slots=1)
    with pytest.raises(ValueError):
        eplb.compute_logical_maps(torch.tensor([[0, 7]]), 4)
    for ci in range(int(GOLD["n_init"])):
        e, r, *want = GOLD[f"init{ci}"].tolist()
        assert eplb.build_initial_global_physical_to_logical_map(e, r) == want


# ------------------------------------------------------------------------------------------ plan
def _simulate(old, new, ranks, plan):
    """apply a plan to slot contents (content = logical id) with staging semantics"""
    per = len(old) // ranks
    cur = np.array(old).copy()
    staged = {}
    for sr, ss, dr, ds, e in plan.p2p:
        assert old[sr * per + ss] == e, "sender does not hold what it sends"
        assert sr != dr
        staged[(dr, ds)] = e
    for r, ss, ds, e in plan.local:
        assert old[r * per + ss] == e
        staged[(r, ds)] = e
    for r, prim, ds, e in plan.fanout:
        assert staged[(r, prim)] == e
        staged[(r, ds)] = e
    for (r, ds), e in staged.items():
        cur[r * per + ds] = e
    return cur


def test_plan_reaches_the_new_placement_and_spreads_senders():
    rng = np.random.default_rng(11)
    for E, P, R, G in [(8, 16, 8, 1), (64, 80, 8, 8), (256, 288, 8, 8), (12, 16, 4, 4), (16, 16, 2, 1)]:
        w0 = rng.random((1, E)).astype(np.float32) ** 4
        w1 = rng.random((1, E)).astype(np.float32) ** 4
        old = Policy.rebalance_experts(w0, P, G, 1, R).numpy()[0]
        new = Policy.rebalance_experts(w1, P, G, 1, R, torch.from_numpy(old[None])).numpy()[0]
        plan = eplb.plan_layer_transfers(old, new, R)
        np.testing.assert_array_equal(_simulate(old, new, R, plan), new)
        per = P // R
        # one receive per (rank, expert); nothing is sent to a rank that already holds the expert
        seen = set()
        for sr, ss, dr, ds, e in plan.p2p:
            assert (dr, e) not in seen
            seen.add((dr, e))
            assert e not in old[dr * per:(dr + 1) * per]
        # unchanged slots are not touched
        touched = {(dr, ds) for _, _, dr, ds, _ in plan.p2p} | {(r, ds) for r, _, ds, _ in plan.local} | \
                  {(r, ds) for r, _, ds, _ in plan.fanout}
        for p in range(P):
            assert ((p // per, p % per) in touched) == (old[p] != new[p])
    # a replicated expert's receivers are dealt over its holders
    old = np.array([0, 1, 0, 2, 0, 3, 4, 5, 6, 7, 8, 9])      # expert 0 on ranks 0, 1, 2 (per = 2)
    new = np.array([0, 1, 0, 2, 0, 3, 0, 5, 0, 7, 0, 9])      # ranks 3, 4, 5 each need expert 0
    plan = eplb.plan_layer_transfers(old, new, 6)
    assert sorted(sr for sr, *_ in plan.p2p) == [0, 1, 2]
    assert plan.egress(6) == [1, 1, 1, 0, 0, 0]
    # the balance carries across the layers of one rearrangement: layer 1 starts with the least-loaded holder
    eg = [0] * 6
    eplb.plan_layer_transfers(old, np.array([0, 1, 0, 2, 0, 3, 0, 5, 6, 7, 8, 9]), 6, eg)      # rank 3 <- expert 0 from rank 0
    nxt = eplb.plan_layer_transfers(old, np.array([0, 1, 0, 2, 0, 3, 4, 5, 0, 7, 8, 9]), 6, eg)  # rank 4 <- expert 0
    assert eg == [1, 1, 0, 0, 0, 0] and nxt.p2p[0][0] == 1
    # no change, no traffic; an expert nobody holds is an error
    assert eplb.plan_layer_transfers(old, old, 6) == eplb.LayerPlan()
    with pytest.raises(ValueError):
        eplb.plan_layer_transfers(np.array([0, 1, 2, 3]), np.array([0, 1, 2, 9]), 2)


def test_plan_with_empty_slots_and_fanout():
    old = np.array([0, 1, 2, -1, 3, -1])            # 2 ranks x 3 slots
    new = np.array([3, 3, 0, 2, 2, -1])             # rank 0 needs 3 twice (one receive + fan-out); rank 1 needs 2 twice
    plan = eplb.plan_layer_transfers(old, new, 2)
    assert len(plan.p2p) == 2 and len(plan.fanout) == 2
    cur = _simulate(old, new, 2, plan)
    np.testing.assert_array_equal(cur[new >= 0], new[new >= 0])


# ------------------------------------------------------------------------------------------ exchange
def _canon(layer: int, e: int, shapes):
    """weights of logical expert e of a layer: a deterministic function of (layer, e), distinct over e within a
    layer, exact in bf16 and uint8"""
    return [torch.full(s, float((37 * layer + 5 * e + i) % 251), dtype=dt) for i, (s, dt) in enumerate(shapes)]


SHAPES = [((6, 4), torch.bfloat16), ((3,), torch.float32), ((5, 2), torch.uint8)]


def _make_store(layer, logical_ids):
    ts = []
    for i, (s, dt) in enumerate(SHAPES):
        t = torch.zeros((len(logical_ids), *s), dtype=dt)
        for j, e in enumerate(logical_ids):
            if e >= 0:
                t[j] = _canon(layer, e, SHAPES)[i]
        ts.append(t)
    return eplb.TensorExpertStore(ts)


def _store_matches(store, layer, logical_ids):
    for j, e in enumerate(logical_ids):
        if e < 0:
            continue
        for t, c in zip(store.tensors, _canon(layer, e, SHAPES)):
            if not torch.equal(t[j], c):
                return False
    return True


def test_exchange_single_rank():
    old = np.array([[0, 1, 2, 3, 0, 1], [3, 2, 1, 0, 3, 2]])
    new = np.array([[3, 1, 0, 2, 2, 1], [0, 1, 2, 3, 0, 0]])
    stores = [_make_store(l, old[l]) for l in range(2)]
    assert stores[0].expert_nbytes == 6 * 4 * 2 + 3 * 4 + 10
    plans = eplb.rearrange_expert_weights_inplace(old, new, stores, None, rank=0, world=1)
    assert all(not p.p2p for p in plans)
    for l in range(2):
        assert _store_matches(stores[l], l, new[l])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, world, port, q, *args)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _placements(world, L, E, P, seed):
    rng = np.random.default_rng(seed)
    w0 = rng.random((L, E)).astype(np.float32) ** 3
    w1 = rng.random((L, E)).astype(np.float32) ** 3
    old = Policy.rebalance_experts(w0, P, 1, 1, world)
    new = Policy.rebalance_experts(w1, P, 1, 1, world, old)
    return old.numpy(), new.numpy()


def _exchange_worker(rank, world, port, q, L, E, P, budget):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        old, new = _placements(world, L, E, P, seed=17)
        per = P // world
        stores = [_make_store(l, old[l, rank * per:(rank + 1) * per]) for l in range(L)]
        plans = eplb.rearrange_expert_weights_inplace(old, new, stores, None, max_staging_bytes=budget)
        ok = all(_store_matches(stores[l], l, new[l, rank * per:(rank + 1) * per]) for l in range(L))
        # a second call with old == new moves nothing and leaves the weights alone
        again = eplb.rearrange_expert_weights_inplace(new, new, stores, None)
        ok = ok and all(p == eplb.LayerPlan() for p in again)
        ok = ok and all(_store_matches(stores[l], l, new[l, rank * per:(rank + 1) * per]) for l in range(L))
        q.put((rank, (ok, sum(len(p.p2p) for p in plans))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,L,E,P,budget", [(2, 3, 8, 12, 8 << 30), (4, 4, 16, 24, 8 << 30),
                                                 (4, 5, 16, 24, 200)])      # 200 B: one layer per batch
def test_exchange_over_gloo(world, L, E, P, budget):
    res = _spawn(_exchange_worker, world, L, E, P, budget)
    assert all(ok for ok, _ in res.values()), res
    assert len({n for _, n in res.values()}) == 1 and next(iter(res.values()))[1] > 0, "no expert crossed ranks: weak test"


def _state_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, E, red = 2, 8, 4
        st = eplb.EplbState(L, E, red, window_size=4, step_interval=3)
        P, per = E + red, (E + red) // world
        st.expert_stores = [_make_store(l, st.local_logical_ids(l)) for l in range(L)]
        init = st.physical_to_logical_map.clone()
        # every rank records the same skew: logical expert 5 takes most tokens (recorded on its physical slots)
        ran = []
        for step in range(3):
            for l in range(L):
                view = st.layer_state(l).expert_load_view
                for p in range(P):
                    view[p] += 100 if int(init[l, p]) == 5 else 1 + (p % 3)
            ran.append(st.step())
        load = None
        ok = ran == [False, False, True]
        p2l = st.physical_to_logical_map
        l2p, cnt = eplb.compute_logical_maps(p2l, E, max_slots=red + 1)
        ok = ok and torch.equal(st.logical_to_physical_map.cpu().long(), l2p) and \
            torch.equal(st.logical_replica_count.cpu().long(), cnt)
        ok = ok and bool((cnt[:, 5] == cnt.max(dim=1).values).all()) and bool((cnt[:, 5] >= 2).all())
        ok = ok and all(_store_matches(st.expert_stores[l], l, st.local_logical_ids(l)) for l in range(L))
        ok = ok and int(st.expert_load_pass.abs().sum()) == 0 and st.rearrangement_step == 0
        q.put((rank, (ok, p2l.tolist(), st.balancedness())))
    finally:
        dist.destroy_process_group()


def test_state_loop_over_gloo():
    res = _spawn(_state_worker, 2)
    assert all(ok for ok, _, _ in res.values()), res
    assert res[0][1] == res[1][1], "ranks disagree on the placement"
    assert res[0][2] == res[1][2] and 0.0 < res[0][2] <= 1.0


def test_state_single_process_and_dummy_steps():
    st = eplb.EplbState(1, 4, 0, window_size=2, step_interval=100)
    assert st.physical_to_logical_map.tolist() == [[0, 1, 2, 3]]
    assert st.logical_to_physical_map.tolist() == [[[0], [1], [2], [3]]]
    st.layer_state(0).expert_load_view += torch.tensor([1, 2, 3, 4], dtype=torch.int32)
    st.step(is_dummy=True)                                                  # dropped
    assert int(st.expert_load_window.sum()) == 0
    st.layer_state(0).expert_load_view += torch.tensor([1, 2, 3, 4], dtype=torch.int32)
    st.step()
    assert st.global_logical_load().tolist() == [[1.0, 2.0, 3.0, 4.0]]
    with pytest.raises(RuntimeError):
        st.rearrange()                                                      # no expert stores attached
    # resuming a saved placement
    saved = torch.tensor([[2, 0, 1, 3, 0, 2], [3, 2, 1, 0, 1, 1]])
    st2 = eplb.EplbState(2, 4, 2, initial_physical_to_logical_map=saved)
    assert torch.equal(st2.physical_to_logical_map, saved)
    assert st2.logical_replica_count.tolist() == [[2, 1, 2, 1], [1, 3, 1, 1]]
    assert st2.logical_to_physical_map[0, 0].tolist() == [1, 4, -1] and st2.logical_to_physical_map[1, 1].tolist() == [2, 4, 5]
    with pytest.raises(ValueError):
        eplb.EplbState(1, 4, 2, initial_physical_to_logical_map=torch.tensor([[0, 1, 2, 2, 1, 0]]))     # expert 3 missing
    with pytest.raises(ValueError):
        eplb.EplbState(1, 4, 2, initial_physical_to_logical_map=torch.tensor([[0, 1, 2, 3]]))           # wrong width


# ------------------------------------------------------------------------------------------ id map restatement
def test_map_restatement_reproduces_the_reference_test_setup():
    """tests/kernels/moe/test_routing.py:155-188: identity map, one replica -> ids unchanged; load = histogram"""
    rng = np.random.default_rng(0)
    E = 64
    ids = rng.integers(0, E, size=(33, 6))
    phys, load = orc.eplb_map_record(ids, np.arange(E)[:, None], np.ones(E, np.int64), np.zeros(E, np.int32))
    np.testing.assert_array_equal(phys, ids)
    np.testing.assert_array_equal(load, np.bincount(ids.reshape(-1), minlength=E))


def test_map_restatement_against_a_slot_loop():
    rng = np.random.default_rng(1)
    E, P, M, K = 16, 24, 41, 4
    w = rng.random((1, E)).astype(np.float32) ** 4
    p2l = Policy.rebalance_experts(w, P, 1, 1, 8)
    l2p, cnt = eplb.compute_logical_maps(p2l, E, max_slots=P - E + 1)
    l2p, cnt = l2p[0].numpy(), cnt[0].numpy()
    ids = rng.integers(-1, E + 1, size=(M, K))                     # includes -1 and one id past the end
    base = rng.integers(0, 5, size=P).astype(np.int32)
    for enabled, unpadded in [(True, None), (True, 17), (False, None), (True, 0)]:
        phys, load = orc.eplb_map_record(ids, l2p, cnt, base, enabled, unpadded)
        want_load = base.copy()
        for i in range(M * K):
            t, e = i // K, int(ids.reshape(-1)[i])
            if 0 <= e < E:
                want = int(l2p[e, ((t * 2654435769) & 0xFFFFFFFF) % max(int(cnt[e]), 1)])
                assert want >= 0 and p2l[0, want] == e
            else:
                want = -1
            assert phys.reshape(-1)[i] == want
            if enabled and want >= 0 and (unpadded is None or i < unpadded * K):
                want_load[want] += 1
        np.testing.assert_array_equal(load, want_load)
    # all slots of a token use the same replica rank
    hot = int(np.argmax(cnt))
    phys, _ = orc.eplb_map_record(np.full((9, 3), hot), l2p, cnt)
    assert (phys == phys[:, :1]).all() and len(set(phys[:, 0].tolist())) > 1


# ------------------------------------------------------------------------------------------ EP + EPLB end to end
def _e2e_worker(rank, world, port, q):
    """route logical ids -> physical ids (CPU restatement as the test double of the HIP kernel) -> expert-parallel
    all-to-all over the PHYSICAL experts (lvllm_amd/ep.py, linear placement of P slots) -> local experts from the
    tensors the EPLB exchange maintains; before and after a rearrangement the result equals the single-rank
    oracle on the logical ids."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lvllm_amd.ep import ExpertParallelExperts
        from tests.helpers import make_routing, torch_to_bits
        from tests.helpers import TorchEpKernels
        E, red, K, H, I, M = 8, 4, 2, 64, 32, 24
        P, per = E + red, (E + red) // world
        g = torch.Generator().manual_seed(5)
        w13 = (torch.randn((E, 2 * I, H), generator=g) / 4).to(torch.bfloat16)
        w2 = (torch.randn((E, H, I), generator=g) / 4).to(torch.bfloat16)
        st = eplb.EplbState(1, E, red, window_size=2, step_interval=1)
        mine = torch.tensor(st.local_logical_ids(0))
        local13, local2 = w13[mine].contiguous(), w2[mine].contiguous()          # this rank's physical slots
        st.expert_stores = [eplb.TensorExpertStore([local13, local2])]
        d_loc = orc.MoeDesc(E=per, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)

        def local_compute(x, lids, ws, out_dtype):
            if x.shape[0] == 0:
                return torch.zeros((0, H), dtype=out_dtype)
            return torch.from_numpy(orc.moe(d_loc, torch_to_bits(local13), torch_to_bits(local2), torch_to_bits(x.contiguous()),
                                            lids.contiguous().numpy(), ws.contiguous().numpy())).to(out_dtype)
        ep = ExpertParallelExperts(local_compute, P, H, mode="a2a", kernels=TorchEpKernels, return_dtype=torch.float32)
        gx = torch.Generator().manual_seed(100 + rank)
        x = (torch.randn((M, H), generator=gx) / 2).to(torch.bfloat16)
        tw, ids = make_routing(M, E, K, seed=200 + rank, skew=2.0)                 # skewed: expert 0 is hot
        d_all = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        want = orc.moe(d_all, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(x), ids, tw)

        def forward():
            ls = st.layer_state(0)
            phys, load = orc.eplb_map_record(ids, ls.logical_to_physical_map.numpy(), ls.logical_replica_count.numpy(),
                                             ls.expert_load_view.numpy())
            ls.expert_load_view.copy_(torch.from_numpy(load))
            return ep.forward(x, torch.from_numpy(tw), torch.from_numpy(phys)).numpy()
        ok = bool(np.allclose(forward(), want, atol=1e-5, rtol=1e-5))
        before = st.physical_to_logical_map.clone()
        ok = ok and st.step()                                                     # interval 1: rearrangement
        moved = not torch.equal(before, st.physical_to_logical_map)
        ok = ok and bool(np.allclose(forward(), want, atol=1e-5, rtol=1e-5))
        hot = int(torch.bincount(torch.from_numpy(ids).reshape(-1).long(), minlength=E).argmax())
        q.put((rank, (ok, moved, int(st.logical_replica_count[0, hot]), int(st.logical_replica_count.max()))))
    finally:
        dist.destroy_process_group()


def test_ep_all_to_all_over_rebalanced_physical_experts():
    res = _spawn(_e2e_worker, 2)
    assert all(ok for ok, *_ in res.values()), res
    assert all(moved for _, moved, *_ in res.values()), "the rearrangement changed nothing: weak test"
    assert all(c_hot == c_max >= 2 for _, _, c_hot, c_max in res.values()), res


# ------------------------------------------------------------------------------------------ overlapped rearrangement
def _overlap_worker(rank, world, port, q):
    """overlap=True: a due step only posts the first batch; `commit_after_steps` steps later the batch is imported,
    ITS layers' maps are switched and the next batch is posted.  Invariant checked after every step: each layer's
    weights match the placement its maps describe."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, E, red = 3, 8, 4
        st = eplb.EplbState(L, E, red, window_size=4, step_interval=2, overlap=True, commit_after_steps=2,
                            max_staging_bytes=1)                                    # one layer per batch
        st.expert_stores = [_make_store(l, st.local_logical_ids(l)) for l in range(L)]
        init = st.physical_to_logical_map.clone()
        P = E + red

        def consistent():
            l2p, cnt = eplb.compute_logical_maps(st.physical_to_logical_map, E, max_slots=red + 1)
            return (torch.equal(st.logical_to_physical_map.long(), l2p) and torch.equal(st.logical_replica_count.long(), cnt)
                    and all(_store_matches(st.expert_stores[l], l, st.local_logical_ids(l)) for l in range(L)))
        ok, switched_at, commits = True, [], 0
        for step in range(40):
            for l in range(L):
                view = st.layer_state(l).expert_load_view
                for p_ in range(P):
                    view[p_] += 50 * (l + 1) if int(init[l, p_]) == (2 * l + 1) % E else 1 + (p_ % 3)
            changed = st.step()
            commits += int(changed)
            ok = ok and consistent()
            n_switched = sum(int(not torch.equal(init[l], st.physical_to_logical_map[l])) for l in range(L))
            switched_at.append(n_switched)
            if step == 0:
                ok = ok and not st.in_flight and not changed
            if step == 1:
                ok = ok and st.in_flight and not changed and n_switched == 0      # posted, nothing switched yet
            if commits == L:
                break
        # posted at step 1; batches land every 2 steps: steps 3, 5, 7
        ok = ok and commits == L and len(switched_at) == 8 and switched_at[-1] >= 1 and switched_at == sorted(switched_at)
        st.drain()                                                                 # a second round may have been posted
        ok = ok and consistent() and not st.in_flight
        hot_ok = all(int(st.logical_replica_count[l, (2 * l + 1) % E]) >= 2 for l in range(L))
        q.put((rank, (ok, hot_ok, st.physical_to_logical_map.tolist())))
    finally:
        dist.destroy_process_group()


def test_overlapped_rearrangement_over_gloo():
    res = _spawn(_overlap_worker, 2)
    assert all(ok for ok, _, _ in res.values()), res
    assert all(hot for _, hot, _ in res.values()), res
    assert res[0][2] == res[1][2], "ranks disagree on the placement"


def test_plan_on_arbitrary_placements():
    """placements that no policy produced: duplicates on a rank, empty slots, experts that vanish, many ranks"""
    rng = np.random.default_rng(123)
    for case in range(300):
        ranks = int(rng.choice([1, 2, 3, 4, 8]))
        per = int(rng.integers(1, 7))
        E = int(rng.integers(1, 12))
        P = ranks * per
        old = rng.integers(-1, E, size=P)
        present = sorted(set(old[old >= 0].tolist()))
        if not present:
            continue
        new = rng.choice(np.array(present + [-1]), size=P)              # only experts somebody holds (or empty)
        plan = eplb.plan_layer_transfers(old, new, ranks)
        cur = _simulate(old, new, ranks, plan)
        np.testing.assert_array_equal(cur[new >= 0], new[new >= 0], err_msg=f"case {case}")
        np.testing.assert_array_equal(cur[new < 0], old[new < 0])      # slots that become empty are left alone
        assert sum(plan.egress(ranks)) == len(plan.p2p)
        # nothing is received that the rank already had
        for sr, ss, dr, ds, e in plan.p2p:
            assert e not in old[dr * per:(dr + 1) * per]


def test_exchange_accepts_the_reference_expert_weights_lists():
    old = np.array([[0, 1, 2, 3]])
    new = np.array([[3, 0, 1, 2]])
    st = _make_store(0, old[0])
    eplb.rearrange_expert_weights_inplace(old, new, [st.tensors], None, rank=0, world=1)      # raw tensor lists
    assert _store_matches(st, 0, new[0])


def test_exchange_with_empty_slots_single_rank():
    old = np.array([[0, -1, 2, 1]])
    new = np.array([[2, 0, -1, 0]])
    store = _make_store(0, old[0])
    eplb.rearrange_expert_weights_inplace(old, new, [store], None, rank=0, world=1)
    assert _store_matches(store, 0, new[0])


def _grouped_state_worker(rank, world, port, q):
    """4 ranks, 16 experts in 4 groups (group-limited routing models: DeepSeek-V3 / GLM), 8 redundant slots: the
    hierarchical policy inside the loop; every rank ends with the same maps and the weights its slots name"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, E, red = 2, 16, 8
        st = eplb.EplbState(L, E, red, num_groups=4, num_nodes=1, window_size=3, step_interval=3)
        st.expert_stores = [_make_store(l, st.local_logical_ids(l)) for l in range(L)]
        P = E + red
        init = st.physical_to_logical_map.clone()
        rng = np.random.default_rng(7)                       # the same loads on every rank (they are summed anyway)
        per_logical = (rng.random((L, E)) ** 4 * 1000).astype(np.int64) + 1
        for step in range(3):
            for l in range(L):
                view = st.layer_state(l).expert_load_view
                cnt = torch.bincount(init[l], minlength=E)
                view += torch.from_numpy(per_logical[l])[init[l]].div(cnt[init[l]], rounding_mode="floor").to(torch.int32)
            ran = st.step()
        p2l = st.physical_to_logical_map
        ok = ran and not torch.equal(p2l, init)
        ok = ok and all(sorted(set(p2l[l].tolist())) == list(range(E)) for l in range(L))
        ok = ok and all(_store_matches(st.expert_stores[l], l, st.local_logical_ids(l)) for l in range(L))
        # the load per rank after the rearrangement (per-replica loads) is flatter than before
        def imbalance(m):
            worst = 0.0
            for l in range(L):
                c = torch.bincount(m[l], minlength=E).double()
                load = (torch.from_numpy(per_logical[l]).double() / c)[m[l]].view(world, -1).sum(1)
                worst = max(worst, float(load.max() / load.mean()))
            return worst
        q.put((rank, (ok, p2l.tolist(), imbalance(init), imbalance(p2l))))
    finally:
        dist.destroy_process_group()


def test_state_loop_with_expert_groups_over_four_ranks():
    res = _spawn(_grouped_state_worker, 4)
    assert all(ok for ok, *_ in res.values()), res
    assert len({str(v[1]) for v in res.values()}) == 1, "ranks disagree on the placement"
    before, after = res[0][2], res[0][3]
    assert after < before and after < 1.25, (before, after)
