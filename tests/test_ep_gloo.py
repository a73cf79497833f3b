"""Expert-parallel data path on world_size 2 / 3 over gloo (CPU).  The local expert computation is
injected (the CPU oracle stands in for the HIP engine -- tests may do that); what is under test is
the sharding logic of lvllm_amd/ep.py: placement, split sizes, both all-to-all directions, combine."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as orc
from tests.helpers import TorchEpKernels, make_routing, torch_to_bits

E, K, H, I, M = 8, 2, 64, 32, 13


def _weights():
    g = torch.Generator().manual_seed(5)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 4).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g) / 4).to(torch.bfloat16)
    return w13, w2


def _tokens(rank):
    g = torch.Generator().manual_seed(100 + rank)
    a = (torch.randn((M, H), generator=g) / 2).to(torch.bfloat16)
    tw, ids = make_routing(M, E, K, seed=200 + rank, drop=0.1)
    return a, torch.from_numpy(tw), torch.from_numpy(ids)


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lvllm_amd.ep import ExpertParallelExperts
        w13, w2 = _weights()
        real_mode = "a2a" if mode.startswith("a2a") else mode
        ep = ExpertParallelExperts(lambda *a: None, E, H, mode=real_mode, kernels=TorchEpKernels,
                                   fixed_max_tokens=0 if mode == "a2a_ragged" else 1024,
                                   return_dtype=torch.float32 if mode != "a2a_bf16" else None,
                                   validate_uniform=True)
        lo = ep.first_expert[rank]
        n_loc = ep.local_num
        d = orc.MoeDesc(E=n_loc, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        w13l, w2l = torch_to_bits(w13[lo:lo + n_loc]), torch_to_bits(w2[lo:lo + n_loc])

        def local_compute(x, lids, ws, out_dtype):
            if x.shape[0] == 0:
                return torch.zeros((0, H), dtype=out_dtype)
            y = orc.moe(d, w13l, w2l, torch_to_bits(x.contiguous()), lids.contiguous().numpy(), ws.contiguous().numpy())
            return torch.from_numpy(y).to(out_dtype)

        ep.local_compute = local_compute
        a, tw, ids = _tokens(rank)
        out = ep.forward(a, tw, ids)
        q.put((rank, out.numpy()))
    finally:
        dist.destroy_process_group()


def _pf_worker(rank, world, port, q):
    """the reference's modular call sequence (modular_kernel.py:1219-1420): prepare -> experts.apply -> finalize"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lvllm_amd.modular import LkmPrepareAndFinalize, _NoOpReduce
        from lvllm_amd.ops import determine_expert_map
        w13, w2 = _weights()
        pf = LkmPrepareAndFinalize(E, H, kernels=TorchEpKernels)
        assert pf.num_dispatchers() == world and pf.output_is_reduced() and pf.topk_indices_dtype() == torch.int32
        n_loc, emap = determine_expert_map(world, rank, E, "linear")
        lo = pf._ep.first_expert[rank]
        d = orc.MoeDesc(E=n_loc, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        a, tw, ids = _tokens(rank)
        a1q, a1q_scale, meta, ids_d, w_d = pf.prepare(a, tw, ids, E, emap, False, None, True)
        assert a1q_scale is None and meta.valid_den == world and ids_d.shape == (world * M, K) and a1q.shape == (world * M, H)
        with pytest.raises(RuntimeError, match="before the matching finalize"):   # one exchange in flight per instance
            pf.prepare(a, tw, ids, E, emap, False, None, True)
        # what LkmExperts.apply does with them: expert_map, then the weighted expert rows (the oracle stands in)
        g = ids_d.contiguous().to(torch.int64)
        lids = torch.where(g >= 0, emap[g.clamp(min=0)].to(torch.int64), torch.full_like(g, -1)).to(torch.int32)
        assert ((lids >= 0) == (g >= 0)).all(), "a dispatched id reached a rank that does not own its expert"
        fused = torch.from_numpy(orc.moe(d, torch_to_bits(w13[lo:lo + n_loc]), torch_to_bits(w2[lo:lo + n_loc]),
                                         torch_to_bits(a1q.contiguous()), lids.numpy(), w_d.contiguous().numpy()))
        out = torch.empty((M, H), dtype=torch.float32)
        pf.finalize(out, fused, tw, ids, False, _NoOpReduce())
        q.put((rank, out.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_prepare_finalize_pair_matches_single_rank_oracle(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pf_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w13, w2 = _weights()
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    for r in range(world):
        a, tw, ids = _tokens(r)
        ref = orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(a), ids.numpy(), tw.numpy())
        np.testing.assert_allclose(results[r], ref, atol=1e-5, rtol=1e-5)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["a2a", "a2a_bf16", "a2a_ragged", "ar"])
def test_ep_matches_single_rank_oracle(world, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w13, w2 = _weights()
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    for r in range(world):
        a, tw, ids = _tokens(r)
        ref = orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(a), ids.numpy(), tw.numpy())
        if mode == "a2a_bf16":
            # the default return leg: each visited rank's partial row is rounded to bf16 once (<= 2^-9 relative
            # per partial, K partials at most) before the fp32 sum
            np.testing.assert_allclose(results[r], ref, atol=K * 2.0 ** -8 * np.abs(ref).max(), rtol=0)
        else:
            # identical per-row arithmetic, only the fp32 sum over K slots may regroup (per rank, then over ranks)
            np.testing.assert_allclose(results[r], ref, atol=1e-5, rtol=1e-5)


def _uneven_worker(rank, world, port, q):
    """ranks hold DIFFERENT token counts (one of them none at all) under a common record capacity"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lvllm_amd.ep import ExpertParallelExperts
        w13, w2 = _weights()
        ep = ExpertParallelExperts(lambda *a: None, E, H, mode="a2a", kernels=TorchEpKernels, capacity_tokens=16,
                                   return_dtype=torch.float32, validate_uniform=True)
        lo, n_loc = ep.first_expert[rank], ep.local_num
        d = orc.MoeDesc(E=n_loc, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)

        def local_compute(x, lids, ws, out_dtype):
            y = orc.moe(d, torch_to_bits(w13[lo:lo + n_loc]), torch_to_bits(w2[lo:lo + n_loc]),
                        torch_to_bits(x.contiguous()), lids.contiguous().numpy(), ws.contiguous().numpy())
            return torch.from_numpy(y).to(out_dtype)
        ep.local_compute = local_compute
        a, tw, ids = _tokens(rank)
        n = [M, 0, 5][rank]
        out = ep.forward(a[:n], tw[:n], ids[:n])
        q.put((rank, out.numpy()))
    finally:
        dist.destroy_process_group()


def test_common_capacity_serves_unequal_token_counts():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w13, w2 = _weights()
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    for r, n in enumerate([M, 0, 5]):
        a, tw, ids = _tokens(r)
        assert results[r].shape == (n, H)
        if n:
            ref = orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(a[:n]), ids[:n].numpy(), tw[:n].numpy())
            np.testing.assert_allclose(results[r], ref, atol=1e-5, rtol=1e-5)


def test_pack_records_capacity_and_overflow_accounting():
    """the record layout and the counted overflow of the token-granular pack (torch restatement; the HIP kernel must
    equal it bit for bit: tests/test_gpu_ep.py)"""
    from lvllm_amd.ep import owner_of
    Mx, Kx, Hx, Ex, ep = 21, 4, 32, 16, 4
    g = torch.Generator().manual_seed(3)
    hidden = torch.randn((Mx, Hx), generator=g).to(torch.bfloat16)
    ids = torch.randint(-1, Ex, (Mx, Kx), generator=g, dtype=torch.int32)
    tw = torch.rand((Mx, Kx), generator=g)
    rowb = TorchEpKernels.ep_row_bytes(Hx, Kx)
    assert rowb % 16 == 0 and rowb >= Hx * 2 + Kx * 8
    for cap in (Mx, 7):
        send = torch.zeros((ep, cap, rowb), dtype=torch.uint8)
        slot_of = torch.full((ep, Mx), -7, dtype=torch.int32)
        overflow = torch.zeros(1, dtype=torch.int32)
        TorchEpKernels.ep_pack_tokens(hidden, tw, ids, Ex, ep, cap, send, slot_of, overflow)
        own = owner_of(ids, Ex, ep)
        dropped = 0
        for p in range(ep):
            want_tokens = [m for m in range(Mx) if (own[m] == p).any()]
            kept, lost = want_tokens[:cap], want_tokens[cap:]
            dropped += len(lost)
            assert [m for m in range(Mx) if slot_of[p, m] >= 0] == kept
            assert [int(slot_of[p, m]) for m in kept] == list(range(len(kept)))      # ascending token order
            rec_ids = send[p, :, Hx * 2:Hx * 2 + 4 * Kx].contiguous().view(torch.int32)
            assert (rec_ids[len(kept):] == -1).all()
            first = ep_first = p * (Ex // ep)
            for c, m in enumerate(kept):
                assert torch.equal(send[p, c, :Hx * 2].view(torch.bfloat16), hidden[m])
                want = torch.where(own[m] == p, ids[m] - first, torch.full_like(ids[m], -1))
                assert torch.equal(rec_ids[c], want.to(torch.int32))
        assert int(overflow) == dropped and (cap < Mx) == (dropped > 0)


def test_owner_of_matches_expert_map():
    from lvllm_amd.ep import owner_of
    for Eg in (8, 10, 7, 256):
        for ep in (1, 2, 3, 4, 8):
            ids = torch.arange(-1, Eg, dtype=torch.int32)
            own = owner_of(ids, Eg, ep)
            assert own[0] == -1
            for r in range(ep):
                _, emap = orc.expert_map(ep, r, Eg, 0)
                np.testing.assert_array_equal((own[1:] == r).numpy(), emap >= 0)


def _grouped_worker(rank, world, port, slack, skew, q):
    """group-limited routing (one expert group per rank, every token picks its experts inside `topk_group` groups):
    the capacity below the worst case, its counted overflow and the eager fallback to the exact bound"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lvllm_amd.ep import ExpertParallelExperts
        Eg, Kg, Mg, n_group, topk_group = 16, 4, 24, world, 2
        g = torch.Generator().manual_seed(7)
        w13 = (torch.randn((Eg, 2 * I, H), generator=g) / 4).to(torch.bfloat16)
        w2 = (torch.randn((Eg, H, I), generator=g) / 4).to(torch.bfloat16)
        ep = ExpertParallelExperts(lambda *a: None, Eg, H, mode="a2a", kernels=TorchEpKernels,
                                   return_dtype=torch.float32, routing_groups=(n_group, topk_group),
                                   capacity_slack=slack, check_overflow=True)
        lo, n_loc = ep.first_expert[rank], ep.local_num
        d = orc.MoeDesc(E=n_loc, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        seen = []

        def local_compute(x, lids, ws, out_dtype, valid_den=None):
            seen.append((x.shape[0], valid_den))
            y = orc.moe(d, torch_to_bits(w13[lo:lo + n_loc]), torch_to_bits(w2[lo:lo + n_loc]),
                        torch_to_bits(x.contiguous()), lids.contiguous().numpy(), ws.contiguous().numpy())
            return torch.from_numpy(y).to(out_dtype)
        ep.local_compute = local_compute
        ep._lc_takes_den = True
        # routing: token m picks `topk_group` groups (= ranks) and K experts inside them: spread evenly (12 records per
        # destination), or every token of the rank (of ONE rank: the decision must still be collective) on groups 0 and 1
        gr = torch.Generator().manual_seed(50 + rank)
        per = Eg // n_group
        ids = torch.empty((Mg, Kg), dtype=torch.int32)
        for m in range(Mg):
            skewed = skew == "all" or (skew == "one_rank" and rank == 0)
            groups = torch.tensor([0, 1]) if skewed else torch.tensor([m % n_group, (m + 1) % n_group])
            pool = torch.cat([torch.arange(per) + int(gi) * per for gi in groups])
            ids[m] = pool[torch.randperm(len(pool), generator=gr)[:Kg]].to(torch.int32)
        tw = torch.rand((Mg, Kg), generator=gr) + 0.1
        a = (torch.randn((Mg, H), generator=gr) / 2).to(torch.bfloat16)
        cap = ep.capacity_for(Mg, None, Kg)
        out = ep.forward(a, tw, ids)
        wb = ep.wire_bytes(Mg, Kg)
        dfull = orc.MoeDesc(E=Eg, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        ref = orc.moe(dfull, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(a), ids.numpy(), tw.numpy())
        q.put((rank, cap, ep.overflow_count(), [s[0] for s in seen], [s[1] for s in seen], wb["dispatch_bytes"],
               float(np.abs(out.numpy() - ref).max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("slack,skew", [(1.25, "none"), (1.0, "one_rank"), (0.5, "all")])
def test_group_limited_capacity_and_overflow_fallback(slack, skew):
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grouped_worker, args=(r, world, port, slack, skew, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Mg, rowb = 24, TorchEpKernels.ep_row_bytes(H, 4)
    for rank, cap, ov, rows_seen, dens, disp_bytes, err in res:
        # ranks per token <= topk_group = 2 of 4 ranks: capacity = ceil(24 * 2 * slack / 4)
        assert cap == -(-int(Mg * 2 * slack) // world)
        assert disp_bytes == (world - 1) * cap * rowb
        assert all(dn == world for dn in dens)              # the engine is told how sparse the records are
        assert err < 1e-5                                    # the result is exact either way
        if skew == "none":
            assert cap == 15 and rows_seen == [world * cap] and ov == 0   # 0.625 x the worst case of 24 record slots
        else:
            # some rank had 24 records for a destination with `cap` slots: counted there, agreed by all, and the step ran
            # once at the exact bound (capacity = tokens) on EVERY rank
            assert rows_seen == [world * Mg]
            assert (ov > 0) == (skew == "all" or rank == 0)


def test_wire_bytes_of_baseline_config3_with_group_limited_capacity():
    """BASELINE.json configs[3] (DeepSeek-V3 routing, 8 groups = 8 ranks, top-4 groups, M = 32 tokens per rank): the
    default capacity with `routing_groups` puts <= 0.63 x the bytes of the worst-case capacity on the wire"""
    from lvllm_amd.ep import ExpertParallelExperts, ranks_per_token
    assert ranks_per_token(8, 8, 4, 8) == 4 and ranks_per_token(8, 4, 2) == 4 and ranks_per_token(2, 8, 4) == 2
    a = ExpertParallelExperts(lambda *x: None, 256, 7168, kernels=TorchEpKernels)
    b = ExpertParallelExperts(lambda *x: None, 256, 7168, kernels=TorchEpKernels, routing_groups=(8, 4))
    a.ep = b.ep = 8                        # (no process group here: the accounting only)
    wa, wb = a.wire_bytes(32, 8), b.wire_bytes(32, 8)
    assert wa["capacity_tokens"] == 32 and wb["capacity_tokens"] == 20
    assert wb["dispatch_bytes"] / wa["dispatch_bytes"] == pytest.approx(0.625)
    assert wb["return_bytes"] / wa["return_bytes"] == pytest.approx(0.625)


def _unequal_no_capacity_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lvllm_amd.ep import ExpertParallelExperts
        ep = ExpertParallelExperts(lambda *a: None, E, H, mode="a2a", kernels=TorchEpKernels)
        Mr = 8 + rank                                  # ranks disagree on the token count and name no common capacity
        a = torch.zeros((Mr, H), dtype=torch.bfloat16)
        ids = torch.zeros((Mr, K), dtype=torch.int32)
        tw = torch.ones((Mr, K))
        try:
            ep.forward(a, tw, ids)
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_unequal_token_counts_without_a_common_capacity_fail_loudly():
    """rank-local token counts size the collectives when no capacity is named; ranks that disagree are told so
    (one all-gather per new token count, outside capture) instead of hanging in mismatched exchanges"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_unequal_no_capacity_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        assert "different token count" in res[rank] and "[8, 9]" in res[rank], res


# ------------------------------------------------------------------------------------------------ ADVICE r3
def _unequal_routing_groups_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lvllm_amd.ep import ExpertParallelExperts
        ep = ExpertParallelExperts(lambda *a: None, E, H, mode="a2a", kernels=TorchEpKernels, routing_groups=(world, 1))
        Mr = 8 + 4 * rank          # the group-limited capacity is derived from the rank-local token count: 8 vs 12 tokens
        a = torch.zeros((Mr, H), dtype=torch.bfloat16)
        ids = torch.zeros((Mr, K), dtype=torch.int32)
        tw = torch.ones((Mr, K))
        try:
            ep.forward(a, tw, ids)
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_unequal_token_counts_with_routing_groups_fail_loudly():
    """`routing_groups` derives the record capacity from the rank-local token count, so the uniformity check
    must stay on (ranks with different counts would size send / recv / back differently and evaluate the overflow
    re-run differently: mismatched collectives)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_unequal_routing_groups_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        assert "different token count" in res[rank] and "[8, 12]" in res[rank], res


def _named_capacity_overflow_worker(rank, world, port, q):
    """a NAMED common capacity below some rank's token count under group-limited routing: the overflow decision and the
    capacity of the re-run must come from group-agreed values (the largest token count of the group), whatever the
    rank-local count is"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lvllm_amd.ep import ExpertParallelExperts
        Eg, Kg, n_group = 8, 2, world
        g = torch.Generator().manual_seed(7)
        w13 = (torch.randn((Eg, 2 * I, H), generator=g) / 4).to(torch.bfloat16)
        w2 = (torch.randn((Eg, H, I), generator=g) / 4).to(torch.bfloat16)
        ep = ExpertParallelExperts(lambda *a: None, Eg, H, mode="a2a", kernels=TorchEpKernels, return_dtype=torch.float32,
                                   routing_groups=(n_group, 1))           # check_overflow defaults to True with routing_groups
        assert ep.check_overflow
        lo, n_loc = ep.first_expert[rank], ep.local_num
        d = orc.MoeDesc(E=n_loc, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        seen = []

        def local_compute(x, lids, ws, out_dtype, valid_den=None):
            seen.append(x.shape[0])
            y = orc.moe(d, torch_to_bits(w13[lo:lo + n_loc]), torch_to_bits(w2[lo:lo + n_loc]),
                        torch_to_bits(x.contiguous()), lids.contiguous().numpy(), ws.contiguous().numpy())
            return torch.from_numpy(y).to(out_dtype)
        ep.local_compute = local_compute
        ep._lc_takes_den = True
        Mr = 20 if rank == 0 else 6                     # rank 0 sends all 20 tokens to rank 1's experts: 20 records > 8 slots
        gr = torch.Generator().manual_seed(90 + rank)
        per = Eg // world
        ids = (torch.randint(0, per, (Mr, Kg), generator=gr) + per * (1 if rank == 0 else rank)).to(torch.int32)
        tw = torch.rand((Mr, Kg), generator=gr) + 0.1
        a = (torch.randn((Mr, H), generator=gr) / 2).to(torch.bfloat16)
        out = ep.forward(a, tw, ids, capacity=8)
        dfull = orc.MoeDesc(E=Eg, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        ref = orc.moe(dfull, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(a), ids.numpy(), tw.numpy())
        q.put((rank, seen, ep.overflow_count(), float(np.abs(out.numpy() - ref).max())))
    finally:
        dist.destroy_process_group()


def test_named_capacity_overflow_reruns_at_the_group_maximum():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_named_capacity_overflow_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, seen, ov, err in res:
        assert seen == [world * 20], (rank, seen)       # ONE expert pass, at the group's largest token count, on every rank
        assert (ov > 0) == (rank == 0) and err < 1e-5


def test_exchange_pool_in_flight_guard_and_pool_tags():
    """the exchange buffers are per (device, group, pool tag), not per instance -- two instances on one pool
    must not both have a dispatch in flight; different tags give independent memory"""
    from lvllm_amd import ep as epm
    def copy(out, inp):
        out.copy_(inp)
    mk = lambda tag: epm.ExpertParallelExperts(lambda *a: None, E, H, mode="a2a", kernels=TorchEpKernels, transport=copy,
                                               return_dtype=torch.float32, pool_tag=tag)
    a, tw, ids = _tokens(0)
    ids = ids.clamp(min=0)
    x, y, z = mk(""), mk(""), mk("mb1")
    rows, rids, rws, hx = x.dispatch_fixed(a, tw, ids, return_handle=True)
    with pytest.raises(RuntimeError, match="already holds a dispatch"):
        y.dispatch_fixed(a, tw, ids)                     # same pool: refused until x combined
    rows_z, _, _, hz = z.dispatch_fixed(a, tw, ids, return_handle=True)       # another tag: its own memory
    assert rows_z.data_ptr() != rows.data_ptr() and torch.equal(rows_z, rows)
    fake = torch.zeros((x.ep * hx[3], H), dtype=torch.float32)
    x.combine_fixed(fake, a.size(0), handle=hx)
    z.combine_fixed(fake, a.size(0), handle=hz)
    r2 = y.dispatch_fixed(a, tw, ids, return_handle=True)                      # free again
    y.abandon_dispatch(r2[3])
    x.dispatch_fixed(a, tw, ids)
    x.abandon_dispatch()


def _two_microbatch_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lvllm_amd.ep import ExpertParallelExperts, forward_two_microbatches
        w13, w2 = _weights()
        eps = [ExpertParallelExperts(lambda *a: None, E, H, mode="a2a", kernels=TorchEpKernels, return_dtype=torch.float32,
                                     pool_tag=t) for t in ("mb0", "mb1", "seq")]
        lo, n_loc = eps[0].first_expert[rank], eps[0].local_num
        d = orc.MoeDesc(E=n_loc, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        w13l, w2l = torch_to_bits(w13[lo:lo + n_loc]), torch_to_bits(w2[lo:lo + n_loc])

        def local_compute(x, lids, ws, out_dtype):
            y = orc.moe(d, w13l, w2l, torch_to_bits(x.contiguous()), lids.contiguous().numpy(), ws.contiguous().numpy())
            return torch.from_numpy(y).to(out_dtype)
        for e in eps:
            e.local_compute = local_compute
        b0, b1 = _tokens(rank), _tokens(rank + 10)
        shared = lambda x: x.float() * 2.0                  # noqa: E731  (stands in for the always-on expert)
        o0, o1, s0, s1 = forward_two_microbatches(eps[0], eps[1], b0, b1, shared=shared)
        want0, want1 = eps[2].forward(*b0).clone(), eps[2].forward(*b1).clone()
        same = bool(torch.equal(o0, want0) and torch.equal(o1, want1) and torch.equal(s0, b0[0].float() * 2) and torch.equal(s1, b1[0].float() * 2))
        # the pools are free again, and the same tag twice is refused
        try:
            forward_two_microbatches(eps[0], eps[0], b0, b1)
            refused = False
        except ValueError:
            refused = True
        o0b, _ = forward_two_microbatches(eps[1], eps[0], b0, b1)
        q.put((rank, same, refused, bool(torch.equal(o0b, want0))))
    finally:
        dist.destroy_process_group()


def test_two_microbatches_in_flight_equal_the_sequential_steps():
    """SURVEY 7 / VERDICT r3 item 7b: the return exchange of micro-batch 0 is issued before the experts of micro-batch 1
    run (on the communicator stream on a GPU); the outputs are those of the two steps run one after the other, bit for
    bit, whichever pool serves which batch"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_microbatch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, refused, swapped in res:
        assert same and refused and swapped, (rank, same, refused, swapped)


def test_captured_capacity_uses_the_capture_slack():
    """while a step is being captured the group-limited capacity is sized for < 1e-4 overflow per pair
    (slack 1.75) instead of the eager path's 1.25 + fallback; BASELINE configs[3]: 28 of 32 record slots -> <= 0.9 x the
    bytes of the worst-case capacity the captured step used before"""
    from lvllm_amd.ep import ExpertParallelExperts
    a = ExpertParallelExperts(lambda *x: None, 256, 7168, kernels=TorchEpKernels)
    b = ExpertParallelExperts(lambda *x: None, 256, 7168, kernels=TorchEpKernels, routing_groups=(8, 4))
    a.ep = b.ep = 8
    assert b.capacity_for(32, None, 8) == 20 and b.capacity_for(32, None, 8, capturing=True) == 28
    wa, wc = a.wire_bytes(32, 8), b.wire_bytes(32, 8, capturing=True)
    assert wc["capacity_tokens"] == 28 and wc["dispatch_bytes"] / wa["dispatch_bytes"] <= 0.9
    # Binomial(32, 1/2) records per (source, destination) pair: P(X > 28)
    from math import comb
    p_over = sum(comb(32, k) for k in range(29, 33)) / 2 ** 32
    assert p_over < 1e-4 / 56, p_over


# ------------------------------------------------------------------ round 5: autotune agreement, micro-batch overflow, pool hygiene
class _FakeTunedEngine:
    """stands in for lk_moe_api._MOE's tuned_plans() / set_tuned_plan() (the real one needs a GPU): what each rank's
    first-call autotune 'chose' for two step shapes"""

    def __init__(self, plans):
        self.plans = dict(plans)

    def tuned_plans(self):
        return sorted(self.plans.items())

    def set_tuned_plan(self, key, index):
        assert key in self.plans
        self.plans[key] = index

    def describe(self):
        return " ".join(f"{k}:{v}" for k, v in sorted(self.plans.items()))


def _agree_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lvllm_amd.ep import agree_tuned_plans
        # two layers' engines; the ranks' timings picked different winners for (layer 0, M=128) and (layer 1, M=32)
        mine = [[{128: 3, 32: 0}, {32: 2}], [{128: 1, 32: 0}, {32: 5}]][rank]
        engs = [_FakeTunedEngine(p) for p in mine]
        changed = agree_tuned_plans(engs)
        desc = [e.describe() for e in engs]
        # a rank that warmed up a shape the others did not: loud
        bad = [_FakeTunedEngine({128: 0, **({64: 1} if rank == 1 else {})})]
        try:
            agree_tuned_plans(bad)
            loud = False
        except RuntimeError as e:
            loud = "different step shapes" in str(e)
        q.put((rank, changed, desc, loud))
    finally:
        dist.destroy_process_group()


def test_autotuned_plans_are_agreed_across_the_group():
    """round-4 verdict item 8: the first-call autotune is timing-dependent, so two ranks of an expert-parallel group may
    pick different plans (different fp32 summation orders inside one group).  ep.agree_tuned_plans makes every rank take
    the MIN candidate index per (engine, step shape): afterwards the ranks report the same describe()."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, ch0, d0, loud0), (_, ch1, d1, loud1) = res
    assert d0 == d1 == ["32:0 128:1", "32:2"], (d0, d1)
    assert (ch0, ch1) == (1, 1) and loud0 and loud1


def _two_microbatch_overflow_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lvllm_amd.ep import ExpertParallelExperts, forward_two_microbatches
        w13, w2 = _weights()
        # group-limited capacity far below what the (ungrouped) routing of _tokens() needs: records WILL overflow
        mk = lambda t, **kw: ExpertParallelExperts(lambda *a: None, E, H, mode="a2a", kernels=TorchEpKernels,   # noqa: E731
                                                   return_dtype=torch.float32, pool_tag=t, **kw)
        tight = dict(routing_groups=(8, 1), capacity_slack=1.0)
        eps = [mk("ov0", **tight), mk("ov1", **tight), mk("ovseq")]
        lo, n_loc = eps[0].first_expert[rank], eps[0].local_num
        d = orc.MoeDesc(E=n_loc, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        w13l, w2l = torch_to_bits(w13[lo:lo + n_loc]), torch_to_bits(w2[lo:lo + n_loc])

        def local_compute(x, lids, ws, out_dtype):
            y = orc.moe(d, w13l, w2l, torch_to_bits(x.contiguous()), lids.contiguous().numpy(), ws.contiguous().numpy())
            return torch.from_numpy(y).to(out_dtype)
        for e in eps:
            e.local_compute = local_compute
        b0, b1 = _tokens(rank), _tokens(rank + 10)
        cap = eps[0].capacity_for(M, None, K)
        o0, o1 = forward_two_microbatches(eps[0], eps[1], b0, b1)
        want0, want1 = eps[2].forward(*b0).clone(), eps[2].forward(*b1).clone()
        dropped = eps[0].overflow_count() + eps[1].overflow_count()
        # ... and the experts raising mid-way leave both pools free for the next step
        def boom(*a, **k):
            raise ValueError("experts failed")
        eps[1].local_compute = boom
        try:
            forward_two_microbatches(eps[0], eps[1], b0, b1)
            raised = False
        except ValueError:
            raised = True
        eps[1].local_compute = local_compute
        o0b, o1b = forward_two_microbatches(eps[0], eps[1], b0, b1)
        q.put((rank, cap, dropped, bool(torch.equal(o0, want0) and torch.equal(o1, want1)), raised,
               bool(torch.equal(o0b, want0) and torch.equal(o1b, want1))))
    finally:
        dist.destroy_process_group()


def test_two_microbatches_rerun_on_overflow_and_free_their_pools_on_errors():
    """forward_two_microbatches sized its capacity from routing_groups but never looked at the overflow
    counters -- an eager step could return rows with dropped contributions.  It now checks collectively after both
    dispatches and re-dispatches BOTH at the exact bound; an exception between dispatch and combine frees both pools."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_microbatch_overflow_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, cap, dropped, same, raised, again in res:
        assert cap < M and dropped > 0, (rank, cap, dropped)        # the tight capacity really overflowed ...
        assert same and raised and again, (rank, same, raised, again)   # ... and the outputs are the exact ones


def test_failed_dispatch_leaves_the_pool_free():
    """_IN_FLIGHT was set before the pack kernel and the transport ran; an exception there left the
    shared pool marked busy for every layer"""
    from lvllm_amd import ep as epm

    def broken(out, inp):
        raise OSError("transport down")

    def copy(out, inp):
        out.copy_(inp)
    a, tw, ids = _tokens(0)
    ids = ids.clamp(min=0)
    bad = epm.ExpertParallelExperts(lambda *x: None, E, H, mode="a2a", kernels=TorchEpKernels, transport=broken,
                                    return_dtype=torch.float32, pool_tag="fail")
    good = epm.ExpertParallelExperts(lambda *x: None, E, H, mode="a2a", kernels=TorchEpKernels, transport=copy,
                                     return_dtype=torch.float32, pool_tag="fail")
    with pytest.raises(OSError):
        bad.dispatch_fixed(a, tw, ids)
    h = good.dispatch_fixed(a, tw, ids, return_handle=True)[3]      # same pool: not "already holds a dispatch"
    good.abandon_dispatch(h)
