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
        assert a1q_scale is None and meta is None and ids_d.shape == (world * M, K) and a1q.shape == (world * M, H)
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
