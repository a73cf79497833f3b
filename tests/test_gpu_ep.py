"""Expert-parallel data path on the GPU: the two exchange kernels against their torch restatement
(bit for bit), the strided engine entry point against the contiguous one (bit for bit), the whole step over a
one-rank RCCL group -- eager AND captured in a hipGraph, both modes -- and over TWO ranks that share the one GPU of
the test box (real HIP kernels and engines on both ranks; the collective itself is staged through gloo because
RCCL refuses two ranks on one device).  The >1-GPU RCCL runs are the driver's (bench.py --gpus N)."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.helpers import TorchEpKernels, make_routing, torch_to_bits

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
DEV = "cuda:0"


def _ops():
    from lvllm_amd import ops
    return ops


@pytest.mark.parametrize("ep,cap", [(1, 37), (2, 37), (3, 40), (8, 37), (4, 9)])
@pytest.mark.parametrize("global_ids", [False, True])
def test_pack_kernel_equals_torch_restatement(ep, cap, global_ids):
    ops = _ops()
    M, K, H, E = 37, 4, 256, 16
    g = torch.Generator().manual_seed(ep * 10 + cap)
    hidden = torch.randn((M, H), generator=g).to(torch.bfloat16)
    ids = torch.randint(-1, E + 2, (M, K), generator=g, dtype=torch.int32)       # incl. -1 and ids >= E
    tw = torch.rand((M, K), generator=g)
    rowb = ops.ep_row_bytes(H, K)
    assert rowb == TorchEpKernels.ep_row_bytes(H, K)
    send = torch.full((ep, cap, rowb), 0xAB, dtype=torch.uint8, device=DEV)
    slot_of = torch.full((ep, M), -9, dtype=torch.int32, device=DEV)
    overflow = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.ep_pack_tokens(hidden.to(DEV), tw.to(DEV), ids.to(DEV), E, ep, cap, send, slot_of, overflow, global_ids)
    rsend = torch.full((ep, cap, rowb), 0xAB, dtype=torch.uint8)
    rslot = torch.full((ep, M), -9, dtype=torch.int32)
    rover = torch.zeros(1, dtype=torch.int32)
    TorchEpKernels.ep_pack_tokens(hidden, tw, ids, E, ep, cap, rsend, rslot, rover, global_ids)
    assert torch.equal(slot_of.cpu(), rslot) and int(overflow.item()) == int(rover.item())
    got = send.cpu()
    id_cols = slice(H * 2, H * 2 + 4 * K)
    assert torch.equal(got[:, :, id_cols], rsend[:, :, id_cols])                   # ids of EVERY record slot (-1 when unused)
    for p in range(ep):
        n = int((rslot[p] >= 0).sum())
        assert torch.equal(got[p, :n, :H * 2 + 8 * K], rsend[p, :n, :H * 2 + 8 * K])   # used records: all three fields


@pytest.mark.parametrize("back_dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("out_dt", [torch.float32, torch.bfloat16])
def test_combine_kernel_equals_torch_restatement(back_dt, out_dt):
    ops = _ops()
    ep, cap, M, H = 4, 19, 23, 520
    g = torch.Generator().manual_seed(1)
    back = torch.randn((ep, cap, H), generator=g).to(back_dt)
    slot_of = torch.randint(-1, cap, (ep, M), generator=g, dtype=torch.int32)
    slot_of[:, 5] = -1                                                             # a token no rank computed
    out = torch.full((M, H), 7.0, dtype=out_dt, device=DEV)
    ops.ep_combine(back.to(DEV), slot_of.to(DEV), out)
    want = TorchEpKernels.ep_combine(back, slot_of, torch.empty((M, H), dtype=out_dt))
    assert torch.equal(out.cpu(), want)
    assert (out.cpu()[5] == 0).all()


def _engine(E, H, I, K, seed=0, fmt="bf16"):
    ops = _ops()
    g = torch.Generator().manual_seed(seed)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 8).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g) / 8).to(torch.bfloat16)
    return ops.RoutedExpertsEngine(w13.to(DEV), w2.to(DEV), top_k=K, act_dtype=torch.bfloat16), w13, w2


@pytest.mark.parametrize("M", [1, 7, 70, 300])
def test_strided_rows_equal_contiguous_rows_bit_for_bit(M):
    """lkm_forward_strided on the record layout of the exchange == lkm_decode / lkm_prefill_device on copies"""
    ops = _ops()
    E, H, I, K = 6, 256, 128, 3
    eng, _, _ = _engine(E, H, I, K)
    g = torch.Generator().manual_seed(M)
    x = (torch.randn((M, H), generator=g) / 2).to(torch.bfloat16)
    tw, ids = make_routing(M, E, K, seed=M, drop=0.15)
    tw, ids = torch.from_numpy(tw), torch.from_numpy(ids)
    rowb = ops.ep_row_bytes(H, K)
    rec = torch.zeros((M, rowb), dtype=torch.uint8)
    rec[:, :H * 2] = x.view(torch.uint8).view(M, H * 2)
    first = 5                                                                      # records carry ids + first
    gid = torch.where(ids >= 0, ids + first, ids)
    gid[0, 0] = first + E                                                          # not one of this engine's experts
    ref_ids = ids.clone()
    ref_ids[0, 0] = -1
    rec[:, H * 2:H * 2 + 4 * K] = gid.contiguous().view(torch.uint8).view(M, 4 * K)
    rec[:, H * 2 + 4 * K:H * 2 + 8 * K] = tw.contiguous().view(torch.uint8).view(M, 4 * K)
    rec = rec.to(DEV)
    rows = rec[:, :H * 2].view(torch.bfloat16)
    rids = rec[:, H * 2:H * 2 + 4 * K].view(torch.int32)
    rws = rec[:, H * 2 + 4 * K:H * 2 + 8 * K].view(torch.float32)
    assert rows.stride(0) == rowb // 2 and (M == 1 or not rows.is_contiguous())
    y32 = eng.forward_rows(rows, rws, rids, out_dtype=torch.float32, id_offset=first)
    y16 = eng.forward_rows(rows, rws, rids, out_dtype=torch.bfloat16, id_offset=first)
    assert torch.equal(y32, eng.decode(x.to(DEV), tw.to(DEV), ref_ids.to(DEV)))
    assert torch.equal(y16, eng.prefill(x.to(DEV), tw.to(DEV), ref_ids.to(DEV)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_one_rank_rccl_step_eager_and_captured():
    """a2a (fp32 and bf16 return) and ar over a ONE-rank RCCL group == the engine called directly; the a2a step is
    then captured in a hipGraph (communicator created before the capture) and replayed on new inputs.  Runs in a
    subprocess under a timeout: a wedged communicator must not take the test session with it."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", PYTHONPATH=str(ROOT))
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "ep_rccl_one_rank.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    tags = ["a2a-f32 eager OK", "a2a-bf16 eager OK", "ar eager OK", "a2a captured OK", "layer ar OK",
            "modular prepare/apply/finalize OK", "two micro-batches overlapped OK"]
    if (ROOT / "oracle" / "_ref" / "modular_kernel_glue.py").exists():      # the reference's FusedMoEKernel over world-1 RCCL
        tags.append("reference FusedMoEKernel OK")
    for tag in tags:
        assert tag in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


# ------------------------------------------------------------------ two ranks on the one GPU of the test box
E2, K2, H2, I2, M2 = 8, 2, 256, 128, 29


def _two_rank_worker(rank, world, port, mode, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from lvllm_amd import ops
        from lvllm_amd.ep import ExpertParallelExperts
        g = torch.Generator().manual_seed(11)
        w13 = (torch.randn((E2, 2 * I2, H2), generator=g) / 8).to(torch.bfloat16)
        w2 = (torch.randn((E2, H2, I2), generator=g) / 8).to(torch.bfloat16)
        n_loc = E2 // world
        lo = rank * n_loc
        eng = ops.RoutedExpertsEngine(w13[lo:lo + n_loc].to(DEV), w2[lo:lo + n_loc].to(DEV), top_k=K2,
                                      act_dtype=torch.bfloat16, num_processes=world, process_id=rank)

        def staged_a2a(out, inp):                      # RCCL refuses two ranks on one device: gloo carries the bytes
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu())
            out.copy_(o)
        ep = ExpertParallelExperts(lambda rows, lids, ws, dt: eng.forward_rows(rows, ws, lids, out_dtype=dt), E2, H2,
                                   mode="a2a", transport=staged_a2a,
                                   return_dtype=torch.float32 if mode == "f32" else None)
        gx = torch.Generator().manual_seed(100 + rank)
        x = (torch.randn((M2, H2), generator=gx) / 2).to(torch.bfloat16)
        tw, ids = make_routing(M2, E2, K2, seed=200 + rank, drop=0.1)
        out = ep.forward(x.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV))
        torch.cuda.synchronize()
        q.put((rank, out.cpu().numpy(), ep.wire_bytes(M2, K2)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_two_ranks_sharing_the_gpu_match_the_single_rank_oracle(mode):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, out, wire = q.get(timeout=300)
        res[r] = (out, wire)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(11)
    w13 = (torch.randn((E2, 2 * I2, H2), generator=g) / 8).to(torch.bfloat16)
    w2 = (torch.randn((E2, H2, I2), generator=g) / 8).to(torch.bfloat16)
    d = orc.MoeDesc(E=E2, H=H2, I=I2, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    for r in range(world):
        gx = torch.Generator().manual_seed(100 + r)
        x = (torch.randn((M2, H2), generator=gx) / 2).to(torch.bfloat16)
        tw, ids = make_routing(M2, E2, K2, seed=200 + r, drop=0.1)
        ref = orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(x), ids, tw)
        out, wire = res[r]
        tol = 2e-3 if mode == "f32" else 2e-3 + K2 * 2.0 ** -8
        np.testing.assert_allclose(out, ref, atol=tol * np.abs(ref).max(), rtol=1e-2)
        assert wire["collectives_per_step"] == 2 and wire["capacity_tokens"] == M2
