"""GPU: the engine as ONE rank of an expert-parallel group sees it at the Mixtral shapes bench.py runs with --gpus 2/4/8:
E_local = 4 / 2 / 1 experts of 8, ep x 32 token records of which ~1/ep carry local ids (valid_den = ep), global ids made
local by id_offset.  Checked against the full 8-expert engine run on the same tokens with every non-local slot masked
(-1): same experts, same rows, so the same bits per (token, expert) partial; the sums differ only by fp32 order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("ep", [2, 4, 8])
def test_rank_local_engine_matches_masked_full_engine(ep):
    import bench
    from lvllm_amd import ops
    wl = dict(bench.WORKLOADS["mixtral8x7b_bf16_decode_m32"])
    E, K, H = wl["E"], wl["K"], wl["H"]
    E_local, first = E // ep, (ep - 1) * (E // ep)            # the last rank's window
    full = bench.build_engine(ops, wl, E, 0, torch.device(DEV))[0]
    loc = bench.build_engine(ops, wl, E_local, first, torch.device(DEV), num_processes=ep, process_id=ep - 1)[0]
    loc.engine.set_tuning(valid_den=ep)
    R = ep * 32                                               # records a rank receives: ep x capacity
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.randn((R, H), generator=g, device=DEV) / 10).to(torch.bfloat16)
    tw, ids = ops.topk_softmax(torch.randn((R, E), generator=g, device=DEV), K, True)
    got = loc.forward_rows(x, tw, ids, id_offset=first)
    masked = torch.where((ids >= first) & (ids < first + E_local), ids, torch.full_like(ids, -1))
    want = full.forward_rows(x, tw, masked)
    assert int((masked >= 0).sum()) > 0
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 1e-5, (err, loc.engine.describe())
    dead = (masked < 0).all(dim=1)
    assert bool((got[dead] == 0).all())


@pytest.mark.parametrize("ep,E,K,M,ragged", [(8, 16, 4, 24, False), (8, 8, 2, 32, False), (4, 16, 6, 17, True), (8, 32, 8, 9, True)])
def test_eight_ranks_emulated_on_one_gpu(ep, E, K, M, ragged):
    """The whole fixed-capacity step as `ep` ranks would run it -- HIP pack kernel, the all-to-all done by hand
    (recv[r][p] = send[p][r]), one REAL engine per rank over its expert window on the received records in place, the
    return all-to-all, the HIP combine kernel -- against the full engine on every rank's tokens.  `ragged`: ranks hold
    different token counts under the common capacity."""
    from lvllm_amd import ops
    H, I = 256, 128
    g = torch.Generator().manual_seed(ep * 100 + E)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 8).to(torch.bfloat16).to(DEV)
    w2 = (torch.randn((E, H, I), generator=g) / 8).to(torch.bfloat16).to(DEV)
    full = ops.RoutedExpertsEngine(w13, w2, top_k=K, act_dtype=torch.bfloat16)
    El = E // ep
    engines = [ops.RoutedExpertsEngine(w13[r * El:(r + 1) * El].contiguous(), w2[r * El:(r + 1) * El].contiguous(),
                                       top_k=K, act_dtype=torch.bfloat16) for r in range(ep)]
    cap = M
    Ms = [M - (r % 3) * 2 if ragged else M for r in range(ep)]
    rowb = ops.ep_row_bytes(H, K)
    xs, tws, idss, sends, slots = [], [], [], [], []
    for r in range(ep):
        x = (torch.randn((Ms[r], H), generator=g) / 4).to(torch.bfloat16).to(DEV)
        tw, ids = ops.topk_softmax(torch.randn((Ms[r], E), generator=g).to(DEV), K, True)
        send = torch.empty((ep, cap, rowb), dtype=torch.uint8, device=DEV)
        slot_of = torch.empty((ep, Ms[r]), dtype=torch.int32, device=DEV)
        ovf = torch.zeros((1,), dtype=torch.int32, device=DEV)
        ops.ep_pack_tokens(x, tw, ids, E, ep, cap, send, slot_of, ovf, True)
        assert int(ovf.item()) == 0
        xs.append(x); tws.append(tw); idss.append(ids); sends.append(send); slots.append(slot_of)
    backs = [torch.empty((ep, cap, H), dtype=torch.float32, device=DEV) for _ in range(ep)]
    for r in range(ep):                                   # rank r: receive, compute on the records in place, return
        recv = torch.stack([sends[p][r] for p in range(ep)]).contiguous()          # the dispatch all-to-all
        rec = recv.view(ep * cap, rowb)
        rows = rec[:, :H * 2].view(torch.bfloat16)
        rids = rec[:, H * 2:H * 2 + 4 * K].view(torch.int32)
        rws = rec[:, H * 2 + 4 * K:H * 2 + 8 * K].view(torch.float32)
        engines[r].engine.set_tuning(valid_den=ep)
        y = engines[r].forward_rows(rows, rws, rids, out_dtype=torch.float32, id_offset=r * El).view(ep, cap, H)
        for p in range(ep):                               # the return all-to-all
            backs[p][r].copy_(y[p])
    for r in range(ep):
        out = ops.ep_combine(backs[r], slots[r], torch.empty((Ms[r], H), dtype=torch.float32, device=DEV))
        want = full.decode(xs[r], tws[r], idss[r])
        err = float((out - want).abs().max() / want.abs().max())
        # a rank's engine may split K differently from the full engine: fp32 order -> an occasional bf16 rounding flip of
        # the intermediate (2^-9 of one element); a routing / slot / window mistake would be O(1)
        assert err < 1e-3, (r, err)
