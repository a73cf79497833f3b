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
