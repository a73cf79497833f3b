#!/usr/bin/env python3
"""Development script (GPU box; lives under tests/ because it quantises with the oracle): the round-3 fp8 x fp8 prefill kernel (pf=9) against the round-2 one (pf=8) and the
128-row tile kernel on one mid-size case, with the error broken down by where a row / column sits in its tile --
an indexing bug shows as a pattern, a synchronisation bug as noise that the serialised mode (dbg=4) removes."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from lvllm_amd import _clib, ops  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import make_routing  # noqa: E402

DEV = "cuda:0"


def main(M=3000, E=6, K=1, H=1024, I=1024, gated=True):
    print(f"#### M={M} E={E} K={K} H={H} I={I} gated={gated}")
    g = torch.Generator().manual_seed(1)
    a = (torch.randn((M, H), generator=g) / 10).to(torch.bfloat16)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 10).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g) / 10).to(torch.bfloat16)
    tw, ids = make_routing(M, E, K, 3, skew=0.3)
    q13, s13 = orc.quant_fp8_block(w13.float().numpy(), 128, 128)
    q2, s2 = orc.quant_fp8_block(w2.float().numpy(), 128, 128)
    eng = ops.RoutedExpertsEngine(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=torch.bfloat16,
                                  fmt="fp8", w13_scale=torch.from_numpy(s13), w2_scale=torch.from_numpy(s2),
                                  group_n=128, group_k=128, fp8_mode=_clib.FP8_W8A8, has_gate_proj=gated,
                                  max_batch_size=4096)
    xa, xtw, xids = a.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV)

    def run(**kw):
        eng.engine.set_tuning(**kw)
        y = eng.decode(xa, xtw, xids).cpu().numpy()
        d = eng.engine.describe()
        return y, d

    base, d0 = run(tiled=128, pf=0, dbg=0, xcd=-1)
    print("base:", d0)
    scale = float(np.abs(base).max())
    # position of each token inside its expert's sorted rows (stable by token index)
    e_of = ids[:, 0]
    pos = np.zeros(M, dtype=np.int64)
    cnt = {}
    for t in range(M):
        e = int(e_of[t])
        pos[t] = cnt.get(e, 0)
        cnt[e] = pos[t] + 1
    print("rows per expert:", {k: v for k, v in sorted(cnt.items())})
    for name, kw in [("pf8", dict(tiled=256, pf=8, dbg=0, xcd=-1)), ("pf9", dict(tiled=256, pf=0, dbg=0, xcd=-1)),
                     ("pf9 serial", dict(tiled=256, pf=0, dbg=4, xcd=-1)), ("pf9 xcd", dict(tiled=256, pf=0, dbg=0, xcd=1)),
                     ("pf9 again", dict(tiled=256, pf=0, dbg=0, xcd=-1))]:
        y, d = run(**kw)
        err = np.abs(y - base)
        bad = err > 4e-3 * scale + 1e-2 * np.abs(base)
        print(f"[{name}] {d.split('|')[-2] if '|' in d else d}")
        print(f"   finite={np.isfinite(y).all()} max|err|/scale={np.nanmax(err) / scale:.3e} bad={bad.mean():.3e}")
        if bad.any():
            rows = bad.any(axis=1)
            cols = bad.any(axis=0)
            print("   bad rows by expert:", {int(e): int(rows[e_of == e].sum()) for e in np.unique(e_of)})
            blk = (pos[rows] // 16)
            print("   bad rows by 16-row block index inside the expert (first 40):", np.bincount(blk)[:40].tolist())
            first_bad = np.nonzero(rows)[0][:12]
            print("   first bad tokens (token, expert, pos in expert):", [(int(t), int(e_of[t]), int(pos[t])) for t in first_bad])
            print("   bad cols by col // 256:", np.bincount(np.nonzero(cols)[0] // 256).tolist())
            print("   bad cols by (col // 16) % 16:", np.bincount((np.nonzero(cols)[0] // 16) % 16, minlength=16).tolist())
            print("   bad cols by col % 16:", np.bincount(np.nonzero(cols)[0] % 16, minlength=16).tolist())
    eng.engine.set_tuning(tiled=0, pf=0, dbg=0, xcd=0)


if __name__ == "__main__":
    main()
    main(M=12000, E=24, H=1152, I=1280)
    main(M=12000, E=24, H=1152, I=1280, gated=False)
