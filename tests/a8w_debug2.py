#!/usr/bin/env python3
"""Development script (GPU box): one expert, many token tiles -> workgroups of the persistent fp8 prefill kernel with two
items; prints which (tile, 16-row block, column group) of the output differ from the 128-row tile kernel."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from lvllm_amd import _clib, ops  # noqa: E402
from oracle import oracle as orc  # noqa: E402

DEV = "cuda:0"


def main(M=4200, H=1024, I=4096):
    E, K = 1, 1
    g = torch.Generator().manual_seed(1)
    a = (torch.randn((M, H), generator=g) / 10).to(torch.bfloat16)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 10).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g) / 10).to(torch.bfloat16)
    tw = np.ones((M, 1), dtype=np.float32)
    ids = np.zeros((M, 1), dtype=np.int32)
    q13, s13 = orc.quant_fp8_block(w13.float().numpy(), 128, 128)
    q2, s2 = orc.quant_fp8_block(w2.float().numpy(), 128, 128)
    eng = ops.RoutedExpertsEngine(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=torch.bfloat16,
                                  fmt="fp8", w13_scale=torch.from_numpy(s13), w2_scale=torch.from_numpy(s2),
                                  group_n=128, group_k=128, fp8_mode=_clib.FP8_W8A8, has_gate_proj=True,
                                  max_batch_size=16384)
    xa, xtw, xids = a.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV)

    def run(**kw):
        eng.engine.set_tuning(**kw)
        return eng.decode(xa, xtw, xids).cpu().numpy(), eng.engine.describe()

    base, d0 = run(tiled=128, pf=0, dbg=0, xcd=-1)
    print("base:", d0)
    scale = float(np.abs(base).max())
    for name, kw in [("pf9", dict(tiled=256, pf=0, dbg=0, xcd=-1)), ("pf9 serial", dict(tiled=256, pf=0, dbg=4, xcd=-1)),
                     ("pf9 reread", dict(tiled=256, pf=0, dbg=256, xcd=-1))]:
        y, d = run(**kw)
        bad = np.abs(y - base) > 4e-3 * scale + 1e-2 * np.abs(base)
        print(f"[{name}] bad={bad.mean():.3e}  max err/scale {np.abs(y - base).max() / scale:.3e}")
        rows = np.nonzero(bad.any(axis=1))[0]
        if len(rows):
            nt = (M + 255) // 256
            npair = (M + 31) // 32
            base_p, rem = npair // nt, npair % nt
            starts = [32 * (i * base_p + min(i, rem)) for i in range(nt + 1)]
            tile_of = np.searchsorted(starts, rows, side="right") - 1
            print("   tiles:", nt, "rows/tile ~", starts[1], " bad rows per tile:", dict(zip(*np.unique(tile_of, return_counts=True))))
            # stale-operand check: a second item's block 0 against the same workgroup's first item (tile - 8)
            for t in np.unique(tile_of)[:3]:
                r0, rp = starts[t], starts[t - 8]
                d_self = np.abs(y[r0:r0 + 16] - base[r0:r0 + 16]).max() / scale
                d_prev = np.abs(y[r0:r0 + 16] - base[rp:rp + 16]).max() / scale
                d_sum = np.abs(y[r0:r0 + 16] - base[r0:r0 + 16] - base[rp:rp + 16]).max() / scale
                print(f"   tile {t} block 0: |y - base(self)| {d_self:.3e}  |y - base(tile {t-8})| {d_prev:.3e}  |y - self - prev| {d_sum:.3e}")
            for t in np.unique(tile_of)[:2]:
                rr = rows[tile_of == t] - starts[t]
                print(f"   tile {t}: bad 16-row blocks {sorted(set((rr // 16).tolist()))}; bad col groups of 128 (count of bad elems):",
                      [int(bad[rows[tile_of == t]][:, c:c + 128].sum()) for c in range(0, H, 128)])
    eng.engine.set_tuning(tiled=0, pf=0, dbg=0, xcd=0)


if __name__ == "__main__":
    main()
