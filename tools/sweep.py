#!/usr/bin/env python3
"""Kernel-geometry sweep on the GPU box: per-kernel HIP-event timings for every launch geometry of
the decode path (lkm_set_tuning), printed as a table + JSON lines.  Development tool."""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from bench import WORKLOADS, build_engine  # noqa: E402
from lvllm_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mixtral8x7b_bf16_decode_m32")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--g1", default="1:1:1,1:2:1,2:1:1,2:2:1,1:2:2,1:4:1,4:1:1", help="nt1:tb:kw list")
    ap.add_argument("--g2", default="1:1,1:2,1:4,2:1,2:2,2:4,2:8,4:2,4:4", help="nt2:sk list")
    ap.add_argument("--M", type=int, default=0)
    ap.add_argument("--routing", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--cfgs", default="", help="';'-separated tuning sets 'k=v,k=v' measured as whole steps (replaces --g1/--g2)")
    args = ap.parse_args()
    wl = dict(WORKLOADS[args.workload])
    if args.M:
        wl["M"] = args.M
    wl.setdefault("M", wl.get("M_global", 0))
    E, K, H, I, M, fmt = wl["E"], wl["K"], wl["H"], wl["I"], wl["M"], wl["fmt"]
    dev = torch.device("cuda", 0)
    eng, bpe = build_engine(ops, wl, E, 0, dev, max_num_seqs=max(256, M), max_batch_size=max(8192, M))[:2]
    gen = torch.Generator(device=dev).manual_seed(7)
    x = (torch.randn((M, H), generator=gen, device=dev) / 10).to(torch.bfloat16)
    logits = torch.randn((M, E), generator=gen, device=dev)
    if args.routing == "zipf":      # log-popularity bias p(e) ~ 1/(e+1), as bench.py
        logits = logits + torch.log(1.0 / torch.arange(1, E + 1, device=dev, dtype=torch.float32))[None, :]
    tw, ids = ops.topk_softmax(logits, K, True)
    e_act = int(torch.unique(ids).numel())
    g1_bytes, g2_bytes = e_act * 2 * I * H * bpe, e_act * H * I * bpe
    out = torch.empty((M, H), dtype=torch.float32, device=dev)
    eng.engine.set_profiling(True)

    def measure():
        acc = {"sort": 0.0, "gemm1": 0.0, "gemm2": 0.0, "combine": 0.0}
        for _ in range(3):
            eng.decode(x, tw, ids, out=out)
        for _ in range(args.reps):
            eng.decode(x, tw, ids, out=out)
            p = eng.engine.get_profile()
            for k in acc:
                acc[k] += p[k] / args.reps
        return acc

    print(f"# rows per expert: {torch.bincount(ids.flatten().long(), minlength=E).tolist()}")
    print(f"# {args.workload} {args.routing} M={M} e_act={e_act} g1={g1_bytes/1e9:.3f} GB g2={g2_bytes/1e9:.3f} GB")
    if args.cfgs:
        keys = ("nt1", "nt2", "kw1", "sk2", "tbmax", "tiled", "waves", "hybrid", "pd1", "pd2", "xcd", "pf", "direct", "valid_den", "dbg", "ydt")
        for spec in args.cfgs.split(";"):
            kv = {k: 0 for k in keys}
            if spec.strip():
                kv.update({k: int(v) for k, v in (x.split("=") for x in spec.split(","))})
            try:
                eng.engine.set_tuning(**kv)
                p = measure()
            except Exception as e:
                print(f"[{spec}] FAILED {e}")
                continue
            tot = sum(p.values())
            print(f"[{spec or 'auto'}] sort {p['sort']*1e3:.1f} gemm1 {p['gemm1']*1e3:.1f} ({g1_bytes/(p['gemm1']*1e-3)/1e9:.0f} GB/s) "
                  f"gemm2 {p['gemm2']*1e3:.1f} ({g2_bytes/(p['gemm2']*1e-3)/1e9:.0f} GB/s) combine {p['combine']*1e3:.1f} "
                  f"total {tot*1e3:.1f} us {M/(tot*1e-3):.0f} tok/s | {eng.engine.describe().split('|',2)[2]}")
            print(json.dumps({"cfg": spec, **{k: v * 1e3 for k, v in p.items()}}))
        return
    base = None
    for spec in args.g1.split(","):
        nt1, tb, kw = map(int, spec.split(":"))
        try:
            eng.engine.set_tuning(nt1=nt1, tbmax=tb, kw1=kw, nt2=0, sk2=0)
            p = measure()
        except Exception as e:
            print(f"g1 nt={nt1} tb={tb} kw={kw}: FAILED {e}")
            continue
        gbs = g1_bytes / (p["gemm1"] * 1e-3) / 1e9
        print(f"g1 nt={nt1} tb={tb} kw={kw}: gemm1 {p['gemm1']*1e3:8.1f} us  {gbs:7.1f} GB/s  ({gbs/80:.1f}% of 8 TB/s)  sort {p['sort']*1e3:.1f} us")
        print(json.dumps({"k": "gemm1", "nt": nt1, "tb": tb, "kw": kw, "us": p["gemm1"] * 1e3, "GBs": gbs}))
    tb2 = 2 if M <= 32 else 4
    for spec in args.g2.split(","):
        nt2, sk = map(int, spec.split(":"))
        for tb in sorted({1 if M <= 16 else tb2, tb2}):
            if nt2 * tb > 8:
                continue
            try:
                eng.engine.set_tuning(nt1=0, tbmax=tb, kw1=0, nt2=nt2, sk2=sk)
                p = measure()
            except Exception as e:
                print(f"g2 nt={nt2} tb={tb} sk={sk}: FAILED {e}")
                continue
            gbs = g2_bytes / (p["gemm2"] * 1e-3) / 1e9
            print(f"g2 nt={nt2} tb={tb} sk={sk}: gemm2 {p['gemm2']*1e3:8.1f} us  {gbs:7.1f} GB/s  ({gbs/80:.1f}%)  combine {p['combine']*1e3:.1f} us")
            print(json.dumps({"k": "gemm2", "nt": nt2, "tb": tb, "sk": sk, "us": p["gemm2"] * 1e3, "GBs": gbs, "combine_us": p["combine"] * 1e3}))
    eng.engine.set_tuning(nt1=0, tbmax=0, kw1=0, nt2=0, sk2=0)
    p = measure()
    tot = sum(p.values())
    print(f"auto: {eng.engine.describe()}")
    print(f"auto: sort {p['sort']*1e3:.1f} gemm1 {p['gemm1']*1e3:.1f} gemm2 {p['gemm2']*1e3:.1f} combine {p['combine']*1e3:.1f} total {tot*1e3:.1f} us "
          f"-> {(g1_bytes+g2_bytes)/(tot*1e-3)/1e9:.0f} GB/s layer, {M/(tot*1e-3):.0f} tok/s")


if __name__ == "__main__":
    main()
