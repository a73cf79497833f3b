#!/usr/bin/env python3
"""One rank's share of an expert-parallel step, timed on ONE GPU (development; feeds the prediction in DESIGN.md section 5).

For ep in 1, 2, 4, 8 the script builds the engine of rank 0 (E / ep experts), hands it what the fixed-capacity exchange
would deliver -- ep x capacity token records with GLOBAL ids, ~1/ep of them local (`valid_den` = ep) -- and times the
captured engine step (sort, GEMM1, GEMM2, combine), plus the pack and combine kernels of the exchange on their own.
  python tools/ep_rank_time.py [mixtral|dsv3]
"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from bench import WORKLOADS, build_engine  # noqa: E402
from lvllm_amd import ops  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "mixtral"
    name = "mixtral8x7b_bf16_decode_m32" if which == "mixtral" else "dsv3_fp8w8a8_ep_decode_b256"
    wl = dict(WORKLOADS[name])
    E, K, H = wl["E"], wl["K"], wl["H"]
    dev = torch.device("cuda", 0)
    out = []
    for ep in (1, 2, 4, 8):
        # Mixtral: weak scaling, 32 tokens per rank; DSv3: the global batch of 256 split over the ranks
        m_rank = wl["M"] if "M" in wl else wl["M_global"] // ep
        cap = m_rank
        R = ep * cap
        E_loc = E // ep
        eng = build_engine(ops, wl, E_loc, 0, dev, max_num_seqs=max(256, R), max_batch_size=max(8192, R),
                           num_processes=ep, process_id=0)[0]
        g = torch.Generator(device=dev).manual_seed(7)
        x = (torch.randn((R, H), generator=g, device=dev) / 10).to(torch.bfloat16)
        logits = torch.randn((R, E), generator=g, device=dev)
        tw, ids = ops.topk_softmax(logits, K, True)
        y = torch.empty((R, H), dtype=torch.bfloat16, device=dev)
        t_eng = timed(lambda: eng.forward_rows(x, tw, ids, out=y, id_offset=0, valid_den=ep if ep > 1 else None))
        rec = {"workload": name, "ep": ep, "tokens_per_rank": m_rank, "records": R, "experts_per_rank": E_loc,
               "engine_us": round(t_eng, 1), "plan": eng.engine.describe().split("|", 2)[2].strip()}
        if ep > 1:
            rowb = ops.ep_row_bytes(H, K)
            send = torch.empty((ep, cap, rowb), dtype=torch.uint8, device=dev)
            slot_of = torch.empty((ep, m_rank), dtype=torch.int32, device=dev)
            ov = torch.zeros((1,), dtype=torch.int32, device=dev)
            xm, twm, idm = x[:m_rank].contiguous(), tw[:m_rank].contiguous(), ids[:m_rank].contiguous()
            rec["pack_us"] = round(timed(lambda: ops.ep_pack_tokens(xm, twm, idm, E, ep, cap, send, slot_of, ov, True)), 1)
            back = torch.randn((ep, cap, H), device=dev).to(torch.bfloat16)
            o = torch.empty((m_rank, H), dtype=torch.float32, device=dev)
            rec["combine_us"] = round(timed(lambda: ops.ep_combine(back, slot_of, o)), 1)
            rec["dispatch_bytes_per_peer"] = cap * rowb
            rec["return_bytes_per_peer"] = cap * H * 2
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
