#!/usr/bin/env python3
"""Generates tests/golden/eplb.npz by RUNNING the reference's own EPLB host code.

Same method as make_golden.py: the reference package cannot be imported here (missing zmq, msgspec,
...), so the pure numpy/torch pieces are taken out of the reference sources with `ast` and exec()ed as
they are; nothing of the reference is copied into the repo, only numeric input/output vectors.

Executed (paths relative to /root/reference):
  DefaultEplbPolicy (balanced_packing, replicate_experts, rebalance_experts_hierarchical,
    preserve_intragpu_slots, rebalance_experts)          vllm/distributed/eplb/policy/default.py:20-332
  compute_logical_maps                                    vllm/distributed/eplb/eplb_state.py:1159-1235
  EplbState.build_initial_global_physical_to_logical_map  vllm/distributed/eplb/eplb_state.py:297-314
  determine_expert_map with num_fused_shared_experts / return_expert_mask
                                                          vllm/model_executor/layers/fused_moe/expert_map_manager.py:22-113
Also stored: the known-answer vectors of the reference's own test of the policy
(tests/distributed/test_eplb_algo.py:12-72, the DeepSeek EPLB example), after checking that the
reference code run here reproduces them.

Loads are drawn WITHOUT exact ties (distinct values per row, and per-replica loads w/c distinct for the
replica counts that can occur): among exactly equal loads the reference's `np.argsort(-w)` order depends
on the numpy build / CPU (x86-simd-sort vs introsort), ours is the stable order -- both are valid EPLB
outputs, but only tie-free inputs have ONE answer to pin.

Run here (needs /root/reference):   python tests/golden/make_golden_eplb.py
"""
from __future__ import annotations

import ast
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def load_policy():
    path = REF / "vllm/distributed/eplb/policy/default.py"
    tree = ast.parse(path.read_text())
    tree.body = [n for n in tree.body if not (isinstance(n, ast.ImportFrom) and n.level > 0)]   # `.abstract`
    ns = {"AbstractEplbPolicy": object}
    exec(compile(tree, str(path), "exec"), ns)
    return ns["DefaultEplbPolicy"]


def load_state_functions():
    path = REF / "vllm/distributed/eplb/eplb_state.py"
    tree = ast.parse(path.read_text())
    ns = {"torch": torch, "Sequence": __import__("typing").Sequence}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "compute_logical_maps":
            exec(compile(ast.Module(body=[node], type_ignores=[]), str(path), "exec"), ns)
        if isinstance(node, ast.ClassDef) and node.name == "EplbState":
            for m in node.body:
                if isinstance(m, ast.FunctionDef) and m.name == "build_initial_global_physical_to_logical_map":
                    m.decorator_list = []
                    exec(compile(ast.Module(body=[m], type_ignores=[]), str(path), "exec"), ns)
    return ns["compute_logical_maps"], ns["build_initial_global_physical_to_logical_map"]


def tie_free_loads(rng, L, E, max_rep):
    """distinct integer-ish loads whose quotients w/c (c <= max_rep) are all distinct too"""
    for _ in range(1000):
        w = rng.permutation(np.arange(1, 50 * E + 1))[: L * E].reshape(L, E).astype(np.float32)
        w = w * np.float32(7.0) + rng.random((L, E)).astype(np.float32)           # break rational ties
        ok = True
        for l in range(L):
            q = np.concatenate([w[l].astype(np.float64) / c for c in range(1, max_rep + 1)])
            if np.unique(q).size != q.size:
                ok = False
                break
        if ok:
            return w
    raise RuntimeError("could not draw tie-free loads")


# (layers, logical, physical, groups, nodes, ranks)
GEOMETRIES = [
    (2, 12, 16, 4, 2, 8),        # the DeepSeek example's geometry
    (1, 8, 16, 1, 1, 8),         # Mixtral-8x7B, one redundant slot per rank
    (3, 128, 144, 1, 1, 8),      # Qwen3-30B-A3B / GLM-4.5-Air: no groups, 8 ranks
    (2, 256, 288, 8, 1, 8),      # DeepSeek-V3: 8 groups on one 8-GPU node, 32+4 per rank
    (2, 256, 320, 8, 2, 16),     # two nodes
    (2, 16, 24, 4, 2, 4),
    (2, 12, 16, 3, 2, 8),        # groups do not divide over nodes -> global policy
    (4, 64, 64, 8, 1, 8),        # no redundancy: pure re-packing
    (1, 6, 10, 3, 1, 2),
]


def main():
    Policy = load_policy()
    compute_logical_maps, build_initial = load_state_functions()
    rng = np.random.default_rng(20260926)
    out: dict[str, np.ndarray] = {}

    # ---- the reference test's known answers (test_eplb_algo.py:12-72), reproduced here first
    w = torch.tensor([[90, 132, 40, 61, 104, 165, 39, 4, 73, 56, 183, 86],
                      [20, 107, 104, 64, 19, 197, 187, 157, 172, 86, 16, 27]])
    exp_p2l = torch.tensor([[5, 6, 5, 7, 8, 4, 3, 4, 10, 9, 10, 2, 0, 1, 11, 1],
                            [7, 10, 6, 8, 6, 11, 8, 9, 2, 4, 5, 1, 5, 0, 3, 1]])
    exp_cnt = torch.tensor([[1, 2, 1, 1, 2, 2, 1, 1, 1, 1, 2, 1], [1, 2, 1, 1, 1, 2, 2, 1, 2, 1, 1, 1]])
    p2l = Policy.rebalance_experts(w, 16, 4, 2, 8)
    l2p, cnt = compute_logical_maps(p2l, 12)
    assert torch.equal(p2l, exp_p2l) and torch.equal(cnt, exp_cnt), "reference code does not reproduce its own test"
    out["ka_weight"], out["ka_phy2log"], out["ka_logcnt"], out["ka_log2phy"] = (
        w.numpy(), exp_p2l.numpy(), exp_cnt.numpy(), l2p.numpy())

    # ---- full policy on tie-free loads, then a second round with the previous placement (slot preservation)
    for gi, (L, E, P, G, N, R) in enumerate(GEOMETRIES):
        max_rep = P - E + 1
        w0 = tie_free_loads(rng, L, E, max_rep)
        w1 = tie_free_loads(rng, L, E, max_rep)
        p0 = Policy.rebalance_experts(torch.from_numpy(w0), P, G, N, R)
        p1 = Policy.rebalance_experts(torch.from_numpy(w1), P, G, N, R, p0)
        p1_free = Policy.rebalance_experts(torch.from_numpy(w1), P, G, N, R)
        l2p0, cnt0 = compute_logical_maps(p0, E)
        out[f"g{gi}_geom"] = np.array([L, E, P, G, N, R], dtype=np.int64)
        out[f"g{gi}_w0"], out[f"g{gi}_w1"] = w0, w1
        out[f"g{gi}_p0"], out[f"g{gi}_p1"], out[f"g{gi}_p1_free"] = p0.numpy(), p1.numpy(), p1_free.numpy()
        out[f"g{gi}_l2p0"], out[f"g{gi}_cnt0"] = l2p0.numpy(), cnt0.numpy()
    out["n_geometries"] = np.array(len(GEOMETRIES))

    # ---- the building blocks on their own
    for ci, (X, n, packs) in enumerate([(3, 16, 4), (2, 36, 9), (5, 8, 8), (1, 64, 2)]):
        w = tie_free_loads(rng, X, n, 1)
        pi, ri = Policy.balanced_packing(w, packs)
        out[f"bp{ci}_w"], out[f"bp{ci}_packs"], out[f"bp{ci}_pack"], out[f"bp{ci}_rank"] = w, np.array(packs), pi, ri
    for ci, (X, n, phy) in enumerate([(3, 16, 24), (2, 32, 32), (4, 8, 20)]):
        w = tie_free_loads(rng, X, n, phy - n + 1)
        p2l, cnt = Policy.replicate_experts(w, phy)
        out[f"re{ci}_w"], out[f"re{ci}_phy"], out[f"re{ci}_p2l"], out[f"re{ci}_cnt"] = w, np.array(phy), p2l, cnt
    out["n_bp"], out["n_re"] = np.array(4), np.array(3)

    # ---- logical maps with unused slots (test_eplb_algo.py:303-340) and with replicas
    p2l = torch.tensor([[0, 1, -1, 2, 3, -1], [3, -1, 2, 1, 0, -1]])
    l2p, cnt = compute_logical_maps(p2l, 4)
    out["lm0_p2l"], out["lm0_l2p"], out["lm0_cnt"] = p2l.numpy(), l2p.numpy(), cnt.numpy()
    p2l = torch.tensor([[2, 0, 2, 1, -1, 2, 0, 3], [1, 1, 1, 0, 2, 3, -1, -1]])
    l2p, cnt = compute_logical_maps(p2l, 4)
    out["lm1_p2l"], out["lm1_l2p"], out["lm1_cnt"] = p2l.numpy(), l2p.numpy(), cnt.numpy()
    p2l1 = torch.tensor([3, 0, 1, 2, 0])
    l2p, cnt = compute_logical_maps(p2l1, 4)
    out["lm2_p2l"], out["lm2_l2p"], out["lm2_cnt"] = p2l1.numpy(), l2p.numpy(), cnt.numpy()

    # ---- initial placement
    for ci, (e, r) in enumerate([(8, 8), (128, 16), (256, 32), (6, 0), (4, 9)]):
        out[f"init{ci}"] = np.array([e, r] + list(build_initial(e, r)), dtype=np.int64)
    out["n_init"] = np.array(5)

    # ---- expert map with fused shared experts (map tail + mask; lvllm_amd/shared_experts.py, ops.determine_expert_map)
    path = REF / "vllm/model_executor/layers/fused_moe/expert_map_manager.py"
    ns = {"torch": torch, "ExpertPlacementStrategy": str}
    for node in ast.parse(path.read_text()).body:
        if isinstance(node, ast.FunctionDef) and node.name == "determine_expert_map":
            exec(compile(ast.Module(body=[node], type_ignores=[]), str(path), "exec"), ns)
    ci = 0
    for E, ep, n_sh in [(8, 2, 1), (8, 4, 2), (10, 3, 1), (128, 8, 1), (256, 8, 2)]:
        for strat in ("linear", "round_robin"):
            for r in range(ep):
                n_loc, emap, mask = ns["determine_expert_map"](ep, r, E, strat, n_sh, True)
                out[f"sm{ci}_meta"] = np.array([E, ep, r, n_sh, 0 if strat == "linear" else 1, n_loc], np.int64)
                out[f"sm{ci}_map"], out[f"sm{ci}_mask"] = emap.numpy(), mask.numpy()
                ci += 1
    out["n_sm"] = np.array(ci)

    np.savez_compressed(OUT / "eplb.npz", **out)
    print(f"wrote {OUT / 'eplb.npz'} ({(OUT / 'eplb.npz').stat().st_size} bytes, {len(out)} arrays)")


if __name__ == "__main__":
    main()
