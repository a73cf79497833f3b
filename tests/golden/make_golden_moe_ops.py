#!/usr/bin/env python3
"""Generates tests/golden/moe_ops.npz by RUNNING the reference's own golden implementations of the scatter / gather
operators (as tests/golden/make_golden.py does for the other rows: the functions are pulled out of the reference's test
files with `ast` and executed unchanged, except that the literal device "cuda" becomes "cpu"; only vectors are saved):
  torch_moe_align_block_size     tests/kernels/moe/test_moe_align_block_size.py:96-172
  torch_permute, torch_unpermute tests/kernels/moe/test_moe_permute_unpermute.py:37-123
  determine_expert_map           vllm/model_executor/layers/fused_moe/expert_map_manager.py:22-113
Inputs follow the reference tests' recipes (randperm ids per token for align, :185-189 / :249-262; fused_topk of randn
gating for permute, :141-146 -- the top-k itself is taken with torch.topk here, the ids only need to be distinct per token).
Run here (needs /root/reference):   python tests/golden/make_golden_moe_ops.py"""
import ast
import sys
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT))
from make_golden import extract  # noqa: E402


class _Cpu(ast.NodeTransformer):
    def visit_Constant(self, node):
        return ast.copy_location(ast.Constant("cpu"), node) if node.value == "cuda" else node


def extract_cpu(path, names, ns):
    tree = _Cpu().visit(ast.parse((REF / path).read_text()))
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.fix_missing_locations(ast.Module(body=[node], type_ignores=[])), str(REF / path), "exec"), ns)
    assert all(n in ns for n in names), names


def main():
    ns = {"torch": torch, "round_up": lambda a, b: -(-a // b) * b, "cdiv": lambda a, b: -(-a // b)}
    extract_cpu("tests/kernels/moe/test_moe_align_block_size.py", ["torch_moe_align_block_size"], ns)
    extract_cpu("tests/kernels/moe/test_moe_permute_unpermute.py", ["torch_permute", "torch_unpermute"], ns)
    emns = {"torch": torch, "get_compressed_expert_map": None}
    import logging
    emns["logger"] = logging.getLogger("golden")
    extract("vllm/model_executor/layers/fused_moe/expert_map_manager.py", ["determine_expert_map"], emns)
    out = {}
    g = torch.Generator().manual_seed(0)

    # ---- moe_align_block_size, no expert map (test_moe_align_block_size:175-240)
    n = 0
    for m, topk, E, bs, pad in [(1, 1, 32, 32, False), (3, 2, 32, 128, True), (256, 16, 160, 32, False), (256, 2, 257, 128, True),
                                (2256, 2, 256, 128, False), (2256, 32, 160, 32, True), (4096, 1, 257, 32, False), (33, 16, 32, 128, False),
                                (3, 32, 256, 32, True)]:
        ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(m)]).to(torch.int32)
        s, e, p = ns["torch_moe_align_block_size"](ids, bs, E, None, pad)
        out.update({f"align{n}_ids": ids.numpy(), f"align{n}_args": np.array([bs, E, int(pad)]), f"align{n}_sorted": s.numpy(),
                    f"align{n}_experts": e.numpy(), f"align{n}_post": p.numpy()})
        n += 1
    out["n_align"] = np.array(n)
    # ---- ... with an expert map (:243-300): every second expert local, inactive ids optionally masked to -1
    n = 0
    for m, topk, E, mask in [(16, 2, 8, False), (32, 4, 64, True), (2048, 2, 8, True), (2048, 4, 64, False)]:
        emap = torch.full((E,), -1, dtype=torch.int32)
        local = list(range(0, E, 2))
        emap[local] = torch.arange(len(local), dtype=torch.int32)
        ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(m)]).to(torch.int32)
        if mask:
            ids = torch.where(emap[ids.long()] >= 0, ids, torch.full_like(ids, -1))
        ids[0, 0] = -1
        # (the golden implementation compares ids with expert indices: -1 matches none of them)
        s, e, p = ns["torch_moe_align_block_size"](ids, 64, E, emap)
        out.update({f"alignm{n}_ids": ids.numpy(), f"alignm{n}_map": emap.numpy(), f"alignm{n}_sorted": s.numpy(),
                    f"alignm{n}_experts": e.numpy(), f"alignm{n}_post": p.numpy()})
        n += 1
    out["n_alignm"] = np.array(n)

    # ---- moe_permute / moe_unpermute (test_moe_permute_unpermute.py:126-216)
    n = 0
    for n_token, H, E, topk, ep in [(1, 64, 16, 2, 1), (33, 64, 64, 6, 4), (1024, 32, 256, 8, 16), (33, 128, 16, 2, 4),
                                    (500, 64, 64, 8, 1), (257, 64, 256, 6, 4)]:
        rank = int(torch.randint(0, ep, (1,), generator=g))
        emap, n_local = None, E
        if ep != 1:
            n_local, emap, _ = emns["determine_expert_map"](ep, rank, E)
        start = n_local * rank
        hidden = torch.randn((n_token, H), generator=g).to(torch.bfloat16)
        gating = torch.randn((n_token, E), generator=g)
        tw, ids = torch.topk(torch.softmax(gating, dim=-1), topk, dim=-1)
        ids = ids.to(torch.int32)
        ph, first, inv, perm, valid = ns["torch_permute"](hidden, ids.long(), topk, E, n_local, start, expert_map=emap)
        res0 = (0.5 * ph.float() + torch.randn(ph.shape, generator=g)).to(torch.bfloat16)
        gold4 = ns["torch_unpermute"](res0.clone(), tw, ids, None, inv, valid, topk, n_local)
        out.update({f"perm{n}_hidden": hidden.view(torch.int16).numpy(), f"perm{n}_ids": ids.numpy(), f"perm{n}_tw": tw.float().numpy(),
                    f"perm{n}_args": np.array([E, n_local, ep, rank]),
                    f"perm{n}_map": (emap if emap is not None else torch.zeros(0, dtype=torch.int32)).numpy(),
                    f"perm{n}_rows": ph.view(torch.int16).numpy(), f"perm{n}_first": first.numpy(), f"perm{n}_inv": inv.numpy(),
                    f"perm{n}_perm": perm.numpy(), f"perm{n}_nvalid": np.array(len(valid)),
                    f"perm{n}_res0": res0.view(torch.int16).numpy(), f"perm{n}_gold4": gold4.view(torch.int16).numpy()})
        n += 1
    out["n_perm"] = np.array(n)
    np.savez_compressed(OUT / "moe_ops.npz", **out)
    print("wrote", OUT / "moe_ops.npz", (OUT / "moe_ops.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
