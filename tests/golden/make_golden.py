#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING the reference's own oracle functions.

The reference package cannot be imported in this container (missing zmq, msgspec, ... -- SURVEY.md
8c), so this script parses the reference source files under /root/reference with `ast`, pulls out
the pure-torch oracle FUNCTIONS the reference's tests use for this path, and exec()s them as they
are, with tiny stand-ins only for the names they look up (enum members, the native SiluAndMul).
Nothing of the reference is copied into the repo: only the numeric input/output vectors are saved.

Functions executed (paths relative to /root/reference):
  torch_topk                        tests/kernels/moe/test_fused_topk.py:18-44
  grouped_topk                      vllm/model_executor/layers/fused_moe/router/grouped_topk_router.py:80-161
  determine_expert_map              vllm/model_executor/layers/fused_moe/expert_map_manager.py:22-113
  ref_fused_moe                     tests/kernels/moe/test_cpu_fused_moe.py:46-107
  _swigluoai_forward_native         vllm/model_executor/layers/fused_moe/cpu_fused_moe.py:30-46
  SiluAndMul.forward_native         vllm/model_executor/layers/activation.py:140-143
  torch_experts / torch_moe         tests/kernels/utils.py:855-1021
  quantize_weights                  vllm/model_executor/layers/quantization/utils/quant_utils.py:642-738
  scalar_types.uint4b8 / uint4 / uint8b128 / uint8   vllm/scalar_type.py
  native_per_token_group_quant_fp8  tests/kernels/quant_utils.py:157-180
  native_w8a8_block_matmul          tests/kernels/quant_utils.py:91-154
  torch_w8a8_block_fp8_moe          tests/kernels/moe/test_block_fp8.py:107-137
  dq_mxfp4_torch (+ e8m0_to_half, upcast_fp4_to_fp16_or_bf16)   tests/quantization/reference_mxfp4.py:28-117
  RoutedExperts._load_w13 / _load_w2 / _load_model_weight_or_group_weight_scale /
    _narrow_expert_data_for_padding / _get_hidden_dim   vllm/model_executor/layers/fused_moe/routed_experts.py:383-612
  dequantize_nvfp4_to_dtype (+ break_fp4_bytes, convert_swizzled_to_linear)
                                    tests/kernels/quantization/nvfp4_utils.py:16-87

Run here (needs /root/reference):   python tests/golden/make_golden.py
The GPU box never runs this; it only reads the committed .npz files.
"""
from __future__ import annotations

import ast
import enum
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def extract(path: str, names: list[str], ns: dict, cls: str | None = None, strip_decorators=True):
    """exec the named top-level functions (or methods of `cls`) of a reference file into ns."""
    src = (REF / path).read_text()
    tree = ast.parse(src)
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    found = []
    for node in body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            if strip_decorators:
                node.decorator_list = []
            mod = ast.Module(body=[node], type_ignores=[])
            code = compile(ast.fix_missing_locations(mod), str(REF / path), "exec")
            exec(code, ns)
            found.append(node.name)
    missing = set(names) - set(found)
    if missing:
        raise RuntimeError(f"{path}: functions not found: {missing}")
    return ns


def extract_assigns(path: str, ns: dict, upto_line: int = 10**9):
    """exec the top-level constant assignments of a reference file (module constants the extracted
    functions look up)."""
    tree = ast.parse((REF / path).read_text())
    for node in tree.body:
        if isinstance(node, ast.Assign) and node.lineno < upto_line:
            try:
                exec(compile(ast.fix_missing_locations(ast.Module(body=[node], type_ignores=[])),
                             str(REF / path), "exec"), ns)
            except Exception:
                pass   # assignments that need names we do not provide are not used by our functions
    return ns


def bits(t: torch.Tensor) -> np.ndarray:
    """bf16/fp16 tensor -> uint16 bit patterns; fp8 -> uint8; others unchanged."""
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.contiguous().view(torch.int16).numpy().view(np.uint16)
    if t.dtype == torch.float8_e4m3fn:
        return t.contiguous().view(torch.uint8).numpy()
    return t.contiguous().numpy()


# ----------------------------------------------------------------------------- stand-ins
class MoEActivation(enum.Enum):
    SILU = "silu"
    SWIGLUOAI = "swigluoai"

    @property
    def custom_op_name(self):
        return {"silu": "silu_and_mul", "swigluoai": "swigluoai_and_mul"}[self.value]


def base_ns() -> dict:
    ns = {"torch": torch, "F": F, "np": np, "MoEActivation": MoEActivation}
    # the reference's own native SiluAndMul
    extract("vllm/model_executor/layers/activation.py", ["forward_native"], ns, cls="SiluAndMul")
    silu_native = ns.pop("forward_native")

    class SiluAndMul:  # shape of vllm.model_executor.layers.activation.SiluAndMul
        forward_native = staticmethod(silu_native)

        def __call__(self, x):
            return silu_native(x)

    ns["SiluAndMul"] = SiluAndMul
    extract("vllm/model_executor/layers/fused_moe/cpu_fused_moe.py", ["_swigluoai_forward_native"], ns)
    ns["_CPU_MOE_ACT_FN"] = {MoEActivation.SILU: silu_native,
                             MoEActivation.SWIGLUOAI: ns["_swigluoai_forward_native"]}
    ns["op_registry"] = {"silu_and_mul": SiluAndMul}
    return ns


# ----------------------------------------------------------------------------- generators
def gen_topk(ns):
    extract("tests/kernels/moe/test_fused_topk.py", ["torch_topk"], ns)
    torch_topk = ns["torch_topk"]
    cases = {}
    idx = 0
    # the reference's own grid (test_fused_topk.py:47-56) + the model shapes of SURVEY 8
    grid = [(m, e, k) for m in (1, 33, 56) for e in (6, 16) for k in (3, 4)]
    grid += [(32, 8, 2), (5, 128, 8), (7, 256, 8), (3, 384, 8)]
    for (m, e, k) in grid:
        for renorm in (True, False):
            for scoring in ("softmax", "sigmoid"):
                for dtype in (torch.float32, torch.bfloat16, torch.float16):
                    for with_bias in (False, True):
                        if with_bias and (idx % 3):
                            idx += 1
                            continue
                        torch.manual_seed(0)
                        _hidden = torch.randn((m, 64), dtype=dtype)
                        gating = torch.randn((m, e), dtype=dtype)
                        bias = torch.randn((e,), dtype=torch.float32) if with_bias else None
                        w, ids = torch_topk(gating_output=gating, topk=k, renormalize=renorm,
                                            e_score_correction_bias=bias, scoring_func=scoring)
                        key = f"c{idx}"
                        cases[key + "_logits"] = bits(gating)
                        cases[key + "_meta"] = np.array(
                            [m, e, k, int(renorm), 0 if scoring == "softmax" else 1,
                             {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype],
                             int(with_bias)], np.int32)
                        if with_bias:
                            cases[key + "_bias"] = bias.numpy()
                        cases[key + "_w"] = w.float().numpy()
                        cases[key + "_ids"] = ids.to(torch.int32).numpy()
                        idx += 1
    cases["n"] = np.array(idx, np.int32)
    np.savez_compressed(OUT / "topk.npz", **cases)
    print("topk.npz:", idx, "cases")


def gen_grouped(ns):
    envs = types.SimpleNamespace(VLLM_USE_FUSED_MOE_GROUPED_TOPK=False, VLLM_BATCH_INVARIANT=False)
    plat = types.SimpleNamespace(is_cuda=lambda: False)
    ns2 = dict(ns, envs=envs, current_platform=plat)
    extract("vllm/model_executor/layers/fused_moe/router/grouped_topk_router.py", ["grouped_topk"], ns2)
    grouped_topk = ns2["grouped_topk"]
    cases = {}
    idx = 0
    # shapes of tests/kernels/moe/test_grouped_topk.py:77-86
    for m in (1, 33):
        for (e, k, ng, tg) in ((16, 2, 8, 2), (128, 2, 8, 2), (256, 8, 8, 4), (384, 8, 1, 1), (128, 8, 1, 1)):
            for renorm in (True, False):
                for scoring in ("softmax", "sigmoid"):
                    for rsf in (1.0, 2.5):
                        for with_bias in (True, False):
                            torch.manual_seed(idx)
                            logits = torch.randn((m, e), dtype=torch.float32)
                            bias = torch.randn((e,), dtype=torch.float32) if with_bias else None
                            w, ids = grouped_topk(torch.empty((m, 0)), logits, k, renorm, ng, tg,
                                                  scoring, rsf, bias)
                            key = f"c{idx}"
                            cases[key + "_logits"] = logits.numpy()
                            cases[key + "_meta"] = np.array(
                                [m, e, k, ng, tg, int(renorm), 0 if scoring == "softmax" else 1,
                                 int(with_bias)], np.int32)
                            cases[key + "_rsf"] = np.array(rsf, np.float32)
                            if with_bias:
                                cases[key + "_bias"] = bias.numpy()
                            cases[key + "_w"] = w.numpy()
                            cases[key + "_ids"] = ids.numpy()
                            idx += 1
    cases["n"] = np.array(idx, np.int32)
    np.savez_compressed(OUT / "grouped_topk.npz", **cases)
    print("grouped_topk.npz:", idx, "cases")


def gen_expert_map(ns):
    ns2 = dict(ns, ExpertPlacementStrategy=str)
    extract("vllm/model_executor/layers/fused_moe/expert_map_manager.py", ["determine_expert_map"], ns2)
    f = ns2["determine_expert_map"]
    cases = {}
    idx = 0
    for E in (8, 10, 128, 256, 7):
        for ep in (1, 2, 3, 4, 8):
            for strat in ("linear", "round_robin"):
                for r in range(ep):
                    n, emap, _ = f(ep, r, E, strat)
                    cases[f"c{idx}_meta"] = np.array([ep, r, E, 0 if strat == "linear" else 1, n], np.int32)
                    cases[f"c{idx}_map"] = (emap.numpy() if emap is not None
                                            else np.arange(E, dtype=np.int32))
                    idx += 1
    cases["n"] = np.array(idx, np.int32)
    np.savez_compressed(OUT / "expert_map.npz", **cases)
    print("expert_map.npz:", idx, "cases")


def _route(score: torch.Tensor, k: int):
    p = torch.softmax(score.float(), dim=-1)
    w, ids = torch.topk(p, k)
    w = w / w.sum(dim=-1, keepdim=True)
    return w.float(), ids.to(torch.int32)


def gen_moe_bf16(ns):
    extract("tests/kernels/moe/test_cpu_fused_moe.py", ["ref_fused_moe"], ns)
    ns2 = dict(ns, moe_kernel_quantize_input=lambda a, s, qd, pt, bs=None: (a, None),
               native_w8a8_block_matmul=None)
    extract("tests/kernels/utils.py", ["torch_experts"], ns2)
    ref_fused_moe, torch_experts = ns["ref_fused_moe"], ns2["torch_experts"]
    cases = {}
    idx = 0
    # (M, I, H, E, K): small members of test_moe.py:292-303 / test_cpu_fused_moe.py grids
    for (m, n, k, e, topk, act) in ((1, 128, 128, 8, 2, "silu"), (33, 256, 128, 4, 2, "silu"),
                                    (17, 64, 256, 16, 6, "silu"), (33, 128, 128, 4, 2, "swigluoai"),
                                    (64, 128, 256, 4, 1, "silu")):
        for dtype in (torch.bfloat16, torch.float16):
            if dtype == torch.float16 and (m, n) not in ((1, 128), (33, 256)):
                continue
            torch.manual_seed(7)
            a = torch.randn((m, k), dtype=dtype) / 10
            w1 = torch.randn((e, 2 * n, k), dtype=dtype) / 10
            w2 = torch.randn((e, k, n), dtype=dtype) / 10
            score = torch.randn((m, e), dtype=dtype)
            tw, ids = _route(score, topk)
            activation = MoEActivation.SILU if act == "silu" else MoEActivation.SWIGLUOAI
            out_cpu = ref_fused_moe(a, w1, w2, None, None, tw, ids.long(), activation)
            key = f"c{idx}"
            cases[key + "_meta"] = np.array([m, n, k, e, topk, 1 if dtype == torch.bfloat16 else 2,
                                             0 if act == "silu" else 1], np.int32)
            cases[key + "_a"] = bits(a)
            cases[key + "_w1"] = bits(w1)
            cases[key + "_w2"] = bits(w2)
            cases[key + "_tw"] = tw.numpy()
            cases[key + "_ids"] = ids.numpy()
            cases[key + "_out_cpu"] = bits(out_cpu)           # test_cpu_fused_moe oracle
            if act == "silu":
                out_gpu = torch_experts(a, w1, w2, tw, ids.long())
                cases[key + "_out_gpu"] = bits(out_gpu)       # in-tree GPU-operator oracle
            idx += 1
    cases["n"] = np.array(idx, np.int32)
    np.savez_compressed(OUT / "moe_dense.npz", **cases)
    print("moe_dense.npz:", idx, "cases")


def gen_moe_int4(ns):
    spec = importlib.util.spec_from_file_location("ref_scalar_type", REF / "vllm/scalar_type.py")
    st = importlib.util.module_from_spec(spec)
    sys.modules["ref_scalar_type"] = st
    spec.loader.exec_module(st)
    ns2 = dict(ns, ScalarType=st.ScalarType, scalar_types=st.scalar_types)
    extract("vllm/model_executor/layers/quantization/utils/quant_utils.py", ["quantize_weights"], ns2)
    extract("tests/kernels/moe/test_cpu_fused_moe.py", ["ref_fused_moe"], ns2)
    quantize_weights, ref_fused_moe = ns2["quantize_weights"], ns2["ref_fused_moe"]
    cases = {}
    idx = 0
    for (m, n, k, e, topk, g) in ((1, 128, 128, 4, 2, 128), (33, 256, 128, 4, 2, 64),
                                  (20, 128, 256, 4, 2, 32), (33, 128, 256, 4, 2, 128)):
        for dtype in (torch.bfloat16, torch.float16):
            if dtype == torch.float16 and g != 64:
                continue
            torch.manual_seed(7)
            a = torch.randn((m, k), dtype=dtype) / 10
            w1 = torch.randn((e, 2 * n, k), dtype=dtype) / 10
            w2 = torch.randn((e, k, n), dtype=dtype) / 10
            score = torch.randn((m, e), dtype=dtype)
            tw, ids = _route(score, topk)
            packs, scales, refs = [], [], []
            for w in (w1, w2):
                qw, sc, rf = [], [], []
                for i in range(e):
                    # test_moe.py:634-641
                    weight, qweight, s, _ = quantize_weights(w[i].T, st.scalar_types.uint4b8, g, False, False)
                    weight = weight.T
                    qweight = qweight.T.contiguous().to(torch.uint8)
                    qweight = qweight[:, 1::2] * 16 + qweight[:, ::2]
                    qw.append(qweight)
                    sc.append(s.T.contiguous())
                    rf.append(weight.contiguous())
                packs.append(torch.stack(qw))
                scales.append(torch.stack(sc))
                refs.append(torch.stack(rf))
            out = ref_fused_moe(a, refs[0], refs[1], None, None, tw, ids.long(), MoEActivation.SILU)
            key = f"c{idx}"
            cases[key + "_meta"] = np.array([m, n, k, e, topk, g, 1 if dtype == torch.bfloat16 else 2], np.int32)
            cases[key + "_a"] = bits(a)
            cases[key + "_w1"] = bits(w1)
            cases[key + "_w2"] = bits(w2)
            cases[key + "_q1"] = packs[0].numpy()
            cases[key + "_q2"] = packs[1].numpy()
            cases[key + "_s1"] = bits(scales[0])
            cases[key + "_s2"] = bits(scales[1])
            if idx == 0:   # dequantised reference weights: one case is enough to pin the dequant
                cases[key + "_ref1"] = bits(refs[0])
                cases[key + "_ref2"] = bits(refs[1])
            cases[key + "_tw"] = tw.numpy()
            cases[key + "_ids"] = ids.numpy()
            cases[key + "_out"] = bits(out)
            idx += 1
    cases["n"] = np.array(idx, np.int32)
    np.savez_compressed(OUT / "moe_int4.npz", **cases)
    print("moe_int4.npz:", idx, "cases")


def gen_moe_wna16(ns):
    """the has_zp x weight_bits grid of tests/kernels/moe/test_moe.py:565-693 (test_fused_moe_wn16): quantize_weights
    with uint4 / uint4b8 / uint8 / uint8b128, packed as the test packs them; expected = torch_moe on w_ref"""
    spec = importlib.util.spec_from_file_location("ref_scalar_type", REF / "vllm/scalar_type.py")
    st = importlib.util.module_from_spec(spec)
    sys.modules["ref_scalar_type"] = st
    spec.loader.exec_module(st)
    ns2 = dict(ns, ScalarType=st.ScalarType, scalar_types=st.scalar_types,
               moe_kernel_quantize_input=lambda a, s, qd, pt, bs=None: (a, None), native_w8a8_block_matmul=None)
    extract("vllm/model_executor/layers/quantization/utils/quant_utils.py", ["quantize_weights"], ns2)
    extract("tests/kernels/utils.py", ["torch_experts"], ns2)
    quantize_weights, torch_experts = ns2["quantize_weights"], ns2["torch_experts"]
    cases = {}
    idx = 0
    dtype = torch.bfloat16
    for (m, n, k, e, topk) in ((1, 128, 128, 4, 2), (33, 128, 256, 4, 2)):
        for g in ((64, 128) if m == 1 else (128,)):
            for has_zp in (True, False):
                for weight_bits in (4, 8):
                    torch.manual_seed(7)
                    a = torch.randn((m, k), dtype=dtype) / 10
                    w1 = torch.randn((e, 2 * n, k), dtype=dtype) / 10
                    w2 = torch.randn((e, k, n), dtype=dtype) / 10
                    score = torch.randn((m, e), dtype=dtype)
                    # test_moe.py:590-596
                    if weight_bits == 4:
                        quant_type = st.scalar_types.uint4 if has_zp else st.scalar_types.uint4b8
                    else:
                        quant_type = st.scalar_types.uint8 if has_zp else st.scalar_types.uint8b128
                    packs, scales, zeros, refs = [], [], [], []
                    for w in (w1, w2):
                        qw, sc, zz, rf = [], [], [], []
                        for i in range(e):
                            # test_moe.py:632-651
                            weight, qweight, s_, qzeros = quantize_weights(w[i].T, quant_type, g, has_zp, False)
                            weight = weight.T
                            qweight = qweight.T.contiguous().to(torch.uint8)
                            s_ = s_.T
                            if has_zp:
                                qzeros = qzeros.T.contiguous().to(torch.uint8)
                            if weight_bits == 4:
                                qweight = qweight[:, 1::2] * 16 + qweight[:, ::2]
                                if has_zp:
                                    qzeros = qzeros[1::2, :] * 16 + qzeros[::2, :]
                            qw.append(qweight)
                            sc.append(s_.contiguous())
                            rf.append(weight.contiguous())
                            if has_zp:
                                zz.append(qzeros)
                        packs.append(torch.stack(qw))
                        scales.append(torch.stack(sc))
                        refs.append(torch.stack(rf))
                        zeros.append(torch.stack(zz) if has_zp else None)
                    # fused_moe(renormalize=False) vs torch_moe: softmax scores, top-k, no renormalisation
                    p_ = torch.softmax(score.float(), dim=-1)
                    tw, ids = torch.topk(p_, topk)
                    out = torch_experts(a, refs[0], refs[1], tw.float(), ids.long())
                    key = f"c{idx}"
                    cases[key + "_meta"] = np.array([m, n, k, e, topk, g, int(has_zp), weight_bits], np.int32)
                    cases[key + "_a"] = bits(a)
                    cases[key + "_q1"] = packs[0].numpy()
                    cases[key + "_q2"] = packs[1].numpy()
                    cases[key + "_s1"] = bits(scales[0])
                    cases[key + "_s2"] = bits(scales[1])
                    if has_zp:
                        cases[key + "_z1"] = zeros[0].numpy()
                        cases[key + "_z2"] = zeros[1].numpy()
                    if m == 1 and g == 64:   # dequantised reference weights: one case per (has_zp, bits) pins the dequant
                        cases[key + "_ref1"] = bits(refs[0])
                        cases[key + "_ref2"] = bits(refs[1])
                    cases[key + "_tw"] = tw.float().numpy()
                    cases[key + "_ids"] = ids.to(torch.int32).numpy()
                    cases[key + "_out"] = bits(out)
                    idx += 1
    cases["n"] = np.array(idx, np.int32)
    np.savez_compressed(OUT / "moe_wna16.npz", **cases)
    print("moe_wna16.npz:", idx, "cases")


def gen_moe_fp8(ns):
    ns2 = dict(ns, is_deep_gemm_e8m0_used=lambda: False, _ceil_to_ue8m0=None,
               FP8_DTYPE=torch.float8_e4m3fn)
    extract("tests/kernels/quant_utils.py",
            ["native_per_token_group_quant_fp8", "native_w8a8_block_matmul"], ns2)
    extract("tests/kernels/moe/test_block_fp8.py", ["torch_w8a8_block_fp8_moe"], ns2)
    f = ns2["torch_w8a8_block_fp8_moe"]
    cases = {}
    idx = 0
    for (m, n, k, e, topk) in ((1, 128, 128, 2, 1), (33, 256, 128, 8, 2), (16, 128, 256, 8, 6)):
        dtype = torch.bfloat16
        torch.manual_seed(0)
        a = torch.randn((m, k), dtype=dtype) / 10
        score = torch.randn((m, e), dtype=dtype)
        w1f = torch.randn((e, 2 * n, k), dtype=torch.float32) / 10
        w2f = torch.randn((e, k, n), dtype=torch.float32) / 10
        blk = [128, 128]

        def blockq(w):
            E_, N_, K_ = w.shape
            nb, kb = -(-N_ // 128), -(-K_ // 128)
            q = torch.empty_like(w, dtype=torch.float8_e4m3fn)
            s = torch.empty((E_, nb, kb), dtype=torch.float32)
            for e_ in range(E_):
                for i in range(nb):
                    for j in range(kb):
                        t = w[e_, i * 128:(i + 1) * 128, j * 128:(j + 1) * 128]
                        sc = t.abs().max().clamp(min=1e-4) / 448.0
                        s[e_, i, j] = sc
                        q[e_, i * 128:(i + 1) * 128, j * 128:(j + 1) * 128] = (t / sc).to(torch.float8_e4m3fn)
            return q, s

        w1, w1s = blockq(w1f)
        w2, w2s = blockq(w2f)
        p = torch.softmax(score.float(), dim=-1)
        tw, ids = torch.topk(p, topk)      # fused_topk(..., renormalize=False), test_block_fp8.py:176
        out = f(a, w1, w2, w1s, w2s, tw, ids, blk)
        key = f"c{idx}"
        cases[key + "_meta"] = np.array([m, n, k, e, topk], np.int32)
        cases[key + "_a"] = bits(a)
        cases[key + "_w1"] = bits(w1)
        cases[key + "_w2"] = bits(w2)
        cases[key + "_w1s"] = w1s.numpy()
        cases[key + "_w2s"] = w2s.numpy()
        cases[key + "_tw"] = tw.float().numpy()
        cases[key + "_ids"] = ids.to(torch.int32).numpy()
        cases[key + "_out"] = bits(out)
        idx += 1
    cases["n"] = np.array(idx, np.int32)
    np.savez_compressed(OUT / "moe_fp8_block.npz", **cases)
    print("moe_fp8_block.npz:", idx, "cases")


def gen_moe_fp4(ns):
    """MXFP4 / NVFP4 expert weights: the reference's own dequantisers produce the dense weights, the
    reference's CPU MoE oracle (ref_fused_moe) the outputs."""
    nm = dict(ns)
    extract_assigns("tests/quantization/reference_mxfp4.py", nm)
    extract("tests/quantization/reference_mxfp4.py",
            ["e8m0_to_half", "upcast_fp4_to_fp16_or_bf16", "dq_mxfp4_torch"], nm)
    nn_ = dict(ns)
    nn_["kE2M1ToFloat"] = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], dtype=torch.float32)
    extract("tests/kernels/quantization/nvfp4_utils.py",
            ["convert_swizzled_to_linear", "convert_swizzled_8x4_layout_to_linear",
             "dequantize_nvfp4_to_dtype", "break_fp4_bytes"], nn_)
    extract("tests/kernels/moe/test_cpu_fused_moe.py", ["ref_fused_moe"], ns)
    dq_mx, dq_nv, to_linear = nm["dq_mxfp4_torch"], nn_["dequantize_nvfp4_to_dtype"], nn_["convert_swizzled_to_linear"]
    ref_fused_moe = ns["ref_fused_moe"]
    cases = {}
    idx = 0
    for fmt in ("mxfp4", "nvfp4"):
        for (m, n, k, e, topk) in ((1, 128, 128, 4, 2), (33, 256, 128, 4, 2), (20, 128, 256, 8, 2)):
            for dtype in (torch.bfloat16, torch.float16):
                if dtype == torch.float16 and m != 33:
                    continue
                g = torch.Generator().manual_seed(100 + idx)
                a = (torch.randn((m, k), generator=g) / 10).to(dtype)
                score = torch.randn((m, e), generator=g).to(dtype)
                tw, ids = _route(score, topk)
                q1 = torch.randint(0, 256, (e, 2 * n, k // 2), generator=g, dtype=torch.uint8)
                q2 = torch.randint(0, 256, (e, k, n // 2), generator=g, dtype=torch.uint8)
                key = f"c{idx}"
                if fmt == "mxfp4":
                    s1 = torch.randint(119, 125, (e, 2 * n, k // 32), generator=g, dtype=torch.uint8)
                    s2 = torch.randint(119, 125, (e, k, n // 32), generator=g, dtype=torch.uint8)
                    w1 = dq_mx(q1, s1, dtype)
                    w2 = dq_mx(q2, s2, dtype)
                    cases[key + "_s1"], cases[key + "_s2"] = s1.numpy(), s2.numpy()
                else:
                    # block scales: random fp8 e4m3fn bytes (finite, positive), generated in the reference's
                    # 128x4-swizzled storage and un-swizzled with the reference's own helper
                    def sf(rows, kk):
                        mt, kt = (rows + 127) // 128, (kk // 16 + 3) // 4
                        raw = torch.randint(0x28, 0x48, (mt * 128 * kt * 4,), generator=g, dtype=torch.uint8)
                        return raw, mt, kt
                    gs1 = torch.tensor([2.0 ** (3 + (i % 3)) for i in range(e)])      # global scales (powers of two:
                    gs2 = torch.tensor([2.0 ** (2 + (i % 2)) for i in range(e)])      #  sf / g == sf * (1/g) exactly)
                    w1l, w2l, s1l, s2l = [], [], [], []
                    for i in range(e):
                        for (q, rows, kk, gsc, wl, sl) in ((q1[i], 2 * n, k, gs1[i], w1l, s1l), (q2[i], k, n, gs2[i], w2l, s2l)):
                            raw, mt, kt = sf(rows, kk)
                            swz = raw.view(torch.float8_e4m3fn)
                            wl.append(dq_nv(q, swz, gsc, dtype, "cpu", block_size=16, is_sf_128x4_layout=True))
                            lin = to_linear(raw, rows, kk, 16)       # the same scales, linear [rows, kk/16]
                            sl.append(lin.contiguous())
                    w1, w2 = torch.stack(w1l), torch.stack(w2l)
                    cases[key + "_s1"], cases[key + "_s2"] = torch.stack(s1l).numpy(), torch.stack(s2l).numpy()
                    cases[key + "_gs1"], cases[key + "_gs2"] = (1.0 / gs1).numpy().astype(np.float32), (1.0 / gs2).numpy().astype(np.float32)
                out = ref_fused_moe(a, w1, w2, None, None, tw, ids.long(), MoEActivation.SILU)
                cases[key + "_meta"] = np.array([m, n, k, e, topk, 0 if fmt == "mxfp4" else 1,
                                                 1 if dtype == torch.bfloat16 else 2], np.int32)
                cases[key + "_a"] = bits(a)
                cases[key + "_q1"], cases[key + "_q2"] = q1.numpy(), q2.numpy()
                if m == 1 or dtype == torch.float16:   # the reference's dequantised weights (pins the dequant)
                    cases[key + "_w1"], cases[key + "_w2"] = bits(w1), bits(w2)
                cases[key + "_tw"], cases[key + "_ids"] = tw.numpy(), ids.numpy()
                cases[key + "_out"] = bits(out)
                idx += 1
    cases["n"] = np.array(idx, np.int32)
    np.savez_compressed(OUT / "moe_fp4.npz", **cases)
    print("moe_fp4.npz:", idx, "cases")


def gen_ingest(ns):
    """SURVEY 8(f1) weight ingest: a tiny synthetic checkpoint (per-expert gate/up/down tensors) is loaded
    with the reference's own RoutedExperts loader helpers (_load_w13, _load_w2,
    _load_model_weight_or_group_weight_scale, _narrow_expert_data_for_padding, _get_hidden_dim;
    routed_experts.py:383-612) for every (tp_rank, ep_rank), then put through the _process_* hand-off
    (routed_experts.py:1457-1466: the compressed-tensors int4 parameters are stored transposed and are
    transposed back + viewed as uint8).  Saved: the checkpoint tensors and, per rank, the tensors the
    reference would pass to lk_moe."""
    import functools
    import types
    fns = {}
    extract("vllm/model_executor/layers/fused_moe/routed_experts.py",
            ["_load_w13", "_load_w2", "_load_model_weight_or_group_weight_scale", "_narrow_expert_data_for_padding",
             "_get_hidden_dim"], fns, cls="RoutedExperts")
    extract("vllm/model_executor/layers/fused_moe/expert_map_manager.py", ["determine_expert_map"],
            ns_em := dict(ns, get_compute_capability=None, logger=types.SimpleNamespace(info=lambda *a, **k: None,
                                                                                       warning=lambda *a, **k: None)))

    def make_self(tp_size, tp_rank):
        me = types.SimpleNamespace()
        me.moe_config = types.SimpleNamespace(is_act_and_mul=True, tp_rank=tp_rank, tp_size=tp_size,
                                              moe_parallel_config=types.SimpleNamespace(tp_size=tp_size))
        me._get_hidden_dim = fns["_get_hidden_dim"]
        me._narrow_expert_data_for_padding = fns["_narrow_expert_data_for_padding"]
        me._load_w13 = functools.partial(fns["_load_w13"], me)
        me._load_w2 = functools.partial(fns["_load_w2"], me)
        me.load = functools.partial(fns["_load_model_weight_or_group_weight_scale"], me)
        return me

    g = torch.Generator().manual_seed(4242)
    cases = {}
    idx = 0
    for quant in ("none", "fp8_block", "int4", "mxfp4"):
        E, H, I = (2, 128, 256) if quant == "fp8_block" else (4, 64, 64)
        # ---- the checkpoint: per expert, w1/w3 [I, K=H], w2 [H, K=I] in the checkpoint's own packing
        ck = {}
        for e in range(E):
            for sid, (n, k) in (("w1", (I, H)), ("w3", (I, H)), ("w2", (H, I))):
                if quant == "none":
                    ck[(e, sid, "weight")] = torch.randn((n, k), generator=g).to(torch.bfloat16)
                elif quant == "fp8_block":
                    ck[(e, sid, "weight")] = torch.randint(0, 256, (n, k), generator=g, dtype=torch.uint8)
                    ck[(e, sid, "weight_scale_inv")] = torch.rand((n // 128, k // 128), generator=g) + 0.5
                elif quant == "int4":      # compressed-tensors pack-quantized: int32 [N, K/8], scales [N, K/g]
                    ck[(e, sid, "weight_packed")] = torch.randint(-2**31, 2**31 - 1, (n, k // 8), generator=g, dtype=torch.int32)
                    ck[(e, sid, "weight_scale")] = (torch.rand((n, k // 32), generator=g) / 8).to(torch.bfloat16)
                else:                      # mxfp4: uint8 [N, K/2], E8M0 [N, K/32]
                    ck[(e, sid, "weight_packed")] = torch.randint(0, 256, (n, k // 2), generator=g, dtype=torch.uint8)
                    ck[(e, sid, "weight_scale")] = torch.randint(118, 130, (n, k // 32), generator=g, dtype=torch.uint8)
        key = f"c{idx}"
        for (e, sid, kind), t in ck.items():
            cases[f"{key}_ck_{e}_{sid}_{kind}"] = bits(t) if t.dtype == torch.bfloat16 else t.numpy()
        ranks = []
        for (tp, ep) in ((1, 1), (2, 1), (1, 2), (2, 2)):
            for tp_rank in range(tp):
                for ep_rank in range(ep):
                    n_local, emap, _ = ns_em["determine_expert_map"](ep, ep_rank, E) if ep > 1 else (E, None, None)
                    Ip = I // tp
                    me = make_self(tp, tp_rank)
                    transposed = quant == "int4"     # CompressedTensorsWNA16MoEMethod: is_transposed params
                    P = {}
                    if quant == "none":
                        P["w13_weight"] = torch.zeros((n_local, 2 * Ip, H), dtype=torch.bfloat16)
                        P["w2_weight"] = torch.zeros((n_local, H, Ip), dtype=torch.bfloat16)
                    elif quant == "fp8_block":
                        P["w13_weight"] = torch.zeros((n_local, 2 * Ip, H), dtype=torch.uint8)
                        P["w2_weight"] = torch.zeros((n_local, H, Ip), dtype=torch.uint8)
                        P["w13_weight_scale_inv"] = torch.zeros((n_local, 2 * Ip // 128, H // 128))
                        P["w2_weight_scale_inv"] = torch.zeros((n_local, H // 128, max(1, Ip // 128)))
                    elif quant == "int4":
                        P["w13_weight_packed"] = torch.zeros((n_local, H // 8, 2 * Ip), dtype=torch.int32)
                        P["w2_weight_packed"] = torch.zeros((n_local, Ip // 8, H), dtype=torch.int32)
                        P["w13_weight_scale"] = torch.zeros((n_local, H // 32, 2 * Ip), dtype=torch.bfloat16)
                        P["w2_weight_scale"] = torch.zeros((n_local, Ip // 32, H), dtype=torch.bfloat16)
                    else:
                        P["w13_weight_packed"] = torch.zeros((n_local, 2 * Ip, H // 2), dtype=torch.uint8)
                        P["w2_weight_packed"] = torch.zeros((n_local, H, Ip // 2), dtype=torch.uint8)
                        P["w13_weight_scale"] = torch.zeros((n_local, 2 * Ip, H // 32), dtype=torch.uint8)
                        P["w2_weight_scale"] = torch.zeros((n_local, H, Ip // 32), dtype=torch.uint8)
                    for (e, sid, kind), t in ck.items():
                        le = e if emap is None else int(emap[e])
                        if le < 0:
                            continue
                        pname = ("w13_" if sid in ("w1", "w3") else "w2_") + kind
                        lw = t.t().contiguous() if transposed else t                      # weight_loader :707-716
                        shard_dim = {"w1": 0, "w2": 1, "w3": 0}[sid]
                        if transposed:
                            shard_dim = int(not shard_dim)                                 # :763-765
                        me.load(shard_dim=shard_dim, expert_data=P[pname][le], shard_id=sid, loaded_weight=lw,
                                tp_rank=tp_rank)
                    if quant == "int4":                                                    # _process_wna16 :1457-1461
                        out = {"w13": P["w13_weight_packed"].transpose(1, 2).contiguous().view(torch.uint8),
                               "w2": P["w2_weight_packed"].transpose(1, 2).contiguous().view(torch.uint8),
                               "s13": P["w13_weight_scale"].transpose(1, 2).contiguous(),
                               "s2": P["w2_weight_scale"].transpose(1, 2).contiguous()}
                    elif quant == "none":
                        out = {"w13": P["w13_weight"], "w2": P["w2_weight"]}
                    elif quant == "fp8_block":
                        out = {"w13": P["w13_weight"], "w2": P["w2_weight"], "s13": P["w13_weight_scale_inv"],
                               "s2": P["w2_weight_scale_inv"]}
                    else:
                        out = {"w13": P["w13_weight_packed"], "w2": P["w2_weight_packed"],
                               "s13": P["w13_weight_scale"], "s2": P["w2_weight_scale"]}
                    tag = f"{key}_r{len(ranks)}"
                    ranks.append((tp, tp_rank, ep, ep_rank))
                    for nm, t in out.items():
                        cases[f"{tag}_{nm}"] = bits(t) if t.dtype == torch.bfloat16 else t.numpy()
        cases[key + "_ranks"] = np.array(ranks, np.int32)
        cases[key + "_meta"] = np.array([E, H, I, {"none": 0, "fp8_block": 1, "int4": 2, "mxfp4": 3}[quant]], np.int32)
        idx += 1
    cases["n"] = np.array(idx, np.int32)
    np.savez_compressed(OUT / "ingest.npz", **cases)
    print("ingest.npz:", idx, "cases")


def gen_residency(ns):
    """the reference's layer-tier predicates (vllm/envs.py:2356-2407) evaluated over a grid of
    LVLLM_GPU_RESIDENT_MOE_LAYERS specs, thresholds and layer names."""
    import json
    env_state = {}
    nr = {"environment_variables": {"LVLLM_GPU_RESIDENT_MOE_LAYERS": lambda: env_state["spec"],
                                    "LVLLM_MOE_NUMA_ENABLED": lambda: env_state["on"],
                                    "LVLLM_GPU_PREFILL_MIN_BATCH_SIZE": lambda: env_state["thr"]}}
    extract("vllm/model_executor/models/utils.py", ["extract_layer_index"], nr)
    extract("vllm/envs.py", ["is_lk_moe_feature_enabled", "is_lk_moe_use_gpu_prefill", "is_lk_moe_mtp_layer",
                             "is_lk_moe_gpu_prefill_layer", "is_lk_moe_cpu_layer", "is_lk_moe_gpu_resident_layer"], nr)
    rows = []
    names = ["model.layers.0.mlp.experts", "model.layers.5.mlp.experts", "model.layers.7.block_sparse_moe",
             "model.layers.12.mlp.experts", "mtp.layers.0.mlp.experts", "model.layers.61.mlp.experts"]
    for spec in ["", "0-5,7", " 3 , 9-8, x, 12-12,", "0-100", "5", "a-b,7-", "61,0"]:
        for on in (True, False):
            for thr in (0, 256):
                env_state.update(spec=spec, on=on, thr=thr)
                for nm in names:
                    rows.append(dict(spec=spec, on=on, thr=thr, name=nm,
                                     resident=bool(nr["is_lk_moe_gpu_resident_layer"](nm)),
                                     prefill=bool(nr["is_lk_moe_gpu_prefill_layer"](nm)),
                                     cpu=bool(nr["is_lk_moe_cpu_layer"](nm))))
    (OUT / "residency.json").write_text(json.dumps(rows))
    print("residency.json:", len(rows), "rows")


def main():
    if not REF.exists():
        sys.exit("needs /root/reference (run in the build container, not on the GPU box)")
    torch.set_num_threads(8)
    ns = base_ns()
    gens = {"topk": gen_topk, "grouped": gen_grouped, "expert_map": gen_expert_map, "bf16": gen_moe_bf16,
            "int4": gen_moe_int4, "wna16": gen_moe_wna16, "fp8": gen_moe_fp8, "fp4": gen_moe_fp4, "ingest": gen_ingest, "residency": gen_residency}
    for name in (sys.argv[1:] or list(gens)):    # `make_golden.py fp4` regenerates one file only
        gens[name](dict(ns))


if __name__ == "__main__":
    main()
