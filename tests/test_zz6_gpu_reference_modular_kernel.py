"""SURVEY 8b, secondary boundary, executed by the REFERENCE's own drivers on the GPU.

1. The reference's `FusedMoEKernel` (vllm/model_executor/layers/fused_moe/modular_kernel.py:1588-1726) and its
   `FusedMoEKernelModularImpl._allocate_buffers / _prepare / _fused_experts / _finalize / apply` (:1096-1525) -- cut out of
   the reference tree byte for byte by oracle/make_ref_glue.py into oracle/_ref/modular_kernel_glue.py (build artifact) --
   are constructed from `lvllm_amd.modular.bind_vllm_prepare_finalize()` + `bind_vllm_base()` and own the call sequence:
   workspace shapes, `prepare` -> `apply` -> `finalize`, `expert_tokens_meta`, the output alias.  The results are compared
   with the CPU oracle for bf16, block-fp8 (W8A16 and W8A8) and wna16 experts, with and without an expert map, top-1 with
   the router weight applied to the input; single rank without a process group here, world-1 RCCL in
   tests/ep_rccl_one_rank.py ("reference FusedMoEKernel").
2. `MoERunner._apply_quant_method` (runner/moe_runner.py:577-664) with `RoutedExperts.should_use_gpu_prefill`
   (routed_experts.py:1344-1357), extracted the same way, drive the reference's `_cpu_decode` / `_gpu_prefill` /
   `_cpu_prefill` glue (oracle/_ref/routed_experts_glue.py) over this repo's `lk_moe`: capturing -> `_cpu_decode`,
   eager batch >= LVLLM_GPU_PREFILL_MIN_BATCH_SIZE -> `_gpu_prefill`, else `_cpu_prefill` -- each branch taken is
   recorded and its output checked against the oracle.
3. The reference's own test grids where the operator supports them: FUSED_MOE_MNK_FACTORS subset incl. E = 192 and
   m = 32 768 / 40 000 rows (tests/kernels/moe/test_moe.py:195-216), the k = 511 rows (hidden size % 8 != 0), and
   the block-fp8 grid N = 4608, K = 7168, E in {2, 8, 16}, top-k 6 (tests/kernels/moe/test_block_fp8.py:60-104).
Tolerances: bf16 / int4 atol 2e-2 of max|ref| (test_moe.py:233-234 scaled to the output range), block-fp8 0.035
(test_block_fp8.py:143-210).
"""
import importlib.util
import os
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch

import bench
from oracle import oracle as orc
from tests.helpers import make_routing, torch_to_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REFDIR = Path(__file__).resolve().parents[1] / "oracle" / "_ref"


def _load(name):
    path = REFDIR / f"{name}.py"
    if not path.exists():
        pytest.skip(f"oracle/_ref/{name}.py not built (python oracle/make_ref_glue.py needs /root/reference)")
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class _QC:
    """the attributes of a FusedMoEQuantConfig (fused_moe/config.py) the driver and LkmQuant.from_vllm read"""
    a1_scale = a2_scale = a1_gscale = a2_gscale = w1_zp = w2_zp = w1_bias = w2_bias = g1_alphas = g2_alphas = None
    quant_dtype = weight_quant_dtype = None
    per_act_token_quant = per_out_ch_quant = False
    use_fp8_w8a8 = use_fp8_w8a16 = use_int4_w4a16 = use_int8_w8a16 = use_mxfp4_w4a16 = False
    w1_scale = w2_scale = block_shape = None

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def _moe_config():
    return types.SimpleNamespace(moe_parallel_config=types.SimpleNamespace(dp_size=1, use_ep=False, ep_size=1, tp_size=1))


def _kernel(mk, E, H, qc, group=None, transport=None):
    from lvllm_amd import modular
    Experts, PF = modular.bind_vllm_base(mk), modular.bind_vllm_prepare_finalize(mk)
    assert not Experts.__abstractmethods__ and not PF.__abstractmethods__
    pf = PF(E, H, group=group, transport=transport, pool_tag="zz6")
    ex = Experts(_moe_config(), qc)
    k = mk.FusedMoEKernel(pf, ex)
    assert isinstance(k.impl, mk.FusedMoEKernelModularImpl) and not k.is_monolithic and k.output_is_reduced()
    return k


def _copy(out, inp):
    out.copy_(inp)


def _case(M, E, K, H, I, seed, dtype=torch.bfloat16, drop=0.0):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn((M, H), generator=g) / 10).to(dtype)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 10).to(dtype)
    w2 = (torch.randn((E, H, I), generator=g) / 10).to(dtype)
    tw, ids = make_routing(M, E, K, seed, drop=drop)
    return a, w13, w2, tw, ids


def _check(out, ref, tol):
    scale = float(np.abs(ref).max())
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=tol * scale, rtol=tol)


@pytest.mark.parametrize("fmt", ["bf16", "fp8_w8a16", "fp8_w8a8", "wna16"])
def test_reference_fused_moe_kernel_drives_the_bound_classes(fmt):
    mk = _load("modular_kernel_glue")
    M, E, K, H, I = 77, 8, 2, 512, 256
    a, w13, w2, tw, ids = _case(M, E, K, H, I, seed=5)
    twd, idd = torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV)
    if fmt == "bf16":
        qc, w1d, w2d = _QC(), w13.to(DEV), w2.to(DEV)
        d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        ref = orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(a), ids, tw)
        tol = 2e-2
    elif fmt.startswith("fp8"):
        q13, s13 = orc.quant_fp8_block(w13.float().numpy(), 128, 128)
        q2, s2 = orc.quant_fp8_block(w2.float().numpy(), 128, 128)
        a8 = fmt == "fp8_w8a8"
        qc = _QC(w1_scale=torch.from_numpy(s13).to(DEV), w2_scale=torch.from_numpy(s2).to(DEV), block_shape=[128, 128],
                 use_fp8_w8a8=a8, use_fp8_w8a16=not a8)
        w1d = torch.from_numpy(q13).to(DEV).view(torch.float8_e4m3fn)
        w2d = torch.from_numpy(q2).to(DEV).view(torch.float8_e4m3fn)
        d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_FP8, groupN=128, groupK=128, round_gemm1=a8, w8a8=a8)
        ref = orc.moe(d, q13, q2, torch_to_bits(a), ids, tw, s13=s13, s2=s2)
        tol = 0.035
    else:
        q13, s13 = orc.quant_int4(torch_to_bits(w13), orc.BF16, 128)
        q2, s2 = orc.quant_int4(torch_to_bits(w2), orc.BF16, 128)
        from tests.helpers import bits_to_torch
        qc = _QC(w1_scale=bits_to_torch(s13, orc.BF16).to(DEV), w2_scale=bits_to_torch(s2, orc.BF16).to(DEV),
                 block_shape=[0, 128], use_int4_w4a16=True)
        w1d, w2d = torch.from_numpy(q13).to(DEV), torch.from_numpy(q2).to(DEV)
        d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_INT4, groupN=1, groupK=128)
        ref = orc.moe(d, q13, q2, torch_to_bits(a), ids, tw, s13=s13, s2=s2)
        tol = 2e-2
    k = _kernel(mk, E, H, qc, transport=_copy)
    out = k.apply(a.to(DEV), w1d, w2d, twd, idd, mk.MoEActivation.SILU, E, None, False)
    assert out.shape == (M, H) and out.dtype == torch.bfloat16
    _check(out, ref, tol)
    # twice through the same kernel object (the engine is cached, the exchange pool is free again after finalize)
    out2 = k.apply(a.to(DEV), w1d, w2d, twd, idd, mk.MoEActivation.SILU, E, None, False)
    assert torch.equal(out2.view(torch.int16), out.view(torch.int16))


def test_reference_wn16_grid_zero_points_and_8bit_through_the_reference_driver():
    """test_fused_moe_wn16's has_zp x weight_bits x group grid (tests/kernels/moe/test_moe.py:565-693) through the
    reference's FusedMoEKernel over the bound classes: expected = torch_moe on quantize_weights' w_ref (golden, produced
    by the reference's own functions), the test's own tolerance atol 2e-2, rtol 0; and against the oracle on the
    oracle-dequantised weights.  4-bit runs the native packed format -- symmetric as uint4b8, with zero points in the engine's
    zero-point mode (LkmConfig.int4_mode = LKM_INT4_ZP: T((q - zp) * s) decoded in registers) -- 8-bit the expanded one."""
    mk = _load("modular_kernel_glue")
    from tests.helpers import bits_to_torch, load_golden
    seen = set()
    for i, c in load_golden("moe_wna16.npz"):
        m, n, k, e, topk, g, has_zp, bits = [int(v) for v in c["meta"]]
        qc = _QC(w1_scale=bits_to_torch(c["s1"], orc.BF16).to(DEV), w2_scale=bits_to_torch(c["s2"], orc.BF16).to(DEV),
                 block_shape=[0, g], use_int4_w4a16=bits == 4, use_int8_w8a16=bits == 8,
                 w1_zp=torch.from_numpy(c["z1"]).to(DEV) if has_zp else None,
                 w2_zp=torch.from_numpy(c["z2"]).to(DEV) if has_zp else None)
        kern = _kernel(mk, e, k, qc, transport=_copy)
        out = kern.apply(bits_to_torch(c["a"], orc.BF16).to(DEV), torch.from_numpy(c["q1"]).to(DEV), torch.from_numpy(c["q2"]).to(DEV),
                         torch.from_numpy(c["tw"]).to(DEV), torch.from_numpy(c["ids"]).to(DEV), mk.MoEActivation.SILU, e, None, False)
        ex = kern.impl.fused_experts
        desc = ex._engine.engine.describe()      # "wf=3" = the native packed 4-bit image (" zp=1": zero-point mode), "wf=0" = 16-bit (expanded)
        native_zp = bits == 4 and has_zp and k % 128 == 0 and n % 128 == 0
        assert ("wf=3 zp=1" if native_zp else ("wf=3 adt" if (bits == 4 and not has_zp) else "wf=0")) in desc, (desc, m, n, k, g)
        np.testing.assert_allclose(out.float().cpu().numpy(), orc.bits_to_f32(c["out"], orc.BF16), atol=2e-2, rtol=0, err_msg=f"case {i}")
        d1 = orc.dequant_wna16(c["q1"], c["s1"], c["z1"] if has_zp else None, bits, g, orc.BF16)
        d2 = orc.dequant_wna16(c["q2"], c["s2"], c["z2"] if has_zp else None, bits, g, orc.BF16)
        _check(out, orc.moe(orc.MoeDesc(E=e, H=k, I=n, act_dtype=orc.BF16), d1, d2, c["a"], c["ids"], c["tw"]), 2e-2)
        seen.add((has_zp, bits, g))
    assert len(seen) == 8


def test_reference_fused_moe_kernel_expert_map_and_router_weight_on_input():
    """an expert-parallel shard (global ids + expert_map, the other rank's experts skipped) and top-1 with
    apply_router_weight_on_input through the reference's driver"""
    mk = _load("modular_kernel_glue")
    from lvllm_amd.ops import determine_expert_map
    M, E, K, H, I = 45, 8, 2, 256, 128
    a, w13, w2, tw, ids = _case(M, E, K, H, I, seed=31)
    for ep_rank in (0, 1):
        n_loc, emap = determine_expert_map(2, ep_rank, E, "linear")
        lo = ep_rank * n_loc
        k = _kernel(mk, E, H, _QC(), transport=_copy)
        out = k.apply(a.to(DEV), w13[lo:lo + n_loc].contiguous().to(DEV), w2[lo:lo + n_loc].contiguous().to(DEV),
                      torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV), mk.MoEActivation.SILU, E, emap.to(DEV), False)
        lids = np.where((ids >= lo) & (ids < lo + n_loc), ids - lo, -1).astype(np.int32)
        d = orc.MoeDesc(E=n_loc, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        _check(out, orc.moe(d, torch_to_bits(w13[lo:lo + n_loc]), torch_to_bits(w2[lo:lo + n_loc]), torch_to_bits(a), lids, tw), 2e-2)
    M, E, K = 20, 4, 1
    a, w13, w2, tw, ids = _case(M, E, K, H, I, seed=33)
    k = _kernel(mk, E, H, _QC(), transport=_copy)
    out = k.apply(a.to(DEV), w13.to(DEV), w2.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV),
                  mk.MoEActivation.SILU, E, None, True)
    a_w = (a.float() * torch.from_numpy(tw)).to(torch.bfloat16)         # what prepare() does to the rows (topk = 1)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    _check(out, orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(a_w), ids, np.ones_like(tw)), 2e-2)
    # relu2 without gate through the reference's MoEActivation member
    g = torch.Generator().manual_seed(7)
    w13n = (torch.randn((E, I, H), generator=g) / 10).to(torch.bfloat16)
    k = _kernel(mk, E, H, _QC(), transport=_copy)
    out = k.apply(a.to(DEV), w13n.to(DEV), w2.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV),
                  mk.MoEActivation.RELU2_NO_MUL, E, None, False)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16, has_gate=False, activation=orc.ACT_RELU2)
    _check(out, orc.moe(d, torch_to_bits(w13n), torch_to_bits(w2), torch_to_bits(a), ids, tw), 2e-2)


# ----------------------------------------------------------------------------------------- MoERunner._apply_quant_method
def _install_forward_context_stubs(mode_none=True):
    """should_use_gpu_prefill imports these two names inside its body (routed_experts.py:1345-1350)"""
    import enum

    class CUDAGraphMode(enum.Enum):
        NONE = 0
        PIECEWISE = 1
        FULL = 2
    for name in ("vllm", "vllm.forward_context", "vllm.config"):
        sys.modules.setdefault(name, types.ModuleType(name))
    ctx = types.SimpleNamespace(cudagraph_runtime_mode=CUDAGraphMode.NONE if mode_none else CUDAGraphMode.FULL)
    fc = sys.modules["vllm.forward_context"]
    fc.ForwardContext, fc.get_forward_context, fc.is_forward_context_available = object, (lambda: ctx), (lambda: True)
    sys.modules["vllm.config"].CUDAGraphMode = CUDAGraphMode
    return ctx, CUDAGraphMode


def test_reference_apply_quant_method_takes_the_three_lk_moe_branches(monkeypatch):
    from tests.test_zz4_gpu_reference_glue import _load_glue, _stand_in
    from lvllm_amd import ops
    glue, run = _load_glue(), _load("moe_runner_glue")
    monkeypatch.setenv("LVLLM_GPU_PREFILL_MIN_BATCH_SIZE", "200")
    ctx, Mode = _install_forward_context_stubs()
    E, K, H, I = 6, 2, 512, 256
    g = torch.Generator().manual_seed(1)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 10).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g) / 10).to(torch.bfloat16)
    a13, a2 = torch_to_bits(w13), torch_to_bits(w2)
    s = _stand_in(glue, E, K, H, I, torch.bfloat16, w13_weight=w13, w2_weight=w2)
    s._process_bf6_fp16()
    s._initialize_cuda_graph_buffers()
    s.clean_weights_after_loading()
    taken = []
    for name in ("_cpu_decode", "_cpu_prefill", "_gpu_prefill"):
        def wrap(fn, name=name):
            def f(*a, **k):
                taken.append(name)
                return fn(*a, **k)
            return f
        setattr(s, name, wrap(getattr(s, name)))
    s.is_gpu_prefill_layer = True
    s.quant_method = types.SimpleNamespace(is_monolithic=False)
    s.should_use_gpu_prefill = types.MethodType(run.RoutedExpertsDispatch.should_use_gpu_prefill, s)
    # the names should_use_gpu_prefill resolves at module level in routed_experts.py
    run.RoutedExpertsDispatch.should_use_gpu_prefill.__globals__["get_gpu_prefill_min_batch_size"] = run.get_gpu_prefill_min_batch_size

    class Router:
        def select_experts(self, hidden_states, router_logits, topk_indices_dtype=None, input_ids=None):
            w, i = ops.topk_softmax(router_logits, K, True)
            return w, i
    runner = run.MoERunner.__new__(run.MoERunner)
    runner.router, runner.routed_experts, runner._shared_experts = Router(), s, None
    runner._quant_method = types.SimpleNamespace(topk_indices_dtype=torch.int32)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)

    def check(out, x, logits, what):
        wr, ir = orc.topk_softmax(logits.cpu().numpy(), K, renormalize=True)
        ref = orc.moe(d, a13, a2, torch_to_bits(x.cpu()), ir, wr)
        _check(out, ref, 2e-2)
        assert taken[-1] == what, taken

    gen = torch.Generator().manual_seed(3)
    # eager, small batch -> _cpu_prefill (host pointers, blocking)
    x = (torch.randn((24, H), generator=gen) / 8).to(torch.bfloat16).to(DEV)
    lg = torch.randn((24, E), generator=gen).to(DEV)
    sh, out = runner._apply_quant_method(x, lg, None)
    assert sh is None
    check(out, x, lg, "_cpu_prefill")
    # eager, batch >= LVLLM_GPU_PREFILL_MIN_BATCH_SIZE -> _gpu_prefill
    x = (torch.randn((300, H), generator=gen) / 8).to(torch.bfloat16).to(DEV)
    lg = torch.randn((300, E), generator=gen).to(DEV)
    _, out = runner._apply_quant_method(x, lg, None)
    check(out, x, lg, "_gpu_prefill")
    # ... but not when the step runs under a cudagraph runtime mode (routed_experts.py:1351-1353)
    ctx.cudagraph_runtime_mode = Mode.FULL
    _, out = runner._apply_quant_method(x[:250], lg[:250], None)
    check(out, x[:250], lg[:250], "_cpu_prefill")
    ctx.cudagraph_runtime_mode = Mode.NONE
    # capturing -> _cpu_decode (device pointers, graph-capturable), replayed
    M = 16
    xs = torch.zeros((M, H), dtype=torch.bfloat16, device=DEV)
    ls = torch.zeros((M, E), dtype=torch.float32, device=DEV)
    x = (torch.randn((M, H), generator=gen) / 8).to(torch.bfloat16).to(DEV)
    lg = torch.randn((M, E), generator=gen).to(DEV)
    xs.copy_(x), ls.copy_(lg)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        s._cpu_decode(xs, *ops.topk_softmax(ls, K, True))      # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        _, out_static = runner._apply_quant_method(xs, ls, None)
    assert taken[-1] == "_cpu_decode"
    gr.replay()
    torch.cuda.synchronize()
    check(out_static, x, lg, "_cpu_decode")


# ----------------------------------------------------------------------------------------- the reference's test grids
def _bf16_grid_case(m, n, k, e, topk, seed=7, rows=None):
    """test_moe.py:292-413 (torch_experts recipe: a, w1, w2 = randn / 10, softmax top-k of randn scores); n = 2 x the
    intermediate size there.  rows: check this seeded subset of the m output rows against the oracle."""
    dev = torch.device(DEV)
    g = torch.Generator(device=dev).manual_seed(seed)
    a = (torch.randn((m, k), generator=g, device=dev) / 10).to(torch.bfloat16)
    w1 = (torch.randn((e, 2 * n, k), generator=g, device=dev) / 10).to(torch.bfloat16)
    w2 = (torch.randn((e, k, n), generator=g, device=dev) / 10).to(torch.bfloat16)
    score = torch.randn((m, e), generator=g, device=dev)
    return a, w1, w2, score


@pytest.mark.parametrize("m,n,k,e,topk", [
    (1, 128, 128, 8, 2), (33, 128, 128, 64, 6), (33, 1024, 2048, 8, 2), (222, 128, 2048, 64, 6), (222, 2048, 128, 8, 6),
    (2, 256, 128, 192, 6),                    # E = 192 (test_moe.py:203-216: the large-expert-count rows)
    (32768, 128, 1024, 8, 2), (40000, 128, 1024, 8, 6),      # the big-m rows (chunked by the operator, :304-333)
])
def test_reference_fused_moe_mnk_grid(m, n, k, e, topk):
    from lvllm_amd import ops
    a, w1, w2, score = _bf16_grid_case(m, n, k, e, topk)
    eng = ops.RoutedExpertsEngine(w1, w2, top_k=topk, act_dtype=torch.bfloat16, max_batch_size=4096)
    tw, ids = ops.topk_softmax(score, topk, False)              # fused_topk(..., renormalize=False), test_moe.py:352
    out = eng.prefill(a, tw, ids)
    assert out.shape == (m, k)
    rows = np.arange(m) if m <= 512 else np.random.default_rng(m).choice(m, 256, replace=False)
    d = orc.MoeDesc(E=e, H=k, I=n, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    ref = orc.moe(d, torch_to_bits(w1.cpu()), torch_to_bits(w2.cpu()), torch_to_bits(a[rows].cpu()),
                  ids[rows].cpu().numpy(), tw[rows].cpu().numpy())
    _check(out[rows], ref, 2e-2)


@pytest.mark.parametrize("m,n,k,e,topk", [
    (2, 2048, 511, 8, 2), (2, 2048, 511, 64, 6),             # FUSED_MOE_MNK_FACTORS_SMALL_M (test_moe.py:203-208)
    (33, 128, 511, 8, 2), (222, 1024, 511, 8, 6),
    (32768, 2048, 511, 8, 2),                                # FUSED_MOE_MNK_FACTORS (test_moe.py:195-201): the big-m k = 511 row
])
def test_reference_fused_moe_mnk_grid_k511(m, n, k, e, topk):
    """k = 511 of the reference's grid (hidden size not a multiple of 8; refused until round 4): lkm_create pads K with
    zeros in the image, the token / output rows pass through aligned scratch rows (lkm_api.hip run_device).  n is the
    intermediate size of the reference's torch_moe recipe (w1 [e, 2n, k])."""
    from lvllm_amd import ops
    if m > 4096:
        n = 256            # (the oracle's CPU time for 256 sampled rows of 8 experts x 2048 x 511 is fine, the weights' are not the point)
    a, w1, w2, score = _bf16_grid_case(m, n, k, e, topk)
    eng = ops.RoutedExpertsEngine(w1, w2, top_k=topk, act_dtype=torch.bfloat16, max_batch_size=4096)
    tw, ids = ops.topk_softmax(score, topk, False)
    out = eng.prefill(a, tw, ids)
    assert out.shape == (m, k) and out.is_contiguous()
    rows = np.arange(m) if m <= 512 else np.random.default_rng(m).choice(m, 256, replace=False)
    d = orc.MoeDesc(E=e, H=k, I=n, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    ref = orc.moe(d, torch_to_bits(w1.cpu()), torch_to_bits(w2.cpu()), torch_to_bits(a[rows].cpu()),
                  ids[rows].cpu().numpy(), tw[rows].cpu().numpy())
    _check(out[rows], ref, 2e-2)
    if m <= 512:
        # the three entry points agree on the odd shape: cpu_decode (fp32), the fused router + scatter step, cpu_prefill
        dec = eng.decode(a, tw, ids)
        np.testing.assert_allclose(dec.cpu().numpy(), ref, atol=2e-3 * float(np.abs(ref).max()), rtol=1e-2)
        fused, fw, fi = eng.forward_logits(a, score, topk, False)
        assert torch.equal(fi, ids) and torch.equal(fw, tw) and torch.equal(fused, dec)
        host = eng.prefill_host(a.cpu(), tw.cpu(), ids.cpu())
        assert torch.equal(host, dec.cpu())


def test_quantised_formats_still_refuse_odd_hidden_sizes():
    """H % 8 != 0 is accepted for unquantised weights only (the quantised formats' groups need whole 32 / 128-k blocks)"""
    from lvllm_amd import _clib, ops
    q13 = torch.zeros((4, 256, 511), dtype=torch.uint8, device=DEV)
    q2 = torch.zeros((4, 511, 128), dtype=torch.uint8, device=DEV)
    s13 = torch.ones((4, 2, 4), dtype=torch.float32, device=DEV)
    s2 = torch.ones((4, 4, 1), dtype=torch.float32, device=DEV)
    with pytest.raises(RuntimeError, match="multiple of 8"):
        ops.RoutedExpertsEngine(q13, q2, top_k=2, act_dtype=torch.bfloat16, fmt="fp8", w13_scale=s13, w2_scale=s2,
                                group_n=128, group_k=128, fp8_mode=_clib.FP8_W8A16)


@pytest.mark.parametrize("M,E", [(1, 2), (83, 8), (2048, 16)])
def test_reference_block_fp8_grid(M, E):
    """MNK_FACTORS of tests/kernels/moe/test_block_fp8.py:60-104: N = 4608 (w13 rows = 2 x 2304), K = 7168, top-k 6,
    128 x 128 blocks, W8A8 (the in-tree block-fp8 semantics); tolerance 0.035 (:143-210)"""
    from lvllm_amd import _clib, ops
    N, K, topk = 4608, 7168, 6
    topk = min(topk, E)
    dev = torch.device(DEV)
    g = torch.Generator(device=dev).manual_seed(M + E)
    a = (torch.randn((M, K), generator=g, device=dev) / 10).to(torch.bfloat16)
    w13 = (torch.randn((E, N, K), generator=g, device=dev) / 10).to(torch.bfloat16)
    w2 = (torch.randn((E, K, N // 2), generator=g, device=dev) / 10).to(torch.bfloat16)
    q13, s13 = bench.quantize_fp8_block(w13)
    q2, s2 = bench.quantize_fp8_block(w2)
    del w13, w2
    eng = ops.RoutedExpertsEngine(q13, q2, top_k=topk, act_dtype=torch.bfloat16, fmt="fp8", w13_scale=s13, w2_scale=s2,
                                  group_n=128, group_k=128, fp8_mode=_clib.FP8_W8A8, max_batch_size=4096)
    score = torch.randn((M, E), generator=g, device=dev)
    tw, ids = ops.topk_softmax(score, topk, False)
    out = eng.prefill(a, tw, ids)
    rows = np.arange(M) if M <= 128 else np.random.default_rng(M).choice(M, 96, replace=False)
    d = orc.MoeDesc(E=E, H=K, I=N // 2, act_dtype=orc.BF16, wfmt=orc.W_FP8, groupN=128, groupK=128, round_gemm1=True, w8a8=True)
    ref = orc.moe(d, q13.cpu().numpy(), q2.cpu().numpy(), torch_to_bits(a[rows].cpu()), ids[rows].cpu().numpy(),
                  tw[rows].cpu().numpy(), s13=s13.cpu().numpy(), s2=s2.cpu().numpy())
    _check(out[rows], ref, 0.035)
