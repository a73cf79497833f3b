"""The reference's OWN lk_moe glue, run unmodified against this repo's `lk_moe` module on the GPU.

oracle/make_ref_glue.py extracts (with `ast`, byte for byte) the methods of the reference's RoutedExperts that hand the
weights to lk_moe and drive its three entry points -- _get_quant_params, _process_{bf6_fp16, wna16, fp8, mxfp4, nvfp4},
_initialize_cuda_graph_buffers, _cpu_decode, _cpu_prefill, _gpu_prefill, clean_weights_after_loading,
global_to_local_expert_ids (vllm/model_executor/layers/fused_moe/routed_experts.py:1332-1342, 1420-1899) -- into
oracle/_ref/routed_experts_glue.py, a build artifact (git-ignored; it travels to the GPU box like oracle/_ref/*.so).  Here
they are bound to a stand-in object that holds CPU weight tensors in the layouts the reference's create_weights produce:
    unquantized   w13_weight [E, 2I, H], w2_weight [E, H, I]                        (unquantized_fused_moe_method.py:98-135)
    wna16         w13_weight_packed int32 [E, H/8, 2I], w13_weight_scale [E, H/g, 2I]   ("Marlin" shapes,
                  compressed_tensors_moe_wna16.py:131-230, 239-420: transposed, the glue transposes them back)
    fp8 block     w13_weight float8_e4m3fn [E, 2I, H], w13_weight_scale_inv fp32 [E, 2I/128, H/128]   (fp8.py:524-668)
    mxfp4 / nvfp4 w13_weight_packed uint8 [E, 2I, H/2], w13_weight_scale uint8 [E, 2I, H/32 | H/16] (+ global scales)
What is checked per MOE_* family: the constructor call as the glue makes it (host pointers, transposed-back views, the
class-static output buffer), the weight tensors deleted right after construction (clean_weights_after_loading: the engine
must own its copy), _cpu_decode inside a captured graph (the reference always captures it, moe_runner.py:609-614) and
replayed on new inputs, _cpu_prefill (host pointers) and _gpu_prefill eagerly -- all against the oracle."""
import enum
import importlib.util
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
GLUE = Path(__file__).resolve().parents[1] / "oracle" / "_ref" / "routed_experts_glue.py"


def _load_glue():
    if not GLUE.exists():
        pytest.skip("oracle/_ref/routed_experts_glue.py not built (python oracle/make_ref_glue.py needs /root/reference)")
    # the one import inside the extracted methods that this image cannot satisfy: an enum of the WNA16 backends
    name = "vllm.model_executor.layers.fused_moe.oracle.int_wna16"
    if name not in sys.modules:
        parts = name.split(".")
        for i in range(1, len(parts) + 1):
            sys.modules.setdefault(".".join(parts[:i]), types.ModuleType(".".join(parts[:i])))

        class WNA16MoEBackend(enum.Enum):
            MARLIN = 0
            FLASHINFER_TRTLLM = 1
        sys.modules[name].WNA16MoEBackend = WNA16MoEBackend
    spec = importlib.util.spec_from_file_location("routed_experts_glue", GLUE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _QuantMethod:          # what _process_wna16 reads off self.quant_method
    def __init__(self, group_size):
        self.group_size, self.num_bits, self.packed_factor = group_size, 4, 8


def _stand_in(glue, E, K, H, I, dtype, **weights):
    for a in ("cuda_graphs", "output_gpu"):      # class statics of the previous family (other hidden size / device)
        if hasattr(glue.RoutedExperts, a):
            delattr(glue.RoutedExperts, a)

    class Layer(glue.RoutedExperts):
        def _ensure_moe_quant_config_init(self):
            pass
    s = Layer.__new__(Layer)
    s.is_gpu_resident_layer = False
    s.use_ep, s.tp_size, s.tp_rank, s.ep_size, s.ep_rank = False, 1, 0, 1, 0
    s.has_gate_proj, s.local_num_experts, s.top_k = True, E, K
    s.hidden_size, s.intermediate_size_per_partition = H, I
    s.max_num_batched_tokens, s.max_num_seqs, s.max_num_group_batch_size = 2048, 64, 2048 + 128
    s.activation_type, s.swiglu_alpha, s.swiglu_limit = 0, None, None
    s.use_gpu_prefill, s.params_dtype, s.check_nan_in_output = True, dtype, True
    for k, v in weights.items():
        setattr(s, k, v)
    return s


def _drive(s, E, K, H, I, ref_fn, atol, rtol):
    """the three entry points as the reference calls them; ref_fn(x_bits, ids, tw) -> fp32 oracle output"""
    s._initialize_cuda_graph_buffers()
    s.clean_weights_after_loading()              # the caller's tensors are gone: the engine owns its copy
    for name in ("w13_weight", "w2_weight", "w13_weight_packed", "w13_weight_scale", "w13_weight_scale_inv"):
        assert not hasattr(s, name)
    dt = s.params_dtype
    gen = torch.Generator().manual_seed(3)

    def inputs(M, seed):
        x = (torch.randn((M, H), generator=gen) / 8).to(dt)
        tw, ids = make_routing(M, E, K, seed=seed, drop=0.1)
        return x, torch.from_numpy(tw), torch.from_numpy(ids)

    def check(out, x, tw, ids, what):
        ref = ref_fn(torch_to_bits(x), ids.numpy(), tw.numpy())
        scale = float(np.abs(ref).max())
        np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=atol * scale, rtol=rtol, err_msg=what)

    # ---- _cpu_decode, captured (static input buffers, the class-static fp32 output buffer) and replayed
    M = 24
    xs = torch.zeros((M, H), dtype=dt, device=DEV)
    tws = torch.zeros((M, K), dtype=torch.float32, device=DEV)
    idss = torch.zeros((M, K), dtype=torch.int32, device=DEV)
    x, tw, ids = inputs(M, 1)
    xs.copy_(x), tws.copy_(tw), idss.copy_(ids)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        s._cpu_decode(xs, tws, idss)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out_static = s._cpu_decode(xs, tws, idss)
    g.replay()
    torch.cuda.synchronize()
    check(out_static, x, tw, ids, "_cpu_decode (captured)")
    x, tw, ids = inputs(M, 2)
    xs.copy_(x), tws.copy_(tw), idss.copy_(ids)
    g.replay()
    torch.cuda.synchronize()
    check(out_static, x, tw, ids, "_cpu_decode (replayed on new inputs)")
    assert out_static.dtype == dt and type(s).output_gpu.dtype == torch.float32
    # ---- _cpu_prefill (host pointers, blocking) and _gpu_prefill (device pointers, activation dtype)
    x, tw, ids = inputs(300, 3)
    check(s._cpu_prefill(x.to(DEV), tw.to(DEV), ids.to(DEV)), x, tw, ids, "_cpu_prefill")
    check(s._gpu_prefill(x.to(DEV), tw.to(DEV), ids.to(DEV)), x, tw, ids, "_gpu_prefill")


def _masters(E, H, I, seed, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 10).to(dtype)
    w2 = (torch.randn((E, H, I), generator=g) / 10).to(dtype)
    return w13, w2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_reference_glue_bf6_fp16(dtype):
    """MOE_BF16 / MOE_FP16 through RoutedExperts._process_bf6_fp16 (routed_experts.py:1618-1668)"""
    glue = _load_glue()
    E, K, H, I = 6, 2, 512, 256
    w13, w2 = _masters(E, H, I, 1, dtype)
    odt = orc.BF16 if dtype == torch.bfloat16 else orc.F16
    a13, a2 = torch_to_bits(w13), torch_to_bits(w2)
    s = _stand_in(glue, E, K, H, I, dtype, w13_weight=w13, w2_weight=w2)
    s._process_bf6_fp16()
    assert type(s.lk_moe).__name__ == ("MOE_BF16" if dtype == torch.bfloat16 else "MOE_FP16")
    del w13, w2
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=odt, wfmt=orc.W_BF16 if dtype == torch.bfloat16 else orc.W_F16)
    _drive(s, E, K, H, I, lambda x, ids, tw: orc.moe(d, a13, a2, x, ids, tw), 3e-3, 1.5e-2)


def test_reference_glue_layer_that_does_not_fit_takes_the_spill_tier(monkeypatch):
    """the spill tier behind the SAME three calls.  With an HBM budget smaller than the layer
    (LKM_HBM_CAP_BYTES; on hardware: lkm_create -> LKM_E_NOMEM) the reference's own glue -- _process_bf6_fp16 with host
    pointers, then _cpu_decode / _cpu_prefill / _gpu_prefill -- ends in lvllm_amd.spill.HostResidentExperts (pinned host
    images, 2 x LVLLM_GPU_PREFETCH_WINDOW device slots): routed_experts.py:1344-1357, 1884-1899; vllm/envs.py:265,1942-1943.
    Eager only, like the reference's gpu_prefill (HostResidentExperts.forward refuses a capturing stream)."""
    glue = _load_glue()
    from lvllm_amd.residency import expert_layer_bytes
    E, K, H, I = 10, 2, 256, 128
    dtype = torch.bfloat16
    w13, w2 = _masters(E, H, I, 7, dtype)
    a13, a2 = torch_to_bits(w13), torch_to_bits(w2)
    monkeypatch.setenv("LKM_HBM_CAP_BYTES", str(expert_layer_bytes(E, H, I, "bf16") // 2))     # half the layer fits
    monkeypatch.setenv("LVLLM_GPU_PREFETCH_WINDOW", "2")
    s = _stand_in(glue, E, K, H, I, dtype, w13_weight=w13, w2_weight=w2)
    s._process_bf6_fp16()
    eng = s.lk_moe
    assert type(eng).__name__ == "MOE_BF16" and type(eng._spill).__name__ == "HostResidentExperts"
    assert eng._spill.slots == 4 and len(eng._spill.images) == E and "spill tier" in eng.describe()
    assert eng.weight_bytes() < expert_layer_bytes(E, H, I, "bf16")           # HBM holds the window, not the layer
    s._initialize_cuda_graph_buffers()
    s.clean_weights_after_loading()
    del w13, w2
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    gen = torch.Generator().manual_seed(5)
    for M, seed in ((24, 1), (300, 2)):
        x = (torch.randn((M, H), generator=gen) / 8).to(dtype)
        tw, ids = make_routing(M, E, K, seed=seed, drop=0.1)
        ref = orc.moe(d, a13, a2, torch_to_bits(x), ids, tw)
        scale = float(np.abs(ref).max())
        xd, twd, idd = x.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV)
        for name in (("_cpu_decode",) if M <= s.max_num_seqs else ()) + ("_cpu_prefill", "_gpu_prefill"):
            out = getattr(s, name)(xd, twd, idd)
            torch.cuda.synchronize()
            np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=3e-3 * scale, rtol=1.5e-2, err_msg=f"{name} M={M}")
    eng.close()


def _drive_spill(s, E, K, H, I, ref_fn, atol, rtol, cls):
    """the three entry points on a layer that took the spill tier (eager: no capture), against the oracle"""
    eng = s.lk_moe
    assert type(eng).__name__ == cls and type(eng._spill).__name__ == "HostResidentExperts", (type(eng).__name__, eng.describe())
    assert len(eng._spill.images) == E and eng._spill.slots < E and "spill tier" in eng.describe()
    s._initialize_cuda_graph_buffers()
    s.clean_weights_after_loading()
    dt = s.params_dtype
    gen = torch.Generator().manual_seed(11)
    for M, seed in ((24, 1), (300, 2)):
        x = (torch.randn((M, H), generator=gen) / 8).to(dt)
        tw, ids = make_routing(M, E, K, seed=seed, drop=0.1)
        ref = ref_fn(torch_to_bits(x), ids, tw)
        scale = float(np.abs(ref).max())
        xd, twd, idd = x.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV)
        for name in (("_cpu_decode",) if M <= s.max_num_seqs else ()) + ("_cpu_prefill", "_gpu_prefill"):
            out = getattr(s, name)(xd, twd, idd)
            torch.cuda.synchronize()
            np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=atol * scale, rtol=rtol, err_msg=f"{name} M={M}")
    eng.close()


@pytest.mark.parametrize("kind", ["wna16", "fp8_block", "mxfp4", "nvfp4"])
def test_reference_glue_quantised_layers_take_the_spill_tier(monkeypatch, kind):
    """the spill tier behind lk_moe.MOE_WNA16 / MOE_FP8 / MOE_MXFP4 (the formats the reference's GPU-prefill tier exists for:
    a quantised layer too big for HBM): the same glue calls as the resident tests above with half the layer as the HBM budget;
    the per-format host views of lk_moe_api._build_spill (shapes of SURVEY 8 a5) against the oracle"""
    glue = _load_glue()
    from lvllm_amd.residency import expert_layer_bytes
    E, K, H, I = 6, 2, 512, 256
    monkeypatch.setenv("LVLLM_GPU_PREFETCH_WINDOW", "2")
    if kind == "wna16":
        g = 64
        w13, w2 = _masters(E, H, I, 12)
        q13, s13 = bench.quantize_int4(w13.to(DEV), g)
        q2, s2 = bench.quantize_int4(w2.to(DEV), g)
        q13, s13, q2, s2 = q13.cpu(), s13.cpu(), q2.cpu(), s2.cpu()
        pk = lambda q: q.contiguous().view(torch.int32).transpose(1, 2).contiguous()       # noqa: E731
        monkeypatch.setenv("LKM_HBM_CAP_BYTES", str(expert_layer_bytes(E, H, I, "int4", group_k=g) // 2))
        s = _stand_in(glue, E, K, H, I, torch.bfloat16, w13_weight_packed=pk(q13), w2_weight_packed=pk(q2),
                      w13_weight_scale=s13.transpose(1, 2).contiguous(), w2_weight_scale=s2.transpose(1, 2).contiguous(),
                      quant_method=_QuantMethod(g))
        s._process_wna16("group")
        d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_INT4, groupN=1, groupK=g)
        a = (q13.numpy(), q2.numpy(), torch_to_bits(s13), torch_to_bits(s2))
        _drive_spill(s, E, K, H, I, lambda x, ids, tw: orc.moe(d, a[0], a[1], x, ids, tw, s13=a[2], s2=a[3]), 3e-3, 1.5e-2, "MOE_WNA16")
    elif kind == "fp8_block":
        w13, w2 = _masters(E, H, I, 13)
        q13, s13 = bench.quantize_fp8_block(w13.to(DEV))
        q2, s2 = bench.quantize_fp8_block(w2.to(DEV))
        q13, s13, q2, s2 = q13.cpu(), s13.cpu(), q2.cpu(), s2.cpu()
        monkeypatch.setenv("LKM_HBM_CAP_BYTES", str(expert_layer_bytes(E, H, I, "fp8") // 2))
        s = _stand_in(glue, E, K, H, I, torch.bfloat16, w13_weight=q13.view(torch.float8_e4m3fn),
                      w2_weight=q2.view(torch.float8_e4m3fn), w13_weight_scale_inv=s13, w2_weight_scale_inv=s2)
        s._process_fp8(True)
        d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_FP8, groupN=128, groupK=128)
        a = (q13.numpy(), q2.numpy(), s13.numpy(), s2.numpy())
        _drive_spill(s, E, K, H, I, lambda x, ids, tw: orc.moe(d, a[0], a[1], x, ids, tw, s13=a[2], s2=a[3]), 3e-3, 1.5e-2, "MOE_FP8")
    elif kind == "nvfp4":
        w13, w2 = _masters(E, H, I, 15)
        q13, s13, m13 = bench.quantize_nvfp4(w13.to(DEV))
        q2, s2, m2 = bench.quantize_nvfp4(w2.to(DEV))
        q13, s13, m13, q2, s2, m2 = q13.cpu(), s13.cpu(), m13.cpu(), q2.cpu(), s2.cpu(), m2.cpu()
        monkeypatch.setenv("LKM_HBM_CAP_BYTES", str(expert_layer_bytes(E, H, I, "nvfp4") // 2))
        s = _stand_in(glue, E, K, H, I, torch.bfloat16, w13_weight_packed=q13, w2_weight_packed=q2,
                      w13_weight_scale=s13.view(torch.float8_e4m3fn), w2_weight_scale=s2.view(torch.float8_e4m3fn),
                      w13_weight_global_scale=1.0 / m13, w2_weight_global_scale=1.0 / m2)
        s._process_nvfp4(need_reciprocal_global_scale=True)
        d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_NVFP4, groupN=1, groupK=16)
        g13, g2 = (1.0 / (1.0 / m13)).numpy(), (1.0 / (1.0 / m2)).numpy()
        a = (q13.numpy(), q2.numpy(), s13.numpy(), s2.numpy())
        _drive_spill(s, E, K, H, I, lambda x, ids, tw: orc.moe(d, a[0], a[1], x, ids, tw, s13=a[2], s2=a[3], gs13=g13, gs2=g2),
                     3e-3, 1.5e-2, "MOE_NVFP4")
    else:
        w13, w2 = _masters(E, H, I, 14)
        q13, s13 = bench.quantize_mxfp4(w13.to(DEV))
        q2, s2 = bench.quantize_mxfp4(w2.to(DEV))
        q13, s13, q2, s2 = q13.cpu(), s13.cpu(), q2.cpu(), s2.cpu()
        monkeypatch.setenv("LKM_HBM_CAP_BYTES", str(expert_layer_bytes(E, H, I, "mxfp4") // 2))
        s = _stand_in(glue, E, K, H, I, torch.bfloat16, w13_weight_packed=q13, w2_weight_packed=q2, w13_weight_scale=s13,
                      w2_weight_scale=s2)
        s._process_mxfp4()
        d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_MXFP4, groupN=1, groupK=32)
        a = (q13.numpy(), q2.numpy(), s13.numpy(), s2.numpy())
        _drive_spill(s, E, K, H, I, lambda x, ids, tw: orc.moe(d, a[0], a[1], x, ids, tw, s13=a[2], s2=a[3]), 3e-3, 1.5e-2, "MOE_MXFP4")


def test_reference_glue_wna16():
    """MOE_WNA16 through RoutedExperts._process_wna16 (:1456-1533): the checkpoint's transposed int32 / scale tensors,
    `.cpu().transpose(1, 2).contiguous().view(torch.uint8)` pointers, group size from _get_quant_params"""
    glue = _load_glue()
    E, K, H, I, g = 4, 2, 512, 256, 64
    w13, w2 = _masters(E, H, I, 2)
    q13, s13 = bench.quantize_int4(w13.to(DEV), g)
    q2, s2 = bench.quantize_int4(w2.to(DEV), g)
    q13, s13, q2, s2 = q13.cpu(), s13.cpu(), q2.cpu(), s2.cpu()
    # create_weights' shapes: [E, K/8, N] int32 (eight nibbles of consecutive k per word), scales [E, K/g, N]
    pk = lambda q: q.contiguous().view(torch.int32).transpose(1, 2).contiguous()       # noqa: E731
    s = _stand_in(glue, E, K, H, I, torch.bfloat16, w13_weight_packed=pk(q13), w2_weight_packed=pk(q2),
                  w13_weight_scale=s13.transpose(1, 2).contiguous(), w2_weight_scale=s2.transpose(1, 2).contiguous(),
                  quant_method=_QuantMethod(g))
    assert s.w13_weight_packed.shape == (E, H // 8, 2 * I) and s.w13_weight_scale.shape == (E, H // g, 2 * I)
    s._process_wna16("group")
    assert type(s.lk_moe).__name__ == "MOE_WNA16" and (s.lk_moe_config.groupN, s.lk_moe_config.groupK) == (1, g)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_INT4, groupN=1, groupK=g)
    a = (q13.numpy(), q2.numpy(), torch_to_bits(s13), torch_to_bits(s2))
    _drive(s, E, K, H, I, lambda x, ids, tw: orc.moe(d, a[0], a[1], x, ids, tw, s13=a[2], s2=a[3]), 3e-3, 1.5e-2)


def test_reference_glue_fp8_block():
    """MOE_FP8 through RoutedExperts._process_fp8(block_quant=True) (:1549-1616): float8_e4m3fn weights and the
    128 x 128 `weight_scale_inv` tensors; lk_moe's fp8 semantics are W8A16 (bf16 activations)"""
    glue = _load_glue()
    E, K, H, I = 4, 2, 512, 256
    w13, w2 = _masters(E, H, I, 3)
    q13, s13 = bench.quantize_fp8_block(w13.to(DEV))
    q2, s2 = bench.quantize_fp8_block(w2.to(DEV))
    q13, s13, q2, s2 = q13.cpu(), s13.cpu(), q2.cpu(), s2.cpu()
    s = _stand_in(glue, E, K, H, I, torch.bfloat16, w13_weight=q13.view(torch.float8_e4m3fn),
                  w2_weight=q2.view(torch.float8_e4m3fn), w13_weight_scale_inv=s13, w2_weight_scale_inv=s2)
    s._process_fp8(True)
    assert type(s.lk_moe).__name__ == "MOE_FP8" and (s.lk_moe_config.groupN, s.lk_moe_config.groupK) == (128, 128)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_FP8, groupN=128, groupK=128)
    a = (q13.numpy(), q2.numpy(), s13.numpy(), s2.numpy())
    _drive(s, E, K, H, I, lambda x, ids, tw: orc.moe(d, a[0], a[1], x, ids, tw, s13=a[2], s2=a[3]), 3e-3, 1.5e-2)


def test_reference_glue_fp8_per_channel():
    """MOE_FP8 through RoutedExperts._process_fp8(block_quant=False) -- what _do_process_weights_after_loading calls for
    CompressedTensorsW8A8Fp8MoEMethod (routed_experts.py:1381-1383): per-output-channel scales `w13_weight_scale`
    [E, 2I, 1], `w2_weight_scale` [E, H, 1]; _get_quant_params (:1440-1453) then hands over groupN = 1 and
    groupK = max(hidden, intermediate): ONE scale group per weight row of each GEMM (include/lkm.h: LkmConfig.groupK)"""
    glue = _load_glue()
    E, K, H, I = 4, 2, 512, 256
    w13, w2 = _masters(E, H, I, 6)

    def per_channel(w):
        s = (w.float().abs().amax(dim=-1, keepdim=True).clamp(min=1e-4) / 448.0)
        return (w.float() / s).to(torch.float8_e4m3fn), s.contiguous()
    q13, s13 = per_channel(w13)
    q2, s2 = per_channel(w2)
    assert s13.shape == (E, 2 * I, 1) and s2.shape == (E, H, 1)
    s = _stand_in(glue, E, K, H, I, torch.bfloat16, w13_weight=q13, w2_weight=q2, w13_weight_scale=s13, w2_weight_scale=s2)
    s._process_fp8(False)
    assert type(s.lk_moe).__name__ == "MOE_FP8" and (s.lk_moe_config.groupN, s.lk_moe_config.groupK) == (1, max(H, I))
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_FP8, groupN=1, groupK=max(H, I))
    a = (q13.view(torch.uint8).numpy(), q2.view(torch.uint8).numpy(), s13.numpy(), s2.numpy())
    # the oracle dequantises fp8 * scale[n] (per channel) and runs the bf16 path on it (SURVEY 8c "fp8 W8A16")
    _drive(s, E, K, H, I, lambda x, ids, tw: orc.moe(d, a[0], a[1], x, ids, tw, s13=a[2], s2=a[3]), 3e-3, 1.5e-2)


def test_reference_glue_wna16_channel():
    """MOE_WNA16 through RoutedExperts._process_wna16("channel"): channel-wise scales, checkpoint shapes
    `w13_weight_scale` [E, 1, 2I] / `w2_weight_scale` [E, 1, H] (transposed back to [E, N, 1] by the glue);
    groupK = max(hidden, intermediate) again.  hidden > intermediate here, the other order in the fp8 case above."""
    glue = _load_glue()
    E, K, H, I = 4, 2, 512, 256
    w13, w2 = _masters(E, H, I, 7)
    q13, s13 = bench.quantize_int4(w13.to(DEV), H)          # one group per row: g = K of the GEMM
    q2, s2 = bench.quantize_int4(w2.to(DEV), I)
    q13, s13, q2, s2 = q13.cpu(), s13.to(torch.bfloat16).cpu(), q2.cpu(), s2.to(torch.bfloat16).cpu()
    assert s13.shape == (E, 2 * I, 1) and s2.shape == (E, H, 1)
    pk = lambda q: q.contiguous().view(torch.int32).transpose(1, 2).contiguous()       # noqa: E731
    s = _stand_in(glue, E, K, H, I, torch.bfloat16, w13_weight_packed=pk(q13), w2_weight_packed=pk(q2),
                  w13_weight_scale=s13.transpose(1, 2).contiguous(), w2_weight_scale=s2.transpose(1, 2).contiguous(),
                  quant_method=_QuantMethod(-1))
    assert s.w13_weight_scale.shape == (E, 1, 2 * I)
    s._process_wna16("channel")
    assert type(s.lk_moe).__name__ == "MOE_WNA16" and (s.lk_moe_config.groupN, s.lk_moe_config.groupK) == (1, max(H, I))
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_INT4, groupN=1, groupK=max(H, I))
    a = (q13.numpy(), q2.numpy(), torch_to_bits(s13), torch_to_bits(s2))
    _drive(s, E, K, H, I, lambda x, ids, tw: orc.moe(d, a[0], a[1], x, ids, tw, s13=a[2], s2=a[3]), 3e-3, 1.5e-2)


def test_reference_glue_mxfp4():
    """MOE_MXFP4 through RoutedExperts._process_mxfp4 (:1748-1815): packed E2M1 + E8M0 scales per 32 k"""
    glue = _load_glue()
    E, K, H, I = 4, 2, 512, 256
    w13, w2 = _masters(E, H, I, 4)
    q13, s13 = bench.quantize_mxfp4(w13.to(DEV))
    q2, s2 = bench.quantize_mxfp4(w2.to(DEV))
    q13, s13, q2, s2 = q13.cpu(), s13.cpu(), q2.cpu(), s2.cpu()
    s = _stand_in(glue, E, K, H, I, torch.bfloat16, w13_weight_packed=q13, w2_weight_packed=q2, w13_weight_scale=s13,
                  w2_weight_scale=s2)
    s._process_mxfp4()
    assert type(s.lk_moe).__name__ == "MOE_MXFP4" and (s.lk_moe_config.groupN, s.lk_moe_config.groupK) == (1, 32)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_MXFP4, groupN=1, groupK=32)
    a = (q13.numpy(), q2.numpy(), s13.numpy(), s2.numpy())
    _drive(s, E, K, H, I, lambda x, ids, tw: orc.moe(d, a[0], a[1], x, ids, tw, s13=a[2], s2=a[3]), 3e-3, 1.5e-2)


def test_reference_glue_nvfp4():
    """MOE_NVFP4 through RoutedExperts._process_nvfp4(need_reciprocal_global_scale=True) (:1674-1744): e4m3 block scales
    per 16 k and the per-expert global scales, whose reciprocals the glue hands over"""
    glue = _load_glue()
    E, K, H, I = 4, 2, 512, 256
    w13, w2 = _masters(E, H, I, 5)
    q13, s13, m13 = bench.quantize_nvfp4(w13.to(DEV))
    q2, s2, m2 = bench.quantize_nvfp4(w2.to(DEV))
    q13, s13, m13, q2, s2, m2 = q13.cpu(), s13.cpu(), m13.cpu(), q2.cpu(), s2.cpu(), m2.cpu()
    s = _stand_in(glue, E, K, H, I, torch.bfloat16, w13_weight_packed=q13, w2_weight_packed=q2,
                  w13_weight_scale=s13.view(torch.float8_e4m3fn), w2_weight_scale=s2.view(torch.float8_e4m3fn),
                  w13_weight_global_scale=1.0 / m13, w2_weight_global_scale=1.0 / m2)
    s._process_nvfp4(need_reciprocal_global_scale=True)
    assert type(s.lk_moe).__name__ == "MOE_NVFP4" and (s.lk_moe_config.groupN, s.lk_moe_config.groupK) == (1, 16)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_NVFP4, groupN=1, groupK=16)
    g13, g2 = (1.0 / (1.0 / m13)).numpy(), (1.0 / (1.0 / m2)).numpy()       # what the glue computes and passes
    a = (q13.numpy(), q2.numpy(), s13.numpy(), s2.numpy())
    _drive(s, E, K, H, I, lambda x, ids, tw: orc.moe(d, a[0], a[1], x, ids, tw, s13=a[2], s2=a[3], gs13=g13, gs2=g2),
           3e-3, 1.5e-2)


def test_reference_glue_global_to_local_expert_ids():
    """RoutedExperts.global_to_local_expert_ids (:1332-1342) against ops.global_to_local_expert_ids on the same expert map"""
    glue = _load_glue()
    from lvllm_amd import ops
    E_global, ep, rank = 64, 4, 2
    _, emap = ops.determine_expert_map(ep, rank, E_global)
    s = glue.RoutedExperts.__new__(glue.RoutedExperts)
    s._expert_map = emap
    rng = np.random.default_rng(0)
    ids = torch.from_numpy(rng.integers(-1, E_global, (97, 6)).astype(np.int64)).to(DEV)
    want = s.global_to_local_expert_ids(ids).cpu().numpy()
    got = ops.global_to_local_expert_ids(ids.to(torch.int32), emap.to(DEV)).cpu().numpy()
    np.testing.assert_array_equal(got, want)
