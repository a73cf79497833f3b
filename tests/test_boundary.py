"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/*.h (lkm.h and its extension lkm_eplb.h) declare, the lk_moe module has the reference's surface, and the product path never
touches the oracle."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    text = "".join(p.read_text() for p in sorted((ROOT / "include").glob("*.h")))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lkm_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from lvllm_amd import build
    lib_path = build.build()
    assert lib_path.exists()
    from lvllm_amd import _clib
    lib = _clib.lib()          # binds the process's single HIP runtime, then dlopens liblkm.so
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported by liblkm.so"
    assert sorted(_clib.EXPORTS) == names, "ctypes binding and header disagree"
    assert _clib.lib().lkm_abi_version() == _clib.LKM_ABI_VERSION


def test_generated_code_has_no_mixed_swizzle_packed_fp32():
    """MI355X returns a wrong low lane now and then for a v_pk_{fma,mul,add}_f32 that takes the HIGH dword of a
    VGPR pair for its LOW lane while the pair is also read through another swizzle (measured:
    tools/probe_hazard.hip; the int4 decoder and the W8A8 scale product hit it until their operands were
    made opaque pairs, lkm_common.h splat2_opaque).  The compiler emits that form on its own for
    `vector * scalar`, so the disassembly of every kernel in the built library is checked."""
    import importlib.util
    from lvllm_amd import build
    lib_path = build.build()
    spec = importlib.util.spec_from_file_location("scan_pk_swizzle", ROOT / "tools" / "scan_pk_swizzle.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        strict, mixed, seen = mod.scan(mod.disassemble_lib(str(lib_path), td), verbose=False)
    assert seen > 10000, "disassembly found no packed fp32 instructions: scanner broken?"
    assert strict == 0 and mixed == 0, (strict, mixed)


def test_config_struct_layout_matches_header():
    from lvllm_amd import _clib
    text = (ROOT / "include" / "lkm.h").read_text()
    body = re.search(r"typedef struct LkmConfig \{(.*?)\} LkmConfig;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(?:int32_t|float)\s+([a-z_A-Z0-9]+)(?:\[\d+\])?;", body)
    assert fields == [f[0] for f in _clib.LkmConfig._fields_]
    assert ctypes.sizeof(_clib.LkmConfig) == 4 * (len(fields) - 1) + 4 * 7 == 124


def test_lk_moe_surface_matches_reference_call_sites():
    """names used by routed_experts.py:1490-1533, 1596-1616, 1648-1668, 1840-1899."""
    import lk_moe
    cfg = lk_moe.MOEConfigV2()
    for attr in ("num_processes", "process_id", "gpu_id", "has_gate_proj", "expert_num", "top_k",
                 "hidden_size", "intermediate_size", "max_batch_size", "max_num_seqs", "stride",
                 "group_min_len", "group_max_len", "groupN", "groupK", "activation_type",
                 "swiglu_alpha", "swiglu_limit", "use_gpu_prefill"):
        assert hasattr(cfg, attr), attr
        setattr(cfg, attr, getattr(cfg, attr))
    for cls in ("MOE_BF16", "MOE_FP16", "MOE_FP8", "MOE_FP8_FP16", "MOE_WNA16", "MOE_WNA16_FP16",
                "MOE_NVFP4", "MOE_NVFP4_FP16", "MOE_MXFP4", "MOE_MXFP4_FP16"):
        c = getattr(lk_moe, cls)
        for m in ("cpu_decode", "cpu_prefill", "gpu_prefill"):
            assert callable(getattr(c, m))


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import lk_moe
    from lvllm_amd._clib import LkmError
    cfg = lk_moe.MOEConfigV2()
    cfg.expert_num, cfg.top_k, cfg.hidden_size, cfg.intermediate_size = 2, 1, 64, 64
    buf = (ctypes.c_uint16 * (2 * 128 * 64))()
    with pytest.raises(LkmError):
        lk_moe.MOE_BF16(cfg, ctypes.addressof(buf), ctypes.addressof(buf), 0, 0, 0, 0)
    from lvllm_amd import ops
    with pytest.raises(ValueError):
        ops.topk_softmax(torch.zeros((2, 4)), 2, True)


def test_product_path_never_references_the_oracle():
    bad = []
    for p in list((ROOT / "lvllm_amd").rglob("*")) + list((ROOT / "lk_moe").rglob("*")):
        if p.suffix in (".py", ".hip", ".h", ".cpp", ".c") and "_obj" not in p.parts:
            t = p.read_text(errors="ignore")
            if re.search(r"\boracle\b", t) and p.name not in ("lkm_common.h", "routing.hip"):
                bad.append(str(p))
            if re.search(r"^\s*(from|import)\s+oracle", t, re.M) or "liblkm_oracle" in t or "lkm_ref" in t:
                bad.append(str(p) + " (imports it)")
    # development scripts measure the product: they may not import the checker either (only tests/, bench.py's
    # cpu_baseline leg and __graft_entry__.smoke() do)
    for p in (ROOT / "tools").glob("*.py"):
        if re.search(r"^\s*(from|import)\s+oracle", p.read_text(errors="ignore"), re.M):
            bad.append(str(p) + " (tools/ imports the oracle)")
    assert not bad, bad
