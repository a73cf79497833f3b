"""The round-3 fp8 x fp8 prefill kernel (lvllm_amd/csrc/gemm_prefill_a8w.h) keeps its operands in a FIXED register map
(v40..v255) behind a hand-counted s_waitcnt ledger.  clang ignores amdgpu_num_vgpr below ~57 registers, so what the
kernel relies on is checked on the generated assembly (tools/scan_a8w_codegen.py): the compiler writes no VGPR of the
map outside the inline asm, uses no scratch, and issues no vector load / vmcnt wait of its own between the barriers.
Cross-compiles one translation unit for gfx950 (about a minute; no GPU needed)."""
import re
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.timeout(900)
def test_generated_code_keeps_the_fixed_register_map(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not available")
    sys.path.insert(0, str(ROOT))
    from concurrent.futures import ThreadPoolExecutor
    from lvllm_amd import build
    flags = [f for f in build.FLAGS if f != "-fPIC"]

    def compile_tu(name):
        asm = tmp_path / f"{name}.s"
        r = subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", "-o", str(asm),
                            str(ROOT / "lvllm_amd" / "csrc" / f"{name}.hip")], capture_output=True, text=True)
        return name, asm, r
    # both activation dtypes (their own translation units), compiled side by side
    with ThreadPoolExecutor(max_workers=2) as ex:
        built = list(ex.map(compile_tu, ["gemm_tiled_fp8a8_bf16", "gemm_tiled_fp8a8_f16"]))
    for name, asm, r in built:
        assert r.returncode == 0, r.stderr[-2000:]
        s = subprocess.run([sys.executable, str(ROOT / "tools" / "scan_a8w_codegen.py"), str(asm)], capture_output=True, text=True)
        assert s.returncode == 0, name + "\n" + s.stdout[-4000:] + s.stderr[-2000:]
        assert "3 kernels, 0 violations" in s.stdout, name + "\n" + s.stdout[-2000:]
        # the three instantiations (gated GEMM1, plain GEMM1, GEMM2) use no scratch and the whole register file
        text = asm.read_text(errors="ignore")
        names = re.findall(r"^(_ZN3lkm23gemm_prefill_a8w_kernel\S*):", text, flags=re.M)
        assert len(names) == 3
        for n in names:
            m = re.search(re.escape(n) + r":.*?; ScratchSize: (\d+)", text, flags=re.S)
            assert m and m.group(1) == "0", (n, m and m.group(1))
