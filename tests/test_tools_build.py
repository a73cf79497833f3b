"""The stand-alone MI355X probes under tools/ (development instruments whose logs DESIGN.md quotes) still build: hipcc
cross-compiles them for gfx950 without a GPU.  Nothing is run."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        pytest.skip("hipcc not available")
    return exe


@pytest.mark.timeout(600)
def test_instruction_stream_probe_builds():
    _hipcc()
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "probe_mfma_valu.py"), "--build-only", "--quick"],
                       capture_output=True, text=True, timeout=580)
    assert r.returncode == 0 and "built" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("src", ["probe_int4_unit.hip", "probe_mfma_issue.hip", "probe_mfma_layout.hip"])
def test_kernel_probes_build(src, tmp_path):
    if not (ROOT / "tools" / src).exists():
        pytest.skip(f"{src} not in this tree")
    r = subprocess.run([_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", str(ROOT / "lvllm_amd" / "csrc"),
                        str(ROOT / "tools" / src), "-o", str(tmp_path / "probe")], capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-2000:]
