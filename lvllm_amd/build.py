"""Builds liblkm.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.

No torch in the build: the library is plain HIP behind a C ABI (include/lkm.h).  Each .hip
translation unit is compiled in parallel, objects are cached by source hash under
lvllm_amd/csrc/_obj/, and the link produces lvllm_amd/liblkm.so (git-ignored; it travels to
the GPU box with the tree).  hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / "_obj"
LIB = PKG / "liblkm.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: liblkm.so cannot be built (ROCm toolchain required)")
    return exe


_INC = None


def _closure(src: Path) -> list[Path]:
    """The files `src` includes, transitively (quoted includes, resolved next to the including file): an object
    depends on these and on nothing else, so touching one header recompiles only the units that see it."""
    import re
    global _INC
    if _INC is None:
        _INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)
    seen: dict[Path, None] = {}
    todo = [src]
    while todo:
        f = todo.pop()
        for name in _INC.findall(f.read_text(errors="ignore")):
            q = (f.parent / name).resolve()
            if q.exists() and q not in seen:
                seen[q] = None
                todo.append(q)
    return sorted(seen)


def _digest(src: Path, flags: list[str]) -> str:
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for p in [src, *_closure(src)]:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()[:16]


def _compile(src: Path, flags: list[str], objdir: Path, verbose: bool) -> Path:
    tag = _digest(src, flags)
    obj = objdir / f"{src.stem}.{tag}.o"
    if obj.exists():
        return obj
    for old in objdir.glob(f"{src.stem}.*.o"):
        old.unlink()
    cmd = [_hipcc(), *flags, "-c", str(src), "-o", str(obj)]
    if verbose:
        print("[lkm build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = False, extra_flags: tuple[str, ...] = (),
          out: Path | None = None, only: str = "") -> Path:
    """Compile (if stale) and link liblkm.so; returns its path.  extra_flags/out build an experimental variant next
    to the default library (development only); `only` (comma list) restricts the extra flags to the sources whose file name
    contains one of its entries -- the others are taken from the default build's object cache."""
    global OBJ, LIB
    base_flags, base_obj = list(FLAGS), CSRC / "_obj"
    var_flags, var_obj = base_flags, base_obj
    if extra_flags or out:
        var_flags = [*base_flags, *extra_flags]
        tag = hashlib.sha256((" ".join(extra_flags) + "|" + only).encode()).hexdigest()[:8]
        var_obj = CSRC / f"_obj_{tag}"
        OBJ = var_obj
        LIB = out or PKG / f"liblkm_{tag}.so"
    base_obj.mkdir(parents=True, exist_ok=True)
    var_obj.mkdir(parents=True, exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if force:
        for o in var_obj.glob("*.o"):
            o.unlink()

    def one(s: Path) -> Path:
        if only and not any(o in s.name for o in only.split(",")):
            return _compile(s, base_flags, base_obj, verbose)
        return _compile(s, var_flags, var_obj, verbose)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(one, srcs))
    stamp = OBJ / "link.stamp"
    want = "no-hip-rt " + " ".join(o.name for o in objs)
    if LIB.exists() and stamp.exists() and stamp.read_text() == want and not force:
        return LIB
    # -no-hip-rt: no DT_NEEDED on libamdhip64; the loader (_clib._bind_hip_runtime) binds the library to
    # the ONE HIP runtime of the process (torch's bundled one when PyTorch-ROCm is loaded).
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-no-hip-rt", "-o", str(LIB), *map(str, objs)]
    if verbose:
        print("[lkm build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link of liblkm.so failed:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(want)
    return LIB


if __name__ == "__main__":
    extra = tuple(a[len("--flag="):] for a in sys.argv if a.startswith("--flag="))
    outs = [a[len("--out="):] for a in sys.argv if a.startswith("--out=")]
    onlys = [a[len("--only="):] for a in sys.argv if a.startswith("--only=")]
    p = build(force="--force" in sys.argv, verbose="-q" not in sys.argv, extra_flags=extra,
              out=Path(outs[0]).resolve() if outs else None, only=onlys[0] if onlys else "")
    print(p)
