#!/usr/bin/env python3
"""Development script (this container only: needs /root/reference): import the reference's
vllm/model_executor/layers/fused_moe/modular_kernel.py with stand-ins for the third-party modules this image lacks, then
run lvllm_amd.modular.bind_vllm_base() against the REAL base classes and report what happened."""
import os
os.environ.setdefault("LVLLM_MOE_NUMA_ENABLED", "1")      # routed_experts.py:37-38 then imports `lk_moe`: this repo's
import importlib
import importlib.abc
import importlib.machinery
import sys
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference")
STUBBED = []


class _Anything(types.ModuleType):
    """a module whose every attribute exists (classes that can be subclassed, called and indexed)"""
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        obj = type(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None,
                              "__class_getitem__": classmethod(lambda cls, item: cls)})
        setattr(self, name, obj)
        return obj


ALLOWED = set()      # top-level third-party packages found missing, one import attempt at a time


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] not in ALLOWED:
            return None
        STUBBED.append(name)
        return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = _Anything(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _import_with_stubs(name):
    """ONE attempt per process (re-importing vllm re-registers its torch custom ops); the parent loop below restarts
    this script with the next missing module added to --allowed"""
    try:
        return importlib.import_module(name)
    except ModuleNotFoundError as e:
        top = (e.name or "").split(".")[0]
        if not top or top in ("vllm", "torch") or top in ALLOWED:
            raise
        print(f"MISSING {top}")
        sys.exit(3)


def main():
    if "--child" not in sys.argv:
        import subprocess
        allowed = []
        for _ in range(80):
            r = subprocess.run([sys.executable, __file__, "--child", ",".join(allowed)], capture_output=True, text=True)
            miss = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("MISSING ")]
            if r.returncode == 3 and miss:
                allowed.append(miss[-1])
                continue
            print(r.stdout[-6000:])
            if r.returncode not in (0, 2):
                print(r.stderr[-3000:])
            return r.returncode
        return 1
    ALLOWED.update(a for a in sys.argv[sys.argv.index("--child") + 1].split(",") if a)
    sys.meta_path.append(_StubFinder())
    try:
        mk = _import_with_stubs("vllm.model_executor.layers.fused_moe.modular_kernel")
    except Exception as e:
        import traceback
        tb = traceback.extract_tb(e.__traceback__)
        print(f"IMPORT FAILED: {type(e).__name__}: {e}")
        print("chain:")
        for fr in tb:
            if "/root/reference" in fr.filename:
                print(f"  {fr.filename.replace('/root/reference/', '')}:{fr.lineno}  {fr.line}")
        print("third-party modules stubbed on the way:", sorted(ALLOWED))
        return 1
    print("imported", mk.__name__, "; third-party modules stubbed:", sorted(ALLOWED))
    from lvllm_amd import modular
    cls = modular.bind_vllm_base()
    print("bind_vllm_base() ->", cls, "mro:", [c.__name__ for c in cls.__mro__][:6])
    abstract = sorted(getattr(cls, "__abstractmethods__", ()))
    print("abstract methods left unimplemented:", abstract)
    pf = getattr(modular, "bind_vllm_prepare_finalize", None)
    if pf is not None:
        c2 = pf()
        print("bind_vllm_prepare_finalize() ->", c2, "abstract left:", sorted(getattr(c2, "__abstractmethods__", ())))
    re = sys.modules.get("vllm.model_executor.layers.fused_moe.routed_experts")
    if re is not None and hasattr(re, "lk_moe"):
        print("routed_experts.lk_moe is", re.lk_moe.__name__, getattr(re.lk_moe, "__version__", "?"), "from", re.lk_moe.__file__)
        print("RoutedExperts has", [m for m in ("_process_fp8", "_cpu_decode", "_gpu_prefill") if hasattr(re.RoutedExperts, m)])
    return 0 if not abstract else 2


if __name__ == "__main__":
    sys.exit(main())
