#!/usr/bin/env python3
"""Check the generated code of the round-3 fp8 prefill kernels (gemm_prefill_a8w.h) against the invariants its
hand-written K loop relies on.  The loop keeps its operands in a fixed register map (v40..v255) that the COMPILER must not
touch -- clang's amdgpu_num_vgpr is ignored below ~57 registers, so this is verified on the assembly instead:

  * outside the inline-asm statements, no instruction writes a VGPR >= 40 -- except between the A8W_EPILOGUE_BEGIN / _END
    markers, where the limit is v78 (nothing of v40..v77 is live across an item's epilogue; v78..v79 spare,
    v80.. = the weight ring with loads in flight, v128.. = the accumulators the epilogue itself reads through asm);
  * no scratch (a spill is a vector-memory operation inside the hand-counted vmcnt ledger);
  * no compiler-issued s_waitcnt vmcnt / vector load between the first and the last s_barrier of a kernel.

usage: scan_a8w_codegen.py file.s        (hipcc -S --cuda-device-only of a translation unit holding the kernels)
exit status 1 on a violation"""
import re
import sys

KERNEL = re.compile(r"^(_ZN3lkm23gemm_prefill_a8w_kernel\S*):")
VREG = re.compile(r"^(?:v(\d+)|v\[(\d+):(\d+)\])$")
NO_DEST = ("global_store", "buffer_store", "ds_write", "scratch_store", "v_cmp", "v_cmpx", "global_load_lds", "buffer_load_dwordx4 v",
           "s_", "v_readlane", "v_readfirstlane", "ds_gws", "v_nop")


def scan(path):
    bad = []
    kern, inasm, epi = None, False, False
    seen_barrier, n_kern = False, 0
    pending = []      # compiler vector-memory ops / vmcnt waits after the first barrier (flushed at the next barrier)
    for ln, raw in enumerate(open(path, errors="ignore"), 1):
        line = raw.rstrip("\n")
        m = KERNEL.match(line)
        if m:
            kern, inasm, epi, seen_barrier, pending = m.group(1), False, False, False, []
            n_kern += 1
            continue
        if kern is None:
            continue
        if line.startswith(".Lfunc_end"):
            kern = None
            continue
        s = line.strip()
        if "ScratchSize:" in s:
            pass
        if ";;#ASMSTART" in s:
            inasm = True
            continue
        if ";;#ASMEND" in s:
            inasm = False
            continue
        if inasm:
            if "A8W_EPILOGUE_BEGIN" in s:
                epi = True
            if "A8W_EPILOGUE_END" in s:
                epi = False
            if re.match(r"s_barrier\b", s):
                seen_barrier = True
                bad += pending
                pending = []
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        if op.startswith("scratch_"):
            bad.append((kern, ln, "scratch access: " + s))
        if seen_barrier and (re.match(r"(global|buffer|flat)_load", op) or (op == "s_waitcnt" and "vmcnt" in args)):
            pending.append((kern, ln, "compiler vector-memory op / vmcnt wait inside the K loop: " + s))
        if op.startswith(NO_DEST) or s.startswith("buffer_load_dwordx4 v") and "lds" in s:
            continue
        dest = args.split(",")[0].strip()
        mm = VREG.match(dest)
        if not mm:
            continue
        hi = int(mm.group(1)) if mm.group(1) is not None else int(mm.group(3))
        limit = 78 if epi else 40
        if hi >= limit:
            bad.append((kern, ln, f"compiler writes v{hi} (limit v{limit - 1}{' inside the epilogue' if epi else ''}): " + s))
    return n_kern, bad


def main():
    rc = 0
    for path in sys.argv[1:]:
        n, bad = scan(path)
        print(f"{path}: {n} kernels, {len(bad)} violations")
        for k, ln, what in bad[:40]:
            print(f"  {k[:60]} line {ln}: {what}")
        if bad or n == 0:
            rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    main()
