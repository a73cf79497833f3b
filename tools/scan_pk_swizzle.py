#!/usr/bin/env python3
"""Scan gfx950 assembly for the packed-fp32 operand forms that MI355X gets wrong now and then.

Measured with tools/probe_hazard.hip: a v_pk_{fma,mul,add}_f32 whose LOW lane takes the HIGH dword of
a VGPR pair (op_sel = 1 for that source) returns 0 for that operand in lanes 48..63 in ~0.05 % of the
executions while MFMAs are in flight, whenever the same pair is also read through a different swizzle
(in the same instruction or in another packed instruction, even 8 slots away), and in ~0.003 % when that
read of src1 is the only one.  Default selects and lo-broadcasts were always right.  Two checks:
  strict : any VGPR source with op_sel = 1                          (the kernels keep this at zero)
  mixed  : one pair read through two swizzles within WINDOW instructions
usage: scan_pk_swizzle.py file.s [...]          (hipcc -S --cuda-device-only)
       scan_pk_swizzle.py --lib liblkm.so       (disassembles every gfx950 code object of the library)
exit status 1 on a strict hit"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

PK = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\s+(.*)$")
WINDOW = 6


def parse(line):
    m = PK.match(line)
    if not m:
        return None
    op, rest = m.group(1), m.group(2).split(";")[0].split("//")[0]
    sel = {"op_sel": None, "op_sel_hi": None}
    for k in sel:
        mm = re.search(k + r":\[([01,]+)\]", rest)
        if mm:
            sel[k] = [int(x) for x in mm.group(1).split(",")]
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", re.sub(r"\s+op_sel.*$", "", rest).strip())]
    srcs = ops[1:]
    n = len(srcs)
    lo = sel["op_sel"] or [0] * n
    hi = sel["op_sel_hi"] or [1] * n
    out = []
    for i, s in enumerate(srcs):
        if s.startswith("v["):
            out.append((s, (lo[i], hi[i])))
    return op, out


def regs_of(tok):
    m = re.match(r"^[va]\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^[va](\d+)$", tok)
    return {int(m.group(1))} if m else set()


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble_lib(lib, outdir):
    """every gfx950 code object bundled in `lib` -> one .s per object (llvm-objdump --offloading + -d)"""
    work = os.path.join(outdir, "bundles")
    os.makedirs(work)
    shutil.copy(lib, os.path.join(work, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
    outs = []
    for f in sorted(os.listdir(work)):
        if "amdgcn" not in f:
            continue
        dst = os.path.join(outdir, f + ".s")
        with open(dst, "w") as fh:
            subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f], cwd=work, check=True, stdout=fh,
                           stderr=subprocess.DEVNULL)
        outs.append(dst)
    return outs


def scan(paths, verbose=True):
    """returns (strict hits, mixed hits, packed instructions seen)"""
    total = mixed_total = seen_total = 0
    for path in paths:
        kern, last, hits, icount, strict = "?", {}, {}, 0, 0
        for ln, line in enumerate(open(path, errors="replace"), 1):
            lab = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w+)>?:", line)
            if lab:
                kern, last = lab.group(1), {}
                continue
            s = line.split(";")[0].split("//")[0].strip()
            if not s or s.startswith(".") or s.endswith(":"):
                continue
            icount += 1
            p = parse(line)
            if p:
                seen_total += 1
                op, srcs = p
                if any(sw[0] == 1 for _, sw in srcs):
                    strict += 1
                    hits.setdefault(kern, []).append((ln, "hi dword -> low lane", s))
                seen = {}
                for reg, sw in srcs:
                    if reg in seen and seen[reg] != sw:
                        hits.setdefault(kern, []).append((ln, "same instruction", s))
                    seen[reg] = sw
                    if reg in last and last[reg][0] != sw and icount - last[reg][1] <= WINDOW:
                        hits.setdefault(kern, []).append((ln, f"{icount - last[reg][1]} instructions after line {last[reg][2]}", s))
                for reg, sw in seen.items():
                    last[reg] = (sw, icount, ln)
            # any write invalidates what was remembered about the overlapping pairs
            toks = s.split(None, 1)
            if len(toks) == 2 and not toks[0].startswith(("s_", "buffer_store", "global_store", "ds_write", "ds_store")):
                dst = regs_of(toks[1].split(",")[0].strip())
                if dst:
                    for reg in [r for r in last if regs_of(r) & dst]:
                        del last[reg]
        n = sum(len(v) for v in hits.values())
        total += strict
        mixed_total += n - strict
        if verbose:
            print(f"{os.path.basename(path)}: {strict} strict, {n - strict} mixed-swizzle packed-fp32 reads in {len(hits)} kernels")
            for k, v in list(hits.items())[:2]:
                for ln, why, s in v[:3]:
                    print(f"   {k[:60]} line {ln} ({why}): {s[:110]}")
    return total, mixed_total, seen_total


def main():
    args = sys.argv[1:]
    if args and args[0] == "--lib":
        with tempfile.TemporaryDirectory() as td:
            strict, mixed, seen = scan(disassemble_lib(args[1], td))
    else:
        strict, mixed, seen = scan(args)
    print(f"total: {seen} packed-fp32 instructions, {strict} strict hits, {mixed} mixed-swizzle hits")
    return 1 if strict else 0


if __name__ == "__main__":
    sys.exit(main())
