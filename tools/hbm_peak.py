#!/usr/bin/env python3
"""Measures the achievable HBM read bandwidth of this MI355X (reference point for roofline.frac)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import torch
from lvllm_amd import _clib
lib = _clib.lib()
buf = torch.empty(2 * 1024**3, dtype=torch.uint8, device="cuda:0").random_(0, 255)
for blocks in (1024, 2048, 4096, 8192):
    for unroll in (2, 4, 8):
        ms = C.c_float()
        _clib.check(lib.lkm_hbm_read_probe(None, C.c_void_p(buf.data_ptr()), buf.numel(), blocks, unroll, 20, C.byref(ms)))
        print(f"blocks={blocks} unroll={unroll}: {ms.value*1e3:.1f} us  {buf.numel()/ms.value/1e6:.0f} GB/s")
