#!/bin/bash
for v in "" _v1 _v2 _v3 _v4; do
  echo "=== lib$v"
  LKM_LIB_PATH=$PWD/lvllm_amd/liblkm$v.so timeout 200 python tools/_dbg/int4_dbg.py 2>&1 | grep "bad$" | awk '{s+=$(NF-1); n++; if ($(NF-1)>0) f++} END {print n" cases, "f" failing, "s" bad elements"}'
done
