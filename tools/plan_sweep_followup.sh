run() { for r in uniform zipf; do echo "== $1 M=$2 $r"; python tools/sweep.py --workload $1 --M $2 --routing $r --reps 30 --cfgs "$3" 2>&1 | grep "^\[" | cut -c1-330; done; }
# (first config of a process measures a few % slow: repeat it)
for m in 32 40; do run dsv3_ep8_rank_fp8w8a16_rows256 $m ";;tiled=64;tiled=32;tiled=-1;hybrid=1"; done
for m in 32 40; do run mixtral8x7b_fp8w8a8_decode_m32 $m ";;hybrid=1;tiled=32"; done
for m in 64 128; do run mixtral8x7b_int4g128_decode_m128 $m ";;pf=-1;pf=5,tiled=64"; done
