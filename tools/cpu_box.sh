cd $GRAFT_REPO_ROOT
cat > /tmp/cpu_t.py <<'PY'
import sys, time, os; sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as orc
E,K,H,I,M=8,2,4096,14336,32
rng=np.random.default_rng(0)
w13=rng.integers(0x3c00,0x3e00,size=(E,2*I,H),dtype=np.uint16)
w2=rng.integers(0x3c00,0x3e00,size=(E,H,I),dtype=np.uint16)
x=rng.integers(0x3c00,0x3e00,size=(M,H),dtype=np.uint16)
tw,ids=orc.topk_softmax(rng.standard_normal((M,E)).astype(np.float32),K)
d=orc.MoeDesc(E=E,H=H,I=I,act_dtype=orc.BF16,wfmt=orc.W_BF16)
orc.moe(d,w13,w2,x,ids,tw)
t=time.perf_counter(); n=5
for _ in range(n): orc.moe(d,w13,w2,x,ids,tw)
dt=(time.perf_counter()-t)/n
print(f"thr={orc.num_threads()} sched={os.environ.get('OMP_SCHEDULE')} bind={os.environ.get('OMP_PROC_BIND')}: {dt*1e3:.1f} ms/step {(w13.nbytes+w2.nbytes)/dt/1e9:.1f} GB/s")
PY
for cfg in "128 dynamic,64 false" "128 static false" "128 dynamic,64 spread" "64 dynamic,64 spread" "256 dynamic,16 false" "32 dynamic,64 spread"; do set -- $cfg; OMP_NUM_THREADS=$1 OMP_SCHEDULE=$2 OMP_PROC_BIND=$3 python /tmp/cpu_t.py 2>&1 | tail -1; done
