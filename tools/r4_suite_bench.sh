# one gpurun call: the whole GPU suite, then bench.py default + one step at a time (tag = $1, extra env for the benches = $2)
set -o pipefail
export SCREAM_NO_BUILD=1; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/$1; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; rc=$?; tail -5 $O/gpu_tests.txt
[ $rc -eq 0 ] || exit 1
env $2 timeout -k 10 300 python bench.py --steps 40 --warmup 6 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
env $2 timeout -k 10 300 python bench.py --steps 40 --warmup 6 --lanes 1 --no-cpu-baseline > $O/bench_l1.json 2> $O/bench_l1.err || exit 1
python - <<PY
import json
O="gpurun_out/$1/"
for f in ("bench","bench_l1"):
    d=json.loads(open(O+f+".json").read().strip().splitlines()[-1])
    print(f, d["value"], d.get("sustained_value"), d["roofline"]["frac"], d["ms_per_step"])
    for r in d["roofline"]["by_kernel"][:4]: print("   ", r)
    for r in d["roofline"].get("by_gemm_shape", []): print("   ", r)
PY
