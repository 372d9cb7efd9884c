# round-4 opening measurements (one gpurun call): suite with the new tests, bench default + one step at a time, ablation of the final projection kernel
set -o pipefail
export SCREAM_NO_BUILD=1; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/gpu_tests.txt 2>&1; rc=$?; tail -5 $O/gpu_tests.txt
grep -h "trained-like\|DEM, trained" $O/gpu_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 40 --warmup 6 > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 python bench.py --steps 40 --warmup 6 --lanes 1 --no-cpu-baseline > $O/bench_l1.json 2> $O/bench_l1.err || exit 1
timeout -k 10 300 python tools/proj_ablate.py run > $O/proj_ablate.txt 2>&1 || exit 1
cat $O/proj_ablate.txt
python - <<'PY'
import json
O="gpurun_out/r4a/"
for f in ("bench","bench_l1"):
    d=json.loads(open(O+f+".json").read().strip().splitlines()[-1])
    print(f, d["value"], d.get("sustained_value"), d["roofline"]["frac"], d["ms_per_step"])
    for r in d["roofline"]["by_gemm_shape"]: print("   ", r)
PY
