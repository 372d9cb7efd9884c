# round-4 measurement set (one gpurun call; tag = $1): bench default + one step at a time, rocprofv3 kernel trace of both, PMC traffic,
# end to end (3DMatch and KITTI harness), the other workloads, the 2-rank rehearsal.  The suite runs separately (tools/r4_suite_bench.sh).
set -o pipefail
export SCREAM_NO_BUILD=1; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/$1; mkdir -p $O
timeout -k 10 400 python bench.py --steps 40 --warmup 6 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
timeout -k 10 300 python bench.py --steps 40 --warmup 6 --lanes 1 --no-cpu-baseline > $O/bench_l1.json 2> $O/bench_l1.err || exit 1
for M in "lanes1:--lanes 1" "default:"; do
  T=${M%%:*}; A=${M#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 bench.py --steps 7 --warmup 3 --gen-procs 1 --no-cpu-baseline --no-power --no-sustain --no-secondary $A > $O/bench_under_rocprof_$T.json 2> $O/rocprof_$T.err || { tail -5 $O/rocprof_$T.err; exit 1; }
  python tools/rocprof_summary.py $O/prof_$T $O/$T && rm -rf $O/prof_$T
done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --gen-procs 1 --no-cpu-baseline --no-power --no-sustain --no-secondary --lanes 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --gen-procs 1 --no-cpu-baseline --no-power --no-sustain --no-secondary --lanes 1 > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
python tools/pmc_traffic.py $O/fetch $O/write $O/traffic.json --lanes 1 --backend h2 > $O/traffic.txt && python tools/rocprof_summary.py $O/fetch $O/pmc_fetch && python tools/rocprof_summary.py $O/write $O/pmc_write && rm -rf $O/fetch $O/write
cat $O/traffic.txt
timeout -k 10 400 python tools/eval_e2e.py 4096 --workers 0,4 > $O/eval_e2e.txt 2>&1 || { tail -5 $O/eval_e2e.txt; exit 1; }
cat $O/eval_e2e.txt
timeout -k 10 600 python tools/eval_e2e_kitti.py 2048 > $O/eval_e2e_kitti.txt 2>&1 || { tail -5 $O/eval_e2e_kitti.txt; exit 1; }
cat $O/eval_e2e_kitti.txt
for wl in kitti uniform64k; do
  P=32; [ $wl = uniform64k ] && P=16
  timeout -k 10 300 python bench.py --workload $wl --pairs $P --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err || exit 1
done
SCREAM_BENCH_BACKEND=gloo SCREAM_BENCH_SINGLE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_2rank.json 2> $O/bench_2rank.err || { tail -5 $O/bench_2rank.err; exit 1; }
python - <<PY
import json
O="gpurun_out/$1/"
for f in ("bench","bench_l1","bench_kitti","bench_uniform64k","bench_2rank"):
    d=json.loads(open(O+f+".json").read().strip().splitlines()[-1])
    print(f, d["value"], d.get("sustained_value"), d["roofline"]["frac"], d["ms_per_step"], d["n_gpus"], d.get("cpu_baseline",{}) and d["cpu_baseline"].get("value"))
PY
