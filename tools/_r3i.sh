set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3i; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
timeout -k 10 300 python bench.py --steps 40 --warmup 5 > $O/bench_l2.json 2> $O/bench_l2.err || exit 1
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --lanes 1 --no-cpu-baseline > $O/bench_l1.json 2> $O/bench_l1.err || exit 1
for L in 1 2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_l$L -- python3 bench.py --steps 7 --warmup 3 --gen-procs 1 --no-cpu-baseline --no-power --no-sustain --lanes $L > $O/bench_under_rocprof_l$L.json 2> $O/rocprof_l$L.err || exit 1
  python tools/rocprof_summary.py $O/prof_l$L $O/lanes$L && rm -rf $O/prof_l$L
done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --gen-procs 1 --no-cpu-baseline --no-power --no-sustain --lanes 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --gen-procs 1 --no-cpu-baseline --no-power --no-sustain --lanes 1 > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
python tools/pmc_traffic.py $O/fetch $O/write $O/traffic.json --lanes 1 --backend h2 > $O/traffic.txt && python tools/rocprof_summary.py $O/fetch $O/pmc_fetch && python tools/rocprof_summary.py $O/write $O/pmc_write && rm -rf $O/fetch $O/write
cat $O/traffic.txt
timeout -k 10 400 python tools/eval_e2e.py > $O/eval_e2e.txt 2>&1 || exit 1
cat $O/eval_e2e.txt
