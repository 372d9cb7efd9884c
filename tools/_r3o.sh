export SCREAM_NO_BUILD=1; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_data_files.py -m gpu -x -q > $O/tests.txt 2>&1; rc=$?; tail -12 $O/tests.txt
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
timeout -k 10 300 python tools/eval_e2e.py --workers 0,4,8 > $O/eval_e2e.txt 2>&1; cat $O/eval_e2e.txt
for wl in kitti; do for gb in h2 h1; do SCREAM_GEMM=$gb timeout -k 10 300 python bench.py --workload $wl --pairs 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_${wl}_$gb.json 2> $O/bench_${wl}_$gb.err; done; done
SCREAM_GEMM=h1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_3dmatch_h1.json 2> $O/bench_3dmatch_h1.err
python -c "
import json
for f in ('kitti_h2','kitti_h1','3dmatch_h1'):
    d=json.loads(open('$O/bench_%s.json'%f).read().strip().splitlines()[-1]);print(f,d['value'],d['sustained_value'],d['roofline']['frac'],d['dtype'])"
