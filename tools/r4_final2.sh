# final validation of the round's last code (one gpurun call): suite, tail stamps, tail / projection soaks, then tools/r4_measure.sh
set -o pipefail
export SCREAM_NO_BUILD=1; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4final2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/tail_stamps.py run > $O/tail_stamps.txt 2>&1 || { tail -5 $O/tail_stamps.txt; exit 1; }
cat $O/tail_stamps.txt
timeout -k 10 300 python tools/tail_soak.py 60 2>&1 | tail -1 | tee $O/tail_soak.txt
timeout -k 10 300 python tools/proj_soak.py 60 2>&1 | tail -1 | tee $O/proj_soak.txt
timeout -k 10 300 python tools/forward_soak.py 40 2>&1 | tail -1 | tee $O/forward_soak.txt
bash tools/r4_measure.sh r4final2m
