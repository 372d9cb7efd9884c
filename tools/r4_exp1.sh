set -o pipefail
export SCREAM_NO_BUILD=1; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4x1; mkdir -p $O
T_TAGS=_r0,_r3,_r6,_r0,_r3,_r6 T_VARIANTS=0 timeout -k 10 500 python tools/tail_ablate.py run 2>&1 | tee $O/tail_ride0.txt || exit 1
for i in 1 2; do
P_KERNELS=ring timeout -k 10 300 python tools/proj_ring_bench.py 2>&1 | grep -E "ring" | tee -a $O/ring_prefetch.txt
SCREAM_LIB=tools/_abl_ring/lib_a16.so P_KERNELS=ring timeout -k 10 300 python tools/proj_ring_bench.py 2>&1 | grep -E "ring" | sed 's/ring /nopf /' | tee -a $O/ring_prefetch.txt
done
for v in full:scream_amd/libscream_hip.so nopf:tools/_abl_ring/lib_a16.so; do
  t=${v%%:*}; l=${v#*:}
  SCREAM_LIB=$l P_NOSMI=1 P_SECS=0.05 P_KERNELS=ring P_SHAPES=stem,crosskv timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$t -- python3 tools/proj_ring_bench.py > $O/pmc_$t.txt 2> $O/pmc_$t.err || { tail -5 $O/pmc_$t.err; exit 1; }
  python tools/rocprof_summary.py $O/fetch_$t $O/pmc_ring_$t && rm -rf $O/fetch_$t
done
cat $O/pmc_ring_full*.txt $O/pmc_ring_nopf*.txt | grep -i "proj_ring\|Kernel" | head
