export SCREAM_NO_BUILD=1; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3n2; mkdir -p $O
timeout -k 10 300 python tools/eval_e2e.py --workers 4 --worker-context spawn > $O/eval_spawn.txt 2>&1; cat $O/eval_spawn.txt
timeout -k 10 300 python tools/eval_e2e.py --workers 4 --worker-context forkserver > $O/eval_forkserver.txt 2>&1; tail -3 $O/eval_forkserver.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_icp -- python3 tools/icp_bench.py > $O/icp_bench.txt 2> $O/icp_prof.err; python tools/rocprof_summary.py $O/prof_icp $O/icp && rm -rf $O/prof_icp; head -14 $O/icp_kernel_stats.txt | cut -c1-170
