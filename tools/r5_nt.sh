# round 4 (late): the non-temporal-hint builds (tools/_ab/lib_nt*.so) against the shipped library: HBM fetch by the counters, then a same-box A/B/A/B
set -o pipefail; VARIANTS=${VARIANTS:-"base nt2 nt3 nt6 nt7"}
export SCREAM_NO_BUILD=1; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5nt; mkdir -p $O
for v in $VARIANTS; do
  if [ $v = base ]; then unset SCREAM_LIB; else export SCREAM_LIB=tools/_ab/lib_$v.so; fi
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -- python3 bench.py --steps 2 --warmup 1 --gen-procs 1 --no-cpu-baseline --no-power --no-sustain --no-secondary --lanes 1 > $O/pmc_fetch_$v.json 2> $O/pmc_fetch_$v.err || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$v -- python3 bench.py --steps 2 --warmup 1 --gen-procs 1 --no-cpu-baseline --no-power --no-sustain --no-secondary --lanes 1 > $O/pmc_write_$v.json 2> $O/pmc_write_$v.err || exit 1
  python tools/pmc_traffic.py $O/fetch_$v $O/write_$v $O/traffic_$v.json --lanes 1 --backend h2 > $O/traffic_$v.txt && rm -rf $O/fetch_$v $O/write_$v
  head -5 $O/traffic_$v.txt
done
unset SCREAM_LIB
for i in 1 2; do for v in $VARIANTS; do
  if [ $v = base ]; then L=""; else L="SCREAM_LIB=tools/_ab/lib_$v.so"; fi
  env $L timeout -k 10 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary > $O/b_${v}_$i.json 2> $O/b_${v}_$i.err || { tail -5 $O/b_${v}_$i.err; exit 1; }
done; done
python - <<PY
import json
for i in (1,2):
  for v in "'$VARIANTS'".strip("'").split():
    d=json.loads(open("gpurun_out/r5nt/b_%s_%d.json"%(v,i)).read().strip().splitlines()[-1])
    print(v, i, d["value"], d.get("sustained_value"), d["roofline"]["frac"], d["ms_per_step"])
PY
