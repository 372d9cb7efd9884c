# default bench line with fewer persistent blocks than CUs (tools/_ab/lib_g<N>.so: -DSCREAM_MAX_GRID=N), same box, two passes
set -o pipefail
export SCREAM_NO_BUILD=1; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5grid; mkdir -p $O
for i in 1 2; do for g in ${GRIDS:-256 240 224 192}; do
  if [ $g = 256 ]; then L=""; else L="SCREAM_LIB=tools/_ab/lib_g$g.so"; fi
  env $L timeout -k 10 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary > $O/b_${g}_$i.json 2> $O/b_${g}_$i.err || { tail -5 $O/b_${g}_$i.err; exit 1; }
done; done
python - <<PY
import json
for i in (1,2):
  for g in [int(x) for x in "'${GRIDS:-256 240 224 192}'".strip("'").split()]:
    d=json.loads(open("gpurun_out/r5grid/b_%d_%d.json"%(g,i)).read().strip().splitlines()[-1])
    print(g, i, d["value"], d.get("sustained_value"), d["ms_per_step"], d.get("power"))
PY
