# same-box A/B/A/B of a variant library against the shipped one on the default bench: tools/r5_ab_lib.sh <tag> <variant.so> ["<extra bench args>"]
set -o pipefail
export SCREAM_NO_BUILD=1; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/$1; mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary $3 > $O/base_$i.json 2> $O/base_$i.err || { tail -5 $O/base_$i.err; exit 1; }
  SCREAM_LIB=$2 timeout -k 10 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary $3 > $O/var_$i.json 2> $O/var_$i.err || { tail -5 $O/var_$i.err; exit 1; }
done
python - <<PY
import json
for i in (1,2,3):
  for v in ("base","var"):
    d=json.loads(open("gpurun_out/$1/%s_%d.json"%(v,i)).read().strip().splitlines()[-1])
    print(v, i, d["value"], d.get("sustained_value"), d["roofline"]["frac"], d["ms_per_step"], d.get("power"))
PY
