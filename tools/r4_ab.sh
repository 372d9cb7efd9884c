# same-box A/B/A/B of an environment switch on the default bench: tools/r4_ab.sh <tag> "<ENV=off>" (e.g. SCREAM_RING_PROJ=0)
set -o pipefail
export SCREAM_NO_BUILD=1; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/$1; mkdir -p $O
for i in 1 2; do
  timeout -k 10 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline $3 > $O/on_$i.json 2> $O/on_$i.err || { tail -5 $O/on_$i.err; exit 1; }
  env $2 timeout -k 10 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline $3 > $O/off_$i.json 2> $O/off_$i.err || { tail -5 $O/off_$i.err; exit 1; }
done
python - <<PY
import json
for f in ("on_1","off_1","on_2","off_2"):
    d=json.loads(open("gpurun_out/$1/"+f+".json").read().strip().splitlines()[-1])
    print(f, d["value"], d.get("sustained_value"), d["roofline"]["frac"], d["ms_per_step"])
PY
