set -o pipefail
export SCREAM_NO_BUILD=1; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4q; mkdir -p $O
for i in 1 2; do for a in 2 3 4; do
  timeout -k 10 400 python bench.py --steps 42 --warmup 6 --no-cpu-baseline --no-secondary --alternate $a > $O/a${a}_$i.json 2> $O/a${a}_$i.err || { tail -5 $O/a${a}_$i.err; exit 1; }
done; done
python - <<PY
import json
for i in (1,2):
  for a in (2,3,4):
    d=json.loads(open("gpurun_out/r4q/a%d_%d.json"%(a,i)).read().strip().splitlines()[-1])
    print("alternate", a, d["value"], d.get("sustained_value"), d["ms_per_step"])
PY
