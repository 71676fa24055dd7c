# usage: bash profiles/micro/run_ab.sh <mode> [extra bench args]: two runs of bench.py --mode <mode>, prints value and ms per step
m=$1; shift
for i in 1 2; do
  python bench.py --mode $m --no-cpu-baseline "$@" > gpurun_out/ab.json 2>gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
  python - <<'PY'
import json
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("value_serial") or d["config"].get("value_serial"))
PY
done
