# the bench lines only (no profiler): default run, the driver's 20 / 5 setting, every --mode at its defaults
OUT=gpurun_out/lines_$1
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err || exit 1
for m in beam train resnet preprocess metrics; do
  python bench.py --mode $m > $OUT/bench_$m.json 2> $OUT/bench_$m.err || exit 1
done
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["unit"], d["ms_per_step"], d.get("value_serial") or d["config"].get("value_serial"), d.get("value_cold_start"))
PY
