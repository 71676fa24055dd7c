python -m pytest tests/test_hip_training.py -x -q -m gpu -k "side_stream or train_step or data_parallel or reference_trainer" > gpurun_out/tr_tests.txt 2>&1 || { grep -n "Error\|assert\|FAILED" gpurun_out/tr_tests.txt | head -20; exit 1; }
tail -1 gpurun_out/tr_tests.txt
for a in "" "" ; do
  python bench.py --mode train --no-cpu-baseline --steps 100 --warmup 10 $a > gpurun_out/tr.json 2>gpurun_out/tr.err || { tail -5 gpurun_out/tr.err; exit 1; }
  python - "$a" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/tr.json").read().strip().splitlines()[-1])
print(repr(sys.argv[1]), d["value"], d["ms_per_step"])
PY
done
