python -m pytest tests -x -q -m gpu -k "group8 or coresident or pipeline" > gpurun_out/dec2_tests.txt 2>&1 || { tail -20 gpurun_out/dec2_tests.txt; exit 1; }
tail -2 gpurun_out/dec2_tests.txt
i=0
for a in "" "--pipe-decoders 2 --pipe-depth 3" "--pipe-decoders 2 --pipe-depth 4" "--pipe-decoders 2 --pipe-encoders 2 --pipe-depth 4" "--pipe-decoders 3 --pipe-depth 5"; do
  i=$((i+1))
  python bench.py --no-cpu-baseline --steps 300 $a > gpurun_out/dec2_$i.json 2>gpurun_out/dec2_$i.err || { tail -5 gpurun_out/dec2_$i.err; exit 1; }
  python - "$a" gpurun_out/dec2_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(repr(sys.argv[1]), d["value"], d["ms_per_step"], d["config"].get("ids_check_pipelined"), d["config"].get("batch_pipeline"))
PY
done
