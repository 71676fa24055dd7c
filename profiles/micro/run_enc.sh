for n in 1 2; do python bench.py --no-cpu-baseline --steps 200 --pipe-encoders $n > gpurun_out/enc_greedy_$n.json 2>gpurun_out/enc_greedy_$n.err || exit 1; done
for n in 1 2 3; do python bench.py --mode resnet --no-cpu-baseline --steps 200 --pipe-encoders $n > gpurun_out/enc_resnet_$n.json 2>gpurun_out/enc_resnet_$n.err || exit 1; done
python - <<'PY'
import json
for m in ("greedy","resnet"):
    for n in (1,2,3):
        try: d=json.loads(open(f"gpurun_out/enc_{m}_{n}.json").read().strip().splitlines()[-1])
        except Exception as e: continue
        print(m,n,d["value"],d["ms_per_step"],d["config"].get("value_serial"),d["config"].get("ids_check_pipelined"))
PY
