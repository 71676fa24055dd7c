import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("value_serial"), d.get("value_pipelined"), d["config"].get("batch_pipeline"))
print(" ".join(f'{s["kernel"]}={s["ms"]}' for s in d["roofline"]["stages"]))
