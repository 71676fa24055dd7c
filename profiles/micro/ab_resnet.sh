# same-box A/B of bench.py --mode resnet: trunk kernel flags (0x20000 = no LDS-patch 3x3 kernel, 0x40000 = 128-column tiles everywhere)
for f in 0 0x20000 0x40000 0x60000; do
python bench.py --mode resnet --no-cpu-baseline --resnet-flags $f > gpurun_out/b_resnet_ab.json 2> gpurun_out/b_resnet_ab.err
python -c "
import json,sys; d=json.loads(open('gpurun_out/b_resnet_ab.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done
