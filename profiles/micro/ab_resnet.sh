for i in 1; do
for f in 0 0x20000 0x40000 0x60000; do
python bench.py --mode resnet --no-cpu-baseline --resnet-flags $f > gpurun_out/b_resnet_$f.json 2> gpurun_out/b_resnet_$f.err
python -c "
import json,sys; d=json.loads(open('gpurun_out/b_resnet_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done; done
