# same-box A/B of two builds of the library: bash profiles/micro/ab_lib.sh <other.so> <bench args...>
other=$1; shift
for i in 1 2; do
python bench.py "$@" --no-cpu-baseline > gpurun_out/ab_a.json 2> gpurun_out/ab_a.err
python -c "import json; d=json.loads(open('gpurun_out/ab_a.json').read().strip().splitlines()[-1]); print('tree ', d['value'], d['ms_per_step'])"
python profiles/micro/bench_with_lib.py $other "$@" --no-cpu-baseline > gpurun_out/ab_b.json 2> gpurun_out/ab_b.err
python -c "import json; d=json.loads(open('gpurun_out/ab_b.json').read().strip().splitlines()[-1]); print('other', d['value'], d['ms_per_step'])"
done
