#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash profiles/collect.sh r01 v3'
# Writes gpurun_out/prof_<tag>/{stats,fetch,write}/ ; profiles/summarize.py turns them into the committed files.
# Counters are collected in their own passes (never together with a trace domain other than kernel-trace).
set -e
ROUND=${1:-r01}; TAG=${2:-v3}
OUT=gpurun_out/prof_${ROUND}_${TAG}
export TMPDIR=/tmp
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python bench.py --steps 3 --warmup 1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- python bench.py --steps 3 --warmup 1 > $OUT/write.log 2>&1
# matrix-pipe busy cycles (summed over all SIMDs) and the GPU-active clock count (summed over the 8 XCDs)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o p -- python bench.py --steps 3 --warmup 1 > $OUT/mfma.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma_train -o p -- python bench.py --mode train --steps 3 --warmup 1 > $OUT/mfma_train.log 2>&1
# the secondary configs (BASELINE configs[2..4]): bench lines with their own roofline / cpu_baseline, per-kernel stats,
# matrix-pipe counters of the ResNet trunk
for m in beam train resnet preprocess metrics; do
  python bench.py --mode $m > $OUT/bench_$m.json 2> $OUT/bench_$m.err      # default steps / warm-up: the pipelined modes need ~25 batches to reach their steady state
done
for m in beam train resnet; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$m -o p -- python bench.py --mode $m --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats_$m.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma_resnet -o p -- python bench.py --mode resnet --steps 3 --warmup 1 --no-cpu-baseline > $OUT/mfma_resnet.log 2>&1
find $OUT -name "*.csv" | head -40
