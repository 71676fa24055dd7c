set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ramp3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python $GRAFT_REPO_ROOT/profiles/micro/ramp3.py > $OUT/log.txt 2>&1
ls $OUT
