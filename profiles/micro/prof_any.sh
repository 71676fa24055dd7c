# usage: bash profiles/micro/prof_any.sh <tag> <python script and args...>   (kernel stats of a short script)
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python $SCRIPT "$@" > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
test -n "$f" && python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:80]:80s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
