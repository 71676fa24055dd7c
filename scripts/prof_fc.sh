export TMPDIR=/tmp
for d in 0 1 2 4 5 7; do
  I2L_DBG=$d rocprofv3 --kernel-trace --stats -d gpurun_out/pf$d -o p -- python scripts/fc.py > /dev/null 2>&1
done
python - <<'PY'
import sqlite3
for d in (0,1,2,4,5,7):
    c=sqlite3.connect(f'gpurun_out/pf{d}/p_results.db')
    for r in c.execute("select name,total_calls,average from top_kernels limit 3"):
        if 'linear' in r[0]: print(d, r[0][:50], r[1], round(r[2],1))
PY
