set -o pipefail
out=gpurun_out/r6k; mkdir -p $out; root=$PWD; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace -d $root/$out/tr -o t -- python3 $root/tools/ksq_probe.py > $root/$out/probe.log 2>&1) || exit 1
python3 - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/r6k/tr/**/*_results.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
for name, start, end, grid in cur.execute("select name, start, end, grid_x from kernels where name like '%k_ksq%' order by start"):
    print("k_ksq grid %6d waves %5d: %.1f us" % (grid, grid // 64, (end - start) / 1e3))
PY
rm -rf $out/tr
