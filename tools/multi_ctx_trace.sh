#!/bin/bash
# How far apart do the contexts of zkp_pairing_batch_multi start?  Builds integration/c/zkp_multi.c, runs it with three contexts on
# this GPU under rocprofv3 --kernel-trace and prints, per HIP stream (= per context), when its first pipeline kernel of each
# multi-context call started.  From page-locked arrays no context's start waits for another context's copy.
#     bash tools/multi_ctx_trace.sh [pairs]      -> gpurun_out/multi_ctx_trace.txt
set -e -o pipefail
root=$PWD
n=${1:-393216}
gcc -O2 -I include integration/c/zkp_multi.c -L zkvm_pairings_amd -lzkp_pairings -Wl,-rpath,$root/zkvm_pairings_amd -o /tmp/zkp_multi
export TMPDIR=/tmp
rm -rf $root/gpurun_out/mct; mkdir -p $root/gpurun_out/mct
(cd /tmp && rocprofv3 --kernel-trace -d $root/gpurun_out/mct -o t -- /tmp/zkp_multi 3 $n > $root/gpurun_out/mct/run.log 2>&1)
python3 - "$(find $root/gpurun_out/mct -name '*_results.db' | head -n 1)" <<'PY' | tee $root/gpurun_out/multi_ctx_trace.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = list(cur.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")))
print("# columns of the kernels view:", cols)
# a multi-context call = a burst of k_prep_lines on three queues; group prep kernels that start within 20 ms of each other
preps = [(s, q) for (n, s, e, q) in rows if "k_prep_lines" in n]
calls, cur_call = [], []
for s, q in preps:
    if cur_call and s - cur_call[0][0] > 150e6:
        calls.append(cur_call); cur_call = []
    cur_call.append((s, q))
if cur_call:
    calls.append(cur_call)
for i, c in enumerate(calls):
    first = {}
    for s, q in c:
        first.setdefault(q, s)
    t0 = min(first.values())
    print("call %d: first k_prep_lines per %s, ms after the earliest: %s" % (i, qcol, ", ".join("%s: %.3f" % (q, (s - t0) / 1e6) for q, s in sorted(first.items(), key=lambda kv: kv[1]))))
    # is the GPU ever idle between the contexts' pipelines?  union of the kernel intervals from this call's first prep kernel
    # to the last kernel that starts before the next call
    t1 = calls[i + 1][0][0] if i + 1 < len(calls) else float("inf")
    iv = sorted((s, e) for (n, s, e, q) in rows if t0 <= s < t1)
    busy, hi = 0, iv[0][0]
    for s, e in iv:
        if e > hi:
            busy += e - max(s, hi); hi = e
    print("        kernels cover %.2f of the %.2f ms from the first to the last kernel of the call (%.1f %%)" % (busy / 1e6, (hi - t0) / 1e6, 100.0 * busy / (hi - t0)))
PY
cat $root/gpurun_out/mct/run.log | tail -4 >> $root/gpurun_out/multi_ctx_trace.txt
rm -rf $root/gpurun_out/mct
