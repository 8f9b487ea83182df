#!/usr/bin/env python3
"""table of tools/batch_sweep.py lines: tools/sweep_table.py gpurun_out/r6b/knobs_*.json  (ms per call, queued; fingerprints compared)"""
import json, sys
files = sorted(sys.argv[1:], key=lambda x: int(x.split('_')[-1].split('.')[0]))
tab, fps = {}, {}
for f in files:
    d = json.load(open(f))
    tag = d['tag'] + "  #" + f.split('_')[-1].split('.')[0]
    tab[tag] = {(r['n'], r['k']): r['ms_queued'] for r in d['rows']}
    for r in d['rows']:
        fps.setdefault((r['n'], r['k']), set()).add(tuple(r.get('fp', [])))
keys = sorted(next(iter(tab.values())).keys(), key=lambda t: (t[1], t[0]))
print("%-86s" % "cfg" + "".join("%9s" % ("%d%s" % (n, "" if k == 1 else "k%d" % k)) for n, k in keys))
for t, v in tab.items():
    print("%-86s" % t + "".join("%9.2f" % v[x] if x in v else "%9s" % "-" for x in keys))
bad = [k for k, v in fps.items() if len(v) > 1]
print("fingerprints differ between settings at:", bad if bad else "none")
