#!/bin/bash
# VGPRs / spilled VGPRs / scratch per kernel and device function of zkp_coop.hip (extra hipcc flags as arguments)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" \
    "$(dirname "$0")/../zkvm_pairings_amd/csrc/zkp_coop.hip" -o /tmp/zkp_regs.o 2>&1 |
python3 -c '
import re, sys
name = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        name = m.group(1); rows[name] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
    if m and name:
        rows[name][m.group(1)] = int(m.group(2))
import subprocess
for n, r in rows.items():
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    d = re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0]
    print("%-28s vgpr %3d  spill %3d  scratch %4d  occ %d" % (d[:28], r.get("VGPRs", -1), r.get("VGPRs Spill", -1), r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1)))
'
