"""Reduce a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (with --kernel-trace) over bench.py to per-kernel-symbol
matrix-core utilisation -> profiles/r02_mfma_busy_pmc.json.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/mfma -o m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/mfma_reduce.py gpurun_out/mfma/m_counter_collection.csv > profiles/r02_mfma_busy_pmc.json

SQ_VALU_MFMA_BUSY_CYCLES counts, summed over all SIMDs of the chip, the cycles a SIMD's matrix pipe is busy: 16 per 16x16x32 bf16
instruction (checked against the FLOP count of the 3x3 forward launches: 4.93e12 FLOP / 16384 per instruction = 3.01e8 instructions,
4.817e9 busy cycles).  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (8 x the launch's cycles).  busy_frac = MFMA_BUSY /
(GUI_ACTIVE / 8 * 1024 SIMDs): the share of the chip's matrix-pipe cycles that were busy while the kernel ran -- in the step, i.e. beside
whatever the other stream ran."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_reduce import symbol

csv.field_size_limit(1 << 30)
busy, act, n = defaultdict(float), defaultdict(float), defaultdict(set)
with open(sys.argv[1], newline="") as f:
    for row in csv.DictReader(f):
        s = symbol(row["Kernel_Name"])
        v = float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[s] += v
            n[s].add(row["Dispatch_Id"])
        elif row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[s] += v
out = []
for s in busy:
    if busy[s] <= 0 or act[s] <= 0:
        continue
    out.append({"kernel": s, "launches": len(n[s]), "mfma_busy_cycles": busy[s], "gui_active_cycles": act[s],
                "busy_frac": round(busy[s] * 8 / (act[s] * 1024), 4)})
out.sort(key=lambda r: -r["mfma_busy_cycles"])
print(json.dumps({"note": __doc__.split("\n\n")[-1].replace("\n", " "), "kernels": out}, indent=1))
