"""Diagnostic: the head region of one train step (between the trunk's average pool forward and backward) from a rocprofv3
--kernel-trace csv: kernels by symbol, total time, gaps.  usage: python tools/head_region.py <kernel_trace.csv> [step]"""
import csv
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_reduce import symbol

csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), symbol(r["Kernel_Name"])) for r in rows)
fw = [i for i, k in enumerate(ks) if k[2].startswith("avgpool_fwd")]
bw = [i for i, k in enumerate(ks) if k[2].startswith("avgpool_bwd")]
step = int(sys.argv[2]) if len(sys.argv) > 2 else len(fw) - 2
i0 = fw[step]
i1 = [j for j in bw if j > i0][0]
h = ks[i0:i1 + 1]
wall = (h[-1][1] - h[0][0]) / 1e3
busy = sum(k[1] - k[0] for k in h) / 1e3
print("head region of step %d: %d kernels, wall %.1f us, kernel time %.1f us, gaps %.1f us" % (step, len(h), wall, busy, wall - busy))
agg = defaultdict(lambda: [0, 0.0])
for k in h:
    agg[k[2]][0] += 1
    agg[k[2]][1] += (k[1] - k[0]) / 1e3
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-60s x%3d %8.1f us" % (name[:60], n, t))
