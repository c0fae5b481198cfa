"""Diagnostic: per-kernel durations and gaps of one rollout frame from a rocprofv3 --kernel-trace csv of tools/rollout_frames.py.
usage: python tools/rollout_timeline.py <kernel_trace.csv> [frame_index]"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_reduce import symbol

csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), symbol(r["Kernel_Name"]),
             "%sx%sx%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))) for r in rows)
# frames are cut at the max-pool launch (once per frame): a printed "frame" runs from one max-pool to the next
starts = [i for i, k in enumerate(ks) if "maxpool" in k[2]]
f = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 3
lo, hi = starts[f], starts[f + 1]
seg = ks[lo:hi]
wall = ks[hi][0] - seg[0][0]
busy = sum(k[1] - k[0] for k in seg)
print("frame %d: %d kernels, wall (start to next frame's start) %.1f us, kernel time %.1f us, gaps %.1f us" % (
    f, len(seg), wall / 1e3, busy / 1e3, (wall - busy) / 1e3))
prev = None
for k in seg:
    gap = (k[0] - prev) / 1e3 if prev else 0.0
    print("  %6.1f us  (+%5.1f gap)  %-52s grid %s" % ((k[1] - k[0]) / 1e3, gap, k[2][:52], k[3]))
    prev = k[1]
