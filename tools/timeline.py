"""Diagnostic: timeline of ONE train step from a rocprofv3 --kernel-trace csv: per-queue busy time, overlap, gaps, and the
main-stream critical path by kernel symbol.  usage: python tools/timeline.py <kernel_trace.csv> [step_index] [--list]
(--list: every launch of the step in start order: start ms, queue M/S, duration us, gap to the previous launch on its queue, grid, symbol)"""
import csv
import sys
import os
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_reduce import symbol

csv.field_size_limit(1 << 30)
LIST = "--list" in sys.argv
if LIST:
    sys.argv.remove("--list")
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), symbol(r["Kernel_Name"]), r["Queue_Id"],
             "%sx%s" % (int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)), r.get("Grid_Size_Y", "1"))) for r in rows)
adam = [i for i, k in enumerate(ks) if k[2] == "adam_kernel"]
step = int(sys.argv[2]) if len(sys.argv) > 2 else -5   # a timed step: negative = counted from the end (bench.py ends with 3 event-bracketed steps; pre-conditioning and warm-up steps come first)
if step < 0:
    step += len(adam)
lo, hi = adam[step - 1] + 1, adam[step] + 1
seg = ks[lo:hi]
t0, t1 = seg[0][0], max(k[1] for k in seg)
print("step %d: %d kernels, wall %.3f ms" % (step, len(seg), (t1 - t0) / 1e6))
byq = defaultdict(list)
for k in seg:
    byq[k[3]].append(k)


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


allbusy = union([(k[0], k[1]) for k in seg])
print("  any-queue busy %.3f ms, idle %.3f ms" % (allbusy / 1e6, (t1 - t0 - allbusy) / 1e6))
for q, lst in byq.items():
    b = union([(k[0], k[1]) for k in lst])
    print("  queue %s: %d kernels, busy %.3f ms, sum of durations %.3f ms" % (q, len(lst), b / 1e6, sum(k[1] - k[0] for k in lst) / 1e6))
qs = sorted(byq, key=lambda q: -len(byq[q]))
if len(qs) > 1:
    main, side = byq[qs[0]], byq[qs[1]]
    ub = union([(k[0], k[1]) for k in main]) + union([(k[0], k[1]) for k in side]) - union([(k[0], k[1]) for k in main + side])
    print("  main/side overlapped time %.3f ms" % (ub / 1e6))
    # phase split: the forward ends with the loss kernel (the second queue also carries the projection shortcuts of the forward)
    loss = [k for k in main if k[2] == "pose_loss_kernel"]
    s0 = loss[0][1] if loss else min(k[0] for k in side)
    s1 = max(k[1] for k in side)
    print("  forward (start -> loss kernel) %.3f ms; backward up to the last side kernel %.3f ms; tail after it %.3f ms" %
          ((s0 - t0) / 1e6, (s1 - s0) / 1e6, (t1 - s1) / 1e6))
    # main-queue kernels during the backward span, by symbol
    agg = defaultdict(lambda: [0, 0.0])
    for k in main:
        ph = "fwd" if k[1] <= s0 else "bwd"
        a = agg[(ph, k[2])]
        a[0] += 1
        a[1] += (k[1] - k[0]) / 1e6
    for k in side:
        a = agg[("side-fwd" if k[1] <= s0 else "side", k[2])]
        a[0] += 1
        a[1] += (k[1] - k[0]) / 1e6
    for ph in ("fwd", "side-fwd", "bwd", "side"):
        tot = sum(v[1] for kk, v in agg.items() if kk[0] == ph)
        print("  -- %s: %.3f ms of kernel time" % (ph, tot))
        for kk, v in sorted(agg.items(), key=lambda x: -x[1][1]):
            if kk[0] == ph and v[1] > 0.08:
                print("       %-42s x%3d %7.3f ms" % (kk[1][:42], v[0], v[1]))
    # gaps on the main queue
    main_s = sorted(main)
    gaps = [(main_s[i + 1][0] - main_s[i][1]) / 1e3 for i in range(len(main_s) - 1)]
    pos = [g for g in gaps if g > 0]
    print("  main-queue gaps: %d, total %.3f ms, median %.2f us, >20us: %d" %
          (len(pos), sum(pos) / 1e3, sorted(pos)[len(pos) // 2] if pos else 0, sum(1 for g in pos if g > 20)))
if LIST:
    last = {}
    mq = qs[0]
    for k in seg:
        g = (k[0] - last[k[3]]) / 1e3 if k[3] in last else 0.0
        last[k[3]] = k[1]
        print("%8.3f %s %8.1f us  gap %6.1f  %-10s %s" % ((k[0] - t0) / 1e6, "M" if k[3] == mq else "S", (k[1] - k[0]) / 1e3, g, k[4], k[2]))
