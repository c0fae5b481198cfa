"""Isolated timing of the y3-free conv3 forward (rpe_conv1x1_fwd_bn) at the benchmark's layer-1 / layer-2 shapes: the row-streaming kernel
(csrc/stream1x1.hip) against the tiled role-5 launch (RPE_NO_STREAM1X1=1 in a second process).   python tools/bench_stream1x1.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rgb_proprioceptive_pose_estimator_amd import ops

DEV = "cuda"
for (b, h, ci, co) in [(256, 56, 64, 256), (256, 28, 128, 512)]:
    rows = b * h * h
    x = torch.relu(torch.randn(rows, ci, device=DEV)).to(torch.bfloat16).reshape(b, h, h, ci)
    w = (torch.randn(co, ci, device=DEV) / ci ** 0.5).to(torch.bfloat16)
    res = torch.randn(rows, co, device=DEV).to(torch.bfloat16).reshape(b, h, h, co)
    scale, shift = torch.rand(co, device=DEV) + 0.5, torch.randn(co, device=DEV) * 0.3
    big = torch.empty(600 * 1000 * 1000, dtype=torch.uint8, device=DEV)
    mb = (rows * ci * 2 + rows * co * 2 * 2 + rows * co / 8) / 1e6
    for rep in range(2):
        ts = []
        for i in range(6):
            big.zero_()                                  # cold caches
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.conv1x1_fwd_bn(x, w, scale, shift, res)
            e.record()
            e.synchronize()
            ts.append(a.elapsed_time(e) * 1e3)
        ts = sorted(ts[1:])
        print("%s %dx%dx%d %d->%d: %.1f us (median of 5, cold caches)  %.2f TB/s" % (ops.last_kernel_name(), b, h, h, ci, co, ts[2], mb / ts[2]), flush=True)
