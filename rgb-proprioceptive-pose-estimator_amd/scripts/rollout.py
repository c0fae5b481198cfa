#!/usr/bin/env python
"""Rollout entry point (model side) on the MI355X path.

The reference's scripts/rollout.py drives a Robosuite simulator and renders video (rollout.py:141-187,
util/learn_utils.py:258-539); the simulator is out of scope, so this script keeps what belongs to the model:
the constructor flags, loading a reference-format state_dict checkpoint (rollout.py:193), the fixed seeds
(np/torch = 3, rollout.py:50-51), `model.eval(); model.rollout = True; model.reset_initial_state(1)`
(learn_utils.py:322-323,342), one call per timestep with the LSTM state carried on the module
(learn_utils.py:446), the per-step error print-out and the `model_outputs.npy` dump.  Frames come from a
seeded synthetic episode instead of `env.step`.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from rgb_proprioceptive_pose_estimator_amd.scripts.train_model import DTYPES, build_model, build_parser  # noqa: E402


def main(argv=None):
    p = build_parser()
    p.add_argument("--model_path", type=str, default=None, help="state_dict .pth to load (reference key names)")
    p.add_argument("--n_episodes", type=int, default=10)
    p.add_argument("--out", type=str, default="model_outputs.npy")
    p.add_argument("--no_graph", action="store_true", help="launch every frame eagerly instead of replaying one captured hipGraph "
                   "(a frame is ~90 launches on one stream; replay: 0.45 ms, eager: 1.05-1.27 ms at batch 1 -- profiles/r03_rollout_latency.txt)")
    args = p.parse_args(argv)
    from rgb_proprioceptive_pose_estimator_amd.models import PoseDistanceLoss
    from rgb_proprioceptive_pose_estimator_amd.util.data_utils import synthetic_batch

    np.random.seed(3)
    torch.manual_seed(3)
    if not torch.cuda.is_available():
        raise SystemExit("rollout.py: no MI355X visible; this path has no CPU fallback")
    model = build_model(args, DTYPES[args.dtype])
    if args.model_path:
        model.load_state_dict(torch.load(args.model_path, map_location=torch.device("cpu")))
    model.cuda().eval()
    model.rollout = True
    val = PoseDistanceLoss(mode="val")
    outs, pos_errs, ori_errs = [], [], []
    two_arm = not hasattr(model, "object_name")
    frame = None   # the captured frame (util.learn_utils.GraphedRolloutFrame): built from the first frame's tensors
    with torch.no_grad():
        for ep in range(args.n_episodes):
            model.reset_initial_state(1)
            ep_b = synthetic_batch((args.horizon, 1), 3 + ep, with_depth=args.use_depth, noise_scale=args.noise_scale)
            for t in range(args.horizon):
                if model.requires_sequence:
                    img, x0bar = ep_b["img"][t:t + 1], ep_b["x0bar"][t:t + 1]
                    depth = None if ep_b["depth"] is None else ep_b["depth"][t:t + 1]
                else:
                    img, x0bar = ep_b["img"][t], ep_b["x0bar"][t]
                    depth = None if ep_b["depth"] is None else ep_b["depth"][t]
                if frame is None and not args.no_graph:
                    from rgb_proprioceptive_pose_estimator_amd.util.learn_utils import GraphedRolloutFrame
                    frame = GraphedRolloutFrame(model, img.cuda(), None if depth is None else depth.cuda(), x0bar.cuda())
                    model.reset_initial_state(1)   # (capture and warm-up frames advanced the carried LSTM state)
                out = model(img, depth, x0bar) if frame is None else frame(img, depth, x0bar)
                out = out[-1] if isinstance(out, tuple) else out
                truth = (ep_b["x1"] if two_arm else ep_b["obj"])[t].reshape(out.shape)
                pe, oe = val(out, truth)
                pos_errs.append(float(pe)), ori_errs.append(float(oe))
                outs.append(out.reshape(7).cpu().numpy())
            print("episode {}: mean pos err {:.4f} m, mean ori err {:.4f} rad".format(ep, np.mean(pos_errs[-args.horizon:]), np.mean(ori_errs[-args.horizon:])))
    np.save(args.out, np.stack(outs))
    print("Mean pos err {:.4f} (std {:.4f}), mean ori err {:.4f} (std {:.4f})".format(np.mean(pos_errs), np.std(pos_errs), np.mean(ori_errs), np.std(ori_errs)))


if __name__ == "__main__":
    main()
