"""smoke(): one tiny train step of the hot path on cuda:0, checked against the oracle (the CPU restatement
under oracle/, used here only as the checker)."""
import os
import sys

import torch


def smoke_step():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import pose_oracle as po  # checker only

    from . import models as M
    from .optim import FusedAdam

    if not torch.cuda.is_available():
        raise RuntimeError("smoke(): no GPU visible -- the HIP path has no CPU fallback")
    torch.cuda.set_device(0)
    kind, cfg, lead = "no", dict(latent_dim=64, hidden=[32, 16], use_depth=False, no_proprioception=False), (2,)
    sd = po.make_state(kind, cfg, 11)
    batch = po.synth_batch(lead, 102)
    ref = po.train_step(kind, cfg, {k: v.clone() for k, v in sd.items()}, batch, dict(metric="combined", scale=1.0, alpha=0.5, mode="pose"), {})
    for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, 5e-2)):
        model = M.NaiveObjectStateEstimator("cube", [32, 16], 50, 64, False, (9,), False, False, False, compute_dtype=dtype)
        model.load_state_dict({k: v for k, v in sd.items()})
        model.cuda().train()
        opt = FusedAdam(model.parameters(), lr=1e-3)
        crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
        out = model(batch["img"].cuda(), None, batch["x0bar"].cuda())
        loss = crit(out, batch["obj"].cuda())
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        err = ((out.detach().cpu() - ref["outputs"]).abs().max() / ref["outputs"].abs().max()).item()
        lerr = abs(loss.item() - ref["loss"].item()) / abs(ref["loss"].item())
        print("smoke[%s]: pose rel err %.3e, loss %.6f vs oracle %.6f (rel %.2e)" % (str(dtype).split(".")[-1], err, loss.item(), ref["loss"].item(), lerr))
        if not (err < tol and lerr < tol):
            raise AssertionError("smoke: HIP %s path deviates from the oracle (pose %.3e, loss %.3e, tol %.1e)" % (dtype, err, lerr, tol))
    return True
