import contextlib, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from rgb_proprioceptive_pose_estimator_amd import models as M
from rgb_proprioceptive_pose_estimator_amd.util import learn_utils as LU
MODELS = {
    "no": lambda dt: M.NaiveObjectStateEstimator("cube", [1024, 256, 64], 50, 512, False, (9,), False, False, False, compute_dtype=dt),
    "tdo": lambda dt: M.TemporallyDependentObjectStateEstimator("hammer", 512, 50, 512, 10, 0.1, False, (9,), False, False, False, compute_dtype=dt),
}
for name in ("no", "tdo"):
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = MODELS[name](torch.bfloat16)
    model.cuda().eval(); model.rollout = True; model.reset_initial_state(1)
    seq = getattr(model, "requires_sequence", False)
    img = torch.randn((1, 1, 3, 224, 224) if seq else (1, 3, 224, 224), device="cuda")
    x0 = torch.randn((1, 1, 7) if seq else (1, 7), device="cuda")
    # replicate GraphedRolloutFrame with debug mode
    with torch.no_grad():
        for _ in range(3): model(img, None, x0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        out = model(img, None, x0)
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize(); print(name, "replay only: %.3f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); print(name, "host time per replay call: %.3f ms" % (th / 200 * 1e3))

    # the same frame eagerly, per-call host time and device time
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): model(img, None, x0)
        th = time.perf_counter() - t0
        torch.cuda.synchronize(); td = time.perf_counter() - t0
    print(name, "eager: host %.3f ms, total %.3f ms per frame" % (th / 200 * 1e3, td / 200 * 1e3))
