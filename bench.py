#!/usr/bin/env python
"""Benchmark of the hot path: images/s of one full TRAIN STEP (stage image -> ResNet-50 trunk -> aux head ->
proprio-fusion MLP -> PoseDistanceLoss (+ on-device val metrics) -> backward -> gradient all-reduce -> Adam)
of NaiveObjectStateEstimator at 224x224, 256 images per GPU, bf16 compute / fp32 accumulate / fp32 masters
(BASELINE.json configs[1]); synthetic Robosuite-shaped data resident in HBM, random-init weights.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel family of the step (HIP-event timed in
a separate profiled pass after the timed region); `cpu_baseline` is the oracle (a CPU port of the reference
path) timed on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import ctypes
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CATS = ["conv_fwd", "conv_dgrad", "conv_wgrad", "bn_fwd", "bn_bwd", "other"]
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
PRECONDITION_MIN_S, PRECONDITION_QUIET_S, PRECONDITION_MAX_S = 3.0, 2.0, 15.0   # untimed train steps before the W warm-up steps (reported in the line as `precondition_s`)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def usable_cores():
    """cores this process may actually use: affinity mask capped by the cgroup CPU quota (a container on a big host)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


C1_CFG = dict(latent_dim=512, hidden=[1024, 256, 64], use_depth=False, no_proprioception=False)
LOSS_CFG = dict(metric="combined", scale=1.0, alpha=0.5, mode="pose")


def cpu_baseline(seconds_budget=25.0):
    """Oracle train step (fp32 torch-CPU ops) of the same model family: config C1 = 32 images, NO model.  Also returns the
    oracle's pose outputs of the FIRST step (pristine seeded weights) -- the reference side of `pose_parity`."""
    from oracle import pose_oracle as po
    torch.set_num_threads(usable_cores())
    log("[bench] cpu baseline on %d threads (os.cpu_count=%s, affinity=%d)" % (torch.get_num_threads(), os.cpu_count(), len(os.sched_getaffinity(0))))
    sd = po.make_state("no", C1_CFG, 0)
    batch = po.synth_batch((32,), 1234)
    opt = {}
    first = po.train_step("no", C1_CFG, sd, batch, LOSS_CFG, opt)  # warm-up; its outputs belong to the pristine weights
    n, t0 = 0, time.time()
    while n < 3 or (time.time() - t0 < seconds_budget and n < 8):
        po.train_step("no", C1_CFG, sd, batch, LOSS_CFG, opt)
        n += 1
    dt = (time.time() - t0) / n
    return {"value": round(32 / dt, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d timed train steps (1 warm-up) of 32 images, fp32, NaiveObjectStateEstimator latent 512 hidden [1024,256,64], oracle/pose_oracle.py" % n}, first


def pose_parity(ref, dtypes, dev):
    """The second half of BASELINE.json's metric ("pose L2 vs CPU ref"): the HIP path and the CPU oracle on identical seeded
    weights and inputs (config C1: 32 images, train-mode forward = step 1 of a run), per compute dtype.  Position L2 in the units
    of the pose (metres), relative error over all 7 outputs against the largest reference magnitude."""
    from oracle import pose_oracle as po
    from rgb_proprioceptive_pose_estimator_amd import models as M
    sd = po.make_state("no", C1_CFG, 0)
    batch = po.synth_batch((32,), 1234)
    want = ref["outputs"]
    out = {}
    for name, dtype in dtypes:
        with contextlib.redirect_stdout(sys.stderr):
            m = M.NaiveObjectStateEstimator("cube", [1024, 256, 64], 50, 512, False, (9,), False, False, False, compute_dtype=dtype)
        m.load_state_dict(sd)
        m.to(dev).train()
        got = m(batch["img"].to(dev), None, batch["x0bar"].to(dev)).detach().float().cpu()
        crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
        loss = float(crit(got.to(dev), batch["obj"].to(dev)).item())
        l2 = (got[:, :3] - want[:, :3]).norm(dim=1)
        out[name] = {"pos_l2_max": float(l2.max()), "pos_l2_mean": float(l2.mean()),
                     "rel_err_max_7": float((got - want).abs().max() / want.abs().max()),
                     "loss": loss, "loss_ref": float(ref["loss"].item())}
        del m
    torch.cuda.empty_cache()
    return {"config": "C1: 32 images 224x224, NaiveObjectStateEstimator latent 512 hidden [1024,256,64], seeded weights + inputs, train-mode forward",
            "reference": "oracle/pose_oracle.py (fp32 CPU restatement, pinned to the reference's vectors: tests/golden/model_no_c1.npz)",
            "tolerance": {"f32": 1e-4, "bf16": 2e-2, "f16": 4e-3}, **out}


def time_f32_path(dev, batch_size):
    """Step time of the exact-fp32 compute path (the one that meets the 1e-4 pose bar), same workload, a few steps."""
    from rgb_proprioceptive_pose_estimator_amd import models as M
    from rgb_proprioceptive_pose_estimator_amd.optim import FusedAdam
    from rgb_proprioceptive_pose_estimator_amd.util.data_utils import synthetic_batch
    from rgb_proprioceptive_pose_estimator_amd.util.learn_utils import GraphedTrainStep, train_step
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        m = M.NaiveObjectStateEstimator("cube", [1024, 256, 64], 50, 512, False, (9,), False, False, False, compute_dtype=torch.float32)
    m.to(dev).train()
    criterion = {"obj_loss": M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose"), "val_loss": M.PoseDistanceLoss(mode="val")}
    opt = FusedAdam(m.parameters(), lr=1e-3)
    b = synthetic_batch((batch_size,), 1234, device=dev)
    batch = (b["img"], None, b["x0bar"], b["x0"], None, b["obj"])
    for _ in range(2):
        train_step(m, batch, criterion, opt, True, "train", None)
    torch.cuda.synchronize()
    n, t0 = 5, time.perf_counter()
    for _ in range(n):
        train_step(m, batch, criterion, opt, True, "train", None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    del m, opt
    torch.cuda.empty_cache()
    return {"ms_per_step": round(dt * 1e3, 2), "images_per_s": round(batch_size / dt, 1), "steps": n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precondition-min", type=float, default=PRECONDITION_MIN_S,
                    help="least seconds of untimed 10-step windows before the W warm-up steps")
    ap.add_argument("--model", default="no", choices=["no", "n", "td", "tdo", "tdo_v2"],
                    help="model family: no = NaiveObjectStateEstimator (BASELINE configs[1], the default and the metric's workload); td / tdo / tdo_v2 = the "
                         "sequence models of configs[2..4] at (S, N) = (4, batch/4), latent 512, hidden 512 (proprio hidden 64)")
    ap.add_argument("--resnet", type=int, default=50, choices=[18, 50, 101, 152], help="num_resnet_layers of the trunk (the metric's workload is 50; the others are "
                    "side records of import_resnet's remaining members, util/model_utils.py:130-136)")
    ap.add_argument("--depth-head", action="store_true", help="use_depth=True (configs[3]: TDO + auxiliary depth head)")
    ap.add_argument("--force-dist", action="store_true", help="N = 1 only: initialise RCCL with one rank and run the data-parallel path "
                    "(parameter broadcast, staged stream joins, bucketed SUM all-reduce) so that its cost on one GPU is measured")
    ap.add_argument("--graph", action="store_true", help="replay one captured hipGraph per step instead of issuing every launch (N = 1 only; measured SLOWER than eager issue here: the host already runs ahead of the device and the replay schedules the two-stream backward worse -- DESIGN.md section 6)")
    args = ap.parse_args()

    from rgb_proprioceptive_pose_estimator_amd import models as M
    from rgb_proprioceptive_pose_estimator_amd._lib import RPE_F32, lib
    from rgb_proprioceptive_pose_estimator_amd.dist import GradSync, broadcast_parameters, init_from_env
    from rgb_proprioceptive_pose_estimator_amd.optim import FusedAdam
    from rgb_proprioceptive_pose_estimator_amd.util.data_utils import synthetic_batch
    from rgb_proprioceptive_pose_estimator_amd.util.learn_utils import GraphedTrainStep, train_step
    import torch.distributed as dist

    if args.force_dist and args.gpus == 1 and "RANK" not in os.environ:
        os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": os.environ.get("MASTER_PORT", "29531"),
                           "RPE_DIST_FORCE_INIT": "1"})
    rank, world, local = init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]

    torch.manual_seed(0)
    dh = args.depth_head
    seq = args.model in ("td", "tdo", "tdo_v2")
    if seq and args.batch % 4:
        raise SystemExit("--model %s runs (S, N) = (4, batch/4) sequences: --batch must be a multiple of 4" % args.model)
    lead = (4, args.batch // 4) if seq else (args.batch,)
    builders = {
        "no": lambda: M.NaiveObjectStateEstimator("cube", [1024, 256, 64], args.resnet, 512, False, (9,), dh, False, False, compute_dtype=dtype),
        "n": lambda: M.NaiveEndEffectorStateEstimator([1024, 256, 64], [1024, 256, 64], args.resnet, 512, False, compute_dtype=dtype),
        "td": lambda: M.TemporallyDependentStateEstimator(512, 512, args.resnet, 512, 4, 0.1, False, (9,), dh, False, compute_dtype=dtype),
        "tdo": lambda: M.TemporallyDependentObjectStateEstimator("hammer", 512, args.resnet, 512, 4, 0.1, False, (9,), dh, False, False, compute_dtype=dtype),
        "tdo_v2": lambda: M.TemporallyDependentObjectStateEstimatorV2("robot1_eef", 512, 64, args.resnet, 512, 4, 0.1, False, (9,), dh, False, compute_dtype=dtype),
    }
    workloads = {
        "no": "NaiveObjectStateEstimator train step (BASELINE.json configs[1]): ResNet-50 trunk + bn1 aux head + proprio MLP [1024,256,64] + PoseDistanceLoss(combined, alpha 0.5) + Adam",
        "n": "NaiveEndEffectorStateEstimator train step: ResNet-50 trunk + pre / post measurement MLPs [1024,256,64] + 2 x PoseDistanceLoss + Adam",
        "td": "TemporallyDependentStateEstimator train step (BASELINE.json configs[2]): ResNet-50 trunk + bn1 aux head + pre / post LSTM(512) + 2 x PoseDistanceLoss + Adam, (S, N) = (4, batch/4)",
        "tdo": "TemporallyDependentObjectStateEstimator train step (BASELINE.json configs[3] per GPU): ResNet-50 trunk + bn1 aux (+ depth) head + LSTM(512) + fc + PoseDistanceLoss + Adam, (S, N) = (4, batch/4)",
        "tdo_v2": "TemporallyDependentObjectStateEstimatorV2 train step (BASELINE.json configs[4] per GPU): ResNet-50 trunk + bn1 aux head + image LSTM(512) + proprio LSTM(64) + fc + PoseDistanceLoss + Adam, (S, N) = (4, batch/4)",
    }
    with contextlib.redirect_stdout(sys.stderr):  # the constructor prints its feature width, as the reference does; stdout is the JSON line only
        model = builders[args.model]()
    model.cuda().train()
    train_obj_pose = hasattr(model, "object_name")     # the reference's own switch (util/learn_utils.py:58)
    mk = lambda: M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
    criterion = {"obj_loss": mk(), "x0_loss": mk(), "x1_loss": mk(), "val_loss": M.PoseDistanceLoss(mode="val")}
    use_graph = world == 1 and args.graph and not args.force_dist
    opt = FusedAdam(model.parameters(), lr=1e-3, capturable=use_graph)
    b = synthetic_batch(lead, 1234 + rank, with_depth=dh, device=dev)
    batch = (b["img"], b["depth"], b["x0bar"], b["x0"], b["x1"], b["obj"])

    # build the parameter arena, make replicas identical, attach the staged gradient reduction -- all before the first step
    model._materialize(dev)
    sync = None
    if world > 1 or args.force_dist:
        broadcast_parameters(model._arena.flat, list(model.buffers()))
        sync = GradSync(model._arena.grad, reduce_single=args.force_dist).attach(model)   # slices are all-reduced under the backward
    precondition_s = 0.0
    precondition_steps = 0
    if use_graph:
        # the whole step as ONE captured hipGraph (forward, loss, val metrics, backward on both streams, Adam); replayed per step
        graphed = GraphedTrainStep(model, criterion, opt, train_obj_pose, batch, warmup=min(3, max(1, args.warmup)))
        batch = graphed.static
        run_step = lambda: graphed(batch)
        for _ in range(max(0, args.warmup - graphed.warmup_steps)):
            run_step()
        _z = torch.zeros((), device=dev)
        torch.zeros(1, dtype=torch.bool, device=dev)[0] = _z != _z   # (loads the two small kernels of the timed loop's NaN test: see below)
    else:
        run_step = lambda: train_step(model, batch, criterion, opt, train_obj_pose, "train", sync)
        # Device pre-conditioning, BEFORE the W warm-up steps and outside every timed region: windows of 10 untimed steps (each closed by a
        # synchronize) for at least --precondition-min seconds and until the best window time has not improved by 0.3 % for
        # PRECONDITION_QUIET_S seconds (at most PRECONDITION_MAX_S): clocks, caches and the allocator settle within the first two windows
        # (33 -> 19.6 ms/step).  The window times go to stderr; `precondition_s` is reported in the line.  (The "first process on a box is
        # 10-25 % slow" of this round's early A/B records was NOT the device: see the warm-up loop below.)
        t_pre, best, t_best = time.perf_counter(), None, None
        while True:
            t_w = time.perf_counter()
            for _ in range(10):
                run_step()
            precondition_steps += 10
            torch.cuda.synchronize()
            now = time.perf_counter()
            w = now - t_w
            log("[bench] pre-conditioning window at %.1f s: %.2f ms/step" % (now - t_pre, w * 100))
            if best is None or w < best * 0.997:
                best, t_best = (w if best is None else min(best, w)), now
            stop = (now - t_best >= PRECONDITION_QUIET_S and now - t_pre >= args.precondition_min) or now - t_pre >= max(PRECONDITION_MAX_S, args.precondition_min)
            if world > 1 or args.force_dist:
                # every rank must run the SAME number of steps (each carries collectives): rank 0 decides for all
                flag = torch.tensor([1 if stop else 0], dtype=torch.int32, device=dev)
                dist.broadcast(flag, 0)
                stop = bool(flag.item())
            if stop:
                break
        precondition_s = time.perf_counter() - t_pre
        # the W warm-up steps run EXACTLY the statements of the timed loop -- incl. the NaN test of the loss value: its two small
        # kernels are loaded on first use (hipModuleLoad of the host framework's code object: ~10 ms in a process on a warm box, ~60 ms
        # in the first process after the box was leased or the repository copied), which inside a 20-step timed region read as
        # +0.5 / +3 ms per step (the "first process is slow" of profiles/r03_ab_*.txt; the pre-conditioning windows, which did not
        # contain those statements, ran at full speed in the same process: profiles/r03_bench_first_process.txt)
        warm_flags = torch.zeros(max(1, args.warmup), dtype=torch.bool, device=dev)
        for i in range(args.warmup):
            loss, _, _ = run_step()
            warm_flags[i] = loss != loss

    def fence():
        if world > 1 or args.force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    log("[bench] warm-up done; timing %d steps" % args.steps)
    fence()
    nan_flags = torch.zeros(args.steps, dtype=torch.bool, device=dev)   # NaN loss VALUES (see loss_note), tested on the device: no sync in the loop
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, _, _ = run_step()
        nan_flags[i] = loss != loss
    host_loop_ms = (time.perf_counter() - t0) / args.steps * 1e3   # host time per step INSIDE the free-running loop (see host_issue_ms below)
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    final_loss = float(loss.item())
    nan_loss_steps = int(nan_flags.sum().item())
    loss_note = None
    params_finite = bool(torch.isfinite(model._arena.flat).all().item())
    if final_loss != final_loss or nan_loss_steps:
        # the reference's loss normalises the predicted quaternion without an epsilon (models/losses.py) and the regressors end
        # in a ReLU, so an all-zero quaternion gives a 0/0 loss VALUE for that step while the gradients stay finite (the ReLU
        # mask zeroes them); the oracle shows the same on this seeded batch (tools/loss_trace.py).  NaN is not valid JSON.
        final_loss = None if final_loss != final_loss else final_loss
        loss_note = ("loss value 0/0 on an all-zero post-ReLU quaternion in %d of %d timed steps, as in the reference (models/losses.py:68-69; pinned by "
                     "tests/golden/model_no_nanloss.npz); gradients and parameters finite" % (nan_loss_steps, args.steps))

    log("[bench] timed region: %.1f ms/step" % (dt / args.steps * 1e3))
    # What the HOST needs to issue one step, measured on ISOLATED steps (outside the timed region): each starts with the device idle and
    # is timed until its last call has returned, so the host never waits for the device.  The per-step host time inside the free-running
    # timed loop (host_loop_ms_per_step) is NOT that: the HIP runtime lets a process run only a few steps ahead of the device (its
    # kernel-argument / signal pools are bounded: the loop's host time follows the device time at a fixed distance -- round 4 measured
    # 15.0 -> 14.1 ms when the DEVICE step went 19.24 -> 18.91 ms, profiles/r04_ab_t_gemm_off_chain.txt), so round 3's 12.3 ms was mostly waiting.
    iso = []
    for _ in range(5):
        fence()
        t1 = time.perf_counter()
        run_step()
        iso.append((time.perf_counter() - t1) * 1e3)
    fence()
    host_issue_ms = sorted(iso)[len(iso) // 2]
    if world > 1:   # the slowest rank's host is the one that matters
        th = torch.tensor([host_issue_ms, host_loop_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(th, op=dist.ReduceOp.MAX)
        host_issue_ms, host_loop_ms = th[0].item(), th[1].item()
    # ---- profiled pass (not timed): HIP events around every launch of the trunk plan, per kernel family ----
    plan = model.trunk._active
    n_prof = 3
    lib.rpe_resnet50_profile(plan.handle, 1)
    for _ in range(n_prof):
        train_step(model, batch, criterion, opt, train_obj_pose, "train", sync)
    torch.cuda.synchronize()
    ms = (ctypes.c_float * 6)()
    launches = (ctypes.c_int * 6)()
    flops = (ctypes.c_double * 6)()
    byts = (ctypes.c_double * 6)()
    lib.rpe_resnet50_profile_read(plan.handle, ms, launches, flops, byts)
    log("[bench] profiled pass read back")
    fam = {}
    for i, c in enumerate(CATS):
        if launches[i]:
            fam[c] = {"ms_per_step": ms[i] / n_prof, "launches_per_step": launches[i] // n_prof,
                      "tflops": (flops[i] / 1e12) / (ms[i] / n_prof / 1e3) if flops[i] else None,
                      "gbs": (byts[i] / 1e9) / (ms[i] / n_prof / 1e3) if byts[i] else None}
    # the same spans per kernel SYMBOL (names as rocprofv3 lists them, short form): the roofline line is for the symbol with
    # the largest summed time; achieved = its algorithmic FLOPs per launch / its average launch duration
    buf = ctypes.create_string_buffer(1 << 16)
    nbytes = lib.rpe_resnet50_profile_kernels(plan.handle, buf, len(buf))
    syms = []
    for line in buf.raw[:max(0, nbytes)].decode().splitlines():
        name, cnt, tms, fl, by = line.split(";")
        syms.append({"kernel": name, "launches": int(cnt), "ms": float(tms), "flops": float(fl), "bytes": float(by)})
    lib.rpe_resnet50_profile(plan.handle, 0)
    dom = max(syms, key=lambda k: k["ms"])   # over EVERY profiled symbol of the trunk plan (GEMMs, BN / pooling passes, packing)
    avg_ms = dom["ms"] / dom["launches"]
    tflops = dom["flops"] / dom["launches"] / 1e12 / (avg_ms / 1e3)   # (0 for a pure streaming kernel)
    gbs = dom["bytes"] / dom["launches"] / 1e9 / (avg_ms / 1e3)
    # which roof bounds this kernel: arithmetic intensity against the ridge peak_flops / peak_bytes
    ai = dom["flops"] / max(dom["bytes"], 1.0)
    hbm_bound = dom["flops"] <= 0 or ai < PEAK_TFLOPS[args.dtype] * 1e12 / (PEAK_HBM_GBS * 1e9)
    traffic, traffic_source = None, None
    build_id = lib.rpe_build_id().decode()   # hash of the sources the LOADED library was built from (compiled into it)
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic_pmc.json")
    if os.path.exists(tpath):  # PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE) of this same command, committed per round
        tj = json.load(open(tpath))
        if tj.get("build_id") != build_id:
            # the counters were collected on another build of the library: no number is better than a number about different kernels
            traffic_source = "profiles/hbm_traffic_pmc.json was measured on build %s, this process loaded build %s: traffic withheld" % (tj.get("build_id"), build_id)
        else:
            for k in tj.get("kernels", []):
                if k["kernel"] == dom["kernel"]:
                    traffic = round(k["per_launch_MB"] * 1e6)
                    traffic_source = ("lookup in profiles/hbm_traffic_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on build %s = the library "
                                      "this process loaded; NOT measured by this run)" % build_id)
    roofline = {"bound": "hbm" if hbm_bound else "mfma", "kernel": dom["kernel"],
                "achieved": round(gbs if hbm_bound else tflops, 2), "peak": PEAK_HBM_GBS if hbm_bound else PEAK_TFLOPS[args.dtype],
                "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": round((gbs / PEAK_HBM_GBS) if hbm_bound else (tflops / PEAK_TFLOPS[args.dtype]), 4), "traffic": traffic, "traffic_source": traffic_source,
                "build_id": build_id,
                "avg_launch_ms": round(avg_ms, 4), "launches_per_step": dom["launches"] // n_prof,
                "algorithmic_mb_per_launch": round(dom["bytes"] / dom["launches"] / 1e6, 2),
                "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 2),
                "arithmetic_intensity": round(ai, 1), "tflops": round(tflops, 1), "gbs": round(gbs, 1),
                # weight-gradient launches run on the engine's second stream beside the data-gradient / BN chain of the caller's
                # stream: their duration (and the chain's) is measured while the two share HBM and the CUs
                "stream": "second (beside the data-gradient chain)" if dom["kernel"].startswith("tn_kernel") else "caller's",
                "kernels": sorted(({"kernel": k["kernel"], "launches_per_step": k["launches"] // n_prof, "ms_per_step": round(k["ms"] / n_prof, 3),
                                    "tflops": round(k["flops"] / 1e12 / (k["ms"] / 1e3), 1) if k["flops"] else None,
                                    "gbs": round(k["bytes"] / 1e9 / (k["ms"] / 1e3), 1) if k["bytes"] else None} for k in syms),
                                  key=lambda k: -k["ms_per_step"])[:int(os.environ.get("RPE_BENCH_TOPK", "8"))],
                "families": {k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in fam.items()}}

    if rank == 0:
        imgs = args.batch * world * args.steps
        out = {
            "metric": "images/sec (train step, 224x224 bs256 per GPU)", "value": round(imgs / dt, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            # optimizer steps taken on the synthetic batch BEFORE the W warm-up steps (untimed 10-step windows): final_loss / nan_loss_steps
            # describe the state after precondition_steps + warmup + steps updates.  The timed loop's statements are `run_step()` and the
            # device-side NaN test of the loss value (two small elementwise launches per step, inside the timed region since round 3).
            "precondition_s": round(precondition_s, 2), "precondition_steps": precondition_steps,
            "host_issue_ms_per_step": round(host_issue_ms, 2), "host_loop_ms_per_step": round(host_loop_ms, 2),
            "config": {"workload": (workloads[args.model] if args.resnet == 50 else workloads[args.model].replace("ResNet-50", "ResNet-%d" % args.resnet).replace("BASELINE.json ", "as BASELINE.json ")) +
                                   (" [use_depth=True]" if dh else ""), "model": args.model,
                       "images_per_gpu": args.batch, "global_batch": args.batch * world, "resolution": 224, "latent_dim": 512,
                       "parallelism": "dp%d" % world + (" (RCCL world 1: staged joins + bucketed all-reduce on one GPU)" if args.force_dist else ""),
                       "launch": "hipGraph replay" if use_graph else "eager", "final_loss": final_loss, "nan_loss_steps": nan_loss_steps,
                       "loss_note": loss_note, "params_finite": params_finite},
            "roofline": roofline,
        }
        # whole-step view against both roofs (24.52 GFLOP and 152.9 MB per image)
        ips = imgs / dt / world
        out["step_roofline"] = {"tflops_per_gpu": round(ips * 24.52e9 / 1e12, 2), "frac_mfma": round(ips * 24.52e9 / 1e12 / PEAK_TFLOPS[args.dtype], 4),
                                "ideal_fused_gbs_per_gpu": round(ips * 152.9e6 / 1e9, 1), "frac_hbm": round(ips * 152.9e6 / 1e9 / PEAK_HBM_GBS, 4)}
        if args.resnet != 50:   # a side record: the per-image constants above and the metric are the ResNet-50 workload's
            out.pop("step_roofline")
            out["metric"] += " [side record: ResNet-%d trunk, not the BASELINE workload]" % args.resnet
        if world == 1 and not args.no_cpu_baseline and args.model == "no" and not dh and not args.force_dist and args.resnet == 50:
            out["cpu_baseline"], ref_first = cpu_baseline()
            del model, opt
            torch.cuda.empty_cache()
            dts = [(args.dtype, dtype)] + ([("f32", torch.float32)] if args.dtype != "f32" else [])
            out["pose_parity"] = pose_parity(ref_first, dts, dev)
            if args.dtype != "f32":
                out["f32_path"] = time_f32_path(dev, args.batch)
        print(json.dumps(out, allow_nan=False))
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
