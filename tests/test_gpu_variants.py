"""The kernel configurations that are selectable by environment switch (measured, rejected as defaults, kept
for experiments) must stay CORRECT: each switch is read once per process, so every variant gets one child process that runs
the conv / Linear parity cases of test_gpu_ops.py under it.  Children run one after the other (one GPU process at a time)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
NT = "conv_fwd_and_stats or conv_dgrad or dgrad_with_fused_bn_backward or linear_fwd_layout"
TN = "conv_wgrad or linear_wgrad"

ALL = NT + " or " + TN

# Independent switches are grouped per child process (each child pays the interpreter + torch start-up): a failure names the
# group, the switches in it are then run one by one by hand.
VARIANTS = [   # (switches, the parity cases of test_gpu_ops.py they can affect)
    # 256-row / 8-wave NT tiles for M >= 4096, K >= 1024; register staging in the weight-gradient kernel; stride-2 data gradient
    # without the parity-class decomposition; non-temporal epilogue stores
    ({"RPE_NT_BIG": "1", "RPE_TN_REG": "1", "RPE_NO_PARITY": "1", "RPE_NT_NTSTORE": "1"}, ALL),
    # 128-byte K rows + 2-slot ring for every K; few, long split-M slices in the weight gradient; 8 waves on the 128x128 tile
    ({"RPE_NT_BK64": "1", "RPE_TN_WGS": "64", "RPE_NT_W8": "1", "RPE_NT_K64": "0"}, ALL),
    # 64-byte K rows for every K; 4-slot ring of 32-row steps in the weight-gradient kernel; 128x256 / 8-wave tiles for N >= 256
    ({"RPE_NT_NOBK64": "1", "RPE_TN_RING": "1,4", "RPE_NT_WIDE": "2"}, ALL),
    ({"RPE_TN_RING": "2,3"}, TN),     # 3-slot ring of 64-row steps for every shape
]


@pytest.mark.parametrize("env,subset", VARIANTS, ids=[",".join("%s=%s" % kv for kv in e.items()) for e, _ in VARIANTS])
def test_kernel_variant_parity(env, subset):
    child_env = dict(os.environ)
    child_env.update(env)
    cmd = [sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_ops.py"), "-q", "-x", "-p", "no:cacheprovider", "-k", subset]
    r = subprocess.run(cmd, env=child_env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-1000:]
    assert r.returncode == 0, "variant %r failed:\n%s" % (env, tail)
    assert " passed" in r.stdout


# schedule / fusion switches of the trunk engine: one golden train step per model family must still match the reference
ENGINE_VARIANTS = [
    {"RPE_NO_OVERLAP": "1", "RPE_GRAM": "1"},        # everything on one stream; BN3 statistics from the Gram matrix of conv3's input (16-bit types)
    # projection-shortcut branch on the main stream; dense early-feature gradient + separate pool / BN backward passes for the
    # stem; conv3 backward through a materialised dy on the main stream; a block's weight gradients issued behind one event
    {"RPE_NO_FWD_OVERLAP": "1", "RPE_STEM_UNFUSED": "1", "RPE_NO_BN_FOLD": "1", "RPE_WGRAD_DEFER": "1"},
    # forward as two concurrent half-batch pipelines; conv3 weight gradient from dy; fp32 atomics instead of slabs
    {"RPE_FWD_SPLIT": "1", "RPE_NO_WGRAD_FOLD": "1", "RPE_WGRAD_ATOMIC": "1"},
    # projection-shortcut backward on the side stream; its BN as a pass of its own; folded weight gradient for layers 1-2 only;
    # inference convs unsplit and few-row Linear layers on the MFMA tile kernel
    # (the golden case's eval / rollout outputs run at 2-4 images: the default takes the split-K and per-column kernels there)
    {"RPE_CD_SIDE": "1", "RPE_NO_SPLITK": "1", "RPE_NO_LINEAR_ROWS": "1", "RPE_NO_DS_FUSE": "1", "RPE_WGRAD_FOLD_MAX": "128", "RPE_FOLD_PREP_UNFUSED": "1", "RPE_NO_LINEAR_SPLITK": "1", "RPE_NO_DS_FOLD": "1"},
]


@pytest.mark.parametrize("env", ENGINE_VARIANTS, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_engine_variant_parity(env):
    child_env = dict(os.environ)
    child_env.update(env)
    cmd = [sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_models.py"), "-q", "-x", "-p", "no:cacheprovider", "-k",
           "(test_model_fp32_matches_reference_and_oracle and tdo_v2) or test_model_bf16_tracks_fp32_reference"]   # (fp32 golden step + the 16-bit path)
    r = subprocess.run(cmd, env=child_env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-1000:]
    assert r.returncode == 0, "variant %r failed:\n%s" % (env, tail)
    assert " passed" in r.stdout
