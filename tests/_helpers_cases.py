"""Golden-case table (must match oracle/gen_golden.py)."""
CASES = {
    # kind: (cfg, lead dims, weight seed, data seed) -- must match oracle/gen_golden.py
    "no": (dict(latent_dim=64, hidden=[32, 16], use_depth=False, no_proprioception=False), (2,), 11, 101),
    "n": (dict(latent_dim=64, hidden=[32]), (2,), 12, 102),
    "td": (dict(latent_dim=64, hidden=32, use_depth=False), (2, 2), 13, 103),
    "tdo": (dict(latent_dim=64, hidden=32, use_depth=True, no_proprioception=False), (2, 2), 14, 104),
    "tdo_v2": (dict(latent_dim=64, hidden=32, proprio_hidden=8, use_depth=False), (2, 2), 15, 105),
}
LOSS_CFG = dict(metric="combined", scale=1.0, alpha=0.5, mode="pose")
# BASELINE.json configs[0] at its own size (32 images, latent 512, hidden [1024, 256, 64]) -- must match oracle/gen_golden.py C1
C1 = (dict(latent_dim=512, hidden=[1024, 256, 64], use_depth=False, no_proprioception=False), (32,), 21, 201)
SAMPLE_STRIDE, SAMPLE_MAX = 997, 4096
# NaiveObjectStateEstimator on ResNet-101 (import_resnet's deeper bottleneck option) -- must match oracle/gen_golden.py R101
R101 = (dict(latent_dim=64, hidden=[32], use_depth=False, no_proprioception=False, depth=101), (2,), 41, 401)
# ... and on ResNet-18, the BasicBlock member import_resnet reaches -- must match oracle/gen_golden.py R18
R18 = (dict(latent_dim=64, hidden=[32], use_depth=False, no_proprioception=False, depth=18), (2,), 45, 451)
# BASELINE.json configs[2]: the two-arm TD model on sequences of four frames -- must match oracle/gen_golden.py TD_S4
TD_S4 = (dict(latent_dim=64, hidden=32, use_depth=False), (4, 2), 51, 501)
# feature_layer_nums other than (9,): every hook the reference can run at 224x224 (given out of order), depth heads on; and None
# -- must match oracle/gen_golden.py HOOKS / NOHOOK
HOOKS = (dict(latent_dim=64, hidden=[32], use_depth=True, no_proprioception=False, hooks=(3, 0, 9, 2, 1)), (2,), 61, 601)
HOOKS18 = (dict(latent_dim=64, hidden=[32], use_depth=True, no_proprioception=False, hooks=(3, 0, 9, 2, 1), depth=18), (2,), 63, 603)   # ... on resnet18
NOHOOK = (dict(latent_dim=64, hidden=[32], use_depth=False, no_proprioception=False, hooks=None), (2,), 62, 602)
# BASELINE.json configs[2..4] at the head sizes the reference's scripts train (latent 512, hidden 512, proprio hidden 64), lead dims
# (S, N) = (4, 8) -- must match oracle/gen_golden.py SEQ_CFG
SEQ_CFG = {
    "td": (dict(latent_dim=512, hidden=512, use_depth=False), (4, 8), 71, 701),
    "tdo": (dict(latent_dim=512, hidden=512, use_depth=True, no_proprioception=False), (4, 8), 72, 702),
    "tdo_v2": (dict(latent_dim=512, hidden=512, proprio_hidden=64, use_depth=False), (4, 8), 73, 703),
}
SEQ_SAMPLE_STRIDE, SEQ_SAMPLE_MAX = 499, 16384
