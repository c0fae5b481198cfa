#!/usr/bin/env python
"""Training entry point on the MI355X path.

Keeps every flag of the reference's scripts/train_model.py:17-47 (names, types, defaults) and its model-constructor
dispatch (:158-218), criterion dict (:100-105), Adam (:228) and train() call (:248-259).  The Robosuite environment
the reference builds at import time (:84-97) is replaced by seeded synthetic Robosuite-shaped episodes; flags that only
configure the simulator (--controller, --robots, --use_placement_initializer, --motion) are accepted and recorded.
Added flags: --dtype {bf16,f16,f32}, --optimizer {fused,torch}, --episodes_seed.

Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/train_model.py ...`;
episodes are sharded over ranks and gradients SUM-all-reduced over RCCL.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODELS = {'n', 'no', 'td', 'tdo', 'tdo_v2'}


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--model", type=str, default="n", help="Which mode to run. Options are 'n', 'no', 'td', 'tdo' or 'tdo_v2'")
    p.add_argument("--controller", type=str, default="OSC_POSE", help="(simulator only) controller name")
    p.add_argument("--camera_name", type=str, default="frontview", help="Name of camera to render for observations")
    p.add_argument("--horizon", type=int, default=100, help="Horizon per episode run")
    p.add_argument("--sequence_length", type=int, default=10, help="Sequence length for LSTMs")
    p.add_argument("--noise_scale", type=float, default=0.001, help="Noise scale for self measurements")
    p.add_argument("--latent_dim", type=int, default=1024, help="Dimension of output from ResNet")
    p.add_argument("--hidden_dim", nargs="+", type=int, default=[512], help="Hidden dimensions in FC network (naive only), or LSTM net (td/o only)")
    p.add_argument("--proprio_hidden_dim", type=int, default=64, help="Hidden dimensions in proprio LSTM net (tdo_v2 only)")
    p.add_argument("--lr", type=float, default=0.001, help="Learning rate for Adam optimizer")
    p.add_argument("--n_train_episodes_per_epoch", type=int, default=10, help="Number of training episodes per epoch")
    p.add_argument("--n_val_episodes_per_epoch", type=int, default=2, help="Number of validation episodes per epoch")
    p.add_argument("--env", type=str, default="TwoArmLift", help="Environment name (two-arm iff it contains 'TwoArm')")
    p.add_argument("--robots", nargs="+", type=str, default=["Panda", "Sawyer"], help="(simulator only) robot names")
    p.add_argument("--use_placement_initializer", action="store_true", help="(simulator only)")
    p.add_argument("--feature_extract", action="store_true", help="Whether ResNet will be set to feature extract mode or not")
    p.add_argument("--no_proprioception", action="store_true", help="If set, will not leverage proprioceptive measurements during training")
    p.add_argument("--use_depth", action="store_true", help="Whether to use depth or not")
    p.add_argument("--use_pretrained", action="store_true", help="Whether to use pretrained ResNet or not")
    p.add_argument("--obj_name", type=str, default=None, help="Object name to generate observations of")
    p.add_argument("--motion", type=str, default="random", help="Type of robot motion to use")
    p.add_argument("--distance_metric", type=str, default="l2", help="Distance metric to use for loss")
    p.add_argument("--loss_mode", type=str, default="pose", help="Type of loss to use. Options are 'position' or 'pose'")
    p.add_argument("--loss_scale_factor", type=float, default=1.0, help="Scaling factor for Pose loss")
    p.add_argument("--alpha", type=float, default=0.5, help="Orientation loss scaling factor relative to position error")
    p.add_argument("--n_epochs", type=int, default=5000, help="Number of epochs")
    p.add_argument("--load_checkpoint", action="store_true", help="Whether to load prior trained model")
    p.add_argument("--checkpoint_model_path", type=str, default="../log/runs/model.pth", help="Path to checkpoint .pth file to load into model")
    # additions
    p.add_argument("--dtype", choices=["bf16", "f16", "f32"], default="bf16",
                   help="compute dtype of the conv trunk (fp32 accumulate either way; f16 adds dynamic loss scaling and needs --optimizer fused)")
    p.add_argument("--optimizer", choices=["fused", "torch"], default="fused", help="FusedAdam (one HIP kernel) or torch.optim.Adam")
    p.add_argument("--episodes_seed", type=int, default=1234, help="seed of the synthetic episode generator")
    p.add_argument("--no_save", action="store_true", help="do not write the best-validation checkpoint")
    return p


DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def build_model(args, compute_dtype):
    from rgb_proprioceptive_pose_estimator_amd import models as M
    num_resnet_layers, feature_layer_nums = 50, (9,)
    assert args.model in MODELS, "Error: Invalid model specified. Options are: {}".format(MODELS)
    if args.model == 'n':
        return M.NaiveEndEffectorStateEstimator(hidden_dims_pre_measurement=args.hidden_dim, hidden_dims_post_measurement=args.hidden_dim,
                                                num_resnet_layers=num_resnet_layers, latent_dim=args.latent_dim,
                                                feature_extract=args.feature_extract, compute_dtype=compute_dtype)
    if args.model == 'no':
        return M.NaiveObjectStateEstimator(object_name=args.obj_name, hidden_dims=args.hidden_dim, num_resnet_layers=num_resnet_layers,
                                           latent_dim=args.latent_dim, feature_extract=args.feature_extract,
                                           feature_layer_nums=feature_layer_nums, use_depth=args.use_depth, use_pretrained=args.use_pretrained,
                                           no_proprioception=args.no_proprioception, compute_dtype=compute_dtype)
    common = dict(num_resnet_layers=num_resnet_layers, latent_dim=args.latent_dim, sequence_length=args.sequence_length,
                  feature_extract=args.feature_extract, feature_layer_nums=feature_layer_nums, use_depth=args.use_depth,
                  use_pretrained=args.use_pretrained, device="cuda", compute_dtype=compute_dtype)
    if args.model == 'td':
        return M.TemporallyDependentStateEstimator(hidden_dim_pre_measurement=args.hidden_dim[0], hidden_dim_post_measurement=args.hidden_dim[0], **common)
    if args.model == 'tdo':
        return M.TemporallyDependentObjectStateEstimator(object_name=args.obj_name, hidden_dim=args.hidden_dim[0],
                                                         no_proprioception=args.no_proprioception, **common)
    return M.TemporallyDependentObjectStateEstimatorV2(object_name=args.obj_name, img_hidden_dim=args.hidden_dim[0],
                                                       proprio_hidden_dim=args.proprio_hidden_dim, **common)


def main(argv=None):
    args = build_parser().parse_args(argv)
    from rgb_proprioceptive_pose_estimator_amd.dist import init_from_env
    from rgb_proprioceptive_pose_estimator_amd.models import PoseDistanceLoss
    from rgb_proprioceptive_pose_estimator_amd.optim import FusedAdam
    from rgb_proprioceptive_pose_estimator_amd.util.data_utils import SyntheticEpisodeDataset
    from rgb_proprioceptive_pose_estimator_amd.util.learn_utils import train

    rank, world, local = init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("train_model.py: no MI355X visible; this path has no CPU fallback")
    torch.cuda.set_device(local)
    device = "cuda:%d" % local
    if rank == 0:
        print("*" * 20 + "\nRunning experiment:\n")
        for k, v in sorted(vars(args).items()):
            print("{}: {}".format(k, v))
        print("world size: {}\n".format(world) + "*" * 20)
    if args.model in ('no', 'tdo', 'tdo_v2') and args.obj_name is None:
        raise SystemExit("--obj_name is required for object-pose models (e.g. cube, hammer, robot1_eef)")
    crit = lambda: PoseDistanceLoss(distance_metric=args.distance_metric, scale_factor=args.loss_scale_factor, alpha=args.alpha, mode=args.loss_mode)
    criterion = {"x0_loss": crit(), "x1_loss": crit(), "obj_loss": crit(), "val_loss": PoseDistanceLoss(mode="val")}
    torch.manual_seed(0)
    model = build_model(args, DTYPES[args.dtype])
    if args.load_checkpoint:
        model.load_state_dict(torch.load(args.checkpoint_model_path, map_location="cpu"))
    if args.dtype == "f16" and args.optimizer != "fused":
        raise SystemExit("--dtype f16 needs --optimizer fused: the loss-scale unscale / skip logic lives in FusedAdam.step (amp.py)")
    opt_cls = FusedAdam if args.optimizer == "fused" else torch.optim.Adam
    optimizer = opt_cls(model.parameters(), lr=args.lr)
    dataset = SyntheticEpisodeDataset(horizon=args.horizon, use_depth=args.use_depth, obj_name=args.obj_name, is_two_arm="TwoArm" in args.env,
                                      motion=args.motion, seed=args.episodes_seed + 1000 * rank, device=device, env_name=args.env)
    params = {"camera_name": args.camera_name, "noise_scale": args.noise_scale}
    if rank == 0:
        print("Training...")
    return train(model=model, dataset=dataset, criterion=criterion, optimizer=optimizer, num_epochs=args.n_epochs,
                 num_train_episodes_per_epoch=args.n_train_episodes_per_epoch, num_val_episodes_per_epoch=args.n_val_episodes_per_epoch,
                 params=params, device=device, save_model=not args.no_save)


if __name__ == "__main__":
    main()
