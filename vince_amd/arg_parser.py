"""Flags of the path (reference arg_parser.py:38-241).  Same names and defaults for everything the encoder +
contrastive path reads; dataset / transform / end-task flags are accepted and ignored so the reference's launch
scripts (vince/*.sh) parse unchanged.  New: --compute-dtype {bf16,fp32}."""
import argparse
import os

from . import constants


def solver_class(name):
    from . import solvers
    from .solvers import vince_solver
    if name != "VinceSolver":
        raise argparse.ArgumentTypeError("only VinceSolver is part of the HIP path, got %s" % name)
    return vince_solver.VinceSolver


def backbone_class(name):
    from .models.building_blocks import backbone_models
    if name not in backbone_models.__all__:
        raise argparse.ArgumentTypeError("backbone must be one of %s" % backbone_models.__all__)
    return getattr(backbone_models, name)


def build_parser():
    p = argparse.ArgumentParser(description="Video Noise Contrastive Estimation training args (MI355X path)")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--title", type=str, default="vince")
    p.add_argument("--description", type=str, default="run")
    p.add_argument("--num-frames", type=int, default=1)
    p.add_argument("--test-first", action="store_true")
    p.add_argument("--saved-variable-prefix", default="", type=str)
    p.add_argument("--new-variable-prefix", default="", type=str)
    p.add_argument("--base-logdir", default=constants.BASE_LOG_DIR, type=str)
    p.add_argument("--tensorboard-dir", default="tensorboard")
    p.add_argument("--checkpoint-dir", default=None)
    p.add_argument("--long-save-checkpoint-dir", default=None)
    p.add_argument("--data-path", default=None)
    p.add_argument("--dataset", default=None)
    p.add_argument("--transform", default="StandardVideoTransform")
    p.add_argument("--solver", type=solver_class, default="VinceSolver")
    p.add_argument("--backbone", type=backbone_class, default="ResNet18")
    p.add_argument("--end-task-classifier-num-classes", default=0, type=int)
    p.add_argument("--use-attention", action="store_true")
    p.add_argument("--jigsaw", action="store_true")
    p.add_argument("--freeze-feature-extractor", action="store_true")
    p.add_argument("--self-batch-comparison", action="store_true")
    p.add_argument("--inter-batch-comparison", action="store_true")
    p.add_argument("--vince-queue-size", default=256, type=int)
    p.add_argument("--vince-embedding-size", default=64, type=int)
    p.add_argument("--vince-momentum", type=float, default=0.999)
    p.add_argument("--vince-temperature", type=float, default=0.07)
    p.add_argument("--vince-self-temperature", type=float, default=0.03)
    p.add_argument("--no-multi-frame", dest="multi_frame", action="store_false")
    p.add_argument("--use-apex", action="store_true", help="accepted for script compatibility; use --compute-dtype bf16")
    p.add_argument("--compute-dtype", default="bf16", choices=["bf16", "fp32", "x3", "x3f"],
                   help="trunk arithmetic (head, similarity and loss are always fp32)")
    p.add_argument("--epochs", default=200, type=int)
    p.add_argument("--lr-decay-type", default="cos", choices=["cos", "step"])
    p.add_argument("--lr-step-schedule", default=[120, 160], nargs="*", type=int)
    p.add_argument("--pytorch-gpu-ids", type=str, default="0")
    p.add_argument("--feature-extractor-gpu-ids", type=str, default="0")
    p.add_argument("-j", "--num-workers", default=0, type=int)
    p.add_argument("-b", "--batch-size", default=256, type=int)
    p.add_argument("--use-videos", action="store_true")
    p.add_argument("-e", "--iterations-per-epoch", default=10000, type=int)
    p.add_argument("--base-lr", default=0.001, type=float)
    p.add_argument("--input-width", default=224, type=int)
    p.add_argument("--input-height", default=224, type=int)
    p.add_argument("--use-imagenet-weights", action="store_true")
    p.add_argument("--no-warmup", dest="use_warmup", action="store_false")
    p.add_argument("--log-frequency", default=10, type=int)
    p.add_argument("--image-log-frequency", default=1000, type=int)
    p.add_argument("--no-save", dest="save", action="store_false")
    p.add_argument("--no-restore", dest="restore", action="store_false")
    p.add_argument("--save-frequency", default=5000, type=int)
    p.add_argument("--long-save-frequency", default=25, type=int)
    p.add_argument("--disable-dataloader", action="store_true")
    p.add_argument("--use-imagenet", action="store_true")
    p.add_argument("--imagenet-data-path", type=str, default=None)
    p.add_argument("--video-sample-rate", default=5, type=int)
    p.add_argument("--max-video-length", type=int, default=512)
    p.add_argument("--only-use-shots", action="store_true")
    p.add_argument("--max-side-size", default=480, type=int)
    return p


def finalize(args):
    """Post-processing of arg_parser.py:200-241 that the path depends on."""
    args.input_size = (args.input_height, args.input_width)
    assert (not args.inter_batch_comparison) or (args.num_frames % 2 == 0), \
        "Must use an even number of frames when not using inter-batch comparison."
    assert (not args.self_batch_comparison) or args.inter_batch_comparison, \
        "self-batch-comparison is only used when inter-batch-comparison is on."
    assert args.multi_frame or args.num_frames == 1
    args.tensorboard_dir = os.path.join(args.base_logdir, args.title, args.tensorboard_dir,
                                        constants.TIME_STR + "_" + args.description)
    if args.checkpoint_dir is None:
        args.checkpoint_dir = os.path.join(args.base_logdir, args.title, "checkpoints_" + args.description)
    if args.long_save_checkpoint_dir is None:
        args.long_save_checkpoint_dir = os.path.join(args.base_logdir, args.title, "long_checkpoints",
                                                     constants.TIME_STR + "_" + args.description)
    # one process per GPU: LOCAL_RANK picks the device; the reference's id lists collapse to that one ordinal
    local = int(os.environ.get("LOCAL_RANK", args.pytorch_gpu_ids.split(",")[0]))
    args.pytorch_gpu_ids = [local]
    args.feature_extractor_gpu_ids = [local]
    args.saved_variable_prefix = args.saved_variable_prefix.split(",")
    args.new_variable_prefix = args.new_variable_prefix.split(",")
    # --use-imagenet (vince/train_moco_v2.sh:39) is kept as parsed: the side decoders are built and train on every batch whose
    # data_source is "IN" (labelled, VinceSolver.process_imagenet_data); batches from other sources never reach them
    return args


def parse_args(argv=None):
    return finalize(build_parser().parse_args(argv))
