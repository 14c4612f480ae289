"""Programmatic construction of the ``args`` namespace the path reads (SURVEY.md 8b config contract), for callers
that do not go through the command line (tests, bench.py, notebooks)."""
import types

from .models.building_blocks import backbone_models

# (backbone, input, batch, K, D, T, dtype) of BASELINE.json's configs
CONFIGS = {
    "C1": dict(backbone="ResNet18", input_size=(64, 64), batch_size=32, vince_queue_size=512, vince_embedding_size=64,
               vince_temperature=0.07, compute_dtype="fp32"),
    "C2": dict(backbone="ResNet18", input_size=(224, 224), batch_size=256, vince_queue_size=4096, vince_embedding_size=64,
               vince_temperature=0.07, compute_dtype="fp32"),
    "C3": dict(backbone="ResNet50", input_size=(224, 224), batch_size=256, vince_queue_size=65536,
               vince_embedding_size=128, vince_temperature=0.2, compute_dtype="bf16", base_lr=0.03),
}


def make_args(**kw):
    d = dict(
        debug=True, title="vince", description="run", num_frames=1, backbone="ResNet18", use_attention=False,
        jigsaw=False, inter_batch_comparison=False, self_batch_comparison=False, batch_size=32, vince_queue_size=512,
        vince_embedding_size=64, vince_momentum=0.999, vince_temperature=0.07, vince_self_temperature=0.03,
        use_imagenet=False, use_imagenet_weights=False, compute_dtype="fp32", base_lr=0.03, epochs=200,
        lr_decay_type="cos", lr_step_schedule=[120, 160], use_warmup=True, pytorch_gpu_ids=[0],
        feature_extractor_gpu_ids=[0], input_size=(64, 64), iterations_per_epoch=10, log_frequency=1, save_frequency=10 ** 9,
        long_save_frequency=25, save=False, restore=False, checkpoint_dir=None, long_save_checkpoint_dir=None,
        saved_variable_prefix=[""], new_variable_prefix=[""], freeze_feature_extractor=False, test_first=False,
        batch_source=None, val_batch_source=None, prefetch_thread=False, base_logdir="logs",
    )
    d.update(kw)
    if isinstance(d["backbone"], str):
        d["backbone"] = getattr(backbone_models, d["backbone"])
    return types.SimpleNamespace(**d)
