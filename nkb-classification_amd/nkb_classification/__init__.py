"""MI355X-native drop-in for the `nkb_classification` package's training hot path.

Same module and function names as the reference (engine.train_epoch / val_epoch, model.get_model,
losses.get_loss, utils.get_optimizer / get_scheduler, metrics.compute_metrics, logging.BaseLogger); the
per-step compute runs in hand-written HIP kernels (libnkbhip.so, C ABI in include/nkbhip.h).
"""
__version__ = "0.1.0"
