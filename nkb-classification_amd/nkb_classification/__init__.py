"""MI355X-native drop-in for the `nkb_classification` package's training hot path.

Same module and function names as the reference (engine.train_epoch / val_epoch, model.get_model,
losses.get_loss, utils.get_optimizer / get_scheduler, metrics.compute_metrics, logging.BaseLogger); the
per-step compute runs in hand-written HIP kernels (libnkbhip.so, C ABI in include/nkbhip.h).
"""
import os as _os

# compute / weight-gradient / gradient-exchange streams (+ RCCL's) need more than the runtime's default of 4 hardware
# queues to actually run side by side (see bench.py); harmless when the HIP runtime is already up
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__version__ = "0.1.0"
