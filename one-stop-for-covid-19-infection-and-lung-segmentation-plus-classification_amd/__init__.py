"""MI355X-native U-Net segmentation engine (drop-in for the reference's U-Net hot path).

Public surface:
  runners.holdout_runner_unet_infection_segmentation / runners.runner_lung_segmentation
  keras_like.UNetModel   -- compile / fit / evaluate / predict / save_weights / load_weights
  engine.HipUNet         -- the HIP backend (libunet_hip.so through the C ABI of include/unet_hip.h)
Importing this package has no side effects and does not need a GPU; constructing the
backend does (there is no CPU fallback).
"""
__all__ = ["runners", "keras_like", "engine", "weights", "data"]
