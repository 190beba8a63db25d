"""Drop-in for the reference module of the same name (Scripts/task1_crossval_4folds_unet.py:6): exports exactly
`four_fold_runner_unet_infection_segmentation`, so the reference's app.py (`from task1_crossval_4folds_unet import *`, app.py:7-12)
works unchanged with this directory on sys.path.  Import has no side effects."""
import os as _os, sys as _sys
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)

__all__ = ["four_fold_runner_unet_infection_segmentation"]


def four_fold_runner_unet_infection_segmentation(**kw):
    from covidseg_amd.runners import four_fold_runner_unet_infection_segmentation as _impl
    return _impl(**kw)


if __name__ == "__main__":
    four_fold_runner_unet_infection_segmentation()
