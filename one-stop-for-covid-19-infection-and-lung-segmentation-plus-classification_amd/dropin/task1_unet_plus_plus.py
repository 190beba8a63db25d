"""Drop-in for the reference module of the same name (Scripts/task1_unet_plus_plus.py:6): exports exactly
`holdout_runner_unetplusplus_infection_segmentation`, so the reference's app.py (`from task1_unet_plus_plus import *`, app.py:7-12)
works unchanged with this directory on sys.path.  Import has no side effects."""
import os as _os, sys as _sys
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)

__all__ = ["holdout_runner_unetplusplus_infection_segmentation"]


def holdout_runner_unetplusplus_infection_segmentation(**kw):
    from covidseg_amd.runners import holdout_runner_unetplusplus_infection_segmentation as _impl
    return _impl(**kw)


if __name__ == "__main__":
    holdout_runner_unetplusplus_infection_segmentation()
