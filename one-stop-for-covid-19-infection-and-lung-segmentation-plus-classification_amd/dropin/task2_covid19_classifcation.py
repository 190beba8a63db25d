"""Drop-in for the reference module of the same name (Scripts/task2_covid19_classifcation.py:6): exports exactly
`runner_classification`, so the reference's app.py (`from task2_covid19_classifcation import *`, app.py:7-12) works unchanged with
this directory on sys.path.  Import has no side effects."""
import os as _os, sys as _sys
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)

__all__ = ["runner_classification"]


def runner_classification(**kw):
    from covidseg_amd.runners import runner_classification as _impl
    return _impl(**kw)


if __name__ == "__main__":
    runner_classification()
