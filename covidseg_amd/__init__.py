"""Import alias: the product package lives in the directory
``one-stop-for-covid-19-infection-and-lung-segmentation-plus-classification_amd/``
(name fixed by the build contract; hyphens make it un-importable with a plain
``import`` statement).  ``import covidseg_amd`` resolves sub-modules from that
directory, so ``covidseg_amd.engine`` is ``<that dir>/engine.py``.
"""
import os as _os

_REAL = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
    "one-stop-for-covid-19-infection-and-lung-segmentation-plus-classification_amd",
)
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
del _f
