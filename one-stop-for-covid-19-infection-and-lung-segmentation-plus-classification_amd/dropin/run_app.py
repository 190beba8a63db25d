#!/usr/bin/env python
"""Run the reference's menu script UNCHANGED against the engine:

    python <package>/dropin/run_app.py [/path/to/reference/Scripts/app.py]

`python Scripts/app.py` itself cannot pick the drop-ins up: the interpreter puts the SCRIPT's directory at sys.path[0], ahead of
PYTHONPATH, so app.py:7-12 (`from task1_..._comments import *` ...) would import the reference's own Keras modules sitting next
to it.  This launcher puts THIS directory (the six same-named modules) at sys.path[0] and executes the given app.py text with
runpy.run_path, which does not add the script's directory.  The path of app.py: argv[1], else $UNET_REFERENCE_APP, else
/root/reference/Scripts/app.py.  With UNET_DROPIN_TRACE=1 the origin of each of the six modules is printed at exit (tests use it
to prove the engine's modules, not the reference's, were imported)."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
MODULES = ("task1_crossval_3folds_unet", "task1_crossval_4folds_unet", "task1_preprocessing_plus_unet_with_comments", "task1_unet_plus_plus",
           "task2_covid19_classifcation", "task3_lung_segmentation_unet")


def main(argv):
    app = argv[1] if len(argv) > 1 else os.environ.get("UNET_REFERENCE_APP", "/root/reference/Scripts/app.py")
    if not os.path.isfile(app):
        sys.exit(f"run_app.py: {app} not found (pass the reference's Scripts/app.py as the first argument)")
    sys.path[:] = [HERE] + [p for p in sys.path if os.path.abspath(p or ".") not in (HERE, os.path.dirname(os.path.abspath(app)))]
    try:
        runpy.run_path(app, run_name="__main__")
    finally:
        if os.environ.get("UNET_DROPIN_TRACE"):
            for m in MODULES:
                print(f"[dropin] {m} <- {getattr(sys.modules.get(m), '__file__', None)}", flush=True)


if __name__ == "__main__":
    main(sys.argv)
