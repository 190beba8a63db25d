"""The module-level drop-in: the reference's menu script, unchanged, must import the ENGINE's six modules
and reach the engine's runners when started through dropin/run_app.py -- and plain `python Scripts/app.py` must be known NOT to."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DROPIN = os.path.join(ROOT, "one-stop-for-covid-19-infection-and-lung-segmentation-plus-classification_amd", "dropin")
REF_APP = "/root/reference/Scripts/app.py"


def run_app(app, word, extra_env=None, timeout=600):
    env = dict(os.environ, UNET_DROPIN_TRACE="1", **(extra_env or {}))
    env.pop("PYTHONPATH", None)
    return subprocess.run([sys.executable, os.path.join(DROPIN, "run_app.py"), app], input=word + "\n", capture_output=True, text=True, env=env, timeout=timeout)


def assert_dropins_imported(out):
    lines = [l for l in out.splitlines() if l.startswith("[dropin] ")]
    assert len(lines) == 6, out
    for l in lines:
        assert os.path.dirname(l.split(" <- ")[1]) == DROPIN, l


@pytest.mark.skipif(not os.path.isfile(REF_APP), reason="the reference checkout exists only in the build container")
def test_reference_app_text_resolves_the_dropins_and_exits_cleanly_on_an_unknown_choice():
    r = run_app(REF_APP, "seven")                         # APP:29 advertises 'seven'; no branch takes it -> the script just ends
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Enter from one of the" in r.stdout            # the reference's own prompt text was executed (APP:29)
    assert_dropins_imported(r.stdout)


@pytest.mark.skipif(not os.path.isfile(REF_APP), reason="the reference checkout exists only in the build container")
def test_reference_app_three_reaches_the_engine_and_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu test below")
    r = run_app(REF_APP, "three", {"UNET_EPOCHS": "1", "UNET_SAMPLES": "8", "UNET_SIZE": "32"})
    assert r.returncode != 0 and "UNetHipError" in r.stderr, (r.returncode, r.stderr[-2000:])       # no CPU fallback: the HIP engine refuses loudly
    assert "get_ipython" not in r.stderr                  # (what the reference's own module dies with when it is imported instead)
    assert_dropins_imported(r.stdout)


def test_menu_stand_in_resolves_the_dropins():
    r = run_app(os.path.join(HERE, "menu_like.py"), "seven")
    assert r.returncode == 0, r.stderr[-2000:]
    assert_dropins_imported(r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("word,needle", [("three", "test loss, test dice coefficient:"), ("six", "test loss, test dice coefficient:")])
def test_menu_choice_runs_on_the_engine(word, needle, tmp_path):
    """APP:44-45 / 56-57 through the launcher on the MI355X: 'three' (infection hold-out) and 'six' (lung) complete and print the reference's
    summary lines.  (The reference's app.py does not exist on the GPU box; tests/menu_like.py has the same contract.)"""
    app = REF_APP if os.path.isfile(REF_APP) else os.path.join(HERE, "menu_like.py")
    env = {"UNET_EPOCHS": "1", "UNET_SAMPLES": "8", "UNET_SIZE": "32"}
    p = subprocess.run([sys.executable, os.path.join(DROPIN, "run_app.py"), app], input=word + "\n", capture_output=True, text=True,
                       env=dict(os.environ, UNET_DROPIN_TRACE="1", **env), timeout=900, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    assert needle in p.stdout and "Best dice score:" in p.stdout and "Best Threshold for Recall:" in p.stdout
    assert_dropins_imported(p.stdout)
    assert os.path.exists(tmp_path / "unet_covid_weights_dice_coeff.hdf5")          # T1:1044 checkpoint written where the reference writes it (cwd)
