"""hdf5_min.py pinned against the REAL libhdf5 (HDF5 1.10.6, /opt/conda of the build image), both directions:

  reader <- tests/golden/hdf5/*.h5: files written by libhdf5 itself (tests/golden/make_hdf5_fixtures.c) in the shapes h5py 2 / h5py 3 /
            Keras give their weight and full-model files (fixed and variable-length strings, continuation blocks, libver='latest' headers,
            compact + big-endian data, tracked times, a two-level group B-tree) -- committed, so this half runs anywhere;
            plus three files the real h5py 3.3.0 wrote with Keras' own call sequence (tests/golden/make_h5py_fixtures.py);
  writer -> what save_weights writes for the three graphs is re-read by libhdf5 through ctypes (tests/h5ref.py), walked by h5dump /
            h5ls, and loaded by the real h5py along Keras' load path (tests/h5py_check.py under /opt/conda/bin/python3.9); every dataset,
            type and attribute must come back identical.  Skipped where no libhdf5 / h5py is installed.
"""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import h5ref                                   # noqa: E402
from covidseg_amd import hdf5_min as H5        # noqa: E402
from covidseg_amd import weights as W          # noqa: E402

FIX = os.path.join(ROOT, "tests", "golden", "hdf5")
H5PY_PYTHON = os.environ.get("H5PY_PYTHON", "/opt/conda/bin/python3.9")
needs_libhdf5 = pytest.mark.skipif(not h5ref.available(), reason="no libhdf5 in this image")


def val(seed, n):
    """the generator's content rule (make_hdf5_fixtures.c: val)"""
    i = np.arange(n, dtype=np.uint64)
    h = ((i + np.uint64(977 * seed)) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    return ((h >> np.uint64(8)).astype(np.float64) / 16777216.0 - 0.5).astype(np.float32)


LAYERS = ["input_1", "conv2d_1", "batch_normalization_1", "a_layer_with_a_rather_long_name_1"]
SHAPES = {"conv2d_1/kernel:0": (3, 3, 1, 4), "conv2d_1/bias:0": (4,), "batch_normalization_1/gamma:0": (4,), "batch_normalization_1/beta:0": (4,),
          "batch_normalization_1/moving_mean:0": (4,), "batch_normalization_1/moving_variance:0": (4,),
          "a_layer_with_a_rather_long_name_1/kernel:0": (5, 7)}


# ---- libhdf5 wrote it, hdf5_min reads it ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,version", [("weights_h5py2_earliest", "2.3.1"), ("weights_tracked_times", "2.3.1"),
                                          ("weights_latest_compact_be", "2.3.1"), ("fullmodel_h5py3_vlen", "2.4.0"),
                                          ("h5py_keras_weights", "2.3.1"), ("h5py_keras_fullmodel", "2.4.0")])
def test_reader_on_files_written_by_libhdf5(name, version):
    path = os.path.join(FIX, name + ".h5")
    assert H5.is_hdf5(path)
    layers, attrs = H5.load_keras_weights(path)
    assert list(layers) == LAYERS and list(layers["input_1"]) == []
    seen = []
    for l, (ln, ws) in enumerate(layers.items()):
        for k, (wn, a) in enumerate(ws.items()):
            assert a.dtype == np.float32 and a.shape == SHAPES[wn], wn
            assert np.array_equal(a.reshape(-1), val(10 * l + k, a.size)), wn        # bit-exact (big-endian and compact storage included)
            seen.append(wn)
    assert seen == list(SHAPES)
    root = H5.read_file(path)
    full = "fullmodel" in name
    g = root["model_weights"] if full else root
    assert H5._strs(g.attrs["backend"]) == ["tensorflow"] and H5._strs(g.attrs["keras_version"]) == [version]
    if full:                                                          # variable-length UTF-8 strings through the global heap
        cfg = json.loads(H5._strs(attrs["model_config"])[0])
        assert cfg["class_name"] == "Model" and [l["name"] for l in cfg["config"]["layers"]] == ["input_1", "conv2d_1"]
        assert json.loads(H5._strs(attrs["training_config"])[0])["optimizer_config"]["config"]["lr"] == 0.0005
        ow = root["optimizer_weights"]
        assert H5._strs(ow.attrs["weight_names"]) == ["Adam/iterations:0", "Adam/conv2d_1/kernel/m:0"]
        assert int(ow["Adam/iterations:0"]) == 1234567890123 and ow["Adam/iterations:0"].dtype == np.int64
        m = ow["Adam/conv2d_1/kernel/m:0"]
        assert m.dtype == (np.float32 if name.startswith("h5py") else np.float64)                      # the C generator stores this one as float64
        assert m.shape == (3, 3, 1, 4) and np.array_equal(m.reshape(-1), val(77, 36).astype(m.dtype))


def test_reader_walks_a_two_level_group_btree_written_by_libhdf5():
    root = H5.read_file(os.path.join(FIX, "wide_group_two_level_btree.h5"))
    assert sorted(root.children) == [f"w{i:03d}:0" for i in range(300)]
    for i in range(300):
        assert np.array_equal(root[f"w{i:03d}:0"], val(i, 1))


@pytest.mark.parametrize("name,what", [("refused_chunked_deflate", "chunked"), ("refused_dense_group", "dense")])
def test_reader_names_the_feature_it_refuses(name, what):
    with pytest.raises(H5.H5FormatError, match=what):
        H5.read_file(os.path.join(FIX, name + ".h5"))


def test_reader_joins_the_attribute_chunks_keras_writes_above_64k():
    """save_attributes_to_hdf5_group splits `layer_names` into layer_names0, layer_names1, ... (h5py wrote this one)"""
    root = H5.read_file(os.path.join(FIX, "h5py_keras_chunked_attrs.h5"))
    names = H5._chunked_attr(root, "layer_names")
    assert len(names) == 2000 and names[0] == "layer_0000".ljust(40, "x") and names[-1] == "layer_1999".ljust(40, "x")
    assert "layer_names" not in root.attrs and len(root.children) == 5


def test_fixture_set_is_the_generators():
    got = sorted(os.path.basename(p) for p in glob.glob(os.path.join(FIX, "*.h5")))
    src = open(os.path.join(ROOT, "tests", "golden", "make_hdf5_fixtures.c")).read() + open(os.path.join(ROOT, "tests", "golden", "make_h5py_fixtures.py")).read()
    assert len(got) == 10 and all(g in src for g in got)


# ---- hdf5_min wrote it, libhdf5 reads it ----------------------------------------------------------------------------------------
@needs_libhdf5
@pytest.mark.parametrize("arch", ["unet", "unetpp", "classifier"])
@pytest.mark.parametrize("full_model", [False, True])
def test_libhdf5_reads_back_every_weight_file_the_writer_produces(tmp_path, arch, full_model):
    hw = (32, 32)
    w = W.init_weights(7, 1, arch, hw)
    rng = np.random.default_rng(2)
    w = {k: (v + rng.standard_normal(v.shape).astype(np.float32) * 0.01) for k, v in w.items()}
    f = str(tmp_path / ("unet_covid_weights_dice_coeff.hdf5" if full_model else "unet_0.8954_cosine_annealer.h5"))      # T1:1044 / T1:1079
    W.save_weights(f, w, 1, arch, hw, full_model=full_model)
    dsets, facts, attrs, groups = h5ref.read_tree(f)
    pre = "/model_weights" if full_model else ""
    kn = W.keras_names(1, arch, hw)
    assert len(dsets) == len(w)
    for k, v in w.items():
        p = f"{pre}/{kn[k].split('/')[0]}/{kn[k]}"
        assert facts[p] == (1, 4) and dsets[p].shape == v.shape and np.array_equal(dsets[p], v), p          # H5T_FLOAT, 4 bytes, same bits
    layer_lists = W._layer_weight_lists(w, 1, arch, hw)
    top = pre or "/"
    assert [s.decode() for s in attrs[(top, "layer_names")]] == [ln for ln, _ in layer_lists]
    assert attrs[(top, "backend")] == b"tensorflow" and attrs[(top, "keras_version")] == b"2.3.1"
    for ln, ws in layer_lists:
        got = attrs[(f"{pre}/{ln}", "weight_names")]
        assert [s.decode() for s in got] == [wn for wn, _ in ws] if ws else len(got) == 0
    if full_model:
        cfg = json.loads(attrs[("/", "model_config")].decode())
        assert [l["config"]["name"] for l in cfg["config"]["layers"]] == [ln for ln, _ in layer_lists]
    # the library's own tools walk the whole file without a complaint
    for tool, argv in ((h5ref.H5DUMP, ["-H"]), (h5ref.H5LS, ["-r"])):
        if tool:
            r = subprocess.run([tool, *argv, f], capture_output=True, text=True, timeout=120)
            assert r.returncode == 0 and "error" not in r.stderr.lower(), r.stderr[-500:]
            if tool == h5ref.H5LS:
                assert sum(1 for line in r.stdout.splitlines() if " Dataset " in line) == len(w)


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no interpreter with h5py in this image")
@pytest.mark.parametrize("arch,full_model", [("unet", False), ("unet", True), ("unetpp", True), ("classifier", False)])
def test_h5py_loads_the_writers_files_the_way_keras_does(tmp_path, arch, full_model):
    """the real h5py (3.3.0, the image's Anaconda interpreter; the system python has none) opens what save_weights wrote and follows
    saving.py's load path: layer_names -> group -> weight_names -> datasets"""
    hw = (32, 32)
    w = W.init_weights(9, 1, arch, hw)
    rng = np.random.default_rng(4)
    w = {k: (v + rng.standard_normal(v.shape).astype(np.float32) * 0.01) for k, v in w.items()}
    f, out = str(tmp_path / "m.h5"), str(tmp_path / "read.npz")
    W.save_weights(f, w, 1, arch, hw, full_model=full_model)
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    r = subprocess.run([H5PY_PYTHON, os.path.join(ROOT, "tests", "h5py_check.py"), f, out], capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out)
    layer_lists = W._layer_weight_lists(w, 1, arch, hw)
    assert list(z["layer_names"]) == [ln for ln, _ in layer_lists]
    assert str(z["group::backend"]) == "tensorflow" and str(z["group::keras_version"]) == "2.3.1"
    for ln, ws in layer_lists:
        assert list(z["weight_names::" + ln]) == [wn for wn, _ in ws]
        for wn, a in ws:
            got = z[f"w::{ln}::{wn}"]
            assert str(z[f"dtype::{ln}::{wn}"]) == "<f4" and got.shape == a.shape and np.array_equal(got, np.asarray(a, np.float32)), (ln, wn)
    if full_model:
        cfg = json.loads(str(z["rootattr::model_config"]))
        assert [l["config"]["name"] for l in cfg["config"]["layers"]] == [ln for ln, _ in layer_lists]


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no interpreter with h5py in this image")
def test_h5py_reads_the_optimizer_of_a_compiled_models_file(tmp_path):
    """model.save of a compiled model (T1:1046-1047): h5py follows saving.py's `_deserialize_model` -- training_config, optimizer_weights.attrs['weight_names'],
    one dataset per name -- and gets Adam's iteration count (int64 scalar) and slots back, in optimizer.weights order (iterations, m, v, vhat placeholders)"""
    hw = (32, 32)
    w = W.init_weights(3, 1, "unet", hw)
    names = W.trainable_names(1, "unet", hw)
    rng = np.random.default_rng(6)
    opt = {"step": 41, "lr": 5e-4, "m": {k: rng.standard_normal(w[k].shape).astype(np.float32) for k in names}, "v": {k: rng.random(w[k].shape).astype(np.float32) for k in names}}
    f, out = str(tmp_path / "unet_covid_weights_val_loss.hdf5"), str(tmp_path / "read.npz")
    W.save_weights(f, w, 1, "unet", hw, full_model=True, optimizer=opt)
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    r = subprocess.run([H5PY_PYTHON, os.path.join(ROOT, "tests", "h5py_check.py"), f, out], capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out)
    P = len(names)
    on = list(z["optimizer_weight_names"])
    assert len(on) == 1 + 3 * P and on[0] == "Adam/iterations:0" and on[1] == "training/Adam/m_0:0" and on[1 + P] == "training/Adam/v_0:0"
    assert z["ow::0"].dtype == np.int64 and z["ow::0"].shape == () and int(z["ow::0"]) == 41
    for i, k in enumerate(names):
        assert np.array_equal(z["ow::%d" % (1 + i)], opt["m"][k]) and np.array_equal(z["ow::%d" % (1 + P + i)], opt["v"][k]), k
        assert z["ow::%d" % (1 + 2 * P + i)].shape == (1,)
    tc = json.loads(str(z["training_config"]))
    assert tc["optimizer_config"] == {"class_name": "Adam", "config": {"beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-07, "decay": 0.0, "amsgrad": False, "learning_rate": 0.0005}}
    assert tc["loss"] == "bce_dice_loss" and tc["metrics"] == ["dice_coeff"]
    if h5ref.H5DUMP:
        r = subprocess.run([h5ref.H5DUMP, "-H", f], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "error" not in r.stderr.lower()


@needs_libhdf5
def test_libhdf5_reads_back_wide_groups_mixed_types_and_scalars(tmp_path):
    """generic trees: 300 links in one group (two-level B-tree, several local-heap growths), int64 / float64 / scalar datasets,
    string-array and numeric attributes on groups"""
    root = H5.Group()
    rng = np.random.default_rng(3)
    want = {}
    for i in rng.permutation(300):
        a = rng.standard_normal((int(i) % 5 + 1, 3)).astype(np.float32)
        root.create_dataset(f"wide/n{int(i):03d}:0", a); want[f"/wide/n{int(i):03d}:0"] = a
    root.create_dataset("types/i64", np.arange(-3, 9, dtype=np.int64).reshape(3, 4)); want["/types/i64"] = np.arange(-3, 9, dtype=np.int64).reshape(3, 4)
    root.create_dataset("types/f64", np.linspace(0, 1, 7)); want["/types/f64"] = np.linspace(0, 1, 7)
    root.create_dataset("types/scalar", np.float32(2.5)); want["/types/scalar"] = np.float32(2.5)
    root["types"].attrs["names"] = ["alpha", "be", "gamma_delta"]
    root["types"].attrs["lr"] = np.float32(5e-4)
    root.attrs["note"] = "bce_dice_loss"
    f = str(tmp_path / "t.h5")
    H5.write_file(f, root)
    dsets, facts, attrs, groups = h5ref.read_tree(f)
    assert sorted(dsets) == sorted(want) and sorted(groups) == ["/", "/types", "/wide"]
    for p, a in want.items():
        assert dsets[p].shape == np.shape(a) and np.array_equal(dsets[p], a), p
    assert facts["/types/i64"] == (0, 8) and facts["/types/f64"] == (1, 8) and facts["/types/scalar"] == (1, 4)
    assert attrs[("/types", "names")] == [b"alpha", b"be", b"gamma_delta"] and attrs[("/", "note")] == b"bce_dice_loss"
    assert np.float32(attrs[("/types", "lr")]) == np.float32(5e-4)


@needs_libhdf5
def test_both_readers_agree_on_the_libhdf5_fixtures():
    """hdf5_min's reading of each readable fixture equals libhdf5's own reading of it"""
    for name in ("weights_h5py2_earliest", "weights_tracked_times", "weights_latest_compact_be", "fullmodel_h5py3_vlen", "wide_group_two_level_btree",
                 "h5py_keras_weights", "h5py_keras_fullmodel", "h5py_keras_chunked_attrs"):
        path = os.path.join(FIX, name + ".h5")
        dsets, _, attrs, _ = h5ref.read_tree(path)
        root = H5.read_file(path)
        for p, a in dsets.items():
            mine = root[p]
            assert mine.shape == a.shape and np.array_equal(np.asarray(mine, a.dtype), a), (name, p)
        for (where, an), v in attrs.items():
            g = root if where == "/" else root[where]
            mine = H5._strs(g.attrs[an]) if isinstance(v, (bytes, list)) else g.attrs[an]
            ref = [x.decode() for x in ([v] if isinstance(v, bytes) else v)] if isinstance(v, (bytes, list)) else v
            assert (mine == ref) if isinstance(ref, list) else (np.size(mine) == np.size(ref)), (name, where, an)
