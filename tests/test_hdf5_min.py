"""hdf5_min.py: the HDF5 subset behind Keras weight files (T1:1046-1047, 1073, 1079).  The pin against the real libhdf5 is
tests/test_hdf5_pinned.py; the checks here need no library: (1) write -> read round trips of arbitrary trees, (2) `walk()` below: an INDEPENDENT decoder written straight from the field
tables of the HDF5 File Format Specification (superblock v0, v1 object headers, symbol-table groups: local heap / v1 B-tree / SNOD,
dataspace v1, datatype classes 0 / 1 / 3, layout v3, attribute v1) that re-derives every address, size, alignment and B-tree key
invariant libhdf5 relies on, (3) the Keras layout of keras/engine/saving.py, incl. files whose layer names carry other counters."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from covidseg_amd import hdf5_min as H5       # noqa: E402
from covidseg_amd import weights as W         # noqa: E402

UNDEF = 0xFFFFFFFFFFFFFFFF


# ---- an independent structural walker -----------------------------------------------------------------------------------------
def u(d, p, n):
    return int.from_bytes(d[p:p + n], "little")


def walk(d):
    """-> {path: ('group', attrs) | ('dataset', shape, class, size, raw bytes)}; asserts the invariants on the way"""
    assert d[:8] == b"\x89HDF\r\n\x1a\n"
    assert d[8] == 0 and d[9] == 0 and d[10] == 0 and d[12] == 0           # superblock / free-space / root-entry / shared-header versions
    assert d[13] == 8 and d[14] == 8                                         # sizes of offsets and lengths
    leaf_k, int_k = u(d, 16, 2), u(d, 18, 2)
    assert u(d, 20, 4) == 0 and u(d, 24, 8) == 0                             # consistency flags, base address
    assert u(d, 32, 8) == UNDEF and u(d, 48, 8) == UNDEF                     # no free-space info, no driver info
    eof = u(d, 40, 8)
    assert eof == len(d)                                                     # end-of-file address == file size
    assert u(d, 56, 8) == 0                                                  # root link-name offset
    root_oh, cache = u(d, 64, 8), u(d, 72, 4)
    out = {}

    def ohdr(addr):
        assert addr % 8 == 0 and addr + 16 <= eof
        assert d[addr] == 1 and d[addr + 1] == 0
        nmsg, refc, size = u(d, addr + 2, 2), u(d, addr + 4, 4), u(d, addr + 8, 4)
        assert refc == 1 and size % 8 == 0 and addr + 16 + size <= eof
        msgs, p = [], addr + 16
        while p < addr + 16 + size:
            t, n, fl = u(d, p, 2), u(d, p + 2, 2), d[p + 4]
            assert n % 8 == 0 and d[p + 5:p + 8] == b"\0\0\0"
            msgs.append((t, d[p + 8:p + 8 + n])); p += 8 + n
        assert p == addr + 16 + size and len(msgs) == nmsg                    # exact message count (libhdf5 rejects a mismatch)
        return msgs

    def dtype(b):
        cls, ver = b[0] & 15, b[0] >> 4
        assert ver == 1
        size = u(b, 4, 4)
        if cls == 1:
            assert b[1] == 0x20 and b[3] == 0 and u(b, 8, 2) == 0 and u(b, 10, 2) == 8 * size
            if size == 4:
                assert b[2] == 31 and tuple(b[12:16]) == (23, 8, 0, 23) and u(b, 16, 4) == 127
            return ("f", size, 20)
        if cls == 0:
            assert u(b, 8, 2) == 0 and u(b, 10, 2) == 8 * size
            return ("i" if b[1] & 8 else "u", size, 12)
        assert cls == 3 and (b[1] & 15) == 1 and (b[1] >> 4) == 0            # NULL-padded ASCII
        return ("S", size, 8)

    def dspace(b):
        assert b[0] == 1 and b[2] == 0 and b[3:8] == b"\0" * 5
        return tuple(u(b, 8 + 8 * i, 8) for i in range(b[1])), 8 + 8 * b[1]

    def attr(b):
        assert b[0] == 1 and b[1] == 0
        ns, ts, ss = u(b, 2, 2), u(b, 4, 2), u(b, 6, 2)
        p = 8
        name = b[p:p + ns]; assert name[-1] == 0; p += ns + (-ns % 8)
        kind, size, tl = dtype(b[p:p + ts]); assert tl == ts; p += ts + (-ts % 8)
        shape, sl = dspace(b[p:p + ss]); assert sl == ss; p += ss + (-ss % 8)
        n = int(np.prod(shape)) if shape else 1
        assert len(b) >= p + n * size
        raw = b[p:p + n * size]
        val = np.frombuffer(raw, dtype=f"S{size}" if kind == "S" else f"<{kind}{size}").reshape(shape)
        return name[:-1].decode(), val

    def group(path, oh):
        msgs = ohdr(oh)
        st = [m for m in msgs if m[0] == 0x11]
        assert len(st) == 1 and all(m[0] in (0x11, 0x0C) for m in msgs)
        bt, hp = u(st[0][1], 0, 8), u(st[0][1], 8, 8)
        out[path or "/"] = ("group", dict(attr(m[1]) for m in msgs if m[0] == 0x0C))
        assert d[hp:hp + 4] == b"HEAP" and d[hp + 4] == 0
        hsize, hfree, hdata = u(d, hp + 8, 8), u(d, hp + 16, 8), u(d, hp + 24, 8)
        assert hsize % 8 == 0 and hdata % 8 == 0 and hdata + hsize <= eof and (hfree == 1 or hfree < hsize)
        assert d[hdata] == 0                                                  # offset 0 = the empty string

        def name_at(off):
            assert off < hsize
            e = d.index(b"\0", hdata + off)
            assert e < hdata + hsize
            return d[hdata + off:e]
        names = []

        levels = {}

        def node(addr, level_expected):
            assert addr % 8 == 0 and d[addr:addr + 4] == b"TREE" and d[addr + 4] == 0
            level, used = d[addr + 5], u(d, addr + 6, 2)
            assert used <= 2 * int_k and addr + 24 + (4 * int_k + 1) * 8 <= eof   # the node is allocated at full size
            assert level_expected is None or level == level_expected
            levels.setdefault(level, []).append((addr, u(d, addr + 8, 8), u(d, addr + 16, 8)))
            keys = [u(d, addr + 24 + 16 * i, 8) for i in range(used + 1)]
            kids = [u(d, addr + 32 + 16 * i, 8) for i in range(used)]
            for i, c in enumerate(kids):
                lo, hi = name_at(keys[i]), name_at(keys[i + 1])
                before = len(names)
                if level:
                    node(c, level - 1)
                else:
                    assert d[c:c + 4] == b"SNOD" and d[c + 4] == 1 and c + 8 + 2 * leaf_k * 40 <= eof
                    ns = u(d, c + 6, 2)
                    assert 1 <= ns <= 2 * leaf_k
                    for j in range(ns):
                        e = c + 8 + 40 * j
                        names.append((name_at(u(d, e, 8)), u(d, e + 8, 8), u(d, e + 16, 4), d[e + 24:e + 40]))
                mine = [n[0] for n in names[before:]]
                assert all(lo < n <= hi for n in mine) and mine[-1] == hi     # (left key, right key] covers the child; right key = its largest name
        node(bt, None)
        for lv in levels.values():                                            # sibling chain of every level, left to right
            for i, (addr, left, right) in enumerate(lv):
                assert left == (lv[i - 1][0] if i else UNDEF) and right == (lv[i + 1][0] if i + 1 < len(lv) else UNDEF)
        assert [n[0] for n in names] == sorted(n[0] for n in names) and len({n[0] for n in names}) == len(names)
        for nm, coh, ctype, scratch in names:
            sub = f"{path}/{nm.decode()}"
            cm = ohdr(coh)
            if any(m[0] == 0x11 for m in cm):
                assert ctype == 1
                stm = [m for m in cm if m[0] == 0x11][0][1]
                assert scratch == stm[:16]                                    # cached B-tree / heap addresses agree with the header message
                group(sub, coh)
            else:
                assert ctype == 0
                t = {m[0]: m[1] for m in cm}
                assert set(t) == {1, 3, 5, 8}
                shape, _ = dspace(t[1]); kind, size, _ = dtype(t[3])
                assert t[5][:8] == bytes([2, 2, 2, 1, 0, 0, 0, 0])
                assert t[8][0] == 3 and t[8][1] == 1
                da, dn = u(t[8], 2, 8), u(t[8], 10, 8)
                n = int(np.prod(shape)) if shape else 1
                assert dn == n * size and (dn == 0 or (da % 8 == 0 and da + dn <= eof))
                out[sub] = ("dataset", shape, kind, size, d[da:da + dn] if dn else b"")
    rm = ohdr(root_oh)
    assert cache == 1 and d[80:96] == [m for m in rm if m[0] == 0x11][0][1][:16]
    group("", root_oh)
    return out


def tree_equal(a, b):
    assert list(a.attrs) == list(b.attrs)
    for k in a.attrs:
        x, y = np.asarray(H5._attr_array(a.attrs[k])), np.asarray(b.attrs[k])
        assert x.shape == y.shape and (x == y).all(), k
    assert sorted(a.children) == sorted(b.children)
    for k, c in a.children.items():
        if isinstance(c, H5.Group):
            tree_equal(c, b.children[k])
        else:
            assert c.dtype == b.children[k].dtype and c.shape == b.children[k].shape and np.array_equal(c, b.children[k]), k


@pytest.mark.parametrize("nchild", [0, 1, 7, 8, 9, 40, 300])
def test_round_trip_and_structure(nchild):
    rng = np.random.default_rng(nchild)
    root = H5.Group()
    root.attrs["backend"] = "tensorflow"; root.attrs["numbers"] = np.arange(5, dtype=np.int64); root.attrs["f"] = np.float32(1.5)
    root.attrs["names"] = [f"layer_{i}" for i in range(nchild)] if nchild else np.zeros((0,), "S1")
    for i in range(nchild):                                          # > 8 children: several SNODs; > 256: a second B-tree level
        g = root.create_group(f"layer_{i}")
        g.attrs["weight_names"] = [f"layer_{i}/kernel:0", f"layer_{i}/bias:0"]
        g.create_dataset(f"layer_{i}/kernel:0", rng.standard_normal((3, 3, 2, 4)).astype(np.float32))
        g.create_dataset(f"layer_{i}/bias:0", rng.standard_normal(4).astype(np.float32))
    root.create_dataset("scalar", np.float64(3.25)); root.create_dataset("empty", np.zeros((0, 3), np.float32)); root.create_dataset("ints", np.arange(7, dtype=np.uint8))
    data = H5.to_bytes(root)
    tree_equal(root, H5.read_file(data))
    w = walk(data)
    assert sum(1 for v in w.values() if v[0] == "dataset") == 2 * nchild + 3 and len(w) == 1 + 3 * nchild + 3 + nchild
    if nchild:
        k = root[f"layer_{nchild - 1}/layer_{nchild - 1}/kernel:0"]
        assert w[f"/layer_{nchild - 1}/layer_{nchild - 1}/kernel:0"] == ("dataset", (3, 3, 2, 4), "f", 4, k.astype("<f4").tobytes())
        assert [s.decode() for s in w["/"][1]["names"]] == [f"layer_{i}" for i in range(nchild)]
    assert w["/"][1]["backend"].tobytes() == b"tensorflow"


@pytest.mark.parametrize("arch", ["unet", "unetpp", "classifier"])
@pytest.mark.parametrize("full_model", [False, True])
def test_keras_weight_file_layout_and_round_trip(tmp_path, arch, full_model):
    hw = (32, 32)
    w = W.init_weights(5, 1, arch, hw)
    rng = np.random.default_rng(1)
    w = {k: (v + rng.standard_normal(v.shape).astype(np.float32) * 0.01) for k, v in w.items()}
    f = str(tmp_path / ("unet_covid_weights_dice_coeff.hdf5" if full_model else "unet_0.8954_cosine_annealer.h5"))      # T1:1044 / T1:1079
    W.save_weights(f, w, 1, arch, hw, full_model=full_model)
    tree = walk(open(f, "rb").read())
    pre = "/model_weights" if full_model else ""
    top = tree[pre or "/"][1]
    kn = W.keras_names(1, arch, hw)
    names = [s.decode() for s in top["layer_names"]]
    assert top["backend"].tobytes() == b"tensorflow" and top["keras_version"].tobytes() == b"2.3.1"
    for k, v in w.items():                                           # saving.py: dataset path = <layer>/<weight name>, weight name = '<layer>/<var>:0'
        ln = kn[k].split("/")[0]
        kind, shape, cls, size, raw = tree[f"{pre}/{ln}/{kn[k]}"]
        assert kind == "dataset" and shape == v.shape and (cls, size) == ("f", 4) and raw == v.astype("<f4").tobytes()
        assert ln in names and kn[k].encode() in list(tree[f"{pre}/{ln}"][1]["weight_names"])
    for ln in names:                                                 # every layer has a group, weight-less ones an empty weight_names
        assert tree[f"{pre}/{ln}"][0] == "group"
    if full_model:
        import json
        cfg = json.loads(tree["/"][1]["model_config"].tobytes().decode())
        assert cfg["class_name"] == ("Sequential" if arch == "classifier" else "Model")
        assert [l["config"]["name"] for l in cfg["config"]["layers"]] == names
    w2 = W.load_weights(f, 1, arch, hw)
    assert list(w2) == list(w) and all(np.array_equal(w[k], w2[k]) for k in w)


def test_keras_file_from_another_session_is_matched_by_order(tmp_path):
    """A model built after others in the same Keras session carries shifted auto-names (conv2d_20 ...): Keras loads by layer order
    (saving.py load_weights_from_hdf5_group); so does weights.load_weights when the names do not match."""
    w = W.init_weights(2)
    layers = W._layer_weight_lists(w, 1, "unet", None)
    ren = []
    for ln, ws in layers:
        base, num = ln.rsplit("_", 1)
        new = f"{base}_{int(num) + 19}"
        ren.append((new, [(wn.replace(ln + "/", new + "/"), a) for wn, a in ws]))
    f = str(tmp_path / "shifted.h5")
    H5.save_keras_weights(f, ren)
    w2 = W.load_weights(f)
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    # and a wrong graph is refused with the offending tensor named
    with pytest.raises(ValueError, match="layers with weights|shape"):
        W.load_weights(f, 1, "unetpp")


def test_reader_accepts_what_other_writers_emit():
    """object-header continuation blocks + NIL messages (libhdf5 adds attributes to an existing header that way), a user block
    in front of the superblock is NOT needed by Keras files; corrupt / foreign files fail with a clear error."""
    root = H5.Group(); root.attrs["a"] = "x"; root.create_dataset("d", np.arange(6, dtype=np.float32).reshape(2, 3))
    data = bytearray(H5.to_bytes(root))
    # move the root header's attribute message into a continuation block at the end of the file
    oh = int.from_bytes(data[64:72], "little")
    size = int.from_bytes(data[oh + 8:oh + 12], "little")
    p, msgs = oh + 16, []
    while p < oh + 16 + size:
        n = int.from_bytes(data[p + 2:p + 4], "little"); msgs.append((p, 8 + n)); p += 8 + n
    (ps, ns), (pa, na) = msgs                                          # symbol table, attribute
    cont = len(data)
    data += data[pa:pa + na]
    assert na >= 24
    data[pa:pa + 8] = struct.pack("<HHB3x", 0x10, 16, 0); data[pa + 8:pa + 24] = struct.pack("<QQ", cont, na)
    data[pa + 24:pa + na] = struct.pack("<HHB3x", 0, na - 32, 0) + b"\0" * (na - 32) if na > 24 else b""
    data[oh + 2:oh + 4] = struct.pack("<H", 4 if na > 24 else 3)
    data[40:48] = struct.pack("<Q", len(data))
    g = H5.read_file(bytes(data))
    assert g.attrs["a"].tobytes() == b"x" and np.array_equal(g["d"], np.arange(6, dtype=np.float32).reshape(2, 3))
    with pytest.raises(H5.H5FormatError):
        H5.read_file(b"PK\x03\x04" + b"\0" * 100)
    with pytest.raises(H5.H5FormatError):
        H5.read_file(bytes(data[:len(data) // 2]))                      # truncated
