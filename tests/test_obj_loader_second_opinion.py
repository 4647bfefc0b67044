"""rtmi_obj_load (the library's OBJ/MTL reader) against a second restatement of ObjLoader.swift in plain Python
(oracle/swift_walkthrough.py) on generated files: lines the reference ignores, quirks it has (quads keep three
indices, `usemtl` takes the last token, blank lines vanish before the MTL block scan sees them, indented keys are
not keys) and input it traps on.  Both must load the same triangles and materials, or both must refuse."""
import numpy as np
import pytest

import raytracer_amd as rt
from oracle import swift_walkthrough as sw

NUMBERS = ["0", "1", "-1", "2.5", "-0.125", "1e3", "-2.5E-2", "+3", ".5", "7.", "1e400", "-1e400", "nan", "inf", "0x1p3", "1e-320", "3.0000000000000004"]
BAD_NUMBERS = ["1/2", "a", "1,5", "--1", "1e", "0x", "1 e3"[:1] + "e"]
NAMES = ["red", "front", "m1", "Material.001", "x"]


def number(rng, bad=0.0):
    return str(rng.choice(BAD_NUMBERS)) if rng.random() < bad else str(rng.choice(NUMBERS))


def gap(rng):
    return " " * int(rng.choice([1, 1, 1, 2, 3]))


def obj_text(rng, bad):
    lines, nv = [], 0
    for _ in range(int(rng.integers(4, 40))):
        kind = rng.choice(["v", "v", "v", "f", "f", "usemtl", "comment", "other", "vt", "vn", "blank", "indented"])
        if kind == "v":
            n = 3 if rng.random() > 0.15 else int(rng.choice([2, 4])) if rng.random() < bad * 4 + 0.3 else 3
            lines.append("v" + "".join(gap(rng) + number(rng, bad) for _ in range(n)) + (" " if rng.random() < 0.2 else ""))
            nv += 1
        elif kind == "f":
            n = int(rng.choice([3, 3, 3, 4, 5, 2 if rng.random() < bad * 3 else 3]))
            hi = max(nv, 1) + (1 if rng.random() < bad * 2 else 0)
            idx = [str(int(rng.integers(1 if rng.random() > bad else 0, hi + 1))) for _ in range(n)]
            if rng.random() < bad:
                idx[0] = idx[0] + "/1/1"
            lines.append("f" + "".join(gap(rng) + i for i in idx))
        elif kind == "usemtl":
            lines.append(rng.choice(["usemtl " + rng.choice(NAMES), "usemtl  " + rng.choice(NAMES) + " " + rng.choice(NAMES), "usemtl", "usemtl\t" + rng.choice(NAMES)]))
        elif kind == "comment":
            lines.append("# v 1 2 3")
        elif kind == "other":
            lines.append(rng.choice(["g group", "o thing", "s off", "mtllib a.mtl", "vp 1 2", "v\t1 2 3", "f\t1 2 3", "vertex 1 2 3"]))
        elif kind == "vt":
            lines.append("vt " + number(rng, bad) + " " + number(rng, bad) + (" " + number(rng) if rng.random() < 0.3 else ""))
        elif kind == "vn":
            lines.append("vn " + " ".join(number(rng, bad) for _ in range(3 if rng.random() > bad else 2)))
        elif kind == "blank":
            lines.append(rng.choice(["", " ", "\t"]))
        else:
            lines.append(" " + rng.choice(["v 9 9 9", "f 1 1 1", "usemtl red"]))
    sep = "\r\n" if rng.random() < bad else "\n"
    return sep.join(lines) + (sep if rng.random() < 0.8 else "")


def mtl_text(rng, bad):
    lines = []
    for _ in range(int(rng.integers(1, 5))):
        if rng.random() < 0.3:
            lines.append(rng.choice(["# materials", "", "Kd 1 1 1", "d 0.5"]))
        head = rng.choice(["newmtl ", " newmtl ", "newmtl  "]) + rng.choice(NAMES)
        if rng.random() < bad:
            head = rng.choice(["newmtl", "newmtlX " + rng.choice(NAMES)])
        lines.append(head)
        keys = ["Ns", "d", "illum", "Kd", "Ka", "Ks"]
        rng.shuffle(keys)
        if rng.random() < bad * 2:
            keys = keys[:-1]
        if rng.random() < 0.3:
            keys.append(str(rng.choice(["Ns", "Kd", "illum"])))  # a second value replaces the first
        for k in keys:
            if k in ("Kd", "Ka", "Ks"):
                n = 3 if rng.random() > bad else int(rng.choice([2, 4]))
                val = " ".join(number(rng, bad) for _ in range(n))
            elif k == "illum":
                val = str(rng.choice(["0", "1", "2", "+2", "-1"])) if rng.random() > bad else str(rng.choice(["2.0", "x"]))
            else:
                val = number(rng, bad)
            indent = " " if rng.random() < bad else ""
            lines.append(indent + k + gap(rng) + val)
            if rng.random() < 0.15:
                lines.append(rng.choice(["", "# note", "Ni 1.45", "map_Kd a.png"]))
        lines.append(rng.choice(["", " ", "  \t", "# end"]))
    return "\n".join(lines) + "\n"


def same(a, b):
    return a == b or (a != a and b != b)


def compare(tmp_path, obj, mtl, tag):
    op, mp = tmp_path / f"{tag}.obj", tmp_path / f"{tag}.mtl"
    op.write_bytes(obj.encode())
    mp.write_bytes(mtl.encode())
    try:
        verts, faces = sw.read_obj_text(obj)
        tri_list = sw.create_triangle_list(verts, faces)
        mats = sw.read_mtl_text(mtl)
        expected = (verts, tri_list, mats)
    except sw.Trap:
        expected = None
    try:
        got = rt.load_obj(str(op), str(mp))
    except rt.RtmiError as e:
        assert e.status == rt._abi.RTMI_ERR_INVALID_ARGUMENT
        got = None
    assert (got is None) == (expected is None), f"{tag}: library {'refused' if got is None else 'loaded'}, walkthrough {'trapped' if expected is None else 'loaded'}\n{obj}\n---\n{mtl}"
    if got is None:
        return False
    tris, gm, names, nverts = got
    verts, tri_list, mats = expected
    assert nverts == len(verts) and len(tris) == len(tri_list)
    assert set(names) == set(mats) and len(names) == len(set(names))
    for name, (ns, d, illum, kd, ka, ks) in mats.items():
        m = gm[names.index(name)]
        assert same(float(m["Ns"]), ns) and same(float(m["d"]), d) and int(m["illum"]) == illum, (tag, name)
        for field, c in (("Kd", kd), ("Ka", ka), ("Ks", ks)):
            assert all(same(float(x), y) for x, y in zip(m[field], (c.red, c.green, c.blue))), (tag, name, field)
    for t, (pts, mname) in zip(tris, tri_list):
        assert all(same(float(t["v"][k][a]), pts[k][a]) for k in range(3) for a in range(3)), tag
        assert (names[t["material"]] if t["material"] >= 0 else None) == (mname if mname in mats else None), tag
    return True


def test_generated_files_load_alike(tmp_path):
    rng = np.random.default_rng(5)
    loaded = 0
    for k in range(400):
        bad = [0.0, 0.0, 0.02, 0.08][k % 4]
        loaded += compare(tmp_path, obj_text(rng, bad), mtl_text(rng, bad), f"g{k}")
    assert 120 <= loaded <= 380  # both outcomes are exercised


@pytest.mark.parametrize("obj,mtl", [
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\n\n\nusemtl a\nf 1 2 3\n", "newmtl a\nNs 1\n\nd 1\nillum 2\nKd 1 1 1\nKa 1 1 1\nKs 1 1 1\n"),  # an EMPTY line does not end a block: split drops it
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n", "newmtl a\nNs 1\n \nd 1\nillum 2\nKd 1 1 1\nKa 1 1 1\nKs 1 1 1\n"),   # a blank-but-not-empty line does: d is nil
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", "newmtl a\nNs 1\nd 1\nillum 2\nKd 1 1 1\nKa 1 1 1\nKs 1 1 1\nnewmtl b\n Ns 1\nd 1\nillum 2\nKd 1 1 1\nKa 1 1 1\nKs 1 1 1\n"),  # indented Ns is no key
    ("v 0 0 0 1\nv 1 0 0 1\nv 0 1 0 1\nf 1 2 3 1 2 3\n", ""),  # extra numbers are read and dropped
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf +1 2 3\n", ""),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\nvt 1\n", ""),      # vt with one number traps
])
def test_hand_picked_quirks(tmp_path, obj, mtl):
    compare(tmp_path, obj, mtl, "q")
