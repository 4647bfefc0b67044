"""Writes the stand-in mesh (the displaced icosphere of raytracer_amd.scenes) as the kind of OBJ/MTL pair the reference's
parser accepts (ObjLoader.swift:160-201, :95-158): `v x y z` lines, one `usemtl`, triangulated `f a b c` lines with
1-based plain vertex indices, LF line ends; the MTL holds the material the reference builds in startRT
(ViewController.swift:155).  Coordinates are written with 17 significant digits, so the loader reads back the same
doubles and the file renders to the same frame, bit for bit, as the generated mesh.

    python tools/write_mesh_obj.py OUT.obj [--level 4]      (level 4: 5,120 triangles; 7: 327,680)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scenes  # noqa: E402


def write_mesh(obj_path, level=4, material_name="pig"):
    v, f = scenes.displaced_icosphere(level)
    mtl_path = os.path.splitext(obj_path)[0] + ".mtl"
    with open(obj_path, "w", newline="\n") as out:
        out.write(f"# displaced icosphere level {level}: {len(v)} vertices, {len(f)} triangles\n")
        out.write(f"mtllib {os.path.basename(mtl_path)}\n")
        out.writelines(f"v {x!r} {y!r} {z!r}\n" for x, y, z in v.tolist())
        out.write(f"usemtl {material_name}\n")
        out.writelines(f"f {a + 1} {b + 1} {c + 1}\n" for a, b, c in f.tolist())
    m = scenes.VC_MATERIAL
    with open(mtl_path, "w", newline="\n") as out:
        out.write(f"newmtl {material_name}\nNs {m['Ns']!r}\nd {m['d']!r}\nillum {m['illum']}\n")
        for key in ("Kd", "Ka", "Ks"):
            out.write(f"{key} " + " ".join(repr(float(c)) for c in m[key]) + "\n")
    return obj_path, mtl_path, len(f)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--level", type=int, default=4)
    a = ap.parse_args()
    obj, mtl, n = write_mesh(a.out, a.level)
    print(f"{obj} + {mtl}: {n} triangles")
