"""Command-line harness: OBJ (+MTL) -> PNG/PPM on one MI355X, the way the reference's UI does it
(loadObjFile, then startRT with the typed-in camera; ViewController.swift:84-105, :130-190).

    python -m raytracer_amd.render mesh.obj out.png [--mtl mesh.mtl] [--size 1000 1000]
           [--camera 0 0 40] [--direction 0 0 -1] [--focal-length 1] [--subdivision 1]
           [--nearest-hit] [--shadow-max-distance] [--shadows-enter-bounded] [--reflections-enter-bounded]

Lights default to the three the reference's window starts with (ViewController.swift:59-70).
"""
import argparse

import numpy as np

import raytracer_amd as rt
from raytracer_amd import _abi, scenes


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("obj")
    ap.add_argument("out")
    ap.add_argument("--mtl")
    ap.add_argument("--size", type=int, nargs=2, default=(1000, 1000), metavar=("W", "H"))
    ap.add_argument("--camera", type=float, nargs=3, default=scenes.VC_CAMERA["pos"])
    ap.add_argument("--direction", type=float, nargs=3, default=scenes.VC_CAMERA["direction"])
    ap.add_argument("--focal-length", type=float, default=scenes.VC_CAMERA["focal_length"])
    ap.add_argument("--subdivision", type=float, default=1.0, help="pixelSubdivision: 1, 0.5 (4 spp), 0.25 (16 spp)")
    ap.add_argument("--ambient", type=float, nargs=3, default=(0.2, 0.2, 0.2))
    ap.add_argument("--device", type=int, default=0)
    # extensions, not the reference's behaviour (include/rtmi.h)
    ap.add_argument("--nearest-hit", action="store_true", help="RTMI_HIT_NEAREST instead of the first hit in list order")
    ap.add_argument("--shadow-max-distance", action="store_true", help="only what lies before a light casts a shadow")
    ap.add_argument("--shadows-enter-bounded", action="store_true", help="shadow rays test the mesh inside its bounding sphere")
    ap.add_argument("--reflections-enter-bounded", action="store_true", help="reflection rays test the mesh inside its bounding sphere")
    args = ap.parse_args(argv)

    lights = np.concatenate([scenes.light(**scenes.L1), scenes.light(**scenes.L2), scenes.light(**scenes.L3)])
    scene = rt.Scene.from_obj(args.obj, args.mtl, lights, device=args.device)
    cam = scenes.make_camera(tuple(args.camera), tuple(args.direction), args.focal_length)
    flags = (_abi.RTMI_FLAG_SHADOW_MAX_DISTANCE if args.shadow_max_distance else 0) | \
            (_abi.RTMI_FLAG_SHADOWS_ENTER_BOUNDED if args.shadows_enter_bounded else 0) | \
            (_abi.RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED if args.reflections_enter_bounded else 0)
    params = scenes.make_params(args.size[0], args.size[1], args.subdivision, ambient=tuple(args.ambient), flags=flags,
                                hit_policy=_abi.RTMI_HIT_NEAREST if args.nearest_hit else _abi.RTMI_HIT_LIST_ORDER)
    rgba, _, cnt = scene.render(cam, params, want_hit_ids=False)
    scene.close()
    rt.write_image(args.out, rgba)
    print(f"{args.out}: {args.size[0]}x{args.size[1]}, {cnt['samples']} samples, {cnt['primary_hits']} hits, "
          f"{cnt['ms_total']:.2f} ms on the device")
    return cnt


if __name__ == "__main__":
    main()
