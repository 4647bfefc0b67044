"""Hand-built scenes shared by the emulated-kernel tests and the GPU tests."""
import numpy as np

from raytracer_amd import scenes


def lights_on_primary_rays(oracle_mod, cam, W, H, pixels):
    """Light positions that the primary rays of `pixels` hit EXACTLY: with the camera at the origin, 4*d is
    representable whenever d is, and (4*d - 0)/d is 4 in every component (Renderables.swift:80-91 asks for equal
    quotients)."""
    ul, u, v, ph, pv = oracle_mod.frame(cam, W, H)
    out = []
    for i, j in pixels:
        d = np.array(oracle_mod.pixel_to_world(ul, u, v, ph, pv, float(i), float(j)))
        d = d / np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        out.append(tuple(4.0 * d))
    return out


def light_hit_scene(oracle_mod):
    W, H = 24, 20
    cam = scenes.make_camera((0, 0, 0), (0.3, -0.2, -1.0), 2.0)
    pixels = [(5, 7), (11, 9), (17, 3), (8, 15)]
    locs = lights_on_primary_rays(oracle_mod, cam, W, H, pixels)
    lights = np.concatenate([scenes.light(loc=loc, specular=(0.8,) * 3, diffuse=(0.5,) * 3) for loc in locs])
    objects = np.concatenate([scenes.light_object(0), scenes.light_object(1), scenes.sphere_object((0.5, -0.3, -3.0), 0.8, 0),
                              scenes.light_object(2), scenes.light_object(3)])
    data = scenes.SceneData(lights=lights, materials=scenes.material(**scenes.VC_MATERIAL), objects=objects,
                            triangles=np.zeros(0, scenes.TRI_DTYPE))
    return data, cam, scenes.make_params(W, H, 1.0), pixels
