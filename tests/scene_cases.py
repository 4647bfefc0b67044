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


def toplevel_mesh_with_company(n_lights, level=2):
    """A non-convex mesh listed at top level, a sphere with a material beside it and a light in the object list:
    reflection rays find the mesh, the sphere and the light; shadow rays are blocked by mesh and sphere.
    n_lights point lights on a ring (more than 16 takes the shading kernel that traces its own rays)."""
    import math
    from raytracer_amd import scenes
    v, f = scenes.displaced_icosphere(level)
    tris = scenes.triangles_from_vertices(v, f, 0)
    tris["material"][::3] = 1
    lights = np.concatenate([
        scenes.light(loc=(6.0 * math.cos(2.4 * k), 3.0 + 0.37 * k, 6.0 * math.sin(2.4 * k)), specular=(0.5, 0.5, 0.5),
                     diffuse=(1.7 / n_lights, 1.5 / n_lights, 1.6 / n_lights)) for k in range(n_lights)])
    mats = np.concatenate([scenes.material(**scenes.VC_MATERIAL),
                           scenes.material(Ns=5.0, d=1.0, illum=2, Kd=(0.8, 0.3, 0.2), Ka=(0.2, 0.5, 0.9), Ks=(0.4, 0.9, 0.1))])
    objects = np.concatenate([
        scenes.sphere_object((1.6, 0.3, 0.4), 0.5, 1),
        scenes.triangles_object(0, len(tris)),
        scenes.light_object(0),
    ])
    return scenes.SceneData(lights=lights, materials=mats, objects=objects, triangles=tris)


def shadow_extension_cases(bounding_sphere):
    """Scenes on which the two shadow-ray extensions change pixels: a non-convex bounded mesh with a material (its
    folds shadow one another once shadow rays enter the bounding sphere), lights between objects (a sphere or a
    top-level triangle behind a light stops shadowing with the distance limit), a light listed as an object."""
    bounded = scenes.mesh_scene(1, "B", 3, bounding_sphere=bounding_sphere, shuffle_seed=4)
    bounded.triangles["material"] = 0
    company = scenes.mesh_scene(1, "B", 2, bounding_sphere=bounding_sphere)
    company.triangles["material"] = 0
    # a light between the mesh and a sphere, and the same light listed as an object; top-level triangles behind another
    lights = np.concatenate([scenes.light(loc=(2.2, 0.3, 0.4), specular=(0.8, 0.8, 0.8), diffuse=(0.5, 0.5, 0.5)),
                             scenes.light(loc=(-0.2, 2.0, 0.6), specular=(0.6, 0.7, 0.8), diffuse=(0.5, 0.4, 0.3))])
    wall = np.zeros(2, scenes.TRI_DTYPE)
    wall["v"][0] = [(-2.5, 3.5, -2.0), (2.5, 3.5, 2.0), (2.5, 3.5, -2.0)]
    wall["v"][1] = [(-2.5, 3.5, -2.0), (-2.5, 3.5, 2.0), (2.5, 3.5, 2.0)]
    wall["material"] = 0
    n = len(company.triangles)
    company = scenes.SceneData(lights=lights, materials=company.materials,
                               triangles=np.concatenate([company.triangles, wall]),
                               objects=np.concatenate([company.objects, scenes.sphere_object((3.6, 0.5, 0.6), 0.8, 0),
                                                       scenes.light_object(0), scenes.triangles_object(n, 2)]))
    # the first half of a mesh listed by two bounding spheres, the second of them smaller than what it lists (a shadow
    # ray that misses that sphere does not test its children, wherever they are) and carrying a material
    shared = scenes.mesh_scene(1, "B", 2, bounding_sphere=bounding_sphere)
    shared.triangles["material"] = 0
    half = len(shared.triangles) // 2
    shared.objects = np.concatenate([shared.objects, scenes.sphere_object((0.35, 0.3, 0.2), 0.55, 0, bounding=True, first=0, count=half)])
    # a sphere beside a bounding sphere that has a material and no children (the only bounding sphere of the scene, so
    # there is no tree of children at all): the reference shows it in the sphere's mirror, a reflection ray that enters
    # it finds nothing
    hollow = scenes.two_sphere_scene()
    hollow.objects["kind"][1] = 2
    hollow.objects["first"][1] = 0
    hollow.objects["count"][1] = 0
    return [(bounded, scenes.make_camera(**scenes.MESH_CAMERA)),
            (company, scenes.make_camera((0.8, 0.9, 5.0), (-0.05, -0.12, -1.0), 7.0)),
            (shared, scenes.make_camera((0.4, 0.2, 4.5), (-0.05, -0.03, -1.0), 9.0)),
            (hollow, scenes.make_camera(**scenes.VC_CAMERA))]
