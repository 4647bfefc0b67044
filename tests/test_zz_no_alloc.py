"""rtmi_render_device_noalloc: the variant of rtmi_render_device whose work buffers are slices of the library's
static device array.  It must give exactly what rtmi_render_device gives (hit ids and RGBA8 byte for byte, same
ray counters), on two calls in a row of which the second is the smaller frame, and refuse frames beyond its
stated limits.  The CPU part runs the emulated build of the kernels (tests/emu) through the same entry point."""
import numpy as np
import pytest
import torch  # before librtmi.so is loaded: one process, one HIP runtime (torch brings its own copy)

from raytracer_amd import _abi, scenes

PARITY_COUNTERS = ("samples", "rays_reflection", "rays_shadow", "sphere_tests", "primary_hits", "tri_exact_tests",
                   "shadow_rays_elided")


def mixed_scene(bounding_sphere):
    """A bounded mesh, a sphere with a material and a top-level triangle run: every kernel of the frame runs."""
    v, f = scenes.displaced_icosphere(2)
    mesh = scenes.triangles_from_vertices(v * 0.8, f, 0)
    floor = np.zeros(2, scenes.TRI_DTYPE)
    floor["v"][0] = [[-3.0, -1.2, -3.0], [3.0, -1.2, 3.0], [3.0, -1.2, -3.0]]
    floor["v"][1] = [[-3.0, -1.2, -3.0], [-3.0, -1.2, 3.0], [3.0, -1.2, 3.0]]
    floor["material"] = 0
    tris = np.concatenate([mesh, floor])
    c, r = bounding_sphere(mesh)
    objs = np.concatenate([scenes.sphere_object((1.6, 0.2, 0.0), 0.6, 0),
                           scenes.sphere_object(c, r, -1, bounding=True, first=0, count=len(mesh)),
                           scenes.triangles_object(len(mesh), 2)])
    lights = np.concatenate([scenes.light(**scenes.L1), scenes.light(**scenes.L2), scenes.light(**scenes.L3)])
    return scenes.SceneData(lights=lights, materials=scenes.material(**scenes.VC_MATERIAL), objects=objs, triangles=tris)


CAM = dict(pos=(0.5, 0.8, 5.0), direction=(-0.05, -0.12, -1.0), focal_length=8.0)
# (width, height, subdivision): 57,600 then 13,824 samples in whole blocks -- the second call is the smaller one
FRAMES = [(160, 90, 0.5), (96, 72, 1.0), (120, 60, 0.5)]


@pytest.mark.gpu
def test_noalloc_matches_render_device_on_consecutive_calls(rt, oracle_mod):
    import copy
    L = rt.lib()
    assert L.rtmi_noalloc_max_samples() == 65536
    for data in (mixed_scene(rt.bounding_sphere), scenes.mesh_scene(3, "B", 3, bounding_sphere=rt.bounding_sphere)):
        sc = rt.Scene(data)
        ref_sc = rt.Scene(data)  # a second scene object, so that neither path leans on the other's cached state
        cam = scenes.make_camera(**CAM)
        for search in (_abi.RTMI_SEARCH_CONE_TREE, _abi.RTMI_SEARCH_LIST_SCAN):
            for (W, H, sub) in FRAMES:
                p = scenes.make_params(W, H, sub, search=search)
                spp = rt.samples_per_pixel(sub)
                out = {}
                for name, scene, fn in (("noalloc", sc, sc.render_device_noalloc), ("device", ref_sc, ref_sc.render_device)):
                    rgba = torch.full((H, W, 4), 0x5A, dtype=torch.uint8, device="cuda")
                    hit = torch.full((H, W, spp), -77, dtype=torch.int32, device="cuda")
                    torch.cuda.synchronize()  # the fills are on torch's stream, the render on the scene's
                    cnt = fn(cam, p, rgba.data_ptr(), hit.data_ptr())
                    out[name] = (rgba.cpu().numpy(), hit.cpu().numpy(), cnt)
                a, b = out["noalloc"], out["device"]
                assert (a[1] == b[1]).all(), f"{W}x{H} sub {sub} search {search}: hit ids differ"
                assert (a[0] == b[0]).all(), f"{W}x{H} sub {sub} search {search}: RGBA8 differs"
                assert (b[1] != -77).any() and (b[1] >= 0).any()
                for k in PARITY_COUNTERS:
                    assert a[2][k] == b[2][k], (W, H, sub, search, k)
                # and against the oracle itself: hit ids and RGBA8 of the rendered (W-1) x (H-1) pixels, the rest untouched
                o_rgba, o_hit, o_cnt = oracle_mod.render(data, cam, copy.copy(p), threads=8)
                m = np.zeros((H, W), bool)
                m[:H - 1, :W - 1] = True
                assert (a[1][m] == o_hit[m]).all(), f"{W}x{H} sub {sub} search {search}: hit ids differ from the oracle's"
                assert (a[0][m] == o_rgba[m]).all(), f"{W}x{H} sub {sub} search {search}: RGBA8 differs from the oracle's"
                assert (a[0][~m] == 0x5A).all() and (a[1][~m] == -77).all()
                assert a[2]["primary_hits"] == o_cnt["primary_hits"] and a[2]["rays_shadow"] == o_cnt["rays_shadow"]
        # beyond the limits: refused with a message, nothing rendered, and the scene still works afterwards
        big = scenes.make_params(400, 300, 0.5)
        rgba = torch.zeros((300, 400, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        with pytest.raises(rt.RtmiError) as e:
            sc.render_device_noalloc(cam, big, rgba.data_ptr())
        assert e.value.status == _abi.RTMI_ERR_UNSUPPORTED and "static array" in str(e.value)
        assert int(rgba.max()) == 0
        wide = scenes.make_params(20000, 16, 1.0, window=(0, 0, 64, 8))  # few samples, but more table entries than fit
        rgba = torch.zeros((16, 20000, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        with pytest.raises(rt.RtmiError) as e:
            sc.render_device_noalloc(cam, wide, rgba.data_ptr())
        assert e.value.status == _abi.RTMI_ERR_UNSUPPORTED
        # the shadow-ray extensions build a tree of their own on first use: refused here
        small = torch.zeros((16, 16, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        with pytest.raises(rt.RtmiError) as e:
            sc.render_device_noalloc(cam, scenes.make_params(16, 16, 1.0, flags=_abi.RTMI_FLAG_SHADOWS_ENTER_BOUNDED), small.data_ptr())
        assert e.value.status == _abi.RTMI_ERR_UNSUPPORTED
        sc.render_device(cam, big, torch.zeros((300, 400, 4), dtype=torch.uint8, device="cuda").data_ptr())
        sc.close()
        ref_sc.close()


@pytest.mark.gpu
def test_noalloc_allocates_no_device_memory(rt):
    """hipMemGetInfo's free byte count stays where it was over 120 calls that alternate between two frame sizes and
    two searches (everything the caller needs -- output buffers -- exists before the first reading)."""
    data = mixed_scene(rt.bounding_sphere)
    sc = rt.Scene(data)
    cam = scenes.make_camera(**CAM)
    frames = []
    for (W, H, sub) in FRAMES[:2]:
        spp = rt.samples_per_pixel(sub)
        frames.append((W, H, sub, torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"),
                       torch.zeros((H, W, spp), dtype=torch.int32, device="cuda")))
    params = [scenes.make_params(W, H, sub, search=search) for (W, H, sub, _, _) in frames
              for search in (_abi.RTMI_SEARCH_CONE_TREE, _abi.RTMI_SEARCH_LIST_SCAN)]
    bufs = [f for f in frames for _ in range(2)]
    torch.cuda.synchronize()
    for p, (_, _, _, rgba, hit) in zip(params, bufs):  # first use of every kernel involved (code objects load lazily)
        sc.render_device_noalloc(cam, p, rgba.data_ptr(), hit.data_ptr())
    free0 = torch.cuda.mem_get_info()[0]
    first = None
    for i in range(120):
        p, (_, _, _, rgba, hit) = params[i % 4], bufs[i % 4]
        cnt = sc.render_device_noalloc(cam, p, rgba.data_ptr(), hit.data_ptr())
        if i == 0:
            first = (rgba.cpu().numpy().copy(), cnt["primary_hits"])
        assert torch.cuda.mem_get_info()[0] == free0, f"free device memory changed at call {i}"
    sc.render_device_noalloc(cam, params[0], bufs[0][3].data_ptr(), bufs[0][4].data_ptr())
    assert (bufs[0][3].cpu().numpy() == first[0]).all() and first[1] > 0
    sc.close()


@pytest.mark.gpu
def test_noalloc_calls_from_two_threads_take_turns(rt):
    """Two scenes on one GPU share the device's static array; the library serialises their calls, so two threads
    that call at the same time both get the frames a single thread gets."""
    import threading
    datas = [mixed_scene(rt.bounding_sphere), scenes.mesh_scene(3, "B", 3, bounding_sphere=rt.bounding_sphere)]
    cam = scenes.make_camera(**CAM)
    W, H, sub = FRAMES[0]
    p = scenes.make_params(W, H, sub)
    scs = [rt.Scene(d) for d in datas]
    bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in datas]
    torch.cuda.synchronize()
    want = []
    for sc, b in zip(scs, bufs):
        sc.render_device_noalloc(cam, p, b.data_ptr())
        want.append(b.cpu().numpy().copy())
    bad = []

    def work(k):
        for _ in range(25):
            b = bufs[k]
            scs[k].render_device_noalloc(cam, p, b.data_ptr())
            if not (b.cpu().numpy() == want[k]).all():
                bad.append(k)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad, f"frames of scene(s) {sorted(set(bad))} changed when two threads rendered at once"
    for sc in scs:
        sc.close()


def test_noalloc_emulated_matches_oracle(oracle_mod, tmp_path):
    """The same entry point in the CPU-emulated build, under AddressSanitizer: two frames in a row on one scene
    object (the static array keeps the first frame's contents), hit ids and RGBA8 equal to the oracle's."""
    import copy
    import emu_util
    emu_util.build()
    data = mixed_scene(oracle_mod.bounding_sphere)
    cams = [scenes.make_camera(**CAM), scenes.make_camera(pos=(0.2, 0.5, 4.5), direction=(0.0, -0.1, -1.0), focal_length=9.0)]
    for search in (_abi.RTMI_SEARCH_CONE_TREE, _abi.RTMI_SEARCH_LIST_SCAN):
        for (W, H, sub) in ((56, 40, 0.5), (40, 24, 1.0)):
            p = scenes.make_params(W, H, sub, search=search)
            frames, stats, _ = emu_util.run_emulated(data, cams, p, tmp_path / f"{search}_{W}", noalloc=True)
            for (rgba, hit, cnt), cam in zip(frames, cams):
                o_rgba, o_hit, o_cnt = oracle_mod.render(data, cam, copy.copy(p), threads=4)
                m = np.zeros((H, W), bool)
                m[:H - 1, :W - 1] = True
                assert (hit[m] == o_hit[m]).all() and (rgba[m] == o_rgba[m]).all()
                assert (rgba[~m] == 0x5A).all()
                assert cnt.primary_hits == o_cnt["primary_hits"] > 0
