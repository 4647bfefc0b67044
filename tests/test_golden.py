"""Golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle)
replayed against the oracle on CPU and against librtmi.so on the GPU.  Bar: bit-exact hit ids,
bit-exact RGBA8 (the FP64 decision path reproduces the reference arithmetic; the only libm
difference is pow(), which would show up as at most 1 LSB in a channel)."""
import numpy as np
import pytest

from golden_util import COUNTER_KEYS, golden_names, load_golden

NAMES = golden_names()


def test_fixture_set_is_complete():
    assert len(NAMES) >= 8
    for required in ("single_sphere_64x48", "two_sphere_64x48", "two_tri_list_order_32x32",
                     "ico2_bounded_s1_48x40", "ico2_bounded_s05_48x40", "ico2_bounded_s025_48x40",
                     "ico2_toplevel_3L_40x40", "mixed_3L_64x48"):
        assert required in NAMES


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(oracle_mod, name):
    data, cam, params, rgba, hit, counters = load_golden(name)
    o_rgba, o_hit, o_cnt = oracle_mod.render(data, cam, params)
    assert (o_hit == hit).all()
    assert (o_rgba == rgba).all()
    assert {k: o_cnt[k] for k in COUNTER_KEYS} == counters


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_golden(rt, name):
    data, cam, params, rgba, hit, counters = load_golden(name)
    sc = rt.Scene(data)
    g_rgba, g_hit, g_cnt = sc.render(cam, params)
    sc.close()
    assert (g_hit == hit).all(), f"{int((g_hit != hit).sum())} hit-id mismatches"
    diff = np.abs(g_rgba.astype(int) - rgba.astype(int))
    assert diff.max() <= 1, "RGBA8 tolerance is 1 LSB (device pow vs libm pow)"
    assert (diff > 0).sum() == 0, "observed: bit-exact on every fixture"
    for k in ("samples", "rays_reflection", "rays_shadow", "sphere_tests", "primary_hits"):
        assert g_cnt[k] == counters[k], k
