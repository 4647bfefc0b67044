"""Host logic of the cone tree: the spatial slot order (no device needed)."""
import numpy as np
import pytest

from raytracer_amd import scenes


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 63, 64, 65, 511, 513, 5120])
def test_slot_order_is_a_permutation_with_padding_at_the_end(n):
    import raytracer_amd as rt
    rng = np.random.default_rng(n)
    tris = np.zeros(n, scenes.TRI_DTYPE)
    tris["v"] = rng.normal(size=(n, 3, 3))
    order = rt.tree_slot_order(tris)
    assert len(order) == max(8, (n + 7) // 8 * 8)
    assert sorted(order[order >= 0].tolist()) == list(range(n))
    assert (order[:n] >= 0).all() and (order[n:] == -1).all()  # only the last leaf is padded


def test_leaves_and_nodes_are_spatially_compact():
    import raytracer_amd as rt
    data = scenes.mesh_scene(4, "B", 1, bounding_sphere=rt.bounding_sphere)  # 5,120 triangles on a unit-size mesh
    tris = data.triangles
    # shuffling the list must not matter: the order is spatial
    perm = np.random.default_rng(0).permutation(len(tris))
    for t in (tris, tris[perm]):
        order = rt.tree_slot_order(t)
        cen = t["v"].mean(axis=1)[order]
        for size, bound in ((8, 0.35), (64, 0.9)):
            groups = cen.reshape(-1, size, 3)
            diag = np.linalg.norm(groups.max(axis=1) - groups.min(axis=1), axis=1)
            assert diag.mean() < bound, (size, diag.mean())  # the whole mesh spans about 4.3


def test_degenerate_input():
    import raytracer_amd as rt
    tris = np.zeros(20, scenes.TRI_DTYPE)
    tris["v"][3] = np.nan  # non-finite centroids sort as 0
    tris["v"][5] = np.inf
    order = rt.tree_slot_order(tris)
    assert sorted(order[order >= 0].tolist()) == list(range(20))
