"""The conservativeness claims behind the FP32 filters, checked on the CPU.

The GPU kernels nominate (ray, triangle) pairs with FP32 tests and decide them in FP64; hit ids are
exact only if no FP32 test ever rejects a pair the reference accepts.  This file restates the
operand construction of k_pack_write / k_pack_tree / merge_cones / tree_sphere_records (kernels_prepare.h,
rtmi_tree.h) in numpy and evaluates the tests the way the kernels do (float32 operands, fused
multiply-add chains), for rays the ORACLE says hit: every such ray must pass the triangle's cone,
the cone of every ancestor, the edge test, and (for arbitrary origins) the bounding spheres.
It validates the mathematics and the margins, not the device code itself (the GPU parity tests do that).
"""
import numpy as np
import pytest

from raytracer_amd import scenes

F = np.float32


def fma32(a, b, c):
    """float32 fma: the product of two float32 is exact in float64."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def dot3_fma(w, d):
    """fma(wz, dz, fma(wy, dy, wx*dx)) as the kernels evaluate it."""
    q = (w[..., 0].astype(np.float64) * d[..., 0].astype(np.float64)).astype(F)
    q = fma32(w[..., 1], d[..., 1], q)
    return fma32(w[..., 2], d[..., 2], q)


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def triangle_cone(v, o):
    """axis and cos(theta) of the cone around the three corner directions (k_pack_tree)."""
    a = unit(v - o)                       # [T,3,3]
    total = a.sum(axis=1)
    axis = unit(total)                    # [T,3]
    cos_t = np.einsum("tkc,tc->tk", a, axis).min(axis=1)
    # narrower cones where they exist: through all three corners, or over one side; each reaches to its farthest corner
    cands = [np.cross(a[:, 1] - a[:, 0], a[:, 2] - a[:, 0]), a[:, 0] + a[:, 1], a[:, 1] + a[:, 2], a[:, 2] + a[:, 0]]
    for cand in cands:
        length = np.linalg.norm(cand, axis=1)
        ok = length > 1e-12
        ax = cand / np.where(ok, length, 1.0)[:, None]
        ax = np.where((np.einsum("tc,tc->t", ax, total) < 0)[:, None], -ax, ax)
        ck = np.einsum("tkc,tc->tk", a, ax).min(axis=1)
        better = ok & (ck > cos_t)
        axis = np.where(better[:, None], ax, axis)
        cos_t = np.where(better, ck, cos_t)
    return axis, cos_t


def cone_vector(axis, cos_t):
    den = cos_t - 2.0 ** -19
    w = (axis / den[:, None]).astype(F)
    wide = ~(den >= 0.0625)
    return w, wide


def fit(axes, corners):
    """node_axis + node_finish (levels 1 to 3 of k_pack_tree): the cone around the normalised sum of the children's axes
    that holds every corner direction below the node.  corners: [n, 3] unit vectors."""
    p = unit(axes.sum(axis=0))
    return p, (corners @ p).min() - 1e-12


def merge(axes, cos_c):
    """merge_cones (levels 4 and up): cone around the normalised axis sum that contains every child cone."""
    p = unit(axes.sum(axis=0))
    ca = axes @ p
    sa = np.linalg.norm(np.cross(np.broadcast_to(p, axes.shape), axes), axis=1)
    sin_c = np.sqrt(np.maximum(0.0, 1.0 - cos_c ** 2))
    cmin = (ca * cos_c - sa * sin_c).min() - 1e-12
    return p, cmin


def edge_vectors(v, o):
    A, B, C = v[:, 0] - o, v[:, 1] - o, v[:, 2] - o
    M = np.stack([np.cross(A, B), np.cross(B, C), np.cross(C, A)], axis=1)  # [T,3,3]
    L = np.abs(o).sum() + np.abs(v).sum(axis=2).max(axis=1)
    E = 16.0 * 2.0 ** -24 * np.abs(M).sum(axis=2) + 2.0 ** -40 * (L * L)[:, None]
    return (M / E[:, :, None]).astype(F)


def hitting_rays(oracle_mod, tri, o, rng, n):
    """Unit directions from o that the oracle's Triangle.intersections accepts, aimed at random points
    of the triangle including its edges and corners."""
    w = rng.dirichlet((0.3, 0.3, 0.3), size=n)
    w[: n // 4] = np.eye(3)[rng.integers(0, 3, n // 4)]                       # corners
    k = rng.integers(0, 3, n // 4)
    e = rng.random(n // 4)
    w[n // 4: n // 2] = 0.0
    w[np.arange(n // 4, n // 2), k] = e
    w[np.arange(n // 4, n // 2), (k + 1) % 3] = 1.0 - e                        # edges
    pts = w @ tri
    d = unit(pts - o)
    keep = np.array([oracle_mod.triangle_intersection(tri, o, di) is not None for di in d], bool)
    return d[keep]


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (1e-3, 0.0), (2.0e4, 3.0e5)])
def test_cone_tree_and_edge_test_never_reject_a_hit(oracle_mod, scale, offset, rt=None):
    rng = np.random.default_rng(7)
    vtx, faces = scenes.displaced_icosphere(2)                                  # 320 triangles
    v = (vtx * scale + offset)[faces]                                           # [T,3,3]
    o = np.array([0.3, -0.2, 4.0]) * scale + offset
    # front-facing triangles only (the pack culls the others: t > 0 fails for every ray from o)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    front = np.einsum("tc,tc->t", n, v[:, 0] - o) < 0
    v = v[front]
    axis, cos_t = triangle_cone(v, o)
    w, wide = cone_vector(axis, cos_t)
    m = edge_vectors(v, o)
    # ancestors: groups of 8 consecutive triangles and groups of 8 of those, fitted to the corners below them; above
    # that a cone around the cones (as the levels that span several workgroups are built)
    corners = unit(v - o)                                                       # [T,3,3]
    groups = [np.arange(i, min(i + 8, len(v))) for i in range(0, len(v), 8)]
    lvl1 = [fit(axis[g], corners[g].reshape(-1, 3)) for g in groups]
    lvl2 = []
    for i in range(0, len(lvl1), 8):
        ax = np.array([p for p, _ in lvl1[i:i + 8]])
        lvl2.append(fit(ax, corners[i * 8:(i + 8) * 8].reshape(-1, 3)))
    lvl3 = [merge(np.array([p for p, _ in lvl2]), np.array([c for _, c in lvl2]))]
    # the fitted cones are no wider than cones around the children's cones
    for g, (p, c) in zip(groups, lvl1):
        assert c >= merge(axis[g], cos_t[g])[1] - 1e-9
    total = 0
    for t in range(0, len(v), 2):
        d64 = hitting_rays(oracle_mod, v[t], o, rng, 64)
        if len(d64) == 0:
            continue
        d = d64.astype(F)
        total += len(d)
        if not wide[t]:
            assert (dot3_fma(w[t][None], d) >= 1.0).all(), "triangle cone rejected a hit"
        for (p, c), idx in ((lvl1[t // 8], t // 8), (lvl2[t // 64], t // 64), (lvl3[0], 0)):
            if c >= 0.25:  # otherwise the node is 'always descend'
                wn, _ = cone_vector(p[None], np.array([c]))
                assert (dot3_fma(wn, d) >= 1.0).all(), "an ancestor cone rejected a hit"
        s = np.stack([dot3_fma(m[t, i][None], d) for i in range(3)], axis=1).max(axis=1)
        assert (s <= 1.0).all(), "edge test rejected a hit"
    assert total > 1000


def test_bounding_spheres_never_reject_a_hit_from_any_origin(oracle_mod):
    rng = np.random.default_rng(11)
    vtx, faces = scenes.displaced_icosphere(2)
    v = vtx[faces]
    c0 = 0.5 * (v.reshape(-1, 3).min(axis=0) + v.reshape(-1, 3).max(axis=0))
    r_o1 = 1.01 * np.abs(np.concatenate([v.reshape(-1, 3), [[50, 50, 50], [-50, -50, -50]]]) - c0).max(axis=0).sum()

    def sphere_entry(c, r):
        s1 = np.abs(c).sum() + r_o1
        nk = r * r * (1 + 1e-6) + 64 * 2.0 ** -24 * s1 * s1 - (c * c).sum()
        return c.astype(F), F(nk)

    cen = v.mean(axis=1) - c0
    rho = np.linalg.norm(v - c0 - cen[:, None], axis=2).max(axis=1) * (1 + 1e-12)
    total = 0
    for t in range(0, len(v), 5):
        # origins on other triangles of the mesh and far away, rays aimed at this triangle
        for o in (v[(t * 7 + 3) % len(v)].mean(axis=0), np.array([30.0, -20.0, 45.0]), v[t].mean(axis=0) + 1e-3):
            d64 = hitting_rays(oracle_mod, v[t], o, rng, 32)
            if len(d64) == 0:
                continue
            total += len(d64)
            d = d64.astype(F)
            of = (o - c0).astype(F)
            ndo = -dot3_plain(d, of)
            noo = -dot3_plain(of[None], of)
            # leaf sphere and the sphere of its group of 8
            g = np.arange(t // 8 * 8, min(t // 8 * 8 + 8, len(v)))
            gc = cen[g].mean(axis=0)
            gr = np.linalg.norm((v[g] - c0).reshape(-1, 3) - gc, axis=1).max() * (1 + 1e-12)  # reaches the farthest corner below the node
            for c, r in ((cen[t], rho[t]), (gc, gr)):
                cf, nk = sphere_entry(c, r)
                A = fma32(np.broadcast_to(cf[2], d[:, 2].shape), d[:, 2], fma32(np.broadcast_to(cf[1], d[:, 1].shape), d[:, 1],
                          fma32(np.broadcast_to(cf[0], d[:, 0].shape), d[:, 0], ndo)))
                o2 = (2 * of).astype(F)
                B = fma32(cf[2:3], o2[2:3], fma32(cf[1:2], o2[1:2], fma32(cf[0:1], o2[0:1], noo)))
                U = fma32(A, A, np.broadcast_to(B, A.shape)) + nk
                assert not (U < 0).any(), "bounding sphere rejected a hit"
    assert total > 1000


def dot3_plain(a, b):
    """dx*ox + dy*oy + dz*oz in float32, left to right (top_tree_trace's ndo/noo)."""
    q = (a[..., 0] * b[..., 0]).astype(F)
    q = (q + (a[..., 1] * b[..., 1]).astype(F)).astype(F)
    return (q + (a[..., 2] * b[..., 2]).astype(F)).astype(F)
