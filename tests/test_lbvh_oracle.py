"""CPU checks of the brute-force mesh-tracing oracle against an independent scalar restatement on tiny inputs
(known-answer cases: the reference has no fixtures for its OptiX path)."""
import numpy as np

from oracle.lbvh_oracle import brute_force_closest_hit, MISS_INDEX


def _scalar_hit(V, F, o, d):
    best, bid = None, MISS_INDEX
    f32 = np.float32
    for fi, (a, b, c) in enumerate(F):
        v0, v1, v2 = V[a], V[b], V[c]
        e1, e2 = v1 - v0, v2 - v0
        pv = np.array([d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]], f32)
        det = f32(f32(e1[0] * pv[0] + e1[1] * pv[1]) + e1[2] * pv[2])
        if det == 0:
            continue
        inv = f32(1.0) / det
        tv = o - v0
        u = f32(f32(f32(tv[0] * pv[0] + tv[1] * pv[1]) + tv[2] * pv[2]) * inv)
        if not (0 <= u <= 1):
            continue
        qv = np.array([tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]], f32)
        v = f32(f32(f32(d[0] * qv[0] + d[1] * qv[1]) + d[2] * qv[2]) * inv)
        if not (v >= 0 and f32(u + v) <= 1):
            continue
        t = f32(f32(f32(e2[0] * qv[0] + e2[1] * qv[1]) + e2[2] * qv[2]) * inv)
        if t > 0 and t < 1e16 and (best is None or t < best):
            best, bid = t, fi
    return bid, best


def test_known_answers_on_a_tetrahedron():
    V = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    F = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], np.int32)
    rays = np.array([[0.2, 0.2, -1, 0, 0, 1],      # enters through the z=0 face (0), exits through the slanted face
                     [0.2, 0.2, 0.1, 0, 0, 1],     # starts inside: only the slanted face (3)
                     [5, 5, 5, 0, 0, 1],           # miss
                     [0.2, 0.2, 2, 0, 0, -1]], np.float32)   # from above: slanted face first
    hit, idx, t = brute_force_closest_hit(V, F, rays)
    assert hit.tolist() == [1, 1, 0, 1]
    assert idx.tolist() == [0, 3, MISS_INDEX, 3]
    np.testing.assert_allclose(t[[0, 1, 3]], [1.0, 0.5, 1.4], rtol=1e-6)


def test_vectorised_oracle_equals_scalar_restatement_bitwise():
    from nu_nerf_amd.lbvh import icosphere
    V, F = icosphere(1, 0.5)
    g = np.random.Generator(np.random.PCG64(3))
    o = g.normal(size=(200, 3)).astype(np.float32)
    o = (o / np.linalg.norm(o, axis=1, keepdims=True) * g.uniform(0.0, 2.0, (200, 1))).astype(np.float32)
    d = g.normal(size=(200, 3)).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    hit, idx, t = brute_force_closest_hit(V, F, np.concatenate([o, d], 1))
    for i in range(200):
        bid, bt = _scalar_hit(V, F, o[i], d[i])
        assert idx[i] == bid
        if bid != MISS_INDEX:
            assert t[i] == bt       # same float32 bits


def test_shared_edge_tie_goes_to_lowest_face_id():
    # two coplanar triangles sharing the edge x = y; a ray through that edge hits both at the same t
    V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    F = np.array([[0, 2, 3], [0, 1, 2]], np.int32)
    rays = np.array([[0.5, 0.5, 1, 0, 0, -1]], np.float32)
    hit, idx, t = brute_force_closest_hit(V, F, rays)
    assert hit[0] == 1 and idx[0] == 0 and t[0] == 1.0
