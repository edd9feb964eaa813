"""Brute-force closest-hit oracle for the mesh tracer (TEST INFRASTRUCTURE).

Restates the semantics of the reference's OptiX programs (cuda/triangle.cu:48-99: closest hit, tmin = 0, tmax = 1e16,
no culling; miss -> (0.0, 10000000); hit -> (1.0, primitive index)) with an O(N*F) numpy sweep.  OptiX's own
ray/triangle arithmetic and tie-breaking are unspecified and cannot run here ("parity unpinned" against OptiX,
SURVEY.md 8(c)); bit-exactness of hit indices is therefore DEFINED against this oracle: Moeller-Trumbore in float32
with one rounding per operation in the order written below, ties in t resolved to the lowest face id, valid hits
0 < t < tmax.  numpy float32 array arithmetic rounds after every operation (no FMA contraction), which is what the
HIP kernel reproduces with __fmul_rn/__fadd_rn/__fsub_rn/__fdiv_rn.
"""
import numpy as np

MISS_INDEX = 10000000
f32 = np.float32


def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


def brute_force_closest_hit(V, F, rays, tmin=0.0, tmax=1e16, chunk=256):
    """V [Nv,3] f32, F [Nf,3] int, rays [N,6] f32 -> (hit f32[N], idx i32[N], t f32[N])."""
    V = np.asarray(V, f32)
    F = np.asarray(F, np.int64)
    rays = np.asarray(rays, f32)
    v0, v1, v2 = V[F[:, 0]][None], V[F[:, 1]][None], V[F[:, 2]][None]      # [1,F,3]
    e1, e2 = v1 - v0, v2 - v0
    N = rays.shape[0]
    hit = np.zeros(N, f32)
    idx = np.full(N, MISS_INDEX, np.int32)
    tt = np.zeros(N, f32)
    tmin, tmax = f32(tmin), f32(tmax)
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        for s in range(0, N, chunk):
            o = rays[s:s + chunk, None, :3]
            d = rays[s:s + chunk, None, 3:]
            pv = _cross(np.broadcast_to(d, (d.shape[0], e2.shape[1], 3)), e2)
            det = _dot(e1, pv)
            inv = f32(1.0) / det
            tv = o - v0
            u = _dot(tv, pv) * inv
            qv = _cross(tv, np.broadcast_to(e1, tv.shape))
            v = _dot(np.broadcast_to(d, qv.shape), qv) * inv
            t = _dot(np.broadcast_to(e2, qv.shape), qv) * inv
            ok = (det != 0) & (u >= 0) & (u <= 1) & (v >= 0) & ((u + v) <= 1) & (t > tmin) & (t < tmax)
            tm = np.where(ok, t, np.inf).astype(f32)
            best = tm.min(1)
            # lowest face id among the minima
            first = (tm == best[:, None]).argmax(1)
            any_hit = ok.any(1)
            hit[s:s + chunk] = any_hit.astype(f32)
            idx[s:s + chunk] = np.where(any_hit, first, MISS_INDEX).astype(np.int32)
            tt[s:s + chunk] = np.where(any_hit, best, 0).astype(f32)
    return hit, idx, tt
