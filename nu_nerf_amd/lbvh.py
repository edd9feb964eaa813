"""Mesh closest-hit tracing on the GPU: HIP LBVH replacing the reference's OptiX path.

`LBVH.intersect(ray[N,6]) -> (hit f32[N], idx i32[N])` keeps the I/O contract of
`optix_mesh.intersect` (network/tracing_optix.py:154-158; device programs cuda/triangle.cu:48-99) without its
device->host->device round trips; `Scene` mirrors the part of `DiffRender.Scene` the stage-2 renderer uses
(network/DiffRender.py:318-360, :410-416, :539-549): angle-weighted vertex normals and the differentiable
re-intersection (u, v, t, interpolated normal) of the hit triangles.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib as L

MISS_INDEX = 10000000


class LBVH:
    def __init__(self, vertices, faces):
        L.require_cuda(vertices, faces)
        self.lib = L.load()
        self.lib.nu_lbvh_bytes.restype = ctypes.c_longlong
        self.V = vertices.detach().to(torch.float32).contiguous()
        self.F = faces.detach().to(torch.int32).contiguous()
        self.n_faces = int(self.F.shape[0])
        nbytes = self.lib.nu_lbvh_bytes(self.n_faces)
        self.buf = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=self.V.device)
        L.check(self.lib.nu_lbvh_build(L.ptr(self.V), int(self.V.shape[0]), L.ptr(self.F), self.n_faces, L.ptr(self.buf),
                                       ctypes.c_longlong(nbytes), L.stream()), "nu_lbvh_build")

    def intersect(self, ray, tmin=0.0, tmax=1e16, return_t=False):
        ray = ray.detach().to(torch.float32).contiguous()
        N = ray.shape[0]
        hit = torch.empty(N, device=ray.device)
        idx = torch.empty(N, dtype=torch.int32, device=ray.device)
        t = torch.empty(N, device=ray.device) if return_t else None
        L.check(self.lib.nu_lbvh_trace(L.ptr(self.buf), self.n_faces, L.ptr(ray), N, ctypes.c_float(tmin), ctypes.c_float(tmax),
                                       L.ptr(hit), L.ptr(idx), L.ptr(t), L.stream()), "nu_lbvh_trace")
        return (hit, idx, t) if return_t else (hit, idx)

    def intersect_brute(self, ray, tmin=0.0, tmax=1e16):
        """O(N*F) device sweep with the same triangle test (cross-check)."""
        ray = ray.detach().to(torch.float32).contiguous()
        N = ray.shape[0]
        hit = torch.empty(N, device=ray.device)
        idx = torch.empty(N, dtype=torch.int32, device=ray.device)
        L.check(self.lib.nu_brute_trace(L.ptr(self.V), L.ptr(self.F), self.n_faces, L.ptr(ray), N, ctypes.c_float(tmin),
                                        ctypes.c_float(tmax), L.ptr(hit), L.ptr(idx), L.ptr(None), L.stream()), "nu_brute_trace")
        return hit, idx


def corner_angles_and_face_normals(tri):
    """DiffRender.py:170-192."""
    u, v, w = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], tri[:, 2] - tri[:, 1]
    fn = torch.cross(u, v, dim=1)
    fn = fn / fn.norm(dim=1, keepdim=True)
    u, v, w = (x / x.norm(dim=1, keepdim=True) for x in (u, v, w))
    a0 = torch.acos(torch.clamp((u * v).sum(1), -1, 1))
    a1 = torch.acos(torch.clamp((-u * w).sum(1), -1, 1))
    return torch.stack([a0, a1, math.pi - a0 - a1], 1), fn


def dintersect(origin, direction, triangles, normals):
    """Differentiable Moeller-Trumbore u, v, t and interpolated unit normal (DiffRender.py:61-125)."""
    v0, v1, v2 = triangles[:, 0], triangles[:, 1], triangles[:, 2]
    e1, e2 = v1 - v0, v2 - v0
    pvec = torch.cross(direction, e2, dim=1)
    inv_det = 1 / (e1 * pvec).sum(1)
    tvec = origin - v0
    u = (tvec * pvec).sum(1) * inv_det
    qvec = torch.cross(tvec, e1, dim=1)
    v = (direction * qvec).sum(1) * inv_det
    t = (e2 * qvec).sum(1) * inv_det
    n = (1 - u - v)[:, None] * normals[:, 0] + u[:, None] * normals[:, 1] + v[:, None] * normals[:, 2]
    return u, v, t, n / n.norm(dim=1, keepdim=True)


def vertex_normals_and_curvature(Vc, Fc):
    """Angle-weighted unit vertex normals [V,3] (DiffRender.py:343-358) and per-vertex Gaussian curvature [V,1], on the host.

    Per-vertex quantities are sums over the incident faces; they are accumulated on the CPU in face order (index_add_ on the device
    adds with float atomics: the vertex normals -- and every refracted direction after them -- would differ by an ulp from run to
    run); a mesh is preprocessed once.  Curvature: DiffRender.py:331,360 takes PyMesh's "vertex_gaussian_curvature" and clips it to
    [-10, 10]; here angle defect 2 pi - sum of the corner angles at the vertex, over the vertex area (a third of the incident
    faces' area).  PyMesh is absent: pinned analytically (sphere: 1 / r^2), not against PyMesh output."""
    Vc, Fc = Vc.detach().to(torch.float32).cpu(), Fc.detach().to(torch.long).cpu()
    tri = Vc[Fc]
    ang, fn = corner_angles_and_face_normals(tri)
    vn = torch.zeros_like(Vc)
    vn.index_add_(0, Fc.reshape(-1), (ang[:, :, None] * fn[:, None, :]).reshape(-1, 3))
    normals = vn / vn.norm(dim=1, keepdim=True)
    area = 0.5 * torch.linalg.norm(torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=1), dim=1)
    ang_sum = torch.zeros(len(Vc), dtype=ang.dtype).index_add_(0, Fc.reshape(-1), ang.reshape(-1))
    v_area = torch.zeros_like(ang_sum).index_add_(0, Fc.reshape(-1), (area / 3.0)[:, None].expand(-1, 3).reshape(-1))
    curv = torch.clamp((2.0 * math.pi - ang_sum) / v_area.clamp_min(1e-20), -10.0, 10.0)[:, None]
    return normals, curv


class Scene:
    """Triangle mesh + LBVH with the reference Scene's tracing surface (DiffRender.py:318-360, :539-549)."""

    def __init__(self, vertices, faces):
        self.vertices = vertices.detach().to(torch.float32).contiguous()
        self.faces = faces.detach().to(torch.long).contiguous()
        self.bvh = LBVH(self.vertices, self.faces)
        dev = self.vertices.device
        normals, curv = vertex_normals_and_curvature(self.vertices.cpu(), self.faces.cpu())
        self.normals, self.gaussian_curvatures = normals.to(dev), curv.to(dev)

    def intersect(self, origin, direction):
        hit, idx = self.bvh.intersect(torch.cat([origin, direction], 1))
        return idx.to(torch.long), hit > 0

    def Dintersect(self, origin, direction):
        """-> dict(u, v, t, n, point, faces_ind), hitted mask."""
        faces_ind, hitted = self.intersect(origin, direction)
        f = self.faces[faces_ind[hitted]]
        o, d = origin[hitted], direction[hitted]
        u, v, t, n = dintersect(o, d, self.vertices[f], self.normals[f])
        return dict(u=u, v=v, t=t, n=n, point=o + t[:, None] * d, origin=o, direction=d, faces_ind=faces_ind[hitted]), hitted


def dintersect_hip(scene, eng, origin, direction, curvature=False):
    """Scene.Dintersect on the HIP kernels (nu_lbvh_trace + nu_s2_hit_fwd/_bwd): -> dict(n, point, t, faces_ind, hit_idx, d
    [, g_k]), hitted mask.  curvature=True adds the interpolated per-vertex Gaussian curvature (DiffRender.py:116,
    Intersection.g_k).  ONE device -> host read (the number of hits): the hit rows are gathered by index, `hit_idx` and the
    gathered directions `d` are handed on so that callers need not ask again."""
    from . import stage2_ops as O
    faces_ind, hitted = scene.intersect(origin, direction)
    hit_idx = hitted.nonzero().flatten()
    fi = faces_ind.index_select(0, hit_idx)
    o, d = origin.index_select(0, hit_idx), direction.index_select(0, hit_idx)
    if curvature:
        point, n, t, gk = O.hit(eng, scene, o, d, fi, curvature=True)
        return dict(n=n, point=point, t=t, faces_ind=fi, hit_idx=hit_idx, d=d, g_k=gk), hitted
    point, n, t = O.hit(eng, scene, o, d, fi)
    return dict(n=n, point=point, t=t, faces_ind=fi, hit_idx=hit_idx, d=d), hitted


def icosphere(subdiv=2, radius=0.5):
    """Unit icosahedron subdivided `subdiv` times (20 * 4^subdiv faces): build-generated stand-in for the stage-1 mesh."""
    p = (1.0 + 5 ** 0.5) / 2.0
    v = np.array([[-1, p, 0], [1, p, 0], [-1, -p, 0], [1, -p, 0], [0, -1, p], [0, 1, p], [0, -1, -p], [0, 1, -p],
                  [p, 0, -1], [p, 0, 1], [-p, 0, -1], [-p, 0, 1]], np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], np.int64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(subdiv):
        cache, verts, nf = {}, list(v), []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = (verts[a] + verts[b]) / 2
                verts.append(m / np.linalg.norm(m))
                cache[k] = len(verts) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.asarray(verts), np.asarray(nf, np.int64)
    return (v * radius).astype(np.float32), f.astype(np.int32)
