"""Stage-2 renderer (zero-thickness variant): drop-in for `Stage2Renderer` of network/renderer_zerothick.py:868-2055
(`name2renderer['stage2']`, registry :2057-2060).

The light paths are a chain of SEGMENTS (`trace_segments`), rendered front to back (`render_segments`).  Everything numeric runs
on HIP kernels with hand-derived backward -- stage 2 needs d L / d position everywhere, because every sample position depends on
the learned index of refraction:
  * closest hit: HIP LBVH (nu_nerf_amd/lbvh.py) instead of OptiX; differentiable hit point / normal: nu_s2_hit_*;
  * IoR network, SDF / NeRF++ / predictor / material networks incl. second-order SDF terms and input gradients: MFMA GEMMs
    (nets.py: IorFn, SdfFn, NerfFn, StackFn, MaterialsFn; one parameter hub per engine and pass);
  * refraction / total internal reflection: nu_s2_refract_*;
  * sample placement: the stage-1 sampler kernels for the inner-SDF up-sampling, nu_s2_far_* for the NeRF++ importance pass;
  * the outer samples of all segments: nu_s2_seg_* + ONE NeRF++ pass; inner segment: nu_s2_neus_alpha_*;
  * shading: one kernel pair for the stacks' inputs (nu_s2_shade_encode_*) + predictor stacks + one BRDF-mix kernel pair
    (nu_s2_shade_combine_*, shading_glue.py);
  * per-segment composite in linear RGB with the running transmittance: nu_s2_composite_*.
torch is left with index bookkeeping (one nonzero per mask, index_select / index_add) and O(rays) glue.
state_dict() names/order equal the reference's (574 entries incl. the `color_network.stage1_network.*` aliases).
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import stage2_ops as O
from . import torch_glue as G
from .engine import Stage1Engine, addr
from .lbvh import Scene, dintersect_hip
from .nets import Stage1Nets
from .shading_glue import shade
from .renderer import (AppShadingNetwork, NeRFNetwork, NeROShapeRenderer, SDFNetwork, SingleVarianceNetwork, WNLinear)


class IoRNetwork(nn.Module):
    """Parameter layout of reference IoRNetwork / ThicknessNetwork (field.py:1046-1087): Sequential indices 0,2,4,5."""

    def __init__(self):
        super().__init__()
        self.module0 = nn.Sequential(WNLinear(39, 256), nn.ReLU(), WNLinear(256, 256), nn.ReLU(), WNLinear(256, 256),
                                     WNLinear(256, 1), nn.Sigmoid())

    def forward(self, x):
        # a parameter holder, like SDFNetwork: the networks run on the HIP GEMMs of the owning renderer's engine (nets.IorFn /
        # IorPairFn via Stage2Renderer.nets()); there is no eager evaluation of them in the package
        raise RuntimeError("IoRNetwork is a parameter holder: evaluate it through its Stage2Renderer (nets().ior / .thickness)")


class AppShadingNetworkS2(nn.Module):
    """Holds no parameters of its own; keeps the stage-1 network as a child like the reference (field.py:798-802), which is
    what puts the `color_network.stage1_network.*` aliases into state_dict()."""
    default_cfg = {'human_light': False, 'sphere_direction': True, 'light_pos_freq': 6, 'inner_init': -0.95,
                   'roughness_init': 0.0, 'metallic_init': 0.0, 'light_exp_max': 5.0, 'refrac_freq': 6}

    def __init__(self, cfg, stage1):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        self.stage1_network = stage1


def read_ply(path):
    """Minimal PLY reader (ascii / binary_little_endian, float xyz vertices, triangle faces)."""
    with open(path, 'rb') as fh:
        fmt, nv, nf, vprops, section = None, 0, 0, [], None
        while True:
            line = fh.readline().decode('ascii', 'replace').strip()
            tok = line.split()
            if tok[:1] == ['format']:
                fmt = tok[1]
            elif tok[:2] == ['element', 'vertex']:
                nv, section = int(tok[2]), 'v'
            elif tok[:2] == ['element', 'face']:
                nf, section = int(tok[2]), 'f'
            elif tok[:1] == ['property'] and section == 'v':
                vprops.append((tok[1], tok[2]))
            elif line == 'end_header':
                break
        if fmt == 'ascii':
            rows = [fh.readline().split() for _ in range(nv)]
            V = np.asarray([[float(r[0]), float(r[1]), float(r[2])] for r in rows], np.float32)
            Fc = np.asarray([[int(x) for x in fh.readline().split()[1:4]] for _ in range(nf)], np.int32)
            return V, Fc
        types = {'float': '<f4', 'float32': '<f4', 'double': '<f8', 'uchar': 'u1', 'uint8': 'u1', 'int': '<i4', 'uint': '<u4',
                 'short': '<i2', 'ushort': '<u2'}
        dt = np.dtype([(n, types[t]) for t, n in vprops])
        vert = np.frombuffer(fh.read(nv * dt.itemsize), dtype=dt, count=nv)
        V = np.stack([vert['x'], vert['y'], vert['z']], 1).astype(np.float32)
        fd = np.dtype([('n', 'u1'), ('i', '<i4', (3,))])
        Fc = np.frombuffer(fh.read(nf * fd.itemsize), dtype=fd, count=nf)['i'].astype(np.int32)
        return V, Fc


class Stage2Renderer(nn.Module):
    default_cfg = {**NeROShapeRenderer.default_cfg, 'train_ray_num': 1024, 'is_nerf': False}

    def __init__(self, cfg, training=True):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        self.is_nerf = self.cfg['is_nerf']
        self.IORs = nn.Parameter(torch.zeros(10))
        self.nerf_network = NeRFNetwork()
        self.stage1_network = NeROShapeRenderer(self._load_stage1_cfg(), training=False)
        self._load_stage1_ckpt()
        self.IORs_pred = IoRNetwork()
        self.IoRint_pred = IoRNetwork()
        self.thickness_pred = IoRNetwork()
        self.outer_nerf = NeRFNetwork()
        self.color_network = AppShadingNetworkS2(self.cfg['shader_config'], self.stage1_network)
        self.sdf_network_inner = SDFNetwork()
        self.deviation_network_inner = SingleVarianceNetwork(self.cfg['inv_s_init'])
        self.color_network_inner = AppShadingNetwork(self.cfg['shader_config'])
        self._init_own_parameters()
        self._mesh = self._load_mesh()
        self.scene = None
        self._nets = None
        if training:
            self._init_dataset()

    # ---- data: the ray-pool store of the stage-1 module (`database_name: synthetic/<n_rays>`; image databases are out of scope) ----
    _init_dataset = NeROShapeRenderer._init_dataset
    _shuffle_train_batch = NeROShapeRenderer._shuffle_train_batch
    # an image database loaded by the caller: the same device-resident ray store as stage 1 (renderer.set_ray_store)
    _construct_ray_batch = staticmethod(NeROShapeRenderer._construct_ray_batch)
    _construct_nerf_ray_batch = staticmethod(NeROShapeRenderer._construct_nerf_ray_batch)
    set_ray_store = NeROShapeRenderer.set_ray_store
    _test_batch_from_store = NeROShapeRenderer._test_batch_from_store
    _process_ray_batch = NeROShapeRenderer._process_ray_batch
    near_far_from_sphere = staticmethod(NeROShapeRenderer.near_far_from_sphere)
    get_human_coordinate_poses = NeROShapeRenderer.get_human_coordinate_poses

    # ---- construction helpers ---------------------------------------------------------------------
    def _load_stage1_cfg(self):
        c = self.cfg
        if 'stage1_cfg' in c:
            return dict(c['stage1_cfg'])
        import yaml
        with open(c['stage1_cfg_dir']) as fh:
            return yaml.safe_load(fh)

    def _load_stage1_ckpt(self):
        c = self.cfg
        if c.get('stage1_ckpt_dir') and os.path.exists(c['stage1_ckpt_dir']):
            ck = torch.load(c['stage1_ckpt_dir'], weights_only=True, map_location='cpu')
            self.stage1_network.load_state_dict(ck['network_state_dict'], strict=False)

    def _load_mesh(self):
        c = self.cfg
        if 'stage1_mesh_arrays' in c:
            V, Fc = c['stage1_mesh_arrays']
            return np.asarray(V, np.float32), np.asarray(Fc, np.int32)
        return read_ply(c['stage1_mesh_dir'])

    def _init_own_parameters(self):
        from .params import init_stage2_params
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        init = init_stage2_params(seed, seed + 1, self.cfg['shader_config'])
        sd = self.state_dict()
        with torch.no_grad():
            for k, v in init.items():
                if k.startswith(('stage1_network.', 'color_network.stage1_network.')):
                    continue
                sd[k].copy_(torch.as_tensor(np.asarray(v)).reshape(sd[k].shape))

    def load_param_dict(self, arrays):
        sd = self.state_dict()
        with torch.no_grad():
            for k, v in arrays.items():
                sd[k].copy_(torch.as_tensor(np.asarray(v)).reshape(sd[k].shape))

    def _apply(self, fn, *a, **k):
        self._nets = None
        self.scene = None
        return super()._apply(fn, *a, **k)

    def nets(self):
        """(stage-1 nets, inner nets) over two HIP engines + the LBVH scene; rebuilt after device moves."""
        dev = self.IORs.device
        if self._nets is None:
            s1 = self.stage1_network
            n1 = Stage1Nets(s1.engine(), s1._named())
            named = {}
            for k, p in self.sdf_network_inner.named_parameters():
                named['sdf_network.' + k] = p
            named['deviation_network.variance'] = self.deviation_network_inner.variance
            for k, p in self.color_network_inner.named_parameters():
                named['color_network.' + k] = p
            named['color_network.FG_LUT'] = self.color_network_inner.FG_LUT
            for k, p in s1.outer_nerf.named_parameters():      # placeholder: the inner engine never evaluates a NeRF++
                named['outer_nerf.' + k] = p
            for k, p in self.IORs_pred.module0.named_parameters():   # the IoR network rides on the inner engine's GEMMs
                named['ior_network.' + k] = p
            ecfg = dict(self.cfg)
            ecfg.update(self.color_network_inner.cfg)
            n2 = Stage1Nets(Stage1Engine(named, dev, ecfg), named)
            V, Fc = self._mesh
            self.scene = Scene(torch.from_numpy(V).to(dev), torch.from_numpy(Fc).to(dev))
            self._nets = (n1, n2)
        return self._nets

    def get_anneal_val(self, step):
        if self.cfg['anneal_end'] < 0:
            return 1.0
        return float(np.min([1.0, step / self.cfg['anneal_end']]))

    compute_rgb_loss = NeROShapeRenderer.compute_rgb_loss

    # ---- pieces -----------------------------------------------------------------------------------
    def _density_alpha(self, n1, pts, dists, dirs):
        """compute_density_alpha with the stage-1 NeRF++ (renderer_zerothick.py:1531-1540); dirs = ray directions."""
        sig, rgb = n1.nerf(pts, dirs)
        alpha = 1.0 - torch.exp(-F.softplus(sig) * dists)
        return alpha, G.linear_to_srgb(torch.exp(torch.clamp(rgb, max=5.0)))

    def _shading(self, nets, scfg, lut, points, normals, view_dirs, feats, s2=False, is_internal=False, aux=None, inter_results=False):
        return shade(nets, scfg, lut, points, normals, view_dirs, feats, s2=s2, is_internal=is_internal, aux=aux,
                     inter_results=inter_results)

    def _inner_occ_loss(self, n2, x, sdf, grads, dirs, aux, step):
        """Hook for the inner segment's occlusion loss: the zero-thickness model has none (renderer_zerothick.py:1890-1915)."""
        return {}

    def _upsample_inner(self, n2, start, dirs, end):
        """Segment-1 hierarchical sampling against the inner SDF (renderer_zerothick.py:1742-1760): 64 uniform fractions,
        two rounds of 32 importance samples on the HIP sampler kernels; returns the 128 sorted fractions (no grad).  As in the
        reference, radii and new SDF samples are taken at start + dir * z with the FRACTION z used as a distance."""
        eng = n2.eng
        lib, S = eng.lib, eng.stream()
        M, dev = start.shape[0], start.device
        cp = ctypes.c_void_p
        zn = torch.linspace(0, 1, 64, device=dev)
        pts = (start[:, None, :] + (end - start)[:, None, :] * zn[None, :, None]).reshape(-1, 3).contiguous()
        sdf = eng.sdf_forward(addr(pts), 3, M * 64, keep=False, want_feat=False)['sdf']
        z = zn[None, :].expand(M, 64).contiguous()
        o, d = start.contiguous(), dirs.contiguous()
        uv = getattr(self, '_uv32', None)         # CPU-computed once, then resident: a pageable host -> device copy per call makes the host
        if uv is None or uv.device != dev:        # wait for the stream to drain (same bits as the per-call copy it replaces)
            uv = self._uv32 = torch.linspace(0.5 / 32, 1.0 - 0.5 / 32, steps=32).to(dev)
        var = self.deviation_network_inner.variance
        sn = 64
        for it in range(2):
            zn_new, Xn = eng.empty(M, 32), eng.empty(M * 32, 3)
            L.check(lib.nu_upsample(cp(addr(o)), cp(addr(d)), cp(addr(z)), cp(addr(sdf)), M, sn, cp(addr(var)),
                                    ctypes.c_float(64.0 * 2 ** it), 1 if self.cfg['clip_sample_variance'] else 0,
                                    cp(addr(uv)), 32, cp(addr(zn_new)), cp(addr(Xn)), S), "nu_upsample")
            last = it == 1
            sdf_n = None if last else eng.sdf_forward(addr(Xn), 3, M * 32, keep=False, want_feat=False)['sdf']
            zo, so = eng.empty(M, sn + 32), (None if last else eng.empty(M, sn + 32))
            L.check(lib.nu_merge_sorted(cp(addr(z)), cp(addr(sdf)), sn, cp(addr(zn_new)), cp(addr(sdf_n)), 32, M,
                                        cp(addr(zo)), cp(addr(so)), S), "nu_merge_sorted")
            z, sdf, sn = zo, so, sn + 32
        return z

    # ---- light paths ------------------------------------------------------------------------------
    def trace_segments(self, rays_o, rays_d):
        """The light paths of a ray batch as a chain of SEGMENTS (renderer_zerothick.py:1571-1828: up to 3 refraction bounces
        against the stage-1 mesh, then the sample placement of every straight piece).

        Segment b holds the rays still alive after b refractions: where they start, where they go, which of them hit the mesh
        (`hit_idx`), which of those refract and continue (`cont_idx`, indices into the segment) and their sample nodes
        x_j = start + v z_j.  The geometry runs on HIP kernels with hand-derived backward -- closest hit (LBVH), differentiable
        hit point / normal, IoR network, Snell refraction / total internal reflection -- and the placement on the no-grad HIP
        samplers; what is left here is index bookkeeping (one `nonzero` per mask, reused)."""
        n1, n2 = self.nets()
        scene, dev = self.scene, rays_o.device
        N0 = rays_o.shape[0]
        segs = []
        start, dirs = rays_o, rays_d
        root = torch.arange(N0, device=dev)                    # camera ray each segment ray descends from
        valid = torch.ones(N0, dtype=torch.bool, device=dev)   # False: total internal reflection somewhere along the path
        for b in range(3):
            inside = b % 2 == 1
            N = start.shape[0]
            S1 = 128 if b == 1 else 256
            inter, hit = dintersect_hip(scene, n1.eng, start, dirs)            # LBVH closest hit + differentiable hit
            hit_idx, d_hit = inter['hit_idx'], inter['d']                      # (one host read for the hit set, shared)
            miss_idx = (~hit).nonzero().flatten() if (b != 1 and hit_idx.numel() < N) else hit_idx[:0]
            point = inter['point']
            normal = F.normalize(inter['n'], dim=-1)
            if inside:
                normal = -normal
            ior = torch.sigmoid(n2.ior(G.embed(point, 6)))                     # IoRNetwork on the HIP GEMMs
            refracts, eta, next_dir, next_start = O.refract(n1.eng, d_hit, normal, ior, point, not inside)
            keep = refracts.nonzero().flatten()
            cont_idx = hit_idx.index_select(0, keep)
            root_hit = root.index_select(0, hit_idx)
            valid[root_hit] = valid.index_select(0, root_hit) & refracts       # (no index list of the lost rays: no host read)
            # ---- sample nodes of this segment: x_j = start + v z_j ----
            v = (start + dirs * 4.5) - start                   # rounded like the reference's `end - start`
            z = torch.linspace(0, 1, S1, device=dev)[None, :].repeat(N, 1)
            if hit_idx.numel() > 0:
                s_hit = start.index_select(0, hit_idx)
                v = v.index_copy(0, hit_idx, point - s_hit)
                if b == 1:                                     # inside the object: hierarchical sampling against the inner SDF
                    with torch.no_grad():
                        z[hit_idx] = self._upsample_inner(n2, s_hit.detach(), d_hit.detach(), point.detach())
            if miss_idx.numel() > 0 and b != 1:                # rays that leave the scene: NeRF++ importance pass, no gradient
                d_miss = dirs.index_select(0, miss_idx)
                with torch.no_grad():
                    z[miss_idx] = O.far_importance_nodes(n1.eng, start.index_select(0, miss_idx), d_miss)
                v = v.index_copy(0, miss_idx, d_miss)
            segs.append(dict(start=start, dirs=dirs, v=v, z=z, cont_idx=cont_idx, n_cont=int(cont_idx.numel()),
                             normal=normal.index_select(0, keep), eta=eta.index_select(0, keep)[:, None], inside=inside))
            if cont_idx.numel() == 0:
                break
            start, dirs = next_start.index_select(0, keep), next_dir.index_select(0, keep)
            root = root.index_select(0, cont_idx)
        return segs, valid[:, None]

    _TWO_STREAM_RAYS = int(os.environ.get('NU_S2_TWO_STREAM_RAYS', 1 << 20))      # 0: inner segment on the caller's stream

    def _inner_segment(self, n2, sg, cos_anneal_ratio, step, s):
        """The samples of the inner segment that lie inside the unit sphere (renderer_zerothick.py:1886-1915): inner SDF + normal,
        NeuS alpha, inner shading.  -> None without such samples, else (ray, sample) indices, alpha, sRGB colour (4 channels) and
        the eikonal / std / occlusion terms."""
        N = sg['start'].shape[0]
        nodes = self.path_points(sg)[:, :-1, :]
        inner = torch.norm(nodes, dim=-1) <= 1.0
        where = inner.nonzero()
        if where.shape[0] == 0:
            return None
        r_i, s_i = where[:, 0], where[:, 1]
        seglen = torch.linalg.norm(nodes[:, 1:] - nodes[:, :-1], dim=-1)
        seglen = torch.cat([seglen, seglen[..., -1:]], -1)
        # (directions through an expanded view, not index_select(0, r_i): its backward would add the many samples of a ray with
        # float atomics; this way the per-sample cotangents land at unique (ray, sample) slots and the sum over a ray's samples is
        # an ordinary reduction -- bitwise reproducible)
        d_in = sg['dirs'][:, None, :].expand(N, nodes.shape[1], 3)[r_i, s_i]
        x_in, len_in = nodes[r_i, s_i], seglen[r_i, s_i]
        y, grads = n2.sdf(x_in)
        a = O.neus_alpha(n2.eng, y[:, 0], grads, d_in, len_in, s, cos_anneal_ratio)
        aux = {}
        c, _ = self._shading(n2, self.color_network_inner.cfg, self.color_network_inner.FG_LUT, x_in, grads, -d_in, y[:, 1:], aux=aux)
        res = dict(r_i=r_i, s_i=s_i, alpha=a, color4=torch.cat([c, torch.zeros_like(c[:, :1])], -1), std=torch.mean(1 / s),
                   gradient_error=(torch.linalg.norm(grads, dim=-1) - 1.0) ** 2)
        res.update(self._inner_occ_loss(n2, x_in, y[:, 0], grads, d_in, aux, step))
        return res

    def _shade_surfaces(self, n1, s1c, segs):
        """(sRGB surface colour, transmitted share) of the surface at the end of every segment that has continuing rays
        (renderer_zerothick.py:1940-1975), all segments in one pass of the network ops.  A surface seen from inside the object
        contributes no colour (`is_internal`: AppShadingNetwork_S2 multiplies it by 0, field.py:1005): its rows are zeroed here."""
        live = [b for b, sg in enumerate(segs) if sg['n_cont'] > 0]
        if not live:
            return {}
        pts, nrm, view = [], [], []
        for b in live:
            sg, cont = segs[b], segs[b]['cont_idx']
            pts.append(sg['start'].index_select(0, cont) + sg['v'].index_select(0, cont) * sg['z'].index_select(0, cont)[:, -1:])
            nrm.append(sg['normal'])
            view.append(-sg['dirs'].index_select(0, cont))
        one = len(live) == 1
        hit_pt, normal, v = (pts[0], nrm[0], view[0]) if one else (torch.cat(pts, 0), torch.cat(nrm, 0), torch.cat(view, 0))
        y, _ = n1.sdf(hit_pt, need_normal=False)
        surf, through = self._shading(n1, s1c.cfg, s1c.FG_LUT, hit_pt, normal, v, y[:, 1:], s2=True, is_internal=False)
        counts = [segs[b]['n_cont'] for b in live]
        surf_b, through_b = (surf,), (through,)
        if not one:
            surf_b, through_b = torch.split(surf, counts, 0), torch.split(through, counts, 0)
        res = {}
        for k, b in enumerate(live):
            res[b] = (surf_b[k] * 0 if segs[b]['inside'] else surf_b[k], through_b[k])
        return res

    @staticmethod
    def path_points(seg):
        return seg['start'][:, None, :] + seg['v'][:, None, :] * seg['z'][..., None]

    # ---- render_core ------------------------------------------------------------------------------
    def render_segments(self, segs, cos_anneal_ratio=0.0, step=None, is_train=True):
        """Front-to-back over the segments in linear RGB with a running transmittance (renderer_zerothick.py:1835-2011, training).
        The outer (|x| > 1) samples of ALL segments go through the NeRF++ in one pass of HIP kernels (stage2_ops.outer_segments);
        each segment's composite is one kernel pair; surface and inner-segment shading are the HIP network ops + one BRDF-mix
        kernel pair (shading_glue.shade)."""
        n1, n2 = self.nets()
        dev = segs[0]['start'].device
        N0 = segs[0]['start'].shape[0]
        T = torch.ones(N0, 3, device=dev)
        colors = []
        out = {'gradient_error': torch.zeros(1, device=dev), 'std': torch.zeros(1, device=dev)}
        s1c = self.stage1_network.color_network
        # The inner segment (inner engine: SDF, NeuS alpha, inner shading) and the outer samples of all segments (stage-1 engine:
        # NeRF++) do not depend on each other, and the inner segment's GEMMs run on small point sets that leave most of the chip
        # idle: the inner segment is enqueued on the inner engine's side stream first and the NeRF++ pass on the current stream
        # beside it (measured: -5 % at 1024 rays, -2..6 % at 4096; `NU_S2_TWO_STREAM_RAYS=0` switches it off); the
        # autograd engine runs each node's backward on its forward stream, so the backward passes overlap the same way.  The two
        # engines have separate reduction arenas and descriptor tables; tensors that cross the fork / join are recorded on the
        # stream that reads them.
        inner_res, side = None, None
        # inv_s of the inner surface (read on the caller's stream: the parameter's gradient accumulates there)
        inv_s = torch.exp(self.deviation_network_inner.variance * 10.0).clip(1e-6, 1e6)
        if self.cfg['freeze_inv_s_step'] is not None and step is not None and step < self.cfg['freeze_inv_s_step']:
            inv_s = inv_s.detach()
        # (the bf16-MFMA modes keep everything on one stream: DESIGN 12, packed-fp32 VALU kernels go wrong beside those GEMMs)
        if (len(segs) > 1 and segs[1]['start'].shape[0] > 0 and N0 <= self._TWO_STREAM_RAYS and dev.type == 'cuda'
                and n2.eng.bf16 == 0):
            main = torch.cuda.current_stream(dev)
            side = n2.eng._fork(mark=False)      # (this engine's ops on the two streams are ordered by op_begin / op_end events)
            for t in [segs[1][k] for k in ('start', 'v', 'z', 'dirs')] + [inv_s]:
                t.record_stream(side)
            with torch.cuda.stream(side):
                inner_res = self._inner_segment(n2, segs[1], cos_anneal_ratio, step, inv_s)
                for t in (inner_res or {}).values():
                    if torch.is_tensor(t):
                        t.record_stream(main)
        outer = O.outer_segments(n1, [(sg['start'], sg['v'], sg['z'], sg['dirs']) for sg in segs])
        # The surfaces the continuing rays cross (stage-1 materials at the hit point, AppShadingNetwork_S2) do not depend on each
        # other or on the transmittance: in training they are shaded in ONE pass over the hit points of all segments -- the rows of
        # a network op are independent, so every value is the one the per-segment calls give, and at the batch sizes of stage 2
        # (10^2-10^3 hit points per segment) the per-segment calls are launch-latency-bound: a third of the launches for the same
        # rows.  Validation keeps the per-segment calls (the first surface also returns its intermediate images).
        surfaces = self._shade_surfaces(n1, s1c, segs) if is_train else None
        for b, sg in enumerate(segs):
            N, cont = sg['start'].shape[0], sg['cont_idx']
            alpha, col = outer[b]
            if b == 1 and N > 0:                               # inside the object: the inner SDF surface (NeuS alpha + shading)
                if side is not None:
                    n2.eng._join()
                else:
                    inner_res = self._inner_segment(n2, sg, cos_anneal_ratio, step, inv_s)
                if inner_res is not None:
                    alpha = alpha.index_put((inner_res['r_i'], inner_res['s_i']), inner_res['alpha'])
                    col = col.index_put((inner_res['r_i'], inner_res['s_i']), inner_res['color4'])
                    out.update({k: inner_res[k] for k in ('std', 'gradient_error', 'loss_occ') if k in inner_res})
            light, T = O.segment_composite(n1.eng, alpha, col, T)
            if sg['n_cont'] == 0:
                colors.append(light)
                break
            # the surface the continuing rays cross: stage-1 materials at the hit point, AppShadingNetwork_S2
            if surfaces is not None:
                surf, through = surfaces[b]
                T_c = T.index_select(0, cont)
                colors.append(light.index_add(0, cont, G.srgb_to_linear(surf) * T_c))
                T = T_c * through
                continue
            hit_pt = sg['start'].index_select(0, cont) + sg['v'].index_select(0, cont) * sg['z'].index_select(0, cont)[:, -1:]
            y, _ = n1.sdf(hit_pt, need_normal=False)
            if b == 0 and not is_train:
                # validation images of the first surface (renderer_zerothick.py:1952-1962 / renderer.py:2274-2289): shading normal
                # mapped to [0,1], specular terms of AppShadingNetwork_S2's intermediate results, scattered to the camera rays
                surf, through, inter = self._shading(n1, s1c.cfg, s1c.FG_LUT, hit_pt, sg['normal'], -sg['dirs'].index_select(0, cont),
                                                     y[:, 1:], s2=True, is_internal=sg['inside'], inter_results=True)
                zero = torch.zeros(N0, 3, device=dev)
                out['normal'] = zero.index_copy(0, cont, (F.normalize(sg['normal'], dim=-1) + 1.0) * 0.5)
                for k in ('specular_color', 'specular_light', 'specular_ref'):
                    out[k] = zero.index_copy(0, cont, inter[k].expand(-1, 3))
            else:
                surf, through = self._shading(n1, s1c.cfg, s1c.FG_LUT, hit_pt, sg['normal'], -sg['dirs'].index_select(0, cont),
                                              y[:, 1:], s2=True, is_internal=sg['inside'])
            T_c = T.index_select(0, cont)
            colors.append(light.index_add(0, cont, G.srgb_to_linear(surf) * T_c))
            T = T_c * through
        for b in range(len(colors) - 1, 0, -1):                # what the deeper segments saw flows back along the paths
            colors[b - 1] = colors[b - 1].index_add(0, segs[b - 1]['cont_idx'], colors[b])
        out['ray_rgb'] = torch.clamp(G.linear_to_srgb(colors[0]), min=0.0, max=1.0)
        out['acc'] = torch.ones(N0, device=dev)
        if not is_train:
            for k in ('normal', 'specular_color', 'specular_light', 'specular_ref'):
                out.setdefault(k, torch.zeros(N0, 3, device=dev))
        return out

    # the reference's method names (renderer_zerothick.py:1571, :1835), kept for callers that use them
    def ray_trace(self, rays_o, rays_d):
        return self.trace_segments(rays_o, rays_d)

    def render_core(self, segs, cos_anneal_ratio=0.0, step=None, is_train=True, **_):
        return self.render_segments(segs, cos_anneal_ratio=cos_anneal_ratio, step=step, is_train=is_train)

    def render(self, rays_o, rays_d, near=None, far=None, human_poses=None, perturb_overwrite=-1, cos_anneal_ratio=0.0,
               is_train=True, step=None, is_nerf=False):
        """renderer_zerothick.py:1442-1466."""
        n1, n2 = self.nets()
        n1.eng.pack()
        n2.eng.pack()
        n1.begin_pass()
        n2.begin_pass()
        segs, valid = self.trace_segments(rays_o, rays_d)
        ret = self.render_segments(segs, cos_anneal_ratio=cos_anneal_ratio, step=step, is_train=is_train)
        ret['tir_mask'] = valid
        with torch.no_grad():
            ret['_paths'] = [self.path_points(sg) for sg in segs]       # materialised for inspection only
        ret['_ior_ratios'] = [sg['eta'] for sg in segs if sg['n_cont'] > 0]
        ret['_directions'] = [sg['dirs'] for sg in segs]
        ret['_normals'] = [sg['normal'] for sg in segs if sg['n_cont'] > 0]
        return ret

    def train_step_rays(self, batch, step):
        """renderer_zerothick.py:1259-1275 on an explicit ray batch."""
        rays_d = F.normalize(batch['rays_d'], dim=-1)
        out = self.render(batch['rays_o'], rays_d, None, None, None, -1, self.get_anneal_val(step), is_train=True, step=step,
                          is_nerf=self.is_nerf)
        tm = out['tir_mask'].detach().float()
        out['loss_rgb'] = self.compute_rgb_loss(out['ray_rgb'] * tm, batch['rgbs'] * tm)
        return out

    def train_step(self, step):
        """renderer_zerothick.py:1259-1275 on the module's device-resident ray store."""
        rn = self.cfg['train_ray_num']
        dev = self.IORs.device
        if getattr(self, 'train_batch', None) is None:
            raise RuntimeError("train_step needs the module's ray store: construct with training=True (and, for an image database, "
                               "hand the loaded images over with set_ray_store)")
        if self._batch_dev != dev:
            self.train_batch = {k: v.to(dev) for k, v in self.train_batch.items()}
            if getattr(self, 'train_poses', None) is not None:
                self.train_poses = self.train_poses.to(dev)
            self._batch_dev = dev
            self._shuffle_train_batch()
        batch = {k: v[self.train_batch_i:self.train_batch_i + rn] for k, v in self.train_batch.items()}
        self.train_batch_i += rn
        if self.train_batch_i + rn >= self.tbn:
            self._shuffle_train_batch()
        if 'dirs' in batch:         # real captures: world-space rays from the camera poses (renderer_zerothick.py:347-361)
            poses, idxs = self.train_poses, batch['idxs'][..., 0]
            rays_o = (poses[:, :, :3].permute(0, 2, 1) @ -poses[:, :, 3:])[idxs, :, 0]
            rays_d = (poses[idxs, :, :3].permute(0, 2, 1) @ batch['dirs'].unsqueeze(-1))[..., 0]
            batch = {'rays_o': rays_o, 'rays_d': rays_d, 'rgbs': batch['rgbs']}
        return self.train_step_rays(batch, step)

    _EVAL_KEYS = ('ray_rgb', 'gradient_error', 'normal', 'tir_mask', 'specular_light', 'specular_color', 'specular_ref')

    def render_eval(self, batch, step, chunk=None):
        """test_step's ray loop (renderer_zerothick.py:1226-1241) over an explicit ray batch: chunks of cfg['test_ray_num'] rays,
        cos_anneal 0, is_train=False; ray_rgb / gt masked by the TIR mask as the reference does."""
        trn = int(chunk or self.cfg['test_ray_num'])
        outs = {k: [] for k in self._EVAL_KEYS}
        n = batch['rays_o'].shape[0]
        for ri in range(0, n, trn):
            rays_d = F.normalize(batch['rays_d'][ri:ri + trn], dim=-1)
            o = self.render(batch['rays_o'][ri:ri + trn], rays_d, None, None, None, 0, 0, is_train=False, step=step, is_nerf=self.is_nerf)
            for k in self._EVAL_KEYS:
                outs[k].append(o[k].detach())
        outs = {k: torch.cat(v, 0) for k, v in outs.items()}
        tm = outs['tir_mask'].float()
        if 'rgbs' in batch:
            outs['loss_rgb'] = self.compute_rgb_loss(outs['ray_rgb'] * tm, batch['rgbs'] * tm)
        return outs

    def test_step(self, index, step):
        """Full-image validation render of camera `index` (renderer_zerothick.py:1209-1257): the test image of the image store
        handed to set_ray_store (down-sampled as cfg says, its 'depths' / 'masks' as gt_depth / gt_mask), or every pixel of the
        down-sampled synthetic camera (gt_depth / gt_mask: the empty scene's zeros)."""
        from . import synthetic
        dev = self.IORs.device
        if getattr(self, 'test_imgs_info', None) is not None:
            batch, h, w, depth, mask = self._test_batch_from_store(index, dev)
        else:
            hw = int(self.cfg.get('synthetic_hw', 800))
            ratio = float(self.cfg['downsample_ratio']) if self.cfg['test_downsample_ratio'] else 1.0
            rays, h, w = synthetic.make_image_rays(index, hw=hw, seed=int(self.cfg.get('ray_seed', 6033)), downsample=ratio)
            batch = {k: torch.from_numpy(v).to(dev) for k, v in rays.items()}
            depth, mask = torch.zeros(h, w, 1), torch.zeros(h, w, 1, dtype=torch.int32)
        with torch.no_grad():
            outputs = self.render_eval(batch, step)
        tm = outputs['tir_mask'].float()
        outputs['gt_rgb'] = (batch['rgbs'] * tm).reshape(h, w, 3)
        outputs['ray_rgb'] = (outputs['ray_rgb'] * tm).reshape(h, w, 3)
        outputs['gt_depth'] = depth
        outputs['gt_mask'] = mask
        self.zero_grad()
        return outputs

    def forward(self, data):
        """trainer protocol (renderer_zerothick.py:2013-2035): {'step'} -> train_step; {'index','eval','step'} (the
        ValidationEvaluator's call, train/train_valid.py:25-29) -> test_step."""
        step = data['step']
        if 'eval' in data:
            index = data['index']
            index = int(index.reshape(-1)[0]) if torch.is_tensor(index) else int(np.asarray(index).reshape(-1)[0])
            return self.test_step(index, step)
        return self.train_step(step)


from .renderer import name2renderer  # noqa: E402

name2renderer['stage2'] = Stage2Renderer          # registry of network/renderer_zerothick.py:2057-2060
