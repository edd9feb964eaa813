"""Stage-2 renderer, NON-zero-thickness variant: drop-in for `Stage2Renderer` of network/renderer.py:907-2398
(`name2renderer['stage2']` of the `zero_thickness: False` configs, registry :2400-2403, dispatch run_training.py:16-20).

What differs from the zero-thickness model (nu_nerf_amd/stage2.py), and where it runs:
  * the surface is a SHELL: at every hit the ray refracts twice, through a shell of learned thickness whose faces are modelled as
    concentric spheres of the local curvature radius (renderer.py:1692-2032) -- one HIP kernel pair per bounce
    (nu_s2_shell_*: forward, and a backward that carries the 12 x 12 Jacobian per ray as forward-mode dual numbers);
    inputs: the hit op's point / normal / interpolated Gaussian curvature (nu_s2_hit_*, per-vertex angle defect over vertex
    area, lbvh.py) and the raw outputs of IORs_pred and thickness_pred on the HIP GEMMs (nets.IorFn); IoRint_pred is
    multiplied by 0 in the reference (:1734) and is not evaluated (its parameters get no gradient instead of zeros);
  * sample layout 64 / 128 / 64 nodes per segment (:2057-2124): hits -> uniform fractions of the segment (the inner segment
    through the stage-1 sampler kernels against the inner SDF), misses -> 64 fixed inverse-depth nodes, no importance pass;
  * the inner surface is shaded by AppShadingNetwork_SpecInner (field.py:1320-1571: 8 position frequencies, 2 refraction
    frequencies, refraction light capped at exp(-0.2)) -- the same network ops and BRDF-mix kernel as every other shading;
  * rays that are inside the object and find no exit are dropped from the paths after the fact (:1660-1670);
  * every surface after the first is shaded black (`is_internal = i != 0`, :2272).
Everything else -- outer samples of all segments in one NeRF++ pass, NeuS alpha, per-segment composite -- is stage2.py's.
state_dict() names / order equal the reference's (565 entries incl. the `color_network.stage1_network.*` and
`infinity_far_bkgr.*` aliases); pinned by tests/golden/stage2_thick_step6000_r24.npz (the reference's own class)."""
import torch
import torch.nn as nn

from . import stage2_ops as O
from . import torch_glue as G
from .engine import Stage1Engine
from .lbvh import Scene, dintersect_hip
from .nets import Stage1Nets
from .renderer import AppShadingNetwork, NeRFNetwork, SDFNetwork, SingleVarianceNetwork
from .renderer_std import NeROShapeRenderer
from .stage2 import AppShadingNetworkS2, IoRNetwork
from .stage2 import Stage2Renderer as _ZeroThickStage2


class AppShadingNetworkSpecInner(AppShadingNetwork):
    """Parameter layout and defaults of AppShadingNetwork_SpecInner (field.py:1320-1379): the stage-1 predictors with
    `light_pos_freq` 8 and `refrac_freq` 2; its refraction light is exp(min(., -0.2)) (field.py:1373)."""
    default_cfg = {**AppShadingNetwork.default_cfg, 'light_pos_freq': 8, 'light_exp_max': 5.0, 'refrac_freq': 2}

    def __init__(self, cfg):
        super().__init__(cfg)
        self.cfg['refrac_exp_max'] = -0.2


class Stage2Renderer(_ZeroThickStage2):
    default_cfg = {**NeROShapeRenderer.default_cfg, 'train_ray_num': 1024, 'is_nerf': False}

    def __init__(self, cfg, training=True):
        nn.Module.__init__(self)
        self.cfg = {**self.default_cfg, **cfg}
        self.is_nerf = self.cfg['is_nerf']
        self.get_mask = self.cfg['get_mask']                           # renderer.py:962 (default False here; the reference has no default)
        self.IORs = nn.Parameter(torch.zeros(10))
        self.nerf_network = NeRFNetwork()
        self.stage1_network = NeROShapeRenderer(self._load_stage1_cfg(), training=False)
        self._load_stage1_ckpt()
        self.infinity_far_bkgr = self.stage1_network.infinity_far_bkgr  # renderer.py:986: the alias is a state_dict() prefix
        self.IORs_pred = IoRNetwork()
        self.IoRint_pred = IoRNetwork()
        self.thickness_pred = IoRNetwork()                               # ThicknessNetwork has IoRNetwork's layout (field.py:1068-1087)
        self.color_network = AppShadingNetworkS2(self.cfg['shader_config'], self.stage1_network)
        self.sdf_network_inner = SDFNetwork()
        self.deviation_network_inner = SingleVarianceNetwork(self.cfg['inv_s_init'])
        self.color_network_inner = AppShadingNetworkSpecInner(self.cfg['shader_config'])
        self._init_own_parameters()
        self._mesh = self._load_mesh()
        self.scene = None
        self._nets = None
        if training:
            self._init_dataset()

    def _init_own_parameters(self):
        from .params import init_stage2_thick_own_params
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        self.load_param_dict(init_stage2_thick_own_params(seed, self.color_network_inner.cfg))

    def nets(self):
        """(stage-1 nets, inner nets) over two HIP engines + the LBVH scene; the IoR and thickness networks ride on the inner
        engine's GEMMs."""
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
            for k, p in self.IORs_pred.module0.named_parameters():
                named['ior_network.' + k] = p
            for k, p in self.thickness_pred.module0.named_parameters():
                named['thickness_network.' + k] = p
            ecfg = dict(self.cfg)
            ecfg.update(self.color_network_inner.cfg)
            n2 = Stage1Nets(Stage1Engine(named, dev, ecfg), named)
            V, Fc = self._mesh
            self.scene = Scene(torch.from_numpy(V).to(dev), torch.from_numpy(Fc).to(dev))
            self._nets = (n1, n2)
        return self._nets

    # ---- light paths ------------------------------------------------------------------------------
    def trace_segments(self, rays_o, rays_d):
        """renderer.py:1610-2148 as a chain of segments (same record as stage2.trace_segments): up to 3 shell crossings against
        the stage-1 mesh, then the sample nodes of every straight piece."""
        n1, n2 = self.nets()
        scene, dev = self.scene, rays_o.device
        N0 = rays_o.shape[0]
        segs = []
        start, dirs = rays_o, rays_d
        root = torch.arange(N0, device=dev)                    # camera ray each segment ray descends from
        valid = torch.ones(N0, dtype=torch.bool, device=dev)   # False: total internal reflection somewhere along the path
        z_far = 1.0 / torch.flip(torch.linspace(1e-3, 1.0 - 1.0 / (64 + 1.0), 64, device=dev), dims=[-1]) + 1.0 / 64   # :2102-2114
        for b in range(3):
            inside = b % 2 == 1
            inter, hit = dintersect_hip(scene, n1.eng, start, dirs, curvature=True)
            hit_idx, d_hit = inter['hit_idx'], inter['d']
            all_hit = hit_idx.numel() == start.shape[0]
            if b == 1 and not all_hit:
                # inside the object without an exit (:1660-1670): the ray leaves the paths; its first surface is not shaded either
                stay = hit_idx
                prev = segs[0]
                prev['cont_idx'] = prev['cont_idx'].index_select(0, stay)
                prev['normal'] = prev['normal'].index_select(0, stay)      # (`ior_ratios[0]` keeps its rows in the reference;
                #                                                              the shader ignores that argument, field.py:909)
                prev['n_cont'] = int(stay.numel())
                start, dirs, root = start.index_select(0, stay), dirs.index_select(0, stay), root.index_select(0, stay)
                hit = torch.ones(stay.numel(), dtype=torch.bool, device=dev)
                hit_idx, all_hit = torch.arange(stay.numel(), device=dev), True
                if stay.numel() == 0:
                    break
            N = start.shape[0]
            point = inter['point']
            pe = G.embed(point, 6)
            ior_raw, thick_raw = n2.ior_and_thickness(pe)     # the two networks as grouped launches
            refracts, tir_ok, eta, normal, p_end, next_start, next_dir = O.shell_refract(
                n1.eng, d_hit, inter['n'], point, ior_raw, inter['g_k'], thick_raw, inside)
            keep = refracts.nonzero().flatten()
            cont_idx = hit_idx.index_select(0, keep)
            root_hit = root.index_select(0, hit_idx)
            valid[root_hit] = valid.index_select(0, root_hit) & tir_ok      # (no index list of the lost rays: no host read)
            # ---- sample nodes x_j = start + v z_j: 64 (128 inside the object) ----
            S1 = 128 if b == 1 else 64
            if b == 1:                                         # every ray of the inner segment hit (see above)
                with torch.no_grad():                          # hierarchical sampling against the inner SDF (:2081-2099)
                    z = self._upsample_inner(n2, start.detach(), dirs.detach(), p_end.detach())
                v = p_end - start
            else:
                # hits: uniform fractions of (end - start); rays that leave the scene: fixed inverse-depth nodes along the
                # direction (:2101-2119) -- chosen per row with `where`, no index list of the misses
                z = torch.where(hit[:, None], torch.linspace(0, 1, S1, device=dev)[None, :], z_far[None, :])
                v = dirs if hit_idx.numel() == 0 else dirs.index_copy(0, hit_idx, p_end - start.index_select(0, hit_idx))
            segs.append(dict(start=start, dirs=dirs, v=v, z=z, cont_idx=cont_idx, n_cont=int(cont_idx.numel()),
                             normal=normal.index_select(0, keep), eta=eta.index_select(0, keep)[:, None], inside=b != 0))
            if cont_idx.numel() == 0:
                break
            start, dirs = next_start.index_select(0, keep), next_dir.index_select(0, keep)
            root = root.index_select(0, cont_idx)
        return segs, valid[:, None]

    def render_segments(self, segs, cos_anneal_ratio=0.0, step=None, is_train=True):
        out = super().render_segments(segs, cos_anneal_ratio=cos_anneal_ratio, step=step, is_train=is_train)
        out.setdefault('loss_occ', torch.zeros(1, device=segs[0]['start'].device))
        return out

    def _inner_occ_loss(self, n2, x, sdf, grads, dirs, aux, step, perm=None):
        """renderer.py:2247-2255 / :1580-1608: L1 between the inner shader's occlusion probability and the hit probability of the
        reflected ray marched through the INNER SDF (no gradient; the stage-1 probe kernels on the inner engine), at the inner
        samples close to the inner surface that face the ray; at most `occ_loss_max_pn` of them (random subset, as the reference)."""
        cfg = self.cfg
        dev = x.device
        if not cfg['apply_occ_loss'] or step < cfg['occ_loss_step'] or 'occ_raw' not in aux:
            return {'loss_occ': torch.zeros(1, device=dev)}
        with torch.no_grad():
            mask = (torch.norm(x, dim=-1) < 0.999) & (torch.sum(grads * dirs, -1) < 0) & (torch.abs(sdf) < cfg['occ_sdf_thresh'])
            idx = torch.nonzero(mask)[:, 0]
            if idx.numel() > cfg['occ_loss_max_pn']:
                if perm is None:
                    perm = torch.randperm(idx.numel(), device=dev)
                idx = torch.sort(idx[perm[:cfg['occ_loss_max_pn']]])[0]
            if idx.numel() == 0:
                return {'loss_occ': torch.zeros(1, device=dev)}
            occ_gt = n2.eng.occ_probe(x.detach()[idx], aux['reflective'].detach()[idx])
        occ_prob = aux['occ_raw'][idx] * 0.5 + 0.5
        return {'loss_occ': torch.nn.functional.l1_loss(occ_prob.reshape(-1, 1), occ_gt.reshape(-1, 1))}

    def train_step_rays(self, batch, step):
        """renderer.py:1315-1330 on an explicit ray batch (`mask` = batch['masks'] or ones)."""
        import torch.nn.functional as F
        rays_d = F.normalize(batch['rays_d'], dim=-1)
        out = self.render(batch['rays_o'], rays_d, None, None, None, -1, self.get_anneal_val(step), is_train=True, step=step,
                          is_nerf=self.is_nerf)
        tm = out['tir_mask'].detach().float()
        if 'masks' in batch:
            tm = tm * batch['masks'].reshape(-1, 1)
        out['loss_rgb'] = self.compute_rgb_loss(out['ray_rgb'] * tm, batch['rgbs'] * tm)
        return out


from .renderer_std import name2renderer  # noqa: E402

name2renderer['stage2'] = Stage2Renderer          # registry of network/renderer.py:2400-2403
