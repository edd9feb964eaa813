"""Golden vectors for the NON-zero-thickness stage-2 model (network/renderer.py: Stage2Renderer), from the
reference's own class under the shims of oracle/gen_golden.py (build container only).

The OptiX / PyMesh-backed `Scene` is replaced by a brute-force scene with the same `Dintersect` contract; PyMesh's
"vertex_gaussian_curvature" (absent here) by the angle-defect estimate of nu_nerf_amd/lbvh.py -- that input is therefore NOT pinned
by the reference, everything computed from it is.  Parameters: the stage-1 part from nu_nerf_amd/params.py, every other state-dict
entry from `nu_nerf_amd.params.params_from_manifest` (names + shapes + seed -> values), so the fixture carries the manifest and
the product can rebuild the same weights.  Output: tests/golden/stage2_thick_step6000_r24.npz and ..._step25000_r24.npz (the latter
past `occ_loss_step`, with the inner occlusion loss in the total): inputs, per-ray outputs, TIR mask, segment geometry, loss terms,
gradient norms of every trained parameter.  The product is nu_nerf_amd/stage2_thick.py (tests/test_stage2_thick_gpu.py)."""
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.gen_golden import install_shims, to_t, OUT   # noqa: E402


def main(step=6000, loss_names=('eikonal', 'std', 'nerf_render'), open_mesh=False):
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import yaml
    import network.renderer as rr
    import network.DiffRender as DR
    from oracle.lbvh_oracle import brute_force_closest_hit
    from nu_nerf_amd.params import init_stage1_params, randomize_for_parity
    from nu_nerf_amd.lbvh import icosphere, corner_angles_and_face_normals

    V, F = icosphere(3, 0.5)
    if open_mesh:
        # an OPEN surface (the cap z > 0.2 removed, unused vertices kept): rays that enter through the shell can find no exit,
        # which is the ragged branch of ray_trace (renderer.py:1660-1670); rays through the opening meet the back of the far side
        F = np.ascontiguousarray(F[V[F].mean(1)[:, 2] <= 0.2])

    class FakeScene:
        def __init__(self, mesh_path):
            self.vertices = torch.from_numpy(V)
            self.faces = torch.from_numpy(F.astype(np.int64))
            tri = self.vertices[self.faces]
            ang, fn = corner_angles_and_face_normals(tri)
            vn = torch.zeros_like(self.vertices)
            vn.index_add_(0, self.faces.reshape(-1), (ang[:, :, None] * fn[:, None, :]).reshape(-1, 3))
            self.normals = vn / vn.norm(dim=1, keepdim=True)
            area = 0.5 * torch.linalg.norm(torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=1), dim=1)
            ang_sum = torch.zeros(len(self.vertices)).index_add_(0, self.faces.reshape(-1), ang.reshape(-1))
            v_area = torch.zeros(len(self.vertices)).index_add_(0, self.faces.reshape(-1), (area / 3.0)[:, None].expand(-1, 3).reshape(-1))
            self.gaussian_curvatures = torch.clamp((2.0 * np.pi - ang_sum) / v_area, -10.0, 10.0)[:, None]

        def Dintersect(self, ray):
            rays = torch.cat([ray.origin, ray.direction], 1).detach().numpy().astype(np.float32)
            hit, idx, _ = brute_force_closest_hit(V, F, rays)
            hitted = torch.from_numpy(hit > 0)
            faces_ind = torch.from_numpy(idx.astype(np.int64))
            f = self.faces[faces_ind[hitted]]
            rh = ray.select(hitted)
            u, v, t, n, gk = DR.JIT_Dintersect(rh.origin, rh.direction, self.vertices[f].float(), self.normals[f].float(),
                                               self.gaussian_curvatures[f].float())
            return DR.Intersection(u=u, v=v, t=t, n=n, g_k=gk, ray=rh, faces_ind=faces_ind[hitted]), hitted

    rr.Scene = FakeScene
    import types
    rr.trimesh.load = lambda path, **k: types.SimpleNamespace(vertices=V.astype(np.float64), faces=F.astype(np.int64))
    tmp = tempfile.mkdtemp()
    s1 = randomize_for_parity(init_stage1_params(6033, sphere_direction=True), seed=1)
    torch.save({'network_state_dict': to_t(s1)}, os.path.join(tmp, 's1.pth'))
    s1cfg = {'name': 's1', 'network': 'shape', 'get_mask': False, 'database_name': 'real/x/raw_1024', 'is_nerf': False, 'apply_occ_loss': True,
             'occ_loss_step': 15000, 'freeze_inv_s_step': 15000, 'zero_thickness': False,
             'shader_config': {'sphere_direction': True, 'human_light': False, 'light_exp_max': 5.0}}
    with open(os.path.join(tmp, 's1.yaml'), 'w') as fh:
        yaml.safe_dump(s1cfg, fh)
    cfg = {'name': 'golden_s2t', 'network': 'stage2', 'get_mask': False, 'database_name': 'real/x/raw_1024', 'is_nerf': False,
           'shader_config': {'sphere_direction': True, 'human_light': False, 'light_exp_max': 5.0},
           'loss': list(loss_names), 'eikonal_weight': 0.02, 'freeze_inv_s_step': 5000, 'occ_loss_step': 20000,
           'stage1_ckpt_dir': os.path.join(tmp, 's1.pth'), 'stage1_cfg_dir': os.path.join(tmp, 's1.yaml'),
           'stage1_mesh_dir': 'unused.ply'}
    net = rr.Stage2Renderer(cfg, training=False)
    keys = list(net.state_dict().keys())
    print("constructed; state_dict entries:", len(keys))
    tops = {}
    for k in keys:
        tops[k.split('.')[0]] = tops.get(k.split('.')[0], 0) + 1
    print(tops)
    from nu_nerf_amd.synthetic import make_object_rays
    from nu_nerf_amd.params import params_from_manifest
    from network.loss import name2loss
    # ---- parameters: stage 1 from its own generator (both aliases), everything else from the manifest rule ----
    sd = net.state_dict()
    own = [k for k in keys if not k.startswith(('stage1_network.', 'color_network.stage1_network.', 'infinity_far_bkgr.'))
           and not k.endswith('FG_LUT')]                         # the LUT is an asset both sides load from the same file
    manifest = [(k, tuple(sd[k].shape)) for k in own]
    p2 = params_from_manifest(manifest, seed=7044)
    # the inner SDF decides where the 64 importance samples of the inner segment fall (inverse CDF of NeuS weights): a network of
    # random weights makes that placement ill-conditioned (1e-7 in an SDF value moves a sample by 1e-2), so the inner SDF and its
    # variance take the perturbed geometric init of the zero-thickness fixture (a radius-0.5 sphere-like surface) instead
    inner = randomize_for_parity(init_stage1_params(7044, sphere_direction=True), seed=3)
    for k, v in inner.items():
        if k.startswith('sdf_network.'):
            p2['sdf_network_inner.' + k[len('sdf_network.'):]] = v
    p2['deviation_network_inner.variance'] = inner['deviation_network.variance']
    for k, v in s1.items():
        p2['stage1_network.' + k] = v
        p2['color_network.stage1_network.' + k] = v
        if k.startswith('infinity_far_bkgr.'):
            p2[k] = v
    for k in keys:
        if k.endswith('FG_LUT') and k not in p2:
            p2[k] = sd[k].numpy()
    missing = [k for k in keys if k not in p2]
    print("own entries", len(own), "missing", missing[:5])
    print("load:", net.load_state_dict(to_t({k: p2[k] for k in keys}), strict=True))
    losses = [name2loss[n](cfg) for n in cfg['loss']]
    R = 24
    rays = make_object_rays(R, seed=500)
    o, d, rgbs = (torch.from_numpy(rays[k]) for k in ('rays_o', 'rays_d', 'rgbs'))
    dn = torch.nn.functional.normalize(d, dim=-1)
    mask = torch.ones(R, 1)
    net.zero_grad()
    pathes, converges, directions, ior_ratios, infinity_bkgr, gradient_mesh, tir_mask = net.ray_trace(o, dn, mask)
    print("segments", len(pathes), "converged per bounce", [int(c.sum()) for c in converges], "rays per segment",
          [int(p.shape[0]) for p in pathes], "samples", [int(p.shape[1]) for p in pathes], "tir", int(tir_mask.sum()))
    net.zero_grad()
    outputs = net.render(o, dn, mask, None, None, None, -1, net.get_anneal_val(step), is_train=True, step=step, is_nerf=False)
    outputs['loss_rgb'] = net.compute_rgb_loss(outputs['ray_rgb'] * outputs['tir_mask'].detach() * mask,
                                               rgbs * outputs['tir_mask'].detach() * mask)
    log = {}
    for ls in losses:
        log.update(ls(outputs, {}, step))
    total = 0
    for k, v in log.items():
        if k.startswith('loss'):
            total = total + torch.mean(v)
    total.backward()
    res = {'rays_o': rays['rays_o'], 'rays_d': rays['rays_d'], 'rgbs': rays['rgbs'], 'step': np.asarray(step),
           'manifest_names': np.asarray([k for k, _ in manifest]),
           'manifest_shapes': np.asarray([','.join(str(x) for x in shp) for _, shp in manifest]),
           'manifest_seed': np.asarray(7044), 'state_dict_keys': np.asarray(keys),
           'vertex_gaussian_curvature': net.scene.gaussian_curvatures.numpy(),
           'total_loss': total.detach().numpy(), 'out_ray_rgb': outputs['ray_rgb'].detach().numpy(),
           'out_tir_mask': outputs['tir_mask'].numpy(), 'out_gradient_error': outputs['gradient_error'].detach().numpy(),
           'out_std': outputs['std'].detach().numpy(), 'out_normal': outputs['normal'].detach().numpy(),
           'out_specular_color': outputs['specular_color'].detach().numpy()}
    for i in range(len(pathes)):
        res['path%d' % i] = pathes[i].detach().numpy()
        res['conv%d' % i] = converges[i].numpy()
        res['dir%d' % i] = directions[i].detach().numpy()
    for i, r in enumerate(ior_ratios):
        res['ior%d' % i] = r.detach().numpy()
    for i, gm in enumerate(gradient_mesh):
        res['normal_mesh%d' % i] = gm.detach().numpy()
    for k, v in log.items():
        if k.startswith('loss'):
            res['term_' + k] = torch.mean(v).detach().numpy()
    gn = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    res['grad_names'] = np.asarray(sorted(gn.keys()))
    res['grad_norms'] = np.asarray([float(gn[k].double().norm()) for k in sorted(gn.keys())])
    os.makedirs(OUT, exist_ok=True)
    res['out_loss_occ'] = outputs['loss_occ'].detach().numpy()
    res['mesh_faces'] = F.astype(np.int32)
    # the validation render of the same rays (test_step's call, renderer.py:1297-1300: perturb 0, cos_anneal 0, is_train=False)
    with torch.no_grad():
        ev = net.render(o, dn, mask, None, None, None, 0, 0, is_train=False, step=step, is_nerf=False)
    for k in ('ray_rgb', 'normal', 'specular_color', 'specular_light', 'specular_ref', 'tir_mask'):
        res['eval_' + k] = ev[k].detach().numpy()
    np.savez_compressed(os.path.join(OUT, "stage2_thick_step%d_r24%s.npz" % (step, "_open" if open_mesh else "")), **res)
    print("loss", float(total), {k: float(v) for k, v in res.items() if k.startswith('term_')}, "n grads", len(gn),
          "rgb range", float(outputs['ray_rgb'].min()), float(outputs['ray_rgb'].max()))
    print("params without grad:", sorted(set(n.split('.')[0] for n, p in net.named_parameters() if p.grad is None)))


if __name__ == "__main__":
    main()
    # past occ_loss_step and freeze_inv_s_step: the inner occlusion probe (renderer.py:2247-2255) in the loss, trainable inner variance
    main(step=25000, loss_names=('eikonal', 'std', 'nerf_render', 'occ'))
    main(step=6000, open_mesh=True)
