"""Golden vectors for the NON-zero-thickness stage-2 model (network/renderer.py: Stage2Renderer), from the
reference's own class under the shims of oracle/gen_golden.py (build container only).  Work in progress: this script first
probes that the class constructs and steps under the shims."""
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.gen_golden import install_shims, to_t, OUT   # noqa: E402


def main():
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
           'loss': ['eikonal', 'std', 'nerf_render'], 'eikonal_weight': 0.02, 'freeze_inv_s_step': 5000,
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
    R, step = 24, 6000
    rays = make_object_rays(R, seed=500)
    o, d, rgbs = (torch.from_numpy(rays[k]) for k in ('rays_o', 'rays_d', 'rgbs'))
    dn = torch.nn.functional.normalize(d, dim=-1)
    mask = torch.ones(R, 1)
    res = net.ray_trace(o, dn, mask)
    pathes, converges = res[0], res[1]
    print("segments", len(pathes), "converged per bounce", [int(c.sum()) for c in converges], "rays per segment",
          [int(p.shape[0]) for p in pathes], "samples", [int(p.shape[1]) for p in pathes])
    out = net.render(o, dn, mask, None, None, None, -1, net.get_anneal_val(step), is_train=True, step=step, is_nerf=False)
    print({k: (tuple(v.shape) if hasattr(v, 'shape') else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
