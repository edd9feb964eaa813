"""Stage-1 parameter inventory and a portable, seed-reproducible initialiser.

Names and shapes follow the reference's `state_dict()` exactly so checkpoints interchange
(legacy weight-norm names `weight_g` / `weight_v`):
  sdf_network.lin{0..8}.{bias,weight_g,weight_v}        network/field.py:94-124
  deviation_network.variance                            network/field.py:191-195
  outer_nerf.{pts_linears.N,views_linears.0,feature_linear,alpha_linear,rgb_linear}.{weight,bias}
                                                        network/field.py:246-261
  color_network.<predictor>.{0,2,4,6}.{bias,weight_g,weight_v}, color_network.FG_LUT
                                                        network/field.py:371-408, :569-611
  infinity_far_bkgr.module0.{0,2,4,6,8}.*               network/field.py:1020-1036 (never used in forward)

The initial DISTRIBUTIONS restate the reference initialisers (geometric SDF init field.py:102-120,
nn.Linear default init, the bias constants of field.py:598-611 and renderer_zerothick.py:156), but the
draws come from numpy's PCG64 so that the same seed gives the same weights on any machine -- tests,
bench and the golden-vector generator all rebuild identical weights without shipping them.
"""
import math
import os
from collections import OrderedDict

import numpy as np

_ASSET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "bsdf_256_256.bin")
FG_LUT_SHA256 = "see tests/test_params.py"


def load_fg_lut():
    """Split-sum BRDF lookup table, fp32 [1,256,256,2] (reference asset assets/bsdf_256_256.bin, field.py:583)."""
    return np.fromfile(_ASSET, dtype=np.float32).reshape(1, 256, 256, 2).copy()


def _linear_default(rng, out_dim, in_dim):
    """nn.Linear default: weight, bias ~ U(-1/sqrt(in), 1/sqrt(in))."""
    b = 1.0 / math.sqrt(in_dim)
    w = rng.uniform(-b, b, size=(out_dim, in_dim)).astype(np.float32)
    bias = rng.uniform(-b, b, size=(out_dim,)).astype(np.float32)
    return w, bias


def _put_wn(p, prefix, w, bias):
    p[prefix + ".bias"] = bias.astype(np.float32)
    p[prefix + ".weight_g"] = np.linalg.norm(w.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    p[prefix + ".weight_v"] = w.astype(np.float32)


def predictor_dims(sphere_direction=False, refrac_freq=6, light_pos_freq=6):
    """(name, in_dim, out_dim, last-bias constant or None) of every make_predictor stack."""
    pos = 3 + 6 * light_pos_freq
    return [
        ("metallic_predictor", 259, 1, None),
        ("roughness_predictor", 259, 1, None),
        ("albedo_predictor", 259, 3, None),
        ("outer_light", 144 if sphere_direction else 72, 3, math.log(0.5)),
        ("inner_light", pos + 72, 3, math.log(0.5)),
        ("inner_weight", pos + 39, 1, -0.95),
        ("transmisstion_weight", 259, 1, None),
        ("iors", 259, 1, None),
        ("refrac_light", 2 * (3 + 6 * refrac_freq), 3, math.log(0.5)),
    ]


def init_stage1_params(seed=6033, sphere_direction=False, sdf_bias=0.5, inv_s_init=0.3, refrac_freq=6, light_pos_freq=6):
    """OrderedDict name -> np.float32 array, in the reference module-construction order
    (renderer_zerothick.py:144-162)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    p = OrderedDict()

    # ---- SDFNetwork, geometric init (field.py:94-124) ----
    dims = [39] + [256] * 8 + [257]
    for l in range(9):
        in_dim = dims[l]
        out_dim = dims[l + 1] - dims[0] if l + 1 == 4 else dims[l + 1]
        if l == 8:
            w = rng.normal(math.sqrt(math.pi) / math.sqrt(in_dim), 1e-4, size=(out_dim, in_dim))
            b = np.full((out_dim,), -sdf_bias)
        elif l == 0:
            w = np.zeros((out_dim, in_dim))
            w[:, :3] = rng.normal(0.0, math.sqrt(2) / math.sqrt(out_dim), size=(out_dim, 3))
            b = np.zeros((out_dim,))
        elif l == 4:
            w = rng.normal(0.0, math.sqrt(2) / math.sqrt(out_dim), size=(out_dim, in_dim))
            w[:, -(dims[0] - 3):] = 0.0
            b = np.zeros((out_dim,))
        else:
            w = rng.normal(0.0, math.sqrt(2) / math.sqrt(out_dim), size=(out_dim, in_dim))
            b = np.zeros((out_dim,))
        _put_wn(p, f"sdf_network.lin{l}", w, b)

    p["deviation_network.variance"] = np.asarray(inv_s_init, np.float32)

    # ---- NeRF++ (field.py:246-261; rgb bias renderer_zerothick.py:156) ----
    in_ch, in_view, W = 84, 27, 256
    for i in range(8):
        k = in_ch if i == 0 else (W + in_ch if i == 5 else W)
        w, b = _linear_default(rng, W, k)
        p[f"outer_nerf.pts_linears.{i}.weight"], p[f"outer_nerf.pts_linears.{i}.bias"] = w, b
    w, b = _linear_default(rng, W // 2, in_view + W)
    p["outer_nerf.views_linears.0.weight"], p["outer_nerf.views_linears.0.bias"] = w, b
    w, b = _linear_default(rng, W, W)
    p["outer_nerf.feature_linear.weight"], p["outer_nerf.feature_linear.bias"] = w, b
    w, b = _linear_default(rng, 1, W)
    p["outer_nerf.alpha_linear.weight"], p["outer_nerf.alpha_linear.bias"] = w, b
    w, b = _linear_default(rng, 3, W // 2)
    p["outer_nerf.rgb_linear.weight"] = w
    p["outer_nerf.rgb_linear.bias"] = np.full((3,), math.log(0.5), np.float32)

    # ---- AppShadingNetwork predictors (field.py:575-611) ----
    p["color_network.FG_LUT"] = load_fg_lut()
    for name, k, n_out, last_bias in predictor_dims(sphere_direction, refrac_freq, light_pos_freq):
        chain = [(k, 256), (256, 256), (256, 256), (256, n_out)]
        for j, (ki, no) in enumerate(chain):
            w, b = _linear_default(rng, no, ki)
            if j == 3 and last_bias is not None:
                b = np.full((no,), last_bias, np.float32)
            _put_wn(p, f"color_network.{name}.{2 * j}", w, b)

    # ---- InfOutNetwork (field.py:1020-1036): constructed, never evaluated ----
    chain = [(63, 256), (256, 256), (256, 256), (256, 256), (256, 3)]
    for j, (ki, no) in enumerate(chain):
        w, b = _linear_default(rng, no, ki)
        _put_wn(p, f"infinity_far_bkgr.module0.{2 * j}", w, b)
    return p


def randomize_for_parity(params, seed=1):
    """Perturb the degenerate geometric init so that parity tests exercise every weight and every
    branch: non-zero embedding columns, non-unit weight_g, materials away from sigmoid(0).  Keeps
    the SDF close to a radius-0.5 sphere so rays still see a surface."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for k, v in params.items():
        v = v.copy()
        if k.endswith("FG_LUT") or k == "deviation_network.variance":
            pass
        elif k.startswith("sdf_network") and k.endswith("weight_v"):
            v += (0.02 * rng.standard_normal(v.shape) * (np.abs(v).mean() + 0.02)).astype(np.float32)
        elif k.endswith("weight_g"):
            v *= (1.0 + 0.05 * rng.standard_normal(v.shape)).astype(np.float32)
        elif k.endswith("bias") and not k.startswith("sdf_network"):
            v += (0.05 * rng.standard_normal(v.shape)).astype(np.float32)
        out[k] = v.astype(np.float32)
    return out


def count_params(params):
    return int(sum(v.size for k, v in params.items() if not k.endswith("FG_LUT")))


def init_ior_params(rng, prefix):
    """IoRNetwork / ThicknessNetwork (field.py:1046-1087): Sequential 0,2,4,5 = weight-normed linears 39-256-256-256-1."""
    p = OrderedDict()
    for idx, (k, n) in zip((0, 2, 4, 5), ((39, 256), (256, 256), (256, 256), (256, 1))):
        w, b = _linear_default(rng, n, k)
        _put_wn(p, f"{prefix}.module0.{idx}", w, b)
    return p


def init_stage2_params(seed=6033, inner_seed=7044, shader_cfg=None):
    """Stage2Renderer.state_dict() names in the reference's registration order (renderer_zerothick.py:919-978):
    nerf_network, IORs, stage1_network.*, IORs_pred, IoRint_pred, thickness_pred, outer_nerf, color_network.stage1_network.*
    (the same tensors again), sdf_network_inner, deviation_network_inner, color_network_inner."""
    shader_cfg = shader_cfg or {}
    sd = bool(shader_cfg.get('sphere_direction', False))
    rf = int(shader_cfg.get('refrac_freq', 6))
    s1 = init_stage1_params(seed, sphere_direction=sd, refrac_freq=rf)
    inner = init_stage1_params(inner_seed, sphere_direction=sd, refrac_freq=rf)
    rng = np.random.Generator(np.random.PCG64(seed + 17))
    p = OrderedDict()
    p['IORs'] = np.zeros(10, np.float32)          # the module's own parameter precedes its children in state_dict()
    extra_nerf = init_stage1_params(seed + 1)
    for k, v in extra_nerf.items():
        if k.startswith('outer_nerf.'):
            p['nerf_network.' + k[len('outer_nerf.'):]] = v
    for k, v in s1.items():
        p['stage1_network.' + k] = v
    p.update(init_ior_params(rng, 'IORs_pred'))
    p.update(init_ior_params(rng, 'IoRint_pred'))
    p.update(init_ior_params(rng, 'thickness_pred'))
    for k, v in init_stage1_params(seed + 2).items():
        if k.startswith('outer_nerf.'):
            p[k] = v
    for k, v in s1.items():
        p['color_network.stage1_network.' + k] = v
    for k, v in inner.items():
        if k.startswith('sdf_network.'):
            p['sdf_network_inner.' + k[len('sdf_network.'):]] = v
    p['deviation_network_inner.variance'] = inner['deviation_network.variance']
    for k, v in inner.items():
        if k.startswith('color_network.'):
            p['color_network_inner.' + k[len('color_network.'):]] = v
    return p


def init_stage2_thick_own_params(seed=7044, shader_cfg=None):
    """Initial values of what the NON-zero-thickness Stage2Renderer owns besides the stage-1 network (renderer.py:963-1024):
    nerf_network, IORs, the IoR / inner-IoR / thickness networks, the inner SDF (geometric init), its variance and the inner
    AppShadingNetwork_SpecInner predictors (8 position / 2 refraction frequencies, field.py:1321-1330)."""
    shader_cfg = shader_cfg or {}
    sd = bool(shader_cfg.get('sphere_direction', False))
    inner = init_stage1_params(seed, sphere_direction=sd, refrac_freq=int(shader_cfg.get('refrac_freq', 2)),
                               light_pos_freq=int(shader_cfg.get('light_pos_freq', 8)))
    rng = np.random.Generator(np.random.PCG64(seed + 17))
    p = OrderedDict()
    p['IORs'] = np.zeros(10, np.float32)
    for k, v in init_stage1_params(seed + 1).items():
        if k.startswith('outer_nerf.'):
            p['nerf_network.' + k[len('outer_nerf.'):]] = v
    p.update(init_ior_params(rng, 'IORs_pred'))
    p.update(init_ior_params(rng, 'IoRint_pred'))
    p.update(init_ior_params(rng, 'thickness_pred'))
    for k, v in inner.items():
        if k.startswith('sdf_network.'):
            p['sdf_network_inner.' + k[len('sdf_network.'):]] = v
    p['deviation_network_inner.variance'] = inner['deviation_network.variance']
    for k, v in inner.items():
        if k.startswith('color_network.'):
            p['color_network_inner.' + k[len('color_network.'):]] = v
    return p


def params_from_manifest(manifest, seed):
    """Seed-reproducible values for a list of (name, shape) state-dict entries, by name pattern only (no module semantics):
    weight-norm directions and plain weights ~ N(0, 1/fan_in), weight_g ~ 1 +- 5 %, biases ~ N(0, 0.05), anything 0-d or 1-d
    without a known suffix ~ N(0, 0.1).  Parity fixtures of models whose initialisers are not restated here carry the manifest and
    the seed (oracle/gen_golden_stage2_thick.py); the values have no training meaning."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for name, shape in manifest:
        shape = tuple(int(x) for x in shape)
        if name.endswith('FG_LUT'):
            raise ValueError("the FG LUT is an asset, not a parameter to generate")
        if name.endswith(('weight_v', 'weight')) and len(shape) == 2:
            v = rng.standard_normal(shape) / np.sqrt(max(shape[1], 1))
        elif name.endswith('weight_g'):
            v = 1.0 + 0.05 * rng.standard_normal(shape)
        elif name.endswith('bias'):
            v = 0.05 * rng.standard_normal(shape)
        else:
            v = 0.1 * rng.standard_normal(shape)
        out[name] = v.astype(np.float32)
    return out

