"""GPU parity of the network-level ops WITH input gradients (nu_nerf_amd/nets.py) against torch autograd on the CPU oracle:
SDF value / normal / second-order input gradient, NeRF++ input gradients, predictor stacks, batched materials."""
import numpy as np
import pytest
import torch

from helpers import parity_params, rel_err
from oracle import stage1_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nets(gpu):
    from nu_nerf_amd.renderer import NeROShapeRenderer
    from nu_nerf_amd.params import init_stage1_params, randomize_for_parity
    from nu_nerf_amd.nets import Stage1Nets
    net = NeROShapeRenderer({'is_nerf': True}, training=False)
    net.load_param_dict(randomize_for_parity(init_stage1_params(6033), seed=1))
    net = net.to(gpu)
    eng = net.engine()
    eng.pack()
    return net, Stage1Nets(eng, net._named())


def test_sdf_value_normal_and_input_gradient(nets, gpu):
    net, N = nets
    g = np.random.Generator(np.random.PCG64(21))
    P = 300
    x = torch.from_numpy(g.uniform(-0.8, 0.8, (P, 3)).astype(np.float32))
    cy = torch.from_numpy(g.standard_normal((P, 257)).astype(np.float32))
    cn = torch.from_numpy(g.standard_normal((P, 3)).astype(np.float32))
    # oracle: y(x), n(x) with x requiring grad -> dL/dx incl. the Hessian-vector term
    Pm = parity_params(requires_grad=True)
    xo = x.clone().requires_grad_(True)
    y = O.sdf_forward(Pm, xo)
    (n,) = torch.autograd.grad(y[:, :1], xo, torch.ones(P, 1), create_graph=True)
    L = (y * cy).sum() + (n * cn).sum()
    gx, gw = torch.autograd.grad(L, [xo, Pm['sdf_network.lin2.weight_v']])
    # HIP
    xg = x.to(gpu).requires_grad_(True)
    net.zero_grad()
    yg, ng = N.sdf(xg)
    assert rel_err(yg.cpu(), y) < 1e-5 and rel_err(ng.cpu(), n) < 1e-5
    ((yg * cy.to(gpu)).sum() + (ng * cn.to(gpu)).sum()).backward()
    assert rel_err(xg.grad.cpu(), gx) < 1e-4
    assert rel_err(net.sdf_network.lin2.weight_v.grad.cpu(), gw) < 1e-4
    # first-order only (no cotangent on n)
    xg2 = x.to(gpu).requires_grad_(True)
    y2, _ = N.sdf(xg2)
    (y2 * cy.to(gpu)).sum().backward()
    xo2 = x.clone().requires_grad_(True)
    (O.sdf_forward(parity_params(), xo2) * cy).sum().backward()
    assert rel_err(xg2.grad.cpu(), xo2.grad) < 1e-4


def test_nerf_input_gradients(nets, gpu):
    net, N = nets
    g = np.random.Generator(np.random.PCG64(22))
    P = 257
    x = torch.from_numpy((g.standard_normal((P, 3)) * 2.0).astype(np.float32))
    x = x * (1.2 / x.norm(dim=1, keepdim=True)).clamp(min=1.0)            # outside the unit sphere
    d = torch.nn.functional.normalize(torch.from_numpy(g.standard_normal((P, 3)).astype(np.float32)), dim=-1)
    cs = torch.from_numpy(g.standard_normal(P).astype(np.float32))
    cr = torch.from_numpy(g.standard_normal((P, 3)).astype(np.float32))
    Pm = parity_params(requires_grad=True)
    xo, do = x.clone().requires_grad_(True), d.clone().requires_grad_(True)
    nn = xo.norm(dim=-1, keepdim=True)
    sig, rgb = O.nerf_forward(Pm, torch.cat([xo / nn, 1.0 / nn], -1), -do)
    ((sig[:, 0] * cs).sum() + (rgb * cr).sum()).backward()
    xg, dg = x.to(gpu).requires_grad_(True), d.to(gpu).requires_grad_(True)
    net.zero_grad()
    s2, r2 = N.nerf(xg, dg)
    assert rel_err(s2.cpu(), sig[:, 0]) < 1e-5 and rel_err(r2.cpu(), rgb) < 1e-5
    ((s2 * cs.to(gpu)).sum() + (r2 * cr.to(gpu)).sum()).backward()
    assert rel_err(xg.grad.cpu(), xo.grad) < 1e-4
    assert rel_err(dg.grad.cpu(), do.grad) < 1e-4
    assert rel_err(net.outer_nerf.pts_linears[5].weight.grad.cpu(), Pm['outer_nerf.pts_linears.5.weight'].grad) < 1e-4


def test_predictor_stack_and_materials(nets, gpu):
    net, N = nets
    g = np.random.Generator(np.random.PCG64(23))
    P = 200
    Pm = parity_params(requires_grad=True)
    X = torch.from_numpy(g.standard_normal((P, 111)).astype(np.float32) * 0.5)
    co = torch.from_numpy(g.standard_normal((P, 3)).astype(np.float32))
    Xo = X.clone().requires_grad_(True)
    (O.predictor(Pm, 'color_network.inner_light', Xo, 'none') * co).sum().backward()
    Xg = X.to(gpu).requires_grad_(True)
    net.zero_grad()
    out = N.predictor('inner_light', Xg)
    (out * co.to(gpu)).sum().backward()
    assert rel_err(Xg.grad.cpu(), Xo.grad) < 1e-4
    assert rel_err(net.color_network.inner_light[0].weight_v.grad.cpu(), Pm['color_network.inner_light.0.weight_v'].grad) < 1e-4
    # materials
    feat = torch.from_numpy((0.3 * g.standard_normal((P, 256))).astype(np.float32))
    x = torch.from_numpy(g.uniform(-0.6, 0.6, (P, 3)).astype(np.float32))
    cm = torch.from_numpy(g.standard_normal((P, 6)).astype(np.float32))
    fo, xo = feat.clone().requires_grad_(True), x.clone().requires_grad_(True)
    fx = torch.cat([fo, xo], -1)
    raws = [O.predictor(Pm, f'color_network.{n}', fx, 'none') for n in
            ('metallic_predictor', 'roughness_predictor', 'albedo_predictor', 'transmisstion_weight')]
    (torch.cat(raws, -1) * cm).sum().backward()
    fg, xg = feat.to(gpu).requires_grad_(True), x.to(gpu).requires_grad_(True)
    m = N.materials(fg, xg)
    assert rel_err(m.cpu(), torch.cat(raws, -1)) < 1e-5
    (m * cm.to(gpu)).sum().backward()
    assert rel_err(fg.grad.cpu(), fo.grad) < 1e-4 and rel_err(xg.grad.cpu(), xo.grad) < 1e-4


CFG16 = {'is_nerf': True, 'n_samples': 32, 'n_importance': 32, 'n_bg_samples': 16, 'freeze_inv_s_step': 15000,
         'apply_occ_loss': True, 'occ_loss_step': 15000, 'eikonal_weight': 0.1, 'mlp_dtype': 'bf16'}


@pytest.mark.parametrize("P", [1, 31, 64, 1000, 8192, 20000, 30001])
def test_fused_sdf_forward_is_bit_identical_to_the_layered_path(gpu, P):
    """csrc/fused_sdf.hip: the no-gradient SDF forward as one kernel (field.py:133-153 + the embedding :47-61 + the skip concat
    :142-143) -- same MFMA k order, same softplus, same head reduction as the layer-by-layer path, so the results are equal BIT for
    bit (32- and 64-row tiles, ragged last tile, one point); and equal to a float64 torch evaluation of the network to fp32
    rounding.  x_ld = 3 (sampler points) and 8 (point records)."""
    import ctypes
    from nu_nerf_amd.engine import addr
    from test_stage1_gpu import make_net
    net = make_net(gpu)
    eng = net.engine()
    eng.pack()
    torch.manual_seed(P)
    for x_ld in (3, 8):
        X = (torch.rand(P, x_ld, device=gpu) * 2 - 1) * 0.9
        eng._fused_sdf, eng._FUSED_SDF_MAX_POINTS = True, 1 << 30
        fused = eng.sdf_forward(addr(X), x_ld, P, keep=False, want_feat=False)['sdf'].clone()
        eng._fused_sdf = False
        layered = eng.sdf_forward(addr(X), x_ld, P, keep=False, want_feat=False)['sdf'].clone()
        eng._fused_sdf = True
        assert torch.equal(fused, layered), float((fused - layered).abs().max())
    # the bf16-storage arithmetic (mlp_dtype 'bf16', BASELINE config 4) has a fused kernel of its own (sdf_fused16_fwd_kernel): bit
    # for bit the layered bf16-storage path (bf16 weight tables, activations rounded to bf16 between layers, fp32 accumulation)
    if P in (1, 31, 1000, 20000, 30001):           # 32-, 64- and 128-row tiles (the last: 512 threads, two row groups per weight stream)
        net16 = make_net(gpu, cfg=dict(CFG16))
        e16 = net16.engine()
        assert e16.h16
        e16.pack()
        X16 = (torch.rand(P, 3, device=gpu) * 2 - 1) * 0.9
        e16._fused_sdf = True
        f16 = e16.sdf_forward(addr(X16), 3, P, keep=False, want_feat=False)['sdf'].clone()
        e16._fused_sdf = False
        l16 = e16.sdf_forward(addr(X16), 3, P, keep=False, want_feat=False)['sdf'].clone()
        assert torch.equal(f16, l16), float((f16 - l16).abs().max())
    # float64 reference of SDFNetwork.forward with the module's own weights
    sd = {k: v.detach().double() for k, v in net.state_dict().items() if k.startswith('sdf_network.')}
    x = X[:, :3].double()
    emb = [x]
    for k in range(6):
        emb += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    inp = torch.cat(emb, -1)
    h = inp
    for l in range(9):
        v, g, b = (sd[f'sdf_network.lin{l}.{n}'] for n in ('weight_v', 'weight_g', 'bias'))
        W = g * v / v.norm(dim=1, keepdim=True)
        if l == 4:
            h = torch.cat([h, inp], -1) / 2 ** 0.5
        h = h @ W.t() + b
        if l < 8:
            h = torch.nn.functional.softplus(h, beta=100)
    torch.testing.assert_close(fused.double(), h[:, 0], rtol=2e-5, atol=2e-6)


def test_pack_writes_the_presplit_weight_planes_of_every_table(gpu):
    """mlp_dtype 'bf16x6': the pack launch writes, next to every fp32 weight table, its exact hi / mid / lo bf16 split in the layout
    gemm_nt6_kernel reads (include/nu_nerf.h NuGemmNT.B6: 256-row blocks, k-group major, half-swap in rows with bit 3 set).  Every
    table of the engine -- per-layer tables, the stacked material tables (row offsets), the transposed tables (column offsets) -- must
    equal the same split computed from the fp32 table on the host side (test_gemm_gpu._p3), bit for bit."""
    from test_gemm_gpu import _p3
    from test_stage1_gpu import make_net
    net = make_net(gpu, cfg=dict(CFG16, mlp_dtype='bf16x6'))
    eng = net.engine()
    assert eng.bf16 == 2 and eng.w6 and not eng.h16
    eng.pack()
    torch.cuda.synchronize()
    seen = 0
    for lay in eng.layers:
        for tab in (lay.Wp, lay.WpT):
            if tab is None:
                continue
            t = tab[0]
            twin = eng._tw[id(t)]
            want = _p3(t.reshape(-1, t.shape[-1]))
            assert twin.numel() == want.numel(), (lay.name if hasattr(lay, 'name') else '?', tuple(t.shape))
            assert torch.equal(twin.view(torch.int16), want.view(torch.int16)), (tuple(t.shape), tab[1])
            seen += 1
    assert seen >= 60
