"""CPU tests of the host-side logic: parameter inventory, synthetic rays, loss registry, schedules, module boundary,
and the 2-rank data-parallel gradient all-reduce over gloo."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from helpers import oracle_cfg
from oracle import stage1_oracle as O


def test_param_inventory_matches_reference_counts():
    from nu_nerf_amd.params import init_stage1_params, count_params
    p = init_stage1_params(6033)
    assert count_params(p) == 2967393                      # SURVEY.md section 8(a) parameter inventory
    sub = lambda pre: sum(v.size for k, v in p.items() if k.startswith(pre) and not k.endswith('FG_LUT'))
    assert (sub('sdf_network'), sub('outer_nerf'), sub('color_network'), sub('infinity_far_bkgr')) == \
        (529076, 606596, 1616162, 215558)
    q = init_stage1_params(6033)
    assert all(np.array_equal(p[k], q[k]) for k in p)      # seed-reproducible
    assert p['sdf_network.lin3.weight_v'].shape == (217, 256) and p['sdf_network.lin8.bias'][0] == -0.5
    assert np.all(p['sdf_network.lin0.weight_v'][:, 3:] == 0) and np.all(p['sdf_network.lin4.weight_v'][:, -36:] == 0)
    assert abs(float(p['outer_nerf.rgb_linear.bias'][0]) - math.log(0.5)) < 1e-7


def test_lut_asset_is_the_reference_table():
    import hashlib
    from nu_nerf_amd.params import load_fg_lut
    lut = load_fg_lut()
    assert lut.shape == (1, 256, 256, 2) and lut.dtype == np.float32
    assert hashlib.sha256(lut.tobytes()).hexdigest() == "aee514f7c7e561a357e529567222da99e84886c31c46a32fe767a5b066bbe196"


def test_module_state_dict_has_reference_names_and_order():
    from nu_nerf_amd.renderer import NeROShapeRenderer
    from nu_nerf_amd.params import init_stage1_params
    torch.manual_seed(6033)
    m = NeROShapeRenderer({'is_nerf': True}, training=False)
    sd = m.state_dict()
    ref = init_stage1_params(1)
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == ref[k].shape for k in ref)
    # seeded construction is reproducible, default cfg keys preserved
    torch.manual_seed(6033)
    m2 = NeROShapeRenderer({'is_nerf': True}, training=False)
    assert torch.equal(m.sdf_network.lin2.weight_v, m2.sdf_network.lin2.weight_v)
    for k in ('n_samples', 'n_importance', 'n_bg_samples', 'up_sample_steps', 'train_ray_num', 'anneal_end', 'occ_loss_step'):
        assert k in m.cfg
    assert m.cfg['train_ray_num'] == 512 and m.cfg['n_samples'] == 64


def test_product_path_fails_loudly_without_gpu():
    from nu_nerf_amd.renderer import NeROShapeRenderer
    from nu_nerf_amd._lib import NuNerfLibraryError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = NeROShapeRenderer({'is_nerf': True}, training=False)
    o = torch.zeros(4, 3)
    with pytest.raises(NuNerfLibraryError):
        m.render(o, o + 1, torch.full((4, 1), 0.8), torch.full((4, 1), 4.5), step=0)


def test_network_holders_have_no_eager_evaluation():
    """The module objects that mirror the reference's networks only hold parameters; their arithmetic runs on the HIP kernels of the
    owning renderer's engine.  A holder called on its own raises instead of quietly computing with torch (no rocBLAS path exists in
    the package): SDFNetwork (field.py:133-170) and the IoR / thickness networks of stage 2 (field.py:1046-1087)."""
    from nu_nerf_amd.renderer import SDFNetwork
    from nu_nerf_amd.stage2 import IoRNetwork
    import inspect
    from nu_nerf_amd import torch_glue
    x = torch.zeros(5, 3)
    with pytest.raises(RuntimeError, match="parameter holder"):
        IoRNetwork()(x)
    with pytest.raises(RuntimeError, match="parameter holder"):
        SDFNetwork()(x)
    assert 'F.linear' not in inspect.getsource(torch_glue) and not hasattr(torch_glue, 'wn_linear')


def test_synthetic_rays():
    from nu_nerf_amd.synthetic import make_rays, make_cameras
    a, b = make_rays(256, seed=5), make_rays(256, seed=5)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert a['rays_o'].shape == (256, 3) and a['rays_d'].dtype == np.float32
    np.testing.assert_allclose(np.linalg.norm(a['rays_o'], axis=1), 4.0, rtol=1e-5)
    poses = make_cameras(10)
    R = poses[:, :, :3]
    np.testing.assert_allclose(np.einsum('nij,nkj->nik', R, R), np.broadcast_to(np.eye(3), (10, 3, 3)), atol=1e-5)
    # cameras look at the origin: -z axis points from the camera towards 0
    np.testing.assert_allclose(-R[:, :, 2], -poses[:, :, 3] / 4.0, atol=1e-5)


def test_loss_registry_matches_oracle_assembly():
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss
    g = torch.Generator().manual_seed(0)
    cfg = oracle_cfg()
    for step in (0, 500, 20000):
        out = {'loss_rgb': torch.rand(32, generator=g), 'gradient_error': torch.rand(100, generator=g),
               'std': torch.rand(1), 'loss_occ': torch.rand(1, generator=g), 'color_bkgr': torch.rand(32, 3, generator=g),
               'color_spec': torch.rand(32, 3, generator=g)}
        pts = torch.randn(500, 3, generator=g) * 0.8
        out['sdf_pts'], out['sdf_vals'] = pts, pts.norm(dim=-1) - 0.5 + 0.3 * torch.randn(500, generator=g)
        total, log = total_loss(out, [name2loss[n](cfg) for n in SPHEREPOT_LOSSES], step)
        ototal, oterms = O.assemble_losses(out, cfg, step)
        assert abs(float(total) - float(ototal)) < 1e-6
        assert set(k for k in log if k.startswith('loss')) == set(oterms.keys())


def test_lr_schedule_matches_oracle():
    import bench
    for s in (0, 100, 4999, 5000, 20000, 299999):
        assert abs(bench.warmup_cos_lr(s) - O.warmup_cos_lr(s)) < 1e-12


def test_ide_table_generation_matches_oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from scripts.gen_ide_table import table
    ml, mat = table()
    assert np.array_equal(mat.T, O._IDE_MAT) and np.array_equal(np.asarray(ml, np.float32), O._IDE_ML)


# ---------------------------------------------------------------------------------------------------------
# data-parallel over rays: 2 ranks, gloo
# ---------------------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nu_nerf_amd.renderer import NeROShapeRenderer
    from nu_nerf_amd.parallel import GradAllReducer, shard_rays, stage1_trainable_names
    torch.manual_seed(1)
    m = NeROShapeRenderer({'is_nerf': True}, training=False)
    red = GradAllReducer(m, world)
    names = stage1_trainable_names(m)
    named = dict(m.named_parameters())
    for i, n in enumerate(names):
        if i % 7 != 3:                     # some grads None (on every rank alike): they must STAY None -- no Adam state is born
            named[n].grad = torch.full_like(named[n], float(rank + 1) * (1 + (i % 5)))
    red.all_reduce()
    ok = red.gathered_calls == 1
    for i, n in enumerate(names):
        if i % 7 == 3:
            ok &= named[n].grad is None
        else:
            ok &= bool(torch.allclose(named[n].grad, torch.full_like(named[n], 1.5 * (1 + (i % 5)))))
    # the renderer's layout: every gradient a view of one flat buffer -> reduced in place, no copies
    flat = torch.full((red.numel,), float(rank + 1))
    off = 0
    for n in names:
        k = named[n].numel()
        named[n].grad = flat[off:off + k].view_as(named[n]).detach()     # what autograd stores: same storage, no view link
        off += k
    sf = red._shared_flat()
    ok &= sf is not None and sf.data_ptr() == flat.data_ptr() and sf.numel() == flat.numel()
    before = red.in_place_calls
    red.all_reduce()
    ok &= bool(torch.all(flat == 1.5)) and red.in_place_calls == before + 1
    ok &= float(named[names[5]].grad.flatten()[0]) == 1.5
    # step < freeze_inv_s_step: the variance has no gradient; its slot is a hole inside the flat range.  Still in place,
    # and .grad stays None (the gathered path used to zero-fill it, which gave Adam a state 15000 steps early)
    flat.fill_(float(rank + 1))
    named['deviation_network.variance'].grad = None
    sf = red._shared_flat()
    ok &= sf is not None and sf.numel() == flat.numel()
    before = red.in_place_calls
    red.all_reduce()
    ok &= red.in_place_calls == before + 1 and bool(torch.all(flat == 1.5)) and named['deviation_network.variance'].grad is None
    # a gradient that lives elsewhere (accumulated from two autograd paths): no clean range -> gathered path
    named[names[0]].grad = named[names[0]].grad.clone()
    ok &= red._shared_flat() is None
    # per-point means: ranks with 10 and 30 inner points weigh 0.5 and 1.5, so the averaged per-rank means equal the
    # mean over all 40 points
    w = red.point_weight(10 if rank == 0 else 30, torch.device('cpu'))
    ok &= abs(float(w) - (0.5 if rank == 0 else 1.5)) < 1e-6
    w2 = red.point_weight(torch.tensor([10 if rank == 0 else 30, 7], dtype=torch.int32), torch.device('cpu'))   # device-count form
    ok &= abs(float(w2) - float(w)) < 1e-6
    vals = torch.arange(10.) if rank == 0 else 10.0 + torch.arange(30.)
    share = torch.mean(vals * w).reshape(1)
    dist.all_reduce(share)
    ok &= abs(float(share) / world - float(torch.arange(40.).mean())) < 1e-4
    ok &= all(p.grad is None for n, p in m.named_parameters() if n.startswith(('color_network.iors', 'infinity_far_bkgr')))
    # several subset sizes in ONE all-reduce (inner points, occlusion-loss points, candidate rays); a subset that is empty on every
    # rank gets weight 1, one that is empty on this rank only gets 0
    cw = red.count_weights([10 if rank == 0 else 30, torch.tensor([4 if rank == 0 else 0]), 0], torch.device('cpu'))
    ok &= bool(torch.allclose(cw, torch.tensor([0.5, 2.0, 1.0]) if rank == 0 else torch.tensor([1.5, 0.0, 1.0])))
    # gathered path with a gradient that exists on ONE rank only (stage 2: a rank whose rays all miss the object trains no inner
    # network that step): the union over ranks decides, the other rank contributes zeros, both end with the same averaged gradient
    for p in red.params:
        p.grad = None
    a, b = named[names[0]], named[names[1]]
    a.grad = torch.full_like(a, float(rank + 1))
    if rank == 0:
        b.grad = torch.full_like(b, 4.0)
    before = red.gathered_calls
    red.all_reduce()
    ok &= red.gathered_calls == before + 1 and bool(torch.all(a.grad == 1.5)) and b.grad is not None and bool(torch.all(b.grad == 2.0))
    ok &= all(p.grad is None for p in red.params[2:])
    # the stage-2 module: the same reducer class; dead parameters at any nesting depth stay out of the bucket, aliases count once
    from nu_nerf_amd.stage2 import Stage2Renderer
    from nu_nerf_amd.lbvh import icosphere
    from nu_nerf_amd.parallel import stage1_trainable_names
    s2 = Stage2Renderer({'name': 's2', 'network': 'stage2', 'is_nerf': True, 'shader_config': {'sphere_direction': False, 'human_light': False},
                         'stage1_cfg': {'is_nerf': True}, 'stage1_mesh_arrays': icosphere(1, 0.5)}, training=False)
    n2 = stage1_trainable_names(s2)
    ok &= len(n2) == len(set(n2)) and not any('iors.' in n or 'infinity_far_bkgr.' in n for n in n2)
    ok &= any(n.startswith('IORs_pred.') for n in n2) and any(n.startswith('stage1_network.sdf_network.') for n in n2)
    red2 = GradAllReducer(s2, world)
    p0 = red2.params[0]
    p0.grad = torch.full_like(p0, float(rank))
    red2.all_reduce()
    ok &= bool(torch.all(p0.grad == 0.5))
    batch = {'rays_o': torch.arange(10.)[:, None].repeat(1, 3), 'rgbs': torch.arange(10.)[:, None].repeat(1, 3)}
    sh = shard_rays(batch, rank, world)
    ok &= sh['rays_o'].shape[0] == 5 and float(sh['rays_o'][0, 0]) == 5.0 * rank
    try:
        shard_rays({'rays_o': torch.zeros(9, 3)}, rank, world)
        ok = False
    except ValueError:
        pass
    q.put((rank, ok, red.numel))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_all_reduce_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == 2552665                         # params that receive gradients in stage 1


def test_capacity_classes_make_allocation_sizes_repeat():
    """Per-point buffers are allocated from a few capacity classes (engine._capacity): point counts that wander by +-10 %
    between steps must map to ONE size after the first request, small buffers stay exact."""
    from types import SimpleNamespace
    from nu_nerf_amd.engine import Stage1Engine
    eng = SimpleNamespace(_cap_classes=[], _ROW_QUANTUM=Stage1Engine._ROW_QUANTUM)
    cap = lambda *shape: Stage1Engine._capacity(eng, shape)
    assert cap(4096, 3) is None and cap(65536, 256) is None            # small: exact allocation
    first = cap(527000, 256)
    assert first % 16384 == 0 and 1.25 * 527000 <= first < 1.25 * 527000 + 16384
    for n in (520000, 545000, 500001, 560000, int(first)):              # the scene's outer-point counts
        assert cap(n, 256) == first
    inner = cap(128000, 288)
    assert inner != first and all(cap(n, 64) == inner for n in (115000, 135000, 150000))
    assert cap(3 * 128000 + 4096, 96) not in (None, inner)             # the row-batched outer_light input: its own class
    assert len(eng._cap_classes) == 3
    bigger = cap(int(first) + 1, 256)                                    # a record: a new class, once
    assert bigger > first and cap(int(first) + 5000, 256) == bigger


# ---------------------------------------------------------------------------------------------------------
# real-capture ray construction: network/renderer.py:346-378
# ---------------------------------------------------------------------------------------------------------
def test_process_ray_batch_and_human_poses_vs_reference_fixture():
    """tests/golden/ray_batch_std.npz comes from the reference's own methods (oracle/gen_golden_r2.py).  Under torch 2.10 the
    reference's get_human_coordinate_poses raises for more than one pose (in-place write into an expanded tensor,
    renderer.py:354-355), so the generator called it pose by pose; _process_ray_batch ends in that call, so its ray part
    (renderer.py:367-376: plain arithmetic on the inputs + the reference's near_far_from_sphere) was evaluated line by line
    -- the fixture says so in `process_ray_batch_restated`."""
    from helpers import check_ray_batch_against_reference_fixture
    check_ray_batch_against_reference_fixture('cpu')


def test_ray_store_construction_vs_reference_fixture_and_unmodified_yaml_names():
    """A YAML's own `database_name` (nerf/spherepot) no longer stops the module from being constructed for training: the
    image files are the caller's to load, `set_ray_store(imgs_info)` turns them into the ray store exactly as the reference's
    _init_dataset does after loading (fixture from the reference's own methods)."""
    from helpers import check_ray_store_against_reference_fixture
    from nu_nerf_amd.renderer import name2renderer
    info = check_ray_store_against_reference_fixture('cpu')
    net = name2renderer['shape']({'database_name': 'nerf/spherepot', 'is_nerf': True, 'train_ray_num': 16}, training=True)
    with pytest.raises(RuntimeError, match='set_ray_store'):
        net.train_step(0)
    net.set_ray_store(info, test_imgs_info=info)
    assert net.tbn == 90 and net.train_num == 3 and net.train_batch['rays_o'].shape == (90, 3)
    # the store is a permutation of the fixture's rays: every (origin, direction, colour) row is one of the reference's
    from helpers import golden
    g = golden("ray_store.npz")
    ref = np.concatenate([g['nerf_rays_o'], g['nerf_rays_d'], g['nerf_rgbs']], 1)
    got = torch.cat([net.train_batch[k] for k in ('rays_o', 'rays_d', 'rgbs')], 1).numpy()
    assert np.allclose(np.sort(got.view([('', got.dtype)] * 9), axis=0).view(got.dtype), np.sort(ref.view([('', ref.dtype)] * 9), axis=0).view(ref.dtype),
                       rtol=1e-6, atol=1e-6)
    real = name2renderer['shape']({'database_name': 'real/bear', 'is_nerf': False, 'train_ray_num': 16}, training=True)
    real.set_ray_store(info)
    assert sorted(real.train_batch) == ['dirs', 'idxs', 'rgbs'] and real.train_poses.shape == (3, 3, 4)


def test_imgs_info_downsample_follows_the_opencv_definitions():
    """renderer_zerothick.py:71-87 restated for tensors (OpenCV is absent offline, so cv2's own output cannot pin it): Gaussian blur
    with sigma = 1 / (3 ratio) and the odd kernel size of utils/base_utils.py:131-137 under BORDER_REFLECT101, bilinear resize
    with half-integer pixel centres, intrinsics scaled by diag(dw / w, dh / h, 1) -- checked against a direct numpy evaluation of
    those definitions, plus the invariants (constant image, shapes, nearest resize of depth / mask)."""
    from nu_nerf_amd.renderer import imgs_info_downsample
    rng = np.random.default_rng(3)
    n, h, w = 2, 11, 14
    img = rng.random((n, 3, h, w)).astype(np.float32)
    K = np.tile(np.array([[500.0, 0, 7.0], [0, 510.0, 5.5], [0, 0, 1]], np.float32), (n, 1, 1))
    depth = rng.random((n, h, w)).astype(np.float32)
    for ratio in (0.5, 0.25):
        out = imgs_info_downsample({'imgs': torch.from_numpy(img), 'Ks': torch.from_numpy(K), 'depths': torch.from_numpy(depth),
                                    'poses': torch.zeros(n, 3, 4)}, ratio)
        dh, dw = int(ratio * h), int(ratio * w)
        assert out['imgs'].shape == (n, 3, dh, dw) and out['depths'].shape == (n, dh, dw) and out['poses'].shape == (n, 3, 4)
        sigma = (1 / ratio) / 3
        ks = int(np.ceil(((sigma - 0.8) / 0.3 + 1) * 2 + 1))
        ks += 1 if ks % 2 == 0 else 0
        xs = np.arange(ks) - (ks - 1) / 2
        k1 = np.exp(-xs ** 2 / (2 * sigma ** 2)); k1 /= k1.sum()
        r = ks // 2

        def refl(i, m):                       # BORDER_REFLECT101: gfedcb|abcdefgh|gfedcba
            return -i if i < 0 else (2 * (m - 1) - i if i >= m else i)
        blur = np.zeros_like(img, dtype=np.float64)
        for y in range(h):
            for x in range(w):
                acc = 0.0
                for dy in range(-r, r + 1):
                    for dx in range(-r, r + 1):
                        acc = acc + k1[dy + r] * k1[dx + r] * img[:, :, refl(y + dy, h), refl(x + dx, w)]
                blur[:, :, y, x] = acc
        ref = np.zeros((n, 3, dh, dw))
        for y in range(dh):
            fy = min(max((y + 0.5) * h / dh - 0.5, 0.0), h - 1.0); y0 = int(np.floor(fy)); y1 = min(y0 + 1, h - 1); ty = fy - y0
            for x in range(dw):
                fx = min(max((x + 0.5) * w / dw - 0.5, 0.0), w - 1.0); x0 = int(np.floor(fx)); x1 = min(x0 + 1, w - 1); tx = fx - x0
                ref[:, :, y, x] = ((1 - ty) * ((1 - tx) * blur[:, :, y0, x0] + tx * blur[:, :, y0, x1])
                                   + ty * ((1 - tx) * blur[:, :, y1, x0] + tx * blur[:, :, y1, x1]))
        np.testing.assert_allclose(out['imgs'].numpy(), ref, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(out['Ks'].numpy(), np.diag([dw / w, dh / h, 1.0]).astype(np.float32)[None] @ K, rtol=1e-6)
        ys, xs2 = (np.arange(dh) * h // dh), (np.arange(dw) * w // dw)
        np.testing.assert_array_equal(out['depths'].numpy(), depth[:, ys][:, :, xs2])
        const = imgs_info_downsample({'imgs': torch.full((1, 3, h, w), 0.37), 'Ks': torch.from_numpy(K[:1])}, ratio)
        np.testing.assert_allclose(const['imgs'].numpy(), 0.37, rtol=1e-6)


def test_forward_routes_eval_requests_to_test_step_cpu_side():
    """forward({'index','eval','step'}) is the ValidationEvaluator's call (train/train_valid.py:25-29), first made at step 0
    (trainer_zero.py:174): it must reach test_step, not raise.  (The render itself needs the GPU: tests/test_eval_gpu.py.)"""
    from nu_nerf_amd.renderer import name2renderer
    from nu_nerf_amd.synthetic import make_image_rays, make_rays
    net = name2renderer['shape']({'database_name': 'synthetic/64', 'is_nerf': True}, training=True)
    seen = {}
    net.test_step = lambda index, step: seen.update(index=index, step=step) or {'ok': True}
    assert net({'index': torch.tensor([3]), 'eval': True, 'step': 0}) == {'ok': True} and seen == {'index': 3, 'step': 0}
    rays, h, w = make_image_rays(5, hw=16, downsample=0.5)
    assert (h, w) == (8, 8) and rays['rays_d'].shape == (64, 3) and np.allclose(rays['rays_o'], rays['rays_o'][0])
    # the image rays are the training pool's construction: pixel (i, j) of the full-resolution camera gives the same ray
    full, _, _ = make_image_rays(5, hw=16, downsample=1.0)
    dn = full['rays_d'] / np.linalg.norm(full['rays_d'], axis=1, keepdims=True)
    assert np.all(dn @ (-full['rays_o'][0] / np.linalg.norm(full['rays_o'][0])) > 0.85)       # looking at the origin (corner pixels: cos 0.89)


def test_stage2_parameter_hub_hands_each_parameter_its_slice_once_cpu_side():
    """nets.ParamHubFn (stage 2): network ops differentiate ONE token of the flat gradient buffer's shape; the hub sums their
    contributions, hands every touched parameter its slice once and leaves untouched parameters at grad None (optimizer state is
    created lazily: train_glue.FusedAdam).  Host logic only: a stand-in engine with the flat-buffer bookkeeping, no library."""
    from nu_nerf_amd import nets as N

    class Eng:
        dev = torch.device('cpu')
        grad_views = {'a.w': (0, (2, 3)), 'a.b': (6, (2,)), 'b.w': (8, (4,)), 'c.w': (12, (3,))}
        n_grad = 15

    eng = Eng()
    params = {n: torch.nn.Parameter(torch.zeros(s)) for n, (_, s) in eng.grad_views.items()}
    names = list(params)
    assert N._own_range(eng, ['a.w', 'a.b']) == (0, 8) and N._own_range(eng, ['b.w']) == (8, 12)
    with pytest.raises(AssertionError):
        N._own_range(eng, ['a.w', 'b.w'])                      # not contiguous: the op would leak another op's slots

    class Op(torch.autograd.Function):
        """A stand-in network op: its backward writes its own slots of a zero-filled flat buffer (as the ops' kernels and
        unpack_grads(flat, layers=...) do) and returns it."""
        @staticmethod
        def forward(ctx, x, own, fill, token):
            ctx.own, ctx.fill = own, fill
            return x * 2.0

        @staticmethod
        def backward(ctx, g):
            flat = torch.zeros(eng.n_grad)
            lo, hi = N._own_range(eng, ctx.own)
            flat[lo:hi] = ctx.fill
            return g * 2.0, None, None, N._token_grad(eng, flat, ctx.own)

    token = N.ParamHubFn.apply(eng, names, *[params[n] for n in names])
    x = torch.ones(3, requires_grad=True)
    y = Op.apply(x, ['a.w', 'a.b'], 1.0, token) + Op.apply(x, ['a.w', 'a.b'], 10.0, token) + Op.apply(x, ['b.w'], 100.0, token)
    y.sum().backward()
    assert torch.equal(params['a.w'].grad, torch.full((2, 3), 11.0)) and torch.equal(params['a.b'].grad, torch.full((2,), 11.0))
    assert torch.equal(params['b.w'].grad, torch.full((4,), 100.0))
    assert params['c.w'].grad is None                          # no op touched it
    assert torch.equal(x.grad, torch.full((3,), 6.0))


def test_stage2_thick_fixture_is_reproducible_from_its_manifest():
    """SURVEY 8(f) N3 (the non-zero-thickness stage-2 model; GPU parity in tests/test_stage2_thick_gpu.py): the reference-generated fixture
    tests/golden/stage2_thick_step6000_r24.npz (oracle/gen_golden_stage2_thick.py) carries a parameter manifest + seed instead of
    13 MB of weights, and the per-vertex Gaussian curvature its run used -- PyMesh's attribute in the reference, the angle-defect
    estimate here.  Pin both: the generator reproduces the same values, the product's curvature equals the fixture's."""
    import hashlib
    from helpers import golden
    from nu_nerf_amd.params import params_from_manifest
    from nu_nerf_amd.lbvh import icosphere, vertex_normals_and_curvature
    g = golden("stage2_thick_step6000_r24.npz")
    names = [str(n) for n in g['manifest_names']]
    shapes = [tuple(int(x) for x in str(s).split(',') if x) for s in g['manifest_shapes']]
    assert len(names) == 197 and len(set(names)) == 197
    assert {n.split('.')[0] for n in names} == {'IORs', 'nerf_network', 'IORs_pred', 'IoRint_pred', 'thickness_pred', 'sdf_network_inner',
                                                'deviation_network_inner', 'color_network_inner'}
    p = params_from_manifest(list(zip(names, shapes)), int(g['manifest_seed']))
    q = params_from_manifest(list(zip(names, shapes)), int(g['manifest_seed']))
    h = hashlib.sha256()
    for n in names:
        assert p[n].shape == shapes[names.index(n)] and p[n].dtype == np.float32 and np.array_equal(p[n], q[n])
        h.update(p[n].tobytes())
    assert sum(v.size for v in p.values()) > 2_000_000
    V, Fc = icosphere(3, 0.5)
    _, curv = vertex_normals_and_curvature(torch.from_numpy(V), torch.from_numpy(Fc.astype(np.int64)))
    np.testing.assert_allclose(curv.numpy(), g['vertex_gaussian_curvature'], rtol=1e-6, atol=0)
    # what the reference run produced with those inputs (the numbers the product will have to match)
    assert g['out_ray_rgb'].shape == (24, 3) and int(g['out_tir_mask'].sum()) == 23 and len(g['grad_names']) == 295
    assert [g['path%d' % i].shape[:2] for i in range(3)] == [(24, 64), (15, 128), (15, 64)]


def test_stage2_thick_module_has_the_reference_state_dict_and_registry_entry():
    """The non-zero-thickness Stage2Renderer (network/renderer.py:907-1024) as a module, on the CPU: `name2renderer['stage2']` of the
    `zero_thickness: False` registry, the reference's 565 state_dict() keys in the reference's order (from the fixture the
    reference itself wrote), parameter shapes of the manifest, and the initial values a from-scratch run starts with (inner SDF:
    geometric init; last biases of the light stacks; refraction-light cap)."""
    from helpers import golden
    from nu_nerf_amd.stage2_thick import name2renderer, Stage2Renderer, AppShadingNetworkSpecInner
    from nu_nerf_amd.renderer_std import NeROShapeRenderer as StdStage1
    from nu_nerf_amd.lbvh import icosphere
    g = golden("stage2_thick_step6000_r24.npz")
    assert name2renderer['stage2'] is Stage2Renderer and name2renderer['shape'] is StdStage1
    shader = {'sphere_direction': True, 'human_light': False, 'light_exp_max': 5.0}
    cfg = {'name': 's2t', 'network': 'stage2', 'get_mask': False, 'is_nerf': False, 'shader_config': shader,
           'stage1_cfg': {'name': 's1', 'network': 'shape', 'get_mask': False, 'is_nerf': False, 'shader_config': shader},
           'stage1_mesh_arrays': icosphere(2, 0.5)}
    net = Stage2Renderer(cfg, training=False)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g['state_dict_keys']] and len(sd) == 565
    for n, shp in zip(g['manifest_names'], g['manifest_shapes']):
        assert tuple(sd[str(n)].shape) == tuple(int(x) for x in str(shp).split(',') if x), n
    assert sd['infinity_far_bkgr.module0.0.weight_v'].data_ptr() == sd['stage1_network.infinity_far_bkgr.module0.0.weight_v'].data_ptr()
    inner = net.color_network_inner
    assert isinstance(inner, AppShadingNetworkSpecInner) and inner.cfg['light_pos_freq'] == 8 and inner.cfg['refrac_freq'] == 2
    assert inner.cfg['refrac_exp_max'] == -0.2
    assert tuple(sd['color_network_inner.inner_light.0.weight_v'].shape) == (256, 51 + 72)
    assert tuple(sd['color_network_inner.refrac_light.0.weight_v'].shape) == (256, 30)
    np.testing.assert_allclose(sd['color_network_inner.outer_light.6.bias'].numpy(), np.log(0.5), rtol=1e-6)
    np.testing.assert_allclose(sd['sdf_network_inner.lin8.bias'].numpy(), -0.5, rtol=1e-6)          # geometric init, bias 0.5


def test_presplit_weight_plane_layout_matches_the_documented_index_formula():
    """include/nu_nerf.h NuGemmNT.B6: bf16 index of element (n, k) of plane p =
    ((n >> 8) * (ld >> 4) + (k >> 4)) * 12288 + (n & 255) * 48 + 16 p + ((k & 15) ^ (n & 8)), planes = exact hi / mid / lo split.
    The host-side builder the GPU tests compare the pack launch against (tests/test_gemm_gpu._p3) must follow that formula."""
    import torch
    from test_gemm_gpu import _p3
    torch.manual_seed(3)
    for rows, ld in ((300, 64), (512, 288), (130, 1024)):
        W = torch.randn(rows, ld) * torch.exp(2 * torch.randn(rows, 1))
        flat = _p3(W)
        assert flat.numel() == 3 * ((rows + 255) // 256 * 256) * ld
        hi = W.bfloat16()
        mid = (W - hi.float()).bfloat16()
        lo = (W - hi.float() - mid.float()).bfloat16()
        assert torch.equal(hi.float() + mid.float() + lo.float(), W)          # the split is exact
        g = torch.Generator().manual_seed(rows)
        for _ in range(400):
            n = int(torch.randint(0, rows, (1,), generator=g))
            k = int(torch.randint(0, ld, (1,), generator=g))
            base = ((n >> 8) * (ld >> 4) + (k >> 4)) * 12288 + (n & 255) * 48 + ((k & 15) ^ (n & 8))
            for p_, plane in enumerate((hi, mid, lo)):
                assert flat[base + 16 * p_] == plane[n, k], (rows, ld, n, k, p_)
