"""Network-level differentiable ops over the HIP engine, WITH input gradients.

Stage 1 never needs d L / d x (sample positions do not depend on parameters), so its whole render_core is one fused
op (renderer._RenderCoreFn).  Stage 2 does: refracted directions, hit points and therefore every sample position
depend on the learned IoR (renderer_zerothick.py:1642-1684).  These Functions expose the three MLP stacks at network
granularity -- value, SDF normal (second order) and gradients w.r.t. parameters AND inputs -- so that the ragged
per-bounce bookkeeping of stage 2 can stay in torch while every GEMM runs in the HIP library.

  SdfFn       x[P,3]                -> y[P,257] (sdf | feature), n[P,3] = d sdf / d x       (field.py:133-170)
  NerfFn      x[P,3], dir[P,3]      -> sigma[P], rgb_raw[P,3]  (inputs (x/|x|, 1/|x|), -dir) (field.py:265-289)
  StackFn     X[rows, K]            -> raw[rows, n_out]   one make_predictor stack          (field.py:371-408)
  MaterialsFn feat[P,256], x[P,3]   -> raw[P,6] (metallic, roughness, albedo(3), transmission; pre-sigmoid)
"""
import os

import torch

from .engine import addr


def _numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


def _grads_from_flat(eng, flat, names):
    cache = eng.__dict__.setdefault('_slice_cache', {})
    plan = cache.get(id(names))
    if plan is None or plan[0] is not names:
        plan = cache[id(names)] = (names, [(eng.grad_views[n][0], _numel(eng.grad_views[n][1]), eng.grad_views[n][1]) for n in names])
    return [flat[off:off + numel].view(shape) for off, numel, shape in plan[1]]


def _own_range(eng, names):
    """[lo, hi) of the flat gradient buffer that the parameters `names` occupy (one network's parameters are contiguous: each
    op's backward writes exactly this range of its zero-filled buffer)."""
    cache = eng.__dict__.setdefault('_range_cache', {})
    hit = cache.get(id(names))
    if hit is not None and hit[0] is names:
        return hit[1]
    lo, hi, tot = None, 0, 0
    for n in names:
        off, shape = eng.grad_views[n]
        numel = _numel(shape)
        lo = off if lo is None else min(lo, off)
        hi = max(hi, off + numel)
        tot += numel
    assert lo is not None and tot == hi - lo, "a network's parameters are expected to be contiguous in the flat gradient buffer"
    cache[id(names)] = (names, (lo, hi))
    return lo, hi


def _token_grad(eng, flat, names):
    """This op's share of the flat gradient buffer: `flat` itself -- it was zero-filled, the op's kernels wrote its bias gradients
    and unpack_grads(flat, layers=<the op's network>) its weight gradients, nothing else."""
    eng._hub_touched.update(names)           # parameters no op touched keep grad None (optimizer state is created lazily)
    return flat


class ParamHubFn(torch.autograd.Function):
    """All parameters of one engine -> one token of the flat gradient buffer's shape.  Every network op of a pass takes the token
    instead of its ~50 parameters and returns its flat gradient contribution for it; autograd sums those (one add per op) and
    this node hands each parameter its slice ONCE -- instead of one small accumulation per parameter per op."""

    @staticmethod
    def forward(ctx, eng, names, *params):
        ctx.eng, ctx.names = eng, names
        eng._hub_touched = set()
        return torch.empty(eng.n_grad, device=eng.dev)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None, None) + (None,) * len(ctx.names)
        touched = ctx.eng._hub_touched
        grads = _grads_from_flat(ctx.eng, g, ctx.names)
        return (None, None) + tuple(gr if n in touched else None for n, gr in zip(ctx.names, grads))


class SdfFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, names, token, need_normal=True):
        x = x.detach().contiguous()
        P = x.shape[0]
        a = eng.sdf_forward(addr(x), 3, P, keep=True, want_feat=True)
        ctx.eng, ctx.a, ctx.names = eng, a, names
        ctx.set_materialize_grads(False)
        if not need_normal:                  # value and feature only (the surface points of stage 2): no reverse sweep
            return a['YX'][:, :257].clone(), None
        n = eng.sdf_normal(a)
        return a['YX'][:, :257].clone(), n.clone()

    @staticmethod
    def backward(ctx, dy, dn):
        eng, a = ctx.eng, ctx.a
        eng.op_begin()
        P = a['P']
        flat = eng.zeros(eng.n_grad)
        dYX = eng.zeros(P, 288)
        if dy is not None:
            dYX[:, :257] = dy
        nbar = dn.contiguous() if dn is not None else None
        dx = eng.empty(P, 3)
        eng.sdf_backward(a, dYX, nbar, flat, dx=dx)
        eng.unpack_grads(flat, eng.sdf)
        eng.op_end()
        ctx.a = None
        return None, dx, None, _token_grad(eng, flat, ctx.names), None


class NerfFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, d, names, token):
        P = x.shape[0]
        pt = eng.zeros(P, 8)
        pt[:, :3] = x.detach()
        pt[:, 4:7] = d.detach()
        b = eng.nerf_forward(pt, None, P, None, None)
        ctx.eng, ctx.b, ctx.pt, ctx.names = eng, b, pt, names
        ctx.set_materialize_grads(False)
        return b['sig'].clone(), b['rgb'][:, :3].clone()

    @staticmethod
    def backward(ctx, dsig, drgb):
        eng, b, pt = ctx.eng, ctx.b, ctx.pt
        eng.op_begin()
        P = b['P']
        flat = eng.zeros(eng.n_grad)
        ds = dsig.contiguous() if dsig is not None else eng.zeros(P)
        dr = eng.zeros(P, 4)
        if drgb is not None:
            dr[:, :3] = drgb
        dx, dd = eng.empty(P, 3), eng.empty(P, 3)
        eng.nerf_backward(b, pt, None, None, None, flat, dsig=ds, drgb=dr, dx=dx, ddir=dd)
        eng.unpack_grads(flat, eng.nerf_all)
        eng.op_end()
        ctx.b = None
        return None, dx, dd, None, _token_grad(eng, flat, ctx.names)


class StackFn(torch.autograd.Function):
    """One make_predictor stack (3 hidden ReLU layers + skinny head) on already-encoded inputs."""

    @staticmethod
    def forward(ctx, eng, layers, X, names, token):
        rows, K = X.shape
        Kp = layers[0].Kp
        if K == Kp:                 # rows already in the stack's padded layout (stage2_ops.shade_encode): no copy
            Xp = X.detach().contiguous()
        else:
            Xp = eng.zeros(rows, Kp)
            Xp[:, :K] = X.detach()
        Hs = eng.relu_stack_fwd(layers, Xp, Kp, rows)
        head = layers[3]
        no = head.N
        out = eng.empty(rows, 4)
        eng.skinny_fwd(addr(Hs[2]), 256, rows, 256, addr(*head.Wp), 256, addr(head.b), no, addr(out), 4)
        ctx.eng, ctx.layers, ctx.Xp, ctx.Hs, ctx.names, ctx.K, ctx.no = eng, layers, Xp, Hs, names, K, no
        ctx.set_materialize_grads(False)
        return out[:, :no].clone()

    @staticmethod
    def backward(ctx, dout):
        eng, layers, Xp, Hs = ctx.eng, ctx.layers, ctx.Xp, ctx.Hs
        eng.op_begin()
        rows, Kp = Xp.shape
        flat = eng.zeros(eng.n_grad)
        dy = eng.zeros(rows, 4)
        if dout is not None:
            dy[:, :ctx.no] = dout
        head = layers[3]
        dH3 = eng.empty(rows, 256)
        eng.skinny_bwd(addr(dy), 4, addr(Hs[2]), 256, rows, 256, addr(*head.Wp), 256, ctx.no, addr(dH3), 256, 1, 0,
                       addr(*head.dWp), head.ldd, addr(flat, head.db_off))
        dX = eng.empty(rows, Kp)
        eng.relu_stack_bwd(layers, Xp, Kp, rows, Hs, dH3, flat, dX, Kp, Kp)
        eng.unpack_grads(flat, layers)
        eng.op_end()
        ctx.Hs = ctx.Xp = None
        return None, None, dX[:, :ctx.K], None, _token_grad(eng, flat, ctx.names)


class StacksFn(torch.autograd.Function):
    """SEVERAL make_predictor stacks of one shading call as one op: the stacks are independent of each other, so level j of all of
    them is one launch (engine.nt_batch: one persistent tile list) and their weight gradients one queue (engine.wgrad_batch) -- at
    the 10^2-10^4 rows of a stage-2 shading call every one of these launches fills a fraction of the chip.  Same kernels and the
    same per-row arithmetic as one StackFn per stack; one zero-filled flat gradient buffer and one unpack for the lot (the stacks'
    layers are consecutive in the engine's layer table)."""

    @staticmethod
    def forward(ctx, eng, stacks, layers_union, names, token, *Xs):
        from .engine import EPI_BIAS_RELU
        Xps, Hss = [], []
        for layers, X in zip(stacks, Xs):
            rows, K = X.shape
            Kp = layers[0].Kp
            if K == Kp:
                Xp = X.detach().contiguous()
            else:
                Xp = eng.zeros(rows, Kp)
                Xp[:, :K] = X.detach()
            Xps.append(Xp)
            Hss.append([eng.empty(rows, 256) for _ in range(3)])
        for j in range(3):
            eng.nt_batch([eng.nt_desc(addr(Xp if j == 0 else Hs[j - 1]), layers[j].Kp if j == 0 else 256, addr(*layers[j].Wp), layers[j].Kp,
                                      Xp.shape[0], 256, layers[j].Kp, addr(Hs[j]), 256, EPI_BIAS_RELU, bias=addr(layers[j].b),
                                      mask=eng.relu_mask(Hs[j], Xp.shape[0], 256))
                          for layers, Xp, Hs in zip(stacks, Xps, Hss)])
        outs = []
        for layers, Xp, Hs in zip(stacks, Xps, Hss):
            head = layers[3]
            out = eng.empty(Xp.shape[0], 4)
            eng.skinny_fwd(addr(Hs[2]), 256, Xp.shape[0], 256, addr(*head.Wp), 256, addr(head.b), head.N, addr(out), 4)
            outs.append(out[:, :head.N].clone())
        ctx.eng, ctx.stacks, ctx.Xps, ctx.Hss, ctx.names, ctx.Ks = eng, stacks, Xps, Hss, names, [X.shape[1] for X in Xs]
        ctx.layers_union = layers_union
        return tuple(outs)       # (an output nobody differentiates arrives as zeros in backward: every stack's packed gradient is rewritten)

    @staticmethod
    def backward(ctx, *douts):
        from .engine import EPI_MUL_DRELU, EPI_PLAIN
        eng, stacks, Xps, Hss = ctx.eng, ctx.stacks, ctx.Xps, ctx.Hss
        eng.op_begin()
        flat = eng.zeros(eng.n_grad)
        live = list(range(len(stacks)))
        dXs = [None] * len(stacks)
        with eng.wgrad_batch() as wb:
            dAs = {}
            for k in live:
                layers, Xp, Hs = stacks[k], Xps[k], Hss[k]
                rows, head = Xp.shape[0], layers[3]
                dy = wb.keep(douts[k].contiguous())                    # [rows, n_out]: read with its own leading dimension
                dH3 = wb.keep(eng.empty(rows, 256))
                eng.skinny_bwd(addr(dy), head.N, addr(Hs[2]), 256, rows, 256, addr(*head.Wp), 256, head.N, addr(dH3), 256, 1, 0,
                               addr(*head.dWp), head.ldd, addr(flat, head.db_off))
                dAs[k] = dH3
                dXs[k] = eng.empty(rows, layers[0].Kp)
            for j in (2, 1, 0):
                descs = []
                for k in live:
                    layers, Xp, Hs = stacks[k], Xps[k], Hss[k]
                    rows, lay = Xp.shape[0], layers[j]
                    u, ldu = (Xp, lay.Kp) if j == 0 else (Hs[j - 1], 256)
                    eng.wgrad(addr(dAs[k]), 256, addr(u), ldu, rows, 256, lay.Kp, addr(*lay.dWp), lay.ldd, addr(flat, lay.db_off))
                    if j > 0:
                        nxt = wb.keep(eng.empty(rows, 256))
                        descs.append(eng.nt_desc(addr(dAs[k]), 256, addr(*lay.WpT), lay.ldT, rows, 256, 256, addr(nxt), 256, EPI_MUL_DRELU,
                                                 H=addr(Hs[j - 1]), ldh=256, mask=getattr(Hs[j - 1], '_nu_mask', None)))
                        dAs[k] = nxt
                    else:
                        descs.append(eng.nt_desc(addr(dAs[k]), 256, addr(*lay.WpT), lay.ldT, rows, lay.Kp, 256, addr(dXs[k]), lay.Kp, EPI_PLAIN))
                eng.nt_batch(descs)
        eng.unpack_grads(flat, ctx.layers_union)
        eng.op_end()
        ctx.Hss = ctx.Xps = None
        return (None, None, None, None, _token_grad(eng, flat, ctx.names)) + tuple(dX[:, :K] for dX, K in zip(dXs, ctx.Ks))


class MaterialsFn(torch.autograd.Function):
    """The four material predictors batched (metallic, roughness, albedo, transmission) on [feature, x]."""

    @staticmethod
    def forward(ctx, eng, feat, x, names, token):
        from .engine import EPI_BIAS_RELU
        P = feat.shape[0]
        YX = eng.zeros(P, 288)
        YX[:, 1:257] = feat.detach()
        YX[:, 257:260] = x.detach()
        e = eng.empty
        M1, M2, M3 = e(P, 1024), e(P, 1024), e(P, 1024)
        eng.nt(addr(YX), 288, addr(eng.WpM0), 288, P, 1024, 288, addr(M1), 1024, EPI_BIAS_RELU, bias=addr(eng.bM0),
               mask=eng.relu_mask(M1, P, 1024))
        for j, (src, dst) in ((1, (M1, M2)), (2, (M2, M3))):
            eng.nt(addr(src), 1024, addr(eng.WpM[j]), 256, P, 256, 256, addr(dst), 1024, EPI_BIAS_RELU,
                   bias=addr(eng.bM[j]), groups=4, sA=256, sB=65536, sC=256, sBias=256, mask=eng.relu_mask(dst, P, 1024))
        Mraw = e(P, 8)
        eng.skinny_fwd(addr(M3), 1024, P, 1024, addr(eng.Ws6), 1024, addr(eng.b6), 6, addr(Mraw), 8)
        ctx.eng, ctx.s, ctx.names = eng, dict(YX=YX, M1=M1, M2=M2, M3=M3), names
        ctx.set_materialize_grads(False)
        return Mraw[:, :6].clone()

    @staticmethod
    def backward(ctx, dM):
        from .engine import EPI_MUL_DRELU, EPI_PLAIN
        eng, s = ctx.eng, ctx.s
        eng.op_begin()
        P = s['YX'].shape[0]
        e = eng.empty
        flat = eng.zeros(eng.n_grad)
        dMraw = eng.zeros(P, 8)
        if dM is not None:
            dMraw[:, :6] = dM
        db0, db12, db6 = eng.mat_db
        with eng.wgrad_batch() as wb:                 # the op's three weight gradients: one queue, launched together
            dM3 = wb.keep(e(P, 1024))
            eng.skinny_bwd(addr(dMraw), 8, addr(s['M3']), 1024, P, 1024, addr(eng.Ws6), 1024, 6, addr(dM3), 1024, 1, 0,
                           addr(eng.dWs6), 1024, addr(flat, db6))
            dA = dM3
            for j, Hin in ((2, s['M2']), (1, s['M1'])):
                eng.wgrad(addr(dA), 1024, addr(Hin), 1024, P, 256, 256, addr(eng.dWpM[j]), 256, addr(flat, db12[j]),
                          groups=4, sA0=256, sB0=256, sW=65536, sDb=256)
                nxt = wb.keep(e(P, 1024))
                eng.nt(addr(dA), 1024, addr(eng.WpTM[j]), 256, P, 256, 256, addr(nxt), 1024, EPI_MUL_DRELU,
                       H=addr(Hin), ldh=1024, groups=4, sA=256, sB=65536, sC=256, sH=256, mask=getattr(Hin, '_nu_mask', None))
                dA = nxt
            eng.wgrad(addr(dA), 1024, addr(s['YX']), 288, P, 1024, 288, addr(eng.dWpM0), 288, addr(flat, db0))
            dYX = e(P, 288)
            eng.nt(addr(dA), 1024, addr(eng.WpTM0), 1024, P, 288, 1024, addr(dYX), 288, EPI_PLAIN)
        eng.unpack_grads(flat, eng.mat_layers)
        eng.op_end()
        ctx.s = None
        return None, dYX[:, 1:257], dYX[:, 257:260], None, _token_grad(eng, flat, ctx.names)


class IorFn(torch.autograd.Function):
    """The stage-2 IoR / thickness networks on the HIP GEMMs (field.py:1046-1087; `ls` = the engine's layer list): X [rows, 39] (the 6-frequency embedding of the hit point)
    -> 256 ReLU -> 256 ReLU -> 256 (no activation) -> 1 raw output (the caller applies the sigmoid); gradients w.r.t. X and the
    weight-normed parameters."""

    @staticmethod
    def forward(ctx, eng, ls, X, names, token):
        from .engine import EPI_BIAS_NONE, EPI_BIAS_RELU
        rows, K = X.shape
        Xp = eng.zeros(rows, 64)
        Xp[:, :K] = X.detach()
        H = [eng.empty(rows, 256) for _ in range(3)]
        eng.nt(addr(Xp), 64, addr(*ls[0].Wp), 64, rows, 256, 64, addr(H[0]), 256, EPI_BIAS_RELU, bias=addr(ls[0].b))
        eng.nt(addr(H[0]), 256, addr(*ls[1].Wp), 256, rows, 256, 256, addr(H[1]), 256, EPI_BIAS_RELU, bias=addr(ls[1].b))
        eng.nt(addr(H[1]), 256, addr(*ls[2].Wp), 256, rows, 256, 256, addr(H[2]), 256, EPI_BIAS_NONE, bias=addr(ls[2].b))
        out = eng.empty(rows, 1)
        eng.skinny_fwd(addr(H[2]), 256, rows, 256, addr(*ls[3].Wp), 256, addr(ls[3].b), 1, addr(out), 1)
        ctx.eng, ctx.ls, ctx.names, ctx.Xp, ctx.H, ctx.K = eng, ls, names, Xp, H, K
        ctx.set_materialize_grads(False)
        return out[:, 0].clone()

    @staticmethod
    def backward(ctx, dout):
        from .engine import EPI_MUL_DRELU, EPI_PLAIN
        eng, ls, Xp, H = ctx.eng, ctx.ls, ctx.Xp, ctx.H
        eng.op_begin()
        rows = Xp.shape[0]
        flat = eng.zeros(eng.n_grad)
        dy = (dout.contiguous() if dout is not None else eng.zeros(rows)).reshape(rows, 1).contiguous()
        with eng.wgrad_batch():               # the op's weight gradients: one queue (every operand is a local that outlives the block)
            d2 = eng.empty(rows, 256)
            eng.skinny_bwd(addr(dy), 1, addr(H[2]), 256, rows, 256, addr(*ls[3].Wp), 256, 1, addr(d2), 256, 0, 0,
                           addr(*ls[3].dWp), ls[3].ldd, addr(flat, ls[3].db_off))
            eng.wgrad(addr(d2), 256, addr(H[1]), 256, rows, 256, 256, addr(*ls[2].dWp), ls[2].ldd, addr(flat, ls[2].db_off))
            d1 = eng.empty(rows, 256)
            eng.nt(addr(d2), 256, addr(*ls[2].WpT), ls[2].ldT, rows, 256, 256, addr(d1), 256, EPI_MUL_DRELU, H=addr(H[1]), ldh=256)
            eng.wgrad(addr(d1), 256, addr(H[0]), 256, rows, 256, 256, addr(*ls[1].dWp), ls[1].ldd, addr(flat, ls[1].db_off))
            d0 = eng.empty(rows, 256)
            eng.nt(addr(d1), 256, addr(*ls[1].WpT), ls[1].ldT, rows, 256, 256, addr(d0), 256, EPI_MUL_DRELU, H=addr(H[0]), ldh=256)
            eng.wgrad(addr(d0), 256, addr(Xp), 64, rows, 256, 64, addr(*ls[0].dWp), ls[0].ldd, addr(flat, ls[0].db_off))
            dX = eng.empty(rows, 64)
            eng.nt(addr(d0), 256, addr(*ls[0].WpT), ls[0].ldT, rows, 64, 256, addr(dX), 64, EPI_PLAIN)
        eng.unpack_grads(flat, ls)
        eng.op_end()
        ctx.H = ctx.Xp = None
        return None, None, dX[:, :ctx.K], None, _token_grad(eng, flat, ctx.names)


class IorPairFn(torch.autograd.Function):
    """Two networks of IorFn's shape on the SAME input (the non-zero-thickness model evaluates IoR and thickness at every hit
    point, renderer.py:1725-1734) as grouped launches: each GEMM of the pair is one launch of two independent problems at
    constant strides (the pair's weight tables, biases, activations), so the pair costs the launches of one network -- at the
    10^2-10^3 hit points of a stage-2 bounce these launches are latency-bound.  Same kernels, same per-row arithmetic as two
    IorFn calls."""

    @staticmethod
    def forward(ctx, eng, la, lb, X, names, token):
        from .engine import EPI_BIAS_NONE, EPI_BIAS_RELU
        rows, K = X.shape
        Xp = eng.zeros(rows, 64)
        Xp[:, :K] = X.detach()
        H = [eng.empty(2, rows, 256) for _ in range(3)]
        sH = rows * 256

        def st(pa, pb):             # element stride between the two networks' copies of one table (separate allocations)
            return (pb - pa) // 4
        for k, (src, lds, Kd, sA, epi) in enumerate(((Xp, 64, 64, 0, EPI_BIAS_RELU), (H[0], 256, 256, sH, EPI_BIAS_RELU),
                                                      (H[1], 256, 256, sH, EPI_BIAS_NONE))):
            eng.nt(addr(src), lds, addr(*la[k].Wp), Kd, rows, 256, Kd, addr(H[k]), 256, epi, bias=addr(la[k].b), groups=2, sA=sA,
                   sB=st(addr(*la[k].Wp), addr(*lb[k].Wp)), sC=sH, sBias=st(addr(la[k].b), addr(lb[k].b)))
        out = eng.empty(2, rows, 1)
        for z, ls in enumerate((la, lb)):
            eng.skinny_fwd(addr(H[2], z * sH), 256, rows, 256, addr(*ls[3].Wp), 256, addr(ls[3].b), 1, addr(out, z * rows), 1)
        ctx.eng, ctx.la, ctx.lb, ctx.names, ctx.Xp, ctx.H, ctx.K = eng, la, lb, names, Xp, H, K
        ctx.set_materialize_grads(False)
        return out[0, :, 0].clone(), out[1, :, 0].clone()

    @staticmethod
    def backward(ctx, da, db):
        from .engine import EPI_MUL_DRELU, EPI_PLAIN
        eng, la, lb, Xp, H = ctx.eng, ctx.la, ctx.lb, ctx.Xp, ctx.H
        eng.op_begin()
        rows = Xp.shape[0]
        sH = rows * 256
        flat = eng.zeros(eng.n_grad)
        dy = eng.zeros(2, rows, 1)
        if da is not None:
            dy[0, :, 0] = da
        if db is not None:
            dy[1, :, 0] = db

        def st(pa, pb):
            return (pb - pa) // 4
        with eng.wgrad_batch() as wb:         # the op's weight gradients: one queue; the intermediate cotangents are held until it is launched
            d2 = eng.empty(2, rows, 256)
            for z, ls in enumerate((la, lb)):
                eng.skinny_bwd(addr(dy, z * rows), 1, addr(H[2], z * sH), 256, rows, 256, addr(*ls[3].Wp), 256, 1, addr(d2, z * sH), 256, 0, 0,
                               addr(*ls[3].dWp), ls[3].ldd, addr(flat, ls[3].db_off))
            d_prev = d2
            for k in (2, 1):
                eng.wgrad(addr(d_prev), 256, addr(H[k - 1]), 256, rows, 256, 256, addr(*la[k].dWp), la[k].ldd, addr(flat, la[k].db_off),
                          groups=2, sA0=sH, sB0=sH, sW=st(addr(*la[k].dWp), addr(*lb[k].dWp)), sDb=lb[k].db_off - la[k].db_off)
                d_next = wb.keep(eng.empty(2, rows, 256))
                eng.nt(addr(d_prev), 256, addr(*la[k].WpT), la[k].ldT, rows, 256, 256, addr(d_next), 256, EPI_MUL_DRELU, H=addr(H[k - 1]), ldh=256,
                       groups=2, sA=sH, sB=st(addr(*la[k].WpT), addr(*lb[k].WpT)), sC=sH, sH=sH)
                d_prev = d_next
            eng.wgrad(addr(d_prev), 256, addr(Xp), 64, rows, 256, 64, addr(*la[0].dWp), la[0].ldd, addr(flat, la[0].db_off),
                      groups=2, sA0=sH, sB0=0, sW=st(addr(*la[0].dWp), addr(*lb[0].dWp)), sDb=lb[0].db_off - la[0].db_off)
            dX = eng.empty(2, rows, 64)
            eng.nt(addr(d_prev), 256, addr(*la[0].WpT), la[0].ldT, rows, 64, 256, addr(dX), 64, EPI_PLAIN,
                   groups=2, sA=sH, sB=st(addr(*la[0].WpT), addr(*lb[0].WpT)), sC=rows * 64)
        eng.unpack_grads(flat, la)
        eng.unpack_grads(flat, lb)
        eng.op_end()
        ctx.H = ctx.Xp = None
        return None, None, None, (dX[0] + dX[1])[:, :ctx.K], None, _token_grad(eng, flat, ctx.names)


def _same_shape_pair(la, lb):
    return len(la) == len(lb) == 4 and all(a.N == b.N and a.K == b.K and a.Kp == b.Kp and a.ldT == b.ldT and a.ldd == b.ldd
                                           for a, b in zip(la, lb))


class Stage1Nets:
    """Differentiable callables over one Stage1Engine (SDF, variance, NeRF++, the shading predictors)."""

    def __init__(self, eng, named, prefix_map=None):
        if getattr(eng, 'h16', False):
            raise NotImplementedError("stage 2 runs the fp32 / bf16x6 MLP modes; mlp_dtype 'bf16' (bf16 storage) is a stage-1 mode")
        self.eng = eng
        g = list(eng.grad_views.keys())
        self.named = named

        def sel(pred):
            names = [n for n in g if pred(n)]
            return names, [named[n] for n in names]
        self.sdf_names, self.sdf_params = sel(lambda n: n.startswith('sdf_network.'))
        self.nerf_names, self.nerf_params = sel(lambda n: n.startswith('outer_nerf.'))
        mats = ('metallic_predictor', 'roughness_predictor', 'albedo_predictor', 'transmisstion_weight')
        self.mat_names, self.mat_params = sel(lambda n: n.startswith(tuple('color_network.' + m + '.' for m in mats)))
        self.stack = {}
        for nm, layers in (('outer_light', eng.outer_light), ('inner_light', eng.inner_light),
                           ('inner_weight', eng.inner_weight), ('refrac_light', eng.refrac_light)):
            names, params = sel(lambda n, nm=nm: n.startswith('color_network.' + nm + '.'))
            self.stack[nm] = (layers, names, params)
        self.ior_names, _ = sel(lambda n: n.startswith('ior_network.'))
        self.thick_names, _ = sel(lambda n: n.startswith('thickness_network.'))
        self.all_names = [n for n in g if n in named and isinstance(named[n], torch.nn.Parameter)]
        for names in [self.sdf_names, self.nerf_names, self.mat_names, self.ior_names, self.thick_names] + [v[1] for v in self.stack.values()]:
            if names:
                _own_range(eng, names)       # every op's parameters form ONE range of the flat buffer (asserted once, here)
        self._token = None

    def begin_pass(self):
        """Start of a forward pass (a new autograd graph): the next network op makes a fresh parameter hub."""
        self._token = None

    def token(self):
        """The pass's parameter-hub token (ParamHubFn): what every network op differentiates instead of its parameters."""
        if self._token is None or not torch.is_grad_enabled():
            tok = ParamHubFn.apply(self.eng, self.all_names, *[self.named[n] for n in self.all_names])
            if not torch.is_grad_enabled():
                return tok
            self._token = tok
        return self._token

    def sdf(self, x, need_normal=True):
        """(sdf | feature [P,257], normal [P,3]); need_normal=False returns (.., None) and skips the normal's reverse sweep."""
        return SdfFn.apply(self.eng, x, self.sdf_names, self.token(), need_normal)

    def nerf(self, x, d):
        return NerfFn.apply(self.eng, x, d, self.nerf_names, self.token())

    def materials(self, feat, x):
        return MaterialsFn.apply(self.eng, feat, x, self.mat_names, self.token())

    def ior(self, X):
        """Raw (pre-sigmoid) output of the IoR network on encoded points X [rows, 39]."""
        if X.shape[0] == 0:
            return X.new_zeros(0)
        return IorFn.apply(self.eng, self.eng.small['ior_network'], X, self.ior_names, self.token())

    def thickness(self, X):
        """Raw (pre-sigmoid) output of the thickness network (field.py:1068-1087) on encoded points X [rows, 39]."""
        if X.shape[0] == 0:
            return X.new_zeros(0)
        return IorFn.apply(self.eng, self.eng.small['thickness_network'], X, self.thick_names, self.token())

    def ior_and_thickness(self, X):
        """(raw IoR, raw thickness) on encoded points X [rows, 39]: the two networks as grouped launches (IorPairFn)."""
        if X.shape[0] == 0:
            return X.new_zeros(0), X.new_zeros(0)
        la, lb = self.eng.small['ior_network'], self.eng.small['thickness_network']
        if not _same_shape_pair(la, lb) or os.environ.get('NU_S2_IOR_PAIR') == '0':
            return self.ior(X), self.thickness(X)
        return IorPairFn.apply(self.eng, la, lb, X, self.ior_names + self.thick_names, self.token())

    def predictor(self, name, X):
        layers, names, params = self.stack[name]
        return StackFn.apply(self.eng, layers, X, names, self.token())

    def predictors(self, which, Xs):
        """Raw heads of several predictor stacks of ONE shading call (`which` in the engine's table order, e.g. ('outer_light',
        'inner_light', 'inner_weight')) as one op: level j of all stacks is one launch (StacksFn)."""
        key = tuple(which)
        u = self.__dict__.setdefault('_unions', {}).get(key)
        if u is None:
            layers_union = [lay for nm in which for lay in self.stack[nm][0]]
            names_union = [n for nm in which for n in self.stack[nm][1]]
            _own_range(self.eng, names_union)                 # one contiguous range of the flat buffer (asserted)
            u = self._unions[key] = (tuple(self.stack[nm][0] for nm in which), layers_union, names_union)
        if os.environ.get('NU_S2_STACKS') == '0' or any(X.shape[0] == 0 for X in Xs):      # development switch (A/B): one op per stack
            return tuple(self.predictor(nm, X) for nm, X in zip(which, Xs))
        return StacksFn.apply(self.eng, u[0], u[1], u[2], self.token(), *Xs)
