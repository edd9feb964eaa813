"""Host-side orchestration of the stage-1 hot path over the C ABI of libnunerf.so.

This is the Python counterpart of the reference's Python host code (network/renderer_zerothick.py,
network/field.py): it owns the packed-weight tables and sequences the HIP kernels -- fp32-MFMA GEMM
sweeps for the three MLP stacks (forward, SDF input-gradient, backward and the second-order sweeps),
encoders, NeuS alpha, shading combine and the composite.  No arithmetic of the path happens in torch:
torch only provides device memory and the current stream.

Buffer conventions (all fp32 row-major; leading dims are multiples of 32 with zero/finite padding):
  SDF net (P inner points)                                   reference: field.py:133-170
    E  [P, 64]   embedding (39) | 0                U4 [P,256]  h4 (217) | embedding (39)
    H1..H3, H5..H8 [P,256] post-softplus           YX [P,288]  sdf | feat (256) | x (3) | 0
    D0..D7 [P,256] delta_l = gbar_{l+1} * sp'(a_l)  (reverse sweep producing n = d sdf / d x)
    Q*, C*  tangent sweep / second-order terms (Appendix B of SURVEY.md)
  NeRF++ (P outer points)                                    reference: field.py:265-289
    E4 [P,96], N1..N4,N6..N8 [P,256], U5 [P,352] = h5 | emb(84) | 0, V [P,288] = feature | view(27) | 0
  shading stack                                              reference: field.py:684-777
    materials: YX -> M1 [P,1024] -> M2 -> M3 -> Mraw [P,8]  (4 predictors batched / grouped)
    lights: OLin [3P+R, 96], ILin [2P,128], IWin [P,96], RLin [P,96] -> 3 hidden [rows,256] -> raw heads
"""
import ctypes
import os
import math

import numpy as np
import torch

from . import _lib as L

c_int, c_ll, c_f, c_p = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p

EPI_BIAS_NONE, EPI_BIAS_RELU, EPI_BIAS_SOFTPLUS, EPI_MUL_DRELU, EPI_MUL_DSP, EPI_Q_SP, EPI_B_SP, EPI_PLAIN, EPI_B_RELU = range(9)


class GemmNT(ctypes.Structure):
    _fields_ = [("A", c_p), ("lda", c_int), ("B", c_p), ("ldb", c_int), ("M", c_int), ("N", c_int), ("K", c_int),
                ("C", c_p), ("ldc", c_int), ("C2", c_p), ("ldc2", c_int), ("bias", c_p), ("H", c_p), ("ldh", c_int),
                ("D", c_p), ("ldd", c_int), ("Cadd", c_p), ("ldadd", c_int), ("zero_to", c_int), ("act_cols", c_int),
                ("alpha", c_f), ("groups", c_int), ("sA", c_ll), ("sB", c_ll), ("sC", c_ll), ("sC2", c_ll),
                ("sBias", c_ll), ("sH", c_ll), ("sD", c_ll), ("sCadd", c_ll), ("epi", c_int), ("bf16", c_int),
                ("mask", c_p), ("mask_nct", c_int), ("mask_ct0", c_int), ("B6", c_p)]


class GemmTN(ctypes.Structure):
    _fields_ = [("A0", c_p), ("lda0", c_int), ("B0", c_p), ("ldb0", c_int), ("A1", c_p), ("lda1", c_int),
                ("B1", c_p), ("ldb1", c_int), ("P", c_int), ("N1", c_int), ("N2", c_int), ("slab", c_p),
                ("bias_slab", c_p), ("S", c_int), ("groups", c_int), ("sA0", c_ll), ("sB0", c_ll), ("sA1", c_ll),
                ("sB1", c_ll), ("sSlab", c_ll), ("sBiasSlab", c_ll), ("bf16", c_int), ("pad_", c_int)]


class ReduceDesc(ctypes.Structure):
    """Mirror of NuReduceDesc (include/nu_nerf.h): one deferred deterministic split reduction."""
    _fields_ = [("slab", c_p), ("out", c_p), ("ss", c_ll), ("S", c_int), ("N1", c_int), ("N2", c_int), ("rs", c_int),
                ("ldo", c_int), ("accumulate", c_int), ("G", c_int), ("blk_begin", c_int), ("alpha", ctypes.c_float),
                ("pad_", c_int)]


class PackDesc(ctypes.Structure):
    _fields_ = [("v", c_p), ("g", c_p), ("colmap", c_p), ("Wp", c_p), ("WpT", c_p), ("dWp", c_p), ("dv_off", c_ll),
                ("dg_off", c_ll), ("bias", c_p), ("bias_p", c_p), ("scale", c_f), ("N", c_int), ("K", c_int),
                ("Kp", c_int), ("ldT", c_int), ("ldd", c_int), ("row_begin", c_int), ("col_off", c_int), ("Wp16", c_p), ("WpT16", c_p),
                ("planes", c_int), ("pad_", c_int), ("w6_row0", c_int), ("w6_ld", c_int), ("t6_row0", c_int), ("t6_col0", c_int),
                ("t6_ld", c_int), ("pad2_", c_int)]


class Lin(ctypes.Structure):
    """Mirror of NuLin (include/nu_nerf.h): one packed layer."""
    _fields_ = [("Wp", c_p), ("WpT", c_p), ("dWp", c_p), ("bias", c_p), ("db_off", c_ll), ("N", c_int), ("K", c_int), ("Kp", c_int),
                ("ldT", c_int), ("ldd", c_int), ("pad_", c_int), ("Wp16", c_p), ("WpT16", c_p)]


class WgradItem(ctypes.Structure):
    """Mirror of NuWgradItem: one queued weight gradient of a backward pass."""
    _fields_ = [("g", GemmTN), ("dW", c_p), ("ldw", c_int), ("pad_", c_int), ("sW", c_ll), ("db", c_p), ("sDb", c_ll),
                ("flops", ctypes.c_double), ("bytes", ctypes.c_double)]


class OpCtx(ctypes.Structure):
    """Mirror of NuOpCtx: arithmetic mode, flat gradient buffer, deferred-reduction arena (shared by the Python-sequenced path),
    the queue of a pass's weight gradients."""
    _fields_ = [("prec", c_int), ("h16", c_int), ("flat", c_p), ("arena", c_p), ("arena_floats", c_ll), ("arena_off", c_ll),
                ("descs", c_p), ("ndesc", c_int), ("cap", c_int), ("ev", c_p), ("ev_meta", c_p), ("nev", c_int), ("ev_cap", c_int),
                ("forked", c_int), ("pad_", c_int), ("pend", c_p), ("npend", c_int), ("pend_cap", c_int)]


class SdfNet(ctypes.Structure):
    _fields_ = [("lin", Lin * 9)]


class SdfBufs(ctypes.Structure):
    _fields_ = [("P", c_int), ("pad_", c_int), ("E", c_p), ("U4", c_p), ("YX", c_p), ("sdf", c_p), ("H", c_p * 9), ("D", c_p * 8),
                ("G0", c_p), ("n", c_p), ("Q", c_p * 9), ("C", c_p * 8), ("Aux", c_p * 8), ("dE0", c_p)]


class NerfNet(ctypes.Structure):
    _fields_ = [("pts", Lin * 8), ("feat", Lin), ("alpha", Lin), ("view", Lin), ("rgb", Lin)]


class NerfBufs(ctypes.Structure):
    _fields_ = [("P", c_int), ("pad_", c_int), ("H", c_p * 9), ("mask", c_p * 9), ("V", c_p), ("HV", c_p), ("sig", c_p), ("rgb", c_p),
                ("dHV", c_p), ("dF", c_p), ("dH8a", c_p), ("dA", c_p * 9), ("dE4", c_p), ("dx", c_p), ("ddir", c_p)]


class ShadeNet(ctypes.Structure):
    _fields_ = [("WpM0", c_p), ("WpTM0", c_p), ("bM0", c_p), ("dWpM0", c_p), ("WpM", c_p * 3), ("WpTM", c_p * 3), ("bM", c_p * 3),
                ("dWpM", c_p * 3), ("dbM_off", c_ll * 3), ("Ws6", c_p), ("b6", c_p), ("dWs6", c_p), ("db6_off", c_ll),
                ("outer_light", Lin * 4), ("inner_light", Lin * 4), ("inner_weight", Lin * 4), ("refrac_light", Lin * 4),
                ("lut", c_p), ("exp_max", c_f), ("sphere", c_int), ("ld_ol", c_int), ("refrac_dim", c_int), ("ld_rl", c_int), ("pad_", c_int),
                ("WpM0_16", c_p), ("WpTM0_16", c_p), ("WpM16", c_p * 3), ("WpTM16", c_p * 3)]


class ShadeBufs(ctypes.Structure):
    _fields_ = [("P", c_int), ("R", c_int), ("extra_dirs", c_p), ("extra_pts", c_p), ("M", c_p * 3), ("maskM", c_p * 3), ("Mraw", c_p),
                ("OLin", c_p), ("ILin", c_p), ("IWin", c_p), ("RLin", c_p), ("SD", c_p),
                ("OLh", c_p * 3), ("ILh", c_p * 3), ("IWh", c_p * 3), ("RLh", c_p * 3),
                ("maskOL", c_p * 3), ("maskIL", c_p * 3), ("maskIW", c_p * 3), ("maskRL", c_p * 3),
                ("OLo", c_p), ("ILo", c_p), ("IWo", c_p), ("RLo", c_p), ("aux", c_p),
                ("dMraw", c_p), ("dOLo", c_p), ("dILo", c_p), ("dIWo", c_p), ("dRLo", c_p), ("dNoV", c_p),
                ("dH3", c_p * 4), ("tmpOL", c_p * 2), ("tmpIL", c_p * 2), ("tmpIW", c_p * 2), ("tmpRL", c_p * 2),
                ("dOLin", c_p), ("dILin", c_p), ("dn", c_p), ("dM", c_p * 3), ("dYX", c_p)]


def rup(a, b):
    return (a + b - 1) // b * b


def addr(t, off=0):
    """Device address of element `off` of tensor t (fp32 / int32, or bf16 in the bf16-storage mode); 0 for None."""
    if t is None:
        return 0
    return t.data_ptr() + t.element_size() * off


class _Layer:
    """One packed linear layer (python record; the device table is built from these)."""

    def __init__(self, name, v, g, b, N, K, Kp, *, scale=1.0, colmap=None, col_off=0, v_row0=0):
        self.name, self.v, self.g, self.b = name, v, g, b
        self.N, self.K, self.Kp, self.scale, self.colmap, self.col_off, self.v_row0 = N, K, Kp, scale, colmap, col_off, v_row0
        self.Wp = self.WpT = self.dWp = self.bias_p = None  # (tensor, element offset)
        self.ldT = self.ldd = 0
        self.dv_off = self.dg_off = self.db_off = -1


class Stage1Engine:
    """Packed weights + kernel sequencing for NeROShapeRenderer's hot path on one GPU."""

    def __init__(self, params, device, cfg):
        self.lib = L.load()
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise L.NuNerfLibraryError("Stage1Engine needs a CUDA(HIP) device: there is no CPU fallback")
        self._dev_index = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        self.cfg = cfg
        self.p = params  # dict name -> Parameter/Tensor on device
        self.exp_max = float(cfg.get('light_exp_max', 3.0))
        self.sphere_direction = bool(cfg.get('sphere_direction', False))
        # 6 everywhere in stage 1; AppShadingNetwork_SpecInner (the inner surface of the non-zero-thickness stage-2 model,
        # field.py:1321-1330) defaults to 8 -- supported by the network-level ops (nets.py), not by the fused stage-1 shading
        self.light_pos_freq = int(cfg.get('light_pos_freq', 6))
        if 3 + 6 * self.light_pos_freq + 72 > 128:
            raise NotImplementedError("light_pos_freq > 8: the inner_light input would not fit its 128-column tile")
        self.refrac_dim = 3 + 6 * int(cfg.get('refrac_freq', 6))       # field.py:590-591 (real_bottle uses refrac_freq 3)
        self.ld_ol = 160 if self.sphere_direction else 96               # outer_light input 144 / 72 (field.py:594-597)
        self.ld_rl = rup(2 * self.refrac_dim, 32)
        lib = self.lib
        for fn in ("nu_wgrad_workspace_bytes", "nu_skinny_bwd_workspace_bytes", "nu_colsum_workspace_bytes",
                   "nu_gemm_tn_workspace_bytes", "nu_neus_alpha_bwd_workspace_bytes"):
            getattr(lib, fn).restype = c_ll
        assert lib.nu_pack_desc_size() == ctypes.sizeof(PackDesc), "PackDesc ABI mismatch"
        assert lib.nu_reduce_desc_size() == ctypes.sizeof(ReduceDesc), "ReduceDesc ABI mismatch"
        assert lib.nu_gemm_nt_size() == ctypes.sizeof(GemmNT) and lib.nu_gemm_tn_size() == ctypes.sizeof(GemmTN), "GEMM ABI mismatch"
        # MLP arithmetic: 'fp32' = exact fp32 MFMA (the reference's precision); 'bf16' = operands rounded to bf16 on their
        # way into LDS, bf16 MFMA with fp32 accumulation (BASELINE config 4; no reference counterpart, tolerance in the tests)
        md = str(cfg.get('mlp_dtype', os.environ.get('NU_MLP_DTYPE', 'fp32'))).lower()   # env: run a whole test suite in one mode
        if md not in ('fp32', 'f32', 'float32', 'bf16', 'bfloat16', 'bf16x6'):
            raise ValueError(f"mlp_dtype {md!r}: expected 'fp32', 'bf16' or 'bf16x6'")
        # 'bf16x6': fp32-equivalent products on the bf16 pipe (exact 3-way split of both operands, six partial products)
        self.bf16 = 2 if md == 'bf16x6' else (1 if md.startswith('b') else 0)
        # 'bf16' stores what only GEMMs touch as bf16 in HBM (weight tables + most hidden activations, include/nu_nerf.h
        # NuOpCtx.h16); NU_BF16_STORAGE=0 keeps the round-1 behaviour (fp32 in HBM, rounded on load) for A/B runs
        self.h16 = self.bf16 == 1 and os.environ.get('NU_BF16_STORAGE', '1') != '0'
        self.hdt = torch.bfloat16 if self.h16 else torch.float32
        # deferred split reductions (weight gradients, skinny heads, column sums): partial slabs live in a bump arena
        # until flush_reductions() sums them all in a few batched launches (before unpack_grads reads the results)
        self._rd_cap = 1024
        self._rd = (ReduceDesc * self._rd_cap)()
        self._arena = None
        # one context for both sequencing paths (network-level C entries and the launch-by-launch Python path below): the
        # descriptor count and the arena offset live in the struct
        # the weight gradients of a backward pass are queued and launched together (include/nu_nerf.h: nu_wgrad_defer / _flush)
        assert lib.nu_wgrad_item_size() == ctypes.sizeof(WgradItem), "WgradItem ABI mismatch"
        self._pend = (WgradItem * 32)()
        self._ctx = OpCtx(prec=self.bf16, h16=1 if self.h16 else 0, flat=0, arena=0, arena_floats=0, arena_off=0,
                          descs=ctypes.cast(self._rd, c_p).value, ndesc=0, cap=self._rd_cap, ev=0, ev_meta=0, nev=0, ev_cap=0,
                          forked=0, pad_=0, pend=ctypes.cast(self._pend, c_p).value, npend=0, pend_cap=32)
        self._wg_depth, self._wg_held = 0, []
        self._ndesc_p = ctypes.cast(ctypes.addressof(self._ctx) + OpCtx.ndesc.offset, ctypes.POINTER(c_int))
        for fn, st in (("nu_op_ctx_size", OpCtx), ("nu_sdf_net_size", SdfNet), ("nu_sdf_bufs_size", SdfBufs), ("nu_nerf_net_size", NerfNet),
                       ("nu_nerf_bufs_size", NerfBufs), ("nu_shade_net_size", ShadeNet), ("nu_shade_bufs_size", ShadeBufs)):
            assert getattr(lib, fn)() == ctypes.sizeof(st), f"{st.__name__} ABI mismatch"
        # NU_PY_SEQ=1: sequence every launch from Python (the path bench.py's per-launch event timing uses)
        self.py_seq = os.environ.get('NU_PY_SEQ', '0') != '0'
        if self.py_seq and self.h16:
            raise ValueError("NU_PY_SEQ=1 (launch-by-launch sequencing) has no bf16-storage mode: unset it or set NU_BF16_STORAGE=0")
        # NU_FUSED_SDF=0: the no-gradient SDF evaluations through the layered path (development A/B)
        self._fused_sdf = os.environ.get('NU_FUSED_SDF', '1') != '0'
        self._ws = None
        self._ptr_sig = None
        self._ktime = None
        self._ktime_on = True
        self._ev_pool = []
        self._cap_classes = []
        self.last_ctx = None
        self._build_layers()

    # ------------------------------------------------------------------ buffers
    # Point counts change a little from step to step; the caching allocator only reuses a block that is large enough, so
    # every new record size used to cost a burst of hipMalloc calls (one of them now and then 50 ms).  Large per-point
    # buffers are therefore allocated from a few CAPACITY CLASSES and handed out as a prefix view: a request reuses an
    # existing class up to 1.6x its size, otherwise a new class with 25 % headroom (multiple of 16384 rows) is opened --
    # the inner / outer point counts of a scene move by +-10 % between batches.  After the first step every allocation
    # repeats a size the allocator already holds (sized for 288 GB: the step's 11 GB of activations become ~15 GB).
    _ROW_QUANTUM = 16384

    def _capacity(self, shape):
        if not (len(shape) >= 1 and isinstance(shape[0], int) and shape[0] > 4 * self._ROW_QUANTUM):
            return None
        n = shape[0]
        for c in self._cap_classes:
            if n <= c <= 1.6 * n:
                return c
        q = self._ROW_QUANTUM
        c = (int(n * 1.25) + q - 1) // q * q
        self._cap_classes.append(c)
        self._cap_classes.sort()
        return c

    def zeros(self, *shape, dtype=torch.float32):
        cap = self._capacity(shape)
        if cap is None:
            return torch.zeros(*shape, dtype=dtype, device=self.dev)
        return torch.zeros(cap, *shape[1:], dtype=dtype, device=self.dev)[:shape[0]]

    def empty(self, *shape, dtype=torch.float32):
        cap = self._capacity(shape)
        if cap is None:
            return torch.empty(*shape, dtype=dtype, device=self.dev)
        return torch.empty(cap, *shape[1:], dtype=dtype, device=self.dev)[:shape[0]]

    def empty_h(self, *shape):
        """A hidden activation that only GEMMs touch: bf16 in the bf16-storage mode (rule: include/nu_nerf.h NuOpCtx)."""
        return self.empty(*shape, dtype=self.hdt)

    def workspace(self, nbytes):
        n = (int(nbytes) + 3) // 4
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=self.dev)
        return self._ws

    def stream(self):
        return L.stream(self._dev_index)

    # Small batches leave the chip half empty (a 256 -> 256 layer on 8 192 points is 128 tiles for 256 CUs): the NeRF++ chain of the
    # outer points and the SDF / shading chain of the inner points are independent between the partition and the composite, so
    # below `_TWO_STREAM_SAMPLES` ray samples the NeRF++ chain runs on a second HIP stream.  Both chains push their split reductions
    # to the shared arena from this (single) host thread; the batched reduction runs after the join.
    _TWO_STREAM_SAMPLES = int(os.environ.get('NU_TWO_STREAM_SAMPLES', 200000))
    occ_sdf_thresh = None      # set by the renderer for a training step past occ_loss_step: render_forward leaves ctx['occ_idx']
    _FUSED_SDF_MAX_POINTS = int(os.environ.get('NU_FUSED_SDF_MAX_POINTS', 40000))

    def _fork(self, mark=True):
        """mark=False (stage 2): the ops of this engine that run on the two streams are totally ordered by events (op_begin /
        op_end), so a mid-pass flush of its arena is safe and NuOpCtx.forked stays clear -- nothing to restore if the window is
        left by an exception."""
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(self.dev)
        self._side.wait_stream(torch.cuda.current_stream(self.dev))
        if mark:
            self._ctx.forked = 1      # until _join: no mid-pass flush of the shared arena (fail closed, NuOpCtx.forked)
        return self._side

    def _join(self):
        torch.cuda.current_stream(self.dev).wait_stream(self._side)
        self._ctx.forked = 0

    def forked(self):
        """`with eng.forked() as side:` -- the fork / join pair as a context: whatever the block raises (a full arena fails closed
        with NuNerfLibraryError while two streams share it), the caller's stream waits for the side stream again and NuOpCtx.forked
        is cleared, so a later single-stream pass on this engine may flush mid-pass as usual."""
        eng = self

        class _Forked:
            def __enter__(self):
                return eng._fork()

            def __exit__(self, et, ev, tb):
                eng._join()
                return False
        return _Forked()

    # Stage 2 runs the ops of ONE engine on two streams (the inner segment on the side stream, the IoR / thickness networks on the
    # caller's): they share this engine's reduction arena and descriptor list, which a flush resets.  The host issues the ops one
    # after the other, so a total order across the streams is enough: every op that may push reductions waits for the previous such
    # op of the engine when that ran on another stream (an event, not a stream-wide wait: the other engine's work on that stream
    # keeps overlapping).
    def op_begin(self):
        ev = getattr(self, '_op_ev', None)
        if ev is not None:
            cur = torch.cuda.current_stream(self.dev)
            if cur.cuda_stream != self._op_stream:
                cur.wait_event(ev)

    def op_end(self):
        if getattr(self, '_side', None) is None:
            return                                    # this engine never left its caller's stream
        cur = torch.cuda.current_stream(self.dev)
        ev = torch.cuda.Event()
        ev.record(cur)
        self._op_ev, self._op_stream = ev, cur.cuda_stream

    def relu_mask(self, act, rows, ncols):
        """Sign-bit buffer for a [rows, ncols] ReLU activation (2 KB per 128x128 tile): written by the BIAS_RELU GEMM that
        produces `act`, read by the backward GEMMs instead of `act` itself.  Rides on the activation tensor so that it lives
        exactly as long."""
        if os.environ.get('NU_RELU_MASK', '1') == '0':     # development switch: A/B against reading the activation
            return None
        nct = (ncols + 127) // 128
        rows_q = self._capacity((rows,)) or rows
        m = torch.empty(((rows_q + 127) // 128) * nct * 256, dtype=torch.int64, device=self.dev)
        m._nu_nct = nct
        act._nu_mask = m
        return m

    def _arena_take(self, nbytes):
        """Device address of `nbytes` of slab space that stays untouched until the next flush_reductions()."""
        n = (int(nbytes) + 255) // 256 * 64          # floats, 256-byte granules
        if self._arena is None:
            self._ensure_arena(n)
        elif self._ctx.arena_off + n > self._arena.numel():
            self._forced_flush()                      # stream order: later producers may then reuse the space
            if n > self._arena.numel():
                self._ensure_arena(n)
        off = self._ctx.arena_off
        self._ctx.arena_off += n
        return self._arena.data_ptr() + 4 * off, n * 4

    def _ensure_arena(self, n=0):
        if self._arena is None or n > self._arena.numel():
            self._arena = None
            # >= 4 GiB: the split-reduction slabs of one step come to 1.5-2 GB at every batch size (the split count is capped, not
            # the point count), so the arena never fills inside a step and the one batched reduction runs after the last producer
            # -- which is also what makes the two-stream mode safe (an early flush would reduce slabs the other stream still writes)
            self._arena = torch.empty(max(n, int(os.environ.get('NU_ARENA_FLOATS', 1 << 30))), dtype=torch.float32, device=self.dev)
            self._ctx.arena, self._ctx.arena_floats = self._arena.data_ptr(), self._arena.numel()

    def flush_reductions(self):
        L.check(self.lib.nu_ctx_flush(ctypes.byref(self._ctx), self.stream()), "nu_ctx_flush")

    def _forced_flush(self):
        """A flush forced by a full arena / descriptor table in the MIDDLE of a pass.  While the engine is forked (two streams feed
        the one arena) that would reduce slabs the other stream may still be writing and then hand their space out again:
        fail closed, like the C entries (NuOpCtx.forked -> NU_ERR_WORKSPACE)."""
        if self._ctx.forked:
            raise L.NuNerfLibraryError("split-reduction arena (or descriptor table) full while two streams share it: raise "
                                       "NU_ARENA_FLOATS or lower NU_TWO_STREAM_SAMPLES (NU_ERR_WORKSPACE)")
        # reductions only: the weight gradients still queued keep waiting for the end of their pass (their split must not
        # depend on when the arena happened to fill)
        L.check(self.lib.nu_ctx_reduce(ctypes.byref(self._ctx), self.stream()), "nu_ctx_reduce")

    # ------------------------------------------------------------------ layer tables
    def _build_layers(self):
        p = self.p
        z = self.zeros
        layers = []
        self.grad_views = {}   # param name -> (offset, shape)
        self._goff = 0

        def galloc(n):
            o = self._goff
            self._goff += n
            return o

        self.grad_numel = {}   # param name -> element count (cached: the per-step slicing of the flat buffer is host-time critical)

        def reg(name, off):
            self.grad_views[name] = (off, tuple(p[name].shape))
            self.grad_numel[name] = int(p[name].numel())

        # ---- SDF network ----
        sdf = []
        for l in range(9):
            pre = f'sdf_network.lin{l}'
            v = p[pre + '.weight_v']
            N, K = v.shape
            Kp = 64 if l == 0 else 256
            lay = _Layer(pre, v, p[pre + '.weight_g'], p[pre + '.bias'], N, K, Kp,
                         scale=(1.0 / math.sqrt(2.0)) if l == 4 else 1.0)
            lay.Wp = (z(rup(N, 128), Kp), 0)
            ldT = 288 if l == 8 else rup(N, 32)
            lay.WpT, lay.ldT = (z(rup(Kp, 128), ldT), 0), ldT
            lay.dWp, lay.ldd = (z(N, Kp), 0), Kp
            lay.dv_off, lay.dg_off, lay.db_off = galloc(N * K), galloc(N), galloc(N)
            reg(pre + '.weight_v', lay.dv_off); reg(pre + '.weight_g', lay.dg_off); reg(pre + '.bias', lay.db_off)
            sdf.append(lay)
        self.sdf = sdf
        layers += sdf
        self.var_off = galloc(1)
        reg('deviation_network.variance', self.var_off)

        # ---- NeRF++ ----
        nerf = []
        cm5 = np.concatenate([256 + np.arange(84), np.arange(256)]).astype(np.int32)   # [emb(84), h(256)] -> h | emb
        self._colmaps = {'n5': torch.from_numpy(cm5).to(self.dev)}

        def plain(name, Kp, colmap=None, NT_rows=True):
            w = p[name + '.weight']
            N, K = w.shape
            lay = _Layer(name, w, None, p[name + '.bias'], N, K, Kp, colmap=colmap)
            lay.Wp = (z(rup(N, 128), Kp), 0)
            lay.WpT, lay.ldT = (z(rup(Kp, 128), rup(N, 32)), 0), rup(N, 32)
            lay.dWp, lay.ldd = (z(N, Kp), 0), Kp
            lay.dv_off, lay.db_off = galloc(N * K), galloc(N)
            reg(name + '.weight', lay.dv_off); reg(name + '.bias', lay.db_off)
            return lay
        for i in range(8):
            Kp = 96 if i == 0 else (352 if i == 5 else 256)
            nerf.append(plain(f'outer_nerf.pts_linears.{i}', Kp, self._colmaps['n5'] if i == 5 else None))
        self.nerf = nerf
        self.nerf_feat = plain('outer_nerf.feature_linear', 256)
        self.nerf_alpha = plain('outer_nerf.alpha_linear', 256)
        self.nerf_view = plain('outer_nerf.views_linears.0', 288)
        self.nerf_rgb = plain('outer_nerf.rgb_linear', 128)
        self.nerf_all = nerf + [self.nerf_feat, self.nerf_alpha, self.nerf_view, self.nerf_rgb]     # consecutive in the table
        layers += self.nerf_all

        # ---- shading: material predictors (batched) ----
        mats = ['metallic_predictor', 'roughness_predictor', 'albedo_predictor', 'transmisstion_weight']
        cmM = np.concatenate([1 + np.arange(256), 257 + np.arange(3)]).astype(np.int32)   # [feat, x] -> YX columns
        self._colmaps['m0'] = torch.from_numpy(cmM).to(self.dev)
        self.WpM0, self.WpTM0, self.bM0, self.dWpM0 = z(1024, 288), z(384, 1024), z(1024), z(1024, 288)
        self.WpM = [None, z(4, 256, 256), z(4, 256, 256)]
        self.WpTM = [None, z(4, 256, 256), z(4, 256, 256)]
        self.bM = [None, z(4, 256), z(4, 256)]
        self.dWpM = [None, z(4, 256, 256), z(4, 256, 256)]
        self.Ws6, self.b6, self.dWs6 = z(6, 1024), z(8), z(6, 1024)
        self.mat_layers = []
        db0 = galloc(1024)
        db12 = [None, galloc(1024), galloc(1024)]
        db6 = galloc(6)
        head_row = [0, 1, 2, 5]
        self.mat_db = (db0, db12, db6)
        for i, name in enumerate(mats):
            pre = f'color_network.{name}'
            for j, idx in enumerate((0, 2, 4, 6)):
                q = f'{pre}.{idx}'
                v = p[q + '.weight_v']
                N, K = v.shape
                if j == 0:
                    lay = _Layer(q, v, p[q + '.weight_g'], p[q + '.bias'], N, K, 288, colmap=self._colmaps['m0'])
                    lay.Wp = (self.WpM0, i * 256 * 288)
                    lay.WpT, lay.ldT = (self.WpTM0, i * 256), 1024
                    lay.dWp, lay.ldd = (self.dWpM0, i * 256 * 288), 288
                    lay.bias_p = (self.bM0, i * 256)
                    lay.db_off = db0 + i * 256
                elif j < 3:
                    lay = _Layer(q, v, p[q + '.weight_g'], p[q + '.bias'], N, K, 256)
                    lay.Wp = (self.WpM[j], i * 65536)
                    lay.WpT, lay.ldT = (self.WpTM[j], i * 65536), 256
                    lay.dWp, lay.ldd = (self.dWpM[j], i * 65536), 256
                    lay.bias_p = (self.bM[j], i * 256)
                    lay.db_off = db12[j] + i * 256
                else:
                    lay = _Layer(q, v, p[q + '.weight_g'], p[q + '.bias'], N, K, 1024, col_off=i * 256)
                    lay.Wp = (self.Ws6, head_row[i] * 1024)
                    lay.dWp, lay.ldd = (self.dWs6, head_row[i] * 1024), 1024
                    lay.bias_p = (self.b6, head_row[i])
                    lay.db_off = db6 + head_row[i]
                lay.dv_off, lay.dg_off = galloc(N * K), galloc(N)
                reg(q + '.weight_v', lay.dv_off); reg(q + '.weight_g', lay.dg_off); reg(q + '.bias', lay.db_off)
                self.mat_layers.append(lay)
        layers += self.mat_layers

        # ---- shading: light predictors ----
        def predictor(name, Kp0):
            pre = f'color_network.{name}'
            out = []
            for j, idx in enumerate((0, 2, 4, 6)):
                q = f'{pre}.{idx}'
                v = p[q + '.weight_v']
                N, K = v.shape
                Kp = Kp0 if j == 0 else 256
                lay = _Layer(q, v, p[q + '.weight_g'], p[q + '.bias'], N, K, Kp)
                lay.Wp = (z(rup(N, 128) if j < 3 else N, Kp), 0)
                if j < 3:
                    lay.WpT, lay.ldT = (z(rup(Kp, 128), 256), 0), 256
                lay.dWp, lay.ldd = (z(N, Kp), 0), Kp
                lay.dv_off, lay.dg_off, lay.db_off = galloc(N * K), galloc(N), galloc(N)
                reg(q + '.weight_v', lay.dv_off); reg(q + '.weight_g', lay.dg_off); reg(q + '.bias', lay.db_off)
                out.append(lay)
            return out
        self.outer_light = predictor('outer_light', self.ld_ol)
        self.inner_light = predictor('inner_light', 128)
        self.inner_weight = predictor('inner_weight', 96)
        self.refrac_light = predictor('refrac_light', self.ld_rl)
        layers += self.outer_light + self.inner_light + self.inner_weight + self.refrac_light
        # ---- stage 2: the IoR / thickness networks (field.py:1046-1087: 39 -> 256 ReLU -> 256 ReLU -> 256 -> 1), when the owner
        # has them ----
        self.small = {}
        for pre in ('ior_network', 'thickness_network'):
            if pre + '.0.weight_v' not in p:
                continue
            net = []
            for j, idx in enumerate((0, 2, 4, 5)):
                q = f'{pre}.{idx}'
                v = p[q + '.weight_v']
                N, K = v.shape
                Kp = 64 if j == 0 else 256
                lay = _Layer(q, v, p[q + '.weight_g'], p[q + '.bias'], N, K, Kp)
                lay.Wp = (z(rup(N, 128) if j < 3 else N, Kp), 0)
                if j < 3:
                    lay.WpT, lay.ldT = (z(rup(Kp, 128), 256), 0), 256
                lay.dWp, lay.ldd = (z(N, Kp), 0), Kp
                lay.dv_off, lay.dg_off, lay.db_off = galloc(N * K), galloc(N), galloc(N)
                reg(q + '.weight_v', lay.dv_off); reg(q + '.weight_g', lay.dg_off); reg(q + '.bias', lay.db_off)
                net.append(lay)
            self.small[pre] = net
            layers += net
        self.ior = self.small.get('ior_network')
        self.layers = layers
        self.n_grad = self._goff
        # bf16 copies of every NT-side weight table (written by the same pack launch): same shapes, same element offsets
        # 'bf16x6': the same twins hold the exact hi / mid / lo split instead (three times the elements, the fp32 table's offsets
        # times 3 -- include/nu_nerf.h NuGemmNT.B6): the NT kernel then never splits a weight tile (csrc/gemm_nt6.hip)
        self._tw = {}
        self.w6 = self.bf16 == 2 and os.environ.get('NU_PRESPLIT_WEIGHTS', '1') != '0'
        if self.h16 or self.w6:
            for lay in layers:
                for tab in (lay.Wp, lay.WpT):
                    if tab is not None and id(tab[0]) not in self._tw:
                        if self.w6:
                            ld = tab[0].shape[-1]
                            assert ld % 16 == 0, "pre-split weight planes need leading dimensions that are multiples of 16"
                            rows = -(-(tab[0].numel() // ld) // 256) * 256          # whole 256-row blocks
                            self._tw[id(tab[0])] = torch.zeros(3 * rows * ld, dtype=torch.bfloat16, device=tab[0].device)
                        else:
                            self._tw[id(tab[0])] = torch.zeros_like(tab[0], dtype=torch.bfloat16)
        self.lut = p['color_network.FG_LUT']
        # parameters that never receive a gradient in stage 1: color_network.iors.*, infinity_far_bkgr.*
        self._desc_dev = None

    def _signature(self):
        return tuple(l.v.data_ptr() for l in self.layers[:4]) + (self.layers[-1].v.data_ptr(),)

    def _upload_descs(self):
        descs = (PackDesc * len(self.layers))()
        row = 0
        for i, l in enumerate(self.layers):
            d = descs[i]
            d.v = addr(l.v, l.v_row0 * l.K)
            d.g = addr(l.g, l.v_row0) if l.g is not None else 0
            d.colmap = addr(l.colmap) if l.colmap is not None else 0
            d.Wp = addr(*l.Wp)
            d.WpT = addr(*l.WpT) if l.WpT is not None else 0
            d.dWp = addr(*l.dWp)
            d.Wp16, d.WpT16 = self._a16(l.Wp), self._a16(l.WpT)
            d.planes = 3 if self.w6 else 1
            if self.w6:          # the pack launch gets the START of each twin and the layer's place in the table (include/nu_nerf.h)
                d.Wp16, d.w6_ld = addr(self._tw[id(l.Wp[0])]), l.Wp[0].shape[-1]
                d.w6_row0 = l.Wp[1] // d.w6_ld
                assert l.Wp[1] % d.w6_ld == 0
                if l.WpT is not None:
                    d.WpT16, d.t6_ld = addr(self._tw[id(l.WpT[0])]), l.WpT[0].shape[-1]
                    d.t6_row0, d.t6_col0 = l.WpT[1] // d.t6_ld, l.WpT[1] % d.t6_ld
            d.dv_off, d.dg_off = l.dv_off, l.dg_off
            d.bias = addr(l.b) if (l.bias_p is not None) else 0
            d.bias_p = addr(*l.bias_p) if l.bias_p is not None else 0
            d.scale, d.N, d.K, d.Kp, d.ldT, d.ldd, d.row_begin, d.col_off = l.scale, l.N, l.K, l.Kp, l.ldT, l.ldd, row, l.col_off
            l.row_begin = row
            row += l.N
        self.total_rows = row
        raw = bytes(descs)
        host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        self._desc_dev = host.to(self.dev)
        self._ptr_sig = self._signature()
        self._build_net_structs()

    def _a16(self, tab):
        """Device address of the bf16 twin of a (tensor, element offset) weight table; 0 outside the bf16-storage mode."""
        if tab is None or not (self.h16 or self.w6):
            return 0
        return addr(self._tw[id(tab[0])], tab[1] * (3 if self.w6 else 1))

    def _lin(self, lay):
        return Lin(addr(*lay.Wp), addr(*lay.WpT) if lay.WpT is not None else 0, addr(*lay.dWp), addr(lay.b), lay.db_off, lay.N, lay.K,
                   lay.Kp, lay.ldT, lay.ldd, 0, self._a16(lay.Wp), self._a16(lay.WpT))

    def _build_net_structs(self):
        """Packed-layer tables of the network-level C entries (include/nu_nerf.h: NuSdfNet, NuNerfNet, NuShadeNet)."""
        self._sdf_net = SdfNet()
        for l in range(9):
            self._sdf_net.lin[l] = self._lin(self.sdf[l])
        self._nerf_net = NerfNet()
        for i in range(8):
            self._nerf_net.pts[i] = self._lin(self.nerf[i])
        self._nerf_net.feat, self._nerf_net.alpha = self._lin(self.nerf_feat), self._lin(self.nerf_alpha)
        self._nerf_net.view, self._nerf_net.rgb = self._lin(self.nerf_view), self._lin(self.nerf_rgb)
        n = ShadeNet()
        db0, db12, db6 = self.mat_db
        n.WpM0, n.WpTM0, n.bM0, n.dWpM0 = addr(self.WpM0), addr(self.WpTM0), addr(self.bM0), addr(self.dWpM0)
        for j in (1, 2):
            n.WpM[j], n.WpTM[j], n.bM[j], n.dWpM[j] = addr(self.WpM[j]), addr(self.WpTM[j]), addr(self.bM[j]), addr(self.dWpM[j])
            n.dbM_off[j] = db12[j]
        n.dbM_off[0] = db0
        n.Ws6, n.b6, n.dWs6, n.db6_off = addr(self.Ws6), addr(self.b6), addr(self.dWs6), db6
        n.WpM0_16, n.WpTM0_16 = self._a16((self.WpM0, 0)), self._a16((self.WpTM0, 0))
        for j in (1, 2):
            n.WpM16[j], n.WpTM16[j] = self._a16((self.WpM[j], 0)), self._a16((self.WpTM[j], 0))
        for name in ('outer_light', 'inner_light', 'inner_weight', 'refrac_light'):
            arr = getattr(n, name)
            for j, lay in enumerate(getattr(self, name)):
                arr[j] = self._lin(lay)
        n.lut, n.exp_max, n.sphere = addr(self.lut), self.exp_max, 1 if self.sphere_direction else 0
        n.ld_ol, n.refrac_dim, n.ld_rl = self.ld_ol, self.refrac_dim, self.ld_rl
        self._shade_net = n

    def _use_c(self):
        """Network-level C entries unless launch-by-launch sequencing is asked for (NU_PY_SEQ=1, or bench.py's per-launch
        event timing, which brackets individual GEMM launches)."""
        return not self.py_seq

    def pack(self):
        """Fold weight-norm, pad/permute and transpose every layer's weight: one launch."""
        if self._desc_dev is None or self._ptr_sig != self._signature():
            self._upload_descs()
        L.check(self.lib.nu_pack_layers(c_p(self._desc_dev.data_ptr()), len(self.layers), self.total_rows, self.stream()),
                "nu_pack_layers")

    def unpack_grads(self, flat, layers=None):
        """Weight-norm / plain weight gradients from the packed dW tables into `flat`.  layers: only these (consecutive entries of
        self.layers -- one network of a stage-2 op): every other slot of `flat` stays untouched."""
        self.flush_reductions()
        if layers is None:
            L.check(self.lib.nu_unpack_grads(c_p(self._desc_dev.data_ptr()), len(self.layers), self.total_rows,
                                             c_p(flat.data_ptr()), self.stream()), "nu_unpack_grads")
            return
        key = id(layers)
        rng = self.__dict__.setdefault('_unpack_ranges', {}).get(key)
        if rng is None or rng[0] is not layers:
            row0, rows = layers[0].row_begin, 0
            for l in layers:
                if l.row_begin != row0 + rows:
                    raise ValueError("unpack_grads(layers=...): the layers are not consecutive in the descriptor table")
                rows += l.N
            rng = self._unpack_ranges[key] = (layers, row0, rows)
        L.check(self.lib.nu_unpack_grads_range(c_p(self._desc_dev.data_ptr()), len(self.layers), rng[1], rng[2], c_p(flat.data_ptr()),
                                               self.stream()), "nu_unpack_grads_range")

    # ------------------------------------------------------------------ raw launches
    def nt(self, A, lda, B, ldb, M, N, K, C, ldc, epi, *, C2=0, ldc2=0, bias=0, H=0, ldh=0, D=0, ldd=0, Cadd=0,
           ldadd=0, zero_to=0, act_cols=0, alpha=1.0, groups=1, sA=0, sB=0, sC=0, sC2=0, sBias=0, sH=0, sD=0, sCadd=0,
           ktrue=None, ntrue=None, mask=None, B6=0):
        """C = epi(A . B^T) on the fp32-MFMA kernel.  ktrue/ntrue: unpadded extents, used only for the algorithmic
        FLOP count of the roofline report.  B6 (mode 'bf16x6' only): the pre-split planes of B (include/nu_nerf.h NuGemmNT.B6)."""
        if M <= 0:
            return
        g = GemmNT(A, lda, B, ldb, M, N, K, C, ldc, C2, ldc2, bias, H, ldh, D, ldd, Cadd, ldadd, zero_to, act_cols,
                   alpha, groups, sA, sB, sC, sC2, sBias, sH, sD, sCadd, epi, self.bf16,
                   mask.data_ptr() if mask is not None else 0, mask._nu_nct if mask is not None else 0, 0, B6)
        kt = self._ktime if self.ktime_on else None
        if kt is not None:
            e0, e1 = self._event_pair()
            e0.record()
        L.check(self.lib.nu_gemm_nt_ex(ctypes.byref(g), self.stream()), "nu_gemm_nt_ex")
        if kt is not None:
            e1.record()
            reads_h = bool(H) and not (mask is not None and epi in (EPI_MUL_DRELU, EPI_B_RELU))   # sign bits replace H
            nmat = 1 + (1 if C2 else 0) + (1 if reads_h else 0) + (1 if D else 0) + (1 if Cadd else 0)
            abytes = 4.0 * groups * (M * (ktrue or K) + M * (ntrue or N) * nmat + N * (ktrue or K))
            kt['nt'].append((e0, e1, 2.0 * M * (ntrue or N) * (ktrue or K) * groups, abytes))

    def nt_desc(self, A, lda, B, ldb, M, N, K, C, ldc, epi, *, C2=0, ldc2=0, bias=0, H=0, ldh=0, D=0, ldd=0, Cadd=0, ldadd=0, zero_to=0,
                act_cols=0, alpha=1.0, groups=1, sA=0, sB=0, sC=0, sC2=0, sBias=0, sH=0, sD=0, sCadd=0, mask=None):
        """One problem of an `nt_batch` launch (the arguments of `nt`)."""
        return GemmNT(A, lda, B, ldb, M, N, K, C, ldc, C2, ldc2, bias, H, ldh, D, ldd, Cadd, ldadd, zero_to, act_cols,
                      alpha, groups, sA, sB, sC, sC2, sBias, sH, sD, sCadd, epi, self.bf16,
                      mask.data_ptr() if mask is not None else 0, mask._nu_nct if mask is not None else 0, 0, 0)

    def nt_batch(self, descs):
        """Several independent NT problems in as few persistent launches as possible (nu_gemm_nt_batch: runs of one epilogue kind
        share ONE tile list) -- level j of the light predictors of a stage-2 shading call.  Bit-identical to one `nt` per problem."""
        descs = [g for g in descs if g.M > 0]
        if not descs:
            return
        arr = (GemmNT * len(descs))(*descs)
        kt = self._ktime if self.ktime_on else None
        if kt is not None:
            e0, e1 = self._event_pair()
            e0.record()
        L.check(self.lib.nu_gemm_nt_batch(arr, len(descs), self.stream()), "nu_gemm_nt_batch")
        if kt is not None:
            e1.record()
            fl = by = 0.0
            for g in descs:
                nmat = 1 + (1 if g.C2 else 0) + (1 if (g.H and not (g.mask and g.epi in (EPI_MUL_DRELU, EPI_B_RELU))) else 0) + (1 if g.D else 0) + (1 if g.Cadd else 0)
                fl += 2.0 * g.M * g.N * g.K * g.groups
                by += 4.0 * g.groups * (g.M * g.K + g.M * g.N * nmat + g.N * g.K)
            kt['nt'].append((e0, e1, fl, by))

    def begin_kernel_timing(self, reserve=0, py_reserve=None):
        """Bracket every GEMM launch with HIP events on the launch stream (bench.py's roofline leg).  The network-level C
        entries record the events themselves (NuOpCtx.ev); the Python-sequenced path records them here.  The event pairs cost
        about 4 ms per step (two queue barriers per launch), so the bench switches `ktime_on` per step to sample.
        The pool is SPLIT: the first `reserve - py_reserve` events belong to the C entries (used in order, NuOpCtx.ev_cap), the
        rest to the Python-sequenced launches (stage 2 uses both paths in one step) -- no event is ever recorded twice."""
        self._ktime = {'nt': [], 'tn': []}
        # events are created here, outside the timed region: torch makes the HIP event at the first record(), and creating
        # the ~650 events of one bracketed step used to cost that step 70 ms
        n = max(int(reserve), 4) // 4 * 4
        if py_reserve is None:
            py_reserve = n if self.py_seq else n // 4
        py_reserve = min(max(int(py_reserve), 0) // 2 * 2, n)
        pool = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        for ev in pool:
            ev.record()
        torch.cuda.synchronize(self.dev)
        n_c = 0 if self.py_seq else n - py_reserve
        self._ev_c = pool[:n_c]                               # the C entries use these in order
        self._ev_pool = pool[n_c:]                            # the Python-sequenced launches pop pairs from these
        self._ev_py_exhausted = False
        self._ev_handles = (c_p * max(n_c, 1))(*[ev.cuda_event for ev in self._ev_c])
        self._ev_meta = (ctypes.c_double * (3 * max(n_c // 2, 1)))()
        self._ctx.nev, self._ctx.ev_cap, self._ctx.ev_meta = 0, n_c, ctypes.addressof(self._ev_meta)
        self.ktime_on = True

    @property
    def ktime_on(self):
        return self._ktime_on

    @ktime_on.setter
    def ktime_on(self, on):
        self._ktime_on = bool(on)
        live = self._ktime_on and self._ktime is not None and not self.py_seq and self._ctx.ev_cap > 0
        self._ctx.ev = ctypes.addressof(self._ev_handles) if live else 0

    def _event_pair(self):
        pool = self._ev_pool
        if len(pool) >= 2:
            return pool.pop(), pool.pop()
        self._ev_py_exhausted = True                          # reported: events made here cost the bracketed step host time
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def end_kernel_timing(self):
        kt, self._ktime = self._ktime, None
        self._ctx.ev = 0
        torch.cuda.synchronize(self.dev)
        for i in range(self._ctx.nev // 2):                  # launches the C entries bracketed
            kind, flops, nbytes = self._ev_meta[3 * i:3 * i + 3]
            kt['tn' if kind else 'nt'].append((self._ev_c[2 * i], self._ev_c[2 * i + 1], flops, nbytes))
        out = {'event_capacity_reached': bool(self._ctx.ev_cap and self._ctx.nev >= self._ctx.ev_cap),
               'python_event_pool_exhausted': bool(getattr(self, '_ev_py_exhausted', False))}
        self._ctx.nev = 0
        for key, pre in (('nt', ''), ('tn', 'tn_')):
            out[pre + 'seconds'] = sum(a.elapsed_time(b) for a, b, _, _ in kt[key]) * 1e-3
            out[pre + 'flops'] = sum(f for _, _, f, _ in kt[key])
            out[pre + 'bytes'] = sum(b for _, _, _, b in kt[key])
            out[pre + 'launches'] = len(kt[key])
        return out

    def wgrad(self, A0, lda0, B0, ldb0, P, N1, N2, dW, ldw, db, *, A1=0, lda1=0, B1=0, ldb1=0, groups=1, sA0=0, sB0=0,
              sA1=0, sB1=0, sW=0, sDb=0, n2true=None):
        kt = self._ktime if (self.ktime_on and self._wg_depth == 0 and self._ctx.ev == 0) else None     # (queued launches: timed by the library)
        if kt is not None:
            e0, e1 = self._event_pair()
            e0.record()
        self._wgrad(A0, lda0, B0, ldb0, P, N1, N2, dW, ldw, db, A1, lda1, B1, ldb1, groups, sA0, sB0, sA1, sB1, sW, sDb)
        if kt is not None:
            e1.record()
            npair = 2 if A1 else 1
            kt['tn'].append((e0, e1, 2.0 * P * N1 * (n2true or N2) * groups * npair,
                             4.0 * groups * (npair * P * (N1 + (n2true or N2)) + N1 * N2)))

    def _wgrad(self, A0, lda0, B0, ldb0, P, N1, N2, dW, ldw, db, A1, lda1, B1, ldb1, groups, sA0, sB0, sA1, sB1, sW, sDb):
        """Queue one weight gradient (nu_wgrad_defer).  Inside `with self.wgrad_batch():` the queue is launched when the block
        exits -- the caller keeps every operand alive and unmodified until then; outside, at once."""
        if P <= 0:
            return
        self._ensure_arena()
        g = GemmTN(A0, lda0, B0, ldb0, A1, lda1, B1, ldb1, P, N1, N2, 0, 0, 1, groups, sA0, sB0, sA1, sB1, 0, 0, self.bf16, 0)
        npair = 2 if A1 else 1
        L.check(self.lib.nu_wgrad_defer(ctypes.byref(self._ctx), ctypes.byref(g), c_p(dW), ldw, c_ll(sW), c_p(db), c_ll(sDb),
                                        ctypes.c_double(2.0 * P * N1 * N2 * groups * npair),
                                        ctypes.c_double(4.0 * groups * (npair * P * (N1 + N2) + N1 * N2)), self.stream()), "nu_wgrad_defer")
        if self._wg_depth == 0:
            self.flush_wgrads()

    def flush_wgrads(self):
        L.check(self.lib.nu_wgrad_flush(ctypes.byref(self._ctx), self.stream()), "nu_wgrad_flush")

    def wgrad_batch(self):
        """Context of a backward pass sequenced from Python: weight gradients queued inside are launched together at the exit
        (one launch per tile class, like the network-level C entries).  The caller must keep the operands of every queued weight
        gradient referenced until the block ends (`keep(t)` on the returned object holds a tensor for it)."""
        eng = self

        class _Batch:
            def keep(self, *ts):
                eng._wg_held.extend(ts)             # (nested blocks share the list: it is dropped when the outermost one has flushed)
                return ts[0] if len(ts) == 1 else ts

            def __enter__(self):
                eng._wg_depth += 1
                return self

            def __exit__(self, et, ev, tb):
                eng._wg_depth -= 1
                if eng._wg_depth == 0:
                    if et is None:
                        eng.flush_wgrads()
                    else:
                        eng._ctx.npend = 0          # an error is propagating: drop the queue instead of launching on dead buffers
                    eng._wg_held = []
                return False
        return _Batch()

    def skinny_fwd(self, H, ldh, P, K, Ws, ldw, b, NO, out, ldo):
        L.check(self.lib.nu_skinny_fwd(c_p(H), ldh, P, K, c_p(Ws), ldw, c_p(b), NO, c_p(out), ldo, self.stream()),
                "nu_skinny_fwd")

    def skinny_bwd(self, dy, ldy, H, ldh, P, K, Ws, ldw, NO, dH, lddh, relu_mask, accumulate, dWs, lddw, db):
        if self._ctx.ndesc + 2 > self._rd_cap:
            self._forced_flush()
        ws, nb = self._arena_take(self.lib.nu_skinny_bwd_workspace_bytes(K, NO))
        L.check(self.lib.nu_skinny_bwd_enqueue(c_p(dy), ldy, c_p(H), ldh, P, K, c_p(Ws), ldw, NO, c_p(dH), lddh, relu_mask,
                                               accumulate, c_p(dWs), lddw, c_p(db), c_p(ws), c_ll(nb), self._rd,
                                               self._ndesc_p, self._rd_cap, self.stream()), "nu_skinny_bwd_enqueue")

    def colsum(self, A, lda, P, ncols, out, accumulate):
        if self._ctx.ndesc + 1 > self._rd_cap:
            self._forced_flush()
        ws, nb = self._arena_take(self.lib.nu_colsum_workspace_bytes(ncols))
        L.check(self.lib.nu_colsum_enqueue(c_p(A), lda, P, ncols, c_p(out), accumulate, c_p(ws), c_ll(nb), self._rd,
                                           self._ndesc_p, self._rd_cap, self.stream()), "nu_colsum_enqueue")

    # ------------------------------------------------------------------ SDF network
    def sdf_forward(self, X, x_ld, P, *, keep=True, want_feat=True):
        """SDF MLP forward on P points (X: device address of [P, x_ld] rows whose first 3 floats are x).
        keep=False runs the no-grad sampler chain with two ping-pong buffers and only the sdf column.
        Returns a dict of activation tensors."""
        lib, S = self.lib, self.stream()
        e = self.empty
        if (not keep and not want_feat and self.bf16 == 0 and self._fused_sdf and P <= self._FUSED_SDF_MAX_POINTS
                and getattr(self, '_sdf_net', None) is not None):
            # nothing is kept: the whole network in ONE kernel (csrc/fused_sdf.hip), bit-identical to the layered path below.
            # Measured (scripts/bench_fused_sdf.py, profiles/r03): 93 vs 122 us at 8 192 points, 153 vs 186 at 16 384, 304 vs 315
            # at 32 768; from 65 536 points on the layered GEMMs (two workgroups per CU) are 4 % faster, so those keep them
            sdf = e(P)
            L.check(lib.nu_sdf_fused_fwd(ctypes.byref(self._sdf_net), c_p(X), x_ld, P, c_p(addr(sdf)), S), "nu_sdf_fused_fwd")
            return {'P': P, 'sdf': sdf}
        if not keep and not want_feat and self.h16 and self._fused_sdf and getattr(self, '_sdf_net', None) is not None:
            # bf16 storage: the same network in ONE kernel (csrc/fused_sdf.hip, sdf_fused16_fwd_kernel), bit-identical to the layered
            # bf16-storage path below -- a point costs 12 bytes in and 4 out instead of 1 KB per layer
            sdf = e(P)
            L.check(lib.nu_sdf_fused16_fwd(ctypes.byref(self._sdf_net), c_p(X), x_ld, P, c_p(addr(sdf)), S), "nu_sdf_fused16_fwd")
            return {'P': P, 'sdf': sdf}
        a = {'P': P}
        a['E'] = e(P, 64)
        a['U4'] = e(P, 256)
        a['YX'] = e(P, 288) if want_feat else None
        ls = self.sdf
        eh = self.empty_h               # H[1..3], H[5..7]: bf16 in the bf16-storage mode; U4 (= H[4]) and the kept H[8] fp32
        h8 = e if want_feat else eh     # without the feature head only the sdf head reads H[8]: bf16 too (NuOpCtx.h16 rule)
        if keep:
            H = [None] + [eh(P, 256) for _ in range(3)] + [a['U4']] + [eh(P, 256) for _ in range(3)] + [h8(P, 256)]
        else:
            t0, t1 = eh(P, 256), eh(P, 256)
            H = [None, t0, t1, t0, a['U4'], t0, t1, t0, t1 if not want_feat else e(P, 256)]
        a['H'] = H
        if self._use_c():          # one C call sequences the embedding, the eight hidden GEMMs and the output layer
            cb = SdfBufs(P=P, E=addr(a['E']), U4=addr(a['U4']), YX=addr(a['YX']))
            for l in range(1, 9):
                cb.H[l] = addr(H[l])
            a['sdf'] = None if want_feat else e(P)
            cb.sdf = addr(a['sdf'])
            a['cb'] = cb
            L.check(lib.nu_sdf_mlp_fwd(ctypes.byref(self._ctx), ctypes.byref(self._sdf_net), c_p(X), x_ld, ctypes.byref(cb),
                                       1 if want_feat else 0, S), "nu_sdf_mlp_fwd")
            return a
        L.check(lib.nu_sdf_embed(c_p(X), x_ld, P, c_p(addr(a['E'])), c_p(addr(a['U4'])),
                                 c_p(addr(a['YX'])), S), "nu_sdf_embed")
        src, lds, K = a['E'], 64, 64
        for l in range(8):
            N = ls[l].N
            self.nt(addr(src), lds, addr(*ls[l].Wp), ls[l].Kp, P, N, K, addr(H[l + 1]), 256, EPI_BIAS_SOFTPLUS,
                    bias=addr(ls[l].b), zero_to=N, ktrue=ls[l].K)
            src, lds, K = H[l + 1], 256, 256
        # last layer: row 0 = sdf (skinny), rows 1..256 = feature (N = 256 GEMM)
        if want_feat:
            self.skinny_fwd(addr(H[8]), 256, P, 256, addr(*ls[8].Wp), 256, addr(ls[8].b), 1, addr(a['YX']), 288)
            self.nt(addr(H[8]), 256, addr(ls[8].Wp[0], 256), 256, P, 256, 256, addr(a['YX'], 1), 288, EPI_BIAS_NONE,
                    bias=addr(ls[8].b, 1))
            a['sdf'] = None
        else:
            a['sdf'] = e(P)
            self.skinny_fwd(addr(H[8]), 256, P, 256, addr(*ls[8].Wp), 256, addr(ls[8].b), 1, addr(a['sdf']), 1)
        return a

    def sdf_normal(self, a):
        """Reverse sweep: n = d sdf / d x  (field.py:158-170), keeping delta_l for the second-order backward."""
        lib, S, P, ls, H = self.lib, self.stream(), a['P'], self.sdf, a['H']
        e = self.empty
        D = [self.empty_h(P, 256) if l in (0, 1, 2, 4, 5, 6) else e(P, 256) for l in range(8)]
        a['D'] = D
        if self._use_c():
            cb = self._sdf_cb(a)
            a['G0'], a['n'] = e(P, 64), e(P, 3)
            for l in range(8):
                cb.D[l] = addr(D[l])
            cb.G0, cb.n = addr(a['G0']), addr(a['n'])
            L.check(lib.nu_sdf_mlp_normal(ctypes.byref(self._ctx), ctypes.byref(self._sdf_net), ctypes.byref(cb), S), "nu_sdf_mlp_normal")
            return a['n']
        L.check(lib.nu_rowscale_dsp(c_p(addr(H[8])), 256, P, 256, c_p(addr(*ls[8].Wp)), c_p(addr(D[7])), 256, S),
                "nu_rowscale_dsp")
        for l in range(7, 0, -1):
            # G_{u_l} = D_l . Wp_l ; delta_{l-1} = G[:, :N_{l-1}] * sp'(H_l)
            Kred = rup(ls[l].N, 32)                    # reduction over layer l's outputs (217 -> 224 at l = 3)
            Nout = ls[l - 1].N                          # columns that carry an activation derivative
            if l == 4:
                # columns 217..255 of G_{u_4} are the skip gradient w.r.t. the embedding: written plain
                self.nt(addr(D[l]), 256, addr(*ls[l].WpT), ls[l].ldT, P, 256, Kred, addr(D[l - 1]), 256, EPI_MUL_DSP,
                        H=addr(H[l]), ldh=256, act_cols=217)
            else:
                self.nt(addr(D[l]), 256, addr(*ls[l].WpT), ls[l].ldT, P, Nout, Kred, addr(D[l - 1]), 256, EPI_MUL_DSP,
                        H=addr(H[l]), ldh=256, zero_to=256 if l != 4 else 0)
        G0 = e(P, 64)
        self.nt(addr(D[0]), 256, addr(*ls[0].WpT), ls[0].ldT, P, 39, 256, addr(G0), 64, EPI_PLAIN, zero_to=64)
        a['G0'] = G0
        a['n'] = e(P, 3)
        L.check(lib.nu_embed_jt(c_p(addr(a['E'])), c_p(addr(G0)), 64, c_p(addr(D[3], 217)), 256, P, c_p(addr(a['n'])), S),
                "nu_embed_jt")
        return a['n']

    def sdf_backward(self, a, dYX, nbar, flat, dx=None):
        """Backward of (y, n) w.r.t. the SDF parameters given dYX [P,288] (cols 0..256 = d y) and nbar [P,3]
        (may be None: first-order only).  Writes packed weight grads + bias grads (into `flat`).  dx [P,3] (optional):
        receives d L / d x through the network (stage 2), to which the caller adds dYX's x-slot."""
        lib, S, P, ls, H, E = self.lib, self.stream(), a['P'], self.sdf, a['H'], a['E']
        e = self.empty
        second = nbar is not None
        if self._use_c():
            cb = self._sdf_cb(a)
            keep = []
            eh = self.empty_h          # bf16-storage rule: C[l] / Aux[l] like D[l], Q[l] like H[l]
            if second:
                for l in range(8):
                    t = eh(P, 256) if l in (0, 1, 2, 4, 5, 6) else e(P, 256)
                    keep.append(t)
                    cb.C[l] = addr(t)
                for l in range(9):
                    t = e(P, 64) if l == 0 else (eh(P, 256) if l in (1, 2, 3, 5, 6, 7) else e(P, 256))
                    keep.append(t)
                    cb.Q[l] = addr(t)
            else:
                for l in range(8):
                    t = eh(P, 256) if l in (0, 1, 2, 4, 5, 6) else e(P, 256)
                    keep.append(t)
                    cb.Aux[l] = addr(t)
            if dx is not None:
                t = e(P, 64)
                keep.append(t)
                cb.dE0 = addr(t)
            self._ensure_arena()
            self._ctx.flat = addr(flat)
            L.check(lib.nu_sdf_mlp_bwd(ctypes.byref(self._ctx), ctypes.byref(self._sdf_net), ctypes.byref(cb), c_p(addr(dYX)),
                                       c_p(addr(nbar)), c_p(addr(dx)), S), "nu_sdf_mlp_bwd")
            a['_bwd_keep'] = keep          # buffers stay referenced until the caller drops the activation dict
            return
        Cb = [None] * 8
        Q = [None] * 9
        if second:
            D = a['D']
            Q[0] = e(P, 64)
            Q[4] = e(P, 256)
            L.check(lib.nu_embed_j(c_p(addr(E)), c_p(addr(nbar)), P, c_p(addr(Q[0])), c_p(addr(Q[4])), S), "nu_embed_j")
            src, lds, K = Q[0], 64, 64
            for l in range(8):
                N = ls[l].N
                Cb[l] = e(P, 256)
                if l + 1 != 4:
                    Q[l + 1] = e(P, 256)
                self.nt(addr(src), lds, addr(*ls[l].Wp), ls[l].Kp, P, N, K, addr(Q[l + 1]), 256, EPI_Q_SP,
                        C2=addr(Cb[l]), ldc2=256, H=addr(H[l + 1]), ldh=256, D=addr(D[l]), ldd=256,
                        zero_to=N if l == 3 else 256)
                src, lds, K = Q[l + 1], 256, 256
        # B sweep: abar_7 = (dYX . W8) * sp'(H8) + C7 ...
        A = [None] * 8
        for l in range(7, -1, -1):
            if l == 7:
                srcA, lda, K, WT, ldT = dYX, 288, 288, ls[8].WpT, ls[8].ldT
            else:
                srcA, lda, K, WT, ldT = A[l + 1], 256, rup(ls[l + 1].N, 32), ls[l + 1].WpT, ls[l + 1].ldT
            N = ls[l].N
            A[l] = Cb[l] if second else e(P, 256)
            if l == 3 and dx is not None:
                # also keep the plain columns 217..255: gradient w.r.t. the embedding copy of the skip connection
                self.nt(addr(srcA), lda, addr(*WT), ldT, P, 256, K, addr(A[l]), 256, EPI_B_SP if second else EPI_MUL_DSP,
                        H=addr(H[l + 1]), ldh=256, Cadd=addr(Cb[l]) if second else 0, ldadd=256, act_cols=217)
            else:
                self.nt(addr(srcA), lda, addr(*WT), ldT, P, N, K, addr(A[l]), 256, EPI_B_SP if second else EPI_MUL_DSP,
                        H=addr(H[l + 1]), ldh=256, Cadd=addr(Cb[l]) if second else 0, ldadd=256, zero_to=256)
        # weight gradients: one queue, launched together (every operand is held by the lists above until this method returns)
        with self.wgrad_batch():
            for l in range(8):
                u, ldu = (E, 64) if l == 0 else (H[l], 256)
                if second:
                    self.wgrad(addr(A[l]), 256, addr(u), ldu, P, ls[l].N, ls[l].Kp, addr(*ls[l].dWp), ls[l].ldd,
                               addr(flat, ls[l].db_off), A1=addr(a['D'][l]), lda1=256, B1=addr(Q[l]), ldb1=64 if l == 0 else 256)
                else:
                    self.wgrad(addr(A[l]), 256, addr(u), ldu, P, ls[l].N, ls[l].Kp, addr(*ls[l].dWp), ls[l].ldd,
                               addr(flat, ls[l].db_off))
            self.wgrad(addr(dYX), 288, addr(H[8]), 256, P, 257, 256, addr(*ls[8].dWp), 256, addr(flat, ls[8].db_off))
        if second:
            # d W8[sdf row] += sum_p q_8   (the reverse sweep starts from W8's sdf row)
            self.colsum(addr(Q[8]), 256, P, 256, addr(*ls[8].dWp), 1)
        if dx is not None:
            dE0 = e(P, 64)
            self.nt(addr(A[0]), 256, addr(*ls[0].WpT), ls[0].ldT, P, 39, 256, addr(dE0), 64, EPI_PLAIN, zero_to=64)
            L.check(lib.nu_embed_jt2(c_p(addr(E)), c_p(addr(dE0)), 64, c_p(addr(A[3], 217)), 256,
                                     c_p(addr(a['G0']) if second else 0), 64, c_p(addr(a['D'][3], 217) if second else 0), 256,
                                     c_p(addr(nbar) if second else 0), P, c_p(addr(dx)), 0, S), "nu_embed_jt2")

    def _sdf_cb(self, a):
        """The C buffer table of an activation dict (built on demand for dicts made by the Python-sequenced forward)."""
        cb = a.get('cb')
        if cb is None:
            cb = SdfBufs(P=a['P'], E=addr(a['E']), U4=addr(a['U4']), YX=addr(a['YX']), sdf=addr(a.get('sdf')))
            for l in range(1, 9):
                cb.H[l] = addr(a['H'][l])
            a['cb'] = cb
        if 'D' in a:
            for l in range(8):
                cb.D[l] = addr(a['D'][l])
            cb.G0, cb.n = addr(a.get('G0')), addr(a.get('n'))
        return cb

    # ------------------------------------------------------------------ generic ReLU stacks
    def relu_stack_fwd(self, layers, X, ldx, rows):
        """3 hidden ReLU layers (make_predictor, field.py:371-408); returns hidden activations [H1,H2,H3]."""
        Hs = []
        src, lds = X, ldx
        for j in range(3):
            lay = layers[j]
            Hn = self.empty(rows, 256)
            self.nt(addr(src), lds, addr(*lay.Wp), lay.Kp, rows, 256, lay.Kp, addr(Hn), 256, EPI_BIAS_RELU, bias=addr(lay.b),
                    mask=self.relu_mask(Hn, rows, 256))
            Hs.append(Hn)
            src, lds = Hn, 256
        return Hs

    def relu_stack_bwd(self, layers, X, ldx, rows, Hs, dH3, flat, dX=None, lddx=0, dx_cols=0):
        """dH3 = gradient w.r.t. post-ReLU H3 already masked by relu'(H3) (i.e. d pre-activation of layer 2)."""
        with self.wgrad_batch() as wb:        # (inside a caller's batch the queue is launched with the caller's)
            dA = wb.keep(dH3)
            for j in (2, 1, 0):
                lay = layers[j]
                u, ldu = (X, ldx) if j == 0 else (Hs[j - 1], 256)
                self.wgrad(addr(dA), 256, addr(u), ldu, rows, 256, lay.Kp, addr(*lay.dWp), lay.ldd, addr(flat, lay.db_off))
                if j > 0:
                    nxt = wb.keep(self.empty(rows, 256))
                    self.nt(addr(dA), 256, addr(*lay.WpT), lay.ldT, rows, 256, 256, addr(nxt), 256, EPI_MUL_DRELU,
                            H=addr(Hs[j - 1]), ldh=256, mask=getattr(Hs[j - 1], '_nu_mask', None))
                    dA = nxt
                elif dX is not None:
                    self.nt(addr(dA), 256, addr(*lay.WpT), lay.ldT, rows, dx_cols, 256, addr(dX), lddx, EPI_PLAIN)

    # ------------------------------------------------------------------ shading stack
    def shading_forward(self, a, pt, idx, P, color_rm, extra_dirs=None, extra_pts=None):
        """Materials -> encodings -> 4 light predictors -> combine (field.py:684-777).
        a: SDF activations (YX, E, n).  extra_dirs [R,3]: per-ray directions whose mirror query IDE(d,0)
        rides along the outer_light batch (colour_spec, renderer_zerothick.py:780-781)."""
        if self.light_pos_freq != 6:
            raise NotImplementedError("the fused stage-1 shading encodes positions with 6 frequencies (every stage-1 config); "
                                      "light_pos_freq != 6 runs through the network-level ops (nets.py / shading_glue.py)")
        lib, S = self.lib, self.stream()
        e = self.empty
        s = {'P': P}
        YX = a['YX']
        if self._use_c():
            return self._c_shading_forward(a, pt, idx, P, color_rm, extra_dirs, extra_pts)
        # materials: layer 0 batched (N=1024), layers 1-2 grouped x4, block-diagonal 6-wide head
        M1, M2, M3 = e(P, 1024), e(P, 1024), e(P, 1024)
        self.nt(addr(YX), 288, addr(self.WpM0), 288, P, 1024, 288, addr(M1), 1024, EPI_BIAS_RELU, bias=addr(self.bM0),
                mask=self.relu_mask(M1, P, 1024))
        for j, (src, dst) in ((1, (M1, M2)), (2, (M2, M3))):
            self.nt(addr(src), 1024, addr(self.WpM[j]), 256, P, 256, 256, addr(dst), 1024, EPI_BIAS_RELU,
                    bias=addr(self.bM[j]), groups=4, sA=256, sB=65536, sC=256, sBias=256, mask=self.relu_mask(dst, P, 1024))
        Mraw = e(P, 8)
        self.skinny_fwd(addr(M3), 1024, P, 1024, addr(self.Ws6), 1024, addr(self.b6), 6, addr(Mraw), 8)
        s.update(M1=M1, M2=M2, M3=M3, Mraw=Mraw)
        # encodings
        R = 0 if extra_dirs is None else extra_dirs.shape[0]
        rows_ol = 3 * P + R
        ld_ol, ld_rl, sph = self.ld_ol, self.ld_rl, 1 if self.sphere_direction else 0
        OLin, ILin, IWin, RLin, SD = e(rows_ol, ld_ol), e(2 * P, 128), e(P, 96), e(P, ld_rl), e(P, 8)
        L.check(lib.nu_shade_encode_fwd(c_p(addr(a['n'])), c_p(addr(pt)), 8, c_p(addr(a['E'])), c_p(addr(Mraw)), 8, P, sph,
                                        ld_ol, self.refrac_dim, ld_rl, c_p(addr(OLin)), c_p(addr(ILin)), c_p(addr(IWin)),
                                        c_p(addr(RLin)), c_p(addr(SD)), S), "nu_shade_encode_fwd")
        if R:
            L.check(lib.nu_spec_encode(c_p(addr(extra_dirs)), c_p(addr(extra_pts)), R, sph if extra_pts is not None else 0,
                                       c_p(addr(OLin, 3 * P * ld_ol)), ld_ol, S), "nu_spec_encode")
        s.update(OLin=OLin, ILin=ILin, IWin=IWin, RLin=RLin, SD=SD, R=R, rows_ol=rows_ol)
        # light predictors
        s['OLh'] = self.relu_stack_fwd(self.outer_light, OLin, ld_ol, rows_ol)
        s['ILh'] = self.relu_stack_fwd(self.inner_light, ILin, 128, 2 * P)
        s['IWh'] = self.relu_stack_fwd(self.inner_weight, IWin, 96, P)
        s['RLh'] = self.relu_stack_fwd(self.refrac_light, RLin, ld_rl, P)
        OLo, ILo, IWo, RLo = e(rows_ol, 4), e(2 * P, 4), e(P), e(P, 4)
        for lay, Hs, out, rows, no, ldo in ((self.outer_light[3], s['OLh'], OLo, rows_ol, 3, 4),
                                            (self.inner_light[3], s['ILh'], ILo, 2 * P, 3, 4),
                                            (self.inner_weight[3], s['IWh'], IWo, P, 1, 1),
                                            (self.refrac_light[3], s['RLh'], RLo, P, 3, 4)):
            self.skinny_fwd(addr(Hs[2]), 256, rows, 256, addr(*lay.Wp), 256, addr(lay.b), no, addr(out), ldo)
        s.update(OLo=OLo, ILo=ILo, IWo=IWo, RLo=RLo)
        s['aux'] = e(P, 4)
        L.check(lib.nu_shade_combine_fwd(c_p(addr(Mraw)), 8, c_p(addr(OLo)), c_p(addr(ILo)), c_p(addr(IWo)),
                                         c_p(addr(RLo)), c_p(addr(SD)), c_p(addr(self.lut)), c_p(addr(idx)), P,
                                         c_f(self.exp_max), c_p(addr(color_rm)), c_p(addr(s['aux'])), S),
                "nu_shade_combine_fwd")
        return s

    def _c_shading_forward(self, a, pt, idx, P, color_rm, extra_dirs, extra_pts):
        """shading_forward through nu_shading_stack_fwd: this method only allocates the buffers and fills the pointer table."""
        lib, S, e = self.lib, self.stream(), self.empty
        R = 0 if extra_dirs is None else extra_dirs.shape[0]
        rows_ol = 3 * P + R
        ld_ol, ld_rl = self.ld_ol, self.ld_rl
        s = {'P': P, 'R': R, 'rows_ol': rows_ol}
        cb = ShadeBufs(P=P, R=R, extra_dirs=addr(extra_dirs), extra_pts=addr(extra_pts))
        eh = self.empty_h              # the material and light-predictor hidden layers: bf16 in the bf16-storage mode
        M = [eh(P, 1024), eh(P, 1024), eh(P, 1024)]
        for j in range(3):
            cb.M[j], cb.maskM[j] = addr(M[j]), addr(self.relu_mask(M[j], P, 1024))
        s.update(M1=M[0], M2=M[1], M3=M[2], Mraw=e(P, 8), OLin=e(rows_ol, ld_ol), ILin=e(2 * P, 128), IWin=e(P, 96), RLin=e(P, ld_rl),
                 SD=e(P, 8), OLo=e(rows_ol, 4), ILo=e(2 * P, 4), IWo=e(P), RLo=e(P, 4), aux=e(P, 4))
        for k in ('Mraw', 'OLin', 'ILin', 'IWin', 'RLin', 'SD', 'OLo', 'ILo', 'IWo', 'RLo', 'aux'):
            setattr(cb, k, addr(s[k]))
        for key, rows, arr, marr in (('OLh', rows_ol, cb.OLh, cb.maskOL), ('ILh', 2 * P, cb.ILh, cb.maskIL),
                                     ('IWh', P, cb.IWh, cb.maskIW), ('RLh', P, cb.RLh, cb.maskRL)):
            Hs = [eh(rows, 256), eh(rows, 256), eh(rows, 256)]
            for j in range(3):
                arr[j], marr[j] = addr(Hs[j]), addr(self.relu_mask(Hs[j], rows, 256))
            s[key] = Hs
        s['cb'] = cb
        L.check(lib.nu_shading_stack_fwd(ctypes.byref(self._ctx), ctypes.byref(self._shade_net), ctypes.byref(cb), c_p(addr(a['YX'])),
                                         c_p(addr(a['E'])), c_p(addr(a['n'])), c_p(addr(pt)), c_p(addr(idx)), c_p(addr(color_rm)), S),
                "nu_shading_stack_fwd")
        return s

    def _c_shading_backward(self, a, s, pt, idx, dcolor_rm, flat, d_spec_raw, d_occ_raw, d_mat_raw=None):
        lib, S, e, P = self.lib, self.stream(), self.empty, s['P']
        rows_ol, R = s['rows_ol'], s['R']
        cb = s['cb']
        dMraw, dOLo, dILo, dIWo, dRLo, dNoV = e(P, 8), e(rows_ol, 4), e(2 * P, 4), e(P), e(P, 4), e(P)
        cb.dMraw, cb.dOLo, cb.dILo, cb.dIWo, cb.dRLo, cb.dNoV = (addr(t) for t in (dMraw, dOLo, dILo, dIWo, dRLo, dNoV))
        args = (ctypes.byref(self._ctx), ctypes.byref(self._shade_net), ctypes.byref(cb), c_p(addr(a['YX'])), c_p(addr(a['n'])),
                c_p(addr(pt)), c_p(addr(idx)), c_p(addr(dcolor_rm)))
        L.check(lib.nu_shading_stack_bwd(*args, 0, S), "nu_shading_stack_bwd(0)")
        if R:
            if d_spec_raw is not None:
                dOLo[3 * P:, :3] = d_spec_raw
                dOLo[3 * P:, 3] = 0
            else:
                dOLo[3 * P:].zero_()
        if d_occ_raw is not None:
            dIWo += d_occ_raw
        if d_mat_raw is not None:
            dMraw += d_mat_raw
        keep = []
        for i, rows in enumerate((rows_ol, 2 * P, P, P)):
            t = self.empty_h(rows, 256)
            keep.append(t)
            cb.dH3[i] = addr(t)
        for arr, rows in ((cb.tmpOL, rows_ol), (cb.tmpIL, 2 * P), (cb.tmpIW, P), (cb.tmpRL, P)):
            for j in range(2):
                t = self.empty_h(rows, 256)
                keep.append(t)
                arr[j] = addr(t)
        dOLin, dILin, dn, dYX = e(rows_ol, self.ld_ol), e(2 * P, 128), e(P, 3), e(P, 288)
        dM = [self.empty_h(P, 1024), self.empty_h(P, 1024), self.empty_h(P, 1024)]
        cb.dOLin, cb.dILin, cb.dn, cb.dYX = addr(dOLin), addr(dILin), addr(dn), addr(dYX)
        for j in range(3):
            cb.dM[j] = addr(dM[j])
        self._ensure_arena()
        self._ctx.flat = addr(flat)
        L.check(lib.nu_shading_stack_bwd(*args, 1, S), "nu_shading_stack_bwd(1)")
        return dYX, dn

    def shading_backward(self, a, s, pt, idx, dcolor_rm, flat, d_spec_raw=None, d_occ_raw=None, d_mat_raw=None):
        """Returns (dYX [P,288] with feature/x columns filled, dn_shade [P,3])."""
        lib, S, P = self.lib, self.stream(), s['P']
        e = self.empty
        rows_ol, R = s['rows_ol'], s['R']
        if self._use_c() and 'cb' in s:
            return self._c_shading_backward(a, s, pt, idx, dcolor_rm, flat, d_spec_raw, d_occ_raw, d_mat_raw)
        dMraw, dOLo, dILo, dIWo, dRLo, dNoV = e(P, 8), e(rows_ol, 4), e(2 * P, 4), e(P), e(P, 4), e(P)
        L.check(lib.nu_shade_combine_bwd(c_p(addr(s['Mraw'])), 8, c_p(addr(s['OLo'])), c_p(addr(s['ILo'])),
                                         c_p(addr(s['IWo'])), c_p(addr(s['RLo'])), c_p(addr(s['SD'])),
                                         c_p(addr(self.lut)), c_p(addr(idx)), P, c_f(self.exp_max),
                                         c_p(addr(dcolor_rm)), c_p(addr(dMraw)), c_p(addr(dOLo)), c_p(addr(dILo)),
                                         c_p(addr(dIWo)), c_p(addr(dRLo)), c_p(addr(dNoV)), S), "nu_shade_combine_bwd")
        if R:
            if d_spec_raw is not None:
                dOLo[3 * P:, :3] = d_spec_raw
                dOLo[3 * P:, 3] = 0
            else:
                dOLo[3 * P:].zero_()
        if d_occ_raw is not None:
            dIWo += d_occ_raw
        if d_mat_raw is not None:
            dMraw += d_mat_raw
        with self.wgrad_batch() as wb:        # the pass's weight gradients: one queue, launched together at the end of the block
            # heads + hidden stacks of the four light predictors
            ld_ol, ld_rl = self.ld_ol, self.ld_rl
            dOLin, dILin = e(rows_ol, ld_ol), e(2 * P, 128)
            for layers, Hs, dy, ldy, rows, no, X, ldx, dX, lddx, dxc in (
                    (self.outer_light, s['OLh'], dOLo, 4, rows_ol, 3, s['OLin'], ld_ol, dOLin, ld_ol, ld_ol),
                    (self.inner_light, s['ILh'], dILo, 4, 2 * P, 3, s['ILin'], 128, dILin, 128, 128),
                    (self.inner_weight, s['IWh'], dIWo, 1, P, 1, s['IWin'], 96, None, 0, 0),
                    (self.refrac_light, s['RLh'], dRLo, 4, P, 3, s['RLin'], ld_rl, None, 0, 0)):
                head = layers[3]
                dH3 = wb.keep(e(rows, 256))
                self.skinny_bwd(addr(dy), ldy, addr(Hs[2]), 256, rows, 256, addr(*head.Wp), 256, no, addr(dH3), 256, 1, 0,
                                addr(*head.dWp), head.ldd, addr(flat, head.db_off))
                self.relu_stack_bwd(layers, X, ldx, rows, Hs, dH3, flat, dX, lddx, dxc)
            dn = e(P, 3)
            L.check(lib.nu_shade_encode_bwd(c_p(addr(a['n'])), c_p(addr(pt)), 8, c_p(addr(s['SD'])), c_p(addr(dOLin)), ld_ol,
                                            1 if self.sphere_direction else 0, c_p(addr(dILin)), c_p(addr(dNoV)), P,
                                            c_p(addr(dn)), c_p(addr(dMraw)), 8, S), "nu_shade_encode_bwd")
            # materials backward
            db0, db12, db6 = self.mat_db
            dM3 = wb.keep(e(P, 1024))
            self.skinny_bwd(addr(dMraw), 8, addr(s['M3']), 1024, P, 1024, addr(self.Ws6), 1024, 6, addr(dM3), 1024, 1, 0,
                            addr(self.dWs6), 1024, addr(flat, db6))
            dA = dM3
            for j, Hin in ((2, s['M2']), (1, s['M1'])):
                self.wgrad(addr(dA), 1024, addr(Hin), 1024, P, 256, 256, addr(self.dWpM[j]), 256, addr(flat, db12[j]),
                           groups=4, sA0=256, sB0=256, sW=65536, sDb=256)
                nxt = wb.keep(e(P, 1024))
                self.nt(addr(dA), 1024, addr(self.WpTM[j]), 256, P, 256, 256, addr(nxt), 1024, EPI_MUL_DRELU,
                        H=addr(Hin), ldh=1024, groups=4, sA=256, sB=65536, sC=256, sH=256, mask=getattr(Hin, '_nu_mask', None))
                dA = nxt
            self.wgrad(addr(dA), 1024, addr(a['YX']), 288, P, 1024, 288, addr(self.dWpM0), 288, addr(flat, db0))
            dYX = e(P, 288)
            self.nt(addr(dA), 1024, addr(self.WpTM0), 1024, P, 288, 1024, addr(dYX), 288, EPI_PLAIN)
        return dYX, dn

    # ------------------------------------------------------------------ NeRF++ background
    def nerf_forward(self, pt, idx, P, alpha_rm, color_rm):
        """pt [P, 8] point records (x, dist, unit direction).  With alpha_rm = None the fused activation/scatter is skipped
        and the raw heads are returned in the dict (stage 2)."""
        lib, S = self.lib, self.stream()
        e = self.empty
        b = {'P': P}
        E4, U5, V = e(P, 96), e(P, 352), e(P, 288)
        if self._use_c():
            # H[1..4], H[6..8]: bf16 in the bf16-storage mode (H[5] = U5 carries the embedding)
            H = [E4] + [U5 if i == 4 else self.empty_h(P, 256) for i in range(8)]
            cb = NerfBufs(P=P)
            for i in range(9):
                cb.H[i] = addr(H[i])
                if i > 0:
                    cb.mask[i] = addr(self.relu_mask(H[i], P, 256))
            HV, sig, rgb = e(P, 128), e(P), e(P, 4)
            cb.V, cb.HV, cb.sig, cb.rgb = addr(V), addr(HV), addr(sig), addr(rgb)
            L.check(lib.nu_nerfpp_mlp_fwd(ctypes.byref(self._ctx), ctypes.byref(self._nerf_net), c_p(addr(pt)), pt.shape[1],
                                          ctypes.byref(cb), S), "nu_nerfpp_mlp_fwd")
            if alpha_rm is not None:
                L.check(lib.nu_nerf_act_fwd(c_p(addr(sig)), 1, c_p(addr(rgb)), 4, c_p(addr(pt)), c_p(addr(idx)), P,
                                            c_p(addr(alpha_rm)), c_p(addr(color_rm)), S), "nu_nerf_act_fwd")
            b.update(H=H, V=V, HV=HV, sig=sig, rgb=rgb, cb=cb)
            return b
        L.check(lib.nu_nerf_embed(c_p(addr(pt)), pt.shape[1], P, c_p(addr(E4)), c_p(addr(U5)), c_p(addr(V)), S), "nu_nerf_embed")
        H = [E4]
        src, lds = E4, 96
        for i in range(8):
            lay = self.nerf[i]
            dst = U5 if i == 4 else e(P, 256)
            ldc = 352 if i == 4 else 256
            self.nt(addr(src), lds, addr(*lay.Wp), lay.Kp, P, 256, lay.Kp, addr(dst), ldc, EPI_BIAS_RELU, bias=addr(lay.b),
                    mask=self.relu_mask(dst, P, 256))
            H.append(dst)
            src, lds = dst, ldc
        b['H'] = H  # H[i] = input of layer i (H[5] = U5, ld 352), H[8] = last hidden
        sig = e(P)
        self.skinny_fwd(addr(H[8]), 256, P, 256, addr(*self.nerf_alpha.Wp), 256, addr(self.nerf_alpha.b), 1, addr(sig), 1)
        self.nt(addr(H[8]), 256, addr(*self.nerf_feat.Wp), 256, P, 256, 256, addr(V), 288, EPI_BIAS_NONE,
                bias=addr(self.nerf_feat.b))
        HV = e(P, 128)
        self.nt(addr(V), 288, addr(*self.nerf_view.Wp), 288, P, 128, 288, addr(HV), 128, EPI_BIAS_RELU,
                bias=addr(self.nerf_view.b))
        rgb = e(P, 4)
        self.skinny_fwd(addr(HV), 128, P, 128, addr(*self.nerf_rgb.Wp), 128, addr(self.nerf_rgb.b), 3, addr(rgb), 4)
        if alpha_rm is not None:
            L.check(lib.nu_nerf_act_fwd(c_p(addr(sig)), 1, c_p(addr(rgb)), 4, c_p(addr(pt)), c_p(addr(idx)), P,
                                        c_p(addr(alpha_rm)), c_p(addr(color_rm)), S), "nu_nerf_act_fwd")
        b.update(V=V, HV=HV, sig=sig, rgb=rgb)
        return b

    def nerf_backward(self, b, pt, idx, dalpha_rm, dcolor_rm, flat, dsig=None, drgb=None, dx=None, ddir=None):
        """Backward of the NeRF++ stack.  Either (dalpha_rm, dcolor_rm) through the fused activation kernel (stage 1) or raw
        head cotangents (dsig [P], drgb [P,4]) directly (stage 2).  dx/ddir [P,3] (optional) receive the input gradients."""
        lib, S, P = self.lib, self.stream(), b['P']
        e = self.empty
        H = b['H']
        want_in = dx is not None
        if dsig is None:
            dsig, drgb = e(P), e(P, 4)
            L.check(lib.nu_nerf_act_bwd(c_p(addr(b['sig'])), 1, c_p(addr(b['rgb'])), 4, c_p(addr(pt)), c_p(addr(idx)), P,
                                        c_p(addr(dalpha_rm)), c_p(addr(dcolor_rm)), c_p(addr(dsig)), 1, c_p(addr(drgb)), 4, S),
                    "nu_nerf_act_bwd")
        if self._use_c():
            cb = b.get('cb')
            if cb is None:                 # forward was sequenced from Python
                cb = NerfBufs(P=P, V=addr(b['V']), HV=addr(b['HV']), sig=addr(b['sig']), rgb=addr(b['rgb']))
                for i in range(9):
                    cb.H[i] = addr(H[i])
                    if i > 0:
                        cb.mask[i] = addr(getattr(H[i], '_nu_mask', None))
            ldf = 288 if want_in else 256
            keep = [e(P, 128), e(P, ldf), self.empty_h(P, 256)]
            cb.dHV, cb.dF, cb.dH8a = (addr(t) for t in keep)
            for i in range(1, 9):
                t = e(P, 352 if want_in else 256) if i == 5 else self.empty_h(P, 256)
                keep.append(t)
                cb.dA[i] = addr(t)
            if want_in:
                t = e(P, 96)
                keep.append(t)
                cb.dE4, cb.dx, cb.ddir = addr(t), addr(dx), addr(ddir)
            else:
                cb.dE4 = cb.dx = cb.ddir = 0
            self._ensure_arena()
            self._ctx.flat = addr(flat)
            L.check(lib.nu_nerfpp_mlp_bwd(ctypes.byref(self._ctx), ctypes.byref(self._nerf_net), c_p(addr(pt)), pt.shape[1],
                                          ctypes.byref(cb), c_p(addr(dsig)), c_p(addr(drgb)), S), "nu_nerfpp_mlp_bwd")
            return
        with self.wgrad_batch() as wb:        # the pass's weight gradients: one queue, launched together at the end of the block
            # rgb head -> view layer
            dHV = wb.keep(e(P, 128))
            self.skinny_bwd(addr(drgb), 4, addr(b['HV']), 128, P, 128, addr(*self.nerf_rgb.Wp), 128, 3, addr(dHV), 128, 1, 0,
                            addr(*self.nerf_rgb.dWp), 128, addr(flat, self.nerf_rgb.db_off))
            self.wgrad(addr(dHV), 128, addr(b['V']), 288, P, 128, 288, addr(*self.nerf_view.dWp), 288,
                       addr(flat, self.nerf_view.db_off))
            ldf = 288 if want_in else 256
            dF = wb.keep(e(P, ldf))   # gradient w.r.t. V = feature_linear output (256) | view embedding (27, stage 2 only)
            self.nt(addr(dHV), 128, addr(*self.nerf_view.WpT), self.nerf_view.ldT, P, ldf, 128, addr(dF), ldf, EPI_PLAIN)
            self.wgrad(addr(dF), ldf, addr(H[8]), 256, P, 256, 256, addr(*self.nerf_feat.dWp), 256,
                       addr(flat, self.nerf_feat.db_off))
            # density head: dH8_alpha (masked by relu'(H8)), then add the feature path
            dH8a = wb.keep(e(P, 256))
            self.skinny_bwd(addr(dsig), 1, addr(H[8]), 256, P, 256, addr(*self.nerf_alpha.Wp), 256, 1, addr(dH8a), 256, 1, 0,
                            addr(*self.nerf_alpha.dWp), 256, addr(flat, self.nerf_alpha.db_off))
            dA = wb.keep(e(P, 256))
            self.nt(addr(dF), ldf, addr(*self.nerf_feat.WpT), self.nerf_feat.ldT, P, 256, 256, addr(dA), 256, EPI_B_RELU,
                    H=addr(H[8]), ldh=256, Cadd=addr(dH8a), ldadd=256, mask=getattr(H[8], '_nu_mask', None))
            # trunk, layers 7..0 ; dA = d pre-activation of layer i (row stride lda)
            dskip, lda = None, 256
            for i in range(7, -1, -1):
                lay = self.nerf[i]
                ldu = 96 if i == 0 else (352 if i == 5 else 256)
                self.wgrad(addr(dA), lda, addr(H[i]), ldu, P, 256, lay.Kp, addr(*lay.dWp), lay.ldd, addr(flat, lay.db_off))
                if i > 0:
                    if i == 5 and want_in:
                        # columns 256..339 of the layer-5 input are the re-concatenated embedding: keep their plain gradient
                        nxt = wb.keep(e(P, 352))
                        self.nt(addr(dA), lda, addr(*lay.WpT), lay.ldT, P, 340, 256, addr(nxt), 352, EPI_MUL_DRELU,
                                H=addr(H[i]), ldh=ldu, act_cols=256, zero_to=352, mask=getattr(H[i], '_nu_mask', None))
                        dskip = nxt
                        dA, lda = nxt, 352
                    else:
                        nxt = wb.keep(e(P, 256))
                        self.nt(addr(dA), lda, addr(*lay.WpT), lay.ldT, P, 256, 256, addr(nxt), 256, EPI_MUL_DRELU,
                                H=addr(H[i]), ldh=ldu, mask=getattr(H[i], '_nu_mask', None))
                        dA, lda = nxt, 256
                elif want_in:
                    dE4 = wb.keep(e(P, 96))
                    self.nt(addr(dA), lda, addr(*lay.WpT), lay.ldT, P, 84, 256, addr(dE4), 96, EPI_PLAIN, zero_to=96)
                    L.check(lib.nu_nerf_embed_bwd(c_p(addr(pt)), pt.shape[1], c_p(addr(H[0])), c_p(addr(b['V'])), c_p(addr(dE4)), 96,
                                                  c_p(addr(dskip, 256)), 352, c_p(addr(dF, 256)), ldf, P, c_p(addr(dx)),
                                                  c_p(addr(ddir)), S), "nu_nerf_embed_bwd")

    # ------------------------------------------------------------------ sampler (no grad)
    def _sampler_consts(self, Nc, Nbg, n_new):
        key = (Nc, Nbg, n_new)
        if getattr(self, '_sc_key', None) != key:
            lin = torch.linspace(0.0, 1.0, Nc)
            zo = torch.linspace(1e-3, 1.0 - 1.0 / (Nbg + 1.0), Nbg)
            mids = 0.5 * (zo[1:] + zo[:-1])
            upper = torch.cat([mids, zo[-1:]], -1)
            lower = torch.cat([zo[:1], mids], -1)
            uv = torch.linspace(0.5 / n_new, 1.0 - 0.5 / n_new, steps=n_new)
            self._sc = tuple(t.to(self.dev).contiguous() for t in (lin, lower, upper, zo, uv))
            self._sc_key = key
        return self._sc

    def sample_ray(self, o, d, near, far, perturb, u1=None, u2=None):
        """Hierarchical sampler (renderer_zerothick.py:572-612).  o, d [R,3]; near, far [R].  When perturb > 0 the
        two uniform draws are taken from (u1 [R], u2 [R,Nbg]) or drawn with torch.rand on the device."""
        cfg, lib, S = self.cfg, self.lib, self.stream()
        Nc, Nbg, Ni, steps = cfg['n_samples'], cfg['n_bg_samples'], cfg['n_importance'], cfg['up_sample_steps']
        R = o.shape[0]
        n_new = Ni // steps
        lin, lower, upper, zo_lin, uv = self._sampler_consts(Nc, Nbg, n_new)
        if perturb > 0:
            if u1 is None:
                u1 = torch.rand(R, 1, device=self.dev)
                u2 = torch.rand(R, Nbg, device=self.dev)
            u1, u2 = u1.contiguous(), u2.contiguous()
        e = self.empty
        z, zbg, X = e(R, Nc), e(R, Nbg), e(R * Nc, 3)
        L.check(lib.nu_sample_coarse(c_p(addr(o)), c_p(addr(d)), c_p(addr(near)), c_p(addr(far)), c_p(addr(lin)),
                                     c_p(addr(lower if perturb > 0 else zo_lin)), c_p(addr(upper)),
                                     c_p(addr(u1) if perturb > 0 else 0), c_p(addr(u2) if perturb > 0 else 0), R, Nc, Nbg,
                                     c_p(addr(z)), c_p(addr(zbg)), c_p(addr(X)), S), "nu_sample_coarse")
        sdf = self.sdf_forward(addr(X), 3, R * Nc, keep=False, want_feat=False)['sdf']
        sn = Nc
        var = self.p['deviation_network.variance']
        for i in range(steps):
            zn, Xn = e(R, n_new), e(R * n_new, 3)
            L.check(lib.nu_upsample(c_p(addr(o)), c_p(addr(d)), c_p(addr(z)), c_p(addr(sdf)), R, sn, c_p(addr(var)),
                                    c_f(64.0 * 2 ** i), 1 if cfg['clip_sample_variance'] else 0, c_p(addr(uv)), n_new,
                                    c_p(addr(zn)), c_p(addr(Xn)), S), "nu_upsample")
            last = i + 1 == steps
            sdf_n = None if last else self.sdf_forward(addr(Xn), 3, R * n_new, keep=False, want_feat=False)['sdf']
            zo, so = e(R, sn + n_new), (None if last else e(R, sn + n_new))
            L.check(lib.nu_merge_sorted(c_p(addr(z)), c_p(addr(sdf)), sn, c_p(addr(zn)), c_p(addr(sdf_n)), n_new, R,
                                        c_p(addr(zo)), c_p(addr(so)), S), "nu_merge_sorted")
            z, sdf, sn = zo, so, sn + n_new
        out = e(R, sn + Nbg)
        L.check(lib.nu_concat_cols(c_p(addr(z)), sn, c_p(addr(zbg)), Nbg, R, c_p(addr(out)), S), "nu_concat_cols")
        return out

    # ------------------------------------------------------------------ render_core
    def _render_forward_inner(self, ctx, out, o, d, P_in, pt_in, idx_in, alpha_rm, color_rm, anneal, spec_pts):
        """The inner-point chain of render_forward (SDF, normal, NeuS alpha, shading) on the caller's stream."""
        lib, S_, e = self.lib, self.stream(), self.empty
        # unit ray directions for the per-ray mirror query (dirs[:,0,:] in the reference)
        du = torch.nn.functional.normalize(d, dim=-1).contiguous()
        if P_in > 0:
            a = self.sdf_forward(addr(pt_in), 8, P_in, keep=True)
            self.sdf_normal(a)
            if self.occ_sdf_thresh is not None:
                # the occlusion loss's point list (renderer_zerothick.py:699-707) needs only x, the view direction, the SDF and its
                # normal: its nonzero() -- a host wait -- is taken HERE, with the NeRF++ chain already queued and the shading stack
                # about to be, instead of after the forward, where the GPU then sat waiting for the loss and backward launches
                x_, dirs_ = pt_in[:P_in, :3], pt_in[:P_in, 4:7]
                m_ = (torch.norm(x_, dim=-1) < 0.999) & (torch.sum(a['n'] * dirs_, -1) < 0) & (torch.abs(a['YX'][:, 0]) < self.occ_sdf_thresh)
                ctx['occ_idx'] = torch.nonzero(m_)[:, 0]
            gerr = e(P_in)
            var = self.p['deviation_network.variance']
            L.check(lib.nu_neus_alpha_fwd(c_p(addr(a['YX'])), 288, c_p(addr(a['n'])), c_p(addr(pt_in)), c_p(addr(idx_in)),
                                          P_in, c_p(addr(var)), c_f(anneal), c_p(addr(alpha_rm)), c_p(addr(gerr)),
                                          c_p(addr(color_rm)), S_), "nu_neus_alpha_fwd")
            s = self.shading_forward(a, pt_in, idx_in, P_in, color_rm, extra_dirs=du, extra_pts=spec_pts)
            ctx.update(sdf=a, shade=s)
            out['gradient_error'] = gerr
            out['spec_raw'] = s['OLo'][3 * P_in:, :3]
            out['occ_raw'] = s['IWo']
            out['sdf_in'] = a['YX'][:, 0]
            out['aux'] = s['aux']
            out['normal_raw'] = a['n']

    def render_forward(self, o, d, z, anneal, want_weights=False, spec_pts=None):
        """Stage-1 render_core forward (renderer_zerothick.py:725-820) on R rays with S samples each.
        Returns (outputs dict of tensors, ctx) ; one host sync (the inner-point count)."""
        lib, S_ = self.lib, self.stream()
        e = self.empty
        R, S = z.shape
        o, d, z = o.contiguous(), d.contiguous(), z.contiguous()
        cnt, off, tot = e(R, dtype=torch.int32), e(R, dtype=torch.int32), e(2, dtype=torch.int32)
        L.check(lib.nu_partition_count(c_p(addr(o)), c_p(addr(d)), c_p(addr(z)), R, S, c_p(addr(cnt)), c_p(addr(off)),
                                       c_p(addr(tot)), S_), "nu_partition_count")
        P_in = int(tot[0].item())
        P_out = R * S - P_in
        pt_in, idx_in = e(max(P_in, 1), 8), e(max(P_in, 1), dtype=torch.int32)
        pt_out, idx_out = e(max(P_out, 1), 8), e(max(P_out, 1), dtype=torch.int32)
        inner_rm = e(R * S, dtype=torch.uint8)
        L.check(lib.nu_partition_write(c_p(addr(o)), c_p(addr(d)), c_p(addr(z)), R, S, c_p(addr(off)), c_p(addr(pt_in)),
                                       c_p(addr(idx_in)), c_p(addr(pt_out)), c_p(addr(idx_out)), c_p(addr(inner_rm)), S_),
                "nu_partition_write")
        alpha_rm, color_rm = e(R * S), e(R * S, 4)
        ctx = dict(R=R, S=S, P_in=P_in, P_out=P_out, P_in_dev=tot[:1], pt_in=pt_in, idx_in=idx_in, pt_out=pt_out, idx_out=idx_out,
                   inner_rm=inner_rm, alpha_rm=alpha_rm, color_rm=color_rm, anneal=float(anneal))
        # (the bf16-MFMA modes stay on one stream: a packed-fp32 VALU kernel running beside this library's bf16-MFMA GEMMs returns wrong
        # elements now and then -- 'bf16x6' gradients were not reproducible run to run with the overlap; the fp32-MFMA kernels of the
        # default mode do not trigger it.  DESIGN.md 12, scripts/determinism_valu_victim.py, scripts/determinism_probe3.py)
        two = (P_out > 0 and P_in > 0 and R * S <= self._TWO_STREAM_SAMPLES
               and (self.bf16 == 0 or os.environ.get('NU_BF16_TWO_STREAMS') == '1'))
        ctx['two_streams'] = two
        out = {}
        fk = self.forked() if two else None
        side = fk.__enter__() if two else None
        try:
            if two:
                with torch.cuda.stream(side):
                    ctx['nerf'] = self.nerf_forward(pt_out, idx_out, P_out, alpha_rm, color_rm)
            elif P_out > 0:
                ctx['nerf'] = self.nerf_forward(pt_out, idx_out, P_out, alpha_rm, color_rm)
            self._render_forward_inner(ctx, out, o, d, P_in, pt_in, idx_in, alpha_rm, color_rm, anneal, spec_pts)
        finally:
            if two:
                fk.__exit__(None, None, None)
        weights = e(R, S) if want_weights else None
        rgb, acc, rgb_bg, nrm_sum = e(R, 3), e(R), e(R, 3), e(R)
        L.check(lib.nu_composite_fwd(c_p(addr(alpha_rm)), c_p(addr(color_rm)), c_p(addr(inner_rm)), R, S,
                                     c_p(addr(weights)), c_p(addr(rgb)), c_p(addr(acc)), c_p(addr(rgb_bg)),
                                     c_p(addr(nrm_sum)), S_), "nu_composite_fwd")
        out.update(rgb=rgb, acc=acc, rgb_bg=rgb_bg, weights=weights, nrm_sum=nrm_sum)
        self.last_ctx = ctx
        return out, ctx

    def render_backward(self, ctx, d_rgb, d_acc, d_rgb_bg, d_gerr=None, d_spec_raw=None, d_occ_raw=None, d_sdf_in=None,
                        train_inv_s=False, d_nrm_sum=None, d_trans=None, d_metal=None):
        """Hand-derived backward of render_forward w.r.t. every network parameter.  Returns the flat gradient
        buffer (layout: self.grad_views)."""
        lib, S_ = self.lib, self.stream()
        e = self.empty
        R, S, P_in, P_out = ctx['R'], ctx['S'], ctx['P_in'], ctx['P_out']
        flat = self.zeros(self.n_grad)
        dalpha_rm, dcolor_rm = e(R * S), e(R * S, 4)
        # contiguous copies are named locals: they must outlive the launches that read them (a temporary inside the argument
        # list dies as soon as addr() returns and its block may be handed to the next allocation)
        d_rgb, d_acc, d_rgb_bg, d_nrm_sum, d_gerr = (None if t is None else t.contiguous()
                                                     for t in (d_rgb, d_acc, d_rgb_bg, d_nrm_sum, d_gerr))
        L.check(lib.nu_composite_bwd(c_p(addr(ctx['alpha_rm'])), c_p(addr(ctx['color_rm'])), c_p(addr(ctx['inner_rm'])),
                                     R, S, c_p(addr(d_rgb)), c_p(addr(d_acc)), c_p(addr(d_rgb_bg)), c_p(addr(d_nrm_sum)),
                                     c_p(addr(dalpha_rm)), c_p(addr(dcolor_rm)), S_), "nu_composite_bwd")
        two = ctx.get('two_streams', False)
        fk = self.forked() if two else None
        side = fk.__enter__() if two else None
        try:
            self._render_backward_inner(ctx, two, side, flat, dalpha_rm, dcolor_rm, d_gerr, d_spec_raw, d_occ_raw, d_sdf_in, train_inv_s,
                                        d_nrm_sum, d_trans, d_metal)
        finally:
            if two:
                fk.__exit__(None, None, None)
        self.unpack_grads(flat)
        return flat

    def _render_backward_inner(self, ctx, two, side, flat, dalpha_rm, dcolor_rm, d_gerr, d_spec_raw, d_occ_raw, d_sdf_in, train_inv_s,
                               d_nrm_sum, d_trans, d_metal):
        lib, S_, e = self.lib, self.stream(), self.empty
        R, S, P_in, P_out = ctx['R'], ctx['S'], ctx['P_in'], ctx['P_out']
        if two:
            with torch.cuda.stream(side):
                self.nerf_backward(ctx['nerf'], ctx['pt_out'], ctx['idx_out'], dalpha_rm, dcolor_rm, flat)
        elif P_out > 0:
            self.nerf_backward(ctx['nerf'], ctx['pt_out'], ctx['idx_out'], dalpha_rm, dcolor_rm, flat)
        else:
            self.zero_stale_wgrads(self.nerf + [self.nerf_feat, self.nerf_alpha, self.nerf_view, self.nerf_rgb])
        if P_in == 0:
            self.zero_stale_wgrads(self.sdf + self.mat_layers + self.outer_light + self.inner_light +
                                   self.inner_weight + self.refrac_light)
        if P_in > 0:
            a, s = ctx['sdf'], ctx['shade']
            d_mat = None
            if d_trans is not None or d_metal is not None:
                # cotangents of the post-sigmoid transmission weight / metallic (the registry's TransmissionRegLoss and
                # MetallicRegLoss): d raw = d y * y (1 - y), added to the raw material heads' gradient (columns 5 and 0 of Mraw)
                aux = s['aux']
                d_mat = torch.zeros(P_in, 8, device=aux.device)
                if d_trans is not None:
                    t = aux[:, 1]
                    d_mat[:, 5] = d_trans.reshape(-1) * t * (1.0 - t)
                if d_metal is not None:
                    m = aux[:, 2]
                    d_mat[:, 0] = d_metal.reshape(-1) * m * (1.0 - m)
            dYX, dn = self.shading_backward(a, s, ctx['pt_in'], ctx['idx_in'], dcolor_rm, flat,
                                            d_spec_raw=d_spec_raw, d_occ_raw=d_occ_raw, d_mat_raw=d_mat)
            nbar = e(P_in, 3)
            var = self.p['deviation_network.variance']
            ws = nb = 0
            if train_inv_s:        # per-block partials of d variance: a slab of the reduction arena, summed with the weight gradients
                if self._ctx.ndesc + 1 > self._rd_cap:
                    self._forced_flush()
                ws, nb = self._arena_take(lib.nu_neus_alpha_bwd_workspace_bytes(P_in))
            L.check(lib.nu_neus_alpha_bwd(c_p(addr(a['YX'])), 288, c_p(addr(a['n'])), c_p(addr(ctx['pt_in'])),
                                          c_p(addr(ctx['idx_in'])), P_in, c_p(addr(var)), c_f(ctx['anneal']),
                                          c_p(addr(dalpha_rm)), c_p(addr(d_gerr)),
                                          c_p(addr(dn)), c_p(addr(dcolor_rm) if d_nrm_sum is not None else 0), c_p(addr(dYX)), 288,
                                          c_p(addr(nbar)),
                                          c_p(addr(flat, self.var_off) if train_inv_s else 0), c_p(ws), c_ll(nb), self._rd, self._ndesc_p,
                                          self._rd_cap, S_), "nu_neus_alpha_bwd")
            if d_sdf_in is not None:
                dYX[:, 0] += d_sdf_in
            self.sdf_backward(a, dYX, nbar, flat)

    # ------------------------------------------------------------------ occlusion probe (no grad)
    def occ_probe(self, pts, dirs, sn0=64, sn1=16):
        """Hit probability of secondary rays inside the unit sphere: get_intersection + get_weights + sample_pdf
        (field.py:501-554) for points already known to satisfy |x| < 0.999.  Returns occ_prob_gt [M]."""
        lib, S = self.lib, self.stream()
        e = self.empty
        M = pts.shape[0]
        pts, dirs = pts.contiguous(), dirs.contiguous()
        dtx = torch.sum(pts * dirs, dim=-1)
        xtx = torch.sum(pts ** 2, dim=-1)
        max_dist = (-dtx + torch.sqrt(dtx ** 2 - xtx + 1 + 1e-6)).contiguous()
        key = ('probe', sn0, sn1)
        if getattr(self, '_pc_key', None) != key:
            self._pc = (torch.linspace(0, 1, sn0).to(self.dev),
                        torch.linspace(0.5 / sn1, 1.0 - 0.5 / sn1, steps=sn1).to(self.dev))
            self._pc_key = key
        lin, uv = self._pc
        zero = self.zeros(M)
        z, X = e(M, sn0), e(M * sn0, 3)
        L.check(lib.nu_sample_coarse(c_p(addr(pts)), c_p(addr(dirs)), c_p(addr(zero)), c_p(addr(max_dist)), c_p(addr(lin)),
                                     c_p(0), c_p(0), c_p(0), c_p(0), M, sn0, 0, c_p(addr(z)), c_p(0), c_p(addr(X)), S),
                "nu_sample_coarse")
        sdf = self.sdf_forward(addr(X), 3, M * sn0, keep=False, want_feat=False)['sdf']
        var = self.p['deviation_network.variance']
        zn, Xn = e(M, sn1), e(M * sn1, 3)
        L.check(lib.nu_probe_weights(c_p(addr(pts)), c_p(addr(dirs)), c_p(addr(z)), c_p(addr(sdf)), M, sn0, c_p(addr(var)),
                                     c_p(addr(uv)), sn1, c_p(addr(zn)), c_p(addr(Xn)), c_p(0), S), "nu_probe_weights")
        sdf2 = self.sdf_forward(addr(Xn), 3, M * sn1, keep=False, want_feat=False)['sdf']
        wsum = e(M)
        L.check(lib.nu_probe_weights(c_p(addr(pts)), c_p(addr(dirs)), c_p(addr(zn)), c_p(addr(sdf2)), M, sn1,
                                     c_p(addr(var)), c_p(0), 0, c_p(0), c_p(0), c_p(addr(wsum)), S), "nu_probe_weights")
        return wsum

    def zero_stale_wgrads(self, which):
        """When a point set is empty its weight-gradient GEMMs do not run: clear their packed outputs."""
        seen = set()
        for lay in which:
            t = lay.dWp[0]
            if id(t) not in seen:
                t.zero_()
                seen.add(id(t))
