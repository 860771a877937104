"""Thin tensor-level wrappers over the C-ABI (include/semivl_hip.h).

torch is used here for device memory (torch.empty), views and the current HIP stream only; every FLOP of the hot
path is issued through libsemivl_hip.so.  All tensors are fp32 CUDA(HIP) tensors unless stated.
"""
import collections
import ctypes as C
import os
import math
import weakref

import torch

from . import lib as L

A_KC, A_MC, A_CONV, A_PATCH = 0, 1, 2, 3
B_KC, B_NC, B_CONVW = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_RELU, ACT_MUL_DGELU, ACT_MUL_DRELU = 0, 1, 2, 3, 4
OUT_STRIDED, OUT_CONVT2X, OUT_PATCH = 0, 1, 2


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional live kernel timing (bench.py): PROFILE = {} enables HIP-event brackets around the launches of the kernel
# families we report rooflines for.  Events are recorded on the launch stream (torch's current stream).
PROFILE = None
PROFILE_SCOPE = None  # name of the sub-path whose launches are being recorded ("vit" inside the encoder regions)


class prof_scope:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global PROFILE_SCOPE
        self.prev, PROFILE_SCOPE = PROFILE_SCOPE, self.name

    def __exit__(self, *exc):
        global PROFILE_SCOPE
        PROFILE_SCOPE = self.prev


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(name, e0, work, tag=None):
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROFILE.setdefault(name, []).append((e0, e1, work, tag, PROFILE_SCOPE))


def _p(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def _chk(t, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype, (t.device, t.dtype)
    return t


def empty(*shape, dtype=torch.float32, device=None):
    return torch.empty(*shape, dtype=dtype, device=device if device is not None else torch.cuda.current_device())


def zeros(*shape, dtype=torch.float32, device=None):
    """torch.zeros on the library's fill kernel (fp32, or 4 / 8-byte integer types filled through an fp32 view)."""
    t = torch.empty(*shape, dtype=dtype, device=device if device is not None else torch.cuda.current_device())
    if t.numel():
        if not t.is_cuda or t.element_size() % 4 != 0:
            return t.zero_()
        f = t.view(-1).view(torch.float32)
        L.check(L.load().svl_fill_f32(C.c_void_p(f.data_ptr()), 0.0, f.numel(), _st()), "svl_fill_f32")
    return t


def permute4(src, shape, strides):
    """A contiguous fp32 tensor of `shape` (4 dims) read from `src` through the element `strides` (svl_permute4_f32): the
    weight-sized permutes between parameter layouts and kernel packs without an ATen copy kernel."""
    assert src.dtype == torch.float32 and len(shape) == 4 and len(strides) == 4
    out = torch.empty(*shape, dtype=torch.float32, device=src.device)
    if not src.is_cuda:
        return out.copy_(torch.as_strided(src, shape, strides, src.storage_offset()))
    L.check(L.load().svl_permute4_f32(_p(src), _p(out), *[int(v) for v in shape], *[int(v) for v in strides], _st()),
            "svl_permute4_f32")
    return out


class Op:
    """Operand description: tensor (+ element offset), leading dim, batch strides."""
    __slots__ = ("t", "off", "ld", "bso", "bsi")

    def __init__(self, t, ld, off=0, bso=0, bsi=0):
        self.t, self.off, self.ld, self.bso, self.bsi = t, off, ld, bso, bsi

    def c(self):
        return L.Operand(C.c_void_p(self.t.data_ptr() + 4 * self.off), self.ld, self.bso, self.bsi)


# In-register split kernel on fp16 x 2 terms with per-tensor scales (svl_gemm_desc.emu_ws; round 5): SVL_GEMM_EMU_NO_H2=1 keeps
# the bf16 x 3 form for every launch (A/B runs).
EMU_H2 = not os.environ.get("SVL_GEMM_EMU_NO_H2")
# conv_fwd launches on the in-register fp16 x 2 form (implicit-GEMM convolutions the tiled / whole-image kernels do not take).
# Round 5 ended with them OFF (the float64 gate of tests/test_fullsize_gpu.py at 4.0 - 4.4 x); since the split-K slab sums run in
# double and the gate compares under the product's own tie decisions they are ON (mode-6 families <= 2.5 x).
EMU_H2_CONVFWD = True


def gemm(a_mode, b_mode, M, N, K, A, B, Cout, c_off=0, ldc_m=None, ldc_n=1, batch=1, batch_inner=1, ksplit=0,
         c_bso=0, c_bsi=0, alpha=1.0, bias=None, bias_mod=0, act=ACT_NONE, resid=None, r_off=0, ldr_m=None, ldr_n=1,
         r_bso=0, r_bsi=0, accumulate=False, out_mode=OUT_STRIDED, conv=None, ct=(0, 0, 0), preact=None, w_planes=None,
         emu_h2=True):
    d = L.GemmDesc()
    d.conv_w_planes = _p(w_planes)
    d.a_mode, d.b_mode, d.M, d.N, d.K = a_mode, b_mode, M, N, K
    d.batch, d.batch_inner, d.ksplit = batch, batch_inner, ksplit
    d.A, d.B = A.c(), B.c()
    if conv is not None:
        d.conv = conv
    d.C = C.c_void_p(Cout.data_ptr() + 4 * c_off)
    d.out_mode = out_mode
    d.ldc_m = ldc_m if ldc_m is not None else N
    d.ldc_n, d.c_bs_outer, d.c_bs_inner = ldc_n, c_bso, c_bsi
    d.ct_H, d.ct_W, d.ct_Cout = ct
    d.alpha = alpha
    d.bias = _p(bias)
    d.bias_mod, d.act = bias_mod, act
    d.preact = _p(preact)
    if resid is not None:
        d.resid = C.c_void_p(resid.data_ptr() + 4 * r_off)
        d.ldr_m = ldr_m if ldr_m is not None else d.ldc_m
        d.ldr_n, d.r_bs_outer, d.r_bs_inner = ldr_n, r_bso, r_bsi
    d.accumulate = 1 if accumulate else 0
    ws = None
    if (EMU_H2 and emu_h2 and 2.0 * M * N * K >= 4.0e9 and (batch == 1 or ksplit > 0)
            and Cout.is_cuda and get_gemm_emulation() == 6):
        ws = torch.empty(2, dtype=torch.int32, device=Cout.device)      # scratch of the operand-maximum passes (fp16 x 2 form)
        d.emu_ws = _p(ws)
    e0 = _prof_begin()
    L.check(L.load().svl_gemm_f32(C.byref(d), _st()), "svl_gemm_f32")
    if e0 is not None:
        # which matrix pipe served this launch: the library reports the kernel family its dispatch chose
        # (svl_last_gemm_path: 1 = bf16 split products, 4 = fp16 x 2 split products; 0 / 2 / 3 = exact fp32 MFMA, short-K
        # stream, elementwise)
        path = L.load().svl_last_gemm_path()
        _prof_end("gemm_bf16x" if path in (1, 4) else "gemm", e0, 2.0 * M * N * K * (1 if ksplit > 0 else batch),
                  (a_mode, b_mode, M, N, K, batch) if path != 4 else ("split_h2", a_mode, b_mode, M, N, K, batch))


def conv_geom(H, W, C1, KH, KW, dil=1, pad=0, sign=1, C2=0, rep=1, src2=None, ld2=0, patch=0, stride=1, Ho=0, Wo=0):
    g = L.ConvGeom()
    g.H, g.W, g.C1, g.C2, g.rep = H, W, C1, C2, rep
    g.stride, g.Ho, g.Wo = stride, Ho, Wo
    g.KH, g.KW, g.dil, g.pad, g.sign = KH, KW, dil, pad, sign
    g.src2 = _p(src2)
    g.ld2, g.patch = ld2, patch
    return g


# ------------------------------------------------------------------------------------------------ weight-gradient stream
# Backward has one dependency chain (the input gradients) and a lot of work OFF it: the weight-gradient GEMMs / convs, their
# split-K reductions and the bias column sums.  On one stream they queue between the chain's bandwidth-bound passes
# (LayerNorm / GroupNorm backward, packs, resampling), which then run with the matrix pipe idle: 83 ms of a 490 ms step
# had no MFMA kernel in flight (tools/rocpd_attrib.py).  With the off-chain work on a second stream the chip has matrix
# work to run next to those passes.  `with wgrad_side(t1, t2, ...)`: the enclosed launches go to the per-device
# weight-gradient stream, ordered after everything queued so far on the current stream; the listed tensors (produced on
# the current stream, read in the block) are kept alive until the block has run (below).  wgrad_join() orders the current
# stream after the side stream (before anything reads the gradients: all-reduce, optimizer).
# Lifetime of the operands: the block's tensors are NOT handed to the allocator with record_stream (a block released on the
# current stream would then stay unusable until the side stream has caught up -- with the multi-GB decoder tensors of the
# N = 81 / 150 configs the side stream's lag piled up tens of GB of such blocks, the allocator ran into the device limit and
# fell into its free-everything-and-retry path: 1.6 s -> 6.9 s per ADE step).  Instead every block leaves (event, tensors)
# in a short queue; once more than WGRAD_DEPTH blocks are outstanding the current stream WAITS for the oldest one's event
# and only then drops the references: the side stream is never more than WGRAD_DEPTH blocks behind, and a released block
# is reusable at once (the current stream is ordered after its last reader).
# Default: on for a single-GPU process; OFF for the ranks of a multi-GPU job (WORLD_SIZE > 1), where every stream owns a
# hardware queue (semivl_amd/__init__.py raises GPU_MAX_HW_QUEUES to 8 so that the communication stream does not share a
# queue with the compute it overlaps) and weight-gradient GEMMs truly concurrent with the chain's GEMMs cost 20 ms per step
# (DESIGN §9).  SVL_WGRAD_STREAM=1 / SVL_NO_WGRAD_STREAM=1 force either setting; cfg["wgrad_stream"] overrides per step.
def _env_flag(name):
    """An environment switch parsed as an integer (`NAME=0` is off, unlike bool("0"))."""
    v = os.environ.get(name, "").strip()
    if not v:
        return False
    try:
        return int(v) != 0
    except ValueError:
        return v.lower() not in ("false", "no", "off")


WGRAD_STREAM = (_env_flag("SVL_WGRAD_STREAM") or int(os.environ.get("WORLD_SIZE", "1")) <= 1) \
    and not _env_flag("SVL_NO_WGRAD_STREAM")
WGRAD_DEPTH = 2          # blocks the weight-gradient stream may lag behind the chain
_WG = {}
_WG_KEEP = {}   # device index -> deque of (event on the side stream, tensors read before it, block number)
_WG_SEQ = 0     # blocks closed so far


def _wg_stream(dev):
    s = _WG.get(dev.index)
    if s is None:
        # (stream priority was tried: this runtime offers only {high, normal}; a HIGH-priority weight-gradient stream delays
        # the dependency chain, 449 -> 494 ms, so it stays at normal priority)
        s = torch.cuda.Stream(dev)
        _WG[dev.index] = s
        _WG_KEEP[dev.index] = collections.deque()
    return s


def _wg_retire(main, keep_at_most):
    q = _WG_KEEP.get(main.device.index)
    while q and len(q) > keep_at_most:
        ev = q.popleft()[0]
        main.wait_event(ev)


class wgrad_side:
    def __init__(self, *tensors):
        self.tensors, self.ctx, self.main, self.wg = tensors, None, None, None

    def __enter__(self):
        if not WGRAD_STREAM or not torch.cuda.is_available():
            return self
        main = torch.cuda.current_stream()
        wg = _wg_stream(main.device)
        if wg.cuda_stream == main.cuda_stream:
            return self
        ev = torch.cuda.Event()
        ev.record(main)
        wg.wait_event(ev)
        self.main, self.wg = main, wg
        self.ctx = torch.cuda.stream(wg)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            ev = torch.cuda.Event()
            ev.record(self.wg)
            global _WG_SEQ
            _WG_SEQ += 1
            _WG_KEEP[self.main.device.index].append(
                (ev, [t for t in self.tensors if isinstance(t, torch.Tensor) and t.is_cuda], _WG_SEQ))
            _wg_retire(self.main, WGRAD_DEPTH)
            self.tensors = ()
        return False


def wgrad_event():
    """An event after everything queued so far on the weight-gradient stream (None when it was never used)."""
    if not torch.cuda.is_available():
        return None
    wg = _WG.get(torch.cuda.current_device())
    if wg is None:
        return None
    ev = torch.cuda.Event()
    ev.record(wg)
    ev._svl_seq = _WG_SEQ      # every block closed so far is ordered before this event
    return ev


def streams_share_queue(x, y, busy_ms=3.0):
    """Do streams x and y sit on the same hardware queue?  The runtime multiplexes a process's streams onto
    GPU_MAX_HW_QUEUES queues and serialises streams that share one: a `busy_ms` single-wave kernel goes to x,
    a tiny one to y ordered only after an event recorded BEFORE the long one -- if y's kernel finishes after x's they
    share a queue.  Costs `busy_ms` of device time and a synchronize: construction-time probe, never in the step."""
    lib = L.load()
    dev = x.device
    buf = torch.zeros(4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(x)
    L.check(lib.svl_clock_probe(C.c_void_p(buf.data_ptr()), 1, int(busy_ms * 1e5), C.c_void_p(x.cuda_stream)), "svl_clock_probe")
    y.wait_event(e0)
    L.check(lib.svl_clock_probe(C.c_void_p(buf[2:].data_ptr()), 1, 100, C.c_void_p(y.cuda_stream)), "svl_clock_probe")
    e1.record(y)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) > 0.66 * busy_ms


def wgrad_join(ev=None, produced=()):
    """Order the current stream after `ev` (wgrad_event()) or after everything queued on the weight-gradient stream.
    `produced`: tensors allocated inside wgrad_side blocks that the current stream goes on to use (gradients handed back to
    autograd when there is no main_grad arena): marked in use here for the caching allocator."""
    if not torch.cuda.is_available():
        return
    main = torch.cuda.current_stream()
    if ev is not None:
        main.wait_event(ev)
        q, seq = _WG_KEEP.get(main.device.index), getattr(ev, "_svl_seq", -1)
        while q and q[0][2] <= seq:      # blocks the waited event retires: their operands need not be held any longer
            q.popleft()
        return
    wg = _WG.get(main.device.index)
    if wg is not None and wg.cuda_stream != main.cuda_stream:
        main.wait_stream(wg)
        _WG_KEEP[main.device.index].clear()
        for t in produced:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(main)


# ------------------------------------------------------------------------------------------------ pre-split bf16x3 operands
def planes_rows(rows):
    return (rows + 255) // 256 * 256


# Operand format of the packed-planes GEMM: "h2" = fp16 x 2 planes + one power-of-two scale per row, three cross products
# (round 5: half the matrix work and 2/3 of the bytes of "b3" at the same error level against fp64); "b3" = bf16 x 3 planes,
# six products.  SVL_PLANES_FMT=b3 keeps round 4's form for A/B runs; the fused attention kernels emit b3 either way (their
# consumers -- out-projection, in_proj input gradient -- take the format of the A operand they are given).
PLANES_FMT = os.environ.get("SVL_PLANES_FMT", "h2")
assert PLANES_FMT in ("h2", "b3")


class Planes:
    """An fp32 matrix [rows, K] in the fragment-packed operand format of svl_gemm_planes_f32 (1 KiB chunks per (k-group,
    32-row block, plane); include/semivl_hip.h).  fmt "b3": three bf16 planes x = x0 + x1 + x2.  fmt "h2": two fp16 planes
    and one scale exponent per row, x = 2^sexp[row] (h0 + h1); `rnorm` (optional) = upper bounds of the rows' L2 norms,
    which a GEMM needs to scale a planes OUTPUT in this format.  The buffer holds `prow` = rows rounded up to 256 rows."""
    __slots__ = ("buf", "rows", "K", "prow", "fmt", "sexp", "rnorm", "_bd")

    def __init__(self, rows, K, device=None, buf=None, fmt=None, sexp=None, rnorm=None):
        assert K % 16 == 0
        self.fmt = fmt or PLANES_FMT
        self.rows, self.K, self.prow = rows, K, planes_rows(rows)
        dev = device if device is not None else (buf.device if buf is not None else torch.cuda.current_device())
        self._bd = None
        if self.fmt == "h2":
            self.buf = buf if buf is not None else torch.empty(K // 16 * self.prow * 32, dtype=torch.float16, device=dev)
            self.sexp = sexp if sexp is not None else torch.empty(self.prow, dtype=torch.int32, device=dev)
            self.rnorm = rnorm
        else:
            self.buf = buf if buf is not None else torch.empty(K // 16 * self.prow * 48, dtype=torch.bfloat16, device=dev)
            self.sexp = self.rnorm = None

    @property
    def shape(self):
        return (self.rows, self.K)

    def kslice(self, k0, k1):
        """Columns [k0, k1) (multiples of 16) as a Planes view: k-groups are the outermost index of the layout (the row
        scales of an h2 buffer belong to whole rows and stay valid; the row norms stay upper bounds)."""
        assert k0 % 16 == 0 and k1 % 16 == 0 and 0 <= k0 < k1 <= self.K
        per = self.prow * (32 if self.fmt == "h2" else 48)
        return Planes(self.rows, k1 - k0, buf=self.buf[k0 // 16 * per:k1 // 16 * per], fmt=self.fmt, sexp=self.sexp,
                      rnorm=self.rnorm)


def split_planes(x2d, out=None, row_off=0, transpose=False, fmt=None):
    """fp32 [rows, K] -> Planes (one HBM pass: 4 B read + 6 / 4 B written per element; the h2 pass reads a 32-row block
    twice, the second time from L2, to find the row scales).  transpose=True splits x2d^T (used once per weight for the
    input-gradient GEMMs)."""
    assert x2d.dim() == 2 and x2d.dtype == torch.float32
    if transpose:
        K, rows = x2d.shape
        assert x2d.stride(1) == 1
        ld, ks = 1, x2d.stride(0)
    else:
        rows, K = x2d.shape
        assert x2d.stride(1) == 1
        ld, ks = x2d.stride(0), 1
    if out is None:
        out = Planes(rows, K, device=x2d.device, fmt=fmt)
    assert out.K == K and out.rows >= row_off + rows and row_off % 32 == 0
    if out.fmt == "h2":
        if out.rnorm is None:
            out.rnorm = torch.empty(out.prow, dtype=torch.float32, device=x2d.device)
        L.check(L.load().svl_split_planes_f16x2(_p(x2d), ld, ks, rows, K, _p(out.buf), out.prow, row_off, _p(out.sexp),
                                                _p(out.rnorm), _st()), "svl_split_planes_f16x2")
        return out
    L.check(L.load().svl_split_planes_bf16x3(_p(x2d), ld, ks, rows, K, _p(out.buf), out.prow, row_off, _st()),
            "svl_split_planes_bf16x3")
    return out


WEIGHT_EPOCH = 0          # bumped by FusedAdamW.step(): its kernel rewrites the parameter arena behind torch's back
_WPLANES = {}


class StreamCached:
    """A cached device value (tensor, tuple of tensors, Planes) together with the event that completes it on the stream
    that BUILT it.  The step runs its gradient-free passes on a second stream (train.py): whichever stream first needs a
    re-laid-out weight builds it, and a later hit from the OTHER stream must (1) wait for that build and (2) tell the
    caching allocator that the block is in use there -- without this a lagging side stream leaves the main stream
    convolving with a half-written (or, after an optimizer step, stale) pack."""
    __slots__ = ("val", "ev", "seen")

    def __init__(self, val):
        self.val, self.ev, self.seen = val, None, None
        if torch.cuda.is_available():
            s = torch.cuda.current_stream()
            self.ev = torch.cuda.Event()
            self.ev.record(s)
            self.seen = {s.cuda_stream}

    @staticmethod
    def _tensors(v):
        if isinstance(v, torch.Tensor):
            yield v
        elif isinstance(v, (tuple, list)):
            for x in v:
                yield from StreamCached._tensors(x)
        elif hasattr(v, "buf"):
            yield v.buf
            for extra in (getattr(v, "sexp", None), getattr(v, "rnorm", None)):
                if extra is not None:
                    yield extra

    def get(self):
        if self.ev is not None:
            s = torch.cuda.current_stream()
            if s.cuda_stream not in self.seen:
                s.wait_event(self.ev)
                for t in self._tensors(self.val):
                    if t.is_cuda:
                        t.record_stream(s)
                self.seen.add(s.cuda_stream)
        return self.val


def weights_changed():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def weight_planes(W, transpose=False, fmt=None):
    """Planes of a weight matrix (or of a row slice of one), split ONCE and cached: frozen weights for the life of the
    process (keyed on storage + torch version counter), trainable ones until the next optimizer step.  h2 planes carry the
    row norms (`rnorm`): the B side of the bound behind an h2 planes output (_out_bound)."""
    fmt = fmt or PLANES_FMT

    def build():
        return split_planes(W.detach(), transpose=transpose, fmt=fmt)

    base = W._base if W._base is not None else W
    if not isinstance(base, torch.nn.Parameter):      # not a parameter: no identity to key a cache on
        return build()
    key = (W.data_ptr(), tuple(W.shape), W.stride(0), transpose, fmt)
    ver = (base._version, WEIGHT_EPOCH if base.requires_grad else 0)
    hit = _WPLANES.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is base:    # (a freed parameter's address may be handed out again)
        return hit[1].get()
    pl = build()
    if len(_WPLANES) > 4096:
        _WPLANES.clear()
    _WPLANES[key] = (ver, StreamCached(pl), weakref.ref(base))
    return pl


def _out_bound(B, bias):
    """Device float[2] = {max row norm of B, max |bias|} for svl_pgemm_desc::b_bound (one single-block launch,
    svl_bound2_f32), cached on the weight planes per (bias tensor version, stream)."""
    key = (None if bias is None else (bias.data_ptr(), bias._version, WEIGHT_EPOCH if bias.requires_grad else 0),
           torch.cuda.current_stream().cuda_stream)
    if B._bd is None:
        B._bd = {}
    hit = B._bd.get(key)
    if hit is not None:
        return hit
    bd = torch.empty(2, dtype=torch.float32, device=B.buf.device)
    bflat = bias.detach().view(-1) if bias is not None else None
    L.check(L.load().svl_bound2_f32(_p(B.rnorm), B.rows, _p(bflat), bflat.numel() if bflat is not None else 0, _p(bd), _st()),
            "svl_bound2_f32")
    if len(B._bd) > 8:
        B._bd.clear()
    B._bd[key] = bd
    return bd


def planes_eligible(M, N, K):
    """The pre-split path serves the large dense GEMMs of emulation mode 6 (the ViT linears: K, N in {768, 2304, 3072})
    when K is a whole number of MFMA k-groups; narrower ones (the decoder's K = 128 ... 512 per-pixel linears) stay on the
    in-register split kernel, where a separate split pass over A would cost more than it saves."""
    return get_gemm_emulation() == 6 and PLANES_PATH and M >= 256 and N >= 256 and K >= 512 and K % 16 == 0


# Mode 6 runs its dense GEMMs on the packed-planes kernel (csrc/gemm_planes.hip: LDS-DMA + MFMA only in the main loop);
# SVL_GEMM_NO_PLANES=1 keeps the in-register split kernel of gemm.hip for A/B runs.
PLANES_PATH = not os.environ.get("SVL_GEMM_NO_PLANES")


def pgemm(A, B, M, N, out=None, planes_out=None, bias=None, act=ACT_NONE, preact=None, resid=None, accumulate=False,
          m_off=0):
    """out[M, N] (fp32, row-major, optional) and / or planes_out (Planes [M, N], optional) = epi(A @ B^T)."""
    K = A.K
    assert B.K == K and A.rows >= m_off + M and B.rows >= N and A.fmt == B.fmt
    d = L.PGemmDesc()
    d.A, d.B, d.a_rows, d.b_rows = _p(A.buf), _p(B.buf), A.prow, B.prow
    d.m_off, d.M, d.N, d.K = m_off, M, N, K
    if out is not None:
        assert out.stride(1) == 1
        d.C, d.ldc = _p(out), out.stride(0)
    elif preact is not None:
        d.ldc = preact.stride(0)
    keep = None
    if A.fmt == "h2":
        d.fmt, d.a_sexp, d.b_sexp = 1, _p(A.sexp), _p(B.sexp)
    if planes_out is not None:
        d.planes_out, d.p_rows = _p(planes_out.buf), planes_out.prow
        if planes_out.fmt == "h2":     # row scales of the result from the bound |A_m| max|B_n| + max|bias| (include/semivl_hip.h)
            assert A.fmt == "h2" and A.rnorm is not None
            keep = _out_bound(B, bias)
            d.p_fmt, d.a_rnorm, d.b_bound, d.p_sexp = 1, _p(A.rnorm), _p(keep), _p(planes_out.sexp)
    d.bias, d.act, d.preact = _p(bias), act, _p(preact)
    if resid is not None:
        assert resid.stride(1) == 1
        d.resid, d.ldr = _p(resid), resid.stride(0)
    d.accumulate = 1 if accumulate else 0
    e0 = _prof_begin()
    L.check(L.load().svl_gemm_planes_f32(C.byref(d), _st()), "svl_gemm_planes_f32")
    _prof_end("gemm_bf16x", e0, 2.0 * M * N * K, ("planes" if A.fmt == "b3" else "planes_h2", M, N, K, act))


def _planes_out_for(xa, M, N, resid, act):
    """The Planes object a planes_only GEMM writes: h2 when the A operand is h2 and carries row norms (the result's row
    scales come from a bound, which a residual add would break), b3 otherwise (any A format can emit it)."""
    h2 = xa.fmt == "h2" and xa.rnorm is not None and (resid is None or act in (ACT_MUL_DGELU, ACT_MUL_DRELU))
    return Planes(M, N, device=xa.buf.device, fmt="h2" if h2 else "b3")


# ------------------------------------------------------------------------------------------------ dense helpers
def linear(x, W, bias=None, act=ACT_NONE, resid=None, out=None, accumulate=False, preact=None, planes_only=False):
    """out[M,N] = act(x[M,K] @ W[N,K]^T + bias) + resid   (torch F.linear layout).  `x` may be a Planes object (the
    previous GEMM's / producer's pre-split output).  planes_only=True returns the result as Planes instead of fp32 when
    the pre-split path is active (else the fp32 tensor): for values whose only consumer is the next GEMM."""
    M, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K and W.stride(1) == 1
    if isinstance(x, Planes) or planes_eligible(M, N, K):
        xa = x if isinstance(x, Planes) else split_planes(x)
        wb = weight_planes(W, fmt=xa.fmt)
        if preact is not None:
            assert out is None or preact.stride(0) == out.stride(0)
        if planes_only and N % 16 == 0 and out is None and not accumulate:
            po = _planes_out_for(xa, M, N, resid, act)
            pgemm(xa, wb, M, N, None, po, bias, act, preact, resid)
            return po
        if out is None:
            out = empty(M, N, device=xa.buf.device)
        pgemm(xa, wb, M, N, out, None, bias, act, preact, resid, accumulate)
        return out
    assert x.stride(1) == 1 and W.is_contiguous()
    if out is None:
        out = empty(M, N, device=x.device)
    gemm(A_KC, B_KC, M, N, K, Op(x, x.stride(0)), Op(W, K), out, ldc_m=out.stride(0), bias=bias, act=act,
         resid=resid, ldr_m=resid.stride(0) if resid is not None else None, accumulate=accumulate, preact=preact)
    return out


def copy2d(src, s_off, sgrp, src_go, src_ld, dst, d_off, dgrp, dst_go, dst_ld, rows, Cc, accumulate=False):
    """Strided row copy; offsets in elements."""
    L.check(L.load().svl_copy2d_f32(C.c_void_p(src.data_ptr() + 4 * s_off), sgrp, src_go, src_ld,
                                    C.c_void_p(dst.data_ptr() + 4 * d_off), dgrp, dst_go, dst_ld, rows, Cc,
                                    1 if accumulate else 0, _st()), "svl_copy2d_f32")


def matmul_nn(a, b, out=None, accumulate=False, dact=ACT_NONE, z=None, planes_only=False):
    """out[M,N] = a[M,K] @ b[K,N]  (dgrad: dY @ W with W [out,in]).  `dact` = ACT_MUL_DGELU / ACT_MUL_DRELU multiplies
    the result by the activation derivative at the saved pre-activation `z` [M,N] in the epilogue.  `a` may be Planes;
    planes_only as in linear()."""
    M, K = a.shape
    N = b.shape[1]
    assert b.shape[0] == K and b.stride(1) == 1
    if isinstance(a, Planes) or planes_eligible(M, N, K):
        xa = a if isinstance(a, Planes) else split_planes(a)
        wb = weight_planes(b, transpose=True, fmt=xa.fmt)   # B^T planes: rows = N (input features), k = K (output features)
        if dact != ACT_NONE:
            assert z is not None and z.shape == (M, N) and z.stride(1) == 1 and not accumulate
        if planes_only and N % 16 == 0 and out is None and not accumulate:
            po = _planes_out_for(xa, M, N, z if dact != ACT_NONE else None, dact)
            pgemm(xa, wb, M, N, None, po, None, dact, None, z if dact != ACT_NONE else None)
            return po
        if out is None:
            out = empty(M, N, device=xa.buf.device)
        pgemm(xa, wb, M, N, out, None, None, dact, None, z if dact != ACT_NONE else None, accumulate)
        return out
    assert a.stride(1) == 1
    if out is None:
        out = empty(M, N, device=a.device)
    if dact != ACT_NONE:
        assert z is not None and z.shape == (M, N) and z.stride(1) == 1 and not accumulate
        gemm(A_KC, B_NC, M, N, K, Op(a, a.stride(0)), Op(b, b.stride(0)), out, ldc_m=out.stride(0), act=dact, resid=z,
             ldr_m=z.stride(0))
        return out
    gemm(A_KC, B_NC, M, N, K, Op(a, a.stride(0)), Op(b, b.stride(0)), out, ldc_m=out.stride(0), accumulate=accumulate)
    return out


def _ksplit_plan(M, N, K, dense=False, emu_tiles=False):
    bm = 32 if M <= 32 else (64 if M <= 64 else 128)
    bn = 128 if (M <= 64 and N > 64) else (32 if N <= 32 else (64 if N <= 64 else 128))
    tiles = math.ceil(M / bm) * math.ceil(N / bn)
    slots = 768
    if emu_tiles or (dense and M >= 256 and N >= 96 and get_gemm_emulation()):
        tiles, slots = math.ceil(M / 128) * math.ceil(N / 128), 512   # the emulated kernel keeps 2 blocks per CU
    # slices so that tiles x slices fills, but does not exceed, ONE round of resident blocks (256 CUs x 3): 36 tiles x 29
    # slices = 1044 blocks was 1.36 rounds, i.e. a second, mostly idle round
    s = max(1, min(slots // tiles if tiles <= slots else 1, K // 512 if K >= 1024 else 1, 512))
    ks = math.ceil(math.ceil(K / s) / 16) * 16
    s = math.ceil(K / ks)
    return s, ks


def matmul_tn(a, b, out=None, accumulate=False):
    """out[M,N] = a[K,M]^T @ b[K,N] with deterministic split-K (wgrad: dY^T @ X)."""
    K, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == K and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = empty(M, N, device=a.device)
        accumulate = False
    assert out.is_contiguous()
    s, ks = _ksplit_plan(M, N, K, dense=True)
    if s == 1:
        gemm(A_MC, B_NC, M, N, K, Op(a, a.stride(0)), Op(b, b.stride(0)), out, accumulate=accumulate)
        return out
    slabs = empty(s, M, N, device=a.device)
    gemm(A_MC, B_NC, M, N, K, Op(a, a.stride(0)), Op(b, b.stride(0)), slabs, batch=s, ksplit=ks, c_bso=M * N)
    reduce_slabs(out, slabs, accumulate)
    return out


def reduce_slabs(out, slabs, accumulate=False):
    L.check(L.load().svl_reduce_slabs_f32(_p(out), _p(slabs), slabs.shape[0], out.numel(), 1 if accumulate else 0,
                                          _st()), "svl_reduce_slabs_f32")


def colsum(x2d, out=None, accumulate=False, C_=None, ld=None):
    rows = x2d.shape[0]
    Cc = C_ if C_ is not None else x2d.shape[1]
    ldd = ld if ld is not None else x2d.stride(0)
    if out is None:
        out = empty(Cc, device=x2d.device)
        accumulate = False
    if Cc == 1 and ldd == 1 and rows >= 65536 and rows % 1024 == 0:
        # the sum of one long column (the head conv's bias gradient: 19.6 M dlogits per ADE chunk) has ONE column of
        # parallelism in the column kernel -- 2.9 ms at 27 GB/s; as a [rows / 1024, 1024] matrix it is a 16 B-wide column
        # sum (full bandwidth) followed by a 1024-element one.  Fixed summation order either way.
        part = colsum(x2d.reshape(rows // 1024, 1024))
        return colsum(part.view(1024, 1), out=out, accumulate=accumulate)
    lib = L.load()
    ws = empty(int(lib.svl_colsum_ws_floats(rows, Cc)), device=x2d.device)
    L.check(lib.svl_colsum_f32(_p(x2d), rows, Cc, ldd, _p(out), 1 if accumulate else 0, _p(ws), _st()),
            "svl_colsum_f32")
    return out


def eltwise(mode, a, b=None, out=None):
    if out is None:
        out = torch.empty_like(a)
    L.check(L.load().svl_eltwise_f32(mode, _p(a), _p(b), _p(out), a.numel(), _st()), "svl_eltwise_f32")
    return out


def add(a, b, out=None):
    return eltwise(0, a, b, out)


def gelu(a, out=None):
    return eltwise(5, a, None, out)


def fill(t, v):
    L.check(L.load().svl_fill_f32(_p(t), float(v), t.numel(), _st()), "svl_fill_f32")
    return t


def affine_planes(x, k4):
    """((x * k4[0][c] + k4[1][c]) - k4[2][c]) / k4[3][c] on an NCHW tensor (k4 [4, C] on the device)."""
    x = x.contiguous()
    Bn, Cc = x.shape[:2]
    y = torch.empty_like(x)
    L.check(L.load().svl_affine_planes_f32(_p(x), Bn * Cc, Cc, x[0, 0].numel(), _p(k4), _p(y), _st()),
            "svl_affine_planes_f32")
    return y


def chanmask(x, mask, scale, rows_per_img, out=None):
    rows, Cc = x.shape
    if out is None:
        out = torch.empty_like(x)
    L.check(L.load().svl_chanmask_f32(_p(x), _p(mask), float(scale), rows, rows_per_img, Cc, _p(out), _st()),
            "svl_chanmask_f32")
    return out


# ------------------------------------------------------------------------------------------------ normalisation


def layernorm_fwd(x, gamma, beta, eps, planes=False, want_y=True):
    """y = LN(x), stats.  planes=True additionally returns the result as packed Planes (the next GEMM's A operand);
    want_y=False then skips the fp32 copy (returns None for it)."""
    rows, Cc = x.shape
    stats = empty(rows, 2, device=x.device)
    if not planes:
        y = torch.empty_like(x)
        L.check(L.load().svl_layernorm_fwd(_p(x), _p(gamma), _p(beta), float(eps), rows, Cc, _p(y), _p(stats), _st()),
                "svl_layernorm_fwd")
        return y, stats
    if PLANES_FMT == "h2" and not want_y:    # planes only: one kernel (row statistics, exponents, planes)
        pl = Planes(rows, Cc, device=x.device, fmt="h2")
        pl.rnorm = torch.empty(pl.prow, dtype=torch.float32, device=x.device)
        L.check(L.load().svl_layernorm_fwd_planes_f16x2(_p(x), _p(gamma), _p(beta), float(eps), rows, Cc, None, _p(stats),
                                                        _p(pl.buf), pl.prow, _p(pl.sexp), _p(pl.rnorm), _st()),
                "svl_layernorm_fwd_planes_f16x2")
        return None, stats, pl
    if want_y or PLANES_FMT == "h2":   # measured at [32800, 768]: row pass (39 us) + pack pass over the cache-warm result (38 us) beats the fused
        y = torch.empty_like(x)      # kernel (102 us: its 32-row blocks leave 8 sequential rows per wave); planes-only
        L.check(L.load().svl_layernorm_fwd(_p(x), _p(gamma), _p(beta), float(eps), rows, Cc, _p(y), _p(stats), _st()),
                "svl_layernorm_fwd")                                                 # is faster fused (60 us vs 77)
        return (y if want_y else None), stats, split_planes(y)      # (h2: the pack pass finds the row scales; no fused form yet)
    pl = Planes(rows, Cc, device=x.device, fmt="b3")
    L.check(L.load().svl_layernorm_fwd_planes(_p(x), _p(gamma), _p(beta), float(eps), rows, Cc, None, _p(stats),
                                              _p(pl.buf), pl.prow, _st()), "svl_layernorm_fwd_planes")
    return None, stats, pl


def layernorm_bwd(dy, x, stats, gamma, dx_add=None, want_wgrad=False, planes=False):
    """dx (+ dgamma, dbeta).  planes=True: dx is additionally emitted as packed Planes (appended to the result)."""
    rows, Cc = x.shape
    dx = torch.empty_like(x)
    lib = L.load()
    dgp = dbp = None
    if want_wgrad:
        nparts = lib.svl_layernorm_bwd_parts(rows)
        dgp = empty(nparts, Cc, device=x.device)
        dbp = empty(nparts, Cc, device=x.device)
    L.check(lib.svl_layernorm_bwd(_p(dy), _p(x), _p(stats), _p(gamma), rows, Cc, _p(dx_add), _p(dx), _p(dgp), _p(dbp),
                                  _st()), "svl_layernorm_bwd")
    res = (dx, colsum(dgp), colsum(dbp)) if want_wgrad else (dx,)
    if planes:   # (measured: 81 us + a 38 us pack pass over the cache-warm dx vs 162 us for svl_layernorm_bwd_planes)
        res = res + (split_planes(dx),)
    return res if len(res) > 1 else res[0]


def softmax_rows_fwd(s, rows, cols, ld, scale):
    L.check(L.load().svl_softmax_rows_fwd(_p(s), rows, cols, ld, float(scale), _st()), "svl_softmax_rows_fwd")


def softmax_rows_bwd(dp, p, rows, cols, ld, scale):
    L.check(L.load().svl_softmax_rows_bwd(_p(dp), _p(p), rows, cols, ld, float(scale), _st()), "svl_softmax_rows_bwd")


def l2norm_fwd(x, eps=0.0):
    rows, Cc = x.shape
    y = torch.empty_like(x)
    inv = empty(rows, device=x.device)
    L.check(L.load().svl_l2norm_fwd(_p(x), rows, Cc, float(eps), _p(y), _p(inv), _st()), "svl_l2norm_fwd")
    return y, inv


def l2norm_bwd(dy, y, inv):
    rows, Cc = y.shape
    dx = torch.empty_like(y)
    L.check(L.load().svl_l2norm_bwd(_p(dy), _p(y), _p(inv), rows, Cc, _p(dx), _st()), "svl_l2norm_bwd")
    return dx


def groupnorm_fwd(x, ldx, gamma, beta, eps, imgs, HW, Cc, G, relu, y, ldy):
    """x/y are flat tensors viewed as [imgs*HW, ld]; y may be a channel slice of a wider buffer (pass data offset via
    a narrowed view: y.data_ptr() is used)."""
    stats = empty(imgs, G, 2, device=x.device)
    L.check(L.load().svl_groupnorm_fwd(_p(x), ldx, _p(gamma), _p(beta), float(eps), imgs, HW, Cc, G, 1 if relu else 0,
                                       _p(y), ldy, _p(stats), _st()), "svl_groupnorm_fwd")
    return stats


UP_LOSS = not os.environ.get("SVL_NO_UP_LOSS")      # A/B: the logits' resize evaluated inside the pixel-loss kernels (train.py)
GN_DEFER = True     # GroupNorm + ReLU applied by the consuming convolution (model/vlg_head.py; tests flip it for bit-identity checks)


def groupnorm_scale_shift(stats, gamma, beta, imgs, Cc, G):
    """[imgs, 2, C] (scale, shift) table of a GroupNorm from its statistics (the `gn_in` operand of the tiled convolutions)."""
    t = empty(imgs, 2, Cc, device=stats.device)
    L.check(L.load().svl_groupnorm_scale_shift(_p(stats), _p(gamma), _p(beta), imgs, Cc, G, _p(t), _st()),
            "svl_groupnorm_scale_shift")
    return t


def conv3x3_gn(x, ldx, imgs, H, W, C1, wf, Co, eps, src2=None, ld2=0, C2=0, rep=1, gn_in=None):
    """3x3 / pad 1 convolution fused with the statistics of the following GroupNorm (groups of 16 channels): returns
    (pre [imgs*H*W, Co], stats [imgs, Co/16, 2]), or None when the tiled kernel does not take the shape (the caller then
    runs conv_fwd + groupnorm_fwd)."""
    if not (CONV_TILED and Co % 16 == 0):
        return None
    lib = L.load()
    ws = torch.empty(max(1, lib.svl_conv3x3_gn_ws_doubles(imgs, H, W, Co)), dtype=torch.float64, device=x.device)
    pre = empty(imgs * H * W, Co, device=x.device)
    stats = empty(imgs, Co // 16, 2, device=x.device)
    e0 = _prof_begin()
    rc = lib.svl_conv3x3_gn_f32(_p(x), ldx, C1, _p(src2), ld2, C2, rep, _p(wf), imgs, H, W, Co, _p(pre), Co, float(eps),
                                _p(ws), _p(stats), _p(gn_in), _p(w_planes_of(wf)), _st())
    if rc == -3:            # SVL_ERR_UNSUPPORTED: nothing was launched
        return None
    L.check(rc, "svl_conv3x3_gn_f32")
    if e0 is not None:
        x6 = get_gemm_emulation() == 6 and not os.environ.get("SVL_CONV_TILED_NO_EMU")
        K = 9 * (C1 + C2)
        _prof_end("gemm_bf16x" if x6 else "gemm", e0, 2.0 * imgs * H * W * Co * K, (A_CONV, B_KC, imgs * H * W, Co, K, 1))
    return pre, stats


def groupnorm_apply(x, ldx, gamma, beta, imgs, HW, Cc, G, relu, stats, y, ldy):
    """y = relu?(groupnorm(x)) from existing statistics: the forward's apply pass alone (bit-identical result)."""
    L.check(L.load().svl_groupnorm_apply(_p(x), ldx, _p(gamma), _p(beta), imgs, HW, Cc, G, 1 if relu else 0, _p(stats),
                                         _p(y), ldy, _st()), "svl_groupnorm_apply")
    return y


def groupnorm_bwd(dy, lddy, x, ldx, y, ldy, stats, gamma, imgs, HW, Cc, G, relu, dx, lddx, beta=None):
    """With `beta` given the ReLU mask is re-derived from x (the forward's own fma) and y is not read."""
    cs = empty(imgs, 2, Cc, device=x.device)
    L.check(L.load().svl_groupnorm_bwd(_p(dy), lddy, _p(x), ldx, _p(None if beta is not None else y), ldy, _p(stats),
                                       _p(gamma), _p(beta), imgs, HW, Cc, G, 1 if relu else 0, _p(dx), lddx, _p(cs),
                                       _st()), "svl_groupnorm_bwd")
    flat = cs.view(imgs, 2 * Cc)
    dbeta = colsum(flat, C_=Cc, ld=2 * Cc)
    dgamma = colsum(flat[:, Cc:], C_=Cc, ld=2 * Cc)
    return dgamma, dbeta


GN_BWD_FUSED = True      # GroupNorm-backward sums from the producing dgrad's epilogue (round 6: 327.4 -> 326.5 ms VOC, 1121 -> 1114 ms ADE)


def conv3x3_dgrad_gnb(dy, lddy, imgs, H, W, Co, wd, Ci, gn_x, gn_stats, gn_gamma, gn_beta, G):
    """Input gradient dx [pix, Ci] of a narrow 3x3 convolution (dgrad pack wd [Ci, 9 Co]) TOGETHER with the backward channel
    sums of the GroupNorm + ReLU (over gn_x [pix, Ci], G groups of 16 channels) whose output gradient dx is: returns
    (dx, chan_sums [imgs, 2, Ci]) or None when the fused kernel does not take the launch (callers: conv_dgrad + groupnorm_bwd)."""
    if not (GN_BWD_FUSED and CONV_TILED and Ci % 16 == 0 and G * 16 == Ci and dy.is_cuda and get_gemm_emulation() == 6):
        return None
    lib = L.load()
    table = groupnorm_scale_shift(gn_stats, gn_gamma, gn_beta, imgs, Ci, G)
    ws = torch.empty(max(1, lib.svl_conv3x3_gnb_ws_doubles(imgs, H, W, Ci)), dtype=torch.float64, device=dy.device)
    dx = empty(imgs * H * W, Ci, device=dy.device)
    cs = empty(imgs, 2, Ci, device=dy.device)
    e0 = _prof_begin()
    rc = lib.svl_conv3x3_dgrad_gnb_f32(_p(dy), lddy, Co, _p(wd), imgs, H, W, Ci, _p(dx), Ci, 0, _p(gn_x), _p(table), _p(gn_stats),
                                       _p(ws), _p(cs), _p(w_planes_of(wd)), _st())
    if rc == -3:            # SVL_ERR_UNSUPPORTED: nothing was launched
        return None
    L.check(rc, "svl_conv3x3_dgrad_gnb_f32")
    _prof_end("gemm_bf16x", e0, 2.0 * imgs * H * W * Ci * 9 * Co, (A_CONV, B_KC, imgs * H * W, Ci, 9 * Co, 1))
    return dx, cs


def groupnorm_bwd_from_sums(dy, lddy, x, ldx, stats, gamma, beta, imgs, HW, Cc, G, relu, cs, dx, lddx):
    """svl_groupnorm_bwd's apply pass on channel sums `cs` [imgs, 2, C] that a producer's epilogue left (conv3x3_dgrad_gnb)."""
    L.check(L.load().svl_groupnorm_bwd_apply(_p(dy), lddy, _p(x), ldx, _p(stats), _p(gamma), _p(beta), imgs, HW, Cc, G,
                                             1 if relu else 0, _p(cs), _p(dx), lddx, _st()), "svl_groupnorm_bwd_apply")
    flat = cs.view(imgs, 2 * Cc)
    dbeta = colsum(flat, C_=Cc, ld=2 * Cc)
    dgamma = colsum(flat[:, Cc:], C_=Cc, ld=2 * Cc)
    return dgamma, dbeta


# ------------------------------------------------------------------------------------------------ ViT attention (fused, D = 64)
def attention_h2():
    """The fused attention runs on the fp16 x 2 kernels of csrc/attn_h2.hip (pre-packed operands, three products per term):
    emulation mode 6, unless SVL_ATTN_NO_EMU keeps the exact fp32 kernels (A/B runs).  (The bf16 x 6 family of rounds 2-5
    and its own planes outputs were retired in round 6.)"""
    return get_gemm_emulation() == 6 and not os.environ.get("SVL_ATTN_NO_EMU")


def _attn_family():
    """Profile family of the attention launches (bench.py)."""
    return "attention_bf16x" if (PROFILE is not None and attention_h2()) else "attention"


def attention_planes_ok():
    """The attention results can be handed on as packed planes (the fp32 result through the generic pack pass)."""
    return attention_h2() and PLANES_PATH


def _attn_ws(Bn, T, H, backward, dev):
    n = L.load().svl_attention_h2_ws_bytes(Bn, T, H, 1 if backward else 0)
    return torch.empty(n + 1024, dtype=torch.uint8, device=dev), n


def _ws_ptr(ws):
    return (ws.data_ptr() + 1023) // 1024 * 1024


def attention_fwd(qkv, Bn, T, H, want_lse=True, planes=False, want_out=True):
    """Flash-style fused attention: qkv [Bn*T, 3E] -> (out [Bn*T, E] or None, lse [Bn*H*T] or None[, out as Planes]).
    planes=True (attention_planes_ok()) additionally returns the output as packed planes -- 38 us at [32800, 768] for the
    pack pass, less than what the out-projection saves on fp16 x 2 operands; want_out=False then drops the fp32 copy
    (gradient-free passes)."""
    E = H * 64
    if planes and not attention_planes_ok():
        raise RuntimeError("attention_fwd: planes outputs need emulation mode 6 (attention_planes_ok())")
    out = empty(Bn * T, E, device=qkv.device)
    lse = empty(Bn * H * T, device=qkv.device) if want_lse else None
    e0 = _prof_begin()
    if attention_h2():
        ws, n = _attn_ws(Bn, T, H, False, qkv.device)
        L.check(L.load().svl_attention_fwd_h2(_p(qkv), Bn, T, H, _p(out), _p(lse), None, 0, _ws_ptr(ws), n, _st()),
                "svl_attention_fwd_h2")
    else:
        L.check(L.load().svl_attention_fwd(_p(qkv), Bn, T, H, _p(out), _p(lse), None, 0, _st()), "svl_attention_fwd")
    _prof_end(_attn_family(), e0, 4.0 * Bn * H * T * T * 64, ("fwd_h2" if attention_h2() else "fwd", Bn, T, H))
    if not planes:
        return out, lse
    op = split_planes(out)
    return (out if want_out else None), lse, op


def attention_bwd(dout, qkv, out, lse, Bn, T, H, planes=False):
    """-> dqkv [Bn*T, 3E] (and, with planes=True, the same as Planes through the generic pack pass)."""
    if planes and not attention_planes_ok():
        raise RuntimeError("attention_bwd: planes outputs need emulation mode 6 (attention_planes_ok())")
    dqkv = torch.empty_like(qkv)
    ws = empty(Bn * H * T, device=qkv.device)
    e0 = _prof_begin()
    if attention_h2():
        wsb, n = _attn_ws(Bn, T, H, True, qkv.device)
        L.check(L.load().svl_attention_bwd_h2(_p(qkv), _p(out), _p(dout), _p(lse), Bn, T, H, _p(ws), _p(dqkv), None, 0,
                                              _ws_ptr(wsb), n, _st()), "svl_attention_bwd_h2")
    else:
        L.check(L.load().svl_attention_bwd(_p(qkv), _p(out), _p(dout), _p(lse), Bn, T, H, _p(ws), _p(dqkv), None, 0, _st()),
                "svl_attention_bwd")
    _prof_end(_attn_family(), e0, 14.0 * Bn * H * T * T * 64, ("bwd_h2" if attention_h2() else "bwd", Bn, T, H))
    return (dqkv, split_planes(dqkv)) if planes else dqkv


# ------------------------------------------------------------------------------------------------ ViT attention (materialised probabilities; head dims != 64)
def vit_attention_fwd(qkv, Bn, T, H, D):
    """qkv [Bn*T, 3E] -> (out [Bn*T, E], probs [Bn*H, T, Tp]).  q is scaled by D^-0.5 (a power of two here) inside
    the softmax, identical to nn.MultiheadAttention's q-scaling."""
    E = H * D
    Tp = (T + 3) // 4 * 4
    dev = qkv.device
    P = empty(Bn * H, T, Tp, device=dev)
    gemm(A_KC, B_KC, T, T, D, Op(qkv, 3 * E, 0, T * 3 * E, D), Op(qkv, 3 * E, E, T * 3 * E, D), P, ldc_m=Tp,
         batch=Bn * H, batch_inner=H, c_bso=H * T * Tp, c_bsi=T * Tp)
    softmax_rows_fwd(P, Bn * H * T, T, Tp, 1.0 / math.sqrt(D))
    out = empty(Bn * T, E, device=dev)
    gemm(A_KC, B_NC, T, D, T, Op(P, Tp, 0, H * T * Tp, T * Tp), Op(qkv, 3 * E, 2 * E, T * 3 * E, D), out, ldc_m=E,
         batch=Bn * H, batch_inner=H, c_bso=T * E, c_bsi=D)
    return out, P


def vit_attention_bwd(dout, qkv, P, Bn, T, H, D):
    E = H * D
    Tp = P.shape[2]
    dev = qkv.device
    dqkv = empty(Bn * T, 3 * E, device=dev)
    bh = dict(batch=Bn * H, batch_inner=H)
    p_op = Op(P, Tp, 0, H * T * Tp, T * Tp)
    do_op = Op(dout, E, 0, T * E, D)
    # dV = P^T dO
    gemm(A_MC, B_NC, T, D, T, p_op, do_op, dqkv, c_off=2 * E, ldc_m=3 * E, c_bso=T * 3 * E, c_bsi=D, **bh)
    # dP = dO V^T
    dP = empty(Bn * H, T, Tp, device=dev)
    gemm(A_KC, B_KC, T, T, D, do_op, Op(qkv, 3 * E, 2 * E, T * 3 * E, D), dP, ldc_m=Tp, c_bso=H * T * Tp,
         c_bsi=T * Tp, **bh)
    softmax_rows_bwd(dP, P, Bn * H * T, T, Tp, 1.0 / math.sqrt(D))
    ds_op = Op(dP, Tp, 0, H * T * Tp, T * Tp)
    # dQ = dS K ; dK = dS^T Q
    gemm(A_KC, B_NC, T, D, T, ds_op, Op(qkv, 3 * E, E, T * 3 * E, D), dqkv, c_off=0, ldc_m=3 * E, c_bso=T * 3 * E,
         c_bsi=D, **bh)
    gemm(A_MC, B_NC, T, D, T, ds_op, Op(qkv, 3 * E, 0, T * 3 * E, D), dqkv, c_off=E, ldc_m=3 * E, c_bso=T * 3 * E,
         c_bsi=D, **bh)
    return dqkv


# ------------------------------------------------------------------------------------------------ SemanticTransformer attention
def permute_rows(x, outer, A, Bd, Cc):
    """[outer, A, Bd, Cc] -> [outer, Bd, A, Cc] (rows of Cc floats)."""
    out = empty(outer * A * Bd, Cc, device=x.device)
    L.check(L.load().svl_permute_rows_f32(_p(x), outer, A, Bd, Cc, _p(out), _st()), "svl_permute_rows_f32")
    return out


# class sequences of at least this length go through the fused (flash-style, MFMA) attention kernels of the ViT instead of
# the wave-per-query VALU kernel (seqattn.hip): N = 81 / 150 (COCO / ADE); N = 19 / 21 would idle 7 of 8 waves of a block
SEQATTN_MFMA_MIN = int(os.environ.get("SVL_SEQATTN_MFMA_MIN", "64"))


def seqattn_fwd(qkv, groups, inner, seq, heads, outer_stride, inner_stride, seq_stride):
    rows, E3 = qkv.shape
    E = E3 // 3
    out = empty(rows, E, device=qkv.device)
    probs = empty(groups, heads, seq, seq, device=qkv.device)
    d = L.SeqAttnDesc(groups, inner, seq, heads, outer_stride, inner_stride, seq_stride, _p(qkv), _p(out), _p(probs),
                      None, None, None)
    L.check(L.load().svl_seqattn_fwd(C.byref(d), _st()), "svl_seqattn_fwd")
    return out, probs


def seqattn_bwd(dout, qkv, probs, groups, inner, seq, heads, outer_stride, inner_stride, seq_stride):
    dqkv = torch.empty_like(qkv)
    ds = torch.empty_like(probs)
    d = L.SeqAttnDesc(groups, inner, seq, heads, outer_stride, inner_stride, seq_stride, _p(qkv), None, _p(probs),
                      _p(dout), _p(dqkv), _p(ds))
    L.check(L.load().svl_seqattn_bwd(C.byref(d), _st()), "svl_seqattn_bwd")
    return dqkv


# ------------------------------------------------------------------------------------------------ convolutions (NHWC, stride 1)
_PACKS = {}


def cached_pack(W, tag, fn):
    """A re-laid-out copy of a weight (conv operand packs, ConvTranspose packs), built ONCE per parameter version: frozen
    weights for the life of the process, trainable ones until the next optimizer step (`weights_changed`).  Keeps the
    ~130 weight-sized permute kernels per step (and their launches) off the hot path.  Non-parameters are not cached."""
    base = W._base if W._base is not None else W
    if not isinstance(base, torch.nn.Parameter):
        return fn(W.detach())
    key = (W.data_ptr(), tuple(W.shape), tag)
    ver = (base._version, WEIGHT_EPOCH if base.requires_grad else 0)
    hit = _PACKS.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is base:    # weak reference: the address of a freed parameter of an
        return hit[1].get()                                        # earlier model may be handed out again by the allocator
    out = fn(W.detach())
    if len(_PACKS) > 4096:
        _PACKS.clear()
    _PACKS[key] = (ver, StreamCached(out), weakref.ref(base))      # stream-aware: see StreamCached
    return out


def conv3x3_weight_planes(pack, N, Ct):
    """bf16 x 3 planes of a narrow 3x3 convolution's packed weights [N, 9 Ct] in the tiled kernel's LDS image
    (svl_conv3x3_weight_planes), or None when the tiled split kernel cannot use one."""
    if not (pack.is_cuda and N in (32, 64, 128) and Ct % 16 == 0):     # (128: the dilated ASPP convolutions, conv_dil.hip)
        return None
    lib = L.load()
    pl = torch.empty(lib.svl_conv3x3_weight_planes_bytes(N, Ct), dtype=torch.uint8, device=pack.device)
    L.check(lib.svl_conv3x3_weight_planes(_p(pack), N, Ct, _p(pl), _st()), "svl_conv3x3_weight_planes")
    return pl


def w_planes_of(pack):
    """The planes image that travels with a packed weight (pack_conv_w), if any."""
    return getattr(pack, "_svl_planes", None)


def pack_conv_w(W):
    """[Co, Ci, kh, kw] -> forward pack [Co, (kh kw) Ci] and dgrad pack [Ci, (kh kw) Co] (weight-sized permutes, cached).
    The packs of the narrow 3x3 layers carry their pre-split planes (`w_planes_of`): built with the pack, once per
    parameter version, kept alive and stream-tracked with it."""
    def build(w):
        Co, Ci, kh, kw = w.shape
        w = w.contiguous()
        cs = (Ci * kh * kw, kh * kw, kw, 1)                          # element strides of [Co, Ci, kh, kw]
        wf = permute4(w, (Co, kh, kw, Ci), (cs[0], cs[2], cs[3], cs[1])).view(Co, kh * kw * Ci)
        wd = permute4(w, (Ci, kh, kw, Co), (cs[1], cs[2], cs[3], cs[0])).view(Ci, kh * kw * Co)
        out = [wf, wd]
        if kh == 3 and kw == 3:
            for pk, n, ct in ((wf, Co, Ci), (wd, Ci, Co)):
                pl = conv3x3_weight_planes(pk, n, ct)
                if pl is not None:
                    pk._svl_planes = pl
                    out.append(pl)
        return tuple(out)
    r = cached_pack(W, "conv", build)
    return r[0], r[1]


def unpack_conv_wgrad(dwf, Co, Ci, kh, kw):
    return permute4(dwf, (Co, Ci, kh, kw), (kh * kw * Ci, 1, kw * Ci, Ci))     # [Co, kh, kw, Ci] -> [Co, Ci, kh, kw]


CONV_TILED = not os.environ.get("SVL_CONV_NO_TILED")   # narrow 3x3 weight gradients on conv_tiled.hip


def conv_out_size(H, W, KH, KW, dil, pad, stride):
    return (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1, (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1


def conv_fwd(x, ldx, imgs, H, W, C1, wf, Co, KH, KW, dil, pad, bias=None, act=ACT_NONE, out=None, ldo=None, src2=None,
             ld2=0, C2=0, rep=1, resid=None, ldr=None, stride=1):
    """y[pix, Co] = conv(x) with packed weight wf [Co, KH*KW*(C1+C2)]; stride > 1: pix runs over the output grid."""
    Ho, Wo = conv_out_size(H, W, KH, KW, dil, pad, stride) if stride != 1 else (H, W)
    M = imgs * Ho * Wo
    K = KH * KW * (C1 + C2)
    if out is None:
        out = empty(M, Co, device=x.device)
        ldo = Co
    g = conv_geom(H, W, C1, KH, KW, dil, pad, 1, C2, rep, src2, ld2, stride=stride, Ho=Ho, Wo=Wo)
    gemm(A_CONV, B_KC, M, Co, K, Op(x, ldx), Op(wf, K), out, ldc_m=ldo, bias=bias, act=act, conv=g, resid=resid,
         ldr_m=ldr, w_planes=w_planes_of(wf) if (KH, KW, pad, stride) == (3, 3, dil, 1) else None, emu_h2=EMU_H2_CONVFWD)
    return out


def conv_dgrad(dy, lddy, imgs, H, W, Co, wd, Ci, KH, KW, dil, pad, out=None, ldo=None, accumulate=False):
    """dx[pix, Ci] from dy[pix, Co] with the dgrad pack wd [Ci, KH*KW*Co] (taps mirrored via sign=-1)."""
    M = imgs * H * W
    K = KH * KW * Co
    if out is None:
        out = empty(M, Ci, device=dy.device)
        ldo = Ci
    g = conv_geom(H, W, Co, KH, KW, dil, pad, -1)
    gemm(A_CONV, B_KC, M, Ci, K, Op(dy, lddy), Op(wd, K), out, ldc_m=ldo, conv=g, accumulate=accumulate,
         w_planes=w_planes_of(wd) if (KH, KW, pad) == (3, 3, dil) else None, emu_h2=True)
    return out


def conv_wgrad_tiled_ok(imgs, H, W, C1, C2, Co, lddy, ldx, ld2=0):
    """The spatially tiled 3x3 weight-gradient kernels (conv_tiled.hip) take this layer (they also accept `gn_in`)."""
    co_ok = Co in (32, 64) or (Co == 128 and get_gemm_emulation() == 6 and not os.environ.get("SVL_CONV_TILED_NO_EMU"))
    return (CONV_TILED and co_ok and C1 % 4 == 0 and C2 % 4 == 0 and (C1 + C2) % 32 == 0 and H >= 8 and W >= 16 and
            imgs * H * W >= 16384 and lddy % 4 == 0 and ldx % 4 == 0 and (C2 == 0 or ld2 % 4 == 0))


def conv_wgrad(dy, lddy, x, ldx, imgs, H, W, C1, Co, KH, KW, dil, pad, src2=None, ld2=0, C2=0, rep=1, stride=1, gn_in=None):
    """dWf[Co, KH*KW*(C1+C2)] = dy^T im2col(x), deterministic split-K over (output) pixels.  `gn_in`: x is a
    pre-normalisation tensor, the operand is relu(groupnorm(x)) (tiled kernels only: conv_wgrad_tiled_ok)."""
    Ho, Wo = conv_out_size(H, W, KH, KW, dil, pad, stride) if stride != 1 else (H, W)
    Kpix = imgs * Ho * Wo
    N = KH * KW * (C1 + C2)
    co_ok = Co in (32, 64) or (Co == 128 and get_gemm_emulation() == 6 and not os.environ.get("SVL_CONV_TILED_NO_EMU"))
    if (CONV_TILED and KH == 3 and KW == 3 and dil == 1 and pad == 1 and stride == 1 and co_ok and
            C1 % 4 == 0 and C2 % 4 == 0 and (C1 + C2) % 32 == 0 and H >= 8 and W >= 16 and Kpix >= 16384 and lddy % 4 == 0 and ldx % 4 == 0 and
            (C2 == 0 or ld2 % 4 == 0)):
        # narrow layers: spatially tiled weight-gradient kernel (conv_tiled.hip), slabs reduced in fixed order
        lib = L.load()
        groups = lib.svl_conv3x3_wgrad_tiled_groups(imgs, H, W, C1 + C2, Co)
        slabs = empty(groups, Co, N, device=dy.device)
        e0 = _prof_begin()
        L.check(lib.svl_conv3x3_wgrad_tiled(_p(dy), lddy, Co, _p(x), ldx, C1, _p(src2), ld2, C2, rep, imgs, H, W,
                                            _p(slabs), groups, _p(gn_in), _st()), "svl_conv3x3_wgrad_tiled")
        x6 = (get_gemm_emulation() == 6 and not os.environ.get("SVL_CONV_TILED_NO_EMU") and
              (Co >= 64 or (C1 + C2) % 64 == 0 or (C1 + C2 == 32 and H >= 8 and groups >= 2)))   # the dispatch rule of svl_conv3x3_wgrad_tiled
        _prof_end("gemm_bf16x" if x6 else "gemm", e0, 2.0 * Co * N * Kpix, ("wgrad3x3_tiled", Co, N, Kpix, 1))
        out = empty(Co, N, device=dy.device)
        reduce_slabs(out, slabs)
        return out
    assert gn_in is None, "gn_in needs the tiled weight-gradient kernel (conv_wgrad_tiled_ok)"
    g = conv_geom(H, W, C1, KH, KW, dil, pad, 1, C2, rep, src2, ld2, stride=stride, Ho=Ho, Wo=Wo)
    # (the split-emulation kernel serves this launch when Cout >= 96, N >= 96 and the output rows are whole 8-pixel groups:
    # 128-row tiles, two resident blocks per CU)
    x6 = (get_gemm_emulation() in (3, 6) and Co >= 96 and N >= 96 and Wo % 8 == 0 and Kpix >= 1024 and Kpix % 16 == 0)
    s, ks = _ksplit_plan(Co, N, Kpix, emu_tiles=x6)
    out = empty(Co, N, device=dy.device)
    if s == 1:
        gemm(A_MC, B_CONVW, Co, N, Kpix, Op(dy, lddy), Op(x, ldx), out, conv=g)
        return out
    slabs = empty(s, Co, N, device=dy.device)
    gemm(A_MC, B_CONVW, Co, N, Kpix, Op(dy, lddy), Op(x, ldx), slabs, batch=s, ksplit=ks, c_bso=Co * N, conv=g)
    reduce_slabs(out, slabs)
    return out


def conv_cout1_gn_ok(H, W, Cc, KH=3, KW=3, dil=1, pad=1):
    """The LDS-tiled Conv2d(C -> 1) kernel takes this layer (the form that accepts `gn_in`)."""
    return KH == 3 and KW == 3 and dil == 1 and pad == 1 and Cc in (16, 32, 64) and H >= 8 and W >= 16


def conv_cout1_fwd(x, ldx, imgs, H, W, Cc, wf, KH, KW, dil, pad, bias=None, out=None, gn_in=None):
    """Conv2d(C -> 1) forward, HBM-bound direct kernel; wf = forward pack [1, KH*KW*C]; returns [imgs*H*W, 1].
    `gn_in`: x is a pre-normalisation tensor, the operand is relu(groupnorm(x)) (conv_cout1_gn_ok)."""
    y = empty(imgs * H * W, 1, device=x.device) if out is None else out
    assert y.is_contiguous() and y.numel() == imgs * H * W
    L.check(L.load().svl_conv_cout1_fwd(_p(x), ldx, imgs, H, W, Cc, KH, KW, dil, pad, _p(wf), _p(bias), _p(gn_in), _p(y),
                                        _st()), "svl_conv_cout1_fwd")
    return y


def conv_cout1_wgrad(dy, x, ldx, imgs, H, W, Cc, dil, pad, gn_in=None):
    """Weight gradient of Conv2d(C -> 1, 3x3): returns the forward-pack layout [1, 9*C] (`gn_in` as in conv_cout1_fwd)."""
    lib = L.load()
    nb = lib.svl_conv_cout1_wgrad_blocks(imgs, H, W)
    slabs = empty(nb, 9 * Cc, device=x.device)
    L.check(lib.svl_conv_cout1_wgrad(_p(dy), _p(x), ldx, imgs, H, W, Cc, dil, pad, _p(gn_in), _p(slabs), _st()),
            "svl_conv_cout1_wgrad")
    out = empty(1, 9 * Cc, device=x.device)
    reduce_slabs(out, slabs)
    return out


def conv_cin1_dgrad(dy, lddy, imgs, H, W, Co, wtap, KH, KW, dil, pad):
    """Input gradient of Conv2d(1 -> Co): T = dY @ Wtap^T (GEMM, N = KH*KW taps) then the shifted-tap gather.
    wtap [KH*KW, Co]."""
    M = imgs * H * W
    T = empty(M, KH * KW, device=dy.device)
    gemm(A_KC, B_KC, M, KH * KW, Co, Op(dy, lddy), Op(wtap, Co), T)
    out = empty(M, 1, device=dy.device)
    L.check(L.load().svl_tap_gather(_p(T), imgs, H, W, KH, KW, dil, pad, 1, _p(out), _st()), "svl_tap_gather")
    return out


def convT2x_fwd(x, ldx, imgs, H, W, Ci, wp, Co, bias, out, ldo):
    """ConvTranspose2d(k=2,s=2): wp [4*Co, Ci] packed as n=(a,b,co); out [imgs,2H,2W,:] pixel stride ldo."""
    M = imgs * H * W
    gemm(A_KC, B_KC, M, 4 * Co, Ci, Op(x, ldx), Op(wp, Ci), out, ldc_m=ldo, bias=bias, bias_mod=Co,
         out_mode=OUT_CONVT2X, ct=(H, W, Co))
    return out


def convT2x_dgrad(du, lddu, imgs, H, W, Co, wb, Ci):
    """dx[imgs*H*W, Ci] of ConvTranspose2d(k2,s2) = conv(k2, s2) of du [imgs,2H,2W,Co] with wb [Ci, 4*Co] (n=(a,b,co))."""
    M = imgs * H * W
    g = conv_geom(2 * H, 2 * W, Co, 2, 2, 1, 0, 1, stride=2, Ho=H, Wo=W)
    out = empty(M, Ci, device=du.device)
    gemm(A_CONV, B_KC, M, Ci, 4 * Co, Op(du, lddu), Op(wb, 4 * Co), out, conv=g)
    return out


def convT2x_wgrad(x, ldx, du, lddu, imgs, H, W, Ci, Co):
    """dWb[Ci, (a,b,co)] = sum_m x[m,ci] * du[pix(m;a,b), co]."""
    Kpix = imgs * H * W
    N = 4 * Co
    g = conv_geom(2 * H, 2 * W, Co, 2, 2, 1, 0, 1, stride=2, Ho=H, Wo=W)
    x6 = (get_gemm_emulation() in (3, 6) and Ci >= 96 and N >= 96 and W % 8 == 0 and Kpix >= 1024 and Kpix % 16 == 0)
    s, ks = _ksplit_plan(Ci, N, Kpix, emu_tiles=x6)
    out = empty(Ci, N, device=x.device)
    if s == 1:
        gemm(A_MC, B_CONVW, Ci, N, Kpix, Op(x, ldx), Op(du, lddu), out, conv=g)
        return out
    slabs = empty(s, Ci, N, device=x.device)
    gemm(A_MC, B_CONVW, Ci, N, Kpix, Op(x, ldx), Op(du, lddu), slabs, batch=s, ksplit=ks, c_bso=Ci * N, conv=g)
    reduce_slabs(out, slabs)
    return out


# ------------------------------------------------------------------------------------------------ resampling
def bilinear_nhwc_fwd(x, ldx, imgs, h, w, Cc, align, rep, H, W, y, ldy, accumulate=False):
    L.check(L.load().svl_bilinear_nhwc_fwd(_p(x), ldx, imgs, h, w, Cc, 1 if align else 0, rep, H, W, _p(y), ldy,
                                           1 if accumulate else 0, _st()), "svl_bilinear_nhwc_fwd")


def bilinear_nhwc_bwd(dy, lddy, imgs, h, w, Cc, align, rep, H, W, dx, lddx, accumulate=False):
    if rep > 1 and Cc % 4 == 0 and lddy % 4 == 0:
        # sum the class repeats first (one coalesced pass), then a rep = 1 backward on the per-image map
        summed = empty(imgs * H * W, Cc, device=dy.device)
        L.check(L.load().svl_sum_rep_f32(_p(dy), lddy, imgs, rep, H * W, Cc, _p(summed), _st()), "svl_sum_rep_f32")
        dy, lddy, rep = summed, Cc, 1
    L.check(L.load().svl_bilinear_nhwc_bwd(_p(dy), lddy, imgs, h, w, Cc, 1 if align else 0, rep, H, W, _p(dx), lddx,
                                           1 if accumulate else 0, _st()), "svl_bilinear_nhwc_bwd")


def bilinear_planes_fwd(x, h, w, align, H, W, out=None):
    planes = x.numel() // (h * w)
    y = empty(*x.shape[:-2], H, W, device=x.device) if out is None else out
    assert y.is_contiguous() and y.numel() == planes * H * W
    L.check(L.load().svl_bilinear_planes_fwd(_p(x), planes, h, w, 1 if align else 0, H, W, _p(y), _st()),
            "svl_bilinear_planes_fwd")
    return y


def bilinear_planes_bwd(dy, h, w, align, H, W):
    planes = dy.numel() // (H * W)
    dx = empty(*dy.shape[:-2], h, w, device=dy.device)
    L.check(L.load().svl_bilinear_planes_bwd(_p(dy), planes, h, w, 1 if align else 0, H, W, _p(dx), _st()),
            "svl_bilinear_planes_bwd")
    return dx


def avgpool_cat_fwd(x, imgs, H, W, Cc, P, text, nclass):
    """P: int (square window) or (PH, PW)."""
    PH, PW = (P, P) if isinstance(P, int) else P
    Ct = text.shape[1] if text is not None else 0
    y = empty(imgs * (H // PH) * (W // PW), Cc + Ct, device=x.device)
    L.check(L.load().svl_avgpool_cat_fwd(_p(x), imgs, H, W, Cc, PH, PW, _p(text), Ct, nclass, _p(y), _st()),
            "svl_avgpool_cat_fwd")
    return y


def avgpool_cat_bwd(dy, imgs, H, W, Cc, P, Ct, nclass, add_to=None):
    """`add_to`: an existing [imgs*H*W, Cc] gradient the pooled gradient is added to in place (returned as dx)."""
    PH, PW = (P, P) if isinstance(P, int) else P
    dx = empty(imgs * H * W, Cc, device=dy.device) if add_to is None else add_to
    lib = L.load()
    L.check(lib.svl_avgpool_cat_bwd(_p(dy), imgs, H, W, Cc, PH, PW, Ct, _p(dx), 0 if add_to is None else 1, _st()),
            "svl_avgpool_cat_bwd")
    dtext = None
    if Ct > 0:
        dtext, part = empty(nclass, Ct, device=dy.device), empty(imgs, Ct, device=dy.device)
        L.check(lib.svl_avgpool_cat_bwd_text(_p(dy), imgs, (H // PH) * (W // PW), Cc, Ct, nclass, _p(part), _p(dtext), _st()),
                "svl_avgpool_cat_bwd_text")
    return dx, dtext


# ------------------------------------------------------------------------------------------------ pixel losses
def softmax_max(logits):
    Bn, N = logits.shape[:2]
    HW = logits[0, 0].numel()
    conf = empty(Bn, *logits.shape[2:], device=logits.device)
    label = empty(Bn, *logits.shape[2:], dtype=torch.int64, device=logits.device)
    e0 = _prof_begin()
    L.check(L.load().svl_softmax_max_f32(_p(logits), Bn, N, HW, _p(conf), _p(label), _st()), "svl_softmax_max_f32")
    _prof_end("softmax_max", e0, float(Bn * HW) * (4 * N + 12), (Bn, N, HW))
    return conf, label


def cutmix_f32(a, b, box, out=None):
    """where(box==1, b, a); a,b [B,(C,)H,W] fp32, box [B,H,W] fp32."""
    Bn = a.shape[0]
    HW = box[0].numel()
    Cc = a[0].numel() // HW
    if out is None:
        out = torch.empty_like(a)
    L.check(L.load().svl_cutmix_f32(_p(out), _p(a), _p(b), _p(box), Bn, Cc, HW, _st()), "svl_cutmix_f32")
    return out


def cutmix_i64(a, b, box):
    out = torch.empty_like(a)
    L.check(L.load().svl_cutmix_i64(_p(out), _p(a), _p(b), _p(box), a.shape[0], box[0].numel(), _st()),
            "svl_cutmix_i64")
    return out


def count_valid(map_i64, out_count):
    """out_count (int64[1] view, pre-zeroed) += #(map != 255)."""
    L.check(L.load().svl_count_valid_i64(_p(map_i64), map_i64.numel(), _p(out_count), _st()), "svl_count_valid_i64")


def ce_fused(logits, target, use_ignore_t, conf=None, ign=None, conf_thresh=0.0, mc=None, dlogits=None, gscale=None,
             sums_out=None, all_pixels=False, img_weight=None):
    """Returns sums (double[4] device): {sum w*ce_t, sum ce_m, sum conf*valid, #valid}."""
    Bn, N = logits.shape[:2]
    HW = logits[0, 0].numel()
    lib = L.load()
    nblk = lib.svl_ce_num_blocks(Bn, N, HW)
    if nblk <= 0:
        raise RuntimeError(f"svl_ce_fused: unsupported N={N}")
    partials = empty(nblk, 4, device=logits.device)
    d = L.CeDesc(_p(logits), Bn, N, HW, _p(target), 1 if use_ignore_t else 0, _p(conf), _p(ign), float(conf_thresh),
                 1 if all_pixels else 0, _p(mc), _p(partials), _p(dlogits), _p(gscale), _p(img_weight))
    e0 = _prof_begin()
    L.check(lib.svl_ce_fused_f32(C.byref(d), _st()), "svl_ce_fused_f32")
    # algorithmic bytes: fwd (4N+20) + bwd (8N+20) B/px when dlogits is produced, else fwd only
    _prof_end("ce_fused", e0, float(Bn * HW) * ((12 * N + 40) if dlogits is not None else (4 * N + 20)), (Bn, N, HW))
    if sums_out is None:
        sums_out = empty(4, dtype=torch.float64, device=logits.device)
    L.check(lib.svl_ce_finalize(_p(partials), nblk, _p(sums_out), _st()), "svl_ce_finalize")
    return sums_out


def ce_up_ok(Bn, N, h, w, H, W, align):
    """Whether the pixel-loss kernels can evaluate the resize [h, w] -> [H, W] themselves (svl_ce_up_num_blocks)."""
    return L.load().svl_ce_up_num_blocks(int(Bn), int(N), int(h), int(w), int(H), int(W), 1 if align else 0) > 0


def softmax_max_up(logits, H, W, align):
    """softmax(dim 1).max(dim 1) of bilinear(logits [B, N, h, w] -> [H, W]) without writing the resized tensor."""
    Bn, N, h, w = logits.shape
    conf = empty(Bn, H, W, device=logits.device)
    label = empty(Bn, H, W, dtype=torch.int64, device=logits.device)
    e0 = _prof_begin()
    L.check(L.load().svl_softmax_max_up_f32(_p(logits), Bn, N, h, w, H, W, 1 if align else 0, _p(conf), _p(label), _st()),
            "svl_softmax_max_up_f32")
    _prof_end("softmax_max_up", e0, float(Bn) * (4.0 * N * h * w + 12.0 * H * W), (Bn, N, h * w, H * W))
    return conf, label


def ce_up_fused(logits, H, W, align, target, use_ignore_t, conf=None, ign=None, conf_thresh=0.0, mc=None, dlogits=None,
                gscale=None, sums_out=None, all_pixels=False, img_weight=None):
    """ce_fused on bilinear(logits [B, N, h, w] -> [H, W]) with the resize evaluated inside the kernel; `dlogits`
    [B, N, h, w] receives d(loss)/d(logits) at the LOW resolution.  Returns sums (double[4] device)."""
    Bn, N, h, w = logits.shape
    lib = L.load()
    nblk = lib.svl_ce_up_num_blocks(Bn, N, h, w, H, W, 1 if align else 0)
    if nblk <= 0:
        raise RuntimeError(f"svl_ce_up_fused: unsupported geometry N={N} {h}x{w} -> {H}x{W}")
    partials = empty(nblk, 4, device=logits.device)
    d = L.CeUpDesc(_p(logits), Bn, N, h, w, H, W, 1 if align else 0, _p(target), 1 if use_ignore_t else 0, _p(conf),
                   _p(ign), float(conf_thresh), 1 if all_pixels else 0, _p(mc), _p(partials), _p(dlogits), _p(gscale),
                   _p(img_weight))
    e0 = _prof_begin()
    L.check(lib.svl_ce_up_fused_f32(C.byref(d), _st()), "svl_ce_up_fused_f32")
    # bytes the kernel moves: low-resolution logits in (+ gradient out) and 28 B of maps per full-resolution pixel
    _prof_end("ce_up_fused", e0, float(Bn) * ((8.0 if dlogits is not None else 4.0) * N * h * w + 28.0 * H * W),
              (Bn, N, h * w, H * W))
    if sums_out is None:
        sums_out = empty(4, dtype=torch.float64, device=logits.device)
    L.check(lib.svl_ce_finalize(_p(partials), nblk, _p(sums_out), _st()), "svl_ce_finalize")
    return sums_out


def maskclip_labels(dense, H, W, scale, thresh, ign=None):
    Bn, N, h, w = dense.shape
    out = empty(Bn, H, W, dtype=torch.int64, device=dense.device)
    L.check(L.load().svl_maskclip_labels(_p(dense), Bn, N, h, w, H, W, float(scale), float(thresh), _p(ign), _p(out),
                                         _st()), "svl_maskclip_labels")
    return out


def concept_max(pred, offsets_i32, N):
    Bn, NC = pred.shape[:2]
    HW = pred[0, 0].numel()
    out = empty(Bn, N, *pred.shape[2:], device=pred.device)
    L.check(L.load().svl_concept_max_f32(_p(pred), Bn, NC, HW, _p(offsets_i32), N, _p(out), _st()),
            "svl_concept_max_f32")
    return out


def adamw_step(p, g, m, v, seg_off, seg_lr, seg_wd, nseg, beta1, beta2, eps, step, gscale=1.0, ema=None, ema_decay=0.0):
    L.check(L.load().svl_adamw_step(_p(p), _p(g), _p(m), _p(v), _p(seg_off), _p(seg_lr), _p(seg_wd), nseg, p.numel(),
                                    float(beta1), float(beta2), float(eps), int(step), float(gscale), _p(ema),
                                    float(ema_decay), _st()), "svl_adamw_step")


def semivl_gscale(counts_i64, numel_u, lam, gscale_out, factors=None, mc_counts=None):
    L.check(L.load().svl_semivl_gscale(_p(counts_i64), float(numel_u), float(lam), _p(factors), _p(mc_counts),
                                       _p(gscale_out), _st()), "svl_semivl_gscale")


def semivl_loss(sums_f64, numel_u, lam, out8, factors=None, mc_counts=None):
    L.check(L.load().svl_semivl_loss(_p(sums_f64), float(numel_u), float(lam), _p(factors), _p(mc_counts), _p(out8),
                                     _st()), "svl_semivl_loss")


def bernoulli(shape, keep_prob, device):
    """fp32 tensor of independent Bernoulli(keep_prob) draws (svl_bernoulli_f32).  The counter range of a call is drawn from
    torch's CPU generator (one host-side randint, no device sync): the masks follow torch.manual_seed like the reference's
    F.dropout2d does, without reproducing its exact stream."""
    out = empty(*shape, device=device)
    off = int(torch.randint(0, 2 ** 62, (1,)).item())
    L.check(L.load().svl_bernoulli_f32(_p(out), out.numel(), float(keep_prob), torch.initial_seed() & (2 ** 64 - 1), off,
                                       _st()), "svl_bernoulli_f32")
    return out


def conf_ratio(conf, ign, thresh):
    """ratio[b] = #(conf_b >= thresh & valid) / #valid   (conf_mode 'pixelratio', train_utils.py:39-40) -> float[B]."""
    lib = L.load()
    out = empty(conf.shape[0], device=conf.device)
    ws = torch.empty(int(lib.svl_conf_avg_ws_doubles(conf.shape[0])), dtype=torch.float64, device=conf.device)
    L.check(lib.svl_conf_ratio_f32(_p(conf), _p(ign), conf.shape[0], conf[0].numel(), float(thresh), _p(out), _p(ws), _st()),
            "svl_conf_ratio_f32")
    return out


def conf_avg_factor(conf, ign, out_f64):
    """out_f64[0] = sum_b mean_{valid}(conf_b)  ('pixelavg' confidence weighting)."""
    lib = L.load()
    ws = torch.empty(int(lib.svl_conf_avg_ws_doubles(conf.shape[0])), dtype=torch.float64, device=conf.device)
    L.check(lib.svl_conf_avg_factor(_p(conf), _p(ign), conf.shape[0], conf[0].numel(), _p(out_f64), _p(ws), _st()),
            "svl_conf_avg_factor")


def softmax_planes(logits):
    Bn, N = logits.shape[:2]
    out = torch.empty_like(logits)
    L.check(L.load().svl_softmax_planes_f32(_p(logits), Bn, N, logits[0, 0].numel(), _p(out), _st()),
            "svl_softmax_planes_f32")
    return out


def iou_hist(pred, target, K, ignore_index, hist):
    """hist (int64 [3K], accumulated) += intersection / prediction-area / target-area counts."""
    L.check(L.load().svl_iou_hist_i64(_p(pred), _p(target), pred.numel(), K, ignore_index, _p(hist), _st()),
            "svl_iou_hist_i64")


def set_gemm_emulation(mode):
    """0: exact fp32 MFMA (default); 6 / 3: bf16 split emulation for the large dense GEMMs (include/semivl_hip.h)."""
    L.check(L.load().svl_set_gemm_emulation(int(mode)), "svl_set_gemm_emulation")


def get_gemm_emulation():
    return L.load().svl_get_gemm_emulation()


# ------------------------------------------------------------------------------------------------ BatchNorm / MaxPool
def bn_stats(x, C, rows=None, ld=None):
    """[2, C] float64: per-channel (sum, sum of squares) of a channels-last [rows, C] activation."""
    rows = rows if rows is not None else x.shape[0]
    ld = ld if ld is not None else x.stride(0)
    lib = L.load()
    sums = torch.empty(2, C, dtype=torch.float64, device=x.device)
    ws = torch.empty(int(lib.svl_bn_ws_doubles(rows, C)), dtype=torch.float64, device=x.device)
    L.check(lib.svl_bn_stats(_p(x), ld, rows, C, _p(sums), _p(ws), _st()), "svl_bn_stats")
    return sums


def bn_finalize(sums, count, eps, momentum, running_mean, running_var):
    C_ = sums.shape[1]
    mean, invstd = empty(C_, device=sums.device), empty(C_, device=sums.device)
    L.check(L.load().svl_bn_finalize(_p(sums), float(count), eps, momentum, _p(running_mean), _p(running_var), C_,
                                     _p(mean), _p(invstd), _st()), "svl_bn_finalize")
    return mean, invstd


def bn_eval_invstd(running_var, eps):
    out = empty(running_var.numel(), device=running_var.device)
    L.check(L.load().svl_bn_eval_invstd(_p(running_var), eps, running_var.numel(), _p(out), _st()), "svl_bn_eval_invstd")
    return out


def bn_apply(x, C, mean, invstd, gamma, beta, relu=False, resid=None, out=None):
    rows = x.shape[0]
    if out is None:
        out = empty(rows, C, device=x.device)
    L.check(L.load().svl_bn_apply(_p(x), x.stride(0), rows, C, _p(mean), _p(invstd), _p(gamma), _p(beta), _p(resid),
                                  resid.stride(0) if resid is not None else 0, 1 if relu else 0, _p(out), out.stride(0),
                                  _st()), "svl_bn_apply")
    return out


def bn_bwd_reduce(dy, x, y, C, mean, invstd, remask=None):
    """[2, C] float64: (sum dy', sum dy' * xhat); y (the ReLU output) masks dy when given; remask = (gamma, beta): the mask is
    re-derived from x with the forward's own expression instead (BatchNorm + ReLU without a residual input)."""
    rows = x.shape[0]
    lib = L.load()
    sums = torch.empty(2, C, dtype=torch.float64, device=x.device)
    ws = torch.empty(int(lib.svl_bn_ws_doubles(rows, C)), dtype=torch.float64, device=x.device)
    if remask is not None:
        y = None
    L.check(lib.svl_bn_bwd_reduce(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(y), y.stride(0) if y is not None else 0,
                                  _p(mean), _p(invstd), _p(remask[0] if remask else None), _p(remask[1] if remask else None),
                                  rows, C, _p(sums), _p(ws), _st()), "svl_bn_bwd_reduce")
    return sums


def bn_bwd_apply(dy, x, y, C, mean, invstd, gamma, sums, count, want_dres=False, remask_beta=None):
    rows = x.shape[0]
    dx = empty(rows, C, device=x.device)
    dres = empty(rows, C, device=x.device) if want_dres else None
    if remask_beta is not None:
        y = None
    L.check(L.load().svl_bn_bwd_apply(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(y), y.stride(0) if y is not None else 0,
                                      _p(mean), _p(invstd), _p(gamma), _p(remask_beta), _p(sums), float(count), rows, C, _p(dx),
                                      dx.stride(0), _p(dres), dres.stride(0) if dres is not None else 0, _st()),
            "svl_bn_bwd_apply")
    return (dx, dres) if want_dres else dx


def maxpool3x3s2_fwd(x, imgs, H, W, C):
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = empty(imgs * Ho * Wo, C, device=x.device)
    idx = torch.empty(imgs * Ho * Wo, C, dtype=torch.uint8, device=x.device)
    L.check(L.load().svl_maxpool3x3s2_fwd(_p(x), imgs, H, W, C, _p(y), _p(idx), _st()), "svl_maxpool3x3s2_fwd")
    return y, idx, Ho, Wo


def maxpool3x3s2_bwd(dy, idx, imgs, H, W, C):
    dx = empty(imgs * H * W, C, device=dy.device)
    L.check(L.load().svl_maxpool3x3s2_bwd(_p(dy), _p(idx), imgs, H, W, C, _p(dx), _st()), "svl_maxpool3x3s2_bwd")
    return dx
