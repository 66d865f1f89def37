"""Thin torch-tensor wrappers over the C ABI (include/dahitra_hip.h).

torch is used for device memory and the current HIP stream only; every arithmetic result below is
produced by a kernel of libdahitra_hip.so.  All activations are NHWC tensors of dtype float32
(parity mode) or bfloat16 (throughput mode)."""
import ctypes
import os
import sys

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
_DT = {torch.float32: 0, torch.bfloat16: 1}
_vp = ctypes.c_void_p
_ci = ctypes.c_int
_cl = ctypes.c_long
_cf = ctypes.c_float
_cd = ctypes.c_double


def dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError("dahitra_amd: unsupported activation dtype %s" % t.dtype)


def set_f32_mma_mode(mode):
    """dh_set_f32_mma_mode: 0 = exact fp32 MFMA, 1 = split-bf16 three-product form (two bf16 planes, unit roundoff 2^-17), 2 = the
    six-product form (three bf16 planes, 2^-23), 3 = the split-fp16 three-product form (two fp16 planes, ~2^-21; operands inside
    fp16's range) for the matrix products of every fp32 launch this host thread issues from now on (the Engine sets it at each
    of its entry points: compute_dtype="bf16x3" = form 3 forward, form 1 backward)."""
    _lib.check(_lib.lib().dh_set_f32_mma_mode(int(mode)), "dh_set_f32_mma_mode")


def get_f32_mma_mode():
    return int(_lib.lib().dh_get_f32_mma_mode())


class f32_mma_mode:
    """with ops.f32_mma_mode(1): ...  -- the mode for the block, the previous one restored after it"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_f32_mma_mode()
        set_f32_mma_mode(self.mode)
        return self

    def __exit__(self, *exc):
        set_f32_mma_mode(self.prev)
        return False


def chunk_channels(dtype):
    """channels per 64-byte MFMA K-chunk: the granularity of Cin for dh_conv2d_fwd"""
    return 32 if dtype == torch.bfloat16 else 16


def P(t):
    if t is None:
        return _vp(0)
    assert t.is_cuda and t.is_contiguous(), "dahitra_amd ops need contiguous device tensors"
    return _vp(t.data_ptr())


def S():
    return _vp(torch.cuda.current_stream().cuda_stream)


_ws = {}


def workspace(nbytes, device):
    """scratch for ONE launch at a time per stream (keyed by device AND current stream: the levels of the hierarchical model
    run on streams of their own, Engine._run_staged, and must not share it)"""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream) if torch.device(device).type == "cuda" else str(device)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def rekey_workspace(device, old_stream, new_stream):
    """hand the scratch buffer that `old_stream` grew (a graph's eager warm-up) to `new_stream` (its capture stream): the
    capture then finds a buffer of the right size that lives OUTSIDE the graph's private pool, and the warm-up stream --
    which is gone after the capture -- leaves no entry behind"""
    buf = _ws.pop((str(device), old_stream.cuda_stream), None)
    if buf is not None:
        cur = _ws.get((str(device), new_stream.cuda_stream))
        if cur is None or cur.numel() < buf.numel():
            _ws[(str(device), new_stream.cuda_stream)] = buf


_PINNED = []        # buffers a captured HIP graph points into: kept alive for the life of the process


def pin_captured_buffers(net):
    """Called right after a HIP-graph capture of `net`'s step.  The graph's kernel nodes hold RAW pointers into
    buffers that live outside the graph's private pool: the shared scratch workspace, the weight-gradient plan's slabs
    and device job table, the packed-weight plans, the parameter / gradient arenas.  Each of them is normally replaced
    (and the old tensor returned to the allocator) when a later eager call needs more room or another shape -- the
    last partial batch of an epoch, an eval forward at a larger size, another net in the process.  Holding a reference
    here means a replacement allocates a NEW buffer while the one the graph reads and writes stays valid."""
    keep = list(_ws.values())
    eng = net._engine
    if eng._wgrad_plan is not None:
        keep += list(eng._wgrad_plan.slots) + list(eng._wgrad_plan.tables.values())
    for pack, _, xstack, wstack in eng._plans.values():
        keep += [pack.table] + list(pack.keep) + list(xstack.values())
        for fwd_stack, dgrad_stack in wstack.values():
            keep += [fwd_stack, dgrad_stack]
    keep += [net._arena.flat, net._arena.grad]
    keep = [t for t in keep if t is not None]
    _PINNED.extend(keep)
    return keep


def _call(name, *args):
    L = _lib.lib()
    _lib.check(getattr(L, name)(*args), name)


def cdiv(a, b):
    return (a + b - 1) // b


def pad16(c):
    return cdiv(c, 16) * 16


# ---- weights -------------------------------------------------------------------------------------
def pack_weight(w, dtype, want_dgrad=True, dgrad_inner=0, out_scale=None, want_fwd=True, out_dgrad=None):
    """w: OIHW (or [O, I]) fp32 master.  Returns (fwd [taps, OPad, I], dgrad [taps, IPad, OK] | None).
    out_scale [O]: per-output-channel factor folded into the forward form (eval-mode BatchNorm).
    out_dgrad: write the data-gradient form into this (contiguous) buffer instead of a fresh one."""
    if w.dim() == 2:
        O, I = w.shape
        ks = 1
    else:
        O, I, ks, _ = w.shape
    OPad, IPad = pad16(O), pad16(I)
    OK = max(O, dgrad_inner)
    fwd = torch.empty(ks * ks, OPad, I, dtype=dtype, device=w.device) if want_fwd else None
    if out_dgrad is not None:
        assert out_dgrad.numel() == ks * ks * IPad * OK and out_dgrad.dtype == dtype
        dg = out_dgrad
    else:
        dg = torch.empty(ks * ks, IPad, OK, dtype=dtype, device=w.device) if want_dgrad else None
    _call("dh_pack_weight", _ci(_DT[dtype]), P(w), P(out_scale), _ci(O), _ci(I), _ci(ks), _ci(OPad), P(fwd), _ci(IPad), _ci(OK),
          P(dg), S())
    return fwd, dg


class PackPlan:
    """persistent packed-weight buffers + the device job table of dh_pack_weights_multi (one launch per step)"""

    class _Job(ctypes.Structure):
        _fields_ = [("w", ctypes.c_void_p), ("fwd", ctypes.c_void_p), ("dgrad", ctypes.c_void_p), ("O", ctypes.c_int),
                    ("I", ctypes.c_int), ("KS", ctypes.c_int), ("OPad", ctypes.c_int), ("IPad", ctypes.c_int),
                    ("OK", ctypes.c_int), ("dtype", ctypes.c_int), ("first_block", ctypes.c_int), ("nblocks", ctypes.c_int)]

    def __init__(self, device):
        self.device, self.jobs, self.blocks, self.table, self.keep = device, [], 0, None, []

    def add(self, w, dtype, want_fwd=True, want_dgrad=True, dgrad_inner=0, out_dgrad=None, frag=False, out_fwd=None):
        """same contract as pack_weight; returns the (persistent) fwd / dgrad buffers.  frag (bf16 3x3 weights, I and
        dgrad_inner multiples of 32): the buffers are in FRAGMENT order [rows / 16, K / 32, taps, 64, 8] -- what the
        register-resident-weights convolution loads (conv2d's w_frag); add the layer a second time without it for the
        row-major form every other kernel reads"""
        if w.dim() == 2:
            (O, I), ks = w.shape, 1
        else:
            O, I, ks, _ = w.shape
        OPad, IPad, OK = pad16(O), pad16(I), max(O, dgrad_inner)
        ck = chunk_channels(dtype)
        if frag:
            assert ks == 3 and dtype == torch.bfloat16 and I % 32 == 0 and OK % 32 == 0 and out_dgrad is None
            fwd = torch.empty(OPad // 16, I // 32, ks * ks, 64, 8, dtype=dtype, device=w.device) if want_fwd else None
            dg = torch.empty(IPad // 16, OK // 32, ks * ks, 64, 8, dtype=dtype, device=w.device) if want_dgrad else None
        else:
            fwd = out_fwd if out_fwd is not None else \
                (torch.empty(ks * ks, OPad, I, dtype=dtype, device=w.device) if want_fwd else None)
            dg = out_dgrad if out_dgrad is not None else \
                (torch.empty(ks * ks, IPad, OK, dtype=dtype, device=w.device) if want_dgrad else None)
        n = max(fwd.numel() if fwd is not None else 0, dg.numel() if dg is not None else 0)
        assert n < 2 ** 31 and w.numel() < 2 ** 31, "pack_weights_multi indexes a layer with 32-bit arithmetic"
        nb = max(1, min(256, cdiv(n, 2048)))        # a lane packs 8 elements per pass (16-byte store): one pass per lane
        if ks == 3 and O % 32 == 0 and I % 32 == 0 and OK == O:
            nb = min(256, (O // 32) * (I // 32))       # the tiled form (csrc/pointwise.hip pack_job_tiled): one 32 x 32 x 9 tile per workgroup
        self.jobs.append(self._Job(w.data_ptr(), fwd.data_ptr() if fwd is not None else None,
                                   dg.data_ptr() if dg is not None else None, O, I, ks, OPad, IPad, OK,
                                   _DT[dtype] | (0x200 if frag else 0), self.blocks, nb))
        self.blocks += nb
        self.keep += [w, fwd, dg]
        return fwd, dg

    def run(self):
        if not self.jobs:
            return
        if self.table is None:
            assert ctypes.sizeof(self._Job) == _lib.lib().dh_pack_job_size()
            raw = b"".join(bytes(j) for j in self.jobs)
            self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        _call("dh_pack_weights_multi", P(self.table), _ci(len(self.jobs)), _ci(self.blocks), S())


class WgradPlan:
    """The split-K reduces of one backward pass as ONE launch (dh_wgrad_reduce_multi): while a plan is active
    (`with plan:`), conv2d_wgrad writes its partial slabs into a per-layer persistent workspace and records a job;
    `run()` sums them all into the gradient arena.  The job table is built on the first pass and reused while the
    sequence of layers (gradient pointers, shapes, split factors) repeats -- which it does for a fixed network and
    batch shape, and what makes the table capturable in a HIP graph."""

    class _Job(ctypes.Structure):
        _fields_ = [("part", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("splitk", ctypes.c_int), ("taps", ctypes.c_int),
                    ("Oslab", ctypes.c_int), ("O", ctypes.c_int), ("I", ctypes.c_int), ("accumulate", ctypes.c_int),
                    ("first_block", ctypes.c_int), ("nblocks", ctypes.c_int)]

    def __init__(self, device, batch=False):
        """batch: the wave-specialised 3x3 weight gradients of the pass are RECORDED by conv2d_wgrad and issued as one launch by
        run() (dh_wgrad_batch_*: fewer, longer K slices per layer, a quarter of the partial-slab traffic); their x / dy are
        kept alive until then.  Off under ops.PROFILE (the per-launch events need one launch per layer) and with
        DAHITRA_WGRAD_BATCH=0."""
        self.device, self.slots, self.cur, self.blocks = device, [], 0, 0
        self.tables, self.sigs, self.phase = {}, {}, 0      # one job table per run() of a pass (a split backward runs twice)
        self._prev = None
        self.post = []
        self.batch = batch and os.environ.get("DAHITRA_WGRAD_BATCH", "1") != "0"
        self.batching, self.keep = False, []

    def __enter__(self):
        global _WGRAD_PLAN
        self._prev, _WGRAD_PLAN = _WGRAD_PLAN, self
        self.cur = 0
        self.jobs = []
        self.post = []
        self.blocks = 0
        self.phase = 0
        self.keep = []
        self.batching = self.batch and PROFILE is None
        if self.batching:
            _call("dh_wgrad_batch_begin")
        elif self._prev is None:
            _call("dh_wgrad_batch_abort")       # a pass that never finished must not leave its batch open behind it
        return self

    def __exit__(self, *exc):
        global _WGRAD_PLAN
        _WGRAD_PLAN = self._prev
        if self.batching:
            self.batching = False
            if exc and exc[0] is not None:
                _lib.lib().dh_wgrad_batch_abort()      # drop what an aborted pass recorded
            else:
                _call("dh_wgrad_batch_end", S())
            self.keep = []
        return False

    def hold(self, *tensors):
        """operands of a recorded (not yet issued) weight-gradient launch"""
        if self.batching:
            self.keep += [t for t in tensors if t is not None]

    def slab(self, nbytes):
        """persistent workspace of the cur-th deferred layer of the pass"""
        if self.cur == len(self.slots):
            self.slots.append(torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.device))
        elif self.slots[self.cur].numel() < nbytes:
            _PINNED.append(self.slots[self.cur])         # a captured graph may still point at the smaller buffer
            self.slots[self.cur] = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            self.tables = {}
        buf = self.slots[self.cur]
        self.cur += 1
        return buf

    def add(self, part, dw, splitk, taps, oslab, o, i, accumulate):
        nb = cdiv(o * i * taps, _lib.lib().dh_wgrad_reduce_outputs_per_block(i))
        self.jobs.append(self._Job(part.data_ptr(), dw.data_ptr(), splitk, taps, oslab, o, i, int(accumulate),
                                   self.blocks, nb))
        self.blocks += nb

    def run(self):
        """reduce every layer deferred since the last run() of this pass (a split backward calls it once per part)"""
        if self.batching:
            _call("dh_wgrad_batch_launch", S())      # the recorded weight gradients, one launch
            self.keep = []
        if self.jobs:
            self._run_reduce()
        for fn in self.post:          # launches that consume reduced gradients (phase-gradient combine)
            fn()
        self.jobs, self.post, self.blocks = [], [], 0
        self.phase += 1

    def _run_reduce(self):
        sig = b"".join(bytes(j) for j in self.jobs)
        if self.tables.get(self.phase) is None or sig != self.sigs.get(self.phase):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("dahitra_amd: the weight-gradient plan changed during graph capture; run one eager "
                                   "step of the same shapes first")
            assert ctypes.sizeof(self._Job) == _lib.lib().dh_wgrad_reduce_job_size()
            if self.tables.get(self.phase) is not None:
                _PINNED.append(self.tables[self.phase])
            self.tables[self.phase] = torch.frombuffer(bytearray(sig), dtype=torch.uint8).to(self.device)
            self.sigs[self.phase] = sig
        _call("dh_wgrad_reduce_multi", P(self.tables[self.phase]), _ci(len(self.jobs)), _ci(self.blocks), S())


_WGRAD_PLAN = None


class SideStream:
    """A second HIP stream for launches that nothing downstream of the current kernel chain waits for (a weight gradient
    next to the low-occupancy token-side backward kernels).  fork(): the side stream waits for everything enqueued so far
    on the current stream; launches inside `with side:` go to it; join(): the current stream waits for the side stream.
    Works under HIP-graph capture (the fork / join become graph dependencies).  Tensors handed to side launches are kept
    alive until the join (the caching allocator must not give their memory to a later main-stream launch)."""

    def __init__(self, device):
        self.stream, self.keep, self.active, self._ctx = torch.cuda.Stream(device=device), [], False, None

    def fork(self, *tensors):
        global _PERSIST_BLOCK
        self.stream.wait_stream(torch.cuda.current_stream())
        self.keep += [t for t in tensors if t is not None]
        if not self.active:
            _PERSIST_BLOCK += 1       # until the join: no device-wide-barrier kernels next to the side stream's launches
        self.active = True
        return self

    def __enter__(self):
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self._ctx.__exit__(*exc)
        self._ctx = None
        return False

    def join(self):
        global _PERSIST_BLOCK
        if self.active:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.keep, self.active = [], False
            _PERSIST_BLOCK -= 1


# ---- convolution / linear ------------------------------------------------------------------------
PROFILE = None       # set to {} by bench.py to collect (events, algorithmic flops, algorithmic bytes) per launch
REPLAY = None        # set to {"key": class, "calls": []} by bench.py for ONE eager step: the C calls of that kernel class as
                     # closures (stream -> launch) with their tensors kept alive, to be re-issued back to back in a recorded graph


class _Prof:
    """bench.py's per-launch timer: HIP events recorded on the launch stream around ONE kernel launch of class `key`,
    with that launch's algorithmic FLOPs and bytes (what the launch must read + write, each tensor once)"""
    __slots__ = ("ev",)

    def __init__(self, key, flops, nbytes):
        self.ev = None
        if PROFILE is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            PROFILE.setdefault(key, []).append((self.ev, float(flops), float(nbytes)))

    def __enter__(self):
        if self.ev is not None:
            self.ev[0].record()

    def __exit__(self, *exc):
        if self.ev is not None:
            self.ev[1].record()
        return False


def _nb(*tensors):
    return sum(t.numel() * t.element_size() for t in tensors if t is not None)


def _bn_in_args(b):
    if b is None:
        return _vp(0), _vp(0), _ci(0)
    return P(b.scale), P(b.shift), _ci(b.groups)


def _gate_args(gate):
    if gate is None:
        return _vp(0), _vp(0), _vp(0), _vp(0), _ci(0)
    out_relu, y, mean, invstd, groups = gate
    return P(out_relu), P(y), P(mean), P(invstd), _ci(groups)


class BnInput:
    """An activation that exists only as (pre-normalisation conv output y, per-group scale / shift): the consumer
    kernels apply relu(y * scale + shift) while they load it (dh_conv2d_fwd's in_scale, dh_conv2d_wgrad_bn_in)."""
    __slots__ = ("y", "scale", "shift", "groups")

    def __init__(self, y, scale, shift, groups):
        self.y, self.scale, self.shift, self.groups = y, scale, shift, groups

    @property
    def shape(self):
        return self.y.shape

    @property
    def device(self):
        return self.y.device

    @property
    def dtype(self):
        return self.y.dtype


def conv2d(x, wp, cout, ks=3, stride=1, pad=1, bias=None, residual=None, act=ACT_NONE, want_stats=False,
           want_preact=False, npix_valid=0, w_image_stride=0, out_hw=None, alg_flops=0, dilation=1, gate=None, w_frag=None):
    """gate = (out_relu | None, y_pre_bn, mean, invstd, groups): BN-backward gating of a data-gradient launch; the
    call then returns (g, partials) for bn_bwd_from_partials (see include/dahitra_hip.h).
    x may be a BnInput: BatchNorm-apply + ReLU happen on load."""
    if isinstance(x, Up4Input):
        assert ks == 3 and stride == 1 and pad == 1 and dilation == 1 and residual is None and gate is None and not want_preact
        return _conv3x3_up4(x, wp, cout, bias, act, want_stats)
    bn_in = x if isinstance(x, BnInput) else None
    if bn_in is not None:
        x = bn_in.y
    N, H, W, Cin = x.shape
    if gate is not None:
        want_stats = True
    if out_hw is None:
        OH = (H + 2 * pad - dilation * (ks - 1) - 1) // stride + 1
        OW = (W + 2 * pad - dilation * (ks - 1) - 1) // stride + 1
    else:
        OH, OW = out_hw
    cpad = wp.shape[-2]
    y = torch.empty(N, OH, OW, cout, dtype=x.dtype, device=x.device)
    pre = torch.empty_like(y) if want_preact else None
    stats = None
    if want_stats:
        nt = _lib.lib().dh_conv2d_fwd_num_tiles(_DT[x.dtype], N, OH, OW, Cin, ks, stride)
        stats = torch.empty(2, cpad, nt, dtype=torch.float32, device=x.device)      # [sum | sumsq][channel][tile]
    nt_ = 64 if cpad % 64 == 0 else (32 if cpad % 32 == 0 else 16)
    key = "conv_mfma<%s,ks%d,s%d,nt%d>" % ("bf16" if x.dtype == torch.bfloat16 else "f32", ks, stride, nt_)
    flops = alg_flops if alg_flops else 2.0 * N * OH * OW * cout * Cin * ks * ks
    fixed = (_ci(dt(x)), P(x), P(wp), P(y), P(bias), P(residual), P(stats), _ci(N), _ci(H), _ci(W),
             _ci(Cin), _ci(OH), _ci(OW), _ci(cout), _ci(cpad), _ci(ks), _ci(stride), _ci(pad), _ci(act), _ci(npix_valid),
             _cl(w_image_stride), P(pre), _ci(dilation), *_gate_args(gate), *_bn_in_args(bn_in), _ci(0), P(w_frag))
    with _Prof(key, flops, _nb(x, y, wp, residual, pre)):
        _call("dh_conv2d_fwd", *fixed, S())
    if REPLAY is not None and key == REPLAY["key"]:
        REPLAY["calls"].append((lambda stream, fixed=fixed: _call("dh_conv2d_fwd", *fixed, stream),
                                (x, wp, y, bias, residual, stats, pre, w_frag, gate, bn_in), flops))
    out = [y]
    if want_stats:
        out.append(stats)
    if want_preact:
        out.append(pre)
    return out[0] if len(out) == 1 else tuple(out)


class Up4Input:
    """nn.Upsample(4, 'bilinear')(|a - b|) of two [N, h, w, 32] bf16 maps WITHOUT the [N, 4h, 4w, 32] tensor (models/networks.py:
    383-389: the input of classifier.0): conv2d interpolates its tiles from a and b while it loads (dh_conv3x3_up4_fwd);
    conv2d_wgrad materialises the map.  `.shape` is the shape the upsampled tensor would have."""
    __slots__ = ("a", "b")

    def __init__(self, a, b):
        assert a.shape == b.shape and a.dim() == 4 and a.shape[-1] == 32 and a.dtype == torch.bfloat16
        assert a.is_contiguous() and b.is_contiguous()
        self.a, self.b = a, b

    @property
    def shape(self):
        n, h, w, c = self.a.shape
        return torch.Size((n, 4 * h, 4 * w, c))

    def materialize(self):
        return absdiff_upsample4(self.a, self.b)


def _conv3x3_up4(u, wp, cout, bias, act, want_stats):
    N, H, W, _ = u.shape
    assert cout == 32 and tuple(wp.shape) == (9, 32, 32) and wp.dtype == torch.bfloat16
    y = torch.empty(N, H, W, 32, dtype=torch.bfloat16, device=u.a.device)
    stats = None
    if want_stats:
        stats = torch.empty(2, 32, _lib.lib().dh_conv2d_fwd_num_tiles(_DT[torch.bfloat16], N, H, W, 32, 3, 1), dtype=torch.float32, device=y.device)
    key, flops = "conv_mfma<bf16,ks3,s1,nt32>", 2.0 * N * H * W * 32 * 32 * 9
    fixed = (P(u.a), P(u.b), P(wp), P(bias), _ci(act), P(y), P(stats), _ci(N), _ci(H), _ci(W))
    with _Prof(key, flops, _nb(u.a, u.b, y, wp)):
        _call("dh_conv3x3_up4_fwd", *fixed, S())
    if REPLAY is not None and key == REPLAY["key"]:
        REPLAY["calls"].append((lambda stream, fixed=fixed: _call("dh_conv3x3_up4_fwd", *fixed, stream), (u.a, u.b, wp, bias, y, stats), flops))
    return (y, stats) if want_stats else y


class SplitCat:
    """torch.cat([t[:B], t[B:]], channel) of a [2B, H, W, C] tensor WITHOUT the concatenated tensor (models/networks.py:1344:
    cat([a_128, b_128], 1) -- the two temporal streams are the two halves of one batch here).  conv3x3_split / conv2d_wgrad
    read the two halves in place; `.shape` is the shape the concatenation would have."""
    __slots__ = ("t", "B")

    def __init__(self, t):
        assert t.dim() == 4 and t.shape[0] % 2 == 0 and t.is_contiguous()
        self.t, self.B = t, t.shape[0] // 2

    @property
    def shape(self):
        n, h, w, c = self.t.shape
        return torch.Size((n // 2, h, w, 2 * c))

    @property
    def split_bytes(self):
        return self.t[self.B:].data_ptr() - self.t.data_ptr()

    def materialize(self):
        return cat_halves(self.t)


def conv3x3_split_supported(B, H, W, cin, cout, dtype):
    """can the 3x3 / stride 1 / pad 1 layer cin -> cout on B x H x W pixels run with a SplitCat input (forward and weight
    gradient) and a split data-gradient output?  (bf16, register-resident-weights + wave-specialised kernels; dahitra_hip.h)"""
    return dtype == torch.bfloat16 and os.environ.get("DAHITRA_NO_SPLIT_CAT", "0") != "1" and \
        bool(_lib.lib().dh_conv3x3_split_supported(_ci(B), _ci(H), _ci(W), _ci(cin), _ci(cout)))


def conv3x3_split(x, wp, w_frag, cout, want_stats=False, split_out=False, alg_flops=0):
    """3x3 / stride 1 / pad 1 convolution with a SplitCat input and / or (split_out) an output whose channel halves are the two
    batch halves of a [2N, H, W, cout / 2] tensor (the data gradient of a layer that read a SplitCat)."""
    xs = x if isinstance(x, SplitCat) else None
    N, H, W, Cin = x.shape
    xt = xs.t if xs is not None else x
    assert xt.dtype == torch.bfloat16 and w_frag is not None and (xs is not None or split_out)
    if split_out:
        y = torch.empty(2 * N, H, W, cout // 2, dtype=xt.dtype, device=xt.device)
        ysplit = y[N:].data_ptr() - y.data_ptr()
    else:
        y = torch.empty(N, H, W, cout, dtype=xt.dtype, device=xt.device)
        ysplit = 0
    stats = None
    if want_stats:
        nt = _lib.lib().dh_conv2d_fwd_num_tiles(_DT[xt.dtype], N, H, W, Cin, 3, 1)
        stats = torch.empty(2, cout, nt, dtype=torch.float32, device=xt.device)
    flops = alg_flops if alg_flops else 2.0 * N * H * W * cout * Cin * 9
    fixed = (P(xt), _cl(xs.split_bytes if xs is not None else 0), P(wp), P(w_frag), P(y), _cl(ysplit), P(stats), _ci(N), _ci(H),
             _ci(W), _ci(Cin), _ci(cout))
    key = "conv_mfma<bf16,ks3,s1,nt64>"
    with _Prof(key, flops, _nb(xt, y, wp)):
        _call("dh_conv3x3_split_fwd", *fixed, S())
    if REPLAY is not None and key == REPLAY["key"]:
        REPLAY["calls"].append((lambda stream, fixed=fixed: _call("dh_conv3x3_split_fwd", *fixed, stream), (xt, wp, w_frag, y, stats), flops))
    return (y, stats) if want_stats else y


HEAD_FWD = os.environ.get("DAHITRA_NO_HEAD_FWD", "0") != "1"


def conv3x3_head(x, wp, ncls, bias, w_oihw=None):
    """the class head: 3x3 / pad 1 convolution to `ncls` (<= 16) channels, fp32 NCHW logits written by the kernel itself.
    x may be a BnInput (BatchNorm-apply + ReLU on load).  With the fp32 master weights w_oihw [ncls, 32, 3, 3] at hand, bf16
    activations (or fp32 ones under f32_mma_mode 3) of 32 channels and ncls <= 2 go to dh_head_fwd (one MFMA per input pixel group for all nine taps, no halo)."""
    bn_in = x if isinstance(x, BnInput) else None
    if bn_in is not None:
        x = bn_in.y
    N, H, W, Cin = x.shape
    assert wp.shape[-2] == 16 and ncls <= 16
    out = torch.empty(N, ncls, H, W, dtype=torch.float32, device=x.device)
    if HEAD_FWD and w_oihw is not None and Cin == 32 and N * H * W * 128 < 2 ** 31 and _lib.lib().dh_head_fwd_supported(ncls, W) and \
            (x.dtype == torch.bfloat16 or (x.dtype == torch.float32 and get_f32_mma_mode() == 3)):
        assert w_oihw.shape == (ncls, 32, 3, 3) and w_oihw.dtype == torch.float32
        with _Prof("head_fwd", 2.0 * N * H * W * ncls * Cin * 9, _nb(x, out)):
            _call("dh_head_fwd", _ci(dt(x)), P(x), P(w_oihw), P(bias), _ci(ncls), *_bn_in_args(bn_in), P(out), _ci(N), _ci(H), _ci(W), S())
        return out
    key = "conv_mfma<%s,ks3,s1,nt16>" % ("bf16" if x.dtype == torch.bfloat16 else "f32")
    with _Prof(key, 2.0 * N * H * W * ncls * Cin * 9, _nb(x, out, wp)):
        _call("dh_conv3x3_head_fwd", _ci(dt(x)), P(x), P(wp), P(bias), _ci(N), _ci(H), _ci(W), _ci(Cin), _ci(ncls),
              *_bn_in_args(bn_in), P(out), S())
    return out


def rows_view(x2d):
    """[rows, C] -> the [1, ceil(rows/16), 16, C] image view used for linear layers (no copy when
    rows % 16 == 0; otherwise the tail rows are masked through npix_valid)."""
    rows, C = x2d.shape
    return rows, cdiv(rows, 16), C


def linear(x2d, wp, cout, bias=None, residual=None, act=ACT_NONE, want_preact=False, images=1, w_image_stride=0):
    """x2d [rows, Cin] (rows = images * rows_per_image).  Returns [rows, cout]."""
    rows, Cin = x2d.shape
    rpi = rows // images
    Hh = cdiv(rpi, 16)
    cpad = wp.shape[-2]
    y = torch.empty(rows, cout, dtype=x2d.dtype, device=x2d.device)
    pre = torch.empty_like(y) if want_preact else None
    _call("dh_conv2d_fwd", _ci(dt(x2d)), P(x2d), P(wp), P(y), P(bias), P(residual), _vp(0), _ci(images), _ci(Hh),
          _ci(16), _ci(Cin), _ci(Hh), _ci(16), _ci(cout), _ci(cpad), _ci(1), _ci(1), _ci(0), _ci(act), _ci(rpi),
          _cl(w_image_stride), P(pre), _ci(1), *_gate_args(None), *_bn_in_args(None), _ci(0), _vp(0), S())
    # note: with rows_per_image % 16 != 0 the image stride used by the kernel (Hh*16 rows) would differ
    # from rpi; callers guarantee rpi % 16 == 0 whenever images > 1.
    assert images == 1 or rpi % 16 == 0
    return (y, pre) if want_preact else y


def conv2d_wgrad(x, dy, dw, ks, stride, pad, accumulate=False, groups=1, use_tr=True, cout_real=0, cin=0, dilation=1,
                 defer=True):
    """dw (OIHW fp32, or [N, Cout, Cin] when groups == N) (+)= weight gradient.
    cin > 0: use only the first `cin` channels of x (x keeps its own channel pitch).
    x may be a BnInput (gradient against relu(y * scale + shift), computed on load)."""
    if isinstance(x, BnInput):
        return _conv2d_wgrad_bn_in(x, dy, dw, ks, stride, pad, accumulate, use_tr, cout_real, dilation, defer)
    if isinstance(x, SplitCat):
        return _conv2d_wgrad_split(x, dy, dw, ks, stride, pad, accumulate, defer)
    if isinstance(x, Up4Input):
        # (the weight gradient with the interpolation on load was built and measured -- 94 us against 67 us for the plain
        # kernel, and not reproducible from run to run at >= 512 workgroups; removed, DESIGN.md section 6e: the map is formed here)
        x = x.materialize()
    N, H, W, pitch = x.shape
    Cin = cin if cin else pitch
    _, OH, OW, Cout = dy.shape
    L = _lib.lib()
    nbytes = L.dh_conv2d_wgrad_workspace_size(N, OH, OW, Cin, Cout, ks, groups)
    plan = _WGRAD_PLAN
    if plan is not None and groups == 1 and defer:      # defer=False: the caller reads dw right after this call
        # deferred: partial slabs into this layer's persistent workspace, summed later by plan.run()
        ws = plan.slab(nbytes)
        sk = ctypes.c_int(0)
        with _Prof("conv_wgrad<%s,ks%d,s%d>" % ("bf16" if x.dtype == torch.bfloat16 else "f32", ks, stride),
                   2.0 * N * OH * OW * Cout * Cin * ks * ks, _nb(x, dy)):
            _call("dh_conv2d_wgrad_partial", _ci(dt(x)), P(x), P(dy), P(dw), _ci(int(accumulate)), _ci(N), _ci(H), _ci(W),
                  _ci(Cin), _ci(OH), _ci(OW), _ci(Cout), _ci(ks), _ci(stride), _ci(pad), _ci(1), _ci(0), _ci(int(use_tr)),
                  _ci(cout_real), _ci(pitch), _ci(dilation), P(ws), ctypes.byref(sk), S())
        plan.hold(x, dy)
        if sk.value > 0:
            plan.add(ws, dw, sk.value, ks * ks, Cout, cout_real if cout_real else Cout, Cin, accumulate)
        return
    ws = workspace(nbytes, x.device)
    _call("dh_conv2d_wgrad", _ci(dt(x)), P(x), P(dy), P(dw), _ci(int(accumulate)), _ci(N), _ci(H), _ci(W), _ci(Cin),
          _ci(OH), _ci(OW), _ci(Cout), _ci(ks), _ci(stride), _ci(pad), _ci(groups), _ci(0), _ci(int(use_tr)),
          _ci(cout_real), _ci(pitch), _ci(dilation), P(ws), S())


def _conv2d_wgrad_split(xs, dy, dw, ks, stride, pad, accumulate, defer):
    """weight gradient against a SplitCat input (3x3 / stride 1 / pad 1, bf16): the two tensors are read in place"""
    assert ks == 3 and stride == 1 and pad == 1
    N, H, W, Cin = xs.shape
    Cout = dy.shape[-1]
    nbytes = _lib.lib().dh_conv2d_wgrad_workspace_size(N, H, W, Cin, Cout, 3, 1)
    plan, own = (_WGRAD_PLAN if defer else None), False
    if plan is None:
        plan, own = WgradPlan(dy.device), True
        plan.__enter__()
    try:
        ws = plan.slab(nbytes)
        sk = ctypes.c_int(0)
        with _Prof("conv_wgrad<bf16,ks3,s1>", 2.0 * N * H * W * Cout * Cin * 9, _nb(xs.t, dy)):
            _call("dh_conv2d_wgrad_split", P(xs.t), _cl(xs.split_bytes), P(dy), P(dw), _ci(int(accumulate)), _ci(N), _ci(H), _ci(W),
                  _ci(Cin), _ci(Cout), P(ws), ctypes.byref(sk), S())
        plan.hold(xs.t, dy)
        plan.add(ws, dw, sk.value, 9, Cout, Cout, Cin, accumulate)
        if own:
            plan.run()
    finally:
        if own:
            plan.__exit__(*sys.exc_info())


def _conv2d_wgrad_bn_in(b, dy, dw, ks, stride, pad, accumulate, use_tr, cout_real, dilation, defer):
    x = b.y
    N, H, W, Cin = x.shape
    _, OH, OW, Cout = dy.shape
    nbytes = _lib.lib().dh_conv2d_wgrad_workspace_size(N, OH, OW, Cin, Cout, ks, 1)
    plan = _WGRAD_PLAN if defer else None
    ws = plan.slab(nbytes) if plan is not None else workspace(nbytes, x.device)
    sk = ctypes.c_int(0)
    with _Prof("conv_wgrad<%s,ks%d,s%d>" % ("bf16" if x.dtype == torch.bfloat16 else "f32", ks, stride),
               2.0 * N * OH * OW * Cout * Cin * ks * ks, _nb(x, dy)):
        _call("dh_conv2d_wgrad_bn_in", _ci(dt(x)), P(x), P(dy), P(dw), _ci(int(accumulate)), _ci(N), _ci(H), _ci(W), _ci(Cin),
              _ci(OH), _ci(OW), _ci(Cout), _ci(ks), _ci(stride), _ci(pad), _ci(int(use_tr)), _ci(cout_real), _ci(dilation),
              P(b.scale), P(b.shift), _ci(b.groups), P(ws), ctypes.byref(sk) if plan is not None else None, S())
    if plan is not None:
        plan.hold(x, dy, b.scale, b.shift)
    if plan is not None and sk.value > 0:
        plan.add(ws, dw, sk.value, ks * ks, Cout, cout_real if cout_real else Cout, Cin, accumulate)


# ---- conv3x3(nearest-upsample-x2(x)) with 32 output channels as four 2x2 phase convolutions (models/networks.py:251-256) ----
def pack_phase_weights(w, bias, dtype):
    """w OIHW fp32 [32, Cin, 3, 3] -> (fwd [4, 128, Cin], dgrad [4, Cin, 128], bias4 [128])"""
    O, I = w.shape[0], w.shape[1]
    assert O == 32 and tuple(w.shape[2:]) == (3, 3)
    fwd = torch.empty(4, 128, I, dtype=dtype, device=w.device)
    dg = torch.empty(4, I, 128, dtype=dtype, device=w.device)
    b4 = torch.empty(128, dtype=torch.float32, device=w.device)
    _call("dh_pack_phase_weights", _ci(_DT[dtype]), P(w), P(bias), _ci(I), P(fwd), P(dg), P(b4), S())
    return fwd, dg, b4


def conv_up2_fwd(x, wfwd, bias4, act=ACT_NONE):
    """act(conv3x3(nearest-x2(x)) + bias): x [N, H, W, Cin] -> [N, 2H, 2W, 32]; the upsampled tensor is never materialised"""
    N, H, W, Cin = x.shape
    y = torch.empty(N, 2 * H, 2 * W, 32, dtype=x.dtype, device=x.device)
    key = "conv_phase<%s,up2_fwd>" % ("bf16" if x.dtype == torch.bfloat16 else "f32")         # its own class: KS = 2 launches
    with _Prof(key, 2.0 * N * 4 * H * W * 32 * Cin * 9, _nb(x, y, wfwd)):                       # algorithmic FLOPs (of the 3x3)
        _call("dh_conv2d_fwd", _ci(dt(x)), P(x), P(wfwd), P(y), P(bias4), _vp(0), _vp(0), _ci(N), _ci(H), _ci(W), _ci(Cin),
              _ci(H), _ci(W), _ci(128), _ci(128), _ci(2), _ci(1), _ci(1), _ci(act), _ci(0), _cl(0), _vp(0), _ci(1),
              *_gate_args(None), *_bn_in_args(None), _ci(1), _vp(0), S())
    return y


def pack_s2_dgrad_phase_weights(w, dtype):
    """OIHW fp32 [Co, Ci, 3, 3] of a stride-2 conv -> [4, 4 * Ci, Co] phase weights of its data gradient"""
    Co, Ci = w.shape[0], w.shape[1]
    out = torch.empty(4, 4 * Ci, Co, dtype=dtype, device=w.device)
    _call("dh_pack_s2_dgrad_phase_weights", _ci(_DT[dtype]), P(w), _ci(Co), _ci(Ci), P(out), S())
    return out


def conv3x3s2_dgrad(dy, wphase, cin, coarse_residual=None, alg_flops=0):
    """data gradient of a 3x3 / stride-2 / pad-1 convolution from dy [N, OH, OW, Co] straight to [N, 2 OH, 2 OW, cin]: four
    output-parity phases (1 + 2 + 2 + 4 of the 9 taps), no zero-inserted tensor; coarse_residual [N, OH, OW, cin] (the data
    gradient of a parallel 1x1 stride-2 convolution) lands on the even-even positions"""
    N, OH, OW, Co = dy.shape
    dx = torch.empty(N, 2 * OH, 2 * OW, cin, dtype=dy.dtype, device=dy.device)
    key = "conv_phase<%s,s2_dgrad>" % ("bf16" if dy.dtype == torch.bfloat16 else "f32")
    with _Prof(key, alg_flops if alg_flops else 2.0 * N * OH * OW * Co * cin * 9, _nb(dy, dx, wphase, coarse_residual)):
        _call("dh_conv2d_fwd", _ci(dt(dy)), P(dy), P(wphase), P(dx), _vp(0), P(coarse_residual), _vp(0), _ci(N), _ci(OH),
              _ci(OW), _ci(Co), _ci(OH), _ci(OW), _ci(4 * cin), _ci(4 * cin), _ci(2), _ci(1), _ci(1), _ci(ACT_NONE), _ci(0),
              _cl(0), _vp(0), _ci(1), *_gate_args(None), *_bn_in_args(None), _ci(1), _vp(0), S())
    return dx


def conv_up2_dgrad(dy, wdgrad, cin):
    """gradient of conv_up2_fwd with respect to x: dy [N, 2H, 2W, 32] -> [N, H, W, cin]"""
    N, H2, W2, C = dy.shape
    assert C == 32 and H2 % 2 == 0 and W2 % 2 == 0 and (cin % 64 == 0 or cin == 32)
    H, W = H2 // 2, W2 // 2
    dx = torch.empty(N, H, W, cin, dtype=dy.dtype, device=dy.device)
    key = "conv_phase<%s,up2_dgrad>" % ("bf16" if dy.dtype == torch.bfloat16 else "f32")
    with _Prof(key, 2.0 * N * H2 * W2 * 32 * cin * 9, _nb(dy, dx, wdgrad)):
        _call("dh_conv2d_fwd", _ci(dt(dy)), P(dy), P(wdgrad), P(dx), _vp(0), _vp(0), _vp(0), _ci(N), _ci(H), _ci(W), _ci(128),
              _ci(H), _ci(W), _ci(cin), _ci(cin), _ci(2), _ci(1), _ci(1), _ci(ACT_NONE), _ci(0), _cl(0), _vp(0), _ci(1),
              *_gate_args(None), *_bn_in_args(None), _ci(2), _vp(0), S())
    return dx


def conv_up2_wgrad(x, dy, dw, accumulate=True, use_tr=True):
    """dw (OIHW fp32 [32, Cin, 3, 3]) (+)= weight gradient of conv_up2_fwd: four per-phase 2x2 gradients (one launch), their
    split-K reduce (with the pass's other reduces when a WgradPlan is active), then the tap combine"""
    N, H, W, Cin = x.shape
    L = _lib.lib()
    plan, own = _WGRAD_PLAN, False
    if plan is None:
        plan, own = WgradPlan(x.device), True
        plan.__enter__()
    try:
        ws = plan.slab(L.dh_conv2d_wgrad_phase_workspace_size(N, H, W, Cin))
        dwab = plan.slab(4 * 32 * Cin * 4 * 4).view(torch.float32)
        sk = ctypes.c_int(0)
        with _Prof("conv_wgrad<%s,ks3,s1>" % ("bf16" if x.dtype == torch.bfloat16 else "f32"),
                   2.0 * N * 4 * H * W * 32 * Cin * 9, _nb(x, dy)):
            _call("dh_conv2d_wgrad_phase", _ci(dt(x)), P(x), P(dy), _ci(N), _ci(H), _ci(W), _ci(Cin), _ci(int(use_tr)), P(ws),
                  ctypes.byref(sk), S())
        per = sk.value * 4 * 32 * Cin * 4            # bytes of one phase's slabs
        for ph in range(4):
            plan.add(ws[ph * per:], dwab[ph * 32 * Cin * 4:], sk.value, 4, 32, 32, Cin, False)
        plan.post.append(lambda: _call("dh_phase_wgrad_combine", P(dwab), P(dw), _ci(Cin), _ci(int(accumulate)), S()))
        if own:
            plan.run()
    finally:
        if own:
            plan.__exit__()


def linear_wgrad(x2d, dy2d, dw, accumulate=False, images=1, per_image=False, use_tr=True):
    rows, Cin = x2d.shape
    Cout = dy2d.shape[1]
    rpi = rows // images
    Hh = cdiv(rpi, 16)
    groups = images if per_image else 1
    L = _lib.lib()
    nbytes = L.dh_conv2d_wgrad_workspace_size(images, Hh, 16, Cin, Cout, 1, groups)
    ws = workspace(nbytes, x2d.device)
    assert images == 1 or rpi % 16 == 0
    _call("dh_conv2d_wgrad", _ci(dt(x2d)), P(x2d), P(dy2d), P(dw), _ci(int(accumulate)), _ci(images), _ci(Hh), _ci(16),
          _ci(Cin), _ci(Hh), _ci(16), _ci(Cout), _ci(1), _ci(1), _ci(0), _ci(groups), _ci(rpi), _ci(int(use_tr)),
          _ci(0), _ci(0), _ci(1), P(ws), S())


def add_coarse_(x, coarse):
    """x[n, 2y, 2x, :] += coarse[n, y, x, :] in place (the coarse-grid data gradient of a 1x1 stride-2 convolution)"""
    N, H, W, C = x.shape
    n2, OH, OW, c2 = coarse.shape
    assert n2 == N and c2 == C and x.dtype == coarse.dtype and x.is_contiguous() and coarse.is_contiguous()
    _call("dh_add_coarse", _ci(dt(x)), P(x), P(coarse), _ci(N), _ci(OH), _ci(OW), _ci(H), _ci(W), _ci(C), S())
    return x


def zero_insert2(dy, H, W):
    N, OH, OW, C = dy.shape
    z = torch.empty(N, H, W, C, dtype=dy.dtype, device=dy.device)
    _call("dh_zero_insert2", _ci(dt(dy)), P(dy), P(z), _ci(N), _ci(OH), _ci(OW), _ci(H), _ci(W), _ci(C), S())
    return z


# ---- stem ----------------------------------------------------------------------------------------
def stem_space_to_depth(x_nchw, dtype):
    N, C, H, W = x_nchw.shape
    assert C == 3
    cp = chunk_channels(dtype)
    y = torch.empty(N, H // 2, W // 2, cp, dtype=dtype, device=x_nchw.device)
    _call("dh_stem_space_to_depth", _ci(_DT[dtype]), P(x_nchw), P(y), _ci(N), _ci(H), _ci(W), _ci(cp), S())
    return y


def stem_pack_weight(w, dtype, out_scale=None):
    O = w.shape[0]
    cp = chunk_channels(dtype)
    out = torch.empty(16, O, cp, dtype=dtype, device=w.device)
    _call("dh_stem_pack_weight", _ci(_DT[dtype]), P(w), P(out_scale), P(out), _ci(O), _ci(cp), S())
    return out


def stem7_fwd(x1, x2, w, out_scale=None, bias=None, relu=False, want_stats=False, want_xs=False, groups=1):
    """7x7/2 stem on the NCHW fp32 images of both streams (x2 may be None) -> (y bf16 NHWC [N,H/2,W/2,64], stats, xs16)"""
    B, C, H, W = x1.shape
    assert C == 3 and x1.dtype == torch.float32 and x1.is_contiguous() and (x2 is None or (x2.shape == x1.shape and x2.is_contiguous()))
    N = B if x2 is None else 2 * B
    y = torch.empty(N, H // 2, W // 2, 64, dtype=torch.bfloat16, device=x1.device)
    nt = _lib.lib().dh_stem7_fwd_num_slots(N, H, W, groups)
    stats = torch.empty(2, 64, nt, dtype=torch.float32, device=x1.device) if want_stats else None
    xs = torch.empty(N, H // 2, W // 2, 16, dtype=torch.bfloat16, device=x1.device) if want_xs else None
    with _Prof("stem7_fwd", 2.0 * N * (H // 2) * (W // 2) * 64 * 147, _nb(x1, x2, y, xs)):
        _call("dh_stem7_fwd", P(x1), P(x2), _ci(B), _ci(N), _ci(H), _ci(W), P(w), P(out_scale), P(bias), _ci(int(relu)), P(y),
              P(stats), _ci(groups), P(xs), S())
    return y, stats, xs


def stem_pool_bn_bwd(arg, dpool, y, scale, shift, mean, invstd, gamma, dgamma, dbeta, groups, accumulate=True, extra=None):
    """maxpool backward + ReLU mask + BatchNorm-backward sums of the stem's tail in one pass -> (d masked, coef [groups,3,C]).
    extra: a second gradient of the pre-pool activation (same shape as y), added before the mask"""
    N, H, W, C = y.shape
    d = torch.empty_like(y)
    coef = torch.empty(groups, 3, C, dtype=torch.float32, device=y.device)
    L = _lib.lib()
    ws = workspace(L.dh_stem_pool_bn_bwd_workspace_size(C, groups), y.device)
    assert extra is None or (extra.shape == y.shape and extra.dtype == y.dtype)
    with _Prof("bn_bwd", 0, _nb(arg, dpool, y, d, extra)):
        _call("dh_stem_pool_bn_bwd_plus", P(arg), P(dpool), P(extra), P(y), P(scale), P(shift), P(mean), P(invstd), P(gamma), _ci(N), _ci(H),
              _ci(W), _ci(C), _ci(groups), P(d), P(coef), P(dgamma), P(dbeta), _ci(int(accumulate)), P(ws), S())
    return d, coef


def stem_wgrad(x_s2d, dy, dw, accumulate=False, use_tr=True, bn=None):
    """bn = (y, coef, groups): dy is the masked gradient of the BatchNorm output; BatchNorm backward is applied on load"""
    N, H2, W2, cp = x_s2d.shape
    O = dy.shape[-1]
    dw2 = torch.empty(O, 16, 4, 4, dtype=torch.float32, device=dy.device)     # 12 real + 4 zero channels
    if bn is not None:
        y, coef, groups = bn
        assert cp == 16 and O == 64 and y.shape == dy.shape
        L = _lib.lib()
        ws = workspace(L.dh_conv2d_wgrad_workspace_size(N, H2, W2, 16, O, 4, 1), dy.device)
        with _Prof("conv_wgrad<bf16,ks4,s1>", 2.0 * N * H2 * W2 * O * 16 * 16, _nb(x_s2d, dy, y)):
            _call("dh_stem_wgrad_bn", P(x_s2d), P(dy), P(y), P(coef), _ci(groups), _ci(N), _ci(H2), _ci(W2), P(dw2),
                  _ci(int(use_tr)), P(ws), S())
    else:
        conv2d_wgrad(x_s2d, dy, dw2, ks=4, stride=1, pad=2, use_tr=use_tr, cin=16, defer=False)     # dw2 is unpacked next
    _call("dh_stem_unpack_grad", P(dw2), P(dw), _ci(O), _ci(16), _ci(int(accumulate)), S())


# ---- normalisation -------------------------------------------------------------------------------
def bn_finalize(stats, C, groups, count, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, nbt=None):
    """nbt: the layer's int64 num_batches_tracked buffer (device), incremented by `groups` in the same launch"""
    _, cp, nt = stats.shape
    dev = stats.device
    mean = torch.empty(groups, C, dtype=torch.float32, device=dev)
    invstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    _call("dh_bn_finalize", P(stats), _ci(nt), _ci(cp), _ci(C), _ci(groups), _cd(float(count)), P(gamma), P(beta),
          P(running_mean), P(running_var), _cf(momentum), _cf(eps), P(mean), P(invstd), P(scale), P(shift), P(nbt), S())
    return mean, invstd, scale, shift


def bn_eval_params(gamma, beta, rm, rv, eps=1e-5):
    C = gamma.numel()
    scale = torch.empty(1, C, dtype=torch.float32, device=gamma.device)
    shift = torch.empty_like(scale)
    _call("dh_bn_eval_params", P(gamma), P(beta), P(rm), P(rv), _cf(eps), _ci(C), P(scale), P(shift), S())
    return scale, shift


RELU_BITS = os.environ.get("DAHITRA_NO_RELU_BITS", "0") != "1"      # A/B switch of bn_apply(want_bits=True)


def bn_apply(x, scale, shift, groups=1, act=ACT_NONE, residual=None, want_bits=False):
    """want_bits (ReLU): also returns the ReLU mask of y as one byte per 16-byte piece (dh_bn_apply_bits) -- what bn_bwd(bits=...)
    reads in place of the post-activation tensor; None with DAHITRA_NO_RELU_BITS=1"""
    C = x.shape[-1]
    npix = x.numel() // C
    y = torch.empty_like(x)
    bits = None
    v = 8 if x.dtype == torch.bfloat16 else 4
    if want_bits and RELU_BITS and act == ACT_RELU and C % v == 0:
        bits = torch.empty(x.numel() // v, dtype=torch.uint8, device=x.device)
    with _Prof("bn_apply", 0, _nb(x, residual, y)):
        if bits is not None:
            _call("dh_bn_apply_bits", _ci(dt(x)), P(x), P(residual), P(y), P(scale), P(shift), _cl(npix), _ci(C), _ci(groups),
                  _ci(act), P(bits), S())
        else:
            _call("dh_bn_apply", _ci(dt(x)), P(x), P(residual), P(y), P(scale), P(shift), _cl(npix), _ci(C), _ci(groups),
                  _ci(act), S())
    return (y, bits) if want_bits else y


_BN_SYNC = {}
BN_BWD_PERSIST = os.environ.get("DAHITRA_NO_PERSIST_BN", "0") != "1"      # True: where faster; "force": wherever supported (tests)
if os.environ.get("DAHITRA_PERSIST_BN_FORCE", "0") == "1":
    BN_BWD_PERSIST = "force"
_PERSIST_BLOCK = 0          # > 0: inside no_persist_bn() -- another stream may run beside the launches made here
BN_PERSIST_LAUNCHES = 0     # persistent launches issued (or recorded into a graph) by this process: what tests assert the guard on


class no_persist_bn:
    """`with ops.no_persist_bn():` BatchNorm backward takes the two-pass kernels.  The persistent form (dh_bn_bwd_persist)
    holds a device-wide barrier across 256 one-per-CU workgroups: it must not be launched where a kernel of another
    stream -- the RCCL all-reduce of the overlapped data-parallel step, a side-stream weight gradient -- may hold CUs."""

    def __enter__(self):
        global _PERSIST_BLOCK
        _PERSIST_BLOCK += 1

    def __exit__(self, *exc):
        global _PERSIST_BLOCK
        _PERSIST_BLOCK -= 1
        return False


def bn_sync_words(device):
    """the zeroed state of dh_bn_bwd_persist's device-wide barrier + fixed-point accumulators: one block per (device, stream)
    -- two streams must never share the counters -- that lives forever (captured HIP graphs point at it)"""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    t = _BN_SYNC.get(key)
    if t is None:
        t = _BN_SYNC[key] = torch.zeros(8192, dtype=torch.int32, device=device)
    return t


def bn_persist_check(device=None):
    """reads and clears the error word of every persistent-BatchNorm sync block of `device` (all devices if None): raises
    HipLibraryError when a launch since the last check timed out at its device-wide barrier or met a non-finite sum (the
    affected step's dgamma / dbeta are NaN by then).  One 4-byte device-to-host copy per block: call it every N steps."""
    bad = []
    for (dev, stream), t in list(_BN_SYNC.items()):
        if device is not None and dev != str(device):
            continue
        word = _lib.lib().dh_bn_bwd_persist_status(P(t), _vp(torch.cuda.current_stream(t.device).cuda_stream))
        if word:
            bad.append((dev, stream, word))
    if bad:
        raise _lib.HipLibraryError(
            "dahitra_amd: the persistent BatchNorm backward failed (device, stream, word): %s -- bit 0: its device-wide "
            "barrier timed out (another stream's kernel kept workgroups out: set DAHITRA_NO_PERSIST_BN=1 or keep such "
            "launches inside ops.no_persist_bn()), bit 1: non-finite gradient sums; the gradients of that step are NaN" % bad)


def bn_bwd(dout, out_relu, x, mean, invstd, gamma, dgamma, dbeta, groups=1, accumulate=False, want_dres=False,
           mask_scale=None, mask_shift=None, bits=None):
    """bits: the ReLU mask bytes of bn_apply(want_bits=True), used in place of out_relu"""
    C = x.shape[-1]
    npix = x.numel() // C
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    L = _lib.lib()
    ws = workspace(L.dh_bn_bwd_workspace_size(_cl(npix), C, groups), x.device)
    if BN_BWD_PERSIST and not _PERSIST_BLOCK and \
            (L.dh_bn_bwd_persist_preferred if BN_BWD_PERSIST is True else L.dh_bn_bwd_persist_supported)(
            _ci(dt(x)), _cl(npix), C, groups):
        # one persistent launch, tensors held on chip across a device-wide barrier: every tensor is read once
        global BN_PERSIST_LAUNCHES
        BN_PERSIST_LAUNCHES += 1
        with _Prof("bn_bwd", 0, _nb(dout, bits if bits is not None else out_relu, x) + _nb(dx, dres)):
            if bits is not None:
                _call("dh_bn_bwd_persist_bits", P(dout), P(bits), P(x), P(mean), P(invstd), P(gamma), _cl(npix), _ci(C),
                      _ci(groups), P(dx), P(dres), P(dgamma), P(dbeta), _ci(int(accumulate)), P(ws), P(bn_sync_words(x.device)), S())
            else:
                _call("dh_bn_bwd_persist", P(dout), P(out_relu), P(x), P(mean), P(invstd), P(gamma), _cl(npix), _ci(C),
                      _ci(groups), P(dx), P(dres), P(dgamma), P(dbeta), _ci(int(accumulate)), P(mask_scale), P(mask_shift),
                      P(ws), P(bn_sync_words(x.device)), S())
        return (dx, dres) if want_dres else dx
    # two passes (reduce, apply): dout / x (/ out) are read twice, dx (/ dres) written once
    with _Prof("bn_bwd", 0, 2 * _nb(dout, bits if bits is not None else out_relu, x) + _nb(dx, dres)):
        if bits is not None:
            _call("dh_bn_bwd_bits", _ci(dt(x)), P(dout), P(bits), P(x), P(mean), P(invstd), P(gamma), _cl(npix), _ci(C),
                  _ci(groups), P(dx), P(dres), P(dgamma), P(dbeta), _ci(int(accumulate)), P(ws), S())
        else:
            _call("dh_bn_bwd", _ci(dt(x)), P(dout), P(out_relu), P(x), P(mean), P(invstd), P(gamma), _cl(npix), _ci(C),
                  _ci(groups), P(dx), P(dres), P(dgamma), P(dbeta), _ci(int(accumulate)), P(mask_scale), P(mask_shift), P(ws), S())
    return (dx, dres) if want_dres else dx


def bn_bwd_from_partials(g, x, partial, mean, invstd, gamma, dgamma, dbeta, groups=1, accumulate=False):
    """BN backward fed by a gated data-gradient launch (conv2d(..., gate=...)): g already carries the ReLU mask"""
    C = x.shape[-1]
    npix = x.numel() // C
    _, cp, nt = partial.shape
    assert cp == C and g.shape == x.shape
    dx = torch.empty_like(x)
    ws = workspace(groups * 2 * C * 4, x.device)
    _call("dh_bn_bwd_from_partials", _ci(dt(x)), P(g), P(x), P(partial), _ci(nt), P(mean), P(invstd), P(gamma), _cl(npix),
          _ci(C), _ci(groups), P(dx), P(dgamma), P(dbeta), _ci(int(accumulate)), P(ws), S())
    return dx


def layernorm(x2d, gamma, beta, eps=1e-5, want_stats=True):
    rows, C = x2d.shape
    y = torch.empty_like(x2d)
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x2d.device) if want_stats else None
    _call("dh_layernorm_fwd", _ci(dt(x2d)), P(x2d), P(gamma), P(beta), P(y), P(stats), _cl(rows), _ci(C), _cf(eps), S())
    return (y, stats) if want_stats else y


def layernorm_bwd(dy, x2d, stats, gamma, dgamma, dbeta, dx_add=None, accumulate=False):
    rows, C = x2d.shape
    dx = torch.empty_like(x2d)
    L = _lib.lib()
    ws = workspace(L.dh_layernorm_bwd_workspace_size(_cl(rows)), x2d.device)
    _call("dh_layernorm_bwd", _ci(dt(x2d)), P(dy), P(x2d), P(stats), P(gamma), P(dx), P(dx_add), P(dgamma), P(dbeta),
          _ci(int(accumulate)), _cl(rows), _ci(C), P(ws), S())
    return dx


# ---- pointwise -----------------------------------------------------------------------------------
def nchw_to_nhwc(x, dtype, cpad=0):
    N, C, H, W = x.shape
    cp = max(C, cpad)
    y = torch.empty(N, H, W, cp, dtype=dtype, device=x.device)
    _call("dh_nchw_to_nhwc", _ci(_DT[dtype]), P(x), P(y), _ci(N), _ci(C), _cl(H * W), _ci(cp), S())
    return y


def head_dgrad3x3(dy, w_oihw, ncls, relu_out=None):
    """data gradient of the 3x3 class head (32 -> ncls): dy [N,H,W,CP] with one 16-byte piece per pixel -> [N,H,W,32].
    relu_out: the head's input is the output of a ReLU (given here); the gradient comes back already masked by it"""
    N, H, W, CP = dy.shape
    assert w_oihw.shape == (ncls, 32, 3, 3) and w_oihw.dtype == torch.float32
    dx = torch.empty(N, H, W, 32, dtype=dy.dtype, device=dy.device)
    if relu_out is not None:
        if dy.dtype == torch.bfloat16 and CP == 8 and ncls <= 2:
            _call("dh_head_dgrad3x3_relu", P(dy), P(w_oihw), _ci(ncls), P(relu_out), P(dx), _ci(N), _ci(H), _ci(W), S())
            return dx
        return act_bwd(head_dgrad3x3(dy, w_oihw, ncls), relu_out, ACT_RELU)
    _call("dh_head_dgrad3x3", _ci(dt(dy)), P(dy), _ci(CP), P(w_oihw), _ci(ncls), P(dx), _ci(N), _ci(H), _ci(W), S())
    return dx


def head_dgrad3x3_bn(dy, w_oihw, ncls, y, scale, shift, mean, invstd, groups):
    """head_dgrad3x3 gated for the BatchNorm + ReLU behind its 32 output channels: returns (g masked [N,H,W,32], partials
    [2, 32, blocks]) for bn_bwd_from_partials"""
    N, H, W, CP = dy.shape
    assert w_oihw.shape == (ncls, 32, 3, 3) and dy.dtype == torch.bfloat16 and CP == 8 and ncls <= 2 and y.shape == (N, H, W, 32)
    g = torch.empty(N, H, W, 32, dtype=dy.dtype, device=dy.device)
    nb = _lib.lib().dh_head_dgrad3x3_bn_blocks(N, H, W, groups)
    partial = torch.empty(2, 32, nb, dtype=torch.float32, device=dy.device)
    with _Prof("bn_bwd", 0, _nb(dy, y, g)):
        _call("dh_head_dgrad3x3_bn", P(dy), P(w_oihw), _ci(ncls), P(y), P(scale), P(shift), P(mean), P(invstd), _ci(groups), P(g),
              P(partial), _ci(N), _ci(H), _ci(W), S())
    return g, partial


def head_dlogits_pack(dl_nchw, dtype=torch.bfloat16):
    """dlogits [N, ncls <= 8, H, W] fp32 -> the zero-bordered class map of head_bn_bwd / head_relu_bwd (int32 words of bf16
    pairs, _head_dlp_shape); dtype float32 (the bf16x3 mode): the plane of bf16 heads, then the plane of bf16 remainders"""
    N, C, H, W = dl_nchw.shape
    assert C <= 8 and dl_nchw.dtype == torch.float32 and dl_nchw.is_contiguous()
    dlp = torch.empty(_head_dlp_shape(N, H, W, C, dtype), dtype=torch.int32, device=dl_nchw.device)
    _call("dh_head_dlogits_pack", _ci(_DT[dtype]), P(dl_nchw), _ci(N), _ci(C), _ci(H), _ci(W), P(dlp), S())
    return dlp


def _head_dlp_shape(N, H, W, ncls, dtype):
    """[N, H + 2, W + 2] + (planes * words per piece,): a piece = 2 classes (one word) for ncls <= 2, 8 classes (4 words) above;
    one plane for bf16, two (heads, remainders) for the split fp32 form"""
    words = (1 if ncls <= 2 else 4) * (1 if dtype == torch.bfloat16 else 2)
    return (N, H + 2, W + 2) if words == 1 else (N, H + 2, W + 2, words)


def _head_dlp_ok(dlp, y, ncls):
    N, H, W, C = y.shape
    return dlp.dtype == torch.int32 and tuple(dlp.shape) == _head_dlp_shape(N, H, W, ncls, y.dtype) and C == 32 and \
        y.dtype in (torch.bfloat16, torch.float32)


def head_bn_bwd(dlp, w_oihw, ncls, y, scale, shift, mean, invstd, gamma, dgamma, dbeta, groups, accumulate=True, dw=None, db=None):
    """class head's data gradient + the backward of the BatchNorm + ReLU behind it without the gradient tensor in between
    (dh_head_bn_bwd): dlp = head_dlogits_pack(dlogits, y.dtype) -> the gradient of the pre-BatchNorm activation y [N,H,W,32].
    dw [ncls,32,3,3] / db [ncls]: the head convolution's own weight / bias gradient from the same pass.  y fp32: the bf16x3
    mode's arithmetic (three split bf16 products) -- not for the exact fp32 mode"""
    assert (dw is None) == (db is None) and (dw is None or (dw.shape == (ncls, 32, 3, 3) and dw.is_contiguous()))
    N, H, W, _ = y.shape
    assert w_oihw.shape == (ncls, 32, 3, 3) and ncls <= 8 and _head_dlp_ok(dlp, y, ncls)
    dx = torch.empty_like(y)
    ws = workspace(_lib.lib().dh_head_bn_bwd_workspace_size(N, H, W, groups), y.device)
    with _Prof("bn_bwd", 0, _nb(y, y, dx)):
        _call("dh_head_bn_bwd", _ci(dt(y)), P(dlp), P(w_oihw), _ci(ncls), P(y), P(scale), P(shift), P(mean), P(invstd), P(gamma),
              _ci(groups), P(dx), P(dgamma), P(dbeta), P(dw), P(db), _ci(1 if accumulate else 0), _ci(N), _ci(H), _ci(W), P(ws), S())
    return dx


def head_relu_bwd(dlp, w_oihw, ncls, relu_out, dw, db, accumulate=True):
    """class head behind a ReLU: data gradient (masked by relu_out > 0) + the head's weight / bias gradient in one pass over
    relu_out [N,H,W,32] (dh_head_relu_bwd); dlp = head_dlogits_pack(dlogits, relu_out.dtype)"""
    N, H, W, C = relu_out.shape
    assert w_oihw.shape == (ncls, 32, 3, 3) and ncls <= 8 and _head_dlp_ok(dlp, relu_out, ncls) and dw.shape == (ncls, 32, 3, 3) and \
        dw.is_contiguous()
    dx = torch.empty_like(relu_out)
    ws = workspace(_lib.lib().dh_head_bn_bwd_workspace_size(N, H, W, 1), dlp.device)
    with _Prof("act_bwd", 0, _nb(relu_out, dx)):
        _call("dh_head_relu_bwd", _ci(dt(relu_out)), P(dlp), P(w_oihw), _ci(ncls), P(relu_out), P(dx), P(dw), P(db),
              _ci(1 if accumulate else 0), _ci(N), _ci(H), _ci(W), P(ws), S())
    return dx


def nhwc_to_nchw(x):
    N, H, W, C = x.shape
    y = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
    _call("dh_nhwc_to_nchw", _ci(dt(x)), P(x), P(y), _ci(N), _ci(C), _cl(H * W), S())
    return y


def copy_channels(src, sc0, dst, dc0, cn):
    Cs, Cd = src.shape[-1], dst.shape[-1]
    Pn = src.numel() // Cs
    _call("dh_copy_channels", _ci(dt(src)), P(src), _ci(Cs), _ci(sc0), P(dst), _ci(Cd), _ci(dc0), _ci(cn), _cl(Pn), S())


# ---- small element-wise launches of independent levels, recorded and issued as one job-table launch (dh_ew_multi) ----
EW_ADD_POS, EW_CAT_HALVES, EW_SPLIT_HALVES, EW_ABSDIFF_HALVES, EW_ABSDIFF_HALVES_BWD, EW_ADD_POS_BWD = 1, 2, 3, 4, 5, 6
EW_MAXJ = 12
_EW_BATCH = None         # while an EncoderBatch(ew=True) is open: [(op, a, b, c, i0, i1, l0)] of the recorded calls (tensors kept alive)


def _ew_record(op, a, b, c, i0, i1, l0):
    """True: the call was recorded (its result is valid after the batch's next launch)"""
    if _EW_BATCH is None:
        return False
    if len(_EW_BATCH) == EW_MAXJ:
        ew_flush()
    _EW_BATCH.append((op, a, b, c, int(i0), int(i1), int(l0)))
    return True


def ew_flush():
    """issue what has been recorded (one launch); a no-op without pending jobs"""
    if not _EW_BATCH:
        return
    n = len(_EW_BATCH)
    ptr = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    ops_ = (ctypes.c_int * n)(*[j[0] for j in _EW_BATCH])
    a = (ctypes.c_void_p * n)(*[ptr(j[1]) for j in _EW_BATCH])
    b = (ctypes.c_void_p * n)(*[ptr(j[2]) for j in _EW_BATCH])
    c = (ctypes.c_void_p * n)(*[ptr(j[3]) for j in _EW_BATCH])
    i0 = (ctypes.c_int * n)(*[j[4] for j in _EW_BATCH])
    i1 = (ctypes.c_int * n)(*[j[5] for j in _EW_BATCH])
    l0 = (ctypes.c_long * n)(*[j[6] for j in _EW_BATCH])
    _call("dh_ew_multi", _ci(n), ops_, a, b, c, i0, i1, l0, S())
    del _EW_BATCH[:]


def cat_halves(t):
    """torch.cat([t[:B], t[B:]], channel) of a contiguous [2B, H, W, C] tensor, one launch"""
    n, h, w, c = t.shape
    cat = torch.empty(n // 2, h, w, 2 * c, dtype=t.dtype, device=t.device)
    assert t.is_contiguous()
    if t.dtype == torch.bfloat16 and c % 8 == 0 and _ew_record(EW_CAT_HALVES, t, None, cat, c, 0, (n // 2) * h * w):
        return cat
    _call("dh_cat_halves", _ci(dt(t)), P(t), P(cat), _ci(c), _cl((n // 2) * h * w), _ci(0), S())
    return cat


def split_halves(cat):
    """the inverse: [B, H, W, 2C] -> [2B, H, W, C] (the gradient of cat_halves)"""
    n, h, w, c2 = cat.shape
    t = torch.empty(2 * n, h, w, c2 // 2, dtype=cat.dtype, device=cat.device)
    assert cat.is_contiguous()
    if cat.dtype == torch.bfloat16 and (c2 // 2) % 8 == 0 and _ew_record(EW_SPLIT_HALVES, cat, None, t, c2 // 2, 0, n * h * w):
        return t
    _call("dh_cat_halves", _ci(dt(cat)), P(t), P(cat), _ci(c2 // 2), _cl(n * h * w), _ci(1), S())
    return t


def add(a, b):
    y = torch.empty_like(a)
    _call("dh_add", _ci(dt(a)), P(a), P(b), P(y), _cl(a.numel()), S())
    return y


def add_pos(x, pos):
    N, H, W, C = x.shape
    y = torch.empty_like(x)
    assert x.is_contiguous() and pos.is_contiguous()
    if x.dtype == torch.bfloat16 and C % 8 == 0 and _ew_record(EW_ADD_POS, x, pos, y, N, C, H * W):
        return y
    _call("dh_add_pos", _ci(dt(x)), P(x), P(pos), P(y), _ci(N), _cl(H * W), _ci(C), S())
    return y


def add_pos_bwd(dy, dpos, accumulate=False):
    N, H, W, C = dy.shape
    assert dy.is_contiguous() and dpos.is_contiguous()
    if dy.dtype == torch.bfloat16 and C == 32 and _ew_record(EW_ADD_POS_BWD, dy, None, dpos, N, int(accumulate), H * W):
        return
    _call("dh_add_pos_bwd", _ci(dt(dy)), P(dy), P(dpos), _ci(N), _cl(H * W), _ci(C), _ci(int(accumulate)), S())


def act_bwd(dy, ref, act):
    dx = torch.empty_like(dy)
    _call("dh_act_bwd", _ci(dt(dy)), P(dy), P(ref), P(dx), _cl(dy.numel()), _ci(act), S())
    return dx


def maxpool(x, want_arg=False, bn=None):
    """bn = (scale, shift, groups): x is a pre-BatchNorm tensor, relu(x * scale + shift) is applied on load"""
    N, H, W, C = x.shape
    y = torch.empty(N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C, dtype=x.dtype, device=x.device)
    arg = torch.empty(y.shape, dtype=torch.uint8, device=x.device) if want_arg else None
    sc, sh, groups = bn if bn is not None else (None, None, 0)
    _call("dh_maxpool3x3s2_fwd", _ci(dt(x)), P(x), P(y), P(arg), _ci(N), _ci(H), _ci(W), _ci(C), P(sc), P(sh), _ci(groups),
          S())
    return (y, arg) if want_arg else y


def maxpool_bwd(arg, dy, in_shape):
    N, H, W, C = in_shape
    dx = torch.empty(N, H, W, C, dtype=dy.dtype, device=dy.device)
    _call("dh_maxpool3x3s2_bwd", _ci(dt(dy)), P(arg), P(dy), P(dx), _ci(N), _ci(H), _ci(W), _ci(C), S())
    return dx


def upsample2(x):
    N, H, W, C = x.shape
    y = torch.empty(N, 2 * H, 2 * W, C, dtype=x.dtype, device=x.device)
    _call("dh_upsample2_nearest_fwd", _ci(dt(x)), P(x), P(y), _ci(N), _ci(H), _ci(W), _ci(C), S())
    return y


def upsample2_bwd(dy):
    N, H2, W2, C = dy.shape
    dx = torch.empty(N, H2 // 2, W2 // 2, C, dtype=dy.dtype, device=dy.device)
    _call("dh_upsample2_nearest_bwd", _ci(dt(dy)), P(dy), P(dx), _ci(N), _ci(H2 // 2), _ci(W2 // 2), _ci(C), S())
    return dx


def absdiff_upsample4(a, b):
    N, H, W, C = a.shape
    y = torch.empty(N, 4 * H, 4 * W, C, dtype=a.dtype, device=a.device)
    _call("dh_absdiff_upsample4_fwd", _ci(dt(a)), P(a), P(b), P(y), _ci(N), _ci(H), _ci(W), _ci(C), S())
    return y


def absdiff_upsample4_bwd(a, b, dy):
    N, H, W, C = a.shape
    da, db = torch.empty_like(a), torch.empty_like(b)
    _call("dh_absdiff_upsample4_bwd", _ci(dt(a)), P(a), P(b), P(dy), P(da), P(db), _ci(N), _ci(H), _ci(W), _ci(C), S())
    return da, db


def conv3x3_dgrad_through_up4(dy, wdgrad, a, b, da=None, db=None):
    """backward of conv3x3(bilinear_x4(|a - b|)) with respect to a and b, from dy [N, 4H, 4W, K] (gradient of the conv's
    output) and the conv's data-gradient pack [9, 32, K]: the fine-grid gradient [N, 4H, 4W, 32] is never written -- the
    convolution reduces each tile to the coarse pixels it interpolates from, a second small launch sums and applies the sign.
    Returns (da, db) like a, b [N, H, W, 32] bf16."""
    N, H, W, C = a.shape
    assert C == 32 and a.dtype == torch.bfloat16 and dy.shape[:3] == (N, 4 * H, 4 * W) and wdgrad.shape[-2] == 32
    L = _lib.lib()
    partial = torch.empty(L.dh_conv3x3_dgrad_up4_partial_floats(N, 4 * H, 4 * W), dtype=torch.float32, device=a.device)
    with _Prof("conv_mfma<bf16,ks3,s1,nt32>", 2.0 * N * 16 * H * W * 32 * dy.shape[-1] * 9, _nb(dy, wdgrad, partial)):
        _call("dh_conv3x3_dgrad_up4", _ci(dt(dy)), P(dy), P(wdgrad), _ci(N), _ci(4 * H), _ci(4 * W), _ci(dy.shape[-1]), P(partial), S())
    da = torch.empty_like(a) if da is None else da
    db = torch.empty_like(b) if db is None else db
    _call("dh_absdiff_up4_combine", P(partial), P(a), P(b), P(da), P(db), _ci(N), _ci(H), _ci(W), S())
    return da, db


def absdiff(a, b):
    y = torch.empty_like(a)
    _call("dh_absdiff", _ci(dt(a)), P(a), P(b), P(y), _cl(a.numel()), S())
    return y


def absdiff_bwd(a, b, dy, da, db, accumulate=True):
    _call("dh_absdiff_bwd", _ci(dt(a)), P(a), P(b), P(dy), P(da), P(db), _cl(a.numel()), _ci(int(accumulate)), S())


def colsum(x2d, out, accumulate=False):
    C = x2d.shape[-1]
    Pn = x2d.numel() // C
    ws = workspace(1024 * C * 4 + 1024, x2d.device)          # up to 1024 partial rows
    _call("dh_colsum", _ci(dt(x2d)), P(x2d), _cl(Pn), _ci(C), P(out), _ci(int(accumulate)), P(ws), S())


def cast_from_f32(src, dtype):
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    _call("dh_cast_from_f32", _ci(_DT[dtype]), P(src), P(dst), _cl(src.numel()), S())
    return dst


def cast_to_f32(src, dst=None, accumulate=False):
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    _call("dh_cast_to_f32", _ci(dt(src)), P(src), P(dst), _cl(src.numel()), _ci(int(accumulate)), S())
    return dst


# ---- token side ----------------------------------------------------------------------------------
def tokenizer_fwd(x, wa, pos, B, L):
    """x [S, H, W, 32]; returns (tok_cat [B, 2L, 32] fp32 written for the S/B streams present, saved).  Tokens are
    fp32 in every compute mode (csrc/tokens.hip header)."""
    Sn, H, W, C = x.shape
    HW = H * W
    dev = x.device
    logits = torch.empty(Sn * HW, L, dtype=torch.float32, device=dev)
    stats = torch.empty(Sn, L, 2, dtype=torch.float32, device=dev)
    pooled = torch.empty(Sn, L, 32, dtype=torch.float32, device=dev)
    tok = torch.empty(B, 2 * L, 32, dtype=torch.float32, device=dev)
    nws = _lib.lib().dh_tokenizer_fwd_workspace_size(Sn, HW, L)
    args = lambda ws: (_ci(dt(x)), P(x), P(wa), P(pos), _ci(Sn), _ci(B), _ci(HW), _ci(L), P(logits), P(stats), P(pooled),
                       P(tok), P(ws), S())
    if _XPREP_BATCH is not None and _ENC_BATCH is not None:
        # recorded (EncoderBatch): issued with the other levels' tokenizers, right before the recorded encoder stacks that read
        # the tokens -- scratch of its own until then.  (A caller that reads `tok` with an immediate launch instead calls
        # flush_recorded_tokens() first: Engine.encoder's layer-wise path.)
        ws = torch.empty(max(int(nws), 1), dtype=torch.uint8, device=dev)
        _XPREP_BATCH.extend((ws, x, pos))
        _call("dh_tokenizer_fwd", *args(ws))
    else:
        with _XprepPaused():
            _call("dh_tokenizer_fwd", *args(workspace(nws, dev)))
    return tok, (logits, stats, pooled)


def flush_recorded_tokens():
    """inside an EncoderBatch: issue the recorded tokenizer / preparation forward launches now (their results are read next by
    a launch that is not a recorded one)"""
    if _XPREP_BATCH is not None:
        _call("dh_xprep_batch_launch_fwd", S())


def tokenizer_bwd(x, wa, saved, dtok_cat, dx_accum, dwa, dpos, B, L, accumulate=False):
    Sn, H, W, C = x.shape
    HW = H * W
    logits, stats, pooled = saved
    Lb = _lib.lib()
    nws = Lb.dh_tokenizer_bwd_workspace_size(Sn, HW, L)
    if _XPREP_BATCH is not None:          # recorded: dx_accum / dwa / dpos are valid after the EncoderBatch's next launch()
        ws = torch.empty(max(int(nws), 1), dtype=torch.uint8, device=x.device)
        _XPREP_BATCH.extend((ws, dtok_cat, dx_accum, dpos))
    else:
        ws = workspace(nws, x.device)
    assert dtok_cat.dtype == torch.float32
    _call("dh_tokenizer_bwd", _ci(dt(x)), P(x), P(wa), _ci(Sn), _ci(B), _ci(HW), _ci(L), P(logits), P(stats), P(pooled),
          P(dtok_cat), P(dx_accum), P(dwa), P(dpos), _ci(int(accumulate)), P(ws), S())


def prep_mfma_supported(dtype, Sn, L, heads, dim_head, HLP):
    return bool(_lib.lib().dh_xattn_prep_mfma_supported(_ci(_DT[dtype]), _ci(Sn), _ci(L), _ci(heads), _ci(dim_head), _ci(HLP)))


DEC_RECORD = None    # set to [] by bench.py for ONE eager step: every fused-decoder call (layer / stack, forward / backward /
                     # finalize) as {"kind", "rows", "depth", "mlp", "call"} in program order, with {"kind": "launch"} where an
                     # EncoderBatch issued what it had recorded -- bench.py re-issues them, grouped the same way, inside a recorded
                     # graph to time the attention blocks by themselves (the `attention` record of its line)


def _dec_recorded(kind, rows_of, depth_of, mlp_of):
    """decorator of the fused-decoder entry points: see DEC_RECORD"""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            if DEC_RECORD is not None:
                DEC_RECORD.append(dict(kind=kind, rows=rows_of(a), depth=depth_of(a), mlp=mlp_of(a, k),
                                       call=functools.partial(fn, *a, **k)))
            return fn(*a, **k)
        return wrapped
    return deco


class XattnPrep:
    """Per-image operands of the re-associated cross attention (see csrc/tokens.hip)."""

    def __init__(self, tok, bstride, sstride, B, Sn, L, heads, dim_head, ln_g, ln_b, wq, wkT, wvT, woT, dtype,
                 scale=32 ** -0.5, eps=1e-5, masters=None):
        """masters = (wk, wv, wo fp32 masters, wqT in `dtype`): lets the matrix-core form run where it serves the shape"""
        dev = tok.device
        assert tok.dtype == torch.float32, "tokens are fp32 in every compute mode"
        inner = heads * dim_head
        self.HLP = cdiv(heads * L, 32) * 32
        self.args = (bstride, sstride, B, Sn, L, heads, dim_head)
        self.scale = scale
        self.mn = torch.empty(Sn, L, 32, dtype=torch.float32, device=dev)
        self.mstats = torch.empty(Sn, L, 2, dtype=torch.float32, device=dev)
        self.k = torch.empty(Sn, L, inner, dtype=torch.float32, device=dev)
        self.v = torch.empty_like(self.k)
        self.kq = torch.empty(Sn, self.HLP, 32, dtype=dtype, device=dev)
        self.kqT = torch.empty(Sn, 32, self.HLP, dtype=dtype, device=dev)
        self.vo = torch.empty(Sn, self.HLP, 32, dtype=dtype, device=dev)
        self.voT = torch.empty(Sn, 32, self.HLP, dtype=dtype, device=dev)
        self.mfma = masters is not None and prep_mfma_supported(dtype, Sn, L, heads, dim_head, self.HLP)
        self._bwd_ops = (wq, woT, wkT, wvT)
        if self.mfma:
            wk, wv, wo, wqT = masters
            with _XprepPaused():
                _call("dh_xattn_prep_fwd_stack_mfma", P(tok), _cl(bstride), _cl(sstride), _ci(B), _ci(Sn), _ci(heads),
                      _ci(dim_head), _ci(self.HLP), _cf(scale), _cf(eps), _ci(1), _cl(0), P(ln_g), P(ln_b), P(wk), P(wv), P(wo),
                      P(wqT), P(self.mn), P(self.mstats), P(self.k), P(self.v), P(self.kq), P(self.kqT), P(self.vo),
                      P(self.voT), S())
            return
        _call("dh_xattn_prep_fwd", _ci(_DT[dtype]), P(tok), _cl(bstride), _cl(sstride), _ci(B), _ci(Sn), _ci(L),
              _ci(heads), _ci(dim_head), _ci(self.HLP), _cf(scale), _cf(eps), P(ln_g), P(ln_b), P(wq), P(wkT), P(wvT),
              P(woT), P(self.mn), P(self.mstats), P(self.k), P(self.v), P(self.kq), P(self.kqT), P(self.vo),
              P(self.voT), S())


class _PrepView:
    """one layer of an XattnPrepStack, with the attributes of XattnPrep"""
    __slots__ = ("mn", "mstats", "k", "v", "kq", "kqT", "vo", "voT", "HLP", "args", "scale")


class XattnPrepStack:
    """XattnPrep for ALL layers of a decoder stack in one launch (they read the same tokens).  ln_g0 / ln_b0 / wq0 are
    the first layer's fp32 parameters, consecutive layers lie `param_stride` floats apart (the net's flat arena);
    wkT / wvT / woT are stacked [layers, 32 * inner] transposes in `dtype`."""

    def __init__(self, tok, bstride, sstride, B, Sn, L, heads, dim_head, layers, param_stride, ln_g0, ln_b0, wq0, wkT, wvT,
                 woT, dtype, scale=32 ** -0.5, eps=1e-5, masters=None, record=True):
        """masters = (wk0, wv0, wo0 fp32 masters of the first layer, stacked wqT): see XattnPrep.
        record=False: launch at once even inside an EncoderBatch (the caller's next launch is not a recorded one)"""
        dev = tok.device
        assert tok.dtype == torch.float32, "tokens are fp32 in every compute mode"
        inner = heads * dim_head
        self.layers, self.param_stride, self.dtype = layers, param_stride, dtype
        self.HLP = cdiv(heads * L, 32) * 32
        self.args = (bstride, sstride, B, Sn, L, heads, dim_head)
        self.scale = scale
        f32 = dict(dtype=torch.float32, device=dev)
        self.mn = torch.empty(layers, Sn, L, 32, **f32)
        self.mstats = torch.empty(layers, Sn, L, 2, **f32)
        self.k = torch.empty(layers, Sn, L, inner, **f32)
        self.v = torch.empty_like(self.k)
        self.kq = torch.empty(layers, Sn, self.HLP, 32, dtype=dtype, device=dev)
        self.kqT = torch.empty(layers, Sn, 32, self.HLP, dtype=dtype, device=dev)
        self.vo = torch.empty_like(self.kq)
        self.voT = torch.empty_like(self.kqT)
        # per-image weight gradients of every layer, filled by the layers' backward kernels
        self.dkq = torch.empty(layers, Sn, self.HLP, 32, **f32)
        self.dvoT = torch.empty(layers, Sn, 32, self.HLP, **f32)
        self.mfma = masters is not None and prep_mfma_supported(dtype, Sn, L, heads, dim_head, self.HLP)
        self._bwd_ops = (wq0, woT, wkT, wvT)          # what the matrix-core backward reads
        if self.mfma:
            wk0, wv0, wo0, wqT = masters
            if _XPREP_BATCH is not None and not record:
                _lib.lib().dh_xprep_batch_pause(1)
            try:
                _call("dh_xattn_prep_fwd_stack_mfma", P(tok), _cl(bstride), _cl(sstride), _ci(B), _ci(Sn), _ci(heads),
                      _ci(dim_head), _ci(self.HLP), _cf(scale), _cf(eps), _ci(layers), _cl(param_stride), P(ln_g0), P(ln_b0),
                      P(wk0), P(wv0), P(wo0), P(wqT), P(self.mn), P(self.mstats), P(self.k), P(self.v), P(self.kq), P(self.kqT),
                      P(self.vo), P(self.voT), S())
            finally:
                if _XPREP_BATCH is not None and not record:
                    _lib.lib().dh_xprep_batch_pause(0)
            if _XPREP_BATCH is not None and record:
                _XPREP_BATCH.append(tok)
            return
        _call("dh_xattn_prep_fwd_stack", _ci(_DT[dtype]), P(tok), _cl(bstride), _cl(sstride), _ci(B), _ci(Sn), _ci(L),
              _ci(heads), _ci(dim_head), _ci(self.HLP), _cf(scale), _cf(eps), _ci(layers), _cl(param_stride), P(ln_g0),
              P(ln_b0), P(wq0), P(wkT), P(wvT), P(woT), P(self.mn), P(self.mstats), P(self.k), P(self.v), P(self.kq),
              P(self.kqT), P(self.vo), P(self.voT), S())

    def layer(self, i):
        v = _PrepView()
        for n in ("mn", "mstats", "k", "v", "kq", "kqT", "vo", "voT"):
            setattr(v, n, getattr(self, n)[i])
        v.HLP, v.args, v.scale = self.HLP, self.args, self.scale
        return v

    def backward(self, tok, dtok_accum, ln_g0, wqT, wk0, wv0, wo0, dln_g0, dln_b0, dwq0, dwk0, dwv0, dwo0):
        """after every layer's backward stored its dkq / dvoT: token gradient (accumulated into dtok_accum), the
        to_q / to_k / to_v / to_out weight gradients and the shared LayerNorm's, for all layers, in one pass"""
        bstride, sstride, B, Sn, L, heads, dim_head = self.args
        dk = torch.empty_like(self.k)
        dv = torch.empty_like(self.v)
        nws = _lib.lib().dh_xattn_prep_bwd_stack_workspace_size(Sn, L, self.layers)
        if _XPREP_BATCH is not None and self.mfma:
            # recorded (EncoderBatch): the launch comes later, next to other stacks' -- scratch of its own until then
            ws = torch.empty(max(int(nws), 1), dtype=torch.uint8, device=tok.device)
            _XPREP_BATCH.extend((ws, dk, dv, tok, dtok_accum))
        else:
            ws = workspace(nws, tok.device)
            if _DEC_BATCH is not None:            # launched at once: a held-back stack finalize produces the dkq / dvoT read here
                _call("dh_decoder_batch_launch", S())
        if self.mfma:
            wq0, woT, wkT, wvT = self._bwd_ops
            _call("dh_xattn_prep_bwd_stack_mfma", P(tok), P(dtok_accum), _cl(bstride), _cl(sstride), _ci(B), _ci(Sn),
                  _ci(heads), _ci(dim_head), _ci(self.HLP), _cf(self.scale), _ci(self.layers), _cl(self.param_stride),
                  P(ln_g0), P(wq0), P(woT), P(wkT), P(wvT), P(self.mn), P(self.mstats), P(self.k), P(self.v), P(self.dkq),
                  P(self.dvoT), P(dk), P(dv), P(dln_g0), P(dln_b0), P(dwq0), P(dwk0), P(dwv0), P(dwo0), _ci(1), P(ws), S())
            return
        _call("dh_xattn_prep_bwd_stack", _ci(_DT[self.dtype]), P(tok), P(dtok_accum), _cl(bstride), _cl(sstride), _ci(B),
              _ci(Sn), _ci(L), _ci(heads), _ci(dim_head), _ci(self.HLP), _cf(self.scale), _ci(self.layers),
              _cl(self.param_stride), P(ln_g0), P(wqT), P(wk0), P(wv0), P(wo0), P(self.mn), P(self.mstats), P(self.k),
              P(self.v), P(self.dkq), P(self.dvoT), P(dk), P(dv), P(dln_g0), P(dln_b0), P(dwq0), P(dwk0), P(dwv0),
              P(dwo0), _ci(1), P(ws), S())


def xattn_prep_bwd(prep, tok, dtok_accum, ln_g, wqT, wk, wv, wo, dkq, dvoT, dln_g, dln_b, dwq, dwk, dwv, dwo,
                   accumulate, dtype):
    bstride, sstride, B, Sn, L, heads, dim_head = prep.args
    dk = torch.empty_like(prep.k)
    dv = torch.empty_like(prep.v)
    Lb = _lib.lib()
    if getattr(prep, "mfma", False):
        wq0, woT, wkT, wvT = prep._bwd_ops
        ws = workspace(Lb.dh_xattn_prep_bwd_stack_workspace_size(Sn, L, 1), tok.device)
        with _XprepPaused():
            _call("dh_xattn_prep_bwd_stack_mfma", P(tok), P(dtok_accum), _cl(bstride), _cl(sstride), _ci(B), _ci(Sn), _ci(heads),
                  _ci(dim_head), _ci(prep.HLP), _cf(prep.scale), _ci(1), _cl(0), P(ln_g), P(wq0), P(woT), P(wkT), P(wvT),
                  P(prep.mn), P(prep.mstats), P(prep.k), P(prep.v), P(dkq), P(dvoT), P(dk), P(dv), P(dln_g), P(dln_b), P(dwq),
                  P(dwk), P(dwv), P(dwo), _ci(int(accumulate)), P(ws), S())
        return
    ws = workspace(Lb.dh_xattn_prep_bwd_workspace_size(Sn), tok.device)
    _call("dh_xattn_prep_bwd", _ci(_DT[dtype]), P(tok), P(dtok_accum), _cl(bstride), _cl(sstride), _ci(B), _ci(Sn),
          _ci(L), _ci(heads), _ci(dim_head), _ci(prep.HLP), _cf(prep.scale), P(ln_g), P(wqT), P(wk), P(wv), P(wo),
          P(prep.mn), P(prep.mstats), P(prep.k), P(prep.v), P(dkq), P(dvoT), P(dk), P(dv), P(dln_g), P(dln_b), P(dwq),
          P(dwk), P(dwv), P(dwo), _ci(int(accumulate)), P(ws), S())


@_dec_recorded("fwd", lambda a: a[0].shape[0], lambda a: 1, lambda a, k: a[12])
def decoder_layer_fwd(x2d, prep, rows_per_image, ln1_g, ln1_b, bo, ln2_g, ln2_b, w1, b1, w2, b2, mlp, eps=1e-5, fp8=False):
    """fused cross-attention + MLP decoder layer (csrc/decoder_fused.hip); x2d [rows, 32] bf16.
    fp8=True: the four MFMA products on OCP e4m3 operands (csrc/decoder_fp8.hip)"""
    y = torch.empty_like(x2d)
    with _Prof("decoder_layer_fwd", 0, _nb(x2d, y)):
        _call("dh_decoder_layer_fwd_fp8" if fp8 else "dh_decoder_layer_fwd", P(x2d), P(y), P(prep.kq), P(prep.voT), P(ln1_g), P(ln1_b), P(bo), P(ln2_g), P(ln2_b),
              P(w1), P(b1), P(w2), P(b2), _cl(x2d.shape[0]), _ci(rows_per_image), _ci(mlp), _cf(eps), S())
    if _DEC_BATCH is not None and not fp8:        # recorded: operands alive until the batch is issued
        _DEC_BATCH.extend((x2d, y, prep.kq, prep.voT))
    return y


@_dec_recorded("bwd", lambda a: a[0].shape[0], lambda a: 1, lambda a, k: a[16])
def decoder_layer_bwd(x2d, dy, prep, rows_per_image, ln1_g, ln1_b, bo, ln2_g, ln2_b, w1, w1T, b1, w2, w2T, b2, grads, mlp,
                      eps=1e-5, dkq=None, dvoT=None, partial=None):
    """returns (dx, dkq [S,32,32] fp32, dvoT [S,32,32] fp32); grads = (dw1, dw2, db1, db2, dbo, dg1, dbe1, dg2, dbe2)
    are accumulated in place.  partial (a per-layer buffer of decoder_layer_bwd_partial_floats floats): only the data
    gradient runs, the parameter gradients are summed later by decoder_stack_bwd_finalize (grads is ignored)."""
    rows = x2d.shape[0]
    images = rows // rows_per_image
    dx = torch.empty_like(x2d)
    if dkq is None:
        dkq = torch.empty(images, 32, 32, dtype=torch.float32, device=x2d.device)
        dvoT = torch.empty(images, 32, 32, dtype=torch.float32, device=x2d.device)
    if partial is not None:
        ws, gp = partial, [_vp(0)] * 9
    else:
        ws = workspace(_lib.lib().dh_decoder_layer_bwd_workspace_size(_cl(rows), rows_per_image, mlp), x2d.device)
        gp = [P(t) for t in grads]
    with _Prof("decoder_layer_bwd", 0, _nb(x2d, dy, dx)):
        _call("dh_decoder_layer_bwd", P(x2d), P(dy), P(dx), P(prep.kq), P(prep.voT), P(prep.vo), P(prep.kqT), P(ln1_g),
              P(ln1_b), P(bo), P(ln2_g), P(ln2_b), P(w1), P(w1T), P(b1), P(w2), P(w2T), P(b2), *gp,
              P(dkq), P(dvoT), _cl(rows), _ci(rows_per_image), _ci(mlp), _cf(eps), P(ws), S())
    if _DEC_BATCH is not None and partial is not None:
        _DEC_BATCH.extend((x2d, dy, dx, ws, prep.kq, prep.voT, prep.vo, prep.kqT))
    return dx, dkq, dvoT


@_dec_recorded("fwd", lambda a: a[0].shape[0], lambda a: a[1].layers, lambda a, k: a[7])
def decoder_stack_fwd(x2d, stack, rows_per_image, params0, w1s, w2s, par_stride, mlp, eps=1e-5):
    """ALL layers of a fused decoder stack in one launch (csrc/decoder_fused.hip, DecArgs::depth): a workgroup takes its
    pixel rows through every layer.  stack: the XattnPrepStack of the layers; params0 = (ln1_g, ln1_b, bo, ln2_g, ln2_b, b1,
    b2) of layer 0, the others `par_stride` floats further; w1s / w2s: the packed MLP weights stacked [depth, ...].
    Returns ys [depth, rows, 32]: every layer's output (ys[-1] = the stack's)."""
    depth, rows = stack.layers, x2d.shape[0]
    ys = torch.empty(depth, rows, 32, dtype=x2d.dtype, device=x2d.device)
    g1, b1_, bo, g2, b2_, fb1, fb2 = params0
    with _Prof("decoder_layer_fwd", 0, depth * _nb(x2d, x2d)):
        _call("dh_decoder_stack_fwd", P(x2d), P(ys), P(stack.kq), P(stack.voT), P(g1), P(b1_), P(bo), P(g2), P(b2_), P(w1s), P(fb1),
              P(w2s), P(fb2), _ci(depth), _cl(stack.kq[0].numel()), _cl(w1s[0].numel()), _cl(par_stride), _cl(rows),
              _ci(rows_per_image), _ci(mlp), _cf(eps), S())
    if _DEC_BATCH is not None:
        _DEC_BATCH.extend((x2d, ys, stack.kq, stack.voT, w1s, w2s))
    return ys


@_dec_recorded("bwd", lambda a: a[0].shape[0], lambda a: a[3].layers, lambda a, k: a[11])
def decoder_stack_bwd(x2d, ys, dy, stack, rows_per_image, params0, w1s, w1Ts, w2s, w2Ts, par_stride, mlp, partials, eps=1e-5):
    """data gradient of decoder_stack_fwd in one launch; the per-workgroup parameter-gradient partials of layer l land in
    partials[l] (decoder_stack_bwd_finalize sums them).  Returns dx."""
    depth, rows = stack.layers, x2d.shape[0]
    dx, dwork = torch.empty_like(x2d), torch.empty_like(x2d)
    g1, b1_, bo, g2, b2_, fb1, fb2 = params0
    assert partials.shape[0] == depth and partials.is_contiguous()
    with _Prof("decoder_layer_bwd", 0, depth * _nb(x2d, dy, dx)):
        _call("dh_decoder_stack_bwd", P(x2d), P(ys), P(dy), P(dx), P(dwork), P(stack.kq), P(stack.voT), P(stack.vo), P(stack.kqT),
              P(g1), P(b1_), P(bo), P(g2), P(b2_), P(w1s), P(w1Ts), P(fb1), P(w2s), P(w2Ts), P(fb2), _ci(depth),
              _cl(stack.kq[0].numel()), _cl(w1s[0].numel()), _cl(par_stride), _cl(rows), _ci(rows_per_image), _ci(mlp), _cf(eps),
              P(partials), S())
    if _DEC_BATCH is not None:
        _DEC_BATCH.extend((x2d, ys, dy, dx, dwork, partials, stack.kq, stack.voT, stack.vo, stack.kqT, w1s, w1Ts, w2s, w2Ts))
    return dx


def decoder_layer_bwd_partial_floats(rows, rows_per_image, mlp):
    return _lib.lib().dh_decoder_layer_bwd_workspace_size(_cl(rows), rows_per_image, mlp) // 4


@_dec_recorded("fin", lambda a: a[1], lambda a: a[0].shape[0], lambda a, k: a[3])
def decoder_stack_bwd_finalize(partials, rows, rows_per_image, mlp, grads0, grad_stride, dkq, dvoT):
    """one launch for the parameter gradients of all layers of a decoder stack: partials [depth, floats], grads0 = the nine
    gradient tensors of layer 0 (layer l's sit grad_stride floats further), dkq / dvoT [depth, images, 32, 32]"""
    _call("dh_decoder_stack_bwd_finalize", P(partials), _ci(partials.shape[0]), _cl(rows), _ci(rows_per_image), _ci(mlp),
          *(P(t) for t in grads0), _cl(grad_stride), P(dkq), P(dvoT), S())


def encoder_supported(n, heads, dim_head, mlp):
    return bool(_lib.lib().dh_encoder_supported(n, heads, dim_head, mlp))


def encoder_fwd(tok2d, B, n, depth, heads, dim_head, mlp, pstride, params, save, scale=32 ** -0.5, eps=1e-5):
    """fused token encoder stack (csrc/encoder_fused.hip).  tok2d [B*n, 32] fp32; params: the 11 first-layer tensors
    (ln1_g, ln1_b, wqkv, wo, bo, ln2_g, ln2_b, w1, b1, w2, b2).  Returns (y, saved per-layer forward images | None)."""
    assert tok2d.dtype == torch.float32 and tok2d.shape == (B * n, 32)
    y = torch.empty_like(tok2d)
    xs = torch.empty(_lib.lib().dh_encoder_saved_floats(B, n, depth, heads, dim_head, mlp), dtype=torch.float32,
                     device=tok2d.device) if save else None          # [depth][B][forward image]
    _call("dh_encoder_fwd", P(tok2d), P(y), P(xs), _ci(B), _ci(n), _ci(depth), _ci(heads), _ci(dim_head), _ci(mlp),
          _cf(scale), _cf(eps), _cl(pstride), *(P(t) for t in params), S())
    if _ENC_BATCH is not None:
        _ENC_BATCH.extend((tok2d, y, xs))
    return y, xs


def encoder_bwd(dy, xs, B, n, depth, heads, dim_head, mlp, pstride, params, grads, scale=32 ** -0.5, eps=1e-5):
    """dx of the fused encoder; `grads` (same order as params, first layer, in the gradient arena) are accumulated"""
    dx = torch.empty_like(dy)
    nbytes = _lib.lib().dh_encoder_bwd_workspace_size(B, n, depth, heads, dim_head, mlp)
    if _ENC_BATCH is not None:         # recorded, issued later with other stacks: a workspace of its own, alive until then
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dy.device)
        _ENC_BATCH.extend((dy, dx, xs, ws))
    else:
        ws = workspace(nbytes, dy.device)
    _call("dh_encoder_bwd", P(dy), P(dx), P(xs), _ci(B), _ci(n), _ci(depth), _ci(heads), _ci(dim_head), _ci(mlp),
          _cf(scale), _cf(eps), _cl(pstride), *(P(t) for t in params), *(P(t) for t in grads), P(ws), S())
    return dx


_ENC_BATCH = None        # while an EncoderBatch is open: the tensors of the recorded launches
_DEC_BATCH = None        # ... and of its recorded decoder layers
_XPREP_BATCH = None      # ... and of its recorded cross-attention preparations (their own workspaces, dk / dv)


def xprep_recording():
    """True while XattnPrepStack launches are only recorded: their results are valid after the EncoderBatch's next launch()"""
    return _XPREP_BATCH is not None


class _XprepPaused:
    """the single-layer preparation calls launch at once even inside an open batch (their callers read the result next)"""

    def __enter__(self):
        if _XPREP_BATCH is not None:
            _lib.lib().dh_xprep_batch_pause(1)

    def __exit__(self, *exc):
        if _XPREP_BATCH is not None:
            _lib.lib().dh_xprep_batch_pause(0)
        return False


class EncoderBatch:
    """`with EncoderBatch() as eb:` -- encoder_fwd / encoder_bwd calls inside only RECORD their launches (dh_encoder_batch_*);
    eb.launch() issues them together (one workgroup per image each: stacks of independent levels share the chip).  Their
    outputs are valid after launch().  Inert under ops.PROFILE (per-launch events) and with DAHITRA_ENC_BATCH=0."""

    def __init__(self, decoder=False, ew=False):
        """ew=True: add_pos / add_pos_bwd / cat_halves / split_halves / absdiff_halves(_bwd) calls are recorded too (one
        dh_ew_multi launch per round, issued FIRST): the caller pauses between such a call and the first use of its result.
        DAHITRA_EW_BATCH=0 switches that part off.
        decoder=True: the fused decoder layers (decoder_layer_fwd, and decoder_layer_bwd with `partial`) are recorded too
        (dh_decoder_batch_*): layers of independent stacks share a launch.  DAHITRA_DEC_BATCH=0 switches that part off."""
        self.on = PROFILE is None and os.environ.get("DAHITRA_ENC_BATCH", "1") != "0"
        self.dec = decoder and PROFILE is None and os.environ.get("DAHITRA_DEC_BATCH", "1") != "0"
        # with the decoder stacks, their token-side preparation (XattnPrepStack and its backward, dh_xprep_batch_*) and the
        # stack's parameter-gradient finalize are recorded as well.  DAHITRA_XPREP_BATCH=0 switches that part off.
        self.xprep = self.dec and os.environ.get("DAHITRA_XPREP_BATCH", "1") != "0"
        self.ew = ew and PROFILE is None and os.environ.get("DAHITRA_EW_BATCH", "1") != "0"

    def __enter__(self):
        global _ENC_BATCH, _DEC_BATCH, _XPREP_BATCH, _EW_BATCH
        if self.ew:
            assert _EW_BATCH is None, "EncoderBatch(ew=True) is not re-entrant"
            _EW_BATCH = []
        if self.on or self.dec:
            assert _ENC_BATCH is None and _DEC_BATCH is None and _XPREP_BATCH is None, "EncoderBatch is not re-entrant"
        if self.on:
            _ENC_BATCH = []
            _call("dh_encoder_batch_begin")
        if self.dec:
            _DEC_BATCH = []
            _call("dh_decoder_batch_begin")
        if self.xprep:
            _XPREP_BATCH = []
            _call("dh_xprep_batch_begin")
        return self

    def launch(self):
        """program order of what one round may have recorded: a stack's preparation before its forward; a stack's finalize
        (which the decoder batch issues after its backward launches) before the preparation's backward"""
        if self.ew:
            ew_flush()
        if self.xprep:
            _call("dh_xprep_batch_launch_fwd", S())
        if self.on:
            _call("dh_encoder_batch_launch", S())
            del _ENC_BATCH[:]
        if self.dec:
            _call("dh_decoder_batch_launch", S())
            del _DEC_BATCH[:]
            if DEC_RECORD is not None:
                DEC_RECORD.append(dict(kind="launch"))
        if self.xprep:
            _call("dh_xprep_batch_launch_bwd", S())
            del _XPREP_BATCH[:]

    def __exit__(self, *exc):
        global _ENC_BATCH, _DEC_BATCH, _XPREP_BATCH, _EW_BATCH
        failed = bool(exc) and exc[0] is not None
        if self.ew:
            if not failed:
                ew_flush()
            _EW_BATCH = None
        if self.xprep:
            if failed:
                _lib.lib().dh_xprep_batch_abort()
                _XPREP_BATCH = None
            else:
                _call("dh_xprep_batch_launch_fwd", S())
        if self.on:
            if failed:
                _lib.lib().dh_encoder_batch_abort()
            else:
                _call("dh_encoder_batch_end", S())
            _ENC_BATCH = None
        if self.dec:
            if failed:
                _lib.lib().dh_decoder_batch_abort()
            else:
                _call("dh_decoder_batch_end", S())
            _DEC_BATCH = None
        if self.xprep and not failed:
            _call("dh_xprep_batch_end", S())      # after the decoder batch: a pending finalize precedes the preparation's backward
            _XPREP_BATCH = None
        return False


def softmax_groups(x2d, heads, L):
    rows, HLP = x2d.shape
    y = torch.empty_like(x2d)
    _call("dh_softmax_groups_fwd", _ci(dt(x2d)), P(x2d), P(y), _cl(rows), _ci(heads), _ci(L), _ci(HLP), S())
    return y


def softmax_groups_bwd(y, dy, heads, L):
    rows, HLP = y.shape
    dx = torch.empty_like(y)
    _call("dh_softmax_groups_bwd", _ci(dt(y)), P(y), P(dy), P(dx), _cl(rows), _ci(heads), _ci(L), _ci(HLP), S())
    return dx


def self_attn(qkv, B, n, heads, dim_head, scale=32 ** -0.5):
    inner = heads * dim_head
    o = torch.empty(B * n, inner, dtype=qkv.dtype, device=qkv.device)
    attn = torch.empty(B, heads, n, n, dtype=torch.float32, device=qkv.device)
    _call("dh_self_attn_fwd", _ci(dt(qkv)), P(qkv), P(o), P(attn), _ci(B), _ci(n), _ci(heads), _ci(dim_head),
          _cf(scale), S())
    return o, attn


def self_attn_bwd(qkv, attn, dout, B, n, heads, dim_head, scale=32 ** -0.5):
    dqkv = torch.empty_like(qkv)
    _call("dh_self_attn_bwd", _ci(dt(qkv)), P(qkv), P(attn), P(dout), P(dqkv), _ci(B), _ci(n), _ci(heads),
          _ci(dim_head), _cf(scale), S())
    return dqkv


# ---- loss / mask / optimizer ---------------------------------------------------------------------
def focal_loss(logits_nchw, target, want_grad=True, grad_scale=1.0, alpha=0.5):
    B, C, H, W = logits_nchw.shape
    loss = torch.empty((), dtype=torch.float32, device=logits_nchw.device)
    dl = torch.empty_like(logits_nchw) if want_grad else None
    ws = workspace(4096, logits_nchw.device)
    _call("dh_focal_loss", P(logits_nchw), P(target), _ci(B), _ci(C), _cl(H * W), _cf(alpha), _cf(grad_scale), P(loss),
          P(dl), P(ws), S())
    return loss, dl


def cross_entropy_fwd(logits_nchw, target, ignore_index=255):
    """returns out [2] (device): mean CE over the non-ignored pixels, their count"""
    B, C, H, W = logits_nchw.shape
    out = torch.empty(2, dtype=torch.float32, device=logits_nchw.device)
    ws = workspace(16384, logits_nchw.device)
    _call("dh_cross_entropy_fwd", P(logits_nchw), P(target), _ci(B), _ci(C), _cl(H * W), _ci(ignore_index), P(out), P(ws), S())
    return out


def cross_entropy_bwd(logits_nchw, target, fwd_out, upstream, ignore_index=255):
    B, C, H, W = logits_nchw.shape
    dl = torch.empty_like(logits_nchw)
    _call("dh_cross_entropy_bwd", P(logits_nchw), P(target), _ci(B), _ci(C), _cl(H * W), _ci(ignore_index), P(fwd_out),
          P(upstream), P(dl), S())
    return dl


def dice_argmax_constant(logits_nchw, target, eps=1e-7):
    B, C, H, W = logits_nchw.shape
    loss = torch.empty((), dtype=torch.float32, device=logits_nchw.device)
    ws = workspace(24576, logits_nchw.device)
    _call("dh_dice_argmax_constant", P(logits_nchw), P(target), _ci(B), _ci(C), _cl(H * W), _cf(eps), P(loss), P(ws), S())
    return loss


def confusion_matrix(logits_nchw, target, counts, want_mask=False):
    """counts [C, C] int64 (device) += confusion of argmax(logits) against target; optionally returns the mask"""
    B, C, H, W = logits_nchw.shape
    mask = torch.empty(B, H, W, dtype=torch.int64, device=logits_nchw.device) if want_mask else None
    _call("dh_confusion_matrix", P(logits_nchw), P(target), _ci(B), _ci(C), _cl(H * W), P(mask), P(counts), S())
    return mask


def argmax_nchw(logits_nchw):
    B, C, H, W = logits_nchw.shape
    mask = torch.empty(B, H, W, dtype=torch.int64, device=logits_nchw.device)
    _call("dh_argmax_nchw", P(logits_nchw), P(mask), _ci(B), _ci(C), _cl(H * W), S())
    return mask


def adamw_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _call("dh_adamw_step", P(param), P(grad), P(exp_avg), P(exp_avg_sq), _cl(param.numel()), _cf(lr), _cf(beta1),
          _cf(beta2), _cf(eps), _cf(weight_decay), _ci(step), _cf(grad_scale), S())


# ---- xBD train step (csrc/xbd_step.hip) ---------------------------------------------------------------
def combo_loss_fwd(logits, masks, weights_dev, dice_weight=1.0, focal_weight=8.0):
    """per-channel ComboLoss{dice, focal} (xBD_code/losses.py:95-126).  Returns (loss [], channel_loss [C], sums [C,4])"""
    B, C, H, W = logits.shape
    dev = logits.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    ch = torch.empty(C, dtype=torch.float32, device=dev)
    sums = torch.empty(C, 4, dtype=torch.float32, device=dev)
    ws = workspace(_lib.lib().dh_combo_loss_workspace_size(C), dev)
    _call("dh_combo_loss_fwd", P(logits), P(masks), _ci(B), _ci(C), _cl(H * W), P(weights_dev), _cf(dice_weight),
          _cf(focal_weight), P(sums), P(ch), P(loss), P(ws), S())
    return loss, ch, sums


def combo_loss_bwd(logits, masks, sums, weights_dev, upstream, dice_weight=1.0, focal_weight=8.0):
    B, C, H, W = logits.shape
    dl = torch.empty_like(logits)
    _call("dh_combo_loss_bwd", P(logits), P(masks), P(sums), P(weights_dev), P(upstream), _cf(dice_weight),
          _cf(focal_weight), _ci(B), _ci(C), _cl(H * W), P(dl), S())
    return dl


def grad_norm_clip_coef(grad_flat, max_norm, out):
    """out [2] fp32 (device): total L2 norm of the flat gradient arena, clip_grad_norm_ coefficient"""
    ws = workspace(_lib.lib().dh_grad_norm_workspace_size(), grad_flat.device)
    _call("dh_grad_norm_clip_coef", P(grad_flat), _cl(grad_flat.numel()), _cf(max_norm), P(out), P(ws), S())


def adamw_xbd_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale_dev=None):
    _call("dh_adamw_xbd_step", P(param), P(grad), P(exp_avg), P(exp_avg_sq), _cl(param.numel()), _cf(lr), _cf(beta1),
          _cf(beta2), _cf(eps), _cf(weight_decay), _ci(step), P(grad_scale_dev), S())


# ---- variants writing into caller-provided (contiguous) buffers ------------------------------------
def stem_space_to_depth_into(x_nchw, out):
    N, C, H, W = x_nchw.shape
    assert C == 3 and out.is_contiguous()
    _call("dh_stem_space_to_depth", _ci(dt(out)), P(x_nchw), P(out), _ci(N), _ci(H), _ci(W), _ci(out.shape[-1]), S())


def absdiff_upsample4_bwd_into(a, b, dy, da, db):
    N, H, W, C = a.shape
    _call("dh_absdiff_upsample4_bwd", _ci(dt(a)), P(a), P(b), P(dy), P(da), P(db), _ci(N), _ci(H), _ci(W), _ci(C), S())


def reduce_rows(partial, nt, n, out, accumulate=False, scale=1.0):
    _call("dh_reduce_partials", P(partial), _cl(nt), _cl(n), _cf(scale), P(out), _ci(int(accumulate)), S())


def scale_into(src, scalar_dev, dst):
    """dst = src * scalar (a 0-d / 1-element device tensor), no host sync"""
    _call("dh_scale_by_scalar", P(src), P(scalar_dev.reshape(1).float().contiguous()), P(dst), _cl(src.numel()), S())


def absdiff_halves(tok3, out):
    """out[b] = |tok3[b, 1] - tok3[b, 0]| for tok3 [B, 2, n] (token difference, networks.py:1311)"""
    B = tok3.shape[0]
    n = out.numel() // B
    assert tok3.is_contiguous() and out.is_contiguous()
    if tok3.dtype == torch.float32 and _ew_record(EW_ABSDIFF_HALVES, tok3, None, out, B, 0, n):
        return
    _call("dh_absdiff_halves", _ci(dt(tok3)), P(tok3), P(out), _ci(B), _cl(n), S())


def absdiff_halves_bwd(tok3, dout, dtok3):
    B = tok3.shape[0]
    n = dout.numel() // B
    assert tok3.is_contiguous() and dout.is_contiguous() and dtok3.is_contiguous()
    if tok3.dtype == torch.float32 and _ew_record(EW_ABSDIFF_HALVES_BWD, tok3, dout, dtok3, B, 0, n):
        return
    _call("dh_absdiff_halves_bwd", _ci(dt(tok3)), P(tok3), P(dout), P(dtok3), _ci(B), _cl(n), S())
