"""Forward / backward pipelines of the change-detection nets, composed from libdahitra_hip kernels.

No autograd graph is built for the interior: each block's forward returns its output plus a closure
that runs the hand-written backward kernels and deposits parameter gradients straight into the flat
fp32 gradient arena.  The two temporal streams are processed as ONE batch of 2B images ("groups" = 2
keeps BatchNorm statistics per stream exactly as the reference's two forward_single calls do,
models/networks.py:360-361), which halves the launch count and doubles the work per launch.

Reference structure followed (not copied): BASE_Transformer.forward models/networks.py:358-392,
ResNet.forward_single :233-257, BasicBlock models/resnet.py:58-73, Transformer / TransformerDecoder
models/help_funcs.py:154-186, BASE_Transformer_UNet.forward models/networks.py:1297-1357.
"""
import os
import sys
import types

import torch

from . import ops
from .netspec import (ATTN_SCALE, BIT_DIM_HEAD, BIT_HEADS, BN_EPS, BN_MOMENTUM, DIM, LN_EPS, UNET_LEVELS,
                      get_config, is_active, state_spec)

RELU, GELU, NONE = ops.ACT_RELU, ops.ACT_GELU, ops.ACT_NONE



def _drain(gen):
    """runs a staged generator (Engine._decoder_gen and friends) straight through and returns its value"""
    try:
        while True:
            next(gen)
    except StopIteration as e:
        return e.value


class Packed:
    """kernel-layout copies of one weight: forward / data-gradient form, and (bf16 3x3 layers the register-resident-weights
    convolution may serve) the same two in fragment order"""
    __slots__ = ("fwd", "dgrad", "fwd_frag", "dgrad_frag")

    def __init__(self, fwd, dgrad, fwd_frag=None, dgrad_frag=None):
        self.fwd, self.dgrad, self.fwd_frag, self.dgrad_frag = fwd, dgrad, fwd_frag, dgrad_frag


class Gated:
    """Output of a data-gradient launch that was gated for the BatchNorm layer it feeds (ops.conv2d(..., gate=...)):
    g = dout * ReLU-mask and the per-tile partial sums (sum g, sum g * xhat) of that layer's backward."""
    __slots__ = ("g", "partial")

    def __init__(self, g, partial):
        self.g, self.partial = g, partial


class HeadGrad:
    """The class head's data gradient NOT formed: the dlogits as zero-bordered bf16 pairs (ops.head_dlogits_pack) and the head's weights, for a BatchNorm + ReLU
    backward that recomputes it in both of its passes (ops.head_bn_bwd) instead of reading a 32-channel tensor twice."""
    __slots__ = ("dl", "w", "ncls", "gw", "gb")

    def __init__(self, dl, w, ncls, gw, gb):
        self.dl, self.w, self.ncls, self.gw, self.gb = dl, w, ncls, gw, gb        # gw / gb: the head's own gradients, (+)= there


class CoarseRes:
    """a data gradient that lives on the COARSE grid of a stride-2 layer ([N, OH, OW, C]): the value of the fine-grid
    gradient at the even-even positions, zero elsewhere (the 1x1 stride-2 shortcut of a BasicBlock)"""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class Engine:
    def __init__(self, net_G, dtype=torch.float32, use_tr=True, attn_fp8=False, mma_x3=False):
        self.net_G = net_G
        # fp32 nets only (compute_dtype="bf16x3"): matrix products on the 16-bit matrix cores as THREE split products per operand
        # pair (ops.set_f32_mma_mode, csrc/common.h f32x3 / f32h3 / f32x6).  The gradients are ~20x more sensitive to activation
        # error than the logits are: with the forward on bf16 planes (unit roundoff 2^-17) they sit 150x above their fp32 distance
        # from the oracle (median 1.9e-3 vs 1.3e-5 on base_transformer_pos_s4), with only the BACKWARD at 2^-17 they stay at it
        # (2.2e-5), measured on MI355X.  So the FORWARD splits into FP16 planes (form 3: 11 mantissa bits per plane, ~2^-21 with
        # three products; activations are O(1) and the weight tile is staged times 2^8, which keeps both inside fp16's range) and
        # the data / weight gradients, whose operands span fp32's range, into BF16 planes (form 1).  Form 2 (three bf16 planes,
        # six products, 2^-23) is the forward's alternative for nets whose activations or weights leave fp16's range (|x| < 65504,
        # |w| < 255): DAHITRA_X3_FWD=2.  DAHITRA_X3_FWD / DAHITRA_X3_BWD = 0 .. 3 override the form of a pass (A/B switches).
        self.mma_x3 = bool(mma_x3) and dtype == torch.float32
        self.mma_fwd = int(os.environ.get("DAHITRA_X3_FWD", "3")) if self.mma_x3 else 0
        self.mma_bwd = int(os.environ.get("DAHITRA_X3_BWD", "1")) if self.mma_x3 else 0
        self._x3_checked = False
        # fp8 (OCP e4m3) MFMA operands in the fused decoder layers' forward products (BASELINE configs[4]); bf16 mode only
        self.attn_fp8 = bool(attn_fp8) or os.environ.get("DAHITRA_ATTN_FP8", "0") == "1"
        self.cfg = get_config(net_G)
        self.dtype = dtype
        self.use_tr = use_tr
        self.fused_decoder = os.environ.get("DAHITRA_NO_FUSED_DECODER", "0") != "1"
        self.fused_encoder = os.environ.get("DAHITRA_NO_FUSED_ENCODER", "0") != "1"
        # BatchNorm backward with the reduction pass folded into the epilogue of the data-gradient conv that produces
        # dout (DAHITRA_BN_FUSION=1).  Measured time-neutral on MI355X: it removes one of the three tensor reads of
        # the reduction, but the gated MFMA launches slow down by as much (4054.6 vs 4054.1 pairs/s) -- so the
        # two-pass backward stays the default and the fused path is kept, tested, as an option.
        self.fused_bn_bwd = os.environ.get("DAHITRA_BN_FUSION", "0") == "1"
        # the 1x1 stride-2 shortcut of a Bottleneck: its data gradient as a GEMM on the COARSE grid + an in-place add at the
        # even-even positions, instead of zero insertion + a 1x1 convolution on the fine grid (DAHITRA_NO_COARSE_SHORTCUT=1)
        self.coarse_shortcut = os.environ.get("DAHITRA_NO_COARSE_SHORTCUT", "0") != "1"
        # BatchNorm-apply + ReLU fused into the load of the consumer convolution (forward and weight gradient): the
        # normalised activation of conv1 of every BasicBlock / of the head's first conv is never materialised
        self.lazy_bn = os.environ.get("DAHITRA_NO_LAZY_BN", "0") != "1"
        # conv_pred(nearest-upsample-x2(x)) as four 2x2 phase convolutions on x (exact up to fp re-association: the 3x3 taps
        # that read the same source pixel are pre-summed): 2.25x fewer FLOPs, no upsampled tensor (models/networks.py:251-256)
        self.phase_conv_pred = os.environ.get("DAHITRA_NO_PHASE_CONV", "0") != "1"
        # data gradient of the 3x3 stride-2 convolutions as four output-parity phase convolutions over dY instead of a
        # stride-1 convolution over the zero-inserted dY (4x the pixels, 75 % zeros)
        self.phase_s2_dgrad = os.environ.get("DAHITRA_NO_PHASE_S2", "0") != "1"
        # the UNet up path's relu(conv3x3(nearest-upsample-x2(x))) (32 -> 32 channels) in the same phase form (_up_conv)
        self.phase_up_conv = os.environ.get("DAHITRA_NO_PHASE_UPCONV", "0") != "1"
        # conv_layer2_0 reads cat([a_128, b_128], 1) in place (ops.SplitCat) instead of through two channel copies each way
        self.split_cat = os.environ.get("DAHITRA_NO_SPLIT_CAT", "0") != "1"
        # bf16: the 7x7/2 stem as one kernel on the NCHW fp32 images (csrc/stem.hip) instead of space-to-depth + 4x4 conv
        self.direct_stem = os.environ.get("DAHITRA_NO_DIRECT_STEM", "0") != "1"
        # ... and its backward without a BatchNorm pass: reduction fused into the max-pool backward, apply into the weight gradient
        self.fused_stem_bwd = os.environ.get("DAHITRA_NO_FUSED_STEM_BWD", "0") != "1"
        # the class head writes its fp32 NCHW logits itself (no NHWC logits tensor, no layout pass)
        self.fused_head_out = os.environ.get("DAHITRA_NO_FUSED_HEAD", "0") != "1"
        # one finalize launch for the parameter gradients of all layers of a fused decoder stack
        self.defer_dec_finalize = os.environ.get("DAHITRA_NO_DEFER_DEC_FINALIZE", "0") != "1"
        # all layers of a fused decoder stack in one launch per direction (csrc/decoder_fused.hip DecArgs::depth): the stack is
        # a per-pixel-row function, so nothing synchronises between its layers but a workgroup's own re-staging of the weights
        self.fuse_dec_stack = os.environ.get("DAHITRA_NO_DEC_STACK", "0") != "1"
        # the three levels' small kernels on streams of their own between the shared launches (_run_staged)
        self.level_streams = os.environ.get("DAHITRA_LEVEL_STREAMS", "0") == "1"
        self._lstreams = []
        # class-head data gradient gated for the classifier's BatchNorm (mask + BN-backward sums in its epilogue).  Measured
        # neutral (8020 vs 8027 pairs/s: the reduction pass it removes costs what the extra read of y costs the head kernel):
        # off by default, DAHITRA_GATED_HEAD=1 turns it on.
        self.gated_head_dgrad = os.environ.get("DAHITRA_GATED_HEAD", "0") == "1"
        # the class head's data gradient never written: both passes of the classifier's BatchNorm backward recompute it from the
        # 16-byte-per-pixel dlogits (one MFMA per 16 pixels x 16 channels) -- 3 x 134 MB less at the bench size
        # (DAHITRA_NO_FUSED_HEAD_BN=1: the three-kernel path)
        self.fused_head_bn = os.environ.get("DAHITRA_NO_FUSED_HEAD_BN", "0") != "1"
        # a second stream for the classifier's weight gradient, next to the low-occupancy token-side backward (bf16 training).
        # Measured: the launch does overlap the encoder's backward in the graph (rocprofv3 trace: 120 us next to encoder_bwd /
        # encoder_wgrad / tok_bwd instead of 63 us alone), but those kernels slow down by 28 us and the forked graph costs more
        # than the rest: 7950 vs 8010 pairs/s over five interleaved 150-step runs.  Off by default (DAHITRA_SIDE_STREAM=1).
        # bf16: the data gradient of classifier.0 goes straight to the coarse |A - B| maps (the 32 x 256 x 256 gradient of the
        # bilinear-upsampled map is never written or re-read): DAHITRA_NO_FUSED_UP4_BWD=1 restores the two-kernel path
        self.fused_up4_bwd = os.environ.get("DAHITRA_NO_FUSED_UP4_BWD", "0") != "1"
        # ... and in the forward |A - B| + bilinear x4 can be formed inside classifier.0's loads (ops.Up4Input: bit-identical).
        # Measured on MI355X (profiles/r04_up4_ab.txt): ~70 VALU instructions per 16-byte halo piece in a kernel that runs one
        # tile per workgroup -- 121 us against absdiff_up4_fwd 32 + the register-resident-weights kernel 76; the same on-load
        # form of the weight gradient ran 94 us against 67 (and is gone again: its results were not reproducible from run to
        # run at >= 512 workgroups).  OFF by default; DAHITRA_FUSED_UP4_FWD=1 turns the convolution's form on (the weight
        # gradient then materialises the map itself).  DESIGN.md section 6e.
        self.fused_up4_fwd = os.environ.get("DAHITRA_FUSED_UP4_FWD", "0") == "1"
        self.use_side = os.environ.get("DAHITRA_SIDE_STREAM", "0") == "1"
        self.pack_side = os.environ.get("DAHITRA_PACK_SIDE", "0") == "1"
        self._pack_stream = None
        self.side = None
        self._deferred_wgrad = None
        self.shapes = {k: s for k, s, _ in state_spec(net_G)}
        self.p = {}        # key -> fp32 parameter / buffer tensors (device)
        self.g = {}        # key -> fp32 gradient views
        self.pk = {}       # key -> Packed
        self.wstack = {}   # (decoder prefix, 'net.0' | 'net.3') -> stacked packed MLP weights (forward, data gradient)
        self.training = False
        self.need_grad = False
        self._bwd = None
        self._plans = {}
        self._wgrad_plan = None

    # ---- binding ---------------------------------------------------------------------------------
    def bind(self, params, grads):
        self.p, self.g = params, grads
        self._plans = {}       # the packed buffers point into the old arena

    def _pack_all(self):
        """kernel-layout copies of every weight the forward (and, with need_grad, the data gradients) will read.
        The destinations and the job table are built once per (arena, need_grad) and then refreshed by ONE launch
        per step (ops.PackPlan); the 7x7 stem has its own space-to-depth pack."""
        plan = self._plans.get(self.need_grad)
        if plan is None:
            plan = self._plans[self.need_grad] = self._build_plan()
        pack, self.pk, self.xstack, self.wstack = plan
        # the per-step re-pack (one launch, 37 - 42 us of strided gathers) next to the stem, which reads the OIHW weights itself:
        # a side stream forked here and joined after the stem (_pack_join) -- in the recorded step two parallel branches
        if self.pack_side and self.training and self.direct_stem and self.dtype == torch.bfloat16 and ops.PROFILE is None:
            if self._pack_stream is None:
                self._pack_stream = ops.SideStream(self.p["resnet.conv1.weight"].device)
            with self._pack_stream.fork():
                pack.run()
        else:
            pack.run()
        key = "resnet.conv1.weight"
        if not (self.direct_stem and self.dtype == torch.bfloat16):      # the direct stem kernel reads the OIHW weights itself
            self.pk[key] = Packed(ops.stem_pack_weight(self.p[key], self.dtype), None)

    def _pack_join(self):
        """the current stream waits for the re-pack issued on the side stream (no-op otherwise): before the first packed weight is read"""
        if self._pack_stream is not None:
            self._pack_stream.join()

    def _build_plan(self):
        ck = ops.chunk_channels(self.dtype)
        pack = ops.PackPlan(self.p["resnet.conv1.weight"].device)
        pk, xstack = {}, {}    # xstack: (decoder prefix, matrix) -> stacked transposes [depth, 32*inner]
        wstack = {}            # (decoder prefix, "net.0" | "net.3") -> (stacked forward packs, stacked data-gradient packs) [depth, O*I]
        for key, shape in self.shapes.items():
            if key not in self.p or not key.endswith("weight") or len(shape) not in (2, 4):
                continue
            if not is_active(self.net_G, key) or key == "resnet.conv1.weight":
                continue
            if key.startswith("conv_a") or key.startswith("conv_token") or key.startswith("resnet.fc") or \
                    key.startswith("resnet.layer4"):
                continue        # consumed in fp32 by the tokenizer kernels, or unused by the forward
            if key == "conv_pred.weight" and self.phase_conv_pred and shape[1] % 64 == 0 and self.cfg["kind"] == "bit":
                continue        # packed in its 2x2 phase form by conv_pred_phase
            if self.phase_up_conv and key in ("conv_layer2.0.weight", "conv_layer3.0.weight", "conv_layer4.0.weight") and \
                    tuple(shape[:2]) == (32, 32):
                continue        # packed in its 2x2 phase form by _up_conv
            if ".to_q." in key or ".to_k." in key or ".to_v." in key or \
                    (".to_out." in key and "transformer_decoder" in key):
                # cross-attention weights: the token-side prep reads their transposes (coalesced).  The transposes of
                # one decoder stack live in ONE [depth, 32*inner] buffer per matrix, so that a single launch can
                # prepare every layer (ops.XattnPrepStack); the per-layer views serve the layer-at-a-time path.
                pfx, rest = key.split(".layers.")
                li, which = int(rest.split(".")[0]), rest.split(".fn.fn.")[1]
                skey = (pfx, which)
                if skey not in xstack:
                    depth = 1 + max(int(k.split(".layers.")[1].split(".")[0]) for k in self.shapes
                                    if k.startswith(pfx + ".layers.") and k.endswith(which))
                    xstack[skey] = torch.empty(depth, shape[0] * shape[1], dtype=self.dtype, device=self.p[key].device)
                view = xstack[skey][li]
                pack.add(self.p[key], self.dtype, want_fwd=False, out_dgrad=view)
                pk[key] = Packed(None, view)
                continue
            O = shape[0]
            dt = self.dtype
            if key.startswith("transformer_decoder") and ".1.fn.fn.net." in key and len(shape) == 2 and shape[0] % 16 == 0 and \
                    shape[1] % 16 == 0:
                # the MLP weights of one decoder stack in ONE buffer per form, layer after layer: the layer-fused stack kernels
                # (ops.decoder_stack_fwd / _bwd) step through them at a constant stride; the per-layer views serve everything else
                pfx, rest = key.split(".layers.")
                li, which = int(rest.split(".")[0]), ("net.0" if ".net.0." in key else "net.3")
                if (pfx, which) not in wstack:
                    depth = 1 + max(int(k.split(".layers.")[1].split(".")[0]) for k in self.shapes
                                    if k.startswith(pfx + ".layers.") and k.endswith("1.fn.fn.%s.weight" % which))
                    mk = lambda: torch.empty(depth, shape[0] * shape[1], dtype=dt, device=self.p[key].device)
                    wstack[(pfx, which)] = (mk(), mk() if self.need_grad else None)
                sf, sd = wstack[(pfx, which)]
                f, d = pack.add(self.p[key], dt, want_dgrad=self.need_grad, dgrad_inner=O, out_fwd=sf[li].view(1, shape[0], shape[1]),
                                out_dgrad=sd[li].view(1, shape[1], shape[0]) if sd is not None else None)
                pk[key] = Packed(f, d)
                continue
            if key.startswith("transformer") and not key.startswith("transformer_decoder"):
                # the token encoder (<= 16 rows per image) runs in fp32 in every mode: its output feeds
                # |token2 - token1|, which bf16 rounding would wipe out (csrc/tokens.hip header)
                dt = torch.float32
            c = ops.chunk_channels(dt)
            f, d = pack.add(self.p[key], dt, want_dgrad=self.need_grad, dgrad_inner=-(-O // c) * c)
            ff = dd = None
            if dt == torch.bfloat16 and len(shape) == 4 and shape[2] == 3 and \
                    ((O in (64, 128, 256) and shape[1] in (64, 128, 256)) or (O == 32 and shape[1] == 32)):
                # the layers conv_wreg.hip may serve (it decides per launch): a second copy in fragment order
                ff, dd = pack.add(self.p[key], dt, want_dgrad=self.need_grad, dgrad_inner=O, frag=True)
            pk[key] = Packed(f, d, ff, dd)
        return pack, pk, xstack, wstack

    # ---- primitive units -------------------------------------------------------------------------
    def s2_phase_ok(self, ks, stride, pad, xshape, dy):
        N, H, W, Cin = xshape
        return self.phase_s2_dgrad and ks == 3 and stride == 2 and pad == 1 and Cin in (32, 64) and \
            H == 2 * dy.shape[1] and W == 2 * dy.shape[2] and dy.shape[-1] % ops.chunk_channels(self.dtype) == 0

    def conv_dgrad(self, dy, wkey, ks, stride, pad, xshape, residual=None, dilation=1, gate=None, coarse=False):
        """data gradient of a convolution; with `gate` (the .gate of the BN layer that produced this conv's input) the
        result is a Gated pair instead of the plain gradient.  coarse=True (1x1 stride 2 only): the gradient on the coarse
        grid, i.e. without the zero positions.  residual may be a CoarseRes (3x3 stride-2 phase path only)."""
        N, H, W, Cin = xshape
        flops = 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * self.shapes[wkey][0] * Cin * ks * ks   # algorithmic
        if coarse:
            assert ks == 1 and stride == 2 and residual is None and gate is None
            return ops.conv2d(dy, self.pk[wkey].dgrad, Cin, 1, 1, 0, alg_flops=flops)
        if gate is None and self.s2_phase_ok(ks, stride, pad, xshape, dy):
            wph = ops.pack_s2_dgrad_phase_weights(self.p[wkey], self.dtype)
            if residual is None or isinstance(residual, CoarseRes):
                return ops.conv3x3s2_dgrad(dy, wph, Cin, residual.t if residual is not None else None, alg_flops=flops)
            return ops.add(ops.conv3x3s2_dgrad(dy, wph, Cin, None, alg_flops=flops), residual)
        assert not isinstance(residual, CoarseRes)
        if stride == 2:
            dy = ops.zero_insert2(dy, H, W)
        r = ops.conv2d(dy, self.pk[wkey].dgrad, Cin, ks, 1, dilation * (ks - 1) - pad, residual=residual,
                       out_hw=(H, W), alg_flops=flops, dilation=dilation, gate=gate,
                       w_frag=self.pk[wkey].dgrad_frag if stride == 1 else None)
        return Gated(*r) if gate is not None else r

    def conv_bn(self, x, wkey, bnkey, ks, stride, pad, groups, relu, residual=None, dilation=1, lazy=False, side_wgrad=False):
        """conv + BatchNorm (+ residual) (+ ReLU).  x may be an ops.BnInput (the previous layer's lazy output).
        lazy=True (train mode, ReLU, no residual; the ONLY consumer must be a 3x3 stride-1 convolution and its weight
        gradient): the normalised activation is never written -- the call returns an ops.BnInput (pre-BN conv output +
        scale / shift) and the consumer applies relu(y * scale + shift) while it loads (the bn_apply pass, one read
        and one write of the activation, disappears)."""
        cout = self.shapes[wkey][0]
        lazy = lazy and self.training and self.lazy_bn and relu and residual is None and not self.fused_bn_bwd
        gamma, beta = self.p[bnkey + ".weight"], self.p[bnkey + ".bias"]
        rm, rv = self.p[bnkey + ".running_mean"], self.p[bnkey + ".running_var"]
        act = RELU if relu else NONE
        split = isinstance(x, ops.SplitCat)          # cat([A, B], channel) read in place (3x3 / stride 1, train mode, bf16)
        if split and not (self.training and ks == 3 and stride == 1 and pad == 1 and dilation == 1 and residual is None):
            x, split = x.materialize(), False
        relu_bits = None
        if self.training:
            if split:
                y, st = ops.conv3x3_split(x, self.pk[wkey].fwd, self.pk[wkey].fwd_frag, cout, want_stats=True)
            else:
                y, st = ops.conv2d(x, self.pk[wkey].fwd, cout, ks, stride, pad, want_stats=True, dilation=dilation,
                                   w_frag=self.pk[wkey].fwd_frag if stride == 1 else None)
            N, OH, OW, _ = y.shape
            mean, invstd, scale, shift = ops.bn_finalize(st, cout, groups, (N // groups) * OH * OW, gamma, beta, rm, rv,
                                                         BN_MOMENTUM, BN_EPS, nbt=self.p[bnkey + ".num_batches_tracked"])
            if lazy:
                out = ops.BnInput(y, scale, shift, groups)
            elif relu and residual is not None and self.need_grad:
                # BatchNorm + residual + ReLU: the backward needs the ReLU mask only -- written here as one byte per 8 elements
                # (bf16) and read there in place of `out` (one tensor read less in each of its passes)
                out, relu_bits = ops.bn_apply(y, scale, shift, groups, act, residual, want_bits=True)
            else:
                out = ops.bn_apply(y, scale, shift, groups, act, residual)
        else:
            # eval: BatchNorm folds into the convolution -- scale into the packed weights, shift as the bias
            scale, shift = ops.bn_eval_params(gamma, beta, rm, rv, BN_EPS)
            wp = ops.pack_weight(self.p[wkey], self.dtype, want_dgrad=False, out_scale=scale)[0]
            out = ops.conv2d(x, wp, cout, ks, stride, pad, bias=shift, residual=residual, act=act, dilation=dilation)
            y = mean = invstd = None
        if not self.need_grad:
            return out, None
        has_res = residual is not None

        def bwd(dout, need_dx=True, dx_res=None, next_gate=None, coarse_dx=False, through_up4=None):
            """through_up4 = (a, b, da, db): this conv's input is bilinear_x4(|a - b|); the gradients land in da, db"""
            if isinstance(dout, HeadGrad):    # the class head's gradient, recomputed by both passes of this backward
                dy = ops.head_bn_bwd(dout.dl, dout.w, dout.ncls, y, scale, shift, mean, invstd, gamma,
                                     self.g[bnkey + ".weight"], self.g[bnkey + ".bias"], groups, dw=dout.gw, db=dout.gb)
                dres = None
            elif isinstance(dout, Gated):     # ReLU mask and the reduction pass were done by the producer of dout
                dy = ops.bn_bwd_from_partials(dout.g, y, dout.partial, mean, invstd, gamma, self.g[bnkey + ".weight"],
                                              self.g[bnkey + ".bias"], groups, accumulate=True)
                dres = dout.g if has_res else None
            elif relu and not has_res:
                # no residual: the ReLU mask is recomputed from y (x * scale + shift > 0) -- `out` is not read
                dy, dres = ops.bn_bwd(dout, None, y, mean, invstd, gamma, self.g[bnkey + ".weight"],
                                      self.g[bnkey + ".bias"], groups, accumulate=True, mask_scale=scale,
                                      mask_shift=shift), None
            else:
                r = ops.bn_bwd(dout, out if relu else None, y, mean, invstd, gamma, self.g[bnkey + ".weight"],
                               self.g[bnkey + ".bias"], groups, accumulate=True, want_dres=has_res, bits=relu_bits)
                dy, dres = r if has_res else (r, None)
            if side_wgrad and self.side is not None:
                # Deferred to the side stream: nothing until the batched split-K reduce needs this weight gradient.  It is
                # launched (run_deferred_wgrad) when the main stream reaches the token encoder's backward -- one workgroup per
                # image, 3/4 of the CUs idle for ~150 us; forked right here it only competed with the data gradient below.
                self._deferred_wgrad = (x, dy, wkey, ks, stride, pad, dilation)
            else:
                ops.conv2d_wgrad(x, dy, self.g[wkey], ks, stride, pad, accumulate=True, use_tr=self.use_tr,
                                 dilation=dilation)
            if through_up4 is not None:
                return ops.conv3x3_dgrad_through_up4(dy, self.pk[wkey].dgrad, *through_up4), dres
            if split and need_dx:
                # the gradient of the concatenation, as the two batch halves of ONE [2B, H, W, C / 2] tensor (ops.SplitCat)
                assert dx_res is None and next_gate is None and not coarse_dx
                cin = x.shape[-1]
                return ops.conv3x3_split(dy, self.pk[wkey].dgrad, self.pk[wkey].dgrad_frag, cin, split_out=True,
                                         alg_flops=2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * cout * cin * 9), dres
            dx = self.conv_dgrad(dy, wkey, ks, stride, pad, x.shape, residual=dx_res, dilation=dilation,
                                 gate=next_gate, coarse=coarse_dx) if need_dx else None
            return dx, dres
        # what an upstream data-gradient launch needs to gate for this layer (None: keep the two-pass backward)
        bwd.gate = (out if relu else None, y, mean, invstd, groups) if (self.fused_bn_bwd and cout % 16 == 0) else None
        # what the class head's data gradient needs to gate itself for this layer (ReLU mask recomputed from y)
        bwd.bn = (y, scale, shift, mean, invstd, groups) if (self.training and relu and not has_res) else None
        return out, bwd

    def conv_act(self, x, wkey, bkey, ks, pad, act, cpad_grad=False):
        cout = self.shapes[wkey][0]
        out = ops.conv2d(x, self.pk[wkey].fwd, cout, ks, 1, pad, bias=self.p[bkey] if bkey else None, act=act)
        if not self.need_grad:
            return out, None

        def bwd(dout, need_dx=True, dx_res=None, masked=False):
            dy = ops.act_bwd(dout, out, RELU) if (act == RELU and not masked) else dout
            ops.conv2d_wgrad(x, dy, self.g[wkey], ks, 1, pad, accumulate=True, use_tr=self.use_tr)
            if bkey:
                ops.colsum(dy.view(-1, cout), self.g[bkey], accumulate=True)
            dx = self.conv_dgrad(dy, wkey, ks, 1, pad, x.shape, residual=dx_res) if need_dx else None
            return dx
        return out, bwd

    def conv_pred_phase(self, x):
        """upsamplex2 + conv_pred (models/networks.py:251-256) without the upsampled tensor: ops.conv_up2_*"""
        wkey, bkey = "conv_pred.weight", "conv_pred.bias"
        wf, wd, b4 = ops.pack_phase_weights(self.p[wkey], self.p[bkey], self.dtype)
        out = ops.conv_up2_fwd(x, wf, b4)
        if not self.need_grad:
            return out, None

        def bwd(dout):
            ops.conv_up2_wgrad(x, dout, self.g[wkey], accumulate=True, use_tr=self.use_tr)
            ops.colsum(dout.view(-1, 32), self.g[bkey], accumulate=True)
            return ops.conv_up2_dgrad(dout, wd, x.shape[-1])
        return out, bwd

    def stem(self, x1, x2, groups, pool=False):
        """7x7/2 stem + BN + ReLU as a 4x4 conv on the space-to-depth image (csrc/loss_optim.hip).
        pool=True (the BiT nets, where the 3x3/2 max-pool is the stem's only consumer): returns
        (pooled, argmax, stem output shape, bwd); in train mode BN + ReLU are applied while the pool loads, so the
        stem's activation is never written."""
        B = x1.shape[0]
        H, W = x1.shape[2], x1.shape[3]
        wkey, bnkey = "resnet.conv1.weight", "resnet.bn1"
        gamma, beta = self.p[bnkey + ".weight"], self.p[bnkey + ".bias"]
        rm, rv = self.p[bnkey + ".running_mean"], self.p[bnkey + ".running_var"]
        oh, ow = H // 2, W // 2
        direct = self.direct_stem and self.dtype == torch.bfloat16      # one kernel on the NCHW images (csrc/stem.hip)
        if not direct:
            cp = ops.chunk_channels(self.dtype)
            xs = torch.empty(2 * B, H // 2, W // 2, cp, dtype=self.dtype, device=x1.device)
            ops.stem_space_to_depth_into(x1, xs[:B])
            ops.stem_space_to_depth_into(x2, xs[B:])
        if self.training:
            if direct:
                y, st, xs = ops.stem7_fwd(x1, x2, self.p[wkey], want_stats=True, want_xs=self.need_grad, groups=groups)
            else:
                y, st = ops.conv2d(xs, self.pk[wkey].fwd, 64, 4, 1, 2, want_stats=True, out_hw=(oh, ow))
            mean, invstd, scale, shift = ops.bn_finalize(st, 64, groups, B * oh * ow, gamma, beta, rm, rv, BN_MOMENTUM,
                                                         BN_EPS, nbt=self.p[bnkey + ".num_batches_tracked"])
            if pool:
                out = None
                pooled, parg = ops.maxpool(y, want_arg=True, bn=(scale, shift, groups))
            else:
                out = ops.bn_apply(y, scale, shift, groups, RELU)
        else:
            scale, shift = ops.bn_eval_params(gamma, beta, rm, rv, BN_EPS)
            if direct:
                out = ops.stem7_fwd(x1, x2, self.p[wkey], out_scale=scale, bias=shift, relu=True)[0]
            else:
                wp = ops.stem_pack_weight(self.p[wkey], self.dtype, out_scale=scale)
                out = ops.conv2d(xs, wp, 64, 4, 1, 2, bias=shift, act=RELU, out_hw=(oh, ow))
            y = mean = invstd = None
            if pool:
                pooled, parg = ops.maxpool(out), None
        oshape = (2 * B, oh, ow, 64)

        def bwd(dout):
            dy = ops.bn_bwd(dout, None, y, mean, invstd, gamma, self.g[bnkey + ".weight"], self.g[bnkey + ".bias"],
                            groups, accumulate=True, mask_scale=scale, mask_shift=shift)
            ops.stem_wgrad(xs, dy, self.g[wkey], accumulate=True, use_tr=self.use_tr)

        def from_pool(parg_, dpool, extra=None):
            """backward entered at the max-pool's output: pool backward + ReLU mask + BN sums in one pass, BN backward itself
            inside the weight gradient's loads (the stem has no data gradient).  extra: a second gradient of the stem's output
            (the hierarchical model's level-2 tap), summed in the same pass"""
            d, coef = ops.stem_pool_bn_bwd(parg_, dpool, y, scale, shift, mean, invstd, gamma, self.g[bnkey + ".weight"],
                                           self.g[bnkey + ".bias"], groups, extra=extra)
            ops.stem_wgrad(xs, d, self.g[wkey], accumulate=True, use_tr=self.use_tr, bn=(y, coef, groups))
        if self.training and direct and self.fused_stem_bwd and self.need_grad:
            bwd.from_pool = from_pool        # (pool=False: the caller pooled the materialised output itself and kept the arg-max)
        if pool:
            return pooled, parg, oshape, (bwd if self.need_grad else None)
        if not self.need_grad:
            return out, None
        return out, bwd

    def basic_block(self, x, pfx, stride, groups):
        h, b1 = self.conv_bn(x, pfx + ".conv1.weight", pfx + ".bn1", 3, stride, 1, groups, True, lazy=True)
        has_ds = (pfx + ".downsample.0.weight") in self.shapes
        if has_ds:
            idt, bds = self.conv_bn(x, pfx + ".downsample.0.weight", pfx + ".downsample.1", 1, stride, 0, groups, False)
        else:
            idt, bds = x, None
        out, b2 = self.conv_bn(h, pfx + ".conv2.weight", pfx + ".bn2", 3, 1, 1, groups, True, residual=idt)
        if not self.need_grad:
            return out, None

        def bwd(dout, next_gate=None):
            dh, dres = b2(dout, next_gate=b1.gate)
            if has_ds:
                if stride == 2 and next_gate is None and self.phase_s2_dgrad and x.shape[-1] in (32, 64):
                    # shortcut gradient on the coarse grid; conv1's phase data gradient adds it at the even-even positions
                    dxds, _ = bds(dres, coarse_dx=True)
                    dx, _ = b1(dh, dx_res=CoarseRes(dxds), next_gate=next_gate)
                else:
                    dxds, _ = bds(dres)
                    dx, _ = b1(dh, dx_res=dxds, next_gate=next_gate)
            else:
                dx, _ = b1(dh, dx_res=dres, next_gate=next_gate)
            return dx
        bwd.gate = b2.gate
        return out, bwd

    def bottleneck(self, x, pfx, stride, dilation, groups):
        """Bottleneck (models/resnet.py:76-122): 1x1 -> 3x3 (stride / dilation, pad = dilation) -> 1x1 (+identity)."""
        h1, b1 = self.conv_bn(x, pfx + ".conv1.weight", pfx + ".bn1", 1, 1, 0, groups, True,
                              lazy=(stride == 1 and dilation == 1))           # consumed by the 3x3 conv2 only
        h2, b2 = self.conv_bn(h1, pfx + ".conv2.weight", pfx + ".bn2", 3, stride, dilation, groups, True,
                              dilation=dilation)
        has_ds = (pfx + ".downsample.0.weight") in self.shapes
        if has_ds:
            idt, bds = self.conv_bn(x, pfx + ".downsample.0.weight", pfx + ".downsample.1", 1, stride, 0, groups, False)
        else:
            idt, bds = x, None
        out, b3 = self.conv_bn(h2, pfx + ".conv3.weight", pfx + ".bn3", 1, 1, 0, groups, True, residual=idt)
        if not self.need_grad:
            return out, None

        def bwd(dout, next_gate=None):
            dh2, dres = b3(dout, next_gate=b2.gate)
            dh1, _ = b2(dh2, next_gate=b1.gate)
            if has_ds and stride == 2 and next_gate is None and self.coarse_shortcut:
                # the shortcut's gradient on its own coarse grid, added at the even-even positions of conv1's data gradient
                dxds, _ = bds(dres, coarse_dx=True)
                dx, _ = b1(dh1, next_gate=next_gate)
                ops.add_coarse_(dx, dxds)
            elif has_ds:
                dxds, _ = bds(dres)
                dx, _ = b1(dh1, dx_res=dxds, next_gate=next_gate)
            else:
                dx, _ = b1(dh1, dx_res=dres, next_gate=next_gate)
            return dx
        bwd.gate = b3.gate
        return out, bwd

    def res50_layer(self, x, li, stride, first_dilation, dilation, groups):
        """_make_layer (resnet.py:178-199): block 0 keeps the previous dilation, the others use the new one."""
        from .netspec import RESNET50_BLOCKS
        bwds = []
        x, b = self.bottleneck(x, "resnet.layer%d.0" % li, stride, first_dilation, groups)
        bwds.append(b)
        for i in range(1, RESNET50_BLOCKS[li - 1]):
            x, b = self.bottleneck(x, "resnet.layer%d.%d" % (li, i), 1, dilation, groups)
            bwds.append(b)
        if not self.need_grad:
            return x, None

        def bwd(d, next_gate=None):
            for i in range(len(bwds) - 1, -1, -1):
                d = bwds[i](d, next_gate=bwds[i - 1].gate if i > 0 else next_gate)
            return d
        bwd.gate = bwds[-1].gate
        return x, bwd

    def res_layer(self, x, li, stride, groups):
        x, ba = self.basic_block(x, "resnet.layer%d.0" % li, stride, groups)
        x, bb = self.basic_block(x, "resnet.layer%d.1" % li, 1, groups)
        if not self.need_grad:
            return x, None

        def bwd(d, next_gate=None):
            return ba(bb(d, next_gate=ba.gate), next_gate=next_gate)
        bwd.gate = bb.gate
        return x, bwd

    # ---- transformer pieces ----------------------------------------------------------------------
    def mlp_block(self, x1, f):
        """x + W2 gelu(W1 LN(x) + b1) + b2 on rows [R, 32] (help_funcs.py:52-63)."""
        g2, b2 = self.p[f + ".norm.weight"], self.p[f + ".norm.bias"]
        w1k, w2k = f + ".fn.net.0.weight", f + ".fn.net.3.weight"
        mlp = self.shapes[w1k][0]
        ln2, st2 = ops.layernorm(x1, g2, b2, LN_EPS)
        h, z = ops.linear(ln2, self.pk[w1k].fwd, mlp, bias=self.p[f + ".fn.net.0.bias"], act=GELU, want_preact=True)
        x2 = ops.linear(h, self.pk[w2k].fwd, DIM, bias=self.p[f + ".fn.net.3.bias"], residual=x1)
        if not self.need_grad:
            return x2, None

        def bwd(dx2):
            dh = ops.linear(dx2, self.pk[w2k].dgrad, mlp)
            ops.linear_wgrad(h, dx2, self.g[w2k], accumulate=True, use_tr=self.use_tr)
            ops.colsum(dx2, self.g[f + ".fn.net.3.bias"], accumulate=True)
            dz = ops.act_bwd(dh, z, GELU)
            dln2 = ops.linear(dz, self.pk[w1k].dgrad, DIM)
            ops.linear_wgrad(ln2, dz, self.g[w1k], accumulate=True, use_tr=self.use_tr)
            ops.colsum(dz, self.g[f + ".fn.net.0.bias"], accumulate=True)
            return ops.layernorm_bwd(dln2, x1, st2, g2, self.g[f + ".norm.weight"], self.g[f + ".norm.bias"],
                                     dx_add=dx2, accumulate=True)
        return x2, bwd

    def encoder(self, tok, pfx, depth, heads, dim_head, B, n):
        """token self-attention stack on [B*n, 32] rows (models/networks.py:457-512)."""
        fused = self._encoder_fused(tok, pfx, depth, heads, dim_head, B, n) if self.fused_encoder else None
        if fused is not None:
            return fused
        ops.flush_recorded_tokens()          # (staged levels) the layer-wise kernels below launch at once
        x = tok
        bw = []
        for i in range(depth):
            a, f = "%s.layers.%d.0.fn" % (pfx, i), "%s.layers.%d.1.fn" % (pfx, i)
            x, b1 = self._enc_attn(x, a, heads, dim_head, B, n)
            x, b2 = self.mlp_block(x, f)
            bw += [b1, b2]
        if not self.need_grad:
            return x, None

        def bwd(d):
            for b in reversed(bw):
                d = b(d)
            return d
        return x, bwd

    _ENC_NAMES = ("0.fn.norm.weight", "0.fn.norm.bias", "0.fn.fn.to_qkv.weight", "0.fn.fn.to_out.0.weight",
                  "0.fn.fn.to_out.0.bias", "1.fn.norm.weight", "1.fn.norm.bias", "1.fn.fn.net.0.weight",
                  "1.fn.fn.net.0.bias", "1.fn.fn.net.3.weight", "1.fn.fn.net.3.bias")

    def _encoder_fused(self, tok, pfx, depth, heads, dim_head, B, n):
        """the whole encoder stack as csrc/encoder_fused.hip (3 launches instead of ~22 per layer); None when the
        shape is outside that kernel or the layers' parameters are not at one constant pitch in the arenas"""
        mlp = self.shapes["%s.layers.0.1.fn.fn.net.0.weight" % pfx][0]
        if tok.dtype != torch.float32 or not ops.encoder_supported(n, heads, dim_head, mlp):
            return None
        key = lambda i, nm: "%s.layers.%d.%s" % (pfx, i, nm)
        stride = 0
        tables = (self.p, self.g) if self.need_grad else (self.p,)
        for table in tables:
            for nm in self._ENC_NAMES:
                if any(key(i, nm) not in table for i in range(depth)):
                    return None
                for i in range(1, depth):
                    d = table[key(i, nm)].data_ptr() - table[key(i - 1, nm)].data_ptr()
                    if d % 4 or (stride and d // 4 != stride):
                        return None
                    stride = d // 4
        params = [self.p[key(0, nm)] for nm in self._ENC_NAMES]
        y, xs = ops.encoder_fwd(tok, B, n, depth, heads, dim_head, mlp, stride, params, self.need_grad, ATTN_SCALE, LN_EPS)
        if not self.need_grad:
            return y, None
        grads = [self.g[key(0, nm)] for nm in self._ENC_NAMES]

        def bwd(d):
            return ops.encoder_bwd(d.contiguous(), xs, B, n, depth, heads, dim_head, mlp, stride, params, grads,
                                   ATTN_SCALE, LN_EPS)
        return y, bwd

    def _enc_attn(self, x0, a, heads, dim_head, B, n):
        g1, b1 = self.p[a + ".norm.weight"], self.p[a + ".norm.bias"]
        wqkv, wout = a + ".fn.to_qkv.weight", a + ".fn.to_out.0.weight"
        inner = heads * dim_head
        xn, st1 = ops.layernorm(x0, g1, b1, LN_EPS)
        qkv = ops.linear(xn, self.pk[wqkv].fwd, 3 * inner)
        o, attn = ops.self_attn(qkv, B, n, heads, dim_head, ATTN_SCALE)
        x1 = ops.linear(o, self.pk[wout].fwd, DIM, bias=self.p[a + ".fn.to_out.0.bias"], residual=x0)
        if not self.need_grad:
            return x1, None

        def bwd(dx1):
            do = ops.linear(dx1, self.pk[wout].dgrad, inner)
            ops.linear_wgrad(o, dx1, self.g[wout], accumulate=True, use_tr=self.use_tr)
            ops.colsum(dx1, self.g[a + ".fn.to_out.0.bias"], accumulate=True)
            dqkv = ops.self_attn_bwd(qkv, attn, do, B, n, heads, dim_head, ATTN_SCALE)
            dxn = ops.linear(dqkv, self.pk[wqkv].dgrad, DIM)
            ops.linear_wgrad(xn, dqkv, self.g[wqkv], accumulate=True, use_tr=self.use_tr)
            return ops.layernorm_bwd(dxn, x0, st1, g1, self.g[a + ".norm.weight"], self.g[a + ".norm.bias"],
                                     dx_add=dx1, accumulate=True)
        return x1, bwd

    def decoder(self, *args):
        """cross-attention stack: pixel rows x2d [images*HW, 32] attend to the L tokens of their image
        (help_funcs.py:170-186); dtok accumulates the token gradients.  (_decoder_gen run straight through.)"""
        x, bg = _drain(self._decoder_gen(*args))
        return x, (None if bg is None else (lambda d: _drain(bg(d))))

    def _decoder_gen(self, x2d, images, tok, tok_b, tok_s, B, dtok, pfx, depth, heads, dim_head, L):
        """decoder() as a GENERATOR that pauses after every fused layer launch, forward and (the generator its second return
        value makes) backward: inside an ops.EncoderBatch(decoder=True) those launches are only recorded, and _run_staged
        issues the layers the independent levels have reached as one launch"""
        bw = []
        x = x2d
        rpi = x2d.shape[0] // images
        mlp0 = self.shapes["%s.layers.0.1.fn.fn.net.0.weight" % pfx][0]
        fused = self.fused_decoder and self.dtype == torch.bfloat16 and L == 4 and heads * L <= 32 and rpi % 128 == 0 \
            and mlp0 in (32, 64)
        # launches that are NOT recorded read their inputs at once: what the caller left in the element-wise batch goes out first
        recorded = fused and ops._DEC_BATCH is not None and not self.attn_fp8
        if not recorded:
            ops.ew_flush()
        stack = self._prep_stack(tok, tok_b, tok_s, B, images, L, heads, dim_head, pfx, depth) if fused and depth > 1 \
            else None
        # deferred parameter gradients: the fused backward launches of a stack leave their per-workgroup partials in a per-layer
        # buffer and ONE launch sums them for all layers (needs the layers' parameters at one constant pitch in the arena)
        defer = None
        if fused and stack is not None and self.need_grad and self.defer_dec_finalize:
            gstride = self._layer_grad_stride(pfx, depth)
            mlps = {self.shapes["%s.layers.%d.1.fn.fn.net.0.weight" % (pfx, i)][0] for i in range(depth)}
            if gstride is not None and len(mlps) == 1:
                rows = x.shape[0]
                defer = types.SimpleNamespace(
                    partials=torch.empty(depth, ops.decoder_layer_bwd_partial_floats(rows, rpi, mlp0), dtype=torch.float32,
                                         device=x.device), gstride=gstride, rows=rows)
        # the whole stack in ONE launch per direction (ops.decoder_stack_fwd: a workgroup takes its rows through every layer)
        sfuse = None
        if fused and stack is not None and self.fuse_dec_stack and not self.attn_fp8 and (defer is not None or not self.need_grad):
            sfuse = self._stack_operands(pfx, depth)
        if sfuse is not None:
            params0, pstride, w1s, w1Ts, w2s, w2Ts = sfuse
            x0 = x
            ys = ops.decoder_stack_fwd(x0, stack, rpi, params0, w1s, w2s, pstride, mlp0, LN_EPS)
            yield                   # ys is valid from here on
            if not self.need_grad:
                return ys[depth - 1], None

            def bwd_stack(d):
                if not recorded:
                    ops.ew_flush()
                dx = ops.decoder_stack_bwd(x0, ys, d.contiguous(), stack, rpi, params0, w1s, w1Ts, w2s, w2Ts, pstride, mlp0,
                                           defer.partials, LN_EPS)
                yield               # dx is valid from here on
                a0, f0 = "%s.layers.0.0.fn" % pfx, "%s.layers.0.1.fn" % pfx
                grads0 = (self.g[f0 + ".fn.net.0.weight"], self.g[f0 + ".fn.net.3.weight"], self.g[f0 + ".fn.net.0.bias"],
                          self.g[f0 + ".fn.net.3.bias"], self.g[a0 + ".fn.to_out.0.bias"], self.g[a0 + ".norm.weight"],
                          self.g[a0 + ".norm.bias"], self.g[f0 + ".norm.weight"], self.g[f0 + ".norm.bias"])
                ops.decoder_stack_bwd_finalize(defer.partials, defer.rows, rpi, mlp0, grads0, defer.gstride, stack.dkq, stack.dvoT)
                stack.backward(tok, dtok, self.p[a0 + ".norm.weight"], self.xstack[(pfx, "to_q.weight")],
                               *(self.p[a0 + ".fn.to_%s.weight" % n] for n in ("k", "v", "out.0")),
                               self.g[a0 + ".norm.weight"], self.g[a0 + ".norm.bias"],
                               *(self.g[a0 + ".fn.to_%s.weight" % n] for n in ("q", "k", "v", "out.0")))
                yield               # (the finalize and the token-side backward may only be recorded) dtok is valid from here on
                return dx
            return ys[depth - 1], bwd_stack
        for i in range(depth):
            a, f = "%s.layers.%d.0.fn" % (pfx, i), "%s.layers.%d.1.fn" % (pfx, i)
            mlp = self.shapes[f + ".fn.net.0.weight"][0]
            if fused:
                x, b1 = self._dec_layer_fused(x, images, rpi, tok, tok_b, tok_s, B, dtok, a, f, heads, dim_head, L, mlp,
                                              stack=stack, li=i, partial=defer.partials[i] if defer is not None else None)
                bw.append(b1)
                yield               # x is valid from here on
                continue
            x, b1 = self._dec_attn(x, images, tok, tok_b, tok_s, B, dtok, a, heads, dim_head, L)
            x, b2 = self.mlp_block(x, f)
            bw += [b1, b2]
        if not self.need_grad:
            return x, None

        def bwd(d):
            if not recorded:
                ops.ew_flush()
            for b in reversed(bw):
                d = b(d)
                if fused:
                    yield           # d is valid from here on
            if defer is not None:
                a0, f0 = "%s.layers.0.0.fn" % pfx, "%s.layers.0.1.fn" % pfx
                grads0 = (self.g[f0 + ".fn.net.0.weight"], self.g[f0 + ".fn.net.3.weight"], self.g[f0 + ".fn.net.0.bias"],
                          self.g[f0 + ".fn.net.3.bias"], self.g[a0 + ".fn.to_out.0.bias"], self.g[a0 + ".norm.weight"],
                          self.g[a0 + ".norm.bias"], self.g[f0 + ".norm.weight"], self.g[f0 + ".norm.bias"])
                ops.decoder_stack_bwd_finalize(defer.partials, defer.rows, rpi, mlp0, grads0, defer.gstride, stack.dkq, stack.dvoT)
            if stack is not None:            # token-side backward of ALL layers in one pass (they stored dkq / dvoT)
                a0 = "%s.layers.0.0.fn" % pfx
                stack.backward(tok, dtok, self.p[a0 + ".norm.weight"], self.xstack[(pfx, "to_q.weight")],
                               *(self.p[a0 + ".fn.to_%s.weight" % n] for n in ("k", "v", "out.0")),
                               self.g[a0 + ".norm.weight"], self.g[a0 + ".norm.bias"],
                               *(self.g[a0 + ".fn.to_%s.weight" % n] for n in ("q", "k", "v", "out.0")))
                if fused:
                    yield           # (may only be recorded) dtok is valid from here on
            return d
        return x, bwd

    def _stack_operands(self, pfx, depth):
        """what the layer-fused stack kernels step through at constant strides, or None: (the seven fp32 parameter vectors of
        layer 0, their stride in floats, the stacked packed MLP weights w1 / w1T / w2 / w2T)"""
        names = ("0.fn.norm.weight", "0.fn.norm.bias", "0.fn.fn.to_out.0.bias", "1.fn.norm.weight", "1.fn.norm.bias",
                 "1.fn.fn.net.0.bias", "1.fn.fn.net.3.bias")
        stride = None
        for n in names:
            for i in range(1, depth):
                k0, k1 = "%s.layers.%d.%s" % (pfx, i - 1, n), "%s.layers.%d.%s" % (pfx, i, n)
                if k0 not in self.p or k1 not in self.p:
                    return None
                d = self.p[k1].data_ptr() - self.p[k0].data_ptr()
                if d % 4 or (stride is not None and d // 4 != stride):
                    return None
                stride = d // 4
        w1, w2 = self.wstack.get((pfx, "net.0")), self.wstack.get((pfx, "net.3"))
        if stride is None or w1 is None or w2 is None or w1[0].shape[0] != depth or w2[0].shape[0] != depth:
            return None
        if self.need_grad and (w1[1] is None or w2[1] is None):
            return None
        params0 = tuple(self.p["%s.layers.0.%s" % (pfx, n)] for n in names)
        return params0, stride, w1[0], w1[1], w2[0], w2[1]

    def _layer_grad_stride(self, pfx, depth):
        """floats between consecutive layers' gradients of the nine tensors the fused decoder backward accumulates, or None"""
        names = ("1.fn.fn.net.0.weight", "1.fn.fn.net.3.weight", "1.fn.fn.net.0.bias", "1.fn.fn.net.3.bias", "0.fn.fn.to_out.0.bias",
                 "0.fn.norm.weight", "0.fn.norm.bias", "1.fn.norm.weight", "1.fn.norm.bias")
        stride = None
        for n in names:
            for i in range(1, depth):
                k0, k1 = "%s.layers.%d.%s" % (pfx, i - 1, n), "%s.layers.%d.%s" % (pfx, i, n)
                if k0 not in self.g or k1 not in self.g:
                    return None
                d = self.g[k1].data_ptr() - self.g[k0].data_ptr()
                if d % 4 or (stride is not None and d // 4 != stride):
                    return None
                stride = d // 4
        return stride

    def _prep_stack(self, tok, tok_b, tok_s, B, images, L, heads, dim_head, pfx, depth):
        """ops.XattnPrepStack for the `depth` layers of decoder `pfx`, or None when the layers' parameters do not sit
        at one constant pitch in the flat arenas (then every layer is prepared on its own)"""
        names = ("norm.weight", "norm.bias", "fn.to_q.weight", "fn.to_k.weight", "fn.to_v.weight", "fn.to_out.0.weight")
        key = lambda i, n: "%s.layers.%d.0.fn.%s" % (pfx, i, n)
        stride = None
        for table in (self.p, self.g):
            for n in names:
                if any(key(i, n) not in table for i in range(depth)):
                    if table is self.g and not self.need_grad:
                        continue
                    return None
                for i in range(1, depth):
                    d = table[key(i, n)].data_ptr() - table[key(i - 1, n)].data_ptr()
                    if d % 4 or (stride is not None and d // 4 != stride):
                        return None
                    stride = d // 4
        a0 = "%s.layers.0.0.fn" % pfx
        return ops.XattnPrepStack(tok, tok_b, tok_s, B, images, L, heads, dim_head, depth, stride,
                                  self.p[a0 + ".norm.weight"], self.p[a0 + ".norm.bias"], self.p[a0 + ".fn.to_q.weight"],
                                  self.xstack[(pfx, "to_k.weight")], self.xstack[(pfx, "to_v.weight")],
                                  self.xstack[(pfx, "to_out.0.weight")], self.dtype, ATTN_SCALE, LN_EPS,
                                  masters=(self.p[a0 + ".fn.to_k.weight"], self.p[a0 + ".fn.to_v.weight"],
                                           self.p[a0 + ".fn.to_out.0.weight"], self.xstack[(pfx, "to_q.weight")]),
                                  record=not self.attn_fp8)      # the fp8 layers launch at once: so must what they read

    def _dec_layer_fused(self, x0, images, rpi, tok, tok_b, tok_s, B, dtok, a, f, heads, dim_head, L, mlp, stack=None,
                         li=0, partial=None):
        """one decoder layer = one HIP kernel per direction (csrc/decoder_fused.hip) + the per-image operand prep
        (`stack`: prepared for all layers at once by ops.XattnPrepStack; its backward runs once after the last layer)"""
        g1, b1 = self.p[a + ".norm.weight"], self.p[a + ".norm.bias"]
        wq, wk, wv = (self.p[a + ".fn.to_%s.weight" % n] for n in "qkv")
        wo, bo = self.p[a + ".fn.to_out.0.weight"], self.p[a + ".fn.to_out.0.bias"]
        g2, b2 = self.p[f + ".norm.weight"], self.p[f + ".norm.bias"]
        w1k, w2k = f + ".fn.net.0.weight", f + ".fn.net.3.weight"
        fb1, fb2 = self.p[f + ".fn.net.0.bias"], self.p[f + ".fn.net.3.bias"]
        wqT, wkT, wvT = (self.pk[a + ".fn.to_%s.weight" % n].dgrad for n in "qkv")
        woT = self.pk[a + ".fn.to_out.0.weight"].dgrad
        prep = stack.layer(li) if stack is not None else \
            ops.XattnPrep(tok, tok_b, tok_s, B, images, L, heads, dim_head, g1, b1, wq, wkT, wvT, woT, self.dtype,
                          ATTN_SCALE, LN_EPS, masters=(wk, wv, wo, wqT))
        y = ops.decoder_layer_fwd(x0, prep, rpi, g1, b1, bo, g2, b2, self.pk[w1k].fwd, fb1, self.pk[w2k].fwd, fb2, mlp,
                                  LN_EPS, fp8=self.attn_fp8)
        if not self.need_grad:
            return y, None

        def bwd(dy):
            grads = (self.g[w1k], self.g[w2k], self.g[f + ".fn.net.0.bias"], self.g[f + ".fn.net.3.bias"],
                     self.g[a + ".fn.to_out.0.bias"], self.g[a + ".norm.weight"], self.g[a + ".norm.bias"],
                     self.g[f + ".norm.weight"], self.g[f + ".norm.bias"])
            if stack is not None:
                dx, _, _ = ops.decoder_layer_bwd(x0, dy, prep, rpi, g1, b1, bo, g2, b2, self.pk[w1k].fwd,
                                                 self.pk[w1k].dgrad, fb1, self.pk[w2k].fwd, self.pk[w2k].dgrad, fb2,
                                                 grads, mlp, LN_EPS, dkq=stack.dkq[li], dvoT=stack.dvoT[li], partial=partial)
                return dx
            dx, dkq, dvoT = ops.decoder_layer_bwd(x0, dy, prep, rpi, g1, b1, bo, g2, b2, self.pk[w1k].fwd,
                                                  self.pk[w1k].dgrad, fb1, self.pk[w2k].fwd, self.pk[w2k].dgrad, fb2,
                                                  grads, mlp, LN_EPS)
            ops.xattn_prep_bwd(prep, tok, dtok, g1, wqT, wk, wv, wo, dkq, dvoT, self.g[a + ".norm.weight"],
                               self.g[a + ".norm.bias"], *(self.g[a + ".fn.to_%s.weight" % n] for n in "qkv"),
                               self.g[a + ".fn.to_out.0.weight"], True, self.dtype)
            return dx
        return y, bwd

    def _dec_attn(self, x0, images, tok, tok_b, tok_s, B, dtok, a, heads, dim_head, L):
        g1, b1 = self.p[a + ".norm.weight"], self.p[a + ".norm.bias"]
        wq, wk, wv = (self.p[a + ".fn.to_%s.weight" % n] for n in "qkv")
        wo, bo = self.p[a + ".fn.to_out.0.weight"], self.p[a + ".fn.to_out.0.bias"]
        xn, st1 = ops.layernorm(x0, g1, b1, LN_EPS)
        wqT, wkT, wvT = (self.pk[a + ".fn.to_%s.weight" % n].dgrad for n in "qkv")
        woT = self.pk[a + ".fn.to_out.0.weight"].dgrad
        prep = ops.XattnPrep(tok, tok_b, tok_s, B, images, L, heads, dim_head, g1, b1, wq, wkT, wvT, woT, self.dtype,
                             ATTN_SCALE, LN_EPS)
        HLP = prep.HLP
        dots = ops.linear(xn, prep.kq, HLP, images=images, w_image_stride=HLP * DIM)
        attn = ops.softmax_groups(dots, heads, L)
        x1 = ops.linear(attn, prep.voT, DIM, bias=bo, residual=x0, images=images, w_image_stride=HLP * DIM)
        if not self.need_grad:
            return x1, None
        dev = x0.device

        def bwd(dx1):
            dattn = ops.linear(dx1, prep.vo, HLP, images=images, w_image_stride=HLP * DIM)
            dvoT = torch.empty(images, DIM, HLP, dtype=torch.float32, device=dev)
            ops.linear_wgrad(attn, dx1, dvoT, images=images, per_image=True, use_tr=self.use_tr)
            ops.colsum(dx1, self.g[a + ".fn.to_out.0.bias"], accumulate=True)
            ddots = ops.softmax_groups_bwd(attn, dattn, heads, L)
            dxn = ops.linear(ddots, prep.kqT, DIM, images=images, w_image_stride=HLP * DIM)
            dkq = torch.empty(images, HLP, DIM, dtype=torch.float32, device=dev)
            ops.linear_wgrad(xn, ddots, dkq, images=images, per_image=True, use_tr=self.use_tr)
            dx = ops.layernorm_bwd(dxn, x0, st1, g1, self.g[a + ".norm.weight"], self.g[a + ".norm.bias"],
                                   dx_add=dx1, accumulate=True)
            ops.xattn_prep_bwd(prep, tok, dtok, g1, wqT, wk, wv, wo, dkq, dvoT, self.g[a + ".norm.weight"],
                               self.g[a + ".norm.bias"], *(self.g[a + ".fn.to_%s.weight" % n] for n in "qkv"),
                               self.g[a + ".fn.to_out.0.weight"], True, self.dtype)
            return dx
        return x1, bwd

    # ---- whole nets ------------------------------------------------------------------------------
    def forward(self, x1, x2, training, need_grad):
        ops.set_f32_mma_mode(self.mma_fwd)        # library state per host thread: every entry point sets it
        self.training, self.need_grad = training, need_grad
        if need_grad and self.use_side and self.side is None and self.dtype == torch.bfloat16:
            self.side = ops.SideStream(x1.device)
        self._pack_all()
        if self.cfg["kind"] == "bit":
            logits, bwd = self._bit(x1, x2)
        else:
            logits, bwd = self._unet(x1, x2)        # "unet" and "xbd" share the trunk / up path; levels differ
        if need_grad:
            self._bwd = bwd        # a no-grad forward (validation, metrics) leaves a pending backward alone
        # fp16-plane forward (form 3): operands beyond fp16's range (|x| >= 65504; |w| >= 255, which BatchNorm-folded eval weights
        # of near-dead channels can reach) become inf / NaN without any other sign.  The FIRST forward of an engine in that form is
        # checked once (one synchronisation, outside any stream capture) and fails loudly, naming the six-product bf16 form.
        if self.mma_fwd == 3 and not self._x3_checked and not torch.cuda.is_current_stream_capturing():
            self._x3_checked = True
            if not bool(torch.isfinite(logits).all()):
                raise FloatingPointError(
                    "dahitra_amd: compute_dtype='bf16x3' produced non-finite logits in its fp16-plane forward (operands beyond "
                    "fp16's range: |x| < 65504, |w| < 255); set DAHITRA_X3_FWD=2 (three bf16 planes, six products, fp32's range)")
        return logits

    def split_prefixes(self):
        """key prefixes of the parameters whose gradients the SECOND part of a split backward writes (backward_second)"""
        if self.cfg["kind"] == "bit":
            return ("resnet.conv1.", "resnet.bn1.", "resnet.layer1.", "resnet.layer2.")
        return ("resnet.",)

    def take_backward(self):
        """the backward closure of the forward that just ran (the autograd node of THAT forward keeps it, so several
        grad-enabled forwards may be outstanding, as with the reference's autograd graph)"""
        bwd, self._bwd = self._bwd, None
        return bwd

    def backward(self, dlogits_nchw, bwd=None):
        ops.set_f32_mma_mode(self.mma_bwd)
        if bwd is None:
            bwd, self._bwd = self._bwd, None
        if bwd is None:
            raise RuntimeError("dahitra_amd: backward called without a grad-enabled forward")
        plan = self._plans.get(True)
        if plan is not None:       # a no-grad forward in between switched self.pk to the forward-only packed weights
            self.pk, self.xstack, self.wstack = plan[1], plan[2], plan[3]
        if self._wgrad_plan is None:
            self._wgrad_plan = ops.WgradPlan(dlogits_nchw.device, batch=True)
        with self._wgrad_plan as plan:      # the conv layers' split-K reduces: one launch at the end of the pass
            bwd(dlogits_nchw)
            self._side_join()
            plan.run()

    # A backward pass in TWO parts (data-parallel overlap, dahitra_amd/graph.py): after backward_first() every gradient
    # from resnet.layer3 to the end of the arena is final (its split-K reduces have run) and can be all-reduced while
    # backward_second() computes the stem / layer1 / layer2 gradients.
    def backward_first(self, dlogits_nchw, bwd):
        ops.set_f32_mma_mode(self.mma_bwd)
        split = getattr(bwd, "split", None)
        if split is None:
            raise RuntimeError("dahitra_amd: this net's backward has no split point")
        plan = self._plans.get(True)
        if plan is not None:
            self.pk, self.xstack, self.wstack = plan[1], plan[2], plan[3]
        if self._wgrad_plan is None:
            self._wgrad_plan = ops.WgradPlan(dlogits_nchw.device, batch=True)
        self._wgrad_plan.__enter__()
        try:
            self._split_state = (split[1], split[0](dlogits_nchw))
            self._side_join()
            self._wgrad_plan.run()
        except BaseException:
            self._wgrad_plan.__exit__(*sys.exc_info())      # abort: what the failed pass recorded must not be launched
            raise

    def run_deferred_wgrad(self):
        d, self._deferred_wgrad = self._deferred_wgrad, None
        if d is None:
            return
        x, dy, wkey, ks, stride, pad, dilation = d
        with self.side.fork(x.y if isinstance(x, ops.BnInput) else (x.a if isinstance(x, ops.Up4Input) else x), dy):
            ops.conv2d_wgrad(x, dy, self.g[wkey], ks, stride, pad, accumulate=True, use_tr=self.use_tr, dilation=dilation)

    def _side_join(self):
        if self.side is not None:
            self.run_deferred_wgrad()
            self.side.join()

    def backward_second(self):
        ops.set_f32_mma_mode(self.mma_bwd)
        second, state = self._split_state
        self._split_state = None
        try:
            second(state)
            self._wgrad_plan.run()
        finally:
            self._wgrad_plan.__exit__(*sys.exc_info())      # (None, None, None) on success; an exception aborts the batch

    def _head_out(self, h, wkey, bkey):
        """final 3x3 conv to n_class logits, returned as NCHW fp32 (the reference's output layout)"""
        ncls = self.shapes[wkey][0]
        ck = ops.chunk_channels(self.dtype)
        if self.pk[wkey].fwd.shape[-2] == 16 and self.fused_head_out:
            out = ops.conv3x3_head(h, self.pk[wkey].fwd, ncls, self.p[bkey], w_oihw=self.p[wkey])      # fp32 NCHW logits from the conv itself
        else:
            out = ops.nhwc_to_nchw(ops.conv2d(h, self.pk[wkey].fwd, ncls, 3, 1, 1, bias=self.p[bkey]))
        if not self.need_grad:
            return out, None

        small = 8 if (self.dtype == torch.bfloat16 or ncls > 4) else 4          # channels of one 16-byte piece
        direct = h.shape[-1] == 32 and ncls <= small

        def bwd(dl_nchw, next_gate=None, head_bn=None, relu_out=None):
            """relu_out: `h` is the output of a ReLU (the tensor itself): the returned gradient is already masked by it"""
            if direct and next_gate is None:
                # n_class (2..5) real channels: keep dlogits at ONE 16-byte piece per pixel instead of padding them to
                # a 32-channel K-chunk (4x the bytes, written and read twice), and take the data gradient with the
                # dedicated head kernel instead of an MFMA convolution over 94 % zeros
                # bf16, or the fp32 pipeline in its split-product mode (three bf16 products: NOT the exact fp32 mode)
                fuse = self.fused_head_bn and ncls <= 8 and self.g[wkey].is_contiguous() and \
                    (self.dtype == torch.bfloat16 or (self.dtype == torch.float32 and ops.get_f32_mma_mode() != 0))
                if head_bn is not None and fuse:
                    # data gradient, weight and bias gradient of this convolution: all inside the BatchNorm backward behind it
                    return HeadGrad(ops.head_dlogits_pack(dl_nchw, self.dtype), self.p[wkey], ncls, self.g[wkey], self.g[bkey])
                if relu_out is not None and fuse and relu_out is h:
                    # the head behind a ReLU: data gradient (masked) + weight / bias gradient in one pass over the head's input
                    return ops.head_relu_bwd(ops.head_dlogits_pack(dl_nchw, self.dtype), self.p[wkey], ncls, h, self.g[wkey],
                                             self.g[bkey])
                dl = ops.nchw_to_nhwc(dl_nchw, self.dtype, cpad=small)
                ops.conv2d_wgrad(h, dl, self.g[wkey], 3, 1, 1, accumulate=True, use_tr=self.use_tr, cout_real=ncls)
                tmp = torch.empty(small, dtype=torch.float32, device=h.device)
                ops.colsum(dl.view(-1, small), tmp)
                ops.reduce_rows(tmp, 1, ncls, self.g[bkey], accumulate=True)
                if head_bn is not None and self.gated_head_dgrad and self.dtype == torch.bfloat16 and ncls <= 2:
                    # the BatchNorm behind these 32 channels: mask + reduction here, only its apply pass remains
                    return Gated(*ops.head_dgrad3x3_bn(dl, self.p[wkey], ncls, *head_bn))
                return ops.head_dgrad3x3(dl, self.p[wkey], ncls, relu_out=relu_out)
            dl = ops.nchw_to_nhwc(dl_nchw, self.dtype, cpad=ck)          # channels zero-padded to one K-chunk
            ops.conv2d_wgrad(h, dl, self.g[wkey], 3, 1, 1, accumulate=True, use_tr=self.use_tr, cout_real=ncls)
            tmp = torch.empty(ck, dtype=torch.float32, device=h.device)
            ops.colsum(dl.view(-1, ck), tmp)
            ops.reduce_rows(tmp, 1, ncls, self.g[bkey], accumulate=True)
            dh = self.conv_dgrad(dl, wkey, 3, 1, 1, h.shape, gate=next_gate)
            return ops.act_bwd(dh, relu_out, RELU) if relu_out is not None else dh
        return out, bwd

    def _bit(self, x1, x2):
        cfg, L = self.cfg, self.cfg["token_len"]
        B = x1.shape[0]
        S2 = 2 * B
        xp, xarg, xshape, b_stem = self.stem(x1, x2, 2, pool=True)
        self._pack_join()
        if cfg.get("backbone") == "resnet50":
            l1, b_l1 = self.res50_layer(xp, 1, 1, 1, 1, 2)
            l2, b_l2 = self.res50_layer(l1, 2, 2, 1, 1, 2)
            l3, b_l3 = self.res50_layer(l2, 3, 1, 1, 2, 2)       # stride replaced by dilation 2 (honoured)
        else:
            l1, b_l1 = self.res_layer(xp, 1, 1, 2)
            l2, b_l2 = self.res_layer(l1, 2, 2, 2)
            l3, b_l3 = self.res_layer(l2, 3, 1, 2)
        phased = self.phase_conv_pred and l3.shape[-1] % 64 == 0
        if phased:
            feat, b_pred = self.conv_pred_phase(l3)
        else:
            up = ops.upsample2(l3)
            feat, b_pred = self.conv_act(up, "conv_pred.weight", "conv_pred.bias", 3, 1, NONE)
        _, fh, fw, _ = feat.shape
        hw = fh * fw
        wa = self.p["conv_a.weight"]
        tok_cat, tsaved = ops.tokenizer_fwd(feat, wa, self.p["pos_embedding"], B, L)
        tok2d, b_enc = self.encoder(tok_cat.view(B * 2 * L, DIM), "transformer", cfg["enc_depth"], BIT_HEADS,
                                    BIT_DIM_HEAD, B, 2 * L)
        dtok = torch.zeros_like(tok2d) if self.need_grad else None
        dec, b_dec = self.decoder(feat.view(S2 * hw, DIM), S2, tok2d, 2 * L * DIM, L * DIM, B, dtok,
                                  "transformer_decoder", cfg["dec_depth"], BIT_HEADS, cfg["dec_dim_head"], L)
        dec4 = dec.view(S2, fh, fw, DIM)
        # bf16: |A - B| + bilinear x4 happen inside classifier.0's loads (ops.Up4Input: forward and weight gradient; the data
        # gradient goes straight to the coarse maps below) -- the 32 x 4h x 4w map is never written or read
        k0 = "classifier.0.weight"
        if self.fused_up4_fwd and self.dtype == torch.bfloat16 and DIM == 32 and tuple(self.shapes[k0][:2]) == (32, 32) and \
                tuple(self.pk[k0].fwd.shape) == (9, 32, 32):
            upd = ops.Up4Input(dec4[:B].contiguous(), dec4[B:].contiguous())
        else:
            upd = ops.absdiff_upsample4(dec4[:B], dec4[B:])
        h, b_c0 = self.conv_bn(upd, k0, "classifier.1", 3, 1, 1, 1, True, lazy=True, side_wgrad=True)
        logits, b_out = self._head_out(h, "classifier.3.weight", "classifier.3.bias")
        if not self.need_grad:
            return logits, None

        def bwd_first(dl):
            """head, decoder, tokens, conv_pred, layer3: everything whose parameters sit behind layer2 in the arena"""
            dh = b_out(dl, next_gate=b_c0.gate, head_bn=getattr(b_c0, "bn", None))
            fuse = self.fused_up4_bwd and self.dtype == torch.bfloat16 and fh % 2 == 0 and fw % 4 == 0 and \
                self.pk["classifier.0.weight"].dgrad.dim() == 3 and self.pk["classifier.0.weight"].dgrad.shape[1] == 32
            if fuse:
                dec_g = torch.empty_like(dec4)
                b_c0(dh, through_up4=(dec4[:B], dec4[B:], dec_g[:B], dec_g[B:]))
            else:
                dupd, _ = b_c0(dh)
                dec_g = torch.empty_like(dec4)
                ops.absdiff_upsample4_bwd_into(dec4[:B], dec4[B:], dupd, dec_g[:B], dec_g[B:])
            dfeat = b_dec(dec_g.view(S2 * hw, DIM))                    # also fills dtok
            if self.side is not None:
                self.run_deferred_wgrad()            # the classifier's weight gradient, next to the encoder's backward
            dtok_cat = b_enc(dtok)
            dfeat4 = dfeat.view(S2, fh, fw, DIM)
            ops.tokenizer_bwd(feat, wa, tsaved, dtok_cat, dfeat4, self.g["conv_a.weight"], self.g["pos_embedding"],
                              B, L, accumulate=True)
            dl3 = b_pred(dfeat4) if phased else ops.upsample2_bwd(b_pred(dfeat4))
            return b_l3(dl3, next_gate=b_l2.gate)

        def bwd_second(dl2):
            dxp = b_l1(b_l2(dl2, next_gate=b_l1.gate))
            if hasattr(b_stem, "from_pool"):
                b_stem.from_pool(xarg, dxp)
            else:
                b_stem(ops.maxpool_bwd(xarg, dxp, xshape))

        def bwd(dl):
            bwd_second(bwd_first(dl))
        bwd.split = (bwd_first, bwd_second, ("resnet.conv1.", "resnet.bn1.", "resnet.layer1.", "resnet.layer2."))
        return logits, bwd

    # hierarchical model -------------------------------------------------------------------------
    def _zeros_like(self, t):
        """zeros of t's shape (fp32): a slice of the pool _unet zeroed with one launch while it lasts, else torch.zeros_like"""
        zp, n = getattr(self, "_zpool", None), t.numel()
        if zp is not None and t.dtype == torch.float32 and zp[1] + n <= zp[0].numel():
            out = zp[0][zp[1]:zp[1] + n].view(t.shape)
            zp[1] += (n + 63) // 64 * 64
            return out
        return torch.zeros_like(t)

    def _run_staged(self, gens, ew=False):
        """GENERATORS that pause right after every token-encoder call and every fused decoder stack (forward: `_level`;
        backward: the `bwd` it returns).  They run in rounds inside one ops.EncoderBatch: the launches they reach are only
        RECORDED, and after each round the recorded ones go out together -- the levels are independent, an encoder stack
        occupies 2 x batch workgroups, the decoder stacks of the 16 x 16 and 32 x 32 levels a fraction of the chip.
        Between two such launches every level issues its OWN small kernels (1x1 squeeze, tokenizer, cross-attention operand
        preparation and its gradients, positional adds, channel copies: 138 launches under 8 us in the DAHiTra step, most of
        them a few dozen workgroups): with `level_streams` each level's segment goes to a stream of its own, forked from and
        joined into the main stream around the round -- in the recorded step they are parallel branches of the graph and run
        side by side instead of one after the other.
        ew=True (the `_level` generators, which pause after them): the levels' positional adds, channel concatenations and token
        differences (and their gradients) are recorded as well and go out as one job-table launch per round (ops.EwBatch).
        Returns the generators' return values."""
        vals = {}
        streams = None
        if self.level_streams and len(gens) > 1 and ops.PROFILE is None:
            dev = torch.cuda.current_device()
            while len(self._lstreams) < len(gens):
                self._lstreams.append(ops.SideStream(torch.device("cuda", dev)))
            streams = self._lstreams
        with ops.EncoderBatch(decoder=True, ew=ew and streams is None) as eb:
            live = list(enumerate(gens))
            while live:
                paused = []
                for i, g in live:
                    try:
                        if streams is not None:
                            with streams[i].fork():
                                next(g)
                        else:
                            next(g)
                        paused.append((i, g))
                    except StopIteration as e:
                        vals[id(g)] = e.value
                if streams is not None:
                    for st in streams:
                        st.join()
                eb.launch()          # what the levels recorded in this round: one launch per kernel family
                live = paused
        return [vals[id(g)] for g in gens]

    def _level(self, l, xa_b, B):
        """one _forward_trans_module (networks.py:1297-1318) on the [A;B] batch of trunk taps.  A generator (see _run_staged):
        pauses after the encoder call, returns (out4, bwd) where bwd(dout4) is a generator of the same kind"""
        lv, L = UNET_LEVELS[l], self.cfg["token_len"]
        S2 = 2 * B
        sq, b_sq = self.conv_act(xa_b, "conv_squeeze_%d.0.weight" % l, None, 1, 0, RELU)
        _, fh, fw, _ = sq.shape
        hw = fh * fw
        wa = self.p["conv_token_%d.weight" % l]
        tok_cat, tsaved = ops.tokenizer_fwd(sq, wa, self.p["pos_embedding_%d" % l], B, L)
        tok2d, b_enc = self.encoder(tok_cat.view(B * 2 * L, DIM), "transformer_%d" % l, self.cfg["enc_depth"],
                                    lv["heads"], lv["dim_head"], B, 2 * L)
        pos = self.p["pos_embedding_decoder_%d" % l]
        xin = ops.add_pos(sq, pos)  # (may only be recorded: issued with the round's other launches)
        yield                       # tok2d and xin are valid from here on (_run_staged)
        dtok = self._zeros_like(tok2d) if self.need_grad else None
        dp = "transformer_decoder_%d" % l
        dec, b_dec = yield from self._decoder_gen(xin.view(S2 * hw, DIM), S2, tok2d, 2 * L * DIM, L * DIM, B, dtok, dp,
                                                  lv["dec_depth"], lv["heads"], lv["dim_head"], L)
        dec4 = dec.view(S2, fh, fw, DIM)
        # third pass: decoder(conv_decode(cat[x1, x2]), |tok2 - tok1|) with the SAME weights
        cat = ops.cat_halves(dec4)
        tk3 = tok2d.view(B, 2, L * DIM)                      # [b][stream][L*32]
        dtk = torch.empty(B, L, DIM, dtype=torch.float32, device=sq.device)
        ops.absdiff_halves(tk3, dtk)
        ddtk = self._zeros_like(dtk) if self.need_grad else None
        yield                       # cat and dtk are valid from here on
        dxc, b_cd = self.conv_act(cat, "conv_decode_%d.weight" % l, None, 3, 1, NONE)
        xin3 = ops.add_pos(dxc, pos)        # (recorded: a round issues the element-wise jobs before its decoder stacks)
        out, b_dec3 = yield from self._decoder_gen(xin3.view(B * hw, DIM), B, dtk.view(B * L, DIM), L * DIM, 0, B, ddtk, dp,
                                                   lv["dec_depth"], lv["heads"], lv["dim_head"], L)
        out4 = out.view(B, fh, fw, DIM)
        if not self.need_grad:
            return out4, None
        gpos = self.g["pos_embedding_decoder_%d" % l]

        def bwd(dout4):
            dxin3 = (yield from b_dec3(dout4.reshape(B * hw, DIM))).view(B, fh, fw, DIM)
            ops.add_pos_bwd(dxin3, gpos, accumulate=True)
            dcat = b_cd(dxin3)
            ops.absdiff_halves_bwd(tk3, ddtk, dtok)                    # accumulates into both token halves
            ddec = ops.split_halves(dcat)      # (these three may only be recorded: issued before the round's decoder stacks)
            dxin = (yield from b_dec(ddec.view(S2 * hw, DIM))).view(S2, fh, fw, DIM)
            ops.add_pos_bwd(dxin, gpos, accumulate=True)
            dtok_cat = b_enc(dtok)
            yield                   # dtok_cat is valid from here on (_run_staged)
            ops.tokenizer_bwd(sq, wa, tsaved, dtok_cat, dxin, self.g["conv_token_%d.weight" % l],
                              self.g["pos_embedding_%d" % l], B, L, accumulate=True)
            yield                   # (may only be recorded) dxin is complete from here on
            return b_sq(dxin)
        return out4, bwd

    def _xbd_level(self, l, xa_b, B):
        """(a generator like _level) _forward_trans_module of the xBD copy (xBD_code/zoo/model_transformer_encoding.py:385-406): squeeze, tokens,
        encoder, then ONE decoder pass on conv_decode(cat[x1, x2]) against |token2 - token1|.  Positional terms exist
        only in the level-5 call, which receives the *_3 embeddings (layer index 3, lines 358-383)."""
        lv, L = UNET_LEVELS[l], self.cfg["token_len"]
        sq, b_sq = self.conv_act(xa_b, "conv_squeeze_%d.0.weight" % l, None, 1, 0, RELU)
        _, fh, fw, _ = sq.shape
        hw = fh * fw
        wa = self.p["conv_token_%d.weight" % l]
        with_pos = l == 5
        pos_tok = self.p["pos_embedding_3"] if with_pos else torch.zeros(1, 2 * L, DIM, device=sq.device)
        tok_cat, tsaved = ops.tokenizer_fwd(sq, wa, pos_tok, B, L)
        tok2d, b_enc = self.encoder(tok_cat.view(B * 2 * L, DIM), "transformer_%d" % l, self.cfg["enc_depth"],
                                    lv["heads"], lv["dim_head"], B, 2 * L)
        yield                       # tok2d is valid from here on (_run_staged)
        tk3 = tok2d.view(B, 2, L * DIM)
        dtk = torch.empty(B, L, DIM, dtype=torch.float32, device=sq.device)
        ops.absdiff_halves(tk3, dtk)
        ddtk = self._zeros_like(dtk) if self.need_grad else None
        cat = ops.cat_halves(sq)
        dxc, b_cd = self.conv_act(cat, "conv_decode_%d.weight" % l, None, 3, 1, NONE)
        pos = self.p["pos_embedding_decoder_3"] if (with_pos and self.cfg["decoder_pos"]) else None
        xin = ops.add_pos(dxc, pos) if pos is not None else dxc
        out, b_dec = yield from self._decoder_gen(xin.view(B * hw, DIM), B, dtk.view(B * L, DIM), L * DIM, 0, B, ddtk,
                                                  "transformer_decoder_%d" % l, lv["dec_depth"], lv["heads"], lv["dim_head"], L)
        out4 = out.view(B, fh, fw, DIM)
        if not self.need_grad:
            return out4, None

        def bwd(dout4):
            dxin = (yield from b_dec(dout4.reshape(B * hw, DIM))).view(B, fh, fw, DIM)
            if pos is not None:
                ops.add_pos_bwd(dxin, self.g["pos_embedding_decoder_3"], accumulate=True)
            dcat = b_cd(dxin)
            dtok = torch.zeros_like(tok2d)
            ops.absdiff_halves_bwd(tk3, ddtk, dtok)
            dtok_cat = b_enc(dtok)
            yield                   # dtok_cat is valid from here on (_run_staged)
            dsq = ops.split_halves(dcat)
            gpos = self.g["pos_embedding_3"] if with_pos else torch.zeros(1, 2 * L, DIM, device=sq.device)
            ops.tokenizer_bwd(sq, wa, tsaved, dtok_cat, dsq, self.g["conv_token_%d.weight" % l], gpos, B, L,
                              accumulate=True)
            yield                   # (may only be recorded) dsq is complete from here on
            return b_sq(dsq)
        return out4, bwd

    def _up_conv(self, l, x):
        """relu(conv_layer<l>(nearest-upsample-x2(x))) (models/networks.py:1341-1351).  Default: the four 2x2 phase convolutions
        of conv_pred_phase on x itself -- the upsampled tensor (134 MB at 256 x 256, batch 32) is neither written nor read, by
        the forward, the data gradient or the weight gradient, and the matrix work is 2.25x smaller.  DAHITRA_NO_PHASE_UPCONV=1:
        upsample kernel + 3x3 convolution."""
        wkey, bkey = "conv_layer%d.0.weight" % l, "conv_layer%d.0.bias" % l
        if not (self.phase_up_conv and x.shape[-1] == 32 and self.shapes[wkey][0] == 32):
            up = ops.upsample2(x)
            out, b = self.conv_act(up, wkey, bkey, 3, 1, RELU)
            if not self.need_grad:
                return out, None
            return out, (lambda d, masked=False: ops.upsample2_bwd(b(d, masked=masked)))
        wf, wd, b4 = ops.pack_phase_weights(self.p[wkey], self.p[bkey], self.dtype)
        out = ops.conv_up2_fwd(x, wf, b4, act=RELU)
        if not self.need_grad:
            return out, None

        def bwd(dout, masked=False):
            """masked: dout already carries this layer's ReLU mask (the class head's data gradient applied it)"""
            dy = dout if masked else ops.act_bwd(dout, out, RELU)
            ops.conv_up2_wgrad(x, dy, self.g[wkey], accumulate=True, use_tr=self.use_tr)
            ops.colsum(dy.view(-1, 32), self.g[bkey], accumulate=True)
            return ops.conv_up2_dgrad(dy, wd, 32)
        return out, bwd

    def _unet(self, x1, x2):
        B = x1.shape[0]
        s2, b_stem = self.stem(x1, x2, 2)                      # [2B,128,128,64] (post-ReLU tap)
        self._pack_join()
        p4, arg4 = ops.maxpool(s2, want_arg=True)
        s4, b_l1 = self.res_layer(p4, 1, 1, 2)                 # 64x64x64
        s8, b_l2 = self.res_layer(s4, 2, 2, 2)                 # 32x32x128
        p16, arg16 = ops.maxpool(s8, want_arg=True)
        s16, b_l3 = self.res_layer(p16, 3, 1, 2)               # 16x16x256
        level = self._xbd_level if self.cfg["kind"] == "xbd" else self._level
        if self.need_grad:
            # the levels' token-gradient accumulators (dtok, ddtk: 3 B L 32 floats per level) from ONE zeroed buffer
            self._zpool = [torch.zeros(3 * 3 * B * self.cfg["token_len"] * DIM, dtype=torch.float32, device=s4.device), 0]
        ew = self.cfg["kind"] != "xbd"          # (_level pauses after its recorded element-wise calls; _xbd_level does not)
        (o5, b5), (t4, b4), (t3, b3) = self._run_staged([level(5, s16, B), level(4, s8, B), level(3, s4, B)], ew=ew)
        self._zpool = None
        o5u = ops.upsample2(o5)
        o4, bu4 = self._up_conv(4, ops.add(t4, o5u))
        o3, bu3 = self._up_conv(3, ops.add(t3, o4))
        _, h2, w2, c2 = s2.shape
        # cat([a_128, b_128], 1) (networks.py:1344) is read IN PLACE by conv_layer2_0.0, its weight gradient, and written in
        # place by its data gradient (ops.SplitCat: the two streams are the halves of s2); otherwise two channel copies each way
        k20 = "conv_layer2_0.0.weight"
        split2 = self.split_cat and self.training and self.pk[k20].fwd_frag is not None and \
            (not self.need_grad or self.pk[k20].dgrad_frag is not None) and \
            ops.conv3x3_split_supported(B, h2, w2, 2 * c2, self.shapes[k20][0], self.dtype)
        cat2 = ops.SplitCat(s2) if split2 else ops.SplitCat(s2).materialize()
        y, b20 = self.conv_bn(cat2, k20, "conv_layer2_0.1", 3, 1, 1, 1, True, lazy=True)
        y2 = ops.conv2d(y, self.pk["conv_layer2_0.3.weight"].fwd, 32, 3, 1, 1, bias=self.p["conv_layer2_0.3.bias"],
                        residual=o3)
        o2, bu2 = self._up_conv(2, y2)
        logits, b_out = self._head_out(o2, "classifier.weight", "classifier.bias")
        if not self.need_grad:
            return logits, None

        def bwd_first(dl):
            """head, top-down path and the three levels: everything whose parameters sit behind the trunk in the arena"""
            do2 = b_out(dl, relu_out=o2)                       # (the ReLU of conv_layer2 applied to the gradient in that kernel)
            dy2 = bu2(do2, masked=True)                        # grad of (conv_layer2_0 out + o3)
            ops.conv2d_wgrad(y, dy2, self.g["conv_layer2_0.3.weight"], 3, 1, 1, accumulate=True, use_tr=self.use_tr)
            ops.colsum(dy2.view(-1, 32), self.g["conv_layer2_0.3.bias"], accumulate=True)
            dy = self.conv_dgrad(dy2, "conv_layer2_0.3.weight", 3, 1, 1, y.shape, gate=b20.gate)
            dcat2, _ = b20(dy)
            if split2:
                ds2 = dcat2                                    # already the [2B, H, W, c2] gradient of s2's two halves
            else:
                ds2 = ops.split_halves(dcat2)
            dsum3 = bu3(dy2)                                   # d(t3 + o4)
            dsum4 = bu4(dsum3)                                 # d(t4 + o5u)
            ds4, ds8, ds16 = self._run_staged([b3(dsum3), b4(dsum4), b5(ops.upsample2_bwd(dsum4))], ew=ew)
            return ds16, ds8, ds4, ds2

        def bwd_second(state):
            """the trunk: layer3, layer2, layer1, stem (the `resnet.` keys, one contiguous run at the head of the arena)"""
            ds16, ds8, ds4, ds2 = state
            dp16 = b_l3(ds16)
            ds8 = ops.add(ds8, ops.maxpool_bwd(arg16, dp16, s8.shape))
            ds4 = ops.add(ds4, b_l2(ds8))
            dp4 = b_l1(ds4)
            if hasattr(b_stem, "from_pool") and os.environ.get("DAHITRA_NO_FUSED_STEM_BWD_UNET", "0") != "1":
                b_stem.from_pool(arg4, dp4, extra=ds2)             # add + max-pool backward + both BatchNorm passes in one
            else:
                b_stem(ops.add(ds2, ops.maxpool_bwd(arg4, dp4, s2.shape)))

        def bwd(dl):
            bwd_second(bwd_first(dl))
        bwd.split = (bwd_first, bwd_second, ("resnet.",))
        return logits, bwd
