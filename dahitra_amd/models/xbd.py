"""The xBD 5-class damage-assessment step on the HIP pipelines.

Mirrored interfaces (same names, argument meaning, error behaviour):
    BASE_Transformer_UNet(input_nc, output_nc, ...)   xBD_code/zoo/model_transformer_encoding.py:242-449
        net(x) with ONE [B, 6, H, W] tensor (pre | post), logits [B, 5, H, W]; built at xBD_code/train.py:44-45
    ComboLoss(weights, per_image=False)               xBD_code/losses.py:95-126   (dice + focal on the sigmoid)
    xbd_loss(out, msks)                               xBD_code/train.py:348-353   (the five weighted channel losses)
    clip_grad_norm_(parameters, max_norm)             torch.nn.utils.clip_grad_norm_ as called at train.py:373
    AdamW(params, lr, weight_decay)                   xBD_code/adamw.py:6-86      (hand-rolled; eps before bias correction)
Everything computes through libdahitra_hip.so (csrc/xbd_step.hip + the shared model kernels); CPU tensors are refused."""
import torch
import torch.nn as nn

from .. import _lib, ops
from ..optim import AdamW as _ArenaAdamW
from .networks import CDNet

CHANNEL_WEIGHTS = (0.05, 0.2, 0.8, 0.7, 0.4)          # xBD_code/train.py:353
_weights = {}


def BASE_Transformer_UNet(input_nc=3, output_nc=5, with_pos='learned', resnet_stages_num=4, token_len=4, token_trans=True,
                          enc_depth=1, dec_depth=8, dim_head=64, decoder_dim_head=64, tokenizer=True,
                          if_upsample_2x=True, pool_mode='max', pool_size=2, backbone='resnet18', decoder_softmax=True,
                          with_decoder_pos=None, with_decoder=True, compute_dtype=None):
    """Constructor of the xBD copy.  `dec_depth` is accepted and ignored exactly as the reference does (the per-level
    decoder depths 4/4/8/1 are hard-coded, model_transformer_encoding.py:318-334).  Default torch initialisation (the
    xBD scripts never call init_weights)."""
    if (input_nc, output_nc, with_pos, resnet_stages_num, token_len, enc_depth, dim_head, decoder_dim_head, backbone) != \
            (3, 5, 'learned', 4, 4, 1, 64, 64, 'resnet18') or not (tokenizer and token_trans and with_decoder and
                                                                  decoder_softmax):
        raise NotImplementedError("only the configuration of xBD_code/train.py:44-45 is built "
                                  "(input_nc=3, output_nc=5, token_len=4, with_pos='learned', resnet18)")
    if with_decoder_pos not in (None, 'learned'):
        raise NotImplementedError("with_decoder_pos must be None or 'learned'")
    print("using UNet Transformer !!!!")
    return CDNet("xbd_unet_transformer" if with_decoder_pos == 'learned' else "xbd_unet_transformer_nodecpos",
                 compute_dtype)


# ---- loss ------------------------------------------------------------------------------------------------
class _Combo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, masks, weights_dev, dice_w, focal_w):
        loss, channel, sums = ops.combo_loss_fwd(logits, masks, weights_dev, dice_w, focal_w)
        ctx.save_for_backward(logits, masks, sums, weights_dev)
        ctx.w = (dice_w, focal_w)
        ctx.mark_non_differentiable(channel)
        return loss, channel

    @staticmethod
    def backward(ctx, dloss, _dchannel):
        logits, masks, sums, weights_dev = ctx.saved_tensors
        up = dloss.detach().to(torch.float32).reshape(1).contiguous()
        return ops.combo_loss_bwd(logits, masks, sums, weights_dev, up, *ctx.w), None, None, None, None


def _check(outputs, targets):
    if not (outputs.is_cuda and targets.is_cuda):
        raise _lib.HipLibraryError("dahitra_amd xBD loss runs on MI355X only (no CPU fallback)")
    if outputs.shape != targets.shape:
        raise ValueError("logits %s and masks %s must have the same shape" % (tuple(outputs.shape), tuple(targets.shape)))


class ComboLoss(nn.Module):
    """ComboLoss({'dice': a, 'focal': b}): a * soft dice over the whole batch + b * FocalLoss2d(gamma 2), both on
    sigmoid(outputs) (xBD_code/losses.py:95-126).  Other terms of the reference's mapping (bce, jaccard, lovasz, ...)
    are not on the executed path (train.py:316) and raise."""

    def __init__(self, weights, per_image=False):
        super().__init__()
        extra = [k for k, v in weights.items() if v and k not in ("dice", "focal")]
        if extra or per_image:
            raise NotImplementedError("ComboLoss terms %s / per_image are outside the executed xBD step" % extra)
        self.weights = dict(weights)
        self.values = {}

    def forward(self, outputs, targets):
        """outputs / targets: one channel [B, H, W] (the reference's call, train.py:348-352) or [B, C, H, W]"""
        _check(outputs, targets)
        lo = outputs.float().contiguous()
        lo = lo.unsqueeze(1) if lo.dim() == 3 else lo
        ta = targets.float().contiguous().view(lo.shape)
        key = (str(lo.device), (1.0,) * lo.shape[1])
        ones = _weights.get(key)
        if ones is None:
            ones = _weights[key] = torch.ones(lo.shape[1], dtype=torch.float32, device=lo.device)
        loss, _ = _Combo.apply(lo, ta, ones, float(self.weights.get("dice", 0)), float(self.weights.get("focal", 0)))
        return loss


def channel_weights_dev(device, channel_weights=CHANNEL_WEIGHTS):
    """the channel weights as a cached device tensor (no host-to-device copy inside a HIP-graph capture)"""
    key = (str(device), tuple(float(v) for v in channel_weights))
    w = _weights.get(key)
    if w is None:
        w = _weights[key] = torch.tensor(key[1], dtype=torch.float32, device=device)
    return w


def xbd_loss(out, msks, channel_weights=CHANNEL_WEIGHTS, dice=1.0, focal=8.0, want_channels=False):
    """train.py:348-353 in one pass: sum_c w_c * ComboLoss{dice:1, focal:8}(out[:, c], msks[:, c])"""
    _check(out, msks)
    key = (str(out.device), tuple(float(v) for v in channel_weights))
    w = _weights.get(key)
    if w is None:                    # cached: no host-to-device copy inside a HIP-graph capture
        w = _weights[key] = torch.tensor(key[1], dtype=torch.float32, device=out.device)
    loss, channel = _Combo.apply(out.float().contiguous(), msks.float().contiguous(), w, float(dice), float(focal))
    return (loss, channel) if want_channels else loss


# ---- clip + optimizer ------------------------------------------------------------------------------------
def clip_grad_norm_(parameters, max_norm):
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) for the parameters of ONE dahitra_amd net: the total L2
    norm over the flat gradient arena and the in-place scaling by min(1, max_norm / (norm + 1e-6)), without a host
    synchronisation.  Returns the total norm as a device scalar."""
    params = [p for p in parameters if p.grad is not None]
    nets = {id(getattr(p, "_dh_arena", (None,))[0]): getattr(p, "_dh_arena", (None,))[0] for p in params}
    if len(nets) != 1 or None in nets.values():
        raise _lib.HipLibraryError("clip_grad_norm_: parameters must belong to one dahitra_amd net on the GPU")
    net = next(iter(nets.values()))
    if len(params) != len(net._active_keys):
        raise ValueError("clip_grad_norm_: pass all of net.parameters() (the norm is taken over the whole arena)")
    _, grad = net.flat_params()
    out = torch.empty(2, dtype=torch.float32, device=grad.device)
    ops.grad_norm_clip_coef(grad, float(max_norm), out)
    ops.scale_into(grad, out[1:2], grad)
    return out[0]


class AdamW(_ArenaAdamW):
    """xBD_code/adamw.py: m, v as Adam; denom = sqrt(v) + eps; step = lr * sqrt(1 - b2^t) / (1 - b1^t);
    w -= weight_decay * lr * w before the Adam term.  One launch over the net's flat arena."""
    _rule = "xbd"

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, capturable=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=capturable)
