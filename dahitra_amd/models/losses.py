"""Drop-in surface of the reference's models/losses.py for the executed train step
(models/trainer.py:254-262): focal_loss (+ one_hot's 1e-6, losses.py:58-196) with its gradient
produced by the same HIP kernel; cross_entropy (losses.py:9-26) for the batch-size-1 branch.

diceloss(argmax(logits), gt) in the reference contributes no gradient (argmax) and comes from
segmentation_models_pytorch, which is neither vendored nor version-pinned (parity
of that constant is UNPINNED); `diceloss` restates smp's binary DiceLoss on the HIP arg-max mask so that
G_loss carries the same additive constant the reference logs."""
import torch

from .. import ops


class _Focal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, alpha):
        tgt = target
        if tgt.dim() == logits.dim():
            tgt = tgt[:, 0]
        tgt = tgt.to(torch.int64).contiguous()
        loss, dl = ops.focal_loss(logits.detach().float().contiguous(), tgt, want_grad=True, alpha=alpha)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        out = torch.empty_like(dl)      # dl * upstream scalar, read on the device (no host sync)
        ops.scale_into(dl, g, out)
        return out, None, None


def focal_loss(input, target, alpha=0.5, gamma=2.0, reduction='mean', eps=None):
    """models/losses.py:106-196 (gamma fixed at 2, mean reduction: the only form the trainer uses)."""
    if gamma != 2.0 or reduction != 'mean':
        raise NotImplementedError("dahitra_amd.focal_loss: only gamma=2, reduction='mean' (trainer.py:257)")
    if not input.is_cuda:
        raise RuntimeError("dahitra_amd.focal_loss needs a CUDA(HIP) tensor; there is no CPU fallback")
    return _Focal.apply(input, target, float(alpha))


def argmax_mask(logits):
    """torch.argmax(G_pred, dim=1) (models/trainer.py:170, evaluator.py:101) on the HIP kernel"""
    return ops.argmax_nchw(logits.detach().float().contiguous())


def _target3(logits, target):
    tgt = target[:, 0] if target.dim() == logits.dim() else target
    return tgt.to(torch.int64).contiguous()


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        lg = logits.detach().float().contiguous()
        out = ops.cross_entropy_fwd(lg, target, ignore_index)
        ctx.save_for_backward(lg, target, out)
        ctx.ignore_index = ignore_index
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        lg, target, out = ctx.saved_tensors
        up = g.detach().float().reshape(1).contiguous()
        return ops.cross_entropy_bwd(lg, target, out, up, ctx.ignore_index), None, None


def cross_entropy(input, target, weight=None, reduction='mean', ignore_index=255):
    """models/losses.py:9-26: F.cross_entropy with the hard-coded class weights [1, 1] (so `weight` is ignored, as in
    the reference), ignore_index 255, mean reduction.  The reference's bilinear resize for mismatched sizes is not on
    the trainer's path (logits and labels share H, W) and raises here."""
    if not input.is_cuda:
        raise RuntimeError("dahitra_amd.cross_entropy needs a CUDA(HIP) tensor; there is no CPU fallback")
    if reduction != 'mean':
        raise NotImplementedError("dahitra_amd.cross_entropy: reduction='mean' only (trainer.py:261)")
    tgt = _target3(input, target)
    if tuple(input.shape[-2:]) != tuple(tgt.shape[-2:]):
        raise NotImplementedError("cross_entropy: logits %s vs target %s -- the resize branch (losses.py:21-22) is "
                                  "outside the trainer's path" % (tuple(input.shape), tuple(tgt.shape)))
    return _CrossEntropy.apply(input, tgt, int(ignore_index))


def diceloss(input, target, weight=None):
    """models/losses.py:333-339: DiceLoss(mode='binary') of segmentation_models_pytorch on argmax(input) -- a constant
    with respect to the parameters (argmax).  Returned as a gradient-free device scalar."""
    if not input.is_cuda:
        raise RuntimeError("dahitra_amd.diceloss needs a CUDA(HIP) tensor; there is no CPU fallback")
    return ops.dice_argmax_constant(input.detach().float().contiguous(), _target3(input, target))
