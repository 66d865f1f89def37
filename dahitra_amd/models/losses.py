"""Drop-in surface of the reference's models/losses.py for the executed train step
(models/trainer.py:254-262): focal_loss (+ one_hot's 1e-6, losses.py:58-196) with its gradient
produced by the same HIP kernel, cross_entropy for the B == 1 branch is not on the measured path.

diceloss(argmax(logits), gt) in the reference contributes no gradient (argmax) and comes from
segmentation_models_pytorch, which is neither vendored nor version-pinned;
`dice_constant` reproduces smp's binary DiceLoss on the HIP arg-max mask for logging only."""
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
