"""CDEvaluator: the evaluation loop of the reference's models/evaluator.py on the HIP pipelines.

Mirrored: __init__(args, dataloader) evaluator.py:25-63, _load_checkpoint :66-86 (FileNotFoundError when the
checkpoint is missing), eval_models(checkpoint_name) :166-182 -> scores dict (acc / mIoU / mF1 / per-class,
misc/metric_tool.py:96-138).  net.eval() folds every BatchNorm into its convolution; arg-max and the confusion
matrix are one device kernel per batch (no per-step device->host copy, cf. evaluator.py:89-104,
metric_tool.py:141-158).  The visualisation jpgs of evaluator.py:118-131 are host plumbing and not written."""
import os

import numpy as np
import torch

from .. import ops
from .networks import define_G


def cm2score(cm):
    """misc/metric_tool.py:96-138"""
    cm = np.asarray(cm, dtype=np.float64)
    tp = np.diag(cm)
    sum_r, sum_c = cm.sum(axis=1), cm.sum(axis=0)
    eps = np.finfo(np.float32).eps
    acc = tp.sum() / (cm.sum() + eps)
    recall, precision = tp / (sum_r + eps), tp / (sum_c + eps)
    f1 = 2 * recall * precision / (recall + precision + eps)
    iou = tp / (sum_r + sum_c - tp + eps)
    out = {'acc': float(acc), 'miou': float(np.nanmean(iou)), 'mf1': float(np.nanmean(f1))}
    for i in range(len(tp)):
        out['iou_%d' % i], out['F1_%d' % i] = float(iou[i]), float(f1[i])
        out['precision_%d' % i], out['recall_%d' % i] = float(precision[i]), float(recall[i])
    return out


class CDEvaluator:
    def __init__(self, args, dataloader):
        self.dataloader = dataloader
        self.n_class = args.n_class
        self.net_G = define_G(args=args, gpu_ids=args.gpu_ids)
        if not (torch.cuda.is_available() and len(args.gpu_ids) > 0):
            raise RuntimeError("dahitra_amd.CDEvaluator needs a GPU id (there is no CPU fallback)")
        self.device = torch.device("cuda:%s" % args.gpu_ids[0])
        self.checkpoint_dir = getattr(args, "checkpoint_dir", ".")
        self.confusion = torch.zeros(self.n_class, self.n_class, dtype=torch.int64, device=self.device)
        self.G_pred = None
        self.batch = None
        self.best_val_acc = 0.0
        self.best_epoch_id = 0

    def _load_checkpoint(self, checkpoint_name='best_ckpt.pt'):
        path = os.path.join(self.checkpoint_dir, checkpoint_name)
        if not os.path.exists(path):
            raise FileNotFoundError('no such checkpoint %s' % checkpoint_name)
        ck = torch.load(path, map_location="cpu")
        sd = ck['model_G_state_dict']
        # nn.DataParallel checkpoints carry a "module." prefix (xBD_code/train.py:450-453)
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.net_G.load_state_dict(sd)
        self.best_val_acc = ck.get('best_val_acc', 0.0)
        self.best_epoch_id = ck.get('best_epoch_id', 0)

    def _forward_pass(self, batch):
        self.batch = batch
        self.G_pred = self.net_G(batch['A'].to(self.device), batch['B'].to(self.device))

    def _collect_running_batch_states(self, want_mask=False):
        gt = self.batch['L'].to(self.device).long().contiguous()
        return ops.confusion_matrix(self.G_pred.detach().float().contiguous(), gt, self.confusion, want_mask)

    def eval_models(self, checkpoint_name='best_ckpt.pt'):
        self._load_checkpoint(checkpoint_name)
        self.confusion.zero_()
        self.net_G.eval()
        for batch in self.dataloader:
            with torch.no_grad():
                self._forward_pass(batch)
            self._collect_running_batch_states()
        return cm2score(self.confusion.cpu().numpy())
