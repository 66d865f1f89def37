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
from ..misc import metric_tool
from ..misc.logger_tool import Logger
from .networks import define_G


def cm2score(cm):
    """misc/metric_tool.py:96-138 on a confusion matrix (plain floats)"""
    return {k: float(v) for k, v in metric_tool.cm2score(np.asarray(cm, dtype=np.float64)).items()}


class CDEvaluator:
    def __init__(self, args, dataloader):
        self.dataloader = dataloader
        self.n_class = args.n_class
        self.net_G = define_G(args=args, gpu_ids=args.gpu_ids)
        if not (torch.cuda.is_available() and len(args.gpu_ids) > 0):
            raise RuntimeError("dahitra_amd.CDEvaluator needs a GPU id (there is no CPU fallback)")
        self.device = torch.device("cuda:%s" % args.gpu_ids[0])
        self.checkpoint_dir = getattr(args, "checkpoint_dir", ".")
        self.vis_dir = getattr(args, "vis_dir", None)
        for d in (self.checkpoint_dir, self.vis_dir):
            if d and not os.path.exists(d):
                os.makedirs(d, exist_ok=True)
        self.model_str = args.net_G
        self.logger = None
        if getattr(args, "checkpoint_dir", None):          # evaluator.py:37-40: log_test.txt next to the checkpoint
            self.logger = Logger(os.path.join(self.checkpoint_dir, 'log_test.txt'))
            self.logger.write_dict_str(args.__dict__)
        self.confusion = torch.zeros(self.n_class, self.n_class, dtype=torch.int64, device=self.device)
        self.G_pred = None
        self.batch = None
        self.epoch_acc = 0
        self.best_val_acc = 0.0
        self.best_epoch_id = 0
        self.steps_per_epoch = len(dataloader) if hasattr(dataloader, '__len__') else 0
        self.is_training = False
        self.batch_id = 0
        self.epoch_id = 0
        # the evaluation forward + the confusion count as one recorded HIP graph per batch shape (dahitra_amd.graph.GraphedEvalStep);
        # args.hip_graph = False / DAHITRA_NO_GRAPH=1, or a batch of another shape (the ragged last one): the eager forward
        self.use_graph = bool(getattr(args, "hip_graph", True)) and os.environ.get("DAHITRA_NO_GRAPH", "0") != "1"
        self._graph, self._graph_key = None, None

    def _log(self, message):
        if self.logger is not None:
            self.logger.write(message)

    def _load_checkpoint(self, checkpoint_name='best_ckpt.pt'):
        path = os.path.join(self.checkpoint_dir, checkpoint_name)
        if not os.path.exists(path):
            raise FileNotFoundError('no such checkpoint %s' % checkpoint_name)
        ck = torch.load(path, map_location="cpu", weights_only=False)
        sd = ck['model_G_state_dict']
        # nn.DataParallel checkpoints carry a "module." prefix (xBD_code/train.py:450-453)
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.net_G.load_state_dict(sd)
        self.best_val_acc = ck.get('best_val_acc', 0.0)
        self.best_epoch_id = ck.get('best_epoch_id', 0)
        self._log('Eval Historical_best_acc = %.4f (at epoch %d)\n\n' % (self.best_val_acc, self.best_epoch_id))

    def _visualize_pred(self):
        from .losses import argmax_mask
        return argmax_mask(self.G_pred).unsqueeze(1) * 255

    def _forward_pass(self, batch):
        self.batch = batch
        self.G_pred = self.net_G(batch['A'].to(self.device), batch['B'].to(self.device))

    def _collect_running_batch_states(self, want_mask=False):
        gt = self.batch['L'].to(self.device).long().contiguous()
        return ops.confusion_matrix(self.G_pred.detach().float().contiguous(), gt, self.confusion, want_mask)

    def _collect_epoch_states(self):
        scores = cm2score(self.confusion.cpu().numpy())
        self.epoch_acc = scores['mf1']
        if self.logger is not None:                              # evaluator.py:138-139: an empty "<mF1>.txt" marker
            with open(os.path.join(self.checkpoint_dir, '%s.txt' % self.epoch_acc), mode='a'):
                pass
        self._log(''.join('%s: %.5f ' % (k, v) for k, v in scores.items()) + '\n\n')
        return scores

    def eval_models(self, checkpoint_name='best_ckpt.pt'):
        self._load_checkpoint(checkpoint_name)
        self._log('Begin evaluation...\n')
        self.confusion.zero_()
        self.is_training = False
        self.net_G.eval()
        for self.batch_id, batch in enumerate(self.dataloader, 0):
            if not self._graphed_batch(batch):
                with torch.no_grad():
                    self._forward_pass(batch)
                self._collect_running_batch_states()
        return self._collect_epoch_states()

    def _graphed_batch(self, batch):
        """forward + confusion count of `batch` through the recorded graph; False when this batch takes the eager path"""
        if not self.use_graph:
            return False
        a, b, lab = batch['A'].to(self.device), batch['B'].to(self.device), batch['L'].to(self.device).long().contiguous()
        key = (tuple(a.shape), a.dtype, tuple(lab.shape))
        if self._graph is None:
            from ..graph import GraphedEvalStep
            self._graph = GraphedEvalStep(self.net_G, a.float(), b.float(), lab, confusion=self.confusion)
            self._graph_key = key
        if key != self._graph_key:
            return False
        self.batch = batch
        self.G_pred = self._graph(a.float(), b.float(), lab)
        return True
