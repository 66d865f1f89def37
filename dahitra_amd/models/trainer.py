"""CDTrainer: the train step of the reference's models/trainer.py on the HIP pipelines.

Mirrored (same names, argument meaning, order of operations):
    __init__(args, dataloaders)      trainer.py:21-103  net_G = define_G(args, gpu_ids), AdamW, scheduler
    _forward_pass(batch)             trainer.py:247-252 batch {'A','B','L'} -> self.G_pred
    _backward_G()                    trainer.py:254-262 B != 1: diceloss(argmax) + focal_loss; B == 1: cross_entropy
    train_models()                   trainer.py:288-334 forward, zero_grad, backward, step (then the no-op clip)
Out of scope here (host plumbing): Logger/Timer files, visualisation jpgs,
numpy accuracy curves.  The running confusion matrix is kept on the device instead of the per-step
device->host copy of trainer.py:163-173."""
import os

import torch

from .. import ops
from ..optim import AdamW
from . import losses
from .networks import define_G, get_scheduler


class CDTrainer:
    def __init__(self, args, dataloaders):
        self.dataloaders = dataloaders
        self.n_class = args.n_class
        self.net_G = define_G(args=args, gpu_ids=args.gpu_ids)
        if not (torch.cuda.is_available() and len(args.gpu_ids) > 0):
            raise RuntimeError("dahitra_amd.CDTrainer needs a GPU id (the reference's CPU mode, gpu_ids=-1, "
                               "has no counterpart: there is no CPU fallback)")
        self.device = torch.device("cuda:%s" % args.gpu_ids[0])
        self.lr = args.lr
        self.optimizer_G = AdamW(self.net_G.parameters(), lr=self.lr, betas=(0.9, 0.999), weight_decay=0.01)
        self.exp_lr_scheduler_G = get_scheduler(self.optimizer_G, args)
        self.batch_size = args.batch_size
        self.max_num_epochs = args.max_epochs
        self.epoch_to_start = 0
        self.checkpoint_dir = getattr(args, "checkpoint_dir", None)
        self.G_pred = None
        self.G_loss = None
        self.batch = None
        self.is_training = False
        self.best_val_acc = 0.0
        self.best_epoch_id = 0
        self.epoch_id = 0
        self.confusion = torch.zeros(self.n_class, self.n_class, dtype=torch.int64, device=self.device)

    # ---- the hot step --------------------------------------------------------------------------------
    def _forward_pass(self, batch):
        self.batch = batch
        img_in1 = batch['A'].to(self.device)
        img_in2 = batch['B'].to(self.device)
        self.G_pred = self.net_G(img_in1, img_in2)
        self.G_final_pred = self.G_pred

    def _backward_G(self):
        gt = self.batch['L'].to(self.device).long()
        self._pxl_loss1 = losses.diceloss
        self._pxl_loss2 = losses.focal_loss
        if gt.shape[0] != 1:
            # the dice term (trainer.py:259) is a gradient-free constant (argmax): it shifts the logged value only
            self.G_loss = self._pxl_loss1(self.G_pred, gt) + self._pxl_loss2(self.G_pred, gt)
        else:
            self.G_loss = losses.cross_entropy(self.G_pred, gt)
        self.G_loss.backward()

    def train_step(self, batch):
        self._forward_pass(batch)
        self.optimizer_G.zero_grad()
        self._backward_G()
        self.optimizer_G.step()
        # trainer.py:308 clips AFTER the step; gradients are zeroed before the next use => no effect
        return self.G_loss

    def _update_metric(self):
        gt = self.batch['L'].to(self.device).long().contiguous()
        ops.confusion_matrix(self.G_pred.detach().float().contiguous(), gt, self.confusion)   # arg-max + counts, one kernel

    def scores(self):
        """acc / mIoU / mF1 from the confusion matrix (misc/metric_tool.py:96-138)"""
        import numpy as np
        cm = self.confusion.cpu().numpy().astype(np.float64)
        tp = np.diag(cm)
        sum_r, sum_c = cm.sum(1), cm.sum(0)
        eps = np.finfo(np.float32).eps
        acc = tp.sum() / (cm.sum() + eps)
        recall, precision = tp / (sum_r + eps), tp / (sum_c + eps)
        f1 = 2 * recall * precision / (recall + precision + eps)
        iou = tp / (sum_r + sum_c - tp + eps)
        return dict(acc=float(acc), miou=float(np.nanmean(iou)), mf1=float(np.nanmean(f1)))

    # ---- checkpoints (trainer.py:106-134, 150-158) ---------------------------------------------------
    def _load_checkpoint(self, ckpt_name='best_ckpt.pt'):
        """resume: net, optimizer (flat Adam moments + step count) and scheduler states, epoch counters"""
        path = os.path.join(self.checkpoint_dir, ckpt_name) if self.checkpoint_dir else None
        if not path or not os.path.exists(path):
            print('training from scratch...')
            return False
        checkpoint = torch.load(path, map_location=self.device)
        sd = checkpoint['model_G_state_dict']
        if all(k.startswith('module.') for k in sd):           # written under nn.DataParallel
            sd = {k[len('module.'):]: v for k, v in sd.items()}
        self.net_G.load_state_dict(sd)
        self.net_G.to(self.device)
        self.optimizer_G.load_state_dict(checkpoint['optimizer_G_state_dict'])
        self.exp_lr_scheduler_G.load_state_dict(checkpoint['exp_lr_scheduler_G_state_dict'])
        self.epoch_to_start = checkpoint['epoch_id'] + 1
        self.best_val_acc = checkpoint['best_val_acc']
        self.best_epoch_id = checkpoint['best_epoch_id']
        print('Epoch_to_start = %d, Historical_best_acc = %.4f (at epoch %d)' %
              (self.epoch_to_start, self.best_val_acc, self.best_epoch_id))
        return True

    def _save_checkpoint(self, ckpt_name):
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        torch.save({'epoch_id': self.epoch_id, 'best_val_acc': self.best_val_acc,
                    'best_epoch_id': self.best_epoch_id, 'model_G_state_dict': self.net_G.state_dict(),
                    'optimizer_G_state_dict': self.optimizer_G.state_dict(),
                    'exp_lr_scheduler_G_state_dict': self.exp_lr_scheduler_G.state_dict()},
                   os.path.join(self.checkpoint_dir, ckpt_name))

    def train_models(self):
        self._load_checkpoint()
        for self.epoch_id in range(self.epoch_to_start, self.max_num_epochs):
            self.confusion.zero_()
            self.is_training = True
            self.net_G.train()
            for batch in self.dataloaders['train']:
                self.train_step(batch)
                self._update_metric()
            train_scores = self.scores()
            self.exp_lr_scheduler_G.step()
            self.confusion.zero_()
            self.is_training = False
            self.net_G.eval()
            for batch in self.dataloaders['val']:
                with torch.no_grad():
                    self._forward_pass(batch)
                self._update_metric()
            val = self.scores()
            print("epoch %d train mF1 %.5f val mF1 %.5f loss %.6f" % (self.epoch_id, train_scores["mf1"], val["mf1"],
                                                                    float(self.G_loss)))
            if val["mf1"] > self.best_val_acc and self.checkpoint_dir:      # trainer.py:216-232 (last_ckpt is commented
                self.best_val_acc, self.best_epoch_id = val["mf1"], self.epoch_id     # out in the reference)
                self._save_checkpoint('best_ckpt.pt')
