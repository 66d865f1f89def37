"""CDTrainer: the training loop of the reference's models/trainer.py on the HIP pipelines.

Mirrored (same names, argument meaning, order of operations):
    __init__(args, dataloaders)      trainer.py:21-103   net_G = define_G(args, gpu_ids), AdamW, scheduler, Logger / Timer,
                                                         running ConfuseMatrixMeter, accuracy curves, checkpoint / vis dirs
    _load_checkpoint / _save_checkpoint   :106-158       resume net + optimizer + scheduler + epoch counters
    _forward_pass(batch)             :247-252            batch {'A','B','L'} -> self.G_pred
    _backward_G()                    :254-262            B != 1: diceloss(argmax) + focal_loss; B == 1: cross_entropy
    _update_metric / _collect_running_batch_states / _collect_epoch_states   :163-214
    _update_checkpoints / _update_*_acc_curve / _update_lr_schedulers        :216-245
    train_models()                   :288-334            load, per epoch: train batches, scheduler step, val batches, ckpt
Differences, all on the host side of the step:
  * THE STEP IS A HIP GRAPH.  With static batch shapes (every batch but a ragged last one) `train_step` replays ONE recorded
    graph -- forward, focal loss, backward, AdamW and the confusion-matrix count (dahitra_amd.graph.GraphedTrainStep): one
    hipGraphLaunch instead of ~500 ctypes launches (3.8 ms of host time per step).  A batch of another shape, a batch of one
    (the reference's cross-entropy branch) and `args.hip_graph = False` / DAHITRA_NO_GRAPH=1 take the eager step, which runs
    the same kernels and gives bit-identical parameters.
  * the confusion matrix is accumulated ON THE DEVICE across the epoch and read once per epoch (and at the reference's log
    lines, every 2000th batch): no logits / mask / matrix copy per step (trainer.py:163-173 copies logits and mask);
  * `args.compute_dtype` ('fp32' parity mode | 'bf16' throughput mode), `args.gpu_loader` (utils.get_loaders: the
    device-resident input pipeline) have no counterpart in the reference;
  * under torchrun (WORLD_SIZE > 1) the trainer joins the process group (dahitra_amd.parallel): replicas start from rank
    0's parameters, gradients are averaged per step (inside the graphed step, or after the eager backward); the epoch's
    confusion matrix is summed over the ranks (all ranks see the same epoch_acc / best_val_acc), and only rank 0 writes
    log.txt, the accuracy curves and the checkpoints (the others wait at a barrier for the file);
  * `args.checkpoint_dir` / `args.vis_dir` / `args.loss` are optional (no log files without a checkpoint_dir); the
    visualisation grid of trainer.py:194-203 is not assembled (its imsave is commented out there)."""
import os

import numpy as np
import torch

from .. import ops, parallel
from ..misc.logger_tool import Logger, Timer
from ..misc.metric_tool import ConfuseMatrixMeter
from ..optim import AdamW
from . import losses
from .networks import define_G, get_scheduler


class _NullLogger:
    def write(self, message):
        pass

    def write_dict_str(self, d):
        pass


class CDTrainer:
    def __init__(self, args, dataloaders):
        self.dataloaders = dataloaders
        self.n_class = args.n_class
        self.rank, local, self.world = parallel.init_from_env()      # torchrun: one process per GPU (no-op otherwise)
        if self.world > 1:
            args.gpu_ids = [local]                                   # each rank trains on the GPU of its LOCAL_RANK
        self.net_G = define_G(args=args, gpu_ids=args.gpu_ids)
        if not (torch.cuda.is_available() and len(args.gpu_ids) > 0):
            raise RuntimeError("dahitra_amd.CDTrainer needs a GPU id (the reference's CPU mode, gpu_ids=-1, "
                               "has no counterpart: there is no CPU fallback)")
        self.device = torch.device("cuda:%s" % args.gpu_ids[0])
        print(self.device)
        if self.world > 1:                                           # replicas start from rank 0's parameters and buffers
            self.net_G._ensure_arena(self.device)
            parallel.broadcast_params_(self.net_G)
        self.use_graph = bool(getattr(args, "hip_graph", True)) and os.environ.get("DAHITRA_NO_GRAPH", "0") != "1"
        self.lr = args.lr
        # capturable: lr / step count / bias corrections live on the device, so that the step can be recorded and replayed
        # (also without the graph: the eager step then runs the very same update kernel, bit for bit)
        self.optimizer_G = AdamW(self.net_G.parameters(), lr=self.lr, betas=(0.9, 0.999), weight_decay=0.01, capturable=True)
        self._graph, self._graph_key = None, None
        self._val_graph, self._val_graph_key = None, None      # the validation forward + confusion count (GraphedEvalStep)
        self.exp_lr_scheduler_G = get_scheduler(self.optimizer_G, args)
        self.running_metric = ConfuseMatrixMeter(n_class=self.n_class)
        self.checkpoint_dir = getattr(args, "checkpoint_dir", None)
        self.vis_dir = getattr(args, "vis_dir", None)
        for d in (self.checkpoint_dir, self.vis_dir):
            if d and not os.path.exists(d):
                os.makedirs(d, exist_ok=True)
        # one process per GPU: only rank 0 owns the files of the run (log.txt, *_acc.npy, best_ckpt.pt) -- N ranks appending to
        # one log and racing torch.save() on one path leave torn files behind; the other ranks log nowhere
        self.is_main = self.rank == 0
        if self.checkpoint_dir and self.is_main:
            self.logger = Logger(os.path.join(self.checkpoint_dir, 'log.txt'))
            self.logger.write_dict_str(args.__dict__)
        else:
            self.logger = _NullLogger()
        self.timer = Timer()
        self.batch_size = args.batch_size
        self.epoch_acc = 0
        self.best_val_acc = 0.0
        self.best_epoch_id = 0
        self.epoch_to_start = 0
        self.max_num_epochs = args.max_epochs
        self.global_step = 0
        self.steps_per_epoch = len(dataloaders['train']) if dataloaders and hasattr(dataloaders.get('train'), '__len__') else 1
        self.total_steps = (self.max_num_epochs - self.epoch_to_start) * self.steps_per_epoch
        self.G_pred = None
        self.G_final_pred = None
        self.pred_vis = None
        self.batch = None
        self.G_loss = None
        self.is_training = False
        self.batch_id = 0
        self.epoch_id = 0
        loss_name = getattr(args, "loss", "ce")
        if loss_name == 'ce':
            self._pxl_loss = losses.cross_entropy
        elif loss_name == 'focal':
            self._pxl_loss = losses.focal_loss
        else:       # ce_multi / ce_dice exist in the reference's table but _backward_G never reads self._pxl_loss
            raise NotImplementedError(loss_name)
        self.VAL_ACC = self._load_curve('val_acc.npy')
        self.TRAIN_ACC = self._load_curve('train_acc.npy')
        self.confusion = torch.zeros(self.n_class, self.n_class, dtype=torch.int64, device=self.device)
        self._synced = np.zeros((self.n_class, self.n_class), np.int64)     # part of `confusion` the host meter has seen
        self._focal = self._dice_args = None
        self._counted = False

    def _load_curve(self, name):
        path = os.path.join(self.checkpoint_dir, name) if self.checkpoint_dir else None
        return np.load(path) if path and os.path.exists(path) else np.array([], np.float32)

    # ---- checkpoints (trainer.py:106-134, 150-158) ---------------------------------------------------
    def _load_checkpoint(self, ckpt_name='best_ckpt.pt'):
        """resume: net, optimizer (flat Adam moments + step count) and scheduler states, epoch counters"""
        path = os.path.join(self.checkpoint_dir, ckpt_name) if self.checkpoint_dir else None
        if not path or not os.path.exists(path):
            print('training from scratch...')
            return False
        self.logger.write('loading last checkpoint...\n')
        checkpoint = torch.load(path, map_location=self.device, weights_only=False)
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
        self.total_steps = (self.max_num_epochs - self.epoch_to_start) * self.steps_per_epoch
        self.logger.write('Epoch_to_start = %d, Historical_best_acc = %.4f (at epoch %d)\n\n' %
                          (self.epoch_to_start, self.best_val_acc, self.best_epoch_id))
        return True

    def _save_checkpoint(self, ckpt_name):
        """rank 0 writes (to a temporary name, then an atomic rename); every rank waits for the file before going on, so a
        resume on any rank never sees a half-written checkpoint"""
        if self.is_main:
            self._write_checkpoint(ckpt_name)
        parallel.barrier()

    def _write_checkpoint(self, ckpt_name):
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        path = os.path.join(self.checkpoint_dir, ckpt_name)
        # (plain python numbers: the file then also loads under torch.load's weights_only default; the reference's own
        # checkpoints carry numpy scalars, which the loaders here accept with weights_only=False)
        torch.save({'epoch_id': int(self.epoch_id), 'best_val_acc': float(self.best_val_acc),
                    'best_epoch_id': int(self.best_epoch_id), 'model_G_state_dict': self.net_G.state_dict(),
                    'optimizer_G_state_dict': self.optimizer_G.state_dict(),
                    'exp_lr_scheduler_G_state_dict': self.exp_lr_scheduler_G.state_dict()}, path + '.tmp')
        os.replace(path + '.tmp', path)

    # ---- bookkeeping ----------------------------------------------------------------------------------
    def _timer_update(self):
        self.global_step = (self.epoch_id - self.epoch_to_start) * self.steps_per_epoch + self.batch_id
        self.timer.update_progress((self.global_step + 1) / max(self.total_steps, 1))
        est = self.timer.estimated_remaining()
        imps = (self.global_step + 1) * self.batch_size / max(self.timer.get_stage_elapsed(), 1e-9)
        return imps, est

    def _visualize_pred(self):
        return losses.argmax_mask(self.G_final_pred).unsqueeze(1) * 255

    def _update_lr_schedulers(self):
        self.exp_lr_scheduler_G.step()

    def _count_batch(self):
        """arg-max + confusion counts of this batch INTO the device-side running matrix (one kernel, no host traffic);
        the graphed step has done it inside the graph"""
        if not self._counted:
            gt = self.batch['L'].to(self.device).long().contiguous()
            ops.confusion_matrix(self.G_final_pred.detach().float().contiguous(), gt, self.confusion)
            self._counted = True

    def _sync_metric(self, all_ranks=False):
        """hand what the device has counted since the last call to the reference's meter (ONE read-back of n_class^2 numbers);
        returns the mean F1 of that increment (for a per-batch call: the batch's F1, the reference's `running_mf1`).
        all_ranks (the epoch's last call): the counts of every rank's shard are summed first, so that all ranks hold the SAME
        epoch scores and take the same best-model decision"""
        cm = self.confusion
        if all_ranks and parallel.exchange_enabled():
            cm = cm.clone()
            parallel.allreduce_counts_(cm)
        cm = cm.cpu().numpy().copy()            # (a copy also when `confusion` is host memory: _synced must not alias it)
        inc, self._synced = cm - self._synced, cm
        return self.running_metric.update_from_matrix(inc) if inc.sum() > 0 else 0.0

    def _update_metric(self):
        """trainer.py:163-173 (kept for callers of the reference's name): count this batch and sync the host meter now"""
        self._count_batch()
        return self._sync_metric()

    @property
    def G_loss(self):
        """dice(argmax) + focal as the reference logs it (trainer.py:259).  The dice term carries no gradient; it is evaluated
        only when somebody reads G_loss (the log line of every 2000th batch), not once per step"""
        if self._focal is None:
            return None
        if self._dice_args is not None:
            pred, gt = self._dice_args
            self._focal, self._dice_args = losses.diceloss(pred, gt) + self._focal.detach(), None
        return self._focal

    @G_loss.setter
    def G_loss(self, value):
        self._focal, self._dice_args = value, None

    def _collect_running_batch_states(self):
        self._count_batch()
        loader = self.dataloaders['train'] if self.is_training else self.dataloaders['val']
        m = len(loader) if hasattr(loader, '__len__') else -1
        imps, est = self._timer_update()
        if np.mod(self.batch_id, 2000) == 1:
            running_acc = self._sync_metric()
            loss = self.G_loss
            self.logger.write('Is_training: %s. [%d,%d][%d,%d], imps: %.2f, est: %.2fh, G_loss: %.5f, running_mf1: %.5f\n' %
                              (self.is_training, self.epoch_id, self.max_num_epochs - 1, self.batch_id, m,
                               imps * self.batch_size, est, float(loss.detach()) if loss is not None else float('nan'),
                               running_acc))

    def _collect_epoch_states(self):
        self._sync_metric(all_ranks=True)         # the one device read of the epoch (+ one 4-number all-reduce under torchrun)
        scores = self.running_metric.get_scores()
        self.epoch_acc = scores['mf1']
        self.logger.write('Is_training: %s. Epoch %d / %d, epoch_mF1= %.5f\n' %
                          (self.is_training, self.epoch_id, self.max_num_epochs - 1, self.epoch_acc))
        self.logger.write(''.join('%s: %.5f ' % (k, v) for k, v in scores.items()) + '\n\n')
        return scores

    def _update_checkpoints(self):
        self.logger.write('Lastest model updated. Epoch_acc=%.4f, Historical_best_acc=%.4f (at epoch %d)\n\n'
                          % (self.epoch_acc, self.best_val_acc, self.best_epoch_id))
        if self.epoch_acc > self.best_val_acc:              # (last_ckpt.pt is commented out in the reference, trainer.py:219)
            self.best_val_acc = self.epoch_acc
            self.best_epoch_id = self.epoch_id
            if self.checkpoint_dir:
                self._save_checkpoint(ckpt_name='best_ckpt.pt')
            self.logger.write('*' * 10 + 'Best model updated!\n\n')

    def _update_training_acc_curve(self):
        self.TRAIN_ACC = np.append(self.TRAIN_ACC, [self.epoch_acc])
        if self.checkpoint_dir and self.is_main:
            np.save(os.path.join(self.checkpoint_dir, 'train_acc.npy'), self.TRAIN_ACC)

    def _update_val_acc_curve(self):
        self.VAL_ACC = np.append(self.VAL_ACC, [self.epoch_acc])
        if self.checkpoint_dir and self.is_main:
            np.save(os.path.join(self.checkpoint_dir, 'val_acc.npy'), self.VAL_ACC)

    def _clear_cache(self):
        self.running_metric.clear()
        self.confusion.zero_()
        self._synced = np.zeros_like(self._synced)

    def scores(self):
        """acc / mIoU / mF1 / per-class figures of the device-side running confusion matrix (misc/metric_tool.py:96-138)"""
        from ..misc.metric_tool import cm2score
        return {k: float(v) for k, v in cm2score(self.confusion.cpu().numpy().astype(np.float64)).items()}

    # ---- the hot step --------------------------------------------------------------------------------
    def _forward_pass(self, batch):
        self.batch = batch
        img_in1 = batch['A'].to(self.device)
        img_in2 = batch['B'].to(self.device)
        self.G_pred = self.net_G(img_in1, img_in2)
        self.G_final_pred = self.G_pred
        self._counted = False

    def _backward_G(self):
        gt = self.batch['L'].to(self.device).long()
        self._pxl_loss1 = losses.diceloss
        self._pxl_loss2 = losses.focal_loss
        if gt.shape[0] != 1:
            # the dice term (trainer.py:259) is a gradient-free constant (argmax): it shifts the logged value only
            focal = self._pxl_loss2(self.G_pred, gt)
            focal.backward()
            self._focal, self._dice_args = focal.detach(), (self.G_pred.detach(), gt)
        else:
            self.G_loss = losses.cross_entropy(self.G_pred, gt)
            self.G_loss.backward()

    def _eager_step(self, batch):
        self._forward_pass(batch)
        self.optimizer_G.zero_grad()
        self._backward_G()
        scale = parallel.allreduce_net_grads_(self.net_G)           # 1.0 without a process group
        self.optimizer_G.step(grad_scale=scale)
        # trainer.py:308 clips AFTER the step; gradients are zeroed before the next use => no effect

    def train_step(self, batch):
        """one step of trainer.py:302-308; returns G_loss (dice constant + focal, or the cross entropy of a batch of one)"""
        self._step(batch)
        return self.G_loss

    def _step(self, batch):
        """the step without materialising G_loss.  Static shapes -> replay of the recorded HIP graph; anything else -> eager."""
        a, b, lab = batch['A'], batch['B'], batch['L']
        graphable = self.use_graph and self.net_G.training and a.shape[0] != 1
        if not graphable:
            return self._eager_step(batch)
        a = a.to(self.device, non_blocking=True)
        b = b.to(self.device, non_blocking=True)
        lab = lab.to(self.device, non_blocking=True)
        key = (tuple(a.shape), a.dtype, tuple(lab.shape), lab.dtype)
        if self._graph is None:
            from ..graph import GraphedTrainStep
            self._graph = GraphedTrainStep(self.net_G, self.optimizer_G, a.float(), b.float(), lab, confusion=self.confusion)
            self._graph_key = key
        if key != self._graph_key:                # the ragged last batch of an epoch
            return self._eager_step({'A': a, 'B': b, 'L': lab})
        self.batch = batch
        loss = self._graph(a, b, lab)
        self.G_pred = self.G_final_pred = self._graph.logits
        self._counted = True                      # counted inside the graph
        self._focal, self._dice_args = loss, (self.G_pred, self._graph.lab)

    def _val_graphed(self, batch):
        """the validation forward + confusion count of a batch of the recorded shape as one hipGraphLaunch
        (dahitra_amd.graph.GraphedEvalStep); False: this batch takes the eager forward (another shape, graphs off)"""
        if not self.use_graph:
            return False
        a, b = batch['A'].to(self.device).float(), batch['B'].to(self.device).float()
        lab = batch['L'].to(self.device).long().contiguous()
        key = (tuple(a.shape), tuple(lab.shape))
        if self._val_graph is None:
            from ..graph import GraphedEvalStep
            self._val_graph = GraphedEvalStep(self.net_G, a, b, lab, confusion=self.confusion)
            self._val_graph_key = key
        if key != self._val_graph_key:
            return False
        self.batch = batch
        self.G_pred = self.G_final_pred = self._val_graph(a, b, lab)
        self._counted = True                      # counted inside the graph
        return True

    def train_models(self):
        self._load_checkpoint()
        for self.epoch_id in range(self.epoch_to_start, self.max_num_epochs):
            # ---- train ----
            self._clear_cache()
            self.is_training = True
            self.net_G.train()
            self.logger.write('lr: %0.7f\n' % self.optimizer_G.param_groups[0]['lr'])
            sampler = getattr(self.dataloaders['train'], 'sampler', None)
            if hasattr(sampler, 'set_epoch'):           # DistributedSampler (utils.get_loaders under torchrun)
                sampler.set_epoch(self.epoch_id)
            for self.batch_id, batch in enumerate(self.dataloaders['train'], 0):
                self._step(batch)
                self._collect_running_batch_states()
            self._collect_epoch_states()
            self._update_training_acc_curve()
            self._update_lr_schedulers()
            # ---- eval ----
            self.logger.write('Begin evaluation...\n')
            self._clear_cache()
            self.is_training = False
            self.net_G.eval()
            # under torchrun the replicas' BatchNorm running statistics differ (per-replica BN, as nn.DataParallel) while the
            # validation split is sharded over the ranks and best_ckpt.pt stores rank 0's buffers: every rank scores rank
            # 0's model, so that the logged epoch score IS the score of the checkpointed model
            parallel.broadcast_buffers_(self.net_G)
            for self.batch_id, batch in enumerate(self.dataloaders['val'], 0):
                if not self._val_graphed(batch):
                    with torch.no_grad():
                        self._forward_pass(batch)
                self._collect_running_batch_states()
            self._collect_epoch_states()
            # ---- checkpoints ----
            self._update_val_acc_curve()
            self._update_checkpoints()
