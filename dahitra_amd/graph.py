"""One train step as ONE HIP graph: ~500 (BiT) to ~2000 (newUNetTrans) kernel launches recorded once and
replayed per step, so the host cost of a step drops from milliseconds of Python/ctypes to a single
hipGraphLaunch (MI355X guidance: capture launch-bound inner loops in hipGraphs, not a tracing compiler).

    step = GraphedTrainStep(net, opt, a, b, lab)      # opt = dahitra_amd.optim.AdamW(..., capturable=True)
    loss = step(a, b, lab)                            # device scalar, same semantics as the eager step

Recorded: forward, zero_grad, focal loss, backward (at engine level, no autograd graph) and (single process) the AdamW
kernel.  With
torch.distributed the step is TWO graphs around the exchange:
    graph 1: forward, loss and the first part of the backward -- BiT nets: down to and including resnet.layer3, so that
             every gradient from layer3 to the end of the flat arena (~77 % of its bytes for base_transformer_pos_s4) is
             final; newUNetTrans / the xBD model: head, top-down path and the three transformer levels (the 5 MB behind the
             trunk in the arena);
    all-reduce of that arena tail, ASYNC (RCCL's stream) ..........  } concurrently
    graph 2: the rest of the backward (BiT: layer2 / layer1 / stem;  }
             newUNetTrans / xBD: the whole ResNet trunk)             }
    all-reduce of the arena head, wait for both, then the update: AdamW with 1/world folded into its grad_scale, or for the
    xBD step the mean over ranks, clip_grad_norm_ over the complete arena and the hand-rolled AdamW (train.py:373-374).
DAHITRA_OVERLAP=auto (default) takes this form only where the all-reduce it hides is modelled longer than the form costs
(parallel.split_offset: world size and tail bytes; 1 forces it, 0 / DAHITRA_NO_OVERLAP=1 gives one graph, then one all-reduce and
the update).  The warm-up steps torch needs before capture are
undone (parameters, BN buffers and optimizer state are restored), so the first graphed step is step 1."""
import os

import torch
import torch.distributed as dist

from . import ops, parallel
from .models import losses


class GraphedTrainStep:
    CHECK_EVERY = 64          # replays between two read-backs of the persistent-BatchNorm error words

    def __init__(self, net, opt, a, b, lab, warmup=3, confusion=None):
        """confusion: an int64 [n_class, n_class] device tensor; the recorded step then also counts arg-max(logits) against
        the labels into it (dh_confusion_matrix, one more kernel inside the graph: the running metric of the reference's
        trainer without any per-step launch or host read)"""
        if not getattr(opt, "capturable", False):
            raise ValueError("GraphedTrainStep needs dahitra_amd.optim.AdamW(..., capturable=True)")
        self.net, self.opt = net, opt
        self.confusion = confusion
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.exchange = parallel.exchange_enabled()      # gradient all-reduce + AdamW after the replay
        self.split_off = None                            # arena offset where the overlapped (two-graph) form splits
        self.persist_bn_launches = None                  # (graph 1, graph 2) counts of the overlapped form
        self._set_inputs(a, b, lab)
        net._ensure_arena(a.device)
        # ---- snapshot the training state, warm up eagerly on a side stream, restore -------------------
        flat0 = net._arena.flat.clone()
        bufs0 = [t.clone() for t in net.buffers()]
        conf0 = confusion.clone() if confusion is not None else None
        opt0 = opt.snapshot_flat_state(net)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        split = self.exchange and self._can_split()
        with torch.cuda.stream(s):
            for _ in range(warmup):
                if split:      # the same launches the two captures make (the split backward has its own reduce tables)
                    self._split_first()
                    self._second()
                    self._update()
                else:
                    self._body(include_opt=True)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        net._arena.flat.copy_(flat0)
        for t, t0 in zip(net.buffers(), bufs0):
            t.copy_(t0)
        opt.restore_flat_state(net, opt0)     # the optimizer keeps what it carried (e.g. a resumed checkpoint)
        if confusion is not None:
            confusion.copy_(conf0)
        # ---- capture ---------------------------------------------------------------------------------
        # A captured graph holds RAW pointers: the shared scratch workspace, the weight-gradient plan's slabs and
        # job table, the packed-weight buffers.  ops.pin_captured_buffers() makes every buffer the capture touched
        # immutable for the life of the process (a later, larger eager call allocates a NEW buffer instead of
        # freeing the one the graph still reads and writes).
        # The capture runs on a stream of our own that inherits the warm-up stream's scratch workspace (ops.workspace is keyed
        # by stream): sized by the warm-up, allocated outside the graph's private pool, and no entry of a dead stream remains.
        # (kept for the life of the step: released, PyTorch's stream pool could hand its handle to a later torch.cuda.Stream(),
        # whose eager launches would then look up -- and scribble over -- the workspace the recorded graph reads and writes)
        cs = self._capture_stream = torch.cuda.Stream()
        ops.rekey_workspace(a.device, s, cs)
        self.graph = torch.cuda.CUDAGraph()
        if split:
            self.graph2 = torch.cuda.CUDAGraph()
            n0 = ops.BN_PERSIST_LAUNCHES
            with torch.cuda.graph(self.graph, stream=cs):
                self.loss = self._split_first()
            n1 = ops.BN_PERSIST_LAUNCHES
            with torch.cuda.graph(self.graph2, pool=self.graph.pool(), stream=cs):
                self._second()
            # persistent BatchNorm launches recorded into (graph 1, graph 2): graph 2 replays beside RCCL's kernels and must hold none
            self.persist_bn_launches = (n1 - n0, ops.BN_PERSIST_LAUNCHES - n1)
            if self.persist_bn_launches[1] and os.environ.get("DAHITRA_OVERLAP_PERSIST_BN") != "1":
                raise RuntimeError("dahitra_amd: %d persistent BatchNorm launches were recorded into the graph that overlaps the "
                                   "gradient all-reduce" % self.persist_bn_launches[1])
        else:
            self.split_off = None
            with torch.cuda.graph(self.graph, stream=cs):
                self.loss = self._body(include_opt=not self.exchange)
        self._pinned = ops.pin_captured_buffers(net)
        self._generation = net._arena.generation
        self._calls = 0
        torch.cuda.synchronize()
        ops.bn_persist_check(a.device)         # the warm-up steps and the capture left no barrier timeout behind

    def _set_inputs(self, a, b, lab):
        self.a, self.b, self.lab = a.clone(), b.clone(), lab.clone()

    def _copy_inputs(self, a, b, lab):
        self.a.copy_(a, non_blocking=True)
        self.b.copy_(b, non_blocking=True)
        self.lab.copy_(lab, non_blocking=True)

    def _eager_body(self, include_opt):
        logits = self.net(self.a, self.b)
        self.logits = logits.detach()          # static output buffer of the graph: valid after every replay
        self.opt.zero_grad()
        loss = losses.focal_loss(logits, self.lab)
        loss.backward()
        self._count(self.logits)
        if include_opt:
            self.opt.step()
        return loss.detach()

    def _body(self, include_opt):
        """the one-graph step.  The focal-loss step runs at engine level (no autograd graph): the same kernels as
        `_eager_body` minus what autograd adds around them -- the fill of the upstream gradient 1.0 and the pass that scales
        dloss/dlogits by it (two launches, 34 MB per step at batch 32).  Subclasses with their own `_eager_body` keep it."""
        if type(self)._eager_body is not GraphedTrainStep._eager_body or os.environ.get("DAHITRA_GRAPH_AUTOGRAD") == "1":
            return self._eager_body(include_opt)           # (DAHITRA_GRAPH_AUTOGRAD=1: A/B switch back to the autograd step)
        net = self.net
        logits = self._forward_split()
        bwd = net._engine.take_backward()
        self.logits = logits
        loss, dl = self._loss_and_grad(logits)
        self._count(logits)
        net._arena.grad.zero_()
        net._engine.backward(dl, bwd)
        net._bind_grad_views()           # the optimizer skips parameters without a .grad, as torch does
        if include_opt:
            self.opt.step()
        return loss

    def _forward_split(self):
        return self.net._run_forward(self.a, self.b, need_grad=True)

    def _loss_and_grad(self, logits):
        """(loss, dloss/dlogits) at engine level (no autograd graph): focal loss of the BiT / DAHiTra trainers"""
        tgt = self.lab[:, 0] if self.lab.dim() == logits.dim() else self.lab
        return ops.focal_loss(logits, tgt.to(torch.int64).contiguous(), want_grad=True)

    def _count(self, logits):
        if self.confusion is not None:
            tgt = self.lab[:, 0] if self.lab.dim() == logits.dim() else self.lab
            ops.confusion_matrix(logits, tgt.to(torch.int64).contiguous(), self.confusion)

    # ---- overlapped form ------------------------------------------------------------------------------
    def _can_split(self):
        """the net's backward has a split point and every gradient the second graph writes (bit: stem, layer1, layer2; unet /
        xbd: the whole trunk) lies below the first offset of what the first graph completes: the arena tail [split, end) is
        final after the first graph.  Keys below the split that the first graph writes (positional embeddings are registered
        first) just ride with the second all-reduce."""
        self.split_off = parallel.split_offset(self.net, self.world)
        return self.split_off is not None

    def _split_first(self):
        """forward, loss and the first part of the backward, called under capture (engine level: one autograd-free pass)"""
        net = self.net
        logits = self._forward_split()
        bwd = net._engine.take_backward()
        self.logits = logits
        loss, dl = self._loss_and_grad(logits)
        self._count(logits)
        net._arena.grad.zero_()
        net._engine.backward_first(dl, bwd)
        net._bind_grad_views()           # the optimizer skips parameters without a .grad, as torch does
        return loss

    def _second(self):
        """graph 2 (layer2 / layer1 / stem backward) replays WHILE the all-reduce of the arena tail runs on RCCL's stream: its
        BatchNorm backward must not be the persistent one-launch form, whose device-wide barrier needs every CU"""
        if os.environ.get("DAHITRA_OVERLAP_PERSIST_BN") == "1":      # EXPERIMENT (one rank only): what the two-pass BatchNorm of graph 2 costs
            self.net._engine.backward_second()
            return
        with ops.no_persist_bn():
            self.net._engine.backward_second()

    def _after_replay(self):
        """world > 1: the exchange step and the update run eagerly after the replayed forward/backward"""
        parallel.allreduce_net_grads_(self.net)
        self._update()

    def _replay_overlapped(self):
        _, grad = self.net.flat_params()
        self.graph.replay()
        w1 = dist.all_reduce(grad[self.split_off:], op=dist.ReduceOp.SUM, async_op=True)      # waits for graph 1 only
        self.graph2.replay()                                                                  # ... while this runs
        w2 = dist.all_reduce(grad[:self.split_off], op=dist.ReduceOp.SUM, async_op=True)
        w1.wait()
        w2.wait()
        self._update()

    def _update(self):
        """after both all-reduces: the (eager) parameter update"""
        self.opt.step()

    def __call__(self, *inputs):
        if self.net._arena.generation != self._generation:
            raise RuntimeError("dahitra_amd: the net's parameter arena was rebuilt (moved to another device / "
                               "parameters replaced) after this step was captured; build a new GraphedTrainStep")
        if inputs and inputs[0] is not None:
            self._copy_inputs(*inputs)
        self._calls += 1
        if self._calls in (2, 8) or self._calls % self.CHECK_EVERY == 0:
            # the persistent BatchNorm backward's device-wide barrier: a timeout (bit 0) or a non-finite sum (bit 1) of any
            # replay since the last check raises here (one 4-byte read-back per sync block: a host sync, hence not per step)
            ops.bn_persist_check(self.a.device)
        self.opt.sync_hyper(1.0 / self.world)
        if self.split_off is not None:
            self._replay_overlapped()
            return self.loss
        self.graph.replay()
        if self.exchange:
            self._after_replay()
        return self.loss


class GraphedEvalStep:
    """The evaluation forward (models/evaluator.py:156-164: net.eval(), no gradient state) as ONE HIP graph: the eval-mode weight
    re-pack (every BatchNorm folded into its convolution), the forward and -- with `confusion` -- the arg-max + confusion-matrix
    count against the labels (one kernel, no host read).  ~130 (BiT) / ~200 (DAHiTra) launches per batch become one
    hipGraphLaunch: the eager evaluation forward of DAHiTra proper is bound by its launches from Python.

        step = GraphedEvalStep(net, a, b, lab, confusion)      # net.eval() is called here
        logits = step(a, b, lab)                                # static shapes; the caller runs other shapes eagerly

    The parameters are read when the graph replays, so a checkpoint loaded into `net` afterwards is what the next replay scores."""

    def __init__(self, net, a, b, lab=None, confusion=None, warmup=2):
        self.net, self.confusion = net, confusion
        net.eval()
        self.a, self.b = a.clone(), b.clone()
        self.lab = lab.clone() if lab is not None else None
        if confusion is not None and lab is None:
            raise ValueError("GraphedEvalStep: a confusion matrix needs the labels")
        net._ensure_arena(a.device)
        conf0 = confusion.clone() if confusion is not None else None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if confusion is not None:
            confusion.copy_(conf0)
        cs = self._capture_stream = torch.cuda.Stream()       # (kept: see GraphedTrainStep)
        ops.rekey_workspace(a.device, s, cs)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=cs):
            self.logits = self._body()
        self._pinned = ops.pin_captured_buffers(net)
        self._generation = net._arena.generation
        torch.cuda.synchronize()

    def _body(self):
        with torch.no_grad():
            logits = self.net(self.a, self.b)
        if self.confusion is not None:
            tgt = self.lab[:, 0] if self.lab.dim() == logits.dim() else self.lab
            ops.confusion_matrix(logits.detach().float().contiguous(), tgt.to(torch.int64).contiguous(), self.confusion)
        return logits

    def __call__(self, a=None, b=None, lab=None):
        if self.net._arena.generation != self._generation:
            raise RuntimeError("dahitra_amd: the net's parameter arena was rebuilt after this evaluation step was captured; "
                               "build a new GraphedEvalStep")
        if self.net.training:
            raise RuntimeError("dahitra_amd: GraphedEvalStep replays the eval-mode forward; call net.eval() (a train-mode forward "
                               "in between re-packs the weights for training)")
        if a is not None:
            self.a.copy_(a, non_blocking=True)
            self.b.copy_(b, non_blocking=True)
            if lab is not None and self.lab is not None:
                self.lab.copy_(lab, non_blocking=True)
        self.graph.replay()
        return self.logits


class GraphedXbdStep(GraphedTrainStep):
    """The xBD step (xBD_code/train.py:331-374) as one HIP graph: forward of the 6-channel model, the five weighted
    ComboLoss terms, backward, clip_grad_norm_(0.999) and the hand-rolled AdamW.

        step = GraphedXbdStep(net, xbd.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-6, capturable=True), imgs, msks)
        loss = step(imgs, msks)"""

    def __init__(self, net, opt, imgs, msks, max_norm=0.999, warmup=3):
        self.max_norm = max_norm
        super().__init__(net, opt, imgs, None, msks, warmup=warmup)

    def _set_inputs(self, imgs, _unused, msks):
        self.a, self.lab = imgs.clone(), msks.clone()

    def _copy_inputs(self, imgs, msks):
        self.a.copy_(imgs, non_blocking=True)
        self.lab.copy_(msks, non_blocking=True)

    def _eager_body(self, include_opt):
        from .models import xbd
        self.net.zero_grad()
        logits = self.net(self.a)
        self.logits = logits.detach()
        loss = xbd.xbd_loss(logits, self.lab)
        loss.backward()
        if include_opt:
            xbd.clip_grad_norm_(self.net.parameters(), self.max_norm)
            self.opt.step()
        return loss.detach()

    def _forward_split(self):
        x = self.a
        return self.net._run_forward(x[:, :3], x[:, 3:], need_grad=True)

    def _loss_and_grad(self, logits):
        """the five weighted ComboLoss terms and their gradient at engine level (xBD_code/train.py:348-353)"""
        from .models import xbd
        w = xbd.channel_weights_dev(logits.device)
        lo, ms = logits.float().contiguous(), self.lab.float().contiguous()
        loss, _, sums = ops.combo_loss_fwd(lo, ms, w, 1.0, 8.0)
        one = self._one if getattr(self, "_one", None) is not None else torch.ones(1, dtype=torch.float32, device=lo.device)
        self._one = one
        return loss, ops.combo_loss_bwd(lo, ms, sums, w, one, 1.0, 8.0)

    def _after_replay(self):
        parallel.allreduce_net_grads_(self.net)
        self._update()

    def _update(self):
        """mean over the ranks, clip_grad_norm_(0.999) over the WHOLE (now complete) arena, then the hand-rolled AdamW:
        the clip needs the total norm, so it cannot start before both all-reduces have landed (train.py:373-374)"""
        from .models import xbd
        _, grad = self.net.flat_params()
        if self.world > 1:
            if getattr(self, "_inv_world", None) is None:
                self._inv_world = torch.tensor([1.0 / self.world], device=grad.device)
            ops.scale_into(grad, self._inv_world, grad)
        xbd.clip_grad_norm_(self.net.parameters(), self.max_norm)
        self.opt.step()
