"""One train step as ONE HIP graph: ~500 (BiT) to ~2000 (newUNetTrans) kernel launches recorded once and
replayed per step, so the host cost of a step drops from milliseconds of Python/ctypes to a single
hipGraphLaunch (MI355X guidance: capture launch-bound inner loops in hipGraphs, not a tracing compiler).

    step = GraphedTrainStep(net, opt, a, b, lab)      # opt = dahitra_amd.optim.AdamW(..., capturable=True)
    loss = step(a, b, lab)                            # device scalar, same semantics as the eager step

Recorded: forward, zero_grad, focal loss, backward and (single process) the AdamW kernel.  With
torch.distributed the gradient all-reduce and AdamW run eagerly after the replay (two calls), the
1/world factor folded into the optimizer's grad_scale.  The warm-up steps torch needs before capture are
undone (parameters, BN buffers and optimizer state are restored), so the first graphed step is step 1."""
import torch
import torch.distributed as dist

from . import parallel
from .models import losses


class GraphedTrainStep:
    def __init__(self, net, opt, a, b, lab, warmup=3):
        if not getattr(opt, "capturable", False):
            raise ValueError("GraphedTrainStep needs dahitra_amd.optim.AdamW(..., capturable=True)")
        self.net, self.opt = net, opt
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.a, self.b, self.lab = a.clone(), b.clone(), lab.clone()
        net._ensure_arena(a.device)
        # ---- snapshot the training state, warm up eagerly on a side stream, restore -------------------
        flat0 = net._arena.flat.clone()
        bufs0 = [t.clone() for t in net.buffers()]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._eager_body(include_opt=True)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        net._arena.flat.copy_(flat0)
        for t, t0 in zip(net.buffers(), bufs0):
            t.copy_(t0)
        st = opt._flat_state[id(net)]
        st[0].zero_()
        st[1].zero_()
        st[2].zero_()
        # ---- capture ---------------------------------------------------------------------------------
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._eager_body(include_opt=self.world == 1)
        torch.cuda.synchronize()

    def _eager_body(self, include_opt):
        logits = self.net(self.a, self.b)
        self.opt.zero_grad()
        loss = losses.focal_loss(logits, self.lab)
        loss.backward()
        if include_opt:
            self.opt.step()
        return loss.detach()

    def __call__(self, a=None, b=None, lab=None):
        if a is not None:
            self.a.copy_(a, non_blocking=True)
            self.b.copy_(b, non_blocking=True)
            self.lab.copy_(lab, non_blocking=True)
        self.opt.sync_hyper(1.0 / self.world)
        self.graph.replay()
        if self.world > 1:
            parallel.allreduce_net_grads_(self.net)
            self.opt.step()
        return self.loss
