"""AdamW with torch.optim.AdamW's update rule (models/trainer.py:39-40: lr, betas (0.9, 0.999),
eps 1e-8, weight_decay 0.01, decoupled decay, bias correction, no amsgrad) as ONE HIP launch over the
net's flat fp32 arena.  Parameters whose .grad is None are skipped exactly as torch does, which for
these nets is the statically known set the forward never touches (resnet.layer4/fc, the *_2 modules).

It is a torch.optim.Optimizer subclass, so lr schedulers (get_scheduler) and state_dict() work.
`capturable=True` keeps lr / step count / bias corrections in device memory so that the step can be
recorded in a HIP graph (dahitra_amd.graph) and replayed."""
import ctypes

import torch

from . import _lib, ops


class AdamW(torch.optim.Optimizer):
    _rule = "torch"      # "xbd": the hand-rolled rule of xBD_code/adamw.py (subclass in models/xbd.py)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.capturable = capturable
        self._flat_state = {}       # id(net) -> [exp_avg, exp_avg_sq, step(int) | step_dev, hyper_dev, hyper_host]

    # ---- flat (arena) state -----------------------------------------------------------------------
    def _state_for(self, net, group):
        param, _ = net.flat_params()
        st = self._flat_state.get(id(net))
        if st is None or st[0].numel() != param.numel() or st[0].device != param.device:
            st = [torch.zeros_like(param), torch.zeros_like(param), 0, None, None]
            if self.capturable:
                st[2] = torch.zeros(1, dtype=torch.int32, device=param.device)
                st[3] = torch.zeros(8, dtype=torch.float32, device=param.device)
            self._flat_state[id(net)] = st
            sd_p = dict(net.named_parameters())
            for k in net._active_keys:      # per-parameter views for state_dict() interchange
                o, n = net._arena.offsets[k]
                self.state[sd_p[k]] = dict(step=torch.tensor(0.0), exp_avg=st[0][o:o + n].view(sd_p[k].shape),
                                           exp_avg_sq=st[1][o:o + n].view(sd_p[k].shape))
        return st

    def sync_hyper(self, grad_scale=1.0):
        """capturable mode: push lr / betas / eps / weight_decay / grad_scale to the device when they changed
        (call OUTSIDE graph capture; a replay then reads the new values)"""
        for group in self.param_groups:
            for st in self._flat_state.values():
                if st[3] is None:
                    continue
                host = (group["lr"], group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"],
                        float(grad_scale))
                if st[4] != host:
                    st[3][:6].copy_(torch.tensor(host, dtype=torch.float32))
                    st[4] = host

    def step_count(self, net):
        st = self._flat_state.get(id(net))
        if st is None:
            return 0
        return int(st[2]) if not torch.is_tensor(st[2]) else int(st[2].item())

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            todo = [p for p in group["params"] if p.grad is not None]
            nets = {}
            loose = []
            for p in todo:
                tag = getattr(p, "_dh_arena", None)
                if tag is not None and p.grad.data_ptr() == tag[0]._grad_views[tag[1]].data_ptr():
                    nets.setdefault(id(tag[0]), (tag[0], []))[1].append(p)
                else:
                    loose.append(p)
            for net, plist in nets.values():
                if len(plist) != len(net._active_keys):
                    loose += plist          # partial coverage: fall back to per-tensor launches
                    continue
                param, grad = net.flat_params()
                st = self._state_for(net, group)
                if self.capturable:
                    if st[4] is None:
                        self.sync_hyper(grad_scale)
                    if self._rule == "xbd":
                        ops._call("dh_adamw_xbd_step_graph", ops.P(param), ops.P(grad), ops.P(st[0]), ops.P(st[1]),
                                  ctypes.c_long(param.numel()), ops.P(st[3]), ops.P(st[2]), ops.P(None), ops.S())
                    else:
                        ops._call("dh_adamw_step_graph", ops.P(param), ops.P(grad), ops.P(st[0]), ops.P(st[1]),
                                  ctypes.c_long(param.numel()), ops.P(st[3]), ops.P(st[2]), ops.S())
                else:
                    st[2] += 1
                    self._launch(param, grad, st[0], st[1], group, st[2], grad_scale)
            for p in loose:
                if not p.is_cuda:
                    raise _lib.HipLibraryError("dahitra_amd.AdamW: parameters must live on the GPU (no CPU fallback)")
                s = self.state[p]
                if "_n" not in s:
                    s["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    s["_n"] = 0
                s["_n"] += 1
                self._launch(p.data, p.grad.contiguous(), s["exp_avg"], s["exp_avg_sq"], group, s["_n"], grad_scale)
        return loss

    def _launch(self, param, grad, m, v, group, step, grad_scale):
        beta1, beta2 = group["betas"]
        if self._rule == "xbd":
            if grad_scale != 1.0:
                raise ValueError("the xBD AdamW takes no host-side grad_scale (clip with clip_grad_norm_)")
            ops.adamw_xbd_step(param, grad, m, v, group["lr"], beta1, beta2, group["eps"], group["weight_decay"], step)
        else:
            ops.adamw_step(param, grad, m, v, group["lr"], beta1, beta2, group["eps"], group["weight_decay"], step,
                           grad_scale)
