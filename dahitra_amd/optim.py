"""AdamW with torch.optim.AdamW's update rule (models/trainer.py:39-40: lr, betas (0.9, 0.999),
eps 1e-8, weight_decay 0.01, decoupled decay, bias correction, no amsgrad) as ONE HIP launch over the
net's flat fp32 arena.  Parameters whose .grad is None are skipped exactly as torch does, which for
these nets is the statically known set the forward never touches (resnet.layer4/fc, the *_2 modules).

It is a torch.optim.Optimizer subclass, so lr schedulers (get_scheduler) work, and state_dict() /
load_state_dict() interchange with torch.optim.AdamW's per-parameter {step, exp_avg, exp_avg_sq}
(the reference resumes with optimizer_G.load_state_dict, models/trainer.py:116): the flat moment
arenas and the step counter are filled from a loaded state and written back on save.
`capturable=True` keeps lr / step count / bias corrections in device memory so that the step can be
recorded in a HIP graph (dahitra_amd.graph) and replayed."""
import ctypes

import torch

from . import _lib, ops


class _Flat:
    """Adam state of one net: flat moment arenas + the step counter (host int, or a device int32 when capturable)"""
    __slots__ = ("net", "group", "m", "v", "step", "hyper", "host")

    def __init__(self, net, group, param, capturable):
        self.net, self.group = net, group
        self.m, self.v = torch.zeros_like(param), torch.zeros_like(param)
        self.step = torch.zeros(1, dtype=torch.int32, device=param.device) if capturable else 0
        self.hyper = torch.zeros(8, dtype=torch.float32, device=param.device) if capturable else None
        self.host = None          # the values last pushed into `hyper`

    def count(self):
        return int(self.step.item()) if torch.is_tensor(self.step) else int(self.step)

    def set_count(self, n):
        if torch.is_tensor(self.step):
            self.step.fill_(int(n))
        else:
            self.step = int(n)


class AdamW(torch.optim.Optimizer):
    _rule = "torch"      # "xbd": the hand-rolled rule of xBD_code/adamw.py (subclass in models/xbd.py)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.capturable = capturable
        self._flat_state = {}       # id(net) -> _Flat

    # ---- flat (arena) state -----------------------------------------------------------------------
    def _state_for(self, net, group):
        param, _ = net.flat_params()
        st = self._flat_state.get(id(net))
        if st is None or st.m.numel() != param.numel() or st.m.device != param.device:
            st = _Flat(net, group, param, self.capturable)
            self._flat_state[id(net)] = st
            self._adopt_param_state(st)
        return st

    def _adopt_param_state(self, st):
        """per-parameter state (loaded by load_state_dict, or absent) -> the flat arenas; afterwards self.state[p]
        holds VIEWS of the arenas, so state_dict() sees the live moments"""
        net = st.net
        sd_p = dict(net.named_parameters())
        steps = set()
        for k in net._active_keys:
            p = sd_p[k]
            o, n = net._arena.offsets[k]
            mv, vv = st.m[o:o + n].view(p.shape), st.v[o:o + n].view(p.shape)
            old = self.state.get(p)
            if old and "exp_avg" in old and old["exp_avg"].data_ptr() != mv.data_ptr():
                mv.copy_(old["exp_avg"].to(mv.device, torch.float32))
                vv.copy_(old["exp_avg_sq"].to(vv.device, torch.float32))
                steps.add(int(float(old.get("step", 0))))
            self.state[p] = dict(step=torch.tensor(0.0), exp_avg=mv, exp_avg_sq=vv)
        if len(steps) > 1:
            raise ValueError("dahitra_amd.AdamW: a loaded state with different step counts per parameter (%s) cannot "
                             "be represented by the arena's single counter" % sorted(steps))
        if steps:
            st.set_count(steps.pop())

    def state_dict(self):
        """torch.optim.AdamW layout; `step` of every parameter is the arena's real step count"""
        for st in self._flat_state.values():
            n = float(st.count())
            sd_p = dict(st.net.named_parameters())
            for k in st.net._active_keys:
                s = self.state.get(sd_p[k])
                if s is not None:
                    s["step"] = torch.tensor(n)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)        # fills self.state[p] with fresh tensors (not arena views)
        for st in list(self._flat_state.values()):
            st.m.zero_()
            st.v.zero_()
            st.set_count(0)
            self._adopt_param_state(st)
            st.host = None                         # hyper-parameters may have changed with the param_groups

    def _group_of(self, net):
        for group in self.param_groups:
            if id(net) in self._nets_of(group):
                return group
        return None

    def snapshot_flat_state(self, net):
        """(m, v, step count) of `net`, or None when this optimizer does not hold its parameters.  A state that exists
        only per parameter so far -- load_state_dict() on a fresh optimizer, before any step -- is adopted into the flat
        arenas first, so that a snapshot / restore round trip (dahitra_amd.graph's warm-up) keeps a resumed checkpoint"""
        st = self._flat_state.get(id(net))
        if st is None:
            group = self._group_of(net)
            if group is None:
                return None
            st = self._state_for(net, group)
        return st.m.clone(), st.v.clone(), st.count()

    def restore_flat_state(self, net, snap):
        st = self._flat_state.get(id(net))
        if st is None:
            return
        if snap is None:
            st.m.zero_()
            st.v.zero_()
            st.set_count(0)
        else:
            st.m.copy_(snap[0])
            st.v.copy_(snap[1])
            st.set_count(snap[2])

    def _nets_of(self, group):
        nets = {}
        for p in group["params"]:
            tag = getattr(p, "_dh_arena", None)
            if tag is not None:
                nets[id(tag[0])] = tag[0]
        return nets

    def sync_hyper(self, grad_scale=None):
        """capturable mode: push each group's lr / betas / eps / weight_decay (and grad_scale; None keeps the value
        pushed last, 1.0 initially) to the device state of that group's nets when they changed.  Call OUTSIDE graph
        capture; a replay then reads the new values."""
        for group in self.param_groups:
            for nid in self._nets_of(group):
                st = self._flat_state.get(nid)
                if st is None or st.hyper is None:
                    continue
                gs = float(grad_scale) if grad_scale is not None else (st.host[5] if st.host is not None else 1.0)
                host = (group["lr"], group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"], gs)
                if st.host != host:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("dahitra_amd.AdamW: hyper-parameters / grad_scale changed inside a graph "
                                           "capture; call sync_hyper() before capturing")
                    st.hyper[:6].copy_(torch.tensor(host, dtype=torch.float32))
                    st.host = host

    def step_count(self, net):
        st = self._flat_state.get(id(net))
        return 0 if st is None else st.count()

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """grad_scale: factor applied to the gradient inside the kernel (1/world of the data-parallel mean).  In
        capturable mode None keeps the device value (sync_hyper)."""
        loss = closure() if closure is not None else None
        for group in self.param_groups:
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
                    # ALWAYS compare the group's lr / betas / eps / weight_decay with what the device holds (a host tuple
                    # compare): an lr scheduler changes group['lr'] between steps, and the eager step has to see it exactly
                    # as the graphed step does (GraphedTrainStep syncs before every replay).  Inside a capture nothing may
                    # change; sync_hyper raises there if it did.
                    self.sync_hyper(grad_scale)
                    if self._rule == "xbd":
                        ops._call("dh_adamw_xbd_step_graph", ops.P(param), ops.P(grad), ops.P(st.m), ops.P(st.v),
                                  ctypes.c_long(param.numel()), ops.P(st.hyper), ops.P(st.step), ops.P(None), ops.S())
                    else:
                        ops._call("dh_adamw_step_graph", ops.P(param), ops.P(grad), ops.P(st.m), ops.P(st.v),
                                  ctypes.c_long(param.numel()), ops.P(st.hyper), ops.P(st.step), ops.S())
                else:
                    st.step += 1
                    self._launch(param, grad, st.m, st.v, group, st.step, 1.0 if grad_scale is None else grad_scale)
            for p in loose:
                if not p.is_cuda:
                    raise _lib.HipLibraryError("dahitra_amd.AdamW: parameters must live on the GPU (no CPU fallback)")
                s = self.state[p]
                if "_n" not in s:
                    if "exp_avg" not in s:
                        s["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                        s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    s["_n"] = int(float(s.get("step", 0)))
                s["_n"] += 1
                s["step"] = torch.tensor(float(s["_n"]))
                self._launch(p.data, p.grad.contiguous(), s["exp_avg"], s["exp_avg_sq"], group, s["_n"],
                             1.0 if grad_scale is None else grad_scale)
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
