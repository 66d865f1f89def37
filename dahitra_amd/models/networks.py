"""Drop-in surface of the reference's models/networks.py for the change-detection hot path.

    define_G(args, init_type='normal', init_gain=0.02, gpu_ids=[])   models/networks.py:130-168
    init_net / init_weights                                          models/networks.py:77-127
    get_scheduler(optimizer, args)                                   models/networks.py:22-49
    net(x1, x2) -> logits [B, n_class, H, W] (fp32, NCHW)            models/networks.py:358-392, 1321-1357

The returned nn.Module owns parameters / buffers under the reference's state-dict names (so
checkpoints interchange, models/trainer.py:150-158) but holds no torch operators: forward and
backward run the HIP pipelines of dahitra_amd.engine.  Parameters live as views into one flat fp32
arena per net (grad-carrying parameters first) so that the optimizer step and the data-parallel
gradient all-reduce are single launches over contiguous memory.

Compute type: args.compute_dtype / DAHITRA_DTYPE in {"fp32", "bf16x3", "bf16"}.  fp32 is the parity mode
(exact-fp32 MFMA); bf16x3 is the same fp32 pipeline with every matrix product on the 16-bit matrix cores as
three split products (dh_set_f32_mma_mode: fp32 tensors; fp16 planes in the forward, bf16 planes in the
backward -- the fast parity mode, held to the exact mode's test bounds); bf16 is the throughput mode (bf16 activations + MFMA, fp32 accumulation / master weights)."""
import math
import os

import torch
import torch.nn as nn
from torch.optim import lr_scheduler

from .. import _lib
from ..engine import Engine
from ..netspec import get_config, is_active, is_alias, is_buffer, state_spec

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
           "bf16x3": torch.float32}


def get_scheduler(optimizer, args):
    """linear | step | multistep, as models/networks.py:22-49 (returns, not raises, on unknown)."""
    if args.lr_policy == 'linear':
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: 1.0 - epoch / float(args.max_epochs + 1))
    if args.lr_policy == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=args.max_epochs // 3, gamma=0.1)
    if args.lr_policy == 'multistep':
        return lr_scheduler.MultiStepLR(
            optimizer, milestones=[2, 4, 7, 11, 15, 25, 35, 47, 60, 70, 90, 110, 130, 150, 170, 180, 190], gamma=0.5)
    return NotImplementedError('learning rate policy [%s] is not implemented', args.lr_policy)


class _Node(nn.Module):
    """name-space holder so that state_dict() keys equal the reference's dotted names"""


class _Arena:
    """flat fp32 storage: [active params | inactive params], matching grads for the active part"""

    def __init__(self):
        self.flat = None
        self.grad = None
        self.n_active = 0
        self.offsets = {}
        self.generation = 0


class _NetFunction(torch.autograd.Function):
    """autograd boundary: one node for the whole net.  Parameter gradients are deposited by the HIP
    backward directly into the arena (p.grad views); the images need no gradient."""

    @staticmethod
    def forward(ctx, anchor, net, x1, x2):
        ctx.net = net
        out = net._run_forward(x1, x2)
        ctx.bwd = net._engine.take_backward()      # this forward's saved activations + backward kernels
        return out

    @staticmethod
    def backward(ctx, dlogits):
        bwd, ctx.bwd = ctx.bwd, None
        if bwd is None:
            raise RuntimeError("dahitra_amd: this forward's backward already ran (no retain_graph: the saved "
                               "activations are released after one pass)")
        ctx.net._run_backward(dlogits.contiguous(), bwd)
        return None, None, None, None


class CDNet(nn.Module):
    """BASE_Transformer / BASE_Transformer_UNet replacement selected by net_G."""

    def __init__(self, net_G, compute_dtype=None, attn_dtype=None):
        super().__init__()
        self.net_G = net_G
        self.cfg = get_config(net_G)
        name = compute_dtype or os.environ.get("DAHITRA_DTYPE", "fp32")
        if name not in _DTYPES:
            raise ValueError("compute dtype must be fp32, bf16x3 or bf16, got %r" % name)
        self.compute_dtype = _DTYPES[name]
        # fp32 tensors, split-bf16 matrix products (DAHITRA_F32_MMA=bf16x3: every fp32 net of the process, an A/B switch)
        self.mma_x3 = name == "bf16x3" or (self.compute_dtype == torch.float32 and os.environ.get("DAHITRA_F32_MMA", "") == "bf16x3")
        self._spec = state_spec(net_G)
        for key, shape, role in self._spec:
            node = self
            parts = key.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            if is_alias(role):
                # the reference registers the same sub-module twice (nn.ModuleList holders of the xBD model):
                # one Parameter object under two names -- state_dict() lists both, parameters() once
                node.register_parameter(parts[-1], self.get_parameter(role[6:]))
            elif role == "bn_nbt":
                node.register_buffer(parts[-1], torch.tensor(0, dtype=torch.long))
            elif is_buffer(role):
                node.register_buffer(parts[-1], torch.zeros(shape) if role == "bn_rm" else torch.ones(shape))
            else:
                if role in ("bn_w", "ln_w"):
                    t = torch.ones(shape)
                elif role in ("bn_b", "ln_b", "bias"):
                    t = torch.zeros(shape)
                elif role == "pos":
                    t = torch.randn(shape)                       # nn.Parameter(torch.randn(...)), networks.py:294
                else:
                    fan_in = int(math.prod(shape[1:]))
                    t = torch.empty(shape).uniform_(-1.0, 1.0).mul_(fan_in ** -0.5)
                node.register_parameter(parts[-1], nn.Parameter(t))
        if attn_dtype not in (None, "bf16", "fp8"):
            raise ValueError("attn_dtype must be 'bf16' or 'fp8', got %r" % attn_dtype)
        if attn_dtype == "fp8" and self.compute_dtype != torch.bfloat16:
            raise ValueError("attn_dtype='fp8' (fp8 MFMA operands in the decoder layers) needs compute_dtype='bf16'")
        self._engine = Engine(net_G, self.compute_dtype, use_tr=os.environ.get("DAHITRA_NO_TR", "0") != "1",
                              attn_fp8=attn_dtype == "fp8", mma_x3=self.mma_x3)
        self._arena = _Arena()
        self._anchor = None
        self.tokens_ = None      # attributes the reference stashes on the module (networks.py:373-374)
        self.tokens = None

    # ---- arena ---------------------------------------------------------------------------------------
    def _named_state(self):
        sd_p = dict(self.named_parameters())
        sd_b = dict(self.named_buffers())
        return sd_p, sd_b

    def _ensure_arena(self, device):
        ar = self._arena
        sd_p, sd_b = self._named_state()
        if ar.flat is not None and ar.flat.device == device and all(
                sd_p[k].data_ptr() == ar.flat.data_ptr() + 4 * off for k, (off, _) in ar.offsets.items()):
            return      # every parameter is still the arena view handed out earlier
        own = [k for k, _, r in self._spec if not is_buffer(r) and not is_alias(r)]
        active = [k for k in own if is_active(self.net_G, k)]
        inactive = [k for k in own if not is_active(self.net_G, k)]
        # every parameter starts on a 16-byte boundary (4 floats): kernels read the fp32 masters with 16-byte vector loads
        # straight from the arena (encoder_fused.hip ksplit_rows, the tiled weight re-pack), and e.g. classifier.3.bias (2
        # floats) would otherwise leave everything behind it 8-byte aligned.  Padding floats are zero, carry zero gradients and
        # stay zero under AdamW; the per-layer sizes of the transformer stacks are multiples of 32, so their constant layer
        # pitch is unchanged.
        pad4 = lambda n: (n + 3) & ~3
        total = sum(pad4(sd_p[k].numel()) for k in active + inactive)
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        off = 0
        ar.offsets = {}
        for k in active + inactive:
            p = sd_p[k]
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            p.grad = None
            ar.offsets[k] = (off, n)
            off += pad4(n)
            if k == active[-1]:
                ar.n_active = off
        ar.flat = flat
        ar.grad = torch.zeros(ar.n_active, dtype=torch.float32, device=device)
        ar.generation += 1
        for b in sd_b.values():
            if b.device != device:
                b.data = b.data.to(device)
        params = {k: sd_p[k].data for k in sd_p}
        params.update({k: sd_b[k] for k in sd_b})
        grads = {k: ar.grad[o:o + n].view(sd_p[k].shape) for k, (o, n) in ar.offsets.items() if o < ar.n_active}
        self._engine.bind(params, grads)
        self._grad_views = grads
        self._active_keys = active
        for k in active:
            sd_p[k]._dh_arena = (self, k)
        self._anchor = torch.zeros((), device=device, requires_grad=True)

    def flat_params(self):
        """(param, grad) flat fp32 views over the grad-carrying parameters (optimizer / all-reduce)"""
        return self._arena.flat[:self._arena.n_active], self._arena.grad

    # ---- forward / backward -------------------------------------------------------------------------
    def forward(self, x1, x2=None):
        if self.cfg["kind"] == "xbd":
            # ONE 6-channel tensor, split in the forward (xBD_code/zoo/model_transformer_encoding.py:409-412)
            if x2 is not None or x1.dim() != 4 or x1.shape[1] != 6:
                raise ValueError("the xBD model takes one [B, 6, H, W] tensor (pre | post image)")
            if self.cfg["decoder_pos"] and tuple(x1.shape[2:]) != (1024, 1024):
                raise RuntimeError("pos_embedding_decoder_3 [1,32,64,64] is added to the 1/16-scale map "
                                   "(model_transformer_encoding.py:372-383): input must be 1024x1024, got %dx%d"
                                   % tuple(x1.shape[2:]))
            x1, x2 = x1[:, :3], x1[:, 3:]
        elif x2 is None:
            raise TypeError("forward() missing the second image")
        if not x1.is_cuda:
            raise _lib.HipLibraryError("dahitra_amd runs on MI355X only: inputs must be CUDA(HIP) tensors; "
                                       "there is no CPU fallback")
        _lib.lib()
        self._ensure_arena(x1.device)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if need_grad:
            if not self.training:
                raise NotImplementedError("dahitra_amd: backward through eval-mode BatchNorm is not supported "
                                          "(the reference trains in train mode, models/trainer.py:297)")
            return _NetFunction.apply(self._anchor, self, x1, x2)
        with torch.no_grad():
            return self._run_forward(x1, x2, need_grad=False)

    def _run_forward(self, x1, x2, need_grad=True):
        x1 = x1.detach().float().contiguous()
        x2 = x2.detach().float().contiguous()
        return self._engine.forward(x1, x2, self.training, need_grad)

    def _run_backward(self, dlogits, bwd=None):
        ar = self._arena
        sd_p = dict(self.named_parameters())
        fresh = all(sd_p[k].grad is None for k in self._active_keys)
        if fresh:
            ar.grad.zero_()
        self._engine.backward(dlogits, bwd)
        self._bind_grad_views(sd_p)

    def _bind_grad_views(self, sd_p=None):
        """every active parameter's .grad = its view of the flat gradient arena (host bookkeeping, no launch)"""
        sd_p = sd_p if sd_p is not None else dict(self.named_parameters())
        for k in self._active_keys:
            p = sd_p[k]
            if p.grad is None:
                p.grad = self._grad_views[k]
            elif p.grad.data_ptr() != self._grad_views[k].data_ptr():
                raise RuntimeError("dahitra_amd: parameter %s has a foreign .grad tensor; use zero_grad(set_to_none=True)" % k)


def load_pretrained_trunk(net, path_or_state):
    """Fill the ResNet trunk from a torchvision-style state dict (`conv1.weight`, `layer1.0.bn1.running_mean`, ...), e.g. a
    saved `torchvision.models.resnet18(pretrained=True).state_dict()` -- what the reference gets from the ImageNet download
    in models/resnet.py:228-244 (BiT nets) and resnet18(pretrained=True) in xBD_code/zoo/model_transformer_encoding.py:195.
    No download happens here: pass a file path or the dict.  Keys the net does not have (`fc.*`, `layer4.*` for the BiT nets)
    and keys of a different shape are skipped; returns the list of loaded keys."""
    sd = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, (str, bytes, os.PathLike)) else path_or_state
    own = net.state_dict()
    prefix = "resnet." if any(k.startswith("resnet.") for k in own) else ""
    picked = {}
    for k, v in sd.items():
        k2 = prefix + (k[7:] if k.startswith("module.") else k)
        if k2 in own and tuple(own[k2].shape) == tuple(v.shape):
            picked[k2] = v
    own.update(picked)
    net.load_state_dict(own)
    return sorted(picked)


def init_weights(net, init_type='normal', init_gain=0.02):
    """models/networks.py:77-108: every Conv / Linear weight ~ N(0, gain), biases 0, BN gamma ~ N(1, gain);
    LayerNorm and positional embeddings untouched."""
    if init_type != 'normal':
        raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
    roles = {k: r for k, _, r in state_spec(net.net_G)}
    with torch.no_grad():
        for k, p in net.named_parameters():
            r = roles[k]
            if r in ("conv_w", "lin_w"):
                p.normal_(0.0, init_gain)
            elif r == "bias":
                p.zero_()
            elif r == "bn_w":
                p.normal_(1.0, init_gain)
            elif r == "bn_b":
                p.zero_()
    print('initialize network with %s' % init_type)


MULTI_GPU_RECIPE = (
    "dahitra_amd runs one process per GPU (RCCL all-reduce of the flat gradient arena), not nn.DataParallel threads in one "
    "process: launch\n    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 "
    "main_cd.py --gpu_ids 0 ...\n(each rank takes the GPU of its LOCAL_RANK; CDTrainer joins the process group through "
    "dahitra_amd.parallel.init_from_env, shards the device-resident loader by rank and averages gradients per step; "
    "BatchNorm statistics stay per replica, as under the reference's DataParallel, models/networks.py:121-125)")


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    """models/networks.py:111-127.  More than one id selected nn.DataParallel there; here multi-GPU is one process per
    GPU, so a list of several ids is refused with the launch recipe instead of silently training on the first one."""
    if len(gpu_ids) > 1:
        raise ValueError("gpu_ids=%s: %s" % (list(gpu_ids), MULTI_GPU_RECIPE))
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(gpu_ids[0])
    init_weights(net, init_type, init_gain=init_gain)
    return net


def define_G(args, init_type='normal', init_gain=0.02, gpu_ids=[]):
    """models/networks.py:130-168; reads args.net_G (and the optional args.compute_dtype)."""
    if get_config(args.net_G).get("ctor_only"):   # NotImplementedError for unknown names, as the reference
        raise NotImplementedError("Generator model name [%s] is not recognized" % args.net_G)
    net = CDNet(args.net_G, getattr(args, "compute_dtype", None), getattr(args, "attn_dtype", None))
    return init_net(net, init_type, init_gain, gpu_ids)


def BASE_Transformer(input_nc=3, output_nc=2, with_pos='learned', resnet_stages_num=4, token_len=4, enc_depth=1,
                     dec_depth=1, decoder_dim_head=64, backbone='resnet18', **kw):
    """constructor-style access (models/networks.py:260-310) for the configurations define_G exposes, plus the
    ResNet-50 trunk (backbone='resnet50', networks.py:192-195) that only a constructor call reaches"""
    from ..netspec import NET_CONFIGS
    if backbone not in ("resnet18", "resnet50") or input_nc != 3 or with_pos != 'learned' or resnet_stages_num != 4:
        raise NotImplementedError("BASE_Transformer(backbone=%r, input_nc=%r, with_pos=%r, resnet_stages_num=%r) is "
                                  "outside the supported configurations" % (backbone, input_nc, with_pos,
                                                                           resnet_stages_num))
    for name, c in NET_CONFIGS.items():
        if c["kind"] == "bit" and c.get("backbone", "resnet18") == backbone and \
                (c["n_class"], c["token_len"], c["enc_depth"], c["dec_depth"], c["dec_dim_head"]) == \
                (output_nc, token_len, enc_depth, dec_depth, decoder_dim_head):
            return CDNet(name, kw.get("compute_dtype"), kw.get("attn_dtype"))
    raise NotImplementedError("BASE_Transformer configuration not covered by define_G's net_G table")


def BASE_Transformer_UNet(input_nc=3, output_nc=2, **kw):
    if output_nc != 2:
        raise NotImplementedError("newUNetTrans is defined with 2 classes (models/networks.py:163-165)")
    return CDNet("newUNetTrans", kw.get("compute_dtype"))
