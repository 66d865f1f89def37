"""CPU fp32 restatement of the DAHiTra change-detection hot path (ORACLE -- TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product package (dahitra_amd/) never does and fails loudly when its HIP library is missing.

This is a from-scratch *functional* restatement (state-dict in, tensors out) of the reference's
nn.Module code, written against torch.nn.functional on CPU.  It is pinned two ways:
  * in the build container, tests/test_oracle_vs_reference.py imports the real reference
    (oracle/ref_import.py) and compares logits / loss / gradients / AdamW-updated parameters;
  * everywhere (incl. the GPU box) tests/test_oracle_golden.py compares it with the committed
    fixtures tests/golden/*.npz, which were produced BY THE REFERENCE (oracle/make_golden.py).

Reference citations (relative to /root/reference):
  factory / net_G names ........ models/networks.py:130-168
  Siamese trunk ................ models/networks.py:233-257, models/resnet.py:35-73,125-225
  tokenizer .................... models/networks.py:312-319
  token encoder ................ models/networks.py:332-336,434-512
  cross-attention decoder ...... models/networks.py:338-347, models/help_funcs.py:66-114,170-186
  BiT tail ..................... models/networks.py:383-392, models/help_funcs.py:7-15
  hierarchical (newUNetTrans) .. models/networks.py:1118-1138,1146-1357
  focal loss / one_hot ......... models/losses.py:58-104,106-196
  train step ................... models/trainer.py:39-40,247-262,302-308
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# net_G -> configuration (models/networks.py:130-165)
# --------------------------------------------------------------------------------------------
NET_CONFIGS = {
    "base_transformer_pos_s4": dict(kind="bit", n_class=2, token_len=4, enc_depth=1, dec_depth=1,
                                    dec_dim_head=64),
    "base_transformer_pos_s4_dd8": dict(kind="bit", n_class=2, token_len=4, enc_depth=1,
                                        dec_depth=8, dec_dim_head=64),
    "base_transformer_pos_s4_dd8_o5": dict(kind="bit", n_class=5, token_len=4, enc_depth=1,
                                           dec_depth=8, dec_dim_head=64),
    "base_transformer_pos_s4_dd8_dedim8": dict(kind="bit", n_class=2, token_len=4, enc_depth=1,
                                               dec_depth=8, dec_dim_head=8),
    "base_transformer_pos_s4_dd8_t8_e2d4": dict(kind="bit", n_class=2, token_len=8, enc_depth=2,
                                                dec_depth=4, dec_dim_head=8),
    "newUNetTrans": dict(kind="unet", n_class=2, token_len=4, enc_depth=1),
    # NOT a define_G name: the reference reaches this model only through the constructor call
    # BASE_Transformer(..., backbone='resnet50') (networks.py:192-195).  It is
    # keyed here so that the same spec/forward/train helpers serve it.
    "base_transformer_pos_s4_resnet50": dict(kind="bit", n_class=2, token_len=4, enc_depth=1, dec_depth=1,
                                             dec_dim_head=64, backbone="resnet50"),
    # xBD copy of the hierarchical model (xBD_code/zoo/model_transformer_encoding.py:242-449, built at
    # xBD_code/train.py:44-45): ONE 6-channel input, 5 output channels, diff-only decoder pass, ModuleList
    # aliases in the state dict, and the `layer == 5/4/3` positional-embedding quirk.
    # "xbd_unet_transformer" is train.py's model (with_decoder_pos='learned' => 1024x1024 input only);
    # "_nodecpos" is the same constructor with with_decoder_pos=None, which runs at any multiple of 64.
    "xbd_unet_transformer": dict(kind="xbd", n_class=5, token_len=4, enc_depth=1, decoder_pos=True),
    "xbd_unet_transformer_nodecpos": dict(kind="xbd", n_class=5, token_len=4, enc_depth=1, decoder_pos=False),
}
DIM = 32           # transformer width everywhere (networks.py:289, 1178)
ENC_HEADS = 8      # BiT encoder/decoder heads (networks.py:305-310)
ENC_DIM_HEAD = 64
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LN_EPS = 1e-5

# hierarchical model: level suffix -> (trunk channels, heads, decoder depth, dim_head, pos size)
UNET_LEVELS = {
    5: dict(cin=256, heads=4, dec_depth=4, dim_head=64, size=16),
    4: dict(cin=128, heads=4, dec_depth=4, dim_head=64, size=32),
    3: dict(cin=64, heads=8, dec_depth=8, dim_head=64, size=64),
    2: dict(cin=64, heads=1, dec_depth=1, dim_head=32, size=64),   # registered, unused in forward
}


def get_config(net_G):
    if net_G not in NET_CONFIGS:
        raise NotImplementedError("Generator model name [%s] is not recognized" % net_G)
    return dict(NET_CONFIGS[net_G], net_G=net_G)


# --------------------------------------------------------------------------------------------
# state-dict specification (key, shape, role) in the reference's registration order
# --------------------------------------------------------------------------------------------
def _bn_spec(pfx, c):
    return [(pfx + ".weight", (c,), "bn_w"), (pfx + ".bias", (c,), "bn_b"),
            (pfx + ".running_mean", (c,), "bn_rm"), (pfx + ".running_var", (c,), "bn_rv"),
            (pfx + ".num_batches_tracked", (), "bn_nbt")]


def _resnet18_spec(pfx="resnet"):
    s = [(pfx + ".conv1.weight", (64, 3, 7, 7), "conv_w")] + _bn_spec(pfx + ".bn1", 64)
    cin = 64
    for li, planes in zip((1, 2, 3, 4), (64, 128, 256, 512)):
        for b in (0, 1):
            p = "%s.layer%d.%d" % (pfx, li, b)
            s.append((p + ".conv1.weight", (planes, cin, 3, 3), "conv_w"))
            s += _bn_spec(p + ".bn1", planes)
            s.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv_w"))
            s += _bn_spec(p + ".bn2", planes)
            if b == 0 and cin != planes:
                s.append((p + ".downsample.0.weight", (planes, cin, 1, 1), "conv_w"))
                s += _bn_spec(p + ".downsample.1", planes)
            cin = planes
    s += [(pfx + ".fc.weight", (1000, 512), "lin_w"), (pfx + ".fc.bias", (1000,), "bias")]
    return s


RESNET50_BLOCKS = (3, 4, 6, 3)


def _resnet50_spec(pfx="resnet"):
    """Bottleneck trunk (models/resnet.py:76-122, 261-270): width = planes, expansion 4; every layer's
    first block carries a 1x1 downsample (resnet.py:186-190)."""
    s = [(pfx + ".conv1.weight", (64, 3, 7, 7), "conv_w")] + _bn_spec(pfx + ".bn1", 64)
    cin = 64
    for li, planes, blocks in zip((1, 2, 3, 4), (64, 128, 256, 512), RESNET50_BLOCKS):
        for b in range(blocks):
            p = "%s.layer%d.%d" % (pfx, li, b)
            s.append((p + ".conv1.weight", (planes, cin, 1, 1), "conv_w"))
            s += _bn_spec(p + ".bn1", planes)
            s.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv_w"))
            s += _bn_spec(p + ".bn2", planes)
            s.append((p + ".conv3.weight", (4 * planes, planes, 1, 1), "conv_w"))
            s += _bn_spec(p + ".bn3", 4 * planes)
            if b == 0:
                s.append((p + ".downsample.0.weight", (4 * planes, cin, 1, 1), "conv_w"))
                s += _bn_spec(p + ".downsample.1", 4 * planes)
            cin = 4 * planes
    s += [(pfx + ".fc.weight", (1000, 2048), "lin_w"), (pfx + ".fc.bias", (1000,), "bias")]
    return s


def _encoder_spec(pfx, depth, heads, dim_head, mlp_dim):
    inner = heads * dim_head
    s = []
    for i in range(depth):
        a = "%s.layers.%d.0.fn" % (pfx, i)
        f = "%s.layers.%d.1.fn" % (pfx, i)
        s += [(a + ".norm.weight", (DIM,), "ln_w"), (a + ".norm.bias", (DIM,), "ln_b"),
              (a + ".fn.to_qkv.weight", (3 * inner, DIM), "lin_w"),
              (a + ".fn.to_out.0.weight", (DIM, inner), "lin_w"),
              (a + ".fn.to_out.0.bias", (DIM,), "bias"),
              (f + ".norm.weight", (DIM,), "ln_w"), (f + ".norm.bias", (DIM,), "ln_b"),
              (f + ".fn.net.0.weight", (mlp_dim, DIM), "lin_w"),
              (f + ".fn.net.0.bias", (mlp_dim,), "bias"),
              (f + ".fn.net.3.weight", (DIM, mlp_dim), "lin_w"),
              (f + ".fn.net.3.bias", (DIM,), "bias")]
    return s


def _decoder_spec(pfx, depth, heads, dim_head, mlp_dim):
    inner = heads * dim_head
    s = []
    for i in range(depth):
        a = "%s.layers.%d.0.fn" % (pfx, i)
        f = "%s.layers.%d.1.fn" % (pfx, i)
        s += [(a + ".norm.weight", (DIM,), "ln_w"), (a + ".norm.bias", (DIM,), "ln_b"),
              (a + ".fn.to_q.weight", (inner, DIM), "lin_w"),
              (a + ".fn.to_k.weight", (inner, DIM), "lin_w"),
              (a + ".fn.to_v.weight", (inner, DIM), "lin_w"),
              (a + ".fn.to_out.0.weight", (DIM, inner), "lin_w"),
              (a + ".fn.to_out.0.bias", (DIM,), "bias"),
              (f + ".norm.weight", (DIM,), "ln_w"), (f + ".norm.bias", (DIM,), "ln_b"),
              (f + ".fn.net.0.weight", (mlp_dim, DIM), "lin_w"),
              (f + ".fn.net.0.bias", (mlp_dim,), "bias"),
              (f + ".fn.net.3.weight", (DIM, mlp_dim), "lin_w"),
              (f + ".fn.net.3.bias", (DIM,), "bias")]
    return s


def state_spec(net_G):
    """[(key, shape, role)] in the same order as the reference module's state_dict()."""
    cfg = get_config(net_G)
    L = cfg["token_len"]
    if cfg["kind"] == "bit":
        s = [("pos_embedding", (1, 2 * L, DIM), "pos")]
        r50 = cfg.get("backbone") == "resnet50"
        s += _resnet50_spec() if r50 else _resnet18_spec()
        s += [("classifier.0.weight", (32, 32, 3, 3), "conv_w")] + _bn_spec("classifier.1", 32)
        s += [("classifier.3.weight", (cfg["n_class"], 32, 3, 3), "conv_w"),
              ("classifier.3.bias", (cfg["n_class"],), "bias"),
              ("conv_pred.weight", (32, 1024 if r50 else 256, 3, 3), "conv_w"), ("conv_pred.bias", (32,), "bias"),
              ("conv_a.weight", (L, 32, 1, 1), "conv_w")]
        s += _encoder_spec("transformer", cfg["enc_depth"], ENC_HEADS, ENC_DIM_HEAD, 2 * DIM)
        s += _decoder_spec("transformer_decoder", cfg["dec_depth"], ENC_HEADS,
                           cfg["dec_dim_head"], 2 * DIM)
        return s
    if cfg["kind"] == "xbd":
        return _xbd_spec(cfg)
    # hierarchical model (networks.py:1146-1249)
    s = [("pos_embedding_%d" % l, (1, 2 * L, DIM), "pos") for l in (5, 4, 3, 2)]
    s += [("pos_embedding_decoder_5", (1, DIM, 16, 16), "pos"),
          ("pos_embedding_decoder_4", (1, DIM, 32, 32), "pos"),
          ("pos_embedding_decoder_3", (1, DIM, 64, 64), "pos"),
          ("pos_embedding_decoder_2", (1, DIM, 64, 64), "pos")]
    s += _resnet18_spec()
    s += [("conv_pred.weight", (32, 384, 3, 3), "conv_w"), ("conv_pred.bias", (32,), "bias")]
    for l in (5, 4, 3, 2):
        s.append(("conv_squeeze_%d.0.weight" % l, (DIM, UNET_LEVELS[l]["cin"], 1, 1), "conv_w"))
    for l in (5, 4, 3, 2):
        s.append(("conv_token_%d.weight" % l, (L, DIM, 1, 1), "conv_w"))
    for l in (5, 4, 3, 2):
        s.append(("conv_decode_%d.weight" % l, (DIM, 2 * DIM, 3, 3), "conv_w"))
    for l in (5, 4, 3, 2):
        lv = UNET_LEVELS[l]
        s += _encoder_spec("transformer_%d" % l, cfg["enc_depth"], lv["heads"], lv["dim_head"], DIM)
        s += _decoder_spec("transformer_decoder_%d" % l, lv["dec_depth"], lv["heads"],
                           lv["dim_head"], DIM)
    s += [("conv_layer2_0.0.weight", (128, 128, 3, 3), "conv_w")] + _bn_spec("conv_layer2_0.1", 128)
    s += [("conv_layer2_0.3.weight", (32, 128, 3, 3), "conv_w"), ("conv_layer2_0.3.bias", (32,), "bias")]
    for l in (2, 3, 4):
        s += [("conv_layer%d.0.weight" % l, (32, 32, 3, 3), "conv_w"),
              ("conv_layer%d.0.bias" % l, (32,), "bias")]
    s += [("classifier.weight", (cfg["n_class"], 32, 3, 3), "conv_w"),
          ("classifier.bias", (cfg["n_class"],), "bias")]
    return s


def _xbd_spec(cfg):
    """xBD_code/zoo/model_transformer_encoding.py:255-340.  The nn.ModuleList holders (285-334) register every
    level's modules a second time, so state_dict() carries alias keys; role "alias:<primary key>"."""
    L = cfg["token_len"]
    s = [("pos_embedding_%d" % l, (1, 2 * L, DIM), "pos") for l in (5, 4, 3)]
    if cfg["decoder_pos"]:
        s += [("pos_embedding_decoder_%d" % l, (1, DIM, UNET_LEVELS[l]["size"], UNET_LEVELS[l]["size"]), "pos")
              for l in (5, 4, 3)]
    s += _resnet18_spec()
    s += [("conv_pred.weight", (32, 384, 3, 3), "conv_w"), ("conv_pred.bias", (32,), "bias")]

    def with_aliases(groups, holder):
        """groups: {level: [(key, shape, role)]} registered 5,4,3,2; the holder lists them 2,3,4,5"""
        out = []
        for l in (5, 4, 3, 2):
            out += groups[l][1]
        for i, l in enumerate((2, 3, 4, 5)):
            pfx, items = groups[l]
            out += [("%s.%d%s" % (holder, i, k[len(pfx):]), shp, "alias:" + k) for k, shp, _ in items]
        return out

    s += with_aliases({l: ("conv_squeeze_%d" % l, [("conv_squeeze_%d.0.weight" % l, (DIM, UNET_LEVELS[l]["cin"], 1, 1),
                                                     "conv_w")]) for l in (5, 4, 3, 2)}, "conv_squeeze_layers")
    s += with_aliases({l: ("conv_token_%d" % l, [("conv_token_%d.weight" % l, (L, DIM, 1, 1), "conv_w")])
                       for l in (5, 4, 3, 2)}, "conv_tokens_layers")
    s += with_aliases({l: ("conv_decode_%d" % l, [("conv_decode_%d.weight" % l, (DIM, 2 * DIM, 3, 3), "conv_w")])
                       for l in (5, 4, 3, 2)}, "conv_decode_layers")
    enc, dec = {}, {}
    for l in (5, 4, 3, 2):
        lv = UNET_LEVELS[l]
        enc[l] = ("transformer_%d" % l, _encoder_spec("transformer_%d" % l, cfg["enc_depth"], lv["heads"],
                                                       lv["dim_head"], DIM))
        dec[l] = ("transformer_decoder_%d" % l, _decoder_spec("transformer_decoder_%d" % l, lv["dec_depth"],
                                                               lv["heads"], lv["dim_head"], DIM))
        s += enc[l][1] + dec[l][1]
    for holder, groups in (("transformer_layers", enc), ("transformer_decoder_layers", dec)):
        for i, l in enumerate((2, 3, 4, 5)):
            pfx, items = groups[l]
            s += [("%s.%d%s" % (holder, i, k[len(pfx):]), shp, "alias:" + k) for k, shp, _ in items]
    s += [("conv_layer2_0.0.weight", (128, 128, 3, 3), "conv_w")] + _bn_spec("conv_layer2_0.1", 128)
    s += [("conv_layer2_0.3.weight", (32, 128, 3, 3), "conv_w"), ("conv_layer2_0.3.bias", (32,), "bias")]
    for l in (2, 3, 4):
        s += [("conv_layer%d.0.weight" % l, (32, 32, 3, 3), "conv_w"),
              ("conv_layer%d.0.bias" % l, (32,), "bias")]
    s += [("classifier.weight", (cfg["n_class"], 32, 3, 3), "conv_w"),
          ("classifier.bias", (cfg["n_class"],), "bias")]
    return s


def is_alias(role):
    return role.startswith("alias:")


def is_buffer(role):
    return role in ("bn_rm", "bn_rv", "bn_nbt")


# --------------------------------------------------------------------------------------------
# deterministic, RNG-independent state generator (SURVEY.md section 8c "Weights")
# --------------------------------------------------------------------------------------------
def _crc32(s):
    import zlib
    return zlib.crc32(s.encode()) & 0xFFFFFFFF


def _hash_uniform(seed, n):
    """n floats in [-1, 1): a splitmix64-style counter hash; exact in fp32 (24-bit mantissa)."""
    import numpy as np
    with np.errstate(over="ignore"):                       # wrap-around mod 2**64 is intended
        x = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) \
            + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)      # [0,1), 24 bits
    return (2.0 * u - 1.0).astype(np.float32)


def deterministic_state(net_G, salt=0, gain=1.0):
    """Well-conditioned synthetic weights (activations stay O(1), logit margins are not tiny).

    conv/linear ~ U(-a, a), a = gain*sqrt(3/fan_in); BN gamma 1+-0.1, beta +-0.1,
    running_mean +-0.1, running_var 1+-0.2; LN gamma 1+-0.1, beta +-0.1; bias +-0.05; pos +-0.5.
    """
    import numpy as np
    sd = OrderedDict()
    for key, shape, role in state_spec(net_G):
        if is_alias(role):
            sd[key] = sd[role[6:]]          # the same tensor object, as in the reference's state_dict()
            continue
        n = int(np.prod(shape)) if len(shape) else 1
        u = _hash_uniform(_crc32(key) + 7919 * salt, n)
        if role in ("conv_w", "lin_w"):
            fan_in = int(np.prod(shape[1:]))
            v = u * np.float32(gain * math.sqrt(3.0 / fan_in))
        elif role in ("bn_w", "ln_w"):
            v = np.float32(1.0) + np.float32(0.1) * u
        elif role in ("bn_b", "ln_b", "bn_rm"):
            v = np.float32(0.1) * u
        elif role == "bn_rv":
            v = np.float32(1.0) + np.float32(0.2) * u
        elif role == "bias":
            v = np.float32(0.05) * u
        elif role == "pos":
            v = np.float32(0.5) * u
        elif role == "bn_nbt":
            sd[key] = torch.zeros((), dtype=torch.int64)
            continue
        else:
            raise KeyError(role)
        sd[key] = torch.from_numpy(np.ascontiguousarray(v.reshape(shape)))
    return sd


def large_margin_state(net_G, salt=0):
    """deterministic_state with an ANTISYMMETRIC 2-class head (class-1 filter = -class-0 filter), so that the two logits
    are l and -l and the arg-max margin is 2|l|: the fraction of pixels inside a 4e-4 * max|logit| tie band drops from
    ~1 % to < 0.1 %, which makes "masks identical outside the band" a statement about (almost) every pixel."""
    sd = deterministic_state(net_G, salt)
    head = "classifier.3" if get_config(net_G)["kind"] == "bit" else "classifier"
    w, b = sd[head + ".weight"], sd[head + ".bias"]
    assert w.shape[0] == 2, "large_margin_state is defined for the 2-class heads"
    w[1] = -w[0]
    b[1] = -b[0]
    return sd


def synthetic_batch(batch, size, seed=1234, n_class=2, positive_frac=0.05):
    """A, B in [-1,1] and a sparse label map (SURVEY.md section 8d 'Synthetic inputs')."""
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(batch, 3, size, size, generator=g).clamp_(-1, 1)
    b = torch.randn(batch, 3, size, size, generator=g).clamp_(-1, 1)
    r = torch.rand(batch, 1, size, size, generator=g)
    if n_class == 2:
        lab = (r > 1.0 - positive_frac).to(torch.int64)
    else:
        lab = torch.where(r < 0.85, torch.zeros_like(r),
                          1 + torch.floor((r - 0.85) / 0.15 * (n_class - 1)).clamp_(max=n_class - 2)
                          ).to(torch.int64)
    return a, b, lab


# --------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------
def _bn(sd, pfx, x, training):
    y = F.batch_norm(x, sd[pfx + ".running_mean"], sd[pfx + ".running_var"], sd[pfx + ".weight"],
                     sd[pfx + ".bias"], training, BN_MOMENTUM, BN_EPS)
    if training:
        sd[pfx + ".num_batches_tracked"] += 1
    return y


def _basic_block(sd, pfx, x, stride, training):
    """models/resnet.py:58-73 (dilation is forced to 1, resnet.py:45-47)."""
    y = F.conv2d(x, sd[pfx + ".conv1.weight"], None, stride, 1)
    y = F.relu(_bn(sd, pfx + ".bn1", y, training))
    y = F.conv2d(y, sd[pfx + ".conv2.weight"], None, 1, 1)
    y = _bn(sd, pfx + ".bn2", y, training)
    if (pfx + ".downsample.0.weight") in sd:
        x = F.conv2d(x, sd[pfx + ".downsample.0.weight"], None, stride, 0)
        x = _bn(sd, pfx + ".downsample.1", x, training)
    return F.relu(y + x)


def _bottleneck(sd, pfx, x, stride, dilation, training):
    """models/resnet.py:102-122: 1x1 -> 3x3 (stride and dilation live here, padding = dilation) -> 1x1."""
    y = F.conv2d(x, sd[pfx + ".conv1.weight"])
    y = F.relu(_bn(sd, pfx + ".bn1", y, training))
    y = F.conv2d(y, sd[pfx + ".conv2.weight"], None, stride, dilation, dilation)
    y = F.relu(_bn(sd, pfx + ".bn2", y, training))
    y = F.conv2d(y, sd[pfx + ".conv3.weight"])
    y = _bn(sd, pfx + ".bn3", y, training)
    if (pfx + ".downsample.0.weight") in sd:
        x = F.conv2d(x, sd[pfx + ".downsample.0.weight"], None, stride, 0)
        x = _bn(sd, pfx + ".downsample.1", x, training)
    return F.relu(y + x)


def _res50_layer(sd, li, x, stride, first_dilation, dilation, training):
    """_make_layer (resnet.py:178-199): block 0 uses the PREVIOUS dilation, the rest the new one."""
    x = _bottleneck(sd, "resnet.layer%d.0" % li, x, stride, first_dilation, training)
    for b in range(1, RESNET50_BLOCKS[li - 1]):
        x = _bottleneck(sd, "resnet.layer%d.%d" % (li, b), x, 1, dilation, training)
    return x


def _res_layer(sd, li, x, stride, training):
    x = _basic_block(sd, "resnet.layer%d.0" % li, x, stride, training)
    return _basic_block(sd, "resnet.layer%d.1" % li, x, 1, training)


def _stem(sd, x, training):
    y = F.conv2d(x, sd["resnet.conv1.weight"], None, 2, 3)
    return F.relu(_bn(sd, "resnet.bn1", y, training))


def _tokenizer(x, w):
    """softmax over pixels of a 1x1 conv, then weighted pooling (networks.py:312-319)."""
    b, c, h, w_ = x.shape
    att = F.conv2d(x, w).reshape(b, -1, h * w_)
    att = torch.softmax(att, dim=-1)
    return att @ x.reshape(b, c, h * w_).transpose(1, 2)           # [b, L, c]


def _layer_norm(sd, pfx, x):
    return F.layer_norm(x, (DIM,), sd[pfx + ".weight"], sd[pfx + ".bias"], LN_EPS)


def _split_heads(t, heads):
    b, n, inner = t.shape
    return t.reshape(b, n, heads, inner // heads).permute(0, 2, 1, 3)   # b h n d


def _merge_heads(t):
    b, h, n, d = t.shape
    return t.permute(0, 2, 1, 3).reshape(b, n, h * d)


def _mlp(sd, pfx, x):
    h = F.gelu(F.linear(x, sd[pfx + ".net.0.weight"], sd[pfx + ".net.0.bias"]))
    return F.linear(h, sd[pfx + ".net.3.weight"], sd[pfx + ".net.3.bias"])


def _encoder(sd, pfx, x, depth, heads):
    """Token self-attention; scale is dim**-0.5 with dim=32, not dim_head (networks.py:461)."""
    scale = DIM ** -0.5
    for i in range(depth):
        a = "%s.layers.%d.0.fn" % (pfx, i)
        xn = _layer_norm(sd, a + ".norm", x)
        q, k, v = F.linear(xn, sd[a + ".fn.to_qkv.weight"]).chunk(3, dim=-1)
        q, k, v = (_split_heads(t, heads) for t in (q, k, v))
        att = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)
        o = _merge_heads(att @ v)
        x = F.linear(o, sd[a + ".fn.to_out.0.weight"], sd[a + ".fn.to_out.0.bias"]) + x
        f = "%s.layers.%d.1.fn" % (pfx, i)
        x = _mlp(sd, f + ".fn", _layer_norm(sd, f + ".norm", x)) + x
    return x


def _decoder(sd, pfx, x, m, depth, heads):
    """Cross attention of pixels x [b,n,32] on tokens m [b,L,32] (help_funcs.py:66-114,170-186).

    One LayerNorm is shared by x and m (PreNorm2, help_funcs.py:43-49); the residual adds the
    un-normalised x (Residual2, help_funcs.py:26-31)."""
    scale = DIM ** -0.5
    for i in range(depth):
        a = "%s.layers.%d.0.fn" % (pfx, i)
        xn = _layer_norm(sd, a + ".norm", x)
        mn = _layer_norm(sd, a + ".norm", m)
        q = _split_heads(F.linear(xn, sd[a + ".fn.to_q.weight"]), heads)
        k = _split_heads(F.linear(mn, sd[a + ".fn.to_k.weight"]), heads)
        v = _split_heads(F.linear(mn, sd[a + ".fn.to_v.weight"]), heads)
        att = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)
        o = _merge_heads(att @ v)
        x = F.linear(o, sd[a + ".fn.to_out.0.weight"], sd[a + ".fn.to_out.0.bias"]) + x
        f = "%s.layers.%d.1.fn" % (pfx, i)
        x = _mlp(sd, f + ".fn", _layer_norm(sd, f + ".norm", x)) + x
    return x


def _decode_map(sd, pfx, x, m, depth, heads, pos=None):
    b, c, h, w = x.shape
    if pos is not None:
        x = x + pos
    y = _decoder(sd, pfx, x.reshape(b, c, h * w).transpose(1, 2), m, depth, heads)
    return y.transpose(1, 2).reshape(b, c, h, w)


# --------------------------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------------------------
def _bit_forward(sd, cfg, x1, x2, training, taps):
    def trunk(x):
        x = _stem(sd, x, training)
        x = F.max_pool2d(x, 3, 2, 1)
        if cfg.get("backbone") == "resnet50":
            x = _res50_layer(sd, 1, x, 1, 1, 1, training)
            x = _res50_layer(sd, 2, x, 2, 1, 1, training)
            x = _res50_layer(sd, 3, x, 1, 1, 2, training)  # stride -> dilation 2, honoured by Bottleneck
        else:
            x = _res_layer(sd, 1, x, 1, training)
            x = _res_layer(sd, 2, x, 2, training)
            x = _res_layer(sd, 3, x, 1, training)          # stride replaced by (ignored) dilation
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return F.conv2d(x, sd["conv_pred.weight"], sd["conv_pred.bias"], 1, 1)

    f1 = trunk(x1)
    f2 = trunk(x2)
    t1 = _tokenizer(f1, sd["conv_a.weight"])
    t2 = _tokenizer(f2, sd["conv_a.weight"])
    tok = torch.cat([t1, t2], dim=1) + sd["pos_embedding"]
    tok = _encoder(sd, "transformer", tok, cfg["enc_depth"], ENC_HEADS)
    m1, m2 = tok.chunk(2, dim=1)
    d1 = _decode_map(sd, "transformer_decoder", f1, m1, cfg["dec_depth"], ENC_HEADS)
    d2 = _decode_map(sd, "transformer_decoder", f2, m2, cfg["dec_depth"], ENC_HEADS)
    diff = torch.abs(d1 - d2)
    up = F.interpolate(diff, scale_factor=4, mode="bilinear", align_corners=False)
    y = F.conv2d(up, sd["classifier.0.weight"], None, 1, 1)
    y = F.relu(_bn(sd, "classifier.1", y, training))
    logits = F.conv2d(y, sd["classifier.3.weight"], sd["classifier.3.bias"], 1, 1)
    if taps is not None:
        taps.update(feat1=f1, feat2=f2, tokens=tok, dec1=d1, dec2=d2, diff=diff)
    return logits


def _unet_forward(sd, cfg, x1, x2, training, taps):
    def trunk(x):
        s2 = _stem(sd, x, training)                      # in-place ReLU: the tap is post-ReLU
        s4 = _res_layer(sd, 1, F.max_pool2d(s2, 3, 2, 1), 1, training)
        s8 = _res_layer(sd, 2, s4, 2, training)
        s16 = _res_layer(sd, 3, F.max_pool2d(s8, 3, 2, 1), 1, training)
        return s2, s4, s8, s16

    def level(l, xa, xb):
        lv = UNET_LEVELS[l]
        wsq = sd["conv_squeeze_%d.0.weight" % l]
        xa = F.relu(F.conv2d(xa, wsq))
        xb = F.relu(F.conv2d(xb, wsq))
        ta = _tokenizer(xa, sd["conv_token_%d.weight" % l])
        tb = _tokenizer(xb, sd["conv_token_%d.weight" % l])
        tok = torch.cat([ta, tb], dim=1) + sd["pos_embedding_%d" % l]
        tok = _encoder(sd, "transformer_%d" % l, tok, cfg["enc_depth"], lv["heads"])
        ma, mb = tok.chunk(2, dim=1)
        pos = sd["pos_embedding_decoder_%d" % l]
        dp = "transformer_decoder_%d" % l
        da = _decode_map(sd, dp, xa, ma, lv["dec_depth"], lv["heads"], pos)
        db = _decode_map(sd, dp, xb, mb, lv["dec_depth"], lv["heads"], pos)
        dtok = torch.abs(mb - ma)
        dx = F.conv2d(torch.cat([da, db], dim=1), sd["conv_decode_%d.weight" % l], None, 1, 1)
        return _decode_map(sd, dp, dx, dtok, lv["dec_depth"], lv["heads"], pos)

    def up_conv(l, x):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return F.relu(F.conv2d(x, sd["conv_layer%d.0.weight" % l], sd["conv_layer%d.0.bias" % l], 1, 1))

    a2, a4, a8, a16 = trunk(x1)
    b2, b4, b8, b16 = trunk(x2)
    o5 = F.interpolate(level(5, a16, b16), scale_factor=2, mode="nearest")
    o4 = up_conv(4, level(4, a8, b8) + o5)
    o3 = up_conv(3, level(3, a4, b4) + o4)
    y = F.conv2d(torch.cat([a2, b2], dim=1), sd["conv_layer2_0.0.weight"], None, 1, 1)
    y = F.relu(_bn(sd, "conv_layer2_0.1", y, training))
    y = F.conv2d(y, sd["conv_layer2_0.3.weight"], sd["conv_layer2_0.3.bias"], 1, 1)
    o2 = up_conv(2, y + o3)
    logits = F.conv2d(o2, sd["classifier.weight"], sd["classifier.bias"], 1, 1)
    if taps is not None:
        taps.update(out5=o5, out4=o4, out3=o3, out2=o2)
    return logits


def _xbd_forward(sd, cfg, x1, x2, training, taps):
    """xBD_code/zoo/model_transformer_encoding.py:356-449.  Differences from _unet_forward: only the
    difference pass of the decoder runs (385-406), it reads conv_decode(cat[squeezed x1, squeezed x2]) and
    |token2 - token1|; positional embeddings are picked by `if layer == 5/4/3` while the callers pass the
    ModuleList index 3/2/1 (416-432), so ONLY the level-5 call (layer=3) adds pos_embedding_3 to its tokens
    and pos_embedding_decoder_3 [1,32,64,64] to its 1/16-scale map -- which therefore must be 64x64."""
    def trunk(x):
        s2 = _stem(sd, x, training)
        s4 = _res_layer(sd, 1, F.max_pool2d(s2, 3, 2, 1), 1, training)
        s8 = _res_layer(sd, 2, s4, 2, training)
        s16 = _res_layer(sd, 3, F.max_pool2d(s8, 3, 2, 1), 1, training)
        return s2, s4, s8, s16

    def level(l, xa, xb):
        lv = UNET_LEVELS[l]
        wsq = sd["conv_squeeze_%d.0.weight" % l]
        xa = F.relu(F.conv2d(xa, wsq))
        xb = F.relu(F.conv2d(xb, wsq))
        ta = _tokenizer(xa, sd["conv_token_%d.weight" % l])
        tb = _tokenizer(xb, sd["conv_token_%d.weight" % l])
        tok = torch.cat([ta, tb], dim=1)
        if l == 5:                                    # layer index 3 == the `if layer == 3` branch
            tok = tok + sd["pos_embedding_3"]
        tok = _encoder(sd, "transformer_%d" % l, tok, cfg["enc_depth"], lv["heads"])
        ma, mb = tok.chunk(2, dim=1)
        dtok = torch.abs(mb - ma)
        dx = F.conv2d(torch.cat([xa, xb], dim=1), sd["conv_decode_%d.weight" % l], None, 1, 1)
        pos = sd["pos_embedding_decoder_3"] if (l == 5 and cfg["decoder_pos"]) else None
        return _decode_map(sd, "transformer_decoder_%d" % l, dx, dtok, lv["dec_depth"], lv["heads"], pos)

    def up_conv(l, x):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return F.relu(F.conv2d(x, sd["conv_layer%d.0.weight" % l], sd["conv_layer%d.0.bias" % l], 1, 1))

    a2, a4, a8, a16 = trunk(x1)
    b2, b4, b8, b16 = trunk(x2)
    o5 = F.interpolate(level(5, a16, b16), scale_factor=2, mode="nearest")
    o4 = up_conv(4, level(4, a8, b8) + o5)
    o3 = up_conv(3, level(3, a4, b4) + o4)
    y = F.conv2d(torch.cat([a2, b2], dim=1), sd["conv_layer2_0.0.weight"], None, 1, 1)
    y = F.relu(_bn(sd, "conv_layer2_0.1", y, training))
    y = F.conv2d(y, sd["conv_layer2_0.3.weight"], sd["conv_layer2_0.3.bias"], 1, 1)
    o2 = up_conv(2, y + o3)
    logits = F.conv2d(o2, sd["classifier.weight"], sd["classifier.bias"], 1, 1)
    if taps is not None:
        taps.update(out5=o5, out4=o4, out3=o3, out2=o2)
    return logits


def forward(sd, net_G, x1, x2=None, training=False, taps=None):
    """logits [B, n_class, H, W]; BN buffers in `sd` are updated in place when training.  The xBD model
    takes ONE 6-channel tensor (model_transformer_encoding.py:409-412): pass it as x1 with x2=None."""
    cfg = get_config(net_G)
    if cfg["kind"] == "bit":
        return _bit_forward(sd, cfg, x1, x2, training, taps)
    if cfg["kind"] == "xbd":
        if x2 is None:
            x1, x2 = x1[:, :3], x1[:, 3:]
        return _xbd_forward(sd, cfg, x1, x2, training, taps)
    return _unet_forward(sd, cfg, x1, x2, training, taps)


# --------------------------------------------------------------------------------------------
# loss, mask, train step
# --------------------------------------------------------------------------------------------
def focal_loss(logits, target, alpha=0.5, gamma=2.0):
    """models/losses.py:106-196 with reduction='mean'; one-hot carries +1e-6 (losses.py:104)."""
    if target.dim() == logits.dim():
        target = target[:, 0]
    p = torch.softmax(logits, dim=1)
    logp = torch.log_softmax(logits, dim=1)
    oh = torch.zeros_like(logits).scatter_(1, target.unsqueeze(1), 1.0) + 1e-6
    focal = -alpha * torch.pow(1.0 - p, gamma) * logp
    return (oh * focal).sum(dim=1).mean()


def cross_entropy(logits, target, ignore_index=255):
    """models/losses.py:9-26, the batch-size-1 branch of trainer.py:260-261: F.cross_entropy with the hard-coded
    class weights [1, 1] (created with .cuda() in the reference, so that function itself cannot run on a CPU:
    restated here), ignore_index 255, mean reduction."""
    if target.dim() == logits.dim():
        target = target[:, 0]
    w = torch.ones(logits.shape[1], dtype=logits.dtype)
    return F.cross_entropy(logits, target.long(), weight=w, ignore_index=ignore_index, reduction="mean")


def dice_constant(logits, target, eps=1e-7):
    """The gradient-free dice term of trainer.py:256-259 (smp DiceLoss(mode='binary') applied to
    the argmax mask: from_logits => logsigmoid().exp(), dims (0,2), smooth 0).  PARITY UNPINNED:
    segmentation_models_pytorch is not vendored and its version is not stated."""
    pred = torch.argmax(logits, dim=1).to(torch.float32)
    bs = target.shape[0]
    y_true = target.reshape(bs, 1, -1).to(torch.float32)
    y_pred = F.logsigmoid(pred).exp().reshape(bs, 1, -1)
    inter = (y_pred * y_true).sum(dim=(0, 2))
    card = (y_pred + y_true).sum(dim=(0, 2))
    dice = (2.0 * inter) / card.clamp_min(eps)
    loss = (1.0 - dice) * (y_true.sum(dim=(0, 2)) > 0).to(torch.float32)
    return loss.mean()


def argmax_mask(logits):
    """trainer.py:170 / evaluator.py:101: first maximum wins."""
    return torch.argmax(logits, dim=1)


def trainable_keys(net_G):
    return [k for k, _, role in state_spec(net_G) if not is_buffer(role) and not is_alias(role)]


# ---- xBD train step (xBD_code/train.py:310-374) ---------------------------------------------------
XBD_CHANNEL_WEIGHTS = (0.05, 0.2, 0.8, 0.7, 0.4)        # train.py:353
XBD_EPS = 1e-6                                           # xBD_code/losses.py:12


def xbd_masks(label, n_class=5):
    """float masks [B,5,H,W] like TrainData's `msk` (train.py:150-166): channel 0 = any building,
    channels 1..4 = damage level; built from an integer map with values 0..4."""
    m = [(label > 0)] + [(label == c) for c in range(1, n_class)]
    return torch.stack([t.reshape(label.shape[0], *label.shape[-2:]) for t in m], dim=1).to(torch.float32)


def combo_loss_channel(logit, target):
    """ComboLoss({'dice': 1, 'focal': 8}) on one channel (xBD_code/losses.py:95-126): both terms read the
    sigmoid; dice over the whole batch (per_image=False, losses.py:24-34), FocalLoss2d gamma 2 with both
    operands clamped to [1e-6, 1-1e-6] (losses.py:273-288)."""
    s = torch.sigmoid(logit)
    t = target.float()
    so, to = s.reshape(1, -1), t.reshape(1, -1)
    inter = (so * to).sum(1)
    union = so.sum(1) + to.sum(1) + XBD_EPS
    dice = (1 - (2 * inter + XBD_EPS) / union).mean()
    o = s.reshape(-1).clamp(XBD_EPS, 1.0 - XBD_EPS)
    tt = t.reshape(-1).clamp(XBD_EPS, 1.0 - XBD_EPS)
    pt = (1 - tt) * (1 - o) + tt * o
    focal = (-(1.0 - pt) ** 2 * torch.log(pt)).mean()
    return 1 * dice + 8 * focal


def xbd_loss(logits, masks):
    """train.py:348-353"""
    return sum(w * combo_loss_channel(logits[:, c], masks[:, c]) for c, w in enumerate(XBD_CHANNEL_WEIGHTS))


class XbdTrainState:
    """train.py:331-374 + the hand-rolled AdamW of xBD_code/adamw.py:37-86 (eps is added to sqrt(v) BEFORE the
    bias correction, unlike torch.optim.AdamW) + clip_grad_norm_(0.999) BEFORE the step (train.py:373)."""

    def __init__(self, net_G, sd, lr=1e-4, weight_decay=1e-6, betas=(0.9, 0.999), eps=1e-8, max_norm=0.999):
        self.net_G = net_G
        self.sd = OrderedDict()
        for k, _, role in state_spec(net_G):
            if is_alias(role):
                self.sd[k] = self.sd[role[6:]]
                continue
            t = sd[k].detach().clone()
            if not is_buffer(role):
                t.requires_grad_(True)
            self.sd[k] = t
        self.params = [self.sd[k] for k in trainable_keys(net_G)]
        self.lr, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_norm
        self.state = {}
        self.last_total_norm = None

    def step(self, x6, masks):
        for p in self.params:
            p.grad = None
        logits = forward(self.sd, self.net_G, x6, None, training=True)
        loss = xbd_loss(logits, masks)
        loss.backward()
        grads = [p.grad for p in self.params if p.grad is not None]
        total = torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in grads]), 2.0)
        self.last_total_norm = float(total)
        coef = torch.clamp(self.max_norm / (total + 1e-6), max=1.0)
        b1, b2 = self.betas
        with torch.no_grad():
            for i, p in enumerate(self.params):
                if p.grad is None:
                    continue
                g = p.grad * coef
                st = self.state.setdefault(i, dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
                st["step"] += 1
                st["m"].mul_(b1).add_(g, alpha=1 - b1)
                st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = st["v"].sqrt().add_(self.eps)
                bc1, bc2 = 1 - b1 ** st["step"], 1 - b2 ** st["step"]
                step_size = self.lr * math.sqrt(bc2) / bc1
                if self.wd != 0:
                    p.add_(p, alpha=-self.wd * self.lr)
                p.addcdiv_(st["m"], denom, value=-step_size)
        return logits.detach(), float(loss.detach())


class TrainState:
    """Parameters as leaf tensors + a stock torch AdamW exactly as trainer.py:39-40 builds it."""

    def __init__(self, net_G, sd, lr=0.01):
        self.net_G = net_G
        self.sd = OrderedDict()
        for k, _, role in state_spec(net_G):
            if is_alias(role):
                self.sd[k] = self.sd[role[6:]]
                continue
            t = sd[k].detach().clone()
            if not is_buffer(role):
                t.requires_grad_(True)
            self.sd[k] = t
        self.params = [self.sd[k] for k in trainable_keys(net_G)]
        self.opt = torch.optim.AdamW(self.params, lr=lr, betas=(0.9, 0.999), weight_decay=0.01)

    def step(self, a, b, label):
        """forward -> zero_grad -> focal.backward -> AdamW.step (trainer.py:302-308).  The clip at
        trainer.py:308 runs after the step and never affects an update, so it is omitted."""
        logits = forward(self.sd, self.net_G, a, b, training=True)
        self.opt.zero_grad(set_to_none=True)
        loss = focal_loss(logits, label)
        loss.backward()
        self.opt.step()
        return logits.detach(), float(loss.detach())
