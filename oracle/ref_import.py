"""Container-only harness: import the read-only reference at /root/reference as a parity oracle.

TEST INFRASTRUCTURE -- never imported by the product package (dahitra_amd/).  It is used only by
oracle/make_golden.py (to generate tests/golden/*.npz) and by the `-m "not gpu"` tests that pin
oracle/cdnet_ref.py against the reference itself.  /root/reference does not exist on the GPU box;
everything here is skipped there.

Stubs: torchvision / timm / segmentation_models_pytorch are imported by
the reference (models/networks.py:6-7,17; models/losses.py:3) but are not needed by the hot path.
The ImageNet download in models/resnet.py:228-234 is replaced by a no-op (init_weights overwrites
every conv/linear weight anyway, models/networks.py:88-105).
"""
import os
import sys
import types
import contextlib
import io

REF_ROOT = os.environ.get("DAHITRA_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_loaded = {}


def load():
    """Returns (networks_module, losses_module) of the reference."""
    if _loaded:
        return _loaded["networks"], _loaded["losses"]
    if not available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    import torch

    def _na(*a, **k):
        raise RuntimeError("stubbed third-party symbol called on the hot path")

    tv = _stub("torchvision")
    tvm = _stub("torchvision.models", resnet34=_na)
    tv.models = tvm
    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.layers", DropPath=torch.nn.Identity, to_2tuple=lambda x: (x, x),
          trunc_normal_=lambda t, std=0.02: t)
    smp = _stub("segmentation_models_pytorch")
    smp.losses = _stub("segmentation_models_pytorch.losses", DiceLoss=_na)
    # the reference package is called `models`; make sure ours never shadows it
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    sys.path.insert(0, REF_ROOT)
    try:
        import models.resnet as ref_resnet
        ref_resnet.load_state_dict_from_url = lambda *a, **k: None
        _orig = ref_resnet.ResNet.load_state_dict
        ref_resnet.ResNet.load_state_dict = (
            lambda self, sd, *a, **k: None if sd is None else _orig(self, sd, *a, **k))
        import models.networks as ref_networks
        import models.losses as ref_losses
    finally:
        sys.path.remove(REF_ROOT)
    _loaded["networks"] = ref_networks
    _loaded["losses"] = ref_losses
    return ref_networks, ref_losses


def define_G(net_G):
    """Reference define_G (models/networks.py:130-168) on CPU; chatter silenced."""
    nets, _ = load()
    args = types.SimpleNamespace(net_G=net_G)
    with contextlib.redirect_stdout(io.StringIO()):
        return nets.define_G(args=args, gpu_ids=[])


def build_resnet50_variant(output_nc=2):
    """C5 model: BASE_Transformer(backbone='resnet50') (models/networks.py:186-197)."""
    nets, _ = load()
    with contextlib.redirect_stdout(io.StringIO()):
        net = nets.BASE_Transformer(input_nc=3, output_nc=output_nc, token_len=4,
                                    resnet_stages_num=4, with_pos='learned', backbone='resnet50')
        return nets.init_net(net, 'normal', 0.02, [])


# ---------------------------------------------------------------------------------------------------
# xBD copy of the hierarchical model
# ---------------------------------------------------------------------------------------------------
_xbd = {}


def load_xbd():
    """Returns (model_module, losses_module, adamw_module) of xBD_code/.  The model file loads its trunk
    with SourceFileLoader('zoo/bit_resnet.py') relative to the working directory
    (zoo/model_transformer_encoding.py:10-11), hence the temporary chdir."""
    if _xbd:
        return _xbd["model"], _xbd["losses"], _xbd["adamw"]
    load()          # installs the torchvision stubs
    import importlib.util
    import torch
    tvm = sys.modules["torchvision.models"]
    for n in ("resnet34", "resnet18", "efficientnet_b0"):
        setattr(tvm, n, getattr(tvm, n, None) or (lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))))
    import torch.hub
    xroot = os.path.join(REF_ROOT, "xBD_code")
    cwd = os.getcwd()
    hub_orig = torch.hub.load_state_dict_from_url
    os.chdir(xroot)
    try:
        def imp(name, rel):
            spec = importlib.util.spec_from_file_location(name, os.path.join(xroot, rel))
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            return m
        torch.hub.load_state_dict_from_url = lambda *a, **k: None
        model = imp("_xbd_model_transformer_encoding", "zoo/model_transformer_encoding.py")
        bm = model.bitmodule
        bm.load_state_dict_from_url = lambda *a, **k: None
        _o = bm.ResNet.load_state_dict
        bm.ResNet.load_state_dict = lambda self, sd, *a, **k: None if sd is None else _o(self, sd, *a, **k)
        _xbd["model"] = model
        _xbd["losses"] = imp("_xbd_losses", "losses.py")
        _xbd["adamw"] = imp("_xbd_adamw", "adamw.py")
    finally:
        os.chdir(cwd)
        torch.hub.load_state_dict_from_url = hub_orig
    return _xbd["model"], _xbd["losses"], _xbd["adamw"]


def build_xbd_model(with_decoder_pos='learned'):
    """the model xBD_code/train.py:44-45 builds (default torch init; no init_weights in that script)"""
    model, _, _ = load_xbd()
    with contextlib.redirect_stdout(io.StringIO()):
        return model.BASE_Transformer_UNet(input_nc=3, output_nc=5, token_len=4, resnet_stages_num=4,
                                           with_pos='learned', with_decoder_pos=with_decoder_pos,
                                           enc_depth=1, dec_depth=8)


# ---------------------------------------------------------------------------------------------------
# data path of the reference (datasets/CD_dataset.py, datasets/data_utils.py, misc/metric_tool.py)
# ---------------------------------------------------------------------------------------------------
_data = {}


def load_data():
    """Returns (CD_dataset module, data_utils module, metric_tool module) of the reference.

    datasets/data_utils.py needs five torchvision.transforms.functional primitives (to_pil_image, hflip, vflip, to_tensor,
    normalize) and imports cv2 / sklearn it never calls on the CDDataset path.  torchvision is not installed here, so the
    five primitives are supplied with their documented semantics (PIL transposes, uint8 HWC -> float CHW / 255,
    (x - mean) / std); every decision of the augmentation -- crop origin, which flips, the blur radius, the order of the
    `random` draws, PIL's GaussianBlur -- is the reference's own code."""
    if _data:
        return _data["cd"], _data["du"], _data["metric"]
    load()
    import numpy as np
    import torch
    from PIL import Image

    def to_pil_image(img):
        return img if isinstance(img, Image.Image) else Image.fromarray(np.asarray(img))

    def to_tensor(img):
        a = np.asarray(img)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return t.float().div(255) if t.dtype == torch.uint8 else t.float()

    def normalize(t, mean, std):
        m = torch.tensor(mean, dtype=t.dtype).view(-1, 1, 1)
        s = torch.tensor(std, dtype=t.dtype).view(-1, 1, 1)
        return (t - m) / s

    tf = _stub("torchvision.transforms.functional", to_pil_image=to_pil_image, to_tensor=to_tensor, normalize=normalize,
               hflip=lambda im: im.transpose(Image.FLIP_LEFT_RIGHT), vflip=lambda im: im.transpose(Image.FLIP_TOP_BOTTOM),
               rotate=lambda im, angle: im.rotate(angle))
    tt = _stub("torchvision.transforms", functional=tf)
    sys.modules["torchvision"].transforms = tt
    _stub("cv2")
    if "sklearn.model_selection" not in sys.modules:
        try:
            import sklearn.model_selection  # noqa: F401
        except Exception:
            _stub("sklearn")
            _stub("sklearn.model_selection", train_test_split=None)
    if not hasattr(np, "str"):
        np.str = str                    # datasets/CD_dataset.py:32 (default argument evaluated lazily, numpy < 1.24 name)
    for k in [k for k in sys.modules if k == "datasets" or k.startswith("datasets.") or k == "misc" or k.startswith("misc.")]:
        del sys.modules[k]
    # the reference's datasets/ and misc/ have no __init__.py (namespace packages): a regular `datasets` package in
    # site-packages (HuggingFace) would win the import, so the two packages are registered by path
    for pkg in ("datasets", "misc"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF_ROOT, pkg)]
        sys.modules[pkg] = m
    sys.path.insert(0, REF_ROOT)
    try:
        import datasets.data_utils as du
        import datasets.CD_dataset as cd
        import misc.metric_tool as metric
    finally:
        sys.path.remove(REF_ROOT)
        for pkg in ("datasets", "misc"):          # leave no shadow of the reference's package names behind
            for k in [k for k in sys.modules if k == pkg or k.startswith(pkg + ".")]:
                del sys.modules[k]
    _data.update(cd=cd, du=du, metric=metric)
    return cd, du, metric
