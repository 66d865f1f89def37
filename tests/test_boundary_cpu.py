"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol the header
declares, the module surface matches the reference's state-dict names, and the product fails loudly
(no fallback) when asked to compute without the GPU.  No compute calls here."""
import json
import os
import subprocess
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    lib = os.path.join(ROOT, "dahitra_amd", "lib", "libdahitra_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-j8", "-C", ROOT])
    return lib


def test_library_exports_every_declared_symbol(built):
    from dahitra_amd import _lib
    lib = _lib.lib()
    syms = _lib.declared_symbols()
    assert len(syms) >= 50
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.dh_abi_version() == 1


def test_header_has_no_torch_types():
    txt = open(os.path.join(ROOT, "include", "dahitra_hip.h")).read()
    import re
    code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)          # declarations only (comments cite torch call sites)
    code = re.sub(r"//[^\n]*", "", code)
    assert "torch" not in code.lower() and "at::" not in code and "c10::" not in code
    assert "at::Tensor" not in txt and "#include <torch" not in txt


def test_state_dict_keys_equal_reference(golden_dir):
    from dahitra_amd.models.networks import BASE_Transformer, define_G
    ref = json.load(open(os.path.join(golden_dir, "state_keys.json")))
    for name, keys in ref.items():
        if name.endswith("_resnet50"):          # only a constructor call reaches it (networks.py:192-195)
            with pytest.raises(NotImplementedError):
                define_G(types.SimpleNamespace(net_G=name))
            net = BASE_Transformer(input_nc=3, output_nc=2, token_len=4, resnet_stages_num=4, with_pos='learned',
                                   backbone='resnet50')
        elif name.startswith("xbd_"):           # xBD_code/train.py:44-45 builds it by constructor
            from dahitra_amd.models.xbd import BASE_Transformer_UNet
            with pytest.raises(NotImplementedError):
                define_G(types.SimpleNamespace(net_G=name))
            net = BASE_Transformer_UNet(input_nc=3, output_nc=5, token_len=4, resnet_stages_num=4, with_pos='learned',
                                        with_decoder_pos=None if name.endswith("nodecpos") else 'learned',
                                        enc_depth=1, dec_depth=8)
            assert sum(p.numel() for p in net.parameters()) == 13250765 - (0 if not name.endswith("nodecpos") else
                                                                           32 * (16 * 16 + 32 * 32 + 64 * 64))
        else:
            net = define_G(types.SimpleNamespace(net_G=name))
        sd = net.state_dict()
        assert [[k, list(v.shape)] for k, v in sd.items()] == keys, name


def test_netspec_agrees_with_oracle_spec():
    import cdnet_ref as O
    from dahitra_amd import netspec
    for name in O.NET_CONFIGS:
        assert netspec.state_spec(name) == O.state_spec(name), name


def test_define_G_contract():
    from dahitra_amd.models.networks import define_G, get_scheduler
    with pytest.raises(NotImplementedError):
        define_G(types.SimpleNamespace(net_G="no_such_net"))
    net = define_G(types.SimpleNamespace(net_G="base_transformer_pos_s4"))
    # init_weights semantics (models/networks.py:88-105): N(0, 0.02) weights, zero biases, BN gamma ~ N(1, 0.02)
    sd = net.state_dict()
    assert abs(float(sd["resnet.layer1.0.conv1.weight"].std()) - 0.02) < 2e-3
    assert float(sd["conv_pred.bias"].abs().max()) == 0.0
    assert abs(float(sd["resnet.bn1.weight"].mean()) - 1.0) < 0.02
    assert float(sd["transformer.layers.0.0.fn.norm.weight"].min()) == 1.0      # LayerNorm untouched
    opt = torch.optim.SGD(net.parameters(), lr=1.0)
    sch = get_scheduler(opt, types.SimpleNamespace(lr_policy="linear", max_epochs=9))
    sch.step()
    assert abs(opt.param_groups[0]["lr"] - 0.9) < 1e-9
    assert isinstance(get_scheduler(opt, types.SimpleNamespace(lr_policy="bogus", max_epochs=9)), NotImplementedError)


def test_unused_parameter_contract():
    """17 (BiT) / 48 (newUNetTrans) tensors never receive a gradient"""
    from dahitra_amd import netspec
    for name, n in (("base_transformer_pos_s4", 17), ("newUNetTrans", 48), ("base_transformer_pos_s4_resnet50", 32), ("xbd_unet_transformer", 50),
                    ("xbd_unet_transformer_nodecpos", 48)):
        inactive = [k for k, _, r in netspec.state_spec(name)
                    if not netspec.is_buffer(r) and not netspec.is_alias(r) and not netspec.is_active(name, k)]
        assert len(inactive) == n, (name, len(inactive))


def test_cpu_tensors_are_refused():
    from dahitra_amd import _lib
    from dahitra_amd.models.networks import define_G
    net = define_G(types.SimpleNamespace(net_G="base_transformer_pos_s4"))
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(_lib.HipLibraryError):
        net(x, x)


def test_product_never_imports_oracle():
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "dahitra_amd")):
        for fn in fns:
            if fn.endswith(".py") and fn != "smoke.py":
                txt = open(os.path.join(dp, fn)).read()
                if "cdnet_ref" in txt or "ref_import" in txt or "import oracle" in txt:
                    bad.append(fn)
    assert not bad, bad


def test_every_test_name_the_documents_cite_exists():
    """DESIGN.md / README.md / INTEGRATION.md cite tests as evidence: every `test_...` name in them must resolve to a test
    function (names ending in `_` or `_*` are prefixes) or a test file of this tree"""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    defined, files = set(), set()
    for f in glob.glob(os.path.join(root, "tests", "test_*.py")):
        files.add(os.path.basename(f)[:-3])
        defined.update(re.findall(r"^def (test_[A-Za-z0-9_]+)\(", open(f).read(), flags=re.M))
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        for name in sorted(set(re.findall(r"\btest_[A-Za-z0-9_]+", open(os.path.join(root, doc)).read()))):
            if name in defined or name in files:
                continue
            if name.endswith("_") and any(d.startswith(name) for d in defined | files):
                continue
            missing.append((doc, name))
    assert not missing, missing


def test_bf16x3_compute_mode_is_the_fp32_pipeline_with_split_products(monkeypatch):
    """compute_dtype="bf16x3" (host logic, no GPU): fp32 tensors, the forward in the fp16-plane three-product form, the backward
    in the bf16-plane three-product form; DAHITRA_F32_MMA=bf16x3 switches every fp32 net; an unknown name is refused like the reference refuses
    an unknown net_G"""
    from dahitra_amd.models.networks import CDNet
    net = CDNet("base_transformer_pos_s4", "bf16x3")
    assert net.compute_dtype == torch.float32 and net.mma_x3
    assert (net._engine.dtype, net._engine.mma_fwd, net._engine.mma_bwd) == (torch.float32, 3, 1)
    plain = CDNet("base_transformer_pos_s4", "fp32")
    assert not plain.mma_x3 and (plain._engine.mma_fwd, plain._engine.mma_bwd) == (0, 0)
    assert not CDNet("base_transformer_pos_s4", "bf16").mma_x3
    monkeypatch.setenv("DAHITRA_F32_MMA", "bf16x3")
    assert CDNet("base_transformer_pos_s4", "fp32").mma_x3 and not CDNet("base_transformer_pos_s4", "bf16").mma_x3
    with pytest.raises(ValueError):
        CDNet("base_transformer_pos_s4", "fp16x3")


def test_bench_line_contract_fields_are_produced_by_the_source():
    """bench.py (static check, no GPU): the sub-records the driver's line carries since round 5 and the switches that drop them"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    for key in ('"parity_mode"', '"secondary"', '"ddp_rehearsal"', '"exact_fp32"', '"bf16x3_vs_fp32"', '"roofline"', '"cpu_baseline"'):
        assert key in src, key
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True).stdout
    for flag in ("--no-secondary", "--no-ddp-rehearsal", "--no-roofline", "--no-parity-mode", "bf16x3"):
        assert flag in out, flag
