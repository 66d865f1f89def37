"""ctypes binding of libdahitra_hip.so (C ABI: include/dahitra_hip.h).

The library is the product's only arithmetic path.  There is no CPU / eager fallback: if the
shared object is missing or a call fails, a RuntimeError is raised."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# DAHITRA_HIP_LIB: an experiment build of the same library (tools/*_timeline.py); the product path is the in-tree one
LIB_PATH = os.environ.get("DAHITRA_HIP_LIB") or os.path.join(_HERE, "lib", "libdahitra_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dahitra_hip.h")

_lib = None


class HipLibraryError(RuntimeError):
    pass


def declared_symbols():
    """Every dh_* function declared in include/dahitra_hip.h."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dh_[a-z0-9_]+)\s*\(", txt)))


_LONG_RET = {"dh_conv3x3_dgrad_up4_partial_floats", "dh_encoder_bwd_workspace_size", "dh_encoder_saved_floats", "dh_xattn_prep_bwd_stack_workspace_size", "dh_combo_loss_workspace_size", "dh_grad_norm_workspace_size", "dh_conv2d_wgrad_workspace_size", "dh_bn_bwd_workspace_size", "dh_layernorm_bwd_workspace_size",
             "dh_stem_pool_bn_bwd_workspace_size", "dh_conv2d_wgrad_phase_workspace_size", "dh_tokenizer_bwd_workspace_size", "dh_decoder_layer_bwd_workspace_size", "dh_tokenizer_fwd_workspace_size", "dh_xattn_prep_bwd_workspace_size", "dh_head_bn_bwd_workspace_size"}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                "dahitra_amd: %s not found -- build it with `make` (or __graft_entry__.build()); "
                "there is no fallback path" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.dh_last_error.restype = ctypes.c_char_p
        for name in _LONG_RET:
            getattr(_lib, name).restype = ctypes.c_long
    return _lib


def check(rc, what):
    if rc != 0:
        raise HipLibraryError("%s failed: %s" % (what, lib().dh_last_error().decode()))
