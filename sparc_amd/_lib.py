"""ctypes binding of libsparc_hip.so (the C ABI declared in include/sparc_hip.h).

The library is mandatory: there is no CPU or eager fallback.  If it has not been
built (python -m sparc_amd.build) importing this module raises.
"""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPARC_HIP_LIB") or os.path.join(_HERE, "lib", "libsparc_hip.so")   # SPARC_HIP_LIB: another build of it

ERRORS = {-1: "SPARC_ERR_ARG (bad argument)", -2: "SPARC_ERR_WS (workspace too small or misaligned)",
          -3: "SPARC_ERR_LAUNCH (HIP launch failed)",
          -4: "SPARC_ERR_SIZE (problem larger than the fixed-size entry point supports)"}


class SparcHipError(RuntimeError):
    pass


class Tiling(ctypes.Structure):
    """sparc_tiling_t of include/sparc_hip.h: a host struct describing a device tiling buffer."""
    _fields_ = [("buf", c_void_p), ("ndesc", c_int32), ("rows", c_int32), ("vec", c_int32), ("threads", c_int32),
                ("kp", c_int32), ("dual", c_int32), ("s_addr", c_void_p), ("s_val", c_void_p), ("s_valT", c_void_p),
                ("o_row", c_void_p), ("o_val", c_void_p), ("o_valT", c_void_p), ("row_scale", c_void_p)]


class AdamTensor(ctypes.Structure):
    """sparc_adam_tensor_t of include/sparc_hip.h."""
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n", c_int64)]


class Blocks(ctypes.Structure):
    """sparc_blocks_t of include/sparc_hip.h: a host struct describing the dense diagonal blocks of a batch."""
    _fields_ = [("desc", c_void_p), ("ndesc", c_int32), ("max_p", c_int32), ("blk", c_void_p), ("blkT", c_void_p)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise SparcHipError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -m sparc_amd.build); "
            "sparc_amd has no fallback path")
    return ctypes.CDLL(LIB_PATH)


lib = _load()

P = c_void_p
_SIGS = {
    "sparc_abi_version": (c_int, []),
    "sparc_init": (c_int, []),
    "sparc_csr_build_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sparc_csr_build": (c_int, [c_int64, c_int64, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "sparc_affinity_prep_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sparc_affinity_prep": (c_int, [c_int64, c_int64, P, P, P, P, c_double, P, P, P, P, c_size_t, P]),
    "sparc_lap_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "sparc_lap_fwd": (c_int, [c_int64, c_int64, P, P, P, P, P, P, c_int, c_int64, P, P, P, P, P, c_size_t, P]),
    "sparc_spmm": (c_int, [c_int64, c_int64, P, P, P, c_int, P, P, P, P, P]),
    "sparc_spmm_split": (c_int, [c_int64, c_int64, P, P, P, c_int, P, P, P, c_int64, c_int64, P, P, c_int64, P, P, P, P]),
    "sparc_lap_fwd_blocks": (c_int, [c_int64, c_int64, P, P, P, P, P, P, c_int, c_int64, P, P, P, P, P, c_size_t, P]),
    "sparc_blocks_spmm": (c_int, [c_int64, c_int, P, P, P, P, P]),
    "sparc_profile_begin": (c_int, [c_int, c_int]),
    "sparc_profile_end": (c_int, [P, P]),
    "sparc_gram_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "sparc_gram": (c_int, [c_int64, c_int, P, c_int64, P, c_int64, P, P, c_size_t, P]),
    "sparc_gram_static": (c_int, [c_int64, c_int, P, c_int64, P, c_int64, P, P]),
    "sparc_gram_reduce": (c_int, [c_int, c_int, P, P, P]),
    "sparc_gram_apply": (c_int, [c_int64, c_int, P, c_int64, P, c_int64, P, P, c_int, P, c_int64, P, c_size_t, P]),
    "sparc_orth_weights": (c_int, [c_int64, c_int, P, P, c_int64, P, c_int, P, P, P, P, P, P]),
    "sparc_orth_weights_host_idx": (c_int, [c_int64, c_int, P, P, c_int64, P, c_int, c_int, P, P, P, P, P, P]),
    "sparc_orth_apply": (c_int, [c_int64, c_int, P, c_int64, P, c_int, P, c_int64, P]),
    "sparc_orth_weights_bwd": (c_int, [c_int64, c_int, P, P, P, P, c_int, P, P]),
    "sparc_orth_weights_bwd_f64": (c_int, [c_int64, c_int, P, P, P, P, c_int, P, P]),
    "sparc_orth_apply_f64": (c_int, [c_int64, c_int, P, c_int64, P, P, c_int64, P]),
    "sparc_act_bwd_bias_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "sparc_act_bwd_bias": (c_int, [c_int64, c_int, c_int, P, c_int64, P, c_int64, P, c_int64, P, P, c_size_t, P]),
    "sparc_wgrad_small_workspace_bytes": (c_size_t, [c_int, c_int]),
    "sparc_wgrad_small": (c_int, [c_int64, c_int, c_int, P, c_int64, P, c_int64, P, P, c_size_t, P]),
    "sparc_first_layer_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "sparc_first_layer_bwd": (c_int, [c_int64, c_int, c_int, P, c_int64, P, c_int64, P, c_int64, P, P, c_size_t, P]),
    "sparc_first_layer_bwd_recompute": (c_int, [c_int64, c_int, c_int, P, c_int64, P, c_int64, P, P, P, P, c_size_t, P]),
    "sparc_linear_narrow_fwd": (c_int, [c_int64, c_int, c_int, c_int, P, c_int64, P, P, P, c_int64, P]),
    "sparc_linear_narrow_fwd_gram_workspace_bytes": (c_size_t, [c_int]),
    "sparc_linear_narrow_fwd_gram": (c_int, [c_int64, c_int, c_int, c_int, P, c_int64, P, P, P, c_int64, c_int64, P, P, c_size_t, P]),
    "sparc_linear_narrow_in_fwd": (c_int, [c_int64, c_int, c_int, c_int, P, c_int64, P, P, P, c_int64, P]),
    "sparc_dgrad_narrow_relu_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "sparc_dgrad_narrow_relu": (c_int, [c_int64, c_int, c_int, P, c_int64, P, P, c_int64, P, c_int64, P, P, c_size_t, P]),
    "sparc_dgrad_wide_relu_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "sparc_dgrad_wide_relu": (c_int, [c_int64, c_int, c_int, P, c_int64, P, P, c_int64, P, c_int64, P, P, c_size_t, P]),
    "sparc_dgrad_wide_first_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "sparc_dgrad_wide_first": (c_int, [c_int64, c_int, c_int, c_int, P, c_int64, P, P, c_int64, P, P, P, P, c_size_t, P]),
    "sparc_adam_step": (c_int, [c_int, P, c_double, c_double, c_double, c_double, c_double, c_int64, P]),
    "sparc_row_rank_values": (c_int, [c_int64, c_int64, P, c_int64, c_int, P, P, P]),
    "sparc_label_scores": (c_int, [c_int64, P, P, P, P, c_int64, P, c_int, P, P, P, P]),
    "sparc_subblock_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sparc_subblock_count": (c_int, [c_int64, P, P, P, c_int64, P, c_int, P, P, P, c_size_t, P]),
    "sparc_subblock_fill": (c_int, [c_int64, P, P, P, c_int64, P, c_int, c_double, P, P, P, P, P, c_size_t, P]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTS = tuple(_SIGS)


_initialised = set()


def init_device(index: int) -> None:
    """sparc_init() once per device, with that device current."""
    if index in _initialised:
        return
    import torch
    with torch.cuda.device(index):
        check(lib.sparc_init(), "sparc_init")
    _initialised.add(index)


def check(rc, what):
    if rc != 0:
        raise SparcHipError(f"{what} failed: {ERRORS.get(rc, rc)}")
