"""The C-ABI library loads without a GPU and exports every symbol include/sparc_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sparc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sparc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for s in ["sparc_csr_build", "sparc_affinity_prep", "sparc_lap_fwd", "sparc_gram", "sparc_orth_weights",
              "sparc_orth_apply", "sparc_orth_weights_bwd", "sparc_subblock_count", "sparc_subblock_fill"]:
        assert s in syms


def test_library_exports_every_declared_symbol():
    from sparc_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    assert set(declared_symbols()) == set(_lib.EXPORTS)
    assert lib.sparc_abi_version() == 1


def test_workspace_queries_without_gpu():
    from sparc_amd._lib import lib
    assert lib.sparc_lap_workspace_bytes(1000, 64, 0) >= 3 * 1000 * 128 * 4
    assert lib.sparc_lap_workspace_bytes(1000, 65, 0) == 0          # k out of range
    assert lib.sparc_gram_workspace_bytes(1000, 64) > 0
    assert lib.sparc_affinity_prep_workspace_bytes(1000, 5000) > 0
    assert lib.sparc_csr_build_workspace_bytes(1000, 5000) > 0
    assert lib.sparc_subblock_workspace_bytes(1000, 10) > 0


def test_bad_arguments_are_rejected_before_any_launch():
    from sparc_amd._lib import lib
    assert lib.sparc_spmm(-1, 0, None, None, None, 64, None, None, None, None, 0, 0, 0, None) == -1
    assert lib.sparc_gram(10, 65, None, 65, None, 65, None, None, 0, None) == -1
    assert lib.sparc_orth_apply(0, 8, None, 8, None, 0, None, 8, None) == -1


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "sparc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f


def test_every_entry_point_rejects_null_pointers_without_touching_the_device():
    """Every int-returning entry point, called with null pointers (all sizes zero, then plausible sizes), returns an
    error code: the checks come before the first launch, so this runs (and must not crash) on a machine without a GPU.
    A child process makes the calls, so that a crash is reported as one."""
    import subprocess
    import sys
    code = r'''
import ctypes, sys
sys.path.insert(0, %r)
from sparc_amd import _lib
skip = {"sparc_abi_version", "sparc_profile_begin", "sparc_profile_end"}
for name, (res, args) in _lib._SIGS.items():
    if res is not ctypes.c_int or name in skip:
        continue
    fn = getattr(_lib.lib, name)
    for fill in (0, 10):
        vals = []
        for a in args:
            if a is ctypes.c_void_p:
                vals.append(None)
            elif a is ctypes.c_double:
                vals.append(0.0)
            elif a is ctypes.c_size_t:
                vals.append(0)
            else:
                vals.append(fill)
        print(name, fill, flush=True)
        rc = fn(*vals)
        assert rc < 0, (name, fill, rc)
print("ALL_REJECTED")
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ALL_REJECTED" in out.stdout, out.stdout[-600:] + out.stderr[-1500:]


def test_out_of_range_widths_are_rejected_before_any_launch():
    """Entry points whose kernels are built for k (or a narrow width) up to 64 refuse 65 with an argument / size error
    instead of launching: non-null pointers, plausible row counts, every int argument 65."""
    from sparc_amd import _lib
    narrow = ["sparc_gram", "sparc_gram_static", "sparc_orth_weights", "sparc_orth_weights_host_idx", "sparc_orth_apply", "sparc_orth_weights_bwd",
              "sparc_orth_weights_bwd_f64", "sparc_orth_apply_f64", "sparc_act_bwd_bias", "sparc_wgrad_small",
              "sparc_first_layer_bwd", "sparc_linear_narrow_fwd", "sparc_linear_narrow_fwd_gram", "sparc_dgrad_narrow_relu",
              "sparc_gram_reduce", "sparc_gram_apply", "sparc_dgrad_wide_relu", "sparc_dgrad_wide_first"]
    for name in narrow:
        res, args = _lib._SIGS[name]
        vals = []
        for a in args:
            vals.append(4096 if a is ctypes.c_void_p else 0.0 if a is ctypes.c_double else (1 << 30) if a is ctypes.c_size_t
                        else 65 if a is ctypes.c_int else 100)
        vals[-1] = None
        assert getattr(_lib.lib, name)(*vals) in (-1, -4), name
