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
