"""CPU: the C-ABI library builds, loads, and exports every symbol include/sde_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "sde_hip.h")
LIB = os.path.join(ROOT, "simpledepthestimation_amd", "libsde_hip.so")


def declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sde_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(LIB)


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "sde_photo_fwd" in syms and "sde_last_error" in syms and len(syms) >= 15


def test_library_exports_every_declared_symbol(built):
    missing = [s for s in declared_symbols() if not hasattr(built, s)]
    assert not missing, f"declared in sde_hip.h but not exported: {missing}"


def test_python_binding_covers_header():
    from simpledepthestimation_amd.hip import lib as L
    import simpledepthestimation_amd.hip.photometric  # noqa: F401  (registers nothing extra, but must import without a GPU)
    try:
        import simpledepthestimation_amd.hip.nn  # noqa: F401
    except ImportError:
        pass
    import simpledepthestimation_amd.hip.evaluation  # noqa: F401
    import simpledepthestimation_amd.data.device_aug  # noqa: F401  (sde_image_prep_u8)
    bound = set(L._PROTOS) | {"sde_last_error"}
    missing = [s for s in declared_symbols() if s not in bound]
    assert not missing, f"no ctypes prototype for: {missing}"
    L.lib()   # loads + binds argtypes; raises if a prototype names a symbol the .so lacks


def declared_arity():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for name, params in re.findall(r"\b(sde_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        params = params.strip()
        out[name] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def test_python_prototypes_have_the_declared_number_of_arguments():
    """A ctypes prototype with one argument too few or too many still binds and only fails (or corrupts the call) on the GPU: compare every
    prototype's length with the parameter list of its declaration in include/sde_hip.h."""
    from simpledepthestimation_amd.hip import lib as L
    import simpledepthestimation_amd.hip.photometric  # noqa: F401
    import simpledepthestimation_amd.hip.nn  # noqa: F401
    import simpledepthestimation_amd.hip.evaluation  # noqa: F401
    arity = declared_arity()
    assert len(arity) >= 60, len(arity)
    bad = {n: (len(L._PROTOS[n][0]), k) for n, k in arity.items() if n in L._PROTOS and len(L._PROTOS[n][0]) != k}
    assert not bad, f"(prototype arguments, declared parameters) differ for: {bad}"


def test_error_path_without_gpu(built):
    built.sde_last_error.restype = ctypes.c_char_p
    built.sde_resize.restype = ctypes.c_int
    rc = built.sde_resize(None, None, 0, 0, 0, 0, 0, 0, None)     # rejected on the host before any launch
    assert rc < 0 and b"sde_resize" in built.sde_last_error()


def test_product_refuses_cpu_tensors():
    import torch
    from simpledepthestimation_amd.hip import photometric as P
    from simpledepthestimation_amd.hip.lib import SdeHipError
    with pytest.raises(SdeHipError):
        P.resize(torch.rand(1, 3, 8, 8), (4, 4))


def test_product_does_not_import_oracle():
    import subprocess, sys
    code = ("import sys; import simpledepthestimation_amd, simpledepthestimation_amd.hip.photometric; "
            "bad=[m for m in sys.modules if m=='oracle' or m.startswith('oracle.')]; sys.exit(1 if bad else 0)")
    assert subprocess.run([sys.executable, "-c", code], cwd=ROOT).returncode == 0
