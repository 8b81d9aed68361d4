import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_oracle():
    import importlib.util
    spec = importlib.util.spec_from_file_location("oracle_build", os.path.join(ROOT, "oracle", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.ensure_built()


def _build_product():
    from zyg_amd import build
    return build.ensure_built()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; never used by the product)."""
    import zyg_amd
    lib = ctypes.CDLL(_build_oracle())
    lib.zyg_oracle_hash.restype = ctypes.c_uint32
    lib.zyg_oracle_hash.argtypes = [ctypes.c_uint32]
    lib.zyg_oracle_random_light.restype = ctypes.c_uint32
    lib.zyg_oracle_random_light.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                            ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    lib.zyg_oracle_light_pdf.restype = ctypes.c_float
    lib.zyg_oracle_light_pdf.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                         ctypes.c_float, ctypes.c_uint32]
    lib.zyg_oracle_render.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                      ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.c_int32, ctypes.POINTER(zyg_amd.Stats)]
    lib.zyg_oracle_resolve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    lib.zyg_oracle_pcg.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    return lib


@pytest.fixture(scope="session")
def product_lib():
    """libzyg_amd.so (host + HIP).  Loading needs no GPU; compute calls do."""
    import zyg_amd
    _build_product()
    return zyg_amd.load_library()


@pytest.fixture()
def engine(product_lib):
    import zyg_amd
    e = zyg_amd.Engine(product_lib)
    yield e
    e.release()


def _variant_library():
    """variants/lib_all.so: the library with the kernels and options that lost their A/B (kernel_table.hpp ZYG_VARIANTS), built by
    `python -m zyg_amd.build --variant all --units 4,5,6,7,-1 -DZYG_VARIANTS=1` from the same sources; None when it is absent or older than they."""
    import zyg_amd
    path = os.path.join(ROOT, "variants", "lib_all.so")
    if not os.path.exists(path):
        return None
    src = os.path.join(ROOT, "zyg_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(d, f)) for d, _, files in os.walk(src) for f in files)
    if os.path.getmtime(path) < newest and not os.environ.get("ZYG_VARIANT_LIB_ANY_AGE"):
        return None
    return zyg_amd.load_library(path)


@pytest.fixture(scope="session")
def variant_lib_or_none(product_lib):
    return _variant_library()


@pytest.fixture(scope="session")
def variant_lib(variant_lib_or_none):
    if variant_lib_or_none is None:
        pytest.skip("variants/lib_all.so is not built for these sources (python -m zyg_amd.build --variant all --units 4,5,6,7,-1 -DZYG_VARIANTS=1)")
    return variant_lib_or_none


@pytest.fixture()
def variant_engine(variant_lib):
    import zyg_amd
    e = zyg_amd.Engine(variant_lib)
    yield e
    e.release()
