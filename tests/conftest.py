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
