"""Device functions against the oracle, bit for bit, through the C ABI's unit-test hook (zygpu_test_eval):
integer work (PCG32, Sobol + Owen scrambling, ray offset) and the floating-point building blocks whose exact
rounding the image parity rests on (restated sin / cos / acos, SphQuad solid-angle sampling)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(product_lib):
    c = ctypes.c_void_p()
    assert product_lib.zygpu_create(0, ctypes.byref(c)) == 0, product_lib.zygpu_last_error()
    product_lib.zygpu_test_eval.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                            ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    yield c
    product_lib.zygpu_destroy(c)


def _eval(lib, ctx, op, inp, out_stride):
    inp = np.ascontiguousarray(inp)
    n = inp.shape[0]
    in_stride = inp.size // n
    out = np.zeros((n, out_stride), dtype=np.float32)
    assert lib.zygpu_test_eval(ctx, op, n, inp.ctypes.data, in_stride, out.ctypes.data, out_stride) == 0
    return out


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_trig_bit_exact(product_lib, oracle, ctx):
    rng = np.random.default_rng(3)
    n = 200000
    x = rng.uniform(-7, 7, n).astype(np.float32)
    x[: n // 2] = rng.uniform(-1, 1, n // 2).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1e-9, 3.1415927]
    g = _eval(product_lib, ctx, 0, x.reshape(-1, 1), 3)
    s, c, a = (np.zeros(n, np.float32) for _ in range(3))
    oracle.zyg_oracle_trig(n, _vp(x), _vp(s), _vp(c), _vp(a))
    assert np.array_equal(g[:, 0].view(np.uint32), s.view(np.uint32))
    assert np.array_equal(g[:, 1].view(np.uint32), c.view(np.uint32))
    both_nan = np.isnan(g[:, 2]) & np.isnan(a)
    assert np.all((g[:, 2].view(np.uint32) == a.view(np.uint32)) | both_nan)


def test_exp_atan2_bit_exact(product_lib, oracle, ctx):
    """Beer-Lambert attenuation inside glass (exp) and the spherical uv of Sphere fragments (atan2)."""
    rng = np.random.default_rng(12)
    n = 200000
    inp = np.zeros((n, 2), np.float32)
    inp[:, 0] = -rng.exponential(3.0, n)
    inp[: n // 4, 0] = rng.uniform(-100, 2, n // 4)
    inp[:6, 0] = [0.0, -0.0, -1e-5, -0.3, -0.4, -88.0]
    inp[:, 1] = rng.uniform(-3, 3, n)
    g = _eval(product_lib, ctx, 6, inp, 2)
    e, a = np.zeros(n, np.float32), np.zeros(n, np.float32)
    oracle.zyg_oracle_exp(n, _vp(np.ascontiguousarray(inp[:, 0])), _vp(e))
    oracle.zyg_oracle_atan2(n, _vp(np.ascontiguousarray(inp[:, 1])), _vp(np.ascontiguousarray(inp[:, 0])), _vp(a))
    assert np.array_equal(g[:, 0].view(np.uint32), e.view(np.uint32))
    assert np.array_equal(g[:, 1].view(np.uint32), a.view(np.uint32))


def test_tan_bit_exact(product_lib, oracle, ctx):
    """The disk light's equi-angular sampling: tan of angles in (-pi/2, pi/2) (disk.zig:229-234), musl's tanf on both sides."""
    rng = np.random.default_rng(21)
    n = 200000
    x = rng.uniform(-1.5707, 1.5707, n).astype(np.float32)
    x[:8] = [0.0, -0.0, 1e-5, -1e-5, 0.7853982, -0.7853982, 1.5707, -1.5707]
    g = _eval(product_lib, ctx, 7, x.reshape(-1, 1), 1)
    c = np.zeros(n, np.float32)
    oracle.zyg_oracle_tan(n, _vp(x), _vp(c))
    assert np.array_equal(g[:, 0].view(np.uint32), c.view(np.uint32))
    ref = np.tan(x.astype(np.float64))
    assert np.max(np.abs(c - ref) / np.maximum(np.abs(ref), 1e-30)) < 2e-7  # within an ulp of the true tangent


def test_sphquad_bit_exact(product_lib, oracle, ctx):
    rng = np.random.default_rng(4)
    n = 100000
    inp = np.zeros((n, 7), np.float32)
    inp[:, 0:2] = rng.uniform(0.2, 2.0, (n, 2))
    inp[:, 2:5] = rng.uniform(-2, 2, (n, 3))
    inp[:, 5:7] = rng.uniform(0, 1, (n, 2))
    g = _eval(product_lib, ctx, 1, inp, 5)
    c = np.zeros((n, 5), np.float32)
    oracle.zyg_oracle_sphquad_eval(n, _vp(inp), _vp(c))
    same = (g.view(np.uint32) == c.view(np.uint32)) | (np.isnan(g) & np.isnan(c))
    assert same.all()


def test_sobol_stream_bit_exact(product_lib, oracle, ctx):
    rng = np.random.default_rng(6)
    n = 5000
    seeds = rng.integers(0, 2 ** 32, (n, 2), dtype=np.uint64).astype(np.uint32)
    seeds[0] = [0, 0]
    seeds[1] = [0xFFFFFFFF, 0xFFFFFFFF]
    g = _eval(product_lib, ctx, 2, seeds.view(np.float32), 10)
    buf = (ctypes.c_float * 10)()
    for i in range(0, n, 37):
        oracle.zyg_oracle_sobol_stream(ctypes.c_uint32(int(seeds[i, 0])), ctypes.c_uint32(int(seeds[i, 1])), 10, 5, buf)
        assert list(g[i]) == list(buf), i
    assert np.all((g >= 0) & (g <= 1))


def test_pcg_bit_exact(product_lib, oracle, ctx):
    rng = np.random.default_rng(7)
    n = 2000
    st = rng.integers(0, 2 ** 32, (n, 2), dtype=np.uint64).astype(np.uint32)
    st[0] = [42, 54]
    g = _eval(product_lib, ctx, 3, st.view(np.float32), 4)
    u = (ctypes.c_uint32 * 4)()
    f = (ctypes.c_float * 4)()
    for i in range(0, n, 13):
        oracle.zyg_oracle_pcg(int(st[i, 0]), int(st[i, 1]), 4, u, f)
        assert list(g[i]) == list(f), i


def test_offset_ray_bit_exact(product_lib, oracle, ctx):
    rng = np.random.default_rng(8)
    n = 20000
    p = rng.uniform(-50, 50, (n, 3)).astype(np.float32)
    p[: n // 4] *= 1e-3  # exercise the |p| < 1/32 float path
    nn = rng.normal(size=(n, 3))
    nn = (nn / np.linalg.norm(nn, axis=1, keepdims=True)).astype(np.float32)
    g = _eval(product_lib, ctx, 4, np.concatenate([p, nn], axis=1), 3)
    out = (ctypes.c_float * 3)()
    for i in range(0, n, 11):
        oracle.zyg_oracle_offset_ray(_vp(p[i]), _vp(nn[i]), out)
        assert list(g[i]) == list(out), i
