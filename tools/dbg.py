import ctypes, sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import zyg_amd
l = zyg_amd.load_library()
O = ctypes.CDLL('oracle/liboracle.so')
ctx = ctypes.c_void_p(); assert l.zygpu_create(0, ctypes.byref(ctx)) == 0
l.zygpu_test_eval.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
rng = np.random.default_rng(3)
n = 200000
x = rng.uniform(-7, 7, n).astype(np.float32); x[:n//2] = rng.uniform(-1,1,n//2).astype(np.float32)
g = np.zeros((n,3), np.float32)
assert l.zygpu_test_eval(ctx, 0, n, x.ctypes.data, 1, g.ctypes.data, 3) == 0
s = np.zeros(n, np.float32); c = np.zeros(n, np.float32); a = np.zeros(n, np.float32)
O.zyg_oracle_trig(n, x.ctypes.data_as(ctypes.c_void_p), s.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p), a.ctypes.data_as(ctypes.c_void_p))
eq = lambda u,v: np.mean((u==v)|(np.isnan(u)&np.isnan(v)))
print('sin eq', eq(g[:,0],s), 'cos eq', eq(g[:,1],c), 'acos eq', eq(g[:,2],a))
inp = np.zeros((n,7), np.float32)
inp[:,0:2] = rng.uniform(0.2, 2.0, (n,2)); inp[:,2:5] = rng.uniform(-2, 2, (n,3)); inp[:,5:7] = rng.uniform(0,1,(n,2))
go = np.zeros((n,5), np.float32); co = np.zeros((n,5), np.float32)
assert l.zygpu_test_eval(ctx, 1, n, inp.ctypes.data, 7, go.ctypes.data, 5) == 0
O.zyg_oracle_sphquad_eval(n, inp.ctypes.data_as(ctypes.c_void_p), co.ctypes.data_as(ctypes.c_void_p))
for k,name in enumerate(['S','pdf','x','y','z']):
    print(name, 'eq', eq(go[:,k], co[:,k]))
ab = rng.uniform(-2, 2, (n,2)).astype(np.float32)
go = np.zeros((n,6), np.float32); co = np.zeros((n,6), np.float32)
assert l.zygpu_test_eval(ctx, 5, n, ab.ctypes.data, 2, go.ctypes.data, 6) == 0
O.zyg_oracle_basic_ops(n, ab.ctypes.data_as(ctypes.c_void_p), co.ctypes.data_as(ctypes.c_void_p))
for k,name in enumerate(['div','sqrt','muladd','rcp','acosR','poly']):
    print(name, 'eq', eq(go[:,k], co[:,k]))
