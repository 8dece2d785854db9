import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import models as omdl
from mici_amd import _ffi, models, systems
from mici_amd.runtime import DeviceBatch, default_context
dim=256; n=1
rng=np.random.default_rng(dim)
om=omdl.Rank1Metric(omdl.make_spd(dim, rng))
system=systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(om.base))
x=rng.standard_normal((n,dim)); b=rng.standard_normal((n,dim))
ctx=default_context(); fn=ctx._lib.mm_debug_blk16la_linalg
fn.restype=C.c_int; fn.argtypes=[C.c_void_p,C.c_void_p,C.c_void_p,C.c_int,_ffi.c_double_p,_ffi.c_int32_p,C.c_int,_ffi.c_double_p]
batch=DeviceBatch(ctx,n,dim); batch.upload(x,b,np.ones(n,dtype=np.int8))
out=np.zeros((n,256,256)); st=np.zeros(n,dtype=np.int32)
_ffi.check(fn(ctx.handle, system.device_model(ctx).handle, batch.handle, 0, out.ctypes.data_as(_ffi.c_double_p), st.ctypes.data_as(_ffi.c_int32_p), 0, None), ctx.handle, "dbg")
want=np.linalg.inv(om.metric_func(x[0]))
err=np.abs(out[0]-want).reshape(16,16,16,16).max(axis=(1,3))
np.set_printoptions(linewidth=250, precision=1)
print("status", st)
print((np.log10(err+1e-300)).round(0).astype(int))
