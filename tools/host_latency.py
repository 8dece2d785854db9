import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np
from mici_amd import integrators, models, systems
from mici_amd.states import ChainState
sys_ = systems.EuclideanMetricSystem(models.GaussIso(32))
integ = integrators.LeapfrogIntegrator(sys_, 0.1)
st = ChainState(pos=np.random.randn(32), mom=np.random.randn(32), dir=1)
for _ in range(50): st = integ.step(st)
t0=time.perf_counter(); n=2000
for _ in range(n): st = integ.step(st)
print("single-state step: %.1f us" % ((time.perf_counter()-t0)/n*1e6))
q=np.random.randn(4,32); p=np.random.randn(4,32)
for _ in range(20): integ.step_batch(q,p,1,n_steps=1)
t0=time.perf_counter()
for _ in range(n): q,p,_,_=integ.step_batch(q,p,1,n_steps=1)
print("step_batch N=4 n_steps=1: %.1f us" % ((time.perf_counter()-t0)/n*1e6))
t0=time.perf_counter()
for _ in range(200): integ.step_batch(q,p,1,n_steps=1000)
print("step_batch N=4 n_steps=1000: %.1f us per call" % ((time.perf_counter()-t0)/200*1e6))
h=sys_.h(st)
t0=time.perf_counter()
for _ in range(n): sys_.h(st)
print("system.h: %.1f us" % ((time.perf_counter()-t0)/n*1e6))

# ---- where the single-state step's time goes: the floor of "one kernel launch + one stream synchronisation" on this box
from mici_amd.runtime import DeviceBatch, default_context
ctx = default_context()
b = DeviceBatch(ctx, 1, 32, mapped=True)
b.upload(np.random.randn(1, 32), np.random.randn(1, 32), [1])
for _ in range(100):
    integ.step_device(b, 1, ctx)
ctx.sync()
t0 = time.perf_counter()
for _ in range(n):
    integ.step_device(b, 1, ctx)
ctx.sync()
print("launch only (asynchronous, amortised): %.1f us per call" % ((time.perf_counter() - t0) / n * 1e6))
t0 = time.perf_counter()
for _ in range(n):
    integ.step_device(b, 1, ctx)
    ctx.sync()
print("launch + stream synchronisation (the floor of a single-state step through any kernel): %.1f us"
      % ((time.perf_counter() - t0) / n * 1e6))
t0 = time.perf_counter()
for _ in range(n):
    ctx.sync()
print("stream synchronisation of an idle stream: %.1f us" % ((time.perf_counter() - t0) / n * 1e6))
t0 = time.perf_counter()
for _ in range(n):
    b.upload(st.pos[None], st.mom[None], [1])
    b.download_all()
print("upload + download_all of the mapped single-state buffer (no kernel): %.1f us" % ((time.perf_counter() - t0) / n * 1e6))
