"""Where do the ~0.9 ms of fixed cost in a timed bench region come from?  (diagnostic, GPU box)
  a) first launch of a graph exec vs later launches (per schedule level)
  b) first call of params72() / gather (torch kernels loaded lazily)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
fit, prob = bench.build_problem(0, 119, dev, True, 3)
s = torch.cuda.Stream(dev)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return th * 1e3, (time.perf_counter() - t0) * 1e3


def steps(n):
    def f():
        with torch.cuda.stream(s):
            fit.step(n, use_graph=True)
    return f


with torch.cuda.stream(s):
    fit.prepare(5); fit.prepare(20); fit.prepare(100)
print('after prepare(5, 20, 100) [capture + instantiate + upload, nothing launched]')
for n in (1, 1, 5, 5, 20, 20, 20, 100, 100):
    h, t = timed(steps(n))
    print(f'  step({n:3d}): host {h:7.3f} ms  total {t:7.3f} ms  -> {t / n * 1e3:7.1f} us/iteration')
h, t = timed(lambda: fit.params72())
print(f'params72() first: host {h:.3f} total {t:.3f} ms')
h, t = timed(lambda: fit.params72())
print(f'params72() second: host {h:.3f} total {t:.3f} ms')
from lemo_amd.sharding import gather_fitted_params
h, t = timed(lambda: gather_fitted_params(fit.params72()[None]))
print(f'gather first: host {h:.3f} total {t:.3f} ms')
h, t = timed(lambda: gather_fitted_params(fit.params72()[None]))
print(f'gather second: host {h:.3f} total {t:.3f} ms')
for n in (20, 20):
    h, t = timed(steps(n))
    print(f'  step({n:3d}): host {h:7.3f} ms  total {t:7.3f} ms  -> {t / n * 1e3:7.1f} us/iteration')
