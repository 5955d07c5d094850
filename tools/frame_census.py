"""phase stamps of the per-frame kernels (pose fwd / pose bwd / lbs_bwd_frame), block 60 thread 0.
Needs the census build: tools/ab_build.sh census -DLEMO_CENSUS ; LEMO_HIP_LIB=lemo_amd/csrc/build_ab/census.so"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
fit, _ = bench.build_problem(0, 119, dev, True, conv_variant=bench.DEFAULT_CONV_VARIANT)
lib = fit.lib
buf = torch.zeros(4 * 32, dtype=torch.int64, device=dev)
lib._dll.lemo_census_set.argtypes = [C.c_void_p]
assert lib._dll.lemo_census_set(buf.data_ptr()) == 0
fit.step(30, use_graph=False)
torch.cuda.synchronize()
for rep in range(3):
    buf.zero_()
    fit.step(10, use_graph=False)
    torch.cuda.synchronize()
    d = buf.cpu().numpy().reshape(4, 32)
    for k, name in enumerate(('pose_fwd', 'pose_bwd', 'lbs_bwd_frame')):
        t = d[k][d[k] > 0]
        if len(t) > 1:
            print(rep, name, 'total', t[-1] - t[0], 'phases', list(np.diff(t)))
