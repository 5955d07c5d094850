"""phase breakdown of the LBS vertex-forward kernel (diagnostic, GPU box only)"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import _hip, synthetic
from lemo_amd._hip import ptr
from lemo_amd.body_model import BodyModelData, DeviceBody, alloc_pose_ws
lib = _hip.get_lib(); dev = torch.device('cuda:0')
data = BodyModelData(synthetic.make_synthetic_smplx(seed=0)); db = DeviceBody(data, dev)
B = 119
ws, tt, Bp = alloc_pose_ws(B, data.nj, dev, db.blend_f16)
tt['Xg'].normal_(); tt['A'].normal_()
PRE = os.environ.get('LBS_PRE', '1') == '1'          # B operand pre-split by the pose kernel (values do not matter for the timing)
XGS = ptr(tt['XgS']) if PRE else None
verts = torch.empty(B, data.V, 3, device=dev); vp = torch.empty_like(verts)
nblk = (data.V + 41) // 42
dbg = torch.zeros(nblk * 8 * 4, dtype=torch.int64, device=dev)
s = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    lib.check(lib.lbs_verts_fwd_census(C.byref(db.skin), ptr(tt['Xg']), Bp, ptr(tt['A']), data.nj, None, data.V, B, ptr(verts), ptr(vp), ptr(dbg), s, XGS))
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nblk, 8, 4)
t0, tp, tg, t1 = d[..., 0], d[..., 1], d[..., 2], d[..., 3]
print('pre-split B' if PRE else 'B converted per workgroup', '| blocks', nblk, 'median cycles: prologue %d  gemm %d  handover+skinning %d  total %d (max %d)' % (
    np.median(tp - t0), np.median(tg - tp), np.median(t1 - tg), np.median(t1 - t0), (t1 - t0).max()))
# wall time of the product kernel (no census stamps), HIP events around 20 launches
def run():
    lib.check(lib.lbs_verts_fwd_xs(C.byref(db.skin), ptr(tt['Xg']), XGS, Bp, ptr(tt['A']), data.nj, None, None, data.V, B, ptr(verts), ptr(vp), s))
for _ in range(5): run()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
print('lbs_verts_fwd (product kernel): %.2f us per launch (best of 5 x 20 back-to-back launches)' % best)
