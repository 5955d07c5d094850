"""Where does the AE step engine's weight-gradient launch spend its time?  Runs the launch alone on a loaded engine in three
modes (lemo_ae_wgrad_probe): 0 = the product, 1 = operands loaded once per wave (MFMA + structure only), 2 = loads without MFMAs.
us per launch from HIP events around 20 launches.  Diagnostic, GPU box only.  Usage: python tools/ae_wgrad_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import synthetic, infill, _hip
from lemo_amd.infill import AE, finetune_and_infill
dev = torch.device('cuda:0')
lib = _hip.get_lib()
w = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_ae_weights(7).items()}
ae = AE().to(dev); ae.load_state_dict(w)
x = torch.randn(1, 4, 210, 135, device=dev); mask = torch.ones(210, 135, device=dev) > 0
finetune_and_infill(ae, w, x, mask, steps=2, use_graph=False)
ses = next(iter(infill._SESSIONS.values()))
torch.cuda.synchronize()
s = ses.stream.cuda_stream
with torch.cuda.stream(ses.stream):
    for mode, name in ((0, 'product'), (1, 'operands loaded once per wave'), (2, 'loads, no MFMAs'), (3, 'single register set (more waves)'), (4, 'no scheduling fences'), (0, 'product again')):
        for _ in range(3): lib.check(lib.ae_wgrad_probe(ses.h, mode, s))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ses.stream)
        for _ in range(20): lib.check(lib.ae_wgrad_probe(ses.h, mode, s))
        e1.record(ses.stream); torch.cuda.synchronize()
        print('%-32s %.1f us per launch' % (name, e0.elapsed_time(e1) * 1e3 / 20), flush=True)
