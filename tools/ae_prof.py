"""rocprof target: 20 eager finetune steps of the infilling AE at [1,4,210,135] (diagnostic).
Usage: python tools/ae_prof.py [engine|autograd] [graph]   (default: the native step engine, eager launches)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import synthetic
from lemo_amd.infill import AE, finetune_and_infill
dev = torch.device('cuda:0')
engine = (sys.argv[1] if len(sys.argv) > 1 else 'engine') == 'engine'
w = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_ae_weights(7).items()}
ae = AE().to(dev); ae.load_state_dict(w)
x = torch.randn(1, 4, 210, 135, device=dev); mask = torch.ones(210, 135, device=dev) > 0
finetune_and_infill(ae, w, x, mask, steps=20, lr=3e-6, use_graph='graph' in sys.argv[2:], engine=engine)
torch.cuda.synchronize()
