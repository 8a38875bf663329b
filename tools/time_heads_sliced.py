"""cfg-4 pair kernel timing, fp64 vs sliced (GPU box); L2Q_LIB_NAME selects an A/B build"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops
sys.path.insert(0, os.path.dirname(__file__))
from probe_heads_sliced_lib import make
m, n = int(os.environ.get('M', 256)), int(os.environ.get('NN', 147456))
z, heads, v, f = make(m, n)
heads['sliced'] = ops.heads_sliced_build(heads)
for sl in ((False, True) if os.environ.get('BOTH', '1') == '1' else (True,)):
    ops.USE_SLICED_HEADS[0] = sl
    vv = v.clone()
    for _ in range(3):
        ops.vnet_heads_vupdate_pair_(z, heads, (1.0, 1.0, 1.0), vv, f, 1e-3, True, False, 1e-3, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.vnet_heads_vupdate_pair_(z, heads, (1.0, 1.0, 1.0), vv, f, 1e-3, True, False, 1e-3, True)
    e1.record(); torch.cuda.synchronize()
    print(f'{os.environ.get("L2Q_LIB_NAME", "libl2q.so")} M {m} N {n} pair sliced {sl}: {e0.elapsed_time(e1) / 20:.4f} ms', flush=True)
