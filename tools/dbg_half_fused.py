"""Diagnostic: fused vs un-fused half-precision heads (test_u1_half_precision_networks case
64x64 / relu / bf16): where does the log-det difference of the outlier chains come from?"""
import sys
import numpy as np
import torch
sys.path[:0] = ['/root/repo', '/root/repo/l2hmc-qcd_amd', '/root/repo/tests']
import l2hmc.configs as cfgs
from l2hmc.dynamics.pytorch.dynamics import Dynamics
from l2hmc.lattice.u1.pytorch.lattice import LatticeU1
from l2hmc.network.pytorch.network import NetworkFactory

hd = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
lat, nb, units, act, bn = (64, 64), 24, [128, 96], 'relu', False
torch.set_default_dtype(torch.float32)
torch.manual_seed(5)
np.random.seed(5)
dc = cfgs.DynamicsConfig(nchains=nb, group='U1', latvolume=list(lat), nleapfrog=2, eps=0.1,
                         eps_hmc=0.1, verbose=False)
nc = cfgs.NetworkConfig(units=units, activation_fn=act, dropout_prob=0.0, use_batch_norm=bn)
spec = cfgs.InputSpec(xshape=tuple(dc.xshape), xnet={'x': [dc.xdim, 2], 'v': [dc.xdim]},
                      vnet={'x': [dc.xdim], 'v': [dc.xdim]})
latt = LatticeU1(nb, list(lat))
dyn = Dynamics(latt.action, dc, NetworkFactory(spec, nc, cfgs.ConvolutionConfig())).eval()
g = torch.Generator().manual_seed(7)
with torch.no_grad():
    for n_, p in dyn.named_parameters():
        if n_.endswith('coeff'):
            p.copy_(0.3 * torch.randn(p.shape, generator=g).to(p.device))
x0 = latt.random().to(dyn.device)
v0 = torch.randn(nb, dc.xdim, generator=g).to(dyn.device)
dyn.set_net_precision(hd)
dyn.fuse_u1_steps = False
from l2hmc import _ops as ops
for forward in (True, False):
    out = {}
    for fused in (True, False):
        dyn.fuse_half_heads = fused
        # individual sub-updates of LF step 1
        xn, vn = dyn._pack(x0), v0.clone()
        lds = []
        real_v = dyn._v_update_n if hasattr(dyn, '_v_update_n') else None
        ld = dyn._lf_n(1, xn, vn, 2.5, forward)
        out[fused] = (xn.clone(), vn.clone(), ld.clone())
    d = (out[True][2] - out[False][2]).abs()
    print('forward', forward, 'ld diff per chain', [f'{float(a):.4f}' for a in d])
    dv = (out[True][1] - out[False][1]).abs()
    print('  v diff max per chain', [f'{float(a):.1e}' for a in dv.max(1).values])
    big = int(d.argmax())
    row = dv[big]
    top = torch.topk(row, 5)
    print('  chain', big, 'top v diffs', top.values.tolist(), top.indices.tolist())
    dx = (torch.remainder(out[True][0] - out[False][0] + np.pi, 2 * np.pi) - np.pi).abs().reshape(nb, -1)
    top = torch.topk(dx[big], 5)
    print('  chain', big, 'top x diffs', top.values.tolist(), top.indices.tolist())
