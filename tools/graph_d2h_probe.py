"""Which captured transitions does a >= 512 KB device-to-host copy on the null stream poison?  (round 5: the ones whose launch path contained a hipMemsetAsync -- memset NODES of a replayed graph stop taking effect after such a copy on ROCm 7.0; fixed by csrc/l2q_common.hpp::launch_zero)
CASE = su3_verbose | su3_hmc | u1_fb | u1_hmc"""
import os, sys
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(__file__), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
case = os.environ.get('CASE', 'su3_verbose')
import bench


def poison():
    t = torch.zeros(4 << 20, dtype=torch.uint8, device='cuda')
    h = t.cpu()
    torch.cuda.synchronize()


if case.startswith('su3'):
    sys.argv = ['bench.py', '--nchains', '64']
    args = bench.parse()
    dyn, lat = bench.build(args, 9992)
    x = bench.hot_start(args, seed=1)
    beta = float(args.beta)
    dyn.config.verbose = True
    if 'noheads' in case:
        dyn.sliced_heads = False
    if 'noinput' in case:
        dyn.sliced_input = False
    if 'nopair' in case:
        dyn.pair_v_updates = False
    if 'novec8' in case:
        dyn.fuse_x_vec8 = False
    if case == 'su3_hmc':
        g = dyn.make_graphed(x, beta, mode='hmc', eps=0.01, nleapfrog=4)
    else:
        g = dyn.make_graphed(x, beta)
else:
    import l2hmc.configs as cfgs
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.lattice.u1.pytorch.lattice import LatticeU1
    from l2hmc.network.pytorch.network import NetworkFactory
    torch.set_default_dtype(torch.float32)
    torch.manual_seed(1); np.random.seed(1)
    nb, L = 2048, [16, 16]
    dc = cfgs.DynamicsConfig(nchains=nb, group='U1', latvolume=L, nleapfrog=4, eps=0.1, eps_hmc=0.1, verbose=True)
    nc = cfgs.NetworkConfig(units=[16, 16], activation_fn='relu', dropout_prob=0.0, use_batch_norm=False)
    spec = cfgs.InputSpec(xshape=tuple(dc.xshape), xnet={'x': [dc.xdim, 2], 'v': [dc.xdim]},
                          vnet={'x': [dc.xdim], 'v': [dc.xdim]})
    lat = LatticeU1(nb, L)
    dyn = Dynamics(lat.action, dc, NetworkFactory(spec, nc, cfgs.ConvolutionConfig())).eval()
    x = lat.random()
    beta = 4.0
    g = dyn.make_graphed(x, beta, mode='hmc' if case == 'u1_hmc' else 'fb', eps=0.1 if case == 'u1_hmc' else None,
                         nleapfrog=4 if case == 'u1_hmc' else None)


def show(tag):
    for _ in range(2):
        xo, m = g(x)
    e = m['energy']                                  # [steps + 1, nb]
    fin = torch.isfinite(e).all(dim=1).cpu().tolist()
    print(f'[{case}] {tag:7s} acc finite {bool(torch.isfinite(m["acc"]).all())} x_out finite {bool(torch.isfinite(xo).all())} '
          f'energy rows finite {fin}', flush=True)


show('before')
poison()
show('after')
