"""Shared helpers for the parity tests: build oracle networks/dynamics from a golden file."""
import numpy as np

from oracle import network as onet
from oracle.dynamics import DynamicsOracle


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def su3_oracle(g):
    """DynamicsOracle for the su3_l2hmc golden file."""
    w = sub(g, 'vnet.')

    def vnet(step, xv, fv):
        return onet.leapfrog_layer(xv, fv, w, nunits=1, activation='tanh')
    L = tuple(int(i) for i in g['latvolume'])
    return DynamicsOracle('SU3', L, int(g['nleapfrog']), g['xeps'], g['veps'], g['masks'],
                          vnet=vnet)


def u1_net_kwargs(g):
    conv = None
    if g['conv_filters'].size:
        conv = {'filters': [int(i) for i in g['conv_filters']],
                'sizes': [int(i) for i in g['conv_sizes']],
                'pool': [int(i) for i in g['conv_pool']]}
    return dict(nunits=len(g['units']), activation=str(g['activation']), conv=conv,
                use_batch_norm=bool(g['use_batch_norm']))


def u1_oracle(g, dtype=np.float32):
    sd = sub(g, 'sd.')
    kw = u1_net_kwargs(g)
    nlf = int(g['nleapfrog'])

    def vnet(step, x, f):
        return onet.leapfrog_layer(x, f, sub(sd, f'vnet.{step}.'), **kw)

    def xnet(step, first, x, v):
        which = 'first' if first else 'second'
        return onet.leapfrog_layer(x, v, sub(sd, f'xnet.{step}.{which}.'), **kw)
    xeps = [sd[f'xeps.{i}'] for i in range(nlf)]
    veps = [sd[f'veps.{i}'] for i in range(nlf)]
    L = tuple(int(i) for i in g['latvolume'])
    return DynamicsOracle('U1', L, nlf, xeps, veps, g['masks'], vnet=vnet, xnet=xnet,
                          dtype=dtype)


# ------------------------------------------------------------------ product-side builders
def build_su3_dynamics(g, with_nets=True, nb=None, verbose=True):
    """l2hmc.Dynamics (the HIP-backed product) configured like the su3_* golden files."""
    import torch
    import l2hmc.configs as cfgs
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3
    from l2hmc.network.pytorch.network import NetworkFactory
    L = [int(i) for i in g['latvolume']]
    nb = int(g['x'].shape[0]) if nb is None else nb
    nlf = int(g['nleapfrog']) if with_nets else 2
    dc = cfgs.DynamicsConfig(nchains=nb, group='SU3', latvolume=L, nleapfrog=nlf, eps=0.01,
                             eps_hmc=0.01, verbose=verbose, use_split_xnets=False,
                             use_separate_networks=False, merge_directions=True)
    lat = LatticeSU3(nb, L)
    nf = None
    if with_nets:
        nc = cfgs.NetworkConfig(units=[4], activation_fn='tanh', dropout_prob=0.0,
                                use_batch_norm=False)
        V = int(np.prod(L))
        spec = cfgs.InputSpec(xshape=tuple(dc.xshape), xnet={'x': [32 * V], 'v': [32 * V]},
                              vnet={'x': [32 * V], 'v': [32 * V]})
        nf = NetworkFactory(input_spec=spec, network_config=nc,
                            conv_config=cfgs.ConvolutionConfig())
    dyn = Dynamics(potential_fn=lat.action, config=dc, network_factory=nf)
    if with_nets:
        sd = {k: torch.from_numpy(v) for k, v in sub(g, 'vnet.').items()}
        missing, unexpected = dyn.vnet.load_state_dict(sd, strict=True), None
        dyn.set_masks(g['masks'])
        with torch.no_grad():
            for i in range(nlf):
                dyn.xeps[i].copy_(torch.tensor(float(g['xeps'][i])))
                dyn.veps[i].copy_(torch.tensor(float(g['veps'][i])))
    dyn.eval()
    return dyn, lat


def build_u1_dynamics(g, verbose=True):
    import torch
    import l2hmc.configs as cfgs
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.lattice.u1.pytorch.lattice import LatticeU1
    from l2hmc.network.pytorch.network import NetworkFactory
    L = [int(i) for i in g['latvolume']]
    nb = int(g['x'].shape[0])
    nlf = int(g['nleapfrog'])
    dc = cfgs.DynamicsConfig(nchains=nb, group='U1', latvolume=L, nleapfrog=nlf, eps=0.1,
                             eps_hmc=0.1, use_ncp=True, verbose=verbose, use_split_xnets=True,
                             use_separate_networks=True, merge_directions=True)
    kw = u1_net_kwargs(g)
    nc = cfgs.NetworkConfig(units=[int(i) for i in g['units']], activation_fn=kw['activation'],
                            dropout_prob=0.2, use_batch_norm=kw['use_batch_norm'])
    cc = cfgs.ConvolutionConfig(**kw['conv']) if kw['conv'] else cfgs.ConvolutionConfig()
    xdim = dc.xdim
    spec = cfgs.InputSpec(xshape=tuple(dc.xshape), xnet={'x': [xdim, 2], 'v': [xdim]},
                          vnet={'x': [xdim], 'v': [xdim]})
    lat = LatticeU1(nb, L)
    nf = NetworkFactory(input_spec=spec, network_config=nc, conv_config=cc)
    dyn = Dynamics(potential_fn=lat.action, config=dc, network_factory=nf)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sub(g, 'sd.').items()}
    res = dyn.load_state_dict(sd, strict=False)
    assert all(k.startswith('networks.') for k in res.missing_keys), res.missing_keys
    assert not res.unexpected_keys, res.unexpected_keys
    dyn._eps_cache = {}
    dyn.set_masks(g['masks'])
    dyn.eval()
    return dyn, lat
