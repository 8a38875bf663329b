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
