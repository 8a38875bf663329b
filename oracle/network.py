"""numpy oracle: the xnet / vnet leapfrog networks (eval mode).  TEST INFRASTRUCTURE ONLY.

Restates ``network/pytorch/network.py`` of the reference: ``LeapfrogLayer.forward``
(:522-551), ``InputLayer.forward`` (:430-451), ``ConvStack`` (:240-346),
``PeriodicPadding`` (:151-172), ``ScaledTanh`` (:175-206).  Weights are given as a dict
keyed like the reference ``state_dict`` of one LeapfrogLayer.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np


def act(name: str, z: np.ndarray) -> np.ndarray:
    """ACTIVATION_FNS, network/pytorch/network.py:40-46"""
    if name == 'tanh':
        return np.tanh(z)
    if name == 'relu':
        return np.maximum(z, 0)
    if name == 'leaky_relu':
        return np.where(z > 0, z, z * z.dtype.type(0.01))
    if name == 'elu':
        return np.where(z > 0, z, np.expm1(np.minimum(z, 0)))
    if name == 'swish':
        return z / (1.0 + np.exp(-z))
    raise ValueError(name)


def linear(z, w, b):
    return z @ w.T + b


def periodic_pad(x, size):
    """network/pytorch/network.py:158-172"""
    if size == 0:
        return x
    x = np.concatenate([x[:, :, -size:, :], x, x[:, :, :size, :]], axis=2)
    x = np.concatenate([x[:, :, :, -size:], x, x[:, :, :, :size]], axis=3)
    return x


def conv2d_valid(x, w, b):
    """cross-correlation, stride 1, no padding: x[nb,C,H,W], w[F,C,kh,kw]"""
    nb, c, h, wd = x.shape
    f, _, kh, kw = w.shape
    ho, wo = h - kh + 1, wd - kw + 1
    xt = np.ascontiguousarray(x.transpose(0, 2, 3, 1))             # [nb, H, W, C]
    out = np.zeros((nb, ho, wo, f), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            out += xt[:, i:i + ho, j:j + wo, :] @ w[:, :, i, j].T    # channel sum on BLAS
    return out.transpose(0, 3, 1, 2) + b[None, :, None, None]


def maxpool2d(x, p):
    nb, c, h, w = x.shape
    ho, wo = h // p, w // p
    return x[:, :, :ho * p, :wo * p].reshape(nb, c, ho, p, wo, p).max((3, 5))


def conv_stack(x, wts, prefix, filters, sizes, pool, activation):
    """ConvStack.forward.  Layer numbering follows the reference ModuleList
    (network/pytorch/network.py:283-326)."""
    li = 0
    x = periodic_pad(x, sizes[0] - 1)
    li += 1
    x = conv2d_valid(x, wts[f'{prefix}layers.{li}.weight'], wts[f'{prefix}layers.{li}.bias'])
    li += 1
    for idx, (f, n) in enumerate(zip(filters[1:], sizes[1:])):
        x = periodic_pad(x, n - 1)
        li += 1
        x = conv2d_valid(x, wts[f'{prefix}layers.{li}.weight'],
                         wts[f'{prefix}layers.{li}.bias'])
        li += 1
        if (idx + 1) % 2 == 0:
            p = 2 if pool is None else pool[idx]
            x = maxpool2d(x, p)
            li += 1
        x = act(activation, x)
        li += 1
    x = x.reshape(x.shape[0], -1)
    li += 1                                           # Flatten
    x = linear(x, wts[f'{prefix}layers.{li}.weight'], wts[f'{prefix}layers.{li}.bias'])
    return act(activation, x)


def leapfrog_layer(
        x: np.ndarray,
        v: np.ndarray,
        wts: dict,
        *,
        nunits: int,
        activation: str,
        nw=(1.0, 1.0, 1.0),
        conv: Optional[dict] = None,
        use_batch_norm: bool = False,
        bn_eps: float = 1e-5,
):
    """(s, t, q) = LeapfrogLayer((x, v)) in eval mode.

    x is the already prepared network input (U1 xnet: [nb,4,T,X]; U1 vnet: [nb,2,T,X];
    SU3 vnet: [nb,4,T,X,Y,Z,8]); v likewise.  ``nunits`` = len(network_config.units).
    """
    if conv is not None and conv.get('filters'):
        x = conv_stack(x, wts, 'input_layer.conv_stack.', conv['filters'], conv['sizes'],
                       conv.get('pool'), activation)
    xf = x.reshape(x.shape[0], -1)
    vf = v.reshape(v.shape[0], -1)
    z = act(activation,
            linear(xf, wts['input_layer.xlayer.weight'], wts['input_layer.xlayer.bias'])
            + linear(vf, wts['input_layer.vlayer.weight'], wts['input_layer.vlayer.bias']))
    for i in range(nunits - 1):
        z = act(activation, linear(z, wts[f'hidden_layers.{i}.weight'],
                                   wts[f'hidden_layers.{i}.bias']))
    if use_batch_norm:
        z = ((z - wts['batch_norm.running_mean'])
             / np.sqrt(wts['batch_norm.running_var'] + bn_eps)
             * wts['batch_norm.weight'] + wts['batch_norm.bias'])
    s = nw[0] * np.exp(wts['scale.coeff']) * np.tanh(
        linear(z, wts['scale.layer.weight'], wts['scale.layer.bias']))
    t = nw[1] * linear(z, wts['transl.weight'], wts['transl.bias'])
    q = nw[2] * np.exp(wts['transf.coeff']) * np.tanh(
        linear(z, wts['transf.layer.weight'], wts['transf.layer.bias']))
    return s, t, q
