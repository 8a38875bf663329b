"""xnet / vnet leapfrog networks -- API and ``state_dict`` key layout of the reference's
``src/l2hmc/network/pytorch/network.py`` (PeriodicPadding :151-172, ScaledTanh :175-206,
ConvStack :240-346, InputLayer :349-451, LeapfrogLayer :454-551, NetworkFactory :634-801).

The modules only *hold* parameters (so ``state_dict`` / ``parameters()`` / checkpoints look
like the reference's); the arithmetic of ``forward`` is the MFMA GEMM with fused
bias + activation + ScaledTanh epilogues (``l2q_gemm_f64`` / ``l2q_gemm_f32``) and the fused
periodic-pad conv + max-pool kernel (``l2q_conv2d_periodic_f32``).  Eval-mode semantics
(dropout off, batch-norm running statistics) -- the training path is SURVEY.md 8(f) item 1.
"""
from __future__ import annotations

import logging
from typing import Any, Callable, Optional, Sequence

import numpy as np
import torch
from torch import nn

from l2hmc import DEVICE
from l2hmc import _ops as ops
from l2hmc.configs import ConvolutionConfig, NetWeight, NetworkConfig
from l2hmc.group.su3.pytorch.group import SU3
from l2hmc.group.u1.pytorch.group import U1Phase
from l2hmc.network.factory import BaseNetworkFactory

log = logging.getLogger(__name__)
Tensor = torch.Tensor

ACTIVATION_FNS = {
    'elu': nn.ELU(inplace=True),
    'tanh': nn.Tanh(),
    'relu': nn.ReLU(inplace=True),
    'swish': nn.SiLU(),
    'leaky_relu': nn.LeakyReLU(inplace=True),
}
_ACT_NAME = {nn.ELU: 'elu', nn.Tanh: 'tanh', nn.ReLU: 'relu', nn.SiLU: 'swish',
             nn.LeakyReLU: 'leaky_relu'}


def act_name(fn: Any) -> str:
    if isinstance(fn, str):
        return fn
    for cls, name in _ACT_NAME.items():
        if isinstance(fn, cls):
            return name
    raise ValueError(f'unsupported activation {fn}')


def flatten(x: Tensor) -> Tensor:
    return x.reshape(x.shape[0], -1)


def nested_children(m: nn.Module) -> dict:
    """{name: nested dict of children | leaf module} (network.py:49-57)"""
    children = dict(m.named_children())
    if not children:
        return {m._get_name(): m}
    return {name: nested_children(child) if isinstance(child, nn.Module) else child
            for name, child in children.items()}


def xy_repr(x: Tensor) -> Tensor:
    return torch.stack((torch.cos(x), torch.sin(x)), dim=1)


def init_all(model: nn.Module, init_func: Callable, *params, **kwargs) -> None:
    """init_func(p, *params, **kwargs) on every parameter (network.py:80-90)"""
    for p in model.parameters():
        init_func(p, *params, **kwargs)


def init_all_by_shape(model: nn.Module, init_funcs: dict) -> None:
    """Pick the initialiser by tensor rank: init_funcs[str(rank)] or init_funcs['default']
    (network.py:93-118)."""
    assert 'default' in init_funcs, 'init_funcs must have `default` entry'
    for p in model.parameters():
        if hasattr(p, 'shape'):
            init_funcs.get(str(len(p.shape)), init_funcs['default'])(p)


def init_weights(m: nn.Module, method: str = 'xavier_uniform') -> None:
    """`model.apply(init_weights)` helper for nn.Linear layers (network.py:121-141)."""
    if not isinstance(m, nn.Linear):
        return
    if method == 'zeros':
        nn.init.zeros_(m.weight)
        nn.init.zeros_(m.bias)
        return
    name = method if method.endswith('_') else method + '_'
    fn = getattr(nn.init, name, None) or getattr(nn.init, method, None)
    if callable(fn):
        fn(m.weight)
    else:
        log.warning(f'Unable to initialize weights with {method}; keeping the default')


def calc_output_size(hw: tuple[int, int], kernel_size, stride: int = 1, pad: int = 0,
                     dilation: int = 1) -> tuple[int, int]:
    """Conv2d output extent (network.py:209-236)"""
    from math import floor
    if isinstance(kernel_size, int):
        kernel_size = (kernel_size, kernel_size)
    h = floor(1 + (hw[0] + 2 * pad - dilation * (kernel_size[0] - 1) - 1) / stride)
    w = floor(1 + (hw[1] + 2 * pad - dilation * (kernel_size[1] - 1) - 1) / stride)
    return h, w


def dummy_network(inputs: tuple[Tensor, Tensor]) -> tuple[Tensor, Tensor, Tensor]:
    x, _ = inputs
    return torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)


def zero_weights(m):
    if isinstance(m, nn.Linear):
        nn.init.zeros_(m.weight.data)
        if m.bias is not None:
            nn.init.constant_(m.bias.data, 0)


# ---------------------------------------------------------------- seed-level initialisation
# The reference builds its networks on the host and consumes the GLOBAL torch generator in an
# order that the results depend on ("same (lattice, beta, seed) -> same chain"):
#   get_and_call_network (network.py:572-631): group.random(xshape), group.random_momentum(xshape),
#   then LeapfrogLayer.__init__ (:454-516): hidden Linear layers, scale, transf, transl are
#   initialised at construction, whereas the conv stack, its Linear, vlayer and xlayer are Lazy
#   modules that draw their weights at the first forward -- conv layers in order, the stack's
#   Linear, then vlayer BEFORE xlayer (InputLayer.forward :449-450) -- and that dummy forward runs
#   in train mode, so Dropout draws one mask and BatchNorm1d takes one running-statistics step.
# Here every parameter is drawn on the host generator in that order and then moved to DEVICE.
def _uninit(cls, *args, **kw) -> nn.Module:
    """`cls(*args)` with uninitialised host parameters and no generator use: the stand-in for
    the reference's Lazy* layers until `reset_parameters()` runs at their first-forward slot."""
    return torch.nn.utils.skip_init(cls, *args, device='cpu', **kw)


_BURN_CHUNK = 1 << 24        # multiple of 16: torch's CPU normal fill transforms groups of 16


def advance_randn(numel: int, dtype: Optional[torch.dtype] = None) -> None:
    """Advance the global host generator exactly like ``torch.randn(numel, dtype=dtype)`` without
    holding the values: chunks that are multiples of 16 draw the same uniforms in the same
    groups as one call (ATen normal_fill: all uniforms first, Box-Muller per 16; a tail that is
    not a multiple of 16 redraws the last 16 -- kept inside the final chunk)."""
    dtype = dtype or torch.get_default_dtype()
    left = int(numel)
    while left > 0:
        n = left if left < 2 * _BURN_CHUNK else _BURN_CHUNK
        torch.randn(n, dtype=dtype)
        left -= n


def advance_rand(numel: int, dtype: Optional[torch.dtype] = None) -> None:
    """Same for ``torch.rand`` (serial per element on the host: any chunking is equivalent)."""
    dtype = dtype or torch.get_default_dtype()
    left = int(numel)
    while left > 0:
        n = min(left, _BURN_CHUNK)
        torch.rand(n, dtype=dtype)
        left -= n


def _linear_bwd(dpre: Tensor, x: Tensor, weight, bias,
                need_dx: bool = True, xT: Optional[Tensor] = None,
                wgrad: Optional[Tensor] = None, bgrad: Optional[Tensor] = None,
                bias_done: bool = False, skip_dw: bool = False) -> Optional[Tensor]:
    """Backward of y = x W^T + b given dpre = dL/dy: W.grad += dpre^T x, b.grad += colsum(dpre),
    returns dL/dx = dpre W (MFMA GEMMs on transposed operands; accumulation into the flat
    gradient arena the parameters' .grad are views of)."""
    # both operands are read transposed in the GEMM's tile loader and the result is added into
    # the gradient arena by its epilogue: no transposed copies, no separate accumulation pass
    # (rows that are not 16-byte multiples fall back to explicit transposes inside gemm_ex)
    # wgrad / bgrad: accumulate there instead of weight.grad / bias.grad (native-order shadows of
    # the SU(3) vnet, LeapfrogLayer.native_train_begin)
    # skip_dw: the caller has parked (dpre, x) in the tape's arenas and forms W.grad for all network calls of
    # the step in ONE GEMM afterwards (LeapfrogLayer.flush_deferred)
    if not skip_dw:
        ops.gemm_ex(dpre, x, a_trans=True, w_trans=True,
                    out=weight.grad if wgrad is None else wgrad, accumulate=True)
    if bias is not None and not bias_done:
        ops.colsum_(bias.grad if bgrad is None else bgrad, dpre)
    if not need_dx:
        return None
    return ops.gemm_ex(dpre, weight.detach(), w_trans=True)


class PeriodicPadding(nn.Module):
    """Parameter-free; kept so that ConvStack.layers has the reference's numbering.  Inside
    ConvStack the padding is fused into the conv kernel (index arithmetic, no copy)."""

    def __init__(self, size: int):
        super().__init__()
        self.size = size

    def forward(self, x: Tensor) -> Tensor:
        x = torch.concat([x[:, :, -self.size:, :], x, x[:, :, 0:self.size, :]], 2)
        return torch.concat([x[:, :, :, -self.size:], x, x[:, :, :, 0:self.size]], 3)


class ScaledTanh(nn.Module):
    """exp(coeff) * tanh(W z + b)  (network.py:175-206) -- one GEMM with fused epilogue."""

    def __init__(self, in_features: int, out_features: int) -> None:
        super().__init__()
        self.coeff = nn.parameter.Parameter(torch.zeros(1, out_features))
        self.layer = nn.Linear(in_features=in_features, out_features=out_features,
                               device='cpu')                  # host generator (see above)
        self.to(DEVICE)

    def forward(self, x):
        return ops.gemm(x.to(DEVICE).contiguous(), self.layer.weight.detach(),
                        self.layer.bias.detach(), coeff=self.coeff.detach().reshape(-1),
                        act='tanh')


class ConvStack(nn.Module):
    def __init__(self, xshape: Sequence[int], conv_config: ConvolutionConfig,
                 activation_fn: Any, use_batch_norm: bool = False,
                 in_channels: Optional[int] = None, lazy: bool = False) -> None:
        """lazy=True leaves the parameters uninitialised for the owner to `materialize()` at
        the reference's first-forward slot; stand-alone stacks draw them here."""
        super().__init__()
        if len(xshape) == 3:
            d, nt, nx = xshape[0], xshape[1], xshape[2]
        elif len(xshape) == 4:
            _, d, nt, nx = xshape
        else:
            raise ValueError(f'ConvStack is 2D only (U1); got xshape {xshape}')
        self.d, self.nt, self.nx = d, nt, nx
        self.xshape = xshape
        self.xdim = int(np.cumprod(xshape[1:])[-1])
        self.activation_fn = activation_fn
        self.act = act_name(activation_fn)
        self.layers = nn.ModuleList()
        cin = (d + 2) if in_channels is None else in_channels
        self.in_channels = cin
        self.plan = []                       # (layer index of conv, k, pool, act)
        self._wcache: dict = {}
        h, w = nt, nx
        filters = list(conv_config.filters or [])
        sizes = list(conv_config.sizes or [])
        if filters:
            assert len(filters) == len(sizes)
            self.layers.append(PeriodicPadding(sizes[0] - 1))
            self.layers.append(_uninit(nn.Conv2d, cin, filters[0], sizes[0]))
            self.plan.append((1, sizes[0], 1, None))          # no activation after conv #1
            h, w, cin = h + sizes[0] - 1, w + sizes[0] - 1, filters[0]
            for idx, (f, n) in enumerate(zip(filters[1:], sizes[1:])):
                self.layers.append(PeriodicPadding(n - 1))
                self.layers.append(_uninit(nn.Conv2d, cin, f, n))
                ci = len(self.layers) - 1
                pool = 1
                if (idx + 1) % 2 == 0:
                    pool = 2 if conv_config.pool is None else int(conv_config.pool[idx])
                    self.layers.append(nn.MaxPool2d(pool))
                self.layers.append(self.activation_fn)
                self.plan.append((ci, n, pool, self.act))
                h, w, cin = (h + n - 1) // pool, (w + n - 1) // pool, f
        self.layers.append(nn.Flatten())
        if use_batch_norm:
            raise NotImplementedError('ConvStack(use_batch_norm=True) is never used by the '
                                      'reference (network.py:407-411)')
        self.layers.append(_uninit(nn.Linear, cin * h * w, self.xdim))
        self.linear_index = len(self.layers) - 1
        self.layers.append(self.activation_fn)
        if not lazy:
            self.materialize()
            self.to(DEVICE)

    def materialize(self) -> None:
        """Draw the Lazy layers' weights in first-forward order (network.py:341-344)."""
        for layer in self.layers:
            if isinstance(layer, (nn.Conv2d, nn.Linear)):
                layer.reset_parameters()

    def _clast_weight(self, ci: int) -> Tensor:
        """conv weight as [cout, k, k, cin] (K columns of the implicit GEMM in the order in which
        an NHWC activation is contiguous), cached per weight version."""
        conv = self.layers[ci]
        ver = (conv.weight._version, ops.PARAM_GENERATION[0])
        hit = self._wcache.get(ci)
        if hit is None or hit[0] != ver:
            w = conv.weight.detach().permute(0, 2, 3, 1)
            if ci == self.plan[0][0] and w.shape[-1] % 4:     # first layer: lattice channels
                # padded to a 16-byte group (its input is padded alike in forward)
                w = torch.nn.functional.pad(w, (0, 4 - w.shape[-1] % 4))
            hit = (ver, w.contiguous())
            self._wcache[ci] = hit
        return hit[1]

    def _half_weights(self, hd: torch.dtype) -> dict:
        """16-bit copies for the half-precision path, cached per weight version: conv weights in
        the K order of their input layout, biases rounded like autocast does, and the Linear
        weight with its input columns permuted from the reference's NCHW flatten order to NHWC
        (so the activations need no transposition)."""
        lin = self.layers[self.linear_index]
        ver = tuple(self.layers[ci].weight._version for ci, *_ in self.plan) + \
            (lin.weight._version, lin.bias._version, ops.PARAM_GENERATION[0], hd)
        hit = self._wcache.get('half')
        if hit is not None and hit['ver'] == ver:
            return hit
        r16 = lambda t: t.detach().to(hd).float().contiguous()
        convs = []
        for n, (ci, k, pool, act) in enumerate(self.plan):
            w = self.layers[ci].weight.detach()
            w16 = w.permute(0, 2, 3, 1).to(hd).contiguous()              # [cout, k, k, cin]
            if n == 0 and w16.shape[-1] % 8:
                # the first layer sees the 2 / 4 lattice channels padded to 8 (zeros): one
                # 16-byte gather per tap instead of cin scalar ones
                pad = 8 - w16.shape[-1] % 8
                w16 = torch.nn.functional.pad(w16, (0, pad)).contiguous()
            convs.append((w16, r16(self.layers[ci].bias)))
        h, w_ = self.nt, self.nx
        c = self.in_channels
        for ci, k, pool, act in self.plan:
            h, w_, c = (h + k - 1) // pool, (w_ + k - 1) // pool, self.layers[ci].out_channels
        wl = lin.weight.detach()
        if self.plan:                               # [out, C, H, W] -> [out, H, W, C]
            wl = wl.reshape(wl.shape[0], c, h, w_).permute(0, 2, 3, 1).reshape(wl.shape[0], -1)
        hit = {'ver': ver, 'convs': convs, 'lin': (wl.to(hd).contiguous(), r16(lin.bias))}
        self._wcache['half'] = hit
        return hit

    def _forward_half(self, x: Tensor, hd: torch.dtype) -> Tensor:
        """autocast's view of this stack: Conv2d / Linear in 16 bit, fp32 accumulation.  Returns
        the fp32 container of the 16-bit result (it is the fp32-typed `x` input of the
        LeapfrogLayer's first GEMM, which rounds on load: exact)."""
        hw = self._half_weights(hd)
        if self.plan:
            x = ops.nchw_to_nhwc_pad_h(x, hd, hw['convs'][0][0].shape[-1])
        for (ci, k, pool, act), (w16, b) in zip(self.plan, hw['convs']):
            x = ops.conv2d_periodic_gemm_h(x, 'nhwc', w16, b, pool, act)
        wl, bl = hw['lin']
        return ops.gemm_h(x.reshape(x.shape[0], -1), wl, bl, act=self.act, out_dtype=torch.float32)

    def forward(self, x: Tensor) -> Tensor:
        x = x.to(DEVICE)
        x = x.reshape(x.shape[0], self.in_channels, self.nt, self.nx).contiguous()
        if x.dtype != torch.float32:
            raise NotImplementedError('the conv kernels are fp32 (the U(1) configs)')
        if getattr(self, 'half_dtype', None) is not None:
            return self._forward_half(x, self.half_dtype)
        # implicit GEMM: periodic im2col -> f32 MFMA GEMM (NHWC activations) -> pool + act
        layout = 'nchw'
        if self.plan:                 # NHWC from the start (channels padded to 4): vector gathers
            x = ops.nchw_to_nhwc_pad(x, self._clast_weight(self.plan[0][0]).shape[-1])
            layout = 'nhwc'
        for ci, k, pool, act in self.plan:
            conv = self.layers[ci]
            wc = self._clast_weight(ci)
            x = ops.conv2d_periodic_gemm(x, layout, wc.permute(0, 3, 1, 2), conv.bias.detach(),
                                         pool, act, w_clast=wc)
            layout = 'nhwc'
        if layout == 'nhwc':                       # reference flattens NCHW
            nb, H, W, C = x.shape
            x = ops.transpose(x.reshape(nb, H * W, C), nb, H * W, C)
        lin = self.layers[self.linear_index]
        return ops.gemm(x.reshape(x.shape[0], -1).contiguous(), lin.weight.detach(),
                        lin.bias.detach(), act=self.act)

    # ---- training path: forward keeping what the backward needs, and the backward itself
    def forward_train(self, x: Tensor, half: Optional[torch.dtype] = None) -> tuple[Tensor, dict]:
        """half: the stack under torch.autocast(dtype=half) -- every Conv2d / Linear with 16-bit operands
        and outputs (ops.conv2d_periodic_gemm_train's `half`), values kept in fp32 containers for the tape."""
        x = x.to(DEVICE)
        nb = x.shape[0]
        x = x.reshape(nb, self.in_channels, self.nt, self.nx).contiguous()
        if x.dtype != torch.float32:
            raise NotImplementedError('the conv kernels are fp32 (the U(1) configs)')
        layout, ctxs = 'nchw', []
        for ci, k, pool, act in self.plan:
            conv = self.layers[ci]
            x, c = ops.conv2d_periodic_gemm_train(x, layout, conv.weight.detach(),
                                                  conv.bias.detach(), pool, act, half=half)
            ctxs.append(c)
            layout = 'nhwc'
        nhwc_shape = None
        if layout == 'nhwc':
            nhwc_shape = tuple(x.shape)
            _, H, W, C = x.shape
            x = ops.transpose(x.reshape(nb, H * W, C), nb, H * W, C)
        flat = x.reshape(nb, -1).contiguous()
        lin = self.layers[self.linear_index]
        if half is not None:
            y = ops.gemm_h(flat, lin.weight.detach().to(half).contiguous(),
                           lin.bias.detach().to(half).float().contiguous(), act=self.act,
                           out_dtype=torch.float32)
        else:
            y = ops.gemm(flat, lin.weight.detach(), lin.bias.detach(), act=self.act)
        return y, {'convs': ctxs, 'nhwc_shape': nhwc_shape, 'flat': flat, 'y': y}

    def backward(self, ctx: dict, dy: Tensor) -> Tensor:
        """Accumulates the layers' .grad; returns dL/dx [nb, C, T, X]."""
        lin = self.layers[self.linear_index]
        if self.act == 'swish':
            raise NotImplementedError('ConvStack.backward with swish: the tape holds post-activations')
        dpre = ops.act_bwd(dy.contiguous().clone(), ctx['y'], self.act)
        d = _linear_bwd(dpre, ctx['flat'], lin.weight, lin.bias)
        nb = d.shape[0]
        if ctx['nhwc_shape'] is not None:
            _, H, W, C = ctx['nhwc_shape']
            d = ops.transpose(d.reshape(nb, C, H * W), nb, C, H * W).reshape(nb, H, W, C)
        for (ci, k, pool, act), c in zip(reversed(self.plan), reversed(ctx['convs'])):
            conv = self.layers[ci]
            d = ops.conv2d_periodic_gemm_bwd(c, d, conv.weight.detach(), conv.weight.grad,
                                             conv.bias.grad)
        return d


class InputLayer(nn.Module):
    def __init__(self, xshape: Sequence[int], network_config: NetworkConfig,
                 activation_fn: Callable[[Tensor], Tensor],
                 conv_config: Optional[ConvolutionConfig] = None,
                 input_shapes: Optional[dict[str, Sequence[int] | int]] = None,
                 x_features: Optional[int] = None, v_features: Optional[int] = None,
                 conv_channels: Optional[int] = None, lazy: bool = False) -> None:
        super().__init__()
        self.xshape = xshape
        self.net_config = network_config
        self.units = self.net_config.units
        self.xdim = int(np.cumprod(self.xshape[1:])[-1])
        if input_shapes is None:
            input_shapes = {'x': self.xdim, 'v': self.xdim}
        self.input_shapes = {k: (v if isinstance(v, int) else int(np.cumprod(v)[-1]))
                             for k, v in input_shapes.items()}
        self.conv_config = conv_config
        self.activation_fn = activation_fn
        self.act = act_name(activation_fn)
        conv_stack: nn.Module = nn.Identity()
        if conv_config is not None and conv_config.filters is not None \
                and len(conv_config.filters) > 0:
            conv_stack = ConvStack(xshape=xshape, conv_config=conv_config,
                                   activation_fn=self.activation_fn,
                                   in_channels=conv_channels, lazy=True)
        self.conv_stack = conv_stack
        has_conv = isinstance(conv_stack, ConvStack)
        xin = self.xdim if has_conv else (x_features or self.input_shapes['x'])
        vin = v_features or self.input_shapes['v']
        self.xlayer = _uninit(nn.Linear, xin, self.net_config.units[0])
        self.vlayer = _uninit(nn.Linear, vin, self.net_config.units[0])
        if not lazy:
            self.materialize()
            self.to(DEVICE)

    def materialize(self) -> None:
        """First-forward order of the reference's Lazy layers: the conv stack, then vlayer, then
        xlayer (network.py:446-450)."""
        if isinstance(self.conv_stack, ConvStack):
            self.conv_stack.materialize()
        self.vlayer.reset_parameters()
        self.xlayer.reset_parameters()

    def forward(self, inputs: tuple[Tensor, Tensor]) -> Tensor:
        x, v = inputs
        x, v = x.to(DEVICE), v.to(DEVICE)
        if isinstance(self.conv_stack, ConvStack):
            x = self.conv_stack(x)
        return ops.gemm(flatten(x).contiguous(), self.xlayer.weight.detach(),
                        self.xlayer.bias.detach(), a2=flatten(v).contiguous(),
                        w2=self.vlayer.weight.detach(), bias2=self.vlayer.bias.detach(),
                        act=self.act)


class LeapfrogLayer(nn.Module):
    def __init__(self, xshape: Sequence[int], network_config: NetworkConfig,
                 input_shapes: Optional[dict[str, int | Sequence[int]]] = None,
                 net_weight: Optional[NetWeight] = None,
                 conv_config: Optional[ConvolutionConfig] = None,
                 name: Optional[str] = None, x_features: Optional[int] = None,
                 v_features: Optional[int] = None, conv_channels: Optional[int] = None):
        super().__init__()
        if net_weight is None:
            net_weight = NetWeight(1., 1., 1.)
        self.xshape = xshape
        self.nw = net_weight
        self.net_config = network_config
        self.name = name if name is not None else 'network'
        self.xdim = int(np.cumprod(xshape[1:])[-1])
        act_fn = self.net_config.activation_fn
        if isinstance(act_fn, str):
            act_fn = ACTIVATION_FNS.get(act_fn, None)
        assert isinstance(act_fn, Callable)
        self.activation_fn = act_fn
        self.act = act_name(act_fn)
        self.input_layer = InputLayer(
            xshape=xshape, network_config=network_config, activation_fn=self.activation_fn,
            conv_config=conv_config, input_shapes=input_shapes, x_features=x_features,
            v_features=v_features, conv_channels=conv_channels, lazy=True)
        self.units = self.net_config.units
        self.hidden_layers = nn.ModuleList()
        for idx, units in enumerate(self.units[1:]):
            self.hidden_layers.append(nn.Linear(self.units[idx], units, device='cpu'))
        self.scale = ScaledTanh(self.units[-1], self.xdim)
        self.transf = ScaledTanh(self.units[-1], self.xdim)
        self.transl = nn.Linear(self.units[-1], self.xdim, device='cpu')
        self.dropout = nn.Dropout(self.net_config.dropout_prob)
        if self.net_config.use_batch_norm:
            self.batch_norm = nn.BatchNorm1d(self.units[-1])
        # the reference's Lazy input layers draw their weights at the first forward, i.e. after
        # everything above (nothing else touches the generator in between)
        self.input_layer.materialize()
        self.to(DEVICE)
        self._head_cache: dict = {}

    def set_net_weight(self, net_weight: NetWeight):
        self.nw = net_weight

    # -------------------------------------------------------------- weights for the kernels
    def _versions(self):
        vs = [p._version for p in self.parameters()]
        vs += [b._version for b in self.buffers()]
        return tuple(vs) + (self.nw.s, self.nw.t, self.nw.q, ops.PARAM_GENERATION[0],
                            getattr(self, 'half_dtype', None))

    def set_precision(self, half: Optional[torch.dtype]) -> None:
        """half = torch.float16 | torch.bfloat16: the Linear layers run on the 16-bit MFMA path
        with autocast's rounding points (what the reference gets from torch.autocast around
        Dynamics.forward, trainers/pytorch/trainer.py:211-219); None: full precision."""
        if half not in (None, torch.float16, torch.bfloat16):
            raise ValueError(f'LeapfrogLayer.set_precision: {half}')
        if half is not None and self.transl.weight.dtype != torch.float32:
            raise ValueError('half-precision layers need fp32 master weights (the U(1) configs)')
        self.half_dtype = half
        if isinstance(self.input_layer.conv_stack, ConvStack):
            self.input_layer.conv_stack.half_dtype = half

    def kernel_weights(self, in_perm: Optional[Tensor] = None,
                       out_perm: Optional[Tensor] = None) -> dict:
        """Contiguous (optionally permuted) weight copies the GEMM kernels read.

        in_perm / out_perm: ``idx[j_native] = j_reference`` maps (SU(3) native layout, see
        include/l2q.h) applied to the input columns of xlayer/vlayer and to the output rows of
        the three heads.  Eval-mode BatchNorm1d is folded into the heads:
        W (a*z + c) + b = (W diag(a)) z + (W c + b).  Rebuilt when any parameter changes."""
        key = (None if in_perm is None else in_perm.data_ptr(),
               None if out_perm is None else out_perm.data_ptr())
        ver = self._versions()
        hit = self._head_cache.get(key)
        if hit is not None and hit['ver'] == ver:
            return hit
        with torch.no_grad():
            il = self.input_layer
            wx, wv = il.xlayer.weight, il.vlayer.weight
            if in_perm is not None and not isinstance(il.conv_stack, ConvStack):
                wx = wx[:, in_perm]
            if in_perm is not None:
                wv = wv[:, in_perm]
            out = {'ver': ver, 'wx': wx.contiguous(), 'bx': il.xlayer.bias.contiguous(),
                   'wv': wv.contiguous(), 'bv': il.vlayer.bias.contiguous(),
                   'hidden': [(h.weight.contiguous(), h.bias.contiguous())
                              for h in self.hidden_layers]}
            heads = {}
            for nm, lin, coeff in (('s', self.scale.layer, self.scale.coeff),
                                   ('t', self.transl, None),
                                   ('q', self.transf.layer, self.transf.coeff)):
                w, b = lin.weight, lin.bias
                if self.net_config.use_batch_norm:
                    bn = self.batch_norm
                    a = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    c = bn.bias - bn.running_mean * a
                    b = b + w @ c
                    w = w * a[None, :]
                co = None if coeff is None else coeff.reshape(-1)
                if out_perm is not None:
                    w, b = w[out_perm], b[out_perm]
                    co = None if co is None else co[out_perm]
                heads[nm] = (w.contiguous(), b.contiguous(),
                             None if co is None else co.contiguous())
            out['heads'] = heads
            # per-entry multipliers nw * exp(coeff) for the fused heads + v-update kernel
            out['heads_scaled'] = {
                's': (heads['s'][0], heads['s'][1], (self.nw.s * heads['s'][2].exp()).contiguous()),
                't': (heads['t'][0], heads['t'][1], None),
                'q': (heads['q'][0], heads['q'][1], (self.nw.q * heads['q'][2].exp()).contiguous()),
            }
            if in_perm is None and out_perm is None and wx.dtype == torch.float32 \
                    and not isinstance(il.conv_stack, ConvStack) \
                    and max(self.units) <= 64 and len(self.units) <= 8:
                # layout of the fused U(1) sub-update kernels (l2q_u1_vstep_f32 / _xstep_f32)
                import ctypes
                hidden = [torch.cat([h.weight.reshape(-1), h.bias.reshape(-1)])
                          for h in self.hidden_layers]
                hs = out['heads_scaled']
                ones = torch.ones_like(hs['t'][1])
                out['fused_u1'] = {
                    'wxT': wx.t().contiguous(), 'wvT': wv.t().contiguous(),
                    'b0': (il.xlayer.bias + il.vlayer.bias).contiguous(),
                    'hidden': torch.cat(hidden).contiguous() if hidden else None,
                    'units_c': (ctypes.c_int * len(self.units))(*[int(u) for u in self.units]),
                    'nl': len(self.units), 'act': self.act, 'scale_t': float(self.nw.t),
                    'heads': {'s': (hs['s'][0], hs['s'][1], hs['s'][2]),
                              't': (hs['t'][0], hs['t'][1], ones),
                              'q': (hs['q'][0], hs['q'][1], hs['q'][2])}}
            if getattr(self, 'half_dtype', None) is not None:
                hd = self.half_dtype
                r16 = lambda t: t.to(hd).float().contiguous()     # autocast casts the bias too
                out['h'] = {
                    'wx': out['wx'].to(hd), 'wv': out['wv'].to(hd),
                    'bx': r16(out['bx']), 'bv': r16(out['bv']),
                    'hidden': [(hw.to(hd), r16(hb)) for hw, hb in out['hidden']],
                    'heads': {k: (w_.to(hd), r16(b_), c_) for k, (w_, b_, c_) in heads.items()},
                    # per-entry fp32 multipliers of the fused heads + update kernel
                    'heads_scaled': {k: (heads[k][0].to(hd), r16(heads[k][1]),
                                         out['heads_scaled'][k][2]) for k in ('s', 't', 'q')}}
        self._head_cache[key] = out
        return out

    def _check_mode(self):
        if self.training and (self.net_config.dropout_prob > 0 or self.net_config.use_batch_norm):
            raise NotImplementedError(
                'LeapfrogLayer: train-mode dropout / batch-norm statistics belong to the training '
                'entry points (forward_train / Trainer.train_step); this is the eval-mode forward '
                '-- call .eval()')

    def forward_flat(self, x: Tensor, v: Tensor, w: Optional[dict] = None
                     ) -> tuple[Tensor, Tensor, Tensor]:
        """(s, t, q) from already flattened inputs [nb, Kx], [nb, Kv] (conv stack, if any,
        already applied to x)."""
        self._check_mode()
        w = self.kernel_weights() if w is None else w
        if getattr(self, 'half_dtype', None) is not None:
            h = w['h']
            z = ops.gemm_h(x, h['wx'], h['bx'], a2=v, w2=h['wv'], bias2=h['bv'], act=self.act)
            for hw, hb in h['hidden']:
                z = ops.gemm_h(z, hw, hb, act=self.act)
            ws, bs, cs = h['heads']['s']
            wt, bt, _ = h['heads']['t']
            wq, bq, cq = h['heads']['q']
            f32 = torch.float32
            s = ops.gemm_h(z, ws, bs, coeff=cs, scale=self.nw.s, act='tanh', out_dtype=f32)
            t = ops.gemm_h(z, wt, bt, scale=self.nw.t, out_dtype=f32)
            q = ops.gemm_h(z, wq, bq, coeff=cq, scale=self.nw.q, act='tanh', out_dtype=f32)
            return s, t, q
        z = ops.gemm(x, w['wx'], w['bx'], a2=v, w2=w['wv'], bias2=w['bv'], act=self.act)
        for hw, hb in w['hidden']:
            z = ops.gemm(z, hw, hb, act=self.act)
        ws, bs, cs = w['heads']['s']
        wt, bt, _ = w['heads']['t']
        wq, bq, cq = w['heads']['q']
        s = ops.gemm(z, ws, bs, coeff=cs, scale=self.nw.s, act='tanh')
        t = ops.gemm(z, wt, bt, scale=self.nw.t)
        q = ops.gemm(z, wq, bq, coeff=cq, scale=self.nw.q, act='tanh')
        return s, t, q

    def hidden_flat_h(self, x: Tensor, v: Tensor, w: Optional[dict] = None) -> Tensor:
        """z = last hidden activation in 16 bit (half-precision mode)."""
        self._check_mode()
        h = (self.kernel_weights() if w is None else w)['h']
        z = ops.gemm_h(x, h['wx'], h['bx'], a2=v, w2=h['wv'], bias2=h['bv'], act=self.act)
        for hw, hb in h['hidden']:
            z = ops.gemm_h(z, hw, hb, act=self.act)
        return z

    def hidden_flat_h_u1x(self, x: Tensor, mask: Tensor, complement: bool, v: Tensor,
                          w: Optional[dict] = None) -> Tensor:
        """hidden_flat_h for the U(1) xnet without a conv stack: the [cos(m x), sin(m x)] input
        is formed from the link angles inside the first GEMM's tile loader."""
        self._check_mode()
        h = (self.kernel_weights() if w is None else w)['h']
        z = ops.gemm_h_u1x(x, mask, complement, h['wx'], h['bx'], v, h['wv'], h['bv'], self.act)
        for hw, hb in h['hidden']:
            z = ops.gemm_h(z, hw, hb, act=self.act)
        return z

    def hidden_flat(self, x: Tensor, v: Tensor, w: dict, sliced_exp: Optional[int] = None) -> Tensor:
        """z = last hidden activation [nb, units[-1]] (input layer + hidden layers).

        sliced_exp = e: the caller guarantees |x|, |v| < 2^e entry-wise (the SU(3) vnet inputs
        su3_to_vec(projectSU(.)) with e = 2), which lets the fp64 input layer run on the int8 matrix
        cores (csrc/gemm_sliced.hip) when its shapes qualify; the digit images of xlayer / vlayer live in
        the version-keyed weight cache `w` and are rebuilt whenever a parameter changes."""
        self._check_mode()
        if getattr(self, 'half_dtype', None) is not None:
            raise NotImplementedError('hidden_flat feeds the fp64 fused heads kernel (SU(3))')
        z = None
        if sliced_exp is not None and ops.USE_SLICED_INPUT[0] and x.dtype == torch.float64 and \
                ops.gemm_sliced_pays(x.shape[0], w['wx'].shape[0], x.shape[1], v.shape[1]):
            if 'input_img' not in w:
                # the images take 7/8 of the weights' bytes again (2 x 235 MB per vnet at cfg-4, 2 x 3.8 GB
                # at 16^4): built only while that leaves most of the free memory alone
                need = 2 * (w['wx'].numel() + w['wv'].numel()) * 8
                ix = iv = None
                if not x.is_cuda or ops.mem_gate('sliced input-layer images', need, 0.25, x.device):
                    ix, iv = ops.gemm_sliced_build(w['wx']), ops.gemm_sliced_build(w['wv'])
                w['input_img'] = (ix, iv) if ix is not None and iv is not None else None
            if w['input_img'] is not None:
                z = ops.gemm_sliced(x, w['input_img'][0], w['wx'].shape[0], w['bx'], a_exp=sliced_exp,
                                    a2=v, image2=w['input_img'][1], a2_exp=sliced_exp, bias2=w['bv'],
                                    act=self.act)
        if z is None:
            z = ops.gemm(x, w['wx'], w['bx'], a2=v, w2=w['wv'], bias2=w['bv'], act=self.act)
        for hw, hb in w['hidden']:
            z = ops.gemm(z, hw, hb, act=self.act)
        return z



    # ---- native-order shadows for training (SU(3) vnet)
    def native_train_begin(self, in_perm: Tensor, out_perm: Tensor) -> None:
        """Training in the kernels' native entry order: the big matrices of this layer (input
        columns of xlayer / vlayer, output rows of the three heads, their biases and ScaledTanh
        coefficients) are gathered once per step into native-order shadows, forward_train /
        backward run on the shadows (gradients accumulate in native-order buffers), and
        native_train_end scatters the gradients back into the checkpoint-ordered .grad views.
        Replaces the five reference <-> native transposes of the activations around every
        network call of the tape (2 x 268 MB + 3 x 302 MB per call at cfg-4, forward and
        backward) by one gather + one scatter of the weights per step."""
        il = self.input_layer
        if isinstance(il.conv_stack, ConvStack):
            raise NotImplementedError('native-order training: dense input layers only')
        nat = getattr(self, '_nat', None)
        if nat is None or nat['in'] is not in_perm or nat['out'] is not out_perm:
            nat = {'in': in_perm, 'out': out_perm, 'w': {}, 'g': {}}
            self._nat = nat
        src = {'wx': (il.xlayer.weight, 1, in_perm), 'wv': (il.vlayer.weight, 1, in_perm),
               'ws': (self.scale.layer.weight, 0, out_perm), 'bs': (self.scale.layer.bias, 0, out_perm),
               'cs': (self.scale.coeff, -1, out_perm),
               'wt': (self.transl.weight, 0, out_perm), 'bt': (self.transl.bias, 0, out_perm),
               'wq': (self.transf.layer.weight, 0, out_perm), 'bq': (self.transf.layer.bias, 0, out_perm),
               'cq': (self.transf.coeff, -1, out_perm)}
        with torch.no_grad():
            for k, (par, dim, perm) in src.items():
                t = par.detach()
                if dim == -1:
                    t = t.reshape(-1)
                    dim = 0
                buf = nat['w'].get(k)
                if buf is None or buf.shape != t.shape:
                    buf = torch.empty_like(t)
                    nat['w'][k] = buf
                    nat['g'][k] = torch.zeros_like(t)
                torch.index_select(t, dim, perm, out=buf)
                nat['g'][k].zero_()
        nat['src'] = src
        nat.pop('sliced', None)            # this step's slice image of the heads: built on first use
        nat.pop('input_img', None)         # ... and the digit images of the input layer
        if nat.get('defer') is not None and not nat['defer'].get('off'):
            nat['defer']['i'] = 0
            nat['defer']['seen'] = set()
        nat['active'] = True

    def native_train_end(self, keep_active: bool = False, scatter: bool = True, on_ready=None) -> None:
        """Scatter-add the native-order gradients into the parameters' .grad, leave native mode.
        keep_active: flush only (the gradient buffers are zeroed, the weight shadows stay: another
        recorded trajectory of the same optimiser step still needs them -- dynamics/pytorch/autograd.py);
        scatter=False: leave without touching .grad (a recorded trajectory that was never reversed);
        on_ready([param]): called per matrix right after ITS deferred-gradient GEMM and scatter -- the
        data-parallel exchange of that slab then overlaps the next matrix's GEMM (training.GradReducer)."""
        nat = getattr(self, '_nat', None)
        if nat is None or not nat.get('active'):
            return
        if not scatter:
            d = nat.get('defer')
            if d is not None and not d.get('off'):
                d['i'] = 0
                d['seen'] = set()
            nat['active'] = False
            return
        # the deferred weight-gradient GEMMs run matrix by matrix, each just in front of its scatter (all at
        # once here when nobody listens; a no-op when the reverse sweep has flushed already)
        jobs = self._deferred_jobs()
        if on_ready is None:
            for run in jobs.values():
                run()
            jobs = {}
        with torch.no_grad():
            inv = nat.setdefault('inv', {})
            # big matrices first: their exchange takes longest, the small vectors ride in its shadow
            order = sorted(nat['src'].items(), key=lambda kv: -kv[1][0].numel()) if on_ready is not None \
                else list(nat['src'].items())
            for k, (par, dim, perm) in order:
                if k in jobs:
                    jobs.pop(k)()
                g = nat['g'][k]
                if par.grad is None:
                    if keep_active:
                        g.zero_()
                    continue
                # perm is a bijection: grad[perm] += g  ==  grad += g[perm^-1]  (a gather and an
                # add instead of index_add_'s atomics: 0.25 instead of 0.8 ms per 302 MB matrix)
                ip = inv.get(id(perm))
                if ip is None:
                    ip = torch.empty_like(perm)
                    ip[perm] = torch.arange(perm.numel(), device=perm.device, dtype=perm.dtype)
                    inv[id(perm)] = ip
                if dim == -1:
                    par.grad.reshape(-1).add_(torch.index_select(g, 0, ip))
                else:
                    par.grad.add_(torch.index_select(g, dim, ip))
                if keep_active:
                    g.zero_()
                if on_ready is not None:
                    on_ready([par])
            if on_ready is not None:
                # the layers that never had shadows (hidden layers, the input layer's biases, BatchNorm)
                shadowed = {id(par) for par, _d, _p in nat['src'].values()}
                on_ready([q for q in self.parameters() if id(q) not in shadowed])
        nat['active'] = bool(keep_active)

    def native_train_release(self) -> None:
        """Drop what native-order training keeps between optimiser steps -- weight / gradient shadows,
        the deferred-gradient arenas (8 GB at cfg-4), the slice images of the tape -- when the layer leaves
        training (`Dynamics.eval()`): a sampler that runs next must not find that memory pinned (its own
        slice images are gated on free memory).  A later train step allocates them again."""
        nat = getattr(self, '_nat', None)
        if nat is None or nat.get('active'):
            return                                   # (never inside a step)
        self._nat = None

    def native_active(self) -> bool:
        nat = getattr(self, '_nat', None)
        return bool(nat is not None and nat.get('active'))

    # ---- training path (train-mode semantics: dropout active, BatchNorm batch statistics)
    def training_needs_fresh_forward(self) -> bool:
        """True when two forward_train calls on the same input differ on purpose (fresh dropout
        mask per call, BatchNorm running statistics updated per call): the trajectory tape must
        then evaluate every v-update separately, like the reference's autograd graph does."""
        return self.training and (float(self.net_config.dropout_prob) > 0
                                  or bool(self.net_config.use_batch_norm))

    def forward_train(self, x: Tensor, v: Tensor, drop_keep: Optional[Tensor] = None,
                      hidden_only: bool = False, sliced_input_exp: Optional[int] = None
                      ) -> tuple[Tensor, Tensor, Tensor, dict]:
        """(s, t, q, ctx).  drop_keep: a given dropout keep-mask [nb, units[-1]] instead of a
        fresh draw (the construction-time dummy forward replays the host generator's mask).  x: the network's x input ([nb, C, T, X] when there is a conv stack,
        otherwise anything flattenable to [nb, Kx]); v likewise.  reference: network.py:522-551
        under autograd.  hidden_only: stop in front of the heads and return (z, ctx) -- the caller runs
        heads_vupdate_train_sliced, which completes ctx.  sliced_input_exp = e: the caller guarantees
        |x|, |v| < 2^e (the SU(3) vnet inputs, e = 2) and native-order training is on: the input layer may run
        on the digit images of this step's weights (sliced_train_input_images)."""
        il = self.input_layer
        nb = x.shape[0]
        if getattr(self, 'half_dtype', None) is not None and getattr(self, 'half_train', True):
            return self._forward_train_half(x, v, drop_keep)
        conv_ctx = None
        if isinstance(il.conv_stack, ConvStack):
            xf, conv_ctx = il.conv_stack.forward_train(x)
        else:
            xf = x.reshape(nb, -1).contiguous()
        vf = v.reshape(nb, -1).contiguous()
        # swish is not invertible: its derivative needs the pre-activation, so for swish the
        # layers run without the fused activation and the tape keeps z_pre next to act(z_pre)
        swish = self.act == 'swish'
        if swish and conv_ctx is not None:
            raise NotImplementedError('training: swish inside the conv stack (its kernels fuse the '
                                      'activation with the max-pool) -- use another activation_fn')
        fused = None if swish else self.act
        nw_ = self._nat['w'] if self.native_active() else None     # inputs / outputs in native order
        imgs = self.sliced_train_input_images(nb) if (sliced_input_exp is not None and nw_ is not None
                                                      and fused is not None and conv_ctx is None) else None
        if imgs is not None:
            # the input layer of the tape on the int8 matrix cores (csrc/gemm_sliced.hip), like the sampler's
            z = ops.gemm_sliced(xf, imgs[0], nw_['wx'].shape[0], il.xlayer.bias.detach(), a_exp=sliced_input_exp,
                                a2=vf, image2=imgs[1], a2_exp=sliced_input_exp, bias2=il.vlayer.bias.detach(),
                                act=fused)
        else:
            z = ops.gemm(xf, il.xlayer.weight.detach() if nw_ is None else nw_['wx'],
                         il.xlayer.bias.detach(), a2=vf,
                         w2=il.vlayer.weight.detach() if nw_ is None else nw_['wv'],
                         bias2=il.vlayer.bias.detach(), act=fused)
        pre = [z] if swish else None
        if swish:
            z = ops.act_fwd(z, 'swish')
        acts = [z]
        for h in self.hidden_layers:
            z = ops.gemm(z, h.weight.detach(), h.bias.detach(), act=fused)
            if swish:
                pre.append(z)
                z = ops.act_fwd(z, 'swish')
            acts.append(z)
        ctx: dict = {'xf': xf, 'vf': vf, 'acts': acts, 'pre': pre, 'conv': conv_ctx,
                     'xshape': tuple(x.shape), 'vshape': tuple(v.shape)}
        p = float(self.net_config.dropout_prob)
        if p > 0 and self.training:
            keep = torch.bernoulli(torch.full_like(z, 1.0 - p)) if drop_keep is None \
                else drop_keep.to(device=z.device, dtype=z.dtype)
            z = ops.mul(z, keep, 1.0 / (1.0 - p))
            ctx['drop'] = keep
        if self.net_config.use_batch_norm:
            bn = self.batch_norm
            ctx['bn_in'] = z
            mom = 0.1 if bn.momentum is None else float(bn.momentum)
            z, ctx['bn_mean'], ctx['bn_invstd'] = ops.bn_train_fwd(
                z, bn.weight.detach(), bn.bias.detach(), bn.eps, mom, bn.running_mean,
                bn.running_var)
            bn.num_batches_tracked += 1
        ctx['z'] = z
        ctx['native'] = nw_ is not None
        if hidden_only:
            return z, ctx
        if nw_ is not None:
            s = ops.gemm(z, nw_['ws'], nw_['bs'], coeff=nw_['cs'], scale=self.nw.s, act='tanh')
            t = ops.gemm(z, nw_['wt'], nw_['bt'], scale=self.nw.t)
            q = ops.gemm(z, nw_['wq'], nw_['bq'], coeff=nw_['cq'], scale=self.nw.q, act='tanh')
        else:
            s = ops.gemm(z, self.scale.layer.weight.detach(), self.scale.layer.bias.detach(),
                         coeff=self.scale.coeff.detach().reshape(-1), scale=self.nw.s, act='tanh')
            t = ops.gemm(z, self.transl.weight.detach(), self.transl.bias.detach(), scale=self.nw.t)
            q = ops.gemm(z, self.transf.layer.weight.detach(), self.transf.layer.bias.detach(),
                         coeff=self.transf.coeff.detach().reshape(-1), scale=self.nw.q, act='tanh')
        ctx['s'], ctx['q'] = s, q
        return s, t, q, ctx

    def _half_train_weights(self) -> dict:
        """16-bit copies of the RAW weights for the train-mode forward (kernel_weights() folds the
        eval-mode BatchNorm into the heads; train mode normalises with batch statistics).  What
        torch.autocast's weight cache holds during the reference's forward_step."""
        ver = self._versions()
        hit = getattr(self, '_half_train_cache', None)
        if hit is not None and hit['ver'] == ver:
            return hit
        hd = self.half_dtype
        with torch.no_grad():
            il = self.input_layer
            r16 = lambda t: t.detach().to(hd).float().contiguous()
            w16 = lambda t: t.detach().to(hd).contiguous()
            out = {'ver': ver, 'wx': w16(il.xlayer.weight), 'bx': r16(il.xlayer.bias),
                   'wv': w16(il.vlayer.weight), 'bv': r16(il.vlayer.bias),
                   'hidden': [(w16(h.weight), r16(h.bias)) for h in self.hidden_layers],
                   'ws': w16(self.scale.layer.weight), 'bs': r16(self.scale.layer.bias),
                   'cs': self.scale.coeff.detach().reshape(-1).contiguous(),
                   'wt': w16(self.transl.weight), 'bt': r16(self.transl.bias),
                   'wq': w16(self.transf.layer.weight), 'bq': r16(self.transf.layer.bias),
                   'cq': self.transf.coeff.detach().reshape(-1).contiguous()}
        self._half_train_cache = out
        return out

    def _forward_train_half(self, x: Tensor, v: Tensor, drop_keep: Optional[Tensor] = None):
        """forward_train under the reference's `autocast_context_train` (trainers/pytorch/trainer.py:211-219,
        1276-1280): every nn.Linear runs on the 16-bit MFMA kernels with autocast's rounding points (operands
        and outputs rounded to fp16 / bf16, fp32 accumulation -- the layers the eval path is pinned to the
        reference's autocast fixtures with), Dropout / BatchNorm act on the 16-bit activations, the ScaledTanh
        multipliers and everything downstream are fp32.  The tape keeps fp32 copies of the activations: the
        reverse sweep differentiates THIS forward with the casts as identities (what autocast's backward does)
        on the fp32 master weights, in fp32 -- the reference runs those products in 16 bit; the difference is
        below its own fp16-vs-fp32 distance, which is what the tests pin."""
        il = self.input_layer
        if self.act == 'swish':
            raise NotImplementedError('half-precision training: swish keeps pre-activations; use another activation_fn')
        hd = self.half_dtype
        nb = x.shape[0]
        h = self._half_train_weights()
        conv_ctx = None
        if isinstance(il.conv_stack, ConvStack):
            xf, conv_ctx = il.conv_stack.forward_train(x, half=hd)
        else:
            xf = x.reshape(nb, -1).float().contiguous()
        vf = v.reshape(nb, -1).float().contiguous()
        z = ops.gemm_h(xf, h['wx'], h['bx'], a2=vf, w2=h['wv'], bias2=h['bv'], act=self.act)
        acts = [z.float()]
        for hw, hb in h['hidden']:
            z = ops.gemm_h(z, hw, hb, act=self.act)
            acts.append(z.float())
        ctx: dict = {'xf': xf, 'vf': vf, 'acts': acts, 'pre': None, 'conv': conv_ctx,
                     'xshape': tuple(x.shape), 'vshape': tuple(v.shape), 'native': False}
        zf = acts[-1]
        p = float(self.net_config.dropout_prob)
        if p > 0 and self.training:
            keep = torch.bernoulli(torch.full_like(zf, 1.0 - p)) if drop_keep is None \
                else drop_keep.to(device=zf.device, dtype=zf.dtype)
            zf = ops.mul(zf, keep, 1.0 / (1.0 - p)).to(hd).float()
            ctx['drop'] = keep
        if self.net_config.use_batch_norm:
            bn = self.batch_norm
            ctx['bn_in'] = zf
            mom = 0.1 if bn.momentum is None else float(bn.momentum)
            zf, ctx['bn_mean'], ctx['bn_invstd'] = ops.bn_train_fwd(
                zf, bn.weight.detach(), bn.bias.detach(), bn.eps, mom, bn.running_mean, bn.running_var)
            bn.num_batches_tracked += 1
            zf = zf.to(hd).float()
        ctx['z'] = zf
        z16 = zf.to(hd)
        f32 = torch.float32
        s = ops.gemm_h(z16, h['ws'], h['bs'], coeff=h['cs'], scale=self.nw.s, act='tanh', out_dtype=f32)
        t = ops.gemm_h(z16, h['wt'], h['bt'], scale=self.nw.t, out_dtype=f32)
        q = ops.gemm_h(z16, h['wq'], h['bq'], coeff=h['cq'], scale=self.nw.q, act='tanh', out_dtype=f32)
        ctx['s'], ctx['q'] = s, q
        return s, t, q, ctx

    # ---- weight gradients of all network calls of a step in one GEMM per matrix
    DEFER_MAX_FREE_FRACTION = 0.25

    def defer_slot(self, nb: int, kx: int, kv: int, capacity: int):
        """Native-order training: activation slots of the next network call of this step inside contiguous
        arenas ([capacity, nb, k]), or None.  The weight-gradient GEMMs dW += dpre^T x have the chains as their
        contraction dimension -- 256 deep per call, where the fp64 MFMA kernel spends as long on its prologue,
        its accumulate epilogue (W.grad read and written per call) and tile quantisation as on the products:
        0.53 / 0.42 ms per call for the heads / input matrices at cfg-4.  With the inputs (vec8 of x and of
        the force), the last hidden activations and the heads' pre-activation cotangents of ALL calls of the
        step kept side by side, flush_deferred forms every W.grad in ONE GEMM with K = calls x chains:
        0.30 / 0.26 ms per call (tools/gemm_ex_probe.py), W.grad touched once.  Costs the cotangent arenas
        (3 x calls x chains x N x 8 bytes: 8 GB at cfg-4), taken only while that is a small part of the free
        device memory; the input arenas replace per-call tensors of the same size."""
        nat = getattr(self, '_nat', None)
        if nat is None or not nat.get('active') or capacity <= 0:
            return None
        if not getattr(self, 'defer_weight_grads', True):
            return None
        d = nat.get('defer')
        n_out = nat['w']['bs'].numel()
        units = nat['w']['ws'].shape[1]                # last hidden width (the heads' input)
        units_in = nat['w']['wx'].shape[0]             # first hidden width (the input layer's output)
        key = (nb, kx, kv, capacity)
        if d is None or d['key'] != key:
            nat.pop('defer', None)
            dt = nat['w']['ws'].dtype
            es = 8 if dt == torch.float64 else 4
            extra = 3 * capacity * nb * n_out * es
            total = extra + capacity * nb * (kx + kv + units + units_in) * es
            if not ops.mem_gate('deferred weight-gradient arenas', total, self.DEFER_MAX_FREE_FRACTION,
                                nat['w']['ws'].device):
                nat['defer'] = {'key': key, 'off': True}
                return None
            dev = nat['w']['ws'].device
            d = {'key': key, 'off': False, 'i': 0, 'seen': set(),
                 'xv': torch.empty((capacity, nb, kx), dtype=dt, device=dev),
                 'fv': torch.empty((capacity, nb, kv), dtype=dt, device=dev),
                 'z': torch.empty((capacity, nb, units), dtype=dt, device=dev),
                 'dpre_in': torch.empty((capacity, nb, units_in), dtype=dt, device=dev),
                 'dpre': {t: torch.empty((capacity, nb, n_out), dtype=dt, device=dev) for t in 'stq'}}
            nat['defer'] = d
        if d.get('off') or d['i'] >= capacity:
            return None
        i = d['i']
        d['i'] += 1
        return i, d['xv'][i], d['fv'][i]

    def _deferred_jobs(self) -> dict:
        """matrix key -> thunk forming that matrix's W.grad from the arenas filled by this step's backward
        calls (one GEMM with K = calls x chains each); the arenas are marked consumed."""
        nat = getattr(self, '_nat', None)
        d = None if nat is None else nat.get('defer')
        if d is None or d.get('off') or not d['seen']:
            return {}
        n = d['i']
        if d['seen'] != set(range(n)):
            raise RuntimeError(f'flush_deferred: backward ran for calls {sorted(d["seen"])} of {n}')
        nb = d['key'][0]
        ng_ = nat['g']
        rows = lambda a: a[:n].reshape(n * nb, -1)

        def job(dpre, act, key):
            return lambda: ops.gemm_ex(rows(dpre), rows(act), a_trans=True, w_trans=True, out=ng_[key],
                                       accumulate=True)
        jobs = {'w' + tag: job(d['dpre'][tag], d['z'], 'w' + tag) for tag in 'stq'}
        jobs['wx'] = job(d['dpre_in'], d['xv'], 'wx')
        jobs['wv'] = job(d['dpre_in'], d['fv'], 'wv')
        d['i'] = 0
        d['seen'] = set()
        return jobs

    def flush_deferred(self) -> None:
        """W.grad of the five big matrices from the arenas filled by this step's backward calls."""
        for run in self._deferred_jobs().values():
            run()

    # ---- the heads of the training tape on the int8-sliced kernel (csrc/heads_sliced.hip, TAPE instances)
    def sliced_train_image(self):
        """Slice image of this step's native-order head weights, or None when the layer does not qualify
        (not in native-order training, fp32, units[-1] != 256, an unbounded activation in front of the heads,
        dropout / BatchNorm between them, or weights the 54-bit fixed point refuses).  Built on first use
        after native_train_begin (one pass over the 3 x N x 256 weights + a stream synchronisation per
        optimiser step), into the previous step's buffer."""
        nat = getattr(self, '_nat', None)
        if nat is None or not nat.get('active'):
            return None
        if 'sliced' in nat:
            return nat['sliced']
        w = nat['w']
        ok = (w['ws'].dtype == torch.float64 and w['ws'].is_cuda and w['ws'].shape[1] == ops.SLICED_K
              and self.act == 'tanh' and not self.net_config.use_batch_norm
              and not (float(self.net_config.dropout_prob) > 0 and self.training)
              and isinstance(self.scale, ScaledTanh) and isinstance(self.transf, ScaledTanh))
        image = None
        if ok:
            buf, usable = ops.heads_sliced_build_into(w['ws'], w['wt'], w['wq'], nat.get('sliced_buf'))
            nat['sliced_buf'] = buf
            if usable:
                image = {'image': buf, 'cs': float(self.nw.s) * torch.exp(w['cs']),
                         'cq': float(self.nw.q) * torch.exp(w['cq'])}
        nat['sliced'] = image
        return image

    def sliced_train_input_images(self, nb: int):
        """Digit images of this step's native-order xlayer / vlayer weights (csrc/gemm_sliced.hip), or None
        when the shapes do not qualify (`_ops.gemm_sliced_pays`), the images do not fit comfortably or a
        matrix is refused.  Built on first use after native_train_begin (two passes over the weights + two
        stream synchronisations per optimiser step)."""
        nat = getattr(self, '_nat', None)
        if nat is None or not nat.get('active'):
            return None
        if 'input_img' in nat:
            return nat['input_img']
        wx, wv = nat['w']['wx'], nat['w']['wv']
        imgs = None
        if (ops.USE_SLICED_INPUT[0] and wx.dtype == torch.float64 and wx.is_cuda
                and ops.gemm_sliced_pays(nb, wx.shape[0], wx.shape[1], wv.shape[1])):
            need = 2 * (wx.numel() + wv.numel()) * 8
            if ops.mem_gate('sliced input-layer images (training tape)', need, 0.25, wx.device):
                ix, iv = ops.gemm_sliced_build(wx), ops.gemm_sliced_build(wv)
                imgs = (ix, iv) if ix is not None and iv is not None else None
        nat['input_img'] = imgs
        return imgs

    def heads_vupdate_train_sliced(self, z: Tensor, ctx: dict, v: Tensor, force: Tensor, eps: float,
                                   forward: bool):
        """(v', logdet, s, t, q) from the hidden activations of forward_train(..., hidden_only=True): the three
        heads and the first momentum update that consumes them in one kernel; completes ctx for backward."""
        im = self.sliced_train_image()
        w = self._nat['w']
        v_new, ld, s, t, q = ops.vnet_heads_vupdate_sliced_tape(
            z, im['image'], w['bs'], im['cs'], w['bt'], float(self.nw.t), w['bq'], im['cq'], v, force, eps,
            forward)
        ctx['s'], ctx['q'] = s, q
        return v_new, ld, s, t, q

    def backward(self, ctx: dict, ds: Tensor, dt: Tensor, dq: Tensor) -> tuple[Tensor, Tensor]:
        """Accumulates every parameter's .grad; returns (dL/dx_in, dL/dv_in) shaped like the
        inputs of forward_train."""
        z = ctx['z']
        zT = None            # (only the fall-back path of _linear_bwd transposes)
        dz = None
        native = bool(ctx.get('native'))
        if native and not self.native_active():
            raise RuntimeError('LeapfrogLayer.backward: the tape was recorded in native order but '
                               'native_train_end() has already run')
        nw_ = self._nat['w'] if native else None
        ng_ = self._nat['g'] if native else None
        # weight gradients deferred to flush_deferred (defer_slot): this call parks its operands in slot di
        di = ctx.get('defer_idx') if native else None
        dfr = self._nat.get('defer') if di is not None else None
        if dfr is not None:
            dfr['z'][di].copy_(z)
            dfr['seen'].add(di)
        for head, cot, out, tag in ((self.scale, ds, ctx['s'], 's'), (self.transl, dt, None, 't'),
                                    (self.transf, dq, ctx['q'], 'q')):
            # the head's VJP with its bias and coefficient gradients (column sums over the chains; d s / d coeff
            # = s) formed in the same pass over (cot, out)
            lin = head.layer if isinstance(head, ScaledTanh) else head
            bg = ng_['b' + tag] if native else lin.bias.grad
            slot = None if dfr is None else dfr['dpre'][tag][di]
            if isinstance(head, ScaledTanh):
                nw = self.nw.s if head is self.scale else self.nw.q
                co = head.coeff.detach().reshape(-1) if not native else nw_['c' + tag]
                cg = head.coeff.grad.reshape(-1) if not native else ng_['c' + tag]
                dpre = ops.scaled_tanh_bwd_sums(cot, out, co, nw, bg, cg, out=slot)
            else:
                dpre = ops.scaled_tanh_bwd_sums(cot, None, None, self.nw.t, bg, None, out=slot)
            if native:
                d = _linear_bwd(dpre, z, nw_['w' + tag], nw_['b' + tag], xT=zT,
                                wgrad=ng_['w' + tag], bgrad=ng_['b' + tag], bias_done=True,
                                skip_dw=dfr is not None)
            else:
                d = _linear_bwd(dpre, z, lin.weight, lin.bias, xT=zT, bias_done=True)
            dz = d if dz is None else ops.add_(dz, d)
        if self.net_config.use_batch_norm:
            bn = self.batch_norm
            dz = ops.bn_bwd(dz, ctx['bn_in'], ctx['bn_mean'], ctx['bn_invstd'], bn.weight.detach(),
                            bn.weight.grad, bn.bias.grad)
        if 'drop' in ctx:
            dz = ops.mul(dz, ctx['drop'], 1.0 / (1.0 - float(self.net_config.dropout_prob)))
        acts = ctx['acts']
        pre = ctx.get('pre') is not None
        dact = ctx['pre'] if pre else acts                          # swish: from the pre-activation
        for i in range(len(self.hidden_layers) - 1, -1, -1):
            h = self.hidden_layers[i]
            dpre = ops.act_bwd(dz, dact[i + 1], self.act, from_preact=pre)
            dz = _linear_bwd(dpre, acts[i], h.weight, h.bias)
        il = self.input_layer
        dpre = ops.act_bwd(dz, dact[0], self.act, from_preact=pre)
        if native:
            if dfr is not None:
                dfr['dpre_in'][di].copy_(dpre)
            dxf = _linear_bwd(dpre, ctx['xf'], nw_['wx'], il.xlayer.bias, wgrad=ng_['wx'], skip_dw=dfr is not None)
            dvf = _linear_bwd(dpre, ctx['vf'], nw_['wv'], il.vlayer.bias, wgrad=ng_['wv'], skip_dw=dfr is not None)
        else:
            dxf = _linear_bwd(dpre, ctx['xf'], il.xlayer.weight, il.xlayer.bias)
            dvf = _linear_bwd(dpre, ctx['vf'], il.vlayer.weight, il.vlayer.bias)
        if ctx['conv'] is not None:
            dx = il.conv_stack.backward(ctx['conv'], dxf)
        else:
            dx = dxf
        return dx.reshape(ctx['xshape']), dvf.reshape(ctx['vshape'])

    def forward(self, inputs: tuple[Tensor, Tensor]) -> tuple[Tensor, Tensor, Tensor]:
        x, v = inputs
        x, v = x.to(DEVICE), v.to(DEVICE)
        if isinstance(self.input_layer.conv_stack, ConvStack):
            x = self.input_layer.conv_stack(x)
        dt = self.transl.weight.dtype
        return self.forward_flat(flatten(x).to(dt).contiguous(), flatten(v).to(dt).contiguous())


def get_network(xshape, network_config, input_shapes=None, net_weight=None, conv_config=None,
                name=None, **kw) -> LeapfrogLayer:
    return LeapfrogLayer(xshape=xshape, network_config=network_config,
                         input_shapes=input_shapes, net_weight=net_weight,
                         conv_config=conv_config, name=name, **kw)


def _dummy_inputs(group, xshape: Sequence[int], values: bool):
    """The two draws `get_and_call_network` of the reference makes before building a network
    (network.py:589-590): `group.random(xshape)` and `group.random_momentum(xshape)`, on the
    host generator, in the default dtype.  values=False only advances the generator (the draws
    feed a dummy forward whose outputs are discarded; nothing but BatchNorm's running statistics
    remembers them)."""
    n = int(np.prod(xshape))
    if isinstance(group, SU3) or getattr(group, '_name', None) == 'SU3':
        if values:
            return group.random(xshape), group.random_momentum(xshape)
        advance_randn(n)                      # group.py:115-116: real and imaginary normals
        advance_randn(n)
        for _ in range(8):                    # utils.py:171-183 randTAH3: 8 draws of shape[:-2]
            advance_randn(n // 9)
        return None, None
    if values:
        return group.random(xshape), group.random_momentum(xshape)
    advance_rand(n)                           # u1/group.py:158-162
    advance_randn(n)
    return None, None


def get_and_call_network(xshape: Sequence[int], *, network_config: NetworkConfig,
                         is_xnet: bool, group: U1Phase | SU3,
                         input_shapes: Optional[dict[str, int | Sequence[int]]] = None,
                         net_weight: Optional[NetWeight] = None,
                         conv_config: Optional[ConvolutionConfig] = None,
                         name: Optional[str] = None) -> LeapfrogLayer:
    """The reference materialises its Lazy layers with one dummy forward
    (network.py:572-631); the widths that call would discover are computed directly here:
    U1 xnet sees [cos, sin] (4 channels, 2*xdim features), SU3 xnet sees real||imag
    (2*xdim features), SU3 vnet sees the 8-component vectors (input_shapes).

    What that dummy forward leaves behind is reproduced so that a seed gives the reference's
    (CPU-path) networks: the generator advances by the two dummy draws, every layer draws its
    weights at the reference's slot (see `_uninit`), Dropout draws its train-mode mask, and
    BatchNorm1d's running statistics take the one step of that forward (through the training
    kernels, on the drawn inputs)."""
    xdim = int(np.cumprod(xshape[1:])[-1])
    su3 = isinstance(group, SU3) or getattr(group, '_name', None) == 'SU3'
    kw: dict = {}
    if su3:
        if is_xnet:
            kw.update(x_features=2 * xdim, v_features=2 * xdim)
    else:
        kw['conv_channels'] = 4 if is_xnet else 2
        if is_xnet:
            kw['x_features'] = 2 * xdim
    bn = bool(network_config.use_batch_norm)
    x, v = _dummy_inputs(group, xshape, values=bn)
    net = get_network(xshape=xshape, network_config=network_config, input_shapes=input_shapes,
                      net_weight=net_weight, conv_config=conv_config, name=name, **kw)
    p = float(network_config.dropout_prob)
    keep = None
    if p > 0:
        # network.py:538-539 in train mode: F.dropout's mask does not depend on the values
        z = torch.ones(int(xshape[0]), int(network_config.units[-1]))
        keep = (torch.nn.functional.dropout(z, p, training=True) != 0)
    if bn:
        with torch.no_grad():
            nb = int(xshape[0])
            if su3:
                if is_xnet:
                    xi = torch.cat([x.real, x.imag], dim=1).reshape(nb, -1)
                    vi = torch.cat([v.real, v.imag], dim=1).reshape(nb, -1)
                else:
                    xi = group.group_to_vec(x).reshape(nb, -1)
                    vi = group.group_to_vec(v).reshape(nb, -1)
            else:
                xi = group.group_to_vec(x) if is_xnet else x
                vi = v
            dt = net.transl.weight.dtype
            net.forward_train(xi.to(DEVICE).to(dt).contiguous(), vi.to(DEVICE).to(dt).contiguous(),
                              drop_keep=keep)
    return net


class NetworkFactory(BaseNetworkFactory):
    def build_xnet(self, group: SU3 | U1Phase, name: Optional[str] = None) -> LeapfrogLayer:
        xname = 'xnet' if name is None else f'xnet/{name}'
        return get_and_call_network(
            xshape=self.input_spec.xshape, network_config=self.network_config, is_xnet=True,
            group=group, input_shapes=self.input_spec.xnet, net_weight=self.nw.x,
            conv_config=self.conv_config, name=xname)

    def build_vnet(self, group: SU3 | U1Phase, name: Optional[str] = None) -> LeapfrogLayer:
        vname = 'vnet' if name is None else f'vnet/{name}'
        return get_and_call_network(
            xshape=self.input_spec.xshape, network_config=self.network_config, is_xnet=False,
            group=group, input_shapes=self.input_spec.vnet, net_weight=self.nw.v,
            conv_config=self.conv_config, name=vname)

    def build_networks(self, n: int, split_xnets: bool, group: SU3 | U1Phase) -> nn.ModuleDict:
        assert n >= 1, 'Must build at least one network'
        if n == 1:
            return nn.ModuleDict({'xnet': self.build_xnet(group=group),
                                  'vnet': self.build_vnet(group=group)})
        vnet = nn.ModuleDict()
        xnet = nn.ModuleDict()
        for lf in range(n):
            vnet[f'{lf}'] = self.build_vnet(group=group, name=f'{lf}')
            if split_xnets:
                xnet[f'{lf}'] = nn.ModuleDict({
                    'first': self.build_xnet(group=group, name=f'{lf}/first'),
                    'second': self.build_xnet(group=group, name=f'{lf}/second')})
            else:
                xnet[f'{lf}'] = self.build_xnet(group=group, name=f'{lf}')
        return nn.ModuleDict({'xnet': xnet, 'vnet': vnet})
