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


def build_u1_seeded_dynamics(g, nb, device=None):
    """Product Dynamics configured like a tests/golden/u1_cfg*.npz fixture (make_golden_sizes.py)
    with ``nb`` chains: the weights are the counter-based numbers of tests/golden/seeded.py (the
    same ones the generator loaded into the reference), step sizes and masks from the fixture."""
    import os
    import sys
    import torch
    import l2hmc.configs as cfgs
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.lattice.u1.pytorch.lattice import LatticeU1
    from l2hmc.network.pytorch.network import NetworkFactory
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import seeded
    L = [int(i) for i in g['latvolume']]
    nlf = int(g['nleapfrog'])
    dc = cfgs.DynamicsConfig(nchains=nb, group='U1', latvolume=L, nleapfrog=nlf,
                             eps=float(g['eps']), eps_hmc=float(g['eps']), use_ncp=True,
                             verbose=True, use_split_xnets=bool(g['use_split_xnets']),
                             use_separate_networks=bool(g['use_separate_networks']),
                             merge_directions=True)
    kw = u1_net_kwargs(g)
    nc = cfgs.NetworkConfig(units=[int(i) for i in g['units']], activation_fn=kw['activation'],
                            dropout_prob=0.2, use_batch_norm=kw['use_batch_norm'])
    cc = cfgs.ConvolutionConfig(**kw['conv']) if kw['conv'] else cfgs.ConvolutionConfig()
    xdim = dc.xdim
    spec = cfgs.InputSpec(xshape=tuple(dc.xshape), xnet={'x': [xdim, 2], 'v': [xdim]},
                          vnet={'x': [xdim], 'v': [xdim]})
    lat = LatticeU1(nb, L)
    dyn = Dynamics(potential_fn=lat.action, config=dc,
                   network_factory=NetworkFactory(input_spec=spec, network_config=nc, conv_config=cc))
    sd = dyn.state_dict()
    dev_ = device if device is not None else next(dyn.parameters()).device
    seeded.fill_state_dict(sd, int(g['weight_seed']), float(g['head_scale']), torch_device=dev_)
    with torch.no_grad():
        for i in range(nlf):
            dyn.xeps[i].copy_(torch.tensor(float(g['xeps'][i])))
            dyn.veps[i].copy_(torch.tensor(float(g['veps'][i])))
    dyn._eps_cache = {}
    dyn.set_masks(g['masks'])
    dyn.eval()
    return dyn, lat


def u1_seeded_oracle(g, dyn, dtype=np.float32):
    """DynamicsOracle driven by the state_dict of a build_u1_seeded_dynamics product."""
    sd = {k: v.detach().cpu().numpy() for k, v in dyn.state_dict().items()
          if not k.startswith('networks.')}
    kw = u1_net_kwargs(g)
    nlf = int(g['nleapfrog'])

    def vnet(step, x, f):
        return onet.leapfrog_layer(x, f, sub(sd, f'vnet.{step}.'), **kw)

    def xnet(step, first, x, v):
        which = 'first' if first else 'second'
        return onet.leapfrog_layer(x, v, sub(sd, f'xnet.{step}.{which}.'), **kw)
    L = tuple(int(i) for i in g['latvolume'])
    return DynamicsOracle('U1', L, nlf, [sd[f'xeps.{i}'] for i in range(nlf)],
                          [sd[f'veps.{i}'] for i in range(nlf)], g['masks'], vnet=vnet, xnet=xnet,
                          dtype=dtype)


def build_u1_train_dynamics(g, dropout=0.0):
    """Product Dynamics + LatticeLoss configured like a u1_train_* golden file (train mode).
    Call under the default dtype the fixture was generated with."""
    import torch
    import l2hmc.configs as cfgs
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.lattice.u1.pytorch.lattice import LatticeU1
    from l2hmc.loss.pytorch.loss import LatticeLoss
    from l2hmc.network.pytorch.network import NetworkFactory
    L = [int(i) for i in g['latvolume']]
    nb = int(g['x'].shape[0])
    nlf = int(g['nleapfrog'])
    merge = bool(g['merge_directions']) if 'merge_directions' in g else True
    dc = cfgs.DynamicsConfig(nchains=nb, group='U1', latvolume=L, nleapfrog=nlf, eps=0.1,
                             eps_hmc=0.1, use_ncp=True, verbose=False, use_split_xnets=True,
                             use_separate_networks=True, merge_directions=merge)
    kw = u1_net_kwargs(g)
    nc = cfgs.NetworkConfig(units=[int(i) for i in g['units']], activation_fn=kw['activation'],
                            dropout_prob=dropout, use_batch_norm=kw['use_batch_norm'])
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
    dyn.train()
    loss_fn = LatticeLoss(lat, cfgs.LossConfig(use_mixed_loss=bool(g['use_mixed_loss']),
                                               charge_weight=float(g['charge_weight'])))
    return dyn, lat, loss_fn


def check_train_step(g, dyn, loss_fn, rtol, atol_rel, adam_min_grad=0.0, autograd=False):
    """Run the product's forward + reverse sweep + Adam on the fixture's inputs and compare with
    the reference's loss / gradients / updated parameters.  Returns the worst relative errors.

    autograd=True: the reference caller's LITERAL sequence instead of the product's own trainer
    entry (trainers/pytorch/trainer.py:1266-1314): `x.requires_grad_(True); optimizer.zero_grad();
    xout, metrics = dynamics((x, beta)); loss = loss_fn(x, mc_states.proposed.x, acc); loss.backward();
    torch.optim.Adam(dynamics.parameters()).step()` -- nothing of this repo's training API is called."""
    import torch
    from l2hmc.dynamics.pytorch import training as T
    dyn._inject = {'normals': g['normals'], 'u': g['u']}
    if 'forward' in g and not bool(g['merge_directions']):
        dyn._inject['forward'] = bool(g['forward'])          # the direction draw of apply_transition
    x = torch.from_numpy(g['x'])
    beta = torch.tensor(float(g['beta']))
    xin = dyn.g.compat_proj(dyn.unflatten(x.to(dyn.device)))
    if autograd:
        optimizer = torch.optim.Adam(dyn.parameters(), lr=float(g['lr']))
        xin.requires_grad_(True)
        optimizer.zero_grad()
        xout, metrics = dyn((xin, beta))
        assert xout.grad_fn is not None and metrics['acc'].grad_fn is not None
        xprop = metrics['mc_states'].proposed.x
        assert xprop.grad_fn is not None
        loss = loss_fn(x_init=xin, x_prop=xprop, acc=metrics['acc'])
        loss.backward()
        loss = loss.detach()

        class _Opt:                        # the rest of the check steps through `arena.adam_step`
            def adam_step(self, lr):
                optimizer.step()
        arena = _Opt()
    else:
        arena = T.ParamArena(dyn)
        arena.zero_grad()
        xout, metrics, loss = T.train_forward_backward(dyn, loss_fn, xin, beta)
    dyn._inject = None
    out = {}
    acc = metrics['acc'].detach().cpu().numpy()
    np.testing.assert_allclose(acc, g['acc'], rtol=rtol, atol=rtol)
    xp = metrics['mc_states'].proposed.x.detach().cpu().numpy().reshape(g['x_prop'].shape)
    d = np.abs(np.angle(np.exp(1j * (xp - g['x_prop'])))).max()
    assert d < 50 * rtol, f'x_prop differs by {d}'
    assert np.array_equal(metrics['acc_mask'].cpu().numpy(), g['acc_mask'])
    np.testing.assert_allclose(float(loss), float(g['loss']), rtol=20 * rtol)
    worst = 0.0
    gn = np.sqrt(sum(float((g[k] ** 2).sum()) for k in list(g) if k.startswith('grad.')))
    names = dict(dyn.named_parameters())
    for k in list(g):
        if not k.startswith('grad.'):
            continue
        ref = g[k]
        assert names[k[5:]].grad is not None, f'{k}: no gradient reached this parameter'
        got = names[k[5:]].grad.detach().cpu().numpy()
        err = np.abs(got - ref).max()
        scale = max(np.abs(ref).max(), atol_rel * gn)
        worst = max(worst, err / scale)
        assert err <= rtol * 100 * scale, f'{k}: err {err:.3e} vs scale {scale:.3e}'
    out['grad_rel'] = worst
    arena.adam_step(lr=float(g['lr']))
    sd = dyn.state_dict()
    wp = 0.0
    for k in list(g):
        if not k.startswith('sd1.'):
            continue
        name = k[4:]
        if name.endswith('num_batches_tracked'):
            assert int(sd[name]) == int(g[k]), name
            continue
        got = sd[name].detach().cpu().numpy()
        ref = g[k]
        before = g['sd.' + name]
        # Adam's first step moves every touched weight by ~lr: compare the *update*
        diff = np.abs((got - before) - (ref - before))
        gk = 'grad.' + name if ('grad.' + name) in g else 'grad.networks.' + name
        if adam_min_grad > 0 and gk in g:
            # Adam's first update is lr * g / (|g| + 1e-8): where the gradient is rounding noise
            # the sign of the step is noise too -- compare where the gradient is resolved
            diff = np.where(np.abs(g[gk]) > adam_min_grad, diff, 0.0)
        err = diff.max() if diff.size else 0.0
        if err > wp:
            wp, out['param_worst'] = err, name
    out['param_abs'] = wp
    return out


def build_su3_train_dynamics(g):
    """Product Dynamics + LatticeLoss configured like tests/golden/su3_train.npz (train mode,
    float64 default dtype)."""
    import torch
    import l2hmc.configs as cfgs
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3
    from l2hmc.loss.pytorch.loss import LatticeLoss
    from l2hmc.network.pytorch.network import NetworkFactory
    L = [int(i) for i in g['latvolume']]
    nb = int(g['x'].shape[0])
    nlf = int(g['nleapfrog'])
    merge = bool(g['merge_directions']) if 'merge_directions' in g else True
    dc = cfgs.DynamicsConfig(nchains=nb, group='SU3', latvolume=L, nleapfrog=nlf, eps=0.006,
                             eps_hmc=0.006, verbose=False, use_split_xnets=False,
                             use_separate_networks=False, merge_directions=merge)
    nc = cfgs.NetworkConfig(units=[int(i) for i in g['units']], activation_fn=str(g['activation']),
                            dropout_prob=0.0, use_batch_norm=bool(g['use_batch_norm']))
    V = int(np.prod(L))
    spec = cfgs.InputSpec(xshape=tuple(dc.xshape), xnet={'x': [32 * V], 'v': [32 * V]},
                          vnet={'x': [32 * V], 'v': [32 * V]})
    lat = LatticeSU3(nb, L, c1=float(g['c1']) if 'c1' in g else 0.0)
    nf = NetworkFactory(input_spec=spec, network_config=nc, conv_config=cfgs.ConvolutionConfig())
    dyn = Dynamics(potential_fn=lat.action, config=dc, network_factory=nf)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sub(g, 'sd.').items()}
    res = dyn.load_state_dict(sd, strict=False)
    assert all('xnet' in k or k.startswith('networks.') for k in res.missing_keys), res.missing_keys
    assert not res.unexpected_keys, res.unexpected_keys
    dyn._eps_cache = {}
    dyn.set_masks(g['masks'])
    dyn.train()
    loss_fn = LatticeLoss(lat, cfgs.LossConfig(
        use_mixed_loss=bool(g['use_mixed_loss']), charge_weight=float(g['charge_weight']),
        plaq_weight=float(g['plaq_weight']), rmse_weight=float(g['rmse_weight'])))
    return dyn, lat, loss_fn


def check_leapfrog_layer_backward(act: str, device: str, dtype, tol: float):
    """LeapfrogLayer.forward_train / backward (dense layers, no conv / batch-norm) against
    torch.autograd of the same arithmetic written with torch ops -- network.py:430-551 of the
    reference is exactly this arithmetic under autograd."""
    import torch
    import l2hmc.configs as cfgs
    from l2hmc.network.pytorch.network import LeapfrogLayer
    torch.manual_seed(3)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        nb, xdim = 5, 24
        nc = cfgs.NetworkConfig(units=[16, 12], activation_fn=act, dropout_prob=0.0,
                                use_batch_norm=False)
        net = LeapfrogLayer(xshape=(nb, xdim), network_config=nc,
                            input_shapes={'x': xdim, 'v': xdim},
                            net_weight=cfgs.NetWeight(0.7, 1.1, 0.9)).to(device).train()
        with torch.no_grad():
            net.scale.coeff.normal_(0, 0.3)
            net.transf.coeff.normal_(0, 0.3)
        x = torch.randn(nb, xdim, device=device)
        v = torch.randn(nb, xdim, device=device)
        gs, gt, gq = (torch.randn(nb, xdim, device=device) for _ in range(3))
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        s, t, q, ctx = net.forward_train(x, v)
        dx, dv = net.backward(ctx, gs.clone(), gt.clone(), gq.clone())
        got = {n: p.grad.clone() for n, p in net.named_parameters()}
        # torch reference
        f = {'tanh': torch.tanh, 'relu': torch.relu, 'swish': torch.nn.functional.silu,
             'elu': torch.nn.functional.elu,
             'leaky_relu': lambda z: torch.nn.functional.leaky_relu(z, 0.01)}[act]
        xr, vr = x.clone().requires_grad_(True), v.clone().requires_grad_(True)
        for p in net.parameters():
            p.grad = None
        il = net.input_layer
        z = f(torch.nn.functional.linear(xr, il.xlayer.weight, il.xlayer.bias)
              + torch.nn.functional.linear(vr, il.vlayer.weight, il.vlayer.bias))
        for h in net.hidden_layers:
            z = f(torch.nn.functional.linear(z, h.weight, h.bias))
        sr = net.nw.s * torch.exp(net.scale.coeff) * torch.tanh(net.scale.layer(z))
        tr = net.nw.t * net.transl(z)
        qr = net.nw.q * torch.exp(net.transf.coeff) * torch.tanh(net.transf.layer(z))
        ((sr * gs).sum() + (tr * gt).sum() + (qr * gq).sum()).backward()
        worst = max(float((s - sr).abs().max()), float((t - tr).abs().max()),
                    float((q - qr).abs().max()))
        assert worst < tol, ('forward', worst)
        assert float((dx.reshape(nb, -1) - xr.grad).abs().max()) < tol
        assert float((dv.reshape(nb, -1) - vr.grad).abs().max()) < tol
        for n, p in net.named_parameters():
            e = float((got[n] - p.grad).abs().max())
            assert e < tol * max(1.0, float(p.grad.abs().max())), (n, e)
    finally:
        torch.set_default_dtype(old)


# ------------------------------------------------------------------ "modes" fixtures
# tests/golden/modes_*.npz (make_golden_modes.py): the reference built from (lattice, beta, seed)
# alone, in the configuration branches the first fixtures leave out.
MODES_U1 = ['modes_u1_nomerge', 'modes_u1_nomerge_b', 'modes_u1_shared', 'modes_u1_sep_nosplit',
            'modes_u1_shared_split']
MODES_SU3 = ['modes_su3_sep', 'modes_su3_nomerge', 'modes_su3_nomerge_b', 'modes_su3_bn']


def modes_net_kwargs(g):
    conv = None
    if g['conv_filters'].size:
        conv = {'filters': [int(i) for i in g['conv_filters']],
                'sizes': [int(i) for i in g['conv_sizes']],
                'pool': [int(i) for i in g['conv_pool']]}
    return dict(nunits=len(g['units']), activation=str(g['activation']), conv=conv,
                use_batch_norm=bool(g['use_batch_norm']))


def modes_state_dict(g):
    """state_dict of the fixture's trajectory: what the seed gave + the recorded perturbation."""
    sd = sub(g, 'init.')
    sd.update(sub(g, 'pert.'))
    return sd


def modes_oracle(g):
    """DynamicsOracle for a modes_* fixture: which LeapfrogLayer a sub-update calls follows
    use_separate_networks / use_split_xnets like dynamics.py:1112-1135."""
    sd = modes_state_dict(g)
    kw = modes_net_kwargs(g)
    group = str(g['group'])
    sep, split = bool(g['use_separate_networks']), bool(g['use_split_xnets'])
    nlf = int(g['nleapfrog'])
    dtype = np.float64 if group == 'SU3' else np.float32
    if group == 'SU3':
        kw.pop('conv')

    def vnet(step, x, f):
        return onet.leapfrog_layer(x, f, sub(sd, f'vnet.{step}.' if sep else 'vnet.'), **kw)

    def xnet(step, first, x, v):
        pre = 'xnet.'
        if sep:
            pre += f'{step}.'
            if split:
                pre += 'first.' if first else 'second.'
        return onet.leapfrog_layer(x, v, sub(sd, pre), **kw)
    L = tuple(int(i) for i in g['latvolume'])
    return DynamicsOracle(group, L, nlf, [sd[f'xeps.{i}'] for i in range(nlf)],
                          [sd[f'veps.{i}'] for i in range(nlf)], g['masks'], vnet=vnet,
                          xnet=xnet if group == 'U1' else None, use_ncp=bool(g['use_ncp']),
                          merge_directions=bool(g['merge_directions']), dtype=dtype)


def seed_all(s):
    import torch
    torch.manual_seed(int(s))
    np.random.seed(int(s))


def build_from_seed(g, verbose=True):
    """Product Dynamics from the fixture's configuration and SEED only (no weights loaded)."""
    import l2hmc.configs as cfgs
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.network.pytorch.network import NetworkFactory
    group = str(g['group'])
    L = [int(i) for i in g['latvolume']]
    nb = int(g['nchains'])
    seed_all(g['seed'])
    dc = cfgs.DynamicsConfig(nchains=nb, group=group, latvolume=L, nleapfrog=int(g['nleapfrog']),
                             eps=float(g['eps']), eps_hmc=float(g['eps']),
                             use_ncp=bool(g['use_ncp']), verbose=verbose,
                             use_split_xnets=bool(g['use_split_xnets']),
                             use_separate_networks=bool(g['use_separate_networks']),
                             merge_directions=bool(g['merge_directions']))
    kw = modes_net_kwargs(g)
    nc = cfgs.NetworkConfig(units=[int(i) for i in g['units']], activation_fn=kw['activation'],
                            dropout_prob=float(g['dropout_prob']),
                            use_batch_norm=kw['use_batch_norm'])
    cc = cfgs.ConvolutionConfig(**kw['conv']) if kw['conv'] else cfgs.ConvolutionConfig()
    if group == 'U1':
        from l2hmc.lattice.u1.pytorch.lattice import LatticeU1 as Lat
        xdim = dc.xdim
        dims = {'xnet': {'x': [xdim, 2], 'v': [xdim]}, 'vnet': {'x': [xdim], 'v': [xdim]}}
    else:
        from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3 as Lat
        xdim = int(np.prod(dc.xshape[1:-2])) * 8
        dims = {'xnet': {'x': [xdim], 'v': [xdim]}, 'vnet': {'x': [xdim], 'v': [xdim]}}
    spec = cfgs.InputSpec(xshape=tuple(dc.xshape), **dims)
    lat = Lat(nb, L)
    nf = NetworkFactory(input_spec=spec, network_config=nc, conv_config=cc,
                        net_weights=cfgs.NetWeights(x=cfgs.NetWeight(1., 1., 1.),
                                                    v=cfgs.NetWeight(1., 1., 1.)))
    dyn = Dynamics(potential_fn=lat.action, config=dc, network_factory=nf)
    return dyn, lat


def check_init_state(dyn, g, prefix='init.', bn_tol=2e-6):
    """Every parameter the seed determines is BIT-equal to the reference's; BatchNorm running
    statistics (one momentum step of the construction-time dummy forward, computed by different
    arithmetic) within bn_tol.  Returns the number of bit-equal tensors."""
    sd = dyn.state_dict()
    ref = sub(g, prefix)
    mine = {k for k in sd if not k.startswith('networks.')}
    assert mine == set(ref), (sorted(mine - set(ref))[:5], sorted(set(ref) - mine)[:5])
    n = 0
    for k, r in ref.items():
        a = sd[k].detach().cpu().numpy()
        assert a.shape == r.shape and a.dtype == r.dtype, (k, a.shape, r.shape, a.dtype, r.dtype)
        if k.endswith('running_mean') or k.endswith('running_var'):
            assert np.abs(a - r).max() <= bn_tol * max(1.0, np.abs(r).max()), (k, np.abs(a - r).max())
        else:
            assert np.array_equal(a, r), (k, float(np.abs(a - r).max()))
            n += 1
    return n


def apply_pert(dyn, g):
    """Load the fixture's recorded perturbation (coeffs, step sizes, BN statistics)."""
    import torch
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sub(g, 'pert.').items()}
    res = dyn.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    dyn._eps_cache = {}


def check_autograd_bridge_semantics(g, build, tol, sampler=True):
    """What an autograd caller may do around `loss.backward()` (dynamics/pytorch/autograd.py), on a
    train-step fixture `g` with `build(g) -> (dyn, lat, loss_fn)`:
      * a scaled loss gives scaled gradients (GradScaler.scale, trainer.py:1303-1304) and a second
        backward without zero_grad ACCUMULATES into p.grad;
      * torch.autograd.grad returns the same gradients without touching p.grad;
      * two recorded trajectories before one backward (the aux_weight > 0 sequence, :1338-1353) share
        one set of native-order weight shadows and both contribute;
      * a recorded trajectory that is dropped without a backward leaves nothing behind;
      * under torch.no_grad() train mode samples without a graph."""
    import torch
    inject = {'normals': g['normals'], 'u': g['u']}
    if 'forward' in g and not bool(g['merge_directions']):
        inject['forward'] = bool(g['forward'])
    beta = torch.tensor(float(g['beta']))

    def step(dyn, loss_fn, scale=1.0, twice=False):
        xin = dyn.g.compat_proj(dyn.unflatten(torch.from_numpy(g['x']).to(dyn.device)))
        dyn._inject = dict(inject)
        xout, m = dyn((xin, beta))
        loss = loss_fn(x_init=xin, x_prop=m['mc_states'].proposed.x, acc=m['acc'])
        if twice:
            dyn._inject = dict(inject)
            _, m2 = dyn((xin, beta))
            loss = loss + loss_fn(x_init=xin, x_prop=m2['mc_states'].proposed.x, acc=m2['acc'])
        dyn._inject = None
        return xout, m, loss * scale

    def grads(dyn):
        return {k: p.grad.detach().clone() for k, p in dyn.named_parameters() if p.grad is not None}

    def close(a, b, factor=1.0):
        assert a.keys() == b.keys() and len(a) > 4
        for k in a:
            sc = max(float(b[k].abs().max()), 1e-30)
            err = float((a[k] - factor * b[k]).abs().max())
            assert err <= tol * max(sc * abs(factor), 1e-6), (k, err, sc)

    dyn, lat, loss_fn = build(g)
    _, _, loss = step(dyn, loss_fn)
    loss.backward()
    base = grads(dyn)
    # scaled loss, accumulated on top of the first backward: 1 + 1024
    _, _, loss = step(dyn, loss_fn, scale=1024.0)
    loss.backward()
    close(grads(dyn), base, 1025.0)
    # torch.autograd.grad: same numbers, p.grad untouched
    before = grads(dyn)
    _, _, loss = step(dyn, loss_fn)
    names = [k for k, p in dyn.named_parameters() if k in base]
    got = torch.autograd.grad(loss, [dict(dyn.named_parameters())[k] for k in names])
    close({k: t for k, t in zip(names, got)}, base)
    close(grads(dyn), before)
    # two recorded trajectories, one backward
    for p in dyn.parameters():
        p.grad = None
    _, _, loss = step(dyn, loss_fn, twice=True)
    sess = dyn._ag_session
    assert sess.live == 2
    loss.backward()
    close(grads(dyn), base, 2.0)
    del loss
    assert sess.live == 0 and not any(n.native_active() for n in dyn.modules()
                                      if hasattr(n, 'native_active'))
    # a dropped trajectory, then a clean step
    for p in dyn.parameters():
        p.grad = None
    xo, m, loss = step(dyn, loss_fn)
    sess = dyn._ag_session
    assert sess.live == 1
    del xo, m, loss
    import gc
    gc.collect()
    assert sess.live == 0
    _, _, loss = step(dyn, loss_fn)
    loss.backward()
    close(grads(dyn), base)
    # retain_graph is refused loudly (the tape is released by the first backward)
    _, _, loss = step(dyn, loss_fn)
    loss.backward(retain_graph=True)
    try:
        loss.backward()
    except RuntimeError as e:
        assert 'reversed already' in str(e)
    else:
        raise AssertionError('second backward through one tape must raise')
    # no_grad: plain sampler, no graph
    if sampler and not any(getattr(n, 'training_needs_fresh_forward', lambda: False)()
                           for n in dyn.modules()):
        with torch.no_grad():
            xo, m, _ = step(dyn, loss_fn)
        assert xo.grad_fn is None and m['acc'].grad_fn is None


def check_half_train_step(g, route):
    """Mixed-precision train step on a u1_train_{fp16,bf16}* fixture (the REAL reference under
    `torch.autocast(dtype)` + `GradScaler`, tests/golden/make_golden_train.py half).

    route 'trainer': this build's tape with `set_net_precision(half)`, seeds scaled by the fixture's
    loss scale (what Trainer.train_step does with its LossScaler);
    route 'autograd': the reference caller's literal sequence -- `with torch.autocast(dtype): dynamics((x,
    beta))`, `scaler.scale(loss).backward(); scaler.unscale_(opt); scaler.step(opt); scaler.update()` with
    torch's own GradScaler and Adam.

    Returns distances in units the caller asserts on: `grad_vs_ref16` = |g - g_ref16| / |g_ref16| (global L2),
    `ref16_vs_ref32` = the reference's own 16-bit-vs-fp32 distance, per-parameter worst ratio, acc / loss
    differences, parameter update difference."""
    import torch
    from l2hmc.dynamics.pytorch import training as T
    half = {'fp16': torch.float16, 'bf16': torch.bfloat16}[str(g['half'])]
    dyn, lat, loss_fn = build_u1_train_dynamics(g)
    dev = dyn.device.type if isinstance(dyn.device, torch.device) else str(dyn.device).split(':')[0]
    scale = float(g['init_scale'])
    dyn._inject = {'normals': g['normals'], 'u': g['u']}
    x = torch.from_numpy(g['x'])
    beta = torch.tensor(float(g['beta']))
    xin = dyn.g.compat_proj(dyn.unflatten(x.to(dyn.device)))
    out = {}
    if route == 'trainer':
        dyn.set_net_precision(half)
        arena = T.ParamArena(dyn)
        arena.zero_grad()
        xout, metrics, loss = T.train_forward_backward(dyn, loss_fn, xin, beta, loss_weight=scale)
        grads = {k: p.grad.detach().cpu().numpy() / scale for k, p in dyn.named_parameters()
                 if p.grad is not None}
        arena.adam_step(lr=float(g['lr']), grad_scale=1.0 / scale)
    else:
        opt = torch.optim.Adam(dyn.parameters(), lr=float(g['lr']))
        scaler = torch.amp.GradScaler(dev, init_scale=scale)
        xin.requires_grad_(True)
        opt.zero_grad()
        with torch.autocast(dev, dtype=half):
            xout, metrics = dyn((xin, beta))
        assert dyn.net_precision == half
        loss = loss_fn(xin, metrics['mc_states'].proposed.x, metrics['acc'])
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in dyn.named_parameters()
                 if p.grad is not None}
        scaler.step(opt)
        scaler.update()
        out['scale_after'] = float(scaler.get_scale())
        loss = loss.detach()
    dyn._inject = None
    assert np.array_equal(metrics['acc_mask'].cpu().numpy(), g['acc_mask'])
    out['acc'] = float(np.abs(metrics['acc'].detach().cpu().numpy() - g['acc']).max())
    out['acc_ref16_vs_ref32'] = float(np.abs(g['acc32'] - g['acc']).max())
    out['loss'] = abs(float(loss) - float(g['loss'])) / max(1.0, abs(float(g['loss'])))
    xp = metrics['mc_states'].proposed.x.detach().cpu().numpy().reshape(g['x_prop'].shape)
    out['x_prop'] = float(np.abs(np.angle(np.exp(1j * (xp - g['x_prop'])))).max())
    out['x_prop_ref16_vs_ref32'] = float(np.abs(np.angle(np.exp(1j * (g['x_prop32'] - g['x_prop'])))).max())
    keys = [k for k in g if k.startswith('grad.')]
    n16 = np.sqrt(sum(float((g[k].astype(np.float64) ** 2).sum()) for k in keys))
    d = np.sqrt(sum(float(((grads[k[5:]].astype(np.float64) - g[k]) ** 2).sum()) for k in keys))
    dref = np.sqrt(sum(float(((g['grad32.' + k[5:]].astype(np.float64) - g[k]) ** 2).sum()) for k in keys))
    out['grad_vs_ref16'] = d / n16
    out['ref16_vs_ref32'] = dref / n16
    worst = 0.0
    for k in keys:
        nk = np.sqrt(float((g[k].astype(np.float64) ** 2).sum()))
        dk = np.sqrt(float(((grads[k[5:]].astype(np.float64) - g[k]) ** 2).sum()))
        worst = max(worst, dk / max(nk, 1e-3 * n16))
    out['grad_worst_param'] = worst
    sd = dyn.state_dict()
    wp = 0.0
    for k in g:
        if k.startswith('sd1.') and not k.endswith('num_batches_tracked'):
            name = k[4:]
            gk = 'grad.' + name if ('grad.' + name) in g else 'grad.networks.' + name
            diff = np.abs(sd[name].detach().cpu().numpy() - g[k])
            if gk in g:                       # Adam's first step is lr * sign(g) where g is resolved
                diff = np.where(np.abs(g[gk]) > 1e-2 * np.abs(g[gk]).max(), diff, 0.0)
            wp = max(wp, float(diff.max()) if diff.size else 0.0)
    out['param_abs'] = wp
    return out


def assert_half_train_step(g, name, route, out):
    """Tolerances of check_half_train_step: the product must be as close to the reference's 16-bit step as
    the reference's 16-bit step is to its own fp32 step (x 2 for the gradients, x 3 for acc = exp(-dH));
    without BatchNorm the rounding points are reproduced and the product is much closer than that."""
    assert out['grad_vs_ref16'] <= max(2.0 * out['ref16_vs_ref32'], 2e-3), out
    assert out['acc'] <= max(3.0 * out['acc_ref16_vs_ref32'], 2e-3), out
    assert out['x_prop'] <= max(3.0 * out['x_prop_ref16_vs_ref32'], 1e-3), out
    assert out['loss'] <= 1e-2, out
    if name in ('u1_train_fp16', 'u1_train_fp16_conv'):
        assert out['grad_vs_ref16'] <= 5e-4 and out['param_abs'] <= 1e-6, out
    else:
        assert out['param_abs'] <= 2.1 * float(g['lr']), out        # Adam's first step: lr * sign(g)
    if route == 'autograd':
        assert out['scale_after'] == float(g['scale_after']), out
