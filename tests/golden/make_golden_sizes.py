#!/usr/bin/env python3
"""Reference goldens at the BASELINE U(1) shapes themselves.

cfg-2 (16 x 16, beta 4, nleapfrog 8, fp32) and cfg-3 (64 x 64, beta 6, nleapfrog 8, fp16 nets /
fp32 action) with the networks those configs run -- the reference's default conv stack
[8,16,32,64,128] + units [16,16,16,16] (its 51 200 -> 8192 Linear is 1.7 GB per network at 64 x 64)
and dense units [256, 256] -- are far too big to store, so the weights are counter-based numbers
(tests/golden/seeded.py) that the GPU tests regenerate on the device.  The REAL reference's
``Dynamics`` runs one merged trajectory on TWO chains here; the fixture keeps only the inputs
(x, normals, u, masks, step sizes, the weight seed) and the reference's outputs.  The GPU tests
(tests/test_sizes_gpu.py::test_cfg2_* / test_cfg3_*) run 2048 / 8192 chains that are copies of
these two and compare every chain on the device.

The accept uniforms are chosen after a first pass (u = midpoint between acc and 0 resp. 1, one
accept and one reject with the widest margin) and handed to the reference through
``torch.rand_like`` -- the reference code itself is unmodified.

    bash tests/golden/setup_reference_env.sh
    PYTHONPATH=/tmp/oracle_stubs:/tmp/oracle/src python3 tests/golden/make_golden_sizes.py [case ...]
"""
import contextlib
import os
import sys
import time

import numpy as np
import torch

CASES = sys.argv[1:]
sys.argv = [sys.argv[0], 'u1']
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402  (imports the reference, float32 default)
import seeded  # noqa: E402

npy = G.npy
CONV_DEFAULT = {'filters': [8, 16, 32, 64, 128], 'sizes': [5, 3, 3, 3, 2], 'pool': [2, 2, 2, 2, 2]}


def case(name, L, nlf, units, act, conv, bn, beta, seed, head_scale, eps, hd=None, therm=60, therm_eps=0.1,
         sep=True, split=True, save=True):
    t0 = time.time()
    nb = 2
    dyn, lat = G.build_dynamics('U1', L, nb, nlf=nlf, eps=eps, units=units, act=act, conv=conv,
                                sep=sep, split=split, bn=bn, dropout=0.2, seed=seed)
    bt = torch.tensor(beta)
    G.seed_all(seed + 2)
    x = lat.random()
    dyn((x, bt))                           # materialise the lazy layers (the force is autograd)
    n_el = seeded.fill_state_dict(dyn.state_dict(), seed, head_scale)
    with torch.no_grad():
        for i, (xe, ve) in enumerate(zip(dyn.xeps, dyn.veps)):
            xe.mul_(1.0 + 0.002 * (i + 1))
            ve.mul_(1.0 - 0.001 * (i + 1))
    print(f'  {name}: {n_el / 1e6:.1f} M seeded weights ({time.time() - t0:.0f} s)', flush=True)
    for _ in range(therm):                 # thermalise in fp32 with the reference's own HMC
        x, _m = dyn.apply_transition_hmc((x, bt), eps=therm_eps, nleapfrog=int(round(1.0 / therm_eps)))
        x = dyn.g.compat_proj(dyn.unflatten(x)).detach()
    print(f'  {name}: thermalised, last HMC acc {npy(_m["acc"])} plaq {npy(lat.plaqs(x)).mean():.4f}',
          flush=True)
    ac = (torch.autocast('cpu', dtype={'fp16': torch.float16, 'bf16': torch.bfloat16}[hd])
          if hd else contextlib.nullcontext())
    sd_draw = seed + 3
    G.seed_all(sd_draw)
    nrm = torch.randn(nb, 2, *L)
    G.seed_all(sd_draw)
    with ac:
        _xo, m0 = dyn((x, bt))
    acc0 = npy(m0['acc'])
    print(f'  {name}: first pass acc {acc0} ({time.time() - t0:.0f} s)', flush=True)
    if not (np.all(acc0 > 0.02) and np.all(acc0 < 0.98)):
        print(f'  {name}: acc not inside (0, 1) -- recalibrate head_scale / eps')
        if save:
            raise SystemExit(1)
        return
    u = np.where(np.arange(nb) % 2 == 0, 0.5 * (1.0 + acc0), 0.5 * acc0).astype(np.float32)
    real_rand_like = torch.rand_like

    def rand_like(t, *a, **k):
        assert tuple(t.shape) == (nb,)
        return torch.from_numpy(u).to(t.dtype)
    G.seed_all(sd_draw)
    torch.rand_like = rand_like
    try:
        with ac:
            xo, m = dyn((x, bt))
    finally:
        torch.rand_like = real_rand_like
    mc = m['mc_states']
    assert torch.equal(mc.init.v.float().reshape(nb, -1), nrm.reshape(nb, -1))
    assert npy(m['acc_mask']).tolist() == [0.0, 1.0], npy(m['acc_mask'])
    assert np.array_equal(npy(m['acc']), acc0)
    # the fp32 trajectory on the same draws: yardstick for the 16-bit tolerances
    extra = {}
    if hd:
        G.seed_all(sd_draw)
        torch.rand_like = rand_like
        try:
            xo32, m32 = dyn((x, bt))
        finally:
            torch.rand_like = real_rand_like
        extra = dict(acc_fp32=npy(m32['acc']), x_prop_fp32=npy(m32['mc_states'].proposed.x))
        print(f'  {name}: |acc - acc32| {np.abs(acc0 - extra["acc_fp32"]).max():.3e}')
    f32 = lambda t: npy(t.float())
    margin = float(np.abs(acc0 - u).min())
    print(f'  {name}: acc {acc0} u {u} margin {margin:.3f} ({time.time() - t0:.0f} s)', flush=True)
    if not save:
        return
    G.save(name, latvolume=np.array(L), beta=beta, nleapfrog=nlf, eps=eps, x=npy(x),
           normals=npy(nrm), u=u, masks=np.stack([npy(mm)[0] for mm in dyn.masks]),
           xeps=np.array([npy(e) for e in dyn.xeps]), veps=np.array([npy(e) for e in dyn.veps]),
           weight_seed=seed, head_scale=head_scale, precision=hd or 'fp32',
           x_prop=f32(mc.proposed.x), v_prop=f32(mc.proposed.v), x_out=f32(xo),
           acc=f32(m['acc']), acc_mask=f32(m['acc_mask']), sumlogdet=f32(m['sumlogdet']),
           energy=f32(m['energy']), logdet=f32(m['logdet']),
           units=np.array(units), activation=act, use_batch_norm=bn,
           use_separate_networks=sep, use_split_xnets=split,
           conv_filters=np.array(conv['filters'] if conv else []),
           conv_sizes=np.array(conv['sizes'] if conv else []),
           conv_pool=np.array(conv['pool'] if conv else []), **extra)


ALL = {
    # cfg-2: conf/dynamics default (nleapfrog 8, separate + split networks), conf/network default
    'u1_cfg2_conv': dict(L=(16, 16), nlf=8, units=[16, 16, 16, 16], act='leaky_relu',
                         conv=CONV_DEFAULT, bn=True, beta=4.0, seed=2023, head_scale=0.25, eps=0.05),
    'u1_cfg2_dense': dict(L=(16, 16), nlf=8, units=[16, 16, 16, 16], act='leaky_relu',
                          conv=None, bn=True, beta=4.0, seed=2021, head_scale=0.25, eps=0.05),
    # cfg-3: fp16 nets / fp32 action
    'u1_cfg3_fp16_dense': dict(L=(64, 64), nlf=8, units=[256, 256], act='leaky_relu', conv=None,
                               bn=True, beta=6.0, seed=3030, head_scale=0.1, eps=0.05, hd='fp16', therm=150, therm_eps=0.04),
    'u1_cfg3_fp16_conv': dict(L=(64, 64), nlf=2, units=[16, 16, 16, 16], act='leaky_relu',
                              conv=CONV_DEFAULT, bn=True, beta=6.0, seed=3031, head_scale=0.1,
                              eps=0.05, hd='fp16', therm=150, therm_eps=0.04),
}

if __name__ == '__main__':
    dry = 'dry' in CASES
    for k in [c for c in CASES if c != 'dry'] or list(ALL):
        kw = dict(ALL[k.split(':')[0]])
        for ov in k.split(':')[1:]:          # name:head_scale=0.2:eps=0.03  (calibration runs)
            a, b = ov.split('=')
            kw[a] = type(kw[a])(b)
        case(k.split(':')[0], save=not dry, **kw)
