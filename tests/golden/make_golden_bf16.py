#!/usr/bin/env python3
"""16-bit goldens for BASELINE cfg-3 ("fp16 nets / fp32 action"): the REAL reference's U(1)
``Dynamics`` run under ``torch.autocast('cpu', dtype=torch.bfloat16 | torch.float16)`` -- the
context manager the reference's trainer wraps around the forward step
(trainers/pytorch/trainer.py:211-219, 1276-1280; it enables it on CUDA only, the CPU device type
is what can be run here).

    bash tests/golden/setup_reference_env.sh
    PYTHONPATH=/tmp/oracle_stubs:/tmp/oracle/src python3 tests/golden/make_golden_bf16.py [bf16|fp16]

(default bf16 -> u1_bf16*.npz; ``fp16`` -> u1_fp16*.npz, cfg-3's stated dtype.)
Stores inputs (weights, masks, x, draws) and the reference's network outputs, sub-updates and
merged-trajectory results; tests/test_dynamics_gpu.py::test_u1_half_reference_golden compares
``Dynamics.set_net_precision('bf16' | 'fp16')`` with them.
"""
import os
import sys

import numpy as np
import torch

HD = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
HDT = {'bf16': torch.bfloat16, 'fp16': torch.float16}[HD]
sys.argv = [sys.argv[0], 'u1']
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402  (imports the reference, float32 default)
from l2hmc.dynamics.pytorch.dynamics import State  # noqa: E402

npy = G.npy


def case(name, L, nb, nlf, units, act, beta, seed, conv=None):
    dyn, lat = G.build_dynamics('U1', L, nb, nlf=nlf, eps=0.1, units=units, act=act, conv=conv,
                                sep=True, split=True, bn=False, dropout=0.0, seed=seed)
    G.perturb(dyn, seed + 1)
    bt = torch.tensor(beta)
    G.seed_all(seed + 2)
    x = lat.random()
    for _ in range(30):       # thermalise in fp32 with the reference's own HMC
        x, _m = dyn.apply_transition_hmc((x, bt), eps=0.2, nleapfrog=5)
        x = dyn.g.compat_proj(dyn.unflatten(x)).detach()
    ac = torch.autocast('cpu', dtype=HDT)
    best = None
    for sd in range(seed + 3, seed + 103):       # draw with the widest accept margin
        G.seed_all(sd)
        nrm = torch.randn(nb, 2, *L)
        u = torch.rand(nb)
        G.seed_all(sd)
        with ac:
            xo, m = dyn((x, bt))
        margin = float((m['acc'] - u).abs().min())
        mixed = 0 < float(m['acc_mask'].sum()) < nb
        if mixed and (best is None or margin > best[0]):
            best = (margin, sd)
        if mixed and margin > 0.05:
            break
    G.seed_all(best[1])
    nrm = torch.randn(nb, 2, *L)
    u = torch.rand(nb)
    G.seed_all(best[1])
    with ac:
        xo, m = dyn((x, bt))
        mc = m['mc_states']
        v = mc.init.v
        st = State(x, v, bt)
        force = dyn.grad_potential(x.clone(), bt)
        sv, tv, qv = dyn._call_vnet(0, (x, force))
        st_v, ld_v = dyn._update_v_fwd(0, st)
        m0, mb0 = dyn._get_mask(0)
        xm = dyn.unflatten(m0) * x
        sx, tx, qx = dyn._call_xnet(0, (xm, v), first=True)
        st_x, ld_x = dyn._update_x_fwd(0, st, m0, first=True)
        st_xb, ld_xb = dyn._update_x_bwd(0, st, mb0, first=False)
    # the same trajectory in fp32 (how far the 16-bit type moves the result: scale for the tolerances)
    G.seed_all(best[1])
    xo32, m32 = dyn((x, bt))
    f32 = lambda t: npy(t.float())
    sd_ = {k: a for k, a in G.state_dict_np(dyn).items() if not k.startswith('networks.')}
    print(f'  {name}: dtypes s {sv.dtype} t {tv.dtype} q {qv.dtype}  acc {npy(m["acc"])[:6]} '
          f'margin {best[0]:.3f}  |acc - acc32| {float((m["acc"] - m32["acc"]).abs().max()):.3e}')
    G.save(name, latvolume=np.array(L), beta=beta, nleapfrog=nlf, x=npy(x), normals=npy(nrm),
           u=npy(u), masks=np.stack([npy(mm)[0] for mm in dyn.masks]),
           x_prop=f32(mc.proposed.x), v_prop=f32(mc.proposed.v), x_out=f32(xo),
           acc=f32(m['acc']), acc_mask=f32(m['acc_mask']), sumlogdet=f32(m['sumlogdet']),
           energy=f32(m['energy']), acc_fp32=npy(m32['acc']), x_out_fp32=npy(xo32),
           force=f32(force), vnet_s=f32(sv), vnet_t=f32(tv), vnet_q=f32(qv), v_fwd=f32(st_v.v),
           logdet_v_fwd=f32(ld_v), xnet_s=f32(sx), xnet_t=f32(tx), xnet_q=f32(qx),
           x_fwd=f32(st_x.x), logdet_x_fwd=f32(ld_x), x_bwd=f32(st_xb.x), logdet_x_bwd=f32(ld_xb),
           units=np.array(units), activation=act, use_batch_norm=False,
           conv_filters=np.array(conv['filters'] if conv else []),
           conv_sizes=np.array(conv['sizes'] if conv else []),
           conv_pool=np.array(conv['pool'] if conv else []),
           **{'sd.' + k: a for k, a in sd_.items()})


if __name__ == '__main__':
    case(f'u1_{HD}', (8, 8), 32, 2, [32, 32], 'leaky_relu', beta=2.0, seed=300)
    case(f'u1_{HD}_tanh', (8, 16), 12, 2, [24], 'tanh', beta=3.0, seed=320)
    case(f'u1_{HD}_conv', (8, 8), 6, 2, [8, 8], 'relu', beta=2.0, seed=340,
         conv={'filters': [2, 4], 'sizes': [3, 2], 'pool': [2, 2]})
