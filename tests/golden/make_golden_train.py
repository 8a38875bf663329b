#!/usr/bin/env python3
"""Golden *training-step* vectors (loss, per-parameter gradients, BatchNorm running statistics,
parameters after one Adam step) from the REAL reference in train mode.

    bash tests/golden/setup_reference_env.sh
    PYTHONPATH=/tmp/oracle_stubs:/tmp/oracle/src python3 tests/golden/make_golden_train.py f64
    PYTHONPATH=/tmp/oracle_stubs:/tmp/oracle/src python3 tests/golden/make_golden_train.py f32
    PYTHONPATH=/tmp/oracle_stubs:/tmp/oracle/src python3 tests/golden/make_golden_train.py su3
    PYTHONPATH=/tmp/oracle_stubs:/tmp/oracle/src python3 tests/golden/make_golden_train.py nomerge
    PYTHONPATH=/tmp/oracle_stubs:/tmp/oracle/src python3 tests/golden/make_golden_train.py half

`f64`: U(1) 4x4, dense networks, float64 default dtype (tight tolerance); `f32`: U(1) 4x6 with
the conv stack (a pooling layer), float32.  Sequence = Trainer.train_step of the reference
(trainers/pytorch/trainer.py:1316-1367): compat_proj -> dynamics((x, beta)) in train mode ->
LatticeLoss(x_init, x_prop, acc) -> loss.backward() -> Adam(lr_init).step().

`half`: the reference's mixed-precision step (trainer.py:211-219, 1276-1280, 1303-1313): the forward under
`torch.autocast(dtype=float16 | bfloat16)` (device type 'cpu' -- what runs here), then
`grad_scaler.scale(loss).backward(); grad_scaler.unscale_(optimizer); grad_scaler.step(optimizer);
grad_scaler.update()`.  Also stores the fp32 gradients of the same draw (`grad32.*`): the reference's own
fp16-vs-fp32 distance is the yardstick of the product's tolerance.
"""
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
WHICH = sys.argv[1]
if WHICH in ('f64', 'su3', 'su3c1', 'nomerge'):
    torch.set_default_dtype(torch.float64)
sys.argv = [sys.argv[0], 'su3' if WHICH in ('f64', 'su3', 'su3c1', 'nomerge') else 'u1']
sys.path.insert(0, OUT)
import make_golden as mg  # noqa: E402  (imports the reference, sets nothing else)
import l2hmc.configs as cfgs  # noqa: E402
from l2hmc.loss.pytorch.loss import LatticeLoss  # noqa: E402

npy = mg.npy


def train_case(name, L, nb, nlf, units, act, conv, beta, seed, bn, loss_cfg, eps=0.1, lr=1e-3,
               merge=True, want_forward=None, half=None, init_scale=1024.0):
    """merge=False: the reference trains on `apply_transition` -- one direction, drawn with
    torch.rand(1) before the momenta (dynamics.py:704-742), accept probability with the swapped
    arguments of transition_kernel (:1053-1057); want_forward picks a draw seed by direction."""
    dyn, lat = mg.build_dynamics('U1', L, nb, nlf=nlf, eps=eps, units=units, act=act, conv=conv,
                                 sep=True, split=True, bn=bn, dropout=0.0, seed=seed, merge=merge)
    mg.perturb(dyn, seed + 1)
    bt = torch.tensor(beta)
    mg.seed_all(seed + 2)
    x = lat.random()
    for _ in range(40):
        x, _m = dyn.apply_transition_hmc((x, bt), eps=0.2, nleapfrog=5)
        x = dyn.g.compat_proj(dyn.unflatten(x)).detach()
    # .copy(): Tensor.numpy() aliases the live parameters, which Adam then updates in place
    sd0 = {k: a.copy() for k, a in mg.state_dict_np(dyn).items()
           if not k.startswith('networks.')}
    dyn.train()
    loss_fn = LatticeLoss(lat, loss_cfg)
    opt = torch.optim.Adam(dyn.parameters(), lr=lr)
    dseed, fwd = seed + 3, True
    while True:
        mg.seed_all(dseed)
        if not merge:
            fwd = bool(torch.rand(1) > 0.5)
        nrm = torch.randn(nb, 2, *L)
        u = torch.rand(nb)
        if merge or want_forward is None or fwd == want_forward:
            break
        dseed += 1000
    extra = {}
    if half is not None:
        # a draw whose accept decisions are not marginal (the 16-bit forward moves acc by ~1e-3)
        bufs = {k: b.detach().clone() for k, b in dyn.named_buffers()}
        for cand in range(dseed, dseed + 200):
            mg.seed_all(cand)
            nrm_c = torch.randn(nb, 2, *L)
            u_c = torch.rand(nb)
            mg.seed_all(cand)
            with torch.autocast('cpu', dtype=half):          # (the reference's force needs grad mode)
                _xo, m_c = dyn((dyn.g.compat_proj(x.reshape(dyn.xshape)), bt))
            with torch.no_grad():
                for k, b in dyn.named_buffers():
                    b.copy_(bufs[k])
            a_c = m_c['acc'].detach()
            if float((a_c - u_c).abs().min()) > 0.05 and 0 < float((a_c > u_c).sum()) < nb \
                    and int(((a_c > 0.02) & (a_c < 0.98)).sum()) >= nb // 2:
                dseed, nrm, u = cand, nrm_c, u_c
                break
        else:
            raise SystemExit(f'{name}: no draw with an accept margin')
        # the fp32 step on the same draw first (gradients only; BatchNorm statistics restored afterwards)
        bufs = {k: b.detach().clone() for k, b in dyn.named_buffers()}
        mg.seed_all(dseed)
        x32 = dyn.g.compat_proj(x.reshape(dyn.xshape)).detach().requires_grad_(True)
        opt.zero_grad()
        _xo, m32 = dyn((x32, bt))
        l32 = loss_fn(x32, m32['mc_states'].proposed.x, m32['acc'])
        l32.backward()
        for n, p in dyn.named_parameters():
            extra['grad32.' + n] = (npy(p.grad).copy() if p.grad is not None else np.zeros(tuple(p.shape)))
        extra.update(loss32=npy(l32), acc32=npy(m32['acc']), x_prop32=npy(m32['mc_states'].proposed.x))
        with torch.no_grad():
            for k, b in dyn.named_buffers():
                b.copy_(bufs[k])
    mg.seed_all(dseed)
    xinit = dyn.g.compat_proj(x.reshape(dyn.xshape)).detach()
    xinit.requires_grad_(True)
    opt.zero_grad()
    if half is None:
        xout, m = dyn((xinit, bt))
    else:
        with torch.autocast('cpu', dtype=half):            # autocast_context_train (trainer.py:1276-1280)
            xout, m = dyn((xinit, bt))
    mc = m['mc_states']
    assert torch.equal(mc.init.v.detach(), nrm.reshape(nb, -1))
    loss = loss_fn(xinit, mc.proposed.x, m['acc'])
    grads = {}
    if half is None:
        loss.backward()
        for n, p in dyn.named_parameters():
            grads['grad.' + n] = (npy(p.grad).copy() if p.grad is not None
                                  else np.zeros(tuple(p.shape)))
        opt.step()
    else:
        scaler = torch.amp.GradScaler('cpu', init_scale=init_scale)
        scaler.scale(loss).backward()                      # backward_step (trainer.py:1303-1313)
        scaler.unscale_(opt)
        for n, p in dyn.named_parameters():
            grads['grad.' + n] = (npy(p.grad).copy() if p.grad is not None
                                  else np.zeros(tuple(p.shape)))
        scaler.step(opt)
        scaler.update()
        extra.update(init_scale=init_scale, scale_after=float(scaler.get_scale()),
                     half='fp16' if half == torch.float16 else 'bf16')
        finite = all(np.isfinite(a).all() for a in grads.values())
        extra['skipped'] = not finite                      # GradScaler.step skips the optimiser on inf / nan
        if not finite:
            grads = {k: np.zeros_like(a) for k, a in grads.items()}
    sd1 = {k: a for k, a in mg.state_dict_np(dyn).items() if not k.startswith('networks.')}
    if half is not None and extra['skipped']:
        # an overflowing scale: only the GradScaler bookkeeping is the result (parameters untouched)
        assert all(np.array_equal(sd0[k], sd1[k]) for k in sd0)
        mg.save(name, init_scale=init_scale, scale_after=extra['scale_after'], skipped=True,
                same_draw_as=name.replace('_overflow', ''))
        print(f'  {name}: overflow, step skipped, scale {init_scale} -> {extra["scale_after"]}')
        return
    mg.save(name, latvolume=np.array(L), beta=beta, nleapfrog=nlf, x=npy(x), normals=npy(nrm),
            u=npy(u), masks=np.stack([npy(mm)[0] for mm in dyn.masks]),
            x_prop=npy(mc.proposed.x), x_out=npy(xout), acc=npy(m['acc']),
            acc_mask=npy(m['acc_mask']), sumlogdet=npy(m['sumlogdet']), loss=npy(loss),
            lr=lr, charge_weight=loss_cfg.charge_weight, use_mixed_loss=loss_cfg.use_mixed_loss,
            units=np.array(units), activation=act, use_batch_norm=bn,
            merge_directions=merge, forward=fwd,
            conv_filters=np.array(conv['filters'] if conv else []),
            conv_sizes=np.array(conv['sizes'] if conv else []),
            conv_pool=np.array(conv['pool'] if conv else []),
            **{'sd.' + k: a for k, a in sd0.items()}, **{'sd1.' + k: a for k, a in sd1.items()},
            **grads, **extra)
    gn = np.sqrt(sum(float((g ** 2).sum()) for g in grads.values()))
    print(f'  {name}: loss {float(loss):.6g} acc {npy(m["acc"])[:6]} |grad| {gn:.4g}')
    if half is not None and extra['skipped']:
        print(f'    overflow: step skipped, scale {init_scale} -> {extra["scale_after"]}')
    elif half is not None:
        d = np.sqrt(sum(float(((grads[k] - extra['grad32.' + k[5:]]) ** 2).sum()) for k in grads))
        print(f'    |grad16 - grad32| / |grad32| = {d / gn:.3e}; |u - acc| min '
              f'{float(np.abs(npy(u) - npy(m["acc"])).min()):.3f}; scale after {extra["scale_after"]}')


def su3_train_case(name, L, nb, nlf, units, act, beta, seed, bn, loss_cfg, eps=0.006, lr=1e-3,
                   c1=0.0, merge=True):
    """SU(3): vnet only (the xnet is never called, dynamics.py:1420-1425).  Start near
    equilibrium and pick a draw whose acceptance is not saturated so that the gradient also
    flows through acc.  The reference's train_step begins with compat_proj (projectSU)."""
    from l2hmc.group.su3.pytorch import utils as U
    dyn, lat = mg.build_dynamics('SU3', L, nb, nlf=nlf, eps=eps, units=units, act=act, bn=bn,
                                 dropout=0.0, seed=seed, c1=c1, merge=merge)
    mg.perturb(dyn, seed + 1)
    bt = torch.tensor(beta)
    mg.seed_all(seed + 2)
    x = torch.matrix_exp(0.3 * U.randTAH3((nb, 4, *L)))
    shape = tuple(x.shape[:-2])
    sd0 = {k: a.copy() for k, a in mg.state_dict_np(dyn).items()
           if not k.startswith('networks.')}
    dyn.train()
    loss_fn = LatticeLoss(lat, loss_cfg)
    opt = torch.optim.Adam(dyn.parameters(), lr=lr)
    best = None
    for sd in range(seed + 3, seed + 60):
        mg.seed_all(sd)
        xo, m = dyn((dyn.g.compat_proj(x.reshape(dyn.xshape)), bt))
        a = npy(m['acc'])
        inside = int(((a > 0.02) & (a < 0.98)).sum())
        if best is None or inside > best[0]:
            best = (inside, sd)
        if inside == nb:
            break
    sd = best[1]
    mg.seed_all(sd)
    fwd = True
    if not merge:
        fwd = bool(torch.rand(1) > 0.5)
    nrm = torch.stack([torch.randn(shape) for _ in range(8)])
    u = torch.rand(nb)
    mg.seed_all(sd)
    xinit = dyn.g.compat_proj(x.reshape(dyn.xshape)).detach()
    xinit.requires_grad_(True)
    opt.zero_grad()
    xout, m = dyn((xinit, bt))
    mc = m['mc_states']
    assert np.array_equal(npy((m['acc'] > u).float()), npy(m['acc_mask'])), 'uniform replay'
    loss = loss_fn(xinit, mc.proposed.x, m['acc'])
    loss.backward()
    grads = {}
    for n, p in dyn.named_parameters():
        grads['grad.' + n] = (npy(p.grad).copy() if p.grad is not None
                              else np.zeros(tuple(p.shape)))
    opt.step()
    sd1 = {k: a for k, a in mg.state_dict_np(dyn).items() if not k.startswith('networks.')}
    mg.save(name, latvolume=np.array(L), beta=beta, nleapfrog=nlf, x=npy(x), normals=npy(nrm),
            u=npy(u), masks=np.stack([npy(mm)[0] for mm in dyn.masks]),
            x_prop=npy(mc.proposed.x), x_out=npy(xout), acc=npy(m['acc']),
            acc_mask=npy(m['acc_mask']), sumlogdet=npy(m['sumlogdet']), loss=npy(loss), lr=lr,
            charge_weight=loss_cfg.charge_weight, plaq_weight=loss_cfg.plaq_weight,
            rmse_weight=loss_cfg.rmse_weight, use_mixed_loss=loss_cfg.use_mixed_loss,
            units=np.array(units), activation=act, use_batch_norm=bn, c1=c1,
            merge_directions=merge, forward=fwd,
            # the SU(3) xnet is never called: no gradient, no update -- not stored
            **{'sd.' + k: a for k, a in sd0.items() if 'xnet' not in k},
            **{'sd1.' + k: a for k, a in sd1.items() if 'xnet' not in k},
            **{k: a for k, a in grads.items() if 'xnet' not in k})
    assert all(float(np.abs(a).max()) == 0.0 for k, a in grads.items() if 'xnet' in k)
    gn = np.sqrt(sum(float((g ** 2).sum()) for g in grads.values()))
    print(f'  {name}: loss {float(loss):.6g} acc {npy(m["acc"])} |grad| {gn:.4g}')


if __name__ == '__main__':
    if WHICH == 'su3':
        su3_train_case('su3_train', (2, 3, 2, 4), 3, 2, [6], 'tanh', beta=6.0, seed=400, bn=False,
                       loss_cfg=cfgs.LossConfig(use_mixed_loss=True, charge_weight=0.05,
                                                rmse_weight=0.1, plaq_weight=0.1))
    elif WHICH == 'su3c1':
        # improved action: rectangles in H (accept probability) only, Wilson force in the
        # integrator (dynamics.py:134-135); eps small enough for a non-saturated acceptance
        su3_train_case('su3_train_c1', (2, 3, 2, 4), 3, 2, [6], 'tanh', beta=6.0, seed=430, bn=False,
                       loss_cfg=cfgs.LossConfig(use_mixed_loss=True, charge_weight=0.05,
                                                rmse_weight=0.1, plaq_weight=0.1),
                       eps=0.004, c1=-0.331)
    elif WHICH == 'nomerge':
        lc = cfgs.LossConfig(use_mixed_loss=True, charge_weight=0.01)
        train_case('u1_train_nomerge_fwd', (4, 4), 5, 2, [8, 6], 'leaky_relu', None, beta=2.0, seed=500,
                   bn=True, loss_cfg=lc, merge=False, want_forward=True)
        train_case('u1_train_nomerge_bwd', (4, 6), 5, 3, [8], 'tanh', None, beta=3.0, seed=520,
                   bn=False, loss_cfg=cfgs.LossConfig(use_mixed_loss=False, charge_weight=0.5),
                   merge=False, want_forward=False)
        su3_train_case('su3_train_nomerge', (2, 3, 2, 4), 3, 2, [6], 'tanh', beta=6.0, seed=540, bn=False,
                       loss_cfg=cfgs.LossConfig(use_mixed_loss=True, charge_weight=0.05,
                                                rmse_weight=0.1, plaq_weight=0.1), merge=False)
    elif WHICH == 'half':
        lc = cfgs.LossConfig(use_mixed_loss=True, charge_weight=0.01)
        train_case('u1_train_fp16', (4, 8), 8, 2, [16, 16], 'relu', None, beta=2.0, seed=600, bn=False,
                   loss_cfg=lc, half=torch.float16)
        train_case('u1_train_fp16_bn', (4, 6), 6, 2, [8, 6], 'leaky_relu', None, beta=2.5, seed=620, bn=True,
                   loss_cfg=lc, half=torch.float16, init_scale=16.0)
        train_case('u1_train_bf16', (4, 6), 5, 3, [8], 'tanh', None, beta=3.0, seed=640, bn=False,
                   loss_cfg=cfgs.LossConfig(use_mixed_loss=False, charge_weight=0.5), half=torch.bfloat16)
        train_case('u1_train_fp16_conv', (4, 6), 5, 2, [8, 6], 'leaky_relu',
                   {'filters': [2, 3, 4], 'sizes': [3, 2, 2], 'pool': [2, 2, 2]}, beta=2.5, seed=660, bn=False,
                   loss_cfg=lc, half=torch.float16, init_scale=16.0)
        # default GradScaler scale 2^16: the fp16 backward overflows, the step is skipped, the scale halves
        train_case('u1_train_fp16_overflow', (4, 8), 8, 2, [16, 16], 'relu', None, beta=2.0, seed=600, bn=False,
                   loss_cfg=lc, half=torch.float16, init_scale=65536.0)
    elif WHICH == 'f64':
        train_case('u1_train_f64', (4, 4), 6, 2, [8, 6], 'leaky_relu', None, beta=2.0, seed=300,
                   bn=True, loss_cfg=cfgs.LossConfig(use_mixed_loss=True, charge_weight=0.01))
        train_case('u1_train_f64_plain', (4, 6), 5, 3, [8], 'tanh', None, beta=3.0, seed=320,
                   bn=False, loss_cfg=cfgs.LossConfig(use_mixed_loss=False, charge_weight=0.5))
    else:
        train_case('u1_train_conv', (4, 6), 5, 2, [8, 6], 'leaky_relu',
                   {'filters': [2, 3, 4], 'sizes': [3, 2, 2], 'pool': [2, 2, 2]}, beta=2.5,
                   seed=340, bn=True,
                   loss_cfg=cfgs.LossConfig(use_mixed_loss=True, charge_weight=0.01))
