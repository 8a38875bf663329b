"""``python -m l2hmc key=value ...`` -- counterpart of the reference's hydra entry point
(src/l2hmc/__main__.py:100-195): compose the config tree with dotted overrides, build the
Experiment, train if ``steps.nera > 0 and steps.nepoch > 0``, evaluate the trained sampler and
the plain-HMC baseline for ``steps.test`` steps, and report the model improvement
(``<|dQint|>_trained / <|dQint|>_HMC``, utils/plot_helpers.py:189-263) -- as one JSON object
instead of plots / wandb artefacts.

    python -m l2hmc dynamics.group=SU3 dynamics.latvolume=[4,4,4,4] steps.nera=1 steps.nepoch=5
    python -m l2hmc --config-name su3test outdir=runs/su3test     # flat config + dataset dump
"""
from __future__ import annotations

import json
import sys
import time

import l2hmc.configs as cfgs


def _summary(res: dict, nb: int) -> dict:
    rate = res['timer'].get_eval_rate()
    h = res['history']
    out = {'steps': rate['num_steps'], 'LF_per_s': rate['eval_rate'],
           'chain_LF_per_s': rate['eval_rate'] * nb,
           'acc_mean': float(sum(a.mean() for a in h['acc']) / len(h['acc'])),
           'loss_last': h['loss'][-1]}
    if 'dQint' in h and len(h['dQint']) > 1:
        out['dQint_mean'] = float(sum(d.mean() for d in h['dQint'][1:]) / (len(h['dQint']) - 1))
    return out


def get_experiment(cfg: dict, keep=None, skip=None):
    """Seed, set the default dtype, build the Experiment -- the order of the reference's entry
    point (__main__.py:55-93): `setup_torch(seed=cfg.seed, ...)`, float64 default for the fp64
    precisions, then `Experiment(cfg)` continuing that generator stream."""
    import torch
    from l2hmc.utils.dist import setup_torch
    framework = str(cfg.get('framework'))
    if framework not in cfgs.SYNONYMS['pytorch']:
        raise ValueError('Framework must be specified, one of: [pytorch] (the TensorFlow back-end '
                         f'is outside this build); got {framework!r}')
    cfg['framework'] = 'pytorch'
    seed = cfg.get('seed')
    setup_torch(seed=0 if seed is None else int(seed), backend=cfg.get('backend', 'DDP'),
                port=str(cfg.get('port', '2345')))
    if str(cfg.get('precision')) in cfgs.FP64_SYNONYMS + ['f64']:
        torch.set_default_dtype(torch.float64)
    from l2hmc.experiment.pytorch.experiment import Experiment
    return Experiment(cfg, keep=keep, skip=skip)


def main(argv=None) -> dict:
    overrides = list(sys.argv[1:] if argv is None else argv)
    config_name, outdir = 'config', None
    rest = []
    it = iter(overrides)
    for a in it:                                   # hydra's --config-name / -cn; outdir=<dir>
        if a in ('--config-name', '-cn'):
            config_name = next(it)
        elif a.startswith('--config-name='):
            config_name = a.partition('=')[2]
        elif a.startswith('outdir='):
            outdir = a.partition('=')[2]
        else:
            rest.append(a)
    cfg = cfgs.get_config(rest, config_name=config_name)
    ex = get_experiment(cfg)
    out: dict = {}
    nb = ex.config.dynamics.nchains
    x = None
    if ex.config.steps.nera > 0 and ex.config.steps.nepoch > 0:
        t0 = time.time()
        res = ex.train()
        x = res['x']
        out['train'] = {'steps': len(res['history']['loss']), 'seconds': time.time() - t0,
                        'loss_first': res['history']['loss'][0],
                        'loss_last': res['history']['loss'][-1],
                        'beta_last': res['history']['beta'][-1]}
    if ex.config.steps.test > 0:
        for job in ('eval', 'hmc'):
            res = ex.evaluate(job_type=job, x=x)
            if res is not None:
                out[job] = _summary(res, nb)
        if 'dQint_mean' in out.get('eval', {}) and out.get('hmc', {}).get('dQint_mean', 0) > 0:
            out['model_improvement'] = out['eval']['dQint_mean'] / out['hmc']['dQint_mean']
    if outdir is not None:
        # the reference's per-job dataset dump (utils/history.py:894-909, experiment save_dataset)
        for job, hist in ex.trainer.histories.items():
            if hist.history:
                out.setdefault('datasets', {})[job] = str(hist.save_dataset(outdir, job))
    print(json.dumps(out))
    return out


if __name__ == '__main__':
    main()
