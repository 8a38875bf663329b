"""``python -m l2hmc key=value ...`` -- thin counterpart of the reference's hydra entry point
(src/l2hmc/__main__.py:55-274): compose the config tree with dotted overrides, build the
Experiment, run the evaluation and the HMC baseline, print the eval rates as JSON.

    python -m l2hmc +experiment=su3 dynamics.nchains=16 steps.test=10
"""
from __future__ import annotations

import json
import sys

import l2hmc.configs as cfgs


def main(argv=None) -> dict:
    overrides = list(sys.argv[1:] if argv is None else argv)
    cfg = cfgs.get_config(overrides)
    from l2hmc.experiment.pytorch.experiment import Experiment
    ex = Experiment(cfg)
    out = {}
    nb = ex.config.dynamics.nchains
    for job in ('eval', 'hmc'):
        res = ex.evaluate(job_type=job)
        if res is None:
            continue
        rate = res['timer'].get_eval_rate()
        h = res['history']
        out[job] = {'steps': rate['num_steps'], 'LF_per_s': rate['eval_rate'],
                    'chain_LF_per_s': rate['eval_rate'] * nb,
                    'acc_mean': float(sum(a.mean() for a in h['acc']) / len(h['acc'])),
                    'loss_last': h['loss'][-1]}
    print(json.dumps(out))
    return out


if __name__ == '__main__':
    main()
