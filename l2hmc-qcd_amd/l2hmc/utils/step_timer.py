"""``StepTimer`` -- same bookkeeping and ``get_eval_rate`` quantity as the reference's
``src/l2hmc/utils/step_timer.py:14-137`` (the rate this repo's bench.py reports x nchains),
without the pandas dependency."""
from __future__ import annotations

import json
import os
import time
from pathlib import Path
from typing import Optional

import numpy as np


class StepTimer:
    def __init__(self, evals_per_step: int = 1) -> None:
        self.data: list[float] = []
        self.t = time.time()
        self.iterations = 0
        self.evals_per_step = evals_per_step

    def start(self) -> None:
        self.t = time.time()

    def stop(self) -> float:
        dt = time.time() - self.t
        self.data.append(dt)
        self.iterations += 1
        return dt

    def get_eval_rate(self, evals_per_step: Optional[int] = None) -> dict:
        """eval_rate = evals_per_step * n / elapsed (step_timer.py:87-100)"""
        if evals_per_step is None:
            evals_per_step = self.evals_per_step
        elapsed = float(np.sum(self.data))
        num_evals = evals_per_step * len(self.data)
        return {'eval_rate': num_evals / elapsed if elapsed > 0 else 0.0,
                'total_time': elapsed, 'num_evals': num_evals, 'num_steps': len(self.data),
                'evals_per_step': evals_per_step}

    def write_eval_rate(self, outdir: os.PathLike, mode: str = 'a',
                        evals_per_step: Optional[int] = None) -> dict:
        rate = self.get_eval_rate(evals_per_step)
        outfile = Path(outdir).joinpath('step_timer_output.json')
        Path(outdir).mkdir(parents=True, exist_ok=True)
        with open(outfile, mode) as f:
            json.dump(rate, f)
        return rate

    def save_data(self, outfile: os.PathLike, mode: str = 'a') -> None:
        Path(outfile).parent.mkdir(parents=True, exist_ok=True)
        with open(outfile, mode) as f:
            for i, dt in enumerate(self.data):
                f.write(f'{i},{dt}\n')

    def save_and_write(self, outdir: os.PathLike, mode: str = 'a', fname: Optional[str] = None,
                       evals_per_step: Optional[int] = None) -> dict:
        fname = 'step_timer' if fname is None else fname
        self.save_data(Path(outdir).joinpath(f'{fname}.csv'), mode=mode)
        return self.write_eval_rate(outdir, mode=mode, evals_per_step=evals_per_step)
