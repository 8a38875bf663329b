"""``BaseHistory`` -- per-step metric store of a training / evaluation run and its dataset dump.

Counterpart of the reference's ``src/l2hmc/utils/history.py`` (``BaseHistory`` :157-263 --
``update`` / ``_update`` / ``era_summary`` / ``metric_to_numpy`` -- and the dataset part
:854-909 -- ``to_DataArray`` / ``get_dataset``) without its plotting (matplotlib / xarray plot
helpers are out of scope).  The dataset keeps the reference's
axis conventions:

    [ndraws]                -> dims ('draw',)
    [ndraws, nchains]       -> ('chain', 'draw')             (transposed like the reference)
    [ndraws, nlf, nchains]  -> ('chain', 'leapfrog', 'draw')

and is written as ``<job>_data.npz`` (+ ``<job>_dims.json``) and ``<job>_avgs.csv`` (one row per
step: the per-step averages ``update`` returns); ``get_dataset`` returns an ``xarray.Dataset``
when xarray is importable, else the plain ``{name: (dims, array)}`` dict.
"""
from __future__ import annotations

import csv
import json
from pathlib import Path
from typing import Any, Optional

import numpy as np
import torch

from l2hmc.configs import Steps

SCALARS = (float, int, bool, np.floating, np.integer)


def grab(x: Any) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        x = x.detach()
        return (x if x.is_complex() or x.dtype == torch.float64 else x.float()).cpu().numpy()
    return np.asarray(x)


def format_pair(k: str, v) -> str:
    """`key=value` the way the reference prints step summaries (utils/history.py:59-64)."""
    if isinstance(v, (int, bool, np.integer)):
        return f'{k}={v}'
    return f'{k}={v:<.3f}'


def summarize_dict(d: dict) -> str:
    return ' '.join([format_pair(k, v) for k, v in d.items()])


class BaseHistory:
    def __init__(self, steps: Optional[Steps] = None):
        self.steps = steps
        self.history: dict[str, list] = {}
        self.era_metrics: dict[str, dict] = {}
        self.rows: list[dict] = []
        if steps is not None:
            self.era_metrics = {str(era): {} for era in range(steps.nera)}

    def era_summary(self, era) -> str:
        em = self.era_metrics.get(str(era))
        if em is None:
            return ''
        return ', '.join(f'{k}={np.mean(v):<5.4f}' for k, v in em.items()
                         if k not in ('era', 'epoch') and v is not None)

    def metric_to_numpy(self, metric: Any):
        if isinstance(metric, SCALARS) or isinstance(metric, np.ndarray):
            return metric
        if isinstance(metric, torch.Tensor):
            return grab(metric)
        if isinstance(metric, list):
            if isinstance(metric[0], torch.Tensor):
                return grab(torch.stack(metric))
            if isinstance(metric[0], np.ndarray):
                return np.stack(metric)
        return np.array(metric)

    def _update(self, key: str, val: Any):
        if isinstance(val, (list, tuple)) and len(val) and isinstance(val[0], torch.Tensor):
            val = grab(torch.stack(list(val)))
        if isinstance(val, torch.Tensor):
            val = grab(val)
        self.history.setdefault(key, []).append(val)
        if isinstance(val, SCALARS):
            return val
        return np.mean(val).real

    def update(self, metrics: dict) -> dict[str, Any]:
        """Append one step's metrics; returns their per-step averages (history.py:235-263)."""
        avgs: dict[str, Any] = {}
        era = metrics.get('era')
        for key, val in metrics.items():
            if val is None:
                continue
            items = [(f'{key}/{k}', v) for k, v in val.items()] if isinstance(val, dict) else [(key, val)]
            for kk, v in items:
                avg = self._update(kk, v)
                avgs[kk] = avg
                if era is not None:
                    self.era_metrics.setdefault(str(era), {}).setdefault(kk, []).append(avg)
        self.rows.append(avgs)
        return avgs

    # ---- dataset
    @staticmethod
    def to_array(x, therm_frac: Optional[float] = 0.0):
        """(dims, array) with the reference's axis order (history.py:854-892)."""
        arr = np.array(x)
        if not np.iscomplexobj(arr):
            arr = arr.real
        if therm_frac is not None and therm_frac > 0:
            arr = arr[int(therm_frac * arr.shape[0]):]
        if arr.ndim == 1:
            return ('draw',), arr
        if arr.ndim == 2:
            return ('chain', 'draw'), arr.T
        if arr.ndim == 3:
            return ('chain', 'leapfrog', 'draw'), arr.T
        raise ValueError(f'Invalid shape encountered: {arr.shape}')

    def get_dataset(self, data: Optional[dict] = None, therm_frac: Optional[float] = 0.0):
        data = self.history if data is None else data
        out = {}
        for key, val in data.items():
            try:
                out[key.replace('/', '_')] = self.to_array(val, therm_frac)
            except ValueError:
                continue                      # ragged / >3-d entries are skipped like the reference
        try:
            import xarray as xr
        except Exception:
            return out
        return xr.Dataset({k: xr.DataArray(a, dims=d) for k, (d, a) in out.items()})

    def save_dataset(self, outdir, job_type: str = 'train', therm_frac: float = 0.0) -> Path:
        """<outdir>/<job>_data.npz, <job>_dims.json, <job>_avgs.csv."""
        outdir = Path(outdir)
        outdir.mkdir(parents=True, exist_ok=True)
        ds = {}
        for key, val in self.history.items():
            try:
                ds[key.replace('/', '_')] = self.to_array(val, therm_frac)
            except ValueError:
                continue
        f = outdir / f'{job_type}_data.npz'
        np.savez_compressed(f, **{k: a for k, (d, a) in ds.items()})
        (outdir / f'{job_type}_dims.json').write_text(json.dumps({k: list(d) for k, (d, a) in ds.items()}))
        keys = sorted({k for r in self.rows for k in r})
        with open(outdir / f'{job_type}_avgs.csv', 'w', newline='') as fh:
            w = csv.DictWriter(fh, fieldnames=keys)
            w.writeheader()
            for r in self.rows:
                w.writerow({k: (float(np.real(v)) if not isinstance(v, (str, bool)) else v)
                            for k, v in r.items()})
        return f
