"""Counter-based weights for fixtures whose networks are too big to store.

``tests/golden/make_golden_sizes.py`` loads these numbers into the REAL reference's ``Dynamics``
(numpy, CPU) and the GPU tests load the same numbers into the product's ``Dynamics`` (torch, on
the device): element ``i`` of tensor ``name`` is a 32-bit integer hash of ``(crc32(name) ^ seed,
i)`` mapped to [-1, 1) -- integer arithmetic below 2^63 and power-of-two scalings only, so the
numpy and the torch form give identical bits on any device.  Test infrastructure only.
"""
import zlib

import numpy as np

_M = 0x45D9F3B
_MASK = 0xFFFFFFFF


def _mix(h):
    h = (((h >> 16) ^ h) * _M) & _MASK
    h = (((h >> 16) ^ h) * _M) & _MASK
    return (h >> 16) ^ h


def key_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & _MASK


def _mix_u32(h):
    """_mix on uint32 arrays (wrapping multiply == the masked int64 form), in place"""
    m = np.uint32(_M)
    for _ in range(2):
        h ^= h >> np.uint32(16)
        h *= m
    h ^= h >> np.uint32(16)
    return h


def unit_np(n: int, kseed: int) -> np.ndarray:
    """float32 [n] in [-1, 1)"""
    out = np.empty(n, dtype=np.float32)
    step = 1 << 22
    ks = np.uint32(kseed)
    for i0 in range(0, n, step):
        h = np.arange(i0, min(n, i0 + step), dtype=np.uint32)
        h ^= ks
        h = _mix_u32(h)
        h += ks
        h = _mix_u32(h)
        h >>= np.uint32(8)
        o = out[i0:i0 + h.size]
        o[:] = h
        o *= np.float32(2.0 ** -23)
        o -= np.float32(1.0)
    return out


def unit_torch(n: int, kseed: int, device):
    import torch
    out = torch.empty(n, dtype=torch.float32, device=device)
    step = 1 << 26
    for i0 in range(0, n, step):
        i = torch.arange(i0, min(n, i0 + step), dtype=torch.int64, device=device)
        h = _mix((i ^ kseed) & _MASK)
        h = _mix((h + kseed) & _MASK)
        out[i0:i0 + i.numel()] = (h >> 8).to(torch.float32) * (2.0 ** -23) - 1.0
    return out


def plan(name: str, shape, head_scale: float):
    """(kind, a, b): value = a + b * unit  (b rounded to float32 once, on both sides)."""
    leaf = name.rsplit('.', 1)[-1]
    nd = len(shape)
    if leaf == 'coeff':
        return 0.0, 0.3
    if leaf == 'running_mean':
        return 0.0, 0.1
    if leaf == 'running_var':
        return 1.2, 0.2
    if leaf == 'weight' and nd == 1:            # BatchNorm scale
        return 1.0, 0.2
    if leaf == 'bias':
        return 0.0, 0.05 * (head_scale if _is_head(name) else 1.0)
    if leaf == 'weight':
        fan_in = int(np.prod(shape[1:]))
        return 0.0, float(np.float32(1.0 / np.sqrt(fan_in))) * (head_scale if _is_head(name) else 1.0)
    return None


def _is_head(name: str) -> bool:
    return any(s in name for s in ('.scale.layer.', '.transl.', '.transf.layer.'))


def fill_state_dict(sd: dict, seed: int, head_scale: float, torch_device=None):
    """In-place fill of every float weight / buffer of a ``Dynamics.state_dict()`` (keys under the
    aliased ``networks.`` prefix, step sizes and integer buffers are left alone).  ``sd`` values are
    torch tensors (reference: CPU; product: device); returns the number of elements written."""
    import torch
    n_el = 0
    for k in sorted(sd):
        t = sd[k]
        if k.startswith('networks.') or not t.dtype.is_floating_point:
            continue
        if k.split('.')[0] in ('xeps', 'veps'):
            continue
        p = plan(k, tuple(t.shape), head_scale)
        if p is None:
            continue
        a, b = p
        ks = key_seed(k, seed)
        n = t.numel()
        with torch.no_grad():
            if torch_device is None:
                u = torch.from_numpy(unit_np(n, ks))
            else:
                u = unit_torch(n, ks, torch_device)
            val = u * float(np.float32(b)) + float(np.float32(a))
            t.copy_(val.reshape(t.shape).to(t.dtype))
        n_el += n
    return n_el
