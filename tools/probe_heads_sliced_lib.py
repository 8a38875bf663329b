"""shared by tools/probe_heads_sliced.py and tools/time_heads_sliced.py"""
import torch
dev = 'cuda'


def make(m, n, k=256, cplx=True, wscale=0.05):
    z = torch.relu(torch.randn(m, k, dtype=torch.float64, device=dev))
    heads = {}
    for nm in 'stq':
        w = (torch.rand(n, k, dtype=torch.float64, device=dev) * 2 - 1) * wscale
        b = torch.randn(n, dtype=torch.float64, device=dev) * 0.1
        c = None if nm == 't' else (1.0 + 0.1 * torch.randn(n, dtype=torch.float64, device=dev))
        heads[nm] = (w, b, c)
    dt = torch.complex128 if cplx else torch.float64
    v = torch.randn(m, n, dtype=dt, device=dev)
    f = torch.randn(m, n, dtype=dt, device=dev)
    return z, heads, v, f


