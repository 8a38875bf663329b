"""Can HIP events recorded INSIDE a captured graph time its kernels on replay?  (torch.cuda.Event(external=True))"""
import torch
x = torch.randn(1 << 26, device='cuda')
y = torch.empty_like(x)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        y.copy_(x); y.mul_(2.0)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    evs = [torch.cuda.Event(enable_timing=True, external=True) for _ in range(3)]
except TypeError as e:
    print('no external events:', e); raise SystemExit(0)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    evs[0].record()
    y.copy_(x)
    evs[1].record()
    for _ in range(4):
        y.mul_(1.0001)
    evs[2].record()
for rep in range(3):
    g.replay()
    torch.cuda.synchronize()
    print('replay', rep, 'copy ms', evs[0].elapsed_time(evs[1]), '4 x mul ms', evs[1].elapsed_time(evs[2]))
