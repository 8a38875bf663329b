import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/l2hmc-qcd_amd')
import torch
sys.argv = ['bench.py', '--no-cpu-baseline']
import bench
args = bench.parse()
dyn, lat = bench.build(args, seed=9992)
x = bench.hot_start(lat, args, seed=9992)
beta = torch.tensor(6.0)
for _ in range(3):
    xo, m = dyn((x, beta))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    xo, m = dyn((x, beta))
torch.cuda.synchronize(); print('eager  ms', (time.perf_counter() - t0) / 5 * 1e3)
g = dyn.make_graphed(x, 6.0, mode='fb')
for _ in range(2):
    xo, m = g(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    xo, m = g(x)
torch.cuda.synchronize(); print('graph  ms', (time.perf_counter() - t0) / 5 * 1e3, float(m['acc'].mean()))
