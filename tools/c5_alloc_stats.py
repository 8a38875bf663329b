import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/l2hmc-qcd_amd')
import torch, numpy as np
sys.argv = ['bench.py', '--lattice', '16', '16', '16', '16', '--no-cpu-baseline']
import bench
args = bench.parse()
dyn, lat = bench.build(args, seed=9992)
x = bench.hot_start(lat, args, seed=9992)
beta = torch.tensor(6.2)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s0 = torch.cuda.memory_stats()
    xo, m = dyn((x, beta))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s1 = torch.cuda.memory_stats()
    print(f'step {i}: {dt*1e3:.0f} ms  device_alloc +{s1["num_device_alloc"]-s0["num_device_alloc"]} device_free +{s1["num_device_free"]-s0["num_device_free"]} retries {s1["num_alloc_retries"]} reserved {s1["reserved_bytes.all.current"]/2**30:.1f} GiB peak_alloc {s1["allocated_bytes.all.peak"]/2**30:.1f} GiB', flush=True)
    x = xo.reshape(x.shape)
