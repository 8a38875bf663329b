"""cfg-4 trajectory: eager launches vs the HIP-graph replay (Dynamics.make_graphed), same box, interleaved."""
import os, sys, time
import torch
ROOT = os.path.join(os.path.dirname(__file__), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
import bench
sys.argv = ['bench.py'] + sys.argv[1:]
args = bench.parse()
dyn, lat = bench.build(args, 9992)
x = bench.hot_start(args, seed=1)
beta = torch.tensor(args.beta)
for _ in range(2):
    xo, m = dyn((x, beta))
g = dyn.make_graphed(x, beta=float(beta))


def run(fn, n=5):
    global x
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        xo, m = fn(x)
        x = xo
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    te = run(lambda xx: dyn((xx, beta)))
    tg = run(lambda xx: g(xx))
    tc = run(lambda xx: tuple(t.clone() if isinstance(t, torch.Tensor) else t for t in g(xx)))
    tec = run(lambda xx: tuple(t.clone() if isinstance(t, torch.Tensor) else t for t in dyn((xx, beta))))
    print(f'eager {te:.3f} ms  graph {tg:.3f} ms  graph + clone(x_out) {tc:.3f} ms  eager + clone(x_out) {tec:.3f} ms', flush=True)
