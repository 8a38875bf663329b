"""cfg-4 trajectory replayed from a HIP graph: does it matter where the input comes from?  (round 5: a replay fed
with a CLONE of its own output measured 1.2 ms faster than one fed with the output buffer itself.)
MODE=own|clone|const|eager|eager_const  STEPS=n"""
import os, sys, time
import torch
ROOT = os.path.join(os.path.dirname(__file__), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
import bench
mode = os.environ.get('MODE', 'own')
steps = int(os.environ.get('STEPS', 10))
sys.argv = ['bench.py']
args = bench.parse()
dyn, lat = bench.build(args, 9992)
x0 = bench.hot_start(args, seed=1)
beta = torch.tensor(args.beta)
for _ in range(2):
    xo, m = dyn((x0, beta))
g = dyn.make_graphed(x0, beta=float(beta)) if not mode.startswith('eager') else None


def one(x):
    if mode == 'own':
        return g(x)[0]
    if mode == 'clone':
        return g(x)[0].clone()
    if mode == 'const':
        g(x0)
        return x0
    if mode == 'eager':
        return dyn((x, beta))[0]
    dyn((x0, beta))
    return x0


for rep in range(3):
    x = x0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        x = one(x)
    torch.cuda.synchronize()
    print(f'{mode}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per trajectory', flush=True)
