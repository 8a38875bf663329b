#!/usr/bin/env python3
"""bench.py -- chain*leapfrog-steps/sec of the L2HMC leapfrog integrator on MI355X.

One "step" = one ``Dynamics.forward((x, beta))``: fresh momenta, a merged forward+backward
L2HMC trajectory (2 * nleapfrog generalised leapfrog steps, each = 2 force evaluations,
2 vnet calls, 2 masked expm link updates), Metropolis accept/reject and select -- exactly what
the reference's ``Trainer.eval_step`` times (``StepTimer.get_eval_rate``,
src/l2hmc/utils/step_timer.py:87-100).  Workload = BASELINE.json configs[3], the one the
metric is quoted on: 4D SU(3) 8^4, beta = 6.0, 256 chains per GPU, complex128 / fp64,
nleapfrog = 4 (=> 8 executed LF steps), vnet units [256] (conf/su3test.yaml of the reference).
Synthetic hot-start gauge field, random-init networks.  Chains are independent: with N GPUs
every rank runs its own 256 chains (weak scaling), no data-path collective.

Prints ONE JSON line (rank 0).  ``roofline`` is measured live with HIP events recorded on the
launch stream around every kernel launch of the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_MFMA_PEAK_TF = 78.6       # MI355X fp64 matrix peak (datasheet; v_mfma_f64_16x16x4_f64)

# algorithmic bytes per (chain * site) -- SURVEY.md section 8(d)
ALG_BYTES = {
    'l2q_su3_plaq_reduce': 576, 'l2q_su3_force': 1152, 'l2q_su3_force_kick': 1728,
    'l2q_su3_expm_mul': 1728, 'l2q_su3_projsu_vec8': 832, 'l2q_v_update': 2592,
    'l2q_su3_kinetic_reduce': 576, 'l2q_su3_pack': 1152, 'l2q_su3_unpack': 1152,
    'l2q_select_rows': 1728, 'l2q_su3_assemble_tah': 832, 'l2q_scale_f64': 1152,
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--nchains', type=int, default=256, help='chains per GPU')
    ap.add_argument('--lattice', type=int, nargs=4, default=[8, 8, 8, 8])
    ap.add_argument('--nleapfrog', type=int, default=4)
    ap.add_argument('--units', type=int, nargs='+', default=[256])
    ap.add_argument('--beta', type=float, default=6.0)
    ap.add_argument('--mode', choices=['l2hmc', 'hmc'], default='l2hmc')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-chains', type=int, default=32)
    return ap.parse_args()


def build(args, seed):
    import l2hmc.configs as cfgs
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3
    from l2hmc.network.pytorch.network import NetworkFactory
    torch.set_default_dtype(torch.float64)
    L = list(args.lattice)
    V = int(np.prod(L))
    # weights / masks from the base seed on every rank; chains differ per rank
    torch.manual_seed(seed)
    np.random.seed(seed)
    dc = cfgs.DynamicsConfig(nchains=args.nchains, group='SU3', latvolume=L,
                             nleapfrog=args.nleapfrog, eps=0.01, eps_hmc=0.01, verbose=False,
                             use_split_xnets=False, use_separate_networks=False,
                             merge_directions=True)
    nc = cfgs.NetworkConfig(units=list(args.units), activation_fn='tanh', dropout_prob=0.0,
                            use_batch_norm=False)
    spec = cfgs.InputSpec(xshape=tuple(dc.xshape), xnet={'x': [32 * V], 'v': [32 * V]},
                          vnet={'x': [32 * V], 'v': [32 * V]})
    lat = LatticeSU3(args.nchains, L)
    nf = NetworkFactory(spec, nc, cfgs.ConvolutionConfig(),
                        cfgs.NetWeights(x=cfgs.NetWeight(0., 1., 1.), v=cfgs.NetWeight(1., 1., 1.)))
    dyn = Dynamics(lat.action, dc, nf if args.mode == 'l2hmc' else None)
    dyn.eval()
    return dyn, lat


def hot_start(lat, args, seed):
    """x = projectSU(randn + i randn) (SU3.random, group/su3/pytorch/group.py:113-119), drawn on
    the device generator and projected by the HIP kernel."""
    from l2hmc import _ops as ops
    g = torch.Generator(device='cuda')
    g.manual_seed(seed)
    V = int(np.prod(args.lattice))
    z = torch.randn((args.nchains, 4, 9, V, 2), dtype=torch.float64, device='cuda', generator=g)
    xn = ops.su3_project_su_n(torch.view_as_complex(z))
    return ops.su3_unpack(xn, args.lattice)


class KernelTimer:
    """HIP events on the launch stream around every C-ABI call while enabled."""

    def __init__(self):
        self.records = []
        self.enabled = False

    def install(self):
        from l2hmc import native
        orig = native.call
        timer = self

        def timed_call(name, *a):
            if not timer.enabled:
                return orig(name, *a)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(name, *a)
            e1.record()
            fl = 2.0 * a[2] * a[3] * (a[4] + a[7]) if name.startswith('l2q_gemm') else 0.0
            timer.records.append((name, fl, e0, e1))
        native.call = timed_call
        import l2hmc._ops as ops
        ops.N.call = timed_call

    def summary(self):
        out = {}
        for name, fl, e0, e1 in self.records:
            d = out.setdefault(name, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1) * 1e-3
            d[2] += fl
        return out


def cpu_baseline(dyn, args):
    """The numpy oracle (a port of the reference's PyTorch-CPU path, pinned to golden vectors)
    timed on the host cores on a bounded sample of the same workload."""
    from oracle import network as onet, su3 as osu3
    from oracle.dynamics import DynamicsOracle
    nbc = args.cpu_chains
    L = tuple(args.lattice)
    rng = np.random.default_rng(1)
    nlf = 1
    masks = [m.numpy().reshape(-1) for m in dyn.masks[:nlf]]
    vnet = None
    if args.mode == 'l2hmc':
        w = {k: v.detach().cpu().numpy() for k, v in dyn.vnet.state_dict().items()}

        def vnet(step, xv, fv):
            return onet.leapfrog_layer(xv, fv, w, nunits=len(args.units), activation='tanh')
    orc = DynamicsOracle('SU3', L, nlf, [0.01] * nlf, [0.01] * nlf, masks, vnet=vnet)
    x = osu3.project_su(rng.normal(size=(nbc, 4, *L, 3, 3)) + 1j * rng.normal(size=(nbc, 4, *L, 3, 3)))
    nrm = rng.normal(size=(8, nbc, 4, *L))
    u = rng.random(nbc)
    t0 = time.perf_counter()
    if args.mode == 'l2hmc':
        orc.apply_transition_fb(x, args.beta, nrm, u)
        nsteps = 2 * nlf
    else:
        orc.apply_transition_hmc(x, args.beta, nrm, u, 0.01, 2)
        nsteps = 2
    dt = time.perf_counter() - t0
    return {'value': nbc * nsteps / dt, 'unit': 'chain*leapfrog-steps/s', 'cores': 1,
            'kind': 'port',
            'sample': f'{nbc} chains x {nsteps} LF steps of the same {args.mode} trajectory '
                      f'(SU(3) {"x".join(map(str, L))}, units {args.units}) in {dt:.1f} s; '
                      'single-threaded numpy'}


def secondary(dyn, x, beta, args, nlf_exec):
    """Untimed-region extras SURVEY.md 8(d) asks to report beside the headline: the same
    trajectory with per-step metrics (verbose=True, the YAML default) and the plain-HMC
    baseline sampler (apply_transition_hmc), each over 2 steps after 1 warm-up."""
    res = {}

    def rate(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        return round(args.nchains * nlf_exec * 2 / (time.perf_counter() - t0), 1)
    old = dyn.config.verbose
    try:
        dyn.config.verbose = True
        res['l2hmc_verbose_true'] = rate(lambda: dyn((x, beta)))
        dyn.config.verbose = False
        res['hmc'] = rate(lambda: dyn.apply_transition_hmc((x, beta), eps=0.01,
                                                           nleapfrog=nlf_exec))
    finally:
        dyn.config.verbose = old
    res['unit'] = 'chain*leapfrog-steps/s'
    return res


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU path)'
    # L2Q_BENCH_SHARE_GPU=1 + L2Q_BENCH_BACKEND=gloo: functional test of the N>1 path on a
    # 1-GPU box (all ranks on device 0); the real runs use one GPU per rank and RCCL.
    share = os.environ.get('L2Q_BENCH_SHARE_GPU') == '1'
    torch.cuda.set_device(0 if share else local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(os.environ.get('L2Q_BENCH_BACKEND', 'nccl'), rank=rank,
                                world_size=world)                      # nccl = RCCL over xGMI
    dyn, lat = build(args, seed=9992)
    x = hot_start(lat, args, seed=9992 * (rank + 1))
    beta = torch.tensor(args.beta)
    nlf_exec = 2 * args.nleapfrog

    def step(xin):
        if args.mode == 'l2hmc':
            xo, m = dyn((xin, beta))
        else:
            xo, m = dyn.apply_transition_hmc((xin, beta), eps=0.01, nleapfrog=nlf_exec)
        return xo, m

    # set-up, not a measured or counted step: the first two trajectories of a process grow
    # PyTorch's caching-allocator pool (hipMalloc is synchronous: 28 + 5 calls, 2.6 s + 1.4 s at
    # the 16^4 shard) and build the native-order weight copies.  Doing that
    # here keeps the W warm-up steps of the contract what they are meant to be even for W = 0 / 1.
    for _ in range(2):
        step(x)
    torch.cuda.synchronize()
    timer = KernelTimer()
    timer.install()
    for _ in range(args.warmup):
        x, m = step(x)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x, m = step(x)
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    acc = m['acc']
    assert torch.isfinite(acc).all() and torch.isfinite(x).all(), 'non-finite trajectory'

    if rank == 0:
        V = int(np.prod(args.lattice))
        sites = args.nchains * V
        ks = timer.summary()
        total_k = sum(v[1] for v in ks.values())
        kernels = {}
        for name, (cnt, tt, fl) in sorted(ks.items(), key=lambda kv: -kv[1][1]):
            ent = {'launches': cnt, 'avg_ms': round(tt / cnt * 1e3, 4),
                   'share': round(tt / total_k, 4)}
            if name in ALG_BYTES:
                ent['GB/s'] = round(sites * ALG_BYTES[name] / (tt / cnt) / 1e9, 1)
            kernels[name] = ent
        traffic_file = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        pmc = json.load(open(traffic_file)) if os.path.exists(traffic_file) else {}

        def traffic(name):
            t = pmc.get(name)
            return (None, None) if t is None else (t['total_bytes'], t.get('source'))

        def hbm_roof(name):
            cnt, tt, _ = ks[name]
            ach = sites * ALG_BYTES[name] / (tt / cnt) / 1e9
            tr, src = traffic(name)
            return {'kernel': name, 'bound': 'hbm', 'achieved': round(ach, 1),
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                    'traffic': tr, 'traffic_source': src, 'avg_ms': round(tt / cnt * 1e3, 4),
                    'launches': cnt, 'algorithmic_bytes_per_launch': sites * ALG_BYTES[name],
                    'time_s': tt}

        def mfma_roof(names, label, flops):
            cnt = sum(ks[n][0] for n in names)
            tt = sum(ks[n][1] for n in names)
            ach = flops / tt / 1e12
            tr, src = traffic(names[0])
            return {'kernel': label, 'bound': 'mfma', 'achieved': round(ach, 2),
                    'peak': FP64_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
                    'frac': round(ach / FP64_MFMA_PEAK_TF, 4), 'traffic': tr,
                    'traffic_source': src, 'avg_ms': round(tt / cnt * 1e3, 4), 'launches': cnt,
                    'time_s': tt}

        rooflines = []
        force_name = 'l2q_su3_force' if 'l2q_su3_force' in ks else 'l2q_su3_force_kick'
        rooflines.append(hbm_roof(force_name))
        rooflines.append(hbm_roof('l2q_su3_plaq_reduce'))
        heads = [n for n in ('l2q_vnet_heads_vupdate_pair_f64', 'l2q_vnet_heads_vupdate_f64')
                 if n in ks]
        if heads:
            per = 3 * 2.0 * args.nchains * args.units[-1] * 36 * V       # s, t, q heads
            rooflines.append(mfma_roof(heads, 'l2q_vnet_heads_vupdate[_pair]_f64 (3 heads + '
                                       'v-update)', per * sum(ks[n][0] for n in heads)))
        if 'l2q_gemm_f64' in ks:
            rooflines.append(mfma_roof(['l2q_gemm_f64'], 'l2q_gemm_f64 (input + hidden layers)',
                                       ks['l2q_gemm_f64'][2]))
        # `roofline` = the dominant kernel (largest share of the timed region) among those
        roofline = dict(max(rooflines, key=lambda r: r['time_s']))
        for r in rooflines + [roofline]:
            r.pop('time_s', None)
        nchain_lf = world * args.nchains * nlf_exec * args.steps
        out = {
            'metric': 'chain*leapfrog-steps/sec, 4D SU(3) 8^4 fp64 '
                      '(+ plaquette/force-kernel HBM GB/s in "rooflines")',
            'value': round(nchain_lf / dt, 1),
            'unit': 'chain*leapfrog-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'4D SU(3) {"x".join(map(str, args.lattice))}, beta={args.beta}, '
                                   f'{args.nchains} chains/GPU, complex128/fp64, '
                                   f'{"Dynamics.forward merged L2HMC" if args.mode == "l2hmc" else "apply_transition_hmc"}, '
                                   f'nleapfrog={args.nleapfrog} ({nlf_exec} LF steps/trajectory), '
                                   f'vnet units {args.units}, verbose=False',
                       'global_chains': world * args.nchains, 'parallelism': f'chains sharded x{world}'},
            'roofline': roofline,
            'rooflines': rooflines,
            'kernel_time_fraction_of_wall': round(total_k / dt, 4),
            'kernels': kernels,
            'accept_prob_mean': round(float(acc.mean()), 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            out['secondary'] = secondary(dyn, x, beta, args, nlf_exec)
            out['cpu_baseline'] = cpu_baseline(dyn, args)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
