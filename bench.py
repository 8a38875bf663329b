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

``--gpus N`` from a bare shell re-executes itself under ``torch.distributed.run`` with N ranks
(one per GPU, backend nccl = RCCL over xGMI); under torchrun (RANK / WORLD_SIZE set) it checks
that the world size equals N.  ``--mode train`` times ``Trainer.train_step`` instead (tape +
reverse sweep + ONE all-reduce of the flat gradient + fused Adam) and reports the collective's
time and bandwidth.

Prints ONE JSON line (rank 0).  ``value`` / ``ms_per_step`` time exactly K steps with nothing else in
the loop; ``roofline`` / ``kernels`` are measured live with HIP events recorded on the launch stream around
every kernel launch of the same K steps run once more right after the timed region
(``instrumented_ms_per_step``: what the events cost is then visible instead of sitting in the headline).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_MFMA_PEAK_TF = 78.6       # MI355X fp64 matrix peak (datasheet; v_mfma_f64_16x16x4_f64)
# dense int8 matrix peak: v_mfma_i32_16x16x64_i8 = 32768 ops per 16 cycles per SIMD, 1024 SIMDs, 2.4 GHz
# (MI355X_MICROARCH.md: "I8 ~2x the bf16 rate", micro-benchmark floor 3944; tools/microbench/mfma_i8_rate.hip
# reaches 4.2-4.9 POP/s on this part)
INT8_MFMA_PEAK_TOPS = 5033.0

# algorithmic bytes per (chain * site) -- SURVEY.md section 8(d): one 3x3 complex128 link is
# 144 B, a site has 4 links; vec8 = 4 * 8 * 8 B = 256 B per site
ALG_BYTES = {
    'l2q_su3_plaq_reduce': 576,            # read x
    'l2q_su3_force': 1152,                 # read x, write F
    'l2q_su3_force_kick': 1728,            # read x, read + write v
    'l2q_su3_force_vec8': 1408,            # read x, write F and vec8(F)
    'l2q_su3_expm_mul': 1728,              # read x, v; write x
    'l2q_su3_expm_mul2': 1728,
    'l2q_su3_expm_mul2_vec8': 1984,        # + write vec8(x')
    'l2q_su3_projsu_vec8': 832,            # read field, write vec8
    'l2q_v_update': 2592,
    'l2q_su3_kinetic_reduce': 576,
    'l2q_su3_pack': 1152, 'l2q_su3_unpack': 1152,
    'l2q_select_rows': 1152,               # the selected source row is read, the output written
    'l2q_su3_assemble_tah': 832,           # 8 normals per link in, one TAH matrix out
    'l2q_scale_f64': 1152,
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
    ap.add_argument('--mode', choices=['l2hmc', 'hmc', 'train'], default='l2hmc')
    ap.add_argument('--micro-batch', type=int, default=None,
                    help='train mode: chains per tape micro-batch (needed at 16^4)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-spot-check', action='store_true')
    ap.add_argument('--no-comm-probe', action='store_true')
    ap.add_argument('--no-u1', action='store_true', help='skip the untimed U(1) cfg-2 / cfg-3 block')
    ap.add_argument('--cpu-chains', type=int, default=32)
    ap.add_argument('--settle', type=int, default=4,
                    help='graph-replayed sampler: set-up trajectories before the warm-up (clock settling)')
    ap.add_argument('--fp64-input-layer', action='store_true',
                    help='A/B: the vnet input layer on the fp64 MFMA kernel instead of the int8-sliced one')
    ap.add_argument('--fp64-train-heads', action='store_true',
                    help='A/B (--mode train): the heads of the training tape as three fp64 GEMMs + v_update '
                         'instead of the TAPE instances of the int8-sliced heads kernel')
    ap.add_argument('--sliced-train-input', action='store_true',
                    help='A/B (--mode train): the input layer of the tape on digit images of the step\'s weights '
                         '(off by default: the per-step image builds cost what the nine calls save)')
    ap.add_argument('--force-native-training', action='store_true',
                    help='--mode train: native-order weight shadows even where the traffic estimate prefers the '
                         'reference-order path (small micro-batches on large lattices)')
    ap.add_argument('--separate-v-pairs', action='store_true',
                    help='A/B (--mode train): the two v-updates that share a network call reversed by two kernels')
    ap.add_argument('--separate-x-halves', action='store_true',
                    help='A/B (--mode train): the two masked x half-updates of a leapfrog step as two tape entries')
    ap.add_argument('--no-defer-weight-grads', action='store_true',
                    help='A/B (--mode train): weight-gradient GEMMs per network call instead of one per matrix and step')
    return ap.parse_args()


def free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def respawn_under_torchrun(args) -> int:
    """`python bench.py --gpus N` from a bare shell: one rank per GPU through torch.distributed.run
    (the reference's counterpart: process-group init + DDP wrap, utils/dist.py:126-144,
    trainers/pytorch/trainer.py:246-257)."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.run(cmd, env=env).returncode


import numpy as np  # noqa: E402
import torch  # noqa: E402


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
    if args.fp64_input_layer:
        dyn.sliced_input = False
    dyn.eval()
    return dyn, lat


def build_trainer(args, seed):
    """Trainer (train_step) at the same shapes; model from the base seed on every rank."""
    import l2hmc.configs as cfgs
    from l2hmc.trainers.pytorch.trainer import Trainer
    torch.manual_seed(seed)
    np.random.seed(seed)
    L = ','.join(str(i) for i in args.lattice)
    units = ','.join(str(u) for u in args.units)
    cfg = cfgs.get_config([
        'dynamics.group=SU3', f'dynamics.latvolume=[{L}]', f'dynamics.nchains={args.nchains}',
        f'dynamics.nleapfrog={args.nleapfrog}', 'dynamics.eps=0.01', 'dynamics.verbose=false',
        'dynamics.use_split_xnets=false', 'dynamics.use_separate_networks=false',
        f'network.units=[{units}]', 'network.activation_fn=tanh', 'network.dropout_prob=0.0',
        'network.use_batch_norm=false', 'conv=none', 'loss.plaq_weight=0.1',
        'loss.rmse_weight=0.1', 'loss.charge_weight=0.0'])
    tr = Trainer(cfg)
    tr.micro_batch = args.micro_batch
    if args.fp64_train_heads:
        tr.dynamics.sliced_train_heads = False
    if args.sliced_train_input:
        tr.dynamics.sliced_train_input = True
    if args.no_defer_weight_grads:
        tr.dynamics.defer_weight_grads = False
    if args.separate_x_halves:
        tr.dynamics.fuse_x_halves_train = False
    if args.separate_v_pairs:
        tr.dynamics.fuse_v_pairs_bwd = False
    if args.force_native_training:
        tr.dynamics.native_training = 'force'

    return tr


def hot_start(args, seed):
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
            fl = 2.0 * a[2] * a[3] * (a[4] + a[7]) if name in ('l2q_gemm_f64', 'l2q_gemm_f32') else 0.0
            if name == 'l2q_gemm_sliced_f64':      # (A, image, K, e, A2, image2, K2, e2, M, N, ...)
                fl = 2.0 * a[8] * a[9] * (a[2] + a[6])
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


def _cpu_sample(spec: dict, seed: int):
    """One bounded sample of the CPU baseline (this process, spec['threads'] torch threads): returns
    (chains x LF steps, seconds).  Module-level so that worker processes can run it."""
    import torch as th
    from oracle import torch_cpu as tc
    th.set_num_threads(spec['threads'])
    L, nlf, nbc = tuple(spec['lattice']), 1, spec['chains']
    gen = th.Generator().manual_seed(seed)
    w = th.load(spec['weights'], mmap=True) if spec['weights'] else None
    sim = tc.TorchSU3Dynamics(L, nlf, [0.01] * nlf, [0.01] * nlf, spec['masks'], w, nunits=spec['nunits'])
    z = th.randn((nbc, 4, *L, 3, 3, 2), dtype=th.float64, generator=gen)
    x = tc.project_su(th.view_as_complex(z))
    nrm = th.randn((8, nbc, 4, *L), dtype=th.float64, generator=gen)
    u = th.rand(nbc, dtype=th.float64, generator=gen)
    t0 = time.perf_counter()
    if spec['mode'] == 'l2hmc':
        sim.apply_transition_fb(x, spec['beta'], nrm, u)
        nsteps = 2 * nlf
    else:
        sim.apply_transition_hmc(x, spec['beta'], nrm, u, 0.01, 2)
        nsteps = 2
    return nbc * nsteps, time.perf_counter() - t0


def _cpu_worker(spec, seed, start, q):
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        while time.time() < start:                      # common start: the workers overlap
            time.sleep(0.005)
        t0 = time.time()
        units, dt = _cpu_sample(spec, seed)
        q.put((units, dt, t0, time.time()))
    except Exception as e:  # noqa: BLE001
        q.put(('error', f'{type(e).__name__}: {e}'))


def cpu_baseline(dyn, args):
    """oracle/torch_cpu.py -- a torch-CPU restatement of the reference's path with the
    reference's method (roll-based Wilson loops, autograd force, torch.matrix_exp; pinned to the
    reference fixtures by tests/test_oracle_golden.py) -- timed on the host cores of the GPU box on a
    bounded sample of the same workload.  torch's intra-op pool on these small 3x3 batches peaks at ~16
    threads and collapses beyond (tools/cpu_threads_probe.py on the 256-CPU host: 35 / 45 / 54 / 33 / 19 / 8
    chain*LF/s at 4 / 8 / 16 / 32 / 64 / 128 threads), so ALL cores are used the way the chains allow:
    cpu_count / 16 worker processes x 16 threads, each on its own chains (they are independent), started
    together; `value` = all chains x LF steps / the time from the common start to the last worker's end.
    The one-process figure is kept as `single_process`."""
    import multiprocessing as mp
    import tempfile
    ncpu = os.cpu_count() or 1
    threads = int(os.environ.get('L2Q_BENCH_CPU_THREADS', min(16, ncpu)))
    L = tuple(args.lattice)
    spec = {'threads': threads, 'lattice': list(L), 'chains': args.cpu_chains, 'mode': args.mode,
            'beta': float(args.beta), 'nunits': len(args.units),
            'masks': [m.numpy().reshape(-1) for m in dyn.masks[:1]], 'weights': None}
    tmp = None
    if args.mode == 'l2hmc':
        tmp = tempfile.NamedTemporaryFile(suffix='.pt', delete=False)
        tmp.close()
        torch.save({k: v.detach().cpu() for k, v in dyn.vnet.state_dict().items()}, tmp.name)
        spec['weights'] = tmp.name
    out = {}
    try:
        units, dt = _cpu_sample(spec, 1)
        single = {'value': round(units / dt, 2), 'cores': threads, 'seconds': round(dt, 2)}
        nproc = max(1, ncpu // threads)
        multi = None
        # measured on the 256-CPU GPU host (profiles/r05f_bench_l2hmc.json): 16 processes x 16 threads give 14.6
        # chain*LF/s in 70 s against 26.3 for ONE process x 16 threads -- every process streams the 1.4 GB of
        # network weights per layer call and they saturate the host memory system -- so the all-cores attempt
        # only runs on request (L2Q_BENCH_CPU_ALL=1) and the default line stays within its time budget
        if nproc > 1 and os.environ.get('L2Q_BENCH_CPU_ALL') == '1':
            ctx = mp.get_context('spawn')
            q = ctx.Queue()
            start = time.time() + 20.0                  # interpreter + torch import of the workers
            ps = [ctx.Process(target=_cpu_worker, args=(spec, 100 + i, start, q)) for i in range(nproc)]
            for p_ in ps:
                p_.start()
            res = []
            deadline = time.time() + 20.0 + max(90.0, 12.0 * dt)
            while len(res) < nproc and time.time() < deadline:
                try:
                    res.append(q.get(timeout=1.0))
                except Exception:  # noqa: BLE001  (queue.Empty)
                    pass
            for p_ in ps:
                p_.join(timeout=1.0)
                if p_.is_alive():
                    p_.kill()
            good = [r for r in res if r[0] != 'error']
            if len(good) == nproc:
                wall = max(r[3] for r in good) - min(r[2] for r in good)
                multi = {'value': round(sum(r[0] for r in good) / wall, 2), 'processes': nproc,
                         'cores': nproc * threads, 'seconds': round(wall, 2)}
            else:
                multi = {'error': f'{len(good)} of {nproc} workers finished', 'detail': str(res)[:200]}
        best = multi if (multi and 'value' in multi and multi['value'] > single['value']) else single
        nsteps = 2
        out = {'value': best['value'], 'unit': 'chain*leapfrog-steps/s', 'cores': best['cores'],
               'host_cpus': ncpu, 'kind': 'port', 'single_process': single, 'all_cores': multi,
               'sample': f'{args.cpu_chains} chains x {nsteps} LF steps of the same {args.mode} trajectory per '
                         f'process (SU(3) {"x".join(map(str, L))}, units {args.units}); oracle/torch_cpu.py, '
                         f'{best.get("processes", 1)} process(es) x {threads} torch threads in '
                         f'{best["seconds"]} s.  Not os.cpu_count() threads: one process scales to ~16 threads '
                         f'(35 / 45 / 54 / 33 / 19 / 8 chain*LF/s at 4 / 8 / 16 / 32 / 64 / 128 threads), and '
                         f'cpu_count/16 processes x 16 threads on independent chains were measured SLOWER on the '
                         f'256-CPU host (14.6 vs 26.3: memory-bound on the network weights; L2Q_BENCH_CPU_ALL=1 '
                         f're-runs that, profiles/r05f_bench_l2hmc.json holds the record)'}
    finally:
        if tmp is not None:
            try:
                os.unlink(tmp.name)
            except OSError:
                pass
    return out


def spot_check(dyn, lat, x, args):
    """Untimed sanity of what was timed (rank 0): observables of the output configuration and one
    force evaluation of two of its chains through the HIP kernels vs the numpy oracle; acceptance
    of plain HMC after a short thermalisation (the random-init L2HMC networks reject everything
    from a hot start, so `accept_prob_mean` alone says nothing about the integrator)."""
    from oracle import su3 as osu3
    from l2hmc import _ops as ops
    L = tuple(args.lattice)
    x4 = x.reshape(args.nchains, 4, *L, 3, 3)
    pick = [0, args.nchains - 1] if args.nchains > 1 else [0]
    xh = x4[pick].cpu().numpy()
    met = lat.calc_metrics(x4)
    pl = met['plaqs'][pick].cpu().numpy()
    qi = met['intQ'][pick].cpu().numpy()
    f = ops.su3_unpack(ops.su3_force_n(ops.su3_pack(x4[pick]), args.beta, L), L).cpu().numpy()
    av, mx = osu3.check_su(xh)
    out = {'chains': pick,
           'plaq_abs_err': float(np.abs(pl - osu3.plaqs(xh)).max()),
           'intQ_abs_err': float(np.abs(qi - osu3.int_charges(xh)).max()),
           'force_abs_err': float(np.abs(f - osu3.grad_action(xh, args.beta)).max()),
           'checkSU_max': float(mx.max()), 'plaq': [round(float(p), 6) for p in pl]}
    ok = (out['plaq_abs_err'] < 1e-10 and out['intQ_abs_err'] < 1e-9
          and out['force_abs_err'] < 1e-11 and out['checkSU_max'] < 1e-10)
    # plain HMC, eps = 0.05 x 8 steps, 24 trajectories from the hot start at this beta
    old = dyn.config.verbose
    dyn.config.verbose = False
    xs, accs, plq = x, [], []
    beta = torch.tensor(args.beta)
    for i in range(24):
        xs, m = dyn.apply_transition_hmc((xs, beta), eps=0.05, nleapfrog=8)
        accs.append(float(m['acc'].mean()))
    dyn.config.verbose = old
    plq = float(lat.calc_metrics(xs.reshape(x4.shape))['plaqs'].mean())
    out.update({'hmc_acc_last8': round(float(np.mean(accs[-8:])), 4),
                'hmc_plaq_after_24_traj': round(plq, 5), 'ok': bool(ok)})
    assert ok, f'spot check against the oracle failed: {out}'
    if args.mode == 'l2hmc':
        out['heads'] = heads_check(dyn, x4, args)
        out['input_layer'] = input_check(dyn, x4, args)
    return out


def input_check(dyn, x4, args):
    """The vnet's input layer on the inputs of the timed run (su3_to_vec(projectSU(x)), su3_to_vec(projectSU(
    force))) through the int8-sliced kernel (csrc/gemm_sliced.hip) and through the fp64 MFMA kernel
    (untimed): their difference, and for chains 0 / 1 and the first 32 hidden units the distance of each
    from a long-double evaluation on the host.  `sliced` says which one the timed trajectories ran."""
    from l2hmc import _ops as ops
    nb = args.nchains
    xn = ops.su3_pack(x4)
    vnet = dyn._get_vnet(0)
    fn, _, w = dyn._v_inputs_n(vnet, xn, torch.tensor(args.beta), None)
    res = {'sliced': bool(w is not None and w.get('input_img') is not None and ops.USE_SLICED_INPUT[0]
                          and dyn.sliced_input)}
    if not res['sliced']:
        return res
    xv = ops.su3_projsu_vec8_n(xn).reshape(nb, -1)
    fv = ops.su3_projsu_vec8_n(fn).reshape(nb, -1)
    za = vnet.hidden_flat(xv, fv, dict(w, hidden=[]), sliced_exp=ops.SLICED_INPUT_EXP)
    try:
        ops.USE_SLICED_INPUT[0] = False
        zb = vnet.hidden_flat(xv, fv, dict(w, hidden=[]), sliced_exp=ops.SLICED_INPUT_EXP)
    finally:
        ops.USE_SLICED_INPUT[0] = True
    res['max_abs_diff'] = float((za - zb).abs().max())
    res['input_abs_max'] = float(max(xv.abs().max(), fv.abs().max()))
    LD, rows, cols = np.longdouble, [0, min(1, nb - 1)], slice(0, 32)
    ld = lambda t: t.detach().cpu().numpy().astype(LD)
    pre = (ld(xv[rows]) @ ld(w['wx'][cols]).T + ld(fv[rows]) @ ld(w['wv'][cols]).T
           + ld(w['bx'][cols]) + ld(w['bv'][cols]))
    if vnet.act == 'tanh':
        want = np.tanh(pre)
        for name, got in (('sliced', za), ('fp64_mfma', zb)):
            res[f'{name}_vs_long_double'] = float(np.abs(ld(got[rows][:, cols]) - want).max())
    return res


def heads_check(dyn, x4, args):
    """One heads + momentum update with the NETWORK AND STATE OF THE TIMED RUN through the int8-sliced
    kernel and through the fp64 MFMA kernel (untimed): their difference, and for chains 0 / 1 and the
    first 512 entries the distance of each from a long-double evaluation on the host (numpy float128:
    x87 80-bit).  The sliced kernel is the one the timed trajectories ran when `sliced` is true."""
    from l2hmc import _ops as ops
    nb = args.nchains
    xn = ops.su3_pack(x4)
    vnet = dyn._get_vnet(0)
    fn, z, w = dyn._v_inputs_n(vnet, xn, torch.tensor(args.beta), None)
    hs = w['heads_scaled']
    res = {'sliced': hs.get('sliced') is not None and ops.USE_SLICED_HEADS[0]}
    if not res['sliced']:
        return res
    g = torch.Generator(device='cuda').manual_seed(7)
    v = torch.randn(fn.reshape(nb, -1).shape, dtype=torch.float64, device='cuda', generator=g)
    v = torch.complex(v, torch.randn(v.shape, dtype=torch.float64, device='cuda', generator=g))
    f = fn.reshape(nb, -1)
    nw, eps = (vnet.nw.s, vnet.nw.t, vnet.nw.q), 0.05
    va = v.clone()
    la = ops.vnet_heads_vupdate_(z, hs, nw, va, f, eps, True)
    try:
        ops.USE_SLICED_HEADS[0] = False
        vb = v.clone()
        lb = ops.vnet_heads_vupdate_(z, hs, nw, vb, f, eps, True)
    finally:
        ops.USE_SLICED_HEADS[0] = True
    res['max_abs_diff_v'] = float((va - vb).abs().max())
    res['max_abs_diff_logdet'] = float((la - lb).abs().max())
    res['logdet_abs_max'] = float(lb.abs().max())
    LD, rows, cols = np.longdouble, [0, min(1, nb - 1)], slice(0, 512)
    zz = z[rows].cpu().numpy().astype(LD)
    y = {k: zz @ hs[k][0][cols].cpu().numpy().astype(LD).T + hs[k][1][cols].cpu().numpy().astype(LD) for k in 'stq'}
    sv = hs['s'][2][cols].cpu().numpy().astype(LD) * np.tanh(y['s'])
    qv = hs['q'][2][cols].cpu().numpy().astype(LD) * np.tanh(y['q'])
    tv = LD(nw[1]) * y['t']
    h = LD(0.5) * LD(eps)
    es, eq = np.exp(h * sv), np.exp(LD(eps) * qv)
    v0, f0 = v[rows][:, cols].cpu().numpy(), f[rows][:, cols].cpu().numpy()
    wr = es * v0.real.astype(LD) - h * (f0.real.astype(LD) * eq + tv)
    wi = es * v0.imag.astype(LD) - h * (f0.imag.astype(LD) * eq)
    for name, got in (('sliced', va), ('fp64_mfma', vb)):
        gh = got[rows][:, cols].cpu().numpy()
        res[f'{name}_vs_long_double'] = float(max(np.abs(gh.real.astype(LD) - wr).max(),
                                                  np.abs(gh.imag.astype(LD) - wi).max()))
    return res


def secondary(dyn, x, beta, args, nlf_exec):
    """Untimed-region extras SURVEY.md 8(d) asks to report beside the headline: the same
    trajectory with per-step metrics (verbose=True, the YAML default) and the plain-HMC
    baseline sampler (apply_transition_hmc), each over 2 steps after 1 warm-up."""
    res = {}

    bad = []

    def rate(fn, warm=4, n=4):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # every secondary line is a real trajectory: a non-finite acceptance would make its rate meaningless
        if isinstance(out, tuple) and isinstance(out[1], dict) and 'acc' in out[1] \
                and not bool(torch.isfinite(out[1]['acc']).all()):
            bad.append(len(res))
        return round(args.nchains * nlf_exec * n / dt, 1)
    old = dyn.config.verbose
    try:
        dyn.config.verbose = True
        res['l2hmc_verbose_true'] = rate(lambda: dyn((x, beta)))
        dyn.config.verbose = False
        # sampler-loop convenience (opt-in): the transition keeps the native-layout original of the
        # x it returned and reuses it when that very tensor comes straight back
        try:
            dyn.cache_native_output = True
            state = {'x': x}

            def chained():
                state['x'], mm = dyn((state['x'], beta))
                return state['x'], mm
            res['l2hmc_native_output_cache'] = rate(chained)
        finally:
            dyn.cache_native_output = False
            dyn._xcache = None
        # the same trajectory with the three head layers scaled by 0.03 (random-init heads are O(1)
        # per entry: dH ~ -50 from a hot start and every chain rejects; scaled, the accept
        # probability is inside (0, 1) -- tests/test_sizes_gpu.py pins exactly this set-up to the
        # oracle): the accept / select path exercised with both outcomes, same shapes and kernels
        if dyn._networks_built:
            heads = [dyn.vnet.scale.layer, dyn.vnet.transl, dyn.vnet.transf.layer]
            keep = [(l.weight.detach().clone(), l.bias.detach().clone()) for l in heads]
            try:
                with torch.no_grad():
                    for l in heads:
                        l.weight.mul_(0.03)
                        l.bias.mul_(0.03)
                box = {}

                def scaled():
                    xo_, box['m'] = dyn((x, beta))
                    return xo_, box['m']
                r = rate(scaled)
                a = box['m']['acc']
                res['l2hmc_scaled_heads'] = {
                    'value': r, 'head_scale': 0.03, 'accept_prob_mean': round(float(a.mean()), 4),
                    'accept_prob_min': round(float(a.min()), 4), 'accept_prob_max': round(float(a.max()), 4),
                    'accepted_fraction': round(float(box['m']['acc_mask'].mean()), 4)}
            finally:
                with torch.no_grad():
                    for l, (w0, b0) in zip(heads, keep):
                        l.weight.copy_(w0)
                        l.bias.copy_(b0)
        res['hmc'] = rate(lambda: dyn.apply_transition_hmc((x, beta), eps=0.01,
                                                           nleapfrog=nlf_exec))
        # the headline trajectory replayed from a HIP graph behind forward() (Dynamics.auto_graph_su3, opt-in: at this
        # size a replay is worth ~1 %)
        try:
            dyn.auto_graph_su3 = True
            res['l2hmc_auto_graph_su3'] = rate(lambda: dyn((x, beta)))
        finally:
            dyn.auto_graph_su3 = False
            dyn._graphs.clear()
        # an explicit graph (Dynamics.make_graphed): outputs are views of its static buffers, nothing copied out
        try:
            g = dyn.make_graphed(x, beta=float(beta))
            res['l2hmc_hip_graph'] = rate(lambda: g(x))
            del g
        except Exception as e:  # noqa: BLE001  (reported, never fatal for the headline)
            res['l2hmc_hip_graph'] = f'failed: {type(e).__name__}: {e}'[:200]
    finally:
        dyn.config.verbose = old
    res['unit'] = 'chain*leapfrog-steps/s'
    if bad:
        res['non_finite_acceptance_in_entries'] = bad
    return res


FP16_MFMA_PEAK_TF = 2500.0      # dense fp16 / bf16 MFMA (MI355X_MICROARCH.md; not the 2:1-sparsity figure)
FP32_MFMA_PEAK_TF = 157.3       # v_mfma_f32_16x16x4_f32


def _u1_bytes(name, a):
    """algorithmic HBM bytes of one launch of the fused half-precision heads + update kernel: the
    fp32 field it updates in place (read + write), the second fp32 operand, Z and the 3 heads' W"""
    if name == 'l2q_u1_heads_update_h':
        M, K, N = a[2], a[3], a[4]
        return 3.0 * M * N * 4 + 2.0 * M * K + 3.0 * 2.0 * N * K
    return 0.0


def _u1_flops(name, a):
    """MFMA flops of one launch of a U(1) network entry point from its C-ABI arguments."""
    if name == 'l2q_gemm_h':
        return 2.0 * a[4] * a[5] * (a[6] + a[9])
    if name == 'l2q_gemm_h_u1x':
        return 2.0 * a[5] * a[6] * (2 * a[7] + a[10])
    if name == 'l2q_u1_heads_update_h':
        return 6.0 * a[2] * a[3] * a[4]
    if name == 'l2q_conv_gemm_periodic_h':
        nb, C, H, W, k, cout = a[7], a[8], a[9], a[10], a[11], a[15]
        return 2.0 * nb * (H + k - 1) * (W + k - 1) * cout * C * k * k
    if name == 'l2q_conv_gemm_periodic_f32':
        nb, C, H, W, k, cout = a[5], a[6], a[7], a[8], a[9], a[13]
        return 2.0 * nb * (H + k - 1) * (W + k - 1) * cout * C * k * k
    if name == 'l2q_gemm_f32':
        return 2.0 * a[2] * a[3] * (a[4] + a[7])
    return 0.0


def secondary_u1(steps=3, only=None):
    """Untimed-region block (rank 0, N = 1): the other two single-GPU BASELINE configs run as
    themselves -- cfg-2 (2D U(1) 16x16, beta 4, 2048 chains, nleapfrog 8, fp32) and cfg-3 (64x64,
    beta 6, 8192 chains, nleapfrog 8, fp16 nets / fp32 action), each with the reference's default
    network (conf/network + conf/conv defaults: conv [8,16,32,64,128] + units [16]*4) and with a
    dense one -- `steps` merged trajectories after one warm-up, x fed through compat_proj like the
    reference's eval loop.  Reports chain*LF/s and, from HIP events around every C-ABI call, the
    dominant kernel with its share and (MFMA entry points) its fraction of the dense peak.
    Parity of exactly these shapes: tests/test_sizes_gpu.py::test_cfg2_* / test_cfg3_*."""
    import gc
    import l2hmc.configs as cfgs
    from l2hmc import native
    import l2hmc._ops as ops
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.lattice.u1.pytorch.lattice import LatticeU1
    from l2hmc.network.pytorch.network import NetworkFactory
    old_dtype = torch.get_default_dtype()
    torch.set_default_dtype(torch.float32)
    conv_default = dict(filters=[8, 16, 32, 64, 128], sizes=[5, 3, 3, 3, 2], pool=[2, 2, 2, 2, 2])
    cases = [('cfg2_default_net_fp32', [16, 16], 2048, 4.0, [16, 16, 16, 16], True, None),
             ('cfg2_dense_fp32', [16, 16], 2048, 4.0, [16, 16, 16, 16], False, None),
             ('cfg3_dense256_fp16', [64, 64], 8192, 6.0, [256, 256], False, 'fp16'),
             ('cfg3_default_net_fp16', [64, 64], 8192, 6.0, [16, 16, 16, 16], True, 'fp16')]
    out = {}
    orig_native, orig_ops = native.call, ops.N.call
    for tag, L, nb, beta, units, conv, prec in cases:
        if only and tag not in only:
            continue
        try:
            torch.manual_seed(9992)
            np.random.seed(9992)
            nlf = 8
            dc = cfgs.DynamicsConfig(nchains=nb, group='U1', latvolume=L, nleapfrog=nlf, eps=0.1,
                                     eps_hmc=0.1, verbose=False)
            nc = cfgs.NetworkConfig(units=units, activation_fn='leaky_relu', dropout_prob=0.2,
                                    use_batch_norm=True)
            cc = cfgs.ConvolutionConfig(**conv_default) if conv else cfgs.ConvolutionConfig()
            spec = cfgs.InputSpec(xshape=tuple(dc.xshape), xnet={'x': [dc.xdim, 2], 'v': [dc.xdim]},
                                  vnet={'x': [dc.xdim], 'v': [dc.xdim]})
            lat = LatticeU1(nb, L)
            dyn = Dynamics(lat.action, dc, NetworkFactory(spec, nc, cc)).eval()
            dyn.set_net_precision(prec)
            x = lat.random()
            bt = torch.tensor(beta)
            for _ in range(max(1, int(dyn.auto_graph_after))):     # warm-up (weight copies, pool; the default
                xo, m = dyn((x, bt))                               # Dynamics captures its graph on the 3rd sighting)
                x = dyn.g.compat_proj(xo.reshape(x.shape))
            torch.cuda.synchronize()
            recs = []

            def timed(name, *a):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                orig_native(name, *a)
                e1.record()
                recs.append((name, _u1_flops(name, a), e0, e1, _u1_bytes(name, a)))
            # (1) the DEFAULT Dynamics, un-instrumented: launch-bound lattices replay a HIP graph by themselves
            t0 = time.perf_counter()
            for _ in range(steps):
                xo, m = dyn((x, bt))
                x = dyn.g.compat_proj(xo.reshape(x.shape))
            torch.cuda.synchronize()
            dt_default = (time.perf_counter() - t0) / steps
            auto_graphed = bool(dyn._graphs)
            # (2) eager launches with HIP events around every C-ABI call: the kernel table
            dyn.auto_graph = False
            native.call = timed
            ops.N.call = timed
            t0 = time.perf_counter()
            for _ in range(steps):
                xo, m = dyn((x, bt))
                x = dyn.g.compat_proj(xo.reshape(x.shape))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            native.call, ops.N.call = orig_native, orig_ops
            assert bool(torch.isfinite(x).all())
            # the same trajectory replayed from a HIP graph (Dynamics.make_graphed; VERDICT r03 item 7: the
            # dense cfg-2 block is launch-bound in eager mode)
            try:
                gr = dyn.make_graphed(x, beta=float(beta))
                xo, _ = gr(x)
                x = dyn.g.compat_proj(xo.reshape(x.shape))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    xo, mg = gr(x)
                    x = dyn.g.compat_proj(xo.reshape(x.shape))
                torch.cuda.synchronize()
                dtg = (time.perf_counter() - t0) / steps
                assert bool(torch.isfinite(x).all())
                graphed = {'ms_per_trajectory': round(dtg * 1e3, 3), 'value': round(nb * 2 * nlf / dtg, 1),
                           'accept_prob_mean': round(float(mg['acc'].mean()), 4)}
                gr = None
            except Exception as e:  # noqa: BLE001
                graphed = f'failed: {type(e).__name__}: {e}'[:200]
            agg = {}
            for name, fl, e0, e1, by in recs:
                d = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
                d[0] += 1
                d[1] += e0.elapsed_time(e1) * 1e-3
                d[2] += fl
                d[3] += by
            tot = sum(v[1] for v in agg.values())
            name, (cnt, tt, fl, by) = max(agg.items(), key=lambda kv: kv[1][1])
            top = {k: {'share': round(v[1] / tot, 4), 'launches_per_trajectory': v[0] // steps,
                       'avg_ms': round(v[1] / v[0] * 1e3, 4)}
                   for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]}
            dom = {'kernel': name, 'share_of_kernel_time': round(tt / tot, 4),
                   'launches_per_trajectory': cnt // steps, 'avg_ms': round(tt / cnt * 1e3, 4)}
            if fl > 0:
                peak = FP32_MFMA_PEAK_TF if name.endswith('f32') else FP16_MFMA_PEAK_TF
                f_mfma = fl / tt / 1e12 / peak
                f_hbm = by / tt / 1e9 / HBM_PEAK_GBS
                if f_hbm > f_mfma:       # the roofline that binds is the one closer to its peak
                    dom.update({'bound': 'hbm', 'achieved': round(by / tt / 1e9, 1),
                                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(f_hbm, 4),
                                'mfma_frac': round(f_mfma, 4)})
                else:
                    dom.update({'bound': 'mfma', 'achieved': round(fl / tt / 1e12, 2), 'peak': peak,
                                'unit': 'TFLOP/s', 'frac': round(f_mfma, 4)})
            out[tag] = {'workload': f'2D U(1) {L[0]}x{L[1]}, beta={beta}, {nb} chains, nleapfrog={nlf} '
                                    f'({2 * nlf} LF steps/trajectory), '
                                    f'{"default conv stack + " if conv else ""}units {units}, '
                                    f'{prec or "fp32"} nets / fp32 action',
                        'ms_per_trajectory': round(dt_default * 1e3, 3),
                        'value': round(nb * 2 * nlf / dt_default, 1), 'unit': 'chain*leapfrog-steps/s',
                        'default_path': 'HIP-graph replay (Dynamics.auto_graph)' if auto_graphed else 'eager launches',
                        'eager_instrumented_ms_per_trajectory': round(dt * 1e3, 3),
                        'steps': steps, 'accept_prob_mean': round(float(m['acc'].mean()), 4),
                        'kernel_time_fraction_of_wall': round(tot / (dt * steps), 4),
                        'dominant_kernel': dom, 'kernels': top, 'hip_graph': graphed}
            # HBM / fabric bytes per launch of the dominant kernel from the PMC passes taken at exactly this shape
            # (tools/pmc_collect.sh with tools/kprof_u1_cfg3.py): the v- and the x-update are two instantiations
            try:
                pj = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic_u1_cfg3.json')))
                ent = [v for k, v in pj.items() if k.split('@')[0].split(':')[0] == dom.get('kernel')
                       and v.get('lattice') == L and v.get('nchains') == nb]
                if tag == 'cfg3_dense256_fp16' and ent:
                    dom['traffic'] = round(sum(e['total_bytes'] for e in ent) / len(ent), 1)
                    dom['traffic_source'] = '; '.join(sorted({e['source'].split(' ')[0] for e in ent}))
            except Exception:  # noqa: BLE001
                pass
        except Exception as e:  # noqa: BLE001  (reported, never fatal for the headline)
            import traceback
            out[tag] = f'failed: {type(e).__name__}: {e} | ' + traceback.format_exc()[-400:]
        finally:
            native.call, ops.N.call = orig_native, orig_ops
            dyn = lat = x = xo = m = None
            gc.collect()
            torch.cuda.empty_cache()
    torch.set_default_dtype(old_dtype)
    return out


def published_u1(steps=10):
    """Untimed-region block (rank 0, N = 1): THE configuration the reference publishes numbers for
    (reports/l2hmc-2dU1/README.md:366-381, 472-590, 700-733, 826-841) -- 2D U(1) 16x16,
    beta 4, 2048 chains, nleapfrog 4 (8 LF steps), precision fp16, conv none, conf/ defaults otherwise (units
    [16]x4, leaky_relu, dropout 0.2, BatchNorm, separate + split networks, verbose): `Trainer.train_step`
    (autocast-style 16-bit tape forward, LossScaler, reverse sweep, fused Adam), `eval_step` and
    `hmc_step(nleapfrog=8, eps=0.25)` on 128 chains -- seconds per step next to the reference's `dt` on one
    A100-SXM4-80GB.  Cross-hardware context, not a like-for-like comparison and not `vs_baseline`."""
    import gc
    import l2hmc.configs as cfgs
    from l2hmc.trainers.pytorch.trainer import Trainer
    old_dtype = torch.get_default_dtype()
    torch.set_default_dtype(torch.float32)
    out = {}
    try:
        torch.manual_seed(76043)
        np.random.seed(76043)
        cfg = cfgs.get_config(['precision=fp16', 'dynamics.nleapfrog=4', 'dynamics.nchains=2048',
                               'dynamics.eps=0.05', 'dynamics.latvolume=[16,16]', 'conv=none', 'seed=76043'])
        tr = Trainer(cfg)
        nparams = tr.count_parameters()
        beta = 4.0
        x = tr.lattice.random()

        def clock(fn, x, n, warm):
            for _ in range(warm):
                x, m = fn(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                x, m = fn(x)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n, x, m
        dt_tr, x, m = clock(lambda xx: tr.train_step((xx, beta)), x, steps, 3)
        x128 = x[:128].contiguous()
        dt_ev, _x, me = clock(lambda xx: tr.eval_step((xx, beta)), x128, steps, 3)
        dt_h, _x, mh = clock(lambda xx: tr.hmc_step((xx, beta), nleapfrog=8, eps=0.25), x128, steps, 3)
        nlf = 4
        out = {
            'workload': '2D U(1) 16x16, beta=4, nleapfrog=4 (8 LF steps), precision fp16 (16-bit layers, fp32 '
                        'lattice arithmetic), conv none, units [16,16,16,16], dropout 0.2, BatchNorm, separate + '
                        'split networks, verbose=True; train: 2048 chains; eval / hmc: 128 chains',
            'trainable_parameters': nparams, 'reference_parameters': 598344,
            'train_step_s': round(dt_tr, 5), 'train_chain_lf_per_s': round(2048 * 2 * nlf / dt_tr, 1),
            'eval_step_s': round(dt_ev, 5), 'hmc_step_s': round(dt_h, 5),
            'loss': round(float(m['loss']), 5), 'loss_scale': m.get('loss_scale'),
            'train_acc_mean': round(float(m['acc'].mean()), 4),
            'eval_acc_mean': round(float(me['acc'].mean()), 4), 'hmc_acc_mean': round(float(mh['acc'].mean()), 4),
            'reference_a100': {'train_step_s': [0.29, 0.31], 'eval_step_s': 0.119, 'hmc_step_s': [0.017, 0.018],
                               'hardware': '1x NVIDIA A100-SXM4-80GB (ALCF ThetaGPU), PyTorch + autocast fp16',
                               'source': 'reports/l2hmc-2dU1/README.md:366-381, 574-590, 700-706, 826-835'},
            'note': 'cross-hardware context only: the reference publishes these incidental dt fields and nothing '
                    'for SU(3); seconds per step include the trainer-side loss / lattice metrics like the '
                    "reference's timer does"}
    except Exception as e:  # noqa: BLE001  (reported, never fatal for the headline)
        import traceback
        out = f'failed: {type(e).__name__}: {e} | ' + traceback.format_exc()[-600:]
    finally:
        tr = x = None
        gc.collect()
        torch.cuda.empty_cache()
        torch.set_default_dtype(old_dtype)
    return out


def load_traffic(args):
    """profiles/pmc_traffic.json: HBM/fabric bytes per launch from separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes (tools/pmc_collect.sh).  An entry is used only if it was
    measured on THIS lattice / chain count and on the kernel template the library dispatches now
    (l2q_kernel_name); otherwise traffic is null with the reason."""
    from l2hmc import native
    f = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    pmc = json.load(open(f)) if os.path.exists(f) else {}
    L = [int(i) for i in args.lattice]

    def traffic(name):
        t = pmc.get(f'{name}@{"x".join(map(str, L))}x{args.nchains}', pmc.get(name))
        if t is None:
            return None, 'no PMC pass recorded for this entry point'
        if t.get('lattice') != L or t.get('nchains') != args.nchains:
            return None, (f"PMC pass was taken at lattice {t.get('lattice')} x {t.get('nchains')} "
                          f"chains, this run is {L} x {args.nchains}")
        now = native.kernel_name(name, L)
        if now and now not in t.get('kernel', ''):
            return None, f"PMC pass was taken on `{t.get('kernel')}`, the library now dispatches `{now}`"
        return t['total_bytes'], t.get('source')
    return traffic


def allreduce_probe(tr, dist, world, reps=5):
    """Time of the ONE collective of a training step: all-reduce of the flat gradient arena
    (trainers/pytorch/trainer.py:246-257, 1296-1304 = DDP's bucketed all-reduce in the reference)."""
    nbytes = sum(g['grad'].numel() * g['grad'].element_size() for g in tr.arena.groups.values())
    out = {'bytes': nbytes, 'ranks': world}
    if dist is None:
        out['note'] = 'single rank: no collective'
        return out
    for _ in range(2):
        tr.arena.all_reduce()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        tr.arena.all_reduce()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t = torch.tensor([dt], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    out.update({'ms': round(dt * 1e3, 3), 'algbw_GBps': round(nbytes / dt / 1e9, 2),
                'busbw_GBps': round(nbytes / dt / 1e9 * 2 * (world - 1) / world, 2)})
    return out


def collective_probe(dist, world, n_params, reps=5):
    """Sampling needs no collective, so the default mode would never touch RCCL; the one exchange
    of the path is the training-gradient all-reduce (trainers/pytorch/trainer.py:246-257,
    1296-1304 of the reference = DDP's bucketed all-reduce).  This times exactly that exchange --
    ONE all-reduce of a flat fp64 buffer of the gradient arena's size -- outside the timed region,
    so the driver's N = 2, 4, 8 runs also put a number on RCCL over xGMI."""
    buf = torch.ones(n_params, dtype=torch.float64, device='cuda')
    dist.all_reduce(buf)
    torch.cuda.synchronize()
    # self-check: every rank contributed exactly once (a mis-wired communicator must not pass silently)
    if not bool((buf == float(world)).all()):
        raise RuntimeError(f'collective_probe: all-reduce of ones over {world} ranks gave '
                           f'{float(buf.min())}..{float(buf.max())}')
    buf.fill_(1.0)
    dist.all_reduce(buf)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t = torch.tensor([dt], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    nbytes = n_params * 8
    return {'what': 'all-reduce(sum) of a flat fp64 buffer the size of the cfg gradient arena '
                    '(vnet parameters), untimed region', 'bytes': nbytes, 'ranks': world,
            'ms': round(dt * 1e3, 3), 'algbw_GBps': round(nbytes / dt / 1e9, 2),
            'busbw_GBps': round(nbytes / dt / 1e9 * 2 * (world - 1) / world, 2)}


def native_comm_probe(dist, world, n_params, reps=3):
    """The same exchange through the C ABI (l2q_comm_init / l2q_allreduce_grads, librccl resolved at
    run time) -- also with ONE rank, so that a 1-GPU run proves librccl loads and that one flat
    buffer of the gradient arena's size goes through on the launch stream (untimed region; never
    fatal for the headline)."""
    try:
        from l2hmc.utils.dist import NativeComm
        with NativeComm(rank=0 if dist is None else dist.get_rank(), world_size=world) as c:
            buf = torch.ones(n_params, dtype=torch.float64, device='cuda')
            c.all_reduce_(buf)
            torch.cuda.synchronize()
            ok = bool((buf == float(world)).all())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                c.all_reduce_(buf)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
        out = {'route': 'l2q_allreduce_grads (C ABI -> librccl)', 'ranks': world,
               'bytes': n_params * 8, 'ms': round(ms, 3), 'sum_correct': ok}
        if world > 1:             # (one rank moves nothing over xGMI: a bandwidth figure would be meaningless)
            out['algbw_GBps'] = round(n_params * 8 / ms / 1e6, 2)
            out['busbw_GBps'] = round(n_params * 8 / ms / 1e6 * 2 * (world - 1) / world, 2)
        else:
            out['note'] = 'one rank: proves librccl loads and the flat buffer goes through on the launch stream'
        return out
    except Exception as e:  # noqa: BLE001
        return {'route': 'l2q_allreduce_grads', 'error': f'{type(e).__name__}: {e}'[:300]}


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(respawn_under_torchrun(args))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    # stdout carries ONE line, the JSON record: everything else a library may print there (librccl
    # writes its version banner to stdout when the first communicator is created, through C stdio,
    # flushed at exit) is sent to stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with '
                 f'--nproc-per-node {args.gpus} (or from a bare shell, which self-spawns)')
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU path)'
    # L2Q_BENCH_SHARE_GPU=1 + L2Q_BENCH_BACKEND=gloo: functional test of the N>1 path on a
    # 1-GPU box (all ranks on device 0); the real runs use one GPU per rank and RCCL.
    share = os.environ.get('L2Q_BENCH_SHARE_GPU') == '1'
    if not share and torch.cuda.device_count() < world:
        sys.exit(f'bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) visible')
    torch.cuda.set_device(0 if share else local_rank)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = os.environ.get('L2Q_BENCH_BACKEND', 'nccl')             # nccl = RCCL over xGMI
        dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    seed = 9992
    train = args.mode == 'train'
    if train:
        tr = build_trainer(args, seed)
        dyn, lat = tr.dynamics, tr.lattice
    else:
        dyn, lat = build(args, seed)
    from l2hmc.utils.dist import chain_seed
    cseed = chain_seed(seed, rank)                 # chain streams: distinct from the model seed on every rank
    x = hot_start(args, seed=cseed)
    # The headline times the DEFAULT `Dynamics` (what a user gets): every trajectory packs the x it
    # is handed into the native layout, like the reference's eval_step pays compat_proj.  The opt-in
    # native-output cache (skips that 0.2 ms pack when x_out is fed straight back) is reported as
    # `secondary.l2hmc_native_output_cache`.
    if train:
        # per-rank stream for momenta / accept uniforms (chains differ, the model does not)
        torch.manual_seed(cseed)
        torch.cuda.manual_seed(cseed)
    beta = torch.tensor(args.beta)
    nlf_exec = 2 * args.nleapfrog

    def step(xin):
        if train:
            return tr.train_step((xin, args.beta))
        if args.mode == 'l2hmc':
            return dyn((xin, beta))
        return dyn.apply_transition_hmc((xin, beta), eps=0.01, nleapfrog=nlf_exec)

    # set-up, not a measured or counted step: the first two trajectories of a process grow
    # PyTorch's caching-allocator pool (hipMalloc is synchronous: 28 + 5 calls, 2.6 s + 1.4 s at
    # the 16^4 shard) and build the native-order weight copies.  Doing that
    # here keeps the W warm-up steps of the contract what they are meant to be even for W = 0 / 1.
    # (train mode: the first step also builds the parameter arena; weights stay in sync because
    # every rank applies the same all-reduced gradient.)
    for _ in range(1 if train else 2):
        step(x)
    torch.cuda.synchronize()
    setup_steps = 1 if train else 2
    graphed = bool(getattr(dyn, '_graphs', None))
    if graphed:
        # A Dynamics that replays its eval-mode transitions from a HIP graph (Dynamics._auto_graphed: the small U(1)
        # lattices by default, SU(3) with auto_graph_su3): a few set-up trajectories after the capture, so that the W
        # warm-up and K timed steps of the contract see the sampler's steady state.  (--settle 0 skips them.)
        for _ in range(args.settle):
            x, m = step(x)
        setup_steps += args.settle
        torch.cuda.synchronize()
    timer = KernelTimer()
    timer.install()
    for _ in range(args.warmup):
        x, m = step(x)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region carries NO instrumentation: `value` is what a user's loop gets.  The per-kernel HIP
    # events (two per C-ABI call, created and recorded on the launch stream) run over the SAME K steps
    # repeated immediately afterwards: inside the timed region they cost 0.4 ms of a 21.6 ms trajectory and
    # 4-13 ms of a 150 ms train step (profiles/r04u_bench_timer_ab.txt), which is instrumentation, not work.
    # `instrumented_ms_per_step` reports that second pass; the kernel table and the rooflines come from it.
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x, m = step(x)
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = os.environ.get('L2Q_BENCH_NO_KTIMER') != '1'     # (experiment switch: no kernel table)
    # (HIP events cannot be recorded inside a graph replay -- torch: "External events are disallowed in rocm" --
    # so the instrumented pass launches the SAME kernels eagerly: its per-kernel durations are those of the lower
    # clock state, i.e. an upper bound of what ran in the timed region; the rocprofv3 table of the graph replay
    # itself is profiles/r05*_bench_l2hmc_kernel_stats.txt)
    skip_instr = os.environ.get('L2Q_BENCH_SKIP_INSTRUMENTED') == '1'
    # (L2Q_BENCH_SKIP_INSTRUMENTED=1: no second pass -- the LAST K trajectories of the process are then the timed ones,
    # which is what tools/kstats.sh selects for the rocprofv3 table of the graph replay)
    if graphed and not skip_instr:
        dyn.auto_graph = False
        step(x)
    t1 = time.perf_counter()
    for _ in range(0 if skip_instr else args.steps):
        x, m = step(x)
    barrier()
    dt_instr = (time.perf_counter() - t1) if not skip_instr else dt
    timer.enabled = False
    if graphed:
        dyn.auto_graph = True
    per_rank = None
    ranks_check = None
    if dist is not None:
        # the N > 1 run checks itself: N distinct devices, the SAME model on every rank (built from the
        # seed), DIFFERENT chains on every rank -- or it fails loudly instead of reporting a number
        from l2hmc.utils.dist import model_checksum
        msum = model_checksum(dyn)
        xsum = float(x.double().abs().sum() if not x.is_complex() else torch.view_as_real(x).abs().sum())
        try:
            uid = str(torch.cuda.get_device_properties(torch.cuda.current_device()).uuid)
        except Exception:  # noqa: BLE001
            uid = f'device-{torch.cuda.current_device()}'
        info = [None] * world
        from l2hmc import _ops as _l2q_ops
        dist.all_gather_object(info, {'rank': rank, 'device': torch.cuda.current_device(), 'uuid': uid,
                                      'model': msum, 'chains': xsum,
                                      'paths': _l2q_ops.mem_gate_signature()})
        if len({i['model'] for i in info}) != 1:
            raise RuntimeError(f'bench.py: ranks hold different models: {info}')
        if len({i['chains'] for i in info}) != world:
            raise RuntimeError(f'bench.py: ranks run the same chains: {info}')
        if not share and len({i['uuid'] for i in info}) != world:
            raise RuntimeError(f'bench.py: ranks share a device: {info}')
        # the memory-gated kernel choices (int8-sliced vs fp64 MFMA ...) are made by all ranks together before the
        # first kernel that depends on them (l2hmc._ops.mem_gate): every rank must report the same table
        if len({i['paths'] for i in info}) != 1:
            raise RuntimeError(f'bench.py: ranks took different memory-gated kernel paths: {info}')
        ranks_check = {'distinct_devices': len({i['uuid'] for i in info}), 'same_model': True,
                       'distinct_chain_sets': world, 'same_kernel_paths': True,
                       'kernel_paths': [list(kv) for kv in info[0]['paths']]}
    if dist is not None:
        # every rank's own time for its K steps (rank 0 reports min / median / max of the per-rank
        # rates next to the contract's max-over-ranks value)
        ts = [torch.zeros(1, dtype=torch.float64, device='cuda') for _ in range(world)]
        dist.all_gather(ts, torch.tensor([dt], dtype=torch.float64, device='cuda'))
        per_rank = sorted(float(t.item()) for t in ts)
        dt = per_rank[-1]
    acc = m['acc']
    assert torch.isfinite(acc).all() and torch.isfinite(x).all(), 'non-finite trajectory'
    blocking_ms = None
    if train and dist is not None:
        # A/B of the gradient exchange: the same K steps with ONE blocking all-reduce after the reverse sweep
        # instead of the slab-by-slab exchange that overlaps it (Trainer.overlap_grad_exchange); the
        # difference is what the overlap hides of the collective
        tr.overlap_grad_exchange = False
        step(x)
        barrier()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            x, _m2 = step(x)
        barrier()
        tb = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device='cuda')
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        blocking_ms = float(tb.item()) / args.steps * 1e3
        tr.overlap_grad_exchange = True
    ar = allreduce_probe(tr, dist, world) if train else None
    probe = None
    nprobe = None
    if not train and args.mode == 'l2hmc' and not args.no_comm_probe:
        nparam = sum(p.numel() for p in dyn.vnet.parameters())
        if dist is not None:
            probe = collective_probe(dist, world, nparam)
        # the C-ABI route: always with one rank (cannot hang); with N > 1 only on request -- a second
        # communicator next to torch's is not something to discover in the driver's scaling run
        if dist is None or os.environ.get('L2Q_BENCH_NATIVE_COMM') == '1':
            nprobe = native_comm_probe(dist, world, nparam)

    if rank == 0:
        V = int(np.prod(args.lattice))
        sites = args.nchains * V
        ks = timer.summary()
        total_k = sum(v[1] for v in ks.values())
        kernels = {}
        for name, (cnt, tt, fl) in sorted(ks.items(), key=lambda kv: -kv[1][1]):
            ent = {'launches': cnt, 'avg_ms': round(tt / cnt * 1e3, 4),
                   'share': round(tt / total_k, 4)}
            if name in ALG_BYTES and not (train and args.micro_batch):
                ent['GB/s'] = round(sites * ALG_BYTES[name] / (tt / cnt) / 1e9, 1)
            kernels[name] = ent
        if len(kernels) > 24:                      # train mode: ~60 entry points; keep the top
            kernels = dict(list(kernels.items())[:24])
        traffic = load_traffic(args)

        def hbm_roof(name):
            cnt, tt, _ = ks[name]
            ach = sites * ALG_BYTES[name] / (tt / cnt) / 1e9
            tr_, src = traffic(name)
            return {'kernel': name, 'symbol': native_name(name), 'bound': 'hbm',
                    'achieved': round(ach, 1),
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                    'traffic': tr_, 'traffic_source': src, 'avg_ms': round(tt / cnt * 1e3, 4),
                    'launches': cnt, 'algorithmic_bytes_per_launch': sites * ALG_BYTES[name],
                    'time_s': tt}

        def native_name(name):
            from l2hmc import native
            return native.kernel_name(name, [int(i) for i in args.lattice])

        def mfma_roof(names, label, flops, peak=FP64_MFMA_PEAK_TF, unit='TFLOP/s'):
            cnt = sum(ks[n][0] for n in names)
            tt = sum(ks[n][1] for n in names)
            ach = flops / tt / 1e12
            tr_, src = traffic(names[0])
            return {'kernel': label, 'bound': 'mfma', 'achieved': round(ach, 2),
                    'peak': peak, 'unit': unit,
                    'frac': round(ach / peak, 4), 'traffic': tr_,
                    'traffic_source': src, 'avg_ms': round(tt / cnt * 1e3, 4), 'launches': cnt,
                    'time_s': tt}

        rooflines = []
        if not (train and args.micro_batch):
            for fname in ('l2q_su3_force', 'l2q_su3_force_vec8', 'l2q_su3_force_kick'):
                if fname in ks:
                    rooflines.append(hbm_roof(fname))
            if 'l2q_su3_plaq_reduce' in ks:
                rooflines.append(hbm_roof('l2q_su3_plaq_reduce'))
            heads = [n for n in ('l2q_vnet_heads_vupdate_pair_f64', 'l2q_vnet_heads_vupdate_f64')
                     if n in ks]
            if heads:
                per = 3 * 2.0 * args.nchains * args.units[-1] * 36 * V       # s, t, q heads
                rooflines.append(mfma_roof(heads, 'l2q_vnet_heads_vupdate[_pair]_f64 (3 heads + '
                                           'v-update)', per * sum(ks[n][0] for n in heads)))
            if 'l2q_vnet_heads_vupdate_sliced_f64' in ks:
                # the same heads with the fp64 products rebuilt from 28 exact int8 slice products
                # (csrc/heads_sliced.hip).  `achieved` keeps the contract's meaning -- the ALGORITHMIC
                # flops of the three fp64 GEMMs / time, against the fp64 MFMA peak (the instruction this
                # replaces) --; `int8` prices the operations that actually execute (28 x as many) against
                # the int8 MFMA peak, which is the unit that bounds the kernel
                nm = 'l2q_vnet_heads_vupdate_sliced_f64'
                per = 3 * 2.0 * args.nchains * args.units[-1] * 36 * V
                r = mfma_roof([nm], 'l2q_vnet_heads_vupdate_sliced_f64 (3 heads, fp64 products from 28 int8 '
                              'slice GEMMs, + v-update)', per * ks[nm][0])
                tops = 28 * per * ks[nm][0] / ks[nm][1] / 1e12
                r['int8'] = {'ops_per_launch': 28 * per, 'achieved': round(tops, 1),
                             'peak': INT8_MFMA_PEAK_TOPS, 'unit': 'TOP/s',
                             'frac': round(tops / INT8_MFMA_PEAK_TOPS, 4)}
                rooflines.append(r)
            if 'l2q_gemm_sliced_f64' in ks:
                # the vnet's input layer the same way (csrc/gemm_sliced.hip: 28 int8 slice products per
                # fp64 product, the activations sliced on the fly)
                nm = 'l2q_gemm_sliced_f64'
                r = mfma_roof([nm], 'l2q_gemm_sliced_f64 (input layer, fp64 products from 28 int8 slice '
                              'GEMMs)', ks[nm][2])
                tops = 28 * ks[nm][2] / ks[nm][1] / 1e12
                r['int8'] = {'ops_per_launch': 28 * ks[nm][2] / ks[nm][0], 'achieved': round(tops, 1),
                             'peak': INT8_MFMA_PEAK_TOPS, 'unit': 'TOP/s',
                             'frac': round(tops / INT8_MFMA_PEAK_TOPS, 4)}
                rooflines.append(r)
            if 'l2q_gemm_f64' in ks:
                rooflines.append(mfma_roof(['l2q_gemm_f64'], 'l2q_gemm_f64 (input + hidden layers)',
                                           ks['l2q_gemm_f64'][2]))
        # `roofline` = the dominant kernel (largest share of the timed region) among those
        roofline = dict(max(rooflines, key=lambda r: r['time_s'])) if rooflines else None
        for r in rooflines + ([roofline] if roofline else []):
            r.pop('time_s', None)
        if roofline is not None:
            # BASELINE's metric has a second half, "plaquette-kernel HBM GB/s": the lattice kernels' own lines sit
            # INSIDE `roofline` (the record the driver keeps), next to the int8 pricing of the sliced GEMMs
            short = {'l2q_su3_plaq_reduce': 'plaq', 'l2q_su3_force': 'force', 'l2q_su3_force_vec8': 'force_vec8',
                     'l2q_su3_force_kick': 'force_kick'}
            roofline['hbm_kernels'] = {
                short[r['kernel']]: {'symbol': r['symbol'], 'GB/s': r['achieved'], 'peak': r['peak'],
                                     'frac': r['frac'], 'avg_ms': r['avg_ms'], 'launches': r['launches'],
                                     'traffic': r['traffic'],
                                     'algorithmic_bytes_per_launch': r['algorithmic_bytes_per_launch']}
                for r in rooflines if r['bound'] == 'hbm' and r['kernel'] in short}
            i8 = {('heads' if 'heads' in r['kernel'] else 'input_layer'): dict(r['int8'], avg_ms=r['avg_ms'])
                  for r in rooflines if 'int8' in r}
            if i8:
                roofline['int8'] = i8
        nchain_lf = world * args.nchains * nlf_exec * args.steps
        what = {'l2hmc': 'Dynamics.forward merged L2HMC', 'hmc': 'apply_transition_hmc',
                'train': 'Trainer.train_step (tape + reverse sweep + all-reduce + Adam)'}[args.mode]
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
                                   f'{args.nchains} chains/GPU, complex128/fp64, {what}, '
                                   f'nleapfrog={args.nleapfrog} ({nlf_exec} LF steps/trajectory), '
                                   f'vnet units {args.units}, verbose=False, default Dynamics switches '
                                   f'(mc_states.out.v is formed on first access)',
                       'global_chains': world * args.nchains, 'parallelism': f'chains sharded x{world}'},
            # ranks of the RCCL communicator that executed in this run (N = 1: the one-rank
            # communicator of the native probe, if it ran)
            'rccl_ranks': (world if (dist is not None and backend == 'nccl')
                           else (f'{world} ({backend})' if dist is not None
                                 else (1 if (nprobe or {}).get('sum_correct') else 0))),
            'roofline': roofline,
            'rooflines': rooflines,
            'instrumented_ms_per_step': round(dt_instr / args.steps * 1e3, 3),
            'launch_path': ('HIP-graph replay behind Dynamics.forward (Dynamics.auto_graph, default); the kernel '
                            'table / rooflines below come from the instrumented pass on eager launches of the same '
                            'kernels (HIP events cannot be recorded inside a replay): an upper bound of the '
                            'durations in the timed region') if graphed else 'eager launches',
            'setup_steps': setup_steps,
            'kernel_time_fraction_of_wall': round(total_k / dt_instr, 4),
            'kernels': kernels,
            'accept_prob_mean': round(float(acc.mean()), 4),
        }
        if args.mode == 'l2hmc' and float(acc.mean()) == 0.0:
            # default-initialised heads from a hot start reject every proposal (dH ~ -50; the reference would
            # too), so the timed loop takes the reject branch of the final select.  The rate does not depend on
            # it: `secondary.l2hmc_scaled_heads` runs the same trajectory with the heads scaled so that about
            # half of the chains accept (and x changes from step to step)
            out['accept_prob_note'] = ('random-init networks from a hot start reject everything; see '
                                       'secondary.l2hmc_scaled_heads for the same rate at acceptance ~0.5')
        if per_rank is not None:
            unit_per_rank = args.nchains * nlf_exec * args.steps
            rates = sorted(unit_per_rank / t for t in per_rank)
            out['per_rank'] = {'unit': 'chain*leapfrog-steps/s per rank', 'min': round(rates[0], 1),
                               'median': round(rates[len(rates) // 2], 1), 'max': round(rates[-1], 1),
                               'seconds_min': round(per_rank[0], 4), 'seconds_max': round(per_rank[-1], 4),
                               'rates': [round(r, 1) for r in rates], 'self_check': ranks_check}
        if probe is not None:
            out['grad_allreduce_probe'] = probe
        if nprobe is not None:
            out['grad_allreduce_probe_native'] = nprobe
        if train:
            out['train'] = {'params_trained': tr.arena.numel(), 'grad_allreduce': ar,
                            'grad_exchange': tr.last_exchange,
                            'blocking_exchange_ms_per_step': None if blocking_ms is None else round(blocking_ms, 3),
                            'micro_batch': args.micro_batch, 'loss': m.get('loss'),
                            'peak_mem_GiB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
        if world == 1 and not train:
            if not args.no_spot_check:
                out['spot_check'] = spot_check(dyn, lat, x, args)
            if not args.no_cpu_baseline:
                out['secondary'] = secondary(dyn, x, beta, args, nlf_exec)
                sh = out['secondary'].get('l2hmc_scaled_heads')
                if isinstance(sh, dict):
                    # the representative sampling line beside the headline's all-reject one (same kernels and shapes)
                    out['config']['workload'] += (
                        f'; acceptance {out["accept_prob_mean"]} with random-init heads (every chain rejects), '
                        f'{sh["accept_prob_mean"]} with the heads scaled by {sh["head_scale"]}: '
                        f'{sh["value"]} chain*LF/s (secondary.l2hmc_scaled_heads)')
                    out['config']['scaled_heads'] = {'value': sh['value'], 'accept_prob_mean': sh['accept_prob_mean'],
                                                     'accepted_fraction': sh['accepted_fraction']}
                out['cpu_baseline'] = cpu_baseline(dyn, args)
            if not args.no_u1:
                del dyn, lat, x, m
                torch.cuda.empty_cache()
                out['secondary_u1'] = secondary_u1()
                out['secondary_u1']['published_u1_16x16_fp16'] = published_u1()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
