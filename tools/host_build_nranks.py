"""N ranks build the cfg-4 bench model at once on the host ("per-rank host work x8 on one
node is untested"): wall time and peak RSS per rank, and a check that every rank holds the same weights (the
model is replicated by seeding, bench.py never broadcasts it).  gloo, CPU only:
    python tools/host_build_nranks.py [N=8]"""
import os, resource, sys, time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.argv = ['bench.py']
    import bench
    args = bench.parse()
    dist.barrier()
    t0 = time.perf_counter()
    dyn, lat = bench.build(args, 9992)
    dt = time.perf_counter() - t0
    # order-sensitive checksum of every parameter
    cs = torch.zeros(2, dtype=torch.float64)
    for i, p in enumerate(dyn.parameters()):
        q = p.detach().double().reshape(-1)
        cs[0] += q.sum() * (1 + (i % 7))
        cs[1] += (q * q).sum()
    lo, hi = cs.clone(), cs.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    rss = torch.tensor([dt, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6], dtype=torch.float64)
    all_ = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(all_, rss)
    if rank == 0:
        print(f'{world} ranks on {os.cpu_count()} host cores: build {min(a[0] for a in all_):.1f}-'
              f'{max(a[0] for a in all_):.1f} s per rank, peak RSS {max(a[1] for a in all_):.2f} GB per rank '
              f'({sum(a[1] for a in all_):.1f} GB in all); parameter checksums identical on all ranks: '
              f'{bool(torch.equal(lo, hi))} ({cs.tolist()})', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mp.spawn(worker, args=(n, 29000 + os.getpid() % 2000), nprocs=n)
