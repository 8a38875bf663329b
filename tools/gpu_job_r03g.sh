#!/usr/bin/env bash
# Kernel-trace record of the TIMED region (tools/kstats.sh with the per-trajectory marker) for the three
# SU(3) bench workloads, plus the whole-process tables; bench lines beside them.
set -u
cd "$(dirname "$0")/.."
tag="${1:-r03g}"
o="gpurun_out/$tag"; mkdir -p "$o"
export TMPDIR=/tmp
M=su3_assemble_tah_kernel
python bench.py > "$o/bench_l2hmc.json" 2> "$o/bench_l2hmc.err"
python bench.py --mode hmc --no-u1 > "$o/bench_hmc.json" 2>> "$o/bench_l2hmc.err"
python bench.py --lattice 16 16 16 16 --beta 6.2 --steps 3 --warmup 2 --no-cpu-baseline --no-spot-check --no-u1 > "$o/bench_cfg5_shard.json" 2> "$o/bench_cfg5.err"
KSTATS_MARKER=$M KSTATS_LAST=5 bash tools/kstats.sh "$o/bench_l2hmc_kernel_stats.txt" python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/kstats_l2hmc.log" 2>&1
bash tools/kstats.sh "$o/bench_l2hmc_kernel_stats_whole_process.txt" python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > /dev/null 2>&1
KSTATS_MARKER=$M KSTATS_LAST=5 bash tools/kstats.sh "$o/bench_hmc_kernel_stats.txt" python bench.py --mode hmc --no-cpu-baseline --no-spot-check --no-u1 > "$o/kstats_hmc.log" 2>&1
KSTATS_MARKER=$M KSTATS_LAST=3 bash tools/kstats.sh "$o/bench_cfg5_shard_kernel_stats.txt" python bench.py --lattice 16 16 16 16 --beta 6.2 --steps 3 --warmup 2 --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/kstats_cfg5.log" 2>&1
head -30 "$o/bench_l2hmc_kernel_stats.txt"; head -14 "$o/bench_cfg5_shard_kernel_stats.txt"; head -8 "$o/bench_hmc_kernel_stats.txt"; grep -c . "$o/bench_l2hmc_kernel_stats_whole_process.txt"
