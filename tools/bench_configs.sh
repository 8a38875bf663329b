#!/usr/bin/env bash
# Throughput of the BASELINE.json configs other than the bench.py workload (cfg-4), one line each.
# usage: bash tools/bench_configs.sh > profiles/rNN_u1_configs.txt   (on an MI355X)
set -u
cd "$(dirname "$0")/.."
run() { echo "## $*"; timeout 900 python tools/bench_u1.py "$@" 2>&1 | grep -E "forward|hmc" ; }
echo "# cfg-1: U(1) 8x8, beta 2, 128 chains, nleapfrog 4, fp32"
run --L 8 8 --nb 128 --nlf 4 --beta 2 --steps 20
run --L 8 8 --nb 128 --nlf 4 --beta 2 --steps 20 --conv
echo "# cfg-2: U(1) 16x16, beta 4, 2048 chains, nleapfrog 8, fp32"
run --steps 10
run --steps 5 --conv
echo "# cfg-3: U(1) 64x64, beta 6, 8192 chains, nleapfrog 8; fp32 vs fp16 / bf16 layers"
run --L 64 64 --nb 8192 --beta 6 --steps 3 --no-hmc --units 256 256
run --L 64 64 --nb 8192 --beta 6 --steps 3 --no-hmc --units 256 256 --precision fp16
run --L 64 64 --nb 8192 --beta 6 --steps 3 --no-hmc --units 256 256 --precision bf16
run --L 64 64 --nb 8192 --beta 6 --steps 2 --no-hmc --no-graph --conv
run --L 64 64 --nb 8192 --beta 6 --steps 2 --no-hmc --no-graph --conv --precision fp16
