#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r05g; mkdir -p $o
export TMPDIR=/tmp
timeout 900 python tools/graph_vs_eager.py 2>&1 | grep -v amdgpu | tee $o/graph_vs_eager.txt
