#!/usr/bin/env bash
# A/B of alternate libl2q builds on one box: tools/gpu_job_ab.sh <script> <lib names...>
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab
script="$1"; shift
for rep in 1 2; do
for n in "$@"; do
  echo "== $n (rep $rep)"; L2Q_LIB_NAME=libl2q_$n.so timeout 300 python3 $script 2>&1 | grep -v "^$" | tail -${TAILN:-14}
done; done 2>&1 | tee gpurun_out/ab/out.txt
