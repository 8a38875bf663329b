#!/usr/bin/env bash
# A/B of the plaquette-sharing force kernel: timing builds -DL2Q_PQ_EXP=bits against the shipping library (one box)
for lib in libl2q.so $(cd l2hmc-qcd_amd/l2hmc/_lib && ls libl2q_pq*.so 2>/dev/null); do echo "== $lib"; L2Q_LIB_NAME=$lib timeout 200 python tools/force_bench.py --plaq --quick 2>&1 | grep "force_tile=7\|tile=5 su3_force_link_kernel<0"; done
