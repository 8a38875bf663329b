#!/usr/bin/env bash
cd "$(dirname "$0")/../.."
export L2Q_GS_MEMSET=1
MODE=own STEPS=10 timeout 300 python tools/graph_feed_probe.py 2>&1 | grep "per traj"
for c in su3_verbose; do CASE=$c timeout 300 python tools/graph_d2h_probe.py 2>&1 | grep "^\[" | tail -1; done
