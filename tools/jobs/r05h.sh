#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r05h; mkdir -p $o
export TMPDIR=/tmp
for m in own clone const eager eager_const; do MODE=$m timeout 300 python tools/graph_feed_probe.py 2>&1 | grep "per traj"; done | tee $o/feed.txt
for m in own clone; do
  MODE=$m STEPS=5 KSTATS_MARKER=su3_assemble_tah_kernel KSTATS_LAST=5 bash tools/kstats.sh $o/kstats_$m.txt python tools/graph_feed_probe.py > $o/kstats_$m.log 2>&1
  head -22 $o/kstats_$m.txt
done
