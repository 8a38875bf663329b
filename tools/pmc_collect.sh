#!/usr/bin/env bash
# HBM/fabric traffic of the SU(3) kernels at the bench.py shapes (cfg-4: 8^4 x 256 chains), as
# MI355X_MICROARCH.md prescribes: one rocprofv3 --pmc pass per counter (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), no trace domains beside --kernel-trace.  Run ON the GPU box:
#   bash tools/pmc_collect.sh rNN      -> gpurun_out/pmc_rNN/{FETCH_SIZE,WRITE_SIZE,...}/
#                                         profiles/rNN_pmc_counters.txt, profiles/pmc_traffic.json
set -u
tag="${1:-r03}"      # L2Q_KPROF_LATTICE="16 16 16 16" L2Q_KPROF_NB=256 for the cfg-5 shard
cd "$(dirname "$0")/.."
root="$PWD"
out="$root/gpurun_out/pmc_$tag"
mkdir -p "$out"
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"; do
  d="$out/$(echo "$ctr" | tr ' ' '_')"
  mkdir -p "$d"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d "$d" -o p --output-format csv \
      -- python "$root/${L2Q_KPROF_SCRIPT:-tools/kprof.py}" > "$d/stdout.log" 2>&1)
  echo "pass [$ctr]: rc=$? $(ls "$d" | tr '\n' ' ')"
done
# (L2Q_KPROF_SCRIPT=tools/kprof_train.py L2Q_PMC_JSON=profiles/pmc_traffic_train.json: the reverse-sweep kernels)
python tools/pmc_summary.py "$out/*/p_counter_collection.csv" "profiles/${tag}_pmc_counters.txt" \
    "${L2Q_PMC_JSON:-profiles/pmc_traffic.json}"
