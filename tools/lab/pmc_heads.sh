#!/usr/bin/env bash
# PMC comparison of the heads-kernel candidates (tools/lab/heads_lab.hip), one pass per counter group
cd "$(dirname "$0")/../.."
root="$PWD"; out="$root/gpurun_out/pmc_heads"; mkdir -p "$out"; export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1)); d="$out/g$i"; mkdir -p "$d"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d "$d" -o p --output-format csv -- "$root/tools/bin/heads_lab" 256 147456 pmc > "$d/stdout.log" 2>&1)
  echo "pass [$ctr]: rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/pmc_heads/g*/**/p_counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void l2q::', '')
        rows[k].setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
with open('gpurun_out/pmc_heads/summary.txt', 'w') as o:
    for k, d in rows.items():
        o.write(k + '\n')
        for c, v in sorted(d.items()):
            o.write(f'    {c:36s} {sum(v)/len(v):.4g}  (n={len(v)})\n')
print(open('gpurun_out/pmc_heads/summary.txt').read())
PY
