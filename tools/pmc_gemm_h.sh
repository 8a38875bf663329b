#!/usr/bin/env bash
# PMC passes on the big half-precision Linear (8192 x 8192 x 51200 fp16, gemm_f16_dma.hip) and the conv
# kernels around it: HBM/fabric traffic, L2 hit rate, LDS bank conflicts.  One counter group per pass,
# no trace domains beside --kernel-trace.   bash tools/pmc_gemm_h.sh <tag>  (on the GPU box)
set -u
tag="${1:-r02}"
cd "$(dirname "$0")/.."
root="$PWD"
out="$root/gpurun_out/pmc_gemmh_$tag"
mkdir -p "$out"
export TMPDIR=/tmp
cat > /tmp/gemmh_probe.py <<PY
import sys, torch
sys.path.insert(0, '$root/l2hmc-qcd_amd')
from l2hmc import _ops as ops
torch.manual_seed(0)
m, n, k = 8192, 8192, 51200
a = torch.randn(m, k, device='cuda').half(); w = (torch.randn(n, k, device='cuda') / k ** 0.5).half()
b = torch.randn(n, device='cuda')
for _ in range(3):
    c = ops.gemm_h(a, w, b, act='leaky_relu', out_dtype=torch.float32)
torch.cuda.synchronize()
PY
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  d="$out/$(echo "$ctr" | tr ' ' '_')"
  mkdir -p "$d"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d "$d" -o p --output-format csv \
      -- python /tmp/gemmh_probe.py > "$d/stdout.log" 2>&1)
  echo "pass [$ctr]: rc=$?"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_h_dma' in r['Kernel_Name']:
            acc[r['Counter_Name']]['v'].append(float(r['Counter_Value']))
res = {k: sum(v['v']) / len(v['v']) for k, v in acc.items()}   # mean over the dispatches (one row per dispatch)
lines = ['# gemm_h_dma_kernel, 8192 x 8192 x 51200 fp16: PMC values per launch (mean of 3 launches)']
for k, v in sorted(res.items()):
    lines.append(f'{k:24s} {v:.4g}')
if 'FETCH_SIZE' in res:
    lines.append(f'read  {2 * res["FETCH_SIZE"] * 1024 / 1e9:.2f} GB (2 x FETCH_SIZE KiB, the gfx950 correction of MI355X_MICROARCH.md), write {res.get("WRITE_SIZE", 0) * 1024 / 1e9:.2f} GB;  operands 1.68 GB, output 0.27 GB')
if 'TCC_HIT_sum' in res:
    lines.append(f'L2 hit rate {res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"]):.3f}')
if 'SQ_LDS_BANK_CONFLICT' in res and res.get('SQ_LDS_IDX_ACTIVE'):
    lines.append(f'LDS bank-conflict cycles / active cycles {res["SQ_LDS_BANK_CONFLICT"] / res["SQ_LDS_IDX_ACTIVE"]:.3f}')
open(out + '/summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
