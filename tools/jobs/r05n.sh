#!/usr/bin/env bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05n
for pc in 0 1; do
  SKINNY_PC=$pc tools/kstats.sh gpurun_out/r05n/stats_pc$pc.txt python tools/time_gemm_h_input.py 4 8
  echo "== pc $pc"; grep -i "skinny\|splitk\|pack_w" gpurun_out/r05n/stats_pc$pc.txt | awk '{print substr($1,1,60), $2,$3,$4}'
  SKINNY_PC=$pc timeout 300 python tools/time_gemm_h_input.py 4 8 2>&1 | grep "^\["
done
python - <<'PY'
import sys, os, torch
sys.path.insert(0, 'l2hmc-qcd_amd')
from l2hmc import _ops as ops, native
torch.manual_seed(0)
m, n, k = 2048, 256, 4096
x = torch.rand(m, k, device='cuda') * 6 - 3; f = torch.randn(m, k, device='cuda')
wx = (torch.randn(n, k, device='cuda') / k ** 0.5).half(); wv = (torch.randn(n, k, device='cuda') / k ** 0.5).half()
wx2 = (torch.randn(n, 2 * k, device='cuda') / k ** 0.5).half(); b = torch.randn(n, device='cuda')
mask = (torch.rand(k, device='cuda') < 0.5).float()
out = {}
for pc in (0, 1):
    native.set_tuning('gemm_h_skinny_pc', pc)
    out[pc] = (ops.gemm_h(x, wx, b, a2=f, w2=wv, bias2=b, act='leaky_relu'), ops.gemm_h_u1x(x, mask, True, wx2, b, f, wv, b, 'leaky_relu'))
print('pc vs plain max diff', float((out[0][0].float() - out[1][0].float()).abs().max()), float((out[0][1].float() - out[1][1].float()).abs().max()))
PY
