#!/usr/bin/env bash
cd "$(dirname "$0")/../.."
timeout 1500 python -m pytest tests/test_sizes_gpu.py tests/test_dynamics_gpu.py tests/test_kernels_gpu.py -q -x -k "cfg3 or half or bf16 or fp16 or heads_update_h or autocast" 2>&1 | tail -4
timeout 300 python tools/time_heads_h.py 2 2>&1 | grep "^\["
