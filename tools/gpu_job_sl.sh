#!/usr/bin/env bash
# timing decomposition of the sliced heads kernel (A/B builds with L2Q_SL_SKIP)
cd "$(dirname "$0")/.."
python tools/time_heads_sliced.py
for k in "$@"; do BOTH=0 L2Q_LIB_NAME=libl2q_$k.so python tools/time_heads_sliced.py; done
