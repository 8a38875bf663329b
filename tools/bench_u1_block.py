"""bench.py's untimed U(1) block alone: python tools/bench_u1_block.py [case tags ...] (default: cfg3_dense256_fp16)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tags = sys.argv[1:] or ['cfg3_dense256_fp16']
out = bench.secondary_u1(steps=3, only=tags)
for tag, rec in out.items():
    if isinstance(rec, dict):
        rec = {k: v for k, v in rec.items() if k != 'workload'}
    print(tag, json.dumps(rec))
