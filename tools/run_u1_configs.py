"""bench.py's secondary_u1 block alone (cfg-2 / cfg-3 as themselves): target of tools/kstats.sh"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (puts l2hmc-qcd_amd on sys.path)
res = bench.secondary_u1()
print(json.dumps({k: (v if isinstance(v, str) else {'ms_per_trajectory': v['ms_per_trajectory'], 'value': v['value']})
                  for k, v in res.items()}))
