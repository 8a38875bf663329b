#!/usr/bin/env bash
# Test scaffolding ONLY (never shipped, lives in /tmp): make the read-only reference
# importable.  It mkdirs at import, so it needs a writable copy; four of its import-time
# dependencies (mpi4py, enrich, hydra, omegaconf) are absent here and get 25 lines of stubs.
# See SURVEY.md Appendix B.
set -euo pipefail
mkdir -p /tmp/oracle/src /tmp/oracle_stubs/{mpi4py,enrich,hydra/core,omegaconf}
rm -rf /tmp/oracle/src/l2hmc
cp -r /root/reference/src/l2hmc /tmp/oracle/src/l2hmc && chmod -R u+w /tmp/oracle
cat > /tmp/oracle_stubs/mpi4py/__init__.py <<'PY'
class _Comm:
    def Get_rank(self): return 0
    def Get_size(self): return 1
    def bcast(self, x, root=0): return x
    def Barrier(self): pass
class MPI:
    COMM_WORLD = _Comm()
PY
touch /tmp/oracle_stubs/enrich/__init__.py /tmp/oracle_stubs/hydra/__init__.py /tmp/oracle_stubs/hydra/core/__init__.py
echo 'from rich.logging import RichHandler' > /tmp/oracle_stubs/enrich/handler.py
cat > /tmp/oracle_stubs/hydra/core/config_store.py <<'PY'
class ConfigStore:
    _i = None
    @classmethod
    def instance(cls):
        cls._i = cls._i or cls(); return cls._i
    def store(self, *a, **k): pass
PY
printf 'class DictConfig(dict): pass\nclass OmegaConf: pass\n' > /tmp/oracle_stubs/omegaconf/__init__.py
echo "reference importable with PYTHONPATH=/tmp/oracle_stubs:/tmp/oracle/src"
