"""CPU oracle for the L2HMC leapfrog hot path -- TEST INFRASTRUCTURE ONLY.

A numpy restatement of the reference algorithm (saforem2/l2hmc-qcd, PyTorch path).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``l2hmc-qcd_amd/``) never does.

Parity pin: every function here is checked against golden vectors that were produced by
importing the real reference in the build container (``tests/golden/make_golden.py``);
see ``tests/test_oracle_golden.py``.
"""
