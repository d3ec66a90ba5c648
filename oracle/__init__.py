"""CPU oracle for the TRON env path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this package.  The product never does.  Parity: pinned by tests/golden/*.npz
(generated from the reference by tests/golden/make_golden.py).
"""
from .tron_oracle import *  # noqa: F401,F403
