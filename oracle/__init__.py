"""CPU oracle for the KmerCountTable hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py may import
this package.  Nothing under ``oxli_amd/`` does.
"""
from .oracle import *  # noqa: F401,F403
