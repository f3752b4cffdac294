"""TEST INFRASTRUCTURE ONLY — CPU restatement (oracle) of the Factorizer hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  Nothing under ``factorizer_amd/`` imports it.
"""
