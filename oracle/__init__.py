"""ORACLE — test infrastructure only.

CPU restatements of the reference hot path used as the *checker* by tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  Nothing in
``mp_former_amd`` may import from here.
"""
