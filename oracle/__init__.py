"""ORACLE — test infrastructure only.

CPU restatement (numpy) of the reference algorithms on the Cap2Det hot path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package; the
product (`cap2det_amd/`) never does and has no CPU fallback.  See the module docstrings for
what is pinned by the reference's own known-answer tests and what is PARITY UNPINNED.
"""
