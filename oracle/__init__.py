"""CPU oracle for the MISO hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``miso_amd/`` may import this package.  It is used by ``tests/``,
by ``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline`` leg as the
*checker* of the HIP path, never as the thing shipped or measured as product.
"""
