"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the RealPDEBench FNO3d hot path.

Nothing in the product package (``realpdebench_amd``) may import from here.
Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.
"""
