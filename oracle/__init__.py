"""CPU oracle for the mici_amd hot path.  TEST INFRASTRUCTURE ONLY.

This package is a NumPy/SciPy restatement of the integrator hot path of the
reference (matt-graham/mici, ``src/mici/{integrators,solvers,systems,matrices}.py``).
It exists so that the HIP kernels can be checked for parity; it is *not* part of
the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product package ``mici_amd`` never
imports ``oracle`` and has no CPU fallback.

Pinning: the restatement is checked op-for-op against the imported reference in
the build container (``tools/gen_golden.py``), and against the committed
fixtures under ``tests/golden/`` everywhere else (``tests/test_oracle_golden.py``).
"""
