"""CPU oracle for the DDM hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a plain-PyTorch (CPU, fp32/fp64) restatement of the
reference's algorithm for the path named in BASELINE.json (UNet forward/backward +
analytic-schedule q_sample / loss / samplers).  It exists to CHECK the HIP product
path and to be timed as the ``cpu_baseline`` leg of bench.py.

Only ``tests/``, ``__graft_entry__.smoke()``, ``bench.py``'s cpu_baseline leg and
``tools/make_golden.py`` may import it.  The product package ``adm_amd`` never does.

Parity status: PINNED.  ``tools/make_golden.py`` (run in the build container, where
/root/reference is importable) checks every function here against the imported
reference on identical inputs and writes ``tests/golden/*.npz`` +
``tests/golden/oracle_vs_reference_report.json``; ``tests/test_oracle_golden.py``
re-checks the oracle against those committed vectors on every run.
"""
