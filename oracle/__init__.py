"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the geneo4PETSc hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.
The product (``geneo4petsc_amd`` + ``libgeneopc.so``) never imports or links this package.

Pinning status (see DESIGN.md "Oracle"):
  * decomposition / element weighting / local (Neumann) matrix assembly / RHS / converged
    solution are PINNED by the reference's own 84 golden logs ``tst/dummy/*.ref``
    (committed as data in ``tests/golden/dummy_refs.json``).
  * eigenvalues, dimE and Krylov iteration counts are NOT pinned by any reference test
    (``--shortRes`` suppresses them): for those rows parity is
    "unpinned by the reference"; they are pinned instead by ARPACK/SuperLU (scipy) which is
    the same ARPACK mode-3 shift-invert algorithm SLEPc drives at geneo.cpp:626-744.
"""
