"""CPU oracle for the MPBP message-update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker.

What it is: a numpy (LAPACK ``gesdd`` through ``numpy.linalg.svd``) restatement
of the reference algorithm of stecrotti/MatrixProductBP.jl v0.9.0 for the path

    iterate! -> onebpiter! -> compute_prob_ys / cavity / op -> compress!
             -> f_bp_partial -> mpem2 -> compress!(:left) -> set_msg!

plus the observables read afterwards (beliefs, pair_beliefs,
bethe_free_energy) and the brute-force enumeration the reference's own tests
compare with.  Every function cites the reference ``file:line`` it follows
(paths relative to the reference checkout).

Third-party arithmetic that is NOT in the reference tree and is restated here
from its published behaviour: ``TensorTrains.jl`` v0.12 (``Project.toml:48``:
compress!, orthogonalize_left!/right!, SVDTrunc functors, normalize!,
normalize_eachmatrix!, normalization, marginals, accumulate_L/R, _compose) and
``CavityTools.jl`` 0.3/1 (``Project.toml:29``: cavity).

Parity pinning: the reference is Julia and cannot be run here
(no ``julia`` binary, nothing was denied - the toolchain is absent).  The
oracle is pinned by the reference's own known answers and properties:
  * the 7 belief vectors of ``test/sis_infinite_graph.jl:21-29`` (16 digits),
  * exactness on trees against brute-force enumeration, the property asserted
    by ``test/sis_small_tree.jl:46-50``, ``test/glauber_small_tree.jl:63-72``,
    ``test/glauber_pmJ_small_tree.jl``, ``test/sirs_small_tree.jl``,
  * "observe everything => -F_bethe == logprob(X)" (``test/sis_small_tree.jl:100-111``),
  * ``evaluate(mpem2(B)) == evaluate(B)`` (``test/mpems.jl:29-40``),
  * recursive == generic (exhaustive trace) == RecursiveTraceFactor updates
    (``test/sis_small_tree.jl:53-98``).
See ``tests/test_oracle_*.py``.
"""
