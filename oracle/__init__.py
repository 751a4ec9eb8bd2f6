"""CPU oracle for the DQ-VAE hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, from the mathematics, what the reference's CPU path computes for
every row of SURVEY.md section 8(a) that the HIP library implements.  It exists to check
the HIP kernels; it is never imported by the product package
(``dynamicvectorquantization_amd``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.

Pinning: every function here is checked against outputs captured from the real reference
(imported read-only in the build container by ``tools/gen_golden.py``; fixtures under
``tests/golden/``) by ``tests/test_oracle_golden.py``.  Rows that cannot be pinned are
listed in DESIGN.md ("parity unpinned": LPIPS with ImageNet VGG16 weights, device RNG).

Implementation language: numpy / torch-CPU (fp32, with fp64 where the reference's result
is only well defined as the mathematically exact answer -- the VQ argmin).
"""
