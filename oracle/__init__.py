"""CPU oracle for the MM2D3D hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement (numpy / torch-CPU) of the
algorithms on the hot path named in SURVEY.md section 8.  It exists only as the
checker: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package ``mm2d3d_amd`` never imports
it and fails loudly when its HIP library is missing.

Parity pinning:
  * the 3D arithmetic of the reference lives in the un-vendored dependency
    ``sparseconvnet`` (facebookresearch/SparseConvNet @ dcf6a7ff, pinned in
    /root/reference/environment.yml:37) which is absent from the container and
    from /root/reference, and the reference holds no golden vectors for it:
    **the SparseConvNet boundary is "parity unpinned" by the reference**.  The
    oracle restates the published algorithm and is
    pinned instead by (1) dense equivalence against torch CPU
    ``F.conv3d / F.conv_transpose3d / F.batch_norm`` (tests/test_oracle_dense.py)
    and (2) the reference's own composition ``scn_unet.UNet`` executed over the
    oracle's primitives (tests/test_oracle_wiring.py, fixtures under
    tests/golden/).
  * leaf functions that ARE importable from /root/reference
    (``augment_and_scale_3d``, ``Loss``, ``Optimizer``) are pinned by golden
    vectors generated from the reference itself (tests/golden/make_golden.py).
"""
