"""CPU oracle for the DeepIM hot path -- TEST INFRASTRUCTURE ONLY.

Every function here is a CPU restatement (numpy / torch-CPU / plain C) of one
piece of wangg12/mx-DeepIM's refinement path and cites the reference file:line
it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; the product
(``mx-deepim_amd/``) never does and fails loudly without its HIP library.

Pinning status (see DESIGN.md "Oracle"):
  * se3 / flow / pose_error / min_rect restatements are PINNED by golden
    vectors generated from the importable reference modules
    (tests/golden/make_golden.py, run inside the build container).
  * zoom ops, FlowNet forward, rasteriser, Transform3D forward are restated
    from source; their MXNet / OpenGL primitives are absent here, so those
    parts are "parity unpinned" except where a reference self-check could be
    re-created (Transform3D vs RT_transform, ZoomTrans round trip).
"""
