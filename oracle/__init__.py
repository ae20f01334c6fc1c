"""CPU oracle for the render-and-compare hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``foundationpose_amd/`` may import this
package; it is used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` as the checker / CPU baseline.

Parity status:
  * network definitions (``oracle.nets``): PINNED against outputs of the reference's own
    ``learning/models/*.py`` (tests/golden/nets_golden.npz, made by tests/golden/make_golden.py).
  * image-space ops and control flow (``oracle.ops`` / fp_oracle.c / ``oracle.pipeline``): PINNED against golden
    vectors produced by the reference's own hot-path Python -- Utils.py, predict_pose_refine.py, predict_score.py,
    h5_dataset.py, pose_dataset.py imported from /root/reference and run on CPU (tests/golden/ref_harness.py,
    make_golden_pipeline.py -> tests/golden/pipeline_golden.npz; checked by tests/test_oracle_pipeline_golden.py).
  * what stays "parity unpinned": the INTERNALS of three third-party packages that are neither under
    /root/reference nor installed here -- nvdiffrast (rasterize / interpolate / texture), kornia 0.7.2
    (warp_perspective) and pytorch3d (so3_exp_map, rotation_6d_to_matrix).  ref_harness.py substitutes small
    stand-ins written from their published semantics; the reference ships no tests or golden
    vectors for them.  In particular the integer z-buffer is DEFINED by fp_oracle.c: the
    "bit-exact" gate is HIP kernel == this oracle, not == nvdiffrast's internal buffer, which is unobservable.
"""
