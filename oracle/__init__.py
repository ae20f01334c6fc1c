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
    Round 3: 32 poses incl. windows that leave the frame, the N == 2 quirk, use_normal=True and one amp=True pass of
    the reference predictors (make_golden_pipeline_wide.py -> pipeline_golden_wide.npz;
    tests/test_oracle_pipeline_golden_wide.py); the autocast restatement (``oracle.nets_amp``) against the reference
    modules under torch.autocast (make_golden_amp.py -> nets_amp_golden.npz; tests/test_oracle_amp_golden.py).
  * ``fpo_cluster_poses`` (mycpp/src/app/pybind_api.cpp:24-68): the C++ cannot be compiled here (Eigen / Boost absent);
    agrees with an independent float32 numpy restatement inside the reference's own make_rotation_grid on four symmetry
    sets (tests/golden/make_golden_geometry.py -> geometry_golden.npz; tests/test_geometry_vs_reference_golden.py).
  * what stays "parity unpinned": the INTERNALS of three third-party packages that are neither under
    /root/reference nor installed here -- nvdiffrast (rasterize / interpolate / texture), kornia 0.7.2
    (warp_perspective) and pytorch3d (so3_exp_map, rotation_6d_to_matrix).  ref_harness.py substitutes small
    stand-ins written from their published semantics; the reference ships no tests or golden
    vectors for them.  In particular the integer z-buffer is DEFINED by fp_oracle.c: the
    "bit-exact" gate is HIP kernel == this oracle, not == nvdiffrast's internal buffer, which is unobservable.
"""
