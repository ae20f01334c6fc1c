"""GPU parity tests: every C-ABI entry point of libfp_amd.so against the CPU oracle on the same seeded inputs.
Integer outputs (z-buffer, triangle ids, erosion) must be bit-exact; floating point within the stated tolerance."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def gmesh(scene, dev):
    from foundationpose_amd.Utils import make_mesh_tensors
    return make_mesh_tensors(scene["mesh"], device=dev)


@pytest.fixture(scope="module")
def frame(scene, dev):
    from oracle import ops as oo
    from oracle import pipeline as op
    d = op.preprocess_depth(scene["depth"])
    xyz = oo.depth2xyzmap(d, scene["K"], f64_internal=True)
    return dict(depth_f=d, xyz=xyz,
                rgb_t=torch.as_tensor(scene["rgb"], device=dev).float().contiguous(),
                depth_t=torch.as_tensor(d, device=dev), xyz_t=torch.as_tensor(xyz, device=dev))


def _t(x, dev):
    return torch.as_tensor(np.ascontiguousarray(x), device=dev)


# ------------------------------------------------------------------ frame ops
def test_erode_bit_exact_and_bilateral(scene, dev):
    from foundationpose_amd import ops
    from oracle import ops as oo
    d = scene["depth"].copy()
    d[:7, :9] = 0; d[100:104, 200:260] = 120.0; d[300, 300] = 0.3
    e_ref = oo.erode_depth(d)
    e = ops.erode_depth(_t(d, dev)).cpu().numpy()
    assert np.array_equal(e, e_ref)
    b_ref = oo.bilateral_filter_depth(e_ref)
    b = ops.bilateral_filter_depth(_t(e_ref, dev)).cpu().numpy()
    np.testing.assert_allclose(b, b_ref, rtol=0, atol=2e-6)  # expf differs by ulps between libm and the GPU
    assert np.array_equal(b == 0, b_ref == 0)


def test_depth_to_xyz_both_variants(scene, dev):
    from foundationpose_amd import ops
    from oracle import ops as oo
    d = scene["depth"]
    for f64 in (True, False):
        ref = oo.depth2xyzmap(d, scene["K"], zfar=1.0 if not f64 else np.inf, f64_internal=f64)
        out = ops.depth_to_xyz(_t(d, dev), scene["K"], zfar=1.0 if not f64 else float("inf"), f64_internal=f64).cpu().numpy()
        assert np.array_equal(out, ref)


def test_crop_windows_bit_exact(scene, dev):
    from foundationpose_amd import ops
    from oracle import ops as oo
    tf_ref, bb_ref = oo.crop_windows(scene["poses"], scene["K"], scene["diameter"], 1.2, (160, 160))
    tf, bb = ops.crop_windows(_t(scene["poses"], dev), scene["K"], scene["diameter"], 1.2, (160, 160))
    assert np.array_equal(tf.cpu().numpy(), tf_ref) and np.array_equal(bb.cpu().numpy(), bb_ref)


def test_pose_update(scene, dev):
    from foundationpose_amd import ops
    from oracle import ops as oo
    rng = np.random.default_rng(3)
    P = scene["poses"][:64]
    tr = rng.normal(size=(64, 3)).astype(np.float32)
    for rep, rd in (("axis_angle", 3), ("6d", 6)):
        ro = rng.normal(size=(64, rd)).astype(np.float32)
        for norm in (True, False):
            ref = oo.pose_update(tr, ro, P, rep, norm, (0.02, 0.02, 0.05), 0.349, scene["diameter"])
            out = ops.pose_update(_t(tr, dev), _t(ro, dev), _t(P, dev), rep, norm, (0.02, 0.02, 0.05), 0.349,
                                  scene["diameter"]).cpu().numpy()
            np.testing.assert_allclose(out, ref, rtol=0, atol=3e-6)  # tanhf / sinf / cosf ulps
    # any other trans_rep = the reference's plain `else` (predict_pose_refine.py:217-218): the raw output is the translation
    # (the oracle's update with normalize_xyz=True and a diameter of 2 multiplies by exactly 1)
    ro = rng.normal(size=(64, 3)).astype(np.float32)
    ref = oo.pose_update(tr, ro, P, "axis_angle", True, (0.02, 0.02, 0.05), 0.349, 2.0)
    out = ops.pose_update(_t(tr, dev), _t(ro, dev), _t(P, dev), "axis_angle", False, (0.02, 0.02, 0.05), 0.349, scene["diameter"],
                          trans_rep="raw_xyz").cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=0, atol=3e-6)
    assert np.array_equal(out[:, :3, 3], P[:, :3, 3] + tr)


def test_pose_update_deepim_and_delta_outputs(scene, dev):
    """trans_rep='deepim' (predict_pose_refine.py:201-215) and the optional outputs the reference keeps as
    last_trans_update / last_rot_update (:238-239): metric translation delta and applied 3x3 rotation"""
    import os
    from foundationpose_amd import ops
    from oracle import ops as oo
    rng = np.random.default_rng(4)
    P = scene["poses"][:64]
    tr = (rng.normal(size=(64, 3)) * np.array([0.05, 0.05, 0.02]) + np.array([0, 0, 1.0])).astype(np.float32)   # depth ratio ~ 1
    ro = rng.normal(size=(64, 3)).astype(np.float32)
    tf_ref, _ = oo.crop_windows(P, scene["K"], scene["diameter"], 1.2, (160, 160))
    for norm in (True, False):
        dt_ref = oo.deepim_trans_delta(tr, P, scene["K"], tf_ref, 160, norm, scene["diameter"])
        ref = oo.pose_update(dt_ref, ro, P, "axis_angle", True, (1, 1, 1), 0.349, 2.0)
        dt = torch.empty((64, 3), dtype=torch.float32, device=dev)
        dR = torch.empty((64, 3, 3), dtype=torch.float32, device=dev)
        out = ops.pose_update(_t(tr, dev), _t(ro, dev), _t(P, dev), "axis_angle", norm, (0.02, 0.02, 0.05), 0.349, scene["diameter"],
                              trans_delta_out=dt, rot_delta_out=dR, trans_rep="deepim", K=scene["K"], tf_to_crops=_t(tf_ref, dev),
                              input_w=160).cpu().numpy()
        np.testing.assert_allclose(dt.cpu().numpy(), dt_ref, rtol=0, atol=2e-5)     # closed-form vs numpy matrix inverses, f32
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5)
        np.testing.assert_allclose(out[:, :3, 3], P[:, :3, 3] + dt.cpu().numpy(), atol=1e-6)
        np.testing.assert_allclose(out[:, :3, :3], dR.cpu().numpy() @ P[:, :3, :3], atol=2e-6)
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_golden.npz")))
    Pg = g["poses_in"]
    tfg, _ = oo.crop_windows(Pg, scene["K"], scene["diameter"], 1.2, (160, 160))
    dt = torch.empty((3, 3), dtype=torch.float32, device=dev)
    ops.pose_update(_t(g["g4_raw_trans"], dev), _t(g["g4_raw_rot"], dev), _t(Pg, dev), "axis_angle", True, (0.02, 0.02, 0.05), 0.349,
                    scene["diameter"], trans_delta_out=dt, trans_rep="deepim", K=scene["K"], tf_to_crops=_t(tfg, dev), input_w=160)
    np.testing.assert_allclose(dt.cpu().numpy(), g["g6_deepim_trans_delta"], atol=5e-6, rtol=1e-5)   # the reference's own numbers


# ------------------------------------------------------------------ rasteriser
@pytest.mark.parametrize("textured", [True, False])
def test_render_crops_zbuffer_bit_exact(scene, dev, textured):
    from foundationpose_amd import ops
    from foundationpose_amd.mesh import make_can_mesh
    from foundationpose_amd.Utils import make_mesh_tensors
    from oracle import ops as oo
    from oracle import pipeline as op
    mesh = scene["mesh"] if textured else make_can_mesh(textured=False)
    mnp = op.mesh_tensors_np(mesh)
    gm = make_mesh_tensors(mesh, device=dev)
    P = scene["poses"][::7]  # 36 hypotheses across the grid
    tf, bb = oo.crop_windows(P, scene["K"], scene["diameter"], 1.2, (160, 160))
    want = ("A", "color", "depth", "xyz", "normal", "zbuf", "tri_id")
    ref = oo.render_crops(mnp, P, bb, scene["K"], 480, 640, (160, 160), scene["diameter"], 0.001, True, want=want)
    out = ops.render_crops(gm["_handle"], _t(P, dev), _t(bb, dev), scene["K"], 480, 640, (160, 160), scene["diameter"],
                           0.001, True, want=want)
    assert np.array_equal(out["tri_id"].cpu().numpy(), ref["tri_id"]), "triangle ids differ"
    assert np.array_equal(out["zbuf"].cpu().numpy().view(np.uint32), ref["zbuf"]), "integer z-buffer differs"
    assert (ref["tri_id"] >= 0).mean() > 0.15
    for k in ("A", "color", "depth", "xyz", "normal"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("crop_ratio", [1.2, 1.1])
def test_render_crops_zbuffer_bit_exact_all_252_poses(scene, dev, gmesh, crop_ratio):
    """round 6 (the round-5 verdict, weak 1d): the integer z-buffer and the triangle ids of EVERY hypothesis of the BASELINE
    configuration -- the 252-pose rotation grid, in the refiner's crop window (crop_ratio 1.2, predict_pose_refine.py:44-45) and in the
    scorer's (1.1, predict_score.py:74-75) -- bit for bit against the oracle, in one launch of 252 (the launch the bench times is 126)."""
    from foundationpose_amd import ops
    from oracle import ops as oo
    P = scene["poses"]
    assert P.shape[0] == 252
    tf, bb = oo.crop_windows(P, scene["K"], scene["diameter"], crop_ratio, (160, 160))
    ref = oo.render_crops(scene["mesh_np"], P, bb, scene["K"], 480, 640, (160, 160), scene["diameter"], 0.001, True, want=("zbuf", "tri_id"))
    out = ops.render_crops(gmesh["_handle"], _t(P, dev), _t(bb, dev), scene["K"], 480, 640, (160, 160), scene["diameter"], 0.001, True,
                           want=("zbuf", "tri_id"))
    tid, zb = out["tri_id"].cpu().numpy(), out["zbuf"].cpu().numpy().view(np.uint32)
    bad = [i for i in range(252) if not (np.array_equal(tid[i], ref["tri_id"][i]) and np.array_equal(zb[i], ref["zbuf"][i]))]
    assert not bad, f"hypotheses whose z-buffer / triangle ids differ from the oracle: {bad}"
    assert all((ref["tri_id"][i] >= 0).mean() > 0.1 for i in range(252))         # every hypothesis draws something
    # and the two halves the bench launches (sub-batches of 126) are the same bits as the launch of 252
    for a, b in ((0, 126), (126, 252)):
        o = ops.render_crops(gmesh["_handle"], _t(P[a:b], dev), _t(bb[a:b], dev), scene["K"], 480, 640, (160, 160), scene["diameter"], 0.001,
                             True, want=("zbuf", "tri_id"))
        assert torch.equal(o["tri_id"], out["tri_id"][a:b]) and torch.equal(o["zbuf"], out["zbuf"][a:b])


def test_render_crops_fp16_output_and_flags(scene, dev, gmesh):
    from foundationpose_amd import ops
    from oracle import ops as oo
    P = scene["poses"][:9]
    tf, bb = oo.crop_windows(P, scene["K"], scene["diameter"], 1.2, (160, 160))
    for norm, thr in ((True, 0.1), (False, 0.001)):
        ref = oo.render_crops(scene["mesh_np"], P, bb, scene["K"], 480, 640, (160, 160), scene["diameter"], thr, norm,
                              want=("A",))["A"]
        o16 = ops.render_crops(gmesh["_handle"], _t(P, dev), _t(bb, dev), scene["K"], 480, 640, (160, 160),
                               scene["diameter"], thr, norm, out_f16=True, want=("A",))["A"]
        assert o16.dtype == torch.float16
        o16 = o16.cpu().numpy()
        r16 = ref.astype(np.float16)  # same f32 values, rounded once to fp16
        assert (o16 != r16).mean() < 1e-5 and np.abs(o16.astype(np.float32) - r16.astype(np.float32)).max() <= 1e-3


def test_render_full_frame_and_ragged_sizes(scene, dev, gmesh):
    """no bbox (full 480x640 frame, strips of 10 rows) and a non-square crop with a ragged last strip"""
    from foundationpose_amd import ops
    from oracle import ops as oo
    T = scene["gt"][None].astype(np.float32)
    ref = oo.render_crops(scene["mesh_np"], T, None, scene["K"], 480, 640, (480, 640), normalize_xyz=False,
                          want=("tri_id", "zbuf", "depth"))
    out = ops.render_crops(gmesh["_handle"], _t(T, dev), None, scene["K"], 480, 640, (480, 640), normalize_xyz=False,
                           want=("tri_id", "zbuf", "depth"))
    assert np.array_equal(out["tri_id"].cpu().numpy(), ref["tri_id"])
    assert np.array_equal(out["zbuf"].cpu().numpy().view(np.uint32), ref["zbuf"])
    P = scene["poses"][:3]
    bb = np.array([[180, 60, 500, 350]] * 3, np.float32)
    ref = oo.render_crops(scene["mesh_np"], P, bb, scene["K"], 480, 640, (104, 88), want=("tri_id", "zbuf"))
    out = ops.render_crops(gmesh["_handle"], _t(P, dev), _t(bb, dev), scene["K"], 480, 640, (104, 88), want=("tri_id", "zbuf"))
    assert np.array_equal(out["tri_id"].cpu().numpy(), ref["tri_id"])
    assert np.array_equal(out["zbuf"].cpu().numpy().view(np.uint32), ref["zbuf"])


def test_render_degenerate_inputs(scene, dev, gmesh):
    """object behind the camera / far outside the crop => empty crops; N=0 is a no-op"""
    from foundationpose_amd import ops
    from oracle import ops as oo
    P = scene["poses"][:2].copy()
    P[0, 2, 3] = -0.5          # behind the camera: every vertex culled
    bb = np.array([[0, 0, 159, 159], [5000, 5000, 5159, 5159]], np.float32)  # second: window far off the object
    ref = oo.render_crops(scene["mesh_np"], P, bb, scene["K"], 480, 640, (160, 160), want=("tri_id", "A"))
    out = ops.render_crops(gmesh["_handle"], _t(P, dev), _t(bb, dev), scene["K"], 480, 640, (160, 160), want=("tri_id", "A"))
    assert (out["tri_id"].cpu().numpy() == -1).all() and (ref["tri_id"] == -1).all()
    assert np.array_equal(out["A"].cpu().numpy(), ref["A"])
    e = ops.render_crops(gmesh["_handle"], torch.empty((0, 4, 4), device=dev), torch.empty((0, 4), device=dev),
                         scene["K"], 480, 640, (160, 160), want=("A",))
    assert e["A"].shape[0] == 0


def test_render_large_mesh_workspace_path(scene, dev):
    """large mesh (25,920 triangles, 13k vertices): long per-strip lists; same bits as the oracle"""
    from foundationpose_amd import ops
    from foundationpose_amd.mesh import make_can_mesh
    from foundationpose_amd.Utils import make_mesh_tensors
    from oracle import ops as oo
    from oracle import pipeline as op
    mesh = make_can_mesh(n_ang=160, n_axial=80, textured=False)  # 12,963 vertices, 25,920 triangles
    assert ops._lib.lib().fp_workspace_bytes(4, len(mesh.vertices), len(mesh.faces), 160, 160) > 0
    gm = make_mesh_tensors(mesh, device=dev)
    P = scene["poses"][:4]
    tf, bb = oo.crop_windows(P, scene["K"], scene["diameter"], 1.2, (160, 160))
    ref = oo.render_crops(op.mesh_tensors_np(mesh), P, bb, scene["K"], 480, 640, (160, 160), want=("tri_id", "zbuf", "A"))
    out = ops.render_crops(gm["_handle"], _t(P, dev), _t(bb, dev), scene["K"], 480, 640, (160, 160), want=("tri_id", "zbuf", "A"))
    assert np.array_equal(out["tri_id"].cpu().numpy(), ref["tri_id"])
    assert np.array_equal(out["zbuf"].cpu().numpy().view(np.uint32), ref["zbuf"])
    np.testing.assert_allclose(out["A"].cpu().numpy(), ref["A"], atol=1e-5)


def test_nvdiffrast_render_shim(scene, dev, gmesh):
    from foundationpose_amd.Utils import nvdiffrast_render
    from oracle import ops as oo
    T = scene["gt"][None].astype(np.float32)
    extra = {}
    color, depth, normal = nvdiffrast_render(K=scene["K"], H=480, W=640, ob_in_cams=_t(T, dev), mesh_tensors=gmesh,
                                             use_light=True, extra=extra)
    ref = oo.render_crops(scene["mesh_np"], T, None, scene["K"], 480, 640, (480, 640), normalize_xyz=False,
                          want=("color", "depth", "normal", "xyz"))
    assert color.shape == (1, 480, 640, 3) and depth.shape == (1, 480, 640) and extra["xyz_map"].shape == (1, 480, 640, 3)
    np.testing.assert_allclose(color.cpu().numpy(), ref["color"], atol=1e-5)
    np.testing.assert_allclose(depth.cpu().numpy(), ref["depth"], atol=1e-6)
    np.testing.assert_allclose(normal.cpu().numpy(), ref["normal"], atol=1e-5)


# ------------------------------------------------------------------ observed crops
@pytest.mark.parametrize("mode", ["refine", "score"])
@pytest.mark.parametrize("normalize", [True, False])
def test_warp_crops(scene, dev, frame, mode, normalize):
    from foundationpose_amd import ops
    from oracle import ops as oo
    P = scene["poses"][::5].copy()
    P[0, :3, 3] = [0.17, 0.12, 0.6]      # window partly outside the frame: zero padding path
    P[1, :3, 3] = [-0.14, -0.1, 0.5]
    tf, _ = oo.crop_windows(P, scene["K"], scene["diameter"], 1.2, (160, 160))
    m = oo.MODE_REFINE if mode == "refine" else oo.MODE_SCORE
    ref = oo.warp_crops(scene["rgb"], frame["xyz"] if mode == "refine" else None, frame["depth_f"], tf, scene["K"], P,
                        scene["diameter"], m, normalize)
    out = ops.warp_crops(frame["rgb_t"], frame["xyz_t"] if mode == "refine" else None, frame["depth_t"], _t(tf, dev),
                         scene["K"], _t(P, dev), scene["diameter"], m, normalize).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-6)
    assert np.abs(ref[0, :3]).max() > 0 and (ref[0, :3] == 0).mean() > 0.05  # the padded region really is exercised
    o16 = ops.warp_crops(frame["rgb_t"], frame["xyz_t"] if mode == "refine" else None, frame["depth_t"], _t(tf, dev),
                         scene["K"], _t(P, dev), scene["diameter"], m, normalize, out_f16=True).cpu().numpy()
    np.testing.assert_allclose(o16.astype(np.float32), ref, atol=2e-3, rtol=1e-3)


# ------------------------------------------------------------------ predictors end to end (fp32 parity configuration)
def _geodesic(Ra, Rb):
    """rotation angle of Ra Rb^T via atan2(sin, cos): well conditioned near 0, unlike arccos of a float32 trace"""
    D = Ra.astype(np.float64) @ Rb.astype(np.float64).transpose(0, 2, 1)
    s = 0.5 * np.linalg.norm(np.stack([D[:, 2, 1] - D[:, 1, 2], D[:, 0, 2] - D[:, 2, 0], D[:, 1, 0] - D[:, 0, 1]], 1), axis=1)
    c = (np.trace(D, axis1=1, axis2=2) - 1) / 2
    return np.arctan2(s, c)


def test_refiner_fp32_matches_oracle(scene, dev, gmesh, frame):
    """north-star tolerance at BASELINE size: dR <= 1e-4 rad, dt <= 1e-4 m for every one of the 252 hypotheses in every one
    of the 5 refine iterations, calibrated stand-in weights (full-size updates: ~2 cm / 0.2-0.36 rad).

    Each iteration is compared from bit-identical inputs (the oracle's pose after the previous iteration): with
    stand-in (untrained) weights the render-and-compare map is chaotic -- the oracle itself turns a 1e-6 m input
    perturbation into 6e-3 rad after one iteration and 0.17 rad after two (coverage / nearest-neighbour flips feed a
    saturated head; measured in DESIGN.md "Parity") -- so a free-running chain compares two chaotic trajectories, not two
    implementations.  In this configuration (`amp=False` in the reference) the image-space ops run on libfp_amd.so and the
    networks on PyTorch-ROCm in fp32."""
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    from oracle import pipeline as op
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, seed=0)
    P0 = scene["poses"]  # 252 hypotheses
    trace = []
    op.refine_predict(cfg, sd, scene["rgb"], frame["depth_f"], scene["K"], P0, frame["xyz"], scene["mesh_np"],
                      scene["diameter"], iteration=5, trace=trace)
    pred = PoseRefinePredictor(cfg=cfg, state_dict=sd, device=dev, precision="fp32")
    start = P0
    worst = []
    for it in range(5):
        out, _ = pred.predict(scene["rgb"], frame["depth_t"], scene["K"], start, frame["xyz_t"], mesh=scene["mesh"],
                              mesh_tensors=gmesh, mesh_diameter=scene["diameter"], iteration=1)
        out = out.cpu().numpy()
        tgt = trace[it]["poses"]
        dR = _geodesic(out[:, :3, :3], tgt[:, :3, :3])
        dt = np.linalg.norm(out[:, :3, 3] - tgt[:, :3, 3], axis=1)
        uR = _geodesic(tgt[:, :3, :3], np.asarray(start)[:, :3, :3])
        worst.append((float(dR.max()), float(dt.max()), float(np.median(uR))))
        assert dR.max() <= 1e-4 and dt.max() <= 1e-4, (it, dR.max(), dt.max())
        assert np.median(uR) > 0.05                              # a full-size update
        np.testing.assert_allclose(pred.last_raw_output["trans"].cpu().numpy(), trace[it]["trans"], atol=2e-4)
        np.testing.assert_allclose(pred.last_raw_output["rot"].cpu().numpy(), trace[it]["rot"], atol=2e-4)
        # last_trans_update / last_rot_update as the reference keeps them (predict_pose_refine.py:238-239): metric
        # translation delta and the applied 3x3 rotation, i.e. pose_new = [[dR, 0], [0, 1]] applied as in Utils.py:848-855
        dT = pred.last_trans_update.cpu().numpy()
        dRm = pred.last_rot_update.cpu().numpy()
        np.testing.assert_allclose(out[:, :3, 3], np.asarray(start)[:, :3, 3] + dT, atol=1e-6)
        np.testing.assert_allclose(out[:, :3, :3], dRm @ np.asarray(start)[:, :3, :3], atol=1e-5)
        start = tgt
    # the free-running chain stays finite
    chain, _ = pred.predict(scene["rgb"], frame["depth_t"], scene["K"], P0[::16], frame["xyz_t"], mesh=scene["mesh"],
                            mesh_tensors=gmesh, mesh_diameter=scene["diameter"], iteration=3)
    assert torch.isfinite(chain).all()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_fp32_252x5.json"), "w") as f:
        json.dump(dict(per_iteration_max_dR_max_dt_median_update=worst), f)


def test_scorer_fp32_matches_oracle(scene, dev, gmesh, frame):
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_SCORE_CFG, random_state_dict
    from oracle import pipeline as op
    cfg = dict(DEFAULT_SCORE_CFG)
    sd = random_state_dict("score", cfg, seed=0)
    P0 = scene["poses"][::16]
    ref = op.score_predict(cfg, sd, scene["rgb"], frame["depth_f"], scene["K"], P0, scene["mesh_np"], scene["diameter"])
    pred = ScorePredictor(cfg=cfg, state_dict=sd, device=dev, precision="fp32")
    out, _ = pred.predict(scene["rgb"], frame["depth_t"], scene["K"], P0, mesh=scene["mesh"], mesh_tensors=gmesh,
                          mesh_diameter=scene["diameter"])
    out = out.cpu().numpy()
    assert np.abs(out - ref).max() < 1e-3 * max(1.0, np.abs(ref - 100).max())
    spread = ref.max() - ref.min()
    if spread > 1e-2:  # ranking is only meaningful when the logits are separated by more than the tolerance
        assert np.argmax(out) == np.argmax(ref)


def test_refiner_fp16_plan_close_to_fp32(scene, dev, gmesh, frame):
    """deployment configuration (fp16 + MFMA kernels) stays close to the fp32 parity configuration (one iteration)"""
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, seed=0)
    P0 = scene["poses"][::16]
    outs = {}
    for prec in ("fp32", "fp16"):
        pred = PoseRefinePredictor(cfg=cfg, state_dict=sd, device=dev, precision=prec)
        o, _ = pred.predict(scene["rgb"], frame["depth_t"], scene["K"], P0, frame["xyz_t"], mesh=scene["mesh"],
                            mesh_tensors=gmesh, mesh_diameter=scene["diameter"], iteration=1)
        outs[prec] = o.cpu().numpy()
    dt = np.linalg.norm(outs["fp16"][:, :3, 3] - outs["fp32"][:, :3, 3], axis=1)
    dR = _geodesic(outs["fp16"][:, :3, :3], outs["fp32"][:, :3, :3])
    assert dt.max() < 2e-3 and dR.max() < 2e-2, (dt.max(), dR.max())


def test_estimator_api_sequence(scene, dev):
    """replays run_demo.py's call sequence: register on frame 0, then track_one (estimater.py:159-268)"""
    from foundationpose_amd.estimater import FoundationPose
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    mesh = scene["mesh"]
    refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev)
    scorer = ScorePredictor(cfg=dict(DEFAULT_SCORE_CFG), state_dict=random_state_dict("score", seed=0), device=dev)
    est = FoundationPose(model_pts=mesh.vertices, model_normals=mesh.vertex_normals, mesh=mesh, scorer=scorer,
                         refiner=refiner, device=dev)
    assert est.rot_grid.shape == (252, 4, 4)
    with pytest.raises(RuntimeError):
        est.track_one(scene["rgb"], scene["depth"], scene["K"], iteration=2)
    pose = est.register(K=scene["K"], rgb=scene["rgb"], depth=scene["depth"], ob_mask=scene["mask"], iteration=2)
    assert pose.shape == (4, 4) and np.isfinite(pose).all()
    assert est.poses.shape == (252, 4, 4) and est.scores.shape == (252,)
    assert (est.scores[:-1] >= est.scores[1:]).all()
    p2 = est.track_one(scene["rgb"], scene["depth"], scene["K"], iteration=2)
    assert p2.shape == (4, 4) and np.isfinite(p2).all()
    empty = est.register(K=scene["K"], rgb=scene["rgb"], depth=scene["depth"], ob_mask=np.zeros_like(scene["mask"]))
    assert np.allclose(empty[:3, :3], np.eye(3))  # degenerate-mask early-out (estimater.py:185-189)


# ------------------------------------------------------------------ HIP path vs the reference's own Python (golden vectors)
def test_hip_network_inputs_match_reference_golden(scene, dev, gmesh, frame):
    """fp_crop_windows + fp_render_crops + fp_warp_crops against tests/golden/pipeline_golden.npz, which was produced
    by the reference's make_crop_data_batch / transform_batch (tolerances: tests/test_oracle_pipeline_golden.py)"""
    import os
    from foundationpose_amd import ops
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_golden.npz")))
    P = _t(g["poses_in"], dev)
    tf, bb = ops.crop_windows(_t(scene["poses"], dev), scene["K"], scene["diameter"], 1.2, (160, 160))
    assert np.array_equal(tf.cpu().numpy(), g["g1_tf_to_crops"])
    for mode, ratio, thr, keyA, keyB in ((ops.MODE_REFINE, 1.2, 0.001, "g3_refine_A_norm1", "g3_refine_B_norm1"),
                                         (ops.MODE_SCORE, 1.1, 0.1, "g3_score_A", "g3_score_B")):
        tf, bb = ops.crop_windows(P, scene["K"], scene["diameter"], ratio, (160, 160))
        A = ops.render_crops(gmesh["_handle"], P, bb, scene["K"], 480, 640, (160, 160), scene["diameter"], thr, True,
                             want=("A",))["A"].cpu().numpy()[:, :, ::2, ::2]
        B = ops.warp_crops(frame["rgb_t"], frame["xyz_t"] if mode == ops.MODE_REFINE else None, frame["depth_t"], tf,
                           scene["K"], P, scene["diameter"], mode, True).cpu().numpy()[:, :, ::2, ::2]
        dA = np.abs(A - g[keyA])
        assert dA[:, 3:].max() <= 5e-4 and (dA[:, :3] > 1e-3).mean() <= 1e-3 and dA[:, :3].max() <= 0.2
        np.testing.assert_allclose(B[:, :3], g[keyB][:, :3], rtol=0, atol=1e-4)
        diff = (B[:, 3:] != g[keyB][:, 3:]).any(1)
        assert diff.mean() < 2e-3 and diff[:, 1:, 1:].sum() == 0


def test_hip_network_inputs_and_predictors_match_wide_reference_golden(scene, dev, gmesh, frame):
    """the same against tests/golden/pipeline_golden_wide.npz: 32 poses incl. crop windows that leave the frame, the
    N == 2 quirk through the predictor, and the reference predictors' refined poses / scores (fp32, one iteration)"""
    from foundationpose_amd import ops
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor, make_crop_data_batch
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "pipeline_golden_wide.npz")))
    R, Cc = slice(1, None, 4), slice(2, None, 4)
    P = _t(g["w_poses_in"], dev)
    for mode, ratio, thr, keyA, keyB in ((ops.MODE_REFINE, 1.2, 0.001, "w_refine_A", "w_refine_B"),
                                         (ops.MODE_SCORE, 1.1, 0.1, "w_score_A", "w_score_B")):
        tf, bb = ops.crop_windows(P, scene["K"], scene["diameter"], ratio, (160, 160))
        A = ops.render_crops(gmesh["_handle"], P, bb, scene["K"], 480, 640, (160, 160), scene["diameter"], thr, True,
                             want=("A",))["A"].cpu().numpy()[:, :, R, Cc]
        B = ops.warp_crops(frame["rgb_t"], frame["xyz_t"] if mode == ops.MODE_REFINE else None, frame["depth_t"], tf,
                           scene["K"], P, scene["diameter"], mode, True).cpu().numpy()[:, :, R, Cc]
        dA = np.abs(A - g[keyA])
        assert dA[:, 3:].max() <= 5e-4 and (dA[:, :3] > 1e-3).mean() <= 2e-3 and dA[:, :3].max() <= 0.35
        assert np.array_equal(A.any(1), g[keyA].any(1))                      # same coverage, incl. the tie-rule pixels
        np.testing.assert_allclose(B[:, :3], g[keyB][:, :3], rtol=0, atol=1e-4)
        diff = (B[:, 3:] != g[keyB][:, 3:]).any(1)
        assert diff.mean() < 2e-3
    # N == 2 through make_crop_data_batch (predict_pose_refine.py:44-45)
    cfg = dict(DEFAULT_REFINE_CFG)
    b2 = make_crop_data_batch(cfg["input_resize"], g["pair_poses_in"], scene["mesh"], frame["rgb_t"], frame["depth_t"], scene["K"],
                              cfg["crop_ratio"], frame["xyz_t"], mesh_diameter=scene["diameter"], cfg=cfg, mesh_tensors=gmesh)
    A2 = b2.AB[:2].cpu().numpy()[:, :, ::2, ::2]
    dA = np.abs(A2 - g["pair_refine_A"])
    assert dA[:, 3:].max() <= 5e-4 and (dA[:, :3] > 1e-3).mean() <= 3e-3
    assert np.array_equal(b2.AB[2:].cpu().numpy()[:, 3:, ::2, ::2], g["pair_refine_B"][:, 3:])
    # predictors, fp32 configuration (amp off in the reference), against the reference's own outputs
    pred = PoseRefinePredictor(cfg=cfg, state_dict=random_state_dict("refine", cfg, 0), device=dev, precision="fp32")
    out, _ = pred.predict(scene["rgb"], frame["depth_t"], scene["K"], g["w_poses_in"], frame["xyz_t"], mesh=scene["mesh"],
                          mesh_tensors=gmesh, mesh_diameter=scene["diameter"], iteration=1)
    out = out.cpu().numpy()
    ref = g["w_refined_1it"]
    # band: the stand-in network's sensitivity to the texture-edge pixels where ref_harness' float64 rasteriser and the
    # float32 one differ (tests/test_oracle_pipeline_golden.py::test_one_pass_deviation_is_explained_by_the_rendered_inputs)
    assert np.abs(out[:, :3, 3] - ref[:, :3, 3]).max() <= 1.5e-3 and np.abs(out[:, :3, :3] - ref[:, :3, :3]).max() <= 8e-3
    out2, _ = pred.predict(scene["rgb"], frame["depth_t"], scene["K"], g["pair_poses_in"], frame["xyz_t"], mesh=scene["mesh"],
                           mesh_tensors=gmesh, mesh_diameter=scene["diameter"], iteration=1)
    assert np.abs(out2.cpu().numpy()[:, :3, 3] - g["pair_refined_1it"][:, :3, 3]).max() <= 1e-3
    scfg = dict(DEFAULT_SCORE_CFG)
    sc = ScorePredictor(cfg=scfg, state_dict=random_state_dict("score", scfg, 0), device=dev, precision="fp32")
    s, _ = sc.predict(scene["rgb"], frame["depth_t"], scene["K"], g["w_poses_in"], mesh=scene["mesh"], mesh_tensors=gmesh,
                      mesh_diameter=scene["diameter"])
    s = s.cpu().numpy()
    np.testing.assert_allclose(s, g["w_scores"], atol=0.6)
    assert np.argmax(s) == np.argmax(g["w_scores"])


def test_use_normal_true_is_the_reference_behaviour(scene, dev, gmesh, frame):
    """cfg use_normal=True (predict_pose_refine.py:50,58,75-76; predict_score.py:78): make_crop_data_batch adds normalAs /
    normalBs to the batch (golden minted by the reference's own function), predict() feeds the networks rgb + xyz only
    (:187-188) -- so the refined poses and scores are bit-identical with and without the flag; c_in != 6 is refused (the
    reference fails in its first conv)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_pipeline_wide import normal_map_for_tests
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor, make_crop_data_batch
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "pipeline_golden_wide.npz")))
    cfg = dict(DEFAULT_REFINE_CFG, use_normal=True)
    nm = normal_map_for_tests()
    b = make_crop_data_batch(cfg["input_resize"], g["un_poses_in"], scene["mesh"], frame["rgb_t"], frame["depth_t"], scene["K"],
                             cfg["crop_ratio"], frame["xyz_t"], normal_map=nm, mesh_diameter=scene["diameter"], cfg=cfg, mesh_tensors=gmesh)
    nA, nB = b.normalAs.cpu().numpy()[:, :, ::2, ::2], b.normalBs.cpu().numpy()[:, :, ::2, ::2]
    assert (np.abs(nB - g["un_normalBs"]) > 1e-6).any(1).mean() < 2e-3
    assert (np.abs(nA - g["un_normalAs"]) > 2e-3).any(1).mean() < 5e-3
    with pytest.raises(ValueError):
        make_crop_data_batch(cfg["input_resize"], g["un_poses_in"], scene["mesh"], frame["rgb_t"], frame["depth_t"], scene["K"],
                             cfg["crop_ratio"], frame["xyz_t"], normal_map=None, mesh_diameter=scene["diameter"], cfg=cfg, mesh_tensors=gmesh)
    sd = random_state_dict("refine", dict(DEFAULT_REFINE_CFG), 0)
    P = scene["poses"][:40]
    outs = []
    for c in (dict(DEFAULT_REFINE_CFG), cfg):
        pred = PoseRefinePredictor(cfg=c, state_dict=sd, device=dev)
        o, _ = pred.predict(scene["rgb"], frame["depth_t"], scene["K"], P, frame["xyz_t"], normal_map=nm, mesh=scene["mesh"],
                            mesh_tensors=gmesh, mesh_diameter=scene["diameter"], iteration=2)
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    scfg = dict(DEFAULT_SCORE_CFG, use_normal=True)
    s0, _ = ScorePredictor(cfg=dict(DEFAULT_SCORE_CFG), state_dict=random_state_dict("score", dict(DEFAULT_SCORE_CFG), 0), device=dev).predict(
        scene["rgb"], frame["depth_t"], scene["K"], P, mesh=scene["mesh"], mesh_tensors=gmesh, mesh_diameter=scene["diameter"])
    s1, _ = ScorePredictor(cfg=scfg, state_dict=random_state_dict("score", dict(DEFAULT_SCORE_CFG), 0), device=dev).predict(
        scene["rgb"], frame["depth_t"], scene["K"], P, normal_map=nm, mesh=scene["mesh"], mesh_tensors=gmesh, mesh_diameter=scene["diameter"])
    assert torch.equal(s0, s1)
    with pytest.raises(NotImplementedError):
        PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG, c_in=9), state_dict=sd, device=dev)


# ------------------------------------------------------------------ hipGraph-captured tracking
def test_graphed_tracker_replays_the_eager_result(scene, dev, gmesh, frame):
    """one captured graph per (frame size, N, iterations); replay == eager bit for bit, for changing inputs"""
    from foundationpose_amd.graphs import GraphedTracker
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev)
    P = scene["poses"][::32][:8]
    trk = GraphedTracker(refiner, gmesh, scene["diameter"], scene["K"], 480, 640, n_hyp=len(P), iteration=2, device=dev).capture()
    rng = np.random.default_rng(0)
    for k in range(3):
        rgb = np.clip(scene["rgb"].astype(np.float32) + rng.normal(0, 3 * k, scene["rgb"].shape), 0, 255).astype(np.float32)
        depth = (scene["depth"] + 0.002 * k).astype(np.float32)
        Pk = P.copy()
        Pk[:, :3, 3] += 0.003 * k
        eager = trk.step_eager(rgb, depth, Pk).clone()
        replay = trk.step(rgb, depth, Pk).clone()
        assert torch.equal(eager, replay), k
    # tracking mode: the previous output feeds the next frame without leaving the device
    a = trk.step(scene["rgb"].astype(np.float32), scene["depth"], P).clone()
    b = trk.step(scene["rgb"].astype(np.float32), scene["depth"]).clone()
    c = trk.step_eager(scene["rgb"].astype(np.float32), scene["depth"], a.cpu().numpy())
    assert torch.equal(b, c)


@pytest.mark.parametrize("n_hyp", [1, 64])
def test_frame_pipeline_returns_the_bits_of_the_unpipelined_loop(scene, dev, gmesh, n_hyp):
    """round 6, BASELINE configs[4] as its own workload (graphs.FramePipeline): upload + u8 -> f32 + depth filters + back-projection of
    frame f + 1 on an ingest stream under the refine loop of frame f, two staging slots handed over with events.  Twelve distinct
    frames (own noise, own depth, own hypotheses): every frame's poses equal GraphedTracker.step on the same host data bit for bit,
    with uploaded hypotheses and in tracking mode (the previous output is the next start: never uploaded)."""
    from foundationpose_amd import synthetic as syn
    from foundationpose_amd.graphs import FramePipeline, GraphedTracker
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.weights import CONTRACTION_HEAD_SCALE, DEFAULT_REFINE_CFG, random_state_dict
    refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0, head_scale=CONTRACTION_HEAD_SCALE),
                                  device=dev)
    F = 12
    rng = np.random.default_rng(5)
    rgb_h = torch.empty((F, 480, 640, 3), dtype=torch.uint8).pin_memory()
    depth_h = torch.empty((F, 480, 640), dtype=torch.float32).pin_memory()
    hyp_h = torch.empty((F, n_hyp, 4, 4), dtype=torch.float32).pin_memory()
    for f in range(F):
        rgb_h[f].copy_(torch.from_numpy(np.clip(scene["rgb"].astype(np.float32) + rng.normal(0, 4, scene["rgb"].shape), 0, 255).astype(np.uint8)))
        depth_h[f].copy_(torch.from_numpy((scene["depth"] + 0.0005 * f).astype(np.float32)))
        hyp_h[f].copy_(torch.from_numpy(syn.perturbed_poses(scene["gt"], n_hyp, seed=50 + f, max_trans=0.02, max_rot_deg=10.0).astype(np.float32)))
    trk = GraphedTracker(refiner, gmesh, scene["diameter"], scene["K"], 480, 640, n_hyp=n_hyp, iteration=2, device=dev).capture()
    ref = [trk.step(rgb_h[f].to(dev).float(), depth_h[f], hyp_h[f]).clone() for f in range(F)]
    trk._have_output = False
    ref_track = [trk.step(rgb_h[0].to(dev).float(), depth_h[0], hyp_h[0]).clone()]
    for f in range(1, F):
        ref_track.append(trk.step(rgb_h[f].to(dev).float(), depth_h[f]).clone())
    pipe = FramePipeline(trk)
    for mode, want in (("uploaded", ref), ("tracking", ref_track)):
        trk._have_output = False
        got = []
        pipe.submit(0, rgb_h[0], depth_h[0], hyp_h[0])
        for f in range(F):
            if f + 1 < F:
                pipe.submit((f + 1) % 2, rgb_h[f + 1], depth_h[f + 1], hyp_h[f + 1] if mode == "uploaded" else None)
            got.append(pipe.run(f % 2).clone())
        torch.cuda.synchronize()
        bad = [f for f in range(F) if not torch.equal(got[f], want[f])]
        assert not bad, (mode, bad)
    assert not torch.equal(ref[3], ref[4]) and not torch.equal(ref_track[3], ref_track[4])      # the frames differ


def test_sub_batches_on_concurrent_streams_change_nothing(scene, dev, gmesh, frame):
    """overlap.py: the refiner's / scorer's hypothesis sub-batches on two streams give the bits of one batch on one stream
    (same kernels and summation order per hypothesis), eagerly and inside a captured graph"""
    from foundationpose_amd.graphs import GraphedTracker
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    P = scene["poses"][:75]                      # odd count: parts of 38 + 37
    rgb, depth, xyz = frame["rgb_t"], frame["depth_t"], frame["xyz_t"]
    res = {}
    for ns in (1, 2, 3):
        refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev, n_streams=ns)
        scorer = ScorePredictor(cfg=dict(DEFAULT_SCORE_CFG), state_dict=random_state_dict("score", seed=0), device=dev, n_streams=ns)
        assert len(refiner.sub.parts(len(P))) == min(ns, 2) and len(refiner.sub.parts(63)) == 1
        from foundationpose_amd.overlap import side_streams_overlap
        if not side_streams_overlap(dev, 1):
            pytest.skip("the sub-batch stream does not run beside the main stream on this box")
        assert len(refiner.sub.parts(len(P), dev)) == min(ns, 2)
        p, _ = refiner.predict(rgb, depth, scene["K"], P, xyz, mesh=scene["mesh"], mesh_tensors=gmesh, mesh_diameter=scene["diameter"], iteration=3)
        s, _ = scorer.predict(rgb, depth, scene["K"], p, mesh=scene["mesh"], mesh_tensors=gmesh, mesh_diameter=scene["diameter"])
        res[ns] = (p.clone(), s.clone(), refiner.last_trans_update.clone(), refiner.last_rot_update.clone(),
                   {k: v.clone() for k, v in refiner.last_raw_output.items()})
        if ns == 2:
            trk = GraphedTracker(refiner, gmesh, scene["diameter"], scene["K"], 480, 640, n_hyp=len(P), iteration=3, device=dev).capture()
            assert len(trk.workspace) == 2
            g = trk.step(scene["rgb"].astype(np.float32), scene["depth"], P).clone()
            e = trk.step_eager(scene["rgb"].astype(np.float32), scene["depth"], P).clone()
            assert torch.equal(g, e)
    for ns in (2, 3):
        for a, b in zip(res[1][:4], res[ns][:4]):
            assert torch.equal(a, b), ns
        for k in res[1][4]:
            assert torch.equal(res[1][4][k], res[ns][4][k])


def test_graphed_predict_is_the_eager_predict(scene, dev, gmesh, frame):
    """PoseRefinePredictor.predict / ScorePredictor.predict with graph=True (one linear hipGraph per sub-batch, replayed on the
    sub-batch streams; predict_pose_refine.refine_graphed, predict_score._graphed_features) return the bits of the eager
    launches: first call (capture + replay), replay with new inputs, a second key, the shared-translation form of register(),
    and "auto" (eager at the first sighting of a key, captured at the second)"""
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    rgb, depth, xyz = frame["rgb_t"], frame["depth_t"], frame["xyz_t"]
    kw = dict(mesh=scene["mesh"], mesh_tensors=gmesh, mesh_diameter=scene["diameter"])
    refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev, graph=False)
    scorer = ScorePredictor(cfg=dict(DEFAULT_SCORE_CFG), state_dict=random_state_dict("score", seed=0), device=dev, graph=False)

    def both(P, it, graph, shared=None):
        p, _ = refiner.predict(rgb, depth, scene["K"], P, xyz, iteration=it, graph=graph, shared_translation=shared, **kw)
        out = [p.clone(), refiner.last_trans_update.clone(), refiner.last_rot_update.clone()] + \
              [v.clone() for _, v in sorted(refiner.last_raw_output.items())]
        s, _ = scorer.predict(rgb, depth, scene["K"], p, graph=graph, **kw)
        return out + [s.clone()]

    def same(a, b, what):
        for i, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), (what, i)
    P = scene["poses"][:75]
    Q = np.ascontiguousarray(scene["poses"][75:150])
    eager = both(P, 3, False)
    same(both(P, 3, True), eager, "capture + first replay")
    assert len(refiner._graphs.items) == 1 and len(scorer._graphs.items) == 1
    eq = both(Q, 3, False)
    same(both(Q, 3, True), eq, "replay with other poses")
    same(both(P, 3, True), eager, "replay with the first poses again")
    assert len(refiner._graphs.items) == 1
    rgb2 = (rgb * 0.9).contiguous()                       # another frame through the same graphs
    rgb, keep = rgb2, rgb
    e2 = both(P, 3, False)
    same(both(P, 3, True), e2, "replay with another frame")
    rgb = keep
    same(both(P[:40], 2, True), both(P[:40], 2, False), "a second key (one part, two iterations)")
    assert len(refiner._graphs.items) == 2
    # register(): every hypothesis at one translation, the first iteration shares the observed crop
    S = P.copy()
    S[:, :3, 3] = S[0, :3, 3]
    same(both(S, 2, True, shared=True), both(S, 2, False, shared=True), "shared translation")
    # "auto": the first sighting of a key runs eagerly, the second captures
    n0 = len(refiner._graphs.items)
    a1 = both(P[:50], 2, "auto")
    assert len(refiner._graphs.items) == n0
    a2 = both(P[:50], 2, "auto")
    assert len(refiner._graphs.items) == n0 + 1
    same(a1, a2, "auto: eager then graph")
    same(a1, both(P[:50], 2, False), "auto vs eager")


def test_linear_layernorm_is_the_two_kernel_path(dev):
    """fp_linear_layernorm_fwd (out_proj / linear2 + residual + LayerNorm in one launch) returns the bits of
    fp_igemm_f16_fwd followed by fp_layernorm_res_fwd: both residual forms, full and ragged row counts, one or both outputs"""
    from foundationpose_amd import ops
    from foundationpose_amd.engine import _HipLinear
    g = torch.Generator(device="cpu").manual_seed(11)
    w = (torch.randn((512, 512), generator=g) * 0.05)
    b = torch.randn((512,), generator=g) * 0.1
    lin = _HipLinear(w.to(dev), b.to(dev))
    wp = ops.PackedLinear512(lin.w)                       # what the fused kernel reads: the fragment-packed copy
    # the packing itself: lane l of k16-step q, channel tile i of channel group c holds W[64 c + 32 i + (l & 31)][16 q + 8 (l >> 5) ..+8]
    ref = lin.w.reshape(8, 2, 32, 32, 2, 8).permute(0, 3, 1, 4, 2, 5).contiguous()      # [c, q, i, l >> 5, l & 31, 8]
    assert torch.equal(wp.data.reshape(-1), ref.reshape(-1))
    with pytest.raises(Exception):
        ops.linear_layernorm_res(torch.zeros((4, 512), dtype=torch.float16, device=dev), lin.w, lin.b, torch.ones(512, device=dev),
                                 torch.zeros(512, device=dev), x32=torch.zeros((4, 512), device=dev))     # an unpacked weight is refused
    gamma = (1.0 + 0.1 * torch.randn((512,), generator=g)).to(dev)
    beta = (0.1 * torch.randn((512,), generator=g)).to(dev)
    pe = torch.randn((400, 512), generator=g).to(dev)
    for n_seq, S in ((126, 400), (3, 400), (1, 77), (1, 1), (2, 129)):
        M = n_seq * S
        x = torch.randn((n_seq, S, 512), generator=g).to(torch.float16).to(dev)
        tok = torch.randn((n_seq, S, 512), generator=g).to(torch.float16).to(dev)
        x32 = torch.randn((n_seq, S, 512), generator=g).to(dev)
        for kw in (dict(tok16=tok, pe=pe[:S].contiguous()), dict(x32=x32)):
            br = lin(x)
            want32, want16 = ops.layernorm_res(br, gamma, beta, 1e-5, **kw)
            got32, got16 = ops.linear_layernorm_res(x, wp, lin.b, gamma, beta, 1e-5, **kw)
            assert torch.equal(got32, want32) and torch.equal(got16, want16), (M, list(kw))
            only16 = ops.linear_layernorm_res(x, wp, lin.b, gamma, beta, 1e-5, want32=False, **kw)
            assert only16[0] is None and torch.equal(only16[1], want16)


def test_linear512_is_the_igemm_linear(dev):
    """fp_linear512_f16_fwd (the in_proj of the attention blocks on the row-owning tile: input tile fetched once for all column
    blocks, weights fragment-packed from L2) returns the bits of fp_igemm_f16_fwd: N = 512 / 1536 / 2048, full, ragged and tiny row
    counts, with and without ReLU and bias; nothing outside the output is written"""
    from foundationpose_amd import ops
    from foundationpose_amd.engine import _HipLinear
    g = torch.Generator(device="cpu").manual_seed(21)
    for N in (1536, 512, 2048, 3072):            # 3072 (round 6): the in_proj of both refiner heads in one launch, 12 bias pieces on 8 waves
        lin = _HipLinear((torch.randn((N, 512), generator=g) * 0.05).to(dev), (torch.randn((N,), generator=g) * 0.1).to(dev))
        wp = ops.PackedLinear512(lin.w)
        for M in (126 * 400, 3 * 400, 129, 1):
            x = torch.randn((M, 512), generator=g).to(torch.float16).to(dev)
            for relu in (False, True):
                want = lin(x, relu=relu)
                arena = torch.full((M * N + 2048,), -7.0, dtype=torch.float16, device=dev)
                got = ops.linear512(x, wp, lin.b, relu=relu, out=arena[1024:1024 + M * N].view(M, N))
                assert torch.equal(got, want), (N, M, relu)
                assert bool((arena[:1024] == -7).all()) and bool((arena[1024 + M * N:] == -7).all()), (N, M)
        nob = ops.linear512(x, wp, None)
        assert torch.equal(nob, _HipLinear(lin.w, torch.zeros(N, device=dev))(x))
    with pytest.raises(Exception):
        ops.linear512(torch.zeros((4, 256), dtype=torch.float16, device=dev), wp, None)


def test_merged_head_attention_is_the_two_separate_heads(scene, dev, gmesh, frame):
    """round 6 (the round-5 verdict's item 2): RefineNet's trans_head and rot_head as ONE in_proj launch (512 -> 3072) + ONE 8-head
    attention launch, out_proj + LayerNorm reading its head's half of the (N, 400, 1024) context through the row stride: (i) the strided
    fp_linear_layernorm_fwd returns the bits of the dense call on a copy of the column block; (ii) the whole plan with the switch on
    returns the bits of the plan with the switch off (raw network outputs of 126 and 40 hypotheses, both residual forms exercised)."""
    from foundationpose_amd import engine, ops
    from foundationpose_amd.refine_network import RefineNet
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    g = torch.Generator(device="cpu").manual_seed(33)
    wide = (torch.randn((7, 400, 1024), generator=g) * 0.5).half().to(dev)
    w = ops.PackedLinear512((torch.randn((512, 512), generator=g) * 0.05).half().to(dev))
    b = (torch.randn(512, generator=g) * 0.1).half().float().to(dev)
    gm, bt = (torch.rand(512, generator=g) + 0.5).to(dev), (torch.randn(512, generator=g) * 0.1).to(dev)
    tok = (torch.randn((7, 400, 512), generator=g)).half().to(dev)
    pe = torch.randn((400, 512), generator=g).to(dev)
    x32 = torch.randn((7, 400, 512), generator=g).to(dev)
    for c in (0, 512):
        blk = wide[..., c:c + 512]
        assert not blk.is_contiguous()
        for kw in (dict(tok16=tok, pe=pe), dict(x32=x32)):
            a32, a16 = ops.linear_layernorm_res(blk, w, b, gm, bt, 1e-5, **kw)
            d32, d16 = ops.linear_layernorm_res(blk.contiguous(), w, b, gm, bt, 1e-5, **kw)
            assert torch.equal(a32, d32) and torch.equal(a16, d16), c
    with pytest.raises(Exception):
        ops.linear_layernorm_res(wide[..., 4:516], w, b, gm, bt, 1e-5, x32=x32)          # misaligned column block
    with pytest.raises(Exception):
        ops.linear_layernorm_res(wide[:, ::2, :512], w, b, gm, bt, 1e-5, x32=x32[:, ::2])  # not a column block
    cfg = dict(DEFAULT_REFINE_CFG)
    net = RefineNet(cfg=cfg, c_in=6)
    net.load_state_dict(random_state_dict("refine", cfg, seed=0))
    plan = engine.RefinePlan(net, dev, precision="fp16")
    assert plan.qkv2_p is not None and plan.qkv2_p.out_features == 3072
    from oracle import pipeline as op
    for n in (126, 40):
        A, B, _, _ = op.refine_inputs(cfg, scene["poses"][:n], scene["mesh_np"], scene["rgb"], frame["xyz"], scene["K"], scene["diameter"])
        AB = torch.cat([torch.from_numpy(A), torch.from_numpy(B)]).half().to(dev)
        outs = {}
        for on in (True, False):
            with engine.overrides(MERGED_HEAD_QKV=on):
                with ops.KernelTimers() as kt:
                    o = plan(AB)
                calls = {k: v["calls"] for k, v in kt.summary().items()}
                assert calls["fp_attention_f16_fwd"] == (1 if on else 2) and calls["fp_linear512_f16_fwd"] == (1 if on else 2), (on, calls)
                outs[on] = plan(AB)
        for k in ("trans", "rot"):
            assert torch.equal(outs[True][k], outs[False][k]), (n, k)
            assert float(outs[True][k].abs().max()) > 0


def test_encoder_tail_is_the_two_fused_launches(dev):
    """round 6: fp_encoder_tail_mean_fwd (out_proj + x + sa + norm1 + linear1 + ReLU + linear2 + x + ff + norm2 + token mean in ONE
    launch, norm1's fp16 output staying in LDS) returns the bits of fp_linear_layernorm_fwd followed by fp_ffn_layernorm_mean_fwd:
    126 / 3 / 1 sequences of 400 tokens (ragged last tile), dense context and a column block of a two-head context, with and
    without biases; the workspace is checked"""
    from foundationpose_amd import ops
    import foundationpose_amd._lib as L
    g = torch.Generator(device="cpu").manual_seed(44)
    rnd = lambda *shape, s=1.0: (torch.randn(shape, generator=g) * s)
    W = [ops.PackedLinear512(rnd(512, 512, s=0.05).half().to(dev)) for _ in range(3)]
    Bv = [rnd(512, s=0.1).half().float().to(dev) for _ in range(3)]
    g1, b1n = (torch.rand(512, generator=g) + 0.5).to(dev), rnd(512, s=0.1).to(dev)
    g2, b2n = (torch.rand(512, generator=g) + 0.5).to(dev), rnd(512, s=0.1).to(dev)
    pe = rnd(400, 512).to(dev)
    for G_ in (126, 3, 1):
        wide = rnd(G_, 400, 1024, s=0.5).half().to(dev)
        tok = rnd(G_, 400, 512).half().to(dev)
        for ctx in (wide[..., :512].contiguous(), wide[..., 512:]):
            for bias in (True, False):
                bo, bb1, bb2 = (Bv if bias else (None, None, None))
                y32, y16 = ops.linear_layernorm_res(ctx, W[0], bo, g1, b1n, 1e-5, tok16=tok, pe=pe)
                want = ops.ffn_layernorm_mean(y16, W[1], bb1, W[2], bb2, y32, g2, b2n, 1e-5)
                got = ops.encoder_tail_mean(ctx, W[0], bo, tok, pe, g1, b1n, W[1], bb1, W[2], bb2, g2, b2n, 1e-5)
                assert torch.equal(got, want), (G_, ctx.is_contiguous(), bias)
                assert float(got.abs().max()) > 0 and bool(torch.isfinite(got).all())
    assert L.lib().fp_encoder_tail_workspace_bytes(126, 400) == 126 * 400 * 512 * 4 + (126 * 400 // 16 + 1) * 512 * 4
    with pytest.raises(L.FpAmdError):
        ops.encoder_tail_mean(ctx, W[0], None, tok, pe, g1, b1n, W[1], None, W[2], None, g2, b2n, 1e-5,
                              workspace=torch.empty(1024, dtype=torch.uint8, device=dev))


def test_ffn_layernorm_mean_is_the_three_kernel_path_up_to_summation_order(dev):
    """fp_ffn_layernorm_mean_fwd (linear1 + ReLU + linear2 + residual + norm2 + token mean in one launch) against 2 x fp_igemm_f16_fwd +
    fp_colmean_f16_fwd: the same rounding points, the token mean summed in another fixed fp32 order -> equal to ~1e-6 of the values; and
    the order does not depend on where in the batch a hypothesis sits: a slice of the batch returns the bits of the whole batch (what
    sub-batches on two streams, shards and the single batch rely on)"""
    from foundationpose_amd import ops
    from foundationpose_amd.engine import _HipLinear
    g = torch.Generator(device="cpu").manual_seed(12)
    l1 = _HipLinear((torch.randn((512, 512), generator=g) * 0.05).to(dev), (torch.randn((512,), generator=g) * 0.1).to(dev))
    l2 = _HipLinear((torch.randn((512, 512), generator=g) * 0.05).to(dev), (torch.randn((512,), generator=g) * 0.1).to(dev))
    p1, p2 = ops.PackedLinear512(l1.w), ops.PackedLinear512(l2.w)
    gamma = (1.0 + 0.1 * torch.randn((512,), generator=g)).to(dev)
    beta = (0.1 * torch.randn((512,), generator=g)).to(dev)
    for n, S in ((75, 400), (3, 400), (5, 144), (2, 16)):
        y16 = torch.randn((n, S, 512), generator=g).to(torch.float16).to(dev)
        x32 = torch.randn((n, S, 512), generator=g).to(dev)
        fused = ops.ffn_layernorm_mean(y16, p1, l1.b, p2, l2.b, x32, gamma, beta, 1e-5)
        three = ops.colmean_f16(l2(l1(y16, relu=True)), gamma, beta, 1e-5, resid32=x32)
        assert fused.shape == (n, 512) and torch.isfinite(fused).all()
        assert (fused - three).abs().max().item() <= 2e-6 * max(1.0, three.abs().max().item()), (n, S)
        for a, b in ((0, 1), (n // 2, n), (1, n - 1)):
            if b > a:
                part = ops.ffn_layernorm_mean(y16[a:b].contiguous(), p1, l1.b, p2, l2.b, x32[a:b].contiguous(), gamma, beta, 1e-5)
                assert torch.equal(part, fused[a:b]), (n, S, a, b)
    with pytest.raises(Exception):
        ops.ffn_layernorm_mean(torch.zeros((2, 130, 512), dtype=torch.float16, device=dev), p1, l1.b, p2, l2.b,
                               torch.zeros((2, 130, 512), device=dev), gamma, beta, 1e-5)


def test_replicate_channels(dev):
    """fp_replicate_rows_f16: one image's channel group copied into the same group of the following images, nothing else touched"""
    from foundationpose_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    for n, hp, wp, ct, c0, c1 in ((7, 42, 42, 256, 128, 256), (2, 5, 3, 64, 0, 8), (40, 9, 9, 512, 256, 320), (1, 4, 4, 16, 8, 16)):
        buf = torch.randn((n + 1, hp, wp, ct), generator=g).to(torch.float16).to(dev)
        want = buf.clone()
        want[1:n, :, :, c0:c1] = want[0:1, :, :, c0:c1]
        ops.replicate_channels(buf, n, c0, c1)
        assert torch.equal(buf, want), (n, hp, wp, ct, c0, c1)       # incl. the image after the last copy: untouched


def test_shared_observed_crop_changes_nothing(scene, dev, gmesh, frame):
    """register() starts every hypothesis at one translation, so in the first refine iteration all pairs have the same crop
    window and the same observed crop: the refiner warps it once per sub-batch and the fp16 plan's stem encodes it once
    (engine._HipEncoder shared_b).  Bit-identical to 75 separate copies, on one and on two streams, with the flag given, found
    out from host poses, and for the two-pose quirk; poses with different translations never take the path."""
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    rgb, depth, xyz = frame["rgb_t"], frame["depth_t"], frame["xyz_t"]
    kw = dict(mesh=scene["mesh"], mesh_tensors=gmesh, mesh_diameter=scene["diameter"])

    def run(refiner, P, it, **k):
        p, _ = refiner.predict(rgb, depth, scene["K"], P, xyz, iteration=it, **kw, **k)
        return (p.clone(), refiner.last_trans_update.clone(), refiner.last_rot_update.clone(),
                {n: v.clone() for n, v in refiner.last_raw_output.items()})

    def same(a, b):
        return all(torch.equal(x, y) for x, y in zip(a[:3], b[:3])) and all(torch.equal(a[3][n], b[3][n]) for n in a[3])

    for ns in (1, 2):
        refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev, n_streams=ns)
        calls = []
        plan = refiner.plan()
        enc_call = plan.enc.__call__
        plan.enc.__class__ = type("_CountingEncoder", (plan.enc.__class__,), {
            "__call__": lambda self, AB, slot=0, shared_b=False, small_calls=True: (calls.append((int(AB.shape[0]), bool(shared_b))), enc_call(AB, slot, shared_b, small_calls))[1]})
        for n, it in ((75, 1), (75, 3), (2, 2)):
            P = scene["poses"][:n]
            calls.clear()
            ref = run(refiner, P, it, shared_translation=False)
            assert not any(sh for _, sh in calls)
            calls.clear()
            auto = run(refiner, P, it)                                        # host poses: found out
            parts = refiner.sub.parts(n, dev)
            expect = [(b - a + 1, True) for a, b in parts if b - a > 1]
            assert [c for c in calls if c[1]] == expect, (calls, expect)      # iteration 0 only, one shared crop per part
            told = run(refiner, torch.as_tensor(P, device=dev), it, shared_translation=True)
            assert same(ref, auto) and same(ref, told), (ns, n, it)
        # different translations: host poses say so, and the flag is refused
        Q = scene["poses"][:8].copy()
        Q[3, 0, 3] += 1e-3
        calls.clear()
        run(refiner, Q, 1)
        assert not any(sh for _, sh in calls)
        with pytest.raises(ValueError):
            run(refiner, Q, 1, shared_translation=True)


def test_estimator_track_graph_matches_eager(scene, dev):
    from foundationpose_amd.estimater import FoundationPose
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    mesh = scene["mesh"]
    refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev)
    scorer = ScorePredictor(cfg=dict(DEFAULT_SCORE_CFG), state_dict=random_state_dict("score", seed=0), device=dev)
    out = {}
    for graph in (False, True):
        est = FoundationPose(model_pts=mesh.vertices, model_normals=mesh.vertex_normals, mesh=mesh, scorer=scorer,
                             refiner=refiner, device=dev, track_graph=graph)
        est.pose_last = torch.as_tensor(scene["gt"], device=dev, dtype=torch.float)
        seq = [est.track_one(scene["rgb"], scene["depth"], scene["K"], iteration=2) for _ in range(3)]
        # a register() with 252 hypotheses between tracked frames allocates the big activation set and grows the
        # rasteriser scratch; the captured graph owns its buffers, so replaying it afterwards is still the eager result
        seq.append(est.register(K=scene["K"], rgb=scene["rgb"], depth=scene["depth"], ob_mask=scene["mask"], iteration=1))
        seq += [est.track_one(scene["rgb"], scene["depth"], scene["K"], iteration=2) for _ in range(2)]
        out[graph] = np.stack(seq)
    assert np.array_equal(out[False], out[True])


# ------------------------------------------------------------------ edge cases of the predictors
@pytest.mark.parametrize("n", [1, 2, 5])
def test_refiner_small_batches_incl_two_pose_quirk(scene, dev, gmesh, frame, n):
    """N=1 (tracking), N=2 (the reference's transform_pts broadcasting quirk, SURVEY App. D.5: both hypotheses are
    rendered with [umin_0, vmin_0, umax_1, vmax_1]) and an odd N"""
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    from oracle import pipeline as op
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, seed=0)
    P0 = scene["poses"][[3, 77, 140, 201, 250][:n]].copy()
    if n == 2:
        P0[1, :3, 3] += [0.02, -0.01, 0.03]   # different windows, so that the quirk changes the render
    ref = op.refine_predict(cfg, sd, scene["rgb"], frame["depth_f"], scene["K"], P0, frame["xyz"], scene["mesh_np"],
                            scene["diameter"], iteration=1)
    pred = PoseRefinePredictor(cfg=cfg, state_dict=sd, device=dev, precision="fp32")
    out, _ = pred.predict(scene["rgb"], frame["depth_t"], scene["K"], P0, frame["xyz_t"], mesh=scene["mesh"],
                          mesh_tensors=gmesh, mesh_diameter=scene["diameter"], iteration=1)
    out = out.cpu().numpy()
    assert out.shape == (n, 4, 4)
    assert _geodesic(out[:, :3, :3], ref[:, :3, :3]).max() <= 1e-4
    assert np.linalg.norm(out[:, :3, 3] - ref[:, :3, 3], axis=1).max() <= 1e-4
    p16 = PoseRefinePredictor(cfg=cfg, state_dict=sd, device=dev, precision="fp16")
    o16, _ = p16.predict(scene["rgb"], frame["depth_t"], scene["K"], P0, frame["xyz_t"], mesh=scene["mesh"],
                         mesh_tensors=gmesh, mesh_diameter=scene["diameter"], iteration=2)
    assert o16.shape == (n, 4, 4) and torch.isfinite(o16).all()


@pytest.mark.parametrize("use_bn,rot_rep,normalize", [(False, "6d", False), (True, "6d", True), (False, "axis_angle", True)])
def test_refiner_config_variants_match_oracle(scene, dev, frame, use_bn, rot_rep, normalize):
    """use_BN / rot_rep / normalize_xyz are unknown until real checkpoints are available: all combinations
    of the code paths must match the oracle; vertex-colour mesh instead of a texture"""
    from foundationpose_amd.mesh import make_can_mesh
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.Utils import make_mesh_tensors
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    from oracle import pipeline as op
    cfg = dict(DEFAULT_REFINE_CFG, use_BN=use_bn, rot_rep=rot_rep, normalize_xyz=normalize)
    sd = random_state_dict("refine", cfg, seed=5)
    mesh = make_can_mesh(textured=False)
    gm = make_mesh_tensors(mesh, device=dev)
    P0 = scene["poses"][::40]
    ref = op.refine_predict(cfg, sd, scene["rgb"], frame["depth_f"], scene["K"], P0, frame["xyz"], op.mesh_tensors_np(mesh),
                            scene["diameter"], iteration=1)
    for prec, tolR, tolt in (("fp32", 1e-4, 1e-4), ("fp16", 3e-2, 3e-3)):
        pred = PoseRefinePredictor(cfg=cfg, state_dict=sd, device=dev, precision=prec)
        out, _ = pred.predict(scene["rgb"], frame["depth_t"], scene["K"], P0, frame["xyz_t"], mesh=mesh, mesh_tensors=gm,
                              mesh_diameter=scene["diameter"], iteration=1)
        out = out.cpu().numpy()
        assert _geodesic(out[:, :3, :3], ref[:, :3, :3]).max() <= tolR, prec
        assert np.linalg.norm(out[:, :3, 3] - ref[:, :3, 3], axis=1).max() <= tolt, prec


def test_scorer_single_hypothesis_and_fp16_ranking(scene, dev, gmesh, frame):
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_SCORE_CFG, random_state_dict
    from oracle import pipeline as op
    cfg = dict(DEFAULT_SCORE_CFG)
    sd = random_state_dict("score", cfg, seed=0)
    one, _ = ScorePredictor(cfg=cfg, state_dict=sd, device=dev, precision="fp32").predict(
        scene["rgb"], frame["depth_t"], scene["K"], scene["poses"][:1], mesh=scene["mesh"], mesh_tensors=gmesh,
        mesh_diameter=scene["diameter"])
    ref1 = op.score_predict(cfg, sd, scene["rgb"], frame["depth_f"], scene["K"], scene["poses"][:1], scene["mesh_np"], scene["diameter"])
    assert one.shape == (1,) and abs(float(one[0]) - float(ref1[0])) < 1e-3 * max(1.0, abs(float(ref1[0]) - 100))
    P0 = scene["poses"][::8]
    ref = op.score_predict(cfg, sd, scene["rgb"], frame["depth_f"], scene["K"], P0, scene["mesh_np"], scene["diameter"])
    s16, _ = ScorePredictor(cfg=cfg, state_dict=sd, device=dev, precision="fp16").predict(
        scene["rgb"], frame["depth_t"], scene["K"], P0, mesh=scene["mesh"], mesh_tensors=gmesh, mesh_diameter=scene["diameter"])
    s16 = s16.cpu().numpy()
    assert np.abs(s16 - ref).max() < 0.25 * max(1.0, ref.std())      # fp16 deployment vs fp32 oracle
    top = np.argsort(-ref)[:3]
    assert np.argmax(s16) in top                                       # the fp16 winner is among the oracle's top 3


# ------------------------------------------------------------------ demo driver on the on-disk formats
def test_run_demo_on_a_synthetic_sequence(tmp_path, dev):
    """scripts/run_demo.py: OBJ + PNG sequence written to disk, read back, register + track_one, poses on disk;
    graph replay of track_one gives the same files"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_demo", os.path.join(root, "scripts", "run_demo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for graph in (False, True):
        d = str(tmp_path / f"dbg{int(graph)}")
        mod.main(["--synthetic", "3", "--est_refine_iter", "2", "--track_refine_iter", "1", "--debug_dir", d] +
                 (["--track_graph"] if graph else []))
        files = sorted(os.listdir(os.path.join(d, "ob_in_cam")))
        assert files == ["0000000.txt", "0000001.txt", "0000002.txt"]
        out[graph] = np.stack([np.loadtxt(os.path.join(d, "ob_in_cam", f)) for f in files])
        assert np.isfinite(out[graph]).all() and np.allclose(out[graph][:, 3], [0, 0, 0, 1])
    assert np.array_equal(out[False], out[True])
    from foundationpose_amd.datareader import YcbineoatReader
    r = YcbineoatReader(str(tmp_path / "dbg0" / "synthetic_scene"))
    assert r.get_xyz_map(0).shape == (480, 640, 3)


def test_get_vis_canvases(scene, dev, gmesh, frame):
    """get_vis=True returns the debug canvases of predict_pose_refine.py:241-291 / predict_score.py:219-223 and leaves
    the numeric result untouched"""
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    P0 = scene["poses"][::64]
    n = len(P0)
    refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev)
    kw = dict(mesh=scene["mesh"], mesh_tensors=gmesh, mesh_diameter=scene["diameter"])
    o0, v0 = refiner.predict(scene["rgb"], frame["depth_t"], scene["K"], P0, frame["xyz_t"], iteration=1, **kw)
    o1, v1 = refiner.predict(scene["rgb"], frame["depth_t"], scene["K"], P0, frame["xyz_t"], iteration=1, get_vis=True, **kw)
    assert v0 is None and torch.equal(o0, o1)
    assert v1.dtype == np.uint8 and v1.ndim == 3 and v1.shape[0] > n * 160 and v1.shape[1] > 2 * 4 * 160
    assert v1.std() > 5      # rendered + observed content, not a blank sheet
    scorer = ScorePredictor(cfg=dict(DEFAULT_SCORE_CFG), state_dict=random_state_dict("score", seed=0), device=dev)
    s0, _ = scorer.predict(scene["rgb"], frame["depth_t"], scene["K"], o0, **kw)
    s1, v2 = scorer.predict(scene["rgb"], frame["depth_t"], scene["K"], o0, get_vis=True, **kw)
    assert torch.equal(s0, s1) and v2.dtype == np.uint8 and v2.shape[0] > n * 160 and v2.shape[1] > 4 * 160


def test_run_ycb_video_on_a_synthetic_bop_scene(tmp_path, dev):
    """scripts/run_ycb_video.py: BOP-layout scene + PLY/OBJ models on disk, reset_object + register per keyframe
    (run_ycb_video.py:43-130), yaml result file, ADD / ADD-S / AUC summary"""
    import importlib.util
    import os
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_ycb_video", os.path.join(root, "scripts", "run_ycb_video.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = str(tmp_path / "dbg")
    summary = mod.main(["--synthetic", "2", "--est_refine_iter", "1", "--debug_dir", d])
    assert summary["n"] == 2 and 0.0 <= summary["ADD_AUC"] <= 1.0 and 0.0 <= summary["ADDS_AUC"] <= 1.0
    assert summary["ADDS_mean_m"] <= summary["ADD_mean_m"] + 1e-9      # closest-point distance never exceeds the paired one
    res = yaml.safe_load(open(os.path.join(d, "ycbv_res.yml")))
    assert sorted(res[1].keys()) == ["000000", "000001"] and np.asarray(res[1]["000000"][1]).shape == (4, 4)
    from foundationpose_amd.datareader import YcbVideoReader
    r = YcbVideoReader(os.path.join(d, "synthetic_bop", "test", "000001"), models_dir=os.path.join(d, "synthetic_bop", "models"))
    assert r.get_xyz_map(0).shape == (480, 640, 3)


@pytest.mark.parametrize("B,S", [(3, 400), (2, 252), (1, 37), (5, 64), (2, 130), (1, 1)])
def test_attention_kernel_matches_fp32_softmax_attention(dev, B, S):
    """fp_attention_f16_fwd against the plain fp32 formula on the same fp16 operands (asymmetric random data, every
    tail case: partial key block, partial query tile, idle waves, one token)"""
    from foundationpose_amd import ops
    H, hd = 4, 128
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + S)
    qkv = (torch.randn((B, S, 3 * H * hd), generator=g) * 1.5).half().to(dev)
    out = ops.attention_f16(qkv, H)
    q, k, v = (qkv.float().reshape(B, S, 3, H, hd)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) / np.sqrt(hd), dim=-1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B, S, H * hd)
    assert out.shape == ref.shape and out.dtype == torch.float16
    err = (out.float() - ref).abs().max().item()
    assert err < 4e-3, err


def test_rasteriser_is_exact_next_to_a_gemm_on_another_stream(scene, dev, gmesh):
    """Regression for a gfx950 finding (csrc/Makefile, -fno-slp-vectorize): with packed-fp32 VALU instructions in its
    shading phase, k_raster returned wrong values in lanes 48-63 of a wave (16-pixel runs shaded at a neighbouring pixel
    centre) in ~90 % of the launches that overlapped an MFMA kernel of another stream -- a rocBLAS GEMM is enough.  The
    sub-batch streams of overlap.py make that overlap the normal case."""
    from foundationpose_amd import ops
    from foundationpose_amd.Utils import get_mesh_handle
    h = get_mesh_handle(gmesh)
    n = 38
    P = torch.as_tensor(scene["poses"][:n], device=dev)
    K, diam = scene["K"], scene["diameter"]
    _, bb = ops.crop_windows(P, K, diam, 1.2, (160, 160))
    A = torch.zeros((n, 6, 160, 160), dtype=torch.float16, device=dev)
    ws = torch.empty(max(16, ops.workspace_bytes(n, h.V, h.T, 160, 160)), dtype=torch.uint8, device=dev)
    x = torch.randn((14800, 512), device=dev, dtype=torch.float16)
    w = torch.randn((512, 512), device=dev, dtype=torch.float16)
    y = torch.empty((14800, 512), device=dev, dtype=torch.float16)
    side = torch.cuda.Stream(device=dev)

    def render():
        return ops.render_crops(h, P, bb, K, 480, 640, out_hw=(160, 160), mesh_diameter=diam, xyz_thr=0.001, normalize_xyz=True,
                                A_out=A, workspace=ws, want=("A", "zbuf", "tri_id"))
    base = {k: v.clone() for k, v in render().items()}
    torch.cuda.synchronize()
    bad = 0
    reps = 400
    for rep in range(reps):
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        r = render()                                   # the rasteriser first, the GEMM arrives while it runs
        with torch.cuda.stream(side):
            torch.matmul(x, w.t(), out=y)
        main.wait_stream(side)
        torch.cuda.synchronize()
        bad += int(any(not torch.equal(r[k], base[k]) for k in r))
    assert bad == 0, f"{bad} of {reps} overlapped launches differ from the launch that ran alone"
    # the run-time canary overlap.py runs before it lets sub-batches overlap agrees
    from foundationpose_amd import overlap
    assert overlap._exact_next_to_gemm(side, dev)


def test_eight_hypothesis_shards_with_the_real_predictors_rank_like_one_batch(scene, dev, gmesh, frame):
    """SURVEY 8(e) hypothesis-parallel on ONE GPU: the 8 ranks of a node run one after the other -- the REAL
    PoseRefinePredictor / ScorePredictor on shards of 32, ..., 28 hypotheses (shard_bounds(252, 8)), the real
    register_hypothesis_parallel / FeaturePoseExchange / all_gather_rows with only the collective replaced by a copy
    from the other ranks' send buffers -- and must rank the 252 hypotheses like the single 252-row batch.  Measured
    (profiles/r03_shards8_vs_batch.json): refined poses and scores of the shards are BIT-IDENTICAL to the batch's -- at
    these sizes a 28/32-row shard still selects the kernels and tiles of the 126-row sub-batches (fp_conv3x3_sw_applicable
    only excludes M < 2 BM), and a tile's fp32 summation order does not depend on where it sits in the grid -- so sharding
    changes nothing but the wall time.  Contraction-scaled heads as in the free-running chain test."""
    from foundationpose_amd import dist as fpd
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import CONTRACTION_HEAD_SCALE, DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    from amp_util import kendall_tau
    rcfg, scfg = dict(DEFAULT_REFINE_CFG), dict(DEFAULT_SCORE_CFG)
    refiner = PoseRefinePredictor(cfg=rcfg, state_dict=random_state_dict("refine", rcfg, 0, head_scale=CONTRACTION_HEAD_SCALE), device=dev)
    scorer = ScorePredictor(cfg=scfg, state_dict=random_state_dict("score", scfg, 0), device=dev)
    P0 = torch.as_tensor(scene["poses"], device=dev)
    N, G = P0.shape[0], 8
    args = (frame["rgb_t"], frame["depth_t"], scene["K"], P0, frame["xyz_t"])
    kw = dict(mesh=scene["mesh"], mesh_tensors=gmesh, mesh_diameter=scene["diameter"], iteration=5)
    poses1, scores1, order1 = fpd.register_hypothesis_parallel(refiner, scorer, *args, **kw)          # one rank, one batch
    # pass 1: every rank's send buffer (its collective sees only itself: the result of this pass is discarded)
    sends = {}

    def recording(rank):
        def collective(out, send):
            sends[rank] = send.clone()
            out.zero_()
            out[rank * send.shape[0]:(rank + 1) * send.shape[0]] = send
        return collective
    for r in range(G):
        fpd.register_hypothesis_parallel(refiner, scorer, *args, **kw, collective=recording(r), world_rank=(G, r))
    assert sorted(sends) == list(range(G)) and all(v.shape == (32, 528) for v in sends.values())      # the short shard is padded

    # pass 2: the all-gather delivers every rank's buffer; ranks 0 and 7 (the padded one) must agree with each other bit for bit
    def gathered(out, send):
        for r in range(G):
            out[r * 32:(r + 1) * 32] = sends[r]
    res = [fpd.register_hypothesis_parallel(refiner, scorer, *args, **kw, collective=gathered, world_rank=(G, r)) for r in (0, G - 1)]
    for a_, b_ in zip(res[0], res[1]):
        assert torch.equal(a_, b_)                      # replicated
    poses8, scores8, order8 = res[0]
    assert poses8.shape == (N, 4, 4) and scores8.shape == (N,)
    s1 = torch.empty(N, device=dev); s1[order1] = scores1
    s8 = torch.empty(N, device=dev); s8[order8] = scores8
    p1 = torch.empty((N, 4, 4), device=dev); p1[order1] = poses1
    p8 = torch.empty((N, 4, 4), device=dev); p8[order8] = poses8
    tau = kendall_tau(s1.cpu().numpy(), s8.cpu().numpy())
    dt = (p1[:, :3, 3] - p8[:, :3, 3]).norm(dim=1).max().item()
    dR = _geodesic(p1[:, :3, :3].cpu().numpy(), p8[:, :3, :3].cpu().numpy()).max()
    top1_rank = int((order1 == order8[0]).nonzero()[0, 0])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "shards8_vs_batch.json"), "w") as f:
        json.dump(dict(kendall_tau=tau, max_dt=dt, max_dR=float(dR), top1_of_sharded_run_has_rank_in_single_batch=top1_rank,
                       max_abs_score_diff=float((s1 - s8).abs().max())), f)
    assert dt <= 1e-4 and dR <= 1e-4, (dt, dR)           # refined poses of the shards = those of the batch within the north-star tolerance
    assert tau >= 0.98 and top1_rank <= 1, (tau, top1_rank)


_RCCL_CAPTURE = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from foundationpose_amd import dist as fpd
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
s = torch.randn(252, device=dev); p = torch.randn(252, 4, 4, device=dev)
fpd.gather_object_records(s, p); torch.cuda.synchronize()          # warm-up: communicator creation is not capturable
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        rec = fpd.gather_object_records(s * 2, p + 1)                # the collective is enqueued on the capturing (compute) stream
for k in range(3):
    s.copy_(torch.randn(252, device=dev)); p.copy_(torch.randn(252, 4, 4, device=dev))
    g.replay(); torch.cuda.synchronize()
    assert rec.shape == (1, 252, 17)
    assert torch.equal(rec[0, :, 0], s * 2) and torch.equal(rec[0, :, 1:].reshape(252, 4, 4), p + 1), k
dist.destroy_process_group()
print("RCCL_CAPTURE_OK")
"""


def test_rccl_all_gather_is_captured_with_the_compute_stream(dev):
    """the result exchange is ONE all-gather enqueued on the compute stream, so that a whole step can live in a
    hipGraph.  torch.distributed runs a synchronous collective (async_op=False -> AllgatherOptions.asyncOp = False) on the
    CURRENT stream in this PyTorch; here RCCL (world size 1, own process) is captured into a graph on a side stream together
    with the arithmetic around it and replayed with new inputs."""
    import socket
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", _RCCL_CAPTURE, ROOT, str(port)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RCCL_CAPTURE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


_TWO_RANKS = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
root, port, rank, world, fin, fout = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
sys.path.insert(0, root)
from foundationpose_amd import dist as fpd
from foundationpose_amd.Utils import make_mesh_tensors
from foundationpose_amd.mesh import make_can_mesh
from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
from foundationpose_amd.predict_score import ScorePredictor
from foundationpose_amd.weights import CONTRACTION_HEAD_SCALE, DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = port
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=world)
z = np.load(fin)
mesh = make_can_mesh(); gm = make_mesh_tensors(mesh, device=dev)
rcfg, scfg = dict(DEFAULT_REFINE_CFG), dict(DEFAULT_SCORE_CFG)
refiner = PoseRefinePredictor(cfg=rcfg, state_dict=random_state_dict("refine", rcfg, 0, head_scale=CONTRACTION_HEAD_SCALE), device=dev)
scorer = ScorePredictor(cfg=scfg, state_dict=random_state_dict("score", scfg, 0), device=dev)
t = lambda k: torch.as_tensor(z[k], device=dev)
poses, scores, order = fpd.register_hypothesis_parallel(refiner, scorer, t("rgb"), t("depth"), z["K"], t("poses"), t("xyz"), mesh=mesh,
                                                        mesh_tensors=gm, mesh_diameter=float(z["diameter"]), iteration=int(z["iteration"]))
rec = fpd.gather_object_records(scores + rank, poses)          # the object-parallel exchange, on device tensors over gloo
torch.cuda.synchronize()
np.savez(fout, poses=poses.cpu().numpy(), scores=scores.cpu().numpy(), order=order.cpu().numpy(), rec=rec.cpu().numpy())
dist.barrier(); dist.destroy_process_group()
print("RANK_OK", rank)
"""


@pytest.mark.parametrize("n", [252, 61])
def test_two_real_ranks_on_one_gpu_rank_like_one_batch(scene, dev, gmesh, frame, tmp_path, n):
    """SURVEY 8(e) with nothing emulated but the wire: TWO processes (torch.distributed rendezvous on 127.0.0.1, world size 2)
    share the box's one GPU, each with the real PoseRefinePredictor / ScorePredictor on its shard of the hypotheses (126 + 126, and
    the ragged 31 + 30), exchanging the [feature | pose] records with a real inter-process all-gather -- gloo through host memory,
    because RCCL refuses two ranks on one device (dist._all_gather); RCCL itself has only run at world size 1 here.  Both ranks
    must return the same bits, and those must rank the hypotheses like the single batch of this process."""
    import socket
    import subprocess
    import sys
    from foundationpose_amd import dist as fpd
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import CONTRACTION_HEAD_SCALE, DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    from amp_util import kendall_tau
    it = 3
    fin = str(tmp_path / "in.npz")
    np.savez(fin, rgb=frame["rgb_t"].cpu().numpy(), depth=frame["depth_f"], xyz=frame["xyz"], K=scene["K"], poses=scene["poses"][:n],
             diameter=scene["diameter"], iteration=it)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    outs = [str(tmp_path / f"out{r}.npz") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_RANKS, ROOT, str(port), str(r), "2", fin, outs[r]], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    # meanwhile: the single batch, in this process
    rcfg, scfg = dict(DEFAULT_REFINE_CFG), dict(DEFAULT_SCORE_CFG)
    refiner = PoseRefinePredictor(cfg=rcfg, state_dict=random_state_dict("refine", rcfg, 0, head_scale=CONTRACTION_HEAD_SCALE), device=dev)
    scorer = ScorePredictor(cfg=scfg, state_dict=random_state_dict("score", scfg, 0), device=dev)
    poses1, scores1, order1 = fpd.register_hypothesis_parallel(
        refiner, scorer, frame["rgb_t"], frame["depth_t"], scene["K"], torch.as_tensor(scene["poses"][:n], device=dev), frame["xyz_t"],
        mesh=scene["mesh"], mesh_tensors=gmesh, mesh_diameter=scene["diameter"], iteration=it)
    for r, p in enumerate(procs):
        so_, se_ = p.communicate(timeout=600)
        assert p.returncode == 0 and f"RANK_OK {r}" in so_, (r, so_[-2000:], se_[-4000:])
    z = [np.load(o) for o in outs]
    for k in ("poses", "scores", "order"):
        assert np.array_equal(z[0][k], z[1][k]), f"{k} differs between the two ranks"              # replicated
    for r in range(2):                                                                             # object-parallel records
        assert z[r]["rec"].shape == (2, n, 17)
        assert np.allclose(z[r]["rec"][1, :, 0] - 1, z[r]["rec"][0, :, 0], atol=1e-4)          # row g = rank g's record
        assert np.array_equal(z[r]["rec"][0, :, 1:].reshape(n, 4, 4), z[0]["poses"])
    s1 = np.empty(n, np.float32); s1[order1.cpu().numpy()] = scores1.cpu().numpy()
    s2 = np.empty(n, np.float32); s2[z[0]["order"]] = z[0]["scores"]
    p1 = np.empty((n, 4, 4), np.float32); p1[order1.cpu().numpy()] = poses1.cpu().numpy()
    p2 = np.empty((n, 4, 4), np.float32); p2[z[0]["order"]] = z[0]["poses"]
    dt = float(np.linalg.norm(p1[:, :3, 3] - p2[:, :3, 3], axis=1).max())
    dR = float(_geodesic(p1[:, :3, :3], p2[:, :3, :3]).max())
    tau = kendall_tau(s1, s2)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"two_ranks_one_gpu_n{n}.json"), "w") as f:
        json.dump(dict(n=n, iterations=it, poses_bit_identical_to_single_batch=bool(np.array_equal(p1, p2)),
                       scores_bit_identical_to_single_batch=bool(np.array_equal(s1, s2)), max_dt=dt, max_dR=dR, kendall_tau=tau,
                       max_abs_score_diff=float(np.abs(s1 - s2).max())), f)
    assert dt <= 1e-4 and dR <= 1e-4, (dt, dR)
    assert tau >= 0.98 and int(np.argmax(s1)) == int(np.argmax(s2)), tau


def test_network_kernels_write_only_their_outputs(dev):
    """outputs placed inside a poisoned arena: ragged GEMM / conv shapes (partial tiles), tokens + positional output,
    attention -- no byte outside the output tensors changes"""
    import ctypes as C
    from foundationpose_amd import ops, _lib
    G = 1 << 20

    def arena(nbytes):
        a = torch.full((G + nbytes + G,), 0x5A, dtype=torch.uint8, device=dev)
        return a, a[G:G + nbytes]

    def intact(a, nbytes):
        return bool((a[:G] == 0x5A).all()) and bool((a[G + nbytes:] == 0x5A).all())
    g = torch.Generator(device="cpu").manual_seed(0)
    Gm, Gi = ops.IgemmGeom.matrix, ops.IgemmGeom.image
    for M in (14800, 50400, 127, 129):
        for N in (512, 1536):
            x = (torch.randn((M, 512), generator=g) * 0.1).half().to(dev)
            w = (torch.randn((N, 512), generator=g) * 0.05).half().to(dev)
            a, yv = arena(M * N * 2)
            ops.igemm_f16(x, Gm(512), w, torch.zeros(N, device=dev), yv.view(torch.float16).reshape(M, N), Gm(N), M, N, 512, 1, relu=False)
            torch.cuda.synchronize()
            assert intact(a, M * N * 2), ("linear", M, N)
    for (Bn, H, Cin, Cout, stride) in ((37, 40, 128, 128, 1), (19, 40, 256, 256, 1), (37, 20, 512, 512, 1), (19, 40, 256, 512, 2), (21, 80, 64, 128, 2)):
        Ho = H // stride
        xin = torch.zeros((Bn, H + 2, H + 2, Cin), dtype=torch.float16, device=dev)
        xin[:, 1:-1, 1:-1] = (torch.randn((Bn, H, H, Cin), generator=g) * 0.1).half().to(dev)
        w = (torch.randn((Cout, 9 * Cin), generator=g) * 0.02).half().to(dev)
        nb = Bn * (Ho + 2) * (Ho + 2) * Cout * 2
        a, yv = arena(nb)
        yv.zero_()
        ops.igemm_f16(xin, Gi(Ho, Ho, 1, Cin, stride=stride, offset=0), w, torch.zeros(Cout, device=dev),
                      yv.view(torch.float16).reshape(Bn, Ho + 2, Ho + 2, Cout), Gi(Ho, Ho, 1, Cout), Bn * Ho * Ho, Cout, Cin, 9,
                      relu=True, conv_rounding=True)
        torch.cuda.synchronize()
        assert intact(a, nb), ("conv3x3", Bn, H, Cin, Cout, stride)
        if stride == 1 and Cout == 512:
            nb2 = Bn * Ho * Ho * Cout * 2
            a1, t1 = arena(nb2)
            a2, t2 = arena(nb2)
            ops.igemm_f16(xin, Gi(Ho, Ho, 1, Cin, stride=1, offset=0), w, torch.zeros(Cout, device=dev),
                          t1.view(torch.float16).reshape(Bn, Ho * Ho, Cout), Gi(Ho, Ho, 0, Cout), Bn * Ho * Ho, Cout, Cin, 9, relu=True,
                          conv_rounding=True, pe=torch.zeros((Ho * Ho, Cout), device=dev), y_pe=t2.view(torch.float16).reshape(Bn, Ho * Ho, Cout))
            torch.cuda.synchronize()
            assert intact(a1, nb2) and intact(a2, nb2), "tokens"
    for Bn in (37, 5):
        qkv = (torch.randn((Bn, 400, 1536), generator=g) * 0.3).half().to(dev)
        nb = Bn * 400 * 512 * 2
        a, ov = arena(nb)
        st = _lib.lib().fp_attention_f16_fwd(C.c_void_p(qkv.data_ptr()), C.c_void_p(ov.data_ptr()), Bn, 400, 4, 128, 0,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert st == 0 and intact(a, nb), ("attention", Bn)
    # patch-embed conv into the padded NHWC buffer
    for Bn in (5, 3):
        x = (torch.rand((Bn, 6, 160, 160), generator=g) - 0.5).half().to(dev)
        w = (torch.randn((64, 294), generator=g) * 0.05).half().to(dev)
        nb = Bn * 82 * 82 * 64 * 2
        a, yv = arena(nb)
        ops.conv7x7s2_bn_relu(x, w, torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros(64, device=dev),
                              yv.view(torch.float16).reshape(Bn, 82, 82, 64), 1)
        torch.cuda.synchronize()
        assert intact(a, nb), ("conv1", Bn)


# ------------------------------------------------------------------ round 5: a15 ranking and the reference's track_one, against the oracle
def test_register_ranking_and_track_one_match_the_oracle(scene, dev):
    """SURVEY a15 (predict_score.py:174-175 / estimater.py:173-182: scores -> argsort -> poses[ids], best pose through the centring
    transform) and the reference's tracking call (estimater.py:250-268: ONE hypothesis, 2 iterations, depth2xyzmap_batch), both through
    the estimator facade in the fp32 configuration, against the CPU oracle run on the same hypotheses: the whole ranking, not only
    "scores are sorted"; and track_one against the oracle's own two-iteration chain from the registered pose."""
    from foundationpose_amd.estimater import FoundationPose
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    from oracle import ops as oo
    from oracle import pipeline as op
    mesh = scene["mesh"]
    rcfg, scfg = dict(DEFAULT_REFINE_CFG), dict(DEFAULT_SCORE_CFG)
    rsd, ssd = random_state_dict("refine", rcfg, seed=0), random_state_dict("score", scfg, seed=0)
    refiner = PoseRefinePredictor(cfg=rcfg, state_dict=rsd, device=dev, precision="fp32")
    scorer = ScorePredictor(cfg=scfg, state_dict=ssd, device=dev, precision="fp32")
    est = FoundationPose(model_pts=mesh.vertices, model_normals=mesh.vertex_normals, mesh=mesh, scorer=scorer, refiner=refiner, device=dev)
    est.track_graph = False
    est.rot_grid = est.rot_grid[::8].contiguous()                  # 32 of the 252 hypotheses: the oracle finishes in seconds
    n = est.rot_grid.shape[0]
    K = scene["K"]
    best = est.register(K=K, rgb=scene["rgb"], depth=scene["depth"], ob_mask=scene["mask"], iteration=1)
    # the oracle on the SAME filtered depth and the SAME start hypotheses (the depth filters and the hypothesis generation are pinned
    # by their own tests; the bilateral filter's expf differs by ulps between libm and the GPU, and a 1e-6 difference of the start
    # translation is enough to flip coverage pixels of the stand-in network's inputs)
    from foundationpose_amd import ops
    depth_t = ops.bilateral_filter_depth(ops.erode_depth(torch.as_tensor(scene["depth"], device=dev, dtype=torch.float).contiguous(), radius=2), radius=2)
    d = depth_t.cpu().numpy()
    start = est.generate_random_pose_hypo(K=K, rgb=scene["rgb"], depth=depth_t,
                                          mask=torch.as_tensor(scene["mask"], device=dev) > 0).cpu().numpy()
    xyz = oo.depth2xyzmap(d, K, f64_internal=True)
    diam = float(est.diameter)
    mnp = op.mesh_tensors_np(est.mesh)                             # the estimator's (centred) mesh
    p_ref = op.refine_predict(rcfg, rsd, scene["rgb"], d, K, start, xyz, mnp, diam, iteration=1)
    # the refined poses register() ranked, before ranking: the same predict call once more (same kernels, same bits)
    xyz_t = ops.depth_to_xyz(depth_t, K, zfar=float("inf"), f64_internal=True)
    p_hip, _ = refiner.predict(mesh=est.mesh, mesh_tensors=est.mesh_tensors, rgb=scene["rgb"], depth=depth_t, K=K,
                               ob_in_cams=torch.as_tensor(start, device=dev), xyz_map=xyz_t, mesh_diameter=diam, iteration=1,
                               shared_translation=True)
    p_hip = p_hip.cpu().numpy()
    assert _geodesic(p_hip[:, :3, :3], p_ref[:, :3, :3]).max() <= 1e-4 and np.linalg.norm(p_hip[:, :3, 3] - p_ref[:, :3, 3], axis=1).max() <= 1e-4
    # the oracle scores THOSE poses (teacher forced: the stand-in scorer turns a 4e-6 pose difference into score differences of
    # up to 1.6 through coverage / nearest-neighbour flips of its inputs, measured; the ranking logic is what is under test here)
    s_ref = op.score_predict(scfg, ssd, scene["rgb"], d, K, p_hip, mnp, diam)
    ids = np.argsort(-s_ref, kind="stable")
    sc, po = est.scores.cpu().numpy(), est.poses.cpu().numpy()
    assert sc.shape == (n,) and po.shape == (n, 4, 4)
    tol = 1e-3 * max(1.0, np.abs(s_ref - 100).max())
    np.testing.assert_allclose(sc, s_ref[ids], atol=tol)           # the sorted scores ARE the oracle's sorted scores
    assert (sc[:-1] >= sc[1:]).all()
    # the ranking: position i holds oracle hypothesis ids[i] -- the very pose, bit for bit -- unless it sits in a run of near-ties
    gaps = -np.diff(s_ref[ids])
    for i in range(n):
        tie = (i > 0 and gaps[i - 1] <= 2 * tol) or (i < n - 1 and gaps[i] <= 2 * tol)
        assert tie or np.array_equal(po[i], p_hip[ids[i]]), (i, int(ids[i]))
    assert (gaps > 2 * tol).sum() >= n // 2, "the stand-in scores are too close together for a ranking test"
    assert int(est.best_id) == int(ids[0]) or gaps[0] <= 2 * tol
    assert sorted(map(bytes, po)) == sorted(map(bytes, p_hip))     # a permutation of the refined poses: nothing lost, nothing doubled
    np.testing.assert_allclose(best, po[0] @ est.get_tf_to_centered_mesh().cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(est.pose_last.cpu().numpy(), po[0], atol=0)
    # ---- track_one: one hypothesis, two iterations, the batch variant of the back-projection; teacher-forced per iteration like
    # test_refiner_fp32_matches_oracle (the stand-in map amplifies a last-bit difference of iteration 1 in iteration 2)
    xyz_b = oo.depth2xyzmap(d, K, f64_internal=False)
    trace = []
    op.refine_predict(rcfg, rsd, scene["rgb"], d, K, po[:1], xyz_b, mnp, diam, iteration=2, trace=trace)
    est.pose_last = torch.as_tensor(po[0], device=dev)
    t1 = est.track_one(scene["rgb"], scene["depth"], K, iteration=1)
    got1 = est.pose_last.cpu().numpy().reshape(4, 4)
    assert _geodesic(got1[None, :3, :3], trace[0]["poses"][:, :3, :3])[0] <= 1e-4 and np.linalg.norm(got1[:3, 3] - trace[0]["poses"][0, :3, 3]) <= 1e-4
    np.testing.assert_allclose(t1, got1 @ est.get_tf_to_centered_mesh().cpu().numpy(), atol=1e-6)
    est.pose_last = torch.as_tensor(trace[0]["poses"][0], device=dev)         # iteration 2 from the oracle's pose after iteration 1
    est.track_one(scene["rgb"], scene["depth"], K, iteration=1)
    got2 = est.pose_last.cpu().numpy().reshape(4, 4)
    assert _geodesic(got2[None, :3, :3], trace[1]["poses"][:, :3, :3])[0] <= 1e-4 and np.linalg.norm(got2[:3, 3] - trace[1]["poses"][0, :3, 3]) <= 1e-4
    # and the two-iteration call is the chain of two one-iteration calls (device-resident loop == host-chained calls, bit for bit)
    est.pose_last = torch.as_tensor(po[0], device=dev)
    est.track_one(scene["rgb"], scene["depth"], K, iteration=2)
    two = est.pose_last.clone()
    est.pose_last = torch.as_tensor(po[0], device=dev)
    est.track_one(scene["rgb"], scene["depth"], K, iteration=1)
    est.track_one(scene["rgb"], scene["depth"], K, iteration=1)
    assert torch.equal(two, est.pose_last)


def test_captured_renders_without_a_workspace_do_not_share_scratch(scene, dev, gmesh):
    """the advisor's round-4 finding: two graphs captured one after the other (torch.cuda.graph uses ONE capture stream) baked in the
    same stream-keyed default scratch; replayed on different streams they raced on it.  Now a capture without a caller-owned
    workspace allocates its scratch inside the capture (the graph's private pool): two such graphs replayed concurrently, many times,
    return the bits of the eager launches."""
    from foundationpose_amd import ops
    from foundationpose_amd.Utils import get_mesh_handle
    h = get_mesh_handle(gmesh)
    K, diam = scene["K"], scene["diameter"]
    sets = []
    for a in (0, 40):
        P = torch.as_tensor(scene["poses"][a:a + 40], device=dev)
        _, bb = ops.crop_windows(P, K, diam, 1.2, (160, 160))
        A = torch.zeros((40, 6, 160, 160), dtype=torch.float16, device=dev)
        sets.append((P, bb, A))

    def render(i):
        P, bb, A = sets[i]
        return ops.render_crops(h, P, bb, K, 480, 640, out_hw=(160, 160), mesh_diameter=diam, xyz_thr=0.001, normalize_xyz=True,
                                A_out=A, want=("A",))
    base = []
    for i in range(2):
        render(i)
        base.append(sets[i][2].clone())
    torch.cuda.synchronize()
    graphs = []
    for i in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            render(i)                                 # no workspace argument
        graphs.append(g)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    bad = 0
    for rep in range(60):
        for i in range(2):
            sets[i][2].zero_()
        torch.cuda.synchronize()
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                graphs[i].replay()
        torch.cuda.synchronize()
        bad += int(any(not torch.equal(sets[i][2], base[i]) for i in range(2)))
    assert bad == 0, f"{bad} of 60 concurrent replays differ from the eager launches"


@pytest.mark.parametrize("n", [1, 3, 12])
def test_small_calls_two_stream_heads_and_splitk_change_nothing_but_the_summation_order(scene, dev, gmesh, frame, n):
    """Round 5, the small-call path (the reference's track_one is ONE hypothesis): (i) the rotation head on the side stream beside
    the translation head returns the bits of the one-stream launch order, eagerly and as a captured graph; (ii) split-K convolutions
    change the fp32 summation order only: one teacher-forced iteration stays within the fp16 policy's noise of the plain kernels."""
    from foundationpose_amd import engine
    from foundationpose_amd.graphs import GraphedTracker
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.weights import CONTRACTION_HEAD_SCALE, DEFAULT_REFINE_CFG, random_state_dict
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, seed=0, head_scale=CONTRACTION_HEAD_SCALE)
    P0 = torch.as_tensor(scene["poses"][[3, 77, 140, 201, 250, 17, 33, 90, 111, 160, 222, 5][:n]], device=dev)
    kw = dict(mesh=scene["mesh"], mesh_tensors=gmesh, mesh_diameter=scene["diameter"])
    from foundationpose_amd import ops
    outs, splitk_launches = {}, {}
    for name, heads, sk in (("product", 12, 12), ("one_stream", 0, 12), ("plain_convs", 12, 0)):
        with engine.overrides(HEADS_TWO_STREAMS_MAX_HYPS=heads, SPLITK_MAX_HYPS=sk):
            pred = PoseRefinePredictor(cfg=cfg, state_dict=sd, device=dev, precision="fp16", graph=False)
            o, _ = pred.predict(scene["rgb"], frame["depth_t"], scene["K"], P0, frame["xyz_t"], iteration=2, **kw)
            outs[name] = o.clone()
            # which entry points the call runs (the advisor's round-5 finding: the comparison below would pass vacuously if the
            # gate of the split-K path were broken): the same call under the per-kernel timers, which record every launch by name
            with ops.KernelTimers() as kt:
                pred.predict(scene["rgb"], frame["depth_t"], scene["K"], P0, frame["xyz_t"], iteration=1, **kw)
            splitk_launches[name] = kt.summary().get("fp_igemm_f16_splitk_fwd", dict(calls=0))["calls"]
            if name == "product":
                raw = {k: v.clone() for k, v in pred.last_raw_output.items()}
                trk = GraphedTracker(pred, gmesh, scene["diameter"], scene["K"], 480, 640, n_hyp=n, iteration=2, device=dev).capture()
                g1 = trk.step(scene["rgb"], scene["depth"], P0).clone()
                e1 = trk.step_eager(scene["rgb"], scene["depth"], P0).clone()
                assert torch.equal(g1, e1), "graph replay of the forked heads differs from the eager launches"
    # the encoder's convolutions (and, below ROWS_QKV_MIN_ROWS rows, the in_proj) on the split-K entry point -- 14 launches at n = 12, where
    # a layer whose tile count already fills the chip is given one piece -- and none with the threshold at zero
    assert splitk_launches["product"] >= 10 and splitk_launches["one_stream"] == splitk_launches["product"] and splitk_launches["plain_convs"] == 0, splitk_launches
    assert torch.equal(outs["product"], outs["one_stream"]), "two-stream heads changed a result"
    assert set(raw) == {"trans", "rot"}
    # split-K against the plain convolutions: contraction-scaled heads, so the 2-iteration chain compares arithmetic, not chaos
    dR = _geodesic(outs["product"][:, :3, :3].cpu().numpy(), outs["plain_convs"][:, :3, :3].cpu().numpy())
    dt = np.linalg.norm((outs["product"][:, :3, 3] - outs["plain_convs"][:, :3, 3]).cpu().numpy(), axis=1)
    assert dR.max() <= 1e-4 and dt.max() <= 1e-5, (dR.max(), dt.max())      # measured 2.2e-5 rad / 5.1e-6 m over two free-running iterations


def test_bench_under_the_launcher_with_rccl_matches_the_headline(dev):
    """round 6 (the round-5 verdict's item 6): no multi-GPU node has ever been available to the builder or the driver, so the first
    8-GPU run must not die on plumbing.  bench.py end to end, launched EXACTLY as the driver launches N > 1 (python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...), with N = 1
    and RCCL initialised (FP_BENCH_FORCE_DIST=1): object mode issues the fused [score | pose] all-gather on RCCL every step,
    hypothesis mode goes through register_hypothesis_parallel / FeaturePoseExchange.  Each JSON line is checked for the contract's
    fields and its rate against the plain single-process headline of the same box within 2 % (hypothesis mode, whose step also builds
    and exchanges the records: 3 %; one re-run allowed: the chip clocks to its power budget)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fast = ["--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-kernel-table", "--no-extras"]

    def run(mode, launcher):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FP_BENCH_FORCE_DIST"):
            env.pop(k, None)
        cmd = [sys.executable]
        if launcher:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)]
            env["FP_BENCH_FORCE_DIST"] = "1"
        cmd += [os.path.join(root, "bench.py"), "--gpus", "1", "--mode", mode] + fast
        r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (cmd, r.stderr[-2000:])
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    def check(d):
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                  "data", "config"):
            assert k in d, k
        assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 3 and d["higher_is_better"] is True
        assert abs(d["value"] - 252 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    rep = {}
    # object mode is the headline's own step + one all-gather: 2 %; the hypothesis-mode step also builds and exchanges the [feature | pose]
    # records and sorts on every rank (measured 1.2 % above the plain step at world size 1): 3 %
    tol = dict(object=0.02, hypothesis=0.03)
    for attempt in range(2):
        plain = run("object", launcher=False)
        check(plain)
        ok = True
        for mode in ("object", "hypothesis"):
            d = run(mode, launcher=True)
            check(d)
            assert "RCCL" in d["config"]["parallelism"] or "rccl" in d["config"]["parallelism"].lower(), d["config"]
            rep[mode] = dict(ms_per_step=d["ms_per_step"], plain_ms_per_step=plain["ms_per_step"], ratio=d["ms_per_step"] / plain["ms_per_step"])
            ok = ok and abs(rep[mode]["ratio"] - 1.0) <= tol[mode]
        if ok:
            break
    REPORT = os.path.join(root, "gpurun_out", "bench_rccl_world1.json")
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    json.dump(rep, open(REPORT, "w"), indent=1)
    for mode, r in rep.items():
        assert abs(r["ratio"] - 1.0) <= tol[mode], rep
