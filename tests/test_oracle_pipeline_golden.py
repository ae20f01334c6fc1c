"""Pins the image-space oracle (oracle/fp_oracle.c, oracle/pipeline.py) against golden vectors produced by the
REFERENCE's own hot-path Python (tests/golden/pipeline_golden.npz, minted by tests/golden/make_golden_pipeline.py
which imports /root/reference through tests/golden/ref_harness.py).

Bit-exact where the reference's arithmetic is fully determined by its own code (crop windows, back-projection, the
nearest-neighbour xyz crops).  Float tolerances elsewhere are set by what separates the two computations -- the
golden side evaluates nvdiffrast's per-pixel barycentrics in float64 and torch's bilinear taps in torch's order; the
oracle fixes a float32 expression order so that the HIP kernels can match it bit for bit:
  * A (rendered crop): xyz within 5e-4 (normalised units, i.e. < 5e-5 m) everywhere; rgb within 1e-3 on >= 99.9 % of
    the pixels -- the rest are texture-edge pixels where a 1e-7 change of the uv coordinate selects the other texel
  * B (observed crop): rgb within 1e-4, xyz exact; scorer variant exact except "tie" pixels whose hop-2 sample
    coordinate is a half-integer up to float32 noise inside torch's normalised-homography chain (< 0.2 %, all on the crop's first row / column)
  * one refine iteration / one score pass through the reference predictors: 1e-3 m, 5e-3 (rotation elements), 0.3
    logits (the stand-in networks amplify the texture-edge pixels above; see DESIGN.md "Parity")
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


@pytest.fixture(scope="module")
def frame(scene):
    from oracle import ops as oo
    from oracle import pipeline as op
    d = op.preprocess_depth(scene["depth"])
    return dict(depth=d, xyz=oo.depth2xyzmap(d, scene["K"], f64_internal=True))


def test_crop_windows_match_reference(scene, gold):
    from oracle import ops as oo
    tf, bb = oo.crop_windows(scene["poses"], scene["K"], scene["diameter"], 1.2, (160, 160))
    assert np.array_equal(tf, gold["g1_tf_to_crops"])            # Utils.compute_crop_window_tf_batch, bit-exact
    np.testing.assert_allclose(bb, gold["g1_bbox2d"], rtol=0, atol=1e-4)  # torch batched inverse vs closed form (px)


def test_back_projection_matches_reference(scene, gold, frame):
    from oracle import ops as oo
    assert np.array_equal(oo.depth2xyzmap(frame["depth"], scene["K"], f64_internal=True)[::4, ::4], gold["g2_xyz_np"])
    assert np.array_equal(oo.depth2xyzmap(frame["depth"], scene["K"], zfar=1.0, f64_internal=False)[::4, ::4], gold["g2_xyz_batch"])


def _check_A(A, G):
    A = A[:, :, ::2, ::2]
    d = np.abs(A - G)
    assert d[:, 3:].max() <= 5e-4, d[:, 3:].max()
    assert (d[:, :3] > 1e-3).mean() <= 1e-3 and d[:, :3].max() <= 0.2, ((d[:, :3] > 1e-3).mean(), d[:, :3].max())
    assert np.array_equal(A[:, 3:].any(1) | (A[:, :3].any(1)), G[:, 3:].any(1) | G[:, :3].any(1))  # same coverage


@pytest.mark.parametrize("norm", [1, 0])
def test_refiner_inputs_match_reference(scene, gold, frame, norm):
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG
    from oracle import pipeline as op
    cfg = dict(DEFAULT_REFINE_CFG, normalize_xyz=bool(norm))
    A, B, _, _ = op.refine_inputs(cfg, gold["poses_in"], scene["mesh_np"], scene["rgb"], frame["xyz"], scene["K"], scene["diameter"])
    _check_A(A, gold[f"g3_refine_A_norm{norm}"])
    Bs, G = B[:, :, ::2, ::2], gold[f"g3_refine_B_norm{norm}"]
    np.testing.assert_allclose(Bs[:, :3], G[:, :3], rtol=0, atol=1e-4)
    assert np.array_equal(Bs[:, 3:], G[:, 3:])
    assert (G[2, :3] == 0).mean() > 0.05 and np.abs(G[2, :3]).max() > 0   # pose 2 exercises the zero padding


def test_scorer_inputs_match_reference(scene, gold, frame):
    from foundationpose_amd.weights import DEFAULT_SCORE_CFG
    from oracle import pipeline as op
    A, B, _, _ = op.score_inputs(dict(DEFAULT_SCORE_CFG), gold["poses_in"], scene["mesh_np"], scene["rgb"], frame["depth"],
                                 scene["K"], scene["diameter"])
    _check_A(A, gold["g3_score_A"])
    Bs, G = B[:, :, ::2, ::2], gold["g3_score_B"]
    np.testing.assert_allclose(Bs[:, :3], G[:, :3], rtol=0, atol=1e-4)
    ties = (Bs[:, 3:] != G[:, 3:]).any(1)
    assert ties.mean() < 2e-3, ties.mean()
    assert ties[:, 1:, 1:].sum() == 0  # ties sit on the crop's first row / column (frame pixel = window edge)


def test_reference_predictors_one_pass(scene, gold, frame):
    """PoseRefinePredictor.predict (1 iteration) and ScorePredictor.predict of the reference vs the oracle pipeline"""
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    from oracle import pipeline as op
    rcfg, scfg = dict(DEFAULT_REFINE_CFG), dict(DEFAULT_SCORE_CFG)
    P = gold["poses_in"]
    tr = []
    out = op.refine_predict(rcfg, random_state_dict("refine", rcfg, 0), scene["rgb"], frame["depth"], scene["K"], P, frame["xyz"],
                            scene["mesh_np"], scene["diameter"], iteration=1, trace=tr)
    # measured (round 4, after the rasteriser's tie rule was defined in nvdiffrast's window space and the golden re-minted): 3.0e-6 m,
    # 1.8e-5 on the rotation entries, 3.0e-6 on the translation delta -- the 1e-3 / 5e-3 band of rounds 1-2 was the stand-in network's
    # response to a dozen texture-edge pixels that no longer differ (test_one_pass_deviation_is_explained_by_the_rendered_inputs)
    assert np.abs(out[:, :3, 3] - gold["g4_refined_1it"][:, :3, 3]).max() <= 2e-5
    assert np.abs(out[:, :3, :3] - gold["g4_refined_1it"][:, :3, :3]).max() <= 1e-4
    np.testing.assert_allclose(tr[0]["trans"] * np.float32(scene["diameter"] / 2), gold["g4_trans_delta"], atol=2e-5)
    assert np.abs(out[:, :3, 3] - P[:, :3, 3]).max() > 5e-3       # a real update, not a no-op
    trs = []
    s = op.score_predict(scfg, random_state_dict("score", scfg, 0), scene["rgb"], frame["depth"], scene["K"], P, scene["mesh_np"],
                         scene["diameter"], trace=trs)
    # measured: |score - reference| = 0.44, 0.37, 0.034 on logits of spread (std) 8.9.  The two larger ones are the hypotheses whose
    # observed crop has nearest-neighbour TIES on its first row (the frame pixel that falls exactly on the window edge: kornia's
    # stand-in and fp_oracle.c's deterministic rule pick different neighbours there, test_scorer_inputs_match_reference; ~20 of
    # 25 600 xyz pixels each); the hypothesis without a tie pixel agrees to 0.034.  Gate: 0.05 without ties, 0.6 (7 % of the spread)
    # with -- the advisor's round-3 question why 0.3 became 0.6: the re-minted golden moved which hypotheses carry ties, not the
    # arithmetic.
    ties = (trs[0]["B"][:, 3:, ::2, ::2] != gold["g3_score_B"][:, 3:]).any(1).reshape(len(s), -1).sum(1)
    dev = np.abs(s - gold["g4_scores"])
    assert (dev[ties == 0] <= 0.05).all() and (dev <= 0.6).all() and (ties == 0).any() and (ties > 0).any(), (dev, ties)
    assert np.argmax(s) == np.argmax(gold["g4_scores"]) and np.argmin(s) == np.argmin(gold["g4_scores"])


def test_one_pass_deviation_is_explained_by_the_rendered_inputs(scene, gold, frame):
    """The 1e-3 m / 5e-3 band of test_reference_predictors_one_pass, taken apart.  (a) Networks: the oracle's RefineNet on
    the REFERENCE's own network inputs (captured inside its predictor) reproduces the reference's raw outputs to 2e-5
    -- the networks are not the source.  (b) Inputs: the oracle's rendered crop A differs
    from the reference's on a small set of pixels: colour on texture edges (the stand-in rasteriser of ref_harness.py
    evaluates barycentrics in float64, fp_oracle.c in float32 like nvdiffrast; one texel step of the 512x512 texture is a
    jump of up to 0.7 in colour) and the silhouette; xyz agrees to 5e-6.  (c) Sensitivity: the stand-in network turns that
    input difference into the whole output deviation -- swapping only the inputs moves the oracle's output onto the
    reference's."""
    import torch
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    from oracle import nets
    from oracle import pipeline as op
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, 0)
    P = gold["poses_in"]
    A_ref, B_ref = gold["g4_net_A0"].astype(np.float32)[None], gold["g4_net_B0"].astype(np.float32)[None]
    o_ref_in = nets.refine_forward(torch.from_numpy(A_ref), torch.from_numpy(B_ref), sd)
    # (a) same inputs -> same outputs
    assert np.abs(o_ref_in["trans"].numpy()[0] - gold["g4_raw_trans"][0]).max() < 2e-5
    assert np.abs(o_ref_in["rot"].numpy()[0] - gold["g4_raw_rot"][0]).max() < 2e-5
    # (b) what differs between the two renderers
    A, B, _, _ = op.refine_inputs(cfg, P[:1], scene["mesh_np"], scene["rgb"], frame["xyz"], scene["K"], scene["diameter"])
    dA, dB = np.abs(A - A_ref), np.abs(B - B_ref)
    assert dB[:, :3].max() < 1e-4 and (dB[:, 3:] > 0).mean() < 2e-3      # observed crop: rgb 3e-5, xyz exact up to edge ties
    assert dA[:, 3:].max() < 5e-4                                        # rendered xyz agrees ...
    n_rgb = int((dA[0, :3].max(axis=0) > 1e-3).sum())
    # ... rendered colour: a few hundred texture-edge / silhouette pixels in rounds 1-2 (hence the old `0 < n_rgb` lower bound), NONE
    # since round 3's tie-rule fix + re-minted golden: the renderers now agree to 1e-3 on every pixel of this hypothesis
    assert n_rgb < 0.02 * 160 * 160, n_rgb
    # (c) the output deviation of the full pass comes from those pixels
    o_own_in = nets.refine_forward(torch.from_numpy(A), torch.from_numpy(B), sd)
    dev_total = np.abs(o_own_in["rot"].numpy()[0] - gold["g4_raw_rot"][0]).max()
    dev_inputs_only = np.abs(o_own_in["rot"].numpy()[0] - o_ref_in["rot"].numpy()[0]).max()
    assert abs(dev_total - dev_inputs_only) < 5e-5 and dev_total < 1e-4, (dev_total, dev_inputs_only)
    print(f"texture-edge pixels: {n_rgb}, raw rot deviation {dev_total:.2e} (inputs only: {dev_inputs_only:.2e})")


def test_deepim_translation_matches_reference(scene, gold, frame):
    """trans_rep='deepim' (predict_pose_refine.py:201-215) through the oracle pipeline vs the reference predictor"""
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    from oracle import pipeline as op
    cfg = dict(DEFAULT_REFINE_CFG, trans_rep="deepim")
    P = gold["poses_in"]
    out = op.refine_predict(cfg, random_state_dict("refine", dict(DEFAULT_REFINE_CFG), 0), scene["rgb"], frame["depth"], scene["K"], P,
                            frame["xyz"], scene["mesh_np"], scene["diameter"], iteration=1)
    ref = gold["g6_deepim_refined_1it"]
    # the translation now scales with depth * network output: same band as the tracknet pass, relative to the step size
    step = np.abs(ref[:, :3, 3] - P[:, :3, 3]).max()
    assert step > 1e-2
    assert np.abs(out[:, :3, 3] - ref[:, :3, 3]).max() <= 0.03 * step
    assert np.abs(out[:, :3, :3] - ref[:, :3, :3]).max() <= 5e-3
    # closed form on the reference's own raw outputs: exact to float32 rounding
    from oracle import ops as oo
    tf, _ = oo.crop_windows(P, scene["K"], scene["diameter"], cfg["crop_ratio"], (160, 160))
    dt = oo.deepim_trans_delta(gold["g4_raw_trans"], P, scene["K"], tf, 160, True, scene["diameter"])
    np.testing.assert_allclose(dt, gold["g6_deepim_trans_delta"], atol=2e-6, rtol=1e-5)


def test_pose_update_matches_reference(scene, gold):
    """Utils.egocentric_delta_pose_to_pose + the axis-angle / 6d branches of predict_pose_refine.py:217-234"""
    from oracle import ops as oo
    x = gold["g5_inputs"]
    tr, ro, ro6 = x[:, :3], x[:, 3:6], x[:, 6:]
    P = scene["poses"][:8]
    a = oo.pose_update(tr, ro, P, "axis_angle", True, (0.02, 0.02, 0.05), 0.349, scene["diameter"])
    np.testing.assert_allclose(a, gold["g5_axis_angle"], rtol=0, atol=2e-6)
    b = oo.pose_update(tr, ro6, P, "6d", False, (0.02, 0.02, 0.05), 0.349, scene["diameter"])
    np.testing.assert_allclose(b, gold["g5_6d"], rtol=0, atol=2e-6)
