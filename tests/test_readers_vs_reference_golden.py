"""foundationpose_amd/datareader.py against the REFERENCE's own readers: the two small
datasets under tests/golden/reader_seq/ were read by the reference's YcbineoatReader / BopBaseReader classes
(tests/golden/make_golden_readers.py, build container); here the same directories are read by ours."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "reader_seq")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "readers_golden.npz"), allow_pickle=False)


@pytest.mark.parametrize("tag,kw", [("full", dict(zfar=1.2)), ("half", dict(shorter_side=24, zfar=np.inf))])
def test_demo_sequence_reader_matches_reference(gold, tag, kw):
    from foundationpose_amd.datareader import YcbineoatReader
    r = YcbineoatReader(os.path.join(ROOT, "demo"), **kw)
    np.testing.assert_array_equal(r.K, gold[f"demo_{tag}_K"])
    assert [r.H, r.W, len(r)] == gold[f"demo_{tag}_HW"].tolist()
    assert list(r.id_strs) == gold[f"demo_{tag}_ids"].tolist()
    assert r.get_video_name() == str(gold[f"demo_{tag}_video_name"])
    for i in range(len(r)):
        c, d, m = r.get_color(i), r.get_depth(i), r.get_mask(i)
        assert c.dtype == gold[f"demo_{tag}_color{i}"].dtype and m.dtype == gold[f"demo_{tag}_mask{i}"].dtype
        np.testing.assert_array_equal(c, gold[f"demo_{tag}_color{i}"])
        np.testing.assert_array_equal(d, gold[f"demo_{tag}_depth{i}"])          # same float64 arithmetic: exact
        np.testing.assert_array_equal(m, gold[f"demo_{tag}_mask{i}"])
        np.testing.assert_array_equal(r.get_gt_pose(i), gold[f"demo_{tag}_pose{i}"])
    assert (gold[f"demo_{tag}_depth0"] == 0).any() and (gold[f"demo_{tag}_depth0"] > 0).any()
    # get_xyz_map = depth2xyzmap(get_depth, K) (a device op in the product, pinned by the GPU tests); the CPU oracle
    # closes the chain here
    from oracle import ops as oo
    np.testing.assert_allclose(oo.depth2xyzmap(r.get_depth(0).astype(np.float32), r.K, f64_internal=True), gold[f"demo_{tag}_xyz0"], rtol=0, atol=1e-6)


def test_bop_scene_reader_matches_reference(gold):
    from foundationpose_amd.datareader import BopBaseReader
    r = BopBaseReader(os.path.join(ROOT, "bop", "000048"), zfar=1.5, resize=1)
    assert len(r) == int(gold["bop_r1_n"][0])
    assert list(r.id_strs) == gold["bop_r1_ids"].tolist()
    assert r.get_video_id() == int(gold["bop_r1_video_id"][0])
    assert r.bop_depth_scale == float(gold["bop_r1_depth_scale"][0])
    for i in range(len(r)):
        np.testing.assert_array_equal(r.get_K(i), gold[f"bop_r1_K{i}"])
        np.testing.assert_array_equal(r.get_color(i), gold[f"bop_r1_color{i}"])
        np.testing.assert_allclose(r.get_depth(i), gold[f"bop_r1_depth{i}"], rtol=0, atol=1e-15)
        np.testing.assert_array_equal(r.get_instance_ids_in_image(i), gold[f"bop_r1_obs{i}"])
        for ob in (5, 12):
            np.testing.assert_array_equal(r.get_mask(i, ob), gold[f"bop_r1_mask{i}_{ob}"])
            np.testing.assert_allclose(r.get_gt_pose(i, ob), gold[f"bop_r1_pose{i}_{ob}"], rtol=0, atol=1e-15)
            np.testing.assert_allclose(r.get_gt_poses(i, ob), gold[f"bop_r1_poses{i}_{ob}"], rtol=0, atol=1e-15)
    from oracle import ops as oo
    np.testing.assert_allclose(oo.depth2xyzmap(r.get_depth(0).astype(np.float32), r.get_K(0), f64_internal=True), gold["bop_r1_xyz0"], rtol=0, atol=1e-6)


def test_pose_error_metrics_match_reference(gold):
    """ADD, ADD-S, AUC and the projection helper against the reference's Utils.add_err / adds_err / compute_auc_sklearn /
    project_3d_to_2d (run in the build container on the same random poses)"""
    from foundationpose_amd import vis
    pts, P, G = gold["metric_pts"], gold["metric_pred"], gold["metric_gt"]
    np.testing.assert_allclose([vis.add_err(p, g, pts) for p, g in zip(P, G)], gold["metric_add"], rtol=1e-12)
    np.testing.assert_allclose([vis.adds_err(p, g, pts) for p, g in zip(P, G)], gold["metric_adds"], rtol=1e-12)
    errs = gold["metric_errs"]
    ours = [vis.compute_auc(errs), vis.compute_auc(errs, max_val=0.05, step=0.0005), vis.compute_auc(gold["metric_add"]), vis.compute_auc(np.zeros(5))]
    np.testing.assert_allclose(ours, gold["metric_auc"], rtol=1e-12)
    K = gold["metric_K"]
    proj = np.array([vis.project_3d_to_2d(np.append(pts[i], 1.0), K, G[i % 6]) for i in range(12)])
    np.testing.assert_array_equal(proj, gold["metric_proj"])
