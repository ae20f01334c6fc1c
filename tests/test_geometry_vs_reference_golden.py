"""the product's init-time geometry against golden vectors minted by the REFERENCE's own
functions (tests/golden/make_golden_geometry.py -> geometry_golden.npz): sample_views_icosphere (Utils.py:483-507),
make_rotation_grid (estimater.py:106-124), symmetry_tfs_from_info (Utils.py:806-834), guess_translation
(estimater.py:137-156), compute_mesh_diameter (Utils.py:559-574).  [3P] stand-ins inside the golden: trimesh's icosphere,
transformations.euler_matrix, mycpp.cluster_poses (ref_harness.py)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_geometry as mg  # noqa: E402  (inputs of the golden cases; imports nothing from /root/reference)

G = np.load(os.path.join(ROOT, "tests", "golden", "geometry_golden.npz"))


def test_icosphere_views_match_reference():
    from foundationpose_amd.Utils import sample_views_icosphere
    assert np.array_equal(sample_views_icosphere(n_views=40), G["views_n40"])
    assert np.array_equal(sample_views_icosphere(n_views=1, subdivisions=2), G["views_sub2"])


@pytest.mark.parametrize("name", list(mg.SYM_INFOS))
def test_symmetry_tfs_match_reference(name):
    from foundationpose_amd.Utils import symmetry_tfs_from_info
    got = symmetry_tfs_from_info(mg.SYM_INFOS[name], rot_angle_discrete=5)
    ref = G[f"sym_{name}"]
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)          # rotations, mm -> m translations and offsets bit for bit
    if name == "cont_z":
        assert np.array_equal(symmetry_tfs_from_info(mg.SYM_INFOS[name], rot_angle_discrete=30), G["sym_cont_z_step30"])


@pytest.mark.parametrize("name,sym", [("identity", None), ("both", "sym_both"), ("cont_x", "sym_cont_x"), ("discrete", "sym_discrete")])
def test_rotation_grid_matches_reference(name, sym):
    """the 252-pose grid, and what the greedy symmetry-aware clustering keeps of it (order included)"""
    from foundationpose_amd.estimater import FoundationPose
    S = np.eye(4)[None] if sym is None else G[sym]
    me = types.SimpleNamespace(symmetry_tfs=torch.as_tensor(S, dtype=torch.float), device=torch.device("cpu"))
    FoundationPose.make_rotation_grid(me, min_n_views=40, inplane_step=60)
    got, ref = me.rot_grid.numpy(), G[f"grid_{name}"]
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_guess_translation_matches_reference(device):
    """masked median on the device == numpy's (mean of the two middle values for an even count), bbox centre, zeros for
    an empty mask / no valid depth"""
    from foundationpose_amd.estimater import FoundationPose
    me = types.SimpleNamespace(device=torch.device(device))
    for name, depth, mask in mg.guess_cases():
        got = FoundationPose.guess_translation(me, depth=torch.as_tensor(depth, device=device), mask=mask, K=mg.K)
        assert np.array_equal(np.asarray(got, dtype=np.float64).reshape(3), G[f"guess_{name}"]), name
        got2 = FoundationPose.guess_translation(me, depth=depth, mask=torch.as_tensor(mask, device=device), K=mg.K)
        assert np.array_equal(np.asarray(got2, dtype=np.float64).reshape(3), G[f"guess_{name}"]), name


def test_mesh_diameter_matches_reference():
    from foundationpose_amd.Utils import compute_mesh_diameter
    for name, pts, n_sample, seed in mg.diameter_cases():
        np.random.seed(seed)
        got = compute_mesh_diameter(model_pts=pts, n_sample=n_sample)
        assert np.float64(got) == G[f"diam_{name}"], (name, got, G[f"diam_{name}"])


@pytest.mark.parametrize("name,sym", [("identity", None), ("both", "sym_both"), ("cont_x", "sym_cont_x"), ("discrete", "sym_discrete")])
def test_oracle_cluster_poses_matches_reference_grid(name, sym):
    """the oracle's C restatement of mycpp.cluster_poses keeps exactly the poses the reference's make_rotation_grid kept"""
    from oracle import ops as oo
    S = np.eye(4)[None] if sym is None else G[sym]
    full = G["grid_identity"].astype(np.float64)           # identity symmetry keeps all 252 poses, in generation order
    assert full.shape == (252, 4, 4)
    keep = oo.cluster_poses(30, 99999, full, S)
    assert np.array_equal(full[keep].astype(np.float32), G[f"grid_{name}"])
