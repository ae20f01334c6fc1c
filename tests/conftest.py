import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def scene():
    return _build_scene()


def _build_scene():
    """Seeded synthetic scene of SURVEY.md 8(d), frame rendered with the CPU oracle (test-only)."""
    from foundationpose_amd import synthetic as syn
    from foundationpose_amd.mesh import make_can_mesh
    from foundationpose_amd.Utils import euler_matrix, sample_views_icosphere
    from oracle import ops as oo
    from oracle import pipeline as op

    mesh = make_can_mesh()
    mesh_np = op.mesh_tensors_np(mesh)
    K = syn.YCBV_K.copy()
    T = syn.gt_pose(0)
    full = oo.render_crops(mesh_np, T[None].astype(np.float32), None, K, syn.H, syn.W, (syn.H, syn.W),
                           normalize_xyz=False, want=("color", "depth"))
    rgb, depth, mask = syn.compose_frame(full["color"][0], full["depth"][0])
    diameter = float(np.linalg.norm(mesh.vertices.max(0) - mesh.vertices.min(0)))  # cylinder: exact bbox diagonal
    # 252-pose rotation grid (42 views x 6 in-plane), translation = GT translation with a small offset
    cams = sample_views_icosphere(40)
    grid = []
    for c in cams:
        for a in np.deg2rad(np.arange(0, 360, 60)):
            grid.append(np.linalg.inv(c @ euler_matrix(0, 0, a)))
    grid = np.asarray(grid)
    grid[:, :3, 3] = T[:3, 3] + np.array([0.004, -0.003, 0.01])
    return dict(mesh=mesh, mesh_np=mesh_np, K=K, gt=T, rgb=rgb, depth=depth, mask=mask, diameter=diameter,
                poses=grid.astype(np.float32), H=syn.H, W=syn.W)
