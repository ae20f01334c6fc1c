"""Mints tests/golden/geometry_golden.npz: the init-time geometry of SURVEY 8(a) row a4 / 8(f) row 2, computed by the
REFERENCE's own functions (imported from /root/reference through tests/golden/ref_harness.py):

  views_*      Utils.sample_views_icosphere (Utils.py:483-507) for n_views=40 and for subdivisions=2
  grid_*       FoundationPose.make_rotation_grid (estimater.py:106-124) with the identity symmetry set and with a
               discrete + continuous set built by Utils.symmetry_tfs_from_info
  sym_*        Utils.symmetry_tfs_from_info (Utils.py:806-834): continuous about x / y / z, discrete only, both, none
  guess_*      FoundationPose.guess_translation (estimater.py:137-156): odd / even number of valid depths (numpy's
               median averages the two middle values), empty mask, mask without a valid depth
  diam_*       Utils.compute_mesh_diameter(model_pts=...) (Utils.py:559-574) with the numpy seed set as
               estimater.register leaves it (set_seed(0) is called per register; reset_object runs before it, so the test
               seeds explicitly): fewer points than n_sample, and more

What is NOT the reference's code here (absent from this image, replaced by stand-ins in ref_harness.py, [3P]):
trimesh.creation.icosphere, transformations.euler_matrix, mycpp.cluster_poses (C++ needing Eigen/Boost).

    python tests/golden/make_golden_geometry.py        (build container only)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

SYM_INFOS = {
    "none": {},
    "cont_z": {"symmetries_continuous": [{"axis": [0, 0, 1], "offset": [0, 0, 0]}]},
    "cont_y_offset": {"symmetries_continuous": [{"axis": [0, 1, 0], "offset": [0.001, -0.002, 0.003]}]},
    "cont_x": {"symmetries_continuous": [{"axis": [1, 0, 0], "offset": [0, 0, 0]}]},
    # BOP models_info.json stores these as floats (an integer list makes the reference's in-place `*= 0.001` raise)
    "discrete": {"symmetries_discrete": [[-1.0, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1],
                                         [1.0, 0, 0, 12.5, 0, -1, 0, -3.0, 0, 0, -1, 40.0, 0, 0, 0, 1]]},
    "both": {"symmetries_discrete": [[-1.0, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1]],
             "symmetries_continuous": [{"axis": [0, 0, 1], "offset": [0, 0, 0]}]},
}
K = np.array([[1066.778, 0, 312.9869], [0, 1067.487, 241.3109], [0, 0, 1]])


def guess_cases():
    """(name, depth f32 (480,640), mask uint8)"""
    rng = np.random.default_rng(11)
    depth = (0.6 + 0.3 * rng.random((480, 640))).astype(np.float32)
    depth[rng.random((480, 640)) < 0.1] = 0
    cases = []
    m = np.zeros((480, 640), np.uint8)
    m[100:231, 200:333] = 1
    cases.append(("odd_or_even_a", depth, m))
    m2 = m.copy()
    vs, us = np.nonzero((m2 > 0) & (depth >= 0.001))
    m2[vs[0], us[0]] = 0                        # one valid pixel fewer: the other parity of the count
    cases.append(("odd_or_even_b", depth, m2))
    m3 = np.zeros((480, 640), np.uint8)
    m3[5, 7] = 255
    m3[400, 630] = 3
    cases.append(("two_pixels", depth, m3))
    cases.append(("empty_mask", depth, np.zeros((480, 640), np.uint8)))
    d4 = depth.copy()
    d4[m > 0] = 0.0005
    cases.append(("no_valid_depth", d4, m))
    return cases


def diameter_cases():
    from foundationpose_amd.mesh import make_can_mesh
    rng = np.random.default_rng(5)
    return [("can", np.asarray(make_can_mesh().vertices, dtype=np.float64), 10000, 0),
            ("cloud", rng.normal(size=(3000, 3)) * np.array([0.05, 0.02, 0.09]), 1000, 0),
            ("cloud_all", rng.normal(size=(700, 3)), None, 3)]


def main():
    import ref_harness as rh
    ns = rh.load_reference()
    U, E = ns.Utils, ns.estimater
    out = {}
    out["views_n40"] = U.sample_views_icosphere(n_views=40)
    out["views_sub2"] = U.sample_views_icosphere(n_views=1, subdivisions=2)
    for name, info in SYM_INFOS.items():
        out[f"sym_{name}"] = U.symmetry_tfs_from_info(info, rot_angle_discrete=5)
    out["sym_cont_z_step30"] = U.symmetry_tfs_from_info(SYM_INFOS["cont_z"], rot_angle_discrete=30)
    for name, sym in (("identity", np.eye(4)[None]), ("both", out["sym_both"]), ("cont_x", out["sym_cont_x"]),
                      ("discrete", out["sym_discrete"])):
        me = types.SimpleNamespace(symmetry_tfs=torch.as_tensor(sym, dtype=torch.float))
        E.FoundationPose.make_rotation_grid(me, min_n_views=40, inplane_step=60)
        out[f"grid_{name}"] = me.rot_grid.numpy()
    me = types.SimpleNamespace(debug=0)
    for name, depth, mask in guess_cases():
        out[f"guess_{name}"] = np.asarray(E.FoundationPose.guess_translation(me, depth=depth, mask=mask, K=K), dtype=np.float64)
    for name, pts, n_sample, seed in diameter_cases():
        np.random.seed(seed)
        out[f"diam_{name}"] = np.float64(U.compute_mesh_diameter(model_pts=pts, n_sample=n_sample))
    path = os.path.join(HERE, "geometry_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    for k, v in out.items():
        print(f"  {k}: {np.shape(v)} {np.asarray(v).dtype}")


if __name__ == "__main__":
    main()
