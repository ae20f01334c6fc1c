"""On-disk formats either side of the path: OBJ meshes and demo-format RGB-D sequences."""
import numpy as np
import pytest


def test_obj_round_trip(tmp_path):
    from foundationpose_amd.mesh import make_can_mesh
    from foundationpose_amd.mesh_io import load_mesh, save_obj
    for textured in (True, False):
        mesh = make_can_mesh(n_ang=12, n_axial=5, textured=textured, tex_size=32)
        path = str(tmp_path / f"can{int(textured)}.obj")
        save_obj(mesh, path)
        back = load_mesh(path)
        assert back.faces.shape == mesh.faces.shape
        # the loader merges (v, vt) pairs: same triangles in space, possibly renumbered
        tri_a = np.sort(mesh.vertices[mesh.faces].reshape(len(mesh.faces), -1), axis=1)
        tri_b = np.sort(back.vertices[back.faces].reshape(len(back.faces), -1), axis=1)
        np.testing.assert_allclose(tri_a, tri_b, rtol=0, atol=1e-8)
        if textured:
            assert back.visual.uv is not None and np.asarray(back.visual.material.image).shape == (32, 32, 3)
            assert np.array_equal(np.asarray(back.visual.material.image), np.asarray(mesh.visual.material.image)[..., :3])
            ua = mesh.visual.uv[mesh.faces].reshape(len(mesh.faces), -1)
            ub = back.visual.uv[back.faces].reshape(len(back.faces), -1)
            np.testing.assert_allclose(ua, ub, atol=1e-8)       # per-corner texture coordinates survive
        else:
            assert getattr(back.visual, "uv", None) is None
        np.testing.assert_allclose(np.linalg.norm(back.vertex_normals, axis=1), 1.0, atol=1e-6)


def test_obj_quads_negative_indices_and_mtl_colour(tmp_path):
    from foundationpose_amd.mesh_io import load_obj
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1.0 0.5 0.0\n")
    (tmp_path / "q.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf -4 -3 -2 -1\n")
    m = load_obj(str(tmp_path / "q.obj"))
    assert m.faces.shape == (2, 3) and m.vertices.shape == (4, 3)
    assert np.array_equal(np.asarray(m.visual.vertex_colors)[0, :3], [255, 127, 0])
    n = m.vertex_normals
    assert np.allclose(np.abs(n[:, 2]), 1.0)


def test_sequence_round_trip(tmp_path, scene):
    from foundationpose_amd.datareader import YcbineoatReader, write_sequence
    d = str(tmp_path / "seq")
    depth2 = scene["depth"] + 0.01
    write_sequence(d, scene["K"], [scene["rgb"], scene["rgb"][::-1]], [scene["depth"], depth2], [scene["mask"], scene["mask"]],
                   gt_poses=[scene["gt"], scene["gt"]])
    r = YcbineoatReader(d, zfar=1.0)
    assert len(r) == 2 and r.id_strs == ["0000000", "0000001"] and (r.H, r.W) == (480, 640)
    np.testing.assert_allclose(r.K, scene["K"])
    assert np.array_equal(r.get_color(0), scene["rgb"]) and np.array_equal(r.get_color(1), scene["rgb"][::-1])
    assert np.array_equal(r.get_mask(0), (scene["mask"] > 0).astype(np.uint8))
    dep = r.get_depth(0)
    ok = (scene["depth"] >= 0.001) & (scene["depth"] < 0.9995)
    assert np.abs(dep[ok] - scene["depth"][ok]).max() <= 0.5e-3 + 1e-9          # uint16 millimetres
    assert (dep[scene["depth"] >= 1.0005] == 0).all()                            # zfar
    np.testing.assert_allclose(r.get_gt_pose(1), scene["gt"], atol=1e-12)
    half = YcbineoatReader(d, shorter_side=240)
    assert (half.H, half.W) == (240, 320) and np.isclose(half.K[0, 0], scene["K"][0, 0] / 2)
    assert np.array_equal(half.get_color(0), scene["rgb"][::2, ::2])


def test_metrics_and_debug_drawing(scene):
    from foundationpose_amd import vis
    pts = np.asarray(scene["mesh"].vertices)
    gt = scene["gt"]
    moved = gt.copy()
    moved[:3, 3] += [0.01, 0, 0]
    assert abs(vis.add_err(moved, gt, pts) - 0.01) < 1e-9 and vis.add_err(gt, gt, pts) == 0
    assert 0 < vis.adds_err(moved, gt, pts) <= 0.01 + 1e-9
    # a half-turn about the can's axis is invisible to ADD-S (symmetric point set) but not to ADD
    flip = gt @ np.diag([-1.0, -1.0, 1.0, 1.0])
    assert vis.adds_err(flip, gt, pts) < 1e-3 < vis.add_err(flip, gt, pts)
    assert abs(vis.compute_auc([0.0] * 10) - 1.0) < 1e-9 and vis.compute_auc([1.0] * 10) == 0.0
    assert 0.45 < vis.compute_auc(np.linspace(0, 0.1, 101)) < 0.55
    dv = vis.depth_to_vis(scene["depth"], inverse=True)
    assert dv.shape == (480, 640, 3) and dv.dtype == np.uint8
    g = vis.make_grid_image([np.zeros((10, 12, 3)), np.ones((10, 12, 3)) * 200, np.zeros((10, 12, 3))], nrow=2, padding=2)
    assert g.shape == (2 * 12 + 2, 2 * 14 + 2, 3) and g[0, 0, 0] == 255 and g[2, 16, 0] == 200
    box = np.stack([pts.min(0), pts.max(0)])
    img = vis.draw_posed_3d_box(scene["K"], scene["rgb"], gt, box)
    img = vis.draw_xyz_axis(img, gt, scale=0.1, K=scene["K"])
    assert img.shape == scene["rgb"].shape and (img != scene["rgb"]).any()
    ys, xs = np.nonzero((img != scene["rgb"]).any(-1))
    m = np.nonzero(scene["mask"])
    assert xs.min() >= m[1].min() - 80 and xs.max() <= m[1].max() + 80      # the overlay sits on the object
    A = np.random.default_rng(0).random((3, 6, 16, 16)).astype(np.float32)
    c = vis.crop_rows_canvas(A, A)
    assert c.ndim == 3 and c.shape[2] == 3 and c.shape[0] > 3 * 16


def _bop_scene(tmp_path, scene):
    from foundationpose_amd.datareader import write_bop_scene
    gt2 = scene["gt"].copy()
    gt2[:3, 3] += [0.3, 0.0, 0.1]
    m2 = np.zeros_like(scene["mask"]); m2[10:40, 500:560] = 1
    inst = [[(5, scene["gt"], scene["mask"]), (5, gt2, m2), (9, gt2, m2)], [(5, scene["gt"], scene["mask"])]]
    sdir, mdir = str(tmp_path / "ycbv" / "test" / "000048"), str(tmp_path / "ycbv" / "models")
    write_bop_scene(sdir, scene["K"], [scene["rgb"]] * 2, [scene["depth"]] * 2, inst, models_dir=mdir, meshes={5: scene["mesh"]})
    return sdir, mdir, gt2, m2


def test_bop_scene_reader_roundtrip(tmp_path, scene):
    """BOP layout (reference datareader.py:155-365): camera / GT json, per-instance visible masks, mm models"""
    from foundationpose_amd.datareader import BopBaseReader, YcbVideoReader
    sdir, mdir, gt2, m2 = _bop_scene(tmp_path, scene)
    r = BopBaseReader(sdir, zfar=5.0, models_dir=mdir)
    assert len(r) == 2 and r.get_video_id() == 48 and r.id_strs == ["000000", "000001"]
    np.testing.assert_allclose(r.get_K(0), scene["K"])
    assert (r.get_color(0) == scene["rgb"]).all()
    d = r.get_depth(0)
    assert d.dtype == np.float64 and np.abs(d - scene["depth"]).max() <= 0.5e-4 + 1e-9      # 0.1 mm depth units
    assert r.get_instance_ids_in_image(0).tolist() == [5, 5, 9] and r.get_instance_ids_in_image(1).tolist() == [5]
    assert (r.get_mask(0, 5) == (scene["mask"] > 0)).all() and (r.get_mask(0, 9) == (m2 > 0)).all()
    assert r.get_mask(1, 9) is None
    np.testing.assert_allclose(r.get_gt_poses(0, 5), np.stack([scene["gt"], gt2]), atol=1e-9)
    np.testing.assert_allclose(r.get_gt_pose(0, 5), scene["gt"], atol=1e-9)
    np.testing.assert_allclose(r.get_gt_pose(0, 5, mask=m2 > 0), gt2, atol=1e-9)        # the instance under the mask
    y = YcbVideoReader(sdir, models_dir=mdir)
    assert y.dataset_name == "ycbv" and y.ob_ids == [5] and y.is_keyframe(0)
    mesh = y.get_gt_mesh(5)
    np.testing.assert_allclose(np.sort(np.asarray(mesh.vertices), 0), np.sort(np.asarray(scene["mesh"].vertices), 0), atol=1e-6)
    assert abs(y.get_model_diameter(5) - np.hypot(0.102, 0.140)) < 1e-3      # models_info diameter = largest vertex distance
    assert y.symmetry_tfs[5].shape == (1, 4, 4)
    h = BopBaseReader(sdir, resize=0.5, models_dir=mdir)
    assert h.get_color(0).shape == (240, 320, 3) and h.get_depth(0).shape == (240, 320) and h.get_K(0)[0, 0] == scene["K"][0, 0] * 0.5


@pytest.mark.parametrize("binary", [True, False])
def test_ply_roundtrip(tmp_path, scene, binary):
    from foundationpose_amd.mesh_io import load_mesh, save_ply
    f = str(tmp_path / "m.ply")
    save_ply(scene["mesh"], f, binary=binary)
    m = load_mesh(f)
    np.testing.assert_allclose(m.vertices, np.asarray(scene["mesh"].vertices, dtype=np.float32), atol=1e-7)
    assert (m.faces == scene["mesh"].faces).all()
    np.testing.assert_allclose(m.vertex_normals, scene["mesh"].vertex_normals, atol=1e-6)
