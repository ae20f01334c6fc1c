"""CPU: pins the oracle -- networks against the reference's own model code (golden vectors), image ops against
analytic properties (the reference ships no fixtures for them: 'parity unpinned')."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "nets_golden.npz")


def _golden_inputs(n, seed):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.golden_inputs(n, seed)


@pytest.mark.parametrize("use_bn", [True, False])
@pytest.mark.parametrize("rot_rep", ["axis_angle", "6d"])
def test_refine_oracle_matches_reference_golden(use_bn, rot_rep):
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    from oracle import nets
    g = np.load(GOLD)
    cfg = dict(DEFAULT_REFINE_CFG, use_BN=use_bn, rot_rep=rot_rep)
    sd = random_state_dict("refine", cfg, seed=1)
    A, B = _golden_inputs(2, 11)
    out = nets.refine_forward(A, B, sd)
    tag = f"refine_bn{int(use_bn)}_{rot_rep}"
    for k in ("trans", "rot"):
        ref = g[f"{tag}_{k}"]
        np.testing.assert_allclose(out[k].numpy(), ref, rtol=2e-4, atol=2e-4 * np.abs(ref).max())


@pytest.mark.parametrize("use_bn", [True, False])
def test_score_oracle_matches_reference_golden(use_bn):
    from foundationpose_amd.weights import DEFAULT_SCORE_CFG, random_state_dict
    from oracle import nets
    g = np.load(GOLD)
    cfg = dict(DEFAULT_SCORE_CFG, use_BN=use_bn)
    sd = random_state_dict("score", cfg, seed=2)
    A, B = _golden_inputs(4, 12)
    for L in (4, 2):
        out = nets.score_forward(A, B, sd, L=L)["score_logit"].numpy()
        ref = g[f"score_bn{int(use_bn)}_L{L}"]
        np.testing.assert_allclose(out, ref, rtol=2e-4, atol=2e-4 * np.abs(ref).max())


def test_product_modules_match_oracle_and_golden():
    """the product's nn.Modules (state-dict compatible with the reference) agree with the golden vectors too"""
    from foundationpose_amd.refine_network import RefineNet
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    g = np.load(GOLD)
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, seed=1)
    net = RefineNet(cfg=cfg, c_in=6).eval()
    net.load_state_dict(sd, strict=True)
    A, B = _golden_inputs(2, 11)
    with torch.no_grad():
        out = net(A, B)
    ref = g["refine_bn1_axis_angle_trans"]
    np.testing.assert_allclose(out["trans"].numpy(), ref, rtol=2e-4, atol=2e-4 * np.abs(ref).max())


# ---------------------------------------------------------------- rasteriser properties
def test_raster_watertight_and_analytic_depth(scene):
    """Closed mesh => every covered pixel has an even number of surface crossings; the winner is the nearest one.
    Depth of the oracle z-buffer must equal the analytic ray/cylinder intersection to ~1e-4 m (faceting)."""
    from oracle import ops as oo
    mesh, K, T = scene["mesh_np"], scene["K"], scene["gt"].astype(np.float32)
    o = oo.render_crops(mesh, T[None], None, K, 480, 640, (480, 640), normalize_xyz=False,
                        want=("depth", "zbuf", "tri_id", "xyz"))
    zb, tri, dep = o["zbuf"][0], o["tri_id"][0], o["depth"][0]
    cov = tri >= 0
    assert cov.sum() > 20000
    assert (zb[~cov] == 0xFFFFFFFF).all() and (dep[~cov] == 0).all()
    # fixed-point key (depth at the 1/16-px snapped vertices) vs float depth (unsnapped vertices, nvdiffrast's
    # per-pixel pass): equal up to the snapping shift (<= 1/32 px of the local depth slope; steep at grazing facets)
    dz = np.abs(zb[cov].astype(np.float64) / 2 ** 20 - dep[cov])
    assert np.median(dz) < 2e-5 and np.percentile(dz, 99) < 5e-4 and dz.max() < 5e-3
    # analytic cylinder (r=0.051, h=0.14) intersection along the pixel rays
    vs, us = np.nonzero(cov)
    rays = np.stack([(us + 0.5 - K[0, 2]) / K[0, 0], (vs + 0.5 - K[1, 2]) / K[1, 1], np.ones(len(us))], 1)
    R, t = T[:3, :3].astype(np.float64), T[:3, 3].astype(np.float64)
    d = rays @ R  # ray dir in object frame (R^T d)
    o0 = -R.T @ t
    a = d[:, 0] ** 2 + d[:, 1] ** 2
    b = 2 * (o0[0] * d[:, 0] + o0[1] * d[:, 1])
    c = o0[0] ** 2 + o0[1] ** 2 - 0.051 ** 2
    disc = b * b - 4 * a * c
    best = np.full(len(us), np.inf)
    ok = disc >= 0
    s = np.where(ok, (-b - np.sqrt(np.where(ok, disc, 0))) / (2 * a), np.inf)
    zhit = o0[2] + s * d[:, 2]
    best = np.where(ok & (np.abs(zhit) <= 0.07) & (s > 0), s, best)
    for zc in (-0.07, 0.07):
        sc = (zc - o0[2]) / d[:, 2]
        p = o0[None] + sc[:, None] * d
        hit = (sc > 0) & (p[:, 0] ** 2 + p[:, 1] ** 2 <= 0.051 ** 2)
        best = np.where(hit & (sc < best), sc, best)
    good = np.isfinite(best)
    assert good.mean() > 0.97  # silhouette pixels of the faceted mesh may miss the analytic surface
    err = np.abs(best[good] - dep[vs[good], us[good]])  # ray parameter s == z_cam because rays have z=1
    assert np.percentile(err, 99) < 2e-3 and np.median(err) < 1.5e-4  # sagitta of the 50-gon = 1e-4 m, amplified at grazing rays


def test_raster_translation_invariance_of_coverage(scene):
    """Shifting the crop window by whole pixels shifts the integer coverage exactly (fixed-point snapping)."""
    from oracle import ops as oo
    mesh, K = scene["mesh_np"], scene["K"]
    P = scene["poses"][:3]
    bb = np.array([[200, 100, 360, 260]] * 3, np.float32)
    a = oo.render_crops(mesh, P, bb, K, 480, 640, (160, 160), want=("tri_id",))["tri_id"]
    b = oo.render_crops(mesh, P, bb + np.array([16, 8, 16, 8], np.float32), K, 480, 640, (160, 160), want=("tri_id",))["tri_id"]
    inner_a = a[:, 8:, 16:]
    inner_b = b[:, :-8, :-16]
    assert (inner_a != inner_b).mean() < 2e-3  # only float-projection rounding at sub-pixel boundaries may differ


def test_depth_filters_edge_cases():
    from oracle import ops as oo
    d = np.full((12, 16), 0.8, np.float32)
    d[0, 0] = 0.0      # invalid corner
    d[5, 5] = 0.9      # isolated outlier
    d[8, :] = 150.0    # beyond zfar
    e = oo.erode_depth(d)
    assert e[5, 5] == 0.0 and e[0, 0] == 0.0 and e[3, 10] == np.float32(0.8)
    b = oo.bilateral_filter_depth(e)
    assert abs(b[3, 10] - 0.8) < 1e-6 and b[8, 3] > 0.79  # zfar row is re-filled from valid neighbours
    z = oo.bilateral_filter_depth(np.zeros((6, 6), np.float32))
    assert (z == 0).all()


def test_crop_window_closed_forms(scene):
    from oracle import ops as oo
    P = scene["poses"][:5]
    tf, bb = oo.crop_windows(P, scene["K"], scene["diameter"], 1.2, (160, 160))
    left, top = -tf[:, 0, 2] / tf[:, 0, 0], -tf[:, 1, 2] / tf[:, 1, 1]
    assert np.allclose(left, np.round(left), atol=1e-3) and np.allclose(top, np.round(top), atol=1e-3)
    side = 160.0 / tf[:, 0, 0]
    assert np.allclose(side, np.round(side), atol=1e-3)
    np.testing.assert_allclose(bb[:, 2] - bb[:, 0], 159.0 / tf[:, 0, 0], rtol=1e-5)  # 159/160 window (App. D.2)


def test_pose_update_against_scipy(scene):
    from scipy.spatial.transform import Rotation
    from oracle import ops as oo
    rng = np.random.default_rng(0)
    P = scene["poses"][:8]
    tr, ro = rng.normal(size=(8, 3)).astype(np.float32), rng.normal(size=(8, 3)).astype(np.float32)
    out = oo.pose_update(tr, ro, P, "axis_angle", True, (1, 1, 1), 0.349, 0.1737)
    w = np.tanh(ro) * 0.349
    dR = Rotation.from_rotvec(w).as_matrix().transpose(0, 2, 1)
    np.testing.assert_allclose(out[:, :3, :3], dR @ P[:, :3, :3], atol=2e-6)
    np.testing.assert_allclose(out[:, :3, 3], P[:, :3, 3] + tr * (0.1737 / 2), atol=1e-6)
    out6 = oo.pose_update(tr, np.concatenate([ro, tr], 1), P, "6d", False, (0.02, 0.02, 0.05), 0.349, 0.1737)
    np.testing.assert_allclose(out6[:, :3, 3], P[:, :3, 3] + np.tanh(tr) * np.array([0.02, 0.02, 0.05]), atol=1e-6)
    Rd = out6[:, :3, :3] @ P[:, :3, :3].transpose(0, 2, 1)
    np.testing.assert_allclose(Rd @ Rd.transpose(0, 2, 1), np.tile(np.eye(3), (8, 1, 1)), atol=1e-5)


def test_attention_core_definition_matches_nn_multiheadattention():
    """fp_attention_f16_fwd is specified (include/fp_amd.h) as what nn.MultiheadAttention(512, 4, batch_first=True) does
    between in_proj and out_proj when called as att(x, x, x) (refine_network.py:56-70 via nn.TransformerEncoderLayer,
    score_network.py:52-53, :84-88).  This pins that formula -- the one tests/test_gpu_parity.py checks the kernel
    against -- to the torch module itself."""
    import torch
    torch.manual_seed(0)
    att = torch.nn.MultiheadAttention(512, 4, batch_first=True).eval()
    x = torch.randn(3, 37, 512)
    with torch.no_grad():
        want, _ = att(x, x, x, need_weights=False)
        qkv = torch.nn.functional.linear(x, att.in_proj_weight, att.in_proj_bias)          # (B, S, [q | k | v])
        B, S, H, hd = 3, 37, 4, 128
        q, k, v = (qkv.reshape(B, S, 3, H, hd)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        ctx = (torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, S, H * hd)
        got = att.out_proj(ctx)
    assert (got - want).abs().max().item() < 1e-5


def test_attention_kernel_lane_bookkeeping_in_numpy():
    """The index scheme of csrc/attention.hip replayed with a numpy model of v_mfma_f32_32x32x16_f16 (A lane l holds
    A[l&31][8(l>>5)+i], B lane l holds B[8(l>>5)+i][l&31], D lane l reg r = D[(r&3)+8(r>>2)+4(l>>5)][l&31]): scores
    computed transposed, probabilities fed to the second product straight from the registers they are born in, V^T read
    with the matching key permutation.  Guards the permutation documented at the top of the kernel."""
    rng = np.random.default_rng(0)
    nq, nk, d = 32, 64, 128                     # one query tile, one 64-key block

    def mfma(a_frag, b_frag, acc):              # a_frag, b_frag: (64 lanes, 8); acc: (64 lanes, 16)
        A = np.zeros((32, 16)); B = np.zeros((16, 32))
        for l in range(64):
            A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a_frag[l]
            B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = b_frag[l]
        D = A @ B
        out = acc.copy()
        for l in range(64):
            for r in range(16):
                out[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
        return out

    Q, K, V = rng.standard_normal((nq, d)), rng.standard_normal((nk, d)), rng.standard_normal((nk, d))
    lanes = np.arange(64); lq, hi = lanes & 31, lanes >> 5
    # S^T tiles: A = K rows (key = 32 sb + lane & 31), B = Q rows (query = lane & 31), k = d
    s = [np.zeros((64, 16)), np.zeros((64, 16))]
    for sb in range(2):
        for kk in range(8):
            kf = np.stack([K[32 * sb + lq[l], 16 * kk + 8 * hi[l]:16 * kk + 8 * hi[l] + 8] for l in range(64)])
            qf = np.stack([Q[lq[l], 16 * kk + 8 * hi[l]:16 * kk + 8 * hi[l] + 8] for l in range(64)])
            s[sb] = mfma(kf, qf, s[sb])
    # lane l, reg r of tile sb is the score of query l&31 against key 32 sb + (r&3) + 8(r>>2) + 4(l>>5)
    S_ref = Q @ K.T
    for sb in range(2):
        for l in (0, 17, 40, 63):
            for r in range(16):
                assert abs(s[sb][l, r] - S_ref[l & 31, 32 * sb + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)]) < 1e-9
    # softmax over the 64 keys of a query = the lane's 32 registers + those of lane l ^ 32
    m = np.maximum(np.max(s[0], 1), np.max(s[1], 1)); m = np.maximum(m, m[lanes ^ 32])
    p = [np.exp(s[0] - m[:, None]), np.exp(s[1] - m[:, None])]
    rs = p[0].sum(1) + p[1].sum(1); rs = rs + rs[lanes ^ 32]
    # O^T += V^T P^T: k-step ks of the block = registers 8(ks&1)..+7 of tile ks>>1; the V^T fragment of lane (d, hi) is
    # keys 16 ks + 4 hi + {0..3} and 16 ks + 8 + 4 hi + {0..3}
    o = [np.zeros((64, 16)) for _ in range(4)]
    for ks in range(4):
        pf = p[ks >> 1][:, 8 * (ks & 1):8 * (ks & 1) + 8]
        for dt in range(4):
            vf = np.zeros((64, 8))
            for l in range(64):
                k0 = 16 * ks + 4 * hi[l]
                keys = list(range(k0, k0 + 4)) + list(range(k0 + 8, k0 + 12))
                vf[l] = V[keys, 32 * dt + lq[l]]
            o[dt] = mfma(vf, pf, o[dt])
    want = (np.exp(S_ref - S_ref.max(1, keepdims=True)) / np.exp(S_ref - S_ref.max(1, keepdims=True)).sum(1, keepdims=True)) @ V
    for dt in range(4):
        for l in range(64):
            for r in range(16):
                dd = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
                assert abs(o[dt][l, r] / rs[l] - want[l & 31, dd]) < 1e-9
