"""CPU: the exactly-rounded yardstick of the GPU parity gates (tests/golden/acc64_chain_golden.npz, made by make_golden_acc64.py
from oracle/nets_amp.py with ACC64 = True) is what this machine computes, its chains are consistent, and the fp32-accumulating
oracle sits at the distance from it that the gates assume."""
import os
import zlib

import numpy as np
import pytest
import torch

from amp_util import geodesic, ulp16
from conftest import ROOT


@pytest.fixture(scope="module")
def G():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "acc64_chain_golden.npz")))


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def test_golden_is_self_consistent(scene, G):
    assert G["tf_start"].shape == (5, 252, 4, 4) and G["tf_exact"].shape == (5, 252, 4, 4) and G["fr_chain"].shape == (6, 252, 4, 4)
    assert np.array_equal(G["tf_start"][0], scene["poses"]) and np.array_equal(G["fr_chain"][0], scene["poses"])
    for it in range(4):                                    # teacher forcing: iteration i+1 starts from the exact pose of iteration i
        assert np.array_equal(G["tf_start"][it + 1], G["tf_exact"][it])
    for k in ("tf_trans", "tf_rot"):                       # the reference holds the raw outputs in fp16
        assert np.array_equal(G[k], G[k].astype(np.float16).astype(np.float32))
    upd = geodesic(G["tf_exact"][0][:, :3, :3], G["tf_start"][0][:, :3, :3])
    assert np.median(upd) > 0.05                           # full-size updates
    mot = geodesic(G["fr_chain"][5][:, :3, :3], G["fr_chain"][0][:, :3, :3])
    assert 1e-3 < np.median(mot) < 1e-2                    # contraction-scaled chain
    assert G["score_exact"].shape == (252,) and np.isfinite(G["score_exact"]).all()


def test_yardstick_reproduces_here_and_fp32_oracle_is_close(scene, G):
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    from oracle import nets_amp, ops as oo
    from oracle import pipeline as op
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, seed=0)
    d = op.preprocess_depth(scene["depth"])
    xyz = oo.depth2xyzmap(d, scene["K"], f64_internal=True)
    n = 6
    A, B, _, _ = op.refine_inputs(cfg, G["tf_start"][0], scene["mesh_np"], scene["rgb"], xyz, scene["K"], scene["diameter"])
    assert (_crc(A), _crc(B)) == tuple(int(v) for v in G["tf_crc"][0])          # same network inputs as the minting run
    A, B = torch.from_numpy(A[:n]), torch.from_numpy(B[:n])
    assert nets_amp.ACC64 is False
    nets_amp.ACC64 = True
    try:
        o64 = nets_amp.refine_forward(A, B, sd)
    finally:
        nets_amp.ACC64 = False
    o32 = nets_amp.refine_forward(A, B, sd)
    for k, g in (("trans", G["tf_trans"][0][:n]), ("rot", G["tf_rot"][0][:n])):
        d64 = np.abs(o64[k].numpy() - g)
        assert (d64 <= ulp16(g)).all() and np.mean(d64 == 0) >= 0.9, (k, d64.max())
        # fp32 accumulation: a few fp16 ulps of the raw output away from the exactly-rounded one, never far
        d32 = np.abs(o32[k].numpy() - g)
        assert d32.max() <= 16 * ulp16(np.abs(g).max()), (k, d32.max())


def test_fitted_heads_golden_reproduces_here_and_refines(scene):
    """round 5: tests/golden/acc64_fitted_chain_golden.npz (make_golden_acc64_fitted.py; the stand-in refiner with heads fitted by
    tests/golden/fit_contraction_heads.py) -- the starts are the seeded perturbations, the first iteration's exactly-rounded pose is what
    this machine computes for a few hypotheses, the stand-in REFINES (error to the ground truth shrinks), and the golden's own
    fp32-accumulating oracle chain documents the amplification of a free-running chain (the reason its 1e-4 gate stays with the
    contraction-scaled heads)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_acc64_fitted import start_poses
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    from oracle import nets_amp, ops as oo
    from oracle import pipeline as op
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "acc64_fitted_chain_golden.npz")))
    P0 = start_poses(scene["gt"])
    assert np.array_equal(P0, g["start"]) and np.array_equal(g["chain"][0], P0) and np.array_equal(g["oracle_chain"][0], P0)
    assert g["chain"].shape == (6, 252, 4, 4) and np.isfinite(g["chain"]).all()
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, seed=0, heads="fitted")
    ref = random_state_dict("refine", cfg, seed=0)
    for k in sd:                                           # only the two output heads differ from the calibrated seed-0 checkpoint
        same = torch.equal(sd[k], ref[k])
        assert same != k.startswith(("trans_head.1.", "rot_head.1.")), k
    assert float(sd["rot_head.1.weight"].norm()) < 0.5 * float(ref["rot_head.1.weight"].norm())
    d = op.preprocess_depth(scene["depth"])
    xyz = oo.depth2xyzmap(d, scene["K"], f64_internal=True)
    A, B, _, _ = op.refine_inputs(cfg, P0, scene["mesh_np"], scene["rgb"], xyz, scene["K"], scene["diameter"])
    assert (_crc(A), _crc(B)) == tuple(int(v) for v in g["crc"][0])
    n = 4
    nets_amp.ACC64 = True
    try:
        o = nets_amp.refine_forward(torch.from_numpy(A[:n]), torch.from_numpy(B[:n]), sd)
    finally:
        nets_amp.ACC64 = False
    tn = [float(v) for v in cfg["trans_normalizer"]]
    new = oo.pose_update(o["trans"].numpy(), o["rot"].numpy(), P0[:n], cfg["rot_rep"], bool(cfg["normalize_xyz"]), tn, float(cfg["rot_normalizer"]),
                         float(scene["diameter"]))
    assert np.abs(new - g["chain"][1][:n]).max() <= 1e-6          # exactly rounded = reproducible (up to libm ulps in the pose update)
    # it refines: one iteration brings the translation error to the ground truth under half, the rotation error down
    G = np.tile(scene["gt"][None], (252, 1, 1))
    e = lambda P: (geodesic(P[:, :3, :3], G[:, :3, :3]), np.linalg.norm(P[:, :3, 3] - G[:, :3, 3], axis=1))
    e0, e1 = e(g["chain"][0]), e(g["chain"][1])
    assert np.median(e1[1]) < 0.5 * np.median(e0[1]) and np.median(e1[0]) < 0.9 * np.median(e0[0])
    upd = geodesic(g["chain"][1][:, :3, :3], g["chain"][0][:, :3, :3])
    assert np.median(upd) > 0.04                                   # full-size, directed updates
    # one iteration: the fp32-accumulating oracle is at the fp16 policy's noise floor; free running it leaves the exact chain
    d1 = geodesic(g["oracle_chain"][1][:, :3, :3], g["chain"][1][:, :3, :3])
    d5 = geodesic(g["oracle_chain"][5][:, :3, :3], g["chain"][5][:, :3, :3])
    assert np.median(d1) < 6e-4 and np.median(d5) > 20 * np.median(d1)


def test_trained_standin_golden_reproduces_here_and_contracts(scene):
    """round 6: tests/golden/acc64_trained_chain_golden.npz (make_golden_acc64_trained.py) was minted for the shipped checkpoint of the
    TRAINED stand-in refiner (weights.trained_refiner_state_dict; recipe tests/golden/train_standin_refiner.py), its first iteration's
    exactly-rounded raw outputs are what this machine computes, the network IS a contraction with full-size updates (one iteration
    shrinks the pose error by far more than the 3 x the round-5 verdict asked for), and the golden's own fp32-accumulating oracle chain
    says what a free-running implementation of the fp16 policy can be held to: 1e-4 m for every hypothesis, 1e-4 rad for the bulk."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_acc64_fitted import start_poses
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, TRAINED_REFINER_FILE, trained_refiner_state_dict
    from oracle import nets_amp, ops as oo
    from oracle import pipeline as op
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "acc64_trained_chain_golden.npz")))
    assert hashlib.sha256(open(TRAINED_REFINER_FILE, "rb").read()).hexdigest() == str(g["checkpoint_sha256"])
    sd = trained_refiner_state_dict()
    for k, v in sd.items():                                 # matrices and kernels are fp16-valued: autocast's cast of them is exact
        if v.dtype.is_floating_point and v.dim() >= 2 and not k.endswith("pos_embed.pe"):
            assert torch.equal(v, v.half().float()), k
    P0 = start_poses(scene["gt"])
    assert np.array_equal(P0, g["start"]) and np.array_equal(g["chain"][0], P0) and g["chain"].shape == (6, 252, 4, 4)
    cfg = dict(DEFAULT_REFINE_CFG)
    d = op.preprocess_depth(scene["depth"])
    xyz = oo.depth2xyzmap(d, scene["K"], f64_internal=True)
    A, B, _, _ = op.refine_inputs(cfg, P0, scene["mesh_np"], scene["rgb"], xyz, scene["K"], scene["diameter"])
    assert (_crc(A), _crc(B)) == tuple(int(v) for v in g["crc"][0])
    n = 4
    nets_amp.ACC64 = True
    try:
        o = nets_amp.refine_forward(torch.from_numpy(A[:n]), torch.from_numpy(B[:n]), sd)
    finally:
        nets_amp.ACC64 = False
    for k, ref in (("trans", g["raw_trans"][0][:n]), ("rot", g["raw_rot"][0][:n])):
        dd = np.abs(o[k].numpy() - ref)
        assert (dd <= ulp16(ref)).all() and np.mean(dd == 0) >= 0.9, (k, dd.max())
    G = np.tile(scene["gt"][None], (252, 1, 1))
    e = lambda P: (geodesic(P[:, :3, :3], G[:, :3, :3]), np.linalg.norm(P[:, :3, 3] - G[:, :3, 3], axis=1))
    e0, e1, e5 = e(g["chain"][0]), e(g["chain"][1]), e(g["chain"][5])
    assert np.median(e0[0]) > 0.1 and np.median(e0[1]) > 0.01                            # starts 7.5 deg / 1.2 cm off (median)
    assert np.median(e1[0]) * 30 < np.median(e0[0]) and np.median(e1[1]) * 30 < np.median(e0[1])   # one iteration: > 30 x in both
    assert np.median(e5[0]) < 5e-4 and np.median(e5[1]) < 1.5e-4
    # the fp32-accumulating oracle's free-running chain against the exact one: the floor of ANY implementation of the policy
    dR5 = geodesic(g["oracle_chain"][5][:, :3, :3], g["chain"][5][:, :3, :3])
    dt5 = np.linalg.norm(g["oracle_chain"][5][:, :3, 3].astype(np.float64) - g["chain"][5][:, :3, 3], axis=1)
    assert dt5.max() < 1e-4 and np.median(dR5) < 1e-4 and 0.6 < np.mean(dR5 <= 1e-4) < 1.0 and dR5.max() < 1e-3
