"""Fits the two output heads (Linear(512 -> 3) of trans_head / rot_head, refine_network.py:56-70) of the stand-in RefineNet by ridge
regression, so that the stand-in is what a trained refiner is: a CONTRACTION towards the observed pose with full-size first updates
(rotation up to rot_normalizer = 20 deg, translation up to ~2 cm) -- round-4 verdict, item 5.  Everything below the heads stays the
seeded random-init network (weights.random_state_dict, calibrated BatchNorm statistics): its pooled transformer features are random
features of the (rendered, observed) crop pair, and a linear read-out of 512 of them regresses the pose error well enough to shrink it.

    python tests/golden/fit_contraction_heads.py            # ~10 min on 8 cores -> foundationpose_amd/data/standin_fitted_heads.npz

Training set: perturbations of the scene's ground-truth pose (tests/conftest.py), features from the CPU oracle in the deployed
arithmetic (oracle/nets_amp.py, fp32 accumulation).  Targets in the network's own output space (predict_pose_refine.py:195-234):
  trans: (t_gt - t) / (diameter / 2)                       (trans_rep 'tracknet', normalize_xyz: no tanh)
  rot:   atanh(clip(w / rot_normalizer)),  exp(w) = R R_gt^T (rot_rep 'axis_angle': R' = so3_exp_map(w)^T R)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

OUT = os.path.join(ROOT, "foundationpose_amd", "data", "standin_fitted_heads.npz")
N_TRAIN, N_VAL = 1536, 256
# Tikhonov weight.  The table this script prints is the trade-off: a small lambda fits the held-out translation to 21 % residual but
# needs |W_rot| = 23 (the output noise of ANY implementation of the fp16 policy is |W| x the feature noise its rounding flips cause:
# measured 6e-4 rad at the calibrated random heads' |W_rot| = 5.6); lambda = 3 still shrinks the held-out translation error to 33 % and
# the rotation error to 79 % (median) per iteration at |W_trans| = 0.34, |W_rot| = 1.4 -- a genuine, if modest, contraction with a
# quarter of the random heads' noise gain.
LAMBDA = 3.0
MAX_ROT_DEG, MAX_TRANS = 20.0, 0.03
CHUNK = 32


def so3_log(R):
    """rotation vector of R (n,3,3), float64"""
    c = np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1, 1)
    th = np.arccos(c)
    v = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], 1)
    s = np.sin(th)
    k = np.where(s > 1e-9, th / (2 * np.maximum(s, 1e-9)), 0.5)
    return v * k[:, None]


def targets(cfg, poses, gt, diameter):
    P = np.asarray(poses, dtype=np.float64)
    G = np.asarray(gt, dtype=np.float64)
    yt = (G[None, :3, 3] - P[:, :3, 3]) / (diameter / 2)
    w = so3_log(P[:, :3, :3] @ G[:3, :3].T[None])
    yr = np.arctanh(np.clip(w / float(cfg["rot_normalizer"]), -0.95, 0.95))
    return yt, yr


def features(cfg, sd, sc, frame, poses):
    """pooled transformer features of both heads, (n, 512) each: mean over the 400 tokens of the encoder layer's output (the Linear
    that follows commutes with the mean, refine_network.py:90-91)"""
    from oracle import nets_amp
    from oracle import pipeline as op
    ft, fr = [], []
    for a in range(0, len(poses), CHUNK):
        A, B, _, _ = op.refine_inputs(cfg, poses[a:a + CHUNK], sc["mesh_np"], sc["rgb"], frame["xyz"], sc["K"], sc["diameter"])
        with torch.no_grad():
            tok = nets_amp.encoder_tokens(torch.from_numpy(A), torch.from_numpy(B), sd, "encodeA", "encodeAB")
            ft.append(nets_amp.encoder_layer(tok, sd, "trans_head.0").float().mean(dim=1).numpy())
            fr.append(nets_amp.encoder_layer(tok, sd, "rot_head.0").float().mean(dim=1).numpy())
    return np.concatenate(ft), np.concatenate(fr)


def ridge(F, Y, lam):
    """-> W (out, 512), b (out): least squares on centred features with Tikhonov weight lam * trace(F^T F) / 512"""
    mu, my = F.mean(0), Y.mean(0)
    Fc, Yc = F - mu, Y - my
    G = Fc.T @ Fc
    W = np.linalg.solve(G + lam * np.trace(G) / G.shape[0] * np.eye(G.shape[0]), Fc.T @ Yc).T
    return W, my - W @ mu


def main():
    from conftest import _build_scene
    from foundationpose_amd import synthetic as syn
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    from oracle import ops as oo
    from oracle import pipeline as op
    torch.set_num_threads(os.cpu_count() or 8)
    sc = _build_scene()
    d = op.preprocess_depth(sc["depth"])
    frame = dict(xyz=oo.depth2xyzmap(d, sc["K"], f64_internal=True))
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, seed=0)
    t0 = time.time()
    P = syn.perturbed_poses(sc["gt"], N_TRAIN + N_VAL, seed=4242, max_trans=MAX_TRANS, max_rot_deg=MAX_ROT_DEG).astype(np.float32)
    cache = "/tmp/fit_features.npz"
    if os.path.exists(cache) and np.array_equal(np.load(cache)["P"], P):
        Ft, Fr = np.load(cache)["Ft"], np.load(cache)["Fr"]
    else:
        Ft, Fr = features(cfg, sd, sc, frame, P)
    print(f"features of {len(P)} poses: {time.time() - t0:.0f} s", flush=True)
    yt, yr = targets(cfg, P, sc["gt"], sc["diameter"])
    tr, va = slice(0, N_TRAIN), slice(N_TRAIN, None)
    best = None
    for lam in (1e-2, 1e-1, 0.3, 1.0, LAMBDA, 10.0, 30.0):
        Wt, bt = ridge(Ft[tr].astype(np.float64), yt[tr], lam)
        Wr, br = ridge(Fr[tr].astype(np.float64), yr[tr], lam)
        et = np.linalg.norm(Ft[va] @ Wt.T + bt - yt[va], axis=1) / np.maximum(np.linalg.norm(yt[va], axis=1), 1e-9)
        er = np.linalg.norm(np.tanh(Fr[va] @ Wr.T + br) - np.tanh(yr[va]), axis=1) / np.maximum(np.linalg.norm(np.tanh(yr[va]), axis=1), 1e-9)
        print(f"lambda {lam:g}: held-out residual / error  trans median {np.median(et):.3f} p90 {np.percentile(et, 90):.3f}   "
              f"rot median {np.median(er):.3f} p90 {np.percentile(er, 90):.3f}   |Wt| {np.linalg.norm(Wt):.1f} |Wr| {np.linalg.norm(Wr):.1f}", flush=True)
        if lam == LAMBDA:
            best = (0.0, lam, Wt, bt, Wr, br)
    _, lam, Wt, bt, Wr, br = best
    np.savez_compressed(OUT, **{"trans_head.1.weight": Wt.astype(np.float32), "trans_head.1.bias": bt.astype(np.float32),
                                "rot_head.1.weight": Wr.astype(np.float32), "rot_head.1.bias": br.astype(np.float32),
                                "lambda": np.float64(lam), "n_train": np.int64(N_TRAIN), "max_rot_deg": np.float64(MAX_ROT_DEG),
                                "max_trans": np.float64(MAX_TRANS)})
    np.savez_compressed(cache, Ft=Ft, Fr=Fr, P=P, yt=yt, yr=yr)
    print(f"chosen lambda {lam:g}; wrote {OUT}; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
