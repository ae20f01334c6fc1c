"""Trains the stand-in RefineNet into what the reference's released checkpoint is: a CONTRACTION towards the observed pose
(round-5 verdict, item 1).  The reference's weights (readme.md:61) cannot be had here, and a random convolutional trunk turns a 1e-6
pose difference into other features (the free-running chain of ANY fp32-accumulating implementation leaves the
exactly-rounded one x 30 per iteration).  A trained network does not; this script makes one for the synthetic scene of the parity tests:

    python tests/golden/train_standin_refiner.py [--seconds 480] [--batch 48]       # on an MI355X box (gpurun), ~10 GPU-minutes

  model    the product's nn.Module (foundationpose_amd/refine_network.py = learning/models/refine_network.py:26-93), started from the
           seeded stand-in checkpoint (weights.random_state_dict('refine', seed 0)), under torch.autocast(float16) + GradScaler --
           the deployed arithmetic (predict_pose_refine.py:190-191) on PyTorch-ROCm, autograd only
  data     seeded perturbations of the scene's ground-truth pose (multi-scale: uniform up to 16 deg / 3 cm mixed with log-uniform
           down to 0.016 deg / 0.03 mm, so that the map keeps contracting near its fixed point), network inputs from the product's own
           fp_crop_windows / fp_render_crops / fp_warp_crops
  target   the normalised delta the pose update applies (predict_pose_refine.py:195-234; trans_rep 'tracknet' + normalize_xyz:
           dt / (diameter / 2); rot_rep 'axis_angle': tanh(rot) * rot_normalizer = log(R R_gt^T)), L1 loss in that space
  output   foundationpose_amd/data/standin_trained_refiner.npz: the state_dict with every matrix / kernel held in float16 (what
           autocast casts it to anyway; vectors stay float32), + gpurun_out/train_standin_refiner.json: loss curve, held-out contraction
           per iteration, and a first hip / torch_amp comparison of the free-running chain
The checkpoint is data; tests/golden/make_golden_acc64_trained.py mints the exactly-rounded chain for it on the CPU.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

OUT = os.path.join(ROOT, "foundationpose_amd", "data", "standin_trained_refiner.npz")
MAX_ROT_DEG, MAX_TRANS = 16.0, 0.03      # rot_normalizer is 20 deg: the tanh target stays <= 0.8


def sample_poses(gt, n, rng):
    """n perturbations of gt (4,4): half uniform in magnitude up to (MAX_ROT_DEG, MAX_TRANS), half log-uniform over three decades"""
    out = np.tile(np.asarray(gt, dtype=np.float64)[None], (n, 1, 1))
    for i in range(n):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        u, v = rng.uniform(), rng.uniform()
        fr = u if rng.uniform() < 0.5 else 10.0 ** (-3.0 * u)
        ft = v if rng.uniform() < 0.5 else 10.0 ** (-3.0 * v)
        ang = np.deg2rad(MAX_ROT_DEG) * fr
        Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        out[i, :3, :3] = (np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx) @ out[i, :3, :3]
        d = rng.normal(size=3)
        out[i, :3, 3] += d / np.linalg.norm(d) * MAX_TRANS * ft
    return out.astype(np.float32)


def so3_log(R):
    c = np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1, 1)
    th = np.arccos(c)
    v = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], 1)
    s = np.sin(th)
    k = np.where(s > 1e-9, th / (2 * np.maximum(s, 1e-9)), 0.5)
    return v * k[:, None]


def targets(cfg, poses, gt, diameter):
    """-> (trans target, tanh(rot) target) in the network's output space"""
    P, G = np.asarray(poses, dtype=np.float64), np.asarray(gt, dtype=np.float64)
    yt = (G[None, :3, 3] - P[:, :3, 3]) / (diameter / 2)
    w = so3_log(P[:, :3, :3] @ G[:3, :3].T[None])
    return yt.astype(np.float32), np.clip(w / float(cfg["rot_normalizer"]), -0.999, 0.999).astype(np.float32)


def pose_err(P, gt):
    from amp_util import geodesic
    G = np.tile(np.asarray(gt)[None], (len(P), 1, 1))
    return geodesic(P[:, :3, :3], G[:, :3, :3]), np.linalg.norm(P[:, :3, 3].astype(np.float64) - G[:, :3, 3], axis=1)


def pct(x):
    x = np.asarray(x, dtype=np.float64)
    return dict(median=float(np.median(x)), p90=float(np.percentile(x, 90)), max=float(x.max()))


def packed_state_dict(sd):
    """matrices / kernels in float16 (autocast's cast, made once), vectors and buffers in float32"""
    out = {}
    for k, v in sd.items():
        v = v.detach().cpu()
        if not v.dtype.is_floating_point:
            out[k] = v.numpy()
        elif v.dim() >= 2 and not k.endswith("pos_embed.pe"):
            out[k] = v.to(torch.float16).numpy()
        else:
            out[k] = v.float().numpy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=480.0, help="training budget (wall clock on the GPU box)")
    ap.add_argument("--batch", type=int, default=48)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--bn-train-frac", type=float, default=0.55, help="share of the budget with BatchNorm in training mode; then the running "
                    "statistics are re-estimated as plain averages over --bn-batches batches and frozen (the deployed network is eval-mode)")
    ap.add_argument("--bn-batches", type=int, default=150)
    ap.add_argument("--head-init", type=float, default=0.05, help="scale of the two output Linears at the start (the stand-in's random "
                    "heads saturate the tanh of the rotation update: no gradient)")
    ap.add_argument("--out", default=OUT)
    ap.add_argument("--report", default=os.path.join(ROOT, "gpurun_out", "train_standin_refiner.json"))
    args = ap.parse_args()
    from conftest import _build_scene
    from foundationpose_amd import ops
    from foundationpose_amd.Utils import get_mesh_handle, make_mesh_tensors
    from foundationpose_amd.engine import _conv_backend
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.refine_network import RefineNet
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    from make_golden_acc64_fitted import start_poses
    from oracle import ops as oo
    from oracle import pipeline as op
    assert torch.cuda.is_available(), "training needs the GPU box"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    sc = _build_scene()
    d = op.preprocess_depth(sc["depth"])
    xyz = oo.depth2xyzmap(d, sc["K"], f64_internal=True)
    cfg = dict(DEFAULT_REFINE_CFG)
    K, diam, gt = sc["K"], float(sc["diameter"]), sc["gt"]
    gm = make_mesh_tensors(sc["mesh"], device=dev)
    handle = get_mesh_handle(gm)
    rgb_t = torch.as_tensor(sc["rgb"], device=dev).float().contiguous()
    xyz_t = torch.as_tensor(xyz, device=dev, dtype=torch.float).contiguous()
    H, W = sc["H"], sc["W"]
    oh, ow = cfg["input_resize"]

    def inputs(poses):
        """network inputs of the product's own kernels, (2n,6,oh,ow) float32 (autocast rounds them as the fp16 plan's buffer does)"""
        P = torch.as_tensor(poses, device=dev, dtype=torch.float).contiguous()
        n = P.shape[0]
        AB = torch.empty((2 * n, 6, oh, ow), dtype=torch.float32, device=dev)
        tf, bb = ops.crop_windows(P, K, diam, cfg["crop_ratio"], (ow, oh))
        ops.render_crops(handle, P, bb, K, H, W, out_hw=(oh, ow), mesh_diameter=diam, xyz_thr=0.001, normalize_xyz=True, A_out=AB[:n])
        ops.warp_crops(rgb_t, xyz_t, None, tf, K, P, diam, ops.MODE_REFINE, normalize_xyz=True, out_hw=(oh, ow), B_out=AB[n:])
        return AB

    net = RefineNet(cfg=cfg, c_in=6)
    sd0 = random_state_dict("refine", cfg, seed=0)
    for k in ("trans_head.1.weight", "trans_head.1.bias", "rot_head.1.weight", "rot_head.1.bias"):
        sd0[k] = sd0[k] * args.head_init
    net.load_state_dict(sd0)
    net.to(dev)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=args.lr, weight_decay=1e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    rng = np.random.default_rng(20260930)
    val = start_poses(gt, 252)                                           # the parity chain's own starts: <= 15 deg / 2 cm, seed 777
    held = sample_poses(gt, 256, np.random.default_rng(99))

    def one_iteration(poses):
        net.eval()
        with torch.no_grad(), _conv_backend(), torch.autocast("cuda", dtype=torch.float16):
            outs = []
            for a in range(0, len(poses), 64):
                AB = inputs(poses[a:a + 64])
                n = AB.shape[0] // 2
                o = net(AB[:n], AB[n:])
                outs.append({k: v.float() for k, v in o.items()})
        raw = {k: torch.cat([o[k] for o in outs]) for k in outs[0]}
        P = torch.as_tensor(poses, device=dev, dtype=torch.float).contiguous()
        tn = [float(v) for v in cfg["trans_normalizer"]]
        return ops.pose_update(raw["trans"], raw["rot"], P, rot_rep=cfg["rot_rep"], normalize_xyz=True, trans_normalizer=tn,
                               rot_normalizer=float(cfg["rot_normalizer"]), mesh_diameter=diam).cpu().numpy()

    def batch(n):
        poses = sample_poses(gt, n, rng)
        yt, yr = targets(cfg, poses, gt, diam)
        return inputs(poses), torch.as_tensor(yt, device=dev), torch.as_tensor(yr, device=dev)

    def set_mode(bn_eval):
        net.train()
        if bn_eval:
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.eval()

    def reestimate_bn(n_batches):
        """running statistics = plain averages of the batch statistics over n_batches training batches (momentum None), no gradient"""
        bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        for m in bns:
            m.reset_running_stats()
            m.momentum = None
        net.train()
        with torch.no_grad(), _conv_backend(), torch.autocast("cuda", dtype=torch.float16):
            for _ in range(n_batches):
                AB, _, _ = batch(args.batch)
                net(AB[:args.batch], AB[args.batch:])
        for m in bns:
            m.momentum = 0.1

    log, t0, step = [], time.time(), 0
    bn_eval, bad, good = False, 0, None
    while True:
        el = time.time() - t0
        if el >= args.seconds:
            break
        prog = el / args.seconds
        lr = args.lr * min(1.0, (step + 1) / 300.0) * (0.02 + 0.98 * 0.5 * (1 + np.cos(np.pi * prog)))
        for g in opt.param_groups:
            g["lr"] = lr
        if prog >= args.bn_train_frac and not bn_eval:
            reestimate_bn(args.bn_batches)
            bn_eval = True
            e0, e1 = pose_err(held, gt), pose_err(one_iteration(held), gt)
            print(json.dumps(dict(event="BatchNorm frozen", step=step, held_dR_ratio=float(np.median(e1[0]) / np.median(e0[0])),
                                  held_dt_ratio=float(np.median(e1[1]) / np.median(e0[1])))), flush=True)
        set_mode(bn_eval)
        AB, yt, yr = batch(args.batch)
        n = args.batch
        with _conv_backend(), torch.autocast("cuda", dtype=torch.float16):
            o = net(AB[:n], AB[n:])
        lt = (o["trans"].float() - yt).abs().mean()
        lr_ = (torch.tanh(o["rot"].float()) - yr).abs().mean()
        loss = lt + lr_
        if not bool(torch.isfinite(loss)):
            # an fp16 overflow in the forward pass: no step; if it persists, back to the last snapshot that evaluated finite
            bad += 1
            opt.zero_grad(set_to_none=True)
            if bad >= 20 and good is not None:
                net.load_state_dict(good)
                opt.state.clear()
                bad = 0
                print(json.dumps(dict(event="restored the last finite snapshot", step=step)), flush=True)
            continue
        bad = 0
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        scaler.step(opt)
        scaler.update()
        step += 1
        if step % 50 == 0 or step == 1:
            row = dict(step=step, seconds=round(time.time() - t0, 1), lr=float(lr), loss_trans=float(lt), loss_rot=float(lr_), bn_eval=bn_eval)
            if step % 500 == 0:
                e0, e1 = pose_err(held, gt), pose_err(one_iteration(held), gt)
                row.update(held_dR_ratio=float(np.median(e1[0]) / np.median(e0[0])), held_dt_ratio=float(np.median(e1[1]) / np.median(e0[1])))
                if bn_eval and np.isfinite(row["held_dR_ratio"]) and np.isfinite(row["held_dt_ratio"]):
                    good = {k: v.detach().clone() for k, v in net.state_dict().items()}
            log.append(row)
            print(json.dumps(row), flush=True)
    if not bn_eval:
        reestimate_bn(args.bn_batches)
    train_s = time.time() - t0
    # ---- the checkpoint: matrices rounded to float16 once, reloaded so that everything below runs on the shipped values
    packed = packed_state_dict(net.state_dict())
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    np.savez(args.out, **packed)
    sha = hashlib.sha256(open(args.out, "rb").read()).hexdigest()
    from foundationpose_amd.weights import trained_refiner_state_dict
    sd = trained_refiner_state_dict(args.out)
    net.load_state_dict(sd)
    net.eval()
    rep = dict(seconds=train_s, steps=step, batch=args.batch, sha256=sha, bytes=os.path.getsize(args.out), log=log)
    # ---- held-out contraction, iteration by iteration (torch_amp module), on the parity chain's 252 starts
    chain = [val]
    for _ in range(5):
        chain.append(one_iteration(chain[-1]))
    rep["lib_chain_error_to_gt"] = [dict(dR=pct(pose_err(P, gt)[0]), dt=pct(pose_err(P, gt)[1])) for P in chain]
    e0, e1 = pose_err(chain[0], gt), pose_err(chain[1], gt)
    rep["first_iteration_shrinks"] = dict(dR=float(np.median(e0[0]) / np.median(e1[0])), dt=float(np.median(e0[1]) / np.median(e1[1])))
    os.makedirs(os.path.dirname(args.report), exist_ok=True)
    json.dump(rep, open(args.report, "w"), indent=1)
    np.save(os.path.join(os.path.dirname(args.report), "train_standin_lib_chain.npy"), np.stack(chain))
    # ---- first look at the deployed kernels: free-running hip chain against the torch_amp chain, and hip against itself from the
    # torch chain's poses (teacher-forced)
    pred = PoseRefinePredictor(cfg=cfg, state_dict=sd, device=dev, precision="fp16")
    kw = dict(mesh=sc["mesh"], mesh_tensors=gm, mesh_diameter=diam)
    depth_t = torch.as_tensor(d, device=dev)

    def hip(P, it):
        o, _ = pred.predict(sc["rgb"], depth_t, K, P, xyz_t, iteration=it, **kw)
        return o.cpu().numpy()
    from amp_util import geodesic
    dist = lambda P, Q: (geodesic(P[:, :3, :3], Q[:, :3, :3]), np.linalg.norm(P[:, :3, 3].astype(np.float64) - Q[:, :3, 3], axis=1))
    h5 = hip(val, 5)
    rep["hip_free_running_vs_lib"] = dict(dR=pct(dist(h5, chain[5])[0]), dt=pct(dist(h5, chain[5])[1]))
    rep["hip_chain_error_to_gt"] = dict(dR=pct(pose_err(h5, gt)[0]), dt=pct(pose_err(h5, gt)[1]))
    tf = [dist(hip(chain[k], 1), chain[k + 1]) for k in range(5)]
    rep["hip_teacher_forced_vs_lib"] = [dict(dR=pct(a), dt=pct(b)) for a, b in tf]
    os.makedirs(os.path.dirname(args.report), exist_ok=True)
    json.dump(rep, open(args.report, "w"), indent=1)
    print(json.dumps({k: v for k, v in rep.items() if k != "log"}, indent=1))


if __name__ == "__main__":
    main()
