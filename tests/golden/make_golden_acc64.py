"""Generates tests/golden/acc64_chain_golden.npz: the EXACTLY-ROUNDED evaluation of the reference's autocast policy
(oracle/nets_amp.py with ACC64 = True: every reduction accumulated in float64, rounded once to the dtype the policy holds
it in, the policy's own fp16 rounding points kept) on the 252-hypothesis scene of the GPU parity tests (tests/conftest.py).
It is the yardstick of tests/test_gpu_amp.py: the distance of the HIP plan, of the fp32-accumulating oracle and of
PyTorch-ROCm under autocast TO THIS is what the gates bound -- an absolute statement about each implementation, not a
comparison of two noisy ones.  Run in the build container (CPU, ~15 min on 8 cores):

    python tests/golden/make_golden_acc64.py

Stored
  tf_start  (5,252,4,4)  teacher-forced chain, calibrated stand-in weights (|update| ~2 cm / 0.2-0.36 rad): the pose every
                         implementation starts iteration i from (= the exact pose of iteration i-1; tf_start[0] = the grid)
  tf_exact  (5,252,4,4)  exactly-rounded refined pose of iteration i;  tf_trans / tf_rot: the raw network outputs
  tf_crc    (5,2)        crc32 of the network inputs A and B of iteration i (the C oracle's rasteriser / warp on tf_start[i]):
                         the test checks that the GPU box's CPU builds bit-identical inputs before it trusts the yardstick
  tf_exact_fused_it0     iteration 0 with the convolution bias added to the fp32 accumulator (CONV_BIAS = "fused": what ATen's
                         own im2col + GEMM convolution does, i.e. the policy `lib` follows in this image)
  fr_chain  (6,252,4,4)  FREE-RUNNING 5-iteration chain, contraction-scaled heads (weights.CONTRACTION_HEAD_SCALE):
                         fr_chain[i] = poses after i iterations
  score_exact (252,)     ScoreNetMultiPair logits + 100 of fr_chain[5], exactly rounded
"""
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHUNK = 12          # hypotheses per float64 forward (bounds the im2col buffers of the float64 convolution)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def forward_exact(A, B, sd, fn, **kw):
    """nets_amp.<fn> under ACC64 in chunks of hypotheses (every hypothesis is independent up to the scorer's cross attention)"""
    from oracle import nets_amp
    outs = []
    nets_amp.ACC64 = True
    try:
        for a in range(0, A.shape[0], CHUNK):
            outs.append(fn(torch.from_numpy(A[a:a + CHUNK]), torch.from_numpy(B[a:a + CHUNK]), sd, **kw))
    finally:
        nets_amp.ACC64 = False
    if isinstance(outs[0], dict):
        return {k: torch.cat([o[k] for o in outs], 0).numpy() for k in outs[0]}
    return torch.cat(outs, 0)


def refine_exact(cfg, sd, sc, frame, poses):
    """one exactly-rounded refine iteration -> (poses', trans, rot, crcA, crcB)"""
    from oracle import nets_amp, ops as oo
    from oracle import pipeline as op
    A, B, _, _ = op.refine_inputs(cfg, poses, sc["mesh_np"], sc["rgb"], frame["xyz"], sc["K"], sc["diameter"])
    o = forward_exact(A, B, sd, nets_amp.refine_forward)
    tn = [float(v) for v in cfg["trans_normalizer"]]
    new = oo.pose_update(o["trans"], o["rot"], poses, cfg["rot_rep"], bool(cfg["normalize_xyz"]), tn, float(cfg["rot_normalizer"]),
                         float(sc["diameter"]))
    return new, o["trans"], o["rot"], crc(A), crc(B)


def main():
    from conftest import _build_scene
    from foundationpose_amd.weights import CONTRACTION_HEAD_SCALE, DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    from oracle import nets_amp, ops as oo
    from oracle import pipeline as op
    torch.set_num_threads(os.cpu_count() or 8)
    sc = _build_scene()
    d = op.preprocess_depth(sc["depth"])
    frame = dict(depth_f=d, xyz=oo.depth2xyzmap(d, sc["K"], f64_internal=True))
    cfg = dict(DEFAULT_REFINE_CFG)
    out = {}
    t0 = time.time()
    # ---- teacher-forced chain, calibrated weights
    sd = random_state_dict("refine", cfg, seed=0)
    P = sc["poses"].astype(np.float32)
    starts, exact, tr, ro, crcs = [], [], [], [], []
    for it in range(5):
        starts.append(P.copy())
        P, t_, r_, ca, cb = refine_exact(cfg, sd, sc, frame, P)
        exact.append(P.copy()); tr.append(t_); ro.append(r_); crcs.append((ca, cb))
        print(f"teacher-forced iteration {it}: {time.time() - t0:.0f} s", flush=True)
    out.update(tf_start=np.stack(starts), tf_exact=np.stack(exact), tf_trans=np.stack(tr), tf_rot=np.stack(ro),
               tf_crc=np.asarray(crcs, dtype=np.uint32))
    nets_amp.CONV_BIAS = "fused"
    try:
        out["tf_exact_fused_it0"] = refine_exact(cfg, sd, sc, frame, starts[0])[0]
    finally:
        nets_amp.CONV_BIAS = "separate"
    print(f"fused-bias iteration 0: {time.time() - t0:.0f} s", flush=True)
    # ---- free-running chain, contraction-scaled heads
    sdc = random_state_dict("refine", cfg, seed=0, head_scale=CONTRACTION_HEAD_SCALE)
    P = sc["poses"].astype(np.float32)
    chain = [P.copy()]
    for it in range(5):
        P = refine_exact(cfg, sdc, sc, frame, P)[0]
        chain.append(P.copy())
        print(f"free-running iteration {it}: {time.time() - t0:.0f} s", flush=True)
    out["fr_chain"] = np.stack(chain)
    # ---- scores of the free-running chain's result
    scfg = dict(DEFAULT_SCORE_CFG)
    ssd = random_state_dict("score", scfg, seed=0)
    A, B, _, _ = op.score_inputs(scfg, chain[-1], sc["mesh_np"], sc["rgb"], frame["depth_f"], sc["K"], sc["diameter"])
    feats = forward_exact(A, B, ssd, nets_amp.score_features)              # per hypothesis
    nets_amp.ACC64 = True
    try:
        x = nets_amp.mha(feats.reshape(1, feats.shape[0], -1), ssd, "att_cross", explicit=True)
        logit = nets_amp._linear(x, ssd["linear.weight"], ssd["linear.bias"]).reshape(-1)
    finally:
        nets_amp.ACC64 = False
    out["score_exact"] = logit.numpy() + 100.0
    out["score_crc"] = np.asarray([crc(A), crc(B)], dtype=np.uint32)
    out["head_scale"] = np.float64(CONTRACTION_HEAD_SCALE)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "acc64_chain_golden.npz"), **out)
    print(f"done in {time.time() - t0:.0f} s; keys: {sorted(out)}")


if __name__ == "__main__":
    main()
