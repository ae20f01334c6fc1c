"""Deterministic stand-in checkpoints.  The reference's weights/ directory is not shipped,
so tests / bench / golden vectors use a procedurally generated state_dict: every tensor is drawn from a
CPU generator seeded by (seed, crc32(key)), which is stable across processes, machines and torch builds.
Key names and shapes are those of the reference modules (SURVEY.md 8(c) 'State-dict compatibility')."""
import math
import os
import zlib

import torch

DEFAULT_REFINE_CFG = dict(use_BN=True, c_in=6, input_resize=[160, 160], crop_ratio=1.2, normalize_xyz=True,
                          rot_rep="axis_angle", trans_rep="tracknet", trans_normalizer=[0.019999999552965164, 0.019999999552965164, 0.05000000074505806],
                          rot_normalizer=0.3490658503988659, use_normal=False, use_mask=False, n_view=1,
                          zfar=float("inf"), normal_uint8=False)
DEFAULT_SCORE_CFG = dict(use_BN=True, c_in=6, input_resize=[160, 160], crop_ratio=1.1, normalize_xyz=True,
                         use_normal=False, zfar=float("inf"))


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 31 - 1))
    return g


_PE_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "pos_embed_pe_400x512.npy")


def positional_table(max_len=400, d_model=512):
    """network_modules.py:115-130 PositionalEmbedding buffer.  The reference computes it with float32 torch ops
    (`torch.sin(position * div_term)`, arguments up to 400), whose vectorised sin / cos / exp differ between CPUs: the build
    container (Xeon) and the GPU box's host (EPYC) disagree on 1.7 % of the entries by up to 1.5e-5 -- enough to move fp16
    roundings downstream and to make a "deterministic" stand-in checkpoint machine-dependent (round 4: the exactly-rounded
    yardstick minted here did not reproduce there).  A real checkpoint carries the buffer in its state_dict; the stand-in
    checkpoints carry THIS table (computed once in the build container, shipped as data) for the one size both networks use."""
    if (max_len, d_model) == (400, 512) and os.path.exists(_PE_FILE):
        import numpy as np
        return torch.from_numpy(np.load(_PE_FILE)).unsqueeze(0).clone()
    pe = torch.zeros(max_len, d_model, dtype=torch.float32)
    position = torch.arange(0, max_len).float().unsqueeze(1)
    div_term = (torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model)).exp()[None]
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def fill_state_dict(template, seed=0, head_gain=1.0):
    """template: {key: tensor} giving names/shapes/dtypes; returns a new dict with seeded values."""
    out = {}
    for k, t in template.items():
        g = _gen(seed, k)
        shape = tuple(t.shape)
        if k.endswith("num_batches_tracked"):
            v = torch.zeros(shape, dtype=t.dtype)
        elif k.endswith("pos_embed.pe"):
            v = positional_table(shape[1], shape[2])
        elif k.endswith("running_mean"):
            v = torch.randn(shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            v = torch.rand(shape, generator=g) + 0.5
        elif "norm" in k.split(".")[-2] or ".bn" in k or k.split(".")[-2] == "1" and len(shape) == 1 and "net" in k:
            # BatchNorm / LayerNorm affine parameters
            v = (torch.rand(shape, generator=g) + 0.5) if k.endswith("weight") else torch.randn(shape, generator=g) * 0.1
        elif len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            v = torch.randn(shape, generator=g) * math.sqrt(1.0 / fan_in)
            if ("head.1." in k) or k.startswith("linear."):
                v = v * head_gain
        else:
            v = torch.randn(shape, generator=g) * 0.05
        out[k] = v.to(t.dtype) if t.dtype.is_floating_point else v.to(t.dtype)
    return out


def random_state_dict(kind, cfg=None, seed=0, head_scale=1.0, heads="random"):
    """kind in {'refine','score'} -> seeded state_dict for RefineNet / ScoreNetMultiPair.

    head_scale multiplies the refiner's output heads (trans_head.1 / rot_head.1 weight and bias).  A TRAINED refiner is
    a contraction (its update shrinks the pose error); these untrained stand-ins are the opposite -- d(update)/d(pose)
    is ~40-120 for the seed-0 checkpoint on the synthetic scene (measured with the CPU oracle), so a
    free-running chain of iterations amplifies any last-bit difference into a different trajectory.  head_scale < ~0.008
    makes the iteration map non-expanding; the free-running parity tests use CONTRACTION_HEAD_SCALE.

    heads="fitted" (round 5; refiner, seed 0 only): the two final Linear(512 -> 3) layers come from a ridge regression of the pose
    error on the pooled features of THIS seeded network over perturbations of the synthetic scene's ground-truth pose
    (tests/golden/fit_contraction_heads.py -> data/standin_fitted_heads.npz): a genuine contraction towards the observed pose with
    full-size first updates (rotation up to rot_normalizer, translation up to ~3 cm), instead of down-scaled random heads."""
    from .refine_network import RefineNet
    from .score_network import ScoreNetMultiPair
    if kind == "refine":
        cfg = dict(DEFAULT_REFINE_CFG, **(cfg or {}))
        with torch.device("meta"):
            net = RefineNet(cfg=cfg, c_in=cfg["c_in"])
        gain = 8.0
    elif kind == "score":
        cfg = dict(DEFAULT_SCORE_CFG, **(cfg or {}))
        with torch.device("meta"):
            net = ScoreNetMultiPair(cfg=cfg, c_in=cfg["c_in"])
        gain = 8.0
    else:
        raise ValueError(kind)
    template = {k: v for k, v in net.state_dict().items()}
    calibrated = seed == 0 and cfg.get("use_BN", False) and cfg.get("rot_rep", "axis_angle") == "axis_angle"
    sd = fill_state_dict(template, seed=seed, head_gain=1.0 if calibrated else gain)
    if calibrated:
        sd.update(_calibration_overlay(kind, sd))
    if heads == "fitted":
        if not (kind == "refine" and calibrated):
            raise ValueError("heads='fitted' exists for the calibrated seed-0 refiner only")
        import numpy as np
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "standin_fitted_heads.npz")
        fit = np.load(path)
        for k in ("trans_head.1.weight", "trans_head.1.bias", "rot_head.1.weight", "rot_head.1.bias"):
            sd[k] = torch.from_numpy(fit[k]).to(sd[k].dtype).reshape(sd[k].shape)
    elif heads != "random":
        raise ValueError(heads)
    if kind == "refine" and head_scale != 1.0:
        for k in ("trans_head.1.weight", "trans_head.1.bias", "rot_head.1.weight", "rot_head.1.bias"):
            sd[k] = sd[k] * float(head_scale)
    return sd


CONTRACTION_HEAD_SCALE = 0.002


_CALIB = None


def _calibration_overlay(kind, sd):
    """seed-0 checkpoints: BatchNorm statistics and head scales measured on crops of the synthetic scene
    (tests/golden/calibrate_standin.py) so that the stand-in networks are input-sensitive and well scaled."""
    global _CALIB
    import os
    import numpy as np
    if _CALIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "standin_calib.npz")
        _CALIB = dict(np.load(path)) if os.path.exists(path) else {}
    out = {}
    if kind == "score" and "score/att_cross.qk_gain" in _CALIB:
        W = sd["att_cross.in_proj_weight"].clone()
        W[512:1024] = W[:512]
        W[:1024] *= float(_CALIB["score/att_cross.qk_gain"])
        out["att_cross.in_proj_weight"] = W
    for k, v in _CALIB.items():
        pre, key = k.split("/", 1)
        if pre == kind and key in sd:
            out[key] = torch.from_numpy(v).to(sd[key].dtype).reshape(sd[key].shape)
    return out


TRAINED_REFINER_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "standin_trained_refiner.npz")


def trained_refiner_state_dict(path=None):
    """Round 6: the stand-in RefineNet TRAINED on the synthetic scene of the parity tests (tests/golden/train_standin_refiner.py: the
    product's nn.Module under autocast on PyTorch-ROCm, perturbations of the ground-truth pose rendered by fp_render_crops, the
    normalised delta of predict_pose_refine.py:195-234 as the target) -- a genuine contraction towards the observed pose, which is what
    the reference's released checkpoint is and what no random trunk can be.  The file holds matrices and kernels in float16 (the cast
    autocast applies to them anyway, made once) and vectors in float32; returned as a float32 state_dict with the reference's keys."""
    import numpy as np
    path = path or TRAINED_REFINER_FILE
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: the trained stand-in refiner is made by tests/golden/train_standin_refiner.py on a GPU box")
    z = np.load(path)
    out = {}
    for k in z.files:
        v = torch.from_numpy(z[k])
        out[k] = v.float() if v.dtype.is_floating_point else v
    return out
