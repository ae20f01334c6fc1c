"""Generates tests/golden/nets_amp_golden.npz: the REFERENCE's own model code
(/root/reference/learning/models/{refine_network,score_network}.py) run under torch.autocast(fp16) -- the deployed
configuration of predict_pose_refine.py:190-191 / predict_score.py:193-194 (`torch.cuda.amp.autocast`), here with the
CPU autocast backend because the build container has no GPU.  Run in the build container only:

    python tests/golden/make_golden_amp.py

CPU autocast and CUDA autocast put these modules through the same op sequence and cast points (conv / linear / bmm /
SDPA in fp16, LayerNorm and the residual stream in fp32 because their inputs are fp32, softmax output consumed as fp16)
with ONE difference: the CPU convolution adds its bias to the fp32 accumulator, the CUDA / ROCm backends add it to the
rounded fp16 output.  oracle/nets_amp.py has a switch for exactly that (CONV_BIAS); the pin test uses "fused".

Stored: the network outputs, and strided samples of the intermediate activations (forward hooks) so that the
restatement is pinned layer by layer and not only through a 3-number output."""
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict  # noqa: E402
from make_golden import Cfg, golden_inputs  # noqa: E402

STRIDE = 61   # sample stride of the stored intermediates (prime: visits every channel / position class)


def sample(t):
    return t.detach().float().reshape(-1)[::STRIDE].numpy().copy()


def _hook(cap, name):
    def h(m, i, o):
        cap[name] = sample(o[0] if isinstance(o, tuple) else o)
    return h


def main():
    for m in ("Utils", "cv2"):
        sys.modules[m] = types.ModuleType(m)
    sys.path.insert(0, "/root/reference/learning/models")
    rn = importlib.import_module("refine_network")
    sn = importlib.import_module("score_network")
    torch.set_num_threads(8)
    out = {}
    for use_bn in (True, False):
        cfg = dict(DEFAULT_REFINE_CFG, use_BN=use_bn)
        sd = random_state_dict("refine", cfg, seed=1)
        net = rn.RefineNet(cfg=Cfg(cfg), c_in=6).eval()
        net.load_state_dict(sd, strict=True)
        A, B = golden_inputs(3, seed=11)
        cap = {}
        net.encodeA[0].register_forward_hook(_hook(cap, "conv1"))
        net.encodeA.register_forward_hook(_hook(cap, "stem"))
        net.encodeAB.register_forward_hook(_hook(cap, "joint"))
        net.pos_embed.register_forward_hook(_hook(cap, "tok"))
        net.trans_head[0].self_attn.register_forward_hook(_hook(cap, "sa"))
        net.trans_head[0].norm1.register_forward_hook(_hook(cap, "n1"))
        net.trans_head[0].register_forward_hook(_hook(cap, "layer"))
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
            o = net(A, B)
        tag = f"refine_bn{int(use_bn)}"
        assert o["trans"].dtype == torch.float16
        out[tag + "_trans"] = o["trans"].float().numpy()
        out[tag + "_rot"] = o["rot"].float().numpy()
        for k, v in cap.items():
            out[f"{tag}_{k}"] = v
    for use_bn in (True, False):
        cfg = dict(DEFAULT_SCORE_CFG, use_BN=use_bn)
        sd = random_state_dict("score", cfg, seed=2)
        net = sn.ScoreNetMultiPair(cfg=Cfg(cfg), c_in=6).eval()
        net.load_state_dict(sd, strict=True)
        A, B = golden_inputs(6, seed=12)
        cap = {}
        net.encoderAB.register_forward_hook(_hook(cap, "joint"))
        net.att.register_forward_hook(_hook(cap, "att"))
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
            feats = net.extract_feat(A, B)
            o = net(A, B, L=6)
        tag = f"score_bn{int(use_bn)}"
        out[tag + "_feats"] = feats.float().numpy()
        out[tag + "_L6"] = o["score_logit"].float().numpy()
        for k, v in cap.items():
            out[f"{tag}_{k}"] = v
    path = os.path.join(ROOT, "tests", "golden", "nets_amp_golden.npz")
    np.savez_compressed(path, **out)
    for k, v in out.items():
        print(k, v.shape, v.ravel()[:3])


if __name__ == "__main__":
    main()
