"""Generates tests/golden/nets_golden.npz by running the REFERENCE's own model code
(/root/reference/learning/models/{refine_network,score_network}.py, imported with empty stub `Utils`/`cv2`
modules which they import but never use) on seeded inputs with the seeded stand-in checkpoints of
foundationpose_amd.weights.  Run in the build container only (the reference is not on the GPU box):

    python tests/golden/make_golden.py

The npz stores outputs only; inputs and weights are re-derived from seeds by the tests."""
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict  # noqa: E402


class Cfg(dict):
    __getattr__ = dict.__getitem__


def golden_inputs(n, seed):
    """Seeded network inputs in the value range of real A/B tensors: rgb in [0,1], xyz in [-1,1] with zeros."""
    g = torch.Generator().manual_seed(seed)
    def one():
        rgb = torch.rand((n, 3, 160, 160), generator=g)
        xyz = torch.rand((n, 3, 160, 160), generator=g) * 2 - 1
        mask = (torch.rand((n, 1, 160, 160), generator=g) > 0.4).float()
        return torch.cat([rgb * mask, xyz * mask], dim=1)
    return one(), one()


def main():
    for m in ("Utils", "cv2"):
        sys.modules[m] = types.ModuleType(m)
    sys.path.insert(0, "/root/reference/learning/models")
    rn = importlib.import_module("refine_network")
    sn = importlib.import_module("score_network")
    torch.set_num_threads(8)
    out = {}
    for use_bn in (True, False):
        for rot_rep in ("axis_angle", "6d"):
            cfg = dict(DEFAULT_REFINE_CFG, use_BN=use_bn, rot_rep=rot_rep)
            sd = random_state_dict("refine", cfg, seed=1)
            net = rn.RefineNet(cfg=Cfg(cfg), c_in=6).eval()
            net.load_state_dict(sd, strict=True)
            A, B = golden_inputs(2, seed=11)
            with torch.no_grad():
                o = net(A, B)
            tag = f"refine_bn{int(use_bn)}_{rot_rep}"
            out[tag + "_trans"] = o["trans"].numpy()
            out[tag + "_rot"] = o["rot"].numpy()
    for use_bn in (True, False):
        cfg = dict(DEFAULT_SCORE_CFG, use_BN=use_bn)
        sd = random_state_dict("score", cfg, seed=2)
        net = sn.ScoreNetMultiPair(cfg=Cfg(cfg), c_in=6).eval()
        net.load_state_dict(sd, strict=True)
        A, B = golden_inputs(4, seed=12)
        with torch.no_grad():
            o = net(A, B, L=4)
            o2 = net(A, B, L=2)
        out[f"score_bn{int(use_bn)}_L4"] = o["score_logit"].numpy()
        out[f"score_bn{int(use_bn)}_L2"] = o2["score_logit"].numpy()
    path = os.path.join(ROOT, "tests", "golden", "nets_golden.npz")
    np.savez(path, **out)
    for k, v in out.items():
        print(k, v.shape, v.ravel()[:4])


if __name__ == "__main__":
    main()
