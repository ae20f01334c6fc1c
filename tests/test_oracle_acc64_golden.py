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
