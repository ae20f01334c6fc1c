"""CPU: pins oracle/nets_amp.py (the fp16-autocast arithmetic policy) against the reference's own modules run under
torch.autocast(fp16) -- tests/golden/nets_amp_golden.npz, minted by tests/golden/make_golden_amp.py in the build
container.  The first conv block must agree BIT FOR BIT (same cast points, and the 6x7x7 reduction is short enough
that both sides round the same fp32 sums); deeper layers differ only by fp32 summation order, which flips an fp16
rounding in ~0.3 % of the outputs of every convolution (measured: the reference's own CPU kernels against an fp64
accumulation of the same fp16 operands do the same), so they are held to a relative RMS bound instead."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "nets_amp_golden.npz")
STRIDE = 61


def _inputs(n, seed):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.golden_inputs(n, seed)


def _s(t):
    return t.float().reshape(-1)[::STRIDE].numpy()


def _rel_rms(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-12))


def _ulp16(x):
    ax = np.maximum(np.abs(x), 2.0 ** -14)
    return 2.0 ** (np.floor(np.log2(ax)) - 10)


@pytest.fixture()
def amp():
    from oracle import nets_amp
    old = nets_amp.CONV_BIAS
    nets_amp.CONV_BIAS = "fused"     # the CPU backend's convolution (see make_golden_amp.py)
    yield nets_amp
    nets_amp.CONV_BIAS = old


@pytest.mark.parametrize("use_bn", [True, False])
def test_refine_amp_oracle_matches_reference_under_autocast(amp, use_bn):
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    g = np.load(GOLD)
    cfg = dict(DEFAULT_REFINE_CFG, use_BN=use_bn)
    sd = random_state_dict("refine", cfg, seed=1)
    A, B = _inputs(3, 11)
    tag = f"refine_bn{int(use_bn)}"
    tr = {}
    tok = amp.encoder_tokens(A, B, sd, "encodeA", "encodeAB", tr)
    assert np.array_equal(_s(tr["conv1"]), g[tag + "_conv1"])                 # conv + bias + BN + ReLU: bit for bit
    stem = _s(tr["stem"])
    assert np.mean(stem != g[tag + "_stem"]) < 0.3 and _rel_rms(stem, g[tag + "_stem"]) < 1.5e-3
    n = A.shape[0]
    joint = _s(tr["tok16"].permute(0, 2, 1).reshape(n, 512, 20, 20))
    assert _rel_rms(joint, g[tag + "_joint"]) < 2e-3
    assert np.abs(joint - g[tag + "_joint"]).max() <= 8 * _ulp16(g[tag + "_joint"]).max()
    assert _rel_rms(_s(tok), g[tag + "_tok"]) < 2e-3
    p = "trans_head.0"
    sa = amp.mha(tok, sd, p + ".self_attn", explicit=False)
    assert _rel_rms(_s(sa), g[tag + "_sa"]) < 3e-3
    assert _rel_rms(_s(amp._ln(tok + sa, sd, p + ".norm1")), g[tag + "_n1"]) < 3e-3
    assert _rel_rms(_s(amp.encoder_layer(tok, sd, p)), g[tag + "_layer"]) < 3e-3
    out = amp.refine_forward(A, B, sd)
    for k in ("trans", "rot"):
        ref = g[f"{tag}_{k}"]
        assert out[k].dtype == torch.float32 and np.array_equal(out[k].numpy(), out[k].half().float().numpy())
        assert np.all(np.abs(out[k].numpy() - ref) <= 2 * _ulp16(ref)), (k, out[k].numpy(), ref)


@pytest.mark.parametrize("use_bn", [True, False])
def test_score_amp_oracle_matches_reference_under_autocast(amp, use_bn):
    from foundationpose_amd.weights import DEFAULT_SCORE_CFG, random_state_dict
    g = np.load(GOLD)
    cfg = dict(DEFAULT_SCORE_CFG, use_BN=use_bn)
    sd = random_state_dict("score", cfg, seed=2)
    A, B = _inputs(6, 12)
    tag = f"score_bn{int(use_bn)}"
    tr = {}
    tok = amp.encoder_tokens(A, B, sd, "encoderA", "encoderAB", tr)
    joint = _s(tr["tok16"].permute(0, 2, 1).reshape(6, 512, 20, 20))
    assert _rel_rms(joint, g[tag + "_joint"]) < 2e-3
    att = amp.mha(tok, sd, "att", explicit=True)
    assert _rel_rms(_s(att), g[tag + "_att"]) < 3e-3
    feats = amp.score_features(A, B, sd).numpy()
    assert _rel_rms(feats, g[tag + "_feats"]) < 2e-3
    out = amp.score_forward(A, B, sd, L=6)["score_logit"].numpy()
    ref = g[tag + "_L6"]
    assert np.all(np.abs(out - ref) <= 3 * _ulp16(ref)), (out, ref)


def test_policy_deviations_are_at_the_summation_order_floor(amp):
    """What the deployed HIP plan does differently from the autocast op sequence, measured on the oracle itself against
    the reference-under-autocast goldens: (a) the token mean commuted with the output Linear (engine.py), (b) flash-order
    attention in the scorer instead of the fp16-rounded score matrix of the need_weights=True branch.  Both stay at
    the floor that fp32 summation order alone produces (the `fused` oracle against the golden)."""
    import torch.nn.functional as F
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    g = np.load(GOLD)
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, seed=1)
    A, B = _inputs(3, 11)
    tok = amp.encoder_tokens(A, B, sd, "encodeA", "encodeAB")
    for name in ("trans", "rot"):
        h = amp.encoder_layer(tok, sd, f"{name}_head.0")
        w, b = amp.r16(sd[f"{name}_head.1.weight"]), amp.r16(sd[f"{name}_head.1.bias"])
        commuted = amp.r16(F.linear(h.mean(dim=1), w, b)).numpy()
        ref = g[f"refine_bn1_{name}"]
        assert np.all(np.abs(commuted - ref) <= 2 * _ulp16(ref))
    cfg = dict(DEFAULT_SCORE_CFG)
    sd = random_state_dict("score", cfg, seed=2)
    A, B = _inputs(6, 12)
    tok = amp.encoder_tokens(A, B, sd, "encoderA", "encoderAB")
    ref = g["score_bn1_feats"]
    floor = _rel_rms(amp.r16(amp.mha(tok, sd, "att", explicit=True).mean(dim=1)).numpy(), ref)
    flash = _rel_rms(amp.r16(amp.mha(tok, sd, "att", explicit=False).mean(dim=1)).numpy(), ref)
    assert flash < 1.5 * floor + 1e-4, (flash, floor)
