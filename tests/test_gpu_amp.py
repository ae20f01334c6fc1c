"""GPU parity of the DEPLOYED configuration (precision='fp16' = the reference's autocast policy on libfp_amd.so) against
the matched-dtype oracle (oracle/nets_amp.py, pinned against the reference's modules under torch.autocast):
  1. every network kernel against a torch-CPU emulation of the same op with the same rounding points: equal up to
     fp32-summation-order flips (tests/amp_util.py);
  2. encoder / plans against the oracle on the same inputs;
  3. BASELINE size: 252 hypotheses -- each of 5 refine iterations from bit-identical poses (teacher forced, calibrated
     stand-in weights) as a THREE-way comparison HIP plan / the nn.Module under torch.autocast on PyTorch-ROCm / oracle,
     the free-running 5-iteration chain (contraction-scaled heads, weights.CONTRACTION_HEAD_SCALE) and the 252 scores
     (Kendall tau, top-1), also three ways.  The measured error distributions are written to
     gpurun_out/parity_amp.json (committed under profiles/)."""
import json
import os
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from amp_util import assert_equal_up_to_flips, conv_amp_ref, flip_report, geodesic, kendall_tau, r16, ulp16
from conftest import ROOT

pytestmark = pytest.mark.gpu
REPORT = {}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module", autouse=True)
def _write_report():
    yield
    if REPORT:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_amp.json")
        merged = {}
        if os.path.exists(path):          # several pytest invocations (-k subsets) contribute to one report
            try:
                with open(path) as f:
                    merged = json.load(f)
            except Exception:
                merged = {}
        for k, v in REPORT.items():
            if isinstance(v, dict) and isinstance(merged.get(k), dict):
                merged[k].update(v)
            else:
                merged[k] = v
        with open(path, "w") as f:
            json.dump(merged, f, indent=1, sort_keys=True)


def _padded_nhwc(x_nchw, pad, dev):
    B, Cc, H, W = x_nchw.shape
    buf = torch.zeros((B, H + 2 * pad, W + 2 * pad, Cc), dtype=torch.float16, device=dev)
    buf[:, pad:pad + H, pad:pad + W, :] = x_nchw.permute(0, 2, 3, 1).to(dev)
    return buf


def _bn(g, C):
    """random eval BatchNorm as (scale, shift) f32"""
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    mean, var = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    scale = w / torch.sqrt(var + 1e-5)
    return scale, b - mean * scale


# ------------------------------------------------------------------ 1. kernels
@pytest.mark.parametrize("bn,bias", [(True, True), (False, True), (True, False)])
def test_conv7x7_policy(dev, bn, bias):
    from foundationpose_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    x = r16(torch.rand((5, 6, 160, 160), generator=g) * 2 - 1)
    w = r16(torch.randn((64, 6, 7, 7), generator=g) * 0.06 + torch.arange(64)[:, None, None, None] * 1e-4)
    b = r16(torch.randn(64, generator=g) * 0.2) if bias else None
    sb = _bn(g, 64) if bn else None
    ref, mag, slack = conv_amp_ref(x, w, b, sb, 2)
    D = lambda t: None if t is None else t.to(dev)
    for pad in (1, 0):
        buf = torch.full((5, 80 + 2 * pad, 80 + 2 * pad, 64), 7.0, dtype=torch.float16, device=dev)
        ops.conv7x7s2_bn_relu(x.half().to(dev), w.half().reshape(64, -1).contiguous().to(dev), D(b), D(sb[0]) if bn else None,
                              D(sb[1]) if bn else None, buf, pad)
        out = buf[:, pad:pad + 80, pad:pad + 80].permute(0, 3, 1, 2).float().cpu()
        assert_equal_up_to_flips(out.numpy(), ref.numpy(), mag.numpy(), max_frac=0.01, max_ulps=4.2 if bn else 2.0, what=f"conv1 pad={pad}", slack=slack.numpy())
        if pad:   # the border belongs to the caller
            assert float((buf[:, 0] - 7).abs().max()) == 0 and float((buf[:, -1] - 7).abs().max()) == 0
            assert float((buf[:, :, 0] - 7).abs().max()) == 0 and float((buf[:, :, -1] - 7).abs().max()) == 0
    # ragged shapes: odd number of bands, width not a multiple of 32 pixels per tile
    x2 = r16(torch.rand((3, 6, 104, 88), generator=g) * 2 - 1)
    ref2, mag2, slack2 = conv_amp_ref(x2, w, b, sb, 2)
    buf = torch.zeros((3, 52, 44, 64), dtype=torch.float16, device=dev)
    ops.conv7x7s2_bn_relu(x2.half().to(dev), w.half().reshape(64, -1).contiguous().to(dev), D(b), D(sb[0]) if bn else None,
                          D(sb[1]) if bn else None, buf, 0)
    assert_equal_up_to_flips(buf.permute(0, 3, 1, 2).float().cpu().numpy(), ref2.numpy(), mag2.numpy(), max_frac=0.01, max_ulps=4.2 if bn else 2.0,
                             what="conv1 ragged", slack=slack2.numpy())


# (B, H, Cin, Cout, stride, residual, bn): stride-1 shapes with at least two tiles of rows run the shifted-window kernel
# (conv_sw.hip: N % 256 == 0 -> 256x256 tile, otherwise 512x128), the others the generic implicit GEMM
@pytest.mark.parametrize("B,H,Cin,Cout,stride,res,bn", [
    (3, 40, 128, 128, 1, True, True), (2, 40, 256, 256, 1, True, True), (5, 20, 512, 512, 1, True, False),
    (7, 20, 512, 512, 1, False, True), (3, 24, 64, 384, 1, False, True), (1, 20, 256, 256, 1, True, True),
    (2, 80, 64, 128, 2, False, True), (3, 40, 256, 512, 2, False, True), (1, 8, 128, 128, 1, True, True),
    # more tiles than CUs: whole rounds on the shifted-window kernel, the partial round on the 128x128 kernel (side stream)
    (83, 40, 64, 128, 1, True, True)])
def test_igemm_conv3x3_policy(dev, B, H, Cin, Cout, stride, res, bn):
    from foundationpose_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Cin + H)
    x = F.relu(r16(torch.randn((B, Cin, H, H), generator=g) * 0.5))
    w = r16(torch.randn((Cout, Cin, 3, 3), generator=g) * (1.0 / (3 * Cin ** 0.5)) + torch.arange(Cout)[:, None, None, None] * 1e-5)
    bias = r16(torch.randn(Cout, generator=g) * 0.1)
    sb = _bn(g, Cout) if bn else None
    Ho = H // stride
    r = r16(torch.randn((B, Cout, Ho, Ho), generator=g) * 0.5) if res else None
    ref, mag, slack = conv_amp_ref(x, w, bias, sb, stride, residual=r)
    xb = _padded_nhwc(x.half(), 1, dev)
    wk = w.half().permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(dev)
    y = torch.zeros((B, Ho + 2, Ho + 2, Cout), dtype=torch.float16, device=dev)
    rb = _padded_nhwc(r.half(), 1, dev) if res else None
    gin = ops.IgemmGeom.image(Ho, Ho, 1, Cin, stride=stride, offset=0)
    gin.padded_h, gin.padded_w = H + 2, H + 2
    gout = ops.IgemmGeom.image(Ho, Ho, 1, Cout)
    ops.igemm_f16(xb, gin, wk, bias.to(dev), y, gout, B * Ho * Ho, Cout, Cin, 9, relu=True, residual=rb, r_geom=gout if res else None,
                  bn_scale=sb[0].to(dev) if bn else None, bn_shift=sb[1].to(dev) if bn else None, conv_rounding=True)
    out = y[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).float().cpu()
    # every rounding point of the sequence can flip independently (conv, + bias, BatchNorm, + identity), and a flip before
    # BatchNorm is scaled by |scale| <= 2.1 on its way out: the cap on the size of a deviation grows with the sequence,
    # the FRACTION of deviating elements is what separates summation-order noise from different arithmetic
    cap = (2.0 + (2.2 if bn else 0.0)) + (1.0 if res else 0.0)
    rep = assert_equal_up_to_flips(out.numpy(), ref.numpy(), mag.numpy(), max_frac=0.03, max_ulps=cap, what="conv3x3", slack=slack.numpy())
    REPORT.setdefault("kernel_flip_rates", {})[f"conv3x3 B{B} H{H} {Cin}->{Cout} s{stride}"] = rep
    assert float(y[:, 0].abs().max()) == 0 and float(y[:, :, 0].abs().max()) == 0   # the zero border is left untouched
    assert float(y[:, -1].abs().max()) == 0 and float(y[:, :, -1].abs().max()) == 0


def test_igemm_channel_concat_and_linear_policy(dev):
    """bsplit writes image b and image b+n side by side along C (the A|B feature concat); taps=1 is nn.Linear (one
    rounding of accumulator + bias), with ragged last tiles and ReLU"""
    from foundationpose_amd import ops
    g = torch.Generator(device="cpu").manual_seed(77)
    n, H, Cc = 3, 16, 128
    x = r16(torch.randn((2 * n, Cc, H, H), generator=g) * 0.5)
    w = r16(torch.randn((Cc, Cc, 3, 3), generator=g) * 0.03)
    ref, mag, slack = conv_amp_ref(x, w, None, None, 1, relu=False)
    ref, mag, slack = (torch.cat([t[:n], t[n:]], dim=1) for t in (ref, mag, slack))
    y = torch.zeros((n, H + 2, H + 2, 2 * Cc), dtype=torch.float16, device=dev)
    gin = ops.IgemmGeom.image(H, H, 1, Cc, offset=0)
    gout = ops.IgemmGeom.image(H, H, 1, 2 * Cc, bsplit=n, cgroup=Cc)
    ops.igemm_f16(_padded_nhwc(x.half(), 1, dev), gin, w.half().permute(0, 2, 3, 1).reshape(Cc, -1).contiguous().to(dev), None, y, gout,
                  2 * n * H * H, Cc, Cc, 9, conv_rounding=True)
    assert_equal_up_to_flips(y[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float().cpu().numpy(), ref.numpy(), mag.numpy(), what="bsplit",
                             slack=slack.numpy())
    # (11264, 512, 1536): 264 tiles of 256x256 on 256 CUs -> split launch (whole round + remainder rows on the 128x128 kernel)
    for M, K, N, relu in ((1000, 512, 1536, False), (37, 64, 128, True), (4097, 512, 512, True), (252, 512, 1536, False),
                          (11264, 512, 1536, True)):
        xm = r16(torch.randn((M, K), generator=g))
        wm = r16(torch.randn((N, K), generator=g) * 0.05 + torch.arange(N)[:, None] * 1e-4)   # asymmetric: a transposed fragment cannot pass
        b = r16(torch.randn(N, generator=g))
        acc = (xm.double() @ wm.double().t() + b.double()).float()
        slack = (1e-6 * (xm.double().abs() @ wm.double().abs().t())).float()
        refm = r16(acc)
        refm = F.relu(refm) if relu else refm
        ym = torch.empty((M, N), dtype=torch.float16, device=dev)
        pe = torch.randn((100, N), generator=g)           # second output: f16(f32(y) + pe[m % 100]) (the fused positional table)
        ype = torch.empty((M, N), dtype=torch.float16, device=dev)
        ops.igemm_f16(xm.half().to(dev), ops.IgemmGeom.matrix(K), wm.half().to(dev), b.to(dev), ym, ops.IgemmGeom.matrix(N), M, N, K, 1, relu=relu,
                      pe=pe.to(dev), y_pe=ype)
        assert torch.equal(ype.cpu(), (ym.float().cpu() + pe[torch.arange(M) % 100]).half()), "fused positional table"
        assert_equal_up_to_flips(ym.float().cpu().numpy(), refm.numpy(), acc.abs().numpy(), max_frac=0.02, what=f"linear {M}x{K}x{N}",
                                 slack=slack.numpy())


def test_rowops_policy(dev):
    from foundationpose_amd import ops
    g = torch.Generator(device="cpu").manual_seed(21)
    Bn, S = 5, 400
    tok = r16(torch.randn((Bn, S, 512), generator=g) * 2)
    pe = torch.randn((S, 512), generator=g)
    br = r16(torch.randn((Bn, S, 512), generator=g))
    gamma, beta = torch.rand(512, generator=g) + 0.5, torch.randn(512, generator=g) * 0.1
    D = lambda t: t.to(dev)
    # add_pe: fp16(fp32(tok) + pe)
    x16 = ops.add_pe_f16(D(tok.half()), D(pe))
    assert torch.equal(x16.cpu(), (tok + pe).half())
    # LayerNorm on the fp32 stream, residual given as tok16 + pe or as x32
    x32 = tok + pe
    ref = F.layer_norm(x32 + br, (512,), gamma, beta, 1e-5)
    for kw in (dict(tok16=D(tok.half()), pe=D(pe)), dict(x32=D(x32))):
        y32, y16 = ops.layernorm_res(D(br.half()), D(gamma), D(beta), 1e-5, **kw)
        np.testing.assert_allclose(y32.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)
        assert_equal_up_to_flips(y16.float().cpu().numpy(), r16(ref).numpy(), max_frac=0.01, what="LN fp16 copy")
        assert torch.equal(y16.cpu(), y32.cpu().half())          # the fp16 copy is the rounding of the fp32 stream
    only16 = ops.layernorm_res(D(br.half()), D(gamma), D(beta), 1e-5, x32=D(x32), want32=False)
    assert only16[0] is None and torch.equal(only16[1], y16)
    # token mean: fused with residual + LN, LN only, plain
    m = ops.colmean_f16(D(br.half()), D(gamma), D(beta), 1e-5, resid32=D(x32))
    np.testing.assert_allclose(m.cpu().numpy(), ref.mean(dim=1).numpy(), atol=2e-5, rtol=1e-5)
    m1 = ops.colmean_f16(D(br.half()), D(gamma), D(beta))
    np.testing.assert_allclose(m1.cpu().numpy(), F.layer_norm(br, (512,), gamma, beta, 1e-5).mean(dim=1).numpy(), atol=2e-5, rtol=1e-5)
    m0 = ops.colmean_f16(D(br.half()))
    np.testing.assert_allclose(m0.cpu().numpy(), br.mean(dim=1).numpy(), atol=2e-5, rtol=1e-5)
    assert torch.equal(m, ops.colmean_f16(D(br.half()), D(gamma), D(beta), 1e-5, resid32=D(x32)))   # fixed summation order
    # N-row linears
    for M, K, N in ((252, 512, 3), (5, 512, 6), (252, 512, 512), (1, 512, 1), (7, 64, 130)):
        x = torch.randn((M, K), generator=g)
        w = r16(torch.randn((N, K), generator=g) * 0.05 + torch.arange(N)[:, None] * 1e-3)
        b = r16(torch.randn(N, generator=g))
        ref = (x.double() @ w.double().t() + b.double()).float()
        slack = (1e-6 * (x.double().abs() @ w.double().abs().t())).float().numpy()
        y = ops.rows_linear(D(x), D(w.half()), D(b))
        np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=2e-5)
        yr = ops.rows_linear(D(x), D(w.half()), D(b), round_f16=True)
        assert_equal_up_to_flips(yr.cpu().numpy(), r16(ref).numpy(), max_frac=0.02, what="rows_linear round", slack=slack)
        yh = ops.rows_linear(D(x), D(w.half()), D(b), out_f16=True)
        assert yh.dtype == torch.float16 and torch.equal(yh.float(), yr)
        ref16 = (r16(x).double() @ w.double().t() + b.double()).float()
        y16 = ops.rows_linear(D(x.half()), D(w.half()), D(b))
        np.testing.assert_allclose(y16.cpu().numpy(), ref16.numpy(), atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("B,S", [(3, 400), (1, 252), (2, 130), (1, 1)])
def test_attention_fp16_score_policy(dev, B, S):
    """FP_ATT_FP16_SCORES against the need_weights=True branch written out with its roundings (oracle.nets_amp.
    attention_explicit); the kernel keeps flash order for the probabilities (rounded before, normalised after the second
    product), which the bound below covers"""
    from foundationpose_amd import ops
    from oracle import nets_amp
    H, hd = 4, 128
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + S)
    qkv = r16(torch.randn((B, S, 3 * H * hd), generator=g) * 1.5)
    out = ops.attention_f16(qkv.half().to(dev), H, fp16_scores=True).float().cpu()
    ref = nets_amp.attention_explicit(qkv, H)
    plain = ops.attention_f16(qkv.half().to(dev), H).float().cpu()
    ref_flash = nets_amp.attention_flash(qkv, H)
    e1, e0 = (out - ref).abs().max().item(), (plain - ref_flash).abs().max().item()
    assert e1 < 4e-3 and e0 < 4e-3, (e1, e0)
    if S >= 130:
        assert not torch.equal(out, plain)     # the score rounding is visible: the flag is not a no-op


# ------------------------------------------------------------------ 2. encoder / plans vs the oracle
def _net_inputs(n, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    AB = torch.rand((2 * n, 6, 160, 160), generator=g)
    AB[:, 3:] = AB[:, 3:] * 2 - 1
    return r16(AB * (torch.rand((2 * n, 1, 160, 160), generator=g) > 0.3))


@pytest.mark.parametrize("use_bn", [True, False])
def test_hip_encoder_matches_amp_oracle(dev, use_bn):
    from foundationpose_amd import engine
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    from oracle import nets_amp
    cfg = dict(DEFAULT_REFINE_CFG, use_BN=use_bn)
    sd = random_state_dict("refine", cfg, seed=3)
    n = 3
    AB = _net_inputs(n, 8)
    tr = {}
    nets_amp.encoder_tokens(AB[:n], AB[n:], sd, "encodeA", "encodeAB", tr)
    enc = engine._HipEncoder({k: v.to(dev) for k, v in sd.items()}, "encodeA", "encodeAB", dev)
    hip, x16 = enc(AB.half().to(dev))
    assert hip.shape == (n, 400, 512)
    # the fused positional table: the in_proj operand is the fp16 rounding of fp16 tokens + fp32 table
    assert torch.equal(x16.cpu(), (hip.float().cpu() + sd["pos_embed.pe"].float()[:, :400]).half())
    # first block: a 294-term reduction, compared on its own buffer (interior of the padded NHWC activation)
    c1 = enc._bufs[(n, 160, 160, 0)]["P1"][:, 1:-1, 1:-1].permute(0, 3, 1, 2).float().cpu()
    r1 = flip_report(c1.numpy(), tr["conv1"].numpy())
    rep = flip_report(hip.float().cpu().numpy(), tr["tok16"].numpy())
    REPORT.setdefault("encoder_vs_oracle", {})[f"bn{int(use_bn)}"] = dict(conv1=r1, tokens=rep)
    assert r1["frac"] < 0.01 and r1["max_abs"] <= 4 * ulp16(tr["conv1"].abs().max().item()), r1
    assert rep["rel_rms"] < 2e-3 and rep["max_abs"] <= 16 * ulp16(tr["tok16"].abs().max().item()), rep
    assert torch.equal(hip, enc(AB.half().to(dev))[0])     # second call reuses the cached zero-bordered buffers


def test_plans_match_amp_oracle(dev):
    """RefinePlan / ScorePlan (fp16) on 6 pairs against oracle.nets_amp: outputs within 2 fp16 ulps, features at the
    summation-order floor (measured with the oracle's own reversed-order evaluation)"""
    from foundationpose_amd import engine
    from foundationpose_amd.refine_network import RefineNet
    from foundationpose_amd.score_network import ScoreNetMultiPair
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    from oracle import nets_amp
    n = 6
    AB = _net_inputs(n, 4)
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, 0)
    net = RefineNet(cfg=cfg, c_in=6)
    net.load_state_dict(sd)
    o16 = engine.RefinePlan(net, dev, precision="fp16")(AB.half().to(dev))
    ref = nets_amp.refine_forward(AB[:n], AB[n:], sd)
    nets_amp.REVERSED_SUMS = True
    try:
        ref_r = nets_amp.refine_forward(AB[:n], AB[n:], sd)
    finally:
        nets_amp.REVERSED_SUMS = False
    for k in ("trans", "rot"):
        assert o16[k].dtype == torch.float32
        o = o16[k].cpu().numpy()
        assert np.array_equal(o, o.astype(np.float16).astype(np.float32))           # the reference holds these in fp16
        err, floor = np.abs(o - ref[k].numpy()), np.abs(ref_r[k].numpy() - ref[k].numpy())
        REPORT.setdefault("plans_vs_oracle", {})[k] = dict(max_err=float(err.max()), floor_max=float(floor.max()),
                                                           out_abs_mean=float(np.abs(ref[k].numpy()).mean()))
        assert err.max() <= max(4 * floor.max(), 8 * ulp16(np.abs(ref[k].numpy()).max())), (k, err.max(), floor.max())
    cfg = dict(DEFAULT_SCORE_CFG)
    sd = random_state_dict("score", cfg, 0)
    net = ScoreNetMultiPair(cfg=cfg, c_in=6)
    net.load_state_dict(sd)
    plan = engine.ScorePlan(net, dev, precision="fp16")
    f16 = plan.features(AB.half().to(dev))
    assert f16.dtype == torch.float16
    fref = nets_amp.score_features(AB[:n], AB[n:], sd)
    rep = flip_report(f16.float().cpu().numpy(), fref.numpy())
    REPORT["plans_vs_oracle"]["score_features"] = rep
    assert rep["rel_rms"] < 4e-3, rep
    logits = plan.head(f16, L=n).cpu().numpy().reshape(-1)
    lref = nets_amp.score_forward(AB[:n], AB[n:], sd, n)["score_logit"].numpy().reshape(-1)
    assert np.abs(logits - lref).max() <= 0.05 * max(1.0, lref.std()) + 4 * ulp16(np.abs(lref).max()), (logits, lref)


# ------------------------------------------------------------------ 3. BASELINE size
@pytest.fixture(scope="module")
def gmesh(scene, dev):
    from foundationpose_amd.Utils import make_mesh_tensors
    return make_mesh_tensors(scene["mesh"], device=dev)


@pytest.fixture(scope="module")
def frame(scene, dev):
    from oracle import ops as oo
    from oracle import pipeline as op
    d = op.preprocess_depth(scene["depth"])
    xyz = oo.depth2xyzmap(d, scene["K"], f64_internal=True)
    return dict(depth_f=d, xyz=xyz, depth_t=torch.as_tensor(d, device=dev), xyz_t=torch.as_tensor(xyz, device=dev))


def _pct(x):
    x = np.asarray(x, dtype=np.float64)
    return dict(median=float(np.median(x)), p90=float(np.percentile(x, 90)), p99=float(np.percentile(x, 99)), max=float(x.max()))


def test_refiner_252_teacher_forced_three_way(scene, dev, gmesh, frame):
    """252 hypotheses, calibrated stand-in weights (|update| ~ 2 cm / 0.2-0.36 rad), each of the 5 iterations started from
    the oracle's pose of the previous one (bit-identical inputs), THREE implementations of the reference's autocast policy:
      hip     the deployed plan: every network op on libfp_amd.so (precision='fp16')
      lib     the product's nn.Module under torch.autocast('cuda', float16) on PyTorch-ROCm: MIOpen / rocBLAS / ATen
              kernels (precision='torch_amp') -- nothing of it is ours
      oracle  oracle/nets_amp.py on the CPU (explicit casts; pinned against the reference under CPU autocast)
    The refined poses of the three are compared pairwise.  With these weights the policy itself carries ~0.3 % of the update
    as fp16 rounding noise (each implementation rounds a different fp32 summation order), so 1e-4 rad is not reachable by
    ANY pair -- the gate is that the HIP plan is as close to the oracle as the library is (x1.5 on median / p90: if all
    three deviate independently by sigma from the exactly-rounded result, every pair is sqrt(2) sigma apart, but MIOpen's
    fallback convolution here accumulates in float64, i.e. sigma_lib ~ 0, which makes lib-vs-oracle the smallest pair),
    or inside the north-star 1e-4.  No slack against a self-made floor; the reversed-summation floor of the oracle against
    itself is only reported."""
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
    from oracle import nets_amp, ops as oo
    from oracle import pipeline as op
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, seed=0)
    P0 = scene["poses"]
    t0 = time.time()
    trace = []
    op.refine_predict(cfg, sd, scene["rgb"], frame["depth_f"], scene["K"], P0, frame["xyz"], scene["mesh_np"], scene["diameter"],
                      iteration=5, trace=trace, amp=True)
    # reported only: same policy, reversed summation order, same inputs
    nf = 64
    A, B = torch.from_numpy(trace[0]["A"][:nf]), torch.from_numpy(trace[0]["B"][:nf])
    nets_amp.REVERSED_SUMS = True
    try:
        o_r = nets_amp.refine_forward(A, B, sd)
    finally:
        nets_amp.REVERSED_SUMS = False
    tn = [float(v) for v in cfg["trans_normalizer"]]
    p_r = oo.pose_update(o_r["trans"].numpy(), o_r["rot"].numpy(), P0[:nf], cfg["rot_rep"], True, tn, float(cfg["rot_normalizer"]),
                         float(scene["diameter"]))
    floor_R = geodesic(p_r[:, :3, :3], trace[0]["poses"][:nf, :3, :3])
    floor_t = np.linalg.norm(p_r[:, :3, 3] - trace[0]["poses"][:nf, :3, 3], axis=1)
    # which conv-bias policy does the library follow?  cuDNN / MIOpen add the bias to the rounded fp16 output ("separate", what
    # the HIP plan and the oracle's default do); ATen's own convolution (im2col + GEMM, used here because MIOpen has no tuned
    # gfx950 kernels in this image) starts the fp32 accumulation from the bias ("fused", one rounding, like the CPU backend).
    # Evaluated on the same 64 hypotheses of iteration 0; the library is compared with BOTH below.
    nets_amp.CONV_BIAS = "fused"
    try:
        o_f = nets_amp.refine_forward(A, B, sd)
    finally:
        nets_amp.CONV_BIAS = "separate"
    p_fused = oo.pose_update(o_f["trans"].numpy(), o_f["rot"].numpy(), P0[:nf], cfg["rot_rep"], True, tn, float(cfg["rot_normalizer"]),
                             float(scene["diameter"]))
    t_oracle = time.time() - t0
    preds = dict(hip=PoseRefinePredictor(cfg=cfg, state_dict=sd, device=dev, precision="fp16"),
                 lib=PoseRefinePredictor(cfg=cfg, state_dict=sd, device=dev, precision="torch_amp", n_streams=1))
    rep = dict(oracle_seconds=t_oracle, oracle_reversed_sum_floor_dR=_pct(floor_R), oracle_reversed_sum_floor_dt=_pct(floor_t),
               iterations=[], seconds=dict(hip=0.0, lib=0.0))
    start = P0
    for it in range(5):
        tgt = trace[it]["poses"]
        out = {}
        for name, pred in preds.items():
            t1 = time.time()
            o, _ = pred.predict(scene["rgb"], frame["depth_t"], scene["K"], start, frame["xyz_t"], mesh=scene["mesh"],
                                mesh_tensors=gmesh, mesh_diameter=scene["diameter"], iteration=1)
            out[name] = o.cpu().numpy()
            rep["seconds"][name] += time.time() - t1
            raw = {k: v.cpu().numpy() for k, v in pred.last_raw_output.items()}
            for k in ("trans", "rot"):            # the reference holds the raw outputs in fp16
                assert np.array_equal(raw[k], raw[k].astype(np.float16).astype(np.float32)), (name, k)
        out["oracle"] = tgt
        uR, ut = geodesic(tgt[:, :3, :3], start[:, :3, :3]), np.linalg.norm(tgt[:, :3, 3] - start[:, :3, 3], axis=1)
        row = dict(update_dR=_pct(uR), update_dt=_pct(ut))
        for a_, b_ in (("hip", "oracle"), ("lib", "oracle"), ("hip", "lib")):
            dR = geodesic(out[a_][:, :3, :3], out[b_][:, :3, :3])
            dt = np.linalg.norm(out[a_][:, :3, 3] - out[b_][:, :3, 3], axis=1)
            row[f"{a_}_vs_{b_}"] = dict(dR=_pct(dR), dt=_pct(dt), rel_dR=_pct(dR / np.maximum(uR, 1e-9)), rel_dt=_pct(dt / np.maximum(ut, 1e-9)))
        if it == 0:     # the first 64 hypotheses against the oracle evaluated with the other bias policy
            for name in ("hip", "lib"):
                dRf = geodesic(out[name][:nf, :3, :3], p_fused[:, :3, :3])
                dRs = geodesic(out[name][:nf, :3, :3], tgt[:nf, :3, :3])
                row[f"{name}_first64_vs_oracle_bias_fused_dR"] = _pct(dRf)
                row[f"{name}_first64_vs_oracle_bias_separate_dR"] = _pct(dRs)
            rep["library_conv_bias_policy"] = "fused" if row["lib_first64_vs_oracle_bias_fused_dR"]["median"] < \
                row["lib_first64_vs_oracle_bias_separate_dR"]["median"] else "separate"
        rep["iterations"].append(row)
        start = tgt
    REPORT["refiner_252_teacher_forced_three_way"] = rep
    for it, r in enumerate(rep["iterations"]):
        h, l = r["hip_vs_oracle"], r["lib_vs_oracle"]
        for q, tol in (("dR", 1e-4), ("dt", 1e-4)):
            for stat in ("median", "p90"):
                assert h[q][stat] <= 1.5 * max(l[q][stat], tol), (it, q, stat, h[q], l[q])
            assert h[q]["max"] <= 2.0 * max(l[q]["max"], tol), (it, q, h[q], l[q])
        assert h["rel_dR"]["median"] < 0.01 and h["rel_dt"]["median"] < 0.02, (it, h)
        assert r["update_dR"]["median"] > 0.05                 # full-size updates: nothing is scaled down


@pytest.fixture(scope="module")
def chain_ref(scene, frame):
    """the oracle's free-running 5-iteration chain over the 252 hypotheses (autocast policy, contraction-scaled heads)"""
    from foundationpose_amd.weights import CONTRACTION_HEAD_SCALE, DEFAULT_REFINE_CFG, random_state_dict
    from oracle import pipeline as op
    cfg = dict(DEFAULT_REFINE_CFG)
    sd = random_state_dict("refine", cfg, seed=0, head_scale=CONTRACTION_HEAD_SCALE)
    ref = op.refine_predict(cfg, sd, scene["rgb"], frame["depth_f"], scene["K"], scene["poses"], frame["xyz"], scene["mesh_np"],
                            scene["diameter"], iteration=5, amp=True)
    return dict(cfg=cfg, sd=sd, ref=ref)


def test_refiner_252_free_running_chain(scene, dev, gmesh, frame, chain_ref):
    """The chain the metric times (estimater.py:215: 252 hypotheses, iteration=5, free running) in the deployed dtype, with
    contraction-scaled stand-in heads (weights.CONTRACTION_HEAD_SCALE: a trained refiner is a contraction, the unscaled
    stand-in expands a last-bit difference 40-120x per iteration, so an unscaled free-running chain compares chaotic
    trajectories).  Reported next to the same chain on PyTorch-ROCm under autocast.  This is NOT the parity gate of the
    deployed dtype (that is the full-size three-way test above); it checks that five iterations chained on the device -- no
    host round trip, sub-batches on two streams -- end where the oracle's chain ends: the bulk inside the north-star
    1e-4 rad / 1e-4 m, the tail (hypotheses where a crop pixel flips coverage or its nearest-neighbour source between the two
    runs) bounded, and no worse than the library's."""
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.weights import CONTRACTION_HEAD_SCALE
    cfg, sd, ref = chain_ref["cfg"], chain_ref["sd"], chain_ref["ref"]
    P0 = scene["poses"]
    mR, mt = geodesic(ref[:, :3, :3], P0[:, :3, :3]), np.linalg.norm(ref[:, :3, 3] - P0[:, :3, 3], axis=1)
    rep = dict(head_scale=CONTRACTION_HEAD_SCALE, total_motion_dR=_pct(mR), total_motion_dt=_pct(mt))
    preds = {}
    for name, kw in (("hip", dict(precision="fp16")), ("lib", dict(precision="torch_amp", n_streams=1))):
        pred = preds[name] = PoseRefinePredictor(cfg=cfg, state_dict=sd, device=dev, **kw)
        out, _ = pred.predict(scene["rgb"], frame["depth_t"], scene["K"], P0, frame["xyz_t"], mesh=scene["mesh"], mesh_tensors=gmesh,
                              mesh_diameter=scene["diameter"], iteration=5)
        out = out.cpu().numpy()
        dR, dt = geodesic(out[:, :3, :3], ref[:, :3, :3]), np.linalg.norm(out[:, :3, 3] - ref[:, :3, 3], axis=1)
        rep[name] = dict(dR=_pct(dR), dt=_pct(dt), rel_dR=_pct(dR / np.maximum(mR, 1e-9)), frac_within_1e4_rad=float(np.mean(dR <= 1e-4)),
                         frac_within_1e4_m=float(np.mean(dt <= 1e-4)))
    REPORT["refiner_252_free_running_5_iterations"] = rep
    h, l = rep["hip"], rep["lib"]
    assert mR.mean() > 1e-3                                   # the chain moves the poses by >> the tolerance
    assert h["frac_within_1e4_rad"] >= min(0.97, l["frac_within_1e4_rad"] - 0.01) and h["frac_within_1e4_m"] >= 0.99, rep
    assert h["dR"]["median"] <= 1.5 * max(l["dR"]["median"], 1e-5) and h["dR"]["max"] <= 1e-3 and h["dt"]["max"] <= 1e-4, rep
    assert h["rel_dR"]["median"] < 0.05, rep
    # last_trans_update / last_rot_update: the reference's semantics (metric delta, applied 3x3 rotation)
    pred = preds["hip"]
    assert pred.last_trans_update.shape == (252, 3) and pred.last_rot_update.shape == (252, 3, 3)
    Rd = pred.last_rot_update.cpu().numpy()
    assert np.abs(Rd @ Rd.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5


def test_scorer_252_three_way(scene, dev, gmesh, frame, chain_ref):
    """the 252 scores of the oracle's refined poses: HIP plan / PyTorch-ROCm under autocast / autocast oracle (and the fp32
    oracle as a yardstick): Kendall tau, top-1, logit errors"""
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_SCORE_CFG, random_state_dict
    from oracle import pipeline as op
    ref = chain_ref["ref"]
    scfg = dict(DEFAULT_SCORE_CFG)
    ssd = random_state_dict("score", scfg, seed=0)
    sref = op.score_predict(scfg, ssd, scene["rgb"], frame["depth_f"], scene["K"], ref, scene["mesh_np"], scene["diameter"], amp=True)
    s32 = op.score_predict(scfg, ssd, scene["rgb"], frame["depth_f"], scene["K"], ref, scene["mesh_np"], scene["diameter"], amp=False)
    scorer = ScorePredictor(cfg=scfg, state_dict=ssd, device=dev, precision="fp16")
    s, _ = scorer.predict(scene["rgb"], frame["depth_t"], scene["K"], ref, mesh=scene["mesh"], mesh_tensors=gmesh,
                          mesh_diameter=scene["diameter"])
    s = s.cpu().numpy()
    tau, tau_floor = kendall_tau(s, sref), kendall_tau(s32, sref)
    srep = dict(kendall_tau_vs_amp_oracle=tau, kendall_tau_fp32_oracle_vs_amp_oracle=tau_floor, top1_equal=bool(np.argmax(s) == np.argmax(sref)),
                hip_top1_rank_in_oracle=int(np.argsort(-sref).tolist().index(int(np.argmax(s)))), abs_err=_pct(np.abs(s - sref)),
                fp32_vs_amp_abs_err=_pct(np.abs(s32 - sref)), logit_std=float(sref.std()))
    # third implementation: the nn.Module under torch.autocast on PyTorch-ROCm (MIOpen / rocBLAS / ATen)
    lib = ScorePredictor(cfg=scfg, state_dict=ssd, device=dev, precision="torch_amp", n_streams=1)
    sl, _ = lib.predict(scene["rgb"], frame["depth_t"], scene["K"], ref, mesh=scene["mesh"], mesh_tensors=gmesh, mesh_diameter=scene["diameter"])
    sl = sl.cpu().numpy()
    srep.update(kendall_tau_lib_vs_amp_oracle=kendall_tau(sl, sref), kendall_tau_hip_vs_lib=kendall_tau(s, sl),
                lib_abs_err=_pct(np.abs(sl - sref)), hip_vs_lib_abs_err=_pct(np.abs(s - sl)), lib_top1_equal=bool(np.argmax(sl) == np.argmax(sref)))
    REPORT["scorer_252"] = srep
    assert tau >= srep["kendall_tau_lib_vs_amp_oracle"] - 0.01, srep          # as close to the oracle as the library is
    assert np.abs(s - sref).max() <= 1.5 * max(np.abs(sl - sref).max(), 0.02 * sref.std()), srep
    assert tau >= min(0.98, tau_floor - 0.01), srep
    assert srep["hip_top1_rank_in_oracle"] <= 2, srep
    assert np.abs(s - sref).max() <= max(4 * np.abs(s32 - sref).max(), 0.02 * sref.std()), srep
