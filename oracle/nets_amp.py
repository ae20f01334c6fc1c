"""fp16-autocast restatement of the two networks: the arithmetic POLICY of the reference's deployed configuration
(`with torch.cuda.amp.autocast(enabled=self.amp)`, predict_pose_refine.py:190-191, predict_score.py:193-194), written
with explicit casts so that every rounding point is visible.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

All tensors here are fp32 tensors; `r16` rounds to the nearest fp16 value (the value an fp16 tensor would hold).  A
matrix product of fp16 operands with fp32 accumulation is therefore an fp32 product of r16-rounded operands: fp16 x fp16
products are exact in fp32, only the summation order is unspecified (as it is inside cuDNN / cuBLAS / MFMA).

The op sequence is the one autocast produces for these modules (torch/amp lists):
  * nn.Conv2d        -> fp16 in, fp32 accumulate, fp16 out; the bias is a separate fp16 add (ATen Convolution.cpp adds the
                        bias after cudnn_convolution): r16(r16(conv) + r16(b))
  * nn.BatchNorm2d   -> (eval) runs in the input dtype with fp32 statistics: r16(bn_fp32(x16))
  * ReLU, `out += identity` -> fp16 elementwise
  * PositionalEmbedding: fp16 tokens + fp32 buffer -> fp32 (type promotion, network_modules.py:133-137)
  * nn.Linear        -> fp16 in, fp32 accumulate + bias in the GEMM epilogue, one rounding: r16(x16 @ W16^T + b16)
  * nn.TransformerEncoderLayer (refine_network.py:56-70; the fused fast path is disabled under autocast): the residual
    stream `x + sa`, `x + ff` and both LayerNorms are fp32; every Linear re-rounds its fp32 input to fp16;
    self-attention goes through F.scaled_dot_product_attention (need_weights=False): fp32 scores and softmax
    statistics, probabilities rounded to fp16 before P.V, normalisation after the product (flash-attention order)
  * nn.MultiheadAttention called directly (score_network.py:73,86; need_weights=True): q scaled by sqrt(1/d) and
    rounded to fp16, `bmm` scores rounded to fp16, softmax in fp32, probabilities rounded to fp16, `bmm` -> fp16
  * `.mean(dim=1)` of an fp16 tensor: fp32 accumulation, fp16 result
PINNED by tests/test_oracle_amp_golden.py against the reference's own modules run under torch.autocast (CPU, fp16) in the
build container (tests/golden/nets_amp_golden.npz, made by tests/golden/make_golden_amp.py)."""
import math

import torch
import torch.nn.functional as F


def r16(x):
    return x.to(torch.float16).to(torch.float32)


# "separate": the CUDA / ROCm backends of ATen add the bias to the fp16 convolution output (two roundings);
# "fused": the CPU backend adds it to the fp32 accumulator (one rounding) -- used only to pin this file against the
# reference run under CPU autocast (tests/test_oracle_amp_golden.py)
CONV_BIAS = "separate"

# True: every reduction (conv input channels, Linear / attention k) runs over the REVERSED index order -- the same
# arithmetic policy with a different fp32 summation order.  The distance between the two evaluations is the floor any
# other implementation of the policy (cuDNN, MFMA kernels) can be expected to reach (tests/test_gpu_amp.py).
REVERSED_SUMS = False

# True: the EXACTLY-ROUNDED evaluation of the same policy -- the yardstick the GPU parity gates measure every implementation
# against (tests/test_gpu_amp.py, tests/golden/make_golden_acc64.py).  Every reduction (convolution, Linear, the two
# attention products, softmax sums, LayerNorm statistics, token means) is accumulated in float64 from the same fp16-valued
# operands -- products of fp16 values are exact, so a float64 sum of <= 4608 of them is the exact sum to ~1e-13 relative --
# and rounded ONCE to the dtype the policy holds it in (the fp32 accumulator / fp32 tensor), then through the policy's own
# fp16 rounding points.  The fp32-elementwise ops of the policy (BatchNorm affine, exp, LayerNorm affine) are evaluated in
# float64 and rounded to fp32 as well, so that no implementation's choice of fp32 instruction sequence is baked in.  An
# implementation that accumulates in fp32 in ANY order (this file with ACC64 = False, cuDNN, the MFMA kernels) differs from
# it only where its accumulation error moves a value across an fp16 rounding boundary.
ACC64 = False


def _acc(x):
    """operand of a reduction: float64 under ACC64"""
    return x.double() if ACC64 else x


def _f32(x):
    """the reduction's result as the policy holds it: fp32"""
    return x.float()


def _conv(x, sd, p, stride):
    w = r16(sd[p + ".weight"].float())
    if REVERSED_SUMS:
        x, w = x.flip(1), w.flip(1)
    y = _f32(F.conv2d(_acc(x), _acc(w), None, stride=stride, padding=(w.shape[-1] - 1) // 2))
    b = sd.get(p + ".bias")
    if b is None:
        return r16(y)
    if CONV_BIAS == "fused":
        return r16(y + r16(b.float())[None, :, None, None])
    return r16(r16(y) + r16(b.float())[None, :, None, None])


def _bn(x, sd, p):
    return r16(_f32(F.batch_norm(_acc(x), _acc(sd[p + ".running_mean"].float()), _acc(sd[p + ".running_var"].float()),
                                 _acc(sd[p + ".weight"].float()), _acc(sd[p + ".bias"].float()), training=False, eps=1e-5)))


def _conv_bn_relu(x, sd, p, stride):
    x = _conv(x, sd, p + ".net.0", stride)
    if (p + ".net.1.weight") in sd:
        x = _bn(x, sd, p + ".net.1")
    return F.relu(x)


def _basic_block(x, sd, p):
    y = _conv(x, sd, p + ".conv1", 1)
    if (p + ".bn1.weight") in sd:
        y = _bn(y, sd, p + ".bn1")
    y = _conv(F.relu(y), sd, p + ".conv2", 1)
    if (p + ".bn2.weight") in sd:
        y = _bn(y, sd, p + ".bn2")
    return F.relu(r16(y + x))


def _linear(x, w, b):
    """fp16 GEMM with the bias in the epilogue: one rounding"""
    x, w = r16(x), r16(w.float())
    if REVERSED_SUMS:
        x, w = x.flip(-1), w.flip(-1)
    return r16(_f32(F.linear(_acc(x), _acc(w), _acc(r16(b.float())))))


def encoder_tokens(A, B, sd, stem, joint, trace=None):
    """-> fp32 token stream (n, 400, 512) = fp16 features + fp32 positional table"""
    n = A.shape[0]
    x = r16(torch.cat([A, B], dim=0).float())
    x = _conv_bn_relu(x, sd, stem + ".0", 2)
    if trace is not None:
        trace["conv1"] = x
    x = _conv_bn_relu(x, sd, stem + ".1", 2)
    x = _basic_block(x, sd, stem + ".2")
    x = _basic_block(x, sd, stem + ".3")
    if trace is not None:
        trace["stem"] = x
    ab = torch.cat((x[:n], x[n:]), dim=1)
    ab = _basic_block(ab, sd, joint + ".0")
    ab = _basic_block(ab, sd, joint + ".1")
    ab = _conv_bn_relu(ab, sd, joint + ".2", 2)
    ab = _basic_block(ab, sd, joint + ".3")
    ab = _basic_block(ab, sd, joint + ".4")
    tok = ab.reshape(n, ab.shape[1], -1).permute(0, 2, 1)
    if trace is not None:
        trace["tok16"] = tok
    return tok + sd["pos_embed.pe"].float()[:, : tok.shape[1]]


def _heads(t, Bn, L, nhead, hd):
    return t.reshape(Bn, L, nhead, hd).permute(0, 2, 1, 3)


def attention_flash(qkv, nhead=4):
    """F.scaled_dot_product_attention on fp16 q, k, v (flash order): qkv (Bn, L, 3D) fp16 values -> (Bn, L, D) fp16 values"""
    Bn, L, D3 = qkv.shape
    D = D3 // 3
    hd = D // nhead
    q, k, v = (_heads(t, Bn, L, nhead, hd) for t in qkv.split(D, dim=-1))
    s = _f32(_acc(q) @ _acc(k).transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    p = _f32(torch.exp(_acc(s - s.amax(dim=-1, keepdim=True))))
    o = _f32((_acc(r16(p)) @ _acc(v)) / _acc(p).sum(dim=-1, keepdim=True))
    return r16(o.permute(0, 2, 1, 3).reshape(Bn, L, D))


def attention_explicit(qkv, nhead=4):
    """the need_weights=True branch of F.multi_head_attention_forward under autocast: q * sqrt(1/d) -> fp16, bmm -> fp16
    scores, fp32 softmax, fp16 probabilities, bmm -> fp16"""
    Bn, L, D3 = qkv.shape
    D = D3 // 3
    hd = D // nhead
    q, k, v = (_heads(t, Bn, L, nhead, hd) for t in qkv.split(D, dim=-1))
    qs = r16(q * torch.tensor(math.sqrt(1.0 / float(hd)), dtype=torch.float32))
    s = r16(_f32(_acc(qs) @ _acc(k).transpose(-1, -2)))
    p = r16(_f32(torch.softmax(_acc(s), dim=-1)))
    return r16(_f32(_acc(p) @ _acc(v)).permute(0, 2, 1, 3).reshape(Bn, L, D))


def mha(x, sd, p, explicit, nhead=4):
    """nn.MultiheadAttention(batch_first=True) as att(x, x, x): x fp32 or fp16-valued (Bn, L, 512) -> fp16 values"""
    qkv = _linear(x, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"])
    ctx = attention_explicit(qkv, nhead) if explicit else attention_flash(qkv, nhead)
    return _linear(ctx, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _ln(x, sd, p):
    return _f32(F.layer_norm(_acc(x), (x.shape[-1],), _acc(sd[p + ".weight"].float()), _acc(sd[p + ".bias"].float()), 1e-5))


def encoder_layer(x, sd, p):
    """fp32 residual stream in, fp32 out"""
    x = _ln(x + mha(x, sd, p + ".self_attn", explicit=False), sd, p + ".norm1")
    ff = _linear(F.relu(_linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])), sd[p + ".linear2.weight"],
                 sd[p + ".linear2.bias"])
    return _ln(x + ff, sd, p + ".norm2")


@torch.no_grad()
def refine_forward(A, B, sd, trace=None):
    tok = encoder_tokens(A, B, sd, "encodeA", "encodeAB", trace)
    out = {}
    for name in ("trans", "rot"):
        h = encoder_layer(tok, sd, f"{name}_head.0")
        y = _linear(h, sd[f"{name}_head.1.weight"], sd[f"{name}_head.1.bias"])   # (n, 400, 3|6) fp16 values
        out[name] = r16(_f32(_acc(y).mean(dim=1)))
    return out


@torch.no_grad()
def score_features(A, B, sd, trace=None):
    tok = encoder_tokens(A, B, sd, "encoderA", "encoderAB", trace)
    return r16(_f32(_acc(mha(tok, sd, "att", explicit=True)).mean(dim=1)))


@torch.no_grad()
def score_forward(A, B, sd, L, trace=None):
    feats = score_features(A, B, sd, trace)
    bs = A.shape[0] // L
    x = mha(feats.reshape(bs, L, -1), sd, "att_cross", explicit=True)
    return {"score_logit": _linear(x, sd["linear.weight"], sd["linear.bias"]).reshape(bs, L)}
