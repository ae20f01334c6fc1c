"""Inference plans for RefineNet / ScoreNetMultiPair, built once from a module's state_dict (reference checkpoints load
unchanged).

precision='fp16' -- the deployed configuration, the reference's `torch.cuda.amp.autocast` (predict_pose_refine.py:190-191,
predict_score.py:193-194) -- runs the whole network on libfp_amd.so and follows autocast's op sequence rounding point
by rounding point (the restatement the tests check against is oracle/nets_amp.py):
  * patch-embed conv (7x7 s2, 6->64) + BatchNorm + ReLU          -> fp_conv7x7s2_bn_relu_fwd
  * the 15 3x3 convs: fp16(conv) + bias -> fp16, BatchNorm -> fp16, (+ identity -> fp16), ReLU
                                                                -> fp_igemm_f16_fwd (FP_IGEMM_ROUND_ACC)
  * positional table: fp16 tokens + fp32 table = the fp32 residual stream; its fp16 rounding feeds in_proj
                                                                -> second output of the last conv's epilogue (fp_igemm_epilogue.pe;
                                                                   the fp32 sum itself is recomputed by the LayerNorm)
  * every 512-wide Linear (in_proj, out_proj, linear1, linear2): one rounding of accumulator + bias
                                                                -> fp_igemm_f16_fwd
  * self-attention between in_proj and out_proj                  -> fp_attention_f16_fwd (flash order = F.scaled_dot_product_
                                                                   attention; FP_ATT_FP16_SCORES for the scorer's direct
                                                                   nn.MultiheadAttention calls)
  * x + sa -> LayerNorm on the fp32 stream, written as fp32 (next residual) and fp16 (next GEMM operand)
                                                                -> fp_layernorm_res_fwd
  * x + ff -> LayerNorm -> mean over the 400 tokens              -> fp_colmean_f16_fwd (the normalised tensor is never written)
  * the Linear layers behind a token mean (3|6-wide heads, the scorer's out_proj and final Linear) on N rows
                                                                -> fp_rows_linear_fwd
Two algebraic rearrangements against the literal op order, both checked against the reference-under-autocast goldens
(tests/test_oracle_amp_golden.py): the token mean is taken before the Linear it follows (refine_network.py:90-91,
score_network.py:73-74), and attention never materialises the (N*4,400,400) probabilities.
No PyTorch compute kernel is left on this plan (allocations only).

precision='fp32' is the fp32 parity configuration: plain torch ops, no autocast.

precision='torch_amp' is BASELINE configs[1] taken literally -- "HIP rasteriser + PyTorch-ROCm refine/score": the nn.Module
itself (refine_network.py / score_network.py, the reference's module tree) under torch.autocast('cuda', float16), i.e.
MIOpen / rocBLAS / ATen kernels behind the same predictor.  It is the independent third implementation of the autocast
policy that tests/test_gpu_amp.py puts next to the HIP plan and the oracle, and `bench.py --precision torch_amp`.
"""
import contextlib

import torch
import torch.nn.functional as F

from . import ops

# MIOpen in this image has no tuned / pre-compiled gfx950 kernels for these layers and falls back to its naive reference
# convolution (naive_conv_*_fwd_*: ~0.7 s per layer call at 504 images, measured in round 1; one fp32 pass of the encoder at
# N=252 takes minutes).  The two torch configurations therefore run their convolutions on ATen's own path (im2col +
# rocBLAS GEMM, what PyTorch uses when `torch.backends.cudnn` is disabled) unless MIOpen is asked for explicitly.
USE_MIOPEN = False


# out_proj + residual + LayerNorm of the refiner's encoder layers as ONE kernel (csrc/linear_ln.hip, fp_linear_layernorm_fwd):
# bit-identical to the two launches it replaces (tests/test_gpu_parity.py::test_linear_layernorm_is_the_two_kernel_path) and
# 0.6-1.4 % of the step faster.  overrides(FUSED_OUT_PROJ_LN=False) goes back to fp_igemm_f16_fwd + fp_layernorm_res_fwd.
FUSED_OUT_PROJ_LN = True


# linear1 + ReLU + linear2 + residual + norm2 + token mean of the refiner's encoder layers as ONE launch (+ a finish kernel;
# csrc/linear_ln.hip, fp_ffn_layernorm_mean_fwd): the two (M, 512) intermediates stay in LDS.  Same rounding points; the token
# mean is summed in another fixed fp32 order, so it passes the parity gates against the exactly-rounded yardstick
# (tests/test_gpu_amp.py) rather than an equality test: distances to the yardstick unchanged to three digits
# (profiles/r04_parity_amp.json against r04_parity_amp_fused_ffn.json), 35.9 -> 35.1 ms per bench step (profiles/r04_b_bench_*).
# overrides(FUSED_FFN=False) goes back to 2 x fp_igemm_f16_fwd + fp_colmean_f16_fwd.
FUSED_FFN = True


# The in_proj of the self-attention blocks (512 -> 1536) on the row-owning tile (csrc/linear_ln.hip, fp_linear512_f16_fwd): the 128 x 512
# input tile of a workgroup is fetched once for the three column blocks, weights come fragment-packed from L2 into registers.  The bits
# of fp_igemm_f16_fwd (tests/test_gpu_parity.py::test_linear512_is_the_igemm_linear).  overrides(ROWS_QKV=False) goes back to
# fp_igemm_f16_fwd; launches below ROWS_QKV_MIN_ROWS rows (the scorer's cross-hypothesis attention: 252 rows) stay there as well.
ROWS_QKV = True
ROWS_QKV_MIN_ROWS = 4096
# round 5: the encoder's stride-1 3x3 convolutions hand fp_igemm_f16_fwd a tile-packed copy of their weights as well (epilogue.w_tiles): the
# shifted-window kernel then stages a k-step's weight tile from one contiguous 8 KiB run.  Same operands, same order: the same bits.
PACKED_CONV_TILES = True
# round 6: RefineNet's trans_head and rot_head read the same tokens: their in_proj as one 3072-wide launch, their attention as one 8-head
# launch (RefinePlan.__init__); the same bits as two calls each.  overrides(MERGED_HEAD_QKV=False) goes back to them.
MERGED_HEAD_QKV = True
# round 6: the whole encoder layer behind the attention context as ONE launch (csrc/linear_ln.hip k_rows512<.., TAIL>): out_proj + norm1
# and the fused FFN + norm2 + token mean above, with norm1's fp16 output staying in LDS between them.  Same bits as the two launches.
FUSED_TAIL = True


def _conv_backend():
    if USE_MIOPEN or not torch.backends.cudnn.is_available():
        return contextlib.nullcontext()
    return torch.backends.cudnn.flags(enabled=False)


def _bn_affine(sd, bn_p):
    """eval BatchNorm2d as x * scale + shift (fp32), or (None, None)"""
    if bn_p is None or (bn_p + ".weight") not in sd:
        return None, None
    scale = sd[bn_p + ".weight"].float() / torch.sqrt(sd[bn_p + ".running_var"].float() + 1e-5)
    shift = sd[bn_p + ".bias"].float() - sd[bn_p + ".running_mean"].float() * scale
    return scale.contiguous(), shift.contiguous()


def _r16(t):
    """fp32 tensor holding the fp16 rounding of t (autocast casts parameters to fp16)"""
    return t.to(torch.float16).to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------ fp32 torch plan
class _Conv:
    """conv (+ eval BN folded into weights and bias) (+ReLU), torch ops"""

    def __init__(self, sd, conv_p, bn_p, stride, dtype, channels_last):
        w = sd[conv_p + ".weight"].float()
        b = sd.get(conv_p + ".bias")
        b = torch.zeros(w.shape[0], device=w.device) if b is None else b.float()
        scale, shift = _bn_affine(sd, bn_p)
        if scale is not None:
            w, b = w * scale[:, None, None, None], b * scale + shift
        wf = w.to(dtype)
        self.w = wf.contiguous(memory_format=torch.channels_last) if channels_last else wf.contiguous()
        self.b = b.to(dtype)
        self.stride = stride
        self.pad = (w.shape[-1] - 1) // 2

    def __call__(self, x, relu=True):
        y = F.conv2d(x, self.w, self.b, stride=self.stride, padding=self.pad)
        return F.relu_(y) if relu else y


class _Block:
    def __init__(self, sd, p, dtype, cl):
        self.c1 = _Conv(sd, p + ".conv1", p + ".bn1", 1, dtype, cl)
        self.c2 = _Conv(sd, p + ".conv2", p + ".bn2", 1, dtype, cl)

    def __call__(self, x):
        y = self.c2(self.c1(x), relu=False)
        y += x
        return F.relu_(y)


class _Encoder:
    """stem on cat(A,B) along batch + joint encoder on the channel concat -> tokens (N,400,512) + PE (torch ops)"""

    def __init__(self, sd, stem, joint, dtype, channels_last):
        cl = channels_last
        self.dtype, self.cl = dtype, cl
        self.c1 = _Conv(sd, stem + ".0.net.0", stem + ".0.net.1", 2, dtype, cl)
        self.c2 = _Conv(sd, stem + ".1.net.0", stem + ".1.net.1", 2, dtype, cl)
        self.s2, self.s3 = _Block(sd, stem + ".2", dtype, cl), _Block(sd, stem + ".3", dtype, cl)
        self.j0, self.j1 = _Block(sd, joint + ".0", dtype, cl), _Block(sd, joint + ".1", dtype, cl)
        self.j2 = _Conv(sd, joint + ".2.net.0", joint + ".2.net.1", 2, dtype, cl)
        self.j3, self.j4 = _Block(sd, joint + ".3", dtype, cl), _Block(sd, joint + ".4", dtype, cl)
        self.pe = sd["pos_embed.pe"].to(dtype)

    def __call__(self, AB):
        """AB: (2N,6,H,W) -- A in the first half, B in the second."""
        n = AB.shape[0] // 2
        if self.cl:
            AB = AB.contiguous(memory_format=torch.channels_last)
        x = self.s3(self.s2(self.c2(self.c1(AB))))
        ab = torch.cat((x[:n], x[n:]), dim=1)
        ab = self.j4(self.j3(self.j2(self.j1(self.j0(ab)))))
        tok = ab.permute(0, 2, 3, 1).reshape(n, -1, ab.shape[1])  # free view when channels_last
        return tok + self.pe[:, : tok.shape[1]]


class _TorchMHA:
    """self-attention of nn.MultiheadAttention(512, 4, batch_first=True) called as att(x, x, x), torch ops"""

    def __init__(self, sd, p, dtype, nhead=4):
        self.wi, self.bi = sd[p + ".in_proj_weight"].to(dtype), sd[p + ".in_proj_bias"].to(dtype)
        self.wo, self.bo = sd[p + ".out_proj.weight"].to(dtype), sd[p + ".out_proj.bias"].to(dtype)
        self.nhead = nhead

    def __call__(self, x):
        Bn, L, D = x.shape
        hd = D // self.nhead
        qkv = F.linear(x, self.wi, self.bi).reshape(Bn, L, 3, self.nhead, hd)
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        ctx = F.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(Bn, L, D)
        return F.linear(ctx, self.wo, self.bo)


class _TorchEncoderLayer:
    """nn.TransformerEncoderLayer(512, 4, 512, batch_first=True) in eval mode: post-norm, ReLU, LN eps 1e-5"""

    def __init__(self, sd, p, dtype):
        self.att = _TorchMHA(sd, p + ".self_attn", dtype)
        self.l1 = (sd[p + ".linear1.weight"].to(dtype), sd[p + ".linear1.bias"].to(dtype))
        self.l2 = (sd[p + ".linear2.weight"].to(dtype), sd[p + ".linear2.bias"].to(dtype))
        self.n1 = (sd[p + ".norm1.weight"].to(dtype), sd[p + ".norm1.bias"].to(dtype))
        self.n2 = (sd[p + ".norm2.weight"].to(dtype), sd[p + ".norm2.bias"].to(dtype))

    def __call__(self, x):
        x = F.layer_norm(x + self.att(x), (x.shape[-1],), self.n1[0], self.n1[1], 1e-5)
        ff = F.linear(F.relu(F.linear(x, *self.l1)), *self.l2)
        return F.layer_norm(x + ff, (x.shape[-1],), self.n2[0], self.n2[1], 1e-5)


# ------------------------------------------------------------------------------------------------ HIP plan (fp16 autocast policy)
def _conv_params(sd, conv_p, bn_p):
    """-> dict(w (Cout, kh*kw*Cin) fp16 with k ordered (ky, kx, ci), bias f32 (fp16-rounded) | None, scale, shift | None)"""
    w = sd[conv_p + ".weight"].float()
    b = sd.get(conv_p + ".bias")
    scale, shift = _bn_affine(sd, bn_p)
    wk = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(torch.float16).contiguous()
    # round 5: the tile-packed copy the shifted-window kernel fetches as contiguous 8 KiB runs (fp_pack_conv3x3_tiles_f16); 3x3 only
    tiles = None
    if wk.is_cuda and w.shape[2] == 3 and w.shape[3] == 3 and w.shape[0] % 128 == 0 and w.shape[1] % 32 == 0 and PACKED_CONV_TILES:
        tiles = ops.pack_conv3x3_tiles(wk, w.shape[0], w.shape[1])
    return dict(w=wk, w_tiles=tiles, bias=None if b is None else _r16(b.float()), scale=scale, shift=shift)


# Round 5: split-K for the reference's tracking call.  With ONE hypothesis (estimater.py:250-268) the encoder's convolutions are
# launches of 16-26 tiles on 256 CUs, each tile running its whole k loop (55 us per launch on average, 73 % of track_one's GPU time);
# fp_igemm_f16_splitk_fwd cuts the k range instead.  The decision depends on the HYPOTHESIS COUNT of the call only (not on the image
# count of a launch: the shared-observed-crop form of the stem must keep the summation order of the plain form) and the number of
# pieces is a function of the layer and of that count (splitk_pieces), so the two-pose quirk, the shared-crop stem and graph replays
# of one call all see the same arithmetic; calls that are split into sub-batches never come here (see below).
# measured (scripts/bench_small_batches.py, profiles/r05_k_small_batches.log): predict(n, 2 iterations) 1.90 -> 1.05 ms at n = 1, 1.90 -> 1.21 at
# 4, 1.93 -> 1.39 at 8, 2.02 -> 1.71 at 12, 2.05 -> 1.96 at 16, slower from 24 on.  Must stay below the sub-batch minimum (overlap.SubBatches
# min_rows = 32): a call that is split into sub-batches never takes this path, so the parts of a call and the whole call always agree.
SPLITK_MAX_HYPS = 12
HEADS_TWO_STREAMS_MAX_HYPS = 12   # RefinePlan: see __call__
SPLITK_TARGET_WGS = 384        # (tile, piece) workgroups a launch should have: 1.5 per CU
# the hard ceiling of both thresholds: overlap.SubBatches' default min_rows (a call of >= 2 x 32 hypotheses is split into sub-batches)
SMALL_CALL_CEILING = 31

_SWITCHES = ("FUSED_OUT_PROJ_LN", "FUSED_FFN", "ROWS_QKV", "PACKED_CONV_TILES", "MERGED_HEAD_QKV", "FUSED_TAIL", "SPLITK_MAX_HYPS", "HEADS_TWO_STREAMS_MAX_HYPS")


@contextlib.contextmanager
def overrides(**kw):
    """Test / A-B hook (round 6: the release package reads NO environment variable -- which kernels and which summation order a
    call runs are fixed by the constants above).  `with engine.overrides(SPLITK_MAX_HYPS=0): ...` changes a switch for plans BUILT
    AND RUN inside the block and restores it; the small-call thresholds are refused above SMALL_CALL_CEILING, where the parts of a
    sub-batched call and the whole call would no longer see one arithmetic."""
    g = globals()
    for k, v in kw.items():
        if k not in _SWITCHES:
            raise KeyError(f"engine.overrides: unknown switch {k!r} (known: {', '.join(_SWITCHES)})")
        if k.endswith("_MAX_HYPS") and not (0 <= int(v) <= SMALL_CALL_CEILING):
            raise ValueError(f"engine.overrides: {k} = {v} is outside 0..{SMALL_CALL_CEILING}")
    saved = {k: g[k] for k in kw}
    g.update(kw)
    try:
        yield
    finally:
        g.update(saved)


def small_call(n, limit):
    """is a call of n hypotheses a small call under threshold `limit`?  Clamped at use: whatever a test set, never at or above the
    sub-batch minimum"""
    return n <= min(int(limit), SMALL_CALL_CEILING)


def splitk_pieces(rows, cout, cin):
    """pieces of the 9 * cin / 64 k-steps for a 3x3 convolution of `rows` output pixels: enough (tile, piece) workgroups to fill the
    chip, at least two k-steps per piece.  A function of the layer and of the hypothesis count of the CALL (rows = images x pixels;
    the shared-observed-crop stem passes the image count of the plain form), so every launch of one call sees the same pieces."""
    tiles = -(-rows // 128) * (cout // 128)
    nk = 9 * cin // 64
    return max(1, min(nk // 2, round(SPLITK_TARGET_WGS / tiles)))


class _HipEncoder:
    """The encoder on libfp_amd.so only: patch-embed conv (fp_conv7x7s2_bn_relu_fwd) + 15 implicit-GEMM 3x3 convs
    (fp_igemm_f16_fwd) with the conv / BatchNorm / residual / ReLU rounding sequence of autocast in their epilogues.
    Activations are NHWC fp16 with a 1-pixel zero border; one buffer set per (batch, H, W, slot) is allocated on first use
    and kept (a captured hipGraph holds raw pointers into it, so sets are never dropped or reallocated; `slot` separates
    callers that run concurrently on different streams); the A|B channel
    concat is a strided store of the stem's last conv (refine_network.py:82-85), and the last conv writes the
    (N, 400, 512) token matrix directly."""

    def __init__(self, sd, stem, joint, device):
        c1 = sd[stem + ".0.net.0.weight"].float()
        self.c1_w = c1.to(torch.float16).reshape(64, -1).contiguous()            # PyTorch layout (64, 6*7*7)
        b1 = sd.get(stem + ".0.net.0.bias")
        self.c1_b = None if b1 is None else _r16(b1.float())
        self.c1_scale, self.c1_shift = _bn_affine(sd, stem + ".0.net.1")
        names = [("c2", stem + ".1.net.0", stem + ".1.net.1")]
        for blk, pre in (("s2", stem + ".2"), ("s3", stem + ".3"), ("j0", joint + ".0"), ("j1", joint + ".1"),
                         ("j3", joint + ".3"), ("j4", joint + ".4")):
            names += [(blk + "a", pre + ".conv1", pre + ".bn1"), (blk + "b", pre + ".conv2", pre + ".bn2")]
        names.append(("j2", joint + ".2.net.0", joint + ".2.net.1"))
        self.w = {k: _conv_params(sd, c, b) for k, c, b in names}
        self.pe = sd["pos_embed.pe"].float().reshape(-1, sd["pos_embed.pe"].shape[-1]).contiguous()   # (400, 512) f32
        self.device = device
        self._bufs = {}

    def _buffers(self, n, H, W, slot, small=True):
        small = bool(small) and small_call(n, SPLITK_MAX_HYPS)
        key = (n, H, W, slot, small)
        b = self._bufs.get(key)
        if b is None:
            z = lambda *shape: torch.zeros(shape, dtype=torch.float16, device=self.device)
            h1, w1, h2, w2, h3, w3 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
            sk = None
            if small:
                need = max(ops.igemm_splitk_workspace_bytes(r, co, splitk_pieces(r, co, ci))
                           for r, co, ci in ((2 * n * h2 * w2, 128, 64), (2 * n * h2 * w2, 128, 128), (n * h2 * w2, 256, 256),
                                             (n * h3 * w3, 512, 256), (n * h3 * w3, 512, 512)))
                sk = torch.empty(need, dtype=torch.uint8, device=self.device)
            b = dict(SK=sk, P1=z(2 * n, h1 + 2, w1 + 2, 64), P2=z(2 * n, h2 + 2, w2 + 2, 128), P3=z(2 * n, h2 + 2, w2 + 2, 128),
                     T=z(2 * n, h2 + 2, w2 + 2, 128), CAT=z(n, h2 + 2, w2 + 2, 256), J0=z(n, h2 + 2, w2 + 2, 256),
                     T2=z(n, h2 + 2, w2 + 2, 256), Q0=z(n, h3 + 2, w3 + 2, 512), Q1=z(n, h3 + 2, w3 + 2, 512),
                     T3=z(n, h3 + 2, w3 + 2, 512), dims=(h1, w1, h2, w2, h3, w3))
            self._bufs[key] = b
        return b

    def _conv(self, name, x, Bn, Ho, Wo, Cin, Cout, y, stride=1, res=None, relu=True, gout=None, gres=None, pe=None, y_pe=None, sk=None,
              sk_rows=None):
        c = self.w[name]
        G = ops.IgemmGeom.image
        gin = G(Ho, Wo, 1, Cin, stride=stride, offset=0)
        gout = gout if gout is not None else G(Ho, Wo, 1, Cout)
        if res is not None and gres is None:
            gres = G(Ho, Wo, 1, Cout)
        if sk is not None:
            return ops.igemm_f16_splitk(x, gin, c["w"], c["bias"], y, gout, Bn * Ho * Wo, Cout, Cin, 9,
                                        splitk_pieces(sk_rows if sk_rows is not None else Bn * Ho * Wo, Cout, Cin), sk,
                                        relu=relu, residual=res, r_geom=gres, bn_scale=c["scale"], bn_shift=c["shift"],
                                        conv_rounding=True, pe=pe, y_pe=y_pe)
        return ops.igemm_f16(x, gin, c["w"], c["bias"], y, gout, Bn * Ho * Wo, Cout, Cin, 9, relu=relu, residual=res, r_geom=gres,
                             bn_scale=c["scale"], bn_shift=c["shift"], conv_rounding=True, pe=pe, y_pe=y_pe,
                             w_tiles=c["w_tiles"] if stride == 1 else None)

    def __call__(self, AB, slot=0, shared_b=False, small_calls=True):
        """AB (2n,6,H,W) fp16 -> (tokens (n, H/8 * W/8, 512) fp16 before the positional table,
        x16 = f16(f32(tokens) + pe): the in_proj operand, written by the same epilogue).
        shared_b: AB is (n+1,6,H,W) -- n rendered crops and ONE observed crop that all n pairs share (the first refine
        iteration of register(): every hypothesis has the same translation, hence the same crop window).  The shared stem
        then runs on n+1 images instead of 2n and the B half of the concat is replicated; per element the same kernels in
        the same order, so the result is bit-identical to feeding n copies."""
        n2, _, H, W = AB.shape
        n = n2 - 1 if shared_b else n2 // 2
        b = self._buffers(n, H, W, slot, small_calls)
        h1, w1, h2, w2, h3, w3 = b["dims"]
        G = ops.IgemmGeom.image
        ops.conv7x7s2_bn_relu(AB, self.c1_w, self.c1_b, self.c1_scale, self.c1_shift, b["P1"][:n2], 1)
        self._conv("c2", b["P1"], n2, h2, w2, 64, 128, b["P2"], stride=2, sk=b["SK"], sk_rows=2 * n * h2 * w2)
        self._conv("s2a", b["P2"], n2, h2, w2, 128, 128, b["T"], sk=b["SK"], sk_rows=2 * n * h2 * w2)
        self._conv("s2b", b["T"], n2, h2, w2, 128, 128, b["P3"], res=b["P2"], sk=b["SK"], sk_rows=2 * n * h2 * w2)
        self._conv("s3a", b["P3"], n2, h2, w2, 128, 128, b["T"], sk=b["SK"], sk_rows=2 * n * h2 * w2)
        # stem output of image i (A) and image n+i (B) side by side along C: torch.cat((a, b), 1)
        self._conv("s3b", b["T"], n2, h2, w2, 128, 128, b["CAT"], res=b["P3"],
                   gout=G(h2, w2, 1, 256, bsplit=n, cgroup=128), gres=G(h2, w2, 1, 128), sk=b["SK"], sk_rows=2 * n * h2 * w2)
        if shared_b and n > 1:
            ops.replicate_channels(b["CAT"], n, 128, 256)
        self._conv("j0a", b["CAT"], n, h2, w2, 256, 256, b["T2"], sk=b["SK"])
        self._conv("j0b", b["T2"], n, h2, w2, 256, 256, b["J0"], res=b["CAT"], sk=b["SK"])
        self._conv("j1a", b["J0"], n, h2, w2, 256, 256, b["T2"], sk=b["SK"])
        self._conv("j1b", b["T2"], n, h2, w2, 256, 256, b["CAT"], res=b["J0"], sk=b["SK"])
        self._conv("j2", b["CAT"], n, h3, w3, 256, 512, b["Q0"], stride=2, sk=b["SK"])
        self._conv("j3a", b["Q0"], n, h3, w3, 512, 512, b["T3"], sk=b["SK"])
        self._conv("j3b", b["T3"], n, h3, w3, 512, 512, b["Q1"], res=b["Q0"], sk=b["SK"])
        self._conv("j4a", b["Q1"], n, h3, w3, 512, 512, b["T3"], sk=b["SK"])
        tok = torch.empty((n, h3 * w3, 512), dtype=torch.float16, device=AB.device)
        x16 = torch.empty_like(tok)
        self._conv("j4b", b["T3"], n, h3, w3, 512, 512, tok, res=b["Q1"], gout=G(h3, w3, 0, 512), gres=G(h3, w3, 1, 512),
                   pe=self.pe[: h3 * w3], y_pe=x16, sk=b["SK"])
        return tok, x16


class _HipLinear:
    """nn.Linear under autocast on fp_igemm_f16_fwd: fp16 operands, fp32 accumulation + bias, one rounding"""

    def __init__(self, w, b):
        assert w.shape[0] % 128 == 0 and w.shape[1] % 64 == 0
        self.w = w.to(torch.float16).contiguous()
        self.b = _r16(b.float())

    def __call__(self, x, relu=False):
        x2 = x.reshape(-1, x.shape[-1])
        M, K = x2.shape
        No = self.w.shape[0]
        y = torch.empty((M, No), dtype=torch.float16, device=x.device)
        Gm = ops.IgemmGeom.matrix
        ops.igemm_f16(x2, Gm(K), self.w, self.b, y, Gm(No), M, No, K, 1, relu=relu)
        return y.reshape(*x.shape[:-1], No)


class _HipRowsLinear:
    """a Linear applied to N pooled rows (fp32 in, fp16 weights, fp32 accumulation)"""

    def __init__(self, w, b):
        self.w = w.to(torch.float16).contiguous()
        self.b = _r16(b.float())

    def __call__(self, x, round_f16=True, out_f16=False, out=None):
        return ops.rows_linear(x, self.w, self.b, round_f16=round_f16, out_f16=out_f16, out=out)


class _HipMHA:
    def __init__(self, sd, p, fp16_scores, nhead=4, small_calls=True):
        self.qkv = _HipLinear(sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"])
        self.out = _HipLinear(sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])
        self.out_rows = _HipRowsLinear(sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])
        self.nhead, self.fp16_scores = nhead, fp16_scores
        self.qkv_p = ops.PackedLinear512(self.qkv.w) if tuple(self.qkv.w.shape) == (1536, 512) else None
        # small calls: the in_proj of a call of <= SPLITK_MAX_HYPS sequences is a 48-tile launch of 8 k-steps each; split
        # in four it fills the chip.  Not for the scorer's cross-hypothesis attention (one "sequence" of N hypotheses: its row count is
        # the hypothesis count of the call, and every rank of a sharded call has to see the same arithmetic there).
        self.small_calls = small_calls
        self._sk = {}

    def context(self, x16, slot=0, small_calls=True):
        """softmax(q k^T / sqrt(d)) v with heads merged, before the output projection: (Bn, L, D) fp16.  slot: as for the encoder's
        activation sets -- callers that overlap on different streams must not share the split-K slab"""
        Bn, L, D = x16.shape
        if ROWS_QKV and self.qkv_p is not None and Bn * L >= ROWS_QKV_MIN_ROWS:
            qkv = ops.linear512(x16, self.qkv_p, self.qkv.b)
        elif self.small_calls and small_calls and small_call(Bn, SPLITK_MAX_HYPS) and D % 64 == 0 and D // 64 >= 4:
            M, No = Bn * L, self.qkv.w.shape[0]
            pieces = max(1, min(D // 64 // 2, round(SPLITK_TARGET_WGS / (-(-M // 128) * (No // 128)))))
            key = (M, No, pieces, x16.device, slot)
            ws = self._sk.get(key)
            if ws is None:
                ws = self._sk[key] = torch.empty(ops.igemm_splitk_workspace_bytes(M, No, pieces), dtype=torch.uint8, device=x16.device)
            qkv = torch.empty((M, No), dtype=torch.float16, device=x16.device)
            Gm = ops.IgemmGeom.matrix
            ops.igemm_f16_splitk(x16.reshape(M, D), Gm(D), self.qkv.w, self.qkv.b, qkv, Gm(No), M, No, D, 1, pieces, ws)
        else:
            qkv = self.qkv(x16)
        return ops.attention_f16(qkv.reshape(Bn, L, 3 * D), self.nhead, fp16_scores=self.fp16_scores)

    def __call__(self, x16, slot=0, small_calls=True):
        return self.out(self.context(x16, slot, small_calls))


class _HipEncoderLayer:
    """nn.TransformerEncoderLayer under autocast, ending in the token mean: the layer input is the fp32 stream
    f32(tok16) + pe, which exists only inside the kernels that consume it"""

    def __init__(self, sd, p):
        self.att = _HipMHA(sd, p + ".self_attn", fp16_scores=False)
        self.l1 = _HipLinear(sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])
        self.l2 = _HipLinear(sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
        self.n1 = (sd[p + ".norm1.weight"].float().contiguous(), sd[p + ".norm1.bias"].float().contiguous())
        self.n2 = (sd[p + ".norm2.weight"].float().contiguous(), sd[p + ".norm2.bias"].float().contiguous())
        # fragment-packed copies for the row-owning fused kernels (csrc/linear_ln.hip), which read weights from L2 into registers
        self.out_p, self.l1_p, self.l2_p = (ops.PackedLinear512(m.w) for m in (self.att.out, self.l1, self.l2))

    def pooled(self, tok16, x16, pe, slot=0, small_calls=True):
        """-> mean over the tokens of the layer output, (N, 512) fp32"""
        return self.pooled_from_context(self.att.context(x16, slot, small_calls), tok16, pe)

    def pooled_from_context(self, ctx16, tok16, pe):
        """the layer behind the attention context (heads merged, before out_proj): ctx16 (N, L, 512) fp16, possibly a column block of a
        wider tensor (RefinePlan's two heads as one 8-head attention call)"""
        if FUSED_TAIL and FUSED_OUT_PROJ_LN and FUSED_FFN and ctx16.shape[1] % 16 == 0 and tuple(pe.shape) == (ctx16.shape[1], 512):
            # round 6: out_proj + norm1 + linear1 + ReLU + linear2 + norm2 + token mean in ONE launch (fp_encoder_tail_mean_fwd): the bits
            # of the two launches below, norm1's fp16 output never leaves LDS
            return ops.encoder_tail_mean(ctx16, self.out_p, self.att.out.b, tok16, pe, self.n1[0], self.n1[1], self.l1_p, self.l1.b,
                                         self.l2_p, self.l2.b, self.n2[0], self.n2[1], 1e-5)
        if FUSED_OUT_PROJ_LN:
            # out_proj + residual + norm1 in one launch, the projection staying on chip (fp_linear_layernorm_fwd): the same bits
            y32, y16 = ops.linear_layernorm_res(ctx16, self.out_p, self.att.out.b, self.n1[0], self.n1[1], 1e-5, tok16=tok16, pe=pe)
        else:
            sa = self.att.out(ctx16.contiguous())                                # fp16
            y32, y16 = ops.layernorm_res(sa, self.n1[0], self.n1[1], 1e-5, tok16=tok16, pe=pe)   # LN(x + sa): fp32 stream + fp16 copy
        if FUSED_FFN and y16.shape[1] % 16 == 0:
            return ops.ffn_layernorm_mean(y16, self.l1_p, self.l1.b, self.l2_p, self.l2.b, y32, self.n2[0], self.n2[1], 1e-5)
        ff = self.l2(self.l1(y16, relu=True))
        return ops.colmean_f16(ff, self.n2[0], self.n2[1], 1e-5, resid32=y32)    # mean_t LN(y + ff)


def _check_precision(precision):
    if precision not in ("fp16", "fp32", "torch_amp"):
        raise ValueError(f"precision must be 'fp16', 'fp32' or 'torch_amp', got {precision!r}")


def _dev_sd(module_or_sd, device):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    return {k: v.detach().to(device) for k, v in sd.items()}


class RefinePlan:
    # False: a call of <= SPLITK_MAX_HYPS hypotheses runs the kernels (and the summation order) of a large call -- what the shards of
    # a multi-rank call ask for, so that a shard returns the bits of the single batch whatever its size (dist.py)
    small_calls = True

    def __init__(self, model, device, precision="fp16", channels_last=True):
        _check_precision(precision)
        self.dtype = torch.float16 if precision == "fp16" else torch.float32
        self.hip = self.dtype == torch.float16
        self.module = model if precision == "torch_amp" else None
        if self.module is not None:
            return
        sd = _dev_sd(model, device)
        self.heads = {}
        if self.hip:
            self.enc = _HipEncoder(sd, "encodeA", "encodeAB", device)
            for name in ("trans", "rot"):
                self.heads[name] = (_HipEncoderLayer(sd, f"{name}_head.0"),
                                    _HipRowsLinear(sd[f"{name}_head.1.weight"], sd[f"{name}_head.1.bias"]))
            # round 6: the self-attention of the two heads as ONE in_proj launch (512 -> 3072, the token tile fetched once) and ONE
            # 8-head attention launch.  Rows of the merged weight: [q_trans q_rot | k_trans k_rot | v_trans v_rot], i.e. the layout of an
            # 8-head nn.MultiheadAttention whose heads 0-3 are trans_head's and 4-7 rot_head's (refine_network.py:56-70); per element the
            # same dot products in the same order, so the same bits as the two separate calls (tests/test_gpu_parity.py)
            wt, wr = (self.heads[n][0].att.qkv for n in ("trans", "rot"))
            if tuple(wt.w.shape) == (1536, 512) and tuple(wr.w.shape) == (1536, 512):
                cut = lambda t: (t[0:512], t[512:1024], t[1024:1536])
                self.qkv2_p = ops.PackedLinear512(torch.cat([p for pair in zip(cut(wt.w), cut(wr.w)) for p in pair], 0).contiguous())
                self.qkv2_b = torch.cat([p for pair in zip(cut(wt.b), cut(wr.b)) for p in pair], 0).contiguous()
            else:
                self.qkv2_p = None
        else:
            self.enc = _Encoder(sd, "encodeA", "encodeAB", self.dtype, channels_last)
            for name in ("trans", "rot"):
                self.heads[name] = (_TorchEncoderLayer(sd, f"{name}_head.0", self.dtype),
                                    sd[f"{name}_head.1.weight"].to(self.dtype), sd[f"{name}_head.1.bias"].to(self.dtype))

    def _head_side_stream(self, tok16):
        """the side stream the rotation head of a SMALL call runs on, or None: calls of more than HEADS_TWO_STREAMS_MAX_HYPS
        hypotheses fill the chip by themselves (and the side stream belongs to their sub-batches), a per-kernel timing pass wants
        one stream, and a side stream that shares the main stream's hardware queue buys nothing"""
        if not getattr(self, "two_stream_heads", True) or not self.small_calls or not small_call(tok16.shape[0], HEADS_TWO_STREAMS_MAX_HYPS) \
                or tok16.device.type != "cuda" or ops.KernelTimers.active is not None:
            return None
        from . import overlap
        # the probe of a side stream (spin kernels, the rasteriser canary) synchronises: never inside a stream capture.  The predictors
        # reserve the stream when they build the plan; a capture that still gets here first stays on one stream
        if torch.cuda.is_current_stream_capturing() and not overlap.reserved(tok16.device, 1):
            return None
        if not overlap.side_streams_overlap(tok16.device, 1):
            return None
        return overlap.reserve_streams(tok16.device, 1)[0]

    @torch.inference_mode()
    def __call__(self, AB, slot=0, shared_b=False):
        """AB (2N,6,H,W) in the plan's dtype -> {'trans': (N,3) f32, 'rot': (N,3|6) f32}.  slot: activation-buffer set
        (callers that overlap on different streams use different slots).  shared_b (fp16 plan only): AB is (N+1,6,H,W),
        the last image being the observed crop every pair shares (_HipEncoder.__call__)"""
        out = {}
        if shared_b and not self.hip:
            raise ValueError("shared_b is a property of the fp16 plan; expand the observed crop for the torch plans")
        if self.module is not None:
            n = AB.shape[0] // 2
            with _conv_backend(), torch.autocast("cuda", dtype=torch.float16):          # predict_pose_refine.py:190-191
                o = self.module(AB[:n], AB[n:])
            return {k: v.float() for k, v in o.items()}                 # predict_pose_refine.py:192-193
        if self.hip:
            sc = self.small_calls
            tok16, x16 = self.enc(AB, slot, shared_b=shared_b, small_calls=sc)
            # Linear and the token mean commute: mean_t(x_t W^T + b) = (mean_t x_t) W^T + b, so the 512 -> 3|6 head
            # runs on N rows instead of N*400 (refine_network.py:90-91); the result is held in fp16 by the reference
            run = lambda name, hslot: self.heads[name][1](self.heads[name][0].pooled(tok16, x16, self.enc.pe, hslot, sc), round_f16=True)
            side = self._head_side_stream(tok16)
            if side is None:
                n_seq, L = int(tok16.shape[0]), int(tok16.shape[1])
                if MERGED_HEAD_QKV and self.qkv2_p is not None and ROWS_QKV and n_seq * L >= ROWS_QKV_MIN_ROWS and \
                        not (sc and small_call(n_seq, SPLITK_MAX_HYPS)):
                    ctx = ops.attention_f16(ops.linear512(x16, self.qkv2_p, self.qkv2_b), 8)         # (N, L, 1024) = [trans | rot]
                    for i, name in enumerate(("trans", "rot")):
                        layer, lin = self.heads[name]
                        out[name] = lin(layer.pooled_from_context(ctx[..., 512 * i:512 * (i + 1)], tok16, self.enc.pe), round_f16=True)
                    return out
                for name in self.heads:
                    out[name] = run(name, slot)
                return out
            # round 5: a call of a few hypotheses (the reference's track_one: ONE) is a chain of ~100 latency-bound launches on an
            # empty chip, and the two heads are independent given the tokens (refine_network.py:90-91): the rotation head runs on the
            # process's side stream beside the translation head -- the same kernels on the same data, so the same bits; fork / join are
            # stream waits, i.e. parallel branches under hipGraph capture (as for the sub-batches of overlap.py)
            cur = torch.cuda.current_stream(tok16.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                out["rot"] = run("rot", (slot, "side"))
            out["trans"] = run("trans", slot)
            cur.wait_stream(side)
            for t in (tok16, x16):              # allocated on the main stream, read on the side stream
                t.record_stream(side)
            out["rot"].record_stream(cur)       # allocated on the side stream, read on the main stream after the join
            return out
        with _conv_backend():
            tok = self.enc(AB)
        for name, (layer, w, b) in self.heads.items():
            out[name] = F.linear(layer(tok), w, b).float().mean(dim=1)
        return out


class ScorePlan:
    small_calls = True          # see RefinePlan

    def __init__(self, model, device, precision="fp16", channels_last=True):
        _check_precision(precision)
        self.dtype = torch.float16 if precision == "fp16" else torch.float32
        self.hip = self.dtype == torch.float16
        self.module = model if precision == "torch_amp" else None
        if self.module is not None:
            return
        sd = _dev_sd(model, device)
        if self.hip:
            self.enc = _HipEncoder(sd, "encoderA", "encoderAB", device)
            self.att = _HipMHA(sd, "att", fp16_scores=True)
            self.att_cross = _HipMHA(sd, "att_cross", fp16_scores=True, small_calls=False)
            self.lin = _HipRowsLinear(sd["linear.weight"], sd["linear.bias"])
        else:
            self.enc = _Encoder(sd, "encoderA", "encoderAB", self.dtype, channels_last)
            self.att = _TorchMHA(sd, "att", self.dtype)
            self.att_cross = _TorchMHA(sd, "att_cross", self.dtype)
            self.lin_w, self.lin_b = sd["linear.weight"].to(self.dtype), sd["linear.bias"].to(self.dtype)

    @torch.inference_mode()
    def features(self, AB, slot=0, out=None):
        """(2n,6,H,W) -> pooled per-hypothesis features (n,512), fp16 on the HIP plan (score_network.py:60-74); written
        into `out` if given (HIP plan)"""
        if self.module is not None:
            n = AB.shape[0] // 2
            with _conv_backend(), torch.autocast("cuda", dtype=torch.float16):          # predict_score.py:193-194 -> score_network.py:60-74
                f = self.module.extract_feat(AB[:n], AB[n:])
            if out is not None:
                out.copy_(f)
                return out
            return f
        if self.hip:
            _, x16 = self.enc(AB, slot, small_calls=self.small_calls)
            # out_proj and the token mean commute (score_network.py:73-74): pool the attention output, project N rows
            return self.att.out_rows(ops.colmean_f16(self.att.context(x16, slot, self.small_calls)), out_f16=True, out=out)
        with _conv_backend():
            tok = self.enc(AB)
        f = self.att(tok).float().mean(dim=1)
        if out is not None:
            out.copy_(f)
            return out
        return f

    @torch.inference_mode()
    def head(self, feats, L):
        """cross-hypothesis attention + linear (score_network.py:83-88): feats (bs*L,512) -> logits (bs,L) fp32"""
        x = feats.reshape(-1, L, feats.shape[-1])
        if self.module is not None:
            with torch.autocast("cuda", dtype=torch.float16):          # score_network.py:83-88
                x, _ = self.module.att_cross(x, x, x)
                return self.module.linear(x).reshape(-1, L).float()
        if self.hip:
            if x.dtype != torch.float16:           # features that came back from an all-gather in another dtype
                x = x.to(torch.float16)
            o16 = self.att_cross(x.contiguous())
            return self.lin(o16.reshape(-1, o16.shape[-1]), round_f16=True).reshape(-1, L)
        x = self.att_cross(x.to(self.dtype))
        return F.linear(x, self.lin_w, self.lin_b).float().reshape(-1, L)

    def __call__(self, AB, L):
        return {"score_logit": self.head(self.features(AB), L)}
