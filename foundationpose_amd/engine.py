"""Inference plans for RefineNet / ScoreNetMultiPair.

A plan is built once from a module's state_dict (reference checkpoints load unchanged): eval-mode BatchNorm is folded
into the preceding conv and weights are cast to the compute dtype.

precision='fp16' (deployment; mirrors the reference's autocast(fp16), predict_pose_refine.py:190, predict_score.py:193)
runs the whole network on libfp_amd.so:
  * patch-embed conv (7x7 s2, 6->64 + BN + ReLU)            -> fp_conv7x7s2_bn_relu_fwd
  * the 15 3x3 convs and every 512-wide projection          -> fp_igemm_f16_fwd (bias / residual / ReLU fused)
  * self-attention between in_proj and out_proj             -> fp_attention_f16_fwd
  * LayerNorm, LayerNorm + token mean                       -> fp_layernorm_f16_fwd, fp_colmean_f16_fwd
What is left to PyTorch are the tiny N-row tensors after the token mean (head linears on N x 512, the scorer's final
Linear) and elementwise glue (2.6 % of the GPU time, profiles/README.md).
precision='fp32' is the parity configuration: all torch ops, fp32, no autocast.
"""
import os

import torch
import torch.nn.functional as F

from . import ops


def _fold_bn(sd, conv_p, bn_p):
    w = sd[conv_p + ".weight"].float()
    b = sd.get(conv_p + ".bias")
    b = torch.zeros(w.shape[0], device=w.device) if b is None else b.float()
    if bn_p is not None and (bn_p + ".weight") in sd:
        scale = sd[bn_p + ".weight"].float() / torch.sqrt(sd[bn_p + ".running_var"].float() + 1e-5)
        shift = (b - sd[bn_p + ".running_mean"].float()) * scale + sd[bn_p + ".bias"].float()
        return w, scale, shift
    return w, torch.ones_like(b), b


class _Conv:
    """conv (+folded BN) (+ReLU); weights pre-scaled so the epilogue is bias-only."""

    def __init__(self, sd, conv_p, bn_p, stride, dtype, channels_last):
        w, scale, shift = _fold_bn(sd, conv_p, bn_p)
        self.raw_w, self.scale, self.shift = w, scale.contiguous(), shift.contiguous()
        wf = (w * scale[:, None, None, None]).to(dtype)
        self.w = wf.contiguous(memory_format=torch.channels_last) if channels_last else wf.contiguous()
        self.b = shift.to(dtype)
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
    """stem on cat(A,B) along batch + joint encoder on the channel concat -> tokens (N,400,512) + PE."""

    def __init__(self, sd, stem, joint, dtype, channels_last, use_hip):
        cl = channels_last
        self.dtype, self.cl, self.use_hip = dtype, cl, use_hip and dtype == torch.float16
        self.c1 = _Conv(sd, stem + ".0.net.0", stem + ".0.net.1", 2, dtype, cl)
        if self.use_hip:
            self.c1_wflat = self.c1.raw_w.to(torch.float16).reshape(64, -1).contiguous()
        self.c2 = _Conv(sd, stem + ".1.net.0", stem + ".1.net.1", 2, dtype, cl)
        self.s2, self.s3 = _Block(sd, stem + ".2", dtype, cl), _Block(sd, stem + ".3", dtype, cl)
        self.j0, self.j1 = _Block(sd, joint + ".0", dtype, cl), _Block(sd, joint + ".1", dtype, cl)
        self.j2 = _Conv(sd, joint + ".2.net.0", joint + ".2.net.1", 2, dtype, cl)
        self.j3, self.j4 = _Block(sd, joint + ".3", dtype, cl), _Block(sd, joint + ".4", dtype, cl)
        self.pe = sd["pos_embed.pe"].to(dtype)

    def __call__(self, AB):
        """AB: (2N,6,H,W) -- A in the first half, B in the second."""
        n = AB.shape[0] // 2
        if self.use_hip:
            x = ops.conv7x7s2_bn_relu(AB, self.c1_wflat, self.c1.scale, self.c1.shift, channels_last=self.cl)
        else:
            if self.cl:
                AB = AB.contiguous(memory_format=torch.channels_last)
            x = self.c1(AB)
        x = self.s3(self.s2(self.c2(x)))
        ab = torch.cat((x[:n], x[n:]), dim=1)
        ab = self.j4(self.j3(self.j2(self.j1(self.j0(ab)))))
        tok = ab.permute(0, 2, 3, 1).reshape(n, -1, ab.shape[1])  # free view when channels_last
        return tok + self.pe[:, : tok.shape[1]]


def _igemm_weight(conv):
    """(Cout, Cin, 3, 3) BN-folded weights -> (Cout, 9*Cin) fp16 with k ordered (ky, kx, ci); bias f32"""
    w = conv.raw_w * conv.scale[:, None, None, None]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(torch.float16).contiguous(), conv.shift.float().contiguous()


class _HipEncoder:
    """The encoder on libfp_amd.so only: patch-embed conv (fp_conv7x7s2_bn_relu_fwd) + 15 implicit-GEMM 3x3 convs
    (fp_igemm_f16_fwd) with bias / residual / ReLU epilogues.  Activations are NHWC fp16 with a 1-pixel zero border
    (allocated once per batch size and reused), the A|B channel concat is a strided store of the stem's last conv
    (refine_network.py:82-85), and the last conv writes the (N, 400, 512) token matrix directly."""

    def __init__(self, sd, stem, joint, device):
        mk = lambda c, b, s: _Conv(sd, c, b, s, torch.float16, False)
        self.c1 = mk(stem + ".0.net.0", stem + ".0.net.1", 2)
        self.c1_wflat = self.c1.raw_w.to(torch.float16).reshape(64, -1).contiguous()
        names = [("c2", stem + ".1.net.0", stem + ".1.net.1")]
        for blk, pre in (("s2", stem + ".2"), ("s3", stem + ".3"), ("j0", joint + ".0"), ("j1", joint + ".1"),
                         ("j3", joint + ".3"), ("j4", joint + ".4")):
            names += [(blk + "a", pre + ".conv1", pre + ".bn1"), (blk + "b", pre + ".conv2", pre + ".bn2")]
        names.append(("j2", joint + ".2.net.0", joint + ".2.net.1"))
        self.w = {k: _igemm_weight(mk(c, b, 1)) for k, c, b in names}
        self.pe = sd["pos_embed.pe"].to(torch.float16)
        self.device = device
        self._bufs = {}

    def _buffers(self, n, H, W):
        key = (n, H, W)
        b = self._bufs.get(key)
        if b is None:
            z = lambda *shape: torch.zeros(shape, dtype=torch.float16, device=self.device)
            h1, w1, h2, w2, h3, w3 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
            b = dict(P1=z(2 * n, h1 + 2, w1 + 2, 64), P2=z(2 * n, h2 + 2, w2 + 2, 128), P3=z(2 * n, h2 + 2, w2 + 2, 128),
                     T=z(2 * n, h2 + 2, w2 + 2, 128), CAT=z(n, h2 + 2, w2 + 2, 256), J0=z(n, h2 + 2, w2 + 2, 256),
                     T2=z(n, h2 + 2, w2 + 2, 256), Q0=z(n, h3 + 2, w3 + 2, 512), Q1=z(n, h3 + 2, w3 + 2, 512),
                     T3=z(n, h3 + 2, w3 + 2, 512), dims=(h1, w1, h2, w2, h3, w3))
            self._bufs = {key: b}   # one batch size resident at a time
        return b

    def _conv(self, name, x, Bn, Ho, Wo, Cin, Cout, y, stride=1, res=None, relu=True, gout=None, gres=None):
        w, bias = self.w[name]
        G = ops.IgemmGeom.image
        gin = G(Ho, Wo, 1, Cin, stride=stride, offset=0)
        gout = gout if gout is not None else G(Ho, Wo, 1, Cout)
        if res is not None and gres is None:
            gres = G(Ho, Wo, 1, Cout)
        return ops.igemm_f16(x, gin, w, bias, y, gout, Bn * Ho * Wo, Cout, Cin, 9, relu=relu, residual=res, r_geom=gres)

    def __call__(self, AB):
        n2, _, H, W = AB.shape
        n = n2 // 2
        b = self._buffers(n, H, W)
        h1, w1, h2, w2, h3, w3 = b["dims"]
        G = ops.IgemmGeom.image
        ops.conv7x7s2_bn_relu(AB, self.c1_wflat, self.c1.scale, self.c1.shift, out_padded=b["P1"])
        self._conv("c2", b["P1"], n2, h2, w2, 64, 128, b["P2"], stride=2)
        self._conv("s2a", b["P2"], n2, h2, w2, 128, 128, b["T"])
        self._conv("s2b", b["T"], n2, h2, w2, 128, 128, b["P3"], res=b["P2"])
        self._conv("s3a", b["P3"], n2, h2, w2, 128, 128, b["T"])
        # stem output of image i (A) and image n+i (B) side by side along C: torch.cat((a, b), 1)
        self._conv("s3b", b["T"], n2, h2, w2, 128, 128, b["CAT"], res=b["P3"],
                   gout=G(h2, w2, 1, 256, bsplit=n, cgroup=128), gres=G(h2, w2, 1, 128))
        self._conv("j0a", b["CAT"], n, h2, w2, 256, 256, b["T2"])
        self._conv("j0b", b["T2"], n, h2, w2, 256, 256, b["J0"], res=b["CAT"])
        self._conv("j1a", b["J0"], n, h2, w2, 256, 256, b["T2"])
        self._conv("j1b", b["T2"], n, h2, w2, 256, 256, b["CAT"], res=b["J0"])
        self._conv("j2", b["CAT"], n, h3, w3, 256, 512, b["Q0"], stride=2)
        self._conv("j3a", b["Q0"], n, h3, w3, 512, 512, b["T3"])
        self._conv("j3b", b["T3"], n, h3, w3, 512, 512, b["Q1"], res=b["Q0"])
        self._conv("j4a", b["Q1"], n, h3, w3, 512, 512, b["T3"])
        tok = torch.empty((n, h3 * w3, 512), dtype=torch.float16, device=AB.device)
        self._conv("j4b", b["T3"], n, h3, w3, 512, 512, tok, res=b["Q1"], gout=G(h3, w3, 0, 512), gres=G(h3, w3, 1, 512))
        return tok + self.pe[:, : tok.shape[1]]


class _Linear:
    def __init__(self, w, b, dtype, use_hip):
        self.use_hip = use_hip and dtype == torch.float16 and w.shape[0] % 128 == 0 and w.shape[1] % 64 == 0
        self.w = w.to(dtype).contiguous()
        self.b32 = b.float().contiguous()
        self.b = b.to(dtype)

    def __call__(self, x, relu=False, residual=None):
        """y = act(x @ w.T + b (+ residual)); the fp16 plan runs it on fp_igemm_f16_fwd with the epilogue fused"""
        if self.use_hip:
            x2 = x.reshape(-1, x.shape[-1])
            x2 = x2 if x2.is_contiguous() else x2.contiguous()
            M, K = x2.shape
            No = self.w.shape[0]
            y = torch.empty((M, No), dtype=torch.float16, device=x.device)
            r2 = None
            if residual is not None:
                r2 = residual.reshape(-1, No)
                r2 = r2 if r2.is_contiguous() else r2.contiguous()
            Gm = ops.IgemmGeom.matrix
            ops.igemm_f16(x2, Gm(K), self.w, self.b32, y, Gm(No), M, No, K, 1, relu=relu, residual=r2,
                          r_geom=Gm(No) if r2 is not None else None)
            return y.reshape(*x.shape[:-1], No)
        y = F.linear(x, self.w, self.b)
        if residual is not None:
            y = y + residual
        return F.relu_(y) if relu else y


class _MHA:
    """self-attention of nn.MultiheadAttention(512, 4, batch_first=True) called as att(x, x, x)."""

    def __init__(self, sd, p, dtype, use_hip, nhead=4):
        self.qkv = _Linear(sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"], dtype, use_hip)
        self.out = _Linear(sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"], dtype, use_hip)
        self.nhead = nhead
        self.hip = use_hip and dtype == torch.float16 and os.environ.get("FP_ATTENTION", "hip") == "hip"

    def context(self, x):
        """softmax(q k^T / sqrt(d)) v with heads merged, before the output projection: (Bn, L, D)"""
        Bn, L, D = x.shape
        hd = D // self.nhead
        if self.hip and hd == 128:
            # the in_proj output goes to the MFMA attention kernel as it stands ([q | k | v] rows), heads merged on the way out
            return ops.attention_f16(self.qkv(x).reshape(Bn, L, 3 * D), self.nhead)
        qkv = self.qkv(x).reshape(Bn, L, 3, self.nhead, hd)
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        # fused attention: the (Bn*4, L, L) probability tensor (161 M elements at N=252, which the reference
        # materialises because it calls nn.MultiheadAttention with need_weights=True) never reaches HBM
        return F.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(Bn, L, D)

    def __call__(self, x, residual=None):
        return self.out(self.context(x), residual=residual)


class _EncoderLayer:
    """nn.TransformerEncoderLayer(512, 4, 512, batch_first=True) in eval mode: post-norm, ReLU, LN eps 1e-5 (fp32)."""

    def __init__(self, sd, p, dtype, use_hip):
        self.att = _MHA(sd, p + ".self_attn", dtype, use_hip)
        self.l1 = _Linear(sd[p + ".linear1.weight"], sd[p + ".linear1.bias"], dtype, use_hip)
        self.l2 = _Linear(sd[p + ".linear2.weight"], sd[p + ".linear2.bias"], dtype, use_hip)
        self.n1 = (sd[p + ".norm1.weight"].float(), sd[p + ".norm1.bias"].float())
        self.n2 = (sd[p + ".norm2.weight"].float(), sd[p + ".norm2.bias"].float())
        self.dtype = dtype
        self.hip = use_hip and dtype == torch.float16

    def _ln(self, x, wb):
        if self.hip:
            return ops.layernorm_f16(x, wb[0], wb[1], 1e-5)
        return F.layer_norm(x.float(), (x.shape[-1],), wb[0], wb[1], 1e-5).to(self.dtype)

    def __call__(self, x):
        x = self._ln(self.att(x, residual=x), self.n1)
        return self._ln(self.l2(self.l1(x, relu=True), residual=x), self.n2)

    def pooled(self, x):
        """mean over the tokens of the layer output, (N, L, 512) -> (N, 512) fp32.  On the HIP plan the second
        LayerNorm is fused with the token mean (fp_colmean_f16_fwd): the normalised tensor is never written."""
        x = self._ln(self.att(x, residual=x), self.n1)
        z = self.l2(self.l1(x, relu=True), residual=x)
        if self.hip:
            return ops.colmean_f16(z, self.n2[0], self.n2[1], 1e-5)
        return self._ln(z, self.n2).float().mean(dim=1)


def _dev_sd(module_or_sd, device):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    return {k: v.detach().to(device) for k, v in sd.items()}


class RefinePlan:
    def __init__(self, model, device, precision="fp16", channels_last=True, use_hip=True):
        sd = _dev_sd(model, device)
        self.dtype = torch.float16 if precision == "fp16" else torch.float32
        hip = use_hip and self.dtype == torch.float16
        self.enc = _HipEncoder(sd, "encodeA", "encodeAB", device) if hip else \
            _Encoder(sd, "encodeA", "encodeAB", self.dtype, channels_last, use_hip)
        self.heads = {}
        for name in ("trans", "rot"):
            self.heads[name] = (_EncoderLayer(sd, f"{name}_head.0", self.dtype, use_hip),
                                sd[f"{name}_head.1.weight"].to(self.dtype), sd[f"{name}_head.1.bias"].to(self.dtype))
        self.hip = hip

    @torch.inference_mode()
    def __call__(self, AB):
        tok = self.enc(AB)
        out = {}
        for name, (layer, w, b) in self.heads.items():
            if self.hip:
                # Linear and the token mean commute: mean_t(x_t W^T + b) = (mean_t x_t) W^T + b, so the 512 -> 3|6 head
                # runs on N rows instead of N*400 (refine_network.py:90-91)
                out[name] = F.linear(layer.pooled(tok), w.float(), b.float())
            else:
                out[name] = F.linear(layer(tok), w, b).float().mean(dim=1)
        return out


class ScorePlan:
    def __init__(self, model, device, precision="fp16", channels_last=True, use_hip=True):
        sd = _dev_sd(model, device)
        self.dtype = torch.float16 if precision == "fp16" else torch.float32
        hip = use_hip and self.dtype == torch.float16
        self.enc = _HipEncoder(sd, "encoderA", "encoderAB", device) if hip else \
            _Encoder(sd, "encoderA", "encoderAB", self.dtype, channels_last, use_hip)
        self.att = _MHA(sd, "att", self.dtype, use_hip)
        self.att_cross = _MHA(sd, "att_cross", self.dtype, use_hip)
        self.lin_w, self.lin_b = sd["linear.weight"].to(self.dtype), sd["linear.bias"].to(self.dtype)
        self.hip = hip

    @torch.inference_mode()
    def features(self, AB):
        """(2n,6,H,W) -> pooled per-hypothesis features (n,512) (score_network.py:60-74)."""
        tok = self.enc(AB)
        if self.hip:
            # out_proj and the token mean commute (score_network.py:73-74): pool the attention output, project N rows
            pooled = ops.colmean_f16(self.att.context(tok))
            return F.linear(pooled, self.att.out.w.float(), self.att.out.b32).to(self.dtype)
        return self.att(tok).float().mean(dim=1).to(self.dtype)

    @torch.inference_mode()
    def head(self, feats, L):
        """cross-hypothesis attention + linear (score_network.py:83-88): feats (bs*L,512) -> logits (bs,L) fp32."""
        x = feats.reshape(-1, L, feats.shape[-1])
        x = self.att_cross(x)
        return F.linear(x, self.lin_w, self.lin_b).float().reshape(-1, L)

    def __call__(self, AB, L):
        return {"score_logit": self.head(self.features(AB), L)}
