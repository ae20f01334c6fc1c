"""Functional torch-CPU fp32 restatement of the two networks, driven directly by a state_dict.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates learning/models/refine_network.py:73-93, score_network.py:60-90, network_modules.py:37-50,94-111,
133-137 and the torch modules they instantiate (nn.TransformerEncoderLayer post-norm/ReLU/eps 1e-5,
nn.MultiheadAttention).  PINNED: tests/test_oracle_golden.py checks these functions against
outputs of the reference's own modules (tests/golden/nets_golden.npz)."""
import math

import torch
import torch.nn.functional as F


def _conv(x, sd, p, stride):
    w = sd[p + ".weight"]
    return F.conv2d(x, w, sd.get(p + ".bias"), stride=stride, padding=(w.shape[-1] - 1) // 2)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=1e-5)


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
    return F.relu(y + x)


def encoder_tokens(A, B, sd, stem, joint):
    n = A.shape[0]
    x = torch.cat([A, B], dim=0)
    x = _conv_bn_relu(x, sd, stem + ".0", 2)
    x = _conv_bn_relu(x, sd, stem + ".1", 2)
    x = _basic_block(x, sd, stem + ".2")
    x = _basic_block(x, sd, stem + ".3")
    ab = torch.cat((x[:n], x[n:]), dim=1)
    ab = _basic_block(ab, sd, joint + ".0")
    ab = _basic_block(ab, sd, joint + ".1")
    ab = _conv_bn_relu(ab, sd, joint + ".2", 2)
    ab = _basic_block(ab, sd, joint + ".3")
    ab = _basic_block(ab, sd, joint + ".4")
    tok = ab.reshape(n, ab.shape[1], -1).permute(0, 2, 1)
    return tok + sd["pos_embed.pe"][:, : tok.shape[1]]


def mha(x, sd, p, nhead=4):
    """self-attention of nn.MultiheadAttention(batch_first=True) called as att(x,x,x)."""
    Bn, L, D = x.shape
    qkv = F.linear(x, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"])
    q, k, v = qkv.split(D, dim=-1)
    hd = D // nhead

    def heads(t):
        return t.reshape(Bn, L, nhead, hd).permute(0, 2, 1, 3)

    q, k, v = heads(q), heads(k), heads(v)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = (att @ v).permute(0, 2, 1, 3).reshape(Bn, L, D)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def encoder_layer(x, sd, p):
    """nn.TransformerEncoderLayer(d_model=512, nhead=4, dim_feedforward=512, batch_first=True), eval, post-norm."""
    x = F.layer_norm(x + mha(x, sd, p + ".self_attn"), (x.shape[-1],), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    ff = F.linear(F.relu(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])), sd[p + ".linear2.weight"],
                  sd[p + ".linear2.bias"])
    return F.layer_norm(x + ff, (x.shape[-1],), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)


@torch.no_grad()
def refine_forward(A, B, sd):
    tok = encoder_tokens(A, B, sd, "encodeA", "encodeAB")
    out = {}
    for name in ("trans", "rot"):
        h = encoder_layer(tok, sd, f"{name}_head.0")
        out[name] = F.linear(h, sd[f"{name}_head.1.weight"], sd[f"{name}_head.1.bias"]).mean(dim=1)
    return out


@torch.no_grad()
def score_features(A, B, sd):
    tok = encoder_tokens(A, B, sd, "encoderA", "encoderAB")
    return mha(tok, sd, "att").mean(dim=1)


@torch.no_grad()
def score_forward(A, B, sd, L):
    feats = score_features(A, B, sd)
    bs = A.shape[0] // L
    x = feats.reshape(bs, L, -1)
    x = mha(x, sd, "att_cross")
    return {"score_logit": F.linear(x, sd["linear.weight"], sd["linear.bias"]).reshape(bs, L)}
