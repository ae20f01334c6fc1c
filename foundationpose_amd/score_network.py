"""ScoreNetMultiPair (reference: learning/models/score_network.py:27-90): same encoder (own weights), self-attention
over the 400 tokens of each pair, mean-pool to a 512-vector per hypothesis, cross-attention over the L hypotheses of
one object, Linear(512,1) -> (B,L) logits.  State-dict keys match the reference (note ``encoderA`` vs ``encodeA``)."""
import torch
import torch.nn as nn

from .network_modules import PositionalEmbedding, cfg_get, encoder_joint, encoder_stem


class ScoreNetMultiPair(nn.Module):
    def __init__(self, cfg=None, c_in=4):
        super().__init__()
        self.cfg = cfg
        norm = nn.BatchNorm2d if cfg_get(cfg, "use_BN", False) else None
        self.encoderA = encoder_stem(c_in, norm)
        self.encoderAB = encoder_joint(norm)
        self.att = nn.MultiheadAttention(embed_dim=512, num_heads=4, bias=True, batch_first=True)
        self.att_cross = nn.MultiheadAttention(embed_dim=512, num_heads=4, bias=True, batch_first=True)
        self.pos_embed = PositionalEmbedding(d_model=512, max_len=400)
        self.linear = nn.Linear(512, 1)

    def extract_feat(self, A, B):
        n = A.shape[0]
        feat = self.encoderA(torch.cat([A, B], dim=0))
        ab = self.encoderAB(torch.cat((feat[:n], feat[n:]), dim=1))
        tok = self.pos_embed(ab.reshape(n, ab.shape[1], -1).permute(0, 2, 1))
        tok, _ = self.att(tok, tok, tok)
        return tok.mean(dim=1).reshape(n, -1)

    def forward(self, A, B, L):
        bs = A.shape[0] // L
        x = self.extract_feat(A, B).reshape(bs, L, -1)
        x, _ = self.att_cross(x, x, x)
        return {"score_logit": self.linear(x).reshape(bs, L)}
