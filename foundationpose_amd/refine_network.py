"""RefineNet (reference: learning/models/refine_network.py:26-93): shared stem on cat(A,B) along batch,
joint encoder on the channel concat, 400 tokens + sinusoidal PE, two post-norm TransformerEncoderLayer heads
(translation / rotation) each followed by Linear and a mean over tokens.  State-dict keys match the reference."""
import torch
import torch.nn as nn

from .network_modules import PositionalEmbedding, cfg_get, encoder_joint, encoder_stem


class RefineNet(nn.Module):
    def __init__(self, cfg=None, c_in=4, n_view=1):
        super().__init__()
        self.cfg = cfg
        norm = nn.BatchNorm2d if cfg_get(cfg, "use_BN", False) else None
        self.encodeA = encoder_stem(c_in, norm)
        self.encodeAB = encoder_joint(norm)
        self.pos_embed = PositionalEmbedding(d_model=512, max_len=400)
        rot_rep = cfg_get(cfg, "rot_rep", "axis_angle")
        if rot_rep == "axis_angle":
            rot_dim = 3
        elif rot_rep == "6d":
            rot_dim = 6
        else:
            raise RuntimeError(f"unknown rot_rep {rot_rep}")
        self.trans_head = nn.Sequential(
            nn.TransformerEncoderLayer(d_model=512, nhead=4, dim_feedforward=512, batch_first=True), nn.Linear(512, 3))
        self.rot_head = nn.Sequential(
            nn.TransformerEncoderLayer(d_model=512, nhead=4, dim_feedforward=512, batch_first=True), nn.Linear(512, rot_dim))

    def tokens(self, A, B):
        n = A.shape[0]
        feat = self.encodeA(torch.cat([A, B], dim=0))
        ab = self.encodeAB(torch.cat((feat[:n], feat[n:]), dim=1).contiguous())
        return self.pos_embed(ab.reshape(n, ab.shape[1], -1).permute(0, 2, 1))

    def forward(self, A, B):
        tok = self.tokens(A, B)
        return {"trans": self.trans_head(tok).mean(dim=1), "rot": self.rot_head(tok).mean(dim=1)}
