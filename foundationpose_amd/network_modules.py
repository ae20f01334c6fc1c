"""Building blocks of the two networks (reference: learning/models/network_modules.py:37-50,73-111,115-137).
Attribute names are kept (``net``, ``conv1``, ``bn1``, ``conv2``, ``bn2``, ``pe``) so that the reference's
checkpoints load with strict=True."""
import torch
import torch.nn as nn

from .weights import positional_table


class ConvBNReLU(nn.Module):
    """conv(k, stride, pad=(k-1)//2, bias) [+ norm] + ReLU, stored as ``self.net`` = Sequential."""

    def __init__(self, C_in, C_out, kernel_size=3, stride=1, norm_layer=nn.BatchNorm2d, bias=True):
        super().__init__()
        mods = [nn.Conv2d(C_in, C_out, kernel_size, stride, (kernel_size - 1) // 2, bias=bias)]
        if norm_layer is not None:
            mods.append(norm_layer(C_out))
        mods.append(nn.ReLU(inplace=True))
        self.net = nn.Sequential(*mods)

    def forward(self, x):
        return self.net(x)


class ResnetBasicBlock(nn.Module):
    """Two 3x3 convs with identity skip; norm layers only when ``norm_layer`` is given (use_BN)."""

    def __init__(self, inplanes, planes, norm_layer=nn.BatchNorm2d, bias=False):
        super().__init__()
        self.norm_layer = norm_layer
        self.conv1 = nn.Conv2d(inplanes, planes, 3, 1, 1, bias=bias)
        if norm_layer is not None:
            self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=bias)
        if norm_layer is not None:
            self.bn2 = norm_layer(planes)

    def forward(self, x):
        y = self.conv1(x)
        if self.norm_layer is not None:
            y = self.bn1(y)
        y = self.conv2(self.relu(y))
        if self.norm_layer is not None:
            y = self.bn2(y)
        return self.relu(y + x)


class PositionalEmbedding(nn.Module):
    def __init__(self, d_model, max_len=512):
        super().__init__()
        self.register_buffer("pe", positional_table(max_len, d_model))

    def forward(self, x):
        return x + self.pe[:, : x.size(1)]


def encoder_stem(c_in, norm_layer):
    """per-image encoder shared by A and B (refine_network.py:37-42 / score_network.py:36-41)."""
    return nn.Sequential(
        ConvBNReLU(c_in, 64, kernel_size=7, stride=2, norm_layer=norm_layer),
        ConvBNReLU(64, 128, kernel_size=3, stride=2, norm_layer=norm_layer),
        ResnetBasicBlock(128, 128, bias=True, norm_layer=norm_layer),
        ResnetBasicBlock(128, 128, bias=True, norm_layer=norm_layer),
    )


def encoder_joint(norm_layer):
    """joint encoder on channel-concatenated features (refine_network.py:44-50 / score_network.py:43-49)."""
    return nn.Sequential(
        ResnetBasicBlock(256, 256, bias=True, norm_layer=norm_layer),
        ResnetBasicBlock(256, 256, bias=True, norm_layer=norm_layer),
        ConvBNReLU(256, 512, kernel_size=3, stride=2, norm_layer=norm_layer),
        ResnetBasicBlock(512, 512, bias=True, norm_layer=norm_layer),
        ResnetBasicBlock(512, 512, bias=True, norm_layer=norm_layer),
    )


def cfg_get(cfg, key, default=None):
    try:
        return cfg[key]
    except (KeyError, TypeError):
        return getattr(cfg, key, default)
