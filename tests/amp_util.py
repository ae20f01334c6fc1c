"""Helpers of the fp16-policy parity tests: torch-CPU emulation of one op of the autocast sequence with explicit
roundings (same definitions as oracle/nets_amp.py), and the comparison "equal up to fp32-summation-order flips":
two fp32 accumulations of the same fp16 products differ in the last bits, so a value that lies within that distance
of an fp16 rounding boundary comes out one fp16 ulp apart (measured: ~0.3 % of the outputs of a 3x3 convolution,
tests/test_oracle_amp_golden.py).  Anything beyond one ulp of the LARGEST intermediate of the op, or more than a few
per cent of flipped elements, is a different arithmetic."""
import numpy as np
import torch
import torch.nn.functional as F


def r16(x):
    return x.to(torch.float16).to(torch.float32)


def ulp16(x):
    """fp16 ulp at the magnitude of x (np array or tensor -> np array)"""
    ax = np.maximum(np.abs(np.asarray(x, dtype=np.float64)), 2.0 ** -14)
    return 2.0 ** (np.floor(np.log2(ax)) - 10)


def flip_report(out, ref, mag=None, slack=None):
    """-> dict(frac mismatched, max error in ulps of `mag` (default: max(|out|,|ref|)), max abs).  `slack`: elementwise
    absolute error that fp32 accumulation itself may carry into the value BEFORE it is rounded (~1e-6 x sum |a_k b_k| for
    a dot product): a result that cancels to a small number has a tiny fp16 ulp but the full accumulation error."""
    out = np.asarray(out, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    m = np.maximum(np.abs(out), np.abs(ref)) if mag is None else np.maximum(np.asarray(mag, dtype=np.float64), np.maximum(np.abs(out), np.abs(ref)))
    d = np.abs(out - ref)
    dd = d if slack is None else np.maximum(d - np.asarray(slack, dtype=np.float64), 0.0)
    return dict(frac=float(np.mean(d > 0)), max_ulps=float((dd / ulp16(m)).max()), max_abs=float(d.max()),
                rel_rms=float(np.sqrt(np.mean(d ** 2)) / max(np.sqrt(np.mean(ref ** 2)), 1e-30)))


def assert_equal_up_to_flips(out, ref, mag=None, max_frac=0.03, max_ulps=1.0, what="", slack=None):
    rep = flip_report(out, ref, mag, slack)
    assert np.isfinite(np.asarray(out, dtype=np.float64)).all(), what
    assert rep["max_ulps"] <= max_ulps + 1e-6 and rep["frac"] <= max_frac, (what, rep)
    return rep


def conv_amp_ref(x16, w16, bias, bn, stride, residual=None, relu=True):
    """autocast op sequence of conv (+bias) (+eval BN as scale/shift) (+identity) (+ReLU) on fp16-valued fp32 tensors.
    bias: f32 (already fp16-representable) | None; bn: (scale, shift) | None.  -> (result, magnitude of the largest
    intermediate per element, fp32-accumulation slack per element).  The convolution itself is evaluated in float64
    (products of fp16 values are exact, and the CPU backend's float32 path may pick a Winograd algorithm for 3x3
    kernels whose error is far above an fp32 dot product's)."""
    pad = (w16.shape[-1] - 1) // 2
    acc = F.conv2d(x16.double(), w16.double(), None, stride=stride, padding=pad)
    slack = (1e-6 * F.conv2d(x16.double().abs(), w16.double().abs(), None, stride=stride, padding=pad)).float()
    y = r16(acc.float())
    mag = y.abs()
    if bias is not None:
        y = r16(y + bias[None, :, None, None])
        mag = torch.maximum(mag, y.abs())
    if bn is not None:
        y = r16(y * bn[0][None, :, None, None] + bn[1][None, :, None, None])
        mag = torch.maximum(mag, y.abs())
        slack = slack * bn[0].abs()[None, :, None, None]
    if residual is not None:
        mag = torch.maximum(mag, residual.abs())
        y = r16(y + residual)
        mag = torch.maximum(mag, y.abs())
    if relu:
        y = F.relu(y)
    return y, mag, slack


def geodesic(Ra, Rb):
    """rotation angle of Ra Rb^T via atan2(sin, cos): well conditioned near 0, unlike arccos of a float32 trace"""
    D = Ra.astype(np.float64) @ Rb.astype(np.float64).transpose(0, 2, 1)
    s = 0.5 * np.linalg.norm(np.stack([D[:, 2, 1] - D[:, 1, 2], D[:, 0, 2] - D[:, 2, 0], D[:, 1, 0] - D[:, 0, 1]], 1), axis=1)
    c = (np.trace(D, axis1=1, axis2=2) - 1) / 2
    return np.arctan2(s, c)


def kendall_tau(a, b):
    """Kendall rank correlation of two score vectors (O(n^2), n = 252)"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    sa = np.sign(a[:, None] - a[None, :])
    sb = np.sign(b[:, None] - b[None, :])
    n = len(a)
    return float((sa * sb).sum() / (n * (n - 1)))
