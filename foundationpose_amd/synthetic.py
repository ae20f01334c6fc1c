"""Seeded synthetic RGB-D scene of SURVEY.md 8(d) (YCB-V intrinsics, 'can' at 0.75 m, noise + dropout).
The frame is rendered by a caller-supplied ``render_fn`` (the product's HIP rasteriser in bench.py, the CPU oracle in
the CPU tests), so this module has no compute dependency of its own."""
import numpy as np

YCBV_K = np.array([[1066.778, 0.0, 312.9869], [0.0, 1067.487, 241.3109], [0.0, 0.0, 1.0]])
H, W = 480, 640


def quat_to_rot(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def gt_pose(seed=0):
    rng = np.random.default_rng(seed)
    T = np.eye(4)
    T[:3, :3] = quat_to_rot(rng.normal(size=4))
    T[:3, 3] = [0.02, -0.03, 0.75]
    return T


def compose_frame(color, depth, seed=0, noise_std=0.001, dropout=0.02, bg_depth=1.2):
    """color (H,W,3) in [0,1], depth (H,W) metres (0 = empty) of the object rendered at the GT pose ->
    rgb uint8 (H,W,3), depth f32 (H,W) with background plane, gaussian noise and dropout, mask (H,W) bool."""
    rng = np.random.default_rng(seed + 1000)
    color = np.asarray(color, dtype=np.float64)
    depth = np.asarray(depth, dtype=np.float64)
    mask = depth > 0
    bg = rng.uniform(0.3, 0.6, size=(H // 8, W // 8, 3))
    bg = np.kron(bg, np.ones((8, 8, 1)))
    rgb = np.where(mask[..., None], color, bg)
    d = np.where(mask, depth, bg_depth)
    d = d + rng.normal(0, noise_std, size=d.shape)
    d[rng.uniform(size=d.shape) < dropout] = 0.0
    return (np.clip(rgb, 0, 1) * 255).astype(np.uint8), d.astype(np.float32), mask


def perturbed_poses(pose, n, seed=0, max_trans=0.02, max_rot_deg=10.0):
    """n seeded perturbations of a pose (tracking config: <= 2 cm, <= 10 deg)."""
    rng = np.random.default_rng(seed + 2000)
    out = np.tile(np.asarray(pose, dtype=np.float64)[None], (n, 1, 1))
    for i in range(n):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        ang = np.deg2rad(rng.uniform(0, max_rot_deg))
        Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        dR = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
        out[i, :3, :3] = dR @ out[i, :3, :3]
        out[i, :3, 3] += rng.uniform(-1, 1, size=3) * max_trans / np.sqrt(3)
    return out
