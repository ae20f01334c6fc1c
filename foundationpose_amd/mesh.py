"""Minimal triangle-mesh container (duck-types the few trimesh attributes the reference touches:
``vertices``, ``faces``, ``vertex_normals``, ``visual.uv``, ``visual.material.image``, ``visual.vertex_colors``,
``copy()`` -- estimater.py:44-71, Utils.py:104-130) plus the synthetic "can" of BASELINE.md section 3.
trimesh / open3d are not available in the target image, so nothing here depends on them."""
import copy as _copy

import numpy as np


class _Material:
    def __init__(self, image):
        self.image = image  # (Ht,Wt,3) uint8 ndarray (or a PIL image with .convert)


class TextureVisual:
    kind = "texture"

    def __init__(self, uv, image):
        self.uv = np.asarray(uv, dtype=np.float64)
        self.material = _Material(image)
        self.vertex_colors = None


class ColorVisual:
    kind = "vertex"

    def __init__(self, vertex_colors=None):
        self.vertex_colors = None if vertex_colors is None else np.asarray(vertex_colors)


def vertex_normals_from_faces(vertices, faces):
    """Area-weighted average of the incident face normals."""
    v = np.asarray(vertices, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    vn = np.zeros_like(v)
    for k in range(3):
        np.add.at(vn, f[:, k], fn)
    ln = np.linalg.norm(vn, axis=1, keepdims=True)
    return vn / np.maximum(ln, 1e-20)


class SimpleMesh:
    def __init__(self, vertices, faces, vertex_normals=None, uv=None, texture=None, vertex_colors=None):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64)
        self._vn = None if vertex_normals is None else np.asarray(vertex_normals, dtype=np.float64)
        if texture is not None and uv is not None:
            self.visual = TextureVisual(uv, np.asarray(texture))
        else:
            self.visual = ColorVisual(vertex_colors)

    @property
    def vertex_normals(self):
        if self._vn is None:
            self._vn = vertex_normals_from_faces(self.vertices, self.faces)
        return self._vn

    def copy(self):
        return _copy.deepcopy(self)


def make_can_mesh(radius=0.051, height=0.140, n_ang=50, n_axial=48, textured=True, tex_size=512, seed=0):
    """Closed cylinder ~ YCB-V 002_master_chef_can: 4900 triangles, 2501 vertices."""
    ang = np.arange(n_ang + 1) / n_ang * 2 * np.pi  # duplicated seam column keeps the UV map continuous
    zs = np.linspace(-height / 2, height / 2, n_axial + 1)
    verts, uvs = [], []
    for iz, z in enumerate(zs):
        for ia, a in enumerate(ang):
            verts.append([radius * np.cos(a), radius * np.sin(a), z])
            uvs.append([ia / n_ang, iz / n_axial])
    ring = n_ang + 1
    faces = []
    for iz in range(n_axial):
        for ia in range(n_ang):
            a0 = iz * ring + ia
            a1 = a0 + 1
            b0 = a0 + ring
            b1 = b0 + 1
            faces.append([a0, a1, b1])
            faces.append([a0, b1, b0])
    cb = len(verts)
    verts.append([0, 0, -height / 2]); uvs.append([0.5, 0.0])
    ct = len(verts)
    verts.append([0, 0, height / 2]); uvs.append([0.5, 1.0])
    top0 = n_axial * ring
    for ia in range(n_ang):
        faces.append([cb, ia + 1, ia])
        faces.append([ct, top0 + ia, top0 + ia + 1])
    verts = np.asarray(verts); faces = np.asarray(faces)
    if textured:
        return SimpleMesh(verts, faces, uv=np.asarray(uvs), texture=make_texture(tex_size, seed))
    rng = np.random.default_rng(seed)
    cols = (rng.uniform(0.2, 1.0, size=(len(verts), 3)) * 255).astype(np.uint8)
    return SimpleMesh(verts, faces, vertex_colors=cols)


def make_texture(size=512, seed=0):
    """Low-pass noise plus a few high-contrast stripes so that the pose is observable."""
    rng = np.random.default_rng(seed)
    base = rng.uniform(0, 1, size=(size // 16, size // 16, 3))
    img = np.kron(base, np.ones((16, 16, 1)))
    k = np.ones(9) / 9.0
    for ax in (0, 1):
        img = np.apply_along_axis(lambda m: np.convolve(np.concatenate([m[-4:], m, m[:4]]), k, mode="valid"), ax, img)
    yy, xx = np.mgrid[0:size, 0:size]
    stripes = ((xx // 40) % 3 == 0) & ((yy // 64) % 2 == 0)
    img[stripes] = img[stripes] * 0.25
    img[(yy % 128) < 6] = [0.95, 0.1, 0.1]
    img[(xx % 170) < 5] = [0.1, 0.1, 0.9]
    return (np.clip(img, 0, 1) * 255).astype(np.uint8)
