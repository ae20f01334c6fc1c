"""Mesh files: Wavefront OBJ (+ MTL + texture image) in and out, numpy + PIL only.

The reference loads meshes with trimesh (`run_demo.py:29`, `datareader.py:150`); the hot path touches `.vertices`,
`.faces`, `.vertex_normals`, `.visual.uv`, `.visual.material.image` / `.visual.vertex_colors` (Utils.py:104-130).
`load_obj` returns a `SimpleMesh` with exactly those attributes.  Like trimesh it merges the OBJ's separate
position / texture-coordinate index streams into one vertex list (a vertex per distinct (v, vt) pair), so that
`uv_idx == faces`."""
import os

import numpy as np

from .mesh import SimpleMesh


def _parse_mtl(path):
    tex = None
    kd = None
    if not os.path.exists(path):
        return tex, kd
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "map_Kd":
                tex = os.path.join(os.path.dirname(path), t[-1])
            elif t[0] == "Kd" and len(t) >= 4:
                kd = [float(v) for v in t[1:4]]
    return tex, kd


def load_obj(path):
    """-> SimpleMesh.  Polygons are fan-triangulated; negative (relative) indices are supported."""
    v, vt, vn, corners, mtl = [], [], [], [], None
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                v.append([float(x) for x in t[1:4]])
            elif t[0] == "vt":
                vt.append([float(x) for x in t[1:3]])
            elif t[0] == "vn":
                vn.append([float(x) for x in t[1:4]])
            elif t[0] == "mtllib":
                mtl = os.path.join(os.path.dirname(path), " ".join(t[1:]))
            elif t[0] == "f":
                poly = []
                for tok in t[1:]:
                    p = (tok.split("/") + ["", ""])[:3]
                    iv = int(p[0])
                    it = int(p[1]) if p[1] else 0
                    inn = int(p[2]) if p[2] else 0
                    iv = iv - 1 if iv > 0 else len(v) + iv
                    it = (it - 1 if it > 0 else len(vt) + it) if p[1] else -1
                    inn = (inn - 1 if inn > 0 else len(vn) + inn) if p[2] else -1
                    poly.append((iv, it, inn))
                for k in range(1, len(poly) - 1):
                    corners.append((poly[0], poly[k], poly[k + 1]))
    v = np.asarray(v, dtype=np.float64).reshape(-1, 3)
    vt = np.asarray(vt, dtype=np.float64).reshape(-1, 2)
    vn = np.asarray(vn, dtype=np.float64).reshape(-1, 3)
    corners = np.asarray(corners, dtype=np.int64).reshape(-1, 3, 3)
    has_uv = len(vt) > 0 and (corners[..., 1] >= 0).all()
    key = corners[..., :2].reshape(-1, 2) if has_uv else corners[..., :1].reshape(-1, 1)
    uniq, inv = np.unique(key, axis=0, return_inverse=True)
    faces = inv.reshape(-1, 3)
    vertices = v[uniq[:, 0]]
    uv = vt[uniq[:, 1]] if has_uv else None
    normals = None
    if len(vn) > 0 and (corners[..., 2] >= 0).all():
        acc = np.zeros_like(vertices)
        np.add.at(acc, faces.reshape(-1), vn[corners[..., 2].reshape(-1)])
        normals = acc / np.maximum(np.linalg.norm(acc, axis=1, keepdims=True), 1e-20)
    texture, kd = _parse_mtl(mtl) if mtl else (None, None)
    image = None
    if texture is not None and os.path.exists(texture) and uv is not None:
        from PIL import Image
        image = np.asarray(Image.open(texture).convert("RGB"))
    vcol = None
    if image is None and kd is not None:
        vcol = np.tile((np.clip(np.asarray(kd), 0, 1) * 255).astype(np.uint8)[None], (len(vertices), 1))
    return SimpleMesh(vertices, faces, vertex_normals=normals, uv=uv if image is not None else None, texture=image,
                      vertex_colors=vcol)


def save_obj(mesh, path):
    """writes <path>, <stem>.mtl and <stem>.png (when the mesh is textured)"""
    stem = os.path.splitext(path)[0]
    visual = mesh.visual
    image = getattr(getattr(visual, "material", None), "image", None)
    uv = getattr(visual, "uv", None)
    textured = image is not None and uv is not None
    with open(path, "w") as f:
        if textured:
            f.write(f"mtllib {os.path.basename(stem)}.mtl\nusemtl material_0\n")
        for p in np.asarray(mesh.vertices):
            f.write("v %.9g %.9g %.9g\n" % tuple(p))
        if textured:
            for t in np.asarray(uv):
                f.write("vt %.9g %.9g\n" % tuple(t))
        for n in np.asarray(mesh.vertex_normals):
            f.write("vn %.9g %.9g %.9g\n" % tuple(n))
        for a, b, c in np.asarray(mesh.faces) + 1:
            if textured:
                f.write(f"f {a}/{a}/{a} {b}/{b}/{b} {c}/{c}/{c}\n")
            else:
                f.write(f"f {a}//{a} {b}//{b} {c}//{c}\n")
    if textured:
        from PIL import Image
        Image.fromarray(np.asarray(image)[..., :3].astype(np.uint8)).save(stem + ".png")
        with open(stem + ".mtl", "w") as f:
            f.write(f"newmtl material_0\nKa 1 1 1\nKd 1 1 1\nKs 0 0 0\nmap_Kd {os.path.basename(stem)}.png\n")


def load_mesh(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".obj":
        return load_obj(path)
    raise NotImplementedError(f"mesh format '{ext}' is not supported (OBJ only)")
