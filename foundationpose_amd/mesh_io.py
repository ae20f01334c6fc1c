"""Mesh files: Wavefront OBJ (+ MTL + texture image) and PLY in and out, numpy + PIL only.

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


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def load_ply(path):
    """-> SimpleMesh.  PLY as the BOP model sets ship it: `ascii` or `binary_little_endian`, a vertex element with
    x y z (+ nx ny nz, + red green blue, + texture_u texture_v) and a face element with one index list (polygons are
    fan-triangulated); a `comment TextureFile <name>` header line names the texture image."""
    with open(path, "rb") as f:
        fmt, elements, texfile = None, [], None
        line = f.readline().strip()
        if line != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            t = line.decode("ascii", "replace").split()
            if not t:
                continue
            if t[0] == "format":
                fmt = t[1]
            elif t[0] == "comment" and len(t) >= 3 and t[1] == "TextureFile":
                texfile = os.path.join(os.path.dirname(path), t[2])
            elif t[0] == "element":
                elements.append((t[1], int(t[2]), []))
            elif t[0] == "property":
                elements[-1][2].append(t[1:])
            elif t[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian"):
            raise NotImplementedError(f"{path}: PLY format '{fmt}' is not supported")
        data = {}
        for name, count, props in elements:
            scalar = all(pr[0] != "list" for pr in props)
            if fmt == "ascii":
                rows = [f.readline().split() for _ in range(count)]
                if scalar:
                    arr = np.array(rows, dtype=np.float64).reshape(count, len(props))
                    data[name] = {pr[-1]: arr[:, i] for i, pr in enumerate(props)}
                else:
                    data[name] = {props[0][-1]: [np.array(r[1:1 + int(r[0])], dtype=np.int64) for r in rows]}
            elif scalar:
                dt = np.dtype([(pr[-1], "<" + _PLY_TYPES[pr[0]]) for pr in props])
                arr = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
                data[name] = {n: arr[n].astype(np.float64) for n in arr.dtype.names}
            else:
                if len(props) != 1:
                    raise NotImplementedError(f"{path}: element '{name}' mixes list and scalar properties")
                ct, it = np.dtype("<" + _PLY_TYPES[props[0][1]]), np.dtype("<" + _PLY_TYPES[props[0][2]])
                lists = []
                for _ in range(count):
                    n = int(np.frombuffer(f.read(ct.itemsize), dtype=ct, count=1)[0])
                    lists.append(np.frombuffer(f.read(it.itemsize * n), dtype=it, count=n).astype(np.int64))
                data[name] = {props[0][-1]: lists}
    v = data["vertex"]
    vertices = np.stack([v["x"], v["y"], v["z"]], 1)
    polys = next(iter(data.get("face", {"vertex_indices": []}).values()))
    faces = np.array([[p[0], p[k], p[k + 1]] for p in polys for k in range(1, len(p) - 1)], dtype=np.int64).reshape(-1, 3)
    normals = np.stack([v["nx"], v["ny"], v["nz"]], 1) if "nx" in v else None
    uv = np.stack([v["texture_u"], v["texture_v"]], 1) if "texture_u" in v else None
    image = None
    if uv is not None and texfile is not None and os.path.exists(texfile):
        from PIL import Image
        image = np.asarray(Image.open(texfile).convert("RGB"))
    vcol = np.stack([v["red"], v["green"], v["blue"]], 1).astype(np.uint8) if "red" in v and image is None else None
    return SimpleMesh(vertices, faces, vertex_normals=normals, uv=uv if image is not None else None, texture=image, vertex_colors=vcol)


def save_ply(mesh, path, binary=True):
    """vertex positions + normals (+ colours) and triangles, binary little-endian or ascii"""
    verts = np.asarray(mesh.vertices, dtype=np.float32)
    nrm = np.asarray(mesh.vertex_normals, dtype=np.float32)
    col = getattr(mesh.visual, "vertex_colors", None)
    col = None if col is None else np.asarray(col)[:, :3].astype(np.uint8)
    faces = np.asarray(mesh.faces, dtype=np.int32)
    props = "".join(f"property float {n}\n" for n in ("x", "y", "z", "nx", "ny", "nz"))
    if col is not None:
        props += "".join(f"property uchar {n}\n" for n in ("red", "green", "blue"))
    header = (f"ply\nformat {'binary_little_endian' if binary else 'ascii'} 1.0\nelement vertex {len(verts)}\n{props}"
              f"element face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n")
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        if binary:
            fields = [("p", "<f4", (3,)), ("n", "<f4", (3,))] + ([("c", "u1", (3,))] if col is not None else [])
            rec = np.zeros(len(verts), dtype=fields)
            rec["p"], rec["n"] = verts, nrm
            if col is not None:
                rec["c"] = col
            f.write(rec.tobytes())
            frec = np.zeros(len(faces), dtype=[("k", "u1"), ("i", "<i4", (3,))])
            frec["k"], frec["i"] = 3, faces
            f.write(frec.tobytes())
        else:
            for i in range(len(verts)):
                row = list(verts[i]) + list(nrm[i])
                f.write((" ".join("%.9g" % x for x in row) + ("" if col is None else " %d %d %d" % tuple(col[i])) + "\n").encode())
            for a, b, c in faces:
                f.write(f"3 {a} {b} {c}\n".encode())


def load_mesh(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".obj":
        return load_obj(path)
    if ext == ".ply":
        return load_ply(path)
    raise NotImplementedError(f"mesh format '{ext}' is not supported (OBJ and PLY only)")
