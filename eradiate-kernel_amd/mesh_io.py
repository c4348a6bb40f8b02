"""Triangle-mesh ingestion for the `obj` and `ply` shape plugins.

Follows /root/reference/src/shapes/obj.cpp:96-330 and src/shapes/ply.cpp (header grammar, list-typed faces, ascii /
binary_little_endian / binary_big_endian bodies) as far as the render path needs: positions, optional normals and
texture coordinates, faces triangulated as fans in file order (the face order is the primitive order, which decides
ties between coincident hits).  Like the reference's loaders, vertices are transformed by `to_world` at load time and
missing normals are computed with the angle-weighted scheme of Mesh::recompute_vertex_normals
(src/librender/mesh.cpp:200-254) unless `face_normals` is set.
"""
import os
import struct

import numpy as np

f32 = np.float32


def _fail(kind, name, msg):
    raise RuntimeError('Error while loading %s file "%s": %s' % (kind, name, msg))


# --------------------------------------------------------------------------------------------- OBJ
def read_obj(path, flip_tex_coords=True):
    """-> (positions (n,3) f32, normals (n,3) f32 | None, texcoords (n,2) f32 | None, faces (m,3) u32).
    One output vertex per distinct (v, vt, vn) triple, numbered in order of first use (obj.cpp:196-262)."""
    name = os.path.basename(path)
    if not os.path.exists(path):
        _fail("OBJ", name, "file not found")
    vs, vns, vts, tris = [], [], [], []
    vertex_map, keys = {}, []
    with open(path, "r", errors="replace") as fh:
        for line in fh:
            cur = line.strip(" \t\r\n")
            if len(cur) >= 1024:
                _fail("OBJ", name, "file contains an excessively long line! (%i characters)" % len(cur))
            try:
                if cur.startswith(("v ", "v\t")):
                    vs.append([f32(x) for x in cur[2:].split()[:3]])
                    if len(vs[-1]) != 3:
                        raise ValueError
                elif cur.startswith(("vn ", "vn\t")):
                    vns.append([f32(x) for x in cur[3:].split()[:3]])
                    if len(vns[-1]) != 3:
                        raise ValueError
                elif cur.startswith(("vt ", "vt\t")):
                    uv = [f32(x) for x in cur[3:].split()[:2]]
                    if len(uv) != 2:
                        raise ValueError
                    if flip_tex_coords:
                        uv[1] = f32(1) - uv[1]
                    vts.append(uv)
                elif cur.startswith(("f ", "f\t")):
                    tri, count = [0, 0, 0], 0
                    for tok in cur[2:].split():
                        parts = tok.split("/")
                        if len(parts) > 3:
                            raise ValueError
                        key = [0, 0, 0]
                        for k, part in enumerate(parts):
                            key[k] = int(part) if part else 0
                        if key[0] < 0 or key[1] < 0 or key[2] < 0:       # strtoul in the reference: relative indices are not supported
                            raise ValueError
                        if key[0] - 1 >= len(vs) or key[0] < 1:
                            _fail("OBJ", name, "reference to invalid vertex %i!" % key[0])
                        key = tuple(key)
                        vid = vertex_map.get(key)
                        if vid is None:
                            vid = vertex_map[key] = len(keys)
                            keys.append(key)
                        if count < 3:
                            tri[count] = vid
                        else:
                            tri[1], tri[2] = tri[2], vid
                        count += 1
                        if count >= 3:
                            tris.append(tuple(tri))
            except ValueError:
                _fail("OBJ", name, 'could not parse line "%s"' % cur)
    n = len(keys)
    vs = np.asarray(vs, dtype=f32).reshape(-1, 3)
    pos = np.zeros((n, 3), f32)
    nor = np.zeros((n, 3), f32) if vns else None
    tex = np.zeros((n, 2), f32) if vts else None
    vns = np.asarray(vns, dtype=f32).reshape(-1, 3)
    vts = np.asarray(vts, dtype=f32).reshape(-1, 2)
    for vid, key in enumerate(keys):
        pos[vid] = vs[key[0] - 1]
        if key[1] and tex is not None:
            if key[1] - 1 >= len(vts):
                _fail("OBJ", name, "reference to invalid texture coordinate %i!" % key[1])
            tex[vid] = vts[key[1] - 1]
        if key[2] and nor is not None:
            if key[2] - 1 >= len(vns):
                _fail("OBJ", name, "reference to invalid normal %i!" % key[2])
            nor[vid] = vns[key[2] - 1]
    return pos, nor, tex, np.asarray(tris, dtype=np.uint32).reshape(-1, 3)


# --------------------------------------------------------------------------------------------- PLY
_PLY_TYPES = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h", "ushort": "H", "uint16": "H",
              "int": "i", "int32": "i", "uint": "I", "uint32": "I", "float": "f", "float32": "f", "double": "d", "float64": "d"}


def read_ply(path):
    """-> (positions, normals | None, texcoords | None, faces); polygons are triangulated as fans (ply.cpp)."""
    name = os.path.basename(path)
    if not os.path.exists(path):
        _fail("PLY", name, "file not found")
    with open(path, "rb") as fh:
        data = fh.read()
    end = data.find(b"end_header")
    if not data.startswith(b"ply") or end < 0:
        _fail("PLY", name, "invalid PLY header")
    body_start = data.find(b"\n", end) + 1
    header = data[:end].decode("ascii", errors="replace").splitlines()
    fmt, elements = None, []
    for line in header[1:]:
        tok = line.split()
        if not tok or tok[0] in ("comment", "obj_info"):
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append({"name": tok[1], "count": int(tok[2]), "props": []})
        elif tok[0] == "property":
            if not elements:
                _fail("PLY", name, "property before element")
            if tok[1] == "list":
                elements[-1]["props"].append(("list", tok[2], tok[3], tok[4]))
            else:
                elements[-1]["props"].append(("scalar", tok[1], tok[2]))
        else:
            _fail("PLY", name, 'invalid PLY header: "%s"' % line)
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        _fail("PLY", name, "invalid PLY header: unknown format")
    for e in elements:
        for pr in e["props"]:
            for t in pr[1:-1]:
                if t not in _PLY_TYPES:
                    _fail("PLY", name, 'unknown type "%s"' % t)
    vertex, faces = None, []
    if fmt == "ascii":
        tokens = data[body_start:].split()
        pos_ = 0

        def take(t):
            nonlocal pos_
            v = tokens[pos_]
            pos_ += 1
            return float(v) if _PLY_TYPES[t] in "fd" else int(v)
    else:
        endian = "<" if fmt == "binary_little_endian" else ">"
        pos_ = body_start

        def take(t):
            nonlocal pos_
            c = _PLY_TYPES[t]
            v = struct.unpack_from(endian + c, data, pos_)[0]
            pos_ += struct.calcsize(c)
            return v
    try:
        for e in elements:
            if e["name"] == "vertex":
                names = [pr[2] for pr in e["props"]]
                if any(pr[0] == "list" for pr in e["props"]):
                    _fail("PLY", name, "list properties are not supported on vertices")
                if fmt != "ascii":                                       # fast path: fixed-size records
                    dt = np.dtype([(pr[2], endian + _PLY_TYPES[pr[1]]) for pr in e["props"]])
                    vertex = np.frombuffer(data, dtype=dt, count=e["count"], offset=pos_)
                    pos_ += dt.itemsize * e["count"]
                    vertex = {k: vertex[k].astype(np.float64) for k in names}
                else:
                    rows = [[take(pr[1]) for pr in e["props"]] for _ in range(e["count"])]
                    arr = np.asarray(rows, dtype=np.float64).reshape(e["count"], len(names))
                    vertex = {k: arr[:, i] for i, k in enumerate(names)}
            elif e["name"] == "face":
                for _ in range(e["count"]):
                    for pr in e["props"]:
                        if pr[0] == "list":
                            cnt = take(pr[1])
                            idx = [take(pr[2]) for _ in range(cnt)]
                            if pr[3] in ("vertex_index", "vertex_indices"):
                                if cnt < 3:
                                    _fail("PLY", name, "faces must have at least three vertices")
                                for k in range(1, cnt - 1):
                                    faces.append((idx[0], idx[k], idx[k + 1]))
                        else:
                            take(pr[1])
            else:                                                          # skip unknown elements
                for _ in range(e["count"]):
                    for pr in e["props"]:
                        if pr[0] == "list":
                            for _ in range(take(pr[1])):
                                take(pr[2])
                        else:
                            take(pr[1])
    except (IndexError, struct.error, ValueError):
        _fail("PLY", name, "premature end of file")
    if vertex is None or not all(k in vertex for k in "xyz"):
        _fail("PLY", name, "vertex positions are missing")
    pos = np.stack([vertex["x"], vertex["y"], vertex["z"]], 1).astype(f32)
    nor = np.stack([vertex["nx"], vertex["ny"], vertex["nz"]], 1).astype(f32) if all(k in vertex for k in ("nx", "ny", "nz")) else None
    tex = None
    for a, b in (("u", "v"), ("s", "t"), ("texture_u", "texture_v"), ("texture_s", "texture_t")):
        if a in vertex and b in vertex:
            tex = np.stack([vertex[a], vertex[b]], 1).astype(f32)
            break
    faces = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    if faces.size and (faces.min() < 0 or faces.max() >= len(pos)):
        _fail("PLY", name, "face references an invalid vertex")
    return pos, nor, tex, faces.astype(np.uint32)


# --------------------------------------------------------------------------------------------- shared
def _unit_angle(a, b):
    """enoki::unit_angle: numerically robust angle between unit vectors."""
    dot = np.sum(a * b, -1)
    t = f32(2) * np.arcsin(np.clip(f32(0.5) * np.linalg.norm(b - a, axis=-1), 0, 1)).astype(f32)
    t2 = f32(np.pi) - f32(2) * np.arcsin(np.clip(f32(0.5) * np.linalg.norm(b + a, axis=-1), 0, 1)).astype(f32)
    return np.where(dot >= 0, t, t2).astype(f32)


def _normalize(v):
    n = np.linalg.norm(v, axis=-1, keepdims=True).astype(f32)
    return (v / np.where(n > 0, n, 1)).astype(f32)


def compute_vertex_normals(pos, faces):
    """Mesh::recompute_vertex_normals (mesh.cpp:200-254): face normals weighted by the face angle at the vertex."""
    pos = np.asarray(pos, f32)
    v0, v1, v2 = pos[faces[:, 0]], pos[faces[:, 1]], pos[faces[:, 2]]
    s0, s1 = v1 - v0, v2 - v0
    n = np.cross(s0, s1).astype(f32)
    l2 = np.sum(n * n, -1)
    ok = l2 > 0
    n = np.where(ok[:, None], n / np.sqrt(np.where(ok, l2, 1))[:, None], 0).astype(f32)
    angles = np.stack([_unit_angle(_normalize(s0), _normalize(s1)),
                       _unit_angle(_normalize(v2 - v1), _normalize(v0 - v1)),
                       _unit_angle(_normalize(v0 - v2), _normalize(v1 - v2))], 1)
    out = np.zeros_like(pos)
    for j in range(3):
        np.add.at(out, faces[:, j], n * np.where(ok, angles[:, j], 0)[:, None])
    length = np.linalg.norm(out, axis=-1)
    res = np.where(length[:, None] != 0, out / np.where(length != 0, length, 1)[:, None], np.array([1, 0, 0], f32))
    return res.astype(f32)


def load_mesh(kind, path, to_world=None, face_normals=False, flip_tex_coords=True):
    """-> dict of world-space arrays for the in-memory `mesh` shape record."""
    pos, nor, tex, faces = read_obj(path, flip_tex_coords) if kind == "obj" else read_ply(path)
    if to_world is not None:
        m = np.asarray(to_world.matrix, f32)
        it = np.asarray(to_world.inverse_transpose, f32)
        pos = (pos @ m[:3, :3].T + m[:3, 3]).astype(f32)
        if nor is not None:
            nor = _normalize((nor @ it[:3, :3].T).astype(f32))
    if not np.all(np.isfinite(pos)):
        _fail(kind.upper(), os.path.basename(path), "mesh contains invalid vertex position data")
    if face_normals:
        nor = None
    elif nor is None and len(faces):
        nor = compute_vertex_normals(pos, faces)
    out = {"vertex_positions": pos, "faces": faces}
    if nor is not None:
        out["vertex_normals"] = nor
    if tex is not None:
        out["vertex_texcoords"] = tex
    return out
