"""scene.json from a .blend file, without Blender.

The reference produces its scene description by running Blend/Export.py inside Blender (Export.py:226-300: walk
`scene.objects`, write cameras, point lights, spheres, cubes and planes into `..\\ASCII\\scene.json`).  Blender is not
available where this code runs, and most of the reference's scenes exist only as .blend files, so this module reads
the .blend container itself (file blocks + the embedded SDNA struct descriptions, nothing is executed from the file)
and evaluates exactly what Export.py asks Blender for:

  get_camera_data          Export.py:6-49      location, gaze = R @ (0,0,-1), up = R @ (0,1,0), lens, sensor, DOF, resolution
  get_point_light_data     Export.py:53-60     location, energy
  get_sphere_data          Export.py:62-78     world translation at frame_start / frame_end, radius = scale.x
  get_cube_data            Export.py:80-98     same + rotation_euler + scale.x
  get_plane_data           Export.py:100-111   matrix_world @ vertex.co for every mesh vertex, in vertex order
  get_material_data        Export.py:114-190   first Principled BSDF (or Diffuse BSDF) node of the first material
  find_connected_texture   Export.py:205-224   first image texture reachable from the Base Color input
  sample_object_transform  Export.py:229-237   location F-curves evaluated at the frame
  export_to_json           Export.py:240-300   object classification by type and mesh-name prefix, json.dump(indent=4)

Checked against the one pair the reference ships: Blend/dof.blend -> ASCII/scene.json (tests/test_blend_export.py).
Written for the files in the reference (Blender 4.5, little endian, 64-bit pointers, objects without parents or
constraints); anything else raises.  Floats are read as float32 and widened, which is what Python sees inside
Blender; matrix and vector arithmetic follows Blender's own routines (eulO_to_mat3 in double then float, float
matrix product, mathutils' float products summed in double).
"""
from __future__ import annotations

import json
import math
import os
import re
import struct

import numpy as np

F32 = np.float32


class BlendFile:
    """File blocks and SDNA of a .blend file; fields are read by name through the DNA stored in the file itself."""

    _FMT = {"char": "c", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "float": "f", "double": "d",
            "int64_t": "q", "uint64_t": "Q", "int8_t": "b", "uint8_t": "B", "int16_t": "h", "uint16_t": "H",
            "int32_t": "i", "uint32_t": "I", "long": "i", "ulong": "I"}

    def __init__(self, path: str):
        with open(path, "rb") as f:
            d = f.read()
        if d[:7] != b"BLENDER":
            raise ValueError(f"{path}: not an uncompressed .blend file")
        if d[7:8] != b"-" or d[8:9] != b"v":
            raise ValueError(f"{path}: only little-endian files with 64-bit pointers are supported")
        self.version = int(d[9:12])
        self.d = d
        self.blocks, self.by_addr = [], {}
        off = 12
        while off < len(d):
            code = d[off:off + 4]
            size, old, sdna, count = struct.unpack_from("<iQii", d, off + 4)
            b = dict(code=code.rstrip(b"\0").decode("latin1"), size=size, old=old, sdna=sdna, count=count, off=off + 24)
            self.blocks.append(b)
            self.by_addr.setdefault(old, b)
            if code == b"ENDB":
                break
            off += 24 + size
        self._parse_dna()

    def _parse_dna(self):
        blk = next(x for x in self.blocks if x["code"] == "DNA1")
        d, o = self.d, blk["off"]
        base = o

        def align(x):
            return base + ((x - base + 3) & ~3)

        def strings(tag, o):
            assert d[o:o + 4] == tag, (d[o:o + 4], tag)
            n, = struct.unpack_from("<i", d, o + 4)
            o += 8
            out = []
            for _ in range(n):
                z = d.index(b"\0", o)
                out.append(d[o:z].decode("latin1"))
                o = z + 1
            return out, align(o)

        assert d[o:o + 4] == b"SDNA"
        names, o = strings(b"NAME", o + 4)
        types, o = strings(b"TYPE", o)
        assert d[o:o + 4] == b"TLEN"
        tlen = struct.unpack_from("<%dh" % len(types), d, o + 4)
        o = align(o + 4 + 2 * len(types))
        assert d[o:o + 4] == b"STRC"
        n, = struct.unpack_from("<i", d, o + 4)
        o += 8
        self.structs, self.struct_by_name = [], {}
        type_len = dict(zip(types, tlen))
        for _ in range(n):
            t, nf = struct.unpack_from("<hh", d, o)
            o += 4
            lay, pos = {}, 0
            for _ in range(nf):
                ft, fn = struct.unpack_from("<hh", d, o)
                o += 4
                ftype, fname = types[ft], names[fn]
                is_ptr = fname.startswith("*") or fname.startswith("(*")
                dims = [int(x) for x in re.findall(r"\[(\d+)\]", fname)]
                count = int(np.prod(dims)) if dims else 1
                key = re.sub(r"\[.*", "", fname).strip("()").lstrip("*")
                esz = 8 if is_ptr else type_len[ftype]
                lay[key] = (pos, ftype, is_ptr, dims, esz)
                pos += esz * count
            s = dict(name=types[t], layout=lay, size=pos)
            self.structs.append(s)
            self.struct_by_name[s["name"]] = s

    # ---- access -------------------------------------------------------------------------------------------
    def struct_name(self, block) -> str:
        return self.structs[block["sdna"]]["name"]

    def block_at(self, addr: int):
        """The block that starts at the (file-time) address `addr`, or None for a null / dangling pointer."""
        return self.by_addr.get(addr) if addr else None

    def get(self, block, sname: str, path: str, index: int = 0):
        off = block["off"] + index * self.struct_by_name[sname]["size"]
        s = self.struct_by_name[sname]
        parts = path.split(".")
        for i, p in enumerate(parts):
            pos, ftype, is_ptr, dims, esz = s["layout"][p]
            off += pos
            if i + 1 < len(parts):
                s = self.struct_by_name[ftype]
                continue
            count = int(np.prod(dims)) if dims else 1
            if is_ptr:
                v = struct.unpack_from("<%dQ" % count, self.d, off)
                return v[0] if not dims else list(v)
            fmt = self._FMT.get(ftype)
            if fmt is None:
                raise KeyError(f"{sname}.{path} is a struct ({ftype})")
            if ftype == "char" and dims:
                raw = self.d[off:off + count]
                z = raw.find(b"\0")
                return raw[:z if z >= 0 else count].decode("utf8", "replace")
            v = struct.unpack_from("<%d%s" % (count, fmt), self.d, off)
            return v[0] if not dims else list(v)

    def pointer_array(self, addr: int, n: int):
        b = self.block_at(addr)
        return list(struct.unpack_from("<%dQ" % n, self.d, b["off"])) if b and n > 0 else []

    def list_base(self, block, sname: str, path: str, elem: str):
        """Blocks of a ListBase whose elements start with next / prev pointers."""
        out, p, seen = [], self.get(block, sname, path + ".first"), set()
        while p and p not in seen:
            seen.add(p)
            e = self.block_at(p)
            if e is None:
                break
            out.append(e)
            p = self.get(e, elem, "next")
        return out

    def c_string(self, addr: int) -> str:
        b = self.block_at(addr)
        return self.d[b["off"]:b["off"] + b["size"]].split(b"\0")[0].decode("utf8", "replace") if b else ""

    def id_name(self, block) -> str:
        """bpy's `.name`: ID.name without its two-letter type prefix."""
        return self.get(block, self.struct_name(block), "id.name")[2:]


# ---- Blender's arithmetic -----------------------------------------------------------------------------------

def _eul_to_mat3(e):
    """eulO_to_mat3 for the XYZ order (math_rotation.c): double trigonometry, float result, m[col][row]."""
    ci, cj, ch = math.cos(float(e[0])), math.cos(float(e[1])), math.cos(float(e[2]))
    si, sj, sh = math.sin(float(e[0])), math.sin(float(e[1])), math.sin(float(e[2]))
    cc, cs, sc, ss = ci * ch, ci * sh, si * ch, si * sh
    m = np.zeros((3, 3), dtype=F32)
    m[0][0] = cj * ch; m[1][0] = sj * sc - cs; m[2][0] = sj * cc + ss
    m[0][1] = cj * sh; m[1][1] = sj * ss + cc; m[2][1] = sj * cs - sc
    m[0][2] = -sj;     m[1][2] = cj * si;      m[2][2] = cj * ci
    return m


def _object_matrix(loc, rot, scale):
    """BKE_object_to_mat4 for an unparented XYZ-Euler object without deltas: rotation x diag(scale), loc in column 3."""
    r = _eul_to_mat3(rot)
    m = np.zeros((4, 4), dtype=F32)
    for c in range(3):
        for row in range(3):
            m[c][row] = F32(r[c][row]) * F32(scale[c])          # mul_m3_m3m3(mat, rmat, smat) with a diagonal smat
    m[3][0], m[3][1], m[3][2], m[3][3] = F32(loc[0]), F32(loc[1]), F32(loc[2]), F32(1.0)
    return m


def _mat_mul_vec(m, v, n):
    """mathutils Matrix @ Vector (column_vector_multiplication): float products summed in double, float result;
    a 4x4 matrix times a 3D vector uses w = 1."""
    vec = [F32(x) for x in v] + ([F32(1.0)] if n == 4 else [])
    out = []
    for row in range(3):
        dot = 0.0
        for col in range(n):
            dot += float(F32(m[col][row]) * vec[col])
        out.append(F32(dot))
    return out


def _normalized(v):
    """mathutils Vector.normalized (normalize_vn_vn): squared length in double, float sqrt, multiply by 1/len."""
    d = sum(float(x) * float(x) for x in v)
    if not d > 1.0e-35:
        return [F32(0.0)] * len(v)
    inv = F32(1.0) / F32(math.sqrt(d))
    return [F32(x) * inv for x in v]


def _f(x) -> float:
    return float(F32(x))


# ---- animation ----------------------------------------------------------------------------------------------

def _fcurve_value(keys, extend, frame: float) -> float:
    """evaluate_fcurve for keyframed curves without modifiers.  keys = [(handle_left, knot, handle_right, ipo)] with
    (x, y) pairs.  Exact for frames on a key, outside the keyed range with constant extrapolation, and for constant
    and linear segments; Bezier segments are solved numerically (x(t) = frame by bisection in double)."""
    if not keys:
        raise ValueError("F-curve without keyframes")
    first, last = keys[0], keys[-1]
    if frame <= first[1][0]:
        if extend == 1 and len(keys) > 1 and frame < first[1][0]:
            raise NotImplementedError("linear extrapolation before the first key")
        return first[1][1]
    if frame >= last[1][0]:
        if extend == 1 and len(keys) > 1 and frame > last[1][0]:
            raise NotImplementedError("linear extrapolation after the last key")
        return last[1][1]
    for a, b in zip(keys, keys[1:]):
        if abs(frame - b[1][0]) < 1e-4:
            return b[1][1]
        if a[1][0] <= frame < b[1][0]:
            ipo = a[3]
            if ipo == 0:
                return a[1][1]
            if ipo == 1:
                t = (frame - a[1][0]) / (b[1][0] - a[1][0])
                return a[1][1] + t * (b[1][1] - a[1][1])
            if ipo == 2:
                x0, x1, x2, x3 = a[1][0], a[2][0], b[0][0], b[1][0]
                y0, y1, y2, y3 = a[1][1], a[2][1], b[0][1], b[1][1]
                # BKE_fcurve_correct_bezpart: keep the handles inside the segment
                h1, h2, span = x1 - x0, x3 - x2, x3 - x0
                if h1 + h2 > span and h1 + h2 != 0:
                    fac = span / (h1 + h2)
                    x1, y1 = x0 + fac * h1, y0 + fac * (y1 - y0)
                    x2, y2 = x3 - fac * h2, y3 - fac * (y3 - y2)
                bez = lambda p0, p1, p2, p3, t: ((1 - t) ** 3) * p0 + 3 * ((1 - t) ** 2) * t * p1 + 3 * (1 - t) * t * t * p2 + t ** 3 * p3
                lo, hi = 0.0, 1.0
                for _ in range(80):
                    mid = 0.5 * (lo + hi)
                    if bez(x0, x1, x2, x3, mid) < frame:
                        lo = mid
                    else:
                        hi = mid
                return bez(y0, y1, y2, y3, 0.5 * (lo + hi))
            raise NotImplementedError(f"F-curve interpolation mode {ipo}")
    return last[1][1]


# ---- the exporter -------------------------------------------------------------------------------------------

OB_MESH, OB_MBALL, OB_LAMP, OB_CAMERA = 1, 5, 10, 11
LA_LOCAL = 0


class _Exporter:
    def __init__(self, path: str):
        self.b = BlendFile(path)
        b = self.b
        scenes = [x for x in b.blocks if x["code"] == "SC"]
        if len(scenes) != 1:
            raise ValueError("expected exactly one scene")
        self.scene = scenes[0]
        self.frame_start = b.get(self.scene, "Scene", "r.sfra")
        self.frame_end = b.get(self.scene, "Scene", "r.efra")

    # bpy.context.scene.objects: every object of the scene's collection tree, parents before children collections,
    # each collection's objects in list order, an object once
    def scene_objects(self):
        b = self.b
        out, seen = [], set()

        def walk(coll):
            for co in b.list_base(coll, "Collection", "gobject", "CollectionObject"):
                ob = b.block_at(b.get(co, "CollectionObject", "ob"))
                if ob is not None and ob["old"] not in seen:
                    seen.add(ob["old"]); out.append(ob)
            for ch in b.list_base(coll, "Collection", "children", "CollectionChild"):
                c = b.block_at(b.get(ch, "CollectionChild", "collection"))
                if c is not None:
                    walk(c)

        master = b.block_at(b.get(self.scene, "Scene", "master_collection"))
        if master is None:
            raise ValueError("scene without a master collection")
        walk(master)
        return out

    def check_plain(self, ob):
        b = self.b
        if b.get(ob, "Object", "parent") or b.get(ob, "Object", "constraints.first"):
            raise NotImplementedError(f"{b.id_name(ob)}: parented or constrained objects")
        if b.get(ob, "Object", "rotmode") != 1:
            raise NotImplementedError(f"{b.id_name(ob)}: rotation mode other than XYZ Euler")
        if any(b.get(ob, "Object", "dloc")) or any(b.get(ob, "Object", "drot")) or b.get(ob, "Object", "dscale") != [1.0, 1.0, 1.0]:
            raise NotImplementedError(f"{b.id_name(ob)}: delta transforms")

    def fcurves(self, ob):
        """{(rna_path, array_index): (keys, extend)} of the object's assigned action slot."""
        b = self.b
        adt = b.block_at(b.get(ob, "Object", "adt"))
        if adt is None:
            return {}
        act = b.block_at(b.get(adt, "AnimData", "action"))
        if act is None:
            return {}
        handle = b.get(adt, "AnimData", "slot_handle")
        curves = []
        n = b.get(act, "bAction", "strip_keyframe_data_array_num")
        for p in b.pointer_array(b.get(act, "bAction", "strip_keyframe_data_array"), n):
            skd = b.block_at(p)
            nb = b.get(skd, "ActionStripKeyframeData", "channelbag_array_num")
            for q in b.pointer_array(b.get(skd, "ActionStripKeyframeData", "channelbag_array"), nb):
                bag = b.block_at(q)
                if b.get(bag, "ActionChannelBag", "slot_handle") != handle:
                    continue
                nf = b.get(bag, "ActionChannelBag", "fcurve_array_num")
                curves += [b.block_at(x) for x in b.pointer_array(b.get(bag, "ActionChannelBag", "fcurve_array"), nf)]
        if not curves and n == 0:                              # action written by an older Blender: one flat list
            curves = b.list_base(act, "bAction", "curves", "FCurve")
        out = {}
        for fc in curves:
            if b.get(fc, "FCurve", "driver") or b.get(fc, "FCurve", "modifiers.first"):
                raise NotImplementedError("F-curve drivers / modifiers")
            bez = b.block_at(b.get(fc, "FCurve", "bezt"))
            keys = []
            for k in range(b.get(fc, "FCurve", "totvert")):
                v = b.get(bez, "BezTriple", "vec", k)
                ipo = b.get(bez, "BezTriple", "ipo", k)
                keys.append(((v[0], v[1]), (v[3], v[4]), (v[6], v[7]), ipo[0] if isinstance(ipo, bytes) else int(ipo)))
            out[(b.c_string(b.get(fc, "FCurve", "rna_path")), b.get(fc, "FCurve", "array_index"))] = (keys, b.get(fc, "FCurve", "extend"))
        return out

    def transform_at(self, ob, frame):
        """loc, rot, scale of the object with its F-curves evaluated at `frame` (scene.frame_set(frame))."""
        b = self.b
        vals = {"location": list(b.get(ob, "Object", "loc")), "rotation_euler": list(b.get(ob, "Object", "rot")),
                "scale": list(b.get(ob, "Object", "size"))}
        for (path, idx), (keys, extend) in self.fcurves(ob).items():
            if path in vals:
                vals[path][idx] = _f(_fcurve_value(keys, extend, float(frame)))
        return vals["location"], vals["rotation_euler"], vals["scale"]

    def sample_object_transform(self, ob, frame):                  # Export.py:229-237
        loc, rot, scale = self.transform_at(ob, frame)
        m = _object_matrix(loc, rot, scale)
        return {"x": float(m[3][0]), "y": float(m[3][1]), "z": float(m[3][2])}

    # ---- materials (Export.py:114-224) -----------------------------------------------------------------------
    def first_material(self, ob):
        b = self.b
        data = b.block_at(b.get(ob, "Object", "data"))
        sname = b.struct_name(data)
        if b.get(data, sname, "totcol") <= 0:
            return None
        ptrs = b.pointer_array(b.get(data, sname, "mat"), 1)
        return b.block_at(ptrs[0]) if ptrs else None

    def socket_value(self, sock):
        b = self.b
        v = b.block_at(b.get(sock, "bNodeSocket", "default_value"))
        if v is None:
            return None
        sname = b.struct_name(v)
        return b.get(v, sname, "value")

    def find_connected_texture(self, node, links, visited=None):
        b = self.b
        visited = set() if visited is None else visited
        if node["old"] in visited:
            return None
        visited.add(node["old"])
        if b.get(node, "bNode", "idname") == "ShaderNodeTexImage":
            img = b.block_at(b.get(node, "bNode", "id"))
            if img is not None:
                # os.path.basename on the platform Blender ran on (Windows paths in these files) or image.name
                path = b.get(img, "Image", "name")
                base = re.split(r"[\\/]", path)[-1]
                return base or b.id_name(img)
        for sock in b.list_base(node, "bNode", "inputs", "bNodeSocket"):
            for ln in links:
                if b.get(ln, "bNodeLink", "tosock") == sock["old"]:
                    frm = b.block_at(b.get(ln, "bNodeLink", "fromnode"))
                    if frm is not None:
                        tex = self.find_connected_texture(frm, links, visited)
                        if tex:
                            return tex
        return None

    def get_material_data(self, ob):
        b = self.b
        default = {"diffuse": {"r": 1.0, "g": 1.0, "b": 1.0}, "specular": {"r": 0.3, "g": 0.3, "b": 0.3}, "shininess": 32.0,
                   "transparency": 0.0, "ior": 1.0, "texture": None}
        mat = self.first_material(ob)
        if mat is None:
            return default
        diffuse, specular = dict(default["diffuse"]), dict(default["specular"])
        shininess, transparency, ior, texture = 32.0, 0.0, 1.0, None
        tree = b.block_at(b.get(mat, "Material", "nodetree"))
        if tree is not None:
            nodes = b.list_base(tree, "bNodeTree", "nodes", "bNode")
            links = b.list_base(tree, "bNodeTree", "links", "bNodeLink")
            principled = False
            for node in nodes:
                if b.get(node, "bNode", "idname") != "ShaderNodeBsdfPrincipled":
                    continue
                principled = True
                inputs = {}
                for sock in b.list_base(node, "bNode", "inputs", "bNodeSocket"):
                    inputs.setdefault(b.get(sock, "bNodeSocket", "name"), sock)      # node.inputs['Name']: first match
                col = inputs["Base Color"]
                c = self.socket_value(col)
                diffuse = {"r": _f(c[0]), "g": _f(c[1]), "b": _f(c[2])}
                for ln in links:
                    if b.get(ln, "bNodeLink", "tosock") == col["old"]:             # color_input.is_linked, links[0]
                        frm = b.block_at(b.get(ln, "bNodeLink", "fromnode"))
                        tex = self.find_connected_texture(frm, links) if frm is not None else None
                        if tex:
                            texture = tex
                        break
                if "Specular" in inputs:
                    s = min(_f(self.socket_value(inputs["Specular"])), 0.3)
                    specular = {"r": s, "g": s, "b": s}
                if "Roughness" in inputs:
                    shininess = max(0.0, (1.0 - _f(self.socket_value(inputs["Roughness"]))) * 128.0)
                if "Transmission" in inputs:
                    transparency = max(0.0, min(1.0, _f(self.socket_value(inputs["Transmission"]))))
                if "IOR" in inputs:
                    ior = _f(self.socket_value(inputs["IOR"]))
                break
            if not principled:
                for node in nodes:
                    if b.get(node, "bNode", "idname") == "ShaderNodeBsdfDiffuse":
                        for sock in b.list_base(node, "bNode", "inputs", "bNodeSocket"):
                            if b.get(sock, "bNodeSocket", "name") == "Color":
                                c = self.socket_value(sock)
                                diffuse = {"r": _f(c[0]), "g": _f(c[1]), "b": _f(c[2])}
                                specular = {"r": 0.5, "g": 0.5, "b": 0.5}
                                shininess = 32.0
                                break
                        else:
                            continue
                        break
        return {"diffuse": diffuse, "specular": specular, "shininess": shininess, "transparency": transparency, "ior": ior,
                "texture": texture if texture else None}

    # ---- objects ----------------------------------------------------------------------------------------------
    def get_camera_data(self, ob):                                     # Export.py:6-49
        b = self.b
        self.check_plain(ob)
        loc, rot, scale = self.transform_at(ob, self.frame_start)
        m = _object_matrix(loc, rot, scale)
        fwd = _normalized(_mat_mul_vec(m, (0.0, 0.0, -1.0), 3))
        up = _normalized(_mat_mul_vec(m, (0.0, 1.0, 0.0), 3))
        cam = b.block_at(b.get(ob, "Object", "data"))
        return {
            "location": {"x": _f(loc[0]), "y": _f(loc[1]), "z": _f(loc[2])},
            "gaze_vector": {"x": float(fwd[0]), "y": float(fwd[1]), "z": float(fwd[2])},
            "up_vector": {"x": float(up[0]), "y": float(up[1]), "z": float(up[2])},
            "aperture": _f(b.get(cam, "Camera", "dof.aperture_fstop")),
            "focal_distance": _f(b.get(cam, "Camera", "dof.focus_distance")),
            "focal_length": _f(b.get(cam, "Camera", "lens")),
            "sensor": {"width": _f(b.get(cam, "Camera", "sensor_x")), "height": _f(b.get(cam, "Camera", "sensor_y"))},
            "film_resolution": {"width": b.get(self.scene, "Scene", "r.xsch"), "height": b.get(self.scene, "Scene", "r.ysch")},
        }

    def get_point_light_data(self, ob):                                # Export.py:53-60
        b = self.b
        loc, _, _ = self.transform_at(ob, self.frame_start)
        la = b.block_at(b.get(ob, "Object", "data"))
        lay = b.struct_by_name["Lamp"]["layout"]
        energy = b.get(la, "Lamp", "energy_new" if "energy_new" in lay else "energy")
        return {"location": {"x": _f(loc[0]), "y": _f(loc[1]), "z": _f(loc[2])}, "radiant_intensity": _f(energy)}

    def get_sphere_data(self, ob):                                     # Export.py:62-78
        self.check_plain(ob)
        start = self.sample_object_transform(ob, self.frame_start)
        end = self.sample_object_transform(ob, self.frame_end)
        _, _, scale = self.transform_at(ob, self.frame_end)             # read after the last frame_set
        return {"start_location": start, "end_location": end, "radius": _f(scale[0]), "material": self.get_material_data(ob)}

    def get_cube_data(self, ob):                                       # Export.py:80-98
        self.check_plain(ob)
        start = self.sample_object_transform(ob, self.frame_start)
        end = self.sample_object_transform(ob, self.frame_end)
        _, rot, scale = self.transform_at(ob, self.frame_end)
        return {"start_location": start, "end_location": end,
                "rotation": {"x": _f(rot[0]), "y": _f(rot[1]), "z": _f(rot[2])}, "scale": _f(scale[0]),
                "material": self.get_material_data(ob)}

    def mesh_vertices(self, mesh):
        b = self.b
        n = b.get(mesh, "Mesh", "totvert")
        layers = b.block_at(b.get(mesh, "Mesh", "vdata.layers"))
        for k in range(b.get(mesh, "Mesh", "vdata.totlayer")):
            if b.get(layers, "CustomDataLayer", "type", k) == 48 and b.get(layers, "CustomDataLayer", "name", k) == "position":
                data = b.block_at(b.get(layers, "CustomDataLayer", "data", k))
                return np.frombuffer(b.d, dtype="<f4", count=3 * n, offset=data["off"]).reshape(n, 3)
        raise NotImplementedError("mesh without a 'position' attribute layer")

    def get_plane_data(self, ob):                                      # Export.py:100-111
        b = self.b
        self.check_plain(ob)
        # the export loop has called frame_set(frame_end) for any sphere / cube before this plane and never resets the
        # frame inside the loop; planes in the reference's files are not animated, so the frame does not matter here
        loc, rot, scale = self.transform_at(ob, self.frame_start)
        m = _object_matrix(loc, rot, scale)
        corners = []
        for v in self.mesh_vertices(b.block_at(b.get(ob, "Object", "data"))):
            c = _mat_mul_vec(m, v, 4)
            corners.append({"x": float(c[0]), "y": float(c[1]), "z": float(c[2])})
        return {"corners": corners, "material": self.get_material_data(ob)}

    def export(self) -> dict:                                          # Export.py:240-300
        b = self.b
        cameras, lights, spheres, cubes, planes = [], [], [], [], []
        for ob in self.scene_objects():
            typ = b.get(ob, "Object", "type")
            data = b.block_at(b.get(ob, "Object", "data"))
            dname = b.id_name(data) if data is not None else ""
            if typ == OB_CAMERA:
                cameras.append(self.get_camera_data(ob))
            elif typ == OB_LAMP and b.get(data, "Lamp", "type") == LA_LOCAL:
                lights.append(self.get_point_light_data(ob))
            elif (typ == OB_MESH and dname.startswith("Sphere")) or typ == OB_MBALL:
                spheres.append(self.get_sphere_data(ob))
            elif typ == OB_MESH and dname.startswith("Cube"):
                cubes.append(self.get_cube_data(ob))
            elif typ == OB_MESH and dname.startswith("Plane"):
                planes.append(self.get_plane_data(ob))
        return {"properties": {"cameras": cameras, "point_lights": lights, "spheres": spheres, "cubes": cubes, "planes": planes}}


def export_scene(blend_path: str) -> dict:
    """The dictionary Export.py's export_to_json builds for the scene stored in `blend_path`."""
    return _Exporter(os.fspath(blend_path)).export()


def export_scene_json(blend_path: str) -> str:
    """The text Export.py writes: json.dump(data, f, indent=4)."""
    return json.dumps(export_scene(blend_path), indent=4)


if __name__ == "__main__":
    import sys
    sys.stdout.write(export_scene_json(sys.argv[1]))
