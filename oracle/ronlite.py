"""Minimal RON reader for the reference's scene / settings files (test infrastructure only).

Independent of the product's C++ loader so that the two can be checked against each other.
Accepts what the reference's files use (scene.rs:5-14, camera.rs:3-12, sphere.rs:3-13,
materials.rs:10-13,38-41,67-71,98-101): unnamed structs `( name: value, ... )`, lists `[ ... ]`,
numbers, `//` line comments and trailing commas.
"""
import re

import numpy as np

_TOKEN = re.compile(r"\s*(?:(//[^\n]*)|([()\[\],:])|([A-Za-z_][A-Za-z0-9_]*)|([-+]?[0-9.][0-9.eE+-]*))")


def _tokens(text):
    pos, out = 0, []
    text = text.rstrip()
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise ValueError(f"RON: bad character at offset {pos}: {text[pos:pos + 20]!r}")
        pos = m.end()
        if m.group(1):
            continue
        if m.group(2):
            out.append(m.group(2))
        elif m.group(3):
            out.append(("id", m.group(3)))
        else:
            out.append(("num", m.group(4)))
    return out


def _parse(tok, i):
    t = tok[i]
    if t == "(":
        i += 1
        fields, positional = {}, []
        while tok[i] != ")":
            if isinstance(tok[i], tuple) and tok[i][0] == "id" and tok[i + 1] == ":":
                name = tok[i][1]
                val, i = _parse(tok, i + 2)
                fields[name] = val
            else:
                val, i = _parse(tok, i)
                positional.append(val)
            if tok[i] == ",":
                i += 1
        return (fields if fields or not positional else positional), i + 1
    if t == "[":
        i += 1
        items = []
        while tok[i] != "]":
            val, i = _parse(tok, i)
            items.append(val)
            if tok[i] == ",":
                i += 1
        return items, i + 1
    if isinstance(t, tuple) and t[0] == "num":
        s = t[1]
        return (float(s) if any(c in s for c in ".eE") else int(s)), i + 1
    if isinstance(t, tuple) and t[0] == "id":
        # optional struct name in front of '(' (e.g. `Scene(`), or a bare identifier
        if i + 1 < len(tok) and tok[i + 1] == "(":
            return _parse(tok, i + 1)
        return t[1], i + 1
    raise ValueError(f"RON: unexpected token {t!r}")


def loads(text):
    tok = _tokens(text)
    val, i = _parse(tok, 0)
    if i != len(tok):
        raise ValueError("RON: trailing tokens")
    return val


def load(path):
    with open(path, "r") as f:
        return loads(f.read())


SPHERE_DT = np.dtype([("pos", "<f4", 3), ("rad", "<f4"), ("mat_type", "<u4"), ("mat_index", "<u4"),
                      ("pad", "<u4", 2)])
VEC16_DT = np.dtype([("v", "<f4", 3), ("w", "<f4")])
NODE_DT = np.dtype([("mn", "<f4", 3), ("pad0", "<u4"), ("mx", "<f4", 3), ("pad1", "<u4"),
                    ("type", "<u4"), ("index", "<u4"), ("pad2", "<u4", 2)])
assert SPHERE_DT.itemsize == 32 and VEC16_DT.itemsize == 16 and NODE_DT.itemsize == 48


def _v3(d):
    return [np.float32(d["x"]), np.float32(d["y"]), np.float32(d["z"])]


def scene_blobs(scene):
    """Scene dict -> the reference's GPU byte layouts (sphere.rs:26-43, materials.rs Bytes impls)."""
    sp = np.zeros(len(scene["spheres"]), SPHERE_DT)
    for i, s in enumerate(scene["spheres"]):
        sp[i]["pos"] = _v3(s["pos"])
        sp[i]["rad"] = s["rad"]
        sp[i]["mat_type"] = s["mat_type"]
        sp[i]["mat_index"] = s["mat_index"]

    def vec16(items, key, wkey=None):
        a = np.zeros(len(items), VEC16_DT)
        for i, m in enumerate(items):
            a[i]["v"] = _v3(m[key])
            if wkey:
                a[i]["w"] = m[wkey]
        return a

    cam = scene["camera"]
    return {
        "camera": (np.array(_v3(cam["pos"]), np.float32), np.float32(cam["yaw"]), np.float32(cam["pitch"]),
                   np.float32(cam["vfov"])),
        "spheres": sp,
        "lights": vec16(scene["lights"], "colour"),
        "lambertians": vec16(scene["lambertians"], "albedo"),
        "metals": vec16(scene["metals"], "albedo", "roughness"),
        "glass": np.array([g["ir"] for g in scene["glass"]], np.float32),
    }


def default_scene():
    """Scene::default() (scene.rs:15-31) + CameraSettings::default() (camera.rs:13-22)."""
    v = lambda x, y, z: {"x": x, "y": y, "z": z}
    return {
        "camera": {"pos": v(0.0, 0.0, -3.0), "yaw": 0.0, "pitch": 0.0, "vfov": 60.0},
        "spheres": [
            {"pos": v(1.0, 0.0, 0.0), "rad": 1.0, "mat_type": 1, "mat_index": 0},
            {"pos": v(-1.0, -0.0, -0.0), "rad": 1.0, "mat_type": 0, "mat_index": 0},
            {"pos": v(0.0, 1.0, 0.0), "rad": 1.0, "mat_type": 3, "mat_index": 0},
            {"pos": v(-0.0, -1.0, -0.0), "rad": 1.0, "mat_type": 2, "mat_index": 0},
        ],
        "lights": [{"colour": v(1.0, 1.0, 1.0)}],
        "lambertians": [{"albedo": v(0.5, 0.5, 0.5)}],
        "metals": [{"albedo": v(0.5, 0.5, 0.5), "roughness": 0.5}],
        "glass": [{"ir": 1.5}],
    }
