#!/usr/bin/env python3
"""Derive the scene constants the stepper needs from the reference's ASSET files.

Runs only where /root/reference exists (the build container). Output is the small
data file `tennisbot_rl_amd/assets/scene.json`, which IS committed: the GPU box has
no /root/reference and nothing at run time may read it.

What is read (data only, nothing is imported or executed):
  tennisbot/resources/racket.stl        binary STL  -> 2-D convex outline of the racket
  tennisbot/resources/racket.urdf:17-21 mass, inertia, inertial origin
  tennisbot/resources/ball.urdf:11-15,27-32   mass, inertia, sphere radius
  tennisbot/resources/court.urdf:19-24,43-47  two collision boxes
  tennisbot/resources/simplegoal.urdf:17-22   goal cylinder

The racket STL is an extruded 2-D outline (all vertices have x = +-0.0145), and a
dynamic mesh collides as its convex hull, so the collision
geometry is "prism over a convex polygon in the link (y, z) plane".
"""
import json
import os
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np

REF = os.environ.get("TB_REFERENCE", "/root/reference")
RES = os.path.join(REF, "tennisbot", "resources")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                   "tennisbot_rl_amd", "assets", "scene.json")


def read_binary_stl(path):
    raw = open(path, "rb").read()
    (ntri,) = struct.unpack("<I", raw[80:84])
    assert len(raw) == 84 + 50 * ntri, "not a binary STL"
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("attr", "<u2")])
    tris = np.frombuffer(raw[84:], dtype=rec, count=ntri)
    return ntri, tris["v"].reshape(-1, 3)


def convex_polygon_ccw(pts):
    """Andrew monotone chain on float64 points; returns CCW hull vertices."""
    p = sorted(set(map(tuple, pts)))

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lo = []
    for q in p:
        while len(lo) >= 2 and cross(lo[-2], lo[-1], q) <= 0:
            lo.pop()
        lo.append(q)
    up = []
    for q in reversed(p):
        while len(up) >= 2 and cross(up[-2], up[-1], q) <= 0:
            up.pop()
        up.append(q)
    return np.array(lo[:-1] + up[:-1], dtype=np.float64)


def floats(s):
    return [float(x) for x in s.split()]


def main():
    ntri, verts = read_binary_stl(os.path.join(RES, "racket.stl"))
    uniq = np.unique(verts, axis=0)
    xs = np.unique(uniq[:, 0])
    assert len(xs) == 2 and xs[0] == -xs[1], "racket.stl is expected to be an extrusion along x"
    outline = np.unique(uniq[:, 1:], axis=0).astype(np.float64)
    hull = convex_polygon_ccw(outline)
    area = 0.5 * float(np.sum(hull[:, 0] * np.roll(hull[:, 1], -1)
                              - np.roll(hull[:, 0], -1) * hull[:, 1]))
    assert area > 0, "hull must be counter-clockwise in (y, z)"

    racket = ET.parse(os.path.join(RES, "racket.urdf")).getroot().find("link/inertial")
    ball_root = ET.parse(os.path.join(RES, "ball.urdf")).getroot()
    ball_in = ball_root.find("link/inertial")
    court = ET.parse(os.path.join(RES, "court.urdf")).getroot()
    goal = ET.parse(os.path.join(RES, "simplegoal.urdf")).getroot()
    boxes = [floats(c.find("geometry/box").get("size")) for c in court.findall("link/collision")]
    cyl = goal.find("link/collision/geometry/cylinder")
    ri = racket.find("inertia")
    bi = ball_in.find("inertia")

    scene = {
        "_generated_by": "tools/extract_assets.py from the reference's asset files (data, not code)",
        "racket": {
            "stl_triangles": int(ntri),
            "stl_unique_vertices": int(len(uniq)),
            "outline_points": int(len(outline)),
            "half_thickness": float(xs[1]),
            "bbox_min": [float(v) for v in uniq.min(0)],
            "bbox_max": [float(v) for v in uniq.max(0)],
            "hull_area": area,
            "hull_yz_ccw": [[float(y), float(z)] for y, z in hull],
            "mass": float(racket.find("mass").get("value")),
            "inertia_diag": [float(ri.get("ixx")), float(ri.get("iyy")), float(ri.get("izz"))],
            "inertial_origin": floats(racket.find("origin").get("xyz")),
        },
        "ball": {
            "mass": float(ball_in.find("mass").get("value")),
            "inertia_diag": [float(bi.get("ixx")), float(bi.get("iyy")), float(bi.get("izz"))],
            "radius": float(ball_root.find("link/collision/geometry/sphere").get("radius")),
        },
        "court": {"ground_box_size": boxes[0], "net_box_size": boxes[1]},
        "goal": {"radius": float(cyl.get("radius")), "length": float(cyl.get("length"))},
    }
    with open(OUT, "w") as f:
        json.dump(scene, f, indent=1)
    print("wrote", os.path.normpath(OUT), "hull vertices:", len(hull), "area:", area)


if __name__ == "__main__":
    sys.exit(main())
