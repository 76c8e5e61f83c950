#!/usr/bin/env python3
"""Development: one case of tools/fuzz_parity.py again, with what differs printed.

  fuzz_case.py <seed0> <case>
"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.fuzz_parity as fz
from oracle import oracle
from tests.util import render_hip

seed0, case = int(sys.argv[1]), int(sys.argv[2])
oracle.build()
captured = {}
real_render, real_hip = oracle.render, fz.render_hip


def spy_oracle(*a, **k):
    captured["ref"] = real_render(*a, **k)
    captured["scene"] = a[0]
    captured["args"] = a[1:]
    captured.setdefault("kw", {})["oracle"] = k
    return captured["ref"]


def spy_hip(*a, **k):
    captured["got"] = real_hip(*a, **k)
    captured.setdefault("kw", {})["hip"] = k
    return captured["got"]


oracle.render, fz.render_hip = spy_oracle, spy_hip
ok = fz.one_case(seed0 * 1000 + case, case, True)
print("equal" if ok else "DIFFERENT")
ref, got = captured["ref"], captured["got"]
for plane in ("prim", "depth", "color"):
    a, b = ref[plane], got[plane]
    if plane == "depth":
        a, b = a.view(np.uint32), b.view(np.uint32)
    d = (a != b)
    if plane == "color":
        d = d.any(axis=-1)
    print(plane, "differs at", int(d.sum()), "pixels")
    for y, x in np.argwhere(d)[:12]:
        print("   (x %d, y %d): oracle %s  hip %s   prim oracle %s hip %s" % (x, y, a[y, x], b[y, x], ref["prim"][y, x], got["prim"][y, x]))
print("stats", got["stats"])

if not ok and len(sys.argv) > 3:
    # which triangle brings the difference about: the scene again with only its first k triangles
    # (the others moved off the screen, so that slots and draw order stay what they are)
    import copy
    s = captured["scene"]
    d = np.argwhere((ref["color"] != got["color"]).any(axis=-1))[0]
    y, x = int(d[0]), int(d[1])
    oracle.render, fz.render_hip = real_render, real_hip
    kw = captured["kw"]
    n = len(s.vertices) // 3
    first_bad = None
    for k in range(n + 1):
        sk = copy.deepcopy(s)
        sk.vertices["pos"][3 * k:] = (5.0, 5.0, 0.5)
        r = real_render(sk, *captured["args"], **kw["oracle"])
        g = real_hip(sk, **kw["hip"])
        same = np.array_equal(r["color"][y, x], g["color"][y, x])
        if not same and first_bad is None:
            first_bad = k
            print("first difference at pixel (%d, %d) with the first %d triangles: oracle %s hip %s; prim %s / %s depth %r / %r" % (
                x, y, k, r["color"][y, x], g["color"][y, x], r["prim"][y, x], g["prim"][y, x], r["depth"][y, x], g["depth"][y, x]))
            print("triangle %d (clip space):\n%s\nuv %s" % (k - 1, s.vertices["pos"][3 * (k - 1):3 * k], s.vertices["uv"][3 * (k - 1):3 * k]))
            rp = real_render(copy.deepcopy(sk), *captured["args"], **kw["oracle"])
            sk2 = copy.deepcopy(s); sk2.vertices["pos"][3 * (k - 1):] = (5.0, 5.0, 0.5)
            r2 = real_render(sk2, *captured["args"], **kw["oracle"]); g2 = real_hip(sk2, **kw["hip"])
            print("one triangle fewer: oracle %s hip %s prim %s / %s depth %r / %r" % (r2["color"][y, x], g2["color"][y, x], r2["prim"][y, x], g2["prim"][y, x], r2["depth"][y, x], g2["depth"][y, x]))
            break
    print("objects", s.objects, "meshes", s.meshes)

if not ok and len(sys.argv) > 3:
    # the same few triangles under every shading mode, with and without translucent textures
    for opaque in (False, True):
        for shading in range(4):
            sk = copy.deepcopy(s)
            sk.vertices["pos"][3 * first_bad:] = (5.0, 5.0, 0.5)
            if opaque:
                for t in sk.textures:
                    t[..., 3] = 255
            ko, kh = dict(kw["oracle"], shading=shading), dict(kw["hip"], shading=shading)
            r = real_render(sk, *captured["args"], **ko)
            g = real_hip(sk, **kh)
            nd = int((r["color"] != g["color"]).any(axis=-1).sum())
            print("opaque %s shading %d: %d pixels differ; at (%d, %d) oracle %s hip %s" % (opaque, shading, nd, x, y, r["color"][y, x], g["color"][y, x]))
    print("light", kw["oracle"].get("light"))
