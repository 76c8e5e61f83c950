#!/usr/bin/env python3
"""Development: many random scenes, HIP frame against the oracle's, bit for bit (colour, depth,
primitive id).  Triangle soups of every awkward kind at odd frame sizes (one in twelve with NaN, infinities and other wild values in it), random shading mode (the lit extension under random lights among them),
random scissor band, several objects and textures, some objects not rendered, the draw list of the reference
binary (SXE_FLAG_COMPAT_REFERENCE_INDEXING) and the BGRA target order now and then, through either upload entry
point; sphere fields under random cameras.

  fuzz_parity.py [cases] [seed]          (FUZZ_BIG=1: the soups' frames seven times as wide and high)
"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle
from renderer_amd import api, scenes
from tests.util import random_soup, soup_scene, render_hip, IDENT



BIG = os.environ.get("FUZZ_BIG", "") == "1"


def run(cases, seed0, verbose=True):
    """Number of cases whose HIP frame differs from the oracle's."""
    oracle.build()
    bad = 0
    for case in range(cases):
        bad += 0 if one_case(seed0 * 1000 + case, case, verbose) else 1
        if verbose and case % 500 == 499:
            print("%d cases so far, %d mismatches" % (case + 1, bad), flush=True)  # (a long run must be seen to be alive)
    return bad


def one_case(seed, case, verbose):
    rng = np.random.default_rng(seed)
    shading = int(rng.integers(0, 4))
    if case % 3 == 2:
        s = scenes.config_scene(int(rng.choice([2, 3, 5])), scale=float(rng.uniform(0.03, 0.12)))
        w, h = s.size
        s.camera["position"] += rng.uniform(-3, 3, 3).astype(np.float32)
        models = oracle.to_model(s.transforms)
        view, proj = oracle.camera(s.camera, w, h)
        desc = "%s %dx%d" % (s.name, w, h)
    else:
        w, h = int(rng.integers(9, 400)), int(rng.integers(9, 300))
        if BIG:  # frames of up to 2800 x 2100: thousands of tiles, triangles over hundreds of them
            w, h = 7 * w, 7 * h
        n = int(rng.integers(1, 1500))
        n_obj = int(rng.integers(1, 6))
        split = np.diff(np.sort(np.concatenate([[0, n], rng.integers(0, n + 1, n_obj - 1)]))).tolist()
        split = [k for k in split if k > 0]
        tris = random_soup(rng, n, w, h)
        if rng.random() < 0.3:
            # corners on a quarter-pixel grid and a few distinct depths: samples that lie exactly on
            # edges and corners (the top-left rule), exact depth ties, shared edges
            q = float(rng.choice([1.0, 2.0, 4.0]))
            px = np.round((tris[..., 0] + 1.0) * 0.5 * w * q) / q
            py = np.round((tris[..., 1] + 1.0) * 0.5 * h * q) / q
            on = np.abs(tris[..., :2]).max(axis=(1, 2)) < 1.4  # (the huge and the off-screen ones stay as they are)
            tris[on, :, 0] = (2.0 * px / w - 1.0)[on]
            tris[on, :, 1] = (2.0 * py / h - 1.0)[on]
            tris[..., 2] = np.round(tris[..., 2] * 16.0) / 16.0
        texs = [rng.integers(0, 256, size=(int(rng.integers(1, 40)), int(rng.integers(1, 40)), 4), dtype=np.uint8) for _ in range(3)]
        translucent = rng.random() < 0.25  # then the tiles take the ordered (blending) path
        for k, t in enumerate(texs):
            if not (translucent and k < 2):
                t[..., 3] = 255
        uvs, colors = rng.uniform(-0.5, 1.5, (3 * n, 2)).astype(np.float32), rng.uniform(0, 1, (3 * n, 3)).astype(np.float32)
        # one soup in twelve has values that are not finite, near the largest float, denormal or -0
        # among its positions, texture coordinates, colours and matrices (a generator of its own, so
        # that the other cases of a seed stay what they were)
        rng2 = np.random.default_rng(seed ^ 0x5EED)
        wild = rng2.random() < 1.0 / 12.0
        if wild:
            bad = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3.0e38, 1e-40, -0.0], np.float32)
            for arr, p in ((tris, 0.02), (uvs, 0.01), (colors, 0.01)):
                pick = rng2.random(arr.shape) < p
                arr[pick] = rng2.choice(bad, size=int(pick.sum()))
        s = soup_scene(tris, uvs=uvs, colors=colors, textures=texs, width=w,
                       height=h, objects_split=split, tex_ids=rng.integers(0, 3, len(split)))
        models = np.tile(IDENT, (len(s.objects), 1))
        if wild and rng2.random() < 0.5:
            models[int(rng2.integers(0, len(models))), int(rng2.integers(0, 16))] = rng2.choice(bad)
        view = proj = IDENT
        desc = "soup %d tris %dx%d %d objects%s%s" % (n, w, h, len(split), " translucent" if translucent else "", " wild" if wild else "")
    scissor = None
    if rng.random() < 0.4:
        a, b = sorted(rng.integers(0, h + 1, 2).tolist())
        if b > a:
            scissor = (a, b)
    light = None
    if shading == 3 and rng.random() < 0.7:  # (the lit extension: else under the default light)
        light = dict(to_light=rng.normal(size=3).tolist(), to_viewer=rng.normal(size=3).tolist(), ambient=float(rng.uniform(0, 0.5)),
                     diffuse=float(rng.uniform(0, 1.2)), specular=float(rng.uniform(0, 1)), shininess_log2=int(rng.integers(0, 9)))
    # the draw list as the reference binary records it (objects not rendered, texture groups, index
    # ranges that run over several objects' meshes), and the other byte order of the target
    compat, bgra = bool(rng.random() < 0.25), bool(rng.random() < 0.2)
    if compat or rng.random() < 0.15:
        s.objects["rendered"] = (rng.random(len(s.objects)) < 0.75).astype(np.uint32)
    flags = (oracle.FLAG_REFERENCE_INDEXING if compat else 0) | (oracle.FLAG_BGRA if bgra else 0)
    desc += " compat" * compat + " bgra" * bgra
    ref = oracle.render(s, models, view, proj, w, h, shading=shading, scissor=scissor, threads=8, light=light, flags=flags)
    got = render_hip(s, models=models, view=view, proj=proj, shading=shading, scissor=scissor, light=light,
                     reference_indexing=compat, bgra=bgra, use_drawables=bool(rng.random() < 0.3))
    y0, y1 = scissor if scissor else (0, h)
    ok = (np.array_equal(ref["color"][y0:y1], got["color"][y0:y1]) and
          np.array_equal(ref["depth"][y0:y1].view(np.uint32), got["depth"][y0:y1].view(np.uint32)) and
          np.array_equal(ref["prim"][y0:y1], got["prim"][y0:y1]))
    if not ok and verbose:
        print("MISMATCH case %d: %s shading %d scissor %s" % (case, desc, shading, scissor), flush=True)
    return ok


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    bad = run(cases, seed0)
    print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
