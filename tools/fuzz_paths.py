#!/usr/bin/env python3
"""Development: random scenes through the entry points tools/fuzz_parity.py does not take, every
frame against the oracle's bit for bit (colour, depth, primitive id):

  drawables  update_drawables with one mesh copy per object, several texture groups, objects that
             are not rendered;
  views      sxe_draw_views: a set of random cameras in one launch set, every view against the
             oracle under that camera;
  flight     frames enqueued back to back into one to three images while transforms and camera
             change between them, read only after the last one;
  noalloc    sxe_draw_noalloc on the library's static work array, twice in a row.

  fuzz_paths.py [cases] [seed]
"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle
from renderer_amd import api, scenes
from tests.util import random_soup, soup_scene, IDENT


def equal(ref, got):
    return (np.array_equal(ref["color"], got["color"]) and np.array_equal(ref["depth"].view(np.uint32), got["depth"].view(np.uint32)) and
            np.array_equal(ref["prim"], got["prim"]))


def read(d, image):
    return {"color": d.read_color(image), "depth": d.read_depth(image), "prim": d.read_coverage(image)}


def soup(rng, translucent_ok=True):
    w, h = int(rng.integers(9, 300)), int(rng.integers(9, 260))
    n = int(rng.integers(1, 1200))
    n_obj = int(rng.integers(1, 7))
    split = np.diff(np.sort(np.concatenate([[0, n], rng.integers(0, n + 1, n_obj - 1)]))).tolist()
    split = [k for k in split if k > 0]
    texs = [rng.integers(0, 256, size=(int(rng.integers(1, 40)), int(rng.integers(1, 40)), 4), dtype=np.uint8) for _ in range(3)]
    translucent = translucent_ok and rng.random() < 0.25
    for k, t in enumerate(texs):
        if not (translucent and k < 2):
            t[..., 3] = 255
    s = soup_scene(random_soup(rng, n, w, h), uvs=rng.uniform(-0.5, 1.5, (3 * n, 2)), colors=rng.uniform(0, 1, (3 * n, 3)), textures=texs,
                   width=w, height=h, objects_split=split, tex_ids=rng.integers(0, 3, len(split)))
    return s, w, h


def spheres(rng):
    s = scenes.config_scene(int(rng.choice([2, 3, 5])), scale=float(rng.uniform(0.03, 0.1)))
    s.camera["position"] += rng.uniform(-3, 3, 3).astype(np.float32)
    return s, s.size[0], s.size[1]


def case_drawables(rng):
    s, w, h = soup(rng)
    s.objects["rendered"] = (rng.random(len(s.objects)) < 0.75).astype(s.objects["rendered"].dtype)
    shading = int(rng.integers(0, 3))
    models = np.tile(IDENT, (len(s.objects), 1))
    ref = oracle.render(s, models, IDENT, IDENT, w, h, shading=shading, threads=8)
    with api.HipDrawer(w, h, coverage=True, shading=shading) as d:
        d.set_textures(s.textures)
        d.update_drawables(s.drawables())
        d.update_uniforms(models)
        d.update_camera_matrices(IDENT, IDENT)
        d.draw(0)
        return equal(ref, read(d, 0)), "drawables %dx%d %d objects shading %d" % (w, h, len(s.objects), shading)


def case_views(rng):
    s, w, h = spheres(rng)
    n = int(rng.integers(1, 9))
    cams = np.repeat(s.camera, n)
    cams["position"] += rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    shading = int(rng.integers(0, 4))
    models = oracle.to_model(s.transforms)
    with api.HipDrawer(w, h, n_images=n, coverage=True, shading=shading) as d:
        d.update_scene(s)
        d.draw_views(cams, 0)
        d.sync()
        ok = True
        for v in range(n):
            view, proj = oracle.camera(cams[v:v + 1], w, h)
            ok = ok and equal(oracle.render(s, models, view, proj, w, h, shading=shading, threads=8), read(d, v))
    return ok, "views %s %dx%d %d views shading %d" % (s.name, w, h, n, shading)


def case_flight(rng):
    s, w, h = spheres(rng)
    n_images = int(rng.integers(1, 4))
    frames = int(rng.integers(2, 7))
    t = s.transforms.copy()
    cam = s.camera.copy()
    refs = []
    with api.HipDrawer(w, h, n_images=n_images, coverage=True) as d:
        d.update_scene(s)
        d.update_camera(cam)
        for f in range(frames):
            step = float(rng.uniform(10, 500))
            d.rotation_system(9, f, step)
            oracle.rotation_system(t, 9, f, step)
            if rng.random() < 0.5:
                cam["position"] += rng.uniform(-2, 2, 3).astype(np.float32)
                d.update_camera(cam)
            d.draw(f % n_images)
            refs.append(oracle.render(s, oracle.to_model(t), *oracle.camera(cam, w, h), w, h, threads=8))
        d.sync()
        ok = True
        for f in range(max(0, frames - n_images), frames):  # the frames the images still hold
            ok = ok and equal(refs[f], read(d, f % n_images))
    return ok, "flight %s %dx%d %d frames in %d images" % (s.name, w, h, frames, n_images)


def case_noalloc(rng):
    s, w, h = soup(rng) if rng.random() < 0.6 else spheres(rng)
    soup_like = s.name == "soup"
    shading = int(rng.integers(0, 3))
    models = np.tile(IDENT, (len(s.objects), 1)) if soup_like else oracle.to_model(s.transforms)
    view, proj = (IDENT, IDENT) if soup_like else oracle.camera(s.camera, w, h)
    ref = oracle.render(s, models, view, proj, w, h, shading=shading, threads=8)
    with api.HipDrawer(w, h, coverage=True, shading=shading) as d:
        d.update_scene(s)
        d.update_uniforms(models)
        d.update_camera_matrices(view, proj)
        ok = True
        for image in (0, 1):
            try:
                d.draw_noalloc(image)
            except api.SxeError as e:
                return e.code == api.E_CAPACITY, "noalloc refused: %s" % e  # (too large for the static array: not a mismatch)
            ok = ok and equal(ref, read(d, image))
    return ok, "noalloc %s %dx%d shading %d" % (s.name, w, h, shading)


CASES = [case_drawables, case_views, case_flight, case_noalloc]


def run(cases, seed0):
    oracle.build()
    bad = 0
    for case in range(cases):
        rng = np.random.default_rng(seed0 * 100000 + case)
        ok, desc = CASES[case % len(CASES)](rng)
        if not ok:
            bad += 1
            print("MISMATCH case %d: %s" % (case, desc), flush=True)
        if case % 200 == 199:
            print("%d cases so far, %d mismatches" % (case + 1, bad), flush=True)
    return bad


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    bad = run(cases, seed0)
    print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
