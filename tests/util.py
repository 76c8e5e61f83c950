"""Helpers shared by the parity tests: run one scene through the oracle and through the HIP library."""
import numpy as np

from renderer_amd import api


def scene_matrices(orc, scene, width, height):
    models = orc.to_model(scene.transforms)
    view, proj = orc.camera(scene.camera, width, height)
    return models, view, proj


def render_oracle(orc, scene, width=None, height=None, models=None, view=None, proj=None, **kw):
    width = width or scene.size[0]
    height = height or scene.size[1]
    m, v, p = scene_matrices(orc, scene, width, height)
    return orc.render(scene, m if models is None else models, v if view is None else view,
                      p if proj is None else proj, width, height, **kw)


def render_hip(scene, width=None, height=None, models=None, view=None, proj=None, shading=0, scissor=None,
               want_clip=False, use_drawables=False, image=0):
    width = width or scene.size[0]
    height = height or scene.size[1]
    with api.HipDrawer(width, height, coverage=True, shading=shading, scissor=scissor) as d:
        if use_drawables:
            d.set_textures(scene.textures)
            d.update_drawables(scene.drawables())
        else:
            d.update_scene(scene)
        if models is not None:
            d.update_uniforms(models)
        if view is not None:
            d.update_camera_matrices(view, proj)
        else:
            d.update_camera(scene.camera)
        d.draw(image)
        d.sync()
        out = {"color": d.read_color(image), "depth": d.read_depth(image), "prim": d.read_coverage(image),
               "stats": d.stats()}
        if want_clip:
            out["clip"] = d.read_clip_positions()
        return out


def assert_frames_equal(a, b, rows=None, what=""):
    """coverage and depth bit-exact, colour bit-exact (the stated bar is +-1 LSB; the design reaches 0)."""
    sl = slice(None) if rows is None else slice(rows[0], rows[1])
    pa, pb = a["prim"][sl], b["prim"][sl]
    nbad = int((pa != pb).sum())
    assert nbad == 0, "%s coverage differs at %d pixels, first %s" % (what, nbad, np.argwhere(pa != pb)[:4].tolist())
    da, db = a["depth"][sl].view(np.uint32), b["depth"][sl].view(np.uint32)
    nbad = int((da != db).sum())
    assert nbad == 0, "%s depth bits differ at %d pixels, first %s" % (what, nbad, np.argwhere(da != db)[:4].tolist())
    ca, cb = a["color"][sl].astype(np.int16), b["color"][sl].astype(np.int16)
    diff = np.abs(ca - cb).max() if ca.size else 0
    assert diff == 0, "%s colour differs by up to %d LSB at %d pixels" % (what, diff, int((ca != cb).any(axis=-1).sum()))
