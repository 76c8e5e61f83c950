"""ctypes binding of the CPU oracle (oracle/libsxe_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (renderer_amd) never imports this module.
Parity status: "parity unpinned" (see sxe_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsxe_oracle.so")

VERTEX = np.dtype([("pos", "<f4", 3), ("color", "<f4", 3), ("uv", "<f4", 2)])
TRANSFORM = np.dtype([("position", "<f4", 3), ("scale", "<f4", 3), ("rotation", "<f4", 4)])
MESH = np.dtype([("first_vertex", "<u4"), ("n_vertices", "<u4"), ("first_index", "<u4"), ("n_indices", "<u4")])
OBJECT = np.dtype([("mesh", "<u4"), ("texture", "<i4"), ("rendered", "<u4")])

SHADE_TEXTURE, SHADE_VERTEX_COLOR, SHADE_COLOR_TIMES_TEXTURE, SHADE_LIT = 0, 1, 2, 3
# the light of SHADE_LIT (an extension without a reference counterpart): the defaults of sxe_default_light
DEFAULT_LIGHT = dict(to_light=(0.3, 0.8, 0.5), to_viewer=(0.0, 0.0, 1.0), ambient=0.2, diffuse=0.7, specular=0.3, shininess_log2=5)
FLAG_NO_TRIVIAL_REJECT = 1
FLAG_BGRA = 4  # colour target in B, G, R, A byte order
FLAG_REFERENCE_INDEXING = 2  # the draws as the reference binary records them (src/renderer/drawer.rs:313-347)


class Config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("clear", C.c_float * 4),
                ("scissor_y0", C.c_uint32), ("scissor_y1", C.c_uint32),
                ("shading", C.c_uint32), ("flags", C.c_uint32),
                ("to_light", C.c_float * 3), ("to_viewer", C.c_float * 3),
                ("ambient", C.c_float), ("diffuse", C.c_float), ("specular", C.c_float), ("shininess_log2", C.c_uint32)]


class Texture(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("prims", "prims_rejected", "prims_clipped", "subtris",
                                          "subtris_culled", "fragments_tested", "fragments_covered",
                                          "fragments_passed", "subtris_empty")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force=False):
    """Compile the oracle with gcc (seconds). Safe to call repeatedly."""
    src = os.path.join(_HERE, "sxe_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.orc_sincosf.argtypes = [C.c_float, fp, fp]
        L.orc_atan2f.argtypes = [C.c_float, C.c_float]
        L.orc_atan2f.restype = C.c_float
        L.orc_asinf.argtypes = [C.c_float]
        L.orc_asinf.restype = C.c_float
        L.orc_rand01.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_rand01.restype = C.c_float
        L.orc_srgb_encode.argtypes = [C.c_float]
        L.orc_srgb_encode.restype = C.c_uint8
        L.orc_srgb_decode.argtypes = [C.c_uint8]
        L.orc_srgb_decode.restype = C.c_float
        L.orc_rotation_system.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_float]
        L.orc_transform_rotate.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.orc_transform_translate.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.orc_perspective_yflip.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.orc_camera.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_fps_view.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_render.restype = C.c_int
        L.orc_render_mt.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def sincos(x):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincosf(C.c_float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def atan2(y, x):
    return lib().orc_atan2f(y, x)


def asin(x):
    return lib().orc_asinf(x)


def rand01(seed, frame, index):
    return lib().orc_rand01(seed, frame, index)


def srgb_encode(x):
    return int(lib().orc_srgb_encode(x))


def srgb_decode(c):
    return float(lib().orc_srgb_decode(c))


def identity_transforms(n=1):
    t = np.zeros(n, dtype=TRANSFORM)
    for i in range(n):
        lib().orc_transform_identity(_p(t[i:i + 1]))
    return t


def quat_from_euler_deg(x, y, z):
    q = np.zeros(4, dtype=np.float32)
    lib().orc_quat_from_euler_deg(C.c_float(x), C.c_float(y), C.c_float(z), _p(q))
    return q


def quat_mul(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.zeros(4, dtype=np.float32)
    lib().orc_quat_mul(_p(a), _p(b), _p(out))
    return out


def quat_to_euler(q):
    q = np.ascontiguousarray(q, dtype=np.float32)
    e = np.zeros(3, dtype=np.float32)
    lib().orc_quat_to_euler(_p(q), _p(e))
    return e


def rotate(transforms, i, x, y, z):
    lib().orc_transform_rotate(_p(transforms[i:i + 1]), x, y, z)


def translate(transforms, i, x, y, z):
    lib().orc_transform_translate(_p(transforms[i:i + 1]), x, y, z)


def rotation_system(transforms, seed, frame, delta_ms):
    assert transforms.dtype == TRANSFORM and transforms.flags.c_contiguous
    lib().orc_rotation_system(_p(transforms), len(transforms), seed, frame, delta_ms)


def to_model(transforms):
    """Transform::to_ubo for every element -> (n, 16) column-major float32."""
    assert transforms.dtype == TRANSFORM and transforms.flags.c_contiguous
    out = np.zeros((len(transforms), 16), dtype=np.float32)
    for i in range(len(transforms)):
        lib().orc_transform_to_model(_p(transforms[i:i + 1]), _p(out[i]))
    return out


def mat4_mul(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(16)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(16)
    out = np.zeros(16, dtype=np.float32)
    lib().orc_mat4_mul(_p(a), _p(b), _p(out))
    return out


def perspective(fovy_deg, aspect, near, far):
    m = np.zeros(16, dtype=np.float32)
    lib().orc_perspective_yflip(fovy_deg, aspect, near, far, _p(m))
    return m


def fps_view(eye, pitch, yaw):
    eye = np.ascontiguousarray(eye, dtype=np.float32)
    m = np.zeros(16, dtype=np.float32)
    lib().orc_fps_view(_p(eye), pitch, yaw, _p(m))
    return m


def camera(cam_transform, width, height, fovy_deg=45.0, near=0.1, far=1000.0):
    """(view, proj) as update_camera builds them; aspect is the real target W/H."""
    assert cam_transform.dtype == TRANSFORM
    view = np.zeros(16, dtype=np.float32)
    proj = np.zeros(16, dtype=np.float32)
    aspect = np.float32(width) / np.float32(height)
    lib().orc_camera(_p(cam_transform), fovy_deg, float(aspect), near, far, _p(view), _p(proj))
    return view, proj


def render(scene, models, view, proj, width, height, clear=(0.7, 0.2, 0.0, 1.0), scissor=None,
           shading=SHADE_TEXTURE, flags=0, want_clip=False, threads=0, light=None):
    """Render one frame of `scene` (renderer_amd.scenes.Scene-like: .vertices .indices .meshes
    .objects .textures). Returns dict(color, depth, prim, clip, stats)."""
    L = lib()
    y0, y1 = scissor if scissor is not None else (0, height)
    lt = dict(DEFAULT_LIGHT, **(light or {}))
    cfg = Config(width, height, (C.c_float * 4)(*clear), y0, y1, shading, flags, (C.c_float * 3)(*lt["to_light"]),
                 (C.c_float * 3)(*lt["to_viewer"]), lt["ambient"], lt["diffuse"], lt["specular"], lt["shininess_log2"])
    vertices = np.ascontiguousarray(scene.vertices, dtype=VERTEX)
    indices = np.ascontiguousarray(scene.indices, dtype=np.uint32)
    meshes = np.ascontiguousarray(scene.meshes, dtype=MESH)
    objects = np.ascontiguousarray(scene.objects, dtype=OBJECT)
    models = np.ascontiguousarray(models, dtype=np.float32).reshape(len(objects), 16)
    view = np.ascontiguousarray(view, dtype=np.float32).reshape(16)
    proj = np.ascontiguousarray(proj, dtype=np.float32).reshape(16)
    texs = [np.ascontiguousarray(t, dtype=np.uint8) for t in scene.textures]
    tarr = (Texture * max(1, len(texs)))()
    for i, t in enumerate(texs):
        assert t.ndim == 3 and t.shape[2] == 4
        tarr[i] = Texture(t.ctypes.data, t.shape[1], t.shape[0])
    color = np.zeros((height, width, 4), dtype=np.uint8)
    depth = np.zeros((height, width), dtype=np.float32)
    prim = np.zeros((height, width), dtype=np.uint32)
    n_inst_vtx = int(meshes["n_vertices"][objects["mesh"]].sum()) if len(objects) else 0
    clip = np.zeros((n_inst_vtx, 4), dtype=np.float32) if want_clip else None
    stats = Stats()
    if threads and threads > 0:
        rc = L.orc_render_mt(C.byref(cfg), _p(vertices), _p(indices), _p(meshes), C.c_uint32(len(meshes)),
                             _p(objects), C.c_uint32(len(objects)), _p(models), _p(view), _p(proj),
                             tarr, C.c_uint32(len(texs)), _p(color), _p(depth), _p(prim), C.c_int(threads))
    else:
        rc = L.orc_render(C.byref(cfg), _p(vertices), _p(indices), _p(meshes), C.c_uint32(len(meshes)),
                          _p(objects), C.c_uint32(len(objects)), _p(models), _p(view), _p(proj),
                          tarr, C.c_uint32(len(texs)), _p(color), _p(depth), _p(prim),
                          _p(clip) if clip is not None else None, C.byref(stats))
    if rc != 0:
        raise ValueError("orc_render rejected the scene (rc=%d)" % rc)
    return {"color": color, "depth": depth, "prim": prim, "clip": clip, "stats": stats.as_dict()}
