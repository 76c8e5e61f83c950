"""ctypes binding of libsxe_hip.so and a Drawer-shaped wrapper.

`HipDrawer` mirrors the reference's `Drawer` trait (src/renderer/drawer.rs:26-31):
`update_drawables`, `update_uniforms`, `update_camera`, `draw`, with the same call order as
`start_engine` (src/main.rs:165-196), plus the readback that stands in for `Presenter::present`.
There is no CPU rendering path here: without the HIP library or without a GPU every rendering
call raises `SxeError`.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# (SXE_LIB: a development build of the library to load instead, e.g. one of exp/libsxe_*.so of tools/exp.sh)
LIB_PATH = os.environ.get("SXE_LIB") or os.path.join(HERE, "libsxe_hip.so")

VERTEX = np.dtype([("pos", "<f4", 3), ("color", "<f4", 3), ("uv", "<f4", 2)])
TRANSFORM = np.dtype([("position", "<f4", 3), ("scale", "<f4", 3), ("rotation", "<f4", 4)])
MESH = np.dtype([("first_vertex", "<u4"), ("n_vertices", "<u4"), ("first_index", "<u4"), ("n_indices", "<u4")])
OBJECT = np.dtype([("mesh", "<u4"), ("texture", "<i4"), ("rendered", "<u4")])

SHADE_TEXTURE, SHADE_VERTEX_COLOR, SHADE_COLOR_TIMES_TEXTURE = 0, 1, 2
SHADE_LIT = 3  # flat Blinn-Phong over the texture: an extension, the reference's shaders have no lighting
FLAG_WRITE_COVERAGE, FLAG_TIMING = 1, 2
FLAG_DISCARD_DEPTH = 16  # the depth images are not kept (AttachmentStoreOp::DontCare, src/renderer/drawer.rs:544-547)
FLAG_BGRA = 8  # colour images in B, G, R, A byte order (src/renderer/presenter.rs:472-478 on X11)
FLAG_COMPAT_REFERENCE_INDEXING = 4  # draw list as the reference binary records it (src/renderer/drawer.rs:313-347)
E_INVALID, E_DEVICE, E_CAPACITY, E_STATE = -1, -2, -3, -4
MAX_DIM = 8192  # SXE_MAX_DIM: width and height limit of a frame

EXPORTS = [
    "sxe_default_config", "sxe_create", "sxe_destroy", "sxe_last_error", "sxe_version", "sxe_set_textures",
    "sxe_update_drawables", "sxe_update_instances", "sxe_update_uniforms", "sxe_update_transforms",
    "sxe_rotation_system", "sxe_rotation_system_span", "sxe_update_camera", "sxe_update_camera_matrices", "sxe_draw", "sxe_draw_views", "sxe_sync", "sxe_draw_noalloc",
    "sxe_read_color", "sxe_map_color", "sxe_read_depth", "sxe_read_coverage", "sxe_read_clip_positions", "sxe_read_models",
    "sxe_read_transforms", "sxe_color_device_ptr", "sxe_stream", "sxe_get_timings", "sxe_get_timing_totals", "sxe_reset_timing_totals", "sxe_set_timing_stride", "sxe_get_stats", "sxe_read_tile_counts", "sxe_read_tile_parts", "sxe_read_tile_times",
    "sxe_transform_identity", "sxe_transform_translate", "sxe_transform_rotate", "sxe_transform_to_model",
    "sxe_camera_matrices", "sxe_srgb_encode_host", "sxe_default_light", "sxe_set_light", "sxe_light_vectors",
]


def make_light(to_light=None, to_viewer=None, ambient=None, diffuse=None, specular=None, shininess_log2=None):
    """An sxe_light: the library's defaults with the given fields replaced."""
    lt = Light()
    lib().sxe_default_light(C.byref(lt))
    if to_light is not None:
        lt.to_light = (C.c_float * 3)(*to_light)
    if to_viewer is not None:
        lt.to_viewer = (C.c_float * 3)(*to_viewer)
    for name, value in (("ambient", ambient), ("diffuse", diffuse), ("specular", specular), ("shininess_log2", shininess_log2)):
        if value is not None:
            setattr(lt, name, value)
    return lt


def light_vectors(light):
    """(to_light, to_viewer, half) unit vectors of an sxe_light, as the kernels use them."""
    out = [np.zeros(3, dtype=np.float32) for _ in range(3)]
    rc = lib().sxe_light_vectors(C.byref(light), _p(out[0]), _p(out[1]), _p(out[2]))
    if rc != 0:
        raise SxeError(rc, "not a usable light")
    return out


class SxeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("sxe error %d: %s" % (code, message))
        self.code = code


class Config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("fov_deg", C.c_float), ("z_near", C.c_float),
                ("z_far", C.c_float), ("clear", C.c_float * 4), ("device", C.c_int32), ("n_images", C.c_uint32),
                ("scissor_y0", C.c_uint32), ("scissor_y1", C.c_uint32), ("shading", C.c_uint32),
                ("flags", C.c_uint32)]


class TransformC(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("scale", C.c_float * 3), ("rotation", C.c_float * 4)]


class TextureDesc(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class DrawableC(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("n_vertices", C.c_uint32), ("indices", C.c_void_p),
                ("n_indices", C.c_uint32), ("transform", TransformC), ("texture", C.c_int32),
                ("rendered", C.c_uint32)]


class MeshDesc(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("n_vertices", C.c_uint32), ("indices", C.c_void_p),
                ("n_indices", C.c_uint32)]


class InstanceC(C.Structure):
    _fields_ = [("mesh", C.c_uint32), ("texture", C.c_int32), ("rendered", C.c_uint32), ("transform", TransformC)]


INSTANCE = np.dtype([("mesh", "<u4"), ("texture", "<i4"), ("rendered", "<u4"), ("transform", TRANSFORM)])
assert INSTANCE.itemsize == C.sizeof(InstanceC) == 52


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("transform_ms", "object_ms", "setup_ms", "finish_ms",
                                         "raster_shade_ms", "frame_ms")]

    def as_dict(self):
        return {n: float(getattr(self, n)) for n, _ in self._fields_}


class Light(C.Structure):
    """sxe_light: the light of SHADE_LIT."""
    _fields_ = [("to_light", C.c_float * 3), ("to_viewer", C.c_float * 3), ("ambient", C.c_float), ("diffuse", C.c_float),
                ("specular", C.c_float), ("shininess_log2", C.c_uint32)]


class FrameStats(C.Structure):
    _fields_ = [("prims", C.c_uint64), ("records", C.c_uint64), ("bin_entries", C.c_uint64),
                ("tiles", C.c_uint64), ("instance_vertices", C.c_uint64), ("objects", C.c_uint64),
                ("tile_size", C.c_uint32), ("redraws", C.c_uint32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def lib():
    """Load libsxe_hip.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SxeError(E_DEVICE, "%s is missing: build it with `python -m renderer_amd._build` "
                                     "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.sxe_last_error.restype = C.c_char_p
        L.sxe_last_error.argtypes = [C.c_void_p]
        L.sxe_version.restype = C.c_char_p
        L.sxe_default_config.argtypes = [C.POINTER(Config), C.c_uint32, C.c_uint32]
        L.sxe_default_config.restype = None
        L.sxe_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.sxe_destroy.argtypes = [C.c_void_p]
        L.sxe_destroy.restype = None
        L.sxe_set_textures.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.sxe_update_drawables.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.sxe_default_light.argtypes = [C.POINTER(Light)]
        L.sxe_default_light.restype = None
        L.sxe_set_light.argtypes = [C.c_void_p, C.POINTER(Light)]
        L.sxe_light_vectors.argtypes = [C.POINTER(Light), C.c_void_p, C.c_void_p, C.c_void_p]
        L.sxe_update_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.sxe_update_uniforms.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.sxe_update_transforms.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.sxe_rotation_system.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_float]
        L.sxe_rotation_system_span.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float]
        L.sxe_update_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.sxe_update_camera_matrices.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.sxe_draw.argtypes = [C.c_void_p, C.c_uint32]
        L.sxe_sync.argtypes = [C.c_void_p]
        L.sxe_draw_views.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.sxe_draw_noalloc.argtypes = [C.c_void_p, C.c_uint32]
        for name in ("sxe_read_color", "sxe_read_depth", "sxe_read_coverage"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        for name in ("sxe_read_clip_positions", "sxe_read_models", "sxe_read_transforms", "sxe_read_tile_counts", "sxe_read_tile_parts",
                     "sxe_read_tile_times"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.sxe_color_device_ptr.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.sxe_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.sxe_map_color.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.POINTER(C.c_uint8))]
        L.sxe_get_timings.argtypes = [C.c_void_p, C.POINTER(Timings)]
        L.sxe_get_timing_totals.argtypes = [C.c_void_p, C.POINTER(Timings), C.POINTER(C.c_uint64)]
        L.sxe_reset_timing_totals.argtypes = [C.c_void_p]
        L.sxe_set_timing_stride.argtypes = [C.c_void_p, C.c_uint32]
        L.sxe_get_stats.argtypes = [C.c_void_p, C.POINTER(FrameStats)]
        L.sxe_transform_identity.argtypes = [C.c_void_p]
        L.sxe_transform_identity.restype = None
        L.sxe_transform_translate.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.sxe_transform_translate.restype = None
        L.sxe_transform_rotate.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.sxe_transform_rotate.restype = None
        L.sxe_transform_to_model.argtypes = [C.c_void_p, C.c_void_p]
        L.sxe_transform_to_model.restype = None
        L.sxe_camera_matrices.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.sxe_camera_matrices.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- host-side Transform maths (src/components/transform.rs), usable without a GPU ----

def identity_transforms(n=1):
    t = np.zeros(n, dtype=TRANSFORM)
    for i in range(n):
        lib().sxe_transform_identity(_p(t[i:i + 1]))
    return t


def rotate(transforms, i, x, y, z):
    """Transform::rotate on element i (Euler degrees, right-multiplied)."""
    lib().sxe_transform_rotate(_p(transforms[i:i + 1]), x, y, z)


def translate(transforms, i, x, y, z):
    lib().sxe_transform_translate(_p(transforms[i:i + 1]), x, y, z)


def to_model(transforms):
    """Transform::to_ubo for every element -> (n,16) column-major float32."""
    assert transforms.dtype == TRANSFORM and transforms.flags.c_contiguous
    out = np.zeros((len(transforms), 16), dtype=np.float32)
    for i in range(len(transforms)):
        lib().sxe_transform_to_model(_p(transforms[i:i + 1]), _p(out[i]))
    return out


def camera_matrices(cam, width, height, fov_deg=45.0, near=0.1, far=1000.0):
    assert cam.dtype == TRANSFORM
    view = np.zeros(16, dtype=np.float32)
    proj = np.zeros(16, dtype=np.float32)
    aspect = float(np.float32(width) / np.float32(height))
    lib().sxe_camera_matrices(_p(cam), fov_deg, aspect, near, far, _p(view), _p(proj))
    return view, proj


class Drawable:
    """src/primitives/drawable.rs:7-12 with the mesh inlined."""

    def __init__(self, vertices, indices, transform, texture=0, rendered=True):
        self.vertices = np.ascontiguousarray(vertices, dtype=VERTEX)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.transform = np.ascontiguousarray(transform, dtype=TRANSFORM).reshape(1)
        self.texture = int(texture)
        self.rendered = bool(rendered)


class HipDrawer:
    """The `Drawer` trait over the HIP library."""

    def __init__(self, width, height, device=0, n_images=2, scissor=None, shading=SHADE_TEXTURE,
                 coverage=False, timing=False, clear=None, fov_deg=None, z_near=None, z_far=None, debug_flags=0,
                 reference_indexing=False, bgra=False, discard_depth=False):
        self._L = lib()
        self._ctx = C.c_void_p()
        cfg = Config()
        self._L.sxe_default_config(C.byref(cfg), width, height)
        cfg.device = device
        cfg.n_images = n_images
        if scissor is not None:
            cfg.scissor_y0, cfg.scissor_y1 = int(scissor[0]), int(scissor[1])
            if cfg.scissor_y0 == 0 and cfg.scissor_y1 == 0:
                raise SxeError(E_INVALID, "empty scissor")
        cfg.shading = shading
        cfg.flags = ((FLAG_WRITE_COVERAGE if coverage else 0) | (FLAG_TIMING if timing else 0) | debug_flags |
                     (FLAG_COMPAT_REFERENCE_INDEXING if reference_indexing else 0) | (FLAG_BGRA if bgra else 0) |
                     (FLAG_DISCARD_DEPTH if discard_depth else 0))
        if debug_flags:
            # development switches (tools/ only): the library refuses them without this
            os.environ.setdefault("SXE_DEVELOPMENT", "1")
        if clear is not None:
            cfg.clear = (C.c_float * 4)(*clear)
        if fov_deg is not None:
            cfg.fov_deg = fov_deg
        if z_near is not None:
            cfg.z_near = z_near
        if z_far is not None:
            cfg.z_far = z_far
        rc = self._L.sxe_create(C.byref(cfg), C.byref(self._ctx))
        if rc != 0:
            raise SxeError(rc, self._L.sxe_last_error(None).decode())
        self.width, self.height, self.cfg = width, height, cfg
        self.n_objects = 0
        self.n_instance_vertices = 0

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._L.sxe_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise SxeError(rc, self._L.sxe_last_error(self._ctx).decode())

    # ---- scene ----
    def set_textures(self, textures):
        texs = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
        arr = (TextureDesc * max(1, len(texs)))()
        for i, t in enumerate(texs):
            if t.ndim != 3 or t.shape[2] != 4:
                raise SxeError(E_INVALID, "texture %d is not HxWx4 uint8" % i)
            arr[i] = TextureDesc(t.ctypes.data, t.shape[1], t.shape[0])
        self._check(self._L.sxe_set_textures(self._ctx, arr, len(texs)))

    def set_light(self, to_light=None, to_viewer=None, ambient=None, diffuse=None, specular=None, shininess_log2=None):
        """The light of SHADE_LIT; what is left out keeps the library's default."""
        self._check(self._L.sxe_set_light(self._ctx, C.byref(make_light(to_light, to_viewer, ambient, diffuse, specular, shininess_log2))))

    def update_drawables(self, drawables):
        arr = (DrawableC * max(1, len(drawables)))()
        nv = 0
        for i, d in enumerate(drawables):
            tr = TransformC.from_buffer_copy(d.transform.tobytes())
            arr[i] = DrawableC(d.vertices.ctypes.data, len(d.vertices), d.indices.ctypes.data, len(d.indices), tr,
                               d.texture, 1 if d.rendered else 0)
            nv += len(d.vertices)
        self._check(self._L.sxe_update_drawables(self._ctx, arr, len(drawables)))
        self.n_objects = len(drawables)
        self.n_instance_vertices = nv

    def update_scene(self, scene, transforms=None):
        """Upload a scenes.Scene (shared meshes + instances)."""
        vertices = np.ascontiguousarray(scene.vertices, dtype=VERTEX)
        indices = np.ascontiguousarray(scene.indices, dtype=np.uint32)
        meshes = (MeshDesc * max(1, len(scene.meshes)))()
        for i, m in enumerate(scene.meshes):
            meshes[i] = MeshDesc(vertices.ctypes.data + VERTEX.itemsize * int(m["first_vertex"]), int(m["n_vertices"]),
                                 indices.ctypes.data + 4 * int(m["first_index"]), int(m["n_indices"]))
        tr = scene.transforms if transforms is None else transforms
        inst = np.zeros(len(scene.objects), dtype=INSTANCE)
        inst["mesh"] = scene.objects["mesh"]
        inst["texture"] = scene.objects["texture"]
        inst["rendered"] = scene.objects["rendered"]
        inst["transform"] = tr
        self.set_textures(scene.textures)
        self._check(self._L.sxe_update_instances(self._ctx, meshes, len(scene.meshes), _p(inst), len(inst)))
        self.n_objects = len(inst)
        self.n_instance_vertices = int(scene.meshes["n_vertices"][scene.objects["mesh"]].sum()) if len(inst) else 0

    # ---- per frame ----
    def update_uniforms(self, models):
        m = np.ascontiguousarray(models, dtype=np.float32).reshape(-1, 16)
        self._check(self._L.sxe_update_uniforms(self._ctx, _p(m), len(m)))

    def update_transforms(self, transforms):
        t = np.ascontiguousarray(transforms, dtype=TRANSFORM)
        self._check(self._L.sxe_update_transforms(self._ctx, _p(t), len(t)))

    def rotation_system(self, seed, frame, delta_ms, n_frames=1):
        """The rotation update of frames frame .. frame + n_frames - 1, in one launch."""
        if n_frames == 1:
            self._check(self._L.sxe_rotation_system(self._ctx, seed, frame, delta_ms))
        else:
            self._check(self._L.sxe_rotation_system_span(self._ctx, seed, frame, n_frames, delta_ms))

    def update_camera(self, transform):
        t = np.ascontiguousarray(transform, dtype=TRANSFORM).reshape(1)
        self._check(self._L.sxe_update_camera(self._ctx, _p(t)))

    def update_camera_matrices(self, view, proj):
        v = np.ascontiguousarray(view, dtype=np.float32).reshape(16)
        p = np.ascontiguousarray(proj, dtype=np.float32).reshape(16)
        self._check(self._L.sxe_update_camera_matrices(self._ctx, _p(v), _p(p)))

    def draw(self, image_index=0):
        self._check(self._L.sxe_draw(self._ctx, image_index))

    def draw_views(self, cameras, first_image=0):
        """One view per camera Transform into images first_image, first_image + 1, ...; the views
        share their kernel launches."""
        cams = np.ascontiguousarray(cameras, dtype=TRANSFORM)
        self._check(self._L.sxe_draw_views(self._ctx, _p(cams), len(cams), first_image))

    def sync(self):
        self._check(self._L.sxe_sync(self._ctx))

    def draw_noalloc(self, image_index=0):
        """draw + sync with every work buffer in the library's static device array (no allocation)."""
        self._check(self._L.sxe_draw_noalloc(self._ctx, image_index))

    # ---- readback ----
    def read_color(self, image_index=0):
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._check(self._L.sxe_read_color(self._ctx, image_index, _p(out), out.nbytes))
        return out

    def read_depth(self, image_index=0):
        out = np.zeros((self.height, self.width), dtype=np.float32)
        self._check(self._L.sxe_read_depth(self._ctx, image_index, _p(out), out.nbytes))
        return out

    def read_coverage(self, image_index=0):
        out = np.zeros((self.height, self.width), dtype=np.uint32)
        self._check(self._L.sxe_read_coverage(self._ctx, image_index, _p(out), out.nbytes))
        return out

    def read_clip_positions(self):
        out = np.zeros((max(self.n_instance_vertices, 1), 4), dtype=np.float32)
        self._check(self._L.sxe_read_clip_positions(self._ctx, _p(out), out.nbytes))
        return out[:self.n_instance_vertices]

    def read_models(self):
        out = np.zeros((max(self.n_objects, 1), 16), dtype=np.float32)
        self._check(self._L.sxe_read_models(self._ctx, _p(out), out.nbytes))
        return out[:self.n_objects]

    def read_transforms(self):
        out = np.zeros(max(self.n_objects, 1), dtype=TRANSFORM)
        self._check(self._L.sxe_read_transforms(self._ctx, _p(out), out.nbytes))
        return out[:self.n_objects]

    def color_device_ptr(self, image_index=0):
        p = C.c_void_p()
        self._check(self._L.sxe_color_device_ptr(self._ctx, image_index, C.byref(p)))
        return p.value

    def stream_ptr(self):
        """The context's hipStream_t as an integer (for torch.cuda.ExternalStream)."""
        p = C.c_void_p()
        self._check(self._L.sxe_stream(self._ctx, C.byref(p)))
        return p.value or 0

    def tile_counts(self):
        n = int(self.stats()["tiles"])
        out = np.zeros(max(n, 1), dtype=np.uint32)
        self._check(self._L.sxe_read_tile_counts(self._ctx, _p(out), out.nbytes))
        return out[:n]

    def map_color(self, image=0):
        """The frame as a read-only view of the library's pinned copy (no copy into numpy memory);
        valid until the next map_color of this image or close()."""
        ptr = C.POINTER(C.c_uint8)()
        self._check(self._L.sxe_map_color(self._ctx, image, C.byref(ptr)))
        a = np.ctypeslib.as_array(ptr, shape=(self.height, self.width, 4))
        a.flags.writeable = False
        return a

    def tile_parts(self):
        n = int(self.stats()["tiles"])
        out = np.zeros((max(n, 1), 5), dtype=np.uint32)
        self._check(self._L.sxe_read_tile_parts(self._ctx, _p(out), out.nbytes))
        return out[:n]

    def tile_times(self):
        n = int(self.stats()["tiles"])
        out = np.zeros((max(n, 1), 8), dtype=np.uint64)
        self._check(self._L.sxe_read_tile_times(self._ctx, _p(out), out.nbytes))
        return out[:n]

    def timings(self):
        t = Timings()
        self._check(self._L.sxe_get_timings(self._ctx, C.byref(t)))
        return t.as_dict()

    def timing_totals(self):
        """(sums of the per-kernel times, number of frames) since the last reset_timing_totals()."""
        t, n = Timings(), C.c_uint64(0)
        self._check(self._L.sxe_get_timing_totals(self._ctx, C.byref(t), C.byref(n)))
        return t.as_dict(), int(n.value)

    def reset_timing_totals(self):
        self._check(self._L.sxe_reset_timing_totals(self._ctx))

    def set_timing_stride(self, every):
        """Bracket only every `every`-th frame by events (each event costs stream time)."""
        self._check(self._L.sxe_set_timing_stride(self._ctx, int(every)))

    def stats(self):
        s = FrameStats()
        self._check(self._L.sxe_get_stats(self._ctx, C.byref(s)))
        return s.as_dict()


class FramesInFlight:
    """One context per target image: the reference keeps a command buffer and a fence per swapchain
    image and submits frame i + 1 while image i is still being rendered
    (src/renderer/drawer.rs:412-456).  Here every image's context has its own HIP stream and work
    buffers, so the frames of different images may overlap on the GPU (config 3 gains 5 % in frame
    rate that way; a frame dominated by its tile kernel gains nothing).  What the
    caller sees is the Drawer of one context: state calls go to every context, draw(image_index) to
    the context of that image, and the rotation system's updates are applied by each context when it
    next draws, as one span over the frames it did not draw (the replicated state of frame-parallel
    ranks).  Every frame is the frame a single context draws, bit for bit."""

    def __init__(self, width, height, n_images=2, **kw):
        if n_images < 1:
            raise SxeError(E_INVALID, "FramesInFlight needs at least one image")
        self.width, self.height = width, height
        self.contexts = []
        try:
            for _ in range(n_images):
                self.contexts.append(HipDrawer(width, height, n_images=1, **kw))
        except Exception:
            self.close()
            raise
        self._waiting = [[] for _ in range(n_images)]  # rotation updates (seed, first frame, frames, delta) not applied yet

    def close(self):
        for d in self.contexts:
            d.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _all(self, name, *a, **kw):
        for d in self.contexts:
            getattr(d, name)(*a, **kw)

    def _context(self, image_index):
        if not 0 <= image_index < len(self.contexts):
            raise SxeError(E_INVALID, "image index %d of %d" % (image_index, len(self.contexts)))
        return image_index

    def _apply_waiting(self, k):
        """Hand context k the updates it has missed: runs of consecutive frames as one span each."""
        runs = []
        for seed, first, n, delta in self._waiting[k]:
            if runs and runs[-1][0] == seed and runs[-1][3] == delta and runs[-1][1] + runs[-1][2] == first:
                runs[-1][2] += n
            else:
                runs.append([seed, first, n, delta])
        self._waiting[k] = []
        for seed, first, n, delta in runs:
            self.contexts[k].rotation_system(seed, first, delta, n_frames=n)

    # ---- scene and per-frame state: every context ----
    def set_textures(self, textures):
        self._all("set_textures", textures)

    def set_light(self, **kw):
        self._all("set_light", **kw)

    def update_drawables(self, drawables):
        self._forget_updates()
        self._all("update_drawables", drawables)

    def update_scene(self, scene, transforms=None):
        self._forget_updates()
        self._all("update_scene", scene, transforms)

    def update_uniforms(self, models):
        self._forget_updates()
        self._all("update_uniforms", models)

    def update_transforms(self, transforms):
        self._forget_updates()
        self._all("update_transforms", transforms)

    def _forget_updates(self):
        # (what replaces every transform makes the rotations still waiting for it void)
        self._waiting = [[] for _ in self.contexts]

    def update_camera(self, transform):
        self._all("update_camera", transform)

    def update_camera_matrices(self, view, proj):
        self._all("update_camera_matrices", view, proj)

    def rotation_system(self, seed, frame, delta_ms, n_frames=1):
        for w in self._waiting:
            w.append((seed, frame, n_frames, delta_ms))

    # ---- frames ----
    def draw(self, image_index=0):
        k = self._context(image_index)
        self._apply_waiting(k)
        self.contexts[k].draw(0)

    def sync(self):
        self._all("sync")

    def read_color(self, image_index=0):
        return self.contexts[self._context(image_index)].read_color(0)

    def read_depth(self, image_index=0):
        return self.contexts[self._context(image_index)].read_depth(0)

    def read_coverage(self, image_index=0):
        return self.contexts[self._context(image_index)].read_coverage(0)

    def read_transforms(self):
        self._apply_waiting(0)
        return self.contexts[0].read_transforms()

    def set_timing_stride(self, every):
        self._all("set_timing_stride", every)

    def stats(self):
        """Frame statistics of the first context, with the redraws of all of them."""
        s = self.contexts[0].stats()
        s["redraws"] = sum(d.stats()["redraws"] for d in self.contexts)
        return s
