"""ctypes binding of the C99 host library (host/libpt_host.so, declared in host/pt_host.h).

The host library is the counterpart of the reference's C host: it reads .vks/.vkt/.lights/.rt_save files,
computes the camera matrices and the 784-byte constants block and drives the HIP library. Python callers
(bench.py, tests) use it exactly like the C driver host/main.c does.
"""
import ctypes as C
import os

import numpy as np

from . import hip_api

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "host", "libpt_host.so")

EXPORTED_SYMBOLS = [
    "mat_mat_mul", "normalize", "rotation_matrix_from_angles", "invert_mat4",
    "get_world_to_view_space", "get_view_to_projection_space", "get_world_to_projection_space",
    "load_scene", "free_scene", "load_textures", "free_textures",
    "get_scene_file", "quicksave", "quickload", "init_scene_spec", "init_render_settings",
    "load_scene_devices", "create_lit_scene_from_paths_devices", "create_lit_scene_devices", "reduce_render_targets",
    "create_lit_scene_from_paths", "create_lit_scene", "free_lit_scene", "apply_render_settings", "set_light_sampling",
    "create_render_targets", "write_constant_buffer", "render_frames", "save_screenshot", "write_hdr", "write_png",
]


class Rotation(C.Structure):
    _fields_ = [("angles", C.c_float * 3), ("mouse_active", C.c_bool), ("origin_angles", C.c_float * 3)]


class Camera(C.Structure):
    """camera_t, src/camera.h:42-63"""
    _fields_ = [("rotation", Rotation), ("position", C.c_float * 3), ("speed", C.c_float), ("near", C.c_float), ("far", C.c_float),
                ("type", C.c_int32), ("fov", C.c_float), ("height", C.c_float)]


class SceneSpec(C.Structure):
    """scene_spec_t, src/main.h:55-78"""
    _fields_ = [("scene_file", C.c_int32), ("camera", Camera), ("tonemapper", C.c_int32), ("exposure", C.c_float),
                ("frame_index", C.c_uint32), ("sky_color", C.c_float * 3), ("sky_strength", C.c_float),
                ("emission_material_color", C.c_float * 3), ("emission_material_strength", C.c_float), ("params", C.c_float * 4)]


assert C.sizeof(Camera) == 64 and C.sizeof(SceneSpec) == 128


class SceneFileHeader(C.Structure):
    _fields_ = [("marker", C.c_uint32), ("version", C.c_uint32), ("material_count", C.c_uint64), ("triangle_count", C.c_uint64),
                ("dequantization_factor", C.c_float * 3), ("dequantization_summand", C.c_float * 3), ("material_names", C.POINTER(C.c_char_p))]


class HostScene(C.Structure):
    _fields_ = [("header", SceneFileHeader), ("device_scene", C.c_void_p), ("device_context", C.c_void_p), ("emission_material_index", C.c_uint32)]


class LitScene(C.Structure):
    _fields_ = [("scene", HostScene), ("spherical_light_count", C.c_uint32), ("spherical_lights", C.POINTER(C.c_float * 4))]


class RenderTargets(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("accum_frame_count", C.c_uint32)]


class RenderSettings(C.Structure):
    _fields_ = [("sampling_strategy", C.c_int32), ("path_length", C.c_uint32)]


class HostTexture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("mipmap_count", C.c_uint32),
                ("mipmap_sizes", C.POINTER(C.c_uint64)), ("mipmap_data", C.POINTER(C.c_void_p)), ("payload", C.c_void_p)]


class HostTextures(C.Structure):
    _fields_ = [("texture_count", C.c_uint32), ("textures", C.POINTER(HostTexture))]


IMAGE_FILE_FORMAT_PNG, IMAGE_FILE_FORMAT_JPG, IMAGE_FILE_FORMAT_HDR = range(3)
TONEMAPPER_CLAMP, TONEMAPPER_ACES, TONEMAPPER_KHRONOS_PBR_NEUTRAL = range(3)
MAX_SLIDE_COUNT = 100


class Slide(C.Structure):
    _fields_ = [("quicksave", C.c_char_p), ("render_settings", RenderSettings), ("screenshot_path", C.c_char_p),
                ("screenshot_format", C.c_int), ("screenshot_frame", C.c_uint32)]


class Slideshow(C.Structure):
    _fields_ = [("slides", Slide * MAX_SLIDE_COUNT), ("slide_count", C.c_uint32), ("slide_begin", C.c_uint32), ("slide_end", C.c_uint32),
                ("slide_current", C.c_uint32)]

_lib = None


def lib():
    global _lib
    if _lib is None:
        hip_api.lib()  # libpt_host.so links against libpt_hip.so; load it first by absolute path
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing; build it with `make -C path_tracer_amd/host`" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        vp, u32, i32, f32p = C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_float)
        _lib.invert_mat4.argtypes = [vp, vp]
        _lib.invert_mat4.restype = None
        _lib.rotation_matrix_from_angles.argtypes = [vp, vp]
        _lib.rotation_matrix_from_angles.restype = None
        _lib.get_world_to_projection_space.argtypes = [vp, C.POINTER(Camera), C.c_float]
        _lib.get_world_to_projection_space.restype = None
        _lib.get_world_to_view_space.argtypes = [vp, C.POINTER(Camera)]
        _lib.get_world_to_view_space.restype = None
        _lib.load_scene.argtypes = [C.POINTER(HostScene), i32, C.c_char_p, C.c_char_p]
        _lib.free_scene.argtypes = [C.POINTER(HostScene)]
        _lib.free_scene.restype = None
        _lib.load_textures.argtypes = [C.POINTER(HostTextures), C.POINTER(C.c_char_p), u32]
        _lib.free_textures.argtypes = [C.POINTER(HostTextures)]
        _lib.free_textures.restype = None
        _lib.get_scene_file.argtypes = [i32] + [C.POINTER(C.c_char_p)] * 5
        _lib.quickload.argtypes = [C.POINTER(SceneSpec), C.c_char_p]
        _lib.quicksave.argtypes = [C.POINTER(SceneSpec), C.c_char_p]
        _lib.init_scene_spec.argtypes = [C.POINTER(SceneSpec)]
        _lib.init_scene_spec.restype = None
        _lib.init_render_settings.argtypes = [C.POINTER(RenderSettings)]
        _lib.init_render_settings.restype = None
        _lib.create_lit_scene_from_paths.argtypes = [C.POINTER(LitScene), i32, C.c_char_p, C.c_char_p, C.c_char_p]
        _lib.create_lit_scene.argtypes = [C.POINTER(LitScene), i32, C.POINTER(SceneSpec), C.c_char_p]
        _lib.create_lit_scene_devices.argtypes = [C.POINTER(LitScene), C.POINTER(C.c_int), u32, C.POINTER(SceneSpec), C.c_char_p]
        _lib.create_lit_scene_from_paths_devices.argtypes = [C.POINTER(LitScene), C.POINTER(C.c_int), u32, C.c_char_p, C.c_char_p, C.c_char_p]
        _lib.reduce_render_targets.argtypes = [C.POINTER(LitScene)]
        _lib.free_lit_scene.argtypes = [C.POINTER(LitScene)]
        _lib.free_lit_scene.restype = None
        _lib.apply_render_settings.argtypes = [C.POINTER(LitScene), C.POINTER(RenderSettings)]
        _lib.set_light_sampling.argtypes = [C.POINTER(LitScene), C.c_int]
        _lib.create_render_targets.argtypes = [C.POINTER(RenderTargets), C.POINTER(LitScene), u32, u32]
        _lib.write_constant_buffer.argtypes = [C.POINTER(hip_api.Constants), C.POINTER(SceneSpec), C.POINTER(LitScene), C.POINTER(RenderTargets)]
        _lib.render_frames.argtypes = [C.POINTER(SceneSpec), C.POINTER(LitScene), C.POINTER(RenderTargets), u32]
        _lib.save_screenshot.argtypes = [C.c_char_p, i32, C.POINTER(LitScene), C.POINTER(RenderTargets), C.POINTER(SceneSpec)]
        _lib.write_hdr.argtypes = [C.c_char_p, u32, u32, vp]
        _lib.write_png.argtypes = [C.c_char_p, u32, u32, vp]
        _lib.write_jpg.argtypes = [C.c_char_p, u32, u32, vp, i32]
        _lib.tonemap_pixel.argtypes = [f32p, f32p, C.c_float, u32, i32]
        _lib.tonemap_pixel.restype = None
        _lib.tonemap_image.argtypes = [vp, vp, C.c_size_t, C.c_float, u32, i32]
        _lib.tonemap_image.restype = None
        _lib.save_display_image.argtypes = [C.c_char_p, i32, C.POINTER(LitScene), C.POINTER(RenderTargets), C.POINTER(SceneSpec)]
        _lib.create_slides.argtypes = [C.POINTER(Slideshow)]
        _lib.create_slides.restype = u32
        _lib.free_slideshow.argtypes = [C.POINTER(Slideshow)]
        _lib.free_slideshow.restype = None
        _lib.run_slideshow.argtypes = [C.POINTER(Slideshow), i32, C.c_char_p, C.c_char_p, u32, u32, i32, u32]
    return _lib


class HostError(RuntimeError):
    pass


def _check(code, what):
    if code != 0:
        raise HostError("%s failed (code %d)" % (what, code))


def quickload(path) -> SceneSpec:
    spec = SceneSpec()
    _check(lib().quickload(C.byref(spec), os.fsencode(path)), "quickload")
    return spec


def scene_file_paths(scene_file):
    outs = [C.c_char_p() for _ in range(5)]
    _check(lib().get_scene_file(scene_file, *[C.byref(o) for o in outs]), "get_scene_file")
    keys = ("name", "scene", "textures", "lights", "quicksave")
    return {k: o.value.decode() for k, o in zip(keys, outs)}


class Renderer:
    """The host objects of one GPU: lit scene, render settings, render targets (what app_t holds in the reference)."""

    def __init__(self, spec: SceneSpec, data_root, device=0, width=1280, height=1024, sampling_strategy=hip_api.SAMPLING_STRATEGY_NEE, path_length=4, devices=None):
        """devices: a list of GPU indices makes the C host replicate the scene on all of them (a pt_context_t); frames
        are then shared out among them and summed over xGMI before the accumulation is read."""
        self.spec = SceneSpec.from_buffer_copy(bytes(spec))
        self.lit_scene = LitScene()
        self.targets = RenderTargets()
        self.settings = RenderSettings(sampling_strategy, path_length)
        self._alive = False
        if devices is None:
            _check(lib().create_lit_scene(C.byref(self.lit_scene), device, C.byref(self.spec), os.fsencode(data_root)), "create_lit_scene")
        else:
            ids = (C.c_int * len(devices))(*devices)
            _check(lib().create_lit_scene_devices(C.byref(self.lit_scene), ids, len(devices), C.byref(self.spec), os.fsencode(data_root)), "create_lit_scene_devices")
        self._alive = True
        _check(lib().apply_render_settings(C.byref(self.lit_scene), C.byref(self.settings)), "apply_render_settings")
        _check(lib().create_render_targets(C.byref(self.targets), C.byref(self.lit_scene), width, height), "create_render_targets")
        self.width, self.height = width, height

    @property
    def device_scene(self):
        return C.c_void_p(self.lit_scene.scene.device_scene)

    def set_render_settings(self, sampling_strategy, path_length):
        self.settings = RenderSettings(sampling_strategy, path_length)
        _check(lib().apply_render_settings(C.byref(self.lit_scene), C.byref(self.settings)), "apply_render_settings")

    def set_light_sampling(self, light_sampling):
        _check(lib().set_light_sampling(C.byref(self.lit_scene), light_sampling), "set_light_sampling")

    def resize(self, width, height):
        _check(lib().create_render_targets(C.byref(self.targets), C.byref(self.lit_scene), width, height), "create_render_targets")
        self.width, self.height = width, height

    def constants(self) -> hip_api.Constants:
        constants = hip_api.Constants()
        _check(lib().write_constant_buffer(C.byref(constants), C.byref(self.spec), C.byref(self.lit_scene), C.byref(self.targets)), "write_constant_buffer")
        return constants

    def render_frames(self, frame_count):
        _check(lib().render_frames(C.byref(self.spec), C.byref(self.lit_scene), C.byref(self.targets), frame_count), "render_frames")

    @property
    def device_context(self):
        return C.c_void_p(self.lit_scene.scene.device_context) if self.lit_scene.scene.device_context else None

    def synchronize(self):
        if self.device_context:
            _check(hip_api.lib().pt_context_synchronize(self.device_context), "pt_context_synchronize")
        else:
            _check(hip_api.lib().pt_synchronize(self.device_scene), "pt_synchronize")

    def reduce(self):
        """Sums the targets of all GPUs into the first one's (nothing to do on one GPU)."""
        _check(lib().reduce_render_targets(C.byref(self.lit_scene)), "reduce_render_targets")

    def read_accumulation(self):
        self.reduce()
        rgba = np.zeros((self.height, self.width, 4), np.float32)
        count = C.c_uint32()
        _check(hip_api.lib().pt_read_accumulation(self.device_scene, rgba.ctypes.data_as(C.c_void_p), C.byref(count)), "pt_read_accumulation")
        return rgba, count.value

    def save_screenshot(self, path, image_format=None):
        if image_format is None:
            image_format = IMAGE_FILE_FORMAT_HDR if path.endswith(".hdr") else (IMAGE_FILE_FORMAT_JPG if path.endswith(".jpg") else IMAGE_FILE_FORMAT_PNG)
        _check(lib().save_screenshot(os.fsencode(path), image_format, C.byref(self.lit_scene), C.byref(self.targets), C.byref(self.spec)), "save_screenshot")

    def save_display_image(self, path, image_format=None):
        if image_format is None:
            image_format = IMAGE_FILE_FORMAT_JPG if path.endswith(".jpg") else IMAGE_FILE_FORMAT_PNG
        _check(lib().save_display_image(os.fsencode(path), image_format, C.byref(self.lit_scene), C.byref(self.targets), C.byref(self.spec)), "save_display_image")

    def statistics(self):
        stats = hip_api.Statistics()
        _check(hip_api.lib().pt_get_statistics(self.device_scene, C.byref(stats)), "pt_get_statistics")
        return stats

    def reset_statistics(self):
        _check(hip_api.lib().pt_reset_statistics(self.device_scene), "pt_reset_statistics")

    def set_instrumentation(self, kernel_timing=False, visit_counters=False):
        _check(hip_api.lib().pt_set_instrumentation(self.device_scene, int(kernel_timing), int(visit_counters)), "pt_set_instrumentation")

    def bvh_info(self):
        info = hip_api.BvhInfo()
        _check(hip_api.lib().pt_get_bvh_info(self.device_scene, C.byref(info)), "pt_get_bvh_info")
        return info

    def accumulation_device_pointer(self):
        return hip_api.lib().pt_get_accumulation_device_pointer(self.device_scene)

    def close(self):
        if self._alive:
            lib().free_lit_scene(C.byref(self.lit_scene))
            self._alive = False

    def __del__(self):
        self.close()


class ReplicaView:
    """Statistics and instrumentation of one replica of a multi-GPU renderer (pt_context_scene(context, i)); the
    replica belongs to the renderer's context."""

    def __init__(self, renderer, scene_handle):
        self._renderer = renderer  # keeps the context alive
        self._scene = C.c_void_p(scene_handle)

    device_scene = property(lambda self: self._scene)
    statistics = Renderer.statistics
    reset_statistics = Renderer.reset_statistics
    set_instrumentation = Renderer.set_instrumentation
    bvh_info = Renderer.bvh_info
