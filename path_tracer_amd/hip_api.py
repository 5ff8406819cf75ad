"""ctypes binding of the C ABI declared in include/pt_hip.h (libpt_hip.so).

This is plumbing for Python callers (tests, bench.py): every call goes straight to the HIP library.
There is no CPU fallback; if the library cannot be loaded the import of this module's `lib()` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpt_hip.so")

SAMPLING_STRATEGY_SPHERICAL, SAMPLING_STRATEGY_PSA, SAMPLING_STRATEGY_BRDF, SAMPLING_STRATEGY_NEE = range(4)
LIGHT_SAMPLING_EXACT, LIGHT_SAMPLING_TREE = range(2)
SHADING_MATH_EXACT, SHADING_MATH_FAST = range(2)

EXPORTED_SYMBOLS = [
    "pt_create_scene", "pt_destroy_scene", "pt_set_lights", "pt_set_light_sampling", "pt_set_render_settings", "pt_create_render_targets",
    "pt_render_samples", "pt_synchronize", "pt_read_accumulation", "pt_get_accumulation_device_pointer",
    "pt_set_sample_count", "pt_set_instrumentation", "pt_get_statistics", "pt_reset_statistics", "pt_get_bvh_info",
    "pt_create_context", "pt_destroy_context", "pt_context_device_count", "pt_context_scene", "pt_context_set_lights",
    "pt_context_set_light_sampling", "pt_context_set_render_settings", "pt_context_create_render_targets",
    "pt_context_render_samples", "pt_context_synchronize", "pt_context_reduce", "pt_context_last_reduce_ms",
    "pt_trace_primary_rays", "pt_trace_rays", "pt_trace_rays_no_alloc", "pt_evaluate_shading_data", "pt_random_numbers",
    "pt_abi_version", "pt_get_device_count", "pt_context_frame_slice", "pt_evaluate_function",
    "pt_set_shading_math", "pt_context_set_shading_math",
]
ABI_VERSION = 3  # PT_HIP_ABI_VERSION of the include/pt_hip.h that the structures below mirror


class Constants(C.Structure):
    """pt_constants_t = constants_t of the reference (src/main.h:237-256)."""
    _fields_ = [
        ("world_to_projection_space", C.c_float * 16),
        ("projection_to_world_space", C.c_float * 16),
        ("camera_pos", C.c_float * 3), ("camera_type", C.c_int32),
        ("hemispherical_camera_normal", C.c_float * 3), ("pad_1", C.c_float),
        ("dequantization_factor", C.c_float * 3), ("pad_2", C.c_float),
        ("dequantization_summand", C.c_float * 3), ("pad_3", C.c_float),
        ("viewport_size", C.c_float * 2), ("inv_viewport_size", C.c_float * 2),
        ("exposure", C.c_float), ("frame_index", C.c_uint32), ("accum_frame_count", C.c_uint32), ("pad_4", C.c_float),
        ("sky_radiance", C.c_float * 3), ("pad_5", C.c_float),
        ("emission_material_radiance", C.c_float * 3), ("pad_6", C.c_float),
        ("params", C.c_float * 4),
        ("spherical_lights", (C.c_float * 4) * 32),
    ]


assert C.sizeof(Constants) == 784


class SceneHeader(C.Structure):
    _fields_ = [("material_count", C.c_uint64), ("triangle_count", C.c_uint64),
                ("dequantization_factor", C.c_float * 3), ("dequantization_summand", C.c_float * 3)]


class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("mip_count", C.c_uint32),
                ("mip_data", C.POINTER(C.c_void_p)), ("mip_sizes", C.POINTER(C.c_uint64))]


class Statistics(C.Structure):
    _fields_ = [("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("paths", C.c_uint64),
                ("nodes_visited", C.c_uint64 * 2), ("triangles_tested", C.c_uint64 * 2), ("shaded_hits", C.c_uint64),
                ("kernel_ms", C.c_double * 5), ("kernel_launches", C.c_uint64 * 5),
                ("shadow_skipping", C.c_uint32), ("shadow_skipping_fixed", C.c_uint32),
                ("shadow_rays_decided_on", C.c_uint64), ("shadow_rays_blocked", C.c_uint64)]


class BvhInfo(C.Structure):
    _fields_ = [("node_count", C.c_uint32), ("triangle_count", C.c_uint32), ("depth", C.c_uint32),
                ("node_bytes", C.c_uint32), ("triangle_bytes", C.c_uint32), ("build_ms", C.c_float)]


KERNEL_NAMES = ("raygen", "trace_extension", "trace_shadow", "shade", "resolve")

_lib = None


def lib():
    """Loads libpt_hip.so. Raises if it is missing: there is no other implementation to fall back to."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing; build it with `make -C path_tracer_amd/csrc` (or __graft_entry__.build())" % LIB_PATH)
        loaded = C.CDLL(LIB_PATH)
        loaded.pt_abi_version.restype = C.c_uint32
        if loaded.pt_abi_version() != ABI_VERSION:
            raise RuntimeError("%s has ABI version %d, this binding was written for %d; rebuild it with `make -C path_tracer_amd/csrc`" % (LIB_PATH, loaded.pt_abi_version(), ABI_VERSION))
        _lib = loaded
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
        _lib.pt_create_scene.argtypes = [C.POINTER(vp), i32, C.POINTER(SceneHeader), vp, vp, vp, C.POINTER(TextureDesc), u32]
        _lib.pt_destroy_scene.argtypes = [vp]
        _lib.pt_destroy_scene.restype = None
        _lib.pt_set_lights.argtypes = [vp, vp, u32]
        _lib.pt_set_light_sampling.argtypes = [vp, i32]
        _lib.pt_set_render_settings.argtypes = [vp, i32, u32, u32]
        _lib.pt_set_shading_math.argtypes = [vp, i32]
        _lib.pt_context_set_shading_math.argtypes = [vp, i32]
        _lib.pt_create_render_targets.argtypes = [vp, u32, u32]
        _lib.pt_render_samples.argtypes = [vp, C.POINTER(Constants), u32, u32, u32, i32]
        _lib.pt_synchronize.argtypes = [vp]
        _lib.pt_read_accumulation.argtypes = [vp, vp, C.POINTER(u32)]
        _lib.pt_get_accumulation_device_pointer.argtypes = [vp]
        _lib.pt_get_accumulation_device_pointer.restype = vp
        _lib.pt_set_sample_count.argtypes = [vp, u32]
        _lib.pt_set_instrumentation.argtypes = [vp, i32, i32]
        _lib.pt_get_statistics.argtypes = [vp, C.POINTER(Statistics)]
        _lib.pt_reset_statistics.argtypes = [vp]
        _lib.pt_get_bvh_info.argtypes = [vp, C.POINTER(BvhInfo)]
        _lib.pt_trace_primary_rays.argtypes = [vp, C.POINTER(Constants), u32, vp, vp, vp]
        _lib.pt_trace_rays.argtypes = [vp, vp, vp, u32, vp, vp, vp]
        _lib.pt_trace_rays_no_alloc.argtypes = [vp, vp, vp, u32, vp, vp, vp]
        _lib.pt_evaluate_shading_data.argtypes = [vp, C.POINTER(Constants), vp, vp, vp, vp, vp, u32, vp]
        _lib.pt_random_numbers.argtypes = [i32, u32, u32, u32, u32, vp]
        _lib.pt_evaluate_function.argtypes = [vp, i32, vp, vp, u32, vp]
        _lib.pt_create_context.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), u32, C.POINTER(SceneHeader), vp, vp, vp, C.POINTER(TextureDesc), u32]
        _lib.pt_destroy_context.argtypes = [vp]
        _lib.pt_destroy_context.restype = None
        _lib.pt_context_device_count.argtypes = [vp]
        _lib.pt_context_device_count.restype = u32
        _lib.pt_context_scene.argtypes = [vp, u32]
        _lib.pt_context_scene.restype = vp
        _lib.pt_context_set_lights.argtypes = [vp, vp, u32]
        _lib.pt_context_set_light_sampling.argtypes = [vp, i32]
        _lib.pt_context_set_render_settings.argtypes = [vp, i32, u32, u32]
        _lib.pt_context_create_render_targets.argtypes = [vp, u32, u32]
        _lib.pt_context_render_samples.argtypes = [vp, C.POINTER(Constants), u32, u32, i32]
        _lib.pt_context_synchronize.argtypes = [vp]
        _lib.pt_context_reduce.argtypes = [vp]
        _lib.pt_context_last_reduce_ms.argtypes = [vp]
        _lib.pt_context_last_reduce_ms.restype = C.c_double
        _lib.pt_context_frame_slice.argtypes = [u32, u32, u32, u32, C.POINTER(u32), C.POINTER(u32)]
        _lib.pt_get_device_count.argtypes = []
        _lib.pt_get_device_count.restype = u32
    return _lib


def _ptr(array):
    return array.ctypes.data_as(C.c_void_p)


class PtError(RuntimeError):
    pass


def _check(code, what):
    if code != 0:
        raise PtError("%s failed (code %d)" % (what, code))


class Scene:
    """A scene resident on one GPU. Arrays follow the .vks layout (src/scene.h:24-38)."""

    def __init__(self, material_count, factor, summand, positions, normals_uvs, materials, textures, device=0):
        """textures: list of (width, height, format, [mip payload bytes/arrays...]), three per material."""
        positions = np.ascontiguousarray(positions, np.uint32).reshape(-1, 2)
        normals_uvs = np.ascontiguousarray(normals_uvs, np.uint16).reshape(-1, 4)
        materials = np.ascontiguousarray(materials, np.uint8).reshape(-1)
        header = SceneHeader()
        header.material_count = material_count
        header.triangle_count = len(materials)
        header.dequantization_factor[:] = [float(v) for v in factor]
        header.dequantization_summand[:] = [float(v) for v in summand]
        self.factor = np.array(factor, np.float32)
        self.summand = np.array(summand, np.float32)
        descs = (TextureDesc * max(1, len(textures)))()
        keep = []
        for i, (width, height, fmt, mips) in enumerate(textures):
            mips = [np.ascontiguousarray(np.frombuffer(bytes(m), np.uint8) if not isinstance(m, np.ndarray) else m) for m in mips]
            data = (C.c_void_p * len(mips))(*[m.ctypes.data for m in mips])
            sizes = (C.c_uint64 * len(mips))(*[m.nbytes for m in mips])
            keep += [mips, data, sizes]
            descs[i] = TextureDesc(width, height, fmt, len(mips), data, sizes)
        self.handle = C.c_void_p()
        self.device = device
        _check(lib().pt_create_scene(C.byref(self.handle), device, C.byref(header), _ptr(positions), _ptr(normals_uvs), _ptr(materials), descs, len(textures)), "pt_create_scene")
        self.triangle_count = len(materials)
        self.width = self.height = 0

    def close(self):
        if getattr(self, "handle", None):
            lib().pt_destroy_scene(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def set_lights(self, lights):
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 4)
        _check(lib().pt_set_lights(self.handle, _ptr(lights), len(lights)), "pt_set_lights")

    def set_light_sampling(self, light_sampling):
        _check(lib().pt_set_light_sampling(self.handle, light_sampling), "pt_set_light_sampling")

    def set_shading_math(self, shading_math):
        _check(lib().pt_set_shading_math(self.handle, shading_math), "pt_set_shading_math")

    def set_render_settings(self, sampling_strategy=SAMPLING_STRATEGY_NEE, path_length=4, emission_material_index=0):
        _check(lib().pt_set_render_settings(self.handle, sampling_strategy, path_length, emission_material_index), "pt_set_render_settings")

    def create_render_targets(self, width, height):
        _check(lib().pt_create_render_targets(self.handle, width, height), "pt_create_render_targets")
        self.width, self.height = width, height

    def render_samples(self, constants, first_frame_index, sample_count, frame_stride=1, accumulate=True):
        _check(lib().pt_render_samples(self.handle, C.byref(constants), first_frame_index, sample_count, frame_stride, int(accumulate)), "pt_render_samples")

    def synchronize(self):
        _check(lib().pt_synchronize(self.handle), "pt_synchronize")

    def read_accumulation(self):
        rgba = np.zeros((self.height, self.width, 4), np.float32)
        count = C.c_uint32()
        _check(lib().pt_read_accumulation(self.handle, _ptr(rgba), C.byref(count)), "pt_read_accumulation")
        return rgba, count.value

    def accumulation_device_pointer(self):
        return lib().pt_get_accumulation_device_pointer(self.handle)

    def set_sample_count(self, count):
        _check(lib().pt_set_sample_count(self.handle, count), "pt_set_sample_count")

    def set_instrumentation(self, kernel_timing=False, visit_counters=False):
        _check(lib().pt_set_instrumentation(self.handle, int(kernel_timing), int(visit_counters)), "pt_set_instrumentation")

    def statistics(self):
        stats = Statistics()
        _check(lib().pt_get_statistics(self.handle, C.byref(stats)), "pt_get_statistics")
        return stats

    def reset_statistics(self):
        _check(lib().pt_reset_statistics(self.handle), "pt_reset_statistics")

    def bvh_info(self):
        info = BvhInfo()
        _check(lib().pt_get_bvh_info(self.handle, C.byref(info)), "pt_get_bvh_info")
        return info

    def trace_primary_rays(self, constants, frame_index):
        tri = np.zeros((self.height, self.width), np.int32)
        barys = np.zeros((self.height, self.width, 2), np.float32)
        front = np.zeros((self.height, self.width), np.uint8)
        _check(lib().pt_trace_primary_rays(self.handle, C.byref(constants), frame_index, _ptr(tri), _ptr(barys), _ptr(front)), "pt_trace_primary_rays")
        return tri, barys, front

    def trace_rays(self, origins, directions):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        tri = np.zeros(len(o), np.int32)
        tuv = np.zeros((len(o), 3), np.float32)
        front = np.zeros(len(o), np.uint8)
        _check(lib().pt_trace_rays(self.handle, _ptr(o), _ptr(d), len(o), _ptr(tri), _ptr(tuv), _ptr(front)), "pt_trace_rays")
        return tri, tuv, front

    def evaluate_shading_data(self, constants, triangles, barys, front, out_dirs, footprints=None):
        triangles = np.ascontiguousarray(triangles, np.int32).reshape(-1)
        barys = np.ascontiguousarray(barys, np.float32).reshape(-1, 2)
        front = np.ascontiguousarray(front, np.uint8).reshape(-1)
        out_dirs = np.ascontiguousarray(out_dirs, np.float32).reshape(-1, 3)
        out = np.zeros((len(triangles), 20), np.float32)
        _check(lib().pt_evaluate_shading_data(self.handle, C.byref(constants), _ptr(triangles), _ptr(barys), _ptr(front), _ptr(out_dirs),
                                                  None if footprints is None else _ptr(np.ascontiguousarray(footprints, np.float32).reshape(-1)), len(triangles), _ptr(out)), "pt_evaluate_shading_data")
        return out

    def evaluate_function(self, function, shading_data, inputs):
        """pt_evaluate_function(): inputs (n, <= 4) floats -> (n, 4) floats."""
        shading_data = np.ascontiguousarray(shading_data, np.float32).reshape(20)
        inputs = np.atleast_2d(np.asarray(inputs, np.float32))
        padded = np.zeros((len(inputs), 4), np.float32)
        padded[:, :inputs.shape[1]] = inputs
        out = np.zeros((len(padded), 4), np.float32)
        _check(lib().pt_evaluate_function(self.handle, function, _ptr(shading_data), _ptr(padded), len(padded), _ptr(out)), "pt_evaluate_function")
        return out


(FUNCTION_SAMPLE_GGX_IN_DIR, FUNCTION_GGX_IN_DIR_DENSITY, FUNCTION_SAMPLE_HEMISPHERE_PSA, FUNCTION_HEMISPHERE_PSA_DENSITY,
 FUNCTION_SAMPLE_HEMISPHERE_SPHERICAL, FUNCTION_HEMISPHERE_SPHERICAL_DENSITY, FUNCTION_SAMPLE_FROSTBITE_BRDF, FUNCTION_FROSTBITE_BRDF_DENSITY,
 FUNCTION_FROSTBITE_BRDF, FUNCTION_SAMPLE_LIGHTS, FUNCTION_LIGHTS_DENSITY) = range(11)


def random_numbers(x, y, frame_index, draw_count, device=0):
    out = np.zeros((draw_count, 2), np.float32)
    _check(lib().pt_random_numbers(device, x, y, frame_index, draw_count, _ptr(out)), "pt_random_numbers")
    return out
