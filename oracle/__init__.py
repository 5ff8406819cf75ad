"""ctypes binding of the CPU oracle (libpt_oracle.so).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py. The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import formats

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")


def build(force=False):
    sources = [os.path.join(_HERE, n) for n in ("pt_oracle.c", "pt_oracle_host.c", "pt_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in sources):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


class Constants(C.Structure):
    """constants_t of src/main.h:237-256."""
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


class _Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("mip_count", C.c_uint32),
                ("payload", C.c_void_p), ("mip_offsets", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_create_scene.restype = C.c_void_p
        _lib.oracle_create_scene.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        _lib.oracle_destroy_scene.argtypes = [C.c_void_p]
        _lib.oracle_render.restype = C.c_uint64
        _lib.oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_int]
        _lib.oracle_render_light_sampling.restype = C.c_uint64
        _lib.oracle_render_light_sampling.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                                      C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_int]
        _lib.oracle_primary_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_rng_stream.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib.oracle_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_shading_data.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.oracle_primary_cones.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib.oracle_sample_texture.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.oracle_evaluate_function.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.oracle_sincos.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        _lib.oracle_log.restype = C.c_float
        _lib.oracle_log.argtypes = [C.c_float]
        _lib.oracle_erfinv.restype = C.c_float
        _lib.oracle_erfinv.argtypes = [C.c_float]
        _lib.oracle_world_to_projection_space.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        _lib.oracle_invert_mat4.argtypes = [C.c_void_p, C.c_void_p]
        _lib.oracle_write_constants.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    return _lib


def _ptr(array):
    return array.ctypes.data_as(C.c_void_p)


class OracleScene:
    def __init__(self, scene: formats.Scene):
        self.scene = scene
        self._keep = []
        textures = (_Texture * max(1, len(scene.textures)))()
        for i, t in enumerate(scene.textures):
            payload = np.ascontiguousarray(t.payload)
            offsets = np.ascontiguousarray(t.mip_offsets, dtype=np.uint64)
            self._keep += [payload, offsets]
            textures[i] = _Texture(t.width, t.height, t.format, t.mip_count, payload.ctypes.data, offsets.ctypes.data)
        factor = np.ascontiguousarray(scene.dequantization_factor, np.float32)
        summand = np.ascontiguousarray(scene.dequantization_summand, np.float32)
        positions = np.ascontiguousarray(scene.positions, np.uint32)
        normals = np.ascontiguousarray(scene.normals_uvs, np.uint16)
        materials = np.ascontiguousarray(scene.materials, np.uint8)
        self.handle = lib().oracle_create_scene(len(scene.material_names), scene.triangle_count, _ptr(factor), _ptr(summand),
                                                _ptr(positions), _ptr(normals), _ptr(materials), C.byref(textures), len(scene.textures))

    def close(self):
        if self.handle:
            lib().oracle_destroy_scene(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def render(self, constants: Constants, lights, strategy=3, path_length=4, emission_material_index=None,
               first_frame_index=0, frame_count=1, frame_stride=1, accum=None, threads=1, light_sampling=0):
        """light_sampling 0: loops over all lights (the reference); 1: light tree."""
        width, height = int(constants.viewport_size[0]), int(constants.viewport_size[1])
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 4)
        accumulate = accum is not None
        if accum is None:
            accum = np.zeros((height, width, 4), np.float32)
        if emission_material_index is None:
            emission_material_index = self.scene.emission_material_index
        rays = lib().oracle_render_light_sampling(self.handle, C.byref(constants), _ptr(lights), len(lights), light_sampling, strategy, path_length,
                                                  emission_material_index, first_frame_index, frame_count, frame_stride, int(accumulate), _ptr(accum), threads)
        return accum, rays

    def primary_cones(self, constants: Constants, frame_index):
        width, height = int(constants.viewport_size[0]), int(constants.viewport_size[1])
        widths = np.zeros((height, width), np.float32)
        spreads = np.zeros((height, width), np.float32)
        lib().oracle_primary_cones(C.byref(constants), frame_index, _ptr(widths), _ptr(spreads))
        return widths, spreads

    def primary_hits(self, constants: Constants, frame_index):
        width, height = int(constants.viewport_size[0]), int(constants.viewport_size[1])
        triangles = np.zeros((height, width), np.int32)
        barys = np.zeros((height, width, 2), np.float32)
        front = np.zeros((height, width), np.uint8)
        lib().oracle_primary_hits(self.handle, C.byref(constants), frame_index, _ptr(triangles), _ptr(barys), _ptr(front))
        return triangles, barys, front

    def trace_rays(self, origins, directions):
        origins = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        directions = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        n = len(origins)
        triangles = np.zeros(n, np.int32)
        tuv = np.zeros((n, 3), np.float32)
        front = np.zeros(n, np.uint8)
        lib().oracle_trace_rays(self.handle, _ptr(origins), _ptr(directions), n, _ptr(triangles), _ptr(tuv), _ptr(front))
        return triangles, tuv, front

    def shading_data(self, constants, triangles, barys, front, out_dirs, emission_material_index=None, footprints=None):
        triangles = np.ascontiguousarray(triangles, np.int32).reshape(-1)
        barys = np.ascontiguousarray(barys, np.float32).reshape(-1, 2)
        front = np.ascontiguousarray(front, np.uint8).reshape(-1)
        out_dirs = np.ascontiguousarray(out_dirs, np.float32).reshape(-1, 3)
        if emission_material_index is None:
            emission_material_index = self.scene.emission_material_index
        out = np.zeros((len(triangles), 20), np.float32)
        lib().oracle_shading_data(self.handle, C.byref(constants), emission_material_index, _ptr(triangles), _ptr(barys), _ptr(front), _ptr(out_dirs),
                                  None if footprints is None else _ptr(np.ascontiguousarray(footprints, np.float32).reshape(-1)), len(triangles), _ptr(out))
        return out

    def sample_texture(self, texture_index, uvs, lods=None):
        uvs = np.ascontiguousarray(uvs, np.float32).reshape(-1, 2)
        out = np.zeros((len(uvs), 3), np.float32)
        lods = None if lods is None else np.ascontiguousarray(lods, np.float32).reshape(-1)
        lib().oracle_sample_texture(self.handle, texture_index, _ptr(uvs), None if lods is None else _ptr(lods), len(uvs), _ptr(out))
        return out


def evaluate_function(function, shading_data, inputs, lights=None):
    """oracle_evaluate_function(): inputs (n, <= 4) floats -> (n, 4) floats."""
    shading_data = np.ascontiguousarray(shading_data, np.float32).reshape(20)
    inputs = np.atleast_2d(np.asarray(inputs, np.float32))
    padded = np.zeros((len(inputs), 4), np.float32)
    padded[:, :inputs.shape[1]] = inputs
    lights = np.zeros((0, 4), np.float32) if lights is None else np.ascontiguousarray(lights, np.float32).reshape(-1, 4)
    out = np.zeros((len(padded), 4), np.float32)
    lib().oracle_evaluate_function(function, _ptr(shading_data), _ptr(lights), len(lights), _ptr(padded), len(padded), _ptr(out))
    return out


def rng_stream(x, y, frame_index, draw_count):
    randoms = np.zeros((draw_count, 2), np.float32)
    seeds = np.zeros((draw_count, 2), np.uint32)
    lib().oracle_rng_stream(x, y, frame_index, draw_count, _ptr(randoms), _ptr(seeds))
    return randoms, seeds


def sincos(x):
    s, c = C.c_float(), C.c_float()
    lib().oracle_sincos(x, C.byref(s), C.byref(c))
    return s.value, c.value


def log(x):
    return lib().oracle_log(x)


def erfinv(x):
    return lib().oracle_erfinv(x)


def write_constants(spec, width, height, scene: formats.Scene, lights, accum_frame_count=0) -> Constants:
    """write_constant_buffer() of src/main.c:629-676 for a 128-byte scene spec record."""
    spec = np.ascontiguousarray(spec)
    lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 4)
    factor = np.ascontiguousarray(scene.dequantization_factor, np.float32)
    summand = np.ascontiguousarray(scene.dequantization_summand, np.float32)
    constants = Constants()
    lib().oracle_write_constants(C.byref(constants), _ptr(spec), width, height, _ptr(factor), _ptr(summand), _ptr(lights), len(lights), accum_frame_count)
    return constants


def invert_mat4(mat):
    mat = np.ascontiguousarray(mat, np.float32).reshape(16)
    out = np.zeros(16, np.float32)
    lib().oracle_invert_mat4(_ptr(out), _ptr(mat))
    return out.reshape(4, 4)


def world_to_projection_space(spec, aspect):
    spec = np.ascontiguousarray(spec)
    out = np.zeros(16, np.float32)
    # the camera_t starts 4 bytes into scene_spec_t
    lib().oracle_world_to_projection_space(_ptr(out), C.c_void_p(spec.ctypes.data + 4), aspect)
    return out.reshape(4, 4)
