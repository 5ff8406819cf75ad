"""ctypes binding of tests/emu/libpt_emu.so: the per-thread kernel bodies of the HIP library run on the host.

TEST INFRASTRUCTURE ONLY. It lets the CPU test suite check BVH construction, traversal and shading logic of the
gfx950 kernels against the oracle without a GPU. The product library never loads it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libpt_emu.so")
_CSRC = os.path.join(_HERE, "..", "..", "path_tracer_amd", "csrc")


def build(force=False):
    sources = [os.path.join(_HERE, "pt_emu.cpp")] + [os.path.join(_CSRC, n) for n in ("pt_common.h", "pt_bvh_build.h", "pt_trace.h", "pt_shade.h", "pt_light_tree.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in sources):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-o", _LIB, os.path.join(_HERE, "pt_emu.cpp")])


class _Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("mip_count", C.c_uint32),
                ("payload", C.c_void_p), ("mip_offsets", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.emu_create_scene.restype = C.c_void_p
        _lib.emu_create_scene.argtypes = [C.c_uint64, C.c_uint64] + [C.c_void_p] * 6 + [C.c_uint32]
        _lib.emu_destroy_scene.argtypes = [C.c_void_p]
        _lib.emu_set_lights.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        _lib.emu_bvh_info.argtypes = [C.c_void_p, C.c_void_p]
        _lib.emu_light_grid_mismatches.restype = C.c_uint32
        _lib.emu_light_grid_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib.emu_light_grid_info.restype = None
        _lib.emu_light_grid_info.argtypes = [C.c_void_p, C.c_void_p]
        _lib.emu_lights_density.restype = None
        _lib.emu_lights_density.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        _lib.emu_light_tree_reuse_mismatches.restype = C.c_uint32
        _lib.emu_light_tree_reuse_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        _lib.emu_binary_tree_cost.restype = C.c_double
        _lib.emu_binary_tree_cost.argtypes = [C.c_void_p]
        _lib.emu_trace_rays_brute_force.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.emu_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.emu_primary_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.emu_render.restype = C.c_uint64
        _lib.emu_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        _lib.emu_primary_cones.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib.emu_shading_data.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.emu_evaluate_function.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.emu_sincos.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        _lib.emu_log.restype = C.c_float
        _lib.emu_log.argtypes = [C.c_float]
        _lib.emu_erfinv.restype = C.c_float
        _lib.emu_erfinv.argtypes = [C.c_float]
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _footprints(footprints, count):
    """Pointer to count float32 footprints, or NULL for none (which the callee reads as zeros)."""
    if footprints is None:
        return None
    array = np.ascontiguousarray(footprints, np.float32).reshape(-1)
    assert len(array) == count
    _footprints.keep = array
    return array.ctypes.data_as(C.c_void_p)


class EmuScene:
    def __init__(self, scene):
        self.scene = scene
        self._keep = []
        textures = (_Texture * max(1, len(scene.textures)))()
        for i, t in enumerate(scene.textures):
            payload = np.ascontiguousarray(t.payload)
            offsets = np.ascontiguousarray(t.mip_offsets, dtype=np.uint64)
            self._keep += [payload, offsets]
            textures[i] = _Texture(t.width, t.height, t.format, t.mip_count, payload.ctypes.data, offsets.ctypes.data)
        f = np.ascontiguousarray(scene.dequantization_factor, np.float32)
        s = np.ascontiguousarray(scene.dequantization_summand, np.float32)
        p = np.ascontiguousarray(scene.positions, np.uint32)
        n = np.ascontiguousarray(scene.normals_uvs, np.uint16)
        m = np.ascontiguousarray(scene.materials, np.uint8)
        self.handle = lib().emu_create_scene(len(scene.material_names), scene.triangle_count, _ptr(f), _ptr(s), _ptr(p), _ptr(n), _ptr(m), C.byref(textures), len(scene.textures))

    def close(self):
        if self.handle:
            lib().emu_destroy_scene(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def set_lights(self, lights):
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 4)
        lib().emu_set_lights(self.handle, _ptr(lights), len(lights))

    def evaluate_function(self, function, shading_data, inputs):
        """pt_evaluate_function() of the kernel bodies: inputs (n, <= 4) floats -> (n, 4) floats."""
        shading_data = np.ascontiguousarray(shading_data, np.float32).reshape(20)
        inputs = np.atleast_2d(np.asarray(inputs, np.float32))
        padded = np.zeros((len(inputs), 4), np.float32)
        padded[:, :inputs.shape[1]] = inputs
        out = np.zeros((len(padded), 4), np.float32)
        lib().emu_evaluate_function(self.handle, function, _ptr(shading_data), _ptr(padded), len(padded), _ptr(out))
        return out

    def light_grid_mismatches(self, origins, directions):
        """(rays whose lights found through the grid differ from a loop over all lights, rays that met more than four
        lights, lights met in total), or None if the lights have no grid."""
        origins, directions = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (origins, directions))
        overflows, hits = C.c_uint32(0), C.c_uint32(0)
        mismatches = lib().emu_light_grid_mismatches(self.handle, _ptr(origins), _ptr(directions), len(origins), C.byref(overflows), C.byref(hits))
        return None if mismatches == 0xffffffff else (int(mismatches), int(overflows.value), int(hits.value))

    def light_grid_info(self):
        """Cells per axis and entries of the grid over the lights (zeros: none)."""
        out = np.zeros(4, np.uint32)
        lib().emu_light_grid_info(self.handle, _ptr(out))
        return [int(v) for v in out]

    def lights_density(self, positions, normals, directions_0, directions_1, mode):
        """Densities of two directions per vertex under the light tree's sampling; mode 0: through the grid over the
        lights, 1: by the walk of the tree, 2: by a loop over all lights."""
        arrays = [np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (positions, normals, directions_0, directions_1)]
        out = np.zeros((len(arrays[0]), 2), np.float32)
        lib().emu_lights_density(self.handle, *[_ptr(a) for a in arrays], len(arrays[0]), mode, _ptr(out))
        return out

    def light_tree_reuse_mismatches(self, positions, normals, randoms):
        positions, normals = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (positions, normals))
        randoms = np.ascontiguousarray(randoms, np.float32).reshape(-1, 2)
        return int(lib().emu_light_tree_reuse_mismatches(self.handle, _ptr(positions), _ptr(normals), _ptr(randoms), len(positions)))

    def bvh_info(self):
        out = np.zeros(3, np.uint32)
        lib().emu_bvh_info(self.handle, _ptr(out))
        return {"node_count": int(out[0]), "depth": int(out[1]), "max_stack": int(out[2])}

    def trace_rays(self, origins, directions):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        tri = np.zeros(len(o), np.int32)
        tuv = np.zeros((len(o), 3), np.float32)
        front = np.zeros(len(o), np.uint8)
        visits = np.zeros(2, np.uint64)
        lib().emu_trace_rays(self.handle, _ptr(o), _ptr(d), len(o), _ptr(tri), _ptr(tuv), _ptr(front), _ptr(visits))
        return tri, tuv, front, visits

    def trace_rays_brute_force(self, origins, directions):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        tri = np.zeros(len(o), np.int32)
        tuv = np.zeros((len(o), 3), np.float32)
        front = np.zeros(len(o), np.uint8)
        lib().emu_trace_rays_brute_force(self.handle, _ptr(o), _ptr(d), len(o), _ptr(tri), _ptr(tuv), _ptr(front))
        return tri, tuv, front

    def primary_cones(self, constants, frame_index):
        width, height = int(constants.viewport_size[0]), int(constants.viewport_size[1])
        widths = np.zeros((height, width), np.float32)
        spreads = np.zeros((height, width), np.float32)
        lib().emu_primary_cones(C.byref(constants), frame_index, _ptr(widths), _ptr(spreads))
        return widths, spreads

    def primary_hits(self, constants, frame_index):
        w, h = int(constants.viewport_size[0]), int(constants.viewport_size[1])
        tri = np.zeros((h, w), np.int32)
        barys = np.zeros((h, w, 2), np.float32)
        front = np.zeros((h, w), np.uint8)
        lib().emu_primary_hits(self.handle, C.byref(constants), frame_index, _ptr(tri), _ptr(barys), _ptr(front))
        return tri, barys, front

    def render(self, constants, strategy=3, path_length=4, emission_material_index=None, first_frame_index=0, frame_count=1, frame_stride=1, accum=None):
        w, h = int(constants.viewport_size[0]), int(constants.viewport_size[1])
        accumulate = accum is not None
        if accum is None:
            accum = np.zeros((h, w, 4), np.float32)
        if emission_material_index is None:
            emission_material_index = self.scene.emission_material_index
        rays = lib().emu_render(self.handle, C.byref(constants), strategy, path_length, emission_material_index, first_frame_index, frame_count, frame_stride, int(accumulate), _ptr(accum))
        return accum, rays

    def shading_data(self, constants, triangles, barys, front, out_dirs, emission_material_index=None, footprints=None):
        triangles = np.ascontiguousarray(triangles, np.int32).reshape(-1)
        barys = np.ascontiguousarray(barys, np.float32).reshape(-1, 2)
        front = np.ascontiguousarray(front, np.uint8).reshape(-1)
        out_dirs = np.ascontiguousarray(out_dirs, np.float32).reshape(-1, 3)
        if emission_material_index is None:
            emission_material_index = self.scene.emission_material_index
        out = np.zeros((len(triangles), 20), np.float32)
        lib().emu_shading_data(self.handle, C.byref(constants), emission_material_index, _ptr(triangles), _ptr(barys), _ptr(front), _ptr(out_dirs), _footprints(footprints, len(triangles)), len(triangles), _ptr(out))
        return out
