"""Procedural stand-in scenes in the reference's file formats.

The scenes the reference's numbers are quoted on (Bistro, Arcade, ...) are not redistributable with this
repository, so benchmarks and large-scene tests use generated scenes of the same shape instead: the same files in
the same places (data/<scene>.vks, data/<scene>_textures/*.vkt, data/<scene>.lights, data/saves/.../*.rt_save; the
table at src/main.c:27-44 of the reference), millions of small triangles with displaced facades and foliage-like
clutter, about a hundred textured materials, emissive geometry where the light spheres are.

    python -m path_tracer_amd.tools.procedural --scene city --out /tmp/pt_city [--triangles 2800000]
"""
import argparse
import os
import shutil
import time

import numpy as np

from . import vks, vkt

EMISSION_MATERIAL = "_emission"
# Counts the changes of what the generators make of the same arguments (2: random numbers from Pcg2d, triangle counts
# not below what was asked for)
REVISION = 2


class Pcg2d:
    """The random numbers of the generators: the shader's own generator (get_random_numbers(),
    src/shaders/pathtrace.frag.glsl:23-34), so that a scene can be generated again without numpy's generators.

    Number pair i of a stream (i counts from 0 over all draws of the object) is the state uvec2(i mod 2^32,
    seed + (i >> 32)) advanced once by that function; its two words x, y give the uniform numbers (x + 0.5) / 2^32 and
    (y + 0.5) / 2^32 in float64, in that order (never 0 or 1). A draw of n numbers uses ceil(n / 2) pairs and drops the
    last number when n is odd. Normal numbers are Box-Muller over one pair each: sqrt(-2 ln u_x) cos(2 pi u_y), then
    the same with sin, in float64. Arrays are filled in C order."""

    def __init__(self, seed):
        self.seed, self.counter = int(seed), 0

    def _pairs(self, count):
        index = self.counter + np.arange(count, dtype=np.uint64)
        self.counter += count
        x = (index & np.uint64(0xffffffff)).astype(np.uint32)
        y = ((index >> np.uint64(32)) + np.uint64(self.seed & 0xffffffff)).astype(np.uint32)
        multiplier, increment, shift = np.uint32(1664525), np.uint32(1013904223), np.uint32(16)
        with np.errstate(over="ignore"):
            x, y = x * multiplier + increment, y * multiplier + increment
            x = x + y * multiplier
            y = y + x * multiplier
            x, y = x ^ (x >> shift), y ^ (y >> shift)
            x = x + y * multiplier
            y = y + x * multiplier
            x, y = x ^ (x >> shift), y ^ (y >> shift)
        return x, y

    def _unit(self, count):
        x, y = self._pairs((count + 1) // 2)
        numbers = np.stack([x, y], axis=1).reshape(-1)[:count]
        return (numbers.astype(np.float64) + 0.5) / 4294967296.0

    def uniform(self, low=0.0, high=1.0, size=None):
        low, high = np.asarray(low, np.float64), np.asarray(high, np.float64)
        shape = np.broadcast_shapes(low.shape, high.shape) if size is None else (tuple(size) if np.ndim(size) else (int(size),))
        numbers = low + (high - low) * self._unit(int(np.prod(shape, dtype=np.int64))).reshape(shape)
        return float(numbers) if numbers.ndim == 0 else numbers

    def normal(self, loc=0.0, scale=1.0, size=None):
        shape = () if size is None else (tuple(size) if np.ndim(size) else (int(size),))
        count = int(np.prod(shape, dtype=np.int64))
        x, y = self._pairs((count + 1) // 2)
        radius = np.sqrt(-2.0 * np.log((x.astype(np.float64) + 0.5) / 4294967296.0))
        angle = 2.0 * np.pi * (y.astype(np.float64) + 0.5) / 4294967296.0
        numbers = np.stack([radius * np.cos(angle), radius * np.sin(angle)], axis=1).reshape(-1)[:count].reshape(shape)
        numbers = loc + scale * numbers
        return float(numbers) if numbers.ndim == 0 else numbers


class _Mesh:
    def __init__(self):
        self.positions, self.uvs, self.materials = [], [], []

    def add(self, triangles, uvs, material):
        triangles = np.asarray(triangles, np.float32).reshape(-1, 3, 3)
        self.positions.append(triangles)
        self.uvs.append(np.asarray(uvs, np.float32).reshape(-1, 3, 2))
        self.materials.append(np.full(len(triangles), material, np.uint8))

    def count(self):
        return sum(len(p) for p in self.positions)

    def arrays(self):
        positions = np.concatenate(self.positions)
        edge_1, edge_2 = positions[:, 1] - positions[:, 0], positions[:, 2] - positions[:, 0]
        normals = np.cross(edge_1, edge_2)
        length = np.linalg.norm(normals, axis=1, keepdims=True)
        normals = np.where(length > 0.0, normals / np.maximum(length, 1e-30), np.array([0.0, 0.0, 1.0], np.float32))
        return positions, np.repeat(normals[:, None, :], 3, axis=1), np.concatenate(self.uvs), np.concatenate(self.materials)


def _grid(mesh, origin, du, dv, nu, nv, height, material, uv_scale=0.5):
    """A displaced rectangle: counter-clockwise seen from the side cross(du, dv) points to; height: (nv+1, nu+1)."""
    origin, du, dv = (np.asarray(a, np.float64) for a in (origin, du, dv))
    normal = np.cross(du, dv)
    normal /= np.linalg.norm(normal)
    u, v = np.arange(nu + 1) / nu, np.arange(nv + 1) / nv
    points = origin + u[None, :, None] * du + v[:, None, None] * dv + np.asarray(height)[:, :, None] * normal
    coords = np.stack(np.broadcast_arrays(u[None, :] * np.linalg.norm(du) * uv_scale, v[:, None] * np.linalg.norm(dv) * uv_scale), axis=2)
    p00, p10, p11, p01 = points[:-1, :-1], points[:-1, 1:], points[1:, 1:], points[1:, :-1]
    c00, c10, c11, c01 = coords[:-1, :-1], coords[:-1, 1:], coords[1:, 1:], coords[1:, :-1]
    mesh.add(np.concatenate([np.stack([p00, p10, p11], axis=2).reshape(-1, 3, 3), np.stack([p00, p11, p01], axis=2).reshape(-1, 3, 3)]),
             np.concatenate([np.stack([c00, c10, c11], axis=2).reshape(-1, 3, 2), np.stack([c00, c11, c01], axis=2).reshape(-1, 3, 2)]), material)


def _box(mesh, lo, hi, material):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    size = hi - lo
    flat = np.zeros((2, 2))
    x, y, z = np.eye(3) * size
    _grid(mesh, lo, y, x, 1, 1, flat, material)                # bottom (-z)
    _grid(mesh, lo + z, x, y, 1, 1, flat, material)            # top (+z)
    _grid(mesh, lo, x, z, 1, 1, flat, material)                # -y
    _grid(mesh, lo + y, z, x, 1, 1, flat, material)            # +y
    _grid(mesh, lo, z, y, 1, 1, flat, material)                # -x
    _grid(mesh, lo + x, y, z, 1, 1, flat, material)            # +x


def _icosphere(subdivisions):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    vertices = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                         [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    faces = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                      [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    triangles = vertices[faces]
    for _ in range(subdivisions):
        a, b, c = triangles[:, 0], triangles[:, 1], triangles[:, 2]
        ab, bc, ca = a + b, b + c, c + a
        triangles = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    return triangles / np.linalg.norm(triangles, axis=2, keepdims=True)


def _light_spheres(mesh, lights, material, subdivisions=3):
    """Emissive geometry slightly inside each light sphere, so that camera rays see the lamps."""
    unit = _icosphere(subdivisions)
    for x, y, z, radius in np.asarray(lights, np.float64):
        mesh.add(unit * (0.98 * radius) + np.array([x, y, z]), np.zeros((len(unit), 3, 2)), material)


def _facade_height(nu, nv, width, height, rng):
    """Window insets, floor ledges and brick-scale roughness for a facade of width x height metres."""
    u, v = np.arange(nu + 1) / nu * width, np.arange(nv + 1) / nv * height
    window = ((np.mod(u, 1.8)[None, :] > 0.5) & (np.mod(u, 1.8)[None, :] < 1.4) & (np.mod(v, 3.0)[:, None] > 0.9) & (np.mod(v, 3.0)[:, None] < 2.4))
    ledge = (np.mod(v, 3.0)[:, None] < 0.15) & np.ones_like(u, bool)[None, :]
    return -0.25 * window + 0.12 * ledge + rng.uniform(-0.01, 0.01, (nv + 1, nu + 1))


def _building(mesh, rng, x0, y0, x1, y1, height, cell, wall_material, roof_material):
    w, d = x1 - x0, y1 - y0
    nu_w, nu_d, nv = max(1, int(round(w / cell))), max(1, int(round(d / cell))), max(1, int(round(height / cell)))
    up = (0.0, 0.0, height)
    _grid(mesh, (x0, y0, 0), (w, 0, 0), up, nu_w, nv, _facade_height(nu_w, nv, w, height, rng), wall_material)      # -y
    _grid(mesh, (x1, y0, 0), (0, d, 0), up, nu_d, nv, _facade_height(nu_d, nv, d, height, rng), wall_material)      # +x
    _grid(mesh, (x1, y1, 0), (-w, 0, 0), up, nu_w, nv, _facade_height(nu_w, nv, w, height, rng), wall_material)     # +y
    _grid(mesh, (x0, y1, 0), (0, -d, 0), up, nu_d, nv, _facade_height(nu_d, nv, d, height, rng), wall_material)     # -x
    nr_w, nr_d = max(1, nu_w // 2), max(1, nu_d // 2)
    _grid(mesh, (x0, y0, height), (w, 0, 0), (0, d, 0), nr_w, nr_d, rng.uniform(0.0, 0.05, (nr_d + 1, nr_w + 1)), roof_material)


def _tree(mesh, rng, x, y, leaf_count, trunk_material, leaf_material):
    height, radius = rng.uniform(3.0, 5.0), rng.uniform(1.5, 2.5)
    _box(mesh, (x - 0.15, y - 0.15, 0.0), (x + 0.15, y + 0.15, height), trunk_material)
    direction = rng.normal(size=(leaf_count, 3))
    centre = np.array([x, y, height + 0.6 * radius]) + direction / np.linalg.norm(direction, axis=1, keepdims=True) \
        * radius * rng.uniform(0.0, 1.0, (leaf_count, 1)) ** (1.0 / 3.0) * np.array([1.0, 1.0, 0.8])
    leaves = centre[:, None, :] + rng.normal(scale=0.09, size=(leaf_count, 3, 3))
    mesh.add(leaves, np.tile(np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]]), (leaf_count, 1, 1)), leaf_material)


def _smooth_noise(rng, size, cutoff):
    spectrum = np.fft.rfft2(rng.normal(size=(size, size)))
    fy, fx = np.meshgrid(np.fft.fftfreq(size) * size, np.fft.rfftfreq(size) * size, indexing="ij")
    spectrum *= np.exp(-(fx ** 2 + fy ** 2) / (2.0 * cutoff ** 2))
    noise = np.fft.irfft2(spectrum, s=(size, size))
    noise -= noise.min()
    return noise / max(noise.max(), 1e-30)


def write_material_textures(directory, name, rng, size=256, emissive=False):
    """Base colour (BC1 sRGB), specular (BC1: occlusion, roughness, metalness) and normal (BC5) textures."""
    os.makedirs(directory, exist_ok=True)
    if emissive:
        base = np.ones((4, 4, 3))
        specular = np.ones((4, 4, 3)) * np.array([1.0, 1.0, 0.0])
        normal = np.full((4, 4, 2), 0.5)
    else:
        coarse, fine = _smooth_noise(rng, size, 4.0), _smooth_noise(rng, size, 40.0)
        y, x = np.mgrid[0:size, 0:size] / size
        bricks = ((np.mod(y * 16.0, 1.0) < 0.08) | (np.mod(x * 8.0 + 0.5 * np.floor(y * 16.0), 1.0) < 0.04)).astype(np.float64)
        tint, mortar = rng.uniform(0.15, 0.85, 3), rng.uniform(0.3, 0.6)
        pattern = rng.uniform(0.0, 1.0) < 0.6
        shade = (0.7 + 0.3 * coarse) * (0.85 + 0.15 * fine)
        base = tint[None, None, :] * shade[:, :, None]
        if pattern:
            base = base * (1.0 - bricks[:, :, None]) + mortar * bricks[:, :, None]
        roughness = np.clip(rng.uniform(0.25, 0.9) + 0.2 * (fine - 0.5), 0.05, 1.0)
        metalness = np.full((size, size), 1.0 if rng.uniform() < 0.1 else 0.0)
        specular = np.stack([np.ones((size, size)), roughness, metalness], axis=2)
        bump = 0.02 * fine + (0.03 * (1.0 - bricks) if pattern else 0.0)
        dx = (np.roll(bump, -1, axis=1) - np.roll(bump, 1, axis=1)) * size / 2.0 / 4.0
        dy = (np.roll(bump, -1, axis=0) - np.roll(bump, 1, axis=0)) * size / 2.0 / 4.0
        n = np.stack([-dx, -dy, np.ones_like(dx)], axis=2)
        n /= np.linalg.norm(n, axis=2, keepdims=True)
        normal = n[:, :, 0:2] * 0.5 + 0.5
    vkt.write_vkt(os.path.join(directory, name + "_BaseColor.vkt"), base, vkt.FORMAT_BC1_SRGB)
    vkt.write_vkt(os.path.join(directory, name + "_Specular.vkt"), specular, vkt.FORMAT_BC1_UNORM)
    vkt.write_vkt(os.path.join(directory, name + "_Normal.vkt"), normal, vkt.FORMAT_BC5_UNORM)


def _write_scene(root, scene_name, texture_dir, mesh, material_names, lights, save_path, save_writer, rng, texture_size):
    data = os.path.join(root, "data")
    os.makedirs(os.path.join(data, os.path.dirname(save_path)), exist_ok=True)
    positions, normals, uvs, materials = mesh.arrays()
    vks.write_vks(os.path.join(data, scene_name + ".vks"), material_names, positions, normals, uvs, materials)
    for name in material_names:
        write_material_textures(os.path.join(data, texture_dir), name, rng, texture_size, emissive=(name == EMISSION_MATERIAL))
    vks.write_lights(os.path.join(data, scene_name + ".lights"), lights)
    save_writer(os.path.join(data, save_path))
    return len(positions)


def generate_city(root, triangles=2800000, lights_file=None, save_file=None, material_count=100, texture_size=256, seed=1):
    """An outdoor block in the place of data/Bistro_outside.*: buildings with displaced facades on a lot grid,
    a cobbled ground, trees, and a lamp (pole + emissive sphere) at every light of lights_file (the reference's
    Bistro_outside.lights when given). Streets are kept free around the lamps and in front of the camera of the
    reference's bistro/outside quicksave. Returns the triangle count."""
    rng = Pcg2d(seed)
    if lights_file is not None:
        raw = np.fromfile(lights_file, np.uint8)
        lights = raw[4:].view(np.float32).reshape(-1, 4).astype(np.float64)
    else:
        lights = np.array([[x, y, 6.9, 0.37] for x in (-30.0, -10.0, 10.0, 30.0) for y in (-24.0, -8.0, 8.0, 24.0)])
    camera = np.array([-24.389008, 0.47295034, 7.4361873])
    material_names = ["facade_%02d" % i for i in range(material_count - 5)] + ["roof", "ground", "metal", "leaves", EMISSION_MATERIAL]
    roof, ground, metal, leaves, emission = range(material_count - 5, material_count)
    lo, hi = np.array([-66.0, -78.0]), np.array([96.0, 84.0])
    pitch, margin = 13.5, 1.75
    lots = []
    for ix in range(int((hi[0] - lo[0]) / pitch)):
        for iy in range(int((hi[1] - lo[1]) / pitch)):
            x0, y0 = lo[0] + ix * pitch + margin, lo[1] + iy * pitch + margin
            x1, y1 = x0 + pitch - 2 * margin, y0 + pitch - 2 * margin
            nearest = np.maximum(np.maximum(x0 - lights[:, 0], lights[:, 0] - x1), np.maximum(y0 - lights[:, 1], lights[:, 1] - y1))
            corridor = x1 > camera[0] - 4.0 and x0 < camera[0] + 40.0 and y1 > camera[1] - 7.0 and y0 < camera[1] + 7.0
            if nearest.min() > 2.5 and not corridor:
                lots.append((x0, y0, x1, y1))
    heights = rng.uniform(9.0, 26.0, len(lots))
    tree_spots = lights[:, 0:2] + rng.uniform(-1.0, 1.0, (len(lights), 2)) * 0.5 + np.array([2.2, 1.4])

    def build(cell, leaf_count):
        mesh = _Mesh()
        local = Pcg2d(seed + 1)
        for i, (lot, height) in enumerate(zip(lots, heights)):
            _building(mesh, local, *lot, height, cell, i % (material_count - 5), roof)
        n = max(1, int((hi[0] - lo[0]) / (2.0 * cell)))
        _grid(mesh, (lo[0], lo[1], 0.0), (hi[0] - lo[0], 0, 0), (0, hi[1] - lo[1], 0), n, n, local.uniform(0.0, 0.03, (n + 1, n + 1)), ground)
        for x, y, z, radius in lights:
            _box(mesh, (x - 0.06, y - 0.06, 0.0), (x + 0.06, y + 0.06, z - radius - 0.02), metal)
        for x, y in tree_spots:
            _tree(mesh, local, x, y, leaf_count, metal, leaves)
        _light_spheres(mesh, lights, emission)
        return mesh

    facade_area = sum(2.0 * ((x1 - x0) + (y1 - y0)) * h + 0.25 * (x1 - x0) * (y1 - y0) for (x0, y0, x1, y1), h in zip(lots, heights))
    ground_area = 0.25 * (hi[0] - lo[0]) * (hi[1] - lo[1])
    fixed = len(lights) * (1280 + 12 + 12)
    leaf_count = int(0.2 * triangles / max(1, len(tree_spots)))
    cell = (2.0 * (facade_area + ground_area) / max(1.0, triangles - fixed - leaf_count * len(tree_spots))) ** 0.5

    def count(cell):
        """The triangles that build(cell, leaf_count) makes."""
        total = len(lights) * (1280 + 12) + len(tree_spots) * (12 + leaf_count)
        for (x0, y0, x1, y1), height in zip(lots, heights):
            nu_w, nu_d, nv = max(1, int(round((x1 - x0) / cell))), max(1, int(round((y1 - y0) / cell))), max(1, int(round(height / cell)))
            total += 4 * (nu_w + nu_d) * nv + 2 * max(1, nu_w // 2) * max(1, nu_d // 2)
        return total + 2 * max(1, int((hi[0] - lo[0]) / (2.0 * cell))) ** 2

    # The estimate above is good to a few tenths of a per cent; the scene has at least the triangles asked for
    while count(cell) < triangles:
        cell *= 0.9995
    mesh = build(cell, leaf_count)
    assert mesh.count() == count(cell)

    def save_writer(path):
        if save_file is not None:
            shutil.copyfile(save_file, path)
        else:
            vks.write_rt_save(path, camera, (1.5144025, 0.0, -1.6373228), scene_file=0, frame_index=815, sky_strength=0.5, emission_strength=500.0, tonemapper=0)
    return _write_scene(root, "Bistro_outside", "Bistro_textures", mesh, material_names, lights, "saves/bistro/outside.rt_save", save_writer, rng, texture_size)


def generate_interior(root, triangles=1000000, light_count=32, material_count=24, texture_size=256, seed=2):
    """A hall in the place of data/Arcade.*: displaced walls, floor and ceiling, rows of cabinets and light_count
    ceiling lamps, each an analytic spherical light with emissive geometry. Returns the triangle count."""
    rng = Pcg2d(seed)
    sx, sy, sz = 24.0, 16.0, 5.0
    columns = int(np.ceil(light_count ** 0.5 * (sx / sy) ** 0.5))
    rows = int(np.ceil(light_count / columns))
    lights = np.array([[(i % columns + 0.5) * sx / columns, (i // columns + 0.5) * sy / rows, sz - 0.45, 0.12] for i in range(light_count)])
    material_names = ["surface_%02d" % i for i in range(material_count - 1)] + [EMISSION_MATERIAL]
    emission = material_count - 1
    mesh = _Mesh()
    area = 2.0 * (sx * sy + sx * sz + sy * sz)
    cabinets = [(x, y) for x in np.arange(3.0, sx - 2.0, 3.0) for y in np.arange(2.5, sy - 2.0, 3.5)]
    cabinet_area = len(cabinets) * (2.0 * (1.2 + 0.8) * 1.9 + 1.2 * 0.8)
    cell = ((area + cabinet_area) * 2.0 / max(1.0, triangles - light_count * 1280)) ** 0.5

    panels = [((0, 0, 0), (sx, 0, 0), (0, sy, 0), 0, 0.02),                 # floor, normal +z
              ((0, 0, sz), (0, sy, 0), (sx, 0, 0), 1, 0.02),                # ceiling, normal -z
              ((0, 0, 0), (0, 0, sz), (sx, 0, 0), 2, 0.02),                 # wall y = 0, normal +y
              ((0, sy, 0), (sx, 0, 0), (0, 0, sz), 3, 0.02),                # wall y = sy, normal -y
              ((0, 0, 0), (0, sy, 0), (0, 0, sz), 4, 0.02),                 # wall x = 0, normal +x
              ((sx, 0, 0), (0, 0, sz), (0, sy, 0), 5, 0.02)]                # wall x = sx, normal -x
    for i, (x, y) in enumerate(cabinets):
        material = 6 + i % (material_count - 7)
        panels += [((x, y, 0), (1.2, 0, 0), (0, 0, 1.9), material, 0.05), ((x + 1.2, y, 0), (0, 0.8, 0), (0, 0, 1.9), material, 0.05),
                   ((x + 1.2, y + 0.8, 0), (-1.2, 0, 0), (0, 0, 1.9), material, 0.05), ((x, y + 0.8, 0), (0, -0.8, 0), (0, 0, 1.9), material, 0.05),
                   ((x, y, 1.9), (1.2, 0, 0), (0, 0.8, 0), material, 0.01)]

    def divisions(du, dv, cell):
        return max(1, int(np.linalg.norm(du) / cell)), max(1, int(np.linalg.norm(dv) / cell))

    def count(cell):
        return light_count * 1280 + sum(2 * int(np.prod(divisions(du, dv, cell))) for _, du, dv, _, _ in panels)

    # At least the triangles asked for (the estimate of the cell size is good to about a per cent)
    while count(cell) < triangles:
        cell *= 0.999
    for origin, du, dv, material, amplitude in panels:
        nu, nv = divisions(du, dv, cell)
        u, v = np.arange(nu + 1) / nu * np.linalg.norm(du), np.arange(nv + 1) / nv * np.linalg.norm(dv)
        relief = amplitude * (np.sin(u * 5.0)[None, :] * np.sin(v * 5.0)[:, None]) + rng.uniform(-0.004, 0.004, (nv + 1, nu + 1))
        _grid(mesh, origin, du, dv, nu, nv, relief, material)
    _light_spheres(mesh, lights, emission)

    def save_writer(path):
        vks.write_rt_save(path, (1.0, 1.2, 1.7), (1.45, 0.0, -2.2), scene_file=2, frame_index=0, sky_strength=0.0, emission_strength=60.0, tonemapper=0)
    return _write_scene(root, "Arcade", "Arcade_textures", mesh, material_names, lights, "saves/Arcade/default.rt_save", save_writer, rng, texture_size)


def ensure_scene(kind, root, **options):
    """Generates the scene under root unless a finished one is there; returns root."""
    stamp = os.path.join(root, "generated_%s.txt" % kind)
    if not os.path.exists(stamp):
        os.makedirs(root, exist_ok=True)
        count = {"city": generate_city, "interior": generate_interior}[kind](root, **options)
        with open(stamp, "w") as file:
            file.write("%d triangles\n" % count)
    return root


def main():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("--scene", choices=("city", "interior"), default="city")
    parser.add_argument("--out", required=True)
    parser.add_argument("--triangles", type=int, default=None)
    parser.add_argument("--lights", default=None, help="a .lights file to place the lamps of the city at")
    parser.add_argument("--texture-size", type=int, default=256)
    arguments = parser.parse_args()
    start = time.time()
    options = {"texture_size": arguments.texture_size}
    if arguments.triangles:
        options["triangles"] = arguments.triangles
    if arguments.scene == "city":
        count = generate_city(arguments.out, lights_file=arguments.lights, **options)
    else:
        count = generate_interior(arguments.out, **options)
    print("%s: %d triangles under %s in %.1f s" % (arguments.scene, count, arguments.out, time.time() - start))


if __name__ == "__main__":
    main()
