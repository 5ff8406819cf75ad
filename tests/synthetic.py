"""Synthetic scenes in the .vks/.vkt data layout for tests (seeded, built in memory)."""
import numpy as np

from oracle import formats


def encode_octahedral(normals):
    n = normals / np.abs(normals).sum(axis=1, keepdims=True)
    xy = n[:, :2].copy()
    neg = n[:, 2] < 0
    folded = (1 - np.abs(n[neg][:, 1::-1])) * np.where(n[neg][:, :2] >= 0, 1.0, -1.0)
    xy[neg] = folded
    # stored value 32768 <-> 0, 1 <-> -1, 65535 <-> +1 (tools/io_export_vulkan_blender28.py:36-46)
    return np.clip(np.round(xy * 32767.0 + 32768.0), 1, 65535).astype(np.uint16)


def quantize_positions(points):
    lo, hi = points.min(axis=0), points.max(axis=0)
    extent = np.maximum(hi - lo, 1e-6)
    factor = (extent / (2 ** 21 - 1)).astype(np.float32)
    summand = lo.astype(np.float32)
    q = np.clip(np.round((points - lo) / factor.astype(np.float64)), 0, 2 ** 21 - 1).astype(np.uint64)
    word0 = (q[:, 0] | ((q[:, 1] & 0x7ff) << 21)).astype(np.uint32)
    word1 = ((q[:, 1] >> 11) | (q[:, 2] << 10)).astype(np.uint32)
    return np.stack([word0, word1], 1), factor, summand


def random_texture(rng, size, fmt, mips=None):
    """Random blocks on every level; mips=None gives the full chain down to 4x4 like the reference's converter."""
    if mips is None:
        mips = max(1, int(np.log2(size)) - 1)
    block = 16 if fmt == 141 else 8
    payloads, extents, offsets, offset = [], [], [], 0
    for m in range(mips):
        s = max(size >> m, 4)
        data = rng.integers(0, 256, (s // 4) * (s // 4) * block, dtype=np.uint8)
        payloads.append(data)
        extents.append((s, s, len(data)))
        offsets.append(offset)
        offset += len(data)
    return formats.Texture(size, size, fmt, mips, extents, np.array(offsets, np.uint64), np.concatenate(payloads))


def noise_texture(rng, size, fmt):
    """White noise with a proper mip chain: every level is the Gaussian-filtered level 0, compressed by the
    authoring tool of this repository (path_tracer_amd/tools/vkt.py), as a converter would write it."""
    import os
    import tempfile
    from path_tracer_amd.tools import vkt
    image = rng.uniform(0.0, 1.0, (size, size, 2 if fmt == 141 else 3))
    with tempfile.TemporaryDirectory() as directory:
        path = os.path.join(directory, "noise.vkt")
        vkt.write_vkt(path, image, fmt)
        return formats.load_vkt(path)


def triangle_soup(seed=1, triangle_count=4000, material_count=3, extent=10.0, size=0.6, texture_size=32, emission_fraction=0.05, filtered_mips=False):
    """Random small triangles in a cube, random block-compressed textures, first material emits. With filtered_mips
    the textures are noise whose coarser levels are filtered from level 0; otherwise every level is random blocks
    of its own (more distinct values per level, but no level is the mean of the one below)."""
    rng = np.random.default_rng(seed)
    centers = rng.uniform(-extent / 2, extent / 2, (triangle_count, 1, 3))
    verts = centers + rng.normal(scale=size, size=(triangle_count, 3, 3))
    positions, factor, summand = quantize_positions(verts.reshape(-1, 3))
    geometric = np.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0])
    geometric /= np.maximum(np.linalg.norm(geometric, axis=1, keepdims=True), 1e-12)
    normals = np.repeat(geometric, 3, axis=0) + rng.normal(scale=0.2, size=(3 * triangle_count, 3))
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    uv = rng.integers(0, 65536, (3 * triangle_count, 2)).astype(np.uint16)
    normals_uvs = np.concatenate([encode_octahedral(normals), uv], axis=1).astype(np.uint16)
    materials = rng.integers(1, material_count, triangle_count).astype(np.uint8) if material_count > 1 else np.zeros(triangle_count, np.uint8)
    materials[rng.random(triangle_count) < emission_fraction] = 0
    names = ["_emission"] + ["material_%d" % i for i in range(1, material_count)]
    textures = []
    make_texture = noise_texture if filtered_mips else random_texture
    for _ in range(material_count):
        textures += [make_texture(rng, texture_size, 132), make_texture(rng, texture_size, 131), make_texture(rng, texture_size, 141)]
    return formats.Scene(names, triangle_count, factor, summand, positions, normals_uvs, materials, textures)


def random_lights(seed, count, extent=10.0, radius=(0.1, 0.5)):
    rng = np.random.default_rng(seed)
    lights = np.zeros((count, 4), np.float32)
    lights[:, :3] = rng.uniform(-extent / 2, extent / 2, (count, 3))
    lights[:, 3] = rng.uniform(radius[0], radius[1], count)
    return lights


LIGHT_LAYOUTS = ("row", "clusters", "giant", "around_the_scene", "flat", "far_apart")


def light_layout(name):
    """Light sets that stress the lookup of the lights a ray meets: a row of spheres that a ray threads (more hits than
    the grid path keeps, so the walk of the light tree takes over per vertex), dense clusters (crowded cells), one light
    larger than the rest of the scene (listed in many cells), one light around the whole scene (directions sampled from
    inside a light are no unit vectors), all lights in a plane (one layer of cells), two groups a thousand extents
    apart (coarse cells; shading points beyond the grid's reach)."""
    rng = np.random.default_rng(77)
    if name == "row":
        return np.array([[-4.0 + 0.9 * i, 0.02 * i, 2.0, 0.3] for i in range(10)] + [[0.5, 1.0, 3.0, 0.2]], np.float32)
    if name == "clusters":
        centers = rng.uniform(-5, 5, (6, 3))
        return np.concatenate([np.concatenate([c + rng.normal(scale=0.15, size=(40, 3)), rng.uniform(0.02, 0.1, (40, 1))], axis=1) for c in centers]).astype(np.float32)
    if name == "giant":
        return np.concatenate([random_lights(5, 200), np.array([[0, 0, 60.0, 30.0]], np.float32)])
    if name == "around_the_scene":
        return np.concatenate([random_lights(5, 200), np.array([[0, 0, 0.0, 30.0]], np.float32)])
    if name == "flat":
        lights = random_lights(6, 150)
        lights[:, 2] = 2.5
        return lights
    if name == "far_apart":
        return np.concatenate([random_lights(7, 60), random_lights(8, 60) + np.array([9000.0, 0, 0, 0], np.float32)])
    raise ValueError(name)


def lit_room(seed=1, light_count=64, extent=12.0, texture_size=16):
    """A floor, two walls and light_count emissive octahedra floating above the floor, each inscribed in a spherical
    light (like the tessellated light spheres of the reference's scenes). Returns the scene and its lights."""
    rng = np.random.default_rng(seed)
    h = extent / 2
    quads = [[(-h, -h, 0), (h, -h, 0), (h, h, 0), (-h, h, 0)],
             [(-h, h, 0), (h, h, 0), (h, h, h), (-h, h, h)],
             [(-h, -h, 0), (-h, h, 0), (-h, h, h), (-h, -h, h)]]
    tris = []
    for q in quads:
        tris += [[q[0], q[1], q[2]], [q[0], q[2], q[3]]]
    materials = [1, 1, 2, 2, 1, 1]
    lights = np.zeros((light_count, 4), np.float32)
    lights[:, :2] = rng.uniform(-h * 0.9, h * 0.9, (light_count, 2))
    lights[:, 2] = rng.uniform(1.0, h * 0.8, light_count)
    lights[:, 3] = rng.uniform(0.08, 0.3, light_count)
    axes = np.concatenate([np.eye(3), -np.eye(3)])
    faces = [(0, 1, 2), (1, 3, 2), (3, 4, 2), (4, 0, 2), (1, 0, 5), (3, 1, 5), (4, 3, 5), (0, 4, 5)]
    for x, y, z, r in lights:
        corners = np.array([x, y, z]) + 0.999 * r * axes
        tris += [[corners[a], corners[b], corners[c]] for a, b, c in faces]
        materials += [0] * 8
    verts = np.array(tris, np.float64)
    positions, factor, summand = quantize_positions(verts.reshape(-1, 3))
    geometric = np.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0])
    geometric /= np.maximum(np.linalg.norm(geometric, axis=1, keepdims=True), 1e-12)
    # normals towards the room and outwards from the octahedra respectively
    geometric[:6] *= np.sign((geometric[:6] * (np.array([0, 0, 1.0]) - verts[:6].mean(axis=1))).sum(axis=1, keepdims=True))
    uv = rng.integers(0, 65536, (3 * len(tris), 2)).astype(np.uint16)
    normals_uvs = np.concatenate([encode_octahedral(np.repeat(geometric, 3, axis=0)), uv], axis=1).astype(np.uint16)
    textures = []
    for _ in range(3):
        textures += [random_texture(rng, texture_size, 132), random_texture(rng, texture_size, 131), random_texture(rng, texture_size, 141)]
    scene = formats.Scene(["_emission", "floor", "wall"], len(tris), factor, summand, positions, normals_uvs, np.array(materials, np.uint8), textures)
    return scene, lights


def look_at_spec(position, angles, frame_index=100, sky=(0.2, 0.4, 1.0), sky_strength=1.0, emission_strength=10.0, camera_type=0):
    spec = formats.default_scene_spec()
    spec["position"] = position
    spec["angles"] = angles
    spec["frame_index"] = frame_index
    spec["sky_color"] = sky
    spec["sky_strength"] = sky_strength
    spec["emission_material_strength"] = emission_strength
    spec["type"] = camera_type
    return spec


def constant_texture(value, fmt, size=4):
    """A texture of one colour (linear values in [0, 1]; two channels for format 141), written by the authoring tool
    of this repository and read back by the numpy parser."""
    import os
    import tempfile
    from path_tracer_amd.tools import vkt
    image = np.broadcast_to(np.asarray(value, np.float64), (size, size, len(value))).copy()
    with tempfile.TemporaryDirectory() as directory:
        path = os.path.join(directory, "constant.vkt")
        vkt.write_vkt(path, image, fmt)
        return formats.load_vkt(path)


def furnace_scene(base_color=(1.0, 1.0, 1.0), roughness=0.5, metalicity=0.0, walls=False, half_extent=100.0):
    """A floor in the plane z = 0 (normals +z) of one material with constant textures and no emitter: under a sky of
    unit radiance a convex surface reflects exactly its directional albedo. With walls: a box of the same material,
    open at the top, inside which light bounces several times. The specular texture stores (unused, square root of
    the roughness, metalicity) as shading_data.glsl:96-102 reads it, the normal texture the flat normal."""
    h = half_extent
    quads = [[(-h, -h, 0), (h, -h, 0), (h, h, 0), (-h, h, 0)]]
    if walls:
        w, t = 1.0, 1.5
        quads = [[(-w, -w, 0), (w, -w, 0), (w, w, 0), (-w, w, 0)],
                 [(-w, -w, 0), (-w, w, 0), (-w, w, t), (-w, -w, t)], [(w, w, 0), (w, -w, 0), (w, -w, t), (w, w, t)],
                 [(w, -w, 0), (-w, -w, 0), (-w, -w, t), (w, -w, t)], [(-w, w, 0), (w, w, 0), (w, w, t), (-w, w, t)]]
    tris = []
    for q in quads:
        tris += [[q[0], q[1], q[2]], [q[0], q[2], q[3]]]
    verts = np.array(tris, np.float64)
    positions, factor, summand = quantize_positions(verts.reshape(-1, 3))
    geometric = np.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0])
    geometric /= np.linalg.norm(geometric, axis=1, keepdims=True)
    uv = np.tile(np.array([[0, 0], [4096, 0], [4096, 4096]], np.uint16), (len(tris), 1))
    normals_uvs = np.concatenate([encode_octahedral(np.repeat(geometric, 3, axis=0)), uv], axis=1).astype(np.uint16)
    surface = [constant_texture(base_color, 132), constant_texture((0.0, np.sqrt(roughness), metalicity), 131), constant_texture((0.5, 0.5), 141)]
    unused = [constant_texture((0.0, 0.0, 0.0), 132), constant_texture((0.0, 1.0, 0.0), 131), constant_texture((0.5, 0.5), 141)]
    return formats.Scene(["_emission", "surface"], len(tris), factor, summand, positions, normals_uvs, np.ones(len(tris), np.uint8), unused + surface)
