"""numpy parsers for the reference's binary file formats (test infrastructure).

These are independent of the product's C loaders so that the two can be checked
against each other. Layouts follow the reference's readers:
.vks  src/scene.c:293-367, .vkt src/textures.c:63-106, .lights src/main.c:687-698,
.rt_save src/main.c:81-109 with scene_spec_t at src/main.h:55-78.
"""
import os
import struct
from dataclasses import dataclass, field
from typing import List

import numpy as np


@dataclass
class Texture:
    width: int
    height: int
    format: int
    mip_count: int
    mip_extents: List[tuple]
    mip_offsets: np.ndarray  # uint64 byte offsets into payload
    payload: np.ndarray      # uint8


@dataclass
class Scene:
    material_names: List[str]
    triangle_count: int
    dequantization_factor: np.ndarray  # float32[3]
    dequantization_summand: np.ndarray  # float32[3]
    positions: np.ndarray       # uint32[3T, 2]
    normals_uvs: np.ndarray     # uint16[3T, 4]
    materials: np.ndarray       # uint8[T]
    textures: List[Texture] = field(default_factory=list)

    @property
    def emission_material_index(self) -> int:
        # src/main.c:936-939: index of "_emission", 0 if there is none
        index = 0
        for i, name in enumerate(self.material_names):
            if name == "_emission":
                index = i
        return index


def load_vks(path: str) -> Scene:
    data = open(path, "rb").read()
    marker, version = struct.unpack_from("<II", data, 0)
    if marker != 0xabcabc:
        raise ValueError("not a .vks file: marker %#x" % marker)
    if version != 1:
        raise ValueError("unsupported .vks version %d" % version)
    material_count, triangle_count = struct.unpack_from("<QQ", data, 8)
    factor = np.frombuffer(data, np.float32, 3, 24).copy()
    summand = np.frombuffer(data, np.float32, 3, 36).copy()
    offset = 48
    names = []
    for _ in range(material_count):
        length, = struct.unpack_from("<Q", data, offset)
        offset += 8
        names.append(data[offset:offset + length].decode("utf-8"))
        offset += length + 1
    positions = np.frombuffer(data, np.uint32, triangle_count * 6, offset).reshape(-1, 2).copy()
    offset += triangle_count * 24
    normals_uvs = np.frombuffer(data, np.uint16, triangle_count * 12, offset).reshape(-1, 4).copy()
    offset += triangle_count * 24
    materials = np.frombuffer(data, np.uint8, triangle_count, offset).copy()
    offset += triangle_count
    eof, = struct.unpack_from("<I", data, offset)
    if eof != 0xe0fe0f:
        raise ValueError("missing end-of-file marker in .vks file")
    return Scene(names, triangle_count, factor, summand, positions, normals_uvs, materials)


def load_vkt(path: str) -> Texture:
    data = open(path, "rb").read()
    marker, version, mip_count, width, height, fmt, size = struct.unpack_from("<IIIIIIQ", data, 0)
    if marker != 0xbc1bc1:
        raise ValueError("not a .vkt file: marker %#x" % marker)
    if version != 1:
        raise ValueError("unsupported .vkt version %d" % version)
    offset = 32
    extents, offsets = [], []
    for _ in range(mip_count):
        w, h, mip_size, mip_offset = struct.unpack_from("<IIQQ", data, offset)
        offset += 24
        extents.append((w, h, mip_size))
        offsets.append(mip_offset)
    payload = np.frombuffer(data, np.uint8, size, offset).copy()
    eof, = struct.unpack_from("<I", data, offset + size)
    if eof != 0xe0fe0f:
        raise ValueError("missing end-of-file marker in .vkt file")
    return Texture(width, height, fmt, mip_count, extents, np.array(offsets, np.uint64), payload)


def load_scene(vks_path: str, texture_dir: str) -> Scene:
    """load_scene() of src/scene.c:282-404: mesh plus three textures per material."""
    scene = load_vks(vks_path)
    for name in scene.material_names:
        for suffix in ("_BaseColor.vkt", "_Specular.vkt", "_Normal.vkt"):
            scene.textures.append(load_vkt(os.path.join(texture_dir, name + suffix)))
    return scene


def load_lights(path: str, max_count=None) -> np.ndarray:
    data = open(path, "rb").read()
    count, = struct.unpack_from("<I", data, 0)
    if max_count is not None:
        count = min(count, max_count)
    count = min(count, (len(data) - 4) // 16)
    return np.frombuffer(data, np.float32, count * 4, 4).reshape(-1, 4).copy()


RT_SAVE_DTYPE = np.dtype([
    ("scene_file", "<i4"),
    ("angles", "<f4", 3), ("mouse_active", "u1"), ("pad", "u1", 3), ("origin_angles", "<f4", 3),
    ("position", "<f4", 3), ("speed", "<f4"), ("near", "<f4"), ("far", "<f4"),
    ("type", "<i4"), ("fov", "<f4"), ("height", "<f4"),
    ("tonemapper", "<i4"), ("exposure", "<f4"), ("frame_index", "<u4"),
    ("sky_color", "<f4", 3), ("sky_strength", "<f4"),
    ("emission_material_color", "<f4", 3), ("emission_material_strength", "<f4"),
    ("params", "<f4", 4),
])
assert RT_SAVE_DTYPE.itemsize == 128


def load_rt_save(path: str) -> np.ndarray:
    data = open(path, "rb").read()
    if len(data) < 128:
        raise ValueError("quicksave too short")
    return np.frombuffer(data, RT_SAVE_DTYPE, 1, 0).copy()


def default_scene_spec() -> np.ndarray:
    """The Cornell box default of init_scene_spec(), src/main.c:120-146."""
    spec = np.zeros(1, RT_SAVE_DTYPE)
    spec["scene_file"] = 1
    spec["angles"] = (1.570796, 0.0, 0.0)
    spec["position"] = (0.278, 0.379, 0.2744)
    spec["speed"] = 2.0
    spec["near"] = 0.01
    spec["far"] = 1.0e4
    spec["fov"] = np.float32(np.pi * float(np.float32(0.4)))  # M_PI * 0.4f is a double product
    spec["height"] = 10.0
    spec["tonemapper"] = 1
    spec["exposure"] = 1.0
    spec["sky_color"] = (0.2, 0.4, 1.0)
    spec["emission_material_color"] = (1.0, 0.9, 0.6)
    spec["emission_material_strength"] = 20.0
    return spec


def dequantize_positions(scene: Scene) -> np.ndarray:
    """World-space vertex positions (3 * triangle_count, 3), float32, with the arithmetic of src/scene.c:43-55:
    21 bits per coordinate out of two words, multiplied by the factor, then the summand added."""
    words = scene.positions.astype(np.uint32)
    quantized = np.stack([words[:, 0] & 0x1fffff, ((words[:, 0] & 0xffe00000) >> 21) | ((words[:, 1] & 0x3ff) << 11),
                          (words[:, 1] & 0x7ffffc00) >> 10], axis=1).astype(np.float32)
    return quantized * scene.dequantization_factor.astype(np.float32) + scene.dequantization_summand.astype(np.float32)


def decode_normals(scene: Scene) -> np.ndarray:
    """Unit vertex normals (3 * triangle_count, 3) from the octahedral map, as dequantize_normal() at
    src/shaders/mesh_quantization.glsl:16-34 decodes them: 32768 is zero, 1 is -1 (float64 here: for checks with a
    tolerance, not bit-exact)."""
    octahedral = (scene.normals_uvs[:, 0:2].astype(np.float64) - 32768.0) / 32767.0
    normal = np.concatenate([octahedral, (1.0 - np.abs(octahedral[:, 0]) - np.abs(octahedral[:, 1]))[:, None]], axis=1)
    folded = (1.0 - np.abs(normal[:, 1::-1])) * np.where(normal[:, 0:2] >= 0.0, 1.0, -1.0)
    normal[:, 0:2] = np.where(normal[:, 2:3] < 0.0, folded, normal[:, 0:2])
    return normal / np.linalg.norm(normal, axis=1, keepdims=True)
