"""Writer for *.vks scene files, *.lights files and *.rt_save quicksaves.

Follows the reference's exporter tools/io_export_vulkan_blender28.py:458-530: triangles sorted by the 30-bit
Morton code of their centroid, positions quantised to 21 bits per coordinate with dequantisation constants that
reproduce cell centres, octahedral normals in 2x16 bits with 32768 <-> 0, texture coordinates in 2x16 bits
covering [0, 8) per triangle, one material byte per triangle.
"""
import struct

import numpy as np


def _part_1_by_2(x):
    x = x & 0x3ff
    x = (x ^ (x << 16)) & 0xff0000ff
    x = (x ^ (x << 8)) & 0x0300f00f
    x = (x ^ (x << 4)) & 0x030c30c3
    x = (x ^ (x << 2)) & 0x09249249
    return x


def morton_codes(points):
    """30-bit Morton codes of points within their bounding box (io_export_vulkan_blender28.py:66-77)."""
    lo, hi = points.min(axis=0), points.max(axis=0)
    scale = 1024.0 / np.maximum(hi - lo, 1e-30)
    q = np.clip((points - lo) * scale, 0.0, 1023.0).astype(np.uint32)
    return (_part_1_by_2(q[:, 0]) | (_part_1_by_2(q[:, 1]) << 1) | (_part_1_by_2(q[:, 2]) << 2)).astype(np.uint32)


def encode_normals(normals):
    """Octahedral map to 2 x 16 bits (io_export_vulkan_blender28.py:24-46)."""
    normals = np.asarray(normals, np.float64)
    octahedral = normals[:, 0:2] / np.abs(normals).sum(axis=1, keepdims=True)
    sign_not_zero = np.where(octahedral >= 0.0, 1.0, -1.0)
    folded = (1.0 - np.abs(octahedral[:, ::-1])) * sign_not_zero
    octahedral = np.where(normals[:, 2:3] <= 0.0, folded, octahedral)
    return (octahedral * 32767.0 + 32768.5).astype(np.uint16)


def write_vks(path, material_names, positions, normals, uvs, material_indices, sort_triangles=True):
    """positions, normals: float arrays (T, 3, 3); uvs: (T, 3, 2); material_indices: (T,) ints < 256."""
    positions = np.asarray(positions, np.float32).reshape(-1, 3, 3)
    normals = np.asarray(normals, np.float32).reshape(-1, 3, 3)
    uvs = np.asarray(uvs, np.float32).reshape(-1, 3, 2).copy()
    material_indices = np.asarray(material_indices, np.uint8).reshape(-1)
    triangle_count = len(positions)
    if sort_triangles and triangle_count:
        order = np.argsort(morton_codes(positions.mean(axis=1)), kind="stable")
        positions, normals, uvs, material_indices = positions[order], normals[order], uvs[order], material_indices[order]
    flat = positions.reshape(-1, 3).astype(np.float64)
    box_min = flat.min(axis=0) if triangle_count else np.zeros(3)
    box_max = flat.max(axis=0) if triangle_count else np.ones(3)
    quantization_factor = 2.0 ** 21 / np.maximum(box_max - box_min, 1e-30)
    quantized = np.minimum(2 ** 21 - 1, (flat * quantization_factor - box_min * quantization_factor).astype(np.uint32)).astype(np.uint64)
    dequantization_factor = (1.0 / quantization_factor).astype(np.float32)
    dequantization_summand = (box_min + 0.5 / quantization_factor).astype(np.float32)
    packed = np.zeros((len(flat), 2), np.uint32)
    packed[:, 0] = (quantized[:, 0] | ((quantized[:, 1] & 0x7ff) << 21)).astype(np.uint32)
    packed[:, 1] = (((quantized[:, 1] & 0x1ff800) >> 11) | (quantized[:, 2] << 10)).astype(np.uint32)
    uvs -= np.floor(uvs.min(axis=1))[:, np.newaxis, :]
    normal_and_uv = np.zeros((len(flat), 4), np.uint16)
    normal_and_uv[:, 0:2] = encode_normals(normals.reshape(-1, 3))
    normal_and_uv[:, 2:4] = np.clip(uvs.reshape(-1, 2) * (65535.0 / 8.0) + 0.5, 0.0, 65535.0).astype(np.uint16)
    with open(path, "wb") as file:
        file.write(struct.pack("<II", 0xabcabc, 1))
        file.write(struct.pack("<QQ", len(material_names), triangle_count))
        file.write(dequantization_factor.tobytes())
        file.write(dequantization_summand.tobytes())
        for name in material_names:
            encoded = name.encode("utf-8")
            file.write(struct.pack("<Q", len(encoded)) + encoded + b"\0")
        file.write(packed.tobytes())
        file.write(normal_and_uv.tobytes())
        file.write(material_indices.tobytes())
        file.write(struct.pack("<I", 0xe0fe0f))
    return dequantization_factor, dequantization_summand


def write_lights(path, lights):
    """u32 count + count x (x, y, z, radius), as read at src/main.c:687-698 of the reference."""
    lights = np.asarray(lights, np.float32).reshape(-1, 4)
    with open(path, "wb") as file:
        file.write(struct.pack("<I", len(lights)))
        file.write(lights.tobytes())


def write_rt_save(path, position, angles, scene_file=1, frame_index=0, camera_type=0, fov=None, exposure=1.0,
                  sky_color=(0.2, 0.4, 1.0), sky_strength=0.0, emission_color=(1.0, 0.9, 0.6), emission_strength=20.0,
                  near=0.01, far=1.0e4, height=10.0, tonemapper=1):
    """The 128-byte scene_spec_t of src/main.h:55-78."""
    if fov is None:
        fov = float(np.float32(np.pi * float(np.float32(0.4))))
    record = struct.pack("<i3fB3x3f3ffffiffifI3ff3ff4f", scene_file, *angles, 0, 0.0, 0.0, 0.0, *position, 2.0, near, far,
                         camera_type, fov, height, tonemapper, exposure, frame_index, *sky_color, sky_strength,
                         *emission_color, emission_strength, 0.0, 0.0, 0.0, 0.0)
    assert len(record) == 128
    with open(path, "wb") as file:
        file.write(record)
