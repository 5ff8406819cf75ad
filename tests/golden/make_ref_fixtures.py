"""Writes tests/golden/ref/: outputs of the reference's own C code on committed inputs.

Three parts of the reference compile in the authoring container from their own sources with plain gcc
(`make -C oracle ref`, outputs in the git-ignored oracle/_ref/): src/math_utilities.c, the texture converter
tools/texture_conversion/main.c (with its vendored stb_dxt.h and stb_image.h) and src/stb_image_write.c. This script
runs them and stores what they return, so that the tests can compare the host library, the oracle and the authoring
tools with results *of the reference itself* on machines that have neither the reference nor these binaries:

  ref/math_utilities.npz   inputs and outputs of mat_mat_mul, normalize, rotation_matrix_from_angles, invert_mat4
                           (src/math_utilities.c:16-64,197-203) on random inputs and on the angles of every quicksave
  ref/images/*.png         small source images (written here, seeded)
  ref/vkt/*.vkt            what the reference's converter makes of them: BC1 UNORM/sRGB (131/132), BC5 (141) with full
                           Gaussian mip chains (main.c:349-402), and uncompressed RGBA8 UNORM/sRGB (37/43), which show
                           the reference's mip levels without a block encoder in between
  ref/stb/*                what the reference's stb_image_write makes of a small image (PNG, JPG quality 90, HDR), called
                           as src/main.c:2053,2076-2077 call it, plus the pixels that went in

Needs /root/reference and pillow; run once in the authoring container: python tests/golden/make_ref_fixtures.py
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_BUILD = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(HERE, "ref")

# (file name, width, height): power-of-two extents as the converter demands, one not square, one 1x1 (which the
# converter widens to a constant 4x4 block, main.c:295-307)
IMAGES = (("blobs", 64, 64), ("stripes", 64, 16), ("noise", 16, 16), ("flat", 8, 8), ("pixel", 1, 1))
FORMATS = (131, 132, 141, 37, 43)


def ptr(array):
    return array.ctypes.data_as(C.c_void_p)


def make_image(name, width, height):
    """Seeded RGB test images with smooth parts (mips matter), edges (block endpoints matter) and noise."""
    rng = np.random.default_rng(sum(name.encode()))
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    u, v = (x + 0.5) / width, (y + 0.5) / height
    if name == "blobs":
        r = 0.5 + 0.5 * np.sin(2 * np.pi * (u + 0.15 * np.sin(2 * np.pi * v)))
        g = 0.5 + 0.5 * np.cos(2 * np.pi * (v + 0.5 * u))
        b = 0.1 + 0.8 / (1.0 + np.exp(40.0 * (np.sqrt((u - 0.5) ** 2 + (v - 0.5) ** 2) - 0.3)))
    elif name == "stripes":
        r = (np.floor(u * 8) % 2) * 0.9 + 0.05
        g = v
        b = 0.5 + 0.5 * np.sin(2 * np.pi * 3 * u) * np.cos(2 * np.pi * v)
    elif name == "noise":
        r, g, b = rng.uniform(0, 1, (3, height, width))
    elif name == "flat":
        r, g, b = np.full((height, width), 0.737), np.full((height, width), 0.5), np.full((height, width), 0.2)
    else:
        r, g, b = np.full((height, width), 0.31), np.full((height, width), 0.62), np.full((height, width), 0.93)
    return np.clip(np.stack([r, g, b], -1) * 255.0 + 0.5, 0, 255).astype(np.uint8)


def math_fixtures():
    lib = C.CDLL(os.path.join(REF_BUILD, "libref_math.so"))
    lib.normalize.restype = C.c_bool
    rng = np.random.default_rng(20)
    out = {}
    # mat_mat_mul: the shapes the host uses (3x3 by 3x3, 1x3 by 3x3, 4x4 by 4x4) and an odd one
    for index, (m, k, n) in enumerate(((3, 3, 3), (1, 3, 3), (4, 4, 4), (2, 5, 3))):
        lhs = rng.normal(size=(16, m, k)).astype(np.float32)
        rhs = rng.normal(size=(16, k, n)).astype(np.float32)
        product = np.zeros((16, m, n), np.float32)
        for i in range(16):
            lib.mat_mat_mul(ptr(product[i]), ptr(lhs[i]), ptr(rhs[i]), m, k, n)
        out["mul%d_lhs" % index], out["mul%d_rhs" % index], out["mul%d_out" % index] = lhs, rhs, product
    # normalize, with a zero vector (returns false and leaves it alone)
    vectors = (rng.normal(size=(32, 3)) * 10.0 ** rng.uniform(-20, 15, (32, 1))).astype(np.float32)
    vectors[5] = 0.0
    normalized = vectors.copy()
    flags = np.zeros(32, np.uint8)
    for i in range(32):
        flags[i] = lib.normalize(ptr(normalized[i]), 3)
    out["normalize_in"], out["normalize_out"], out["normalize_ok"] = vectors, normalized, flags
    # rotation_matrix_from_angles: random angles and those of every quicksave the reference ships
    angles = [rng.uniform(-7.0, 7.0, 3).astype(np.float32) for _ in range(64)]
    saves = os.path.join(HERE, "data", "saves")
    for root, _, files in sorted(os.walk(saves)):
        for name in sorted(files):
            if name.endswith(".rt_save"):
                record = np.fromfile(os.path.join(root, name), np.uint8)
                angles.append(record[4:16].view(np.float32).copy())
    angles = np.array(angles, np.float32)
    rotations = np.zeros((len(angles), 9), np.float32)
    for i in range(len(angles)):
        lib.rotation_matrix_from_angles(ptr(rotations[i]), ptr(angles[i]))
    out["angles"], out["rotations"] = angles, rotations
    # invert_mat4: random matrices, ill-scaled ones, and projection-like matrices (zeros in the places a
    # perspective matrix has them, so that sums of products that are exactly zero are covered)
    matrices = rng.normal(size=(48, 4, 4)).astype(np.float32)
    matrices[16:32] *= (10.0 ** rng.uniform(-3, 3, (16, 4, 1))).astype(np.float32)
    for i in range(32, 48):
        projection = np.zeros((4, 4), np.float32)
        near, far, top = rng.uniform(0.01, 1.0), rng.uniform(10.0, 1e4), np.tan(rng.uniform(0.2, 1.4))
        projection[0, 0], projection[1, 1] = -1.0 / (top * 16 / 9), 1.0 / top
        projection[2, 2], projection[2, 3], projection[3, 2] = (far + near) / (near - far), 2 * far * near / (near - far), -1.0
        view = np.eye(4, dtype=np.float32)
        view[:3, :3] = rotations[i - 32].reshape(3, 3).T
        view[:3, 3] = rng.normal(size=3) * 10
        matrices[i] = (projection.astype(np.float32) @ view).astype(np.float32)
    inverses = np.zeros_like(matrices)
    for i in range(len(matrices)):
        source = matrices[i].copy()
        lib.invert_mat4(ptr(inverses[i]), ptr(source))
    out["matrices"], out["inverses"] = matrices, inverses
    np.savez(os.path.join(OUT, "math_utilities.npz"), **out)


def texture_fixtures():
    from PIL import Image
    os.makedirs(os.path.join(OUT, "images"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "vkt"), exist_ok=True)
    converter = os.path.join(REF_BUILD, "texture_conversion")
    for name, width, height in IMAGES:
        png = os.path.join(OUT, "images", name + ".png")
        Image.fromarray(make_image(name, width, height), "RGB").save(png)
        for vk_format in FORMATS:
            vkt = os.path.join(OUT, "vkt", "%s_%d.vkt" % (name, vk_format))
            subprocess.check_call([converter, str(vk_format), png, vkt], stdout=subprocess.DEVNULL)


def stb_fixtures():
    os.makedirs(os.path.join(OUT, "stb"), exist_ok=True)
    lib = C.CDLL(os.path.join(REF_BUILD, "libref_stb_image_write.so"))
    width, height = 37, 21
    # smooth in colour (stb_image_write halves the chroma resolution at the reference's quality of 90), with a
    # little luminance texture
    ldr = make_image("blobs", 256, 256)[40:40 + height, 60:60 + width].astype(np.int32)
    ldr += (make_image("noise", 64, 32)[:height, :width, :1].astype(np.int32) - 128) // 8
    ldr = np.clip(ldr, 0, 255).astype(np.uint8)
    rng = np.random.default_rng(9)
    hdr = (rng.uniform(0, 1, (height, width, 3)) * 10.0 ** rng.uniform(-4, 4, (height, width, 1))).astype(np.float32)
    hdr[0, 0] = 0.0
    hdr[1, :] = 0.25  # a run, which the run-length coder of the format picks up
    np.save(os.path.join(OUT, "stb", "ldr_pixels.npy"), ldr)
    np.save(os.path.join(OUT, "stb", "hdr_pixels.npy"), hdr)
    stb = os.path.join(OUT, "stb")
    assert lib.stbi_write_png(os.fsencode(os.path.join(stb, "image.png")), width, height, 3, ptr(ldr), width * 3)
    assert lib.stbi_write_jpg(os.fsencode(os.path.join(stb, "image.jpg")), width, height, 3, ptr(ldr), 90)
    assert lib.stbi_write_hdr(os.fsencode(os.path.join(stb, "image.hdr")), width, height, 3, ptr(hdr))


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("needs the reference under /root/reference")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "-s"])
    os.makedirs(OUT, exist_ok=True)
    math_fixtures()
    texture_fixtures()
    stb_fixtures()
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
