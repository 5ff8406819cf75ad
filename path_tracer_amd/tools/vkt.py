"""Writer for *.vkt texture files with BC1 / BC5 block compression and Gaussian-filtered mipmaps.

Follows the reference's converter tools/texture_conversion/main.c: header and mipmap headers (:323-343), mipmaps
down to 4x4 filtered from the full-resolution image with a Gaussian of standard deviation 0.4 * 2^level and
wrap-around (:349-402), base colour stored sRGB-encoded as BC1 (VkFormat 132), specular as BC1 UNORM (131), normals
as BC5 (141). The block encoders are simple range fits (the reference calls stb_dxt).
"""
import struct

import numpy as np

FORMAT_BC1_UNORM, FORMAT_BC1_SRGB, FORMAT_BC5_UNORM = 131, 132, 141


def linear_to_srgb8(x):
    x = np.clip(x, 0.0, 1.0)
    srgb = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1.0 / 2.4) - 0.055)
    return (srgb * 255.0 + 0.5).astype(np.uint8)


def quantize_linear8(x):
    return (np.clip(x, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)


def _blocks(image):
    """(H, W, C) -> (H/4 * W/4, 16, C) in block row-major order, texel index 4 * y + x."""
    h, w, c = image.shape
    return image.reshape(h // 4, 4, w // 4, 4, c).transpose(0, 2, 1, 3, 4).reshape(-1, 16, c)


def encode_bc1(rgb8):
    """rgb8: uint8 (H, W, 3). Returns the BC1 payload (8 bytes per 4x4 block, four-colour mode)."""
    blocks = _blocks(rgb8).astype(np.int32)
    hi, lo = blocks.max(axis=1), blocks.min(axis=1)

    def to565(c):
        r, g, b = (c[:, 0] * 31 + 127) // 255, (c[:, 1] * 63 + 127) // 255, (c[:, 2] * 31 + 127) // 255
        return (r << 11) | (g << 5) | b

    def from565(v):
        r, g, b = (v >> 11) & 31, (v >> 5) & 63, v & 31
        return np.stack([(r << 3) | (r >> 2), (g << 2) | (g >> 4), (b << 3) | (b >> 2)], axis=1)

    c0, c1 = to565(hi), to565(lo)
    swap = c0 < c1
    c0, c1 = np.where(swap, c1, c0), np.where(swap, c0, c1)
    e0, e1 = from565(c0), from565(c1)
    four = (c0 > c1)[:, None]
    palette = np.stack([e0, e1, np.where(four, (2 * e0 + e1) // 3, (e0 + e1) // 2), np.where(four, (e0 + 2 * e1) // 3, 0)], axis=1)
    distance = ((blocks[:, :, None, :] - palette[:, None, :, :]) ** 2).sum(axis=3)
    # with equal endpoints the block is in three-colour mode where code 3 is black: never pick it there
    distance[:, :, 3] = np.where(four, distance[:, :, 3], 1 << 30)
    codes = distance.argmin(axis=2).astype(np.uint32)
    bits = (codes << (2 * np.arange(16, dtype=np.uint32))[None, :]).sum(axis=1).astype(np.uint32)
    out = np.zeros((len(blocks), 2), np.uint32)
    out[:, 0] = (c0 | (c1 << 16)).astype(np.uint32)
    out[:, 1] = bits
    return out.tobytes()


def _encode_bc4(values):
    """values: int (N, 16) in 0..255 -> uint64 per block (eight-value mode when the block is not constant)."""
    a0, a1 = values.max(axis=1), values.min(axis=1)
    eight = (a0 > a1)[:, None]
    steps = np.arange(1, 7)[None, :]
    interpolated8 = ((7 - steps) * a0[:, None] + steps * a1[:, None]) // 7
    palette = np.concatenate([a0[:, None], a1[:, None], interpolated8], axis=1)
    distance = np.abs(values[:, :, None] - palette[:, None, :])
    codes = np.where(eight, distance.argmin(axis=2), 0).astype(np.uint64)
    bits = (codes << (3 * np.arange(16, dtype=np.uint64))[None, :]).sum(axis=1)
    return a0.astype(np.uint64) | (a1.astype(np.uint64) << 8) | (bits << 16)


def encode_bc5(rg8):
    blocks = _blocks(rg8).astype(np.int64)
    out = np.zeros((len(blocks), 2), np.uint64)
    out[:, 0] = _encode_bc4(blocks[:, :, 0])
    out[:, 1] = _encode_bc4(blocks[:, :, 1])
    return out.tobytes()


def gaussian_mipmap(image, level):
    """Mipmap `level` of a float image (H, W, C), filtered from level 0 like tools/texture_conversion/main.c:356-400."""
    if level == 0:
        return image
    scale = 1 << level
    standard_deviation = 0.4 * scale
    extent = int(np.ceil(3.0 * standard_deviation))
    taps = np.arange(2 * extent) - (extent - 0.5)
    weights = np.exp(-0.5 / standard_deviation ** 2 * taps ** 2)
    weights /= weights.sum()
    offset = scale // 2 - extent
    h, w, _ = image.shape
    rows = (np.arange(h // scale)[:, None] * scale + offset + np.arange(2 * extent)[None, :]) & (h - 1)
    cols = (np.arange(w // scale)[:, None] * scale + offset + np.arange(2 * extent)[None, :]) & (w - 1)
    vertical = np.einsum("yk,ykxc->yxc", np.broadcast_to(weights, rows.shape), image[rows])
    return np.einsum("xk,yxkc->yxc", np.broadcast_to(weights, cols.shape), vertical[:, cols])


def write_vkt(path, image, vk_format):
    """image: float linear (H, W, 3) for BC1 formats, (H, W, 2) for BC5; power-of-two extent >= 4."""
    image = np.asarray(image, np.float64)
    h, w, _ = image.shape
    if h < 4 or w < 4 or (h & (h - 1)) or (w & (w - 1)):
        raise ValueError("texture extent %dx%d must be a power of two and at least 4x4" % (w, h))
    mip_count = max(1, min(int(np.log2(w)), int(np.log2(h))) - 1)
    payloads = []
    for level in range(mip_count):
        mip = gaussian_mipmap(image, level)
        if vk_format == FORMAT_BC1_SRGB:
            payloads.append(encode_bc1(linear_to_srgb8(mip)))
        elif vk_format == FORMAT_BC1_UNORM:
            payloads.append(encode_bc1(quantize_linear8(mip)))
        elif vk_format == FORMAT_BC5_UNORM:
            payloads.append(encode_bc5(quantize_linear8(mip)))
        else:
            raise ValueError("unsupported VkFormat %d" % vk_format)
    with open(path, "wb") as file:
        file.write(struct.pack("<IIIIIIQ", 0xbc1bc1, 1, mip_count, w, h, vk_format, sum(len(p) for p in payloads)))
        offset = 0
        for level, payload in enumerate(payloads):
            file.write(struct.pack("<IIQQ", w >> level, h >> level, len(payload), offset))
            offset += len(payload)
        for payload in payloads:
            file.write(payload)
        file.write(struct.pack("<I", 0xe0fe0f))
