// Per-path-vertex work of the wavefront path tracer: random numbers, camera
// rays, mesh and texture decode, the Frostbite BRDF with its samplers, and
// spherical-light sampling. Each function states the lines of the reference
// shader it implements (paths relative to src/shaders/).
#pragma once
#include "pt_common.h"

// ---- pathtrace.frag.glsl:23-34, PCG2D ----
PT_FN v2 pt_random(uint32_t& sx, uint32_t& sy) {
	sx = 1664525u * sx + 1013904223u;
	sy = 1664525u * sy + 1013904223u;
	sx += 1664525u * sy;
	sy += 1664525u * sx;
	sx ^= sx >> 16;
	sy ^= sy >> 16;
	sx += 1664525u * sy;
	sy += 1664525u * sx;
	sx ^= sx >> 16;
	sy ^= sy >> 16;
	return mk2((float) sx * 2.32830643654e-10f, (float) sy * 2.32830643654e-10f);
}

// ---- pathtrace.frag.glsl:38-45 ----
PT_FN float pt_erfinv(float x) {
	float w = -pt_log(fmaxf(1.0e-37f, 1.0f - x * x));
	float a = w - 2.5f;
	float b = sqrtf(w) - 3.0f;
	float pa = fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(2.81022636e-08f, a, 3.43273939e-07f), a, -3.5233877e-06f), a, -4.39150654e-06f), a, 0.00021858087f), a, -0.00125372503f), a, -0.00417768164f), a, 0.246640727f), a, 1.50140941f);
	float pb = fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(-0.000200214257f, b, 0.000100950558f), b, 0.00134934322f), b, -0.00367342844f), b, 0.00573950773f), b, -0.0076224613f), b, 0.00943887047f), b, 1.00167406f), b, 2.83297682f);
	return x * ((w < 5.0f) ? pa : pb);
}

struct PtFrame { v3 b1, b2, n; };

// ---- brdfs.glsl:126-134 ----
PT_FN PtFrame pt_shading_space(v3 n) {
	float s = (n.z > 0.0f) ? 1.0f : -1.0f;
	float a = -1.0f / (s + n.z);
	float b = n.x * n.y * a;
	PtFrame f;
	f.b1 = mk3(1.0f + s * n.x * n.x * a, s * b, -s * n.x);
	f.b2 = mk3(b, s + n.y * n.y * a, -n.y);
	f.n = n;
	return f;
}
PT_FN v3 pt_to_world(const PtFrame& f, v3 v) { return (f.b1 * v.x + f.b2 * v.y) + f.n * v.z; }
PT_FN v3 pt_to_local(const PtFrame& f, v3 v) { return mk3(dot(f.b1, v), dot(f.b2, v), dot(f.n, v)); }

// ---- pathtrace.frag.glsl:50-55 ----
PT_FN v3 pt_sample_hemisphere_spherical(v2 randoms) {
	float azimuth = (2.0f * PT_PI) * randoms.x - PT_PI;
	float inclination = (0.5f * PT_PI) * randoms.y;
	float si, ci, sa, ca;
	pt_sincos(inclination, &si, &ci);
	pt_sincos(azimuth, &sa, &ca);
	return mk3(si * ca, si * sa, ci);
}

// ---- pathtrace.frag.glsl:60-64 ----
PT_FN float pt_hemisphere_spherical_density(float z) {
	if (z < 0.0f)
		return 0.0f;
	return 1.0f / ((PT_PI * PT_PI) * sqrtf(fmaxf(0.0f, 1.0f - z * z)));
}

// ---- camera_utilities.glsl:9-13 (row_major block: GLSL M * v is the plain
// row-major product) ----
PT_FN v3 pt_camera_ray_origin(v2 tc, const float* m) {
	float px = 2.0f * tc.x - 1.0f, py = 2.0f * tc.y - 1.0f;
	float r[4];
	for (int i = 0; i != 4; ++i)
		r[i] = ((m[4 * i + 0] * px + m[4 * i + 1] * py) + m[4 * i + 2] * 0.0f) + m[4 * i + 3] * 1.0f;
	return mk3(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
}

// ---- camera_utilities.glsl:24-44 (GLSL m[c][r] is element 4 * r + c) ----
PT_FN v3 pt_camera_ray_direction(v2 tc, const float* m) {
	float dx = 2.0f * tc.x - 1.0f, dy = 2.0f * tc.y - 1.0f;
	v3 d;
	d.x = (m[4 * 1 + 1] * m[4 * 3 + 2] - m[4 * 3 + 1] * m[4 * 1 + 2]) * dx;
	d.x += (m[4 * 3 + 1] * m[4 * 0 + 2] - m[4 * 0 + 1] * m[4 * 3 + 2]) * dy;
	d.x += m[4 * 0 + 1] * m[4 * 1 + 2] - m[4 * 1 + 1] * m[4 * 0 + 2];
	d.y = (m[4 * 3 + 0] * m[4 * 1 + 2] - m[4 * 1 + 0] * m[4 * 3 + 2]) * dx;
	d.y += (m[4 * 0 + 0] * m[4 * 3 + 2] - m[4 * 3 + 0] * m[4 * 0 + 2]) * dy;
	d.y += m[4 * 1 + 0] * m[4 * 0 + 2] - m[4 * 0 + 0] * m[4 * 1 + 2];
	d.z = (m[4 * 1 + 0] * m[4 * 3 + 1] - m[4 * 3 + 0] * m[4 * 1 + 1]) * dx;
	d.z += (m[4 * 3 + 0] * m[4 * 0 + 1] - m[4 * 0 + 0] * m[4 * 3 + 1]) * dy;
	d.z += m[4 * 0 + 0] * m[4 * 1 + 1] - m[4 * 1 + 0] * m[4 * 0 + 1];
	return normalize(d);
}

// ---- pathtrace.frag.glsl:365-386: seed, Gaussian pixel jitter, camera ray ----
// The seed of a path once its primary ray has been generated: the initial
// seed of pathtrace.frag.glsl:368 advanced by the one draw of pt_primary_ray()
PT_FN void pt_seed_after_primary_ray(uint32_t x, uint32_t y, uint32_t frame_index, uint32_t& sx, uint32_t& sy) {
	sx = x ^ (frame_index << 16);
	sy = y ^ ((frame_index + 237u) << 16);
	pt_random(sx, sy);
}

// The camera ray through the point (jx, jy) of the image plane in pixels (pathtrace.frag.glsl:374-386)
PT_FN void pt_camera_ray(const PtConstants& c, float jx, float jy, v3* origin, v3* dir) {
	if (c.camera_type <= 1) {
		v2 tc = mk2(jx * c.inv_viewport_size[0], jy * c.inv_viewport_size[1]);
		*origin = pt_camera_ray_origin(tc, c.projection_to_world_space);
		*dir = pt_camera_ray_direction(tc, c.world_to_projection_space);
	}
	else {
		PtFrame f = pt_shading_space(mk3(c.hemispherical_camera_normal[0], c.hemispherical_camera_normal[1], c.hemispherical_camera_normal[2]));
		*origin = mk3(c.camera_pos[0], c.camera_pos[1], c.camera_pos[2]);
		float factor_y = (c.camera_type == 3) ? 2.0f : 1.0f;
		*dir = pt_to_world(f, pt_sample_hemisphere_spherical(mk2(jx * 1.0f * c.inv_viewport_size[0], jy * factor_y * c.inv_viewport_size[1])));
	}
}

PT_FN float pt_length(v3 v) { return sqrtf(dot(v, v)); }

// Seed, jittered position and camera ray of pixel (x, y). The ray's cone (see
// "Ray cones" below) is what one pixel subtends: the rays one pixel to the
// right and one pixel down tell how far apart neighbouring rays start and how
// fast they diverge; the smaller of the two counts (the minor axis).
PT_FN void pt_primary_ray(const PtConstants& c, uint32_t x, uint32_t y, uint32_t frame_index, uint32_t& sx, uint32_t& sy, v3* origin, v3* dir, float* cone_width, float* cone_spread) {
	sx = x ^ (frame_index << 16);
	sy = y ^ ((frame_index + 237u) << 16);
	v2 randoms = pt_random(sx, sy);
	randoms.x = 2.0f * randoms.x - 1.0f;
	randoms.y = 2.0f * randoms.y - 1.0f;
	float scale = 0.9f * sqrtf(2.0f);
	float jx = ((float) x + 0.5f) + scale * pt_erfinv(randoms.x);
	float jy = ((float) y + 0.5f) + scale * pt_erfinv(randoms.y);
	pt_camera_ray(c, jx, jy, origin, dir);
	v3 origin_right, dir_right, origin_down, dir_down;
	pt_camera_ray(c, jx + 1.0f, jy, &origin_right, &dir_right);
	pt_camera_ray(c, jx, jy + 1.0f, &origin_down, &dir_down);
	*cone_width = fminf(pt_length(origin_right - *origin), pt_length(origin_down - *origin));
	*cone_spread = fminf(pt_length(dir_right - *dir), pt_length(dir_down - *dir));
}


// ---- mesh_quantization.glsl:4-11 ----
PT_FN v3 pt_dequantize_position(u32x2 q, const float* factor, const float* summand) {
	float px = (float) (q.x & 0x1fffffu);
	float py = (float) (((q.x >> 21) & 0x7ffu) | ((q.y & 0x3ffu) << 11));
	float pz = (float) ((q.y >> 10) & 0x1fffffu);
	return mk3(fmaf(px, factor[0], summand[0]), fmaf(py, factor[1], summand[1]), fmaf(pz, factor[2], summand[2]));
}

// ---- mesh_quantization.glsl:16-31 ----
PT_FN v3 pt_dequantize_normal(float ox, float oy) {
	const float factor = 2.0f * (65534.0f / 65535.0f);
	const float summand = -(32768.0f / 65535.0f) * factor;
	ox = fmaf(ox, factor, summand);
	oy = fmaf(oy, factor, summand);
	v3 n = mk3(ox, oy, (1.0f - fabsf(ox)) - fabsf(oy));
	float sign_x = (ox >= 0.0f) ? 1.0f : -1.0f, sign_y = (oy >= 0.0f) ? 1.0f : -1.0f;
	if (n.z < 0.0f) {
		float nx = (1.0f - fabsf(n.y)) * sign_x, ny = (1.0f - fabsf(n.x)) * sign_y;
		n.x = nx;
		n.y = ny;
	}
	return normalize(n);
}


// ---- texture(): shading_data.glsl:74-76 with the sampler of
// src/main.c:864-873 (linear min, mag and mip filters, REPEAT addressing, all
// mip levels, anisotropy 16). Blocks are decoded by hand with 8-bit integer
// palettes (once per scene, see below); filtering is trilinear with float
// weights: bilinear on the two mip levels next to the level of detail, which
// comes from the footprint of the path at the hit (pt_texture_lod_base()). ----

// BC1 block (two RGB565 endpoints + 16 2-bit codes) -> 8-bit RGB of one texel
PT_FN void pt_decode_bc1(uint32_t w0, uint32_t w1, uint32_t px, uint32_t py, uint32_t* r, uint32_t* g, uint32_t* b) {
	uint32_t c0 = w0 & 0xffffu, c1 = w0 >> 16;
	uint32_t code = (w1 >> (2 * (4 * py + px))) & 3u;
	uint32_t r0 = ((c0 >> 8) & 0xf8u) | (c0 >> 13), g0 = ((c0 >> 3) & 0xfcu) | ((c0 >> 9) & 3u), b0 = ((c0 << 3) & 0xf8u) | ((c0 >> 2) & 7u);
	uint32_t r1 = ((c1 >> 8) & 0xf8u) | (c1 >> 13), g1 = ((c1 >> 3) & 0xfcu) | ((c1 >> 9) & 3u), b1 = ((c1 << 3) & 0xf8u) | ((c1 >> 2) & 7u);
	bool four = c0 > c1;
	if (code == 0) { *r = r0; *g = g0; *b = b0; }
	else if (code == 1) { *r = r1; *g = g1; *b = b1; }
	else if (code == 2) {
		if (four) { *r = (2 * r0 + r1) / 3; *g = (2 * g0 + g1) / 3; *b = (2 * b0 + b1) / 3; }
		else { *r = (r0 + r1) / 2; *g = (g0 + g1) / 2; *b = (b0 + b1) / 2; }
	}
	else {
		if (four) { *r = (r0 + 2 * r1) / 3; *g = (g0 + 2 * g1) / 3; *b = (b0 + 2 * b1) / 3; }
		else { *r = *g = *b = 0; }
	}
}

// BC4 block (two 8-bit endpoints + 16 3-bit codes) -> 8-bit value of one texel
PT_FN uint32_t pt_decode_bc4(uint32_t w0, uint32_t w1, uint32_t px, uint32_t py) {
	uint32_t a0 = w0 & 0xffu, a1 = (w0 >> 8) & 0xffu;
	uint64_t bits = ((uint64_t) w1 << 16) | (w0 >> 16);
	uint32_t code = (uint32_t) (bits >> (3 * (4 * py + px))) & 7u;
	if (code == 0) return a0;
	if (code == 1) return a1;
	if (a0 > a1) return ((8 - code) * a0 + (code - 1) * a1) / 7;
	if (code == 6) return 0;
	if (code == 7) return 255;
	return ((6 - code) * a0 + (code - 1) * a1) / 5;
}

// Textures are decoded once, when a scene is created, into 8-bit texels of
// four bytes (R, G, B, 0) in tiles of 4x4 texels, one 64-byte line per tile as
// in the block-compressed original: the decode is exact integer arithmetic,
// so sampling the decoded texels gives the same values as decoding per
// sample, at a fifth of the instructions (shading is bound by them) and eight
// times the memory of BC1, which 288 GB of HBM can afford. The mip levels of
// a texture follow each other without gaps (pt_mip_texel_offset()).
PT_FN uint64_t pt_texel_index(uint32_t log_w, uint32_t x, uint32_t y) {
	return ((((uint64_t) (y >> 2) << (log_w - 2)) + (x >> 2)) << 4) + ((y & 3u) << 2) + (x & 3u);
}

// Texel (x, y) of one mip level, 2^log_w texels wide, of a BC1 (formats 131, 132) or BC5 (141) texture
PT_FN uint32_t pt_decode_texel(const uint8_t* mip, uint32_t log_w, uint32_t format, uint32_t x, uint32_t y) {
	uint64_t block = ((uint64_t) (y / 4) << (log_w - 2)) + x / 4;
	if (format == 141) {
		u32x4 w = *(const u32x4*) (mip + block * 16);
		uint32_t r = pt_decode_bc4(w.x, w.y, x & 3, y & 3);
		uint32_t g = pt_decode_bc4(w.z, w.w, x & 3, y & 3);
		return r | (g << 8);
	}
	u32x2 w = *(const u32x2*) (mip + block * 8);
	uint32_t r, g, b;
	pt_decode_bc1(w.x, w.y, x & 3, y & 3, &r, &g, &b);
	return r | (g << 8) | (b << 16);
}

// The sRGB decoding table (256 floats). A shaded vertex looks up twelve
// entries; the kernels keep a copy in LDS (pt_stage_srgb_table() at their
// start), which takes these gathers off the vector memory path that the
// texel and mesh fetches already keep busy.
#if defined(__HIPCC__)
__shared__ float pt_srgb_table_lds[256];
#endif
PT_FN float pt_srgb_to_linear(const PtSceneView& s, uint32_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
	return pt_srgb_table_lds[value];
#else
	return s.srgb_to_linear[value];
#endif
}

// The 8-bit texel (R, G << 8, B << 16) at (x, y) of one mip level, REPEAT addressing
PT_FN uint32_t pt_load_texel(const PtSceneView& s, u32x2 t, uint32_t level, int32_t x, int32_t y) {
	uint32_t log_w = (t.y & 0xffu) - level, log_h = ((t.y >> 8) & 0xffu) - level;
	uint32_t ux = (uint32_t) x & ((1u << log_w) - 1u), uy = (uint32_t) y & ((1u << log_h) - 1u);
	const uint32_t* texels = (const uint32_t*) (s.texture_blob + ((uint64_t) t.x << 4)) + pt_mip_texel_offset(t.y & 0xffu, (t.y >> 8) & 0xffu, level);
	return texels[pt_texel_index(log_w, ux, uy)];
}

// The value of a texel: sRGB decoding for VK_FORMAT_BC1_RGB_SRGB_BLOCK, else unorm
PT_FN v3 pt_texel_value(const PtSceneView& s, uint32_t format, uint32_t texel) {
	uint32_t r = texel & 0xffu, g = (texel >> 8) & 0xffu, b = (texel >> 16) & 0xffu;
	if (format == 132)
		return mk3(pt_srgb_to_linear(s, r), pt_srgb_to_linear(s, g), pt_srgb_to_linear(s, b));
	return mk3(pt_unorm8(r), pt_unorm8(g), pt_unorm8(b));
}

// The four texels around a sample position and its weights. Loading them
// is kept apart from using them: a shaded vertex needs up to 24 texels, and
// the loads only overlap if nothing that waits for one of them (the sRGB
// table lookups, the branch on the format) stands between them.
struct PtTexelQuad {
	uint32_t t00, t10, t01, t11;
	float wx, wy;
};

PT_FN PtTexelQuad pt_load_texel_quad(const PtSceneView& s, u32x2 t, uint32_t level, v2 uv) {
	float x = uv.x * (float) (1u << ((t.y & 0xffu) - level)) - 0.5f, y = uv.y * (float) (1u << (((t.y >> 8) & 0xffu) - level)) - 0.5f;
	float fx0 = floorf(x), fy0 = floorf(y);
	int32_t x0 = (int32_t) fx0, y0 = (int32_t) fy0;
	PtTexelQuad q;
	q.wx = x - fx0;
	q.wy = y - fy0;
	q.t00 = pt_load_texel(s, t, level, x0, y0);
	q.t10 = pt_load_texel(s, t, level, x0 + 1, y0);
	q.t01 = pt_load_texel(s, t, level, x0, y0 + 1);
	q.t11 = pt_load_texel(s, t, level, x0 + 1, y0 + 1);
	return q;
}

// Bilinear filtering of one mip level
PT_FN v3 pt_filter_texel_quad(const PtSceneView& s, uint32_t format, const PtTexelQuad& q) {
	v3 t00 = pt_texel_value(s, format, q.t00), t10 = pt_texel_value(s, format, q.t10);
	v3 t01 = pt_texel_value(s, format, q.t01), t11 = pt_texel_value(s, format, q.t11);
	v3 top = t00 * (1.0f - q.wx) + t10 * q.wx;
	v3 bottom = t01 * (1.0f - q.wx) + t11 * q.wx;
	return top * (1.0f - q.wy) + bottom * q.wy;
}

// The level of detail of a texture lookup, up to the size of the texture.
// The reference's texture() takes it from derivatives across a 2x2 pixel quad,
// which are undefined once neighbouring paths have hit different surfaces;
// here it follows from the footprint of the path itself (a ray cone): with
// uv_area_2 and world_area_2 the doubled areas of the triangle in texture
// coordinates and in the scene, a footprint of diameter `footprint` covers
// uv_area_2 / world_area_2 * footprint^2 of the unit square of a texture and
// 2^(log_w + log_h) times that in texels, so the level whose texels match it is
//   lod = 0.5 * (log2(uv_area_2 / world_area_2 * footprint^2) + log_w + log_h)
// The footprint is the one across the ray (the minor axis of its projection
// onto the surface), as an anisotropic filter takes it; like the sampler's
// maxAnisotropy of 16 (src/main.c:866-872) it grows only where the
// projection is more than 16 times as long as wide (cos = |dot(normal,
// direction)| < 1 / 16). Returns 0.5 * log2(...) without the size terms, or a
// large negative number (level 0) where the triangle has no texture area.
PT_FN float pt_texture_lod_base(float uv_area_2, float world_area_2, float footprint, float cos_out) {
	float stretched = footprint * fmaxf(1.0f, 0.0625f / cos_out);
	float covered = (uv_area_2 / world_area_2) * (stretched * stretched);
	if (!(covered > 1.0e-30f))
		return -1.0e9f;
	return 0.5f * (pt_log(fminf(covered, 1.0e30f)) * 1.44269504088896341f);
}

// Trilinear filtering: the two mip levels around lod_base + (log_w + log_h) / 2,
// clamped to the levels that the texture has. pt_begin_texture_lookup() issues
// the loads of both levels, pt_finish_texture_lookup() filters.
struct PtTextureLookup {
	uint32_t format;
	float weight; // of the coarser level
	PtTexelQuad fine, coarse;
};

PT_FN PtTextureLookup pt_begin_texture_lookup(const PtSceneView& s, uint32_t texture_index, v2 uv, float lod_base) {
	u32x2 t = s.texture_table[texture_index];
	uint32_t last_level = (t.y >> 24) - 1u;
	float lod = lod_base + 0.5f * (float) ((t.y & 0xffu) + ((t.y >> 8) & 0xffu));
	lod = fminf(fmaxf(lod, 0.0f), (float) last_level);
	float floor_lod = floorf(lod);
	uint32_t level = (uint32_t) floor_lod;
	PtTextureLookup lookup;
	lookup.format = (t.y >> 16) & 0xffu;
	lookup.weight = lod - floor_lod;
	lookup.fine = pt_load_texel_quad(s, t, level, uv);
	// Under magnification and at the last level the coarser level has no weight and is not read
	if (lookup.weight > 0.0f)
		lookup.coarse = pt_load_texel_quad(s, t, level + 1u, uv);
	else
		lookup.coarse = lookup.fine;
	return lookup;
}

PT_FN v3 pt_finish_texture_lookup(const PtSceneView& s, const PtTextureLookup& lookup) {
	v3 fine = pt_filter_texel_quad(s, lookup.format, lookup.fine);
	// At full weight the coarser level adds nothing (magnification and the last level end here)
	if (!(lookup.weight > 0.0f))
		return fine;
	v3 coarse = pt_filter_texel_quad(s, lookup.format, lookup.coarse);
	return fine * (1.0f - lookup.weight) + coarse * lookup.weight;
}

PT_FN v3 pt_sample_texture(const PtSceneView& s, uint32_t texture_index, v2 uv, float lod_base) {
	PtTextureLookup lookup = pt_begin_texture_lookup(s, texture_index, uv, lod_base);
	return pt_finish_texture_lookup(s, lookup);
}


// ---- Ray cones: the footprint of a path for pt_texture_lod_base() ----
// A ray carries the diameter of its footprint at its origin and the angle by
// which the footprint widens per unit of distance. Both travel with the ray
// as the upper halves of their floats (8 significant bits are plenty for a
// level of detail) in the one word of the ray record that is free.
PT_FN float pt_cone_quantize(float v) { return pt_as_float(pt_as_uint(v) & 0xffff0000u); }
PT_FN float pt_pack_cone(float width, float spread) { return pt_as_float((pt_as_uint(width) & 0xffff0000u) | (pt_as_uint(spread) >> 16)); }
PT_FN void pt_unpack_cone(float packed, float* width, float* spread) {
	*width = pt_as_float(pt_as_uint(packed) & 0xffff0000u);
	*spread = pt_as_float(pt_as_uint(packed) << 16);
}
// The widening that a bounce adds: half the angle of the cone whose solid
// angle is the reciprocal of the density with which the direction was
// sampled, 1 / sqrt(pi * density), at most 2; the sum is kept below 4
PT_FN float pt_cone_spread_after_bounce(float spread, float density) {
	return fminf(4.0f, spread + fminf(2.0f, 1.0f / sqrtf(PT_PI * density)));
}


#ifndef PT_TEXTURE_TAPS
// Texture lookups take one trilinear tap. A value above one builds the
// measurement variant with an anisotropic filter of up to that many taps
// (see pt_shading_data()), against which the one-tap rule is compared.
#define PT_TEXTURE_TAPS 1
#endif

// ---- shading_data.glsl:19-39 ----
struct PtShadingData {
	v3 pos, normal, out_dir;
	float lambert_out;
	v3 emission, diffuse_albedo, fresnel_0;
	float roughness;
	// Not in the reference: the diameter of the path's footprint at pos
	float footprint;
};

// ---- shading_data.glsl:51-103 ----
// The ray that led here started at cone_origin with a footprint of diameter
// cone_width that widens by cone_spread per unit of distance; that sets the
// level of detail of the three texture lookups.
PT_FN PtShadingData pt_shading_data(const PtSceneView& scene, const PtConstants& c, uint32_t triangle_index, float bary_u, float bary_v, bool front, v3 out_dir,
	v3 cone_origin, float cone_width, float cone_spread)
{
	PtShadingData s;
	float barys[3] = { (1.0f - bary_u) - bary_v, bary_u, bary_v };
	v3 poss[3];
	v2 tcs[3];
	s.pos = mk3(0.0f, 0.0f, 0.0f);
	v3 normal_geo = mk3(0.0f, 0.0f, 0.0f);
	v2 tc = mk2(0.0f, 0.0f);
	// One 64-byte line holds all of the triangle
	const u32x4* record = (const u32x4*) &scene.shade_tris[triangle_index];
	u32x4 r0 = record[0], r1 = record[1], r2 = record[2], r3 = record[3];
	u32x2 vertex_positions[3] = { { r0.x, r0.y }, { r0.z, r0.w }, { r1.x, r1.y } };
	u32x2 vertex_normals_uvs[3] = { { r1.z, r1.w }, { r2.x, r2.y }, { r2.z, r2.w } };
	for (int i = 0; i != 3; ++i) {
		u32x2 qp = vertex_positions[i];
		u32x2 qn = vertex_normals_uvs[i];
		float n0 = pt_unorm16(qn.x & 0xffffu), n1 = pt_unorm16(qn.x >> 16);
		float t0 = pt_unorm16(qn.y & 0xffffu), t1 = pt_unorm16(qn.y >> 16);
		poss[i] = pt_dequantize_position(qp, c.dequantization_factor, c.dequantization_summand);
		s.pos = s.pos + poss[i] * barys[i];
		normal_geo = normal_geo + pt_dequantize_normal(n0, n1) * barys[i];
		tcs[i] = mk2(t0 * 8.0f + 0.0f, t1 * -8.0f + 1.0f);
		tc.x += barys[i] * tcs[i].x;
		tc.y += barys[i] * tcs[i].y;
	}
	normal_geo = normalize(normal_geo);
	uint32_t material = r3.x;
	// mat2(e1, e2) holds e1 and e2 as columns: [0][0] = e1.x, [1][0] = e2.x
	v2 e1 = mk2(tcs[1].x - tcs[0].x, tcs[1].y - tcs[0].y), e2 = mk2(tcs[2].x - tcs[0].x, tcs[2].y - tcs[0].y);
	v3 face = cross(poss[1] - poss[0], poss[2] - poss[0]);
	s.footprint = fmaf(cone_spread, pt_length(s.pos - cone_origin), cone_width);
	float lod_base = pt_texture_lod_base(fabsf(e1.x * e2.y - e2.x * e1.y), pt_length(face), s.footprint, fabsf(dot(normal_geo, out_dir)));
#if PT_TEXTURE_TAPS > 1
	// Measurement variant (never the product's default): an anisotropic filter as
	// a sampler with maxAnisotropy = PT_TEXTURE_TAPS takes it (src/main.c:864-873),
	// N = min(ceil(1 / cos), PT_TEXTURE_TAPS) trilinear taps spread evenly along the
	// major axis of the footprint's ellipse on the surface, the projection of the
	// ray onto the triangle's plane, each at the level of detail of the minor axis
	v3 base_color = mk3(0.0f, 0.0f, 0.0f), specular = mk3(0.0f, 0.0f, 0.0f), normal_tex = mk3(0.0f, 0.0f, 0.0f);
	{
		float cos_out = fmaxf(fabsf(dot(normal_geo, out_dir)), 1.0e-4f);
		float stretch = fminf(1.0f / cos_out, (float) PT_TEXTURE_TAPS);
		int taps = (int) ceilf(stretch);
		v3 along = out_dir - normal_geo * dot(normal_geo, out_dir);
		float along_length = pt_length(along);
		v3 major = (along_length > 1.0e-6f) ? along * (1.0f / along_length) : mk3(0.0f, 0.0f, 0.0f);
		// A step along the surface in texture coordinates: major = a * edge_1 + b * edge_2 (least squares)
		v3 edge_1 = poss[1] - poss[0], edge_2 = poss[2] - poss[0];
		float g11 = dot(edge_1, edge_1), g12 = dot(edge_1, edge_2), g22 = dot(edge_2, edge_2);
		float determinant = g11 * g22 - g12 * g12;
		float r1 = dot(major, edge_1), r2 = dot(major, edge_2);
		float a = 0.0f, b = 0.0f;
		if (determinant > 0.0f) {
			a = (r1 * g22 - r2 * g12) / determinant;
			b = (r2 * g11 - r1 * g12) / determinant;
		}
		v2 uv_step = mk2(a * e1.x + b * e2.x, a * e1.y + b * e2.y);
		float half_length = 0.5f * s.footprint * stretch;
		for (int tap = 0; tap != taps; ++tap) {
			float offset = half_length * ((2.0f * ((float) tap + 0.5f)) / (float) taps - 1.0f);
			v2 tap_tc = mk2(tc.x + offset * uv_step.x, tc.y + offset * uv_step.y);
			base_color = base_color + pt_sample_texture(scene, 3 * material + 0, tap_tc, lod_base);
			specular = specular + pt_sample_texture(scene, 3 * material + 1, tap_tc, lod_base);
			normal_tex = normal_tex + pt_sample_texture(scene, 3 * material + 2, tap_tc, lod_base);
		}
		float weight = 1.0f / (float) taps;
		base_color = base_color * weight;
		specular = specular * weight;
		normal_tex = normal_tex * weight;
	}
#else
	// All texels of the three lookups are requested before the first is used
	PtTextureLookup base_color_lookup = pt_begin_texture_lookup(scene, 3 * material + 0, tc, lod_base);
	PtTextureLookup specular_lookup = pt_begin_texture_lookup(scene, 3 * material + 1, tc, lod_base);
	PtTextureLookup normal_lookup = pt_begin_texture_lookup(scene, 3 * material + 2, tc, lod_base);
	v3 base_color = pt_finish_texture_lookup(scene, base_color_lookup);
	v3 specular = pt_finish_texture_lookup(scene, specular_lookup);
	v3 normal_tex = pt_finish_texture_lookup(scene, normal_lookup);
#endif
	v3 normal_local;
	normal_local.x = normal_tex.x * 2.0f - 1.0f;
	normal_local.y = normal_tex.y * 2.0f - 1.0f;
	normal_local.z = sqrtf(fmaxf(0.0f, (1.0f - normal_local.x * normal_local.x) - normal_local.y * normal_local.y));
	v3 pre_tangent_0 = cross(normal_geo, poss[1] - poss[0]);
	v3 pre_tangent_1 = cross(normal_geo, poss[0] - poss[2]);
	v3 tangent_0 = pre_tangent_1 * e1.x + pre_tangent_0 * e2.x;
	v3 tangent_1 = pre_tangent_1 * e1.y + pre_tangent_0 * e2.y;
	float mean_length = sqrtf(0.5f * (dot(tangent_0, tangent_0) + dot(tangent_1, tangent_1)));
	normal_local.z *= fmaxf(1.0e-8f, mean_length);
	s.normal = normalize((tangent_0 * normal_local.x + tangent_1 * normal_local.y) + normal_geo * normal_local.z);
	s.normal = front ? s.normal : -s.normal;
	s.out_dir = out_dir;
	float normal_offset = fmaxf(0.0f, 1.0e-3f - dot(s.normal, s.out_dir));
	s.normal = normalize(mk3(fmaf(normal_offset, s.out_dir.x, s.normal.x), fmaf(normal_offset, s.out_dir.y, s.normal.y), fmaf(normal_offset, s.out_dir.z, s.normal.z)));
	s.lambert_out = dot(s.normal, s.out_dir);
	float metalicity = specular.z;
	s.diffuse_albedo = base_color - base_color * metalicity;
	s.fresnel_0 = mk3(0.02f, 0.02f, 0.02f) * (1.0f - metalicity) + base_color * metalicity;
	s.roughness = fmaxf(0.006f, specular.y * specular.y);
	s.emission = (material == scene.emission_material_index)
		? mk3(c.emission_material_radiance[0], c.emission_material_radiance[1], c.emission_material_radiance[2])
		: mk3(0.0f, 0.0f, 0.0f);
	return s;
}


// ---- brdfs.glsl:7-12, per channel ----
PT_FN float pt_fresnel_schlick(float f_0, float f_90, float lambert) {
	float flip_1 = 1.0f - lambert;
	float flip_2 = flip_1 * flip_1;
	float flip_5 = flip_2 * flip_1 * flip_2;
	return flip_5 * (f_90 - f_0) + f_0;
}

// ---- brdfs.glsl:19-46 ----
PT_FN v3 pt_frostbite_brdf(const PtShadingData& s, v3 in_dir) {
	float lambert_in = dot(s.normal, in_dir);
	if (fminf(lambert_in, s.lambert_out) < 0.0f)
		return mk3(0.0f, 0.0f, 0.0f);
	v3 half_dir = normalize(in_dir + s.out_dir);
	float half_dot_out = dot(half_dir, s.out_dir);
	float f_90 = (half_dot_out * half_dot_out) * (2.0f * s.roughness) + 0.5f;
	float fresnel_diffuse = pt_fresnel_schlick(1.0f, f_90, s.lambert_out) * pt_fresnel_schlick(1.0f, f_90, lambert_in);
	v3 brdf = s.diffuse_albedo * fresnel_diffuse;
	float half_dot_normal = dot(half_dir, s.normal);
	float roughness_2 = s.roughness * s.roughness;
	float ggx = (roughness_2 * half_dot_normal - half_dot_normal) * half_dot_normal + 1.0f;
	ggx = roughness_2 / (ggx * ggx);
	float masking = lambert_in * sqrtf((s.lambert_out - roughness_2 * s.lambert_out) * s.lambert_out + roughness_2);
	float shadowing = s.lambert_out * sqrtf((lambert_in - roughness_2 * lambert_in) * lambert_in + roughness_2);
	float smith = 0.5f / (masking + shadowing);
	float l = fmaxf(0.0f, half_dot_out);
	v3 fresnel = mk3(pt_fresnel_schlick(s.fresnel_0.x, 1.0f, l), pt_fresnel_schlick(s.fresnel_0.y, 1.0f, l), pt_fresnel_schlick(s.fresnel_0.z, 1.0f, l));
	brdf = brdf + fresnel * (ggx * smith);
	return brdf * (1.0f / PT_PI);
}

// ---- brdfs.glsl:58-75 with roughness equal on both axes ----
PT_FN v3 pt_sample_ggx_vndf(v3 out_dir, float roughness, v2 randoms) {
	v3 out_dir_std = normalize(mk3(out_dir.x * roughness, out_dir.y * roughness, out_dir.z));
	float azimuth = (2.0f * PT_PI) * randoms.x - PT_PI;
	float z = 1.0f - randoms.y * (1.0f + out_dir_std.z);
	float sine = sqrtf(fmaxf(0.0f, 1.0f - z * z));
	float sa, ca;
	pt_sincos(azimuth, &sa, &ca);
	v3 half_dir_std = mk3(sine * ca, sine * sa, z) + out_dir_std;
	return normalize(mk3(half_dir_std.x * roughness, half_dir_std.y * roughness, half_dir_std.z));
}

// ---- brdfs.glsl:88-99 ----
PT_FN float pt_ggx_vndf_density(float lambert_out, float half_dot_normal, float half_dot_out, float roughness) {
	if (half_dot_normal < 0.0f)
		return 0.0f;
	float roughness_2 = roughness * roughness;
	float flip_roughness_2 = 1.0f - roughness * roughness;
	float length_M_inv_out_2 = roughness_2 + flip_roughness_2 * lambert_out * lambert_out;
	float D_vis_std = fmaxf(0.0f, half_dot_out) * (2.0f / PT_PI) / (lambert_out + sqrtf(length_M_inv_out_2));
	float length_M_half_2 = 1.0f - flip_roughness_2 * half_dot_normal * half_dot_normal;
	return D_vis_std * roughness_2 / (length_M_half_2 * length_M_half_2);
}

// ---- brdfs.glsl:104-107: -reflect(o, h) with reflect(I, N) = I - 2 dot(N, I) N ----
PT_FN v3 pt_sample_ggx_in_dir(v3 out_dir, float roughness, v2 randoms) {
	v3 half_dir = pt_sample_ggx_vndf(out_dir, roughness, randoms);
	return -(out_dir - half_dir * (2.0f * dot(half_dir, out_dir)));
}

// ---- brdfs.glsl:112-121 ----
PT_FN float pt_ggx_in_dir_density(float lambert_out, v3 out_dir, v3 in_dir, v3 normal, float roughness) {
	v3 half_dir = normalize(in_dir + out_dir);
	float half_dot_out = dot(half_dir, out_dir);
	float half_dot_normal = dot(half_dir, normal);
	float density = pt_ggx_vndf_density(lambert_out, half_dot_normal, half_dot_out, roughness);
	return density / (4.0f * half_dot_out);
}

// ---- brdfs.glsl:139-146 ----
PT_FN v3 pt_sample_hemisphere_psa(v2 randoms) {
	float azimuth = (2.0f * PT_PI) * randoms.x - PT_PI;
	float radius = sqrtf(randoms.y);
	float z = sqrtf(1.0f - radius * radius);
	float sa, ca;
	pt_sincos(azimuth, &sa, &ca);
	return mk3(radius * ca, radius * sa, z);
}

// ---- brdfs.glsl:151-153 ----
PT_FN float pt_hemisphere_psa_density(float z) { return (1.0f / PT_PI) * fmaxf(0.0f, z); }

// ---- brdfs.glsl:158-164 ----
PT_FN float pt_diffuse_sampling_probability(const PtShadingData& s) {
	return fminf(0.5f, dot(s.diffuse_albedo, mk3(0.2126f, 0.7152f, 0.0722f)));
}

// ---- brdfs.glsl:174-199 ----
PT_FN v3 pt_sample_frostbite_brdf(const PtShadingData& s, v2 randoms) {
	PtFrame frame = pt_shading_space(s.normal);
	float diffuse_prob = pt_diffuse_sampling_probability(s);
	if (randoms.x < diffuse_prob) {
		randoms.x /= diffuse_prob;
		return pt_to_world(frame, pt_sample_hemisphere_psa(randoms));
	}
	randoms.x = (randoms.x - diffuse_prob) / (1.0f - diffuse_prob);
	v3 local_out_dir = pt_to_local(frame, s.out_dir);
	return pt_to_world(frame, pt_sample_ggx_in_dir(local_out_dir, s.roughness, randoms));
}

// ---- brdfs.glsl:203-208; mix(a, b, t) = a * (1 - t) + b * t ----
PT_FN float pt_frostbite_brdf_density(const PtShadingData& s, v3 sampled_dir) {
	float diffuse_prob = pt_diffuse_sampling_probability(s);
	float specular_density = pt_ggx_in_dir_density(s.lambert_out, s.out_dir, sampled_dir, s.normal, s.roughness);
	float diffuse_density = pt_hemisphere_psa_density(dot(s.normal, sampled_dir));
	return specular_density * (1.0f - diffuse_prob) + diffuse_density * diffuse_prob;
}


// The lights as the loops over all lights read them. Every thread reads the
// same light in the same iteration, so on the GPU the read goes through the
// constant address space: a scalar load into registers that the whole wave
// shares, served by the scalar cache, with no vector memory instruction and no
// LDS access per light. (The lights do not change while a kernel runs.)
// The loops over all lights are chains of dependent operations per light
// (two divisions and a square root in the first); unrolled, neighbouring
// lights overlap
#ifndef PT_LIGHT_UNROLL
#define PT_LIGHT_UNROLL 1
#endif
#define PT_STRINGIFY_2(x) #x
#define PT_STRINGIFY(x) PT_STRINGIFY_2(x)
#define PT_LIGHT_LOOP_UNROLL _Pragma(PT_STRINGIFY(unroll PT_LIGHT_UNROLL))
PT_FN f32x4 pt_light(const PtSceneView& s, uint32_t index) {
#if defined(__HIP_DEVICE_COMPILE__)
	typedef float pt_light_words __attribute__((ext_vector_type(4)));
	typedef __attribute__((address_space(4))) const pt_light_words pt_constant_light;
	pt_light_words words = ((pt_constant_light*) s.lights)[index];
	return mk4(words.x, words.y, words.z, words.w);
#else
	return s.lights[index];
#endif
}

// sqrtf(x) for an x that is zero or not below 2^-96, bit for bit: the
// hardware's square root (one unit in the last place) corrected by the signs
// of two exact residuals, which is what the compiler emits for sqrtf() minus
// the rescaling of smaller arguments (nine more instructions around it). For
// x = 0 the root is 0, the lower candidate is a NaN and the upper one leaves a
// residual of zero, so 0 comes out as it is.
PT_FN float pt_sqrt_at_least_2_pow_minus_96(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
	float root = __builtin_amdgcn_sqrtf(x);
	float below = pt_as_float(pt_as_uint(root) - 1u), above = pt_as_float(pt_as_uint(root) + 1u);
	float residual_below = fmaf(-below, root, x), residual_above = fmaf(-above, root, x);
	root = (residual_below <= 0.0f) ? below : root;
	return (residual_above > 0.0f) ? above : root;
#else
	return sqrtf(x);
#endif
}

// ---- pathtrace.frag.glsl:69-83 ----
PT_FN float pt_spherical_light_importance(f32x4 light, v3 shading_pos, v3 normal) {
	v3 center_dir = mk3(light.x, light.y, light.z) - shading_pos;
	if (dot(normal, center_dir) < -light.w)
		return 0.0f;
	float center_dist_2 = dot(center_dir, center_dir);
	float sin_2 = light.w * light.w / center_dist_2;
	// 1 - sin_2 is a multiple of 2^-24 when sin_2 is 1/2 or more and above 1/2
	// otherwise, so the argument of the root is zero or at least 2^-24
	return sin_2 / (1.0f + pt_sqrt_at_least_2_pow_minus_96(fmaxf(0.0f, 1.0f - sin_2)));
}

// The same with the hardware's reciprocal and square root, each good to one
// unit in the last place, instead of two correctly rounded divisions and a
// correctly rounded root: a third of the instructions. The value differs from
// pt_spherical_light_importance() in the last bits, so an image differs from
// the exact one like two images with different seeds do, only far less.
PT_FN float pt_spherical_light_importance_fast(f32x4 light, v3 shading_pos, v3 normal) {
	v3 center_dir = mk3(light.x, light.y, light.z) - shading_pos;
	if (dot(normal, center_dir) < -light.w)
		return 0.0f;
	float center_dist_2 = dot(center_dir, center_dir);
#if defined(__HIP_DEVICE_COMPILE__)
	float sin_2 = (light.w * light.w) * __builtin_amdgcn_rcpf(center_dist_2);
	return sin_2 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - sin_2)));
#else
	float sin_2 = light.w * light.w / center_dist_2;
	return sin_2 / (1.0f + sqrtf(fmaxf(0.0f, 1.0f - sin_2)));
#endif
}

// ---- pathtrace.frag.glsl:95-106 ----
PT_FN v3 pt_sample_spherical_light(f32x4 light, float importance, v3 shading_pos, v2 randoms) {
	float azimuth = (2.0f * PT_PI) * randoms.x - PT_PI;
	float z = 1.0f - importance * randoms.y;
	float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
	float sa, ca;
	pt_sincos(azimuth, &sa, &ca);
	PtFrame frame = pt_shading_space(normalize(mk3(light.x, light.y, light.z) - shading_pos));
	return pt_to_world(frame, mk3(r * ca, r * sa, z));
}

// How far from a shading point the sphere of a light begins, roughly (zero
// inside): shadow rays carry it so that the traversal can tell a hit in front
// of the light they aim at from one behind it. It steers work, not results, so
// the hardware's square root will do.
PT_FN float pt_light_near_distance(f32x4 light, v3 shading_pos) {
	v3 center_dir = mk3(light.x, light.y, light.z) - shading_pos;
#if defined(__HIP_DEVICE_COMPILE__)
	return fmaxf(0.0f, __builtin_amdgcn_sqrtf(dot(center_dir, center_dir)) - light.w);
#else
	return fmaxf(0.0f, sqrtf(dot(center_dir, center_dir)) - light.w);
#endif
}

// Where pt_sample_lights() keeps the importances of its first loop for its
// second one: entry i of a thread is at values[i * stride]. The reference
// evaluates get_spherical_light_importance() anew in both loops; the values
// are the same, so the result is too. Lights beyond capacity are recomputed.
struct PtLightCache {
	float* values;
	uint32_t stride, capacity;
};

// ---- pathtrace.frag.glsl:120-143 ----
PT_FN v3 pt_sample_lights(const PtSceneView& scene, const PtLightCache& cache, float* out_total_importance, v3 shading_pos, v3 normal, v2 randoms, bool fast = false, float* out_near_distance = NULL) {
	float total = 0.0f;
	PT_LIGHT_LOOP_UNROLL
	for (uint32_t i = 0; i != scene.light_count; ++i) {
		float importance = fast ? pt_spherical_light_importance_fast(pt_light(scene, i), shading_pos, normal) : pt_spherical_light_importance(pt_light(scene, i), shading_pos, normal);
		if (i < cache.capacity)
			cache.values[i * cache.stride] = importance;
		total += importance;
	}
	*out_total_importance = total;
	float target_importance = randoms.x * total;
	float prefix_importance = 0.0f;
	for (uint32_t i = 0; i != scene.light_count; ++i) {
		f32x4 light = pt_light(scene, i);
		float importance = (i < cache.capacity) ? cache.values[i * cache.stride]
			: (fast ? pt_spherical_light_importance_fast(light, shading_pos, normal) : pt_spherical_light_importance(light, shading_pos, normal));
		prefix_importance += importance;
		if (prefix_importance > target_importance) {
			randoms.x = ((target_importance + importance) - prefix_importance) / importance;
			if (out_near_distance)
				*out_near_distance = pt_light_near_distance(light, shading_pos);
			return pt_sample_spherical_light(light, importance, shading_pos, randoms);
		}
	}
	return mk3(0.0f, 0.0f, 0.0f);
}

// ---- pathtrace.frag.glsl:150-175 for two directions at once: the first is
// a direction towards a light (is_light_dir true), the second is not. One
// pass over the lights serves both. ----
PT_FN void pt_lights_density_2(const PtSceneView& scene, float total_importance, v3 shading_pos, v3 light_dir, v3 other_dir, float* out_light_dir_density, float* out_other_dir_density) {
	*out_light_dir_density = *out_other_dir_density = 0.0f;
	if (total_importance <= 0.0f)
		return;
	float count_0 = 0.0f, count_1 = 0.0f;
	PT_LIGHT_LOOP_UNROLL
	for (uint32_t i = 0; i != scene.light_count; ++i) {
		f32x4 light = pt_light(scene, i);
		v3 center_dir = mk3(light.x, light.y, light.z) - shading_pos;
		float center_dist_2 = dot(center_dir, center_dir);
		float radius_2 = light.w * light.w;
		float in_sphere = center_dist_2 - radius_2;
		float center_dot_dir_0 = dot(center_dir, light_dir), center_dot_dir_1 = dot(center_dir, other_dir);
		float discriminant_0 = center_dot_dir_0 * center_dot_dir_0 - in_sphere, discriminant_1 = center_dot_dir_1 * center_dot_dir_1 - in_sphere;
		count_0 += (discriminant_0 >= 0.0f && in_sphere >= 0.0f && center_dot_dir_0 >= 0.0f) ? 1.0f : 0.0f;
		count_1 += (discriminant_1 >= 0.0f && in_sphere >= 0.0f && center_dot_dir_1 >= 0.0f) ? 1.0f : 0.0f;
	}
	count_0 = fmaxf(1.0f, count_0);
	*out_light_dir_density = count_0 / (2.0f * PT_PI * total_importance);
	*out_other_dir_density = count_1 / (2.0f * PT_PI * total_importance);
}


// ---- Light tree (structure and estimator: pt_light_tree.h) ----

// Weight of a node for the descent from a shading point, as a fraction: its
// power over the squared distance to the centre of its box (at least the
// squared half diagonal). The power is zero for an empty node and for one whose
// box lies below the horizon of the shading point.
struct PtLightNodeWeight {
	float power, distance_2;
};

PT_FN PtLightNodeWeight pt_light_node_weight(const f32x4* nodes, uint32_t node, v3 shading_pos, v3 normal) {
	PtLightNodeWeight weight;
	weight.power = 0.0f;
	weight.distance_2 = 1.0f;
	f32x4 center_power = nodes[2 * node], half_radius = nodes[2 * node + 1];
	if (!(center_power.w > 0.0f))
		return weight;
	v3 center_dir = mk3(center_power.x, center_power.y, center_power.z) - shading_pos;
	// The largest dot(normal, x - shading_pos) over the points x of the box
	float reach = (fabsf(normal.x) * half_radius.x + fabsf(normal.y) * half_radius.y) + fabsf(normal.z) * half_radius.z;
	if (dot(normal, center_dir) + reach < 0.0f)
		return weight;
	weight.power = center_power.w;
	weight.distance_2 = fmaxf(dot(center_dir, center_dir), half_radius.w);
	return weight;
}

// The weights of the two children of a node over a common denominator:
// power_0 / distance_0 : power_1 / distance_1 = power_0 * distance_1 :
// power_1 * distance_0. The descent and the probability of a leaf want their
// proportion only, and a product costs a twelfth of a correctly rounded
// quotient.
PT_FN void pt_light_child_weights(const f32x4* nodes, uint32_t left_child, v3 shading_pos, v3 normal, float* out_weight_0, float* out_weight_1) {
	PtLightNodeWeight w0 = pt_light_node_weight(nodes, left_child, shading_pos, normal);
	PtLightNodeWeight w1 = pt_light_node_weight(nodes, left_child + 1u, shading_pos, normal);
	*out_weight_0 = w0.power * w1.distance_2;
	*out_weight_1 = w1.power * w0.distance_2;
}

// The leaf that a descent of pt_sample_lights_tree() arrived at (0: none) and
// the probability of arriving there. The density of the sampled direction
// needs that probability again (pt_light_tree_probability() recomputes it by
// the same arithmetic, level by level): kept, it saves that loop for the one
// light that nearly every vertex meets.
struct PtLightTreeSample {
	uint32_t leaf;
	float probability;
};

// Chooses a light by one descent and samples its cone like
// pt_sample_spherical_light(). out_density receives the chosen light's share
// of the density of the returned direction (selection probability times cone
// density); it is zero and the direction is the zero vector if no light was
// sampled.
PT_FN v3 pt_sample_lights_tree(const PtSceneView& scene, float* out_density, PtLightTreeSample* out_sample, v3 shading_pos, v3 normal, v2 randoms, float* out_near_distance = NULL) {
	*out_density = 0.0f;
	out_sample->leaf = 0;
	out_sample->probability = 0.0f;
	uint32_t node = 1;
	float probability = 1.0f;
	while (node < scene.light_leaf_count) {
		float weight_0, weight_1;
		pt_light_child_weights(scene.light_nodes, 2 * node, shading_pos, normal, &weight_0, &weight_1);
		float total = weight_0 + weight_1;
		if (!(total > 0.0f))
			return mk3(0.0f, 0.0f, 0.0f);
		float target = randoms.x * total;
		bool left = target < weight_0 || !(weight_1 > 0.0f);
		float weight = left ? weight_0 : weight_1;
		// The random number is reused as in pathtrace.frag.glsl:137
		randoms.x = fminf(1.0f, (left ? target : (target - weight_0)) / weight);
		probability *= weight / total;
		node = 2 * node + (left ? 0u : 1u);
	}
	out_sample->leaf = node;
	out_sample->probability = probability;
	f32x4 light = scene.tree_lights[node - scene.light_leaf_count];
	// No sample of a light from inside its sphere: pt_ray_meets_light() counts a
	// light only for rays that start outside of it, and the formulas of
	// pathtrace.frag.glsl:69-106 suppose a point outside (inside, z_range exceeds
	// one, z goes below -1 and the sampled vector is no direction any more, which
	// the BRDF answers with negative values). Emission seen from inside is found
	// by the BRDF sample, with a light density of zero.
	v3 center_dir = mk3(light.x, light.y, light.z) - shading_pos;
	if (!(dot(center_dir, center_dir) - light.w * light.w >= 0.0f))
		return mk3(0.0f, 0.0f, 0.0f);
	float importance = pt_spherical_light_importance(light, shading_pos, normal);
	if (!(importance > 0.0f))
		return mk3(0.0f, 0.0f, 0.0f);
	*out_density = probability / ((2.0f * PT_PI) * importance);
	if (out_near_distance)
		*out_near_distance = pt_light_near_distance(light, shading_pos);
	return pt_sample_spherical_light(light, importance, shading_pos, randoms);
}

// The probability with which the descent of pt_sample_lights_tree() arrives
// at a given leaf node, by the same arithmetic
PT_FN float pt_light_tree_probability(const PtSceneView& scene, uint32_t leaf_node, v3 shading_pos, v3 normal) {
	float probability = 1.0f;
	for (uint32_t shift = 31u - (uint32_t) __builtin_clz(leaf_node); shift-- != 0;) {
		uint32_t node = leaf_node >> shift;
		float weight_0, weight_1;
		pt_light_child_weights(scene.light_nodes, node & ~1u, shading_pos, normal, &weight_0, &weight_1);
		float weight = (node & 1u) ? weight_1 : weight_0;
		if (!(weight > 0.0f))
			return 0.0f;
		probability *= weight / (weight_0 + weight_1);
	}
	return probability;
}

// A ray prepared for slab tests against boxes given by centre and half
// extents: with inv = 1 / direction (components too small to invert replaced
// by +-1e-30), the ray is inside the slab of an axis for parameters
// centre * inv - origin * inv -+ half * |inv|
struct PtSlabRay {
	v3 inv, origin_inv, abs_inv;
};

PT_FN PtSlabRay pt_slab_ray(v3 origin, v3 dir) {
	float x = (fabsf(dir.x) > 1.0e-30f) ? dir.x : copysignf(1.0e-30f, dir.x);
	float y = (fabsf(dir.y) > 1.0e-30f) ? dir.y : copysignf(1.0e-30f, dir.y);
	float z = (fabsf(dir.z) > 1.0e-30f) ? dir.z : copysignf(1.0e-30f, dir.z);
	PtSlabRay ray;
	ray.inv = mk3(1.0f / x, 1.0f / y, 1.0f / z);
	ray.origin_inv = mk3(origin.x * ray.inv.x, origin.y * ray.inv.y, origin.z * ray.inv.z);
	ray.abs_inv = mk3(fabsf(ray.inv.x), fabsf(ray.inv.y), fabsf(ray.inv.z));
	return ray;
}

// Whether the ray meets the box at a parameter >= 0
PT_FN bool pt_ray_meets_box(const PtSlabRay& ray, v3 center, v3 half) {
	float cx = fmaf(center.x, ray.inv.x, -ray.origin_inv.x), hx = half.x * ray.abs_inv.x;
	float cy = fmaf(center.y, ray.inv.y, -ray.origin_inv.y), hy = half.y * ray.abs_inv.y;
	float cz = fmaf(center.z, ray.inv.z, -ray.origin_inv.z), hz = half.z * ray.abs_inv.z;
	float t_near = fmaxf(fmaxf(cx - hx, cy - hy), fmaxf(cz - hz, 0.0f));
	float t_far = fminf(fminf(cx + hx, cy + hy), cz + hz);
	return t_near <= t_far;
}

// Leaves whose boxes the rays of pt_lights_density_tree_2() meet and whose
// lights are still to be tested, in the order of discovery. They are worked
// off after the walk (or when four have gathered), so that the threads of a
// wavefront do the sphere tests and the 14-level probability loops at the same
// time instead of one after the other in the middle of their walks.
struct PtLightLeaves {
	uint32_t node_0, node_1, node_2, node_3, count;
};

// Whether the ray from shading_pos along dir meets the light and starts
// outside of it: the condition of pathtrace.frag.glsl:160-170, but with the
// distance between the light's centre and the line taken from a cross product.
// The reference's discriminant subtracts two squared distances and is off by
// about a thousandth of the distance to the light, which no box or grid over
// the lights could follow; this form is off by a few millionths of the
// coordinates involved and holds for directions of any length.
PT_FN bool pt_ray_meets_light(f32x4 light, v3 shading_pos, v3 dir, float dir_length_2) {
	v3 center_dir = mk3(light.x, light.y, light.z) - shading_pos;
	float in_sphere = dot(center_dir, center_dir) - light.w * light.w;
	float center_dot_dir = dot(center_dir, dir);
	v3 lever = cross(center_dir, dir);
	return dot(lever, lever) <= (light.w * light.w) * dir_length_2 && dir_length_2 > 0.0f && in_sphere >= 0.0f && center_dot_dir >= 0.0f;
}

PT_FN bool pt_ray_meets_light(f32x4 light, v3 shading_pos, v3 dir) { return pt_ray_meets_light(light, shading_pos, dir, dot(dir, dir)); }

// Adds the share of a leaf's light to the densities of the rays that meet the
// light
PT_FN void pt_light_leaf_density(const PtSceneView& scene, const PtLightTreeSample& sample, uint32_t node, v3 shading_pos, v3 normal, v3 dir_0, v3 dir_1, float& density_0, float& density_1) {
	f32x4 light = scene.tree_lights[node - scene.light_leaf_count];
	bool hit_0 = pt_ray_meets_light(light, shading_pos, dir_0), hit_1 = pt_ray_meets_light(light, shading_pos, dir_1);
	if (hit_0 || hit_1) {
		float importance = pt_spherical_light_importance(light, shading_pos, normal);
		if (importance > 0.0f) {
			float selection = (node == sample.leaf) ? sample.probability : pt_light_tree_probability(scene, node, shading_pos, normal);
			float density = selection / ((2.0f * PT_PI) * importance);
			density_0 += hit_0 ? density : 0.0f;
			density_1 += hit_1 ? density : 0.0f;
		}
	}
}

PT_FN void pt_light_leaves_flush(const PtSceneView& scene, const PtLightTreeSample& sample, PtLightLeaves& leaves, v3 shading_pos, v3 normal, v3 dir_0, v3 dir_1, float& density_0, float& density_1) {
	if (leaves.count > 0) pt_light_leaf_density(scene, sample, leaves.node_0, shading_pos, normal, dir_0, dir_1, density_0, density_1);
	if (leaves.count > 1) pt_light_leaf_density(scene, sample, leaves.node_1, shading_pos, normal, dir_0, dir_1, density_0, density_1);
	if (leaves.count > 2) pt_light_leaf_density(scene, sample, leaves.node_2, shading_pos, normal, dir_0, dir_1, density_0, density_1);
	if (leaves.count > 3) pt_light_leaf_density(scene, sample, leaves.node_3, shading_pos, normal, dir_0, dir_1, density_0, density_1);
	leaves.count = 0;
}

// Whether one of the two rays meets the box of a node that is not empty (a
// slab test; the box is widened by a ten-thousandth and by slack, a millionth
// of the coordinates involved, so that rounding skips no light that the test
// above would accept)
PT_FN bool pt_light_tree_visit(const PtSceneView& scene, uint32_t node, const PtSlabRay& ray_0, const PtSlabRay& ray_1, float slack) {
	f32x4 center_power = scene.light_nodes[2 * node], half_radius = scene.light_nodes[2 * node + 1];
	v3 center = mk3(center_power.x, center_power.y, center_power.z);
	v3 half = mk3(fmaf(half_radius.x, 1.0001f, slack), fmaf(half_radius.y, 1.0001f, slack), fmaf(half_radius.z, 1.0001f, slack));
	return center_power.w > 0.0f && (pt_ray_meets_box(ray_0, center, half) || pt_ray_meets_box(ray_1, center, half));
}

// Up to four leaves in ascending order (unused slots hold 0xffffffff); a fifth
// sets overflow
struct PtLeafSet {
	uint32_t leaf_0, leaf_1, leaf_2, leaf_3, count;
	bool overflow;
};

PT_FN void pt_leaf_set_insert(PtLeafSet& set, uint32_t leaf) {
	if (leaf == set.leaf_0 || leaf == set.leaf_1 || leaf == set.leaf_2 || leaf == set.leaf_3)
		return;
	if (set.count == 4) {
		set.overflow = true;
		return;
	}
	// Into the last slot, which is free, then down to its place
	set.leaf_3 = leaf;
	++set.count;
	uint32_t low;
	low = (set.leaf_2 < set.leaf_3) ? set.leaf_2 : set.leaf_3; set.leaf_3 = (set.leaf_2 < set.leaf_3) ? set.leaf_3 : set.leaf_2; set.leaf_2 = low;
	low = (set.leaf_1 < set.leaf_2) ? set.leaf_1 : set.leaf_2; set.leaf_2 = (set.leaf_1 < set.leaf_2) ? set.leaf_2 : set.leaf_1; set.leaf_1 = low;
	low = (set.leaf_0 < set.leaf_1) ? set.leaf_0 : set.leaf_1; set.leaf_1 = (set.leaf_0 < set.leaf_1) ? set.leaf_1 : set.leaf_0; set.leaf_0 = low;
}

// One axis of a ray's way through the grid over the lights
struct PtGridAxis {
	float t_next, t_step; // where the ray enters the next cell along this axis, and the distance between such places
	uint32_t remaining;   // cells ahead along this axis
	uint32_t stride;      // what a step adds to the cell's index (wraps around for steps backwards)
};

// Clips the ray to the slab of the grid along one axis (false: it misses the
// slab). The grid ends two margins beyond the outermost light, so a ray that
// rounding lets slip past it meets no light.
PT_FN bool pt_grid_clip(float origin, float dir, float lo, float hi, float& t_enter, float& t_leave) {
	if (dir != 0.0f) {
		float inv = 1.0f / dir;
		float t_0 = (lo - origin) * inv, t_1 = (hi - origin) * inv;
		t_enter = fmaxf(t_enter, fminf(t_0, t_1));
		t_leave = fminf(t_leave, fmaxf(t_0, t_1));
		return true;
	}
	return origin >= lo && origin <= hi;
}

PT_FN PtGridAxis pt_grid_axis(float origin, float dir, float t_enter, float lo, float cell, float inv_cell, uint32_t dim, uint32_t stride, uint32_t& index) {
	float position = fmaf(t_enter, dir, origin);
	float coordinate = fminf(fmaxf(floorf((position - lo) * inv_cell), 0.0f), (float) (dim - 1));
	uint32_t i = (uint32_t) coordinate;
	index += i * stride;
	PtGridAxis axis;
	const bool forward = dir > 0.0f;
	axis.remaining = forward ? (dim - 1u - i) : i;
	axis.stride = forward ? stride : (0u - stride);
	if (dir != 0.0f) {
		float inv = 1.0f / dir;
		axis.t_next = (fmaf(coordinate + (forward ? 1.0f : 0.0f), cell, lo) - origin) * inv;
		axis.t_step = cell * fabsf(inv);
	}
	else {
		axis.t_next = 3.0e38f;
		axis.t_step = 0.0f;
	}
	return axis;
}

// A ray's place in the grid over the lights: its cell, the lights of that cell
// still to be tested, and what the steps to the next cells need
struct PtGridWalk {
	PtGridAxis x, y, z;
	uint32_t cell, entry, entry_end;
};

// The place where the ray from origin along dir (of any length) enters the
// grid. A ray that misses the grid gets a place without lights and without
// cells ahead.
PT_FN PtGridWalk pt_light_grid_begin(const PtSceneView& scene, v3 origin, v3 dir) {
	const uint32_t dim_x = scene.light_grid_dim[0], dim_y = scene.light_grid_dim[1], dim_z = scene.light_grid_dim[2];
	PtGridWalk walk;
	float t_enter = 0.0f, t_leave = 3.0e38f;
	bool meets = pt_grid_clip(origin.x, dir.x, scene.light_grid_lo[0], fmaf((float) dim_x, scene.light_grid_cell[0], scene.light_grid_lo[0]), t_enter, t_leave);
	meets = pt_grid_clip(origin.y, dir.y, scene.light_grid_lo[1], fmaf((float) dim_y, scene.light_grid_cell[1], scene.light_grid_lo[1]), t_enter, t_leave) && meets;
	meets = pt_grid_clip(origin.z, dir.z, scene.light_grid_lo[2], fmaf((float) dim_z, scene.light_grid_cell[2], scene.light_grid_lo[2]), t_enter, t_leave) && meets;
	meets = meets && t_enter <= t_leave;
	// For a ray that misses, t_enter may be anything, NaN included; the cell
	// found for it exists all the same
	walk.cell = 0;
	walk.x = pt_grid_axis(origin.x, dir.x, t_enter, scene.light_grid_lo[0], scene.light_grid_cell[0], scene.light_grid_inv_cell[0], dim_x, 1u, walk.cell);
	walk.y = pt_grid_axis(origin.y, dir.y, t_enter, scene.light_grid_lo[1], scene.light_grid_cell[1], scene.light_grid_inv_cell[1], dim_y, dim_x, walk.cell);
	walk.z = pt_grid_axis(origin.z, dir.z, t_enter, scene.light_grid_lo[2], scene.light_grid_cell[2], scene.light_grid_inv_cell[2], dim_z, dim_x * dim_y, walk.cell);
	walk.entry = scene.light_grid_cells[walk.cell];
	walk.entry_end = scene.light_grid_cells[walk.cell + 1];
	if (!meets) {
		walk.x.remaining = walk.y.remaining = walk.z.remaining = 0u;
		walk.entry_end = walk.entry;
	}
	return walk;
}

// Adds to the set the leaves of all lights that the rays from origin along
// dir_0 and along dir_1 meet, found by following the rays through the grid
// over the lights cell by cell, the second ray after the first. All of it is
// one loop, and a pass of it tests one light of the current cell, if there is
// one left, and then moves on if there is none left: the threads of a
// wavefront, whose rays pass different numbers of cells and lights, thus wait
// for the longest sum of the two rays' work, not for the longest cell, the
// longest first ray and the longest second ray one after the other.
PT_FN void pt_light_grid_collect_2(const PtSceneView& scene, v3 origin, v3 dir_0, v3 dir_1, PtLeafSet& set) {
	v3 dir = dir_0;
	float dir_length_2 = dot(dir, dir);
	bool second = false;
	PtGridWalk walk = pt_light_grid_begin(scene, origin, dir);
	// Every pass tests a light or takes a step or ends a ray, and there are at
	// most 384 steps per ray and 2^26 listed lights; the bound only keeps an
	// error from hanging the GPU
	for (uint32_t pass = 0; pass != (1u << 28); ++pass) {
		if (walk.entry != walk.entry_end) {
#ifdef PT_GRID_STATS // counters of scripts/light_grid_stats.cpp
			++g_tests;
#endif
			if (pt_ray_meets_light(scene.light_grid_lights[walk.entry], origin, dir, dir_length_2))
				pt_leaf_set_insert(set, scene.light_grid_leaves[walk.entry]);
			++walk.entry;
		}
		if (walk.entry == walk.entry_end) {
			// Into the neighbour across the nearest cell boundary ahead
			const bool step_x = walk.x.t_next <= walk.y.t_next && walk.x.t_next <= walk.z.t_next;
			const bool step_y = !step_x && walk.y.t_next <= walk.z.t_next;
			const bool step_z = !step_x && !step_y;
			if ((step_x ? walk.x.remaining : (step_y ? walk.y.remaining : walk.z.remaining)) != 0u) {
#ifdef PT_GRID_STATS
				++g_steps;
#endif
				walk.cell += step_x ? walk.x.stride : (step_y ? walk.y.stride : walk.z.stride);
				walk.x.t_next += step_x ? walk.x.t_step : 0.0f; walk.x.remaining -= step_x ? 1u : 0u;
				walk.y.t_next += step_y ? walk.y.t_step : 0.0f; walk.y.remaining -= step_y ? 1u : 0u;
				walk.z.t_next += step_z ? walk.z.t_step : 0.0f; walk.z.remaining -= step_z ? 1u : 0u;
				walk.entry = scene.light_grid_cells[walk.cell];
				walk.entry_end = scene.light_grid_cells[walk.cell + 1];
			}
			else if (!second) {
				second = true;
				dir = dir_1;
				dir_length_2 = dot(dir, dir);
				walk = pt_light_grid_begin(scene, origin, dir);
			}
			else
				break;
		}
	}
}

PT_FN void pt_lights_density_tree_walk(const PtSceneView& scene, const PtLightTreeSample& sample, v3 shading_pos, v3 normal, v3 dir_0, v3 dir_1, float* out_density_0, float* out_density_1);

// Densities of two directions under pt_sample_lights_tree(): for each, the sum
// over the lights that a ray along it meets of selection probability times
// cone density, in leaf order. The lights come from the grid if there is one;
// more than four of them, or a shading point far outside, go to the walk of
// the tree.
PT_FN void pt_lights_density_tree_2(const PtSceneView& scene, const PtLightTreeSample& sample, v3 shading_pos, v3 normal, v3 dir_0, v3 dir_1, float* out_density_0, float* out_density_1) {
	if (scene.light_grid_cells && (fabsf(shading_pos.x) + fabsf(shading_pos.y)) + fabsf(shading_pos.z) <= scene.light_grid_reach) {
		PtLeafSet set;
		set.leaf_0 = set.leaf_1 = set.leaf_2 = set.leaf_3 = 0xffffffffu;
		set.count = 0;
		set.overflow = false;
		pt_light_grid_collect_2(scene, shading_pos, dir_0, dir_1, set);
		if (!set.overflow) {
			float density_0 = 0.0f, density_1 = 0.0f;
			if (set.count > 0) pt_light_leaf_density(scene, sample, set.leaf_0, shading_pos, normal, dir_0, dir_1, density_0, density_1);
			if (set.count > 1) pt_light_leaf_density(scene, sample, set.leaf_1, shading_pos, normal, dir_0, dir_1, density_0, density_1);
			if (set.count > 2) pt_light_leaf_density(scene, sample, set.leaf_2, shading_pos, normal, dir_0, dir_1, density_0, density_1);
			if (set.count > 3) pt_light_leaf_density(scene, sample, set.leaf_3, shading_pos, normal, dir_0, dir_1, density_0, density_1);
			*out_density_0 = density_0;
			*out_density_1 = density_1;
			return;
		}
	}
	pt_lights_density_tree_walk(scene, sample, shading_pos, normal, dir_0, dir_1, out_density_0, out_density_1);
}

// The same by a walk of the tree. One depth-first walk serves both rays; a
// subtree is entered when one of the rays meets its box. Both children of a
// node are looked at together; a right child still to be entered is remembered
// as one bit per tree level (the node index itself tells the way back).
PT_FN void pt_lights_density_tree_walk(const PtSceneView& scene, const PtLightTreeSample& sample, v3 shading_pos, v3 normal, v3 dir_0, v3 dir_1, float* out_density_0, float* out_density_1) {
	float density_0 = 0.0f, density_1 = 0.0f;
	PtSlabRay ray_0 = pt_slab_ray(shading_pos, dir_0), ray_1 = pt_slab_ray(shading_pos, dir_1);
	f32x4 root_center = scene.light_nodes[2], root_half = scene.light_nodes[3];
	float slack = 1.0e-6f * ((((fabsf(shading_pos.x) + fabsf(shading_pos.y)) + fabsf(shading_pos.z))
		+ ((fabsf(root_center.x) + fabsf(root_center.y)) + fabsf(root_center.z))) + ((root_half.x + root_half.y) + root_half.z));
	PtLightLeaves leaves;
	leaves.node_0 = leaves.node_1 = leaves.node_2 = leaves.node_3 = leaves.count = 0;
	uint32_t node = 1, depth = 0, pending = 0;
	bool inside = pt_light_tree_visit(scene, 1, ray_0, ray_1, slack);
	if (inside && scene.light_leaf_count == 1) {
		// A tree of one light: the root is the leaf
		leaves.node_0 = 1;
		leaves.count = 1;
		inside = false;
	}
	while (inside) {
		bool enter_0 = pt_light_tree_visit(scene, 2 * node, ray_0, ray_1, slack);
		bool enter_1 = pt_light_tree_visit(scene, 2 * node + 1, ray_0, ray_1, slack);
		if (2 * node >= scene.light_leaf_count) {
			// The children are leaves: note those met, left before right
			for (uint32_t child = 0; child != 2; ++child) {
				if (child ? enter_1 : enter_0) {
					if (leaves.count == 4)
						pt_light_leaves_flush(scene, sample, leaves, shading_pos, normal, dir_0, dir_1, density_0, density_1);
					uint32_t leaf = 2 * node + child;
					if (leaves.count == 0) leaves.node_0 = leaf;
					else if (leaves.count == 1) leaves.node_1 = leaf;
					else if (leaves.count == 2) leaves.node_2 = leaf;
					else leaves.node_3 = leaf;
					++leaves.count;
				}
			}
			enter_0 = enter_1 = false;
		}
		if (enter_0 || enter_1) {
			pending |= (enter_0 && enter_1) ? (1u << depth) : 0u;
			node = 2 * node + (enter_0 ? 0u : 1u);
			++depth;
		}
		else if (pending != 0) {
			// Back to the deepest node whose right child waits, and into that child
			uint32_t level = 31u - (uint32_t) __builtin_clz(pending);
			pending &= ~(1u << level);
			node = 2 * (node >> (depth - level)) + 1;
			depth = level + 1;
		}
		else
			inside = false;
	}
	pt_light_leaves_flush(scene, sample, leaves, shading_pos, normal, dir_0, dir_1, density_0, density_1);
	*out_density_0 = density_0;
	*out_density_1 = density_1;
}

// What shading one path vertex produces (the body of one iteration of the
// loops at pathtrace.frag.glsl:224-362 after the extension ray has returned)
struct PtVertexResult {
	bool continues;   // an extension ray leaves this vertex
	v3 next_origin, next_dir;
	bool shadow;      // a shadow ray leaves this vertex (NEE only)
	v3 shadow_dir;
	v3 shadow_weight; // throughput * brdf
	float shadow_factor; // lambert / (light density + brdf density)
	float shadow_near;   // where along the shadow ray the sampled light begins (pt_light_near_distance())
	float next_cone;  // the cone of the extension ray, packed by pt_pack_cone()
};

// Shades vertex k (1-based) of a path. throughput, nee_throughput and the
// seed are the path state and are updated in place; emitted receives what
// the caller has to add to the path's radiance (the weighted emission of the
// surface or the sky). hit_prim == 0xffffffff means the extension ray left
// the scene.
PT_FN PtVertexResult pt_shade_vertex(const PtSceneView& scene, const PtConstants& c, const PtLightCache& light_cache, int strategy, uint32_t path_length, uint32_t k,
	uint32_t hit_prim, float hit_u, float hit_v, bool hit_front, v3 ray_origin, v3 ray_dir, float ray_cone,
	v3& emitted, v3& throughput, v3& nee_throughput, uint32_t& sx, uint32_t& sy)
{
	PtVertexResult r;
	r.continues = false;
	r.shadow = false;
	r.next_origin = r.next_dir = r.shadow_dir = r.shadow_weight = mk3(0.0f, 0.0f, 0.0f);
	r.shadow_factor = 0.0f;
	r.shadow_near = 0.0f;
	r.next_cone = 0.0f;
	// In the NEE estimator emission is weighted by nee_throughput, in the
	// others by throughput (pathtrace.frag.glsl:231,265,293,323)
	bool light_tree = strategy == PT_STRATEGY_NEE_TREE || strategy == PT_STRATEGY_NEE_TREE_FAST;
	bool next_event_estimation = strategy == PT_STRATEGY_NEE || strategy == PT_STRATEGY_NEE_FAST || light_tree;
	v3 emission_weight = next_event_estimation ? nee_throughput : throughput;
	if (hit_prim == 0xffffffffu) {
		emitted = emission_weight * mk3(c.sky_radiance[0], c.sky_radiance[1], c.sky_radiance[2]);
		return r;
	}
	if (k >= path_length) {
		// The last vertex of a path contributes its emission only
		// (pathtrace.frag.glsl:323-325 and the loop bound), and the emission
		// depends on nothing but the material (shading_data.glsl:100-101)
		bool emissive = scene.shade_tris[hit_prim].material == scene.emission_material_index;
		v3 emission = emissive ? mk3(c.emission_material_radiance[0], c.emission_material_radiance[1], c.emission_material_radiance[2]) : mk3(0.0f, 0.0f, 0.0f);
		emitted = emission_weight * emission;
		return r;
	}
	float cone_width, cone_spread;
	pt_unpack_cone(ray_cone, &cone_width, &cone_spread);
	PtShadingData s = pt_shading_data(scene, c, hit_prim, hit_u, hit_v, hit_front, -ray_dir, ray_origin, cone_width, cone_spread);
	emitted = emission_weight * s.emission;
	r.next_origin = s.pos;
	if (strategy == PT_STRATEGY_SPHERICAL || strategy == PT_STRATEGY_PSA) {
		// pathtrace.frag.glsl:234-240 and :268-274
		PtFrame frame = pt_shading_space(s.normal);
		v2 randoms = pt_random(sx, sy);
		v3 sampled_dir = (strategy == PT_STRATEGY_SPHERICAL) ? pt_sample_hemisphere_spherical(randoms) : pt_sample_hemisphere_psa(randoms);
		r.next_dir = pt_to_world(frame, sampled_dir);
		float lambert_in = sampled_dir.z;
		float density = (strategy == PT_STRATEGY_SPHERICAL) ? pt_hemisphere_spherical_density(sampled_dir.z) : pt_hemisphere_psa_density(sampled_dir.z);
		throughput = throughput * ((pt_frostbite_brdf(s, r.next_dir) * lambert_in) / density);
		r.next_cone = pt_pack_cone(s.footprint, pt_cone_spread_after_bounce(cone_spread, density));
		r.continues = true;
	}
	else if (strategy == PT_STRATEGY_BRDF) {
		// pathtrace.frag.glsl:296-304
		r.next_dir = pt_sample_frostbite_brdf(s, pt_random(sx, sy));
		float density = pt_frostbite_brdf_density(s, r.next_dir);
		float lambert_in = dot(s.normal, r.next_dir);
		if (lambert_in <= 0.0f)
			return r;
		throughput = throughput * ((pt_frostbite_brdf(s, r.next_dir) * lambert_in) / density);
		r.next_cone = pt_pack_cone(s.footprint, pt_cone_spread_after_bounce(cone_spread, density));
		r.continues = true;
	}
	else {
		// pathtrace.frag.glsl:326-355. Both directions are sampled first (the
		// light's random numbers are drawn before the BRDF's, as in the
		// reference) so that one loop over the lights yields both densities.
		float total_light_importance = 0.0f, sampled_light_density = 0.0f;
		PtLightTreeSample tree_sample;
		tree_sample.leaf = 0;
		tree_sample.probability = 0.0f;
		v3 light_dir = light_tree
			? pt_sample_lights_tree(scene, &sampled_light_density, &tree_sample, s.pos, s.normal, pt_random(sx, sy), &r.shadow_near)
			: pt_sample_lights(scene, light_cache, &total_light_importance, s.pos, s.normal, pt_random(sx, sy), strategy == PT_STRATEGY_NEE_FAST, &r.shadow_near);
		float lambert_in_0 = dot(s.normal, light_dir);
		r.next_dir = pt_sample_frostbite_brdf(s, pt_random(sx, sy));
		float lambert_in_1 = dot(s.normal, r.next_dir);
		float light_density_0 = 0.0f, light_density_1 = 0.0f;
		if (lambert_in_0 > 0.0f || lambert_in_1 > 0.0f) {
			if (light_tree) {
				pt_lights_density_tree_2(scene, tree_sample, s.pos, s.normal, light_dir, r.next_dir, &light_density_0, &light_density_1);
				// Like max(1, count) at pathtrace.frag.glsl:172-173: the sampled
				// light counts even if rounding lets its own ray miss it
				light_density_0 = fmaxf(light_density_0, sampled_light_density);
			}
			else
				pt_lights_density_2(scene, total_light_importance, s.pos, light_dir, r.next_dir, &light_density_0, &light_density_1);
		}
		if (lambert_in_0 > 0.0f) {
			float brdf_density_0 = pt_frostbite_brdf_density(s, light_dir);
			r.shadow = true;
			r.shadow_dir = light_dir;
			r.shadow_weight = throughput * pt_frostbite_brdf(s, light_dir);
			r.shadow_factor = lambert_in_0 / (light_density_0 + brdf_density_0);
		}
		if (lambert_in_1 <= 0.0f)
			return r;
		float brdf_density_1 = pt_frostbite_brdf_density(s, r.next_dir);
		v3 brdf_lambert_1 = pt_frostbite_brdf(s, r.next_dir) * lambert_in_1;
		nee_throughput = throughput * (brdf_lambert_1 / (light_density_1 + brdf_density_1));
		throughput = throughput * (brdf_lambert_1 * (1.0f / brdf_density_1));
		r.next_cone = pt_pack_cone(s.footprint, pt_cone_spread_after_bounce(cone_spread, brdf_density_1));
		r.continues = true;
	}
	return r;
}


// ---- Test hook: one sampler, density or BRDF of brdfs.glsl /
// pathtrace.frag.glsl for explicit arguments, so that tests can check
// normalisation (the integral of a density is 1) and that every sampler
// follows its density without rendering anything. shading_data holds the 20
// floats of shading_data_t (shading_data.glsl:19-39), input and output four
// floats each; the functions are numbered like PT_FUNCTION_* in pt_hip.h.
PT_FN void pt_evaluate_function(const PtSceneView& scene, int function, const float* shading_data, const float* input, float* output) {
	PtShadingData s;
	s.pos = mk3(shading_data[0], shading_data[1], shading_data[2]);
	s.normal = mk3(shading_data[3], shading_data[4], shading_data[5]);
	s.out_dir = mk3(shading_data[6], shading_data[7], shading_data[8]);
	s.lambert_out = shading_data[9];
	s.emission = mk3(shading_data[10], shading_data[11], shading_data[12]);
	s.diffuse_albedo = mk3(shading_data[13], shading_data[14], shading_data[15]);
	s.fresnel_0 = mk3(shading_data[16], shading_data[17], shading_data[18]);
	s.roughness = shading_data[19];
	s.footprint = 0.0f;
	v2 randoms = mk2(input[0], input[1]);
	v3 dir = mk3(input[0], input[1], input[2]);
	PtFrame frame = pt_shading_space(s.normal);
	v3 result = mk3(0.0f, 0.0f, 0.0f);
	float last = 0.0f;
	PtLightCache no_cache;
	no_cache.values = NULL;
	no_cache.stride = no_cache.capacity = 0;
	switch (function) {
	case 0: result = pt_to_world(frame, pt_sample_ggx_in_dir(pt_to_local(frame, s.out_dir), s.roughness, randoms)); break;
	case 1: result.x = pt_ggx_in_dir_density(s.lambert_out, s.out_dir, dir, s.normal, s.roughness); break;
	case 2: result = pt_to_world(frame, pt_sample_hemisphere_psa(randoms)); break;
	case 3: result.x = pt_hemisphere_psa_density(dot(s.normal, dir)); break;
	case 4: result = pt_to_world(frame, pt_sample_hemisphere_spherical(randoms)); break;
	case 5: result.x = pt_hemisphere_spherical_density(dot(s.normal, dir)); break;
	case 6: result = pt_sample_frostbite_brdf(s, randoms); break;
	case 7: result.x = pt_frostbite_brdf_density(s, dir); break;
	case 8: result = pt_frostbite_brdf(s, dir); break;
	case 9: result = pt_sample_lights(scene, no_cache, &last, s.pos, s.normal, randoms); break;
	case 10: {
		// get_lights_density() for a direction towards a light (x) and for any
		// other direction (y); z is the total importance it was given
		float total = 0.0f;
		for (uint32_t i = 0; i != scene.light_count; ++i)
			total += pt_spherical_light_importance(pt_light(scene, i), s.pos, s.normal);
		pt_lights_density_2(scene, total, s.pos, dir, dir, &result.x, &result.y);
		result.z = total;
		break;
	}
	default: break;
	}
	output[0] = result.x;
	output[1] = result.y;
	output[2] = result.z;
	output[3] = last;
}
