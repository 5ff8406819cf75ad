// Shared definitions for the MI355X path tracer kernels: vector math with a
// fixed order of floating-point operations, elementary functions and the
// memory layouts of everything that lives in HBM.
//
// Every function here is plain C++ over scalars so that, besides hipcc for
// gfx950, a host compiler can run the per-thread kernel bodies in unit tests
// on machines without a GPU. The product library only ever runs them on the
// GPU.
//
// Build with -ffp-contract=off. A fused multiply-add appears exactly where
// the reference's GLSL calls fma() (as fmaf()); everything else rounds after
// every operation. Conventions where GLSL leaves precision open:
//   dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z,  normalize(v) = v * (1/sqrt(dot(v,v)))
//   M * v = (col0*v.x + col1*v.y) + col2*v.z,  sin/cos/log = pt_sincos()/pt_log()
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_FN __host__ __device__ __forceinline__
#else
#define PT_FN inline
#endif

#define PT_PI 3.141592653589793238462643f

struct v3 { float x, y, z; };
struct v2 { float x, y; };
struct alignas(16) f32x4 { float x, y, z, w; };
struct alignas(16) u32x4 { uint32_t x, y, z, w; };
struct alignas(8) u32x2 { uint32_t x, y; };

PT_FN v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_FN v2 mk2(float x, float y) { v2 r; r.x = x; r.y = y; return r; }
PT_FN f32x4 mk4(float x, float y, float z, float w) { f32x4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
PT_FN v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_FN v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_FN v3 operator*(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_FN v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_FN v3 operator/(v3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
PT_FN v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }
PT_FN float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PT_FN v3 cross(v3 a, v3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PT_FN v3 normalize(v3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
PT_FN float comp(v3 v, int i) { return (i == 0) ? v.x : ((i == 1) ? v.y : v.z); }

PT_FN float pt_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
PT_FN uint32_t pt_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// sin and cos: Cody-Waite reduction to [-pi/4, pi/4], then the Cephes
// single-precision minimax polynomials in explicit fmaf form
PT_FN void pt_sincos(float x, float* out_sin, float* out_cos) {
	float kf = rintf(x * 0.636619772367581343f);
	int k = (int) kf;
	float r = fmaf(kf, -1.5703125f, x);
	r = fmaf(kf, -4.837512969970703125e-4f, r);
	r = fmaf(kf, -7.54978995489188216e-8f, r);
	float z = r * r;
	float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
	ps = fmaf(ps, z, -1.6666654611e-1f);
	float s = fmaf(ps * z, r, r);
	float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
	pc = fmaf(pc, z, 4.166664568298827e-2f);
	float c = fmaf(pc, z * z, fmaf(-0.5f, z, 1.0f));
	int q = k & 3;
	float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
	*out_sin = (q & 2) ? -ss : ss;
	*out_cos = ((q + 1) & 2) ? -cc : cc;
}

// Natural logarithm of a positive normal float (Cephes logf form)
PT_FN float pt_log(float x) {
	uint32_t ix = pt_as_uint(x);
	int e = (int) (ix >> 23) - 127;
	float f = pt_as_float((ix & 0x007fffffu) | 0x3f800000u);
	if (f > 1.41421356237f) {
		f *= 0.5f;
		e += 1;
	}
	f -= 1.0f;
	float z = f * f;
	float p = 7.0376836292e-2f;
	p = fmaf(p, f, -1.1514610310e-1f);
	p = fmaf(p, f, 1.1676998740e-1f);
	p = fmaf(p, f, -1.2420140846e-1f);
	p = fmaf(p, f, 1.4249322787e-1f);
	p = fmaf(p, f, -1.6668057665e-1f);
	p = fmaf(p, f, 2.0000714765e-1f);
	p = fmaf(p, f, -2.4999993993e-1f);
	p = fmaf(p, f, 3.3333331174e-1f);
	float fe = (float) e;
	float y = (p * f) * z;
	y = fmaf(-2.12194440e-4f, fe, y);
	y = fmaf(-0.5f, z, y);
	float r = f + y;
	return fmaf(0.693359375f, fe, r);
}


// ---- layouts ----

// The 784-byte uniform block of the reference (src/shaders/constants.glsl,
// constants_t at src/main.h:237-256), std140 with row_major matrices
struct PtConstants {
	float world_to_projection_space[16];
	float projection_to_world_space[16];
	float camera_pos[3];
	int32_t camera_type;
	float hemispherical_camera_normal[3];
	float pad_1;
	float dequantization_factor[3], pad_2, dequantization_summand[3], pad_3;
	float viewport_size[2], inv_viewport_size[2];
	float exposure;
	uint32_t frame_index;
	uint32_t accum_frame_count;
	float pad_4;
	float sky_radiance[3];
	float pad_5;
	float emission_material_radiance[3];
	float pad_6;
	float params[4];
	float spherical_lights[32][4];
};
static_assert(sizeof(PtConstants) == 784, "constants block must match the reference's 784 bytes");

// 8-wide BVH node with child boxes quantised to 8 bits on a per-node
// power-of-two grid (after Ylitie, Karras, Laine: "Efficient incoherent ray
// traversal on GPUs through compressed wide BVHs", HPG 2017). 80 bytes.
// Child i spans origin + q * 2^(e - 127) per axis. Bit i of imask: child i is
// an inner node; bit i of leafmask: child i is one triangle; neither: empty
// slot. Inner children are consecutive nodes from child_base and triangles
// consecutive from tri_base, both in slot order, so a child's index is the base
// plus the number of set mask bits below its slot.
struct alignas(16) PtNode {
	float origin[3];
	uint8_t exponent[3];
	uint8_t imask;
	uint32_t child_base;
	uint32_t tri_base;
	uint8_t leafmask;
	uint8_t pad[7];
	uint8_t qlo[3][8];
	uint8_t qhi[3][8];
};
static_assert(sizeof(PtNode) == 80, "node layout");

// A triangle as the traversal kernel reads it: the float vertices that the
// reference feeds to the acceleration structure build (src/scene.c:43-55)
// plus its index in the scene file. 48 bytes.
struct alignas(16) PtTri {
	float v0[3], v1[3], v2[3];
	uint32_t prim;
	uint32_t pad[2];
};
static_assert(sizeof(PtTri) == 48, "triangle layout");

#define PT_MAX_MIPS 16
// One material texture; mip payloads are raw BC1/BC5 blocks in one device
// blob (format: VkFormat number 131 BC1_RGB_UNORM, 132 BC1_RGB_SRGB, 141
// BC5_UNORM, as in the .vkt header read by src/textures.c:71-84)
// x / 255 and x / 65535 for the integers that 8- and 16-bit unorm texels
// hold, bit-identical to the IEEE quotient for every such x (checked
// exhaustively by the tests): a product with the rounded reciprocal, corrected
// once through its exact residual. Three instructions on the GPU where the
// correctly rounded division takes ten.
PT_FN float pt_unorm8(uint32_t x) {
	const float r = 1.0f / 255.0f;
	float f = (float) x, q = f * r;
	return fmaf(fmaf(-q, 255.0f, f), r, q);
}
PT_FN float pt_unorm16(uint32_t x) {
	const float r = 1.0f / 65535.0f;
	float f = (float) x, q = f * r;
	return fmaf(fmaf(-q, 65535.0f, f), r, q);
}

struct PtTexture {
	uint32_t width, height, format, mip_count;
	uint64_t mip_offsets[PT_MAX_MIPS];
};

// What the shading kernels read per texture, small enough that the table of a
// whole scene (at most 768 textures) stays in the vector L1: the offset of
// the texture's texels in the blob in units of 16 bytes (PtTexture.mip_offsets[0]
// holds it; the other levels follow level 0 without gaps), and
// log2 width | log2 height << 8 | VkFormat << 16 | mip levels << 24
PT_FN u32x2 pt_pack_texture(const PtTexture& t) {
	uint32_t log_w = 0, log_h = 0;
	while ((1u << log_w) < t.width) ++log_w;
	while ((1u << log_h) < t.height) ++log_h;
	u32x2 e;
	e.x = (uint32_t) (t.mip_offsets[0] >> 4);
	e.y = log_w | (log_h << 8) | (t.format << 16) | (t.mip_count << 24);
	return e;
}

// Texels of a 2^log_w x 2^log_h texture that precede mip level `level` when
// the levels are stored back to back: the sum of 2^(log_w + log_h - 2 i) over
// i < level, which is (4^level - 1) / 3 * 2^(log_w + log_h + 2 - 2 level)
PT_FN uint64_t pt_mip_texel_offset(uint32_t log_w, uint32_t log_h, uint32_t level) {
	uint64_t thirds = 0x5555555555555555ull & ((1ull << (2 * level)) - 1ull);
	return thirds << (log_w + log_h + 2 - 2 * level);
}

// Everything that shading reads of a triangle, in one 64-byte line: the
// vertex records of the .vks arrays (src/scene.h:24-38) side by side
struct PtShadeTri {
	u32x2 positions[3];
	u32x2 normals_uvs[3];
	uint32_t material;
	uint32_t pad[3];
};

// Everything a kernel needs to know about the scene
struct PtSceneView {
	const PtNode* nodes;
	const PtTri* tris;
	uint32_t triangle_count;
	const PtShadeTri* shade_tris; // mesh data as stored in the .vks file, regrouped per triangle
	const uint8_t* materials;   // 1 per triangle
	const u32x2* texture_table; // 3 per material, see pt_pack_texture()
	const uint8_t* texture_blob; // decoded texels of all textures, see pt_load_texel()
	const float* srgb_to_linear; // 256 entries
	const f32x4* lights;
	uint32_t light_count;
	uint32_t emission_material_index;
	// Light tree (pt_light_tree.h): two f32x4 per node, node 1 is the root, the
	// leaves are nodes light_leaf_count ... 2 * light_leaf_count - 1 and leaf j
	// holds tree_lights[j]
	const f32x4* light_nodes;
	const f32x4* tree_lights;
	uint32_t light_leaf_count;
	// A grid over the same lights, or light_grid_cells == NULL: which lights
	// can a ray meet? Cell (x, y, z) is entry c = (z * dim[1] + y) * dim[0] + x
	// of light_grid_cells; it lists the lights light_grid_lights[e], which are
	// the leaves light_grid_leaves[e] of the tree, for e from light_grid_cells[c]
	// up to light_grid_cells[c + 1]. A light is listed in every cell that its
	// box, widened by more than rounding can amount to for rays that start
	// within light_grid_reach (sum of absolute coordinates), touches.
	const uint32_t* light_grid_cells;
	const f32x4* light_grid_lights;
	const uint32_t* light_grid_leaves;
	float light_grid_lo[3], light_grid_cell[3], light_grid_inv_cell[3];
	uint32_t light_grid_dim[3];
	float light_grid_reach;
};

// Wavefront buffers. Queues are indexed by slot, path state by path index.
struct PtWavefront {
	// extension ray queues (ping-pong): origin.xyz + path index, direction.xyz
	f32x4* ray_o[2];
	f32x4* ray_d[2];
	// closest hit per extension ray slot: triangle (or ~0), u, v, front
	u32x4* hit;
	// shadow ray queue: origin.xyz + path index, direction.xyz + scalar
	// factor, (throughput * brdf).xyz
	f32x4* sh_o;
	f32x4* sh_d;
	f32x4* sh_w;
	// path state: throughput.xyz + seed.x, nee_throughput.xyz + seed.y, radiance.xyz
	f32x4* state0;
	f32x4* state1;
	f32x4* radiance;
	// ext_count[k], shadow_count[k] for bounce k
	uint32_t* ext_count;
	uint32_t* shadow_count;
};

#define PT_MAX_PATH_LENGTH 16
#define PT_TILE 8

// The first four are the sampling strategies of the reference
// (src/main.h:83-94). The fifth is next event estimation whose light is chosen
// by a descent through the light tree instead of a loop over all lights; it
// has no counterpart in the reference, which stops at 32 lights.
// NEE_FAST is NEE with the hardware's approximate reciprocal and square root (one unit in the last place) in the
// loops over all lights, where NEE rounds every quotient and root correctly; it exists on the GPU only.
// NEE_TREE_FAST is NEE_TREE in the translation unit that is built with a shader compiler's freedoms
enum { PT_STRATEGY_SPHERICAL = 0, PT_STRATEGY_PSA = 1, PT_STRATEGY_BRDF = 2, PT_STRATEGY_NEE = 3, PT_STRATEGY_NEE_TREE = 4, PT_STRATEGY_NEE_FAST = 5, PT_STRATEGY_NEE_TREE_FAST = 6 };

// How one batch of paths maps to pixels and frames: path p belongs to frame
// slot p / paths_per_frame and to the 8x8 pixel tile (p % paths_per_frame) / 64
struct PtBatch {
	uint32_t width, height;
	uint32_t tiles_x, tiles_y;
	uint32_t paths_per_frame; // tiles_x * tiles_y * 64
	uint32_t frame_count;
	uint32_t first_frame_index, frame_stride;
};

PT_FN bool pt_path_to_pixel(const PtBatch& b, uint32_t path, uint32_t* x, uint32_t* y, uint32_t* frame_slot) {
	uint32_t f = path / b.paths_per_frame;
	uint32_t r = path - f * b.paths_per_frame;
	uint32_t tile = r >> 6, in_tile = r & 63;
	uint32_t ty = tile / b.tiles_x, tx = tile - ty * b.tiles_x;
	*x = tx * PT_TILE + (in_tile & 7);
	*y = ty * PT_TILE + (in_tile >> 3);
	*frame_slot = f;
	return *x < b.width && *y < b.height;
}
