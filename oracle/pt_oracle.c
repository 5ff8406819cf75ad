/* CPU oracle: scalar restatement of the reference's light transport.
 * TEST INFRASTRUCTURE ONLY - see pt_oracle.h for what it is and how far its
 * parity with the reference is pinned ("parity unpinned" for ray/triangle
 * intersection and texture filtering, which the reference leaves to the
 * Vulkan driver).
 *
 * Build with -ffp-contract=off: every fused multiply-add below is an explicit
 * fmaf() exactly where the GLSL says fma(); every other a*b+c must stay two
 * roundings so that the HIP kernels (built the same way) agree bit for bit.
 *
 * Float conventions chosen where GLSL leaves precision open:
 *   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   normalize(v)  = v * (1.0f / sqrtf(dot(v,v)))   (IEEE sqrt and divide)
 *   M * v         = (col0*v.x + col1*v.y) + col2*v.z
 *   sin/cos/log   = the polynomial forms in o_sincos()/o_log() below
 */
#include "pt_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define M_PI_F 3.141592653589793238462643f

typedef struct { float x, y, z; } vec3;
typedef struct { float x, y; } vec2;

static inline vec3 v3(float x, float y, float z) { vec3 r = { x, y, z }; return r; }
static inline vec3 add3(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 sub3(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 mul3(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 scale3(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 neg3(vec3 a) { return v3(-a.x, -a.y, -a.z); }
static inline vec3 div3(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline float dot3(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline vec3 cross3(vec3 a, vec3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline vec3 normalize3(vec3 a) { return scale3(a, 1.0f / sqrtf(dot3(a, a))); }
static inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }


/* ---- elementary functions (definitions shared with the HIP kernels) ---- */

/* sin and cos by Cody-Waite reduction to [-pi/4, pi/4] and the Cephes
 * single-precision minimax polynomials, in explicit fmaf form. */
void oracle_sincos(float x, float* out_sin, float* out_cos) {
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
	switch (k & 3) {
		case 0: *out_sin = s; *out_cos = c; break;
		case 1: *out_sin = c; *out_cos = -s; break;
		case 2: *out_sin = -s; *out_cos = -c; break;
		default: *out_sin = -c; *out_cos = s; break;
	}
}

/* Natural logarithm of a positive normal float (Cephes logf form). */
float oracle_log(float x) {
	uint32_t ix = float_to_bits(x);
	int e = (int) (ix >> 23) - 127;
	float f = bits_to_float((ix & 0x007fffffu) | 0x3f800000u);
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

#define o_sincos oracle_sincos
#define o_log oracle_log


/* ---- RNG: pathtrace.frag.glsl:23-34 ---- */
static inline vec2 get_random_numbers(uint32_t seed[2]) {
	seed[0] = 1664525u * seed[0] + 1013904223u;
	seed[1] = 1664525u * seed[1] + 1013904223u;
	seed[0] += 1664525u * seed[1];
	seed[1] += 1664525u * seed[0];
	seed[0] ^= seed[0] >> 16;
	seed[1] ^= seed[1] >> 16;
	seed[0] += 1664525u * seed[1];
	seed[1] += 1664525u * seed[0];
	seed[0] ^= seed[0] >> 16;
	seed[1] ^= seed[1] >> 16;
	vec2 r = { (float) seed[0] * 2.32830643654e-10f, (float) seed[1] * 2.32830643654e-10f };
	return r;
}

/* pathtrace.frag.glsl:38-45 */
float oracle_erfinv(float x) {
	float w = -o_log(fmaxf(1.0e-37f, 1.0f - x * x));
	float a = w - 2.5f;
	float b = sqrtf(w) - 3.0f;
	return x * ((w < 5.0f)
		? fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(2.81022636e-08f, a, 3.43273939e-07f), a, -3.5233877e-06f), a, -4.39150654e-06f), a, 0.00021858087f), a, -0.00125372503f), a, -0.00417768164f), a, 0.246640727f), a, 1.50140941f)
		: fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(-0.000200214257f, b, 0.000100950558f), b, 0.00134934322f), b, -0.00367342844f), b, 0.00573950773f), b, -0.0076224613f), b, 0.00943887047f), b, 1.00167406f), b, 2.83297682f));
}


/* ---- scene representation ---- */

typedef struct {
	float lo[3], hi[3];
	/* internal: left child index (right = left + 1), count = 0;
	   leaf: first triangle slot, count > 0 */
	uint32_t first, count;
} bvh_node_t;

struct oracle_scene {
	uint64_t material_count, triangle_count;
	float factor[3], summand[3];
	uint32_t* positions;
	uint16_t* normals_uvs;
	uint8_t* materials;
	uint32_t texture_count;
	oracle_texture_t* textures;
	uint8_t** texture_payloads;
	uint64_t** texture_offsets;
	/* BVH over the CPU-dequantised float vertices of src/scene.c:38-55 */
	float* bvh_vertices; /* 9 floats per triangle */
	uint32_t* bvh_triangle_order;
	bvh_node_t* nodes;
	uint32_t node_count;
	float srgb_to_linear[256];
};


/* ---- BVH build (binary, median split; only needs to be exact, not fast) ---- */

typedef struct {
	oracle_scene_t* scene;
	float* centroids;
	float* tri_lo;
	float* tri_hi;
	int axis;
} bvh_builder_t;

static bvh_builder_t* g_sort_builder;
static int compare_centroids(const void* a, const void* b) {
	uint32_t ia = *(const uint32_t*) a, ib = *(const uint32_t*) b;
	float ca = g_sort_builder->centroids[3 * ia + g_sort_builder->axis];
	float cb = g_sort_builder->centroids[3 * ib + g_sort_builder->axis];
	if (ca < cb) return -1;
	if (ca > cb) return 1;
	return (ia < ib) ? -1 : (ia > ib);
}

static void build_node(bvh_builder_t* b, uint32_t node_index, uint32_t first, uint32_t count) {
	oracle_scene_t* s = b->scene;
	bvh_node_t* node = &s->nodes[node_index];
	float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (int j = 0; j != 3; ++j) { node->lo[j] = INFINITY; node->hi[j] = -INFINITY; }
	for (uint32_t i = first; i != first + count; ++i) {
		uint32_t t = s->bvh_triangle_order[i];
		for (int j = 0; j != 3; ++j) {
			node->lo[j] = fminf(node->lo[j], b->tri_lo[3 * t + j]);
			node->hi[j] = fmaxf(node->hi[j], b->tri_hi[3 * t + j]);
			clo[j] = fminf(clo[j], b->centroids[3 * t + j]);
			chi[j] = fmaxf(chi[j], b->centroids[3 * t + j]);
		}
	}
	if (count <= 4) {
		node->first = first;
		node->count = count;
		return;
	}
	int axis = 0;
	if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
	if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
	b->axis = axis;
	g_sort_builder = b;
	qsort(s->bvh_triangle_order + first, count, sizeof(uint32_t), compare_centroids);
	uint32_t left_count = count / 2;
	uint32_t left = s->node_count;
	s->node_count += 2;
	node->first = left;
	node->count = 0;
	build_node(b, left, first, left_count);
	build_node(b, left + 1, first + left_count, count - left_count);
}

static void build_bvh(oracle_scene_t* s) {
	uint64_t n = s->triangle_count;
	s->bvh_vertices = malloc(sizeof(float) * 9 * (n ? n : 1));
	s->bvh_triangle_order = malloc(sizeof(uint32_t) * (n ? n : 1));
	s->nodes = malloc(sizeof(bvh_node_t) * (2 * n + 1));
	bvh_builder_t b = { .scene = s };
	b.centroids = malloc(sizeof(float) * 3 * (n ? n : 1));
	b.tri_lo = malloc(sizeof(float) * 3 * (n ? n : 1));
	b.tri_hi = malloc(sizeof(float) * 3 * (n ? n : 1));
	/* Vertex arithmetic of src/scene.c:43-55: separate multiply and add */
	float max_abs = 0.0f;
	for (uint64_t v = 0; v != 3 * n; ++v) {
		uint32_t a = s->positions[2 * v + 0], bb = s->positions[2 * v + 1];
		float pos[3] = {
			(float) (a & 0x1fffff),
			(float) (((a & 0xffe00000u) >> 21) | ((bb & 0x3ff) << 11)),
			(float) ((bb & 0x7ffffc00u) >> 10),
		};
		for (int j = 0; j != 3; ++j) {
			float w = pos[j] * s->factor[j] + s->summand[j];
			s->bvh_vertices[3 * v + j] = w;
			max_abs = fmaxf(max_abs, fabsf(w));
		}
	}
	/* Boxes are padded so that rounding in the slab test can never cull a
	   triangle that the triangle test would accept */
	float pad = max_abs * 1.9073486328125e-06f;
	for (uint64_t t = 0; t != n; ++t) {
		s->bvh_triangle_order[t] = (uint32_t) t;
		for (int j = 0; j != 3; ++j) {
			float v0 = s->bvh_vertices[9 * t + j], v1 = s->bvh_vertices[9 * t + 3 + j], v2 = s->bvh_vertices[9 * t + 6 + j];
			float lo = fminf(v0, fminf(v1, v2)), hi = fmaxf(v0, fmaxf(v1, v2));
			b.tri_lo[3 * t + j] = lo - pad;
			b.tri_hi[3 * t + j] = hi + pad;
			b.centroids[3 * t + j] = 0.5f * (lo + hi);
		}
	}
	s->node_count = 1;
	if (n)
		build_node(&b, 0, 0, (uint32_t) n);
	else {
		memset(&s->nodes[0], 0, sizeof(bvh_node_t));
		s->nodes[0].lo[0] = 1.0f; /* empty box */
	}
	free(b.centroids);
	free(b.tri_lo);
	free(b.tri_hi);
}


oracle_scene_t* oracle_create_scene(uint64_t material_count, uint64_t triangle_count,
	const float dequantization_factor[3], const float dequantization_summand[3],
	const uint32_t* quantized_positions, const uint16_t* normals_and_tex_coords, const uint8_t* material_indices,
	const oracle_texture_t* textures, uint32_t texture_count)
{
	oracle_scene_t* s = calloc(1, sizeof(*s));
	s->material_count = material_count;
	s->triangle_count = triangle_count;
	memcpy(s->factor, dequantization_factor, sizeof(s->factor));
	memcpy(s->summand, dequantization_summand, sizeof(s->summand));
	uint64_t n = triangle_count ? triangle_count : 1;
	s->positions = malloc(n * 24);
	s->normals_uvs = malloc(n * 24);
	s->materials = malloc(n);
	memcpy(s->positions, quantized_positions, triangle_count * 24);
	memcpy(s->normals_uvs, normals_and_tex_coords, triangle_count * 24);
	memcpy(s->materials, material_indices, triangle_count);
	s->texture_count = texture_count;
	s->textures = calloc(texture_count ? texture_count : 1, sizeof(oracle_texture_t));
	s->texture_payloads = calloc(texture_count ? texture_count : 1, sizeof(uint8_t*));
	s->texture_offsets = calloc(texture_count ? texture_count : 1, sizeof(uint64_t*));
	for (uint32_t i = 0; i != texture_count; ++i) {
		const oracle_texture_t* t = &textures[i];
		uint32_t block_size = (t->format == 141) ? 16 : 8;
		uint32_t mw = t->width >> (t->mip_count - 1), mh = t->height >> (t->mip_count - 1);
		if (mw < 4) mw = 4;
		if (mh < 4) mh = 4;
		uint64_t size = t->mip_offsets[t->mip_count - 1] + (uint64_t) ((mw + 3) / 4) * ((mh + 3) / 4) * block_size;
		s->texture_payloads[i] = malloc(size);
		memcpy(s->texture_payloads[i], t->payload, size);
		s->texture_offsets[i] = malloc(sizeof(uint64_t) * t->mip_count);
		memcpy(s->texture_offsets[i], t->mip_offsets, sizeof(uint64_t) * t->mip_count);
		s->textures[i] = *t;
		s->textures[i].payload = s->texture_payloads[i];
		s->textures[i].mip_offsets = s->texture_offsets[i];
	}
	for (int i = 0; i != 256; ++i) {
		double c = i / 255.0;
		s->srgb_to_linear[i] = (float) ((c <= 0.04045) ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
	}
	build_bvh(s);
	return s;
}

void oracle_destroy_scene(oracle_scene_t* s) {
	if (!s) return;
	for (uint32_t i = 0; i != s->texture_count; ++i) {
		free(s->texture_payloads[i]);
		free(s->texture_offsets[i]);
	}
	free(s->texture_payloads);
	free(s->texture_offsets);
	free(s->textures);
	free(s->positions);
	free(s->normals_uvs);
	free(s->materials);
	free(s->bvh_vertices);
	free(s->bvh_triangle_order);
	free(s->nodes);
	free(s);
}


/* ---- ray traversal: closest hit with the semantics of
   rayQueryInitializeEXT(..., gl_RayFlagsOpaqueEXT, 0xff, origin, 1e-3, dir, 1e38)
   at pathtrace.frag.glsl:184-185. The intersection arithmetic itself lives in
   the Vulkan driver in the reference (parity unpinned); defined here as the
   watertight test of Woop, Benthin and Wald (JCGT 2013) without its double
   precision fallback, closest t wins, equal t broken by the lower triangle
   index, t strictly inside (tmin, tmax). ---- */

typedef struct {
	int32_t triangle; /* -1: miss */
	float t, u, v;    /* u, v: weights of vertex 1 and 2 */
	int front;
} hit_t;

typedef struct {
	vec3 origin;
	int kx, ky, kz;
	float sx, sy, sz;
} ray_shear_t;

static inline float comp3(vec3 v, int i) { return (i == 0) ? v.x : ((i == 1) ? v.y : v.z); }

static ray_shear_t get_ray_shear(vec3 origin, vec3 dir) {
	ray_shear_t r;
	r.origin = origin;
	float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
	r.kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
	r.kx = (r.kz + 1) % 3;
	r.ky = (r.kx + 1) % 3;
	if (comp3(dir, r.kz) < 0.0f) {
		int swap = r.kx;
		r.kx = r.ky;
		r.ky = swap;
	}
	float dz = comp3(dir, r.kz);
	r.sx = comp3(dir, r.kx) / dz;
	r.sy = comp3(dir, r.ky) / dz;
	r.sz = 1.0f / dz;
	return r;
}

static inline void intersect_triangle(hit_t* hit, const ray_shear_t* r, const float* verts, uint32_t triangle, float tmin) {
	vec3 a = sub3(v3(verts[0], verts[1], verts[2]), r->origin);
	vec3 b = sub3(v3(verts[3], verts[4], verts[5]), r->origin);
	vec3 c = sub3(v3(verts[6], verts[7], verts[8]), r->origin);
	float az = comp3(a, r->kz), bz = comp3(b, r->kz), cz = comp3(c, r->kz);
	float ax = comp3(a, r->kx) - r->sx * az, ay = comp3(a, r->ky) - r->sy * az;
	float bx = comp3(b, r->kx) - r->sx * bz, by = comp3(b, r->ky) - r->sy * bz;
	float cx = comp3(c, r->kx) - r->sx * cz, cy = comp3(c, r->ky) - r->sy * cz;
	float u = cx * by - cy * bx;
	float v = ax * cy - ay * cx;
	float w = bx * ay - by * ax;
	if ((u < 0.0f || v < 0.0f || w < 0.0f) && (u > 0.0f || v > 0.0f || w > 0.0f))
		return;
	float det = (u + v) + w;
	if (det == 0.0f)
		return;
	float tz = (u * (r->sz * az) + v * (r->sz * bz)) + w * (r->sz * cz);
	float rcp_det = 1.0f / det;
	float t = tz * rcp_det;
	if (!(t > tmin))
		return;
	if (t < hit->t || (t == hit->t && (int32_t) triangle < hit->triangle)) {
		hit->t = t;
		hit->triangle = (int32_t) triangle;
		hit->u = v * rcp_det;
		hit->v = w * rcp_det;
		/* det is minus the signed area of the triangle seen along the ray,
		   so it has the sign of -dot(cross(v1 - v0, v2 - v0), dir); front
		   faces are those whose numeric normal points against the ray (in
		   the Cornell box file every stored vertex normal agrees with that
		   cross product, and the walls must shade as front faces) */
		hit->front = det > 0.0f;
	}
}

static hit_t trace_closest(const oracle_scene_t* s, vec3 origin, vec3 dir, float tmin, float tmax) {
	hit_t hit = { -1, tmax, 0.0f, 0.0f, 0 };
	if (!s->triangle_count)
		return hit;
	ray_shear_t shear = get_ray_shear(origin, dir);
	vec3 idir = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
	uint32_t stack[128];
	uint32_t sp = 0;
	stack[sp++] = 0;
	while (sp) {
		const bvh_node_t* n = &s->nodes[stack[--sp]];
		float t0x = (n->lo[0] - origin.x) * idir.x, t1x = (n->hi[0] - origin.x) * idir.x;
		float t0y = (n->lo[1] - origin.y) * idir.y, t1y = (n->hi[1] - origin.y) * idir.y;
		float t0z = (n->lo[2] - origin.z) * idir.z, t1z = (n->hi[2] - origin.z) * idir.z;
		/* The rounding of these differences, and of those in the triangle
		   test, grows with the distance between box and ray origin: the test
		   gives way by 2^-21 of the largest such distance (the padding of the
		   boxes only covers rays that start near the scene) */
		float reach = 0.0f;
		for (int j = 0; j != 3; ++j) {
			float o = (j == 0) ? origin.x : ((j == 1) ? origin.y : origin.z);
			reach = fmaxf(reach, fmaxf(fabsf(n->lo[j] - o), fabsf(n->hi[j] - o)));
		}
		float give = reach * 4.76837158203125e-07f;
		float gx = give * fabsf(idir.x), gy = give * fabsf(idir.y), gz = give * fabsf(idir.z);
		float tnear = fmaxf(fmaxf(fminf(t0x, t1x) - gx, fminf(t0y, t1y) - gy), fmaxf(fminf(t0z, t1z) - gz, tmin));
		float tfar = fminf(fminf(fmaxf(t0x, t1x) + gx, fmaxf(t0y, t1y) + gy), fminf(fmaxf(t0z, t1z) + gz, hit.t));
		if (!(tnear <= tfar))
			continue;
		if (n->count) {
			for (uint32_t i = n->first; i != n->first + n->count; ++i) {
				uint32_t t = s->bvh_triangle_order[i];
				intersect_triangle(&hit, &shear, &s->bvh_vertices[9 * t], t, tmin);
			}
		}
		else if (sp + 2 <= 128) {
			stack[sp++] = n->first;
			stack[sp++] = n->first + 1;
		}
	}
	if (hit.triangle < 0)
		hit.t = tmax;
	return hit;
}


/* ---- mesh_quantization.glsl ---- */

/* mesh_quantization.glsl:4-11 */
static inline vec3 dequantize_position(const uint32_t q[2], const float factor[3], const float summand[3]) {
	float px = (float) (q[0] & 0x1fffffu);
	float py = (float) (((q[0] >> 21) & 0x7ffu) | ((q[1] & 0x3ffu) << 11));
	float pz = (float) ((q[1] >> 10) & 0x1fffffu);
	return v3(fmaf(px, factor[0], summand[0]), fmaf(py, factor[1], summand[1]), fmaf(pz, factor[2], summand[2]));
}

/* mesh_quantization.glsl:16-31; the input is a pair of unorm16 values */
static inline vec3 dequantize_normal(float ox, float oy) {
	const float factor = 2.0f * (65534.0f / 65535.0f);
	const float summand = -(32768.0f / 65535.0f) * factor;
	ox = fmaf(ox, factor, summand);
	oy = fmaf(oy, factor, summand);
	vec3 normal = v3(ox, oy, (1.0f - fabsf(ox)) - fabsf(oy));
	float sign_x = (ox >= 0.0f) ? 1.0f : -1.0f;
	float sign_y = (oy >= 0.0f) ? 1.0f : -1.0f;
	if (normal.z < 0.0f) {
		float nx = (1.0f - fabsf(normal.y)) * sign_x;
		float ny = (1.0f - fabsf(normal.x)) * sign_y;
		normal.x = nx;
		normal.y = ny;
	}
	return normalize3(normal);
}


/* ---- texture sampling: texture() at shading_data.glsl:74-76 with the sampler
   of src/main.c:864-873 (linear min/mag/mip filters, REPEAT, all mip levels,
   anisotropy 16). Filtering and the choice of the level of detail are done by
   the Vulkan driver in the reference, from derivatives across a 2x2 pixel
   quad that are undefined once neighbouring paths diverge (parity unpinned).
   Defined here as trilinear filtering with float weights of hand-decoded
   blocks (8-bit integer palettes), at a level of detail that follows from the
   footprint of the path at the hit (a ray cone, see texture_lod_base()). ---- */

static inline void decode_bc1_texel(const uint8_t* block, uint32_t px, uint32_t py, uint8_t rgb[3]) {
	uint32_t c0 = block[0] | (block[1] << 8), c1 = block[2] | (block[3] << 8);
	uint32_t bits = block[4] | (block[5] << 8) | (block[6] << 16) | ((uint32_t) block[7] << 24);
	uint32_t code = (bits >> (2 * (4 * py + px))) & 3;
	uint32_t r0 = ((c0 >> 8) & 0xf8) | (c0 >> 13), g0 = ((c0 >> 3) & 0xfc) | ((c0 >> 9) & 3), b0 = ((c0 << 3) & 0xf8) | ((c0 >> 2) & 7);
	uint32_t r1 = ((c1 >> 8) & 0xf8) | (c1 >> 13), g1 = ((c1 >> 3) & 0xfc) | ((c1 >> 9) & 3), b1 = ((c1 << 3) & 0xf8) | ((c1 >> 2) & 7);
	uint32_t r, g, b;
	switch (code) {
		case 0: r = r0; g = g0; b = b0; break;
		case 1: r = r1; g = g1; b = b1; break;
		case 2:
			if (c0 > c1) { r = (2 * r0 + r1) / 3; g = (2 * g0 + g1) / 3; b = (2 * b0 + b1) / 3; }
			else { r = (r0 + r1) / 2; g = (g0 + g1) / 2; b = (b0 + b1) / 2; }
			break;
		default:
			if (c0 > c1) { r = (r0 + 2 * r1) / 3; g = (g0 + 2 * g1) / 3; b = (b0 + 2 * b1) / 3; }
			else { r = g = b = 0; }
			break;
	}
	rgb[0] = (uint8_t) r; rgb[1] = (uint8_t) g; rgb[2] = (uint8_t) b;
}

static inline uint32_t decode_bc4_texel(const uint8_t* block, uint32_t px, uint32_t py) {
	uint32_t a0 = block[0], a1 = block[1];
	uint64_t bits = 0;
	for (int i = 0; i != 6; ++i)
		bits |= (uint64_t) block[2 + i] << (8 * i);
	uint32_t code = (uint32_t) (bits >> (3 * (4 * py + px))) & 7;
	if (code == 0) return a0;
	if (code == 1) return a1;
	if (a0 > a1)
		return ((8 - code) * a0 + (code - 1) * a1) / 7;
	if (code == 6) return 0;
	if (code == 7) return 255;
	return ((6 - code) * a0 + (code - 1) * a1) / 5;
}

static vec3 fetch_texel(const oracle_scene_t* s, const oracle_texture_t* t, uint32_t mip, int32_t x, int32_t y) {
	uint32_t w = t->width >> mip, h = t->height >> mip;
	if (w < 1) w = 1;
	if (h < 1) h = 1;
	uint32_t ux = (uint32_t) x & (w - 1), uy = (uint32_t) y & (h - 1);
	uint32_t blocks_x = (w + 3) / 4;
	uint32_t block_size = (t->format == 141) ? 16 : 8;
	const uint8_t* block = t->payload + t->mip_offsets[mip] + ((uint64_t) (uy / 4) * blocks_x + ux / 4) * block_size;
	if (t->format == 141) {
		uint32_t r = decode_bc4_texel(block, ux & 3, uy & 3), g = decode_bc4_texel(block + 8, ux & 3, uy & 3);
		return v3((float) r / 255.0f, (float) g / 255.0f, 0.0f);
	}
	uint8_t rgb[3];
	decode_bc1_texel(block, ux & 3, uy & 3, rgb);
	if (t->format == 132)
		return v3(s->srgb_to_linear[rgb[0]], s->srgb_to_linear[rgb[1]], s->srgb_to_linear[rgb[2]]);
	return v3((float) rgb[0] / 255.0f, (float) rgb[1] / 255.0f, (float) rgb[2] / 255.0f);
}

static vec3 sample_texture_level(const oracle_scene_t* s, const oracle_texture_t* t, uint32_t mip, vec2 uv) {
	float x = uv.x * (float) (t->width >> mip) - 0.5f, y = uv.y * (float) (t->height >> mip) - 0.5f;
	float fx0 = floorf(x), fy0 = floorf(y);
	float wx = x - fx0, wy = y - fy0;
	int32_t x0 = (int32_t) fx0, y0 = (int32_t) fy0;
	vec3 t00 = fetch_texel(s, t, mip, x0, y0), t10 = fetch_texel(s, t, mip, x0 + 1, y0);
	vec3 t01 = fetch_texel(s, t, mip, x0, y0 + 1), t11 = fetch_texel(s, t, mip, x0 + 1, y0 + 1);
	vec3 top = add3(scale3(t00, 1.0f - wx), scale3(t10, wx));
	vec3 bottom = add3(scale3(t01, 1.0f - wx), scale3(t11, wx));
	return add3(scale3(top, 1.0f - wy), scale3(bottom, wy));
}

/* Trilinear lookup at level of detail lod (already including the size of the
   texture), clamped to the levels that the file holds */
static vec3 sample_texture_lod(const oracle_scene_t* s, uint32_t texture_index, vec2 uv, float lod) {
	const oracle_texture_t* t = &s->textures[texture_index];
	lod = fminf(fmaxf(lod, 0.0f), (float) (t->mip_count - 1));
	float lower = floorf(lod);
	uint32_t mip = (uint32_t) lower;
	float blend = lod - lower;
	vec3 fine = sample_texture_level(s, t, mip, uv);
	if (!(blend > 0.0f))
		return fine;
	vec3 coarse = sample_texture_level(s, t, mip + 1, uv);
	return add3(scale3(fine, 1.0f - blend), scale3(coarse, blend));
}

static uint32_t integer_log2(uint32_t v) {
	uint32_t l = 0;
	while ((1u << l) < v) ++l;
	return l;
}

/* The level of detail rule, defined by this build (the reference leaves it to
   the driver). A path has a footprint of some diameter where it hits a
   triangle. The triangle maps a doubled world area world_area_2 to a doubled
   area uv_area_2 of texture coordinates, so the footprint covers about
   uv_area_2 / world_area_2 * footprint^2 of the unit square, times width *
   height texels. The level at which that is one texel is
     0.5 * (log2(uv_area_2 / world_area_2 * footprint^2) + log2(width) + log2(height)).
   The footprint is measured across the ray (what an anisotropic filter uses
   as its level); only beyond the sampler's anisotropy of 16, i.e. where the
   cosine between normal and ray is below 1/16, it is stretched by
   1 / (16 cosine). This returns the part that does not depend on the
   texture size; very small or undefined coverage means level 0. */
static float texture_lod_base(float uv_area_2, float world_area_2, float footprint, float cosine) {
	float stretch = fmaxf(1.0f, 0.0625f / cosine);
	float diameter = footprint * stretch;
	float coverage = (uv_area_2 / world_area_2) * (diameter * diameter);
	if (!(coverage > 1.0e-30f))
		return -1.0e9f;
	return 0.5f * (o_log(fminf(coverage, 1.0e30f)) * 1.44269504088896341f);
}

static vec3 sample_texture(const oracle_scene_t* s, uint32_t texture_index, vec2 uv, float lod_base) {
	const oracle_texture_t* t = &s->textures[texture_index];
	float size_term = 0.5f * (float) (integer_log2(t->width) + integer_log2(t->height));
	return sample_texture_lod(s, texture_index, uv, lod_base + size_term);
}

/* Ray cones (defined by this build, see texture_lod_base()): a ray carries
   the diameter of the path's footprint at its origin and the angle by which
   it widens per unit distance, both cut to the upper 16 bits of their floats. */
typedef struct { float width, spread; } cone_t;

static float cut_to_upper_half(float v) { return bits_to_float(float_to_bits(v) & 0xffff0000u); }

static cone_t make_cone(float width, float spread) {
	cone_t c = { cut_to_upper_half(width), cut_to_upper_half(spread) };
	return c;
}

/* After a bounce the cone widens by 1 / sqrt(pi * density of the sampled
   direction), at most by 2, to at most 4 in total */
static cone_t cone_after_bounce(float footprint, float spread, float density) {
	float widening = fminf(2.0f, 1.0f / sqrtf(M_PI_F * density));
	return make_cone(footprint, fminf(4.0f, spread + widening));
}


/* ---- shading_data.glsl ---- */

typedef struct {
	vec3 pos, normal, out_dir;
	float lambert_out;
	vec3 emission, diffuse_albedo, fresnel_0;
	float roughness;
} shading_data_t;

typedef struct {
	const oracle_scene_t* scene;
	const oracle_constants_t* c;
	const float* lights;
	uint32_t light_count;
	/* Light tree (see build_light_tree()); light_sampling 1 selects it */
	int light_sampling;
	const float* light_nodes;
	const float* tree_lights;
	uint32_t light_leaf_count;
	uint32_t path_length;
	uint32_t emission_material_index;
	int strategy;
	uint64_t ray_count;
} context_t;

/* shading_data.glsl:51-103; cone_origin and cone describe the ray that led
   here and out_footprint receives the footprint of the path at the hit */
static shading_data_t get_shading_data(const context_t* ctx, int32_t triangle_index, vec2 barycentrics, int front, vec3 out_dir, vec3 cone_origin, cone_t cone, float* out_footprint) {
	const oracle_scene_t* scene = ctx->scene;
	shading_data_t s;
	float barys[3] = { (1.0f - barycentrics.x) - barycentrics.y, barycentrics.x, barycentrics.y };
	vec3 poss[3];
	s.pos = v3(0.0f, 0.0f, 0.0f);
	vec3 normal_geo = v3(0.0f, 0.0f, 0.0f);
	vec2 tex_coords[3];
	vec2 tex_coord = { 0.0f, 0.0f };
	for (int i = 0; i != 3; ++i) {
		uint64_t vert_index = (uint64_t) triangle_index * 3 + i;
		const uint32_t* quantized_pos = &scene->positions[2 * vert_index];
		const uint16_t* nuv = &scene->normals_uvs[4 * vert_index];
		float unorm[4] = { (float) nuv[0] / 65535.0f, (float) nuv[1] / 65535.0f, (float) nuv[2] / 65535.0f, (float) nuv[3] / 65535.0f };
		poss[i] = dequantize_position(quantized_pos, ctx->c->dequantization_factor, ctx->c->dequantization_summand);
		s.pos = add3(s.pos, scale3(poss[i], barys[i]));
		normal_geo = add3(normal_geo, scale3(dequantize_normal(unorm[0], unorm[1]), barys[i]));
		tex_coords[i].x = unorm[2] * 8.0f + 0.0f;
		tex_coords[i].y = unorm[3] * -8.0f + 1.0f;
		tex_coord.x += barys[i] * tex_coords[i].x;
		tex_coord.y += barys[i] * tex_coords[i].y;
	}
	normal_geo = normalize3(normal_geo);
	uint32_t material_index = scene->materials[triangle_index];
	/* Level of detail of the three lookups */
	vec3 travelled = sub3(s.pos, cone_origin);
	float footprint = fmaf(cone.spread, sqrtf(dot3(travelled, travelled)), cone.width);
	if (out_footprint)
		*out_footprint = footprint;
	vec3 face_normal = cross3(sub3(poss[1], poss[0]), sub3(poss[2], poss[0]));
	float uv_area_2 = fabsf((tex_coords[1].x - tex_coords[0].x) * (tex_coords[2].y - tex_coords[0].y) - (tex_coords[2].x - tex_coords[0].x) * (tex_coords[1].y - tex_coords[0].y));
	float lod_base = texture_lod_base(uv_area_2, sqrtf(dot3(face_normal, face_normal)), footprint, fabsf(dot3(normal_geo, out_dir)));
	vec3 base_color_tex = sample_texture(scene, 3 * material_index + 0, tex_coord, lod_base);
	vec3 specular_tex = sample_texture(scene, 3 * material_index + 1, tex_coord, lod_base);
	vec3 normal_tex = sample_texture(scene, 3 * material_index + 2, tex_coord, lod_base);
	vec3 normal_local;
	normal_local.x = normal_tex.x * 2.0f - 1.0f;
	normal_local.y = normal_tex.y * 2.0f - 1.0f;
	normal_local.z = sqrtf(fmaxf(0.0f, (1.0f - normal_local.x * normal_local.x) - normal_local.y * normal_local.y));
	/* mat2(e1, e2): [0] = e1, [1] = e2 (columns) */
	vec2 e1 = { tex_coords[1].x - tex_coords[0].x, tex_coords[1].y - tex_coords[0].y };
	vec2 e2 = { tex_coords[2].x - tex_coords[0].x, tex_coords[2].y - tex_coords[0].y };
	vec3 pre_tangent_0 = cross3(normal_geo, sub3(poss[1], poss[0]));
	vec3 pre_tangent_1 = cross3(normal_geo, sub3(poss[0], poss[2]));
	vec3 tangent_0 = add3(scale3(pre_tangent_1, e1.x), scale3(pre_tangent_0, e2.x));
	vec3 tangent_1 = add3(scale3(pre_tangent_1, e1.y), scale3(pre_tangent_0, e2.y));
	float mean_length = sqrtf(0.5f * (dot3(tangent_0, tangent_0) + dot3(tangent_1, tangent_1)));
	normal_local.z *= fmaxf(1.0e-8f, mean_length);
	s.normal = normalize3(add3(add3(scale3(tangent_0, normal_local.x), scale3(tangent_1, normal_local.y)), scale3(normal_geo, normal_local.z)));
	s.normal = front ? s.normal : neg3(s.normal);
	s.out_dir = out_dir;
	float normal_offset = fmaxf(0.0f, 1.0e-3f - dot3(s.normal, s.out_dir));
	s.normal = normalize3(v3(fmaf(normal_offset, s.out_dir.x, s.normal.x), fmaf(normal_offset, s.out_dir.y, s.normal.y), fmaf(normal_offset, s.out_dir.z, s.normal.z)));
	s.lambert_out = dot3(s.normal, s.out_dir);
	float metalicity = specular_tex.z;
	s.diffuse_albedo = sub3(base_color_tex, scale3(base_color_tex, metalicity));
	/* mix(vec3(0.02), base, metal) = 0.02 * (1 - metal) + base * metal */
	s.fresnel_0 = add3(scale3(v3(0.02f, 0.02f, 0.02f), 1.0f - metalicity), scale3(base_color_tex, metalicity));
	s.roughness = fmaxf(0.006f, specular_tex.y * specular_tex.y);
	if (material_index == ctx->emission_material_index)
		s.emission = v3(ctx->c->emission_material_radiance[0], ctx->c->emission_material_radiance[1], ctx->c->emission_material_radiance[2]);
	else
		s.emission = v3(0.0f, 0.0f, 0.0f);
	return s;
}


/* ---- brdfs.glsl ---- */

/* brdfs.glsl:7-12 (scalar form; all uses have f_90 or f_0 uniform per channel) */
static inline float fresnel_schlick(float f_0, float f_90, float lambert) {
	float flip_1 = 1.0f - lambert;
	float flip_2 = flip_1 * flip_1;
	float flip_5 = flip_2 * flip_1 * flip_2;
	return flip_5 * (f_90 - f_0) + f_0;
}

/* brdfs.glsl:19-46 */
static vec3 frostbite_brdf(const shading_data_t* s, vec3 in_dir) {
	float lambert_in = dot3(s->normal, in_dir);
	if (fminf(lambert_in, s->lambert_out) < 0.0f)
		return v3(0.0f, 0.0f, 0.0f);
	vec3 half_dir = normalize3(add3(in_dir, s->out_dir));
	float half_dot_out = dot3(half_dir, s->out_dir);
	float f_90 = (half_dot_out * half_dot_out) * (2.0f * s->roughness) + 0.5f;
	float fresnel_diffuse = fresnel_schlick(1.0f, f_90, s->lambert_out) * fresnel_schlick(1.0f, f_90, lambert_in);
	vec3 brdf = scale3(s->diffuse_albedo, fresnel_diffuse);
	float half_dot_normal = dot3(half_dir, s->normal);
	float roughness_2 = s->roughness * s->roughness;
	float ggx = (roughness_2 * half_dot_normal - half_dot_normal) * half_dot_normal + 1.0f;
	ggx = roughness_2 / (ggx * ggx);
	float masking = lambert_in * sqrtf((s->lambert_out - roughness_2 * s->lambert_out) * s->lambert_out + roughness_2);
	float shadowing = s->lambert_out * sqrtf((lambert_in - roughness_2 * lambert_in) * lambert_in + roughness_2);
	float smith = 0.5f / (masking + shadowing);
	float l = fmaxf(0.0f, half_dot_out);
	vec3 fresnel = v3(fresnel_schlick(s->fresnel_0.x, 1.0f, l), fresnel_schlick(s->fresnel_0.y, 1.0f, l), fresnel_schlick(s->fresnel_0.z, 1.0f, l));
	float gs = ggx * smith;
	brdf = add3(brdf, scale3(fresnel, gs));
	return scale3(brdf, 1.0f / M_PI_F);
}

/* brdfs.glsl:58-75, isotropic roughness */
static vec3 sample_ggx_vndf(vec3 out_dir, float roughness, vec2 randoms) {
	vec3 out_dir_std = normalize3(v3(out_dir.x * roughness, out_dir.y * roughness, out_dir.z));
	float azimuth = (2.0f * M_PI_F) * randoms.x - M_PI_F;
	float z = 1.0f - randoms.y * (1.0f + out_dir_std.z);
	float sine = sqrtf(fmaxf(0.0f, 1.0f - z * z));
	float sa, ca;
	o_sincos(azimuth, &sa, &ca);
	vec3 cap = v3(sine * ca, sine * sa, z);
	vec3 half_dir_std = add3(cap, out_dir_std);
	return normalize3(v3(half_dir_std.x * roughness, half_dir_std.y * roughness, half_dir_std.z));
}

/* brdfs.glsl:88-99 */
static float get_ggx_vndf_density(float lambert_out, float half_dot_normal, float half_dot_out, float roughness) {
	if (half_dot_normal < 0.0f)
		return 0.0f;
	float roughness_2 = roughness * roughness;
	float flip_roughness_2 = 1.0f - roughness * roughness;
	float length_M_inv_out_2 = roughness_2 + flip_roughness_2 * lambert_out * lambert_out;
	float D_vis_std = fmaxf(0.0f, half_dot_out) * (2.0f / M_PI_F) / (lambert_out + sqrtf(length_M_inv_out_2));
	float length_M_half_2 = 1.0f - flip_roughness_2 * half_dot_normal * half_dot_normal;
	return D_vis_std * roughness_2 / (length_M_half_2 * length_M_half_2);
}

/* brdfs.glsl:104-107; -reflect(o, h) = 2 dot(h, o) h - o, evaluated as
   -(o - 2 dot(h, o) h) like GLSL's reflect() */
static vec3 sample_ggx_in_dir(vec3 out_dir, float roughness, vec2 randoms) {
	vec3 half_dir = sample_ggx_vndf(out_dir, roughness, randoms);
	float d = 2.0f * dot3(half_dir, out_dir);
	return neg3(sub3(out_dir, scale3(half_dir, d)));
}

/* brdfs.glsl:112-121 */
static float get_ggx_in_dir_density(float lambert_out, vec3 out_dir, vec3 in_dir, vec3 normal, float roughness) {
	vec3 half_dir = normalize3(add3(in_dir, out_dir));
	float half_dot_out = dot3(half_dir, out_dir);
	float half_dot_normal = dot3(half_dir, normal);
	float density = get_ggx_vndf_density(lambert_out, half_dot_normal, half_dot_out, roughness);
	density /= 4.0f * half_dot_out;
	return density;
}

typedef struct { vec3 b1, b2, n; } mat3_t;

/* brdfs.glsl:126-134 */
static mat3_t get_shading_space(vec3 n) {
	float s = (n.z > 0.0f) ? 1.0f : -1.0f;
	float a = -1.0f / (s + n.z);
	float b = n.x * n.y * a;
	mat3_t m;
	m.b1 = v3(1.0f + s * n.x * n.x * a, s * b, -s * n.x);
	m.b2 = v3(b, s + n.y * n.y * a, -n.y);
	m.n = n;
	return m;
}

static inline vec3 mat3_mul(const mat3_t* m, vec3 v) {
	return add3(add3(scale3(m->b1, v.x), scale3(m->b2, v.y)), scale3(m->n, v.z));
}

static inline vec3 mat3_transpose_mul(const mat3_t* m, vec3 v) {
	return v3(dot3(m->b1, v), dot3(m->b2, v), dot3(m->n, v));
}

/* brdfs.glsl:139-146 */
static vec3 sample_hemisphere_psa(vec2 randoms) {
	float azimuth = (2.0f * M_PI_F) * randoms.x - M_PI_F;
	float radius = sqrtf(randoms.y);
	float z = sqrtf(1.0f - radius * radius);
	float sa, ca;
	o_sincos(azimuth, &sa, &ca);
	return v3(radius * ca, radius * sa, z);
}

/* brdfs.glsl:151-153 */
static inline float get_hemisphere_psa_density(float sampled_dir_z) {
	return (1.0f / M_PI_F) * fmaxf(0.0f, sampled_dir_z);
}

/* brdfs.glsl:158-164 */
static inline float get_diffuse_sampling_probability(const shading_data_t* s) {
	return fminf(0.5f, dot3(s->diffuse_albedo, v3(0.2126f, 0.7152f, 0.0722f)));
}

/* brdfs.glsl:174-199 */
static vec3 sample_frostbite_brdf(const shading_data_t* s, vec2 randoms) {
	mat3_t shading_to_world_space = get_shading_space(s->normal);
	float diffuse_prob = get_diffuse_sampling_probability(s);
	int diffuse = randoms.x < diffuse_prob;
	if (diffuse) {
		randoms.x /= diffuse_prob;
		return mat3_mul(&shading_to_world_space, sample_hemisphere_psa(randoms));
	}
	else {
		randoms.x = (randoms.x - diffuse_prob) / (1.0f - diffuse_prob);
		vec3 local_out_dir = mat3_transpose_mul(&shading_to_world_space, s->out_dir);
		vec3 local_in_dir = sample_ggx_in_dir(local_out_dir, s->roughness, randoms);
		return mat3_mul(&shading_to_world_space, local_in_dir);
	}
}

/* brdfs.glsl:203-208; mix(spec, diff, p) = spec * (1 - p) + diff * p */
static float get_frostbite_brdf_density(const shading_data_t* s, vec3 sampled_dir) {
	float diffuse_prob = get_diffuse_sampling_probability(s);
	float specular_density = get_ggx_in_dir_density(s->lambert_out, s->out_dir, sampled_dir, s->normal, s->roughness);
	float diffuse_density = get_hemisphere_psa_density(dot3(s->normal, sampled_dir));
	return specular_density * (1.0f - diffuse_prob) + diffuse_density * diffuse_prob;
}


/* ---- pathtrace.frag.glsl ---- */

/* :50-55 */
static vec3 sample_hemisphere_spherical(vec2 randoms) {
	float azimuth = (2.0f * M_PI_F) * randoms.x - M_PI_F;
	float inclination = (0.5f * M_PI_F) * randoms.y;
	float si, ci, sa, ca;
	o_sincos(inclination, &si, &ci);
	o_sincos(azimuth, &sa, &ca);
	float radius = si;
	return v3(radius * ca, radius * sa, ci);
}

/* :60-64 */
static float get_hemisphere_spherical_density(float sampled_dir_z) {
	if (sampled_dir_z < 0.0f)
		return 0.0f;
	return 1.0f / ((M_PI_F * M_PI_F) * sqrtf(fmaxf(0.0f, 1.0f - sampled_dir_z * sampled_dir_z)));
}

/* :69-83 */
static float get_spherical_light_importance(vec3 center, float radius, vec3 shading_pos, vec3 normal) {
	vec3 center_dir = sub3(center, shading_pos);
	if (dot3(normal, center_dir) < -radius)
		return 0.0f;
	float center_dist_2 = dot3(center_dir, center_dir);
	float sin_2 = radius * radius / center_dist_2;
	float z_range = sin_2 / (1.0f + sqrtf(fmaxf(0.0f, 1.0f - sin_2)));
	return z_range;
}

/* :95-106 */
static vec3 sample_spherical_light(vec3 center, float importance, vec3 shading_pos, vec2 randoms) {
	float azimuth = (2.0f * M_PI_F) * randoms.x - M_PI_F;
	float z_range = importance;
	float z = 1.0f - z_range * randoms.y;
	float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
	float sa, ca;
	o_sincos(azimuth, &sa, &ca);
	vec3 local_dir = v3(r * ca, r * sa, z);
	mat3_t light_to_world_space = get_shading_space(normalize3(sub3(center, shading_pos)));
	return mat3_mul(&light_to_world_space, local_dir);
}

/* :120-143 */
static vec3 sample_lights(const context_t* ctx, float* out_total_importance, vec3 shading_pos, vec3 normal, vec2 randoms) {
	float total = 0.0f;
	for (uint32_t i = 0; i != ctx->light_count; ++i) {
		const float* l = &ctx->lights[4 * i];
		total += get_spherical_light_importance(v3(l[0], l[1], l[2]), l[3], shading_pos, normal);
	}
	*out_total_importance = total;
	float target_importance = randoms.x * total;
	float prefix_importance = 0.0f;
	for (uint32_t i = 0; i != ctx->light_count; ++i) {
		const float* l = &ctx->lights[4 * i];
		float importance = get_spherical_light_importance(v3(l[0], l[1], l[2]), l[3], shading_pos, normal);
		prefix_importance += importance;
		if (prefix_importance > target_importance) {
			randoms.x = ((target_importance + importance) - prefix_importance) / importance;
			return sample_spherical_light(v3(l[0], l[1], l[2]), importance, shading_pos, randoms);
		}
	}
	return v3(0.0f, 0.0f, 0.0f);
}

/* :150-175 */
static float get_lights_density(const context_t* ctx, float total_importance, vec3 shading_pos, vec3 normal, vec3 sampled_dir, int is_light_dir) {
	(void) normal;
	if (total_importance <= 0.0f)
		return 0.0f;
	float light_count = 0.0f;
	for (uint32_t i = 0; i != ctx->light_count; ++i) {
		const float* l = &ctx->lights[4 * i];
		vec3 center_dir = sub3(v3(l[0], l[1], l[2]), shading_pos);
		float center_dist_2 = dot3(center_dir, center_dir);
		float center_dot_dir = dot3(center_dir, sampled_dir);
		float radius_2 = l[3] * l[3];
		float in_sphere = center_dist_2 - radius_2;
		float discriminant = center_dot_dir * center_dot_dir - in_sphere;
		light_count += (discriminant >= 0.0f && in_sphere >= 0.0f && center_dot_dir >= 0.0f) ? 1.0f : 0.0f;
	}
	if (is_light_dir)
		light_count = fmaxf(1.0f, light_count);
	return light_count / (2.0f * M_PI_F * total_importance);
}

/* ---- Light tree: no counterpart in the reference (it loops over at most 32
   lights). Restated from the definition the HIP library documents in
   path_tracer_amd/csrc/pt_light_tree.h, with code of its own: a complete
   binary tree in heap order whose leaves hold one light or none; a node holds
   centre and half extents of the box around its spheres, the sum of their
   squared radii and the squared half diagonal (8 floats). */
typedef struct {
	uint32_t coordinate; /* bit pattern of the float, mapped to sort like the number */
	uint32_t index;
} light_key_t;

static int compare_light_keys(const void* a, const void* b) {
	const light_key_t* ka = (const light_key_t*) a;
	const light_key_t* kb = (const light_key_t*) b;
	if (ka->coordinate != kb->coordinate)
		return (ka->coordinate < kb->coordinate) ? -1 : 1;
	return (ka->index < kb->index) ? -1 : ((ka->index > kb->index) ? 1 : 0);
}

/* Distributes the lights indices[0 .. count) over the slot_count leaf slots
   of a subtree: split at the median of the centres along the longest axis of
   their bounding box (lowest axis on ties, lower index on equal coordinates),
   the larger half to the left */
static void place_lights(const float* lights, uint32_t* indices, size_t count, float* leaf_lights, size_t slot_count) {
	if (count == 0)
		return;
	if (slot_count == 1) {
		memcpy(leaf_lights, &lights[4 * (size_t) indices[0]], 4 * sizeof(float));
		return;
	}
	int axis = 0;
	float longest = -1.0f;
	for (int a = 0; a != 3; ++a) {
		float low = lights[4 * (size_t) indices[0] + a], high = low;
		for (size_t i = 1; i != count; ++i) {
			float c = lights[4 * (size_t) indices[i] + a];
			if (c < low) low = c;
			if (c > high) high = c;
		}
		if (high - low > longest) {
			longest = high - low;
			axis = a;
		}
	}
	light_key_t* keys = (light_key_t*) malloc(sizeof(light_key_t) * count);
	for (size_t i = 0; i != count; ++i) {
		uint32_t bits = float_to_bits(lights[4 * (size_t) indices[i] + axis]);
		keys[i].coordinate = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
		keys[i].index = indices[i];
	}
	qsort(keys, count, sizeof(light_key_t), compare_light_keys);
	for (size_t i = 0; i != count; ++i)
		indices[i] = keys[i].index;
	free(keys);
	size_t left_count = (count + 1) / 2;
	place_lights(lights, indices, left_count, leaf_lights, slot_count / 2);
	place_lights(lights, indices + left_count, count - left_count, leaf_lights + 4 * (slot_count / 2), slot_count / 2);
}

static void build_light_tree(const float* lights, uint32_t count, float** out_nodes, float** out_tree_lights, uint32_t* out_leaf_count) {
	uint32_t leaf_count = 1;
	while (leaf_count < count)
		leaf_count <<= 1;
	float* nodes = (float*) calloc(16 * (size_t) leaf_count, sizeof(float));
	float* tree_lights = (float*) calloc(4 * (size_t) leaf_count, sizeof(float));
	float* lo = (float*) calloc(6 * (size_t) leaf_count, sizeof(float));
	float* hi = (float*) calloc(6 * (size_t) leaf_count, sizeof(float));
	uint32_t* indices = (uint32_t*) malloc(sizeof(uint32_t) * (count ? count : 1));
	for (uint32_t i = 0; i != count; ++i)
		indices[i] = i;
	place_lights(lights, indices, count, tree_lights, leaf_count);
	free(indices);
	for (size_t j = 0; j != leaf_count; ++j) {
		const float* l = &tree_lights[4 * j];
		size_t node = (size_t) leaf_count + j;
		float radius = fabsf(l[3]);
		for (int a = 0; a != 3; ++a) {
			lo[3 * node + a] = l[a] - radius;
			hi[3 * node + a] = l[a] + radius;
		}
		nodes[8 * node + 3] = l[3] * l[3];
	}
	for (size_t node = leaf_count - 1; node >= 1; --node) {
		float power_0 = nodes[8 * (2 * node) + 3], power_1 = nodes[8 * (2 * node + 1) + 3];
		/* Children without power (empty leaves, radius zero) do not widen the box */
		float power = 0.0f;
		int first = 1;
		for (size_t child = 2 * node; child != 2 * node + 2; ++child) {
			float child_power = (child == 2 * node) ? power_0 : power_1;
			if (!(child_power > 0.0f))
				continue;
			for (int a = 0; a != 3; ++a) {
				lo[3 * node + a] = first ? lo[3 * child + a] : fminf(lo[3 * node + a], lo[3 * child + a]);
				hi[3 * node + a] = first ? hi[3 * child + a] : fmaxf(hi[3 * node + a], hi[3 * child + a]);
			}
			power = first ? child_power : (power + child_power);
			first = 0;
		}
		nodes[8 * node + 3] = power;
	}
	for (size_t node = 1; node != 2 * (size_t) leaf_count; ++node) {
		float half[3];
		for (int a = 0; a != 3; ++a) {
			nodes[8 * node + a] = 0.5f * (lo[3 * node + a] + hi[3 * node + a]);
			half[a] = 0.5f * (hi[3 * node + a] - lo[3 * node + a]);
			nodes[8 * node + 4 + a] = half[a];
		}
		nodes[8 * node + 7] = (half[0] * half[0] + half[1] * half[1]) + half[2] * half[2];
	}
	free(lo); free(hi);
	*out_nodes = nodes;
	*out_tree_lights = tree_lights;
	*out_leaf_count = leaf_count;
}

/* Selection weight of a node seen from a shading point as a fraction: power
   over squared distance (at least the squared half diagonal); the power is
   zero for an empty node and below the horizon */
static void light_node_weight(const context_t* ctx, uint32_t node, vec3 shading_pos, vec3 normal, float* out_power, float* out_distance_2) {
	const float* n = &ctx->light_nodes[8 * (size_t) node];
	*out_power = 0.0f;
	*out_distance_2 = 1.0f;
	if (!(n[3] > 0.0f))
		return;
	vec3 center_dir = sub3(v3(n[0], n[1], n[2]), shading_pos);
	float reach = (fabsf(normal.x) * n[4] + fabsf(normal.y) * n[5]) + fabsf(normal.z) * n[6];
	if (dot3(normal, center_dir) + reach < 0.0f)
		return;
	*out_power = n[3];
	*out_distance_2 = fmaxf(dot3(center_dir, center_dir), n[7]);
}

/* The weights of the children 2i and 2i + 1 over a common denominator (only
   their proportion matters): p0 / d0 : p1 / d1 = p0 * d1 : p1 * d0 */
static void light_child_weights(const context_t* ctx, uint32_t left_child, vec3 shading_pos, vec3 normal, float* w0, float* w1) {
	float p0, d0, p1, d1;
	light_node_weight(ctx, left_child, shading_pos, normal, &p0, &d0);
	light_node_weight(ctx, left_child + 1, shading_pos, normal, &p1, &d1);
	*w0 = p0 * d1;
	*w1 = p1 * d0;
}

static vec3 sample_lights_tree(const context_t* ctx, float* out_density, vec3 shading_pos, vec3 normal, vec2 randoms) {
	*out_density = 0.0f;
	uint32_t node = 1;
	float probability = 1.0f;
	while (node < ctx->light_leaf_count) {
		float w0, w1;
		light_child_weights(ctx, 2 * node, shading_pos, normal, &w0, &w1);
		float total = w0 + w1;
		if (!(total > 0.0f))
			return v3(0.0f, 0.0f, 0.0f);
		float target = randoms.x * total;
		if (target < w0 || !(w1 > 0.0f)) {
			randoms.x = fminf(1.0f, target / w0);
			probability *= w0 / total;
			node = 2 * node;
		}
		else {
			randoms.x = fminf(1.0f, (target - w0) / w1);
			probability *= w1 / total;
			node = 2 * node + 1;
		}
	}
	const float* l = &ctx->tree_lights[4 * (size_t) (node - ctx->light_leaf_count)];
	/* A shading point inside the light's sphere gets no sample of that light:
	   the density below counts a light only for rays that start outside of it
	   (the condition of :167), and the formulas of :69-106 suppose a point
	   outside (inside, the "cone" is wider than a hemisphere and the sampled
	   vector, whose z goes below -1, is no direction any more). Emission seen
	   from inside is found by the BRDF sample, with a light density of zero. */
	vec3 center_dir = sub3(v3(l[0], l[1], l[2]), shading_pos);
	if (!(dot3(center_dir, center_dir) - l[3] * l[3] >= 0.0f))
		return v3(0.0f, 0.0f, 0.0f);
	float importance = get_spherical_light_importance(v3(l[0], l[1], l[2]), l[3], shading_pos, normal);
	if (!(importance > 0.0f))
		return v3(0.0f, 0.0f, 0.0f);
	*out_density = probability / ((2.0f * M_PI_F) * importance);
	return sample_spherical_light(v3(l[0], l[1], l[2]), importance, shading_pos, randoms);
}

/* Probability of the descent above ending at the given leaf node */
static float light_selection_probability(const context_t* ctx, uint32_t leaf_node, vec3 shading_pos, vec3 normal) {
	uint32_t depth = 0;
	while ((leaf_node >> depth) > 1)
		++depth;
	float probability = 1.0f;
	for (uint32_t level = 1; level <= depth; ++level) {
		uint32_t node = leaf_node >> (depth - level);
		float w0, w1;
		light_child_weights(ctx, node & ~1u, shading_pos, normal, &w0, &w1);
		float w = (node & 1u) ? w1 : w0;
		if (!(w > 0.0f))
			return 0.0f;
		probability *= w / (w0 + w1);
	}
	return probability;
}

/* Adds to *density, for every light below node that a ray from shading_pos
   along dir meets, selection probability times cone density; left subtree
   first. The test is the condition of :160-170 with the distance between the
   light's centre and the line taken from a cross product: the reference's
   discriminant, a difference of two squared distances, is off by about a
   thousandth of the distance to the light, more than the boxes of the tree
   allow for. */
static void accumulate_tree_density(const context_t* ctx, uint32_t node, vec3 shading_pos, vec3 normal, vec3 dir, float* density) {
	const float* n = &ctx->light_nodes[8 * (size_t) node];
	if (!(n[3] > 0.0f))
		return;
	if (node >= ctx->light_leaf_count) {
		const float* l = &ctx->tree_lights[4 * (size_t) (node - ctx->light_leaf_count)];
		vec3 center_dir = sub3(v3(l[0], l[1], l[2]), shading_pos);
		float in_sphere = dot3(center_dir, center_dir) - l[3] * l[3];
		float center_dot_dir = dot3(center_dir, dir);
		vec3 lever = cross3(center_dir, dir);
		float length_2 = dot3(dir, dir);
		if (dot3(lever, lever) <= (l[3] * l[3]) * length_2 && length_2 > 0.0f && in_sphere >= 0.0f && center_dot_dir >= 0.0f) {
			float importance = get_spherical_light_importance(v3(l[0], l[1], l[2]), l[3], shading_pos, normal);
			if (importance > 0.0f)
				*density += light_selection_probability(ctx, node, shading_pos, normal) / ((2.0f * M_PI_F) * importance);
		}
		return;
	}
	/* Skip the subtree unless the ray meets its box: slab test in the form
	   centre / d - origin / d -+ half / |d| per axis, the box widened by a
	   ten-thousandth and by a millionth of the coordinates involved (root box
	   and shading position) against rounding */
	const float* root = &ctx->light_nodes[8];
	float slack = 1.0e-6f * ((((fabsf(shading_pos.x) + fabsf(shading_pos.y)) + fabsf(shading_pos.z))
		+ ((fabsf(root[0]) + fabsf(root[1])) + fabsf(root[2]))) + ((root[4] + root[5]) + root[6]));
	float t_near = 0.0f, t_far = INFINITY;
	const float p[3] = { shading_pos.x, shading_pos.y, shading_pos.z }, d[3] = { dir.x, dir.y, dir.z };
	for (int a = 0; a != 3; ++a) {
		float safe = (fabsf(d[a]) > 1.0e-30f) ? d[a] : copysignf(1.0e-30f, d[a]);
		float inv = 1.0f / safe;
		float t_center = fmaf(n[a], inv, -(p[a] * inv));
		float t_half = fmaf(n[4 + a], 1.0001f, slack) * fabsf(inv);
		t_near = fmaxf(t_near, t_center - t_half);
		t_far = fminf(t_far, t_center + t_half);
	}
	if (t_near <= t_far) {
		accumulate_tree_density(ctx, 2 * node, shading_pos, normal, dir, density);
		accumulate_tree_density(ctx, 2 * node + 1, shading_pos, normal, dir, density);
	}
}

static float get_lights_density_tree(const context_t* ctx, vec3 shading_pos, vec3 normal, vec3 sampled_dir) {
	float density = 0.0f;
	accumulate_tree_density(ctx, 1, shading_pos, normal, sampled_dir, &density);
	return density;
}

/* :181-199 */
static int trace_ray(context_t* ctx, shading_data_t* out, vec3 ray_origin, vec3 ray_dir, cone_t cone, float* out_footprint) {
	++ctx->ray_count;
	hit_t hit = trace_closest(ctx->scene, ray_origin, ray_dir, 1.0e-3f, 1e38f);
	if (hit.triangle < 0) {
		memset(out, 0, sizeof(*out));
		out->emission = v3(ctx->c->sky_radiance[0], ctx->c->sky_radiance[1], ctx->c->sky_radiance[2]);
		return 0;
	}
	vec2 barys = { hit.u, hit.v };
	*out = get_shading_data(ctx, hit.triangle, barys, hit.front, neg3(ray_dir), ray_origin, cone, out_footprint);
	return 1;
}

/* :203-219 */
static vec3 trace_ray_emission(context_t* ctx, vec3 ray_origin, vec3 ray_dir) {
	++ctx->ray_count;
	hit_t hit = trace_closest(ctx->scene, ray_origin, ray_dir, 1.0e-3f, 1e38f);
	if (hit.triangle < 0)
		return v3(ctx->c->sky_radiance[0], ctx->c->sky_radiance[1], ctx->c->sky_radiance[2]);
	if (ctx->scene->materials[hit.triangle] == ctx->emission_material_index)
		return v3(ctx->c->emission_material_radiance[0], ctx->c->emission_material_radiance[1], ctx->c->emission_material_radiance[2]);
	return v3(0.0f, 0.0f, 0.0f);
}

/* :224-247 and :258-281; spherical != 0 selects path_trace_spherical */
static vec3 path_trace_uniform(context_t* ctx, vec3 ray_origin, vec3 ray_dir, cone_t cone, uint32_t seed[2], int spherical) {
	vec3 throughput_weight = v3(1.0f, 1.0f, 1.0f);
	vec3 radiance = v3(0.0f, 0.0f, 0.0f);
	for (uint32_t k = 1; k != ctx->path_length + 1; ++k) {
		shading_data_t s;
		float footprint = 0.0f;
		int hit = trace_ray(ctx, &s, ray_origin, ray_dir, cone, &footprint);
		radiance = add3(radiance, mul3(throughput_weight, s.emission));
		if (hit && k < ctx->path_length) {
			mat3_t shading_to_world_space = get_shading_space(s.normal);
			ray_origin = s.pos;
			vec2 randoms = get_random_numbers(seed);
			vec3 sampled_dir = spherical ? sample_hemisphere_spherical(randoms) : sample_hemisphere_psa(randoms);
			ray_dir = mat3_mul(&shading_to_world_space, sampled_dir);
			float lambert_in = sampled_dir.z;
			float density = spherical ? get_hemisphere_spherical_density(sampled_dir.z) : get_hemisphere_psa_density(sampled_dir.z);
			throughput_weight = mul3(throughput_weight, div3(scale3(frostbite_brdf(&s, ray_dir), lambert_in), density));
			cone = cone_after_bounce(footprint, cone.spread, density);
		}
		else
			break;
	}
	return radiance;
}

/* :286-311 */
static vec3 path_trace_brdf(context_t* ctx, vec3 ray_origin, vec3 ray_dir, cone_t cone, uint32_t seed[2]) {
	vec3 throughput_weight = v3(1.0f, 1.0f, 1.0f);
	vec3 radiance = v3(0.0f, 0.0f, 0.0f);
	for (uint32_t k = 1; k != ctx->path_length + 1; ++k) {
		shading_data_t s;
		float footprint = 0.0f;
		int hit = trace_ray(ctx, &s, ray_origin, ray_dir, cone, &footprint);
		radiance = add3(radiance, mul3(throughput_weight, s.emission));
		if (hit && k < ctx->path_length) {
			ray_origin = s.pos;
			ray_dir = sample_frostbite_brdf(&s, get_random_numbers(seed));
			float density = get_frostbite_brdf_density(&s, ray_dir);
			float lambert_in = dot3(s.normal, ray_dir);
			if (lambert_in <= 0.0f)
				break;
			throughput_weight = mul3(throughput_weight, div3(scale3(frostbite_brdf(&s, ray_dir), lambert_in), density));
			cone = cone_after_bounce(footprint, cone.spread, density);
		}
		else
			break;
	}
	return radiance;
}

/* :315-362 */
static vec3 path_trace_nee(context_t* ctx, vec3 ray_origin, vec3 ray_dir, cone_t cone, uint32_t seed[2]) {
	vec3 throughput_weight = v3(1.0f, 1.0f, 1.0f);
	vec3 nee_throughput_weight = throughput_weight;
	vec3 radiance = v3(0.0f, 0.0f, 0.0f);
	for (uint32_t k = 1; k != ctx->path_length + 1; ++k) {
		shading_data_t s;
		float footprint = 0.0f;
		int hit = trace_ray(ctx, &s, ray_origin, ray_dir, cone, &footprint);
		radiance = add3(radiance, mul3(nee_throughput_weight, s.emission));
		if (hit && k < ctx->path_length) {
			float total_light_importance = 0.0f, sampled_light_density = 0.0f;
			vec3 light_dir = ctx->light_sampling
				? sample_lights_tree(ctx, &sampled_light_density, s.pos, s.normal, get_random_numbers(seed))
				: sample_lights(ctx, &total_light_importance, s.pos, s.normal, get_random_numbers(seed));
			float lambert_in_0 = dot3(s.normal, light_dir);
			if (lambert_in_0 > 0.0f) {
				vec3 light_emission = trace_ray_emission(ctx, s.pos, light_dir);
				/* With the tree the sampled light counts even if rounding lets
				   its own ray miss it, like max(1, count) at :172-173 */
				float light_density_0 = ctx->light_sampling
					? fmaxf(get_lights_density_tree(ctx, s.pos, s.normal, light_dir), sampled_light_density)
					: get_lights_density(ctx, total_light_importance, s.pos, s.normal, light_dir, 1);
				float brdf_density_0 = get_frostbite_brdf_density(&s, light_dir);
				vec3 weight = mul3(mul3(throughput_weight, frostbite_brdf(&s, light_dir)), light_emission);
				radiance = add3(radiance, scale3(weight, lambert_in_0 / (light_density_0 + brdf_density_0)));
			}
			ray_origin = s.pos;
			ray_dir = sample_frostbite_brdf(&s, get_random_numbers(seed));
			float lambert_in_1 = dot3(s.normal, ray_dir);
			if (lambert_in_1 <= 0.0f)
				break;
			float light_density_1 = ctx->light_sampling
				? get_lights_density_tree(ctx, s.pos, s.normal, ray_dir)
				: get_lights_density(ctx, total_light_importance, s.pos, s.normal, ray_dir, 0);
			float brdf_density_1 = get_frostbite_brdf_density(&s, ray_dir);
			vec3 brdf_lambert_1 = scale3(frostbite_brdf(&s, ray_dir), lambert_in_1);
			nee_throughput_weight = mul3(throughput_weight, div3(brdf_lambert_1, light_density_1 + brdf_density_1));
			throughput_weight = mul3(throughput_weight, scale3(brdf_lambert_1, 1.0f / brdf_density_1));
			cone = cone_after_bounce(footprint, cone.spread, brdf_density_1);
		}
		else
			break;
	}
	return radiance;
}

/* camera_utilities.glsl:9-13; the UBO is row_major, so GLSL M * v is the
   plain row-major product */
static vec3 get_camera_ray_origin(vec2 ray_tex_coord, const float m[16]) {
	float px = 2.0f * ray_tex_coord.x - 1.0f, py = 2.0f * ray_tex_coord.y - 1.0f;
	float r[4];
	for (int i = 0; i != 4; ++i)
		r[i] = ((m[4 * i + 0] * px + m[4 * i + 1] * py) + m[4 * i + 2] * 0.0f) + m[4 * i + 3] * 1.0f;
	return v3(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
}

/* camera_utilities.glsl:24-44; GLSL m[c][r] is C m[4 * r + c] */
static vec3 get_camera_ray_direction(vec2 ray_tex_coord, const float m[16]) {
#define W(c, r) m[4 * (r) + (c)]
	float dx = 2.0f * ray_tex_coord.x - 1.0f, dy = 2.0f * ray_tex_coord.y - 1.0f;
	vec3 d;
	d.x = (W(1, 1) * W(2, 3) - W(1, 3) * W(2, 1)) * dx;
	d.x += (W(1, 3) * W(2, 0) - W(1, 0) * W(2, 3)) * dy;
	d.x += W(1, 0) * W(2, 1) - W(1, 1) * W(2, 0);
	d.y = (W(0, 3) * W(2, 1) - W(0, 1) * W(2, 3)) * dx;
	d.y += (W(0, 0) * W(2, 3) - W(0, 3) * W(2, 0)) * dy;
	d.y += W(0, 1) * W(2, 0) - W(0, 0) * W(2, 1);
	d.z = (W(0, 1) * W(1, 3) - W(0, 3) * W(1, 1)) * dx;
	d.z += (W(0, 3) * W(1, 0) - W(0, 0) * W(1, 3)) * dy;
	d.z += W(0, 0) * W(1, 1) - W(0, 1) * W(1, 0);
#undef W
	return normalize3(d);
}

/* The camera ray through a point of the image plane given in pixels, pathtrace.frag.glsl:374-386 */
static void get_camera_ray(const oracle_constants_t* c, vec2 jittered, vec3* ray_origin, vec3* ray_dir) {
	if (c->camera_type <= 1) {
		vec2 ray_tex_coord = { jittered.x * c->inv_viewport_size[0], jittered.y * c->inv_viewport_size[1] };
		*ray_origin = get_camera_ray_origin(ray_tex_coord, c->projection_to_world_space);
		*ray_dir = get_camera_ray_direction(ray_tex_coord, c->world_to_projection_space);
	}
	else {
		mat3_t hemisphere_to_world_space = get_shading_space(v3(c->hemispherical_camera_normal[0], c->hemispherical_camera_normal[1], c->hemispherical_camera_normal[2]));
		*ray_origin = v3(c->camera_pos[0], c->camera_pos[1], c->camera_pos[2]);
		float factor_y = (c->camera_type == 3) ? 2.0f : 1.0f;
		vec2 r = { jittered.x * 1.0f * c->inv_viewport_size[0], jittered.y * factor_y * c->inv_viewport_size[1] };
		*ray_dir = mat3_mul(&hemisphere_to_world_space, sample_hemisphere_spherical(r));
	}
}

static float length3(vec3 v) { return sqrtf(dot3(v, v)); }

/* pathtrace.frag.glsl:365-386: seed, jitter and primary ray of one pixel.
   out_cone (may be NULL) receives the ray's cone: how far apart the rays one
   pixel to the right and one pixel down start, and how far their directions
   are from this one's, the smaller of the two each. */
static void get_primary_ray(const oracle_constants_t* c, uint32_t x, uint32_t y, uint32_t frame_index, uint32_t seed[2], vec3* ray_origin, vec3* ray_dir, cone_t* out_cone) {
	seed[0] = x ^ (frame_index << 16);
	seed[1] = y ^ ((frame_index + 237u) << 16);
	const float std = 0.9f;
	vec2 randoms = get_random_numbers(seed);
	randoms.x = 2.0f * randoms.x - 1.0f;
	randoms.y = 2.0f * randoms.y - 1.0f;
	float scale = std * sqrtf(2.0f);
	vec2 jittered = { ((float) x + 0.5f) + scale * oracle_erfinv(randoms.x), ((float) y + 0.5f) + scale * oracle_erfinv(randoms.y) };
	get_camera_ray(c, jittered, ray_origin, ray_dir);
	if (out_cone) {
		vec2 right = { jittered.x + 1.0f, jittered.y }, down = { jittered.x, jittered.y + 1.0f };
		vec3 o_right, d_right, o_down, d_down;
		get_camera_ray(c, right, &o_right, &d_right);
		get_camera_ray(c, down, &o_down, &d_down);
		float width = fminf(length3(sub3(o_right, *ray_origin)), length3(sub3(o_down, *ray_origin)));
		float spread = fminf(length3(sub3(d_right, *ray_dir)), length3(sub3(d_down, *ray_dir)));
		*out_cone = make_cone(width, spread);
	}
}

/* pathtrace.frag.glsl:365-398 */
static vec3 render_pixel(context_t* ctx, uint32_t x, uint32_t y, uint32_t frame_index) {
	uint32_t seed[2];
	vec3 ray_origin, ray_dir;
	cone_t cone;
	get_primary_ray(ctx->c, x, y, frame_index, seed, &ray_origin, &ray_dir, &cone);
	switch (ctx->strategy) {
		case ORACLE_STRATEGY_SPHERICAL: return path_trace_uniform(ctx, ray_origin, ray_dir, cone, seed, 1);
		case ORACLE_STRATEGY_PSA: return path_trace_uniform(ctx, ray_origin, ray_dir, cone, seed, 0);
		case ORACLE_STRATEGY_BRDF: return path_trace_brdf(ctx, ray_origin, ray_dir, cone, seed);
		default: return path_trace_nee(ctx, ray_origin, ray_dir, cone, seed);
	}
}


/* ---- threaded driver ---- */

typedef struct {
	context_t ctx;
	uint32_t first_frame_index, frame_count, frame_stride;
	int accumulate;
	float* accum;
	uint32_t width, height;
	int thread_index, thread_count;
} render_job_t;

static void* render_rows(void* arg) {
	render_job_t* job = (render_job_t*) arg;
	for (uint32_t y = (uint32_t) job->thread_index; y < job->height; y += (uint32_t) job->thread_count) {
		for (uint32_t x = 0; x != job->width; ++x) {
			float* pixel = &job->accum[4 * ((uint64_t) y * job->width + x)];
			for (uint32_t f = 0; f != job->frame_count; ++f) {
				vec3 radiance = render_pixel(&job->ctx, x, y, job->first_frame_index + f * job->frame_stride);
				/* src/main.c:1000-1021, 1884-1888: the first frame replaces
				   the target, later ones blend ONE, ONE (alpha stays 1) */
				if (f == 0 && !job->accumulate) {
					pixel[0] = radiance.x; pixel[1] = radiance.y; pixel[2] = radiance.z; pixel[3] = 1.0f;
				}
				else {
					pixel[0] += radiance.x; pixel[1] += radiance.y; pixel[2] += radiance.z;
				}
			}
		}
	}
	return NULL;
}

uint64_t oracle_render(const oracle_scene_t* scene, const oracle_constants_t* constants,
	const float* lights, uint32_t light_count,
	int strategy, uint32_t path_length, uint32_t emission_material_index,
	uint32_t first_frame_index, uint32_t frame_count, uint32_t frame_stride,
	int accumulate, float* accum, int thread_count)
{
	return oracle_render_light_sampling(scene, constants, lights, light_count, 0, strategy, path_length, emission_material_index,
		first_frame_index, frame_count, frame_stride, accumulate, accum, thread_count);
}

uint64_t oracle_render_light_sampling(const oracle_scene_t* scene, const oracle_constants_t* constants,
	const float* lights, uint32_t light_count, int light_sampling,
	int strategy, uint32_t path_length, uint32_t emission_material_index,
	uint32_t first_frame_index, uint32_t frame_count, uint32_t frame_stride,
	int accumulate, float* accum, int thread_count)
{
	float* light_nodes = NULL;
	float* tree_lights = NULL;
	uint32_t light_leaf_count = 0;
	if (light_sampling)
		build_light_tree(lights, light_count, &light_nodes, &tree_lights, &light_leaf_count);
	if (thread_count < 1) thread_count = 1;
	if (thread_count > 256) thread_count = 256;
	render_job_t jobs[256];
	pthread_t threads[256];
	for (int i = 0; i != thread_count; ++i) {
		render_job_t* job = &jobs[i];
		memset(job, 0, sizeof(*job));
		job->ctx.scene = scene;
		job->ctx.c = constants;
		job->ctx.lights = lights;
		job->ctx.light_count = light_count;
		job->ctx.light_sampling = light_sampling;
		job->ctx.light_nodes = light_nodes;
		job->ctx.tree_lights = tree_lights;
		job->ctx.light_leaf_count = light_leaf_count;
		job->ctx.path_length = path_length;
		job->ctx.emission_material_index = emission_material_index;
		job->ctx.strategy = strategy;
		job->first_frame_index = first_frame_index;
		job->frame_count = frame_count;
		job->frame_stride = frame_stride;
		job->accumulate = accumulate;
		job->accum = accum;
		job->width = (uint32_t) constants->viewport_size[0];
		job->height = (uint32_t) constants->viewport_size[1];
		job->thread_index = i;
		job->thread_count = thread_count;
	}
	if (thread_count == 1)
		render_rows(&jobs[0]);
	else {
		for (int i = 0; i != thread_count; ++i)
			pthread_create(&threads[i], NULL, render_rows, &jobs[i]);
		for (int i = 0; i != thread_count; ++i)
			pthread_join(threads[i], NULL);
	}
	uint64_t rays = 0;
	for (int i = 0; i != thread_count; ++i)
		rays += jobs[i].ctx.ray_count;
	free(light_nodes);
	free(tree_lights);
	return rays;
}

void oracle_primary_hits(const oracle_scene_t* scene, const oracle_constants_t* constants,
	uint32_t frame_index, int32_t* triangle_indices, float* barycentrics, uint8_t* front)
{
	uint32_t width = (uint32_t) constants->viewport_size[0], height = (uint32_t) constants->viewport_size[1];
	for (uint32_t y = 0; y != height; ++y) {
		for (uint32_t x = 0; x != width; ++x) {
			uint32_t seed[2];
			vec3 o, d;
			get_primary_ray(constants, x, y, frame_index, seed, &o, &d, NULL);
			hit_t hit = trace_closest(scene, o, d, 1.0e-3f, 1e38f);
			uint64_t i = (uint64_t) y * width + x;
			triangle_indices[i] = hit.triangle;
			if (barycentrics) { barycentrics[2 * i] = hit.u; barycentrics[2 * i + 1] = hit.v; }
			if (front) front[i] = (uint8_t) hit.front;
		}
	}
}

void oracle_primary_cones(const oracle_constants_t* constants, uint32_t frame_index, float* widths, float* spreads) {
	uint32_t width = (uint32_t) constants->viewport_size[0], height = (uint32_t) constants->viewport_size[1];
	for (uint32_t y = 0; y != height; ++y) {
		for (uint32_t x = 0; x != width; ++x) {
			uint32_t seed[2];
			vec3 o, d;
			cone_t cone;
			get_primary_ray(constants, x, y, frame_index, seed, &o, &d, &cone);
			widths[(uint64_t) y * width + x] = cone.width;
			spreads[(uint64_t) y * width + x] = cone.spread;
		}
	}
}

void oracle_rng_stream(uint32_t x, uint32_t y, uint32_t frame_index, uint32_t draw_count, float* out_randoms, uint32_t* out_seeds) {
	uint32_t seed[2] = { x ^ (frame_index << 16), y ^ ((frame_index + 237u) << 16) };
	for (uint32_t i = 0; i != draw_count; ++i) {
		vec2 r = get_random_numbers(seed);
		out_randoms[2 * i] = r.x;
		out_randoms[2 * i + 1] = r.y;
		if (out_seeds) { out_seeds[2 * i] = seed[0]; out_seeds[2 * i + 1] = seed[1]; }
	}
}

void oracle_trace_rays(const oracle_scene_t* scene, const float* origins, const float* directions, uint32_t ray_count,
	int32_t* triangle_indices, float* t_u_v, uint8_t* front)
{
	for (uint32_t i = 0; i != ray_count; ++i) {
		hit_t hit = trace_closest(scene, v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), v3(directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]), 1.0e-3f, 1e38f);
		triangle_indices[i] = hit.triangle;
		if (t_u_v) { t_u_v[3 * i] = hit.t; t_u_v[3 * i + 1] = hit.u; t_u_v[3 * i + 2] = hit.v; }
		if (front) front[i] = (uint8_t) hit.front;
	}
}

void oracle_shading_data(const oracle_scene_t* scene, const oracle_constants_t* constants, uint32_t emission_material_index,
	const int32_t* triangle_indices, const float* barycentrics, const uint8_t* front, const float* out_dirs, const float* footprints, uint32_t count, float* out)
{
	context_t ctx;
	memset(&ctx, 0, sizeof(ctx));
	ctx.scene = scene;
	ctx.c = constants;
	ctx.emission_material_index = emission_material_index;
	for (uint32_t i = 0; i != count; ++i) {
		vec2 barys = { barycentrics[2 * i], barycentrics[2 * i + 1] };
		/* A cone that does not widen: the footprint is the given one wherever the hit lies */
		cone_t cone = { footprints ? footprints[i] : 0.0f, 0.0f };
		shading_data_t s = get_shading_data(&ctx, triangle_indices[i], barys, front[i], v3(out_dirs[3 * i], out_dirs[3 * i + 1], out_dirs[3 * i + 2]), v3(0.0f, 0.0f, 0.0f), cone, NULL);
		float* o = &out[20 * i];
		o[0] = s.pos.x; o[1] = s.pos.y; o[2] = s.pos.z;
		o[3] = s.normal.x; o[4] = s.normal.y; o[5] = s.normal.z;
		o[6] = s.out_dir.x; o[7] = s.out_dir.y; o[8] = s.out_dir.z;
		o[9] = s.lambert_out;
		o[10] = s.emission.x; o[11] = s.emission.y; o[12] = s.emission.z;
		o[13] = s.diffuse_albedo.x; o[14] = s.diffuse_albedo.y; o[15] = s.diffuse_albedo.z;
		o[16] = s.fresnel_0.x; o[17] = s.fresnel_0.y; o[18] = s.fresnel_0.z;
		o[19] = s.roughness;
	}
}

/* One sampler, density or BRDF for explicit arguments (see pt_oracle.h) */
void oracle_evaluate_function(int function, const float* shading_data, const float* lights, uint32_t light_count, const float* inputs, uint32_t count, float* outputs) {
	shading_data_t s;
	s.pos = v3(shading_data[0], shading_data[1], shading_data[2]);
	s.normal = v3(shading_data[3], shading_data[4], shading_data[5]);
	s.out_dir = v3(shading_data[6], shading_data[7], shading_data[8]);
	s.lambert_out = shading_data[9];
	s.emission = v3(shading_data[10], shading_data[11], shading_data[12]);
	s.diffuse_albedo = v3(shading_data[13], shading_data[14], shading_data[15]);
	s.fresnel_0 = v3(shading_data[16], shading_data[17], shading_data[18]);
	s.roughness = shading_data[19];
	context_t ctx;
	memset(&ctx, 0, sizeof(ctx));
	ctx.lights = lights;
	ctx.light_count = light_count;
	mat3_t shading_to_world_space = get_shading_space(s.normal);
	for (uint32_t i = 0; i != count; ++i) {
		const float* in = &inputs[4 * (size_t) i];
		float* out = &outputs[4 * (size_t) i];
		vec2 randoms = { in[0], in[1] };
		vec3 dir = v3(in[0], in[1], in[2]);
		vec3 result = v3(0.0f, 0.0f, 0.0f);
		float last = 0.0f;
		switch (function) {
		case 0: result = mat3_mul(&shading_to_world_space, sample_ggx_in_dir(mat3_transpose_mul(&shading_to_world_space, s.out_dir), s.roughness, randoms)); break;
		case 1: result.x = get_ggx_in_dir_density(s.lambert_out, s.out_dir, dir, s.normal, s.roughness); break;
		case 2: result = mat3_mul(&shading_to_world_space, sample_hemisphere_psa(randoms)); break;
		case 3: result.x = get_hemisphere_psa_density(dot3(s.normal, dir)); break;
		case 4: result = mat3_mul(&shading_to_world_space, sample_hemisphere_spherical(randoms)); break;
		case 5: result.x = get_hemisphere_spherical_density(dot3(s.normal, dir)); break;
		case 6: result = sample_frostbite_brdf(&s, randoms); break;
		case 7: result.x = get_frostbite_brdf_density(&s, dir); break;
		case 8: result = frostbite_brdf(&s, dir); break;
		case 9: result = sample_lights(&ctx, &last, s.pos, s.normal, randoms); break;
		case 10: {
			float total = 0.0f;
			for (uint32_t j = 0; j != light_count; ++j)
				total += get_spherical_light_importance(v3(lights[4 * j], lights[4 * j + 1], lights[4 * j + 2]), lights[4 * j + 3], s.pos, s.normal);
			result.x = get_lights_density(&ctx, total, s.pos, s.normal, dir, 1);
			result.y = get_lights_density(&ctx, total, s.pos, s.normal, dir, 0);
			result.z = total;
			break;
		}
		default: break;
		}
		out[0] = result.x;
		out[1] = result.y;
		out[2] = result.z;
		out[3] = last;
	}
}

void oracle_sample_texture(const oracle_scene_t* scene, uint32_t texture_index, const float* uvs, const float* lods, uint32_t count, float* out) {
	for (uint32_t i = 0; i != count; ++i) {
		vec2 uv = { uvs[2 * i], uvs[2 * i + 1] };
		vec3 r = sample_texture_lod(scene, texture_index, uv, lods ? lods[i] : 0.0f);
		out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
	}
}
