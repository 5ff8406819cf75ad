// Persistent closest-hit traversal for gfx950 (device code only).
//
// One ray per lane walks the 8-wide BVH exactly like pt_trace_closest() in
// pt_trace.h (a box test that is conservative like the one there, the same
// triangle arithmetic, the same tie break, hence
// bit-identical hits), organised for 64-lane waves:
//   * Waves are persistent. A lane whose ray is complete does not wait for
//     the slowest ray of its wave: when enough lanes are idle the wave
//     refills them from the ray queue in HBM. A wave reserves rays in bulk
//     with one atomic and then hands them out lane by lane from registers.
//     Each XCD works through its own contiguous eighth of the queue first, so
//     its L2 sees rays that started as neighbouring pixels, and steals from
//     the other eighths afterwards.
//   * Every pass of the loop does at most one node step and one triangle
//     test per lane. A leaf with many triangles therefore costs its lane
//     more passes instead of stalling the other 63 lanes in an inner loop.
//   * The traversal stack holds one entry per BVH level (a group of sibling
//     nodes): PT_WAVE_STACK entries per lane in LDS, laid out entry-major so
//     that a wave's accesses are conflict-free 8-byte words, the rest in HBM.
//   * Two 1024-thread workgroups share a CU: 32 waves, eight per SIMD, is what
//     hides the latency of the gathers (58 VGPRs allow it). Each has half of
//     the 160 KB of LDS. What its stacks leave free (32 KB) holds a prefix of
//     the BVH: the builder emits nodes, and the triangles of their leaves, in
//     breadth-first order, so the prefix is the top of the tree - or all of
//     it for a small scene. Incoherent rays make every lane of a load
//     instruction touch its own cache line, which the vector L1 serves at a
//     fraction of its bandwidth; the same gather from LDS is several times
//     faster and ten times closer.
#pragma once
#include "pt_common.h"
#include "pt_trace.h"

#ifndef PT_WAVES_PER_BLOCK
#define PT_WAVES_PER_BLOCK 16
#endif
#define PT_TRACE_BLOCK (64 * PT_WAVES_PER_BLOCK) // threads per workgroup of the traversal kernels
#define PT_VISIT_COUNTERS 32 // 64-bit words of visit counters and diagnostics
#ifndef PT_TRACE_BLOCKS_PER_CU
#define PT_TRACE_BLOCKS_PER_CU 2 // workgroups that share a CU and its LDS: 32 waves, the most a CU holds
#endif
#ifndef PT_WAVE_STACK
#define PT_WAVE_STACK 6      // stack entries per lane kept in LDS
#endif
// 16-byte words of BVH prefix kept in LDS: what the stacks leave of a workgroup's share of 160 KB
#define PT_CACHE_WORDS ((163840 / PT_TRACE_BLOCKS_PER_CU - PT_WAVES_PER_BLOCK * PT_WAVE_STACK * 512) / 16)
#ifndef PT_REFILL_MIN
#define PT_REFILL_MIN 16     // idle lanes that trigger a refill
#endif
#ifndef PT_RESERVE
#define PT_RESERVE 256       // rays a wave reserves from the queue per atomic, at most
#endif
#ifndef PT_RESERVE_MIN
#define PT_RESERVE_MIN 64    // and at least: short queues are handed out in smaller portions
#endif
#ifndef PT_FETCH_STRIDE
#define PT_FETCH_STRIDE 32   // words between the fetch counters of the eighths (one 128-byte line each)
#endif
#ifndef PT_TRIS_PER_PASS
#define PT_TRIS_PER_PASS 1   // triangle tests per lane and pass
#endif
#ifndef PT_PIPELINE
#define PT_PIPELINE 0        // 1: the node of the next pass is requested before the triangle test of this one
#endif

// Bits of the error word that the traversal sets when it had to give up on rays
#define PT_ERROR_STACK_OVERFLOW 1u // a traversal stack deeper than stack_capacity entries
#define PT_ERROR_PASS_LIMIT 2u     // a wave ran into the bound on its passes

struct PtWaveLds {
	u32x2 stack[PT_WAVE_STACK][64];
};

// The LDS of one workgroup: BVH prefix + 16 * 3 KB of stacks = 160 KB / PT_TRACE_BLOCKS_PER_CU
struct PtTraceLds {
	u32x4 cache[PT_CACHE_WORDS];
	PtWaveLds wave[PT_WAVES_PER_BLOCK];
};

// How much of the BVH the LDS prefix holds: nodes first, triangles after them
struct PtCacheLayout {
	uint32_t node_count; // nodes [0, node_count) live at cache[5 * index]
	uint32_t tri_count;  // triangles [0, tri_count) live at cache[tri_offset + 3 * index]
	uint32_t tri_offset;
};

__host__ __device__ inline PtCacheLayout pt_cache_layout(uint32_t bvh_node_count, uint32_t bvh_triangle_count) {
	PtCacheLayout l;
	l.node_count = (bvh_node_count < PT_CACHE_WORDS / 5) ? bvh_node_count : PT_CACHE_WORDS / 5;
	l.tri_offset = 5 * l.node_count;
	uint32_t room = (PT_CACHE_WORDS - l.tri_offset) / 3;
	l.tri_count = (bvh_triangle_count < room) ? bvh_triangle_count : room;
	return l;
}

// Copies the BVH prefix into LDS; the whole workgroup takes part and is
// synchronised once, after which its waves run independently
__device__ __forceinline__ void pt_fill_cache(PtTraceLds& lds, const PtCacheLayout& layout, const PtNode* nodes, const PtTri* tris) {
	const u32x4* node_words = (const u32x4*) nodes;
	const u32x4* tri_words = (const u32x4*) tris;
	for (uint32_t i = threadIdx.x; i < 5 * layout.node_count; i += PT_TRACE_BLOCK)
		lds.cache[i] = node_words[i];
	for (uint32_t i = threadIdx.x; i < 3 * layout.tri_count; i += PT_TRACE_BLOCK)
		lds.cache[layout.tri_offset + i] = tri_words[i];
	__syncthreads();
}

// Where the rays of one launch come from
struct PtRayQueue {
	const f32x4* origin;    // xyz + a word that is handed to finish() (0xffffffff: skip this slot)
	const f32x4* direction; // xyz + a word that is handed to finish()
	const uint32_t* count;  // number of rays, in device memory
	uint32_t* fetch;        // 8 fetch counters (one per eighth, PT_FETCH_STRIDE words apart), zeroed before the launch
};

// Pointers with an explicit address space: a load through them is a ds_read
// or a global_load by construction. With generic pointers the compiler merges
// the LDS and the HBM branch of a fetch into one flat_load, which sends LDS
// accesses through the vector memory path.
typedef uint32_t pt_vec4 __attribute__((ext_vector_type(4)));
typedef uint32_t pt_vec2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) pt_vec4 pt_lds_vec4;
typedef __attribute__((address_space(1))) pt_vec4 pt_global_vec4;
typedef __attribute__((address_space(3))) pt_vec2 pt_lds_vec2;
typedef __attribute__((address_space(1))) pt_vec2 pt_global_vec2;

__device__ __forceinline__ u32x4 pt_to_u32x4(pt_vec4 v) { u32x4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }
__device__ __forceinline__ u32x4 pt_load_lds(const u32x4* p) { return pt_to_u32x4(*(const pt_lds_vec4*) p); }
__device__ __forceinline__ u32x4 pt_load_global(const u32x4* p) { return pt_to_u32x4(*(const pt_global_vec4*) p); }

// Loads and stores of the ray queues, which a launch touches once: marked
// non-temporal, so that the caches prefer to keep the BVH. (The hit records
// are not: shading reads them right after the traversal has written them.)
typedef float pt_float4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 pt_load_streaming(const f32x4* p) {
	pt_float4 v = __builtin_nontemporal_load((const pt_float4*) p);
	f32x4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
	return r;
}
__device__ __forceinline__ void pt_store_streaming(f32x4* p, f32x4 value) {
	pt_float4 v = { value.x, value.y, value.z, value.w };
	__builtin_nontemporal_store(v, (pt_float4*) p);
}

// a where the mask is all ones, b where it is zero. One three-operand bit
// instruction, which this part executes at the rate of an addition, where a
// conditional move by a lane mask (and the comparison that makes the mask)
// takes twice as long each.
__device__ __forceinline__ uint32_t pt_select_bits(uint32_t mask, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(mask, a, b, 0xca); }
__device__ __forceinline__ float pt_select_bits(uint32_t mask, float a, float b) { return pt_as_float(pt_select_bits(mask, pt_as_uint(a), pt_as_uint(b))); }

// The ray-dependent part of pt_intersect_triangle() for one lane. Instead of
// three component indices, which make every vertex component a choice between
// three registers by two comparisons, the lane keeps masks: two that rotate
// the dominant axis of the ray into z and one that swaps x and y afterwards
// (pt_ray_shear() swaps them for a ray that runs against its dominant axis).
struct PtLaneShear {
	uint32_t rotate_0, rotate_1; // all ones if the dominant axis is x / y
	uint32_t swap;               // all ones if kx is the axis after ky
	float sx, sy, sz;
};

__device__ __forceinline__ PtLaneShear pt_lane_shear(v3 dir) {
	PtRayShear r = pt_ray_shear(dir);
	PtLaneShear l;
	l.rotate_0 = (r.kz == 0) ? 0xffffffffu : 0u;
	l.rotate_1 = (r.kz == 1) ? 0xffffffffu : 0u;
	l.swap = (r.kx != ((r.kz == 2) ? 0 : r.kz + 1)) ? 0xffffffffu : 0u;
	l.sx = r.sx; l.sy = r.sy; l.sz = r.sz;
	return l;
}

// The components kx, ky, kz of v
__device__ __forceinline__ v3 pt_permute(const PtLaneShear& r, v3 v) {
	float x = pt_select_bits(r.rotate_0, v.y, pt_select_bits(r.rotate_1, v.z, v.x));
	float y = pt_select_bits(r.rotate_0, v.z, pt_select_bits(r.rotate_1, v.x, v.y));
	float z = pt_select_bits(r.rotate_0, v.x, pt_select_bits(r.rotate_1, v.y, v.z));
	return mk3(pt_select_bits(r.swap, y, x), pt_select_bits(r.swap, x, y), z);
}

// pt_intersect_triangle() with the lane's masks in place of indices: the same
// operations on the same values
__device__ __forceinline__ void pt_intersect_triangle_lane(PtHit& hit, const PtLaneShear& r, v3 origin, v3 p0, v3 p1, v3 p2, uint32_t prim, float tmin) {
	v3 a = pt_permute(r, p0 - origin), b = pt_permute(r, p1 - origin), c = pt_permute(r, p2 - origin);
	float az = a.z, bz = b.z, cz = c.z;
	float ax = a.x - r.sx * az, ay = a.y - r.sy * az;
	float bx = b.x - r.sx * bz, by = b.y - r.sy * bz;
	float cx = c.x - r.sx * cz, cy = c.y - r.sy * cz;
	float u = cx * by - cy * bx;
	float v = ax * cy - ay * cx;
	float w = bx * ay - by * ax;
	if ((u < 0.0f || v < 0.0f || w < 0.0f) && (u > 0.0f || v > 0.0f || w > 0.0f))
		return;
	float det = (u + v) + w;
	if (det == 0.0f)
		return;
	float tz = (u * (r.sz * az) + v * (r.sz * bz)) + w * (r.sz * cz);
	float rcp_det = 1.0f / det;
	float t = tz * rcp_det;
	if (!(t > tmin))
		return;
	if (t < hit.t || (t == hit.t && prim < hit.prim)) {
		hit.t = t;
		hit.prim = prim;
		hit.u = v * rcp_det;
		hit.v = w * rcp_det;
		hit.front = det > 0.0f;
	}
}

__device__ __forceinline__ uint32_t pt_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Calls finish(slot, origin_w, direction_w, hit) once per ray of the queue.
//
// EMISSIVE_ONLY is for rays of which nothing is asked but whether the triangle
// of their closest hit is emissive (trace_ray_emission() of
// pathtrace.frag.glsl:203-219): hit.prim is then *a* triangle that answers the
// question like the closest one, or 0xffffffff for no hit. Nodes and triangles
// carry marks for it (pt_mark_emissive_*: bit i of a node's first padding byte
// says that an emissive triangle lies below child i, a triangle's first padding
// word that it is emissive). While the closest hit found so far is not
// emissive, only something emissive in front of it could change the answer, so
// children and triangles without the mark are skipped; most rays are blocked
// by a wall early and then have little left to look at. Should the closest hit
// turn out emissive after something was skipped, an occluder may have been
// among the skipped: the ray then walks the tree a second time, up to that
// hit, and ends at the first triangle in front of it that is not emissive.
// With 48 % of the shadow rays blocked (generated city) that saves 17 % of the
// node visits and 12 % of the time; with 12 to 17 % blocked (Cornell box,
// generated interior) it costs 7 to 10 %.
// AIMED (and the argument aimed): the fourth word of a ray's direction says how far away the light
// begins that the ray was aimed at. A hit beyond that distance does not start
// the skipping: the lamp is likely to turn up in front of it, and every ray
// that finds its lamp after something was skipped pays with a second walk
// (9 % of all node steps of the city's shadow rays were second walks; with
// this rule hardly any are left). Where the lights have no emissive geometry
// to find, the rule only delays the skipping (100 M-triangle city with 10 000
// generated lights: +30 % time), so emissive_counts[2] counts the second walks
// and the host turns the rule on where they are frequent.
template <bool COUNT_VISITS, bool EMISSIVE_ONLY, bool AIMED = false, typename Finish>
__device__ __forceinline__ void pt_trace_wave(PtWaveLds& lds, const u32x4* cache, const PtCacheLayout layout, const PtNode* __restrict__ nodes, const PtTri* __restrict__ tris, uint32_t triangle_count,
	const PtRayQueue& queue, u32x2* stack_overflow, uint32_t stack_overflow_stride, uint32_t stack_capacity,
	unsigned long long* visit_counters, uint32_t* error_word, Finish finish, unsigned long long* emissive_counts = NULL, bool aimed = false)
{
	// EMISSIVE_ONLY: rays traced, and rays that ended at a hit that is not
	// emissive, are added to emissive_counts[0] and [1]: the host leaves the
	// skipping off for scenes in which few rays end so (lights mostly in view:
	// little to skip, and keeping track of it costs more than it saves)
	uint32_t finished_count = 0, blocked_count = 0, second_walk_count = 0;
	const uint32_t lane = pt_lane();
	const uint32_t ray_total = *queue.count;
	// Eighths of the queue, in multiples of the reservation size
	// A wave reserves rays in portions: large ones keep the atomics rare,
	// but a queue that is short for the number of waves must be split finely
	// enough for the waves to share it evenly (about eight portions per wave)
	uint32_t reserve = PT_RESERVE;
	while (reserve > PT_RESERVE_MIN && (uint64_t) reserve * gridDim.x * PT_WAVES_PER_BLOCK * 8u > ray_total)
		reserve >>= 1;
	const uint32_t chunk_len = ((ray_total + 8 * reserve - 1) / (8 * reserve)) * reserve;
	uint32_t chunk = blockIdx.x & 7, chunks_tried = 0;
	// Rays reserved by this wave and not yet handed to a lane (wave-uniform)
	uint32_t reserve_next = 0, reserve_end = 0;
	bool rays_remain = ray_total != 0;
	const float tmin = 1.0e-3f;
	// Per-lane ray state. A ray that is complete stays in its lane (unfinished)
	// until the wave refills or ends: finish() then runs for many lanes at once
	// instead of in nearly every pass for the one or two rays that just ended
	bool active = false, unfinished = false;
	uint32_t slot = 0;
	float origin_w = 0.0f, direction_w = 0.0f;
	v3 origin = mk3(0.0f, 0.0f, 0.0f);
	float idx = 0.0f, idy = 0.0f, idz = 0.0f;
	// All ones for a ray that runs against the axis
	uint32_t neg_x = 0, neg_y = 0, neg_z = 0;
	uint32_t octinv = 7;
	PtLaneShear shear;
	shear.rotate_0 = shear.rotate_1 = shear.swap = 0;
	shear.sx = shear.sy = shear.sz = 0.0f;
	PtHit hit;
	hit.prim = 0xffffffffu;
	hit.t = 1e38f;
	hit.u = hit.v = 0.0f;
	hit.front = false;
	u32x2 ngroup, tgroup;
	ngroup.x = ngroup.y = tgroup.x = tgroup.y = 0;
	uint32_t sp = 0;
	// EMISSIVE_ONLY: bit 0: the closest hit so far is not emissive; bit 1:
	// something was skipped on that account; bit 2: this is the second walk
	uint32_t emissive_state = 0;
	uint32_t node_visits = 0, tri_visits = 0, lds_node_visits = 0, lds_tri_visits = 0, empty_node_visits = 0, culled_by_hit_visits = 0, distinct_node_fetches = 0;
	uint32_t skipping_node_visits = 0, second_walk_node_visits = 0;
	uint32_t pass_count = 0, refill_count = 0, node_pass_count = 0, tri_pass_count = 0;
	(void) node_pass_count; (void) tri_pass_count;
	unsigned long long active_lane_passes = 0;

	// The node step in three parts. select_node(): which node a lane without
	// pending triangles looks at next (the next child of the current group or,
	// when that is used up, of the group on top of the stack); false if there is
	// none. load_node() requests its words, test_node() tests its eight boxes
	// and leaves the children that were hit in ngroup and tgroup.
	u32x4 w0, w1, w2, w3, w4;
	uint32_t node_index;
	// PT_PIPELINE: the words of a node are on their way (or there) and its boxes not yet tested
	bool fetched = false;
	auto select_node = [&]() -> bool {
		if (EMISSIVE_ONLY && (emissive_state & 1u)) {
			// Of the children still to be entered, only the marked ones (the
			// second byte holds the marks in the order of the first)
			uint32_t marked = ngroup.y & ((ngroup.y << 16) | 0x00ffffffu);
			emissive_state |= (marked != ngroup.y) ? 2u : 0u;
			ngroup.y = marked;
		}
		if (!(ngroup.y & 0xff000000u) && sp != 0) {
			--sp;
			pt_vec2 entry;
			if (sp < PT_WAVE_STACK)
				entry = *(const pt_lds_vec2*) &lds.stack[sp][lane];
			else
				entry = *(const pt_global_vec2*) &stack_overflow[(uint64_t) (sp - PT_WAVE_STACK) * stack_overflow_stride];
			ngroup.x = entry.x;
			ngroup.y = entry.y;
			if (EMISSIVE_ONLY && (emissive_state & 1u)) {
				uint32_t marked = ngroup.y & ((ngroup.y << 16) | 0x00ffffffu);
				emissive_state |= (marked != ngroup.y) ? 2u : 0u;
				ngroup.y = marked;
			}
		}
		if (!(ngroup.y & 0xff000000u))
			return false;
		uint32_t bit = 31u - (uint32_t) __builtin_clz(ngroup.y);
		ngroup.y &= ~(1u << bit);
		if (ngroup.y & 0xff000000u) {
			pt_vec2 entry;
			entry.x = ngroup.x;
			entry.y = ngroup.y;
			if (sp < PT_WAVE_STACK)
				*(pt_lds_vec2*) &lds.stack[sp][lane] = entry;
			else if (sp < stack_capacity)
				*(pt_global_vec2*) &stack_overflow[(uint64_t) (sp - PT_WAVE_STACK) * stack_overflow_stride] = entry;
			if (sp < stack_capacity)
				++sp;
			else
				atomicOr(error_word, PT_ERROR_STACK_OVERFLOW);
		}
		uint32_t child_slot = (bit - 24u) ^ octinv;
		uint32_t rel = (uint32_t) __builtin_popcount(ngroup.y & 0xffu & ((1u << child_slot) - 1u));
		node_index = ngroup.x + rel;
		if (COUNT_VISITS) {
			++node_visits;
			lds_node_visits += (node_index < layout.node_count) ? 1u : 0u;
			// Node steps of rays that only look for something emissive in front of their hit, and of second walks
			skipping_node_visits += (EMISSIVE_ONLY && (emissive_state & 1u)) ? 1u : 0u;
			second_walk_node_visits += (EMISSIVE_ONLY && (emissive_state & 4u)) ? 1u : 0u;
			// How many different nodes the lanes of this pass fetch: what a walk
			// that shared the fetch and the stack among the lanes would visit
			unsigned long long todo = __ballot(true);
			while (todo) {
				uint32_t first = (uint32_t) __builtin_amdgcn_readlane((int) node_index, (int) __builtin_ctzll(todo));
				todo &= ~__ballot(node_index == first);
				++distinct_node_fetches;
			}
		}
		return true;
	};
	auto load_node = [&]() {
		if (node_index < layout.node_count) {
			const u32x4* words = &cache[5 * node_index];
			w0 = pt_load_lds(words); w1 = pt_load_lds(words + 1); w2 = pt_load_lds(words + 2); w3 = pt_load_lds(words + 3); w4 = pt_load_lds(words + 4);
		}
		else {
			const u32x4* words = (const u32x4*) &nodes[node_index];
			w0 = pt_load_global(words); w1 = pt_load_global(words + 1); w2 = pt_load_global(words + 2); w3 = pt_load_global(words + 3); w4 = pt_load_global(words + 4);
		}
	};
	auto test_node = [&]() {
		float sx = pt_as_float((w0.w & 0xffu) << 23), sy = pt_as_float(((w0.w >> 8) & 0xffu) << 23), sz = pt_as_float(((w0.w >> 16) & 0xffu) << 23);
		uint32_t imask = w0.w >> 24;
		float ax = idx * sx, ay = idy * sy, az = idz * sz;
		float dx = pt_as_float(w0.x) - origin.x, dy = pt_as_float(w0.y) - origin.y, dz = pt_as_float(w0.z) - origin.z;
		float bx = dx * idx, by = dy * idy, bz = dz * idz;
		// Near planes a little nearer, far planes a little farther (pt_box_test_slack())
		float slack = 2.0f * pt_box_test_slack(dx, dy, dz, sx, sy, sz);
		float ex = slack * fabsf(idx), ey = slack * fabsf(idy), ez = slack * fabsf(idz);
		float bxn = bx - ex, byn = by - ey, bzn = bz - ez, bxf = bx + ex, byf = by + ey, bzf = bz + ez;
		// Near and far planes per axis, chosen once per node by the ray's signs
		uint32_t nqx[2] = { pt_select_bits(neg_x, w3.z, w2.x), pt_select_bits(neg_x, w3.w, w2.y) }, fqx[2] = { pt_select_bits(neg_x, w2.x, w3.z), pt_select_bits(neg_x, w2.y, w3.w) };
		uint32_t nqy[2] = { pt_select_bits(neg_y, w4.x, w2.z), pt_select_bits(neg_y, w4.y, w2.w) }, fqy[2] = { pt_select_bits(neg_y, w2.z, w4.x), pt_select_bits(neg_y, w2.w, w4.y) };
		uint32_t nqz[2] = { pt_select_bits(neg_z, w4.z, w3.x), pt_select_bits(neg_z, w4.w, w3.y) }, fqz[2] = { pt_select_bits(neg_z, w3.x, w4.z), pt_select_bits(neg_z, w3.y, w4.w) };
		uint32_t leafmask = w1.z & 0xffu;
		// Eight box tests back to back without branches: bit i of hit8 says
		// whether child slot i was hit (empty slots are masked out below).
		// tn <= tf is read off the sign of tf - tn, shifted into the mask
		// from the last child to the first: a subtraction and a funnel
		// shift per child instead of a comparison, a conditional move and
		// an or. tn and tf are no NaNs (the maxima and minima drop them),
		// equal values give +0, and should two infinities of one sign make
		// the difference a NaN, its sign bit is clear: the child counts as
		// hit, which costs a visit and no more.
		uint32_t missed8 = 0;
#pragma unroll
		for (int i = 7; i >= 0; --i) {
			float nx = fmaf(pt_byte_to_float(nqx[i >> 2], i & 3), ax, bxn), fx = fmaf(pt_byte_to_float(fqx[i >> 2], i & 3), ax, bxf);
			float ny = fmaf(pt_byte_to_float(nqy[i >> 2], i & 3), ay, byn), fy = fmaf(pt_byte_to_float(fqy[i >> 2], i & 3), ay, byf);
			float nz = fmaf(pt_byte_to_float(nqz[i >> 2], i & 3), az, bzn), fz = fmaf(pt_byte_to_float(fqz[i >> 2], i & 3), az, bzf);
			float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, tmin));
			float tf = fminf(fminf(fx, fy), fminf(fz, hit.t));
			missed8 = __builtin_amdgcn_alignbit(missed8, pt_as_uint(tf - tn), 31);
		}
		uint32_t hit8 = ~missed8 & 0xffu;
		if (COUNT_VISITS) {
			// Node steps that found nothing, and among them those that would have found
			// something without the closest hit so far (the node was put on the stack
			// before that hit was known)
			bool empty = (hit8 & (imask | leafmask)) == 0u;
			empty_node_visits += empty ? 1u : 0u;
			uint32_t missed8_unbounded = 0;
#pragma unroll
			for (int i = 7; i >= 0; --i) {
				float nx = fmaf(pt_byte_to_float(nqx[i >> 2], i & 3), ax, bxn), fx = fmaf(pt_byte_to_float(fqx[i >> 2], i & 3), ax, bxf);
				float ny = fmaf(pt_byte_to_float(nqy[i >> 2], i & 3), ay, byn), fy = fmaf(pt_byte_to_float(fqy[i >> 2], i & 3), ay, byf);
				float nz = fmaf(pt_byte_to_float(nqz[i >> 2], i & 3), az, bzn), fz = fmaf(pt_byte_to_float(fqz[i >> 2], i & 3), az, bzf);
				float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, tmin));
				float tf = fminf(fminf(fx, fy), fz);
				missed8_unbounded = __builtin_amdgcn_alignbit(missed8_unbounded, pt_as_uint(tf - tn), 31);
			}
			culled_by_hit_visits += (empty && (~missed8_unbounded & (imask | leafmask) & 0xffu) != 0u) ? 1u : 0u;
		}
		if (EMISSIVE_ONLY && (emissive_state & 1u)) {
			uint32_t marked = hit8 & (w1.z >> 8);
			emissive_state |= (marked != hit8) ? 2u : 0u;
			hit8 = marked;
		}
		// Inner children: from slot order to visiting order (slot ^ octinv),
		// three conditional swaps of neighbouring bits, pairs and nibbles
		// (with EMISSIVE_ONLY the marks of those children ride along in the
		// second byte and end up in the same order)
		uint32_t order = (hit8 & imask) | (EMISSIVE_ONLY ? (w1.z & (imask << 8)) : 0u);
		uint32_t swap = ((order >> 1) ^ order) & (0x5555u & ~neg_x);
		order ^= swap | (swap << 1);
		swap = ((order >> 2) ^ order) & (0x3333u & ~neg_y);
		order ^= swap | (swap << 2);
		swap = ((order >> 4) ^ order) & (0x0f0fu & ~neg_z);
		order ^= swap | (swap << 4);
		ngroup.x = w1.x;
		ngroup.y = (order << 24) | (order & 0xff00u) | imask;
		// Triangles pending in the low byte (by slot), the leaf mask above
		// it: a triangle's index is tri_base plus the leaves below its slot
		tgroup.x = w1.y;
		tgroup.y = (hit8 & leafmask) | (leafmask << 8);
	};

	// Every pass retires, fetches, walks or tests something, so the loop ends
	// when the queue is used up; the bound only guards the GPU against a logic
	// error ever hanging it. Running into it, or into the end of the stack,
	// would leave rays without their closest hit: both set a bit of the error
	// word, which the host checks when it waits for the work.
	bool drained = false;
	for (uint32_t pass = 0; pass != (1u << 26); ++pass) {
		// ---- refill idle lanes ----
		unsigned long long idle_mask = __ballot(!active);
		if (COUNT_VISITS) {
			++pass_count;
			active_lane_passes += 64u - (uint32_t) __popcll(idle_mask);
		}
		if (rays_remain && ((uint32_t) __popcll(idle_mask) >= PT_REFILL_MIN)) {
			if (COUNT_VISITS)
				++refill_count;
			if (EMISSIVE_ONLY) {
				// A completed ray keeps its state until its lane gets the next ray
				finished_count += (uint32_t) __popcll(__ballot(unfinished));
				blocked_count += (uint32_t) __popcll(__ballot(unfinished && (emissive_state & 1u)));
			}
			if (unfinished) {
				finish(slot, origin_w, direction_w, hit);
				unfinished = false;
			}
			if (reserve_next == reserve_end) {
				// Reserve the next block of rays of this wave's current eighth
				uint32_t chunk_begin = chunk * chunk_len;
				uint32_t chunk_end = chunk_begin + chunk_len;
				if (chunk_end > ray_total) chunk_end = ray_total;
				uint32_t base = chunk_len;
				if (chunk_begin < chunk_end) {
					if (lane == 0)
						base = atomicAdd(&queue.fetch[chunk * PT_FETCH_STRIDE], reserve);
					base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
				}
				if (chunk_begin + base < chunk_end) {
					reserve_next = chunk_begin + base;
					reserve_end = reserve_next + reserve;
					if (reserve_end > chunk_end) reserve_end = chunk_end;
				}
				else {
					chunk = (chunk + 1) & 7;
					if (++chunks_tried == 8)
						rays_remain = false;
				}
			}
			uint32_t available = reserve_end - reserve_next;
			if (available) {
				uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) idle_mask, 0u));
				uint32_t want = (uint32_t) __popcll(idle_mask);
				if (!active && rank < available) {
					slot = reserve_next + rank;
					f32x4 o = pt_load_streaming(&queue.origin[slot]), d = pt_load_streaming(&queue.direction[slot]);
					origin_w = o.w;
					direction_w = d.w;
					active = true;
					sp = 0;
					emissive_state = 0;
					tgroup.x = tgroup.y = 0;
					hit.prim = 0xffffffffu;
					hit.t = 1e38f;
					hit.u = hit.v = 0.0f;
					hit.front = false;
					if (pt_as_uint(o.w) == 0xffffffffu || triangle_count == 0) {
						// Padding path or empty scene: a miss without traversal
						ngroup.x = ngroup.y = 0;
					}
					else {
						origin = mk3(o.x, o.y, o.z);
						shear = pt_lane_shear(mk3(d.x, d.y, d.z));
						// The box test may be conservative, so tiny direction
						// components are clamped to keep reciprocals finite
						float cdx = (fabsf(d.x) > 1.0e-30f) ? d.x : copysignf(1.0e-30f, d.x);
						float cdy = (fabsf(d.y) > 1.0e-30f) ? d.y : copysignf(1.0e-30f, d.y);
						float cdz = (fabsf(d.z) > 1.0e-30f) ? d.z : copysignf(1.0e-30f, d.z);
						// The reciprocals only serve the box test, which errs on the side
						// of a hit: the hardware's reciprocal (one unit in the last place,
						// one instruction where the correctly rounded quotient takes
						// twelve) is paid for with twice the slack of pt_box_test_slack()
						idx = __builtin_amdgcn_rcpf(cdx); idy = __builtin_amdgcn_rcpf(cdy); idz = __builtin_amdgcn_rcpf(cdz);
						neg_x = (cdx < 0.0f) ? 0xffffffffu : 0u;
						neg_y = (cdy < 0.0f) ? 0xffffffffu : 0u;
						neg_z = (cdz < 0.0f) ? 0xffffffffu : 0u;
						octinv = 7u - ((neg_x & 1u) | (neg_y & 2u) | (neg_z & 4u));
						ngroup.x = 0;
						ngroup.y = 0x80000000u;
#if PT_PIPELINE
						// The root (in the LDS prefix) at once, so that its boxes are tested in this pass
						if (select_node()) {
							load_node();
							fetched = true;
						}
#endif
					}
				}
				reserve_next += (want < available) ? want : available;
			}
		}
		// ---- exit: nothing in flight and nothing left to fetch ----
		if (!rays_remain && __ballot(active) == 0) {
			if (EMISSIVE_ONLY) {
				finished_count += (uint32_t) __popcll(__ballot(unfinished));
				blocked_count += (uint32_t) __popcll(__ballot(unfinished && (emissive_state & 1u)));
			}
			if (unfinished)
				finish(slot, origin_w, direction_w, hit);
			drained = true;
			break;
		}

#if PT_PIPELINE
		// ---- boxes of the node that was requested in the previous pass ----
		if (COUNT_VISITS)
			node_pass_count += __ballot(fetched) != 0;
		if (fetched) {
			test_node();
			fetched = false;
		}
		// ---- request one triangle per lane that has triangles pending ----
		bool triangle_requested = active && (tgroup.y & 0xffu) != 0;
		if (COUNT_VISITS)
			tri_pass_count += __ballot(triangle_requested) != 0;
		u32x4 t0, t1, t2;
		if (triangle_requested) {
			uint32_t bit = (uint32_t) __builtin_ctz(tgroup.y);
			tgroup.y &= tgroup.y - 1u;
			uint32_t tri_index = tgroup.x + (uint32_t) __builtin_popcount((tgroup.y >> 8) & ((1u << bit) - 1u));
			if (COUNT_VISITS) {
				++tri_visits;
				lds_tri_visits += (tri_index < layout.tri_count) ? 1u : 0u;
			}
			if (tri_index < layout.tri_count) {
				const u32x4* words = &cache[layout.tri_offset + 3 * tri_index];
				t0 = pt_load_lds(words); t1 = pt_load_lds(words + 1); t2 = pt_load_lds(words + 2);
			}
			else {
				const u32x4* words = (const u32x4*) &tris[tri_index];
				t0 = pt_load_global(words); t1 = pt_load_global(words + 1); t2 = pt_load_global(words + 2);
			}
		}
		// ---- request the node of the next pass, behind the triangle: the triangle test
		// below waits for the triangle only and runs while the node is on its way.
		// Which node comes next does not depend on that test (the closest hit so far
		// enters the box test, which comes after it as before; with EMISSIVE_ONLY a
		// child may be entered that the test would have ruled out, which costs a visit) ----
		if (active && (tgroup.y & 0xffu) == 0) {
			if (select_node()) {
				load_node();
				fetched = true;
			}
		}
		// ---- test the triangle ----
		if (triangle_requested) {
			pt_intersect_triangle_lane(hit, shear, origin,
				mk3(pt_as_float(t0.x), pt_as_float(t0.y), pt_as_float(t0.z)),
				mk3(pt_as_float(t0.w), pt_as_float(t1.x), pt_as_float(t1.y)),
				mk3(pt_as_float(t1.z), pt_as_float(t1.w), pt_as_float(t2.x)), t2.y, tmin);
			if (EMISSIVE_ONLY && hit.prim == t2.y) {
				// This triangle is the closest hit now
				if (t2.z != 0u)
					emissive_state &= ~1u;
				else if (emissive_state & 4u) {
					// Second walk: an occluder in front of the emissive hit answers the question
					sp = 0;
					ngroup.y = 0;
					tgroup.y = 0;
					fetched = false;
				}
				else if (!(AIMED && aimed) || hit.t < direction_w)
					emissive_state |= 1u;
			}
		}

#else
		if (COUNT_VISITS)
			node_pass_count += __ballot(active && (tgroup.y & 0xffu) == 0) != 0;
		// ---- one node per lane that has no triangles pending ----
		if (active && (tgroup.y & 0xffu) == 0) {
			if (select_node()) {
				load_node();
				test_node();
			}
		}

		if (COUNT_VISITS)
			tri_pass_count += __ballot(active && (tgroup.y & 0xffu) != 0) != 0;
		// ---- one triangle per lane that has triangles pending ----
		for (int round = 0; round != PT_TRIS_PER_PASS; ++round)
		if (active && (tgroup.y & 0xffu) != 0) {
			uint32_t bit = (uint32_t) __builtin_ctz(tgroup.y);
			tgroup.y &= tgroup.y - 1u;
			uint32_t tri_index = tgroup.x + (uint32_t) __builtin_popcount((tgroup.y >> 8) & ((1u << bit) - 1u));
			if (COUNT_VISITS) {
				++tri_visits;
				lds_tri_visits += (tri_index < layout.tri_count) ? 1u : 0u;
			}
			u32x4 t0, t1, t2;
			if (tri_index < layout.tri_count) {
				const u32x4* words = &cache[layout.tri_offset + 3 * tri_index];
				t0 = pt_load_lds(words); t1 = pt_load_lds(words + 1); t2 = pt_load_lds(words + 2);
			}
			else {
				const u32x4* words = (const u32x4*) &tris[tri_index];
				t0 = pt_load_global(words); t1 = pt_load_global(words + 1); t2 = pt_load_global(words + 2);
			}
			pt_intersect_triangle_lane(hit, shear, origin,
				mk3(pt_as_float(t0.x), pt_as_float(t0.y), pt_as_float(t0.z)),
				mk3(pt_as_float(t0.w), pt_as_float(t1.x), pt_as_float(t1.y)),
				mk3(pt_as_float(t1.z), pt_as_float(t1.w), pt_as_float(t2.x)), t2.y, tmin);
			if (EMISSIVE_ONLY && hit.prim == t2.y) {
				// This triangle is the closest hit now
				if (t2.z != 0u)
					emissive_state &= ~1u;
				else if (emissive_state & 4u) {
					// Second walk: an occluder in front of the emissive hit answers the question
					sp = 0;
					ngroup.y = 0;
					tgroup.y = 0;
				}
				else if (!(AIMED && aimed) || hit.t < direction_w)
					emissive_state |= 1u;
			}
		}

#endif
		// ---- retire complete rays ----
		const bool complete = active && !fetched && !(ngroup.y & 0xff000000u) && sp == 0 && (tgroup.y & 0xffu) == 0;
		const bool walks_again = EMISSIVE_ONLY && complete && emissive_state == 2u;
		if (EMISSIVE_ONLY)
			second_walk_count += (uint32_t) __popcll(__ballot(walks_again));
		if (complete) {
			if (walks_again) {
				// Emissive, but not everything in front of it has been looked at
				emissive_state = 4u;
				ngroup.x = 0;
				ngroup.y = 0x80000000u;
				tgroup.x = tgroup.y = 0;
			}
			else {
				active = false;
				unfinished = true;
			}
		}
	}
	if (!drained && lane == 0)
		atomicOr(error_word, PT_ERROR_PASS_LIMIT);
	if (EMISSIVE_ONLY && lane == 0) {
		atomicAdd(&emissive_counts[0], (unsigned long long) finished_count);
		atomicAdd(&emissive_counts[1], (unsigned long long) blocked_count);
		atomicAdd(&emissive_counts[2], (unsigned long long) second_walk_count);
	}
	if (COUNT_VISITS) {
		atomicAdd(&visit_counters[0], (unsigned long long) node_visits);
		atomicAdd(&visit_counters[1], (unsigned long long) tri_visits);
		atomicAdd(&visit_counters[20], (unsigned long long) lds_node_visits);
		atomicAdd(&visit_counters[21], (unsigned long long) lds_tri_visits);
		atomicAdd(&visit_counters[12], (unsigned long long) empty_node_visits);
		atomicAdd(&visit_counters[13], (unsigned long long) culled_by_hit_visits);
		atomicAdd(&visit_counters[28], (unsigned long long) skipping_node_visits);
		atomicAdd(&visit_counters[29], (unsigned long long) second_walk_node_visits);
		if (lane == 0) {
			// Diagnostics: passes, lanes active in them, refills (summed over waves)
			atomicAdd(&visit_counters[6], (unsigned long long) pass_count);
			atomicAdd(&visit_counters[7], active_lane_passes);
			atomicAdd(&visit_counters[25], (unsigned long long) refill_count); // not [10]: the shadow rays' counters start two words on, and [12] is taken
			atomicAdd(&visit_counters[16], (unsigned long long) node_pass_count);
			atomicAdd(&visit_counters[17], (unsigned long long) tri_pass_count);
			atomicAdd(&visit_counters[24], (unsigned long long) distinct_node_fetches);
		}
	}
}
