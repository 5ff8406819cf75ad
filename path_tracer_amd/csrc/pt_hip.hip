// MI355X (gfx950) kernels and the C ABI of the light transport library.
//
// Wavefront organisation of the reference's fragment shader
// (src/shaders/pathtrace.frag.glsl): one batch holds all pixels of one or
// more frames as independent paths. Per batch:
//   raygen                      seeds, pixel jitter, camera rays (:365-386)
//   for every path vertex k:    (loop bodies of :224-362)
//     trace extension rays      closest hits via the 8-wide BVH (:181-199)
//     shade                     shading data, emission, NEE + BRDF sampling
//     trace shadow rays         closest hit, emission of what was hit (:203-219)
//   resolve                     frame-ordered sum into the RGBA32F target
// Queues of rays live in HBM as structure-of-arrays float4 streams; surviving
// paths are compacted with a wave ballot and one atomic per wave.
#include "pt_common.h"
#include "pt_bvh_build.h"
#include "pt_trace.h"
#include "pt_shade.h"
#include "pt_light_tree.h"
#include "pt_trace_wave.h"
#include "pt_shade_kernel.h"

#ifndef PT_SHORT_BATCH_PATHS
#define PT_SHORT_BATCH_PATHS (12ull << 20) // batches below this get one traversal workgroup per CU and two streams
#endif
#ifndef PT_BATCH_TARGET_PATHS
#define PT_BATCH_TARGET_PATHS (64ull << 20)
#endif
#include "../../include/pt_hip.h"
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <exception>
#include <new>
#include <thread>
#include <vector>

static_assert(sizeof(pt_constants_t) == sizeof(PtConstants), "ABI and device constants must agree");

#define PT_EMISSIVE_COUNTS_OFFSET 50 // three 64-bit counts of k_trace_shadow<..., true> among the queue counters (words 50 to 55), then those of k_trace_extension<..., true> (56 to 61)

#define PT_CHECK(call) do { hipError_t pt_err_ = (call); if (pt_err_ != hipSuccess) { \
	printf("HIP error %s at %s:%d (%s).\n", hipGetErrorString(pt_err_), __FILE__, __LINE__, #call); return 1; } } while (0)

// Makes a GPU current for an entry point and forgets the HIP runtime's
// "last error" of this thread: other users of the runtime in the same process
// (RCCL and PyTorch probe optional features and leave such errors behind)
// must not make this library's own checks of hipGetLastError() fail.
static hipError_t pt_select_device(int device_index) {
	hipError_t result = hipSetDevice(device_index);
	(void) hipGetLastError();
	return result;
}

// Runs the body of an entry point of the C ABI. The callers are C: no
// exception (std::bad_alloc of a std::vector sized by the caller's data, for
// instance) may cross the boundary, so it becomes a message and a nonzero
// return code like every other failure.
template <typename Body>
static int pt_guarded(const char* function_name, Body body) {
	try {
		return body();
	}
	catch (const std::exception& e) {
		printf("%s() failed: %s.\n", function_name, e.what());
	}
	catch (...) {
		printf("%s() failed with an unknown exception.\n", function_name);
	}
	return 1;
}


// ---------------------------------------------------------------- BVH build

__device__ __forceinline__ float pt_wave_min(float v) {
	for (int offset = 32; offset; offset >>= 1)
		v = fminf(v, __shfl_xor(v, offset));
	return v;
}
__device__ __forceinline__ float pt_wave_max(float v) {
	for (int offset = 32; offset; offset >>= 1)
		v = fmaxf(v, __shfl_xor(v, offset));
	return v;
}

// A grid of fixed size strides over the triangles, so that the atomics on the
// six words of the scene bounds stay few (one set per wave of the grid, not
// per 64 triangles: same-address atomics are served one after the other)
__global__ void __launch_bounds__(PT_BLOCK) k_build_dequantize(PtBuild b) {
	float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (uint64_t i = blockIdx.x * PT_BLOCK + threadIdx.x; i < b.n; i += (uint64_t) gridDim.x * PT_BLOCK) {
		float tri_lo[3], tri_hi[3];
		pt_build_dequantize(b, (uint32_t) i, tri_lo, tri_hi);
		for (int j = 0; j != 3; ++j) {
			lo[j] = fminf(lo[j], tri_lo[j]);
			hi[j] = fmaxf(hi[j], tri_hi[j]);
		}
	}
	for (int j = 0; j != 3; ++j) {
		float l = pt_wave_min(lo[j]), h = pt_wave_max(hi[j]);
		if ((threadIdx.x & 63) == 0) {
			atomicMin(&b.bounds[j], pt_float_to_key(l));
			atomicMax(&b.bounds[3 + j], pt_float_to_key(h));
		}
	}
}

__global__ void __launch_bounds__(PT_BLOCK) k_build_morton(PtBuild b) {
	uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i >= b.n)
		return;
	PtSceneBounds sb = pt_build_get_bounds(b.bounds);
	b.keys[i] = pt_build_morton(b, sb, i);
	b.order[i] = i;
}

__global__ void __launch_bounds__(PT_BLOCK) k_build_radix_tree(PtBuild b) {
	uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i + 1 < b.n)
		pt_build_radix_tree(b, i);
}

// Bottom-up passes over the binary tree (boxes, treelets, costs of the wide
// collapse) run in rounds: a round works on the queue of inner nodes whose
// children are both done, one node per thread, and fills the queue of the
// next round with the parents that this makes ready (the second thread to
// arrive at a parent enqueues it). Threads of a wave thus work on different
// nodes of about the same height at the same time. Climbing from every leaf
// in one launch instead (the second thread to arrive goes on to the parent)
// leaves one busy lane per wave after a few levels and needs a device-scope
// release and acquire per arrival, which on this part write back and
// invalidate the L2 of the XCD: that took six times as long.
enum { PT_PASS_BOXES = 1, PT_PASS_TREELETS = 2, PT_PASS_WIDE_COSTS = 4 };

// Appends value to a queue for the lanes that want it, one atomic per wave.
// All lanes of the wave must call this.
__device__ __forceinline__ void pt_queue_push(uint32_t* queue, uint32_t* count, bool want, uint32_t value) {
	unsigned long long mask = __ballot(want);
	if (mask == 0)
		return;
	int leader = __builtin_ctzll(mask);
	uint32_t base = 0;
	if ((int) (threadIdx.x & 63) == leader)
		base = atomicAdd(count, (uint32_t) __popcll(mask));
	base = (uint32_t) __shfl((int) base, leader);
	if (want)
		queue[base + __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u))] = value;
}

// Arrival at the parent of a finished node: true for the second of its children
__device__ __forceinline__ bool pt_build_arrive_second(const PtBuild& b, uint32_t parent) {
	return parent != 0xffffffffu && atomicAdd(&b.flags[parent], 1u) == 1u;
}

// Round 0: the leaves. The flags must be zero.
template <int PASS>
__global__ void __launch_bounds__(PT_BLOCK) k_build_round_leaves(PtBuild b, uint32_t* queue_out, uint32_t* count_out) {
	uint32_t k = blockIdx.x * PT_BLOCK + threadIdx.x;
	bool push = false;
	uint32_t parent = 0xffffffffu;
	if (k < b.n) {
		if (PASS & PT_PASS_BOXES) {
			PtSceneBounds sb = pt_build_get_bounds(b.bounds);
			pt_build_leaf_box(b, sb.pad, k);
		}
		// A single triangle has no inner node above it
		if (b.n > 1) {
			parent = b.parent[b.n - 1 + k];
			push = pt_build_arrive_second(b, parent);
		}
	}
	pt_queue_push(queue_out, count_out, push, parent);
}

// Later rounds: the inner nodes of queue_in (count_in is the counter that the
// previous round has left behind)
template <int PASS>
__global__ void __launch_bounds__(PT_BLOCK) k_build_round(PtBuild b, const uint32_t* queue_in, const uint32_t* count_in, uint32_t* queue_out, uint32_t* count_out) {
	const uint32_t count = *count_in;
	const uint32_t stride = gridDim.x * PT_BLOCK;
	PtTreeletScratch scratch;
	// Whole waves leave the loop together, so that every lane takes part in the pushes
	for (uint32_t first = blockIdx.x * PT_BLOCK + (threadIdx.x & ~63u); first < count; first += stride) {
		uint32_t i = first + (threadIdx.x & 63u);
		bool push = false;
		uint32_t parent = 0xffffffffu;
		if (i < count) {
			uint32_t node = queue_in[i];
			if (PASS & PT_PASS_BOXES)
				pt_build_inner_box(b, node);
			if (PASS & PT_PASS_TREELETS)
				pt_build_treelet(b, node, scratch);
			if (PASS & PT_PASS_WIDE_COSTS)
				pt_build_wide_cost(b, node);
			parent = b.parent[node];
			push = pt_build_arrive_second(b, parent);
		}
		pt_queue_push(queue_out, count_out, push, parent);
	}
}

__global__ void __launch_bounds__(PT_BLOCK) k_build_collapse(PtBuild b, uint32_t level_begin, uint32_t level_end) {
	uint32_t i = level_begin + blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i >= level_end)
		return;
	PtSceneBounds sb = pt_build_get_bounds(b.bounds);
	pt_build_collapse(b, sb.pad, i,
		[&](uint32_t count) { return atomicAdd(&b.counters[0], count); },
		[&](uint32_t count) { return atomicAdd(&b.counters[1], count); });
}


// ---------------------------------------------------------------- wavefront

struct PtTraceArgs {
	PtSceneView scene;
	PtWavefront wf;
	const PtConstants* constants;
	u32x2* stack_overflow;
	uint32_t stack_overflow_stride, stack_capacity;
	uint32_t bounce;  // index into ext_count / shadow_count
	uint32_t parity;  // which ray queue holds this bounce
	uint32_t* fetch;  // 8 zeroed fetch counters for this launch
	PtCacheLayout cache_layout;
	unsigned long long* visit_counters; // nodes, triangles of extension rays [0], [1] and of shadow rays [2], [3], or NULL
	uint32_t* error_word; // PT_ERROR_* bits, checked by pt_synchronize()
	unsigned long long* emissive_counts; // shadow rays traced, shadow rays that ended at a hit that is not emissive, second walks (k_trace_shadow<..., true>)
	uint32_t aimed; // shadow rays start skipping only at hits in front of the light they were aimed at
};

// pathtrace.frag.glsl:365-386 for every path of the batch
__global__ void __launch_bounds__(PT_BLOCK) k_raygen(PtBatch batch, const PtConstants* constants, PtWavefront wf, uint32_t path_count, uint32_t ray_count) {
	uint32_t p = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (p >= path_count)
		return;
	uint32_t x, y, frame_slot;
	bool inside = pt_path_to_pixel(batch, p, &x, &y, &frame_slot);
	uint32_t sx = 0, sy = 0;
	v3 origin = mk3(0.0f, 0.0f, 0.0f), dir = mk3(0.0f, 0.0f, 1.0f);
	float cone_width = 0.0f, cone_spread = 0.0f;
	if (inside)
		pt_primary_ray(*constants, x, y, batch.first_frame_index + frame_slot * batch.frame_stride, sx, sy, &origin, &dir, &cone_width, &cone_spread);
	wf.ray_o[0][p] = mk4(origin.x, origin.y, origin.z, pt_as_float(inside ? p : PT_INVALID));
	wf.ray_d[0][p] = mk4(dir.x, dir.y, dir.z, pt_pack_cone(cone_width, cone_spread));
	// The path state (throughput weights of one, the seed) is not written here:
	// shading the first vertex derives it from the path index again, which is
	// cheaper than 32 bytes written and 32 read per path
	wf.radiance[p] = mk4(0.0f, 0.0f, 0.0f, 0.0f);
	// ray_count is path_count, or 0 when the path length is 0 and no ray is ever traced
	if (p == 0)
		wf.ext_count[0] = ray_count;
}

// trace_ray() of pathtrace.frag.glsl:181-199 up to the hit record.
// Persistent waves: the grid is sized to the chip, not to the queue.
// EMISSIVE_ONLY: for the rays to the last vertex of a path, of which shading only wants to know whether they hit
// something emissive (pathtrace.frag.glsl:323-325); the hit record then names a triangle that answers that like the
// closest one (see pt_trace_wave())
template <bool COUNT_VISITS, bool EMISSIVE_ONLY>
__global__ void __launch_bounds__(PT_TRACE_BLOCK, PT_WAVES_PER_BLOCK * PT_TRACE_BLOCKS_PER_CU / 4) k_trace_extension(PtTraceArgs a) {
	__shared__ PtTraceLds lds;
	pt_fill_cache(lds, a.cache_layout, a.scene.nodes, a.scene.tris);
	PtRayQueue queue;
	queue.origin = a.wf.ray_o[a.parity];
	queue.direction = a.wf.ray_d[a.parity];
	queue.count = &a.wf.ext_count[a.bounce];
	queue.fetch = a.fetch;
	u32x4* hits = a.wf.hit;
	pt_trace_wave<COUNT_VISITS, EMISSIVE_ONLY>(lds.wave[threadIdx.x >> 6], lds.cache, a.cache_layout, a.scene.nodes, a.scene.tris, a.scene.triangle_count, queue,
		a.stack_overflow + (blockIdx.x * PT_TRACE_BLOCK + threadIdx.x), a.stack_overflow_stride, a.stack_capacity, a.visit_counters, a.error_word,
		[&](uint32_t slot, float path_bits, float, const PtHit& hit) {
			// Triangle indices stay below 2^31, so the front flag rides in the top
			// bit (a miss is all ones); the path index comes along so that shading
			// need not read the ray origin
			u32x4 record;
			record.x = (hit.prim == PT_INVALID) ? PT_INVALID : (hit.prim | (hit.front ? 0x80000000u : 0u));
			record.y = pt_as_uint(hit.u);
			record.z = pt_as_uint(hit.v);
			record.w = pt_as_uint(path_bits);
			hits[slot] = record;
		}, a.emissive_counts);
}

// trace_ray_emission() of pathtrace.frag.glsl:203-219 and the MIS sum at :338
// EMISSIVE_ONLY: with the skipping of pt_trace_wave() (the nodes and triangles must carry the marks for it)
template <bool COUNT_VISITS, bool EMISSIVE_ONLY>
__global__ void __launch_bounds__(PT_TRACE_BLOCK, PT_WAVES_PER_BLOCK * PT_TRACE_BLOCKS_PER_CU / 4) k_trace_shadow(PtTraceArgs a) {
	__shared__ PtTraceLds lds;
	pt_fill_cache(lds, a.cache_layout, a.scene.nodes, a.scene.tris);
	PtRayQueue queue;
	queue.origin = a.wf.sh_o;
	queue.direction = a.wf.sh_d;
	queue.count = &a.wf.shadow_count[a.bounce];
	queue.fetch = a.fetch;
	pt_trace_wave<COUNT_VISITS, EMISSIVE_ONLY, true>(lds.wave[threadIdx.x >> 6], lds.cache, a.cache_layout, a.scene.nodes, a.scene.tris, a.scene.triangle_count, queue,
		a.stack_overflow + (blockIdx.x * PT_TRACE_BLOCK + threadIdx.x), a.stack_overflow_stride, a.stack_capacity, a.visit_counters ? a.visit_counters + 2 : NULL, a.error_word,
		[&](uint32_t slot, float path_bits, float, const PtHit& hit) {
			const PtConstants& c = *a.constants;
			v3 emission;
			if (hit.prim == PT_INVALID)
				emission = mk3(c.sky_radiance[0], c.sky_radiance[1], c.sky_radiance[2]);
			else if (a.scene.materials[hit.prim] == a.scene.emission_material_index)
				emission = mk3(c.emission_material_radiance[0], c.emission_material_radiance[1], c.emission_material_radiance[2]);
			else
				emission = mk3(0.0f, 0.0f, 0.0f);
			f32x4 w = pt_load_streaming(&a.wf.sh_w[slot]);
			v3 term = (mk3(w.x, w.y, w.z) * emission) * w.w;
			// Adding a zero leaves the radiance as it is (it is never -0), so the
			// read-modify-write happens only for terms that are not zero; a NaN
			// from an infinite weight compares unequal and is added as well
			if (term.x != 0.0f || term.y != 0.0f || term.z != 0.0f) {
				uint32_t path = pt_as_uint(path_bits);
				f32x4 radiance = a.wf.radiance[path];
				v3 sum = mk3(radiance.x, radiance.y, radiance.z) + term;
				a.wf.radiance[path] = mk4(sum.x, sum.y, sum.z, 0.0f);
			}
		}, a.emissive_counts, a.aimed != 0u);
}

// The marks that shadow rays read (pt_trace_wave<..., EMISSIVE_ONLY = true>): a
// triangle's first padding word says whether its material is the emission
// material ...
__global__ void __launch_bounds__(PT_BLOCK) k_mark_emissive_triangles(PtTri* tris, uint32_t triangle_count, const uint8_t* materials, uint32_t emission_material_index) {
	uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i < triangle_count)
		tris[i].pad[0] = (materials[tris[i].prim] == emission_material_index) ? 1u : 0u;
}

// ... and bit i of a node's first padding byte whether such a triangle lies
// below child i. One launch with inner_children == 0 looks at the triangles
// among the children only; every further launch carries the marks one level
// up, so as many of them as the tree has levels complete them. (A launch may
// see the mark of a child node from before or after that node's thread wrote
// it: marks only ever gain bits towards their final value.)
__global__ void __launch_bounds__(PT_BLOCK) k_mark_emissive_nodes(PtNode* nodes, uint32_t node_count, const PtTri* tris, int inner_children) {
	uint32_t n = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (n >= node_count)
		return;
	const uint32_t imask = nodes[n].imask, leafmask = nodes[n].leafmask, child_base = nodes[n].child_base, tri_base = nodes[n].tri_base;
	uint32_t marks = 0;
	for (uint32_t slot = 0; slot != 8; ++slot) {
		const uint32_t below = (1u << slot) - 1u;
		if (leafmask & (1u << slot))
			marks |= (tris[tri_base + (uint32_t) __builtin_popcount(leafmask & below)].pad[0] != 0u) ? (1u << slot) : 0u;
		else if (inner_children && (imask & (1u << slot)))
			marks |= (((const volatile uint8_t*) nodes[child_base + (uint32_t) __builtin_popcount(imask & below)].pad)[0] != 0) ? (1u << slot) : 0u;
	}
	nodes[n].pad[0] = (uint8_t) marks;
}

// Decodes one mip level (2^log_w texels wide) of a BC1 / BC5 texture into tiled 8-bit texels (pt_load_texel() reads them)
__global__ void __launch_bounds__(PT_BLOCK) k_decode_texture(const uint8_t* bc_mip, uint32_t* texels, uint32_t log_w, uint32_t format, uint32_t texel_count) {
	uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i >= texel_count)
		return;
	uint32_t x = i & ((1u << log_w) - 1u), y = i >> log_w;
	texels[pt_texel_index(log_w, x, y)] = pt_decode_texel(bc_mip, log_w, format, x, y);
}

// The three .vks arrays of a triangle side by side in one 64-byte record
__global__ void __launch_bounds__(PT_BLOCK) k_build_shade_tris(const u32x2* positions, const u32x2* normals_uvs, const uint8_t* materials, PtShadeTri* shade_tris, uint32_t triangle_count) {
	uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i >= triangle_count)
		return;
	PtShadeTri t;
	for (int v = 0; v != 3; ++v) {
		t.positions[v] = positions[3 * (uint64_t) i + v];
		t.normals_uvs[v] = normals_uvs[3 * (uint64_t) i + v];
	}
	t.material = materials[i];
	t.pad[0] = t.pad[1] = t.pad[2] = 0;
	shade_tris[i] = t;
}

// Additive blending into the RGBA32F target in frame order
// (src/main.c:1000-1021, 1884-1888)
__global__ void __launch_bounds__(PT_BLOCK) k_resolve(PtBatch batch, PtWavefront wf, f32x4* accum, int overwrite_first) {
	uint32_t r = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (r >= batch.paths_per_frame)
		return;
	uint32_t x, y, frame_slot;
	if (!pt_path_to_pixel(batch, r, &x, &y, &frame_slot))
		return;
	uint64_t pixel = (uint64_t) y * batch.width + x;
	f32x4 sum = accum[pixel];
	for (uint32_t f = 0; f != batch.frame_count; ++f) {
		f32x4 radiance = wf.radiance[(uint64_t) f * batch.paths_per_frame + r];
		if (f == 0 && overwrite_first)
			sum = mk4(radiance.x, radiance.y, radiance.z, 1.0f);
		else {
			sum.x += radiance.x;
			sum.y += radiance.y;
			sum.z += radiance.z;
		}
	}
	accum[pixel] = sum;
}

// Test hooks
__global__ void __launch_bounds__(PT_BLOCK) k_gather_primary_hits(PtBatch batch, PtWavefront wf, int32_t* triangles, float* barycentrics, uint8_t* front) {
	uint32_t p = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (p >= batch.paths_per_frame)
		return;
	uint32_t x, y, frame_slot;
	if (!pt_path_to_pixel(batch, p, &x, &y, &frame_slot))
		return;
	uint64_t pixel = (uint64_t) y * batch.width + x;
	u32x4 hit = wf.hit[p];
	triangles[pixel] = (hit.x == PT_INVALID) ? -1 : (int32_t) (hit.x & 0x7fffffffu);
	barycentrics[2 * pixel] = pt_as_float(hit.y);
	barycentrics[2 * pixel + 1] = pt_as_float(hit.z);
	front[pixel] = (uint8_t) ((hit.x != PT_INVALID) ? (hit.x >> 31) : 0u);
}

struct PtUserRayArgs {
	PtSceneView scene;
	const f32x4* origins;
	const f32x4* directions;
	const uint32_t* count;
	uint32_t* fetch;
	u32x2* stack_overflow;
	uint32_t stack_overflow_stride, stack_capacity;
	PtCacheLayout cache_layout;
	uint32_t* error_word;
	int32_t* triangles;
	float* t_u_v;
	uint8_t* front;
};

__global__ void __launch_bounds__(PT_TRACE_BLOCK, PT_WAVES_PER_BLOCK * PT_TRACE_BLOCKS_PER_CU / 4) k_trace_user_rays(PtUserRayArgs a) {
	__shared__ PtTraceLds lds;
	pt_fill_cache(lds, a.cache_layout, a.scene.nodes, a.scene.tris);
	PtRayQueue queue;
	queue.origin = a.origins;
	queue.direction = a.directions;
	queue.count = a.count;
	queue.fetch = a.fetch;
	pt_trace_wave<false, false>(lds.wave[threadIdx.x >> 6], lds.cache, a.cache_layout, a.scene.nodes, a.scene.tris, a.scene.triangle_count, queue,
		a.stack_overflow + (blockIdx.x * PT_TRACE_BLOCK + threadIdx.x), a.stack_overflow_stride, a.stack_capacity, NULL, a.error_word,
		[&](uint32_t slot, float, float, const PtHit& hit) {
			a.triangles[slot] = (int32_t) hit.prim;
			a.t_u_v[3 * slot] = hit.t;
			a.t_u_v[3 * slot + 1] = hit.u;
			a.t_u_v[3 * slot + 2] = hit.v;
			a.front[slot] = hit.front ? 1 : 0;
		});
}

// Scratch memory of pt_trace_rays_no_alloc(): a control block (the ray count
// and the eight fetch counters of the traversal, 4096 bytes) followed by the
// staged origins and directions as the traversal reads them (two f32x4 per
// ray). It is part of the code object, so no call has to allocate it; every
// GPU that the library runs on holds its own copy until the process ends.
// In host code this name only stands for the symbol: the array's address on a
// device comes from hipGetSymbolAddress() (pt_no_alloc_scratch_address()).
#define PT_NO_ALLOC_MAX_RAYS PT_TRACE_RAYS_NO_ALLOC_MAX_RAYS
#define PT_NO_ALLOC_CONTROL_BYTES 4096u
static_assert(sizeof(uint32_t) * (1 + 8 * PT_FETCH_STRIDE) <= PT_NO_ALLOC_CONTROL_BYTES, "control block too small");
static __device__ __align__(256) unsigned char pt_no_alloc_scratch[PT_NO_ALLOC_CONTROL_BYTES + 2ull * sizeof(f32x4) * PT_NO_ALLOC_MAX_RAYS];

// Brings caller-supplied rays (three floats per origin and direction) into the
// layout of a ray queue and publishes their number; control[1...] must have
// been zeroed before
__global__ void __launch_bounds__(PT_BLOCK) k_stage_user_rays(const float* origins, const float* directions, uint32_t ray_count, uint32_t* control, f32x4* staged_origins, f32x4* staged_directions) {
	uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i == 0)
		control[0] = ray_count;
	if (i >= ray_count)
		return;
	staged_origins[i] = mk4(origins[3 * (uint64_t) i], origins[3 * (uint64_t) i + 1], origins[3 * (uint64_t) i + 2], pt_as_float(i));
	staged_directions[i] = mk4(directions[3 * (uint64_t) i], directions[3 * (uint64_t) i + 1], directions[3 * (uint64_t) i + 2], 0.0f);
}

__global__ void __launch_bounds__(PT_BLOCK) k_shading_data(PtSceneView scene, const PtConstants* constants, const int32_t* triangles, const float* barycentrics, const uint8_t* front, const float* out_dirs, const float* footprints, uint32_t count, float* out) {
	pt_stage_srgb_table(scene.srgb_to_linear);
	uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i >= count)
		return;
	PtShadingData s = pt_shading_data(scene, *constants, (uint32_t) triangles[i], barycentrics[2 * i], barycentrics[2 * i + 1], front[i] != 0, mk3(out_dirs[3 * i], out_dirs[3 * i + 1], out_dirs[3 * i + 2]),
		mk3(0.0f, 0.0f, 0.0f), footprints ? footprints[i] : 0.0f, 0.0f);
	float* o = &out[20 * (uint64_t) i];
	o[0] = s.pos.x; o[1] = s.pos.y; o[2] = s.pos.z;
	o[3] = s.normal.x; o[4] = s.normal.y; o[5] = s.normal.z;
	o[6] = s.out_dir.x; o[7] = s.out_dir.y; o[8] = s.out_dir.z;
	o[9] = s.lambert_out;
	o[10] = s.emission.x; o[11] = s.emission.y; o[12] = s.emission.z;
	o[13] = s.diffuse_albedo.x; o[14] = s.diffuse_albedo.y; o[15] = s.diffuse_albedo.z;
	o[16] = s.fresnel_0.x; o[17] = s.fresnel_0.y; o[18] = s.fresnel_0.z;
	o[19] = s.roughness;
}

__global__ void k_random_numbers(uint32_t x, uint32_t y, uint32_t frame_index, uint32_t draw_count, float* out) {
	if (blockIdx.x != 0 || threadIdx.x != 0)
		return;
	uint32_t sx = x ^ (frame_index << 16), sy = y ^ ((frame_index + 237u) << 16);
	for (uint32_t i = 0; i != draw_count; ++i) {
		v2 r = pt_random(sx, sy);
		out[2 * i] = r.x;
		out[2 * i + 1] = r.y;
	}
}


// ---------------------------------------------------------------- host side

enum { PT_K_RAYGEN = 0, PT_K_TRACE_EXT = 1, PT_K_TRACE_SHADOW = 2, PT_K_SHADE = 3, PT_K_RESOLVE = 4, PT_K_COUNT = 5 };

struct pt_scene {
	int device;
	hipStream_t stream;
	// scene data
	uint32_t triangle_count, material_count, texture_count;
	u32x2* positions;     // only until the BVH and the shading records are built
	u32x2* normals_uvs;   // likewise
	PtShadeTri* shade_tris;
	uint8_t* materials;
	u32x2* texture_table;
	uint8_t* texture_blob;
	float* srgb_to_linear;
	f32x4* lights;
	f32x4* light_nodes;
	f32x4* tree_lights;
	uint32_t light_leaf_count;
	// The grid over the lights (PtSceneView::light_grid_*), or light_grid_cells NULL
	uint32_t* light_grid_cells;
	f32x4* light_grid_lights;
	uint32_t* light_grid_leaves;
	float light_grid_lo[3], light_grid_cell[3], light_grid_inv_cell[3], light_grid_reach;
	uint32_t light_grid_dim[3];
	int light_sampling;
	int shading_math;
	uint32_t light_count;
	PtNode* nodes;
	PtTri* tris;
	pt_bvh_info_t bvh_info;
	// settings
	int strategy;
	uint32_t path_length, emission_material_index;
	bool emissive_marks_valid; // nodes and triangles carry the marks for emission_marks_index
	// Whether shadow rays skip what cannot change their answer: 0 not decided
	// yet (they do, and count how many of them end blocked; the counts of the
	// first batch decide), 1 yes, 2 no. emissive_skipping_fixed: set by
	// PT_SHADOW_SKIPPING, never decided anew
	int emissive_skipping;
	bool emissive_skipping_fixed;
	bool emissive_skipping_last_vertex; // the rays to the last vertex of a path skip likewise (off with PT_SHADOW_SKIPPING=0)
	uint32_t emissive_marks_index;
	// What the decision was taken from (pt_statistics_t reports it), and the
	// camera of the last pt_render_samples() call: another view has another
	// share of blocked shadow rays, so the decision is taken anew for it
	unsigned long long skipping_decided_on_rays, skipping_decided_on_blocked;
	// Whether skipping shadow rays wait for a hit in front of the light they were aimed at (pt_trace_wave<..., AIMED>):
	// 0 undecided (they do not, and their second walks are counted), 1 yes, 2 no; fixed by PT_SHADOW_AIMED=0/1
	int shadow_aimed;
	bool shadow_aimed_fixed;
	float skipping_camera[20];
	// render targets and wavefront buffers
	uint32_t width, height;
	PtBatch frame_layout; // tiles and paths per frame
	uint32_t max_frames_per_batch;
	uint32_t capacity; // paths per batch
	f32x4* accum;
	uint32_t sample_count;
	PtWavefront wf;
	uint32_t* counters; // ext_count[], shadow_count[] at PT_SHADOW_COUNT_OFFSET, then the fetch counters
	PtConstants* constants;
	u32x2* stack_overflow;
	u32x2* stack_overflow_shadow; // the shadow traversal may run beside the extension traversal
	uint32_t stack_capacity, stack_overflow_stride;
	uint32_t trace_grid; // blocks of the persistent traversal kernels
	// Shadow rays of vertex k and extension rays of vertex k + 1 are independent:
	// the former run on a stream of their own, joined before the next shading
	hipStream_t shadow_stream;
	hipEvent_t shade_done, shadow_done;
	unsigned long long* visit_counters;
	uint32_t* error_word; // device word of PT_ERROR_* bits
	unsigned char* no_alloc_scratch; // this device's copy of pt_no_alloc_scratch
	// pinned host staging for counter readback
	uint32_t* host_counters;
	// instrumentation
	int timing_enabled, visit_counters_enabled;
	pt_statistics_t stats;
	struct timed_launch { int kind; hipEvent_t begin, end; };
	std::vector<timed_launch> pending_events;
	std::vector<hipEvent_t> event_pool;
	// queue size snapshots of finished batches and the number of padding
	// paths (outside the image) that each of them carried in its first queue
	struct counter_read { uint32_t* counters; uint32_t padding_paths; uint32_t emissive_marks_serial; };
	uint32_t emissive_marks_serial; // counts the times the marks were made: the counts of a batch traced with older marks decide nothing
	std::vector<counter_read> pending_counter_reads;
	std::vector<uint32_t*> snapshot_pool; // pinned buffers of PT_COUNTER_WORDS words, reused
};

// ext_count[k] and shadow_count[k] per bounce, then 8 fetch counters for each of the possible trace launches of a batch
// The two families sit on different 128-byte lines: both are targets of atomics in one launch
#define PT_SHADOW_COUNT_OFFSET 32
#define PT_COUNTER_WORDS (2 * PT_SHADOW_COUNT_OFFSET)
static_assert(PT_MAX_PATH_LENGTH + 1 <= PT_SHADOW_COUNT_OFFSET, "queue counters overlap");
#define PT_FETCH_WORDS (8 * PT_FETCH_STRIDE * PT_COUNTER_WORDS)

static uint32_t pt_grid(uint64_t count) { return (uint32_t) ((count + PT_BLOCK - 1) / PT_BLOCK); }
// Grid for kernels that use pt_block_first_slot(): a multiple of 8 blocks
static uint32_t pt_grid_chunked(uint64_t capacity, uint32_t block_size = PT_BLOCK) {
	uint32_t blocks = (uint32_t) ((capacity + block_size - 1) / block_size);
	uint32_t chunk = (blocks + 7) / 8;
	return chunk ? 8 * chunk : 8;
}

// Workgroups for a traversal launch over at most ray_count rays. A launch
// ends when its last wave has finished its last ray, so a short queue (one
// 1080p frame is 2 M rays, four per lane of the full grid) is done sooner by
// one workgroup per CU than by two, and leaves room for a second traversal
// kernel beside it. Measured at 1080p: faster up to four frames per batch
// (8 M paths), slower from eight on.
static uint32_t pt_trace_grid(const pt_scene* s, uint64_t ray_count) {
	if (ray_count < PT_SHORT_BATCH_PATHS)
		return s->trace_grid / PT_TRACE_BLOCKS_PER_CU;
	return s->trace_grid;
}

static PtSceneView pt_scene_view(const pt_scene* s) {
	PtSceneView v;
	v.nodes = s->nodes;
	v.tris = s->tris;
	v.triangle_count = s->triangle_count;
	v.shade_tris = s->shade_tris;
	v.materials = s->materials;
	v.texture_table = s->texture_table;
	v.texture_blob = s->texture_blob;
	v.srgb_to_linear = s->srgb_to_linear;
	v.lights = s->lights;
	v.light_count = s->light_count;
	v.light_nodes = s->light_nodes;
	v.tree_lights = s->tree_lights;
	v.light_leaf_count = s->light_leaf_count;
	v.light_grid_cells = s->light_grid_cells;
	v.light_grid_lights = s->light_grid_lights;
	v.light_grid_leaves = s->light_grid_leaves;
	for (int a = 0; a != 3; ++a) {
		v.light_grid_lo[a] = s->light_grid_lo[a];
		v.light_grid_cell[a] = s->light_grid_cell[a];
		v.light_grid_inv_cell[a] = s->light_grid_inv_cell[a];
		v.light_grid_dim[a] = s->light_grid_dim[a];
	}
	v.light_grid_reach = s->light_grid_reach;
	v.emission_material_index = s->emission_material_index;
	return v;
}

static hipEvent_t pt_get_event(pt_scene* s) {
	if (!s->event_pool.empty()) {
		hipEvent_t e = s->event_pool.back();
		s->event_pool.pop_back();
		return e;
	}
	hipEvent_t e = NULL;
	if (hipEventCreate(&e) != hipSuccess)
		return NULL;
	return e;
}

struct pt_launch_timer {
	pt_scene* s;
	int kind;
	hipEvent_t begin, end;
	hipStream_t stream;
	pt_launch_timer(pt_scene* scene, int kind_, hipStream_t stream_ = NULL) : s(scene), kind(kind_), begin(NULL), end(NULL), stream(stream_ ? stream_ : scene->stream) {
		++s->stats.kernel_launches[kind];
		if (s->timing_enabled) {
			begin = pt_get_event(s);
			end = pt_get_event(s);
			if (begin && end)
				hipEventRecord(begin, stream);
		}
	}
	~pt_launch_timer() {
		if (begin && end) {
			hipEventRecord(end, stream);
			pt_scene::timed_launch t = { kind, begin, end };
			s->pending_events.push_back(t);
		}
	}
};

// Folds finished event pairs and counter snapshots into the statistics;
// requires the stream to be idle
static void pt_collect(pt_scene* s) {
	for (size_t i = 0; i != s->pending_events.size(); ++i) {
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, s->pending_events[i].begin, s->pending_events[i].end) == hipSuccess)
			s->stats.kernel_ms[s->pending_events[i].kind] += ms;
		s->event_pool.push_back(s->pending_events[i].begin);
		s->event_pool.push_back(s->pending_events[i].end);
	}
	s->pending_events.clear();
	for (size_t i = 0; i != s->pending_counter_reads.size(); ++i) {
		uint32_t* c = s->pending_counter_reads[i].counters;
		if (c[0] >= s->pending_counter_reads[i].padding_paths)
			s->stats.extension_rays -= s->pending_counter_reads[i].padding_paths;
		for (uint32_t k = 0; k <= PT_MAX_PATH_LENGTH; ++k) {
			s->stats.extension_rays += c[k];
			s->stats.shadow_rays += c[PT_SHADOW_COUNT_OFFSET + k];
		}
		// The first batch that says enough about the shadow rays of this scene
		// decides whether they go on skipping: yes if a quarter of them are blocked
		unsigned long long traced = 0, blocked = 0, second_walks = 0;
		memcpy(&traced, &c[PT_EMISSIVE_COUNTS_OFFSET], sizeof(traced));
		memcpy(&blocked, &c[PT_EMISSIVE_COUNTS_OFFSET + 2], sizeof(blocked));
		memcpy(&second_walks, &c[PT_EMISSIVE_COUNTS_OFFSET + 4], sizeof(second_walks));
		// The same batch says whether the lamps that rays are aimed at tend to turn up after something was skipped
		// (one ray in thirty-two walking twice): then a hit behind the lamp had better not start the skipping
		if (s->shadow_aimed == 0 && s->pending_counter_reads[i].emissive_marks_serial == s->emissive_marks_serial && traced >= 4096) {
			s->shadow_aimed = (32 * second_walks >= traced) ? 1 : 2;
			if (getenv("PT_TRACE_DIAGNOSTICS"))
				printf("trace diagnostics (shadow): %.1f%% of %llu rays walked twice; a hit behind the light a ray was aimed at %s\n",
					100.0 * (double) second_walks / (double) traced, traced, (s->shadow_aimed == 1) ? "no longer starts the skipping" : "goes on starting the skipping");
		}
		if (s->emissive_skipping == 0 && s->pending_counter_reads[i].emissive_marks_serial == s->emissive_marks_serial && traced >= 4096) {
			s->emissive_skipping = (4 * blocked >= traced) ? 1 : 2;
			s->skipping_decided_on_rays = traced;
			s->skipping_decided_on_blocked = blocked;
			if (getenv("PT_TRACE_DIAGNOSTICS"))
				printf("trace diagnostics (shadow): %.1f%% of %llu rays ended at a hit that is not emissive; skipping %s\n",
					100.0 * (double) blocked / (double) traced, traced, (s->emissive_skipping == 1) ? "stays on" : "is turned off");
		}
		s->snapshot_pool.push_back(c);
	}
	s->pending_counter_reads.clear();
}

template <typename T>
static int pt_alloc(T** out, uint64_t count) {
	*out = NULL;
	PT_CHECK(hipMalloc((void**) out, (count ? count : 1) * sizeof(T)));
	return 0;
}

static void pt_free_render_targets(pt_scene* s) {
	hipFree(s->accum); s->accum = NULL;
	for (int i = 0; i != 2; ++i) {
		hipFree(s->wf.ray_o[i]); s->wf.ray_o[i] = NULL;
		hipFree(s->wf.ray_d[i]); s->wf.ray_d[i] = NULL;
	}
	hipFree(s->wf.hit); s->wf.hit = NULL;
	hipFree(s->wf.sh_o); s->wf.sh_o = NULL;
	hipFree(s->wf.sh_d); s->wf.sh_d = NULL;
	hipFree(s->wf.sh_w); s->wf.sh_w = NULL;
	hipFree(s->wf.state0); s->wf.state0 = NULL;
	hipFree(s->wf.state1); s->wf.state1 = NULL;
	hipFree(s->wf.radiance); s->wf.radiance = NULL;
	s->capacity = 0;
	s->width = s->height = 0;
}

// Allocates wavefront buffers for at least path_count paths
static int pt_reserve_wavefront(pt_scene* s, uint32_t path_count) {
	if (path_count <= s->capacity)
		return 0;
	uint32_t w = s->width, h = s->height;
	f32x4* accum = s->accum;
	s->accum = NULL;
	pt_free_render_targets(s);
	s->accum = accum;
	s->width = w;
	s->height = h;
	for (int i = 0; i != 2; ++i) {
		if (pt_alloc(&s->wf.ray_o[i], path_count) || pt_alloc(&s->wf.ray_d[i], path_count))
			return 1;
	}
	if (pt_alloc(&s->wf.hit, path_count) || pt_alloc(&s->wf.sh_o, path_count) || pt_alloc(&s->wf.sh_d, path_count) || pt_alloc(&s->wf.sh_w, path_count)
		|| pt_alloc(&s->wf.state0, path_count) || pt_alloc(&s->wf.state1, path_count) || pt_alloc(&s->wf.radiance, path_count))
		return 1;
	s->capacity = path_count;
	return 0;
}

// One bottom-up pass over the binary tree in rounds (see k_build_round).
// queues are two arrays of n node indices, counts two counters on the device.
template <int PASS>
static int pt_build_bottom_up(pt_scene* s, const PtBuild& b, uint32_t* queues[2], uint32_t* counts) {
	const uint32_t n = b.n;
	PT_CHECK(hipMemsetAsync(b.flags, 0, sizeof(uint32_t) * n, s->stream));
	PT_CHECK(hipMemsetAsync(counts, 0, 2 * sizeof(uint32_t), s->stream));
	hipLaunchKernelGGL(k_build_round_leaves<PASS>, dim3(pt_grid(n)), dim3(PT_BLOCK), 0, s->stream, b, queues[0], &counts[0]);
	int in = 0;
	uint32_t round = 1, remaining = n > 1 ? 1u : 0u;
	while (remaining != 0) {
		// No tree is deeper than its 63-bit keys and 32-bit indices allow
		if (round > 4096) {
			printf("A bottom-up pass over the BVH did not come to an end.\n");
			return 1;
		}
		// Sixteen rounds between looks at the queue; rounds on an empty queue do nothing
		for (int i = 0; i != 16; ++i, ++round) {
			PT_CHECK(hipMemsetAsync(&counts[in ^ 1], 0, sizeof(uint32_t), s->stream));
			// A node done in round r has at least r + 1 leaves below it, so at most
			// n / (r + 1) nodes queue up; the kernel strides over whatever there is
			uint32_t blocks = pt_grid(n / (round + 1) + 1);
			if (blocks > 16384)
				blocks = 16384;
			hipLaunchKernelGGL(k_build_round<PASS>, dim3(blocks), dim3(PT_BLOCK), 0, s->stream, b, (const uint32_t*) queues[in], (const uint32_t*) &counts[in], queues[in ^ 1], &counts[in ^ 1]);
			in ^= 1;
		}
		PT_CHECK(hipMemcpyAsync(&remaining, &counts[in], sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
		PT_CHECK(hipStreamSynchronize(s->stream));
	}
	PT_CHECK(hipGetLastError());
	return 0;
}

static int pt_build_bvh(pt_scene* s, const pt_scene_header_t* header) {
	uint32_t n = s->triangle_count;
	hipEvent_t begin = NULL, end = NULL;
	if (hipEventCreate(&begin) != hipSuccess || hipEventCreate(&end) != hipSuccess || hipEventRecord(begin, s->stream) != hipSuccess) {
		printf("Failed to create the events that time the BVH build.\n");
		if (begin) hipEventDestroy(begin);
		if (end) hipEventDestroy(end);
		return 1;
	}
	PtBuild b;
	memset(&b, 0, sizeof(b));
	b.n = n;
	b.positions = s->positions;
	memcpy(b.factor, header->dequantization_factor, sizeof(b.factor));
	memcpy(b.summand, header->dequantization_summand, sizeof(b.summand));
	uint64_t* keys_sorted = NULL;
	uint32_t* order_sorted = NULL;
	uint32_t* round_counts = NULL;
	void* sort_temp = NULL;
	int result = 1;
	// The wide BVH has at most one node per two leaf groups; n nodes is a safe bound
	uint32_t max_nodes = n ? n : 1;
	do {
		if (pt_alloc(&s->nodes, max_nodes) || pt_alloc(&s->tris, n))
			break;
		if (n == 0) {
			PtNode empty;
			memset(&empty, 0, sizeof(empty));
			if (hipMemcpyAsync(s->nodes, &empty, sizeof(empty), hipMemcpyHostToDevice, s->stream) != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess)
				break;
			s->bvh_info.node_count = 1;
			s->bvh_info.depth = 1;
			result = 0;
			break;
		}
		if (pt_alloc(&b.verts, 9 * (uint64_t) n) || pt_alloc(&b.bounds, 6) || pt_alloc(&b.keys, n) || pt_alloc(&b.order, n)
			|| pt_alloc(&keys_sorted, n) || pt_alloc(&order_sorted, n)
			|| pt_alloc(&b.left, n) || pt_alloc(&b.right, n) || pt_alloc(&b.parent, 2 * (uint64_t) n)
			|| pt_alloc(&b.size, n) || pt_alloc(&b.sah_cost, n) || pt_alloc(&b.wide_cost, 8 * (uint64_t) n) || pt_alloc(&b.wide_choice, 8 * (uint64_t) n) || pt_alloc(&b.box, 12 * (uint64_t) n)
			|| pt_alloc(&b.flags, n) || pt_alloc(&b.task_binary, max_nodes) || pt_alloc(&b.counters, 2) || pt_alloc(&round_counts, 2))
			break;
		b.nodes = s->nodes;
		b.tris = s->tris;
		uint32_t bounds_init[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
		uint32_t counters_init[2] = { 1u, 0u };
		uint32_t root_task = 0;
		if (hipMemcpyAsync(b.bounds, bounds_init, sizeof(bounds_init), hipMemcpyHostToDevice, s->stream) != hipSuccess
			|| hipMemcpyAsync(b.counters, counters_init, sizeof(counters_init), hipMemcpyHostToDevice, s->stream) != hipSuccess
			|| hipMemcpyAsync(b.task_binary, &root_task, sizeof(root_task), hipMemcpyHostToDevice, s->stream) != hipSuccess
			|| hipMemsetAsync(b.flags, 0, sizeof(uint32_t) * n, s->stream) != hipSuccess)
			break;
		hipLaunchKernelGGL(k_build_dequantize, dim3(pt_grid(n) < 4096 ? pt_grid(n) : 4096), dim3(PT_BLOCK), 0, s->stream, b);
		hipLaunchKernelGGL(k_build_morton, dim3(pt_grid(n)), dim3(PT_BLOCK), 0, s->stream, b);
		size_t temp_bytes = 0;
		if (rocprim::radix_sort_pairs(NULL, temp_bytes, b.keys, keys_sorted, b.order, order_sorted, n, 0, 63, s->stream) != hipSuccess)
			break;
		if (hipMalloc(&sort_temp, temp_bytes ? temp_bytes : 1) != hipSuccess)
			break;
		if (rocprim::radix_sort_pairs(sort_temp, temp_bytes, b.keys, keys_sorted, b.order, order_sorted, n, 0, 63, s->stream) != hipSuccess)
			break;
		{ uint64_t* t = b.keys; b.keys = keys_sorted; keys_sorted = t; }
		{ uint32_t* t = b.order; b.order = order_sorted; order_sorted = t; }
		if (n > 1)
			hipLaunchKernelGGL(k_build_radix_tree, dim3(pt_grid(n - 1)), dim3(PT_BLOCK), 0, s->stream, b);
		// Bottom-up passes in rounds (the old key array serves as the two queues):
		// boxes with the first treelet pass, further treelet passes, then the
		// costs of the wide collapse
		uint32_t* queues[2] = { (uint32_t*) keys_sorted, (uint32_t*) keys_sorted + n };
		bool passes_failed = false;
		for (int pass = 0; pass <= PT_TREELET_PASSES && !passes_failed; ++pass) {
			if (PT_TREELET_PASSES == 0)
				passes_failed = pt_build_bottom_up<PT_PASS_BOXES | PT_PASS_WIDE_COSTS>(s, b, queues, round_counts) != 0;
			else if (pass == 0)
				passes_failed = pt_build_bottom_up<PT_PASS_BOXES | PT_PASS_TREELETS>(s, b, queues, round_counts) != 0;
			else if (pass < PT_TREELET_PASSES)
				passes_failed = pt_build_bottom_up<PT_PASS_TREELETS>(s, b, queues, round_counts) != 0;
			else
				passes_failed = pt_build_bottom_up<PT_PASS_WIDE_COSTS>(s, b, queues, round_counts) != 0;
		}
		if (passes_failed)
			break;
		// Collapse level by level: the nodes allocated while processing one
		// level form the next level
		uint32_t level_begin = 0, level_end = 1, depth = 0;
		bool failed = false;
		while (level_begin < level_end) {
			hipLaunchKernelGGL(k_build_collapse, dim3(pt_grid(level_end - level_begin)), dim3(PT_BLOCK), 0, s->stream, b, level_begin, level_end);
			uint32_t counters[2];
			if (hipMemcpyAsync(counters, b.counters, sizeof(counters), hipMemcpyDeviceToHost, s->stream) != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) {
				failed = true;
				break;
			}
			if (counters[0] > max_nodes || counters[1] > n) {
				printf("BVH collapse overran its buffers (%u nodes, %u triangles).\n", counters[0], counters[1]);
				failed = true;
				break;
			}
			level_begin = level_end;
			level_end = counters[0];
			++depth;
			s->bvh_info.triangle_count = counters[1];
		}
		if (failed || hipGetLastError() != hipSuccess)
			break;
		s->bvh_info.node_count = level_end;
		s->bvh_info.depth = depth;
		s->emissive_marks_valid = false;
		result = 0;
	} while (0);
	hipEventRecord(end, s->stream);
	hipStreamSynchronize(s->stream);
	hipEventElapsedTime(&s->bvh_info.build_ms, begin, end);
	hipEventDestroy(begin);
	hipEventDestroy(end);
	hipFree(b.verts); hipFree(b.bounds); hipFree(b.keys); hipFree(b.order); hipFree(keys_sorted); hipFree(order_sorted);
	hipFree(b.left); hipFree(b.right); hipFree(b.parent); hipFree(b.size); hipFree(b.sah_cost); hipFree(b.wide_cost); hipFree(b.wide_choice); hipFree(b.box);
	hipFree(b.flags); hipFree(b.task_binary); hipFree(b.counters); hipFree(round_counts); hipFree(sort_temp);
	s->bvh_info.node_bytes = sizeof(PtNode);
	s->bvh_info.triangle_bytes = sizeof(PtTri);
	if (result)
		printf("Failed to build the BVH for %u triangles on the GPU.\n", n);
	return result;
}

// The address of pt_no_alloc_scratch on each device, looked up once per
// device (by pt_create_scene(), with that device current)
#define PT_MAX_DEVICES 64
static void* pt_no_alloc_scratch_addresses[PT_MAX_DEVICES];

static int pt_no_alloc_scratch_address(int device_index, void** out_address) {
	*out_address = NULL;
	if (device_index < 0 || device_index >= PT_MAX_DEVICES) {
		printf("GPU index %d is beyond the %d devices that the library keeps scratch addresses for.\n", device_index, PT_MAX_DEVICES);
		return 1;
	}
	if (!pt_no_alloc_scratch_addresses[device_index]) {
		void* address = NULL;
		PT_CHECK(hipGetSymbolAddress(&address, HIP_SYMBOL(pt_no_alloc_scratch)));
		pt_no_alloc_scratch_addresses[device_index] = address;
	}
	*out_address = pt_no_alloc_scratch_addresses[device_index];
	return 0;
}

static int pt_create_scene_impl(pt_scene_t** out_scene, int device_index, const pt_scene_header_t* header,
	const uint32_t* positions, const uint16_t* normals_and_tex_coords, const uint8_t* material_indices,
	const pt_texture_desc_t* textures, uint32_t texture_count)
{
	if (!out_scene)
		return 1;
	*out_scene = NULL;
	if (!header || (header->triangle_count && (!positions || !normals_and_tex_coords || !material_indices))) {
		printf("pt_create_scene() needs a header and all three mesh arrays.\n");
		return 1;
	}
	if (header->triangle_count > 0x55555555ull) {
		printf("pt_create_scene() supports at most %u triangles, got %llu.\n", 0x55555555u, (unsigned long long) header->triangle_count);
		return 1;
	}
	if (texture_count != 3 * header->material_count || (texture_count && !textures)) {
		printf("pt_create_scene() needs three textures per material (%llu materials, %u textures).\n", (unsigned long long) header->material_count, texture_count);
		return 1;
	}
	for (uint32_t i = 0; i != texture_count; ++i) {
		const pt_texture_desc_t* t = &textures[i];
		bool pow2 = t->width >= 4 && t->height >= 4 && !(t->width & (t->width - 1)) && !(t->height & (t->height - 1));
		if ((t->format != 131 && t->format != 132 && t->format != 141) || !pow2 || t->mip_count < 1 || t->mip_count > PT_MAX_MIPS || !t->mip_data || !t->mip_sizes) {
			printf("Texture %u is not supported (format %u, %ux%u, %u mipmaps).\n", i, t->format, t->width, t->height, t->mip_count);
			return 1;
		}
		// Block compression goes down to 4x4 texels (tools/texture_conversion/main.c:291-293),
		// so the smaller extent allows log2(extent) - 1 levels; level m has the extent >> m
		for (uint32_t m = 0; m != t->mip_count; ++m) {
			uint32_t mip_w = t->width >> m, mip_h = t->height >> m;
			if (mip_w < 4 || mip_h < 4) {
				printf("Texture %u (%ux%u) has %u mipmaps, but block-compressed mipmaps end at an extent of 4.\n", i, t->width, t->height, t->mip_count);
				return 1;
			}
			uint64_t expected = (uint64_t) (mip_w / 4) * (mip_h / 4) * ((t->format == 141) ? 16 : 8);
			if (t->mip_sizes[m] != expected || !t->mip_data[m]) {
				printf("The data block for mipmap %u of texture %u was supposed to have %llu bytes but had %llu bytes.\n", m, i, (unsigned long long) expected, (unsigned long long) t->mip_sizes[m]);
				return 1;
			}
		}
	}
	// A material index beyond the table would send the shading kernels past the
	// end of the texture table: such a file is refused here, not on the GPU
	for (uint64_t i = 0; i != header->triangle_count; ++i) {
		if (material_indices[i] >= header->material_count) {
			printf("Triangle %llu uses material %u but the scene has %llu materials.\n", (unsigned long long) i, (uint32_t) material_indices[i], (unsigned long long) header->material_count);
			return 1;
		}
	}
	if (pt_select_device(device_index) != hipSuccess) {
		printf("Failed to select GPU %d.\n", device_index);
		return 1;
	}
	// Set-up for pt_trace_rays_no_alloc(): where this device keeps the scratch array
	void* no_alloc_scratch = NULL;
	if (pt_no_alloc_scratch_address(device_index, &no_alloc_scratch))
		return 1;
	pt_scene* s = new pt_scene();
	s->no_alloc_scratch = (unsigned char*) no_alloc_scratch;
	// From here on the caller's pointer owns the scene: whoever sees this
	// function fail or throw destroys it
	*out_scene = s;
	s->device = device_index;
	s->stream = NULL;
	s->triangle_count = (uint32_t) header->triangle_count;
	s->material_count = (uint32_t) header->material_count;
	s->texture_count = texture_count;
	s->positions = NULL; s->normals_uvs = NULL; s->shade_tris = NULL; s->materials = NULL; s->texture_table = NULL; s->texture_blob = NULL;
	s->srgb_to_linear = NULL; s->lights = NULL; s->light_count = 0; s->light_nodes = NULL; s->tree_lights = NULL; s->light_leaf_count = 0; s->light_grid_cells = NULL; s->light_grid_lights = NULL; s->light_grid_leaves = NULL; s->light_sampling = PT_LIGHT_SAMPLING_EXACT; s->nodes = NULL; s->tris = NULL;
	memset(&s->bvh_info, 0, sizeof(s->bvh_info));
	s->strategy = PT_STRATEGY_NEE;
	// PT_SHADING_MATH=fast makes pt_set_shading_math(scene, PT_SHADING_MATH_FAST) the default of new scenes
	const char* shading_math = getenv("PT_SHADING_MATH");
	s->shading_math = (shading_math && strcmp(shading_math, "fast") == 0) ? PT_SHADING_MATH_FAST : PT_SHADING_MATH_EXACT;
	s->path_length = 4;
	s->emission_material_index = 0;
	s->emissive_marks_valid = false;
	s->emissive_marks_index = 0;
	s->emissive_marks_serial = 0;
	s->skipping_decided_on_rays = s->skipping_decided_on_blocked = 0;
	memset(s->skipping_camera, 0, sizeof(s->skipping_camera));
	// PT_SHADOW_SKIPPING=0 or 1 fixes what is otherwise decided by the rays
	// traced so far (a knob for measurements; the images do not depend on it)
	const char* skipping = getenv("PT_SHADOW_SKIPPING");
	s->emissive_skipping_fixed = skipping && (skipping[0] == '0' || skipping[0] == '1');
	s->emissive_skipping = s->emissive_skipping_fixed ? ((skipping[0] == '1') ? 1 : 2) : 0;
	s->emissive_skipping_last_vertex = !(s->emissive_skipping_fixed && skipping[0] == '0');
	const char* aimed = getenv("PT_SHADOW_AIMED");
	s->shadow_aimed_fixed = aimed && (aimed[0] == '0' || aimed[0] == '1');
	s->shadow_aimed = s->shadow_aimed_fixed ? ((aimed[0] == '1') ? 1 : 2) : 0;
	s->width = s->height = 0;
	memset(&s->frame_layout, 0, sizeof(s->frame_layout));
	s->max_frames_per_batch = 0;
	s->capacity = 0;
	s->accum = NULL;
	s->sample_count = 0;
	memset(&s->wf, 0, sizeof(s->wf));
	s->counters = NULL; s->constants = NULL; s->stack_overflow = NULL; s->stack_overflow_shadow = NULL; s->shadow_stream = NULL; s->shade_done = NULL; s->shadow_done = NULL; s->visit_counters = NULL; s->error_word = NULL; s->host_counters = NULL;
	s->stack_capacity = s->stack_overflow_stride = s->trace_grid = 0;
	s->timing_enabled = s->visit_counters_enabled = 0;
	memset(&s->stats, 0, sizeof(s->stats));
	int failed = 1;
	do {
		if (hipStreamCreate(&s->stream) != hipSuccess || hipStreamCreate(&s->shadow_stream) != hipSuccess
			|| hipEventCreateWithFlags(&s->shade_done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->shadow_done, hipEventDisableTiming) != hipSuccess)
			break;
		uint64_t n = s->triangle_count;
		if (pt_alloc(&s->positions, 3 * n) || pt_alloc(&s->normals_uvs, 3 * n) || pt_alloc(&s->shade_tris, n) || pt_alloc(&s->materials, n)
			|| pt_alloc(&s->counters, PT_COUNTER_WORDS + PT_FETCH_WORDS) || pt_alloc(&s->constants, 1) || pt_alloc(&s->visit_counters, PT_VISIT_COUNTERS)
			|| pt_alloc(&s->srgb_to_linear, 256) || pt_alloc(&s->lights, 1) || pt_alloc(&s->error_word, 1))
			break;
		if (hipMemsetAsync(s->error_word, 0, sizeof(uint32_t), s->stream) != hipSuccess)
			break;
		if (n) {
			if (hipMemcpyAsync(s->positions, positions, n * 24, hipMemcpyHostToDevice, s->stream) != hipSuccess
				|| hipMemcpyAsync(s->normals_uvs, normals_and_tex_coords, n * 24, hipMemcpyHostToDevice, s->stream) != hipSuccess
				|| hipMemcpyAsync(s->materials, material_indices, n, hipMemcpyHostToDevice, s->stream) != hipSuccess)
				break;
		}
		if (hipMemsetAsync(s->visit_counters, 0, PT_VISIT_COUNTERS * sizeof(unsigned long long), s->stream) != hipSuccess)
			break;
		// sRGB decoding table for VK_FORMAT_BC1_RGB_SRGB_BLOCK textures
		float table[256];
		for (int i = 0; i != 256; ++i) {
			double c = i / 255.0;
			table[i] = (float) ((c <= 0.04045) ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
		}
		if (hipMemcpyAsync(s->srgb_to_linear, table, sizeof(table), hipMemcpyHostToDevice, s->stream) != hipSuccess)
			break;
		// The texels of all mip levels of all textures decoded into one blob
		// (pt_load_texel()): the block-compressed payloads go to the GPU as they
		// are, a kernel decodes them
		std::vector<PtTexture> descs(texture_count ? texture_count : 1);
		std::vector<uint64_t> bc_offsets;
		uint64_t texel_bytes = 0, bc_bytes = 0;
		for (uint32_t i = 0; i != texture_count; ++i) {
			descs[i].width = textures[i].width;
			descs[i].height = textures[i].height;
			descs[i].format = textures[i].format;
			descs[i].mip_count = textures[i].mip_count;
			for (uint32_t m = 0; m != textures[i].mip_count; ++m) {
				descs[i].mip_offsets[m] = texel_bytes;
				texel_bytes += (uint64_t) (textures[i].width >> m) * (textures[i].height >> m) * 4;
				bc_offsets.push_back(bc_bytes);
				bc_bytes += (textures[i].mip_sizes[m] + 15) & ~15ull;
			}
			// One 64-byte line per 4x4 tile: every texture starts on one
			texel_bytes = (texel_bytes + 63) & ~63ull;
		}
		std::vector<uint8_t> bc_blob(bc_bytes ? bc_bytes : 16);
		for (uint32_t i = 0, level = 0; i != texture_count; ++i)
			for (uint32_t m = 0; m != textures[i].mip_count; ++m, ++level)
				memcpy(&bc_blob[bc_offsets[level]], textures[i].mip_data[m], textures[i].mip_sizes[m]);
		std::vector<u32x2> packed(descs.size());
		for (size_t i = 0; i != descs.size(); ++i)
			packed[i] = (i < texture_count) ? pt_pack_texture(descs[i]) : u32x2{ 0u, 0u };
		if (texel_bytes >> 36) {
			printf("The textures take %llu bytes decoded, more than the 64 GiB that a scene may hold.\n", (unsigned long long) texel_bytes);
			break;
		}
		uint8_t* device_bc_blob = NULL;
		bool textures_failed = pt_alloc(&s->texture_blob, texel_bytes ? texel_bytes : 16) || pt_alloc(&device_bc_blob, bc_blob.size()) || pt_alloc(&s->texture_table, packed.size())
			|| hipMemcpyAsync(device_bc_blob, bc_blob.data(), bc_blob.size(), hipMemcpyHostToDevice, s->stream) != hipSuccess
			|| hipMemcpyAsync(s->texture_table, packed.data(), packed.size() * sizeof(u32x2), hipMemcpyHostToDevice, s->stream) != hipSuccess;
		for (uint32_t i = 0, level = 0; i != texture_count && !textures_failed; ++i) {
			uint32_t log_w = packed[i].y & 0xffu, log_h = (packed[i].y >> 8) & 0xffu;
			for (uint32_t m = 0; m != textures[i].mip_count; ++m, ++level) {
				uint32_t texels = (textures[i].width >> m) * (textures[i].height >> m);
				// The levels follow each other as pt_mip_texel_offset() says
				if (descs[i].mip_offsets[m] != descs[i].mip_offsets[0] + 4 * pt_mip_texel_offset(log_w, log_h, m)) {
					printf("Internal error: mip level %u of texture %u is misplaced.\n", m, i);
					textures_failed = true;
					break;
				}
				hipLaunchKernelGGL(k_decode_texture, dim3(pt_grid(texels)), dim3(PT_BLOCK), 0, s->stream,
					device_bc_blob + bc_offsets[level], (uint32_t*) (s->texture_blob + descs[i].mip_offsets[m]), log_w - m, textures[i].format, texels);
			}
		}
		textures_failed = textures_failed || hipStreamSynchronize(s->stream) != hipSuccess || hipGetLastError() != hipSuccess;
		hipFree(device_bc_blob);
		if (textures_failed)
			break;
		if (pt_build_bvh(s, header))
			break;
		// Regroup the vertex arrays per triangle for shading; the arrays
		// themselves are not needed any more
		if (n)
			hipLaunchKernelGGL(k_build_shade_tris, dim3(pt_grid((uint32_t) n)), dim3(PT_BLOCK), 0, s->stream, s->positions, s->normals_uvs, s->materials, s->shade_tris, (uint32_t) n);
		if (hipStreamSynchronize(s->stream) != hipSuccess || hipGetLastError() != hipSuccess)
			break;
		hipFree(s->positions); s->positions = NULL;
		hipFree(s->normals_uvs); s->normals_uvs = NULL;
		// The traversal kernels are persistent: one workgroup of sixteen waves per CU
		int cu_count = 0;
		if (hipDeviceGetAttribute(&cu_count, hipDeviceAttributeMultiprocessorCount, device_index) != hipSuccess || cu_count <= 0)
			break;
		s->trace_grid = (((uint32_t) cu_count + 7) & ~7u) * PT_TRACE_BLOCKS_PER_CU;
		s->stack_capacity = s->bvh_info.depth + 2;
		if (s->stack_capacity < PT_WAVE_STACK + 1)
			s->stack_capacity = PT_WAVE_STACK + 1;
		s->stack_overflow_stride = s->trace_grid * PT_TRACE_BLOCK;
		if (pt_alloc(&s->stack_overflow, (uint64_t) s->stack_overflow_stride * (s->stack_capacity - PT_WAVE_STACK))
			|| pt_alloc(&s->stack_overflow_shadow, (uint64_t) s->stack_overflow_stride * (s->stack_capacity - PT_WAVE_STACK)))
			break;
		failed = 0;
	} while (0);
	if (failed) {
		printf("Failed to create the scene on GPU %d.\n", device_index);
		pt_destroy_scene(s);
		*out_scene = NULL;
		return 1;
	}
	return 0;
}

extern "C" void pt_destroy_scene(pt_scene_t* s) {
	if (!s)
		return;
	hipSetDevice(s->device);
	if (s->shadow_stream)
		hipStreamSynchronize(s->shadow_stream);
	if (s->stream)
		hipStreamSynchronize(s->stream);
	pt_guarded("pt_destroy_scene", [&]() { pt_collect(s); return 0; });
	for (size_t i = 0; i != s->pending_events.size(); ++i) {
		hipEventDestroy(s->pending_events[i].begin);
		hipEventDestroy(s->pending_events[i].end);
	}
	for (size_t i = 0; i != s->pending_counter_reads.size(); ++i)
		hipHostFree(s->pending_counter_reads[i].counters);
	for (size_t i = 0; i != s->event_pool.size(); ++i)
		hipEventDestroy(s->event_pool[i]);
	for (size_t i = 0; i != s->snapshot_pool.size(); ++i)
		hipHostFree(s->snapshot_pool[i]);
	pt_free_render_targets(s);
	hipFree(s->positions); hipFree(s->normals_uvs); hipFree(s->shade_tris); hipFree(s->materials); hipFree(s->texture_table); hipFree(s->texture_blob);
	hipFree(s->srgb_to_linear); hipFree(s->lights); hipFree(s->light_nodes); hipFree(s->tree_lights); hipFree(s->light_grid_cells); hipFree(s->light_grid_lights); hipFree(s->light_grid_leaves); hipFree(s->nodes); hipFree(s->tris);
	hipFree(s->counters); hipFree(s->constants); hipFree(s->visit_counters); hipFree(s->error_word); hipFree(s->stack_overflow); hipFree(s->stack_overflow_shadow);
	if (s->shade_done) hipEventDestroy(s->shade_done);
	if (s->shadow_done) hipEventDestroy(s->shadow_done);
	if (s->shadow_stream)
		hipStreamDestroy(s->shadow_stream);
	if (s->stream)
		hipStreamDestroy(s->stream);
	delete s;
}

static int pt_set_lights_impl(pt_scene_t* s, const float* lights, uint32_t count) {
	if (!s || (count && !lights))
		return 1;
	PT_CHECK(pt_select_device(s->device));
	PT_CHECK(hipStreamSynchronize(s->stream));
	// The light tree is built on the host: it is small (64 bytes per light)
	// and changes only here
	PtLightTree tree;
	pt_build_light_tree(tree, lights, count);
	PtLightGrid grid;
#ifndef PT_LIGHTS_PER_CELL
// Lights per cell of the grid over the lights, on average. Shading 10 000
// lights takes 14.7 / 14.5 / 14.8 ms with 0.25 / 0.5 / 1 (smaller cells: more
// steps per ray, fewer lights to test per step).
#define PT_LIGHTS_PER_CELL 0.5f
#endif
	pt_build_light_grid(grid, tree, PT_LIGHTS_PER_CELL);
	f32x4* fresh = NULL;
	f32x4* fresh_nodes = NULL;
	f32x4* fresh_tree_lights = NULL;
	uint32_t* fresh_cells = NULL;
	f32x4* fresh_grid_lights = NULL;
	uint32_t* fresh_grid_leaves = NULL;
	if (pt_alloc(&fresh, count) || pt_alloc(&fresh_nodes, tree.nodes.size()) || pt_alloc(&fresh_tree_lights, tree.lights.size())
		|| (count && hipMemcpy(fresh, lights, sizeof(float) * 4 * count, hipMemcpyHostToDevice) != hipSuccess)
		|| hipMemcpy(fresh_nodes, tree.nodes.data(), sizeof(f32x4) * tree.nodes.size(), hipMemcpyHostToDevice) != hipSuccess
		|| hipMemcpy(fresh_tree_lights, tree.lights.data(), sizeof(f32x4) * tree.lights.size(), hipMemcpyHostToDevice) != hipSuccess
		|| (grid.valid && (pt_alloc(&fresh_cells, grid.cells.size()) || pt_alloc(&fresh_grid_lights, grid.lights.size()) || pt_alloc(&fresh_grid_leaves, grid.leaves.size())
			|| hipMemcpy(fresh_cells, grid.cells.data(), sizeof(uint32_t) * grid.cells.size(), hipMemcpyHostToDevice) != hipSuccess
			|| hipMemcpy(fresh_grid_lights, grid.lights.data(), sizeof(f32x4) * grid.lights.size(), hipMemcpyHostToDevice) != hipSuccess
			|| hipMemcpy(fresh_grid_leaves, grid.leaves.data(), sizeof(uint32_t) * grid.leaves.size(), hipMemcpyHostToDevice) != hipSuccess)))
	{
		printf("Failed to upload %u lights.\n", count);
		hipFree(fresh); hipFree(fresh_nodes); hipFree(fresh_tree_lights); hipFree(fresh_cells); hipFree(fresh_grid_lights); hipFree(fresh_grid_leaves);
		return 1;
	}
	s->emissive_marks_valid = false; // other lights, other shadow rays: the counts behind the skipping start anew
	hipFree(s->lights); hipFree(s->light_nodes); hipFree(s->tree_lights); hipFree(s->light_grid_cells); hipFree(s->light_grid_lights); hipFree(s->light_grid_leaves);
	s->light_grid_cells = fresh_cells;
	s->light_grid_lights = fresh_grid_lights;
	s->light_grid_leaves = fresh_grid_leaves;
	for (int a = 0; a != 3; ++a) {
		s->light_grid_lo[a] = grid.lo[a];
		s->light_grid_cell[a] = grid.cell[a];
		s->light_grid_inv_cell[a] = grid.inv_cell[a];
		s->light_grid_dim[a] = grid.dim[a];
	}
	s->light_grid_reach = grid.reach;
	s->lights = fresh;
	s->light_count = count;
	s->light_nodes = fresh_nodes;
	s->tree_lights = fresh_tree_lights;
	s->light_leaf_count = tree.leaf_count;
	return 0;
}

extern "C" int pt_set_light_sampling(pt_scene_t* s, int light_sampling) {
	if (!s)
		return 1;
	if (light_sampling != PT_LIGHT_SAMPLING_EXACT && light_sampling != PT_LIGHT_SAMPLING_TREE) {
		printf("Unsupported light sampling mode %d.\n", light_sampling);
		return 1;
	}
	s->light_sampling = light_sampling;
	return 0;
}

extern "C" int pt_set_shading_math(pt_scene_t* s, int shading_math) {
	if (!s)
		return 1;
	if (shading_math != PT_SHADING_MATH_EXACT && shading_math != PT_SHADING_MATH_FAST) {
		printf("Unsupported shading math mode %d.\n", shading_math);
		return 1;
	}
	s->shading_math = shading_math;
	return 0;
}

extern "C" int pt_set_render_settings(pt_scene_t* s, int sampling_strategy, uint32_t path_length, uint32_t emission_material_index) {
	if (!s)
		return 1;
	if (sampling_strategy < 0 || sampling_strategy > 3 || path_length > PT_MAX_PATH_LENGTH) {
		printf("Unsupported render settings: sampling strategy %d, path length %u.\n", sampling_strategy, path_length);
		return 1;
	}
	s->strategy = sampling_strategy;
	s->path_length = path_length;
	s->emission_material_index = emission_material_index;
	return 0;
}

static int pt_create_render_targets_impl(pt_scene_t* s, uint32_t width, uint32_t height) {
	if (!s)
		return 1;
	if (width == 0 || height == 0 || (uint64_t) width * height > (1ull << 28)) {
		printf("Unsupported render target extent %ux%u.\n", width, height);
		return 1;
	}
	PT_CHECK(pt_select_device(s->device));
	PT_CHECK(hipStreamSynchronize(s->stream));
	pt_collect(s);
	pt_free_render_targets(s);
	s->width = width;
	s->height = height;
	PtBatch& l = s->frame_layout;
	l.width = width;
	l.height = height;
	l.tiles_x = (width + PT_TILE - 1) / PT_TILE;
	l.tiles_y = (height + PT_TILE - 1) / PT_TILE;
	l.paths_per_frame = l.tiles_x * l.tiles_y * 64;
	// A traversal launch keeps 256 K lanes busy until the last one has finished
	// its last ray; the fewer rays a lane sees per launch, the more of the launch
	// is that tail. 64 M paths (11 GB of queues) give every lane hundreds of rays.
	uint64_t target_paths = PT_BATCH_TARGET_PATHS;
	if (const char* override = getenv("PT_BATCH_PATHS"))
		target_paths = strtoull(override, NULL, 10);
	// Path and slot indices are 32-bit: a batch stays below 2^31 paths whatever the override says
	uint64_t frames = (target_paths + l.paths_per_frame - 1) / l.paths_per_frame;
	const uint64_t frame_limit = (1ull << 31) / l.paths_per_frame;
	if (frames > frame_limit) frames = frame_limit;
	if (frames > 64) frames = 64;
	if (frames < 1) frames = 1;
	s->max_frames_per_batch = (uint32_t) frames;
	if (pt_alloc(&s->accum, (uint64_t) width * height)) {
		pt_free_render_targets(s);
		return 1;
	}
	PT_CHECK(hipMemsetAsync(s->accum, 0, sizeof(f32x4) * (uint64_t) width * height, s->stream));
	s->sample_count = 0;
	return 0;
}

static PtTraceArgs pt_trace_args(pt_scene* s, uint32_t bounce, uint32_t parity, bool shadow) {
	PtTraceArgs a;
	a.scene = pt_scene_view(s);
	a.wf = s->wf;
	a.constants = s->constants;
	a.stack_overflow = shadow ? s->stack_overflow_shadow : s->stack_overflow;
	a.stack_overflow_stride = s->stack_overflow_stride;
	a.stack_capacity = s->stack_capacity;
	a.bounce = bounce;
	a.parity = parity;
	a.cache_layout = pt_cache_layout(s->bvh_info.node_count, s->bvh_info.triangle_count);
	a.fetch = s->counters + PT_COUNTER_WORDS + 8 * PT_FETCH_STRIDE * (shadow ? PT_SHADOW_COUNT_OFFSET + bounce : bounce);
	a.visit_counters = s->visit_counters_enabled ? s->visit_counters : NULL;
	a.error_word = s->error_word;
	a.emissive_counts = (unsigned long long*) (s->counters + PT_EMISSIVE_COUNTS_OFFSET);
	a.aimed = (s->shadow_aimed == 1) ? 1u : 0u;
	return a;
}

static void pt_launch_trace_extension(pt_scene* s, uint32_t bounce, uint32_t parity, uint32_t capacity) {
	pt_launch_timer timer(s, PT_K_TRACE_EXT);
	PtTraceArgs a = pt_trace_args(s, bounce, parity, false);
	// The rays to the last vertex only ask whether they hit something emissive.
	// Most of them hit something that is not, so the skipping pays for them
	// (city: that launch -20 %) unless the whole tree sits in LDS, where a
	// node costs too little (Cornell box: +2 %); their counts go to words of
	// their own and decide nothing
	const bool emissive_only = bounce + 1 == s->path_length && s->emissive_marks_valid && s->emissive_skipping_last_vertex
		&& a.cache_layout.node_count < s->bvh_info.node_count;
	a.emissive_counts = (unsigned long long*) (s->counters + PT_EMISSIVE_COUNTS_OFFSET + 6);
	if (s->visit_counters_enabled && emissive_only)
		hipLaunchKernelGGL((k_trace_extension<true, true>), dim3(pt_trace_grid(s, capacity)), dim3(PT_TRACE_BLOCK), 0, s->stream, a);
	else if (s->visit_counters_enabled)
		hipLaunchKernelGGL((k_trace_extension<true, false>), dim3(pt_trace_grid(s, capacity)), dim3(PT_TRACE_BLOCK), 0, s->stream, a);
	else if (emissive_only)
		hipLaunchKernelGGL((k_trace_extension<false, true>), dim3(pt_trace_grid(s, capacity)), dim3(PT_TRACE_BLOCK), 0, s->stream, a);
	else
		hipLaunchKernelGGL((k_trace_extension<false, false>), dim3(pt_trace_grid(s, capacity)), dim3(PT_TRACE_BLOCK), 0, s->stream, a);
}

// Renders one batch of frame_count frames
static int pt_render_batch(pt_scene* s, uint32_t first_frame_index, uint32_t frame_count, uint32_t frame_stride, int overwrite_first) {
	PtBatch batch = s->frame_layout;
	batch.frame_count = frame_count;
	batch.first_frame_index = first_frame_index;
	batch.frame_stride = frame_stride;
	uint32_t path_count = batch.paths_per_frame * frame_count;
	s->wf.ext_count = s->counters;
	s->wf.shadow_count = s->counters + PT_SHADOW_COUNT_OFFSET;
	PT_CHECK(hipMemsetAsync(s->counters, 0, sizeof(uint32_t) * (PT_COUNTER_WORDS + PT_FETCH_WORDS), s->stream));
	{
		pt_launch_timer timer(s, PT_K_RAYGEN);
		hipLaunchKernelGGL(k_raygen, dim3(pt_grid(path_count)), dim3(PT_BLOCK), 0, s->stream, batch, s->constants, s->wf, path_count, s->path_length ? path_count : 0u);
	}
	if (s->path_length >= 1 && s->bvh_info.triangle_count != 0
		&& (!s->emissive_marks_valid || s->emissive_marks_index != s->emission_material_index)) {
		// The shadow rays' marks for the emission material of these settings
		hipLaunchKernelGGL(k_mark_emissive_triangles, dim3(pt_grid(s->bvh_info.triangle_count)), dim3(PT_BLOCK), 0, s->stream,
			s->tris, s->bvh_info.triangle_count, s->materials, s->emission_material_index);
		for (uint32_t level = 0; level <= s->bvh_info.depth; ++level)
			hipLaunchKernelGGL(k_mark_emissive_nodes, dim3(pt_grid(s->bvh_info.node_count)), dim3(PT_BLOCK), 0, s->stream,
				s->nodes, s->bvh_info.node_count, s->tris, level != 0);
		if (!s->emissive_skipping_fixed)
			s->emissive_skipping = 0;
		if (!s->shadow_aimed_fixed)
			s->shadow_aimed = 0;
		s->emissive_marks_valid = true;
		s->emissive_marks_index = s->emission_material_index;
		++s->emissive_marks_serial;
	}
	uint32_t parity = 0;
	// Short queues get one traversal workgroup per CU, so the shadow rays of a
	// vertex and the extension rays of the next fit on the chip side by side;
	// long queues fill it alone and stay on one stream
	const bool overlap = pt_trace_grid(s, path_count) < s->trace_grid;
	hipStream_t shadow_stream = overlap ? s->shadow_stream : s->stream;
	bool shadow_pending = false;
	for (uint32_t k = 1; k <= s->path_length; ++k) {
		pt_launch_trace_extension(s, k - 1, parity, path_count);
		// Shading vertex k adds to the radiance after the light sample of vertex k - 1
		if (shadow_pending) {
			PT_CHECK(hipStreamWaitEvent(s->stream, s->shadow_done, 0));
			shadow_pending = false;
		}
		{
			pt_launch_timer timer(s, PT_K_SHADE);
			PtShadeArgs a;
			a.scene = pt_scene_view(s);
			a.wf = s->wf;
			a.batch = batch;
			a.constants = s->constants;
			a.strategy = s->strategy;
			a.path_length = s->path_length;
			a.k = k;
			a.parity = parity;
			a.visit_counters = s->visit_counters_enabled ? s->visit_counters : NULL;
			const bool tree = s->light_sampling == PT_LIGHT_SAMPLING_TREE && s->light_nodes;
			a.light_cache_capacity = (s->strategy == PT_STRATEGY_NEE && !tree) ? (s->light_count < PT_LIGHT_CACHE_SIZE ? s->light_count : PT_LIGHT_CACHE_SIZE) : 0;
			const size_t lds = sizeof(float) * PT_SHADE_BLOCK * a.light_cache_capacity;
			switch (a.strategy) {
			case PT_STRATEGY_SPHERICAL: hipLaunchKernelGGL(k_shade<PT_STRATEGY_SPHERICAL>, dim3(pt_grid_chunked(path_count, PT_SHADE_BLOCK)), dim3(PT_SHADE_BLOCK), lds, s->stream, a); break;
			case PT_STRATEGY_PSA: hipLaunchKernelGGL(k_shade<PT_STRATEGY_PSA>, dim3(pt_grid_chunked(path_count, PT_SHADE_BLOCK)), dim3(PT_SHADE_BLOCK), lds, s->stream, a); break;
			case PT_STRATEGY_BRDF: hipLaunchKernelGGL(k_shade<PT_STRATEGY_BRDF>, dim3(pt_grid_chunked(path_count, PT_SHADE_BLOCK)), dim3(PT_SHADE_BLOCK), lds, s->stream, a); break;
			default:
				if (s->shading_math == PT_SHADING_MATH_FAST)
					pt_launch_shade_fast(pt_grid_chunked(path_count, PT_SHADE_BLOCK), lds, s->stream, a, tree);
				else if (tree)
					hipLaunchKernelGGL(k_shade<PT_STRATEGY_NEE_TREE>, dim3(pt_grid_chunked(path_count, PT_SHADE_BLOCK)), dim3(PT_SHADE_BLOCK), lds, s->stream, a);
				else
					hipLaunchKernelGGL(k_shade<PT_STRATEGY_NEE>, dim3(pt_grid_chunked(path_count, PT_SHADE_BLOCK)), dim3(PT_SHADE_BLOCK), lds, s->stream, a);
				break;
			}
		}
		if (s->strategy == PT_STRATEGY_NEE && k < s->path_length) {
			if (overlap) {
				PT_CHECK(hipEventRecord(s->shade_done, s->stream));
				PT_CHECK(hipStreamWaitEvent(shadow_stream, s->shade_done, 0));
			}
			{
				pt_launch_timer timer(s, PT_K_TRACE_SHADOW, shadow_stream);
				PtTraceArgs a = pt_trace_args(s, k, 0, true);
				const bool skipping = s->emissive_skipping != 2 && s->emissive_marks_valid;
				if (s->visit_counters_enabled && skipping)
					hipLaunchKernelGGL((k_trace_shadow<true, true>), dim3(pt_trace_grid(s, path_count)), dim3(PT_TRACE_BLOCK), 0, shadow_stream, a);
				else if (s->visit_counters_enabled)
					hipLaunchKernelGGL((k_trace_shadow<true, false>), dim3(pt_trace_grid(s, path_count)), dim3(PT_TRACE_BLOCK), 0, shadow_stream, a);
				else if (skipping)
					hipLaunchKernelGGL((k_trace_shadow<false, true>), dim3(pt_trace_grid(s, path_count)), dim3(PT_TRACE_BLOCK), 0, shadow_stream, a);
				else
					hipLaunchKernelGGL((k_trace_shadow<false, false>), dim3(pt_trace_grid(s, path_count)), dim3(PT_TRACE_BLOCK), 0, shadow_stream, a);
			}
			if (overlap) {
				PT_CHECK(hipEventRecord(s->shadow_done, shadow_stream));
				shadow_pending = true;
			}
		}
		parity ^= 1;
	}
	if (shadow_pending)
		PT_CHECK(hipStreamWaitEvent(s->stream, s->shadow_done, 0));
	{
		pt_launch_timer timer(s, PT_K_RESOLVE);
		hipLaunchKernelGGL(k_resolve, dim3(pt_grid(batch.paths_per_frame)), dim3(PT_BLOCK), 0, s->stream, batch, s->wf, s->accum, overwrite_first);
	}
	// Snapshot the queue sizes of this batch for the ray statistics
	uint32_t* snapshot = NULL;
	if (!s->snapshot_pool.empty()) {
		snapshot = s->snapshot_pool.back();
		s->snapshot_pool.pop_back();
	}
	else
		PT_CHECK(hipHostMalloc((void**) &snapshot, sizeof(uint32_t) * PT_COUNTER_WORDS, hipHostMallocDefault));
	PT_CHECK(hipMemcpyAsync(snapshot, s->counters, sizeof(uint32_t) * PT_COUNTER_WORDS, hipMemcpyDeviceToHost, s->stream));
	pt_scene::counter_read read = { snapshot, path_count - s->width * s->height * frame_count, s->emissive_marks_serial };
	s->pending_counter_reads.push_back(read);
	s->stats.paths += (uint64_t) s->width * s->height * frame_count;
	PT_CHECK(hipGetLastError());
	return 0;
}

static int pt_render_samples_impl(pt_scene_t* s, const pt_constants_t* constants,
	uint32_t first_frame_index, uint32_t sample_count, uint32_t frame_stride, int accumulate)
{
	if (!s || !constants)
		return 1;
	if (!s->accum) {
		printf("pt_render_samples() needs render targets; call pt_create_render_targets() first.\n");
		return 1;
	}
	if ((uint32_t) constants->viewport_size[0] != s->width || (uint32_t) constants->viewport_size[1] != s->height) {
		printf("The viewport size %gx%g in the constants does not match the %ux%u render targets.\n", constants->viewport_size[0], constants->viewport_size[1], s->width, s->height);
		return 1;
	}
	if (sample_count == 0)
		return 0;
	PT_CHECK(pt_select_device(s->device));
	uint32_t frames_per_batch = s->max_frames_per_batch < sample_count ? s->max_frames_per_batch : sample_count;
	if (pt_reserve_wavefront(s, s->frame_layout.paths_per_frame * frames_per_batch))
		return 1;
	PT_CHECK(hipMemcpyAsync(s->constants, constants, sizeof(PtConstants), hipMemcpyHostToDevice, s->stream));
	// A target that holds no frame has nothing to add to: its first frame takes
	// the discard pipeline whatever the caller says, as the reference does
	// whenever accum_frame_count is zero (src/main.c:1884-1887), so alpha is 1
	if (s->sample_count == 0)
		accumulate = 0;
	// Whether shadow rays skip was decided for the view it was measured on
	float camera[20];
	memcpy(camera, constants->world_to_projection_space, sizeof(float) * 16);
	memcpy(camera + 16, constants->camera_pos, sizeof(float) * 3);
	camera[19] = (float) constants->camera_type;
	if (memcmp(camera, s->skipping_camera, sizeof(camera)) != 0) {
		memcpy(s->skipping_camera, camera, sizeof(camera));
		if ((!s->emissive_skipping_fixed && s->emissive_skipping != 0) || (!s->shadow_aimed_fixed && s->shadow_aimed != 0)) {
			if (!s->emissive_skipping_fixed)
				s->emissive_skipping = 0;
			if (!s->shadow_aimed_fixed)
				s->shadow_aimed = 0;
			++s->emissive_marks_serial; // batches queued for the old view decide nothing
		}
	}
	if (!accumulate)
		s->sample_count = 0;
	uint32_t done = 0;
	while (done < sample_count) {
		uint32_t frames = sample_count - done;
		if (frames > frames_per_batch)
			frames = frames_per_batch;
		int overwrite_first = (!accumulate && done == 0) ? 1 : 0;
		// Keep the backlog of events and counter snapshots bounded, however
		// many batches one call renders
		if (s->pending_counter_reads.size() > 256 || s->pending_events.size() > 4096) {
			PT_CHECK(hipStreamSynchronize(s->stream));
			pt_collect(s);
		}
		if (pt_render_batch(s, first_frame_index + done * frame_stride, frames, frame_stride, overwrite_first))
			return 1;
		done += frames;
	}
	s->sample_count += sample_count;
	return 0;
}

// Reads the error word that the traversal kernels set when they had to give
// up on rays; the stream must be idle. The word is cleared once reported.
static int pt_check_device_errors(pt_scene* s) {
	uint32_t errors = 0;
	PT_CHECK(hipMemcpy(&errors, s->error_word, sizeof(errors), hipMemcpyDeviceToHost));
	if (errors == 0)
		return 0;
	if (errors & PT_ERROR_STACK_OVERFLOW)
		printf("The BVH traversal ran out of stack (%u entries per ray): some rays did not get their closest hit.\n", s->stack_capacity);
	if (errors & PT_ERROR_PASS_LIMIT)
		printf("The BVH traversal stopped at its pass limit before all rays were done.\n");
	PT_CHECK(hipMemset(s->error_word, 0, sizeof(errors)));
	return 1;
}

static int pt_synchronize_impl(pt_scene_t* s) {
	if (!s)
		return 1;
	PT_CHECK(pt_select_device(s->device));
	PT_CHECK(hipStreamSynchronize(s->stream));
	pt_collect(s);
	return pt_check_device_errors(s);
}

static int pt_read_accumulation_impl(pt_scene_t* s, float* rgba, uint32_t* out_sample_count) {
	if (!s)
		return 1;
	if (pt_synchronize_impl(s))
		return 1;
	if (rgba) {
		if (!s->accum) {
			printf("There are no render targets to read.\n");
			return 1;
		}
		PT_CHECK(hipMemcpy(rgba, s->accum, sizeof(f32x4) * (uint64_t) s->width * s->height, hipMemcpyDeviceToHost));
	}
	if (out_sample_count)
		*out_sample_count = s->sample_count;
	return 0;
}

extern "C" void* pt_get_accumulation_device_pointer(pt_scene_t* s) { return s ? (void*) s->accum : NULL; }

extern "C" int pt_set_sample_count(pt_scene_t* s, uint32_t sample_count) {
	if (!s)
		return 1;
	s->sample_count = sample_count;
	return 0;
}

static int pt_set_instrumentation_impl(pt_scene_t* s, int enable_kernel_timing, int enable_visit_counters) {
	if (!s)
		return 1;
	if (pt_synchronize_impl(s))
		return 1;
	s->timing_enabled = enable_kernel_timing;
	s->visit_counters_enabled = enable_visit_counters;
	return 0;
}

static int pt_get_statistics_impl(pt_scene_t* s, pt_statistics_t* out) {
	if (!s || !out)
		return 1;
	if (pt_synchronize_impl(s))
		return 1;
	unsigned long long visits[PT_VISIT_COUNTERS];
	PT_CHECK(hipMemcpy(visits, s->visit_counters, sizeof(visits), hipMemcpyDeviceToHost));
	s->stats.nodes_visited[0] = visits[0];
	s->stats.triangles_tested[0] = visits[1];
	s->stats.nodes_visited[1] = visits[2];
	s->stats.triangles_tested[1] = visits[3];
	s->stats.shaded_hits = visits[4];
	if (getenv("PT_TRACE_DIAGNOSTICS"))
		for (int k = 0; k != 2; ++k) {
			printf("trace diagnostics (%s): %.1f%% of the node fetches and %.1f%% of the triangle fetches served by the BVH prefix in LDS\n", k ? "shadow" : "extension",
				100.0 * (double) visits[20 + 2 * k] / (double) (visits[2 * k] ? visits[2 * k] : 1), 100.0 * (double) visits[21 + 2 * k] / (double) (visits[1 + 2 * k] ? visits[1 + 2 * k] : 1));
			printf("trace diagnostics (%s): %.1f%% of the node steps found no child to enter and no triangle to test, %.1f%% would have without the closest hit found since the node was put on the stack\n", k ? "shadow" : "extension",
				100.0 * (double) visits[12 + 2 * k] / (double) (visits[2 * k] ? visits[2 * k] : 1), 100.0 * (double) visits[13 + 2 * k] / (double) (visits[2 * k] ? visits[2 * k] : 1));
			printf("trace diagnostics (%s): %.1f%% of the node steps belong to rays that only look for something emissive in front of their hit, %.1f%% to second walks\n", k ? "shadow" : "extension",
				100.0 * (double) visits[28 + 2 * k] / (double) (visits[2 * k] ? visits[2 * k] : 1), 100.0 * (double) visits[29 + 2 * k] / (double) (visits[2 * k] ? visits[2 * k] : 1));
			printf("trace diagnostics (%s): the lanes of a pass with a node step fetch %.1f different nodes on average (%.1f lanes take part)\n", k ? "shadow" : "extension",
				(double) visits[24 + 2 * k] / (double) (visits[16 + 2 * k] ? visits[16 + 2 * k] : 1), (double) visits[2 * k] / (double) (visits[16 + 2 * k] ? visits[16 + 2 * k] : 1));
			printf("trace diagnostics (%s): wave passes %llu, lanes holding a ray %.1f%%, refills %llu, passes with a node step %llu (%.1f%% of lanes in them), with a triangle test %llu (%.1f%%)\n",
				k ? "shadow" : "extension", visits[6 + 2 * k], 100.0 * (double) visits[7 + 2 * k] / (64.0 * (double) visits[6 + 2 * k]), visits[25 + 2 * k],
				visits[16 + 2 * k], 100.0 * (double) visits[2 * k] / (64.0 * (double) visits[16 + 2 * k]), visits[17 + 2 * k], 100.0 * (double) visits[1 + 2 * k] / (64.0 * (double) visits[17 + 2 * k]));
		}
	s->stats.shadow_skipping = (uint32_t) s->emissive_skipping;
	s->stats.shadow_skipping_fixed = s->emissive_skipping_fixed ? 1u : 0u;
	s->stats.shadow_rays_decided_on = s->skipping_decided_on_rays;
	s->stats.shadow_rays_blocked = s->skipping_decided_on_blocked;
	*out = s->stats;
	return 0;
}

static int pt_reset_statistics_impl(pt_scene_t* s) {
	if (!s)
		return 1;
	if (pt_synchronize_impl(s))
		return 1;
	memset(&s->stats, 0, sizeof(s->stats));
	PT_CHECK(hipMemset(s->visit_counters, 0, PT_VISIT_COUNTERS * sizeof(unsigned long long)));
	return 0;
}

extern "C" uint32_t pt_abi_version(void) { return PT_HIP_ABI_VERSION; }

extern "C" uint32_t pt_get_device_count(void) {
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count < 0) {
		(void) hipGetLastError();
		return 0;
	}
	return (uint32_t) count;
}

extern "C" int pt_get_bvh_info(pt_scene_t* s, pt_bvh_info_t* out) {
	if (!s || !out)
		return 1;
	*out = s->bvh_info;
	return 0;
}

static int pt_trace_primary_rays_impl(pt_scene_t* s, const pt_constants_t* constants, uint32_t frame_index,
	int32_t* triangle_indices, float* barycentrics, uint8_t* front)
{
	if (!s || !constants || !triangle_indices || !barycentrics || !front)
		return 1;
	if (!s->accum || (uint32_t) constants->viewport_size[0] != s->width || (uint32_t) constants->viewport_size[1] != s->height) {
		printf("pt_trace_primary_rays() needs render targets matching the viewport size in the constants.\n");
		return 1;
	}
	PT_CHECK(pt_select_device(s->device));
	if (pt_reserve_wavefront(s, s->frame_layout.paths_per_frame))
		return 1;
	PtBatch batch = s->frame_layout;
	batch.frame_count = 1;
	batch.first_frame_index = frame_index;
	batch.frame_stride = 1;
	uint32_t path_count = batch.paths_per_frame;
	uint64_t pixels = (uint64_t) s->width * s->height;
	int32_t* d_triangles = NULL;
	float* d_barycentrics = NULL;
	uint8_t* d_front = NULL;
	int result = 1;
	do {
		if (pt_alloc(&d_triangles, pixels) || pt_alloc(&d_barycentrics, 2 * pixels) || pt_alloc(&d_front, pixels))
			break;
		s->wf.ext_count = s->counters;
		s->wf.shadow_count = s->counters + PT_SHADOW_COUNT_OFFSET;
		if (hipMemcpyAsync(s->constants, constants, sizeof(PtConstants), hipMemcpyHostToDevice, s->stream) != hipSuccess
			|| hipMemsetAsync(s->counters, 0, sizeof(uint32_t) * (PT_COUNTER_WORDS + PT_FETCH_WORDS), s->stream) != hipSuccess)
			break;
		hipLaunchKernelGGL(k_raygen, dim3(pt_grid(path_count)), dim3(PT_BLOCK), 0, s->stream, batch, s->constants, s->wf, path_count, path_count);
		PtTraceArgs a = pt_trace_args(s, 0, 0, false);
		hipLaunchKernelGGL((k_trace_extension<false, false>), dim3(pt_trace_grid(s, path_count)), dim3(PT_TRACE_BLOCK), 0, s->stream, a);
		hipLaunchKernelGGL(k_gather_primary_hits, dim3(pt_grid(path_count)), dim3(PT_BLOCK), 0, s->stream, batch, s->wf, d_triangles, d_barycentrics, d_front);
		if (hipMemcpyAsync(triangle_indices, d_triangles, pixels * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream) != hipSuccess
			|| hipMemcpyAsync(barycentrics, d_barycentrics, 2 * pixels * sizeof(float), hipMemcpyDeviceToHost, s->stream) != hipSuccess
			|| hipMemcpyAsync(front, d_front, pixels, hipMemcpyDeviceToHost, s->stream) != hipSuccess
			|| hipStreamSynchronize(s->stream) != hipSuccess || hipGetLastError() != hipSuccess)
			break;
		result = 0;
	} while (0);
	hipFree(d_triangles); hipFree(d_barycentrics); hipFree(d_front);
	if (!result)
		result = pt_check_device_errors(s);
	if (result)
		printf("Failed to trace primary rays on the GPU.\n");
	return result;
}

static int pt_trace_rays_impl(pt_scene_t* s, const float* origins, const float* directions, uint32_t ray_count,
	int32_t* triangle_indices, float* t_u_v, uint8_t* front)
{
	if (!s || (ray_count && (!origins || !directions || !triangle_indices || !t_u_v || !front)))
		return 1;
	if (ray_count == 0)
		return 0;
	PT_CHECK(pt_select_device(s->device));
	PT_CHECK(hipStreamSynchronize(s->stream));
	f32x4* d_origins = NULL; f32x4* d_directions = NULL; int32_t* d_triangles = NULL; float* d_tuv = NULL; uint8_t* d_front = NULL;
	uint32_t* d_control = NULL; // ray count followed by 8 fetch counters
	std::vector<f32x4> staging(2 * (size_t) ray_count);
	for (uint32_t i = 0; i != ray_count; ++i) {
		staging[i] = mk4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], pt_as_float(i));
		staging[ray_count + i] = mk4(directions[3 * i], directions[3 * i + 1], directions[3 * i + 2], 0.0f);
	}
	int result = 1;
	do {
		if (pt_alloc(&d_origins, ray_count) || pt_alloc(&d_directions, ray_count) || pt_alloc(&d_triangles, ray_count)
			|| pt_alloc(&d_tuv, 3 * (uint64_t) ray_count) || pt_alloc(&d_front, ray_count) || pt_alloc(&d_control, 1 + 8 * PT_FETCH_STRIDE))
			break;
		uint32_t control[1 + 8 * PT_FETCH_STRIDE];
		memset(control, 0, sizeof(control));
		control[0] = ray_count;
		if (hipMemcpyAsync(d_origins, staging.data(), sizeof(f32x4) * ray_count, hipMemcpyHostToDevice, s->stream) != hipSuccess
			|| hipMemcpyAsync(d_directions, staging.data() + ray_count, sizeof(f32x4) * ray_count, hipMemcpyHostToDevice, s->stream) != hipSuccess
			|| hipMemcpyAsync(d_control, control, sizeof(control), hipMemcpyHostToDevice, s->stream) != hipSuccess)
			break;
		PtUserRayArgs a;
		a.scene = pt_scene_view(s);
		a.origins = d_origins;
		a.directions = d_directions;
		a.count = d_control;
		a.fetch = d_control + 1;
		a.stack_overflow = s->stack_overflow;
		a.stack_overflow_stride = s->stack_overflow_stride;
		a.stack_capacity = s->stack_capacity;
		a.cache_layout = pt_cache_layout(s->bvh_info.node_count, s->bvh_info.triangle_count);
		a.error_word = s->error_word;
		a.triangles = d_triangles;
		a.t_u_v = d_tuv;
		a.front = d_front;
		hipLaunchKernelGGL(k_trace_user_rays, dim3(pt_trace_grid(s, ray_count)), dim3(PT_TRACE_BLOCK), 0, s->stream, a);
		if (hipMemcpyAsync(triangle_indices, d_triangles, sizeof(int32_t) * ray_count, hipMemcpyDeviceToHost, s->stream) != hipSuccess
			|| hipMemcpyAsync(t_u_v, d_tuv, sizeof(float) * 3 * ray_count, hipMemcpyDeviceToHost, s->stream) != hipSuccess
			|| hipMemcpyAsync(front, d_front, ray_count, hipMemcpyDeviceToHost, s->stream) != hipSuccess
			|| hipStreamSynchronize(s->stream) != hipSuccess || hipGetLastError() != hipSuccess)
			break;
		result = 0;
	} while (0);
	hipFree(d_origins); hipFree(d_directions); hipFree(d_triangles); hipFree(d_tuv); hipFree(d_front); hipFree(d_control);
	if (!result)
		result = pt_check_device_errors(s);
	if (result)
		printf("Failed to trace %u rays on the GPU.\n", ray_count);
	return result;
}

// pt_trace_rays() for callers that cannot allocate: inputs and outputs are
// device memory of the caller, the ray queue and its counters live in the
// scratch array of the code object. No hipMalloc, hipFree, std::vector or
// other allocating routine is called on this path.
static int pt_trace_rays_no_alloc_impl(pt_scene_t* s, const float* device_origins, const float* device_directions, uint32_t ray_count,
	int32_t* device_triangle_indices, float* device_t_u_v, uint8_t* device_front)
{
	if (!s || (ray_count && (!device_origins || !device_directions || !device_triangle_indices || !device_t_u_v || !device_front)))
		return 1;
	if (ray_count > PT_NO_ALLOC_MAX_RAYS) {
		printf("pt_trace_rays_no_alloc() takes at most %u rays per call, got %u.\n", PT_NO_ALLOC_MAX_RAYS, ray_count);
		return 1;
	}
	if (ray_count == 0)
		return 0;
	PT_CHECK(pt_select_device(s->device));
	// The address that hipGetSymbolAddress() returned for this device, never the array's name
	unsigned char* scratch = s->no_alloc_scratch;
	uint32_t* control = (uint32_t*) scratch;
	f32x4* staged_origins = (f32x4*) (scratch + PT_NO_ALLOC_CONTROL_BYTES);
	f32x4* staged_directions = staged_origins + PT_NO_ALLOC_MAX_RAYS;
	// The array keeps what the previous call left: the counters must start at zero
	PT_CHECK(hipMemsetAsync(control, 0, PT_NO_ALLOC_CONTROL_BYTES, s->stream));
	hipLaunchKernelGGL(k_stage_user_rays, dim3(pt_grid(ray_count)), dim3(PT_BLOCK), 0, s->stream, device_origins, device_directions, ray_count, control, staged_origins, staged_directions);
	PtUserRayArgs a;
	a.scene = pt_scene_view(s);
	a.origins = staged_origins;
	a.directions = staged_directions;
	a.count = control;
	a.fetch = control + 1;
	a.stack_overflow = s->stack_overflow;
	a.stack_overflow_stride = s->stack_overflow_stride;
	a.stack_capacity = s->stack_capacity;
	a.cache_layout = pt_cache_layout(s->bvh_info.node_count, s->bvh_info.triangle_count);
	a.error_word = s->error_word;
	a.triangles = device_triangle_indices;
	a.t_u_v = device_t_u_v;
	a.front = device_front;
	hipLaunchKernelGGL(k_trace_user_rays, dim3(pt_trace_grid(s, ray_count)), dim3(PT_TRACE_BLOCK), 0, s->stream, a);
	PT_CHECK(hipGetLastError());
	PT_CHECK(hipStreamSynchronize(s->stream));
	return pt_check_device_errors(s);
}

static int pt_evaluate_shading_data_impl(pt_scene_t* s, const pt_constants_t* constants,
	const int32_t* triangle_indices, const float* barycentrics, const uint8_t* front, const float* out_dirs, const float* footprints,
	uint32_t count, float* out_shading_data)
{
	if (!s || !constants || (count && (!triangle_indices || !barycentrics || !front || !out_dirs || !out_shading_data)))
		return 1;
	if (count == 0)
		return 0;
	for (uint32_t i = 0; i != count; ++i) {
		if (triangle_indices[i] < 0 || (uint32_t) triangle_indices[i] >= s->triangle_count) {
			printf("Triangle index %d of hit %u is out of range.\n", triangle_indices[i], i);
			return 1;
		}
	}
	PT_CHECK(pt_select_device(s->device));
	PT_CHECK(hipStreamSynchronize(s->stream));
	int32_t* d_triangles = NULL; float* d_barycentrics = NULL; uint8_t* d_front = NULL; float* d_dirs = NULL; float* d_footprints = NULL; float* d_out = NULL;
	int result = 1;
	do {
		if (pt_alloc(&d_triangles, count) || pt_alloc(&d_barycentrics, 2 * (uint64_t) count) || pt_alloc(&d_front, count) || pt_alloc(&d_dirs, 3 * (uint64_t) count) || pt_alloc(&d_out, 20 * (uint64_t) count))
			break;
		if (footprints && (pt_alloc(&d_footprints, count) || hipMemcpyAsync(d_footprints, footprints, sizeof(float) * count, hipMemcpyHostToDevice, s->stream) != hipSuccess))
			break;
		if (hipMemcpyAsync(s->constants, constants, sizeof(PtConstants), hipMemcpyHostToDevice, s->stream) != hipSuccess
			|| hipMemcpyAsync(d_triangles, triangle_indices, sizeof(int32_t) * count, hipMemcpyHostToDevice, s->stream) != hipSuccess
			|| hipMemcpyAsync(d_barycentrics, barycentrics, sizeof(float) * 2 * count, hipMemcpyHostToDevice, s->stream) != hipSuccess
			|| hipMemcpyAsync(d_front, front, count, hipMemcpyHostToDevice, s->stream) != hipSuccess
			|| hipMemcpyAsync(d_dirs, out_dirs, sizeof(float) * 3 * count, hipMemcpyHostToDevice, s->stream) != hipSuccess)
			break;
		hipLaunchKernelGGL(k_shading_data, dim3(pt_grid(count)), dim3(PT_BLOCK), 0, s->stream, pt_scene_view(s), s->constants, d_triangles, d_barycentrics, d_front, d_dirs, (const float*) d_footprints, count, d_out);
		if (hipMemcpyAsync(out_shading_data, d_out, sizeof(float) * 20 * count, hipMemcpyDeviceToHost, s->stream) != hipSuccess
			|| hipStreamSynchronize(s->stream) != hipSuccess || hipGetLastError() != hipSuccess)
			break;
		result = 0;
	} while (0);
	hipFree(d_triangles); hipFree(d_barycentrics); hipFree(d_front); hipFree(d_dirs); hipFree(d_footprints); hipFree(d_out);
	if (result)
		printf("Failed to evaluate shading data on the GPU.\n");
	return result;
}

__global__ void __launch_bounds__(PT_BLOCK) k_evaluate_function(PtSceneView scene, int function, const float* shading_data, const float* inputs, uint32_t count, float* outputs) {
	uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i >= count)
		return;
	float s[20], in[4], out[4];
	for (int j = 0; j != 20; ++j)
		s[j] = shading_data[j];
	for (int j = 0; j != 4; ++j)
		in[j] = inputs[4 * (uint64_t) i + j];
	pt_evaluate_function(scene, function, s, in, out);
	for (int j = 0; j != 4; ++j)
		outputs[4 * (uint64_t) i + j] = out[j];
}

static int pt_evaluate_function_impl(pt_scene_t* s, int function, const float* shading_data, const float* inputs, uint32_t count, float* outputs) {
	if (!s || !shading_data || (count && (!inputs || !outputs)))
		return 1;
	if (function < 0 || function >= PT_FUNCTION_COUNT) {
		printf("pt_evaluate_function() knows the functions 0 to %d, got %d.\n", PT_FUNCTION_COUNT - 1, function);
		return 1;
	}
	if (count == 0)
		return 0;
	PT_CHECK(pt_select_device(s->device));
	PT_CHECK(hipStreamSynchronize(s->stream));
	float* d_shading_data = NULL; float* d_inputs = NULL; float* d_outputs = NULL;
	int result = 1;
	do {
		if (pt_alloc(&d_shading_data, 20) || pt_alloc(&d_inputs, 4 * (uint64_t) count) || pt_alloc(&d_outputs, 4 * (uint64_t) count))
			break;
		if (hipMemcpyAsync(d_shading_data, shading_data, sizeof(float) * 20, hipMemcpyHostToDevice, s->stream) != hipSuccess
			|| hipMemcpyAsync(d_inputs, inputs, sizeof(float) * 4 * count, hipMemcpyHostToDevice, s->stream) != hipSuccess)
			break;
		hipLaunchKernelGGL(k_evaluate_function, dim3(pt_grid(count)), dim3(PT_BLOCK), 0, s->stream, pt_scene_view(s), function, d_shading_data, d_inputs, count, d_outputs);
		if (hipMemcpyAsync(outputs, d_outputs, sizeof(float) * 4 * count, hipMemcpyDeviceToHost, s->stream) != hipSuccess
			|| hipStreamSynchronize(s->stream) != hipSuccess || hipGetLastError() != hipSuccess)
			break;
		result = 0;
	} while (0);
	hipFree(d_shading_data); hipFree(d_inputs); hipFree(d_outputs);
	if (result)
		printf("Failed to evaluate function %d on the GPU.\n", function);
	return result;
}

extern "C" int pt_random_numbers(int device_index, uint32_t x, uint32_t y, uint32_t frame_index, uint32_t draw_count, float* out_randoms) {
	if (!out_randoms && draw_count)
		return 1;
	if (draw_count == 0)
		return 0;
	PT_CHECK(pt_select_device(device_index));
	float* d_out = NULL;
	if (pt_alloc(&d_out, 2 * (uint64_t) draw_count))
		return 1;
	hipLaunchKernelGGL(k_random_numbers, dim3(1), dim3(64), 0, 0, x, y, frame_index, draw_count, d_out);
	hipError_t err = hipMemcpy(out_randoms, d_out, sizeof(float) * 2 * draw_count, hipMemcpyDeviceToHost);
	hipFree(d_out);
	if (err != hipSuccess || hipGetLastError() != hipSuccess) {
		printf("Failed to generate random numbers on the GPU.\n");
		return 1;
	}
	return 0;
}


// ---------------------------------------------------------------- several GPUs
//
// Samples are independent given pixel and frame index
// (src/shaders/pathtrace.frag.glsl:368), so a context replicates the scene on
// every GPU, GPU g of n renders the frame indices first + g, first + g + n, ...
// into its own target and one RCCL reduction over xGMI sums the targets into
// that of the first GPU. Nothing else is exchanged. RCCL is loaded when the
// first context is created, so hosts with one GPU do not depend on it.

#include <rccl/rccl.h>
#include <dlfcn.h>

struct pt_rccl_api {
	void* library;
	ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*);
	ncclResult_t (*CommDestroy)(ncclComm_t);
	ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t);
	ncclResult_t (*GroupStart)();
	ncclResult_t (*GroupEnd)();
	const char* (*GetErrorString)(ncclResult_t);
};
static pt_rccl_api pt_rccl;

static int pt_load_rccl() {
	if (pt_rccl.library)
		return 0;
	void* library = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
	if (!library)
		library = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
	if (!library) {
		printf("Failed to load RCCL (librccl.so.1), which sums the render targets of several GPUs: %s.\n", dlerror());
		return 1;
	}
	pt_rccl_api api;
	api.library = library;
	api.CommInitAll = (decltype(api.CommInitAll)) dlsym(library, "ncclCommInitAll");
	api.CommDestroy = (decltype(api.CommDestroy)) dlsym(library, "ncclCommDestroy");
	api.Reduce = (decltype(api.Reduce)) dlsym(library, "ncclReduce");
	api.GroupStart = (decltype(api.GroupStart)) dlsym(library, "ncclGroupStart");
	api.GroupEnd = (decltype(api.GroupEnd)) dlsym(library, "ncclGroupEnd");
	api.GetErrorString = (decltype(api.GetErrorString)) dlsym(library, "ncclGetErrorString");
	if (!api.CommInitAll || !api.CommDestroy || !api.Reduce || !api.GroupStart || !api.GroupEnd || !api.GetErrorString) {
		printf("The RCCL library lacks one of the functions ncclCommInitAll, ncclCommDestroy, ncclReduce, ncclGroupStart, ncclGroupEnd, ncclGetErrorString.\n");
		dlclose(library);
		return 1;
	}
	pt_rccl = api;
	return 0;
}

#define PT_RCCL_CHECK(call) do { ncclResult_t pt_nccl_ = (call); if (pt_nccl_ != ncclSuccess) { \
	printf("RCCL error %s at %s:%d (%s).\n", pt_rccl.GetErrorString(pt_nccl_), __FILE__, __LINE__, #call); return 1; } } while (0)

struct pt_context {
	std::vector<pt_scene*> scenes;
	std::vector<int> devices;
	std::vector<ncclComm_t> comms;
	// A GPU is listed more than once (PT_CONTEXT_LOOPBACK=1, for tests on a
	// machine with one GPU): its replicas are summed by a kernel on that GPU,
	// RCCL has nothing to carry
	bool loopback;
	double reduce_ms; // wall time of the last pt_context_reduce()
};

__global__ void __launch_bounds__(PT_BLOCK) k_add_target(f32x4* sum, const f32x4* term, uint32_t pixel_count) {
	uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i < pixel_count) {
		f32x4 a = sum[i], b = term[i];
		sum[i] = mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
	}
}

// After the sum the alpha channel of the first GPU's target holds the number
// of GPUs whose target was not empty; the reference keeps it at 1 (src/main.c:1015-1017)
__global__ void __launch_bounds__(PT_BLOCK) k_reset_alpha(f32x4* accum, uint32_t pixel_count) {
	uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i < pixel_count)
		accum[i].w = 1.0f;
}

static void pt_destroy_context_impl(pt_context* c) {
	if (!c)
		return;
	for (size_t i = 0; i != c->comms.size(); ++i) {
		if (c->comms[i]) {
			hipSetDevice(c->devices[i]);
			pt_rccl.CommDestroy(c->comms[i]);
		}
	}
	for (size_t i = 0; i != c->scenes.size(); ++i)
		pt_destroy_scene(c->scenes[i]); // NULL where a replica was not created
	delete c;
}

static int pt_create_context_impl(pt_context** out_context, const int* device_ids, uint32_t device_count, const pt_scene_header_t* header,
	const uint32_t* positions, const uint16_t* normals_and_tex_coords, const uint8_t* material_indices, const pt_texture_desc_t* textures, uint32_t texture_count)
{
	if (!device_ids || device_count == 0 || device_count > PT_MAX_DEVICES) {
		printf("pt_create_context() needs between 1 and %d GPU indices.\n", PT_MAX_DEVICES);
		return 1;
	}
	// PT_CONTEXT_LOOPBACK=1 lets a GPU be listed several times: every entry gets
	// a replica of its own and its own share of the frames, which runs the whole
	// multi-GPU logic of this file on one GPU (only the reduction cannot go
	// through RCCL then, which refuses a GPU listed twice)
	const char* loopback_allowed = getenv("PT_CONTEXT_LOOPBACK");
	bool loopback = false;
	for (uint32_t i = 0; i != device_count; ++i) {
		for (uint32_t j = 0; j != i; ++j) {
			if (device_ids[i] == device_ids[j]) {
				if (!(loopback_allowed && loopback_allowed[0] == '1')) {
					printf("pt_create_context() got GPU %d twice.\n", device_ids[i]);
					return 1;
				}
				loopback = true;
			}
		}
	}
	if (loopback) {
		for (uint32_t i = 1; i != device_count; ++i) {
			if (device_ids[i] != device_ids[0]) {
				printf("pt_create_context() with PT_CONTEXT_LOOPBACK=1 takes one GPU listed several times, got GPUs %d and %d.\n", device_ids[0], device_ids[i]);
				return 1;
			}
		}
	}
	// One GPU has nothing to sum: RCCL is loaded and communicators are made
	// for two or more (PT_CONTEXT_FORCE_RCCL=1 takes the reduction through RCCL
	// with one GPU as well, for tests on a one-GPU machine)
	const char* force_rccl = getenv("PT_CONTEXT_FORCE_RCCL");
	const bool use_rccl = !loopback && (device_count > 1 || (force_rccl && force_rccl[0] == '1'));
	if (use_rccl && pt_load_rccl())
		return 1;
	pt_context* c = new pt_context();
	*out_context = c;
	c->reduce_ms = 0.0;
	c->loopback = loopback;
	c->devices.assign(device_ids, device_ids + device_count);
	c->scenes.assign(device_count, (pt_scene*) NULL);
	// The replicas are independent (upload, texture decode and BVH build on
	// each GPU's own stream): one host thread per GPU creates them side by side
	std::vector<int> results(device_count, 1);
	auto create_replica = [&](uint32_t i) {
		results[i] = pt_create_scene(&c->scenes[i], device_ids[i], header, positions, normals_and_tex_coords, material_indices, textures, texture_count);
	};
	if (device_count == 1)
		create_replica(0);
	else {
		std::vector<std::thread> workers;
		for (uint32_t i = 0; i != device_count; ++i)
			workers.emplace_back(create_replica, i);
		for (size_t i = 0; i != workers.size(); ++i)
			workers[i].join();
	}
	for (uint32_t i = 0; i != device_count; ++i) {
		if (results[i]) {
			printf("Failed to replicate the scene on GPU %d.\n", device_ids[i]);
			return 1;
		}
	}
	if (use_rccl) {
		c->comms.assign(device_count, (ncclComm_t) NULL);
		PT_RCCL_CHECK(pt_rccl.CommInitAll(c->comms.data(), (int) device_count, c->devices.data()));
	}
	return 0;
}

// The frames that GPU g of n renders out of sample_count frames from first on: (first, count, stride)
static void pt_context_slice(uint32_t g, uint32_t n, uint32_t first_frame_index, uint32_t sample_count, uint32_t* out_first, uint32_t* out_count) {
	*out_first = first_frame_index + g;
	*out_count = (sample_count > g) ? (sample_count - g + n - 1) / n : 0;
}

static int pt_context_render_samples_impl(pt_context* c, const pt_constants_t* constants, uint32_t first_frame_index, uint32_t sample_count, int accumulate) {
	if (!c || !constants)
		return 1;
	const uint32_t n = (uint32_t) c->scenes.size();
	std::vector<uint32_t> done(n, 0u);
	// The GPUs take turns, one batch each, so that all of them have work
	// queued while the host is still handing out the rest
	bool any = true;
	while (any) {
		any = false;
		for (uint32_t g = 0; g != n; ++g) {
			pt_scene* s = c->scenes[g];
			uint32_t first, count;
			pt_context_slice(g, n, first_frame_index, sample_count, &first, &count);
			if (count == 0 && !accumulate && done[g] == 0) {
				// This GPU has no frame of a fresh accumulation: its target must not keep an old one
				if (!s->accum) {
					printf("pt_context_render_samples() needs render targets; call pt_context_create_render_targets() first.\n");
					return 1;
				}
				PT_CHECK(pt_select_device(s->device));
				PT_CHECK(hipMemsetAsync(s->accum, 0, sizeof(f32x4) * (uint64_t) s->width * s->height, s->stream));
				s->sample_count = 0;
				done[g] = 1;
				continue;
			}
			if (done[g] >= count)
				continue;
			uint32_t frames = count - done[g];
			uint32_t batch = s->max_frames_per_batch ? s->max_frames_per_batch : 1;
			if (frames > batch)
				frames = batch;
			if (pt_render_samples(s, constants, first + done[g] * n, frames, n, (accumulate || done[g] != 0) ? 1 : 0))
				return 1;
			done[g] += frames;
			any = any || done[g] < count;
		}
	}
	return 0;
}

static int pt_context_reduce_impl(pt_context* c) {
	if (!c)
		return 1;
	const uint32_t n = (uint32_t) c->scenes.size();
	pt_scene* root = c->scenes[0];
	if (!root->accum) {
		printf("pt_context_reduce() needs render targets; call pt_context_create_render_targets() first.\n");
		return 1;
	}
	const uint64_t pixel_count = (uint64_t) root->width * root->height;
	for (uint32_t g = 0; g != n; ++g) {
		if (c->scenes[g]->width != root->width || c->scenes[g]->height != root->height || !c->scenes[g]->accum) {
			printf("The render targets of GPU %d do not match those of GPU %d.\n", c->devices[g], c->devices[0]);
			return 1;
		}
	}
	struct timespec begin, end;
	for (uint32_t g = 0; g != n; ++g)
		if (pt_synchronize(c->scenes[g]))
			return 1;
	clock_gettime(CLOCK_MONOTONIC, &begin);
	// One sum over xGMI: every GPU contributes its target, the first receives
	// the total. A context of one GPU has no communicators and nothing to sum
	if (!c->comms.empty()) {
		PT_RCCL_CHECK(pt_rccl.GroupStart());
		for (uint32_t g = 0; g != n; ++g) {
			pt_scene* s = c->scenes[g];
			PT_CHECK(pt_select_device(s->device));
			PT_RCCL_CHECK(pt_rccl.Reduce(s->accum, s->accum, 4 * pixel_count, ncclFloat, ncclSum, 0, c->comms[g], s->stream));
		}
		PT_RCCL_CHECK(pt_rccl.GroupEnd());
	}
	else if (c->loopback) {
		// All replicas live on one GPU and their streams are idle: the first one's stream adds them up
		PT_CHECK(pt_select_device(root->device));
		for (uint32_t g = 1; g != n; ++g)
			hipLaunchKernelGGL(k_add_target, dim3(pt_grid(pixel_count)), dim3(PT_BLOCK), 0, root->stream, root->accum, c->scenes[g]->accum, (uint32_t) pixel_count);
		PT_CHECK(hipStreamSynchronize(root->stream));
	}
	uint32_t total_samples = 0;
	for (uint32_t g = 0; g != n; ++g) {
		pt_scene* s = c->scenes[g];
		total_samples += s->sample_count;
		PT_CHECK(pt_select_device(s->device));
		if (g == 0) {
			if (!c->comms.empty() || c->loopback)
				hipLaunchKernelGGL(k_reset_alpha, dim3(pt_grid(pixel_count)), dim3(PT_BLOCK), 0, s->stream, s->accum, (uint32_t) pixel_count);
		}
		else {
			// What this GPU rendered is part of the total now; it starts over,
			// so that another reduction after more frames counts nothing twice
			PT_CHECK(hipMemsetAsync(s->accum, 0, sizeof(f32x4) * pixel_count, s->stream));
			s->sample_count = 0;
		}
	}
	root->sample_count = total_samples;
	for (uint32_t g = 0; g != n; ++g) {
		PT_CHECK(hipSetDevice(c->scenes[g]->device));
		PT_CHECK(hipStreamSynchronize(c->scenes[g]->stream));
	}
	clock_gettime(CLOCK_MONOTONIC, &end);
	c->reduce_ms = 1.0e3 * (double) (end.tv_sec - begin.tv_sec) + 1.0e-6 * (double) (end.tv_nsec - begin.tv_nsec);
	PT_CHECK(hipGetLastError());
	return 0;
}

extern "C" int pt_create_context(pt_context_t** out_context, const int* device_ids, uint32_t device_count, const pt_scene_header_t* header,
	const uint32_t* positions, const uint16_t* normals_and_tex_coords, const uint8_t* material_indices, const pt_texture_desc_t* textures, uint32_t texture_count)
{
	if (!out_context)
		return 1;
	*out_context = NULL;
	int result = pt_guarded("pt_create_context", [&]() { return pt_create_context_impl(out_context, device_ids, device_count, header, positions, normals_and_tex_coords, material_indices, textures, texture_count); });
	if (result && *out_context) {
		pt_destroy_context_impl(*out_context);
		*out_context = NULL;
	}
	return result;
}

extern "C" void pt_destroy_context(pt_context_t* context) {
	pt_guarded("pt_destroy_context", [&]() { pt_destroy_context_impl(context); return 0; });
}

extern "C" uint32_t pt_context_device_count(const pt_context_t* context) { return context ? (uint32_t) context->scenes.size() : 0u; }

extern "C" pt_scene_t* pt_context_scene(pt_context_t* context, uint32_t index) {
	return (context && index < context->scenes.size()) ? context->scenes[index] : NULL;
}

// Applies one call to the scene of every GPU
template <typename Call>
static int pt_context_for_each(pt_context_t* context, Call call) {
	if (!context)
		return 1;
	for (size_t i = 0; i != context->scenes.size(); ++i)
		if (call(context->scenes[i]))
			return 1;
	return 0;
}

extern "C" int pt_context_set_lights(pt_context_t* context, const float* lights, uint32_t count) {
	return pt_context_for_each(context, [&](pt_scene* s) { return pt_set_lights(s, lights, count); });
}
extern "C" int pt_context_set_light_sampling(pt_context_t* context, int light_sampling) {
	return pt_context_for_each(context, [&](pt_scene* s) { return pt_set_light_sampling(s, light_sampling); });
}
extern "C" int pt_context_set_shading_math(pt_context_t* context, int shading_math) {
	return pt_context_for_each(context, [&](pt_scene* s) { return pt_set_shading_math(s, shading_math); });
}
extern "C" int pt_context_set_render_settings(pt_context_t* context, int sampling_strategy, uint32_t path_length, uint32_t emission_material_index) {
	return pt_context_for_each(context, [&](pt_scene* s) { return pt_set_render_settings(s, sampling_strategy, path_length, emission_material_index); });
}
extern "C" int pt_context_create_render_targets(pt_context_t* context, uint32_t width, uint32_t height) {
	return pt_context_for_each(context, [&](pt_scene* s) { return pt_create_render_targets(s, width, height); });
}
extern "C" int pt_context_synchronize(pt_context_t* context) {
	return pt_context_for_each(context, [&](pt_scene* s) { return pt_synchronize(s); });
}
extern "C" int pt_context_render_samples(pt_context_t* context, const pt_constants_t* constants, uint32_t first_frame_index, uint32_t sample_count, int accumulate) {
	return pt_guarded("pt_context_render_samples", [&]() { return pt_context_render_samples_impl(context, constants, first_frame_index, sample_count, accumulate); });
}
extern "C" int pt_context_reduce(pt_context_t* context) {
	return pt_guarded("pt_context_reduce", [&]() { return pt_context_reduce_impl(context); });
}
extern "C" double pt_context_last_reduce_ms(const pt_context_t* context) { return context ? context->reduce_ms : 0.0; }

extern "C" int pt_context_frame_slice(uint32_t gpu, uint32_t gpu_count, uint32_t first_frame_index, uint32_t sample_count, uint32_t* out_first_frame_index, uint32_t* out_sample_count) {
	if (gpu_count == 0 || gpu >= gpu_count || !out_first_frame_index || !out_sample_count)
		return 1;
	pt_context_slice(gpu, gpu_count, first_frame_index, sample_count, out_first_frame_index, out_sample_count);
	return 0;
}


// ---------------------------------------------------------------- C ABI
// The entry points proper: each runs its implementation above under
// pt_guarded(), so that no C++ exception reaches the C caller.

extern "C" int pt_create_scene(pt_scene_t** out_scene, int device_index, const pt_scene_header_t* header,
	const uint32_t* positions, const uint16_t* normals_and_tex_coords, const uint8_t* material_indices,
	const pt_texture_desc_t* textures, uint32_t texture_count)
{
	int result = pt_guarded("pt_create_scene", [&]() { return pt_create_scene_impl(out_scene, device_index, header, positions, normals_and_tex_coords, material_indices, textures, texture_count); });
	if (result && out_scene && *out_scene) {
		// An exception left a partially built scene behind
		pt_destroy_scene(*out_scene);
		*out_scene = NULL;
	}
	return result;
}

extern "C" int pt_set_lights(pt_scene_t* scene, const float* lights, uint32_t count) {
	return pt_guarded("pt_set_lights", [&]() { return pt_set_lights_impl(scene, lights, count); });
}

extern "C" int pt_create_render_targets(pt_scene_t* scene, uint32_t width, uint32_t height) {
	return pt_guarded("pt_create_render_targets", [&]() { return pt_create_render_targets_impl(scene, width, height); });
}

extern "C" int pt_render_samples(pt_scene_t* scene, const pt_constants_t* constants, uint32_t first_frame_index, uint32_t sample_count, uint32_t frame_stride, int accumulate) {
	return pt_guarded("pt_render_samples", [&]() { return pt_render_samples_impl(scene, constants, first_frame_index, sample_count, frame_stride, accumulate); });
}

extern "C" int pt_synchronize(pt_scene_t* scene) {
	return pt_guarded("pt_synchronize", [&]() { return pt_synchronize_impl(scene); });
}

extern "C" int pt_read_accumulation(pt_scene_t* scene, float* rgba, uint32_t* out_sample_count) {
	return pt_guarded("pt_read_accumulation", [&]() { return pt_read_accumulation_impl(scene, rgba, out_sample_count); });
}

extern "C" int pt_set_instrumentation(pt_scene_t* scene, int enable_kernel_timing, int enable_visit_counters) {
	return pt_guarded("pt_set_instrumentation", [&]() { return pt_set_instrumentation_impl(scene, enable_kernel_timing, enable_visit_counters); });
}

extern "C" int pt_get_statistics(pt_scene_t* scene, pt_statistics_t* out_statistics) {
	return pt_guarded("pt_get_statistics", [&]() { return pt_get_statistics_impl(scene, out_statistics); });
}

extern "C" int pt_reset_statistics(pt_scene_t* scene) {
	return pt_guarded("pt_reset_statistics", [&]() { return pt_reset_statistics_impl(scene); });
}

extern "C" int pt_trace_primary_rays(pt_scene_t* scene, const pt_constants_t* constants, uint32_t frame_index, int32_t* triangle_indices, float* barycentrics, uint8_t* front) {
	return pt_guarded("pt_trace_primary_rays", [&]() { return pt_trace_primary_rays_impl(scene, constants, frame_index, triangle_indices, barycentrics, front); });
}

extern "C" int pt_trace_rays(pt_scene_t* scene, const float* origins, const float* directions, uint32_t ray_count, int32_t* triangle_indices, float* t_u_v, uint8_t* front) {
	return pt_guarded("pt_trace_rays", [&]() { return pt_trace_rays_impl(scene, origins, directions, ray_count, triangle_indices, t_u_v, front); });
}

extern "C" int pt_evaluate_shading_data(pt_scene_t* scene, const pt_constants_t* constants, const int32_t* triangle_indices, const float* barycentrics, const uint8_t* front, const float* out_dirs, const float* footprints, uint32_t count, float* out_shading_data) {
	return pt_guarded("pt_evaluate_shading_data", [&]() { return pt_evaluate_shading_data_impl(scene, constants, triangle_indices, barycentrics, front, out_dirs, footprints, count, out_shading_data); });
}

extern "C" int pt_evaluate_function(pt_scene_t* scene, int function, const float* shading_data, const float* inputs, uint32_t count, float* outputs) {
	return pt_guarded("pt_evaluate_function", [&]() { return pt_evaluate_function_impl(scene, function, shading_data, inputs, count, outputs); });
}

extern "C" int pt_trace_rays_no_alloc(pt_scene_t* scene, const float* device_origins, const float* device_directions, uint32_t ray_count, int32_t* device_triangle_indices, float* device_t_u_v, uint8_t* device_front) {
	return pt_guarded("pt_trace_rays_no_alloc", [&]() { return pt_trace_rays_no_alloc_impl(scene, device_origins, device_directions, ray_count, device_triangle_indices, device_t_u_v, device_front); });
}
