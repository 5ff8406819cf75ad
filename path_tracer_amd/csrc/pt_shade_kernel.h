// The shading kernel of the wavefront path tracer and what it needs from the
// workgroup (queue compaction, dealing hits and misses to separate waves),
// shared by the two translation units that instantiate it: pt_hip.hip with
// the arithmetic that equals the CPU oracle bit for bit, pt_shade_fast.hip
// with the arithmetic a GPU's shader compiler would use (PT_SHADING_MATH_FAST).
#pragma once
#include "pt_common.h"
#include "pt_shade.h"
#include "pt_light_tree.h"
#include "pt_trace_wave.h" // pt_load_streaming(), pt_store_streaming()
#include <hip/hip_runtime.h>

#define PT_BLOCK 256
#ifndef PT_LIGHT_CACHE_SIZE
// Lights whose importance a shading thread keeps in LDS between the loops of
// pt_sample_lights(), at most: five workgroups with that many fit the 160 KB of a CU
#define PT_LIGHT_CACHE_SIZE 28
#endif
#ifndef PT_SHADE_BLOCK
// Threads per workgroup of the shading kernel. Every workgroup costs one
// atomic per queue (64 threads: 16 ms per Cornell step, 128: 8.7, 256: 6.1,
// 512: 5.9), but the kernel needs just under 96 registers, which allows five
// waves per SIMD, and only workgroups of four waves (one per SIMD) stack up
// to five: with 512 threads four waves per SIMD fit, with 640 (ten waves,
// unevenly spread over the SIMDs) a second workgroup no longer does. Five
// waves are worth more than the atomics saved (Cornell 5.7 instead of 5.9 ms,
// generated city 23.5 instead of 25.6).
#define PT_SHADE_BLOCK 256
#endif
static_assert(sizeof(float) * PT_LIGHT_CACHE_SIZE * PT_SHADE_BLOCK <= 64 * 1024, "dynamic LDS beyond 64 KB has to be requested with hipFuncSetAttribute()");
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 5 // waves per SIMD that the shading kernel is compiled for
#endif
#define PT_INVALID 0xffffffffu


// Blocks are dealt round-robin to the 8 XCDs, each with its own L2. Giving
// XCD x the x-th contiguous eighth of a queue keeps rays that started as
// neighbouring pixels on one L2. Returns false for surplus blocks.
__device__ __forceinline__ bool pt_block_first_slot(uint32_t count, uint32_t* out_first_slot, uint32_t block_size = PT_BLOCK) {
	uint32_t block_count = (count + block_size - 1) / block_size;
	uint32_t chunk = (block_count + 7) / 8;
	uint32_t group = blockIdx.x & 7, j = blockIdx.x >> 3;
	uint32_t virtual_block = group * chunk + j;
	if (j >= chunk || virtual_block >= block_count)
		return false;
	*out_first_slot = virtual_block * block_size;
	return true;
}

// Reserves slots in two queues for the lanes of a whole workgroup with two
// atomics per workgroup. Counters that every wave of a launch adds to are
// served one after the other by a single L2 channel, a few nanoseconds each:
// with one atomic per wave and queue that alone took as long as the shading
// of 66 M paths. All threads of the workgroup must call this.
template <uint32_t BLOCK>
__device__ __forceinline__ void pt_block_push_2(uint32_t* counter_a, bool want_a, uint32_t* counter_b, bool want_b, uint32_t* out_slot_a, uint32_t* out_slot_b) {
	__shared__ uint32_t wave_offsets[2][BLOCK / 64];
	__shared__ uint32_t bases[2];
	const uint32_t wave = threadIdx.x >> 6;
	unsigned long long mask_a = __ballot(want_a), mask_b = __ballot(want_b);
	if ((threadIdx.x & 63) == 0) {
		wave_offsets[0][wave] = (uint32_t) __popcll(mask_a);
		wave_offsets[1][wave] = (uint32_t) __popcll(mask_b);
	}
	__syncthreads();
	if (threadIdx.x < 2) {
		uint32_t total = 0;
		for (uint32_t w = 0; w != BLOCK / 64; ++w) {
			uint32_t count = wave_offsets[threadIdx.x][w];
			wave_offsets[threadIdx.x][w] = total;
			total += count;
		}
		bases[threadIdx.x] = total ? atomicAdd(threadIdx.x ? counter_b : counter_a, total) : 0u;
	}
	__syncthreads();
	*out_slot_a = bases[0] + wave_offsets[0][wave] + __builtin_amdgcn_mbcnt_hi((uint32_t) (mask_a >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask_a, 0u));
	*out_slot_b = bases[1] + wave_offsets[1][wave] + __builtin_amdgcn_mbcnt_hi((uint32_t) (mask_b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask_b, 0u));
}

// Sorts the threads of a workgroup into three groups by a class per thread
// (0, 1, or anything else for "none"): returns the position of this thread
// among the threads of class 0 followed by those of class 1, each in thread
// order, and the sizes of both groups. All threads of the workgroup must call
// this.
template <uint32_t BLOCK>
__device__ __forceinline__ uint32_t pt_block_partition(uint32_t cls, uint32_t* out_count_0, uint32_t* out_count_1) {
	__shared__ uint32_t wave_counts[2][BLOCK / 64];
	const uint32_t wave = threadIdx.x >> 6;
	unsigned long long mask_0 = __ballot(cls == 0u), mask_1 = __ballot(cls == 1u);
	if ((threadIdx.x & 63) == 0) {
		wave_counts[0][wave] = (uint32_t) __popcll(mask_0);
		wave_counts[1][wave] = (uint32_t) __popcll(mask_1);
	}
	__syncthreads();
	uint32_t before_0 = 0, before_1 = 0, total_0 = 0, total_1 = 0;
#pragma unroll
	for (uint32_t w = 0; w != BLOCK / 64; ++w) {
		uint32_t c0 = wave_counts[0][w], c1 = wave_counts[1][w];
		before_0 += (w < wave) ? c0 : 0u;
		before_1 += (w < wave) ? c1 : 0u;
		total_0 += c0;
		total_1 += c1;
	}
	*out_count_0 = total_0;
	*out_count_1 = total_1;
	uint32_t rank_0 = before_0 + __builtin_amdgcn_mbcnt_hi((uint32_t) (mask_0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask_0, 0u));
	uint32_t rank_1 = total_0 + before_1 + __builtin_amdgcn_mbcnt_hi((uint32_t) (mask_1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask_1, 0u));
	return (cls == 0u) ? rank_0 : rank_1;
}

// Copies the sRGB table into LDS for pt_srgb_to_linear(); all threads of the workgroup take part
__device__ __forceinline__ void pt_stage_srgb_table(const float* table) {
	for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x)
		pt_srgb_table_lds[i] = table[i];
	__syncthreads();
}

struct PtShadeArgs {
	PtSceneView scene;
	PtWavefront wf;
	PtBatch batch;
	const PtConstants* constants;
	int strategy;
	uint32_t path_length;
	uint32_t k;       // 1-based path vertex; reads ext queue k - 1
	uint32_t parity;  // ray queue of this bounce; the next one is parity ^ 1
	uint32_t light_cache_capacity; // lights per thread in the dynamic LDS of the launch
	unsigned long long* visit_counters; // [4] counts shaded hits, or NULL
};

// Dynamic LDS of k_shade: light_cache_capacity floats per thread
extern __shared__ float pt_shade_light_importances[];

// One instantiation per sampling strategy: the strategy is a compile-time
// constant inside, so each kernel carries the registers of one estimator only
template <int STRATEGY>
__global__ void __launch_bounds__(PT_SHADE_BLOCK, PT_SHADE_WAVES) k_shade(PtShadeArgs a) {
	uint32_t first_slot;
	uint32_t count = a.wf.ext_count[a.k - 1];
	if (!pt_block_first_slot(count, &first_slot, PT_SHADE_BLOCK))
		return;
	pt_stage_srgb_table(a.scene.srgb_to_linear);
	// Per-light importances of this thread's vertex between the two loops of
	// pt_sample_lights(): lane-contiguous, hence free of bank conflicts
	PtLightCache light_cache;
	light_cache.values = &pt_shade_light_importances[threadIdx.x];
	light_cache.stride = PT_SHADE_BLOCK;
	light_cache.capacity = (STRATEGY == PT_STRATEGY_NEE || STRATEGY == PT_STRATEGY_NEE_FAST) ? a.light_cache_capacity : 0;
	uint32_t slot = first_slot + threadIdx.x;
	bool active = slot < count;
	uint32_t path = PT_INVALID;
	PtVertexResult r;
	r.continues = false;
	r.shadow = false;
	u32x4 hit;
	hit.x = hit.y = hit.z = 0u;
	hit.w = PT_INVALID;
	if (active) {
		hit = a.wf.hit[slot];
		active = hit.w != PT_INVALID;
	}
	// Rays that left the scene need a few instructions, hits need thousands:
	// a wave in which both occur keeps the lanes of the former idle for as long
	// as the latter take. The workgroup therefore deals its slots out anew,
	// hits to its first threads and misses to the following ones, so that all
	// but one of its waves are of one kind. Which thread shades which slot does
	// not change what a path computes.
	{
		__shared__ uint32_t dealt_slots[PT_SHADE_BLOCK];
		uint32_t hit_count, miss_count;
		uint32_t cls = !active ? 2u : ((hit.x == PT_INVALID) ? 1u : 0u);
		uint32_t position = pt_block_partition<PT_SHADE_BLOCK>(cls, &hit_count, &miss_count);
		if (active)
			dealt_slots[position] = slot;
		__syncthreads();
		active = threadIdx.x < hit_count + miss_count;
		if (active) {
			slot = dealt_slots[threadIdx.x];
			hit = a.wf.hit[slot];
		}
	}
	if (active) {
		path = hit.w;
		// The direction carries the ray's cone in its fourth word; the origin is
		// needed for the distance that the cone has travelled
		f32x4 d = pt_load_streaming(&a.wf.ray_d[a.parity][slot]), o = pt_load_streaming(&a.wf.ray_o[a.parity][slot]);
		v3 throughput = mk3(1.0f, 1.0f, 1.0f), nee_throughput = mk3(1.0f, 1.0f, 1.0f), emitted = mk3(0.0f, 0.0f, 0.0f);
		uint32_t sx, sy;
		if (a.k == 1) {
			// The state of a path at its first vertex follows from its index
			uint32_t x, y, frame_slot;
			pt_path_to_pixel(a.batch, path, &x, &y, &frame_slot);
			pt_seed_after_primary_ray(x, y, a.batch.first_frame_index + frame_slot * a.batch.frame_stride, sx, sy);
		}
		else {
			f32x4 s0 = a.wf.state0[path], s1 = a.wf.state1[path];
			throughput = mk3(s0.x, s0.y, s0.z);
			nee_throughput = mk3(s1.x, s1.y, s1.z);
			sx = pt_as_uint(s0.w);
			sy = pt_as_uint(s1.w);
		}
		bool missed = hit.x == PT_INVALID;
		if (a.visit_counters && !missed)
			atomicAdd(&a.visit_counters[4], 1ull);
		r = pt_shade_vertex(a.scene, *a.constants, light_cache, STRATEGY, a.path_length, a.k,
			missed ? PT_INVALID : (hit.x & 0x7fffffffu), pt_as_float(hit.y), pt_as_float(hit.z), (hit.x >> 31) != 0, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), d.w,
			emitted, throughput, nee_throughput, sx, sy);
		// Emission is zero at most vertices; radiance + 0 is the radiance (it is
		// never -0), so only other terms, NaN included, touch the path's radiance
		if (emitted.x != 0.0f || emitted.y != 0.0f || emitted.z != 0.0f) {
			f32x4 rad = a.wf.radiance[path];
			a.wf.radiance[path] = mk4(rad.x + emitted.x, rad.y + emitted.y, rad.z + emitted.z, 0.0f);
		}
		if (r.continues) {
			a.wf.state0[path] = mk4(throughput.x, throughput.y, throughput.z, pt_as_float(sx));
			a.wf.state1[path] = mk4(nee_throughput.x, nee_throughput.y, nee_throughput.z, pt_as_float(sy));
		}
	}
	// Compact survivors into the next queues (the whole workgroup takes part)
	uint32_t next_slot, shadow_slot;
	pt_block_push_2<PT_SHADE_BLOCK>(&a.wf.ext_count[a.k], r.continues, &a.wf.shadow_count[a.k], r.shadow, &next_slot, &shadow_slot);
	if (r.continues) {
		pt_store_streaming(&a.wf.ray_o[a.parity ^ 1][next_slot], mk4(r.next_origin.x, r.next_origin.y, r.next_origin.z, pt_as_float(path)));
		pt_store_streaming(&a.wf.ray_d[a.parity ^ 1][next_slot], mk4(r.next_dir.x, r.next_dir.y, r.next_dir.z, r.next_cone));
	}
	if (r.shadow) {
		pt_store_streaming(&a.wf.sh_o[shadow_slot], mk4(r.next_origin.x, r.next_origin.y, r.next_origin.z, pt_as_float(path)));
		// The direction's fourth word stays in a register of the traversal for the life of the ray: the distance at
		// which the sampled light begins, which tells it a hit in front of that light from one behind it
		pt_store_streaming(&a.wf.sh_d[shadow_slot], mk4(r.shadow_dir.x, r.shadow_dir.y, r.shadow_dir.z, r.shadow_near));
		pt_store_streaming(&a.wf.sh_w[shadow_slot], mk4(r.shadow_weight.x, r.shadow_weight.y, r.shadow_weight.z, r.shadow_factor));
	}
}


// k_shade<PT_STRATEGY_NEE_FAST> or, with light_tree, k_shade<PT_STRATEGY_NEE_TREE_FAST>, compiled in pt_shade_fast.hip with fused
// multiply-adds and approximate division and square root in device code
void pt_launch_shade_fast(uint32_t grid, size_t dynamic_lds_bytes, hipStream_t stream, const PtShadeArgs& args, bool light_tree);
