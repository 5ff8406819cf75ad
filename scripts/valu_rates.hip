// Measures what vector instructions cost on gfx950 when a SIMD is full (8 waves): SIMD cycles per wave64
// instruction, for the instructions that the traversal and shading kernels are made of.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates scripts/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { printf("%s failed: %s\n", #call, hipGetErrorString(e_)); exit(1); } } while (0)

// Eight independent chains, so that no instruction waits for the one before it
#define OP8(text) \
	asm volatile(text : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s10", "s11"); asm volatile(text : "+v"(a1) : "v"(b), "v"(c) : "vcc", "s10", "s11"); \
	asm volatile(text : "+v"(a2) : "v"(b), "v"(c) : "vcc", "s10", "s11"); asm volatile(text : "+v"(a3) : "v"(b), "v"(c) : "vcc", "s10", "s11"); \
	asm volatile(text : "+v"(a4) : "v"(b), "v"(c) : "vcc", "s10", "s11"); asm volatile(text : "+v"(a5) : "v"(b), "v"(c) : "vcc", "s10", "s11"); \
	asm volatile(text : "+v"(a6) : "v"(b), "v"(c) : "vcc", "s10", "s11"); asm volatile(text : "+v"(a7) : "v"(b), "v"(c) : "vcc", "s10", "s11");

#define KERNEL(name, text) \
__global__ void __launch_bounds__(1024, 8) name(float* out, int iterations, float b, float c) { \
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
	for (int i = 0; i != iterations; ++i) { OP8(text) OP8(text) OP8(text) OP8(text) } \
	out[blockIdx.x * 1024 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)); \
}

KERNEL(k_fma, "v_fma_f32 %0, %1, %2, %0")
KERNEL(k_mul, "v_mul_f32 %0, %1, %0")
KERNEL(k_add, "v_add_f32 %0, %1, %0")
KERNEL(k_sub, "v_sub_f32 %0, %1, %0")
KERNEL(k_max, "v_max_f32 %0, %1, %0")
KERNEL(k_max3, "v_max3_f32 %0, %1, %2, %0")
KERNEL(k_min3, "v_min3_f32 %0, %1, %2, %0")
KERNEL(k_med3, "v_med3_f32 %0, %1, %2, %0")
KERNEL(k_cndmask_vcc, "v_cndmask_b32 %0, %1, %0, vcc")
KERNEL(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %1, %0, s[10:11]")
KERNEL(k_cvt_ubyte0, "v_cvt_f32_ubyte0 %0, %1")
KERNEL(k_cvt_ubyte2, "v_cvt_f32_ubyte2 %0, %0")
KERNEL(k_cvt_u32, "v_cvt_f32_u32 %0, %0")
KERNEL(k_and, "v_and_b32 %0, %1, %0")
KERNEL(k_or3, "v_or3_b32 %0, %1, %2, %0")
KERNEL(k_and_or, "v_and_or_b32 %0, %1, %2, %0")
KERNEL(k_xor, "v_xor_b32 %0, %1, %0")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %1, 3, %0")
KERNEL(k_add_u32, "v_add_u32 %0, %1, %0")
KERNEL(k_add3_u32, "v_add3_u32 %0, %1, %2, %0")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
KERNEL(k_ffbh, "v_ffbh_u32 %0, %0")
KERNEL(k_cmp_f32_vcc, "v_cmp_le_f32 vcc, %1, %0")
KERNEL(k_cmp_f32_sgpr, "v_cmp_le_f32_e64 s[10:11], %1, %0")
KERNEL(k_cmp_u32_sgpr, "v_cmp_eq_u32_e64 s[10:11], %1, %0")
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_rsq, "v_rsq_f32 %0, %0")
KERNEL(k_div_fixup, "v_div_fixup_f32 %0, %1, %2, %0")
KERNEL(k_div_scale, "v_div_scale_f32 %0, vcc, %1, %2, %0")
KERNEL(k_div_fmas, "v_div_fmas_f32 %0, %1, %2, %0")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_perm, "v_perm_b32 %0, %1, %2, %0")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %1, %2, %0 bitop3:0x30")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL(k_bitop3_select, "v_bitop3_b32 %0, %1, %2, %0 bitop3:0xca")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %2, %0")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 31")
KERNEL(k_or, "v_or_b32 %0, %1, %0")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0")
KERNEL(k_sub_u32, "v_sub_u32 %0, %1, %0")
KERNEL(k_min, "v_min_f32 %0, %1, %0")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL(k_mul_u32_u24, "v_mul_u32_u24 %0, %1, %0")
KERNEL(k_not, "v_not_b32 %0, %0")
KERNEL(k_max_u32, "v_max_u32 %0, %1, %0")
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL(k_floor, "v_floor_f32 %0, %0")
KERNEL(k_fract, "v_fract_f32 %0, %0")
KERNEL(k_add_f32_e64_neg, "v_add_f32_e64 %0, -%1, %0")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %1")
KERNEL(k_exp, "v_exp_f32 %0, %0")
KERNEL(k_log, "v_log_f32 %0, %0")
KERNEL(k_sin, "v_sin_f32 %0, %0")
KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %1, %0")
KERNEL(k_fma_abs, "v_fma_f32 %0, |%1|, %2, %0")
KERNEL(k_mul_e64, "v_mul_f32_e64 %0, |%1|, %0")

// Packed half precision: two results per lane and instruction
KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %1, %2, %0")
KERNEL(k_pk_mul_f16, "v_pk_mul_f16 %0, %1, %0")
KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %1, %0")
KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %1, %0")
KERNEL(k_pk_min_f16, "v_pk_min_f16 %0, %1, %0")
KERNEL(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %1, %0")
KERNEL(k_fma_f16, "v_fma_f16 %0, %1, %2, %0")
KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %1, %0")
KERNEL(k_pk_sub_i16, "v_pk_sub_i16 %0, %1, %0")
KERNEL(k_pk_mad_i16, "v_pk_mad_i16 %0, %1, %2, %0")
KERNEL(k_pk_lshlrev_b16, "v_pk_lshlrev_b16 %0, 3, %0")

// Two different instructions in turn (16 of each per 32): do their costs add, or do they run on different pipes?
#define OP8_MIXED(text_a, text_b) \
	asm volatile(text_a : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s10", "s11"); asm volatile(text_b : "+v"(a1) : "v"(b), "v"(c) : "vcc", "s10", "s11"); \
	asm volatile(text_a : "+v"(a2) : "v"(b), "v"(c) : "vcc", "s10", "s11"); asm volatile(text_b : "+v"(a3) : "v"(b), "v"(c) : "vcc", "s10", "s11"); \
	asm volatile(text_a : "+v"(a4) : "v"(b), "v"(c) : "vcc", "s10", "s11"); asm volatile(text_b : "+v"(a5) : "v"(b), "v"(c) : "vcc", "s10", "s11"); \
	asm volatile(text_a : "+v"(a6) : "v"(b), "v"(c) : "vcc", "s10", "s11"); asm volatile(text_b : "+v"(a7) : "v"(b), "v"(c) : "vcc", "s10", "s11");
#define KERNEL_MIXED(name, text_a, text_b) \
__global__ void __launch_bounds__(1024, 8) name(float* out, int iterations, float b, float c) { \
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
	for (int i = 0; i != iterations; ++i) { OP8_MIXED(text_a, text_b) OP8_MIXED(text_a, text_b) OP8_MIXED(text_a, text_b) OP8_MIXED(text_a, text_b) } \
	out[blockIdx.x * 1024 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)); \
}
KERNEL_MIXED(k_mix_fma_cndmask, "v_fma_f32 %0, %1, %2, %0", "v_cndmask_b32_e64 %0, %1, %0, s[10:11]")
KERNEL_MIXED(k_mix_fma_cvt, "v_fma_f32 %0, %1, %2, %0", "v_cvt_f32_ubyte0 %0, %1")
KERNEL_MIXED(k_mix_fma_max3, "v_fma_f32 %0, %1, %2, %0", "v_max3_f32 %0, %1, %2, %0")
KERNEL_MIXED(k_mix_cndmask_cvt, "v_cndmask_b32_e64 %0, %1, %0, s[10:11]", "v_cvt_f32_ubyte0 %0, %1")
KERNEL_MIXED(k_mix_fma_rcp, "v_fma_f32 %0, %1, %2, %0", "v_rcp_f32 %0, %0")
KERNEL_MIXED(k_mix_max3_rcp, "v_max3_f32 %0, %1, %2, %0", "v_rcp_f32 %0, %0")
KERNEL_MIXED(k_mix_fma_add, "v_fma_f32 %0, %1, %2, %0", "v_add_f32 %0, %1, %0")
KERNEL_MIXED(k_mix_fma_and, "v_fma_f32 %0, %1, %2, %0", "v_and_b32 %0, %1, %0")
KERNEL_MIXED(k_mix_pkfma16_perm, "v_pk_fma_f16 %0, %1, %2, %0", "v_perm_b32 %0, %1, %2, %0")
KERNEL_MIXED(k_mix_pkfma16_pkmax16, "v_pk_fma_f16 %0, %1, %2, %0", "v_pk_max_f16 %0, %1, %0")
KERNEL_MIXED(k_mix_pkfma16_and, "v_pk_fma_f16 %0, %1, %2, %0", "v_and_b32 %0, %1, %0")
KERNEL_MIXED(k_mix_pkmax16_pkmin16, "v_pk_max_f16 %0, %1, %0", "v_pk_min_f16 %0, %1, %0")

// Streams of two-cycle and four-cycle instructions in other proportions (eight per group, then repeated)
#define OP8_PATTERN(t0, t1, t2, t3, t4, t5, t6, t7) \
	asm volatile(t0 : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s10", "s11"); asm volatile(t1 : "+v"(a1) : "v"(b), "v"(c) : "vcc", "s10", "s11"); \
	asm volatile(t2 : "+v"(a2) : "v"(b), "v"(c) : "vcc", "s10", "s11"); asm volatile(t3 : "+v"(a3) : "v"(b), "v"(c) : "vcc", "s10", "s11"); \
	asm volatile(t4 : "+v"(a4) : "v"(b), "v"(c) : "vcc", "s10", "s11"); asm volatile(t5 : "+v"(a5) : "v"(b), "v"(c) : "vcc", "s10", "s11"); \
	asm volatile(t6 : "+v"(a6) : "v"(b), "v"(c) : "vcc", "s10", "s11"); asm volatile(t7 : "+v"(a7) : "v"(b), "v"(c) : "vcc", "s10", "s11");
#define KERNEL_PATTERN(name, t0, t1, t2, t3, t4, t5, t6, t7) \
__global__ void __launch_bounds__(1024, 8) name(float* out, int iterations, float b, float c) { \
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
	for (int i = 0; i != iterations; ++i) { OP8_PATTERN(t0, t1, t2, t3, t4, t5, t6, t7) OP8_PATTERN(t0, t1, t2, t3, t4, t5, t6, t7) OP8_PATTERN(t0, t1, t2, t3, t4, t5, t6, t7) OP8_PATTERN(t0, t1, t2, t3, t4, t5, t6, t7) } \
	out[blockIdx.x * 1024 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)); \
}
#define FAST_OP "v_fma_f32 %0, %1, %2, %0"
#define FAST_OP_2 "v_sub_f32 %0, %1, %0"
#define SLOW_OP "v_cvt_f32_ubyte0 %0, %1"
#define SLOW_OP_2 "v_max3_f32 %0, %1, %2, %0"
KERNEL_PATTERN(k_mix_5fast_3slow, FAST_OP, SLOW_OP, FAST_OP_2, FAST_OP, SLOW_OP_2, FAST_OP, SLOW_OP, FAST_OP_2)
KERNEL_PATTERN(k_mix_6fast_2slow, FAST_OP, SLOW_OP, FAST_OP_2, FAST_OP, FAST_OP, FAST_OP_2, SLOW_OP_2, FAST_OP)
KERNEL_PATTERN(k_mix_7fast_1slow, FAST_OP, FAST_OP_2, FAST_OP, SLOW_OP, FAST_OP, FAST_OP_2, FAST_OP, FAST_OP)
KERNEL_PATTERN(k_mix_3fast_5slow, SLOW_OP, FAST_OP, SLOW_OP_2, SLOW_OP, FAST_OP_2, SLOW_OP_2, FAST_OP, SLOW_OP)
KERNEL_PATTERN(k_mix_2fast_6slow, SLOW_OP, FAST_OP, SLOW_OP_2, SLOW_OP, SLOW_OP, SLOW_OP_2, FAST_OP_2, SLOW_OP)
KERNEL_PATTERN(k_mix_4fast_4slow_runs, FAST_OP, FAST_OP_2, FAST_OP, FAST_OP_2, SLOW_OP, SLOW_OP_2, SLOW_OP, SLOW_OP_2)
KERNEL(k_cndmask_e64_vcc, "v_cndmask_b32_e64 %0, %1, %0, vcc")
// Fused multiply-add with a half-precision operand converted on the fly
#define FMA_MIX "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]"
KERNEL(k_fma_mix, FMA_MIX)
KERNEL_PATTERN(k_mix_2fmamix_1perm, FMA_MIX, FMA_MIX, "v_perm_b32 %0, %1, %2, %0", FMA_MIX, FMA_MIX, "v_perm_b32 %0, %1, %2, %0", FAST_OP_2, SLOW_OP_2)
KERNEL_PATTERN(k_mix_fmamix_max3, FMA_MIX, SLOW_OP_2, FMA_MIX, SLOW_OP_2, FMA_MIX, SLOW_OP_2, FMA_MIX, SLOW_OP_2)
KERNEL_PATTERN(k_mix_fmamix_sub, FMA_MIX, FAST_OP_2, FMA_MIX, FAST_OP_2, FMA_MIX, FAST_OP_2, FMA_MIX, FAST_OP_2)

// Packed single precision: two results per lane and instruction
#define OP8P(text) \
	asm volatile(text : "+v"(p0) : "v"(pb), "v"(pc) : "vcc", "s10", "s11"); asm volatile(text : "+v"(p1) : "v"(pb), "v"(pc) : "vcc", "s10", "s11"); \
	asm volatile(text : "+v"(p2) : "v"(pb), "v"(pc) : "vcc", "s10", "s11"); asm volatile(text : "+v"(p3) : "v"(pb), "v"(pc) : "vcc", "s10", "s11"); \
	asm volatile(text : "+v"(p4) : "v"(pb), "v"(pc) : "vcc", "s10", "s11"); asm volatile(text : "+v"(p5) : "v"(pb), "v"(pc) : "vcc", "s10", "s11"); \
	asm volatile(text : "+v"(p6) : "v"(pb), "v"(pc) : "vcc", "s10", "s11"); asm volatile(text : "+v"(p7) : "v"(pb), "v"(pc) : "vcc", "s10", "s11");
typedef float float2v __attribute__((ext_vector_type(2)));
#define KERNEL_PACKED(name, text) \
__global__ void __launch_bounds__(1024, 8) name(float* out, int iterations, float b, float c) { \
	float2v pb = { b, b }, pc = { c, c }; \
	float2v p0 = { (float) threadIdx.x, 1.0f }, p1 = p0 + 1.0f, p2 = p0 + 2.0f, p3 = p0 + 3.0f, p4 = p0 + 4.0f, p5 = p0 + 5.0f, p6 = p0 + 6.0f, p7 = p0 + 7.0f; \
	for (int i = 0; i != iterations; ++i) { OP8P(text) OP8P(text) OP8P(text) OP8P(text) } \
	float2v sum = ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)); \
	out[blockIdx.x * 1024 + threadIdx.x] = sum.x + sum.y; \
}
KERNEL_PACKED(k_pk_fma, "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL_PACKED(k_pk_mul, "v_pk_mul_f32 %0, %1, %0")
KERNEL_PACKED(k_pk_add, "v_pk_add_f32 %0, %1, %0")

typedef void (*kernel_t)(float*, int, float, float);

int main() {
	int cu_count = 0, clock_khz = 0;
	CHECK(hipDeviceGetAttribute(&cu_count, hipDeviceAttributeMultiprocessorCount, 0));
	CHECK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
	// Two workgroups of 1024 threads per CU: 32 waves, eight per SIMD
	const int blocks = 2 * cu_count, iterations = 20000;
	float* out = NULL;
	CHECK(hipMalloc(&out, sizeof(float) * 1024 * blocks));
	hipEvent_t begin, end;
	CHECK(hipEventCreate(&begin));
	CHECK(hipEventCreate(&end));
	struct { const char* name; kernel_t kernel; } cases[] = {
#define CASE(k) { #k, k }
		CASE(k_fma), CASE(k_mul), CASE(k_add), CASE(k_sub), CASE(k_max), CASE(k_max3), CASE(k_min3), CASE(k_med3), CASE(k_cndmask_vcc), CASE(k_cndmask_sgpr),
		CASE(k_cvt_ubyte0), CASE(k_cvt_ubyte2), CASE(k_cvt_u32), CASE(k_and), CASE(k_or3), CASE(k_and_or), CASE(k_xor), CASE(k_lshl), CASE(k_lshl_add), CASE(k_add_u32), CASE(k_add3_u32),
		CASE(k_bfe), CASE(k_bcnt), CASE(k_ffbh), CASE(k_cmp_f32_vcc), CASE(k_cmp_f32_sgpr), CASE(k_cmp_u32_sgpr), CASE(k_mul_lo_u32), CASE(k_mad_u32_u24),
		CASE(k_rcp), CASE(k_sqrt), CASE(k_rsq), CASE(k_div_fixup), CASE(k_div_scale), CASE(k_div_fmas), CASE(k_mov), CASE(k_perm), CASE(k_bitop3), CASE(k_mbcnt),
		CASE(k_bitop3_select), CASE(k_bfi), CASE(k_alignbit), CASE(k_or), CASE(k_lshr), CASE(k_sub_u32), CASE(k_min), CASE(k_fmac), CASE(k_lshl_or), CASE(k_mul_u32_u24),
		CASE(k_not), CASE(k_max_u32), CASE(k_cvt_i32_f32), CASE(k_floor), CASE(k_fract), CASE(k_add_f32_e64_neg), CASE(k_ldexp), CASE(k_exp), CASE(k_log), CASE(k_sin), CASE(k_add_co),
		CASE(k_mix_fma_cndmask), CASE(k_mix_fma_cvt), CASE(k_mix_fma_max3), CASE(k_mix_cndmask_cvt), CASE(k_mix_fma_rcp), CASE(k_mix_max3_rcp), CASE(k_mix_fma_add), CASE(k_mix_fma_and),
		CASE(k_fma_abs), CASE(k_mul_e64), CASE(k_pk_fma), CASE(k_pk_mul), CASE(k_pk_add),
		CASE(k_pk_fma_f16), CASE(k_pk_mul_f16), CASE(k_pk_add_f16), CASE(k_pk_max_f16), CASE(k_pk_min_f16), CASE(k_cvt_pkrtz), CASE(k_fma_f16),
		CASE(k_pk_max_i16), CASE(k_pk_sub_i16), CASE(k_pk_mad_i16), CASE(k_pk_lshlrev_b16),
		CASE(k_mix_7fast_1slow), CASE(k_mix_6fast_2slow), CASE(k_mix_5fast_3slow), CASE(k_mix_4fast_4slow_runs), CASE(k_mix_3fast_5slow), CASE(k_mix_2fast_6slow), CASE(k_cndmask_e64_vcc), CASE(k_fma_mix), CASE(k_mix_2fmamix_1perm), CASE(k_mix_fmamix_max3), CASE(k_mix_fmamix_sub),
		CASE(k_mix_pkfma16_perm), CASE(k_mix_pkfma16_pkmax16), CASE(k_mix_pkfma16_and), CASE(k_mix_pkmax16_pkmin16),
	};
	printf("%d CUs, %d MHz nominal; cycles are SIMD cycles per wave64 instruction with 8 waves per SIMD at the nominal clock\n", cu_count, clock_khz / 1000);
	for (auto& c : cases) {
		hipLaunchKernelGGL(c.kernel, dim3(blocks), dim3(1024), 0, 0, out, 100, 1.0001f, 0.5f);
		CHECK(hipDeviceSynchronize());
		float best = 1e30f;
		for (int repeat = 0; repeat != 3; ++repeat) {
			CHECK(hipEventRecord(begin, 0));
			hipLaunchKernelGGL(c.kernel, dim3(blocks), dim3(1024), 0, 0, out, iterations, 1.0001f, 0.5f);
			CHECK(hipEventRecord(end, 0));
			CHECK(hipEventSynchronize(end));
			float ms = 0.0f;
			CHECK(hipEventElapsedTime(&ms, begin, end));
			if (ms < best) best = ms;
		}
		// Instructions per SIMD: 8 waves x iterations x 32
		double instructions = 8.0 * iterations * 32.0;
		double cycles = best * 1e-3 * clock_khz * 1e3 / instructions;
		printf("%-18s %7.3f ms  %5.2f cycles\n", c.name + 2, best, cycles);
	}
	CHECK(hipGetLastError());
	hipFree(out);
	return 0;
}
