// PT_SHADING_MATH_FAST: the shading kernel of next event estimation compiled the way a GPU's shader compiler
// treats the reference's GLSL, which asks for no particular rounding: multiplications and additions may fuse,
// division and square root are the hardware's approximations refined to about one unit in the last place
// instead of correctly rounded, and the loops over all lights use the bare reciprocal and square root
// instructions (pt_spherical_light_importance_fast()). This file is the same source as the exact kernel in
// pt_hip.hip, built with other options for device code (see the Makefile); host code is built alike in both, so the
// inline functions that both translation units share are the same on the host.
#include "pt_shade_kernel.h"

void pt_launch_shade_fast(uint32_t grid, size_t dynamic_lds_bytes, hipStream_t stream, const PtShadeArgs& args, bool light_tree) {
	if (light_tree)
		hipLaunchKernelGGL(k_shade<PT_STRATEGY_NEE_TREE_FAST>, dim3(grid), dim3(PT_SHADE_BLOCK), dynamic_lds_bytes, stream, args);
	else
		hipLaunchKernelGGL(k_shade<PT_STRATEGY_NEE_FAST>, dim3(grid), dim3(PT_SHADE_BLOCK), dynamic_lds_bytes, stream, args);
}
