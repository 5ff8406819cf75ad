// Closest-hit ray traversal of the 8-wide compressed BVH. Replaces the ray
// queries of the reference (rayQueryInitializeEXT(..., gl_RayFlagsOpaqueEXT,
// 0xff, origin, 1e-3, dir, 1e38) + rayQueryProceedEXT at
// src/shaders/pathtrace.frag.glsl:183-185 and :205-207).
//
// Intersection semantics (the reference leaves them to the Vulkan driver):
// watertight ray/triangle test of Woop, Benthin and Wald (JCGT 2013) in single
// precision, t strictly inside (tmin, tmax), the closest t wins and equal t is
// broken by the lower triangle index, so the result does not depend on the
// shape of the BVH or the traversal order.
#pragma once
#include "pt_common.h"

struct PtHit {
	uint32_t prim; // index of the triangle in the scene file, 0xffffffff: miss
	float t;
	float u, v;    // barycentric weights of vertex 1 and vertex 2
	bool front;
};

struct PtRayShear {
	int kx, ky, kz;
	float sx, sy, sz;
};

PT_FN PtRayShear pt_ray_shear(v3 dir) {
	PtRayShear r;
	float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
	r.kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
	r.kx = (r.kz == 2) ? 0 : r.kz + 1;
	r.ky = (r.kx == 2) ? 0 : r.kx + 1;
	if (comp(dir, r.kz) < 0.0f) {
		int swap = r.kx;
		r.kx = r.ky;
		r.ky = swap;
	}
	float dz = comp(dir, r.kz);
	r.sx = comp(dir, r.kx) / dz;
	r.sy = comp(dir, r.ky) / dz;
	r.sz = 1.0f / dz;
	return r;
}

PT_FN void pt_intersect_triangle(PtHit& hit, const PtRayShear& r, v3 origin, v3 p0, v3 p1, v3 p2, uint32_t prim, float tmin) {
	v3 a = p0 - origin, b = p1 - origin, c = p2 - origin;
	float az = comp(a, r.kz), bz = comp(b, r.kz), cz = comp(c, r.kz);
	float ax = comp(a, r.kx) - r.sx * az, ay = comp(a, r.ky) - r.sy * az;
	float bx = comp(b, r.kx) - r.sx * bz, by = comp(b, r.ky) - r.sy * bz;
	float cx = comp(c, r.kx) - r.sx * cz, cy = comp(c, r.ky) - r.sy * cz;
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
		// det is minus the signed area of the triangle seen along the ray;
		// front faces have dot(cross(p1 - p0, p2 - p0), dir) < 0
		hit.front = det > 0.0f;
	}
}

// How far, in scene units, the box test has to err on the side of a hit. Both
// the slab test and the triangle test subtract the ray origin from scene
// coordinates first; the rounding of that difference grows with the distance
// between the two, not with the size of the scene, so for a ray that starts
// far outside (an orthographic camera, a caller's own rays) no padding of the
// boxes that is fixed at build time can cover it. The triangle test shears
// the other two axes by the coordinate along the ray's dominant axis, which
// carries that rounding across axes, hence the largest of the three
// distances to the node (d: node origin minus ray origin; a node spans at most
// 255 grid steps of size s per axis beyond its origin) counts for all axes:
// 2^-21 of it, eight units in the last place.
PT_FN float pt_box_test_slack(float dx, float dy, float dz, float sx, float sy, float sz) {
	float reach = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz)) + 255.0f * fmaxf(fmaxf(sx, sy), sz);
	return reach * 4.76837158203125e-07f;
}

PT_FN float pt_byte_to_float(uint32_t word, int byte) { return (float) ((word >> (8 * byte)) & 0xffu); }

// Stack must provide push(u32x2), pop() and empty(). An entry is a group of
// sibling nodes still to visit: x = index of the first inner child of their
// parent, y = pending children (bits 24-31, most urgent highest) | imask.
// With COUNT_VISITS, visits[0] and visits[1] receive the numbers of nodes
// fetched and triangles tested.
template <bool COUNT_VISITS, typename Stack>
PT_FN PtHit pt_trace_closest(const PtNode* nodes, const PtTri* tris, uint32_t triangle_count, v3 origin, v3 dir, float tmin, float tmax, Stack& stack, uint32_t* visits) {
	PtHit hit;
	hit.prim = 0xffffffffu;
	hit.t = tmax;
	hit.u = hit.v = 0.0f;
	hit.front = false;
	if (triangle_count == 0)
		return hit;
	PtRayShear shear = pt_ray_shear(dir);
	// The box test may be conservative, so tiny direction components are
	// clamped to keep reciprocals finite
	float cdx = (fabsf(dir.x) > 1.0e-30f) ? dir.x : copysignf(1.0e-30f, dir.x);
	float cdy = (fabsf(dir.y) > 1.0e-30f) ? dir.y : copysignf(1.0e-30f, dir.y);
	float cdz = (fabsf(dir.z) > 1.0e-30f) ? dir.z : copysignf(1.0e-30f, dir.z);
	float idx = 1.0f / cdx, idy = 1.0f / cdy, idz = 1.0f / cdz;
	bool neg_x = cdx < 0.0f, neg_y = cdy < 0.0f, neg_z = cdz < 0.0f;
	uint32_t octinv = 7u - ((neg_x ? 1u : 0u) | (neg_y ? 2u : 0u) | (neg_z ? 4u : 0u));
	u32x2 ngroup;
	ngroup.x = 0;
	ngroup.y = 0x80000000u;
	while (true) {
		u32x2 tgroup;
		tgroup.x = tgroup.y = 0;
		if (ngroup.y & 0xff000000u) {
			uint32_t bit = 31u - (uint32_t) __builtin_clz(ngroup.y);
			ngroup.y &= ~(1u << bit);
			if (ngroup.y & 0xff000000u)
				stack.push(ngroup);
			uint32_t slot = (bit - 24u) ^ octinv;
			uint32_t rel = (uint32_t) __builtin_popcount(ngroup.y & 0xffu & ((1u << slot) - 1u));
			const u32x4* words = (const u32x4*) &nodes[ngroup.x + rel];
			if (COUNT_VISITS)
				++visits[0];
			u32x4 w0 = words[0], w1 = words[1], w2 = words[2], w3 = words[3], w4 = words[4];
			float sx = pt_as_float((w0.w & 0xffu) << 23), sy = pt_as_float(((w0.w >> 8) & 0xffu) << 23), sz = pt_as_float(((w0.w >> 16) & 0xffu) << 23);
			uint32_t imask = w0.w >> 24;
			float ax = idx * sx, ay = idy * sy, az = idz * sz;
			float dx = pt_as_float(w0.x) - origin.x, dy = pt_as_float(w0.y) - origin.y, dz = pt_as_float(w0.z) - origin.z;
			float bx = dx * idx, by = dy * idy, bz = dz * idz;
			// Slack for the rounding of this test and of the triangle test (see pt_box_test_slack())
			float slack = pt_box_test_slack(dx, dy, dz, sx, sy, sz);
			float ex = slack * fabsf(idx), ey = slack * fabsf(idy), ez = slack * fabsf(idz);
			float bxn = bx - ex, byn = by - ey, bzn = bz - ez, bxf = bx + ex, byf = by + ey, bzf = bz + ez;
			uint32_t qlox[2] = { w2.x, w2.y }, qloy[2] = { w2.z, w2.w }, qloz[2] = { w3.x, w3.y };
			uint32_t qhix[2] = { w3.z, w3.w }, qhiy[2] = { w4.x, w4.y }, qhiz[2] = { w4.z, w4.w };
			uint32_t leafmask = w1.z & 0xffu;
			uint32_t hit8 = 0;
#pragma unroll
			for (int i = 0; i != 8; ++i) {
				float lx = pt_byte_to_float(qlox[i >> 2], i & 3), hx = pt_byte_to_float(qhix[i >> 2], i & 3);
				float ly = pt_byte_to_float(qloy[i >> 2], i & 3), hy = pt_byte_to_float(qhiy[i >> 2], i & 3);
				float lz = pt_byte_to_float(qloz[i >> 2], i & 3), hz = pt_byte_to_float(qhiz[i >> 2], i & 3);
				float nx = fmaf(neg_x ? hx : lx, ax, bxn), fx = fmaf(neg_x ? lx : hx, ax, bxf);
				float ny = fmaf(neg_y ? hy : ly, ay, byn), fy = fmaf(neg_y ? ly : hy, ay, byf);
				float nz = fmaf(neg_z ? hz : lz, az, bzn), fz = fmaf(neg_z ? lz : hz, az, bzf);
				float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, tmin));
				float tf = fminf(fminf(fx, fy), fminf(fz, hit.t));
				if (tn <= tf)
					hit8 |= 1u << i;
			}
			// Inner children in visiting order (slot ^ octinv) in the top byte
			uint32_t order = 0;
			for (uint32_t i = 0; i != 8; ++i)
				if (hit8 & imask & (1u << i))
					order |= 1u << (i ^ octinv);
			ngroup.x = w1.x;
			ngroup.y = (order << 24) | imask;
			// Triangles pending in the low byte (by slot), the leaf mask above it
			tgroup.x = w1.y;
			tgroup.y = (hit8 & leafmask) | (leafmask << 8);
		}
		while (tgroup.y & 0xffu) {
			uint32_t bit = (uint32_t) __builtin_ctz(tgroup.y);
			tgroup.y &= tgroup.y - 1u;
			uint32_t rank = (uint32_t) __builtin_popcount((tgroup.y >> 8) & ((1u << bit) - 1u));
			const u32x4* words = (const u32x4*) &tris[tgroup.x + rank];
			if (COUNT_VISITS)
				++visits[1];
			u32x4 t0 = words[0], t1 = words[1], t2 = words[2];
			v3 p0 = mk3(pt_as_float(t0.x), pt_as_float(t0.y), pt_as_float(t0.z));
			v3 p1 = mk3(pt_as_float(t0.w), pt_as_float(t1.x), pt_as_float(t1.y));
			v3 p2 = mk3(pt_as_float(t1.z), pt_as_float(t1.w), pt_as_float(t2.x));
			pt_intersect_triangle(hit, shear, origin, p0, p1, p2, t2.y, tmin);
		}
		if (!(ngroup.y & 0xff000000u)) {
			if (stack.empty())
				break;
			ngroup = stack.pop();
		}
	}
	return hit;
}
