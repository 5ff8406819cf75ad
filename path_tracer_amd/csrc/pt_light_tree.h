// Light tree: next event estimation among thousands of spherical lights
// without a loop over all of them per path vertex. The reference has no
// counterpart (it stops at 32 lights, src/main.c:691-694, and loops over them,
// pathtrace.frag.glsl:120-175); what is kept of it is the light model: a light
// is sampled uniformly within the cone it subtends (pathtrace.frag.glsl:95-106)
// and emits nothing when it lies behind the horizon of the shading point
// (pathtrace.frag.glsl:69-83).
//
// Structure: a complete binary tree in heap order (node 1 is the root, node i
// has the children 2i and 2i + 1, leaf j is node leaf_count + j; leaf_count is
// the light count rounded up to a power of two, so some leaves stay empty).
// The lights of a node are split between its children at the median of their
// centres along the longest axis of the box around those centres, the larger
// half to the left. A node records the box around the spheres below it as
// centre and half extents, their summed squared radii ("power") and the
// squared half diagonal.
//
// Estimator: a light is chosen by one descent from the root; at every node a
// child is taken with probability proportional to its weight
//   power / max(squared distance to the box centre, squared half diagonal),
// zero when the whole box lies below the horizon. Light i is thus chosen with
// a probability P(i) > 0 whenever it can contribute, and a direction w has the
// density sum over the lights i whose cone contains w of P(i) / (2 pi z_i),
// where z_i is the light's importance (solid angle / 2 pi). That sum is what
// multiple importance sampling needs for light and BRDF samples alike; it is
// found by walking the tree along the ray (stackless, by node index) and, per
// light hit, once from the root to its leaf for P(i). The estimate stays
// unbiased; only its variance differs from the loop over all lights.
//
// Which lights a ray meets is looked up in a uniform grid over the lights
// where there is one (about one light per cell, followed cell by cell along
// the ray); the walk of the tree serves what the grid does not: rays that meet
// more than four lights, rays from far outside, light sets without a grid.
#pragma once
#include "pt_common.h"

#include <algorithm>
#include <cmath>
#include <vector>

// Host side: the arrays of PtSceneView::light_nodes and ::tree_lights
struct PtLightTree {
	uint32_t leaf_count = 0;
	std::vector<f32x4> nodes;  // 2 per node, 2 * leaf_count nodes (node 0 unused)
	std::vector<f32x4> lights; // the light of every leaf; empty leaves are zero
};

// Unsigned integers in the order of the floats they are made from (-0 before
// +0, NaN beyond the infinities)
static inline uint32_t pt_light_sort_key(float value) {
	uint32_t bits = pt_as_uint(value);
	return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}

// Places the lights order[begin, end) in a subtree that has
// slot_count leaf slots, starting at leaf slot first_slot
static inline void pt_place_lights(PtLightTree& tree, const float* lights, std::vector<uint32_t>& order, size_t begin, size_t end, size_t first_slot, size_t slot_count) {
	if (begin == end)
		return;
	if (slot_count == 1) {
		const float* l = &lights[4 * (size_t) order[begin]];
		tree.lights[first_slot] = mk4(l[0], l[1], l[2], l[3]);
		return;
	}
	// Median split along the longest axis of the box around the centres
	// (the lowest axis wins ties, the lower index equal coordinates)
	int axis = 0;
	float longest = -1.0f;
	for (int a = 0; a != 3; ++a) {
		float low = lights[4 * (size_t) order[begin] + a], high = low;
		for (size_t i = begin + 1; i != end; ++i) {
			low = fminf(low, lights[4 * (size_t) order[i] + a]);
			high = fmaxf(high, lights[4 * (size_t) order[i] + a]);
		}
		if (high - low > longest) {
			longest = high - low;
			axis = a;
		}
	}
	// Sorted by the bit pattern mapped so that it orders like the number: a
	// total order even if a coordinate is NaN
	std::sort(order.begin() + begin, order.begin() + end, [&](uint32_t lhs, uint32_t rhs) {
		uint32_t l = pt_light_sort_key(lights[4 * (size_t) lhs + axis]), r = pt_light_sort_key(lights[4 * (size_t) rhs + axis]);
		return (l != r) ? (l < r) : (lhs < rhs);
	});
	size_t left_count = (end - begin + 1) / 2;
	pt_place_lights(tree, lights, order, begin, begin + left_count, first_slot, slot_count / 2);
	pt_place_lights(tree, lights, order, begin + left_count, end, first_slot + slot_count / 2, slot_count / 2);
}

static inline void pt_build_light_tree(PtLightTree& tree, const float* lights, uint32_t count) {
	uint32_t leaf_count = 1;
	while (leaf_count < count)
		leaf_count *= 2;
	tree.leaf_count = leaf_count;
	tree.nodes.assign(4 * (size_t) leaf_count, mk4(0.0f, 0.0f, 0.0f, 0.0f));
	tree.lights.assign(leaf_count, mk4(0.0f, 0.0f, 0.0f, 0.0f));
	std::vector<uint32_t> order(count);
	for (uint32_t i = 0; i != count; ++i)
		order[i] = i;
	pt_place_lights(tree, lights, order, 0, count, 0, leaf_count);
	// Boxes as minimum and maximum first
	std::vector<float> box_min(3 * 2 * (size_t) leaf_count, 0.0f), box_max(3 * 2 * (size_t) leaf_count, 0.0f), power(2 * (size_t) leaf_count, 0.0f);
	for (size_t j = 0; j != leaf_count; ++j) {
		f32x4 l = tree.lights[j];
		size_t node = (size_t) leaf_count + j;
		float radius = fabsf(l.w);
		const float center[3] = { l.x, l.y, l.z };
		for (int a = 0; a != 3; ++a) {
			box_min[3 * node + a] = center[a] - radius;
			box_max[3 * node + a] = center[a] + radius;
		}
		power[node] = l.w * l.w;
	}
	for (size_t node = leaf_count - 1; node >= 1; --node) {
		size_t c0 = 2 * node, c1 = 2 * node + 1;
		// A child without power (surplus leaves, lights of radius zero) does not
		// widen the box
		bool use_0 = power[c0] > 0.0f, use_1 = power[c1] > 0.0f;
		for (int a = 0; a != 3; ++a) {
			if (use_0 && use_1) {
				box_min[3 * node + a] = fminf(box_min[3 * c0 + a], box_min[3 * c1 + a]);
				box_max[3 * node + a] = fmaxf(box_max[3 * c0 + a], box_max[3 * c1 + a]);
			}
			else if (use_0 || use_1) {
				box_min[3 * node + a] = box_min[3 * (use_0 ? c0 : c1) + a];
				box_max[3 * node + a] = box_max[3 * (use_0 ? c0 : c1) + a];
			}
		}
		power[node] = (use_0 ? power[c0] : 0.0f) + (use_1 ? power[c1] : 0.0f);
	}
	for (size_t node = 1; node != 2 * (size_t) leaf_count; ++node) {
		float center[3], half[3];
		for (int a = 0; a != 3; ++a) {
			center[a] = 0.5f * (box_min[3 * node + a] + box_max[3 * node + a]);
			half[a] = 0.5f * (box_max[3 * node + a] - box_min[3 * node + a]);
		}
		tree.nodes[2 * node] = mk4(center[0], center[1], center[2], power[node]);
		tree.nodes[2 * node + 1] = mk4(half[0], half[1], half[2], (half[0] * half[0] + half[1] * half[1]) + half[2] * half[2]);
	}
}

// Host side: the arrays of PtSceneView::light_grid_*
struct PtLightGrid {
	bool valid = false;
	uint32_t dim[3] = { 1, 1, 1 };
	float lo[3] = { 0.0f, 0.0f, 0.0f }, cell[3] = { 1.0f, 1.0f, 1.0f }, inv_cell[3] = { 1.0f, 1.0f, 1.0f };
	float reach = 0.0f;
	std::vector<uint32_t> cells;  // first entry of each cell, and the end of the last
	std::vector<f32x4> lights;    // per entry
	std::vector<uint32_t> leaves; // per entry: the light's node in the tree
};

// A grid over the lights of the tree with about lights_per_cell lights per
// cell. grid.valid stays false for light sets that it does not suit (none
// that emit, coordinates that are not finite or astronomically large, lights
// that each cover most of the cells); the tree alone serves those.
static inline void pt_build_light_grid(PtLightGrid& grid, const PtLightTree& tree, float lights_per_cell = 0.5f) {
	grid = PtLightGrid();
	float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
	size_t count = 0;
	for (size_t j = 0; j != tree.lights.size(); ++j) {
		f32x4 l = tree.lights[j];
		if (!(l.w * l.w > 0.0f))
			continue;
		const float center[3] = { l.x, l.y, l.z };
		for (int a = 0; a != 3; ++a) {
			if (!(fabsf(center[a]) <= 1.0e18f) || !(fabsf(l.w) <= 1.0e18f))
				return;
			lo[a] = fminf(lo[a], center[a] - fabsf(l.w));
			hi[a] = fmaxf(hi[a], center[a] + fabsf(l.w));
		}
		++count;
	}
	if (count == 0)
		return;
	float largest = 0.0f, magnitude = 0.0f;
	for (int a = 0; a != 3; ++a) {
		largest = fmaxf(largest, hi[a] - lo[a]);
		magnitude = fmaxf(magnitude, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
	}
	// Lights are listed with this margin around their boxes, and the grid ends
	// two margins beyond the outermost light: a thousandth of the extent, where
	// following a ray through up to 384 cells is off by some 10^-5 of it
	const float margin = 1.0e-3f * largest + 1.0e-5f * magnitude;
	if (!(margin > 0.0f))
		return;
	float extent[3];
	for (int a = 0; a != 3; ++a) {
		lo[a] -= 2.0f * margin;
		hi[a] += 2.0f * margin;
		extent[a] = hi[a] - lo[a];
	}
	// Cells as close to cubes as the limits allow: an axis along which the
	// lights hardly spread gets one layer, and the others share the cells
	const double cell_target = fmax(1.0, (double) count / fmax((double) lights_per_cell, 1.0e-3));
	const uint32_t dim_max = 128;
	bool single[3] = { false, false, false };
	for (int pass = 0; pass != 3; ++pass) {
		double volume = 1.0;
		int axes = 0;
		for (int a = 0; a != 3; ++a)
			if (!single[a]) {
				volume *= (double) extent[a];
				++axes;
			}
		if (axes == 0)
			break;
		double size = pow(volume / cell_target, 1.0 / axes);
		bool changed = false;
		for (int a = 0; a != 3; ++a) {
			if (single[a])
				continue;
			double cells_along = (double) extent[a] / size;
			if (cells_along < 1.5) {
				single[a] = true;
				changed = true;
				grid.dim[a] = 1;
			}
			else
				grid.dim[a] = (uint32_t) fmin(floor(cells_along + 0.5), (double) dim_max);
		}
		if (!changed)
			break;
	}
	for (int a = 0; a != 3; ++a) {
		grid.lo[a] = lo[a];
		grid.cell[a] = extent[a] / (float) grid.dim[a];
		grid.inv_cell[a] = 1.0f / grid.cell[a];
		if (!(grid.cell[a] > 0.0f) || !(grid.inv_cell[a] <= 3.0e38f))
			return;
	}
	// A ray's origin enters the arithmetic with an error of about 10^-7 of its
	// coordinates, which must stay well below the margin
	grid.reach = 2.5e5f * margin;
	const size_t cell_count = (size_t) grid.dim[0] * grid.dim[1] * grid.dim[2];
	const size_t entry_limit = 64 * count + 65536;
	std::vector<uint32_t> range(6 * tree.lights.size(), 0u);
	grid.cells.assign(cell_count + 1, 0u);
	size_t entry_count = 0;
	for (int fill = 0; fill != 2; ++fill) {
		for (size_t j = 0; j != tree.lights.size(); ++j) {
			f32x4 l = tree.lights[j];
			if (!(l.w * l.w > 0.0f))
				continue;
			uint32_t* r = &range[6 * j];
			if (!fill) {
				const float center[3] = { l.x, l.y, l.z };
				for (int a = 0; a != 3; ++a) {
					float reach = fabsf(l.w) + margin;
					float first = floorf(((center[a] - reach) - lo[a]) * grid.inv_cell[a]), last = floorf(((center[a] + reach) - lo[a]) * grid.inv_cell[a]);
					r[2 * a] = (uint32_t) fminf(fmaxf(first, 0.0f), (float) (grid.dim[a] - 1));
					r[2 * a + 1] = (uint32_t) fminf(fmaxf(last, 0.0f), (float) (grid.dim[a] - 1));
				}
				entry_count += (size_t) (r[1] - r[0] + 1) * (r[3] - r[2] + 1) * (r[5] - r[4] + 1);
				if (entry_count > entry_limit)
					return;
			}
			for (uint32_t z = r[4]; z <= r[5]; ++z)
				for (uint32_t y = r[2]; y <= r[3]; ++y)
					for (uint32_t x = r[0]; x <= r[1]; ++x) {
						size_t c = ((size_t) z * grid.dim[1] + y) * grid.dim[0] + x;
						if (!fill)
							++grid.cells[c + 1];
						else {
							uint32_t e = grid.cells[c]++;
							grid.lights[e] = l;
							grid.leaves[e] = (uint32_t) (tree.leaf_count + j);
						}
					}
		}
		if (!fill) {
			// Counts to first entries; the fill pass advances them to the ends
			for (size_t c = 0; c != cell_count; ++c)
				grid.cells[c + 1] += grid.cells[c];
			grid.lights.assign(entry_count, mk4(0.0f, 0.0f, 0.0f, 0.0f));
			grid.leaves.assign(entry_count, 0u);
		}
		else {
			for (size_t c = cell_count; c != 0; --c)
				grid.cells[c] = grid.cells[c - 1];
			grid.cells[0] = 0;
		}
	}
	grid.valid = true;
}
