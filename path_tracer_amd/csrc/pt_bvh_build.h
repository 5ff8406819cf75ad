// GPU construction of the 8-wide compressed BVH that replaces the Vulkan
// acceleration structure build of the reference (src/scene.c:32-82,90-264):
//   1. dequantise vertices exactly like the reference's CPU loop (separate
//      multiply and add, src/scene.c:43-55) and find the scene bounds,
//   2. 63-bit Morton codes of triangle centroids, radix sort,
//   3. binary radix tree over the sorted codes (Karras, HPG 2012),
//   4. bottom-up bounding boxes with one atomic counter per inner node, then
//      bottom-up treelet restructuring for a lower surface area cost, then the
//      costs of the wide collapse,
//   5. top-down collapse into 8-wide nodes by those costs, one triangle per
//      leaf child, child boxes quantised to 8 bits, children placed in slots
//      so that slot ^ ray octant orders them front to back.
// Each step is a per-thread function; the kernels that run them are in
// pt_hip.hip.
#pragma once
#include "pt_common.h"

struct PtBuild {
	uint32_t n; // triangle count
	const u32x2* positions;
	float factor[3], summand[3];
	float* verts;        // 9 floats per triangle, file order
	uint32_t* bounds;    // 6 order-preserving uint keys: min xyz, max xyz
	uint64_t* keys;      // Morton codes
	uint32_t* order;     // triangle index per sorted position
	// binary radix tree: inner nodes 0 .. n-2, leaf k is node n-1+k
	uint32_t* left;
	uint32_t* right;
	uint32_t* parent;    // per node (inner and leaf)
	uint32_t* size;      // triangles below each inner node
	float* sah_cost;     // surface area cost of the subtree of each inner node
	float* wide_cost;    // 8 per inner node: [j-1] = cost of the subtree as at most j children of a wide node
	uint8_t* wide_choice; // 8 per inner node: [j-1] = 0: one child (a wide node of its own), k: left subtree gets k of the j slots
	float* box;          // 6 floats per node (inner and leaf)
	uint32_t* flags;     // per inner node
	// wide BVH
	PtNode* nodes;
	PtTri* tris;
	uint32_t* task_binary; // binary node that wide node i collapses
	uint32_t* counters;    // [0] wide nodes allocated, [1] triangles written
};

// Order-preserving map from float to uint so that integer atomics can
// compute float minima and maxima
PT_FN uint32_t pt_float_to_key(float f) {
	uint32_t u = pt_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
PT_FN float pt_key_to_float(uint32_t k) {
	return pt_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Step 1: vertex arithmetic of src/scene.c:43-55
PT_FN void pt_build_dequantize(const PtBuild& b, uint32_t tri, float out_lo[3], float out_hi[3]) {
	for (int j = 0; j != 3; ++j) { out_lo[j] = INFINITY; out_hi[j] = -INFINITY; }
	for (uint32_t v = 0; v != 3; ++v) {
		u32x2 q = b.positions[3 * (uint64_t) tri + v];
		float pos[3] = {
			(float) (q.x & 0x1fffffu),
			(float) (((q.x & 0xffe00000u) >> 21) | ((q.y & 0x3ffu) << 11)),
			(float) ((q.y & 0x7ffffc00u) >> 10),
		};
		for (int j = 0; j != 3; ++j) {
			float w = pos[j] * b.factor[j] + b.summand[j];
			b.verts[9 * (uint64_t) tri + 3 * v + j] = w;
			out_lo[j] = fminf(out_lo[j], w);
			out_hi[j] = fmaxf(out_hi[j], w);
		}
	}
}

struct PtSceneBounds { float lo[3], hi[3], pad; };

PT_FN PtSceneBounds pt_build_get_bounds(const uint32_t* keys) {
	PtSceneBounds r;
	float max_abs = 0.0f;
	for (int j = 0; j != 3; ++j) {
		r.lo[j] = pt_key_to_float(keys[j]);
		r.hi[j] = pt_key_to_float(keys[3 + j]);
		max_abs = fmaxf(max_abs, fmaxf(fabsf(r.lo[j]), fabsf(r.hi[j])));
	}
	// Boxes are padded by this much so that rounding in the ray/box test can
	// never cull a triangle that the ray/triangle test would accept
	r.pad = max_abs * 1.9073486328125e-06f;
	return r;
}

PT_FN void pt_tri_box(const float* verts, float pad, float lo[3], float hi[3]) {
	for (int j = 0; j != 3; ++j) {
		float v0 = verts[j], v1 = verts[3 + j], v2 = verts[6 + j];
		lo[j] = fminf(v0, fminf(v1, v2)) - pad;
		hi[j] = fmaxf(v0, fmaxf(v1, v2)) + pad;
	}
}

PT_FN uint64_t pt_spread_bits_21(uint64_t v) {
	v &= 0x1fffffull;
	v = (v | (v << 32)) & 0x1f00000000ffffull;
	v = (v | (v << 16)) & 0x1f0000ff0000ffull;
	v = (v | (v << 8)) & 0x100f00f00f00f00full;
	v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
	v = (v | (v << 2)) & 0x1249249249249249ull;
	return v;
}

// Step 2
PT_FN uint64_t pt_build_morton(const PtBuild& b, const PtSceneBounds& sb, uint32_t tri) {
	float lo[3], hi[3];
	pt_tri_box(&b.verts[9 * (uint64_t) tri], 0.0f, lo, hi);
	uint64_t code = 0;
	for (int j = 0; j != 3; ++j) {
		float extent = sb.hi[j] - sb.lo[j];
		float c = 0.5f * (lo[j] + hi[j]);
		float t = (extent > 0.0f) ? (c - sb.lo[j]) / extent : 0.0f;
		t = fminf(fmaxf(t * 2097152.0f, 0.0f), 2097151.0f);
		code |= pt_spread_bits_21((uint64_t) t) << j;
	}
	return code;
}

PT_FN int pt_clz64(uint64_t v) { return v ? __builtin_clzll(v) : 64; }
PT_FN int pt_clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }
PT_FN int pt_popc(uint32_t v) { return __builtin_popcount(v); }

// Length of the common prefix of the keys at sorted positions i and j, with
// the position as tie breaker for equal keys; -1 outside the array
PT_FN int pt_build_delta(const PtBuild& b, int i, int j) {
	if (j < 0 || j >= (int) b.n)
		return -1;
	uint64_t x = b.keys[i] ^ b.keys[j];
	if (x == 0)
		return 64 + pt_clz32((uint32_t) i ^ (uint32_t) j);
	return pt_clz64(x);
}

// Step 3: inner node i of the binary radix tree
PT_FN void pt_build_radix_tree(const PtBuild& b, uint32_t index) {
	int i = (int) index;
	int d = (pt_build_delta(b, i, i + 1) - pt_build_delta(b, i, i - 1)) >= 0 ? 1 : -1;
	int delta_min = pt_build_delta(b, i, i - d);
	int l_max = 2;
	while (pt_build_delta(b, i, i + l_max * d) > delta_min)
		l_max *= 2;
	int l = 0;
	for (int t = l_max / 2; t >= 1; t /= 2)
		if (pt_build_delta(b, i, i + (l + t) * d) > delta_min)
			l += t;
	int j = i + l * d;
	int delta_node = pt_build_delta(b, i, j);
	int s = 0;
	int t = l;
	do {
		t = (t + 1) >> 1;
		if (pt_build_delta(b, i, i + (s + t) * d) > delta_node)
			s += t;
	} while (t > 1);
	int gamma = i + s * d + ((d < 0) ? -1 : 0);
	int first = (i < j) ? i : j, last = (i < j) ? j : i;
	uint32_t leaf_base = b.n - 1;
	uint32_t left = (first == gamma) ? leaf_base + (uint32_t) gamma : (uint32_t) gamma;
	uint32_t right = (last == gamma + 1) ? leaf_base + (uint32_t) (gamma + 1) : (uint32_t) (gamma + 1);
	b.left[i] = left;
	b.right[i] = right;
	b.parent[left] = index;
	b.parent[right] = index;
	if (i == 0)
		b.parent[0] = 0xffffffffu;
}

// Step 4: called once per leaf (sorted position k). Writes the leaf box, then
// climbs: the second thread to reach an inner node merges its children's
// boxes. arrive(flag_ptr) must be an atomic increment that returns the old
// value and orders this thread's earlier box stores before it and its later
// box loads after it (release/acquire at device scope).
PT_FN void pt_build_leaf_box(const PtBuild& b, float pad, uint32_t k) {
	float* box = &b.box[6 * (uint64_t) (b.n - 1 + k)];
	float lo[3], hi[3];
	pt_tri_box(&b.verts[9 * (uint64_t) b.order[k]], pad, lo, hi);
	for (int j = 0; j != 3; ++j) { box[j] = lo[j]; box[3 + j] = hi[j]; }
}

// The box of an inner node from the boxes of its children
PT_FN void pt_build_inner_box(const PtBuild& b, uint32_t node) {
	const float* l = &b.box[6 * (uint64_t) b.left[node]];
	const float* r = &b.box[6 * (uint64_t) b.right[node]];
	float* out = &b.box[6 * (uint64_t) node];
	for (int j = 0; j != 3; ++j) {
		out[j] = fminf(l[j], r[j]);
		out[3 + j] = fmaxf(l[j + 3], r[j + 3]);
	}
}

template <typename Arrive>
PT_FN void pt_build_refit(const PtBuild& b, float pad, uint32_t k, Arrive arrive) {
	pt_build_leaf_box(b, pad, k);
	if (b.n < 2)
		return;
	uint32_t parent = b.parent[b.n - 1 + k];
	while (parent != 0xffffffffu) {
		if (arrive(&b.flags[parent]) == 0)
			return;
		pt_build_inner_box(b, parent);
		parent = b.parent[parent];
	}
}

// A bottom-up pass over the finished binary tree: called once per leaf
// (sorted position k); visit(node) runs for every inner node after both of its
// children, in the thread that reaches the node second. arrive() as above;
// the flags must be zero before the pass.
template <typename Arrive, typename Visit>
PT_FN void pt_build_climb(const PtBuild& b, uint32_t k, Arrive arrive, Visit visit) {
	if (b.n < 2)
		return;
	uint32_t parent = b.parent[b.n - 1 + k];
	while (parent != 0xffffffffu) {
		if (arrive(&b.flags[parent]) == 0)
			return;
		visit(parent);
		parent = b.parent[parent];
	}
}

// Surface area of a box given as 6 floats (only compared, never output)
PT_FN float pt_box_area(const float* box) {
	float dx = box[3] - box[0], dy = box[4] - box[1], dz = box[5] - box[2];
	return (dx * dy + dy * dz) + dz * dx;
}

// Step 4b: treelet restructuring (Karras and Aila, "Fast parallel construction
// of high-quality bounding volume hierarchies", HPG 2013). The radix tree
// splits space in the middle; its topology is improved bottom-up: at every
// inner node with at least PT_TREELET_GAMMA triangles below it, the node and
// its largest descendants form a treelet with up to eight leaves (subtrees
// that stay as they are), and the binary tree over those leaves with the
// smallest surface area cost, found by dynamic programming over all subsets,
// replaces the treelet's inner nodes. Results of rays do not depend on the
// tree, only the number of nodes they visit.
// Leaves per treelet, at most 8 (a partition is stored in one byte). Measured
// on MI355X against 7: the same on the Cornell box and the hall, +4.6 % on the
// generated city, for twice the time of the pass (5 and 6 leaves: -4 % on
// the Cornell box)
#ifndef PT_TREELET_LEAVES
#define PT_TREELET_LEAVES 8
#endif
#ifndef PT_TREELET_GAMMA
#define PT_TREELET_GAMMA 7
#endif
// One pass gives most of what treelets give: measured on MI355X, 8.08 -> 8.89
// Grays/s on the Cornell box and 3.21 -> 3.33 on a hall with 32 lights; two to
// five passes change that by under one per cent and cost build time
#ifndef PT_TREELET_PASSES
#define PT_TREELET_PASSES 1
#endif
#ifndef PT_SAH_NODE_COST
#define PT_SAH_NODE_COST 1.2f
#endif
#define PT_SAH_TRIANGLE_COST 1.0f

struct PtTreeletScratch {
	float area[1 << PT_TREELET_LEAVES];
	float cost[1 << PT_TREELET_LEAVES];
	uint8_t partition[1 << PT_TREELET_LEAVES];
};

PT_FN float pt_build_subtree_cost(const PtBuild& b, uint32_t node) {
	if (node >= b.n - 1)
		return PT_SAH_TRIANGLE_COST * pt_box_area(&b.box[6 * (uint64_t) node]);
	return b.sah_cost[node];
}

PT_FN void pt_build_union_of(const PtBuild& b, const uint32_t* leaves, uint32_t subset, float out[6]) {
	for (int a = 0; a != 3; ++a) { out[a] = INFINITY; out[3 + a] = -INFINITY; }
	for (uint32_t i = 0; subset != 0; ++i, subset >>= 1) {
		if (subset & 1u) {
			const float* box = &b.box[6 * (uint64_t) leaves[i]];
			for (int a = 0; a != 3; ++a) { out[a] = fminf(out[a], box[a]); out[3 + a] = fmaxf(out[3 + a], box[3 + a]); }
		}
	}
}

// Size and cost of an inner node from its children, then the treelet rooted
// at it. Called bottom-up, after both children are final.
PT_FN void pt_build_treelet(const PtBuild& b, uint32_t root, PtTreeletScratch& t) {
	uint32_t left = b.left[root], right = b.right[root];
	uint32_t size = ((left >= b.n - 1) ? 1u : b.size[left]) + ((right >= b.n - 1) ? 1u : b.size[right]);
	b.size[root] = size;
	float root_area = pt_box_area(&b.box[6 * (uint64_t) root]);
	b.sah_cost[root] = PT_SAH_NODE_COST * root_area + pt_build_subtree_cost(b, left) + pt_build_subtree_cost(b, right);
	if (size < PT_TREELET_GAMMA)
		return;
	// Form the treelet: open the leaf with the largest area until there are enough
	uint32_t leaves[PT_TREELET_LEAVES], inner[PT_TREELET_LEAVES - 1];
	uint32_t leaf_count = 2, inner_count = 1;
	leaves[0] = left; leaves[1] = right; inner[0] = root;
	while (leaf_count < PT_TREELET_LEAVES) {
		int best = -1;
		float best_area = -1.0f;
		for (uint32_t i = 0; i != leaf_count; ++i) {
			if (leaves[i] >= b.n - 1)
				continue;
			float area = pt_box_area(&b.box[6 * (uint64_t) leaves[i]]);
			if (area > best_area) { best_area = area; best = (int) i; }
		}
		if (best < 0)
			break;
		uint32_t open = leaves[best];
		inner[inner_count++] = open;
		leaves[best] = b.left[open];
		leaves[leaf_count++] = b.right[open];
	}
	if (leaf_count < 3)
		return;
	const uint32_t full = (1u << leaf_count) - 1u;
	// Surface area of every subset of the leaves
	for (uint32_t subset = 1; subset <= full; ++subset) {
		float box[6];
		pt_build_union_of(b, leaves, subset, box);
		t.area[subset] = pt_box_area(box);
	}
	// Cheapest binary tree over every subset. In numerical order every proper
	// subset of a set comes before the set, so its cost is known by then.
	for (uint32_t i = 0; i != leaf_count; ++i)
		t.cost[1u << i] = pt_build_subtree_cost(b, leaves[i]);
	for (uint32_t subset = 3; subset <= full; ++subset) {
		// Partitions (p, subset ^ p): p runs over the subsets of the rest that
		// leave out the lowest leaf, so every split is seen once
		uint32_t rest = subset & (subset - 1u);
		if (rest == 0)
			continue;
		float best = INFINITY;
		uint32_t best_p = rest;
		for (uint32_t p = rest; p != 0; p = (p - 1u) & rest) {
			float c = t.cost[p] + t.cost[subset ^ p];
			if (c < best) { best = c; best_p = p; }
		}
		t.cost[subset] = PT_SAH_NODE_COST * t.area[subset] + best;
		t.partition[subset] = (uint8_t) best_p;
	}
	// Keep the present topology unless the best one is cheaper
	if (!(t.cost[full] < b.sah_cost[root]))
		return;
	// Rebuild the treelet's inner nodes after the best partitions, reusing
	// their indices; the root keeps its own
	uint32_t next_inner = 1;
	uint32_t stack_subset[PT_TREELET_LEAVES], stack_node[PT_TREELET_LEAVES];
	uint32_t sp = 0;
	stack_subset[sp] = full; stack_node[sp++] = root;
	while (sp != 0) {
		--sp;
		uint32_t subset = stack_subset[sp], node = stack_node[sp];
		uint32_t halves[2] = { t.partition[subset], subset ^ (uint32_t) t.partition[subset] };
		uint32_t children[2];
		for (int h = 0; h != 2; ++h) {
			if ((halves[h] & (halves[h] - 1u)) == 0) {
				uint32_t index = 0;
				while (!((halves[h] >> index) & 1u)) ++index;
				children[h] = leaves[index];
			}
			else {
				children[h] = inner[next_inner++];
				stack_subset[sp] = halves[h]; stack_node[sp++] = children[h];
			}
			b.parent[children[h]] = node;
		}
		b.left[node] = children[0];
		b.right[node] = children[1];
		float box[6];
		pt_build_union_of(b, leaves, subset, box);
		for (int a = 0; a != 6; ++a)
			b.box[6 * (uint64_t) node + a] = box[a];
		b.sah_cost[node] = t.cost[subset];
		uint32_t triangles = 0;
		for (uint32_t i = 0; i != leaf_count; ++i)
			if ((subset >> i) & 1u)
				triangles += (leaves[i] >= b.n - 1) ? 1u : b.size[leaves[i]];
		b.size[node] = triangles;
	}
}

// Which binary nodes become wide nodes is chosen to minimise the summed
// surface area of all wide nodes, the expected number of node visits of a
// random ray (the dynamic programme of Ylitie, Karras and Laine, "Efficient
// incoherent ray traversal on GPUs through compressed wide BVHs", HPG 2017,
// section 3.1, for leaves of one triangle). For the subtree below binary node
// n and 1 <= j <= 8,
//   cost(n, 1) = area(n) + split(n, 8)                (n becomes a wide node)
//   split(n, j) = min over 0 < k < j of cost(left, k) + cost(right, j - k)
//   cost(n, j) = min(cost(n, j - 1), split(n, j))      (n dissolves into its parent's slots)
// and a triangle costs nothing. Called bottom-up, after both children.
PT_FN void pt_build_wide_cost(const PtBuild& b, uint32_t node) {
	uint32_t left = b.left[node], right = b.right[node];
	float lc[8], rc[8];
	for (int j = 0; j != 8; ++j) {
		lc[j] = (left >= b.n - 1) ? 0.0f : b.wide_cost[8 * (uint64_t) left + j];
		rc[j] = (right >= b.n - 1) ? 0.0f : b.wide_cost[8 * (uint64_t) right + j];
	}
	float split[9];
	uint8_t split_k[9];
	for (int j = 2; j <= 8; ++j) {
		float best = INFINITY;
		int best_k = 1;
		for (int k = 1; k < j; ++k) {
			float c = lc[k - 1] + rc[j - k - 1];
			if (c < best) { best = c; best_k = k; }
		}
		split[j] = best;
		split_k[j] = (uint8_t) best_k;
	}
	float* cost = &b.wide_cost[8 * (uint64_t) node];
	uint8_t* choice = &b.wide_choice[8 * (uint64_t) node];
	cost[0] = pt_box_area(&b.box[6 * (uint64_t) node]) + split[8];
	choice[0] = 0;
	for (int j = 2; j <= 8; ++j) {
		if (split[j] < cost[j - 2]) { cost[j - 1] = split[j]; choice[j - 1] = split_k[j]; }
		else { cost[j - 1] = cost[j - 2]; choice[j - 1] = choice[j - 2]; }
	}
}

// The children of the wide node rooted at binary node root, by the choices of
// pt_build_wide_cost(); returns their number (2 to 8)
PT_FN uint32_t pt_build_wide_children(const PtBuild& b, uint32_t root, uint32_t children[8]) {
	uint32_t count = 0;
	uint32_t stack_node[8], stack_slots[8];
	uint32_t sp = 0;
	// The root itself always splits into 8 slots
	uint8_t k = 0;
	{
		// split(root, 8): recompute the k that pt_build_wide_cost() found
		uint32_t left = b.left[root], right = b.right[root];
		float best = INFINITY;
		for (int t = 1; t < 8; ++t) {
			float lc = (left >= b.n - 1) ? 0.0f : b.wide_cost[8 * (uint64_t) left + t - 1];
			float rc = (right >= b.n - 1) ? 0.0f : b.wide_cost[8 * (uint64_t) right + 8 - t - 1];
			if (lc + rc < best) { best = lc + rc; k = (uint8_t) t; }
		}
		stack_node[sp] = right; stack_slots[sp++] = 8u - k;
		stack_node[sp] = left; stack_slots[sp++] = k;
	}
	while (sp != 0) {
		--sp;
		uint32_t node = stack_node[sp], slots = stack_slots[sp];
		uint8_t choice = (node >= b.n - 1) ? 0 : b.wide_choice[8 * (uint64_t) node + slots - 1];
		if (choice == 0)
			children[count++] = node;
		else {
			stack_node[sp] = b.right[node]; stack_slots[sp++] = slots - choice;
			stack_node[sp] = b.left[node]; stack_slots[sp++] = choice;
		}
	}
	return count;
}

// Power of two with biased exponent e, and the smallest e such that
// extent / 2^(e-127) rounds up to at most 255
PT_FN float pt_exp2_biased(uint32_t e) { return pt_as_float(e << 23); }
PT_FN uint32_t pt_quantization_exponent(float extent) {
	// extent / 255 <= 2^(e-127); start from the exponent of extent / 255
	float target = extent * (1.0f / 255.0f);
	uint32_t e = (pt_as_uint(target) >> 23) & 0xffu;
	if (e < 1) e = 1;
	// round up to the next power of two unless target is one already
	if ((pt_as_uint(target) & 0x007fffffu) != 0 || e == 1) e += 1;
	while (e < 254 && ceilf(extent / pt_exp2_biased(e)) > 255.0f) ++e;
	return e;
}

// Step 5: builds wide node wide_index from the binary subtree rooted at
// b.task_binary[wide_index]. alloc_nodes(count) / alloc_tris(count) reserve
// consecutive wide nodes / triangle slots and return the first index.
template <typename AllocNodes, typename AllocTris>
PT_FN void pt_build_collapse(const PtBuild& b, float pad, uint32_t wide_index, AllocNodes alloc_nodes, AllocTris alloc_tris) {
	uint32_t children[8];
	uint32_t count;
	uint32_t root = b.task_binary[wide_index];
	if (root >= b.n - 1) {
		// A scene of one triangle: the root node holds it
		children[0] = root;
		count = 1;
	}
	else
		count = pt_build_wide_children(b, root, children);
	// Node frame
	float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (uint32_t c = 0; c != count; ++c) {
		const float* box = &b.box[6 * (uint64_t) children[c]];
		for (int j = 0; j != 3; ++j) {
			lo[j] = fminf(lo[j], box[j]);
			hi[j] = fmaxf(hi[j], box[3 + j]);
		}
	}
	PtNode node;
	memset(&node, 0, sizeof(node));
	float scale[3];
	for (int j = 0; j != 3; ++j) {
		node.origin[j] = lo[j];
		uint32_t e = pt_quantization_exponent(hi[j] - lo[j]);
		node.exponent[j] = (uint8_t) e;
		scale[j] = pt_exp2_biased(e);
	}
	// Greedy assignment of children to slots: slot s prefers children lying
	// towards the low side of every axis whose bit is clear in s, so that
	// slot ^ (7 - ray octant) enumerates the children roughly front to back
	float center[3] = { 0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2]) };
	int slot_of[8], child_in_slot[8];
	for (int i = 0; i != 8; ++i) { slot_of[i] = -1; child_in_slot[i] = -1; }
	for (uint32_t round = 0; round != count; ++round) {
		float best_cost = -INFINITY;
		int best_c = -1, best_s = -1;
		for (uint32_t c = 0; c != count; ++c) {
			if (slot_of[c] >= 0)
				continue;
			const float* box = &b.box[6 * (uint64_t) children[c]];
			float dx = 0.5f * (box[0] + box[3]) - center[0];
			float dy = 0.5f * (box[1] + box[4]) - center[1];
			float dz = 0.5f * (box[2] + box[5]) - center[2];
			for (int s = 0; s != 8; ++s) {
				if (child_in_slot[s] >= 0)
					continue;
				float cost = (((s & 1) ? dx : -dx) + ((s & 2) ? dy : -dy)) + ((s & 4) ? dz : -dz);
				if (cost > best_cost) {
					best_cost = cost;
					best_c = (int) c;
					best_s = s;
				}
			}
		}
		slot_of[best_c] = best_s;
		child_in_slot[best_s] = best_c;
	}
	// Count inner children and triangles, reserve their storage
	uint32_t inner_count = 0, tri_count = 0;
	for (int s = 0; s != 8; ++s) {
		if (child_in_slot[s] < 0)
			continue;
		if (children[child_in_slot[s]] >= b.n - 1)
			++tri_count;
		else {
			++inner_count;
			node.imask |= (uint8_t) (1u << s);
		}
	}
	node.child_base = inner_count ? alloc_nodes(inner_count) : 0;
	node.tri_base = tri_count ? alloc_tris(tri_count) : 0;
	uint32_t inner_offset = 0, tri_offset = 0;
	for (int s = 0; s != 8; ++s) {
		if (child_in_slot[s] < 0)
			continue;
		uint32_t child = children[child_in_slot[s]];
		const float* box = &b.box[6 * (uint64_t) child];
		for (int j = 0; j != 3; ++j) {
			float qlo = floorf((box[j] - node.origin[j]) / scale[j]);
			float qhi = ceilf((box[3 + j] - node.origin[j]) / scale[j]);
			node.qlo[j][s] = (uint8_t) fminf(fmaxf(qlo, 0.0f), 255.0f);
			node.qhi[j][s] = (uint8_t) fminf(fmaxf(qhi, 0.0f), 255.0f);
		}
		if (child >= b.n - 1) {
			// A leaf child is one triangle: every triangle has a box of its own
			// in its parent, so few triangles are tested in vain, and a lane
			// never holds a backlog of triangle tests
			uint32_t tri = b.order[child - (b.n - 1)];
			PtTri out;
			const float* v = &b.verts[9 * (uint64_t) tri];
			for (int j = 0; j != 3; ++j) { out.v0[j] = v[j]; out.v1[j] = v[3 + j]; out.v2[j] = v[6 + j]; }
			out.prim = tri;
			out.pad[0] = out.pad[1] = 0;
			b.tris[node.tri_base + tri_offset] = out;
			node.leafmask |= (uint8_t) (1u << s);
			++tri_offset;
		}
		else {
			b.task_binary[node.child_base + inner_offset] = child;
			++inner_offset;
		}
	}
	(void) pad;
	b.nodes[wide_index] = node;
}
