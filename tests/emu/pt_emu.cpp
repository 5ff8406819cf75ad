// Host-side execution of the per-thread kernel bodies of path_tracer_amd/csrc
// for unit tests on machines without a GPU. TEST INFRASTRUCTURE: this file is
// never part of libpt_hip.so and nothing in the product calls it. It runs the
// same BVH build steps, traversal and shading functions that the gfx950
// kernels run, one "thread" after the other, so that their logic can be
// checked against the CPU oracle before any GPU time is spent.
#include "../../path_tracer_amd/csrc/pt_common.h"
#include "../../path_tracer_amd/csrc/pt_bvh_build.h"
#include "../../path_tracer_amd/csrc/pt_trace.h"
#include "../../path_tracer_amd/csrc/pt_shade.h"
#include "../../path_tracer_amd/csrc/pt_light_tree.h"
#include <algorithm>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct EmuStack {
	u32x2 data[256];
	uint32_t sp = 0, max_sp = 0;
	void push(u32x2 e) { data[sp++] = e; if (sp > max_sp) max_sp = sp; }
	u32x2 pop() { return data[--sp]; }
	bool empty() const { return sp == 0; }
};

struct EmuTexture { uint32_t width, height, format, mip_count; const uint8_t* payload; const uint64_t* mip_offsets; };

struct EmuScene {
	uint32_t n;
	std::vector<u32x2> positions, normals_uvs;
	std::vector<uint8_t> materials;
	std::vector<PtTexture> textures;
	std::vector<u32x2> texture_table;
	std::vector<PtShadeTri> shade_tris;
	std::vector<uint8_t> blob;
	std::vector<float> srgb;
	std::vector<f32x4> lights;
	PtLightTree light_tree;
	PtLightGrid light_grid;
	std::vector<PtNode> nodes;
	std::vector<PtTri> tris;
	uint32_t node_count, depth, max_stack;
	double binary_cost = 0.0;
	uint32_t emission_material_index;
	PtSceneView view() const {
		PtSceneView v;
		v.nodes = nodes.data(); v.tris = tris.data(); v.triangle_count = n;
		v.shade_tris = shade_tris.data(); v.materials = materials.data();
		v.texture_table = texture_table.data(); v.texture_blob = blob.data(); v.srgb_to_linear = srgb.data();
		v.lights = lights.data(); v.light_count = (uint32_t) lights.size(); v.emission_material_index = emission_material_index;
		v.light_nodes = light_tree.nodes.data(); v.tree_lights = light_tree.lights.data(); v.light_leaf_count = light_tree.leaf_count;
		// PT_EMU_NO_LIGHT_GRID=1: the walk of the light tree instead of the grid, for comparing the two
		bool use_grid = light_grid.valid && !getenv("PT_EMU_NO_LIGHT_GRID");
		v.light_grid_cells = use_grid ? light_grid.cells.data() : NULL;
		v.light_grid_lights = use_grid ? light_grid.lights.data() : NULL;
		v.light_grid_leaves = use_grid ? light_grid.leaves.data() : NULL;
		for (int a = 0; a != 3; ++a) {
			v.light_grid_lo[a] = light_grid.lo[a]; v.light_grid_cell[a] = light_grid.cell[a];
			v.light_grid_inv_cell[a] = light_grid.inv_cell[a]; v.light_grid_dim[a] = light_grid.dim[a];
		}
		v.light_grid_reach = light_grid.reach;
		return v;
	}
};

// Top-down builder with a binned surface area heuristic, used as a yardstick
// for the quality of the trees that the GPU builder makes. Fills the same
// arrays as the radix tree and refit steps.
struct SahBuilder {
	PtBuild& b;
	float pad;
	uint32_t next_inner = 0;
	std::vector<float> centroid;
	SahBuilder(PtBuild& build, float padding) : b(build), pad(padding), centroid(3 * (size_t) build.n) {
		for (uint32_t t = 0; t != b.n; ++t) {
			float lo[3], hi[3];
			pt_tri_box(&b.verts[9 * (size_t) t], 0.0f, lo, hi);
			for (int j = 0; j != 3; ++j) centroid[3 * (size_t) t + j] = 0.5f * (lo[j] + hi[j]);
		}
		for (uint32_t k = 0; k != b.n; ++k) b.order[k] = k;
	}
	static void grow(float* box, const float* lo, const float* hi) {
		for (int j = 0; j != 3; ++j) { box[j] = std::min(box[j], lo[j]); box[3 + j] = std::max(box[3 + j], hi[j]); }
	}
	static float area(const float* box) { return (box[3] < box[0]) ? 0.0f : pt_box_area(box); }
	uint32_t build(uint32_t begin, uint32_t end, uint32_t parent) {
		if (end - begin == 1) {
			uint32_t node = b.n - 1 + begin;
			float lo[3], hi[3];
			pt_tri_box(&b.verts[9 * (size_t) b.order[begin]], pad, lo, hi);
			for (int j = 0; j != 3; ++j) { b.box[6 * (size_t) node + j] = lo[j]; b.box[6 * (size_t) node + 3 + j] = hi[j]; }
			b.parent[node] = parent;
			return node;
		}
		uint32_t node = next_inner++;
		b.parent[node] = parent;
		float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
		for (uint32_t k = begin; k != end; ++k)
			for (int j = 0; j != 3; ++j) { clo[j] = std::min(clo[j], centroid[3 * (size_t) b.order[k] + j]); chi[j] = std::max(chi[j], centroid[3 * (size_t) b.order[k] + j]); }
		const int BINS = 32;
		float best_cost = INFINITY;
		int best_axis = -1, best_bin = 0;
		for (int axis = 0; axis != 3; ++axis) {
			float extent = chi[axis] - clo[axis];
			if (!(extent > 0.0f)) continue;
			float boxes[BINS][6];
			uint32_t counts[BINS];
			for (int i = 0; i != BINS; ++i) { counts[i] = 0; for (int j = 0; j != 3; ++j) { boxes[i][j] = INFINITY; boxes[i][3 + j] = -INFINITY; } }
			for (uint32_t k = begin; k != end; ++k) {
				uint32_t t = b.order[k];
				int bin = std::min(BINS - 1, (int) ((centroid[3 * (size_t) t + axis] - clo[axis]) / extent * BINS));
				float lo[3], hi[3];
				pt_tri_box(&b.verts[9 * (size_t) t], 0.0f, lo, hi);
				grow(boxes[bin], lo, hi);
				++counts[bin];
			}
			float right_area[BINS];
			uint32_t right_count[BINS];
			float acc[6] = { INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY };
			uint32_t count = 0;
			for (int i = BINS - 1; i > 0; --i) { grow(acc, boxes[i], boxes[i] + 3); count += counts[i]; right_area[i] = area(acc); right_count[i] = count; }
			float left[6] = { INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY };
			count = 0;
			for (int i = 1; i != BINS; ++i) {
				grow(left, boxes[i - 1], boxes[i - 1] + 3);
				count += counts[i - 1];
				if (count == 0 || right_count[i] == 0) continue;
				float cost = area(left) * (float) count + right_area[i] * (float) right_count[i];
				if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = i; }
			}
		}
		uint32_t mid;
		if (best_axis < 0)
			mid = begin + (end - begin) / 2;
		else {
			float extent = chi[best_axis] - clo[best_axis];
			mid = (uint32_t) (std::partition(b.order + begin, b.order + end, [&](uint32_t t) {
				return std::min(BINS - 1, (int) ((centroid[3 * (size_t) t + best_axis] - clo[best_axis]) / extent * BINS)) < best_bin; }) - b.order);
			if (mid == begin || mid == end) mid = begin + (end - begin) / 2;
		}
		uint32_t l = build(begin, mid, node), r = build(mid, end, node);
		b.left[node] = l;
		b.right[node] = r;
		const float* lb = &b.box[6 * (size_t) l]; const float* rb = &b.box[6 * (size_t) r];
		for (int j = 0; j != 3; ++j) { b.box[6 * (size_t) node + j] = std::min(lb[j], rb[j]); b.box[6 * (size_t) node + 3 + j] = std::max(lb[3 + j], rb[3 + j]); }
		return node;
	}
};

// Surface area cost of the binary tree: sum over inner nodes of area / root area
static double binary_tree_cost(const PtBuild& b) {
	if (b.n < 2) return 0.0;
	double total = 0.0;
	double largest = 0.0;
	for (uint32_t i = 0; i + 1 < b.n; ++i) { double a = pt_box_area(&b.box[6 * (size_t) i]); total += a; largest = std::max(largest, a); }
	return total / largest;
}

extern "C" {

EmuScene* emu_create_scene(uint64_t material_count, uint64_t triangle_count, const float* factor, const float* summand,
	const uint32_t* positions, const uint16_t* normals_uvs, const uint8_t* materials, const EmuTexture* textures, uint32_t texture_count)
{
	EmuScene* s = new EmuScene();
	uint32_t n = s->n = (uint32_t) triangle_count;
	s->positions.resize(3 * (size_t) n + 1);
	s->normals_uvs.resize(3 * (size_t) n + 1);
	memcpy(s->positions.data(), positions, 24 * (size_t) n);
	memcpy(s->normals_uvs.data(), normals_uvs, 24 * (size_t) n);
	s->materials.assign(materials, materials + n);
	s->materials.push_back(0);
	s->srgb.resize(256);
	for (int i = 0; i != 256; ++i) {
		double c = i / 255.0;
		s->srgb[i] = (float) ((c <= 0.04045) ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
	}
	s->textures.resize(texture_count ? texture_count : 1);
	// As pt_create_scene(): every mip level of every texture decoded into tiled 8-bit texels, level after level
	for (uint32_t i = 0; i != texture_count; ++i) {
		const EmuTexture& t = textures[i];
		PtTexture& d = s->textures[i];
		d.width = t.width; d.height = t.height; d.format = t.format; d.mip_count = t.mip_count;
		uint32_t block = (t.format == 141) ? 16 : 8;
		for (uint32_t m = 0; m != t.mip_count; ++m) {
			uint32_t mw = t.width >> m, mh = t.height >> m;
			size_t offset = s->blob.size();
			d.mip_offsets[m] = offset;
			s->blob.resize(offset + (size_t) mw * mh * 4);
			uint32_t log_w = 0;
			while ((1u << log_w) < mw) ++log_w;
			std::vector<uint8_t> aligned((size_t) (mw / 4) * (mh / 4) * block + 16);
			memcpy(aligned.data(), t.payload + t.mip_offsets[m], aligned.size() - 16);
			uint32_t* texels = (uint32_t*) &s->blob[offset];
			for (uint32_t y = 0; y != mh; ++y)
				for (uint32_t x = 0; x != mw; ++x)
					texels[pt_texel_index(log_w, x, y)] = pt_decode_texel(aligned.data(), log_w, t.format, x, y);
		}
	}
	s->blob.resize(s->blob.size() + 16);
	s->texture_table.resize(s->textures.size());
	for (uint32_t i = 0; i != texture_count; ++i)
		s->texture_table[i] = pt_pack_texture(s->textures[i]);
	// As k_build_shade_tris() of pt_hip.hip
	s->shade_tris.resize((size_t) n + 1);
	for (uint32_t i = 0; i != n; ++i) {
		PtShadeTri& t = s->shade_tris[i];
		for (int v = 0; v != 3; ++v) { t.positions[v] = s->positions[3 * (size_t) i + v]; t.normals_uvs[v] = s->normals_uvs[3 * (size_t) i + v]; }
		t.material = materials[i];
		t.pad[0] = t.pad[1] = t.pad[2] = 0;
	}
	s->emission_material_index = 0;
	s->nodes.resize(n ? n : 1);
	s->tris.resize(n ? n : 1);
	s->node_count = 1; s->depth = 1; s->max_stack = 0;
	if (n == 0) {
		memset(&s->nodes[0], 0, sizeof(PtNode));
		return s;
	}
	// The build, step by step as in pt_build_bvh() of pt_hip.hip
	PtBuild b;
	memset(&b, 0, sizeof(b));
	b.n = n;
	b.positions = s->positions.data();
	memcpy(b.factor, factor, 12); memcpy(b.summand, summand, 12);
	std::vector<float> verts(9 * (size_t) n), box(12 * (size_t) n);
	std::vector<uint32_t> bounds(6), order(n), left(n), right(n), parent(2 * (size_t) n), flags(n, 0), task_binary(n), counters(2);
	std::vector<float> wide_cost(8 * (size_t) n);
	std::vector<uint8_t> wide_choice(8 * (size_t) n);
	b.wide_cost = wide_cost.data(); b.wide_choice = wide_choice.data();
	std::vector<uint64_t> keys(n);
	b.verts = verts.data(); b.bounds = bounds.data(); b.keys = keys.data(); b.order = order.data();
	b.left = left.data(); b.right = right.data(); b.parent = parent.data(); 
	b.box = box.data(); b.flags = flags.data(); b.nodes = s->nodes.data(); b.tris = s->tris.data(); b.task_binary = task_binary.data(); b.counters = counters.data();
	for (int j = 0; j != 3; ++j) { bounds[j] = 0xffffffffu; bounds[3 + j] = 0; }
	for (uint32_t i = 0; i != n; ++i) {
		float lo[3], hi[3];
		pt_build_dequantize(b, i, lo, hi);
		for (int j = 0; j != 3; ++j) {
			bounds[j] = std::min(bounds[j], pt_float_to_key(lo[j]));
			bounds[3 + j] = std::max(bounds[3 + j], pt_float_to_key(hi[j]));
		}
	}
	PtSceneBounds sb = pt_build_get_bounds(b.bounds);
	for (uint32_t i = 0; i != n; ++i) { keys[i] = pt_build_morton(b, sb, i); order[i] = i; }
	std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return keys[x] < keys[y]; });
	{
		std::vector<uint64_t> sorted(n);
		for (uint32_t i = 0; i != n; ++i) sorted[i] = keys[order[i]];
		keys.swap(sorted);
		b.keys = keys.data();
	}
	for (uint32_t i = 0; i + 1 < n; ++i) pt_build_radix_tree(b, i);
	for (uint32_t i = 0; i != n; ++i)
		pt_build_refit(b, sb.pad, i, [](uint32_t* flag) { return (*flag)++; });
	// Treelet passes and the wide-collapse costs, as in pt_build_bvh()
	std::vector<uint32_t> size(n);
	std::vector<float> sah_cost(n);
	b.size = size.data(); b.sah_cost = sah_cost.data();
	const char* passes_text = getenv("PT_EMU_TREELET_PASSES");
	int passes = passes_text ? atoi(passes_text) : PT_TREELET_PASSES;
	for (int pass = 0; pass != passes; ++pass) {
		std::fill(flags.begin(), flags.end(), 0u);
		PtTreeletScratch scratch;
		for (uint32_t i = 0; i != n; ++i)
			pt_build_climb(b, i, [](uint32_t* flag) { return (*flag)++; }, [&](uint32_t node) { pt_build_treelet(b, node, scratch); });
	}
	std::fill(flags.begin(), flags.end(), 0u);
	for (uint32_t i = 0; i != n; ++i)
		pt_build_climb(b, i, [](uint32_t* flag) { return (*flag)++; }, [&](uint32_t node) { pt_build_wide_cost(b, node); });
	const char* builder = getenv("PT_EMU_BUILDER");
	uint32_t root = 0;
	if (builder && !strcmp(builder, "sah") && n > 1) {
		SahBuilder sah(b, sb.pad);
		sah.build(0, n, 0xffffffffu);
	}
	s->binary_cost = binary_tree_cost(b);
	if (builder && !strcmp(builder, "sah") && n > 1) {
		// Wide-collapse costs for the yardstick tree, children before parents
		std::vector<uint32_t> walk, post;
		walk.push_back(root);
		while (!walk.empty()) {
			uint32_t node = walk.back();
			walk.pop_back();
			if (node >= n - 1) continue;
			post.push_back(node);
			walk.push_back(b.left[node]);
			walk.push_back(b.right[node]);
		}
		for (size_t i = post.size(); i-- != 0;)
			pt_build_wide_cost(b, post[i]);
	}
	counters[0] = 1; counters[1] = 0; task_binary[0] = root;
	uint32_t level_begin = 0, level_end = 1, depth = 0;
	while (level_begin < level_end) {
		for (uint32_t i = level_begin; i != level_end; ++i)
			pt_build_collapse(b, sb.pad, i,
				[&](uint32_t c) { uint32_t r = counters[0]; counters[0] += c; return r; },
				[&](uint32_t c) { uint32_t r = counters[1]; counters[1] += c; return r; });
		level_begin = level_end;
		level_end = counters[0];
		++depth;
	}
	s->node_count = level_end;
	s->depth = depth;
	if (counters[1] != n)
		printf("emu: collapse wrote %u of %u triangles\n", counters[1], n);
	return s;
}

void emu_destroy_scene(EmuScene* s) { delete s; }

void emu_set_lights(EmuScene* s, const float* lights, uint32_t count) {
	s->lights.resize(count);
	if (count) memcpy(s->lights.data(), lights, 16 * (size_t) count);
	pt_build_light_tree(s->light_tree, lights, count);
	const char* per_cell = getenv("PT_EMU_LIGHTS_PER_CELL");
	pt_build_light_grid(s->light_grid, s->light_tree, per_cell ? (float) atof(per_cell) : 0.5f);
}

// For every ray the leaves of the lights it meets, found through the grid over the lights, against a loop over all
// lights with the same test. Returns the number of rays for which the two differ (rays that meet more than four
// lights, which the grid path hands to the walk of the tree, are counted in out_overflows instead), or 0xffffffff
// if the lights have no grid.
uint32_t emu_light_grid_mismatches(EmuScene* s, const float* origins, const float* directions, uint32_t count, uint32_t* out_overflows, uint32_t* out_hits) {
	if (!s->light_grid.valid)
		return 0xffffffffu;
	PtSceneView view = s->view();
	uint32_t mismatches = 0, overflows = 0, hits = 0;
	for (uint32_t i = 0; i != count; ++i) {
		v3 origin = mk3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), dir = mk3(directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]);
		PtLeafSet set;
		set.leaf_0 = set.leaf_1 = set.leaf_2 = set.leaf_3 = 0xffffffffu;
		set.count = 0;
		set.overflow = false;
		// The same ray twice: the set keeps a leaf once
		pt_light_grid_collect_2(view, origin, dir, dir, set);
		std::vector<uint32_t> expected;
		for (uint32_t j = 0; j != s->light_tree.leaf_count; ++j) {
			f32x4 l = s->light_tree.lights[j];
			if (l.w * l.w > 0.0f && pt_ray_meets_light(l, origin, dir))
				expected.push_back(s->light_tree.leaf_count + j);
		}
		hits += (uint32_t) expected.size();
		if (set.overflow) {
			++overflows;
			mismatches += expected.size() <= 4;
			continue;
		}
		uint32_t found[4] = { set.leaf_0, set.leaf_1, set.leaf_2, set.leaf_3 };
		bool same = expected.size() == set.count;
		for (uint32_t k = 0; same && k != set.count; ++k)
			same = found[k] == expected[k];
		mismatches += !same;
	}
	if (out_overflows) *out_overflows = overflows;
	if (out_hits) *out_hits = hits;
	return mismatches;
}

// Cells per axis of the grid over the lights and the number of its entries (zeros: no grid)
void emu_light_grid_info(EmuScene* s, uint32_t* out) {
	for (int a = 0; a != 3; ++a) out[a] = s->light_grid.valid ? s->light_grid.dim[a] : 0u;
	out[3] = s->light_grid.valid ? (uint32_t) s->light_grid.leaves.size() : 0u;
}

// The densities of two directions under the light tree's sampling, three ways: 0 through the grid over the lights
// (with its fallbacks), 1 by the walk of the tree, 2 by a loop over all leaves in order (the definition)
void emu_lights_density(EmuScene* s, const float* positions, const float* normals, const float* directions_0, const float* directions_1, uint32_t count, int mode, float* out_densities) {
	PtSceneView view = s->view();
	PtLightTreeSample none;
	none.leaf = 0;
	none.probability = 0.0f;
	for (uint32_t i = 0; i != count; ++i) {
		v3 pos = mk3(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]), normal = mk3(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
		v3 d0 = mk3(directions_0[3 * i], directions_0[3 * i + 1], directions_0[3 * i + 2]), d1 = mk3(directions_1[3 * i], directions_1[3 * i + 1], directions_1[3 * i + 2]);
		float density_0 = 0.0f, density_1 = 0.0f;
		if (mode == 0)
			pt_lights_density_tree_2(view, none, pos, normal, d0, d1, &density_0, &density_1);
		else if (mode == 1)
			pt_lights_density_tree_walk(view, none, pos, normal, d0, d1, &density_0, &density_1);
		else
			for (uint32_t j = 0; j != s->light_tree.leaf_count; ++j)
				if (s->light_tree.lights[j].w * s->light_tree.lights[j].w > 0.0f)
					pt_light_leaf_density(view, none, s->light_tree.leaf_count + j, pos, normal, d0, d1, density_0, density_1);
		out_densities[2 * i] = density_0;
		out_densities[2 * i + 1] = density_1;
	}
}

// Samples a light through the tree for every vertex and returns how often the density of the sampled direction
// differs between reusing the descent's selection probability and recomputing it
uint32_t emu_light_tree_reuse_mismatches(EmuScene* s, const float* positions, const float* normals, const float* randoms, uint32_t count) {
	PtSceneView view = s->view();
	uint32_t mismatches = 0;
	for (uint32_t i = 0; i != count; ++i) {
		v3 pos = mk3(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]), normal = mk3(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
		PtLightTreeSample sample, none;
		none.leaf = 0; none.probability = 0.0f;
		float sampled_density = 0.0f;
		v2 r; r.x = randoms[2 * i]; r.y = randoms[2 * i + 1];
		v3 dir = pt_sample_lights_tree(view, &sampled_density, &sample, pos, normal, r);
		v3 other = mk3(normal.y, normal.z, normal.x);
		float a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
		pt_lights_density_tree_2(view, sample, pos, normal, dir, other, &a0, &a1);
		pt_lights_density_tree_2(view, none, pos, normal, dir, other, &b0, &b1);
		mismatches += pt_as_uint(a0) != pt_as_uint(b0) || pt_as_uint(a1) != pt_as_uint(b1);
	}
	return mismatches;
}

void emu_bvh_info(EmuScene* s, uint32_t* out) { out[0] = s->node_count; out[1] = s->depth; out[2] = s->max_stack; }
double emu_binary_tree_cost(EmuScene* s) { return s->binary_cost; }

// Number of integers x for which the reciprocal-based unorm conversions differ from the IEEE quotient
uint32_t emu_unorm_mismatches() {
	uint32_t mismatches = 0;
	for (uint32_t x = 0; x != 256; ++x) mismatches += pt_as_uint(pt_unorm8(x)) != pt_as_uint((float) x / 255.0f);
	for (uint32_t x = 0; x != 65536; ++x) mismatches += pt_as_uint(pt_unorm16(x)) != pt_as_uint((float) x / 65535.0f);
	return mismatches;
}

void emu_trace_rays(EmuScene* s, const float* origins, const float* directions, uint32_t count, int32_t* triangles, float* tuv, uint8_t* front, uint64_t* visits_out) {
	uint64_t total[2] = { 0, 0 };
	for (uint32_t i = 0; i != count; ++i) {
		EmuStack stack;
		uint32_t visits[2] = { 0, 0 };
		PtHit hit = pt_trace_closest<true>(s->nodes.data(), s->tris.data(), s->n, mk3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]),
			mk3(directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]), 1.0e-3f, 1e38f, stack, visits);
		s->max_stack = std::max(s->max_stack, stack.max_sp);
		total[0] += visits[0]; total[1] += visits[1];
		triangles[i] = (int32_t) hit.prim;
		tuv[3 * i] = hit.t; tuv[3 * i + 1] = hit.u; tuv[3 * i + 2] = hit.v;
		front[i] = hit.front;
	}
	if (visits_out) { visits_out[0] = total[0]; visits_out[1] = total[1]; }
}

// The closest hit by a loop over all triangles with the traversal's own triangle test: what any BVH must return
void emu_trace_rays_brute_force(EmuScene* s, const float* origins, const float* directions, uint32_t count, int32_t* triangles, float* tuv, uint8_t* front) {
	for (uint32_t i = 0; i != count; ++i) {
		v3 origin = mk3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), dir = mk3(directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]);
		PtRayShear shear = pt_ray_shear(dir);
		PtHit hit;
		hit.prim = 0xffffffffu; hit.t = 1e38f; hit.u = hit.v = 0.0f; hit.front = false;
		for (uint32_t j = 0; j != s->n; ++j) {
			const PtTri& t = s->tris[j];
			pt_intersect_triangle(hit, shear, origin, mk3(t.v0[0], t.v0[1], t.v0[2]), mk3(t.v1[0], t.v1[1], t.v1[2]), mk3(t.v2[0], t.v2[1], t.v2[2]), t.prim, 1.0e-3f);
		}
		triangles[i] = (int32_t) hit.prim;
		tuv[3 * i] = hit.t; tuv[3 * i + 1] = hit.u; tuv[3 * i + 2] = hit.v;
		front[i] = hit.front;
	}
}

void emu_primary_hits(EmuScene* s, const PtConstants* c, uint32_t frame_index, int32_t* triangles, float* barys, uint8_t* front) {
	uint32_t w = (uint32_t) c->viewport_size[0], h = (uint32_t) c->viewport_size[1];
	for (uint32_t y = 0; y != h; ++y) for (uint32_t x = 0; x != w; ++x) {
		uint32_t sx, sy;
		v3 o, d;
		float cone_width, cone_spread;
		pt_primary_ray(*c, x, y, frame_index, sx, sy, &o, &d, &cone_width, &cone_spread);
		EmuStack stack;
		PtHit hit = pt_trace_closest<false>(s->nodes.data(), s->tris.data(), s->n, o, d, 1.0e-3f, 1e38f, stack, NULL);
		size_t i = (size_t) y * w + x;
		triangles[i] = (int32_t) hit.prim;
		barys[2 * i] = hit.u; barys[2 * i + 1] = hit.v;
		front[i] = hit.front;
	}
}

void emu_primary_cones(const PtConstants* c, uint32_t frame_index, float* widths, float* spreads) {
	uint32_t w = (uint32_t) c->viewport_size[0], h = (uint32_t) c->viewport_size[1];
	for (uint32_t y = 0; y != h; ++y) for (uint32_t x = 0; x != w; ++x) {
		uint32_t sx, sy;
		v3 o, d;
		float cone_width, cone_spread;
		pt_primary_ray(*c, x, y, frame_index, sx, sy, &o, &d, &cone_width, &cone_spread);
		// What travels with the ray
		pt_unpack_cone(pt_pack_cone(cone_width, cone_spread), &widths[(size_t) y * w + x], &spreads[(size_t) y * w + x]);
	}
}

// One path per pixel and frame, vertex by vertex, with the radiance updates
// in the order in which the wavefront kernels apply them
uint64_t emu_render(EmuScene* s, const PtConstants* c, int strategy, uint32_t path_length, uint32_t emission_material_index,
	uint32_t first_frame, uint32_t frame_count, uint32_t frame_stride, int accumulate, float* accum)
{
	s->emission_material_index = emission_material_index;
	PtSceneView scene = s->view();
	uint32_t w = (uint32_t) c->viewport_size[0], h = (uint32_t) c->viewport_size[1];
	uint64_t rays = 0;
	for (uint32_t y = 0; y != h; ++y) for (uint32_t x = 0; x != w; ++x) {
		float* pixel = &accum[4 * ((size_t) y * w + x)];
		for (uint32_t f = 0; f != frame_count; ++f) {
			uint32_t sx, sy;
			v3 o, d;
			float cone_width, cone_spread;
			pt_primary_ray(*c, x, y, first_frame + f * frame_stride, sx, sy, &o, &d, &cone_width, &cone_spread);
			float cone = pt_pack_cone(cone_width, cone_spread);
			v3 throughput = mk3(1, 1, 1), nee = mk3(1, 1, 1), radiance = mk3(0, 0, 0);
			for (uint32_t k = 1; k <= path_length; ++k) {
				EmuStack stack;
				PtHit hit = pt_trace_closest<false>(scene.nodes, scene.tris, s->n, o, d, 1.0e-3f, 1e38f, stack, NULL);
				++rays;
				// A cache smaller than some tests' light counts exercises both of its branches
				float cached_importances[8];
				PtLightCache light_cache = { cached_importances, 1, 8 };
				v3 emitted;
				PtVertexResult r = pt_shade_vertex(scene, *c, light_cache, strategy, path_length, k, hit.prim, hit.u, hit.v, hit.front, o, d, cone, emitted, throughput, nee, sx, sy);
				radiance = radiance + emitted;
				if (r.shadow) {
					EmuStack sstack;
					PtHit sh = pt_trace_closest<false>(scene.nodes, scene.tris, s->n, r.next_origin, r.shadow_dir, 1.0e-3f, 1e38f, sstack, NULL);
					++rays;
					v3 emission;
					if (sh.prim == 0xffffffffu) emission = mk3(c->sky_radiance[0], c->sky_radiance[1], c->sky_radiance[2]);
					else if (scene.materials[sh.prim] == emission_material_index) emission = mk3(c->emission_material_radiance[0], c->emission_material_radiance[1], c->emission_material_radiance[2]);
					else emission = mk3(0, 0, 0);
					radiance = radiance + (r.shadow_weight * emission) * r.shadow_factor;
				}
				if (!r.continues) break;
				o = r.next_origin; d = r.next_dir; cone = r.next_cone;
			}
			if (f == 0 && !accumulate) { pixel[0] = radiance.x; pixel[1] = radiance.y; pixel[2] = radiance.z; pixel[3] = 1.0f; }
			else { pixel[0] += radiance.x; pixel[1] += radiance.y; pixel[2] += radiance.z; }
		}
	}
	return rays;
}

void emu_shading_data(EmuScene* s, const PtConstants* c, uint32_t emission_material_index, const int32_t* triangles, const float* barys, const uint8_t* front, const float* out_dirs, const float* footprints, uint32_t count, float* out) {
	s->emission_material_index = emission_material_index;
	PtSceneView scene = s->view();
	for (uint32_t i = 0; i != count; ++i) {
		PtShadingData d = pt_shading_data(scene, *c, (uint32_t) triangles[i], barys[2 * i], barys[2 * i + 1], front[i] != 0, mk3(out_dirs[3 * i], out_dirs[3 * i + 1], out_dirs[3 * i + 2]),
			mk3(0.0f, 0.0f, 0.0f), footprints ? footprints[i] : 0.0f, 0.0f);
		float* o = &out[20 * (size_t) i];
		o[0] = d.pos.x; o[1] = d.pos.y; o[2] = d.pos.z; o[3] = d.normal.x; o[4] = d.normal.y; o[5] = d.normal.z;
		o[6] = d.out_dir.x; o[7] = d.out_dir.y; o[8] = d.out_dir.z; o[9] = d.lambert_out;
		o[10] = d.emission.x; o[11] = d.emission.y; o[12] = d.emission.z;
		o[13] = d.diffuse_albedo.x; o[14] = d.diffuse_albedo.y; o[15] = d.diffuse_albedo.z;
		o[16] = d.fresnel_0.x; o[17] = d.fresnel_0.y; o[18] = d.fresnel_0.z; o[19] = d.roughness;
	}
}

// pt_evaluate_function() of the kernel bodies, evaluated serially (the numbering is PT_FUNCTION_* of include/pt_hip.h)
void emu_evaluate_function(EmuScene* s, int function, const float* shading_data, const float* inputs, uint32_t count, float* outputs) {
	PtSceneView view = s->view();
	for (uint32_t i = 0; i != count; ++i)
		pt_evaluate_function(view, function, shading_data, inputs + 4 * (size_t) i, outputs + 4 * (size_t) i);
}

void emu_sincos(float x, float* s, float* c) { pt_sincos(x, s, c); }
float emu_log(float x) { return pt_log(x); }
float emu_erfinv(float x) { return pt_erfinv(x); }

}
