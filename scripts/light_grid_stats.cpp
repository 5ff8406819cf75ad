// Host-side statistics of the walk through the grid over the lights (pt_light_grid_collect_2) for 10 000 generated
// lights in a 200 x 50 x 200 box: cell steps and light tests per vertex (two rays), and how many of them the slowest of
// 64 vertices near each other needs, which is what a wavefront waits for.
// build and run from scripts/: g++ -O2 -std=c++17 -ffp-contract=off -o /tmp/light_grid_stats light_grid_stats.cpp && /tmp/light_grid_stats [lights per cell]
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../path_tracer_amd/csrc/pt_common.h"
static unsigned long long g_steps, g_tests;
#define PT_GRID_STATS 1
#include "../path_tracer_amd/csrc/pt_shade.h"
#include "../path_tracer_amd/csrc/pt_light_tree.h"
int main(int argc, char** argv) {
	float per_cell = argc > 1 ? atof(argv[1]) : 1.0f;
	float box[3] = {200, 50, 200};
	std::mt19937 rng(4);
	std::uniform_real_distribution<float> u(0, 1);
	std::vector<float> lights(4 * 10000);
	for (int i = 0; i != 10000; ++i) { for (int a = 0; a != 3; ++a) lights[4*i+a] = u(rng) * box[a]; lights[4*i+3] = 0.1f + 0.4f * u(rng); }
	PtLightTree tree; pt_build_light_tree(tree, lights.data(), 10000);
	PtLightGrid grid; pt_build_light_grid(grid, tree, per_cell);
	PtSceneView v = {};
	v.light_nodes = tree.nodes.data(); v.tree_lights = tree.lights.data(); v.light_leaf_count = tree.leaf_count;
	v.light_grid_cells = grid.cells.data(); v.light_grid_lights = grid.lights.data(); v.light_grid_leaves = grid.leaves.data();
	for (int a = 0; a != 3; ++a) { v.light_grid_lo[a] = grid.lo[a]; v.light_grid_cell[a] = grid.cell[a]; v.light_grid_inv_cell[a] = grid.inv_cell[a]; v.light_grid_dim[a] = grid.dim[a]; }
	v.light_grid_reach = grid.reach;
	printf("dims %u %u %u entries %zu\n", grid.dim[0], grid.dim[1], grid.dim[2], grid.leaves.size());
	std::normal_distribution<float> n(0, 1);
	double sum_steps = 0, sum_tests = 0, sum_wave_max = 0; int waves = 2000;
	for (int w = 0; w != waves; ++w) {
		unsigned long long wave_max = 0;
		// A wave: shading points near each other (within 2 m), directions anywhere in the upper half space
		float base[3] = { u(rng) * box[0], u(rng) * 5.0f, u(rng) * box[2] };
		for (int l = 0; l != 64; ++l) {
			v3 o = mk3(base[0] + 2 * u(rng), base[1] + u(rng), base[2] + 2 * u(rng));
			v3 d = mk3(n(rng), fabsf(n(rng)), n(rng));
			g_steps = g_tests = 0;
			PtLeafSet set; set.leaf_0 = set.leaf_1 = set.leaf_2 = set.leaf_3 = 0xffffffffu; set.count = 0; set.overflow = false;
			pt_light_grid_collect_2(v, o, d, mk3(n(rng), fabsf(n(rng)), n(rng)), set);
			sum_steps += g_steps; sum_tests += g_tests;
			if (g_steps + g_tests > wave_max) wave_max = g_steps + g_tests;
		}
		sum_wave_max += wave_max;
	}
	printf("per vertex: steps %.1f tests %.1f; slowest of 64 vertices: steps + tests %.1f\n", sum_steps / (64.0 * waves), sum_tests / (64.0 * waves), sum_wave_max / waves);
}
