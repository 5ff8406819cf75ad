/* Headless command line driver: loads a quicksave and its scene, renders a
 * number of samples per pixel on one GPU and writes a screenshot. It covers
 * what one slide of the reference's slideshow does (src/slides.c,
 * show_slide() at src/main.c:1482: quicksave + render settings + sample count
 * -> image) without window or GUI. */
#include "pt_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static void print_usage(void) {
	printf("Usage: pt_render [-data ROOT] [-save FILE.rt_save] [-scene N] [-wWIDTH] [-hHEIGHT] [-spp N] [-path_length N]\n"
		"                 [-strategy spherical|psa|brdf|nee] [-device N] [-o OUT.png|OUT.hdr]\n"
		"ROOT is the directory holding data/ (default: the working directory, like the reference).\n");
}

int main(int argc, char** argv) {
	const char* data_root = "";
	const char* save_path = NULL;
	const char* out_path = "screenshot.png";
	int scene_override = -1, device = 0;
	uint32_t width = 1280, height = 1024, spp = 64;
	render_settings_t settings;
	init_render_settings(&settings);
	for (int i = 1; i < argc; ++i) {
		const char* arg = argv[i];
		if (!strcmp(arg, "-data") && i + 1 < argc) data_root = argv[++i];
		else if (!strcmp(arg, "-save") && i + 1 < argc) save_path = argv[++i];
		else if (!strcmp(arg, "-scene") && i + 1 < argc) scene_override = atoi(argv[++i]);
		else if (!strcmp(arg, "-spp") && i + 1 < argc) spp = (uint32_t) atoi(argv[++i]);
		else if (!strcmp(arg, "-path_length") && i + 1 < argc) settings.path_length = (uint32_t) atoi(argv[++i]);
		else if (!strcmp(arg, "-device") && i + 1 < argc) device = atoi(argv[++i]);
		else if (!strcmp(arg, "-o") && i + 1 < argc) out_path = argv[++i];
		else if (!strcmp(arg, "-strategy") && i + 1 < argc) {
			const char* name = argv[++i];
			if (!strcmp(name, "spherical")) settings.sampling_strategy = sampling_strategy_spherical;
			else if (!strcmp(name, "psa")) settings.sampling_strategy = sampling_strategy_psa;
			else if (!strcmp(name, "brdf")) settings.sampling_strategy = sampling_strategy_brdf;
			else settings.sampling_strategy = sampling_strategy_nee;
		}
		/* -wN and -hN as in the reference (src/main.c:2112-2113) */
		else if (arg[0] == '-' && arg[1] == 'w' && arg[2]) width = (uint32_t) atoi(arg + 2);
		else if (arg[0] == '-' && arg[1] == 'h' && arg[2]) height = (uint32_t) atoi(arg + 2);
		else {
			print_usage();
			return 1;
		}
	}
	scene_spec_t spec;
	init_scene_spec(&spec);
	if (scene_override >= 0)
		spec.scene_file = (scene_file_t) scene_override;
	if (!save_path) {
		const char* default_save = NULL;
		if (!get_scene_file(spec.scene_file, NULL, NULL, NULL, NULL, &default_save) && default_save) {
			size_t length = strlen(data_root) + strlen(default_save) + 2;
			char* path = malloc(length);
			snprintf(path, length, "%s%s%s", data_root, data_root[0] ? "/" : "", default_save);
			quickload(&spec, path);
			free(path);
		}
	}
	else if (quickload(&spec, save_path))
		return 1;
	lit_scene_t lit_scene;
	if (create_lit_scene(&lit_scene, device, &spec, data_root))
		return 1;
	render_targets_t targets;
	if (apply_render_settings(&lit_scene, &settings) || create_render_targets(&targets, &lit_scene, width, height)) {
		free_lit_scene(&lit_scene);
		return 1;
	}
	struct timespec begin, end;
	clock_gettime(CLOCK_MONOTONIC, &begin);
	int result = render_frames(&spec, &lit_scene, &targets, spp) || pt_synchronize(lit_scene.scene.device_scene);
	clock_gettime(CLOCK_MONOTONIC, &end);
	if (!result) {
		double seconds = (double) (end.tv_sec - begin.tv_sec) + 1.0e-9 * (double) (end.tv_nsec - begin.tv_nsec);
		pt_statistics_t stats;
		if (!pt_get_statistics(lit_scene.scene.device_scene, &stats))
			printf("Rendered %u spp at %ux%u in %.3f s: %.1f Mrays/s, %.1f Msamples/s.\n", spp, width, height, seconds,
				(double) (stats.extension_rays + stats.shadow_rays) / seconds * 1.0e-6, (double) width * height * spp / seconds * 1.0e-6);
		size_t length = strlen(out_path);
		image_file_format_t format = (length > 4 && !strcmp(out_path + length - 4, ".hdr")) ? image_file_format_hdr : image_file_format_png;
		result = save_screenshot(out_path, format, &lit_scene, &targets, &spec);
		if (!result)
			printf("Saved %s.\n", out_path);
	}
	free_lit_scene(&lit_scene);
	return result;
}
