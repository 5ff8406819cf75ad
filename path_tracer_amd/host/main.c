/* Headless command line driver: loads a quicksave and its scene, renders a
 * number of samples per pixel on one GPU (or, with -devices, shared out among
 * several and summed over xGMI) and writes a screenshot, or runs a
 * range of the reference's slides (-bN -eN, src/main.c:2111-2112) as a batch
 * job. No window, no GUI. */
#include "pt_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static void print_usage(void) {
	printf("Usage: pt_render [-data ROOT] [-save FILE.rt_save] [-scene N] [-wWIDTH] [-hHEIGHT] [-spp N] [-path_length N]\n"
		"                 [-strategy spherical|psa|brdf|nee] [-device N | -devices N,N,...] [-o OUT.png|OUT.jpg|OUT.hdr] [-display] [-light_tree]\n"
		"       pt_render -bBEGIN -eEND [-data ROOT] [-out ROOT] [-no_screenshots] [-max_spp N] [-wWIDTH] [-hHEIGHT]\n"
		"ROOT is the directory holding data/ (default: the working directory, like the reference).\n"
		"-devices renders on several GPUs of this node: each takes every n-th frame, RCCL sums the results.\n"
		"-display saves what the window would show (tonemapper and exposure of the quicksave) instead of a screenshot.\n"
		"-light_tree lets next event estimation choose its light through a tree over the lights (for thousands of lights).\n"
		"-bBEGIN -eEND runs slides BEGIN to END-1; screenshots go to renders/ below -out.\n");
}

int main(int argc, char** argv) {
	const char* data_root = "";
	const char* save_path = NULL;
	const char* out_path = "screenshot.png";
	const char* output_root = ".";
	int scene_override = -1, device = 0, display = 0, slide_screenshots = 1, light_tree = 0;
	int devices[64];
	uint32_t device_count = 0;
	uint32_t slide_begin = 0, slide_end = 0, max_spp = 0;
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
		else if (!strcmp(arg, "-devices") && i + 1 < argc) {
			const char* list = argv[++i];
			device_count = 0;
			while (*list && device_count < 64) {
				char* end = NULL;
				long value = strtol(list, &end, 10);
				if (end == list)
					break;
				devices[device_count++] = (int) value;
				list = (*end == ',') ? end + 1 : end;
				if (*end && *end != ',')
					break;
			}
			if (device_count == 0 || *list) {
				print_usage();
				return 1;
			}
		}
		else if (!strcmp(arg, "-o") && i + 1 < argc) out_path = argv[++i];
		else if (!strcmp(arg, "-out") && i + 1 < argc) output_root = argv[++i];
		else if (!strcmp(arg, "-display")) display = 1;
		else if (!strcmp(arg, "-light_tree")) light_tree = 1;
		else if (!strcmp(arg, "-no_screenshots")) slide_screenshots = 0;
		else if (!strcmp(arg, "-max_spp") && i + 1 < argc) max_spp = (uint32_t) atoi(argv[++i]);
		else if (arg[0] == '-' && arg[1] == 'b' && arg[2]) slide_begin = (uint32_t) atoi(arg + 2);
		else if (arg[0] == '-' && arg[1] == 'e' && arg[2]) slide_end = (uint32_t) atoi(arg + 2);
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
	if (slide_begin < slide_end) {
		slideshow_t* show = calloc(1, sizeof(slideshow_t));
		create_slides(show);
		show->slide_begin = slide_begin;
		show->slide_end = slide_end;
		int failed = run_slideshow(show, device, data_root, output_root, width, height, slide_screenshots, max_spp);
		free_slideshow(show);
		free(show);
		return failed;
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
	if (device_count ? create_lit_scene_devices(&lit_scene, devices, device_count, &spec, data_root) : create_lit_scene(&lit_scene, device, &spec, data_root))
		return 1;
	if (device_count == 0)
		device_count = 1;
	render_targets_t targets;
	if (apply_render_settings(&lit_scene, &settings) || set_light_sampling(&lit_scene, light_tree ? PT_LIGHT_SAMPLING_TREE : PT_LIGHT_SAMPLING_EXACT) || create_render_targets(&targets, &lit_scene, width, height)) {
		free_lit_scene(&lit_scene);
		return 1;
	}
	struct timespec begin, end;
	clock_gettime(CLOCK_MONOTONIC, &begin);
	/* The timed region ends when the image is complete on the first GPU: rendering on all GPUs and the sum */
	int result = render_frames(&spec, &lit_scene, &targets, spp)
		|| (lit_scene.scene.device_context ? reduce_render_targets(&lit_scene) : pt_synchronize(lit_scene.scene.device_scene));
	clock_gettime(CLOCK_MONOTONIC, &end);
	if (!result) {
		double seconds = (double) (end.tv_sec - begin.tv_sec) + 1.0e-9 * (double) (end.tv_nsec - begin.tv_nsec);
		uint64_t rays = 0;
		int counted = 1;
		for (uint32_t i = 0; i != device_count; ++i) {
			pt_statistics_t stats;
			pt_scene_t* replica = lit_scene.scene.device_context ? pt_context_scene(lit_scene.scene.device_context, i) : lit_scene.scene.device_scene;
			if (pt_get_statistics(replica, &stats))
				counted = 0;
			else
				rays += stats.extension_rays + stats.shadow_rays;
		}
		if (counted)
			printf("Rendered %u spp at %ux%u on %u GPU%s in %.3f s: %.1f Mrays/s, %.1f Msamples/s.\n", spp, width, height, device_count, device_count == 1 ? "" : "s", seconds,
				(double) rays / seconds * 1.0e-6, (double) width * height * spp / seconds * 1.0e-6);
		if (lit_scene.scene.device_context)
			printf("Summing the render targets over xGMI took %.3f ms.\n", pt_context_last_reduce_ms(lit_scene.scene.device_context));
		size_t length = strlen(out_path);
		image_file_format_t format = image_file_format_png;
		if (length > 4 && !strcmp(out_path + length - 4, ".hdr")) format = image_file_format_hdr;
		if (length > 4 && !strcmp(out_path + length - 4, ".jpg")) format = image_file_format_jpg;
		result = display ? save_display_image(out_path, format, &lit_scene, &targets, &spec) : save_screenshot(out_path, format, &lit_scene, &targets, &spec);
		if (!result)
			printf("Saved %s.\n", out_path);
	}
	free_lit_scene(&lit_scene);
	return result;
}
