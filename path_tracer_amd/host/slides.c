/* The slideshow of the reference as a batch job: each slide is a quicksave,
 * render settings and a sample count, and ends in a screenshot
 * (slide_t at src/main.h:120-136, create_slides() in src/slides.c,
 * show_slide() at src/main.c:1482-1494, the advance rule at :1658-1690).
 * The reference renders one frame per loop iteration and takes the screenshot
 * when accum_frame_count reaches screenshot_frame; here the frames of a slide
 * are rendered back to back. Screenshot paths are relative to an output
 * directory instead of the author's home directory. */
#include "pt_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

/* Creates the directories above file_path (the reference expects them to exist) */
static void make_parent_directories(const char* file_path) {
	char* path = malloc(strlen(file_path) + 1);
	strcpy(path, file_path);
	for (char* c = path + 1; *c; ++c) {
		if (*c == '/') {
			*c = 0;
			mkdir(path, 0777);
			*c = '/';
		}
	}
	free(path);
}

static char* join_path(const char* directory, const char* name) {
	size_t size = strlen(directory) + strlen(name) + 2;
	char* path = malloc(size);
	snprintf(path, size, "%s/%s", directory, name);
	return path;
}

static void add_slide(slideshow_t* show, const char* quicksave, sampling_strategy_t strategy, uint32_t path_length, uint32_t frames, const char* screenshot, image_file_format_t format) {
	if (show->slide_count >= MAX_SLIDE_COUNT) {
		printf("WARNING: More than MAX_SLIDE_COUNT = %u slides. Increase it.\n", MAX_SLIDE_COUNT);
		return;
	}
	slide_t* slide = &show->slides[show->slide_count++];
	slide->quicksave = join_path("data/saves", quicksave);
	slide->render_settings.sampling_strategy = strategy;
	slide->render_settings.path_length = path_length;
	slide->screenshot_frame = frames;
	slide->screenshot_path = join_path("renders", screenshot);
	slide->screenshot_format = format;
}

uint32_t create_slides(slideshow_t* show) {
	/* The same 44 configurations as src/slides.c:5-140, as a table */
	static const struct { const char* quicksave; sampling_strategy_t strategy; uint32_t path_length, frames; const char* screenshot; } fixed[] = {
		{ "bistro/pretty.rt_save", sampling_strategy_nee, 4, 2048, "pretty_bistro.png" },
		{ "bistro/pretty.rt_save", sampling_strategy_nee, 1, 1, "pretty_bistro_emission.png" },
		{ "bistro/path_overview.rt_save", sampling_strategy_nee, 4, 2048, "path_overview_bistro.png" },
		{ "bistro/pretty.rt_save", sampling_strategy_spherical, 4, 128, "pretty_bistro_spherical.png" },
		{ "bistro/pretty_day.rt_save", sampling_strategy_spherical, 4, 128, "pretty_bistro_spherical_day.png" },
		{ "bistro/mirror.rt_save", sampling_strategy_spherical, 4, 4, "mirror_spherical.png" },
		{ "bistro/mirror.rt_save", sampling_strategy_psa, 4, 4, "mirror_psa.png" },
		{ "bistro/mirror.rt_save", sampling_strategy_brdf, 4, 4, "mirror_brdf.png" },
		{ "bistro/shading_point_day.rt_save", sampling_strategy_nee, 4, 2048, "shading_point_incoming.png" },
		{ "bistro/pretty_day.rt_save", sampling_strategy_brdf, 4, 16, "pretty_bistro_brdf_day.png" },
		{ "bistro/pretty.rt_save", sampling_strategy_brdf, 4, 16, "pretty_bistro_brdf.png" },
		{ "bistro/pretty.rt_save", sampling_strategy_nee, 4, 16, "pretty_bistro_nee.png" },
		{ "bistro/shading_point_dark.rt_save", sampling_strategy_nee, 4, 2048, "shading_point_dark_incoming.png" },
		{ "bistro/pretty.rt_save", sampling_strategy_nee, 4, 1, "pretty_bistro_1_spp.hdr" },
	};
	show->slide_count = 0;
	for (size_t i = 0; i != sizeof(fixed) / sizeof(fixed[0]); ++i) {
		size_t length = strlen(fixed[i].screenshot);
		image_file_format_t format = strcmp(fixed[i].screenshot + length - 4, ".hdr") ? image_file_format_png : image_file_format_hdr;
		add_slide(show, fixed[i].quicksave, fixed[i].strategy, fixed[i].path_length, fixed[i].frames, fixed[i].screenshot, format);
	}
	static const struct { const char* quicksave; uint32_t frames; const char* directory; } series[] = {
		{ "bistro/pretty.rt_save", 16 * 2048, "bistro_path_lengths" },
		{ "cornell_box/default.rt_save", 2048, "cornell_box_path_lengths" },
		{ "living_room/day.rt_save", 16 * 2048, "living_room_path_lengths" },
	};
	for (size_t s = 0; s != sizeof(series) / sizeof(series[0]); ++s) {
		for (uint32_t i = 0; i != 10; ++i) {
			char name[128];
			snprintf(name, sizeof(name), "%s/%u.png", series[s].directory, i);
			add_slide(show, series[s].quicksave, sampling_strategy_nee, i, series[s].frames, name, image_file_format_png);
		}
	}
	printf("Defined %u slides.\n", show->slide_count);
	return show->slide_count;
}

void free_slideshow(slideshow_t* show) {
	for (uint32_t i = 0; i != show->slide_count; ++i) {
		free(show->slides[i].quicksave);
		free(show->slides[i].screenshot_path);
	}
	memset(show, 0, sizeof(*show));
}

int run_slideshow(slideshow_t* show, int device_index, const char* data_root, const char* output_root, uint32_t width, uint32_t height, int slide_screenshots, uint32_t frame_limit) {
	if (show->slide_end > show->slide_count) {
		printf("There are only %u slides but the slideshow was requested to end at slide %u.\n", show->slide_count, show->slide_end);
		return 1;
	}
	lit_scene_t lit_scene;
	memset(&lit_scene, 0, sizeof(lit_scene));
	render_targets_t targets;
	memset(&targets, 0, sizeof(targets));
	int scene_loaded = 0, result = 0;
	scene_file_t loaded_file = scene_file_count;
	for (show->slide_current = show->slide_begin; show->slide_current < show->slide_end && !result; ++show->slide_current) {
		const slide_t* slide = &show->slides[show->slide_current];
		scene_spec_t spec;
		char* quicksave = join_path(data_root, slide->quicksave);
		result = quickload(&spec, quicksave);
		free(quicksave);
		if (result) {
			printf("Failed to load the quicksave for slide %u.\n", show->slide_current);
			break;
		}
		printf("Showing slide %u.\n", show->slide_current);
		/* A scene stays loaded while consecutive slides use it (src/main.c:110-113) */
		if (!scene_loaded || loaded_file != spec.scene_file) {
			if (scene_loaded)
				free_lit_scene(&lit_scene);
			scene_loaded = 0;
			if ((result = create_lit_scene(&lit_scene, device_index, &spec, data_root)))
				break;
			scene_loaded = 1;
			loaded_file = spec.scene_file;
		}
		if ((result = apply_render_settings(&lit_scene, &slide->render_settings) || create_render_targets(&targets, &lit_scene, width, height)))
			break;
		uint32_t frames = slide->screenshot_frame;
		if (frame_limit && frames > frame_limit)
			frames = frame_limit;
		if ((result = render_frames(&spec, &lit_scene, &targets, frames)))
			break;
		if (slide_screenshots && slide->screenshot_path) {
			char* path = join_path(output_root, slide->screenshot_path);
			make_parent_directories(path);
			result = save_screenshot(path, slide->screenshot_format, &lit_scene, &targets, &spec);
			if (!result)
				printf("Saved %s after %u frames.\n", path, targets.accum_frame_count);
			free(path);
		}
	}
	if (scene_loaded)
		free_lit_scene(&lit_scene);
	return result;
}
