/* Scene specification, lights, constants and the render loop of the host:
 * the parts of the reference's src/main.c that feed the light transport,
 * without window, GUI and swapchain. */
#include "pt_host.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

_Static_assert(sizeof(camera_t) == 64, "camera_t must match the quicksave layout");
_Static_assert(sizeof(scene_spec_t) == 128, "scene_spec_t must match the 128-byte quicksave record");
_Static_assert(sizeof(constants_t) == 784, "constants_t must match the std140 block");

/* src/main.c:9-77 */
int get_scene_file(scene_file_t scene_file, const char** scene_name, const char** scene_file_path, const char** texture_path, const char** light_path, const char** quicksave_path) {
	static const struct { const char *name, *file, *textures, *lights, *save; } table[scene_file_count] = {
		[scene_file_bistro_outside] = { "Bistro outside", "data/Bistro_outside.vks", "data/Bistro_textures", "data/Bistro_outside.lights", "data/saves/bistro/outside.rt_save" },
		[scene_file_cornell_box] = { "Cornell box", "data/cornell_box.vks", "data/cornell_box_textures", "data/cornell_box.lights", "data/saves/cornell_box/default.rt_save" },
		[scene_file_arcade] = { "Arcade", "data/Arcade.vks", "data/Arcade_textures", "data/Arcade.lights", "data/saves/Arcade/default.rt_save" },
		[scene_file_attic] = { "Attic", "data/attic.vks", "data/attic_textures", "data/attic.lights", "data/saves/attic/default.rt_save" },
		[scene_file_bistro_inside] = { "Bistro inside", "data/Bistro_inside.vks", "data/Bistro_textures", "data/Bistro_inside.lights", "data/saves/bistro/inside.rt_save" },
		[scene_file_living_room_day] = { "Living room day", "data/living_room_day.vks", "data/living_room_textures", "data/living_room_day.lights", "data/saves/living_room/day.rt_save" },
		[scene_file_living_room_night] = { "Living room night", "data/living_room_night.vks", "data/living_room_textures", "data/living_room_night.lights", "data/saves/living_room/night.rt_save" },
	};
	bool known = (int) scene_file >= 0 && scene_file < scene_file_count;
	if (scene_name) (*scene_name) = known ? table[scene_file].name : NULL;
	if (scene_file_path) (*scene_file_path) = known ? table[scene_file].file : NULL;
	if (texture_path) (*texture_path) = known ? table[scene_file].textures : NULL;
	if (light_path) (*light_path) = known ? table[scene_file].lights : NULL;
	if (quicksave_path) (*quicksave_path) = known ? table[scene_file].save : NULL;
	return !known;
}

/* src/main.c:81-90 */
int quicksave(const scene_spec_t* spec, const char* save_path) {
	if (save_path == NULL)
		save_path = "data/quicksave.rt_save";
	FILE* file = fopen(save_path, "wb");
	if (!file)
		return 1;
	size_t written = fwrite(spec, sizeof(*spec), 1, file);
	fclose(file);
	if (written == 1)
		printf("Quicksave.\n");
	return written != 1;
}

/* src/main.c:93-109 */
int quickload(scene_spec_t* spec, const char* save_path) {
	if (save_path == NULL)
		save_path = "data/quicksave.rt_save";
	FILE* file = fopen(save_path, "rb");
	if (file) {
		scene_spec_t loaded;
		size_t read = fread(&loaded, sizeof(loaded), 1, file);
		fclose(file);
		if (read == 1) {
			(*spec) = loaded;
			printf("Quickload from %s.\n", save_path);
			return 0;
		}
	}
	printf("Could not open the quicksave file %s.\n", save_path);
	return 1;
}

/* src/main.c:124-146 */
void init_scene_spec(scene_spec_t* spec) {
	memset(spec, 0, sizeof(*spec));
	spec->scene_file = scene_file_cornell_box;
	spec->camera.type = camera_type_first_person;
	spec->camera.near = 0.01f;
	spec->camera.far = 1.0e4f;
	spec->camera.fov = 3.14159265358979323846 * 0.4f;
	spec->camera.height = 10.0f;
	spec->camera.speed = 2.0f;
	spec->camera.rotation.angles[0] = 1.570796f;
	spec->camera.position[0] = 0.278f;
	spec->camera.position[1] = 0.379f;
	spec->camera.position[2] = 0.2744f;
	spec->tonemapper = tonemapper_aces;
	spec->exposure = 1.0f;
	spec->sky_color[0] = 0.2f; spec->sky_color[1] = 0.4f; spec->sky_color[2] = 1.0f;
	spec->sky_strength = 0.0f;
	spec->emission_material_color[0] = 1.0f; spec->emission_material_color[1] = 0.9f; spec->emission_material_color[2] = 0.6f;
	spec->emission_material_strength = 20.0f;
	printf("Using a default scene specification.\n");
}

/* src/main.c:149-155 */
void init_render_settings(render_settings_t* settings) {
	settings->sampling_strategy = sampling_strategy_nee;
	settings->path_length = 4;
}

/* src/main.c:687-704 */
static int create_lit_scene_on(lit_scene_t* lit_scene, const int* device_indices, uint32_t device_count, bool use_context, const char* scene_path, const char* textures_path, const char* lights_path);

int create_lit_scene_from_paths(lit_scene_t* lit_scene, int device_index, const char* scene_path, const char* textures_path, const char* lights_path) {
	return create_lit_scene_on(lit_scene, &device_index, 1, false, scene_path, textures_path, lights_path);
}

int create_lit_scene_from_paths_devices(lit_scene_t* lit_scene, const int* device_indices, uint32_t device_count, const char* scene_path, const char* textures_path, const char* lights_path) {
	return create_lit_scene_on(lit_scene, device_indices, device_count, true, scene_path, textures_path, lights_path);
}

static int create_lit_scene_on(lit_scene_t* lit_scene, const int* device_indices, uint32_t device_count, bool use_context, const char* scene_path, const char* textures_path, const char* lights_path) {
	memset(lit_scene, 0, sizeof(*lit_scene));
	FILE* file = lights_path ? fopen(lights_path, "rb") : NULL;
	if (file) {
		uint32_t count = 0;
		if (fread(&count, sizeof(uint32_t), 1, file) != 1)
			count = 0;
		if (count > (1u << 24))
			count = 1u << 24;
		lit_scene->spherical_lights = calloc(count ? count : 1, sizeof(float) * 4);
		lit_scene->spherical_light_count = (uint32_t) fread(lit_scene->spherical_lights, sizeof(float) * 4, count, file);
		printf("Loaded %u spherical lights from %s.\n", lit_scene->spherical_light_count, lights_path);
		fclose(file);
	}
	else
		lit_scene->spherical_lights = calloc(1, sizeof(float) * 4);
	int result = use_context ? load_scene_devices(&lit_scene->scene, device_indices, device_count, scene_path, textures_path)
		: load_scene(&lit_scene->scene, device_indices[0], scene_path, textures_path);
	if (!result) {
		printf("Loaded %lu triangles and %lu materials from %s.\n", (unsigned long) lit_scene->scene.header.triangle_count, (unsigned long) lit_scene->scene.header.material_count, scene_path);
		if (lit_scene->scene.device_context)
			result = pt_context_set_lights(lit_scene->scene.device_context, &lit_scene->spherical_lights[0][0], lit_scene->spherical_light_count);
		else
			result = pt_set_lights(lit_scene->scene.device_scene, &lit_scene->spherical_lights[0][0], lit_scene->spherical_light_count);
	}
	if (result)
		free_lit_scene(lit_scene);
	return result;
}

static char* join_path(const char* root, const char* path) {
	size_t length = (root ? strlen(root) + 1 : 0) + strlen(path) + 1;
	char* result = malloc(length);
	if (root && root[0])
		snprintf(result, length, "%s/%s", root, path);
	else
		snprintf(result, length, "%s", path);
	return result;
}

/* src/main.c:679-704 */
static int create_lit_scene_spec_on(lit_scene_t* lit_scene, const int* device_indices, uint32_t device_count, bool use_context, const scene_spec_t* scene_spec, const char* data_root);

int create_lit_scene(lit_scene_t* lit_scene, int device_index, const scene_spec_t* scene_spec, const char* data_root) {
	return create_lit_scene_spec_on(lit_scene, &device_index, 1, false, scene_spec, data_root);
}

int create_lit_scene_devices(lit_scene_t* lit_scene, const int* device_indices, uint32_t device_count, const scene_spec_t* scene_spec, const char* data_root) {
	return create_lit_scene_spec_on(lit_scene, device_indices, device_count, true, scene_spec, data_root);
}

static int create_lit_scene_spec_on(lit_scene_t* lit_scene, const int* device_indices, uint32_t device_count, bool use_context, const scene_spec_t* scene_spec, const char* data_root) {
	const char* scene_path;
	const char* textures_path;
	const char* lights_path;
	if (get_scene_file(scene_spec->scene_file, NULL, &scene_path, &textures_path, &lights_path, NULL)) {
		printf("Failed to load the scene, because the requested scene file is unknown.\n");
		memset(lit_scene, 0, sizeof(*lit_scene));
		return 1;
	}
	char* scene = join_path(data_root, scene_path);
	char* textures = join_path(data_root, textures_path);
	char* lights = join_path(data_root, lights_path);
	int result = create_lit_scene_on(lit_scene, device_indices, device_count, use_context, scene, textures, lights);
	free(scene);
	free(textures);
	free(lights);
	return result;
}

/* src/main.c:707-710 */
void free_lit_scene(lit_scene_t* lit_scene) {
	free_scene(&lit_scene->scene);
	free(lit_scene->spherical_lights);
	memset(lit_scene, 0, sizeof(*lit_scene));
}

int apply_render_settings(lit_scene_t* lit_scene, const render_settings_t* settings) {
	if (lit_scene->scene.device_context)
		return pt_context_set_render_settings(lit_scene->scene.device_context, (int) settings->sampling_strategy, settings->path_length, lit_scene->scene.emission_material_index);
	return pt_set_render_settings(lit_scene->scene.device_scene, (int) settings->sampling_strategy, settings->path_length, lit_scene->scene.emission_material_index);
}

int set_light_sampling(lit_scene_t* lit_scene, int light_sampling) {
	if (lit_scene->scene.device_context)
		return pt_context_set_light_sampling(lit_scene->scene.device_context, light_sampling);
	return pt_set_light_sampling(lit_scene->scene.device_scene, light_sampling);
}

int reduce_render_targets(lit_scene_t* lit_scene) {
	if (!lit_scene->scene.device_context)
		return 0;
	if (pt_context_reduce(lit_scene->scene.device_context)) {
		printf("Failed to sum the render targets of %u GPUs.\n", pt_context_device_count(lit_scene->scene.device_context));
		return 1;
	}
	return 0;
}

/* src/main.c:519-560 */
int create_render_targets(render_targets_t* render_targets, lit_scene_t* lit_scene, uint32_t width, uint32_t height) {
	memset(render_targets, 0, sizeof(*render_targets));
	if (lit_scene->scene.device_context ? pt_context_create_render_targets(lit_scene->scene.device_context, width, height) : pt_create_render_targets(lit_scene->scene.device_scene, width, height)) {
		printf("Failed to create render targets with resolution %ux%u.\n", width, height);
		return 1;
	}
	render_targets->width = width;
	render_targets->height = height;
	return 0;
}

/* src/main.c:629-676 */
int write_constant_buffer(constants_t* constants, const scene_spec_t* scene_spec, const lit_scene_t* lit_scene, const render_targets_t* render_targets) {
	const camera_t* camera = &scene_spec->camera;
	constants_t cts;
	memset(&cts, 0, sizeof(cts));
	cts.viewport_size[0] = (float) render_targets->width;
	cts.viewport_size[1] = (float) render_targets->height;
	cts.inv_viewport_size[0] = 1.0f / (float) render_targets->width;
	cts.inv_viewport_size[1] = 1.0f / (float) render_targets->height;
	cts.camera_type = (int) camera->type;
	cts.exposure = scene_spec->exposure;
	cts.frame_index = scene_spec->frame_index;
	cts.accum_frame_count = render_targets->accum_frame_count + 1;
	memcpy(cts.camera_pos, camera->position, sizeof(cts.camera_pos));
	float world_to_view[4 * 4];
	get_world_to_view_space(world_to_view, camera);
	for (uint32_t i = 0; i != 3; ++i)
		cts.hemispherical_camera_normal[i] = world_to_view[2 * 4 + i];
	const scene_t* scene = &lit_scene->scene;
	memcpy(cts.dequantization_factor, scene->header.dequantization_factor, sizeof(cts.dequantization_factor));
	memcpy(cts.dequantization_summand, scene->header.dequantization_summand, sizeof(cts.dequantization_summand));
	for (uint32_t i = 0; i != 3; ++i) {
		cts.sky_radiance[i] = scene_spec->sky_color[i] * scene_spec->sky_strength;
		cts.emission_material_radiance[i] = scene_spec->emission_material_color[i] * scene_spec->emission_material_strength;
	}
	memcpy(cts.params, scene_spec->params, sizeof(cts.params));
	uint32_t light_count = lit_scene->spherical_light_count;
	if (light_count > MAX_SPHERICAL_LIGHT_COUNT)
		light_count = MAX_SPHERICAL_LIGHT_COUNT;
	if (light_count)
		memcpy(cts.spherical_lights, lit_scene->spherical_lights, sizeof(float) * 4 * light_count);
	float aspect = ((float) render_targets->width) / ((float) render_targets->height);
	get_world_to_projection_space(cts.world_to_projection_space, camera, aspect);
	invert_mat4(cts.projection_to_world_space, cts.world_to_projection_space);
	(*constants) = cts;
	return 0;
}

int render_frames(scene_spec_t* scene_spec, lit_scene_t* lit_scene, render_targets_t* render_targets, uint32_t frame_count) {
	if (frame_count == 0)
		return 0;
	constants_t constants;
	if (write_constant_buffer(&constants, scene_spec, lit_scene, render_targets))
		return 1;
	/* The reference increments the frame index before each frame, so the
	   first of these frames uses frame_index + 1 */
	uint32_t first_frame_index = scene_spec->frame_index + 1;
	int accumulate = render_targets->accum_frame_count != 0;
	if (lit_scene->scene.device_context ? pt_context_render_samples(lit_scene->scene.device_context, &constants, first_frame_index, frame_count, accumulate)
		: pt_render_samples(lit_scene->scene.device_scene, &constants, first_frame_index, frame_count, 1, accumulate)) {
		printf("Failed to render %u frames.\n", frame_count);
		return 1;
	}
	scene_spec->frame_index += frame_count;
	render_targets->accum_frame_count += frame_count;
	return 0;
}

/* src/main.c:1986-2086 */
int save_screenshot(const char* file_path, image_file_format_t format, lit_scene_t* lit_scene, const render_targets_t* render_targets, const scene_spec_t* scene_spec) {
	uint32_t width = render_targets->width, height = render_targets->height;
	size_t pixel_count = (size_t) width * height;
	float* floats = malloc(pixel_count * 4 * sizeof(float));
	uint32_t accum_frame_count = 0;
	if (reduce_render_targets(lit_scene) || pt_read_accumulation(lit_scene->scene.device_scene, floats, &accum_frame_count)) {
		printf("Failed to copy the off-screen render target to host memory for taking a screenshot.\n");
		free(floats);
		return 1;
	}
	int result = 0;
	if (format == image_file_format_hdr) {
		float* hdr_copy = malloc(pixel_count * 3 * sizeof(float));
		for (size_t p = 0; p != pixel_count; ++p)
			for (uint32_t i = 0; i != 3; ++i)
				hdr_copy[3 * p + i] = floats[4 * p + i] / ((float) accum_frame_count);
		result = write_hdr(file_path, width, height, hdr_copy);
		if (result)
			printf("Failed to save a HDR screenshot to %s. Please check path, permissions and available disk space.\n", file_path);
		free(hdr_copy);
	}
	else if (format == image_file_format_png || format == image_file_format_jpg) {
		uint8_t* ldr_copy = malloc(pixel_count * 3);
		float factor = scene_spec->exposure / ((float) accum_frame_count);
		for (size_t p = 0; p != pixel_count; ++p) {
			for (uint32_t i = 0; i != 3; ++i) {
				float rgb = floats[4 * p + i] * factor;
				rgb = (rgb >= 0.0f) ? rgb : 0.0f;
				rgb = (rgb <= 1.0f) ? rgb : 1.0f;
				float srgb = (rgb <= 0.0031308f) ? (12.92f * rgb) : (1.055f * powf(rgb, 1.0f / 2.4f) - 0.055f);
				ldr_copy[3 * p + i] = (uint8_t) (srgb * 255.0f + 0.5f);
			}
		}
		result = (format == image_file_format_jpg) ? write_jpg(file_path, width, height, ldr_copy, 90) : write_png(file_path, width, height, ldr_copy);
		if (result)
			printf("Failed to save a screenshot to %s. Please check path, permissions and available disk space.\n", file_path);
		free(ldr_copy);
	}
	else {
		printf("Failed to save a screenshot to %s. The image file format %d is unknown.\n", file_path, (int) format);
		result = 1;
	}
	free(floats);
	return result;
}
