/* The display transform of the reference's tonemap pass, on the host: what
 * the window shows for an accumulation target. Follows
 * src/shaders/tonemap.frag.glsl:17-82 (operators and main()) and
 * src/shaders/srgb_utility.glsl:2-4 for the encoding that the sRGB swapchain
 * (src/vulkan_basics.c:212) applies to the shader's output. */
#include "pt_host.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

/* tonemap.frag.glsl:17-43 */
static void apply_khronos_pbr_neutral(float color[3]) {
	const float start_compression = 0.8f - 0.04f;
	const float desaturation = 0.15f;
	float darkest = fminf(color[0], fminf(color[1], color[2]));
	float offset = (darkest < 0.08f) ? (darkest - 6.25f * darkest * darkest) : 0.04f;
	for (int i = 0; i != 3; ++i)
		color[i] -= offset;
	float brightest = fmaxf(color[0], fmaxf(color[1], color[2]));
	if (brightest < start_compression)
		return;
	float compressed = 1.0f - start_compression;
	float new_brightest = 1.0f - compressed * compressed / (brightest + compressed - start_compression);
	float weight = 1.0f - 1.0f / (desaturation * (brightest - new_brightest) + 1.0f);
	for (int i = 0; i != 3; ++i) {
		float scaled = color[i] * (new_brightest / brightest);
		color[i] = scaled * (1.0f - weight) + new_brightest * weight;
	}
}

/* tonemap.frag.glsl:49-64 */
static void apply_aces_fit(float color[3]) {
	static const float aces_in[3][3] = { { 0.59719f, 0.35458f, 0.04823f }, { 0.07600f, 0.90834f, 0.01566f }, { 0.02840f, 0.13383f, 0.83777f } };
	static const float aces_out[3][3] = { { 1.60475f, -0.53108f, -0.07367f }, { -0.10208f, 1.10813f, -0.00605f }, { -0.00327f, -0.07276f, 1.07602f } };
	float w[3];
	for (int i = 0; i != 3; ++i) {
		float v = aces_in[i][0] * color[0] + aces_in[i][1] * color[1] + aces_in[i][2] * color[2];
		w[i] = (v * (v + 0.0245786f) - 0.000090537f) / (v * (0.983729f * v + 0.4329510f) + 0.238081f);
	}
	for (int i = 0; i != 3; ++i)
		color[i] = aces_out[i][0] * w[0] + aces_out[i][1] * w[1] + aces_out[i][2] * w[2];
}

void tonemap_pixel(float out_rgb[3], const float hdr_radiance[4], float exposure, uint32_t accum_frame_count, tonemapper_t tonemapper) {
	float factor = exposure / (float) accum_frame_count;
	for (int i = 0; i != 3; ++i)
		out_rgb[i] = hdr_radiance[i] * factor;
	if (tonemapper == tonemapper_clamp)
		for (int i = 0; i != 3; ++i)
			out_rgb[i] = fminf(fmaxf(out_rgb[i], 0.0f), 1.0f);
	else if (tonemapper == tonemapper_aces)
		apply_aces_fit(out_rgb);
	else if (tonemapper == tonemapper_khronos_pbr_neutral)
		apply_khronos_pbr_neutral(out_rgb);
	int any_nan = 0, any_inf = 0;
	for (int i = 0; i != 4; ++i) {
		any_nan |= isnan(hdr_radiance[i]);
		any_inf |= isinf(hdr_radiance[i]);
	}
	/* NaN pixels are shown violet, infinite ones red (tonemap.frag.glsl:77-82) */
	if (any_nan) { out_rgb[0] = 1.0f; out_rgb[1] = 0.0f; out_rgb[2] = 1.0f; }
	if (any_inf) { out_rgb[0] = 1.0f; out_rgb[1] = 0.0f; out_rgb[2] = 0.0f; }
}

void tonemap_image(uint8_t* out_srgb, const float* hdr_radiance, size_t pixel_count, float exposure, uint32_t accum_frame_count, tonemapper_t tonemapper) {
	for (size_t p = 0; p != pixel_count; ++p) {
		float rgb[3];
		tonemap_pixel(rgb, &hdr_radiance[4 * p], exposure, accum_frame_count, tonemapper);
		for (int i = 0; i != 3; ++i) {
			/* UNORM conversion of an sRGB attachment: clamp, encode, round */
			float linear = fminf(fmaxf(rgb[i], 0.0f), 1.0f);
			float srgb = (linear <= 0.0031308f) ? (12.92f * linear) : (1.055f * powf(linear, 1.0f / 2.4f) - 0.055f);
			out_srgb[3 * p + i] = (uint8_t) (srgb * 255.0f + 0.5f);
		}
	}
}

int save_display_image(const char* file_path, image_file_format_t format, lit_scene_t* lit_scene, const render_targets_t* render_targets, const scene_spec_t* scene_spec) {
	if (format == image_file_format_hdr) {
		printf("Failed to save the displayed image to %s. The display is low dynamic range; use PNG or JPG.\n", file_path);
		return 1;
	}
	size_t pixel_count = (size_t) render_targets->width * render_targets->height;
	float* floats = malloc(pixel_count * 4 * sizeof(float));
	uint8_t* ldr = malloc(pixel_count * 3);
	uint32_t accum_frame_count = 0;
	int result = reduce_render_targets(lit_scene) || pt_read_accumulation(lit_scene->scene.device_scene, floats, &accum_frame_count);
	if (result)
		printf("Failed to copy the off-screen render target to host memory.\n");
	else {
		tonemap_image(ldr, floats, pixel_count, scene_spec->exposure, accum_frame_count, scene_spec->tonemapper);
		result = (format == image_file_format_jpg) ? write_jpg(file_path, render_targets->width, render_targets->height, ldr, 90)
			: write_png(file_path, render_targets->width, render_targets->height, ldr);
		if (result)
			printf("Failed to save the displayed image to %s. Please check path, permissions and available disk space.\n", file_path);
	}
	free(floats);
	free(ldr);
	return result;
}
