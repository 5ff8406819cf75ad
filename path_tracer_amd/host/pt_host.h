/* C99 host of the MI355X path tracer. It keeps the scene/camera/lights
 * surface of the reference (MomentsInGraphics/path_tracer) - same structs,
 * same function names and argument meaning, same file formats, same error
 * behaviour (0 = success, message on stdout) - and calls the GPU through the
 * C ABI of include/pt_hip.h where the reference calls Vulkan. A Vulkan
 * device_t becomes a GPU index. Citations are files of the reference.
 */
#ifndef PT_HOST_H
#define PT_HOST_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/pt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- src/math_utilities.h ---- */
#define M_PI_F 3.141592653589793238462643f
void mat_mat_mul(float* out, const float* lhs, const float* rhs, uint32_t m, uint32_t k, uint32_t n);
static inline void mat_vec_mul(float* out, const float* mat, const float* vec, uint32_t m, uint32_t n) { mat_mat_mul(out, mat, vec, m, n, 1); }
bool normalize(float* vec, uint32_t entry_count);
void rotation_matrix_from_angles(float out_rotation[3 * 3], const float angles[3]);
void invert_mat4(float out_inv[4 * 4], float mat[4 * 4]);

/* ---- src/camera.h:11-63 ---- */
typedef struct {
	float angles[3];
	bool mouse_active;
	float origin_angles[3];
} controllable_rotation_t;

typedef enum {
	camera_type_first_person,
	camera_type_ortho,
	camera_type_hemispherical,
	camera_type_spherical,
	camera_type_count,
} camera_type_t;

typedef struct {
	controllable_rotation_t rotation;
	float position[3];
	float speed;
	float near, far;
	camera_type_t type;
	float fov;
	float height;
} camera_t;

void get_world_to_view_space(float world_to_view[4 * 4], const camera_t* camera);
void get_view_to_projection_space(float view_to_projection[4 * 4], const camera_t* camera, float aspect_ratio);
void get_world_to_projection_space(float world_to_projection[4 * 4], const camera_t* camera, float aspect_ratio);

/* ---- src/scene.h ---- */
typedef struct {
	uint32_t marker;
	uint32_t version;
	uint64_t material_count;
	uint64_t triangle_count;
	float dequantization_factor[3], dequantization_summand[3];
	char** material_names;
} scene_file_header_t;

typedef enum {
	material_texture_type_base_color,
	material_texture_type_specular,
	material_texture_type_normal,
	material_texture_type_count,
} material_texture_type_t;

/* A scene that has been loaded from a scene file and is now resident in
 * the HBM of one GPU (scene_t at src/scene.h:89-98: mesh buffers, textures
 * and acceleration structures all live behind the device handle here). */
typedef struct {
	scene_file_header_t header;
	pt_scene_t* device_scene;
	/* Not in the reference: set when the scene is replicated on several GPUs
	 * (load_scene_devices()); device_scene is then its replica on the first */
	pt_context_t* device_context;
	/* Index of the material called _emission, 0 if there is none
	 * (src/main.c:936-939) */
	uint32_t emission_material_index;
} scene_t;

/* src/scene.h:104; device_index replaces the Vulkan device */
int load_scene(scene_t* scene, int device_index, const char* file_path, const char* texture_path);
/* The same for several GPUs of one node, which share out the frames among
 * themselves (no counterpart in the reference) */
int load_scene_devices(scene_t* scene, const int* device_indices, uint32_t device_count, const char* file_path, const char* texture_path);
void free_scene(scene_t* scene);

/* ---- src/textures.h: textures are read into host memory and handed to
 * pt_create_scene() instead of being copied into VkImages ---- */
typedef struct {
	uint32_t width, height;
	uint32_t format;
	uint32_t mipmap_count;
	uint64_t* mipmap_sizes;
	void** mipmap_data;
	uint8_t* payload;
} texture_t;

typedef struct {
	uint32_t texture_count;
	texture_t* textures;
} textures_t;

int load_textures(textures_t* textures, const char* const* texture_file_paths, uint32_t texture_count);
void free_textures(textures_t* textures);

/* ---- src/main.h ---- */
#define MAX_SPHERICAL_LIGHT_COUNT 32

typedef enum {
	scene_file_bistro_outside,
	scene_file_cornell_box,
	scene_file_arcade,
	scene_file_attic,
	scene_file_bistro_inside,
	scene_file_living_room_day,
	scene_file_living_room_night,
	scene_file_count,
} scene_file_t;

typedef enum {
	tonemapper_clamp,
	tonemapper_aces,
	tonemapper_khronos_pbr_neutral,
	tonemapper_count,
	tonemapper_force_int_32 = 0x7fffffff,
} tonemapper_t;

/* src/main.h:55-78: what quicksave/quickload store (128 bytes) */
typedef struct {
	scene_file_t scene_file;
	camera_t camera;
	tonemapper_t tonemapper;
	float exposure;
	uint32_t frame_index;
	float sky_color[3];
	float sky_strength;
	float emission_material_color[3];
	float emission_material_strength;
	float params[4];
} scene_spec_t;

typedef enum {
	sampling_strategy_spherical,
	sampling_strategy_psa,
	sampling_strategy_brdf,
	sampling_strategy_nee,
	sampling_strategy_count,
} sampling_strategy_t;

typedef struct {
	sampling_strategy_t sampling_strategy;
	uint32_t path_length;
} render_settings_t;

typedef enum {
	image_file_format_png,
	image_file_format_jpg,
	image_file_format_hdr,
} image_file_format_t;

typedef pt_constants_t constants_t;

/* lit_scene_t at src/main.h:276-283. All lights of the file are kept (the
 * reference drops those beyond 32); only the first 32 fit the constants. */
typedef struct {
	scene_t scene;
	uint32_t spherical_light_count;
	float (*spherical_lights)[4];
} lit_scene_t;

/* The render target of src/main.h:226-232 */
typedef struct {
	uint32_t width, height;
	uint32_t accum_frame_count;
} render_targets_t;

/* src/main.c:9-77; paths are relative to the working directory like in the
 * reference ("data/cornell_box.vks", ...) */
int get_scene_file(scene_file_t scene_file, const char** scene_name, const char** scene_file_path, const char** texture_path, const char** light_path, const char** quicksave_path);
/* src/main.c:81-109 */
int quicksave(const scene_spec_t* spec, const char* save_path);
int quickload(scene_spec_t* spec, const char* save_path);
/* src/main.c:120-146 without the attempt to quickload first */
void init_scene_spec(scene_spec_t* spec);
/* src/main.c:149-155 */
void init_render_settings(render_settings_t* settings);
/* src/main.c:679-704 with explicit paths (data_root is prepended to the
 * paths that get_scene_file() reports when the spec variant is used) */
int create_lit_scene_from_paths(lit_scene_t* lit_scene, int device_index, const char* scene_path, const char* textures_path, const char* lights_path);
int create_lit_scene(lit_scene_t* lit_scene, int device_index, const scene_spec_t* scene_spec, const char* data_root);
/* The same on several GPUs (no counterpart in the reference): every function
 * below that takes the lit scene then acts on all of them, render_frames()
 * shares the frames out and the screenshot functions sum the targets first */
int create_lit_scene_from_paths_devices(lit_scene_t* lit_scene, const int* device_indices, uint32_t device_count, const char* scene_path, const char* textures_path, const char* lights_path);
int create_lit_scene_devices(lit_scene_t* lit_scene, const int* device_indices, uint32_t device_count, const scene_spec_t* scene_spec, const char* data_root);
/* Sums the render targets of all GPUs into that of the first one; a no-op
 * for a lit scene on one GPU. save_screenshot() and save_display_image() call it. */
int reduce_render_targets(lit_scene_t* lit_scene);
void free_lit_scene(lit_scene_t* lit_scene);
/* Applies render_settings_t and the light list to the device scene (what
 * create_scene_subpass() bakes into shader defines at src/main.c:936-949) */
int apply_render_settings(lit_scene_t* lit_scene, const render_settings_t* settings);
/* Not in the reference: PT_LIGHT_SAMPLING_EXACT (its loops over all lights,
 * the default) or PT_LIGHT_SAMPLING_TREE for next event estimation */
int set_light_sampling(lit_scene_t* lit_scene, int light_sampling);
/* src/main.c:519-560 */
int create_render_targets(render_targets_t* render_targets, lit_scene_t* lit_scene, uint32_t width, uint32_t height);
/* src/main.c:629-676 */
int write_constant_buffer(constants_t* constants, const scene_spec_t* scene_spec, const lit_scene_t* lit_scene, const render_targets_t* render_targets);
/* One or more iterations of the main loop body of the reference:
 * handle_user_input() increments scene_spec->frame_index before every
 * frame (src/main.c:1624), then the frame is rendered and accumulated
 * (src/main.c:1842-1891). */
int render_frames(scene_spec_t* scene_spec, lit_scene_t* lit_scene, render_targets_t* render_targets, uint32_t frame_count);
/* src/main.c:1986-2086. PNG, JPG (quality 90) and HDR are written by this
 * library's own encoders. */
int save_screenshot(const char* file_path, image_file_format_t format, lit_scene_t* lit_scene, const render_targets_t* render_targets, const scene_spec_t* scene_spec);

/* Image writers (replace stbi_write_hdr / stbi_write_png of the vendored
 * stb_image_write.h) */
int write_hdr(const char* file_path, uint32_t width, uint32_t height, const float* rgb);
int write_png(const char* file_path, uint32_t width, uint32_t height, const uint8_t* rgb);
int write_jpg(const char* file_path, uint32_t width, uint32_t height, const uint8_t* rgb, int quality);

/* The tonemap pass (src/shaders/tonemap.frag.glsl:67-82) and the sRGB encoding
 * of the swapchain on the host. hdr_radiance is RGBA, the sum over
 * accum_frame_count frames; out_rgb is linear Rec. 709 before the display
 * encoding, out_srgb 3 bytes per pixel after it. */
void tonemap_pixel(float out_rgb[3], const float hdr_radiance[4], float exposure, uint32_t accum_frame_count, tonemapper_t tonemapper);
void tonemap_image(uint8_t* out_srgb, const float* hdr_radiance, size_t pixel_count, float exposure, uint32_t accum_frame_count, tonemapper_t tonemapper);
/* Saves what the window of the reference would show for the current
 * accumulation (tonemapper and exposure of scene_spec) as PNG or JPG. */
int save_display_image(const char* file_path, image_file_format_t format, lit_scene_t* lit_scene, const render_targets_t* render_targets, const scene_spec_t* scene_spec);

/* Slides (src/main.h:120-150) */
#define MAX_SLIDE_COUNT 100
typedef struct {
	/* Quicksave to load, relative to the data root (e.g. data/saves/bistro/pretty.rt_save) */
	char* quicksave;
	render_settings_t render_settings;
	/* Where the screenshot goes, relative to the output root, or NULL */
	char* screenshot_path;
	image_file_format_t screenshot_format;
	/* The number of accumulated frames at which the screenshot is taken */
	uint32_t screenshot_frame;
} slide_t;
typedef struct {
	slide_t slides[MAX_SLIDE_COUNT];
	uint32_t slide_count;
	uint32_t slide_begin, slide_end;
	uint32_t slide_current;
} slideshow_t;
/* src/slides.c:5-142 */
uint32_t create_slides(slideshow_t* slideshow);
void free_slideshow(slideshow_t* slideshow);
/* Shows slides [slide_begin, slide_end) one after the other: quickload, render
 * settings, screenshot_frame frames (at most frame_limit if that is nonzero),
 * screenshot below output_root unless slide_screenshots is 0 (the
 * -no_screenshots switch of src/main.c:2110). */
int run_slideshow(slideshow_t* slideshow, int device_index, const char* data_root, const char* output_root, uint32_t width, uint32_t height, int slide_screenshots, uint32_t frame_limit);

#ifdef __cplusplus
}
#endif
#endif
