/* CPU oracle for the per-pixel light transport of MomentsInGraphics/path_tracer.
 *
 * THIS IS TEST INFRASTRUCTURE. It restates the reference's fragment shader
 * (src/shaders/pathtrace.frag.glsl and its includes) as scalar C so that the
 * HIP implementation can be checked against it. Nothing in the product
 * library links, loads or calls this code; only tests/, the smoke test and
 * the cpu_baseline leg of bench.py do.
 *
 * Parity status: the reference ships no tests, golden images or known-answer
 * vectors, and its renderer cannot be built in this environment (it needs
 * Vulkan with ray queries, GLFW, a display and glslangValidator). What pins
 * the oracle: (a) the published PCG2D algorithm and hand-derived known
 * answers, (b) facts parsed from the reference's own data files, (c) outputs
 * of the reference's own C where it compiles alone (math_utilities.c, the
 * texture converter, stb_image_write; oracle/Makefile target "ref", fixtures
 * under tests/golden/ref), (d) properties that hold for any correct
 * restatement: every density integrates to one, every sampler follows its
 * density, a surface under a uniform sky reflects the directional albedo that
 * a double-precision numpy restatement of frostbite_brdf() gives, and the four
 * estimators agree (tests/test_estimator_pins.py), (e) restatements in numpy
 * and double precision that share no code with this file: ray queries against
 * an intersection test over all triangles, get_shading_data() with both
 * dequantisations, the rays of all four cameras against unprojection and
 * spherical coordinates (tests/test_oracle_pins.py). Ray/triangle intersection
 * and texture filtering happen inside the Vulkan driver in the reference and
 * are "parity unpinned": the oracle defines them (watertight ray/triangle
 * test, float trilinear filtering of hand-decoded BC1/BC5 blocks at a level
 * of detail taken from the footprint of the path, a ray cone).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirror of constants_t, src/main.h:237-256 (784 bytes, std140 row_major). */
typedef struct {
	float world_to_projection_space[16];
	float projection_to_world_space[16];
	float camera_pos[3];
	int32_t camera_type;
	float hemispherical_camera_normal[3];
	float pad_1;
	float dequantization_factor[3], pad_2, dequantization_summand[3], pad_3;
	float viewport_size[2], inv_viewport_size[2];
	float exposure;
	uint32_t frame_index;
	uint32_t accum_frame_count;
	float pad_4;
	float sky_radiance[3];
	float pad_5;
	float emission_material_radiance[3];
	float pad_6;
	float params[4];
	float spherical_lights[32][4];
} oracle_constants_t;

/* One material texture: raw block-compressed mip payloads as stored in a
 * .vkt file (src/textures.c:8-33). format is the VkFormat number: 131
 * BC1_RGB_UNORM, 132 BC1_RGB_SRGB, 141 BC5_UNORM. */
typedef struct {
	uint32_t width, height, format, mip_count;
	const uint8_t* payload;      /* all mips back to back */
	const uint64_t* mip_offsets; /* byte offset of each mip in payload */
} oracle_texture_t;

typedef struct oracle_scene oracle_scene_t;

/* Render settings: sampling strategy enum of src/main.h:83-94. */
enum { ORACLE_STRATEGY_SPHERICAL = 0, ORACLE_STRATEGY_PSA = 1, ORACLE_STRATEGY_BRDF = 2, ORACLE_STRATEGY_NEE = 3 };

oracle_scene_t* oracle_create_scene(uint64_t material_count, uint64_t triangle_count,
	const float dequantization_factor[3], const float dequantization_summand[3],
	const uint32_t* quantized_positions, const uint16_t* normals_and_tex_coords, const uint8_t* material_indices,
	const oracle_texture_t* textures, uint32_t texture_count);
void oracle_destroy_scene(oracle_scene_t* scene);

/* Renders frame_count frames with frame indices first_frame_index +
 * i * frame_stride into accum (width*height RGBA float, top-left origin):
 * the first frame overwrites when accumulate == 0, every other frame adds
 * (src/main.c:1884-1888). Lights come from the separate array (x,y,z,r each),
 * not from constants->spherical_lights, so that more than 32 are possible.
 * Returns the number of rays traced (extension + shadow). */
uint64_t oracle_render(const oracle_scene_t* scene, const oracle_constants_t* constants,
	const float* lights, uint32_t light_count,
	int strategy, uint32_t path_length, uint32_t emission_material_index,
	uint32_t first_frame_index, uint32_t frame_count, uint32_t frame_stride,
	int accumulate, float* accum, int thread_count);

/* oracle_render() with a choice of how next event estimation picks its light:
 * 0 = the reference's loops over all lights (oracle_render() itself), 1 = a
 * descent through a binary tree over the lights. The second has no
 * counterpart in the reference; it restates the definition that the HIP
 * library documents for its light tree and serves to check that library's
 * arithmetic and, against mode 0, that the estimate converges to the same
 * image. */
uint64_t oracle_render_light_sampling(const oracle_scene_t* scene, const oracle_constants_t* constants,
	const float* lights, uint32_t light_count, int light_sampling,
	int strategy, uint32_t path_length, uint32_t emission_material_index,
	uint32_t first_frame_index, uint32_t frame_count, uint32_t frame_stride,
	int accumulate, float* accum, int thread_count);

/* Primary-ray closest hits for one frame: triangle index (-1 on miss),
 * barycentrics (weights of vertex 1 and 2) and front flag per pixel. */
void oracle_primary_hits(const oracle_scene_t* scene, const oracle_constants_t* constants,
	uint32_t frame_index, int32_t* triangle_indices, float* barycentrics, uint8_t* front);

/* The cone of every primary ray of one frame (width * height floats each):
 * the diameter of the footprint at the ray's origin and its growth per unit
 * distance, as the level-of-detail rule of this build defines them. */
void oracle_primary_cones(const oracle_constants_t* constants, uint32_t frame_index, float* widths, float* spreads);

/* The first draw_count outputs (2 floats each) of get_random_numbers() for
 * pixel (x, y) in frame frame_index, and the seed after each draw. */
void oracle_rng_stream(uint32_t x, uint32_t y, uint32_t frame_index, uint32_t draw_count, float* out_randoms, uint32_t* out_seeds);

/* Traces arbitrary rays (origin xyz, direction xyz per ray; tmin 1e-3, tmax
 * 1e38) and returns triangle index, t, barycentrics and front flag. */
void oracle_trace_rays(const oracle_scene_t* scene, const float* origins, const float* directions, uint32_t ray_count,
	int32_t* triangle_indices, float* t_u_v, uint8_t* front);

/* Evaluates get_shading_data() for given hits; out has 20 floats per hit in
 * the member order of shading_data_t (shading_data.glsl:19-39). footprints
 * (may be NULL = zeros) holds the diameter of the path's footprint at each
 * hit, which sets the level of detail of the texture lookups. */
void oracle_shading_data(const oracle_scene_t* scene, const oracle_constants_t* constants, uint32_t emission_material_index,
	const int32_t* triangle_indices, const float* barycentrics, const uint8_t* front, const float* out_dirs, const float* footprints, uint32_t count, float* out);

/* Samples texture texture_index at the given uv coordinates (2 floats
 * each) and levels of detail (NULL = level 0), trilinearly; writes 3 floats
 * per sample (BC5: r, g, 0). */
void oracle_sample_texture(const oracle_scene_t* scene, uint32_t texture_index, const float* uvs, const float* lods, uint32_t count, float* out);

/* One sampler, density or BRDF of the shaders for explicit arguments, for
 * tests of normalisation and of sampler against density. shading_data: the 20
 * floats of one shading_data_t; inputs and outputs: four floats per
 * evaluation (a sampler takes two random numbers and returns a world-space
 * direction, a density or the BRDF takes a world-space direction). function:
 * 0 sample_ggx_in_dir (brdfs.glsl:104-107, in the frame of :126-134),
 * 1 get_ggx_in_dir_density (:112-121), 2 sample_hemisphere_psa (:139-146),
 * 3 get_hemisphere_psa_density (:151-153), 4 sample_hemisphere_spherical
 * (pathtrace.frag.glsl:50-55), 5 get_hemisphere_spherical_density (:60-64),
 * 6 sample_frostbite_brdf (brdfs.glsl:174-199), 7 get_frostbite_brdf_density
 * (:203-208), 8 frostbite_brdf (:19-46), 9 sample_lights
 * (pathtrace.frag.glsl:120-143; fourth output: total importance),
 * 10 get_lights_density (:150-175; outputs: for a direction towards a light,
 * for any other direction, the total importance). */
void oracle_evaluate_function(int function, const float* shading_data, const float* lights, uint32_t light_count, const float* inputs, uint32_t count, float* outputs);

/* Elementary functions shared by definition with the HIP kernels. */
void oracle_sincos(float x, float* out_sin, float* out_cos);
float oracle_log(float x);
float oracle_erfinv(float x);

/* Host-side restatements (src/camera.c:102-167, src/math_utilities.c:16-64,
 * 197-203, src/main.c:629-676). camera is the 64-byte camera_t of
 * src/camera.h:42-63 inside scene_spec_t. */
typedef struct {
	float angles[3];
	uint8_t mouse_active, pad[3];
	float origin_angles[3];
	float position[3];
	float speed;
	float near_plane, far_plane;
	int32_t type;
	float fov;
	float height;
} oracle_camera_t;

/* scene_spec_t, src/main.h:55-78: the 128-byte .rt_save record. */
typedef struct {
	int32_t scene_file;
	oracle_camera_t camera;
	int32_t tonemapper;
	float exposure;
	uint32_t frame_index;
	float sky_color[3];
	float sky_strength;
	float emission_material_color[3];
	float emission_material_strength;
	float params[4];
} oracle_scene_spec_t;

void oracle_world_to_projection_space(float out[16], const oracle_camera_t* camera, float aspect_ratio);
void oracle_invert_mat4(float out[16], const float mat[16]);
void oracle_write_constants(oracle_constants_t* out, const oracle_scene_spec_t* spec, uint32_t width, uint32_t height,
	const float dequantization_factor[3], const float dequantization_summand[3],
	const float* lights, uint32_t light_count, uint32_t accum_frame_count);

#ifdef __cplusplus
}
#endif
#endif
