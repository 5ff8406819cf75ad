/* C ABI of the MI355X light transport library (libpt_hip.so).
 *
 * This is the boundary at which a C host shaped like the reference
 * (MomentsInGraphics/path_tracer) hands its scene to the GPU: what the
 * reference does with Vulkan buffers, an acceleration structure, a uniform
 * block and one full-screen draw per frame, a caller does here with plain
 * pointers and sizes. All functions return 0 on success and nonzero on
 * failure after printing a message to stdout, like the reference's
 * create_*() functions; none throws, exits or keeps host pointers beyond the
 * call. The library is single-caller per scene handle; calls on one handle
 * execute in call order.
 *
 * Citations name files of the reference relative to its root.
 */
#ifndef PT_HIP_H
#define PT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Counts changes of this header that a caller built against an older one
 * would not survive (arguments inserted, structures that grew). A binding
 * compares its own copy with pt_abi_version() of the library it loaded and
 * refuses to go on when they differ. No counterpart in the reference, whose
 * host and shaders are built together. */
#define PT_HIP_ABI_VERSION 3u
uint32_t pt_abi_version(void);

/* The number of GPUs that device_index / device_ids may name (0 when there
 * is none or the HIP runtime fails). Replaces the enumeration of physical
 * devices in create_device() at src/vulkan_basics.c:43-68. */
uint32_t pt_get_device_count(void);

/* The uniform block that the reference uploads once per frame:
 * constants_t at src/main.h:237-256 = src/shaders/constants.glsl:1-39
 * (784 bytes, std140, row_major matrices). frame_index and
 * accum_frame_count are ignored here: pt_render_samples() takes the frame
 * indices as arguments and the library counts accumulated frames itself. */
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
} pt_constants_t;

/* Numbers from the .vks header, scene_file_header_t at src/scene.h:5-21 */
typedef struct {
	uint64_t material_count;
	uint64_t triangle_count;
	float dequantization_factor[3];
	float dequantization_summand[3];
} pt_scene_header_t;

/* One texture as a .vkt file holds it (texture_header_t and mipmap_header_t
 * at src/textures.c:8-33): raw block-compressed mip levels. format is the
 * VkFormat number; 131 (BC1_RGB_UNORM), 132 (BC1_RGB_SRGB) and 141
 * (BC5_UNORM) are supported. Extents must be powers of two, at least 4.
 * Mip level m has the extent (width >> m) x (height >> m) and levels end at an
 * extent of 4, as the reference's converter writes them
 * (tools/texture_conversion/main.c:283-293). All levels are used: lookups
 * are trilinear, as the reference's sampler asks (src/main.c:864-873). */
typedef struct {
	uint32_t width, height;
	uint32_t format;
	uint32_t mip_count;
	const void* const* mip_data;
	const uint64_t* mip_sizes;
} pt_texture_desc_t;

/* sampling_strategy_t at src/main.h:83-94 */
enum {
	PT_SAMPLING_STRATEGY_SPHERICAL = 0,
	PT_SAMPLING_STRATEGY_PSA = 1,
	PT_SAMPLING_STRATEGY_BRDF = 2,
	PT_SAMPLING_STRATEGY_NEE = 3,
};

typedef struct pt_scene pt_scene_t;

/* Uploads a scene and builds its BVH on the GPU. Replaces load_scene() at
 * src/scene.c:282-404 from the point where the file has been read: the three
 * mesh arrays of src/scene.h:24-38 (positions: 3 x 2 uint32 per triangle,
 * normals_and_tex_coords: 3 x 4 uint16 per triangle, material_indices: 1
 * uint8 per triangle), create_bvh() at src/scene.c:90-264 and the texture
 * upload of load_textures() at src/textures.c:112-167. textures holds three
 * entries per material (base colour, specular, normal; src/scene.h:41-59).
 * device_index selects the GPU. On failure *out_scene is NULL. */
int pt_create_scene(pt_scene_t** out_scene, int device_index, const pt_scene_header_t* header,
	const uint32_t* positions, const uint16_t* normals_and_tex_coords, const uint8_t* material_indices,
	const pt_texture_desc_t* textures, uint32_t texture_count);

/* free_scene() at src/scene.c:407-420. NULL is allowed. */
void pt_destroy_scene(pt_scene_t* scene);

/* Replaces lit_scene_t.spherical_lights (src/main.h:276-283, filled at
 * src/main.c:687-698): count lights of four floats (x, y, z, radius). Unlike
 * the reference there is no limit of 32; the spherical_lights member of
 * pt_constants_t is ignored. */
int pt_set_lights(pt_scene_t* scene, const float* lights, uint32_t count);

/* How next event estimation (sampling strategy 3) chooses its light. EXACT is
 * the reference's sample_lights() and get_lights_density()
 * (src/shaders/pathtrace.frag.glsl:120-175): two loops over all lights per
 * path vertex, selection probability proportional to the solid angle. TREE
 * has no counterpart in the reference, which truncates to 32 lights
 * (src/main.c:691-694): one descent through a binary tree over the lights per
 * vertex and a lookup of the lights that a sampled direction meets in a grid
 * over them, for scenes with thousands of lights; the estimate stays
 * unbiased, its noise differs. A surface point inside the sphere of a light
 * takes no sample of that light under TREE (the reference's formulas at
 * :69-106 leave their domain there and EXACT follows them, negative radiance
 * included). The default is EXACT; the other strategies ignore it. */
enum { PT_LIGHT_SAMPLING_EXACT = 0, PT_LIGHT_SAMPLING_TREE = 1 };
int pt_set_light_sampling(pt_scene_t* scene, int light_sampling);

/* The arithmetic of the shading kernel under sampling strategy 3, with
 * either way of sampling lights. EXACT, the default, fuses a multiplication and an
 * addition only where the shaders say fma() and rounds every division and
 * square root correctly, as IEEE arithmetic does, which makes images equal a
 * CPU restatement of the shader bit for bit. FAST is what a GPU's shader
 * compiler is free to make of the reference's GLSL, which asks for no
 * particular rounding: fused multiply-adds, division and square root good to
 * about one unit in the last place, and the bare reciprocal and square root
 * instructions in the two loops over all lights of sample_lights()
 * (src/shaders/pathtrace.frag.glsl:69-83,120-143; under PT_LIGHT_SAMPLING_TREE
 * there are no such loops and only the first two freedoms apply). Random numbers, primary
 * rays and their hits stay the same bits; radiance differs like a few more
 * samples would change it (relative RMSE against EXACT at equal seeds below
 * 2 % at 32 samples per pixel and falling with the sample count; image means
 * within 0.1 %); shading takes a quarter less time. The environment variable
 * PT_SHADING_MATH=fast makes FAST the default of scenes created afterwards. */
enum { PT_SHADING_MATH_EXACT = 0, PT_SHADING_MATH_FAST = 1 };
int pt_set_shading_math(pt_scene_t* scene, int shading_math);

/* Replaces the shader defines SAMPLING_STRATEGY_*, PATH_LENGTH and
 * EMISSION_MATERIAL_INDEX of src/main.c:936-949 (render_settings_t at
 * src/main.h:99-105). path_length is at most 16. */
int pt_set_render_settings(pt_scene_t* scene, int sampling_strategy, uint32_t path_length, uint32_t emission_material_index);

/* create_render_targets() at src/main.c:519-560: allocates the RGBA32F
 * accumulation target for this resolution and resets the accumulated sample
 * count. The ray queues and path states (176 bytes per path) are allocated by
 * the first pt_render_samples() call for as many frames as it renders at
 * once: up to 64 M paths, 11.8 GB, when many samples are requested. */
int pt_create_render_targets(pt_scene_t* scene, uint32_t width, uint32_t height);

/* Renders sample_count frames; sample s uses g_frame_index =
 * first_frame_index + s * frame_stride. Replaces the per-frame uniform
 * upload and draw of record_render_frame_commands() at
 * src/main.c:1842-1891: the first frame overwrites the target when
 * accumulate is 0 (pipeline_discard), all others add to it (blend ONE, ONE,
 * alpha stays 1; src/main.c:1000-1021,1884-1888). Returns once the work is
 * queued; pt_synchronize() or pt_read_accumulation() wait for it. */
int pt_render_samples(pt_scene_t* scene, const pt_constants_t* constants,
	uint32_t first_frame_index, uint32_t sample_count, uint32_t frame_stride, int accumulate);

/* Waits for all queued work of this scene. */
int pt_synchronize(pt_scene_t* scene);

/* Copies the accumulation target to the host: width * height RGBA floats,
 * row-major from the top-left pixel, the sum (not the mean) over all
 * accumulated frames with alpha 1, as save_screenshot() finds it at
 * src/main.c:2043-2051. *out_sample_count receives the number of frames
 * accumulated (render_targets_t.accum_frame_count). Either may be NULL. */
int pt_read_accumulation(pt_scene_t* scene, float* rgba, uint32_t* out_sample_count);

/* The device address of the accumulation target (width * height * 4
 * floats) and a way to adjust the sample count, for hosts that sum the
 * targets of several GPUs themselves (RCCL all-reduce). No counterpart in
 * the reference, which is single-GPU. */
void* pt_get_accumulation_device_pointer(pt_scene_t* scene);
int pt_set_sample_count(pt_scene_t* scene, uint32_t sample_count);

/* Several GPUs of one node. No counterpart in the reference, which renders
 * on one GPU. Samples are independent given pixel and frame index
 * (src/shaders/pathtrace.frag.glsl:368): a context holds a replica of the
 * scene on every GPU of device_ids, GPU g of n renders the frame indices
 * first + g, first + g + n, ... into its own accumulation target, and
 * pt_context_reduce() sums the targets over xGMI (one RCCL reduction) into
 * the target of the first GPU. The union of the slices is exactly the set of
 * frames a single GPU would render; only the order of the float additions
 * differs. The scene arguments are those of pt_create_scene(); lights,
 * settings and render targets are set for all replicas at once, with the
 * meaning of the pt_set_*() / pt_create_render_targets() calls above.
 * pt_context_scene(context, i) is the replica on GPU device_ids[i], for
 * calls per GPU (statistics, BVH info) and, with i = 0 after
 * pt_context_reduce(), for pt_read_accumulation(). A context with one GPU
 * renders the same bits as that GPU's scene alone. */
typedef struct pt_context pt_context_t;
int pt_create_context(pt_context_t** out_context, const int* device_ids, uint32_t device_count, const pt_scene_header_t* header,
	const uint32_t* positions, const uint16_t* normals_and_tex_coords, const uint8_t* material_indices,
	const pt_texture_desc_t* textures, uint32_t texture_count);
void pt_destroy_context(pt_context_t* context);
uint32_t pt_context_device_count(const pt_context_t* context);
pt_scene_t* pt_context_scene(pt_context_t* context, uint32_t index);
int pt_context_set_lights(pt_context_t* context, const float* lights, uint32_t count);
int pt_context_set_light_sampling(pt_context_t* context, int light_sampling);
int pt_context_set_shading_math(pt_context_t* context, int shading_math);
int pt_context_set_render_settings(pt_context_t* context, int sampling_strategy, uint32_t path_length, uint32_t emission_material_index);
int pt_context_create_render_targets(pt_context_t* context, uint32_t width, uint32_t height);
/* pt_render_samples() for sample_count frames in total, shared out as
 * described above. With accumulate 0 the targets of all GPUs start over. */
int pt_context_render_samples(pt_context_t* context, const pt_constants_t* constants, uint32_t first_frame_index, uint32_t sample_count, int accumulate);
int pt_context_synchronize(pt_context_t* context);
/* Waits for the GPUs, then sums their targets into that of the first one
 * (alpha stays 1, its sample count becomes the total) and clears the others,
 * so that rendering may go on and be reduced again without counting
 * anything twice. pt_context_last_reduce_ms() is the wall time the last
 * reduction took (sum, alpha, clears), for throughput reports. */
int pt_context_reduce(pt_context_t* context);
double pt_context_last_reduce_ms(const pt_context_t* context);
/* Which of sample_count frames from first_frame_index on GPU gpu of
 * gpu_count renders in pt_context_render_samples(): *out_sample_count frames
 * (possibly none), the first at *out_first_frame_index, the others gpu_count
 * apart. Host arithmetic only; needs no GPU. */
int pt_context_frame_slice(uint32_t gpu, uint32_t gpu_count, uint32_t first_frame_index, uint32_t sample_count, uint32_t* out_first_frame_index, uint32_t* out_sample_count);

/* Counters and timings since the last reset. Ray counts are the numbers of
 * extension and shadow rays actually traced (queue sizes). Node and triangle
 * visits are only counted while profiling is enabled. Kernel times are sums
 * of HIP event intervals on the library's stream, recorded only while timing
 * is enabled. No counterpart in the reference besides the two timestamps
 * around its draw (src/main.c:1890-1892). */
typedef struct {
	uint64_t extension_rays;
	uint64_t shadow_rays;
	uint64_t paths;
	/* index 0: extension rays, 1: shadow rays */
	uint64_t nodes_visited[2];
	uint64_t triangles_tested[2];
	uint64_t shaded_hits;
	/* index: 0 raygen, 1 extension trace, 2 shadow trace, 3 shade, 4 resolve */
	double kernel_ms[5];
	uint64_t kernel_launches[5];
	/* Whether shadow rays skip what cannot change their answer: 0 not decided
	 * yet (they do, and the next batch decides), 1 yes, 2 no;
	 * shadow_skipping_fixed is 1 when the environment variable
	 * PT_SHADOW_SKIPPING fixed it. The decision was taken from
	 * shadow_rays_decided_on rays of which shadow_rays_blocked ended at
	 * something that is not emissive. It changes speed only, never the image;
	 * reported so that timings of two runs can be compared. */
	uint32_t shadow_skipping;
	uint32_t shadow_skipping_fixed;
	uint64_t shadow_rays_decided_on;
	uint64_t shadow_rays_blocked;
} pt_statistics_t;
int pt_set_instrumentation(pt_scene_t* scene, int enable_kernel_timing, int enable_visit_counters);
int pt_get_statistics(pt_scene_t* scene, pt_statistics_t* out_statistics);
int pt_reset_statistics(pt_scene_t* scene);

/* Facts about the BVH that was built: numbers of 8-wide nodes and of
 * triangles in its arrays in HBM, levels, bytes per node and per triangle,
 * and the time the build took on the GPU. */
typedef struct {
	uint32_t node_count;
	uint32_t triangle_count;
	uint32_t depth;
	uint32_t node_bytes;
	uint32_t triangle_bytes;
	float build_ms;
} pt_bvh_info_t;
int pt_get_bvh_info(pt_scene_t* scene, pt_bvh_info_t* out_info);

/* Test hooks: the closest hit of every primary ray of one frame in pixel
 * order (what rayQueryGetIntersection*EXT return at
 * src/shaders/pathtrace.frag.glsl:187-195), and of caller-supplied rays
 * (origin and direction, 3 floats each; tmin 1e-3, tmax 1e38).
 * triangle_indices receives -1 for a miss; t_u_v three floats per ray. */
int pt_trace_primary_rays(pt_scene_t* scene, const pt_constants_t* constants, uint32_t frame_index,
	int32_t* triangle_indices, float* barycentrics, uint8_t* front);
int pt_trace_rays(pt_scene_t* scene, const float* origins, const float* directions, uint32_t ray_count,
	int32_t* triangle_indices, float* t_u_v, uint8_t* front);

/* pt_trace_rays() without any allocation. It has no counterpart in the
 * reference; it is for callers that cannot allocate device memory: tight
 * serving loops, harnesses that forbid hipMalloc. (It is not a call to record
 * into a HIP graph: it launches on the library's own stream, waits for it and
 * reads the library's error word back.)
 * All five arrays are device memory of the caller on the scene's GPU
 * (origins and directions three floats per ray, results as pt_trace_rays()
 * returns them). The ray queue and its counters live in an array that is
 * part of the library's code object (33 MiB per GPU, held until the process
 * ends), hence at most 1048576 rays per call; more are refused with a
 * message. Calls on one GPU share that array: calls that overlap in time on
 * one GPU, from whatever scene or thread, race. Returns when the results are
 * complete. */
#define PT_TRACE_RAYS_NO_ALLOC_MAX_RAYS 1048576u
int pt_trace_rays_no_alloc(pt_scene_t* scene, const float* device_origins, const float* device_directions, uint32_t ray_count,
	int32_t* device_triangle_indices, float* device_t_u_v, uint8_t* device_front);

/* Test hook: get_shading_data() of src/shaders/shading_data.glsl:51-103 for
 * given hits; 20 floats per hit in the member order of shading_data_t.
 * footprints holds, per hit, the diameter of the path's footprint on the
 * surface in scene units, from which the level of detail of the texture
 * lookups follows (what the implicit derivatives of texture() are to the
 * reference); NULL stands for zeros, i.e. mip level 0. */
int pt_evaluate_shading_data(pt_scene_t* scene, const pt_constants_t* constants,
	const int32_t* triangle_indices, const float* barycentrics, const uint8_t* front, const float* out_dirs, const float* footprints,
	uint32_t count, float* out_shading_data);

/* Test hook: one sampler, density or BRDF of the shaders for explicit
 * arguments, so that normalisation and the agreement of every sampler with
 * its density can be tested without rendering. shading_data holds the 20
 * floats of one shading_data_t (src/shaders/shading_data.glsl:19-39) shared by
 * all evaluations; inputs and outputs hold four floats per evaluation. A
 * sampler takes two random numbers (x, y) and returns a world-space direction
 * (x, y, z); a density or the BRDF takes a world-space direction (x, y, z).
 * The lights are those of pt_set_lights(). */
enum {
	PT_FUNCTION_SAMPLE_GGX_IN_DIR = 0,          /* src/shaders/brdfs.glsl:104-107 in the frame of :126-134 */
	PT_FUNCTION_GGX_IN_DIR_DENSITY = 1,         /* brdfs.glsl:112-121 -> x */
	PT_FUNCTION_SAMPLE_HEMISPHERE_PSA = 2,      /* brdfs.glsl:139-146 */
	PT_FUNCTION_HEMISPHERE_PSA_DENSITY = 3,     /* brdfs.glsl:151-153 -> x */
	PT_FUNCTION_SAMPLE_HEMISPHERE_SPHERICAL = 4, /* src/shaders/pathtrace.frag.glsl:50-55 */
	PT_FUNCTION_HEMISPHERE_SPHERICAL_DENSITY = 5, /* pathtrace.frag.glsl:60-64 -> x */
	PT_FUNCTION_SAMPLE_FROSTBITE_BRDF = 6,      /* brdfs.glsl:174-199 */
	PT_FUNCTION_FROSTBITE_BRDF_DENSITY = 7,     /* brdfs.glsl:203-208 -> x */
	PT_FUNCTION_FROSTBITE_BRDF = 8,             /* brdfs.glsl:19-46 -> x, y, z */
	PT_FUNCTION_SAMPLE_LIGHTS = 9,              /* pathtrace.frag.glsl:120-143; w receives the total importance */
	PT_FUNCTION_LIGHTS_DENSITY = 10,            /* pathtrace.frag.glsl:150-175: x for a direction towards a light, y for any other, z the total importance */
	PT_FUNCTION_COUNT = 11
};
int pt_evaluate_function(pt_scene_t* scene, int function, const float* shading_data, const float* inputs, uint32_t count, float* outputs);

/* Test hook: the first draw_count pairs of get_random_numbers()
 * (src/shaders/pathtrace.frag.glsl:23-34) for the seed of pixel (x, y) in
 * frame frame_index (:368), computed on the GPU. */
int pt_random_numbers(int device_index, uint32_t x, uint32_t y, uint32_t frame_index, uint32_t draw_count, float* out_randoms);

#ifdef __cplusplus
}
#endif
#endif
