/* Reader for *.vks scene files (src/scene.c:282-404 of the reference). Where
 * the reference fills Vulkan buffers, builds acceleration structures and
 * creates images, this loader hands the three mesh arrays and the raw
 * textures to pt_create_scene(), which uploads them and builds the BVH on the
 * GPU. */
#include "pt_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Where the texture of one kind of a material is looked for: <directory>/<material><suffix>, with the suffixes of
   src/scene.c:380-383. The caller frees the string. */
static char* texture_file_path(const char* directory, const char* material_name, material_texture_type_t type) {
	static const char* const suffixes[material_texture_type_count] = {
		[material_texture_type_base_color] = "_BaseColor.vkt",
		[material_texture_type_specular] = "_Specular.vkt",
		[material_texture_type_normal] = "_Normal.vkt",
	};
	size_t size = strlen(directory) + 1 + strlen(material_name) + strlen(suffixes[type]) + 1;
	char* path = malloc(size);
	if (path)
		snprintf(path, size, "%s/%s%s", directory, material_name, suffixes[type]);
	return path;
}

static int load_scene_on(scene_t* scene, const int* device_indices, uint32_t device_count, bool use_context, const char* file_path, const char* texture_path);

int load_scene(scene_t* scene, int device_index, const char* file_path, const char* texture_path) {
	return load_scene_on(scene, &device_index, 1, false, file_path, texture_path);
}

/* load_scene() for a list of GPUs: the scene is replicated through a
 * pt_context_t (frames are shared out among the GPUs and summed over xGMI,
 * also when the list has one entry); device_scene is the replica on the
 * first GPU. */
int load_scene_devices(scene_t* scene, const int* device_indices, uint32_t device_count, const char* file_path, const char* texture_path) {
	return load_scene_on(scene, device_indices, device_count, true, file_path, texture_path);
}

static int load_scene_on(scene_t* scene, const int* device_indices, uint32_t device_count, bool use_context, const char* file_path, const char* texture_path) {
	memset(scene, 0, sizeof(*scene));
	if (!device_indices || device_count == 0) {
		printf("Loading the scene file at %s needs at least one GPU.\n", file_path);
		return 1;
	}
	if (pt_abi_version() != PT_HIP_ABI_VERSION) {
		printf("libpt_hip.so has ABI version %u but this host was built against version %u of pt_hip.h. Please rebuild both.\n", pt_abi_version(), PT_HIP_ABI_VERSION);
		return 1;
	}
	FILE* file = fopen(file_path, "rb");
	if (!file) {
		printf("Failed to open the scene file at %s. Please check path and permissions.\n", file_path);
		return 1;
	}
	uint32_t* positions = NULL;
	uint16_t* normals_and_tex_coords = NULL;
	uint8_t* material_indices = NULL;
	textures_t textures;
	memset(&textures, 0, sizeof(textures));
	char** texture_file_paths = NULL;
	uint64_t texture_count = 0;
	int result = 1;
	do {
		scene_file_header_t* header = &scene->header;
		/* src/scene.c:293-308 */
		if (fread(&header->marker, sizeof(uint32_t), 1, file) != 1 || header->marker != 0xabcabc) {
			printf("The scene file at %s is not a valid *.vks file. Its marker does not match.\n", file_path);
			break;
		}
		if (fread(&header->version, sizeof(uint32_t), 1, file) != 1 || header->version != 1) {
			printf("This renderer only supports *.vks files using version 1 of the file format, but the scene file at %s uses version %u.\n", file_path, header->version);
			break;
		}
		if (fread(&header->material_count, sizeof(uint64_t), 1, file) != 1 || fread(&header->triangle_count, sizeof(uint64_t), 1, file) != 1
			|| fread(header->dequantization_factor, sizeof(float), 3, file) != 3 || fread(header->dequantization_summand, sizeof(float), 3, file) != 3
			|| header->material_count > 256 || header->triangle_count > 0x55555555ull)
		{
			printf("The header of the scene file at %s is incomplete or out of range.\n", file_path);
			break;
		}
		/* src/scene.c:310-316 */
		header->material_names = calloc(header->material_count ? header->material_count : 1, sizeof(char*));
		bool names_ok = true;
		for (uint64_t i = 0; i != header->material_count && names_ok; ++i) {
			uint64_t length = 0;
			if (fread(&length, sizeof(length), 1, file) != 1 || length > 4096) {
				names_ok = false;
				break;
			}
			header->material_names[i] = calloc(length + 1, 1);
			if (fread(header->material_names[i], sizeof(char), length + 1, file) != length + 1)
				names_ok = false;
			header->material_names[i][length] = 0;
		}
		if (!names_ok) {
			printf("Failed to read the material names of the scene file at %s.\n", file_path);
			break;
		}
		/* src/scene.c:318-359: the three mesh arrays */
		uint64_t t = header->triangle_count;
		positions = malloc(t ? t * 24 : 1);
		normals_and_tex_coords = malloc(t ? t * 24 : 1);
		material_indices = malloc(t ? t : 1);
		if (fread(positions, 24, t, file) != t || fread(normals_and_tex_coords, 24, t, file) != t || fread(material_indices, 1, t, file) != t) {
			printf("Failed to read mesh data of the scene file at %s with %lu triangles.\n", file_path, (unsigned long) t);
			break;
		}
		/* src/scene.c:360-367 */
		uint32_t eof_marker = 0;
		if (fread(&eof_marker, sizeof(eof_marker), 1, file) != 1 || eof_marker != 0xe0fe0f) {
			printf("Finished reading data from the scene file at %s but did not encounter an end-of-file marker where expected. Either the file is invalid or the loader is buggy.\n", file_path);
			break;
		}
		/* src/scene.c:377-399: three textures per material */
		texture_count = header->material_count * material_texture_type_count;
		texture_file_paths = calloc(texture_count ? texture_count : 1, sizeof(char*));
		bool paths_complete = texture_file_paths != NULL;
		for (uint64_t k = 0; paths_complete && k != texture_count; ++k) {
			texture_file_paths[k] = texture_file_path(texture_path, header->material_names[k / material_texture_type_count], (material_texture_type_t) (k % material_texture_type_count));
			paths_complete = texture_file_paths[k] != NULL;
		}
		if (!paths_complete) {
			printf("Out of memory for the texture file names of the scene file at %s.\n", file_path);
			break;
		}
		if (load_textures(&textures, (const char* const*) texture_file_paths, (uint32_t) texture_count)) {
			printf("Failed to load textures for the scene file at %s.\n", file_path);
			break;
		}
		/* src/main.c:936-939 */
		for (uint32_t i = 0; i != header->material_count; ++i)
			if (strcmp(header->material_names[i], "_emission") == 0)
				scene->emission_material_index = i;
		/* Upload and BVH build (create_bvh() at src/scene.c:90-264) */
		pt_texture_desc_t* descs = calloc(texture_count ? texture_count : 1, sizeof(pt_texture_desc_t));
		for (uint64_t i = 0; i != texture_count; ++i) {
			const texture_t* texture = &textures.textures[i];
			descs[i].width = texture->width;
			descs[i].height = texture->height;
			descs[i].format = texture->format;
			descs[i].mip_count = texture->mipmap_count;
			descs[i].mip_data = (const void* const*) texture->mipmap_data;
			descs[i].mip_sizes = texture->mipmap_sizes;
		}
		pt_scene_header_t device_header = { .material_count = header->material_count, .triangle_count = header->triangle_count };
		memcpy(device_header.dequantization_factor, header->dequantization_factor, sizeof(float) * 3);
		memcpy(device_header.dequantization_summand, header->dequantization_summand, sizeof(float) * 3);
		int create_result;
		if (!use_context)
			create_result = pt_create_scene(&scene->device_scene, device_indices[0], &device_header, positions, normals_and_tex_coords, material_indices, descs, (uint32_t) texture_count);
		else {
			create_result = pt_create_context(&scene->device_context, device_indices, device_count, &device_header, positions, normals_and_tex_coords, material_indices, descs, (uint32_t) texture_count);
			if (!create_result)
				scene->device_scene = pt_context_scene(scene->device_context, 0);
		}
		free(descs);
		if (create_result) {
			printf("Failed to create ray-tracing acceleration structures for the scene file at %s.\n", file_path);
			break;
		}
		result = 0;
	} while (0);
	fclose(file);
	free(positions);
	free(normals_and_tex_coords);
	free(material_indices);
	free_textures(&textures);
	if (texture_file_paths)
		for (uint64_t i = 0; i != texture_count; ++i)
			free(texture_file_paths[i]);
	free(texture_file_paths);
	if (result)
		free_scene(scene);
	return result;
}

/* src/scene.c:407-420 */
void free_scene(scene_t* scene) {
	/* The replicas of a context belong to the context */
	if (scene->device_context)
		pt_destroy_context(scene->device_context);
	else
		pt_destroy_scene(scene->device_scene);
	if (scene->header.material_names)
		for (uint64_t i = 0; i != scene->header.material_count; ++i)
			free(scene->header.material_names[i]);
	free(scene->header.material_names);
	memset(scene, 0, sizeof(*scene));
}
