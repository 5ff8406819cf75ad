/* Reader for *.vkt texture files (src/textures.c of the reference). The
 * block-compressed mipmaps stay as they are in the file; the GPU decodes them
 * by hand, so there is no image object to create. */
#include "pt_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* src/textures.c:63-94 plus the payload that write_mipmap() (:99-106) streams
 * to the GPU in the reference. Returns 0 upon success; the caller frees. */
static int init_texture(texture_t* texture, const char* texture_file_path) {
	memset(texture, 0, sizeof(*texture));
	FILE* file = fopen(texture_file_path, "rb");
	if (!file) {
		printf("Failed to open the texture file at %s. Please check path and permissions.\n", texture_file_path);
		return 1;
	}
	uint32_t marker = 0, version = 0;
	uint64_t payload_size = 0;
	int result = 1;
	do {
		if (fread(&marker, sizeof(uint32_t), 1, file) != 1 || marker != 0xbc1bc1) {
			printf("The file at %s does not appear to be a valid *.vkt texture file as it does not start with the appropriate file marker. Images from widely used formats must be piped through a texture conversion tool before loading them.\n", texture_file_path);
			break;
		}
		if (fread(&version, sizeof(uint32_t), 1, file) != 1 || version != 1) {
			printf("The texture file at %s uses file format version %u, which is not supported.\n", texture_file_path, version);
			break;
		}
		uint32_t extent[2];
		if (fread(&texture->mipmap_count, sizeof(uint32_t), 1, file) != 1 || fread(extent, sizeof(uint32_t), 2, file) != 2
			|| fread(&texture->format, sizeof(uint32_t), 1, file) != 1 || fread(&payload_size, sizeof(uint64_t), 1, file) != 1
			|| texture->mipmap_count == 0 || texture->mipmap_count > 32)
		{
			printf("The header of the texture file at %s is incomplete or invalid.\n", texture_file_path);
			break;
		}
		texture->width = extent[0];
		texture->height = extent[1];
		texture->mipmap_sizes = calloc(texture->mipmap_count, sizeof(uint64_t));
		texture->mipmap_data = calloc(texture->mipmap_count, sizeof(void*));
		uint64_t* offsets = calloc(texture->mipmap_count, sizeof(uint64_t));
		bool headers_ok = true;
		for (uint32_t i = 0; i != texture->mipmap_count; ++i) {
			uint32_t mip_extent[2];
			if (fread(mip_extent, sizeof(uint32_t), 2, file) != 2 || fread(&texture->mipmap_sizes[i], sizeof(uint64_t), 1, file) != 1
				|| fread(&offsets[i], sizeof(uint64_t), 1, file) != 1
				/* written so that no sum can wrap around */
				|| offsets[i] > payload_size || texture->mipmap_sizes[i] > payload_size - offsets[i])
				headers_ok = false;
		}
		/* The payload cannot be larger than what is left of the file */
		long payload_begin = ftell(file);
		uint64_t bytes_left = 0;
		if (payload_begin >= 0 && fseek(file, 0, SEEK_END) == 0) {
			long file_end = ftell(file);
			if (file_end >= payload_begin)
				bytes_left = (uint64_t) (file_end - payload_begin);
			if (fseek(file, payload_begin, SEEK_SET) != 0)
				bytes_left = 0;
		}
		if (payload_size > bytes_left)
			headers_ok = false;
		if (headers_ok)
			texture->payload = malloc(payload_size ? payload_size : 1);
		uint32_t eof_marker = 0;
		if (!headers_ok || !texture->payload || fread(texture->payload, 1, payload_size, file) != payload_size
			|| fread(&eof_marker, sizeof(eof_marker), 1, file) != 1 || eof_marker != 0xe0fe0f)
		{
			printf("The texture file at %s is truncated or its mipmap headers are inconsistent.\n", texture_file_path);
			free(offsets);
			break;
		}
		for (uint32_t i = 0; i != texture->mipmap_count; ++i)
			texture->mipmap_data[i] = texture->payload + offsets[i];
		free(offsets);
		result = 0;
	} while (0);
	fclose(file);
	return result;
}

/* src/textures.c:112-167 */
int load_textures(textures_t* textures, const char* const* texture_file_paths, uint32_t texture_count) {
	memset(textures, 0, sizeof(*textures));
	textures->texture_count = texture_count;
	textures->textures = calloc(texture_count ? texture_count : 1, sizeof(texture_t));
	for (uint32_t i = 0; i != texture_count; ++i) {
		if (init_texture(&textures->textures[i], texture_file_paths[i])) {
			printf("Failed to load texture %u out of %u. Its file path is %s. Aborting.\n", i, texture_count, texture_file_paths[i]);
			free_textures(textures);
			return 1;
		}
	}
	return 0;
}

/* src/textures.c:170-181 */
void free_textures(textures_t* textures) {
	if (textures->textures) {
		for (uint32_t i = 0; i != textures->texture_count; ++i) {
			free(textures->textures[i].mipmap_sizes);
			free(textures->textures[i].mipmap_data);
			free(textures->textures[i].payload);
		}
	}
	free(textures->textures);
	memset(textures, 0, sizeof(*textures));
}
