/* Minimal image encoders for screenshots: Radiance RGBE (*.hdr) and PNG with
 * stored (uncompressed) deflate blocks. They take the place of the vendored
 * stb_image_write.h in the reference (src/main.c:2053, :2077). */
#include "pt_host.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int write_hdr(const char* file_path, uint32_t width, uint32_t height, const float* rgb) {
	FILE* file = fopen(file_path, "wb");
	if (!file)
		return 1;
	fprintf(file, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %u +X %u\n", height, width);
	uint8_t* row = malloc(4 * (size_t) width);
	for (uint32_t y = 0; y != height; ++y) {
		for (uint32_t x = 0; x != width; ++x) {
			const float* p = &rgb[3 * ((size_t) y * width + x)];
			float max_comp = fmaxf(p[0], fmaxf(p[1], p[2]));
			uint8_t* out = &row[4 * x];
			if (!(max_comp >= 1e-32f))
				out[0] = out[1] = out[2] = out[3] = 0;
			else {
				int exponent;
				float scale = frexpf(max_comp, &exponent) * 256.0f / max_comp;
				out[0] = (uint8_t) (p[0] > 0.0f ? p[0] * scale : 0.0f);
				out[1] = (uint8_t) (p[1] > 0.0f ? p[1] * scale : 0.0f);
				out[2] = (uint8_t) (p[2] > 0.0f ? p[2] * scale : 0.0f);
				out[3] = (uint8_t) (exponent + 128);
			}
		}
		fwrite(row, 4, width, file);
	}
	free(row);
	return fclose(file) != 0;
}

static uint32_t crc_table[256];
static void init_crc_table(void) {
	if (crc_table[1])
		return;
	for (uint32_t n = 0; n != 256; ++n) {
		uint32_t c = n;
		for (int k = 0; k != 8; ++k)
			c = (c & 1) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
		crc_table[n] = c;
	}
}

static uint32_t crc32_update(uint32_t crc, const uint8_t* data, size_t size) {
	for (size_t i = 0; i != size; ++i)
		crc = crc_table[(crc ^ data[i]) & 0xff] ^ (crc >> 8);
	return crc;
}

static void put_be32(uint8_t* out, uint32_t v) {
	out[0] = (uint8_t) (v >> 24); out[1] = (uint8_t) (v >> 16); out[2] = (uint8_t) (v >> 8); out[3] = (uint8_t) v;
}

static void write_chunk(FILE* file, const char* type, const uint8_t* data, uint32_t size) {
	uint8_t head[8];
	put_be32(head, size);
	memcpy(head + 4, type, 4);
	fwrite(head, 1, 8, file);
	if (size)
		fwrite(data, 1, size, file);
	uint32_t crc = crc32_update(0xffffffffu, head + 4, 4);
	crc = crc32_update(crc, data, size) ^ 0xffffffffu;
	uint8_t tail[4];
	put_be32(tail, crc);
	fwrite(tail, 1, 4, file);
}

int write_png(const char* file_path, uint32_t width, uint32_t height, const uint8_t* rgb) {
	init_crc_table();
	FILE* file = fopen(file_path, "wb");
	if (!file)
		return 1;
	static const uint8_t signature[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
	fwrite(signature, 1, 8, file);
	uint8_t ihdr[13];
	put_be32(ihdr, width);
	put_be32(ihdr + 4, height);
	ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
	write_chunk(file, "IHDR", ihdr, 13);
	/* Raw scanlines with filter type 0, wrapped into stored deflate blocks */
	size_t stride = 1 + 3 * (size_t) width;
	size_t raw_size = stride * height;
	size_t block_count = (raw_size + 65534) / 65535;
	if (block_count == 0) block_count = 1;
	size_t zsize = 2 + raw_size + 5 * block_count + 4;
	if (zsize > 0x7fffffffu) {
		fclose(file);
		return 1;
	}
	uint8_t* z = malloc(zsize);
	size_t o = 0;
	z[o++] = 0x78; z[o++] = 0x01;
	uint32_t a = 1, b = 0;
	size_t done = 0;
	for (size_t block = 0; block != block_count; ++block) {
		size_t n = raw_size - done;
		if (n > 65535) n = 65535;
		z[o++] = (block + 1 == block_count) ? 1 : 0;
		z[o++] = (uint8_t) n; z[o++] = (uint8_t) (n >> 8);
		z[o++] = (uint8_t) ~n; z[o++] = (uint8_t) (~n >> 8);
		for (size_t i = 0; i != n; ++i) {
			size_t pos = done + i;
			size_t y = pos / stride, x = pos % stride;
			uint8_t v = (x == 0) ? 0 : rgb[3 * y * (size_t) width + (x - 1)];
			z[o++] = v;
			a = (a + v) % 65521u;
			b = (b + a) % 65521u;
		}
		done += n;
	}
	put_be32(&z[o], (b << 16) | a);
	o += 4;
	write_chunk(file, "IDAT", z, (uint32_t) o);
	write_chunk(file, "IEND", NULL, 0);
	free(z);
	return fclose(file) != 0;
}
