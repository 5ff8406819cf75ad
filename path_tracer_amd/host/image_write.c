/* Image encoders for screenshots: Radiance RGBE (*.hdr), PNG (filtered
 * scanlines, deflate with fixed Huffman codes) and baseline JPEG. They take
 * the place of the vendored stb_image_write.h in the reference
 * (src/main.c:2053, :2077). */
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

/* ---- Deflate (RFC 1951) with the fixed Huffman codes and greedy LZ77
 * matching over hash chains, enough for screenshots: smooth images shrink to a
 * fraction, noise grows by an eighth at worst. ---- */
typedef struct {
	uint8_t* data;
	size_t size;
	uint32_t bit_buffer;
	int bit_count;
} deflate_writer_t;

/* Bits are packed starting from the least significant bit of each byte */
static void deflate_put_bits(deflate_writer_t* w, uint32_t bits, int count) {
	w->bit_buffer |= bits << w->bit_count;
	w->bit_count += count;
	while (w->bit_count >= 8) {
		w->data[w->size++] = (uint8_t) w->bit_buffer;
		w->bit_buffer >>= 8;
		w->bit_count -= 8;
	}
}

/* Huffman codes go out starting from their most significant bit */
static void deflate_put_code(deflate_writer_t* w, uint32_t code, int length) {
	uint32_t reversed = 0;
	for (int i = 0; i != length; ++i)
		reversed |= ((code >> i) & 1u) << (length - 1 - i);
	deflate_put_bits(w, reversed, length);
}

/* A literal byte or a length symbol (256: end of block) in the fixed code */
static void deflate_put_symbol(deflate_writer_t* w, uint32_t symbol) {
	if (symbol < 144) deflate_put_code(w, 0x30 + symbol, 8);
	else if (symbol < 256) deflate_put_code(w, 0x190 + (symbol - 144), 9);
	else if (symbol < 280) deflate_put_code(w, symbol - 256, 7);
	else deflate_put_code(w, 0xc0 + (symbol - 280), 8);
}

static const uint16_t deflate_length_base[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint8_t deflate_length_extra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t deflate_distance_base[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
static const uint8_t deflate_distance_extra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

static void deflate_put_match(deflate_writer_t* w, uint32_t length, uint32_t distance) {
	int l = 28;
	while (deflate_length_base[l] > length) --l;
	deflate_put_symbol(w, 257 + (uint32_t) l);
	deflate_put_bits(w, length - deflate_length_base[l], deflate_length_extra[l]);
	int d = 29;
	while (deflate_distance_base[d] > distance) --d;
	deflate_put_code(w, (uint32_t) d, 5);
	deflate_put_bits(w, distance - deflate_distance_base[d], deflate_distance_extra[d]);
}

/* Compresses size bytes into a zlib stream (RFC 1950); the caller frees the
 * result. Returns NULL if memory runs out. */
static uint8_t* zlib_compress(const uint8_t* raw, size_t size, size_t* out_size) {
	enum { WINDOW = 32768, HASH_SIZE = 1 << 15, MAX_CHAIN = 32, MIN_MATCH = 3, MAX_MATCH = 258 };
	deflate_writer_t w;
	/* Nine bits per byte at worst, plus header, end of block and checksum */
	w.data = malloc(size + size / 8 + 64);
	int32_t* head = malloc(sizeof(int32_t) * HASH_SIZE);
	int32_t* previous = malloc(sizeof(int32_t) * WINDOW);
	if (!w.data || !head || !previous || size > 0x7fffffff) {
		free(w.data); free(head); free(previous);
		return NULL;
	}
	w.size = 0; w.bit_buffer = 0; w.bit_count = 0;
	w.data[w.size++] = 0x78; w.data[w.size++] = 0x9c;
	deflate_put_bits(&w, 1, 1); /* last block */
	deflate_put_bits(&w, 1, 2); /* fixed Huffman codes */
	for (int i = 0; i != HASH_SIZE; ++i) head[i] = -1;
	size_t position = 0;
	while (position < size) {
		uint32_t best_length = 0, best_distance = 0;
		if (position + MIN_MATCH <= size) {
			uint32_t hash = ((uint32_t) raw[position] << 10 ^ (uint32_t) raw[position + 1] << 5 ^ raw[position + 2]) & (HASH_SIZE - 1);
			int32_t candidate = head[hash];
			size_t limit = size - position;
			if (limit > MAX_MATCH) limit = MAX_MATCH;
			for (int chain = 0; chain != MAX_CHAIN && candidate >= 0 && position - (size_t) candidate <= WINDOW; ++chain) {
				uint32_t length = 0;
				while (length < limit && raw[(size_t) candidate + length] == raw[position + length]) ++length;
				if (length > best_length) {
					best_length = length;
					best_distance = (uint32_t) (position - (size_t) candidate);
					if (length == limit) break;
				}
				int32_t next = previous[candidate & (WINDOW - 1)];
				/* Chains end where an entry has been overwritten by a later position */
				if (next >= candidate) break;
				candidate = next;
			}
		}
		size_t step = 1;
		if (best_length >= MIN_MATCH) {
			deflate_put_match(&w, best_length, best_distance);
			step = best_length;
		}
		else
			deflate_put_symbol(&w, raw[position]);
		/* Enter the positions passed into the hash chains */
		for (size_t i = 0; i != step; ++i, ++position) {
			if (position + MIN_MATCH <= size) {
				uint32_t hash = ((uint32_t) raw[position] << 10 ^ (uint32_t) raw[position + 1] << 5 ^ raw[position + 2]) & (HASH_SIZE - 1);
				previous[position & (WINDOW - 1)] = head[hash];
				head[hash] = (int32_t) position;
			}
		}
	}
	deflate_put_symbol(&w, 256);
	if (w.bit_count)
		deflate_put_bits(&w, 0, 8 - w.bit_count);
	uint32_t a = 1, b = 0;
	for (size_t i = 0; i != size; ++i) {
		a = (a + raw[i]) % 65521u;
		b = (b + a) % 65521u;
	}
	put_be32(&w.data[w.size], (b << 16) | a);
	w.size += 4;
	free(head); free(previous);
	*out_size = w.size;
	return w.data;
}

static int paeth(int a, int b, int c) {
	int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
	return (pa <= pb && pa <= pc) ? a : ((pb <= pc) ? b : c);
}

int write_png(const char* file_path, uint32_t width, uint32_t height, const uint8_t* rgb) {
	init_crc_table();
	size_t row_bytes = 3 * (size_t) width, stride = 1 + row_bytes;
	if (width == 0 || height == 0 || stride * (double) height > 2.0e9)
		return 1;
	/* Scanlines, each with the filter (none, sub, up, average, Paeth) that
	 * leaves the smallest sum of absolute residuals */
	uint8_t* raw = malloc(stride * height);
	uint8_t* candidate = malloc(row_bytes);
	if (!raw || !candidate) {
		free(raw); free(candidate);
		return 1;
	}
	for (uint32_t y = 0; y != height; ++y) {
		const uint8_t* row = &rgb[row_bytes * y];
		const uint8_t* above = y ? &rgb[row_bytes * (y - 1)] : NULL;
		uint8_t* out = &raw[stride * y];
		uint64_t best_cost = ~0ull;
		for (int filter = 0; filter != 5; ++filter) {
			uint64_t cost = 0;
			for (size_t i = 0; i != row_bytes; ++i) {
				int left = (i >= 3) ? row[i - 3] : 0, up = above ? above[i] : 0, up_left = (above && i >= 3) ? above[i - 3] : 0;
				int predicted = (filter == 0) ? 0 : (filter == 1) ? left : (filter == 2) ? up : (filter == 3) ? (left + up) / 2 : paeth(left, up, up_left);
				uint8_t residual = (uint8_t) (row[i] - predicted);
				candidate[i] = residual;
				cost += (residual < 128) ? residual : 256 - residual;
			}
			if (cost < best_cost) {
				best_cost = cost;
				out[0] = (uint8_t) filter;
				memcpy(out + 1, candidate, row_bytes);
			}
		}
	}
	free(candidate);
	size_t zsize = 0;
	uint8_t* z = zlib_compress(raw, stride * height, &zsize);
	free(raw);
	if (!z || zsize > 0x7fffffffu) {
		free(z);
		return 1;
	}
	FILE* file = fopen(file_path, "wb");
	if (!file) {
		free(z);
		return 1;
	}
	static const uint8_t signature[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
	fwrite(signature, 1, 8, file);
	uint8_t ihdr[13];
	put_be32(ihdr, width);
	put_be32(ihdr + 4, height);
	ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
	write_chunk(file, "IHDR", ihdr, 13);
	write_chunk(file, "IDAT", z, (uint32_t) zsize);
	write_chunk(file, "IEND", NULL, 0);
	free(z);
	return fclose(file) != 0;
}


/* ---- Baseline JPEG (JFIF, 8-bit, YCbCr 4:4:4, the example Huffman and
 * quantisation tables of ITU-T T.81 annex K). Replaces stbi_write_jpg of the
 * vendored stb_image_write.h, which save_screenshot() calls with quality 90
 * (src/main.c:2068). ---- */

static const uint8_t jpg_zigzag[64] = {
	0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
	35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
static const uint8_t jpg_quant_luma[64] = {
	16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
	18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99 };
static const uint8_t jpg_quant_chroma[64] = {
	17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
	99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99 };
static const uint8_t jpg_dc_luma_bits[16] = { 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
static const uint8_t jpg_dc_chroma_bits[16] = { 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
static const uint8_t jpg_dc_values[12] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
static const uint8_t jpg_ac_luma_bits[16] = { 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d };
static const uint8_t jpg_ac_luma_values[162] = {
	0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
	0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
	0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
	0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
	0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
	0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
	0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa };
static const uint8_t jpg_ac_chroma_bits[16] = { 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 };
static const uint8_t jpg_ac_chroma_values[162] = {
	0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
	0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
	0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
	0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
	0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
	0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
	0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa };

typedef struct { uint16_t code[256]; uint8_t length[256]; } jpg_huffman_t;
typedef struct { FILE* file; uint32_t bits; int bit_count; } jpg_writer_t;

static void jpg_build_huffman(jpg_huffman_t* table, const uint8_t* bits, const uint8_t* values) {
	memset(table, 0, sizeof(*table));
	uint32_t code = 0, k = 0;
	for (uint32_t length = 1; length <= 16; ++length) {
		for (uint32_t i = 0; i != bits[length - 1]; ++i, ++k) {
			table->code[values[k]] = (uint16_t) code++;
			table->length[values[k]] = (uint8_t) length;
		}
		code <<= 1;
	}
}

static void jpg_put_bits(jpg_writer_t* w, uint32_t code, int length) {
	w->bits = (w->bits << length) | (code & ((1u << length) - 1u));
	w->bit_count += length;
	while (w->bit_count >= 8) {
		uint8_t byte = (uint8_t) (w->bits >> (w->bit_count - 8));
		fputc(byte, w->file);
		if (byte == 0xff)
			fputc(0, w->file);
		w->bit_count -= 8;
	}
}

static void jpg_put_segment(FILE* file, uint8_t marker, const uint8_t* payload, uint32_t size) {
	uint8_t head[4] = { 0xff, marker, (uint8_t) ((size + 2) >> 8), (uint8_t) (size + 2) };
	fwrite(head, 1, 4, file);
	fwrite(payload, 1, size, file);
}

static void jpg_encode_block(jpg_writer_t* w, const float* samples, const uint8_t* quant, int* previous_dc, const jpg_huffman_t* dc, const jpg_huffman_t* ac) {
	static float cosines[8][8];
	static int cosines_ready = 0;
	if (!cosines_ready) {
		for (int u = 0; u != 8; ++u)
			for (int x = 0; x != 8; ++x)
				cosines[u][x] = (float) (((u == 0) ? sqrt(0.125) : 0.5) * cos((2 * x + 1) * u * M_PI / 16.0));
		cosines_ready = 1;
	}
	float rows[64], coefficients[64];
	for (int y = 0; y != 8; ++y)
		for (int u = 0; u != 8; ++u) {
			float sum = 0.0f;
			for (int x = 0; x != 8; ++x)
				sum += cosines[u][x] * samples[8 * y + x];
			rows[8 * y + u] = sum;
		}
	for (int v = 0; v != 8; ++v)
		for (int u = 0; u != 8; ++u) {
			float sum = 0.0f;
			for (int y = 0; y != 8; ++y)
				sum += cosines[v][y] * rows[8 * y + u];
			coefficients[8 * v + u] = sum;
		}
	int quantized[64];
	for (int i = 0; i != 64; ++i)
		quantized[i] = (int) lrintf(coefficients[jpg_zigzag[i]] / (float) quant[jpg_zigzag[i]]);
	// DC difference, then run-length coded AC coefficients
	int values[2] = { quantized[0] - *previous_dc, 0 };
	*previous_dc = quantized[0];
	int run = 0;
	for (int i = 0; i != 64; ++i) {
		int value = (i == 0) ? values[0] : quantized[i];
		if (i != 0 && value == 0) {
			++run;
			continue;
		}
		int magnitude = abs(value), size = 0;
		while (magnitude >> size)
			++size;
		uint32_t extra = (uint32_t) ((value < 0) ? value - 1 : value);
		if (i == 0)
			jpg_put_bits(w, dc->code[size], dc->length[size]);
		else {
			for (; run >= 16; run -= 16)
				jpg_put_bits(w, ac->code[0xf0], ac->length[0xf0]);
			jpg_put_bits(w, ac->code[(run << 4) | size], ac->length[(run << 4) | size]);
			run = 0;
		}
		if (size)
			jpg_put_bits(w, extra, size);
	}
	if (run)
		jpg_put_bits(w, ac->code[0], ac->length[0]);
}

int write_jpg(const char* file_path, uint32_t width, uint32_t height, const uint8_t* rgb, int quality) {
	if (width == 0 || height == 0 || width > 65535 || height > 65535)
		return 1;
	FILE* file = fopen(file_path, "wb");
	if (!file)
		return 1;
	quality = (quality < 1) ? 1 : ((quality > 100) ? 100 : quality);
	int scale = (quality < 50) ? 5000 / quality : 200 - 2 * quality;
	uint8_t tables[2][64], segment[2][65];
	for (int t = 0; t != 2; ++t) {
		segment[t][0] = (uint8_t) t;
		for (int i = 0; i != 64; ++i) {
			int q = (((t == 0) ? jpg_quant_luma[i] : jpg_quant_chroma[i]) * scale + 50) / 100;
			tables[t][i] = (uint8_t) ((q < 1) ? 1 : ((q > 255) ? 255 : q));
		}
		for (int i = 0; i != 64; ++i)
			segment[t][1 + i] = tables[t][jpg_zigzag[i]];
	}
	static const uint8_t soi[2] = { 0xff, 0xd8 };
	static const uint8_t jfif[14] = { 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0 };
	fwrite(soi, 1, 2, file);
	jpg_put_segment(file, 0xe0, jfif, sizeof(jfif));
	jpg_put_segment(file, 0xdb, segment[0], 65);
	jpg_put_segment(file, 0xdb, segment[1], 65);
	uint8_t frame[15] = { 8, (uint8_t) (height >> 8), (uint8_t) height, (uint8_t) (width >> 8), (uint8_t) width, 3, 1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1 };
	jpg_put_segment(file, 0xc0, frame, sizeof(frame));
	const uint8_t* bit_lists[4] = { jpg_dc_luma_bits, jpg_ac_luma_bits, jpg_dc_chroma_bits, jpg_ac_chroma_bits };
	const uint8_t* value_lists[4] = { jpg_dc_values, jpg_ac_luma_values, jpg_dc_values, jpg_ac_chroma_values };
	static const uint8_t classes[4] = { 0x00, 0x10, 0x01, 0x11 };
	jpg_huffman_t huffman[4];
	for (int t = 0; t != 4; ++t) {
		uint8_t payload[1 + 16 + 162];
		uint32_t count = 0;
		for (int i = 0; i != 16; ++i)
			count += bit_lists[t][i];
		payload[0] = classes[t];
		memcpy(payload + 1, bit_lists[t], 16);
		memcpy(payload + 17, value_lists[t], count);
		jpg_put_segment(file, 0xc4, payload, 17 + count);
		jpg_build_huffman(&huffman[t], bit_lists[t], value_lists[t]);
	}
	static const uint8_t scan[10] = { 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0 };
	jpg_put_segment(file, 0xda, scan, sizeof(scan));
	jpg_writer_t writer = { file, 0, 0 };
	int previous_dc[3] = { 0, 0, 0 };
	for (uint32_t by = 0; by < height; by += 8) {
		for (uint32_t bx = 0; bx < width; bx += 8) {
			float planes[3][64];
			for (uint32_t y = 0; y != 8; ++y) {
				for (uint32_t x = 0; x != 8; ++x) {
					// Blocks that reach over the border repeat the last row and column
					uint32_t sx = (bx + x < width) ? bx + x : width - 1, sy = (by + y < height) ? by + y : height - 1;
					const uint8_t* p = &rgb[3 * ((size_t) sy * width + sx)];
					float r = p[0], g = p[1], b = p[2];
					planes[0][8 * y + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
					planes[1][8 * y + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
					planes[2][8 * y + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
				}
			}
			jpg_encode_block(&writer, planes[0], tables[0], &previous_dc[0], &huffman[0], &huffman[1]);
			jpg_encode_block(&writer, planes[1], tables[1], &previous_dc[1], &huffman[2], &huffman[3]);
			jpg_encode_block(&writer, planes[2], tables[1], &previous_dc[2], &huffman[2], &huffman[3]);
		}
	}
	if (writer.bit_count)
		jpg_put_bits(&writer, 0x7f, 8 - writer.bit_count);
	static const uint8_t eoi[2] = { 0xff, 0xd9 };
	fwrite(eoi, 1, 2, file);
	int failed = ferror(file);
	return (fclose(file) != 0) || failed;
}
