/* Small matrix utilities of the host, with the arithmetic of the reference's
 * src/math_utilities.c so that the constants handed to the GPU agree with it
 * bit for bit. */
#include "pt_host.h"
#include <math.h>

/* src/math_utilities.c:16-26 */
void mat_mat_mul(float* out, const float* lhs, const float* rhs, uint32_t m, uint32_t k, uint32_t n) {
	for (uint32_t row = 0; row != m; ++row) {
		for (uint32_t col = 0; col != n; ++col) {
			float sum = 0.0f;
			for (uint32_t l = 0; l != k; ++l)
				sum += lhs[row * k + l] * rhs[l * n + col];
			out[row * n + col] = sum;
		}
	}
}

/* src/math_utilities.c:29-40 */
bool normalize(float* vec, uint32_t entry_count) {
	float length_sq = 0.0f;
	for (uint32_t i = 0; i != entry_count; ++i)
		length_sq += vec[i] * vec[i];
	if (length_sq == 0.0f)
		return false;
	float scale = 1.0f / sqrtf(length_sq);
	for (uint32_t i = 0; i != entry_count; ++i)
		vec[i] *= scale;
	return true;
}

/* src/math_utilities.c:43-64: R = Rz * (Ry * Rx) */
void rotation_matrix_from_angles(float out_rotation[3 * 3], const float angles[3]) {
	float s[3], c[3];
	for (int i = 0; i != 3; ++i) {
		s[i] = sinf(angles[i]);
		c[i] = cosf(angles[i]);
	}
	const float rot_x[9] = { 1.0f, 0.0f, 0.0f,  0.0f, c[0], s[0],  0.0f, -s[0], c[0] };
	const float rot_y[9] = { c[1], 0.0f, s[1],  0.0f, 1.0f, 0.0f,  -s[1], 0.0f, c[1] };
	const float rot_z[9] = { c[2], s[2], 0.0f,  -s[2], c[2], 0.0f,  0.0f, 0.0f, 1.0f };
	float rot_xy[9];
	mat_mat_mul(rot_xy, rot_y, rot_x, 3, 3, 3);
	mat_mat_mul(out_rotation, rot_z, rot_xy, 3, 3, 3);
}

/* Parity of the permutation row -> col[row] of 4 elements */
static int permutation_is_odd(const int col[4]) {
	int inversions = 0;
	for (int a = 0; a != 4; ++a)
		for (int b = a + 1; b != 4; ++b)
			if (col[a] > col[b])
				++inversions;
	return inversions & 1;
}

/* src/math_utilities.c:67-203. The reference writes out the Leibniz sums of
 * the determinant (24 terms with a factor 24) and of the adjugate (96 terms
 * with a factor 6) and scales by 4 / det. Floating-point sums depend on the
 * order of the terms, so the same order is produced here: columns are
 * assigned to rows in lexicographic order; for the adjugate the omitted row
 * runs from 3 down to 0 and the product skips it. */
void invert_mat4(float out_inv[4 * 4], float mat[4 * 4]) {
	float det = 0.0f;
	bool det_started = false;
	bool started[16] = { false };
	int col[4];
	for (col[0] = 0; col[0] != 4; ++col[0])
	for (col[1] = 0; col[1] != 4; ++col[1])
	for (col[2] = 0; col[2] != 4; ++col[2])
	for (col[3] = 0; col[3] != 4; ++col[3]) {
		if ((1 << col[0] | 1 << col[1] | 1 << col[2] | 1 << col[3]) != 15)
			continue;
		float term = 24.0f * mat[col[0]] * mat[4 + col[1]] * mat[8 + col[2]] * mat[12 + col[3]];
		if (!det_started) {
			det = term;
			det_started = true;
		}
		else if (permutation_is_odd(col))
			det -= term;
		else
			det += term;
	}
	for (int omitted = 3; omitted >= 0; --omitted) {
		int rows[3];
		for (int r = 0, i = 0; r != 4; ++r)
			if (r != omitted)
				rows[i++] = r;
		int c[3];
		for (c[0] = 0; c[0] != 4; ++c[0])
		for (c[1] = 0; c[1] != 4; ++c[1])
		for (c[2] = 0; c[2] != 4; ++c[2]) {
			int used = 1 << c[0] | 1 << c[1] | 1 << c[2];
			if (c[0] == c[1] || c[0] == c[2] || c[1] == c[2])
				continue;
			int free_col = 0;
			while (used & (1 << free_col))
				++free_col;
			col[rows[0]] = c[0];
			col[rows[1]] = c[1];
			col[rows[2]] = c[2];
			col[omitted] = free_col;
			bool odd = permutation_is_odd(col);
			float* dst = &out_inv[4 * free_col + omitted];
			if (!started[4 * free_col + omitted]) {
				*dst = (odd ? -6.0f : 6.0f) * mat[4 * rows[0] + c[0]] * mat[4 * rows[1] + c[1]] * mat[4 * rows[2] + c[2]];
				started[4 * free_col + omitted] = true;
			}
			else {
				float term = 6.0f * mat[4 * rows[0] + c[0]] * mat[4 * rows[1] + c[1]] * mat[4 * rows[2] + c[2]];
				if (odd)
					*dst -= term;
				else
					*dst += term;
			}
		}
	}
	float scale = 4.0f / det;
	for (uint32_t i = 0; i != 16; ++i)
		out_inv[i] *= scale;
}
