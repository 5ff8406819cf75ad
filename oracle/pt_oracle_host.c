/* CPU oracle, host part: restatement of the reference's camera and constant
 * marshalling (src/camera.c:102-167, src/math_utilities.c:16-64,197-203,
 * src/main.c:629-676). TEST INFRASTRUCTURE ONLY - see pt_oracle.h.
 * Uses libm's sinf/cosf/tanf like the reference host does.
 */
#include "pt_oracle.h"
#include <math.h>
#include <string.h>

/* src/math_utilities.c:16-26 */
static void mat_mat_mul(float* out, const float* lhs, const float* rhs, uint32_t m, uint32_t k, uint32_t n) {
	for (uint32_t i = 0; i != m; ++i)
		for (uint32_t j = 0; j != n; ++j) {
			float entry = 0.0f;
			for (uint32_t l = 0; l != k; ++l)
				entry += lhs[i * k + l] * rhs[l * n + j];
			out[i * n + j] = entry;
		}
}

/* src/math_utilities.c:43-64 */
static void rotation_matrix_from_angles(float out_rotation[9], const float angles[3]) {
	float sins[3] = { sinf(angles[0]), sinf(angles[1]), sinf(angles[2]) };
	float coss[3] = { cosf(angles[0]), cosf(angles[1]), cosf(angles[2]) };
	float rot_x[9] = { 1.0f, 0.0f, 0.0f, 0.0f, coss[0], sins[0], 0.0f, -sins[0], coss[0] };
	float rot_y[9] = { coss[1], 0.0f, sins[1], 0.0f, 1.0f, 0.0f, -sins[1], 0.0f, coss[1] };
	float rot_z[9] = { coss[2], sins[2], 0.0f, -sins[2], coss[2], 0.0f, 0.0f, 0.0f, 1.0f };
	float rot_xy[9];
	mat_mat_mul(rot_xy, rot_y, rot_x, 3, 3, 3);
	mat_mat_mul(out_rotation, rot_z, rot_xy, 3, 3, 3);
}

/* src/camera.c:102-117 */
static void world_to_view_space(float world_to_view[16], const oracle_camera_t* camera) {
	float rotation[9];
	rotation_matrix_from_angles(rotation, camera->angles);
	float translation[3];
	mat_mat_mul(translation, camera->position, rotation, 1, 3, 3);
	for (int r = 0; r != 3; ++r) {
		for (int c = 0; c != 3; ++c)
			world_to_view[4 * r + c] = rotation[3 * c + r];
		world_to_view[4 * r + 3] = -translation[r];
	}
	world_to_view[12] = world_to_view[13] = world_to_view[14] = 0.0f;
	world_to_view[15] = 1.0f;
}

/* src/camera.c:120-157 */
static void view_to_projection_space(float m[16], const oracle_camera_t* camera, float aspect_ratio) {
	float near = camera->near_plane, far = camera->far_plane;
	memset(m, 0, sizeof(float) * 16);
	if (camera->type == 0 || camera->type == 2 || camera->type == 3) {
		float top = tanf(0.5f * camera->fov);
		float right = aspect_ratio * top;
		m[0] = -1.0f / right;
		m[5] = 1.0f / top;
		m[10] = (far + near) / (near - far);
		m[11] = 2.0f * far * near / (near - far);
		m[14] = -1.0f;
	}
	else if (camera->type == 1) {
		float height = camera->height;
		float width = aspect_ratio * height;
		m[0] = 2.0f / width;
		m[5] = 2.0f / height;
		m[10] = -2.0f / (far - near);
		m[11] = -(far + near) / (far - near);
		m[15] = 1.0f;
	}
}

/* src/camera.c:160-167 */
void oracle_world_to_projection_space(float out[16], const oracle_camera_t* camera, float aspect_ratio) {
	float world_to_view[16], view_to_projection[16];
	world_to_view_space(world_to_view, camera);
	view_to_projection_space(view_to_projection, camera, aspect_ratio);
	mat_mat_mul(out, view_to_projection, world_to_view, 4, 4, 4);
}

/* src/math_utilities.c:67-203. The reference spells out the Leibniz
 * expansion term by term; the same terms in the same order are generated
 * here. Determinant: 24 * m[0][p0] * m[1][p1] * m[2][p2] * m[3][p3] over the
 * permutations p in lexicographic order. Adjugate: for each omitted row r
 * (3, 2, 1, 0) and each lexicographic assignment of distinct columns to the
 * remaining rows, +-6 * (product in row order) goes to out[4 * c + r] with c
 * the unused column. Scale by 4 / det. */
void oracle_invert_mat4(float out[16], const float mat[16]) {
	float det = 0.0f;
	int first = 1;
	for (int p0 = 0; p0 != 4; ++p0) for (int p1 = 0; p1 != 4; ++p1) for (int p2 = 0; p2 != 4; ++p2) for (int p3 = 0; p3 != 4; ++p3) {
		if (p0 == p1 || p0 == p2 || p0 == p3 || p1 == p2 || p1 == p3 || p2 == p3) continue;
		int p[4] = { p0, p1, p2, p3 };
		int inversions = 0;
		for (int i = 0; i != 4; ++i) for (int j = i + 1; j != 4; ++j) inversions += p[i] > p[j];
		float term = 24.0f * mat[p0] * mat[4 + p1] * mat[8 + p2] * mat[12 + p3];
		if (first) { det = term; first = 0; }
		else if (inversions & 1) det -= term;
		else det += term;
	}
	int touched[16] = { 0 };
	for (int r = 3; r >= 0; --r) {
		int rows[3], n = 0;
		for (int i = 0; i != 4; ++i) if (i != r) rows[n++] = i;
		for (int c0 = 0; c0 != 4; ++c0) for (int c1 = 0; c1 != 4; ++c1) for (int c2 = 0; c2 != 4; ++c2) {
			if (c0 == c1 || c0 == c2 || c1 == c2) continue;
			int c = 6 - c0 - c1 - c2;
			int p[4];
			p[rows[0]] = c0; p[rows[1]] = c1; p[rows[2]] = c2; p[r] = c;
			int inversions = 0;
			for (int i = 0; i != 4; ++i) for (int j = i + 1; j != 4; ++j) inversions += p[i] > p[j];
			int index = 4 * c + r;
			float sign = (inversions & 1) ? -6.0f : 6.0f;
			if (!touched[index]) {
				out[index] = sign * mat[4 * rows[0] + c0] * mat[4 * rows[1] + c1] * mat[4 * rows[2] + c2];
				touched[index] = 1;
			}
			else if (inversions & 1)
				out[index] -= 6.0f * mat[4 * rows[0] + c0] * mat[4 * rows[1] + c1] * mat[4 * rows[2] + c2];
			else
				out[index] += 6.0f * mat[4 * rows[0] + c0] * mat[4 * rows[1] + c1] * mat[4 * rows[2] + c2];
		}
	}
	float scale = 4.0f / det;
	for (int i = 0; i != 16; ++i)
		out[i] *= scale;
}

/* src/main.c:629-676 */
void oracle_write_constants(oracle_constants_t* out, const oracle_scene_spec_t* spec, uint32_t width, uint32_t height,
	const float dequantization_factor[3], const float dequantization_summand[3],
	const float* lights, uint32_t light_count, uint32_t accum_frame_count)
{
	memset(out, 0, sizeof(*out));
	out->viewport_size[0] = (float) width;
	out->viewport_size[1] = (float) height;
	out->inv_viewport_size[0] = 1.0f / (float) width;
	out->inv_viewport_size[1] = 1.0f / (float) height;
	out->camera_type = spec->camera.type;
	out->exposure = spec->exposure;
	out->frame_index = spec->frame_index;
	out->accum_frame_count = accum_frame_count + 1;
	memcpy(out->camera_pos, spec->camera.position, sizeof(out->camera_pos));
	float world_to_view[16];
	world_to_view_space(world_to_view, &spec->camera);
	out->hemispherical_camera_normal[0] = world_to_view[8];
	out->hemispherical_camera_normal[1] = world_to_view[9];
	out->hemispherical_camera_normal[2] = world_to_view[10];
	memcpy(out->dequantization_factor, dequantization_factor, sizeof(float) * 3);
	memcpy(out->dequantization_summand, dequantization_summand, sizeof(float) * 3);
	for (int i = 0; i != 3; ++i) {
		out->sky_radiance[i] = spec->sky_color[i] * spec->sky_strength;
		out->emission_material_radiance[i] = spec->emission_material_color[i] * spec->emission_material_strength;
	}
	memcpy(out->params, spec->params, sizeof(out->params));
	if (light_count > 32) light_count = 32;
	memcpy(out->spherical_lights, lights, sizeof(float) * 4 * light_count);
	float aspect = ((float) width) / ((float) height);
	oracle_world_to_projection_space(out->world_to_projection_space, &spec->camera, aspect);
	oracle_invert_mat4(out->projection_to_world_space, out->world_to_projection_space);
}
