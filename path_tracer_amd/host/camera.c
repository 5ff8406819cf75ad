/* Camera transforms of the host (src/camera.c:102-167 of the reference). The
 * interactive controls of the reference (control_camera(), src/camera.c:10-99)
 * have no place in a headless renderer. */
#include "pt_host.h"
#include <math.h>
#include <string.h>

/* src/camera.c:102-117 */
void get_world_to_view_space(float world_to_view[4 * 4], const camera_t* camera) {
	/* View to world rotation; world to view uses its transpose */
	float rotation[3 * 3];
	rotation_matrix_from_angles(rotation, camera->rotation.angles);
	float translation[3];
	mat_mat_mul(translation, camera->position, rotation, 1, 3, 3);
	for (uint32_t row = 0; row != 3; ++row) {
		for (uint32_t col = 0; col != 3; ++col)
			world_to_view[4 * row + col] = rotation[3 * col + row];
		world_to_view[4 * row + 3] = -translation[row];
	}
	world_to_view[12] = 0.0f;
	world_to_view[13] = 0.0f;
	world_to_view[14] = 0.0f;
	world_to_view[15] = 1.0f;
}

/* src/camera.c:120-157 */
void get_view_to_projection_space(float view_to_projection[4 * 4], const camera_t* camera, float aspect_ratio) {
	float near = camera->near, far = camera->far;
	memset(view_to_projection, 0, sizeof(float) * 16);
	switch (camera->type) {
		case camera_type_first_person:
		case camera_type_hemispherical:
		case camera_type_spherical: {
			float top = tanf(0.5f * camera->fov);
			float right = aspect_ratio * top;
			view_to_projection[0] = -1.0f / right;
			view_to_projection[5] = 1.0f / top;
			view_to_projection[10] = (far + near) / (near - far);
			view_to_projection[11] = 2.0f * far * near / (near - far);
			view_to_projection[14] = -1.0f;
			break;
		}
		case camera_type_ortho: {
			float height = camera->height;
			float width = aspect_ratio * height;
			view_to_projection[0] = 2.0f / width;
			view_to_projection[5] = 2.0f / height;
			view_to_projection[10] = -2.0f / (far - near);
			view_to_projection[11] = -(far + near) / (far - near);
			view_to_projection[15] = 1.0f;
			break;
		}
		default:
			break;
	}
}

/* src/camera.c:160-167 */
void get_world_to_projection_space(float world_to_projection[4 * 4], const camera_t* camera, float aspect_ratio) {
	float world_to_view[4 * 4], view_to_projection[4 * 4];
	get_world_to_view_space(world_to_view, camera);
	get_view_to_projection_space(view_to_projection, camera, aspect_ratio);
	mat_mat_mul(world_to_projection, view_to_projection, world_to_view, 4, 4, 4);
}
