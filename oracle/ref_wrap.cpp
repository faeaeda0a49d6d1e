// ref_wrap.cpp -- extern "C" shims over the REFERENCE's own vector_math.cpp / camera.cpp, which
// oracle/Makefile compiles unmodified from /root/reference/src into oracle/_ref/libbb_ref.so.
// Authoring-container only (the reference does not exist on the GPU box); used to pin the
// oracle's row-A0 restatement and to generate tests/golden/math_golden.json.
// TEST INFRASTRUCTURE ONLY.
#include "vector_math.h"
#include "camera.h"
#include <cstring>

using namespace bb;

static void put(const Mat4 &m, float *out) { std::memcpy(out, m.M, sizeof(float) * 16); }
static Mat4 get(const float *in) { Mat4 m; std::memcpy(m.M, in, sizeof(float) * 16); return m; }

extern "C" {
void ref_mat4_mul(const float *a, const float *b, float *out) { put(get(a) * get(b), out); }
void ref_mat4_inverse(const float *a, float *out) { put(get(a).inverse(), out); }
void ref_mat4_transpose(const float *a, float *out) { put(get(a).transpose(), out); }
void ref_mat4_translate(float x, float y, float z, float *out) { put(Mat4::translate({x, y, z}), out); }
void ref_mat4_scale(float x, float y, float z, float *out) { put(Mat4::scale({x, y, z}), out); }
void ref_mat4_rotate_x(float d, float *out) { put(Mat4::rotateX(d), out); }
void ref_mat4_rotate_y(float d, float *out) { put(Mat4::rotateY(d), out); }
void ref_mat4_rotate_z(float d, float *out) { put(Mat4::rotateZ(d), out); }
void ref_mat4_look_at(const float *eye, const float *target, const float *up, float *out) {
  put(Mat4::lookAt({eye[0], eye[1], eye[2]}, {target[0], target[1], target[2]}, {up[0], up[1], up[2]}), out);
}
void ref_mat4_perspective(float fov, float aspect, float n, float f, float *out) {
  put(Mat4::perspective(fov, aspect, n, f), out);
}
void ref_camera_look(const float *pos, float yaw, float pitch, float *out3) {
  FreeLookCamera c = {{pos[0], pos[1], pos[2]}, yaw, pitch};
  Float3 l = c.getLook();
  out3[0] = l.X; out3[1] = l.Y; out3[2] = l.Z;
}
void ref_camera_view(const float *pos, float yaw, float pitch, float *out) {
  FreeLookCamera c = {{pos[0], pos[1], pos[2]}, yaw, pitch};
  put(c.getViewMatrix(), out);
}
unsigned ref_sizeof_mat4() { return (unsigned)sizeof(Mat4); }
}

// sphericalToCartesian (src/vector_math.cpp:284-292): pins the light-marker sphere of the overlay oracle
extern "C" void ref_spherical_to_cartesian(float r, float theta, float phi, float *out3) {
  Float3 c = sphericalToCartesian({r, theta, phi});
  out3[0] = c.X; out3[1] = c.Y; out3[2] = c.Z;
}
