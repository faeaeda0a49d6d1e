// bb_scene.cpp -- C++ host shim: the reference's math / camera / scene / "draw a frame" surface for the
// forward path, issuing bbr_* calls (include/bibim_hip.h) where the reference records Vulkan commands.
//
// The float results of every function here are checked bit-for-bit against the reference's own
// vector_math.cpp / camera.cpp (tests/golden/math_golden.json, minted from oracle/_ref) -- which fixes the
// evaluation order: plain left-to-right mul/add, no fused operations (this file is built -ffp-contract=off).
#include "../../include/bibim_scene.h"
#include "../../include/bibim_assets.h"

#include <cmath>
#include <cstring>
#include <memory>

namespace bb {

// ------------------------------------------------------------------------------------------------
// Float3 / Mat4 (src/vector_math.cpp:18-282)
// ------------------------------------------------------------------------------------------------

float Float3::lengthSq() const { return X * X + Y * Y + Z * Z; }
float Float3::length() const { return sqrtf(lengthSq()); }
Float3 Float3::normalize() const { return *this / length(); }
Float3 Float3::operator+(const Float3 &o) const { return {X + o.X, Y + o.Y, Z + o.Z}; }
Float3 Float3::operator-(const Float3 &o) const { return {X - o.X, Y - o.Y, Z - o.Z}; }
Float3 Float3::operator*(float s) const { return {X * s, Y * s, Z * s}; }
Float3 Float3::operator/(float s) const { return {X / s, Y / s, Z / s}; }
float dot(const Float3 &a, const Float3 &b) { return a.X * b.X + a.Y * b.Y + a.Z * b.Z; }
Float3 cross(const Float3 &a, const Float3 &b) {
  return {a.Y * b.Z - a.Z * b.Y, a.Z * b.X - a.X * b.Z, a.X * b.Y - a.Y * b.X};
}

Mat4 Mat4::identity() {
  Mat4 m;
  for (int i = 0; i < 4; ++i) m.M[i][i] = 1.f;
  return m;
}

Mat4 Mat4::transpose() const {
  Mat4 t;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) t.M[r][c] = M[c][r];
  return t;
}

namespace {
// signed 3x3 minor of m with `row` and `col` struck out; expansion order as the reference's
// Mat3::determinant (first row of the column-major minor)
float signedMinor(const Mat4 &m, int row, int col) {
  int cs[3], rs[3];
  for (int i = 0, k = 0; i < 4; ++i)
    if (i != col) cs[k++] = i;
  for (int i = 0, k = 0; i < 4; ++i)
    if (i != row) rs[k++] = i;
  auto e = [&](int c, int r) { return m.M[cs[c]][rs[r]]; };
  float d = e(0, 0) * (e(1, 1) * e(2, 2) - e(2, 1) * e(1, 2)) - e(1, 0) * (e(0, 1) * e(2, 2) - e(2, 1) * e(0, 2)) +
            e(2, 0) * (e(0, 1) * e(1, 2) - e(1, 1) * e(0, 2));
  return ((row + col) & 1) ? d * -1.f : d * 1.f;
}
}  // namespace

Mat4 Mat4::inverse() const {
  float cof[4][4];  // cof[row][col]
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) cof[r][c] = signedMinor(*this, r, c);
  float det = 0.f;
  for (int i = 0; i < 4; ++i) det += M[i][0] * cof[0][i];
  Mat4 inv;  // inverse = adjugate / det; adjugate[col c][row r] = cof[c][r]
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) inv.M[c][r] = cof[c][r] / det;
  return inv;
}

Mat4 Mat4::translate(const Float3 &d) {
  Mat4 m = identity();
  m.M[3][0] = d.X; m.M[3][1] = d.Y; m.M[3][2] = d.Z;
  return m;
}

Mat4 Mat4::scale(const Float3 &s) {
  Mat4 m = identity();
  m.M[0][0] = s.X; m.M[1][1] = s.Y; m.M[2][2] = s.Z;
  return m;
}

Mat4 Mat4::scale(float s) { return scale(Float3{s, s, s}); }

namespace {
// rotation in the plane of axes (a, b): M[a][a] = M[b][b] = cos, M[a][b] = sin, M[b][a] = -sin
Mat4 planeRotation(int a, int b, float degrees) {
  float rad = degToRad(degrees);
  float c = cosf(rad), s = sinf(rad);
  Mat4 m = Mat4::identity();
  m.M[a][a] = c; m.M[a][b] = s; m.M[b][a] = -s; m.M[b][b] = c;
  return m;
}
}  // namespace

Mat4 Mat4::rotateX(float degrees) { return planeRotation(1, 2, degrees); }
Mat4 Mat4::rotateY(float degrees) { return planeRotation(0, 2, degrees); }  // M[0][2] = sin, M[2][0] = -sin
Mat4 Mat4::rotateZ(float degrees) { return planeRotation(0, 1, degrees); }

Mat4 Mat4::lookAt(const Float3 &eye, const Float3 &target, const Float3 &upAxis) {
  Float3 f = (target - eye).normalize();
  Float3 r = cross(upAxis, f).normalize();
  Float3 u = cross(f, r).normalize();
  Mat4 m;
  m.M[0][0] = r.X; m.M[0][1] = u.X; m.M[0][2] = f.X;
  m.M[1][0] = r.Y; m.M[1][1] = u.Y; m.M[1][2] = f.Y;
  m.M[2][0] = r.Z; m.M[2][1] = u.Z; m.M[2][2] = f.Z;
  m.M[3][0] = -dot(eye, r); m.M[3][1] = -dot(eye, u); m.M[3][2] = -dot(eye, f);
  m.M[3][3] = 1.f;
  return m;
}

Mat4 Mat4::perspective(float fovDegrees, float aspectRatio, float nearZ, float farZ) {
  // left-handed, Y flipped, reverse-Z: near -> NDC z 1, far -> 0
  // the reference's unqualified `tan` is the double function when built with g++ (the pinned build):
  // d is rounded once from binary64
  float d = (float)(1.0 / ::tan((double)(degToRad(fovDegrees) * 0.5f)));
  float span = farZ - nearZ;
  Mat4 m;
  m.M[0][0] = d / aspectRatio;
  m.M[1][1] = -d;
  m.M[2][2] = -nearZ / span;
  m.M[2][3] = 1.f;
  m.M[3][2] = nearZ * farZ / span;
  return m;
}

Mat4 operator*(const Mat4 &a, const Mat4 &b) {
  Mat4 r;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i)
      r.M[j][i] = a.M[0][i] * b.M[j][0] + a.M[1][i] * b.M[j][1] + a.M[2][i] * b.M[j][2] + a.M[3][i] * b.M[j][3];
  return r;
}

// ------------------------------------------------------------------------------------------------
// FreeLookCamera (src/camera.cpp:5-20)
// ------------------------------------------------------------------------------------------------

Float3 FreeLookCamera::getLook() const {
  float yaw = degToRad(Yaw), pitch = degToRad(Pitch);
  float cp = cosf(pitch);
  return {-sinf(yaw) * cp, sinf(pitch), cosf(yaw) * cp};
}
Float3 FreeLookCamera::getRight() const { return cross(Float3{0, 1, 0}, getLook()).normalize(); }
Mat4 FreeLookCamera::getViewMatrix() const { return Mat4::lookAt(Pos, Pos + getLook()); }

// ------------------------------------------------------------------------------------------------
// meshes and scenes
// ------------------------------------------------------------------------------------------------

void generatePlaneMesh(std::vector<Vertex> &vertices, std::vector<uint32_t> &indices) {
  // unit quad in the XZ plane, +Y normal, +X tangent; two clockwise triangles (src/render.cpp:1743-1757)
  const uint32_t base = (uint32_t)indices.size();
  const float xs[4] = {-0.5f, -0.5f, 0.5f, 0.5f}, zs[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
  for (int i = 0; i < 4; ++i) {
    Vertex v;
    v.Pos = {xs[i], 0.f, zs[i]};
    v.UV = {xs[i] + 0.5f, zs[i] + 0.5f};
    v.Normal = {0, 1, 0};
    v.Tangent = {1, 0, 0};
    vertices.push_back(v);
  }
  const uint32_t quad[6] = {0, 1, 2, 2, 3, 0};
  for (uint32_t q : quad) indices.push_back(q + base);
}

TriangleScene::TriangleScene(bbr_context *ctx) : Ctx(ctx) {
  Lights.resize(1);  // src/scene.h:141-147
  Lights[0].Dir = {-1, -1, 0};
  Lights[0].Type = LightType::Directional;
  Lights[0].Color = {0.0347f, 0.0131f, 0.2079f};
  Lights[0].Intensity = 10.f;
  Vertex v[3];
  v[0].Pos = {0, 1, 5};   v[0].UV = {0.5f, 1};
  v[1].Pos = {1, -1, 5};  v[1].UV = {1, 0};
  v[2].Pos = {-1, -1, 5}; v[2].UV = {0, 0};
  if (Ctx) bbr_upload_mesh(Ctx, v, 3, nullptr, 0, &Mesh);
  Instance.ModelMat = Mat4::identity();
  Instance.InvModelMat = Mat4::identity();
}

TriangleScene::~TriangleScene() {
  if (Ctx && Mesh >= 0) bbr_free_mesh(Ctx, Mesh);
}

int TriangleScene::drawScene(const Frame &frame) { return bbr_draw(frame.Ctx, Mesh, frame.Material, &Instance, 1); }

ShaderBallScene::ShaderBallScene(bbr_context *ctx, const Vertex *ballVertices, uint32_t numBallVertices, int grid)
    : Ctx(ctx), Grid(grid < 1 ? 1 : grid) {
  // src/scene.cpp:18-36 -- note the cut-offs are stored in radians and compared against a cosine upstream
  Lights.resize(3);
  Lights[0].Dir = {-1, -1, 0};
  Lights[0].Type = LightType::Directional;
  Lights[0].Color = {0.2347f, 0.2131f, 0.2079f};
  Lights[0].Intensity = 10.f;
  Lights[1].Pos = {0, 2, 0};
  Lights[1].Type = LightType::Point;
  Lights[1].Color = {1, 0.8f, 0.8f};
  Lights[1].Intensity = 50;
  Lights[2].Pos = {4, 2, 0};
  Lights[2].Dir = {0, -1, 0};
  Lights[2].Type = LightType::Point;
  Lights[2].Color = {0.8f, 1, 0.8f};
  Lights[2].Intensity = 50;
  Lights[2].InnerCutOff = degToRad(30);
  Lights[2].OuterCutOff = degToRad(25);

  std::vector<Vertex> pv;
  std::vector<uint32_t> pi;
  generatePlaneMesh(pv, pi);
  Plane.NumIndices = (uint32_t)pi.size();
  if (Ctx) bbr_upload_mesh(Ctx, pv.data(), (uint32_t)pv.size(), pi.data(), Plane.NumIndices, &Plane.Mesh);
  Plane.InstanceData.resize(1);
  Plane.InstanceData[0].ModelMat = Mat4::translate({0, -10, 0}) * Mat4::scale({100.f, 100.f, 100.f});
  Plane.InstanceData[0].InvModelMat = Plane.InstanceData[0].ModelMat.inverse();

  ShaderBall.NumVertices = numBallVertices;
  if (Ctx && ballVertices && numBallVertices)
    bbr_upload_mesh(Ctx, ballVertices, numBallVertices, nullptr, 0, &ShaderBall.Mesh);
  ShaderBall.InstanceData.resize((size_t)Grid * Grid);
}

ShaderBallScene::~ShaderBallScene() {
  if (!Ctx) return;
  if (ShaderBall.Mesh >= 0) bbr_free_mesh(Ctx, ShaderBall.Mesh);
  if (Plane.Mesh >= 0) bbr_free_mesh(Ctx, Plane.Mesh);
}

void ShaderBallScene::updateScene(float) {
  if (ShaderBall.Angle > 360) ShaderBall.Angle -= 360;
  const int n = (int)ShaderBall.InstanceData.size();
  for (int i = 0; i < n; ++i) {
    Float3 at = Grid == 1 ? Float3{(float)(i * 2), -1, 2}
                          : Float3{2.f * (float)(i % Grid) - (float)(Grid - 1), -1, 2.f + 2.f * (float)(i / Grid)};
    InstanceBlock &ib = ShaderBall.InstanceData[i];
    ib.ModelMat = Mat4::translate(at) * Mat4::rotateY(ShaderBall.Angle) * Mat4::rotateX(-90) *
                  Mat4::scale({0.01f, 0.01f, 0.01f});
    ib.InvModelMat = ib.ModelMat.inverse();
  }
}

int ShaderBallScene::drawScene(const Frame &frame) {
  // API order matters: ball instances first, plane last (later primitives win depth ties)
  int rc = BBR_OK;
  if (ShaderBall.Mesh >= 0)
    rc = bbr_draw(frame.Ctx, ShaderBall.Mesh, frame.Material, ShaderBall.InstanceData.data(),
                  (uint32_t)ShaderBall.InstanceData.size());
  if (rc != BBR_OK) return rc;
  return bbr_draw(frame.Ctx, Plane.Mesh, frame.Material, Plane.InstanceData.data(), (uint32_t)Plane.InstanceData.size());
}

void fillUniforms(const SceneBase &scene, const FreeLookCamera &cam, const FrameSettings &s, int width, int height,
                  FrameUniformBlock &fb, ViewUniformBlock &vb) {
  fb = FrameUniformBlock();
  size_t n = scene.Lights.size();
  if (n > (size_t)MaxNumLights - 1) n = MaxNumLights - 1;  // upstream asserts size < MAX_NUM_LIGHTS
  fb.NumLights = (int)n;
  std::memcpy(static_cast<void *>(fb.Lights), scene.Lights.data(), n * sizeof(Light));
  fb.EnableToneMapping = s.EnableToneMapping ? 1 : 0;
  fb.Exposure = s.Exposure;
  vb = ViewUniformBlock();
  vb.ViewMat = cam.getViewMatrix();
  vb.ProjMat = Mat4::perspective(s.FovDegrees, (float)width / (float)height, s.NearZ, s.FarZ);
  vb.ViewPos = cam.Pos;
  vb.EnableNormalMap = s.EnableNormalMap ? 1 : 0;
}

int drawFrame(bbr_context *ctx, SceneBase &scene, const FreeLookCamera &cam, const FrameSettings &settings,
              int32_t material, int width, int height, float dt) {
  scene.updateScene(dt);
  // heap, not stack: the blocks are 6.4 KB and callers may run on small fibres
  auto fb = std::make_unique<FrameUniformBlock>();
  ViewUniformBlock vb;
  fillUniforms(scene, cam, settings, width, height, *fb, vb);
  int rc = bbr_set_frame_uniforms(ctx, fb.get());
  if (rc != BBR_OK) return rc;
  rc = bbr_set_view_uniforms(ctx, &vb);
  if (rc != BBR_OK) return rc;
  rc = bbr_set_option(ctx, "render_pass", scene.SceneRenderPassType == RenderPassType::Deferred ? 1 : 0);
  if (rc != BBR_OK) return rc;
  rc = bbr_begin_frame(ctx);
  if (rc != BBR_OK) return rc;
  Frame frame;
  frame.Ctx = ctx;
  frame.Material = material;
  rc = scene.drawScene(frame);
  if (rc != BBR_OK) return rc;
  return bbr_end_frame(ctx);
}

}  // namespace bb

// ================================================================================================
// C surface
// ================================================================================================

struct bbs_scene {
  std::unique_ptr<bb::SceneBase> scene;
  int kind = 0;  // 0 shader balls, 1 triangle
};

namespace {
bb::Mat4 loadMat(const float *p) {
  bb::Mat4 m;
  std::memcpy(m.M, p, sizeof m.M);
  return m;
}
void storeMat(const bb::Mat4 &m, float *p) { std::memcpy(p, m.M, sizeof m.M); }
bb::FreeLookCamera makeCam(const float *pos, float yaw, float pitch) {
  bb::FreeLookCamera c;
  c.Pos = {pos[0], pos[1], pos[2]};
  c.Yaw = yaw;
  c.Pitch = pitch;
  return c;
}
bb::FrameSettings makeSettings(int32_t nm, int32_t tm, float exposure, float fov, float n, float f) {
  bb::FrameSettings s;
  s.EnableNormalMap = nm != 0;
  s.EnableToneMapping = tm != 0;
  s.Exposure = exposure;
  s.FovDegrees = fov;
  s.NearZ = n;
  s.FarZ = f;
  return s;
}
}  // namespace

extern "C" {

void bbs_mat4_mul(const float *a, const float *b, float *out) { storeMat(loadMat(a) * loadMat(b), out); }
void bbs_mat4_inverse(const float *a, float *out) { storeMat(loadMat(a).inverse(), out); }
void bbs_mat4_translate(float x, float y, float z, float *out) { storeMat(bb::Mat4::translate({x, y, z}), out); }
void bbs_mat4_scale(float x, float y, float z, float *out) { storeMat(bb::Mat4::scale(bb::Float3{x, y, z}), out); }
void bbs_mat4_rotate(int axis, float degrees, float *out) {
  storeMat(axis == 0 ? bb::Mat4::rotateX(degrees) : axis == 1 ? bb::Mat4::rotateY(degrees) : bb::Mat4::rotateZ(degrees), out);
}
void bbs_mat4_look_at(const float *eye, const float *target, const float *up, float *out) {
  storeMat(bb::Mat4::lookAt({eye[0], eye[1], eye[2]}, {target[0], target[1], target[2]}, {up[0], up[1], up[2]}), out);
}
void bbs_mat4_perspective(float fov, float aspect, float n, float f, float *out) {
  storeMat(bb::Mat4::perspective(fov, aspect, n, f), out);
}
void bbs_camera_look(float yaw, float pitch, float *out3) {
  float zero[3] = {0, 0, 0};
  bb::Float3 l = makeCam(zero, yaw, pitch).getLook();
  out3[0] = l.X; out3[1] = l.Y; out3[2] = l.Z;
}
void bbs_camera_view(const float *pos, float yaw, float pitch, float *out) {
  storeMat(makeCam(pos, yaw, pitch).getViewMatrix(), out);
}
void bbs_plane_mesh(void *out_vertices4, uint32_t *out_indices6) {
  std::vector<bb::Vertex> v;
  std::vector<uint32_t> i;
  bb::generatePlaneMesh(v, i);
  std::memcpy(out_vertices4, v.data(), 4 * sizeof(bb::Vertex));
  std::memcpy(out_indices6, i.data(), 6 * sizeof(uint32_t));
}

bbs_scene *bbs_shaderball_scene_create(bbr_context *ctx, const void *ball_vertices, uint32_t n_vertices, int32_t grid) {
  bbs_scene *s = new bbs_scene();
  s->scene.reset(new bb::ShaderBallScene(ctx, static_cast<const bb::Vertex *>(ball_vertices), n_vertices, grid));
  s->kind = 0;
  return s;
}
bbs_scene *bbs_shaderball_scene_create_from_file(bbr_context *ctx, const char *fbx_path, int32_t grid) {
  // what the reference's constructor does itself: import ShaderBall.fbx and expand it to a triangle list
  // (src/scene.cpp:57-86), here with the in-repo FBX reader instead of assimp
  void *v = nullptr;
  uint32_t n = 0;
  if (bba_load_fbx_vertices(fbx_path, &v, &n) != BBA_OK) return nullptr;  // bba_last_error() has the reason
  bbs_scene *s = bbs_shaderball_scene_create(ctx, v, n, grid);
  bba_free(v);
  return s;
}
bbs_scene *bbs_triangle_scene_create(bbr_context *ctx) {
  bbs_scene *s = new bbs_scene();
  s->scene.reset(new bb::TriangleScene(ctx));
  s->kind = 1;
  return s;
}
void bbs_scene_destroy(bbs_scene *scene) { delete scene; }

int bbs_scene_set_render_pass(bbs_scene *scene, int32_t render_pass) {
  if (!scene || !scene->scene || (render_pass != 0 && render_pass != 1)) return BBR_ERR_INVALID_ARGUMENT;
  scene->scene->SceneRenderPassType = render_pass ? bb::RenderPassType::Deferred : bb::RenderPassType::Forward;
  return BBR_OK;
}

int bbs_scene_set_lights(bbs_scene *scene, const void *lights, uint32_t n) {
  if (!scene || (!lights && n) || n >= (uint32_t)bb::MaxNumLights) return BBR_ERR_INVALID_ARGUMENT;
  scene->scene->Lights.resize(n);
  if (n) std::memcpy(static_cast<void *>(scene->scene->Lights.data()), lights, n * sizeof(bb::Light));
  return BBR_OK;
}
uint32_t bbs_scene_num_lights(const bbs_scene *scene) { return scene ? (uint32_t)scene->scene->Lights.size() : 0; }
int bbs_scene_get_lights(const bbs_scene *scene, void *out) {
  if (!scene || !out) return BBR_ERR_INVALID_ARGUMENT;
  std::memcpy(out, scene->scene->Lights.data(), scene->scene->Lights.size() * sizeof(bb::Light));
  return BBR_OK;
}

int bbs_scene_instances(bbs_scene *scene, int32_t draw_index, void *out, uint32_t capacity, uint32_t *out_n) {
  if (!scene || !out_n) return BBR_ERR_INVALID_ARGUMENT;
  scene->scene->updateScene(0.f);
  const std::vector<bb::InstanceBlock> *src = nullptr;
  std::vector<bb::InstanceBlock> one;
  if (scene->kind == 0) {
    auto *sb = static_cast<bb::ShaderBallScene *>(scene->scene.get());
    src = draw_index == 0 ? &sb->ShaderBall.InstanceData : draw_index == 1 ? &sb->Plane.InstanceData : nullptr;
  } else if (draw_index == 0) {
    one.push_back(static_cast<bb::TriangleScene *>(scene->scene.get())->Instance);
    src = &one;
  }
  if (!src) return BBR_ERR_INVALID_ARGUMENT;
  *out_n = (uint32_t)src->size();
  if (out) {
    if (capacity < src->size()) return BBR_ERR_INVALID_ARGUMENT;
    std::memcpy(out, src->data(), src->size() * sizeof(bb::InstanceBlock));
  }
  return BBR_OK;
}

int bbs_fill_uniforms(const bbs_scene *scene, const float *cam_pos, float yaw, float pitch, int32_t enable_normal_map,
                      int32_t enable_tone_mapping, float exposure, float fov, float near_z, float far_z, int32_t width,
                      int32_t height, void *out_frame_block, void *out_view_block) {
  if (!scene || !cam_pos || !out_frame_block || !out_view_block || width <= 0 || height <= 0)
    return BBR_ERR_INVALID_ARGUMENT;
  auto fb = std::make_unique<bb::FrameUniformBlock>();
  bb::ViewUniformBlock vb;
  bb::fillUniforms(*scene->scene, makeCam(cam_pos, yaw, pitch),
                   makeSettings(enable_normal_map, enable_tone_mapping, exposure, fov, near_z, far_z), width, height, *fb, vb);
  std::memcpy(out_frame_block, fb.get(), sizeof(bb::FrameUniformBlock));
  std::memcpy(out_view_block, &vb, sizeof vb);
  return BBR_OK;
}

int bbs_draw_frame(bbr_context *ctx, bbs_scene *scene, const float *cam_pos, float yaw, float pitch,
                   int32_t enable_normal_map, int32_t enable_tone_mapping, float exposure, float fov, float near_z,
                   float far_z, int32_t material, int32_t width, int32_t height) {
  if (!ctx || !scene || !cam_pos) return BBR_ERR_INVALID_ARGUMENT;
  return bb::drawFrame(ctx, *scene->scene, makeCam(cam_pos, yaw, pitch),
                       makeSettings(enable_normal_map, enable_tone_mapping, exposure, fov, near_z, far_z), material, width,
                       height, 0.f);
}

}  // extern "C"
