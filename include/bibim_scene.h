/*
 * bibim_scene.h -- C++ host shim above the C ABI: the reference's Scene / Camera / "draw a frame"
 * surface for the forward path, rebuilt on bbr_* calls instead of Vulkan.
 *
 * Mirrors (same names, argument meaning and conventions):
 *   bb::Float3, bb::Mat4 (column-major M[col][row], degrees, pi32 = 3.141592f, reverse-Z LH perspective)
 *                                  src/vector_math.h:6-86, src/vector_math.cpp:84-282
 *   bb::FreeLookCamera             src/camera.h:6-14, src/camera.cpp:5-20
 *   bb::Light / Vertex / InstanceBlock / FrameUniformBlock / ViewUniformBlock
 *                                  src/render.h:96-99,112-117,310-334
 *   bb::SceneBase / TriangleScene / ShaderBallScene (updateScene, drawScene)
 *                                  src/scene.h:75-221, src/scene.cpp:12-211
 *   bb::generatePlaneMesh          src/render.cpp:1743-1757
 *   bb::drawFrame                  = updateScene + uniform fill (src/main.cpp:1286-1342) +
 *                                    recordCommand's ForwardLighting branch (src/main.cpp:106-112) + submit
 *
 * The reference has no symbol called drawFrame; BASELINE.json's name is used for the function that does what
 * one iteration of the reference's main loop does for the forward path.
 *
 * A C surface (bbs_*) at the bottom lets non-C++ hosts and the parity tests drive the same code.
 */
#ifndef BIBIM_SCENE_H
#define BIBIM_SCENE_H

#include <stdint.h>

#include "bibim_hip.h"

#ifdef __cplusplus
#include <vector>

namespace bb {

constexpr float pi32 = 3.141592f;
inline float degToRad(float degrees) { return degrees * pi32 / 180.f; }

struct Float2 {
  float X = 0.f, Y = 0.f;
};

struct Float3 {
  float X = 0.f, Y = 0.f, Z = 0.f;
  float lengthSq() const;
  float length() const;
  Float3 normalize() const;
  Float3 operator+(const Float3 &o) const;
  Float3 operator-(const Float3 &o) const;
  Float3 operator*(float s) const;
  Float3 operator/(float s) const;
};
float dot(const Float3 &a, const Float3 &b);
Float3 cross(const Float3 &a, const Float3 &b);

struct Mat4 {
  float M[4][4] = {};
  Mat4 inverse() const;
  Mat4 transpose() const;
  static Mat4 identity();
  static Mat4 translate(const Float3 &delta);
  static Mat4 scale(const Float3 &scale);
  static Mat4 scale(float s);
  static Mat4 rotateX(float degrees);
  static Mat4 rotateY(float degrees);
  static Mat4 rotateZ(float degrees);
  static Mat4 lookAt(const Float3 &eye, const Float3 &target, const Float3 &upAxis = {0, 1, 0});
  static Mat4 perspective(float fovDegrees, float aspectRatio, float nearZ, float farZ);
};
Mat4 operator*(const Mat4 &a, const Mat4 &b);

struct FreeLookCamera {
  Float3 Pos;
  float Yaw = 0.f;
  float Pitch = 0.f;
  Mat4 getViewMatrix() const;
  Float3 getRight() const;
  Float3 getLook() const;
};

enum class LightType : int32_t { Point = 0, Spot = 1, Directional = 2 };

struct alignas(16) Light {
  Float3 Pos;
  LightType Type = LightType::Point;
  Float3 Dir;
  float Intensity = 0.f;
  Float3 Color;
  float InnerCutOff = 0.f;
  float OuterCutOff = 0.f;
};

struct Vertex {
  Float3 Pos;
  Float2 UV;
  Float3 Normal = {0, 0, -1};
  Float3 Tangent = {0, -1, 0};
};

struct InstanceBlock {
  Mat4 ModelMat;
  Mat4 InvModelMat;
};

constexpr int MaxNumLights = 100;
struct FrameUniformBlock {
  int NumLights = 0;
  Light Lights[MaxNumLights];
  int VisualizedGBufferAttachmentIndex = 0;
  int EnableToneMapping = 0;
  float Exposure = 1.f;
};

struct ViewUniformBlock {
  Mat4 ViewMat;
  Mat4 ProjMat;
  Float3 ViewPos;
  int EnableNormalMap = 0;
};

static_assert(sizeof(Vertex) == BBR_SIZEOF_VERTEX, "Vertex layout");
static_assert(sizeof(InstanceBlock) == BBR_SIZEOF_INSTANCE_BLOCK, "InstanceBlock layout");
static_assert(sizeof(Light) == BBR_SIZEOF_LIGHT, "Light layout");
static_assert(sizeof(FrameUniformBlock) == BBR_SIZEOF_FRAME_UNIFORM_BLOCK, "FrameUniformBlock layout");
static_assert(sizeof(ViewUniformBlock) == BBR_SIZEOF_VIEW_UNIFORM_BLOCK, "ViewUniformBlock layout");

void generatePlaneMesh(std::vector<Vertex> &vertices, std::vector<uint32_t> &indices);

// What the scenes draw into: the bbr context plus the bound material (GUI.SelectedMaterial upstream).
struct Frame {
  bbr_context *Ctx = nullptr;
  int32_t Material = -1;
};

enum class RenderPassType : int32_t { Forward = 0, Deferred = 1 };  // src/scene.h:76 (enum), :77 (member)

struct SceneBase {
  // The reference defaults to Deferred (src/scene.h:77); this shim defaults to the forward path, the one BASELINE
  // measures.  drawFrame selects forward_brdf.* or gbuffer.* + brdf.* with it, as recordCommand does
  // (src/main.cpp:89-112).
  RenderPassType SceneRenderPassType = RenderPassType::Forward;
  std::vector<Light> Lights;
  virtual ~SceneBase() = default;
  virtual void updateScene(float dt) = 0;
  virtual int drawScene(const Frame &frame) = 0;  // records bbr_draw calls in API order
};

struct TriangleScene : SceneBase {
  explicit TriangleScene(bbr_context *ctx);
  ~TriangleScene() override;
  void updateScene(float) override {}
  int drawScene(const Frame &frame) override;
  bbr_context *Ctx;
  int32_t Mesh = -1;
  InstanceBlock Instance;
};

struct ShaderBallScene : SceneBase {
  // ballVertices: the non-indexed triangle list ShaderBall.fbx expands to (src/scene.cpp:62-79).
  // grid == 1 reproduces the reference (instance i at x = 2i); grid G > 1 places G*G instances on the
  // benchmark lattice translate(2*(i%G) - (G-1), -1, 2 + 2*(i/G)).
  ShaderBallScene(bbr_context *ctx, const Vertex *ballVertices, uint32_t numBallVertices, int grid = 1);
  ~ShaderBallScene() override;
  void updateScene(float dt) override;
  int drawScene(const Frame &frame) override;

  bbr_context *Ctx;
  struct {
    int32_t Mesh = -1;
    uint32_t NumIndices = 0;
    std::vector<InstanceBlock> InstanceData;
  } Plane;
  struct {
    int32_t Mesh = -1;
    uint32_t NumVertices = 0;
    std::vector<InstanceBlock> InstanceData;
    float Angle = -90;
  } ShaderBall;
  int Grid = 1;
};

struct FrameSettings {
  bool EnableNormalMap = false;   // src/main.cpp:1302 (static, default off)
  bool EnableToneMapping = false; // :1303
  float Exposure = 1.f;           // :1304
  float FovDegrees = 60.f;        // :1331
  float NearZ = 0.1f;
  float FarZ = 1000.f;            // :1332
};

void fillUniforms(const SceneBase &scene, const FreeLookCamera &cam, const FrameSettings &settings, int width,
                  int height, FrameUniformBlock &frameBlock, ViewUniformBlock &viewBlock);

// One iteration of the reference's render loop for the forward path.  Asynchronous; returns a bbr_status.
int drawFrame(bbr_context *ctx, SceneBase &scene, const FreeLookCamera &cam, const FrameSettings &settings,
              int32_t material, int width, int height, float dt = 0.f);

}  // namespace bb

extern "C" {
#endif /* __cplusplus */

/* ---- C surface over the shim (used by the Python harness and by non-C++ hosts) ---- */
/* A scene owns meshes of the context it was created on: destroy scenes before bbr_destroy(ctx).  (If the order is
 * reversed -- garbage-collected hosts -- bbr_free_mesh on the dead context returns BBR_ERR_BAD_HANDLE and touches
 * nothing.) */
typedef struct bbs_scene bbs_scene;

void bbs_mat4_mul(const float *a, const float *b, float *out);
void bbs_mat4_inverse(const float *a, float *out);
void bbs_mat4_translate(float x, float y, float z, float *out);
void bbs_mat4_scale(float x, float y, float z, float *out);
void bbs_mat4_rotate(int axis, float degrees, float *out); /* 0 x, 1 y, 2 z */
void bbs_mat4_look_at(const float *eye, const float *target, const float *up, float *out);
void bbs_mat4_perspective(float fov_degrees, float aspect, float near_z, float far_z, float *out);
void bbs_camera_look(float yaw, float pitch, float *out3);
void bbs_camera_view(const float *pos, float yaw, float pitch, float *out);
void bbs_plane_mesh(void *out_vertices4, uint32_t *out_indices6);

bbs_scene *bbs_shaderball_scene_create(bbr_context *ctx, const void *ball_vertices, uint32_t n_vertices, int32_t grid);
/* the same, importing the ball from a binary FBX file like the reference's constructor (src/scene.cpp:57-86) with
 * bba_load_fbx_vertices (include/bibim_assets.h); NULL on failure, reason in bba_last_error() */
bbs_scene *bbs_shaderball_scene_create_from_file(bbr_context *ctx, const char *fbx_path, int32_t grid);
bbs_scene *bbs_triangle_scene_create(bbr_context *ctx);
void bbs_scene_destroy(bbs_scene *scene);
/* replace the scene's lights with n Light records (64 B each) */
int bbs_scene_set_lights(bbs_scene *scene, const void *lights, uint32_t n);
/* SceneBase::SceneRenderPassType: 0 forward, 1 deferred */
int bbs_scene_set_render_pass(bbs_scene *scene, int32_t render_pass);
uint32_t bbs_scene_num_lights(const bbs_scene *scene);
int bbs_scene_get_lights(const bbs_scene *scene, void *out_lights);
/* run updateScene and copy out the instance data of draw `draw_index` (0 ball, 1 plane) */
int bbs_scene_instances(bbs_scene *scene, int32_t draw_index, void *out_instances, uint32_t capacity, uint32_t *out_n);
int bbs_fill_uniforms(const bbs_scene *scene, const float *cam_pos, float yaw, float pitch, int32_t enable_normal_map,
                      int32_t enable_tone_mapping, float exposure, float fov, float near_z, float far_z, int32_t width,
                      int32_t height, void *out_frame_block, void *out_view_block);
int bbs_draw_frame(bbr_context *ctx, bbs_scene *scene, const float *cam_pos, float yaw, float pitch,
                   int32_t enable_normal_map, int32_t enable_tone_mapping, float exposure, float fov, float near_z,
                   float far_z, int32_t material, int32_t width, int32_t height);

#ifdef __cplusplus
}
#endif
#endif
