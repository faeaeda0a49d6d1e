/*
 * bb_oracle.h -- CPU restatement of bibim-renderer's forward PBR path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bibim_renderer_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use
 * it, and only as the checker.
 *
 * Byte layouts are the reference's own (all offsets verified against the reference's
 * headers compiled in the authoring container):
 *   bbo_vertex          = bb::Vertex            src/render.h:112-117      44 B
 *   bbo_instance        = bb::InstanceBlock     src/render.h:96-99       128 B
 *   bbo_light           = bb::Light             src/render.h:310-318      64 B
 *   bbo_frame_uniforms  = bb::FrameUniformBlock src/render.h:321-327    6432 B
 *   bbo_view_uniforms   = bb::ViewUniformBlock  src/render.h:329-334     144 B
 *   bbo_gizmo_vertex    = bb::GizmoVertex       src/render.h:122-126      36 B
 */
#ifndef BB_ORACLE_H
#define BB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float M[4][4]; } bbo_mat4; /* column-major M[col][row], src/vector_math.h:64 */

typedef struct {
  float pos[3];
  float uv[2];
  float normal[3];
  float tangent[3];
} bbo_vertex;

typedef struct {
  float pos[3];
  float color[3];
  float normal[3];
} bbo_gizmo_vertex;

typedef struct {
  bbo_mat4 model;
  bbo_mat4 inv_model;
} bbo_instance;

typedef struct {
  float pos[3];
  int32_t type; /* 0 point, 1 spot, 2 directional */
  float dir[3];
  float intensity;
  float color[3];
  float inner_cutoff;
  float outer_cutoff;
  float _pad[3];
} bbo_light;

#define BBO_MAX_LIGHTS 100
typedef struct {
  int32_t num_lights;
  int32_t _pad0[3];
  bbo_light lights[BBO_MAX_LIGHTS];
  int32_t visualized_gbuffer_attachment_index;
  int32_t enable_tone_mapping;
  float exposure;
  int32_t _pad1;
} bbo_frame_uniforms;

typedef struct {
  bbo_mat4 view;
  bbo_mat4 proj;
  float view_pos[3];
  int32_t enable_normal_map;
} bbo_view_uniforms;

/* One RGBA8 image (stb_image STBI_rgb_alpha layout, src/resource.cpp:159-160).  rgba == NULL
 * selects the `default` material's map for that slot (src/render.cpp:1328-1336). */
typedef struct {
  const uint8_t *rgba;
  int32_t w, h;
} bbo_image;

/* PBRMapType order, src/render.h:235-243 */
enum { BBO_MAP_ALBEDO = 0, BBO_MAP_METALLIC, BBO_MAP_ROUGHNESS, BBO_MAP_AO, BBO_MAP_NORMAL, BBO_MAP_HEIGHT, BBO_MAP_COUNT };

typedef struct {
  bbo_image maps[BBO_MAP_COUNT];
} bbo_material;

/* One draw call, as ShaderBallScene::drawScene records it (src/scene.cpp:193-211). */
typedef struct {
  const bbo_vertex *vertices;
  uint32_t n_vertices;
  const uint32_t *indices; /* NULL => non-indexed vkCmdDraw */
  uint32_t n_indices;
  const bbo_instance *instances;
  uint32_t n_instances;
  const bbo_material *material;
} bbo_draw;

#define BBO_NO_PRIM 0xFFFFFFFFu

/* flags for bbo_render */
#define BBO_FLAG_FORWARD_SHADE 1 /* shade every fragment that passes the depth test, in API order
                                    (literal forward pipeline) instead of shading the winner once */

#define BBO_FLAG_DEFERRED 2      /* the reference's deferred path (its default, src/scene.h:77): gbuffer.vert/.frag into
                                    four RGBA16F attachments, then brdf.frag on every pixel */

#define BBO_FLAG_LITERAL 4       /* evaluate the light loop statement by statement as the GLSL is written
                                    (forward_brdf.frag:29-70) instead of in the shipped evaluation order, which
                                    re-associates the well-conditioned products behind the GGX denominator (see
                                    light_surface_contract in bb_oracle.c).  The GPU is bit-identical to the
                                    default and within BASELINE's 1e-4 of the literal form. */

/* Revision of the arithmetic contract (the default, shipped evaluation order).  The frozen frames under tests/golden/
 * record the revision that minted them (tests/golden/CONTRACT.json): a change of the contract form is a new number here,
 * a line in the history of that file and re-minted fixtures, never a silent re-freeze.  The LITERAL form has no
 * revision: it is the GLSL statement by statement.
 *   1  round 1: three divisions in the specular term, integer-seed rcp / rsqrt sequences
 *   2  round 2: one correctly rounded reciprocal behind the GGX denominator, hoisted kD*albedo/PI and color*intensity,
 *      saturate for N.V, N.L, H.V (light_surface_contract) */
#define BBO_CONTRACT_REVISION 2
uint32_t bbo_contract_revision(void);

#define BBO_FLAG_OUTPUT_UV 8     /* measurement aid (tools/texel_lines.py): out_rgba = (vUV.x, vUV.y, 0, 1) of the winning
                                    fragment instead of its colour: which texels the frame's fetches touch */

typedef struct {
  uint64_t n_prims;         /* triangles submitted */
  uint64_t n_raster_tris;   /* sub-triangles that survived clip + cull */
  uint64_t n_clipped_prims; /* prims that went through the polygon clipper */
  uint64_t n_fragments;     /* coverage hits inside [y0,y1) */
  uint64_t n_shaded;        /* pixels in [y0,y1) whose winning fragment is geometry */
} bbo_stats;

/*
 * Render rows [y0,y1) of a width x height frame.  Outputs are full-frame sized; only rows in
 * range are written.  out_rgba: width*height*4 floats (cleared to 0 per src/main.cpp:84).
 * out_prim (optional): width*height uint32, global primitive index in API order or BBO_NO_PRIM.
 * out_depth (optional): width*height floats (cleared to 0).
 */
int bbo_render(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const bbo_draw *draws,
               uint32_t n_draws, int32_t width, int32_t height, int32_t y0, int32_t y1, uint32_t flags,
               float *out_rgba, uint32_t *out_prim, float *out_depth, bbo_stats *stats);

/* The same forward frame (flags: 0 or BBO_FLAG_LITERAL), whole, on `n_threads` threads: primitives set up once in parallel,
 * then bands of `band_rows` rows taken from a queue.  Bit for bit the frame bbo_render makes; exists so that the CPU baseline
 * bench.py reports can use every core of the host. */
int bbo_render_parallel(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const bbo_draw *draws,
                        uint32_t n_draws, int32_t width, int32_t height, uint32_t flags, int32_t n_threads,
                        int32_t band_rows, float *out_rgba, bbo_stats *stats);

/* Deferred path: src/main.cpp:89-104 (G-buffer subpass, then brdf.frag over a full-screen triangle),
 * src/shaders/gbuffer.vert:17-36, gbuffer.frag:17-33, brdf.frag:11-73, attachments R16G16B16A16_SFLOAT
 * (src/main.cpp:443) cleared to 0.  out_rgba: the HDR colour of every pixel (alpha 1 everywhere: brdf.frag also runs
 * on cleared texels).  out_gbuffer (optional): width*height*16 floats, per pixel position.xyz 1 | normal.xyz 0 |
 * albedo.rgb 0 | metallic roughness ao height, each value exactly representable in binary16 (RNE, as bbo_present). */
int bbo_render_deferred(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const bbo_draw *draws,
                        uint32_t n_draws, int32_t width, int32_t height, int32_t y0, int32_t y1, float *out_rgba,
                        float *out_gbuffer, uint32_t *out_prim, float *out_depth, bbo_stats *stats);

/* as bbo_render_deferred, with flags (BBO_FLAG_LITERAL) */
int bbo_render_deferred_flags(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const bbo_draw *draws,
                              uint32_t n_draws, int32_t width, int32_t height, int32_t y0, int32_t y1, uint32_t flags,
                              float *out_rgba, float *out_gbuffer, uint32_t *out_prim, float *out_depth, bbo_stats *stats);

/* BASELINE config #1: gizmo.vert/.frag (src/shaders/gizmo.vert:12-28, gizmo.frag:10-17) rasterised
 * with the same fixed-function rules into a width x height target (viewport = whole target). */
int bbo_render_gizmo(const bbo_view_uniforms *view, const bbo_gizmo_vertex *vertices, uint32_t n_vertices,
                     const uint32_t *indices, uint32_t n_indices, int32_t width, int32_t height,
                     float *out_rgba, uint32_t *out_prim, float *out_depth, bbo_stats *stats);

/* Overlay subpass (src/main.cpp:128-171): the light markers (light.vert/.frag on generateUVSphereMesh(0.1, 16, 16),
 * one instance per light) and the corner gizmo (gizmo.vert/.frag in a gizmo_extent^2 viewport at the top-right, depth of
 * that rectangle cleared first), drawn over the presented RGBA8 image `rgba8` (in/out) and depth-tested against
 * `scene_depth` (width*height floats as bbo_render returns them; not modified).  Colours are sRGB-encoded like
 * bbo_present does.  gizmo_vertices may be NULL (markers only). */
int bbo_overlay(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, int32_t width, int32_t height,
                const float *scene_depth, uint8_t *rgba8, const bbo_gizmo_vertex *gizmo_vertices, uint32_t n_gizmo_vertices,
                const uint32_t *gizmo_indices, uint32_t n_gizmo_indices, int32_t gizmo_extent, bbo_stats *stats);
/* generateUVSphereMesh (src/render.cpp:1774-1833): positions (3 floats each) and triangle indices; either output may be
 * NULL to query the counts */
void bbo_uv_sphere(float radius, int32_t hdiv, int32_t vdiv, float *out_pos3, uint32_t *out_indices, uint32_t *out_n_vertices,
                   uint32_t *out_n_indices);

/* ---- stage-level entry points (known-answer tests, stage parity) ---- */

/* forward_brdf.vert: out_clip[4], out_vary[14] = uv(2) posWorld(3) N(3) T(3) B(3) */
void bbo_vertex_stage(const bbo_view_uniforms *view, const bbo_instance *inst, const bbo_vertex *v,
                      float *out_clip, float *out_vary);
/* P*V as the vertex stage uses it */
void bbo_proj_view(const bbo_view_uniforms *view, bbo_mat4 *out);
/* bilinear REPEAT sample of one map; out[4] in [0,1] */
void bbo_sample(const bbo_image *img, int map_type, float u, float v, float *out);
/* forward_brdf.frag on explicit varyings */
void bbo_shade_fragment(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view,
                        const bbo_material *mat, const float *vary, float *out_rgba);
/* the same in the shipped evaluation order (what k_shade computes, bit for bit) */
void bbo_shade_fragment_contract(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view,
                                 const bbo_material *mat, const float *vary, float *out_rgba);
/* the light loop + ambient term on an explicit surface point, surf[12] = P(3) normal(3) albedo(3) metallic roughness
 * ao; literal != 0: statement by statement, 0: shipped evaluation order */
void bbo_light_surface(int literal, const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const float *surf,
                       float *out_rgba);
/* brdf.glsl scalar pieces */
float bbo_distribution_ggx(const float *N, const float *H, float roughness);
float bbo_geometry_smith(const float *N, const float *V, const float *L, float roughness);
void bbo_fresnel_schlick(const float *H, const float *V, const float *F0, float *out);
/* hdr_tone_mapping.frag:9-18 applied in place on n RGBA pixels (alpha := 1) */
void bbo_tone_map(float *rgba, uint64_t n_pixels, int32_t enable, float exposure);
/* presentation: (optional) binary16 rounding of the HDR value, tone map, sRGB encode, UNORM8; alpha = 255.
 * hdr_tone_mapping.frag:9-18, src/render.h:94, src/render.cpp:242-254.  See bb_oracle.c for the contract. */
void bbo_present(const float *rgba, uint64_t n_pixels, int32_t enable, float exposure, int32_t hdr16, uint8_t *out_rgba8);
float bbo_half_round(float x);
void bbo_half_round_n(const float *in, float *out, size_t n); /* the same, element by element */
float bbo_exp(float x);
void bbo_srgb_thresholds(float *out255);

/* ---- vector_math.cpp / camera.cpp restatement (row A0) ---- */
void bbo_mat4_identity(bbo_mat4 *out);
void bbo_mat4_mul(const bbo_mat4 *a, const bbo_mat4 *b, bbo_mat4 *out);
void bbo_mat4_inverse(const bbo_mat4 *a, bbo_mat4 *out);
void bbo_mat4_transpose(const bbo_mat4 *a, bbo_mat4 *out);
void bbo_mat4_translate(float x, float y, float z, bbo_mat4 *out);
void bbo_mat4_scale(float x, float y, float z, bbo_mat4 *out);
void bbo_mat4_rotate_x(float degrees, bbo_mat4 *out);
void bbo_mat4_rotate_y(float degrees, bbo_mat4 *out);
void bbo_mat4_rotate_z(float degrees, bbo_mat4 *out);
void bbo_mat4_look_at(const float *eye, const float *target, const float *up, bbo_mat4 *out);
void bbo_mat4_perspective(float fov_degrees, float aspect, float near_z, float far_z, bbo_mat4 *out);
void bbo_camera_look(float yaw, float pitch, float *out3);
void bbo_camera_view(const float *pos, float yaw, float pitch, bbo_mat4 *out);

uint32_t bbo_sizeof(int what); /* 0 vertex 1 instance 2 light 3 frame 4 view 5 gizmo vertex */

#ifdef __cplusplus
}
#endif
#endif
