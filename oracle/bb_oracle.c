/*
 * bb_oracle.c -- scalar CPU restatement of bibim-renderer's forward PBR path
 * (vertex transform -> clip/cull/raster/depth -> Cook-Torrance/GGX fragment shade).
 *
 * TEST INFRASTRUCTURE ONLY (see bb_oracle.h).  The shipped HIP path never calls this.
 *
 * Parity status
 *   - Row A0 (vector_math.cpp / camera.cpp): PINNED.  The bbo_mat4_* / bbo_camera_* functions are
 *     checked bit-for-bit against the reference's own sources compiled unmodified into
 *     oracle/_ref (see oracle/Makefile, oracle/ref_wrap.cpp) and against tests/golden/math_golden.json
 *     generated from that build.
 *   - Rows A2-A6 (GLSL stages + Vulkan fixed function): the reference holds no golden images, no tests
 *     and no CPU implementation; glslc / a Vulkan driver do not exist in the authoring container, so
 *     the shader arithmetic is a restatement of the GLSL text pinned by closed-form known-answer
 *     vectors only (tests/test_oracle_kat.py).  Driver latitude (sub-pixel bits, FMA contraction,
 *     filter weight precision, anisotropic taps) is "parity unpinned": the choices below ARE the contract.
 *
 * Arithmetic contract (shared, by construction, with the HIP kernels; compiled -ffp-contract=off):
 *   IEEE binary32, round-to-nearest-even, denormals kept, correctly rounded / and sqrtf.
 *   dot3(a,b)      = fmaf(a.z,b.z, fmaf(a.y,b.y, a.x*b.x))
 *   cross(a,b).x   = fmaf(a.y,b.z, -(a.z*b.y))   (cyclic)
 *   rsqrt(x)       = GLSL inversesqrt as a FIXED sequence: integer seed 0x5F375A86 - (bits(x) >> 1), three
 *                    Newton steps y = y * fma(-(0.5x*y), y, 1.5).  Max error 1.1 ulp (GLSL allows 2), and --
 *                    unlike a hardware v_rsq_f32 or a correctly rounded 1/sqrt -- it is the same bits on CPU and
 *                    GPU at a third of the GPU instruction cost of IEEE sqrt + divide.  Arguments outside the
 *                    normal positive range fall back to 1.0f / sqrtf(x) (zero -> inf, negative -> NaN).
 *   rcp(x)         = the correctly rounded 1/x (an IEEE division here).  The GPU reaches the same bits with
 *                    v_rcp_f32 plus ONE Newton step for every normal input with a normal reciprocal (checked
 *                    exhaustively by bbr_selftest_rcp) and takes the IEEE division outside that range.  GLSL
 *                    divisions a / b of the fragment stage are evaluated as a * rcp(b) (GLSL allows 2.5 ulp).
 *   normalize(v)   = v * rsqrt(dot3(v,v));  length-based attenuation 1/(d*d) = rsqrt(d2)^2
 *   mat*vec        = fmaf(c3,w, fmaf(c2,z, fmaf(c1,y, c0*x)))   per row
 *   mix(a,b,t)     = fmaf(b,t, a*(1-t));  pow(x,5) = ((x*x)*(x*x))*x;  x/PI = x*(float)(1/pi)
 *   Setup quantities that are evaluated once per triangle use binary64 and are rounded once.
 */
#include "bb_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* small vector helpers                                                                       */
/* ------------------------------------------------------------------------------------------ */

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_ld(const float *p) { v3 r = {p[0], p[1], p[2]}; return r; }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 add3(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 scale3(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline v3 cross3(v3 a, v3 b) {
  return v3_make(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float bb_rsqrt(float x) {
  if (!(x >= 1.17549435e-38f && x <= 3.40282347e+38f)) return 1.0f / sqrtf(x);
  union { float f; uint32_t u; } c;
  c.f = x;
  c.u = 0x5F375A86u - (c.u >> 1);
  float y = c.f;
  const float h = 0.5f * x;
  y = y * fmaf(-(h * y), y, 1.5f);
  y = y * fmaf(-(h * y), y, 1.5f);
  y = y * fmaf(-(h * y), y, 1.5f);
  return y;
}
/* Reciprocal: the correctly rounded 1/x (IEEE division).  The GPU reaches the same bits for every input with
 * v_rcp_f32 plus ONE Newton step -- verified exhaustively over all 2 113 929 216 normal inputs with a normal result
 * (tools/microbench/exact_rcp.hip, and bbr_selftest_rcp in the library) -- and IEEE division outside that range. */
static inline float bb_rcp(float x) { return 1.0f / x; }
static inline v3 normalize3(v3 a) { return scale3(a, bb_rsqrt(dot3(a, a))); }
static inline float max0(float a) { return a > 0.0f ? a : 0.0f; } /* GLSL max(a,0): NaN -> 0 */

static inline v4 mat4_mul_v4(const bbo_mat4 *m, v4 v) {
  v4 r;
  r.x = fmaf(m->M[3][0], v.w, fmaf(m->M[2][0], v.z, fmaf(m->M[1][0], v.y, m->M[0][0] * v.x)));
  r.y = fmaf(m->M[3][1], v.w, fmaf(m->M[2][1], v.z, fmaf(m->M[1][1], v.y, m->M[0][1] * v.x)));
  r.z = fmaf(m->M[3][2], v.w, fmaf(m->M[2][2], v.z, fmaf(m->M[1][2], v.y, m->M[0][2] * v.x)));
  r.w = fmaf(m->M[3][3], v.w, fmaf(m->M[2][3], v.z, fmaf(m->M[1][3], v.y, m->M[0][3] * v.x)));
  return r;
}

/* ------------------------------------------------------------------------------------------ */
/* forward_brdf.vert (src/shaders/forward_brdf.vert:24-37)                                    */
/* ------------------------------------------------------------------------------------------ */

#define NVARY 14 /* uv(2) posWorld(3) N(3) T(3) B(3); vNormalWorld == vTBN[2] == N */

void bbo_proj_view(const bbo_view_uniforms *view, bbo_mat4 *out) {
  /* GLSL `uProjMat * uViewMat * posWorld` associates left: (P*V) first. forward_brdf.vert:27 */
  for (int c = 0; c < 4; ++c) {
    v4 col = {view->view.M[c][0], view->view.M[c][1], view->view.M[c][2], view->view.M[c][3]};
    v4 r = mat4_mul_v4(&view->proj, col);
    out->M[c][0] = r.x; out->M[c][1] = r.y; out->M[c][2] = r.z; out->M[c][3] = r.w;
  }
}

/* view == NULL: forward_brdf.vert (gl_Position = (P*V) * posWorld, `pv` = P*V).
 * view != NULL: gbuffer.vert:19-22 (posView = V * posWorld; gl_Position = P * posView), `pv` = P.  The other
 * outputs are the same in both shaders (gbuffer.vert:25-35 == forward_brdf.vert:31-36). */
static void vertex_stage(const bbo_mat4 *pv, const bbo_mat4 *view, const bbo_instance *inst, const bbo_vertex *v,
                         float *clip, float *vary) {
  v4 p = {v->pos[0], v->pos[1], v->pos[2], 1.0f};
  v4 pw = mat4_mul_v4(&inst->model, p);   /* :25 */
  v4 c = view ? mat4_mul_v4(pv, mat4_mul_v4(view, pw)) : mat4_mul_v4(pv, pw); /* :27 */
  clip[0] = c.x; clip[1] = c.y; clip[2] = c.z; clip[3] = c.w;
  /* normalMat = transpose(mat3(aInvModel)) (:31): (normalMat*n)_i = dot(InvModel column i, n) */
  const bbo_mat4 *im = &inst->inv_model;
  v3 n = v3_ld(v->normal), t = v3_ld(v->tangent);
  v3 N = normalize3(v3_make(dot3(v3_ld(im->M[0]), n), dot3(v3_ld(im->M[1]), n), dot3(v3_ld(im->M[2]), n)));
  v3 T = normalize3(v3_make(dot3(v3_ld(im->M[0]), t), dot3(v3_ld(im->M[1]), t), dot3(v3_ld(im->M[2]), t)));
  v3 B = cross3(N, T);                    /* :35 */
  vary[0] = v->uv[0]; vary[1] = v->uv[1];
  vary[2] = pw.x; vary[3] = pw.y; vary[4] = pw.z;
  vary[5] = N.x; vary[6] = N.y; vary[7] = N.z;
  vary[8] = T.x; vary[9] = T.y; vary[10] = T.z;
  vary[11] = B.x; vary[12] = B.y; vary[13] = B.z;
}

void bbo_vertex_stage(const bbo_view_uniforms *view, const bbo_instance *inst, const bbo_vertex *v,
                      float *out_clip, float *out_vary) {
  bbo_mat4 pv;
  bbo_proj_view(view, &pv);
  vertex_stage(&pv, NULL, inst, v, out_clip, out_vary);
}

/* ------------------------------------------------------------------------------------------ */
/* texture sampling: SMP_LINEAR, REPEAT, single mip (src/render.cpp:1338-1371, :860)          */
/* ------------------------------------------------------------------------------------------ */

/* `default` material maps, decoded from the PNGs under resources/pbr/default/ (all uniform 16x16):
 * albedo 255 white, metallic 0, roughness 0, ao 255, normal (127,127,255), height 0. */
static const uint8_t k_default_texel[BBO_MAP_COUNT][4] = {
    {255, 255, 255, 255}, {0, 0, 0, 255}, {0, 0, 0, 255}, {255, 255, 255, 255}, {127, 127, 255, 255}, {0, 0, 0, 255}};

static inline int wrap_repeat(int i, int n) {
  int m = i % n;
  return m < 0 ? m + n : m;
}

static void sample_bilinear(const bbo_image *img, int map_type, float u, float v, float *out) {
  const uint8_t *px = img->rgba;
  int w = img->w, h = img->h;
  if (!px || w <= 0 || h <= 0) { /* missing map => default map (uniform => filter is the identity) */
    px = k_default_texel[map_type];
    w = 1; h = 1;
  }
  /* Vulkan: unnormalised = u*size, texel centre at +0.5, i0 = floor(x-0.5), weights = frac */
  float x = fmaf(u, (float)w, -0.5f);
  float y = fmaf(v, (float)h, -0.5f);
  if (!(fabsf(x) < 1073741824.0f)) x = 0.0f; /* non-finite / absurd coordinates are undefined upstream */
  if (!(fabsf(y) < 1073741824.0f)) y = 0.0f;
  float xf = floorf(x), yf = floorf(y);
  float fx = x - xf, fy = y - yf;
  int ix = (int)xf, iy = (int)yf;
  int x0 = wrap_repeat(ix, w), x1 = wrap_repeat(ix + 1, w);
  int y0 = wrap_repeat(iy, h), y1 = wrap_repeat(iy + 1, h);
  const uint8_t *t00 = px + 4 * ((size_t)y0 * w + x0);
  const uint8_t *t10 = px + 4 * ((size_t)y0 * w + x1);
  const uint8_t *t01 = px + 4 * ((size_t)y1 * w + x0);
  const uint8_t *t11 = px + 4 * ((size_t)y1 * w + x1);
  for (int c = 0; c < 4; ++c) {
    float a = (float)t00[c], b = (float)t10[c], cc = (float)t01[c], d = (float)t11[c];
    float top = fmaf(fx, b - a, a);
    float bot = fmaf(fx, d - cc, cc);
    out[c] = fmaf(fy, bot - top, top) * (1.0f / 255.0f);
  }
}

void bbo_sample(const bbo_image *img, int map_type, float u, float v, float *out) {
  sample_bilinear(img, map_type, u, v, out);
}

/* ------------------------------------------------------------------------------------------ */
/* brdf.glsl (src/shaders/brdf.glsl:2-36)                                                     */
/* ------------------------------------------------------------------------------------------ */

#define BB_PI 3.14159265358979323846f     /* brdf.glsl:2, rounds to 0x40490FDB */
#define BB_INV_PI 0.31830988618379067154f /* (float)(1/pi) */

static float distribution_ggx(v3 N, v3 H, float roughness) {
  float a = roughness * roughness;
  float a2 = a * a;
  float NdotH = max0(dot3(N, H));
  float NdotH2 = NdotH * NdotH;
  float denom = fmaf(NdotH2, a2 - 1.0f, 1.0f);
  denom = (BB_PI * denom) * denom;
  return a2 * bb_rcp(denom);
}

static float geometry_schlick_ggx(float NdotV, float roughness) {
  float r = roughness + 1.0f;
  float k = (r * r) * 0.125f;
  float denom = fmaf(NdotV, 1.0f - k, k);
  return NdotV * bb_rcp(denom);
}

static float geometry_smith(v3 N, v3 V, v3 L, float roughness) {
  float NdotV = max0(dot3(N, V));
  float NdotL = max0(dot3(N, L));
  return geometry_schlick_ggx(NdotV, roughness) * geometry_schlick_ggx(NdotL, roughness);
}

static v3 fresnel_schlick(v3 H, v3 V, v3 F0) {
  float x = 1.0f - max0(dot3(H, V));
  float x2 = x * x;
  float p5 = (x2 * x2) * x;
  return v3_make(fmaf(1.0f - F0.x, p5, F0.x), fmaf(1.0f - F0.y, p5, F0.y), fmaf(1.0f - F0.z, p5, F0.z));
}

float bbo_distribution_ggx(const float *N, const float *H, float roughness) {
  return distribution_ggx(v3_ld(N), v3_ld(H), roughness);
}
float bbo_geometry_smith(const float *N, const float *V, const float *L, float roughness) {
  return geometry_smith(v3_ld(N), v3_ld(V), v3_ld(L), roughness);
}
void bbo_fresnel_schlick(const float *H, const float *V, const float *F0, float *out) {
  v3 f = fresnel_schlick(v3_ld(H), v3_ld(V), v3_ld(F0));
  out[0] = f.x; out[1] = f.y; out[2] = f.z;
}

/* ------------------------------------------------------------------------------------------ */
/* forward_brdf.frag (src/shaders/forward_brdf.frag:15-76)                                    */
/* ------------------------------------------------------------------------------------------ */

static inline float mixf(float a, float b, float t) { return fmaf(b, t, a * (1.0f - t)); }
static inline float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

typedef struct {
  v3 P, normal, albedo;
  float metallic, roughness, ao, height;
} surface;

/* texture fetches + normal selection.  forward_brdf.frag:16-25 (gbuffer = 0: the unmapped normal is normalised, :24)
 * or gbuffer.frag:19-32 (gbuffer = 1: the unmapped normal is written as interpolated, :29; height is fetched, :22) */
static void surface_inputs(const bbo_view_uniforms *vu, const bbo_material *mat, const float *vary, int gbuffer, surface *s) {
  float u = vary[0], v = vary[1];
  s->P = v3_ld(vary + 2);
  float tex[4];
  sample_bilinear(&mat->maps[BBO_MAP_ALBEDO], BBO_MAP_ALBEDO, u, v, tex);       /* :16 */
  s->albedo = v3_make(tex[0], tex[1], tex[2]);
  sample_bilinear(&mat->maps[BBO_MAP_METALLIC], BBO_MAP_METALLIC, u, v, tex);   /* :17 */
  s->metallic = tex[0];
  sample_bilinear(&mat->maps[BBO_MAP_ROUGHNESS], BBO_MAP_ROUGHNESS, u, v, tex); /* :18 */
  s->roughness = tex[0];
  sample_bilinear(&mat->maps[BBO_MAP_AO], BBO_MAP_AO, u, v, tex);               /* :19 */
  s->ao = tex[0];
  s->height = 0.0f;
  if (gbuffer) {
    sample_bilinear(&mat->maps[BBO_MAP_HEIGHT], BBO_MAP_HEIGHT, u, v, tex);     /* gbuffer.frag:22 */
    s->height = tex[0];
  }
  if (vu->enable_normal_map != 0) {                                             /* :21-22 */
    sample_bilinear(&mat->maps[BBO_MAP_NORMAL], BBO_MAP_NORMAL, u, v, tex);
    v3 nt = v3_make(fmaf(tex[0], 2.0f, -1.0f), fmaf(tex[1], 2.0f, -1.0f), fmaf(tex[2], 2.0f, -1.0f));
    v3 N = v3_ld(vary + 5), T = v3_ld(vary + 8), B = v3_ld(vary + 11);
    /* vTBN = mat3(T,B,N); vTBN * nt */
    s->normal.x = fmaf(N.x, nt.z, fmaf(B.x, nt.y, T.x * nt.x));
    s->normal.y = fmaf(N.y, nt.z, fmaf(B.y, nt.y, T.y * nt.x));
    s->normal.z = fmaf(N.z, nt.z, fmaf(B.z, nt.y, T.z * nt.x));
  } else {
    s->normal = gbuffer ? v3_ld(vary + 5) : normalize3(v3_ld(vary + 5));        /* :24 / gbuffer.frag:29 */
  }
}

/* the light loop and the ambient term: forward_brdf.frag:27-75 == brdf.frag:26-72 (same statements) */
static void light_surface(const bbo_frame_uniforms *fu, const bbo_view_uniforms *vu, const surface *s, float *out) {
  const v3 P = s->P, normal = s->normal, albedo = s->albedo;
  const float metallic = s->metallic, roughness = s->roughness, ao = s->ao;
  v3 Lo = v3_make(0.0f, 0.0f, 0.0f);
  int n_lights = fu->num_lights;
  if (n_lights > BBO_MAX_LIGHTS) n_lights = BBO_MAX_LIGHTS;
  for (int i = 0; i < n_lights; ++i) {                                          /* :29 */
    const bbo_light *light = &fu->lights[i];
    v3 L;
    float att;
    if (light->type == 0) {
      /* d = length(L); att = 1/(d*d); L = normalize(L)  (:33-36) with 1/d = rsqrt(dot(L,L)) */
      v3 Lv = sub3(v3_ld(light->pos), P);
      float inv_d = bb_rsqrt(dot3(Lv, Lv));
      att = inv_d * inv_d;
      L = scale3(Lv, inv_d);
    } else if (light->type == 1) {
      v3 Lv = sub3(v3_ld(light->pos), P);
      float inv_d = bb_rsqrt(dot3(Lv, Lv));
      att = inv_d * inv_d;
      L = scale3(Lv, inv_d);
      float theta = dot3(L, normalize3(neg3(v3_ld(light->dir))));
      float epsilon = light->inner_cutoff - light->outer_cutoff;
      att *= clamp01((theta - light->outer_cutoff) * bb_rcp(epsilon));
    } else if (light->type == 2) {
      L = neg3(normalize3(v3_ld(light->dir)));
      att = 1.0f;
    } else {
      continue; /* upstream leaves L/att uninitialised (undefined); the contract contributes nothing */
    }

    v3 V = normalize3(sub3(v3_ld(vu->view_pos), P));                            /* :51 */
    v3 N = normalize3(normal);
    v3 H = normalize3(add3(L, V));

    float D = distribution_ggx(N, H, roughness);
    v3 F0 = v3_make(mixf(0.04f, albedo.x, metallic), mixf(0.04f, albedo.y, metallic), mixf(0.04f, albedo.z, metallic));
    v3 F = fresnel_schlick(H, V, F0);
    float G = geometry_smith(N, V, L, roughness);

    v3 radiance = v3_make((att * light->color[0]) * light->intensity, (att * light->color[1]) * light->intensity,
                          (att * light->color[2]) * light->intensity);

    float NdotV = max0(dot3(V, N));
    float NdotL = max0(dot3(L, N));
    float sden = (4.0f * NdotV) * NdotL;
    if (!(sden > 0.001f)) sden = 0.001f;                                        /* max(.., 0.001) */
    float rden = bb_rcp(sden);
    v3 spec = v3_make(((D * F.x) * G) * rden, ((D * F.y) * G) * rden, ((D * F.z) * G) * rden);
    float om = 1.0f - metallic;
    v3 kD = v3_make((1.0f - F.x) * om, (1.0f - F.y) * om, (1.0f - F.z) * om);

    Lo.x = fmaf(fmaf(kD.x * albedo.x, BB_INV_PI, spec.x) * radiance.x, NdotL, Lo.x);
    Lo.y = fmaf(fmaf(kD.y * albedo.y, BB_INV_PI, spec.y) * radiance.y, NdotL, Lo.y);
    Lo.z = fmaf(fmaf(kD.z * albedo.z, BB_INV_PI, spec.z) * radiance.z, NdotL, Lo.z);
  }

  out[0] = fmaf(0.03f * albedo.x, ao, Lo.x);                                    /* :72-73 */
  out[1] = fmaf(0.03f * albedo.y, ao, Lo.y);
  out[2] = fmaf(0.03f * albedo.z, ao, Lo.z);
  out[3] = 1.0f;
}

/* ------------------------------------------------------------------------------------------ */
/* The SHIPPED evaluation order of the same light loop ("contract" mode; the default of bbo_render /
 * bbo_render_deferred).  light_surface() above follows the GLSL statement by statement; this one is what
 * k_shade executes, bit for bit.  It differs from the literal form only DOWNSTREAM of the one
 * ill-conditioned quantity of the shader -- the GGX denominator q = NdotH^2 (a^2 - 1) + 1, which cancels to
 * ~a^2 at a highlight's peak and amplifies a 1-ulp change of N, V, L or H about 2000-fold.  Everything that
 * feeds q (V, N, L, att, H, NdotH) is computed exactly as in light_surface().  Behind q the products are
 * re-associated the way a shader compiler does for a non-`precise` GLSL expression:
 *   D G / max(4 NdotV NdotL, .001)  =  (a2 NdotV NdotL) / ((q q) (PI dV dL) sden)     one reciprocal instead of four
 *       with dX = NdotX (1 - k) + k (geometrySchlickGGX's denominator), sden = max(4 NdotV NdotL, .001)
 *   kD albedo / PI                  =  (1 - F) ((1 - metallic) albedo (1/PI))         hoisted out of the loop
 *   radiance NdotL                  =  (color intensity) (att NdotL)                  color*intensity once per light
 *   max(N.V,0), max(N.L,0), max(H.V,0) = saturate(...)  for the three cosines that do not feed q (N.H keeps max(.,0)):
 *                                      differs only where rounding puts a cosine of unit vectors above 1 (<= 2 ulp)
 * The per-light constants (color*intensity, the normalised spot / directional direction, 1/epsilon) are
 * evaluated once per light and frame ("cooked" lights) -- the same operations as the literal form, hoisted.
 * The two forms agree to a few ulp of the result (tests/test_oracle_contract.py checks <= 1e-5 relative on
 * random inputs and <= 1e-4 max(1,|ref|) on whole frames); the GPU is compared bit for bit with THIS form
 * and within BASELINE's tolerance with the literal one.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  v3 pos;
  int32_t type;
  v3 ci;      /* color * intensity */
  v3 dirn;    /* type 1: normalize(-dir); type 2: -normalize(dir) */
  float outer, inv_eps;
} cooked_light;

static void cook_light(const bbo_light *l, cooked_light *c) {
  c->pos = v3_ld(l->pos);
  c->type = l->type;
  c->ci = v3_make(l->color[0] * l->intensity, l->color[1] * l->intensity, l->color[2] * l->intensity);
  c->dirn = v3_make(0.0f, 0.0f, 0.0f);
  c->outer = l->outer_cutoff;
  c->inv_eps = 0.0f;
  if (l->type == 1) {
    c->dirn = normalize3(neg3(v3_ld(l->dir)));
    c->inv_eps = bb_rcp(l->inner_cutoff - l->outer_cutoff);
  } else if (l->type == 2) {
    c->dirn = neg3(normalize3(v3_ld(l->dir)));
  }
}

static inline float fmax_nan_lo(float a, float lo) { return a > lo ? a : lo; } /* max(a, lo), NaN -> lo */
static inline float sat01(float a) { return a > 0.0f ? (a < 1.0f ? a : 1.0f) : 0.0f; } /* clamp to [0,1], NaN -> 0 */

static void light_surface_contract(const bbo_frame_uniforms *fu, const bbo_view_uniforms *vu, const surface *s, float *out) {
  const v3 P = s->P, albedo = s->albedo;
  const float metallic = s->metallic, roughness = s->roughness, ao = s->ao;
  int n_lights = fu->num_lights;
  if (n_lights > BBO_MAX_LIGHTS) n_lights = BBO_MAX_LIGHTS;
  /* per-pixel invariants */
  const v3 V = normalize3(sub3(v3_ld(vu->view_pos), P));
  const v3 N = normalize3(s->normal);
  const float NdotV = sat01(dot3(V, N));
  const float rr = roughness + 1.0f;
  const float kk = (rr * rr) * 0.125f, omk = 1.0f - kk;
  const float pidV = BB_PI * fmaf(NdotV, omk, kk);
  const float a = roughness * roughness, a2 = a * a, a2m1 = a2 - 1.0f;
  const float a2nv = a2 * NdotV, c4 = 4.0f * NdotV;
  const v3 F0 = v3_make(mixf(0.04f, albedo.x, metallic), mixf(0.04f, albedo.y, metallic), mixf(0.04f, albedo.z, metallic));
  const v3 omF0 = v3_make(1.0f - F0.x, 1.0f - F0.y, 1.0f - F0.z);
  const float om = 1.0f - metallic;
  const v3 kda = v3_make((om * albedo.x) * BB_INV_PI, (om * albedo.y) * BB_INV_PI, (om * albedo.z) * BB_INV_PI);

  v3 Lo = v3_make(0.0f, 0.0f, 0.0f);
  for (int i = 0; i < n_lights; ++i) {
    cooked_light cl;
    cook_light(&fu->lights[i], &cl);
    v3 L;
    float att;
    if (cl.type == 0 || cl.type == 1) {
      v3 Lv = sub3(cl.pos, P);
      float inv_d = bb_rsqrt(dot3(Lv, Lv));
      att = inv_d * inv_d;
      L = scale3(Lv, inv_d);
      if (cl.type == 1) att *= clamp01((dot3(L, cl.dirn) - cl.outer) * cl.inv_eps);
    } else if (cl.type == 2) {
      L = cl.dirn;
      att = 1.0f;
    } else {
      continue;
    }
    const v3 H = normalize3(add3(L, V));
    const float NdotH = max0(dot3(N, H));
    const float q = fmaf(NdotH * NdotH, a2m1, 1.0f);
    const float x = 1.0f - sat01(dot3(H, V));
    const float x2 = x * x;
    const float p5 = (x2 * x2) * x;
    const float NdotL = sat01(dot3(N, L));
    const float dL = fmaf(NdotL, omk, kk);
    const float sden = fmax_nan_lo(c4 * NdotL, 0.001f);
    const float den = ((q * q) * (pidV * dL)) * sden;
    const float S = (a2nv * NdotL) * bb_rcp(den);
    const v3 F = v3_make(fmaf(omF0.x, p5, F0.x), fmaf(omF0.y, p5, F0.y), fmaf(omF0.z, p5, F0.z));
    const float rl = att * NdotL;
    Lo.x = fmaf(fmaf(1.0f - F.x, kda.x, F.x * S), cl.ci.x * rl, Lo.x);
    Lo.y = fmaf(fmaf(1.0f - F.y, kda.y, F.y * S), cl.ci.y * rl, Lo.y);
    Lo.z = fmaf(fmaf(1.0f - F.z, kda.z, F.z * S), cl.ci.z * rl, Lo.z);
  }
  out[0] = fmaf(0.03f * albedo.x, ao, Lo.x);
  out[1] = fmaf(0.03f * albedo.y, ao, Lo.y);
  out[2] = fmaf(0.03f * albedo.z, ao, Lo.z);
  out[3] = 1.0f;
}

/* literal != 0: the statement-by-statement form (BBO_FLAG_LITERAL); 0: the shipped evaluation order */
static void light_surface_mode(int literal, const bbo_frame_uniforms *fu, const bbo_view_uniforms *vu, const surface *s, float *out) {
  if (literal) light_surface(fu, vu, s, out);
  else light_surface_contract(fu, vu, s, out);
}

static void shade_fragment(int literal, const bbo_frame_uniforms *fu, const bbo_view_uniforms *vu, const bbo_material *mat,
                           const float *vary, float *out) {
  surface s;
  surface_inputs(vu, mat, vary, 0, &s);
  light_surface_mode(literal, fu, vu, &s, out);
}

/* One G-buffer texel: four R16G16B16A16_SFLOAT attachments (src/main.cpp:443, 453-459) as 16 binary32 values that
 * are exactly representable in binary16: position.xyz 1 | normal.xyz 0 | albedo.rgb 0 | metallic roughness ao height.
 * (vPosWorld.w interpolates the constant 1 and rounds to 1; the alpha of the vec3 outputs is undefined in Vulkan, 0
 * here; outMaterialIndex, gbuffer.frag:33, is a constant nobody reads.) */
static void gbuffer_fragment(const bbo_view_uniforms *vu, const bbo_material *mat, const float *vary, float *g) {
  surface s;
  surface_inputs(vu, mat, vary, 1, &s);
  g[0] = bbo_half_round(s.P.x); g[1] = bbo_half_round(s.P.y); g[2] = bbo_half_round(s.P.z); g[3] = 1.0f;
  g[4] = bbo_half_round(s.normal.x); g[5] = bbo_half_round(s.normal.y); g[6] = bbo_half_round(s.normal.z); g[7] = 0.0f;
  g[8] = bbo_half_round(s.albedo.x); g[9] = bbo_half_round(s.albedo.y); g[10] = bbo_half_round(s.albedo.z); g[11] = 0.0f;
  g[12] = bbo_half_round(s.metallic); g[13] = bbo_half_round(s.roughness); g[14] = bbo_half_round(s.ao);
  g[15] = bbo_half_round(s.height);
}

/* brdf.frag:12-73 on one G-buffer texel (nearest fetch of the pixel's own texel) */
static void brdf_pixel(int literal, const bbo_frame_uniforms *fu, const bbo_view_uniforms *vu, const float *g, float *out) {
  surface s;
  s.P = v3_ld(g); s.normal = v3_ld(g + 4); s.albedo = v3_ld(g + 8);
  s.metallic = g[12]; s.roughness = g[13]; s.ao = g[14]; s.height = g[15];
  light_surface_mode(literal, fu, vu, &s, out);
}

void bbo_shade_fragment(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const bbo_material *mat,
                        const float *vary, float *out_rgba) {
  shade_fragment(1, frame, view, mat, vary, out_rgba);
}
void bbo_shade_fragment_contract(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const bbo_material *mat,
                                 const float *vary, float *out_rgba) {
  shade_fragment(0, frame, view, mat, vary, out_rgba);
}
/* both forms of the light loop on an explicit surface point: surface[12] = P(3) normal(3) albedo(3) metallic roughness ao */
void bbo_light_surface(int literal, const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const float *surf, float *out_rgba) {
  surface s;
  s.P = v3_ld(surf); s.normal = v3_ld(surf + 3); s.albedo = v3_ld(surf + 6);
  s.metallic = surf[9]; s.roughness = surf[10]; s.ao = surf[11]; s.height = 0.0f;
  light_surface_mode(literal, frame, view, &s, out_rgba);
}

/* ------------------------------------------------------------------------------------------ */
/* presentation: HDR attachment -> tone map -> sRGB 8-bit swapchain image      */
/*   hdr_tone_mapping.frag:9-18 (mapped = 1 - exp(-hdr * exposure), alpha = 1)                      */
/*   HDR attachment is R16G16B16A16_SFLOAT (src/render.h:94, src/main.cpp:463-472): the tone-map    */
/*     subpass reads binary16 values                                                                */
/*   swapchain format R8G8B8A8_SRGB / B8G8R8A8_SRGB (src/render.cpp:242-254): the colour output is  */
/*     sRGB-encoded and rounded to UNORM8 by the attachment write                                   */
/* Contract (what a Vulkan driver leaves open, fixed here so that GPU and oracle agree on every      */
/* byte): binary16 conversion rounds to nearest even; exp is the fixed sequence bb_exp below (<= 1   */
/* ulp from the real exp); the sRGB byte is the number of thresholds t_k <= c, t_k = float(decode(   */
/* (k - 0.5) / 255)) for k = 1..255 -- the exactly rounded ideal curve (NaN -> 0).  Parity with a     */
/* particular driver's ROP is unpinned (the Vulkan spec allows 0.6 ULP8 of slack).                   */
/* ------------------------------------------------------------------------------------------ */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* round to the nearest binary16 value (ties to even) and return it as binary32 */
float bbo_half_round(float x) {
  const uint32_t u = f2u(x), sign = u & 0x80000000u, a = u & 0x7FFFFFFFu;
  if (a >= 0x7F800000u) return x;                          /* inf, NaN */
  if (a >= 0x477FF000u) return u2f(sign | 0x7F800000u);    /* >= 65520 = halfway between 65504 and 2^16 -> inf */
  if (a < 0x38800000u) {                                   /* below 2^-14: the subnormal grid, spacing 2^-24 */
    const float r = (u2f(a) + 0.5f) - 0.5f;               /* ulp(0.5) = 2^-24: the add rounds to the grid (RNE) */
    return u2f(sign | f2u(r));
  }
  const uint32_t r = (a + 0x00000FFFu + ((a >> 13) & 1u)) & 0xFFFFE000u; /* keep 10 mantissa bits, RNE */
  return u2f(sign | r);
}

void bbo_half_round_n(const float *in, float *out, size_t n) {
  for (size_t i = 0; i < n; ++i) out[i] = bbo_half_round(in[i]);
}

/* exp as a fixed sequence of binary32 operations: Cody-Waite reduction by ln2 (hi/lo), degree-7 Taylor/Horner in
 * fma, scaling by 2^n in two exact-until-the-last multiplications.  Max error measured over 2^24 points: < 1 ulp. */
float bbo_exp(float x) {
  if (!(x >= -104.0f)) return x < -104.0f ? 0.0f : x; /* underflow to 0; NaN stays NaN */
  if (x > 88.7228317f) return INFINITY;
  const float n = rintf(x * 1.44269502f);
  float r = fmaf(n, -0.693145752f, x);
  r = fmaf(n, -1.42860677e-06f, r);
  float p = 1.98412701e-04f;         /* 1/5040 */
  p = fmaf(p, r, 1.38888892e-03f);   /* 1/720 */
  p = fmaf(p, r, 8.33333377e-03f);   /* 1/120 */
  p = fmaf(p, r, 4.16666679e-02f);   /* 1/24 */
  p = fmaf(p, r, 1.66666672e-01f);   /* 1/6 */
  p = fmaf(p, r, 0.5f);
  p = fmaf(p, r, 1.0f);
  p = fmaf(p, r, 1.0f);
  const int32_t ni = (int32_t)n, h = ni / 2;
  const float s1 = u2f((uint32_t)(h + 127) << 23), s2 = u2f((uint32_t)(ni - h + 127) << 23);
  return (p * s1) * s2;
}

/* out[k-1] = t_k, k = 1..255 (ascending) */
void bbo_srgb_thresholds(float *out255) {
  for (int k = 1; k <= 255; ++k) {
    const double b = ((double)k - 0.5) / 255.0;
    const double lin = b <= 0.04045 ? b / 12.92 : pow((b + 0.055) / 1.055, 2.4);
    out255[k - 1] = (float)lin;
  }
}

static inline uint8_t srgb8(float c, const float *thr) { /* number of thresholds <= c; thr[255] = +inf */
  uint32_t pos = 0;
  for (uint32_t step = 128; step; step >>= 1)
    if (thr[pos + step - 1] <= c) pos += step;
  return (uint8_t)pos;
}

void bbo_tone_map(float *rgba, uint64_t n_pixels, int32_t enable, float exposure) {
  for (uint64_t i = 0; i < n_pixels; ++i) {
    float *p = rgba + 4 * i;
    if (enable) {
      for (int c = 0; c < 3; ++c) p[c] = 1.0f - bbo_exp(-p[c] * exposure);
    }
    p[3] = 1.0f;
  }
}

void bbo_present(const float *rgba, uint64_t n_pixels, int32_t enable, float exposure, int32_t hdr16, uint8_t *out_rgba8) {
  float thr[256];
  bbo_srgb_thresholds(thr);
  thr[255] = INFINITY;
  for (uint64_t i = 0; i < n_pixels; ++i) {
    for (int c = 0; c < 3; ++c) {
      float v = rgba[4 * i + c];
      if (hdr16) v = bbo_half_round(v);
      if (enable) v = 1.0f - bbo_exp(-v * exposure);
      out_rgba8[4 * i + c] = srgb8(v, thr);
    }
    out_rgba8[4 * i + 3] = 255; /* outColor.a = 1.0; alpha is not sRGB-encoded */
  }
}

/* ------------------------------------------------------------------------------------------ */
/* fixed function: clip, viewport, snap, cull, coverage, depth                                */
/* (state: src/render.cpp:1069-1125; forward params src/main.cpp:332-350; clears src/main.cpp:84)*/
/* ------------------------------------------------------------------------------------------ */

#define GUARD_BAND 32.0f  /* x/y clip planes sit at +-32w; the scissor does the rest */
#define SUBPIXEL_BITS 8
#define SUBPIXEL_ONE 256
#define MAX_CLIP_VERTS 12
#define MAX_SUBTRIS 8     /* primitive id occupies key >> 3 */

typedef struct {
  float c[4]; /* clip x y z w */
  float b[3]; /* barycentrics with respect to the unclipped triangle */
} clip_vert;

typedef struct {
  int32_t X[3], Y[3];         /* 24.8 snapped framebuffer coordinates */
  float z0, dzdx, dzdy;       /* depth plane relative to vertex 0, per sub-pixel unit */
  float l1dx, l1dy, l2dx, l2dy; /* screen-space barycentric planes relative to vertex 0 */
  float rw[3];                /* 1/w at the (sub-)triangle's own vertices */
  float bary[3][3];           /* own vertex j -> barycentrics wrt the original triangle */
  int clipped;
} raster_tri;

static inline float plane_dist(const float *c, int plane) {
  switch (plane) {
  case 0: return c[3] - c[2];                 /* near: z <= w (reverse-Z: NDC z = 1) */
  case 1: return c[2];                        /* far:  z >= 0 */
  case 2: return fmaf(GUARD_BAND, c[3], c[0]);
  case 3: return fmaf(GUARD_BAND, c[3], -c[0]);
  case 4: return fmaf(GUARD_BAND, c[3], c[1]);
  default: return fmaf(GUARD_BAND, c[3], -c[1]);
  }
}

/* intersection is always evaluated from the inside vertex towards the outside one, so the two
 * triangles sharing an edge produce the same new vertex */
static void clip_lerp(const clip_vert *in, float din, const clip_vert *out, float dout, clip_vert *r) {
  float t = din / (din - dout);
  for (int k = 0; k < 4; ++k) r->c[k] = fmaf(t, out->c[k] - in->c[k], in->c[k]);
  for (int k = 0; k < 3; ++k) r->b[k] = fmaf(t, out->b[k] - in->b[k], in->b[k]);
}

static int clip_polygon(clip_vert *poly, int n) {
  clip_vert tmp[MAX_CLIP_VERTS];
  for (int plane = 0; plane < 6; ++plane) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const clip_vert *a = &poly[i];
      const clip_vert *b = &poly[(i + 1) % n];
      float da = plane_dist(a->c, plane), db = plane_dist(b->c, plane);
      int ina = da >= 0.0f, inb = db >= 0.0f;
      if (ina) tmp[m++] = *a;
      if (ina != inb) {
        if (ina) clip_lerp(a, da, b, db, &tmp[m]);
        else clip_lerp(b, db, a, da, &tmp[m]);
        ++m;
      }
    }
    n = m;
    if (n < 3) return 0;
    memcpy(poly, tmp, sizeof(clip_vert) * (size_t)n);
  }
  return n;
}

/* project + snap one clip-space vertex; returns 0 if it cannot be represented */
/* Viewport transform xf = (px / 2) * xd + (x + px / 2), Vulkan 1.2 sec. 26.9: half extents and centre in pixels.  The
 * main passes use the whole target (centre = half extent); the gizmo overlay uses a 100 x 100 corner (src/main.cpp:760-772). */
typedef struct {
  float half_w, half_h, cx, cy;
} viewport;

static int project_vertex(const float *c, const viewport *vp, int32_t *X, int32_t *Y, float *rw, float *zndc) {
  float w = c[3];
  if (!(w > 0.0f)) return 0;
  float r = 1.0f / w;
  float xs = fmaf(c[0] * r, vp->half_w, vp->cx);
  float ys = fmaf(c[1] * r, vp->half_h, vp->cy);
  if (!(fabsf(xs) <= 4194304.0f) || !(fabsf(ys) <= 4194304.0f)) return 0;
  *X = (int32_t)rintf(xs * 256.0f);
  *Y = (int32_t)rintf(ys * 256.0f);
  *rw = r;
  *zndc = c[2] * r;
  return 1;
}

/* Finish one (sub-)triangle: cull (front = CLOCKWISE in y-down framebuffer space = positive
 * doubled area, src/render.cpp:1097-1098 + Vulkan 1.2 sec. 27.12.1), then planes. */
static int setup_tri(raster_tri *t, const float *z) {
  int64_t dx1 = (int64_t)t->X[1] - t->X[0], dy1 = (int64_t)t->Y[1] - t->Y[0];
  int64_t dx2 = (int64_t)t->X[2] - t->X[0], dy2 = (int64_t)t->Y[2] - t->Y[0];
  int64_t S = dx1 * dy2 - dx2 * dy1;
  if (S <= 0) return 0; /* back-facing or zero area */
  double rS = 1.0 / (double)S;
  t->l1dx = (float)((double)dy2 * rS);
  t->l1dy = (float)(-(double)dx2 * rS);
  t->l2dx = (float)(-(double)dy1 * rS);
  t->l2dy = (float)((double)dx1 * rS);
  double dz1 = (double)z[1] - (double)z[0], dz2 = (double)z[2] - (double)z[0];
  t->z0 = z[0];
  t->dzdx = (float)((dz1 * (double)dy2 - dz2 * (double)dy1) * rS);
  t->dzdy = (float)((dz2 * (double)dx1 - dz1 * (double)dx2) * rS);
  return 1;
}

/* Build the raster triangles of one primitive from its three clip-space vertices. */
static int build_prim(const float clip[3][4], const viewport *vp, raster_tri *out, int *was_clipped) {
  *was_clipped = 0;
  /* trivial reject against the true frustum planes (cannot change any pixel: the viewport scissor
   * removes everything outside anyway) */
  {
    int o_l = 1, o_r = 1, o_t = 1, o_b = 1, o_n = 1, o_f = 1;
    for (int i = 0; i < 3; ++i) {
      const float *c = clip[i];
      o_l &= (c[3] + c[0] < 0.0f); o_r &= (c[3] - c[0] < 0.0f);
      o_t &= (c[3] + c[1] < 0.0f); o_b &= (c[3] - c[1] < 0.0f);
      o_n &= (c[3] - c[2] < 0.0f); o_f &= (c[2] < 0.0f);
    }
    if (o_l | o_r | o_t | o_b | o_n | o_f) return 0;
  }
  int all_in = 1;
  for (int i = 0; i < 3 && all_in; ++i)
    for (int p = 0; p < 6; ++p)
      if (!(plane_dist(clip[i], p) >= 0.0f)) { all_in = 0; break; }

  if (all_in) {
    raster_tri *t = &out[0];
    float z[3];
    for (int i = 0; i < 3; ++i)
      if (!project_vertex(clip[i], vp, &t->X[i], &t->Y[i], &t->rw[i], &z[i])) return 0;
    t->clipped = 0;
    memset(t->bary, 0, sizeof t->bary);
    t->bary[0][0] = t->bary[1][1] = t->bary[2][2] = 1.0f;
    return setup_tri(t, z);
  }

  *was_clipped = 1;
  clip_vert poly[MAX_CLIP_VERTS];
  for (int i = 0; i < 3; ++i) {
    memcpy(poly[i].c, clip[i], sizeof(float) * 4);
    poly[i].b[0] = poly[i].b[1] = poly[i].b[2] = 0.0f;
    poly[i].b[i] = 1.0f;
  }
  int n = clip_polygon(poly, 3);
  if (n < 3) return 0;
  int32_t X[MAX_CLIP_VERTS], Y[MAX_CLIP_VERTS];
  float rw[MAX_CLIP_VERTS], z[MAX_CLIP_VERTS];
  for (int i = 0; i < n; ++i)
    if (!project_vertex(poly[i].c, vp, &X[i], &Y[i], &rw[i], &z[i])) return 0;
  /* fan (0, i, i+1); slot index = i-1 is kept even when a fan triangle is culled so that the sub-triangle
   * number is a pure function of the clipped polygon */
  int count = 0;
  for (int i = 1; i + 1 < n && i - 1 < MAX_SUBTRIS; ++i) {
    raster_tri *t = &out[i - 1];
    const int id[3] = {0, i, i + 1};
    float zz[3];
    for (int k = 0; k < 3; ++k) {
      t->X[k] = X[id[k]]; t->Y[k] = Y[id[k]]; t->rw[k] = rw[id[k]]; zz[k] = z[id[k]];
      memcpy(t->bary[k], poly[id[k]].b, sizeof(float) * 3);
    }
    t->clipped = setup_tri(t, zz) ? 1 : -1; /* -1 marks a culled slot */
    count = i;
  }
  return count; /* number of slots (some may be culled) */
}

static inline int edge_top_left(int64_t dx, int64_t dy) { return dy < 0 || (dy == 0 && dx > 0); }

/* coverage test of pixel centre (Xc,Yc) in 24.8 */
static inline int covers(const raster_tri *t, int64_t Xc, int64_t Yc) {
  for (int i = 0; i < 3; ++i) {
    int j = i == 2 ? 0 : i + 1;
    int64_t dx = (int64_t)t->X[j] - t->X[i], dy = (int64_t)t->Y[j] - t->Y[i];
    int64_t E = dx * (Yc - t->Y[i]) - dy * (Xc - t->X[i]);
    if (E < 0 || (E == 0 && !edge_top_left(dx, dy))) return 0;
  }
  return 1;
}

static inline float tri_depth(const raster_tri *t, int32_t Xc, int32_t Yc) {
  float dxp = (float)(Xc - t->X[0]), dyp = (float)(Yc - t->Y[0]);
  float z = fmaf(t->dzdx, dxp, fmaf(t->dzdy, dyp, t->z0));
  if (!(z >= 0.0f)) z = 0.0f; /* depth is clamped to the [0,1] viewport range; NaN -> 0 */
  if (z > 1.0f) z = 1.0f;
  return z;
}

/* perspective-correct barycentrics wrt the ORIGINAL triangle at a pixel centre */
static inline void tri_bary(const raster_tri *t, int32_t Xc, int32_t Yc, float *beta) {
  float dxp = (float)(Xc - t->X[0]), dyp = (float)(Yc - t->Y[0]);
  float l1 = fmaf(t->l1dx, dxp, t->l1dy * dyp);
  float l2 = fmaf(t->l2dx, dxp, t->l2dy * dyp);
  float l0 = (1.0f - l1) - l2;
  float u0 = l0 * t->rw[0], u1 = l1 * t->rw[1], u2 = l2 * t->rw[2];
  float r = bb_rcp((u0 + u1) + u2);
  float b0 = u0 * r, b1 = u1 * r, b2 = u2 * r;
  if (t->clipped) {
    for (int k = 0; k < 3; ++k) beta[k] = fmaf(b2, t->bary[2][k], fmaf(b1, t->bary[1][k], b0 * t->bary[0][k]));
  } else {
    beta[0] = b0; beta[1] = b1; beta[2] = b2;
  }
}

static inline void interpolate(const float *beta, const float vary[3][NVARY], int n, float *out) {
  for (int k = 0; k < n; ++k) out[k] = fmaf(beta[2], vary[2][k], fmaf(beta[1], vary[1][k], beta[0] * vary[0][k]));
}

/* ------------------------------------------------------------------------------------------ */
/* frame driver                                                                               */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  const bbo_draw *draw;
  uint32_t first_prim; /* global primitive index of this draw's first triangle */
  uint32_t tris_per_instance;
} draw_info;

typedef void (*prim_fetch_fn)(const void *ctx, uint32_t prim, float clip[3][4], float vary[3][NVARY], const void **mat);

typedef struct {
  const bbo_frame_uniforms *fu;
  const bbo_view_uniforms *vu;
  bbo_mat4 pv;
  int deferred; /* gbuffer.vert: clip = P * (V * posWorld) */
  const draw_info *draws;
  uint32_t n_draws;
} pbr_ctx;

static void pbr_fetch(const void *vctx, uint32_t prim, float clip[3][4], float vary[3][NVARY], const void **mat) {
  const pbr_ctx *c = (const pbr_ctx *)vctx;
  uint32_t d = 0;
  while (d + 1 < c->n_draws && prim >= c->draws[d + 1].first_prim) ++d;
  const draw_info *di = &c->draws[d];
  uint32_t local = prim - di->first_prim;
  uint32_t inst = local / di->tris_per_instance, tri = local % di->tris_per_instance;
  for (int k = 0; k < 3; ++k) {
    uint32_t vi = di->draw->indices ? di->draw->indices[3 * tri + k] : 3 * tri + k;
    if (c->deferred)
      vertex_stage(&c->vu->proj, &c->vu->view, &di->draw->instances[inst], &di->draw->vertices[vi], clip[k], vary[k]);
    else
      vertex_stage(&c->pv, NULL, &di->draw->instances[inst], &di->draw->vertices[vi], clip[k], vary[k]);
  }
  *mat = di->draw->material;
}

typedef struct {
  const bbo_view_uniforms *vu;
  bbo_mat4 view, pv;
  const bbo_gizmo_vertex *verts;
  const uint32_t *indices;
} gizmo_ctx;

static void gizmo_fetch(const void *vctx, uint32_t prim, float clip[3][4], float vary[3][NVARY], const void **mat) {
  const gizmo_ctx *c = (const gizmo_ctx *)vctx;
  for (int k = 0; k < 3; ++k) {
    const bbo_gizmo_vertex *gv = &c->verts[c->indices ? c->indices[3 * prim + k] : 3 * prim + k];
    v4 p = {gv->pos[0], gv->pos[1], gv->pos[2], 1.0f};
    v4 pc = mat4_mul_v4(&c->pv, p); /* gl_Position = (projMat * viewMat) * pos, gizmo.vert:25 */
    clip[k][0] = pc.x; clip[k][1] = pc.y; clip[k][2] = pc.z; clip[k][3] = pc.w;
    memset(vary[k], 0, sizeof(float) * NVARY);
    vary[k][0] = gv->color[0]; vary[k][1] = gv->color[1]; vary[k][2] = gv->color[2];
    /* vNormal = mat3(viewMat) * aNormal (gizmo.vert:27) */
    v3 n = v3_ld(gv->normal);
    vary[k][3] = fmaf(c->view.M[2][0], n.z, fmaf(c->view.M[1][0], n.y, c->view.M[0][0] * n.x));
    vary[k][4] = fmaf(c->view.M[2][1], n.z, fmaf(c->view.M[1][1], n.y, c->view.M[0][1] * n.x));
    vary[k][5] = fmaf(c->view.M[2][2], n.z, fmaf(c->view.M[1][2], n.y, c->view.M[0][2] * n.x));
  }
  *mat = NULL;
}

static void gizmo_shade(const float *vary, float *out) {
  /* gizmo.frag:10-17: L = -(0,0,1); diff = max(dot(L, normalize(vNormal)), 0) */
  v3 N = normalize3(v3_ld(vary + 3));
  float diff = max0(dot3(v3_make(-0.0f, -0.0f, -1.0f), N));
  out[0] = vary[0] * diff; out[1] = vary[1] * diff; out[2] = vary[2] * diff; out[3] = 1.0f;
}

typedef struct {
  int program; /* 0 = forward PBR, 1 = gizmo, 2 = deferred PBR (G-buffer write + brdf.frag) */
  const void *ctx;
  prim_fetch_fn fetch;
  const bbo_frame_uniforms *fu;
  const bbo_view_uniforms *vu;
  float *gbuffer; /* program 2, optional: width*height*16 floats */
  int32_t width;
  int literal;    /* BBO_FLAG_LITERAL: statement-by-statement light loop instead of the shipped evaluation order */
  int output_uv;  /* BBO_FLAG_OUTPUT_UV: the interpolated vUV instead of the colour (measurement aid, forward program only) */
} pipeline;

static void shade_pixel(const pipeline *pl, const raster_tri *t, const float vary[3][NVARY], const void *mat,
                        int32_t px, int32_t py, float *out) {
  float beta[3], attr[NVARY];
  tri_bary(t, px * SUBPIXEL_ONE + SUBPIXEL_ONE / 2, py * SUBPIXEL_ONE + SUBPIXEL_ONE / 2, beta);
  if (pl->program == 0) {
    interpolate(beta, vary, NVARY, attr);
    if (pl->output_uv) { out[0] = attr[0]; out[1] = attr[1]; out[2] = 0.0f; out[3] = 1.0f; return; }
    shade_fragment(pl->literal, pl->fu, pl->vu, (const bbo_material *)mat, attr, out);
  } else if (pl->program == 2) {
    float g[16];
    interpolate(beta, vary, NVARY, attr);
    gbuffer_fragment(pl->vu, (const bbo_material *)mat, attr, g);
    if (pl->gbuffer) memcpy(pl->gbuffer + 16 * ((size_t)py * (size_t)pl->width + (size_t)px), g, sizeof g);
    brdf_pixel(pl->literal, pl->fu, pl->vu, g, out);
  } else {
    interpolate(beta, vary, 6, attr);
    gizmo_shade(attr, out);
  }
}

static int render_core(const pipeline *pl, uint32_t n_prims, int32_t width, int32_t height, int32_t y0, int32_t y1,
                       uint32_t flags, float *out_rgba, uint32_t *out_prim, float *out_depth, bbo_stats *stats) {
  if (width <= 0 || height <= 0 || !out_rgba) return -1;
  if (y0 < 0) y0 = 0;
  if (y1 > height) y1 = height;
  if (n_prims >= (1u << 29)) return -2;
  size_t npx = (size_t)width * (size_t)height;
  int own_depth = 0;
  uint32_t *key = (uint32_t *)malloc(npx * sizeof(uint32_t)); /* (prim << 3 | sub) + 1, 0 = empty */
  if (!key) return -3;
  if (!out_depth) {
    out_depth = (float *)malloc(npx * sizeof(float));
    if (!out_depth) { free(key); return -3; }
    own_depth = 1;
  }
  bbo_stats st;
  memset(&st, 0, sizeof st);
  st.n_prims = n_prims;
  if (pl->program == 2) flags &= ~(uint32_t)BBO_FLAG_FORWARD_SHADE; /* the G-buffer keeps the depth-test winner */
  for (int32_t y = y0; y < y1; ++y) {
    if (pl->gbuffer) memset(pl->gbuffer + 16 * (size_t)y * width, 0, sizeof(float) * 16 * (size_t)width); /* src/main.cpp:84 */
    memset(out_rgba + 4 * (size_t)y * width, 0, sizeof(float) * 4 * (size_t)width);   /* clear colour 0 */
    memset(out_depth + (size_t)y * width, 0, sizeof(float) * (size_t)width);            /* clear depth 0  */
    memset(key + (size_t)y * width, 0, sizeof(uint32_t) * (size_t)width);
  }
  const viewport whole = {0.5f * (float)width, 0.5f * (float)height, 0.5f * (float)width, 0.5f * (float)height};

  /* ---- pass 1: visibility in API order, depth op GREATER_OR_EQUAL (src/render.cpp:1121) ---- */
  for (uint32_t prim = 0; prim < n_prims; ++prim) {
    float clip[3][4], vary[3][NVARY];
    const void *mat;
    pl->fetch(pl->ctx, prim, clip, vary, &mat);
    raster_tri tris[MAX_SUBTRIS];
    int was_clipped;
    int n = build_prim(clip, &whole, tris, &was_clipped);
    st.n_clipped_prims += (uint64_t)was_clipped;
    for (int s = 0; s < n; ++s) {
      const raster_tri *t = &tris[s];
      if (t->clipped < 0) continue;
      ++st.n_raster_tris;
      int32_t minX = t->X[0], maxX = t->X[0], minY = t->Y[0], maxY = t->Y[0];
      for (int k = 1; k < 3; ++k) {
        if (t->X[k] < minX) minX = t->X[k];
        if (t->X[k] > maxX) maxX = t->X[k];
        if (t->Y[k] < minY) minY = t->Y[k];
        if (t->Y[k] > maxY) maxY = t->Y[k];
      }
      /* pixels whose centres px*256+128 lie in [min,max] (arithmetic shift = floor) */
      int32_t px0 = (minX - 128 + 255) >> 8, px1 = (maxX - 128) >> 8;
      int32_t py0 = (minY - 128 + 255) >> 8, py1 = (maxY - 128) >> 8;
      if (px0 < 0) px0 = 0;
      if (py0 < y0) py0 = y0;
      if (px1 > width - 1) px1 = width - 1;
      if (py1 > y1 - 1) py1 = y1 - 1;
      for (int32_t py = py0; py <= py1; ++py) {
        for (int32_t px = px0; px <= px1; ++px) {
          int32_t Xc = px * SUBPIXEL_ONE + 128, Yc = py * SUBPIXEL_ONE + 128;
          if (!covers(t, Xc, Yc)) continue;
          ++st.n_fragments;
          float z = tri_depth(t, Xc, Yc);
          size_t o = (size_t)py * width + px;
          if (z >= out_depth[o]) {
            out_depth[o] = z;
            key[o] = ((prim << 3) | (uint32_t)s) + 1u;
            if (flags & BBO_FLAG_FORWARD_SHADE) shade_pixel(pl, t, vary, mat, px, py, out_rgba + 4 * o);
          }
        }
      }
    }
  }

  /* ---- pass 2: shade the winning fragment of every covered pixel ---- */
  uint32_t cached_prim = BBO_NO_PRIM;
  float clip[3][4], vary[3][NVARY];
  const void *mat = NULL;
  raster_tri tris[MAX_SUBTRIS];
  int n_tris = 0;
  for (int32_t py = y0; py < y1; ++py) {
    for (int32_t px = 0; px < width; ++px) {
      size_t o = (size_t)py * width + px;
      uint32_t k = key[o];
      if (out_prim) out_prim[o] = k ? (k - 1u) >> 3 : BBO_NO_PRIM;
      if (!k && pl->program == 2) {
        /* brdf.frag runs on every pixel of its full-screen triangle (src/main.cpp:101-104): the cleared texel too */
        static const float cleared[16] = {0};
        brdf_pixel(pl->literal, pl->fu, pl->vu, cleared, out_rgba + 4 * o);
      }
      if (!k) continue;
      ++st.n_shaded;
      if (flags & BBO_FLAG_FORWARD_SHADE) continue;
      uint32_t prim = (k - 1u) >> 3, sub = (k - 1u) & 7u;
      if (prim != cached_prim) {
        int wc;
        pl->fetch(pl->ctx, prim, clip, vary, &mat);
        n_tris = build_prim(clip, &whole, tris, &wc);
        cached_prim = prim;
      }
      if ((int)sub >= n_tris) { free(key); if (own_depth) free(out_depth); return -4; }
      shade_pixel(pl, &tris[sub], vary, mat, px, py, out_rgba + 4 * o);
    }
  }
  free(key);
  if (own_depth) free(out_depth);
  if (stats) *stats = st;
  return 0;
}

static int render_pbr(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const bbo_draw *draws, uint32_t n_draws,
                      int32_t width, int32_t height, int32_t y0, int32_t y1, uint32_t flags, float *out_rgba,
                      float *out_gbuffer, uint32_t *out_prim, float *out_depth, bbo_stats *stats) {
  if (!frame || !view || (!draws && n_draws)) return -1;
  draw_info *di = (draw_info *)calloc(n_draws ? n_draws : 1, sizeof(draw_info));
  if (!di) return -3;
  uint64_t total = 0;
  for (uint32_t d = 0; d < n_draws; ++d) {
    uint32_t n = draws[d].indices ? draws[d].n_indices : draws[d].n_vertices;
    di[d].draw = &draws[d];
    di[d].first_prim = (uint32_t)total;
    di[d].tris_per_instance = n / 3;
    total += (uint64_t)(n / 3) * draws[d].n_instances;
    if (draws[d].indices)
      for (uint32_t i = 0; i < n / 3 * 3; ++i)
        if (draws[d].indices[i] >= draws[d].n_vertices) { free(di); return -5; }
  }
  if (total >= (1u << 29)) { free(di); return -2; }
  /* drop draws without triangles so pbr_fetch's search stays simple */
  uint32_t m = 0;
  for (uint32_t d = 0; d < n_draws; ++d)
    if (di[d].tris_per_instance && draws[d].n_instances) di[m++] = di[d];
  pbr_ctx ctx;
  ctx.fu = frame; ctx.vu = view; ctx.draws = di; ctx.n_draws = m;
  ctx.deferred = (flags & BBO_FLAG_DEFERRED) != 0;
  bbo_proj_view(view, &ctx.pv);
  pipeline pl = {ctx.deferred ? 2 : 0, &ctx, pbr_fetch, frame, view, ctx.deferred ? out_gbuffer : NULL, width,
                 (flags & BBO_FLAG_LITERAL) != 0, (flags & BBO_FLAG_OUTPUT_UV) != 0};
  int rc = render_core(&pl, (uint32_t)total, width, height, y0, y1, flags, out_rgba, out_prim, out_depth, stats);
  free(di);
  return rc;
}

int bbo_render(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const bbo_draw *draws, uint32_t n_draws,
               int32_t width, int32_t height, int32_t y0, int32_t y1, uint32_t flags, float *out_rgba,
               uint32_t *out_prim, float *out_depth, bbo_stats *stats) {
  return render_pbr(frame, view, draws, n_draws, width, height, y0, y1, flags, out_rgba, NULL, out_prim, out_depth, stats);
}

/* ------------------------------------------------------------------------------------------ */
/* the same forward frame on all cores ("std::thread over tile rows on all cores") */
/* ------------------------------------------------------------------------------------------ */
/* bbo_render called once per band repeats the vertex stage and the primitive setup of EVERY primitive in every call (22 ms of a
 * 4K band's 30-80 ms): fine for a checker, a poor all-cores baseline.  Here the primitives are set up ONCE, in parallel
 * (chunks of 1024 in API order), and the bands then walk the surviving sub-triangles of all chunks in that order -- so every
 * pixel sees the same fragments in the same order, through the same covers / tri_depth / shade_pixel: the frame is bit for bit
 * bbo_render's (tests/test_oracle_golden_frames.py). */
#define PAR_CHUNK 1024u
typedef struct {
  raster_tri t;
  uint32_t key;        /* ((prim << 3) | sub) + 1 */
  int32_t py0, py1;    /* pixel rows the bounding box holds a centre of */
} par_tri;
typedef struct {
  const pipeline *pl;
  uint32_t n_prims, n_chunks;
  int32_t width, height, band_rows, n_bands;
  float *out_rgba;
  par_tri **chunk_tris;
  uint32_t *chunk_n;
  atomic_uint next_chunk, next_band;
  atomic_ullong n_raster_tris, n_clipped, n_fragments, n_shaded;
  atomic_int error;
} par_job;

static void par_setup_chunk(par_job *j, uint32_t c) {
  const viewport whole = {0.5f * (float)j->width, 0.5f * (float)j->height, 0.5f * (float)j->width, 0.5f * (float)j->height};
  uint32_t p0 = c * PAR_CHUNK, p1 = p0 + PAR_CHUNK < j->n_prims ? p0 + PAR_CHUNK : j->n_prims;
  par_tri *out = (par_tri *)malloc(sizeof(par_tri) * (size_t)(p1 - p0) * MAX_SUBTRIS);
  if (!out) { atomic_store(&j->error, -3); return; }
  uint32_t n_out = 0;
  unsigned long long clipped = 0;
  for (uint32_t prim = p0; prim < p1; ++prim) {
    float clip[3][4], vary[3][NVARY];
    const void *mat;
    j->pl->fetch(j->pl->ctx, prim, clip, vary, &mat);
    raster_tri tris[MAX_SUBTRIS];
    int was_clipped;
    int n = build_prim(clip, &whole, tris, &was_clipped);
    clipped += (unsigned long long)was_clipped;
    for (int s = 0; s < n; ++s) {
      const raster_tri *t = &tris[s];
      if (t->clipped < 0) continue;
      int32_t minY = t->Y[0], maxY = t->Y[0];
      for (int k = 1; k < 3; ++k) {
        if (t->Y[k] < minY) minY = t->Y[k];
        if (t->Y[k] > maxY) maxY = t->Y[k];
      }
      par_tri *o = &out[n_out++];
      o->t = *t;
      o->key = ((prim << 3) | (uint32_t)s) + 1u;
      o->py0 = (minY - 128 + 255) >> 8;
      o->py1 = (maxY - 128) >> 8;
    }
  }
  j->chunk_tris[c] = out;
  j->chunk_n[c] = n_out;
  atomic_fetch_add(&j->n_raster_tris, n_out);
  atomic_fetch_add(&j->n_clipped, clipped);
}

static void par_render_band(par_job *j, int32_t b, uint32_t *key, float *depth) {
  const int32_t W = j->width, y0 = b * j->band_rows, y1 = y0 + j->band_rows < j->height ? y0 + j->band_rows : j->height;
  const size_t rows = (size_t)(y1 - y0);
  memset(key, 0, sizeof(uint32_t) * rows * (size_t)W);
  memset(depth, 0, sizeof(float) * rows * (size_t)W);
  memset(j->out_rgba + 4 * (size_t)y0 * W, 0, sizeof(float) * 4 * rows * (size_t)W);
  unsigned long long fragments = 0, shaded = 0;
  for (uint32_t c = 0; c < j->n_chunks; ++c) {
    const par_tri *pt = j->chunk_tris[c];
    for (uint32_t i = 0; i < j->chunk_n[c]; ++i) {
      if (pt[i].py1 < y0 || pt[i].py0 >= y1) continue;
      const raster_tri *t = &pt[i].t;
      int32_t minX = t->X[0], maxX = t->X[0];
      for (int k = 1; k < 3; ++k) {
        if (t->X[k] < minX) minX = t->X[k];
        if (t->X[k] > maxX) maxX = t->X[k];
      }
      int32_t px0 = (minX - 128 + 255) >> 8, px1 = (maxX - 128) >> 8;
      int32_t py0 = pt[i].py0 < y0 ? y0 : pt[i].py0, py1 = pt[i].py1 > y1 - 1 ? y1 - 1 : pt[i].py1;
      if (px0 < 0) px0 = 0;
      if (px1 > W - 1) px1 = W - 1;
      for (int32_t py = py0; py <= py1; ++py)
        for (int32_t px = px0; px <= px1; ++px) {
          int32_t Xc = px * SUBPIXEL_ONE + 128, Yc = py * SUBPIXEL_ONE + 128;
          if (!covers(t, Xc, Yc)) continue;
          ++fragments;
          float z = tri_depth(t, Xc, Yc);
          size_t o = (size_t)(py - y0) * W + px;
          if (z >= depth[o]) {
            depth[o] = z;
            key[o] = pt[i].key;
          }
        }
    }
  }
  /* shade the winning fragment of every covered pixel (the primitive is fetched and set up again: render_core's pass 2) */
  const viewport whole = {0.5f * (float)W, 0.5f * (float)j->height, 0.5f * (float)W, 0.5f * (float)j->height};
  uint32_t cached_prim = BBO_NO_PRIM;
  float clip[3][4], vary[3][NVARY];
  const void *mat = NULL;
  raster_tri tris[MAX_SUBTRIS];
  int n_tris = 0;
  for (int32_t py = y0; py < y1; ++py)
    for (int32_t px = 0; px < W; ++px) {
      uint32_t k = key[(size_t)(py - y0) * W + px];
      if (!k) continue;
      ++shaded;
      uint32_t prim = (k - 1u) >> 3, sub = (k - 1u) & 7u;
      if (prim != cached_prim) {
        int wc;
        j->pl->fetch(j->pl->ctx, prim, clip, vary, &mat);
        n_tris = build_prim(clip, &whole, tris, &wc);
        cached_prim = prim;
      }
      if ((int)sub >= n_tris) { atomic_store(&j->error, -4); return; }
      shade_pixel(j->pl, &tris[sub], vary, mat, px, py, j->out_rgba + 4 * ((size_t)py * W + px));
    }
  atomic_fetch_add(&j->n_fragments, fragments);
  atomic_fetch_add(&j->n_shaded, shaded);
}

static void *par_setup_worker(void *arg) {
  par_job *j = (par_job *)arg;
  for (;;) {
    uint32_t c = atomic_fetch_add(&j->next_chunk, 1u);
    if (c >= j->n_chunks || atomic_load(&j->error)) break;
    par_setup_chunk(j, c);
  }
  return NULL;
}

static void *par_band_worker(void *arg) {
  par_job *j = (par_job *)arg;
  uint32_t *key = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)j->band_rows * (size_t)j->width);
  float *depth = (float *)malloc(sizeof(float) * (size_t)j->band_rows * (size_t)j->width);
  if (!key || !depth) atomic_store(&j->error, -3);
  else
    for (;;) {
      uint32_t b = atomic_fetch_add(&j->next_band, 1u);
      if ((int32_t)b >= j->n_bands || atomic_load(&j->error)) break;
      par_render_band(j, (int32_t)b, key, depth);
    }
  free(key);
  free(depth);
  return NULL;
}

static int par_run(par_job *j, void *(*fn)(void *), int n_threads) {
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  if (!th) return -3;
  int started = 0;
  for (; started < n_threads; ++started)
    if (pthread_create(&th[started], NULL, fn, j) != 0) break;
  if (started == 0) fn(j); /* no thread could be started: this one does the work */
  for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
  free(th);
  return atomic_load(&j->error);
}

int bbo_render_parallel(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const bbo_draw *draws, uint32_t n_draws,
                        int32_t width, int32_t height, uint32_t flags, int32_t n_threads, int32_t band_rows, float *out_rgba,
                        bbo_stats *stats) {
  if (!frame || !view || (!draws && n_draws) || width <= 0 || height <= 0 || !out_rgba) return -1;
  if (flags & (BBO_FLAG_DEFERRED | BBO_FLAG_FORWARD_SHADE | BBO_FLAG_OUTPUT_UV)) return -1; /* the forward frame only */
  if (n_threads < 1) n_threads = 1;
  if (band_rows < 1) band_rows = 8;
  draw_info *di = (draw_info *)calloc(n_draws ? n_draws : 1, sizeof(draw_info));
  if (!di) return -3;
  uint64_t total = 0;
  for (uint32_t d = 0; d < n_draws; ++d) {
    uint32_t n = draws[d].indices ? draws[d].n_indices : draws[d].n_vertices;
    di[d].draw = &draws[d];
    di[d].first_prim = (uint32_t)total;
    di[d].tris_per_instance = n / 3;
    total += (uint64_t)(n / 3) * draws[d].n_instances;
    if (draws[d].indices)
      for (uint32_t i = 0; i < n / 3 * 3; ++i)
        if (draws[d].indices[i] >= draws[d].n_vertices) { free(di); return -5; }
  }
  if (total >= (1u << 29)) { free(di); return -2; }
  uint32_t m = 0;
  for (uint32_t d = 0; d < n_draws; ++d)
    if (di[d].tris_per_instance && draws[d].n_instances) di[m++] = di[d];
  pbr_ctx ctx;
  ctx.fu = frame; ctx.vu = view; ctx.draws = di; ctx.n_draws = m; ctx.deferred = 0;
  bbo_proj_view(view, &ctx.pv);
  pipeline pl = {0, &ctx, pbr_fetch, frame, view, NULL, width, (flags & BBO_FLAG_LITERAL) != 0, 0};
  par_job j;
  memset(&j, 0, sizeof j);
  j.pl = &pl;
  j.n_prims = (uint32_t)total;
  j.n_chunks = (uint32_t)((total + PAR_CHUNK - 1) / PAR_CHUNK);
  j.width = width; j.height = height; j.band_rows = band_rows; j.n_bands = (height + band_rows - 1) / band_rows;
  j.out_rgba = out_rgba;
  j.chunk_tris = (par_tri **)calloc(j.n_chunks ? j.n_chunks : 1, sizeof(par_tri *));
  j.chunk_n = (uint32_t *)calloc(j.n_chunks ? j.n_chunks : 1, sizeof(uint32_t));
  int rc = (j.chunk_tris && j.chunk_n) ? 0 : -3;
  if (!rc) rc = par_run(&j, par_setup_worker, n_threads);
  if (!rc) rc = par_run(&j, par_band_worker, n_threads);
  if (j.chunk_tris)
    for (uint32_t c = 0; c < j.n_chunks; ++c) free(j.chunk_tris[c]);
  free(j.chunk_tris);
  free(j.chunk_n);
  free(di);
  if (!rc && stats) {
    memset(stats, 0, sizeof *stats);
    stats->n_prims = total;
    stats->n_raster_tris = atomic_load(&j.n_raster_tris);
    stats->n_clipped_prims = atomic_load(&j.n_clipped);
    stats->n_fragments = atomic_load(&j.n_fragments);
    stats->n_shaded = atomic_load(&j.n_shaded);
  }
  return rc;
}

int bbo_render_deferred(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const bbo_draw *draws,
                        uint32_t n_draws, int32_t width, int32_t height, int32_t y0, int32_t y1, float *out_rgba,
                        float *out_gbuffer, uint32_t *out_prim, float *out_depth, bbo_stats *stats) {
  return render_pbr(frame, view, draws, n_draws, width, height, y0, y1, BBO_FLAG_DEFERRED, out_rgba, out_gbuffer, out_prim,
                    out_depth, stats);
}

int bbo_render_deferred_flags(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, const bbo_draw *draws,
                              uint32_t n_draws, int32_t width, int32_t height, int32_t y0, int32_t y1, uint32_t flags,
                              float *out_rgba, float *out_gbuffer, uint32_t *out_prim, float *out_depth, bbo_stats *stats) {
  return render_pbr(frame, view, draws, n_draws, width, height, y0, y1, (flags & BBO_FLAG_LITERAL) | BBO_FLAG_DEFERRED, out_rgba,
                    out_gbuffer, out_prim, out_depth, stats);
}

/* ------------------------------------------------------------------------------------------ */
/* overlay subpass: light markers and the corner gizmo, drawn into the      */
/* presented image after tone mapping, depth-tested against the scene's depth                    */
/*   recordCommand src/main.cpp:128-171; light.vert:10-16, light.frag; marker mesh               */
/*   generateUVSphereMesh(0.1, 16, 16) src/main.cpp:953-957, src/render.cpp:1774-1833;           */
/*   pipelines src/main.cpp:746-784 (gizmo: 100 x 100 viewport + scissor in the top-right        */
/*   corner), :825-861 (markers: whole target); both cull BACK, depth test + write,              */
/*   GREATER_OR_EQUAL; depth of the gizmo rectangle cleared to 0 in between (:150-160)           */
/* ------------------------------------------------------------------------------------------ */

#define BB_PI32 3.141592f              /* src/vector_math.h:6 */
#define BB_HALF_PI32 (BB_PI32 * 0.5f) /* src/vector_math.h:7 */
#define BB_TWO_PI32 (BB_PI32 * 2.f)   /* :8 */

/* Positions and indices of generateUVSphereMesh (the markers only keep Pos, src/main.cpp:956): (h+1)*(v+1) vertices,
 * 6*h*(v-1) indices.  sphericalToCartesian: src/vector_math.cpp:284-292. */
void bbo_uv_sphere(float radius, int32_t hdiv, int32_t vdiv, float *out_pos3, uint32_t *out_indices, uint32_t *out_n_vertices,
                   uint32_t *out_n_indices) {
  uint32_t nv = 0, ni = 0;
  for (int v = 0; v <= vdiv; ++v) {
    float theta = -BB_HALF_PI32 + BB_PI32 * ((float)v / (float)vdiv);
    for (int h = 0; h <= hdiv; ++h) {
      float phi = BB_TWO_PI32 * ((float)h / (float)hdiv);
      float cos_theta = cosf(theta);
      if (out_pos3) {
        out_pos3[3 * nv + 0] = radius * cos_theta * cosf(phi);
        out_pos3[3 * nv + 1] = radius * sinf(theta);
        out_pos3[3 * nv + 2] = radius * cos_theta * sinf(phi);
      }
      ++nv;
    }
  }
  for (int v = 0; v < vdiv; ++v)
    for (int h = 0; h < hdiv; ++h) {
      uint32_t base = (uint32_t)((hdiv + 1) * v + h);
      if (v < vdiv - 1) {
        if (out_indices) { out_indices[ni] = base; out_indices[ni + 1] = base + hdiv + 1; out_indices[ni + 2] = base + hdiv + 2; }
        ni += 3;
      }
      if (v > 0) {
        if (out_indices) { out_indices[ni] = base + hdiv + 2; out_indices[ni + 1] = base + 1; out_indices[ni + 2] = base; }
        ni += 3;
      }
    }
  if (out_n_vertices) *out_n_vertices = nv;
  if (out_n_indices) *out_n_indices = ni;
}

/* the gizmo's own matrices, gizmo.vert:13-24 */
static void gizmo_matrices(const bbo_view_uniforms *view, bbo_mat4 *gview, bbo_mat4 *gpv) {
  const bbo_mat4 *uv = &view->view;
  v3 right = v3_make(uv->M[0][0], uv->M[1][0], uv->M[2][0]);
  v3 up = v3_make(uv->M[0][1], uv->M[1][1], uv->M[2][1]);
  v3 look = v3_make(uv->M[0][2], uv->M[1][2], uv->M[2][2]);
  v3 view_pos = scale3(look, -27.0f);
  *gview = *uv;
  gview->M[3][0] = -dot3(view_pos, right);
  gview->M[3][1] = -dot3(view_pos, up);
  gview->M[3][2] = -dot3(view_pos, look);
  bbo_view_uniforms gv = *view;
  gv.view = *gview;
  float d = 1.0f / tanf(0.261799f);
  gv.proj.M[0][0] = d;
  gv.proj.M[1][1] = -d;
  bbo_proj_view(&gv, gpv);
}

/* one overlay triangle: clip, set up, rasterise inside the scissor rectangle with depth test GREATER_OR_EQUAL and
 * depth write; the colour is flat (markers) or gizmo.frag on the interpolated colour + normal */
static void overlay_triangle(const float clip[3][4], const float vary[3][NVARY], int n_vary, const viewport *vp, const int32_t *sc,
                             int32_t width, float *depth, uint8_t *rgba8, const float *thr, bbo_stats *st) {
  raster_tri tris[MAX_SUBTRIS];
  int was_clipped;
  int n = build_prim(clip, vp, tris, &was_clipped);
  st->n_clipped_prims += (uint64_t)was_clipped;
  for (int s = 0; s < n; ++s) {
    const raster_tri *t = &tris[s];
    if (t->clipped < 0) continue;
    ++st->n_raster_tris;
    int32_t minX = t->X[0], maxX = t->X[0], minY = t->Y[0], maxY = t->Y[0];
    for (int k = 1; k < 3; ++k) {
      if (t->X[k] < minX) minX = t->X[k];
      if (t->X[k] > maxX) maxX = t->X[k];
      if (t->Y[k] < minY) minY = t->Y[k];
      if (t->Y[k] > maxY) maxY = t->Y[k];
    }
    int32_t px0 = (minX - 128 + 255) >> 8, px1 = (maxX - 128) >> 8;
    int32_t py0 = (minY - 128 + 255) >> 8, py1 = (maxY - 128) >> 8;
    if (px0 < sc[0]) px0 = sc[0];
    if (py0 < sc[1]) py0 = sc[1];
    if (px1 > sc[2] - 1) px1 = sc[2] - 1;
    if (py1 > sc[3] - 1) py1 = sc[3] - 1;
    for (int32_t py = py0; py <= py1; ++py)
      for (int32_t px = px0; px <= px1; ++px) {
        int32_t Xc = px * SUBPIXEL_ONE + 128, Yc = py * SUBPIXEL_ONE + 128;
        if (!covers(t, Xc, Yc)) continue;
        ++st->n_fragments;
        float z = tri_depth(t, Xc, Yc);
        size_t o = (size_t)py * width + px;
        if (!(z >= depth[o])) continue;
        depth[o] = z;
        float beta[3], attr[NVARY], col[4];
        tri_bary(t, Xc, Yc, beta);
        interpolate(beta, vary, n_vary, attr);
        if (n_vary == 3) { col[0] = attr[0]; col[1] = attr[1]; col[2] = attr[2]; } /* light.frag: outColor = vec4(vColor, 1) */
        else gizmo_shade(attr, col);
        for (int c = 0; c < 3; ++c) rgba8[4 * o + c] = srgb8(col[c], thr);
        rgba8[4 * o + 3] = 255;
      }
  }
}

int bbo_overlay(const bbo_frame_uniforms *frame, const bbo_view_uniforms *view, int32_t width, int32_t height,
                const float *scene_depth, uint8_t *rgba8, const bbo_gizmo_vertex *gizmo_vertices, uint32_t n_gizmo_vertices,
                const uint32_t *gizmo_indices, uint32_t n_gizmo_indices, int32_t gizmo_extent, bbo_stats *stats) {
  if (!frame || !view || !scene_depth || !rgba8 || width <= 0 || height <= 0) return -1;
  size_t npx = (size_t)width * (size_t)height;
  float *depth = (float *)malloc(npx * sizeof(float));
  if (!depth) return -3;
  memcpy(depth, scene_depth, npx * sizeof(float));
  float thr[256];
  bbo_srgb_thresholds(thr);
  thr[255] = INFINITY;
  bbo_stats st;
  memset(&st, 0, sizeof st);
  const viewport whole = {0.5f * (float)width, 0.5f * (float)height, 0.5f * (float)width, 0.5f * (float)height};
  const int32_t sc_whole[4] = {0, 0, width, height};

  /* ---- light markers: instance i sits at uLights[i].pos and has its colour (light.vert:11-15) ---- */
  {
    enum { HD = 16, VD = 16 };
    float pos[(HD + 1) * (VD + 1) * 3];
    uint32_t idx[6 * HD * (VD - 1)], nv, ni;
    bbo_uv_sphere(0.1f, HD, VD, pos, idx, &nv, &ni);
    bbo_mat4 pv;
    bbo_proj_view(view, &pv); /* uProjMat * uViewMat */
    int n_lights = frame->num_lights;
    if (n_lights < 0) n_lights = 0;
    if (n_lights > BBO_MAX_LIGHTS) n_lights = BBO_MAX_LIGHTS;
    for (int li = 0; li < n_lights; ++li) {
      const bbo_light *light = &frame->lights[li];
      /* (P*V) * modelMat, modelMat = identity with column 3 = (pos, 1): columns 0..2 are those of P*V */
      bbo_mat4 pvm = pv;
      v4 t = {light->pos[0], light->pos[1], light->pos[2], 1.0f};
      v4 c3 = mat4_mul_v4(&pv, t);
      pvm.M[3][0] = c3.x; pvm.M[3][1] = c3.y; pvm.M[3][2] = c3.z; pvm.M[3][3] = c3.w;
      for (uint32_t k = 0; k + 2 < ni; k += 3) {
        float clip[3][4], vary[3][NVARY];
        for (int j = 0; j < 3; ++j) {
          const float *p = pos + 3 * idx[k + j];
          v4 pc = mat4_mul_v4(&pvm, (v4){p[0], p[1], p[2], 1.0f});
          clip[j][0] = pc.x; clip[j][1] = pc.y; clip[j][2] = pc.z; clip[j][3] = pc.w;
          memset(vary[j], 0, sizeof vary[j]);
          vary[j][0] = light->color[0]; vary[j][1] = light->color[1]; vary[j][2] = light->color[2];
        }
        ++st.n_prims;
        overlay_triangle(clip, vary, 3, &whole, sc_whole, width, depth, rgba8, thr, &st);
      }
    }
  }

  /* ---- gizmo: depth of its rectangle cleared, own viewport and scissor (src/main.cpp:150-160, 760-772) ---- */
  if (gizmo_vertices && n_gizmo_vertices && gizmo_extent > 0) {
    int32_t x0 = width - gizmo_extent, y0 = 0;
    int32_t sc[4] = {x0 < 0 ? 0 : x0, y0, width, gizmo_extent < height ? gizmo_extent : height};
    for (int32_t y = sc[1]; y < sc[3]; ++y)
      for (int32_t x = sc[0]; x < sc[2]; ++x) depth[(size_t)y * width + x] = 0.0f;
    const float half = 0.5f * (float)gizmo_extent;
    const viewport vp = {half, half, (float)x0 + half, (float)y0 + half};
    bbo_mat4 gview, gpv;
    gizmo_matrices(view, &gview, &gpv);
    uint32_t n = gizmo_indices ? n_gizmo_indices : n_gizmo_vertices;
    for (uint32_t k = 0; k + 2 < n; k += 3) {
      float clip[3][4], vary[3][NVARY];
      for (int j = 0; j < 3; ++j) {
        uint32_t vi = gizmo_indices ? gizmo_indices[k + j] : k + j;
        if (vi >= n_gizmo_vertices) { free(depth); return -5; }
        const bbo_gizmo_vertex *gv = &gizmo_vertices[vi];
        v4 pc = mat4_mul_v4(&gpv, (v4){gv->pos[0], gv->pos[1], gv->pos[2], 1.0f});
        clip[j][0] = pc.x; clip[j][1] = pc.y; clip[j][2] = pc.z; clip[j][3] = pc.w;
        memset(vary[j], 0, sizeof vary[j]);
        vary[j][0] = gv->color[0]; vary[j][1] = gv->color[1]; vary[j][2] = gv->color[2];
        v3 nn = v3_ld(gv->normal);
        vary[j][3] = fmaf(gview.M[2][0], nn.z, fmaf(gview.M[1][0], nn.y, gview.M[0][0] * nn.x));
        vary[j][4] = fmaf(gview.M[2][1], nn.z, fmaf(gview.M[1][1], nn.y, gview.M[0][1] * nn.x));
        vary[j][5] = fmaf(gview.M[2][2], nn.z, fmaf(gview.M[1][2], nn.y, gview.M[0][2] * nn.x));
      }
      ++st.n_prims;
      overlay_triangle(clip, vary, 6, &vp, sc, width, depth, rgba8, thr, &st);
    }
  }
  free(depth);
  if (stats) *stats = st;
  return 0;
}

int bbo_render_gizmo(const bbo_view_uniforms *view, const bbo_gizmo_vertex *vertices, uint32_t n_vertices,
                     const uint32_t *indices, uint32_t n_indices, int32_t width, int32_t height, float *out_rgba,
                     uint32_t *out_prim, float *out_depth, bbo_stats *stats) {
  if (!view || !vertices) return -1;
  uint32_t n = indices ? n_indices : n_vertices;
  if (indices)
    for (uint32_t i = 0; i < n / 3 * 3; ++i)
      if (indices[i] >= n_vertices) return -5;
  gizmo_ctx ctx;
  ctx.vu = view; ctx.verts = vertices; ctx.indices = indices;
  /* gizmo.vert:13-24 */
  const bbo_mat4 *uv = &view->view;
  v3 right = v3_make(uv->M[0][0], uv->M[1][0], uv->M[2][0]);
  v3 up = v3_make(uv->M[0][1], uv->M[1][1], uv->M[2][1]);
  v3 look = v3_make(uv->M[0][2], uv->M[1][2], uv->M[2][2]);
  v3 view_pos = scale3(look, -27.0f);
  ctx.view = *uv;
  ctx.view.M[3][0] = -dot3(view_pos, right);
  ctx.view.M[3][1] = -dot3(view_pos, up);
  ctx.view.M[3][2] = -dot3(view_pos, look);
  bbo_view_uniforms gv = *view;
  gv.view = ctx.view;
  float d = 1.0f / tanf(0.261799f);
  gv.proj.M[0][0] = d;
  gv.proj.M[1][1] = -d;
  bbo_proj_view(&gv, &ctx.pv);
  pipeline pl = {1, &ctx, gizmo_fetch, NULL, view, NULL, width, 0, 0};
  return render_core(&pl, n / 3, width, height, 0, height, 0, out_rgba, out_prim, out_depth, stats);
}

/* ------------------------------------------------------------------------------------------ */
/* vector_math.cpp / camera.cpp restatement (src/vector_math.cpp:84-282, src/camera.cpp:5-20)  */
/* plain mul/add order of the reference, no fused ops                                          */
/* ------------------------------------------------------------------------------------------ */

static inline float deg_to_rad(float d) { return d * BB_PI32 / 180.f; }

void bbo_mat4_identity(bbo_mat4 *out) {
  memset(out, 0, sizeof *out);
  out->M[0][0] = out->M[1][1] = out->M[2][2] = out->M[3][3] = 1.f;
}

static inline float dot4_plain(const float *a, const float *b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; /* src/vector_math.cpp:71-73 */
}

void bbo_mat4_mul(const bbo_mat4 *a, const bbo_mat4 *b, bbo_mat4 *out) {
  bbo_mat4 r; /* src/vector_math.cpp:262-272: result.M[j][i] = dot(a.row(i), b.column(j)) */
  for (int i = 0; i < 4; ++i) {
    float row[4] = {a->M[0][i], a->M[1][i], a->M[2][i], a->M[3][i]};
    for (int j = 0; j < 4; ++j) r.M[j][i] = dot4_plain(row, b->M[j]);
  }
  *out = r;
}

void bbo_mat4_transpose(const bbo_mat4 *a, bbo_mat4 *out) {
  bbo_mat4 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) r.M[j][i] = a->M[i][j];
  *out = r;
}

static float det3(float m[3][3]) { /* src/vector_math.cpp:75-80 */
  return m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) +
         m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
}

static float cofactor(const bbo_mat4 *a, int row, int col) { /* src/vector_math.cpp:92-113 */
  float minor[3][3];
  int mr = 0, mc = 0;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      if (r != row && c != col) {
        minor[mc][mr++] = a->M[c][r];
        if (mr == 3) { mr = 0; ++mc; }
      }
  float sign = ((row + col) % 2) ? -1.f : 1.f;
  return det3(minor) * sign;
}

void bbo_mat4_inverse(const bbo_mat4 *a, bbo_mat4 *out) { /* src/vector_math.cpp:115-134 */
  bbo_mat4 adj;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) adj.M[c][r] = cofactor(a, r, c);
  float det = 0.f;
  for (int i = 0; i < 4; ++i) det += a->M[i][0] * adj.M[i][0];
  bbo_mat4 t;
  bbo_mat4_transpose(&adj, &t);
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) t.M[c][r] /= det;
  *out = t;
}

void bbo_mat4_translate(float x, float y, float z, bbo_mat4 *out) {
  bbo_mat4_identity(out);
  out->M[3][0] = x; out->M[3][1] = y; out->M[3][2] = z;
}

void bbo_mat4_scale(float x, float y, float z, bbo_mat4 *out) {
  bbo_mat4_identity(out);
  out->M[0][0] = x; out->M[1][1] = y; out->M[2][2] = z;
}

void bbo_mat4_rotate_x(float degrees, bbo_mat4 *out) { /* src/vector_math.cpp:187-199 */
  float r = deg_to_rad(degrees), cr = cosf(r), sr = sinf(r);
  bbo_mat4_identity(out);
  out->M[1][1] = cr; out->M[1][2] = sr; out->M[2][1] = -sr; out->M[2][2] = cr;
}

void bbo_mat4_rotate_y(float degrees, bbo_mat4 *out) { /* :201-213 */
  float r = deg_to_rad(degrees), cr = cosf(r), sr = sinf(r);
  bbo_mat4_identity(out);
  out->M[0][0] = cr; out->M[0][2] = sr; out->M[2][0] = -sr; out->M[2][2] = cr;
}

void bbo_mat4_rotate_z(float degrees, bbo_mat4 *out) { /* :215-227 */
  float r = deg_to_rad(degrees), cr = cosf(r), sr = sinf(r);
  bbo_mat4_identity(out);
  out->M[0][0] = cr; out->M[0][1] = sr; out->M[1][0] = -sr; out->M[1][1] = cr;
}

static inline float dot3_plain(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross_plain(const float *a, const float *b, float *o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline void normalize_plain(float *v) { /* Float3::normalize = *this / length() */
  float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  v[0] /= len; v[1] /= len; v[2] /= len;
}

void bbo_mat4_look_at(const float *eye, const float *target, const float *up_axis, bbo_mat4 *out) {
  float f[3] = {target[0] - eye[0], target[1] - eye[1], target[2] - eye[2]}; /* :231-245 */
  normalize_plain(f);
  float r[3], u[3];
  cross_plain(up_axis, f, r);
  normalize_plain(r);
  cross_plain(f, r, u);
  normalize_plain(u);
  memset(out, 0, sizeof *out);
  out->M[0][0] = r[0]; out->M[0][1] = u[0]; out->M[0][2] = f[0];
  out->M[1][0] = r[1]; out->M[1][1] = u[1]; out->M[1][2] = f[1];
  out->M[2][0] = r[2]; out->M[2][1] = u[2]; out->M[2][2] = f[2];
  out->M[3][0] = -dot3_plain(eye, r); out->M[3][1] = -dot3_plain(eye, u); out->M[3][2] = -dot3_plain(eye, f);
  out->M[3][3] = 1.f;
}

void bbo_mat4_perspective(float fov_degrees, float aspect, float near_z, float far_z, bbo_mat4 *out) {
  /* :247-260 -- the unqualified `tan` resolves to the C double function in the reference as compiled here
   * (g++, oracle/_ref): d = (float)(1.0 / tan((double)x)).  Differs from tanf by 1 ulp for some fovs;
   * pinned by tests/test_oracle_math.py against oracle/_ref on random fovs. */
  float d = (float)(1.0 / tan((double)(deg_to_rad(fov_degrees) * 0.5f)));
  float f_sub_n = far_z - near_z;
  memset(out, 0, sizeof *out);
  out->M[0][0] = d / aspect;
  out->M[1][1] = -d;
  out->M[2][2] = -near_z / f_sub_n;
  out->M[2][3] = 1.f;
  out->M[3][2] = near_z * far_z / f_sub_n;
}

void bbo_camera_look(float yaw, float pitch, float *out3) { /* src/camera.cpp:14-20 */
  float yr = deg_to_rad(yaw), pr = deg_to_rad(pitch), cp = cosf(pr);
  out3[0] = -sinf(yr) * cp; out3[1] = sinf(pr); out3[2] = cosf(yr) * cp;
}

void bbo_camera_view(const float *pos, float yaw, float pitch, bbo_mat4 *out) { /* src/camera.cpp:5-7 */
  float look[3], target[3], up[3] = {0.f, 1.f, 0.f};
  bbo_camera_look(yaw, pitch, look);
  for (int i = 0; i < 3; ++i) target[i] = pos[i] + look[i];
  bbo_mat4_look_at(pos, target, up, out);
}

uint32_t bbo_contract_revision(void) { return BBO_CONTRACT_REVISION; }

uint32_t bbo_sizeof(int what) {
  switch (what) {
  case 0: return (uint32_t)sizeof(bbo_vertex);
  case 1: return (uint32_t)sizeof(bbo_instance);
  case 2: return (uint32_t)sizeof(bbo_light);
  case 3: return (uint32_t)sizeof(bbo_frame_uniforms);
  case 4: return (uint32_t)sizeof(bbo_view_uniforms);
  case 5: return (uint32_t)sizeof(bbo_gizmo_vertex);
  default: return 0;
  }
}
