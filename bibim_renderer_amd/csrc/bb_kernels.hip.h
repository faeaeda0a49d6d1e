// bb_kernels.hip.h -- HIP kernels of the forward PBR path for gfx950 (MI355X / CDNA4).
//
//   k_geometry   forward_brdf.vert + clip/cull/viewport/snap + triangle setup + tile binning
//                (reference: src/shaders/forward_brdf.vert:24-37; state src/render.cpp:1069-1125)
//   k_tile       per screen tile: LDS-resident 64-bit visibility keys (depth | primitive) filled with
//                ds_max_u64, ballot/popcount compaction of covered pixels, then forward_brdf.frag +
//                brdf.glsl once per visible pixel (src/shaders/forward_brdf.frag:15-76, brdf.glsl:2-36)
//
// Arithmetic contract: every floating-point expression below has the same operand order and the same
// explicit fmaf() placement as the CPU oracle; the file is compiled with -ffp-contract=off, IEEE
// division and sqrt (hipcc default), denormals on.  Integer coverage is exact (24.8 fixed point,
// 64-bit edge functions, top-left rule).
#pragma once

#include <hip/hip_runtime.h>

#include "bb_types.h"

namespace bbr {

#define BB_DEV __device__ __forceinline__

constexpr float kGuardBand = 32.0f;
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;

struct f3 {
  float x, y, z;
};
struct f4 {
  float x, y, z, w;
};

BB_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
BB_DEV f3 ld3(const float *p) { return f3{p[0], p[1], p[2]}; }
BB_DEV float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
BB_DEV f3 add3(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
BB_DEV f3 sub3(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
BB_DEV f3 scale3(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
BB_DEV f3 neg3(f3 a) { return f3{-a.x, -a.y, -a.z}; }
BB_DEV f3 cross3(f3 a, f3 b) {
  return f3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
BB_DEV f3 normalize3(f3 a) { return scale3(a, 1.0f / sqrtf(dot3(a, a))); }
BB_DEV float max0(float a) { return a > 0.0f ? a : 0.0f; }

BB_DEV f4 mat4_mul(const Mat4 &m, f4 v) {
  f4 r;
  r.x = fmaf(m.M[3][0], v.w, fmaf(m.M[2][0], v.z, fmaf(m.M[1][0], v.y, m.M[0][0] * v.x)));
  r.y = fmaf(m.M[3][1], v.w, fmaf(m.M[2][1], v.z, fmaf(m.M[1][1], v.y, m.M[0][1] * v.x)));
  r.z = fmaf(m.M[3][2], v.w, fmaf(m.M[2][2], v.z, fmaf(m.M[1][2], v.y, m.M[0][2] * v.x)));
  r.w = fmaf(m.M[3][3], v.w, fmaf(m.M[2][3], v.z, fmaf(m.M[1][3], v.y, m.M[0][3] * v.x)));
  return r;
}

// ------------------------------------------------------------------------------------------------
// geometry: vertex stage, clip, setup, binning
// ------------------------------------------------------------------------------------------------

struct ClipVert {
  float c[4];
  float b[3];
};

BB_DEV float plane_dist(const float *c, int plane) {
  switch (plane) {
    case 0: return c[3] - c[2];  // near (reverse-Z: z <= w)
    case 1: return c[2];         // far
    case 2: return fmaf(kGuardBand, c[3], c[0]);
    case 3: return fmaf(kGuardBand, c[3], -c[0]);
    case 4: return fmaf(kGuardBand, c[3], c[1]);
    default: return fmaf(kGuardBand, c[3], -c[1]);
  }
}

BB_DEV void clip_lerp(const ClipVert &in, float din, const ClipVert &out, float dout, ClipVert &r) {
  float t = din / (din - dout);
  for (int k = 0; k < 4; ++k) r.c[k] = fmaf(t, out.c[k] - in.c[k], in.c[k]);
  for (int k = 0; k < 3; ++k) r.b[k] = fmaf(t, out.b[k] - in.b[k], in.b[k]);
}

constexpr int kMaxClipVerts = 12;

// Sutherland-Hodgman against near, far and the four guard-band planes.  Rare path (a handful of
// primitives per frame), so it is kept out of line and unoptimised.
__device__ __noinline__ int clip_polygon(ClipVert *poly, int n) {
  ClipVert tmp[kMaxClipVerts];
  for (int plane = 0; plane < 6; ++plane) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const ClipVert &a = poly[i];
      const ClipVert &b = poly[(i + 1) % n];
      float da = plane_dist(a.c, plane), db = plane_dist(b.c, plane);
      bool ina = da >= 0.0f, inb = db >= 0.0f;
      if (ina) tmp[m++] = a;
      if (ina != inb) {
        if (ina) clip_lerp(a, da, b, db, tmp[m]);
        else clip_lerp(b, db, a, da, tmp[m]);
        ++m;
      }
    }
    n = m;
    if (n < 3) return 0;
    for (int i = 0; i < n; ++i) poly[i] = tmp[i];
  }
  return n;
}

BB_DEV bool project_vertex(const float *c, float half_w, float half_h, int32_t &X, int32_t &Y, float &rw,
                           float &zndc) {
  float w = c[3];
  if (!(w > 0.0f)) return false;
  float r = 1.0f / w;
  float xs = fmaf(c[0] * r, half_w, half_w);
  float ys = fmaf(c[1] * r, half_h, half_h);
  if (!(fabsf(xs) <= 4194304.0f) || !(fabsf(ys) <= 4194304.0f)) return false;
  X = (int32_t)rintf(xs * 256.0f);
  Y = (int32_t)rintf(ys * 256.0f);
  rw = r;
  zndc = c[2] * r;
  return true;
}

// cull + plane setup (binary64, rounded once -- same expressions as the oracle's setup_tri)
BB_DEV bool setup_tri(RasterTri &t, float z0, float z1, float z2) {
  long long dx1 = (long long)t.X1 - t.X0, dy1 = (long long)t.Y1 - t.Y0;
  long long dx2 = (long long)t.X2 - t.X0, dy2 = (long long)t.Y2 - t.Y0;
  long long S = dx1 * dy2 - dx2 * dy1;
  if (S <= 0) return false;
  double rS = 1.0 / (double)S;
  t.l1dx = (float)((double)dy2 * rS);
  t.l1dy = (float)(-(double)dx2 * rS);
  t.l2dx = (float)(-(double)dy1 * rS);
  t.l2dy = (float)((double)dx1 * rS);
  double dz1 = (double)z1 - (double)z0, dz2 = (double)z2 - (double)z0;
  t.z0 = z0;
  t.dzdx = (float)((dz1 * (double)dy2 - dz2 * (double)dy1) * rS);
  t.dzdy = (float)((dz2 * (double)dx1 - dz1 * (double)dx2) * rS);
  return true;
}

template <int TILE_W, int TILE_H>
BB_DEV void bin_triangle(const RasterTri &t, uint32_t ref, const FrameParams &fp, Counters *ctr,
                         uint32_t *tile_count, uint32_t *bins, uint32_t *broad_list) {
  int32_t minX = min(t.X0, min(t.X1, t.X2)), maxX = max(t.X0, max(t.X1, t.X2));
  int32_t minY = min(t.Y0, min(t.Y1, t.Y2)), maxY = max(t.Y0, max(t.Y1, t.Y2));
  int32_t px0 = max((minX - 128 + 255) >> 8, 0), px1 = min((maxX - 128) >> 8, fp.width - 1);
  int32_t py0 = max((minY - 128 + 255) >> 8, 0), py1 = min((maxY - 128) >> 8, fp.height - 1);
  if (px0 > px1 || py0 > py1) return;  // no pixel centre inside the bounding box
  atomicAdd(&ctr->n_raster_tris, 1ull);
  int tx0 = px0 / TILE_W, tx1 = px1 / TILE_W, ty0 = py0 / TILE_H, ty1 = py1 / TILE_H;
  uint32_t ntiles = (uint32_t)(tx1 - tx0 + 1) * (uint32_t)(ty1 - ty0 + 1);
  if (ntiles > fp.broad_threshold) {
    uint32_t slot = atomicAdd(&ctr->n_broad, 1u);
    if (slot < fp.broad_cap) broad_list[slot] = ref;
    else atomicOr(&ctr->overflow, 2u);
    return;
  }
  for (int ty = ty0; ty <= ty1; ++ty) {
    if (fp.world > 1 && ((ty / fp.band_tiles) % fp.world) != fp.rank) continue;
    for (int tx = tx0; tx <= tx1; ++tx) {
      uint32_t tile = (uint32_t)ty * (uint32_t)fp.tiles_x + (uint32_t)tx;
      uint32_t slot = atomicAdd(&tile_count[tile], 1u);
      if (slot < fp.bin_cap) bins[(size_t)tile * fp.bin_cap + slot] = ref;
      else atomicOr(&ctr->overflow, 1u);
    }
  }
  atomicAdd(&ctr->n_bin_refs, (unsigned long long)ntiles);
}

// Rare path: the primitive crosses a clip plane.
template <int TILE_W, int TILE_H>
__device__ __noinline__ void clip_and_bin(const float (*clip)[4], uint32_t prim, const FrameParams &fp, RasterTri *tris,
                                          ClipSlot *clip_arena, Counters *ctr, uint32_t *tile_count, uint32_t *bins,
                                          uint32_t *broad_list, bool &any_valid) {
  any_valid = false;
  atomicAdd(&ctr->n_clipped_prims, 1ull);
  ClipVert poly[kMaxClipVerts];
  for (int i = 0; i < 3; ++i) {
    for (int k = 0; k < 4; ++k) poly[i].c[k] = clip[i][k];
    poly[i].b[0] = poly[i].b[1] = poly[i].b[2] = 0.0f;
    poly[i].b[i] = 1.0f;
  }
  int n = clip_polygon(poly, 3);
  if (n < 3) return;
  int32_t X[kMaxClipVerts], Y[kMaxClipVerts];
  float rw[kMaxClipVerts], z[kMaxClipVerts];
  for (int i = 0; i < n; ++i)
    if (!project_vertex(poly[i].c, fp.half_w, fp.half_h, X[i], Y[i], rw[i], z[i])) return;
  int n_slots = min(n - 2, kMaxSubTris);
  uint32_t base = atomicAdd(&ctr->n_clip_slots, (uint32_t)n_slots);
  if (base + n_slots > fp.clip_cap) {
    atomicOr(&ctr->overflow, 4u);
    return;
  }
  for (int i = 1; i <= n_slots; ++i) {
    ClipSlot s;
    const int id[3] = {0, i, i + 1};
    s.tri.X0 = X[id[0]]; s.tri.Y0 = Y[id[0]];
    s.tri.X1 = X[id[1]]; s.tri.Y1 = Y[id[1]];
    s.tri.X2 = X[id[2]]; s.tri.Y2 = Y[id[2]];
    s.tri.rw0 = rw[id[0]]; s.tri.rw1 = rw[id[1]]; s.tri.rw2 = rw[id[2]];
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c) s.bary[k][c] = poly[id[k]].b[c];
    s.pad[0] = s.pad[1] = 0;
    bool ok = setup_tri(s.tri, z[id[0]], z[id[1]], z[id[2]]);
    s.valid = ok ? 1u : 0u;
    clip_arena[base + i - 1] = s;
    if (ok) {
      any_valid = true;
      bin_triangle<TILE_W, TILE_H>(s.tri, (prim << 3) | (uint32_t)(i - 1), fp, ctr, tile_count, bins, broad_list);
    }
  }
  RasterTri head;
  head.X0 = kClippedSentinel;
  head.Y0 = (int32_t)base;
  head.X1 = n_slots;
  head.Y1 = head.X2 = head.Y2 = 0;
  head.z0 = head.dzdx = head.dzdy = head.l1dx = head.l1dy = head.l2dx = head.l2dy = 0.0f;
  head.rw0 = head.rw1 = head.rw2 = 0.0f;
  tris[prim] = head;
}

// One thread per primitive of one draw call.
template <int TILE_W, int TILE_H>
__global__ __launch_bounds__(256) void k_geometry(DrawDesc draw, Mat4 pv, FrameParams fp, RasterTri *__restrict__ tris,
                                                  PrimAttr *__restrict__ attrs, ClipSlot *__restrict__ clip_arena,
                                                  Counters *__restrict__ ctr, uint32_t *__restrict__ tile_count,
                                                  uint32_t *__restrict__ bins, uint32_t *__restrict__ broad_list) {
  uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t n_local = draw.n_instances * draw.tris_per_instance;
  if (local >= n_local) return;
  uint32_t inst = local / draw.tris_per_instance;
  uint32_t tri = local - inst * draw.tris_per_instance;
  uint32_t prim = draw.first_prim + local;

  const InstanceBlock &ib = draw.instances[inst];
  float clip[3][4];
  PrimAttr pa;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    uint32_t vi = draw.indices ? draw.indices[3 * tri + k] : 3 * tri + k;
    const Vertex &v = draw.vertices[vi];
    // forward_brdf.vert:25,27
    f4 pw = mat4_mul(ib.model, f4{v.pos[0], v.pos[1], v.pos[2], 1.0f});
    f4 c = mat4_mul(pv, pw);
    clip[k][0] = c.x; clip[k][1] = c.y; clip[k][2] = c.z; clip[k][3] = c.w;
    // :31-36  normalMat = transpose(mat3(aInvModel))
    f3 n = ld3(v.normal), t = ld3(v.tangent);
    const Mat4 &im = ib.inv_model;
    f3 N = normalize3(mk3(dot3(ld3(im.M[0]), n), dot3(ld3(im.M[1]), n), dot3(ld3(im.M[2]), n)));
    f3 T = normalize3(mk3(dot3(ld3(im.M[0]), t), dot3(ld3(im.M[1]), t), dot3(ld3(im.M[2]), t)));
    f3 B = cross3(N, T);
    float *o = pa.vary[k];
    o[0] = v.uv[0]; o[1] = v.uv[1];
    o[2] = pw.x; o[3] = pw.y; o[4] = pw.z;
    o[5] = N.x; o[6] = N.y; o[7] = N.z;
    o[8] = T.x; o[9] = T.y; o[10] = T.z;
    o[11] = B.x; o[12] = B.y; o[13] = B.z;
  }
  pa.material = draw.material;
  pa.pad = 0;

  // trivial reject against the true frustum (cannot change any pixel)
  {
    bool o_l = true, o_r = true, o_t = true, o_b = true, o_n = true, o_f = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float *c = clip[i];
      o_l &= (c[3] + c[0] < 0.0f); o_r &= (c[3] - c[0] < 0.0f);
      o_t &= (c[3] + c[1] < 0.0f); o_b &= (c[3] - c[1] < 0.0f);
      o_n &= (c[3] - c[2] < 0.0f); o_f &= (c[2] < 0.0f);
    }
    if (o_l | o_r | o_t | o_b | o_n | o_f) return;
  }
  bool all_in = true;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int p = 0; p < 6; ++p) all_in &= (plane_dist(clip[i], p) >= 0.0f);

  if (all_in) {
    RasterTri t;
    float z0, z1, z2;
    if (!project_vertex(clip[0], fp.half_w, fp.half_h, t.X0, t.Y0, t.rw0, z0)) return;
    if (!project_vertex(clip[1], fp.half_w, fp.half_h, t.X1, t.Y1, t.rw1, z1)) return;
    if (!project_vertex(clip[2], fp.half_w, fp.half_h, t.X2, t.Y2, t.rw2, z2)) return;
    if (!setup_tri(t, z0, z1, z2)) return;
    tris[prim] = t;
    attrs[prim] = pa;
    bin_triangle<TILE_W, TILE_H>(t, prim << 3, fp, ctr, tile_count, bins, broad_list);
  } else {
    bool any_valid;
    clip_and_bin<TILE_W, TILE_H>(clip, prim, fp, tris, clip_arena, ctr, tile_count, bins, broad_list, any_valid);
    if (any_valid) attrs[prim] = pa;
  }
}

// ------------------------------------------------------------------------------------------------
// texture sampling (SMP_LINEAR / REPEAT / one mip; src/render.cpp:1338-1371)
// ------------------------------------------------------------------------------------------------

BB_DEV int wrap_repeat(int i, int n) {
  int m = i % n;
  return m < 0 ? m + n : m;
}

struct BilinearTaps {
  uint32_t o00, o10, o01, o11;  // texel indices
  float fx, fy;
};

BB_DEV BilinearTaps bilinear_taps(float u, float v, int w, int h) {
  float x = fmaf(u, (float)w, -0.5f);
  float y = fmaf(v, (float)h, -0.5f);
  if (!(fabsf(x) < 1073741824.0f)) x = 0.0f;
  if (!(fabsf(y) < 1073741824.0f)) y = 0.0f;
  float xf = floorf(x), yf = floorf(y);
  BilinearTaps t;
  t.fx = x - xf;
  t.fy = y - yf;
  int ix = (int)xf, iy = (int)yf;
  int x0, x1, y0, y1;
  if (((w & (w - 1)) | (h & (h - 1))) == 0) {  // power-of-two sizes: wrap is a mask
    x0 = ix & (w - 1); x1 = (ix + 1) & (w - 1);
    y0 = iy & (h - 1); y1 = (iy + 1) & (h - 1);
  } else {
    x0 = wrap_repeat(ix, w); x1 = wrap_repeat(ix + 1, w);
    y0 = wrap_repeat(iy, h); y1 = wrap_repeat(iy + 1, h);
  }
  t.o00 = (uint32_t)y0 * (uint32_t)w + (uint32_t)x0;
  t.o10 = (uint32_t)y0 * (uint32_t)w + (uint32_t)x1;
  t.o01 = (uint32_t)y1 * (uint32_t)w + (uint32_t)x0;
  t.o11 = (uint32_t)y1 * (uint32_t)w + (uint32_t)x1;
  return t;
}

BB_DEV float filter_channel(uint32_t t00, uint32_t t10, uint32_t t01, uint32_t t11, int shift, float fx, float fy) {
  float a = (float)((t00 >> shift) & 0xFFu), b = (float)((t10 >> shift) & 0xFFu);
  float c = (float)((t01 >> shift) & 0xFFu), d = (float)((t11 >> shift) & 0xFFu);
  float top = fmaf(fx, b - a, a);
  float bot = fmaf(fx, d - c, c);
  return fmaf(fy, bot - top, top) * (1.0f / 255.0f);
}

// ------------------------------------------------------------------------------------------------
// brdf.glsl
// ------------------------------------------------------------------------------------------------

BB_DEV float distribution_ggx(float NdotH_raw, float roughness) {
  float a = roughness * roughness;
  float a2 = a * a;
  float NdotH = max0(NdotH_raw);
  float NdotH2 = NdotH * NdotH;
  float denom = fmaf(NdotH2, a2 - 1.0f, 1.0f);
  denom = (kPi * denom) * denom;
  return a2 / denom;
}

BB_DEV float geometry_schlick_ggx(float NdotX, float k) {
  float denom = fmaf(NdotX, 1.0f - k, k);
  return NdotX / denom;
}

BB_DEV float mixf(float a, float b, float t) { return fmaf(b, t, a * (1.0f - t)); }
BB_DEV float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

// ------------------------------------------------------------------------------------------------
// tile kernel
// ------------------------------------------------------------------------------------------------

struct ShadeParams {
  float view_pos[3];
  int32_t enable_normal_map;
  int32_t num_lights;
};

BB_DEV RasterTri load_tri(const RasterTri *__restrict__ tris, const ClipSlot *__restrict__ clip_arena, uint32_t ref,
                          bool &clipped, uint32_t &slot_index) {
  uint32_t prim = ref >> 3, sub = ref & 7u;
  RasterTri t = tris[prim];
  clipped = (t.X0 == kClippedSentinel);
  slot_index = 0;
  if (clipped) {
    slot_index = (uint32_t)t.Y0 + sub;
    t = clip_arena[slot_index].tri;
  }
  return t;
}

// Exact coverage of one pixel centre + depth; updates the LDS key with ds_max_u64.
BB_DEV void raster_pixel(const RasterTri &t, long long b0, long long b1, long long b2, int px, int py, uint32_t ref,
                         unsigned long long *keys, int key_index) {
  int Xc = px * 256 + 128, Yc = py * 256 + 128;
  long long dx0 = (long long)t.X1 - t.X0, dy0 = (long long)t.Y1 - t.Y0;
  long long dx1 = (long long)t.X2 - t.X1, dy1 = (long long)t.Y2 - t.Y1;
  long long dx2 = (long long)t.X0 - t.X2, dy2 = (long long)t.Y0 - t.Y2;
  long long E0 = dx0 * (long long)(Yc - t.Y0) - dy0 * (long long)(Xc - t.X0) + b0;
  long long E1 = dx1 * (long long)(Yc - t.Y1) - dy1 * (long long)(Xc - t.X1) + b1;
  long long E2 = dx2 * (long long)(Yc - t.Y2) - dy2 * (long long)(Xc - t.X2) + b2;
  if ((E0 | E1 | E2) < 0) return;
  float dxp = (float)(Xc - t.X0), dyp = (float)(Yc - t.Y0);
  float z = fmaf(t.dzdx, dxp, fmaf(t.dzdy, dyp, t.z0));
  if (!(z >= 0.0f)) z = 0.0f;
  if (z > 1.0f) z = 1.0f;
  unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)(ref + 1u);
  atomicMax(&keys[key_index], key);
}

BB_DEV long long edge_bias(long long dx, long long dy) {
  // top-left rule: a pixel centre exactly on an edge belongs to the triangle only for top/left edges
  return (dy < 0 || (dy == 0 && dx > 0)) ? 0ll : -1ll;
}

// pixel index inside the tile, 8x8-blocked so that 64 consecutive indices form one 8x8 quad block
template <int TILE_W>
BB_DEV void tile_pixel(int p, int &x, int &y) {
  int block = p >> 6, within = p & 63;
  constexpr int BX = TILE_W / 8;
  x = (block % BX) * 8 + (within & 7);
  y = (block / BX) * 8 + (within >> 3);
}
template <int TILE_W>
BB_DEV int tile_index(int x, int y) {
  constexpr int BX = TILE_W / 8;
  return (((y >> 3) * BX + (x >> 3)) << 6) | ((y & 7) << 3) | (x & 7);
}

template <int TILE_W, int TILE_H>
BB_DEV void raster_triangle_wave(const RasterTri &t, uint32_t ref, int tile_x0, int tile_y0, const FrameParams &fp,
                                 unsigned long long *keys, int lane) {
  int32_t minX = min(t.X0, min(t.X1, t.X2)), maxX = max(t.X0, max(t.X1, t.X2));
  int32_t minY = min(t.Y0, min(t.Y1, t.Y2)), maxY = max(t.Y0, max(t.Y1, t.Y2));
  int px0 = max(max((minX - 128 + 255) >> 8, 0), tile_x0);
  int px1 = min(min((maxX - 128) >> 8, fp.width - 1), tile_x0 + TILE_W - 1);
  int py0 = max(max((minY - 128 + 255) >> 8, 0), tile_y0);
  int py1 = min(min((maxY - 128) >> 8, fp.height - 1), tile_y0 + TILE_H - 1);
  if (px0 > px1 || py0 > py1) return;
  long long b0 = edge_bias((long long)t.X1 - t.X0, (long long)t.Y1 - t.Y0);
  long long b1 = edge_bias((long long)t.X2 - t.X1, (long long)t.Y2 - t.Y1);
  long long b2 = edge_bias((long long)t.X0 - t.X2, (long long)t.Y0 - t.Y2);
  // sweep the clipped bounding box in 8x8 blocks, one pixel per lane
  int bx0 = px0 & ~7, by0 = py0 & ~7;
  for (int by = by0; by <= py1; by += 8) {
    for (int bx = bx0; bx <= px1; bx += 8) {
      int px = bx + (lane & 7), py = by + (lane >> 3);
      if (px >= px0 && px <= px1 && py >= py0 && py <= py1)
        raster_pixel(t, b0, b1, b2, px, py, ref, keys, tile_index<TILE_W>(px - tile_x0, py - tile_y0));
    }
  }
}

template <int TILE_W, int TILE_H>
BB_DEV void raster_triangle_lane(const RasterTri &t, uint32_t ref, int tile_x0, int tile_y0, const FrameParams &fp,
                                 unsigned long long *keys, int px0, int px1, int py0, int py1) {
  long long b0 = edge_bias((long long)t.X1 - t.X0, (long long)t.Y1 - t.Y0);
  long long b1 = edge_bias((long long)t.X2 - t.X1, (long long)t.Y2 - t.Y1);
  long long b2 = edge_bias((long long)t.X0 - t.X2, (long long)t.Y0 - t.Y2);
  for (int py = py0; py <= py1; ++py)
    for (int px = px0; px <= px1; ++px)
      raster_pixel(t, b0, b1, b2, px, py, ref, keys, tile_index<TILE_W>(px - tile_x0, py - tile_y0));
}

constexpr int kTileThreads = 256;
constexpr int kSmallTriPixels = 48;  // bounding boxes up to this many pixels are rasterised one triangle per lane

template <int TILE_W, int TILE_H>
__global__ __launch_bounds__(kTileThreads) void k_tile(
    FrameParams fp, ShadeParams sp, const Light *__restrict__ lights, const RasterTri *__restrict__ tris,
    const PrimAttr *__restrict__ attrs, const ClipSlot *__restrict__ clip_arena, Counters *__restrict__ ctr,
    Counters *__restrict__ ctr_next, uint32_t *__restrict__ tile_count, const uint32_t *__restrict__ bins,
    const uint32_t *__restrict__ broad_list, const MaterialDesc *__restrict__ materials, float4 *__restrict__ out,
    uint32_t *__restrict__ vis_prim, float *__restrict__ vis_depth) {
  constexpr int TILE_PIXELS = TILE_W * TILE_H;
  __shared__ unsigned long long keys[TILE_PIXELS];
  __shared__ uint16_t list[TILE_PIXELS];
  __shared__ uint32_t big_list[kTileThreads * 4];
  __shared__ uint32_t s_count, s_big;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = blockIdx.x;
  // owned tile row -> global tile row
  int ty, out_tile_row;
  if (fp.world > 1) {
    int lb = blockIdx.y / fp.band_tiles, r = blockIdx.y - lb * fp.band_tiles;
    ty = (lb * fp.world + fp.rank) * fp.band_tiles + r;
    out_tile_row = blockIdx.y;
  } else {
    ty = blockIdx.y;
    out_tile_row = ty;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid < (int)(sizeof(Counters) / 4))
    reinterpret_cast<uint32_t *>(ctr_next)[tid] = 0;  // next frame's counter block
  if (ty >= fp.tiles_y) return;
  const uint32_t tile = (uint32_t)ty * (uint32_t)fp.tiles_x + (uint32_t)tx;
  const int tile_x0 = tx * TILE_W, tile_y0 = ty * TILE_H;

  for (int i = tid; i < TILE_PIXELS; i += kTileThreads) keys[i] = 0ull;
  if (tid == 0) {
    s_count = 0;
    s_big = 0;
  }
  const uint32_t n_bin = min(tile_count[tile], fp.bin_cap);
  const uint32_t n_broad = min(ctr->n_broad, fp.broad_cap);
  __syncthreads();
  if (tid == 0) tile_count[tile] = 0;  // ready for the next frame

  // ---- raster phase A: one triangle per lane for small bounding boxes; big ones are deferred ----
  const uint32_t *my_bin = bins + (size_t)tile * fp.bin_cap;
  for (uint32_t base = 0; base < n_bin; base += kTileThreads) {
    uint32_t i = base + tid;
    if (i < n_bin) {
      uint32_t ref = my_bin[i];
      bool clipped;
      uint32_t slot;
      RasterTri t = load_tri(tris, clip_arena, ref, clipped, slot);
      int32_t minX = min(t.X0, min(t.X1, t.X2)), maxX = max(t.X0, max(t.X1, t.X2));
      int32_t minY = min(t.Y0, min(t.Y1, t.Y2)), maxY = max(t.Y0, max(t.Y1, t.Y2));
      int px0 = max(max((minX - 128 + 255) >> 8, 0), tile_x0);
      int px1 = min(min((maxX - 128) >> 8, fp.width - 1), tile_x0 + TILE_W - 1);
      int py0 = max(max((minY - 128 + 255) >> 8, 0), tile_y0);
      int py1 = min(min((maxY - 128) >> 8, fp.height - 1), tile_y0 + TILE_H - 1);
      if (px0 <= px1 && py0 <= py1) {
        int area = (px1 - px0 + 1) * (py1 - py0 + 1);
        if (area <= kSmallTriPixels) {
          raster_triangle_lane<TILE_W, TILE_H>(t, ref, tile_x0, tile_y0, fp, keys, px0, px1, py0, py1);
        } else {
          uint32_t s = atomicAdd(&s_big, 1u);
          big_list[s & (kTileThreads * 4 - 1)] = ref;  // capacity handled by flushing below
        }
      }
    }
    // flush deferred big triangles whenever the list could overflow on the next round
    __syncthreads();
    uint32_t nb = s_big;
    if (nb > kTileThreads * 3 || base + kTileThreads >= n_bin) {
      for (uint32_t j = wave; j < nb; j += kTileThreads / 64) {
        uint32_t ref = big_list[j];
        bool clipped;
        uint32_t slot;
        RasterTri t = load_tri(tris, clip_arena, ref, clipped, slot);
        raster_triangle_wave<TILE_W, TILE_H>(t, ref, tile_x0, tile_y0, fp, keys, lane);
      }
      __syncthreads();
      if (tid == 0) s_big = 0;
      __syncthreads();
    }
  }
  // ---- raster phase B: broad list (triangles that touch many tiles), one triangle per wave ----
  for (uint32_t j = wave; j < n_broad; j += kTileThreads / 64) {
    uint32_t ref = broad_list[j];
    bool clipped;
    uint32_t slot;
    RasterTri t = load_tri(tris, clip_arena, ref, clipped, slot);
    raster_triangle_wave<TILE_W, TILE_H>(t, ref, tile_x0, tile_y0, fp, keys, lane);
  }
  __syncthreads();

  // ---- compaction: covered pixels -> list (ballot + popcount prefix); background written here ----
  const int out_y0 = out_tile_row * TILE_H;
  for (int base = 0; base < TILE_PIXELS; base += kTileThreads) {
    int p = base + tid;
    int x, y;
    tile_pixel<TILE_W>(p, x, y);
    int gx = tile_x0 + x, gy = tile_y0 + y;
    bool in_frame = gx < fp.width && gy < fp.height;
    unsigned long long key = keys[p];
    bool covered = in_frame && key != 0ull;
    unsigned long long mask = __ballot(covered);
    uint32_t wave_base = 0;
    if (lane == 0 && mask) wave_base = atomicAdd(&s_count, (uint32_t)__popcll(mask));
    wave_base = __shfl(wave_base, 0);
    if (covered) {
      uint32_t rank_in_wave = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      list[wave_base + rank_in_wave] = (uint16_t)p;
    } else if (in_frame) {
      size_t o = (size_t)(out_y0 + y) * (size_t)fp.width + (size_t)gx;
      out[o] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // clear colour, src/main.cpp:84
    }
    if (vis_prim && in_frame) {
      size_t o = (size_t)gy * (size_t)fp.width + (size_t)gx;
      vis_prim[o] = key ? (((uint32_t)key - 1u) >> 3) : 0xFFFFFFFFu;
      vis_depth[o] = __uint_as_float((uint32_t)(key >> 32));
    }
  }
  __syncthreads();
  const uint32_t n_cov = s_count;
  if (tid == 0 && n_cov) atomicAdd(&ctr->n_shaded, (unsigned long long)n_cov);

  // ---- shading: forward_brdf.frag once per visible pixel ----
  for (uint32_t i = tid; i < n_cov; i += kTileThreads) {
    int p = list[i];
    int x, y;
    tile_pixel<TILE_W>(p, x, y);
    int gx = tile_x0 + x, gy = tile_y0 + y;
    uint32_t ref = (uint32_t)keys[p] - 1u;
    uint32_t prim = ref >> 3;
    bool clipped;
    uint32_t slot;
    RasterTri t = load_tri(tris, clip_arena, ref, clipped, slot);

    // perspective-correct barycentrics
    int Xc = gx * 256 + 128, Yc = gy * 256 + 128;
    float dxp = (float)(Xc - t.X0), dyp = (float)(Yc - t.Y0);
    float l1 = fmaf(t.l1dx, dxp, t.l1dy * dyp);
    float l2 = fmaf(t.l2dx, dxp, t.l2dy * dyp);
    float l0 = (1.0f - l1) - l2;
    float u0 = l0 * t.rw0, u1 = l1 * t.rw1, u2 = l2 * t.rw2;
    float r = 1.0f / ((u0 + u1) + u2);
    float b0 = u0 * r, b1 = u1 * r, b2 = u2 * r;
    if (clipped) {
      const ClipSlot &cs = clip_arena[slot];
      float c0 = fmaf(b2, cs.bary[2][0], fmaf(b1, cs.bary[1][0], b0 * cs.bary[0][0]));
      float c1 = fmaf(b2, cs.bary[2][1], fmaf(b1, cs.bary[1][1], b0 * cs.bary[0][1]));
      float c2 = fmaf(b2, cs.bary[2][2], fmaf(b1, cs.bary[1][2], b0 * cs.bary[0][2]));
      b0 = c0; b1 = c1; b2 = c2;
    }

    const PrimAttr &pa = attrs[prim];
    float a[kNumVary];
#pragma unroll
    for (int k = 0; k < kNumVary; ++k) a[k] = fmaf(b2, pa.vary[2][k], fmaf(b1, pa.vary[1][k], b0 * pa.vary[0][k]));
    const MaterialDesc &md = materials[pa.material];

    // texture fetches, forward_brdf.frag:16-22
    const float u = a[0], v = a[1];
    f3 albedo;
    float metallic, roughness, ao;
    {
      const TexDesc &td = md.maps[kMapAlbedo];
      BilinearTaps tp = bilinear_taps(u, v, td.w, td.h);
      const uint32_t *tx32 = reinterpret_cast<const uint32_t *>(td.texels);
      uint32_t t00 = tx32[tp.o00], t10 = tx32[tp.o10], t01 = tx32[tp.o01], t11 = tx32[tp.o11];
      albedo.x = filter_channel(t00, t10, t01, t11, 0, tp.fx, tp.fy);
      albedo.y = filter_channel(t00, t10, t01, t11, 8, tp.fx, tp.fy);
      albedo.z = filter_channel(t00, t10, t01, t11, 16, tp.fx, tp.fy);
    }
    {
      const TexDesc &td = md.maps[kMapMetallic];
      BilinearTaps tp = bilinear_taps(u, v, td.w, td.h);
      const uint32_t *tx32 = reinterpret_cast<const uint32_t *>(td.texels);
      metallic = filter_channel(tx32[tp.o00], tx32[tp.o10], tx32[tp.o01], tx32[tp.o11], 0, tp.fx, tp.fy);
    }
    {
      const TexDesc &td = md.maps[kMapRoughness];
      BilinearTaps tp = bilinear_taps(u, v, td.w, td.h);
      const uint32_t *tx32 = reinterpret_cast<const uint32_t *>(td.texels);
      roughness = filter_channel(tx32[tp.o00], tx32[tp.o10], tx32[tp.o01], tx32[tp.o11], 0, tp.fx, tp.fy);
    }
    {
      const TexDesc &td = md.maps[kMapAO];
      BilinearTaps tp = bilinear_taps(u, v, td.w, td.h);
      const uint32_t *tx32 = reinterpret_cast<const uint32_t *>(td.texels);
      ao = filter_channel(tx32[tp.o00], tx32[tp.o10], tx32[tp.o01], tx32[tp.o11], 0, tp.fx, tp.fy);
    }
    f3 normal;
    if (sp.enable_normal_map != 0) {
      const TexDesc &td = md.maps[kMapNormal];
      BilinearTaps tp = bilinear_taps(u, v, td.w, td.h);
      const uint32_t *tx32 = reinterpret_cast<const uint32_t *>(td.texels);
      uint32_t t00 = tx32[tp.o00], t10 = tx32[tp.o10], t01 = tx32[tp.o01], t11 = tx32[tp.o11];
      f3 nt = mk3(fmaf(filter_channel(t00, t10, t01, t11, 0, tp.fx, tp.fy), 2.0f, -1.0f),
                  fmaf(filter_channel(t00, t10, t01, t11, 8, tp.fx, tp.fy), 2.0f, -1.0f),
                  fmaf(filter_channel(t00, t10, t01, t11, 16, tp.fx, tp.fy), 2.0f, -1.0f));
      // vTBN * nt, vTBN = mat3(T, B, N)
      normal.x = fmaf(a[5], nt.z, fmaf(a[11], nt.y, a[8] * nt.x));
      normal.y = fmaf(a[6], nt.z, fmaf(a[12], nt.y, a[9] * nt.x));
      normal.z = fmaf(a[7], nt.z, fmaf(a[13], nt.y, a[10] * nt.x));
    } else {
      normal = normalize3(mk3(a[5], a[6], a[7]));
    }

    // loop invariants of forward_brdf.frag:51-52 hoisted (bit-identical: same inputs, same operations)
    const f3 P = mk3(a[2], a[3], a[4]);
    const f3 V = normalize3(sub3(ld3(sp.view_pos), P));
    const f3 N = normalize3(normal);
    const float NdotV = max0(dot3(V, N));
    const float rr = roughness + 1.0f;
    const float kk = (rr * rr) * 0.125f;
    const float G_V = geometry_schlick_ggx(NdotV, kk);
    const f3 F0 = mk3(mixf(0.04f, albedo.x, metallic), mixf(0.04f, albedo.y, metallic), mixf(0.04f, albedo.z, metallic));
    const float om = 1.0f - metallic;

    f3 Lo = mk3(0.0f, 0.0f, 0.0f);
    for (int li = 0; li < sp.num_lights; ++li) {
      const Light &light = lights[li];
      f3 L;
      float att;
      if (light.type == 0 || light.type == 1) {
        f3 Lv = sub3(ld3(light.pos), P);
        float d = sqrtf(dot3(Lv, Lv));
        att = 1.0f / (d * d);
        L = scale3(Lv, 1.0f / d);
        if (light.type == 1) {
          float theta = dot3(L, normalize3(neg3(ld3(light.dir))));
          float epsilon = light.inner_cutoff - light.outer_cutoff;
          att *= clamp01((theta - light.outer_cutoff) / epsilon);
        }
      } else if (light.type == 2) {
        L = neg3(normalize3(ld3(light.dir)));
        att = 1.0f;
      } else {
        continue;
      }
      f3 H = normalize3(add3(L, V));
      float D = distribution_ggx(dot3(N, H), roughness);
      float x = 1.0f - max0(dot3(H, V));
      float x2 = x * x;
      float p5 = (x2 * x2) * x;
      f3 F = mk3(fmaf(1.0f - F0.x, p5, F0.x), fmaf(1.0f - F0.y, p5, F0.y), fmaf(1.0f - F0.z, p5, F0.z));
      float NdotL = max0(dot3(N, L));
      float G = G_V * geometry_schlick_ggx(NdotL, kk);
      f3 radiance = mk3((att * light.color[0]) * light.intensity, (att * light.color[1]) * light.intensity,
                        (att * light.color[2]) * light.intensity);
      float sden = (4.0f * NdotV) * NdotL;
      if (!(sden > 0.001f)) sden = 0.001f;
      float rden = 1.0f / sden;
      f3 spec = mk3(((D * F.x) * G) * rden, ((D * F.y) * G) * rden, ((D * F.z) * G) * rden);
      f3 kD = mk3((1.0f - F.x) * om, (1.0f - F.y) * om, (1.0f - F.z) * om);
      Lo.x = fmaf(fmaf(kD.x * albedo.x, kInvPi, spec.x) * radiance.x, NdotL, Lo.x);
      Lo.y = fmaf(fmaf(kD.y * albedo.y, kInvPi, spec.y) * radiance.y, NdotL, Lo.y);
      Lo.z = fmaf(fmaf(kD.z * albedo.z, kInvPi, spec.z) * radiance.z, NdotL, Lo.z);
    }
    float4 color;
    color.x = fmaf(0.03f * albedo.x, ao, Lo.x);
    color.y = fmaf(0.03f * albedo.y, ao, Lo.y);
    color.z = fmaf(0.03f * albedo.z, ao, Lo.z);
    color.w = 1.0f;
    size_t o = (size_t)(out_y0 + y) * (size_t)fp.width + (size_t)gx;
    out[o] = color;
  }
}

// ------------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------------

// [world][shard_rows][width] float4 -> row-major frame (screen-band un-interleave after the all-gather)
__global__ void k_unpack_gathered(const float4 *__restrict__ gathered, float4 *__restrict__ frame, int width, int height,
                                  int world, int band_rows, int shard_rows) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t n = (size_t)width * (size_t)height;
  if (i >= n) return;
  int y = (int)(i / (size_t)width), x = (int)(i - (size_t)y * width);
  int band = y / band_rows, r = y - band * band_rows;
  int rank = band % world, lb = band / world;
  size_t src = ((size_t)rank * shard_rows + (size_t)lb * band_rows + r) * (size_t)width + x;
  frame[i] = gathered[src];
}

// hdr_tone_mapping.frag:9-18 on the fp32 frame (next row)
__global__ void k_tone_map(float4 *__restrict__ frame, size_t n, int enable, float exposure) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 c = frame[i];
  if (enable) {
    c.x = 1.0f - expf(-c.x * exposure);
    c.y = 1.0f - expf(-c.y * exposure);
    c.z = 1.0f - expf(-c.z * exposure);
  }
  c.w = 1.0f;
  frame[i] = c;
}

}  // namespace bbr
