// bb_kernels.hip.h -- HIP kernels of the forward PBR path for gfx950 (MI355X / CDNA4).
//
//   k_geometry   forward_brdf.vert + clip/cull/viewport/snap + triangle setup + tile binning
//                (reference: src/shaders/forward_brdf.vert:24-37; state src/render.cpp:1069-1125)
//   k_raster     per screen tile: LDS-resident 64-bit visibility keys (depth | primitive) filled with
//                ds_max_u64, ballot/popcount compaction of the covered pixels into the tile's fragment list
//   k_shade_items  the frame's shading work list: one item per 64 fragments of a tile's list (scan of the per-tile counts);
//                also cooks the frame's light table
//   k_shade      forward_brdf.frag + brdf.glsl once per visible pixel (src/shaders/forward_brdf.frag:15-76,
//                brdf.glsl:2-36): one wave per item; a TAIL instantiation loops over what the main launch's estimate missed
//   k_present, k_tone_map, k_deferred_background, k_shade_overlay, k_pack_shard / k_unpack_*: the rows either side of the path
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

// Diagnostic ablations (skip parts of the pipeline to see what they cost) exist only in -DBB_ABLATE builds; in the
// shipped library the tests fold to constants.
#ifdef BB_ABLATE
#define BB_ABLATE(bits) ((fp.ablate & (bits)) != 0u)
#else
#define BB_ABLATE(bits) false
#endif

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
// GLSL inversesqrt as a fixed sequence (integer seed + three Newton steps, max error 1.1 ulp): the same bits as the
// oracle's bb_rsqrt, at a third of the instruction count of IEEE sqrt + divide.
// The IEEE paths behind the guards are real function calls: taken for zero, denormal, infinite and NaN arguments only,
// they would otherwise be expanded inline (a 12-instruction division, a 25-instruction square root) at each of the
// shader's 25 call sites -- a tenth of k_shade's code.
BB_DEV float bb_rsqrt_slow(float x) { return 1.0f / sqrtf(x); }
BB_DEV float bb_rcp_slow(float x) { return 1.0f / x; }
// The guards are WAVE-level: the fast sequence runs unconditionally and a uniform, almost never taken branch repairs
// the lanes whose argument was not a normal number.  (As a per-lane if/else the compiler wrapped every call site in an
// exec-mask save / restore -- five scalar instructions and a branch around each of the shader's sixteen guards per
// pixel, a tenth of a wave's issue slots.)
BB_DEV float bb_rsqrt(float x) {
  float y = __uint_as_float(0x5F375A86u - (__float_as_uint(x) >> 1));
  const float h = 0.5f * x;
  y = y * fmaf(-(h * y), y, 1.5f);
  y = y * fmaf(-(h * y), y, 1.5f);
  y = y * fmaf(-(h * y), y, 1.5f);
  const bool odd = !__builtin_amdgcn_classf(x, 0x100);  // anything but a positive normal number (one v_cmp_class)
  if (__builtin_expect(__ballot(odd) != 0ull, 0)) {
    if (odd) y = bb_rsqrt_slow(x);
  }
  return y;
}
// Reciprocal: the correctly rounded 1/x, i.e. what the oracle computes with an IEEE division.  v_rcp_f32 (1 ulp) plus
// ONE Newton step lands on the correctly rounded value for every one of the 2 113 929 216 normal inputs whose
// reciprocal is normal (k_selftest_rcp below checks all of them against the IEEE division in 0.3 s; so does
// tools/microbench/exact_rcp.hip) -- the result does not depend on which 1-ulp seed the hardware returns.  The same
// idea does not work for 1/sqrt (13 % of the inputs end up off by an ulp), so bb_rsqrt keeps its integer seed.
BB_DEV float bb_rcp(float x) {
  const float y0 = __builtin_amdgcn_rcpf(x);
  float y = fmaf(y0, fmaf(-x, y0, 1.0f), y0);
  // a seed that is not a normal number (x zero, denormal, huge, infinite or NaN): the IEEE division.  One v_cmp_class
  // on the seed replaces two range compares on x.
  const bool odd = !__builtin_amdgcn_classf(y0, 0x108);
  if (__builtin_expect(__ballot(odd) != 0ull, 0)) {
    if (odd) y = bb_rcp_slow(x);
  }
  return y;
}
// bb_rcp for a call site that can PROVE its argument is a normal number with a normal reciprocal (the guard and its
// branch are a twentieth of k_shade's instruction stream).  Each use states its proof; k_selftest_rcp checks the
// whole range [2^-100, 2^100] against the IEEE division.
BB_DEV float bb_rcp_normal(float x) {
  const float y = __builtin_amdgcn_rcpf(x);
  return fmaf(y, fmaf(-x, y, 1.0f), y);
}
// nearest binary16 value (ties to even), as binary32: v_cvt_f16_f32 / v_cvt_f32_f16 do exactly this on gfx950
// (round-to-nearest-even, binary16 subnormals kept -- the default float mode of HIP kernels; every binary16
// midpoint +-3 ulp is checked against the oracle's integer formulation in tests/test_gpu_parity.py)
// The empty asm pins the binary32 value: without it the compiler folds a producing fma and the conversion into one
// v_fma_mix*_f16, which rounds the exact fma result ONCE to binary16 -- different from rounding the binary32 result
// (what an attachment write does) exactly when that result sits on a binary16 tie.
BB_DEV float bb_half_round(float x) {
  asm("" : "+v"(x));
  return (float)(_Float16)x;
}

BB_DEV float bb_exp(float x) {
  if (!(x >= -104.0f)) return x < -104.0f ? 0.0f : x;
  if (x > 88.7228317f) return __uint_as_float(0x7F800000u);
  const float n = __builtin_rintf(x * 1.44269502f);
  float r = fmaf(n, -0.693145752f, x);
  r = fmaf(n, -1.42860677e-06f, r);
  float p = 1.98412701e-04f;
  p = fmaf(p, r, 1.38888892e-03f);
  p = fmaf(p, r, 8.33333377e-03f);
  p = fmaf(p, r, 4.16666679e-02f);
  p = fmaf(p, r, 1.66666672e-01f);
  p = fmaf(p, r, 0.5f);
  p = fmaf(p, r, 1.0f);
  p = fmaf(p, r, 1.0f);
  const int ni = (int)n, h = ni / 2;
  const float s1 = __uint_as_float((uint32_t)(h + 127) << 23), s2 = __uint_as_float((uint32_t)(ni - h + 127) << 23);
  return (p * s1) * s2;
}

// sRGB byte = number of thresholds t_k <= c (NaN -> 0).  Instead of a binary search (eight dependent LDS probes)
// the byte is read from a table keyed by the top 16 bits of the float: 256 cells per octave over the 13 octaves
// [2^-13, 1) that contain all thresholds.  A cell is narrower than the closest pair of thresholds (0.39 % against
// >= 0.88 %; checked when the table is built), so it holds the count at its lower edge and at most one more
// threshold, which one compare settles: two LDS reads, same bytes as the search for every float.
constexpr uint32_t kSrgbLutFirstExp = 114u;             // 2^-13
constexpr uint32_t kSrgbLutCells = 13u * 256u;          // biased exponents 114..126
struct SrgbTables {                                      // device copy built by the host
  float thr[256];                                        // t_1..t_255, then +inf
  uint8_t lut[kSrgbLutCells];                            // thresholds <= lower edge of the cell
};

BB_DEV uint32_t srgb8(float c, const SrgbTables &t) {
  const uint32_t u = __float_as_uint(c);
  if ((int32_t)u < (int32_t)(kSrgbLutFirstExp << 23)) return 0u;  // below 2^-13, zero, negative, negative NaN
  if (u >= (127u << 23)) return u <= 0x7F800000u ? 255u : 0u;     // >= 1 (and +inf) -> 255, NaN -> 0
  const uint32_t k = t.lut[(u >> 15) - (kSrgbLutFirstExp << 8)];
  return k + (t.thr[k] <= c ? 1u : 0u);
}

// One presented pixel (hdr_tone_mapping.frag:9-18 + the sRGB UNORM8 attachment write): binary16 HDR value, tone map,
// sRGB byte per channel, alpha 255.  Shared by k_present and the fused output of k_shade / k_raster.
BB_DEV uint32_t present_pixel(float r, float g, float b, const SrgbTables &t, int enable, float exposure, int hdr16) {
  float v[3] = {r, g, b};
  uint32_t px = 0xFF000000u;  // outColor.a = 1.0
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float x = v[k];
    if (hdr16) x = bb_half_round(x);
    if (enable) x = 1.0f - bb_exp(-x * exposure);
    px |= srgb8(x, t) << (8 * k);
  }
  return px;
}

BB_DEV f3 normalize3(f3 a) { return scale3(a, bb_rsqrt(dot3(a, a))); }
// max(a, 0) with the oracle's `a > 0 ? a : 0` semantics: NaN -> +0, -0 -> +0.  v_max_f32 does exactly that (IEEE
// maxNum with -0 < +0; k_selftest_rcp checks every bit pattern), and costs one instruction where the compiler, which
// may not assume the -0 case, emits a compare and a select.
BB_DEV float max0(float a) { return __builtin_fmaxf(a, 0.0f); }
// saturate: clamp to [0, 1], NaN -> +0 (the `clamp` output modifier of the producing instruction under DX10_CLAMP, the
// float mode HIP kernels run in: no instruction of its own, where v_max_f32 is a half-rate one that also breaks the
// 2-cycle issue cadence of the fma / mul stream around it -- profiles/r02_issue_rate.txt)
BB_DEV float sat01(float a) { return __builtin_amdgcn_fmed3f(a, 0.0f, 1.0f); }

// 32-bit integer multiply that stays v_mul_lo_u32.  The 24-bit forms the compiler prefers for small operands (v_mul_u32_u24,
// v_mad_i32_i24, ...) cost ~10 issue cycles each on gfx950 where v_mul_lo_u32 costs 3.5 (profiles/r02_issue_rate.txt, rows
// iso_*): the empty asm hides the operand's known-zero bits from the instruction selector.
BB_DEV int mul32(int a, int b) {
  asm("" : "+v"(a));
  return a * b;
}

// A pixel of the frame is written once and never read again by these kernels: a NON-TEMPORAL store (global_store ... nt).
// 133 MB of pixels per 4K frame otherwise stream through the L2s as ordinary dirty lines and push out what the shading
// does re-read -- texels and primitive records: with nt the pipelined C3 frame went 122.2 -> 110.6 us, C2 26.5 -> 24.9
// (plain / nt / sc1: 122.2 / 110.6 / 120.0 us; sc1 drops the line from the L2 once written back, nt marks it
// first-to-go from the start).
typedef float v4f __attribute__((ext_vector_type(4)));
BB_DEV void store_pixel(float4 *p, float4 v) {
  const v4f q = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(q, reinterpret_cast<v4f *>(p));  // one global_store_dwordx4 ... nt
}
BB_DEV void store_pixel(uint32_t *p, uint32_t v) { __builtin_nontemporal_store(v, p); }
// (The intermediates that are written once and read once -- fragment words, RasterTri records -- stay ordinary accesses:
//  non-temporal they made the frame 2 % slower, 112.1 -> 114.8 / 113.8 us: their one reader does find them in the L2.)

template <typename M4>  // (Mat4 in any address space)
BB_DEV f4 mat4_mul(const M4 &m, f4 v) {
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

// Per-wave LDS workspace of the polygon clipper.  The clipper indexes its vertex arrays dynamically; in
// registers that would spill to scratch (and a kernel that owns scratch pays for it on every launch), in LDS
// it is a plain ds_read/ds_write.  Clipped primitives are rare; the wave clips them one after the other,
// all lanes working on the same polygon.
struct ClipWork {
  ClipVert poly[2][kMaxClipVerts];  // ping-pong
  int32_t X[kMaxClipVerts], Y[kMaxClipVerts];
  float rw[kMaxClipVerts], z[kMaxClipVerts];
  uint32_t prim, base;  // primitive being clipped, its first clip-arena slot
  int32_t n_slots;      // fan triangles of the clipped polygon (0: nothing survived)
  int32_t n_valid;      // of which front-facing and covering a pixel centre
};

// Wave-parallel Sutherland-Hodgman against near, far and the four guard-band planes: lane i owns edge
// (v_i, v_i+1), emits 0, 1 or 2 vertices, and a ballot/popcount prefix gives every lane its output position -- the
// same vertices in the same order as the serial algorithm, in six steps instead of six loops over the polygon.
// The input polygon (n vertices) is in w.poly[0]; returns the buffer index holding the result and its size.
BB_DEV int clip_polygon_wave(ClipWork &w, int &n) {
  const int lane = threadIdx.x & 63;
  int cur = 0;
  for (int plane = 0; plane < 6; ++plane) {
    bool ina = false, cross = false;
    ClipVert a, b;
    float da = 0.0f, db = 0.0f;
    if (lane < n) {
      a = w.poly[cur][lane];
      b = w.poly[cur][lane + 1 == n ? 0 : lane + 1];
      da = plane_dist(a.c, plane);
      db = plane_dist(b.c, plane);
      ina = da >= 0.0f;
      cross = ina != (db >= 0.0f);
    }
    const unsigned long long ma = __ballot(ina), mx = __ballot(cross);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int pos = (int)__popcll(ma & lt) + (int)__popcll(mx & lt);
    __builtin_amdgcn_wave_barrier();
    if (ina) w.poly[cur ^ 1][pos++] = a;
    if (cross) {
      ClipVert r;
      if (ina) clip_lerp(a, da, b, db, r);
      else clip_lerp(b, db, a, da, r);
      w.poly[cur ^ 1][pos] = r;
    }
    __builtin_amdgcn_wave_barrier();
    n = (int)__popcll(ma) + (int)__popcll(mx);
    cur ^= 1;
    if (n < 3) {
      n = 0;
      break;
    }
  }
  return cur;
}

// viewport transform: half extents and centre in pixels (the main passes cover the whole target: centre = half extent)
struct Viewport {
  float half_w, half_h, cx, cy;
};

BB_DEV bool project_vertex(const float *c, const Viewport &vp, int32_t &X, int32_t &Y, float &rw, float &zndc) {
  float w = c[3];
  if (!(w > 0.0f)) return false;
  float r = 1.0f / w;
  float xs = fmaf(c[0] * r, vp.half_w, vp.cx);
  float ys = fmaf(c[1] * r, vp.half_h, vp.cy);
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

// ---- statistics: accumulated per workgroup in LDS, written out as one BlockStats record per workgroup ----

struct TileRange {
  int tx0, tx1, ty0, ty1;
};

// pixel-centre bounding box -> tile range; false if the box holds no pixel centre
template <int TILE_W, int TILE_H>
BB_DEV bool tile_range(const RasterTri &t, const FrameParams &fp, TileRange &r) {
  int32_t minX = min(t.X0, min(t.X1, t.X2)), maxX = max(t.X0, max(t.X1, t.X2));
  int32_t minY = min(t.Y0, min(t.Y1, t.Y2)), maxY = max(t.Y0, max(t.Y1, t.Y2));
  int32_t px0 = max((minX - 128 + 255) >> 8, 0), px1 = min((maxX - 128) >> 8, fp.width - 1);
  int32_t py0 = max((minY - 128 + 255) >> 8, 0), py1 = min((maxY - 128) >> 8, fp.height - 1);
  if (px0 > px1 || py0 > py1) return false;
  r.tx0 = px0 / TILE_W; r.tx1 = px1 / TILE_W; r.ty0 = py0 / TILE_H; r.ty1 = py1 / TILE_H;
  return true;
}

// Raster classes: how many lanes of k_raster work on one triangle.  Decided once per triangle from its
// full pixel bounding box, so that every loop of the raster kernel runs over triangles of similar cost.
//   0: tiny  (box <= 64 px, spans <= 32 px)      one triangle per lane, 32-bit stepped edge functions
//   1: small (spans <= 64 px)                    16 lanes per triangle, 4x4 pixel blocks, 24-bit multiply-adds
//   2: large                                     one wave per triangle, 8x8 blocks, trivial accept / reject
constexpr uint32_t kBinClasses = 3;

BB_DEV uint32_t raster_class(const RasterTri &t, const FrameParams &fp) {
  int32_t minX = min(t.X0, min(t.X1, t.X2)), maxX = max(t.X0, max(t.X1, t.X2));
  int32_t minY = min(t.Y0, min(t.Y1, t.Y2)), maxY = max(t.Y0, max(t.Y1, t.Y2));
  int32_t px0 = max((minX - 128 + 255) >> 8, 0), px1 = min((maxX - 128) >> 8, fp.width - 1);
  int32_t py0 = max((minY - 128 + 255) >> 8, 0), py1 = min((maxY - 128) >> 8, fp.height - 1);
  int ext = max(maxX - minX, maxY - minY);
  int area = (px1 - px0 + 1) * (py1 - py0 + 1);
  if (ext <= 32 * 256 && area <= 64) return 0u;
  if (ext <= 64 * 256) return 1u;
  return 2u;
}

BB_DEV void broad_insert(const RasterTri &t, uint32_t ref, const FrameParams &fp, Counters *ctr, BroadTri *broad_list) {
  uint32_t slot = atomicAdd(&ctr->n_broad, 1u);
  if (slot < fp.broad_cap) {
    BroadTri b;
    b.tri = t;
    b.ref = ref;
    b.pad[0] = b.pad[1] = b.pad[2] = 0;
    broad_list[slot] = b;
  } else {
    atomicOr(&ctr->overflow, 2u);
  }
}

// Wave-aggregated bin insertion.  Consecutive primitives of a mesh land in the same few tiles, so the lanes of
// a wave that target one bin segment are grouped (ballot loop, ALU only) and the lowest lane of each group
// reserves all the group's slots with ONE returning atomic; the leaders of all groups issue their atomics in the
// same instruction, so a wave pays one memory round trip however many tiles it touches, and a tile that
// receives thousands of tiny triangles sees tens of atomics instead of thousands.
// Must be called by all lanes of the wave (has = false for lanes with nothing to insert).
// The reservation is split from the write so that a caller with several insertions per lane can issue all their
// atomics before waiting for the first result (one memory round trip for the batch instead of one per insertion).
struct BinTicket {
  uint32_t base;   // valid in the group's leader lane until redeemed
  uint32_t rank;
  uint32_t gsize;  // slots the leader reserved (leader lane only)
  int leader;
};

BB_DEV BinTicket wave_bin_reserve(bool has, uint32_t seg, uint32_t *tile_count) {
  const int lane = threadIdx.x & 63;
  unsigned long long pending = __ballot(has);
  BinTicket t;
  t.leader = lane;
  t.rank = 0;
  t.base = 0;
  uint32_t gsize = 0;
  while (pending) {
    int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)pending) - 1);
    uint32_t ls = (uint32_t)__builtin_amdgcn_readlane((int)seg, l);
    bool mine = has && seg == ls;
    unsigned long long m = __ballot(mine);
    if (mine) {
      t.leader = l;
      t.rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      gsize = (uint32_t)__popcll(m);
    }
    pending &= ~m;
  }
  if (has && lane == t.leader) t.base = atomicAdd(&tile_count[seg], gsize);
  t.gsize = gsize;
  return t;
}

// heavy: the frame's heavy-tile list (fp.heavy_threshold != 0).  The ONE reservation that takes a bin's count across the
// threshold appends the tile -- as the launch slot k_raster's screen order gives it, and the class of the bin -- so a tile
// is listed at most once per class, and which of its entries does the work follows from the final counts (k_raster).
BB_DEV void wave_bin_write(bool has, uint32_t seg, uint32_t ref, const BinTicket &t, const FrameParams &fp, Counters *ctr,
                           uint32_t *bins, uint32_t *heavy) {
  const uint32_t base = (uint32_t)__shfl((int)t.base, t.leader);
  if (heavy && has && (int)(threadIdx.x & 63) == t.leader && t.base < fp.heavy_threshold && t.base + t.gsize >= fp.heavy_threshold) {
    const uint32_t tile = seg / kBinClasses, c = seg - tile * kBinClasses;
    const uint32_t ty = tile / (uint32_t)fp.tiles_x, tx = tile - ty * (uint32_t)fp.tiles_x;
    uint32_t gy = ty;  // owned tile row -> grid row: the inverse of tile_row
    if (fp.world > 1) {
      const uint32_t band = ty / (uint32_t)fp.band_tiles;
      gy = (band / (uint32_t)fp.world) * (uint32_t)fp.band_tiles + (ty - band * (uint32_t)fp.band_tiles);
    }
    const uint32_t k = atomicAdd(&ctr->n_heavy, 1u);
    if (k < kBinClasses * (uint32_t)(fp.tiles_x * fp.tiles_y)) heavy[k] = (gy * (uint32_t)fp.tiles_x + tx) | (c << 30);
  }
  if (has) {
    uint32_t slot = base + t.rank;
    if (slot < fp.bin_cap) {
      bins[(size_t)seg * fp.bin_cap + slot] = ref;
    } else {
      atomicOr(&ctr->overflow, 1u);
      atomicMax(&ctr->bin_need, slot + 1u);
    }
  }
}

// Rare path: the primitive of lane `owner` crosses a clip plane.  The whole wave works on it: parallel polygon
// clip, lane i projects vertex i and then sets up fan triangle i (binary64 planes), stores its ClipSlot and its
// entry of the every-tile list -- a clipped primitive is typically huge (the ground plane), so its sub-triangles skip
// the bins.  One atomic reserves the arena slots, one the list entries.  Results in w.n_valid / w.base.
template <int TILE_W, int TILE_H>
BB_DEV void clip_primitive_wave(ClipWork &w, int owner, const float (*clip)[4], uint32_t prim, const FrameParams &fp,
                                const Viewport &vp, ClipSlot *clip_arena, Counters *ctr, BroadTri *broad_list) {
  const int lane = threadIdx.x & 63;
  if (lane == owner) {
    w.prim = prim;
    w.n_slots = 0;
    w.n_valid = 0;
    w.base = kNotClipped;
    for (int i = 0; i < 3; ++i) {
      ClipVert v;
      for (int k = 0; k < 4; ++k) v.c[k] = clip[i][k];
      v.b[0] = i == 0 ? 1.0f : 0.0f;
      v.b[1] = i == 1 ? 1.0f : 0.0f;
      v.b[2] = i == 2 ? 1.0f : 0.0f;
      w.poly[0][i] = v;
    }
  }
  __builtin_amdgcn_wave_barrier();
  int n = 3;
  const int cur = clip_polygon_wave(w, n);
  if (n < 3) return;
  // projection: lane i -> vertex i
  bool bad = false;
  if (lane < n) {
    const ClipVert v = w.poly[cur][lane];
    int32_t X = 0, Y = 0;
    float rw = 0.0f, z = 0.0f;
    bad = !project_vertex(v.c, vp, X, Y, rw, z);
    w.X[lane] = X; w.Y[lane] = Y; w.rw[lane] = rw; w.z[lane] = z;
  }
  if (__ballot(bad) != 0ull) return;
  const int n_slots = min(n - 2, kMaxSubTris);
  __builtin_amdgcn_wave_barrier();
  // fan triangle i = (v0, v_i, v_i+1): lane i - 1.  The setup needs no reservation, so it runs first and BOTH reservations
  // -- arena slots (lane 0) and every-tile entries (lane 1) -- go out together afterwards: one memory round trip per clipped
  // primitive instead of two (the ground plane's two primitives are k_geometry's critical path at 1080p: in-kernel stamps).
  bool ok = false;
  ClipSlot s;
  RasterTri tri = {};
  if (lane < n_slots) {
    const int i = lane + 1;
    tri.X0 = w.X[0]; tri.Y0 = w.Y[0];
    tri.X1 = w.X[i]; tri.Y1 = w.Y[i];
    tri.X2 = w.X[i + 1]; tri.Y2 = w.Y[i + 1];
    tri.rw0 = w.rw[0]; tri.rw1 = w.rw[i]; tri.rw2 = w.rw[i + 1];
    const ClipVert v0 = w.poly[cur][0], v1 = w.poly[cur][i], v2 = w.poly[cur][i + 1];
    for (int c = 0; c < 3; ++c) {
      s.bary[0][c] = v0.b[c];
      s.bary[1][c] = v1.b[c];
      s.bary[2][c] = v2.b[c];
    }
    s.pad = 0;
    ok = setup_tri(tri, w.z[0], w.z[i], w.z[i + 1]);
    TileRange tr;
    ok = ok && tile_range<TILE_W, TILE_H>(tri, fp, tr);
    s.valid = ok ? 1u : 0u;
    s.h = PlaneHead{tri.X0, tri.Y0, tri.l1dx, tri.l1dy, tri.l2dx, tri.l2dy, tri.rw0, tri.rw1, tri.rw2};
  }
  const unsigned long long m = __ballot(ok);
  const int n_valid = (int)__popcll(m);
  uint32_t got = 0;
  if (lane == 0) got = atomicAdd(&ctr->n_clip_slots, (uint32_t)n_slots);
  if (lane == 1 && n_valid) got = atomicAdd(&ctr->n_broad, (uint32_t)n_valid);
  const uint32_t base = (uint32_t)__shfl((int)got, 0), slot = (uint32_t)__shfl((int)got, 1);
  const bool arena_fits = base + (uint32_t)n_slots <= fp.clip_cap;
  const bool list_fits = slot + (uint32_t)n_valid <= fp.broad_cap;
  if (!arena_fits && lane == 0) atomicOr(&ctr->overflow, 4u);
  if (n_valid && !list_fits && lane == 0) atomicOr(&ctr->overflow, 2u);
  if (arena_fits && lane < n_slots) clip_arena[base + (uint32_t)lane] = s;
  if (lane == owner && arena_fits) {
    w.base = base;
    w.n_slots = n_slots;
  }
  if (n_valid == 0 || !list_fits) return;  // (an overflowed list: the frame takes none of its entries, k_raster)
  if (ok) {
    // the run of entries is reserved and fits, so every one of them is written: a real entry, or -- when the arena had
    // no room for the sub-triangles, and the frame is rendered again anyway -- an entry no tile can touch (all zero)
    BroadTri b = {};
    if (arena_fits) {
      b.tri = tri;
      b.ref = (w.prim << 3) | (uint32_t)lane;  // the OWNER's primitive (prim is per lane)
      b.pad[0] = base + (uint32_t)lane + 1u;     // its clip-arena slot + 1: travels into the fragment word (k_raster)
    }
    broad_list[slot + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = b;
  }
  if (lane == owner && arena_fits) w.n_valid = n_valid;
}

// One thread per primitive, all draw calls of the frame in one launch (API order = primitive index order).
// OVERLAY = true is the overlay subpass (light markers, corner gizmo): other vertex programs and a
// per-primitive viewport, everything downstream of the vertex stage shared.
template <int TILE_W, int TILE_H, bool OVERLAY = false>
__global__ __launch_bounds__(256) void k_geometry(const DrawDesc *__restrict__ draws, uint32_t n_draws, uint32_t n_prims,
                                                  FirstPrims first_prims,
                                                  RasterTri *__restrict__ tris, ShadeRec *__restrict__ recs,
                                                  uint32_t *__restrict__ tile_count, uint32_t *__restrict__ bins,
                                                  Counters *__restrict__ ctr,
                                                  Mat4 pv, Mat4 view, FrameParams fp, ClipSlot *__restrict__ clip_arena,
                                                  BroadTri *__restrict__ broad_list,
                                                  const MaterialDesc *__restrict__ materials,
                                                  BlockStats *__restrict__ block_stats, uint32_t *__restrict__ heavy) {
#ifdef BB_STAMPS
#define BB_STAMP(i) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long *>(clip_arena + fp.clip_cap)[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define BB_STAMP(i) do { } while (0)
#endif
  // This kernel's waves issue one instruction in ~18 cycles and wait for memory most of their lives; with frames in flight
  // they share their SIMDs with k_shade's, which want to issue all the time.  At the highest issue priority their few
  // instructions go out when they are ready and the wave gives its slot back sooner (k_raster does the same; C3 frame
  // -2 % on two boxes of three, 0 on the third; C5 the same as without).
  __builtin_amdgcn_s_setprio(3);
  BB_STAMP(0);
  __shared__ ClipWork s_clip[4];  // one per wave
  // (No workgroup barrier anywhere in this kernel: the four waves of a workgroup share nothing -- each has its own clip
  //  workspace and leaves its own statistics record -- so a wave retires when ITS primitives are done.)
  const uint32_t prim = blockIdx.x * blockDim.x + threadIdx.x;
  bool needs_clip = false;
  float clip[3][4];
  bool survives_out = false;  // this lane's primitive is rasterised unclipped (statistics)
  bool binned = false;  // this lane holds an unclipped, set-up triangle that goes to tile bins
  uint32_t cls = 0;     // raster class of the triangle: bin segment (kBinClasses per tile)
  TileRange tr = {0, -1, 0, -1};
  uint32_t clipped_raster = 0;  // wave-uniform: sub-triangles of this wave's clipped primitives that reached the every-tile list
  Viewport vp = {fp.half_w, fp.half_h, fp.half_w, fp.half_h};
  if (prim < n_prims) {
    // which draw: the first few draws' first_prim are kernel arguments (no load); more draws than that: the table
    uint32_t d = 0;
#pragma unroll
    for (int q = 0; q < kInlineFirstPrims; ++q) d += prim >= first_prims.v[q] ? 1u : 0u;
    if (n_draws > (uint32_t)kInlineFirstPrims + 1u)
      while (d + 1 < n_draws && prim >= draws[d + 1].first_prim) ++d;
    const DrawDesc draw = draws[d];
    const uint32_t local = prim - draw.first_prim;
    const uint32_t inst = local / draw.tris_per_instance;
    const uint32_t tri = local - inst * draw.tris_per_instance;

    BB_STAMP(1);
    typedef const InstanceBlock __attribute__((address_space(1))) *GlobalInstance;  // (global, not flat, loads)
    const auto &ib = ((GlobalInstance)draw.instances)[inst];
    // The three 44-byte vertices as few, wide loads (a non-indexed triangle is 132 contiguous bytes: nine
    // dwordx4/x3/x2 loads instead of 33 dword loads).  Lanes are 132 bytes apart, so every load instruction touches
    // ~64 cache lines and the L1's tag rate, not HBM, bounds this phase: fewer instructions is what counts.
    // (One shape of loads for indexed and non-indexed meshes -- three 44-byte vertices at three indices, through pointers the
    //  compiler knows to be global memory: left as an if / else of two copies into the same array it merged them into 33
    //  flat dword loads with 33 selected addresses.)
    typedef const Vertex __attribute__((address_space(1))) *GlobalVertex;
    typedef const uint32_t __attribute__((address_space(1))) *GlobalIndex;
    uint32_t vi[3] = {3u * tri, 3u * tri + 1u, 3u * tri + 2u};
    if (draw.indices) {
      const GlobalIndex ix = (GlobalIndex)draw.indices + 3u * tri;
      vi[0] = ix[0]; vi[1] = ix[1]; vi[2] = ix[2];
    }
    const GlobalVertex gv = (GlobalVertex)draw.vertices;
    Vertex vtx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const GlobalVertex q = gv + vi[k];
#pragma unroll
      for (int c = 0; c < 3; ++c) { vtx[k].pos[c] = q->pos[c]; vtx[k].normal[c] = q->normal[c]; vtx[k].tangent[c] = q->tangent[c]; }
      vtx[k].uv[0] = q->uv[0]; vtx[k].uv[1] = q->uv[1];
    }
    // The upper 3 x 3 of the instance's inverse model matrix (the survivors' normal matrix; the overlay programs' colour /
    // view rows) is asked for in the SAME batch as the vertices and the model matrix: every lane of an instance reads the
    // same 128 bytes (L1 hits), and asked for only behind the cull it was one more dependent round trip in a kernel that is
    // nothing but a chain of them.  The fence keeps the compiler from sinking the loads back to their first use.
    float im_[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) im_[r][cc] = ib.inv_model.M[r][cc];
    Mat4 model;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) model.M[r][cc] = ib.model.M[r][cc];
    asm volatile("" :: "v"(im_[0][0]), "v"(im_[0][1]), "v"(im_[0][2]), "v"(im_[1][0]), "v"(im_[1][1]), "v"(im_[1][2]), "v"(im_[2][0]),
                 "v"(im_[2][1]), "v"(im_[2][2]), "v"(vtx[0].pos[0]), "v"(vtx[1].pos[0]), "v"(vtx[2].pos[0]), "v"(model.M[0][0]),
                 "v"(model.M[0][1]), "v"(model.M[0][2]), "v"(model.M[0][3]), "v"(model.M[1][0]), "v"(model.M[1][1]), "v"(model.M[1][2]),
                 "v"(model.M[1][3]), "v"(model.M[2][0]), "v"(model.M[2][1]), "v"(model.M[2][2]), "v"(model.M[2][3]), "v"(model.M[3][0]),
                 "v"(model.M[3][1]), "v"(model.M[3][2]), "v"(model.M[3][3]) : "memory");
#ifdef BB_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    BB_STAMP(6);
#endif
    // ---- positions first: gl_Position of the three vertices decides whether the primitive can touch a pixel at all.  Half
    // of a closed mesh faces away from the camera and is culled below; only the survivors pay for the rest of the vertex
    // stage (normal matrix, two normalisations and a cross product per vertex) and for their 224-byte record ----
    float pw[3][3];  // posWorld (main passes)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const Vertex &v = vtx[k];
      f4 c;
      if (OVERLAY) {
        // The host folds the matrices: ib.model = (P*V)*modelMat of the light (light.vert:11-14) or the gizmo's own
        // projMat*viewMat (gizmo.vert:13-24)
        c = mat4_mul(model, f4{v.pos[0], v.pos[1], v.pos[2], 1.0f});
        pw[k][0] = pw[k][1] = pw[k][2] = 0.0f;
      } else {
        // forward_brdf.vert:25,27
        const f4 w = mat4_mul(model, f4{v.pos[0], v.pos[1], v.pos[2], 1.0f});
        // forward_brdf.vert:27 multiplies (P*V) * posWorld (pv = P*V); gbuffer.vert:19-22 P * (V * posWorld) (pv = P)
        c = fp.deferred ? mat4_mul(pv, mat4_mul(view, w)) : mat4_mul(pv, w);
        pw[k][0] = w.x; pw[k][1] = w.y; pw[k][2] = w.z;
      }
      clip[k][0] = c.x; clip[k][1] = c.y; clip[k][2] = c.z; clip[k][3] = c.w;
    }
    if (OVERLAY && prim >= fp.ov_first_gizmo_prim) vp = Viewport{fp.ov_half, fp.ov_half, fp.ov_cx, fp.ov_cy};
    // trivial reject against the true frustum (cannot change any pixel)
    bool o_l = true, o_r = true, o_t = true, o_b = true, o_n = true, o_f = true, all_in = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float *c = clip[i];
      o_l &= (c[3] + c[0] < 0.0f); o_r &= (c[3] - c[0] < 0.0f);
      o_t &= (c[3] + c[1] < 0.0f); o_b &= (c[3] - c[1] < 0.0f);
      o_n &= (c[3] - c[2] < 0.0f); o_f &= (c[2] < 0.0f);
#pragma unroll
      for (int p = 0; p < 6; ++p) all_in &= (plane_dist(c, p) >= 0.0f);
    }
    RasterTri t = {};
    bool survives = false;  // unclipped, front-facing, holds a pixel centre (of this rank's bands)
    if (!(o_l | o_r | o_t | o_b | o_n | o_f)) {
      if (all_in) {
        float z0, z1, z2;
        if (project_vertex(clip[0], vp, t.X0, t.Y0, t.rw0, z0) && project_vertex(clip[1], vp, t.X1, t.Y1, t.rw1, z1) &&
            project_vertex(clip[2], vp, t.X2, t.Y2, t.rw2, z2) && setup_tri(t, z0, z1, z2) &&
            tile_range<TILE_W, TILE_H>(t, fp, tr)) {
          survives = true;
          if (fp.world > 1) {
            // screen-band partition: a primitive whose tile rows hold none of this rank's bands (band b belongs to rank
            // b mod world) is somebody else's: no record, no bin entry.  (Every rank still runs the vertex positions of
            // every primitive -- that is what tells it whose they are.)
            const int b0 = tr.ty0 / fp.band_tiles, b1 = tr.ty1 / fp.band_tiles;
            const int first = b0 + ((fp.rank - b0 % fp.world) + fp.world) % fp.world;
            survives = first <= b1;
          }
        }
      } else if (!BB_ABLATE(128u)) {
        needs_clip = true;
      }
    }
#ifdef BB_STAMPS
    BB_STAMP(7);
#endif
    if (survives || needs_clip) {
      // ---- the rest of the vertex stage and the primitive's record ----
      ShadeRec pa;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const Vertex &v = vtx[k];
        float o[kNumVary];
        if (OVERLAY) {
          // ib.inv_model row 0 = the light's colour, or the gizmo's viewMat whose upper 3x3 turns the normals
          // (gizmo.vert:27).  draw.material is the program: 1 marker, 2 gizmo.
#pragma unroll
          for (int j = 0; j < kNumVary; ++j) o[j] = 0.0f;
          if (draw.material == 1u) {
            o[0] = im_[0][0]; o[1] = im_[0][1]; o[2] = im_[0][2];
          } else {
            f3 n = ld3(v.normal);
            o[0] = v.tangent[0]; o[1] = v.tangent[1]; o[2] = v.tangent[2];  // the gizmo mesh keeps its colour there
            o[3] = fmaf(im_[2][0], n.z, fmaf(im_[1][0], n.y, im_[0][0] * n.x));
            o[4] = fmaf(im_[2][1], n.z, fmaf(im_[1][1], n.y, im_[0][1] * n.x));
            o[5] = fmaf(im_[2][2], n.z, fmaf(im_[1][2], n.y, im_[0][2] * n.x));
          }
        } else {
          // :31-36  normalMat = transpose(mat3(aInvModel))
          f3 n = ld3(v.normal), tg = ld3(v.tangent);
          const f3 im0 = mk3(im_[0][0], im_[0][1], im_[0][2]), im1 = mk3(im_[1][0], im_[1][1], im_[1][2]),
                   im2 = mk3(im_[2][0], im_[2][1], im_[2][2]);
          f3 N = normalize3(mk3(dot3(im0, n), dot3(im1, n), dot3(im2, n)));
          f3 T = normalize3(mk3(dot3(im0, tg), dot3(im1, tg), dot3(im2, tg)));
          f3 B = cross3(N, T);
          o[0] = v.uv[0]; o[1] = v.uv[1];
          o[2] = pw[k][0]; o[3] = pw[k][1]; o[4] = pw[k][2];
          o[5] = N.x; o[6] = N.y; o[7] = N.z;
          o[8] = T.x; o[9] = T.y; o[10] = T.z;
          o[11] = B.x; o[12] = B.y; o[13] = B.z;
        }
        // varying j of vertex k: 0, 1 in the record's head (the texture coordinates), the rest in its body
        pa.uv[k][0] = o[0]; pa.uv[k][1] = o[1];
#pragma unroll
        for (int j = 0; j < kNumBodyVary; ++j) pa.vary[j][k] = o[2 + j];
      }
      pa.material = draw.material;  // (overlay pass: the overlay program, not an index into the material table)
      pa.packed = nullptr;
      pa.packed_dims = 0u;
      if (!OVERLAY) {
        // (the material's packed form travels in the draw descriptor: no load of the material table here)
        pa.packed = draw.packed;
        pa.packed_dims = draw.packed ? draw.packed_dims : 0u;
      }
      pa.clip_base = kNotClipped;  // (a clipped primitive's is patched in once the clipper has its arena slots, below)
      // planes of the unclipped triangle; zero for a primitive that goes through the clipper (its sub-triangles have their own)
      pa.h = PlaneHead{t.X0, t.Y0, t.l1dx, t.l1dy, t.l2dx, t.l2dy, t.rw0, t.rw1, t.rw2};
      if (needs_clip) pa.h = PlaneHead{0, 0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      if (!BB_ABLATE(64u)) {
        recs[prim] = pa;
        if (survives) tris[prim] = t;
      }
      survives_out = survives;
      if (survives) {
        uint32_t ntiles = (uint32_t)(tr.tx1 - tr.tx0 + 1) * (uint32_t)(tr.ty1 - tr.ty0 + 1);
        if (ntiles > fp.broad_threshold) broad_insert(t, prim << 3, fp, ctr, broad_list);
        else binned = true;
        cls = raster_class(t, fp);
      }
    }
  }
  BB_STAMP(2);
  // ---- tile bins: walk each lane's tile range in lock-step, aggregating per tile across the wave; four insertions
  // per lane are reserved back to back, so their returning atomics share one memory round trip ----
  uint32_t refs = 0;
  {
    const int tw = binned ? tr.tx1 - tr.tx0 + 1 : 0;
    const int nt = (binned && !BB_ABLATE(32u)) ? tw * (tr.ty1 - tr.ty0 + 1) : 0;
    for (int k0 = 0; __ballot(k0 < nt) != 0ull; k0 += 4) {
      bool has[4];
      uint32_t seg[4];
      BinTicket tk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + j;
        has[j] = k < nt;
        const int ty = has[j] ? tr.ty0 + k / tw : 0, tx = has[j] ? tr.tx0 + k % tw : 0;
        if (has[j] && fp.world > 1 && ((ty / fp.band_tiles) % fp.world) != fp.rank) has[j] = false;  // another rank's band
        seg[j] = ((uint32_t)ty * (uint32_t)fp.tiles_x + (uint32_t)tx) * kBinClasses + cls;
        tk[j] = wave_bin_reserve(has[j], seg[j], tile_count);
        refs += (uint32_t)__popcll(__ballot(has[j]));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) wave_bin_write(has[j], seg[j], prim << 3, tk[j], fp, ctr, bins, (OVERLAY || !fp.heavy_threshold) ? nullptr : heavy);
    }
  }
  BB_STAMP(3);
  // ---- clipper (after the bins, so the other lanes' insertions never wait for it): the wave clips its clipped
  // primitives one after the other, all lanes on the same polygon ----
  for (unsigned long long cm = __ballot(needs_clip); cm; cm &= cm - 1ull) {
    const int owner = __builtin_amdgcn_readfirstlane(__ffsll((long long)cm) - 1);
    ClipWork &w = s_clip[threadIdx.x >> 6];
    // the owner's clip-space vertices and primitive index are the wave's input
    float cv[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) cv[i][k] = clip[i][k];
    Viewport ovp = vp;  // the owner's viewport for the whole wave
    if (OVERLAY) {
      ovp.half_w = __shfl(vp.half_w, owner); ovp.half_h = __shfl(vp.half_h, owner);
      ovp.cx = __shfl(vp.cx, owner); ovp.cy = __shfl(vp.cy, owner);
    }
    clip_primitive_wave<TILE_W, TILE_H>(w, owner, cv, prim, fp, ovp, clip_arena, ctr, broad_list);
    __builtin_amdgcn_wave_barrier();
    clipped_raster += (uint32_t)w.n_valid;
    if ((int)(threadIdx.x & 63) == owner && w.n_valid) recs[prim].clip_base = w.base;  // (the record itself was written above, with zero planes)
    __builtin_amdgcn_wave_barrier();
  }
  BB_STAMP(4);
  // statistics leave the kernel as one record per WAVE (summed on the host on demand): a few thousand atomics on ONE
  // counter word would serialise at ~90 per microsecond and dominate this kernel
  {
    const uint32_t n_survivors = (uint32_t)__popcll(__ballot(survives_out));
    const uint32_t n_clipped = (uint32_t)__popcll(__ballot(needs_clip));
    if ((threadIdx.x & 63) == 0)
      block_stats[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = BlockStats{n_survivors + clipped_raster, n_clipped, refs};
  }
  BB_STAMP(5);
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

template <bool BLOCKED = false>
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
  if (BLOCKED) {
    // the packed material is stored block-linear, 4 x 4 texels per 144-byte block (written by bbr_upload_material):
    // texel (x, y) = record ((y >> 2) * ceil(w / 4) + (x >> 2)) * 16 + (y & 3) * 4 + (x & 3).  A minified tap set touches
    // one or two 128-byte lines instead of always two rows 9 w bytes apart.
    const int w4 = (w + 3) >> 2;
    const uint32_t row0 = ((uint32_t)mul32(y0 >> 2, w4) << 4) + (((uint32_t)y0 & 3u) << 2);
    const uint32_t row1 = ((uint32_t)mul32(y1 >> 2, w4) << 4) + (((uint32_t)y1 & 3u) << 2);
    const uint32_t col0 = (((uint32_t)x0 >> 2) << 4) + ((uint32_t)x0 & 3u), col1 = (((uint32_t)x1 >> 2) << 4) + ((uint32_t)x1 & 3u);
    t.o00 = row0 + col0;
    t.o10 = row0 + col1;
    t.o01 = row1 + col0;
    t.o11 = row1 + col1;
    return t;
  }
  const uint32_t row0 = (uint32_t)mul32(y0, w), row1 = (uint32_t)mul32(y1, w);
  t.o00 = row0 + (uint32_t)x0;
  t.o10 = row0 + (uint32_t)x1;
  t.o01 = row1 + (uint32_t)x0;
  t.o11 = row1 + (uint32_t)x1;
  return t;
}

// byte k of a dword as a float: v_cvt_f32_ubyteN.  Spelled out, so that the compiler keeps the conversion: left to
// itself it rewrites float(b) - float(a) as float(int(b) - int(a)) -- v_sub_u32_sdwa + v_cvt_f32_i32, two of the expensive
// (non-fma-class) instructions where the float subtraction is a cheap one; 18 more of them per pixel.
// byte offset of packed texel i (9-byte records): i * 8 + i as one v_lshl_add_u32, not the v_mad_i32_i24 the compiler picks
BB_DEV uint32_t texel_offset(uint32_t i) {
  static_assert(kPackedTexelBytes == 9, "texel_offset");
  asm("" : "+v"(i));
  return (i << 3) + i;
}
BB_DEV float byte_f32(uint32_t t, int shift) {
  float f;
  if (shift == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(t));
  else if (shift == 8) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(t));
  else if (shift == 16) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(t));
  else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(t));
  return f;
}
BB_DEV float filter_channel(uint32_t t00, uint32_t t10, uint32_t t01, uint32_t t11, int shift, float fx, float fy) {
  float a = byte_f32(t00, shift), b = byte_f32(t10, shift);
  float c = byte_f32(t01, shift), d = byte_f32(t11, shift);
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
  return a2 * bb_rcp(denom);
}

// roughness comes out of filter_channel (a bilinear blend of bytes / 255, possibly rounded to binary16): it is in
// [0, 1], so k = (r + 1)^2 / 8 is in [0.125, 0.5]; NdotX = max0(dot of two vectors that are unit length, zero or NaN) is
// in [0, 1 + 2^-20] (max0 removes the NaN).  Hence denom is in [0.125, 1.01]: a normal number, no guard needed.
BB_DEV float geometry_schlick_ggx(float NdotX, float k) {
  float denom = fmaf(NdotX, 1.0f - k, k);
  return NdotX * bb_rcp_normal(denom);
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
  int32_t tone_enable;  // fused presentation only (option "present_fused"): FrameUniformBlock.EnableToneMapping / Exposure
  float exposure;
};

// pixel index inside the tile, 8x8-blocked so that 64 consecutive indices form one 8x8 pixel block
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

// A triangle against a rectangle of pixel centres [x0,x1] x [y0,y1]: 0 = no centre covered, 1 = some, 2 = all.
// Edge functions are affine, so their extrema over the rectangle sit at corners picked by the gradient signs.
struct EdgeSetup {
  int dx[3], dy[3], bias[3];
  int X[3], Y[3];
};

BB_DEV EdgeSetup edge_setup(const RasterTri &t) {
  EdgeSetup e;
  e.X[0] = t.X0; e.Y[0] = t.Y0; e.X[1] = t.X1; e.Y[1] = t.Y1; e.X[2] = t.X2; e.Y[2] = t.Y2;
  e.dx[0] = t.X1 - t.X0; e.dy[0] = t.Y1 - t.Y0;  // |coordinates| < 2^30 (project_vertex), differences fit 32 bits
  e.dx[1] = t.X2 - t.X1; e.dy[1] = t.Y2 - t.Y1;
  e.dx[2] = t.X0 - t.X2; e.dy[2] = t.Y0 - t.Y2;
#pragma unroll
  for (int i = 0; i < 3; ++i) e.bias[i] = (e.dy[i] < 0 || (e.dy[i] == 0 && e.dx[i] > 0)) ? 0 : -1;  // top-left rule
  return e;
}

// exact: 32x32 -> 64-bit products (v_mad_i64_i32)
BB_DEV long long edge_eval(const EdgeSetup &e, int i, int px, int py) {
  int Xc = px * 256 + 128, Yc = py * 256 + 128;
  return (long long)e.dx[i] * (long long)(Yc - e.Y[i]) - (long long)e.dy[i] * (long long)(Xc - e.X[i]) + (long long)e.bias[i];
}

BB_DEV int classify_rect(const EdgeSetup &e, int x0, int x1, int y0, int y1) {
  bool all_in = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    // dE/dx = -dy*256, dE/dy = +dx*256
    int xmax = e.dy[i] <= 0 ? x1 : x0, xmin = e.dy[i] <= 0 ? x0 : x1;
    int ymax = e.dx[i] >= 0 ? y1 : y0, ymin = e.dx[i] >= 0 ? y0 : y1;
    if (edge_eval(e, i, xmax, ymax) < 0) return 0;
    all_in &= edge_eval(e, i, xmin, ymin) >= 0;
  }
  return all_in ? 2 : 1;
}

// zbias (overlay pass only): added to the depth BITS of the key, order-preserving -- 0x40000000 lifts the gizmo's
// fragments above every depth in [0, 1] drawn before (the reference clears the rectangle's depth before the gizmo)
// while they still compete among themselves by depth.
BB_DEV void depth_max(const RasterTri &t, int px, int py, uint32_t ref, unsigned long long *keys, int key_index,
                      uint32_t zbias = 0u) {
  int Xc = px * 256 + 128, Yc = py * 256 + 128;
  float dxp = (float)(Xc - t.X0), dyp = (float)(Yc - t.Y0);
  float z = fmaf(t.dzdx, dxp, fmaf(t.dzdy, dyp, t.z0));
  if (!(z >= 0.0f)) z = 0.0f;
  if (z > 1.0f) z = 1.0f;
  unsigned long long key = ((unsigned long long)(__float_as_uint(z) + zbias) << 32) | (unsigned long long)(ref + 1u);
  atomicMax(&keys[key_index], key);
}

// One wave rasterises one (wave-uniform) triangle into the tile's LDS keys: 8x8 pixel blocks of bounding box ^ tile,
// one pixel per lane.  Rectangles fully inside the triangle skip the edge tests.  Triangles spanning <= 64 px use
// 32-bit edge functions stepped with 24-bit multiply-adds (exact: every term < 2^30); larger ones use 64-bit products.
template <int TILE_W, int TILE_H>
BB_DEV void raster_triangle_wave(const RasterTri &t, uint32_t ref, int tile_x0, int tile_y0, const FrameParams &fp,
                                 unsigned long long *keys, int lane, uint32_t zbias = 0u, int sx0 = 0, int sy0 = 0,
                                 int sx1 = 1 << 30, int sy1 = 1 << 30) {
  int32_t minX = min(t.X0, min(t.X1, t.X2)), maxX = max(t.X0, max(t.X1, t.X2));
  int32_t minY = min(t.Y0, min(t.Y1, t.Y2)), maxY = max(t.Y0, max(t.Y1, t.Y2));
  int px0 = max(max((minX - 128 + 255) >> 8, sx0), tile_x0);
  int px1 = min(min((maxX - 128) >> 8, min(fp.width, sx1) - 1), tile_x0 + TILE_W - 1);
  int py0 = max(max((minY - 128 + 255) >> 8, sy0), tile_y0);
  int py1 = min(min((maxY - 128) >> 8, min(fp.height, sy1) - 1), tile_y0 + TILE_H - 1);
  if (px0 > px1 || py0 > py1) return;
  const EdgeSetup e = edge_setup(t);
  const int w = px1 - px0 + 1, h = py1 - py0 + 1;
  const int lx = lane & 7, ly = lane >> 3;
  if (max(maxX - minX, maxY - minY) <= (1 << 14)) {
    const int Xc0 = px0 * 256 + 128, Yc0 = py0 * 256 + 128;
    int o[3], sx[3], sy[3];
    bool none = false, all = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      o[i] = e.dx[i] * (Yc0 - e.Y[i]) - e.dy[i] * (Xc0 - e.X[i]) + e.bias[i];
      sx[i] = -e.dy[i] * 256;
      sy[i] = e.dx[i] * 256;
      int hi = o[i] + (sx[i] > 0 ? sx[i] * (w - 1) : 0) + (sy[i] > 0 ? sy[i] * (h - 1) : 0);
      int lo = o[i] + (sx[i] < 0 ? sx[i] * (w - 1) : 0) + (sy[i] < 0 ? sy[i] * (h - 1) : 0);
      none |= hi < 0;
      all &= lo >= 0;
    }
    if (none) return;
    for (int by = 0; by < h; by += 8) {
      for (int bx = 0; bx < w; bx += 8) {
        int x = bx + lx, y = by + ly;
        bool in = x < w && y < h;
        if (in && !all) {
          int e0 = mul32(x, sx[0]) + mul32(y, sy[0]) + o[0];
          int e1 = mul32(x, sx[1]) + mul32(y, sy[1]) + o[1];
          int e2 = mul32(x, sx[2]) + mul32(y, sy[2]) + o[2];
          in = (e0 | e1 | e2) >= 0;
        }
        if (in) depth_max(t, px0 + x, py0 + y, ref, keys, tile_index<TILE_W>(px0 + x - tile_x0, py0 + y - tile_y0), zbias);
      }
    }
    return;
  }
  const int cls = classify_rect(e, px0, px1, py0, py1);
  if (cls == 0) return;
  for (int by = 0; by < h; by += 8) {
    for (int bx = 0; bx < w; bx += 8) {
      int px = px0 + bx + lx, py = py0 + by + ly;
      bool in = px <= px1 && py <= py1;
      if (in && cls == 1) in = (edge_eval(e, 0, px, py) | edge_eval(e, 1, px, py) | edge_eval(e, 2, px, py)) >= 0;
      if (in) depth_max(t, px, py, ref, keys, tile_index<TILE_W>(px - tile_x0, py - tile_y0), zbias);
    }
  }
}

constexpr int kItemsThreads = 256;  // launch slots per workgroup of k_shade_items
// words between two of k_raster's chunk counters: a 128-byte cache line each.  Atomics on one LINE serialise like atomics on
// one address (packed 32 to a line they cost k_raster 5 us at C5's 32 640 tiles)
constexpr int kItemGroupStride = 32;
constexpr int kItemGroupSlots = 32, kItemGroups = 2048;  // k_raster's chunk totals: one counter per 32 launch slots (fewer
                                                          // addresses made the atomics cost k_raster 7 us at C3)
constexpr int kItemChunkBits = 6, kItemTxBits = 12;  // item = full << 31 | grid row << 18 | tile column << 6 | chunk of 64 fragments
// A tile that ONE triangle covers completely and nothing else touches has no fragment list: its count word carries this flag
// and the tile's list holds a single word (pixel field 0); fragment p of the tile is that word with p in the pixel field.
// k_shade_items hands the flag on in the item word, so that k_shade knows before it loads anything.
constexpr uint32_t kFullTile = 0x80000000u;
constexpr int kFragPixBits = 12;  // pixel-in-tile field of a fragment's high word (tiles of up to 64x64); the clip slot + 1 sits above it
constexpr int kTileThreads = 256;
constexpr int kTileWaves = kTileThreads / 64;

// ONE ROW of a small triangle (raster classes 0 and 1: spans <= 64 px, so every edge-function term fits 32 bits and a
// step is a plain add) by one lane: the row's pixels of bounding box ^ tile, left to right.  k_raster hands the rows of all
// small triangles of a chunk out as one list (see there), so a lane's work is a row whatever the triangle's size.
// Same values as the 64-bit form (edge_eval) at every pixel centre: E' = E + (top-left ? 0 : -1), covered <=> all E' >= 0.
// The depth is depth_max's expression with its row-constant half hoisted: z = fma(dzdx, dxp, fma(dzdy, dyp, z0)), and
// dxp = float(Xc - X0) is stepped by 256.0f (exact: |Xc - X0| <= 2^14 + 2^8 inside the bounding box).
template <int TILE_W, int TILE_H>
BB_DEV void raster_triangle_row(int X0, int Y0, int X1, int Y1, int X2, int Y2, float z0, float dzdx, float dzdy, uint32_t ref,
                                int px0, int px1, int py, int tile_x0, int tile_y0, unsigned long long *keys, uint32_t zbias) {
  const int dx0 = X1 - X0, dy0 = Y1 - Y0;
  const int dx1 = X2 - X1, dy1 = Y2 - Y1;
  const int dx2 = X0 - X2, dy2 = Y0 - Y2;
  const int Xc0 = px0 * 256 + 128, Yc = py * 256 + 128;
  int e0 = mul32(dx0, Yc - Y0) - mul32(dy0, Xc0 - X0) + ((dy0 < 0 || (dy0 == 0 && dx0 > 0)) ? 0 : -1);
  int e1 = mul32(dx1, Yc - Y1) - mul32(dy1, Xc0 - X1) + ((dy1 < 0 || (dy1 == 0 && dx1 > 0)) ? 0 : -1);
  int e2 = mul32(dx2, Yc - Y2) - mul32(dy2, Xc0 - X2) + ((dy2 < 0 || (dy2 == 0 && dx2 > 0)) ? 0 : -1);
  const int sx0 = dy0 * 256, sx1 = dy1 * 256, sx2 = dy2 * 256;  // E(x+1) = E - sx
  const float zrow = fmaf(dzdy, (float)(Yc - Y0), z0);
  float dxp = (float)(Xc0 - X0);
  const int y = py - tile_y0;
  constexpr int BX = TILE_W / 8;
  const int row_index = (((y >> 3) * BX) << 6) | ((y & 7) << 3);  // tile_index<TILE_W>(x, y) = row_index + (x >> 3 << 6 | x & 7)
  const unsigned long long key_lo = (unsigned long long)(ref + 1u);
  // (Two variations of this loop were built, bit-exact, and measured slower in round 5 -- the boxes of small triangles are
  //  4 (class 0) / 10 (class 1) pixels wide and 30 % covered at C3, tools/raster_stats.py: (a) bounding the row's span first,
  //  x <= e_i / sx_i per edge from binary32 quotients widened by 0.001 px, the exact test still deciding: 48 instructions per
  //  row, k_raster's VALU count 12.9 M -> 14.5 M; (b) two passes, a coverage bit per pixel and then the depth code over the set
  //  bits only: 15.2 M, k_raster alone 45.6 -> 47.6 us.)
  for (int x = px0 - tile_x0; x <= px1 - tile_x0; ++x) {
    if ((e0 | e1 | e2) >= 0) {
      float z = fmaf(dzdx, dxp, zrow);
      if (!(z >= 0.0f)) z = 0.0f;
      if (z > 1.0f) z = 1.0f;
      atomicMax(&keys[row_index + (((x >> 3) << 6) | (x & 7))], ((unsigned long long)(__float_as_uint(z) + zbias) << 32) | key_lo);
    }
    e0 -= sx0; e1 -= sx1; e2 -= sx2;
    dxp += 256.0f;
  }
}

// owned tile row (grid y) -> global tile row; false if past the frame
BB_DEV bool tile_row(const FrameParams &fp, int grid_y, int &ty, int &out_tile_row) {
  if (fp.world > 1) {
    int lb = grid_y / fp.band_tiles, r = grid_y - lb * fp.band_tiles;
    ty = (lb * fp.world + fp.rank) * fp.band_tiles + r;
  } else {
    ty = grid_y;
  }
  out_tile_row = grid_y;
  return ty < fp.tiles_y;
}

// ------------------------------------------------------------------------------------------------
// k_raster: one workgroup per screen tile.  LDS-resident 64-bit keys (depth bits << 32 | primitive) filled with
// ds_max_u64 -- depth op GREATER_OR_EQUAL with "later primitive wins ties" falls out of the key order -- then
// ballot/popcount compaction of the covered pixels into the tile's fragment list.  Background pixels get the
// clear colour here; covered pixels are coloured by k_shade.
//
// The kernel is latency-bound (bin -> triangle record -> LDS atomics), so the triangle data of up to 256 bin
// entries is staged through LDS by all threads at once: two memory round trips per chunk however the entries
// split over the raster classes, instead of two per loop iteration.  The every-tile list rides along as extra
// "large" entries after a per-tile accept/reject test.
// ------------------------------------------------------------------------------------------------
constexpr int kStage = kTileThreads;  // staged entries per chunk (one per thread)
constexpr uint32_t kBroadSpec = 16;    // every-tile-list entries a light tile fetches before it knows the count

struct StagedTri {  // struct-of-arrays in LDS: thread j owns column j when filling
  int X0[kStage], Y0[kStage], X1[kStage], Y1[kStage], X2[kStage], Y2[kStage];
  float z0[kStage], dzdx[kStage], dzdy[kStage];
  uint32_t ref[kStage];
  uint32_t box[kStage];  // px0 | px1 << 8 | py0 << 16 | py1 << 24 relative to the tile; 0xFFFFFFFF = skip
};

BB_DEV RasterTri staged_tri(const StagedTri &st, int j) {
  RasterTri t;
  t.X0 = st.X0[j]; t.Y0 = st.Y0[j]; t.X1 = st.X1[j]; t.Y1 = st.Y1[j]; t.X2 = st.X2[j]; t.Y2 = st.Y2[j];
  t.z0 = st.z0[j]; t.dzdx = st.dzdx[j]; t.dzdy = st.dzdy[j];
  t.l1dx = t.l1dy = t.l2dx = t.l2dy = t.rw0 = t.rw1 = t.rw2 = 0.0f;  // not needed for coverage / depth
  return t;
}

// One light as the loop consumes it: 48 bytes, written once per frame by cook_light.
struct CookedLight {
  float px, py, pz;
  int32_t type;
  float cr, cg, cb;  // color * intensity
  float outer;
  float dx, dy, dz;  // type 1: normalize(-dir); type 2: -normalize(dir)
  float inv_eps;     // type 1: 1 / (inner - outer)
};
struct ShadeShared {
  CookedLight lights[kMaxNumLights];
};

BB_DEV CookedLight cook_light(const Light &l) {
  CookedLight c;
  c.px = l.pos[0]; c.py = l.pos[1]; c.pz = l.pos[2];
  c.type = l.type;
  c.cr = l.color[0] * l.intensity; c.cg = l.color[1] * l.intensity; c.cb = l.color[2] * l.intensity;
  c.outer = l.outer_cutoff;
  c.dx = c.dy = c.dz = 0.0f;
  c.inv_eps = 0.0f;
  if (l.type == 1) {
    const f3 d = normalize3(neg3(ld3(l.dir)));
    c.dx = d.x; c.dy = d.y; c.dz = d.z;
    c.inv_eps = bb_rcp(l.inner_cutoff - l.outer_cutoff);
  } else if (l.type == 2) {
    const f3 d = neg3(normalize3(ld3(l.dir)));
    c.dx = d.x; c.dy = d.y; c.dz = d.z;
  }
  return c;
}

// OVERLAY = true (overlay subpass): the keys start from the scene's resolved depth (`depth_io`, read) instead of 0,
// gizmo primitives are scissored to their rectangle and biased above everything else, pixels no overlay primitive
// wins are left alone (no background fill).  OVERLAY = false with depth_io != nullptr stores the resolved depth.
#ifndef BB_RASTER_WAVES
#define BB_RASTER_WAVES 8  // waves per SIMD the 32 x 32 instantiation is compiled for (64 registers; 20.3 KB of LDS allow eight
                           // workgroups per CU).  64 x 64 tiles hold 32 KB of keys: three.
#endif
#ifndef BB_RASTER_PREFETCH
#define BB_RASTER_PREFETCH 1
#endif
template <int TILE_W, int TILE_H, bool OVERLAY = false>
__global__ __launch_bounds__(kTileThreads) __attribute__((amdgpu_waves_per_eu(TILE_W > 32 ? 3 : (OVERLAY ? 7 : BB_RASTER_WAVES)))) void k_raster(
    // (first: what a tile needs before its first load -- these arrive in scalar registers with the wave, "kernarg preload";
    //  the Makefile asks for it.  Everything behind them comes with one scalar load from the kernel-argument segment.)
    uint32_t *__restrict__ tile_count, Counters *__restrict__ ctr, const BroadTri *__restrict__ broad_list,
    uint32_t *__restrict__ frag_count, unsigned long long *__restrict__ frags, float4 *__restrict__ out,
    FrameParams fp, const RasterTri *__restrict__ tris, const ClipSlot *__restrict__ clip_arena,
    const uint32_t *__restrict__ bins,
    uint32_t *__restrict__ vis_prim, float *__restrict__ vis_depth,
    const float4 *__restrict__ background, float *__restrict__ depth_io, uint32_t *__restrict__ host_flags,
    uint32_t *__restrict__ out8, uint32_t *__restrict__ item_groups, uint32_t *__restrict__ item_head,
    uint32_t *__restrict__ items, const Light *__restrict__ lights, int num_lights, CookedLight *__restrict__ cooked,
    const uint32_t *__restrict__ heavy) {
  constexpr int TILE_PIXELS = TILE_W * TILE_H;
  __shared__ unsigned long long keys[TILE_PIXELS];
  __shared__ StagedTri st;
  __shared__ uint32_t s_count;
  // the row list of the chunk's small triangles (see the chunk loop)
  __shared__ uint16_t s_row0[kStage];          // first row of staged entry j in its wave's list
  __shared__ uint32_t s_wave_rows[kTileWaves];  // rows in each wave's list

  __builtin_amdgcn_s_setprio(3);  // (as k_geometry: a latency-bound wave's instruction goes out when it is ready)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // {overflow bits, bin_need, -, every-tile entries asked for, clip slots asked for} of this frame straight into pinned host
  // memory (final since k_geometry ended): the host looks at them when it reuses the frame's slot -- a few stores instead
  // of a copy kernel on the stream.  (Word 2 is the frame's shade item count, stored by k_shade_items.)
  if (host_flags && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    host_flags[0] = ctr->overflow;
    host_flags[1] = ctr->bin_need;
    host_flags[3] = ctr->n_broad;
    host_flags[4] = ctr->n_clip_slots;
    host_flags[5] = ctr->n_heavy;   // sizes the heavy rows of this slot's next frame
  }
  // SHORT FRAMES (item_head != nullptr; the host decides by the frame's tile count): there is no k_shade_items launch.  The
  // tiles append their items to the frame's list themselves -- one returning atomic per tile on the list's head word, which
  // the frame's staging copy has zeroed -- and this kernel's first workgroup cooks the light table.  A 1080p frame is a
  // chain of dependent kernels whose length over the frames in flight IS its rate; the scan kernel and its boundary were
  // 9 of its 90 us.  The list is then in tile COMPLETION order, not screen order: at 4K that cost 13 % more texel traffic
  // than it saved (round 2), which is why long frames keep k_shade_items.
  if (item_head && cooked && blockIdx.x == 0 && blockIdx.y == 0)
    for (int li = tid; li < num_lights; li += kTileThreads) cooked[li] = cook_light(lights[li]);
  // (called by the first wave of the workgroup, all of its lanes)
  auto append_items = [&](uint32_t chunks, uint32_t flag) {
    uint32_t at = 0u;
    if (lane == 0) at = atomicAdd(item_head, chunks);
    at = (uint32_t)__shfl((int)at, 0);
    if ((uint32_t)lane < chunks)
      items[1u + at + (uint32_t)lane] = flag | ((uint32_t)blockIdx.y << (kItemChunkBits + kItemTxBits)) | ((uint32_t)blockIdx.x << kItemChunkBits) | (uint32_t)lane;
  };  // (short frames only: they have no heavy rows, blockIdx is the tile's launch slot)
  static_assert(TILE_PIXELS / 64 <= 64, "a tile's items are written by one wave");
  // launch slot -> tile: plain row order ...
  uint32_t slot = blockIdx.y * gridDim.x + blockIdx.x;
  // ... behind fp.heavy_rows rows of HEAVY SLOTS (long frames only).  A tile's life is 1.6 us when nothing is binned to it and
  // 14-39 us when a ball is; in plain screen order the last heavy tile starts when the kernel's work is half done and the
  // launch ends in a tail as long as that tile.  k_geometry lists the tiles whose bins reach fp.heavy_threshold references
  // (launch slot | class << 30, at most one entry per class and tile); the workgroups of the first rows take one entry each
  // and rasterise THAT tile, and the tile's own workgroup, which sees the same final counts, leaves at once.  Which entry
  // of a tile listed twice does the work follows from the counts too (the highest class at the threshold), so nobody has to
  // claim anything.  The rows are sized by the host from the list length of the slot's previous frame; a list longer than
  // the rows is not used at all -- every decision is all-or-nothing on values that are final since k_geometry ended.
  const uint32_t heavy_slots = (uint32_t)fp.heavy_rows * gridDim.x;
  bool heavy_slot = false;
  uint32_t heavy_class = 0u;
  if (!OVERLAY && fp.heavy_rows) {
    if (blockIdx.y < (uint32_t)fp.heavy_rows) {
      const uint32_t n_heavy = ctr->n_heavy;
      if (slot >= n_heavy || n_heavy > heavy_slots) return;
      const uint32_t e = heavy[slot];
      heavy_class = e >> 30;
      slot = e & 0x3FFFFFFFu;
      heavy_slot = true;
    } else {
      slot -= heavy_slots;
    }
  }
  const int grid_row = (int)(slot / gridDim.x);
  const int tx = (int)(slot - (uint32_t)grid_row * gridDim.x);
  int ty, out_tile_row;
  const bool live = tile_row(fp, grid_row, ty, out_tile_row);
  if (!live) return;
  const uint32_t tile = (uint32_t)ty * (uint32_t)fp.tiles_x + (uint32_t)tx;
#ifdef BB_STAMPS
#define BB_RSTAMP(i) do { if (tid == 0) reinterpret_cast<unsigned long long *>(frag_count + fp.tiles_x * fp.tiles_y)[tile * 8 + (i)] = wall_clock64(); } while (0)
#else
#define BB_RSTAMP(i) do { } while (0)
#endif
  BB_RSTAMP(0);
  const int tile_x0 = tx * TILE_W, tile_y0 = ty * TILE_H;
  const int out_y0 = out_tile_row * TILE_H;
  const int rx1 = min(tile_x0 + TILE_W, fp.width) - 1, ry1 = min(tile_y0 + TILE_H, fp.height) - 1;

  // Clipped sub-triangles that touch this tile: (reference, clip-arena slot + 1).  The slot goes into the fragment word,
  // so that k_shade can fetch the sub-triangle's planes together with the primitive record instead of after it.
  // More than kClipRefs of them: the field stays 0 and k_shade finds the slot through the record (one more round trip).
  constexpr uint32_t kClipRefs = 32;
  __shared__ uint32_t s_clip_ref[kClipRefs], s_clip_slot[kClipRefs], s_n_clip_refs;
  __shared__ unsigned long long s_pad_frag;
  // The tile's bin counts: every thread reads them itself (uniform address: one line), decides with them, and only AFTER
  // the workgroup's first barrier are they cleared for the slot's next frame -- a wave arrives at that barrier with its
  // loads returned, so no wave, however late it started, can see the cleared value.  (Cleared right after the read, a
  // late wave disagreed with its workgroup about the loop bounds: ~100 wrong pixels in one frame of a few hundred
  // whenever other processes shared the GPU.)
  uint32_t n_cls[kBinClasses];
  bool at_threshold[kBinClasses];
#pragma unroll
  for (uint32_t c = 0; c < kBinClasses; ++c) {
    const uint32_t raw = tile_count[tile * kBinClasses + c];
    at_threshold[c] = raw >= fp.heavy_threshold;
    n_cls[c] = BB_ABLATE(1u | (256u << c)) ? 0u : min(raw, fp.bin_cap);
  }
  // A frame whose every-tile list overflowed (bit 1 of ctr->overflow, final since k_geometry ended) is rendered again after
  // the host has grown the list: the clip path reserves a run of entries and writes none of them when the run does not
  // fit, so a prefix of the list may hold entries never written this frame -- the overflowed frame takes none of it.
  // (Both counter words are read unconditionally and combined without a branch -- ONE round trip, together with the bin
  //  counts above, in front of the tile's first decision.  As `overflow & 2 ? 0 : n_broad` the compiler made the second
  //  load wait for the first.)
  const uint32_t ctr_overflow = ctr->overflow, ctr_n_broad = ctr->n_broad, ctr_n_heavy = ctr->n_heavy;
  const uint32_t n_broad = BB_ABLATE(5u) ? 0u : (min(ctr_n_broad, fp.broad_cap) & (((ctr_overflow >> 1) & 1u) - 1u));
  if (!OVERLAY && fp.heavy_rows) {
    if (heavy_slot) {   // (the list is in use, or this workgroup had left above)
      const uint32_t top = at_threshold[2] ? 2u : (at_threshold[1] ? 1u : 0u);
      if (heavy_class != top) return;                          // the tile's other entry does it
    } else if ((at_threshold[0] || at_threshold[1] || at_threshold[2]) && ctr_n_heavy <= heavy_slots) {
      return;                                                  // a heavy slot does it (or has done it)
    }
  }

  // the pixel's colour when no geometry covers it.  forward: the clear colour (src/main.cpp:84); deferred: brdf.frag on the
  // cleared G-buffer texel (k_deferred_background); fused presentation: the same colours as presented pixels (the clear
  // colour presents as (0, 0, 0, 255))
  // (read ONCE, with the tile's first loads: inside store_background it was a scalar load and a wait per pixel pass)
  float4 bg = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  uint32_t bg8 = 0xFF000000u;
  if (background) {
    bg = background[0];
    bg8 = __float_as_uint(background[1].x);
    // (uniform values: kept in scalar registers -- five vector registers that would otherwise live through the whole chunk loop)
    bg.x = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(bg.x)));
    bg.y = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(bg.y)));
    bg.z = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(bg.z)));
    bg.w = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(bg.w)));
    bg8 = (uint32_t)__builtin_amdgcn_readfirstlane((int)bg8);
  }
  auto store_background = [&](int x, int y) {
    const size_t o = (size_t)(out_y0 + y) * (size_t)fp.width + (size_t)(tile_x0 + x);
    if (out8) store_pixel(&out8[o], bg8);
    else store_pixel(&out[o], bg);
  };

  // ---- light tiles: no LDS, no barrier ----
  // Most tiles of a frame hold no binned triangle at all: sky, or ground that ONE huge (every-tile-list) triangle covers
  // completely -- 6595 of C3's 8160.  Each of the four waves settles that by itself from the same uniform data (the bin
  // counts above and a classification of the <= 64 list entries, one per lane), so the workgroup agrees without talking:
  //   nothing touches the tile       -> its pixels get the background colour, its fragment count is 0
  //   one triangle covers all of it  -> every pixel's winner is known without a depth atomic: the fragment list is ONE
  //                                     word and the count carries kFullTile (k_shade makes the 64 fragment words of an
  //                                     item from the lane index)
  // and the workgroup is gone after two memory round trips.  Through the general path below these tiles lived 4.8 us each
  // (8 KB of keys cleared, a barrier, one staged entry, another barrier: in-kernel stamps) and held 28 % of the frame's
  // wave-slot time between them.  Anything else falls through.
  // (The first kBroadSpec entries of the every-tile list are fetched BEFORE the counts have arrived -- the list always has
  //  that many entries allocated, and zero-filled ones can touch no tile -- so that a light tile is ONE memory round trip:
  //  counts and entries come back together.  A frame with more entries fetches the rest behind the count.)
  BroadTri spec;
  if (!OVERLAY && (uint32_t)lane < kBroadSpec) spec = broad_list[lane];
  if (!OVERLAY && !vis_prim && !depth_io && (n_cls[0] | n_cls[1] | n_cls[2]) == 0u && n_broad <= 64u) {
    bool ok = false, full = false;
    uint32_t ref = 0u, clip_slot1 = 0u;
    if ((uint32_t)lane < n_broad) {
      BroadTri b;
      if ((uint32_t)lane < kBroadSpec) b = spec;
      else b = broad_list[lane];
      const RasterTri t = b.tri;
      ref = b.ref;
      clip_slot1 = b.pad[0];
      const int32_t minX = min(t.X0, min(t.X1, t.X2)), maxX = max(t.X0, max(t.X1, t.X2));
      const int32_t minY = min(t.Y0, min(t.Y1, t.Y2)), maxY = max(t.Y0, max(t.Y1, t.Y2));
      const int px0 = max((minX - 128 + 255) >> 8, tile_x0), px1 = min((maxX - 128) >> 8, rx1);
      const int py0 = max((minY - 128 + 255) >> 8, tile_y0), py1 = min((maxY - 128) >> 8, ry1);
      if (px0 <= px1 && py0 <= py1) {
        const int cls_rect = classify_rect(edge_setup(t), px0, px1, py0, py1);
        ok = cls_rect != 0;
        full = cls_rect == 2 && px1 - px0 + 1 == TILE_W && py1 - py0 + 1 == TILE_H;
      }
    }
    const unsigned long long m_ok = __ballot(ok);
    if (m_ok == 0ull) {
      // the same pixels in the same order as `for (p = tid; p < TILE_PIXELS; p += kTileThreads) tile_pixel(p)` would visit them,
      // without that function's arithmetic per pixel: pass k of thread tid is pixel (x, y + k * STEP) of the tile
      // (kTileThreads = 256 pixels = 4 blocks of 8 x 8 = a strip TILE_W wide or, for 64-pixel tiles, half of one)
      constexpr int BX = TILE_W / 8;                      // 8 x 8 blocks per row of blocks
      constexpr int PASSES = TILE_PIXELS / kTileThreads;
      static_assert((kTileThreads / 64) % BX == 0 || BX % (kTileThreads / 64) == 0, "fill pattern");
#pragma unroll
      for (int k = 0; k < PASSES; ++k) {
        const int block = (tid >> 6) + k * (kTileThreads / 64);
        const int x = (block % BX) * 8 + (tid & 7), y = (block / BX) * 8 + ((tid >> 3) & 7);
        if (tile_x0 + x < fp.width && tile_y0 + y < fp.height) store_background(x, y);
      }
      if (tid == 0) frag_count[tile] = 0u;
      BB_RSTAMP(1); BB_RSTAMP(2); BB_RSTAMP(3); BB_RSTAMP(4);   // (diagnostic build: a light tile's phases all end here)
      return;
    }
    if (__popcll(m_ok) == 1 && __ballot(full) == m_ok) {
      if (tid == 0) {
        const int src = __ffsll((long long)m_ok) - 1;
        const unsigned long long fref = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)ref, src);
        const unsigned long long fhi = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)clip_slot1, src) << kFragPixBits;
        frags[(size_t)tile * TILE_PIXELS] = (fhi << 32) | fref;  // the whole list: 8 bytes instead of 8 KB
        frag_count[tile] = (uint32_t)TILE_PIXELS | kFullTile;
        if (item_groups) atomicAdd(&item_groups[(slot / kItemGroupSlots) * kItemGroupStride], (uint32_t)(TILE_PIXELS / 64));
      }
      if (item_head && wave == 0) append_items((uint32_t)(TILE_PIXELS / 64), kFullTile);
      BB_RSTAMP(1); BB_RSTAMP(2); BB_RSTAMP(3); BB_RSTAMP(4);   // (diagnostic build: a light tile's phases all end here)
      return;
    }
  }

  for (int p = tid; p < TILE_PIXELS; p += kTileThreads) {
    unsigned long long k0 = 0ull;
    if (OVERLAY) {  // depth test against what the scene left behind; low word 0 = "no overlay primitive here"
      int x, y;
      tile_pixel<TILE_W>(p, x, y);
      const int gx = tile_x0 + x, gy = tile_y0 + y;
      if (gx < fp.width && gy < fp.height)
        k0 = (unsigned long long)__float_as_uint(depth_io[(size_t)gy * (size_t)fp.width + (size_t)gx]) << 32;
    }
    keys[p] = k0;
  }
  if (tid == 0) {
    s_count = 0;
    s_n_clip_refs = 0;
  }
  __syncthreads();  // keys cleared; every wave has its copy of the counts
  // (a tile rasterised from a heavy slot keeps its counts: its own workgroup may not have looked at them yet.  k_shade_items
  //  clears them.)
  if (!heavy_slot && tid < (int)kBinClasses && (tid == 0 ? n_cls[0] : (tid == 1 ? n_cls[1] : n_cls[2])))
    tile_count[tile * kBinClasses + tid] = 0;  // ready for the slot's next frame

  // entry order: class 0 | class 1 | class 2 | every-tile list
  const uint32_t e1 = n_cls[0], e2 = e1 + n_cls[1], e3 = e2 + n_cls[2], e_end = e3 + n_broad;
  const uint32_t *bin0 = bins + (size_t)(tile * kBinClasses) * fp.bin_cap;

  // overlay pass: the gizmo's fragments are lifted above every other depth (see depth_max); 0 everywhere else
  auto zbias_of = [&](uint32_t ref) -> uint32_t {
    return (OVERLAY && (ref >> 3) >= fp.ov_first_gizmo_prim) ? 0x40000000u : 0u;
  };
  BB_RSTAMP(1);
  // ---- the chunk loop: up to kStage entries at a time ----
  // An entry is fetched in two dependent trips (bin reference -> triangle record; an every-tile entry in one).  Both run
  // AHEAD of the chunk they belong to: the references of chunk k + 2 and the triangles of chunk k + 1 are asked for before
  // chunk k is rasterised, so a tile with several chunks (a far ball puts 2500 tiny triangles into one tile: ten chunks)
  // pays the two trips once, not once per chunk -- they had been the critical path of the frame's heaviest tiles.
  struct Fetched {  // what the staging keeps of an entry (the first 36 bytes of its RasterTri)
    int X0, Y0, X1, Y1, X2, Y2;
    float z0, dzdx, dzdy;
    uint32_t ref, clip_slot1;
  };
  auto fetch_ref = [&](uint32_t e) -> uint32_t {  // bin reference of entry e (0 for the every-tile list and past the end)
    if (e >= e3) return 0u;
    const uint32_t c = e < e1 ? 0u : (e < e2 ? 1u : 2u);
    const uint32_t i = e - (c == 0u ? 0u : (c == 1u ? e1 : e2));
    return bin0[(size_t)c * fp.bin_cap + i];
  };
  auto fetch_tri = [&](uint32_t e, uint32_t ref) -> Fetched {
    Fetched f = {};
    if (e >= e_end) return f;
    const RasterTri *t = e < e3 ? &tris[ref >> 3] : &broad_list[e - e3].tri;  // binned triangles are never clipped
    f.X0 = t->X0; f.Y0 = t->Y0; f.X1 = t->X1; f.Y1 = t->Y1; f.X2 = t->X2; f.Y2 = t->Y2;
    f.z0 = t->z0; f.dzdx = t->dzdx; f.dzdy = t->dzdy;
    f.ref = ref;
    if (e >= e3) {
      f.ref = broad_list[e - e3].ref;
      f.clip_slot1 = broad_list[e - e3].pad[0];
    }
    return f;
  };
  uint32_t ref_next = fetch_ref((uint32_t)tid + (uint32_t)kStage);
  Fetched cur = fetch_tri((uint32_t)tid, fetch_ref((uint32_t)tid));
  for (uint32_t base = 0; base < e_end; base += kStage) {
    if (base) __syncthreads();  // previous chunk consumed
    // ---- stage: every thread files one entry; the rows of the small triangles (classes 0 and 1) are counted ----
    {
      const uint32_t e = base + (uint32_t)tid;
      uint32_t box = 0xFFFFFFFFu;
      int rows = 0;
      if (e < e_end) {
        const Fetched &t = cur;
        int32_t minX = min(t.X0, min(t.X1, t.X2)), maxX = max(t.X0, max(t.X1, t.X2));
        int32_t minY = min(t.Y0, min(t.Y1, t.Y2)), maxY = max(t.Y0, max(t.Y1, t.Y2));
        int px0 = max((minX - 128 + 255) >> 8, tile_x0), px1 = min((maxX - 128) >> 8, rx1);
        int py0 = max((minY - 128 + 255) >> 8, tile_y0), py1 = min((maxY - 128) >> 8, ry1);
        if (OVERLAY && (t.ref >> 3) >= fp.ov_first_gizmo_prim) {  // the gizmo's scissor rectangle (src/main.cpp:767-772)
          px0 = max(px0, fp.ov_x0); px1 = min(px1, fp.ov_x1 - 1);
          py0 = max(py0, fp.ov_y0); py1 = min(py1, fp.ov_y1 - 1);
        }
        bool ok = px0 <= px1 && py0 <= py1;
        if (ok && e >= e3) {  // every-tile list: accept / reject
          RasterTri rt;
          rt.X0 = t.X0; rt.Y0 = t.Y0; rt.X1 = t.X1; rt.Y1 = t.Y1; rt.X2 = t.X2; rt.Y2 = t.Y2;
          ok = classify_rect(edge_setup(rt), px0, px1, py0, py1) != 0;
        }
        if (ok) {
          if (t.clip_slot1) {
            const uint32_t k = atomicAdd(&s_n_clip_refs, 1u);
            if (k < kClipRefs) {
              s_clip_ref[k] = t.ref;
              s_clip_slot[k] = t.clip_slot1;
            }
          }
          box = (uint32_t)(px0 - tile_x0) | ((uint32_t)(px1 - tile_x0) << 8) | ((uint32_t)(py0 - tile_y0) << 16) |
                ((uint32_t)(py1 - tile_y0) << 24);
          st.X0[tid] = t.X0; st.Y0[tid] = t.Y0; st.X1[tid] = t.X1; st.Y1[tid] = t.Y1; st.X2[tid] = t.X2; st.Y2[tid] = t.Y2;
          st.z0[tid] = t.z0; st.dzdx[tid] = t.dzdx; st.dzdy[tid] = t.dzdy;
          st.ref[tid] = t.ref;
          if (e < e2) rows = py1 - py0 + 1;
        }
      }
      st.box[tid] = box;
      // The rows of the wave's small triangles, as one list per wave: lane l's rows are items [first, first + rows) of it.
      // (exclusive prefix sum over the wave, bit-sliced: rows <= TILE_H needs seven ballots and mbcnt pairs -- no LDS traffic, no
      //  shuffle addresses to keep in registers)
      int first = 0;
      uint32_t wave_rows = 0u;
#pragma unroll
      for (int b = 0; (1 << b) <= TILE_H; ++b) {
        const unsigned long long m = __ballot(((rows >> b) & 1) != 0);
        first += (int)(__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << b);
        wave_rows += (uint32_t)__popcll(m) << b;
      }
      s_row0[tid] = (uint16_t)first;
      if (lane == 0) s_wave_rows[wave] = wave_rows;
    }
    __syncthreads();
    if (base == 0) BB_RSTAMP(2);
    const uint32_t hi = min(base + (uint32_t)kStage, e_end);
    // ---- class 2 and the every-tile list: one wave per large triangle ----
    {
      const uint32_t lo = max(base, e2);
      for (uint32_t e = lo + (uint32_t)wave; e < hi; e += kTileWaves) {
        const int j = __builtin_amdgcn_readfirstlane((int)(e - base));
        if (st.box[j] == 0xFFFFFFFFu) continue;
        const RasterTri t = staged_tri(st, j);
        if (OVERLAY && (st.ref[j] >> 3) >= fp.ov_first_gizmo_prim)
          raster_triangle_wave<TILE_W, TILE_H>(t, st.ref[j], tile_x0, tile_y0, fp, keys, lane, zbias_of(st.ref[j]), fp.ov_x0,
                                               fp.ov_y0, fp.ov_x1, fp.ov_y1);
        else
          raster_triangle_wave<TILE_W, TILE_H>(t, st.ref[j], tile_x0, tile_y0, fp, keys, lane);
      }
    }
    // (behind the wave-per-triangle loop, whose 64-bit edge functions need the registers)
    // the next chunk's triangles and the references of the one after it: in flight while this chunk is rasterised
#if BB_RASTER_PREFETCH
    {
      const uint32_t e_next = base + (uint32_t)kStage + (uint32_t)tid;
      cur = fetch_tri(e_next, ref_next);
      ref_next = fetch_ref(e_next + (uint32_t)kStage);
    }
#endif
    // ---- classes 0 and 1 (small triangles): one ROW of one triangle per lane ----
    // Round 4 gave a tiny triangle one lane (its whole bounding box, pixel by pixel: 16 % of the lanes busy) and a small one
    // sixteen lanes (4 x 4 pixel blocks over its box, a 70-instruction edge setup repeated in each of them: 56 % busy); a tile's
    // life was the life of its unluckiest lane group.  Now the rows of all of them are one list and the 256 threads take rows
    // from it in turn: every lane has work of the same kind whatever the triangle's size, consecutive lanes hold consecutive
    // rows of one triangle (the triangle's words are LDS broadcasts, the rows about equally long), and a row's setup is
    // paid once per row, not per 4 x 4 block.  Same pixels, same keys: the edge functions and the depth expression are
    // the ones of the 64-bit form.
    {
      const uint32_t r0 = s_wave_rows[0], r1 = r0 + s_wave_rows[1], r2 = r1 + s_wave_rows[2], r3 = r2 + s_wave_rows[3];
      static_assert(kTileWaves == 4, "row list: four wave regions");
      for (uint32_t i = (uint32_t)tid; i < r3; i += kTileThreads) {
        const uint32_t w = (i >= r0 ? 1u : 0u) + (i >= r1 ? 1u : 0u) + (i >= r2 ? 1u : 0u);
        const uint32_t local = i - (w == 0u ? 0u : (w == 1u ? r0 : (w == 2u ? r1 : r2)));
        // whose row: the last entry of wave w's 64 whose first row is <= local (entries without rows share their successor's
        // first row, so the last one is the one that has it)
        int j = (int)(w * 64u);
#pragma unroll
        for (int step = 32; step; step >>= 1)
          if ((uint32_t)s_row0[j + step] <= local) j += step;
        const uint32_t box = st.box[j];
        const int py = tile_y0 + (int)((box >> 16) & 255u) + (int)(local - (uint32_t)s_row0[j]);
        const uint32_t ref = st.ref[j];
        raster_triangle_row<TILE_W, TILE_H>(st.X0[j], st.Y0[j], st.X1[j], st.Y1[j], st.X2[j], st.Y2[j], st.z0[j], st.dzdx[j],
                                            st.dzdy[j], ref, tile_x0 + (int)(box & 255u), tile_x0 + (int)((box >> 8) & 255u), py,
                                            tile_x0, tile_y0, keys, zbias_of(ref));
      }
    }
#if !BB_RASTER_PREFETCH
    if (base + (uint32_t)kStage < e_end) {
      const uint32_t e_next = base + (uint32_t)kStage + (uint32_t)tid;
      cur = fetch_tri(e_next, fetch_ref(e_next));
    }
#endif
  }
  __syncthreads();
  BB_RSTAMP(3);

  // ---- compaction: covered pixels -> fragment list (ballot + popcount prefix); background written here ----
  // fragment = ((clip slot + 1) << kFragPixBits | pixel in tile) << 32 | reference
  unsigned long long *my_frags = frags + (size_t)tile * TILE_PIXELS;
  const uint32_t n_clip_refs = s_n_clip_refs <= kClipRefs ? s_n_clip_refs : 0u;  // too many: k_shade looks the slots up
  // (The wave counts the covered pixels of ALL its passes first and gets the place of its run from a prefix over the four
  //  waves' totals -- no atomic at all; round 4 reserved per pass, four dependent returning LDS atomics per wave of every ball tile.)
  constexpr int PASSES = TILE_PIXELS / kTileThreads;
  unsigned long long pass_mask[PASSES];
  uint32_t wave_total = 0u;
  // Which 8 x 8 block of the tile a wave takes in pass k: CONSECUTIVE blocks (for 32-pixel tiles a row of four, a 32 x 8 strip).
  // The wave's covered pixels go into the fragment list as one run, 64 consecutive fragments are an item of k_shade and
  // four consecutive items a workgroup (one CU; the next workgroup is on the next XCD): with the runs in wave order (below)
  // workgroup j of a tile shades strip j, in every tile and every frame.  Measured with the runs in wave order, k_shade's
  // fetch traffic per launch (FETCH_SIZE, KiB, C3): strips 53.8 k -- what round 4's one-atomic-per-pass compaction had --, a
  // quarter of the tile per wave 62.2 k, a column of blocks per wave 67.1 k (gpurun_out/r5/t12_*).
  auto pass_pixel = [&](int k) -> int { return (wave * PASSES + k) * 64 + lane; };
#pragma unroll
  for (int k = 0; k < PASSES; ++k) {
    const int p = pass_pixel(k);
    int x, y;
    tile_pixel<TILE_W>(p, x, y);
    const bool in_frame = tile_x0 + x < fp.width && tile_y0 + y < fp.height;
    const unsigned long long key = keys[p];
    pass_mask[k] = __ballot(in_frame && (OVERLAY ? (uint32_t)key != 0u : key != 0ull));
    wave_total += (uint32_t)__popcll(pass_mask[k]);
  }
  // The four runs stand in the list in WAVE order -- an exclusive prefix over the waves' totals through LDS, one more barrier
  // -- not in the order in which four atomics happen to arrive: the fragment list of a tile is then the same from frame to
  // frame, and so is which of k_shade's workgroups (which XCD, which L2) shades which part of the tile.  With a returning
  // atomic per wave the runs came in any order and k_shade's fetch traffic rose by a fifth (the same L1 misses, more of them
  // L2 misses: neighbouring tiles no longer sent their shared texel lines to the same L2s).
  if (lane == 0) s_wave_rows[wave] = wave_total;   // (the row list's totals are done with: behind the chunk loop's last barrier)
  __syncthreads();
  uint32_t wave_base = 0u;
#pragma unroll
  for (int w = 0; w < kTileWaves; ++w) wave_base += w < wave ? s_wave_rows[w] : 0u;
  if (tid == 0) s_count = s_wave_rows[0] + s_wave_rows[1] + s_wave_rows[2] + s_wave_rows[3];
#pragma unroll
  for (int k = 0; k < PASSES; ++k) {
    int p = pass_pixel(k);
    int x, y;
    tile_pixel<TILE_W>(p, x, y);
    int gx = tile_x0 + x, gy = tile_y0 + y;
    bool in_frame = gx < fp.width && gy < fp.height;
    unsigned long long key = keys[p];
    // main passes: any key; overlay pass: only pixels an overlay primitive won (low word = reference + 1)
    const unsigned long long mask = pass_mask[k];
    const bool covered = ((mask >> lane) & 1ull) != 0ull;
    if (covered) {
      uint32_t rank_in_wave = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      const uint32_t ref = (uint32_t)key - 1u;
      uint32_t clip_slot1 = 0u;
      for (uint32_t k = 0; k < n_clip_refs; ++k)  // (uniform trip count: a handful of clipped sub-triangles per tile at most)
        if (s_clip_ref[k] == ref) clip_slot1 = s_clip_slot[k];
      const unsigned long long fw =
          ((((unsigned long long)clip_slot1 << kFragPixBits) | (unsigned long long)(uint32_t)p) << 32) | (unsigned long long)ref;
      my_frags[wave_base + rank_in_wave] = fw;
      if (wave_base + rank_in_wave == 0u) s_pad_frag = fw;  // the list's first fragment doubles as its padding (below)
    } else if (in_frame && !OVERLAY) {
      store_background(x, y);
    }
    wave_base += (uint32_t)__popcll(mask);
    if (vis_prim && in_frame) {
      size_t o = (size_t)gy * (size_t)fp.width + (size_t)gx;
      vis_prim[o] = key ? (((uint32_t)key - 1u) >> 3) : 0xFFFFFFFFu;
      vis_depth[o] = __uint_as_float((uint32_t)(key >> 32));
    }
    // the resolved depth, kept for the overlay subpass (option "overlays")
    if (!OVERLAY && depth_io && in_frame) depth_io[(size_t)gy * (size_t)fp.width + (size_t)gx] = __uint_as_float((uint32_t)(key >> 32));
  }
  __syncthreads();
  // k_shade works in units of 64 fragments and loads them before it knows the count: the list is padded to a multiple of
  // 64 with copies of its first fragment (the copies are shaded again and not stored).
  {
    const uint32_t n = s_count;
    if (n != 0u && (n & 63u) != 0u && tid < (int)(64u - (n & 63u))) my_frags[n + (uint32_t)tid] = s_pad_frag;
  }
  if (tid == 0) {
    frag_count[tile] = s_count;  // (N_shaded = sum of these, taken on the host on demand)
    // 64-fragment chunks per group of 256 launch slots: what k_shade_items needs to place this group's items
    if (item_groups && s_count) atomicAdd(&item_groups[(slot / kItemGroupSlots) * kItemGroupStride], (s_count + 63u) >> 6);
  }
  if (item_head && wave == 0 && s_count) append_items((s_count + 63u) >> 6, 0u);
  BB_RSTAMP(4);
  if (tid == 0) {
    BB_RSTAMP(5);
#ifdef BB_STAMPS
    reinterpret_cast<unsigned long long *>(frag_count + fp.tiles_x * fp.tiles_y)[tile * 8 + 6] =
        ((unsigned long long)e_end << 32) | s_count;
    reinterpret_cast<unsigned long long *>(frag_count + fp.tiles_x * fp.tiles_y)[tile * 8 + 7] =
        (unsigned long long)n_cls[0] | ((unsigned long long)n_cls[1] << 16) | ((unsigned long long)n_cls[2] << 32) |
        ((unsigned long long)n_broad << 48);
#endif
  }
}

// The light loop and the ambient term: forward_brdf.frag:27-75 == brdf.frag:26-72 (same statements), on a surface
// point given by position, (unnormalised) normal, albedo, metallic, roughness, ao.
//
// Evaluation order = the CPU checker's "contract" form of the light loop, bit for bit.  Everything that feeds the
// one ill-conditioned quantity of the shader, the GGX denominator q = NdotH^2 (a^2 - 1) + 1, follows the GLSL statement
// by statement (V, N, L, att, H, NdotH); the well-conditioned products behind it are re-associated:
//   D G / max(4 NdotV NdotL, .001) = (a2 NdotV NdotL) / ((q q) (PI dV dL) sden)   one reciprocal instead of four
//   kD albedo / PI                 = (1 - F) ((1 - metallic) albedo / PI)         hoisted out of the loop
//   radiance NdotL                 = (color intensity) (att NdotL)                color * intensity once per light
//   max(N.V, 0), max(N.L, 0), max(H.V, 0) = saturate(...)   the three cosines that do not feed q: differs only where
//                                                           rounding puts a cosine of unit vectors above 1 (<= 2 ulp)
// Why the shape matters on gfx950 (tools/microbench/issue_rate.hip, profiles/r02_issue_rate.txt): with two or more waves on
// a SIMD, v_fma/mul/add_f32 issue every 2.05-2.4 cycles -- also with one scalar-register operand (row iso_fma_sgpr_operand:
// 2.7) -- while v_max / v_cmp / conversions cost ~4, a transcendental 13 and a packed, DPP or 24-bit-multiply instruction in
// a stream of plain ones 9-10.  Per-light constants are cooked once per FRAME (k_shade_items) and reach k_shade's loop through
// the scalar cache (ConstLights below: s_load into scalar registers, the light's type already a scalar); only
// k_deferred_background still stages them in LDS (LdsLights).
// ------------------------------------------------------------------------------------------------

// every thread of the workgroup; the caller synchronises
template <int THREADS>
BB_DEV void stage_lights(const ShadeParams &sp, const Light *__restrict__ lights, ShadeShared &sh) {
  for (int li = (int)threadIdx.x; li < sp.num_lights; li += THREADS) sh.lights[li] = cook_light(lights[li]);
}

// Where the light loop finds its cooked lights.
//  * LdsLights: a table staged by the workgroup itself (stage_lights + barrier): k_deferred_background.
//  * ConstLights: the frame's table in global memory, cooked once per frame by k_shade_items and read through the SCALAR
//    cache (constant address space -> s_load into scalar registers): k_shade.  No LDS, no staging code, and above all no
//    workgroup barrier in front of the light loop -- the four waves of a workgroup never wait for each other, and a light's
//    type is already a scalar (no v_readfirstlane, 9 issue cycles each).  The price: a VALU instruction that reads one of
//    the light's fields takes it as its one scalar operand.
struct LdsLights {
  const ShadeShared &sh;
  BB_DEV CookedLight get(int li) const { return sh.lights[li]; }
  BB_DEV static int type_of(const CookedLight &c) { return __builtin_amdgcn_readfirstlane(c.type); }
};
typedef const CookedLight __attribute__((address_space(4))) *ConstCooked;
struct ConstLights {
  ConstCooked table;
  BB_DEV CookedLight get(int li) const {
    ConstCooked c = table + li;
    CookedLight o;
    o.px = c->px; o.py = c->py; o.pz = c->pz; o.type = c->type;
    o.cr = c->cr; o.cg = c->cg; o.cb = c->cb; o.outer = c->outer;
    o.dx = c->dx; o.dy = c->dy; o.dz = c->dz; o.inv_eps = c->inv_eps;
    return o;
  }
  BB_DEV static int type_of(const CookedLight &c) { return c.type; }
};

template <typename Lights>
BB_DEV float4 light_surface(const ShadeParams &sp, const Lights lights, f3 P, f3 normal, f3 albedo, float metallic,
                            float roughness, float ao) {
  // per-pixel invariants
  const f3 V = normalize3(sub3(mk3(sp.view_pos[0], sp.view_pos[1], sp.view_pos[2]), P));
  const f3 N = normalize3(normal);
  const float NdotV = sat01(dot3(V, N));
  const float rr = roughness + 1.0f;
  const float kk = (rr * rr) * 0.125f, omk = 1.0f - kk;
  const float pidV = kPi * fmaf(NdotV, omk, kk);
  const float a = roughness * roughness, a2 = a * a, a2m1 = a2 - 1.0f;
  const float a2nv = a2 * NdotV, c4 = 4.0f * NdotV;
  const f3 F0 = mk3(mixf(0.04f, albedo.x, metallic), mixf(0.04f, albedo.y, metallic), mixf(0.04f, albedo.z, metallic));
  const f3 omF0 = mk3(1.0f - F0.x, 1.0f - F0.y, 1.0f - F0.z);
  const float om = 1.0f - metallic;
  const f3 kda = mk3((om * albedo.x) * kInvPi, (om * albedo.y) * kInvPi, (om * albedo.z) * kInvPi);

  f3 Lo = mk3(0.0f, 0.0f, 0.0f);
  // The next light is fetched one iteration ahead (LDS broadcasts / scalar loads), so the loop never waits for it.
  CookedLight nxt = {};
  if (sp.num_lights > 0) nxt = lights.get(0);
  for (int li = 0; li < sp.num_lights; ++li) {
    const CookedLight cl = nxt;
    const float px = cl.px, py = cl.py, pz = cl.pz, cr = cl.cr, cg = cl.cg, cb = cl.cb;
    const int type = Lights::type_of(cl);
    nxt = lights.get(li + 1 < sp.num_lights ? li + 1 : li);
    f3 L;
    float att;
    if (type == 0 || type == 1) {
      const f3 Lv = sub3(mk3(px, py, pz), P);
      const float inv_d = bb_rsqrt(dot3(Lv, Lv));
      att = inv_d * inv_d;
      L = scale3(Lv, inv_d);
      if (type == 1) att *= clamp01((dot3(L, mk3(cl.dx, cl.dy, cl.dz)) - cl.outer) * cl.inv_eps);
    } else if (type == 2) {
      L = mk3(cl.dx, cl.dy, cl.dz);
      att = 1.0f;
    } else {
      continue;  // upstream leaves L / att uninitialised; the contract contributes nothing
    }
    const f3 H = normalize3(add3(L, V));
    const float NdotH = max0(dot3(N, H));
    const float q = fmaf(NdotH * NdotH, a2m1, 1.0f);
    const float x = 1.0f - sat01(dot3(H, V));
    const float x2 = x * x;
    const float p5 = (x2 * x2) * x;
    const float NdotL = sat01(dot3(N, L));
    const float dL = fmaf(NdotL, omk, kk);
    const float sden = __builtin_fmaxf(c4 * NdotL, 0.001f);  // NaN -> 0.001
    const float den = ((q * q) * (pidV * dL)) * sden;
    const float S = (a2nv * NdotL) * bb_rcp(den);
    const f3 F = mk3(fmaf(omF0.x, p5, F0.x), fmaf(omF0.y, p5, F0.y), fmaf(omF0.z, p5, F0.z));
    const float rl = att * NdotL;
    Lo.x = fmaf(fmaf(1.0f - F.x, kda.x, F.x * S), cr * rl, Lo.x);
    Lo.y = fmaf(fmaf(1.0f - F.y, kda.y, F.y * S), cg * rl, Lo.y);
    Lo.z = fmaf(fmaf(1.0f - F.z, kda.z, F.z * S), cb * rl, Lo.z);
  }
  float4 color;
  color.x = fmaf(0.03f * albedo.x, ao, Lo.x);
  color.y = fmaf(0.03f * albedo.y, ao, Lo.y);
  color.z = fmaf(0.03f * albedo.z, ao, Lo.z);
  color.w = 1.0f;
  return color;
}

// Deferred path: brdf.frag runs on every pixel of its full-screen triangle (src/main.cpp:101-104), also where the
// G-buffer still holds its clear value 0; that colour is the same for all such pixels (it depends on the lights and
// the camera only -- and is not always 0: a light at the world origin makes it NaN), so it is evaluated once.
constexpr int kBackgroundThreads = 64;
__global__ __launch_bounds__(kBackgroundThreads) void k_deferred_background(ShadeParams sp, const Light *__restrict__ lights,
                                                                            float4 *__restrict__ out,
                                                                            const SrgbTables *__restrict__ tables, int gbuffer_view) {
  __shared__ ShadeShared sh;
  stage_lights<kBackgroundThreads>(sp, lights, sh);
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    // (buffer_visualize.frag on a cleared texel: rgb 0, alpha 1)
    const float4 c = gbuffer_view >= 0 ? make_float4(0.f, 0.f, 0.f, 1.f)
                                       : light_surface(sp, LdsLights{sh}, mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 0.f), 0.f, 0.f, 0.f);
    out[0] = c;
    // out[1].x: the same colour as a presented pixel (fused presentation)
    if (tables) out[1] = make_float4(__uint_as_float(present_pixel(c.x, c.y, c.z, *tables, sp.tone_enable, sp.exposure, 1)), 0.f, 0.f, 0.f);
  }
}

// ------------------------------------------------------------------------------------------------
// k_shade_items: the work list of k_shade.  One item = 64 consecutive fragments of one tile's list; k_shade runs one
// wave per item (no wave for an empty part of a tile, none half empty).  items[0] = number of items, items[1 + j] =
// full-tile flag << 31 | grid row << 18 | tile column << 6 | chunk of 64 fragments (a tile of 64x64 pixels has 64 chunks: the
// chunk field is 6 bits), in launch-slot (= screen) order: neighbouring waves shade neighbouring pixels.
// ------------------------------------------------------------------------------------------------

// chunks of 64 fragments in the list of launch slot `slot` (0 past the frame)
BB_DEV uint32_t slot_chunks(const FrameParams &fp, const uint32_t *__restrict__ frag_count, uint32_t slot, uint32_t n_slots, int grid_x) {
  if (slot >= n_slots) return 0u;
  const int gy = (int)(slot / (uint32_t)grid_x), tx = (int)(slot - (uint32_t)gy * (uint32_t)grid_x);
  int ty, out_tile_row;
  if (!tile_row(fp, gy, ty, out_tile_row)) return 0u;
  const uint32_t n = frag_count[(uint32_t)ty * (uint32_t)fp.tiles_x + (uint32_t)tx];
  return (((n & ~kFullTile) + 63u) >> 6) | (n & kFullTile);  // (chunks, with the full-tile flag kept on top)
}

// One workgroup per 256 launch slots, no communication between workgroups: the items in front of a workgroup's slots are
// the sum of k_raster's per-group totals (eight independent loads per thread), its own 256 slots get a scan, then their items are
// written.  (A single workgroup doing all of it took 25 us at C3 -- 70 000 scattered 4-byte stores through ONE compute
// unit's address unit; every workgroup summing the raw counts in front of it took 13 us of dependent loads.)
template <int TILE_W, int TILE_H>
__global__ __launch_bounds__(kItemsThreads) void k_shade_items(FrameParams fp, const uint32_t *__restrict__ frag_count,
                                                               const uint32_t *__restrict__ item_groups,
                                                               uint32_t *__restrict__ items, int grid_x, int grid_y,
                                                               uint32_t *__restrict__ host_count,
                                                               const Light *__restrict__ lights, int num_lights,
                                                               CookedLight *__restrict__ cooked,
                                                               uint32_t *__restrict__ tile_count, const Counters *__restrict__ ctr,
                                                               const uint32_t *__restrict__ heavy) {
  __shared__ uint32_t s_wave[2][kItemsThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // tiles that k_raster rasterised from a heavy slot still have their bin counts (their own workgroups had to see them):
  // cleared here for the slot's next frame
  if (fp.heavy_rows) {
    const uint32_t n_heavy = ctr->n_heavy;
    if (n_heavy <= (uint32_t)fp.heavy_rows * (uint32_t)grid_x)
      for (uint32_t i = blockIdx.x * (uint32_t)kItemsThreads + (uint32_t)tid; i < n_heavy; i += gridDim.x * (uint32_t)kItemsThreads) {
        const uint32_t hs = heavy[i] & 0x3FFFFFFFu;
        const int gy = (int)(hs / (uint32_t)grid_x), tx = (int)(hs - (uint32_t)gy * (uint32_t)grid_x);
        int ty, out_tile_row;
        if (tile_row(fp, gy, ty, out_tile_row)) {
          const uint32_t tile = (uint32_t)ty * (uint32_t)fp.tiles_x + (uint32_t)tx;
#pragma unroll
          for (uint32_t c = 0; c < kBinClasses; ++c) tile_count[tile * kBinClasses + c] = 0u;
        }
      }
  }
  // the frame's cooked light table (k_shade reads it through the scalar cache): once per frame, by the last workgroup
  // (the one with the fewest slots to scan)
  if (blockIdx.x == gridDim.x - 1)
    for (int li = tid; li < num_lights; li += kItemsThreads) cooked[li] = cook_light(lights[li]);
  const uint32_t n_slots = (uint32_t)grid_x * (uint32_t)grid_y;
  const uint32_t first = blockIdx.x * (uint32_t)kItemsThreads;
  uint32_t before = 0u;  // (independent loads: one round trip)
#pragma unroll
  for (int q = 0; q < kItemGroups / kItemsThreads; ++q) {
    const uint32_t g = (uint32_t)q * kItemsThreads + (uint32_t)tid;
    before += g < first / kItemGroupSlots ? item_groups[g * kItemGroupStride] : 0u;
  }
  const uint32_t chunks_flag = slot_chunks(fp, frag_count, first + (uint32_t)tid, n_slots, grid_x);
  const uint32_t chunks = chunks_flag & ~kFullTile;
  uint32_t incl = chunks;  // inclusive scan of this workgroup's slots, wave level
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
    if (lane >= d) incl += up;
    before += (uint32_t)__shfl_xor((int)before, d);  // (butterfly sum: every lane ends with the wave's total)
  }
  if (lane == 63) s_wave[0][wave] = incl;
  if (lane == 0) s_wave[1][wave] = before;
  __syncthreads();
  uint32_t at = 1u + incl - chunks;
#pragma unroll
  for (int w = 0; w < kItemsThreads / 64; ++w) {
    at += s_wave[1][w];
    if (w < wave) at += s_wave[0][w];
  }
  const uint32_t slot = first + (uint32_t)tid;
  if (chunks != 0u) {
    const uint32_t gy = slot / (uint32_t)grid_x, tx = slot - gy * (uint32_t)grid_x;
    const uint32_t word = (chunks_flag & kFullTile) | (gy << (kItemChunkBits + kItemTxBits)) | (tx << kItemChunkBits);
    for (uint32_t c = 0; c < chunks; ++c) items[at + c] = word | c;
  }
  // the workgroup of the last slot knows the total
  if (first + (uint32_t)kItemsThreads >= n_slots && tid == kItemsThreads - 1) {
    items[0] = at + chunks - 1u;
    if (host_count) *host_count = at + chunks - 1u;  // pinned host memory: sizes the launch of this slot's next frame
  }
}

// ------------------------------------------------------------------------------------------------
// k_shade: forward_brdf.frag + brdf.glsl once per visible pixel.
//
// What bounded the first version of this kernel (one workgroup per 256 fragments of a tile) was not arithmetic but the
// chain of dependent memory round trips in front of it -- kernel arguments -> fragment count -> fragment -> primitive
// record -> clip slot -> texels, ~0.8 us each under load -- against ~0.9 us of issue time per 64 fragments: its
// duration followed T = 66 us + 264 us / (waves per SIMD) at C3 (measured by padding LDS: 2, 3, 4, 7 waves -> 198,
// 151, 132, 109 us), and a body that only loaded the fragment and stored a constant still took 52 us.  Hence:
//  * ONE WAVE = ONE ITEM of 64 fragments from the list of k_shade_items (item = grid row, tile column, chunk of the
//    tile's fragment list): no wave is launched for an empty part of a tile, every launched wave is full (k_raster pads
//    each list to a multiple of 64 with copies of its first fragment), and no count has to be read before the fragments.
//    The main launch is sized on the host from the item count the same frame slot produced one frame earlier (pinned
//    word written by k_shade_items) plus 3 % + 64; a second, 32-workgroup instantiation (TAIL) walks whatever lies beyond
//    that estimate in a loop, so a frame that suddenly has more fragments is still complete.  (A SHORT frame -- 1080p --
//    has neither: its raster tiles append their items themselves, k_raster above, the count is read through `item_count`,
//    and it is launched at full coverage.)
//  * the chain is cut to three round trips per item (item word -> fragment words -> record + texels): the fragment
//    word carries the clip-arena slot of a clipped sub-triangle, so its planes are fetched together with the primitive
//    record instead of after it; a wave whose 64 fragments share one primitive fetches the record through the scalar
//    cache (constant address space), and so does every wave the frame's cooked light table: no LDS, no barrier.
//  * many waves per SIMD (64 VGPRs -> the hardware's eight) rather than a persistent, software-pipelined loop: a gfx950 wave issues at
//    most one instruction of ANY kind every ~4 cycles (profiles/r02_issue_rate.txt), so the vector ALU (one instruction
//    every 2 cycles) only fills up with several waves that are all busy issuing.  The persistent forms were built and
//    measured in round 2 ("dead ends"): loop-carried state costs 36 VGPRs, and every forced
//    occupancy spilled.
// ------------------------------------------------------------------------------------------------
#ifndef BB_SHADE_THREADS
#define BB_SHADE_THREADS 256
#endif
#ifndef BB_SHADE_WAVES
#define BB_SHADE_WAVES 8  // waves per SIMD the main forward instantiation is compiled for (64 registers)
#endif
constexpr int kShadeThreads = BB_SHADE_THREADS;
constexpr int kFrontPriorityLights = 6;  // k_shade: from this many lights on, a wave's load phases run at a raised issue priority
#ifdef BB_STAMPS
__device__ unsigned long long g_shade_stamps[4096 * 8];  // diagnostic build: per-wave phase cycles of k_shade
#endif
constexpr int kShadeWaves = kShadeThreads / 64;
typedef const uint8_t __attribute__((address_space(1))) *GlobalBytes;  // a pointer known to point to global memory
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 __attribute__((aligned(1))) u32x3_any;  // 12 bytes at any byte address: one global_load_dwordx3
typedef const u32x3_any __attribute__((address_space(1))) *GlobalTap;

// one work item as the loop carries it: where its 64 fragments are (all uniform, held in scalar registers) and the
// fragment word of this lane
struct ItemFrag {
  unsigned long long frag;
  uint32_t n_frag;                  // fragments in the tile's list
  int tx, ty, out_tile_row, chunk;
  bool live;                        // the item exists
};

// PRESENT = true (option "present_fused"): the colour goes through present_pixel and is stored as RGBA8 -- the tone-map
// subpass fused into the producing kernel: 4 bytes written per pixel instead of 16, and no k_present pass (16 B read +
// 4 B written per pixel) afterwards.
// TAIL = false: one item per wave -- wave v of workgroup w shades item first_item + 4 w + v and exits; the host sizes the
// grid from the item count of the frame this slot rendered before (frames are coherent), so nearly every wave launched
// has an item.  TAIL = true: a small fixed grid loops over whatever lies behind the part the main launch covered (a
// scene that suddenly grew); normally nothing.  Two instantiations because the loop form costs registers: the compiler
// keeps ~100 live around a loop where the straight-line body fits 64 -- four waves per SIMD instead of eight.  The main
// instantiation asks for eight waves per SIMD (amdgpu_waves_per_eu): the allocator then gets from 67 registers to 64
// without a spill, and the eighth wave is worth 3.4 us of the kernel's 76 at C3 (latency hiding is what this kernel runs on).
// MIXED = false: every material of the frame is packed (its five shaded maps share one size, or are the uniform defaults:
// what bbr_upload_material produces for every ShaderBall material) -- the instantiation then has no per-map sampling path
// at all, and that path's registers and code are not the main launch's problem.  The host knows (upload_material_table).
template <int TILE_W, int TILE_H, bool DEFERRED, bool PRESENT = false, bool TAIL = false, bool MIXED = true>
__global__ __launch_bounds__(kShadeThreads) __attribute__((amdgpu_waves_per_eu(TAIL ? 1 : ((DEFERRED || MIXED) ? 7 : BB_SHADE_WAVES)))) void k_shade(
    // (first: what stands between a wave's launch and its first loads -- the item word, the item count, then the fragment
    //  words and the record: these pointers arrive in scalar registers with the wave, "kernarg preload", 14 dwords at most)
    const uint32_t *__restrict__ items, const uint32_t *__restrict__ item_count, uint32_t first_item,
    const unsigned long long *__restrict__ frags, const uint32_t *__restrict__ frag_count,
    const ShadeRec *__restrict__ recs, const ClipSlot *__restrict__ clip_arena,
    FrameParams fp, ShadeParams sp, const CookedLight *__restrict__ cooked,
    const MaterialDesc *__restrict__ materials, float4 *__restrict__ out,
    uint2 *__restrict__ gbuffer, const SrgbTables *__restrict__ tables, uint32_t *__restrict__ out8,
    Counters *__restrict__ ctr, Counters *__restrict__ ctr_done, uint32_t *__restrict__ item_groups) {
  constexpr int TILE_PIXELS = TILE_W * TILE_H;
  const ConstLights lights_c{(ConstCooked)cooked};
  // The frame's counter block has done its job (k_geometry filled it, k_raster read it): keep a copy for the host's
  // statistics / overflow check and clear the block for the next frame of this slot.  Frames of different slots
  // share nothing, so their kernels may overlap freely.
  if (!TAIL && blockIdx.x == 0 && threadIdx.x < sizeof(Counters) / 4) {  // (the main launch only)
    reinterpret_cast<uint32_t *>(ctr_done)[threadIdx.x] = reinterpret_cast<uint32_t *>(ctr)[threadIdx.x];
    reinterpret_cast<uint32_t *>(ctr)[threadIdx.x] = 0u;
  }
  if (!TAIL && blockIdx.x == 0 && item_groups)  // (k_shade_items is done with them)
    for (uint32_t g = threadIdx.x; g < (uint32_t)kItemGroups; g += kShadeThreads) item_groups[g * kItemGroupStride] = 0u;
#ifdef BB_STAMPS
  unsigned long long st_t[8];
  st_t[0] = __builtin_amdgcn_s_memtime();
#define BB_KSTAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BB_KSTAMP(i) do { } while (0)
#endif
  // With a LONG light loop (C5's eight lights: two thirds of a wave's instructions) the waves that are still asking for their
  // fragments, records and texels go first: their loads are out sooner and there is always another wave's light loop to
  // issue behind them (C5 frame 509-514 -> 500-503 us on two boxes).  With four lights the same priority costs 2-3 %
  // (C3 101 -> 104 us): there the waves in their light loops are the ones about to give their slots back.
  if (sp.num_lights >= kFrontPriorityLights) __builtin_amdgcn_s_setprio(2);
  const int lane = (int)(threadIdx.x & 63u);
  // this wave's (first) item: wave-uniform, everything derived from it lives in scalar registers.  The item word is read
  // together with the item count, not after it (the list has room for every index a launch can produce): one dependent
  // round trip less in front of the fragments.
  // (Workgroup b runs on XCD b % 8 -- tools/microbench/xcd_map.hip -- so neighbouring items, which share records and
  //  texels, land in eight different L2s.  Giving each XCD a contiguous eighth of the item list cut k_shade's fetch traffic
  //  by 23 % and made it 12 % SLOWER (98 vs 88 us): a hot region then loads one L2 / one XCD's texture units instead of
  //  eight; runs of 4 / 16 / 64 workgroups per XCD: 87 / 89 / 94 us, no traffic gain.  Plain round-robin stays.)
  uint32_t j = first_item + (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (uint32_t)kShadeWaves + (threadIdx.x >> 6)));
  uint32_t item = items[1u + (TAIL ? 0u : j)];
  uint32_t n_items = *item_count;  // (items[0], or the head word k_raster's tiles appended through: short frames)
  // (both loads are in flight before either is waited for: left to itself the compiler reads the count, branches, and only
  //  then asks for the item word -- one more dependent round trip in front of every wave's fragments)
  //  -- and the frame parameters the item is decoded with arrive in the same round trip, not in one of their own behind it)
  asm volatile("" : "+s"(item), "+s"(n_items), "+s"(fp.tiles_x), "+s"(fp.world), "+s"(fp.rank), "+s"(fp.band_tiles), "+s"(fp.width));
  if (BB_ABLATE(2048u)) sp.num_lights = 0;
  if (j >= n_items) return;  // a wave without an item (the kernel has no barrier: waves come and go on their own)
  do {  // (a loop only in the TAIL instantiation)
  if (TAIL) item = items[1u + j];
  const int chunk = (int)(item & 63u);
  const int tx = (int)((item >> kItemChunkBits) & ((1u << kItemTxBits) - 1u));
  int ty, out_tile_row;
  tile_row(fp, (int)((item & ~kFullTile) >> (kItemChunkBits + kItemTxBits)), ty, out_tile_row);
  const uint32_t tile = (uint32_t)ty * (uint32_t)fp.tiles_x + (uint32_t)tx;
  bool valid;
  unsigned long long frag;
  if (item & kFullTile) {
    // a tile one triangle covers completely: no list, no count -- one word through the scalar cache, the pixel is the
    // lane's own (and the wave is uniform: its primitive record comes through the scalar cache too)
    typedef const unsigned long long __attribute__((address_space(4))) *ConstFrags;
    frag = ((ConstFrags)frags)[(size_t)tile * TILE_PIXELS] + ((unsigned long long)((uint32_t)chunk * 64u + (uint32_t)lane) << 32);
    valid = true;
  } else {
    const uint32_t n_frag = frag_count[tile];
    valid = (uint32_t)chunk * 64u + (uint32_t)lane < n_frag;
    frag = frags[(size_t)tile * TILE_PIXELS + (uint32_t)chunk * 64u + (uint32_t)lane];
  }
  BB_KSTAMP(1);  // item word, count and fragment arrived
  const uint32_t ref = (uint32_t)frag;
  uint32_t prim = BB_ABLATE(16u) ? 0u : (ref >> 3);
  if (BB_ABLATE(1024u)) prim = (uint32_t)__builtin_amdgcn_readfirstlane((int)prim);  // the record through the scalar cache
  int x, y;
  tile_pixel<TILE_W>((int)(frag >> 32) & (TILE_PIXELS - 1), x, y);
  const int gx = tx * TILE_W + x, gy = ty * TILE_H + y;
  // the pixel's index in the output (32 bits: a frame has at most 2^30 pixels) -- formed here, so that ONE register, not the
  // pixel's coordinates, lives through the two load groups below (the kernel runs at the 64 registers of eight waves per SIMD)
  uint32_t opix = (uint32_t)(out_tile_row * TILE_H + y) * (uint32_t)fp.width + (uint32_t)gx;
  asm volatile("" : "+v"(opix));
  const size_t o = (size_t)opix;

  // ---- The dependent loads between a fragment and its colour, in TWO round trips ----
  //   A  the HEAD of the primitive record (planes, 1/w, the three texture coordinates, the material binding: 80 bytes) and,
  //      for a fragment of a clipped primitive, its sub-triangle's clip slot -- whose index the fragment word carries, so
  //      nothing has to be read first to know whether and where
  //   B  the texels (four 12-byte taps, addressed from the interpolated uv) TOGETHER WITH the body of the record (the other
  //      twelve varyings of each vertex: 144 bytes)
  // Written as one sequence the compiler serialised it into five (round 3's build, by its listing: clip_base -> planes ->
  // barycentrics -> varyings -> material pointer -> texels; it sinks every load to its first use to stay within 64
  // registers): each group below ends in a fence -- an empty asm that takes every loaded value -- so that all loads of the
  // group are in flight before the first of them is waited for.
  // Two forms of the same statements:
  //  * every fragment of the wave belongs to ONE (sub-)triangle -- the ground plane's waves, about half of C3's: record and
  //    clip slot come ONCE through the scalar cache (s_load; constant address space) instead of 64 lanes gathering them;
  //  * otherwise each lane gathers the record of its own fragment's primitive (neighbouring pixels share primitives, so
  //    the loads of a wave hit few L1 lines).
  float a[kNumVary];
  uint32_t packed_dims, material;
  const uint8_t *packed_texels;
  float b0, b1, b2;  // perspective-correct barycentrics with respect to the (unclipped) primitive
  // perspective-correct barycentrics from the screen-space planes of the (sub-)triangle
  auto barycentrics = [&](const PlaneHead &h, bool clipped, const float (&cb)[3][3]) {
    const int Xc = gx * 256 + 128, Yc = gy * 256 + 128;
    const float dxp = (float)(Xc - h.X0), dyp = (float)(Yc - h.Y0);
    const float l1 = fmaf(h.l1dx, dxp, h.l1dy * dyp);
    const float l2 = fmaf(h.l2dx, dxp, h.l2dy * dyp);
    const float l0 = (1.0f - l1) - l2;
    const float u0 = l0 * h.rw0, u1 = l1 * h.rw1, u2 = l2 * h.rw2;
    const float r = bb_rcp((u0 + u1) + u2);
    b0 = u0 * r; b1 = u1 * r; b2 = u2 * r;
    if (clipped) {  // barycentrics with respect to the unclipped primitive
      const float c0 = fmaf(b2, cb[2][0], fmaf(b1, cb[1][0], b0 * cb[0][0]));
      const float c1 = fmaf(b2, cb[2][1], fmaf(b1, cb[1][1], b0 * cb[0][1]));
      const float c2 = fmaf(b2, cb[2][2], fmaf(b1, cb[1][2], b0 * cb[0][2]));
      b0 = c0; b1 = c1; b2 = c2;
    }
  };
  // texels of the packed material (one set of taps, four 12-byte loads of 9-byte records) -- global loads, issued in group B
  BilinearTaps tp = {};
  uint32_t t00[3] = {}, t10[3] = {}, t01[3] = {}, t11[3] = {};
  // (Issued by EVERY lane, without a branch around them: a lane whose material is not packed -- maps of different sizes, the
  //  per-map path further down -- reads twelve bytes of its own record instead and ignores them.  Behind a branch the loads
  //  still in flight at the join are unknown to the compiler, and it waits for all of them where the record's first part
  //  would do.)
  auto issue_taps = [&](const void *harmless) {
    const float u = BB_ABLATE(8u) ? 0.5f : a[0], v = BB_ABLATE(8u) ? 0.5f : a[1];
    const bool packed = !MIXED || packed_dims != 0u;
    tp = bilinear_taps<true>(u, v, packed ? (int)(packed_dims & 0xFFFFu) : 1, packed ? (int)(packed_dims >> 16) : 1);
    const GlobalBytes tb = packed ? (GlobalBytes)packed_texels : (GlobalBytes)harmless;
    const u32x3 q00 = *(GlobalTap)(tb + texel_offset(tp.o00)), q10 = *(GlobalTap)(tb + texel_offset(tp.o10));
    const u32x3 q01 = *(GlobalTap)(tb + texel_offset(tp.o01)), q11 = *(GlobalTap)(tb + texel_offset(tp.o11));
    t00[0] = q00.x; t00[1] = q00.y; t00[2] = q00.z;
    t10[0] = q10.x; t10[1] = q10.y; t10[2] = q10.z;
    t01[0] = q01.x; t01[1] = q01.y; t01[2] = q01.z;
    t11[0] = q11.x; t11[1] = q11.y; t11[2] = q11.z;
  };
  {
    const uint32_t clip1 = (uint32_t)(frag >> (32 + kFragPixBits));  // clip-arena slot + 1 of a clipped sub-triangle, 0: none / unknown
    const uint32_t ref_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)ref);
    if ((__ballot(ref != ref_u) == 0ull && !BB_ABLATE(4096u)) || BB_ABLATE(8192u)) {
      // ---- uniform wave: scalar loads (the constant address space is what makes them scalar -- and keeps the compiler from
      // merging the two forms back into one that gathers: both buffers were written by k_geometry and are read-only here) ----
      typedef const ShadeRec __attribute__((address_space(4))) *ConstRec;
      typedef const ClipSlot __attribute__((address_space(4))) *ConstClip;
      typedef const PlaneHead __attribute__((address_space(4))) *ConstHead;
      const ConstRec rp = (ConstRec)recs + (BB_ABLATE(16u) ? 0u : (ref_u >> 3));
      const uint32_t clip1_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)clip1);
      bool clipped = clip1_u != 0u;
      const ConstClip cp = (ConstClip)clip_arena + (clipped ? clip1_u - 1u : 0u);
      const ConstHead hp = clipped ? &cp->h : &rp->h;
      // group A
      PlaneHead h;
      h.X0 = hp->X0; h.Y0 = hp->Y0; h.l1dx = hp->l1dx; h.l1dy = hp->l1dy; h.l2dx = hp->l2dx; h.l2dy = hp->l2dy;
      h.rw0 = hp->rw0; h.rw1 = hp->rw1; h.rw2 = hp->rw2;
      float uv[3][2], cb[3][3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { uv[k][0] = rp->uv[k][0]; uv[k][1] = rp->uv[k][1]; }
#pragma unroll
      for (int jj = 0; jj < 3; ++jj)
#pragma unroll
        for (int k = 0; k < 3; ++k) cb[jj][k] = cp->bary[jj][k];  // (slot 0 when the fragment is not clipped: loaded, not used)
      packed_dims = rp->packed_dims;
      material = rp->material;
      unsigned long long packed_bits = (unsigned long long)(uintptr_t)rp->packed;
      uint32_t clip_base = rp->clip_base;
      asm volatile("" :: "s"(h.X0), "s"(h.Y0), "s"(h.l1dx), "s"(h.l1dy), "s"(h.l2dx), "s"(h.l2dy), "s"(h.rw0), "s"(h.rw1), "s"(h.rw2), "s"(uv[0][0]), "s"(uv[0][1]), "s"(uv[1][0]), "s"(uv[1][1]), "s"(uv[2][0]), "s"(uv[2][1]), "s"(packed_dims), "s"(material), "s"(packed_bits), "s"(clip_base), "s"(cb[0][0]), "s"(cb[0][1]), "s"(cb[0][2]), "s"(cb[1][0]), "s"(cb[1][1]), "s"(cb[1][2]), "s"(cb[2][0]), "s"(cb[2][1]), "s"(cb[2][2]) : "memory");
      packed_texels = (const uint8_t *)(uintptr_t)packed_bits;
      if (!clipped && clip_base != kNotClipped) {
        // a clipped primitive whose slot k_raster could not put into the fragment word (more than 32 clipped sub-triangles
        // on one tile): found through the record, one round trip later
        const ConstClip cq = (ConstClip)clip_arena + (clip_base + (ref_u & 7u));
        h.X0 = cq->h.X0; h.Y0 = cq->h.Y0; h.l1dx = cq->h.l1dx; h.l1dy = cq->h.l1dy; h.l2dx = cq->h.l2dx; h.l2dy = cq->h.l2dy;
        h.rw0 = cq->h.rw0; h.rw1 = cq->h.rw1; h.rw2 = cq->h.rw2;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
          for (int k = 0; k < 3; ++k) cb[jj][k] = cq->bary[jj][k];
        clipped = true;
      }
      barycentrics(h, clipped, cb);
      a[0] = fmaf(b2, uv[2][0], fmaf(b1, uv[1][0], b0 * uv[0][0]));
      a[1] = fmaf(b2, uv[2][1], fmaf(b1, uv[1][1], b0 * uv[0][1]));
      BB_KSTAMP(2);  // head (+ clip slot) arrived, barycentrics and uv done
      // group B: the texel taps first (vector loads, the long pole), the body of the record behind them (scalar loads)
      issue_taps((const void *)(const ShadeRec *)(uintptr_t)rp);
      float body[kNumBodyVary][3];
#pragma unroll
      for (int jj = 0; jj < kNumBodyVary; ++jj)
#pragma unroll
        for (int k = 0; k < 3; ++k) body[jj][k] = rp->vary[jj][k];
#pragma unroll
      for (int q = 0; q < kNumBodyVary; q += 4)
        asm volatile("" :: "s"(body[q][0]), "s"(body[q][1]), "s"(body[q][2]), "s"(body[q + 1][0]), "s"(body[q + 1][1]), "s"(body[q + 1][2]), "s"(body[q + 2][0]), "s"(body[q + 2][1]), "s"(body[q + 2][2]), "s"(body[q + 3][0]), "s"(body[q + 3][1]), "s"(body[q + 3][2]) : "memory");
#pragma unroll
      for (int jj = 0; jj < kNumBodyVary; ++jj) a[2 + jj] = fmaf(b2, body[jj][2], fmaf(b1, body[jj][1], b0 * body[jj][0]));
    } else {
      // ---- each lane gathers the record of its own fragment's primitive ----
      const ShadeRec *rp = recs + prim;
      bool clipped = clip1 != 0u;
      const ClipSlot *cp = clip_arena + (clipped ? clip1 - 1u : 0u);
      const PlaneHead *hp = clipped ? &cp->h : &rp->h;
      // group A: three loads of the head (from the record or the clip slot), three of the rest of the record's head,
      // and, for the lanes of clipped fragments, three of the slot's barycentrics
      PlaneHead h = *hp;
      float uv[3][2], cb[3][3] = {};
#pragma unroll
      for (int k = 0; k < 3; ++k) { uv[k][0] = rp->uv[k][0]; uv[k][1] = rp->uv[k][1]; }
      packed_dims = rp->packed_dims;
      material = rp->material;
      unsigned long long packed_bits = (unsigned long long)(uintptr_t)rp->packed;
      uint32_t clip_base = rp->clip_base;
      if (clipped) {
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
          for (int k = 0; k < 3; ++k) cb[jj][k] = cp->bary[jj][k];
      }
      asm volatile("" :: "v"(h.X0), "v"(h.Y0), "v"(h.l1dx), "v"(h.l1dy), "v"(h.l2dx), "v"(h.l2dy), "v"(h.rw0), "v"(h.rw1), "v"(h.rw2), "v"(uv[0][0]), "v"(uv[0][1]), "v"(uv[1][0]), "v"(uv[1][1]), "v"(uv[2][0]), "v"(uv[2][1]), "v"(packed_dims), "v"(material), "v"(packed_bits), "v"(clip_base), "v"(cb[0][0]), "v"(cb[0][1]), "v"(cb[0][2]), "v"(cb[1][0]), "v"(cb[1][1]), "v"(cb[1][2]), "v"(cb[2][0]), "v"(cb[2][1]), "v"(cb[2][2]) : "memory");
      packed_texels = (const uint8_t *)(uintptr_t)packed_bits;
      const bool late = !clipped && clip_base != kNotClipped;  // (see the uniform form)
      if (__builtin_expect(__ballot(late) != 0ull, 0)) {
        if (late) {
          const ClipSlot cs = clip_arena[clip_base + (ref & 7u)];
          h = cs.h;
#pragma unroll
          for (int jj = 0; jj < 3; ++jj)
#pragma unroll
            for (int k = 0; k < 3; ++k) cb[jj][k] = cs.bary[jj][k];
          clipped = true;
        }
      }
      barycentrics(h, clipped, cb);
      a[0] = fmaf(b2, uv[2][0], fmaf(b1, uv[1][0], b0 * uv[0][0]));
      a[1] = fmaf(b2, uv[2][1], fmaf(b1, uv[1][1], b0 * uv[0][1]));
      BB_KSTAMP(2);  // head (+ clip slot) arrived, barycentrics and uv done
      // group B: the first 80 bytes of the record's body (varyings 0..5 whole and two thirds of the sixth), the four texel
      // taps behind them, and -- as soon as the first part has arrived and its six varyings are interpolated, their eighteen
      // registers free again -- the other 64 bytes of the body.  Loads return in order: the second part (the same cache
      // lines as the first: L1 hits) comes back behind the taps, so the group is still one round trip long, and its peak is
      // 20 + 12 loaded registers instead of 36 + 12 (the kernel has 64).
      const float *bp = &rp->vary[0][0];
      float p1[20];
#pragma unroll
      for (int q = 0; q < 20; ++q) p1[q] = bp[q];
      issue_taps(rp);
      asm volatile("" :: "v"(p1[0]), "v"(p1[1]), "v"(p1[2]), "v"(p1[3]), "v"(p1[4]), "v"(p1[5]), "v"(p1[6]), "v"(p1[7]), "v"(p1[8]), "v"(p1[9]), "v"(p1[10]), "v"(p1[11]), "v"(p1[12]), "v"(p1[13]), "v"(p1[14]), "v"(p1[15]), "v"(p1[16]), "v"(p1[17]), "v"(p1[18]), "v"(p1[19]) : "memory");
#pragma unroll
      for (int jj = 0; jj < 6; ++jj) a[2 + jj] = fmaf(b2, p1[3 * jj + 2], fmaf(b1, p1[3 * jj + 1], b0 * p1[3 * jj]));
      asm volatile("" :: "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");  // (the second part is asked for HERE, not earlier)
      float p2[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) p2[q] = bp[20 + q];
      asm volatile("" :: "v"(p2[0]), "v"(p2[1]), "v"(p2[2]), "v"(p2[3]), "v"(p2[4]), "v"(p2[5]), "v"(p2[6]), "v"(p2[7]), "v"(p2[8]), "v"(p2[9]), "v"(p2[10]), "v"(p2[11]), "v"(p2[12]), "v"(p2[13]), "v"(p2[14]), "v"(p2[15]) : "memory");
      a[8] = fmaf(b2, p2[0], fmaf(b1, p1[19], b0 * p1[18]));
#pragma unroll
      for (int jj = 7; jj < kNumBodyVary; ++jj) a[2 + jj] = fmaf(b2, p2[3 * jj - 18], fmaf(b1, p2[3 * jj - 19], b0 * p2[3 * jj - 20]));
    }
  }
  // the taps are used from here on: pin them behind the body's fence (they were issued in front of it)
  asm volatile("" :: "v"(t00[0]), "v"(t00[1]), "v"(t00[2]), "v"(t10[0]), "v"(t10[1]), "v"(t10[2]), "v"(t01[0]), "v"(t01[1]), "v"(t01[2]), "v"(t11[0]), "v"(t11[1]), "v"(t11[2]) : "memory");
  __builtin_amdgcn_s_setprio(0);  // (the loads are out: filtering and the light loop at the default priority)

  // texture filtering, forward_brdf.frag:16-22
  const float u = BB_ABLATE(8u) ? 0.5f : a[0], v = BB_ABLATE(8u) ? 0.5f : a[1];
  f3 albedo, normal;
  float metallic, roughness, ao;
  if (!MIXED || packed_dims != 0u) {
    albedo.x = filter_channel(t00[0], t10[0], t01[0], t11[0], 0, tp.fx, tp.fy);
    albedo.y = filter_channel(t00[0], t10[0], t01[0], t11[0], 8, tp.fx, tp.fy);
    albedo.z = filter_channel(t00[0], t10[0], t01[0], t11[0], 16, tp.fx, tp.fy);
    metallic = filter_channel(t00[0], t10[0], t01[0], t11[0], 24, tp.fx, tp.fy);
    roughness = filter_channel(t00[1], t10[1], t01[1], t11[1], 24, tp.fx, tp.fy);
    ao = filter_channel(t00[2], t10[2], t01[2], t11[2], 0, tp.fx, tp.fy);
    if (sp.enable_normal_map != 0) {
      const f3 nt = mk3(fmaf(filter_channel(t00[1], t10[1], t01[1], t11[1], 0, tp.fx, tp.fy), 2.0f, -1.0f),
                        fmaf(filter_channel(t00[1], t10[1], t01[1], t11[1], 8, tp.fx, tp.fy), 2.0f, -1.0f),
                        fmaf(filter_channel(t00[1], t10[1], t01[1], t11[1], 16, tp.fx, tp.fy), 2.0f, -1.0f));
      // vTBN * nt, vTBN = mat3(T, B, N)
      normal.x = fmaf(a[5], nt.z, fmaf(a[11], nt.y, a[8] * nt.x));
      normal.y = fmaf(a[6], nt.z, fmaf(a[12], nt.y, a[9] * nt.x));
      normal.z = fmaf(a[7], nt.z, fmaf(a[13], nt.y, a[10] * nt.x));
    } else {
      normal = DEFERRED ? mk3(a[5], a[6], a[7]) : normalize3(mk3(a[5], a[6], a[7]));  // gbuffer.frag:29 / forward :24
    }
  } else {
    // maps of different sizes: one set of taps per map
    const MaterialDesc &md = materials[material];
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
    if (sp.enable_normal_map != 0) {
      const TexDesc &td = md.maps[kMapNormal];
      BilinearTaps tp = bilinear_taps(u, v, td.w, td.h);
      const uint32_t *tx32 = reinterpret_cast<const uint32_t *>(td.texels);
      uint32_t t00 = tx32[tp.o00], t10 = tx32[tp.o10], t01 = tx32[tp.o01], t11 = tx32[tp.o11];
      f3 nt = mk3(fmaf(filter_channel(t00, t10, t01, t11, 0, tp.fx, tp.fy), 2.0f, -1.0f),
                  fmaf(filter_channel(t00, t10, t01, t11, 8, tp.fx, tp.fy), 2.0f, -1.0f),
                  fmaf(filter_channel(t00, t10, t01, t11, 16, tp.fx, tp.fy), 2.0f, -1.0f));
      normal.x = fmaf(a[5], nt.z, fmaf(a[11], nt.y, a[8] * nt.x));
      normal.y = fmaf(a[6], nt.z, fmaf(a[12], nt.y, a[9] * nt.x));
      normal.z = fmaf(a[7], nt.z, fmaf(a[13], nt.y, a[10] * nt.x));
    } else {
      normal = DEFERRED ? mk3(a[5], a[6], a[7]) : normalize3(mk3(a[5], a[6], a[7]));
    }
  }

  BB_KSTAMP(3);  // varyings, taps arrived, filtered
  float4 color;
  if (DEFERRED) {
    // gbuffer.frag:24-32 into four RGBA16F attachments (binary16, round to nearest even), then brdf.frag:12-73 on
    // the pixel's own texel: fused, the texel only goes to memory when somebody asked to see it
    f3 P = mk3(bb_half_round(a[2]), bb_half_round(a[3]), bb_half_round(a[4]));
    normal = mk3(bb_half_round(normal.x), bb_half_round(normal.y), bb_half_round(normal.z));
    albedo = mk3(bb_half_round(albedo.x), bb_half_round(albedo.y), bb_half_round(albedo.z));
    metallic = bb_half_round(metallic); roughness = bb_half_round(roughness); ao = bb_half_round(ao);
    if (gbuffer && valid) {
      const MaterialDesc &md = materials[material];
      const TexDesc &td = md.maps[kMapHeight];
      BilinearTaps tp = bilinear_taps(u, v, td.w, td.h);
      const uint32_t *tx32 = reinterpret_cast<const uint32_t *>(td.texels);
      const float height = bb_half_round(filter_channel(tx32[tp.o00], tx32[tp.o10], tx32[tp.o01], tx32[tp.o11], 0, tp.fx, tp.fy));
      _Float16 g[16] = {(_Float16)P.x, (_Float16)P.y, (_Float16)P.z, (_Float16)1.0f,
                        (_Float16)normal.x, (_Float16)normal.y, (_Float16)normal.z, (_Float16)0.0f,
                        (_Float16)albedo.x, (_Float16)albedo.y, (_Float16)albedo.z, (_Float16)0.0f,
                        (_Float16)metallic, (_Float16)roughness, (_Float16)ao, (_Float16)height};
      uint4 *dst = reinterpret_cast<uint4 *>(gbuffer) + 2 * ((size_t)gy * (size_t)fp.width + (size_t)gx);
      uint4 lo, hi;
      __builtin_memcpy(&lo, g, 16);
      __builtin_memcpy(&hi, g + 8, 16);
      dst[0] = lo;
      dst[1] = hi;
    }
    if (fp.gbuffer_view >= 0) {
      // buffer_visualize.frag:8-12 instead of brdf.frag (recordCommand, src/main.cpp:96-121): the rgb of one attachment
      const f3 shown = fp.gbuffer_view == 0 ? P : (fp.gbuffer_view == 1 ? normal : (fp.gbuffer_view == 2 ? albedo : mk3(metallic, roughness, ao)));
      color = make_float4(shown.x, shown.y, shown.z, 1.0f);
    } else {
      color = light_surface(sp, lights_c, P, normal, albedo, metallic, roughness, ao);
    }
  } else {
    color = light_surface(sp, lights_c, mk3(a[2], a[3], a[4]), normal, albedo, metallic, roughness, ao);
  }
  BB_KSTAMP(4);  // light loop done
  if (BB_ABLATE(2u)) color = make_float4(1.f, 1.f, 1.f, 1.f);
  if (valid) {
    if (PRESENT) store_pixel(&out8[o], present_pixel(color.x, color.y, color.z, *tables, sp.tone_enable, sp.exposure, 1));
    else store_pixel(&out[o], color);
  }
#ifdef BB_STAMPS
  {
    BB_KSTAMP(5);
    const uint32_t wv = blockIdx.x * (uint32_t)kShadeWaves + (threadIdx.x >> 6);
    if (!TAIL && lane == 0 && wv >= 30000u && wv < 34096u) {  // waves from the middle of the launch: steady state
      unsigned long long *d = g_shade_stamps + (size_t)(wv - 30000u) * 8;
      for (int q = 0; q < 5; ++q) d[q] = st_t[q + 1] - st_t[q];
      d[5] = 1;
    }
  }
#endif
  j += gridDim.x * (uint32_t)kShadeWaves;
  asm volatile("" ::: "memory");
  } while (TAIL && j < n_items);
}

// ------------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------------

// bb_rcp / bb_rcp_normal against the IEEE division and max0 against its select form, for every float with bit pattern
// in [lo, hi) and its negation: number of differing results
__global__ void k_selftest_rcp(unsigned long long *__restrict__ mismatches, uint32_t lo, uint32_t hi) {
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long bad = 0;
  for (uint64_t u = (uint64_t)lo + blockIdx.x * blockDim.x + threadIdx.x; u < hi; u += stride) {
    const float x = __uint_as_float((uint32_t)u);
    const float a = bb_rcp(x), b = 1.0f / x, c = bb_rcp(-x);
    bad += (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) || (__float_as_uint(c) != (__float_as_uint(b) ^ 0x80000000u) && !(c != c && b != b));
    // the unguarded variant on its documented domain [2^-100, 2^100]
    if (u >= 0x0D800000u && u <= 0x71800000u) bad += __float_as_uint(bb_rcp_normal(x)) != __float_as_uint(b);
    // max0 == (a > 0 ? a : 0), both signs, every pattern (NaN -> +0, -0 -> +0)
    const float nx = -x;
    bad += __float_as_uint(max0(x)) != __float_as_uint(x > 0.0f ? x : 0.0f);
    bad += __float_as_uint(max0(nx)) != __float_as_uint(nx > 0.0f ? nx : 0.0f);
    bad += __float_as_uint(__builtin_fmaxf(x, 0.001f)) != __float_as_uint(!(x > 0.001f) ? 0.001f : x);
    bad += __float_as_uint(__builtin_fmaxf(nx, 0.001f)) != __float_as_uint(!(nx > 0.001f) ? 0.001f : nx);
  }
  if (bad) atomicAdd(mismatches, bad);
}

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

// ------------------------------------------------------------------------------------------------
// presentation: HDR attachment (binary16) -> hdr_tone_mapping.frag:9-18 -> sRGB UNORM8.
// Same fixed sequences as the CPU oracle (binary16 rounding, exp, threshold table): byte-exact.
// ------------------------------------------------------------------------------------------------

// hdr_tone_mapping.frag:9-18 on the fp32 frame, in place
__global__ void k_tone_map(float4 *__restrict__ frame, size_t n, int enable, float exposure) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 c = frame[i];
  if (enable) {
    c.x = 1.0f - bb_exp(-c.x * exposure);
    c.y = 1.0f - bb_exp(-c.y * exposure);
    c.z = 1.0f - bb_exp(-c.z * exposure);
  }
  c.w = 1.0f;
  frame[i] = c;
}

// 16 B read + 4 B written per pixel: 20 algorithmic bytes, HBM-bound by construction.  A workgroup loads the 4.3 KB
// of tables once and converts 2048 pixels (eight coalesced rounds of 256).
constexpr int kPresentThreads = 256;
constexpr int kPresentPerThread = 8;
__global__ __launch_bounds__(kPresentThreads) void k_present(const float4 *__restrict__ frame, uint32_t *__restrict__ out_rgba8,
                                                              size_t n, const SrgbTables *__restrict__ tables, int enable,
                                                              float exposure, int hdr16) {
  __shared__ SrgbTables t;
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(tables);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&t);
    for (uint32_t w = threadIdx.x; w < sizeof(SrgbTables) / 4; w += kPresentThreads) dst[w] = src[w];
  }
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * (kPresentThreads * kPresentPerThread) + threadIdx.x;
#pragma unroll 2
  for (int j = 0; j < kPresentPerThread; ++j) {
    const size_t i = base + (size_t)j * kPresentThreads;
    if (i >= n) break;
    // (the fp32 frame is read once, the presented image written once: both non-temporal)
    const v4f c = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(frame) + i);
    const uint32_t px = present_pixel(c.x, c.y, c.z, t, enable, exposure, hdr16);
    store_pixel(&out_rgba8[i], px);
  }
}

// Lossless 12.1-byte form of a shard for the all-gather (a quarter less than RGBA32F over links that are the bottleneck
// at N > 1): alpha is 1.0 where geometry was shaded and 0.0 on cleared pixels (forward_brdf.frag:75 writes 1, the clear
// colour is 0, src/main.cpp:84; the deferred path writes 1 everywhere), so it travels as one bit.
//   packed block of a rank = rgb[n][3] float, then (8-byte aligned) one 64-bit mask per 64 pixels, n = shard_rows * width
__global__ void k_pack_shard(const float4 *__restrict__ shard, float *__restrict__ rgb, unsigned long long *__restrict__ mask,
                             size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    p = shard[i];
    rgb[3 * i + 0] = p.x;
    rgb[3 * i + 1] = p.y;
    rgb[3 * i + 2] = p.z;
  }
  const unsigned long long m = __ballot(live && __float_as_uint(p.w) == 0x3F800000u);
  if ((threadIdx.x & 63) == 0 && live) mask[i >> 6] = m;
}

// [world] packed blocks -> row-major RGBA32F frame (the un-interleave of k_unpack_gathered on the packed form)
__global__ void k_unpack_gathered_packed(const uint8_t *__restrict__ gathered, float4 *__restrict__ frame, int width, int height,
                                         int world, int band_rows, int shard_rows, size_t block_bytes, size_t mask_offset) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t n = (size_t)width * (size_t)height;
  if (i >= n) return;
  int y = (int)(i / (size_t)width), x = (int)(i - (size_t)y * width);
  int band = y / band_rows, r = y - band * band_rows;
  int rank = band % world, lb = band / world;
  const size_t j = ((size_t)lb * band_rows + r) * (size_t)width + x;
  const uint8_t *block = gathered + (size_t)rank * block_bytes;
  const float *rgb = reinterpret_cast<const float *>(block) + 3 * j;
  const unsigned long long m = reinterpret_cast<const unsigned long long *>(block + mask_offset)[j >> 6];
  frame[i] = make_float4(rgb[0], rgb[1], rgb[2], ((m >> (j & 63)) & 1ull) ? 1.0f : 0.0f);
}

// The reference's own HDR attachment format as a wire form (R16G16B16A16_SFLOAT, src/render.h:94, src/main.cpp:463-472):
// 8 bytes per pixel, half of the fp32 shard.  Lossy -- every channel is rounded to the nearest binary16 value (ties to even,
// the rounding an attachment write performs: bb_half_round) -- and therefore a separate output, never the default.
__global__ void k_pack_shard_half(const float4 *__restrict__ shard, uint2 *__restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = shard[i];
  const _Float16 h[4] = {(_Float16)p.x, (_Float16)p.y, (_Float16)p.z, (_Float16)p.w};
  uint2 w;
  __builtin_memcpy(&w, h, 8);
  out[i] = w;
}

// [world][shard_rows][width] RGBA16F -> row-major frame, widened to RGBA32F (every binary16 value is a binary32 value)
__global__ void k_unpack_gathered_half(const uint2 *__restrict__ gathered, float4 *__restrict__ frame, int width, int height,
                                       int world, int band_rows, int shard_rows) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t n = (size_t)width * (size_t)height;
  if (i >= n) return;
  int y = (int)(i / (size_t)width), x = (int)(i - (size_t)y * width);
  int band = y / band_rows, r = y - band * band_rows;
  int rank = band % world, lb = band / world;
  const uint2 w = gathered[((size_t)rank * shard_rows + (size_t)lb * band_rows + r) * (size_t)width + x];
  _Float16 h[4];
  __builtin_memcpy(h, &w, 8);
  frame[i] = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}

// [world][shard_rows][width] RGBA8 -> row-major presented frame (same un-interleave as k_unpack_gathered)
__global__ void k_unpack_gathered_rgba8(const uint32_t *__restrict__ gathered, uint32_t *__restrict__ frame, int width,
                                        int height, int world, int band_rows, int shard_rows) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t n = (size_t)width * (size_t)height;
  if (i >= n) return;
  int y = (int)(i / (size_t)width), x = (int)(i - (size_t)y * width);
  int band = y / band_rows, r = y - band * band_rows;
  int rank = band % world, lb = band / world;
  frame[i] = gathered[((size_t)rank * shard_rows + (size_t)lb * band_rows + r) * (size_t)width + x];
}

// The peer form of the exchange as ONE kernel (bbr_push_shard, option "push_mode" 1): a workgroup loads a piece of this
// rank's block once and stores it into the same place of EVERY peer's gather buffer -- world - 1 stores per load, each going
// out over its own xGMI link, so all links of the full mesh carry a block at the same time (the direct pattern of SURVEY
// section 5 / 8(e): shard / link bandwidth, where copies queued one behind the other take world - 1 times that).  The
// peers' buffers are peer-accessible device memory (hipDeviceEnablePeerAccess or an opened IPC handle).
constexpr int kMaxPushPeers = 15;
struct PushTargets {
  void *dst[kMaxPushPeers];
};
constexpr int kPushThreads = 256;
template <typename V>
__global__ __launch_bounds__(kPushThreads) void k_push_block(const V *__restrict__ src, PushTargets t, int n_targets, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kPushThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kPushThreads) {
    const V v = src[i];
#pragma unroll 1
    for (int k = 0; k < n_targets; ++k) reinterpret_cast<V *>(t.dst[k])[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// k_shade_overlay: light.frag / gizmo.frag on the fragments of the overlay pass, written sRGB-encoded into the
// presented RGBA8 image (the overlay subpass draws into the swapchain image, src/main.cpp:128-171).
// ------------------------------------------------------------------------------------------------
template <int TILE_W, int TILE_H>
__global__ __launch_bounds__(kShadeThreads) void k_shade_overlay(
    FrameParams fp, const ShadeRec *__restrict__ recs, const ClipSlot *__restrict__ clip_arena,
    const unsigned long long *__restrict__ frags, const uint32_t *__restrict__ frag_count,
    const SrgbTables *__restrict__ tables, uint32_t *__restrict__ out_rgba8) {
  constexpr int TILE_PIXELS = TILE_W * TILE_H;
  constexpr int CHUNKS = TILE_PIXELS / kShadeThreads;
  const int tx = blockIdx.x / CHUNKS, chunk = blockIdx.x - tx * CHUNKS;
  const int ty = blockIdx.y;
  if (ty >= fp.tiles_y) return;
  const uint32_t tile = (uint32_t)ty * (uint32_t)fp.tiles_x + (uint32_t)tx;
  const uint32_t i = (uint32_t)chunk * kShadeThreads + threadIdx.x;
  const uint32_t n_frag = frag_count[tile];
  if (i >= n_frag) return;
  const unsigned long long frag = frags[(size_t)tile * TILE_PIXELS + i];
  const uint32_t ref = (uint32_t)frag, prim = ref >> 3;
  int x, y;
  tile_pixel<TILE_W>((int)(frag >> 32) & (TILE_PIXELS - 1), x, y);
  const int gx = tx * TILE_W + x, gy = ty * TILE_H + y;
  const ShadeRec pa = recs[prim];
  // perspective-correct barycentrics: the same statements as k_shade
  const bool clipped = pa.clip_base != kNotClipped;
  const ClipSlot *cs = clipped ? &clip_arena[pa.clip_base + (ref & 7u)] : nullptr;
  const PlaneHead h = clipped ? cs->h : pa.h;
  const int Xc = gx * 256 + 128, Yc = gy * 256 + 128;
  const float dxp = (float)(Xc - h.X0), dyp = (float)(Yc - h.Y0);
  const float l1 = fmaf(h.l1dx, dxp, h.l1dy * dyp);
  const float l2 = fmaf(h.l2dx, dxp, h.l2dy * dyp);
  const float l0 = (1.0f - l1) - l2;
  const float u0 = l0 * h.rw0, u1 = l1 * h.rw1, u2 = l2 * h.rw2;
  const float r = bb_rcp((u0 + u1) + u2);
  float b0 = u0 * r, b1 = u1 * r, b2 = u2 * r;
  if (clipped) {
    const float c0 = fmaf(b2, cs->bary[2][0], fmaf(b1, cs->bary[1][0], b0 * cs->bary[0][0]));
    const float c1 = fmaf(b2, cs->bary[2][1], fmaf(b1, cs->bary[1][1], b0 * cs->bary[0][1]));
    const float c2 = fmaf(b2, cs->bary[2][2], fmaf(b1, cs->bary[1][2], b0 * cs->bary[0][2]));
    b0 = c0; b1 = c1; b2 = c2;
  }
  // varyings 0..5 of the overlay programs (k_geometry<..., OVERLAY>): 0, 1 sit in the record's head, 2..5 in its body
  float a[6];
#pragma unroll
  for (int k = 0; k < 2; ++k) a[k] = fmaf(b2, pa.uv[2][k], fmaf(b1, pa.uv[1][k], b0 * pa.uv[0][k]));
#pragma unroll
  for (int k = 0; k < 4; ++k) a[2 + k] = fmaf(b2, pa.vary[k][2], fmaf(b1, pa.vary[k][1], b0 * pa.vary[k][0]));
  float col[3] = {a[0], a[1], a[2]};  // light.frag: outColor = vec4(vColor, 1)
  if (pa.material == 2u) {             // gizmo.frag:10-17: L = -(0,0,1); diff = max(dot(L, normalize(vNormal)), 0)
    const f3 N = normalize3(mk3(a[3], a[4], a[5]));
    const float diff = max0(dot3(mk3(-0.0f, -0.0f, -1.0f), N));
    col[0] = a[0] * diff; col[1] = a[1] * diff; col[2] = a[2] * diff;
  }
  uint32_t px = 0xFF000000u;
#pragma unroll
  for (int k = 0; k < 3; ++k) px |= srgb8(col[k], *tables) << (8 * k);
  out_rgba8[(size_t)gy * (size_t)fp.width + (size_t)gx] = px;
}

}  // namespace bbr
