// bb_types.h -- byte layouts of the forward path's inputs, exactly as the reference hands them to Vulkan,
// plus the device-side records of the MI355X pipeline.  Shared by the HIP kernels, the C-ABI and the
// C++ Scene/Camera/drawFrame shim.
//
// Reference layouts (sizes checked against the reference's headers compiled in the authoring container):
//   Vertex            src/render.h:112-117   44 B   Pos@0 UV@12 Normal@20 Tangent@32
//   InstanceBlock     src/render.h:96-99    128 B   ModelMat@0 InvModelMat@64
//   Light             src/render.h:310-318   64 B   Pos@0 Type@12 Dir@16 Intensity@28 Color@32 Inner@44 Outer@48
//   FrameUniformBlock src/render.h:321-327 6432 B   NumLights@0 Lights@16 ... EnableToneMapping@6420 Exposure@6424
//   ViewUniformBlock  src/render.h:329-334  144 B   ViewMat@0 ProjMat@64 ViewPos@128 EnableNormalMap@140
#pragma once
#include <cstddef>
#include <cstdint>

namespace bbr {

struct Mat4 {
  float M[4][4];  // column-major M[col][row], src/vector_math.h:64
};

struct Vertex {
  float pos[3];
  float uv[2];
  float normal[3];
  float tangent[3];
};

struct InstanceBlock {
  Mat4 model;
  Mat4 inv_model;
};

struct alignas(16) Light {
  float pos[3];
  int32_t type;  // 0 point, 1 spot, 2 directional (src/shaders/standard_sets.glsl:8)
  float dir[3];
  float intensity;
  float color[3];
  float inner_cutoff;
  float outer_cutoff;
};

constexpr int kMaxNumLights = 100;

struct FrameUniformBlock {
  int32_t num_lights;
  Light lights[kMaxNumLights];
  int32_t visualized_gbuffer_attachment_index;
  int32_t enable_tone_mapping;
  float exposure;
};

struct ViewUniformBlock {
  Mat4 view;
  Mat4 proj;
  float view_pos[3];
  int32_t enable_normal_map;
};

static_assert(sizeof(Mat4) == 64, "Mat4");
static_assert(sizeof(Vertex) == 44 && offsetof(Vertex, uv) == 12 && offsetof(Vertex, normal) == 20 &&
                  offsetof(Vertex, tangent) == 32, "Vertex");
static_assert(sizeof(InstanceBlock) == 128, "InstanceBlock");
static_assert(sizeof(Light) == 64 && offsetof(Light, type) == 12 && offsetof(Light, dir) == 16 &&
                  offsetof(Light, intensity) == 28 && offsetof(Light, color) == 32 &&
                  offsetof(Light, inner_cutoff) == 44 && offsetof(Light, outer_cutoff) == 48, "Light");
static_assert(sizeof(FrameUniformBlock) == 6432 && offsetof(FrameUniformBlock, lights) == 16 &&
                  offsetof(FrameUniformBlock, enable_tone_mapping) == 6420 &&
                  offsetof(FrameUniformBlock, exposure) == 6424, "FrameUniformBlock");
static_assert(sizeof(ViewUniformBlock) == 144 && offsetof(ViewUniformBlock, proj) == 64 &&
                  offsetof(ViewUniformBlock, view_pos) == 128 && offsetof(ViewUniformBlock, enable_normal_map) == 140,
              "ViewUniformBlock");

// PBRMapType order, src/render.h:235-243
enum MapType { kMapAlbedo = 0, kMapMetallic, kMapRoughness, kMapAO, kMapNormal, kMapHeight, kMapCount };

// ------------------------------------------------------------------------------------------------
// device-side records (implementation intermediates; not part of the ABI)
// ------------------------------------------------------------------------------------------------

constexpr int kNumVary = 14;  // uv(2) posWorld(3) N(3) T(3) B(3)
constexpr int kSubpixelBits = 8;
constexpr int kMaxSubTris = 8;  // visibility key low word = prim*8 + sub + 1
constexpr uint32_t kMaxPrims = 1u << 29;

// One rasterisable triangle: 24.8 snapped coordinates + planes relative to vertex 0.  64 B.
// (tris[prim] is written for unclipped survivors only; sub-triangles of clipped primitives live in the clip arena.)
struct RasterTri {
  int32_t X0, Y0, X1, Y1, X2, Y2;
  float z0, dzdx, dzdy;          // NDC depth plane, per sub-pixel unit
  float l1dx, l1dy, l2dx, l2dy;  // screen-space barycentric planes
  float rw0, rw1, rw2;           // 1/w at the triangle's own vertices
};
static_assert(sizeof(RasterTri) == 64, "RasterTri");

// What k_shade needs of a (sub-)triangle before it can interpolate anything: vertex 0 in 24.8, the screen-space
// barycentric planes relative to it, 1/w at the three vertices.  36 B.  A primitive record and a clip-arena slot both START
// with one, so that a lane fetches it from either place with the same three loads.
struct PlaneHead {
  int32_t X0, Y0;
  float l1dx, l1dy, l2dx, l2dy;
  float rw0, rw1, rw2;
};
static_assert(sizeof(PlaneHead) == 36, "PlaneHead");

// Sub-triangle produced by the polygon clipper, as k_shade reads it (k_raster gets its copy of the triangle through the
// every-tile list).  80 B.
struct ClipSlot {
  PlaneHead h;
  float bary[3][3];  // own vertex j -> barycentrics with respect to the unclipped primitive
  uint32_t valid;
  uint32_t pad;
};
static_assert(sizeof(ClipSlot) == 80 && offsetof(ClipSlot, bary) == 36, "ClipSlot");

// Entry of the every-tile list: the triangle itself travels with the reference so that the raster kernel's
// per-tile classification is one load deep.  80 B.
struct BroadTri {
  RasterTri tri;
  uint32_t ref;
  uint32_t pad[3];
};
static_assert(sizeof(BroadTri) == 80, "BroadTri");

// One texel of a packed material: the five maps the forward shader reads, interleaved into a 9-byte record so that a
// bilinear tap is ONE 12-byte load (global_load_dwordx3 at an unaligned address; the last three bytes belong to the
// next record and are ignored) instead of five 4-byte loads from five arrays -- and so that a bilinear footprint costs
// 36 bytes of HBM traffic, not 64 (16-byte texels) or 80 (five RGBA8 maps):
//   byte 0..2 albedo.rgb, 3 metallic.r, 4..6 normal.xyz, 7 roughness.r, 8 ao.r
#ifndef BB_PACKED_STRIDE
#define BB_PACKED_STRIDE 9  // (12 and 16 were measured too: DESIGN.md)
#endif
constexpr uint32_t kPackedTexelBytes = BB_PACKED_STRIDE;
constexpr uint32_t kPackedTexelPad = 16;  // bytes allocated behind the last record (the 12-byte load of the last texel)

// Everything k_shade needs about one primitive, in ONE 224-byte record (56 dwords), in the order it is needed:
//   head (80 B)  what stands in front of the texel fetch: the planes and 1/w of the (unclipped) triangle, the texture
//                coordinates of its three vertices, its material binding
//   body (144 B) the other twelve varyings, varying-major ([varying][vertex]) -- fetched together WITH the texels, not in
//                front of them, in two parts (80 + 64 bytes): the first five 16-byte loads hold varyings 0..5 whole, so their
//                registers are free again before the second part is asked for
// Every lane gathers the record of its own fragment's primitive (neighbouring pixels share primitives, so the loads of a
// wave hit few L1 lines); a wave whose fragments share one primitive reads it through the scalar cache.
constexpr int kNumBodyVary = kNumVary - 2;  // posWorld(3) N(3) T(3) B(3)
struct ShadeRec {
  PlaneHead h;                   // (zero for a primitive that went through the clipper: its sub-triangles have their own)
  float uv[3][2];
  uint32_t packed_dims;          // width | height << 16, 0 = not packed
  const uint8_t *packed;         // packed material texels, 9-byte records (nullptr: use the material table)
  uint32_t material;
  uint32_t clip_base;            // first clip-arena slot of a clipped primitive, kNotClipped otherwise
  float vary[kNumBodyVary][3];   // [varying][vertex]
};
static_assert(sizeof(ShadeRec) == 224 && offsetof(ShadeRec, uv) == 36 && offsetof(ShadeRec, packed) == 64 &&
                  offsetof(ShadeRec, vary) == 80, "ShadeRec");
constexpr uint32_t kNotClipped = 0xFFFFFFFFu;
constexpr int kShadeRecDwords = sizeof(ShadeRec) / 4;

struct DrawDesc {
  const Vertex *vertices;
  const uint32_t *indices;  // nullptr => non-indexed
  const InstanceBlock *instances;
  uint32_t n_instances;
  uint32_t tris_per_instance;
  uint32_t first_prim;
  uint32_t material;
  // the material's packed form as the primitive records carry it (nullptr / 0: maps of different sizes, the material table):
  // filled in by the host, so that k_geometry has no dependent load of the material table in front of its record stores
  const uint8_t *packed;
  uint32_t packed_dims;  // width | height << 16
  uint32_t pad;
};
static_assert(sizeof(DrawDesc) == 56, "DrawDesc");
// first_prim of draws 1 .. kInlineFirstPrims travel as kernel arguments (0xFFFFFFFF: no such draw): k_geometry finds its
// draw without a load for frames of up to kInlineFirstPrims + 1 draws
constexpr int kInlineFirstPrims = 3;
struct FirstPrims {
  uint32_t v[kInlineFirstPrims];
};

struct TexDesc {
  const uint8_t *texels;  // RGBA8, row-major
  int32_t w, h;
};

struct MaterialDesc {
  TexDesc maps[kMapCount];
  const uint8_t *packed;  // 9-byte records; non-null when the five shaded maps share one size (or are defaults)
  int32_t pw, ph;
};

// per-frame allocation counters and flags (one block per frame slot).  Statistics are NOT accumulated here:
// same-address atomics from thousands of workgroups serialise; they travel as per-workgroup records instead.
struct Counters {
  uint32_t n_broad;
  uint32_t n_clip_slots;
  uint32_t overflow;  // bit0 bins, bit1 broad list, bit2 clip arena
  uint32_t bin_need;  // largest per-tile reference count seen when a bin overflowed
  uint32_t n_heavy;   // entries of the frame's heavy-tile list (k_geometry appends, k_raster starts those tiles first)
  uint32_t pad[27];
};
static_assert(sizeof(Counters) == 128, "Counters");

// k_geometry statistics of one workgroup
struct BlockStats {
  uint32_t raster_tris, clipped_prims, bin_refs;
};

struct FrameParams {
  int32_t width, height;
  float half_w, half_h;
  int32_t tiles_x, tiles_y;          // tiles over the whole frame
  uint32_t bin_cap, broad_cap, clip_cap;
  uint32_t broad_threshold;          // triangles touching more tiles than this go to the broad list
  // screen-band partition (band = band_tiles tile rows; band b belongs to rank b % world)
  int32_t rank, world, band_tiles;
  int32_t shard_rows;                // rows of the compact output when world > 1
  // heavy tiles first (k_raster): a tile one of whose bins reaches heavy_threshold references is appended to the frame's
  // heavy list by k_geometry (launch slot | class << 30), and the first heavy_rows grid rows of k_raster's launch work
  // through that list before the screen-ordered rest starts.  0 / 0: off.
  uint32_t heavy_threshold;
  int32_t heavy_rows;
  uint32_t ablate;                   // diagnostics only: bit0 skip raster, bit1 skip shading, bit2 skip broad list
  int32_t deferred;                  // 1: the reference's deferred path (gbuffer.vert/.frag + brdf.frag), 0: forward
  int32_t gbuffer_view;              // deferred only: -1 the lit scene, 0..3 buffer_visualize.frag on that G-buffer attachment
  // overlay pass only (k_*<..., OVERLAY>): primitives >= ov_first_gizmo_prim are the gizmo -- own viewport (centre
  // ov_cx/ov_cy, half extent ov_half), scissor rectangle [ov_x0, ov_x1) x [ov_y0, ov_y1), and a depth bias that lets
  // them win over everything drawn before (the reference clears the rectangle's depth first, src/main.cpp:150-160)
  uint32_t ov_first_gizmo_prim;
  float ov_half, ov_cx, ov_cy;
  int32_t ov_x0, ov_y0, ov_x1, ov_y1;
};

}  // namespace bbr
