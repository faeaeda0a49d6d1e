// bibim_hip.hip -- C-ABI (include/bibim_hip.h) over the HIP kernels in bb_kernels.hip.h.
//
// One context = one GPU, one HIP stream, all buffers resident in HBM for the context's lifetime.
// A frame is: [H2D of lights + draw descriptors + instances, one async copy from pinned staging] -> k_geometry
// (all draws) -> k_raster (per tile) -> k_shade (per visible pixel).  No host synchronisation inside a frame; capacities (bins, broad list, clip arena) are
// checked lazily at the next synchronising call and the frame is re-rendered once after growing them.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is opened at bbr_comm_init (see the exchange section)

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <memory>
#include <mutex>
#include <unordered_set>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bibim_hip.h"
#include "bb_kernels.hip.h"

using namespace bbr;

namespace {

std::string g_create_error;

// RCCL, opened once per process at the first bbr_comm_unique_id / bbr_comm_init and never closed (it owns threads): the
// single-GPU path loads and runs without it.
struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclCommCount) comm_count = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

struct Mesh {
  Vertex *d_vertices = nullptr;
  uint32_t *d_indices = nullptr;
  uint32_t n_vertices = 0, n_indices = 0;
  bool alive = false;
};

struct Material {
  uint8_t *d_texels[kMapCount] = {};
  uint8_t *d_packed = nullptr;
  MaterialDesc desc = {};
  bool alive = false;
};

struct RecordedDraw {
  int32_t mesh, material;
  uint32_t n_instances, first_instance, first_prim, tris_per_instance;
};

// hipMemset runs on the NULL stream and may return before the fill has happened; the context's streams are created
// hipStreamNonBlocking, i.e. they do NOT order themselves after the NULL stream.  A kernel launched right after an
// allocation could therefore run BEFORE the zero fill and have its counters / fragment counts wiped afterwards
// (seen as an empty first frame when several processes share the GPU).  Every zero fill is completed here.
inline hipError_t zero_fill_sync(void *ptr, size_t bytes) {
  hipError_t e = hipMemset(ptr, 0, bytes);
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  return e;
}

// Host -> device copies that kernels on the (non-blocking) context streams read right afterwards: completed on the
// NULL stream before returning, for the same reason.
inline hipError_t upload_sync(void *dst, const void *src, size_t bytes) {
  hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  return e;
}

template <typename T>
struct DeviceBuffer {
  T *ptr = nullptr;
  size_t cap = 0;  // elements
  hipError_t ensure(size_t n, bool zero = false) {
    if (n <= cap) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&ptr, n * sizeof(T));
    if (e != hipSuccess) return e;
    cap = n;
    if (zero) e = zero_fill_sync(ptr, n * sizeof(T));
    return e;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
};

}  // namespace

// Everything one frame in flight owns.
struct FrameSlot {
  void *h_staging = nullptr;  // pinned: lights, draw descriptors, instances
  static constexpr int kHostFlagWords = 8;
  uint32_t *h_flags = nullptr;  // pinned, device-visible {overflow bits, bin_need, shade items, every-tile entries, clip slots} of the frame last rendered in this
                                // slot (stored by its k_raster): lets a host that never synchronises still grow capacities
  size_t staging_cap = 0;
  DeviceBuffer<uint8_t> d_staging;
  DeviceBuffer<RasterTri> d_tris;
  DeviceBuffer<ShadeRec> d_attrs;
  DeviceBuffer<ClipSlot> d_clip;
  int ctr_index = 0;                        // the slot's counter block (slot index; the overlay slot has the last one)
  DeviceBuffer<BlockStats> d_block_stats;  // one record per k_geometry wave
  DeviceBuffer<uint32_t> d_tile_count;
  DeviceBuffer<uint32_t> d_bins;
  DeviceBuffer<BroadTri> d_broad;
  DeviceBuffer<unsigned long long> d_frags;  // per tile: compacted (pixel << 32 | primitive ref) of covered pixels
  DeviceBuffer<uint32_t> d_frag_count;
  DeviceBuffer<uint32_t> d_items;       // k_shade's work list: [0] = count, then slot << 6 | chunk of 64 fragments
  DeviceBuffer<CookedLight> d_cooked;   // the frame's light table as the light loop consumes it (k_shade_items -> k_shade)
  DeviceBuffer<uint32_t> d_heavy;       // the frame's heavy-tile list (k_geometry -> k_raster; option "heavy_tiles")
  DeviceBuffer<uint32_t> d_item_groups; // 64-fragment chunks per group of 256 launch slots (k_raster -> k_shade_items)
  DeviceBuffer<float4> d_frame;
  DeviceBuffer<float4> d_background;  // deferred path: colour of the pixels no geometry covers
  DeviceBuffer<float> d_depth;        // option "overlays": the frame's resolved depth, for bbr_draw_overlays
  bool has_depth = false;
  bool fused = false;                 // rendered with option "present_fused": d_present holds the frame, d_frame nothing
  FrameUniformBlock frame_u = {};     // the uniforms the frame in this slot was rendered with (overlays need them)
  ViewUniformBlock view_u = {};
  DeviceBuffer<uint32_t> d_present;  // RGBA8 presented image of this slot's frame (bbr_present)
  struct {
    bool active = false;  // bbr_present was queued for the frame in this slot (re-queued if the frame is replayed)
    uint32_t *out = nullptr;
    void *copy_to = nullptr;  // fused presentation: the caller's buffer the image was copied to
    int32_t enable = 0, hdr16 = 1;
    float exposure = 1.f;
  } present;
  hipEvent_t ev_geom_done = nullptr, ev_raster_done = nullptr, ev_shade_done = nullptr, ev_tail_done = nullptr;
  bool in_flight = false;
  int32_t tone_enable = 0;  // FrameUniformBlock.EnableToneMapping / Exposure of the frame in this slot
  float tone_exposure = 1.f;
  float4 *out_used = nullptr;  // where the frame in this slot wrote its pixels
  hipStream_t stream_used = nullptr;  // the stream its k_shade ran on
  uint32_t n_prims = 0;

  void release_tile_buffers() {
    d_tile_count.release(); d_bins.release(); d_frags.release(); d_frag_count.release(); d_items.release(); d_item_groups.release(); d_cooked.release(); d_heavy.release();
  }
  // native exchange (bbr_allgather_frame / bbr_push_shard) with library-owned buffers: every rank's block, the whole frame
  DeviceBuffer<uint8_t> d_gathered, d_whole;
  void release_all() {
    d_staging.release(); d_tris.release(); d_attrs.release(); d_clip.release(); d_gathered.release(); d_whole.release();
    d_block_stats.release();
    release_tile_buffers(); d_broad.release(); d_frame.release(); d_present.release(); d_background.release(); d_depth.release();
    if (h_staging) (void)hipHostFree(h_staging);
    h_staging = nullptr;
    if (h_flags) (void)hipHostFree(h_flags);
    h_flags = nullptr;
    staging_cap = 0;
  }
};

// Live contexts: a scene object (bb::SceneBase, bbs_scene) frees its meshes through the context it was created on; if
// the host destroys the context first (garbage collectors do), those late calls must fail cleanly instead of touching
// freed memory.
namespace {
std::mutex g_live_mutex;
std::unordered_set<const bbr_context *> g_live;
bool is_live(const bbr_context *c) {
  std::lock_guard<std::mutex> lock(g_live_mutex);
  return g_live.count(c) != 0;
}
}  // namespace

struct bbr_context {
  int device = 0;
  int32_t width = 0, height = 0;
  hipStream_t s_geom = nullptr, s_raster = nullptr, s_shade = nullptr, s_present = nullptr;  // context-owned streams
  hipStream_t user_stream = nullptr;                 // bbr_set_stream: everything on the caller's stream, 1 frame in flight
  std::string last_error;

  std::vector<Mesh> meshes;
  std::vector<Material> materials;
  uint8_t *d_default_texels = nullptr;  // 6 x RGBA8 (1x1 default maps)
  DeviceBuffer<MaterialDesc> d_materials;
  bool materials_dirty = true;
  bool all_packed = true;  // every live material has a packed form (k_shade's MIXED = false instantiation may be used)

  FrameUniformBlock frame_u = {};
  ViewUniformBlock view_u = {};

  bool in_frame = false, have_frame = false;
  std::vector<RecordedDraw> draws;
  std::vector<InstanceBlock> host_instances;
  uint32_t n_prims = 0;
  uint32_t n_live_draws = 0;
  FirstPrims first_prims = {{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}};  // of the recorded frame's draws 1 .. 3 (k_geometry)

  static constexpr int kMaxSlots = 4;
  static constexpr int kCounterBlocks = kMaxSlots + 1;  // one per frame slot + one for the overlay pass
  FrameSlot slots[kMaxSlots];
  DeviceBuffer<Counters> d_counters;       // zero between frames: a frame's k_shade clears its slot's block ...
  DeviceBuffer<Counters> d_counters_done;  // ... after copying it here (statistics, overflow check)
  int frames_in_flight = 2;
  uint64_t frame_counter = 0;
  int last_slot = -1;

  DeviceBuffer<SrgbTables> d_srgb_tables;  // thresholds t_k (linear value at which the sRGB byte becomes k) + the keyed table
  DeviceBuffer<uint32_t> d_vis_prim;
  DeviceBuffer<float> d_vis_depth;
  void *ext_out = nullptr;
  uint64_t ext_out_bytes = 0;

  int tile_mode = 1;  // 0: 64x64, 1: 32x32 (default: finer tiles balance better; 16x16 with one-wave workgroups measured slower)
  uint32_t bin_cap = 512, broad_cap = 4096, clip_cap = 4096, broad_threshold = 16;
  int32_t rank = 0, world = 1, band_rows = 0;
  bool dump_vis = false;
  bool present_fused = false;  // option "present_fused": frames are written as presented RGBA8, no fp32 frame
  bool overlays = false;  // option "overlays": frames keep their depth so that bbr_draw_overlays can test against it
  Mesh marker_mesh, gizmo_mesh;  // generateUVSphereMesh(0.1, 16, 16) and the caller's gizmo, both as Vertex meshes
  FrameSlot ov;           // buffers of the overlay pass (its own little frame)
  int gbuffer_view = -1;  // option "gbuffer_view": GBufferVisualizingOption (src/scene.h:27-35) while the deferred path runs
  bool deferred = false;  // option "render_pass": the reference's deferred path (its default) instead of the forward one
  bool dump_gbuffer = false;
  DeviceBuffer<uint2> d_gbuffer;  // width*height*4 (four RGBA16F texels per pixel), only while bbr_read_gbuffer runs
  uint32_t ablate = 0;
  int n_cus = 256;         // compute units of the device (hipDeviceProp_t::multiProcessorCount)
  int timing = 0;  // 0 off, 1 five events per frame, 2 only the two events around k_shade
  int timing_stride = 1;  // option "timing_stride": events on every n-th frame only (two events a frame cost ~4 % at C3)
  uint64_t timing_tick = 0;
  bool timing_this = false;  // the frame being submitted carries events
  // timing ring: (frame start, geometry done, raster done, shade start, shade done) per frame since the last reset
  std::vector<hipEvent_t> ring;
  uint32_t ring_frames = 0;
  std::vector<hipEvent_t> present_ring;  // (start, stop) around each k_present while timing is on
  uint32_t present_launches = 0;
  static constexpr uint32_t kRingCap = 512;
  static constexpr uint32_t kRingEvents = 5;
  int retries = 0;
  // host side of the frame loop since the last bbr_host_timing_reset (steady_clock; no GPU call is made to keep them):
  // frames submitted, time inside the submit (bbr_end_frame / bbr_replay_frame), and the part of it spent blocked because
  // the frame slot's previous frame had not left the GPU yet
  uint64_t host_frames = 0, host_submit_ns = 0, host_blocked_ns = 0, host_blocked_frames = 0;

  hipStream_t geom_stream() const { return user_stream ? user_stream : s_geom; }
  // k_raster on a stream of its own: geometry of frame N+1 (other slot, other counter block) need not wait for the
  // raster of frame N.  At 1080p that chain -- not the GPU -- set the frame rate (C2: 72 -> 41 us per frame)
  // Stream layout of the frames in flight (option "stream_layout").  All three render the same bits:
  //   0  geometry + raster on s_geom, shade on s_shade, present on s_present          (stage streams)
  //   1  as 0, but k_raster on s_raster: geometry of frame N+1 need not wait for the raster of frame N
  //   2  every kernel of a frame (copy, geometry, raster, shade, present) on the stream of its slot: frames share
  //      nothing (each slot has its own buffers and counter block), so whole frames overlap and no event is needed
  //      inside a frame.  The four context streams double as the slot streams: a process that owns more than a
  //      handful of HIP streams gets slower as a whole -- with seven streams every layout lost 60 %
  // A plain option with a fixed default, 2: measured with three frames in flight (us per frame, layouts 0 / 1 / 2) it is
  // the fastest everywhere -- C2 1080p 84.5 / 67.1 / 37.3, C3 4K 191.5 / 153.6 / 148.6 -- because a frame's chain of
  // dependent kernels (copy -> geometry -> raster -> items -> shade) then only waits for itself.  (Round 1 timed the
  // layouts on the first ~500 frames of every workload and switched by itself; the measurement was fragile and made
  // the frame rate a function of history.)
  ncclComm_t comm = nullptr;
  int comm_rank = -1, comm_world = 0;
  int exchange_slot = -1, exchange_form = -1;  // where the last exchange left the whole frame (library-owned buffers)
  void *exchange_whole = nullptr;
  int push_mode = 1;  // option "push_mode": 1 one kernel storing to every peer (all links at once), 0 copies one after the other
  std::unordered_set<int> peer_mapped;  // devices whose memory this context's device can store to (peer access enabled)
  bool last_push_direct = false;
  int64_t no_tail_items = 40000;  // option "no_tail_items": frames with at most this many item slots get no tail launch
  // option "heavy_tiles": k_raster starts the tiles one of whose bins holds at least this many references before the
  // screen-ordered rest (long frames only; 0: plain screen order; -1, the default: 64 while ONE frame is in flight, 0
  // otherwise).  Measured at C3: k_raster alone 61.9 -> 51.0 us and a frame's latency 177 -> 167 us with 64 (48: the same;
  // 96 / 128 / 256: 58.7 / 61.0 / 61.4; 1 / 16 / 32: 55 / 54 / 53.5 and k_geometry +9 / +10 / +6 us for its appends) -- and
  // with three frames in flight the frame period 1-2 % LONGER (32: 7 %): the other frames' kernels fill the tail that
  // the order removes, and the light tiles' background stores then come in one burst instead of spread over the launch.
  int64_t heavy_tiles = -1;
  static constexpr int kLayouts = 3;
  int layout_mode = 2;  // the option
  int layout = 2;       // layout of the frame being submitted
  bool pipelined() const { return !user_stream && frames_in_flight > 1; }
  hipStream_t slot_stream(int i) const { return i == 0 ? s_raster : (i == 1 ? s_shade : (i == 2 ? s_present : s_geom)); }
  hipStream_t frame_geom_stream(int slot) const { return (pipelined() && layout == 2) ? slot_stream(slot) : geom_stream(); }
  std::vector<hipEvent_t> pending_waits;  // bbr_wait_event: applied to the first stream of the next frame
  hipStream_t raster_stream(int slot) const {
    return !pipelined() ? geom_stream() : (layout == 2 ? slot_stream(slot) : (layout == 1 ? s_raster : s_geom));
  }
  hipStream_t frame_shade_stream(int slot) const { return (pipelined() && layout == 2) ? slot_stream(slot) : shade_stream(); }
  hipStream_t shade_stream() const { return user_stream ? user_stream : (frames_in_flight > 1 ? s_shade : s_geom); }
  // k_present is bandwidth-bound, k_shade issue-bound: on its own stream the presentation of frame N overlaps the
  // shading of frame N+1 instead of delaying it
  hipStream_t present_stream() const { return (user_stream || frames_in_flight == 1) ? shade_stream() : s_present; }
  int n_slots() const { return user_stream ? 1 : frames_in_flight; }
  int tile_w() const { return tile_mode == 0 ? 64 : 32; }
  int tile_h() const { return tile_mode == 0 ? 64 : 32; }
  int tiles_x() const { return (width + tile_w() - 1) / tile_w(); }
  int tiles_y() const { return (height + tile_h() - 1) / tile_h(); }
  int eff_band_rows() const { return band_rows > 0 ? band_rows : tile_h(); }
  int n_bands() const { return (height + eff_band_rows() - 1) / eff_band_rows(); }
  int local_bands() const { return world > 1 ? (n_bands() - rank + world - 1) / world : n_bands(); }
  int shard_rows() const { return world > 1 ? ((n_bands() + world - 1) / world) * eff_band_rows() : height; }
};

namespace {

int fail(bbr_context *ctx, int code, const std::string &msg) {
  if (ctx) ctx->last_error = msg;
  else g_create_error = msg;
  return code;
}

// Every entry point that can allocate, copy or launch makes the context's device current first: a process that drives
// several GPUs (one context each) may call with another device current.
#define BBR_ON_DEVICE(ctx)                                                                                     \
  do {                                                                                                         \
    hipError_t _e = hipSetDevice((ctx)->device);                                                               \
    if (_e != hipSuccess) return fail(ctx, BBR_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(_e)); \
  } while (0)

#define HIP_TRY(ctx, expr)                                                                                     \
  do {                                                                                                         \
    hipError_t _e = (expr);                                                                                    \
    if (_e != hipSuccess)                                                                                      \
      return fail(ctx, _e == hipErrorOutOfMemory ? BBR_ERR_OUT_OF_MEMORY : BBR_ERR_HIP,                        \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                                          \
  } while (0)

// P*V exactly as the vertex stage's contract evaluates it (column j = P * V[j], fmaf chain)
Mat4 proj_view(const ViewUniformBlock &v) {
  Mat4 r;
  for (int c = 0; c < 4; ++c) {
    const float *col = v.view.M[c];
    for (int i = 0; i < 4; ++i)
      r.M[c][i] = std::fmaf(v.proj.M[3][i], col[3],
                            std::fmaf(v.proj.M[2][i], col[2], std::fmaf(v.proj.M[1][i], col[1], v.proj.M[0][i] * col[0])));
  }
  return r;
}

FrameParams make_params(const bbr_context *c) {
  FrameParams fp = {};  // (the overlay fields stay 0 for the main pass)
  fp.width = c->width;
  fp.height = c->height;
  fp.half_w = 0.5f * (float)c->width;
  fp.half_h = 0.5f * (float)c->height;
  fp.tiles_x = c->tiles_x();
  fp.tiles_y = c->tiles_y();
  fp.bin_cap = c->bin_cap;
  fp.broad_cap = c->broad_cap;
  fp.clip_cap = c->clip_cap;
  fp.broad_threshold = c->broad_threshold;
  fp.rank = c->rank;
  fp.world = c->world;
  fp.band_tiles = c->eff_band_rows() / c->tile_h();
  fp.shard_rows = c->shard_rows();
  fp.ablate = c->ablate;
  fp.deferred = c->deferred ? 1 : 0;
  fp.gbuffer_view = c->deferred ? c->gbuffer_view : -1;
  return fp;
}

// Wait for everything this context has queued.
int drain(bbr_context *c) {
  HIP_TRY(c, hipStreamSynchronize(c->geom_stream()));
  if (c->shade_stream() != c->geom_stream()) HIP_TRY(c, hipStreamSynchronize(c->shade_stream()));
  if (c->s_raster) HIP_TRY(c, hipStreamSynchronize(c->s_raster));

  if (c->s_present) HIP_TRY(c, hipStreamSynchronize(c->s_present));
  for (FrameSlot &s : c->slots) s.in_flight = false;
  return BBR_OK;
}

int ensure_srgb_tables(bbr_context *c);

int ensure_slot_buffers(bbr_context *c, FrameSlot &s) {
  size_t tiles = (size_t)c->tiles_x() * c->tiles_y();
  size_t out_rows = (size_t)std::max(c->height, c->shard_rows());
  HIP_TRY(c, s.d_tris.ensure(std::max<size_t>(c->n_prims, 1)));
  HIP_TRY(c, s.d_attrs.ensure(std::max<size_t>(c->n_prims, 1)));
#ifdef BB_STAMPS
  HIP_TRY(c, s.d_clip.ensure(c->clip_cap + 8192));  // diagnostic build: room for per-workgroup time stamps
#else
  HIP_TRY(c, s.d_clip.ensure(c->clip_cap));
#endif
  HIP_TRY(c, c->d_counters.ensure(bbr_context::kCounterBlocks, true));
  HIP_TRY(c, c->d_counters_done.ensure(bbr_context::kCounterBlocks, true));
  HIP_TRY(c, s.d_block_stats.ensure(std::max<size_t>((c->n_prims + 255) / 256, 1) * 4, true));
  HIP_TRY(c, s.d_tile_count.ensure(tiles * kBinClasses, true));
  HIP_TRY(c, s.d_bins.ensure(tiles * kBinClasses * c->bin_cap));
  // (at least kBroadSpec entries: k_raster's light-tile path reads that many before it knows how many the frame wrote)
  HIP_TRY(c, s.d_broad.ensure(std::max<size_t>(c->broad_cap, kBroadSpec), true));
  HIP_TRY(c, s.d_frags.ensure(tiles * (size_t)(c->tile_w() * c->tile_h())));
#ifdef BB_STAMPS
  HIP_TRY(c, s.d_frag_count.ensure(tiles * 17, true));  // diagnostic build: 8 x u64 raster stamps per tile behind the counts
#else
  HIP_TRY(c, s.d_frag_count.ensure(tiles, true));
#endif
  // (+ kShadeWaves: the last workgroup of k_shade's main launch reads the item words of all its waves before it knows the count)
  HIP_TRY(c, s.d_items.ensure(1 + tiles * (size_t)(c->tile_w() * c->tile_h() / 64) + kShadeWaves, true));
  if (tiles > (size_t)kItemGroupSlots * kItemGroups)  // 65536 launch slots: 8192 x 8192 pixels at 32 x 32 tiles
    return fail(c, BBR_ERR_INVALID_ARGUMENT, "frame too large for this tile size: set option tile_mode to 0 (64 x 64 tiles)");
  HIP_TRY(c, s.d_item_groups.ensure((size_t)kItemGroups * kItemGroupStride, true));
  HIP_TRY(c, s.d_cooked.ensure(kMaxNumLights));
  HIP_TRY(c, s.d_heavy.ensure(tiles * kBinClasses, true));
  if (c->deferred) HIP_TRY(c, s.d_background.ensure(2));
  if (c->overlays && &s != &c->ov) HIP_TRY(c, s.d_depth.ensure((size_t)c->width * c->height));
  if (c->present_fused && &s != &c->ov) {
    HIP_TRY(c, s.d_present.ensure((size_t)c->width * out_rows));
    int rc_t = ensure_srgb_tables(c);
    if (rc_t) return rc_t;
  }
  if (c->dump_gbuffer) HIP_TRY(c, c->d_gbuffer.ensure((size_t)c->width * c->height * 4, true));
  if (!c->ext_out) HIP_TRY(c, s.d_frame.ensure(out_rows * c->width));
  if (c->dump_vis) {
    HIP_TRY(c, c->d_vis_prim.ensure((size_t)c->width * c->height));
    HIP_TRY(c, c->d_vis_depth.ensure((size_t)c->width * c->height));
  }
  if (!s.h_flags) {
    HIP_TRY(c, hipHostMalloc((void **)&s.h_flags, FrameSlot::kHostFlagWords * sizeof(uint32_t), hipHostMallocDefault));
    for (int i = 0; i < FrameSlot::kHostFlagWords; ++i) s.h_flags[i] = 0u;
  }
  if (!s.ev_geom_done) HIP_TRY(c, hipEventCreateWithFlags(&s.ev_geom_done, hipEventDisableTiming));
  if (!s.ev_raster_done) HIP_TRY(c, hipEventCreateWithFlags(&s.ev_raster_done, hipEventDisableTiming));
  if (!s.ev_shade_done) HIP_TRY(c, hipEventCreateWithFlags(&s.ev_shade_done, hipEventDisableTiming));
  if (!s.ev_tail_done) HIP_TRY(c, hipEventCreateWithFlags(&s.ev_tail_done, hipEventDisableTiming));
  return BBR_OK;
}

int upload_material_table(bbr_context *c) {
  if (!c->materials_dirty) return BBR_OK;
  size_t n = std::max<size_t>(c->materials.size(), 1);
  std::vector<MaterialDesc> h(n);
  c->all_packed = true;
  for (size_t i = 0; i < c->materials.size(); ++i) {
    h[i] = c->materials[i].desc;
    if (c->materials[i].alive && !c->materials[i].desc.packed) c->all_packed = false;
  }
  int rc = drain(c);
  if (rc) return rc;
  HIP_TRY(c, c->d_materials.ensure(n));
  HIP_TRY(c, upload_sync(c->d_materials.ptr, h.data(), n * sizeof(MaterialDesc)));
  c->materials_dirty = false;
  return BBR_OK;
}

template <int TW, int TH>
void launch_frame(bbr_context *c, FrameSlot &s, const FrameSlot *prev, const FrameParams &fp_in, const Mat4 &pv,
                  const Mat4 &view, const ShadeParams &sp, const Light *d_lights, const DrawDesc *d_draws, uint32_t n_draws, float4 *out,
                  uint32_t *d_item_head) {
  const int slot_index = (int)(&s - c->slots);
  FrameParams fp = fp_in;
  // a rank without bands (more ranks than bands) still launches one row: its blocks fall off tiles_y and exit
  const int grid_y = std::max(1, c->world > 1 ? c->local_bands() * fp.band_tiles : fp.tiles_y);
  // A SHORT frame (at most option "no_tail_items" item slots: 1080p has 32 640) has no k_shade_items launch: k_raster's tiles
  // append their items themselves through the head word the staging copy has just zeroed, and its first workgroup cooks
  // the lights.  One kernel and one kernel boundary less on the frame's chain of dependent kernels; such a frame is also
  // shaded at full coverage, without the tail launch (below).
  const uint32_t max_items = (uint32_t)(fp.tiles_x * grid_y) * (uint32_t)(TW * TH / 64);
  const bool short_frame = max_items <= (uint32_t)c->no_tail_items;
  // Heavy tiles first (long frames; k_raster): the heavy rows in front of the launch are sized from the length of the list
  // the slot's previous frame produced (pinned word 5, written by k_raster) + 1/8 + 8.  A list that does not fit them is not
  // used by that frame at all (plain screen order), so the first frame of a slot, or one after a jump, is merely slower.
  const int64_t heavy_tiles = c->heavy_tiles >= 0 ? c->heavy_tiles : (c->pipelined() ? 0 : 64);
  if (!short_frame && heavy_tiles > 0 && !c->dump_vis) {
    fp.heavy_threshold = (uint32_t)heavy_tiles;
    const uint32_t seen_heavy = s.h_flags ? s.h_flags[5] : 0u;
    if (seen_heavy) fp.heavy_rows = (int32_t)((seen_heavy + seen_heavy / 8u + 8u + (uint32_t)fp.tiles_x - 1u) / (uint32_t)fp.tiles_x);
  }
  hipStream_t sg = c->frame_geom_stream(slot_index), sr = c->raster_stream(slot_index), ss = c->frame_shade_stream(slot_index);
  Counters *ctr = c->d_counters.ptr + s.ctr_index, *ctr_done = c->d_counters_done.ptr + s.ctr_index;
  hipEvent_t *ev = c->timing_this ? &c->ring[bbr_context::kRingEvents * (c->ring_frames % bbr_context::kRingCap)] : nullptr;
#ifdef BB_ABLATE
  // diagnostic builds only (tools/_gpu_overlap.py): bit 17 = launch nothing but k_shade -- the lists of the slot's last whole
  // frame are shaded again -- which measures the shading's own pipelined rate.  (There is no "front half only" twin: k_shade is
  // what clears the slot's counters and chunk totals, and a frame without it overruns the item list.)
  const bool skip_front = (c->ablate & (1u << 17)) != 0u;
#else
  constexpr bool skip_front = false;
#endif
  if (c->n_prims && !skip_front)
    hipLaunchKernelGGL((k_geometry<TW, TH>), dim3((c->n_prims + 255) / 256), dim3(256), 0, sg, d_draws, n_draws,
                       c->n_prims, c->first_prims, s.d_tris.ptr, s.d_attrs.ptr, s.d_tile_count.ptr, s.d_bins.ptr, ctr, pv, view, fp, s.d_clip.ptr,
                       s.d_broad.ptr, c->d_materials.ptr, s.d_block_stats.ptr, s.d_heavy.ptr);
  if (ev && c->timing == 1) (void)hipEventRecord(ev[1], sg);
  if (sr != sg) {
    (void)hipEventRecord(s.ev_geom_done, sg);
    (void)hipStreamWaitEvent(sr, s.ev_geom_done, 0);
  }
  // k_raster writes the background pixels of `out`: if the frame still shading on the other stream writes the
  // same buffer (single external output), raster has to wait for it; geometry above still overlapped
  // (every frame still in flight, not just the previous one: with one stream per slot, or while the layout is being
  //  switched, "after the previous frame" no longer implies "after the one before")
  (void)prev;
  for (const FrameSlot &o : c->slots)
    if (&o != &s && o.in_flight && o.out_used == out && o.stream_used != sr) (void)hipStreamWaitEvent(sr, o.ev_shade_done, 0);
  // option "present_fused": k_raster / k_shade write presented pixels into the slot's RGBA8 image
  uint32_t *out8 = c->present_fused ? s.d_present.ptr : nullptr;
  const SrgbTables *tables = c->present_fused ? c->d_srgb_tables.ptr : nullptr;
  if (fp.deferred)
    hipLaunchKernelGGL(k_deferred_background, dim3(1), dim3(kBackgroundThreads), 0, sr, sp, d_lights, s.d_background.ptr, tables, fp.gbuffer_view);
  uint32_t *item_head = short_frame ? d_item_head : nullptr;
  if (!skip_front)
  hipLaunchKernelGGL((k_raster<TW, TH>), dim3(fp.tiles_x, fp.heavy_rows + grid_y), dim3(kTileThreads), 0, sr, s.d_tile_count.ptr, ctr,
                     s.d_broad.ptr, s.d_frag_count.ptr, s.d_frags.ptr, out, fp, s.d_tris.ptr, s.d_clip.ptr, s.d_bins.ptr,
                     c->dump_vis ? c->d_vis_prim.ptr : nullptr,
                     c->dump_vis ? c->d_vis_depth.ptr : nullptr,
                     fp.deferred ? s.d_background.ptr : nullptr, (c->overlays && c->world == 1) ? s.d_depth.ptr : nullptr,
                     s.h_flags, out8, short_frame ? nullptr : s.d_item_groups.ptr, item_head, s.d_items.ptr, d_lights, sp.num_lights,
                     s.d_cooked.ptr, s.d_heavy.ptr);
  s.has_depth = c->overlays && c->world == 1;
  // k_shade's work list (64 fragments per item), built from the per-tile fragment counts as soon as k_raster is done; the
  // same launch cooks the frame's light table
  if (!short_frame && !skip_front)
    hipLaunchKernelGGL((k_shade_items<TW, TH>), dim3((unsigned)((fp.tiles_x * grid_y + kItemsThreads - 1) / kItemsThreads)), dim3(kItemsThreads), 0, sr, fp, s.d_frag_count.ptr, s.d_item_groups.ptr, s.d_items.ptr,
                       fp.tiles_x, grid_y, s.h_flags ? s.h_flags + 2 : nullptr, d_lights, sp.num_lights, s.d_cooked.ptr,
                       s.d_tile_count.ptr, ctr, s.d_heavy.ptr);
  const uint32_t *item_count = short_frame ? item_head : s.d_items.ptr;   // where k_shade finds the number of items
  if (ev && c->timing == 1) (void)hipEventRecord(ev[2], sr);
  if (ss != sr) {
    (void)hipEventRecord(s.ev_raster_done, sr);
    (void)hipStreamWaitEvent(ss, s.ev_raster_done, 0);
  }
  // (ev[3], the start of the shading interval, is recorded in front of the MAIN launch below -- behind the tail launch when
  //  both are on one stream: the interval is then the dominant kernel's own, which the kernel trace can be held against)
  uint2 *gbuf = (fp.deferred && c->dump_gbuffer) ? c->d_gbuffer.ptr : nullptr;
  // Main launch: one item (64 fragments) per wave, four per workgroup, sized from the item count of the frame this slot
  // rendered last (k_shade_items leaves it in pinned host memory) plus 3 %; tail launch: a small persistent grid for
  // whatever lies behind that (a scene that suddenly grew; normally nothing, and its workgroups exit at once).
  const uint32_t seen = s.h_flags ? s.h_flags[2] : 0u;
  uint32_t est = seen ? seen + seen / 32u + 64u : max_items;
  if (est > max_items) est = max_items;
  // A small frame is launched at full coverage instead: workgroups without an item leave after one scalar load, and a few
  // thousand of them cost less than the tail launch's kernel boundary on the frame's chain of dependent kernels (at
  // 1080p the chain's length over the frames in flight IS the frame rate).
  if (short_frame) est = max_items;
  const uint32_t main_wgs = std::max(1u, (est + kShadeWaves - 1) / kShadeWaves);
  // The tail runs on the raster stream, beside the main launch (their items are disjoint): in front of or behind it on
  // one stream an empty tail would still cost the frame a kernel boundary (~4 us).
  const bool tail = main_wgs * kShadeWaves < max_items;
  auto shade = [&](auto deferred, auto present) {
    if (tail) {
      hipLaunchKernelGGL((k_shade<TW, TH, decltype(deferred)::value, decltype(present)::value, true, true>), dim3(32), dim3(kShadeThreads),
                         0, sr, s.d_items.ptr, item_count, main_wgs * (uint32_t)kShadeWaves, s.d_frags.ptr, s.d_frag_count.ptr, s.d_attrs.ptr,
                         s.d_clip.ptr, fp, sp, s.d_cooked.ptr, c->d_materials.ptr, out, gbuf, tables, out8, ctr, ctr_done, nullptr);
      if (ss != sr) (void)hipEventRecord(s.ev_tail_done, sr);
    }
    if (ev) (void)hipEventRecord(ev[3], ss);
    // (the main launch without the per-map sampling path when every material is packed: k_shade, MIXED)
    auto main_launch = [&](auto mixed) {
      hipLaunchKernelGGL((k_shade<TW, TH, decltype(deferred)::value, decltype(present)::value, false, decltype(mixed)::value>), dim3(main_wgs),
                         dim3(kShadeThreads), 0, ss, s.d_items.ptr, item_count, 0u, s.d_frags.ptr, s.d_frag_count.ptr, s.d_attrs.ptr,
                         s.d_clip.ptr, fp, sp, s.d_cooked.ptr, c->d_materials.ptr, out, gbuf, tables, out8, ctr, ctr_done,
                         short_frame ? nullptr : s.d_item_groups.ptr);
    };
    if (c->all_packed) main_launch(std::false_type{});
    else main_launch(std::true_type{});
    if (tail && ss != sr) (void)hipStreamWaitEvent(ss, s.ev_tail_done, 0);
  };
  if (fp.deferred) {
    if (out8) shade(std::true_type{}, std::true_type{});
    else shade(std::true_type{}, std::false_type{});
  } else {
    if (out8) shade(std::false_type{}, std::true_type{});
    else shade(std::false_type{}, std::false_type{});
  }
  (void)hipEventRecord(s.ev_shade_done, ss);
  s.stream_used = ss;
  if (ev) {
    (void)hipEventRecord(ev[4], ss);
    ++c->ring_frames;
  }
}

// A capacity overflowed (bit0 bins, bit1 every-tile list, bit2 clip arena): grow it.  Everything must have left the GPU.
int apply_growth(bbr_context *c, uint32_t overflow, uint32_t bin_need, uint32_t broad_need, uint32_t clip_need) {
  if (getenv("BBR_DEBUG"))
    fprintf(stderr, "[bbr] capacity growth: overflow bits %u, bin_need %u (bin_cap %u, broad_cap %u, clip_cap %u), frame %llu\n", overflow,
            bin_need, c->bin_cap, c->broad_cap, c->clip_cap, (unsigned long long)c->frame_counter);
  if (overflow & 1u) {
    // bin_need is exact for the frame that overflowed; round up with headroom so small scene changes do not re-trigger
    uint32_t need = std::max(bin_need + bin_need / 4u, c->bin_cap * 2u);
    c->bin_cap = (need + 255u) & ~255u;
    for (FrameSlot &s : c->slots) s.d_bins.release();
  }
  // the every-tile list and the clip arena: what the overflowed frame asked for (a lower bound: primitives dropped by a
  // full clip arena asked for no list entry) with headroom, at least double
  auto grown = [](uint32_t cap, uint32_t need) { return std::max(cap * 2u, (need + need / 4u + 255u) & ~255u); };
  if (overflow & 2u) c->broad_cap = grown(c->broad_cap, broad_need);
  if (overflow & 4u) c->clip_cap = grown(c->clip_cap, clip_need);
  ++c->retries;
  for (FrameSlot &s : c->slots) {
    if (s.h_flags) s.h_flags[0] = s.h_flags[1] = s.h_flags[2] = s.h_flags[3] = s.h_flags[4] = 0u;  // they describe frames rendered with the old capacities
    // tile counters may hold residue of references that did not fit
    if (s.d_tile_count.ptr) HIP_TRY(c, zero_fill_sync(s.d_tile_count.ptr, s.d_tile_count.cap * sizeof(uint32_t)));
  }
  return BBR_OK;
}

// Queue the recorded frame into slot `slot_index`.  Asynchronous.
int submit_frame_into(bbr_context *c, int slot_index) {
  if (!c->have_frame) return fail(c, BBR_ERR_NOT_IN_FRAME, "no recorded frame");
  int rc = upload_material_table(c);
  if (rc) return rc;
  FrameSlot &s = c->slots[slot_index];
  // the slot's previous frame (two frames ago) must have left the GPU before its buffers are reused
  if (s.in_flight) {
    if (hipEventQuery(s.ev_shade_done) != hipSuccess) {  // (hipErrorNotReady: still on the GPU -- the host waits, and says so)
      const auto b0 = std::chrono::steady_clock::now();
      HIP_TRY(c, hipEventSynchronize(s.ev_shade_done));
      c->host_blocked_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - b0).count();
      ++c->host_blocked_frames;
    }
    s.in_flight = false;
  }
  // Self-healing for hosts that only stream frames: the frame that last used this slot reported an overflow.  Its
  // pixels (and those of the frames queued since) are incomplete and already handed out; from this frame on the
  // capacities fit.  (Synchronising calls do better: they re-render the overflowed frame, sync_and_fix.)
  if (s.h_flags && s.h_flags[0]) {
    const uint32_t overflow = s.h_flags[0], bin_need = s.h_flags[1], broad_need = s.h_flags[3], clip_need = s.h_flags[4];
    rc = drain(c);
    if (rc) return rc;
    rc = apply_growth(c, overflow, bin_need, broad_need, clip_need);
    if (rc) return rc;
  }
  rc = ensure_slot_buffers(c, s);
  if (rc) return rc;

  // staging layout: [lights (100 * 64 B)] [draw descriptors] [instances]
  size_t lights_bytes = sizeof(Light) * kMaxNumLights;
  size_t draws_bytes = (sizeof(DrawDesc) * c->draws.size() + 127) & ~(size_t)127;
  size_t inst_bytes = sizeof(InstanceBlock) * c->host_instances.size();
  // ... [item head: one zero word, 16-byte slot] -- k_raster's tiles count a short frame's items through it (launch_frame)
  const size_t head_offset = (lights_bytes + draws_bytes + inst_bytes + 15) & ~(size_t)15;
  size_t total = head_offset + 16;
  if (total > s.staging_cap) {
    if (s.h_staging) (void)hipHostFree(s.h_staging);
    s.h_staging = nullptr;
    size_t cap = std::max<size_t>(total * 2, 1 << 16);
    HIP_TRY(c, hipHostMalloc(&s.h_staging, cap, hipHostMallocDefault));
    s.staging_cap = cap;
    HIP_TRY(c, s.d_staging.ensure(cap));
  }
  std::memcpy(s.h_staging, c->frame_u.lights, lights_bytes);
  {
    // draw descriptors point into the device copy of the instance data made by the same transfer
    const InstanceBlock *d_inst_base = reinterpret_cast<const InstanceBlock *>(s.d_staging.ptr + lights_bytes + draws_bytes);
    DrawDesc *hd = reinterpret_cast<DrawDesc *>((uint8_t *)s.h_staging + lights_bytes);
    uint32_t k = 0;
    for (const RecordedDraw &rd : c->draws) {
      if (!rd.n_instances || !rd.tris_per_instance) continue;  // empty draws would break the first_prim search
      const Mesh &m = c->meshes[rd.mesh];
      DrawDesc d;
      d.vertices = m.d_vertices;
      d.indices = m.d_indices;
      d.instances = d_inst_base + rd.first_instance;
      d.n_instances = rd.n_instances;
      d.tris_per_instance = rd.tris_per_instance;
      d.first_prim = rd.first_prim;
      d.material = (uint32_t)rd.material;
      const MaterialDesc &md = c->materials[rd.material].desc;
      d.packed = md.packed;
      d.packed_dims = md.packed ? ((uint32_t)md.pw | ((uint32_t)md.ph << 16)) : 0u;
      d.pad = 0;
      if (k >= 1 && k <= (uint32_t)kInlineFirstPrims) c->first_prims.v[k - 1] = d.first_prim;
      hd[k++] = d;
    }
    for (uint32_t q = k; q <= (uint32_t)kInlineFirstPrims; ++q)
      if (q >= 1) c->first_prims.v[q - 1] = 0xFFFFFFFFu;
    c->n_live_draws = k;
  }
  if (inst_bytes) std::memcpy((uint8_t *)s.h_staging + lights_bytes + draws_bytes, c->host_instances.data(), inst_bytes);
  std::memset((uint8_t *)s.h_staging + head_offset, 0, 16);
  uint32_t *d_item_head = reinterpret_cast<uint32_t *>(s.d_staging.ptr + head_offset);

  const int n_lights = std::min(std::max(c->frame_u.num_lights, 0), kMaxNumLights);
  c->layout = c->pipelined() ? c->layout_mode : 0;
  hipStream_t sg = c->frame_geom_stream(slot_index);
  for (hipEvent_t e : c->pending_waits) HIP_TRY(c, hipStreamWaitEvent(sg, e, 0));  // bbr_wait_event
  c->pending_waits.clear();
  c->timing_this = c->timing && (c->timing_tick++ % (uint64_t)c->timing_stride) == 0;
  if (c->timing_this) {
    if (c->ring.empty()) {
      c->ring.resize(bbr_context::kRingEvents * bbr_context::kRingCap);
      for (auto &e : c->ring) HIP_TRY(c, hipEventCreate(&e));
    }
    if (c->timing == 1) HIP_TRY(c, hipEventRecord(c->ring[bbr_context::kRingEvents * (c->ring_frames % bbr_context::kRingCap)], sg));
  }
  s.ctr_index = slot_index;
  const Light *d_lights = reinterpret_cast<const Light *>(s.d_staging.ptr);
  const DrawDesc *d_draws = reinterpret_cast<const DrawDesc *>(s.d_staging.ptr + lights_bytes);
  FrameParams fp = make_params(c);
  // forward: gl_Position = (P*V) * posWorld; deferred: P * (V * posWorld) -- the kernel gets P and V separately
  Mat4 pv = c->deferred ? c->view_u.proj : proj_view(c->view_u);
  const Mat4 view = c->view_u.view;
  ShadeParams sp = {};
  std::memcpy(sp.view_pos, c->view_u.view_pos, sizeof sp.view_pos);
  sp.enable_normal_map = c->view_u.enable_normal_map;
  sp.tone_enable = c->frame_u.enable_tone_mapping;
  sp.exposure = c->frame_u.exposure;
  sp.num_lights = n_lights;
  float4 *out = c->ext_out ? reinterpret_cast<float4 *>(c->ext_out) : s.d_frame.ptr;
  const FrameSlot *prev = (c->last_slot >= 0 && c->last_slot != slot_index) ? &c->slots[c->last_slot] : nullptr;

  HIP_TRY(c, hipMemcpyAsync(s.d_staging.ptr, s.h_staging, total, hipMemcpyHostToDevice, sg));
  if (c->tile_mode == 0) launch_frame<64, 64>(c, s, prev, fp, pv, view, sp, d_lights, d_draws, c->n_live_draws, out, d_item_head);
  else launch_frame<32, 32>(c, s, prev, fp, pv, view, sp, d_lights, d_draws, c->n_live_draws, out, d_item_head);
  HIP_TRY(c, hipGetLastError());
  s.in_flight = true;
  s.fused = c->present_fused;
  s.present.active = c->present_fused;  // fused presentation: the frame IS the presented image
  s.present.out = c->present_fused ? s.d_present.ptr : nullptr;
  s.present.enable = c->frame_u.enable_tone_mapping;
  s.present.exposure = c->frame_u.exposure;
  s.present.hdr16 = 1;
  s.present.copy_to = nullptr;
  s.frame_u = c->frame_u;
  s.view_u = c->view_u;
  s.tone_enable = c->frame_u.enable_tone_mapping;
  s.tone_exposure = c->frame_u.exposure;
  s.out_used = out;
  s.n_prims = c->n_prims;
  c->last_slot = slot_index;
  return BBR_OK;
}

int ensure_srgb_tables(bbr_context *c) {
  if (!c->d_srgb_tables.ptr) {
    // t_k = float(decode((k - 0.5) / 255)) (the same table the CPU oracle builds), then the table keyed by the
    // float's top bits: thresholds <= lower edge of each cell; a cell may contain at most one threshold
    auto hp = std::make_unique<SrgbTables>();  // 4.3 KB: not on the stack, not shared between contexts
    SrgbTables &h = *hp;
    for (int k = 1; k <= 255; ++k) {
      const double b = ((double)k - 0.5) / 255.0;
      h.thr[k - 1] = (float)(b <= 0.04045 ? b / 12.92 : std::pow((b + 0.055) / 1.055, 2.4));
    }
    h.thr[255] = std::numeric_limits<float>::infinity();
    for (uint32_t cell = 0; cell < kSrgbLutCells; ++cell) {
      auto edge = [](uint32_t cl) {
        uint32_t bits = ((kSrgbLutFirstExp << 8) + cl) << 15;
        float f;
        std::memcpy(&f, &bits, 4);
        return f;
      };
      const float lo = edge(cell), hi = edge(cell + 1);
      uint32_t below = 0, inside = 0;
      for (int k = 0; k < 255; ++k) {
        below += h.thr[k] <= lo;
        inside += h.thr[k] > lo && h.thr[k] < hi;
      }
      if (inside > 1) return fail(c, BBR_ERR_HIP, "sRGB table: two thresholds in one cell");
      h.lut[cell] = (uint8_t)below;
    }
    HIP_TRY(c, c->d_srgb_tables.ensure(1));
    HIP_TRY(c, upload_sync(c->d_srgb_tables.ptr, &h, sizeof h));
  }
  return BBR_OK;
}

// k_present for the frame in slot `s`, on the shade stream (ordered after the frame's k_shade)
int queue_present(bbr_context *c, FrameSlot &s) {
  int rc_tables = ensure_srgb_tables(c);
  if (rc_tables) return rc_tables;
  const size_t n = (size_t)c->width * (c->world > 1 ? c->shard_rows() : c->height);
  // layout 2: on the slot's own stream, behind its k_shade (the other slots' streams keep the GPU busy meanwhile)
  hipStream_t ps = (c->pipelined() && s.stream_used && s.stream_used != c->shade_stream()) ? s.stream_used : c->present_stream();
  if (ps != s.stream_used) HIP_TRY(c, hipStreamWaitEvent(ps, s.ev_shade_done, 0));  // behind the frame's k_shade
  hipEvent_t *pe = nullptr;
  if (c->timing) {
    if (c->present_ring.empty()) {
      c->present_ring.resize(2 * bbr_context::kRingCap);
      for (auto &e : c->present_ring) HIP_TRY(c, hipEventCreate(&e));
    }
    pe = &c->present_ring[2 * (c->present_launches % bbr_context::kRingCap)];
    HIP_TRY(c, hipEventRecord(pe[0], ps));
  }
  const size_t per_block = (size_t)kPresentThreads * kPresentPerThread;
  hipLaunchKernelGGL(k_present, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(kPresentThreads), 0,
                     ps, s.out_used, s.present.out, n, c->d_srgb_tables.ptr, s.present.enable,
                     s.present.exposure, s.present.hdr16);
  HIP_TRY(c, hipGetLastError());
  if (pe) {
    HIP_TRY(c, hipEventRecord(pe[1], ps));
    ++c->present_launches;
  }
  // the slot is busy until the presented image exists
  HIP_TRY(c, hipEventRecord(s.ev_shade_done, ps));
  return BBR_OK;
}

int submit_frame(bbr_context *c) {
  const auto t0 = std::chrono::steady_clock::now();
  int slot = (int)(c->frame_counter % (uint64_t)c->n_slots());
  int rc = submit_frame_into(c, slot);
  if (rc == BBR_OK) ++c->frame_counter;
  c->host_submit_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  ++c->host_frames;
  return rc;
}

// Synchronise and, if a capacity overflowed in the most recent frame, grow it and render that frame again.
// Render the recorded frame once more into the slot it was last rendered in (after a capacity growth, or for a diagnostic
// dump) and put the slot's presentation state back: the presented image is made again from the new frame, a caller's
// copy of it is refreshed, and bbr_read_presented / bbr_draw_overlays keep working afterwards.
int resubmit_last_frame(bbr_context *c) {
  FrameSlot &s = c->slots[c->last_slot];
  const auto present = s.present;
  int rc = submit_frame_into(c, c->last_slot);
  if (rc) return rc;
  if (present.active && !s.fused) {
    s.present = present;
    rc = queue_present(c, s);
    if (rc) return rc;
  } else if (s.fused && present.copy_to) {  // fused: the re-rendered frame is the image; redo the caller's copy
    const size_t n = (size_t)c->width * (c->world > 1 ? c->shard_rows() : c->height);
    s.present.copy_to = present.copy_to;
    HIP_TRY(c, hipStreamWaitEvent(c->present_stream(), s.ev_shade_done, 0));
    HIP_TRY(c, hipMemcpyAsync(present.copy_to, s.d_present.ptr, n * 4, hipMemcpyDeviceToDevice, c->present_stream()));
    HIP_TRY(c, hipEventRecord(s.ev_shade_done, c->present_stream()));
  }
  return BBR_OK;
}

int sync_and_fix(bbr_context *c, Counters *out_counters) {
  for (int attempt = 0; attempt < 8; ++attempt) {
    int rc = drain(c);
    if (rc) return rc;
    Counters h = {};
    if (c->have_frame && c->last_slot >= 0 && c->d_counters.ptr)
      HIP_TRY(c, hipMemcpy(&h, c->d_counters_done.ptr + c->slots[c->last_slot].ctr_index, sizeof h, hipMemcpyDeviceToHost));
    if (out_counters) *out_counters = h;
    if (!h.overflow) return BBR_OK;
    rc = apply_growth(c, h.overflow, h.bin_need, h.n_broad, h.n_clip_slots);
    if (rc) return rc;
    rc = resubmit_last_frame(c);
    if (rc) return rc;
  }
  return fail(c, BBR_ERR_CAPACITY, "bin capacity still exceeded after 8 growth steps");
}

const void *last_output(const bbr_context *c) {
  return c->last_slot >= 0 ? (const void *)c->slots[c->last_slot].out_used : nullptr;
}

}  // namespace

// ================================================================================================
// overlay subpass: light markers + corner gizmo over the presented image
// ================================================================================================
namespace {

int upload_internal_mesh(bbr_context *c, Mesh &m, const std::vector<Vertex> &v, const std::vector<uint32_t> &idx) {
  if (m.d_vertices) (void)hipFree(m.d_vertices);
  if (m.d_indices) (void)hipFree(m.d_indices);
  m = Mesh();
  HIP_TRY(c, hipMalloc(&m.d_vertices, v.size() * sizeof(Vertex)));
  HIP_TRY(c, upload_sync(m.d_vertices, v.data(), v.size() * sizeof(Vertex)));
  HIP_TRY(c, hipMalloc(&m.d_indices, idx.size() * sizeof(uint32_t)));
  HIP_TRY(c, upload_sync(m.d_indices, idx.data(), idx.size() * sizeof(uint32_t)));
  m.n_vertices = (uint32_t)v.size();
  m.n_indices = (uint32_t)idx.size();
  m.alive = true;
  return BBR_OK;
}

// generateUVSphereMesh(0.1f, 16, 16) as the light markers use it: positions only (src/main.cpp:953-957,
// src/render.cpp:1774-1833; sphericalToCartesian src/vector_math.cpp:284-292; pi32 = 3.141592f)
int ensure_marker_mesh(bbr_context *c) {
  if (c->marker_mesh.alive) return BBR_OK;
  constexpr float pi32 = 3.141592f, half_pi32 = pi32 * 0.5f, two_pi32 = pi32 * 2.f, radius = 0.1f;
  constexpr int hdiv = 16, vdiv = 16;
  std::vector<Vertex> v;
  std::vector<uint32_t> idx;
  for (int iv = 0; iv <= vdiv; ++iv) {
    const float theta = -half_pi32 + pi32 * ((float)iv / (float)vdiv);
    for (int ih = 0; ih <= hdiv; ++ih) {
      const float phi = two_pi32 * ((float)ih / (float)hdiv);
      const float cos_theta = cosf(theta);
      Vertex vx = {};
      vx.pos[0] = radius * cos_theta * cosf(phi);
      vx.pos[1] = radius * sinf(theta);
      vx.pos[2] = radius * cos_theta * sinf(phi);
      v.push_back(vx);
    }
  }
  for (int iv = 0; iv < vdiv; ++iv)
    for (int ih = 0; ih < hdiv; ++ih) {
      const uint32_t base = (uint32_t)((hdiv + 1) * iv + ih);
      if (iv < vdiv - 1) { idx.push_back(base); idx.push_back(base + hdiv + 1); idx.push_back(base + hdiv + 2); }
      if (iv > 0) { idx.push_back(base + hdiv + 2); idx.push_back(base + 1); idx.push_back(base); }
    }
  return upload_internal_mesh(c, c->marker_mesh, v, idx);
}

inline float dot3_fma(const float *a, const float *b) { return std::fmaf(a[2], b[2], std::fmaf(a[1], b[1], a[0] * b[0])); }

// column-major M * v with the vertex stage's fma chain
inline void mat_vec(const Mat4 &m, const float *v, float *out) {
  for (int i = 0; i < 4; ++i)
    out[i] = std::fmaf(m.M[3][i], v[3], std::fmaf(m.M[2][i], v[2], std::fmaf(m.M[1][i], v[1], m.M[0][i] * v[0])));
}

}  // namespace

extern "C" int bbr_upload_gizmo(bbr_context *c, const void *gizmo_vertices, uint32_t n_vertices, const uint32_t *indices,
                                uint32_t n_indices) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  if (!gizmo_vertices || !n_vertices) return fail(c, BBR_ERR_INVALID_ARGUMENT, "upload_gizmo: null/empty input");
  int rc = drain(c);
  if (rc) return rc;
  // bb::GizmoVertex {Pos, Color, Normal} (src/render.h:122-126) -> Vertex {pos, uv, normal, tangent := colour}
  const float *g = static_cast<const float *>(gizmo_vertices);
  std::vector<Vertex> v(n_vertices);
  for (uint32_t i = 0; i < n_vertices; ++i) {
    Vertex vx = {};
    for (int k = 0; k < 3; ++k) {
      vx.pos[k] = g[9 * i + k];
      vx.tangent[k] = g[9 * i + 3 + k];
      vx.normal[k] = g[9 * i + 6 + k];
    }
    v[i] = vx;
  }
  std::vector<uint32_t> idx;
  if (indices) {
    for (uint32_t i = 0; i < n_indices; ++i)
      if (indices[i] >= n_vertices) return fail(c, BBR_ERR_INVALID_ARGUMENT, "upload_gizmo: index out of range");
    idx.assign(indices, indices + n_indices / 3 * 3);
  } else {
    for (uint32_t i = 0; i < n_vertices / 3 * 3; ++i) idx.push_back(i);
  }
  return upload_internal_mesh(c, c->gizmo_mesh, v, idx);
}

extern "C" int bbr_draw_overlays(bbr_context *c, int32_t gizmo_extent) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!c->have_frame || c->last_slot < 0) return fail(c, BBR_ERR_NOT_IN_FRAME, "draw_overlays: nothing rendered");
  if (c->world > 1) return fail(c, BBR_ERR_INVALID_ARGUMENT, "draw_overlays: not available with a partition");
  int rc = sync_and_fix(c, nullptr);  // the overlay pass is synchronous: it is a debugging aid, not part of the hot path
  if (rc) return rc;
  FrameSlot &fs = c->slots[c->last_slot];
  if (!fs.has_depth) return fail(c, BBR_ERR_INVALID_ARGUMENT, "draw_overlays: the frame was rendered without option \"overlays\"");
  if (!fs.present.active) return fail(c, BBR_ERR_NOT_IN_FRAME, "draw_overlays: call bbr_present first (overlays go into the presented image)");
  rc = ensure_marker_mesh(c);
  if (rc) return rc;
  rc = ensure_srgb_tables(c);
  if (rc) return rc;

  // ---- the pass's draw list: instanced markers, then the gizmo (src/main.cpp:138-171) ----
  const int n_lights = std::max(0, std::min(fs.frame_u.num_lights, kMaxNumLights));
  const bool gizmo = gizmo_extent > 0 && c->gizmo_mesh.alive;
  const Mat4 pv = proj_view(fs.view_u);  // uProjMat * uViewMat
  std::vector<InstanceBlock> inst((size_t)n_lights + (gizmo ? 1 : 0));
  for (int i = 0; i < n_lights; ++i) {
    const Light &l = fs.frame_u.lights[i];
    InstanceBlock ib = {};
    ib.model = pv;  // (P*V) * modelMat, modelMat = identity with column 3 = (pos, 1): light.vert:11-14
    const float p[4] = {l.pos[0], l.pos[1], l.pos[2], 1.0f};
    mat_vec(pv, p, ib.model.M[3]);
    ib.inv_model.M[0][0] = l.color[0]; ib.inv_model.M[0][1] = l.color[1]; ib.inv_model.M[0][2] = l.color[2];
    inst[i] = ib;
  }
  if (gizmo) {  // gizmo.vert:13-24
    const Mat4 &uv = fs.view_u.view;
    const float right[3] = {uv.M[0][0], uv.M[1][0], uv.M[2][0]}, up[3] = {uv.M[0][1], uv.M[1][1], uv.M[2][1]};
    const float look[3] = {uv.M[0][2], uv.M[1][2], uv.M[2][2]};
    const float view_pos[3] = {look[0] * -27.0f, look[1] * -27.0f, look[2] * -27.0f};
    ViewUniformBlock gv = fs.view_u;
    gv.view.M[3][0] = -dot3_fma(view_pos, right);
    gv.view.M[3][1] = -dot3_fma(view_pos, up);
    gv.view.M[3][2] = -dot3_fma(view_pos, look);
    const float d = 1.0f / tanf(0.261799f);
    gv.proj.M[0][0] = d;
    gv.proj.M[1][1] = -d;
    InstanceBlock ib = {};
    ib.model = proj_view(gv);
    ib.inv_model = gv.view;
    inst[(size_t)n_lights] = ib;
  }
  const uint32_t marker_tris = c->marker_mesh.n_indices / 3, gizmo_tris = gizmo ? c->gizmo_mesh.n_indices / 3 : 0;
  const uint32_t n_prims = marker_tris * (uint32_t)n_lights + gizmo_tris;
  if (!n_prims) return BBR_OK;
  std::vector<DrawDesc> draws;
  FrameSlot &s = c->ov;
  const size_t draws_bytes = 2 * sizeof(DrawDesc), inst_bytes = inst.size() * sizeof(InstanceBlock);
  HIP_TRY(c, s.d_staging.ensure(draws_bytes + inst_bytes));
  const InstanceBlock *d_inst = reinterpret_cast<const InstanceBlock *>(s.d_staging.ptr + draws_bytes);
  if (n_lights) {
    DrawDesc d = {};
    d.vertices = c->marker_mesh.d_vertices; d.indices = c->marker_mesh.d_indices; d.instances = d_inst;
    d.n_instances = (uint32_t)n_lights; d.tris_per_instance = marker_tris; d.first_prim = 0; d.material = 1u;  // program: marker
    draws.push_back(d);
  }
  if (gizmo) {
    DrawDesc d = {};
    d.vertices = c->gizmo_mesh.d_vertices; d.indices = c->gizmo_mesh.d_indices; d.instances = d_inst + n_lights;
    d.n_instances = 1; d.tris_per_instance = gizmo_tris; d.first_prim = marker_tris * (uint32_t)n_lights; d.material = 2u;  // program: gizmo
    draws.push_back(d);
  }
  HIP_TRY(c, upload_sync(s.d_staging.ptr, draws.data(), draws.size() * sizeof(DrawDesc)));
  HIP_TRY(c, upload_sync(s.d_staging.ptr + draws_bytes, inst.data(), inst_bytes));
  const DrawDesc *d_draws = reinterpret_cast<const DrawDesc *>(s.d_staging.ptr);
  FirstPrims ov_first = {{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}};
  for (size_t q = 1; q < draws.size() && q <= (size_t)kInlineFirstPrims; ++q) ov_first.v[q - 1] = draws[q].first_prim;

  for (int attempt = 0; attempt < 8; ++attempt) {
    const size_t tiles = (size_t)c->tiles_x() * c->tiles_y();
    HIP_TRY(c, s.d_tris.ensure(n_prims));
    HIP_TRY(c, s.d_attrs.ensure(n_prims));
    HIP_TRY(c, s.d_clip.ensure(c->clip_cap));
    HIP_TRY(c, c->d_counters.ensure(bbr_context::kCounterBlocks, true));
    HIP_TRY(c, s.d_block_stats.ensure(((n_prims + 255) / 256) * 4, true));
    HIP_TRY(c, s.d_tile_count.ensure(tiles * kBinClasses, true));
    HIP_TRY(c, s.d_bins.ensure(tiles * kBinClasses * c->bin_cap));
    // (at least kBroadSpec entries: k_raster's light-tile path reads that many before it knows how many the frame wrote)
  HIP_TRY(c, s.d_broad.ensure(std::max<size_t>(c->broad_cap, kBroadSpec), true));
    HIP_TRY(c, s.d_frags.ensure(tiles * (size_t)(c->tile_w() * c->tile_h())));
    HIP_TRY(c, s.d_frag_count.ensure(tiles, true));
    FrameParams fp = make_params(c);
    fp.deferred = 0;
    fp.gbuffer_view = -1;
    fp.ov_first_gizmo_prim = gizmo ? marker_tris * (uint32_t)n_lights : 0xFFFFFFFFu;
    const int x0 = c->width - gizmo_extent;
    fp.ov_half = 0.5f * (float)gizmo_extent;
    fp.ov_cx = (float)x0 + fp.ov_half;
    fp.ov_cy = fp.ov_half;
    fp.ov_x0 = std::max(x0, 0); fp.ov_y0 = 0; fp.ov_x1 = c->width; fp.ov_y1 = std::min(gizmo_extent, c->height);
    s.ctr_index = bbr_context::kMaxSlots;
    Counters *ctr = c->d_counters.ptr + s.ctr_index;
    hipStream_t st = c->shade_stream();
    const Mat4 ident = {};
    auto launch = [&](auto tw, auto th) {
      constexpr int TW = decltype(tw)::value, TH = decltype(th)::value;
      hipLaunchKernelGGL((k_geometry<TW, TH, true>), dim3((n_prims + 255) / 256), dim3(256), 0, st, d_draws, (uint32_t)draws.size(),
                         n_prims, ov_first, s.d_tris.ptr, s.d_attrs.ptr, s.d_tile_count.ptr, s.d_bins.ptr, ctr, ident, ident, fp, s.d_clip.ptr,
                         s.d_broad.ptr, c->d_materials.ptr, s.d_block_stats.ptr, (uint32_t *)nullptr);
      hipLaunchKernelGGL((k_raster<TW, TH, true>), dim3(fp.tiles_x, fp.tiles_y), dim3(kTileThreads), 0, st, s.d_tile_count.ptr, ctr,
                         s.d_broad.ptr, s.d_frag_count.ptr, s.d_frags.ptr, (float4 *)nullptr, fp, s.d_tris.ptr, s.d_clip.ptr,
                         s.d_bins.ptr, (uint32_t *)nullptr, (float *)nullptr,
                         (const float4 *)nullptr, fs.d_depth.ptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                         (uint32_t *)nullptr, (uint32_t *)nullptr, (const Light *)nullptr, 0, (CookedLight *)nullptr,
                         (const uint32_t *)nullptr);
      constexpr int kChunks = TW * TH / kShadeThreads;
      hipLaunchKernelGGL((k_shade_overlay<TW, TH>), dim3(fp.tiles_x * kChunks, fp.tiles_y), dim3(kShadeThreads), 0, st, fp,
                         s.d_attrs.ptr, s.d_clip.ptr, s.d_frags.ptr, s.d_frag_count.ptr, c->d_srgb_tables.ptr, fs.present.out);
    };
    if (c->tile_mode == 0) launch(std::integral_constant<int, 64>{}, std::integral_constant<int, 64>{});
    else launch(std::integral_constant<int, 32>{}, std::integral_constant<int, 32>{});
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(st));
    Counters h = {};
    HIP_TRY(c, hipMemcpy(&h, ctr, sizeof h, hipMemcpyDeviceToHost));
    HIP_TRY(c, zero_fill_sync(ctr, sizeof(Counters)));  // (the pass is synchronous: cleared here, not by a kernel)
    if (!h.overflow) return BBR_OK;
    // an overlay triangle did not fit: the presented pixels it already wrote are a subset of the right ones (same
    // colours), so growing and drawing again on top is correct
    rc = apply_growth(c, h.overflow, h.bin_need, h.n_broad, h.n_clip_slots);
    if (rc) return rc;
    if (s.d_tile_count.ptr) HIP_TRY(c, zero_fill_sync(s.d_tile_count.ptr, s.d_tile_count.cap * sizeof(uint32_t)));
  }
  return fail(c, BBR_ERR_CAPACITY, "draw_overlays: capacity still exceeded after 8 growth steps");
}

// ================================================================================================
// C ABI
// ================================================================================================

extern "C" {

int bbr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *bbr_last_error(const bbr_context *ctx) { return ctx ? ctx->last_error.c_str() : g_create_error.c_str(); }

// Everything a context owns, whatever state its construction reached (bbr_destroy, and bbr_create when a step fails).
static void release_context(bbr_context *c) {
  (void)hipSetDevice(c->device);
  (void)drain(c);
  if (c->comm) (void)g_rccl.comm_destroy(c->comm);
  for (Mesh &m : c->meshes) {
    if (m.d_vertices) (void)hipFree(m.d_vertices);
    if (m.d_indices) (void)hipFree(m.d_indices);
  }
  for (Material &m : c->materials) {
    for (auto &p : m.d_texels)
      if (p) (void)hipFree(p);
    if (m.d_packed) (void)hipFree(m.d_packed);
  }
  if (c->d_default_texels) (void)hipFree(c->d_default_texels);
  c->d_materials.release();
  c->d_counters.release();
  c->d_counters_done.release();
  c->d_vis_prim.release();
  c->d_vis_depth.release();
  c->d_gbuffer.release();
  for (FrameSlot &s : c->slots) {
    s.release_all();
    if (s.ev_geom_done) (void)hipEventDestroy(s.ev_geom_done);
    if (s.ev_raster_done) (void)hipEventDestroy(s.ev_raster_done);
    if (s.ev_shade_done) (void)hipEventDestroy(s.ev_shade_done);
    if (s.ev_tail_done) (void)hipEventDestroy(s.ev_tail_done);
  }
  for (Mesh *m : {&c->marker_mesh, &c->gizmo_mesh}) {
    if (m->d_vertices) (void)hipFree(m->d_vertices);
    if (m->d_indices) (void)hipFree(m->d_indices);
  }
  c->ov.release_all();
  if (c->ov.ev_geom_done) (void)hipEventDestroy(c->ov.ev_geom_done);
  if (c->ov.ev_raster_done) (void)hipEventDestroy(c->ov.ev_raster_done);
  if (c->ov.ev_shade_done) (void)hipEventDestroy(c->ov.ev_shade_done);
  if (c->ov.ev_tail_done) (void)hipEventDestroy(c->ov.ev_tail_done);
  for (auto &e : c->ring)
    if (e) (void)hipEventDestroy(e);
  for (auto &e : c->present_ring)
    if (e) (void)hipEventDestroy(e);
  c->d_srgb_tables.release();
  if (c->s_geom) (void)hipStreamDestroy(c->s_geom);
  if (c->s_raster) (void)hipStreamDestroy(c->s_raster);
  if (c->s_shade) (void)hipStreamDestroy(c->s_shade);
  if (c->s_present) (void)hipStreamDestroy(c->s_present);
  delete c;
}

int bbr_create(int32_t width, int32_t height, int32_t device, bbr_context **out_ctx) {
  if (!out_ctx) return fail(nullptr, BBR_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
  *out_ctx = nullptr;
  if (width <= 0 || height <= 0 || width > 32768 || height > 32768)
    return fail(nullptr, BBR_ERR_INVALID_ARGUMENT, "width/height out of range");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(nullptr, BBR_ERR_NO_DEVICE, "no HIP device: this library has no CPU fallback");
  if (device < 0 || device >= n) return fail(nullptr, BBR_ERR_INVALID_ARGUMENT, "device index out of range");
  bbr_context *c = new bbr_context();
  c->device = device;
  c->width = width;
  c->height = height;
#define CREATE_TRY(expr)                                                                       \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      g_create_error = std::string(#expr) + ": " + hipGetErrorString(_e);                      \
      release_context(c);                                                                      \
      return _e == hipErrorOutOfMemory ? BBR_ERR_OUT_OF_MEMORY : BBR_ERR_HIP;                  \
    }                                                                                          \
  } while (0)
  CREATE_TRY(hipSetDevice(device));
  {
    int cus = 0;
    CREATE_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    if (cus > 0) c->n_cus = cus;
  }
  {
    // geometry + raster are latency-bound and short: give them the high-priority queue so that their workgroups
    // slot in between the (ALU-bound, GPU-filling) shade kernel of the previous frame
    int prio_low = 0, prio_high = 0;
    CREATE_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    // (measured on C3/C5: swapping or equalising the two priorities changes the pipelined frame time by < 2 %)
    CREATE_TRY(hipStreamCreateWithPriority(&c->s_geom, hipStreamNonBlocking, prio_high));
    // (created here, next to s_geom: a stream made later, once the others are busy, was measured to serialise with them)
    CREATE_TRY(hipStreamCreateWithPriority(&c->s_raster, hipStreamNonBlocking, prio_high));
    CREATE_TRY(hipStreamCreateWithPriority(&c->s_shade, hipStreamNonBlocking, prio_low));
    CREATE_TRY(hipStreamCreateWithPriority(&c->s_present, hipStreamNonBlocking, prio_low));
  }
  // `default` material maps (resources/pbr/default/*.png are uniform images): 1x1 RGBA8 each
  static const uint8_t k_default[kMapCount][4] = {{255, 255, 255, 255}, {0, 0, 0, 255},       {0, 0, 0, 255},
                                                  {255, 255, 255, 255}, {127, 127, 255, 255}, {0, 0, 0, 255}};
  CREATE_TRY(hipMalloc(&c->d_default_texels, sizeof k_default));
  CREATE_TRY(upload_sync(c->d_default_texels, k_default, sizeof k_default));
#undef CREATE_TRY
  {
    std::lock_guard<std::mutex> lock(g_live_mutex);
    g_live.insert(c);
  }
  *out_ctx = c;
  return BBR_OK;
}

int bbr_destroy(bbr_context *c) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  {
    std::lock_guard<std::mutex> lock(g_live_mutex);
    if (!g_live.erase(c)) return BBR_ERR_BAD_HANDLE;  // never created, or destroyed already
  }
  release_context(c);
  return BBR_OK;
}

int bbr_upload_mesh(bbr_context *c, const void *vertices, uint32_t n_vertices, const uint32_t *indices,
                    uint32_t n_indices, int32_t *out_mesh) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!vertices || !n_vertices || !out_mesh) return fail(c, BBR_ERR_INVALID_ARGUMENT, "upload_mesh: null/empty input");
  if (indices) {
    for (uint32_t i = 0; i < n_indices; ++i)
      if (indices[i] >= n_vertices) return fail(c, BBR_ERR_INVALID_ARGUMENT, "upload_mesh: index out of range");
  } else if (n_indices) {
    return fail(c, BBR_ERR_INVALID_ARGUMENT, "upload_mesh: n_indices without indices");
  }
  Mesh m;
  m.n_vertices = n_vertices;
  m.n_indices = indices ? n_indices : 0;
  HIP_TRY(c, hipMalloc(&m.d_vertices, (size_t)n_vertices * sizeof(Vertex)));
  HIP_TRY(c, upload_sync(m.d_vertices, vertices, (size_t)n_vertices * sizeof(Vertex)));
  if (m.n_indices) {
    HIP_TRY(c, hipMalloc(&m.d_indices, (size_t)n_indices * sizeof(uint32_t)));
    HIP_TRY(c, upload_sync(m.d_indices, indices, (size_t)n_indices * sizeof(uint32_t)));
  }
  m.alive = true;
  c->meshes.push_back(m);
  *out_mesh = (int32_t)c->meshes.size() - 1;
  return BBR_OK;
}

int bbr_free_mesh(bbr_context *c, int32_t mesh) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  if (!is_live(c)) return BBR_ERR_BAD_HANDLE;  // the context is gone and took the mesh with it (checked before c is touched)
  BBR_ON_DEVICE(c);
  if (mesh < 0 || mesh >= (int32_t)c->meshes.size() || !c->meshes[mesh].alive)
    return fail(c, BBR_ERR_BAD_HANDLE, "free_mesh: bad handle");
  int rc = drain(c);
  if (rc) return rc;
  Mesh &m = c->meshes[mesh];
  if (m.d_vertices) (void)hipFree(m.d_vertices);
  if (m.d_indices) (void)hipFree(m.d_indices);
  m = Mesh();
  c->have_frame = false;
  return BBR_OK;
}

int bbr_upload_material(bbr_context *c, const bbr_image maps[BBR_MAP_COUNT], int32_t *out_material) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!maps || !out_material) return fail(c, BBR_ERR_INVALID_ARGUMENT, "upload_material: null input");
  Material m;
  struct Guard {  // frees what a failed upload had already allocated
    Material *m;
    ~Guard() {
      if (!m) return;
      for (auto &p : m->d_texels)
        if (p) (void)hipFree(p);
      if (m->d_packed) (void)hipFree(m->d_packed);
    }
  } guard{&m};
  for (int i = 0; i < kMapCount; ++i) {
    const bbr_image &im = maps[i];
    if (im.rgba && im.width > 0 && im.height > 0) {
      if (im.width > 16384 || im.height > 16384) return fail(c, BBR_ERR_INVALID_ARGUMENT, "upload_material: map too large");
      size_t bytes = (size_t)im.width * im.height * 4;
      HIP_TRY(c, hipMalloc(&m.d_texels[i], bytes));
      HIP_TRY(c, upload_sync(m.d_texels[i], im.rgba, bytes));
      m.desc.maps[i] = TexDesc{m.d_texels[i], im.width, im.height};
    } else {
      // missing map => the `default` material's map (src/render.cpp:1328-1336)
      m.desc.maps[i] = TexDesc{c->d_default_texels + 4 * i, 1, 1};
    }
  }
  // Interleave the five shaded maps when they agree on one size (missing maps are uniform, so they broadcast):
  // a bilinear tap then costs one 12-byte load of a 9-byte record instead of five 4-byte loads from five arrays.
  {
    static const uint8_t k_default[kMapCount][4] = {{255, 255, 255, 255}, {0, 0, 0, 255},       {0, 0, 0, 255},
                                                    {255, 255, 255, 255}, {127, 127, 255, 255}, {0, 0, 0, 255}};
    const int used[5] = {kMapAlbedo, kMapMetallic, kMapRoughness, kMapAO, kMapNormal};
    int pw = 1, ph = 1;
    bool ok = true;
    for (int k : used) {
      const bbr_image &im = maps[k];
      if (!(im.rgba && im.width > 0 && im.height > 0)) continue;
      if (pw == 1 && ph == 1) {
        pw = im.width;
        ph = im.height;
      } else if (im.width != pw || im.height != ph) {
        ok = false;
      }
    }
    if (ok) {
      // block-linear: 4 x 4 texel blocks of 144 bytes, blocks in row-major order (packed_texel_index in bb_kernels.hip.h):
      // the 2 x 2 footprint of a bilinear tap set then falls into one block 9 times out of 16
      const size_t n_texels = (size_t)pw * ph;
      const size_t w4 = ((size_t)pw + 3) / 4, h4 = ((size_t)ph + 3) / 4;
      std::vector<uint8_t> host(w4 * h4 * 16 * kPackedTexelBytes + kPackedTexelPad, 0);
      auto texel = [&](int k, size_t i) -> const uint8_t * {
        const bbr_image &im = maps[k];
        return (im.rgba && im.width > 0 && im.height > 0) ? im.rgba + 4 * i : k_default[k];
      };
      for (size_t i = 0; i < n_texels; ++i) {
        const uint8_t *al = texel(kMapAlbedo, i), *me = texel(kMapMetallic, i), *ro = texel(kMapRoughness, i);
        const uint8_t *ao = texel(kMapAO, i), *no = texel(kMapNormal, i);
        const size_t x = i % (size_t)pw, y = i / (size_t)pw;
        uint8_t *t = host.data() + (((y >> 2) * w4 + (x >> 2)) * 16 + (y & 3) * 4 + (x & 3)) * kPackedTexelBytes;
        t[0] = al[0]; t[1] = al[1]; t[2] = al[2]; t[3] = me[0];
        t[4] = no[0]; t[5] = no[1]; t[6] = no[2]; t[7] = ro[0];
        t[8] = ao[0];
      }
      HIP_TRY(c, hipMalloc(&m.d_packed, host.size()));
      HIP_TRY(c, upload_sync(m.d_packed, host.data(), host.size()));
      m.desc.packed = m.d_packed;
      m.desc.pw = pw;
      m.desc.ph = ph;
    }
  }
  m.alive = true;
  guard.m = nullptr;  // ownership passes to the context
  c->materials.push_back(m);
  c->materials_dirty = true;
  *out_material = (int32_t)c->materials.size() - 1;
  return BBR_OK;
}

int bbr_free_material(bbr_context *c, int32_t material) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  if (!is_live(c)) return BBR_ERR_BAD_HANDLE;  // the context is gone and took the material with it (checked before c is touched)
  BBR_ON_DEVICE(c);
  if (material < 0 || material >= (int32_t)c->materials.size() || !c->materials[material].alive)
    return fail(c, BBR_ERR_BAD_HANDLE, "free_material: bad handle");
  int rc = drain(c);
  if (rc) return rc;
  Material &m = c->materials[material];
  for (auto &p : m.d_texels)
    if (p) (void)hipFree(p);
  if (m.d_packed) (void)hipFree(m.d_packed);
  m = Material();
  c->materials_dirty = true;
  c->have_frame = false;
  return BBR_OK;
}

int bbr_set_frame_uniforms(bbr_context *c, const void *block) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!block) return fail(c, BBR_ERR_INVALID_ARGUMENT, "set_frame_uniforms: NULL");
  std::memcpy(&c->frame_u, block, sizeof(FrameUniformBlock));
  // the reference asserts NumLights < MAX_NUM_LIGHTS (src/main.cpp:1289-1290)
  if (c->frame_u.num_lights < 0 || c->frame_u.num_lights >= kMaxNumLights)
    return fail(c, BBR_ERR_INVALID_ARGUMENT, "set_frame_uniforms: NumLights must be in [0, 100)");
  return BBR_OK;
}

int bbr_set_view_uniforms(bbr_context *c, const void *block) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!block) return fail(c, BBR_ERR_INVALID_ARGUMENT, "set_view_uniforms: NULL");
  std::memcpy(&c->view_u, block, sizeof(ViewUniformBlock));
  return BBR_OK;
}

int bbr_begin_frame(bbr_context *c) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  c->draws.clear();
  c->host_instances.clear();
  c->n_prims = 0;
  c->in_frame = true;
  c->have_frame = false;
  return BBR_OK;
}

int bbr_draw(bbr_context *c, int32_t mesh, int32_t material, const void *instance_blocks, uint32_t n_instances) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!c->in_frame) return fail(c, BBR_ERR_NOT_IN_FRAME, "draw outside begin_frame/end_frame");
  if (mesh < 0 || mesh >= (int32_t)c->meshes.size() || !c->meshes[mesh].alive)
    return fail(c, BBR_ERR_BAD_HANDLE, "draw: bad mesh handle");
  if (material < 0 || material >= (int32_t)c->materials.size() || !c->materials[material].alive)
    return fail(c, BBR_ERR_BAD_HANDLE, "draw: bad material handle");
  if (n_instances && !instance_blocks) return fail(c, BBR_ERR_INVALID_ARGUMENT, "draw: NULL instance data");
  const Mesh &m = c->meshes[mesh];
  uint32_t tris = (m.d_indices ? m.n_indices : m.n_vertices) / 3;
  uint64_t total = (uint64_t)c->n_prims + (uint64_t)tris * n_instances;
  if (total >= kMaxPrims) return fail(c, BBR_ERR_TOO_MANY_PRIMITIVES, "draw: more than 2^29 primitives in one frame");
  RecordedDraw rd;
  rd.mesh = mesh;
  rd.material = material;
  rd.n_instances = n_instances;
  rd.first_instance = (uint32_t)c->host_instances.size();
  rd.first_prim = c->n_prims;
  rd.tris_per_instance = tris;
  const InstanceBlock *ib = static_cast<const InstanceBlock *>(instance_blocks);
  c->host_instances.insert(c->host_instances.end(), ib, ib + n_instances);
  c->draws.push_back(rd);
  c->n_prims = (uint32_t)total;
  return BBR_OK;
}

int bbr_end_frame(bbr_context *c) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!c->in_frame) return fail(c, BBR_ERR_NOT_IN_FRAME, "end_frame without begin_frame");
  c->in_frame = false;
  c->have_frame = true;
  return submit_frame(c);
}

int bbr_replay_frame(bbr_context *c) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!c->have_frame) return fail(c, BBR_ERR_NOT_IN_FRAME, "replay_frame: no recorded frame");
  return submit_frame(c);
}

int bbr_synchronize(bbr_context *c) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  return sync_and_fix(c, nullptr);
}

int bbr_read_framebuffer(bbr_context *c, float *host) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!host) return fail(c, BBR_ERR_INVALID_ARGUMENT, "read_framebuffer: NULL");
  if (c->world > 1) return fail(c, BBR_ERR_INVALID_ARGUMENT, "read_framebuffer on a partitioned context: use bbr_read_shard");
  if (!c->have_frame) return fail(c, BBR_ERR_NOT_IN_FRAME, "read_framebuffer: nothing rendered");
  if (c->last_slot >= 0 && c->slots[c->last_slot].fused)
    return fail(c, BBR_ERR_INVALID_ARGUMENT, "read_framebuffer: the frame was rendered with option present_fused (no fp32 frame); use bbr_read_presented");
  int rc = sync_and_fix(c, nullptr);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpy(host, last_output(c), (size_t)c->width * c->height * 16, hipMemcpyDeviceToHost));
  return BBR_OK;
}

int bbr_read_shard(bbr_context *c, float *host) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!host) return fail(c, BBR_ERR_INVALID_ARGUMENT, "read_shard: NULL");
  if (!c->have_frame) return fail(c, BBR_ERR_NOT_IN_FRAME, "read_shard: nothing rendered");
  if (c->last_slot >= 0 && c->slots[c->last_slot].fused)
    return fail(c, BBR_ERR_INVALID_ARGUMENT, "read_shard: the frame was rendered with option present_fused (no fp32 frame); use bbr_read_presented");
  int rc = sync_and_fix(c, nullptr);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpy(host, last_output(c), (size_t)c->width * c->shard_rows() * 16, hipMemcpyDeviceToHost));
  return BBR_OK;
}

int bbr_framebuffer_device_ptr(bbr_context *c, void **out_ptr, uint64_t *out_bytes) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!out_ptr) return fail(c, BBR_ERR_INVALID_ARGUMENT, "framebuffer_device_ptr: NULL");
  if (!c->have_frame || c->last_slot < 0) return fail(c, BBR_ERR_NOT_IN_FRAME, "framebuffer_device_ptr: nothing rendered");
  *out_ptr = (void *)c->slots[c->last_slot].out_used;
  if (out_bytes)
    *out_bytes = c->ext_out ? c->ext_out_bytes : (uint64_t)c->width * std::max(c->height, c->shard_rows()) * 16;
  return BBR_OK;
}

int bbr_set_output_device_ptr(bbr_context *c, void *device_ptr, uint64_t bytes) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (device_ptr) {
    uint64_t need = (uint64_t)c->width * (c->world > 1 ? c->shard_rows() : c->height) * 16;
    if (bytes < need) return fail(c, BBR_ERR_INVALID_ARGUMENT, "set_output_device_ptr: buffer too small");
    if ((uintptr_t)device_ptr & 15u) return fail(c, BBR_ERR_INVALID_ARGUMENT, "set_output_device_ptr: need 16-byte alignment");
  }
  c->ext_out = device_ptr;
  c->ext_out_bytes = device_ptr ? bytes : 0;
  return BBR_OK;
}

int bbr_set_stream(bbr_context *c, void *stream) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  int rc = drain(c);
  if (rc) return rc;
  c->user_stream = (hipStream_t)stream;
  c->frame_counter = 0;
  return BBR_OK;
}

int bbr_wait_event(bbr_context *c, void *hip_event) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!hip_event) return fail(c, BBR_ERR_INVALID_ARGUMENT, "wait_event: NULL");
  c->pending_waits.push_back((hipEvent_t)hip_event);  // the next frame's first stream waits for it (which stream that is
                                                      // depends on the stream layout)
  return BBR_OK;
}

int bbr_stream_wait_frame(bbr_context *c, void *stream) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!c->have_frame || c->last_slot < 0) return fail(c, BBR_ERR_NOT_IN_FRAME, "stream_wait_frame: nothing rendered");
  HIP_TRY(c, hipStreamWaitEvent((hipStream_t)stream, c->slots[c->last_slot].ev_shade_done, 0));
  return BBR_OK;
}

int bbr_tile_height(const bbr_context *c, int32_t *out) {
  if (!c || !out) return BBR_ERR_INVALID_ARGUMENT;
  *out = c->tile_h();
  return BBR_OK;
}

int bbr_set_partition(bbr_context *c, int32_t rank, int32_t world, int32_t band_rows) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (world < 1 || rank < 0 || rank >= world) return fail(c, BBR_ERR_INVALID_ARGUMENT, "set_partition: bad rank/world");
  if (band_rows <= 0) band_rows = c->tile_h();
  if (band_rows % c->tile_h()) return fail(c, BBR_ERR_INVALID_ARGUMENT, "set_partition: band_rows must be a multiple of the tile height");
  int rc = drain(c);
  if (rc) return rc;
  c->rank = rank;
  c->world = world;
  c->band_rows = band_rows;
  c->exchange_slot = -1;  // (block sizes change with the partition)
  c->exchange_whole = nullptr;
  if (c->ext_out) {
    uint64_t need = (uint64_t)c->width * (world > 1 ? c->shard_rows() : c->height) * 16;
    if (c->ext_out_bytes < need) {
      c->ext_out = nullptr;
      c->ext_out_bytes = 0;
    }
  }
  return BBR_OK;
}

// onWindowResize (src/main.cpp:1042-1061): wait for the device, drop everything whose size follows the swap-chain
// extent, keep meshes, materials and options.  Buffers come back lazily with the next frame.
int bbr_resize(bbr_context *c, int32_t width, int32_t height) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (width <= 0 || height <= 0 || width > 32768 || height > 32768)
    return fail(c, BBR_ERR_INVALID_ARGUMENT, "resize: width/height out of range");
  if (c->in_frame) return fail(c, BBR_ERR_INVALID_ARGUMENT, "resize: between begin_frame and end_frame");
  int rc = drain(c);
  if (rc) return rc;
  if (width == c->width && height == c->height) return BBR_OK;
  c->width = width;
  c->height = height;
  auto drop = [](FrameSlot &s) {
    s.release_tile_buffers();
    s.d_frame.release(); s.d_present.release(); s.d_depth.release();
    s.has_depth = false;
    s.fused = false;
    s.present.active = false;
    s.present.out = nullptr;
    s.present.copy_to = nullptr;
    s.out_used = nullptr;
    s.in_flight = false;
    if (s.h_flags) s.h_flags[0] = s.h_flags[1] = s.h_flags[2] = 0u;
  };
  for (FrameSlot &s : c->slots) drop(s);
  drop(c->ov);
  c->d_vis_prim.release();
  c->d_vis_depth.release();
  c->d_gbuffer.release();
  c->exchange_slot = -1;  // (the whole frame of the last exchange had the old extent)
  c->exchange_whole = nullptr;
  c->have_frame = false;
  c->last_slot = -1;
  // a caller-owned output buffer was sized for the old extent: the caller sets it again (bbr_set_output_device_ptr)
  c->ext_out = nullptr;
  c->ext_out_bytes = 0;
  return BBR_OK;
}

int bbr_stream_layout_state(const bbr_context *c, int32_t *out_layout, int32_t *out_decided, float *out_ms) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  if (out_layout) *out_layout = c->pipelined() ? c->layout_mode : 0;
  if (out_decided) *out_decided = 1;  // (the layout is a plain option: nothing is timed or decided at run time)
  if (out_ms)
    for (int l = 0; l < bbr_context::kLayouts; ++l) out_ms[l] = 0.f;
  return BBR_OK;
}

int bbr_shard_rows(const bbr_context *c, int32_t *out_rows) {
  if (!c || !out_rows) return BBR_ERR_INVALID_ARGUMENT;
  *out_rows = c->shard_rows();
  return BBR_OK;
}

int bbr_unpack_gathered(bbr_context *c, const void *gathered, void *frame, void *stream) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!gathered || !frame) return fail(c, BBR_ERR_INVALID_ARGUMENT, "unpack_gathered: NULL");
  size_t n = (size_t)c->width * c->height;
  hipStream_t st = stream ? (hipStream_t)stream : c->shade_stream();
  hipLaunchKernelGGL(k_unpack_gathered, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float4 *)gathered,
                     (float4 *)frame, c->width, c->height, c->world, c->eff_band_rows(), c->shard_rows());
  HIP_TRY(c, hipGetLastError());
  return BBR_OK;
}

int bbr_get_stats(bbr_context *c, bbr_stats *out) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!out) return fail(c, BBR_ERR_INVALID_ARGUMENT, "get_stats: NULL");
  if (!c->have_frame) return fail(c, BBR_ERR_NOT_IN_FRAME, "get_stats: nothing rendered");
  Counters h;
  int rc = sync_and_fix(c, &h);
  if (rc) return rc;
  const FrameSlot &s = c->slots[c->last_slot];
  std::memset(out, 0, sizeof *out);
  out->n_prims = c->n_prims;
  {
    size_t nb = ((c->n_prims + 255) / 256) * 4;  // (one record per wave of k_geometry)
    std::vector<BlockStats> bs(nb);
    if (nb) HIP_TRY(c, hipMemcpy(bs.data(), s.d_block_stats.ptr, nb * sizeof(BlockStats), hipMemcpyDeviceToHost));
    for (const BlockStats &b : bs) {
      out->n_raster_tris += b.raster_tris;
      out->n_clipped_prims += b.clipped_prims;
      out->n_bin_refs += b.bin_refs;
    }
  }
  {
    // N_shaded = sum of the per-tile fragment counts of the tiles this rank owns
    size_t tiles = (size_t)c->tiles_x() * c->tiles_y();
    std::vector<uint32_t> fc(tiles);
    HIP_TRY(c, hipMemcpy(fc.data(), s.d_frag_count.ptr, tiles * sizeof(uint32_t), hipMemcpyDeviceToHost));
    int band_tiles = c->eff_band_rows() / c->tile_h();
    for (int ty = 0; ty < c->tiles_y(); ++ty) {
      if (c->world > 1 && ((ty / band_tiles) % c->world) != c->rank) continue;
      for (int tx = 0; tx < c->tiles_x(); ++tx) out->n_shaded += fc[(size_t)ty * c->tiles_x() + tx] & ~kFullTile;
    }
  }
  out->n_broad_tris = h.n_broad;
  out->bin_overflow = (uint32_t)c->retries;
  out->tile_w = (uint32_t)c->tile_w();
  out->tile_h = (uint32_t)c->tile_h();
  out->n_tiles = (uint32_t)(c->tiles_x() * c->tiles_y());
  return BBR_OK;
}

int bbr_read_visibility(bbr_context *c, uint32_t *prim_host, float *depth_host) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!c->have_frame) return fail(c, BBR_ERR_NOT_IN_FRAME, "read_visibility: nothing rendered");
  if (c->world > 1) return fail(c, BBR_ERR_INVALID_ARGUMENT, "read_visibility: not available on a partitioned context");
  int rc = sync_and_fix(c, nullptr);
  if (rc) return rc;
  c->dump_vis = true;
  rc = resubmit_last_frame(c);
  if (!rc) rc = sync_and_fix(c, nullptr);
  c->dump_vis = false;
  if (rc) return rc;
  size_t n = (size_t)c->width * c->height;
  if (prim_host) HIP_TRY(c, hipMemcpy(prim_host, c->d_vis_prim.ptr, n * 4, hipMemcpyDeviceToHost));
  if (depth_host) HIP_TRY(c, hipMemcpy(depth_host, c->d_vis_depth.ptr, n * 4, hipMemcpyDeviceToHost));
  return BBR_OK;
}

int bbr_read_gbuffer(bbr_context *c, float *host) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!host) return fail(c, BBR_ERR_INVALID_ARGUMENT, "read_gbuffer: NULL");
  if (!c->have_frame || c->last_slot < 0) return fail(c, BBR_ERR_NOT_IN_FRAME, "read_gbuffer: nothing rendered");
  if (!c->deferred) return fail(c, BBR_ERR_INVALID_ARGUMENT, "read_gbuffer: option render_pass is not 1 (deferred)");
  if (c->world > 1) return fail(c, BBR_ERR_INVALID_ARGUMENT, "read_gbuffer: not available with a partition");
  int rc = sync_and_fix(c, nullptr);
  if (rc) return rc;
  // the fused deferred kernel keeps the G-buffer in registers; render the frame once more with the dump enabled
  c->dump_gbuffer = true;
  rc = resubmit_last_frame(c);
  if (rc == BBR_OK) rc = sync_and_fix(c, nullptr);
  c->dump_gbuffer = false;
  if (rc) return rc;
  const size_t n = (size_t)c->width * c->height * 16;
  std::vector<_Float16> h(n);
  HIP_TRY(c, hipMemcpy(h.data(), c->d_gbuffer.ptr, n * 2, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) host[i] = (float)h[i];
  c->d_gbuffer.release();
  return BBR_OK;
}

int bbr_last_frame_time_ms(bbr_context *c, float *out_frame_ms, float *out_shade_ms) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!c->timing) return fail(c, BBR_ERR_INVALID_ARGUMENT, "last_frame_time_ms: enable option \"timing\" first");
  if (!c->ring_frames) return fail(c, BBR_ERR_NOT_IN_FRAME, "last_frame_time_ms: no timed frame yet");
  int rc = drain(c);
  if (rc) return rc;
  const hipEvent_t *e = &c->ring[bbr_context::kRingEvents * ((c->ring_frames - 1) % bbr_context::kRingCap)];
  float a = 0.f, b = 0.f;
  if (c->timing == 1) HIP_TRY(c, hipEventElapsedTime(&a, e[0], e[4]));
  HIP_TRY(c, hipEventElapsedTime(&b, e[3], e[4]));
  if (out_frame_ms) *out_frame_ms = a;
  if (out_shade_ms) *out_shade_ms = b;
  return BBR_OK;
}

int bbr_host_timing(const bbr_context *c, uint64_t *out_frames, uint64_t *out_submit_ns, uint64_t *out_blocked_ns,
                    uint64_t *out_blocked_frames) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  if (out_frames) *out_frames = c->host_frames;
  if (out_submit_ns) *out_submit_ns = c->host_submit_ns;
  if (out_blocked_ns) *out_blocked_ns = c->host_blocked_ns;
  if (out_blocked_frames) *out_blocked_frames = c->host_blocked_frames;
  return BBR_OK;
}

int bbr_host_timing_reset(bbr_context *c) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  c->host_frames = c->host_submit_ns = c->host_blocked_ns = c->host_blocked_frames = 0;
  return BBR_OK;
}

int bbr_capacity_growths(const bbr_context *c, uint32_t *out_count) {
  if (!c || !out_count) return BBR_ERR_INVALID_ARGUMENT;
  *out_count = (uint32_t)c->retries;
  return BBR_OK;
}

int bbr_timing_reset(bbr_context *c) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  int rc = drain(c);
  if (rc) return rc;
  c->ring_frames = 0;
  c->present_launches = 0;
  return BBR_OK;
}

int bbr_timing_summary(bbr_context *c, uint32_t *out_frames, float *out_avg_frame_ms, float *out_avg_geometry_ms,
                       float *out_avg_raster_ms, float *out_avg_shade_ms) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!c->timing) return fail(c, BBR_ERR_INVALID_ARGUMENT, "timing_summary: enable option \"timing\" first");
  int rc = drain(c);
  if (rc) return rc;
  uint32_t n = std::min(c->ring_frames, bbr_context::kRingCap);
  double f = 0, g = 0, r = 0, t = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const hipEvent_t *e = &c->ring[bbr_context::kRingEvents * i];
    float a = 0.f, b = 0.f, d = 0.f, h = 0.f;
    if (c->timing == 1) {
      HIP_TRY(c, hipEventElapsedTime(&a, e[0], e[4]));
      HIP_TRY(c, hipEventElapsedTime(&b, e[0], e[1]));
      HIP_TRY(c, hipEventElapsedTime(&d, e[1], e[2]));
    }
    HIP_TRY(c, hipEventElapsedTime(&h, e[3], e[4]));
    f += a; g += b; r += d; t += h;
  }
  if (out_frames) *out_frames = n;
  if (out_avg_frame_ms) *out_avg_frame_ms = n ? (float)(f / n) : 0.f;
  if (out_avg_geometry_ms) *out_avg_geometry_ms = n ? (float)(g / n) : 0.f;
  if (out_avg_raster_ms) *out_avg_raster_ms = n ? (float)(r / n) : 0.f;
  if (out_avg_shade_ms) *out_avg_shade_ms = n ? (float)(t / n) : 0.f;
  return BBR_OK;
}

int bbr_present_timing(bbr_context *c, uint32_t *out_launches, float *out_avg_ms) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!c->timing) return fail(c, BBR_ERR_INVALID_ARGUMENT, "present_timing: enable option \"timing\" first");
  int rc = drain(c);
  if (rc) return rc;
  uint32_t n = std::min(c->present_launches, bbr_context::kRingCap);
  double t = 0;
  for (uint32_t i = 0; i < n; ++i) {
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->present_ring[2 * i], c->present_ring[2 * i + 1]));
    t += ms;
  }
  if (out_launches) *out_launches = n;
  if (out_avg_ms) *out_avg_ms = n ? (float)(t / n) : 0.f;
  return BBR_OK;
}

int bbr_set_option(bbr_context *c, const char *name, int64_t value) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!name) return fail(c, BBR_ERR_INVALID_ARGUMENT, "set_option: NULL name");
  std::string n(name);
  if (n == "gbuffer_view") {  // like render_pass: takes effect with the next submitted frame
    if (value < -1 || value > 3) return fail(c, BBR_ERR_INVALID_ARGUMENT, "gbuffer_view: -1 (the lit scene) or 0..3 (position, normal, albedo, MRAH)");
    c->gbuffer_view = (int)value;
    return BBR_OK;
  }
  if (n == "render_pass") {  // takes effect with the next submitted frame; no need to wait for the ones in flight
    if (value != 0 && value != 1) return fail(c, BBR_ERR_INVALID_ARGUMENT, "render_pass: 0 (forward) or 1 (deferred)");
    c->deferred = value == 1;
    return BBR_OK;
  }
  int rc = drain(c);
  if (rc) return rc;
  if (n == "timing") {
    if (value < 0 || value > 2) return fail(c, BBR_ERR_INVALID_ARGUMENT, "timing: 0, 1 or 2");
    c->timing = (int)value;
    if (value && c->ring.empty()) {  // here, not in the first timed frame: 2560 hipEventCreate calls take half a millisecond
      c->ring.resize(bbr_context::kRingEvents * bbr_context::kRingCap);
      for (auto &e : c->ring) HIP_TRY(c, hipEventCreate(&e));
    }
    c->ring_frames = 0;
    c->present_launches = 0;
    c->timing_tick = 0;
  } else if (n == "timing_stride") {
    if (value < 1 || value > 1024) return fail(c, BBR_ERR_INVALID_ARGUMENT, "timing_stride: 1 .. 1024");
    c->timing_stride = (int)value;
    c->timing_tick = 0;
  }
  else if (n == "frames_in_flight") {
    if (value < 1 || value > bbr_context::kMaxSlots) return fail(c, BBR_ERR_INVALID_ARGUMENT, "frames_in_flight: 1 .. 4");
    c->frames_in_flight = (int)value;
    c->frame_counter = 0;
  } else if (n == "tile_mode") {
    if (value != 0 && value != 1) return fail(c, BBR_ERR_INVALID_ARGUMENT, "tile_mode: 0 (64x64) or 1 (32x32)");
    if (c->world > 1 && c->band_rows % (value == 0 ? 64 : 32))
      return fail(c, BBR_ERR_INVALID_ARGUMENT, "tile_mode: band_rows not a multiple of the new tile height");
    c->tile_mode = (int)value;
    // bins and fragment lists are laid out per tile: drop them so that ensure() re-zeroes the counters
    for (FrameSlot &s : c->slots) s.release_tile_buffers();
    c->ov.release_tile_buffers();
  } else if (n == "bin_cap") {
    if (value < 1 || value > (1 << 20)) return fail(c, BBR_ERR_INVALID_ARGUMENT, "bin_cap out of range");
    c->bin_cap = (uint32_t)value;
    for (FrameSlot &s : c->slots) s.d_bins.release();
  } else if (n == "ablate") {
#ifdef BB_ABLATE
    c->ablate = (uint32_t)value;
#else
    if (value != 0) return fail(c, BBR_ERR_INVALID_ARGUMENT, "ablate: diagnostic builds only (make EXTRA=-DBB_ABLATE)");
#endif
  } else if (n == "present_fused") {
    c->present_fused = value != 0;
  } else if (n == "overlays") {
    c->overlays = value != 0;
  } else if (n == "stream_layout") {
    if (value < 0 || value >= bbr_context::kLayouts) return fail(c, BBR_ERR_INVALID_ARGUMENT, "stream_layout: 0, 1 or 2");
    c->layout_mode = (int)value;
  } else if (n == "push_mode") {
    if (value != 0 && value != 1) return fail(c, BBR_ERR_INVALID_ARGUMENT, "push_mode: 0 (copies) or 1 (one kernel, all peers)");
    c->push_mode = (int)value;
  } else if (n == "no_tail_items") {
    if (value < 0) return fail(c, BBR_ERR_INVALID_ARGUMENT, "no_tail_items must be >= 0");
    c->no_tail_items = value;
  } else if (n == "heavy_tiles") {
    if (value < -1 || value > (1 << 24)) return fail(c, BBR_ERR_INVALID_ARGUMENT, "heavy_tiles: -1 (automatic), 0 (off) or a reference count");
    c->heavy_tiles = value;
  } else if (n == "broad_cap" || n == "clip_cap") {
    // starting capacity of the every-tile list / the clip arena (entries); both double when a frame overflows them
    if (value < 1 || value > (1 << 24)) return fail(c, BBR_ERR_INVALID_ARGUMENT, n + " out of range (1 .. 2^24)");
    int rc = drain(c);
    if (rc) return rc;
    (n == "broad_cap" ? c->broad_cap : c->clip_cap) = (uint32_t)value;
    for (FrameSlot &s : c->slots) {
      if (n == "broad_cap") s.d_broad.release();
      else s.d_clip.release();
    }
    if (n == "broad_cap") c->ov.d_broad.release();
    else c->ov.d_clip.release();
  } else if (n == "broad_threshold") {
    if (value < 1) return fail(c, BBR_ERR_INVALID_ARGUMENT, "broad_threshold must be >= 1");
    c->broad_threshold = (uint32_t)value;
  } else {
    return fail(c, BBR_ERR_INVALID_ARGUMENT, "unknown option: " + n);
  }
  return BBR_OK;
}

#ifdef BB_STAMPS
int bbr_debug_raster_stamps(bbr_context *c, unsigned long long *out) {
  if (!c || c->last_slot < 0) return BBR_ERR_INVALID_ARGUMENT;
  int rc = drain(c);
  if (rc) return rc;
  size_t tiles = (size_t)c->tiles_x() * c->tiles_y();
  HIP_TRY(c, hipMemcpy(out, c->slots[c->last_slot].d_frag_count.ptr + tiles, tiles * 64, hipMemcpyDeviceToHost));
  return BBR_OK;
}
// diagnostic build only: copy out the k_geometry time stamps of the last frame (8 x u64 per workgroup)
int bbr_debug_stamps(bbr_context *c, unsigned long long *out, uint32_t n_blocks) {
  if (!c || c->last_slot < 0) return BBR_ERR_INVALID_ARGUMENT;
  int rc = drain(c);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpy(out, c->slots[c->last_slot].d_clip.ptr + c->clip_cap, (size_t)n_blocks * 64, hipMemcpyDeviceToHost));
  return BBR_OK;
}
#endif

int bbr_present(bbr_context *c, void *rgba8_device, int32_t hdr16) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!c->have_frame || c->last_slot < 0) return fail(c, BBR_ERR_NOT_IN_FRAME, "present: nothing rendered");
  FrameSlot &s = c->slots[c->last_slot];
  const size_t n = (size_t)c->width * (c->world > 1 ? c->shard_rows() : c->height);
  if (c->present_fused && s.present.active && s.present.out == s.d_present.ptr) {
    // the frame was rendered as presented pixels already (binary16 stage included); a caller buffer gets a copy
    if (!hdr16) return fail(c, BBR_ERR_INVALID_ARGUMENT, "present: option present_fused always applies the binary16 stage");
    s.present.copy_to = rgba8_device;
    if (rgba8_device) {
      HIP_TRY(c, hipStreamWaitEvent(c->present_stream(), s.ev_shade_done, 0));
      HIP_TRY(c, hipMemcpyAsync(rgba8_device, s.d_present.ptr, n * 4, hipMemcpyDeviceToDevice, c->present_stream()));
      HIP_TRY(c, hipEventRecord(s.ev_shade_done, c->present_stream()));
    }
    return BBR_OK;
  }
  if (!rgba8_device) HIP_TRY(c, s.d_present.ensure(n));
  s.present.active = true;
  s.present.out = rgba8_device ? (uint32_t *)rgba8_device : s.d_present.ptr;
  s.present.enable = s.tone_enable;
  s.present.exposure = s.tone_exposure;
  s.present.hdr16 = hdr16 != 0;
  return queue_present(c, s);
}

int bbr_present_buffer(bbr_context *c, const void *rgba32f_device, void *rgba8_device, uint64_t n_pixels, int32_t enable,
                       float exposure, int32_t hdr16, void *stream) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!rgba32f_device || !rgba8_device) return fail(c, BBR_ERR_INVALID_ARGUMENT, "present_buffer: NULL");
  if (!n_pixels) return BBR_OK;
  int rc = ensure_srgb_tables(c);
  if (rc) return rc;
  const size_t per_block = (size_t)kPresentThreads * kPresentPerThread;
  hipLaunchKernelGGL(k_present, dim3((unsigned)((n_pixels + per_block - 1) / per_block)), dim3(kPresentThreads), 0,
                     stream ? (hipStream_t)stream : c->shade_stream(), (const float4 *)rgba32f_device,
                     (uint32_t *)rgba8_device, (size_t)n_pixels, c->d_srgb_tables.ptr, enable, exposure, hdr16 != 0);
  HIP_TRY(c, hipGetLastError());
  return BBR_OK;
}

int bbr_read_presented(bbr_context *c, uint8_t *host) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!host) return fail(c, BBR_ERR_INVALID_ARGUMENT, "read_presented: NULL");
  if (!c->have_frame || c->last_slot < 0 || !c->slots[c->last_slot].present.active)
    return fail(c, BBR_ERR_NOT_IN_FRAME, "read_presented: bbr_present was not called for the last frame");
  int rc = sync_and_fix(c, nullptr);
  if (rc) return rc;
  const size_t n = (size_t)c->width * (c->world > 1 ? c->shard_rows() : c->height);
  HIP_TRY(c, hipMemcpy(host, c->slots[c->last_slot].present.out, n * 4, hipMemcpyDeviceToHost));
  return BBR_OK;
}

int bbr_presented_device_ptr(bbr_context *c, void **out_ptr, uint64_t *out_bytes) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!out_ptr) return fail(c, BBR_ERR_INVALID_ARGUMENT, "presented_device_ptr: NULL");
  if (!c->have_frame || c->last_slot < 0 || !c->slots[c->last_slot].present.active)
    return fail(c, BBR_ERR_NOT_IN_FRAME, "presented_device_ptr: bbr_present was not called for the last frame");
  *out_ptr = c->slots[c->last_slot].present.out;
  if (out_bytes) *out_bytes = (uint64_t)c->width * (c->world > 1 ? c->shard_rows() : c->height) * 4;
  return BBR_OK;
}

int bbr_unpack_gathered_rgba8(bbr_context *c, const void *gathered, void *frame, void *stream) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!gathered || !frame) return fail(c, BBR_ERR_INVALID_ARGUMENT, "unpack_gathered_rgba8: NULL");
  size_t n = (size_t)c->width * c->height;
  hipStream_t st = stream ? (hipStream_t)stream : c->shade_stream();
  hipLaunchKernelGGL(k_unpack_gathered_rgba8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     (const uint32_t *)gathered, (uint32_t *)frame, c->width, c->height, c->world, c->eff_band_rows(),
                     c->shard_rows());
  HIP_TRY(c, hipGetLastError());
  return BBR_OK;
}

namespace {
// packed shard: rgb[n][3] float, padding to 8 bytes, one 64-bit alpha mask per 64 pixels, padding to 16 bytes
size_t packed_mask_offset(const bbr_context *c) { return (((size_t)c->width * c->shard_rows() * 12) + 7) & ~(size_t)7; }
size_t packed_block_bytes(const bbr_context *c) {
  const size_t n = (size_t)c->width * c->shard_rows();
  return (packed_mask_offset(c) + ((n + 63) / 64) * 8 + 15) & ~(size_t)15;
}
}  // namespace

int bbr_packed_shard_bytes(const bbr_context *c, uint64_t *out_bytes) {
  if (!c || !out_bytes) return BBR_ERR_INVALID_ARGUMENT;
  *out_bytes = packed_block_bytes(c);
  return BBR_OK;
}

int bbr_pack_shard(bbr_context *c, void *packed, void *stream) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!packed) return fail(c, BBR_ERR_INVALID_ARGUMENT, "pack_shard: NULL");
  if (!c->have_frame || c->last_slot < 0) return fail(c, BBR_ERR_NOT_IN_FRAME, "pack_shard: nothing rendered");
  const FrameSlot &s = c->slots[c->last_slot];
  if (s.fused) return fail(c, BBR_ERR_INVALID_ARGUMENT, "pack_shard: no fp32 frame with option present_fused");
  const size_t n = (size_t)c->width * c->shard_rows();
  hipStream_t st = stream ? (hipStream_t)stream : s.stream_used;  // a caller's stream must already wait for the frame
  hipLaunchKernelGGL(k_pack_shard, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float4 *)s.out_used,
                     (float *)packed, (unsigned long long *)((uint8_t *)packed + packed_mask_offset(c)), n);
  HIP_TRY(c, hipGetLastError());
  return BBR_OK;
}

int bbr_unpack_gathered_packed(bbr_context *c, const void *gathered, void *frame, void *stream) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!gathered || !frame) return fail(c, BBR_ERR_INVALID_ARGUMENT, "unpack_gathered_packed: NULL");
  size_t n = (size_t)c->width * c->height;
  hipStream_t st = stream ? (hipStream_t)stream : c->shade_stream();
  hipLaunchKernelGGL(k_unpack_gathered_packed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint8_t *)gathered,
                     (float4 *)frame, c->width, c->height, c->world, c->eff_band_rows(), c->shard_rows(),
                     packed_block_bytes(c), packed_mask_offset(c));
  HIP_TRY(c, hipGetLastError());
  return BBR_OK;
}

// ================================================================================================
// native exchange (BASELINE config #4): the step in which every rank gets the whole frame
//   collective form: RCCL all-gather of the ranks' blocks (ring over xGMI), one process per GPU
//   peer form:       every rank puts its block into every rank's gather buffer -- ONE kernel that stores to all peers at once
//                    (option push_mode 1, the default: every xGMI link of the mesh busy together), or hipMemcpyPeerAsync
//                    copies one behind the other (push_mode 0, and the fallback when a peer cannot be mapped); for one
//                    process that drives several GPUs, or several processes that exchanged IPC handles
// Both run on the stream of the frame's slot, right behind its k_shade: with stream layout 2 a frame's kernels share a
// stream with nothing else, so the exchange is simply the frame's last step, the next frame of the same slot is ordered
// behind it without an event, and the frames of the other slots render while the links are busy.
// ================================================================================================
namespace {

size_t exchange_block_bytes(const bbr_context *c, int form) {
  const size_t n = (size_t)c->width * c->shard_rows();
  return form == BBR_SHARD_RGBA32F ? n * 16 : (form == BBR_SHARD_PACKED ? packed_block_bytes(c) : (form == BBR_SHARD_RGBA16F ? n * 8 : n * 4));
}
// bytes per pixel of the whole frame an exchange of this form leaves behind (the binary16 form is widened to fp32)
size_t whole_pixel_bytes(int form) { return form == BBR_SHARD_RGBA8 ? 4 : 16; }

int open_rccl(bbr_context *c) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.lib) return BBR_OK;
  void *lib = nullptr;
  for (const char *name : {"librccl.so.1", "librccl.so"}) {
    lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (lib) break;
  }
  if (!lib) return fail(c, BBR_ERR_HIP, std::string("RCCL not found (librccl.so.1): ") + dlerror());
  Rccl r;
  r.get_unique_id = (decltype(r.get_unique_id))dlsym(lib, "ncclGetUniqueId");
  r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(lib, "ncclCommInitRank");
  r.all_gather = (decltype(r.all_gather))dlsym(lib, "ncclAllGather");
  r.comm_destroy = (decltype(r.comm_destroy))dlsym(lib, "ncclCommDestroy");
  r.comm_count = (decltype(r.comm_count))dlsym(lib, "ncclCommCount");
  r.error_string = (decltype(r.error_string))dlsym(lib, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy || !r.comm_count || !r.error_string)
    return fail(c, BBR_ERR_HIP, "librccl lacks an entry point");
  r.lib = lib;
  g_rccl = r;
  return BBR_OK;
}

#define RCCL_TRY(ctx, expr)                                                                                    \
  do {                                                                                                         \
    ncclResult_t _r = (expr);                                                                                  \
    if (_r != ncclSuccess) return fail(ctx, BBR_ERR_HIP, std::string(#expr) + ": " + g_rccl.error_string(_r)); \
  } while (0)

// Put this rank's block of the last frame where the exchange reads it: `dst` (the rank's slot of a gather buffer, or a
// peer's).  rgba32f / rgba8 blocks that already live there (the frame was rendered into the slot) cost nothing.
int stage_block(bbr_context *c, FrameSlot &s, int form, void *dst, hipStream_t st) {
  const size_t n = (size_t)c->width * c->shard_rows();
  if (form == BBR_SHARD_PACKED) {
    hipLaunchKernelGGL(k_pack_shard, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float4 *)s.out_used, (float *)dst,
                       (unsigned long long *)((uint8_t *)dst + packed_mask_offset(c)), n);
    HIP_TRY(c, hipGetLastError());
    return BBR_OK;
  }
  if (form == BBR_SHARD_RGBA16F) {
    hipLaunchKernelGGL(k_pack_shard_half, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float4 *)s.out_used, (uint2 *)dst, n);
    HIP_TRY(c, hipGetLastError());
    return BBR_OK;
  }
  const void *src = form == BBR_SHARD_RGBA32F ? (const void *)s.out_used : (const void *)s.present.out;
  if (src != dst) HIP_TRY(c, hipMemcpyAsync(dst, src, exchange_block_bytes(c, form), hipMemcpyDeviceToDevice, st));
  return BBR_OK;
}

int check_exchange(bbr_context *c, int form, const char *who) {
  if (form != BBR_SHARD_RGBA32F && form != BBR_SHARD_PACKED && form != BBR_SHARD_RGBA8 && form != BBR_SHARD_RGBA16F)
    return fail(c, BBR_ERR_INVALID_ARGUMENT, std::string(who) + ": form must be BBR_SHARD_RGBA32F, _PACKED, _RGBA8 or _RGBA16F");
  if (!c->have_frame || c->last_slot < 0) return fail(c, BBR_ERR_NOT_IN_FRAME, std::string(who) + ": nothing rendered");
  const FrameSlot &s = c->slots[c->last_slot];
  if (form == BBR_SHARD_RGBA8) {
    if (!s.present.active || !s.present.out) return fail(c, BBR_ERR_NOT_IN_FRAME, std::string(who) + ": BBR_SHARD_RGBA8 needs bbr_present first");
  } else if (s.fused) {
    return fail(c, BBR_ERR_INVALID_ARGUMENT, std::string(who) + ": no fp32 frame with option present_fused (use BBR_SHARD_RGBA8)");
  }
  return BBR_OK;
}

int unpack_whole(bbr_context *c, int form, const void *gathered, void *whole, hipStream_t st) {
  const size_t n = (size_t)c->width * c->height;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (form == BBR_SHARD_RGBA32F)
    hipLaunchKernelGGL(k_unpack_gathered, grid, block, 0, st, (const float4 *)gathered, (float4 *)whole, c->width, c->height, c->world,
                       c->eff_band_rows(), c->shard_rows());
  else if (form == BBR_SHARD_PACKED)
    hipLaunchKernelGGL(k_unpack_gathered_packed, grid, block, 0, st, (const uint8_t *)gathered, (float4 *)whole, c->width, c->height,
                       c->world, c->eff_band_rows(), c->shard_rows(), packed_block_bytes(c), packed_mask_offset(c));
  else if (form == BBR_SHARD_RGBA16F)
    hipLaunchKernelGGL(k_unpack_gathered_half, grid, block, 0, st, (const uint2 *)gathered, (float4 *)whole, c->width, c->height,
                       c->world, c->eff_band_rows(), c->shard_rows());
  else
    hipLaunchKernelGGL(k_unpack_gathered_rgba8, grid, block, 0, st, (const uint32_t *)gathered, (uint32_t *)whole, c->width, c->height,
                       c->world, c->eff_band_rows(), c->shard_rows());
  HIP_TRY(c, hipGetLastError());
  return BBR_OK;
}

}  // namespace

int bbr_comm_unique_id(bbr_context *c, uint8_t *out_id) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!out_id) return fail(c, BBR_ERR_INVALID_ARGUMENT, "comm_unique_id: NULL");
  int rc = open_rccl(c);
  if (rc) return rc;
  static_assert(BBR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "BBR_COMM_ID_BYTES");
  ncclUniqueId id;
  RCCL_TRY(c, g_rccl.get_unique_id(&id));
  memcpy(out_id, id.internal, BBR_COMM_ID_BYTES);
  return BBR_OK;
}

int bbr_comm_init(bbr_context *c, int32_t rank, int32_t world, const uint8_t *unique_id) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!unique_id || world < 1 || rank < 0 || rank >= world) return fail(c, BBR_ERR_INVALID_ARGUMENT, "comm_init: bad rank / world / id");
  if (c->comm) return fail(c, BBR_ERR_INVALID_ARGUMENT, "comm_init: the context already has a communicator (bbr_comm_destroy first)");
  int rc = open_rccl(c);
  if (rc) return rc;
  ncclUniqueId id;
  memcpy(id.internal, unique_id, BBR_COMM_ID_BYTES);
  RCCL_TRY(c, g_rccl.comm_init_rank(&c->comm, world, id, rank));  // collective: returns once every rank has called it
  c->comm_rank = rank;
  c->comm_world = world;
  return BBR_OK;
}

int bbr_comm_probe(bbr_context *c) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  return open_rccl(c);  // dlopen + dlsym only: nothing collective, safe to call on one rank alone
}

int bbr_comm_count(bbr_context *c, int32_t *out_ranks) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  if (!out_ranks) return fail(c, BBR_ERR_INVALID_ARGUMENT, "comm_count: NULL");
  if (!c->comm) return fail(c, BBR_ERR_NOT_IN_FRAME, "comm_count: no communicator (bbr_comm_init)");
  int n = 0;
  RCCL_TRY(c, g_rccl.comm_count(c->comm, &n));
  *out_ranks = n;
  return BBR_OK;
}

int bbr_stage_shard(bbr_context *c, int32_t form, void *block_device, void *stream) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  int rc = check_exchange(c, form, "stage_shard");
  if (rc) return rc;
  if (!block_device) return fail(c, BBR_ERR_INVALID_ARGUMENT, "stage_shard: NULL");
  FrameSlot &s = c->slots[c->last_slot];
  hipStream_t st = stream ? (hipStream_t)stream : s.stream_used;  // a caller's stream must already wait for the frame
  return stage_block(c, s, form, block_device, st);
}

int bbr_comm_destroy(bbr_context *c) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!c->comm) return BBR_OK;
  int rc = drain(c);
  if (rc) return rc;
  ncclComm_t comm = c->comm;
  c->comm = nullptr;
  c->comm_rank = -1;
  c->comm_world = 0;
  RCCL_TRY(c, g_rccl.comm_destroy(comm));
  return BBR_OK;
}

int bbr_exchange_block_bytes(const bbr_context *c, int32_t form, uint64_t *out_bytes) {
  if (!c || !out_bytes || form < 0 || form > BBR_SHARD_RGBA16F) return BBR_ERR_INVALID_ARGUMENT;
  *out_bytes = exchange_block_bytes(c, form);
  return BBR_OK;
}

int bbr_allgather_frame(bbr_context *c, int32_t form, void *gathered, void *whole, void *stream) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  int rc = check_exchange(c, form, "allgather_frame");
  if (rc) return rc;
  if (!c->comm) return fail(c, BBR_ERR_NOT_IN_FRAME, "allgather_frame: no communicator (bbr_comm_init)");
  if (c->comm_rank != c->rank || c->comm_world != c->world)
    return fail(c, BBR_ERR_INVALID_ARGUMENT, "allgather_frame: the communicator's rank / world differ from the partition's (bbr_set_partition)");
  FrameSlot &s = c->slots[c->last_slot];
  const size_t block = exchange_block_bytes(c, form);
  if (!gathered) {
    HIP_TRY(c, s.d_gathered.ensure(block * (size_t)c->world));
    gathered = s.d_gathered.ptr;
  }
  if (!whole) {
    HIP_TRY(c, s.d_whole.ensure((size_t)c->width * c->height * whole_pixel_bytes(form)));
    whole = s.d_whole.ptr;
  }
  hipStream_t st = stream ? (hipStream_t)stream : s.stream_used;
  // (also on the slot's own stream: a separate presentation pass may have run on another one and moved the event there)
  HIP_TRY(c, hipStreamWaitEvent(st, s.ev_shade_done, 0));
  uint8_t *mine = (uint8_t *)gathered + block * (size_t)c->rank;
  rc = stage_block(c, s, form, mine, st);
  if (rc) return rc;
  // in place: the send buffer is this rank's slot of the receive buffer
  RCCL_TRY(c, g_rccl.all_gather(mine, gathered, block, ncclUint8, c->comm, st));
  rc = unpack_whole(c, form, gathered, whole, st);
  if (rc) return rc;
  if (!stream) HIP_TRY(c, hipEventRecord(s.ev_shade_done, st));  // "the frame is done" now includes its exchange
  c->exchange_slot = c->last_slot;
  c->exchange_form = form;
  c->exchange_whole = whole;
  return BBR_OK;
}

int bbr_push_shard(bbr_context *c, int32_t form, void *const *peer_gathered, const int32_t *peer_devices, void *stream) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  int rc = check_exchange(c, form, "push_shard");
  if (rc) return rc;
  if (!peer_gathered || !peer_devices) return fail(c, BBR_ERR_INVALID_ARGUMENT, "push_shard: NULL");
  FrameSlot &s = c->slots[c->last_slot];
  const size_t block = exchange_block_bytes(c, form);
  hipStream_t st = stream ? (hipStream_t)stream : s.stream_used;
  // (also on the slot's own stream: a separate presentation pass may have run on another one and moved the event there)
  HIP_TRY(c, hipStreamWaitEvent(st, s.ev_shade_done, 0));
  for (int p = 0; p < c->world; ++p)
    if (!peer_gathered[p]) return fail(c, BBR_ERR_INVALID_ARGUMENT, "push_shard: a peer's gather buffer is NULL");
  // the block is made once, in this rank's own gather buffer, and goes from there to the peers
  uint8_t *mine = (uint8_t *)peer_gathered[c->rank] + block * (size_t)c->rank;
  rc = stage_block(c, s, form, mine, st);
  if (rc) return rc;
  // push_mode 1 (default): ONE kernel stores the block into every peer's buffer, all links busy at once.  It needs the
  // peers' memory mapped into this device's address space: the same device, an opened IPC handle, or peer access, which
  // is switched on here the first time a device shows up.  A peer that cannot be mapped: the copies below instead.
  bool direct = c->push_mode == 1 && c->world > 1;
  for (int p = 0; direct && p < c->world; ++p) {
    const int dev = peer_devices[p];
    if (dev == c->device || c->peer_mapped.count(dev)) continue;
    int can = 0;
    hipError_t e = hipDeviceCanAccessPeer(&can, c->device, dev);
    if (e == hipSuccess && can) e = hipDeviceEnablePeerAccess(dev, 0);
    (void)hipGetLastError();  // (neither outcome may linger as the runtime's "last error": the copies below are a full substitute)
    if (!can || (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)) { direct = false; break; }
    c->peer_mapped.insert(dev);
  }
  if (direct) {
    bool wide = block % 16 == 0;   // 16-byte accesses when every address involved allows them, 4-byte ones otherwise
    for (int p = 0; p < c->world; ++p) wide = wide && ((uintptr_t)peer_gathered[p] % 16) == 0;
    const size_t n = wide ? block / 16 : block / 4;   // (every block form is a whole number of 4-byte words)
    // a grid of one workgroup per CU: the copy is bound by the links (world - 1 x ~60 GB/s), not by the CUs it occupies
    const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)c->n_cus, (n + kPushThreads - 1) / kPushThreads));
    for (int k0 = 1; k0 < c->world; k0 += kMaxPushPeers) {
      PushTargets t = {};
      int nt = 0;
      for (int k = k0; k < c->world && nt < kMaxPushPeers; ++k)
        t.dst[nt++] = (uint8_t *)peer_gathered[(c->rank + k) % c->world] + block * (size_t)c->rank;
      if (wide) hipLaunchKernelGGL(k_push_block<uint4>, dim3(grid), dim3(kPushThreads), 0, st, (const uint4 *)mine, t, nt, n);
      else hipLaunchKernelGGL(k_push_block<uint32_t>, dim3(grid), dim3(kPushThreads), 0, st, (const uint32_t *)mine, t, nt, n);
    }
    HIP_TRY(c, hipGetLastError());
  } else {
    // push_mode 0: copies queued one behind the other on the one stream, nearest rank first (rank + 1, rank + 2, ...): at any
    // moment this rank drives ONE of its links -- ring timing on a full mesh
    for (int k = 1; k < c->world; ++k) {
      const int p = (c->rank + k) % c->world;
      HIP_TRY(c, hipMemcpyPeerAsync((uint8_t *)peer_gathered[p] + block * (size_t)c->rank, peer_devices[p], mine, c->device, block, st));
    }
  }
  c->last_push_direct = direct;
  if (!stream) HIP_TRY(c, hipEventRecord(s.ev_shade_done, st));
  return BBR_OK;
}

int bbr_push_state(const bbr_context *c, int32_t *out_direct) {
  if (!c || !out_direct) return BBR_ERR_INVALID_ARGUMENT;
  *out_direct = c->last_push_direct ? 1 : 0;
  return BBR_OK;
}

int bbr_unpack_whole(bbr_context *c, int32_t form, const void *gathered, void *whole, void *stream) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (form < 0 || form > BBR_SHARD_RGBA16F || !gathered) return fail(c, BBR_ERR_INVALID_ARGUMENT, "unpack_whole: bad form / NULL");
  if (c->last_slot < 0) return fail(c, BBR_ERR_NOT_IN_FRAME, "unpack_whole: nothing rendered");
  FrameSlot &s = c->slots[c->last_slot];
  if (!whole) {
    HIP_TRY(c, s.d_whole.ensure((size_t)c->width * c->height * whole_pixel_bytes(form)));
    whole = s.d_whole.ptr;
  }
  int rc = unpack_whole(c, form, gathered, whole, stream ? (hipStream_t)stream : s.stream_used);
  if (rc) return rc;
  c->exchange_slot = c->last_slot;
  c->exchange_form = form;
  c->exchange_whole = whole;
  return BBR_OK;
}

int bbr_whole_frame_device_ptr(bbr_context *c, void **out_ptr, uint64_t *out_bytes) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  if (!out_ptr) return fail(c, BBR_ERR_INVALID_ARGUMENT, "whole_frame_device_ptr: NULL");
  if (c->exchange_slot < 0 || !c->exchange_whole) return fail(c, BBR_ERR_NOT_IN_FRAME, "whole_frame_device_ptr: no exchange yet");
  *out_ptr = c->exchange_whole;
  if (out_bytes) *out_bytes = (uint64_t)c->width * c->height * (c->exchange_form == BBR_SHARD_RGBA8 ? 4 : 16);
  return BBR_OK;
}

int bbr_read_whole_frame(bbr_context *c, void *host) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!host) return fail(c, BBR_ERR_INVALID_ARGUMENT, "read_whole_frame: NULL");
  if (c->exchange_slot < 0 || !c->exchange_whole) return fail(c, BBR_ERR_NOT_IN_FRAME, "read_whole_frame: no exchange yet");
  int rc = drain(c);  // (not sync_and_fix: a re-render on one rank alone would leave the collective unmatched)
  if (rc) return rc;
  HIP_TRY(c, hipMemcpy(host, c->exchange_whole, (size_t)c->width * c->height * (c->exchange_form == BBR_SHARD_RGBA8 ? 4 : 16),
                       hipMemcpyDeviceToHost));
  return BBR_OK;
}

int bbr_device_alloc(bbr_context *c, uint64_t bytes, void **out_ptr) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!out_ptr || !bytes) return fail(c, BBR_ERR_INVALID_ARGUMENT, "device_alloc: NULL / zero bytes");
  *out_ptr = nullptr;
  HIP_TRY(c, hipMalloc(out_ptr, bytes));
  HIP_TRY(c, zero_fill_sync(*out_ptr, bytes));
  return BBR_OK;
}

int bbr_device_free(bbr_context *c, void *ptr) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!ptr) return BBR_OK;
  int rc = drain(c);
  if (rc) return rc;
  HIP_TRY(c, hipFree(ptr));
  return BBR_OK;
}

int bbr_copy_to_host(bbr_context *c, void *host, const void *device_ptr, uint64_t bytes) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!host || !device_ptr) return fail(c, BBR_ERR_INVALID_ARGUMENT, "copy_to_host: NULL");
  int rc = drain(c);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpy(host, device_ptr, bytes, hipMemcpyDeviceToHost));
  return BBR_OK;
}

int bbr_ipc_export(bbr_context *c, void *device_ptr, uint8_t *out_handle) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!device_ptr || !out_handle) return fail(c, BBR_ERR_INVALID_ARGUMENT, "ipc_export: NULL");
  static_assert(BBR_IPC_HANDLE_BYTES == sizeof(hipIpcMemHandle_t), "BBR_IPC_HANDLE_BYTES");
  hipIpcMemHandle_t h;
  HIP_TRY(c, hipIpcGetMemHandle(&h, device_ptr));
  memcpy(out_handle, &h, sizeof h);
  return BBR_OK;
}

int bbr_ipc_open(bbr_context *c, const uint8_t *handle, void **out_ptr) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!handle || !out_ptr) return fail(c, BBR_ERR_INVALID_ARGUMENT, "ipc_open: NULL");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof h);
  HIP_TRY(c, hipIpcOpenMemHandle(out_ptr, h, hipIpcMemLazyEnablePeerAccess));
  return BBR_OK;
}

int bbr_ipc_close(bbr_context *c, void *ptr) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!ptr) return fail(c, BBR_ERR_INVALID_ARGUMENT, "ipc_close: NULL");
  HIP_TRY(c, hipIpcCloseMemHandle(ptr));
  return BBR_OK;
}

int bbr_selftest_rcp(bbr_context *c, uint32_t lo_bits, uint32_t hi_bits, uint64_t *out_mismatches) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!out_mismatches || hi_bits < lo_bits) return fail(c, BBR_ERR_INVALID_ARGUMENT, "selftest_rcp: bad arguments");
  int rc = drain(c);
  if (rc) return rc;
  unsigned long long *d = nullptr, h = 0;
  HIP_TRY(c, hipMalloc(&d, sizeof h));
  HIP_TRY(c, zero_fill_sync(d, sizeof h));
  hipLaunchKernelGGL(k_selftest_rcp, dim3(4096), dim3(256), 0, c->geom_stream(), d, lo_bits, hi_bits);
  hipError_t e = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(c, e);
  *out_mismatches = h;
  return BBR_OK;
}

int bbr_tone_map(bbr_context *c, int32_t enable, float exposure) {
  if (!c) return BBR_ERR_INVALID_ARGUMENT;
  BBR_ON_DEVICE(c);
  if (!c->have_frame || c->last_slot < 0) return fail(c, BBR_ERR_NOT_IN_FRAME, "tone_map: nothing rendered");
  if (c->slots[c->last_slot].fused) return fail(c, BBR_ERR_INVALID_ARGUMENT, "tone_map: no fp32 frame with option present_fused");
  float4 *frame = c->slots[c->last_slot].out_used;
  size_t n = (size_t)c->width * (c->world > 1 ? c->shard_rows() : c->height);
  // same stream as the frame's shade kernel: ordered after it
  hipLaunchKernelGGL(k_tone_map, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->slots[c->last_slot].stream_used, frame, n, enable, exposure);
  HIP_TRY(c, hipGetLastError());
  return BBR_OK;
}

}  // extern "C"

#ifdef BB_STAMPS
// diagnostic build only: per-wave phase cycles of the last k_shade launch (8 x u64 per wave, 4096 waves)
extern "C" int bbr_debug_shade_stamps(bbr_context *c, unsigned long long *out) {
  if (!c || !out) return BBR_ERR_INVALID_ARGUMENT;
  int rc = drain(c);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(bbr::g_shade_stamps), sizeof(unsigned long long) * 4096 * 8));
  return BBR_OK;
}
#endif
