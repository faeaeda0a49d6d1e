/*
 * bibim_hip.h -- C ABI of the MI355X-native forward PBR path (libbibim_hip.so).
 *
 * The reference (chromedays/bibim-renderer) has no plugin / FFI layer: its forward path is reached
 * through SceneBase::drawScene(const Frame&) (src/scene.h:84) and the DATA it hands to Vulkan.  Each
 * entry point below replaces one of those hand-offs and consumes the reference's byte layouts unchanged:
 *
 *   bbr_upload_mesh          createVertexBuffer / createIndexBuffer            src/scene.h:86-108
 *                            (bb::Vertex 44 B, src/render.h:112-117; VK_INDEX_TYPE_UINT32, src/scene.cpp:209)
 *   bbr_upload_material      createPBRMaterialSet + set 2 binding               src/render.cpp:1243-1336,
 *                            (6 RGBA8 images in PBRMapType order)               src/shaders/standard_sets.glsl:45-50
 *   bbr_set_frame_uniforms   FrameUniformBlock map/memcpy (6432 B)              src/main.cpp:1288-1327
 *   bbr_set_view_uniforms    ViewUniformBlock map/memcpy (144 B)                src/main.cpp:1329-1342
 *   bbr_begin_frame          vkCmdBeginRenderPass, all clears = 0               src/main.cpp:78-86
 *   bbr_draw                 updateInstanceBufferMemory + vkCmdDraw[Indexed]    src/scene.h:120-132,
 *                            (bb::InstanceBlock 128 B, src/render.h:96-99)      src/scene.cpp:203-210
 *   bbr_end_frame            vkCmdEndRenderPass + vkQueueSubmit                 src/main.cpp:174,1364
 *   bbr_read_framebuffer     the HDR colour attachment (kept fp32 RGBA)         src/main.cpp:463-472
 *   pipeline state           createPipeline + forward params (fixed here)       src/render.cpp:1044-1178,
 *                                                                               src/main.cpp:332-350
 *
 * Conventions: every call returns BBR_OK (0) or a negative bbr_status; bbr_last_error() gives text.
 * A context is externally synchronised (one thread at a time), as the reference's single render thread.
 * Caller-owned host memory is copied before the call returns (as createDeviceLocalBufferFromMemory does,
 * src/render.cpp:706-726).  All device work is queued on one HIP stream; bbr_end_frame does not block.
 * There is NO CPU fallback: without a HIP device every compute entry point fails with BBR_ERR_NO_DEVICE.
 */
#ifndef BIBIM_HIP_H
#define BIBIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bbr_context bbr_context;

typedef enum bbr_status {
  BBR_OK = 0,
  BBR_ERR_INVALID_ARGUMENT = -1,
  BBR_ERR_NO_DEVICE = -2,
  BBR_ERR_HIP = -3,
  BBR_ERR_OUT_OF_MEMORY = -4,
  BBR_ERR_BAD_HANDLE = -5,
  BBR_ERR_NOT_IN_FRAME = -6,
  BBR_ERR_TOO_MANY_PRIMITIVES = -7,
  BBR_ERR_CAPACITY = -8
} bbr_status;

/* One RGBA8 image, stb_image STBI_rgb_alpha layout (src/resource.cpp:159-160).
 * rgba == NULL selects the `default` material's map for that slot (src/render.cpp:1328-1336). */
typedef struct bbr_image {
  const uint8_t *rgba;
  int32_t width;
  int32_t height;
} bbr_image;

/* PBRMapType order (src/render.h:235-243) */
enum { BBR_MAP_ALBEDO = 0, BBR_MAP_METALLIC, BBR_MAP_ROUGHNESS, BBR_MAP_AO, BBR_MAP_NORMAL, BBR_MAP_HEIGHT, BBR_MAP_COUNT };

/* Byte sizes the ABI accepts verbatim */
#define BBR_SIZEOF_VERTEX 44
#define BBR_SIZEOF_INSTANCE_BLOCK 128
#define BBR_SIZEOF_LIGHT 64
#define BBR_SIZEOF_FRAME_UNIFORM_BLOCK 6432
#define BBR_SIZEOF_VIEW_UNIFORM_BLOCK 144
#define BBR_MAX_NUM_LIGHTS 100

typedef struct bbr_stats {
  uint64_t n_prims;        /* triangles submitted this frame */
  uint64_t n_raster_tris;  /* (sub-)triangles that survived clip + cull + "covers a pixel centre" */
  uint64_t n_clipped_prims;
  uint64_t n_bin_refs;     /* (tile, triangle) pairs written to bins */
  uint64_t n_broad_tris;   /* triangles routed to the every-tile list */
  uint64_t n_shaded;       /* pixels of the owned rows whose winning fragment is geometry */
  uint32_t bin_overflow;   /* non-zero if a capacity was exceeded (frame is re-rendered transparently) */
  uint32_t tile_w, tile_h;
  uint32_t n_tiles;
} bbr_stats;

/* ---- lifetime ---- */
int bbr_create(int32_t width, int32_t height, int32_t device, bbr_context **out_ctx);
int bbr_destroy(bbr_context *ctx);
/* onWindowResize (src/main.cpp:1042-1061: vkDeviceWaitIdle, cleanupReloadableResources, initReloadableResources): waits
 * for the frames in flight, drops every buffer whose size follows the extent and continues with the new one.  Meshes,
 * materials, options, the partition and the grown capacities stay.  There is no current frame afterwards (read-backs
 * fail with BBR_ERR_NOT_IN_FRAME until one is rendered) and a caller-owned output buffer has to be set again.
 * Not allowed between bbr_begin_frame and bbr_end_frame. */
int bbr_resize(bbr_context *ctx, int32_t width, int32_t height);
const char *bbr_last_error(const bbr_context *ctx); /* ctx may be NULL: last creation error */
int bbr_device_count(void);

/* ---- resources ---- */
int bbr_upload_mesh(bbr_context *ctx, const void *vertices, uint32_t n_vertices, const uint32_t *indices_or_null,
                    uint32_t n_indices, int32_t *out_mesh);
int bbr_upload_material(bbr_context *ctx, const bbr_image maps[BBR_MAP_COUNT], int32_t *out_material);
int bbr_free_mesh(bbr_context *ctx, int32_t mesh);
int bbr_free_material(bbr_context *ctx, int32_t material);

/* ---- per-frame state ---- */
int bbr_set_frame_uniforms(bbr_context *ctx, const void *frame_uniform_block /* 6432 B */);
int bbr_set_view_uniforms(bbr_context *ctx, const void *view_uniform_block /* 144 B */);

/* ---- frame ---- */
int bbr_begin_frame(bbr_context *ctx);
int bbr_draw(bbr_context *ctx, int32_t mesh, int32_t material, const void *instance_blocks, uint32_t n_instances);
/* Capacities that depend on the scene (tile bins, every-tile list, clip arena) grow by themselves.  A synchronising
 * call (bbr_synchronize, bbr_get_stats, any read-back) that finds the last frame overflowed re-renders it with larger
 * buffers before returning.  A host that only streams frames is covered too: every frame stores its overflow bits
 * into pinned host memory (two words, from the raster kernel) and the next frame that reuses its slot grows the
 * capacities first -- the overflowed
 * frames themselves (at most frames_in_flight + 1 of them) are incomplete and stay so. */
int bbr_end_frame(bbr_context *ctx);   /* queues the kernels; asynchronous */
int bbr_replay_frame(bbr_context *ctx); /* re-submit the last recorded frame (same draws and uniforms) */
int bbr_synchronize(bbr_context *ctx);

/* ---- output ---- */
/* Full frame: height*width*4 floats, row-major, RGBA.  On a partitioned context (world > 1) this fails with
 * BBR_ERR_INVALID_ARGUMENT: the output is a compact shard, use bbr_read_shard. */
int bbr_read_framebuffer(bbr_context *ctx, float *rgba32f_host);
int bbr_framebuffer_device_ptr(bbr_context *ctx, void **out_device_ptr, uint64_t *out_bytes);
/* Render into caller-provided device memory instead (e.g. a torch tensor that RCCL all-gathers);
 * NULL restores the internal buffer.  bytes must cover the frame (or the shard when partitioned). */
int bbr_set_output_device_ptr(bbr_context *ctx, void *device_ptr, uint64_t bytes);
/* By default a context owns four HIP streams and keeps two frames in flight, like the reference (numFrames,
 * src/main.cpp:38): every frame slot has its own buffers, counter block and stream ("stream_layout" 2, the default: all
 * kernels of a frame on the stream of its slot, so whole frames overlap and nothing inside a frame needs an event); the
 * host blocks in bbr_end_frame only when the slot it wants to reuse is still on the GPU (bbr_host_timing counts that).
 * Options "frames_in_flight" 1..4 and "stream_layout" change this.  bbr_read_framebuffer / bbr_framebuffer_device_ptr refer
 * to the most recently submitted frame.
 * bbr_set_stream(stream != NULL) puts ALL of the context's work on the caller's stream instead (one frame in flight,
 * plain stream ordering with the caller's other work); NULL returns to the context's own streams. */
int bbr_set_stream(bbr_context *ctx, void *hip_stream);
/* Cross-stream hand-offs for callers that pipeline frames against their own streams (e.g. an RCCL all-gather of
 * frame N overlapping the rendering of frame N+1):
 *   bbr_wait_event        device work submitted after this call waits for `hip_event` (a hipEvent_t the caller recorded,
 *                         e.g. "the consumer of the output buffer has finished with it");
 *   bbr_stream_wait_frame `hip_stream` (a hipStream_t) waits until the most recently submitted frame is complete. */
int bbr_wait_event(bbr_context *ctx, void *hip_event);
int bbr_stream_wait_frame(bbr_context *ctx, void *hip_stream);

/* ---- screen-band partition across GPUs (no reference counterpart) ---- */
/* Band b (band_rows framebuffer rows, a multiple of the tile height) belongs to rank b % world.  The
 * rank's output becomes a compact shard [local band][row in band][x][rgba]; every rank's shard is padded
 * to bbr_shard_rows() rows so that an all-gather of equal-sized shards reassembles the frame. */
int bbr_set_partition(bbr_context *ctx, int32_t rank, int32_t world, int32_t band_rows);
int bbr_shard_rows(const bbr_context *ctx, int32_t *out_rows);
int bbr_read_shard(bbr_context *ctx, float *rgba32f_host); /* shard_rows*width*4 floats */
/* Device-side un-interleave of an all-gathered buffer [world][shard_rows][width][4] into a row-major frame, queued on
 * `hip_stream` (NULL = the context's shading stream). */
int bbr_unpack_gathered(bbr_context *ctx, const void *gathered_device, void *frame_device, void *hip_stream);
/* The same exchange with a quarter less payload, lossless: alpha is 1.0 on shaded pixels and 0.0 on cleared ones
 * (forward_brdf.frag:75, clear colour src/main.cpp:84; the deferred path writes 1 everywhere), so a shard travels as
 * rgb[n][3] float + one bit per pixel, n = shard_rows * width: bbr_packed_shard_bytes() bytes per rank.
 *   bbr_pack_shard              the last frame's shard -> `packed_device`, queued on `hip_stream` (NULL = the stream the
 *                               frame was shaded on; a caller's stream must already wait for the frame, bbr_stream_wait_frame)
 *   bbr_unpack_gathered_packed  [world] packed blocks -> the row-major RGBA32F frame, bit for bit what bbr_unpack_gathered
 *                               makes of the RGBA32F shards */
int bbr_packed_shard_bytes(const bbr_context *ctx, uint64_t *out_bytes);
int bbr_pack_shard(bbr_context *ctx, void *packed_device, void *hip_stream);
int bbr_unpack_gathered_packed(bbr_context *ctx, const void *gathered_device, void *frame_device, void *hip_stream);
int bbr_tile_height(const bbr_context *ctx, int32_t *out_tile_h);

/* ---- diagnostics ---- */
int bbr_get_stats(bbr_context *ctx, bbr_stats *out);                            /* synchronises (and re-renders an overflowed frame) */
/* How often a capacity (bins, every-tile list, clip arena) has been grown since bbr_create (= bbr_stats.bin_overflow), without
 * synchronising or touching the GPU: a host that times frames can tell afterwards whether one of them overflowed. */
int bbr_capacity_growths(const bbr_context *ctx, uint32_t *out_count);          /* host-side counter: does NOT synchronise */
/* The host's side of the frame loop since bbr_host_timing_reset, kept without touching the GPU (steady_clock around the
 * submit): frames submitted by bbr_end_frame / bbr_replay_frame, nanoseconds spent inside those calls, and the part of
 * it the host was BLOCKED because the frame slot it wanted to reuse was still on the GPU (and in how many frames).  A
 * host that is never blocked is the bottleneck itself; one that is blocked every frame is waiting for the GPU.  This is
 * the frame loop the reference paces with a fence per frame in flight (src/main.cpp:1277-1279). */
int bbr_host_timing(const bbr_context *ctx, uint64_t *out_frames, uint64_t *out_submit_ns, uint64_t *out_blocked_ns,
                    uint64_t *out_blocked_frames);
int bbr_host_timing_reset(bbr_context *ctx);
/* winning primitive (global API-order index, 0xFFFFFFFF = none) and depth per pixel; synchronises.
 * Re-runs the frame once with the visibility dump enabled. */
int bbr_read_visibility(bbr_context *ctx, uint32_t *prim_host, float *depth_host);
/* device time of the last bbr_end_frame/bbr_replay_frame in ms, and of its dominant kernel (k_shade); synchronises */
int bbr_last_frame_time_ms(bbr_context *ctx, float *out_frame_ms, float *out_shade_kernel_ms);
/* With option "timing" = 1 every frame records HIP events on the context's stream (frame start, geometry kernel
 * done, raster kernel done, shade kernel done) into a ring of 512 frames.  bbr_timing_summary averages the frames
 * recorded since bbr_timing_reset: whole frame, geometry (incl. the H2D of instances/lights), raster, shade. */
int bbr_timing_reset(bbr_context *ctx);
int bbr_timing_summary(bbr_context *ctx, uint32_t *out_frames, float *out_avg_frame_ms, float *out_avg_geometry_ms,
                       float *out_avg_raster_ms, float *out_avg_shade_ms);
/* Options (all but "render_pass" drain the context first):
 *   "timing" 0|1|2           1: five HIP events per frame (frame start, geometry, raster, shade start, shade done);
 *                            2: only the two around k_shade (frame/geometry/raster averages read 0); restarts the ring
 *   "timing_stride" n        with "timing" on, only every n-th frame carries events (default 1): the events themselves
 *                            perturb a pipelined frame stream (two per frame: ~4 % of the C3 frame rate)
 *   "frames_in_flight" 1..4  default 2 (the reference's numFrames); bench.py runs 4K with 3 and 1080p with 4
 *   "tile_mode" 0|1          1 (default): 32x32 screen tiles; 0: 64x64 tiles -- kept for frames beyond 8192 x 8192 pixels
 *                            (65 536 tile slots), slower everywhere else (k_raster 120 vs 65 us at 4K, round 1)
 *   "bin_cap" n              initial references per (tile, raster class); grows by itself on overflow
 *   "broad_threshold" n      triangles touching more than n tiles go to the every-tile list
 *   "broad_cap" n            initial entries of the every-tile list (default 4096); doubles when a frame overflows it
 *   "clip_cap" n             initial sub-triangle slots of the clip arena (default 4096); doubles likewise
 *   "gbuffer_view" -1..3      deferred path only: instead of brdf.frag, buffer_visualize.frag shows the rgb of one G-buffer
 *                            attachment (GBufferVisualizingOption, src/scene.h:27-35; recordCommand src/main.cpp:96-121):
 *                            0 position, 1 normal, 2 albedo, 3 metallic / roughness / ao; -1 (default) the lit scene.
 *                            The frame then goes through presentation like any other
 *   "render_pass" 0|1        0: forward path (default, the path BASELINE measures), 1: deferred path
 *   "present_fused" 0|1      frames are produced as presented RGBA8 pixels directly (binary16 stage, tone map and sRGB
 *                            encode fused into the raster / shade kernels): no fp32 frame, no k_present pass; bbr_present
 *                            then only marks (or copies to a caller buffer), bbr_read_framebuffer / bbr_read_shard fail
 *   "overlays" 0|1           keep every frame's resolved depth for bbr_draw_overlays (default 0)
 *   "stream_layout" 0|1|2    how the kernels of the frames in flight are spread over HIP streams.  Same pixels in every
 *                            layout.  0: geometry + raster on one stream, shade on a second, present on a third.
 *                            1: as 0 with k_raster on a stream of its own (the geometry of frame N+1 overlaps the
 *                            raster of frame N).  2 (default): every kernel of a frame on the stream of its frame slot
 *                            (frames share nothing, whole frames overlap).  Three frames in flight, us per frame for
 *                            0 / 1 / 2: 1080p, one ShaderBall 84.5 / 67.1 / 37.3; 4K, sixteen 191.5 / 153.6 / 148.6.
 *   "push_mode" 0|1          bbr_push_shard: 1 (default) one kernel storing to every peer at once, 0 peer copies one after
 *                            the other (see "native exchange" below)
 *   "no_tail_items" n        (default 40000) what counts as a SHORT frame: at most n item slots (tiles x 64-fragment chunks
 *                            per tile: 1080p has 32 640).  A long frame's shading work list is built by a scan kernel
 *                            (k_shade_items, screen order), its main launch is sized from the item count the frame slot
 *                            produced one frame earlier and a small tail launch covers what that estimate misses.  A
 *                            short frame is a chain of dependent kernels whose length is its rate: its raster tiles append
 *                            their items themselves (no k_shade_items launch) and it is shaded at full coverage without
 *                            the tail launch (C2: chain 95 -> 85 us, 27.3 -> 24.6 us per frame with four in flight)
 *   "heavy_tiles" n          long frames: the raster kernel starts the tiles one of whose bins holds at least n triangle
 *                            references before the screen-ordered rest.  Same pixels.  0: plain screen order; -1 (default):
 *                            64 while one frame is in flight, 0 otherwise -- it shortens a single frame (C3: k_raster alone
 *                            61.9 -> 51.0 us, frame latency 177 -> 167 us) and costs 1-2 % of the pipelined frame rate
 *   "ablate" bits            diagnostic builds only (make EXTRA=-DBB_ABLATE): skip parts of the pipeline */
int bbr_set_option(bbr_context *ctx, const char *name, int64_t value);
/* The stream layout in use (option "stream_layout"; 0 while only one frame is in flight).  *out_decided is always 1 and
 * out_ms[3] all zero: the layout is a plain option, nothing is timed at run time (round 1 did). */
int bbr_stream_layout_state(const bbr_context *ctx, int32_t *out_layout, int32_t *out_decided, float *out_ms);
/* Self-test of the arithmetic contract on this device: the shader's reciprocal (v_rcp_f32 + one Newton step) against
 * the IEEE division for +x and -x of every float with bit pattern in [lo_bits, hi_bits); 0 mismatches expected.
 * The whole positive range 0 .. 0x7FFFFFFF takes about half a second. */
int bbr_selftest_rcp(bbr_context *ctx, uint32_t lo_bits, uint32_t hi_bits, uint64_t *out_mismatches);

/* ---- deferred variant ----
 * bbr_set_option(ctx, "render_pass", 1) renders with the reference's deferred path, its default
 * (SceneBase::SceneRenderPassType, src/scene.h:77; recordCommand src/main.cpp:89-104): gbuffer.vert/.frag into four
 * R16G16B16A16_SFLOAT attachments (src/main.cpp:443), then brdf.frag on every pixel.  The two subpasses are fused
 * (the G-buffer texel of a pixel is only read by the same pixel), so the G-buffer is not stored unless asked for:
 * bbr_read_gbuffer re-renders the last frame and returns width*height*16 floats, per pixel
 * position.xyz 1 | normal.xyz 0 | albedo.rgb 0 | metallic roughness ao height (binary16 values widened); synchronises. */
int bbr_read_gbuffer(bbr_context *ctx, float *gbuffer_host);

/* ---- overlay subpass ----
 * The reference draws its light markers and the corner gizmo into the swapchain image after tone mapping, depth-tested
 * against the scene (recordCommand, src/main.cpp:128-171; light.vert/.frag on generateUVSphereMesh(0.1, 16, 16), one
 * instance per light; gizmo.vert/.frag in a gizmo_extent^2 viewport at the top-right whose depth is cleared first).
 *   bbr_set_option(ctx, "overlays", 1)   frames keep their resolved depth (4 bytes per pixel) from now on
 *   bbr_upload_gizmo                     the gizmo mesh: 36-byte bb::GizmoVertex records (Pos, Color, Normal,
 *                                        src/render.h:122-126), optional uint32 indices (see bba_load_obj_gizmo)
 *   bbr_draw_overlays                    after bbr_present of the last frame: draws markers (the frame's own lights) and,
 *                                        if gizmo_extent > 0 and a gizmo was uploaded, the gizmo (the reference uses
 *                                        100) over the presented image.  Synchronous: a debugging aid, not part of the
 *                                        hot path.  Not available with a partition. */
int bbr_upload_gizmo(bbr_context *ctx, const void *gizmo_vertices, uint32_t n_vertices, const uint32_t *indices,
                     uint32_t n_indices);
int bbr_draw_overlays(bbr_context *ctx, int32_t gizmo_extent);

/* ---- presentation: the step after the path ----
 * Replaces the tone-map subpass + swapchain write (src/main.cpp:123-126, src/shaders/hdr_tone_mapping.frag:9-18,
 * HDR attachment R16G16B16A16_SFLOAT src/render.h:94, sRGB swapchain format src/render.cpp:242-254):
 * per pixel of the last frame  rgb -> [binary16 round, if hdr16] -> EnableToneMapping ? 1 - exp(-rgb * Exposure) : rgb
 * -> sRGB encode -> UNORM8, alpha = 255; EnableToneMapping / Exposure are the ones of the FrameUniformBlock the frame
 * was rendered with.  Asynchronous, queued behind the frame's shading.  `rgba8_device` = NULL writes a buffer owned by
 * the context (bbr_presented_device_ptr / bbr_read_presented); with a partition the image is this rank's shard
 * (bbr_shard_rows() rows), gathered with ncclAllGather + bbr_unpack_gathered_rgba8 at a quarter of the fp32 payload. */
int bbr_present(bbr_context *ctx, void *rgba8_device, int32_t hdr16);
int bbr_read_presented(bbr_context *ctx, uint8_t *rgba8_host); /* rows*width*4 bytes; synchronises */
/* the same conversion on any device buffer of n_pixels RGBA32F pixels (e.g. an all-gathered frame), queued on
 * `hip_stream` (NULL = the context's shading stream) */
int bbr_present_buffer(bbr_context *ctx, const void *rgba32f_device, void *rgba8_device, uint64_t n_pixels,
                       int32_t enable_tone_mapping, float exposure, int32_t hdr16, void *hip_stream);
int bbr_presented_device_ptr(bbr_context *ctx, void **out_ptr, uint64_t *out_bytes);
int bbr_unpack_gathered_rgba8(bbr_context *ctx, const void *gathered_device, void *frame_device, void *hip_stream);
/* with option "timing" on: launches of k_present since bbr_timing_reset and their average duration (HIP events) */
int bbr_present_timing(bbr_context *ctx, uint32_t *out_launches, float *out_avg_ms);
/* hdr_tone_mapping.frag:9-18 alone on the fp32 frame, in place (no quantisation) */
int bbr_tone_map(bbr_context *ctx, int32_t enable_tone_mapping, float exposure);

/* ---- native exchange: every rank gets the whole frame (BASELINE config #4) ----
 * The reference renders on one GPU; with a partition (bbr_set_partition) each rank holds a compact shard and the frame
 * is completed by one exchange step.  Two native forms, both queued on the stream of the frame's slot right behind its
 * shading (with stream layout 2 that stream carries nothing else: the next frame of the slot is ordered behind the
 * exchange without an event, the other slots render meanwhile), or on `hip_stream` when one is given (made to wait for
 * the frame first):
 *   collective  bbr_comm_unique_id on rank 0 -> the 128 bytes reach every rank by the host's own means ->
 *               bbr_comm_init on every rank (ncclCommInitRank; one process per GPU) -> per frame bbr_allgather_frame:
 *               [pack into this rank's slot of `gathered`] -> ncclAllGather in place (ring over xGMI) -> un-interleave into
 *               `whole`.  librccl is opened with dlopen at the first of these calls; a single-GPU host never loads it.
 *   peer        bbr_push_shard: this rank's block goes into every rank's gather buffer -- for one process driving several
 *               GPUs (one context each) or processes that exchanged bbr_ipc_export handles.  Option "push_mode" 1
 *               (default): ONE kernel loads the block once and stores it to all world - 1 peers, every xGMI link of the
 *               full mesh busy at the same time (the direct pattern: block / link bandwidth); peer access to the other
 *               devices is enabled on first use, and a peer that cannot be mapped falls back to mode 0.  "push_mode" 0:
 *               hipMemcpyPeerAsync copies queued one behind the other, nearest rank first -- one link at a time, i.e.
 *               ring timing.  bbr_push_state says which form the last push took.  The host orders "all pushes have
 *               landed" (events or a barrier) before bbr_unpack_whole, and must not let a rank push frame n + 1 into a
 *               buffer a peer is still un-interleaving frame n from: two gather buffers per rank, used in turn.
 * Block forms (what travels per rank; bbr_exchange_block_bytes): BBR_SHARD_RGBA32F the fp32 shard (16 B/pixel),
 * BBR_SHARD_PACKED rgb + one alpha bit per pixel (12.1 B, lossless: alpha is 0 or 1; = bbr_pack_shard), BBR_SHARD_RGBA8
 * the presented shard (4 B; needs bbr_present), BBR_SHARD_RGBA16F the shard rounded to binary16 (8 B, lossy: a separate
 * output, see the define).  `gathered` / `whole` = NULL use buffers owned by the frame's slot
 * (bbr_whole_frame_device_ptr, bbr_read_whole_frame).  An exchange never re-renders (one rank alone must not repeat a
 * collective): let the capacities settle with one synchronised frame first, as after any scene change.
 * Not yet run on more than one GPU: see DESIGN.md section 5. */
#define BBR_COMM_ID_BYTES 128
#define BBR_IPC_HANDLE_BYTES 64
#define BBR_SHARD_RGBA32F 0
#define BBR_SHARD_PACKED 1
#define BBR_SHARD_RGBA8 2
#define BBR_SHARD_RGBA16F 3 /* every channel rounded to binary16 -- the reference's own HDR attachment format, R16G16B16A16_SFLOAT
                             * (src/render.h:94, src/main.cpp:463-472): 8 B/pixel, LOSSY; the whole frame is widened back to RGBA32F */
int bbr_comm_unique_id(bbr_context *ctx, uint8_t *out_id /* BBR_COMM_ID_BYTES */);
int bbr_comm_init(bbr_context *ctx, int32_t rank, int32_t world, const uint8_t *unique_id /* BBR_COMM_ID_BYTES */);
int bbr_comm_destroy(bbr_context *ctx);
/* opens librccl (dlopen + symbol lookup) and nothing else: NOT collective.  Hosts vote on this before they enter
 * bbr_comm_init together -- a rank that cannot load the library must not leave the others waiting inside ncclCommInitRank */
int bbr_comm_probe(bbr_context *ctx);
/* ranks in the context's communicator as RCCL itself reports them (ncclCommCount) */
int bbr_comm_count(bbr_context *ctx, int32_t *out_ranks);
/* this rank's block of the last frame in `form`, written to `block_device` (bbr_exchange_block_bytes bytes) on `hip_stream`
 * (NULL = the frame's own stream): what bbr_allgather_frame / bbr_push_shard do first, for hosts that run the collective
 * themselves and finish with bbr_unpack_whole */
int bbr_stage_shard(bbr_context *ctx, int32_t form, void *block_device, void *hip_stream);
int bbr_exchange_block_bytes(const bbr_context *ctx, int32_t form, uint64_t *out_bytes);
int bbr_allgather_frame(bbr_context *ctx, int32_t form, void *gathered_device /* world blocks, or NULL */,
                        void *whole_device /* height*width pixels, or NULL */, void *hip_stream);
int bbr_push_shard(bbr_context *ctx, int32_t form, void *const *peer_gathered /* [world] device pointers */,
                   const int32_t *peer_devices /* [world] HIP device of each */, void *hip_stream);
/* 1 if the last bbr_push_shard stored through the one-kernel direct form, 0 if it queued copies */
int bbr_push_state(const bbr_context *ctx, int32_t *out_direct);
int bbr_unpack_whole(bbr_context *ctx, int32_t form, const void *gathered_device, void *whole_device, void *hip_stream);
int bbr_whole_frame_device_ptr(bbr_context *ctx, void **out_ptr, uint64_t *out_bytes);
int bbr_read_whole_frame(bbr_context *ctx, void *host /* height*width*16 bytes (RGBA8 form: *4); synchronises */);
/* plain device memory on the context's GPU (zero-filled), for hosts that do not link HIP themselves: gather buffers */
int bbr_device_alloc(bbr_context *ctx, uint64_t bytes, void **out_ptr);
int bbr_device_free(bbr_context *ctx, void *ptr);
int bbr_copy_to_host(bbr_context *ctx, void *host, const void *device_ptr, uint64_t bytes); /* drains the context first */
int bbr_ipc_export(bbr_context *ctx, void *device_ptr, uint8_t *out_handle /* BBR_IPC_HANDLE_BYTES */);
int bbr_ipc_open(bbr_context *ctx, const uint8_t *handle, void **out_ptr);
int bbr_ipc_close(bbr_context *ctx, void *ptr);

#ifdef __cplusplus
}
#endif
#endif
