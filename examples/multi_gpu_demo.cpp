// multi_gpu_demo.cpp -- BASELINE's configuration #4 natively in C++: ONE host process drives N ranks (a bbr_context
// each, on GPU rank % bbr_device_count()), every rank renders its interleaved screen bands of the same frame, and the
// frame is completed by the library's peer exchange: bbr_push_shard (the rank's block into every rank's gather buffer, one
// kernel storing to all peers at once) + bbr_unpack_whole.  No torch, no MPI, no HIP headers: g++ and libbibim_hip.so.
//
//   g++ -std=c++17 -O2 -Iinclude examples/multi_gpu_demo.cpp -Lbibim_renderer_amd -lbibim_hip -Wl,-rpath,$PWD/bibim_renderer_amd -o mgdemo
//   ./mgdemo --vertices-bin ball.bin --ranks 8 [--size 3840 2160] [--grid 4] [--frames 50] [--form packed|rgba32f|rgba8]
//            [--tone-map 1.0] [--out frame.ppm]
// With fewer GPUs than ranks several ranks share a device (that is how the test runs it on one GPU).  The loop is the
// simple one: all ranks render and push, the host waits for every rank ("all pushes have landed"), all ranks
// un-interleave -- and every rank owns TWO gather buffers used in turn: the un-interleave of frame n is only queued when
// frame() returns, and the pushes of frame n + 1 must not overwrite the buffer it is still reading.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bibim_scene.h"

static int die(const char *what, bbr_context *ctx) {
  std::fprintf(stderr, "%s: %s\n", what, bbr_last_error(ctx));
  return 1;
}

int main(int argc, char **argv) {
  int width = 3840, height = 2160, grid = 4, frames = 50, ranks = 2, form = BBR_SHARD_PACKED;
  bool tone = false;
  float exposure = 1.f;
  std::string vbin, out = "frame.ppm";
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--vertices-bin" && i + 1 < argc) vbin = argv[++i];
    else if (a == "--size" && i + 2 < argc) { width = std::atoi(argv[++i]); height = std::atoi(argv[++i]); }
    else if (a == "--grid" && i + 1 < argc) grid = std::atoi(argv[++i]);
    else if (a == "--frames" && i + 1 < argc) frames = std::atoi(argv[++i]);
    else if (a == "--ranks" && i + 1 < argc) ranks = std::atoi(argv[++i]);
    else if (a == "--tone-map" && i + 1 < argc) { tone = true; exposure = (float)std::atof(argv[++i]); }
    else if (a == "--form" && i + 1 < argc) {
      std::string f = argv[++i];
      // rgba16f: the reference's own HDR attachment format on the wire (8 bytes per pixel); presenting the widened frame
      // gives the same image, because presentation starts by rounding to binary16 anyway
      form = f == "rgba32f" ? BBR_SHARD_RGBA32F : (f == "rgba8" ? BBR_SHARD_RGBA8 : (f == "rgba16f" ? BBR_SHARD_RGBA16F : BBR_SHARD_PACKED));
    }
    else if (a == "--out" && i + 1 < argc) out = argv[++i];
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  if (vbin.empty() || ranks < 1 || ranks > 64) { std::fprintf(stderr, "need --vertices-bin and 1 <= --ranks <= 64\n"); return 2; }
  std::vector<bb::Vertex> ball;
  {
    FILE *f = std::fopen(vbin.c_str(), "rb");
    if (!f) { std::perror(vbin.c_str()); return 1; }
    std::fseek(f, 0, SEEK_END);
    long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    ball.resize((size_t)bytes / sizeof(bb::Vertex));
    if (std::fread(ball.data(), sizeof(bb::Vertex), ball.size(), f) != ball.size()) return 1;
    std::fclose(f);
  }
  const int n_gpus = bbr_device_count();
  if (n_gpus < 1) { std::fprintf(stderr, "no HIP device\n"); return 1; }

  struct Rank {
    bbr_context *ctx = nullptr;
    int32_t device = 0, material = -1;
    bb::ShaderBallScene *scene = nullptr;
    void *gathered[2] = {nullptr, nullptr};
  };
  std::vector<Rank> rk(ranks);
  std::vector<void *> peer_gathered[2] = {std::vector<void *>(ranks), std::vector<void *>(ranks)};
  std::vector<int32_t> peer_devices(ranks);
  uint64_t block = 0;
  for (int r = 0; r < ranks; ++r) {
    Rank &k = rk[r];
    k.device = r % n_gpus;
    if (bbr_create(width, height, k.device, &k.ctx) != BBR_OK) { std::fprintf(stderr, "bbr_create: %s\n", bbr_last_error(nullptr)); return 1; }
    if (bbr_set_partition(k.ctx, r, ranks, 0) != BBR_OK) return die("bbr_set_partition", k.ctx);
    bbr_image none[BBR_MAP_COUNT] = {};
    if (bbr_upload_material(k.ctx, none, &k.material) != BBR_OK) return die("bbr_upload_material", k.ctx);
    k.scene = new bb::ShaderBallScene(k.ctx, ball.data(), (uint32_t)ball.size(), grid);
    if (bbr_exchange_block_bytes(k.ctx, form, &block) != BBR_OK) return die("bbr_exchange_block_bytes", k.ctx);
    for (int b = 0; b < 2; ++b) {
      if (bbr_device_alloc(k.ctx, block * (uint64_t)ranks, &k.gathered[b]) != BBR_OK) return die("bbr_device_alloc", k.ctx);
      peer_gathered[b][r] = k.gathered[b];
    }
    peer_devices[r] = k.device;
  }
  bb::FreeLookCamera cam;
  if (grid > 1) { cam.Pos = {0.f, 2.f, -2.f}; cam.Pitch = -15.f; }
  bb::FrameSettings settings;
  settings.EnableNormalMap = true;
  settings.EnableToneMapping = tone;
  settings.Exposure = exposure;

  int frame_no = 0;
  auto frame = [&]() -> int {
    const int b = frame_no++ & 1;  // this frame's gather buffers; the other set may still be read by the previous frame's unpack
    for (Rank &k : rk) {  // every rank: its bands of the frame, then its block into every rank's gather buffer
      if (bb::drawFrame(k.ctx, *k.scene, cam, settings, k.material, width, height, 0.f) != BBR_OK) return die("drawFrame", k.ctx);
      if (form == BBR_SHARD_RGBA8 && bbr_present(k.ctx, nullptr, 1) != BBR_OK) return die("bbr_present", k.ctx);
      if (bbr_push_shard(k.ctx, form, peer_gathered[b].data(), peer_devices.data(), nullptr) != BBR_OK) return die("bbr_push_shard", k.ctx);
    }
    for (Rank &k : rk)  // all pushes have landed (and the unpack of frame n - 1, which read the other buffers, is done)
      if (bbr_synchronize(k.ctx) != BBR_OK) return die("bbr_synchronize", k.ctx);
    for (Rank &k : rk)  // every rank un-interleaves its copy of the gathered blocks into the whole frame
      if (bbr_unpack_whole(k.ctx, form, k.gathered[b], nullptr, nullptr) != BBR_OK) return die("bbr_unpack_whole", k.ctx);
    return 0;
  };
  // the first frames size the capacities (a synchronising call re-renders an overflowed frame, the exchange does not)
  for (Rank &k : rk)
    if (bb::drawFrame(k.ctx, *k.scene, cam, settings, k.material, width, height, 0.f) != BBR_OK || bbr_synchronize(k.ctx) != BBR_OK)
      return die("drawFrame", k.ctx);
  if (frame()) return 1;
  for (Rank &k : rk) bbr_synchronize(k.ctx);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < frames; ++i)
    if (frame()) return 1;
  for (Rank &k : rk)
    if (bbr_synchronize(k.ctx) != BBR_OK) return die("bbr_synchronize", k.ctx);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("%dx%d over %d ranks on %d GPU(s), %s blocks of %llu bytes: %.1f us/frame = %.0f Mpixels/s (simple loop, one frame at a time)\n",
              width, height, ranks, n_gpus < ranks ? n_gpus : ranks, form == BBR_SHARD_PACKED ? "packed" : (form == BBR_SHARD_RGBA8 ? "RGBA8" : (form == BBR_SHARD_RGBA16F ? "RGBA16F" : "RGBA32F")),
              (unsigned long long)block, frames ? dt / frames * 1e6 : 0.0, frames ? (double)width * height * frames / dt / 1e6 : 0.0);

  // the LAST rank's whole frame -> PPM (fp32 forms: presented on the host side of this demo with the library's own step)
  Rank &last = rk[ranks - 1];
  std::vector<uint8_t> rgba((size_t)width * height * 4);
  if (form == BBR_SHARD_RGBA8) {
    if (bbr_read_whole_frame(last.ctx, rgba.data()) != BBR_OK) return die("bbr_read_whole_frame", last.ctx);
  } else {
    void *whole = nullptr, *presented = nullptr;
    if (bbr_whole_frame_device_ptr(last.ctx, &whole, nullptr) != BBR_OK) return die("bbr_whole_frame_device_ptr", last.ctx);
    if (bbr_device_alloc(last.ctx, (uint64_t)width * height * 4, &presented) != BBR_OK) return die("bbr_device_alloc", last.ctx);
    if (bbr_present_buffer(last.ctx, whole, presented, (uint64_t)width * height, tone ? 1 : 0, exposure, 1, nullptr) != BBR_OK ||
        bbr_synchronize(last.ctx) != BBR_OK)
      return die("bbr_present_buffer", last.ctx);
    if (bbr_copy_to_host(last.ctx, rgba.data(), presented, (uint64_t)width * height * 4) != BBR_OK) return die("bbr_copy_to_host", last.ctx);
    bbr_device_free(last.ctx, presented);
  }
  FILE *f = std::fopen(out.c_str(), "wb");
  if (!f) { std::perror(out.c_str()); return 1; }
  std::fprintf(f, "P6\n%d %d\n255\n", width, height);
  for (size_t p = 0; p < (size_t)width * height; ++p) std::fwrite(&rgba[4 * p], 1, 3, f);
  std::fclose(f);
  std::printf("wrote %s\n", out.c_str());
  for (Rank &k : rk) {
    delete k.scene;
    bbr_device_free(k.ctx, k.gathered[0]);
    bbr_device_free(k.ctx, k.gathered[1]);
    bbr_destroy(k.ctx);
  }
  return 0;
}
