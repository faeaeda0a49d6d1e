// shaderball_demo.cpp -- the reference's render loop for the ShaderBall scene, natively in C++ on the shim:
// bb::ShaderBallScene + bb::FreeLookCamera + bb::drawFrame (include/bibim_scene.h) -> libbibim_hip.so, then
// bbr_present and a binary PPM of the presented image.  What a maintainer's main.cpp looks like after INTEGRATION.md.
//
//   g++ -std=c++17 -O2 -Iinclude examples/shaderball_demo.cpp -Lbibim_renderer_amd -lbibim_hip -Wl,-rpath,$PWD/bibim_renderer_amd -o demo
//   ./demo --fbx resources/ShaderBall.fbx            (or --vertices-bin file: raw bb::Vertex records)
//          [--size 1920 1080] [--grid 4] [--frames 100] [--frames-in-flight 2] [--deferred] [--tone-map 1.0] [--pbr-dir resources/pbr/bark1]
//          [--gizmo resources/gizmo.obj] [--out frame.ppm]        (--gizmo also turns the light markers on)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bibim_assets.h"
#include "bibim_scene.h"

static int die(const char *what, bbr_context *ctx = nullptr) {
  std::fprintf(stderr, "%s: %s\n", what, ctx ? bbr_last_error(ctx) : bba_last_error());
  return 1;
}

int main(int argc, char **argv) {
  int width = 1920, height = 1080, grid = 1, frames = 100, in_flight = 2;  // the reference keeps 2 (src/main.cpp:38)
  bool deferred = false, tone = false;
  float exposure = 1.f;
  std::string fbx, vbin, out = "frame.ppm", pbr, gizmo;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--fbx" && i + 1 < argc) fbx = argv[++i];
    else if (a == "--vertices-bin" && i + 1 < argc) vbin = argv[++i];
    else if (a == "--size" && i + 2 < argc) { width = std::atoi(argv[++i]); height = std::atoi(argv[++i]); }
    else if (a == "--grid" && i + 1 < argc) grid = std::atoi(argv[++i]);
    else if (a == "--frames" && i + 1 < argc) frames = std::atoi(argv[++i]);
    else if (a == "--frames-in-flight" && i + 1 < argc) in_flight = std::atoi(argv[++i]);
    else if (a == "--deferred") deferred = true;
    else if (a == "--tone-map" && i + 1 < argc) { tone = true; exposure = (float)std::atof(argv[++i]); }
    else if (a == "--pbr-dir" && i + 1 < argc) pbr = argv[++i];
    else if (a == "--gizmo" && i + 1 < argc) gizmo = argv[++i];
    else if (a == "--out" && i + 1 < argc) out = argv[++i];
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }

  // --- the ball: imported like ShaderBallScene's constructor does (src/scene.cpp:57-86), or raw Vertex records
  std::vector<bb::Vertex> ball;
  if (!fbx.empty()) {
    void *v = nullptr;
    uint32_t n = 0;
    if (bba_load_fbx_vertices(fbx.c_str(), &v, &n) != BBA_OK) return die("bba_load_fbx_vertices");
    ball.assign(static_cast<bb::Vertex *>(v), static_cast<bb::Vertex *>(v) + n);
    bba_free(v);
  } else if (!vbin.empty()) {
    FILE *f = std::fopen(vbin.c_str(), "rb");
    if (!f) { std::perror(vbin.c_str()); return 1; }
    std::fseek(f, 0, SEEK_END);
    long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    ball.resize((size_t)bytes / sizeof(bb::Vertex));
    if (std::fread(ball.data(), sizeof(bb::Vertex), ball.size(), f) != ball.size()) return 1;
    std::fclose(f);
  } else {
    std::fprintf(stderr, "need --fbx or --vertices-bin\n");
    return 2;
  }

  bbr_context *ctx = nullptr;
  if (bbr_create(width, height, 0, &ctx) != BBR_OK) return die("bbr_create", nullptr), 1;
  if (bbr_set_option(ctx, "frames_in_flight", in_flight) != BBR_OK) return die("frames_in_flight", ctx), 1;
  int rc = 0;
  {
    // --- material: a pbr/<name>/ directory of the reference, or the built-in default maps
    int32_t material = -1;
    if (!pbr.empty()) {
      if (bba_load_material_dir(ctx, pbr.c_str(), &material) != BBA_OK) return die("bba_load_material_dir");
    } else {
      bbr_image none[BBR_MAP_COUNT] = {};
      if (bbr_upload_material(ctx, none, &material) != BBR_OK) return die("bbr_upload_material", ctx);
    }

    // --- overlay subpass (src/main.cpp:128-171): the gizmo mesh, and frames that keep their depth
    if (!gizmo.empty()) {
      void *gv = nullptr;
      uint32_t *gi = nullptr, ngv = 0, ngi = 0;
      if (bba_load_obj_gizmo(gizmo.c_str(), &gv, &ngv, &gi, &ngi) != BBA_OK) return die("bba_load_obj_gizmo");
      if (bbr_set_option(ctx, "overlays", 1) != BBR_OK || bbr_upload_gizmo(ctx, gv, ngv, gi, ngi) != BBR_OK) return die("bbr_upload_gizmo", ctx);
      bba_free(gv);
      bba_free(gi);
    }

    bb::ShaderBallScene scene(ctx, ball.data(), (uint32_t)ball.size(), grid);
    scene.SceneRenderPassType = deferred ? bb::RenderPassType::Deferred : bb::RenderPassType::Forward;
    bb::FreeLookCamera cam;  // reference default: origin, yaw = pitch = 0 (src/main.cpp:1123)
    if (grid > 1) { cam.Pos = {0.f, 2.f, -2.f}; cam.Pitch = -15.f; }
    bb::FrameSettings settings;
    settings.EnableNormalMap = true;
    settings.EnableToneMapping = tone;
    settings.Exposure = exposure;

    // first frame sizes the capacities; then the timed loop
    if (bb::drawFrame(ctx, scene, cam, settings, material, width, height) != BBR_OK || bbr_synchronize(ctx) != BBR_OK)
      return die("drawFrame", ctx);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < frames; ++i)
      if (bb::drawFrame(ctx, scene, cam, settings, material, width, height, 0.f) != BBR_OK) return die("drawFrame", ctx);
    if (bbr_synchronize(ctx) != BBR_OK) return die("bbr_synchronize", ctx);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    bbr_stats st;
    bbr_get_stats(ctx, &st);
    int32_t layout = 0, decided = 0;
    bbr_stream_layout_state(ctx, &layout, &decided, nullptr);
    std::printf("%dx%d, %u triangles, %llu shaded pixels, %s pass, %d frames in flight, stream layout %d%s: %.1f us/frame = %.0f Mpixels/s\n",
                width, height, (unsigned)st.n_prims, (unsigned long long)st.n_shaded, deferred ? "deferred" : "forward", in_flight,
                layout, decided ? "" : " (still being timed)", frames ? dt / frames * 1e6 : 0.0,
                frames ? (double)width * height * frames / dt / 1e6 : 0.0);

    // --- present (tone map + sRGB + RGBA8) and write it out
    std::vector<uint8_t> rgba((size_t)width * height * 4);
    if (bbr_present(ctx, nullptr, 1) != BBR_OK) return die("bbr_present", ctx);
    if (!gizmo.empty() && bbr_draw_overlays(ctx, 100) != BBR_OK) return die("bbr_draw_overlays", ctx);
    if (bbr_read_presented(ctx, rgba.data()) != BBR_OK) return die("bbr_read_presented", ctx);
    FILE *f = std::fopen(out.c_str(), "wb");
    if (!f) { std::perror(out.c_str()); rc = 1; }
    else {
      std::fprintf(f, "P6\n%d %d\n255\n", width, height);
      for (size_t p = 0; p < (size_t)width * height; ++p) std::fwrite(&rgba[4 * p], 1, 3, f);
      std::fclose(f);
      std::printf("wrote %s\n", out.c_str());
    }
  }  // the scene frees its meshes before the context goes
  bbr_destroy(ctx);
  return rc;
}
