// ASan/UBSan driver for the asset readers (CPU only): every input file plus truncations and byte flips of it.
// g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iinclude tools/asan_assets_fuzz.cpp bibim_renderer_amd/csrc/bb_assets.cpp -lz -o /tmp/fuzz && /tmp/fuzz files...
// Round 1: 1647 loads (140 accepted, 1507 rejected), no sanitizer report.
#include "bibim_assets.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <random>
extern "C" int bbr_upload_material(bbr_context*, const bbr_image*, int32_t*) { return -1; }
extern "C" const char* bbr_last_error(const bbr_context*) { return "stub"; }
static std::vector<unsigned char> slurp(const char* p){ std::vector<unsigned char> v; FILE* f=fopen(p,"rb"); if(!f) return v; fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET); v.resize(n); if(n) (void)!fread(v.data(),1,n,f); fclose(f); return v; }
static void spit(const char* p, const std::vector<unsigned char>& v){ FILE* f=fopen(p,"wb"); fwrite(v.data(),1,v.size(),f); fclose(f); }
static int try_file(const char* path, int kind){
  if(kind==0){ uint8_t* px=nullptr; int32_t w,h; int rc=bba_load_png(path,&px,&w,&h); if(rc==0) bba_free(px); return rc; }
  if(kind==1){ void* v=nullptr; uint32_t n; int rc=bba_load_fbx_vertices(path,&v,&n); if(rc==0) bba_free(v); return rc; }
  void* v=nullptr; uint32_t nv, ni; uint32_t* ix=nullptr; int rc=bba_load_obj_gizmo(path,&v,&nv,&ix,&ni); if(rc==0){ bba_free(v); bba_free(ix);} return rc;
}
int main(int argc, char** argv){
  std::mt19937 rng(1234); long ok=0, bad=0;
  for(int a=1;a<argc;++a){
    std::string p=argv[a]; int kind = p.size()>4 && p.substr(p.size()-4)==".png" ? 0 : (p.substr(p.size()-4)==".fbx" ? 1 : 2);
    auto orig=slurp(argv[a]); if(orig.empty()) continue;
    if(try_file(argv[a],kind)==0) ++ok; else ++bad;
    const char* tmp = kind==2 ? "/tmp/bba_fuzz_m.obj" : (kind==1 ? "/tmp/bba_fuzz_m.fbx" : "/tmp/bba_fuzz_m.png");
    int rounds = orig.size() > 4000000 ? 6 : 60;
    for(int r=0;r<rounds;++r){
      auto v=orig;
      if(r%3==0) v.resize(rng()%v.size());
      else { int flips = 1 + rng()%8; for(int k=0;k<flips;++k){ size_t i = (r%3==1) ? rng()%std::min<size_t>(v.size(),512) : rng()%v.size(); v[i]^= (unsigned char)(1u<<(rng()%8)); } }
      spit(tmp,v);
      if(try_file(tmp,kind)==0) ++ok; else ++bad;
    }
  }
  printf("loads ok %ld, rejected %ld\n", ok, bad);
  return 0;
}
