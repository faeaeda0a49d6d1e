// bb_assets.cpp -- asset ingestion (include/bibim_assets.h): binary FBX geometry, OBJ/MTL gizmo, PNG -> RGBA8, the
// pbr/<name>/ directory convention.  Host code; the only dependency besides the C++ runtime is zlib (inflate), which
// both FBX arrays and PNG IDAT streams use.
#include "bibim_assets.h"

#include <dirent.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg) {
  g_error = msg;
  return code;
}

bool read_file(const char *path, std::vector<uint8_t> &out) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (n < 0) {
    std::fclose(f);
    return false;
  }
  out.resize((size_t)n);
  size_t got = n ? std::fread(out.data(), 1, (size_t)n, f) : 0;
  std::fclose(f);
  return got == (size_t)n;
}

bool inflate_all(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t expected) {
  // deflate cannot expand by more than ~1032:1: a declared size beyond that is a corrupt (or hostile) header, not a
  // reason to allocate gigabytes
  if (expected > n * 1100 + 4096) return false;
  out.resize(expected);
  z_stream zs;
  std::memset(&zs, 0, sizeof zs);
  if (inflateInit(&zs) != Z_OK) return false;
  zs.next_in = const_cast<Bytef *>(src);
  zs.avail_in = (uInt)n;
  zs.next_out = out.data();
  zs.avail_out = (uInt)expected;
  int rc = inflate(&zs, Z_FINISH);
  size_t produced = zs.total_out;
  inflateEnd(&zs);
  // a stream that still has output pending after `expected` bytes is tolerated for PNG (extra data is ignored there
  // too by stb_image); FBX arrays must match exactly, which the caller checks through `produced`
  if (rc != Z_STREAM_END && !(rc == Z_BUF_ERROR || rc == Z_OK)) return false;
  out.resize(produced);
  return true;
}

template <class T>
T rd(const uint8_t *p) {
  T v;
  std::memcpy(&v, p, sizeof v);
  return v;
}

// ------------------------------------------------------------------------------------------------
// binary FBX
// ------------------------------------------------------------------------------------------------

struct FbxProp {
  char type = 0;
  double scalar = 0;
  std::vector<double> f64;  // 'd' and 'f' arrays, widened
  std::vector<int64_t> i64; // 'i', 'l', 'b' arrays
  std::string str;          // 'S', 'R'
};

struct FbxNode {
  std::string name;
  std::vector<FbxProp> props;
  std::vector<FbxNode> children;
  const FbxNode *find(const char *n) const {
    for (const FbxNode &c : children)
      if (c.name == n) return &c;
    return nullptr;
  }
};

struct FbxReader {
  const uint8_t *buf;
  size_t size;
  std::string err;
  bool wide = false;  // FBX >= 7500: EndOffset / NumProperties / PropertyListLen are 64-bit, the NULL record has 25 bytes

  bool need(size_t pos, size_t n) {
    if (pos > size || n > size - pos) {
      err = "truncated FBX record";
      return false;
    }
    return true;
  }

  bool props(size_t &pos, uint32_t count, std::vector<FbxProp> &out) {
    for (uint32_t i = 0; i < count; ++i) {
      if (!need(pos, 1)) return false;
      FbxProp p;
      p.type = (char)buf[pos++];
      switch (p.type) {
        case 'Y': if (!need(pos, 2)) return false; p.scalar = rd<int16_t>(buf + pos); pos += 2; break;
        case 'C': if (!need(pos, 1)) return false; p.scalar = buf[pos]; pos += 1; break;
        case 'I': if (!need(pos, 4)) return false; p.scalar = rd<int32_t>(buf + pos); pos += 4; break;
        case 'F': if (!need(pos, 4)) return false; p.scalar = rd<float>(buf + pos); pos += 4; break;
        case 'D': if (!need(pos, 8)) return false; p.scalar = rd<double>(buf + pos); pos += 8; break;
        case 'L': if (!need(pos, 8)) return false; p.scalar = (double)rd<int64_t>(buf + pos); pos += 8; break;
        case 'f': case 'd': case 'l': case 'i': case 'b': {
          if (!need(pos, 12)) return false;
          uint32_t n = rd<uint32_t>(buf + pos), enc = rd<uint32_t>(buf + pos + 4), clen = rd<uint32_t>(buf + pos + 8);
          pos += 12;
          if (!need(pos, clen)) return false;
          const size_t esz = (p.type == 'f' || p.type == 'i') ? 4 : (p.type == 'b' ? 1 : 8);
          std::vector<uint8_t> raw;
          const uint8_t *data = buf + pos;
          if (enc == 1) {
            if (!inflate_all(buf + pos, clen, raw, (size_t)n * esz) || raw.size() != (size_t)n * esz) {
              err = "FBX array: zlib stream does not inflate to its declared size";
              return false;
            }
            data = raw.data();
          } else if (enc != 0 || clen != (uint64_t)n * esz) {
            err = "FBX array: unknown encoding";
            return false;
          }
          pos += clen;
          if (p.type == 'd' || p.type == 'f') {
            p.f64.resize(n);
            for (uint32_t k = 0; k < n; ++k) p.f64[k] = p.type == 'd' ? rd<double>(data + 8 * k) : (double)rd<float>(data + 4 * k);
          } else {
            p.i64.resize(n);
            for (uint32_t k = 0; k < n; ++k)
              p.i64[k] = p.type == 'l' ? rd<int64_t>(data + 8 * k) : (p.type == 'i' ? (int64_t)rd<int32_t>(data + 4 * k) : (int64_t)data[k]);
          }
          break;
        }
        case 'S': case 'R': {
          if (!need(pos, 4)) return false;
          uint32_t n = rd<uint32_t>(buf + pos);
          pos += 4;
          if (!need(pos, n)) return false;
          p.str.assign((const char *)buf + pos, n);
          pos += n;
          break;
        }
        default:
          err = std::string("unknown FBX property type '") + p.type + "'";
          return false;
      }
      out.push_back(std::move(p));
    }
    return true;
  }

  // returns false on error; `node.name` empty + end == 0 marks the NULL record that closes a child list
  bool node(size_t &pos, FbxNode &out, bool &is_null, int depth) {
    if (depth > 64) {
      err = "FBX nesting too deep";
      return false;
    }
    const size_t hdr = wide ? 25 : 13;
    if (!need(pos, hdr)) return false;
    uint64_t end, nprops64, plen;
    if (wide) {
      end = rd<uint64_t>(buf + pos); nprops64 = rd<uint64_t>(buf + pos + 8); plen = rd<uint64_t>(buf + pos + 16);
    } else {
      end = rd<uint32_t>(buf + pos); nprops64 = rd<uint32_t>(buf + pos + 4); plen = rd<uint32_t>(buf + pos + 8);
    }
    uint8_t nlen = buf[pos + hdr - 1];
    if (end == 0) {
      is_null = true;
      pos += hdr;
      return true;
    }
    is_null = false;
    pos += hdr;
    if (nprops64 > size) {  // every property takes at least one byte
      err = "bad FBX record (property count)";
      return false;
    }
    const uint32_t nprops = (uint32_t)nprops64;
    if (!need(pos, nlen) || end > size || end < pos) {
      err = "bad FBX record offsets";
      return false;
    }
    out.name.assign((const char *)buf + pos, nlen);
    pos += nlen;
    size_t p0 = pos;
    if (!props(pos, nprops, out.props)) return false;
    if (pos != p0 + plen) {
      err = "FBX property list length mismatch";
      return false;
    }
    while (pos < end) {
      FbxNode child;
      bool null_rec = false;
      if (!node(pos, child, null_rec, depth + 1)) return false;
      if (null_rec) break;
      out.children.push_back(std::move(child));
    }
    pos = end;
    return true;
  }
};

struct Layer {
  std::string mapping, ref;
  const std::vector<double> *data = nullptr;
  const std::vector<int64_t> *index = nullptr;
};

bool get_layer(const FbxNode &geom, const char *name, const char *data_name, const char *index_name, Layer &out, std::string &err) {
  const FbxNode *el = geom.find(name);
  if (!el) {
    err = std::string("FBX geometry has no ") + name;
    return false;
  }
  const FbxNode *m = el->find("MappingInformationType"), *r = el->find("ReferenceInformationType"), *d = el->find(data_name);
  if (!m || !r || !d || m->props.empty() || r->props.empty() || d->props.empty()) {
    err = std::string(name) + ": incomplete layer element";
    return false;
  }
  out.mapping = m->props[0].str;
  out.ref = r->props[0].str;
  out.data = &d->props[0].f64;
  if (out.ref == "IndexToDirect") {
    const FbxNode *ix = el->find(index_name);
    if (!ix || ix->props.empty()) {
      err = std::string(name) + ": IndexToDirect without an index array";
      return false;
    }
    out.index = &ix->props[0].i64;
  }
  if (out.mapping != "ByPolygonVertex") {
    err = std::string(name) + ": mapping " + out.mapping + " not supported (ByPolygonVertex only)";
    return false;
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// PNG
// ------------------------------------------------------------------------------------------------

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

// Undo the scanline filters of one (sub)image of w x h pixels in place; rows are 1 filter byte + stride bytes.
bool unfilter(uint8_t *data, size_t avail, uint32_t w, uint32_t h, int channels, int depth, std::vector<uint8_t> &out) {
  const size_t stride = ((size_t)w * channels * depth + 7) / 8;
  const int bpp = std::max(1, channels * depth / 8);
  if (avail < (stride + 1) * (size_t)h) return false;
  out.assign(stride * h, 0);
  std::vector<uint8_t> zero(stride, 0);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t *src = data + (stride + 1) * (size_t)y;
    const int ft = src[0];
    ++src;
    uint8_t *cur = out.data() + stride * (size_t)y;
    const uint8_t *prev = y ? cur - stride : zero.data();
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
      int v;
      switch (ft) {
        case 0: v = src[i]; break;
        case 1: v = src[i] + a; break;
        case 2: v = src[i] + b; break;
        case 3: v = src[i] + ((a + b) >> 1); break;
        case 4: v = src[i] + paeth(a, b, c); break;
        default: return false;
      }
      cur[i] = (uint8_t)v;
    }
  }
  return true;
}

int decode_png(const uint8_t *bytes, size_t n, uint8_t **out_rgba, int32_t *out_w, int32_t *out_h) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (n < 8 || std::memcmp(bytes, sig, 8) != 0) return fail(BBA_ERR_FORMAT, "not a PNG file");
  uint32_t w = 0, h = 0;
  int depth = 0, color = -1, interlace = 0;
  std::vector<uint8_t> idat;
  uint8_t palette[256][4];
  uint32_t pal_len = 0;
  bool has_trans = false;
  uint16_t tc[3] = {0, 0, 0};
  for (int i = 0; i < 256; ++i) palette[i][0] = palette[i][1] = palette[i][2] = 0, palette[i][3] = 255;
  size_t pos = 8;
  bool seen_end = false;
  while (!seen_end && pos + 8 <= n) {
    const uint32_t len = be32(bytes + pos);
    const uint8_t *type = bytes + pos + 4, *data = bytes + pos + 8;
    if ((uint64_t)pos + 12 + len > n) return fail(BBA_ERR_FORMAT, "truncated PNG chunk");
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) return fail(BBA_ERR_FORMAT, "bad IHDR");
      w = be32(data); h = be32(data + 4);
      depth = data[8]; color = data[9]; interlace = data[12];
      if (!w || !h || w > (1u << 24) || h > (1u << 24) || (uint64_t)w * h > (1ull << 28)) return fail(BBA_ERR_FORMAT, "bad PNG size");
      if (data[10] != 0 || data[11] != 0 || interlace > 1) return fail(BBA_ERR_FORMAT, "bad PNG compression/filter/interlace method");
      const bool ok = (color == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                      (color == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                      ((color == 2 || color == 4 || color == 6) && (depth == 8 || depth == 16));
      if (!ok) return fail(BBA_ERR_FORMAT, "bad PNG colour type / bit depth");
    } else if (!std::memcmp(type, "PLTE", 4)) {
      if (len > 768 || len % 3) return fail(BBA_ERR_FORMAT, "bad PLTE");
      pal_len = len / 3;
      for (uint32_t i = 0; i < pal_len; ++i) palette[i][0] = data[3 * i], palette[i][1] = data[3 * i + 1], palette[i][2] = data[3 * i + 2];
    } else if (!std::memcmp(type, "tRNS", 4)) {
      if (color == 3) {
        if (len > pal_len) return fail(BBA_ERR_FORMAT, "bad tRNS length");
        for (uint32_t i = 0; i < len; ++i) palette[i][3] = data[i];
      } else if (color == 0 || color == 2) {
        const uint32_t nc = color == 0 ? 1 : 3;
        if (len != 2 * nc) return fail(BBA_ERR_FORMAT, "bad tRNS length");
        has_trans = true;
        for (uint32_t k = 0; k < nc; ++k) tc[k] = (uint16_t)((data[2 * k] << 8) | data[2 * k + 1]);
      } else {
        return fail(BBA_ERR_FORMAT, "tRNS with alpha");
      }
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      seen_end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (color < 0) return fail(BBA_ERR_FORMAT, "PNG without IHDR");
  if (idat.empty()) return fail(BBA_ERR_FORMAT, "PNG without IDAT");
  if (color == 3 && pal_len == 0) return fail(BBA_ERR_FORMAT, "paletted PNG without PLTE");
  const int channels = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4;

  // sub-images: the whole picture, or the seven Adam7 passes
  struct Pass { uint32_t x0, y0, dx, dy; };
  static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  static const Pass whole = {0, 0, 1, 1};
  const int n_pass = interlace ? 7 : 1;
  size_t raw_size = 0;
  for (int p = 0; p < n_pass; ++p) {
    const Pass &ps = interlace ? adam7[p] : whole;
    const uint32_t pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
    if (w <= ps.x0 || h <= ps.y0 || !pw || !ph) continue;
    raw_size += (((size_t)pw * channels * depth + 7) / 8 + 1) * ph;
  }
  std::vector<uint8_t> raw;
  if (!inflate_all(idat.data(), idat.size(), raw, raw_size) || raw.size() < raw_size) return fail(BBA_ERR_FORMAT, "PNG: corrupt IDAT stream");

  uint8_t *rgba = (uint8_t *)std::malloc((size_t)w * h * 4);
  if (!rgba) return fail(BBA_ERR_IO, "out of memory");
  // grey samples of fewer than 8 bits are scaled to 0..255 (x255, x85, x17); palette indices are not
  static const int scale_tab[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
  const int scale = color == 0 ? scale_tab[std::min(depth, 8)] : 1;
  size_t off = 0;
  std::vector<uint8_t> img;
  for (int p = 0; p < n_pass; ++p) {
    const Pass &ps = interlace ? adam7[p] : whole;
    if (w <= ps.x0 || h <= ps.y0) continue;
    const uint32_t pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
    if (!pw || !ph) continue;
    const size_t stride = ((size_t)pw * channels * depth + 7) / 8;
    if (!unfilter(raw.data() + off, raw.size() - off, pw, ph, channels, depth, img)) {
      std::free(rgba);
      return fail(BBA_ERR_FORMAT, "PNG: bad filter type");
    }
    off += (stride + 1) * ph;
    for (uint32_t y = 0; y < ph; ++y) {
      const uint8_t *row = img.data() + stride * y;
      for (uint32_t x = 0; x < pw; ++x) {
        uint16_t s[4] = {0, 0, 0, 0};  // samples at file precision
        for (int c = 0; c < channels; ++c) {
          const size_t k = (size_t)x * channels + c;
          if (depth == 16) s[c] = (uint16_t)((row[2 * k] << 8) | row[2 * k + 1]);
          else if (depth == 8) s[c] = row[k];
          else {
            const size_t bit = k * depth;
            s[c] = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
          }
        }
        uint8_t *o = rgba + 4 * ((size_t)(ps.y0 + y * ps.dy) * w + (ps.x0 + x * ps.dx));
        auto to8 = [&](uint16_t v) -> uint8_t { return depth == 16 ? (uint8_t)(v >> 8) : (uint8_t)(v * scale); };
        if (color == 3) {
          const uint8_t *pe = palette[s[0] < 256 ? s[0] : 0];  // out-of-range indices read entry 0 (never with valid files)
          o[0] = pe[0]; o[1] = pe[1]; o[2] = pe[2]; o[3] = pe[3];
        } else if (color == 0 || color == 4) {
          o[0] = o[1] = o[2] = to8(s[0]);
          o[3] = color == 4 ? to8(s[1]) : 255;
          // colour-key transparency compares at file precision for 16-bit samples, after scaling otherwise
          if (has_trans && (depth == 16 ? s[0] == tc[0] : o[0] == (uint8_t)((tc[0] & 255) * scale))) o[3] = 0;
        } else {
          o[0] = to8(s[0]); o[1] = to8(s[1]); o[2] = to8(s[2]);
          o[3] = color == 6 ? to8(s[3]) : 255;
          if (has_trans && (depth == 16 ? (s[0] == tc[0] && s[1] == tc[1] && s[2] == tc[2])
                                        : (o[0] == (uint8_t)(tc[0] & 255) && o[1] == (uint8_t)(tc[1] & 255) && o[2] == (uint8_t)(tc[2] & 255))))
            o[3] = 0;
        }
      }
    }
  }
  *out_rgba = rgba;
  *out_w = (int32_t)w;
  *out_h = (int32_t)h;
  return BBA_OK;
}

bool is_dir(const std::string &p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
bool is_file(const std::string &p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

std::vector<std::string> split_ws(const std::string &line) {
  std::vector<std::string> t;
  size_t i = 0;
  while (i < line.size()) {
    while (i < line.size() && (line[i] == ' ' || line[i] == '\t' || line[i] == '\r' || line[i] == '\n')) ++i;
    size_t j = i;
    while (j < line.size() && !(line[j] == ' ' || line[j] == '\t' || line[j] == '\r' || line[j] == '\n')) ++j;
    if (j > i) t.push_back(line.substr(i, j - i));
    i = j;
  }
  return t;
}

std::vector<std::string> lines_of(const std::vector<uint8_t> &bytes) {
  std::vector<std::string> out;
  std::string cur;
  for (uint8_t b : bytes) {
    if (b == '\n') {
      out.push_back(cur);
      cur.clear();
    } else {
      cur.push_back((char)b);
    }
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}

}  // namespace

extern "C" {

const char *bba_last_error(void) { return g_error.c_str(); }
void bba_free(void *p) { std::free(p); }

int bba_load_fbx_vertices(const char *path, void **out_vertices, uint32_t *out_n) {
  try {
    if (!path || !out_vertices || !out_n) return fail(BBA_ERR_ARGUMENT, "load_fbx_vertices: NULL argument");
    *out_vertices = nullptr;
    *out_n = 0;
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return fail(BBA_ERR_IO, std::string("cannot read ") + path);
    if (file.size() < 27 || std::memcmp(file.data(), "Kaydara FBX Binary  ", 20) != 0) return fail(BBA_ERR_FORMAT, "not a binary FBX file");
    const uint32_t version = rd<uint32_t>(file.data() + 23);
    FbxReader r{file.data(), file.size(), {}};
    r.wide = version >= 7500;
    FbxNode root;
    size_t pos = 27;
    while (pos < file.size()) {
      FbxNode nd;
      bool null_rec = false;
      if (!r.node(pos, nd, null_rec, 0)) return fail(BBA_ERR_FORMAT, r.err);
      if (null_rec) break;
      root.children.push_back(std::move(nd));
    }
    const FbxNode *objects = root.find("Objects");
    const FbxNode *geom = objects ? objects->find("Geometry") : nullptr;
    if (!geom) return fail(BBA_ERR_FORMAT, "FBX: no Objects/Geometry");
    const FbxNode *vn = geom->find("Vertices"), *pn = geom->find("PolygonVertexIndex");
    if (!vn || !pn || vn->props.empty() || pn->props.empty()) return fail(BBA_ERR_FORMAT, "FBX geometry without Vertices / PolygonVertexIndex");
    const std::vector<double> &ctrl = vn->props[0].f64;
    const std::vector<int64_t> &pvi = pn->props[0].i64;
    const size_t n_pv = pvi.size(), n_ctrl = ctrl.size() / 3;
    if (n_pv == 0) return fail(BBA_ERR_FORMAT, "FBX: empty PolygonVertexIndex");
    // Polygon-vertex k of every emitted triangle corner.  A polygon ends at the negative (bit-inverted) index; triangles
    // pass through, larger polygons are fanned from their first corner -- (0, i, i+1) -- which is what assimp's
    // aiProcess_Triangulate (src/scene.cpp:61) does for convex quads; ShaderBall.fbx has triangles only.
    std::vector<uint32_t> corner;
    corner.reserve(n_pv);
    for (size_t first = 0, k = 0; k < n_pv; ++k) {
      if (pvi[k] >= 0) continue;
      const size_t count = k - first + 1;
      if (count < 3) return fail(BBA_ERR_FORMAT, "FBX: polygon with fewer than three vertices");
      for (size_t i = 1; i + 1 < count; ++i) {
        corner.push_back((uint32_t)first);
        corner.push_back((uint32_t)(first + i));
        corner.push_back((uint32_t)(first + i + 1));
      }
      first = k + 1;
    }
    if (pvi[n_pv - 1] >= 0) return fail(BBA_ERR_FORMAT, "FBX: last polygon is not closed");
    const size_t n = corner.size();
    std::string err;
    Layer ln, lt, lu;
    if (!get_layer(*geom, "LayerElementNormal", "Normals", "NormalsIndex", ln, err) ||
        !get_layer(*geom, "LayerElementTangent", "Tangents", "TangentsIndex", lt, err) ||
        !get_layer(*geom, "LayerElementUV", "UV", "UVIndex", lu, err))
      return fail(BBA_ERR_UNSUPPORTED, err);
    auto fetch = [&](const Layer &l, size_t k, int width, float *dst) -> bool {
      size_t e = k;
      if (l.index) {
        if (k >= l.index->size() || (*l.index)[k] < 0) return false;
        e = (size_t)(*l.index)[k];
      }
      if ((e + 1) * width > l.data->size()) return false;
      for (int c = 0; c < width; ++c) dst[c] = (float)(*l.data)[e * width + c];
      return true;
    };
    float *v = (float *)std::malloc(n * 11 * sizeof(float));
    if (!v) return fail(BBA_ERR_IO, "out of memory");
    for (size_t t = 0; t < n; ++t) {
      const size_t k = corner[t];
      const int64_t raw = pvi[k];
      const size_t ci = (size_t)(raw < 0 ? ~raw : raw);
      float *o = v + 11 * t;
      if (ci >= n_ctrl || !fetch(lu, k, 2, o + 3) || !fetch(ln, k, 3, o + 5) || !fetch(lt, k, 3, o + 8)) {
        std::free(v);
        return fail(BBA_ERR_FORMAT, "FBX: index out of range");
      }
      o[0] = (float)ctrl[3 * ci]; o[1] = (float)ctrl[3 * ci + 1]; o[2] = (float)ctrl[3 * ci + 2];
    }
    *out_vertices = v;
    *out_n = (uint32_t)n;
    return BBA_OK;
  } catch (const std::exception &e) {
    return fail(BBA_ERR_IO, std::string("bba_load_fbx_vertices: ")+e.what());
  }
}

int bba_load_obj_gizmo(const char *path, void **out_vertices, uint32_t *out_nv, uint32_t **out_indices, uint32_t *out_ni) {
  try {
    if (!path || !out_vertices || !out_nv || !out_indices || !out_ni) return fail(BBA_ERR_ARGUMENT, "load_obj_gizmo: NULL argument");
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return fail(BBA_ERR_IO, std::string("cannot read ") + path);
    std::string dir(path);
    size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    std::vector<float> pos, nrm, verts;
    std::vector<uint32_t> idx;
    std::map<std::string, std::array<float, 3>> mats;
    std::array<float, 3> color = {1.f, 1.f, 1.f};
    for (const std::string &line : lines_of(file)) {
      std::vector<std::string> t = split_ws(line);
      if (t.empty() || t[0][0] == '#') continue;
      if (t[0] == "mtllib" && t.size() > 1) {
        std::vector<uint8_t> mtl;
        if (!read_file((dir + "/" + t[1]).c_str(), mtl)) return fail(BBA_ERR_IO, "cannot read material library " + t[1]);
        std::string cur;
        for (const std::string &ml : lines_of(mtl)) {
          std::vector<std::string> m = split_ws(ml);
          if (m.empty()) continue;
          if (m[0] == "newmtl" && m.size() > 1) {
            cur = m[1];
            mats[cur] = {1.f, 1.f, 1.f};
          } else if (m[0] == "Kd" && m.size() > 3 && !cur.empty()) {
            mats[cur] = {(float)std::atof(m[1].c_str()), (float)std::atof(m[2].c_str()), (float)std::atof(m[3].c_str())};
          }
        }
      } else if (t[0] == "v" && t.size() > 3) {
        for (int k = 1; k <= 3; ++k) pos.push_back((float)std::atof(t[k].c_str()));
      } else if (t[0] == "vn" && t.size() > 3) {
        for (int k = 1; k <= 3; ++k) nrm.push_back((float)std::atof(t[k].c_str()));
      } else if (t[0] == "usemtl" && t.size() > 1) {
        auto it = mats.find(t[1]);
        if (it == mats.end()) return fail(BBA_ERR_FORMAT, "usemtl of an unknown material: " + t[1]);
        color = it->second;
      } else if (t[0] == "f") {
        std::vector<uint32_t> corners;
        for (size_t k = 1; k < t.size(); ++k) {
          // v, v/vt, v//vn, v/vt/vn
          long vi = std::atol(t[k].c_str()), ni = 0;
          size_t s1 = t[k].find('/'), s2 = s1 == std::string::npos ? s1 : t[k].find('/', s1 + 1);
          if (s2 != std::string::npos && s2 + 1 < t[k].size()) ni = std::atol(t[k].c_str() + s2 + 1);
          const long np = (long)pos.size() / 3, nn = (long)nrm.size() / 3;
          vi = vi > 0 ? vi - 1 : np + vi;
          ni = ni > 0 ? ni - 1 : nn + ni;
          if (vi < 0 || vi >= np || (nn && (ni < 0 || ni >= nn))) return fail(BBA_ERR_FORMAT, "OBJ face index out of range");
          corners.push_back((uint32_t)(verts.size() / 9));
          for (int c = 0; c < 3; ++c) verts.push_back(pos[3 * vi + c]);
          for (int c = 0; c < 3; ++c) verts.push_back(color[c]);
          for (int c = 0; c < 3; ++c) verts.push_back(nn ? nrm[3 * ni + c] : 0.f);
        }
        for (size_t k = 1; k + 1 < corners.size(); ++k) {
          idx.push_back(corners[0]);
          idx.push_back(corners[k]);
          idx.push_back(corners[k + 1]);
        }
      }
    }
    float *v = (float *)std::malloc(std::max<size_t>(verts.size(), 1) * sizeof(float));
    uint32_t *ix = (uint32_t *)std::malloc(std::max<size_t>(idx.size(), 1) * sizeof(uint32_t));
    if (!v || !ix) {
      std::free(v);
      std::free(ix);
      return fail(BBA_ERR_IO, "out of memory");
    }
    if (!verts.empty()) std::memcpy(v, verts.data(), verts.size() * sizeof(float));
    if (!idx.empty()) std::memcpy(ix, idx.data(), idx.size() * sizeof(uint32_t));
    *out_vertices = v;
    *out_nv = (uint32_t)(verts.size() / 9);
    *out_indices = ix;
    *out_ni = (uint32_t)idx.size();
    return BBA_OK;
  } catch (const std::exception &e) {
    return fail(BBA_ERR_IO, std::string("bba_load_obj_gizmo: ")+e.what());
  }
}

int bba_decode_png(const uint8_t *bytes, uint64_t n, uint8_t **out_rgba, int32_t *out_w, int32_t *out_h) {
  try {
    if (!bytes || !out_rgba || !out_w || !out_h) return fail(BBA_ERR_ARGUMENT, "decode_png: NULL argument");
    *out_rgba = nullptr;
    return decode_png(bytes, (size_t)n, out_rgba, out_w, out_h);
  } catch (const std::exception &e) {
    return fail(BBA_ERR_IO, std::string("bba_decode_png: ")+e.what());
  }
}

int bba_load_png(const char *path, uint8_t **out_rgba, int32_t *out_w, int32_t *out_h) {
  try {
    if (!path || !out_rgba || !out_w || !out_h) return fail(BBA_ERR_ARGUMENT, "load_png: NULL argument");
    *out_rgba = nullptr;
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return fail(BBA_ERR_IO, std::string("cannot read ") + path);
    return decode_png(file.data(), file.size(), out_rgba, out_w, out_h);
  } catch (const std::exception &e) {
    return fail(BBA_ERR_IO, std::string("bba_load_png: ")+e.what());
  }
}

int bba_load_material_dir(bbr_context *ctx, const char *dir, int32_t *out_material) {
  try {
    if (!ctx || !dir || !out_material) return fail(BBA_ERR_ARGUMENT, "load_material_dir: NULL argument");
    // PBRMapType order: src/render.h:235-243
    static const char *names[BBR_MAP_COUNT] = {"albedo.png", "metallic.png", "roughness.png", "ao.png", "normal.png", "height.png"};
    bbr_image maps[BBR_MAP_COUNT];
    uint8_t *owned[BBR_MAP_COUNT] = {};
    int rc = BBA_OK;
    for (int i = 0; i < BBR_MAP_COUNT && rc == BBA_OK; ++i) {
      maps[i].rgba = nullptr;
      maps[i].width = maps[i].height = 0;
      const std::string p = std::string(dir) + "/" + names[i];
      if (!is_file(p)) continue;  // missing map: the default material's map (src/render.cpp:1328-1336)
      rc = bba_load_png(p.c_str(), &owned[i], &maps[i].width, &maps[i].height);
      maps[i].rgba = owned[i];
    }
    if (rc == BBA_OK) {
      int brc = bbr_upload_material(ctx, maps, out_material);
      if (brc != BBR_OK) rc = fail(BBA_ERR_IO, std::string("bbr_upload_material: ") + bbr_last_error(ctx));
    }
    for (uint8_t *p : owned) std::free(p);
    return rc;
  } catch (const std::exception &e) {
    return fail(BBA_ERR_IO, std::string("bba_load_material_dir: ")+e.what());
  }
}

int bba_load_material_set(bbr_context *ctx, const char *root, int32_t *out_materials, char (*out_names)[64], uint32_t capacity,
                          uint32_t *out_n) {
  try {
    if (!ctx || !root || !out_n) return fail(BBA_ERR_ARGUMENT, "load_material_set: NULL argument");
    *out_n = 0;
    DIR *d = opendir(root);
    if (!d) return fail(BBA_ERR_IO, std::string("cannot open directory ") + root);
    std::vector<std::string> dirs;
    while (dirent *e = readdir(d)) {
      const std::string name = e->d_name;
      if (name == "." || name == "..") continue;
      if (is_dir(std::string(root) + "/" + name)) dirs.push_back(name);
    }
    closedir(d);
    std::sort(dirs.begin(), dirs.end());  // FindFirstFile order on NTFS
    // "default" is swapped with the last entry and popped (src/render.cpp:1297-1306)
    for (size_t i = 0; i < dirs.size(); ++i)
      if (dirs[i] == "default") {
        std::swap(dirs[i], dirs.back());
        dirs.pop_back();
        break;
      }
    for (size_t i = 0; i < dirs.size(); ++i) {
      if (i >= capacity) break;
      int32_t id = -1;
      int rc = bba_load_material_dir(ctx, (std::string(root) + "/" + dirs[i]).c_str(), &id);
      if (rc != BBA_OK) return rc;
      if (out_materials) out_materials[i] = id;
      if (out_names) {
        std::memset(out_names[i], 0, 64);
        std::strncpy(out_names[i], dirs[i].c_str(), 63);
      }
      ++*out_n;
    }
    return BBA_OK;
  } catch (const std::exception &e) {
    return fail(BBA_ERR_IO, std::string("bba_load_material_set: ")+e.what());
  }
}

}  // extern "C"
