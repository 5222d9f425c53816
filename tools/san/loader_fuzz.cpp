// AddressSanitizer / UBSan harness for the dataset loader (host-only code of librtxn.so: rtx_nerf_amd/csrc/loader.cpp --
// JSON reader, PNG decoder, .npy reader).  Built by tests/test_loader_sanitized.py with
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=undefined  loader.cpp loader_fuzz.cpp -lz
// (GPU sanitizers are not available on the pool; this is the CPU build the loader can be checked in).  It writes a small valid
// NeRF-synthetic scene and a small LLFF scene into <dir>, loads them, then loads `iters` corrupted copies: bytes flipped, runs
// overwritten, files truncated or extended, chunk lengths and header fields set to extreme values.  Every load must return --
// RTXN_OK or an error code -- without a sanitizer report; a load that succeeds must hand back buffers of the advertised size
// (they are read end to end).   loader_fuzz <dir> <iters> <seed>
// `--emit <dir> <count> <seed>` only writes the structured random PNGs: tests/golden/make_loader_golden.py decodes them with the
// reference's stb_image.h for the differential fixtures.
#include <sys/stat.h>
#include <zlib.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtxn.h"

// librtxn's error sink lives in common.hip beside a kernel; the harness links its own
namespace rtxn {
static char g_err[512];
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace rtxn

namespace {
uint64_t g_state;
uint64_t rnd() {   // xorshift64*
  g_state ^= g_state >> 12;
  g_state ^= g_state << 25;
  g_state ^= g_state >> 27;
  return g_state * 2685821657736338717ull;
}
std::vector<unsigned char> slurp(const std::string& p) {
  std::vector<unsigned char> v;
  if (FILE* f = fopen(p.c_str(), "rb")) {
    unsigned char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
  }
  return v;
}
void spit(const std::string& p, const std::vector<unsigned char>& v) {
  FILE* f = fopen(p.c_str(), "wb");
  if (!f) { perror(p.c_str()); exit(2); }
  if (!v.empty()) fwrite(v.data(), 1, v.size(), f);
  fclose(f);
}
void mutate(std::vector<unsigned char>& v) {
  const int kind = (int)(rnd() % 8);
  if (v.empty()) return;
  switch (kind) {
    case 0: for (int k = 0, n = 1 + (int)(rnd() % 4); k < n; ++k) v[rnd() % v.size()] ^= (unsigned char)(1u << (rnd() % 8)); break;
    case 1: for (int k = 0, n = 1 + (int)(rnd() % 8); k < n; ++k) v[rnd() % v.size()] = (unsigned char)rnd(); break;
    case 2: v.resize(rnd() % v.size()); break;                                                    // truncated
    case 3: { const size_t a = rnd() % v.size(), n = 1 + rnd() % 64;                              // a run of one value
              const unsigned char c = (rnd() & 1) ? 0xff : 0x00;
              for (size_t i = a; i < v.size() && i < a + n; ++i) v[i] = c; } break;
    case 4: { const size_t a = rnd() % v.size();                                                  // a 32-bit field set to an extreme
              static const uint32_t ext[] = {0u, 1u, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0x00ffffffu, 65536u};
              const uint32_t x = ext[rnd() % 7];
              for (int b = 0; b < 4 && a + b < v.size(); ++b) v[a + b] = (unsigned char)(x >> (8 * (3 - b))); } break;
    case 5: for (int k = 0, n = (int)(rnd() % 300); k < n; ++k) v.push_back((unsigned char)rnd()); break;   // trailing bytes
    case 6: { const size_t a = rnd() % v.size(), b = rnd() % v.size();                             // a slice duplicated elsewhere
              const size_t n = 1 + rnd() % 32;
              for (size_t i = 0; i < n && a + i < v.size() && b + i < v.size(); ++i) v[b + i] = v[a + i]; } break;
    default: { const size_t a = rnd() % v.size(); v.erase(v.begin() + a, v.begin() + a + std::min<size_t>(1 + rnd() % 16, v.size() - a)); } break;
  }
}
void put32(std::vector<unsigned char>& v, uint32_t x) { for (int b = 3; b >= 0; --b) v.push_back((unsigned char)(x >> (8 * b))); }
void chunk(std::vector<unsigned char>& png, const char* type, const std::vector<unsigned char>& body) {
  put32(png, (uint32_t)body.size());
  const size_t at = png.size();
  png.insert(png.end(), type, type + 4);
  png.insert(png.end(), body.begin(), body.end());
  put32(png, (uint32_t)crc32(0, png.data() + at, (uInt)(png.size() - at)));
}
// A structurally valid PNG around RANDOM content: every colour type / bit depth / interlace mode (sometimes an invalid
// combination), random filter bytes (0-4, sometimes beyond), a palette and tRNS of random length, and scanline data that is
// the expected length or a little off -- what byte flips of one RGB8 file never reach past the zlib checksum.
std::vector<unsigned char> random_png(int& w, int& h) {
  static const int types[] = {0, 2, 3, 4, 6}, chans[] = {1, 0, 3, 1, 2, 0, 4};
  static const int depths[] = {1, 2, 4, 8, 16};
  w = 1 + (int)(rnd() % 20);
  h = 1 + (int)(rnd() % 20);
  const int ct = (rnd() % 40 == 0) ? (int)(rnd() % 8) : types[rnd() % 5];
  int depth = depths[rnd() % 5];
  if (rnd() % 8) {                                        // mostly a legal depth for the colour type
    if (ct == 2 || ct == 4 || ct == 6) depth = (rnd() & 1) ? 8 : 16;
    if (ct == 3 && depth == 16) depth = 8;
  }
  const int interlace = (rnd() % 30 == 0) ? 2 : (int)(rnd() & 1);
  std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'}, ihdr;
  put32(ihdr, (uint32_t)w);
  put32(ihdr, (uint32_t)h);
  ihdr.push_back((unsigned char)depth);
  ihdr.push_back((unsigned char)ct);
  ihdr.push_back(0);
  ihdr.push_back(0);
  ihdr.push_back((unsigned char)interlace);
  chunk(png, "IHDR", ihdr);
  if (ct == 3 || rnd() % 10 == 0) {
    std::vector<unsigned char> pl((size_t)(rnd() % 5 ? 3 * (1 + rnd() % 256) : rnd() % 800));
    for (auto& c : pl) c = (unsigned char)rnd();
    chunk(png, "PLTE", pl);
  }
  if (rnd() % 3 == 0) {
    std::vector<unsigned char> tr((size_t)(rnd() % 300));
    for (auto& c : tr) c = (unsigned char)rnd();
    chunk(png, "tRNS", tr);
  }
  const int ch = ct < 7 ? chans[ct] : 1;
  // scanlines pass by pass: a filter byte (legal on most files) + random sample bytes
  std::vector<unsigned char> raw;
  const bool legal_filters = rnd() % 5 != 0;
  auto add_pass = [&](int pw, int ph) {
    if (pw <= 0 || ph <= 0) return;
    const size_t row = ((size_t)pw * ch * depth + 7) / 8;
    for (int y = 0; y < ph; ++y) {
      raw.push_back((unsigned char)(legal_filters ? rnd() % 5 : rnd()));
      for (size_t i = 0; i < row; ++i) raw.push_back((unsigned char)rnd());
    }
  };
  if (interlace == 1) {
    static const int x0[] = {0, 4, 0, 2, 0, 1, 0}, y0[] = {0, 0, 4, 0, 2, 0, 1}, dx[] = {8, 8, 4, 4, 2, 2, 1}, dy[] = {8, 8, 8, 4, 4, 2, 2};
    for (int p = 0; p < 7; ++p) add_pass((w - x0[p] + dx[p] - 1) / dx[p], (h - y0[p] + dy[p] - 1) / dy[p]);
  } else add_pass(w, h);
  if (rnd() % 8 == 0) raw.resize((size_t)std::max<long>(0, (long)raw.size() + (long)(rnd() % 9) - 4));   // a little short or long
  const size_t raw_n = raw.size();
  uLongf zn = compressBound((uLong)raw_n);
  std::vector<unsigned char> z(zn);
  compress(z.data(), &zn, raw.data(), (uLong)raw_n);
  z.resize(zn);
  if (rnd() % 5 == 0 && z.size() > 8) {                   // split over two IDAT chunks
    const size_t cut = 1 + rnd() % (z.size() - 1);
    chunk(png, "IDAT", std::vector<unsigned char>(z.begin(), z.begin() + cut));
    chunk(png, "IDAT", std::vector<unsigned char>(z.begin() + cut, z.end()));
  } else chunk(png, "IDAT", z);
  chunk(png, "IEND", {});
  return png;
}
double touch(const rtxn_image_dataset& d) {   // read everything a successful load advertises
  double s = 0;
  const size_t px = (size_t)d.image_width * d.image_height * 3;
  for (size_t i = 0; i < (size_t)d.n_images * px; ++i) s += d.images[i];
  for (size_t i = 0; i < (size_t)d.n_images * 16; ++i) s += d.poses[i];
  return s;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc >= 5 && !strcmp(argv[1], "--emit")) {          // loader_fuzz --emit <dir> <count> <seed>: random_png() files only
    g_state = 0x9E3779B97F4A7C15ull ^ (uint64_t)atoll(argv[4]);
    for (int i = 0, n = atoi(argv[3]); i < n; ++i) {
      int w, h;
      std::vector<unsigned char> png = random_png(w, h);
      if (rnd() % 10 == 0 || getenv("LOADER_FUZZ_MUTATE_ALL")) mutate(png);   // the variable: every file damaged once
      char name[64];
      snprintf(name, sizeof(name), "/r%05d.png", i);
      spit(std::string(argv[2]) + name, png);
    }
    return 0;
  }
  if (argc < 4) { fprintf(stderr, "usage: loader_fuzz <dir> <iters> <seed> | --emit <dir> <count> <seed>\n"); return 2; }
  const std::string dir = argv[1];
  const int iters = atoi(argv[2]);
  g_state = 0x9E3779B97F4A7C15ull ^ (uint64_t)atoll(argv[3]);
  // ---- a valid synthetic scene: 3 frames of 9x7 and transforms_train.json
  const std::string syn = dir + "/syn", llff = dir + "/llff";
  mkdir(syn.c_str(), 0755);
  mkdir((syn + "/train").c_str(), 0755);
  mkdir(llff.c_str(), 0755);
  mkdir((llff + "/images").c_str(), 0755);
  const int W = 9, H = 7, N = 3;
  std::string json = "{\n \"camera_angle_x\": 0.6911,\n \"frames\": [\n";
  for (int i = 0; i < N; ++i) {
    std::vector<unsigned char> rgb((size_t)W * H * 3);
    for (auto& c : rgb) c = (unsigned char)rnd();
    char name[64];
    snprintf(name, sizeof(name), "/train/r_%d.png", i);
    if (rtxn_write_png_rgb8((syn + name).c_str(), rgb.data(), W, H) != RTXN_OK) { fprintf(stderr, "write failed: %s\n", rtxn::g_err); return 2; }
    snprintf(name, sizeof(name), "/images/f%02d.png", i);
    if (rtxn_write_png_rgb8((llff + name).c_str(), rgb.data(), W, H) != RTXN_OK) return 2;
    char buf[512];
    snprintf(buf, sizeof(buf), "  {\"file_path\": \"./train/r_%d\", \"rotation\": 0.01, \"transform_matrix\": [[1,0,0,%d],[0,1,0,0.5],[0,0,1,-2e0],[0,0,0,1]]}%s\n",
             i, i, i + 1 < N ? "," : "");
    json += buf;
  }
  json += " ]\n}\n";
  spit(syn + "/transforms_train.json", std::vector<unsigned char>(json.begin(), json.end()));
  {   // poses_bounds.npy: float64[N][17]
    std::string hdr = "{'descr': '<f8', 'fortran_order': False, 'shape': (3, 17), }";
    while ((10 + hdr.size() + 1) % 64) hdr += ' ';
    hdr += '\n';
    std::vector<unsigned char> npy = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(hdr.size() & 255), (unsigned char)(hdr.size() >> 8)};
    npy.insert(npy.end(), hdr.begin(), hdr.end());
    for (int i = 0; i < N; ++i)
      for (int k = 0; k < 17; ++k) {
        double v = k == 4 ? H : k == 9 ? W : k == 14 ? 11.5 : k == 15 ? 0.7 : k == 16 ? 9.0 : ((k % 5) == (k / 5) ? 1.0 : 0.1 * i);
        unsigned char b[8];
        memcpy(b, &v, 8);
        npy.insert(npy.end(), b, b + 8);
      }
    spit(llff + "/poses_bounds.npy", npy);
  }
  rtxn_image_dataset d;
  if (rtxn_load_images_json(syn.c_str(), "train", 0, &d) != RTXN_OK || d.n_images != N || d.image_width != (unsigned)W) {
    fprintf(stderr, "the valid synthetic scene does not load: %s\n", rtxn::g_err);
    return 2;
  }
  double sink = touch(d);
  rtxn_free_image_dataset(&d);
  float* bounds = nullptr;
  if (rtxn_load_llff(llff.c_str(), 1, 0, &d, &bounds) != RTXN_OK || d.n_images != N) {
    fprintf(stderr, "the valid LLFF scene does not load: %s\n", rtxn::g_err);
    return 2;
  }
  sink += touch(d) + bounds[2 * N - 1];
  rtxn_free_image_dataset(&d);
  rtxn_free_llff_bounds(bounds);
  // ---- corrupted copies
  const std::string victims[] = {syn + "/transforms_train.json", syn + "/train/r_1.png", llff + "/poses_bounds.npy", llff + "/images/f01.png"};
  int ok = 0, rejected = 0;
  for (int it = 0; it < iters; ++it) {
    int which = (int)(rnd() % 6);
    const bool structured = which >= 4;                   // half of the PNG cases: random_png() in place of every frame
    if (structured) which = which == 4 ? 1 : 3;
    const std::vector<unsigned char> orig = slurp(victims[which]);
    std::vector<unsigned char> bad = orig;
    std::vector<std::vector<unsigned char>> saved;
    if (structured) {
      int pw, ph;
      bad = random_png(pw, ph);
      if (rnd() % 4 == 0) mutate(bad);
      for (int i = 0; i < N; ++i) {                       // all frames of the scene get the same file: sizes agree
        char name[64];
        snprintf(name, sizeof(name), which == 1 ? "/train/r_%d.png" : "/images/f%02d.png", i);
        const std::string path = (which == 1 ? syn : llff) + name;
        saved.push_back(slurp(path));
        spit(path, bad);
      }
    } else {
      for (int k = 0, n = 1 + (int)(rnd() % 3); k < n; ++k) mutate(bad);
      spit(victims[which], bad);
    }
    int rc;
    bounds = nullptr;
    if (which < 2) rc = rtxn_load_images_json(syn.c_str(), "train", (int)(rnd() % 4), &d);
    else rc = rtxn_load_llff(llff.c_str(), 1, (int)(rnd() % 4), &d, (rnd() & 1) ? &bounds : nullptr);
    if (rc == RTXN_OK) {
      sink += touch(d);
      if (bounds) sink += bounds[2 * d.n_images - 1];
      ++ok;
      rtxn_free_image_dataset(&d);
      rtxn_free_llff_bounds(bounds);
    } else {
      ++rejected;
      if (d.images || d.poses || bounds) { fprintf(stderr, "iteration %d: a failed load left buffers behind\n", it); return 1; }
    }
    if (structured) {
      for (int i = 0; i < N; ++i) {
        char name[64];
        snprintf(name, sizeof(name), which == 1 ? "/train/r_%d.png" : "/images/f%02d.png", i);
        spit((which == 1 ? syn : llff) + name, saved[i]);
      }
    } else spit(victims[which], orig);
  }
  printf("loader_fuzz: %d corrupted loads, %d accepted, %d rejected, no sanitizer report (checksum %g)\n", iters, ok, rejected, sink);
  return 0;
}
