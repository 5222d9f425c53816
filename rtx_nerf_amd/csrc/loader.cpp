// Dataset loader: NeRF-synthetic transforms_<split>.json + PNG frames -> linear float RGB
// images and row-major 4x4 poses.  Replaces loader/data_loader.cpp of the reference
// (load_images_json :34-94, load_synthetic_data :96-107, load_data :109-149) and the two
// third-party pieces it leans on: jsoncpp 1.9.3 (only root["camera_angle_x"],
// frames[].file_path, frames[].transform_matrix are read, :47-71) and stb_image's
// stbi_loadf(path, &w, &h, &n, 3) (:63).  Both are re-implemented here from their published
// behaviour: a small recursive-descent JSON reader, and a PNG decoder (zlib inflate + the
// five PNG filters, plain and Adam7-interlaced, 1/2/4/8/16-bit, gray/gray-alpha/RGB/RGBA/palette) followed by
// stb's conversions: channels reduced to 3 by DROPPING alpha (no compositing, quirk Q11),
// then ldr->hdr  out = (float)pow(v/255.0f, 2.2f)  (stb_image.h v2.28 stbi__ldr_to_hdr).
// Host-only code; no GPU involvement.
#include <dirent.h>
#include <sys/stat.h>
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cctype>
#include <cstring>
#include <exception>
#include <new>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace {

// ------------------------------------------------------------------------- JSON
struct JVal {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  double num = 0;
  bool b = false;
  std::string str;
  std::vector<JVal> arr;
  std::vector<std::pair<std::string, JVal>> obj;
  const JVal* get(const char* key) const {
    for (const auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

struct JParser {
  const char* p;
  const char* end;      // the buffer is NUL-terminated AT end (strncmp/strtod below may look at it)
  std::string err;
  int depth = 0;        // nesting guard: a hostile file must not overflow the stack
  static constexpr int kMaxDepth = 64;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool fail(const char* m) { if (err.empty()) err = m; return false; }
  bool parse_string(std::string& out) {
    if (p >= end || *p != '"') return fail("expected string");
    ++p;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        ++p;
        if (p >= end) return fail("bad escape");
        switch (*p) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {
            if (end - p < 5) return fail("bad \\u escape");
            unsigned cp = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            p += 4;
            break;
          }
          default: out += *p;
        }
        ++p;
      } else {
        out += *p++;
      }
    }
    if (p >= end) return fail("unterminated string");
    ++p;
    return true;
  }
  bool parse(JVal& v) {
    if (depth >= kMaxDepth) return fail("nesting too deep");
    struct Scope { int& d; explicit Scope(int& x) : d(x) { ++d; } ~Scope() { --d; } } scope(depth);
    ws();
    if (p >= end) return fail("unexpected end");
    if (*p == '{') {
      v.kind = JVal::Obj;
      ++p;
      ws();
      if (p < end && *p == '}') { ++p; return true; }
      for (;;) {
        ws();
        std::string k;
        if (!parse_string(k)) return false;
        ws();
        if (p >= end || *p != ':') return fail("expected ':'");
        ++p;
        JVal child;
        if (!parse(child)) return false;
        v.obj.emplace_back(std::move(k), std::move(child));
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == '}') { ++p; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (*p == '[') {
      v.kind = JVal::Arr;
      ++p;
      ws();
      if (p < end && *p == ']') { ++p; return true; }
      for (;;) {
        JVal child;
        if (!parse(child)) return false;
        v.arr.push_back(std::move(child));
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == ']') { ++p; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (*p == '"') { v.kind = JVal::Str; return parse_string(v.str); }
    if (end - p >= 4 && !strncmp(p, "true", 4)) { v.kind = JVal::Bool; v.b = true; p += 4; return true; }
    if (end - p >= 5 && !strncmp(p, "false", 5)) { v.kind = JVal::Bool; v.b = false; p += 5; return true; }
    if (end - p >= 4 && !strncmp(p, "null", 4)) { v.kind = JVal::Null; p += 4; return true; }
    char* q = nullptr;
    v.num = strtod(p, &q);                     // stops at the terminating NUL at the latest
    if (q == p || q > end) return fail("bad token");
    v.kind = JVal::Num;
    p = q;
    return true;
  }
};

// Regular files only, and of a size a frame or a pose table can have: fopen() also opens a directory (a corrupted
// file_path can name one), for which ftell() reports LONG_MAX.
constexpr off_t kMaxFileBytes = (off_t)1 << 30;
bool read_file(const std::string& path, std::vector<unsigned char>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  struct stat st;
  if (fstat(fileno(f), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 0 || st.st_size > kMaxFileBytes) {
    fclose(f);
    return false;
  }
  const size_t n = (size_t)st.st_size;
  out.resize(n);
  const size_t got = n ? fread(out.data(), 1, n, f) : 0;
  fclose(f);
  return got == n;
}

// ------------------------------------------------------------------------- PNG
inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c) {
  int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

// The zlib stream of the IDAT chunks, inflated the way stb_image's own inflater treats it: the two header bytes are checked
// (multiple of 31, method 8, no preset dictionary), the deflate blocks are decoded to the end of the final block, and the
// Adler-32 that follows is NOT read -- a file with a damaged checksum loads in the reference, so it loads here.  Output beyond
// `keep` bytes is produced (an error late in the stream still rejects the file) but not stored; `total` is its full length.
bool inflate_like_stb(const std::vector<unsigned char>& z, bool zlib_header, size_t keep, std::vector<unsigned char>& out, size_t& total,
                      std::string& err) {
  size_t off = 0;
  if (zlib_header) {
    if (z.size() < 2) { err = "bad zlib header"; return false; }
    const int cmf = z[0], flg = z[1];
    if ((cmf * 256 + flg) % 31 != 0) { err = "bad zlib header"; return false; }
    if (flg & 32) { err = "preset dictionary"; return false; }
    if ((cmf & 15) != 8) { err = "bad compression"; return false; }
    off = 2;
  }
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, -15) != Z_OK) { err = "inflate init"; return false; }
  out.assign(keep, 0);
  unsigned char spill[16384];
  zs.next_in = const_cast<unsigned char*>(z.data()) + off;
  size_t in_left = z.size() - off;
  total = 0;
  int rc = Z_OK;
  while (rc != Z_STREAM_END) {
    if (zs.avail_in == 0 && in_left) {
      zs.avail_in = (uInt)std::min<size_t>(in_left, 1u << 30);
      in_left -= zs.avail_in;
    }
    const bool stored = total < keep;
    const size_t room = stored ? std::min<size_t>(keep - total, 1u << 30) : sizeof(spill);
    zs.next_out = stored ? out.data() + total : spill;
    zs.avail_out = (uInt)room;
    rc = inflate(&zs, Z_NO_FLUSH);
    total += room - zs.avail_out;
    if (rc == Z_STREAM_END) break;
    if (rc != Z_OK || (zs.avail_in == 0 && in_left == 0 && zs.avail_out != 0)) {   // corrupt, or the data ends inside a block
      inflateEnd(&zs);
      err = "inflate failed";
      return false;
    }
  }
  inflateEnd(&zs);
  return true;
}

// Decodes to 8 bits per channel, `channels` in {1,2,3,4} (palette expanded to RGB or RGBA; a tRNS colour key becomes an alpha
// channel).  WHICH files load and which do not follows the reference's decoder, stb_image.h v2.28 (stbi__parse_png_file and
// stbi__create_png_image_raw, restated): the first chunk must be IHDR (after an optional CgBI), one IHDR only, PLTE at most
// 256 whole entries, tRNS before the first IDAT / not longer than the palette / exactly one 16-bit value per colour channel /
// never with an alpha colour type, unknown critical chunks refuse the file and ancillary ones are skipped, chunk CRCs and the
// zlib checksum are not verified, scanline data may be longer than the image needs but not shorter, and bit depths below 8
// are unpacked for any colour type (scaled to 0..255 for grey only) -- tests/golden/loader_stb_fuzz.npz holds what the
// reference's header returned for a few hundred random files of every such kind.
bool decode_png(const std::vector<unsigned char>& file, int& width, int& height, int& channels,
                std::vector<unsigned char>& pixels, std::string& err) {
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (file.size() < 8 || memcmp(file.data(), sig, 8)) { err = "not a PNG"; return false; }
  size_t pos = 8;
  int depth = 0, ctype = -1, interlace = 0, src_ch = 0;
  std::vector<unsigned char> idat, trns;
  unsigned char plte[256 * 3] = {0};
  size_t pal_len = 0;
  bool first = true, done = false, iphone = false, idat_seen = false, pal_alpha = false, has_key = false;
  unsigned key[3] = {0, 0, 0};
  while (!done) {
    // a chunk header past the end of the file reads as zeros in the reference: length 0, type 0 = an unknown critical chunk
    if (pos + 8 > file.size()) { err = "truncated file"; return false; }
    const uint32_t len = be32(&file[pos]);
    const unsigned char* type = &file[pos + 4];
    const size_t body = pos + 8;
    const auto is = [&](const char* t) { return !memcmp(type, t, 4); };
    if (is("IEND")) {                                       // its length and CRC are not looked at
      if (first) { err = "first chunk is not IHDR"; return false; }
      done = true;
      break;
    }
    if (body + (size_t)len > file.size()) { err = "truncated chunk"; return false; }
    const unsigned char* data = file.data() + body;
    if (is("CgBI")) {
      iphone = true;                                        // Apple's variant: a bare deflate stream (channel order is left alone)
    } else if (is("IHDR")) {
      if (!first) { err = "multiple IHDR"; return false; }
      if (len != 13) { err = "bad IHDR"; return false; }
      const uint32_t w = be32(data), h = be32(data + 4);
      // stb_image refuses sides beyond STBI_MAX_DIMENSIONS = 2^24 and images whose samples exceed 2^30
      if (w > (1u << 24) || h > (1u << 24)) { err = "image too large"; return false; }
      depth = data[8]; ctype = data[9]; interlace = data[12];
      if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) { err = "unsupported bit depth"; return false; }
      if (ctype > 6 || (ctype == 3 && depth == 16) || (ctype != 3 && (ctype & 1))) { err = "bad colour type"; return false; }
      if (data[10] != 0) { err = "bad compression method"; return false; }
      if (data[11] != 0) { err = "bad filter method"; return false; }
      if (interlace > 1) { err = "bad interlace method"; return false; }
      if (w == 0 || h == 0) { err = "0-pixel image"; return false; }
      src_ch = ctype == 3 ? 1 : ((ctype & 2) ? 3 : 1) + ((ctype & 4) ? 1 : 0);
      if ((1u << 30) / w / (uint32_t)(ctype == 3 ? 4 : src_ch) < h) { err = "image too large"; return false; }
      width = (int)w; height = (int)h;
    } else if (first) {
      err = "first chunk is not IHDR";
      return false;
    } else if (is("PLTE")) {
      if (len > 256 * 3 || len % 3) { err = "invalid PLTE"; return false; }
      pal_len = len / 3;
      memcpy(plte, data, len);
    } else if (is("tRNS")) {
      if (idat_seen) { err = "tRNS after IDAT"; return false; }
      if (ctype == 3) {
        if (pal_len == 0) { err = "tRNS before PLTE"; return false; }
        if (len > pal_len) { err = "bad tRNS length"; return false; }
        pal_alpha = true;
        trns.assign(data, data + len);
      } else {
        if (!(src_ch & 1)) { err = "tRNS with alpha"; return false; }
        if (len != (uint32_t)src_ch * 2) { err = "bad tRNS length"; return false; }
        has_key = true;
        for (int k = 0; k < src_ch; ++k) {
          const unsigned v = ((unsigned)data[2 * k] << 8) | data[2 * k + 1];
          // 16-bit images compare all 16 bits; the others the low byte, scaled by 255 / (2^depth - 1) and kept to 8 bits as stb does
          key[k] = depth == 16 ? v : ((v & 255) * (255u / ((1u << depth) - 1))) & 255u;
        }
      }
    } else if (is("IDAT")) {
      if (ctype == 3 && pal_len == 0) { err = "no PLTE"; return false; }
      if (len > (1u << 30) || idat.size() + (size_t)len > ((size_t)1 << 31)) { err = "IDAT too large"; return false; }
      idat.insert(idat.end(), data, data + len);
      if (len) idat_seen = true;
    } else if (!(type[0] & 0x20)) {                         // an unknown CRITICAL chunk (upper-case first letter)
      err = "unknown critical chunk";
      return false;
    }
    first = false;
    pos = body + (size_t)len + 4;                           // the CRC is skipped, not checked
  }
  if (!idat_seen) { err = "no IDAT"; return false; }
  const size_t bpp_bits = (size_t)src_ch * depth;
  const size_t fbpp = depth < 8 ? 1 : bpp_bits / 8;         // filter byte distance
  // Adam7 (interlace method 1): seven reduced images, each filtered as an image of its own and scattered to
  // (x0 + i*dx, y0 + j*dy) -- stb_image.h's stbi__create_png_image does the same
  static const int kX0[7] = {0, 4, 0, 2, 0, 1, 0}, kY0[7] = {0, 0, 4, 0, 2, 0, 1};
  static const int kDx[7] = {8, 8, 4, 4, 2, 2, 1}, kDy[7] = {8, 8, 8, 4, 4, 2, 2};
  const int n_pass = interlace ? 7 : 1;
  size_t raw_size = 0;
  for (int ps = 0; ps < n_pass; ++ps) {
    const int pw = interlace ? (width - kX0[ps] + kDx[ps] - 1) / kDx[ps] : width;
    const int ph = interlace ? (height - kY0[ps] + kDy[ps] - 1) / kDy[ps] : height;
    if (pw > 0 && ph > 0) raw_size += (((size_t)pw * bpp_bits + 7) / 8 + 1) * (size_t)ph;
  }
  std::vector<unsigned char> raw;
  size_t raw_total = 0;
  if (!inflate_like_stb(idat, !iphone, raw_size, raw, raw_total, err)) return false;
  if (raw_total < raw_size) { err = "not enough pixels"; return false; }
  const size_t npx = (size_t)width * height;
  std::vector<unsigned char> s8(npx * src_ch), lo8;          // 8-bit samples of the whole image (16-bit: high bytes; low bytes beside
  if (depth == 16 && has_key) lo8.resize(npx * src_ch);      // them where a colour key has to be compared)
  const unsigned maxv = (1u << depth) - 1;
  size_t raw_pos = 0;
  std::vector<unsigned char> img;
  for (int ps = 0; ps < n_pass; ++ps) {
    const int x0 = interlace ? kX0[ps] : 0, y0 = interlace ? kY0[ps] : 0, dx = interlace ? kDx[ps] : 1, dy = interlace ? kDy[ps] : 1;
    const int pw = (width - x0 + dx - 1) / dx, ph = (height - y0 + dy - 1) / dy;
    if (pw <= 0 || ph <= 0) continue;
    const size_t stride = ((size_t)pw * bpp_bits + 7) / 8;
    // below 8 bits the reference unpacks a row in place inside a buffer of one byte per sample position and refuses rows that
    // would not fit: more than 8 bits per pixel (RGB / RGBA at depth 4) never load
    if (depth < 8 && stride > (size_t)pw) { err = "invalid width"; return false; }
    img.assign(stride * (size_t)ph, 0);
    for (int y = 0; y < ph; ++y) {
      const unsigned char* in = &raw[raw_pos + (stride + 1) * (size_t)y];
      const int ft = in[0];
      ++in;
      unsigned char* out = &img[stride * (size_t)y];
      const unsigned char* up = y ? out - stride : nullptr;
      if (ft > 4) { err = "bad filter"; return false; }
      for (size_t x = 0; x < stride; ++x) {
        const int a = x >= fbpp ? out[x - fbpp] : 0, b = up ? up[x] : 0, c = (up && x >= fbpp) ? up[x - fbpp] : 0;
        int v = in[x];
        switch (ft) {
          case 0: break;
          case 1: v += a; break;
          case 2: v += b; break;
          case 3: v += (a + b) >> 1; break;
          default: v += paeth(a, b, c); break;
        }
        out[x] = (unsigned char)v;
      }
    }
    raw_pos += (stride + 1) * (size_t)ph;
    // to 8-bit samples, scattered to the pass's pixel positions
    for (int y = 0; y < ph; ++y) {
      const unsigned char* row = &img[stride * (size_t)y];
      for (int x = 0; x < pw; ++x) {
        const size_t at = ((size_t)(y0 + y * dy) * width + (size_t)(x0 + x * dx)) * src_ch;
        unsigned char* dst = &s8[at];
        if (depth == 8) {
          for (int k = 0; k < src_ch; ++k) dst[k] = row[(size_t)x * src_ch + k];
        } else if (depth == 16) {
          for (int k = 0; k < src_ch; ++k) {
            dst[k] = row[2 * ((size_t)x * src_ch + k)];     // stb: the high byte
            if (!lo8.empty()) lo8[at + k] = row[2 * ((size_t)x * src_ch + k) + 1];
          }
        } else {
          for (int k = 0; k < src_ch; ++k) {                // samples follow one another bit by bit, whatever the colour type
            const size_t bit = ((size_t)x * src_ch + k) * depth;
            const unsigned v = (row[bit / 8] >> (8 - depth - (bit % 8))) & maxv;
            dst[k] = (unsigned char)(ctype == 0 ? v * 255 / maxv : v);
          }
        }
      }
    }
  }
  if (ctype == 3) {
    channels = pal_alpha ? 4 : 3;
    pixels.resize(npx * channels);
    for (size_t i = 0; i < npx; ++i) {
      const size_t idx = s8[i];                             // an index beyond the palette: black, opaque (the reference reads
      for (int k = 0; k < 3; ++k) pixels[i * channels + k] = idx < pal_len ? plte[idx * 3 + k] : 0;   // uninitialised memory)
      if (channels == 4) pixels[i * 4 + 3] = idx < trns.size() ? trns[idx] : 255;
    }
  } else if (has_key) {
    channels = src_ch + 1;
    pixels.resize(npx * channels);
    for (size_t i = 0; i < npx; ++i) {
      bool match = true;
      for (int k = 0; k < src_ch; ++k) {
        const unsigned v = depth == 16 ? ((unsigned)s8[i * src_ch + k] << 8) | lo8[i * src_ch + k] : s8[i * src_ch + k];
        match = match && v == key[k];
        pixels[i * channels + k] = s8[i * src_ch + k];
      }
      pixels[i * channels + src_ch] = match ? 0 : 255;
    }
  } else {
    channels = src_ch;
    pixels.swap(s8);
  }
  return true;
}

// stbi_loadf(..., desired_channels = 3): convert to 3 channels (alpha dropped, gray replicated), then
// ldr -> hdr with gamma 2.2 (flags bit 1 set: keep v/255; bit 0 set: composite alpha over white first).
void to_float_rgb(const std::vector<unsigned char>& px, int channels, size_t npx, int flags, float* out) {
  for (size_t i = 0; i < npx; ++i) {
    float rgb[3];
    const unsigned char* p = &px[i * channels];
    if (channels >= 3) { rgb[0] = p[0]; rgb[1] = p[1]; rgb[2] = p[2]; }
    else { rgb[0] = rgb[1] = rgb[2] = p[0]; }
    const bool has_alpha = channels == 2 || channels == 4;
    const float alpha = has_alpha ? p[channels - 1] / 255.0f : 1.0f;
    for (int k = 0; k < 3; ++k) {
      float v = rgb[k] / 255.0f;
      if (!(flags & 2)) v = (float)(pow(v, 2.2f) * 1.0f);
      if ((flags & 1) && has_alpha) v = v * alpha + (1.0f - alpha);
      out[i * 3 + k] = v;
    }
  }
}

bool all_finite(const std::vector<float>& v) {
  for (float x : v)
    if (!std::isfinite(x)) return false;
  return true;
}

}  // namespace

// ============================================================================ C ABI
static int load_images_json_impl(const char* basename, const char* split, int flags, rtxn_image_dataset* out);

// No C++ exception may cross the C ABI (a bad_alloc from a hostile or corrupt file would otherwise terminate the host).
extern "C" int rtxn_load_images_json(const char* basename, const char* split, int flags, rtxn_image_dataset* out) {
  RTXN_REQUIRE(basename && split && out, "rtxn_load_images_json: NULL argument");
  memset(out, 0, sizeof(*out));
  try {
    return load_images_json_impl(basename, split, flags, out);
  } catch (const std::exception& e) {
    rtxn::set_error("Failed to load the image set %s (%s): %s", basename, split, e.what());
  } catch (...) {
    rtxn::set_error("Failed to load the image set %s (%s)", basename, split);
  }
  free(out->images);
  free(out->poses);
  memset(out, 0, sizeof(*out));
  return RTXN_ERR_IO;
}

static int load_images_json_impl(const char* basename, const char* split, int flags, rtxn_image_dataset* out) {
  const std::string base(basename);
  const std::string json_path = base + "/transforms_" + split + ".json";
  std::vector<unsigned char> text;
  if (!read_file(json_path, text)) {
    rtxn::set_error("Failed to open transform JSON file: %s", json_path.c_str());
    return RTXN_ERR_IO;
  }
  JVal root;
  text.push_back(0);   // NUL terminator: the number/keyword scanners never read past the buffer
  JParser jp{reinterpret_cast<const char*>(text.data()), reinterpret_cast<const char*>(text.data()) + text.size() - 1, {}};
  if (!jp.parse(root) || root.kind != JVal::Obj) {
    rtxn::set_error("%s: JSON parse error: %s", json_path.c_str(), jp.err.c_str());
    return RTXN_ERR_IO;
  }
  const JVal* cam = root.get("camera_angle_x");
  const JVal* frames = root.get("frames");
  if (!cam || cam->kind != JVal::Num || !frames || frames->kind != JVal::Arr) {
    rtxn::set_error("%s: missing camera_angle_x / frames", json_path.c_str());
    return RTXN_ERR_IO;
  }
  const float camera_angle_x = (float)cam->num;
  const size_t n = frames->arr.size();
  int W = 0, H = 0;
  std::vector<float> images, poses(n * 16);
  for (size_t i = 0; i < n; ++i) {
    const JVal& fr = frames->arr[i];
    const JVal* fp = fr.get("file_path");
    const JVal* tm = fr.get("transform_matrix");
    if (!fp || fp->kind != JVal::Str || !tm || tm->kind != JVal::Arr || tm->arr.size() != 4) {
      rtxn::set_error("%s: frame %zu lacks file_path / transform_matrix", json_path.c_str(), i);
      return RTXN_ERR_IO;
    }
    for (int r = 0; r < 4; ++r) {
      if (tm->arr[r].kind != JVal::Arr || tm->arr[r].arr.size() != 4) {
        rtxn::set_error("%s: frame %zu transform_matrix is not 4x4", json_path.c_str(), i);
        return RTXN_ERR_IO;
      }
      for (int c = 0; c < 4; ++c) poses[i * 16 + r * 4 + c] = (float)tm->arr[r].arr[c].num;  // data_loader.cpp:66-71
    }
    const std::string png_path = base + "/" + fp->str + ".png";                               // :61
    std::vector<unsigned char> file, px;
    int w = 0, h = 0, ch = 0;
    std::string err;
    if (!read_file(png_path, file) || !decode_png(file, w, h, ch, px, err)) {
      rtxn::set_error("Failed to load the image %s%s%s", png_path.c_str(), err.empty() ? "" : ": ", err.c_str());
      return RTXN_ERR_IO;  // the reference returns an EMPTY dataset here (:74-78); the C++ wrapper mirrors that
    }
    if (i == 0) { W = w; H = h; images.resize(n * (size_t)W * H * 3); }
    if (w != W || h != H) {
      rtxn::set_error("%s: %dx%d differs from the first frame's %dx%d", png_path.c_str(), w, h, W, H);
      return RTXN_ERR_IO;
    }
    to_float_rgb(px, ch, (size_t)W * H, flags, &images[i * (size_t)W * H * 3]);
  }
  // a pose or field of view that is not a number (a JSON reader accepts "nan" / "1e999") would only surface as garbage rays
  if (!all_finite(poses) || !std::isfinite(camera_angle_x)) {
    rtxn::set_error("%s: camera_angle_x or a transform_matrix entry is not finite", json_path.c_str());
    return RTXN_ERR_IO;
  }
  out->n_images = (int)n;
  out->image_width = (unsigned)W;
  out->image_height = (unsigned)H;
  out->image_channels = 3;                                                                    // desired_channels, :52
  out->focal = (float)(.5 * 800 / std::tan(.5 * camera_angle_x));                             // :85 (800 hard-coded, Q12)
  out->camera_angle_x = camera_angle_x;
  out->images = (float*)malloc(images.size() * sizeof(float));
  out->poses = (float*)malloc(poses.size() * sizeof(float));
  if ((!out->images && !images.empty()) || (!out->poses && !poses.empty())) {
    free(out->images); free(out->poses);
    memset(out, 0, sizeof(*out));
    rtxn::set_error("rtxn_load_images_json: out of memory");
    return RTXN_ERR_IO;
  }
  if (!images.empty()) memcpy(out->images, images.data(), images.size() * sizeof(float));
  if (!poses.empty()) memcpy(out->poses, poses.data(), poses.size() * sizeof(float));
  return RTXN_OK;
}

// ------------------------------------------------------------------------- LLFF (forward-facing) scenes
// The reference declares SceneType::LLFF and stops at the directory name (loader/data_loader.cpp:140-142: load_data
// returns an empty vector).  This fills that stub with the published LLFF layout (Mildenhall et al., "Local Light Field
// Fusion"; NeRF's load_llff.py): <dir>/poses_bounds.npy = float64[N][17], per image a row-major 3x5 matrix
// [R | t | (H, W, focal)] in LLFF's (down, right, backwards) camera axes followed by the near/far depth bounds, and the
// frames themselves in <dir>/images_<factor>/ (PNG; the full-size images/ directory of the published scenes is JPEG,
// which the reference's stbi path would read but this PNG-only decoder does not).  Output poses use the same convention
// as the synthetic loader: row-major 4x4 camera-to-world, axes (right, up, backwards).
namespace {

bool parse_npy_f64_2d(const std::vector<unsigned char>& f, size_t& rows, size_t& cols, std::vector<double>& data, std::string& err) {
  if (f.size() < 10 || memcmp(f.data(), "\x93NUMPY", 6)) { err = "not an .npy file"; return false; }
  const int major = f[6];
  size_t hlen, hoff;
  if (major == 1) { hlen = f[8] | (f[9] << 8); hoff = 10; }
  else if (major == 2 || major == 3) {
    if (f.size() < 12) { err = "truncated .npy header"; return false; }
    hlen = (size_t)f[8] | ((size_t)f[9] << 8) | ((size_t)f[10] << 16) | ((size_t)f[11] << 24); hoff = 12;
  } else { err = "unsupported .npy version"; return false; }
  if (hoff + hlen > f.size()) { err = "truncated .npy header"; return false; }
  const std::string hdr(reinterpret_cast<const char*>(f.data()) + hoff, hlen);
  if (hdr.find("'<f8'") == std::string::npos && hdr.find("\"<f8\"") == std::string::npos) { err = ".npy dtype is not <f8"; return false; }
  if (hdr.find("'fortran_order': False") == std::string::npos) { err = ".npy is Fortran-ordered"; return false; }
  const size_t sp = hdr.find("'shape':");
  const size_t lp = sp == std::string::npos ? sp : hdr.find('(', sp);
  if (lp == std::string::npos) { err = ".npy header lacks a shape"; return false; }
  unsigned long long r = 0, c = 0;
  if (sscanf(hdr.c_str() + lp, "(%llu, %llu", &r, &c) != 2) { err = ".npy array is not 2-D"; return false; }
  if (r > (1ull << 24) || c > 4096 || (hoff + hlen + r * c * 8ull) > f.size()) { err = ".npy data shorter than its shape"; return false; }
  rows = (size_t)r; cols = (size_t)c;
  // copied out: numpy pads the header so that the data is 64-byte aligned, but nothing obliges a file to (a double read
  // through a pointer into the byte buffer was a misaligned load then -- found by the UBSan run of tools/san/loader_fuzz.cpp)
  data.resize(rows * cols);
  if (!data.empty()) memcpy(data.data(), f.data() + hoff + hlen, data.size() * sizeof(double));
  return true;
}

int load_llff_impl(const char* basedir, int factor, int flags, rtxn_image_dataset* out, float** bounds_out) {
  const std::string base(basedir);
  std::vector<unsigned char> npy;
  if (!read_file(base + "/poses_bounds.npy", npy)) {
    rtxn::set_error("Failed to open %s/poses_bounds.npy", basedir);
    return RTXN_ERR_IO;
  }
  size_t n = 0, cols = 0;
  std::vector<double> pbv;
  std::string err;
  const bool parsed = parse_npy_f64_2d(npy, n, cols, pbv, err);
  const double* pb = pbv.data();
  if (!parsed || cols != 17) {
    rtxn::set_error("%s/poses_bounds.npy: %s", basedir, err.empty() ? "expected float64[N][17]" : err.c_str());
    return RTXN_ERR_IO;
  }
  // frames: the .png files of images_<factor>/ (images/ for factor <= 1) in lexicographic order, as load_llff.py sorts them
  const std::string img_dir = base + (factor > 1 ? "/images_" + std::to_string(factor) : std::string("/images"));
  std::vector<std::string> files;
  if (DIR* d = opendir(img_dir.c_str())) {
    while (dirent* e = readdir(d)) {
      const std::string nm(e->d_name);
      if (nm.size() > 4) {
        std::string ext = nm.substr(nm.size() - 4);
        for (auto& ch : ext) ch = (char)tolower((unsigned char)ch);
        if (ext == ".png") files.push_back(nm);
      }
    }
    closedir(d);
  }
  std::sort(files.begin(), files.end());
  if (files.size() != n) {
    rtxn::set_error("%s holds %zu .png frames but poses_bounds.npy has %zu poses", img_dir.c_str(), files.size(), n);
    return RTXN_ERR_IO;
  }
  int W = 0, H = 0;
  std::vector<float> images, poses(n * 16), bounds(n * 2);
  for (size_t i = 0; i < n; ++i) {
    const double* row = pb + i * 17;   // row-major 3x5 then near, far
    // LLFF camera axes (down, right, back) -> (right, up, back): columns [1, -0, 2]; column 3 = camera position
    for (int r = 0; r < 3; ++r) {
      poses[i * 16 + r * 4 + 0] = (float)row[r * 5 + 1];
      poses[i * 16 + r * 4 + 1] = (float)-row[r * 5 + 0];
      poses[i * 16 + r * 4 + 2] = (float)row[r * 5 + 2];
      poses[i * 16 + r * 4 + 3] = (float)row[r * 5 + 3];
    }
    poses[i * 16 + 12] = poses[i * 16 + 13] = poses[i * 16 + 14] = 0.0f;
    poses[i * 16 + 15] = 1.0f;
    bounds[2 * i] = (float)row[15];
    bounds[2 * i + 1] = (float)row[16];
    std::vector<unsigned char> file, px;
    int w = 0, h = 0, ch = 0;
    const std::string png_path = img_dir + "/" + files[i];
    if (!read_file(png_path, file) || !decode_png(file, w, h, ch, px, err)) {
      rtxn::set_error("Failed to load the image %s%s%s", png_path.c_str(), err.empty() ? "" : ": ", err.c_str());
      return RTXN_ERR_IO;
    }
    if (i == 0) { W = w; H = h; images.resize(n * (size_t)W * H * 3); }
    if (w != W || h != H) {
      rtxn::set_error("%s: %dx%d differs from the first frame's %dx%d", png_path.c_str(), w, h, W, H);
      return RTXN_ERR_IO;
    }
    to_float_rgb(px, ch, (size_t)W * H, flags, &images[i * (size_t)W * H * 3]);
  }
  const double f_full = n ? pb[14] : 0.0, w_full = n ? pb[9] : 0.0;   // hwf column of the first pose: H = [4], W = [9], focal = [14]
  if (!all_finite(poses) || !all_finite(bounds) || !std::isfinite(f_full) || !std::isfinite(w_full)) {
    rtxn::set_error("%s/poses_bounds.npy: an entry is not finite", basedir);
    return RTXN_ERR_IO;
  }
  out->n_images = (int)n;
  out->image_width = (unsigned)W;
  out->image_height = (unsigned)H;
  out->image_channels = 3;
  out->focal = (float)(w_full > 0 ? f_full * (double)W / w_full : f_full);   // focal in pixels of the LOADED resolution
  out->camera_angle_x = out->focal > 0 ? (float)(2.0 * std::atan(0.5 * W / out->focal)) : 0.0f;
  out->images = (float*)malloc(images.size() * sizeof(float) + 1);
  out->poses = (float*)malloc(poses.size() * sizeof(float) + 1);
  float* b = (float*)malloc(bounds.size() * sizeof(float) + 1);
  if (!out->images || !out->poses || !b) {
    free(b);
    throw std::bad_alloc();
  }
  memcpy(out->images, images.data(), images.size() * sizeof(float));
  memcpy(out->poses, poses.data(), poses.size() * sizeof(float));
  memcpy(b, bounds.data(), bounds.size() * sizeof(float));
  if (bounds_out) *bounds_out = b; else free(b);
  return RTXN_OK;
}

}  // namespace

extern "C" int rtxn_load_llff(const char* basedir, int factor, int flags, rtxn_image_dataset* out, float** bounds) {
  RTXN_REQUIRE(basedir && out, "rtxn_load_llff: NULL argument");
  RTXN_REQUIRE(factor >= 0 && factor <= 64, "rtxn_load_llff: factor = %d", factor);
  memset(out, 0, sizeof(*out));
  if (bounds) *bounds = nullptr;
  try {
    const int rc = load_llff_impl(basedir, factor, flags, out, bounds);
    if (rc == RTXN_OK) return rc;
  } catch (const std::exception& e) {
    rtxn::set_error("Failed to load the LLFF scene %s: %s", basedir, e.what());
  } catch (...) {
    rtxn::set_error("Failed to load the LLFF scene %s", basedir);
  }
  free(out->images);
  free(out->poses);
  memset(out, 0, sizeof(*out));
  return RTXN_ERR_IO;
}

extern "C" void rtxn_free_llff_bounds(float* bounds) { free(bounds); }

extern "C" void rtxn_free_image_dataset(rtxn_image_dataset* d) {
  if (!d) return;
  free(d->images);
  free(d->poses);
  memset(d, 0, sizeof(*d));
}

// stb_image_write's role (included but never called by the reference, main.cu:19-21): 8-bit RGB PNG.
extern "C" int rtxn_write_png_rgb8(const char* path, const unsigned char* rgb, int width, int height) {
  RTXN_REQUIRE(path && rgb && width > 0 && height > 0, "rtxn_write_png_rgb8: bad argument");
  const size_t stride = (size_t)width * 3;
  std::vector<unsigned char> raw((stride + 1) * (size_t)height);
  for (int y = 0; y < height; ++y) {
    raw[(stride + 1) * (size_t)y] = 0;  // filter: none
    memcpy(&raw[(stride + 1) * (size_t)y + 1], rgb + stride * (size_t)y, stride);
  }
  uLongf clen = compressBound((uLong)raw.size());
  std::vector<unsigned char> comp(clen);
  if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) {
    rtxn::set_error("rtxn_write_png_rgb8: deflate failed");
    return RTXN_ERR_IO;
  }
  FILE* f = fopen(path, "wb");
  if (!f) { rtxn::set_error("rtxn_write_png_rgb8: cannot open %s", path); return RTXN_ERR_IO; }
  auto put32 = [](unsigned char* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
  auto chunk = [&](const char* type, const unsigned char* data, uint32_t len) {
    unsigned char hdr[8];
    put32(hdr, len);
    memcpy(hdr + 4, type, 4);
    fwrite(hdr, 1, 8, f);
    if (len) fwrite(data, 1, len, f);
    uLong crc = crc32(0L, reinterpret_cast<const Bytef*>(type), 4);
    if (len) crc = crc32(crc, data, len);
    unsigned char c[4];
    put32(c, (uint32_t)crc);
    fwrite(c, 1, 4, f);
  };
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  fwrite(sig, 1, 8, f);
  unsigned char ihdr[13];
  put32(ihdr, (uint32_t)width);
  put32(ihdr + 4, (uint32_t)height);
  ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", comp.data(), (uint32_t)clen);
  chunk("IEND", nullptr, 0);
  fclose(f);
  return RTXN_OK;
}
