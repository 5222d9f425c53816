// oracle/_ref/stb_loadf -- the ONE piece of the reference that compiles in this image untouched:
// /root/reference/loader/stb_image.h (v2.28), the decoder behind load_images_json's
// stbi_loadf(path, &w, &h, &n, 3) (loader/data_loader.cpp:63).  This file is only a main():
// the reference header is #included from where it lies (-I/root/reference/loader, see
// oracle/Makefile `ref`), nothing of it is copied, and no stand-in headers are involved.
//
// Compiled as C++ after <cmath> and <math.h>, exactly as the reference does (main.cu:1-20 defines
// STB_IMAGE_IMPLEMENTATION in a C++ translation unit that has both included): stb's
// `pow(v/255.0f, stbi__l2h_gamma)` (stb_image.h:1867) then resolves to the float overload, which
// differs from C's double pow in the last ulp of some values.
//
// TEST INFRASTRUCTURE (fixture generation only): tests/golden/make_loader_golden.py runs it in the
// build container; only the fixtures it writes travel to the GPU box.
//
// usage: stb_loadf <out.bin> <png>...   -> per file: int32 {ok, w, h, channels_in_file}, then w*h*3 floats if ok
#include <cmath>
#include <math.h>
#include <cstdio>
#include <cstdint>

#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s out.bin png...\n", argv[0]); return 2; }
  FILE* f = std::fopen(argv[1], "wb");
  if (!f) return 2;
  for (int i = 2; i < argc; ++i) {
    int w = 0, h = 0, n = 0;
    float* img = stbi_loadf(argv[i], &w, &h, &n, 3);   // data_loader.cpp:63, desired_channels = 3 (:52)
    const int32_t hdr[4] = {img ? 1 : 0, w, h, n};
    std::fwrite(hdr, sizeof(int32_t), 4, f);
    if (img) {
      std::fwrite(img, sizeof(float), (size_t)w * h * 3, f);
      stbi_image_free(img);
    } else {
      std::fprintf(stderr, "%s: %s\n", argv[i], stbi_failure_reason());
    }
  }
  std::fclose(f);
  return 0;
}
