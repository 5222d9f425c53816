// Drop-in for the reference's loader/data_loader.h (lines 1-32): same enums, struct and three
// free functions, implemented over rtxn_load_images_json (librtxn.so) instead of jsoncpp +
// stb_image.  Behaviour of loader/data_loader.cpp is kept: a missing transforms JSON prints and
// exit(1)s (:36-39); a frame that fails to load yields an EMPTY dataset (:74-78); only the
// "train" split is loaded (the `break` at :103); LLFF -- where the reference stops at the directory name and returns an
// empty vector (:140-148) -- loads poses_bounds.npy + images_8/*.png when the scene exists and returns empty otherwise;
// SyntheticName::MATERIALS maps to "fern/" (:128-130, quirk Q12); images are owned by the
// caller and never freed by the library (the reference leaks them too).
#ifndef DATA_LOADER_H
#define DATA_LOADER_H

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtxn.h"

enum class SceneType { LLFF, SYNTHETIC };
enum class SyntheticName {
  CHAIR,
  DRUMS,
  FICUS,
  HOTDOG,
  LEGO,
  MATERIALS,
  MIC,
  SHIP};

struct ImageDataset {
  std::vector<float*> images;
  std::vector<float*> poses;
  float focal;
  unsigned int image_width;
  unsigned int image_height;
  unsigned int image_channels;
};

inline ImageDataset load_images_json(std::string basename, std::string s) {
  rtxn_image_dataset d;
  int rc = rtxn_load_images_json(basename.c_str(), s.c_str(), 0, &d);
  ImageDataset dataset{};
  if (rc != RTXN_OK) {
    std::fprintf(stderr, "%s\n", rtxn_last_error());
    if (std::strstr(rtxn_last_error(), "transform JSON")) std::exit(1);  // data_loader.cpp:36-39
    return dataset;                                                         // :74-78
  }
  const size_t npx = (size_t)d.image_width * d.image_height * 3;
  for (int i = 0; i < d.n_images; ++i) {
    float* image = (float*)std::malloc(npx * sizeof(float));
    float* pose = new float[16];
    std::memcpy(image, d.images + (size_t)i * npx, npx * sizeof(float));
    std::memcpy(pose, d.poses + (size_t)i * 16, 16 * sizeof(float));
    dataset.images.push_back(image);
    dataset.poses.push_back(pose);
  }
  dataset.focal = d.focal;
  dataset.image_width = d.image_width;
  dataset.image_height = d.image_height;
  dataset.image_channels = d.image_channels;
  rtxn_free_image_dataset(&d);
  return dataset;
}

// Same observable behaviour as the reference's function of this name (data_loader.cpp:96-107): it names three splits and
// returns after the first, so the result holds exactly one dataset, the "train" split.
inline std::vector<ImageDataset> load_synthetic_data(std::string directory) {
  return std::vector<ImageDataset>(1, load_images_json(directory, "train"));
}

// Not in the reference (its LLFF branch is a stub): <directory>/poses_bounds.npy + <directory>/images_<factor>/*.png.
// bounds (optional): near/far per image, 2 floats each.
inline std::vector<ImageDataset> load_llff_data(std::string directory, int factor = 8, std::vector<float>* bounds = nullptr) {
  rtxn_image_dataset d;
  float* b = nullptr;
  std::vector<ImageDataset> datasets;
  if (rtxn_load_llff(directory.c_str(), factor, 0, &d, &b) != RTXN_OK) {
    std::fprintf(stderr, "%s\n", rtxn_last_error());
    return datasets;
  }
  ImageDataset dataset{};
  const size_t npx = (size_t)d.image_width * d.image_height * 3;
  for (int i = 0; i < d.n_images; ++i) {
    float* image = (float*)std::malloc(npx * sizeof(float));
    float* pose = new float[16];
    std::memcpy(image, d.images + (size_t)i * npx, npx * sizeof(float));
    std::memcpy(pose, d.poses + (size_t)i * 16, 16 * sizeof(float));
    dataset.images.push_back(image);
    dataset.poses.push_back(pose);
  }
  if (bounds) bounds->assign(b, b + 2 * (size_t)d.n_images);
  dataset.focal = d.focal;
  dataset.image_width = d.image_width;
  dataset.image_height = d.image_height;
  dataset.image_channels = d.image_channels;
  rtxn_free_llff_bounds(b);
  rtxn_free_image_dataset(&d);
  datasets.push_back(dataset);
  return datasets;
}

inline std::vector<ImageDataset> load_data(SceneType type, SyntheticName name) {
  std::string directory, filename;
  switch (name) {
    case SyntheticName::CHAIR: filename = "chair/"; break;
    case SyntheticName::DRUMS: filename = "drums/"; break;
    case SyntheticName::FICUS: filename = "ficus/"; break;
    case SyntheticName::HOTDOG: filename = "hotdog/"; break;
    case SyntheticName::LEGO: filename = "lego/"; break;
    case SyntheticName::MATERIALS: filename = "fern/"; break;  // sic, data_loader.cpp:128-130
    case SyntheticName::MIC: filename = "mic/"; break;
    case SyntheticName::SHIP: filename = "ship/"; break;
  }
  switch (type) {
    case SceneType::LLFF:
      directory = "./data/nerf_llff_data/" + filename;
      return load_llff_data(directory);   // the reference falls through to the empty vector here (:140-142,148)
    case SceneType::SYNTHETIC:
      directory = "./data/nerf_synthetic/" + filename;
      return load_synthetic_data(directory);
  }
  return std::vector<ImageDataset>();
}

#endif
