"""Host mirror of loader/data_loader.h over the C ABI (rtxn_load_images_json):
load_images_json / load_synthetic_data / load_data with the reference's names and
behaviour (loader/data_loader.cpp:34-149), returning numpy arrays instead of raw pointers."""
import ctypes as C
import sys

import numpy as np

from . import _lib

SYNTHETIC_NAMES = {"CHAIR": "chair/", "DRUMS": "drums/", "FICUS": "ficus/", "HOTDOG": "hotdog/", "LEGO": "lego/",
                   "MATERIALS": "fern/",   # sic: data_loader.cpp:128-130 (quirk Q12)
                   "MIC": "mic/", "SHIP": "ship/"}


class ImageDataset:
    """loader/data_loader.h:20-27: images float[n,H,W,3], poses float[n,16], focal, width, height, channels."""

    def __init__(self, images, poses, focal, width, height, channels, camera_angle_x=0.0):
        self.images, self.poses, self.focal = images, poses, focal
        self.image_width, self.image_height, self.image_channels = width, height, channels
        self.camera_angle_x = camera_angle_x


def load_images_json(basename, s, flags=0):
    """data_loader.cpp:34-94.  A missing JSON exits the process as the reference does (:36-39); a frame
    that fails to load returns an empty dataset (:74-78)."""
    d = _lib.ImageDataset()
    rc = _lib.lib().rtxn_load_images_json(str(basename).encode(), s.encode(), flags, C.byref(d))
    if rc != 0:
        msg = _lib.lib().rtxn_last_error().decode()
        print(msg, file=sys.stderr)
        if "transform JSON" in msg:
            sys.exit(1)
        return ImageDataset(np.zeros((0, 0, 0, 3), np.float32), np.zeros((0, 16), np.float32), 0.0, 0, 0, 0)
    n, w, h = d.n_images, d.image_width, d.image_height
    images = np.ctypeslib.as_array(d.images, shape=(n, h, w, 3)).copy() if n else np.zeros((0, h, w, 3), np.float32)
    poses = np.ctypeslib.as_array(d.poses, shape=(n, 16)).copy() if n else np.zeros((0, 16), np.float32)
    out = ImageDataset(images, poses, d.focal, w, h, d.image_channels, d.camera_angle_x)
    _lib.lib().rtxn_free_image_dataset(C.byref(d))
    return out


def load_synthetic_data(directory, flags=0):
    """data_loader.cpp:96-107: only the first split ("train") is loaded (the `break` at :103)."""
    datasets = []
    for split in ("train", "val", "test"):
        datasets.append(load_images_json(directory, split, flags))
        break
    return datasets


def load_llff_data(directory, factor=8, flags=0):
    """Fills the reference's LLFF stub (data_loader.cpp:140-142 sets the directory and returns nothing): poses_bounds.npy
    + the PNG frames of images_<factor>/ -> [ImageDataset] with `.bounds` float[n,2] (near, far); [] when the scene is
    absent or unreadable, which is what the reference returns for every LLFF request."""
    d = _lib.ImageDataset()
    b = C.POINTER(C.c_float)()
    rc = _lib.lib().rtxn_load_llff(str(directory).encode(), int(factor), flags, C.byref(d), C.byref(b))
    if rc != 0:
        print(_lib.lib().rtxn_last_error().decode(), file=sys.stderr)
        return []
    n, w, h = d.n_images, d.image_width, d.image_height
    images = np.ctypeslib.as_array(d.images, shape=(n, h, w, 3)).copy() if n else np.zeros((0, h, w, 3), np.float32)
    poses = np.ctypeslib.as_array(d.poses, shape=(n, 16)).copy() if n else np.zeros((0, 16), np.float32)
    out = ImageDataset(images, poses, d.focal, w, h, d.image_channels, d.camera_angle_x)
    out.bounds = np.ctypeslib.as_array(b, shape=(n, 2)).copy() if n else np.zeros((0, 2), np.float32)
    _lib.lib().rtxn_free_llff_bounds(b)
    _lib.lib().rtxn_free_image_dataset(C.byref(d))
    return [out]


def load_data(scene_type, name, root="./data", flags=0, llff_factor=8):
    """data_loader.cpp:109-149.  scene_type: "SYNTHETIC" | "LLFF".  The reference stops at the directory name for LLFF
    (:140-142) and returns []; here the scene is loaded when it exists (load_llff_data) and [] is returned otherwise."""
    filename = SYNTHETIC_NAMES[name]
    if scene_type == "SYNTHETIC":
        return load_synthetic_data(f"{root}/nerf_synthetic/{filename}", flags)
    return load_llff_data(f"{root}/nerf_llff_data/{filename}", llff_factor, flags)


def write_png(path, rgb_float):
    """Rendered frame float[H,W,3] in [0,1] -> 8-bit PNG (stb_image_write's unused role, main.cu:19-21)."""
    img = np.ascontiguousarray(np.rint(np.clip(rgb_float, 0.0, 1.0) * 255.0).astype(np.uint8))
    h, w = img.shape[:2]
    _lib.check(_lib.lib().rtxn_write_png_rgb8(str(path).encode(), img.ctypes.data_as(C.c_void_p), w, h), "rtxn_write_png_rgb8")
