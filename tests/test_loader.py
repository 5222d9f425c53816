"""Loader (loader/data_loader.h drop-in): transforms JSON + PNG frames, checked against an
independent decode (Pillow) + stb_image's published ldr->hdr rule.  CPU only."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
from PIL import Image

from rtx_nerf_amd import loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_scene(tmp, n=3, w=7, h=5, mode="RGBA", split="train", seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(tmp / split, exist_ok=True)
    frames, imgs = [], []
    for i in range(n):
        ch = {"RGBA": 4, "RGB": 3, "L": 1, "LA": 2}[mode]
        arr = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        Image.fromarray(arr.squeeze() if ch == 1 else arr, mode).save(tmp / split / f"r_{i}.png")
        imgs.append(arr)
        m = rng.standard_normal((4, 4)).round(6)
        m[3] = [0, 0, 0, 1]
        frames.append({"file_path": f"./{split}/r_{i}", "rotation": 0.0125, "transform_matrix": m.tolist()})
    with open(tmp / f"transforms_{split}.json", "w") as f:
        json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, f, indent=2)
    return imgs, frames


def _stb_loadf(arr):
    """stbi_loadf(..., 3): alpha dropped / gray replicated, then (float)pow(v/255.0f, 2.2f) (stb_image.h v2.28)."""
    a = arr.astype(np.float32)
    if a.shape[2] in (1, 2):
        rgb = np.repeat(a[:, :, :1], 3, axis=2)
    else:
        rgb = a[:, :, :3]
    return np.power((rgb / np.float32(255.0)).astype(np.float64), np.float64(np.float32(2.2))).astype(np.float32)


@pytest.mark.parametrize("mode", ["RGBA", "RGB", "L", "LA"])
def test_load_images_json_matches_stb_semantics(tmp_path, mode):
    imgs, frames = _make_scene(tmp_path, mode=mode)
    ds = loader.load_images_json(str(tmp_path), "train")
    assert ds.images.shape == (3, 5, 7, 3) and ds.image_width == 7 and ds.image_height == 5 and ds.image_channels == 3
    for i in range(3):
        np.testing.assert_array_equal(ds.images[i], _stb_loadf(imgs[i]))               # Q11: alpha dropped, gamma 2.2
        np.testing.assert_array_equal(ds.poses[i], np.array(frames[i]["transform_matrix"], np.float32).reshape(16))
    # focal = .5 * 800 / tan(.5 * camera_angle_x), 800 hard-coded (data_loader.cpp:85)
    assert abs(ds.focal - 0.5 * 800 / np.tan(0.5 * np.float32(0.6911112070083618))) < 1e-2
    assert abs(ds.focal - 1111.111) < 0.01


def test_flags_composite_and_no_gamma(tmp_path):
    imgs, _ = _make_scene(tmp_path, n=1, mode="RGBA")
    a = imgs[0].astype(np.float32) / 255.0
    ds = loader.load_images_json(str(tmp_path), "train", flags=3)
    want = a[:, :, :3] * a[:, :, 3:] + (1 - a[:, :, 3:])
    np.testing.assert_allclose(ds.images[0], want, atol=1e-6)


def test_palette_and_filters(tmp_path):
    os.makedirs(tmp_path / "train")
    rng = np.random.default_rng(1)
    # a smooth gradient makes the encoder pick Sub/Up/Average/Paeth filters; a palette image exercises PLTE
    g = np.add.outer(np.arange(40), np.arange(64)).astype(np.uint8)
    rgb = np.stack([g, (g.astype(np.int32) * 2 % 256).astype(np.uint8), 255 - g], axis=2).astype(np.uint8)
    Image.fromarray(rgb, "RGB").save(tmp_path / "train" / "r_0.png", optimize=True)
    pal = Image.fromarray(rng.integers(0, 256, (40, 64, 3), dtype=np.uint8), "RGB").quantize(16)
    pal.save(tmp_path / "train" / "r_1.png")
    frames = [{"file_path": f"./train/r_{i}", "transform_matrix": np.eye(4).tolist()} for i in range(2)]
    json.dump({"camera_angle_x": 0.5, "frames": frames}, open(tmp_path / "transforms_train.json", "w"))
    ds = loader.load_images_json(str(tmp_path), "train", flags=2)
    np.testing.assert_allclose(ds.images[0], rgb.astype(np.float32) / 255.0, atol=1e-7)
    np.testing.assert_allclose(ds.images[1], np.asarray(pal.convert("RGB"), np.float32) / 255.0, atol=1e-7)


def test_missing_png_gives_empty_dataset_and_train_only(tmp_path):
    _make_scene(tmp_path, n=2)
    _make_scene(tmp_path, n=2, split="val", seed=5)
    sets = loader.load_synthetic_data(str(tmp_path))
    assert len(sets) == 1 and sets[0].images.shape[0] == 2          # `break` after train (data_loader.cpp:103)
    os.remove(tmp_path / "train" / "r_1.png")
    ds = loader.load_images_json(str(tmp_path), "train")
    assert ds.images.shape[0] == 0 and len(ds.poses) == 0            # data_loader.cpp:74-78
    assert loader.load_data("LLFF", "LEGO") == []                    # :140-148
    assert loader.SYNTHETIC_NAMES["MATERIALS"] == "fern/"            # :128-130


def test_missing_json_exits_like_the_reference(tmp_path):
    code = ("import sys; sys.path.insert(0, %r); from rtx_nerf_amd import loader; "
            "loader.load_images_json(%r, 'train')" % (ROOT, str(tmp_path / "nope")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to open transform JSON file" in r.stderr     # data_loader.cpp:36-39


def test_write_png_roundtrip(tmp_path):
    rng = np.random.default_rng(2)
    img = rng.uniform(-0.1, 1.1, (9, 13, 3)).astype(np.float32)
    loader.write_png(tmp_path / "out.png", img)
    back = np.asarray(Image.open(tmp_path / "out.png").convert("RGB"))
    np.testing.assert_array_equal(back, np.rint(np.clip(img, 0, 1) * 255).astype(np.uint8))


def test_cpp_dropin_header_compiles_and_loads(tmp_path):
    """A reference-style translation unit using loader/data_loader.h builds against the drop-in and returns the same data."""
    imgs, frames = _make_scene(tmp_path, n=2)
    src = tmp_path / "t.cpp"
    src.write_text('#include "data_loader.h"\n#include <cstdio>\nint main(int argc, char** argv) {\n'
                   '  ImageDataset d = load_images_json(argv[1], "train");\n'
                   '  std::printf("%zu %u %u %u %.4f %.9g %.9g\\n", d.images.size(), d.image_width, d.image_height, d.image_channels,\n'
                   '              d.focal, d.images[1][5], d.poses[1][7]);\n  return 0; }\n')
    exe = tmp_path / "t"
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["g++", "-std=c++17", f"-I{inc}", f"-I{inc}/rtxn_dropin/loader", str(src), "-o", str(exe),
                           f"-L{ROOT}/rtx_nerf_amd", "-lrtxn", f"-Wl,-rpath,{ROOT}/rtx_nerf_amd", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe), str(tmp_path)], text=True).split()
    assert out[:4] == ["2", "7", "5", "3"]
    want = _stb_loadf(imgs[1]).reshape(-1)[5]
    assert abs(float(out[5]) - want) < 1e-7
    assert abs(float(out[6]) - np.float32(frames[1]["transform_matrix"][1][3])) < 1e-6


# ---- pinned by the reference's own decoder -----------------------------------------------------------------------
# tests/golden/loader_stb.npz: PNG files + what the REFERENCE's vendored loader/stb_image.h returns for
# stbi_loadf(path, &w, &h, &n, 3) (loader/data_loader.cpp:63), generated in the build container by
# tests/golden/make_loader_golden.py through oracle/_ref/stb_loadf (the untouched reference header behind a main()).
_GOLD = np.load(os.path.join(ROOT, "tests", "golden", "loader_stb.npz"))


def _one_frame_scene(tmp, png_bytes):
    os.makedirs(tmp / "train", exist_ok=True)
    (tmp / "train" / "r_0.png").write_bytes(png_bytes)
    frames = [{"file_path": "./train/r_0", "transform_matrix": np.eye(4).tolist()}]
    json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, open(tmp / "transforms_train.json", "w"))


@pytest.mark.parametrize("name", [str(n) for n in _GOLD["names"]])
def test_png_decode_is_bit_exact_with_the_references_stb_image(tmp_path, name):
    ok, w, h, _ = (int(v) for v in _GOLD["hdr_" + name])
    _one_frame_scene(tmp_path, _GOLD["png_" + name].tobytes())
    ds = loader.load_images_json(str(tmp_path), "train")
    if not ok:                                   # stbi_loadf returned NULL -> the reference returns an empty dataset (:74-78)
        assert ds.images.shape[0] == 0
        return
    assert ds.images.shape == (1, h, w, 3), (name, ds.images.shape)
    want = _GOLD["out_" + name]
    got = ds.images[0]
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))
    assert bad[0].size == 0, f"{name}: {bad[0].size} of {want.size} values differ, e.g. got {got[bad][:4]} want {want[bad][:4]}"


# ---- LLFF: the stub at data_loader.cpp:140-148 filled in -----------------------------------------------------------
def _make_llff(tmp, n=4, w=12, h=9, factor=8, seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(tmp / f"images_{factor}", exist_ok=True)
    pb = np.zeros((n, 17))
    imgs = []
    for i in range(n):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        pb[i, :15] = np.concatenate([q, rng.standard_normal((3, 1)), np.array([[h * factor], [w * factor], [0.9 * w * factor]])], 1).reshape(-1)
        pb[i, 15:] = [1.2 + i, 9.5 + i]
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(arr, "RGB").save(tmp / f"images_{factor}" / f"image{(n - 1 - i):03d}.png")   # written out of order
        imgs.append(arr)
    np.save(tmp / "poses_bounds.npy", pb)
    return pb, imgs[::-1]   # frames are taken in NAME order


def test_llff_poses_bounds_and_frames(tmp_path):
    pb, imgs = _make_llff(tmp_path)
    sets = loader.load_llff_data(str(tmp_path), factor=8)
    assert len(sets) == 1
    ds = sets[0]
    assert ds.images.shape == (4, 9, 12, 3) and ds.image_width == 12 and ds.image_height == 9
    for i in range(4):
        np.testing.assert_array_equal(ds.images[i], _stb_loadf(imgs[i]))
        m = pb[i, :15].reshape(3, 5)
        want = np.eye(4)
        want[:3, :4] = np.concatenate([m[:, 1:2], -m[:, 0:1], m[:, 2:3], m[:, 3:4]], 1)   # (down,right,back) -> (right,up,back)
        np.testing.assert_array_equal(ds.poses[i].reshape(4, 4), want.astype(np.float32))
        c2w = ds.poses[i].reshape(4, 4)[:3, :3].astype(np.float64)
        assert abs(np.linalg.det(c2w) - 1.0) < 1e-5                                      # still a proper rotation
    np.testing.assert_array_equal(ds.bounds, pb[:, 15:].astype(np.float32))
    assert abs(ds.focal - 0.9 * 12) < 1e-5                                               # focal scaled to the loaded resolution
    assert abs(ds.camera_angle_x - 2 * np.arctan(0.5 * 12 / (0.9 * 12))) < 1e-6


def test_llff_errors_give_the_references_empty_vector(tmp_path, capfd):
    assert loader.load_llff_data(str(tmp_path / "absent")) == []                         # what the reference returns for LLFF
    assert "poses_bounds.npy" in capfd.readouterr().err
    _make_llff(tmp_path, n=3)
    os.remove(tmp_path / "images_8" / "image001.png")
    assert loader.load_llff_data(str(tmp_path)) == []
    assert "2 .png frames but poses_bounds.npy has 3 poses" in capfd.readouterr().err
    np.save(tmp_path / "poses_bounds.npy", np.zeros((3, 16)))
    assert loader.load_llff_data(str(tmp_path)) == []
    assert loader.load_data("LLFF", "LEGO", root=str(tmp_path / "nowhere")) == []


# ---- hostile / corrupt inputs must come back as errors, never as a crash of the host process -----------
def test_loader_survives_hostile_files(tmp_path, capfd):
    good = _GOLD["png_rgb8"].tobytes()
    huge = bytearray(good)
    huge[16:24] = (0x7fffffff).to_bytes(4, "big") * 2                                    # IHDR 2^31-1 x 2^31-1
    _one_frame_scene(tmp_path, bytes(huge))
    assert loader.load_images_json(str(tmp_path), "train").images.shape[0] == 0
    assert "too large" in capfd.readouterr().err
    _one_frame_scene(tmp_path, good)
    (tmp_path / "transforms_train.json").write_text('{"camera_angle_x": 0.69, "frames": ' + "[" * 100000)
    assert loader.load_images_json(str(tmp_path), "train").images.shape[0] == 0
    assert "nesting too deep" in capfd.readouterr().err
    (tmp_path / "transforms_train.json").write_text('{"camera_angle_x": 0.6911112')      # truncated inside a number
    assert loader.load_images_json(str(tmp_path), "train").images.shape[0] == 0
    (tmp_path / "transforms_train.json").write_text('{"camera_angle_x": tru')             # truncated inside a keyword
    assert loader.load_images_json(str(tmp_path), "train").images.shape[0] == 0


# tests/golden/loader_stb_fuzz.npz: 311 structurally valid PNGs with RANDOM content (tools/san/loader_fuzz.cpp --emit: every
# colour type / bit depth / interlace mode incl. illegal combinations, random filter bytes, PLTE / tRNS of random length,
# scanline data a little short or long, one file in ten damaged) and what the reference's stbi_loadf did with each --
# refused (174) or decoded (137, the floats).  make_loader_golden.py::fuzz_fixture wrote it through oracle/_ref/stb_loadf.
def test_png_accept_reject_and_values_follow_the_references_stb_image_on_random_files(tmp_path):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "loader_stb_fuzz.npz"))
    wrong = []
    n_ok = 0
    for name in (str(n) for n in gold["names"]):
        ok, w, h, _ = (int(v) for v in gold["hdr_" + name])
        _one_frame_scene(tmp_path, gold["png_" + name].tobytes())
        ds = loader.load_images_json(str(tmp_path), "train")
        if not ok:
            if ds.images.shape[0] != 0:
                wrong.append(f"{name}: the reference refuses this file, we decoded {ds.images.shape}")
            continue
        n_ok += 1
        if ds.images.shape != (1, h, w, 3):
            wrong.append(f"{name}: the reference decodes {w}x{h}, we returned {ds.images.shape}")
        elif not np.array_equal(ds.images[0].view(np.uint32), gold["out_" + name].view(np.uint32)):
            wrong.append(f"{name}: {int((ds.images[0] != gold['out_' + name]).sum())} values differ")
    assert n_ok > 100 and not wrong, wrong[:10]
