#!/usr/bin/env python3
"""End-to-end use of the harness on a NeRF-synthetic scene directory (transforms_train.json + PNG frames, the format the
reference's loader expects at ./data/nerf_synthetic/<scene>, loader/data_loader.cpp:144): load -> device ray dataset ->
train (hash-grid or frequency model) with periodic occupancy refresh -> held-out PSNR -> PNG of a rendered view.
No dataset ships with this image; pass --make-demo to first write a small procedural scene in the same format.

  python examples/train_synthetic.py --data /path/to/lego [--steps 2000] [--encoding hash]
  python examples/train_synthetic.py --make-demo /tmp/demo_scene --data /tmp/demo_scene --steps 400
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

from rtx_nerf_amd import loader, scenes
from rtx_nerf_amd.train import RayDataset, Trainer, camera_rays, psnr


def make_demo(path, n_frames=16, res=64, grid=32):
    """Writes transforms_train.json + PNGs rendered from the analytic teacher field of tools/train_demo.py."""
    from train_demo import teacher_field
    os.makedirs(os.path.join(path, "train"), exist_ok=True)
    occ = torch.from_numpy(scenes.pack_occupancy(scenes.sphere_density(grid, 0.72)).view(np.int32).copy()).cuda()
    tr = Trainer(grid, occ, encoding="freq", n_neurons=64, n_hidden_layers=2, batch_rays=res * res, max_segments=res * res * 40,
                 density_scale=150.0)
    focal = scenes.lego_focal_length(True)
    frames = []
    for i in range(n_frames):
        pose = scenes.pose_spherical(360.0 * i / n_frames, -20.0 - 20.0 * (i % 3), origin_scale=10.0)
        o, d = camera_rays(pose, focal, res, res)
        img = tr.render_rays(o, d, radiance_fn=teacher_field).reshape(res, res, 3).cpu().numpy()
        loader.write_png(os.path.join(path, "train", f"r_{i}.png"), img)
        frames.append({"file_path": f"./train/r_{i}", "rotation": 0.0, "transform_matrix": pose.tolist()})
    with open(os.path.join(path, "transforms_train.json"), "w") as f:
        json.dump({"camera_angle_x": scenes.LEGO_CAMERA_ANGLE_X, "frames": frames}, f)
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True)
    ap.add_argument("--make-demo")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--grid", type=int, default=32)
    ap.add_argument("--encoding", default="hash")
    ap.add_argument("--out", default="train_synthetic_view.png")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    if a.make_demo:
        make_demo(a.make_demo)
    # flags=2: keep the PNGs' sRGB values (the reference's stbi_loadf applies a 2.2 gamma, Q11); corrected focal (Q1);
    # origin/10 as in the reference (Q2) so that Blender's radius-4 cameras sit just outside the unit grid
    ds = loader.load_images_json(a.data, "train", flags=2)
    if ds.images.shape[0] == 0:
        sys.exit("no frames loaded")
    n_hold = max(1, ds.images.shape[0] // 8)
    train_ds = loader.ImageDataset(ds.images[:-n_hold], ds.poses[:-n_hold], ds.focal, ds.image_width, ds.image_height, 3, ds.camera_angle_x)
    rays, focal = RayDataset.from_images(train_ds, origin_scale=0.1)
    R = a.grid
    tr = Trainer(R, None, encoding=a.encoding, n_neurons=64, n_hidden_layers=2 if a.encoding == "hash" else 4,
                 hashgrid=dict(n_levels=8, n_features=2, log2_hashmap_size=15, base_resolution=8, per_level_scale=1.5),
                 batch_rays=max(a.batch, ds.image_width * ds.image_height), max_segments=max(a.batch, ds.image_width * ds.image_height) * (3 * R),
                 lr=1e-2 if a.encoding == "hash" else 2e-3, density_scale=150.0)
    g = torch.Generator(device="cuda").manual_seed(0)
    W, H = ds.image_width, ds.image_height
    o_t, d_t = camera_rays(ds.poses[-1], focal, W, H, origin_scale=0.1)
    gt = torch.from_numpy(ds.images[-1].reshape(-1, 3)).cuda()
    print(f"{rays.n} training rays from {train_ds.images.shape[0]} frames ({W}x{H}); held-out PSNR before: {psnr(tr.render_rays(o_t, d_t), gt):.2f} dB")
    for it in range(a.steps):
        loss = tr.step(*rays.sample_batch(a.batch, g))
        if (it + 1) % 100 == 0:
            frac = tr.update_occupancy(threshold=0.01) if it + 1 >= 200 else 1.0
            print(f"step {it + 1:5d} loss {float(loss.item()):.6f} occupied {100 * frac:.1f}% held-out PSNR {psnr(tr.render_rays(o_t, d_t), gt):.2f} dB", flush=True)
    img = tr.render_rays(o_t, d_t).reshape(H, W, 3).cpu().numpy()
    loader.write_png(a.out, img)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
