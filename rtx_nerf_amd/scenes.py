"""Deterministic synthetic inputs (host side, numpy): camera poses on the
NeRF-synthetic hemisphere, the procedural "Lego stand-in" occupancy and seeded
MLP weights.  No dataset ships with this image, so every bench and
test input is generated here from a seed.  Pure data generation: nothing in
this module is part of the measured or parity-checked arithmetic.
"""
import math

import numpy as np

# NeRF-synthetic Lego: camera_angle_x = 0.6911112070083618, 800x800, radius ~4.03
LEGO_CAMERA_ANGLE_X = 0.6911112070083618
LEGO_RADIUS = 4.031128874


def lego_focal_length(corrected=True, width=800):
    """params.focal_length (rtx/include/params.h:33).

    corrected: focal in units of the [-1,1] image half-width, 1/tan(fov/2).
    literal (Q1): the reference's 1/tan(0.5*focal_px) (main.cu:363-365,
    loader/data_loader.cpp:85), which treats a pixel focal as an angle."""
    if corrected:
        return float(1.0 / math.tan(0.5 * LEGO_CAMERA_ANGLE_X))
    focal_px = np.float32(0.5 * 800 / math.tan(0.5 * LEGO_CAMERA_ANGLE_X))
    return float(np.float32(1.0) / np.tan(np.float32(0.5) * focal_px))


def pose_spherical(theta_deg, phi_deg, radius=LEGO_RADIUS, origin_scale=1.0):
    """Blender-convention camera-to-world (NeRF-synthetic transform_matrix):
    camera looks down its -z at the world origin, +y up; row-major 4x4.

    origin_scale: the reference divides the translation by 10 at ray
    generation (optixPrograms.cu:76-78, quirk Q2).  Pass 10 to place the
    effective camera at `radius` (outside the [-1,1]^3 grid)."""
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    trans_t = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius * origin_scale], [0, 0, 0, 1]], np.float64)
    rot_phi = np.array([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0],
                        [0, 0, 0, 1]], np.float64)
    rot_theta = np.array([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0],
                          [0, 0, 0, 1]], np.float64)
    c2w = rot_theta @ rot_phi @ trans_t
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float64) @ c2w
    return c2w.astype(np.float32)


def pack_occupancy(dense_bool):
    """bool[R,R,R] indexed [x,y,z] -> uint32 words, bit ((x*R+y)*R+z) (make_grid order, main.cu:158-160)."""
    flat = np.ascontiguousarray(dense_bool, dtype=bool).reshape(-1)
    pad = (-flat.size) % 32
    if pad:
        flat = np.concatenate([flat, np.zeros(pad, bool)])
    bits = flat.reshape(-1, 32).astype(np.uint32)
    return (bits << np.arange(32, dtype=np.uint32)).sum(axis=1, dtype=np.uint64).astype(np.uint32)


def coarse_occupancy(dense_bool):
    R = dense_bool.shape[0]
    rc = R // 4
    return dense_bool.reshape(rc, 4, rc, 4, rc, 4).any(axis=(1, 3, 5))


def lego_standin_density(R, seed=0):
    """bool[R,R,R]: union of a base plate, a body box, a cabin, two cylinders
    (wheels/arm) and a few seeded studs, roughly filling [-0.65,0.65]^3 like
    the Lego bulldozer does its bounding box.  ~5-8 % of cells occupied."""
    rng = np.random.default_rng(seed)
    ax = (np.arange(R, dtype=np.float32) + 0.5) * (2.0 / R) - 1.0
    x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
    occ = np.zeros((R, R, R), bool)

    def box(c, h):
        return (np.abs(x - c[0]) <= h[0]) & (np.abs(y - c[1]) <= h[1]) & (np.abs(z - c[2]) <= h[2])

    occ |= box((0.0, 0.0, -0.42), (0.62, 0.40, 0.05))            # base plate
    occ |= box((-0.05, 0.0, -0.20), (0.42, 0.28, 0.18))          # body
    occ |= box((-0.22, 0.0, 0.12), (0.20, 0.22, 0.16))           # cabin
    occ |= box((0.50, 0.0, -0.22), (0.10, 0.36, 0.14))           # blade
    for cy in (-0.34, 0.34):                                     # tracks: cylinders along x
        occ |= (((y - cy) ** 2 + (z + 0.30) ** 2) <= 0.12 ** 2) & (np.abs(x) <= 0.55)
    occ |= (((x - 0.15) ** 2 + (y) ** 2) <= 0.05 ** 2) & (z >= 0.0) & (z <= 0.45)  # exhaust/arm
    for _ in range(24):                                          # studs
        c = rng.uniform([-0.4, -0.25, -0.02], [0.3, 0.25, 0.02])
        occ |= ((x - c[0]) ** 2 + (y - c[1]) ** 2 <= 0.03 ** 2) & (np.abs(z - c[2] - 0.02) <= 0.03)
    return occ


def sphere_density(R, radius=0.5):
    """config 1's sphere-occupancy variant: cells whose centre is inside the sphere."""
    ax = (np.arange(R, dtype=np.float32) + 0.5) * (2.0 / R) - 1.0
    x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
    return (x * x + y * y + z * z) <= radius * radius


def xavier_params_fp16(n_neurons, n_hidden_layers, enc_padded, seed=1337):
    """Seeded Xavier-uniform fp16 weights in the tcnn layout (numpy generator;
    rtxn_mlp_initialize_params is the PCG32 product path)."""
    rng = np.random.default_rng(seed)
    parts = []

    def fill(rows, cols):
        s = math.sqrt(6.0 / (rows + cols))
        parts.append(rng.uniform(-s, s, size=rows * cols).astype(np.float32))

    fill(n_neurons, enc_padded)
    for _ in range(n_hidden_layers - 1):
        fill(n_neurons, n_neurons)
    fill(16, n_neurons)
    return np.concatenate(parts).astype(np.float16)


def llff_standin_density(R, seed=0, n_blobs=40):
    """bool[R,R,R] for the forward-facing config (BASELINE configs[4]): a sparse cloud of small blobs and thin
    fronds inside a slab |z| < 0.6, ~2 % of cells occupied ("fern-like": irregular, mostly empty)."""
    rng = np.random.default_rng(seed)
    occ = np.zeros((R, R, R), bool)
    ax = (np.arange(R, dtype=np.float32) + 0.5) * (2.0 / R) - 1.0
    for _ in range(n_blobs):
        c = rng.uniform([-0.8, -0.8, -0.55], [0.8, 0.8, 0.55])
        rad = rng.uniform(0.03, 0.10)
        lo = np.clip(((c - rad + 1) / 2 * R).astype(int), 0, R - 1)
        hi = np.clip(((c + rad + 1) / 2 * R).astype(int) + 1, 1, R)
        x, y, z = np.meshgrid(ax[lo[0]:hi[0]], ax[lo[1]:hi[1]], ax[lo[2]:hi[2]], indexing="ij")
        occ[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] |= ((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) <= rad * rad
    for _ in range(n_blobs):                                  # fronds: thin slanted sticks
        p0 = rng.uniform([-0.8, -0.8, -0.5], [0.8, 0.8, 0.5])
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        for t in np.linspace(0, 0.35, 4 * R // 8):
            q = ((p0 + t * d + 1) / 2 * R).astype(int)
            if np.all(q >= 1) and np.all(q < R - 1):
                occ[q[0] - 1:q[0] + 1, q[1] - 1:q[1] + 1, q[2] - 1:q[2] + 1] = True
    return occ


def pose_forward_facing(dx=0.0, dy=0.0, dist=2.6, origin_scale=10.0):
    """LLFF-style forward-facing camera: at (dx, dy, dist) looking down -z, +y up (row-major 4x4);
    translation pre-multiplied by origin_scale to undo the reference's origin/10 (optixPrograms.cu:76-78)."""
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 3] = np.array([dx, dy, dist], np.float32) * origin_scale
    return c2w


def teacher_field(samples):
    """Analytic radiance for training benches/tests (torch, on the samples' device): a soft sphere with
    position-dependent colour; sigma in (0,1) like the student's sigmoid output.  samples: float[S,5] -> float[S,4]."""
    import torch
    x = samples[:, :3]
    r = x.norm(dim=1)
    sigma = torch.sigmoid(30.0 * (0.5 - r))
    rgb = 0.5 + 0.5 * torch.sin(4.0 * x + torch.tensor([0.0, 2.0, 4.0], device=x.device))
    return torch.cat([rgb, sigma[:, None]], dim=1).contiguous()
