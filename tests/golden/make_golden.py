#!/usr/bin/env python3
"""Generates the fixtures in this directory.

The reference (owensgroup/rtx_nerf) holds NO golden vectors, tests or fixtures and cannot be built or run here
(DESIGN.md section 0), so nothing below comes from the reference itself:

  kat_closed_form.npz   inputs + expected outputs derived INDEPENDENTLY of the oracle, in numpy float64 from the
                        formulas the reference source spells out (file:line in tests/test_golden.py).  Both the
                        oracle and the HIP kernels are checked against these.
  oracle_snapshot.npz   outputs of oracle/rtxn_oracle.c on one small seeded scene.  A REGRESSION pin only: it
                        guards the oracle (the checker) against accidental change; it is not evidence of parity
                        with the reference.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def closed_form():
    rng = np.random.default_rng(20261003)
    out = {}
    # --- sampler REGULAR (sampler.cu:52-66): sample = t*dir + origin, t = i/32; t_vals = (i+1)/32
    sp = rng.uniform(-1, 1, (5, 3)).astype(np.float32)
    ep = rng.uniform(-1, 1, (5, 3)).astype(np.float32)
    nh = np.array([2, 0, 3], np.int32)
    vd = rng.uniform(-3, 3, (3, 2)).astype(np.float32)
    i = np.arange(32, dtype=np.float64) / 32
    pos = sp[:, None, :].astype(np.float64) + i[None, :, None] * (ep.astype(np.float64) - sp.astype(np.float64))[:, None, :]
    seg_ray = np.repeat(np.arange(3), nh)
    samples = np.concatenate([pos, np.broadcast_to(vd[seg_ray][:, None, :].astype(np.float64), (5, 32, 2))], axis=2)
    out.update(smp_start=sp, smp_end=ep, smp_num_hits=nh, smp_view=vd, smp_samples=samples.reshape(-1, 5),
               smp_t_vals=np.tile((np.arange(32) + 1) / 32.0, 5))
    # --- minstd_rand (thrust default engine, sampler.cu:117): first 64 draws and the 10000th
    x, seq = 1, []
    for _ in range(10000):
        x = (x * 48271) % 2147483647
        seq.append(x)
    out.update(minstd_first64=np.array(seq[:64], np.uint32), minstd_10000th=np.array([seq[-1]], np.uint32))
    # --- volume render forward COMPAT (vol_render.cu:19-73) on a ragged CSR, float64
    nh = np.array([3, 0, 1, 5], np.int32)
    idx = np.concatenate([[0], np.cumsum(nh)[:-1]]).astype(np.int32)
    P = int(nh.sum())
    rad = rng.uniform(0, 1, (P * 32, 4)).astype(np.float32)
    t = rng.uniform(0, 1, P * 32).astype(np.float32)
    pix = np.zeros((4, 3))
    for r in range(4):
        T, tp = 0.0, 0.0
        for s in range(idx[r] * 32, (idx[r] + nh[r]) * 32):
            d = abs(float(t[s]) - tp)                       # :56, t_prev not reset per segment
            tp = float(t[s])
            T += d * float(rad[s, 3])                       # :60, inclusive
            w = np.exp(-T) * (1 - np.exp(-d * float(rad[s, 3])))   # :61-63
            pix[r] += w * rad[s, :3].astype(np.float64)
    out.update(vr_radiance=rad, vr_t=t, vr_num_hits=nh, vr_indices=idx, vr_pixels=pix)
    # --- volume render backward COMPAT (vol_render.cu:75-143), float64 from fp16 loss gradients
    g = rng.standard_normal((4, 3)).astype(np.float16)
    grads = np.zeros((P * 32, 4))
    for r in range(4):
        tp = 0.0
        for s in range(idx[r] * 32, (idx[r] + nh[r]) * 32):
            d = abs(float(t[s]) - tp)
            tp = float(t[s])
            sig = float(rad[s, 3])
            tr = d * sig                                    # :118 assigned, not accumulated
            grads[s, :3] = g[r].astype(np.float64) * tr * (1 - np.exp(-d * sig))              # :133-135
            grads[s, 3] = (g[r].astype(np.float64) * tr * rad[s, :3] * d * np.exp(-sig * d)).sum()   # :127-129
    out.update(vr_loss_grads=g, vr_grads=grads)
    # --- frequency encoding (tcnn Frequency: feature j of F freqs -> dim j/2F, f (j/2)%F, sin/cos), float64
    x = rng.uniform(-1, 1, (6, 5)).astype(np.float32)
    x[:, 3:] *= 3.0
    enc = []
    for row in x:
        e = []
        for dim, F in [(0, 10), (1, 10), (2, 10), (3, 12), (4, 12)]:
            for f in range(F):
                a = np.pi * float(row[dim]) * 2.0 ** f
                e += [np.sin(a), np.cos(a)]
        enc.append(e + [1.0] * 4)
    out.update(enc_in=x, enc_out=np.array(enc))
    # --- traversal: axis-aligned and diagonal rays through R=8 (main.cu:154-174 cells of width 0.25)
    ro = np.array([[-2.0, 0.1, 0.1], [0.3, -3.0, -0.6], [-2.0, -2.0, -2.0]], np.float32)
    rd = np.array([[1, 0, 0], [0, 1, 0], [1, 1, 1] / np.sqrt(3)], np.float32)
    edges = -1.0 + np.arange(9) * 0.25
    out.update(tr_rays_o=ro, tr_rays_d=rd, tr_num_hits=np.array([8, 8, 8], np.int32), tr_edges=edges)
    # --- L2 + Adam (tcnn; main.cu:36-46)
    pred, tgt = rng.uniform(0, 1, 30).astype(np.float32), rng.uniform(0, 1, 30).astype(np.float32)
    d = pred.astype(np.float64) - tgt
    out.update(l2_pred=pred, l2_target=tgt, l2_values=d * d / 30, l2_grads=64.0 * 2 * d / 30)
    np.savez_compressed(os.path.join(HERE, "kat_closed_form.npz"), **out)


def north_star_closed_form():
    """The north_star model's arithmetic that the reference's config does not use, restated in numpy float64 from the published
    algorithms, independently of oracle/rtxn_oracle.c: tiny-cuda-nn's multiresolution hash encoding (grid.h: level scale
    base * s^l - 1, resolution ceil(scale) + 1, pos = x * scale + 0.5, dense index x + y r + z r^2 while r^3 fits the level's
    parameter count, else (x * 1) ^ (y * 2654435761) ^ (z * 805459861) in 32-bit arithmetic, both modulo the level's parameter
    count; trilinear weights), and the
    emission-absorption quadrature of the corrected compositor, w_i = exp(-sum_{j<i} sigma_j d_j) (1 - exp(-sigma_i d_i)), with
    its gradient taken numerically (central differences in float64) -- not from the closed form the kernels implement."""
    rng = np.random.default_rng(20261004)
    out = {}
    L, F, T, base, s = 6, 2, 10, 4, 1.5
    pts = rng.uniform(-1, 1, (40, 3)).astype(np.float32)
    pts[0] = (-1.0, -1.0, -1.0)
    pts[1] = (1.0, 1.0, 1.0)
    pts[2] = (0.0, 0.25, -0.5)
    levels, off = [], 0
    for l in range(L):
        scale = np.float32(np.exp2(np.float32(l) * np.log2(np.float32(s))) * np.float32(base) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        dense = (res ** 3 + 7) // 8 * 8
        size = min(dense, 1 << T)
        levels.append((float(scale), res, size, off))
        off += size
    n_params = off * F
    table = (rng.standard_normal(n_params) * 0.5).astype(np.float16)
    tab = table.astype(np.float64).reshape(-1, F)
    enc = np.zeros((pts.shape[0], L * F))
    for i, p in enumerate(pts):
        x01 = p.astype(np.float64) * 0.5 + 0.5
        for l, (scale, res, size, o) in enumerate(levels):
            pos = x01 * scale + 0.5
            g = np.floor(pos).astype(np.int64)
            fr = pos - g
            for c in range(8):
                w, q = 1.0, []
                for a in range(3):
                    hi = (c >> a) & 1
                    w *= fr[a] if hi else 1.0 - fr[a]
                    q.append(int(g[a]) + hi)
                if res ** 3 <= size:
                    idx = q[0] + q[1] * res + q[2] * res * res       # the +1 corner of a boundary cell runs past the level ...
                else:
                    idx = ((q[0] * 1) & 0xffffffff) ^ ((q[1] * 2654435761) & 0xffffffff) ^ ((q[2] * 805459861) & 0xffffffff)
                idx %= size                                          # ... grid_index() ends in `index % hashmap_size` either way
                enc[i, l * F:(l + 1) * F] += w * tab[o + idx]
    out.update(hg_cfg=np.array([L, F, T, base], np.int64), hg_scale=np.array([s], np.float64), hg_points=pts, hg_table=table, hg_enc=enc,
               hg_level_sizes=np.array([lv[2] for lv in levels], np.int64))
    # --- corrected compositor on a ragged CSR, K = 32 samples per segment
    K = 32
    nh = np.array([2, 0, 1, 3], np.int32)
    idx = np.concatenate([[0], np.cumsum(nh)[:-1]]).astype(np.int32)
    P = int(nh.sum())
    rad = rng.uniform(0, 1, (P * K, 4)).astype(np.float32)
    rad[:, 3] *= 3.0
    step = rng.uniform(0.01, 0.08, P * K).astype(np.float32)

    def render(r64):
        pix = np.zeros((nh.size, 3))
        for r in range(nh.size):
            Tacc = 0.0
            for q in range(idx[r] * K, (idx[r] + nh[r]) * K):
                x = float(step[q]) * r64[q, 3]
                pix[r] += np.exp(-Tacc) * (1.0 - np.exp(-x)) * r64[q, :3]
                Tacc += x
        return pix

    r64 = rad.astype(np.float64)
    pix = render(r64)
    g = rng.standard_normal((nh.size, 3)).astype(np.float16)
    grads = np.zeros((P * K, 4))
    eps = 1e-6
    for q in range(P * K):
        for ch in range(4):
            a, b = r64.copy(), r64.copy()
            a[q, ch] += eps
            b[q, ch] -= eps
            grads[q, ch] = ((render(a) - render(b)) * g.astype(np.float64)).sum() / (2 * eps)
    out.update(nerf_radiance=rad, nerf_step=step, nerf_num_hits=nh, nerf_indices=idx, nerf_pixels=pix, nerf_loss_grads=g, nerf_grads=grads)
    np.savez_compressed(os.path.join(HERE, "kat_north_star.npz"), **out)


def oracle_snapshot():
    import oracle as O
    from rtx_nerf_amd import scenes
    R, W, H = 16, 16, 12
    cfg = O.mlp_cfg(n_neurons=64, n_hidden_layers=2)
    params = scenes.xavier_params_fp16(64, 2, O.mlp_enc_padded(cfg), seed=99)
    occ = scenes.pack_occupancy(scenes.sphere_density(R, 0.7))
    la = scenes.pose_spherical(25.0, -35.0, origin_scale=10.0)
    f = scenes.lego_focal_length(True)
    out = {}
    for mode in (0, 1):
        pk = O.trace_packed(look_at=la, focal=f, aspect=W / H, W=W, H=H, R=R, occ=occ, mode=mode)
        pix, ns = O.render(la, f, W / H, W, H, R, occ, mode, cfg, params, np.arange(W * H))
        out.update({f"m{mode}_num_hits": pk["num_hits"], f"m{mode}_start": pk["start"], f"m{mode}_end": pk["end"],
                    f"m{mode}_pixels": pix, f"m{mode}_samples": np.array([ns])})
    out.update(params=params, occ=occ, look_at=la, focal=np.array([f], np.float64), dims=np.array([R, W, H]))
    np.savez_compressed(os.path.join(HERE, "oracle_snapshot.npz"), **out)


if __name__ == "__main__":
    closed_form()
    north_star_closed_form()
    oracle_snapshot()
    print("wrote", sorted(p for p in os.listdir(HERE) if p.endswith(".npz")))
