#!/usr/bin/env python3
"""The staged hash-grid inference path on a full 800x800 / 128^3 frame's segments: rtxn_hashgrid_encode_segments (all levels
per block, or the levels dealt to the XCDs: RTXN_HASH_ENCODE_XCD) + the 64-wide layer stack over the encoded input, timed by HIP
events, beside the fused kernel of the same frame.  encode_frame_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rtx_nerf_amd import api, render, scenes
R, W, H = 128, 800, 800
occ = torch.from_numpy(scenes.pack_occupancy(scenes.lego_standin_density(R, seed=0)).view(np.int32).copy()).cuda()
hgd = dict(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.5)
hg = api.HashGrid(n_dir_freqs=4, **hgd)
E = hg.encoded_width()
net = api.Network(n_neurons=64, n_hidden_layers=4, n_encoded_features=E)
net.set_params(torch.from_numpy(scenes.xavier_params_fp16(64, 4, E, seed=5)).cuda())
g = torch.Generator().manual_seed(1)
table = ((torch.rand(hg.n_params(), generator=g) * 2 - 1) * 1e-1).half().cuda()
focal = scenes.lego_focal_length(True)
pipe = render.RenderPipeline(net, R, W, H, focal, occupancy=occ, max_segments=1024, hashgrid=hg, table=table, sample_type=api.SAMPLING_MIDPOINT_WORLD)
la = scenes.pose_spherical(30.0, -30.0, origin_scale=10.0)
pipe.calibrate([la])
pipe.set_pose(la)
pipe.render()
torch.cuda.synchronize()
v = pipe._slot_views(0)
P = int(v.total.item())
S = P * 32
Sp = api.padded_samples(S)
encT = torch.empty((E, Sp), dtype=torch.float16, device="cuda")
out = torch.empty((S, 16), dtype=torch.float16, device="cuda")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_enc = timed(lambda: hg.encode_segments(table, v.start, v.end, v.seg_view, P, api.SAMPLING_MIDPOINT_WORLD, encT))
t_mlp = timed(lambda: net.train_forward_outputs(encT, S, out))
t_fused = timed(lambda: pipe.shade_again(0))
print(f"{P} segments = {S} samples; RTXN_HASH_ENCODE_XCD={os.environ.get('RTXN_HASH_ENCODE_XCD', '1')}: encode {t_enc:.3f} ms "
      f"({16 * 8 * S / t_enc / 1e9:.2f} T gathers/s), layer stack over encT {t_mlp:.3f} ms, sum {t_enc + t_mlp:.3f} ms; fused kernel {t_fused:.3f} ms")
