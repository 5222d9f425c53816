#!/usr/bin/env python3
"""bench.py -- Mrays/s of the 800x800 NeRF inference render (BASELINE.json configs[1]).

One "step" = one full 800x800 frame (640,000 rays) of the hot path
  trace(count) -> scan -> trace(write) -> sampler+encode+MLP -> composite
on synthetic, HBM-resident inputs: 128^3 procedural "Lego stand-in" occupancy,
the reference's 8x128 ReLU MLP with Composite-Frequency encoding (main.cu:35-69),
seeded random fp16 weights, 32 samples per crossed occupied cell
(sampler/sampler.h:4), poses on the NeRF-synthetic hemisphere.

Multi-GPU (--gpus N, launched by torch.distributed.run, one rank per GPU): the
frame's image rows are dealt round-robin to the ranks (ray sharding, no
data-path collective), each rank renders its rows, and the rendered rows are
gathered on rank 0 with one RCCL gather per frame (0.96 MB per rank at N=8).
Total work per step is fixed, so this is strong scaling.  Rank 0 checks the gathered frame bit for
bit against its own single-GPU render after the timed loop ("gather_check").

Frames are software-pipelined across three HIP streams (render.py, render_async): the next frame's traversal and
the previous frame's compositing run underneath this frame's MLP kernel; every step still enqueues one whole frame
of every stage (--serial times the one-stream form).

At N = 1 the same JSON line also carries two more records measured in the same process after the headline loop
(BASELINE.json configs[2] and configs[4] at full size; --no-extras skips them):
  "train_config3": one optimisation step of 4096 rays, hash grid (L=16, F=2, T=2^19) + 4x64 MLP, 128^3 grid: timed as one
                   hipGraph per step (Trainer.capture_step: device-side segment count, traversal one batch ahead) with the
                   host-segment-count form beside it (ms_per_step_host_count), the per-stage HIP-event split and the hash
                   gather / scatter kernels against the HBM roofline;
  "config5":       the 1008x756 forward-facing frame, 256^3 sparse grid, 8x256 MLP, with mlp_fwd256x16_kernel against the
                   MFMA roofline.
The headline (metric/value/roofline/cpu_baseline) stays configs[1].

Prints ONE JSON line on rank 0; see DESIGN.md for the roofline arithmetic.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E ~8 TB/s
def pmc_kernel(key):
    """The newest committed rocprofv3 PMC summary of kernel `key` (profiles/rNN/pmc_kernels.json, tools/pmc_kernels_json.py:
    separate FETCH_SIZE / WRITE_SIZE / SQ / TCC passes over the same workload this bench runs) -- or None if there is none, or
    if it was measured on different machine code than the library being run carries for that kernel (isa_sha16: sha-256 of the
    kernel's ISA, tools/kernel_isa_hash.py; a stale figure is not reported)."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from kernel_isa_hash import kernel_isa_sha16
        from pmc_kernels_json import KERNELS
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_kernels.json")))
        if not files or key not in KERNELS:
            return None
        d = json.load(open(files[-1]))["kernels"].get(key)
        if not d or d.get("isa_sha16") is None or d["isa_sha16"] != kernel_isa_sha16(KERNELS[key][1]):
            return None
        return d
    except Exception:
        return None


def pmc_traffic(key):
    """HBM-side bytes per launch of kernel `key` from its PMC summary (reads at the doubled gfx950 figure, MI355X_MICROARCH.md), or None."""
    d = pmc_kernel(key)
    return d.get("hbm_bytes_per_launch_high") if d else None


def time_shade(pipe, poses_d, ray_begin, n_local, steps):
    """The fused sampler+encode+MLP launch exactly as a frame makes it (RenderPipeline.shade_again: the same C entry point over
    the segments a frame just left in slot 0), bracketed by HIP events on the launch stream; returns (mean kernel ms, mean
    samples per launch)."""
    from rtx_nerf_amd import api
    ms, smp = [], []
    for i in range(steps):
        pipe.look_at.copy_(poses_d[i % len(poses_d)], non_blocking=True)
        pipe.render(ray_begin=ray_begin, ray_count=n_local)          # leaves this pose's packed segments in slot 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pipe.shade_again(0)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
        smp.append(min(int(pipe.total.item()), pipe.max_segments) * api.NUM_SAMPLES_PER_SEGMENT)
    return (float(np.mean(ms)), float(np.mean(smp))) if ms else (None, 0.0)


def extra_train_config3(steps, warmup, kernel_steps=5, train_to=1500, frames=12):
    """BASELINE.json configs[2] at full size: 4096 rays/batch, hash grid L=16 F=2 T=2^19 (base 16, scale 1.5) +
    Frequency(4) directions + 4x64 MLP, 128^3 Lego stand-in grid, K = 32, corrected ("nerf") compositor, L2 + Adam;
    targets rendered from an analytic teacher field.  One step = one full optimisation step (main.cu:619-805)."""
    from rtx_nerf_amd import scenes
    from rtx_nerf_amd.train import Trainer, camera_rays
    R, B = 128, 4096
    dense = scenes.lego_standin_density(R, seed=0)
    occ = torch.from_numpy(scenes.pack_occupancy(dense).view(np.int32).copy()).cuda()
    hgd = dict(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.5)
    tr = Trainer(R, occ, encoding="hash", n_neurons=64, n_hidden_layers=4, hashgrid=hgd, n_dir_freqs=4,
                 batch_rays=128 * 128, max_segments=128 * 128 * 32, lr=1e-2, loss_scale=128.0, density_scale=300.0, mode="nerf")
    focal = scenes.lego_focal_length(True)
    ro, rd, tg = [], [], []
    for i in range(8):
        o, d = camera_rays(scenes.pose_spherical(45.0 * i + 15.0, -30.0, origin_scale=10.0), focal, 128, 128)
        ro.append(o); rd.append(d); tg.append(tr.render_rays(o, d, radiance_fn=scenes.teacher_field).clone())
    ro, rd, tg = torch.cat(ro), torch.cat(rd), torch.cat(tg)
    g = torch.Generator(device="cuda").manual_seed(42)

    def batch():
        idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
        return ro[idx].contiguous(), rd[idx].contiguous(), tg[idx].contiguous()

    losses, samples = [], 0
    for _ in range(warmup):
        losses.append(tr.step(*batch()))
    torch.cuda.synchronize()
    first = float(losses[0].item()) if losses else None
    # (a) the step with the segment count read on the host between traversal and the rest (what the reference does, main.cu:632)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(*batch())
        samples += int(tr.total.item()) * 32     # the step has synchronised on this count already
    torch.cuda.synchronize()
    dt_host = time.perf_counter() - t0
    # (b) the same step as ONE hipGraph (Trainer.capture_step: device-side segment count, rtxn_train_gradients), sized for 1.5x
    # the largest batch seen so far; the batch gather writes straight into the graph's input buffers.  This is the headline.
    cap = int(1.5 * samples / steps / 32) + 1024
    tr.capture_step(B, launch_segments=cap, prefetch=True)   # traversal of batch i+1 beside the gradient kernels of batch i

    def batch_into_graph():
        idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
        torch.index_select(ro, 0, idx, out=tr.graph_rays_o)
        torch.index_select(rd, 0, idx, out=tr.graph_rays_d)
        torch.index_select(tg, 0, idx, out=tr.graph_targets)

    for _ in range(max(2, warmup) + 1):          # the first call only traverses its batch
        batch_into_graph()
        tr.step_captured()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):                       # each call: one batch gathered + traversed, one batch trained = one full step
        batch_into_graph()
        loss = tr.step_captured()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tr.flush_captured()
    S = samples / steps
    stages = tr.time_stages(*batch(), steps=5)
    # The hash-grid kernels against what bounds them.
    #  * encode (hashgrid_encode_f2_kernel): 16 levels x 8 corners x 4 B = 512 B and 128 gathers per sample out of a 25-MB table
    #    that lives in L2 / Infinity Cache -- a gather-rate figure, priced against the guide's Infinity-Cache gather rate, not HBM;
    #  * scatter (hashgrid_backward_kernel): only the LIVE samples (segments with a non-zero loss gradient, rtxn_live_segments)
    #    add anything: per live sample 8 corners x (one 4-B packed-fp16 atomic per hashed level, two 4-B fp32 atomics per dense
    #    level), before the wave-level run aggregation removes duplicates.  Atomics execute at the memory side at ~1.3 TB/s of
    #    added bytes chip-wide (MI355X_MICROARCH.md, Global float atomics) -- that, not the 8 TB/s stream rate, is the ceiling.
    L, F = hgd["n_levels"], hgd["n_features"]
    n_hashed = sum(1 for l in range(L) if tr.hg.level_offset(l) >= tr.hashed_lo) if tr.hash_fp16 else 0
    live = 32 * int(tr.live_ws[0].item()) if tr.live_ws is not None else int(S)     # of the last time_stages batch
    enc_b = L * 8 * F * 2
    bwd_b = n_hashed * 8 * 4 + (L - n_hashed) * 8 * F * 4
    ATOMIC_PEAK_GBS = 1300.0
    kern = {}
    if stages.get("encode"):
        # the gather's ceiling is the vector L1's line rate (one 128-B line lookup per clock and CU), as for hashmlp_fwd_kernel: see extra_render_hash
        ms = stages["encode"]
        gbs = enc_b * S / (ms * 1e-3) / 1e9
        pm = pmc_kernel("hashgrid_encode_f2")
        clock_ghz = (pm or {}).get("effective_clock_ghz") or 2.1
        n_cu = torch.cuda.get_device_properties(0).multi_processor_count
        lps = pm["l1_line_accesses_per_launch"] / max(S, 1.0) if pm and pm.get("l1_line_accesses_per_launch") else 64.0
        ach = lps * S / (ms * 1e-3) / 1e9
        kern["encode"] = {"kernel": "hashgrid_encode_f2_kernel", "ms": round(ms, 4), "bound": "vector-L1 line rate (one 128-B line lookup per clock and CU)", "bytes_per_sample": enc_b,
                          "gathers_per_s_T": round(L * 8 * S / (ms * 1e-3) / 1e12, 3), "useful_gather_gbs": round(gbs, 1),
                          "lines_per_sample": round(lps, 2), "lines_per_sample_source": "TCP_TOTAL_CACHE_ACCESSES_sum (PMC)" if pm and pm.get("l1_line_accesses_per_launch") else "assumed",
                          "achieved": round(ach, 1), "peak": round(n_cu * clock_ghz, 1), "unit": "G lines/s", "frac": round(ach / (n_cu * clock_ghz), 4),
                          "traffic": pmc_traffic("hashgrid_encode_f2")}
    if stages.get("hash_bwd"):
        ms = stages["hash_bwd"]
        gbs = bwd_b * live / (ms * 1e-3) / 1e9
        pk, f32 = pmc_kernel("hashgrid_backward_pk"), pmc_kernel("hashgrid_backward_f32")
        kern["hash_bwd"] = {"kernel": "hashgrid_backward_kernel<pk_f16> + <f32>, live segments, run-aggregated", "ms": round(ms, 4), "bound": "memory-side atomics",
                            "live_samples": live, "live_fraction": round(live / max(S, 1), 4), "atomic_bytes_per_live_sample": bwd_b,
                            "achieved": round(gbs, 1), "peak": ATOMIC_PEAK_GBS, "unit": "GB/s of added bytes before run aggregation", "frac": round(gbs / ATOMIC_PEAK_GBS, 4),
                            "traffic": (pk["hbm_bytes_per_launch_low"] + f32["hbm_bytes_per_launch_low"]) if (pk and f32 and "hbm_bytes_per_launch_low" in pk and "hbm_bytes_per_launch_low" in f32) else None,
                            "atomic_requests_leaving_l2": (pk["l2_to_memory_atomic_requests_per_launch"] + f32["l2_to_memory_atomic_requests_per_launch"])
                            if (pk and f32 and "l2_to_memory_atomic_requests_per_launch" in pk and "l2_to_memory_atomic_requests_per_launch" in f32) else None}
    if "hash_bwd" in kern and kern["hash_bwd"].get("atomic_requests_leaving_l2"):
        # What the kernel is really paced by: every atomic is executed at the memory side (PMC: TCC_EA0_ATOMIC == TCC_ATOMIC), and a wave
        # instruction whose 64 lanes hit 64 different lines leaves the L2 as 64 requests; the guide measured that shape at 0.08 TB/s of
        # added bytes = 20 G requests/s chip-wide (MI355X_MICROARCH.md, Global float atomics, row 'access shape').  Requests counted
        # AFTER the run aggregation, from the committed counter pass; the kernel time of this run.
        hb = kern["hash_bwd"]
        req_s = hb["atomic_requests_leaving_l2"] / (hb["ms"] * 1e-3) / 1e9
        hb["scattered"] = {"achieved": round(req_s, 2), "peak": 20.0, "unit": "G atomic requests/s leaving the L2 (one lane, one line)", "frac": round(req_s / 20.0, 4),
                           "requests_per_live_sample": round(hb["atomic_requests_leaving_l2"] / max(live, 1), 1)}
    dom = "hash_bwd" if "hash_bwd" in kern else "encode"
    rec = {
        "workload": "4096 rays/batch, hash grid L=16 F=2 T=2^19 base 16 x1.5 + Frequency(4) dirs + 4x64 ReLU MLP, 128^3 grid "
                    f"({100.0 * dense.mean():.1f}% cells), K=32, NeRF compositor, L2 + Adam; analytic teacher targets",
        "ms_per_step": round(1e3 * dt / steps, 4), "mrays_s": round(B * steps / dt / 1e6, 4), "steps": steps, "warmup": warmup,
        "step_form": "one hipGraph per step (device-side segment count; traversal one batch ahead as a parallel branch)", "ms_per_step_host_count": round(1e3 * dt_host / steps, 4),
        "launch_segments": cap, "truncated_steps": tr.truncated_steps, "samples_per_step": int(S), "loss_first": first, "loss_last": float(loss.item()), "dtype": "f16 MFMA / f32 accumulate",
        "stage_ms": {k: round(v, 4) for k, v in stages.items()},
        "roofline": dict(kern[dom], samples_per_launch=int(S)),
        "kernels": kern,
    }
    mlp_flop = 2 * (tr.E * 64 + 3 * 64 * 64 + 16 * 64)
    for name, mult in (("mlp_fwd", 1.0), ("mlp_bwd+wgrad", 2.0)):
        ms = stages.get(name)
        if ms:
            tf = mult * mlp_flop * S / (ms * 1e-3) / 1e12
            rec["kernels"][name] = {"ms": round(ms, 4), "flop_per_sample": int(mult * mlp_flop), "achieved": round(tf, 2),
                                    "unit": "TFLOP/s", "frac": round(tf / MFMA_F16_DENSE_PEAK_TFLOPS, 4)}
    rec_hash = extra_render_hash(tr, step_captured=lambda: (batch_into_graph(), tr.step_captured()), trained_steps=warmup + 2 * steps + max(2, warmup) + 6,
                                 kernel_steps=kernel_steps, train_to=train_to, frames=frames)
    del tr
    torch.cuda.empty_cache()
    return rec, rec_hash


def extra_render_hash(tr, step_captured, trained_steps, kernel_steps, train_to=1500, frames=12):
    """train -> render -> PSNR on the fast path (north_star: "the fully-fused MLP + hash-grid encoding"): the configs[2] model is
    trained on to `train_to` optimisation steps (captured step, wall-clocked), then drawn at 800x800 / 128^3 through
    rtxn_render_frame with the fused hash-encode + 4x64 kernel (hashmlp_fwd_kernel) on the trainer's live tensors: PSNR of two
    held-out views against the analytic teacher, pipelined Mrays/s, and the kernel against what bounds it -- the gather rate of
    the cache hierarchy (16 levels x 8 corners x 4 B = 512 B and 128 gathers per sample from a 25-MB table that lives in L2 /
    Infinity Cache), with the MFMA fraction beside it."""
    from rtx_nerf_amd import api, scenes
    from rtx_nerf_amd.train import camera_rays
    W = H = 800
    focal = scenes.lego_focal_length(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    more = max(0, train_to - trained_steps)
    for _ in range(more):
        step_captured()
    tr.flush_captured()
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    held = [scenes.pose_spherical(100.0, -35.0, origin_scale=10.0), scenes.pose_spherical(250.0, -20.0, origin_scale=10.0)]
    poses = held + [scenes.pose_spherical(360.0 * i / 4 + 15.0, -30.0, origin_scale=10.0) for i in range(2)]
    pipe = tr.render_pipeline(W, H, focal, max_segments=1024, stable_inputs=True)    # poses_d: uploaded once, never rewritten
    worst = pipe.calibrate(poses)
    # teacher frames of the held-out views: the trainer's staged path with the analytic field in place of the network, in chunks
    psnrs = []
    B = tr.B // 2                      # 8192 rays per chunk: an 800x800 view's chunks stay inside the trainer's segment capacity
    for la in held:
        o, d = camera_rays(la, focal, W, H)
        gt = torch.cat([tr.render_rays(o[i:i + B].contiguous(), d[i:i + B].contiguous(), radiance_fn=scenes.teacher_field).clone()
                        for i in range(0, W * H, B)])
        pipe.set_pose(la)
        pred = pipe.render().clone()
        torch.cuda.synchronize()
        mse = float(((pred - gt) ** 2).mean())
        psnrs.append(10.0 * np.log10(1.0 / max(mse, 1e-12)))
    poses_d = [torch.from_numpy(p.reshape(16)).cuda() for p in poses]
    outs = [torch.empty((W * H, 3), device="cuda") for _ in range(2)]
    for i in range(3):
        pipe.render_async(poses_d[i % len(poses_d)], out=outs[i & 1])
    pipe.drain_async()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        pipe.render_async(poses_d[i % len(poses_d)], out=outs[i & 1])
    pipe.drain_async()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert not pipe.overflowed(), "render_hash4x64: segment capacity overflow"
    ms, smp = time_shade(pipe, poses_d, 0, W * H, kernel_steps)
    L = tr.hg.cfg.n_levels
    gather_b = L * 8 * 4                                   # fp16 pairs: 4 B per corner
    flop = 2 * (tr.E * 64 + (tr.net.cfg.n_hidden_layers - 1) * 64 * 64 + 16 * 64)
    gbs = gather_b * smp / (ms * 1e-3) / 1e9
    tf = flop * smp / (ms * 1e-3) / 1e12
    # What bounds the kernel: the vector L1 looks up ONE 128-byte line per clock and CU, and an 8-byte gather occupies it for a
    # whole line.  peak = 256 CUs x clock lines/s; achieved = the lines the kernel actually asked its L1 for
    # (TCP_TOTAL_CACHE_ACCESSES_sum, rocprofv3 --pmc, profiles/rNN/pmc_kernels.json -- used only while the running library still
    # carries the measured code) over the kernel time measured here.  Without a matching PMC entry: the 64 lines per sample of
    # DESIGN 5.2's count (12 hashed levels x 4 (y, z) rows + the dense levels), marked as assumed.
    pm = pmc_kernel("hashmlp_fwd_2")
    clock_ghz = (pm or {}).get("effective_clock_ghz") or 2.1
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    peak_lines = n_cu * clock_ghz                          # G lines/s
    if pm and pm.get("l1_line_accesses_per_launch"):
        lines_per_sample, src = pm["l1_line_accesses_per_launch"] / max(smp, 1.0), "TCP_TOTAL_CACHE_ACCESSES_sum per launch / samples per launch"
    else:
        lines_per_sample, src = 64.0, "assumed (DESIGN 5.2): no PMC entry for the running kernel"
    ach_lines = lines_per_sample * smp / (ms * 1e-3) / 1e9
    rec = {
        "workload": f"{W}x{H} inference render of the TRAINED configs[2] model (hash grid L={L} F=2 T=2^{tr.hg.cfg.log2_hashmap_size} + Frequency(4) dirs + "
                    f"4x64 MLP, {tr.step_count} optimisation steps on the analytic teacher), 128^3 Lego stand-in grid, 32 midpoint samples/segment, "
                    "NeRF compositor; rtxn_render_frame_async on the trainer's live tensors",
        "ms_per_step": round(1e3 * dt / frames, 4), "mrays_s": round(W * H * frames / dt / 1e6, 4), "steps": frames, "rays_per_step": W * H,
        "segments_per_frame_max": worst, "mean_samples_per_ray": round(smp / (W * H), 2), "dtype": "f16 table + MFMA / f32 accumulate",
        "psnr_vs_teacher_db": [round(p, 2) for p in psnrs], "held_out_views": 2, "train_steps": tr.step_count,
        "train_seconds_after_bench": round(train_s, 3), "train_steps_in_those_seconds": more,
        "roofline": {"kernel": "hashmlp_fwd_kernel<2>", "bound": "vector-L1 line rate (one 128-B line lookup per clock and CU; the 25-MB table is served by L2 / Infinity Cache, not HBM)",
                     "achieved": round(ach_lines, 1), "peak": round(peak_lines, 1), "unit": "G lines/s", "frac": round(ach_lines / peak_lines, 4),
                     "lines_per_sample": round(lines_per_sample, 2), "lines_per_sample_source": src, "lines_per_sample_assumed_r03": 64,
                     "clock_ghz": clock_ghz, "texture_path_busy_frac": (round(pm["TA_TA_BUSY_sum_avg"] / (n_cu * pm["kernel_ms_under_counters"] * 1e-3 * clock_ghz * 1e9), 4)
                                                                          if pm and pm.get("TA_TA_BUSY_sum_avg") else None),
                     "traffic": pmc_traffic("hashmlp_fwd_2"), "kernel_ms": round(ms, 4), "bytes_per_sample": gather_b, "gathers_per_sample": L * 8,
                     "gathers_per_s": round(L * 8 * smp / (ms * 1e-3) / 1e12, 3), "gathers_unit": "T/s", "useful_gather_gbs": round(gbs, 1), "samples_per_launch": smp,
                     "mfma": {"flop_per_sample": flop, "achieved": round(tf, 1), "unit": "TFLOP/s", "frac": round(tf / MFMA_F16_DENSE_PEAK_TFLOPS, 4)}},
    }
    del pipe
    return rec


def train_ref_record(batch_rays, grid_res, steps, warmup, dense_grid=True, mode="compat", captured=True):
    """One optimisation step of the reference's OWN training iteration (main.cu:612-805): the 8x128 ReLU model with
    Composite-Frequency(10, 12) encoding (main.cu:35-69), REGULAR sampler, the reference's compositor forward and backward
    (RTXN_VR_COMPAT, vol_render.cu:19-143), L2, Adam 1e-3, loss scale 1 (main.cu:36-46); dense_grid: the reference's 8^3 dense
    grid (main.cu:394) with 32 samples per crossed cell, batch_rays = BATCH_SIZE_GRANULARITY x 176 (main.cu:185-186: 22,528 or
    45,056 depending on the tiny-cuda-nn version); otherwise the 128^3 Lego stand-in occupancy (continuity with round 2's
    4096-ray figure).  MFMA fraction = 3 x 262,144 FLOP x samples / step time / 2.5 PFLOP/s (forward + dgrad + wgrad)."""
    from rtx_nerf_amd import scenes
    from rtx_nerf_amd.train import Trainer, camera_rays
    R, B = grid_res, batch_rays
    if dense_grid:
        occ, occ_frac = None, 1.0
    else:
        dense = scenes.lego_standin_density(R, seed=0)
        occ = torch.from_numpy(scenes.pack_occupancy(dense).view(np.int32).copy()).cuda()
        occ_frac = float(dense.mean())
    focal = scenes.lego_focal_length(True)
    side = 256
    ro, rd = [], []
    for i in range(8):
        o, d = camera_rays(scenes.pose_spherical(45.0 * i + 15.0, -30.0, origin_scale=10.0), focal, side, side)
        ro.append(o); rd.append(d)
    ro, rd = torch.cat(ro), torch.cat(rd)
    g = torch.Generator(device="cuda").manual_seed(42)
    tg = torch.rand((ro.shape[0], 3), device="cuda", generator=g)
    # capacity from a counting pass over a few batches (the reference sizes 3R slots per ray, main.cu:486)
    probe = Trainer(R, occ, encoding="freq", n_neurons=64, n_hidden_layers=1, n_dir_freqs=12, batch_rays=B, max_segments=1024, mode=mode)
    need = 0
    for _ in range(4):
        idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
        o, d = ro[idx].contiguous(), rd[idx].contiguous()
        kw = dict(grid_res=R, rays_o=o, rays_d=d, width=B, height=1, ray_begin=0, ray_count=B, occupancy=probe.occ, occupancy_coarse=probe.coarse,
                  occupancy_bricks=probe.bricks, occupancy_super=probe.super_mip, mode=1, viewing_direction=probe.view_dirs, num_hits=probe.num_hits,
                  sub_rays=0)
        from rtx_nerf_amd import api
        api.trace_grid(None, **kw)
        need = max(need, int(probe.num_hits[:B].sum().item()))
    del probe
    torch.cuda.empty_cache()
    cap = int(need * 1.15) + 1024
    tr = Trainer(R, occ, encoding="freq", n_neurons=128, n_hidden_layers=8, n_dir_freqs=12, batch_rays=B, max_segments=cap, lr=1e-3,
                 loss_scale=1.0 if mode == "compat" else 128.0, density_scale=1.0 if mode == "compat" else 300.0, mode=mode)

    # (batches are gathered into buffers allocated once: a fresh tensor per step went through torch's caching allocator, and after
    # an earlier record's torch.cuda.empty_cache() that made the EAGER loop below 2-4x slower on the host while every kernel and
    # the captured step took what they always take -- tools/probe/order_probe.py, profiles/r04/order_probe.txt)
    bo, bd, bt = (torch.empty((B, 3), device="cuda") for _ in range(3))

    def batch():
        idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
        torch.index_select(ro, 0, idx, out=bo)
        torch.index_select(rd, 0, idx, out=bd)
        torch.index_select(tg, 0, idx, out=bt)
        return bo, bd, bt

    for _ in range(warmup):
        loss = tr.step(*batch())
    torch.cuda.synchronize()
    samples = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(*batch())
        samples += min(int(tr.total.item()), tr.max_segments) * 32
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    S = samples / steps
    stages = tr.time_stages(*batch(), steps=3)
    live_frac = (float(tr.live_ws[0].item()) * 32 / max(float(tr.total.item()) * 32, 1.0)) if tr.live_segments else 1.0   # of the last batch
    ms = 1e3 * dt / steps
    # Matrix work and bytes of the MLP kernels AS EXECUTED (per sample of the batch; lf = fraction of segments with a loss gradient):
    #  lean path (8 x 128: Trainer.lean): forward (all samples) 1 x 262,144 FLOP, 224 B of encoding in, 32 B out, 128 B of sign masks;
    #    dgrad chain (live) 1 x, 168 B in (masks, dout, outputs), 2,080 B of dZ out; weight gradient (live) 1 x + 15 recomputed
    #    layer-forwards of 8 (three passes: 2 + 5 + 8) = 2.875 x, 2,080 B of dZ + 3 x 224 B of encoding in;
    #  saved-activation path: forward 224 + 2,048 + 128 + 32 B (all samples, or outputs-only for all + saving pass for the live
    #    ones when two_pass), dgrad 168 + 2,080 B, weight gradient 4,096 + 224 + 32 B (live).
    nominal = 3 * 262144
    if getattr(tr, "lean", False):
        path = "lean: sign masks only, weight gradient recomputes the activations (three passes)"
        flop_exec = 262144 * (1.0 + live_frac * (1.0 + 1.0 + 15.0 / 8.0))
        bytes_exec = 384 + live_frac * (2248 + 2080 + 3 * 224)
        if getattr(tr, "lean_fused", False):
            # sampler + encoder folded into the forward and into the weight gradient: no encT (forward: 1 B of segment constants in, 32 B out,
            # 128 B of masks, 4 B of t_vals); the last hidden layer's dZ is formed in the weight-gradient kernel, so the chain writes and the
            # weight gradient reads 7 x 256 + 32 B of dZ
            path = "lean, encoder folded in: sign masks only, weight gradient recomputes encoding, activations and the last layer's dZ (three passes)"
            bytes_exec = 165 + live_frac * (168 + 1824 + 1824)
    else:
        path = "saved activations" + (" (outputs-only forward + saving pass over the live segments)" if tr.two_pass else "")
        flop_exec = 262144 * (1.0 + 2.0 * live_frac + (live_frac if tr.two_pass else 0.0))
        bytes_exec = ((256 + live_frac * 2400) if tr.two_pass else 2432) + live_frac * (2248 + 4352)
    tf = flop_exec * S / (ms * 1e-3) / 1e12
    # HBM bytes of the three lean kernels from the committed counter passes (tools/pmc_lean_bytes.py: FETCH_SIZE doubled + WRITE_SIZE per
    # sample of the 22,528-ray batch, 88 % of its segments live), quoted only for the dense batches they were measured on and only while
    # all three kernels still carry the measured machine code
    traffic = None
    if getattr(tr, "lean", False) and dense_grid and all(pmc_kernel(k) for k in ("mlp_train_fwd_128_masks", "mlp_bwd_128", "wgrad_recompute_all")):
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r04", "lean_pmc_bytes_per_sample.json")) as f:
                traffic = int(json.load(f)["lean path, three kernels"]["bytes_per_sample_fetch_doubled"] * S)
        except (OSError, KeyError, ValueError):
            traffic = None
    rec = {
        "workload": f"{B} rays/batch, 8x128 ReLU MLP + Composite-Frequency(10, 12), REGULAR sampler, "
                    f"{'the reference compositor fwd/bwd (RTXN_VR_COMPAT)' if mode == 'compat' else 'NeRF compositor'}, L2, Adam 1e-3; "
                    f"{R}^3 grid ({'dense, as main.cu:394' if dense_grid else f'Lego stand-in, {100 * occ_frac:.1f}% cells'}), K=32; random targets",
        "ms_per_step": round(ms, 4), "mrays_s": round(B * steps / dt / 1e6, 4), "steps": steps, "warmup": warmup, "samples_per_step": int(S),
        "segment_capacity": cap, "truncated_steps": tr.truncated_steps, "loss_last": float(loss.item()), "step_form": "eager (segment count on the host, as main.cu:632)",
        "mlp_path": path, "mlp_workspace_gib": round(tr.ws.numel() * 2 / 2 ** 30, 3) if tr.ws is not None else 0.0,
        "stage_ms": {k: round(v, 4) for k, v in stages.items()},
        "roofline": {"kernels": ("mlp_train_fwd_kernel<128, masks> + mlp_bwd_kernel<128> + wgrad_recompute_all_kernel" if getattr(tr, "lean", False)
                                 else "mlp_train_fwd_kernel<128> + mlp_bwd_kernel<128> + wgrad_lds_kernel"),
                     "bound": "mfma", "achieved": round(tf, 1),
                     "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_F16_DENSE_PEAK_TFLOPS, 4),
                     "flop_executed_per_sample": int(flop_exec), "flop_nominal_per_sample": nominal, "samples_per_launch": int(S), "traffic": traffic,
                     "live_fraction": round(live_frac, 4),
                     "note": "whole step; achieved / frac count the matrix work EXECUTED (forward for every sample; dgrad, weight gradient and "
                             "its recomputed forward layers for the segments that carry a loss gradient); frac_nominal = the asked "
                             "3 x 262,144 FLOP for every sample whether visited or not"},
    }
    if captured:
        # the same step as ONE hipGraph with the traversal of the next batch beside the gradient kernels (Trainer.capture_step,
        # as train_config3's headline): no host read of the segment count (main.cu:632 synchronises for it), no launch gaps
        tr.capture_step(B, launch_segments=min(cap, int(1.3 * S / 32) + 1024), prefetch=True)

        def into_graph():
            idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
            torch.index_select(ro, 0, idx, out=tr.graph_rays_o)
            torch.index_select(rd, 0, idx, out=tr.graph_rays_d)
            torch.index_select(tg, 0, idx, out=tr.graph_targets)

        for _ in range(max(2, warmup) + 1):
            into_graph()
            tr.step_captured()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            into_graph()
            loss_c = tr.step_captured()
        torch.cuda.synchronize()
        dt_c = time.perf_counter() - t0
        tr.flush_captured()
        ms_c = 1e3 * dt_c / steps
        rec["ms_per_step_host_count"] = rec["ms_per_step"]
        rec["ms_per_step"], rec["mrays_s"] = round(ms_c, 4), round(B * steps / dt_c / 1e6, 4)
        rec["step_form"] = "one hipGraph per step, traversal of the next batch beside the gradient kernels (segment count on the device)"
        rec["truncated_steps"] = tr.truncated_steps
        rec["loss_last_captured"] = float(loss_c.item())
        tf = flop_exec * S / (ms_c * 1e-3) / 1e12
        rec["roofline"]["achieved"], rec["roofline"]["frac"] = round(tf, 1), round(tf / MFMA_F16_DENSE_PEAK_TFLOPS, 4)
    rec["roofline"]["frac_nominal"] = round(nominal * S / (rec["ms_per_step"] * 1e-3) / 1e12 / MFMA_F16_DENSE_PEAK_TFLOPS, 4)
    mlp_ms = sum(stages.get(k, 0.0) for k in ("mlp_fwd", "mlp_fwd_live", "mlp_bwd+wgrad"))
    if mlp_ms > 0:
        # the three MLP kernels alone (HIP events around each stage of one more batch): executed FLOP and executed bytes over their time
        rec["roofline"]["mlp_kernels_ms"] = round(mlp_ms, 4)
        rec["roofline"]["mlp_kernels_frac"] = round(flop_exec * S / (mlp_ms * 1e-3) / 1e12 / MFMA_F16_DENSE_PEAK_TFLOPS, 4)
        rec["roofline"]["mlp_kernels_bytes_per_sample"] = int(bytes_exec)
        gbs = bytes_exec * S / (mlp_ms * 1e-3) / 1e9
        rec["roofline"]["mlp_kernels_hbm_gbs"] = round(gbs, 1) if gbs <= HBM_PEAK_GBS else None      # above the peak the byte model is wrong: say nothing
        rec["roofline"]["mlp_kernels_hbm_frac"] = round(gbs / HBM_PEAK_GBS, 4) if gbs <= HBM_PEAK_GBS else None
    del tr
    torch.cuda.empty_cache()
    return rec


def extra_train_ref8x128(steps, warmup):
    # (round 3 measured the 4096-ray variant first because after the two large ones "its steps took twice as long".  Round 4 looked:
    # tools/probe/order_probe.py -- kernel times by HIP events and the captured step are the same before and after a 43-GiB
    # workspace has come and gone (ratio 0.99-1.01); what doubled was the host side of the EAGER loop after torch.cuda.empty_cache(),
    # through per-step tensor allocations that train_ref_record no longer makes.  The order is kept for continuity of the records.)
    small = train_ref_record(4096, 128, 3 * steps, warmup, dense_grid=False, mode="nerf")
    return {"b22528_dense8": train_ref_record(128 * 176, 8, steps, warmup, captured=False),
            "b45056_dense8": train_ref_record(256 * 176, 8, max(4, steps // 2), warmup, captured=False),
            "b4096_lego128_nerf": small}


def extra_config5(steps, warmup, kernel_steps):
    """BASELINE.json configs[4] at full size: 1008x756 forward-facing frame, 256^3 sparse fern-like grid, 8x256 MLP."""
    from rtx_nerf_amd import api, render, scenes
    W, H, R = 1008, 756, 256
    dense = scenes.llff_standin_density(R, seed=3)
    occ = torch.from_numpy(scenes.pack_occupancy(dense).view(np.int32).copy()).cuda()
    net = api.Network(n_neurons=256, n_hidden_layers=8)
    net.set_params(torch.from_numpy(scenes.xavier_params_fp16(256, 8, net.encoded_width(), seed=1337)).cuda())
    poses = [scenes.pose_forward_facing(0.3 * np.cos(i), 0.2 * np.sin(i)) for i in range(4)]
    pipe = render.RenderPipeline(net, R, W, H, 1.6, occupancy=occ, max_segments=1024, stable_inputs=True)
    worst = pipe.calibrate(poses)
    poses_d = [torch.from_numpy(p.reshape(16)).cuda() for p in poses]
    out = [torch.empty((W * H, 3), device="cuda") for _ in range(2)]
    for i in range(warmup):
        pipe.render_async(poses_d[i % 4], out=out[i & 1])
    pipe.drain_async()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        pipe.render_async(poses_d[(warmup + i) % 4], out=out[i & 1])
    pipe.drain_async()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert not pipe.overflowed(), "config5: segment capacity overflow"
    ms, smp = time_shade(pipe, poses_d, 0, W * H, kernel_steps)
    flops = net.flops_per_sample()
    ach = flops * smp / (ms * 1e-3) / 1e12
    # A 256-wide layer is 128 KiB of A fragments and streams through LDS once per 256-sample block tile: layer 0 4 x 16 KiB, seven
    # hidden layers 7 x 128 KiB, the output layer's two rotations 16 KiB = 976 KiB per 256 samples = 3,904 B of LDS-DMA fill per
    # sample (L2 -> LDS, no HBM).  MI355X_MICROARCH.md ('ldsdma-fill') puts the chip-wide ceiling of that path at 6.4 TB/s.
    LDSDMA_PEAK_GBS, fill_b = 6400.0, (4 * 16 + 7 * 128 + 16) * 1024 // 256
    fill = fill_b * smp / (ms * 1e-3) / 1e9
    frac_mfma, frac_fill = ach / MFMA_F16_DENSE_PEAK_TFLOPS, fill / LDSDMA_PEAK_GBS
    rec = {
        "workload": f"{W}x{H} inference render, {R}^3 grid (procedural LLFF-fern stand-in, {100.0 * dense.mean():.1f}% cells), "
                    "8x256 ReLU MLP + Composite-Frequency encoding, 32 samples/segment, 4 forward-facing poses, seeded random fp16 weights",
        "ms_per_step": round(1e3 * dt / steps, 4), "mrays_s": round(W * H * steps / dt / 1e6, 4), "steps": steps, "warmup": warmup,
        "rays_per_step": W * H, "segments_per_frame_max": worst, "mean_samples_per_ray": round(smp / (W * H), 2), "dtype": "f16",
        "roofline": {"kernel": "mlp_fwd256x16_kernel<3,10,2,12,segments,half4>",
                     "mfma": "v_mfma_f32_16x16x32_f16", "bound": "lds-dma fill (L2 -> LDS weight stream)" if frac_fill > frac_mfma else "mfma",
                     "ldsdma_fill": {"bytes_per_sample": fill_b, "achieved": round(fill, 1), "peak": LDSDMA_PEAK_GBS, "unit": "GB/s", "frac": round(frac_fill, 4)},
                     "achieved": round(ach, 2),
                     "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(frac_mfma, 4),
                     "traffic": pmc_traffic("mlp_fwd256x16_seg_half4"), "flop_per_sample": flops, "samples_per_launch": smp, "kernel_ms": round(ms, 4)},
    }
    del pipe, net
    torch.cuda.empty_cache()
    return rec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--neurons", type=int, default=128)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--poses", type=int, default=4)
    ap.add_argument("--scene", default="lego", choices=["lego", "llff"], help="lego: hemisphere poses around the Lego stand-in (configs[1]); "
                    "llff: forward-facing frustum over a sparse fern-like grid (configs[4], with --grid 256 --neurons 256)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the train_config3 / config5 records (N = 1 only)")
    ap.add_argument("--extra-steps", type=int, default=30, help="timed steps of each extra record")
    ap.add_argument("--kernel-steps", type=int, default=5, help="extra frames with HIP events around the MLP kernel")
    ap.add_argument("--no-compact", action="store_true", help="fp32 float4 radiance + t_vals between the MLP kernel and the compositor "
                    "(the reference's convertHalfToFloat layout) instead of the network's half outputs")
    ap.add_argument("--reserve-cus", type=int, default=-1, help="CUs kept free of the persistent MLP grid (default: 4 at N > 1, else 0)")
    ap.add_argument("--serial", action="store_true", help="one stream, stages of a frame strictly one after another (no frame pipelining)")
    ap.add_argument("--emulate-shard-of", type=int, default=0, metavar="N", help="diagnostic, single process: render only rank 0's "
                    "row shard of an N-rank run (no collective) to see what one rank's frame costs; the JSON line is marked "
                    "'emulated_shard_of' and its value is NOT a whole-job figure")
    return ap.parse_args()


def self_launch(n):
    """`python3 bench.py --gpus N` without a launcher around it: start the N ranks ourselves.  This process has made no HIP call
    (importing torch makes none), and it never execs: the ranks are a CHILD `python -m torch.distributed.run --nproc-per-node N
    bench.py <same arguments>` on 127.0.0.1 and a free port; rank 0's JSON line reaches our stdout through the inherited
    descriptor, and the child's exit status is ours."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher's --nproc-per-node must equal --gpus",
                  file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # RTXN_REHEARSE_ON_ONE_GPU=1: every rank uses cuda:0 and the gather goes over gloo -- only to rehearse the
    # N>1 code path on a one-GPU box; never used by the driver's real runs (one GPU per rank, RCCL).
    rehearse = os.environ.get("RTXN_REHEARSE_ON_ONE_GPU") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))

    from rtx_nerf_amd import api, render, scenes
    from rtx_nerf_amd.shard import RowShard

    W, H, R = args.width, args.height, args.grid
    dense = scenes.lego_standin_density(R, seed=0) if args.scene == "lego" else scenes.llff_standin_density(R, seed=3)
    words = scenes.pack_occupancy(dense)
    occ = torch.from_numpy(words.view(np.int32).copy()).cuda()
    net = api.Network(n_neurons=args.neurons, n_hidden_layers=args.layers)
    params = scenes.xavier_params_fp16(args.neurons, args.layers, net.encoded_width(), seed=1337)
    net.set_params(torch.from_numpy(params).cuda())
    if args.reserve_cus >= 0:
        net.set_reserved_cus(args.reserve_cus)
    elif world > 1 and not rehearse:
        net.set_reserved_cus(4)   # the per-frame RCCL gather runs beside the MLP kernel: give its kernels somewhere to land
                                  # (free: the chip is power-limited under this kernel, 4 idle CUs cost no time -- DESIGN 3.4)
    if args.scene == "lego":
        focal = scenes.lego_focal_length(True)
        poses = [scenes.pose_spherical(360.0 * i / args.poses + 15.0, -30.0, origin_scale=10.0) for i in range(args.poses)]
    else:
        focal = 1.6
        poses = [scenes.pose_forward_facing(0.3 * np.cos(i), 0.2 * np.sin(i)) for i in range(args.poses)]

    # ray shard of this rank: image rows rank, rank+world, ... (rtx_nerf_amd/shard.py)
    sh = RowShard(W, H, 0, args.emulate_shard_of) if (args.emulate_shard_of > 1 and world == 1) else RowShard(W, H, rank, world)
    n_local, ray_begin = sh.n_local, sh.ray_begin
    pipe = render.RenderPipeline(net, R, W, H, focal, occupancy=occ, max_rays=n_local, max_segments=1024,
                                 window=sh.window, compact=False if args.no_compact else None,
                                 stable_inputs=True)    # poses_d below: one device buffer per pose, uploaded once, never rewritten
    worst = pipe.calibrate(poses, ray_begin=ray_begin, ray_count=n_local)
    poses_d = [torch.from_numpy(p.reshape(16)).cuda() for p in poses]

    # gather plumbing: shards are padded to the largest one so every rank sends n_max rows
    n_max = sh.n_max
    pix_bufs = [torch.zeros((n_max, 3), device="cuda") for _ in range(2)]
    gather_bufs = [[torch.empty((n_max, 3), device="cuda") for _ in range(world)] for _ in range(2)] \
        if (world > 1 and rank == 0) else [None, None]
    pending = [None, None]

    # Frames are software-pipelined (render.py, render_async): traversal of frame i+1 and compositing of frame i-1 run on
    # side streams underneath the MLP kernel of frame i; every step still enqueues exactly one full frame of every stage.
    def step(i):
        b = i & 1
        pix, done, comp = pipe.render_async(poses_d[i % len(poses_d)], ray_begin=ray_begin, ray_count=n_local,
                                            out=pix_bufs[b][:n_local]) if not args.serial else (None, None, None)
        if args.serial:
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
            pipe.look_at.copy_(poses_d[i % len(poses_d)], non_blocking=True)
            pipe.render(ray_begin=ray_begin, ray_count=n_local, out=pix_bufs[b][:n_local])
            if world > 1:
                pending[b] = sh.gather(pix_bufs[b], gather_bufs[b], async_op=True)
        elif world > 1:
            with torch.cuda.stream(comp):          # composite(i) -> gather(i) -> composite(i+2) on the compositor stream
                w = sh.gather(pix_bufs[b], gather_bufs[b], async_op=True)
                w.wait()                           # RCCL: a stream-side wait, it orders the next writer of this buffer

    def drain():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        pipe.drain_async()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert not pipe.overflowed(), "segment capacity overflow: calibrate() margin too small"

    # ---- N > 1: the gathered frame must be the single-GPU frame, bit for bit (outside the timed region) ----
    gather_check = None
    if world > 1:
        last = args.warmup + args.steps - 1
        if rank == 0:
            full = render.RenderPipeline(net, R, W, H, focal, occupancy=occ, max_segments=1024)
            full.calibrate([poses[last % len(poses)]])
            full.set_pose(poses[last % len(poses)])
            want = full.render().reshape(H, W, 3)
            got = sh.assemble(gather_bufs[last & 1])
            torch.cuda.synchronize()
            gather_check = "bit-identical to the single-GPU frame" if torch.equal(got, want) else \
                f"MISMATCH: {int((got != want).any(dim=2).sum())} of {W * H} pixels differ"
            del full

    # ---- dominant kernel (fused sampler+encode+MLP), HIP events on the launch stream ----
    kern_ms, kern_smp = time_shade(pipe, poses_d, ray_begin, n_local, args.kernel_steps)

    rays_per_step = W * H
    value = rays_per_step * args.steps / elapsed / 1e6
    out = {
        "metric": "Mrays/sec rendered (NeRF inference render, traverse->sample->MLP->composite)",
        "value": round(value, 4),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
        "config": {
            "workload": f"{W}x{H} inference render, {R}^3 grid (procedural {'Lego' if args.scene == 'lego' else 'LLFF-fern'} stand-in occupancy, "
                        f"{100.0 * dense.mean():.1f}% cells), {args.layers}x{args.neurons} ReLU MLP + Composite-Frequency "
                        f"encoding, 32 samples/segment, {args.poses} hemisphere poses, seeded random fp16 weights",
            "rays_per_step": rays_per_step,
            "parallelism": (f"ray-shard x{world} (rows round-robin) + " + ("gloo gather (one-GPU rehearsal)" if rehearse
                            else "RCCL gather")) if world > 1 else "single GPU",
            "trace_mode": "dda+mip",
        },
    }
    if gather_check is not None:
        out["gather_check"] = gather_check
    if args.emulate_shard_of > 1 and world == 1:
        out["emulated_shard_of"] = args.emulate_shard_of
        out["value"] = None   # one rank's shard only: not a throughput of the workload
    if rank == 0:
        flops = net.flops_per_sample()
        ms, smp = kern_ms, kern_smp
        if ms:
            ach = flops * smp / (ms * 1e-3) / 1e12
            out["roofline"] = {
                "kernel": (f"mlp_fwd256x16_kernel<3,10,2,12,segments," if args.neurons == 256 else f"mlp_fwd16_kernel<{args.neurons},3,10,2,12,segments,")
                          + f"{'half4' if pipe.compact else 'radiance'}>",
                "mfma": "v_mfma_f32_16x16x32_f16",
                "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_F16_DENSE_PEAK_TFLOPS, 4),
                "traffic": None,
                "flop_per_sample": flops, "samples_per_launch": smp, "kernel_ms": round(ms, 4),
            }
            # PMC bytes were collected on the default frame (all 640,000 rays: 100.09 M samples per launch); a rank's shard launches
            # 1/N of that, and the kernel's traffic is per sample (8 B written + the segment records), so it is scaled by the samples
            full = args.width == 800 and args.height == 800 and args.grid == 128 and args.neurons == 128 and args.layers == 8 and pipe.compact
            t = pmc_traffic("mlp_fwd16_128_seg_half4") if full else None
            if t:
                out["roofline"]["traffic"] = int(t) if world == 1 else int(t * smp / 100.0878e6)
                if world > 1:
                    out["roofline"]["traffic_note"] = "PMC bytes of the full frame's launch x this rank's share of the samples"
        out["config"]["segments_per_frame_local_max"] = worst
        out["config"]["mean_samples_per_ray"] = round(smp / max(n_local, 1), 2)

    # ---- cpu_baseline + PSNR vs oracle on a bounded ray sample spread over ALL bench poses (rank 0, N=1 only) ----
    if rank == 0 and world == 1 and not args.no_cpu:
        import oracle as O   # checker / baseline only
        cfg = O.mlp_cfg(n_neurons=args.neurons, n_hidden_layers=args.layers)

        def every(nrays, phase=0):
            stride = max(1, (W * H) // nrays)
            ids = (np.arange(min(nrays, W * H), dtype=np.int64) * stride + phase % stride)
            return stride, ids[ids < W * H].astype(np.uint32)

        # the timed leg is the oracle's TILED form (orc_render_tiled: 64 samples at a time through an AVX2 + FMA micro-kernel, fp16
        # roundings by F16C, OpenMP over rays) -- first held to the scalar restatement orc_render on a probe of the same rays
        _, probe_ids = every(1024)
        exact, _ = O.render(poses[0], focal, W / H, W, H, R, words, 1, cfg, params, probe_ids[:256])
        t1 = time.perf_counter()
        tiled, _ = O.render_tiled(poses[0], focal, W / H, W, H, R, words, 1, cfg, params, probe_ids)
        probe_s = max(time.perf_counter() - t1, 1e-3)
        tiled_err = float(np.abs(tiled[:256] - exact).max())
        assert tiled_err < 1e-3, f"cpu_baseline: orc_render_tiled differs from orc_render by {tiled_err}"
        n_cpu = int(min(W * H * len(poses), max(4096, 1024 * args.cpu_seconds / probe_s)))
        per_pose = max(256, n_cpu // len(poses))
        cpu_s, cpu_rays, cpu_samples, sq_err, max_err, stride = 0.0, 0, 0, 0.0, 0.0, 1
        for pi, pose in enumerate(poses):
            pipe.look_at.copy_(poses_d[pi])
            gpu_pix = pipe.render(ray_begin=0, ray_count=n_local).cpu().numpy()
            stride, ray_ids = every(per_pose, phase=pi * 7919)
            t1 = time.perf_counter()
            cpu_pix, ns = O.render_tiled(pose, focal, W / H, W, H, R, words, 1, cfg, params, ray_ids)
            cpu_s += time.perf_counter() - t1
            cpu_rays += len(ray_ids)
            cpu_samples += int(ns)
            diff = gpu_pix[ray_ids] - cpu_pix
            sq_err += float((diff.astype(np.float64) ** 2).sum())
            max_err = max(max_err, float(np.abs(diff).max()))
        mse = sq_err / (3.0 * cpu_rays)
        out["psnr_vs_oracle_db"] = round(10.0 * np.log10(1.0 / max(mse, 1e-20)), 2)
        out["max_abs_err_vs_oracle"] = max_err
        flops = net.flops_per_sample() * cpu_samples / cpu_s
        out["cpu_baseline"] = {
            "value": round(cpu_rays / cpu_s / 1e6, 6), "unit": "Mrays/s", "cores": O.num_threads(),
            "kind": "port",
            "note": "the CPU restatement's tiled form: 64 samples per tile through an AVX2 + FMA micro-kernel (4 samples x 16 outputs), fp16 "
                    "weights widened once, fp32 accumulate, activations rounded to fp16 by F16C, single-precision sines on an exactly reduced "
                    "argument, OpenMP over rays; gcc -O3 -mavx2 -mfma -mf16c (no AVX-512 path).  Held to the scalar restatement "
                    f"(orc_render) on 256 of these rays before timing: max |difference| {tiled_err:.1e}",
            "gflops": round(flops / 1e9, 1), "gflops_per_core": round(flops / 1e9 / max(O.num_threads(), 1), 2),
            "sample": f"{cpu_rays} rays (every {stride}th ray of each of the {len(poses)} bench poses, {cpu_samples} samples), "
                      f"oracle/rtxn_oracle.c orc_render_tiled, {cpu_s:.1f} s",
        }

    # ---- configs[2] and configs[4] at full size, same process, after the headline (N = 1 only) ----
    if rank == 0 and world == 1 and not args.no_extras and not args.emulate_shard_of:
        del pipe
        torch.cuda.empty_cache()
        out["train_config3"], out["render_hash4x64"] = extra_train_config3(args.extra_steps, 5, args.kernel_steps)
        out["train_ref8x128"] = extra_train_ref8x128(max(6, args.extra_steps // 3), 3)
        out["config5"] = extra_config5(max(8, args.extra_steps // 2), 3, args.kernel_steps)
    if rank == 0:
        print(json.dumps(out), flush=True)
    bad = gather_check is not None and gather_check.startswith("MISMATCH")
    if world > 1:
        flag = torch.tensor([1 if bad else 0], dtype=torch.int32, device="cpu" if rehearse else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)     # every rank leaves with the same status
        bad = bool(flag.item())
        dist.destroy_process_group()
    if bad:
        print("bench.py: gathered frame differs from the single-GPU frame", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
