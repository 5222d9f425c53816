#!/usr/bin/env python3
"""Per-kernel summary of separate rocprofv3 --pmc passes, fingerprinted by the measured kernel's ISA.
  python tools/pmc_kernels_json.py [--merge <earlier.json>] <out.json> <note> <pass.csv> [<pass.csv> ...]
(--merge: kernels that these passes did not measure are carried over from an earlier summary of the same round, but only while
the library still holds the code they were measured on -- same isa_sha16 -- and marked `carried_from`; the rest is dropped)
Every *counter_collection.csv is one pass of the SAME command with its own counter set (counters are never collected together
with trace domains, and FETCH_SIZE / WRITE_SIZE each get their own pass as MI355X_MICROARCH.md prescribes).  For each kernel
of interest (KERNELS below: name pattern -> mangled-name substrings for the ISA hash) the output holds the mean of every
counter over the kernel's launches, the launch count and duration under counters, derived HBM-side bytes
(hbm_bytes_low = FETCH + WRITE as counted; hbm_bytes_high = 2 x FETCH + WRITE: on gfx950 FETCH_SIZE tallies a wide coalesced
read at half its bytes) and `isa_sha16` = tools/kernel_isa_hash.py of the library the passes ran on.  bench.py quotes
`traffic` from this file only while the library it runs still carries the same code for that kernel."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_isa_hash import kernel_isa_sha16

# record key -> (substring of the demangled kernel name in the CSV, substrings of the mangled name for the ISA hash)
KERNELS = {
    "mlp_fwd16_128_seg_half4": ("mlp_fwd16_kernel<128, 3, 10, 2, 12, 1, 3>", ["mlp_fwd16_kernelILi128ELi3ELi10ELi2ELi12ELi1ELi3E"]),
    "mlp_fwd256x16_seg_half4": ("mlp_fwd256x16_kernel<3, 10, 2, 12, 1, 3>", ["mlp_fwd256x16_kernelILi3ELi10ELi2ELi12ELi1ELi3E"]),
    "hashmlp_fwd_2": ("hashmlp_fwd_kernel<2>", ["hashmlp_fwd_kernelILi2E"]),
    "mlp_enc_fwd16_2": ("mlp_enc_fwd16_kernel<2>", ["mlp_enc_fwd16_kernelILi2E"]),
    "hashgrid_encode_f2": ("hashgrid_encode_f2_kernel", ["hashgrid_encode_f2_kernel"]),
    # (rocprofv3 leaves these two names mangled -- _Float16 in the signature -- so the pattern is the mangled prefix)
    "hashgrid_backward_pk": ("hashgrid_backward_kernelILb1ELb0E", ["hashgrid_backward_kernelILb1ELb0E"]),
    "hashgrid_backward_f32": ("hashgrid_backward_kernelILb0ELb0E", ["hashgrid_backward_kernelILb0ELb0E"]),
    "mlp_bwd_fused64_4_3": ("mlp_bwd_fused64_kernel<4, 3>", ["mlp_bwd_fused64_kernelILi4ELi3E"]),
    "volrender_l2_fused_multi": ("volrender_l2_fused_multi_kernel<4>", ["volrender_l2_fused_multi_kernelILi4E"]),
    "volrender_fwd_pair_nerf_compact": ("volrender_fwd_pair_kernel<1, true>", ["volrender_fwd_pair_kernelILi1ELb1E"]),
    "mlp_train_fwd_128_save": ("mlp_train_fwd_kernel<128, 1, 4, 0>", ["mlp_train_fwd_kernelILi128ELi1ELi4ELi0E"]),
    "mlp_train_fwd_128_out": ("mlp_train_fwd_kernel<128, 0, 4, 0>", ["mlp_train_fwd_kernelILi128ELi0ELi4ELi0E"]),
    # the lean pair as the trainer runs it for the reference's model: sampler + encoder folded in (ENC = 1 / true)
    "mlp_train_fwd_128_masks": ("mlp_train_fwd_kernel<128, 2, 4, 1>", ["mlp_train_fwd_kernelILi128ELi2ELi4ELi1E"]),
    "wgrad_recompute_all": ("wgrad_recompute_all_kernel<7, true>", ["wgrad_recompute_all_kernelILi7ELb1E"]),
    "wgrad_recompute_0_3": ("wgrad_recompute_kernel<7, 0, 3, false, 8>", ["wgrad_recompute_kernelILi7ELi0ELi3ELb0ELi8E"]),
    "wgrad_recompute_3_6": ("wgrad_recompute_kernel<7, 3, 6, false, 8>", ["wgrad_recompute_kernelILi7ELi3ELi6ELb0ELi8E"]),
    "wgrad_recompute_6_8": ("wgrad_recompute_kernel<7, 6, 8, true, 8>", ["wgrad_recompute_kernelILi7ELi6ELi8ELb1ELi8E"]),
    "mlp_bwd_128": ("mlp_bwd_kernel<128>", ["mlp_bwd_kernelILi128E"]),
    "wgrad_lds": ("wgrad_lds_kernel", ["wgrad_lds_kernel"]),
    "adam": ("adam_kernel", ["adam_kernel"]),
}


def main():
    argv = sys.argv[1:]
    earlier = None
    if argv and argv[0] == "--merge":
        earlier = json.load(open(argv[1]))
        argv = argv[2:]
    out, note, files = argv[0], argv[1], argv[2:]
    acc = {k: defaultdict(list) for k in KERNELS}
    dur = {k: [] for k in KERNELS}
    names = {}
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                kn = row["Kernel_Name"]
                for key, (pat, _) in KERNELS.items():
                    if pat in kn:
                        names[key] = kn
                        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                        dur[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    res = {"source": note, "kernels": {}}
    for key, (_, subs) in KERNELS.items():
        if not acc[key]:
            continue
        c = {k: sum(v) / len(v) for k, v in acc[key].items()}
        d = {"kernel": names[key][:160], "isa_sha16": kernel_isa_sha16(subs), "launches_per_pass": max(len(v) for v in acc[key].values()),
             "kernel_ms_under_counters": round(sum(dur[key]) / len(dur[key]), 4), **{k + "_avg": round(v, 2) for k, v in sorted(c.items())}}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:       # KiB
            d["hbm_bytes_per_launch_low"] = int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
            d["hbm_bytes_per_launch_high"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        if "GRBM_GUI_ACTIVE" in c:
            cycles = c["GRBM_GUI_ACTIVE"] / 8          # summed over the 8 XCDs
            d["effective_clock_ghz"] = round(cycles / (d["kernel_ms_under_counters"] * 1e-3) / 1e9, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                d["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles), 4)
            if "SQ_LDS_IDX_ACTIVE" in c:
                d["lds_busy_frac"] = round(c["SQ_LDS_IDX_ACTIVE"] / (256 * cycles), 4)
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:            # vector-L1 tag lookups = 128-byte lines asked of the L1 per launch
            d["l1_line_accesses_per_launch"] = int(c["TCP_TOTAL_CACHE_ACCESSES_sum"])
        if "TCP_TCC_READ_REQ_sum" in c:
            d["l1_to_l2_read_requests_per_launch"] = int(c["TCP_TCC_READ_REQ_sum"])
        if "TCC_EA0_ATOMIC_sum" in c:                      # atomic requests the L2 sends on to the memory side
            d["l2_to_memory_atomic_requests_per_launch"] = int(c["TCC_EA0_ATOMIC_sum"])
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        res["kernels"][key] = d
    if earlier:
        merged = {}
        for key in KERNELS:                            # keep the table's order
            if key in res["kernels"]:
                merged[key] = res["kernels"][key]
            elif key in earlier["kernels"] and earlier["kernels"][key]["isa_sha16"] == kernel_isa_sha16(KERNELS[key][1]):
                merged[key] = dict(earlier["kernels"][key])
                merged[key].setdefault("carried_from", earlier["source"])
        res["kernels"] = merged
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: {x: v[x] for x in v if x in ("isa_sha16", "kernel_ms_under_counters", "hbm_bytes_per_launch_low", "l2_hit_rate", "mfma_busy_frac", "launches_per_pass")}
                      for k, v in res["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
