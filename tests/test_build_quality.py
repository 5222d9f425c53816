"""Build-time properties of the gfx950 code objects (no GPU needed: hipcc cross-compiles): no kernel may use private
(scratch) memory or spill vector registers -- a select on an array element once made hipcc index an input array through
scratch in two variants without anyone noticing -- and the fused MLP kernels must keep the two-waves-per-SIMD budget."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = sorted(glob.glob(os.path.join(ROOT, "rtx_nerf_amd", "csrc", "*.hip")))
ONE_WAVE_PER_SIMD = ("mlp_bwd_fused64_kernel",)   # __launch_bounds__(256, 1): see the kernel's header comment


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", SOURCES, ids=[os.path.basename(s) for s in SOURCES])
def test_kernels_use_no_scratch_and_do_not_spill(src, tmp_path):
    out = tmp_path / "k.s"
    extra = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if os.path.basename(src) == "train.hip" else []    # as the Makefile builds it
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", *extra, f"-I{ROOT}/include",
                           f"-I{ROOT}/rtx_nerf_amd/csrc", "-S", "--cuda-device-only", "-o", str(out), src],
                          stderr=subprocess.DEVNULL, timeout=600)
    txt = out.read_text()
    kernels = re.findall(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S)
    if "__global__" not in open(src).read():
        return   # host-only translation unit (common.hip)
    assert kernels, "no kernel metadata found"
    for name, body in kernels:
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", body).group(1))
        spills = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", body).group(1))
        vgprs = int(re.search(r"\.vgpr_count:\s+(\d+)", body).group(1))
        if any(k in name for k in ONE_WAVE_PER_SIMD):
            # designed for one wave per SIMD and the whole 512-entry register file (VGPR + AGPR): values beyond the 256
            # architectural VGPRs live in AGPRs, which hipcc reports as "spills" although nothing leaves the register file
            assert scratch == 0 and vgprs <= 512, f"{name}: {scratch} B scratch, {vgprs} registers"
            continue
        assert scratch == 0 and spills == 0, f"{name}: {scratch} B scratch, {spills} VGPR spills"
        if "mlp_fwd" in name or "mlp_train_fwd" in name or "mlp_bwd" in name:
            assert vgprs <= 256, f"{name}: {vgprs} VGPRs break the two-waves-per-SIMD budget"
