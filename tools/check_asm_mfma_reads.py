#!/usr/bin/env python3
"""Static check of the gfx950 ISA for the hazard class hipcc cannot cover: a VGPR written by an MFMA that sits INSIDE an
inline-asm statement and read by a COMPILER-generated instruction (outside any asm statement) too few wait states later.
hipcc's hazard recogniser pads MFMA-result -> VALU/VMEM/LDS reads for the MFMAs it emitted itself; it does not look into asm
text, so for asm MFMAs the wait states are the source's job (rtxn::mfma_results_settle) -- and a plain asm `s_nop` statement
orders nothing held in registers, so whether the reads really stay behind it has to be read off the ISA.

    check_asm_mfma_reads.py file.s [...]        exit status 1 and one line per finding

Wait states: one per instruction, N + 1 for `s_nop N`; an intervening MFMA counts as ONE (conservative: it really holds the
pipe for its passes).  Required: 19 (16 passes + 3, the gfx950 figure for the longest MFMA used here).  The scan is linear over
each kernel's text (labels and branches are ignored: the kernels' asm pipelines are straight-line code)."""
import re
import sys

NEED = 19
RE_KERNEL = re.compile(r"^(_Z\w+):")
# VGPRs and AGPRs are tracked in separate namespaces ("v", n) / ("a", n): the weight-gradient accumulators of
# mlp_bwd_fused64_kernel and wgrad_recompute_* live in AGPRs through asm "+a" operands, and a compiler-generated
# v_accvgpr_read (or any other instruction naming an a-register) inside the wait states is the same hazard.
RE_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok):
    out = []
    for m in RE_REG.finditer(tok):
        if m.group(1) is not None:
            out.append((m.group(1), int(m.group(2))))
        else:
            out.extend((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def check(path):
    findings = []
    kernel, in_asm = None, False
    pending = {}            # (file, number) of a VGPR / AGPR -> wait states since an asm MFMA wrote it
    for n, raw in enumerate(open(path), 1):
        line = raw.strip()
        m = RE_KERNEL.match(line)
        if m:
            kernel, pending, in_asm = m.group(1), {}, False
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line[0] in ";." or line.endswith(":") or kernel is None:
            continue
        op, _, rest = line.partition(" ")
        rest = rest.split(";")[0]
        ops = [o.strip() for o in rest.split(",")]
        if op.startswith("s_nop"):
            states = int(ops[0], 0) + 1
        else:
            states = 1
        is_store = op.startswith(("global_store", "buffer_store", "ds_write", "flat_store", "global_atomic", "ds_add"))
        dst_ops = [] if (is_store or op.startswith(("s_", "v_cmp", "v_cmpx"))) else ops[:1]
        src_ops = ops if (is_store or op.startswith("v_cmp")) else ops[1:]
        if not in_asm and not op.startswith("s_"):
            for r in {r for o in src_ops for r in regs(o)}:
                if r in pending and pending[r] < NEED:
                    findings.append(f"{path}:{n}: {kernel[:70]}: `{line}` reads {r[0]}{r[1]} {pending[r]} wait states after an asm MFMA wrote it")
                    break
        for r in list(pending):
            pending[r] += states
            if pending[r] >= 64:
                del pending[r]
        written = [r for o in dst_ops for r in regs(o)]
        if in_asm and op.startswith(("v_mfma", "v_smfmac")):
            for r in written:
                pending[r] = 0
        else:
            for r in written:
                pending.pop(r, None)
    return findings


if __name__ == "__main__":
    bad = [f for p in sys.argv[1:] for f in check(p)]
    print("\n".join(bad) if bad else f"no compiler-generated read of an asm MFMA result within {NEED} wait states ({len(sys.argv) - 1} files)")
    sys.exit(1 if bad else 0)
