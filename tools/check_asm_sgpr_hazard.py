#!/usr/bin/env python3
"""Static check of the device listings for one hazard hipcc pads for its own instructions but not for inline asm: a vector-memory
instruction may read an SGPR that a VALU instruction (v_readlane / v_readfirstlane: a restored spill) wrote only 5 wait states
later.  An asm `global_load... s[a:b]` placed right behind the restore reads the OLD pair -- a wild address (round 4: the first
register-staged weight loads of the lean weight-gradient kernel faulted exactly so).  The asm statement carries its own `s_nop 4`.
    check_asm_sgpr_hazard.py build/*.s        exit status 1 and one line per finding"""
import re
import sys

NEED = 5
RE_VALU_SGPR = re.compile(r"^(v_readlane_b32|v_readfirstlane_b32)\s+s(\d+)\b")
RE_VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)")
RE_S = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def check(path):
    findings, kernel, in_asm, recent = [], None, False, {}   # recent: sgpr -> wait states since a VALU wrote it
    for n, raw in enumerate(open(path), 1):
        ls = raw.strip()
        m = re.match(r"^(_Z\w+):", ls)
        if m:
            kernel, recent = m.group(1), {}
            continue
        if ls.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if ls.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if kernel is None or not ls or ls[0] in ";." or ls.endswith(":"):
            continue
        if in_asm and RE_VMEM.match(ls):
            for mm in RE_S.finditer(ls):
                rs = [int(mm.group(1))] if mm.group(1) else range(int(mm.group(2)), int(mm.group(3)) + 1)
                hit = [r for r in rs if r in recent]
                if hit:
                    findings.append(f"{path}:{n}: {kernel[:70]}: asm `{ls[:60]}` reads s{hit[0]} {recent[hit[0]]} wait states after a VALU wrote it")
                    break
        step = 1
        m = re.match(r"^s_nop\s+(\d+)", ls)
        if m:
            step = int(m.group(1)) + 1
        recent = {r: w + step for r, w in recent.items() if w + step < NEED}
        m = RE_VALU_SGPR.match(ls)
        if m:
            recent[int(m.group(2))] = 0
    return findings


if __name__ == "__main__":
    bad = []
    for p in sys.argv[1:]:
        bad += check(p)
    print("\n".join(bad) if bad else f"no asm vector-memory instruction reads an SGPR within {NEED} wait states of a VALU write ({len(sys.argv) - 1} files)")
    sys.exit(1 if bad else 0)
