#!/usr/bin/env python3
"""Static scan of a gfx950 assembly listing (hipcc -S --cuda-device-only) for one suspected hazard pattern: a ReLU/convert
instruction (the asm units hipcc cannot see into) writing a register that an OUT-OF-PLACE v_mfma issued fewer than eight
wait states earlier still names as SrcC.  Found in passing AND in failing builds of the builtin-MFMA pipeline,
i.e. not the cause of the column-tile-0 errors; zero hits since the k-steps are single asm statements (in-place MFMAs).
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Irtx_nerf_amd/csrc -S --cuda-device-only -o /tmp/mlp.s rtx_nerf_amd/csrc/mlp.hip
  python tools/probe/scan_mfma_war.py /tmp/mlp.s"""
import re, sys
def regs(op):
    m = re.match(r"v\[(\d+):(\d+)\]", op)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", op)
    if m: return {int(m.group(1))}
    return set()
def scan(path, verbose=True):
    name, ins, out = None, [], {}
    def flush():
        if name is None: return
        hits = []
        for j, l in enumerate(ins):
            if not l.startswith("v_mfma"): continue
            ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
            if len(ops) < 4: continue
            d, c = regs(ops[0]), regs(ops[3])
            if not c or c == d: continue
            ws = 0
            for n in ins[j + 1:j + 12]:
                if n.startswith("s_nop"):
                    ws += int(n.split()[1]) + 1
                    continue
                if n.startswith(("v_cvt_pk_f16_f32", "v_pk_max_i16")):
                    w = regs(n.split(None, 1)[1].split(",")[0].strip())
                    if w & c: hits.append((l, n, ws))
                ws += 1
                if ws >= 8: break
        if hits: out[name] = hits
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            flush(); name, ins = m.group(1), []
            continue
        t = line.split(";")[0].strip()
        if t and not t.startswith(".") and not t.endswith(":") and name is not None: ins.append(t)
    flush()
    return out
if __name__ == "__main__":
    r = scan(sys.argv[1])
    for k, v in r.items():
        print(k[:70], len(v), v[0])
    print("kernels with hits:", len(r))
