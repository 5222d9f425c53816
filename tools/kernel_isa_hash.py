#!/usr/bin/env python3
"""Fingerprint of a kernel AS BUILT: sha-256 over the machine code of the gfx950 function(s) in librtxn.so whose mangled names
contain the given substrings.  A PMC summary under profiles/ is stamped with the fingerprint of the kernel it was measured on
(tools/make_pmc_json.py) and bench.py quotes its `traffic` only while the library being run still carries the same code --
the measured ISA, not the source file, is what is compared (a comment edit does not invalidate a measurement, a compiler
flag that changes the code does).
  python tools/kernel_isa_hash.py mlp_fwd16_kernelILi128ELi3ELi10ELi2ELi12ELi1ELi3E [more substrings...]"""
import hashlib
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(lib_path):
    """every gfx950 code object bundled in the library's .hip_fatbin section (one bundle per translation unit)"""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
        blob = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, q)
            ident = blob[q + 24:q + 24 + idlen].decode()
            q += 24 + idlen
            if "gfx950" in ident and size:
                out.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def _functions(co):
    """{mangled name: machine code bytes} of one code object (ELF64 little endian, parsed with llvm-readelf)"""
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co)
        f.flush()
        secs = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-SW", f.name], capture_output=True, text=True).stdout
        syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", f.name], capture_output=True, text=True).stdout
    m = re.search(r"\]\s+\.text\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", secs)
    if not m:
        return {}
    t_addr, t_off = int(m.group(1), 16), int(m.group(2), 16)
    fns = {}
    for line in syms.splitlines():
        p = line.split()
        if len(p) >= 8 and p[3] == "FUNC" and p[6] != "UND":
            addr, size, name = int(p[1], 16), int(p[2]), p[7]
            fns[name] = co[t_off + addr - t_addr:t_off + addr - t_addr + size]
    return fns


def kernel_isa_sha16(substrings, lib_path=None):
    """sha-256 (first 16 hex digits) over the code of every function whose name contains one of `substrings`, in name order;
    None if no function matches or the tools are missing."""
    lib_path = lib_path or os.path.join(ROOT, "rtx_nerf_amd", "librtxn.so")
    try:
        found = {}
        for co in _code_objects(lib_path):
            for name, code in _functions(co).items():
                if any(s in name for s in substrings):
                    found[name] = code
    except (OSError, subprocess.CalledProcessError, struct.error):
        return None
    if not found:
        return None
    h = hashlib.sha256()
    for name in sorted(found):
        h.update(name.encode() + b"\0" + found[name])
    return h.hexdigest()[:16]


if __name__ == "__main__":
    subs = sys.argv[1:] or ["mlp_fwd16_kernelILi128ELi3ELi10ELi2ELi12ELi1ELi3E"]
    print(kernel_isa_sha16(subs))
