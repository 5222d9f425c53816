#!/usr/bin/env python3
"""Generates tests/golden/loader_stb.npz: PNG files (inputs) and what the REFERENCE's own decoder returns for them.

The reference decodes its training images with stbi_loadf(path, &w, &h, &n, 3) (loader/data_loader.cpp:63) from its
vendored loader/stb_image.h.  That header compiles in the build container as it lies (oracle/Makefile `ref` ->
oracle/_ref/stb_loadf, a 30-line main of ours around the untouched header; no stand-in headers), so this is the one
row of the hot-path table whose oracle is pinned by the reference itself rather than by a restatement.

The PNG set is synthesised here by a small encoder of our own so that every decode path stbi_loadf can take is
covered: colour types 0/2/3/4/6, bit depths 1/2/4/8/16, Adam7 interlacing, all five scanline filters, palette with and
without tRNS, tRNS colour keys on gray/RGB, several IDAT chunks, 1x1 and odd sizes.  Only the fixture (PNG bytes +
expected floats) travels to the GPU box; the reference never does.

A second fixture, loader_stb_fuzz.npz (fuzz_fixture below), holds random files instead of designed ones.

Run from the repo root in the build container:   make -C oracle ref && python tests/golden/make_loader_golden.py
"""
import os
import struct
import subprocess
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "stb_loadf")


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _pack_rows(samples, depth):
    """samples: uint16 [h, w*channels] -> list of bytes per scanline at `depth` bits per sample."""
    rows = []
    for r in samples:
        if depth == 8:
            rows.append(bytes(r.astype(np.uint8)))
        elif depth == 16:
            rows.append(r.astype(">u2").tobytes())
        else:
            bits = "".join(format(int(v), f"0{depth}b") for v in r)
            bits += "0" * ((-len(bits)) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


def _filter_rows(rows, bpp, filters):
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        ft = filters[y % len(filters)]
        if len(prev) != len(row):
            prev = bytes(len(row))
        enc = bytearray(len(row))
        for x, v in enumerate(row):
            a = row[x - bpp] if x >= bpp else 0
            b = prev[x]
            c = prev[x - bpp] if x >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[ft]
            enc[x] = (v - pred) & 0xFF
        out.append(ft)
        out += enc
        prev = row
    return bytes(out)


ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def encode_png(img, ctype, depth, interlace=False, filters=(0,), palette=None, trns=None, idat_split=0):
    """img: uint16 [h, w, channels] of raw sample values (palette indices for ctype 3)."""
    h, w, ch = img.shape
    assert ch == {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, ch * depth // 8)
    raw = b""
    if not interlace:
        raw = _filter_rows(_pack_rows(img.reshape(h, w * ch), depth), bpp, filters)
    else:
        for x0, y0, dx, dy in ADAM7:
            sub = img[y0::dy, x0::dx]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            raw += _filter_rows(_pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, filters)
    comp = zlib.compress(raw, 6)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += _chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).reshape(-1)))
    if trns is not None:
        out += _chunk(b"tRNS", bytes(trns))
    if idat_split and len(comp) > idat_split:
        for i in range(0, len(comp), idat_split):
            out += _chunk(b"IDAT", comp[i:i + idat_split])
    else:
        out += _chunk(b"IDAT", comp)
    return out + _chunk(b"IEND", b"")


def cases():
    rng = np.random.default_rng(20261004)
    out = {}

    def rnd(h, w, ch, depth):
        return rng.integers(0, 1 << depth, (h, w, ch), dtype=np.uint16)

    def grad(h, w, ch, depth):    # smooth content: every filter type produces non-trivial residues
        y, x = np.mgrid[0:h, 0:w]
        base = (x * 7 + y * 13)[:, :, None] + np.arange(ch)[None, None, :] * 29
        return (base * ((1 << depth) - 1) // max(1, base.max())).astype(np.uint16)

    allf = (0, 1, 2, 3, 4)
    out["rgba8_nerf_like"] = encode_png(rnd(9, 13, 4, 8), 6, 8, filters=allf)           # the NeRF-synthetic format
    out["rgba8_grad_paeth"] = encode_png(grad(12, 10, 4, 8), 6, 8, filters=(4, 3, 1, 2))
    out["rgb8"] = encode_png(rnd(7, 5, 3, 8), 2, 8, filters=allf)
    out["gray8"] = encode_png(rnd(6, 11, 1, 8), 0, 8, filters=allf)
    out["graya8"] = encode_png(rnd(5, 7, 2, 8), 4, 8, filters=allf)
    out["rgb8_1x1"] = encode_png(rnd(1, 1, 3, 8), 2, 8)
    out["rgba8_multi_idat"] = encode_png(rnd(16, 16, 4, 8), 6, 8, filters=allf, idat_split=97)
    for d in (1, 2, 4):
        out[f"gray{d}"] = encode_png(rnd(7, 13, 1, d), 0, d, filters=(0, 1, 2))
    pal = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    out["pal8"] = encode_png(rnd(8, 9, 1, 8), 3, 8, palette=pal, filters=allf)
    out["pal8_trns"] = encode_png(rnd(8, 9, 1, 8), 3, 8, palette=pal, trns=rng.integers(0, 256, 200, dtype=np.uint8), filters=allf)
    out["pal4_trns"] = encode_png(rnd(5, 11, 1, 4), 3, 4, palette=pal[:16], trns=rng.integers(0, 256, 7, dtype=np.uint8))
    out["pal2"] = encode_png(rnd(6, 9, 1, 2), 3, 2, palette=pal[:4])
    out["pal1"] = encode_png(rnd(3, 17, 1, 1), 3, 1, palette=pal[:2])
    out["rgb16"] = encode_png(rnd(6, 5, 3, 16), 2, 16, filters=allf)
    out["rgba16"] = encode_png(rnd(5, 6, 4, 16), 6, 16, filters=allf)
    out["gray16"] = encode_png(rnd(4, 9, 1, 16), 0, 16, filters=allf)
    out["graya16"] = encode_png(rnd(4, 5, 2, 16), 4, 16, filters=allf)
    out["rgb8_colorkey"] = encode_png(rnd(4, 4, 3, 2).astype(np.uint16) * 85, 2, 8, trns=struct.pack(">HHH", 85, 170, 0))
    out["gray8_colorkey"] = encode_png(rnd(5, 5, 1, 2).astype(np.uint16) * 85, 0, 8, trns=struct.pack(">H", 170))
    out["gray4_colorkey"] = encode_png(rnd(5, 6, 1, 4), 0, 4, trns=struct.pack(">H", 3))
    out["rgb16_colorkey"] = encode_png(rnd(3, 4, 3, 1).astype(np.uint16) * 0xFFFF, 2, 16, trns=struct.pack(">HHH", 0xFFFF, 0, 0xFFFF))
    out["rgba8_adam7"] = encode_png(rnd(11, 13, 4, 8), 6, 8, interlace=True, filters=allf)
    out["rgba8_adam7_small"] = encode_png(rnd(3, 2, 4, 8), 6, 8, interlace=True)
    out["rgb16_adam7"] = encode_png(rnd(9, 10, 3, 16), 2, 16, interlace=True, filters=allf)
    out["gray4_adam7"] = encode_png(rnd(9, 11, 1, 4), 0, 4, interlace=True, filters=(0, 2))
    out["gray1_adam7"] = encode_png(rnd(10, 21, 1, 1), 0, 1, interlace=True)
    out["pal8_trns_adam7"] = encode_png(rnd(8, 8, 1, 8), 3, 8, interlace=True, palette=pal, trns=rng.integers(0, 256, 256, dtype=np.uint8), filters=allf)
    out["pal2_adam7"] = encode_png(rnd(7, 9, 1, 2), 3, 2, interlace=True, palette=pal[:4])
    # things stbi_loadf refuses: both sides must fail
    good = out["rgb8"]
    out["bad_signature"] = b"\x89PNX" + good[4:]
    out["bad_truncated"] = good[: len(good) // 2]
    out["bad_depth3"] = good[:24] + b"\x03" + good[25:]      # IHDR bit depth 3 (CRC not checked by stb)
    return out


def main():
    if not os.path.exists(REF_BIN):
        sys.exit(f"{REF_BIN} missing: run `make -C oracle ref` in the build container (needs /root/reference)")
    cs = cases()
    names = sorted(cs)
    arrays = {"names": np.array(names)}
    with tempfile.TemporaryDirectory() as tmp:
        paths = []
        for n in names:
            p = os.path.join(tmp, n + ".png")
            open(p, "wb").write(cs[n])
            paths.append(p)
        outbin = os.path.join(tmp, "out.bin")
        subprocess.run([REF_BIN, outbin] + paths, check=True, stderr=subprocess.DEVNULL)
        blob = open(outbin, "rb").read()
    pos = 0
    for n in names:
        ok, w, h, ch = struct.unpack_from("<4i", blob, pos)
        pos += 16
        arrays["png_" + n] = np.frombuffer(cs[n], np.uint8)
        arrays["hdr_" + n] = np.array([ok, w, h, ch], np.int32)
        if ok:
            cnt = w * h * 3
            arrays["out_" + n] = np.frombuffer(blob, np.float32, cnt, pos).reshape(h, w, 3).copy()
            pos += 4 * cnt
        print(f"{n:22s} ok={ok} {w}x{h} channels_in_file={ch}")
    assert pos == len(blob)
    np.savez_compressed(os.path.join(HERE, "loader_stb.npz"), **arrays)
    print("wrote tests/golden/loader_stb.npz:", os.path.getsize(os.path.join(HERE, "loader_stb.npz")), "bytes")


def _palette_overrun(png):
    """True if a palette image may index beyond its PLTE (the reference then reads an uninitialised stack array: its
    output for such a file is not a fact about the format, and the fixture leaves the file out)."""
    o, ctype, depth, pal = 8, None, 8, 0
    while o + 8 <= len(png):
        n, t = struct.unpack(">I4s", png[o:o + 8])
        if t == b"IHDR" and ctype is None and n >= 13:
            depth, ctype = png[o + 16], png[o + 17]
        if t == b"PLTE":
            pal = n // 3
        o += 12 + n
    return ctype == 3 and pal < (1 << min(depth, 8))


def fuzz_fixture(count=320, seed=11):
    """tests/golden/loader_stb_fuzz.npz: `count` structurally valid PNGs with RANDOM content from tools/san/loader_fuzz.cpp
    (--emit: every colour type / bit depth / interlace mode, legal and illegal combinations, random filter bytes, PLTE and
    tRNS of random length, scanline data of the right length or a little off, one file in ten damaged afterwards) and what
    the reference's stbi_loadf returned for each: refused, or the floats.  The same generator, 30,000 files over four seeds
    (8,000 of them all damaged), agreed with our decoder on every file before this fixture was cut."""
    gen = os.path.join(tempfile.gettempdir(), "rtxn_loader_fuzz_emit")
    subprocess.run(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "rtx_nerf_amd", "csrc"), os.path.join(ROOT, "rtx_nerf_amd", "csrc", "loader.cpp"),
                    os.path.join(ROOT, "tools", "san", "loader_fuzz.cpp"), "-lz", "-o", gen], check=True)
    arrays, names, skipped = {}, [], 0
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([gen, "--emit", tmp, str(count), str(seed)], check=True)
        files = sorted(f for f in os.listdir(tmp) if f.endswith(".png"))
        outbin = os.path.join(tmp, "out.bin")
        subprocess.run([REF_BIN, outbin] + [os.path.join(tmp, f) for f in files], check=True, stderr=subprocess.DEVNULL)
        blob = open(outbin, "rb").read()
        pos = 0
        for f in files:
            png = open(os.path.join(tmp, f), "rb").read()
            ok, w, h, ch = struct.unpack_from("<4i", blob, pos)
            pos += 16
            img = None
            if ok:
                img = np.frombuffer(blob, np.float32, w * h * 3, pos).reshape(h, w, 3).copy()
                pos += 4 * w * h * 3
            if ok and _palette_overrun(png):
                skipped += 1
                continue
            n = f[:-4]
            names.append(n)
            arrays["png_" + n] = np.frombuffer(png, np.uint8)
            arrays["hdr_" + n] = np.array([ok, w, h, ch], np.int32)
            if ok:
                arrays["out_" + n] = img
        assert pos == len(blob)
    arrays["names"] = np.array(names)
    path = os.path.join(HERE, "loader_stb_fuzz.npz")
    np.savez_compressed(path, **arrays)
    n_ok = sum(int(arrays["hdr_" + n][0]) for n in names)
    print(f"wrote tests/golden/loader_stb_fuzz.npz: {len(names)} files ({n_ok} decoded, {len(names) - n_ok} refused by the reference; "
          f"{skipped} palette files with out-of-range indices left out), {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
    fuzz_fixture()
