import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import oracle
from rtx_nerf_amd import api
rng = np.random.default_rng(16)
n = 3000
def _inputs(rng, n):
    return np.concatenate([rng.uniform(-1, 1, (n, 3)), rng.uniform(0, 3.1416, (n, 1)), rng.uniform(-3.1416, 3.1416, (n, 1))], axis=1).astype(np.float32)
import inspect, tests.test_gpu_train as T
x = T._inputs(rng, n)
hg = api.HashGrid(16, 2, 19, 16, 1.5, n_dir_freqs=4)
ocfg = oracle.hg_cfg(16, 2, 19, 16, 1.5)
table = rng.uniform(-1, 1, hg.n_params()).astype(np.float16)
encT = hg.encode(torch.from_numpy(table).cuda(), torch.from_numpy(x).cuda()).cpu().numpy()
want = oracle.encode_hg(ocfg, 4, table, x)
got = encT[:32, :n].T
bad = np.argwhere(got != want[:, :32])
print("mismatches", bad.tolist())
for s, j in bad:
    l = j // 2
    scale = np.float32(16 * 1.5 ** l - 1)
    p = np.float32(np.float32(x[s, :3] * np.float32(0.5) + np.float32(0.5)) * np.float32(np.exp2(np.float32(l) * np.log2(np.float32(1.5))) * 16 - 1) + np.float32(0.5))
    print("sample", s, "feat", j, "level", l, "got", got[s, j], "want", want[s, j], "x", x[s, :3], "p", p, "cell", np.floor(p), "fr", p - np.floor(p))

# emulate sample 2215, level 2 (dense index) in float32 with two association orders of the weight product
f32 = np.float32
def emu(s, l, order):
    res = int(np.ceil(16 * 1.5 ** l - 1)) + 1
    off = 0
    for k in range(l):
        r = int(np.ceil(16 * 1.5 ** k - 1)) + 1
        off += min((r ** 3 + 7) // 8 * 8, 2 ** 19)
    size = min((res ** 3 + 7) // 8 * 8, 2 ** 19)
    sc = f32(np.exp2(f32(l) * np.log2(f32(1.5))) * f32(16) - f32(1))
    x01 = (x[s, :3] * f32(0.5) + f32(0.5)).astype(f32)   # fma in the kernel; products by 0.5 are exact
    p = np.array([np.float32(np.float64(x01[a]) * np.float64(sc) + 0.5) for a in range(3)], f32)   # fmaf: one rounding
    g = np.floor(p).astype(np.int64); fr = (p - np.floor(p)).astype(f32)
    acc = [f32(0), f32(0)]
    for c in range(8):
        hi = [(c >> a) & 1 for a in range(3)]
        wa = [fr[a] if hi[a] else f32(1) - fr[a] for a in range(3)]
        w = f32(f32(wa[0] * wa[1]) * wa[2]) if order == 0 else f32(wa[0] * f32(wa[1] * wa[2]))
        px, py, pz = g[0] + hi[0], g[1] + hi[1], g[2] + hi[2]
        if res ** 3 <= size:
            idx = (px + py * res + pz * res * res) % size
        else:
            idx = ((px * 1) ^ ((py * 2654435761) & 0xffffffff) ^ ((pz * 805459861) & 0xffffffff)) % size
        for f in range(2):
            v = np.float64(f32(table[(off + idx) * 2 + f]))
            acc[f] = f32(np.float64(w) * v + np.float64(acc[f]))     # fmaf
    return [np.float16(a) for a in acc]
for s, l in ((2215, 2), (2702, 15)):
    print(s, l, "order (x y) z:", emu(s, l, 0), " order x (y z):", emu(s, l, 1), " oracle", want[s, 2 * l: 2 * l + 2], "gpu", got[s, 2 * l: 2 * l + 2])
