"""The data-parallel exchange of the hashed levels' gradient (rtx_nerf_amd/dp.py) on CPU: 2 and 3 gloo ranks drive
Half2GradExchange with torch stand-ins for the three device primitives (the product path runs librtxn's rtxn_half2_* kernels;
tests/test_gpu_train.py checks those against these same semantics on the GPU).  Every rank must end with the same bits, equal
to the lists added in rank order / the dense sum; levels are sent in the smaller form; a rank with an all-zero gradient and a
round in which nobody has anything both work."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rtx_nerf_amd.dp import Half2GradExchange

BLOCK, NB = 256, 5


class CpuOps:
    """what rtxn_half2_count_nonzero / _pack_nonzero / _add_pairs do, in torch on the CPU"""

    @staticmethod
    def _bits(values):
        return values.view(torch.int32)

    @staticmethod
    def workspace(values, block):
        return torch.zeros((values.numel() // 2 + block - 1) // block, dtype=torch.int32)

    @staticmethod
    def count(values, block, ws):
        nz = (CpuOps._bits(values) & 0x7fff7fff) != 0
        for b in range(ws.numel()):
            ws[b] = int(nz[b * block:(b + 1) * block].sum())

    @staticmethod
    def pack(values, block, ws, mask, pairs, count):
        bits = CpuOps._bits(values)
        sel = torch.tensor([(mask >> (i // block)) & 1 for i in range(bits.numel())], dtype=torch.bool)
        idx = torch.nonzero(((bits & 0x7fff7fff) != 0) & sel).flatten()
        idx = idx[torch.randperm(idx.numel())]            # the device list has no particular order
        k = min(idx.numel(), pairs.shape[0])
        pairs[:k, 0] = idx[:k].to(torch.int32)
        pairs[:k, 1] = bits[idx[:k]]
        count[0] = idx.numel()
        bits[sel] = 0

    @staticmethod
    def add(values, pairs, n):
        v = values.view(-1, 2)
        idx = pairs[:n, 0].long()
        add = pairs[:n, 1].contiguous().view(torch.float16).view(-1, 2)
        v[idx] = v[idx] + add


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gradient(rank, rnd):
    """fp16[2 * NB * BLOCK]: block 0 dense on every rank, block 1 sparse, block 2 sparse on rank 0 and dense on the last rank
    (-> dense), block 3 empty everywhere, block 4 sparse with a negative zero in it; round 1: rank 1 holds only zeros; round 2:
    nobody holds anything."""
    g = np.random.default_rng(100 * rnd + rank)
    v = np.zeros((NB * BLOCK, 2), np.float16)
    if rnd == 2 or (rnd == 1 and rank == 1):
        return torch.from_numpy(v.reshape(-1))
    v[:BLOCK] = g.standard_normal((BLOCK, 2)).astype(np.float16)
    for b, k in ((1, 9), (2, 7 if rank == 0 else BLOCK - 3), (4, 5)):
        at = g.choice(BLOCK, k, replace=False) + b * BLOCK
        v[at] = g.standard_normal((k, 2)).astype(np.float16)
    v[4 * BLOCK + 1, 0] = -0.0
    return torch.from_numpy(v.reshape(-1))


def _worker(rank, world, port, force, out):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        res = []
        values = torch.zeros(2 * NB * BLOCK, dtype=torch.float16)
        ex = Half2GradExchange(values, BLOCK, ops=CpuOps, force_lists=force)
        for rnd in range(3):
            values.copy_(_gradient(rank, rnd))
            for w in ex.exchange():
                w.wait()
            res.append((values.clone(), dict(ex.last)))
        if rank == 0:
            torch.save(res, out)
        ref = [r[0].clone() for r in res]
        for t in ref:
            dist.broadcast(t, src=0)
        for t, (mine, _) in zip(ref, res):
            assert torch.equal(t.view(torch.int16), mine.view(torch.int16)), f"rank {rank} differs from rank 0"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,force", [(2, False), (3, False), (2, True)])
def test_half2_exchange_sums_identically_on_every_rank(tmp_path, world, force):
    out = str(tmp_path / "r.pt")
    mp.spawn(_worker, args=(world, _free_port(), force, out), nprocs=world, join=True)
    res = torch.load(out)
    for rnd, (got, last) in enumerate(res):
        parts = [_gradient(r, rnd) for r in range(world)]
        want = torch.zeros_like(parts[0])
        for p in parts:                                   # rank order, fp16 adds: what the lists give bit for bit
            want = want + p
        g, w = got.view(-1, 2), want.view(-1, 2)
        sparse_blocks = list(range(NB)) if force else [1, 3, 4]
        for b in sparse_blocks:
            assert torch.equal(g[b * BLOCK:(b + 1) * BLOCK], w[b * BLOCK:(b + 1) * BLOCK]), (rnd, b)
        if world == 2:
            assert torch.equal(g, w)                      # two terms: the ring's order cannot differ either
        else:
            torch.testing.assert_close(g.float(), sum(p.float() for p in parts).view(-1, 2), rtol=0, atol=2e-2)
        if rnd == 0 and not force:
            assert last["sparse_blocks"] == 3 and last["dense_blocks"] == 2
            assert last["bytes_lists"] == (world - 1) * 8.0 * max(last["list_entries"]) and max(last["list_entries"]) == 14
            assert last["bytes"] < 2.0 * (world - 1) / world * 4 * NB * BLOCK      # less than all blocks dense
        if rnd == 2:
            assert not got.any() and last["bytes_lists"] == 0


def test_choice_follows_the_wire_cost():
    ex = Half2GradExchange(torch.zeros(2 * 1000, dtype=torch.float16), 500, ops=CpuOps)
    # N x max count < entries -> lists
    assert ex.choose([[124, 125], [10, 0]]) == [0, 1]
    assert ex.choose([[249, 250], [10, 0]]) == [0]
    assert ex.choose([[62, 63]] * 8) == [0]               # 8 x 62 = 496 < 500, 8 x 63 = 504
