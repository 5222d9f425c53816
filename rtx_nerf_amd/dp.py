"""Data-parallel exchange of the hash-grid gradient (the reference is single-GPU, main.cu has no counterpart).

One rank's batch touches a small part of every hashed level -- 0.2 % of the entries late in training, 1 .. 25 % in the first
steps (configs[2], 4096 rays; tools/probe/hash_grad_density.py) -- because the scatter visits only the segments that carry a
loss gradient.  So a level is summed across the ranks in whichever form is smaller on the wire:

  dense   all-reduce of the level's fp16 gradient:                 2 (N-1)/N x 4 B x entries      per rank (ring)
  sparse  all-gather of (entry index, half2) lists, then every
          rank adds all N lists into its cleared level:            (N-1) x 8 B x max_r count_r    per rank (ring)

sparse wins where N x max_r count_r < entries.  The ranks agree on the form per level from the all-gathered counts (one
host read of N x levels integers per step: NCCL has to be told the list length); the lists are padded to the longest.
Every rank adds the lists in rank order (its own included) into zeroed entries, so all ranks hold bit-identical sums.

The three device primitives (count / pack / add) are librtxn's rtxn_half2_* entry points; `ops` lets the CPU tests
(gloo, no GPU in the build container) drive the same exchange logic with stand-ins.
"""
import torch
import torch.distributed as dist


class _HipOps:
    """rtxn_half2_count_nonzero / rtxn_half2_pack_nonzero / rtxn_half2_add_pairs on the current stream."""

    @staticmethod
    def workspace(values, block_entries):
        from . import api
        return api.half2_workspace(values, block_entries)

    @staticmethod
    def count(values, block_entries, ws):
        from . import api
        api.half2_count_nonzero(values, block_entries, ws)

    @staticmethod
    def pack(values, block_entries, ws, mask, pairs, count):
        from . import api
        api.half2_pack_nonzero(values, block_entries, ws, mask, pairs, count, clear=True)

    @staticmethod
    def add(values, pairs, n):
        from . import api
        api.half2_add_pairs(values, pairs, n)


class Half2GradExchange:
    """Sums `values` (fp16 tensor, two halves per entry, blocks of `block_entries` entries = hash-grid levels) over the ranks."""

    def __init__(self, values, block_entries, ops=None, group=None, force_lists=False):
        assert values.dtype == torch.float16 and values.numel() % 2 == 0
        self.values, self.block = values, int(block_entries)
        self.n = values.numel() // 2
        self.nb = (self.n + self.block - 1) // self.block
        if self.nb > 64:
            raise ValueError(f"Half2GradExchange: {self.nb} blocks (the pack entry point takes a 64-bit block mask)")
        self.ops = ops or _HipOps
        self.group = group
        self.force_lists = force_lists     # tests: every block as lists whatever the counts say
        dev = values.device
        self.ws = self.ops.workspace(values, self.block)     # int32: per-block counts first, then what pack needs of count
        self.counts = self.ws[:self.nb]
        self.count1 = torch.zeros(1, dtype=torch.int32, device=dev)
        self.pairs = None            # int32[cap][2], grown on demand
        self.gathered = None         # int32[world][cap][2]
        self.last = None             # what the last exchange did (bytes per rank, ring model; blocks in each form)

    def _entries(self, b):
        return min(self.block, self.n - b * self.block)

    def choose(self, counts_by_rank):
        """counts_by_rank: int[world][nb] (host) -> list of blocks that go as lists; the rest go dense."""
        world = len(counts_by_rank)
        sparse = []
        for b in range(self.nb):
            mx = max(int(counts_by_rank[r][b]) for r in range(world))
            if self.force_lists or world * mx < self._entries(b):
                sparse.append(b)
        return sparse

    def exchange(self):
        """Returns the async work handles of the dense part (wait on them before reading `values`); the sparse part is
        complete -- in stream order -- when this returns."""
        world = dist.get_world_size(self.group)
        self.ops.count(self.values, self.block, self.ws)
        lst = [torch.empty_like(self.counts) for _ in range(world)]
        dist.all_gather(lst, self.counts, group=self.group)
        c = torch.stack(lst).cpu().tolist()                      # the one host read: list lengths
        sparse = self.choose(c)
        sp = set(sparse)
        pending = []
        dense_bytes = 0
        b = 0
        while b < self.nb:                                       # contiguous runs of dense blocks: one all-reduce each
            if b in sp:
                b += 1
                continue
            e = b
            while e + 1 < self.nb and (e + 1) not in sp:
                e += 1
            lo, hi = 2 * b * self.block, 2 * min(self.n, (e + 1) * self.block)
            pending.append(dist.all_reduce(self.values[lo:hi], async_op=True, group=self.group))
            dense_bytes += 2 * (hi - lo)
            b = e + 1
        need = [sum(c[r][b] for b in sparse) for r in range(world)]
        cap = max(need) if sparse else 0
        if cap > 0:
            if self.pairs is None or self.pairs.shape[0] < cap:
                grow = max(cap + cap // 2, 1024)
                self.pairs = torch.empty((grow, 2), dtype=torch.int32, device=self.values.device)
                self.gathered = torch.empty((world, grow, 2), dtype=torch.int32, device=self.values.device)
            mask = 0
            for b in sparse:
                mask |= 1 << b
            mine = self.pairs[:cap]
            self.ops.pack(self.values, self.block, self.ws, mask, mine, self.count1)
            parts = [self.gathered[r, :cap] for r in range(world)]
            dist.all_gather(parts, mine, group=self.group)
            for r in range(world):                               # rank order on every rank: bit-identical sums
                if need[r]:
                    self.ops.add(self.values, parts[r], need[r])
        f = (world - 1) / world
        self.last = dict(world=world, sparse_blocks=len(sparse), dense_blocks=self.nb - len(sparse), list_entries=need,
                         bytes_dense=2.0 * f * dense_bytes, bytes_lists=(world - 1) * 8.0 * cap,
                         bytes_counts=(world - 1) * 4.0 * self.nb)
        self.last["bytes"] = self.last["bytes_dense"] + self.last["bytes_lists"] + self.last["bytes_counts"]
        return pending
