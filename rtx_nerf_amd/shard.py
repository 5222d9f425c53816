"""Ray sharding of one W x H launch across the GPUs of a node (host logic only).

Rays are independent (optixPrograms.cu:45,184,241: every thread touches only
its own output slots), so a frame is sharded with no data-path collective:
rank r owns image rows r, r+world, r+2*world, ...  Round-robin rows rather than
contiguous bands because occupancy is concentrated in the middle rows.  The
shard is expressed to the traversal kernel as the strided ray window of
rtxn_trace_params (window_chunk = W rays, window_stride = world*W rays).  The
rendered rows are collected on rank 0 with ONE gather per frame
(torch.distributed: RCCL on GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist


class RowShard:
    def __init__(self, width, height, rank, world):
        self.W, self.H, self.rank, self.world = width, height, rank, world
        self.rows = list(range(rank, height, world))
        self.n_local = len(self.rows) * width
        self.ray_begin = rank * width if world > 1 else 0
        self.window = (width, world * width) if world > 1 else (0, 0)
        self.n_max = ((height + world - 1) // world) * width   # padded shard size (equal on every rank)

    def local_to_global(self, i):
        """launch ray id of local ray i (the kernel's formula)."""
        if self.window[0] == 0:
            return self.ray_begin + i
        c, s = self.window
        return self.ray_begin + (i // c) * s + (i % c)

    def gather(self, local_pixels, gather_list=None, async_op=False):
        """local_pixels: [n_max, 3] (rows beyond n_local are padding).  Rank 0 passes gather_list."""
        if self.world == 1:
            return None
        return dist.gather(local_pixels, gather_list if self.rank == 0 else None, dst=0, async_op=async_op)

    def assemble(self, gather_list):
        """rank 0: list of world tensors [n_max, 3] -> image [H, W, 3]."""
        img = torch.empty((self.H, self.W, 3), dtype=gather_list[0].dtype, device=gather_list[0].device)
        for r, buf in enumerate(gather_list):
            nrows = len(range(r, self.H, self.world))
            img[r::self.world] = buf[:nrows * self.W].reshape(nrows, self.W, 3)
        return img
