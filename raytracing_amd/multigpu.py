"""Scanline-band sharding of one frame over the GPUs of a node (BASELINE.json north_star).

Pixels are independent in the reference's render loop (src/render.rs:248-306), so the frame is cut
into 16-row blocks dealt round-robin to the ranks (sky rows are cheap, diorama rows are not); each
rank renders its blocks into one compact buffer (rt_render_device_part) and the only exchange step is
the gather of those buffers to rank 0, which scatters the rows back into frame order.  The functions
here are backend-agnostic (RCCL on GPUs, gloo in the CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist

ROW_BLOCK = 16  # kRowBlock in csrc/rtx950_device.h


def part_row_indices(rows, n_parts, part, block=ROW_BLOCK):
    """Frame rows (band-relative) held by `part`, in the order of its compact buffer."""
    idx = []
    for b in range(part, (rows + block - 1) // block, n_parts):
        idx.extend(range(b * block, min(rows, (b + 1) * block)))
    return np.asarray(idx, dtype=np.int64)


def part_rows(rows, n_parts, part, block=ROW_BLOCK):
    return int(part_row_indices(rows, n_parts, part, block).size)


def max_part_rows(rows, n_parts, block=ROW_BLOCK):
    return max(part_rows(rows, n_parts, p, block) for p in range(n_parts))


class FrameGather:
    """Gathers the ranks' compact bands to rank 0 and reassembles the frame there.

    Every rank passes a [max_part_rows, width] int32 tensor (padded so all messages have the same
    size); rank 0 owns the [rows, width] frame."""

    def __init__(self, rows, width, world, rank, device, group=None):
        self.rows, self.width, self.world, self.rank, self.group = rows, width, world, rank, group
        self.max_rows = max_part_rows(rows, world)
        self.local = torch.zeros((self.max_rows, width), dtype=torch.int32, device=device)
        self.frame = None
        self.recv = None
        self.index = None
        if rank == 0:
            self.frame = torch.zeros((rows, width), dtype=torch.int32, device=device)
            self.recv = [torch.zeros_like(self.local) for _ in range(world)] if world > 1 else None
            self.index = [torch.from_numpy(part_row_indices(rows, world, p)).to(device) for p in range(world)]

    def gather(self):
        """Collective: after it returns (stream-ordered), rank 0's `frame` holds the whole frame."""
        if self.world > 1:
            dist.gather(self.local, self.recv, dst=0, group=self.group)
        if self.rank == 0:
            parts = self.recv if self.world > 1 else [self.local]
            for p, buf in enumerate(parts):
                n = int(self.index[p].numel())
                if n:
                    self.frame.index_copy_(0, self.index[p], buf[:n])
        return self.frame
