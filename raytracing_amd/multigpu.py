"""Scanline-block sharding of one frame over the GPUs of a node (BASELINE.json north_star).

Pixels are independent in the reference's render loop (src/render.rs:248-306), so the frame is cut
into 16-row blocks dealt round-robin to the ranks (sky rows are cheap, diorama rows are not); each
rank renders its blocks into one compact buffer (rt_render_device_part) and the only exchange step is
the gather of those buffers to rank 0, which scatters the rows back into frame order.

Two things keep that exchange off the critical path:
  * only the rows that can hold geometry travel (rt_frame_row_extent, a host-side bound every rank
    computes identically): rank 0 fills the other rows with the sky colour itself;
  * FramePipeline renders frame f+1 while frame f is being gathered (two message buffers, the gather
    on its own stream).
The functions here are backend-agnostic (RCCL on GPUs, gloo in the CPU tests).
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


def message_window(rows, n_parts, extent, block=ROW_BLOCK):
    """Rows [a, b) of every rank's compact buffer that must travel so that all frame rows of
    `extent` = (y_first, y_end) reach rank 0, and the frame rows [fa, fb) those messages cover.

    Compact block k of rank p is frame block k * n_parts + p, so frame blocks [B0, B1) need compact
    blocks floor(B0 / n) .. ceil(B1 / n) - 1 on some rank; the window is the same on every rank (equal
    message sizes).  An empty extent gives an empty window: no message at all."""
    max_rows = max_part_rows(rows, n_parts, block)
    if extent is None:
        return 0, max_rows, 0, rows
    y_first, y_end = max(0, int(extent[0])), min(rows, int(extent[1]))
    if y_first >= y_end:
        return 0, 0, 0, 0
    b0, b1 = y_first // block, (y_end + block - 1) // block
    k0, k1 = b0 // n_parts, (b1 + n_parts - 1) // n_parts
    a, b = min(k0 * block, max_rows), min(k1 * block, max_rows)
    return a, b, min(k0 * n_parts * block, rows), min(k1 * n_parts * block, rows)


class FrameGather:
    """Gathers the ranks' compact bands to rank 0 and reassembles the frame there.

    Every rank owns `depth` message buffers `locals[slot]`, [max_part_rows, width] int32 (padded so all
    messages have the same size); rank 0 owns the [rows, width] frame.  `unpainted` is the
    band-relative (row, column) of the reference's never-painted pixel i == width (render.rs:316-322),
    or None when the band does not hold it: it matters only when rank 0 fills that row itself."""

    def __init__(self, rows, width, world, rank, device, group=None, depth=1, unpainted=(1, 0)):
        self.rows, self.width, self.world, self.rank, self.group = rows, width, world, rank, group
        self.max_rows = max_part_rows(rows, world)
        self.locals = [torch.zeros((self.max_rows, width), dtype=torch.int32, device=device) for _ in range(depth)]
        self.local = self.locals[0]
        self.unpainted = unpainted if (unpainted is not None and unpainted[0] < rows and unpainted[1] < width) else None
        self.frame = None
        # gloo moves host memory only: with device buffers (rehearsals of the N-GPU path on one GPU) stage through the host
        self._host_staging = world > 1 and torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"
        self._plans = {}
        self._filled = None  # (fa, fb, sky): rows outside [fa, fb) hold this sky colour already
        if rank == 0:
            # one spare row takes the padding rows of the messages, so the scatter is ONE index_copy_
            self._frame_buf = torch.zeros((rows + 1, width), dtype=torch.int32, device=device)
            self.frame = self._frame_buf[:rows]
            self._recv_flat = torch.zeros(world * self.max_rows * width, dtype=torch.int32, device=device) if world > 1 else None
            pad = np.full((world, self.max_rows), rows, dtype=np.int64)
            for p in range(world):
                idx = part_row_indices(rows, world, p)
                pad[p, : idx.size] = idx
            self._index_pad = pad

    def _plan(self, a, b):
        """Receive views and scatter index for the message window [a, b) (cached per window)."""
        plan = self._plans.get((a, b))
        if plan is None:
            m = b - a
            index = torch.from_numpy(np.ascontiguousarray(self._index_pad[:, a:b]).reshape(-1)).to(self._frame_buf.device)
            if self.world > 1:
                block = self._recv_flat[: self.world * m * self.width].view(self.world, m, self.width)
                plan = (index, [block[p] for p in range(self.world)], block.view(self.world * m, self.width))
            else:
                plan = (index, None, None)
            self._plans[(a, b)] = plan
        return plan

    def gather(self, slot=0, extent=None, sky=None):
        """Collective: after it returns (stream-ordered), rank 0's `frame` holds the whole frame.

        extent = (y_first, y_end) from rt_frame_row_extent and sky = the frame's sky colour: rows outside
        the extent are not sent, rank 0 paints them.  Every rank must pass the same extent."""
        if extent is not None and sky is None:
            raise ValueError("an extent needs the sky colour for the rows that are not sent")
        a, b, fa, fb = message_window(self.rows, self.world, extent)
        local = self.locals[slot]
        if self.rank != 0:
            if b > a:
                dist.gather(local[a:b].cpu() if self._host_staging else local[a:b], None, dst=0, group=self.group)
            return None
        if (fa, fb) != (0, self.rows):
            if self._filled != (fa, fb, sky):
                self._frame_buf[:fa].fill_(sky)
                self._frame_buf[fb : self.rows].fill_(sky)
                if self.unpainted is not None and not (fa <= self.unpainted[0] < fb):
                    self._frame_buf[self.unpainted[0], self.unpainted[1]] = 0
                self._filled = (fa, fb, sky)
        else:
            self._filled = None
        if b > a:
            index, views, flat = self._plan(a, b)
            if self.world > 1:
                if self._host_staging:
                    mine = local[a:b].cpu()
                    parts = [torch.empty_like(mine) for _ in range(self.world)]
                    dist.gather(mine, parts, dst=0, group=self.group)
                    flat.copy_(torch.cat(parts))
                else:
                    dist.gather(local[a:b], views, dst=0, group=self.group)
                self._frame_buf.index_copy_(0, index, flat)
            else:
                self._frame_buf.index_copy_(0, index, local[a:b])
        return self.frame


class FramePipeline:
    """Renders frame f+1 while frame f is gathered: message buffers alternate, the gather (and rank 0's
    fill and scatter) run on their own stream, events order buffer reuse.

    render(buffer) enqueues this rank's part of the next frame on `render_stream` (a torch stream
    wrapping the library's: torch.cuda.ExternalStream(ctx.stream())).  On a CPU device (gloo tests)
    everything runs in program order."""

    def __init__(self, gather: FrameGather, render, render_stream=None):
        self.g, self.render, self.render_stream = gather, render, render_stream
        self.depth = len(gather.locals)
        self.n = 0
        self.cuda = render_stream is not None
        if self.cuda:
            dev = render_stream.device
            self.comm_stream = torch.cuda.Stream(device=dev)
            self.rendered = [torch.cuda.Event() for _ in range(self.depth)]
            self.gathered = [torch.cuda.Event() for _ in range(self.depth)]

    def submit(self, extent=None, sky=None):
        slot = self.n % self.depth
        if self.cuda and self.n >= self.depth:
            self.render_stream.wait_event(self.gathered[slot])  # the buffer's previous message has left
        self.render(self.g.locals[slot])
        if self.cuda:
            self.rendered[slot].record(self.render_stream)
            self.comm_stream.wait_event(self.rendered[slot])
            with torch.cuda.stream(self.comm_stream):
                self.g.gather(slot, extent, sky)
                self.gathered[slot].record(self.comm_stream)
        else:
            self.g.gather(slot, extent, sky)
        self.n += 1

    def drain(self):
        """Host wait until every submitted frame has reached rank 0's frame."""
        if self.cuda:
            self.comm_stream.synchronize()
            self.render_stream.synchronize()
