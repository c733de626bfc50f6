"""Scanline-block sharding of one frame over the GPUs of a node (BASELINE.json north_star).

Pixels are independent in the reference's render loop (src/render.rs:248-306), so the frame is cut
into 16-row blocks dealt round-robin to the ranks (sky rows are cheap, diorama rows are not); each
rank renders its blocks into one compact buffer (rt_render_device_part) and the only exchange step is
the gather of those buffers to rank 0, which scatters the rows back into frame order.

Two things keep that exchange off the critical path:
  * only the rows that can hold geometry travel (rt_frame_row_extent, a host-side bound every rank
    computes identically): rank 0 fills the other rows with the sky colour itself;
  * FramePipeline keeps two frames in flight per rank (two message buffers, two library contexts, each
    frame's render, gather and scatter in order on its slot's stream): frame f+1 renders while frame f
    is gathered.
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


def message_window(rows, n_parts, extent, block=ROW_BLOCK, max_rows=None):
    """Rows [a, b) of every rank's compact buffer that must travel so that all frame rows of
    `extent` = (y_first, y_end) reach rank 0, and the frame rows [fa, fb) those messages cover.

    Compact block k of rank p is frame block k * n_parts + p, so frame blocks [B0, B1) need compact
    blocks floor(B0 / n) .. ceil(B1 / n) - 1 on some rank; the window is the same on every rank (equal
    message sizes).  An empty extent gives an empty window: no message at all."""
    if max_rows is None:
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
    messages have the same size); rank 0 owns a [rows, width] frame and a receive buffer per slot
    (`frames[slot]`, a swap chain: frames in flight never share memory).  `unpainted` is the
    band-relative (row, column) of the reference's never-painted pixel i == width (render.rs:316-322),
    or None when the band does not hold it: it matters only when rank 0 fills that row itself."""

    def __init__(self, rows, width, world, rank, device, group=None, depth=1, unpainted=(1, 0)):
        self.rows, self.width, self.world, self.rank, self.group = rows, width, world, rank, group
        self.max_rows = max_part_rows(rows, world)
        self.locals = [torch.zeros((self.max_rows, width), dtype=torch.int32, device=device) for _ in range(depth)]
        self.local = self.locals[0]
        self.unpainted = unpainted if (unpainted is not None and unpainted[0] < rows and unpainted[1] < width) else None
        self.frames = None
        self.frame = None  # rank 0: the frame of the latest gather() call
        # gloo moves host memory only: with device buffers (rehearsals of the N-GPU path on one GPU) stage through the host
        self._host_staging = world > 1 and torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"
        self._plans = {}
        self._windows = {}
        self._filled = [None] * depth  # per slot (fa, fb, sky): rows outside [fa, fb) hold this sky colour already
        if rank == 0:
            # one spare row takes the padding rows of the messages, so the scatter is ONE index_copy_
            self._frame_bufs = [torch.zeros((rows + 1, width), dtype=torch.int32, device=device) for _ in range(depth)]
            self.frames = [b[:rows] for b in self._frame_bufs]
            self.frame = self.frames[0]
            self._recv_flat = [torch.zeros(world * self.max_rows * width, dtype=torch.int32, device=device) if world > 1 else None
                               for _ in range(depth)]
            pad = np.full((world, self.max_rows), rows, dtype=np.int64)
            for p in range(world):
                idx = part_row_indices(rows, world, p)
                pad[p, : idx.size] = idx
            self._index_pad = pad

    def _plan(self, slot, a, b):
        """Receive views and scatter index for the message window [a, b) of a slot (cached)."""
        plan = self._plans.get((slot, a, b))
        if plan is None:
            m = b - a
            index = self._plans.get(("index", a, b))
            if index is None:
                index = torch.from_numpy(np.ascontiguousarray(self._index_pad[:, a:b]).reshape(-1)).to(self._frame_bufs[0].device)
                self._plans[("index", a, b)] = index
            if self.world > 1:
                block = self._recv_flat[slot][: self.world * m * self.width].view(self.world, m, self.width)
                plan = (index, [block[p] for p in range(self.world)], block.view(self.world * m, self.width))
            else:
                plan = (index, None, None)
            self._plans[(slot, a, b)] = plan
        return plan

    def gather(self, slot=0, extent=None, sky=None):
        """Collective: after it returns (stream-ordered), rank 0's `frames[slot]` (= `frame`) holds the
        whole frame.

        extent = (y_first, y_end) from rt_frame_row_extent and sky = the frame's sky colour: rows outside
        the extent are not sent, rank 0 paints them.  Every rank must pass the same extent."""
        if extent is not None and sky is None:
            raise ValueError("an extent needs the sky colour for the rows that are not sent")
        key = None if extent is None else (int(extent[0]), int(extent[1]))
        win = self._windows.get(key)
        if win is None:  # per-frame host work is a dictionary look-up
            win = self._windows[key] = message_window(self.rows, self.world, key, max_rows=self.max_rows)
        a, b, fa, fb = win
        local = self.locals[slot]
        if self.rank != 0:
            if b > a:
                dist.gather(local[a:b].cpu() if self._host_staging else local[a:b], None, dst=0, group=self.group)
            return None
        buf = self._frame_bufs[slot]
        if (fa, fb) != (0, self.rows):
            if self._filled[slot] != (fa, fb, sky):
                buf[:fa].fill_(sky)
                buf[fb : self.rows].fill_(sky)
                if self.unpainted is not None and not (fa <= self.unpainted[0] < fb):
                    buf[self.unpainted[0], self.unpainted[1]] = 0
                self._filled[slot] = (fa, fb, sky)
        else:
            self._filled[slot] = None
        if b > a:
            index, views, flat = self._plan(slot, a, b)
            if self.world > 1:
                if self._host_staging:
                    mine = local[a:b].cpu()
                    parts = [torch.empty_like(mine) for _ in range(self.world)]
                    dist.gather(mine, parts, dst=0, group=self.group)
                    flat.copy_(torch.cat(parts))
                else:
                    dist.gather(local[a:b], views, dst=0, group=self.group)
                buf.index_copy_(0, index, flat)
            else:
                buf.index_copy_(0, index, local[a:b])
        self.frame = self.frames[slot]
        return self.frame


class FramePipeline:
    """Keeps `depth` frames in flight per rank: frame f+1 renders while frame f is gathered - and while
    frame f's slowest tiles are still being traced.  A rank of an N-GPU job holds 1/N of the frame's
    work but all of its critical path (the slowest 8x8 tile takes 80 % of the whole frame's kernel time
    by itself), so one frame in flight leaves its GPU idle most of the time.

    Slot s = f mod depth owns a message buffer, a library context with its stream, and on rank 0 a
    receive buffer and a frame (FrameGather.frames[s]).  Everything of a frame - render, collective,
    rank 0's fill and scatter - is enqueued on its slot's stream, in that order, so a slot needs no
    events: its next frame queues up behind; different slots overlap.  (RCCL runs the collective on its
    own stream, fenced both ways against the stream it is called on, and executes collectives in call
    order, which is the same on every rank.)

    render(buffer, slot) enqueues this rank's part of the next frame on render_streams[slot] (torch
    streams wrapping the library's: torch.cuda.ExternalStream(ctx.stream())).  On a CPU device (gloo
    tests, render_streams=None) everything runs in program order."""

    def __init__(self, gather: FrameGather, render, render_streams=None, time_every=0):
        self.g, self.render = gather, render
        self.depth = len(gather.locals)
        self.n = 0
        # time_every = k > 0: every k-th frame's exchange (collective + rank 0's fill and scatter) is bracketed
        # by events on its slot's stream (gather_stats)
        self.time_every = int(time_every)
        self._timed = []
        self.last_window_rows = 0
        self.cuda = render_streams is not None
        if self.cuda:
            if len(render_streams) != self.depth or len(set(s.cuda_stream for s in render_streams)) != self.depth:
                raise ValueError("one render stream per message buffer")
            self.render_streams = list(render_streams)

    def submit(self, extent=None, sky=None):
        """Returns the slot: on rank 0 the frame will be FrameGather.frames[slot]."""
        slot = self.n % self.depth
        self.render(self.g.locals[slot], slot)
        if self.cuda:
            timed = self.time_every > 0 and self.n % self.time_every == 0
            with torch.cuda.stream(self.render_streams[slot]):
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                self.g.gather(slot, extent, sky)
                if timed:
                    e1.record()
                    self._timed.append((e0, e1))
        else:
            self.g.gather(slot, extent, sky)
        a, b, _, _ = message_window(self.g.rows, self.g.world, None if extent is None else (int(extent[0]), int(extent[1])), max_rows=self.g.max_rows)
        self.last_window_rows = b - a
        self.n += 1
        return slot

    def gather_stats(self):
        """(average ms of the timed exchanges since the last call, their number, bytes this rank sent per
        frame in the last frame's message window).  Waits for the timed frames."""
        ms = []
        for e0, e1 in self._timed:
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        self._timed = []
        sent = 0 if self.g.rank == 0 else self.last_window_rows * self.g.width * 4
        return (sum(ms) / len(ms) if ms else 0.0), len(ms), sent

    def drain(self):
        """Host wait until every submitted frame has reached rank 0."""
        if self.cuda:
            for rs in self.render_streams:
                rs.synchronize()
