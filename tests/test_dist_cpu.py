"""N>1 path on CPU: world_size-2 gloo run of the band partition + gather + reassembly (no GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import multigpu  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, rows, width, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = multigpu.FrameGather(rows, width, world, rank, torch.device("cpu"))
    # stand-in for rt_render_device_part: pixel value encodes (frame row, column)
    idx = multigpu.part_row_indices(rows, world, rank)
    g.local.zero_()
    g.local[: len(idx)] = torch.from_numpy((idx[:, None] * 100000 + np.arange(width)[None, :]).astype(np.int32))
    frame = g.gather()
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rows,width,world", [(2160, 64, 2), (45, 7, 2), (16, 3, 3)])
def test_gather_reassembles_frame(tmp_path, rows, width, world):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), rows, width, out), nprocs=world, join=True)
    frame = np.load(out)
    expect = (np.arange(rows)[:, None] * 100000 + np.arange(width)[None, :]).astype(np.int32)
    assert np.array_equal(frame, expect)


def test_partition_is_a_partition():
    for rows in (1, 15, 16, 17, 720, 2160, 4320):
        for n in (1, 2, 3, 4, 8):
            allrows = np.concatenate([multigpu.part_row_indices(rows, n, p) for p in range(n)])
            assert sorted(allrows.tolist()) == list(range(rows))
            sizes = [multigpu.part_rows(rows, n, p) for p in range(n)]
            assert max(sizes) - min(sizes) <= 16
