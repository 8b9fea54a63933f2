"""world_size-2/3 gloo tests of the window sharding + curve reassembly (the N>1 path of the harness)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_windows, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mpv_amd import distributed as D
        from oracle import nsdiff_oracle as O
        assert D.active() and D.world_size() == world and D.rank() == rank
        lo, hi = D.shard_bounds(n_windows)
        # a per-window computation that depends only on the window index (stands in for the sampler):
        g = torch.Generator().manual_seed(5)
        gx_all = torch.rand(n_windows, 3, 7, 1, generator=g) + 0.1
        local = [O.gx_summary(gx_all[w]) for w in range(lo, hi)]
        a = torch.tensor([v[0] for v in local], dtype=torch.float32)
        b = torch.tensor([v[1] for v in local], dtype=torch.float32)
        full_a, full_b = D.gather_curve(n_windows, lo, hi, a, b)
        want = [O.gx_summary(gx_all[w]) for w in range(n_windows)]
        np.testing.assert_allclose(full_a, np.array([v[0] for v in want], dtype=np.float32), rtol=0, atol=0)
        np.testing.assert_allclose(full_b, np.array([v[1] for v in want], dtype=np.float32), rtol=0, atol=0)
        np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_windows", [(2, 181), (2, 1), (3, 10), (2, 0)])
def test_sharded_curve_equals_unsharded(tmp_path, world, n_windows):
    mp.spawn(_worker, args=(world, _free_port(), n_windows, str(tmp_path)), nprocs=world, join=True)
    bounds = [np.load(tmp_path / ("r%d.npy" % r)) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n_windows
    for a, b in zip(bounds[:-1], bounds[1:]):
        assert a[1] == b[0]                                                # contiguous, no gaps/overlap
    sizes = [int(b[1] - b[0]) for b in bounds]
    assert max(sizes) - min(sizes) <= 1


def test_block_bounds_cover_everything():
    from mpv_amd import distributed as D
    for n in (0, 1, 7, 181, 1000):
        for world in (1, 2, 3, 8):
            spans = [D.block_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
    assert not D.active() and D.rank() == 0 and D.world_size() == 1 and D.shard_bounds(5) == (0, 5)
