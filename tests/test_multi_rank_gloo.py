"""World-size-2 CPU tests (gloo) of the multi-GPU path's host logic and protocol:
  * exchange_key_lists / merged_keys over a real process group;
  * the row-sharded frame protocol of supereight_amd/multi_gpu.py (scan own rows -> all-gather of
    [count, new keys, re-activated blocks] -> allocate the union -> replicated sweep) produces the same
    map as the unsharded reference, with the CPU oracle standing in for the per-rank engine.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ACTIVATE = np.uint64(1) << np.uint64(63)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _pack(keys, cap):
    buf = np.zeros(cap, np.uint64)
    buf[0] = len(keys)
    buf[1:1 + len(keys)] = keys
    return torch.from_numpy(buf.view(np.int64))


def _worker_exchange(rank, world, port, q):
    _init(rank, world, port)
    from supereight_amd.multi_gpu import exchange_key_lists, merged_keys
    rng = np.random.default_rng(100 + rank)
    mine = rng.integers(1, 2 ** 62, size=10 + 7 * rank, dtype=np.uint64)
    got = merged_keys(exchange_key_lists(_pack(mine, 64), world), world)
    q.put((rank, mine.tolist(), got.tolist()))
    # overflow is reported, never silently truncated
    big = _pack(mine, 64)
    big[0] = 1000
    try:
        merged_keys(exchange_key_lists(big, world), world)
        q.put((rank, "no-error", None))
    except OverflowError:
        q.put((rank, "overflow-detected", None))
    dist.destroy_process_group()


def test_exchange_key_lists_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exchange, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in range(4)]
    [p.join(60) for p in procs]
    lists = {r: (mine, got) for r, mine, got in res if isinstance(mine, list)}
    union = lists[0][0] + lists[1][0]
    assert lists[0][1] == union and lists[1][1] == union      # rank order, identical on every rank
    assert sorted(m for _, m, g in res if g is None) == ["overflow-detected", "overflow-detected"]


def _worker_protocol(rank, world, port, field, frames, q):
    _init(rank, world, port)
    from oracle.binding import OraclePipeline
    from supereight_amd.multi_gpu import exchange_key_lists, merged_keys, row_partition
    from supereight_amd.synthetic import SyntheticStream
    W, H, N, dim = 160, 120, 256, 2.4
    mu = 0.1 if field == 0 else 0.02
    rows = row_partition(H, world)[rank]
    stream = SyntheticStream(W, H, dim)
    o = OraclePipeline(field, N, dim, W, H)
    for f in range(frames):
        depth, pose = stream.depth(f), stream.pose(f)
        mine = np.zeros_like(depth)
        mine[rows[0]:rows[1]] = depth[rows[0]:rows[1]]           # a pixel with depth 0 is skipped by the scan
        c0, _, _, a0 = o.blocks()
        keys = np.unique(o.scan_keys(mine, pose, stream.k, mu, f))
        _, _, _, a1 = o.blocks()
        woke = c0[(a0 == 0) & (a1 == 1)]                         # blocks this rank's rays re-activated
        leaf = int(np.log2(N)) - 3
        woke_keys = np.array([o.lib.so_encode(int(x), int(y), int(z), leaf, int(np.log2(N))) for x, y, z in woke], np.uint64) | ACTIVATE
        msg = np.concatenate([keys, woke_keys]).astype(np.uint64)
        allk = merged_keys(exchange_key_lists(_pack(msg, 1 << 16), world), world)
        act = allk[(allk & ACTIVATE) != 0] & ~ACTIVATE
        new = allk[(allk & ACTIVATE) == 0]
        o.allocate_keys(new)
        if len(act):
            out = np.zeros(3, np.int32)
            coords = []
            for kk in act:
                o.lib.so_decode(int(kk), out)
                coords.append(out.copy())
            assert o.activate(np.array(coords)) == len(coords)
        o.sweep(depth, pose, stream.k, mu, f)
    c, x, y, a = o.blocks()
    code, side, nx, ny = o.nodes()
    q.put((rank, c, x, y, a, code, nx, ny))
    dist.destroy_process_group()


@pytest.mark.parametrize("field", [0, 1], ids=["sdf", "ofusion"])
def test_row_sharded_protocol_matches_unsharded_reference(field):
    frames = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_protocol, args=(r, 2, port, field, frames, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    [p.join(60) for p in procs]
    sys.path.insert(0, ROOT)
    from oracle.binding import OraclePipeline
    from supereight_amd.synthetic import SyntheticStream
    W, H, N, dim = 160, 120, 256, 2.4
    mu = 0.1 if field == 0 else 0.02
    s = SyntheticStream(W, H, dim)
    ref = OraclePipeline(field, N, dim, W, H)
    for f in range(frames):
        ref.integrate(s.depth(f), s.pose(f), s.k, mu, f)
    c, x, y, a = ref.blocks()
    code, side, nx, ny = ref.nodes()
    assert len(c) > 500
    for r in res:
        _, rc, rx, ry, ra, rcode, rnx, rny = r
        assert rc.shape == c.shape and (rc == c).all()
        assert (rx.view(np.uint32) == x.view(np.uint32)).all() and (ry.view(np.uint32) == y.view(np.uint32)).all()
        assert (ra == a).all()
        assert (rcode == code).all() and (rnx.view(np.uint32) == nx.view(np.uint32)).all() and (rny.view(np.uint32) == ny.view(np.uint32)).all()


def _worker_tiles(rank, world, port, H, q):
    _init(rank, world, port)
    import torch
    from supereight_amd.multi_gpu import gather_row_tiles, row_partition
    W = 40
    full = torch.arange(H * W * 3, dtype=torch.float32).reshape(H, W, 3)        # what the unsharded raycast would hold
    parts = row_partition(H, world)
    rows = parts[rank]
    mine = torch.full((H, W, 3), -1.0)
    mine[rows[0]:rows[1]] = full[rows[0]:rows[1]]                                # a rank's raycast fills its own rows only
    out = gather_row_tiles(mine, rows, parts)
    q.put((rank, bool(torch.equal(out, full)), [list(p) for p in parts]))
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [120, 116, 8])
def test_vertex_normal_row_tiles_all_gather_world2(H):
    """the ranks' raycast row tiles (unequal when the 8-row units do not divide by the world size; H = 8 leaves
    rank 0 with no rows at all) -> the full image on every rank, through a real gloo all-gather of fixed-size padded tiles."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_tiles, args=(r, 2, port, H, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in range(2)]
    [p.join(60) for p in procs]
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2] and res[0][2][-1][1] == H
