"""The N > 1 path on ONE GPU: R row-sharded replicas live in one process on the same device and
the all-gather is emulated by concatenating their key-list buffers.  Every replica must end up with
the map of the unsharded pipeline, and the stitched row tiles of their raycasts must equal its
vertex / normal images -- bit for bit."""
import numpy as np
import pytest

from oracle.binding import OFUSION, SDF
from supereight_amd.multi_gpu import row_partition
from supereight_amd.pipeline import DenseSLAMPipeline
from supereight_amd.synthetic import SyntheticStream

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("field,mu,R,H", [(SDF, 0.1, 2, 120), (SDF, 0.1, 4, 120), (OFUSION, 0.02, 2, 120), (SDF, 0.1, 8, 116), (OFUSION, 0.02, 8, 116),
                                          (SDF, 0.1, 4, -120), (OFUSION, 0.02, 3, -120)],
                         ids=["sdf-2", "sdf-4", "ofusion-2", "sdf-8-odd-height", "ofusion-8-odd-height", "sdf-4-stress", "ofusion-3-stress"])
@pytest.mark.parametrize("streaming", [False, True], ids=["two-queue", "one-queue"])
def test_sharded_replicas_equal_single(field, mu, R, H, streaming):
    # H = 116: 14.5 raycast tiles of 8 rows -> shards of 1 or 2 tile rows, the last one ending in a half tile
    # H < 0: the ICL-like stress stream (r03), every 3rd frame of its path: blocks leave the frustum of one rank's rows and are woken
    # by another rank's rays, hundreds of new keys per frame and rank, samples outside the volume
    import torch
    W, N, dim, frames = 160, 256, 2.4, 6
    dev = torch.device("cuda", 0)
    if H < 0:
        from supereight_amd.synthetic import StressStream
        H, frames, dim = -H, 24, 4.8

        class _Every3rd:
            def __init__(self):
                self.s = StressStream(W, H, dim)
                self.k = self.s.k

            def depth(self, f):
                d = None
                for g in range(3 * f - 2 if f else 0, 3 * f + 1):
                    d = self.s.depth(g)
                return d

            def pose(self, f):
                return self.s.pose(3 * f)
        stream = _Every3rd()
    else:
        stream = SyntheticStream(W, H, dim)
    single = DenseSLAMPipeline((W, H), N, dim, field_type=field)
    parts = row_partition(H, R)
    reps = [DenseSLAMPipeline((W, H), N, dim, field_type=field, rows=parts[r]) for r in range(R)]
    words = 1 << 15
    send = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(R)]
    for r in range(R):
        reps[r].set_new_keys_buffer(send[r].data_ptr(), words, keepalive=send[r])
        # one-queue (r05): a replica's raycast of frame f is held back and launched with its scan of frame f+1 (k_raycast_scan over the replica's rows,
        # the scan writing the send buffer); the gathered lists are committed behind that launch
        assert reps[r].set_streaming(streaming) == streaming
    for f in range(frames):
        depth, pose = stream.depth(f), stream.pose(f)
        single.set_depth(depth); single.setPose(pose)
        single.integration(stream.k, 1, mu, f)
        single.raycasting(stream.k, mu, f)
        for p in reps:
            p.set_depth(depth); p.setPose(pose)
            assert p.alloc_scan(stream.k, 1, mu, f)
        for p in reps:
            p.sync()
        recv = torch.cat(send)                       # what all_gather_into_tensor delivers on every rank
        torch.cuda.synchronize()                     # (the replicas run on their own streams)
        for p in reps:
            p.alloc_commit(recv.data_ptr(), R, words)
            p.integrate_sweep(stream.k, 1, mu, f)
            if streaming:
                assert p.raycasting_deferred(stream.k, mu, f) == (f > 2)
                assert p.launch_counts()["pending"] == (1 if f > 2 else 0)
            else:
                p.raycasting(stream.k, mu, f)
        if not streaming:
            for p in reps:
                p.sync()
    if streaming:
        for p in reps:
            assert p.launch_counts()["fused"] == frames - 4 and p.launch_counts()["raycast"] == frames - 4    # (the last raycast is still held back)
    c, x, y, a = single.blocks()
    code, side, nx, ny = single.nodes()
    v, n = single.vertex_normal()
    assert len(c) > 500 and (n[..., 0] != -2).sum() > 5000
    vs, ns = np.zeros_like(v), np.zeros_like(n)
    for r, p in enumerate(reps):
        rc, rx, ry, ra = p.blocks()
        assert rc.shape == c.shape and (rc == c).all()
        assert (rx.view(np.uint32) == x.view(np.uint32)).all() and (ry.view(np.uint32) == y.view(np.uint32)).all()
        assert (ra == a).all()
        rcode, rside, rnx, rny = p.nodes()
        assert (rcode == code).all() and (rnx.view(np.uint32) == nx.view(np.uint32)).all()
        rv, rn = p.vertex_normal()
        b, e = parts[r]
        vs[b:e], ns[b:e] = rv[b:e], rn[b:e]
        p.close()
    assert (vs.view(np.uint32) == v.view(np.uint32)).all() and (ns.view(np.uint32) == n.view(np.uint32)).all()
    single.close()


@pytest.mark.parametrize("field,mu,R,pooled", [(SDF, 0.1, 2, False), (SDF, 0.1, 8, False), (OFUSION, 0.02, 4, False), (SDF, 0.1, 4, True)],
                         ids=["sdf-2", "sdf-8", "ofusion-4", "sdf-4-pooled"])
def test_sharded_sweep_equals_single(field, mu, R, pooled):
    """SURVEY 8(e) option 4 behind its flag: every replica integrates only the blocks it owns and receives the others'
    bricks (the all-gather of the send segments is emulated by torch.cat, as for the key lists above).  Map and stitched
    raycast must equal the unsharded pipeline's, bit for bit -- also when the bricks live in the pool (slots differ
    between replicas, positions do not)."""
    import torch
    W, H, N, dim, frames = 160, 120, 256, 2.4, 6
    dev = torch.device("cuda", 0)
    stream = SyntheticStream(W, H, dim)
    mb = 20000 if pooled else 0
    single = DenseSLAMPipeline((W, H), N, dim, field_type=field, max_blocks=mb)
    parts = row_partition(H, R)
    reps = [DenseSLAMPipeline((W, H), N, dim, field_type=field, rows=parts[r], max_blocks=mb) for r in range(R)]
    words, cap = 1 << 15, 4096
    seg = reps[0].sweep_shard_bytes(cap)
    send = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(R)]
    bsend = [torch.zeros(seg, dtype=torch.uint8, device=dev) for _ in range(R)]
    for r in range(R):
        reps[r].set_new_keys_buffer(send[r].data_ptr(), words, keepalive=send[r])
        reps[r].set_sweep_shard(r, R, bsend[r].data_ptr(), cap, keepalive=bsend[r])
    packed = 0
    for f in range(frames):
        depth, pose = stream.depth(f), stream.pose(f)
        single.set_depth(depth); single.setPose(pose)
        single.integration(stream.k, 1, mu, f)
        single.raycasting(stream.k, mu, f)
        for p in reps:
            p.set_depth(depth); p.setPose(pose)
            assert p.alloc_scan(stream.k, 1, mu, f)
        for p in reps:
            p.sync()
        recv = torch.cat(send)
        torch.cuda.synchronize()
        for p in reps:
            p.alloc_commit(recv.data_ptr(), R, words)
            p.integrate_sweep(stream.k, 1, mu, f)
        for p in reps:
            p.sync()
        brecv = torch.cat(bsend)                     # what the all-gather of the segments delivers on every rank
        torch.cuda.synchronize()
        counts = torch.stack([b[:512].view(torch.int64) for b in bsend]).cpu().numpy()   # 64 record counters per segment
        assert counts.max() <= cap // 64 and counts.sum(axis=1).min() > 0, counts
        packed += int(counts.sum())
        for p in reps:
            p.apply_bricks(brecv.data_ptr(), R)
            p.raycasting(stream.k, mu, f)
        for p in reps:
            p.sync()
    c, x, y, a = single.blocks()
    code, side, nx, ny = single.nodes()
    v, n = single.vertex_normal()
    assert len(c) > 500 and packed > len(c)
    order = np.lexsort((c[:, 0], c[:, 1], c[:, 2]))
    vs, ns = np.zeros_like(v), np.zeros_like(n)
    for r, p in enumerate(reps):
        rc, rx, ry, ra = p.blocks()
        ro = np.lexsort((rc[:, 0], rc[:, 1], rc[:, 2]))
        assert rc.shape == c.shape and (rc[ro] == c[order]).all()
        assert (rx[ro].view(np.uint32) == x[order].view(np.uint32)).all() and (ry[ro].view(np.uint32) == y[order].view(np.uint32)).all()
        assert (ra[ro] == a[order]).all()
        rcode, rside, rnx, rny = p.nodes()
        assert (rcode == code).all() and (rnx.view(np.uint32) == nx.view(np.uint32)).all() and (rny.view(np.uint32) == ny.view(np.uint32)).all()
        rv, rn = p.vertex_normal()
        b, e = parts[r]
        vs[b:e], ns[b:e] = rv[b:e], rn[b:e]
        p.close()
    assert (vs.view(np.uint32) == v.view(np.uint32)).all() and (ns.view(np.uint32) == n.view(np.uint32)).all()
    single.close()


def test_brick_segment_overflow_is_reported():
    import torch
    from supereight_amd.pipeline import SeHipError
    W, H, N, dim, mu = 160, 120, 256, 2.4, 0.1
    stream = SyntheticStream(W, H, dim)
    p = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
    cap = 64                                        # rank 0 of 2 owns ~1000 of the blocks frame 0 brings into view
    bsend = torch.zeros(p.sweep_shard_bytes(cap), dtype=torch.uint8, device="cuda")
    p.set_sweep_shard(0, 2, bsend.data_ptr(), cap, keepalive=bsend)
    with pytest.raises(SeHipError, match="brick exchange segment overflow"):
        for f in range(3):
            p.set_depth(stream.depth(f)); p.setPose(stream.pose(f))
            p.integration(stream.k, 1, mu, f)
            p.sync()
    p.close()


def test_key_list_overflow_is_reported_on_the_frame_path():
    """A key list that does not fit its exchange buffer must not go unnoticed: the peers would miss blocks and the
    replicas would diverge.  The stage calls of the following frame fail with SE_HIP_E_CAPACITY."""
    import torch
    from supereight_amd.pipeline import SeHipError
    W, H, N, dim, mu = 160, 120, 256, 2.4, 0.1
    stream = SyntheticStream(W, H, dim)
    p = DenseSLAMPipeline((W, H), N, dim, field_type=SDF, rows=(0, 64))
    words = 64                                      # frame 0 allocates ~1000 blocks in these rows
    send = torch.zeros(words, dtype=torch.int64, device="cuda")
    p.set_new_keys_buffer(send.data_ptr(), words, keepalive=send)
    with pytest.raises(SeHipError, match="key list overflow"):
        for f in range(3):
            p.set_depth(stream.depth(f)); p.setPose(stream.pose(f))
            p.alloc_scan(stream.k, 1, mu, f)
            p.alloc_commit(send.data_ptr(), 1, words)
            p.integrate_sweep(stream.k, 1, mu, f)
            p.sync()
    p.close()


@pytest.mark.parametrize("streaming", [False, True], ids=["two-queue", "one-queue"])
@pytest.mark.parametrize("exchange", ["direct", "torch"])
def test_sharded_pipeline_stream_plan_single_rank_rccl(exchange, streaming, monkeypatch):
    """ShardedPipeline with a one-rank RCCL group: the exchange stream, the all-gather call and the
    commit of the gathered list run exactly as with R > 1 (the own list is committed again, a no-op);
    the result must equal the plain pipeline's."""
    import os
    import torch
    import torch.distributed as dist
    from supereight_amd.multi_gpu import ShardedPipeline
    W, H, N, dim, mu, frames = 160, 120, 256, 2.4, 0.1, 8
    stream = SyntheticStream(W, H, dim)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        single = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
        # "direct": ncclAllGather called by the C library on the process group's communicator;
        # "torch": torch.distributed's collective (the fallback)
        if exchange == "torch":
            monkeypatch.setenv("SE_EXCHANGE", "torch")
        sp = ShardedPipeline((W, H), N, dim, SDF, 0, 1, 0, exchange_always=True, streaming=streaming)
        assert sp.direct == (exchange == "direct") and sp.streaming == streaming
        depth = torch.from_numpy(np.stack([stream.depth(f) for f in range(frames)])).cuda()
        for f in range(frames):
            single.set_depth_device(depth[f].data_ptr()); single.setPose(stream.pose(f))
            single.integration(stream.k, 1, mu, f)
            single.raycasting(stream.k, mu, f)
            sp.frame(depth[f].data_ptr(), stream.pose(f), stream.k, mu, f)
        if streaming:   # raycast(f) + scan(f+1) were one launch, the ncclAllGather followed it on the same stream
            n = sp.p.launch_counts()
            assert n["fused"] == frames - 4 and n["pending"] == 1, n
        torch.cuda.synchronize()
        # the collective really moved the list (committing one's own list again would pass without it)
        w = sp._words
        assert torch.equal(sp.recv[:w], sp.send[:w]) and int(sp.send[:w].ne(0).sum()) > 10
        c, x, y, a = single.blocks()
        rc, rx, ry, ra = sp.p.blocks()
        assert len(c) > 500 and rc.shape == c.shape and (rc == c).all() and (ra == a).all()
        assert (rx.view(np.uint32) == x.view(np.uint32)).all() and (ry.view(np.uint32) == y.view(np.uint32)).all()
        v, n = single.vertex_normal()
        rv, rn = sp.p.vertex_normal()
        assert (rv.view(np.uint32) == v.view(np.uint32)).all() and (rn.view(np.uint32) == n.view(np.uint32)).all()
        sp.close(); single.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_create_replicas_partitions_the_image_rows():
    """se_hip_create_replicas (the device_ids[] form of the constructor): one row-sharded replica per listed
    device, shares aligned to the 8-row raycast tiles -- here the same device three times.  Together the replicas
    raycast the whole image: their tiles stitched = the unsharded pipeline's."""
    import ctypes as C
    from supereight_amd.pipeline import _Config, load_library
    lib = load_library()
    W, H, N, dim, mu = 160, 116, 256, 2.4, 0.1
    cfg = _Config(W, H, N, dim, SDF, 0, 0, 0, 0)
    ids = (C.c_int32 * 3)(0, 0, 0)
    handles = (C.c_void_p * 3)()
    assert lib.se_hip_create_replicas(C.byref(cfg), ids, 3, handles) == 0
    parts = row_partition(H, 3)
    stream = SyntheticStream(W, H, dim)
    reps = []
    for r in range(3):
        p = DenseSLAMPipeline.__new__(DenseSLAMPipeline)          # adopt the handle the C call made
        p.lib, p.W, p.H, p.size, p.dim, p.field, p._h, p._keepalive = lib, W, H, N, dim, SDF, C.c_void_p(handles[r]), None
        p.pose_ = np.eye(4, dtype=np.float32)
        reps.append(p)
    for f in range(4):
        depth, pose = stream.depth(f), stream.pose(f)
        for p in reps:
            p.set_depth(depth); p.setPose(pose)
            p.integration(stream.k, 1, mu, f)      # (each replica only allocates what its own rows see: the exchange is the caller's)
            p.raycasting(stream.k, mu, f)
    assert parts[0][0] == 0 and parts[-1][1] == H and all(parts[i][1] == parts[i + 1][0] for i in range(2))
    for r, p in enumerate(reps):
        b, e = parts[r]
        assert (b % 8 == 0) and (e % 8 == 0 or e == H)
        v, n = p.vertex_normal()
        touched = (n != 0).any(axis=-1)            # raycastKernel writes a unit normal or (INVALID, 0, 0) to every pixel it owns
        assert touched[b:e].all() and not touched[:b].any() and not touched[e:].any()
    # more devices than tiles is refused, and nothing leaks
    ids20 = (C.c_int32 * 20)(*([0] * 20))
    h20 = (C.c_void_p * 20)()
    assert lib.se_hip_create_replicas(C.byref(cfg), ids20, 20, h20) < 0 and all(not h for h in h20)
    for p in reps:
        p.close()


def test_unsharded_handle_with_caller_key_list_orders_commit_behind_scan():
    """a handle that covers the whole image but writes its key list into a caller buffer (a one-rank
    exchange, or a C caller of se_hip_new_keys_device + se_hip_alloc_commit) and synchronises every frame found its main
    stream idle, so the scan went onto the main stream while exchange / commit stayed on the scan stream -- unordered.
    The scan of such a handle now always runs on the scan stream and commit / exchange follow the stream the scan used.
    Result must be the plain pipeline's, bit for bit, and the consumer of the list must see a complete list."""
    import torch
    W, H, N, dim, mu, frames = 160, 120, 256, 2.4, 0.1, 7
    stream = SyntheticStream(W, H, dim)
    single = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
    p = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
    assert p.scan_overlaps()
    words = 1 << 15
    send = torch.zeros(words, dtype=torch.int64, device="cuda")
    p.set_new_keys_buffer(send.data_ptr(), words, keepalive=send)
    total_keys = 0
    for f in range(frames):
        depth, pose = stream.depth(f), stream.pose(f)
        for q in (single, p):
            q.set_depth(depth); q.setPose(pose)
        before = single.counts()[0]
        single.integration(stream.k, 1, mu, f); single.raycasting(stream.k, mu, f); single.sync()
        p.sync()                                      # main stream idle: the situation of the finding
        assert p.alloc_scan(stream.k, 1, mu, f)
        p.alloc_commit(send.data_ptr(), 1, words)     # no sync in between: ordered by the library
        p.integrate_sweep(stream.k, 1, mu, f)
        p.raycasting(stream.k, mu, f)
        p.sync()
        n_keys = int(send[0].item())
        new_blocks = single.counts()[0] - before
        keys = send[1:1 + n_keys].cpu().numpy().view(np.uint64)
        assert int(((keys >> np.uint64(63)) == 0).sum()) >= new_blocks   # the list holds (at least) every new block of the frame
        total_keys += n_keys
    assert total_keys > 500
    c, x, y, a = single.blocks()
    rc, rx, ry, ra = p.blocks()
    assert rc.shape == c.shape and (rc == c).all() and (ra == a).all()
    assert (rx.view(np.uint32) == x.view(np.uint32)).all() and (ry.view(np.uint32) == y.view(np.uint32)).all()
    v, n = single.vertex_normal(); rv, rn = p.vertex_normal()
    assert (rv.view(np.uint32) == v.view(np.uint32)).all() and (rn.view(np.uint32) == n.view(np.uint32)).all()
    single.close(); p.close()


def test_capacity_error_is_sticky_until_cleared():
    """SE_HIP_E_CAPACITY stays raised on every later stage call and on sync (se_hip.h conventions) until the caller
    acknowledges it with se_hip_clear_overflow, which names what overflowed."""
    from supereight_amd.pipeline import SeHipError
    W, H, N, dim, mu = 160, 120, 256, 2.4, 0.1
    stream = SyntheticStream(W, H, dim)
    p = DenseSLAMPipeline((W, H), N, dim, field_type=SDF, max_blocks=256)     # frame 0 needs ~2000 blocks
    with pytest.raises(SeHipError, match="pool exhausted"):
        for f in range(3):
            p.set_depth(stream.depth(f)); p.setPose(stream.pose(f))
            p.integration(stream.k, 1, mu, f)
            p.sync()
    with pytest.raises(SeHipError, match="pool exhausted"):
        p.sync()                                     # sticky
    assert p.clear_overflow() == 1
    p.sync()                                         # acknowledged
    assert p.clear_overflow() == 0
    assert p.counts()[0] == 256                      # the pool is full, its counter stayed bounded
    p.close()


@pytest.mark.parametrize("R,H", [(2, 120), (4, 116)], ids=["R2", "R4-odd-height"])
def test_sharded_tracking_sees_the_full_images(R, H):
    """SURVEY 8e-5 / VERDICT r02 missing #4: a row-sharded replica raycasts its own rows only, but tracking() reads the whole
    vertex_ / normal_ images.  The replicas' row tiles are packed, "all-gathered" (concatenated, as ncclAllGather delivers
    them) and applied; afterwards every replica holds the single pipeline's images and its ICP gives the single pipeline's
    pose, TrackData and reduction sums, bit for bit."""
    import torch
    W, N, dim, mu, frames = 160, 256, 2.4, 0.1, 6
    stream = SyntheticStream(W, H, dim)
    single = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
    parts = row_partition(H, R)
    max_rows = max(e - b for b, e in parts)
    reps = [DenseSLAMPipeline((W, H), N, dim, field_type=SDF, rows=parts[r]) for r in range(R)]
    words = 1 << 15
    send = [torch.zeros(words, dtype=torch.int64, device="cuda") for _ in range(R)]
    tile_bytes = reps[0].image_tile_bytes(max_rows)
    assert tile_bytes == 2 * max_rows * W * 12
    tiles = [torch.zeros(tile_bytes, dtype=torch.uint8, device="cuda") for _ in range(R)]
    for r in range(R):
        reps[r].set_new_keys_buffer(send[r].data_ptr(), words, keepalive=send[r])
    for f in range(frames):
        depth, pose = stream.depth(f), stream.pose(f)
        for p in [single] + reps:
            p.set_depth(depth); p.setPose(pose)
        if f == frames - 1:
            # the last frame is tracked from a slightly wrong pose against the raycast of the frame before
            start = stream.pose(f - 1)
            single.setPose(start)
            ok_s = single.tracking(stream.k, 1e-5, 1, f)
            td_s, red_s, it_s = single.track_data()
            # a row-sharded handle refuses to track against images of which it only holds its own rows
            from supereight_amd.pipeline import SeHipError
            with pytest.raises(SeHipError, match="row-sharded"):
                reps[0].tracking(stream.k, 1e-5, 1, f)
            for r, p in enumerate(reps):
                p.pack_image_tile(tiles[r].data_ptr(), max_rows)
            for p in reps:
                p.sync()
            recv = torch.cat(tiles)
            torch.cuda.synchronize()
            for p in reps:
                p.apply_image_tiles(recv.data_ptr(), parts, max_rows)
                v, n = p.vertex_normal()
                vs, ns = single.vertex_normal()
                assert (v.view(np.uint32) == vs.view(np.uint32)).all() and (n.view(np.uint32) == ns.view(np.uint32)).all()
                p.setPose(start)
                assert p.tracking(stream.k, 1e-5, 1, f) == ok_s
                td, red, it = p.track_data()
                assert it == it_s and (red.view(np.uint32) == red_s.view(np.uint32)).all()
                assert (td["result"] == td_s["result"]).all()
                assert (p.getPose().view(np.uint32) == single.getPose().view(np.uint32)).all()
            assert ok_s
            break
        single.integration(stream.k, 1, mu, f); single.raycasting(stream.k, mu, f)
        for p in reps:
            assert p.alloc_scan(stream.k, 1, mu, f)
        for p in reps:
            p.sync()
        recv = torch.cat(send)
        torch.cuda.synchronize()
        for p in reps:
            p.alloc_commit(recv.data_ptr(), R, words)
            p.integrate_sweep(stream.k, 1, mu, f)
            p.raycasting(stream.k, mu, f)
        for p in reps:
            p.sync()
    for p in [single] + reps:
        p.close()


def _two_rank_worker(rank, world, port, streaming, exchange, q):
    """One of two processes sharing GPU 0 (gloo transport, as bench.py's dry run of --gpus 2): streams the frames through ShardedPipeline without
    ever synchronising, then compares its replica with an unsharded pipeline fed the same frames."""
    import os
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from supereight_amd.multi_gpu import ShardedPipeline
        W, H, N, dim, mu, frames = 320, 240, 512, 4.8, 0.1, 30
        s = SyntheticStream(W, H, dim)
        depth = torch.from_numpy(np.stack([s.depth(f) for f in range(frames)])).cuda()
        sp = ShardedPipeline((W, H), N, dim, SDF, rank, world, 0, streaming=streaming)
        assert sp.gloo and sp.streaming == streaming
        for f in range(frames):
            sp.frame(depth[f].data_ptr(), s.pose(f), s.k, mu, f)        # no synchronisation between frames
        fused = sp.p.launch_counts()["fused"]
        torch.cuda.synchronize()
        single = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
        for f in range(frames):
            single.set_depth_device(depth[f].data_ptr()); single.setPose(s.pose(f))
            single.integration(s.k, 1, mu, f)
            single.raycasting(s.k, mu, f)
        c, x, y, a = single.blocks()
        rc, rx, ry, ra = sp.p.blocks()
        b, e = sp.rows
        v, n = single.vertex_normal()
        rv, rn = sp.p.vertex_normal()
        res = {"rank": rank, "fused": fused, "blocks": [int(len(c)), int(len(rc))],
               "same_set": bool(rc.shape == c.shape and (rc == c).all())}
        if res["same_set"]:
            res["x_mismatch"] = int((rx.view(np.uint32) != x.view(np.uint32)).sum())
            res["y_mismatch"] = int((ry.view(np.uint32) != y.view(np.uint32)).sum())
            res["active_mismatch"] = int((ra != a).sum())
        res["image_mismatch"] = int((rv[b:e].view(np.uint32) != v[b:e].view(np.uint32)).sum() + (rn[b:e].view(np.uint32) != n[b:e].view(np.uint32)).sum())
        sp.close(); single.close()
        q.put(res)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as ex:     # (a worker that dies silently would leave the parent waiting)
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc()})
        raise ex


@pytest.mark.parametrize("streaming", [False, True], ids=["two-queue", "one-queue"])
def test_two_ranks_on_one_gpu_stream_frames_without_synchronising(streaming):
    """The N = 2 frame loop exactly as bench.py --gpus 2 drives it (two processes, every frame enqueued behind the last, the key lists exchanged
    through torch.distributed) -- on one GPU with the gloo transport, which is what this box offers.  The in-process replica tests above separate
    the stages with synchronisations; here nothing does, so the stream ordering between the library's launches and torch's copies / collectives is
    what is tested.  Both replicas must end with the unsharded pipeline's map and their rows of its last raycast, bit for bit."""
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, streaming, "gloo", q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for res in sorted(out, key=lambda r: r["rank"]):
        print(res)
        assert "error" not in res, res["error"]
        assert res["same_set"] and res["blocks"][0] > 5000, res
        assert res["x_mismatch"] == 0 and res["y_mismatch"] == 0 and res["active_mismatch"] == 0 and res["image_mismatch"] == 0, res
        assert res["fused"] == (30 - 4 if streaming else 0), res
