"""Multi-GPU driver of the dense-fusion path: one process per GPU, image rows sharded across ranks,
one RCCL all-gather of the per-rank new-block key lists per frame (BASELINE.json north_star,
SURVEY.md section 8e).

Per frame and rank r of R:
  1. allocation scan over the rank's image rows -> blocks allocated locally + a key list
     ``[count, key_1, ...]`` written straight into the all-gather send buffer;
  2. ``all_gather_into_tensor`` of the fixed-capacity lists (count in word 0, so no separate size
     exchange; over xGMI every peer is one hop away and the message is latency-bound);
  3. every rank inserts the other ranks' keys -> all replicas hold the same block set;
  4. integration sweep (replicated: every replica must hold every voxel update because later
     frames raycast the map from other view points);
  5. raycast of the rank's image rows.

The map is replicated, the image-space work (1, 5) is sharded.  ``torch.distributed`` is only the
transport (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests of the exchange logic).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

# frames <= 3 always integrate (DenseSLAMSystem.cpp:209) and build the map from nothing: their
# key lists are large.  Later frames only add what newly came into view.
BIG_FRAMES = 3


def row_partition(height: int, world: int, align: int = 8) -> List[Tuple[int, int]]:
    """Contiguous row tiles, boundaries aligned to the 8-row raycast tile, covering [0, height)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    units = (height + align - 1) // align
    bounds = [min(height, align * ((units * r) // world)) for r in range(world)] + [height]
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def exchange_key_lists(local, world: int, group=None):
    """All-gather fixed-capacity key lists.  ``local`` is a 1-D int64 tensor ``[count, keys...]``;
    returns a ``world * len(local)`` tensor holding every rank's list in rank order."""
    import torch
    import torch.distributed as dist
    out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    if world == 1:
        out.copy_(local)
    else:
        dist.all_gather_into_tensor(out, local, group=group)
    return out


def pack_row_tile(image, rows: Tuple[int, int], max_rows: int):
    """The rows [rows[0], rows[1]) of an (H, W, C) image tensor as a (max_rows, W, C) tile, zero-padded: the fixed-size
    message of the vertex / normal all-gather."""
    import torch
    tile = torch.zeros((max_rows,) + tuple(image.shape[1:]), dtype=image.dtype, device=image.device)
    tile[: rows[1] - rows[0]] = image[rows[0]:rows[1]]
    return tile


def stitch_row_tiles(gathered, parts: List[Tuple[int, int]], out):
    """Writes the tiles of an all-gather of pack_row_tile() messages (``gathered``: (world, max_rows, W, C)) into the
    (H, W, C) image ``out`` according to the row partition; returns ``out``."""
    for r, (b, e) in enumerate(parts):
        out[b:e] = gathered[r, : e - b]
    return out


def gather_row_tiles(image, rows: Tuple[int, int], parts: List[Tuple[int, int]], group=None):
    """Full (H, W, C) image on every rank from the ranks' row tiles, through torch.distributed (gloo on CPU tensors in the
    tests; the RCCL path of ShardedPipeline.gather_images issues the collective from C instead)."""
    import torch
    import torch.distributed as dist
    world = len(parts)
    max_rows = max(e - b for b, e in parts)
    tile = pack_row_tile(image, rows, max_rows)
    recv = torch.empty(world * tile.numel(), dtype=tile.dtype, device=tile.device)      # flat, like the key lists
    if world == 1:
        recv.copy_(tile.reshape(-1))
    else:
        dist.all_gather_into_tensor(recv, tile.reshape(-1).contiguous(), group=group)
    return stitch_row_tiles(recv.reshape((world,) + tuple(tile.shape)), parts, image.clone())


def merged_keys(gathered, world: int) -> np.ndarray:
    """Host-side view of an exchanged buffer: the concatenation of all valid keys (uint64)."""
    g = gathered.detach().cpu().numpy().view(np.uint64).reshape(world, -1)
    parts = []
    for r in range(world):
        n = int(g[r, 0])
        if n > g.shape[1] - 1:
            raise OverflowError(f"rank {r} produced {n} keys, exchange capacity is {g.shape[1] - 1}")
        parts.append(g[r, 1:1 + n])
    return np.concatenate(parts) if parts else np.zeros(0, np.uint64)


class ShardedPipeline:
    """Row-sharded DenseSLAMPipeline replica of one rank.

    Stream plan (R > 1): the allocation scan of frame f+1 and the all-gather of its key lists are
    ordered on an exchange stream that only waits for the sweep of frame f, so both hide behind the
    raycast of frame f, which runs on the main stream; the main stream joins the exchange stream
    before it inserts the other ranks' keys.  The host never synchronises inside ``frame``.

    Ordering contract for the depth image: ``frame`` enqueues on the streams chosen at construction -- torch's current stream then, or, if
    that was the legacy default stream (handle 0, on which the library cannot launch), a private stream joined to it ONCE, at construction.  A caller that
    produces each frame's depth image on the device must therefore either produce it on ``self.main`` / ``self.xs`` (``with torch.cuda.stream(sp.xs or
    sp.main)``) or make those streams wait for its producer before calling ``frame`` (``sp.main.wait_stream(producer)``, and ``sp.xs`` likewise); images
    that are resident before the loop starts (bench.py, the tests) need nothing.
    """

    def __init__(self, input_size, volume_resolution, volume_dimension, field_type, rank, world, device,
                 small_words: int = 0, big_words: int = 0, max_blocks: int = 0, group=None,
                 exchange_always: bool = False, shard_sweep: bool = False, brick_cap: int = 0, streaming: bool = False):
        import torch
        from .pipeline import DenseSLAMPipeline
        self.torch = torch
        self.rank, self.world, self.group = rank, world, group
        self.parts = row_partition(int(input_size[1]), world)
        rows = self.parts[rank]
        self.rows = rows
        self.max_rows = max(e - b for b, e in self.parts)
        self._img_send = self._img_recv = None
        self._images_full = world == 1
        self.p = DenseSLAMPipeline(input_size, volume_resolution, volume_dimension, field_type=field_type,
                                   device=device, max_blocks=max_blocks, rows=rows)
        dev = torch.device("cuda", device)
        _, cap = self.p.new_keys_device()
        self.big_words = int(big_words) if big_words else int(min(cap, 1 << 22))
        # steady state: the surface newly in view per frame; grows with the square of the resolution
        if not small_words:
            small_words = 16384 * max(1, int(volume_resolution) // 512) ** 2
        self.small_words = int(min(small_words, self.big_words))
        self.send = torch.zeros(self.big_words, dtype=torch.int64, device=dev)
        self.recv = torch.zeros(world * self.big_words, dtype=torch.int64, device=dev)
        self._views = {w: (self.send[:w], self.recv[: world * w]) for w in {self.big_words, self.small_words}}
        self._words = 0
        self.gloo = False
        # exchange_always: run the exchange + commit steps with a single rank too (test hook: the own list
        # is committed a second time, which changes nothing)
        self.exchange = world > 1 or exchange_always
        self.main = torch.cuda.current_stream(dev)
        self.xs = None
        self._pg = None
        self.direct = False
        if self.exchange:
            import torch.distributed as dist
            self.gloo = dist.get_backend(group) == "gloo"
            # scan + collective are ordered on one stream, the sweep and the raycast get a stream of their own.  That stream must have a
            # handle the library can launch on: torch's default stream is the legacy null stream (handle 0, which se_hip_set_scan_stream reads as
            # "make your own") -- a scan on the library's own stream and a torch copy / collective on the default stream are not ordered at all
            # (r05: tests/test_gpu_sharded.py::test_two_ranks_on_one_gpu_stream_frames_without_synchronising lost 34 of 10 534 blocks that way;
            # the direct RCCL path issues its all-gather from C on the scan's stream and was not affected)
            self.xs = torch.cuda.current_stream(dev)
            if self.xs.cuda_stream == 0:
                cur = self.xs
                self.xs = torch.cuda.Stream(dev)
                self.xs.wait_stream(cur)
                self.main = self.xs
            if self.p.scan_overlaps():
                if not streaming:
                    self.main = torch.cuda.Stream(dev)
                # streaming (the one-queue schedule with an exchange): raycast(f) + scan(f+1) are one launch on the main stream, the all-gather, the
                # commit and the sweep follow on it -- scan stream = main stream, so that the scans that do not ride in a raycast's launch (frames
                # 0..3, a frame behind any call that launched the held-back raycast) are ordered with torch's copies / collectives as well
                self.p.set_scan_stream(self.xs.cuda_stream)
            # else (pooled bricks / overlap switched off): the scan runs on the main stream, so everything,
            # the collective included, is ordered on the one current stream
            pg = group if group is not None else dist.group.WORLD
            self._pg = pg if (not self.gloo and hasattr(pg, "_allgather_base")) else None
            self.direct = (not self.gloo) and self._direct_rccl(pg, dev)
        self.p.set_stream(self.main.cuda_stream)
        self.streaming = False
        if streaming:
            # the one-queue schedule (include/se_hip.h, se_hip_set_streaming): a frame's raycast is launched with the next frame's scan
            self.streaming = bool(self.p.set_streaming(True))
        # Sharded sweep (SURVEY 8e option 4; off by default, DESIGN.md section 7 has the price): owner-computes integration
        # + an all-gather of the updated bricks instead of every replica sweeping every block.
        self.shard_sweep = bool(shard_sweep) and world > 1
        if self.shard_sweep:
            if not brick_cap:   # blocks in view per frame ~ surface area in blocks; x2 for ownership imbalance
                brick_cap = int(2 * 20000 * max(1, int(volume_resolution) // 512) ** 2 / world)
            self.brick_cap = 64 * ((int(brick_cap) + 63) // 64)   # 64 sub-segments with a record counter each
            seg = self.p.sweep_shard_bytes(self.brick_cap)
            self.bsend = torch.zeros(seg, dtype=torch.uint8, device=dev)
            self.brecv = torch.zeros(world * seg, dtype=torch.uint8, device=dev)
            self.p.set_sweep_shard(rank, world, self.bsend.data_ptr(), self.brick_cap, keepalive=self.bsend)

    def _exchange_bricks(self):
        """All-gather of the send segments on the MAIN stream (behind the sweep that packed them), then apply."""
        p, torch = self.p, self.torch
        if self.direct:
            p.brick_exchange(self.brecv.data_ptr())
            return
        with torch.cuda.stream(self.main):
            if self.gloo:
                import torch.distributed as dist
                host = self.bsend.cpu()
                out = torch.empty(self.world * host.numel(), dtype=host.dtype)
                dist.all_gather_into_tensor(out, host, group=self.group)
                self.brecv.copy_(out)
            elif self._pg is not None:
                self._pg._allgather_base(self.brecv, self.bsend).wait()
            else:
                import torch.distributed as dist
                dist.all_gather_into_tensor(self.brecv, self.bsend, group=self.group)
        p.apply_bricks(self.brecv.data_ptr(), self.world)

    def gather_images(self):
        """after this call every rank holds the FULL vertex_ / normal_ images of the last raycast (each rank
        raycasts its own rows only).  One all-gather of fixed-size row tiles on the main stream; a no-op for one rank and
        when the images are already complete.  Needed by tracking() and the render methods, not by the integrate + raycast
        loop itself, so it is only issued on demand."""
        if self._images_full:
            return
        p, torch = self.p, self.torch
        nbytes = p.image_tile_bytes(self.max_rows)
        if self._img_send is None:
            dev = self.send.device
            self._img_send = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            self._img_recv = torch.zeros(self.world * nbytes, dtype=torch.uint8, device=dev)
        if self.direct:
            p.gather_images(self._img_send.data_ptr(), self._img_recv.data_ptr(), self.parts, self.max_rows)
        else:
            p.pack_image_tile(self._img_send.data_ptr(), self.max_rows)
            with torch.cuda.stream(self.main):
                if self.gloo:
                    import torch.distributed as dist
                    host = self._img_send.cpu()
                    out = torch.empty(self.world * host.numel(), dtype=host.dtype)
                    dist.all_gather_into_tensor(out, host, group=self.group)
                    self._img_recv.copy_(out)
                elif self._pg is not None:
                    self._pg._allgather_base(self._img_recv, self._img_send).wait()
                else:
                    import torch.distributed as dist
                    dist.all_gather_into_tensor(self._img_recv, self._img_send, group=self.group)
            p.apply_image_tiles(self._img_recv.data_ptr(), self.parts, self.max_rows)
        self._images_full = True

    def tracking(self, k, icp_threshold: float, tracking_rate: int, frame: int, pyramid=(10, 5, 4)) -> bool:
        """DenseSLAMSystem::tracking in the sharded mode: the ICP reads the whole vertex_ / normal_ images of the last raycast,
        so the row tiles are all-gathered first; every rank then runs the same (replicated, deterministic) tracker on the
        full images and ends with the same pose, bit for bit."""
        self.gather_images()
        return self.p.tracking(k, icp_threshold, tracking_rate, frame, pyramid)

    def _direct_rccl(self, pg, dev) -> bool:
        """Hands the process group's ncclComm_t and the address of ncclAllGather (of the RCCL torch has loaded)
        to the C library, so that the per-frame exchange is one C call instead of a torch collective
        (host time per frame 78 -> ~45 us).  Falls back to torch.distributed when torch does not expose them."""
        import ctypes
        import os
        if os.environ.get("SE_EXCHANGE", "") == "torch":
            return False
        try:
            comm = pg._get_backend(dev)._comm_ptr()
            lib = None
            for name in ("librccl.so", "librccl.so.1"):
                path = os.path.join(os.path.dirname(self.torch.__file__), "lib", name)
                if os.path.exists(path):
                    lib = ctypes.CDLL(path)
                    break
            if lib is None:
                lib = ctypes.CDLL("librccl.so.1")
            fn = ctypes.cast(lib.ncclAllGather, ctypes.c_void_p).value
            if not comm or not fn:
                return False
            self.p.set_exchange(comm, fn, self.world)
            return True
        except Exception:
            return False

    def frame(self, depth_ptr: int, pose, k, mu: float, frame: int, integration_rate: int = 1):
        p = self.p
        if not self.exchange and getattr(pose, "shape", None) == (16,):
            # single replica, pose already in the C-ABI layout (to_colmajor): the whole frame is one FFI call
            # (no exchange: the scan writes the handle's own key lists, whose count words the sweep kernel clears)
            return p.frame(depth_ptr, pose, k, mu, frame, integration_rate) & 1
        if getattr(pose, "shape", None) == (16,):
            pose = np.asarray(pose, np.float32).reshape(4, 4).T
        p.set_depth_device(depth_ptr)
        p.setPose(pose)
        words = self.big_words if frame <= BIG_FRAMES else self.small_words
        if self.exchange and words != self._words:
            p.set_new_keys_buffer(self.send.data_ptr(), words, keepalive=self.send)
            self._words = words
        ran = p.alloc_scan(k, integration_rate, mu, frame)
        if ran:
            if self.exchange:
                send, recv = self._views[words]
                if self.gloo:
                    # dry-run transport (several ranks sharing one GPU, no RCCL): stage through the host
                    with self.torch.cuda.stream(self.xs):
                        host = exchange_key_lists(send.cpu(), self.world, self.group)   # .cpu(): ordered on xs, blocks the host
                        recv.copy_(host)
                elif self.direct:
                    p.alloc_exchange(recv.data_ptr(), words)
                    p.integrate_sweep(k, integration_rate, mu, frame)
                    if self.shard_sweep:
                        self._exchange_bricks()
                    self._raycast(k, mu, frame)
                    return ran
                else:
                    with self.torch.cuda.stream(self.xs):
                        if self._pg is not None:
                            self._pg._allgather_base(recv, send).wait()   # wait() = the issuing stream waits, not the host
                        else:
                            import torch.distributed as dist
                            dist.all_gather_into_tensor(recv, send, group=self.group)
                    # no stream join here: se_hip_alloc_commit fences the scan stream (= xs) itself
                p.alloc_commit(recv.data_ptr(), self.world, words)
            p.integrate_sweep(k, integration_rate, mu, frame)
            if self.shard_sweep:
                self._exchange_bricks()
        self._raycast(k, mu, frame)
        return ran

    def _raycast(self, k, mu, frame):
        if self.streaming:
            self.p.raycasting_deferred(k, mu, frame)    # launched with the next frame's scan, or by whatever call comes first
        else:
            self.p.raycasting(k, mu, frame)
        self._images_full = self.world == 1

    def close(self):
        self.p.close()
