"""Parity outside the comfort zone: the "ICL-like stress" stream of supereight_amd/synthetic.py --
a room larger than the volume on two sides (surfaces clipped by the cube, depth points and allocation band steps outside
it, rays ended by the cube exit or the far plane), a pillar and a sphere cut by the x = 0 face (depth discontinuities),
~1.3 cm and 2 deg per frame with a +-90 deg pan there and back (thousands of new keys per frame at 1024^3, blocks leave the
frustum -> active(false) and come back), sigma = 1 mm sensor noise, 2 % holes, depths beyond farPlane, ICL-NUIM intrinsics
(negative fy) -- through the CPU oracle and through the C ABI, compared bit for bit.

Out-of-volume semantics (documented in oracle/se_oracle.cpp, Octree::count_oob, and DESIGN.md 2): the allocation scan skips
steps outside the volume (kfusion/alloc_impl.hpp:92-96, guarded in the reference); the read paths of the raycast
(get_fine / fetch / the cached get, octree.hpp:357-408,439-458) are NOT guarded in the reference: their tree walk wraps
modulo size, which returns "not allocated" unless a block exists at the wrapped position, in which case the reference
indexes that block's array out of bounds (undefined).  Oracle and HIP path return "not allocated" for every such read.
`oob` counts them, `oob_ub` the undefined kind: the tests assert oob > 0 (the regime is really exercised) and
oob_ub == 0 (the reference itself is defined on every read of the stream), i.e. there is no `oob == 0` crutch any more.
"""
import json

import numpy as np
import pytest

from oracle.binding import OFUSION, SDF
from tests.parity_util import compare_maps, compare_raycast, run_both

pytestmark = pytest.mark.gpu

CASES = [
    # name, field, W, H, N, mu, frames, max_blocks
    ("stress_sdf_640x480_512", SDF, 640, 480, 512, 0.1, 36, 0),            # BASELINE configs[0] geometry
    ("stress_sdf_640x480_1024", SDF, 640, 480, 1024, 0.1, 32, 0),          # configs[2]: allocation churn (1-3 k new keys per frame)
    ("stress_ofusion_640x480_512", OFUSION, 640, 480, 512, 0.008, 32, 0),
    ("stress_ofusion_640x480_1024", OFUSION, 640, 480, 1024, 0.008, 30, 0),
    # the whole pan there and back (0 -> +90 -> -90 deg -> ...): almost every block leaves the frustum and is found again
    ("stress_sdf_320x240_512_long", SDF, 320, 240, 512, 0.1, 150, 0),
    ("stress_sdf_320x240_512_long_pooled", SDF, 320, 240, 512, 0.1, 150, 1 << 15),
    ("stress_ofusion_320x240_512_long", OFUSION, 320, 240, 512, 0.02, 110, 0),
]


@pytest.mark.parametrize("name,field,W,H,N,mu,frames,max_blocks", CASES, ids=[c[0] for c in CASES])
def test_stress_stream_parity(name, field, W, H, N, mu, frames, max_blocks):
    dim = 4.8
    seen = {"min_active_frac": 1.0, "max_new": 0, "prev_blocks": 0, "raycasts": 0}

    def on_frame(f, cpu, gpu, rec):
        # every frame: raycast images bit for bit (cheap); block bookkeeping for the regime assertions below
        if rec["raycast"]:
            r = compare_raycast(rec, dim / N)
            assert r["hitmask_mismatch"] == 0 and r["vertex_bit_mismatch_px"] == 0 and r["normal_bit_mismatch_px"] == 0, (f, r)
            seen["raycasts"] += 1
        nb, _ = cpu.counts()
        seen["max_new"] = max(seen["max_new"], nb - seen["prev_blocks"]) if f > 3 else seen["max_new"]
        seen["prev_blocks"] = nb
        if f % 10 == 9 or f == frames - 1:
            m = compare_maps(cpu, gpu)
            assert m["same_block_set"] and m["same_node_set"], (f, m)
            assert m["x_mismatch"] == 0 and m["y_mismatch"] == 0 and m["active_mismatch"] == 0, (f, m)
            assert m["node_x_mismatch"] == 0 and m["node_y_mismatch"] == 0, (f, m)
            act = cpu.blocks()[3]
            seen["min_active_frac"] = min(seen["min_active_frac"], float(act.mean()))

    cpu, gpu, recs = run_both(field, W, H, N, dim, mu, frames, max_blocks=max_blocks, on_frame=on_frame, stream_kind="stress")
    st = cpu.stats()
    print(name, json.dumps({"blocks": cpu.counts()[0], "oob": st["oob"], "oob_ub": st["oob_ub"], "truncated": st["truncated"],
                            "max_new_blocks_per_frame": seen["max_new"], "min_active_frac": round(seen["min_active_frac"], 3)}))
    assert seen["raycasts"] == frames - 3
    assert st["truncated"] == 0          # the reference's key buffer never saturated -> its block set is deterministic
    assert st["oob"] > 0                 # samples did leave the volume: the regime is exercised ...
    assert st["oob_ub"] == 0             # ... and the reference itself is defined on every one of them
    assert seen["max_new"] > (50 if N == 512 else 300)            # centimetre-scale motion: real allocation churn after the start-up frames
    if frames >= 100:
        assert seen["min_active_frac"] < 0.5                       # most of the map out of view at some point (deactivated), then found again
    cpu.close()
    gpu.close()


PIPELINED = [
    # id, field, N, mu, frames, max_blocks
    ("sdf-512", SDF, 512, 0.1, 48, 0),                 # every occupancy level staged in LDS: the SHALLOW / O32 instantiations
    ("ofusion-512", OFUSION, 512, 0.02, 40, 0),
    ("sdf-512-pooled", SDF, 512, 0.1, 48, 1 << 15),
    ("ofusion-512-pooled", OFUSION, 512, 0.02, 40, 1 << 15),
    ("sdf-1024", SDF, 1024, 0.1, 30, 0),               # levels beyond the staged ones: the generic (has_deep) instantiations
    ("sdf-1024-pooled", SDF, 1024, 0.1, 30, 1 << 17),
    ("ofusion-1024", OFUSION, 1024, 0.02, 24, 0),
]


@pytest.mark.parametrize("streaming", [True, False], ids=["one-queue", "two-queue"])
@pytest.mark.parametrize("name,field,N,mu,frames,max_blocks", PIPELINED, ids=[c[0] for c in PIPELINED])
def test_stress_stream_pipelined(name, field, N, mu, frames, max_blocks, streaming):
    _pipelined_case("stress", 320, 240, field, N, mu, frames, max_blocks, streaming)


# The same, at the SHAPE the benchmark times: 640x480 is 2 400 raycast workgroups on a chip that holds 2 560, with the next frame's
# 2 400 scan workgroups queued behind them -- at 320x240 both halves of k_raycast_scan are co-resident from the first microsecond, a different interleaving of
# "the scan inserts into tab[] / lbits / cbits / fbits while the raycast's beam start and march read them".  bench.py's own workload and stream first.
BENCH_SHAPE = [
    # id, stream, field, N, mu, frames, max_blocks
    ("room-sdf-512", "room", SDF, 512, 0.1, 28, 0),                 # = bench.py's headline workload (SHALLOW / O32 instantiation)
    ("room-ofusion-512", "room", OFUSION, 512, 0.008, 24, 0),       # BASELINE configs[4]: the OFusion leap inside the fused launch
    ("room-sdf-1024", "room", SDF, 1024, 0.1, 16, 0),               # the has_deep instantiation at full width
    ("stress-sdf-512", "stress", SDF, 512, 0.1, 24, 0),             # 100-300 insertions per frame beside the raycast
    ("room-sdf-512-pooled", "room", SDF, 512, 0.1, 16, 1 << 16),    # pooled bricks: the raycast reads the index the scan half writes
]


@pytest.mark.parametrize("name,stream,field,N,mu,frames,max_blocks", BENCH_SHAPE, ids=[c[0] for c in BENCH_SHAPE])
def test_fused_launch_at_bench_shape(name, stream, field, N, mu, frames, max_blocks):
    _pipelined_case(stream, 640, 480, field, N, mu, frames, max_blocks, True, min_hits_per_frame=100000)


def _pipelined_case(stream_kind, W, H, field, N, mu, frames, max_blocks, streaming, min_hits_per_frame=300):
    """The stress stream enqueued back to back with NO call between the se_hip_frame calls, every frame's vertex / normal images kept in an
    image ring (se_hip_set_image_ring) and EVERY slot compared with the oracle's raycast of that frame, bit for bit, plus the final map.
      one-queue (se_hip_set_streaming): the raycast of frame f runs inside k_raycast_scan, the launch that also scans frame f+1 -- the kernel
        the benchmark's headline times.  Its scan half inserts into the index (and, pooled bricks, hands out bricks) that its raycast half is
        reading; asserted through the launch counters: all raycasts but the last (flushed by the final sync) were fused launches.
      two-queue (the eager schedule): k_raycast on the main stream beside k_alloc_scan on the scan stream, released by the host gate.
    Pooled maps: the serial schedule (SE_HIP_POOLED_OVERLAP=0) must give the same images."""
    import os
    import torch
    from oracle.binding import OraclePipeline
    from supereight_amd.pipeline import DenseSLAMPipeline
    from supereight_amd.synthetic import make_stream, to_colmajor
    dim = 4.8
    s = make_stream(stream_kind, W, H, dim)
    depths = [s.depth(f) for f in range(frames)]
    poses = [s.pose(f) for f in range(frames)]
    dev = torch.from_numpy(np.stack(depths)).cuda()
    k = np.ascontiguousarray(s.k, np.float32)
    pcm = [to_colmajor(q) for q in poses]

    def run(expect_overlap=True, stream_mode=streaming):
        gpu = DenseSLAMPipeline((W, H), N, dim, field_type=field, max_blocks=max_blocks)
        assert gpu.scan_overlaps() == expect_overlap
        ring = torch.zeros((frames, 2, H, W, 3), dtype=torch.float32, device="cuda")
        assert gpu.image_tile_bytes(H) == ring[0].numel() * 4
        gpu.set_image_ring(ring.data_ptr(), frames, keepalive=ring)
        if stream_mode:
            assert gpu.set_streaming(True) == expect_overlap and gpu.frame_is_fused() == expect_overlap
        gpu.launch_counts(reset=True)
        for f in range(frames):
            assert gpu.frame(dev[f].data_ptr(), pcm[f], k, mu, f) == (3 if f > 2 else 1)      # nothing else is called between the frames
        n = gpu.launch_counts()
        if stream_mode and expect_overlap:
            assert n["pending"] and n["raycast"] == frames - 4 and n["fused"] == frames - 4, n     # frames 3 .. frames-2: fused; the last one is held back
        else:
            assert not n["pending"] and n["raycast"] == frames - 3 and n["fused"] == 0, n
        assert n["integrate"] == frames and n["alloc_scan"] == frames, n
        gpu.sync()
        n = gpu.launch_counts()
        assert not n["pending"] and n["raycast"] == frames - 3, n
        return gpu, ring.cpu().numpy()

    gpu, ring = run()
    cpu = OraclePipeline(field, N, dim, W, H)
    hits = 0
    for f in range(frames):
        cpu.integrate(depths[f], poses[f], s.k, mu, f)
        ran, v_c, n_c = cpu.raycast(poses[f], s.k, mu, f)
        if ran:
            r = compare_raycast({"v_c": v_c, "n_c": n_c, "v_g": ring[f, 0], "n_g": ring[f, 1]}, dim / N)
            hits += r["hits_gpu"]
            assert r["hitmask_mismatch"] == 0 and r["vertex_bit_mismatch_px"] == 0 and r["normal_bit_mismatch_px"] == 0, (f, r)
        else:
            assert not ring[f].any()
    assert hits > min_hits_per_frame * (frames - 3), hits
    m = compare_maps(cpu, gpu)
    assert m["same_block_set"] and m["same_node_set"], m
    assert m["x_mismatch"] == 0 and m["y_mismatch"] == 0 and m["active_mismatch"] == 0, m
    assert m["node_x_mismatch"] == 0 and m["node_y_mismatch"] == 0, m
    v_g, n_g = gpu.vertex_normal()       # vertex_ / normal_ ARE the last slot
    assert (v_g.view(np.uint32) == ring[frames - 1, 0].view(np.uint32)).all() and (n_g.view(np.uint32) == ring[frames - 1, 1].view(np.uint32)).all()
    cpu.close(); gpu.close()
    if max_blocks and streaming:
        os.environ["SE_HIP_POOLED_OVERLAP"] = "0"
        try:
            gpu2, ring2 = run(expect_overlap=False)
        finally:
            del os.environ["SE_HIP_POOLED_OVERLAP"]
        assert (ring2.view(np.uint32) == ring.view(np.uint32)).all()
        gpu2.close()


def test_raw_image_pointers_end_deferral():
    """se_hip_vertex_normal_device on a streaming handle without an image ring switches deferral off for good: a consumer that reads vertex_ /
    normal_ through the raw pointers from work of its own, ordered on the handle's stream behind se_hip_frame(f), must find frame f's images --
    not those of frame f-1, which is what a held-back raycast would leave there.  The handle runs on a torch stream; after each
    frame the images are copied out on that stream through the raw pointers, without any se_hip call; every copy must be the oracle's."""
    import torch
    from oracle.binding import OraclePipeline
    from supereight_amd.pipeline import DenseSLAMPipeline
    from supereight_amd.synthetic import SyntheticStream, to_colmajor
    W, H, N, dim, mu, frames = 160, 120, 256, 2.4, 0.1, 9
    s = SyntheticStream(W, H, dim)
    depths = [s.depth(f) for f in range(frames)]
    poses = [s.pose(f) for f in range(frames)]
    dev = torch.from_numpy(np.stack(depths)).cuda()
    k = np.ascontiguousarray(s.k, np.float32)
    ts = torch.cuda.Stream()
    gpu = DenseSLAMPipeline((W, H), N, dim, field_type=SDF, streaming=True)
    gpu.set_stream(ts.cuda_stream)
    assert gpu.frame_is_fused()
    for f in range(4):
        gpu.frame(dev[f].data_ptr(), to_colmajor(poses[f]), k, mu, f)
    assert gpu.launch_counts()["pending"]                 # frame 3's raycast is held back ...
    pv, pn = gpu.vertex_normal_device()                   # ... launched by this call, which also ends deferral
    assert not gpu.frame_is_fused() and not gpu.launch_counts()["pending"]

    class Raw:   # the raw device pointer as a CUDA array for torch
        def __init__(self, ptr): self.__cuda_array_interface__ = {"shape": (H, W, 3), "typestr": "<f4", "data": (ptr, False), "version": 2}
    tv, tn = torch.as_tensor(Raw(pv), device="cuda"), torch.as_tensor(Raw(pn), device="cuda")
    assert tv.data_ptr() == pv and tn.data_ptr() == pn
    copies = {}
    for f in range(4, frames):
        gpu.frame(dev[f].data_ptr(), to_colmajor(poses[f]), k, mu, f)
        with torch.cuda.stream(ts):                        # the caller's own work on the handle's stream, behind se_hip_frame(f)
            copies[f] = (tv.clone(), tn.clone())
    ts.synchronize()
    assert gpu.launch_counts()["fused"] == 0
    cpu = OraclePipeline(SDF, N, dim, W, H)
    for f in range(frames):
        cpu.integrate(depths[f], poses[f], s.k, mu, f)
        _, v_c, n_c = cpu.raycast(poses[f], s.k, mu, f)
        if f in copies:
            r = compare_raycast({"v_c": v_c, "n_c": n_c, "v_g": copies[f][0].cpu().numpy(), "n_g": copies[f][1].cpu().numpy()}, dim / N)
            assert r["hits_gpu"] > 1000 and r["hitmask_mismatch"] == 0 and r["vertex_bit_mismatch_px"] == 0 and r["normal_bit_mismatch_px"] == 0, (f, r)
    cpu.close(); gpu.close()
