"""The raycast's scheduling state (previous-frame tile costs -> issue priorities, cost-sorted deal of the workgroups'
tile pairs over the compute units) must never reach the results: the same frames with the deal in image order
(SE_HIP_RAY_DEAL=1), without the cost feedback at all (SE_HIP_PRIO=0) and with the defaults give the same map and the
same vertex / normal images, bit for bit -- on an image whose tile count is odd (21 x 15 tiles: the last workgroup has one
tile, the pairs wrap around the row ends) -- and those equal the oracle's."""
import numpy as np
import pytest

from oracle.binding import SDF, OraclePipeline
from supereight_amd.pipeline import DenseSLAMPipeline
from supereight_amd.synthetic import SyntheticStream, to_colmajor
from tests.parity_util import compare_maps, compare_raycast

pytestmark = pytest.mark.gpu

W, H, N, DIM, MU, FRAMES = 168, 116, 256, 2.4, 0.1, 9


def _run(monkeypatch, env):
    import torch
    for k in ("SE_HIP_RAY_DEAL", "SE_HIP_PRIO"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s = SyntheticStream(W, H, DIM)
    depths = [s.depth(f) for f in range(FRAMES)]
    poses = [s.pose(f) for f in range(FRAMES)]
    dev = torch.from_numpy(np.stack(depths)).cuda()
    p = DenseSLAMPipeline((W, H), N, DIM, field_type=SDF, streaming=True)     # the knobs are read once, here; one-queue schedule: the fused launch deals its raycast workgroups the same way
    k = np.ascontiguousarray(s.k, np.float32)
    for f in range(FRAMES):
        p.frame(dev[f].data_ptr(), to_colmajor(poses[f]), k, MU, f)
    v, n = p.vertex_normal()
    return p, v, n, depths, poses, s.k


def test_schedule_never_changes_results(monkeypatch):
    ref, v0, n0, depths, poses, k = _run(monkeypatch, {})
    assert (n0[..., 0] != -2).sum() > 5000
    for env in ({"SE_HIP_RAY_DEAL": "1"}, {"SE_HIP_PRIO": "0"}, {"SE_HIP_RAY_DEAL": "7"}):
        p, v, n, _, _, _ = _run(monkeypatch, env)
        assert (v.view(np.uint32) == v0.view(np.uint32)).all() and (n.view(np.uint32) == n0.view(np.uint32)).all(), env
        c0, x0, y0, a0 = ref.blocks(); c, x, y, a = p.blocks()
        assert (c == c0).all() and (x.view(np.uint32) == x0.view(np.uint32)).all() and (y.view(np.uint32) == y0.view(np.uint32)).all() and (a == a0).all(), env
        p.close()
    cpu = OraclePipeline(SDF, N, DIM, W, H)
    for f in range(FRAMES):
        cpu.integrate(depths[f], poses[f], k, MU, f)
        _, v_c, n_c = cpu.raycast(poses[f], k, MU, f)
    m = compare_maps(cpu, ref)
    assert m["same_block_set"] and m["x_mismatch"] == 0 and m["y_mismatch"] == 0 and m["active_mismatch"] == 0, m
    r = compare_raycast({"v_c": v_c, "n_c": n_c, "v_g": v0, "n_g": n0}, DIM / N)
    assert r["hitmask_mismatch"] == 0 and r["vertex_bit_mismatch_px"] == 0 and r["normal_bit_mismatch_px"] == 0, r
    cpu.close(); ref.close()


@pytest.mark.parametrize("replicas", [1, 2], ids=["one-handle", "two-row-shards"])
def test_block_list_order_never_changes_results(monkeypatch, replicas):
    """Dense grids of >= 1024^3 keep their block list in address order (sort_block_list, se_hip_api.hip: the list is only a set there).  Forced on at 512^3
    (SE_HIP_SORT_BLOCKS=1: a sort in front of every sweep once the list is half again as long as its sorted part) the stress stream -- new blocks every
    frame, blocks deactivated and found again -- must give the map and the images of the unsorted run, bit for bit, on one handle and on two row-sharded
    replicas (whose sweeps see their own, differently ordered lists)."""
    from supereight_amd.synthetic import StressStream
    Ws, Hs, Ns, frames = 320, 240, 512, 40
    s = StressStream(Ws, Hs, 4.8)
    depths = [s.depth(f) for f in range(frames)]
    poses = [s.pose(f) for f in range(frames)]
    k = np.ascontiguousarray(s.k, np.float32)

    def run(sort):
        monkeypatch.delenv("SE_HIP_SORT_BLOCKS", raising=False)
        if sort:
            monkeypatch.setenv("SE_HIP_SORT_BLOCKS", "1")
        import torch
        if replicas == 1:
            ps = [DenseSLAMPipeline((Ws, Hs), Ns, 4.8, field_type=SDF)]
        else:
            half = Hs // 2
            ps = [DenseSLAMPipeline((Ws, Hs), Ns, 4.8, field_type=SDF, rows=(0, half)), DenseSLAMPipeline((Ws, Hs), Ns, 4.8, field_type=SDF, rows=(half, Hs))]
            words = 1 << 15
            send = [torch.zeros(words, dtype=torch.int64, device="cuda") for _ in ps]
            for p, buf in zip(ps, send):
                p.set_new_keys_buffer(buf.data_ptr(), words, keepalive=buf)
        for f in range(frames):
            for p in ps:
                p.set_depth(depths[f]); p.setPose(poses[f])
            if replicas == 1:
                ps[0].integration(k, 1, MU, f)
            else:
                for p in ps:
                    assert p.alloc_scan(k, 1, MU, f)
                for p in ps:
                    p.sync()
                recv = torch.cat(send)                   # what an all-gather delivers on every rank (tests/test_gpu_sharded.py)
                torch.cuda.synchronize()
                for p in ps:
                    p.alloc_commit(recv.data_ptr(), len(ps), words)
                    p.integrate_sweep(k, 1, MU, f)
                for p in ps:
                    p.sync()
        out = []
        for p in ps:
            out.append((p.blocks(), p.memory_info()["device_bytes"]))
            p.close()
        return out

    ref, got = run(False), run(True)
    for (b0, bytes0), (b1, bytes1) in zip(ref, got):
        assert len(b0[0]) > 4096                      # (the sort's own lower bound)
        assert bytes1 > bytes0                        # the sort ran: its buffers are part of the replica's memory
        for a0, a1 in zip(b0, b1):
            assert a0.shape == a1.shape and (np.ascontiguousarray(a0).view(np.uint8) == np.ascontiguousarray(a1).view(np.uint8)).all()
