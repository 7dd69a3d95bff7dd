"""renderDepth / renderTrack / renderVolume on the device against the oracle (byte-exact)."""
import numpy as np
import pytest

from oracle.binding import OFUSION, SDF, OraclePipeline, load, oracle_tracking
from supereight_amd.pipeline import DenseSLAMPipeline
from supereight_amd.synthetic import SyntheticStream

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("field,mu", [(SDF, 0.1), (OFUSION, 0.02)], ids=["sdf", "ofusion"])
def test_render_kernels(field, mu):
    W, H, N, dim, frames = 320, 240, 256, 4.8, 6
    lib = load()
    s = SyntheticStream(W, H, dim)
    cpu = OraclePipeline(field, N, dim, W, H)
    gpu = DenseSLAMPipeline((W, H), N, dim, field_type=field)
    for f in range(frames):
        depth, pose = s.depth(f), s.pose(f)
        gpu.set_depth(depth); gpu.setPose(pose)
        cpu.integrate(depth, pose, s.k, mu, f); gpu.integration(s.k, 1, mu, f)
        ran, v, n = cpu.raycast(pose, s.k, mu, f); gpu.raycasting(s.k, mu, f)
    # renderDepth
    ref = np.zeros((H, W, 4), np.uint8)
    lib.so_render_depth(ref.reshape(-1), np.ascontiguousarray(depth, np.float32).reshape(-1), W, H)
    assert (gpu.renderDepth() == ref).all() and len(np.unique(ref.reshape(-1, 4), axis=0)) > 50
    # renderVolume, view == raycast pose: shades the cached vertex / normal maps
    largestep = 0.75 * mu
    a = gpu.renderVolume(pose, s.k, mu, largestep)
    b = cpu.render_volume(pose, pose, s.k, mu, largestep, v, n)
    assert (a == b).all() and (a[..., 0] > 0).mean() > 0.7
    assert gpu.renderVolume(pose, s.k, mu, largestep, frame=3, rate=2) is None      # frame % rate gate
    # renderVolume from another view: re-raycasts from the near plane with far = 2 * farPlane
    view = s.pose(frames + 20)
    a = gpu.renderVolume(view, s.k, mu, largestep)
    b = cpu.render_volume(view, pose, s.k, mu, largestep, v, n)
    assert (a == b).all() and (a[..., 0] > 0).mean() > 0.5
    # renderTrack after one tracking call
    d2 = s.depth(frames)
    gpu.set_depth(d2)
    ok_g = gpu.tracking(s.k, 1e-5, 1, frames)
    ok_c, pose_c, track_c, red_c, it_c = oracle_tracking(d2, s.k, pose, pose, v, n)
    ref = np.zeros((H, W, 4), np.uint8)
    lib.so_render_track(ref.reshape(-1), track_c.ctypes.data, W, H)
    assert ok_g == ok_c and (gpu.renderTrack() == ref).all()
    cpu.close(); gpu.close()
