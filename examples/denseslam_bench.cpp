// Throughput of the drop-in surface itself: the loop of se_apps/src/benchmark.cpp:115-177 (preprocessing -> [tracking] -> integration ->
// raycasting, wall clock per stage, one line per frame in the reference's column layout :110-112) through the C++ DenseSLAMSystem mirror
// (include/se/DenseSLAMSystem.h) -- what a supereight application that swaps the library in gets, measured without Python in the loop.
//
//   usage: denseslam_bench <stream.bin> <volume_res> <volume_dim> <mu> <warmup> <frames> [log.tsv]
//   stream.bin (written by bench.py from the synthetic stream): int32 W, H, F; float k[4]; F x { float pose[16] row-major camera->world,
//   float depth[W*H] metres }.
//
// Three passes over the same frames, each on a fresh map, ground-truth poses via setPose (tracking is not run: SURVEY 8(d)):
//   closed     the reference's bracketing: every frame ends with synchroniseDevices(), nothing of frame f+1 is issued before frame f is done;
//              depth handed over in HBM (preprocessingDevice);
//   streaming  frames issued back to back, one synchroniseDevices() at the end (poses known in advance); raycasting() of frame f is held back
//              and launched with integration(f+1)'s allocation scan -- the one-queue schedule (Configuration::hip_streaming);
//   upload     as `closed`, but every frame's depth comes from host memory as uint16 millimetres through preprocessing(): the PCIe-inclusive rate;
//   streaming + upload   as `streaming` with that host input: what the reference's own loop (read a frame, preprocessing, ..., benchmark.cpp:115-150) gets
//              when its poses do not depend on the previous frame's images -- the input of frame f+1 crosses PCIe inside the launch that raycasts frame f;
//   tracked    the reference's loop with tracking on (preprocessing -> tracking -> integration -> raycasting, nothing else between the frames), host input:
//              tracking(f+1) needs raycasting(f)'s images, so the class launches raycasts eagerly after the first frames (se_hip.h, streaming callers).
// Prints one JSON line; the per-frame log of the closed pass goes to log.tsv.
#ifndef SE_FIELD_TYPE
#define SE_FIELD_TYPE SDF
#endif
#include <se/DenseSLAMSystem.h>

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace {
typedef std::chrono::steady_clock Clock;
double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct Stream {
  int W = 0, H = 0, F = 0;
  float k[4] = {0, 0, 0, 0};
  std::vector<Eigen::Matrix4f> pose;
  std::vector<float> depth;               // F x W*H, host
  std::vector<unsigned short> depth_mm;   // the same frames as the sensor delivers them
  unsigned short* depth_mm_pinned = nullptr;   // ... and in page-locked memory (DenseSLAMSystem::allocateInput): what a reader that decodes into pinned buffers hands over
  float* dev = nullptr;                   // F x W*H, HBM
  bool load(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    int32_t hdr[3];
    if (std::fread(hdr, 4, 3, f) != 3 || std::fread(k, 4, 4, f) != 4) return false;
    W = hdr[0]; H = hdr[1]; F = hdr[2];
    const size_t n = (size_t)W * H;
    pose.resize(F); depth.resize(n * F); depth_mm.resize(n * F);
    for (int i = 0; i < F; ++i) {
      float p[16];
      if (std::fread(p, 4, 16, f) != 16 || std::fread(&depth[n * i], 4, n, f) != n) return false;
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) pose[i](r, c) = p[r * 4 + c];
    }
    std::fclose(f);
    for (size_t i = 0; i < depth.size(); ++i) depth_mm[i] = (unsigned short)(depth[i] * 1000.f + 0.5f);
    depth_mm_pinned = (unsigned short*)DenseSLAMSystem::allocateInput(depth_mm.size() * sizeof(unsigned short));
    if (!depth_mm_pinned) return false;
    std::copy(depth_mm.begin(), depth_mm.end(), depth_mm_pinned);
    if (hipMalloc((void**)&dev, depth.size() * sizeof(float)) != hipSuccess) return false;
    return hipMemcpy(dev, depth.data(), depth.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
  }
};

Configuration make_config(int res, float dim, float mu, const float k[4], bool streaming, bool pinned) {
  Configuration c;
  c.compute_size_ratio = 1; c.tracking_rate = 1; c.integration_rate = 1; c.rendering_rate = 4;
  c.volume_resolution = Eigen::Vector3i(res, res, res); c.volume_size = Eigen::Vector3f(dim, dim, dim);
  c.voxel_block_size = 8; c.initial_pos_factor = Eigen::Vector3f(0.f, 0.f, 0.f); c.pyramid = {10, 5, 4};
  c.gt_transform = Eigen::Matrix4f::Identity(); c.camera = Eigen::Vector4f(k[0], k[1], k[2], k[3]); c.camera_overrided = true;
  c.mu = mu; c.fps = 0; c.blocking_read = false; c.icp_threshold = 1e-5f; c.no_gui = true;
  c.render_volume_fullsize = false; c.bilateralFilter = false; c.colouredVoxels = false; c.multiResolution = false; c.bayesian = false;
  c.hip_streaming = streaming;
  c.hip_pinned_input = pinned;
  return c;
}

struct Pass { double fps = 0, integration_ms = 0, raycasting_ms = 0, preprocessing_ms = 0, tracking_ms = 0; int blocks = 0; };

enum Mode { CLOSED, STREAMING, UPLOAD, STREAMING_UPLOAD, TRACKED };

Pass run(const Stream& s, int res, float dim, float mu, int warm, int frames, Mode mode, FILE* log, bool pinned = false) {
  std::vector<int> pyramid = {10, 5, 4};
  const bool stream_mode = mode == STREAMING || mode == STREAMING_UPLOAD;
  const Configuration config = make_config(res, dim, mu, s.k, mode != CLOSED && mode != UPLOAD, pinned);
  const unsigned short* host_mm = pinned ? s.depth_mm_pinned : s.depth_mm.data();
  DenseSLAMSystem pipeline(Eigen::Vector2i(s.W, s.H), config.volume_resolution, config.volume_size, Eigen::Vector3f(0.f, 0.f, 0.f), pyramid, config);
  const Eigen::Vector4f camera(s.k[0], s.k[1], s.k[2], s.k[3]);
  const size_t n = (size_t)s.W * s.H;
  Pass out;
  if (log) std::fprintf(log, "frame\tacquisition\tpreprocessing\ttracking\tintegration\traycasting\trendering\tcomputation\ttotal    \tX          \tY          \tZ         \ttracked   \tintegrated\n");
  Clock::time_point t_begin, timings[7];
  timings[0] = Clock::now();
  for (int frame = 0; frame < warm + frames; ++frame) {
    if (frame == warm) { synchroniseDevices(); t_begin = Clock::now(); timings[0] = t_begin; }
    timings[1] = Clock::now();
    if (mode == UPLOAD || mode == STREAMING_UPLOAD || mode == TRACKED) pipeline.preprocessing(host_mm + n * frame, Eigen::Vector2i(s.W, s.H), false);
    else pipeline.preprocessingDevice(s.dev + n * frame);
    timings[2] = Clock::now();
    bool tracked = true;
    if (mode == TRACKED && frame > 3) tracked = pipeline.tracking(camera, config.icp_threshold, config.tracking_rate, (unsigned)frame);
    else pipeline.setPose(s.pose[frame]);         // ground truth in place of tracking() (se_apps/src/mainQt.cpp:257-265 does the same)
    timings[3] = Clock::now();
    const bool integrated = (tracked || frame <= 3) ? pipeline.integration(camera, config.integration_rate, config.mu, (unsigned)frame) : false;   // benchmark.cpp:137-147
    timings[4] = Clock::now();
    pipeline.raycasting(camera, config.mu, (unsigned)frame);
    if (!stream_mode && mode != TRACKED) synchroniseDevices();  // the reference's kernels are synchronous: its raycasting column ends when the images exist
    timings[5] = Clock::now();
    timings[6] = timings[5];                      // (no rendering in the measured loop)
    if (frame >= warm) {
      out.preprocessing_ms += 1e3 * secs(timings[1], timings[2]);
      out.tracking_ms += 1e3 * secs(timings[2], timings[3]);
      out.integration_ms += 1e3 * secs(timings[3], timings[4]);
      out.raycasting_ms += 1e3 * secs(timings[4], timings[5]);
    }
    if (log) {
      const Eigen::Vector3f pos = pipeline.getPosition();
      std::fprintf(log, "%d\t%.6f\t%.6f\t%.6f\t%.6f\t%.6f\t%.6f\t%.6f\t%.6f\t%.6f\t%.6f\t%.6f\t%d        \t%d\n", frame, secs(timings[0], timings[1]),
                   secs(timings[1], timings[2]), secs(timings[2], timings[3]), secs(timings[3], timings[4]), secs(timings[4], timings[5]),
                   secs(timings[5], timings[6]), secs(timings[1], timings[5]), secs(timings[0], timings[6]), pos.x(), pos.y(), pos.z(), 0, (int)integrated);
    }
    timings[0] = Clock::now();
  }
  synchroniseDevices();
  const double total = secs(t_begin, Clock::now());
  out.fps = frames / total;
  out.preprocessing_ms /= frames; out.integration_ms /= frames; out.raycasting_ms /= frames; out.tracking_ms /= frames;
  MapSnapshot snap;
  int nb = 0, nn = 0;
  se_hip_counts(pipeline.handle(), &nb, &nn);
  out.blocks = nb;
  return out;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 7) { std::fprintf(stderr, "usage: %s stream.bin res dim mu warmup frames [log.tsv]\n", argv[0]); return 2; }
  Stream s;
  if (!s.load(argv[1])) { std::fprintf(stderr, "cannot load %s\n", argv[1]); return 2; }
  const int res = std::atoi(argv[2]);
  const float dim = (float)std::atof(argv[3]), mu = (float)std::atof(argv[4]);
  const int warm = std::atoi(argv[5]), frames = std::atoi(argv[6]);
  if (warm < 4 || frames < 1 || warm + frames > s.F) { std::fprintf(stderr, "need 4 <= warmup and warmup + frames <= %d\n", s.F); return 2; }
  FILE* log = argc > 7 ? std::fopen(argv[7], "w") : nullptr;
  // one throw-away pass first: clocks, code objects and the allocator warm (each measured pass then starts from the same state)
  run(s, res, dim, mu, warm, std::min(frames, 20), STREAMING, nullptr);
  const Pass closed = run(s, res, dim, mu, warm, frames, CLOSED, log);
  const Pass streaming = run(s, res, dim, mu, warm, frames, STREAMING, nullptr);
  const Pass upload = run(s, res, dim, mu, warm, frames, UPLOAD, nullptr);
  const Pass stream_up = run(s, res, dim, mu, warm, frames, STREAMING_UPLOAD, nullptr);
  const Pass tracked = run(s, res, dim, mu, warm, frames, TRACKED, nullptr);
  // the three host-input passes again with the frames in page-locked memory and Configuration::hip_pinned_input: no copy into the handle's own ring
  const Pass upload_p = run(s, res, dim, mu, warm, frames, UPLOAD, nullptr, true);
  const Pass stream_up_p = run(s, res, dim, mu, warm, frames, STREAMING_UPLOAD, nullptr, true);
  const Pass tracked_p = run(s, res, dim, mu, warm, frames, TRACKED, nullptr, true);
  if (log) std::fclose(log);
  std::printf("{\"surface\": \"DenseSLAMSystem (include/se/DenseSLAMSystem.h) over libse_hip.so\", \"frames\": %d, \"warmup\": %d, \"blocks\": %d, "
              "\"closed_loop_fps\": %.1f, \"closed_loop_stage_ms\": {\"preprocessing\": %.4f, \"integration\": %.4f, \"raycasting\": %.4f}, "
              "\"streaming_fps\": %.1f, \"streaming_enqueue_ms\": {\"preprocessing\": %.4f, \"integration\": %.4f, \"raycasting\": %.4f}, "
              "\"closed_loop_fps_with_upload\": %.1f, \"upload_stage_ms\": {\"preprocessing\": %.4f, \"integration\": %.4f, \"raycasting\": %.4f}, "
              "\"streaming_fps_with_upload\": %.1f, \"streaming_upload_enqueue_ms\": {\"preprocessing\": %.4f, \"integration\": %.4f, \"raycasting\": %.4f}, "
              "\"tracked_fps_with_upload\": %.1f, \"tracked_stage_ms\": {\"preprocessing\": %.4f, \"tracking\": %.4f, \"integration\": %.4f, \"raycasting\": %.4f}, "
              "\"pinned_input\": {\"closed_loop_fps_with_upload\": %.1f, \"preprocessing_ms\": %.4f, \"streaming_fps_with_upload\": %.1f, \"tracked_fps_with_upload\": %.1f}}\n",
              frames, warm, closed.blocks, closed.fps, closed.preprocessing_ms, closed.integration_ms, closed.raycasting_ms, streaming.fps,
              streaming.preprocessing_ms, streaming.integration_ms, streaming.raycasting_ms, upload.fps, upload.preprocessing_ms, upload.integration_ms,
              upload.raycasting_ms, stream_up.fps, stream_up.preprocessing_ms, stream_up.integration_ms, stream_up.raycasting_ms,
              tracked.fps, tracked.preprocessing_ms, tracked.tracking_ms, tracked.integration_ms, tracked.raycasting_ms,
              upload_p.fps, upload_p.preprocessing_ms, stream_up_p.fps, tracked_p.fps);
  hipFree(s.dev);
  DenseSLAMSystem::freeInput(s.depth_mm_pinned);
  return 0;
}
