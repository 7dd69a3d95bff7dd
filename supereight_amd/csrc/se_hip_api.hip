// C ABI (include/se_hip.h) of the gfx950 dense-fusion path: handle management, per-frame
// host-side matrix set-up (the few 4x4 products DenseSLAMSystem.cpp does before calling its
// kernels) and kernel launches.  There is no CPU fallback: without a usable HIP device
// se_hip_create() fails with SE_HIP_E_NOGPU.
#include "../../include/se_hip.h"

#include <hip/hip_runtime.h>
#include <hipcub/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <numeric>
#include <string>
#include <vector>

#include "se_kernels.h"
#include "se_track_kernels.h"
#include "se_mesh_kernels.h"

int flush_pending_raycast(se_hip_pipeline* p);   // (defined next to se_hip_frame)

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail(SE_HIP_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// ---- host-side matrix helpers (row-major m[r][c]); same operation order as the kernels use
struct M4 { float m[4][4]; };
M4 from_colmajor(const float* p) {
  M4 a;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) a.m[r][c] = p[c * 4 + r];
  return a;
}
M4 mul(const M4& a, const M4& b) {
  M4 c;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      c.m[i][j] = ((a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j]) + a.m[i][3] * b.m[3][j];
  return c;
}
// getCameraMatrix / getInverseCameraMatrix (se_denseslam/include/se/commons.h:255-271)
M4 camera_matrix(const float k[4]) { return {{{k[0], 0, k[2], 0}, {0, k[1], k[3], 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}; }
M4 inverse_camera_matrix(const float k[4]) {
  return {{{1.0f / k[0], 0, -k[2] / k[0], 0}, {0, 1.0f / k[1], -k[3] / k[1], 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
}
void mul3(const float r[9], const float v[3], float out[3]) {
  out[0] = (r[0] * v[0] + r[1] * v[1]) + r[2] * v[2];
  out[1] = (r[3] * v[0] + r[4] * v[1]) + r[5] * v[2];
  out[2] = (r[6] * v[0] + r[7] * v[1]) + r[8] * v[2];
}

struct TimedLaunch { int kernel; hipEvent_t start, stop; };

// spin-wait hint, portable (the host side must also build on aarch64 hosts)
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
  asm volatile("yield" ::: "memory");
#else
  std::this_thread::yield();
#endif
}
// Polls `done()` for at most `limit_us` of wall-clock time (checked every 256 polls); false = gave up.
// A successful poll is followed by an acquire fence: what the device wrote before the word the caller polls (the ICP record behind
// its sequence number) must not be read by loads hoisted above the poll (aarch64 hosts, and the C++ memory model).
template <typename F> bool spin_until(F done, long long limit_us) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned n = 1;; ++n) {
    if (done()) { std::atomic_thread_fence(std::memory_order_acquire); return true; }
    cpu_relax();
    if ((n & 255u) == 0u && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > limit_us) {
      const bool ok = done();
      std::atomic_thread_fence(std::memory_order_acquire);
      return ok;
    }
  }
}

}  // namespace

struct se_hip_pipeline {
  se_hip_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // side stream: the allocation scan (and the depth upload feeding it) of frame f+1 runs here,
  // concurrently with the raycast of frame f on `stream`
  hipStream_t side = nullptr;
  bool own_side = false;       // false after se_hip_set_scan_stream handed one in
  hipEvent_t ev_sweep = nullptr, ev_scan = nullptr;
  bool overlap = false;
  // host gate (see RayArgs::gate): replaces the event between the sweep and the next frame's scan for unsharded replicas
  bool host_gate = false;
  uint32_t* gate_host = nullptr;   // pinned word the raycast kernel writes its sequence number to
  uint32_t ray_seq = 0;            // raycast launches so far
  uint32_t gate_target = 0;        // sequence number of the first raycast behind the last sweep
  bool gate_armed = false;         // a sweep was enqueued that no scan / upload has waited for yet
  bool gate_followed = false;      // ... and a raycast was enqueued behind it
  bool sharded = false;        // this replica scans / raycasts a row range of the image (multi-GPU)
  // One queue for streaming callers (se_hip_set_streaming, off by default): se_hip_frame / se_hip_raycast_deferred defer a frame's raycast to the next
  // frame's allocation scan, which launches it in ONE kernel with that scan on the main stream (k_raycast_scan) -- no second queue, no wait in front of
  // the sweep.  Any other API call launches the deferred raycast first (check()), so results are always in place when somebody looks through the API;
  // a caller that was handed the raw image pointers (se_hip_vertex_normal_device, no image ring) could look past the API: that switches deferral off for good.
  bool fuse = false, in_frame = false, has_pending = false;
  // A held-back raycast that some other call has to launch (se_hip_track needs vertex_ / normal_, a getter, a per-frame se_hip_sync) was held back for
  // nothing: it starts later than the eager schedule would have started it and fuses with no scan (the reference's loop with tracking on).
  // Two such launches in a row without a fused one in between -> this caller looks at every frame: raycasts are launched eagerly from then on
  // (se_hip_set_streaming(p, 1) arms deferral again).
  int flush_streak = 0;
  bool pinned_input = false;   // se_hip_set_pinned_input: page-locked caller images are read in place
  // dense grid: the block list kept (mostly) in address order, see sort_block_list
  int sort_every = 0;           // 0 = off
  uint32_t* bpos_sorted = nullptr; void* sort_tmp = nullptr; size_t sort_tmp_bytes = 0, sort_cap = 0;
  uint32_t sorted_n = 0; int sweeps_since_sort = 0;
  bool ptrs_exposed = false;   // sticky: se_hip_vertex_normal_device handed out vertex_ / normal_ of a handle without an image ring
  float pend_pose[16] = {0}, pend_k[4] = {0}, pend_mu = 0.f;
  uint32_t pend_frame = 0;
  bool images_complete = true; // vertex_ / normal_ hold every row of the last raycast (a row-sharded replica: only after se_hip_apply_image_tiles / se_hip_gather_images)
  bool scan_pending = false;   // a scan was enqueued on `side` and not yet joined by `stream`
  bool scan_on_side = false;   // stream the LAST allocation scan ran on: se_hip_alloc_exchange / se_hip_alloc_commit follow it
  const float* scaled0 = nullptr;   // scaled_depth_[0] of the last se_hip_track
  // direct RCCL exchange of the key lists (se_hip_set_exchange): the caller's communicator and ncclAllGather
  void* xcomm = nullptr;
  int (*xgather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int xworld = 0;
  // sharded sweep (se_hip_set_sweep_shard): owner-computes + brick exchange instead of the replicated sweep
  int shard_world = 0, shard_rank = 0;
  unsigned char* shard_send = nullptr;   // caller's send segment
  size_t shard_cap = 0;                  // bricks per segment
  bool mc_table_ready = false;   // SE_MC_TRI uploaded to constant memory
  unsigned long long* mesh_ctr = nullptr;
  bool filter_input = false;   // preprocessing(..., filterInput): tracking sees the bilateral-filtered depth
  bool occ_commit_due = false; // the next sweep kernel must publish the scan's occupancy bits
  OccLists occ_lists{nullptr, 0, 0};   // ... of these key lists (own list, or every rank's after se_hip_alloc_commit)
  // tracking
  float raycast_pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   // column-major, pose of the last raycast
  std::vector<float*> pyr_depth, pyr_vertex, pyr_normal;  // level 0 depth aliases the current depth image
  TrackData* track = nullptr;
  float* reduce_partial = nullptr;   // 8 x SE_TRACK_SEGMENTS x 32 partial sums of the running ICP iteration
  IcpState* icp = nullptr;           // device: what one ICP iteration hands to the next (two copies: launch j reads [j & 1], writes [(j + 1) & 1])
  IcpHostRecord* icp_host = nullptr; // pinned: the one record the host reads per tracked frame
  unsigned reduce_seq = 0;
  int track_iterations = 0;
  int icp_lookahead = 2;             // launches of a level the host stays ahead of the device (0: all iterations enqueued up front; se_hip_track)
  bool scan_on_main_once = false;    // se_hip_frame_tracked: this frame's scan follows the ICP on the main stream (nothing to overlap with)
  unsigned char* rgbw = nullptr;   // render target (W*H*4)
  DevMap map{};
  int leaf_level = 0, max_level = 0;
  size_t tab_entries = 0;
  size_t occ_words = 0, lbits_words = 0, cbits_words = 0, fbits_words = 0;
  bool of_leap = true;         // OFusion march: leap over block-free space (se_of_leap); SE_HIP_OF_LEAP=0 turns it off (A/B knob, same results)
  int beam = 2;                // raycast: beam start (se_beam_start): 0 off, 1 coarse stage only, 2 both stages; SE_HIP_BEAM (A/B knob: results are the same either way)
  size_t slots = 0;
  size_t cap_blocks = 0, cap_nodes = 0;
  int ray_cache_levels = -1;  // -1: choose automatically
  const float* depth = nullptr;     // what the kernels read: a slot of the handle's own ring or the caller's device image (se_hip_set_depth_device)
  // Host input (se_hip_upload_depth / se_hip_upload_depth_mm; r06): the caller's image is copied into a slot of a ring of pinned host buffers -- the call
  // returns when the caller's buffer has been read, as the reference's synchronous preprocessing() does -- and NOTHING is enqueued: the first kernel that
  // needs float_depth_ reads the pinned image over PCIe and writes the matching slot of the device ring (DepthSrc, se_kernels.h): the frame's allocation
  // scan, or k_depth_from_host in front of a consumer that comes earlier (tracking, renderDepth, a row-sharded scan).  No DMA packet, no second queue,
  // no wait for the previous sweep: slot i is reused kIn uploads later, once a raycast enqueued behind the last sweep that read it has started (the
  // host gate's sequence word; without a host gate: an event recorded behind that sweep).
  static constexpr int kIn = 3;
  float* depth_ring[kIn] = {nullptr, nullptr, nullptr};   // device, width*height floats each
  void* in_host[kIn] = {nullptr, nullptr, nullptr};       // pinned
  size_t in_cap = 0;
  int in_next = 0;
  int in_state[kIn] = {0, 0, 0};            // 0 free, 1 read by enqueued work that no raycast launch has followed yet, 2 followed: reusable once gate_host reaches in_seq
  uint32_t in_seq[kIn] = {0, 0, 0};
  hipEvent_t in_done[kIn] = {nullptr, nullptr, nullptr};   // handles without a host gate
  bool in_event[kIn] = {false, false, false};
  int cur_in = -1;                          // ring slot p->depth points into (-1: the caller's device image)
  DepthSrc in_pending{nullptr, nullptr, 0, 0, 0};   // host-resident input not yet materialised on the device
  float* vertex = nullptr;       // vertex_ / normal_ as every consumer (tracking, rendering, the getters) sees them: the images of the LAST raycast
  float* normal = nullptr;       // launched -- the handle's own buffers, or the slot of the image ring that raycast wrote
  float* vertex_own = nullptr;
  float* normal_own = nullptr;
  float* ring = nullptr;         // se_hip_set_image_ring: slot s = [vertex W*H*3 floats][normal W*H*3 floats] at ring + s * 2 * W*H*3
  int ring_slots = 0;
  int64_t n_launch[SE_HIP_K_COUNT + 1] = {0};   // kernel launches per kind since the last reset, counted on the host (no events needed); [SE_HIP_K_COUNT] = fused
  float* bspline = nullptr;
  float* logodds = nullptr;
  unsigned long long* chain = nullptr;  // 3 candidates for the keys[0] quirk
  unsigned long long* newkeys_own = nullptr;    // the handle's own key lists: two, used alternately by successive scans; the sweep
  unsigned long long* newkeys_own2 = nullptr;   // kernel of a frame clears the count word of the list the next scan will use
  unsigned long long cap_keys_own = 0;
  int own_next = 0;                             // which own list the next scan appends to
  bool own_clean[2] = {false, false};           // its count word is known to be zero (cleared by a sweep or a memset, not used since)
  uint32_t* ctr_host = nullptr;         // pinned
  bool timing = false, stats = false;
  std::vector<TimedLaunch> pending;
  std::vector<hipEvent_t> event_pool;
  double ms_sum[SE_HIP_K_COUNT] = {0};
  int64_t launches[SE_HIP_K_COUNT] = {0};
  int row_begin = 0, row_end = 0;
  int integ_grid = 0;  // > 0: fixed number of workgroups for the integration sweep (tuning knob)
  bool ieee_sweep = false;   // SE_HIP_IEEE_SWEEP=1: never select the shared-reciprocal sweep (k_integrate<.., FAST = false, ..>)
  unsigned short* tile_cost = nullptr;   // raycast scheduling hint: per wave tile, cost in the previous launch (see RayArgs)
  int* prio_thr = nullptr;               // its three priority thresholds (device; written by the integration sweep)
  bool prio_hint = true;                 // SE_HIP_PRIO=0 switches the hint off
  int prio_permille[3] = {400, 150, 50}; // share of the tiles raised to priority >= 1 / >= 2 / 3 (SE_HIP_PRIO_SHARE="a,b,c", per mille)
  uint32_t* ray_order = nullptr;   // raycast schedule: the workgroups' tile pairs by descending previous cost (RayArgs::ray_order)
  int n_cus = 256;
};

namespace {

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

hipEvent_t get_event(se_hip_pipeline* p) {
  if (!p->event_pool.empty()) { hipEvent_t e = p->event_pool.back(); p->event_pool.pop_back(); return e; }
  hipEvent_t e;
  // timing only: without the system-scope fence a record would otherwise put between the kernels it brackets
  hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
  return e;
}
struct ScopedTimer {
  se_hip_pipeline* p; int k; hipStream_t s; hipEvent_t a{}, b{};
  ScopedTimer(se_hip_pipeline* p_, int k_, hipStream_t s_ = nullptr) : p(p_), k(k_), s(s_ ? s_ : p_->stream) {
    p->n_launch[k]++;
    if (p->timing) { a = get_event(p); b = get_event(p); hipEventRecord(a, s); }
  }
  ~ScopedTimer() {
    if (p->timing) { hipEventRecord(b, s); p->pending.push_back({k, a, b}); }
  }
};
void drain_timings(se_hip_pipeline* p) {
  if (p->pending.empty()) return;
  if (p->side) hipStreamSynchronize(p->side);
  hipStreamSynchronize(p->stream);
  for (auto& t : p->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) { p->ms_sum[t.kernel] += ms; p->launches[t.kernel]++; }
    p->event_pool.push_back(t.start);
    p->event_pool.push_back(t.stop);
  }
  p->pending.clear();
}

// bspline_lookup (se_denseslam/src/bfusion/bspline_lookup.cc:36-37): the reference's 1000 literals
// are the cubic B-spline CDF at t_i = -3 + 6 i / 999 printed with 12 significant digits; they are
// regenerated here (tests check the regeneration against the reference's literals).
double bspline_cdf(double t) {
  if (t >= -3.0 && t <= -1.0) return std::pow(3 + t, 3) / 48.0;
  if (t > -1 && t <= 1) return 0.5 + (t * (3 + t) * (3 - t)) / 24.0;
  if (t > 1 && t <= 3) return 1 - std::pow(3 - t, 3) / 48.0;
  if (t > 3) return 1.0;
  return 0.0;
}
void make_bspline(std::vector<float>& lut) {
  lut.resize(1000);
  for (int i = 0; i < 1000; ++i) {
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.12g", bspline_cdf(-3.0 + 6.0 * i / 999.0));
    lut[i] = (float)std::strtod(buf, nullptr);
  }
}
// updateLogs() of the reference calls the C library's log2f (se_denseslam/src/bfusion/mapping_impl.hpp:146-149).
// HNew() only ever yields sample = Q1 - 0.5*Q2 with Q1, Q2 drawn from the 1000-entry table (or the
// constants 0 and 1), so log2f(s / (1 - s)) after the [0.03, 0.97] clamp is tabulated here for every
// index pair with the same C library call; the kernel then needs no transcendental at all.
void make_logodds(const std::vector<float>& lut, std::vector<float>& tab) {
  tab.resize((size_t)SE_LO_DIM * SE_LO_DIM);
  auto q = [&](int i) { return i < 1000 ? lut[i] : (i == 1001 ? 1.f : 0.f); };
  for (int i1 = 0; i1 < SE_LO_DIM; ++i1)
    for (int i2 = 0; i2 < SE_LO_DIM; ++i2) {
      float sample = q(i1) - q(i2) * 0.5f;
      sample = std::max(0.03f, std::min(sample, 0.97f));
      tab[(size_t)i1 * SE_LO_DIM + i2] = log2f(sample / (1.f - sample));
    }
}

int run_zero_chain(se_hip_pipeline* p, const unsigned long long* lists, int nlists, long long stride_words) {
  ScopedTimer t(p, SE_HIP_K_ALLOC_COMMIT);
  hipLaunchKernelGGL(k_zero_chain, dim3(1), dim3(SE_WG), 0, p->stream, p->map, lists, nlists, stride_words);
  return SE_HIP_OK;
}


M4 rigid_inverse(const M4& a) {   // raycast_pose_.inverse() of a rigid transform (DenseSLAMSystem.cpp:164)
  M4 r{};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  for (int i = 0; i < 3; ++i) r.m[i][3] = -((r.m[i][0] * a.m[0][3] + r.m[i][1] * a.m[1][3]) + r.m[i][2] * a.m[2][3]);
  r.m[3][3] = 1.f;
  return r;
}

struct RayLaunchArgs { RayArgs a; size_t smem; dim3 grid; };

// RayArgs of raycastKernel for pose * K^-1 (DenseSLAMSystem.cpp:197-200)
RayLaunchArgs make_ray_args(se_hip_pipeline* p, const float pose_cm[16], const float k[4], float mu) {
  RayLaunchArgs L{};
  const DevMap& m = p->map;
  const M4 view = mul(from_colmajor(pose_cm), inverse_camera_matrix(k));  // DenseSLAMSystem.cpp:199
  RayArgs& a = L.a;
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) a.view3[i * 3 + j] = view.m[i][j]; a.org[i] = view.m[i][3]; }
  a.nearp = 0.4f; a.farp = 4.0f;  // constant_parameters.h:22-32
  a.near_n = a.nearp / m.dim; a.far_n = a.farp / m.dim;
  for (int i = 0; i < 3; ++i) a.scaled_origin[i] = a.org[i] / m.dim + 1.f;
  a.mu = mu;
  a.step = m.dim / (float)m.size;           // DenseSLAMSystem.cpp:197
  a.largestep = a.step * 8;                 // step * BLOCK_SIDE
  a.inv_voxel = (float)m.size / m.dim;      // volume_template.hpp:78
  a.grad_scale = 0.5f * m.dim / (float)m.size;  // octree.hpp:736
  a.epsilon = exp2f(-(float)p->max_level);  // ray_iterator.hpp:63
  a.min_scale = 23 - p->leaf_level;         // ray_iterator.hpp:62
  a.W = p->cfg.width; a.H = p->cfg.height; a.row_begin = p->row_begin; a.row_end = p->row_end;
  // occupancy levels staged in LDS: levels 1..5 (4.7 KB; measured: level 6 = +32 KB costs more
  // occupancy and staging time than the leaf-level bit tests it saves) unless overridden
  int cl = p->ray_cache_levels >= 0 ? p->ray_cache_levels : 5;
  cl = std::min(cl, p->leaf_level);
  a.cache_levels = cl;
  // staged region = words [0, end of level cl)
  a.cache_words = (int)occ_words_upto(cl);
  a.cache_codes = 2u << (3 * cl);
  a.has_deep = cl < p->leaf_level - 1 ? 1 : 0;
  a.stack_depth = p->leaf_level;
  a.tile_cost = p->prio_hint ? p->tile_cost : nullptr;
  a.prio_thr = p->prio_thr;
  a.cost_shift = std::max(0, p->leaf_level - 6);
  a.ray_order = p->ray_order; a.n_cus = p->n_cus;
  // beam start: 64 samples on a tile's centre ray, half a coarse cell apart (or whatever spacing covers near .. far)
  a.beam = (p->beam >= 2 && !m.fbits) ? 1 : p->beam;
  const int flevel = m.fbits ? se_flevel(m) : m.leaf_level;   // (no fbits: leaf_level <= clevel, the coarse grid is the block grid)
  a.beam_cellf = m.dim / (float)(1 << flevel);
  a.beam_inv_cellf = (float)(1 << flevel) / m.dim;
  a.beam_dt2 = 0.4f * a.beam_cellf;
  if (a.beam >= 2) {
    // the fine stage pays (one more dependent load per wave) only where its clearance bound can hold at working distance: an 8x8 pixel beam at 3/4 of the
    // far plane must fit inside one cell's margin.  r05 tested on the block grid, where that fails from 1024^3 on (3.75 cm blocks); the level-6 grid (7.5 cm)
    // holds it for any 640x480-like camera, at every volume resolution
    const float rad = 1.05f * std::hypot(0.5f * (SE_TILE_W - 1) / std::fabs(k[0]), 0.5f * (SE_TILE_H - 1) / std::fabs(k[1])) + a.epsilon;
    if (!((0.75f * a.farp + 0.5f * a.beam_dt2) * rad + 0.5f * a.beam_dt2 <= 0.9f * a.beam_cellf)) a.beam = 1;
  }
  a.beam_cell = m.dim / (float)(1 << m.clevel);
  a.beam_inv_cell = (float)(1 << m.clevel) / m.dim;
  a.beam_dt = std::max(0.5f * a.beam_cell, (a.farp - a.nearp) / 64.f);
  a.inv_dim = 1.f / m.dim;
  // OFusion march: leap over block-free space on the dilated block grid (fbits; for leaf levels <= 5 the coarse grid IS the block grid)
  a.leap_bits = !p->of_leap ? nullptr : m.fbits ? m.fbits : (m.clevel == m.leaf_level ? m.cbits : nullptr);
  a.leap_level = flevel;
  a.leap_dt = 0.9f * a.beam_cellf;
  a.wlog = nullptr;
#ifdef SE_WAVE_PROBE
  {
    // probe builds (tools/wave_timeline.py): the per-wave records of the PREVIOUS raycast launch are written to $SE_HIP_WLOG when the next one is prepared
    static unsigned long long* wl = nullptr;
    static bool tried = false;
    if (!tried) { tried = true; if (std::getenv("SE_HIP_WLOG")) { hipHostMalloc((void**)&wl, 4 * 8 * 32768, 0); std::memset(wl, 0, 4 * 8 * 32768); } }
    a.wlog = wl;
    if (wl) { hipStreamSynchronize(p->stream); if (FILE* f = std::fopen(std::getenv("SE_HIP_WLOG"), "wb")) { std::fwrite(wl, 8, 4 * 32768, f); std::fclose(f); } std::memset(wl, 0, 4 * 8 * 32768); }
  }
#endif
  L.smem = ((size_t)a.cache_words + (size_t)2 * a.stack_depth * SE_WG_RAY) * sizeof(uint32_t);
  const int tiles_x = (a.W + SE_TILE_W - 1) / SE_TILE_W, tiles_y = (a.row_end - a.row_begin + SE_TILE_H - 1) / SE_TILE_H;
  // one workgroup per tile pair, in whole rounds over the compute units (workgroups of the last round beyond the list idle)
  const int n_pairs = (tiles_x * tiles_y + 1) / 2;
  L.grid = dim3((unsigned)(((n_pairs + p->n_cus - 1) / p->n_cus) * p->n_cus));
  return L;
}

// Host gate: returns once the last integration sweep has completed -- known from the sequence word the raycast kernel
// behind it wrote when it started (no raycast behind it: the frames before the first raycast, integration-only
// callers -> a stream synchronisation).
void wait_last_sweep(se_hip_pipeline* p) {
  if (!p->gate_armed) return;
  p->gate_armed = false;
  if (p->gate_followed) {
    // a frame period normally (the raycast behind the sweep starts within tens of microseconds); 20 ms of wall clock at most,
    // then the ordinary stream synchronisation -- e.g. when the caller's stream is held behind work of its own
    volatile uint32_t* w = p->gate_host;
    const uint32_t target = p->gate_target;
    if (!spin_until([&] { return (int32_t)(*w - target) >= 0; }, 20000)) hipStreamSynchronize(p->stream);
    return;
  }
  hipStreamSynchronize(p->stream);
}

// float_depth_ on the device for a consumer on stream `s` that is not the frame's allocation scan: a host-resident input is converted / copied by
// one kernel in stream order (reads the pinned image over PCIe)
void materialise_depth(se_hip_pipeline* p, hipStream_t s) {
  if (p->in_pending.kind == 0) return;
  const int W = p->cfg.width, H = p->cfg.height;
  hipLaunchKernelGGL(k_depth_from_host, dim3((W + 255) / 256, H), dim3(256), 0, s, p->in_pending, W, H);
  p->in_pending.kind = 0;
}
// enqueued work reads the current input slot (a scan, a sweep, the tracker's pyramid): it may not be overwritten before that work is done
void input_slot_in_use(se_hip_pipeline* p) { if (p->cur_in >= 0) { p->in_state[p->cur_in] = 1; p->in_event[p->cur_in] = false; } }
// a raycast launch with gate sequence `seq` was enqueued: once it has started, everything enqueued before it on the main stream is done
void input_slots_followed(se_hip_pipeline* p, uint32_t seq) {
  for (int i = 0; i < se_hip_pipeline::kIn; ++i) if (p->in_state[i] == 1) { p->in_state[i] = 2; p->in_seq[i] = seq; }
}

// Makes the main stream see a scan that ran on the side stream: wait for it, then publish the
// occupancy bits of what it inserted (the scan left occ[] alone because the previous frame's raycast
// may still have been walking it) and apply OFusion's keys[0] quirk.
int join_scan(se_hip_pipeline* p, bool fold_into_sweep = false) {
  if (p->scan_pending) {
    p->scan_pending = false;
    HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_scan, 0));
    p->occ_commit_due = true;
    if (!p->occ_lists.lists) p->occ_lists = OccLists{p->map.newkeys, 1, (long long)p->map.cap_keys + 1};
    // keys[0] quirk: a row-sharded replica applies it in se_hip_alloc_commit, over every rank's list
    if (p->cfg.field_type == SE_HIP_FIELD_OFUSION && !p->sharded)
      if (int r = run_zero_chain(p, p->map.newkeys, 1, (long long)p->map.cap_keys + 1)) return r;
  }
  // the sweep kernel publishes the bits itself when it is the next launch (fold_into_sweep)
  if (p->occ_commit_due && !fold_into_sweep) {
    p->occ_commit_due = false;
    ScopedTimer t(p, SE_HIP_K_ALLOC_COMMIT);
    hipLaunchKernelGGL(k_occ_commit, dim3(64), dim3(SE_WG), 0, p->stream, p->map, p->occ_lists);
    p->occ_lists = OccLists{nullptr, 0, 0};
  }
  return SE_HIP_OK;
}

int check(se_hip_pipeline* p) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");
  int cur = -1;
  if (!(hipGetDevice(&cur) == hipSuccess && cur == p->device)) {
    hipError_t e = hipSetDevice(p->device);
    if (e != hipSuccess) return fail(SE_HIP_E_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  }
  if (p->has_pending && !p->in_frame) { ++p->flush_streak; return flush_pending_raycast(p); }   // whoever calls anything but se_hip_frame gets the deferred raycast first
  return SE_HIP_OK;
}

// The sweep kernel mirrors the device counters into pinned memory when it starts, so an overflow raised by a scan /
// commit is visible to the host by the next stage call without any synchronisation: the frame path reports it
// (a frame late at most) instead of carrying on with a truncated key list or a map that lost octants.
int check_overflow(se_hip_pipeline* p) {
  const uint32_t o = p->ctr_host[C_OVERFLOW];
  if (o == 3) return fail(SE_HIP_E_CAPACITY, "brick exchange segment overflow: the peers missed block updates (raise cap_bricks of se_hip_set_sweep_shard)");
  if (o) return fail(SE_HIP_E_CAPACITY, o == 2 ? "new-key list overflow: blocks were allocated locally but not reported to the peers (raise the exchange capacity)"
                                              : "block / node pool exhausted (raise max_blocks)");
  return SE_HIP_OK;
}

// the images the raycast of `frame` writes, and every consumer (tracking, rendering, the getters) reads from then on
void select_image_target(se_hip_pipeline* p, uint32_t frame) {
  if (p->ring) {
    const size_t n3 = (size_t)p->cfg.width * p->cfg.height * 3;
    p->vertex = p->ring + (size_t)(frame % (uint32_t)p->ring_slots) * 2 * n3;
    p->normal = p->vertex + n3;
  } else {
    p->vertex = p->vertex_own; p->normal = p->normal_own;
  }
}
// scope of an entry point that may run with a deferred raycast outstanding (check() then leaves it alone)
struct InFrame { se_hip_pipeline* p; bool was; explicit InFrame(se_hip_pipeline* q, bool on = true) : p(q), was(q->in_frame) { q->in_frame = on; } ~InFrame() { p->in_frame = was; } };

// A pose or intrinsics with a NaN or an infinity in it is refused (SE_HIP_E_INVALID) instead of integrated: the reference would fuse garbage, and the
// kernels' cheap conversions (hardware float -> int in the march, the sweep's pixel index) are only argued equal to the reference's for finite rays.
bool finite_pose(const float* pose_cm, const float* k) {
  bool ok = true;
  for (int i = 0; i < 16; ++i) ok = ok && std::isfinite(pose_cm[i]);
  for (int i = 0; i < 4; ++i) ok = ok && std::isfinite(k[i]);
  return ok && k[0] != 0.f && k[1] != 0.f;
}

bool stage_runs_integration(uint32_t frame, uint32_t rate) { return ((frame % rate) == 0) || (frame <= 3); }

// Octree::init (se_core/include/se/octree.hpp:425-437) on the device: empty index, root node, every brick and node value
// at initValue().  Enqueued on the main stream.
void reset_map_state(se_hip_pipeline* p) {
  DevMap& m = p->map;
  hipMemsetAsync(m.tab, 0, p->tab_entries * sizeof(uint32_t), p->stream);
  hipMemsetAsync(m.occ, 0, p->occ_words * sizeof(uint32_t), p->stream);
  hipMemsetAsync(m.lbits, 0, p->lbits_words * sizeof(uint32_t), p->stream);
  hipMemsetAsync(m.cbits, 0, 2 * p->cbits_words * sizeof(uint32_t), p->stream);   // (dilated bits + the cells' own "has a block" bits)
  if (m.fbits) hipMemsetAsync(m.fbits, 0, p->fbits_words * sizeof(uint32_t), p->stream);
  hipMemsetAsync(m.bpos, 0, p->cap_blocks * sizeof(uint32_t), p->stream);
  hipMemsetAsync(m.bactive, 0, p->slots, p->stream);
  hipMemsetAsync(m.npos, 0, p->cap_nodes * sizeof(uint32_t), p->stream);
  hipMemsetAsync(m.nlevel, 0, p->cap_nodes, p->stream);
  hipMemsetAsync(m.stats, 0, S_COUNT * sizeof(unsigned long long), p->stream);
  hipMemsetAsync(m.newkeys, 0, sizeof(unsigned long long), p->stream);
  if (p->newkeys_own) hipMemsetAsync(p->newkeys_own, 0, sizeof(unsigned long long), p->stream);
  if (p->newkeys_own2) hipMemsetAsync(p->newkeys_own2, 0, sizeof(unsigned long long), p->stream);
  p->own_clean[0] = p->newkeys_own != nullptr; p->own_clean[1] = p->newkeys_own2 != nullptr;
  // node 0 = root: level 0, side = size
  const uint32_t ctr0[C_COUNT] = {0u, 1u, 0u, 0u, 0u, 0u, 0u, 0u};
  hipMemcpyAsync(m.ctr, ctr0, sizeof ctr0, hipMemcpyHostToDevice, p->stream);
  std::memset(p->ctr_host, 0, C_COUNT * sizeof(uint32_t));
  p->ctr_host[C_NODES] = 1u;
  p->sorted_n = 0; p->sweeps_since_sort = 0;
  hipLaunchKernelGGL(k_fill_bricks, dim3(16384), dim3(256), 0, p->stream, m.vx, m.init_x, m.init_y, p->slots * 1024);
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, p->stream, m.nx, m.init_x, p->cap_nodes * 8);
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, p->stream, m.ny, m.init_y, p->cap_nodes * 8);
}

int grid_for(size_t n, int wg, int cap) { size_t g = (n + wg - 1) / wg; if (g < 1) g = 1; if (g > (size_t)cap) g = cap; return (int)g; }

}  // namespace

extern "C" {

const char* se_hip_last_error(void) { return g_err.c_str(); }

int se_hip_create(const se_hip_config* cfg, se_hip_pipeline** out) {
  if (!cfg || !out) return fail(SE_HIP_E_INVALID, "null argument");
  *out = nullptr;
  const int N = cfg->volume_resolution;
  if (cfg->width <= 0 || cfg->height <= 0) return fail(SE_HIP_E_INVALID, "bad image size");
  if (N < 64 || N > 4096 || (N & (N - 1))) return fail(SE_HIP_E_INVALID, "volume_resolution must be a power of two in [64, 4096]");
  if (!(cfg->volume_dimension > 0)) return fail(SE_HIP_E_INVALID, "volume_dimension must be positive");
  if (cfg->field_type != SE_HIP_FIELD_SDF && cfg->field_type != SE_HIP_FIELD_OFUSION) return fail(SE_HIP_E_INVALID, "unknown field type");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SE_HIP_E_NOGPU, "no HIP device visible (this library has no CPU path)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(SE_HIP_E_INVALID, "device ordinal out of range");
  HIP_TRY(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return fail(SE_HIP_E_NOGPU, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");

  se_hip_pipeline* p = new se_hip_pipeline();
  p->cfg = *cfg;
  p->device = cfg->device;
  p->row_begin = cfg->row_begin;
  p->row_end = (cfg->row_end > cfg->row_begin) ? cfg->row_end : cfg->height;
  if (p->row_begin < 0 || p->row_end > cfg->height) { delete p; return fail(SE_HIP_E_INVALID, "bad row range"); }
  if (const char* ev = std::getenv("SE_HIP_RAY_CACHE_LEVELS")) p->ray_cache_levels = std::atoi(ev);  // tuning knob
  if (const char* ev = std::getenv("SE_HIP_ICP_LOOKAHEAD")) p->icp_lookahead = std::max(0, std::atoi(ev));   // A/B knob (0: every ICP iteration enqueued up front)
  if (const char* ev = std::getenv("SE_HIP_INTEG_GRID")) p->integ_grid = std::atoi(ev);            // tuning knob
  if (const char* ev = std::getenv("SE_HIP_BEAM")) p->beam = std::max(0, std::min(2, std::atoi(ev)));   // A/B + test knob
  if (const char* ev = std::getenv("SE_HIP_OF_LEAP")) p->of_leap = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("SE_HIP_IEEE_SWEEP")) p->ieee_sweep = std::atoi(ev) != 0;       // A/B + test knob: the sweep instantiation with the compiler's divisions
  if (const char* ev = std::getenv("SE_HIP_PRIO")) p->prio_hint = std::atoi(ev) != 0;              // tuning knob
  if (const char* ev = std::getenv("SE_HIP_PRIO_SHARE")) {                                         // tuning knob
    int a = 0, b = 0, c = 0;
    if (std::sscanf(ev, "%d,%d,%d", &a, &b, &c) == 3 && a >= b && b >= c && c >= 0 && a <= 1000) { p->prio_permille[0] = a; p->prio_permille[1] = b; p->prio_permille[2] = c; }
  }
  p->max_level = ilog2(N);
  p->leaf_level = p->max_level - 3;
  DevMap& m = p->map;
  m.size = N; m.max_level = p->max_level; m.leaf_level = p->leaf_level; m.dim = cfg->volume_dimension;
  size_t off = 0;
  for (int l = 0; l < SE_MAX_LEVELS; ++l) m.off[l] = 0;
  for (int l = 1; l <= p->leaf_level; ++l) { m.off[l] = (uint32_t)off; off += (size_t)1 << (3 * l); }
  p->tab_entries = off;
  m.leaf_off = m.off[p->leaf_level];
  p->occ_words = occ_words_upto(p->leaf_level);
  const size_t cells = (size_t)1 << (3 * p->leaf_level);
  // Dense mode (default): one 4 KB brick slot per cell of the block grid, addressed by position --
  // 1 GiB at 512^3, 8 GiB at 1024^3, 64 GiB at 2048^3 of the 288 GB; bricks are pre-set to
  // initValue() so an unallocated voxel reads exactly what the reference's tree walk returns.
  // Pooled mode (max_blocks > 0, or a grid that does not fit): max_blocks bricks behind the index.
  size_t free_b = 0, total_b = 0;
  hipMemGetInfo(&free_b, &total_b);
  // A dense grid is the default while it costs <= SE_HIP_DENSE_MAX_GIB and a third of what is free.  r04-r05 drew the line at 16 GiB (2048^3 = 64 GiB for a
  // ~2 GB payload went pooled: "2.6 % / 3.0 % behind dense", profiles/r04h_pooled_vs_dense.log).  r06, measured again on the current kernels, same box
  // (profiles/r06h_dense2048_ab.log, r06a_dense_sdf2048_pmc_summary.md): at 1280x960 -> 2048^3 the dense raycast is 215 us beside the scan against 286 us
  // (163 / 203 us stand-alone), the scan 226 against 308 us, the sweep 738 against 702 us (bricks one 4 KB page each: 74 k address-translation misses per
  // launch against 2.5 k) -- 1 010 against 956 frames/s: a 288 GB part has the 64 GiB, so the line is 64 GiB now.  The pooled layout's default capacity
  // follows what a scan can allocate -- surfaces, not volume: four layers of blocks on each of the six faces of the volume's cube, 24 (N/8)^2 bricks
  // (1.5 GiB at 1024^3, 6 GiB at 2048^3), at least 65 536; a pool that runs out is reported (SE_HIP_E_CAPACITY), max_blocks raises it.
  size_t dense_max_gib = 64;
  if (const char* ev = std::getenv("SE_HIP_DENSE_MAX_GIB")) dense_max_gib = (size_t)std::max(0, std::atoi(ev));

  bool dense = cfg->max_blocks <= 0 && cells * 4096 <= free_b / 3 && cells * 4096 <= (dense_max_gib << 30);
  if (const char* ev = std::getenv("SE_HIP_DENSE")) dense = std::atoi(ev) != 0 && cells * 4096 <= free_b / 2;
  const size_t nb_side = (size_t)cfg->volume_resolution / 8;
  const size_t cap_default = std::max((size_t)1 << 16, ((24 * nb_side * nb_side + 4095) / 4096) * 4096);
  size_t cap = cfg->max_blocks > 0 ? (size_t)cfg->max_blocks : (dense ? cells : std::max((size_t)1 << 16, std::min(cap_default, free_b / 3 / 4096)));
  cap = std::min(cap, cells);
  m.dense = dense ? 1 : 0;
  p->sort_every = (dense && nb_side >= 128) ? 4 : 0;   // (sort_block_list)
  if (const char* ev = std::getenv("SE_HIP_SORT_BLOCKS")) p->sort_every = dense ? std::max(0, std::atoi(ev)) : 0;
  const size_t slots = dense ? cells : cap;   // voxel bricks / active flags
  p->slots = slots;
  size_t capn = std::min(off - cells + 1, cap / 2 + 4096);  // internal nodes (+ root)
  m.cap_blocks = (uint32_t)cap; m.cap_nodes = (uint32_t)capn;
  m.cap_keys = cap + capn;
  if (cfg->field_type == SE_HIP_FIELD_SDF) { m.init_x = 1.f; m.init_y = 0.f; m.empty_x = 1.f; }
  else { m.init_x = 0.f; m.init_y = 0.f; m.empty_x = 0.f; }

  auto bail = [&](hipError_t e, const char* what) { std::string msg = std::string(what) + ": " + hipGetErrorString(e); se_hip_destroy(p); return fail(SE_HIP_E_DEVICE, msg); };
#define ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc((void**)&(ptr), (bytes)); if (e_ != hipSuccess) return bail(e_, "hipMalloc " #ptr); } while (0)
  hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
  if (e != hipSuccess) return bail(e, "hipStreamCreate");
  p->own_stream = true;
  e = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking);
  if (e != hipSuccess) return bail(e, "hipStreamCreate");
  p->own_side = true;
  // these two only order streams of this device against each other: no system-scope fence (its cache write-back /
  // invalidate sits between the sweep and the raycast otherwise), no timing
  hipEventCreateWithFlags(&p->ev_sweep, hipEventDisableTiming | hipEventDisableSystemFence);
  hipEventCreateWithFlags(&p->ev_scan, hipEventDisableTiming | hipEventDisableSystemFence);
  p->sharded = (p->row_begin != 0 || p->row_end != cfg->height);
  // r03: pooled bricks overlap too (their raycast reads the index the scan writes: see se_block_entry for why that race is
  // benign); SE_HIP_POOLED_OVERLAP=0 restores the serial schedule for pooled maps
  p->overlap = (dense || !(std::getenv("SE_HIP_POOLED_OVERLAP") && std::atoi(std::getenv("SE_HIP_POOLED_OVERLAP")) == 0)) && !std::getenv("SE_HIP_NO_OVERLAP");
  p->host_gate = p->overlap && !p->sharded;
  if (const char* ev = std::getenv("SE_HIP_HOST_GATE")) p->host_gate = p->host_gate && std::atoi(ev) != 0;   // tuning knob
  if (p->host_gate) {
    e = hipHostMalloc((void**)&p->gate_host, 64);
    if (e != hipSuccess) return bail(e, "hipHostMalloc");
    std::memset(p->gate_host, 0, 64);
  }
  ALLOC(m.tab, p->tab_entries * sizeof(uint32_t));
  ALLOC(m.occ, p->occ_words * sizeof(uint32_t));
  p->lbits_words = (cells + 31) / 32;
  ALLOC(m.lbits, p->lbits_words * sizeof(uint32_t));
  m.clevel = std::min(p->leaf_level, 5);      // coarse cells of dim / 32 (15 cm at 4.8 m): see se_beam_start
  p->cbits_words = std::max<size_t>(1, ((size_t)1 << (3 * m.clevel)) / 32);
  ALLOC(m.cbits, 2 * p->cbits_words * sizeof(uint32_t));   // [dilated bits][undilated bits of the same grid: se_mark_coarse]
  m.fbits = nullptr;
  // the level-min(leaf, 6) grid, cells of dim / 64 (7.5 cm at 4.8 m): second stage of the beam start, OFusion leap
  if (se_flevel(m) > m.clevel) { p->fbits_words = ((size_t)1 << (3 * se_flevel(m))) / 32; ALLOC(m.fbits, p->fbits_words * sizeof(uint32_t)); }
  ALLOC(m.vx, slots * 1024 * sizeof(float));   // [512 x | 512 y] per brick
  m.ybyte = cfg->field_type == SE_HIP_FIELD_SDF ? 1 : 0;   // SDF weights are stored as bytes (se_device.h)
  ALLOC(m.bpos, cap * sizeof(uint32_t));
  ALLOC(m.bactive, (slots + 3) & ~(size_t)3);   // whole 32-bit words: se_set_active_once
  ALLOC(m.nx, capn * 8 * sizeof(float));
  ALLOC(m.ny, capn * 8 * sizeof(float));
  ALLOC(m.npos, capn * sizeof(uint32_t));
  ALLOC(m.nlevel, capn);
  ALLOC(m.ctr, C_COUNT * sizeof(uint32_t));
  ALLOC(m.stats, S_COUNT * sizeof(unsigned long long));
  ALLOC(m.newkeys, (m.cap_keys + 1) * sizeof(unsigned long long));
  p->newkeys_own = m.newkeys; p->cap_keys_own = m.cap_keys;
  ALLOC(p->newkeys_own2, (m.cap_keys + 1) * sizeof(unsigned long long));
  for (int i = 0; i < se_hip_pipeline::kIn; ++i) ALLOC(p->depth_ring[i], (size_t)cfg->width * cfg->height * sizeof(float));
  ALLOC(p->vertex_own, (size_t)cfg->width * cfg->height * 3 * sizeof(float));
  ALLOC(p->normal_own, (size_t)cfg->width * cfg->height * 3 * sizeof(float));
  p->vertex = p->vertex_own; p->normal = p->normal_own;
  ALLOC(p->chain, 4 * sizeof(unsigned long long));
  const size_t n_tiles = ((size_t)(cfg->width + SE_TILE_W - 1) / SE_TILE_W) * ((size_t)(cfg->height + SE_TILE_H - 1) / SE_TILE_H);
  ALLOC(p->tile_cost, (n_tiles + 4096) * sizeof(unsigned short));   // (+ padding: se_ray_schedule reads it in 16-byte pieces)
  ALLOC(p->prio_thr, 4 * sizeof(int));
  hipMemsetAsync(p->tile_cost, 0, (n_tiles + 4096) * sizeof(unsigned short), p->stream);
  {
    hipDeviceProp_t prop{};
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) p->n_cus = prop.multiProcessorCount;
    if (const char* ev = std::getenv("SE_HIP_RAY_DEAL")) p->n_cus = std::max(1, std::atoi(ev));   // tuning knob: width of the snake deal (1 = image order)
    const size_t n_pairs = (n_tiles + 1) / 2;
    ALLOC(p->ray_order, n_pairs * sizeof(uint32_t));
    std::vector<uint32_t> ident(n_pairs);
    for (size_t i = 0; i < n_pairs; ++i) ident[i] = (uint32_t)i;
    hipMemcpyAsync(p->ray_order, ident.data(), n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, p->stream);
    hipStreamSynchronize(p->stream);   // (the host vector goes out of scope)
  }
  { const int off[4] = {256, 256, 256, 0}; hipMemcpyAsync(p->prio_thr, off, sizeof off, hipMemcpyHostToDevice, p->stream); }
  p->depth = p->depth_ring[0]; p->cur_in = 0;
  e = hipHostMalloc((void**)&p->ctr_host, C_COUNT * sizeof(uint32_t));
  if (e != hipSuccess) return bail(e, "hipHostMalloc");
  std::memset(p->ctr_host, 0, C_COUNT * sizeof(uint32_t));

  p->cap_blocks = cap; p->cap_nodes = capn;
  reset_map_state(p);
  for (int i = 0; i < se_hip_pipeline::kIn; ++i) hipMemsetAsync(p->depth_ring[i], 0, (size_t)cfg->width * cfg->height * sizeof(float), p->stream);
  hipMemsetAsync(p->vertex, 0, (size_t)cfg->width * cfg->height * 3 * sizeof(float), p->stream);
  hipMemsetAsync(p->normal, 0, (size_t)cfg->width * cfg->height * 3 * sizeof(float), p->stream);
  if (cfg->field_type == SE_HIP_FIELD_OFUSION) {
    std::vector<float> lut, lo;
    make_bspline(lut);
    make_logodds(lut, lo);
    ALLOC(p->bspline, lut.size() * sizeof(float));
    ALLOC(p->logodds, lo.size() * sizeof(float));
    hipMemcpy(p->bspline, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice);
    hipMemcpy(p->logodds, lo.data(), lo.size() * sizeof(float), hipMemcpyHostToDevice);
  }
#undef ALLOC
  e = hipStreamSynchronize(p->stream);
  if (e != hipSuccess) return bail(e, "initialisation");
  *out = p;
  return SE_HIP_OK;
}

int se_hip_destroy(se_hip_pipeline* p) {
  if (!p) return SE_HIP_OK;
  p->has_pending = false;   // (a deferred raycast nobody will look at)
  hipSetDevice(p->device);
  if (p->side) hipStreamSynchronize(p->side);
  if (p->stream) hipStreamSynchronize(p->stream);
  for (auto& t : p->pending) { hipEventDestroy(t.start); hipEventDestroy(t.stop); }
  for (auto& ev : p->event_pool) hipEventDestroy(ev);
  DevMap& m = p->map;
  void* ptrs[] = {m.occ, m.lbits, m.cbits, m.fbits, m.tab, m.vx, m.bpos, m.bactive, m.nx, m.ny, m.npos, m.nlevel, m.ctr, m.stats, p->newkeys_own, p->newkeys_own2,
                  p->depth_ring[0], p->depth_ring[1], p->depth_ring[2], p->vertex_own, p->normal_own, p->bspline, p->logodds, p->chain, p->tile_cost, p->prio_thr, p->ray_order};
  for (void* q : ptrs) if (q) hipFree(q);
  for (auto* q : p->pyr_depth) if (q) hipFree(q);
  for (auto* q : p->pyr_vertex) if (q) hipFree(q);
  for (auto* q : p->pyr_normal) if (q) hipFree(q);
  if (p->track) hipFree(p->track);
  if (p->rgbw) hipFree(p->rgbw);
  if (p->reduce_partial) hipFree(p->reduce_partial);
  if (p->icp) hipFree(p->icp);
  if (p->icp_host) hipHostFree(p->icp_host);
  if (p->ctr_host) hipHostFree(p->ctr_host);
  if (p->bpos_sorted) hipFree(p->bpos_sorted);
  if (p->sort_tmp) hipFree(p->sort_tmp);
  if (p->gate_host) hipHostFree(p->gate_host);
  if (p->mesh_ctr) hipFree(p->mesh_ctr);
  for (int i = 0; i < se_hip_pipeline::kIn; ++i) { if (p->in_host[i]) hipHostFree(p->in_host[i]); if (p->in_done[i]) hipEventDestroy(p->in_done[i]); }
  if (p->own_side && p->side) hipStreamDestroy(p->side);
  if (p->ev_sweep) hipEventDestroy(p->ev_sweep);
  if (p->ev_scan) hipEventDestroy(p->ev_scan);
  if (p->own_stream && p->stream) hipStreamDestroy(p->stream);
  delete p;
  return SE_HIP_OK;
}

int se_hip_sync(se_hip_pipeline* p) {
  if (int r = check(p)) return r;
  // (polling hipStreamQuery before blocking was tried for the closed loop, one sync per frame: 10.3 k vs 10.8 k frames/s -- worse)
  if (p->side) HIP_TRY(hipStreamSynchronize(p->side));
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->gate_armed = false;   // every sweep enqueued so far is done
  return check_overflow(p);
}

int se_hip_clear_overflow(se_hip_pipeline* p) {
  if (int r = check(p)) return r;
  if (p->side) HIP_TRY(hipStreamSynchronize(p->side));
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->gate_armed = false;
  uint32_t dev = 0;
  HIP_TRY(hipMemcpy(&dev, p->map.ctr + C_OVERFLOW, sizeof dev, hipMemcpyDeviceToHost));
  const uint32_t pending = std::max(dev, p->ctr_host[C_OVERFLOW]);
  HIP_TRY(hipMemsetAsync(p->map.ctr + C_OVERFLOW, 0, sizeof(uint32_t), p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->ctr_host[C_OVERFLOW] = 0u;
  return (int)pending;
}

int se_hip_set_stream(se_hip_pipeline* p, void* hip_stream) {
  if (int r = check(p)) return r;
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->gate_armed = false;
  drain_timings(p);
  if (p->own_stream && p->stream) hipStreamDestroy(p->stream);
  p->stream = (hipStream_t)hip_stream;
  p->own_stream = false;
  return SE_HIP_OK;
}

int se_hip_scan_overlaps(se_hip_pipeline* p) {
  if (int r = check(p)) return r;
  return p->overlap ? 1 : 0;
}

int se_hip_set_scan_stream(se_hip_pipeline* p, void* hip_stream) {
  if (int r = check(p)) return r;
  if (p->side) HIP_TRY(hipStreamSynchronize(p->side));
  HIP_TRY(hipStreamSynchronize(p->stream));
  drain_timings(p);
  if (p->own_side && p->side) hipStreamDestroy(p->side);
  p->side = nullptr; p->own_side = false;
  if (hip_stream) { p->side = (hipStream_t)hip_stream; return SE_HIP_OK; }
  HIP_TRY(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
  p->own_side = true;
  return SE_HIP_OK;
}

// Host image -> the next slot of the pinned input ring (see se_hip_pipeline::in_host); nothing is enqueued.
static int stage_input(se_hip_pipeline* p, const void* host, size_t bytes, int kind, int in_w, int ratio) {
  constexpr int R = se_hip_pipeline::kIn;
  if (p->in_cap < bytes) {
    // (first call, or a larger input image: the slots' pending readers first)
    if (p->side) HIP_TRY(hipStreamSynchronize(p->side));
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int i = 0; i < R; ++i) { if (p->in_host[i]) hipHostFree(p->in_host[i]); p->in_host[i] = nullptr; p->in_state[i] = 0; p->in_event[i] = false; }
    p->in_cap = 0; p->in_pending.kind = 0;
    for (int i = 0; i < R; ++i) HIP_TRY(hipHostMalloc(&p->in_host[i], bytes));
    p->in_cap = bytes;
  }
  const int i = p->in_next;
  p->in_next = (i + 1) % R;
  // the slot's last readers (three uploads ago) must be done: normally long true
  if (p->in_state[i] == 2 && p->host_gate) {
    volatile uint32_t* w = p->gate_host;
    const uint32_t target = p->in_seq[i];
    if (!spin_until([&] { return (int32_t)(*w - target) >= 0; }, 20000)) { if (p->side) hipStreamSynchronize(p->side); hipStreamSynchronize(p->stream); }
  } else if (p->in_event[i]) {
    HIP_TRY(hipEventSynchronize(p->in_done[i]));
  } else if (p->in_state[i] != 0) {
    // read by enqueued work that nothing has followed yet (integration-only callers, three uploads without a raycast)
    if (p->side) HIP_TRY(hipStreamSynchronize(p->side));
    HIP_TRY(hipStreamSynchronize(p->stream));
  }
  p->in_state[i] = 0; p->in_event[i] = false;
  const void* src = p->in_host[i];
  if (p->pinned_input) {
    // caller-pinned image (se_hip_set_pinned_input): read where it lies; the slot only lends its device image and its place in the ring
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, host) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) src = at.devicePointer;
    else (void)hipGetLastError();   // (pageable memory: an error code, not an error)
  }
  if (src == p->in_host[i]) std::memcpy(p->in_host[i], host, bytes);   // (r06, measured: non-temporal stores here took 21 us instead of 16 us for the 614 KB image and moved nothing on the device side)
  p->in_pending = DepthSrc{src, p->depth_ring[i], kind, in_w, ratio};
  p->depth = p->depth_ring[i];
  p->cur_in = i;
  return SE_HIP_OK;
}

int se_hip_upload_depth(se_hip_pipeline* p, const float* host_depth_m) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");
  InFrame guard(p);   // (a deferred raycast reads the map, not the depth image: the next frame's input may arrive before it is launched)
  if (int r = check(p)) return r;
  if (!host_depth_m) return fail(SE_HIP_E_INVALID, "null depth");
  return stage_input(p, host_depth_m, (size_t)p->cfg.width * p->cfg.height * sizeof(float), 2, p->cfg.width, 1);
}

int se_hip_upload_depth_mm(se_hip_pipeline* p, const uint16_t* host_mm, int32_t in_w, int32_t in_h) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");
  InFrame guard(p);
  if (int r = check(p)) return r;
  if (!host_mm) return fail(SE_HIP_E_INVALID, "null depth");
  const int W = p->cfg.width, H = p->cfg.height;
  // the reference prints "Invalid ratio." and exits (preprocessing.cpp:165-176); here it is an error code
  if (in_w < W || in_h < H || in_w % W != 0 || in_h % H != 0 || in_w / W != in_h / H) return fail(SE_HIP_E_INVALID, "Invalid ratio.");
  return stage_input(p, host_mm, (size_t)in_w * in_h * sizeof(unsigned short), 1, in_w, in_w / W);   // mm2metersKernel happens where the image is first read
}

int se_hip_set_pinned_input(se_hip_pipeline* p, int32_t on) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");   // (a flag only: does not launch a deferred raycast)
  p->pinned_input = on != 0;
  return SE_HIP_OK;
}
void* se_hip_host_alloc(size_t bytes) {
  void* h = nullptr;
  if (bytes == 0 || hipHostMalloc(&h, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return h;
}
void se_hip_host_free(void* host) { if (host) (void)hipHostFree(host); }

int se_hip_set_depth_device(se_hip_pipeline* p, const float* device_depth_m) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");
  InFrame guard(p);
  if (int r = check(p)) return r;
  if (device_depth_m) { p->depth = device_depth_m; p->cur_in = -1; p->in_pending.kind = 0; }   // (an upload not yet consumed is dropped: the caller replaced the input)
  else if (p->cur_in < 0) { p->depth = p->depth_ring[0]; p->cur_in = 0; }
  return SE_HIP_OK;
}

// The deferred raycast of the previous se_hip_frame call + this frame's allocation scan as one launch on the main stream (k_raycast_scan).
int launch_raycast_scan(se_hip_pipeline* p, const DevMap& ms, const AllocArgs& sa, int scan_wgs, const DepthSrc& ds) {
  const DevMap& m = p->map;
  // (an overflow reported here must not drop the held-back raycast -- se_hip_frame(f) has already said "raycast ran": it stays pending and is
  // launched by the first call after se_hip_clear_overflow)
  if (int r = check_overflow(p)) return r;
  p->has_pending = false;
  p->flush_streak = 0;
  std::memcpy(p->raycast_pose, p->pend_pose, sizeof p->raycast_pose);   // raycast_pose_ = pose_ (DenseSLAMSystem.cpp:196)
  p->images_complete = !p->sharded;
  select_image_target(p, p->pend_frame);
  p->n_launch[SE_HIP_K_ALLOC_SCAN]++; p->n_launch[SE_HIP_K_COUNT]++;   // (the raycast half is counted by the timer scope below)
  RayLaunchArgs L = make_ray_args(p, p->pend_pose, p->pend_k, p->pend_mu);
  if (p->host_gate) { L.a.gate = p->gate_host; L.a.gate_seq = ++p->ray_seq; if (p->gate_armed) p->gate_followed = true; input_slots_followed(p, L.a.gate_seq); }
  const RayArgs& a = L.a;
  const int ray_wgs = (int)L.grid.x;
  const size_t smem = std::max(L.smem, (size_t)SE_SCAN_SLOTS * SE_WG_SCAN * sizeof(uint32_t));
  const dim3 grid((unsigned)(ray_wgs + scan_wgs)), block(SE_WG_RAY);
  const bool sdf = p->cfg.field_type == SE_HIP_FIELD_SDF;
  const size_t nb = (size_t)(m.size >> 3);
  // (a dense grid of > 4 GiB with every level staged -- only with SE_HIP_RAY_CACHE_LEVELS raised -- takes the generic instantiation)
  const bool shallow = !a.has_deep && !(m.dense && nb * nb * nb * (size_t)SE_BRICK_STRIDE * sizeof(float) > ((size_t)4 << 30));
  {
    ScopedTimer t(p, SE_HIP_K_RAYCAST);
#define SE_RS(OF, DN, SH, O3) hipLaunchKernelGGL((k_raycast_scan<OF, DN, SH, O3>), grid, block, smem, p->stream, m, a, p->vertex, p->normal, ray_wgs, ms, p->depth, sa, ds)
    if (sdf) {
      if (m.dense) { if (shallow) SE_RS(false, true, true, true); else SE_RS(false, true, false, false); }
      else { if (shallow) SE_RS(false, false, true, false); else SE_RS(false, false, false, false); }
    } else {
      if (m.dense) { if (shallow) SE_RS(true, true, true, true); else SE_RS(true, true, false, false); }
      else { if (shallow) SE_RS(true, false, true, false); else SE_RS(true, false, false, false); }
    }
#undef SE_RS
  }
  HIP_TRY(hipGetLastError());
  return SE_HIP_OK;
}

// ---------------------------------------------------------------------------------- integrate
static bool frame_can_fuse(se_hip_pipeline* p);
int se_hip_alloc_scan(se_hip_pipeline* p, const float pose_cm[16], const float k[4], uint32_t rate, float mu, uint32_t frame) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");
  // the stage call of a streaming caller (multi_gpu.ShardedPipeline: scan -> exchange -> sweep -> deferred raycast): a raycast that is waiting rides in this
  // scan's launch exactly as inside se_hip_integrate
  const bool ride = p->has_pending && !p->in_frame && rate != 0 && stage_runs_integration(frame, rate) && frame_can_fuse(p);
  InFrame guard(p, ride || p->in_frame);
  if (int r = check(p)) return r;
  if (!pose_cm || !k || rate == 0) return fail(SE_HIP_E_INVALID, "bad argument");
  if (!finite_pose(pose_cm, k)) return fail(SE_HIP_E_INVALID, "non-finite pose or intrinsics");
  if (!stage_runs_integration(frame, rate)) return 0;  // DenseSLAMSystem.cpp:209
  const DevMap& m = p->map;
  const M4 pose = from_colmajor(pose_cm);
  const bool sdf = p->cfg.field_type == SE_HIP_FIELD_SDF;
  AllocArgs a{};
  const float voxelsize = m.dim / (float)m.size;                  // DenseSLAMSystem.cpp:211
  const M4 kPose = mul(pose, inverse_camera_matrix(k));           // alloc_impl.hpp:63-64 (K.inverse(): closed form)
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) a.kpose[i * 4 + j] = kPose.m[i][j];
  a.cam[0] = pose.m[0][3]; a.cam[1] = pose.m[1][3]; a.cam[2] = pose.m[2][3];
  a.band = sdf ? 2 * mu : 6 * mu;                                 // DenseSLAMSystem.cpp:223,228
  a.voxel = voxelsize;
  a.inv_voxel = sdf ? 1 / voxelsize : 1.f / voxelsize;
  a.num_steps = (int)std::ceil(a.band * a.inv_voxel);
  a.W = p->cfg.width; a.H = p->cfg.height; a.row_begin = p->row_begin; a.row_end = p->row_end;
  a.sharded = (p->row_begin != 0 || p->row_end != p->cfg.height) ? 1 : 0;
  // step_to_depth (bfusion/alloc_impl.hpp:48-51) for the three step sizes of compute_stepsize
  auto s2d = [&](float step) { return (int)(floorf(log2f(voxelsize / step)) + m.max_level); };
  a.depth_fine = s2d(voxelsize); a.depth_mid = s2d(10.f * voxelsize); a.depth_coarse = s2d(30.f * voxelsize);
  // Overlap mode: the scan runs on the side stream as soon as the previous sweep is done, i.e.
  // concurrently with the previous frame's raycast, and defers its occ[] updates (join_scan).
  // ... unless there is nothing to overlap with: a caller that synchronises every frame (a SLAM loop whose next pose
  // depends on this frame's raycast) finds the main stream idle here, and the scan then goes straight onto it -- no
  // cross-stream events between scan and sweep (closed loop 110 -> 9x us per frame).  Row-sharded replicas keep the scan
  // stream: their all-gather is ordered on it.
  // ... and so do handles whose key list is a caller buffer or goes into an exchange (se_hip_set_new_keys_buffer,
  // se_hip_set_exchange): whoever consumes that list is told "scan stream" by se_hip_scan_overlaps().
  const bool caller_list = p->map.newkeys != p->newkeys_own && p->map.newkeys != p->newkeys_own2;
  // r04: a deferred raycast of the previous se_hip_frame call is waiting -> this scan rides in its launch, on the main stream (k_raycast_scan)
  const bool fuse_now = p->has_pending && p->in_frame && p->overlap && !p->stats && p->shard_world <= 1;
  // (se_hip_frame_tracked: the main stream still holds the tail of the ICP's last launch, but there is no previous raycast to hide the scan under)
  const bool chain = p->scan_on_main_once && !p->sharded && !caller_list && p->xgather == nullptr;
  p->scan_on_main_once = false;
  const bool ov = !fuse_now && !chain && p->overlap && (p->sharded || caller_list || p->xgather != nullptr || hipStreamQuery(p->stream) == hipErrorNotReady);
  p->scan_on_side = ov;
  hipStream_t s = ov ? p->side : p->stream;
  DevMap ms = m;
  ms.defer_occ = (ov || fuse_now) ? 1 : 0;
  ms.defer_mark = 1;   // the beam-start bitmaps of the new blocks are set from the key list by the sweep in front of the next raycast (se_occ_commit), whatever the schedule
  if (ov) { if (p->host_gate) wait_last_sweep(p); else HIP_TRY(hipStreamWaitEvent(p->side, p->ev_sweep, 0)); }
  else if (p->overlap) { if (int r = join_scan(p)) return r; }   // a depth upload that went to the scan stream is joined here
  const bool own_list = p->map.newkeys == p->newkeys_own || p->map.newkeys == p->newkeys_own2;
  if (own_list) {
    // the own lists alternate; the previous frame's sweep kernel has already cleared this one's count word (no fill
    // kernel in front of the scan: 5 us on the chain sweep -> scan -> next sweep that the scan stream adds to a frame)
    const int w = p->own_next;
    p->map.newkeys = w ? p->newkeys_own2 : p->newkeys_own;
    ms.newkeys = p->map.newkeys;
    if (!p->own_clean[w]) HIP_TRY(hipMemsetAsync(p->map.newkeys, 0, sizeof(unsigned long long), s));
    p->own_clean[w] = false;
  } else {
    HIP_TRY(hipMemsetAsync(m.newkeys, 0, sizeof(unsigned long long), s));   // caller's buffer (multi-GPU send buffer)
  }
  const int npix = (p->row_end - p->row_begin) * p->cfg.width;
  const dim3 grid((npix + SE_WG_SCAN - 1) / SE_WG_SCAN), block(SE_WG_SCAN);
  // an input image still in host memory: this scan visits every pixel once and materialises float_depth_ on its way (DepthSrc); a row-sharded replica's
  // scan sees only its own rows, the sweep behind it needs all of them
  if (a.sharded) materialise_depth(p, s);
  const DepthSrc ds = p->in_pending;
  p->in_pending.kind = 0;
  // SDF: a wave = an 8x8 pixel tile, or 64 pixels of a row when it reads the image from host memory (se_scan_sdf_wg)
  const int sdf_tiles = ds.kind ? ((p->cfg.width + 63) / 64) * (p->row_end - p->row_begin) : ((p->cfg.width + 7) / 8) * ((p->row_end - p->row_begin + 7) / 8);
  if (!sdf) {
    // the three stages' levels (fetch_octant stops at the leaves) and their offsets in the index pyramid (tiled kernel)
    const int dep[3] = {a.depth_fine, a.depth_mid, a.depth_coarse};
    for (int i = 0; i < 3; ++i) { a.of_lvl[i] = std::min(dep[i], m.leaf_level); a.of_off[i] = a.of_lvl[i] >= 1 ? m.off[a.of_lvl[i]] : 0u; }
  }
  if (fuse_now) {
    const int scan_wgs = sdf ? (sdf_tiles + SE_WG_SCAN / 64 - 1) / (SE_WG_SCAN / 64) : (int)grid.x;
    if (int r = launch_raycast_scan(p, ms, a, scan_wgs, ds)) return r;
    input_slot_in_use(p);
    p->occ_commit_due = true;
    if (!p->occ_lists.lists) p->occ_lists = OccLists{p->map.newkeys, 1, (long long)p->map.cap_keys + 1};
    if (!sdf && !a.sharded) { if (int r = run_zero_chain(p, p->map.newkeys, 1, (long long)p->map.cap_keys + 1)) return r; }   // (sharded: se_hip_alloc_commit, over every rank's list)
    HIP_TRY(hipGetLastError());
    return 1;
  }
  {
    ScopedTimer t(p, SE_HIP_K_ALLOC_SCAN, s);
    if (sdf) {
      const dim3 sgrid((sdf_tiles + SE_WG_SCAN / 64 - 1) / (SE_WG_SCAN / 64));
      if (m.dense) {
        if (p->stats) hipLaunchKernelGGL((k_alloc_scan_sdf<true, true>), sgrid, block, 0, s, ms, p->depth, a, ds);
        else hipLaunchKernelGGL((k_alloc_scan_sdf<false, true>), sgrid, block, 0, s, ms, p->depth, a, ds);
      } else {
        if (p->stats) hipLaunchKernelGGL((k_alloc_scan_sdf<true, false>), sgrid, block, 0, s, ms, p->depth, a, ds);
        else hipLaunchKernelGGL((k_alloc_scan_sdf<false, false>), sgrid, block, 0, s, ms, p->depth, a, ds);
      }
    } else {
      if (p->stats) hipLaunchKernelGGL(k_alloc_scan_ofusion<true>, grid, block, 0, s, ms, p->depth, a, ds);
      else hipLaunchKernelGGL(k_alloc_scan_ofusion<false>, grid, block, 0, s, ms, p->depth, a, ds);
    }
  }
  input_slot_in_use(p);
  if (ov) {
    HIP_TRY(hipEventRecord(p->ev_scan, p->side));
    p->scan_pending = true;
    HIP_TRY(hipGetLastError());
    return 1;
  }
  // (serial schedule: occupancy bits in place, the new blocks' beam-start marks with the commit pass of the sweep / k_occ_commit)
  p->occ_commit_due = true;
  if (!p->occ_lists.lists) p->occ_lists = OccLists{p->map.newkeys, 1, (long long)p->map.cap_keys + 1};
  // keys[0] quirk of unique_multiscale (see k_zero_chain): needs the frame's complete key list, so a
  // row-sharded replica defers it to se_hip_alloc_commit (which sees every rank's list)
  if (!sdf && !a.sharded) {
    if (int r = run_zero_chain(p, m.newkeys, 1, (long long)m.cap_keys + 1)) return r;
  }
  HIP_TRY(hipGetLastError());
  return 1;
}

int se_hip_new_keys_device(se_hip_pipeline* p, uint64_t** device_list, int64_t* capacity_words) {
  if (int r = check(p)) return r;
  if (int r = join_scan(p)) return r;
  if (device_list) *device_list = (uint64_t*)p->map.newkeys;
  if (capacity_words) *capacity_words = (int64_t)p->map.cap_keys + 1;
  return SE_HIP_OK;
}

int se_hip_set_new_keys_buffer(se_hip_pipeline* p, uint64_t* device_list, int64_t capacity_words) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");
  InFrame guard(p);   // (a deferred raycast does not care where the next scan writes its list)
  if (int r = check(p)) return r;
  if (device_list && capacity_words < 2) return fail(SE_HIP_E_INVALID, "key buffer too small");
  p->map.newkeys = device_list ? (unsigned long long*)device_list : p->newkeys_own;
  p->map.cap_keys = device_list ? (unsigned long long)(capacity_words - 1) : p->cap_keys_own;
  return SE_HIP_OK;
}

int se_hip_alloc_commit(se_hip_pipeline* p, const uint64_t* device_lists, int32_t nlists, int64_t stride_words) {
  if (int r = check(p)) return r;
  if (!device_lists || nlists <= 0 || stride_words < 1) return fail(SE_HIP_E_INVALID, "bad argument");
  const unsigned long long* lists = (const unsigned long long*)device_lists;
  if (p->overlap && p->side && p->scan_on_side) {
    // The gathered lists were produced on the scan stream (scan kernel, then the caller's all-gather ordered
    // behind it).  The insertion of the peers' keys stays on that stream -- beside the previous frame's raycast,
    // off the sweep -> raycast critical path -- and, like the scan, leaves occ[] to the sweep kernel, which
    // publishes the bits of every gathered list (the own list is one of them).
    DevMap md = p->map;
    md.defer_occ = 1; md.defer_mark = 1;
    {
      ScopedTimer t(p, SE_HIP_K_ALLOC_COMMIT, p->side);
      hipLaunchKernelGGL(k_alloc_commit, dim3(64, nlists), dim3(SE_WG), 0, p->side, md, lists, nlists, (long long)stride_words);
    }
    HIP_TRY(hipEventRecord(p->ev_scan, p->side));
    p->scan_pending = true;
    p->occ_lists = OccLists{lists, nlists, (long long)stride_words};
  } else if (p->occ_commit_due && !p->scan_pending) {
    // the scan rode in the previous raycast's launch on the main stream (k_raycast_scan) and left the bits of its own keys to the sweep, which is still
    // to come: no k_occ_commit launch between all-gather and sweep.  The peers' keys get their bits here (that raycast is over: stream order).
    ScopedTimer t(p, SE_HIP_K_ALLOC_COMMIT);
    hipLaunchKernelGGL(k_alloc_commit, dim3(64, nlists), dim3(SE_WG), 0, p->stream, p->map, lists, nlists, (long long)stride_words);
  } else {
    if (int r = join_scan(p)) return r;
    ScopedTimer t(p, SE_HIP_K_ALLOC_COMMIT);
    hipLaunchKernelGGL(k_alloc_commit, dim3(64, nlists), dim3(SE_WG), 0, p->stream, p->map, lists, nlists, (long long)stride_words);
  }
  if (p->cfg.field_type == SE_HIP_FIELD_OFUSION) {
    if (int r = join_scan(p, true)) return r;   // the keys[0] chain runs on the main stream, behind scan + commit
    if (int r = run_zero_chain(p, lists, nlists, (long long)stride_words)) return r;
  }
  HIP_TRY(hipGetLastError());
  return SE_HIP_OK;
}

int se_hip_set_exchange(se_hip_pipeline* p, void* nccl_comm, void* nccl_all_gather, int32_t world) {
  if (int r = check(p)) return r;
  if ((nccl_comm == nullptr) != (nccl_all_gather == nullptr) || (nccl_comm && world < 1)) return fail(SE_HIP_E_INVALID, "bad argument");
  p->xcomm = nccl_comm;
  p->xgather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))nccl_all_gather;
  p->xworld = nccl_comm ? world : 0;
  return SE_HIP_OK;
}

int se_hip_alloc_exchange(se_hip_pipeline* p, uint64_t* recv_device, int64_t words) {
  if (int r = check(p)) return r;
  if (!p->xgather) return fail(SE_HIP_E_INVALID, "no exchange set (se_hip_set_exchange)");
  if (!recv_device || words < 2 || (unsigned long long)words > p->map.cap_keys + 1) return fail(SE_HIP_E_INVALID, "bad argument");
  // on the stream the scan ran on: behind the scan, beside the previous frame's raycast
  hipStream_t s = (p->overlap && p->side && p->scan_on_side) ? p->side : p->stream;
  const int rc = p->xgather(p->map.newkeys, recv_device, (size_t)words, /* ncclInt64 */ 4, p->xcomm, s);
  if (rc != 0) return fail(SE_HIP_E_DEVICE, "ncclAllGather failed with code " + std::to_string(rc));
  return se_hip_alloc_commit(p, recv_device, p->xworld, words);
}

size_t se_hip_sweep_shard_bytes(size_t cap_bricks) { return SE_SHARD_SUB * 8 + cap_bricks * 4 + cap_bricks * 4096; }

int se_hip_set_sweep_shard(se_hip_pipeline* p, int32_t rank, int32_t world, void* send_device, size_t cap_bricks) {
  if (int r = check(p)) return r;
  if (world <= 1) { p->shard_world = 0; p->shard_send = nullptr; p->shard_cap = 0; return SE_HIP_OK; }
  if (rank < 0 || rank >= world || !send_device || ((uintptr_t)send_device & 15u) || cap_bricks == 0 || (cap_bricks % SE_SHARD_SUB) || cap_bricks > ((size_t)1 << 30))
    return fail(SE_HIP_E_INVALID, "bad argument (send segment 16-byte aligned, cap_bricks a positive multiple of 64)");
  p->shard_world = world; p->shard_rank = rank; p->shard_send = (unsigned char*)send_device; p->shard_cap = cap_bricks;
  return SE_HIP_OK;
}

int se_hip_apply_bricks(se_hip_pipeline* p, const void* recv_device, int32_t world) {
  if (int r = check(p)) return r;
  if (p->shard_world <= 1 || world != p->shard_world || !recv_device) return fail(SE_HIP_E_INVALID, "no sharded sweep set (se_hip_set_sweep_shard) or bad argument");
  {
    ScopedTimer t(p, SE_HIP_K_APPLY_BRICKS);
    const size_t est = (size_t)p->ctr_host[C_BLOCKS] / (size_t)world + 1024;   // records per segment, roughly
    const size_t wgs = std::min<size_t>((est + 3) / 4, 16384);
    const dim3 grid((unsigned)(((wgs + 15) / 16) * 16), (unsigned)world);   // 4 waves per workgroup: a multiple of SE_SHARD_SUB waves
    hipLaunchKernelGGL(k_apply_bricks, grid, dim3(SE_WG), 0, p->stream, p->map, (const unsigned char*)recv_device,
                       se_hip_sweep_shard_bytes(p->shard_cap), (uint32_t)p->shard_cap, p->shard_rank);
  }
  // the next frame's scan reads the active flags this kernel wrote: it is what that scan has to wait for
  if (!p->host_gate) hipEventRecord(p->ev_sweep, p->stream);
  HIP_TRY(hipGetLastError());
  return SE_HIP_OK;
}

int se_hip_brick_exchange(se_hip_pipeline* p, void* recv_device) {
  if (int r = check(p)) return r;
  if (!p->xgather) return fail(SE_HIP_E_INVALID, "no exchange set (se_hip_set_exchange)");
  if (p->shard_world <= 1 || p->xworld != p->shard_world || !recv_device) return fail(SE_HIP_E_INVALID, "no sharded sweep set (se_hip_set_sweep_shard) or bad argument");
  const int rc = p->xgather(p->shard_send, recv_device, se_hip_sweep_shard_bytes(p->shard_cap), /* ncclUint8 */ 1, p->xcomm, p->stream);
  if (rc != 0) return fail(SE_HIP_E_DEVICE, "ncclAllGather failed with code " + std::to_string(rc));
  return se_hip_apply_bricks(p, recv_device, p->xworld);
}

// Dense brick grid: the sweep visits the bricks in the order of the block list, one wave per entry, and the list is in allocation order -- scattered over a
// grid of 8 GiB at 1024^3 (one 4 KB brick per page-sized stride), where the pooled layout, whose bricks ARE in list order, sweeps 10 % faster.  The list is
// only a set there (a block's slot is its grid position, nothing refers to a list index), so it may be put in address order: a radix sort of the packed
// positions (x | y << 10 | z << 20: the grid's own order) of the entries the host already knows about, in stream order in front of a sweep -- at most every
// `sort_every` sweeps, and only once the list is half again as long as its sorted part (a map that keeps growing is sorted O(log) times; ~45 us each at
// 70 k blocks).  Entries appended since stay behind the sorted part until then.  Measured (profiles/r06o_sort_blocks_ab.log, room stream): sweep 108.3 ->
// 101.6 us at 1024^3, 749 -> 709 us at 2048^3 (+2-3.5 % frames/s), nothing at 512^3 (1 GiB grid) and nothing on the stress stream, whose camera pans:
// default for dense grids of >= 1024^3 (SE_HIP_SORT_BLOCKS=0 / n: off / at most every n sweeps).  Results cannot depend on it (same SHA-1 column).
static int sort_block_list(se_hip_pipeline* p) {
  if (p->sort_every <= 0 || !p->map.dense) return SE_HIP_OK;
  const uint32_t n0 = std::min<uint32_t>(p->ctr_host[C_BLOCKS], p->map.cap_blocks);   // (written by an earlier sweep: never more than the list holds)
  if (++p->sweeps_since_sort < p->sort_every || n0 < 4096 || (size_t)n0 * 2 < (size_t)p->sorted_n * 3) return SE_HIP_OK;
  if (p->sort_cap < n0) {
    if (p->bpos_sorted) hipFree(p->bpos_sorted);
    if (p->sort_tmp) hipFree(p->sort_tmp);
    p->bpos_sorted = nullptr; p->sort_tmp = nullptr;
    const size_t cap = std::min<size_t>((size_t)n0 + n0 / 2 + 65536, p->map.cap_blocks);
    HIP_TRY(hipMalloc((void**)&p->bpos_sorted, cap * sizeof(uint32_t)));
    size_t bytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, p->map.bpos, p->bpos_sorted, (int)cap, 0, 30, p->stream));
    HIP_TRY(hipMalloc(&p->sort_tmp, bytes));
    p->sort_tmp_bytes = bytes; p->sort_cap = cap;
  }
  size_t bytes = p->sort_tmp_bytes;
  HIP_TRY(hipcub::DeviceRadixSort::SortKeys(p->sort_tmp, bytes, p->map.bpos, p->bpos_sorted, (int)n0, 0, 30, p->stream));
  HIP_TRY(hipMemcpyAsync(p->map.bpos, p->bpos_sorted, (size_t)n0 * sizeof(uint32_t), hipMemcpyDeviceToDevice, p->stream));
  p->sorted_n = n0; p->sweeps_since_sort = 0;
  return SE_HIP_OK;
}

int se_hip_integrate_sweep(se_hip_pipeline* p, const float pose_cm[16], const float k[4], uint32_t rate, float mu, uint32_t frame) {
  if (int r = check(p)) return r;
  if (!pose_cm || !k || rate == 0) return fail(SE_HIP_E_INVALID, "bad argument");
  if (!finite_pose(pose_cm, k)) return fail(SE_HIP_E_INVALID, "non-finite pose or intrinsics");
  if (!stage_runs_integration(frame, rate)) return 0;
  if (int r = check_overflow(p)) return r;
  if (int r = join_scan(p, true)) return r;
  materialise_depth(p, p->stream);   // (a sweep without its scan: the stage API used out of order)
  input_slot_in_use(p);
  if (int r = sort_block_list(p)) return r;
  const DevMap& m = p->map;
  const M4 pose = from_colmajor(pose_cm);
  const bool sdf = p->cfg.field_type == SE_HIP_FIELD_SDF;
  IntegArgs a{};
  a.commit_occ = p->occ_commit_due ? 1 : 0;
  a.occ_lists = p->occ_lists;
  p->occ_commit_due = false;
  p->occ_lists = OccLists{nullptr, 0, 0};
  // Sophus::SE3f(pose_).inverse() (DenseSLAMSystem.cpp:237): (R^T, R^T * (t * -1)) taken from the matrix
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a.R[i * 3 + j] = pose.m[j][i];
  const float nt[3] = {pose.m[0][3] * -1.f, pose.m[1][3] * -1.f, pose.m[2][3] * -1.f};
  mul3(a.R, nt, a.t);
  const M4 K = camera_matrix(k);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a.K3[i * 3 + j] = K.m[i][j];
  M4 Tcw{};
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Tcw.m[i][j] = a.R[i * 3 + j]; Tcw.m[i][3] = a.t[i]; }
  Tcw.m[3][0] = Tcw.m[3][1] = Tcw.m[3][2] = 0.f; Tcw.m[3][3] = 1.f;
  const M4 cam = mul(K, Tcw);                        // projective_functor.hpp:64
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) a.cam[i * 4 + j] = cam.m[i][j];
  a.voxel = m.dim / (float)m.size;
  const float vs3[3] = {a.voxel, 0, 0};
  mul3(a.R, vs3, a.delta);                           // projective_functor.hpp:76-77
  mul3(a.K3, a.delta, a.cdelta);
  a.mu = mu;
  {
    // The sweep's shared-reciprocal divisions (se_rcp_refined, se_kernels.h) are the IEEE divisions while their operands stay in range; the
    // host bounds them here for every voxel of the volume: pos = R p + t with 0 <= p < dim on every axis.
    auto fin = [](float v) { return std::isfinite(v); };
    bool ok = fin(mu) && mu >= 0x1p-30f && mu <= 0x1p30f && fin(a.voxel) && a.voxel >= 0x1p-60f && m.dim <= 0x1p30f;
    for (int i = 0; i < 3 && ok; ++i) {
      float bound = std::fabs(a.t[i]);
      for (int j = 0; j < 3; ++j) bound += std::fabs(a.R[i * 3 + j]) * m.dim;
      ok = fin(bound) && bound < 0x1p40f;
    }
    // K of the form getCameraMatrix builds (commons.h:255-262), finite, so that K3 * start needs no products by zero and cam.z == pos.z
    ok = ok && a.K3[1] == 0.f && a.K3[3] == 0.f && a.K3[6] == 0.f && a.K3[7] == 0.f && a.K3[8] == 1.f;
    for (int i = 0; i < 9; ++i) ok = ok && fin(a.K3[i]) && std::fabs(a.K3[i]) < 0x1p40f;
    a.fast_div = (ok && !p->ieee_sweep) ? 1 : 0;
  }
  a.maxweight = 100.f;                               // DenseSLAMSystem.cpp:235
  a.timestamp = (1.f / 30.f) * frame;                // DenseSLAMSystem.cpp:243
  a.W = p->cfg.width; a.H = p->cfg.height;
  a.bspline = p->bspline; a.logodds = p->logodds;
  a.ctr_mirror = p->ctr_host;   // the kernel refreshes the host copy of the counters (next frame's launch geometry)
  if (p->map.newkeys == p->newkeys_own || p->map.newkeys == p->newkeys_own2) {
    const int other = p->map.newkeys == p->newkeys_own ? 1 : 0;   // the list this frame's scan did NOT use: consumed a frame ago
    a.zero_count = other ? p->newkeys_own2 : p->newkeys_own;
    p->own_clean[other] = true;
    p->own_next = other;
  }
  if (p->shard_world > 1) {
    a.shard_world = p->shard_world; a.shard_rank = p->shard_rank; a.shard_cap = (uint32_t)p->shard_cap;
    a.shard_count = (unsigned long long*)p->shard_send;
    a.shard_recs = (uint32_t*)(p->shard_send + SE_SHARD_SUB * 8);
    a.shard_vx = (float*)(p->shard_send + SE_SHARD_SUB * 8 + p->shard_cap * 4);
    a.shard_vy = a.shard_vx + p->shard_cap * 512;
    HIP_TRY(hipMemsetAsync(p->shard_send, 0, SE_SHARD_SUB * 8, p->stream));
  }
  if (p->prio_hint) {
    a.tile_cost = p->tile_cost; a.prio_thr = p->prio_thr;
    for (int i = 0; i < 3; ++i) a.prio_permille[i] = p->prio_permille[i];
    a.n_tiles = ((p->cfg.width + SE_TILE_W - 1) / SE_TILE_W) * ((p->row_end - p->row_begin + SE_TILE_H - 1) / SE_TILE_H);
    // the cost-sorted deal pays for the SDF march (raycast 40.8 -> 39.2 us at 512^3, 81.3 -> 79.1 at 1024^3); OFusion's cost
    // figure predicts its launch less well (126 -> 130 us): its workgroups stay in image order
    a.ray_order = sdf ? p->ray_order : nullptr;
  }
  const dim3 block(SE_WG);
  {
    // one launch: blocks (one wave each) then nodes.  The block count lives on the device; the grid is
    // sized from the count read back asynchronously after the previous sweep (+ headroom), and the
    // kernel's grid-stride loop covers whatever that estimate misses.
    ScopedTimer t(p, SE_HIP_K_INTEGRATE);
    const size_t est = (size_t)p->ctr_host[C_BLOCKS] + (size_t)p->ctr_host[C_BLOCKS] / 8 + 2048;
    const size_t wgs = std::min<size_t>(std::max<size_t>((est + 3) / 4, 2048), 65536);
    const dim3 grid(p->integ_grid > 0 ? (unsigned)p->integ_grid : (unsigned)wgs);
#define SE_SWEEP(OF, FD, SHD) hipLaunchKernelGGL((k_integrate<OF, FD, SHD>), grid, block, 0, p->stream, m, p->depth, a)
    a.stats = p->stats ? 1 : 0;
    switch ((sdf ? 0 : 4) | (a.fast_div ? 2 : 0) | (a.shard_world > 1 ? 1 : 0)) {
      case 0: SE_SWEEP(false, false, false); break;
      case 1: SE_SWEEP(false, false, true); break;
      case 2: SE_SWEEP(false, true, false); break;
      case 3: SE_SWEEP(false, true, true); break;
      case 4: SE_SWEEP(true, false, false); break;
      case 5: SE_SWEEP(true, false, true); break;
      case 6: SE_SWEEP(true, true, false); break;
      case 7: SE_SWEEP(true, true, true); break;
    }
#undef SE_SWEEP
  }
  // the next frame's scan / depth upload may start behind this point: an event for the scan stream to wait on, or (host
  // gate) the sequence number of the raycast that follows
  if (p->host_gate) { p->gate_armed = true; p->gate_followed = false; p->gate_target = p->ray_seq + 1u; }
  else {
    hipEventRecord(p->ev_sweep, p->stream);
    if (p->cur_in >= 0) {   // the input slot's last reader in a frame: what a later upload into the slot waits for (stage_input)
      const int i = p->cur_in;
      if (!p->in_done[i]) HIP_TRY(hipEventCreateWithFlags(&p->in_done[i], hipEventDisableTiming));
      HIP_TRY(hipEventRecord(p->in_done[i], p->stream));
      p->in_event[i] = true;
    }
  }
  HIP_TRY(hipGetLastError());
  return 1;
}

// The one-queue streaming schedule (se_hip_set_streaming): a frame's raycast is not launched by se_hip_frame / se_hip_raycast_deferred but held back
// until the next frame's allocation scan, which takes it along in one launch (launch_raycast_scan); any other API call launches it first (check()), so a
// caller that looks at a frame's images through the API, synchronises or tracks sees exactly what the eager schedule shows.  Plain single-device handles
// only, and only while the raycast's workgroups fit the chip in one round -- 640x480: 2 400 of 2 560 --: in a launch of several rounds the scan's
// workgroups inherit the raycast's register and LDS footprint and no longer slip into the gaps (1280x960 -> 2048^3: 356 us fused against 285 us side by
// side, profiles/r04p_march_skip_ab.log), so larger images keep the two-queue schedule.
static bool frame_can_fuse(se_hip_pipeline* p) {
  const int ray_pairs = (((p->cfg.width + SE_TILE_W - 1) / SE_TILE_W) * ((p->row_end - p->row_begin + SE_TILE_H - 1) / SE_TILE_H) + 1) / 2;
  // (r05: row-sharded replicas and handles whose key list goes into an exchange ride along too -- the all-gather and the commit of the gathered lists
  // then follow the fused launch on the main stream, se_hip_alloc_exchange / se_hip_alloc_commit; opt-in like everything here)
  return p->fuse && !p->ptrs_exposed && p->overlap && !p->stats && p->shard_world <= 1 && ray_pairs <= 10 * p->n_cus;
}
int se_hip_set_streaming(se_hip_pipeline* p, int32_t on) {
  if (int r = check(p)) return r;   // (switching off launches an outstanding raycast first)
  p->fuse = on != 0;
  p->flush_streak = 0;
  return frame_can_fuse(p) ? 1 : 0;
}
int se_hip_frame_is_fused(se_hip_pipeline* p) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");
  return frame_can_fuse(p) ? 1 : 0;
}
int se_hip_set_image_ring(se_hip_pipeline* p, float* device_ring, int32_t slots) {
  if (int r = check(p)) return r;
  if ((device_ring != nullptr) != (slots > 0)) return fail(SE_HIP_E_INVALID, "bad argument (ring and slots go together)");
  if (p->ring && device_ring != p->ring && p->vertex != p->vertex_own) {
    // leaving a ring (for none, or for another one: ADVICE r05 -- the caller may free the old ring when this returns): vertex_ / normal_ stay what they
    // were, the last raycast's images move into the handle's own buffers
    const size_t bytes = (size_t)p->cfg.width * p->cfg.height * 3 * sizeof(float);
    HIP_TRY(hipMemcpyAsync(p->vertex_own, p->vertex, bytes, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->normal_own, p->normal, bytes, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->vertex = p->vertex_own; p->normal = p->normal_own;
  }
  p->ring = device_ring; p->ring_slots = device_ring ? slots : 0;
  return SE_HIP_OK;
}
extern "C++" int flush_pending_raycast(se_hip_pipeline* p) {
  if (!p->has_pending) return SE_HIP_OK;
  InFrame guard(p);      // (the nested check() must not recurse)
  const int r = se_hip_raycast(p, p->pend_pose, p->pend_k, p->pend_mu, p->pend_frame);
  if (r >= 0) p->has_pending = false;   // (an error -- a pool overflow not yet acknowledged -- leaves it pending: launched by the first call that succeeds)
  return r < 0 ? r : SE_HIP_OK;
}

int se_hip_integrate(se_hip_pipeline* p, const float pose[16], const float k[4], uint32_t rate, float mu, uint32_t frame) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");
  InFrame guard(p);
  if (int r = check(p)) return r;
  if (!pose || !k || rate == 0) return fail(SE_HIP_E_INVALID, "bad argument");
  if (!finite_pose(pose, k)) return fail(SE_HIP_E_INVALID, "non-finite pose or intrinsics");
  // a deferred raycast must run before this frame's sweep: together with this frame's scan if there is one, else on its own, now
  if (p->has_pending && !(frame_can_fuse(p) && stage_runs_integration(frame, rate))) { if (int r = flush_pending_raycast(p)) return r; }
  int r = se_hip_alloc_scan(p, pose, k, rate, mu, frame);
  if (p->has_pending) { if (int q = flush_pending_raycast(p)) return q; }   // (cannot happen: a scan that ran took it along)
  if (r <= 0) return r;
  return se_hip_integrate_sweep(p, pose, k, rate, mu, frame);
}

// raycasting() of a streaming caller: launched with the next frame's allocation scan (or by whatever call comes first)
int se_hip_raycast_deferred(se_hip_pipeline* p, const float pose[16], const float k[4], float mu, uint32_t frame) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");
  InFrame guard(p);
  if (int r = check(p)) return r;
  if (!pose || !k) return fail(SE_HIP_E_INVALID, "bad argument");
  if (!finite_pose(pose, k)) return fail(SE_HIP_E_INVALID, "non-finite pose or intrinsics");
  if (p->has_pending) { if (int r = flush_pending_raycast(p)) return r; }   // two raycasts without an integration between them
  if (!(frame > 2)) return 0;      // DenseSLAMSystem.cpp:195
  if (!frame_can_fuse(p) || p->flush_streak >= 2) return se_hip_raycast(p, pose, k, mu, frame);
  if (int r = check_overflow(p)) return r;
  std::memcpy(p->pend_pose, pose, sizeof p->pend_pose); std::memcpy(p->pend_k, k, sizeof p->pend_k);
  p->pend_mu = mu; p->pend_frame = frame; p->has_pending = true;
  return 1;
}

// One frame of the hot path in one call: float_depth_ hand-over (device pointer) + integration() + raycasting().
// The same three calls a host makes per frame, without crossing the FFI three times (ctypes: ~5 us each).
int se_hip_frame(se_hip_pipeline* p, const float* device_depth_m, const float pose[16], const float k[4], uint32_t rate, float mu, uint32_t frame) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");
  InFrame guard(p);
  if (int r = check(p)) return r;
  if (!pose || !k || rate == 0) return fail(SE_HIP_E_INVALID, "bad argument");
  if (!finite_pose(pose, k)) return fail(SE_HIP_E_INVALID, "non-finite pose or intrinsics");
  if (device_depth_m) { p->depth = device_depth_m; p->cur_in = -1; p->in_pending.kind = 0; }
  int ran = 0;
  int r = se_hip_integrate(p, pose, k, rate, mu, frame);
  if (r < 0) return r;
  ran |= r > 0 ? 1 : 0;
  r = se_hip_raycast_deferred(p, pose, k, mu, frame);
  if (r < 0) return r;
  ran |= r > 0 ? 2 : 0;
  return ran;   // bit 0: integration ran, bit 1: raycasting ran (or is deferred)
}

// ------------------------------------------------------------------------------------ raycast
int se_hip_raycast(se_hip_pipeline* p, const float pose_cm[16], const float k[4], float mu, uint32_t frame) {
  if (int r = check(p)) return r;
  if (!pose_cm || !k) return fail(SE_HIP_E_INVALID, "bad argument");
  if (!finite_pose(pose_cm, k)) return fail(SE_HIP_E_INVALID, "non-finite pose or intrinsics");
  if (!(frame > 2)) return 0;  // DenseSLAMSystem.cpp:195
  if (int r = check_overflow(p)) return r;
  if (int r = join_scan(p)) return r;
  std::memcpy(p->raycast_pose, pose_cm, sizeof p->raycast_pose);   // raycast_pose_ = pose_ (DenseSLAMSystem.cpp:196)
  p->images_complete = !p->sharded;   // a row-sharded replica has just overwritten its own rows only
  select_image_target(p, frame);
  const DevMap& m = p->map;
  RayLaunchArgs L = make_ray_args(p, pose_cm, k, mu);
  if (p->host_gate) { L.a.gate = p->gate_host; L.a.gate_seq = ++p->ray_seq; if (p->gate_armed) p->gate_followed = true; input_slots_followed(p, L.a.gate_seq); }
  const RayArgs& a = L.a;
  const size_t smem = L.smem;
  const dim3 grid = L.grid, block(SE_WG_RAY);
  const bool sdf = p->cfg.field_type == SE_HIP_FIELD_SDF;
  {
    ScopedTimer t(p, SE_HIP_K_RAYCAST);
#define SE_RAY(OF, ST, DN, SH) hipLaunchKernelGGL((k_raycast<OF, ST, DN, SH>), grid, block, smem, p->stream, m, a, p->vertex, p->normal)
    const bool shallow = !a.has_deep;   // every non-leaf occupancy level is in LDS (volumes <= 512^3): specialised traversal loop
    // dense voxel planes of <= 4 GiB (512^3: 1 GiB): the lean march addresses them by 32-bit byte offsets from the scalar base
    const size_t nb = (size_t)(m.size >> 3);
    const bool o32 = m.dense && shallow && nb * nb * nb * (size_t)SE_BRICK_STRIDE * sizeof(float) <= ((size_t)4 << 30);
    const int variant = (sdf ? 0 : 4) | (p->stats ? 2 : 0) | (m.dense ? 1 : 0);
    switch (variant) {
      case 0: if (shallow) SE_RAY(false, false, false, true); else SE_RAY(false, false, false, false); break;
      case 1: if (o32) hipLaunchKernelGGL((k_raycast<false, false, true, true, true>), grid, block, smem, p->stream, m, a, p->vertex, p->normal);
              else if (shallow) SE_RAY(false, false, true, true); else SE_RAY(false, false, true, false); break;
      case 2: SE_RAY(false, true, false, false); break;
      case 3: SE_RAY(false, true, true, false); break;
      case 4: if (shallow) SE_RAY(true, false, false, true); else SE_RAY(true, false, false, false); break;
      case 5: if (o32) hipLaunchKernelGGL((k_raycast<true, false, true, true, true>), grid, block, smem, p->stream, m, a, p->vertex, p->normal);
              else if (shallow) SE_RAY(true, false, true, true); else SE_RAY(true, false, true, false); break;
      case 6: SE_RAY(true, true, false, false); break;
      case 7: SE_RAY(true, true, true, false); break;
    }
#undef SE_RAY
  }
  HIP_TRY(hipGetLastError());
  return 1;
}

// ---- the vertex_ / normal_ row tiles of the ranks -> full images on every rank
size_t se_hip_image_tile_bytes(se_hip_pipeline* p, int32_t max_rows) {
  if (!p || max_rows <= 0) return 0;
  return (size_t)2 * (size_t)max_rows * (size_t)p->cfg.width * 3 * sizeof(float);
}
int se_hip_pack_image_tile(se_hip_pipeline* p, void* send_device, int32_t max_rows) {
  if (int r = check(p)) return r;
  const int rows = p->row_end - p->row_begin;
  if (!send_device || max_rows < rows) return fail(SE_HIP_E_INVALID, "bad argument (max_rows smaller than this handle's row share)");
  const size_t row_bytes = (size_t)p->cfg.width * 3 * sizeof(float);
  char* dst = (char*)send_device;
  HIP_TRY(hipMemcpyAsync(dst, (const char*)p->vertex + (size_t)p->row_begin * row_bytes, (size_t)rows * row_bytes, hipMemcpyDeviceToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(dst + (size_t)max_rows * row_bytes, (const char*)p->normal + (size_t)p->row_begin * row_bytes, (size_t)rows * row_bytes, hipMemcpyDeviceToDevice, p->stream));
  return SE_HIP_OK;
}
int se_hip_apply_image_tiles(se_hip_pipeline* p, const void* recv_device, int32_t world, int32_t max_rows, const int32_t* row_begin, const int32_t* row_end) {
  if (int r = check(p)) return r;
  if (!recv_device || world < 1 || max_rows <= 0 || !row_begin || !row_end) return fail(SE_HIP_E_INVALID, "bad argument");
  const size_t row_bytes = (size_t)p->cfg.width * 3 * sizeof(float);
  const size_t tile = (size_t)2 * max_rows * row_bytes;
  for (int r = 0; r < world; ++r) {
    const int b = row_begin[r], e = row_end[r];
    if (b < 0 || e > p->cfg.height || e < b || e - b > max_rows) return fail(SE_HIP_E_INVALID, "bad row partition");
    if (b == p->row_begin && e == p->row_end) continue;    // the own rows are in place
    if (e == b) continue;
    const char* src = (const char*)recv_device + (size_t)r * tile;
    HIP_TRY(hipMemcpyAsync((char*)p->vertex + (size_t)b * row_bytes, src, (size_t)(e - b) * row_bytes, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync((char*)p->normal + (size_t)b * row_bytes, src + (size_t)max_rows * row_bytes, (size_t)(e - b) * row_bytes, hipMemcpyDeviceToDevice, p->stream));
  }
  p->images_complete = true;   // (the copies are on the main stream, in front of any later se_hip_track)
  return SE_HIP_OK;
}
int se_hip_gather_images(se_hip_pipeline* p, void* send_device, void* recv_device, int32_t max_rows, const int32_t* row_begin, const int32_t* row_end) {
  if (int r = check(p)) return r;
  if (!p->xgather) return fail(SE_HIP_E_INVALID, "no exchange set (se_hip_set_exchange)");
  if (int r = se_hip_pack_image_tile(p, send_device, max_rows)) return r;
  const int rc = p->xgather(send_device, recv_device, se_hip_image_tile_bytes(p, max_rows), /* ncclUint8 */ 1, p->xcomm, p->stream);
  if (rc != 0) return fail(SE_HIP_E_DEVICE, "ncclAllGather failed with code " + std::to_string(rc));
  return se_hip_apply_image_tiles(p, recv_device, p->xworld, max_rows, row_begin, row_end);
}

int se_hip_download_vertex_normal(se_hip_pipeline* p, float* v, float* n) {
  if (int r = check(p)) return r;
  const size_t bytes = (size_t)p->cfg.width * p->cfg.height * 3 * sizeof(float);
  if (v) HIP_TRY(hipMemcpyAsync(v, p->vertex, bytes, hipMemcpyDeviceToHost, p->stream));
  if (n) HIP_TRY(hipMemcpyAsync(n, p->normal, bytes, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return SE_HIP_OK;
}

int se_hip_vertex_normal_device(se_hip_pipeline* p, float** v, float** n) {
  if (int r = check(p)) return r;
  // Whoever holds these pointers can read the images without going through the API (its own kernel ordered on the handle's stream): with the raycast
  // of a frame deferred it would see the previous frame's.  So the hand-out ends deferral on this handle for good -- unless the images go into a ring
  // the caller supplied, whose contract (se_hip_set_image_ring) says when a slot is complete.
  if (!p->ring) p->ptrs_exposed = true;
  if (v) *v = p->vertex;
  if (n) *n = p->normal;
  return SE_HIP_OK;
}


// -------------------------------------------------------------------------------------- tracking
int se_hip_track(se_hip_pipeline* p, const float k[4], float icp_threshold, uint32_t tracking_rate, uint32_t frame,
                 const int32_t* pyramid, int32_t n_levels, float pose_cm[16]) {
  if (int r = check(p)) return r;
  if (!k || !pose_cm || !pyramid || n_levels < 1 || n_levels > 8 || tracking_rate == 0) return fail(SE_HIP_E_INVALID, "bad argument");
  if (frame % tracking_rate != 0) return 0;   // DenseSLAMSystem.cpp:146
  // the ICP reads the WHOLE of vertex_ / normal_; a row-sharded replica holds only its own rows of the last raycast until
  // the peers' tiles have been applied -- tracking against the stale rows of an earlier frame would be silent garbage
  if (p->sharded && !p->images_complete) return fail(SE_HIP_E_INVALID, "row-sharded handle: se_hip_gather_images / se_hip_apply_image_tiles must follow the raycast before se_hip_track");
  if (int r = join_scan(p)) return r;
  const int W = p->cfg.width, H = p->cfg.height;
  if ((W >> (n_levels - 1)) < 1 || (H >> (n_levels - 1)) < 1) return fail(SE_HIP_E_INVALID, "too many pyramid levels");
  hipStream_t s = p->stream;
  // buffers (first call)
  if ((int)p->pyr_vertex.size() < n_levels) {
    p->pyr_depth.resize(n_levels, nullptr); p->pyr_vertex.resize(n_levels, nullptr); p->pyr_normal.resize(n_levels, nullptr);
    for (int i = 0; i < n_levels; ++i) {
      const size_t n = (size_t)(W >> i) * (H >> i);
      if (!p->pyr_depth[i]) HIP_TRY(hipMalloc((void**)&p->pyr_depth[i], n * sizeof(float)));
      if (!p->pyr_vertex[i]) { HIP_TRY(hipMalloc((void**)&p->pyr_vertex[i], n * 3 * sizeof(float))); HIP_TRY(hipMemsetAsync(p->pyr_vertex[i], 0, n * 3 * sizeof(float), s)); }
      if (!p->pyr_normal[i]) { HIP_TRY(hipMalloc((void**)&p->pyr_normal[i], n * 3 * sizeof(float))); HIP_TRY(hipMemsetAsync(p->pyr_normal[i], 0, n * 3 * sizeof(float), s)); }
    }
  }
  if (!p->track) {
    HIP_TRY(hipMalloc((void**)&p->track, (size_t)W * H * sizeof(TrackData)));
    HIP_TRY(hipMemsetAsync(p->track, 0, (size_t)W * H * sizeof(TrackData), s));
    HIP_TRY(hipMalloc((void**)&p->reduce_partial, 2 * 8 * SE_TRACK_SEGMENTS * 32 * sizeof(float)));   // the iterations' partial rows alternate
    HIP_TRY(hipMalloc((void**)&p->icp, 2 * sizeof(IcpState)));                                       // ... and so does the state (k_icp_iter)
    HIP_TRY(hipHostMalloc((void**)&p->icp_host, sizeof(IcpHostRecord)));
    std::memset(p->icp_host, 0, sizeof(IcpHostRecord));
  }
  // pyramid (DenseSLAMSystem.cpp:149-163): scaled_depth_[0] is the current depth image
  materialise_depth(p, s);
  input_slot_in_use(p);
  const float* d0 = p->depth;
  if (p->filter_input) {   // DenseSLAMSystem.cpp:132-135; gaussian_: DenseSLAMSystem.cpp:111-118
    Gauss5 G;
    for (unsigned int i = 0; i < 5; i++) { const int x = (int)i - 2; G.g[i] = expf(-(x * x) / (2 * 4.0f * 4.0f)); }
    hipLaunchKernelGGL(k_bilateral_filter, dim3((W + 255) / 256, H), dim3(256), 0, s, p->pyr_depth[0], d0, W, H, G, 0.1f);
    d0 = p->pyr_depth[0];
  }
  // scaled_depth_[0] is a copy of float_depth_ in the reference too (DenseSLAMSystem.cpp:149-152): the tracker's level 0 must not alias an input
  // buffer that the next upload (or the caller, for zero-copy depth) may overwrite.  The copy and the first two half-sampling passes are one launch
  // (k_depth_pyramid; r04 -- before: a runtime copy, ~20 us of host time in front of the frame's first kernel, and two launches).
  {
    float* l1 = n_levels > 1 ? p->pyr_depth[1] : nullptr;
    float* l2 = n_levels > 2 ? p->pyr_depth[2] : nullptr;
    hipLaunchKernelGGL(k_depth_pyramid, dim3(((W + 3) / 4 + 255) / 256, (H + 3) / 4), dim3(256), 0, s, p->pyr_depth[0], l1, l2, d0, W, H, 0.1f * 3, p->filter_input ? 0 : 1);   // e_delta * 3
    d0 = p->pyr_depth[0];
  }
  p->scaled0 = d0;
  for (int i = 3; i < n_levels; ++i) {
    const int w = W >> i, h = H >> i;
    hipLaunchKernelGGL(k_half_sample, dim3((w + 255) / 256, h), dim3(256), 0, s, p->pyr_depth[i], w, h, p->pyr_depth[i - 1], W >> (i - 1), 0.1f * 3, 1);
  }
  {
    // depth2vertex + vertex2normal of every level: one launch (grid.z = level; the blocks beyond a coarser level's size return at once)
    PyrLevels L{};
    for (int i = 0; i < n_levels; ++i) {
      const float kk[4] = {k[0] / float(1 << i), k[1] / float(1 << i), k[2] / float(1 << i), k[3] / float(1 << i)};
      const M4 invK = inverse_camera_matrix(kk);
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) L.K[i].m[r * 4 + c] = invK.m[r][c];
      L.w[i] = W >> i; L.h[i] = H >> i;
      L.depth[i] = i == 0 ? d0 : p->pyr_depth[i];
      L.vertex[i] = p->pyr_vertex[i]; L.normal[i] = p->pyr_normal[i];
    }
    hipLaunchKernelGGL(k_vertex_normal_levels, dim3((W + 255) / 256, H, n_levels), dim3(256), 0, s, L, k[1] < 0 ? 1 : 0);
  }
  // The ICP loop (DenseSLAMSystem.cpp:165-186) is device-resident: the iterations of every level are enqueued from here, ONE launch each
  // (k_icp_iter: the previous iteration's final sums + updatePoseKernel as a prologue, then trackKernel + reduceKernel's partial sums); the
  // pose, the convergence flags and the sums travel from launch to launch in device memory (double-buffered), and the host reads one
  // pinned record when the frame's last launch (k_icp_finish_rows) has run its first workgroup.
  const M4 pose0 = from_colmajor(pose_cm);
  const M4 projectReference = mul(camera_matrix(k), rigid_inverse(from_colmajor(p->raycast_pose)));
  TrackArgs a{};
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) a.view[r * 4 + c] = projectReference.m[r][c];
  a.dist_threshold = 0.1f; a.normal_threshold = 0.8f;   // constant_parameters.h:19-20
  a.refW = W; a.refH = H;
  a.icp_threshold = icp_threshold;
  Pose16 P0;
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) P0.m[r * 4 + c] = pose0.m[r][c];
  const size_t part_words = (size_t)8 * SE_TRACK_SEGMENTS * 32;
  const unsigned seq = ++p->reduce_seq;
  // A level that has converged turns its remaining launches into launches that return at once (4.4 us each; 7 of the 10 fine-level launches on the
  // room stream).  The host therefore stays only `look` launches ahead of the device inside a level: every launch reports (launch index, stop flags)
  // in one pinned word, and before launch j of a level is enqueued the host has seen launch j - look store its state -- once that shows the level's
  // stop flag, the level's remaining launches (which would only copy the state) are not enqueued.  The result is the same state either way; the
  // wait is bounded, and a report that does not arrive switches the pruning off for the frame.  SE_HIP_ICP_LOOKAHEAD=0: everything up front (r03).
  int look = p->icp_lookahead;
  volatile unsigned long long* prog = &p->icp_host->progress;
  int j = 0, prev_level = -1;
  for (int level = n_levels - 1; level >= 0; --level) {
    a.inW = W / (1 << level); a.inH = H / (1 << level);
    a.level = level;
    for (int i = 0; i < pyramid[level]; ++i, ++j) {
      if (look > 0 && i >= look + 1) {
        const unsigned long long need = (unsigned long long)(j - look + 1);
        unsigned long long v = 0;
        const bool seen = spin_until([&] { v = *prog; return (unsigned)(v >> 32) == seq && ((v >> 8) & 0xffffffull) >= need; }, 20000);
        if (!seen) look = 0;
        else if ((v >> level) & 1ull) break;
      }
      hipLaunchKernelGGL(k_icp_iter, dim3(SE_TRACK_SEGMENTS, 8), dim3(SE_TRACK_LANES), 0, s, p->icp + (j & 1), p->icp + ((j + 1) & 1), p->pyr_vertex[level],
                         p->pyr_normal[level], p->vertex, p->normal, p->reduce_partial + ((j + 1) & 1) * part_words, p->reduce_partial + (j & 1) * part_words, a, prev_level, P0,
                         (unsigned long long*)&p->icp_host->progress, seq, j);
      prev_level = level;
    }
  }
  {
    // the frame's last launch: k_icp_finish (workgroup (0, 0)) + tracking_result_ of the finest level that ran any iteration
    int rows_level = -1;
    for (int level = 0; level < n_levels; ++level) if (pyramid[level] > 0) { rows_level = level; break; }
    if (rows_level < 0) {
      hipLaunchKernelGGL(k_icp_finish, dim3(1), dim3(SE_TRACK_LANES), 0, s, p->icp + (j & 1), p->icp + ((j + 1) & 1), p->reduce_partial + ((j + 1) & 1) * part_words,
                         p->icp_host, W, H, seq, icp_threshold, prev_level, P0);
    } else {
      a.inW = W / (1 << rows_level); a.inH = H / (1 << rows_level); a.level = rows_level;
      hipLaunchKernelGGL(k_icp_finish_rows, dim3((a.inW + SE_TRACK_LANES - 1) / SE_TRACK_LANES, a.inH + 1), dim3(SE_TRACK_LANES), 0, s, p->icp + (j & 1), p->icp + ((j + 1) & 1),
                         p->reduce_partial + ((j + 1) & 1) * part_words, p->icp_host, seq, prev_level, P0, p->track, p->pyr_vertex[rows_level], p->pyr_normal[rows_level],
                         p->vertex, p->normal, a);
    }
  }
  HIP_TRY(hipGetLastError());
  // the one host wait of the frame: the record lands in pinned memory (bounded spin, then a stream synchronisation)
  volatile unsigned* seq_word = &p->icp_host->seq;
  if (!spin_until([&] { return *seq_word == seq; }, 200000)) {
    HIP_TRY(hipStreamSynchronize(s));
    if (*seq_word != seq) return fail(SE_HIP_E_DEVICE, "tracking did not complete");
  }
  p->track_iterations = p->icp_host->iterations;
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) pose_cm[c * 4 + r] = p->icp_host->pose[r * 4 + c];
  return p->icp_host->tracked ? 1 : 0;
}

// One frame of the reference's loop with tracking on (se_apps/src/benchmark.cpp:115-150) in one call: float_depth_ hand-over,
//   tracked = tracking(); if (tracked || frame <= 3) integrated = integration(); raycasting();
// -- the calls a host makes per frame, without crossing the FFI four times and with the scan chained behind the ICP on the main queue.
// pose_cm: in = pose_, out = pose_ after tracking.  Returns bit 0: integrated, bit 1: raycast ran, bit 2: tracked.
int se_hip_frame_tracked(se_hip_pipeline* p, const float* device_depth_m, const float k[4], float icp_threshold, uint32_t tracking_rate,
                         const int32_t* pyramid, int32_t n_levels, float pose_cm[16], uint32_t integration_rate, float mu, uint32_t frame) {
  if (int r = check(p)) return r;
  if (!k || !pose_cm || !pyramid || integration_rate == 0 || tracking_rate == 0) return fail(SE_HIP_E_INVALID, "bad argument");
  if (device_depth_m) { p->depth = device_depth_m; p->cur_in = -1; p->in_pending.kind = 0; }
  int out = 0;
  int r = se_hip_track(p, k, icp_threshold, tracking_rate, frame, pyramid, n_levels, pose_cm);
  if (r < 0) return r;
  const bool tracked = r > 0;
  if (tracked) out |= 4;
  if (tracked || frame <= 3) {
    p->scan_on_main_once = true;
    r = se_hip_integrate(p, pose_cm, k, integration_rate, mu, frame);
    p->scan_on_main_once = false;
    if (r < 0) return r;
    if (r > 0) out |= 1;
  }
  r = se_hip_raycast(p, pose_cm, k, mu, frame);
  if (r < 0) return r;
  if (r > 0) out |= 2;
  return out;
}

int se_hip_filter_depth(se_hip_pipeline* p, int32_t on) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");   // (a flag only: does not launch a deferred raycast)
  p->filter_input = on != 0;
  return SE_HIP_OK;
}

int se_hip_download_scaled_depth(se_hip_pipeline* p, int32_t level, float* host_out) {
  if (int r = check(p)) return r;
  if (!host_out || level < 0 || level >= (int)p->pyr_depth.size() || !p->scaled0) return fail(SE_HIP_E_INVALID, "no such pyramid level (se_hip_track builds it)");
  const float* src = level == 0 ? p->scaled0 : p->pyr_depth[level];
  const size_t n = (size_t)(p->cfg.width >> level) * (p->cfg.height >> level);
  HIP_TRY(hipMemcpyAsync(host_out, src, n * sizeof(float), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return SE_HIP_OK;
}

int se_hip_download_track(se_hip_pipeline* p, void* host_trackdata, float host_reduce32[32], int32_t* iterations) {
  if (int r = check(p)) return r;
  if (!p->track) return fail(SE_HIP_E_INVALID, "se_hip_track has not run");
  if (host_trackdata) HIP_TRY(hipMemcpy(host_trackdata, p->track, (size_t)p->cfg.width * p->cfg.height * sizeof(TrackData), hipMemcpyDeviceToHost));
  if (host_reduce32) std::memcpy(host_reduce32, p->icp_host->reduce0, 32 * sizeof(float));
  if (iterations) *iterations = p->track_iterations;
  return SE_HIP_OK;
}


// -------------------------------------------------------------------------------------- rendering
static int render_target(se_hip_pipeline* p) {
  if (!p->rgbw) HIP_TRY(hipMalloc((void**)&p->rgbw, (size_t)p->cfg.width * p->cfg.height * 4));
  return SE_HIP_OK;
}
static int render_download(se_hip_pipeline* p, uint8_t* host) {
  HIP_TRY(hipMemcpyAsync(host, p->rgbw, (size_t)p->cfg.width * p->cfg.height * 4, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return SE_HIP_OK;
}

int se_hip_render_volume(se_hip_pipeline* p, uint8_t* host_rgbw, const float view_cm[16], const float k[4], float mu, float largestep,
                         uint32_t frame, uint32_t rate) {
  if (int r = check(p)) return r;
  if (!host_rgbw || !view_cm || !k || rate == 0) return fail(SE_HIP_E_INVALID, "bad argument");
  if (frame % rate != 0) return 0;   // DenseSLAMSystem.cpp:281
  if (int r = join_scan(p)) return r;
  if (int r = render_target(p)) return r;
  RayLaunchArgs L = make_ray_args(p, view_cm, k, mu);
  L.a.farp = 4.0f * 2.0f;            // farPlane * 2.0f (DenseSLAMSystem.cpp:285)
  L.a.largestep = largestep;
  L.a.row_begin = 0; L.a.row_end = p->cfg.height;
  ShadeArgs sh{};
  sh.light[0] = view_cm[12]; sh.light[1] = view_cm[13]; sh.light[2] = view_cm[14];
  sh.ambient[0] = sh.ambient[1] = sh.ambient[2] = 0.1f;   // constant_parameters.h:37
  // !viewPose_->isApprox(raycast_pose_): Eigen's fuzzy compare, ||a-b||^2 <= 1e-10 * min(||a||^2, ||b||^2)
  float d2 = 0, na = 0, nb = 0;
  for (int i = 0; i < 16; ++i) { const float d = view_cm[i] - p->raycast_pose[i]; d2 += d * d; na += view_cm[i] * view_cm[i]; nb += p->raycast_pose[i] * p->raycast_pose[i]; }
  sh.render = !(d2 <= 1e-5f * 1e-5f * std::min(na, nb)) ? 1 : 0;
  const int tiles = ((L.a.W + 7) / 8) * ((L.a.H + 7) / 8);
  const dim3 grid((tiles + SE_WG_RAY / 64 - 1) / (SE_WG_RAY / 64)), block(SE_WG_RAY);
  const bool sdf = p->cfg.field_type == SE_HIP_FIELD_SDF;
  const DevMap& m = p->map;
  if (sdf) { if (m.dense) hipLaunchKernelGGL((k_render_volume<false, true>), grid, block, 0, p->stream, m, L.a, sh, p->vertex, p->normal, p->rgbw);
             else hipLaunchKernelGGL((k_render_volume<false, false>), grid, block, 0, p->stream, m, L.a, sh, p->vertex, p->normal, p->rgbw); }
  else { if (m.dense) hipLaunchKernelGGL((k_render_volume<true, true>), grid, block, 0, p->stream, m, L.a, sh, p->vertex, p->normal, p->rgbw);
         else hipLaunchKernelGGL((k_render_volume<true, false>), grid, block, 0, p->stream, m, L.a, sh, p->vertex, p->normal, p->rgbw); }
  HIP_TRY(hipGetLastError());
  if (int r = render_download(p, host_rgbw)) return r;
  return 1;
}

int se_hip_render_depth(se_hip_pipeline* p, uint8_t* host_rgbw) {
  if (int r = check(p)) return r;
  if (!host_rgbw) return fail(SE_HIP_E_INVALID, "bad argument");
  if (int r = join_scan(p)) return r;
  if (int r = render_target(p)) return r;
  const int n = p->cfg.width * p->cfg.height;
  materialise_depth(p, p->stream);
  input_slot_in_use(p);
  hipLaunchKernelGGL(k_render_depth, dim3((n + 255) / 256), dim3(256), 0, p->stream, p->rgbw, p->depth, n, 0.4f, 4.0f);
  HIP_TRY(hipGetLastError());
  return render_download(p, host_rgbw);
}

int se_hip_render_track(se_hip_pipeline* p, uint8_t* host_rgbw) {
  if (int r = check(p)) return r;
  if (!host_rgbw) return fail(SE_HIP_E_INVALID, "bad argument");
  if (!p->track) return fail(SE_HIP_E_INVALID, "se_hip_track has not run");
  if (int r = render_target(p)) return r;
  const int n = p->cfg.width * p->cfg.height;
  hipLaunchKernelGGL(k_render_track, dim3((n + 255) / 256), dim3(256), 0, p->stream, p->rgbw, p->track, n);
  HIP_TRY(hipGetLastError());
  return render_download(p, host_rgbw);
}

// ----------------------------------------------------------------------------------- read-back
static int fetch_counters(se_hip_pipeline* p) {
  if (int r = join_scan(p)) return r;
  HIP_TRY(hipMemcpyAsync(p->ctr_host, p->map.ctr, C_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return check_overflow(p);
}

// ---------------------------------------------------------------------------------- mesh export
namespace {
int run_mesh(se_hip_pipeline* p, float* dev_out, unsigned long long capacity, unsigned long long* n) {
  if (int r = join_scan(p)) return r;
  if (!p->mc_table_ready) {
    signed char table[256][SE_MC_WIDTH];
    se_mc_expand(table);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(SE_MC_TRI), table, sizeof(table)));
    p->mc_table_ready = true;
  }
  if (!p->mesh_ctr) HIP_TRY(hipMalloc((void**)&p->mesh_ctr, 2 * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(p->mesh_ctr, 0, 2 * sizeof(unsigned long long), p->stream));
  MeshArgs a{dev_out, p->mesh_ctr, capacity};
  hipLaunchKernelGGL(k_mesh, dim3(4096), dim3(SE_WG), 0, p->stream, p->map, a);
  unsigned long long h[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(h, p->mesh_ctr, sizeof h, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  *n = dev_out ? h[1] : h[0];
  return SE_HIP_OK;
}
}  // namespace

int se_hip_mesh_count(se_hip_pipeline* p, int64_t* n_triangles) {
  if (int r = check(p)) return r;
  if (!n_triangles) return fail(SE_HIP_E_INVALID, "null argument");
  unsigned long long n = 0;
  if (int r = run_mesh(p, nullptr, 0, &n)) return r;
  *n_triangles = (int64_t)n;
  return SE_HIP_OK;
}

int se_hip_mesh_download(se_hip_pipeline* p, float* host_triangles, int64_t capacity_triangles, int64_t* n_written) {
  if (int r = check(p)) return r;
  if (!host_triangles || capacity_triangles < 0 || !n_written) return fail(SE_HIP_E_INVALID, "bad argument");
  *n_written = 0;
  if (capacity_triangles == 0) return SE_HIP_OK;
  float* dev = nullptr;
  HIP_TRY(hipMalloc((void**)&dev, (size_t)capacity_triangles * 9 * sizeof(float)));
  unsigned long long n = 0;
  int r = run_mesh(p, dev, (unsigned long long)capacity_triangles, &n);
  if (r == SE_HIP_OK) {
    const size_t w = (size_t)std::min<unsigned long long>(n, (unsigned long long)capacity_triangles);
    if (hipMemcpy(host_triangles, dev, w * 9 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) r = fail(SE_HIP_E_DEVICE, "hipMemcpy (mesh)");
    *n_written = (int64_t)w;
    if (r == SE_HIP_OK && n > (unsigned long long)capacity_triangles) r = fail(SE_HIP_E_CAPACITY, "triangle buffer too small (se_hip_mesh_count gives the size)");
  }
  hipFree(dev);
  return r;
}

// DenseSLAMSystem::dump_mesh (DenseSLAMSystem.cpp:302-322) + writeVtkMesh (se_denseslam/include/se/commons.h:325-390,
// no point / cell data).  The reference appends triangles in OpenMP completion order; here they are sorted by
// their bit patterns so that the file is reproducible.
int se_hip_dump_mesh(se_hip_pipeline* p, const char* filename) {
  if (int r = check(p)) return r;
  if (!filename) return fail(SE_HIP_E_INVALID, "null filename");
  int64_t n = 0;
  if (int r = se_hip_mesh_count(p, &n)) return r;
  std::vector<float> tri((size_t)n * 9);
  int64_t w = 0;
  if (n) if (int r = se_hip_mesh_download(p, tri.data(), n, &w)) return r;
  std::vector<size_t> order((size_t)w);
  std::iota(order.begin(), order.end(), (size_t)0);
  std::sort(order.begin(), order.end(), [&](size_t i, size_t j) { return std::memcmp(&tri[9 * i], &tri[9 * j], 36) < 0; });
  FILE* f = std::fopen(filename, "w");
  if (!f) return fail(SE_HIP_E_INVALID, std::string("cannot open ") + filename);
  std::fprintf(f, "# vtk DataFile Version 1.0\nvtk mesh generated from KFusion\nASCII\nDATASET POLYDATA\n");
  std::fprintf(f, "POINTS %lld FLOAT\n", (long long)(3 * w));
  for (size_t i : order)
    for (int v = 0; v < 3; ++v) std::fprintf(f, "%g %g %g\n", tri[9 * i + 3 * v], tri[9 * i + 3 * v + 1], tri[9 * i + 3 * v + 2]);
  std::fprintf(f, "POLYGONS %lld %lld\n", (long long)w, (long long)(4 * w));
  for (long long i = 0; i < (long long)w; ++i) std::fprintf(f, "3 %lld %lld %lld\n", 3 * i, 3 * i + 1, 3 * i + 2);
  std::fprintf(f, "\n");
  std::fclose(f);
  return SE_HIP_OK;
}

int se_hip_memory_info(se_hip_pipeline* p, int64_t out[4]) {
  if (!p || !out) return fail(SE_HIP_E_INVALID, "bad argument");
  const DevMap& m = p->map;
  const size_t px = (size_t)p->cfg.width * p->cfg.height;
  const size_t bricks = p->slots * 1024 * sizeof(float);
  size_t all = bricks + p->tab_entries * 4 + (p->occ_words + p->lbits_words + 2 * p->cbits_words + p->fbits_words) * 4 + p->cap_blocks * 4 + ((p->slots + 3) & ~(size_t)3) +
               p->cap_nodes * (2 * 8 * 4 + 4 + 1) + 2 * (m.cap_keys + 1) * 8 + se_hip_pipeline::kIn * px * 4 + 2 * px * 3 * 4;
  all += p->sort_cap * sizeof(uint32_t) + p->sort_tmp_bytes;   // (sort_block_list: allocated at the first sort)
  out[0] = m.dense ? 1 : 0; out[1] = (int64_t)p->slots; out[2] = (int64_t)bricks; out[3] = (int64_t)all;
  return SE_HIP_OK;
}

int se_hip_counts(se_hip_pipeline* p, int32_t* nb, int32_t* nn) {
  if (int r = check(p)) return r;
  if (int r = fetch_counters(p)) return r;
  if (nb) *nb = (int32_t)p->ctr_host[C_BLOCKS];
  if (nn) *nn = (int32_t)p->ctr_host[C_NODES];
  return SE_HIP_OK;
}

int se_hip_download_blocks(se_hip_pipeline* p, int32_t* coords, float* x, float* y, uint8_t* active) {
  if (int r = check(p)) return r;
  if (int r = fetch_counters(p)) return r;
  const size_t n = p->ctr_host[C_BLOCKS];
  if (n == 0) return SE_HIP_OK;
  std::vector<uint32_t> pos(n);
  HIP_TRY(hipMemcpy(pos.data(), p->map.bpos, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  const int L = p->leaf_level;
  auto slot_of = [&](size_t i) -> size_t {
    if (!p->map.dense) return i;
    const uint32_t bp = pos[i];
    return ((((size_t)(bp >> 20) << L) | ((bp >> 10) & 1023u)) << L) | (bp & 1023u);
  };
  std::vector<uint8_t> act_all(p->slots);
  HIP_TRY(hipMemcpy(act_all.data(), p->map.bactive, p->slots, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> key(n);
  for (size_t i = 0; i < n; ++i)
    key[i] = se_make_key(pos[i] & 1023u, (pos[i] >> 10) & 1023u, pos[i] >> 20, p->leaf_level, p->max_level);
  std::vector<size_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
  std::vector<uint32_t> slots_sorted(n);
  for (size_t i = 0; i < n; ++i) {
    const size_t s = order[i];
    const size_t slot = slot_of(s);
    slots_sorted[i] = (uint32_t)slot;
    if (coords) { coords[3 * i] = (int)(pos[s] & 1023u) << 3; coords[3 * i + 1] = (int)((pos[s] >> 10) & 1023u) << 3; coords[3 * i + 2] = (int)(pos[s] >> 20) << 3; }
    if (active) active[i] = act_all[slot];
  }
  if (x || y) {
    uint32_t* d_slots = nullptr;
    float* d_pack = nullptr;
    HIP_TRY(hipMalloc((void**)&d_slots, n * sizeof(uint32_t)));
    hipError_t e2 = hipMalloc((void**)&d_pack, n * 512 * sizeof(float));
    if (e2 != hipSuccess) { hipFree(d_slots); return fail(SE_HIP_E_DEVICE, "hipMalloc (download staging)"); }
    hipMemcpyAsync(d_slots, slots_sorted.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, p->stream);
    for (int plane = 0; plane < 2; ++plane) {
      float* dst = plane ? y : x;
      if (!dst) continue;
      if (plane) hipLaunchKernelGGL(k_gather_bricks_y, dim3(2048), dim3(SE_WG), 0, p->stream, p->map, d_slots, n, d_pack);
      else hipLaunchKernelGGL(k_gather_bricks, dim3(2048), dim3(SE_WG), 0, p->stream, p->map.vx, d_slots, n, d_pack);
      hipMemcpyAsync(dst, d_pack, n * 512 * sizeof(float), hipMemcpyDeviceToHost, p->stream);
    }
    hipError_t e3 = hipStreamSynchronize(p->stream);
    hipFree(d_slots);
    hipFree(d_pack);
    if (e3 != hipSuccess) return fail(SE_HIP_E_DEVICE, std::string("download_blocks: ") + hipGetErrorString(e3));
  }
  return SE_HIP_OK;
}

int se_hip_download_nodes(se_hip_pipeline* p, uint64_t* code, uint32_t* side, float* x, float* y) {
  if (int r = check(p)) return r;
  if (int r = fetch_counters(p)) return r;
  const size_t n = p->ctr_host[C_NODES];
  std::vector<uint32_t> pos(n);
  std::vector<uint8_t> lvl(n);
  std::vector<float> hx(n * 8), hy(n * 8);
  HIP_TRY(hipMemcpy(pos.data(), p->map.npos, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(lvl.data(), p->map.nlevel, n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hx.data(), p->map.nx, n * 8 * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hy.data(), p->map.ny, n * 8 * sizeof(float), hipMemcpyDeviceToHost));
  std::vector<unsigned long long> key(n);
  for (size_t i = 0; i < n; ++i)
    key[i] = lvl[i] == 0 ? 0ull : se_make_key(pos[i] & 1023u, (pos[i] >> 10) & 1023u, pos[i] >> 20, lvl[i], p->max_level);
  std::vector<size_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
  for (size_t i = 0; i < n; ++i) {
    const size_t s = order[i];
    if (code) code[i] = key[s];
    if (side) side[i] = (uint32_t)p->map.size >> lvl[s];
    if (x) std::memcpy(x + i * 8, hx.data() + s * 8, 8 * sizeof(float));
    if (y) std::memcpy(y + i * 8, hy.data() + s * 8, 8 * sizeof(float));
  }
  return SE_HIP_OK;
}


// ------------------------------------------------------------------------------------ map export
int se_hip_save_map(se_hip_pipeline* p, const char* filename) {
  if (int r = check(p)) return r;
  if (!filename) return fail(SE_HIP_E_INVALID, "bad argument");
  int32_t nb = 0, nn = 0;
  if (int r = se_hip_counts(p, &nb, &nn)) return r;
  std::vector<uint64_t> ncode(nn);
  std::vector<uint32_t> nside(nn);
  std::vector<float> nx((size_t)nn * 8), ny((size_t)nn * 8);
  if (int r = se_hip_download_nodes(p, ncode.data(), nside.data(), nx.data(), ny.data())) return r;
  std::vector<int32_t> coords((size_t)nb * 3);
  std::vector<float> bx((size_t)nb * 512), by((size_t)nb * 512);
  if (nb) if (int r = se_hip_download_blocks(p, coords.data(), bx.data(), by.data(), nullptr)) return r;
  FILE* f = std::fopen(filename, "wb");
  if (!f) return fail(SE_HIP_E_INVALID, std::string("cannot open ") + filename);
  const bool sdf = p->cfg.field_type == SE_HIP_FIELD_SDF;
  auto put_value = [&](float x, float y) {
    if (sdf) { const float v[2] = {x, y}; std::fwrite(v, 4, 2, f); }
    else { const float xv = x; const uint32_t pad = 0; const double yv = y; std::fwrite(&xv, 4, 1, f); std::fwrite(&pad, 4, 1, f); std::fwrite(&yv, 8, 1, f); }
  };
  const int32_t size = p->map.size;
  const float dim = p->map.dim;
  std::fwrite(&size, 4, 1, f);
  std::fwrite(&dim, 4, 1, f);
  uint64_t n = (uint64_t)nn;
  std::fwrite(&n, 8, 1, f);
  for (int i = 0; i < nn; ++i) {
    const int32_t side = (int32_t)nside[i];
    std::fwrite(&ncode[i], 8, 1, f);
    std::fwrite(&side, 4, 1, f);
    for (int j = 0; j < 8; ++j) put_value(nx[(size_t)i * 8 + j], ny[(size_t)i * 8 + j]);
  }
  n = (uint64_t)nb;
  std::fwrite(&n, 8, 1, f);
  for (int i = 0; i < nb; ++i) {
    const uint64_t code = se_make_key(coords[3 * i] >> 3, coords[3 * i + 1] >> 3, coords[3 * i + 2] >> 3, p->leaf_level, p->max_level);
    std::fwrite(&code, 8, 1, f);
    std::fwrite(&coords[3 * (size_t)i], 4, 3, f);
    for (int j = 0; j < 512; ++j) put_value(bx[(size_t)i * 512 + j], by[(size_t)i * 512 + j]);
  }
  const bool okw = std::ferror(f) == 0;
  std::fclose(f);
  return okw ? SE_HIP_OK : fail(SE_HIP_E_INVALID, "write error");
}

// Octree::load (se_core/include/se/octree.hpp:917-950) for the device map: reads the byte layout of Octree::save /
// se_hip_save_map, re-initialises the map (the reference's load() calls init()), inserts every node and block and
// restores their values.  Two defects of the reference's load() are NOT reproduced: it reads `dim` into an int (the file
// holds a float, octree.hpp:921-924) and it copies only the first voxel of every block (sizeof(*getBlockRawPtr()),
// octree.hpp:945-946); here `dim` is read as the float it is and all 512 voxels are restored.  As in the reference,
// blocks come back with active_ = true (Octree::insert, octree.hpp:518).  The file must have been written for the same
// volume (size, dim) and field type as this handle.
int se_hip_load_map(se_hip_pipeline* p, const char* filename) {
  if (int r = check(p)) return r;
  if (!filename) return fail(SE_HIP_E_INVALID, "bad argument");
  FILE* f = std::fopen(filename, "rb");
  if (!f) return fail(SE_HIP_E_INVALID, std::string("cannot open ") + filename);
  const bool sdf = p->cfg.field_type == SE_HIP_FIELD_SDF;
  const size_t vbytes = sdf ? 8 : 16;
  int32_t size = 0; float dim = 0.f; uint64_t nn = 0, nb = 0;
  bool okr = std::fread(&size, 4, 1, f) == 1 && std::fread(&dim, 4, 1, f) == 1 && std::fread(&nn, 8, 1, f) == 1;
  if (!okr || size != p->map.size || dim != p->map.dim) { std::fclose(f); return fail(SE_HIP_E_INVALID, "map file does not match this volume (size / dim)"); }
  if (nn > p->cap_nodes) { std::fclose(f); return fail(SE_HIP_E_CAPACITY, "map file holds more nodes than the pool"); }
  auto get_values = [&](size_t count, float* x, float* y) -> bool {
    std::vector<unsigned char> raw(count * vbytes);
    if (std::fread(raw.data(), 1, raw.size(), f) != raw.size()) return false;
    for (size_t i = 0; i < count; ++i) {
      std::memcpy(&x[i], &raw[i * vbytes], 4);
      if (sdf) std::memcpy(&y[i], &raw[i * vbytes + 4], 4);
      else { double d; std::memcpy(&d, &raw[i * vbytes + 8], 8); y[i] = (float)d; }
    }
    return true;
  };
  std::vector<unsigned long long> keys; keys.reserve((size_t)nn + 1);
  std::vector<float> nx((size_t)nn * 8), ny((size_t)nn * 8);
  for (uint64_t i = 0; i < nn && okr; ++i) {
    unsigned long long code = 0; int32_t side = 0;
    okr = std::fread(&code, 8, 1, f) == 1 && std::fread(&side, 4, 1, f) == 1 && get_values(8, &nx[i * 8], &ny[i * 8]);
    keys.push_back(code);
  }
  okr = okr && std::fread(&nb, 8, 1, f) == 1;
  if (okr && nb > p->cap_blocks) { std::fclose(f); return fail(SE_HIP_E_CAPACITY, "map file holds more blocks than the pool (raise max_blocks)"); }
  std::vector<int32_t> coords((size_t)nb * 3);
  std::vector<float> bx((size_t)nb * 512), by((size_t)nb * 512);
  std::vector<unsigned long long> bkeys((size_t)nb);
  for (uint64_t i = 0; i < nb && okr; ++i) {
    okr = std::fread(&bkeys[i], 8, 1, f) == 1 && std::fread(&coords[i * 3], 4, 3, f) == 3 && get_values(512, &bx[i * 512], &by[i * 512]);
    if (okr) for (int c = 0; c < 3; ++c) okr = okr && coords[i * 3 + c] >= 0 && coords[i * 3 + c] < size && (coords[i * 3 + c] & 7) == 0;
  }
  std::fclose(f);
  if (!okr) return fail(SE_HIP_E_INVALID, "truncated or malformed map file");
  if (p->map.ybyte) {
    // SDF weights live in one byte each on the device (se_device.h): what sdf_update produces -- integers up to maxweight = 100 -- fits; anything else is not
    // a map of this pipeline or of the reference's and is refused rather than rounded
    for (float w : by) if (!(w >= 0.f && w <= 255.f && w == (float)(int)w)) return fail(SE_HIP_E_INVALID, "map file: a voxel weight is not an integer in 0..255 (SDF weights are stored as bytes)");
  }
  // node keys straight from the file: level and position must be those of an internal node of THIS tree
  for (unsigned long long kx : keys) {
    const int level = (int)(kx & 0x1FFull);
    if (level == 0 && kx != 0ull) return fail(SE_HIP_E_INVALID, "malformed map file (node key)");
    if (level == 0) continue;
    if (level >= p->leaf_level) return fail(SE_HIP_E_INVALID, "malformed map file (node level)");
    const unsigned long long code = kx & ~0x1FFull;
    const int sh = p->max_level - level;
    const unsigned long long lim = 1ull << level;
    if ((se_compact21(code) >> sh) >= lim || (se_compact21(code >> 1) >> sh) >= lim || (se_compact21(code >> 2) >> sh) >= lim)
      return fail(SE_HIP_E_INVALID, "malformed map file (node position)");
  }
  std::vector<unsigned long long> list;
  list.reserve(1 + keys.size() + bkeys.size());
  list.push_back(0ull);
  for (unsigned long long kx : keys) if ((kx & 0x1FFull) != 0ull) list.push_back(kx);   // (the root, key 0, exists already)
  for (uint64_t i = 0; i < nb; ++i)
    list.push_back(se_make_key(coords[i * 3] >> 3, coords[i * 3 + 1] >> 3, coords[i * 3 + 2] >> 3, p->leaf_level, p->max_level));
  list[0] = list.size() - 1;
  unsigned long long* d_list = nullptr; float *d_x = nullptr, *d_y = nullptr; int32_t* d_c = nullptr; unsigned long long* d_k = nullptr;
  auto cleanup = [&]() { for (void* q : {(void*)d_list, (void*)d_x, (void*)d_y, (void*)d_c, (void*)d_k}) if (q) hipFree(q); };
  const size_t nval = std::max((size_t)nb * 512, (size_t)nn * 8) + 8;
  if (hipMalloc((void**)&d_list, list.size() * 8) != hipSuccess || hipMalloc((void**)&d_x, nval * 4) != hipSuccess || hipMalloc((void**)&d_y, nval * 4) != hipSuccess ||
      hipMalloc((void**)&d_c, ((size_t)nb * 3 + 4) * 4) != hipSuccess || hipMalloc((void**)&d_k, ((size_t)nn + 1) * 8) != hipSuccess) { cleanup(); return fail(SE_HIP_E_DEVICE, "hipMalloc (load staging)"); }
  // everything that can fail without touching the map has succeeded: quiesce, re-initialise, insert
  if (p->side) HIP_TRY(hipStreamSynchronize(p->side));
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->scan_pending = false; p->occ_commit_due = false; p->occ_lists = OccLists{nullptr, 0, 0};
  reset_map_state(p);
  hipMemcpyAsync(d_list, list.data(), list.size() * 8, hipMemcpyHostToDevice, p->stream);
  hipLaunchKernelGGL(k_alloc_commit, dim3(256, 1), dim3(SE_WG), 0, p->stream, p->map, d_list, 1, (long long)list.size());
  if (nn) {
    hipMemcpyAsync(d_k, keys.data(), (size_t)nn * 8, hipMemcpyHostToDevice, p->stream);
    hipMemcpyAsync(d_x, nx.data(), (size_t)nn * 8 * 4, hipMemcpyHostToDevice, p->stream);
    hipMemcpyAsync(d_y, ny.data(), (size_t)nn * 8 * 4, hipMemcpyHostToDevice, p->stream);
    hipLaunchKernelGGL(k_load_nodes, dim3(grid_for((size_t)nn * 8, SE_WG, 4096)), dim3(SE_WG), 0, p->stream, p->map, d_k, d_x, d_y, (size_t)nn);
    hipStreamSynchronize(p->stream);   // the staging buffers are reused for the blocks
  }
  if (nb) {
    hipMemcpyAsync(d_c, coords.data(), (size_t)nb * 3 * 4, hipMemcpyHostToDevice, p->stream);
    hipMemcpyAsync(d_x, bx.data(), (size_t)nb * 512 * 4, hipMemcpyHostToDevice, p->stream);
    hipMemcpyAsync(d_y, by.data(), (size_t)nb * 512 * 4, hipMemcpyHostToDevice, p->stream);
    hipLaunchKernelGGL(k_load_blocks, dim3(grid_for((size_t)nb * 512, SE_WG, 16384)), dim3(SE_WG), 0, p->stream, p->map, d_c, d_x, d_y, (size_t)nb);
  }
  const hipError_t e = hipStreamSynchronize(p->stream);
  cleanup();
  if (e != hipSuccess) return fail(SE_HIP_E_DEVICE, std::string("load_map: ") + hipGetErrorString(e));
  int32_t gb = 0, gn = 0;
  if (int r = se_hip_counts(p, &gb, &gn)) return r;
  if ((uint64_t)gb != nb || (uint64_t)gn != nn) return fail(SE_HIP_E_INVALID, "map file is not ancestor-closed (octants without parents) or holds duplicates");
  return SE_HIP_OK;
}

// DenseSLAMSystem::dump_volume (DenseSLAMSystem.h:219) has an empty body in the reference (DenseSLAMSystem.cpp:270-272); the
// useful thing behind the name -- the whole volume on disk -- is se_hip_save_map.

int se_hip_create_replicas(const se_hip_config* cfg, const int32_t* device_ids, int32_t n_devices, se_hip_pipeline** out_handles) {
  if (!cfg || !device_ids || !out_handles || n_devices < 1) return fail(SE_HIP_E_INVALID, "bad argument");
  const int units = (cfg->height + 7) / 8;
  for (int i = 0; i < n_devices; ++i) out_handles[i] = nullptr;
  for (int i = 0; i < n_devices; ++i) {
    se_hip_config c = *cfg;
    c.device = device_ids[i];
    c.row_begin = std::min(cfg->height, 8 * ((units * i) / n_devices));
    c.row_end = (i + 1 == n_devices) ? cfg->height : std::min(cfg->height, 8 * ((units * (i + 1)) / n_devices));
    const int r = (c.row_end > c.row_begin) ? se_hip_create(&c, &out_handles[i]) : fail(SE_HIP_E_INVALID, "more devices than 8-row image tiles");
    if (r != SE_HIP_OK) { const std::string msg = g_err; for (int j = 0; j < i; ++j) { se_hip_destroy(out_handles[j]); out_handles[j] = nullptr; } return fail(r, msg); }
  }
  return SE_HIP_OK;
}

// --------------------------------------------------------------------------------- measurement
int se_hip_enable_timing(se_hip_pipeline* p, int32_t on) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");   // (a flag only: does not launch a deferred raycast, see se_hip_frame)
  p->timing = on != 0;  // no synchronisation here: pending events are resolved by se_hip_get_timings
  return SE_HIP_OK;
}

int se_hip_get_timings(se_hip_pipeline* p, double ms_sum[SE_HIP_K_COUNT], int64_t launches[SE_HIP_K_COUNT], int32_t reset) {
  if (int r = check(p)) return r;
  drain_timings(p);
  for (int i = 0; i < SE_HIP_K_COUNT; ++i) {
    if (ms_sum) ms_sum[i] = p->ms_sum[i];
    if (launches) launches[i] = p->launches[i];
    if (reset) { p->ms_sum[i] = 0; p->launches[i] = 0; }
  }
  return SE_HIP_OK;
}

int se_hip_get_launch_counts(se_hip_pipeline* p, int64_t counts[SE_HIP_K_COUNT + 1], int32_t reset) {
  if (!p) return fail(SE_HIP_E_INVALID, "null handle");   // (host counters only: does not launch a deferred raycast)
  for (int i = 0; i <= SE_HIP_K_COUNT; ++i) {
    if (counts) counts[i] = p->n_launch[i];
    if (reset) p->n_launch[i] = 0;
  }
  return p->has_pending ? 1 : 0;
}

int se_hip_enable_stats(se_hip_pipeline* p, int32_t on) {
  if (int r = check(p)) return r;
  p->stats = on != 0;
  HIP_TRY(hipMemsetAsync(p->map.stats, 0, S_COUNT * sizeof(unsigned long long), p->stream));
  return SE_HIP_OK;
}

int se_hip_get_stats(se_hip_pipeline* p, uint64_t out[16], int32_t reset) {
  if (int r = check(p)) return r;
  if (int r = join_scan(p)) return r;
  unsigned long long h[S_COUNT];
  HIP_TRY(hipMemcpyAsync(h, p->map.stats, sizeof h, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (int r = fetch_counters(p)) return r;
  h[S_NODES] = p->ctr_host[C_NODES];
  for (int i = 0; i < S_COUNT; ++i) out[i] = h[i];
  if (reset) HIP_TRY(hipMemsetAsync(p->map.stats, 0, S_COUNT * sizeof(unsigned long long), p->stream));
  return SE_HIP_OK;
}


}  // extern "C"
