/*
 * DenseSLAMSystem -- C++ mirror of the reference's pipeline class for the hot path
 *   se_denseslam/include/se/DenseSLAMSystem.h:58-411  (class DenseSLAMSystem)
 * on top of the C ABI of include/se_hip.h.  Header-only: an application written against the
 * reference class includes this header instead, defines SE_FIELD_TYPE (SDF or OFusion, as the
 * reference requires: DenseSLAMSystem.h:54) and links libse_hip.so.
 *
 * What is kept (same names, argument meaning and "did the stage run" return values):
 *   the two constructors, preprocessing() (mm -> metres, preprocessing.cpp:161-188, fused into the
 *   upload; the optional bilateral filter runs on the device in front of tracking()), integration(),
 *   tracking(), raycasting(), setPose()/getPose()/getPosition()/getInitPos(), getIntegrated(), getTracked(),
 *   getModelDimensions()/getModelResolution()/getComputationResolution(), renderVolume()/renderTrack()/
 *   renderDepth(), dump_mesh(), setViewPose()/getViewPose(), synchroniseDevices().
 * What differs, because the map lives in HBM:
 *   getMap() returns a host snapshot (MapSnapshot: blocks sorted by Morton key) instead of a
 *   shared_ptr<se::Octree>; getVertex()/getNormal() download vertex_/normal_.
 *   tracking() runs the reference's ICP on the device; poses can still be injected
 *   with setPose(), as the reference's GUI does with ground truth (se_apps/src/mainQt.cpp:257-265).
 *   renderVolume / renderTrack / renderDepth shade on the device and copy the RGBW image out.
 * Errors of the C ABI are reported like the reference reports its own failures: message on
 * std::cerr; constructors additionally throw std::runtime_error (the reference would dereference
 * an unallocated map).
 */
#ifndef SE_HIP_DENSESLAMSYSTEM_H
#define SE_HIP_DENSESLAMSYSTEM_H

#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../se_hip.h"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define SE_HIP_HAVE_EIGEN 1
#endif
#endif
#ifndef SE_HIP_HAVE_EIGEN
/* Eigen is not installed: minimal PODs with the storage layout and the few accessors the
 * DenseSLAMSystem interface needs (column-major Matrix4f, .data(), operator()). */
namespace Eigen {
template <typename T, int N> struct SeVec {
  T v[N];
  SeVec() : v() {}
  SeVec(T a, T b) { static_assert(N == 2, ""); v[0] = a; v[1] = b; }
  SeVec(T a, T b, T c) { static_assert(N == 3, ""); v[0] = a; v[1] = b; v[2] = c; }
  SeVec(T a, T b, T c, T d) { static_assert(N == 4, ""); v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
  T& operator()(int i) { return v[i]; }
  const T& operator()(int i) const { return v[i]; }
  T x() const { return v[0]; }
  T y() const { return v[1]; }
  T z() const { return v[2]; }
  T w() const { return v[3]; }
  const T* data() const { return v; }
};
typedef SeVec<int, 2> Vector2i;
typedef SeVec<int, 3> Vector3i;
typedef SeVec<float, 3> Vector3f;
typedef SeVec<float, 4> Vector4f;
struct Matrix4f {
  float m[16];  // column-major
  Matrix4f() : m() {}
  static Matrix4f Identity() { Matrix4f a; a.m[0] = a.m[5] = a.m[10] = a.m[15] = 1.f; return a; }
  float& operator()(int r, int c) { return m[c * 4 + r]; }
  const float& operator()(int r, int c) const { return m[c * 4 + r]; }
  const float* data() const { return m; }
  float* data() { return m; }
};
}  // namespace Eigen
#endif

struct SDF {};      /* field-type tags: se_denseslam/include/se/volume_traits.hpp:41-72 */
struct OFusion {};
#ifndef SE_FIELD_TYPE
#error "define SE_FIELD_TYPE to SDF or OFusion before including se/DenseSLAMSystem.h (as the reference requires)"
#endif
typedef SE_FIELD_TYPE FieldType;

/* the fields of the reference's Configuration (se_denseslam/include/se/config.h) this path reads */
struct Configuration {
  float mu = 0.1f;
  int device = 0;
  long long max_blocks = 0;
};

struct MapSnapshot {
  int n_blocks = 0, n_nodes = 0;
  std::vector<int32_t> coords;   /* [n][3] block min corners (voxels), sorted by Morton key */
  std::vector<float> x, y;       /* [n][512], voxel index x + 8y + 64z */
  std::vector<uint8_t> active;
};

class DenseSLAMSystem {
 public:
  DenseSLAMSystem(const Eigen::Vector2i& inputSize, const Eigen::Vector3i& volumeResolution,
                  const Eigen::Vector3f& volumeDimensions, const Eigen::Vector3f& initPose, std::vector<int>& pyramid,
                  const Configuration& config)
      : DenseSLAMSystem(inputSize, volumeResolution, volumeDimensions, toMatrix4f(initPose), pyramid, config) {}

  DenseSLAMSystem(const Eigen::Vector2i& inputSize, const Eigen::Vector3i& volumeResolution,
                  const Eigen::Vector3f& volumeDimensions, const Eigen::Matrix4f& initPose, std::vector<int>& pyramid,
                  const Configuration& config)
      : computation_size_(inputSize), volume_resolution_(volumeResolution), volume_dimension_(volumeDimensions) {
    iterations_.assign(pyramid.begin(), pyramid.end());
    if (iterations_.empty()) iterations_ = {10, 5, 4};
    init_pose_ = Eigen::Vector3f(initPose(0, 3), initPose(1, 3), initPose(2, 3));
    mu_ = config.mu;
    pose_ = initPose;
    raycast_pose_ = initPose;
    se_hip_config c{};
    c.width = inputSize.x(); c.height = inputSize.y();
    c.volume_resolution = volumeResolution.x(); c.volume_dimension = volumeDimensions.x();
    c.field_type = is_sdf() ? SE_HIP_FIELD_SDF : SE_HIP_FIELD_OFUSION;
    c.device = config.device; c.max_blocks = config.max_blocks;
    if (se_hip_create(&c, &h_) != SE_HIP_OK) throw std::runtime_error(std::string("DenseSLAMSystem: ") + se_hip_last_error());
  }
  ~DenseSLAMSystem() { se_hip_destroy(h_); }
  DenseSLAMSystem(const DenseSLAMSystem&) = delete;
  DenseSLAMSystem& operator=(const DenseSLAMSystem&) = delete;

  /* DenseSLAMSystem.h:147 / DenseSLAMSystem.cpp:128-141 */
  bool preprocessing(const unsigned short* inputDepth, const Eigen::Vector2i& inputSize, const bool filterInput) {
    // bilateralFilterKernel feeds only the tracking pyramid (scaled_depth_[0]); it runs inside se_hip_track
    return ok(se_hip_filter_depth(h_, filterInput ? 1 : 0)) && ok(se_hip_upload_depth_mm(h_, inputDepth, inputSize.x(), inputSize.y()));
  }
  /* the reference's float_depth_ handed over directly (metres) */
  bool preprocessing(const float* depthMetres) { return ok(se_hip_upload_depth(h_, depthMetres)); }

  /* DenseSLAMSystem.h:173 / DenseSLAMSystem.cpp:143-189: ICP against the last raycast, on the device */
  bool tracking(const Eigen::Vector4f& k, float icp_threshold, unsigned tracking_rate, unsigned frame) {
    const int r = se_hip_track(h_, k.data(), icp_threshold, tracking_rate, frame, iterations_.data(), (int32_t)iterations_.size(), pose_.data());
    ok(r);
    tracked_ = r > 0;
    return tracked_;
  }

  /* DenseSLAMSystem.h:193 / DenseSLAMSystem.cpp:206-268 */
  bool integration(const Eigen::Vector4f& k, unsigned int integration_rate, float mu, unsigned int frame) {
    const int r = se_hip_integrate(h_, pose_.data(), k.data(), integration_rate, mu, frame);
    ok(r);
    integrated_ = r > 0;
    return r > 0;
  }
  /* DenseSLAMSystem.h:212 / DenseSLAMSystem.cpp:191-204 */
  bool raycasting(const Eigen::Vector4f& k, float mu, unsigned int frame) {
    const int r = se_hip_raycast(h_, pose_.data(), k.data(), mu, frame);
    ok(r);
    if (r > 0) raycast_pose_ = pose_;
    return r > 0;
  }

  /* DenseSLAMSystem.h:241-286 / DenseSLAMSystem.cpp:274-300: RGBW images of the computation size */
  void renderVolume(unsigned char* out, const Eigen::Vector2i& outputSize, int frame, int raycast_rendering_rate,
                    const Eigen::Vector4f& k, float largestep) {
    (void)outputSize;
    ok(se_hip_render_volume(h_, out, viewPose_->data(), k.data(), mu_, largestep, (uint32_t)frame, (uint32_t)raycast_rendering_rate));
  }
  void renderTrack(unsigned char* out, const Eigen::Vector2i& outputSize) { (void)outputSize; ok(se_hip_render_track(h_, out)); }
  void renderDepth(unsigned char* out, const Eigen::Vector2i& outputSize) { (void)outputSize; ok(se_hip_render_depth(h_, out)); }

  /* DenseSLAMSystem.h:224 / DenseSLAMSystem.cpp:302-322: marching cubes of the map into a VTK file */
  void dump_mesh(const std::string filename) { ok(se_hip_dump_mesh(h_, filename.c_str())); }
  void setViewPose(Eigen::Matrix4f* value = NULL) { viewPose_ = value ? value : &pose_; }   /* DenseSLAMSystem.h:363-372 */
  Eigen::Matrix4f* getViewPose() { return viewPose_; }

  void getMap(MapSnapshot& out) {
    int nb = 0, nn = 0;
    if (!ok(se_hip_counts(h_, &nb, &nn))) return;
    out.n_blocks = nb; out.n_nodes = nn;
    out.coords.resize((size_t)nb * 3); out.x.resize((size_t)nb * 512); out.y.resize((size_t)nb * 512); out.active.resize(nb);
    if (nb) ok(se_hip_download_blocks(h_, out.coords.data(), out.x.data(), out.y.data(), out.active.data()));
  }
  /* vertex_ / normal_ of the last raycasting(): width*height packed xyz */
  bool getVertexNormal(std::vector<float>& vertex, std::vector<float>& normal) {
    const size_t n = (size_t)computation_size_.x() * computation_size_.y() * 3;
    vertex.resize(n); normal.resize(n);
    return ok(se_hip_download_vertex_normal(h_, vertex.data(), normal.data()));
  }

  bool getTracked() { return tracked_; }
  bool getIntegrated() { return integrated_; }
  Eigen::Vector3f getPosition() {
    return Eigen::Vector3f(pose_(0, 3) - init_pose_.x(), pose_(1, 3) - init_pose_.y(), pose_(2, 3) - init_pose_.z());
  }
  Eigen::Vector3f getInitPos() { return init_pose_; }
  Eigen::Matrix4f getPose() { return pose_; }
  /* DenseSLAMSystem.h:353-356: the translation is relative to the initial position */
  void setPose(const Eigen::Matrix4f pose) {
    pose_ = pose;
    pose_(0, 3) += init_pose_.x(); pose_(1, 3) += init_pose_.y(); pose_(2, 3) += init_pose_.z();
  }
  Eigen::Vector3f getModelDimensions() { return volume_dimension_; }
  Eigen::Vector3i getModelResolution() { return volume_resolution_; }
  Eigen::Vector2i getComputationResolution() { return computation_size_; }
  se_hip_pipeline* handle() { return h_; }

 private:
  static bool is_sdf();
  static Eigen::Matrix4f toMatrix4f(const Eigen::Vector3f& t) {  /* se_core/include/se/utils/math_utils.h:86-93 */
    Eigen::Matrix4f m = Eigen::Matrix4f::Identity();
    m(0, 3) = t.x(); m(1, 3) = t.y(); m(2, 3) = t.z();
    return m;
  }
  bool ok(int status) {
    if (status < 0) { std::cerr << "DenseSLAMSystem: " << se_hip_last_error() << std::endl; return false; }
    return true;
  }
  se_hip_pipeline* h_ = nullptr;
  Eigen::Vector2i computation_size_;
  Eigen::Vector3i volume_resolution_;
  Eigen::Vector3f volume_dimension_;
  Eigen::Vector3f init_pose_;
  Eigen::Matrix4f pose_, raycast_pose_;
  Eigen::Matrix4f* viewPose_ = &pose_;
  float mu_ = 0.1f;
  std::vector<int32_t> iterations_;
  bool tracked_ = false, integrated_ = false;
  friend void synchroniseDevices();
};

namespace se_hip_detail {
template <typename T> struct is_sdf_tag { static const bool value = false; };
template <> struct is_sdf_tag<SDF> { static const bool value = true; };
}  // namespace se_hip_detail
inline bool DenseSLAMSystem::is_sdf() { return se_hip_detail::is_sdf_tag<FieldType>::value; }

/* declared and never defined in the reference (DenseSLAMSystem.h:418) */
inline void synchroniseDevices(DenseSLAMSystem& s) { se_hip_sync(s.handle()); }

#endif /* SE_HIP_DENSESLAMSYSTEM_H */
