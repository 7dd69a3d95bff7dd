/* The Eigen types of the DenseSLAMSystem interface: the real ones when <Eigen/Dense> is installed, otherwise minimal
 * PODs with the same storage layout (column-major Matrix4f) and the few accessors the interface needs. */
#ifndef SE_HIP_EIGEN_PODS_H
#define SE_HIP_EIGEN_PODS_H

#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define SE_HIP_HAVE_EIGEN 1
#endif
#endif
#ifndef SE_HIP_HAVE_EIGEN
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

#endif /* SE_HIP_EIGEN_PODS_H */
