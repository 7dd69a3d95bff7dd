// TEST INFRASTRUCTURE (not part of the product library): exhaustive comparison, on the GPU, of the sweep's shared-reciprocal divisions and
// its range-restricted square root (se_rcp_refined / se_div_refined / se_inv_refined / se_sqrt_ge1, supereight_amd/csrc/se_kernels.h) with the
// operations the compiler emits for `/` and sqrtf -- over the operand ranges IntegArgs::fast_div promises, bit for bit.
// Built by __graft_entry__.build() (and on demand by tests/test_gpu_sweep_arith.py) with the library's own flags.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../supereight_amd/csrc/se_kernels.h"

namespace {
struct Bad { unsigned long long count; uint32_t a, b, got, want; };

__device__ void report(Bad* bad, uint32_t a, uint32_t b, uint32_t got, uint32_t want) {
  if (atomicAdd(&bad->count, 1ull) == 0ull) { bad->a = a; bad->b = b; bad->got = got; bad->want = want; }
}

// x / z and 1 / z: every mantissa of the numerator in the binades listed, for each divisor of the list.  grid.y = divisor, grid.x covers 2^23 mantissas.
// Compared where the sweep uses the quotient: through its square (the only use of x / z, y / z) and directly (1 / z).
__global__ void k_chk_div_z(const float* __restrict__ dens, const int* __restrict__ exps, int n_exps, Bad* bad) {
  const float d = dens[blockIdx.y];
  const SeRcp r = se_rcp_refined(d);
  const float inv = se_inv_refined(r), inv_ref = 1.f / d;
  if (blockIdx.x == 0 && threadIdx.x == 0 && __float_as_uint(inv) != __float_as_uint(inv_ref)) report(bad, 0x3F800000u, __float_as_uint(d), __float_as_uint(inv), __float_as_uint(inv_ref));
  for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < (1u << 23); m += gridDim.x * blockDim.x) {
    for (int e = 0; e < n_exps; ++e) {
      const uint32_t bits = ((uint32_t)(exps[e] + 127) << 23) | m;
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        const float n = __uint_as_float(bits | (sg ? 0x80000000u : 0u));
        const float q = se_div_refined(n, r), q_ref = n / d;
        const float s = 1 + sqf(q), s_ref = 1 + sqf(q_ref);
        // exact wherever the IEEE sequence need not rescale (|n| >= 2^-100, quotient exponent in range); always equal through the square
        const int ed = (int)((__float_as_uint(d) >> 23) & 255u) - 127;
        const bool in_range = exps[e] >= -100 && exps[e] - ed > -120 && exps[e] - ed < 96;   // every residual of the sequence is exact without rescaling, quotient normal
        if ((in_range && __float_as_uint(q) != __float_as_uint(q_ref)) || __float_as_uint(s) != __float_as_uint(s_ref))
          report(bad, __float_as_uint(n), __float_as_uint(d), __float_as_uint(q), __float_as_uint(q_ref));
      }
    }
  }
}

// fminf(1, diff / mu) for EVERY float `diff` the sweep can produce and use (+0 or |diff| >= 2^-100, and diff > -mu; +inf included), for each mu of the list
__global__ void k_chk_div_mu(const float* __restrict__ mus, Bad* bad) {
  const float mu = mus[blockIdx.y];
  const SeRcp r = se_rcp_refined(mu);
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
    const uint32_t bits = (uint32_t)i;
    const uint32_t mag = bits & 0x7FFFFFFFu;
    if (bits == 0x80000000u || (mag != 0u && mag < ((uint32_t)(-100 + 127) << 23))) continue;   // -0 and 0 < |diff| < 2^-100: cannot occur (se_kernels.h)
    const float diff = __uint_as_float(bits);
    if (!(diff > -mu)) continue;     // sdf_update looks at the quotient only under this test (kfusion/mapping_impl.hpp:49); NaN fails it too
    const float got = fminf(1.f, se_div_refined(diff, r)), want = fminf(1.f, diff / mu);
    if (__float_as_uint(got) != __float_as_uint(want)) report(bad, bits, __float_as_uint(mu), __float_as_uint(got), __float_as_uint(want));
  }
}

// sqrtf(s) for every float s >= 1, +inf, and the NaNs (compared as "is NaN")
__global__ void k_chk_sqrt(Bad* bad) {
  for (unsigned long long i = 0x3F800000ull + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i <= 0x7FFFFFFFull; i += (unsigned long long)gridDim.x * blockDim.x) {
    const float s = __uint_as_float((uint32_t)i);
    const float got = se_sqrt_ge1(s), want = sqrtf(s);
    const bool same = (got != got && want != want) || __float_as_uint(got) == __float_as_uint(want);
    if (!same) report(bad, (uint32_t)i, 0u, __float_as_uint(got), __float_as_uint(want));
  }
}
}  // namespace

// which: 0 = divisions by z (dens[n_dens], exps[n_exps]), 1 = diff / mu (dens = the mu values), 2 = square root.
// out[0] = mismatches, out[1..4] = operands and results of the first one.  Returns 0, or a negative HIP error.
extern "C" int se_arith_check(int which, const float* dens, int n_dens, const int* exps, int n_exps, uint64_t out[5]) {
  Bad* bad = nullptr;
  float* d_dens = nullptr;
  int* d_exps = nullptr;
  if (hipMalloc((void**)&bad, sizeof(Bad)) != hipSuccess) return -1;
  hipMemset(bad, 0, sizeof(Bad));
  if (n_dens > 0) { hipMalloc((void**)&d_dens, n_dens * sizeof(float)); hipMemcpy(d_dens, dens, n_dens * sizeof(float), hipMemcpyHostToDevice); }
  if (n_exps > 0) { hipMalloc((void**)&d_exps, n_exps * sizeof(int)); hipMemcpy(d_exps, exps, n_exps * sizeof(int), hipMemcpyHostToDevice); }
  if (which == 0) hipLaunchKernelGGL(k_chk_div_z, dim3(256, n_dens), dim3(256), 0, 0, d_dens, d_exps, n_exps, bad);
  else if (which == 1) hipLaunchKernelGGL(k_chk_div_mu, dim3(4096, n_dens), dim3(256), 0, 0, d_dens, bad);
  else hipLaunchKernelGGL(k_chk_sqrt, dim3(4096), dim3(256), 0, 0, bad);
  const hipError_t e = hipDeviceSynchronize();
  Bad h{};
  hipMemcpy(&h, bad, sizeof h, hipMemcpyDeviceToHost);
  out[0] = h.count; out[1] = h.a; out[2] = h.b; out[3] = h.got; out[4] = h.want;
  hipFree(bad); if (d_dens) hipFree(d_dens); if (d_exps) hipFree(d_exps);
  return e == hipSuccess ? 0 : -2;
}
