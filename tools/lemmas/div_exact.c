/* Exhaustive check of a 3-instruction correctly-rounded division by a run-time constant b:
 *     r = RN(1/b)          (once per b, on the host)
 *     q0 = RN(a * r);  rem = fma(-q0, b, a);  q = fma(rem, r, q0)
 * against the IEEE quotient RN(a / b), for EVERY finite float a (2^32 bit patterns), for the divisors the integration
 * sweep divides by: the weight + 1 (integers 1 .. 101, mapping_impl.hpp:57-58) and mu (kfusion band).  Prints, per divisor,
 * the number of numerators that differ and the range of |a| they lie in.  Groundwork for replacing two of the five IEEE
 * division sequences (11 VALU instructions each) of update_block / sdf_update; NOT used by the product yet.
 *   gcc -O2 -ffp-contract=off -fopenmp -o /tmp/div_exact tools/lemmas/div_exact.c -lm
 *   /tmp/div_exact binade   every divisor, one binade of numerators each (seconds)
 *   /tmp/div_exact          all 2^32 numerators for a sample of divisors (minutes);  /tmp/div_exact full: for all of 1..101 (hours) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static float f_of(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t u_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static void check(float b) {
  const float r = (float)(1.0 / (double)b);   /* RN(1/b): double division is exact enough to round correctly for these b */
  unsigned long long bad = 0;
  float lo = INFINITY, hi = 0.f;
  unsigned long long bad_normal_range = 0;    /* mismatches with 2^-100 <= |a| <= 2^100 */
#pragma omp parallel
  {
    unsigned long long my_bad = 0, my_badn = 0;
    float my_lo = INFINITY, my_hi = 0.f;
#pragma omp for schedule(static)
    for (long long i = 0; i < (1ll << 32); ++i) {
      const float a = f_of((uint32_t)i);
      if (!isfinite(a)) continue;
      const float ref = a / b;
      const float q0 = a * r;
      const float rem = fmaf(-q0, b, a);
      const float q = fmaf(rem, r, q0);
      if (u_of(q) != u_of(ref)) {
        ++my_bad;
        const float m = fabsf(a);
        if (m < my_lo) my_lo = m;
        if (m > my_hi) my_hi = m;
        if (m >= 0x1p-100f && m <= 0x1p100f) ++my_badn;
      }
    }
#pragma omp critical
    { bad += my_bad; bad_normal_range += my_badn; if (my_lo < lo) lo = my_lo; if (my_hi > hi) hi = my_hi; }
  }
  if (bad) printf("b = %-12.9g r = %a : %llu numerators differ, |a| in [%a, %a]; inside 2^-100..2^100: %llu\n", b, r, bad, lo, hi, bad_normal_range);
  else printf("b = %-12.9g r = %a : exact for every finite numerator\n", b, r);
  fflush(stdout);
}

/* One binade of numerators [1, 2) (2^23 mantissas, both signs): the three operations are homogeneous under scaling by
 * powers of two as long as nothing underflows or overflows, so this covers every exponent of the normal range at once. */
static unsigned long long check_binade(float b) {
  const float r = (float)(1.0 / (double)b);
  unsigned long long bad = 0;
  for (uint32_t m = 0; m < (1u << 23); ++m)
    for (int sgn = 0; sgn < 2; ++sgn) {
      const float a = f_of((sgn ? 0x80000000u : 0u) | 0x3F800000u | m);
      const float ref = a / b, q0 = a * r, rem = fmaf(-q0, b, a), q = fmaf(rem, r, q0);
      if (u_of(q) != u_of(ref)) ++bad;
    }
  return bad;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "binade")) {   /* every divisor 1 .. 101 and the mu values, one binade each: seconds */
    unsigned long long total = 0;
    for (int k = 1; k <= 101; ++k) { const unsigned long long n = check_binade((float)k); total += n; if (n) printf("b = %d: %llu differ\n", k, n); }
    const float mus2[] = {0.1f, 0.05f, 0.02f, 0.008f, 0.3f, 0.01f, 0.2f};
    for (unsigned i = 0; i < sizeof mus2 / sizeof *mus2; ++i) { const unsigned long long n = check_binade(mus2[i]); total += n; if (n) printf("b = %g: %llu differ\n", mus2[i], n); }
    printf("binade check, divisors 1..101 and 7 mu values: %llu mismatches\n", total);
    return total != 0;
  }
  const int full = argc > 1;   /* any argument: all of 1..101; default: a sample */
  if (full) for (int k = 1; k <= 101; ++k) check((float)k);
  else { const int ks[] = {1, 2, 3, 5, 7, 10, 33, 64, 100, 101}; for (unsigned i = 0; i < sizeof ks / sizeof *ks; ++i) check((float)ks[i]); }
  const float mus[] = {0.1f, 0.05f, 0.02f, 0.008f, 0.3f};
  for (unsigned i = 0; i < sizeof mus / sizeof *mus; ++i) check(mus[i]);
  return 0;
}
