// gsim_hostlog.hpp -- the HOST's log(double), restated so that the device draws the reference's normals bit for bit.
//
// The reference's noise is std::normal_distribution<double> (include/GameSim.h:73), whose polar method calls std::log
// (libstdc++ random.tcc:1826), i.e. the C library's log() of the machine the reference runs on.  In this image that is
// glibc 2.35 (NOT part of /root/reference): sysdeps/ieee754/dbl-64/e_log.c, Szabolcs Nagy's table-driven log from the ARM
// optimized routines, in the variant its ifunc selects on CPUs with FMA and AVX2 (__log_fma).  Restated here from the
// published algorithm and the order of operations of that variant:
//
//   x in [1 - 2^-4, 1 + 0x1.09p-4):  r = x - 1;  log(x) = r - r^2/2 + r^3 * P(r), the first two terms in double-double
//                                     (r split at 2^27), P of degree 10 (B[1..10], B[0] = -1/2);
//   otherwise:                        x = 2^k z, z in [0x1.6p-1, 0x1.6p0) cut into 128 subintervals i with centres c_i:
//                                     r = fma(z, 1/c_i, -1),  log(x) = k ln2 + log(c_i) + log1p(r), ln2 = hi + lo, the
//                                     polynomial of degree 5 in r (A[0..4]).
//
// The two tables (1/c_i and log c_i, chosen by the authors so that the rounding errors are small: not reproducible from a
// formula) and the coefficients are DATA of the installed libm.so.6: tools/glibc_log_tables.py reads them from that file
// at build time (located by the ln2hi / ln2lo pair that heads them) into host_log_tables.h, and csrc/host_log_check.cpp
// -- this very header compiled for the CPU -- must agree with the host's log() bit for bit on 3e7 arguments before the
// build accepts the tables.  If it does not (another libm, another CPU variant), GSIM_HOST_LOG stays 0, the generator
// uses the device library's log and the noise mode is statistical as before.
//
// Only finite, positive, normal x (what the polar method passes: 0 < r2 <= 1): no special cases.
#pragma once
#ifndef GSIM_HOST_LOG_TABLES_H                               // (the build step checks candidate headers by their full path)
#define GSIM_HOST_LOG_TABLES_H "host_log_tables.h"
#endif
#include GSIM_HOST_LOG_TABLES_H
#ifdef GSIM_FORCE_DEVICE_LOG                                 // A/B and fall-back testing: build as if no table had been accepted
#undef GSIM_HOST_LOG
#define GSIM_HOST_LOG 0
#endif

#if GSIM_HOST_LOG
#ifdef __HIPCC__
#define GSIM_HLOG_FN __device__ __forceinline__
#else
#define GSIM_HLOG_FN inline
#endif

namespace gsim {

// tab: the 128 {1/c, log c} pairs (kHostLogTab on the host; on the device a copy of kHostLogTabDev in LDS -- 64 lanes
// look up 64 different entries)
GSIM_HLOG_FN double host_log(double x, const double *tab) {
  using namespace hostlog;
  unsigned long long ix;
  __builtin_memcpy(&ix, &x, 8);
  if (ix - 0x3FEE000000000000ull <= 0x308FFFFFFFFFFull) {                      // 1 - 2^-4 <= x < 1 + 0x1.09p-4
    if (ix == 0x3FF0000000000000ull) return 0.0;
    const double r = x - 1.0;
    const double r2 = r * r;
    const double p2 = __builtin_fma(r2, B[3], __builtin_fma(r, B[2], B[1]));
    const double p3 = __builtin_fma(r2, B[6], __builtin_fma(r, B[5], B[4]));
    const double r3 = r * r2;
    double t = __builtin_fma(r3, B[10], __builtin_fma(r2, B[9], __builtin_fma(r, B[8], B[7])));
    t = __builtin_fma(t, r3, p3);
    t = __builtin_fma(t, r3, p2);
    const double rhi = __builtin_fma(-0x1p27, r, __builtin_fma(r, 0x1p27, r));   // r + w - w with w = r * 2^27, as that variant fuses it
    const double rhi2 = rhi * rhi;
    const double rlo = r - rhi;
    const double hi = __builtin_fma(rhi2, B[0], r);
    double lo = __builtin_fma(rhi2, B[0], r - hi);
    lo = __builtin_fma(B[0] * rlo, r + rhi, lo);
    return __builtin_fma(t, r3, lo) + hi;
  }
  const unsigned long long tmp = ix - 0x3FE6000000000000ull;
  const int i = (int)((tmp >> 45) & 127ull);
  const int k = (int)((long long)tmp >> 52);
  const unsigned long long iz = ix - (tmp & 0xFFF0000000000000ull);
  double z;
  __builtin_memcpy(&z, &iz, 8);
  const double invc = tab[2 * i], logc = tab[2 * i + 1];
  const double r = __builtin_fma(z, invc, -1.0);
  const double kd = (double)k;
  const double w = __builtin_fma(kd, Ln2hi, logc);
  const double hi = r + w;
  const double r2 = r * r;
  double lo = (w - hi) + r;
  lo = __builtin_fma(kd, Ln2lo, lo);
  lo = __builtin_fma(r2, A[0], lo);
  const double q = __builtin_fma(__builtin_fma(r, A[4], A[3]), r2, __builtin_fma(r, A[2], A[1]));
  return __builtin_fma(r * r2, q, lo) + hi;
}

}  // namespace gsim
#endif
