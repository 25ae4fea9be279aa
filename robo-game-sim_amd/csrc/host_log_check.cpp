// Build-time check of gsim_hostlog.hpp + host_log_tables.h against the host's own log(): run by tools/glibc_log_tables.py.
// Exit code 0: bit-identical on every argument tried.  g++ -O2 -mfma -ffp-contract=off host_log_check.cpp
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "gsim_hostlog.hpp"

#if !GSIM_HOST_LOG
int main() { return 2; }
#else
static double (*volatile libm_log)(double) = static_cast<double (*)(double)>(std::log);   // no constant folding, no builtin

static uint64_t bits(double v) { uint64_t u; std::memcpy(&u, &v, 8); return u; }

int main() {
  uint64_t s = 0x9E3779B97F4A7C15ull, bad = 0, n = 0;
  auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  auto check = [&](double x) {
    if (!(x > 0.0) || !std::isfinite(x) || x < 2.3e-308) return;
    ++n;
    if (bits(gsim::host_log(x, gsim::kHostLogTab)) != bits(libm_log(x))) { if (++bad <= 5) std::fprintf(stderr, "host_log(%a) = %a, log = %a\n", x, gsim::host_log(x, gsim::kHostLogTab), libm_log(x)); }
  };
  for (int i = 0; i < 10000000; ++i) check((double)(next() >> 11) * 0x1p-53);                 // uniform in (0, 1): what the polar method passes
  for (int i = 0; i < 10000000; ++i) check(0.9375 + (double)(next() >> 11) * 0x1p-53 * 0.13);  // the branch around 1
  for (int i = 0; i < 10000000; ++i) { double x; uint64_t u = (next() >> 2) & 0x7FEFFFFFFFFFFFFFull; std::memcpy(&x, &u, 8); check(x); }   // any exponent
  for (int e = -1000; e <= 1000; ++e) for (int m = 0; m < 130; ++m) check(std::ldexp(1.0 + m / 128.0, e));   // subinterval edges
  const double edge[] = {1.0, 0.9375, 0x1.dffffffffffffp-1, 1.0 + 0x1.09p-4, 0x1.108ffffffffffp0, 0x1.fffffffffffffp-1, 0x1.0000000000001p0};
  for (double x : edge) check(x);
  std::printf("host_log_check: %llu arguments, %llu differ\n", (unsigned long long)n, (unsigned long long)bad);
  return bad == 0 ? 0 : 1;
}
#endif
