// rgs_kernels_n.hip -- every kernel instantiation for one team size n = GSIM_N (make builds kern_n1.o ..
// kern_n7.o from this file in parallel).  gfx950 only.
#include <hip/hip_runtime.h>

#include <atomic>

#include "gsim_kernels.hpp"

#ifndef GSIM_N
#error "compile with -DGSIM_N=<1..7>"
#endif

using namespace gsim;

namespace {

constexpr int N = GSIM_N;

template <int MODE, bool REC = false>
hipError_t launch_hybrid_mode(const StepArgs &args, hipStream_t stream) {
  const long long blocks = (args.B + 63) / 64;
  const size_t lds = sizeof(double) * WaveLayout<N>::DOUBLES * HybridGroups<N>::GPW;
  hipLaunchKernelGGL((k_step_hybrid<N, MODE, REC>), dim3((unsigned)blocks), dim3(64), lds, stream, args);
  return hipGetLastError();
}

template <int MODE, bool ASYNC>
hipError_t launch_pool_mode(const StepArgs &args, hipStream_t stream) {
  const long long blocks = (args.B + 64 * kPoolWaves - 1) / (64 * kPoolWaves);
  const size_t lds = sizeof(double) * WaveLayout<N>::DOUBLES * HybridGroups<N>::GPW * kPoolWaves + sizeof(PoolQueue);
  const void *fn = ASYNC ? (const void *)k_step_apool<N, MODE> : (const void *)k_step_pool<N, MODE>;
  // > 64 KB of dynamic LDS has to be asked for once per instantiation AND per device (the attribute belongs to the
  // device's function object); one bit per device, set after the call succeeded -- a second thread racing here at
  // worst repeats the (idempotent) call.
  static std::atomic<unsigned long long> attr_set{0ull};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (dev >= 64 || !(attr_set.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (dev < 64) attr_set.fetch_or(bit, std::memory_order_release);
  }
  if (ASYNC) hipLaunchKernelGGL((k_step_apool<N, MODE>), dim3((unsigned)blocks), dim3(64 * kPoolWaves), lds, stream, args);
  else hipLaunchKernelGGL((k_step_pool<N, MODE>), dim3((unsigned)blocks), dim3(64 * kPoolWaves), lds, stream, args);
  return hipGetLastError();
}

template <int MODE>
hipError_t launch_hybrid2_mode(const StepArgs &args, hipStream_t stream) {
  constexpr bool SPLITV = Hybrid2Split<N>::value;
  const long long blocks = (args.B + 31) / 32;
  const size_t lds = sizeof(double) * GridView<N, SPLITV>::LDS_DOUBLES * HybridGroups<N, SPLITV, Hybrid2Split<N>::LDS_KB>::GPW;
  hipLaunchKernelGGL((k_step_hybrid2<N, MODE>), dim3((unsigned)blocks), dim3(64), lds, stream, args);
  return hipGetLastError();
}

template <int FAST>
hipError_t launch_wave_rng(const StepArgs &args, hipStream_t stream) {
  const long long blocks = (args.B + kWavesPerBlock - 1) / kWavesPerBlock;
  const size_t lds = sizeof(double) * ((WaveLayout<N>::DOUBLES + rng_window<N>()) * kWavesPerBlock + kLogTabDoubles);
  static std::atomic<unsigned long long> attr_set{0ull};      // > 64 KB of dynamic LDS at the large team sizes: per device, see launch_pool_mode
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (lds > 65536 && (dev >= 64 || !(attr_set.load(std::memory_order_acquire) & bit))) {
    e = hipFuncSetAttribute((const void *)k_step_wave_rng<N, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (dev < 64) attr_set.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((k_step_wave_rng<N, FAST>), dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds, stream, args);
  return hipGetLastError();
}

template <int FAST, int NOISE>
hipError_t launch_wave_level(const StepArgs &args, hipStream_t stream) {
  const long long blocks = (args.B + kWavesPerBlock - 1) / kWavesPerBlock;
  const size_t lds = sizeof(double) * WaveLayout<N>::DOUBLES * kWavesPerBlock;
  hipLaunchKernelGGL((k_step_wave<N, 64, FAST, NOISE>), dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds, stream, args);
  return hipGetLastError();
}

hipError_t step(const StepArgs &args, hipStream_t stream, int kind, int mode) {
  switch (kind) {
    case LAUNCH_WAVE:
      return mode == 3 ? launch_wave_level<3, 0>(args, stream)
             : mode == 1 ? launch_wave_level<1, 0>(args, stream) : launch_wave_level<0, 0>(args, stream);
    case LAUNCH_WAVE_NOISE:
      return launch_wave_level<0, 1>(args, stream);
    case LAUNCH_WAVE_RNG:
      return mode == 3 ? launch_wave_rng<3>(args, stream) : mode == 1 ? launch_wave_rng<1>(args, stream) : launch_wave_rng<0>(args, stream);
    case LAUNCH_HYBRID:
      return mode == DIV_POW2_ONESTEP ? launch_hybrid_mode<DIV_POW2_ONESTEP>(args, stream)
             : mode == DIV_PLAYER_POW2 ? launch_hybrid_mode<DIV_PLAYER_POW2>(args, stream)
             : mode == DIV_MARKSTEIN ? launch_hybrid_mode<DIV_MARKSTEIN>(args, stream)
                                     : launch_hybrid_mode<DIV_IEEE>(args, stream);
    case LAUNCH_HYBRID_REC:
      return mode == DIV_POW2_ONESTEP ? launch_hybrid_mode<DIV_POW2_ONESTEP, true>(args, stream)
             : mode == DIV_PLAYER_POW2 ? launch_hybrid_mode<DIV_PLAYER_POW2, true>(args, stream)
             : mode == DIV_MARKSTEIN ? launch_hybrid_mode<DIV_MARKSTEIN, true>(args, stream)
                                     : launch_hybrid_mode<DIV_IEEE, true>(args, stream);
    case LAUNCH_POOL:
      return mode == DIV_POW2_ONESTEP ? launch_pool_mode<DIV_POW2_ONESTEP, false>(args, stream)
             : mode == DIV_PLAYER_POW2 ? launch_pool_mode<DIV_PLAYER_POW2, false>(args, stream)
             : mode == DIV_MARKSTEIN ? launch_pool_mode<DIV_MARKSTEIN, false>(args, stream)
                                     : launch_pool_mode<DIV_IEEE, false>(args, stream);
    case LAUNCH_APOOL:
      return mode == DIV_POW2_ONESTEP ? launch_pool_mode<DIV_POW2_ONESTEP, true>(args, stream)
             : mode == DIV_PLAYER_POW2 ? launch_pool_mode<DIV_PLAYER_POW2, true>(args, stream)
             : mode == DIV_MARKSTEIN ? launch_pool_mode<DIV_MARKSTEIN, true>(args, stream)
                                     : launch_pool_mode<DIV_IEEE, true>(args, stream);
    case LAUNCH_HYBRID2:
      return mode == DIV_POW2_ONESTEP ? launch_hybrid2_mode<DIV_POW2_ONESTEP>(args, stream)
             : mode == DIV_PLAYER_POW2 ? launch_hybrid2_mode<DIV_PLAYER_POW2>(args, stream)
             : mode == DIV_MARKSTEIN ? launch_hybrid2_mode<DIV_MARKSTEIN>(args, stream)
                                     : launch_hybrid2_mode<DIV_IEEE>(args, stream);
  }
  return hipErrorInvalidValue;
}

hipError_t reset(double *state, double *tcur, double *damage, const double *x0, int x0_batched, long long B,
                 hipStream_t stream) {
  const int tb = 256;
  hipLaunchKernelGGL((k_reset<N>), dim3((unsigned)((B + tb - 1) / tb)), dim3(tb), 0, stream, state, tcur, damage, x0,
                     x0_batched, B);
  return hipGetLastError();
}

size_t vgrid_doubles(long long B) {
  constexpr bool SPLITV = Hybrid2Split<N>::value;
  if (!SPLITV) return 0;
  const long long blocks = (B + 31) / 32;
  return (size_t)blocks * HybridGroups<N, SPLITV, Hybrid2Split<N>::LDS_KB>::GPW * GridView<N, SPLITV>::GV_DOUBLES;
}

const GsimOps kOps = {step, reset, vgrid_doubles};

}  // namespace

#define GSIM_CAT2(a, b) a##b
#define GSIM_CAT(a, b) GSIM_CAT2(a, b)
extern "C" const gsim::GsimOps *GSIM_CAT(gsim_ops_n, GSIM_N)() { return &kOps; }
