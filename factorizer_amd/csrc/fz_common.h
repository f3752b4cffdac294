// fz_common.h — shared host/device helpers for libfactorizer_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/factorizer_hip.h"

namespace fz {

// ---- error plumbing (thread-local message, integer status) ------------------------------
std::string& last_error();
std::atomic<int64_t>& launch_counter();

inline int fail(int code, const char* msg) {
  last_error() = msg;
  return code;
}

#define FZ_HIP_OK(expr)                                                             \
  do {                                                                              \
    hipError_t e__ = (expr);                                                        \
    if (e__ != hipSuccess) {                                                        \
      fz::last_error() = std::string(#expr) + ": " + hipGetErrorString(e__);        \
      return FZ_E_HIP;                                                              \
    }                                                                               \
  } while (0)

#define FZ_LAUNCH_CHECK()                  \
  do {                                     \
    fz::launch_counter().fetch_add(1);     \
    FZ_HIP_OK(hipGetLastError());          \
  } while (0)

#if defined(__HIPCC__)
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): branch-free, 1 rcp + 1 exp + 5 fma —
// the libm erff costs ~4x more VALU and carries a branch per element.  Used for the exact-erf
// GELU of the MLP (layers/mlp.py:56); the 1e-4 parity budget dwarfs its error.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);  // 1 ulp; IEEE division is ~10 VALU ops
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * __expf(-ax * ax);
  return x < 0.f ? -r : r;
}

// ---- wave64 all-lanes sum on the DPP network (no LDS traffic) ----------------------------
// xor-1, xor-2 (quad_perm), half-mirror, mirror give every lane its 16-lane row total;
// row_bcast15/31 fold the four rows into lane 63; readlane makes it a scalar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_take<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_take<0x140, 0xf>(v);  // row_mirror
  v += dpp_take<0x142, 0xa>(v);  // row_bcast15 into rows 1,3
  v += dpp_take<0x143, 0xc>(v);  // row_bcast31 into rows 2,3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#endif

}  // namespace fz
