// fz_common.h — shared host/device helpers for libfactorizer_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/factorizer_hip.h"

namespace fz {

// ---- error plumbing (thread-local message, integer status) ------------------------------
std::string& last_error();
std::atomic<int64_t>& launch_counter();
// Walking order of the NEXT launches of this thread's streaming kernels over their column / patch tiles: 0 ascending, 1
// descending (fz_set_tile_order, include/factorizer_hip.h).  A launch that walks a tensor in the direction opposite to the
// launch that produced it starts where the 256 MiB Infinity Cache still holds that tensor.
int& tile_order_ref();
inline int tile_order() { return tile_order_ref() ? 1 : 0; }

inline int fail(int code, const char* msg) {
  last_error() = msg;
  return code;
}

// Diagnostic environment knobs exist only in PROBE builds of the library (-DFZ_PROBE: tools/probes/build_alt.py builds a
// second .so, loaded through FZ_LIB_PATH for same-box A/B runs).  The shipped library reads ONE environment variable,
// FZ_GEMM_BX (the process default of the `products` descriptor field; both settings are valid results), and nothing else:
// FZ_KNOB(name) is a compile-time "unset" there, so every tuning / A-B branch folds to its default.  In a probe build a knob
// is read once per process, at the first launch that consults it (a function-local static at the call site).
struct EnvKnob { bool set; int val; const char* str; };
#ifdef FZ_PROBE
inline EnvKnob env_knob(const char* name) {
  const char* e = getenv(name);
  return EnvKnob{e != nullptr, e ? atoi(e) : 0, e};
}
#define FZ_KNOB(name) ([]() -> const fz::EnvKnob& { static const fz::EnvKnob k_ = fz::env_knob(name); return k_; }())
#else
#define FZ_KNOB(name) (fz::EnvKnob{false, 0, nullptr})
#endif

// fp32 products of a layer: the descriptor's `products` field, or the process default (fz_gemm_bx_enable) when it is unset
inline bool products_split(int field) {
  return field == FZ_PRODUCTS_SPLIT_BF16 ? true : (field == FZ_PRODUCTS_FP32_MFMA ? false : fz_gemm_bx_enable(-1) != 0);
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
// ---- activation storage types -----------------------------------------------------------------
// Mixed precision (BASELINE configs[4]): activations and their gradients may be STORED
// as bf16 in HBM; every kernel converts to fp32 on load and rounds (RNE, v_cvt_pk_bf16_f32) on store,
// so all arithmetic — MFMA accumulation, LayerNorm statistics, and the NMF factors / Grams / eps of
// matrix_factorization.py:200,236 in particular — stays fp32.  Parameters, statistics and weight
// gradients are always fp32.  Kernels are templates on the storage type AT in {float, bf16}.
typedef __bf16 bf16;
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16v4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef float f32v2 __attribute__((ext_vector_type(2)));
typedef float f32v4 __attribute__((ext_vector_type(4)));
typedef float f32v8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float aget(const float* p) { return *p; }
__device__ __forceinline__ float aget(const bf16* p) { return (float)*p; }
__device__ __forceinline__ void aput(float* p, float v) { *p = v; }
__device__ __forceinline__ void aput(bf16* p, float v) { *p = (bf16)v; }

// N consecutive elements <-> N floats (N = 1, 2, 4, 8); the address must be N-element aligned
template <int N>
__device__ __forceinline__ void aload(const float* p, float (&v)[N]) {
  if constexpr (N == 8) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else if constexpr (N == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (N == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = *p;
  }
}
template <int N>
__device__ __forceinline__ void aload(const bf16* p, float (&v)[N]) {
  if constexpr (N == 8) {
    const f32v8 f = __builtin_convertvector(*reinterpret_cast<const bf16v8*>(p), f32v8);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f[i];
  } else if constexpr (N == 4) {
    const f32v4 f = __builtin_convertvector(*reinterpret_cast<const bf16v4*>(p), f32v4);
    v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
  } else if constexpr (N == 2) {
    const f32v2 f = __builtin_convertvector(*reinterpret_cast<const bf16v2*>(p), f32v2);
    v[0] = f[0]; v[1] = f[1];
  } else {
    v[0] = (float)*p;
  }
}
template <int N>
__device__ __forceinline__ void astore(float* p, const float (&v)[N]) {
  if constexpr (N == 8) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else if constexpr (N == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (N == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
    *p = v[0];
  }
}
template <int N>
__device__ __forceinline__ void astore(bf16* p, const float (&v)[N]) {
  if constexpr (N == 8) {
    f32v8 f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = v[i];
    *reinterpret_cast<bf16v8*>(p) = __builtin_convertvector(f, bf16v8);
  } else if constexpr (N == 4) {
    f32v4 f = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<bf16v4*>(p) = __builtin_convertvector(f, bf16v4);
  } else if constexpr (N == 2) {
    f32v2 f = {v[0], v[1]};
    *reinterpret_cast<bf16v2*>(p) = __builtin_convertvector(f, bf16v2);
  } else {
    *p = (bf16)v[0];
  }
}

// 4 consecutive activation elements <-> float4
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16* p) {
  const f32v4 f = __builtin_convertvector(*reinterpret_cast<const bf16v4*>(p), f32v4);
  return make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16* p, const float4& v) {
  const f32v4 f = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<bf16v4*>(p) = __builtin_convertvector(f, bf16v4);
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): branch-free, 1 rcp + 1 exp + 5 fma —
// the libm erff costs ~4x more VALU and carries a branch per element.  Used for the exact-erf
// GELU of the MLP (layers/mlp.py:56); the 1e-4 parity budget dwarfs its error.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);  // 1 ulp; IEEE division is ~10 VALU ops
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * __expf(-ax * ax);
  return __builtin_copysignf(r, x);   // one v_bfi_b32 (r = erf(|x|) >= 0 up to one rounding at x = 0: |r| there is < 2e-7 either way)
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

// ---- EIGHT wave64 sums at once (v[i] <- all-lanes total of v[i], uniform) ---------------------------
// A multi-value butterfly: every exchange stage halves the number of live registers instead of folding
// each value separately, so 8 sums cost 18 vector ops + 8 readlanes (8 independent wave_sum calls: 48
// dependent DPP adds, each behind a VALU->DPP wait state, + 8 readlanes).  The NMF wave program spends a
// third of its VALU instructions in such reductions (X·V, gY·V, X·ga: 8 rows at a time).
//   distance 32: v_permlane32_swap (v[k] | v[k+4])  -> 4 registers, halves hold 2 values
//   distance 16: v_permlane16_swap                  -> 2 registers, 16-lane rows hold 4 values each
//   distance  8: select + row_ror:8                 -> 1 register, 8-lane groups hold the 8 values
//   distance 1, 2, 4: quad_perm / row_half_mirror   -> every lane of group i holds total i (lane 8i read back)
// Order of the additions per value: x[j]+x[j+32] ; +[j+16] ; +[j+8] ; xor 1 ; xor 2 ; half-mirror
// (tests/emul/emul.cpp mirrors it).
typedef unsigned fz_u32x2 __attribute__((ext_vector_type(2)));
// The same eight totals LEFT DISTRIBUTED: every lane of the 8-lane group i (lanes 8i .. 8i + 7) holds total i.  The NMF
// wave program continues on them in that form (one factor row per lane group, nmf_core.h) instead of broadcasting each
// total to all lanes and repeating the row arithmetic 64 times.
__device__ __forceinline__ float wave_sum8_dist(const float (&v)[8], int lane) {
  float y[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const fz_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 4]), false, false);
    y[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  float z[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const fz_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(y[k]), __float_as_uint(y[k + 2]), false, false);
    z[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  const bool hi8 = (lane & 8) != 0;
  const float keep = hi8 ? z[1] : z[0], give = hi8 ? z[0] : z[1];
  float w = keep + dpp_take<0x128, 0xf>(give);  // row_ror:8: lane l takes lane l ^ 8 of its row
  w += dpp_take<0xB1, 0xf>(w);                  // quad_perm [1,0,3,2]
  w += dpp_take<0x4E, 0xf>(w);                  // quad_perm [2,3,0,1]
  w += dpp_take<0x141, 0xf>(w);                 // row_half_mirror
  return w;
}
// total over the eight 8-lane groups of a value that is constant inside each group (distances 8, 16, 32), in every lane
__device__ __forceinline__ float wave_group_sum(float v) {
  v += dpp_take<0x128, 0xf>(v);                 // row_ror:8
  const fz_u32x2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const fz_u32x2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// ---- the same reductions over the 32 lanes of each wave HALF (two independent problems per wave: nmf_pcf.hip) ----
// total of v over the lane's half (rows 0-1 resp. 2-3 of the DPP network), in every lane of that half
__device__ __forceinline__ float half_sum_all(float v) {
  v += dpp_take<0xB1, 0xf>(v);   // xor 1
  v += dpp_take<0x4E, 0xf>(v);   // xor 2
  v += dpp_take<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_take<0x140, 0xf>(v);  // row_mirror -> 16-lane row totals
  const fz_u32x2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);   // row + its neighbour (lane ^ 16)
}
// EIGHT half totals at once, left distributed: every lane of the 4-lane group i = (lane >> 2) & 7 of a half holds total i
// of THAT half.  Multi-value butterfly as in wave_sum8_dist, one stage shorter:
//   distance 16: v_permlane16_swap (v[k] | v[k+4]) -> 4 registers; distance 8: select + row_ror:8 -> 2; distance 4:
//   select + row_half_mirror (lane i <-> 7 - i of its 8-lane group: the partner is always in the other quad) -> 1;
//   distances 1, 2: quad_perm.  19 vector operations for the sixteen sums of a wave.
__device__ __forceinline__ float half_sum8_dist(const float (&v)[8], int lane) {
  float y[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const fz_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 4]), false, false);
    y[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  const bool hi8 = (lane & 8) != 0;
  float z[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float keep = hi8 ? y[k + 2] : y[k], give = hi8 ? y[k] : y[k + 2];
    z[k] = keep + dpp_take<0x128, 0xf>(give);   // row_ror:8
  }
  const bool hi4 = (lane & 4) != 0;
  const float keep = hi4 ? z[1] : z[0], give = hi4 ? z[0] : z[1];
  float w = keep + dpp_take<0x141, 0xf>(give);  // row_half_mirror
  w += dpp_take<0xB1, 0xf>(w);                  // quad_perm [1,0,3,2]
  w += dpp_take<0x4E, 0xf>(w);                  // quad_perm [2,3,0,1]
  return w;
}
// the value lane `src` (0..31) of the lane's own half holds, in every lane (LDS crossbar, no LDS memory)
__device__ __forceinline__ float half_take(float d, int lane, int src) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(((lane & 32) + src) << 2, __float_as_int(d)));
}
__device__ __forceinline__ void wave_sum8(float (&v)[8], int lane) {
  float y[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const fz_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 4]), false, false);
    y[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  float z[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const fz_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(y[k]), __float_as_uint(y[k + 2]), false, false);
    z[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  const bool hi8 = (lane & 8) != 0;
  const float keep = hi8 ? z[1] : z[0], give = hi8 ? z[0] : z[1];
  float w = keep + dpp_take<0x128, 0xf>(give);  // row_ror:8: lane l takes lane l ^ 8 of its row
  w += dpp_take<0xB1, 0xf>(w);                  // quad_perm [1,0,3,2]
  w += dpp_take<0x4E, 0xf>(w);                  // quad_perm [2,3,0,1]
  w += dpp_take<0x141, 0xf>(w);                 // row_half_mirror
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), 8 * i));
}
#endif

}  // namespace fz
