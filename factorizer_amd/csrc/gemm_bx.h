// gemm_bx.h — operand splitting and product helpers of the split-bf16 MFMA GEMM family (gemm_bx.hip, gemm_bxk.hip).
#pragma once
#include "gemm_common.h"

namespace fz {

typedef __bf16 bx8 __attribute__((ext_vector_type(8)));
typedef __bf16 bx2 __attribute__((ext_vector_type(2)));
typedef float fx2 __attribute__((ext_vector_type(2)));

enum { BXPRO_NONE = 0, BXPRO_LN = 1, BXPRO_GELU = 2, BXPRO_BMUL = 3 };

// x[0..7] -> up to three bf16 levels, round-to-nearest at each (v_cvt_pk_bf16_f32 converts two floats)
template <int NT>
__device__ __forceinline__ void bx_split(const float (&x)[8], bx8 (&t)[NT]) {
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    fx2 v = {x[i], x[i + 1]};
    const bx2 a = __builtin_convertvector(v, bx2);
    t[0][i] = a[0]; t[0][i + 1] = a[1];
    if constexpr (NT >= 2) {
      v = v - __builtin_convertvector(a, fx2);
      const bx2 b = __builtin_convertvector(v, bx2);
      t[1][i] = b[0]; t[1][i + 1] = b[1];
      if constexpr (NT >= 3) {
        v = v - __builtin_convertvector(b, fx2);
        const bx2 c = __builtin_convertvector(v, bx2);
        t[2][i] = c[0]; t[2][i + 1] = c[1];
      }
    }
  }
}

// acc += Σ_{i + j <= max(NTA, NTB) - 1} a_i · b_j, smallest terms first
template <int NTA, int NTB>
__device__ __forceinline__ void bx_mfma(f32x16& acc, const bx8 (&a)[NTA], const bx8 (&b)[NTB]) {
  constexpr int L = (NTA > NTB ? NTA : NTB) - 1;
#pragma unroll
  for (int s = L; s >= 0; --s)
#pragma unroll
    for (int i = 0; i < NTA; ++i) {
      const int jj = s - i;
      if (jj >= 0 && jj < NTB) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[jj], acc, 0, 0, 0);
    }
}

// Terms per operand.  fp32 storage: three bf16 levels each (six products).  bf16 storage (mixed-precision mode): the
// column operand IS bf16 — one term, exact — unless a prologue has produced new fp32 values from it (LayerNorm's
// x - pivot, GELU): those keep two levels (16 significand bits) so that the fused prologue adds no rounding the unfused
// layer chain (which would store that tensor as bf16 once, after normalisation) does not have; weights keep two levels.
template <typename AT> struct BxTerms { static constexpr int A = 3; };
template <> struct BxTerms<bf16> { static constexpr int A = 2; };
template <typename AT>
__host__ __device__ constexpr int bx_terms_b(int pro) {
  return sizeof(AT) == 4 ? 3 : ((pro == BXPRO_LN || pro == BXPRO_GELU) ? 2 : 1);
}

// K-split form (gemm_bxk.hip); nacc, mb: tile; returns FZ_OK or an error
template <typename AT>
int gemm_bxk_launch(const GemmArgsT<AT>& a, int loader, int epilogue, int pro, int nacc, int mb, fz_stream_t stream);

}  // namespace fz
