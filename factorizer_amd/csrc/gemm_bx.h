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

// ---- "LayerNorm + Linear(32 -> 32, no bias) + ReLU" on a 32-channel tile that sits in MFMA ACCUMULATOR layout ------------------
// The first layer of a FactorizerBlock, t = relu(in_proj(LN1(x))) (factorizer.py:38,44,75; norm.py:29-34), applied by the kernel
// that PRODUCES the block input x while the tile is in registers: the launch that would read x back (fz_gemm: LayerNorm + in_proj)
// disappears.  Layout: register r of lane (j, h) = channel (r & 3) + 8 (r >> 2) + 4 h of the lane's NQ voxels.
//   Aw: LDS image of W·diag(γ) pre-split for that operand order, [g (2)][level (3)][lane (64)] x 16 B (ln_inproj_stage)
//   y : the tile (bias added; bf16 storage: ALREADY rounded to the stored value — what a separate launch would read)
//   out: mu, rs (exact two-pass statistics of each voxel) and tacc = W·(γ ∘ x̂) — the caller adds tw[row] = (W β)[row], applies
//   the ReLU and stores.
struct BlockPrologueArgs {
  const float* ln_g;   // (32)
  const float* ln_b;   // (32)
  float ln_eps;
  const float* w;      // (32, 32) in_proj weight W[m][k]
  void* t;             // (B, 32, V) activation storage type of the launch
  float* stats;        // (B, 2, V): mean | rstd
};

__device__ __forceinline__ void ln_inproj_stage(__bf16* Aw, float* tw, const BlockPrologueArgs& a, int tid, int nthreads) {
  for (int it = tid; it < 128; it += nthreads) {
    const int l = it & 63, g = it >> 6;
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 8 * g + e;
      const int kk = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      wv[e] = a.w[(l & 31) * 32 + kk] * a.ln_g[kk];
    }
    bx8 t3[3];
    bx_split<3>(wv, t3);
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<bx8*>(Aw + ((g * 3 + i) * 64 + l) * 8) = t3[i];
  }
  for (int r = tid; r < 32; r += nthreads) {
    float s = 0.f;
    for (int k = 0; k < 32; ++k) s += a.w[r * 32 + k] * a.ln_b[k];
    tw[r] = s;
  }
}

template <int NQ>
__device__ __forceinline__ void ln_inproj_tile(float (&y)[NQ][16], const __bf16* Aw, float eps, int lane, float (&mu)[NQ],
                                               float (&rs)[NQ], f32x16 (&tacc)[NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += y[q][r];
    s += __shfl_xor(s, 32, 64);
    mu[q] = s / 32.0f;
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = y[q][r] - mu[q];
      v += d * d;
    }
    v += __shfl_xor(v, 32, 64);
    rs[q] = 1.0f / sqrtf(v / 32.0f + eps);
#pragma unroll
    for (int r = 0; r < 16; ++r) y[q][r] = (y[q][r] - mu[q]) * rs[q];
#pragma unroll
    for (int r = 0; r < 16; ++r) tacc[q][r] = 0.f;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    bx8 aop[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) aop[i] = *reinterpret_cast<const bx8*>(Aw + ((g * 3 + i) * 64 + lane) * 8);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float x8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x8[e] = y[q][8 * g + e];
      bx8 bop[3];
      bx_split<3>(x8, bop);
      bx_mfma<3, 3>(tacc[q], aop, bop);
    }
  }
}

// Terms per operand.  fp32 storage: three bf16 levels each (six products).  bf16 storage (mixed-precision mode): the
// column operand IS bf16 — one term, exact — unless a prologue has produced new fp32 values from it (LayerNorm's
// x - pivot, GELU): those are split in three levels like any fp32 value; the fp32 weights always are.  So the mode differs
// from fp32 ONLY by the storage roundings of activations: every product is fp32-accurate before the one rounding of the
// stored result.  (A first version kept two levels — 16 significand bits — for weights and prologue values: 2^-17
// relative per product is 256x below the storage rounding, but it moves pre-rounding values enough to flip the bf16
// rounding of ~0.4 % of a stored tensor's elements, and the gradient through ten rank-2 HALS sweeps amplifies one
// flipped element of the NMF input 35x: tests/test_gpu_bf16.py::test_block_cfg5_bf16_rank2_t10 moved from 0.74 % to
// 1.72 % of max|gx| against the storage-emulating oracle.)
template <typename AT> struct BxTerms { static constexpr int A = 3; };
template <typename AT>
__host__ __device__ constexpr int bx_terms_b(int pro) {
  return sizeof(AT) == 4 ? 3 : ((pro == BXPRO_LN || pro == BXPRO_GELU) ? 3 : 1);
}

// uniform base (SGPR pair) + per-lane byte offset (one VGPR): global_load ... v_off, s[base]
template <int N, typename AT>
__device__ __forceinline__ void uload(const AT* ubase, unsigned lane_bytes, float (&v)[N]) {
  vload<N>(reinterpret_cast<const AT*>(reinterpret_cast<const char*>(ubase) + lane_bytes), v);
}

// Epilogue of both kernel forms.  Bias rows come from LDS (staged at kernel start), residual / gate operands are
// requested for a batch of rows before the first is used, every address is `scalar base + 32-bit lane offset`, and there
// is no branch per row (M % 32 == 0 is a host-side condition of the family).
template <int MB, int NACC, int EPI, bool S2D, bool LN, typename AT>
__device__ __forceinline__ void bx_epilogue(const GemmArgsT<AT>& p, f32x16 (&acc)[MB][NACC], int b, int m0, int64_t n0, int j, int h,
                                            int64_t col_off, const float* sBias, const float* sW, const float* tW,
                                            const float (&rstd)[NACC], const float (&mu_d)[NACC]) {
  constexpr int ES = (int)sizeof(AT);
  if constexpr (EPI == EPI_PLAIN) {
    // y[b][m][ncol..]: lane part (4h·Ncol + ncol), uniform part (b·M + m0 + 32mb + rbase(r))·Ncol
    const int64_t ncol = S2D ? n0 + 2 * j : col_off;   // (s2d: col_off addresses the FINE input grid)
    const unsigned yoff = (unsigned)(((int64_t)4 * h * p.Ncol + ncol) * ES);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int64_t urow = (int64_t)b * p.M + m0 + mb * 32;
      // residual / gate operands of RBT rows at a time are requested before the first of them is used
      constexpr int RBT = NACC == 4 ? 4 : (NACC == 2 ? 8 : 16);
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += RBT) {
      float ev[RBT][NACC], rv[RBT][NACC];
      if (p.emul != nullptr) {
#pragma unroll
        for (int rr = 0; rr < RBT; ++rr) uload<NACC>(p.emul + (urow + ((r0 + rr) & 3) + 8 * ((r0 + rr) >> 2)) * p.Ncol, yoff, ev[rr]);
      }
      if (p.res != nullptr) {
#pragma unroll
        for (int rr = 0; rr < RBT; ++rr) uload<NACC>(p.res + (urow + ((r0 + rr) & 3) + 8 * ((r0 + rr) >> 2)) * p.Ncol, yoff, rv[rr]);
      }
#pragma unroll
      for (int rr = 0; rr < RBT; ++rr) {
        const int r = r0 + rr;
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
        float add = sBias[mb * 32 + rl];
        float v[NACC];
        if (LN) {
          const float sw = sW[mb * 32 + rl];
          add += tW[mb * 32 + rl];
#pragma unroll
          for (int q = 0; q < NACC; ++q) v[q] = rstd[q] * (acc[mb][q][r] - mu_d[q] * sw) + add;
        } else {
#pragma unroll
          for (int q = 0; q < NACC; ++q) v[q] = acc[mb][q][r] + add;
        }
        if (p.eact) {
#pragma unroll
          for (int q = 0; q < NACC; ++q) v[q] = act_f(p.eact, v[q]);
        }
        if (p.emul != nullptr) {
#pragma unroll
          for (int q = 0; q < NACC; ++q) v[q] *= act_grad_f(p.emul_kind, ev[rr][q]);
        }
        if (p.res != nullptr) {
#pragma unroll
          for (int q = 0; q < NACC; ++q) v[q] += rv[rr][q];
        }
        AT* yp = reinterpret_cast<AT*>(reinterpret_cast<char*>(p.y + (urow + (r & 3) + 8 * (r >> 2)) * p.Ncol) + yoff);
        vstore<NACC>(yp, v);
      }
      }
    }
  } else {
    // depth-to-space: rows (o, td, th, tw), td = h, th = (r >> 1) & 1, tw = r & 1, o_local = r >> 2; the NACC coarse
    // voxels of a lane are neighbours along W (even Wo), so a lane owns 2·NACC consecutive fine voxels per (o, td, th)
    const int Wf = 2 * p.Wo, Hf = 2 * p.Ho;
    const int64_t Vf = 8 * p.Ncol;
    const int Mo = p.M >> 3;
    const int wo = (int)(col_off % p.Wo);
    const int64_t t2 = col_off / p.Wo;
    const int ho = (int)(t2 % p.Ho);
    const int dz = (int)(t2 / p.Ho);
    const unsigned foff = (unsigned)((((int64_t)(2 * dz + h) * Hf + 2 * ho) * Wf + 2 * wo) * ES);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int64_t uo = (int64_t)b * Mo + ((m0 + mb * 32) >> 3);
      float rv[8][2 * NACC];
      if (p.res != nullptr) {
#pragma unroll
        for (int rp = 0; rp < 8; ++rp)
          uload<2 * NACC>(p.res + (uo + (rp >> 1)) * Vf + (int64_t)(rp & 1) * Wf, foff, rv[rp]);
      }
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        const int r0 = 2 * rp;
        const float bs = sBias[mb * 32 + 8 * (rp >> 1)];   // bias of o_local = rp >> 1 (first row of that o)
        float v[2 * NACC];
#pragma unroll
        for (int q = 0; q < NACC; ++q) { v[2 * q] = acc[mb][q][r0] + bs; v[2 * q + 1] = acc[mb][q][r0 + 1] + bs; }
        if (p.res != nullptr) {
#pragma unroll
          for (int i = 0; i < 2 * NACC; ++i) v[i] += rv[rp][i];
        }
        AT* yp = reinterpret_cast<AT*>(reinterpret_cast<char*>(p.y + (uo + (rp >> 1)) * Vf + (int64_t)(rp & 1) * Wf) + foff);
        vstore<2 * NACC>(yp, v);
      }
    }
  }
}

// K-split form (gemm_bxk.hip); nacc, mb: tile; returns FZ_OK or an error
template <typename AT>
int gemm_bxk_launch(const GemmArgsT<AT>& a, int loader, int epilogue, int pro, int nacc, int mb, fz_stream_t stream);

}  // namespace fz
