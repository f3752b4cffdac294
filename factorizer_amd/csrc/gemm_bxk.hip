// gemm_bxk.hip — K-split form of the split-bf16 MFMA GEMM family (arithmetic: gemm_bx.hip / gemm_bx.h) for the narrow,
// deep problems of the U-shape: stages 2-4 (<= 64 K voxels, 128-1024 channels), where a launch is a handful of
// microseconds of matrix work and its duration is set by dependent memory round trips, not by bytes or flops.
//
// Design rules (each one measured on the fp32 family's deep-stage launches, 25-60 us for 3-7 us of work):
//   * The 4 waves of a workgroup share ONE tile of 32·MB rows x 32·NACC columns and take the K16-groups round-robin;
//     BOTH operands stream through a register ring of RD groups per wave (no LDS staging of weights, no barrier in
//     front of the first MFMA): 4·RD·16 reduction indices are in flight per workgroup, so K = 512 is two memory
//     round trips instead of a chain of 32 dependent ones.
//   * No branch and no select sits between a load and its use: shapes are restricted on the host (K % 64 == 0,
//     M % 32 == 0, concat boundary % 16 == 0; everything else falls back to the streaming kernels), out-of-range
//     columns read a clamped address and are simply not stored.
//   * Every global access is `scalar base + 32-bit lane offset`: the per-group address arithmetic is SALU work.
//   * Bias rows wait in LDS from the first instruction on; residual / gate operands of the epilogue are requested for
//     a whole 32-row block before the first of them is used.
//   * Partial accumulators of the four waves meet in LDS and are added in wave order (bitwise reproducible).
#include "gemm_bx.h"

namespace fz {

template <int MB, int NACC, int LOADER, int EPI, int PRO, int RD, bool WT, typename AT>
__global__ __launch_bounds__(256, 2) void gemm_bxk_kernel(GemmArgsT<AT> p) {
  constexpr int NTA = BxTerms<AT>::A, NTB = bx_terms_b<AT>(PRO);
  constexpr int TN = 32 * NACC;
  constexpr bool S2D = LOADER == LOAD_S2D;
  static_assert(!S2D || NACC == 2, "space-to-depth loader: two coarse voxels per lane");
  constexpr int NL = S2D ? 4 : NACC;
  constexpr int SPG = S2D ? 4 : 8;
  constexpr int ES = (int)sizeof(AT);
  constexpr bool LN = PRO == BXPRO_LN, GATE = PRO == BXPRO_BMUL;
  constexpr int kExtra = LN ? 2 * NACC + 2 * MB : 0;           // s1, s2, sW, tW partials
  constexpr int kPer = (MB * NACC * 16 + kExtra) * 64;          // floats parked per wave
  __shared__ float red[3 * kPer];
  __shared__ float sW[32 * MB];
  __shared__ float tW[32 * MB];
  __shared__ float sBias[32 * MB];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int tiles_per_sample = (int)((p.Ncol + TN - 1) / TN);
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.ygroups > 1) {   // XCD-aware: the row-block groups of one column tile share an XCD's L2 (gemm.hip)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    by = slot % p.ygroups;
    bx = (slot / p.ygroups) * 8 + xcd;
    if (bx >= p.xtiles) return;
  }
  const int b = bx / tiles_per_sample;
  const int64_t n0 = (int64_t)(bx % tiles_per_sample) * TN;
  const int m0 = by * 32 * MB;
  const int nIt = p.K >> 6;            // K16-groups per wave (K % 64 == 0): g = 4·it + wave

  // bias rows of this workgroup -> LDS (read back in the epilogue, behind the reduction barrier)
  if (threadIdx.x < 32 * MB) {
    const int m = m0 + threadIdx.x;
    float bv = 0.f;
    if (p.bias != nullptr) bv = (EPI == EPI_D2S) ? p.bias[m >> 3] : p.bias[m];
    sBias[threadIdx.x] = bv;
  }

  // ---- per-lane byte offsets (32 bit) ----
  int64_t col_off;
  bool col_ok;
  if (S2D) {
    const int64_t n = n0 + 2 * j;
    col_ok = n < p.Ncol;
    const int64_t nn = col_ok ? n : 0;
    const int wo = (int)(nn % p.Wo);
    const int64_t t2 = nn / p.Wo;
    const int ho = (int)(t2 % p.Ho);
    const int dz = (int)(t2 / p.Ho);
    col_off = ((int64_t)(2 * dz) * p.Hi + 2 * ho) * p.Wi + 2 * wo;
  } else {
    col_off = n0 + NACC * j;
    col_ok = col_off < p.Ncol;
  }
  const int64_t coff = col_ok ? col_off : 0;
  // column operand: channel (16g + 8h + e) [plain] / (2g + h) [s2d] of the lane's voxels
  const unsigned boff = (unsigned)(((int64_t)(S2D ? h : 8 * h) * p.Vin + coff) * ES);
  // weights: A[m][k]: row m0 + 32mb + j, k = 16g + 8h + e
  unsigned aoff[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = m0 + mb * 32 + j;
    aoff[mb] = WT ? (unsigned)(((int64_t)8 * h * p.ldw + m) * 4) : (unsigned)(((int64_t)m * p.ldw + 8 * h) * 4);
  }
  const unsigned goff = (unsigned)(32 * h);   // LayerNorm gamma / beta of the lane's 8 k

  struct Slot {
    float bv[SPG][GATE ? 2 * NL : NL];
    float av[MB][8];
    float gb[LN ? 16 : 1];
  };
  auto fetch = [&](int it, Slot& sl) {
    // past the end: harmless re-read of the last group (never consumed)
    const int g = 4 * (it < nIt ? it : nIt - 1) + wave;
    if constexpr (!S2D) {
      const int c16 = 16 * g;                       // (concat boundary % 16 == 0: a group lies in one source)
      const bool first = c16 < p.c0;
      const AT* src = first ? p.x[0] : p.x[1];
      const int cs = first ? p.c0 : p.Cin - p.c0;
      const int ci = first ? c16 : c16 - p.c0;
      const AT* ub = src + ((int64_t)b * cs + ci) * p.Vin;
      const AT* ug = GATE ? p.bmul + ((int64_t)b * p.Cin + c16) * p.Vin : nullptr;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a[NL];
        uload<NL>(ub + (int64_t)e * p.Vin, boff, a);
#pragma unroll
        for (int i = 0; i < NL; ++i) sl.bv[e][i] = a[i];
        if constexpr (GATE) {
          float t[NL];
          uload<NL>(ug + (int64_t)e * p.Vin, boff, t);
#pragma unroll
          for (int i = 0; i < NL; ++i) sl.bv[e][NL + i] = t[i];
        }
      }
    } else {
      const AT* ub = p.x[0] + ((int64_t)b * p.Cin + 2 * g) * p.Vin;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a[NL];
        uload<NL>(ub + (int64_t)(e >> 1) * p.Hi * p.Wi + (int64_t)(e & 1) * p.Wi, boff, a);
#pragma unroll
        for (int i = 0; i < NL; ++i) sl.bv[e][i] = a[i];
      }
    }
    if constexpr (!WT) {
      const float* uw = p.w + 16 * g;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) uload<8>(uw, aoff[mb], sl.av[mb]);
    } else {
      const float* uw = p.w + (int64_t)16 * g * p.ldw;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t[1];
          uload<1>(uw + (int64_t)e * p.ldw, aoff[mb], t);
          sl.av[mb][e] = t[0];
        }
    }
    if constexpr (LN) {
      float t[8];
      uload<8>(p.ln_g + 16 * g, goff, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) sl.gb[e] = t[e];
      uload<8>(p.ln_b + 16 * g, goff, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) sl.gb[8 + e] = t[e];
    }
  };

  Slot ring[RD];
#pragma unroll
  for (int i = 0; i < RD; ++i) fetch(i, ring[i]);

  float shift[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e) shift[e] = 0.f;
  if (LN) {
    // pivot = channel-0 value of the lane's voxels (every wave needs the SAME one): well-conditioned single-pass variance
    float pv[NL];
    vload<NL>(p.x[0] + ((int64_t)b * p.c0) * p.Vin + coff, pv);
#pragma unroll
    for (int e = 0; e < NACC; ++e) shift[e] = pv[e % NL];
  }

  f32x16 acc[MB][NACC];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][q][r] = 0.f;
  float s1[NACC], s2[NACC], sWp[MB], tWp[MB];
#pragma unroll
  for (int e = 0; e < NACC; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) sWp[mb] = tWp[mb] = 0.f;

  for (int it0 = 0; it0 < nIt; it0 += RD) {
#pragma unroll
    for (int rd = 0; rd < RD; ++rd) {
      const int it = it0 + rd;
      if (RD > 1 && it >= nIt) break;
      Slot& sl = ring[rd];
      // ---- column operands: prologue + split ----
      bx8 bop[NACC][NTB];
#pragma unroll
      for (int q = 0; q < NACC; ++q) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t;
          if constexpr (S2D) {
            t = sl.bv[e >> 1][2 * q + (e & 1)];     // k = c·8 + td·4 + th·2 + tw
          } else {
            t = sl.bv[e][q];
            if (GATE) t = sl.bv[e][NL + q] > 0.f ? t : 0.f;
            if (LN) {
              t -= shift[q];
              s1[q] += t;
              s2[q] += t * t;
            }
            if (PRO == BXPRO_GELU) t = gelu_f(t);
          }
          x[e] = t;
        }
        bx_split<NTB>(x, bop[q]);
      }
      // ---- weights ----
      bx8 aop[MB][NTA];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        float wv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = sl.av[mb][e];
          if (LN) {
            tWp[mb] += v * sl.gb[8 + e];
            v *= sl.gb[e];
            sWp[mb] += v;
          }
          wv[e] = v;
        }
        bx_split<NTA>(wv, aop[mb]);
      }
      // ---- refill the slot (group RD rounds ahead), then the products ----
      fetch(it + RD, sl);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int q = 0; q < NACC; ++q) bx_mfma<NTA, NTB>(acc[mb][q], aop[mb], bop[q]);
    }
  }

  // ---- add the K-slices: waves 1..3 park their partial sums, wave 0 adds them in wave order ----
  if (LN) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {   // the two lane halves hold the two 8-k halves of every group of the lane's row
      sWp[mb] += __shfl_xor(sWp[mb], 32, 64);
      tWp[mb] += __shfl_xor(tWp[mb], 32, 64);
    }
  }
  if (wave > 0) {
    float* dst = red + (wave - 1) * kPer + lane;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((mb * NACC + q) * 16 + r) * 64] = acc[mb][q][r];
    if (LN) {
#pragma unroll
      for (int e = 0; e < NACC; ++e) {
        dst[(MB * NACC * 16 + e) * 64] = s1[e];
        dst[(MB * NACC * 16 + NACC + e) * 64] = s2[e];
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        dst[(MB * NACC * 16 + 2 * NACC + mb) * 64] = sWp[mb];
        dst[(MB * NACC * 16 + 2 * NACC + MB + mb) * 64] = tWp[mb];
      }
    }
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll 1
  for (int w = 0; w < 3; ++w) {   // (rolled, one 16-register block at a time: all 3·64·MB·NACC reads at once would spill)
    const float* src = red + w * kPer + lane;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int q = 0; q < NACC; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][q][r] += src[((mb * NACC + q) * 16 + r) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
    if (LN) {
#pragma unroll
      for (int e = 0; e < NACC; ++e) {
        s1[e] += src[(MB * NACC * 16 + e) * 64];
        s2[e] += src[(MB * NACC * 16 + NACC + e) * 64];
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        sWp[mb] += src[(MB * NACC * 16 + 2 * NACC + mb) * 64];
        tWp[mb] += src[(MB * NACC * 16 + 2 * NACC + MB + mb) * 64];
      }
    }
  }

  float mu_d[NACC], rstd[NACC];
  if (LN) {
    if (h == 0) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) { sW[mb * 32 + j] = sWp[mb]; tW[mb * 32 + j] = tWp[mb]; }
    }
#pragma unroll
    for (int e = 0; e < NACC; ++e) {
      const float t1 = s1[e] + __shfl_xor(s1[e], 32, 64);
      const float t2 = s2[e] + __shfl_xor(s2[e], 32, 64);
      const float inv = 1.0f / (float)p.Cin;
      const float md = t1 * inv;
      float var = t2 * inv - md * md;
      var = var > 0.f ? var : 0.f;
      mu_d[e] = md;
      rstd[e] = 1.0f / sqrtf(var + p.ln_eps);
    }
    if (p.stats_out != nullptr && by == 0 && h == 0 && col_ok) {
      float mean[NACC];
#pragma unroll
      for (int e = 0; e < NACC; ++e) mean[e] = shift[e] + mu_d[e];
      float* so = p.stats_out + (int64_t)b * 2 * p.Vin;
      vstore<NACC>(so + col_off, mean);
      vstore<NACC>(so + p.Vin + col_off, rstd);
    }
  }
  if (!col_ok) return;

  bx_epilogue<MB, NACC, EPI, S2D, LN, AT>(p, acc, b, m0, n0, j, h, col_off, sBias, sW, tW, rstd, mu_d);
}

// shapes this form takes (everything else stays with the streaming kernels)
template <typename AT>
bool gemm_bxk_ok(const GemmArgsT<AT>& a, int loader, int epilogue, int pro, int nacc) {
  if (a.K % 64 != 0 || a.M % 32 != 0) return false;
  if (loader == LOAD_PLAIN && (a.c0 % 16 != 0 || a.Cin != a.K)) return false;
  if (loader == LOAD_S2D && (8 * a.Cin != a.K || pro != BXPRO_NONE || epilogue != EPI_PLAIN || (a.Wo & 1))) return false;
  if (epilogue == EPI_D2S && (pro != BXPRO_NONE || (nacc == 2 && (a.Wo & 1)))) return false;
  if (a.Ncol % nacc != 0 || a.Vin % nacc != 0) return false;
  if (!a.w_t && ((a.ldw & 3) != 0 || (reinterpret_cast<uintptr_t>(a.w) & 15) != 0)) return false;   // 32-byte weight reads
  // 32-bit lane offsets
  const int64_t es = (int64_t)sizeof(AT);
  if ((8 * a.Vin + a.Vin) * es >= ((int64_t)1 << 31)) return false;
  if (((int64_t)a.M * a.ldw + 8 * a.ldw) * 4 >= ((int64_t)1 << 31)) return false;
  if ((4 * a.Ncol + a.Ncol) * es * 8 >= ((int64_t)1 << 31)) return false;
  return true;
}

template <typename AT>
int gemm_bxk_launch(const GemmArgsT<AT>& a0, int loader, int epilogue, int pro, int nacc, int mb, fz_stream_t stream) {
  GemmArgsT<AT> a = a0;
  hipStream_t st = (hipStream_t)stream;
  if (!gemm_bxk_ok(a, loader, epilogue, pro, nacc)) return FZ_E_UNSUPPORTED;  // (no message: the caller falls through)
  const int mblocks = a.M / 32;
  if (mblocks % mb != 0) mb = 1;
  const int TN = 32 * nacc;
  const int64_t tiles = (a.Ncol + TN - 1) / TN;
  const int ygr = mblocks / mb;
  a.ygroups = (ygr > 1 && ygr <= 8 && tiles * a.B >= 64) ? ygr : 0;
  a.xtiles = (int)(tiles * a.B);
  dim3 grid((unsigned)(tiles * a.B), (unsigned)ygr), block(256);
  if (a.ygroups > 1) grid = dim3((unsigned)(((tiles * a.B + 7) / 8) * 8 * ygr), 1);
  const bool wt = a.w_t != 0;
#define FZ_BXK(MBv, NAv, L, E, PR, RDv)                                                                            \
  do {                                                                                                             \
    if (wt) hipLaunchKernelGGL((gemm_bxk_kernel<MBv, NAv, L, E, PR, RDv, true, AT>), grid, block, 0, st, a);       \
    else hipLaunchKernelGGL((gemm_bxk_kernel<MBv, NAv, L, E, PR, RDv, false, AT>), grid, block, 0, st, a);         \
  } while (0)
// prefetch depth by tile: the ring (RD slots of 8·NACC (x2 gated) + 8·MB (+16 LayerNorm) registers) must fit beside the accumulators
#define FZ_BXK_TILES(L, E, PR)                                                                                     \
  do {                                                                                                             \
    constexpr bool heavy = (PR == BXPRO_LN || PR == BXPRO_BMUL);                                                   \
    if (nacc == 2) { if (mb == 2) FZ_BXK(2, 2, L, E, PR, (heavy ? 1 : 2)); else FZ_BXK(1, 2, L, E, PR, (heavy ? 2 : 4)); } \
    else { if (mb == 2) FZ_BXK(2, 1, L, E, PR, (heavy ? 2 : 4)); else FZ_BXK(1, 1, L, E, PR, (heavy ? 2 : 4)); }    \
  } while (0)
  if (nacc != 1 && nacc != 2) return FZ_E_UNSUPPORTED;
  if (loader == LOAD_S2D) {
    if (mb == 2) FZ_BXK(2, 2, LOAD_S2D, EPI_PLAIN, BXPRO_NONE, 2); else FZ_BXK(1, 2, LOAD_S2D, EPI_PLAIN, BXPRO_NONE, 4);
  } else if (epilogue == EPI_D2S) FZ_BXK_TILES(LOAD_PLAIN, EPI_D2S, BXPRO_NONE);
  else if (pro == BXPRO_LN) FZ_BXK_TILES(LOAD_PLAIN, EPI_PLAIN, BXPRO_LN);
  else if (pro == BXPRO_GELU) FZ_BXK_TILES(LOAD_PLAIN, EPI_PLAIN, BXPRO_GELU);
  else if (pro == BXPRO_BMUL) FZ_BXK_TILES(LOAD_PLAIN, EPI_PLAIN, BXPRO_BMUL);
  else FZ_BXK_TILES(LOAD_PLAIN, EPI_PLAIN, BXPRO_NONE);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

template int gemm_bxk_launch<float>(const GemmArgsT<float>&, int, int, int, int, int, fz_stream_t);
template int gemm_bxk_launch<bf16>(const GemmArgsT<bf16>&, int, int, int, int, int, fz_stream_t);

}  // namespace fz
