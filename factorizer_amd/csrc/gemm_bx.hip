// gemm_bx.hip — channels-first GEMM family on the bf16 matrix cores with SPLIT fp32 operands.
//
//   Out[m, n] = epilogue( Σ_k A[m, k] · prologue(In)[k, n] )        n = voxel (column)
//
// Same layers as gemm.hip's streaming kernel (reference call sites: layers/linear.py:53-58, factorizer.py:38,53,116,
// layers/mlp.py:54-63, unet.py:53,123,128 and autograd through them) for reduction lengths K >= 64, i.e. every dense
// layer and k2s2 (transposed) convolution of stages 1-4 of the U-shape.
//
// Why: gfx950's f32-input MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate, 1/16 of the bf16 MFMA rate.
// An fp32 value is EXACTLY the sum of three bf16 values x = x1 + x2 + x3 (8 + 8 + 8 significand bits, each level
// rounded to nearest), so an fp32 product is a·b = Σ_{i,j} a_i·b_j with every a_i·b_j exact in the fp32 accumulator
// of v_mfma_f32_32x32x16_bf16.  Keeping the six terms with i + j <= 4,
//     a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1),
// drops a2b3 + a3b2 + a3b3 <= 2^-23·|ab| worst case (|x2| <= 2^-8|x|, |x3| <= 2^-16|x|), typically well under
// the 2^-24 rounding of the fp32 FMA it replaces: the result is an fp32 GEMM with fp32 accumulation in 6/16 of the
// matrix-pipe time (measured error against fp64 next to the fp32-MFMA kernel: tools/probes/bx6_accuracy.hip,
// tests/test_gpu_bx.py).  With bf16 activations in HBM (mixed-precision mode, BASELINE configs[4]) the column
// operand IS bf16: one term, and the fp32 weights keep two (16 significand bits) — 2/16.
//
// MFMA mapping (wave64, 32x32x16 bf16): lane l = (j = l&31, h = l>>5) holds B[k = 8h + e][column j], e = 0..7, so a
// K-step of 16 channels needs, per lane, 8 channels of its voxels: lane (j, h) loads ONE vector of NACC consecutive
// voxels of channel 16g + 8h + e for e = 0..7 (every load instruction moves two contiguous 128·NACC-byte runs),
// converts, and the NACC voxel components feed NACC column groups (voxel NACC·j + q) exactly as in gemm.hip — the
// accumulator layout (row in (register, h), column in j) and with it every epilogue of gemm_common.h is unchanged.
// Weights are split once per workgroup into LDS in operand order: As[K16-step][row block][term][lane] x 16 B.
#include "gemm_bx.h"

namespace fz {

// =================================================================================================
// Streaming form (stage 1 and the wide k2s2 convolutions: > 64 K voxels): MB row blocks of 32 x (4 waves x 32·NACC
// columns) per workgroup; each wave owns a column tile and streams its column operand through a register ring of RD
// K16-groups; the weights of the workgroup's rows are split once per 64-index chunk into LDS (double-buffered: the next
// chunk travels global -> registers under the products of the current one).  Same rules as the K-split form
// (gemm_bxk.hip): no branch between a load and its use (shapes restricted on the host), scalar base + 32-bit lane
// offset addressing, bias rows in LDS, batched epilogue operands.  LayerNorm prologue: the row sums s[m] = Σ_k W[m][k]γ[k],
// t[m] = Σ_k W[m][k]β[k] fall out of the weight staging (every thread stages the SAME row in every chunk).
// =================================================================================================
template <int MB, int NACC, int LOADER, int EPI, int PRO, int RD, bool WT, typename AT>
__global__ __launch_bounds__(256, 2) void gemm_bx_kernel(GemmArgsT<AT> p) {
  constexpr int NTA = BxTerms<AT>::A, NTB = bx_terms_b<AT>(PRO);
  constexpr int TN = 32 * NACC;
  constexpr bool S2D = LOADER == LOAD_S2D;
  static_assert(!S2D || NACC == 2, "space-to-depth loader: two coarse voxels per lane");
  constexpr int NL = S2D ? 4 : NACC;
  constexpr int SPG = S2D ? 4 : 8;
  constexpr int ES = (int)sizeof(AT);
  constexpr bool LN = PRO == BXPRO_LN, GATE = PRO == BXPRO_BMUL;
  constexpr int CH = 4;                        // K16-groups per weight chunk
  static_assert(CH % RD == 0, "ring slots must be static inside a chunk");
  constexpr int kItemHalfs = 8;                // one operand item = 8 bf16 = 16 B per lane
  constexpr int kBufItems = CH * MB * NTA * 64;
  constexpr int IPT = CH * MB * 64 / 256;      // weight items (8 consecutive k of one row) per thread and chunk
  __shared__ __attribute__((aligned(16))) __bf16 As[2 * kBufItems * kItemHalfs];
  __shared__ float sW[32 * MB];
  __shared__ float tW[32 * MB];
  __shared__ float sBias[32 * MB];
  __shared__ float sRed[LN ? 2 * 256 : 1];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int tiles_per_sample = (int)((p.Ncol + TN * 4 - 1) / (TN * 4));
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.ygroups > 1) {   // XCD-aware: the row-block groups of one column tile share an XCD's L2 (gemm.hip)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    by = slot % p.ygroups;
    bx = (slot / p.ygroups) * 8 + xcd;
    if (bx >= p.xtiles) return;
  }
  const int b = bx / tiles_per_sample;
  const int64_t n0 = ((int64_t)(bx % tiles_per_sample) * 4 + wave) * TN;
  const int m0 = by * 32 * MB;
  const int nG = p.K >> 4;   // K % 32 == 0: a trailing half chunk runs on zero weights (bf16 storage, K = 32: stage 0)

  if (threadIdx.x < 32 * MB) {
    const int m = m0 + threadIdx.x;
    float bv = 0.f;
    if (p.bias != nullptr) bv = (EPI == EPI_D2S) ? p.bias[m >> 3] : p.bias[m];
    sBias[threadIdx.x] = bv;
  }

  // ---- per-lane byte offsets ----
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
  const unsigned boff = (unsigned)(((int64_t)(S2D ? h : 8 * h) * p.Vin + coff) * ES);

  // weight staging: thread -> items idx = tid + 256u: lane l = tid & 63 and row block mb = (tid >> 6) % MB are the same
  // for every u and every chunk; the K16-group inside the chunk is gl = (tid >> 6) / MB + (4 / MB)·u
  const int wl = threadIdx.x & 63;
  const int wmb = (threadIdx.x >> 6) % MB;
  const int wgl0 = (threadIdx.x >> 6) / MB;
  const int wrow = m0 + wmb * 32 + (wl & 31);
  const unsigned woff = WT ? (unsigned)(((int64_t)8 * (wl >> 5) * p.ldw + wrow) * 4)
                           : (unsigned)(((int64_t)wrow * p.ldw + 8 * (wl >> 5)) * 4);
  const unsigned goff = (unsigned)(32 * (wl >> 5));

  struct Slot { float bv[SPG][GATE ? 2 * NL : NL]; };
  auto fetch = [&](int g, Slot& sl) {
    g = g < nG ? g : nG - 1;   // past the end: harmless re-read of the last group (never consumed)
    if constexpr (!S2D) {
      const int c16 = 16 * g;
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
  };

  Slot ring[RD];
#pragma unroll
  for (int i = 0; i < RD; ++i) fetch(i, ring[i]);

  float shift[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e) shift[e] = 0.f;
  if (LN) {
    // pivot = channel-0 value of the lane's voxels (both lane halves need it): well-conditioned single-pass variance
    float pv[NL];
    vload<NL>(p.x[0] + ((int64_t)b * p.c0) * p.Vin + coff, pv);
#pragma unroll
    for (int e = 0; e < NACC; ++e) shift[e] = pv[e % NL];
  }

  // ---- weights: chunk of CH groups -> registers (raw fp32, times gamma under LayerNorm) -> split -> LDS ----
  float wraw[IPT][8];
  float sWp = 0.f, tWp = 0.f;
  auto load_chunk = [&](int g0) {
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
      int g = g0 + wgl0 + (4 / MB) * u;
      const float live = g < nG ? 1.0f : 0.0f;   // groups past the end of K (half chunk): zero weights, loads clamped
      g = g < nG ? g : nG - 1;
      if constexpr (!WT) {
        uload<8>(p.w + 16 * g, woff, wraw[u]);
      } else {
        const float* uw = p.w + (int64_t)16 * g * p.ldw;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t[1];
          uload<1>(uw + (int64_t)e * p.ldw, woff, t);
          wraw[u][e] = t[0];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) wraw[u][e] *= live;
      if constexpr (LN) {
        float gm[8], bt[8];
        uload<8>(p.ln_g + 16 * g, goff, gm);
        uload<8>(p.ln_b + 16 * g, goff, bt);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          tWp += wraw[u][e] * bt[e];
          wraw[u][e] *= gm[e];
          sWp += wraw[u][e];
        }
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
      const int gm = (threadIdx.x >> 6) + 4 * u;   // gl * MB + mb
      bx8 t[NTA];
      bx_split<NTA>(wraw[u], t);
#pragma unroll
      for (int i = 0; i < NTA; ++i)
        *reinterpret_cast<bx8*>(&As[((buf * CH * MB + gm) * NTA + i) * 64 * kItemHalfs + wl * kItemHalfs]) = t[i];
    }
  };

  f32x16 acc[MB][NACC];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][q][r] = 0.f;
  float s1[NACC], s2[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e) s1[e] = s2[e] = 0.f;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  int cbuf = 0;
  for (int g0 = 0; g0 < nG; g0 += CH) {
    const bool more = g0 + CH < nG;
    if (more) load_chunk(g0 + CH);   // in flight during the products below
    const __bf16* Ab = As + cbuf * kBufItems * kItemHalfs;
#pragma unroll
    for (int gl = 0; gl < CH; ++gl) {
      Slot& sl = ring[gl % RD];
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
              const float tv = (g0 + gl < nG) ? t : 0.f;   // (a clamped re-read past the end of K is not part of the row)
              s1[q] += tv;
              s2[q] += tv * tv;
            }
            if (PRO == BXPRO_GELU) t = gelu_f(t);
          }
          x[e] = t;
        }
        bx_split<NTB>(x, bop[q]);
      }
      fetch(g0 + gl + RD, sl);   // refill the slot just consumed
      // ---- products ----
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        bx8 aop[NTA];
#pragma unroll
        for (int i = 0; i < NTA; ++i)
          aop[i] = *reinterpret_cast<const bx8*>(&Ab[((gl * MB + mb) * NTA + i) * 64 * kItemHalfs + lane * kItemHalfs]);
#pragma unroll
        for (int q = 0; q < NACC; ++q) bx_mfma<NTA, NTB>(acc[mb][q], aop, bop[q]);
      }
    }
    if (more) {
      store_chunk(cbuf ^ 1);
      cbuf ^= 1;
      __syncthreads();
    }
  }

  float mu_d[NACC], rstd[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e) { mu_d[e] = 0.f; rstd[e] = 1.f; }
  if (LN) {
    // row sums: the threads that staged row (mb, i) are tid = 64·w + 32·hh + i with w % MB == mb
    sRed[threadIdx.x] = sWp;
    sRed[256 + threadIdx.x] = tWp;
    __syncthreads();
    if (threadIdx.x < 32 * MB) {
      const int mb = threadIdx.x >> 5, i = threadIdx.x & 31;
      float a = 0.f, c = 0.f;
      for (int w = mb; w < 4; w += MB)
        for (int hh = 0; hh < 2; ++hh) { a += sRed[64 * w + 32 * hh + i]; c += sRed[256 + 64 * w + 32 * hh + i]; }
      sW[threadIdx.x] = a;
      tW[threadIdx.x] = c;
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
    __syncthreads();
  }
  if (!col_ok) return;
  bx_epilogue<MB, NACC, EPI, S2D, LN, AT>(p, acc, b, m0, n0, j, h, col_off, sBias, sW, tW, rstd, mu_d);
}

// shapes the family takes (everything else stays with the fp32-MFMA kernels of gemm.hip)
template <typename AT>
bool gemm_bx_ok(const GemmArgsT<AT>& a, int loader, int epilogue, int pro, int nacc) {
  if (a.K % 32 != 0 || a.M % 32 != 0) return false;
  if (loader == LOAD_PLAIN && (a.c0 % 16 != 0 || a.Cin != a.K)) return false;
  if (loader == LOAD_S2D && (8 * a.Cin != a.K || pro != BXPRO_NONE || epilogue != EPI_PLAIN || (a.Wo & 1))) return false;
  if (epilogue == EPI_D2S && (pro != BXPRO_NONE || (nacc >= 2 && (a.Wo % nacc) != 0))) return false;
  if (a.Ncol % nacc != 0 || a.Vin % nacc != 0) return false;
  if (!a.w_t && ((a.ldw & 3) != 0 || (reinterpret_cast<uintptr_t>(a.w) & 15) != 0)) return false;   // 32-byte weight reads
  if (pro == BXPRO_LN && ((reinterpret_cast<uintptr_t>(a.ln_g) & 15) != 0 || (reinterpret_cast<uintptr_t>(a.ln_b) & 15) != 0)) return false;
  const int64_t es = (int64_t)sizeof(AT);
  if ((8 * a.Vin + a.Vin) * es >= ((int64_t)1 << 31)) return false;                 // 32-bit lane offsets
  if (((int64_t)a.M * a.ldw + 8 * a.ldw) * 4 >= ((int64_t)1 << 31)) return false;
  if ((4 * a.Ncol + a.Ncol) * es * 8 >= ((int64_t)1 << 31)) return false;
  return true;
}

// Host side: tile selection and launch.  Returns FZ_E_UNSUPPORTED (without setting an error) when the descriptor is
// outside this family, so that fz_gemm falls through to the fp32-MFMA kernels.
template <typename AT>
int gemm_bx_launch(const GemmArgsT<AT>& a0, int loader, int epilogue, int pro, fz_stream_t stream) {
  GemmArgsT<AT> a = a0;
  hipStream_t st = (hipStream_t)stream;
  const int mblocks = (a.M + 31) / 32;
  int nacc, mb;
  {
    // narrow problems (stages 2-4): K-split form — tile chosen for >= 512 workgroups where the problem has them
    // (measured, profiles/r04_gemm_bx_sweep.md: the streaming form wins from 16^3 x 2 columns on — except for deep reductions:
    // at 8 192 columns the K = 1 024 convolution 128 -> 256 takes 34 us K-split against 57 us streaming, GELU + 512 -> 256 29 against 39)
    int ks = (a.Ncol * a.B <= 4096 || (a.Ncol * a.B <= 8192 && a.K >= 512)) ? 1 : 0;
    int kn = 2, km = mblocks >= 2 ? 2 : 1;
    auto wgs = [&](int na, int mbb) { return ((a.Ncol + 32 * na - 1) / (32 * na)) * a.B * ((mblocks + mbb - 1) / mbb); };
    if (wgs(kn, km) < 512 && km == 2) km = 1;
    if (wgs(kn, km) < 512 && loader != LOAD_S2D) kn = 1;
    if (a.tune >= 100) {   // per-call form of the same diagnostics (fz_gemm_desc.tune)
      ks = a.tune / 100 == 2;
      if (ks) { kn = (a.tune / 10) % 10; km = a.tune % 10; if (km > mblocks) km = 1; if (loader == LOAD_S2D) kn = 2; if (kn < 1 || kn > 2) kn = 2; if (km < 1 || km > 2) km = 1; }
    }
    if (ks) {
      const int rc = gemm_bxk_launch<AT>(a, loader, epilogue, pro, kn, km, stream);
      if (rc != FZ_E_UNSUPPORTED) return rc;   // shapes outside the K-split form: the streaming kernel below
    }
  }
  if (loader == LOAD_S2D) {
    nacc = 2;
    mb = mblocks >= 2 ? 2 : 1;
  } else {
    // widest tile that still gives every CU about two workgroups
    nacc = 4; mb = mblocks >= 2 ? 2 : 1;
    auto wgs = [&](int na, int mbb) { return ((a.Ncol + 128 * na - 1) / (128 * na)) * a.B * ((mblocks + mbb - 1) / mbb); };
    if (wgs(nacc, mb) < 512) nacc = 2;
    if (wgs(nacc, mb) < 512 && mb == 2) mb = 1;
    if (a.tune / 100 == 1) { nacc = (a.tune / 10) % 10; mb = a.tune % 10; if (mb > mblocks || mb < 1) mb = 1; }
    if (nacc != 2 && nacc != 4) nacc = 2;
    // the LayerNorm prologue (row sums, statistics) and the gated operand (second ring) do not fit beside 128-voxel wave
    // tiles without spilling: 64-voxel tiles
    if (pro == BXPRO_LN || pro == BXPRO_BMUL) nacc = 2;
  }
  if (mblocks % mb != 0) mb = 1;
  if (!gemm_bx_ok(a, loader, epilogue, pro, nacc)) return FZ_E_UNSUPPORTED;
  const int TN = 32 * nacc;
  const int64_t tiles = (a.Ncol + TN * 4 - 1) / (TN * 4);
  const int ygr = mblocks / mb;
  a.ygroups = (ygr > 1 && ygr <= 8 && tiles * a.B >= 64) ? ygr : 0;
  a.xtiles = (int)(tiles * a.B);
  dim3 grid((unsigned)(tiles * a.B), (unsigned)ygr), block(256);
  if (a.ygroups > 1) grid = dim3((unsigned)(((tiles * a.B + 7) / 8) * 8 * ygr), 1);
  const bool wt = a.w_t != 0;
#define FZ_BX(MBv, NAv, L, E, PR, RDv)                                                                           \
  do {                                                                                                           \
    if (wt) hipLaunchKernelGGL((gemm_bx_kernel<MBv, NAv, L, E, PR, RDv, true, AT>), grid, block, 0, st, a);      \
    else hipLaunchKernelGGL((gemm_bx_kernel<MBv, NAv, L, E, PR, RDv, false, AT>), grid, block, 0, st, a);        \
  } while (0)
// ring depth by tile: RD slots of 8·NACC registers (x2 with the gate operand) beside MB·NACC·16 accumulators
#define FZ_BX_TILES(L, E, PR)                                                                                    \
  do {                                                                                                           \
    if (nacc == 4) { if (mb == 2) FZ_BX(2, 4, L, E, PR, 1); else FZ_BX(1, 4, L, E, PR, 2); }                      \
    else { if (mb == 2) FZ_BX(2, 2, L, E, PR, 4); else FZ_BX(1, 2, L, E, PR, 4); }                                \
  } while (0)
#define FZ_BX_TILES2(L, E, PR, RD2)                                                                              \
  do {                                                                                                           \
    if (mb == 2) FZ_BX(2, 2, L, E, PR, RD2); else FZ_BX(1, 2, L, E, PR, 4);                                       \
  } while (0)
  if (loader == LOAD_S2D) {
    if (mb == 2) FZ_BX(2, 2, LOAD_S2D, EPI_PLAIN, BXPRO_NONE, 4); else FZ_BX(1, 2, LOAD_S2D, EPI_PLAIN, BXPRO_NONE, 4);
  } else if (epilogue == EPI_D2S) FZ_BX_TILES(LOAD_PLAIN, EPI_D2S, BXPRO_NONE);
  else if (pro == BXPRO_LN) FZ_BX_TILES2(LOAD_PLAIN, EPI_PLAIN, BXPRO_LN, 4);
  else if (pro == BXPRO_GELU) FZ_BX_TILES(LOAD_PLAIN, EPI_PLAIN, BXPRO_GELU);
  else if (pro == BXPRO_BMUL) FZ_BX_TILES2(LOAD_PLAIN, EPI_PLAIN, BXPRO_BMUL, 2);
  else FZ_BX_TILES(LOAD_PLAIN, EPI_PLAIN, BXPRO_NONE);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

template int gemm_bx_launch<float>(const GemmArgsT<float>&, int, int, int, fz_stream_t);
template int gemm_bx_launch<bf16>(const GemmArgsT<bf16>&, int, int, int, fz_stream_t);

}  // namespace fz
