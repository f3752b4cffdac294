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
// MB row blocks of 32 x (4 waves x 32·NACC columns) per workgroup; RD K16-groups of operand loads in flight per lane.
// =================================================================================================
template <int MB, int NACC, int LOADER, int EPI, int PRO, int RD, typename AT>
__global__ __launch_bounds__(256, 2) void gemm_bx_kernel(GemmArgsT<AT> p) {
  constexpr int NTA = BxTerms<AT>::A, NTB = bx_terms_b<AT>(PRO);
  constexpr int TN = 32 * NACC;
  constexpr bool S2D = LOADER == LOAD_S2D;
  static_assert(!S2D || NACC == 2, "space-to-depth loader: two coarse voxels per lane");
  constexpr int NL = S2D ? 4 : NACC;          // elements per load
  constexpr int SPG = S2D ? 4 : 8;            // loads (ring slots) per K16-group
  constexpr int NR = (PRO == BXPRO_BMUL) ? 2 * NL : NL;
  constexpr int CH = 4;                       // K16-steps per weight chunk (64 reduction indices)
  constexpr int kItemHalfs = 8;               // one operand item = 8 bf16 = 16 B per lane
  constexpr int kBufItems = CH * MB * NTA * 64;
  constexpr int IPT = CH * MB * 64 / 256;     // weight items (8 consecutive k of one row) per thread and chunk
  __shared__ __attribute__((aligned(16))) __bf16 As[2 * kBufItems * kItemHalfs];
  __shared__ float sW[32 * MB];
  __shared__ float tW[32 * MB];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
  const int nG = (p.K + 15) / 16;  // K16-groups

  // ---- per-lane input addressing ----
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

  // slot s (global load index): plain: group s/8, channel 16g + 8h + s%8 ; s2d: group s/4, channel 2g + h, rows (td, th) = s%4
  auto fetch = [&](int s, float (&v)[NR]) {
    if constexpr (!S2D) {
      const int c = 16 * (s >> 3) + 8 * h + (s & 7);
      const int cc = c < p.Cin ? c : p.Cin - 1;
      const bool first = cc < p.c0;
      const AT* base = first ? p.x[0] : p.x[1];
      const int cs = first ? p.c0 : p.Cin - p.c0;
      const int ci = first ? cc : cc - p.c0;
      float a[NL];
      vload<NL>(base + ((int64_t)b * cs + ci) * p.Vin + coff, a);
#pragma unroll
      for (int i = 0; i < NL; ++i) v[i] = a[i];
      if constexpr (PRO == BXPRO_BMUL) {
        float e[NL];
        vload<NL>(p.bmul + ((int64_t)b * p.Cin + cc) * p.Vin + coff, e);
#pragma unroll
        for (int i = 0; i < NL; ++i) v[NL + i] = e[i];
      }
    } else {
      const int c = 2 * (s >> 2) + h;
      const int cc = c < p.Cin ? c : p.Cin - 1;
      const int64_t off = coff + (int64_t)((s >> 1) & 1) * p.Hi * p.Wi + (int64_t)(s & 1) * p.Wi;
      float a[NL];
      vload<NL>(p.x[0] + ((int64_t)b * p.Cin + cc) * p.Vin + off, a);
#pragma unroll
      for (int i = 0; i < NL; ++i) v[i] = a[i];
    }
  };

  const int nslots = nG * SPG;
  float ring[RD * SPG][NR];
#pragma unroll
  for (int i = 0; i < RD * SPG; ++i) fetch(i < nslots ? i : nslots - 1, ring[i]);

  if (PRO == BXPRO_LN) {
    // s[m] = Σ_k W[m][k]·γ[k], t[m] = Σ_k W[m][k]·β[k]  (8 threads per row)
    for (int r0 = 0; r0 < 32 * MB; r0 += 32) {
      const int r = r0 + (threadIdx.x >> 3), part = threadIdx.x & 7;
      const int m = m0 + r;
      float s = 0.f, t = 0.f;
      if (m < p.M)
        for (int k = part; k < p.K; k += 8) {
          const float wv = weight_at(p, m, k);
          s += wv * p.ln_g[k];
          t += wv * p.ln_b[k];
        }
      s += __shfl_xor(s, 1, 64); t += __shfl_xor(t, 1, 64);
      s += __shfl_xor(s, 2, 64); t += __shfl_xor(t, 2, 64);
      s += __shfl_xor(s, 4, 64); t += __shfl_xor(t, 4, 64);
      if (part == 0) { sW[r] = s; tW[r] = t; }
    }
  }

  // ---- weights: chunk [g0, g0 + CH) of K16-groups -> registers (raw fp32) -> split -> LDS in operand order ----
  float wraw[IPT][8];
  auto load_chunk = [&](int g0) {
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
      const int idx = threadIdx.x + u * 256;           // (gl, mb, lane)
      const int l = idx & 63;
      const int mb = (idx >> 6) % MB;
      const int g = g0 + idx / (64 * MB);
      const int m = m0 + mb * 32 + (l & 31);
      const int k0 = 16 * g + 8 * (l >> 5);
      const int mc = m < p.M ? m : p.M - 1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        const int kc = k < p.K ? k : p.K - 1;
        float wv = weight_at(p, mc, kc);
        if (PRO == BXPRO_LN) wv *= p.ln_g[kc];
        wraw[u][e] = (m < p.M && k < p.K) ? wv : 0.f;
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
      const int idx = threadIdx.x + u * 256;
      const int l = idx & 63;
      const int gm = idx >> 6;                          // gl * MB + mb
      bx8 t[NTA];
      bx_split<NTA>(wraw[u], t);
#pragma unroll
      for (int i = 0; i < NTA; ++i)
        *reinterpret_cast<bx8*>(&As[((buf * CH * MB + gm) * NTA + i) * 64 * kItemHalfs + l * kItemHalfs]) = t[i];
    }
  };

  f32x16 acc[MB][NACC];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][q][r] = 0.f;

  float s1[NACC], s2[NACC], shift[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e) s1[e] = s2[e] = shift[e] = 0.f;
  if (PRO == BXPRO_LN) {
    // pivot = channel-0 value (lane half 0, slot 0): well-conditioned single-pass variance
#pragma unroll
    for (int e = 0; e < NACC; ++e) shift[e] = __shfl(ring[0][e % NL], j, 64);
  }

  load_chunk(0);
  store_chunk(0);
  __syncthreads();  // sW / tW and chunk 0
  int cbuf = 0;
  for (int g0 = 0; g0 < nG; g0 += CH) {
    const bool more = g0 + CH < nG;
    if (more) load_chunk(g0 + CH);  // in flight during the MFMAs below
    const __bf16* Ab = As + cbuf * kBufItems * kItemHalfs;
    // RD groups per unrolled body so that ring indices are static
    for (int gl = 0; gl < CH; gl += RD) {
#pragma unroll
      for (int rd = 0; rd < RD; ++rd) {
        const int g = g0 + gl + rd;
        if (RD > 1 && CH % RD != 0 && gl + rd >= CH) break;
        // ---- column operands of this group: prologue + split, column group by column group ----
        bx8 bop[NACC][NTB];
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t;
            if constexpr (S2D) {
              // k = c*8 + td*4 + th*2 + tw: slot (td, th) = e >> 1, component 2q + tw
              t = ring[rd * SPG + (e >> 1)][2 * q + (e & 1)];
              const bool cok = col_ok && (2 * g + h) < p.Cin;
              t = cok ? t : 0.f;
            } else {
              t = ring[rd * SPG + e][q % NL];
              const bool cok = col_ok && (16 * g + 8 * h + e) < p.Cin;
              if (PRO == BXPRO_BMUL) t = ring[rd * SPG + e][(NL + q) % NR] > 0.f ? t : 0.f;
              if (PRO == BXPRO_LN) {
                t = cok ? t - shift[q] : 0.f;
                s1[q] += t;
                s2[q] += t * t;
              } else {
                t = cok ? t : 0.f;
              }
              if (PRO == BXPRO_GELU) t = gelu_f(t);
            }
            x[e] = t;
          }
          bx_split<NTB>(x, bop[q]);
        }
        // ---- refill the slots just consumed (group g + RD) ----
#pragma unroll
        for (int e = 0; e < SPG; ++e) {
          const int sn = (g + RD) * SPG + e;
          fetch(sn < nslots ? sn : nslots - 1, ring[rd * SPG + e]);
        }
        // ---- MFMAs ----
        const int al = gl + rd;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          bx8 aop[NTA];
#pragma unroll
          for (int i = 0; i < NTA; ++i)
            aop[i] = *reinterpret_cast<const bx8*>(&Ab[((al * MB + mb) * NTA + i) * 64 * kItemHalfs + lane * kItemHalfs]);
#pragma unroll
          for (int q = 0; q < NACC; ++q) bx_mfma<NTA, NTB>(acc[mb][q], aop, bop[q]);
        }
      }
    }
    if (more) {
      store_chunk(cbuf ^ 1);
      cbuf ^= 1;
      __syncthreads();
    }
  }

  float mu_d[NACC], rstd[NACC];
  if (PRO == BXPRO_LN) {
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
  const int64_t ncol = S2D ? n0 + 2 * j : col_off;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    if (PRO == BXPRO_LN) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float sw = sW[mb * 32 + rl];
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[mb][q][r] = rstd[q] * (acc[mb][q][r] - mu_d[q] * sw);
      }
    }
    store_block<NACC, EPI, S2D>(p, acc[mb], b, m0 + mb * 32, ncol, h, PRO == BXPRO_LN ? tW + mb * 32 : nullptr);
  }
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
    int ks = (a.Ncol * a.B <= 65536) ? 1 : 0;
    int kn = 2, km = mblocks >= 2 ? 2 : 1;
    auto wgs = [&](int na, int mbb) { return ((a.Ncol + 32 * na - 1) / (32 * na)) * a.B * ((mblocks + mbb - 1) / mbb); };
    if (wgs(kn, km) < 512 && km == 2) km = 1;
    if (wgs(kn, km) < 512 && loader != LOAD_S2D) kn = 1;
    const char* e = getenv("FZ_BX_KS");  // diagnostics: "0" | "1" | "1<nacc><mb>"
    if (e && e[0]) { ks = e[0] - '0'; if (e[1] && e[2]) { kn = e[1] - '0'; km = e[2] - '0'; if (km > mblocks) km = 1; if (loader == LOAD_S2D) kn = 2; } }
    if (ks) {
      const int rc = gemm_bxk_launch<AT>(a, loader, epilogue, pro, kn, km, stream);
      if (rc != FZ_E_UNSUPPORTED) return rc;   // shapes outside the K-split form: the streaming kernel below
    }
  }
  if (loader == LOAD_S2D) {
    nacc = 2;
    mb = mblocks >= 2 ? 2 : 1;
    auto wgs = [&](int mbb) { return ((a.Ncol + 255) / 256) * a.B * ((mblocks + mbb - 1) / mbb); };
    if (mb == 2 && wgs(2) < 256) mb = 1;
  } else {
    // widest tile that still gives every CU about two workgroups
    nacc = 4; mb = mblocks >= 2 ? 2 : 1;
    auto wgs = [&](int na, int mbb) { return ((a.Ncol + 128 * na - 1) / (128 * na)) * a.B * ((mblocks + mbb - 1) / mbb); };
    if (wgs(nacc, mb) < 512) nacc = 2;
    if (wgs(nacc, mb) < 512 && mb == 2) mb = 1;
    if (wgs(nacc, mb) < 256) nacc = 1;
    const char* e = getenv("FZ_BX_CFG");  // diagnostics: "<nacc><mb>"
    if (e && e[0] && e[1]) { nacc = e[0] - '0'; mb = e[1] - '0'; if (mb > mblocks) mb = 1; }
    if (epilogue == EPI_D2S && nacc == 1) nacc = 2;
  }
  const int TN = 32 * nacc;
  const int64_t tiles = (a.Ncol + TN * 4 - 1) / (TN * 4);
  const int ygr = (mblocks + mb - 1) / mb;
  a.ygroups = (ygr > 1 && ygr <= 8 && tiles * a.B >= 64) ? ygr : 0;
  a.xtiles = (int)(tiles * a.B);
  dim3 grid((unsigned)(tiles * a.B), (unsigned)ygr), block(256);
  if (a.ygroups > 1) grid = dim3((unsigned)(((tiles * a.B + 7) / 8) * 8 * ygr), 1);
#define FZ_BX(MBv, NAv, L, E, PR, RDv) hipLaunchKernelGGL((gemm_bx_kernel<MBv, NAv, L, E, PR, RDv, AT>), grid, block, 0, st, a)
#define FZ_BX_TILES(L, E, PR)                                                   \
  do {                                                                          \
    if (nacc == 4) { if (mb == 2) FZ_BX(2, 4, L, E, PR, 1); else FZ_BX(1, 4, L, E, PR, 1); } \
    else if (nacc == 2) { if (mb == 2) FZ_BX(2, 2, L, E, PR, 2); else FZ_BX(1, 2, L, E, PR, 2); } \
    else { if (mb == 2) FZ_BX(2, 1, L, E, PR, 2); else FZ_BX(1, 1, L, E, PR, 2); } \
  } while (0)
  if (loader == LOAD_S2D) {
    if (mb == 2) FZ_BX(2, 2, LOAD_S2D, EPI_PLAIN, BXPRO_NONE, 4); else FZ_BX(1, 2, LOAD_S2D, EPI_PLAIN, BXPRO_NONE, 4);
  } else if (epilogue == EPI_D2S) {
    if (nacc == 4) { if (mb == 2) FZ_BX(2, 4, LOAD_PLAIN, EPI_D2S, BXPRO_NONE, 1); else FZ_BX(1, 4, LOAD_PLAIN, EPI_D2S, BXPRO_NONE, 1); }
    else { if (mb == 2) FZ_BX(2, 2, LOAD_PLAIN, EPI_D2S, BXPRO_NONE, 2); else FZ_BX(1, 2, LOAD_PLAIN, EPI_D2S, BXPRO_NONE, 2); }
  } else if (pro == BXPRO_LN) FZ_BX_TILES(LOAD_PLAIN, EPI_PLAIN, BXPRO_LN);
  else if (pro == BXPRO_GELU) FZ_BX_TILES(LOAD_PLAIN, EPI_PLAIN, BXPRO_GELU);
  else if (pro == BXPRO_BMUL) FZ_BX_TILES(LOAD_PLAIN, EPI_PLAIN, BXPRO_BMUL);
  else FZ_BX_TILES(LOAD_PLAIN, EPI_PLAIN, BXPRO_NONE);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

template int gemm_bx_launch<float>(const GemmArgsT<float>&, int, int, int, fz_stream_t);
template int gemm_bx_launch<bf16>(const GemmArgsT<bf16>&, int, int, int, fz_stream_t);

}  // namespace fz
