// upcat.hip — decoder level forward in ONE pass:  out = W_a·skip + D2S(W_bt·deep) + bias'
//
// Replaces, for the full-resolution decoder level (C = 32 adapter outputs, 32 skip channels, 64 deep channels):
//   x1  = ConvTranspose3d(k2, s2)(deep)              factorizer/unet.py:125   (launch 1: writes the up-sampled tensor)
//   out = adapter(cat([skip, x1], 1))                factorizer/unet.py:126-127, factorizer.py:116   (launch 2: reads it)
// Both are linear with nothing in between, so  out[m][fine(n, tap)] = Σ_c W_a[m][c]·skip[c][fine] + Σ_k Wbt[tap][m][k]·deep[k][n]
// + bias'[m]  with the composed weights  Wbt[tap][m][k] = Σ_c W_b[m][c]·W_t[k][c][tap]  and  bias' = b_ad + W_b·b_t  (formed by
// the host wrapper, a few kflop).  The up-sampled tensor (one full-resolution activation tensor written and read back) is
// never formed: 2.5 U of traffic instead of 4.5 U at this level.
//
// Mapping.  A wave owns 64 consecutive fine voxels of one (d, h) row (W_fine % 64 == 0): lane (j, half) holds the two
// voxels w = w0 + 2j + q, q = 0, 1 — the two fine voxels of ONE coarse voxel along W, i.e. tap tw = q — so the deep
// operand is loaded once per lane and channel and multiplied with the weights of tap (td, th, q) into accumulator q:
// no masked or duplicated MFMA.  (td, th) = (d & 1, h & 1) is wave-uniform per tile.  All eight taps of Wbt and W_a sit
// pre-split in LDS (gemm_bx.h: six exact bf16 products per fp32 product; bf16 storage: the activations are exact single
// terms); operand reads from LDS through one opaque per-lane base per image (see gemm_chain64).
#include "gemm_bx.h"
#include "finish.h"

namespace fz {

template <typename AT>
struct UpcatArgsT {
  const AT* skip;     // (B, 32, 2D, 2H, 2W)
  const AT* deep;     // (B, 64, D, H, W)
  const float* wa;    // W_a[m][c]: (32, lda)
  const float* wbt;   // Wbt[tap][m][k]: (8, 32, 64)
  const float* bias;  // bias'[m] or null
  AT* out;            // (B, 32, 2D, 2H, 2W)
  int lda;
  int B, D, H, W;     // coarse extent
  BlockPrologueArgs pro;   // (PRO) the consuming FactorizerBlock's LayerNorm 1 + in_proj + ReLU, applied to `out` in registers
};

// One PERSISTENT workgroup of 256 threads per CU (one wave per SIMD): W_a and all eight taps of Wbt are split into bf16
// levels and staged ONCE (102 KB of LDS), every wave then walks 64-voxel tiles with a stride of the grid, the operands of
// its next TWO tiles in flight while the products of the current one run.  (A first form staged four taps per 256-voxel
// workgroup: the staging — 1 152 weight items per tile of 24 KB of operand data — was half of the kernel.)
// Four waves, not eight: the same program as a 512-thread workgroup (two waves of one workgroup per SIMD) returned run-to-run
// different values in a few hundred elements at 128^3 — also with the tap images beyond 64 KB unused — while the 256-thread
// form replays bit for bit (tools/probes/upcat_check.py, profiles/r03_two_stream_interaction.md); cause not found.
#ifndef UPCAT_WAVES   // (-DUPCAT_WAVES=8: the eight-wave form, for tools/probes/upcat_variants.sh only)
#define UPCAT_WAVES 4
#endif
#define UPCAT_THREADS (UPCAT_WAVES * 64)
template <typename AT, bool PRO = false>
__global__ __launch_bounds__(UPCAT_THREADS, 1) void upcat_bx_kernel(UpcatArgsT<AT> p, unsigned ntiles) {
  constexpr int C = 32, CD = 64;
  constexpr int GS = C / 16, GD = CD / 16;          // K16-groups of the two segments
  constexpr int NTB = bx_terms_b<AT>(BXPRO_NONE);   // 3 (fp32 storage) / 1 (bf16 storage: exact)
  // LDS images: [slot][level][lane] x 16 B; skip: slot = g (2); deep: slot = GS + tap * GD + g, tap = td*4 + th*2 + tw (32)
  extern __shared__ __attribute__((aligned(16))) float As[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 31, hk = lane >> 5;
  const int Wf = 2 * p.W, Hf = 2 * p.H, Df = 2 * p.D;
  const int64_t Vf = (int64_t)Df * Hf * Wf, Vc = (int64_t)p.D * p.H * p.W;
  const unsigned tiles_per_sample = (unsigned)(Vf / 64);   // 32-bit index arithmetic throughout (fz_upcat_supported bounds the extent): the
                                                             // 64-bit scalar divisions were a visible share of a tile with one wave per SIMD

  // ---- weights -> three bf16 levels -> LDS (once per workgroup) ----
  for (int item = threadIdx.x; item < (GS + 8 * GD) * 64; item += UPCAT_THREADS) {
    const int l = item & 63, slot = item >> 6;
    const int m = l & 31, kh = l >> 5;
    float a8[8];
    if (slot < GS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) a8[e] = p.wa[(int64_t)m * p.lda + 16 * slot + 8 * kh + e];
    } else {
      const int s2 = slot - GS, tap = s2 / GD, g = s2 % GD;
#pragma unroll
      for (int e = 0; e < 8; ++e) a8[e] = p.wbt[((int64_t)tap * 32 + m) * CD + 16 * g + 8 * kh + e];
    }
    bx8 t3[3];
    bx_split<3>(a8, t3);
    bx8* dst = reinterpret_cast<bx8*>(As) + (slot * 3) * 64 + l;
    dst[0] = t3[0]; dst[64] = t3[1]; dst[128] = t3[2];
  }
  // (PRO) behind the 34 weight slots: the in_proj image (2 groups x 3 levels x 64 lanes x 16 B = 6 KB) and (W β)[32]
  __bf16* Apro = reinterpret_cast<__bf16*>(As + (GS + 8 * GD) * 3 * 256);
  float* twp = As + (GS + 8 * GD) * 3 * 256 + 2 * 3 * 256;
  if constexpr (PRO) ln_inproj_stage(Apro, twp, p.pro, (int)threadIdx.x, UPCAT_THREADS);
  __syncthreads();
  float badd[16];   // the bias' entries of this lane's 16 output rows
#pragma unroll
  for (int r = 0; r < 16; ++r) badd[r] = p.bias != nullptr ? p.bias[(r & 3) + 8 * (r >> 2) + 4 * hk] : 0.f;

  int lane4 = lane * 4, lane4d = lane * 4 + GS * 3 * 256;   // opaque per-lane float indices of the two images (64 KB immediates)
  asm volatile("" : "+v"(lane4));
  asm volatile("" : "+v"(lane4d));

  // ---- operand loads of one tile: skip (two fine voxels per lane and channel), deep (their one coarse voxel) ----
  typedef float XsT[GS][8][2];
  typedef float XdT[GD][8];
  auto fetch = [&](unsigned t, XsT& xs, XdT& xd) {
    const unsigned b = t / tiles_per_sample;
    const unsigned n0 = (t - b * tiles_per_sample) * 64u;
    const unsigned row = n0 / (unsigned)Wf;
    const unsigned w0 = n0 - row * (unsigned)Wf;
    const unsigned df = row / (unsigned)Hf, hf = row - df * (unsigned)Hf;
    const unsigned soff = (unsigned)((((int64_t)8 * hk) * Vf + n0 + 2 * j) * (int64_t)sizeof(AT));
    const int64_t ncoarse = ((int64_t)(df >> 1) * p.H + (hf >> 1)) * p.W + (w0 >> 1) + j;
    const unsigned doff = (unsigned)((((int64_t)8 * hk) * Vc + ncoarse) * (int64_t)sizeof(AT));
#pragma unroll
    for (int g = 0; g < GS; ++g) {
      const AT* ub = p.skip + ((int64_t)b * C + 16 * g) * Vf;
#pragma unroll
      for (int e = 0; e < 8; ++e) uload<2>(ub + (int64_t)e * Vf, soff, xs[g][e]);
    }
#pragma unroll
    for (int g = 0; g < GD; ++g) {
      const AT* ub = p.deep + ((int64_t)b * CD + 16 * g) * Vc;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t1[1];
        uload<1>(ub + (int64_t)e * Vc, doff, t1);
        xd[g][e] = t1[0];
      }
    }
  };

  const unsigned tstep = gridDim.x * UPCAT_WAVES;
  // The row operands live in REGISTERS: W_a (6 x 16 B per lane) for the whole walk, the two taps (td, th, tw = 0 / 1) of Wbt
  // (24 x 16 B) until a tile with another (td, th) comes — with one wave per SIMD nothing would hide an LDS read per product
  // group, and the tiles a wave walks are a whole number of row pairs apart in the common extents, so the reload is rare.
  bx8 wsk[GS][3], wdp[2][GD][3];
#pragma unroll
  for (int g = 0; g < GS; ++g)
#pragma unroll
    for (int t = 0; t < 3; ++t) wsk[g][t] = *reinterpret_cast<const bx8*>(As + (g * 3 + t) * 256 + lane4);
  int tap_cur = -1;
  // one tile: split the operands, refill their registers with the tile two steps ahead, products, epilogue
  auto run = [&](unsigned tile, XsT& xs, XdT& xd) {
    const unsigned b = tile / tiles_per_sample;
    const unsigned n0 = (tile - b * tiles_per_sample) * 64u;
    const unsigned row = n0 / (unsigned)Wf;
    const unsigned df = row / (unsigned)Hf, hf = row - df * (unsigned)Hf;
    const int tap0 = __builtin_amdgcn_readfirstlane(((df & 1) * 4 + (hf & 1) * 2) * GD);   // slot of (td, th, tw = 0, g = 0) inside the deep image
    if (tap0 != tap_cur) {   // wave-uniform
      tap_cur = tap0;
      const float* Ad = As + tap0 * 3 * 256;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int g = 0; g < GD; ++g)
#pragma unroll
          for (int t = 0; t < 3; ++t) wdp[q][g][t] = *reinterpret_cast<const bx8*>(Ad + ((q * GD + g) * 3 + t) * 256 + lane4d);
    }

    bx8 bs[GS][2][NTB], bd[GD][NTB];
#pragma unroll
    for (int g = 0; g < GS; ++g)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float x8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x8[e] = xs[g][e][q];
        bx_split<NTB>(x8, bs[g][q]);
      }
#pragma unroll
    for (int g = 0; g < GD; ++g) bx_split<NTB>(xd[g], bd[g]);
    if (tile + 2 * tstep < ntiles) fetch(tile + 2 * tstep, xs, xd);

    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    // ---- skip segment: out += W_a · skip ----
#pragma unroll
    for (int g = 0; g < GS; ++g) {
#pragma unroll
      for (int t = 2; t >= 0; --t) {
        const bx8 a = wsk[g][t];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int jj = NTB - 1; jj >= 0; --jj)
            if (t + jj <= 2) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bs[g][q][jj], acc[q], 0, 0, 0);
      }
    }
    // ---- deep segment: out[.., voxel q] += Wbt[tap (td, th, q)] · deep[coarse voxel] ----
#pragma unroll
    for (int g = 0; g < GD; ++g) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 2; t >= 0; --t) {
          const bx8 a = wdp[q][g][t];
#pragma unroll
          for (int jj = NTB - 1; jj >= 0; --jj)
            if (t + jj <= 2) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bd[g][jj], acc[q], 0, 0, 0);
        }
    }

#ifdef FZ_UPCAT_NOP   // probe: extra wait states between the last MFMA and the first reader of its result
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
#endif
    // ---- epilogue: rows (r & 3) + 8 (r >> 2) + 4 hk, the lane's two voxels as one 8-byte (4-byte) store ----
    const unsigned yoff = (unsigned)((((int64_t)4 * hk) * Vf + n0 + 2 * j) * (int64_t)sizeof(AT));
    AT* yb = p.out + (int64_t)b * C * Vf;
    float yv[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rb = (r & 3) + 8 * (r >> 2);
      const float add = badd[r];
      float v[2] = {acc[0][r] + add, acc[1][r] + add};
      vstore<2>(reinterpret_cast<AT*>(reinterpret_cast<char*>(yb + (int64_t)rb * Vf) + yoff), v);
      if constexpr (PRO) {   // bf16 storage: the block's first layer sees the STORED value, as a separate launch would
        yv[0][r] = sizeof(AT) == 2 ? (float)(AT)v[0] : v[0];
        yv[1][r] = sizeof(AT) == 2 ? (float)(AT)v[1] : v[1];
      }
    }
    if constexpr (PRO) {
      float mu[2], rs[2];
      f32x16 tacc[2];
      ln_inproj_tile<2>(yv, Apro, p.pro.ln_eps, lane, mu, rs, tacc);
      AT* tb = reinterpret_cast<AT*>(p.pro.t) + (int64_t)b * C * Vf;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rb = (r & 3) + 8 * (r >> 2);
        const float add = twp[rb + 4 * hk];
        const float v[2] = {fmaxf(tacc[0][r] + add, 0.f), fmaxf(tacc[1][r] + add, 0.f)};
        vstore<2>(reinterpret_cast<AT*>(reinterpret_cast<char*>(tb + (int64_t)rb * Vf) + yoff), v);
      }
      if (hk == 0) {
        float* so = p.pro.stats + (int64_t)b * 2 * Vf + n0 + 2 * j;
        *reinterpret_cast<float2*>(so) = make_float2(mu[0], mu[1]);
        *reinterpret_cast<float2*>(so + Vf) = make_float2(rs[0], rs[1]);
      }
    }
  };

  // two tiles of operands in flight per wave (one wave per SIMD: up to 512 registers are this wave's)
  XsT xsA, xsB;
  XdT xdA, xdB;
  unsigned tile = blockIdx.x * UPCAT_WAVES + (unsigned)wave;
  if (tile < ntiles) fetch(tile, xsA, xdA);
  if (tile + tstep < ntiles) fetch(tile + tstep, xsB, xdB);
  for (; tile < ntiles; tile += 2 * tstep) {
    run(tile, xsA, xdA);
    if (tile + tstep < ntiles) run(tile + tstep, xsB, xdB);
  }
}

// ---- the weight compositions of the node (tiny: a few hundred kflop .. Mflop; one launch each instead of ~10 framework ops) ----
// wc[k][m][t] = Σ_c w_t[k][c][t]·w_b[m][c]  (backward: correlation weights),  wbt[t][m][k] = the same value (forward layout),
// bias'[m] = b_ad[m] + Σ_c w_b[m][c]·b_t[c].  One workgroup per deep channel k (+ one for the bias).
__global__ __launch_bounds__(256) void upcat_compose_kernel(const float* __restrict__ w_t, const float* __restrict__ w_b, int ldb,
                                                            const float* __restrict__ b_t, const float* __restrict__ b_ad,
                                                            float* __restrict__ wc, float* __restrict__ wbt,
                                                            float* __restrict__ bias, int Cd, int O, int M) {
  const int k = blockIdx.x;
  if (k == Cd) {
    if (bias == nullptr) return;
    for (int m = threadIdx.x; m < M; m += 256) {
      float t = b_ad != nullptr ? b_ad[m] : 0.f;
      if (b_t != nullptr)
        for (int c = 0; c < O; ++c) t += w_b[(int64_t)m * ldb + c] * b_t[c];
      bias[m] = t;
    }
    return;
  }
  const float* wk = w_t + (int64_t)k * O * 8;
  for (int i = threadIdx.x; i < M * 8; i += 256) {
    const int m = i >> 3, t = i & 7;
    float acc = 0.f;
    for (int c = 0; c < O; ++c) acc += wk[c * 8 + t] * w_b[(int64_t)m * ldb + c];
    if (wc != nullptr) wc[((int64_t)k * M + m) * 8 + t] = acc;
    if (wbt != nullptr) wbt[((int64_t)t * M + m) * Cd + k] = acc;
  }
}

// (dW_t, dW_b, db_t from the (deep x g) correlation: the FK_UPCAT job of the finish kernel, finish.h — phase 1: it reads the
// correlation and the adapter bias gradient, both outputs of phase-0 finish jobs)

}  // namespace fz

using namespace fz;

extern "C" int fz_upcat_compose(const float* w_t, const float* w_b, int ldb, const float* b_t, const float* b_ad, float* wc, float* wbt,
                                float* bias, int Cd, int O, int M, fz_stream_t stream) {
  if (!w_t || !w_b || (!wc && !wbt && !bias) || Cd < 1 || O < 1 || M < 1 || ldb < O) return fail(FZ_E_ARG, "fz_upcat_compose: bad arguments");
  hipLaunchKernelGGL(upcat_compose_kernel, dim3((unsigned)(Cd + 1)), dim3(256), 0, (hipStream_t)stream, w_t, w_b, ldb, b_t, b_ad, wc, wbt,
                     bias, Cd, O, M);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_upcat_wgrads(const float* gt, const float* w_t, const float* w_b, int ldb, const float* gb_ad, const float* b_t,
                               float* gw_t, float* gw_b, int ldg, float* gb_t, int Cd, int O, int M, fz_stream_t stream) {
  if (!gt || !w_t || !w_b || !gb_ad || !gw_t || !gw_b || Cd < 1 || O < 1 || M < 1 || ldb < O || ldg < O)
    return fail(FZ_E_ARG, "fz_upcat_wgrads: bad arguments");
  FinishJob fj = finish_job(FK_UPCAT, fin_upcat_blocks(Cd, O, M), 1);
  fj.u.up = FinUpcat{gt, w_t, w_b, gb_ad, b_t, gw_t, gw_b, gb_t, ldb, ldg, Cd, O, M};
  return finish_run(&fj, 1, (hipStream_t)stream);
}

extern "C" int fz_upcat_supported(int C, int Cd, int D, int H, int W) {
  // (coarse extent) fine rows of 2W voxels in 64-voxel wave tiles, 256-voxel workgroups inside one d
  return C == 32 && Cd == 64 && D >= 1 && H >= 1 && W >= 1 && (2 * W) % 64 == 0 && ((int64_t)2 * H * 2 * W) % 256 == 0 &&
         (int64_t)64 * D * H * W * 8 * 4 < ((int64_t)1 << 31);
}

template <typename AT>
static int upcat_launch(const void* skip, const void* deep, const float* wa, int lda, const float* wbt, const float* bias, void* out,
                        int B, int D, int H, int W, const fz_block_prologue* pro, fz_stream_t stream) {
  UpcatArgsT<AT> a;
  a.pro = BlockPrologueArgs{};
  if (pro) a.pro = BlockPrologueArgs{pro->ln_g, pro->ln_b, pro->ln_eps, pro->w, pro->t, pro->stats};
  a.skip = (const AT*)skip; a.deep = (const AT*)deep; a.wa = wa; a.wbt = wbt; a.bias = bias; a.out = (AT*)out;
  a.lda = lda; a.B = B; a.D = D; a.H = H; a.W = W;
  const int64_t ntiles = (int64_t)8 * D * H * W / 64 * B;   // 64-voxel wave tiles
  if (ntiles >= ((int64_t)1 << 30)) return fail(FZ_E_ARG, "fz_upcat: more than 2^30 wave tiles (32-bit tile arithmetic)");
  constexpr int lds = (2 + 8 * 4) * 3 * 64 * 16 + 2 * 3 * 64 * 16 + 32 * 4;   // 34 slots x 3 levels x 64 lanes x 16 B = 102 KB (+ the prologue's 6 KB)
  const int64_t wgs = (ntiles + UPCAT_WAVES - 1) / UPCAT_WAVES;
  if (pro) {
    auto kern = upcat_bx_kernel<AT, true>;
    FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)(wgs < 256 ? wgs : 256)), dim3(UPCAT_THREADS), lds, (hipStream_t)stream, a, (unsigned)ntiles);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }
  auto kern = upcat_bx_kernel<AT>;
  FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)(wgs < 256 ? wgs : 256)), dim3(UPCAT_THREADS), lds, (hipStream_t)stream, a, (unsigned)ntiles);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_upcat2(const void* skip, const void* deep, const float* wa, int lda, const float* wbt, const float* bias, void* out,
                         int B, int C, int Cd, int D, int H, int W, int act_dtype, const fz_block_prologue* pro, fz_stream_t stream) {
  if (!skip || !deep || !wa || !wbt || !out) return fail(FZ_E_ARG, "fz_upcat: null pointer");
  if (pro && (!pro->ln_g || !pro->ln_b || !pro->w || !pro->t || !pro->stats)) return fail(FZ_E_ARG, "fz_upcat2: incomplete block prologue");
  if (!fz_upcat_supported(C, Cd, D, H, W) || lda < C) return fail(FZ_E_UNSUPPORTED, "fz_upcat: needs C = 32, 64 deep channels, 2W % 64 == 0, 4HW % 256 == 0");
  if (B <= 0) return B == 0 ? FZ_OK : fail(FZ_E_SHAPE, "fz_upcat: negative batch");
  if (act_dtype == FZ_STORE_F32) return upcat_launch<float>(skip, deep, wa, lda, wbt, bias, out, B, D, H, W, pro, stream);
  if (act_dtype == FZ_STORE_BF16) return upcat_launch<bf16>(skip, deep, wa, lda, wbt, bias, out, B, D, H, W, pro, stream);
  return fail(FZ_E_ARG, "fz_upcat: act_dtype must be FZ_STORE_F32 or FZ_STORE_BF16");
}

extern "C" int fz_upcat(const void* skip, const void* deep, const float* wa, int lda, const float* wbt, const float* bias, void* out,
                        int B, int C, int Cd, int D, int H, int W, int act_dtype, fz_stream_t stream) {
  return fz_upcat2(skip, deep, wa, lda, wbt, bias, out, B, C, Cd, D, H, W, act_dtype, nullptr, stream);
}
