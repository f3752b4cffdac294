// nmf_cf_gram.hip — backward of the fused FactMixer core for HALS rank 1 behind a ReLU, in the row space (nmf_gram.h).
// Replaces autograd through SWMatricize.forward → NMF(rank 1, "hals") → SWMatricize.inverse_forward
// (factorizer/factorizer.py:41-50; operations.py:417-434; factorization/matrix_factorization.py:210-229,506-533).
//
// relu_gate = 1 promises t = relu(z) >= 0 (factorizer.py:44), which is what lets the iteration be carried by K = X Xᵀ:
// 2 450 instead of 3 600 vector instructions per matrix and no per-column history in LDS (12.7 KB per wave in the general
// kernel: two workgroups per CU).  dL/dY is consumed row by row as it comes out of the exchange, dL/dX is produced row by
// row into it — neither ever occupies 64 registers next to X: <= 168 registers, three workgroups of four waves per CU.
// Same tile / exchange / store scheme as cf_bwd_tile_body (nmf_cf.hip).
//
// What the launch is bound by (tools/probes/gram_floor.sh, profiles/r05_gram_floor.md): with the arithmetic compiled out
// (-DFZ_PROBE_GRAM_NOMATH: loads, exchanges, stores) it takes 345-357 us for window 0 and 485-493 us for window 1 at stage
// 0, the full kernel 360-370 / 478-487 — the tile's memory skeleton, not the wave program, is what the time is.  Variants
// of how the reads are requested that did not change it: channels 4-7 of dL/da and of the running sum by LDS-DMA
// (global_load_lds_dwordx4 into a swizzled 32 KB image, everything in flight from kernel start), two or three waves per
// SIMD.  Channel planes 4 KiB further apart than the dense 8 MiB: window 0 -10 %.
//
// MODE: CFG_HALVES (fp32) — dL/da four channels at a time through 32 registers, channels 0-3 requested once the 52 partial
// sums of K and s are reduced, 4-7 when 0-3 have left for the exchange; CFG_RAW (bf16 storage) — a lane's four elements are
// 8 bytes: all 8 channels as raw pairs in 32 registers at once.
#include "nmf_cf.h"
#include "nmf_gram.h"

#ifndef FZ_GRAM_WAVES
#define FZ_GRAM_WAVES 3
#endif

namespace fz {

enum { CFG_HALVES = 0, CFG_RAW = 2 };

#ifdef FZ_PROBE_GRAM_NOMATH
#define FZ_GRAM_OUT_ROW(m, grow) probe_acc += grow[(m) & 7]
#else
#define FZ_GRAM_OUT_ROW(m, grow) P.out_row(m, grow)
#endif

// 4 consecutive elements as they lie in memory (no conversion): float4 for fp32, two dwords for bf16
template <typename AT> struct CfRaw;
template <> struct CfRaw<float> { using T = float4; };
template <> struct CfRaw<bf16> { using T = uint2; };

template <bool HALF>
__device__ __forceinline__ float4 cf_ld_raw(const float* p, unsigned o, unsigned o2) { return cf_ld4<HALF>(p, o, o2); }
template <bool HALF>
__device__ __forceinline__ uint2 cf_ld_raw(const bf16* p, unsigned o, unsigned o2) {
  if (HALF) {
    const unsigned a = *reinterpret_cast<const unsigned*>(cf_at(p, o)), b = *reinterpret_cast<const unsigned*>(cf_at(p, o2));
    return make_uint2(a, b);
  }
  return *reinterpret_cast<const uint2*>(cf_at(p, o));
}
__device__ __forceinline__ float4 cf_raw_f4(const float4& v) { return v; }
__device__ __forceinline__ float4 cf_raw_f4(const uint2& v) {
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}

template <int WPB, bool HALF, typename AT, int MODE>
__global__ __launch_bounds__(WPB * 64, WPB == 4 ? FZ_GRAM_WAVES : 1) void nmf_cf_bwd_gram_kernel(
    const AT* __restrict__ t, const float* __restrict__ v0, const AT* __restrict__ ga, AT* __restrict__ gt, CfGeom q,
    int T, int G, float eps, int xcd_remap) {
  extern __shared__ __attribute__((aligned(16))) float fz_lds_cfg[];
  using TL = CfTile<WPB>;
  using Raw = typename CfRaw<AT>::T;
  constexpr int NB = MODE == CFG_RAW ? 8 : 4;            // channels per register batch
  constexpr int IMG = 64 * TL::LW;                      // floats of one channel image
  float* S = fz_lds_cfg;
  const CfTileId id = cf_tile_id<WPB>(q, cf_logical_block(xcd_remap));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t base, V;
  unsigned off[2], off2[2];
  int lidx[2];
  cf_tile_decode<WPB>(q, id, tid, base, V, off, lidx, off2);
  const int own0 = cf_owner_lidx<WPB>(lane, wave, 0), own1 = cf_owner_lidx<WPB>(lane, wave, 1);
  float* hist = S + TL::STAGE_FLOATS + wave * gram_hist_floats(G - 1);

  float x[8][8];
#pragma unroll
  for (int dd = 0; dd < 8; ++dd)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 v = cf_ld4<HALF>(t + base + dd * V, off[k], off2[k]);
      x[dd][k * 4 + 0] = v.x; x[dd][k * 4 + 1] = v.y; x[dd][k * 4 + 2] = v.z; x[dd][k * 4 + 3] = v.w;
    }
  cf_to_owner<WPB>(S, lidx, own0, own1, x);

  CfWave w{lane};
  GramBwd<8, CfWave> P;
  Raw gq[NB][2];
  auto request_g = [&]() {
#pragma unroll
    for (int dd = 0; dd < NB; ++dd)
#pragma unroll
      for (int k = 0; k < 2; ++k) gq[dd][k] = cf_ld_raw<HALF>(ga + base + dd * V, off[k], off2[k]);
  };
#ifdef FZ_PROBE_GRAM_NOMATH   // timing probe (tools/probes/gram_floor.sh): loads, exchanges and stores only
  float probe_acc = 0.f;
  request_g();
#else
  P.forward(w, x, v0, 8, 512, T, G, eps, hist, request_g);
#endif
  asm volatile("" : "+s"(base));
  // dL/da: coalesced -> owner through the stage, two rows at a time, consumed at once
#pragma unroll
  for (int bt = 0; bt < 8 / NB; ++bt) {
#pragma unroll
    for (int s = 0; s < NB / 2; ++s) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < 2; ++k) *reinterpret_cast<float4*>(S + c * IMG + lidx[k]) = cf_raw_f4(gq[2 * s + c][k]);
      if (MODE == CFG_HALVES && bt == 0 && s == NB / 2 - 1) {
#pragma unroll
        for (int dd = 0; dd < NB; ++dd)
#pragma unroll
          for (int k = 0; k < 2; ++k) gq[dd][k] = cf_ld_raw<HALF>(ga + base + (NB + dd) * V, off[k], off2[k]);
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float4 a0 = *reinterpret_cast<const float4*>(S + c * IMG + own0);
        const float4 a1 = *reinterpret_cast<const float4*>(S + c * IMG + own1);
        const float grow[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        FZ_GRAM_OUT_ROW(NB * bt + 2 * s + c, grow);
      }
      __syncthreads();
    }
  }
  // The running sum of the earlier windows is requested stage by stage, right in front of the rows it is added to
  // (requested earlier — by DMA when dL/da has been consumed, or inside the reverse sweep — window 1 took 10-25 us longer)
#ifdef FZ_PROBE_GRAM_NOMATH
#else
  P.reverse(w, x, 1.0f / q.gscale_div, [] {});
#endif
#pragma unroll
  for (int bt = 0; bt < 8 / NB; ++bt) {
#pragma unroll
    for (int s = 0; s < NB / 2; ++s) {
      Raw old[2][2];
      if (q.accumulate) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int k = 0; k < 2; ++k) old[c][k] = cf_ld_raw<HALF>(gt + base + (NB * bt + 2 * s + c) * V, off[k], off2[k]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int m = NB * bt + 2 * s + c;
        float o[8];
#ifdef FZ_PROBE_GRAM_NOMATH
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = x[m][e] + probe_acc;
#else
        float srow[8], gsm, ga1m;
        P.row_coeffs(w, m, srow, gsm, ga1m);
        P.gx_row(m, x, srow, gsm, ga1m, o);
#endif
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = x[m][e] > 0.f ? o[e] : 0.f;
        *reinterpret_cast<float4*>(S + c * IMG + own0) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(S + c * IMG + own1) = make_float4(o[4], o[5], o[6], o[7]);
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          float4 z = *reinterpret_cast<const float4*>(S + c * IMG + lidx[k]);
          if (q.accumulate) {
            const float4 o = cf_raw_f4(old[c][k]);
            z.x += o.x; z.y += o.y; z.z += o.z; z.w += o.w;
          }
          cf_st4<HALF>(gt + base + (NB * bt + 2 * s + c) * V, off[k], off2[k], z);
        }
      __syncthreads();
    }
  }
}

template <typename AT>
int cf_bwd_gram_launch(const AT* t, const float* v0, const AT* ga, AT* gt, const CfGeom& q, int64_t nmat, int T, int G,
                       float eps, int xcd_remap, hipStream_t st) {
  const bool half = (q.s2 % 4) != 0;
#ifdef FZ_PROBE_GRAM_WPB   // timing probe (tools/probes/gram_tile.sh): FZ_PROBE_GRAM_WPB patches along W per workgroup instead of 4
  const int twpb = (q.G2 % FZ_PROBE_GRAM_WPB) == 0 ? FZ_PROBE_GRAM_WPB : ((q.G2 % 4) == 0 ? 4 : 1);
#else
  const int twpb = (q.G2 % 4) == 0 ? 4 : 1;
#endif
  constexpr bool kF32 = sizeof(AT) == 4;
#ifdef FZ_PROBE_GRAM_WPB
  const int stage = twpb == FZ_PROBE_GRAM_WPB ? CfTile<FZ_PROBE_GRAM_WPB>::STAGE_FLOATS : (twpb == 4 ? CfTile<4>::STAGE_FLOATS : CfTile<1>::STAGE_FLOATS);
#else
  const int stage = twpb == 4 ? CfTile<4>::STAGE_FLOATS : CfTile<1>::STAGE_FLOATS;
#endif
  const int glds = (stage + twpb * gram_hist_floats(G - 1)) * (int)sizeof(float);
#ifndef FZ_PROBE_GRAM_WPB
  if (glds > 64 * 1024) return FZ_E_UNSUPPORTED;
#endif
  const unsigned nblk = (unsigned)(nmat / twpb);
#define FZ_CF_BWD_GRAM(WW, HH, MM) \
  hipLaunchKernelGGL((nmf_cf_bwd_gram_kernel<WW, HH, AT, MM>), dim3(nblk), dim3(64 * WW), glds, st, t, v0, ga, gt, q, T, G, eps, xcd_remap)
  constexpr int kRegMode = kF32 ? CFG_HALVES : CFG_RAW;
#ifdef FZ_PROBE_GRAM_WPB
  if (!half && twpb == FZ_PROBE_GRAM_WPB) {
    auto kern = nmf_cf_bwd_gram_kernel<FZ_PROBE_GRAM_WPB, false, AT, kRegMode>;
    if (glds > 64 * 1024) FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, glds));
    FZ_CF_BWD_GRAM(FZ_PROBE_GRAM_WPB, false, kRegMode);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }
#endif
  if (half) { if (twpb == 4) FZ_CF_BWD_GRAM(4, true, kRegMode); else FZ_CF_BWD_GRAM(1, true, kRegMode); }
  else if (twpb == 4) FZ_CF_BWD_GRAM(4, false, kRegMode);
  else return FZ_E_UNSUPPORTED;
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}
template int cf_bwd_gram_launch<float>(const float*, const float*, const float*, float*, const CfGeom&, int64_t, int, int, float, int, hipStream_t);
template int cf_bwd_gram_launch<bf16>(const bf16*, const float*, const bf16*, bf16*, const CfGeom&, int64_t, int, int, float, int, hipStream_t);

}  // namespace fz
