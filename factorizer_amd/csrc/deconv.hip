// deconv.hip — grouped "same" cross-correlation for the Deconver family.
//
// The multiplicative updates of the reference's blind deconvolution layer
// (factorizer/factorization/deconvolution.py:136-156: s ← s ∘ (Hᵀx + ε)/(HᵀH s + ε)) are made of one operator,
//
//     out[b, g·Co + o, v] = Σ_i Σ_τ  in[b, g·Ci + i, v + τ − p] · w[b or 0, g, o, i, τ]        (zero padding, p = k/2)
//
// applied as H (filters h: Ci = sources, Co = C/G), as Hᵀ (channel-transposed, flipped filters) and — with the
// output-side gradient as input — as the input gradient of either.  The reference evaluates it by folding the batch
// into the groups of F.conv{2,3}d (deconvolution.py:21-40); its channel counts per group are tiny (1..16), far
// below an MFMA tile, and the op is a 27..343-tap stencil: a direct VALU kernel with the input tile (+ halo) of one
// channel at a time in LDS.
//
//   workgroup  = 256 threads = a 4 x 4 x 64 voxel tile of one (sample, group); thread = 4 voxels along W x ALL Co
//                outputs (Co·4 accumulators), so every staged input value is used Co·k³ times;
//   weights    = wave-uniform: read through the scalar cache (addresses depend on blockIdx only);
//   epilogue   = plain (+ε) or the fused multiplicative update  out = a ∘ b / (acc + ε).
// 2-D layers run as depth-1 volumes with kernel depth 1.
#include "fz_common.h"

namespace fz {

struct GcArgs {
  const float* in;    // (B, G·Ci, D, H, W)
  const float* w;     // (Bw, G, Co, Ci, KD, KH, KW), Bw in {1, B}
  float* out;         // (B, G·Co, D, H, W)
  const float* mul_a; // epilogue 1: (B, G·Co, D, H, W)
  const float* mul_b;
  int B, G, Ci, Co, D, H, W;
  int w_batched;      // 1: filters per sample
  int epilogue;       // 0: out = acc + add_eps ; 1: out = mul_a * mul_b / (acc + add_eps)
  float add_eps;
  int tiles_h, tiles_w;
};

constexpr int kTD = 4, kTH = 4, kTW = 64;

template <int CO, int KD, int KH, int KW>
__global__ __launch_bounds__(256) void gcorr_kernel(GcArgs a) {
  constexpr int PD = KD / 2, PH = KH / 2, PW = KW / 2;
  constexpr int LD = kTD + 2 * PD, LH = kTH + 2 * PH, LW = kTW + 2 * PW;
  constexpr int LWS = LW + 1;                       // row stride (odd: spreads the rows over the banks)
  __shared__ float tile[LD * LH * LWS];
  const int tid = threadIdx.x;
  const int tw = tid & 15, th = (tid >> 4) & 3, td = tid >> 6;
  int bid = blockIdx.x;
  const int twi = bid % a.tiles_w; bid /= a.tiles_w;
  const int thi = bid % a.tiles_h; bid /= a.tiles_h;
  const int tiles_d = (a.D + kTD - 1) / kTD;
  const int tdi = bid % tiles_d;
  const int g = blockIdx.y, b = blockIdx.z;
  const int d0 = tdi * kTD, h0 = thi * kTH, w0 = twi * kTW;
  const int64_t V = (int64_t)a.D * a.H * a.W;
  const float* inb = a.in + ((int64_t)b * a.G + g) * a.Ci * V;
  const float* wg = a.w + (((int64_t)(a.w_batched ? b : 0) * a.G + g) * a.Co) * a.Ci * (KD * KH * KW);

  float acc[CO][4];
#pragma unroll
  for (int o = 0; o < CO; ++o)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[o][e] = 0.f;

  for (int ci = 0; ci < a.Ci; ++ci) {
    __syncthreads();   // previous channel's tile fully consumed
    const float* inc = inb + (int64_t)ci * V;
    for (int idx = tid; idx < LD * LH * LW; idx += 256) {
      const int lw = idx % LW, lh = (idx / LW) % LH, ld = idx / (LW * LH);
      const int zd = d0 + ld - PD, zh = h0 + lh - PH, zw = w0 + lw - PW;
      const bool ok = zd >= 0 && zd < a.D && zh >= 0 && zh < a.H && zw >= 0 && zw < a.W;
      tile[(ld * LH + lh) * LWS + lw] = ok ? inc[((int64_t)zd * a.H + zh) * a.W + zw] : 0.f;
    }
    __syncthreads();
    // (the depth taps stay a rolled loop: CO·KD·KH·KW·4 FMAs fully unrolled is 88 K instructions at CO = 16, k = 7)
#pragma unroll 1
    for (int kd = 0; kd < KD; ++kd)
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
        float row[4 + KW - 1];
        const float* rp = tile + ((td + kd) * LH + (th + kh)) * LWS + tw * 4;
#pragma unroll
        for (int e = 0; e < 4 + KW - 1; ++e) row[e] = rp[e];
#pragma unroll
        for (int o = 0; o < CO; ++o) {
          if (o < a.Co) {
            const float* wp = wg + ((int64_t)o * a.Ci + ci) * (KD * KH * KW) + (kd * KH + kh) * KW;   // uniform
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
              const float wv = wp[kw];
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[o][e] = acc[o][e] + row[e + kw] * wv;
            }
          }
        }
      }
  }
  const int zd = d0 + td, zh = h0 + th, zw = w0 + tw * 4;
  if (zd >= a.D || zh >= a.H) return;
  const int64_t vo = ((int64_t)zd * a.H + zh) * a.W + zw;
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    if (o >= a.Co) continue;
    const int64_t base = (((int64_t)b * a.G + g) * a.Co + o) * V + vo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (zw + e >= a.W) continue;
      float r = acc[o][e] + a.add_eps;
      if (a.epilogue == 1) r = a.mul_a[base + e] * a.mul_b[base + e] / r;
      a.out[base + e] = r;
    }
  }
}


// ---- filter gradient -------------------------------------------------------------------------------
//   gw[bw, g, o, i, τ] = Σ_b Σ_v  gout[b, g·Co + o, v] · in[b, g·Ci + i, v + τ − p]     (b = bw when the filters are per sample)
// — the lag-correlation of the input with the output gradient, a reduction over all voxels.  The reference gets it
// from autograd through F.conv{2,3}d (deconvolution.py:21-40).  Two deterministic stages, no float atomics:
//   1. one workgroup per (sample, group, input channel, chunk of voxel tiles): the gout tile (all Co channels) and
//      the input tile (+ halo) sit in LDS; a work item is one (o, kd, kh) filter row × one slice of the tile's voxel
//      quads and accumulates the KW taps of that row; slices are added through LDS in slice order, chunks of tiles
//      in registers; the per-chunk sums go to the caller's workspace;
//   2. `gcorr_wgrad_finish_kernel` adds chunks (and samples, for shared filters) in index order.
struct GcwArgs {
  const float* in;    // (B, G·Ci, D, H, W)
  const float* gout;  // (B, G·Co, D, H, W)
  float* part;        // (B, G, nchunks, Co, Ci, KD·KH·KW)
  int B, G, Ci, Co, D, H, W;
  int tiles_d, tiles_h, tiles_w, nchunks, tiles_per_chunk;
};

constexpr int kGoStride = kTD * kTH * kTW + 4;   // gout tile row stride (floats): rows of different o on different banks

template <int CO, int KD, int KH, int KW>
__global__ __launch_bounds__(256) void gcorr_wgrad_kernel(GcwArgs a) {
  constexpr int PD = KD / 2, PH = KH / 2, PW = KW / 2;
  constexpr int LD = kTD + 2 * PD, LH = kTH + 2 * PH, LW = kTW + 2 * PW;
  constexpr int LWS = LW + 1;
  constexpr int P = CO * KD * KH;                       // filter rows
  constexpr int S = P >= 256 ? 1 : 256 / P;             // voxel-quad slices per row
  constexpr int NI = (P * S + 255) / 256;               // work items per thread
  constexpr int NR = P * KW;                            // sums per (input channel, chunk)
  constexpr int NS = (NR + 255) / 256;
  constexpr int NQ = kTD * kTH * kTW / 4;               // voxel quads per tile
  extern __shared__ __attribute__((aligned(16))) float fz_lds_gcw[];
  float* go = fz_lds_gcw;                               // [CO][kGoStride]
  float* tile = go + CO * kGoStride;                    // [LD][LH][LWS]
  float* red = tile + LD * LH * LWS;                    // [S][NR]
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x;
  const int g = blockIdx.y / a.Ci, ci = blockIdx.y % a.Ci, b = blockIdx.z;
  const int64_t V = (int64_t)a.D * a.H * a.W;
  const float* inc = a.in + (((int64_t)b * a.G + g) * a.Ci + ci) * V;
  const float* gob = a.gout + ((int64_t)b * a.G + g) * a.Co * V;
  const int ntiles = a.tiles_d * a.tiles_h * a.tiles_w;

  float tot[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) tot[s] = 0.f;

  const int t_lo = chunk * a.tiles_per_chunk;
  const int t_hi = min(ntiles, t_lo + a.tiles_per_chunk);
  for (int t = t_lo; t < t_hi; ++t) {
    const int twi = t % a.tiles_w, thi = (t / a.tiles_w) % a.tiles_h, tdi = t / (a.tiles_w * a.tiles_h);
    const int d0 = tdi * kTD, h0 = thi * kTH, w0 = twi * kTW;
    __syncthreads();   // the previous tile's sweep and reduction are done with go / tile / red
    for (int idx = tid; idx < CO * kTD * kTH * kTW; idx += 256) {
      const int lw = idx % kTW, lh = (idx / kTW) % kTH, ld = (idx / (kTW * kTH)) % kTD, o = idx / (kTW * kTH * kTD);
      const int zd = d0 + ld, zh = h0 + lh, zw = w0 + lw;
      const bool ok = o < a.Co && zd < a.D && zh < a.H && zw < a.W;
      go[o * kGoStride + (ld * kTH + lh) * kTW + lw] = ok ? gob[(int64_t)o * V + ((int64_t)zd * a.H + zh) * a.W + zw] : 0.f;
    }
    for (int idx = tid; idx < LD * LH * LW; idx += 256) {
      const int lw = idx % LW, lh = (idx / LW) % LH, ld = idx / (LW * LH);
      const int zd = d0 + ld - PD, zh = h0 + lh - PH, zw = w0 + lw - PW;
      const bool ok = zd >= 0 && zd < a.D && zh >= 0 && zh < a.H && zw >= 0 && zw < a.W;
      tile[(ld * LH + lh) * LWS + lw] = ok ? inc[((int64_t)zd * a.H + zh) * a.W + zw] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int item = tid + 256 * n;
      if (item < P * S) {
        const int pr = item / S, sl = item % S;
        const int kh = pr % KH, kd = (pr / KH) % KD, o = pr / (KH * KD);
        float acc[KW];
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) acc[kw] = 0.f;
        for (int q = sl; q < NQ; q += S) {
          const int tw = q % (kTW / 4), th = (q / (kTW / 4)) % kTH, td = q / (kTW / 4 * kTH);
          const float4 g4 = *reinterpret_cast<const float4*>(go + o * kGoStride + q * 4);
          const float* rp = tile + ((td + kd) * LH + (th + kh)) * LWS + tw * 4;
          float row[4 + KW - 1];
#pragma unroll
          for (int e = 0; e < 4 + KW - 1; ++e) row[e] = rp[e];
#pragma unroll
          for (int kw = 0; kw < KW; ++kw)
            acc[kw] = acc[kw] + g4.x * row[kw] + g4.y * row[kw + 1] + g4.z * row[kw + 2] + g4.w * row[kw + 3];
        }
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) red[sl * NR + pr * KW + kw] = acc[kw];
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int r = tid + 256 * s;
      if (r < NR) {
        float v = 0.f;
        for (int sl = 0; sl < S; ++sl) v += red[sl * NR + r];
        tot[s] += v;
      }
    }
  }
  constexpr int K3 = KD * KH * KW;
  float* pp = a.part + (((int64_t)b * a.G + g) * a.nchunks + chunk) * a.Co * a.Ci * K3;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int r = tid + 256 * s;
    if (r < NR) {
      const int o = r / K3, tau = r % K3;
      if (o < a.Co) pp[((int64_t)o * a.Ci + ci) * K3 + tau] = tot[s];
    }
  }
}

// ---- the same two operators for ANY odd kernel extent <= 7 per axis and ANY channel count per group [r6] ------------------
// The templates above cover the reference's defaults (cubic / square 3-5-7, <= 16 channels per group); the reference's own test
// suite also builds anisotropic kernels — kernel_size=(5, 3, 3) in tests/test_deconver.py — and `Deconv` allows any channel
// count.  Those used to fall back to framework convolutions on device; they now run here: kernel extents are run-time values
// (tap loops with compile-time bounds of 7 and a run-time guard, so every register array keeps compile-time indices — no
// scratch), the output channels of a group go eight at a time (grid.y = groups x blocks of 8).  Same tile, same halo staging,
// same deterministic two-stage reduction; a fallback in speed (half the register blocking, guards in the tap loops), not in
// kind.
constexpr int kAnyCO = 8, kAnyK = 7;

__global__ __launch_bounds__(256) void gcorr_any_kernel(GcArgs a, int KD, int KH, int KW, int nob) {
  const int PD = KD / 2, PH = KH / 2, PW = KW / 2;
  const int LD = kTD + 2 * PD, LH = kTH + 2 * PH, LW = kTW + 2 * PW;
  const int LWS = LW + 1;
  extern __shared__ __attribute__((aligned(16))) float fz_lds_gca[];
  float* tile = fz_lds_gca;
  const int tid = threadIdx.x;
  const int tw = tid & 15, th = (tid >> 4) & 3, td = tid >> 6;
  int bid = blockIdx.x;
  const int twi = bid % a.tiles_w; bid /= a.tiles_w;
  const int thi = bid % a.tiles_h; bid /= a.tiles_h;
  const int tiles_d = (a.D + kTD - 1) / kTD;
  const int tdi = bid % tiles_d;
  const int g = blockIdx.y / nob, o0 = (blockIdx.y % nob) * kAnyCO, b = blockIdx.z;
  const int d0 = tdi * kTD, h0 = thi * kTH, w0 = twi * kTW;
  const int64_t V = (int64_t)a.D * a.H * a.W;
  const int K3 = KD * KH * KW;
  const float* inb = a.in + ((int64_t)b * a.G + g) * a.Ci * V;
  const float* wg = a.w + (((int64_t)(a.w_batched ? b : 0) * a.G + g) * a.Co) * a.Ci * K3;

  float acc[kAnyCO][4];
#pragma unroll
  for (int o = 0; o < kAnyCO; ++o)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[o][e] = 0.f;

  for (int ci = 0; ci < a.Ci; ++ci) {
    __syncthreads();
    const float* inc = inb + (int64_t)ci * V;
    for (int idx = tid; idx < LD * LH * LW; idx += 256) {
      const int lw = idx % LW, lh = (idx / LW) % LH, ld = idx / (LW * LH);
      const int zd = d0 + ld - PD, zh = h0 + lh - PH, zw = w0 + lw - PW;
      const bool ok = zd >= 0 && zd < a.D && zh >= 0 && zh < a.H && zw >= 0 && zw < a.W;
      tile[(ld * LH + lh) * LWS + lw] = ok ? inc[((int64_t)zd * a.H + zh) * a.W + zw] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int kd = 0; kd < KD; ++kd)
#pragma unroll 1
      for (int kh = 0; kh < KH; ++kh) {
        float row[4 + kAnyK - 1];
        const float* rp = tile + ((td + kd) * LH + (th + kh)) * LWS + tw * 4;
#pragma unroll
        for (int e = 0; e < 4 + kAnyK - 1; ++e) row[e] = e < 4 + KW - 1 ? rp[e] : 0.f;
#pragma unroll
        for (int o = 0; o < kAnyCO; ++o) {
          if (o0 + o < a.Co) {
            const float* wp = wg + ((int64_t)(o0 + o) * a.Ci + ci) * K3 + (kd * KH + kh) * KW;   // uniform
#pragma unroll
            for (int kw = 0; kw < kAnyK; ++kw) {
              if (kw < KW) {
                const float wv = wp[kw];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[o][e] = acc[o][e] + row[e + kw] * wv;
              }
            }
          }
        }
      }
  }
  const int zd = d0 + td, zh = h0 + th, zw = w0 + tw * 4;
  if (zd >= a.D || zh >= a.H) return;
  const int64_t vo = ((int64_t)zd * a.H + zh) * a.W + zw;
#pragma unroll
  for (int o = 0; o < kAnyCO; ++o) {
    if (o0 + o >= a.Co) continue;
    const int64_t base = (((int64_t)b * a.G + g) * a.Co + o0 + o) * V + vo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (zw + e >= a.W) continue;
      float r = acc[o][e] + a.add_eps;
      if (a.epilogue == 1) r = a.mul_a[base + e] * a.mul_b[base + e] / r;
      a.out[base + e] = r;
    }
  }
}

// filter gradient, run-time extents: a work item is one (o, kd, kh) filter row of the block's eight output channels x one slice
// of the tile's voxel quads; the per-item KW sums sit in registers (seven, guarded), slices are added through LDS in slice order,
// tiles in LDS as well (`tot`: one float per (row, kw), owned by one thread — an LDS array because its length is a run-time value)
__global__ __launch_bounds__(256) void gcorr_wgrad_any_kernel(GcwArgs a, int KD, int KH, int KW, int nob, int S) {
  const int PD = KD / 2, PH = KH / 2, PW = KW / 2;
  const int LD = kTD + 2 * PD, LH = kTH + 2 * PH, LW = kTW + 2 * PW;
  const int LWS = LW + 1;
  const int P = kAnyCO * KD * KH;                      // filter rows of the block
  const int NR = P * KW;
  constexpr int NQ = kTD * kTH * kTW / 4;
  extern __shared__ __attribute__((aligned(16))) float fz_lds_gcwa[];
  float* go = fz_lds_gcwa;                              // [8][kGoStride]
  float* tile = go + kAnyCO * kGoStride;                // [LD][LH][LWS]
  float* red = tile + LD * LH * LWS;                    // [S][NR]
  float* tot = red + S * NR;                            // [NR]
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x;
  const int gi = blockIdx.y / nob, o0 = (blockIdx.y % nob) * kAnyCO;
  const int g = gi / a.Ci, ci = gi % a.Ci, b = blockIdx.z;
  const int64_t V = (int64_t)a.D * a.H * a.W;
  const float* inc = a.in + (((int64_t)b * a.G + g) * a.Ci + ci) * V;
  const float* gob = a.gout + (((int64_t)b * a.G + g) * a.Co + o0) * V;
  const int ntiles = a.tiles_d * a.tiles_h * a.tiles_w;
  for (int r = tid; r < NR; r += 256) tot[r] = 0.f;

  const int t_lo = chunk * a.tiles_per_chunk;
  const int t_hi = min(ntiles, t_lo + a.tiles_per_chunk);
  for (int t = t_lo; t < t_hi; ++t) {
    const int twi = t % a.tiles_w, thi = (t / a.tiles_w) % a.tiles_h, tdi = t / (a.tiles_w * a.tiles_h);
    const int d0 = tdi * kTD, h0 = thi * kTH, w0 = twi * kTW;
    __syncthreads();
    for (int idx = tid; idx < kAnyCO * kTD * kTH * kTW; idx += 256) {
      const int lw = idx % kTW, lh = (idx / kTW) % kTH, ld = (idx / (kTW * kTH)) % kTD, o = idx / (kTW * kTH * kTD);
      const int zd = d0 + ld, zh = h0 + lh, zw = w0 + lw;
      const bool ok = o0 + o < a.Co && zd < a.D && zh < a.H && zw < a.W;
      go[o * kGoStride + (ld * kTH + lh) * kTW + lw] = ok ? gob[(int64_t)o * V + ((int64_t)zd * a.H + zh) * a.W + zw] : 0.f;
    }
    for (int idx = tid; idx < LD * LH * LW; idx += 256) {
      const int lw = idx % LW, lh = (idx / LW) % LH, ld = idx / (LW * LH);
      const int zd = d0 + ld - PD, zh = h0 + lh - PH, zw = w0 + lw - PW;
      const bool ok = zd >= 0 && zd < a.D && zh >= 0 && zh < a.H && zw >= 0 && zw < a.W;
      tile[(ld * LH + lh) * LWS + lw] = ok ? inc[((int64_t)zd * a.H + zh) * a.W + zw] : 0.f;
    }
    __syncthreads();
    for (int item = tid; item < P * S; item += 256) {
      const int pr = item / S, sl = item % S;
      const int kh = pr % KH, kd = (pr / KH) % KD, o = pr / (KH * KD);
      float acc[kAnyK];
#pragma unroll
      for (int kw = 0; kw < kAnyK; ++kw) acc[kw] = 0.f;
      for (int q = sl; q < NQ; q += S) {
        const int tw = q % (kTW / 4), th = (q / (kTW / 4)) % kTH, td = q / (kTW / 4 * kTH);
        const float4 g4 = *reinterpret_cast<const float4*>(go + o * kGoStride + q * 4);
        const float* rp = tile + ((td + kd) * LH + (th + kh)) * LWS + tw * 4;
        float row[4 + kAnyK - 1];
#pragma unroll
        for (int e = 0; e < 4 + kAnyK - 1; ++e) row[e] = e < 4 + KW - 1 ? rp[e] : 0.f;
#pragma unroll
        for (int kw = 0; kw < kAnyK; ++kw)
          acc[kw] = acc[kw] + g4.x * row[kw] + g4.y * row[kw + 1] + g4.z * row[kw + 2] + g4.w * row[kw + 3];
      }
#pragma unroll
      for (int kw = 0; kw < kAnyK; ++kw)
        if (kw < KW) red[sl * NR + pr * KW + kw] = acc[kw];
    }
    __syncthreads();
    for (int r = tid; r < NR; r += 256) {
      float v = 0.f;
      for (int sl = 0; sl < S; ++sl) v += red[sl * NR + r];
      tot[r] += v;
    }
  }
  __syncthreads();
  const int K3 = KD * KH * KW;
  float* pp = a.part + (((int64_t)b * a.G + g) * a.nchunks + chunk) * a.Co * a.Ci * K3;
  for (int r = tid; r < NR; r += 256) {
    const int o = r / K3, tau = r % K3;
    if (o0 + o < a.Co) pp[((int64_t)(o0 + o) * a.Ci + ci) * K3 + tau] = tot[r];
  }
}

// gw[bw, g, e] = Σ_{b ∈ samples of bw} Σ_chunk part[b, g, chunk, e], ascending b then chunk
__global__ void gcorr_wgrad_finish_kernel(const float* part, float* gw, int B, int G, int nchunks, int E, int w_batched) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)(w_batched ? B : 1) * G * E;
  if (i >= n) return;
  const int e = (int)(i % E), g = (int)((i / E) % G), bw = (int)(i / ((int64_t)E * G));
  float v = 0.f;
  for (int b = w_batched ? bw : 0; b < (w_batched ? bw + 1 : B); ++b)
    for (int c = 0; c < nchunks; ++c) v += part[(((int64_t)b * G + g) * nchunks + c) * E + e];
  gw[i] = v;
}

}  // namespace fz

using namespace fz;

static bool gc_fast_shape(int Co, int kd, int kh, int kw) {   // the compile-time instantiations
  if (Co > 16) return false;
  if (kd == kh && kh == kw && (kw == 3 || kw == 5 || kw == 7)) return true;
  return kd == 1 && kh == kw && (kw == 3 || kw == 5 || kw == 7);
}
static bool gc_odd7(int k) { return k >= 1 && k <= kAnyK && (k & 1); }

extern "C" int fz_gcorr_supported(int Ci, int Co, int kd, int kh, int kw) {
  if (Ci < 1 || Co < 1) return 0;
  return gc_odd7(kd) && gc_odd7(kh) && gc_odd7(kw) ? 1 : 0;   // (the generic kernels take what the instantiations do not)
}

// out = corr(in, w) [+ eps]  or, with mul_a / mul_b, the fused multiplicative update  mul_a ∘ mul_b / (corr + eps).
// in (B, G·Ci, D, H, W); w (Bw, G, Co, Ci, kd, kh, kw) with Bw = B if w_batched else 1; out (B, G·Co, D, H, W). fp32.
extern "C" int fz_gcorr(const float* in, const float* w, float* out, const float* mul_a, const float* mul_b, int B, int G,
                        int Ci, int Co, int D, int H, int W, int kd, int kh, int kw, int w_batched, float add_eps,
                        fz_stream_t stream) {
  if (!in || !w || !out || ((mul_a == nullptr) != (mul_b == nullptr))) return fail(FZ_E_ARG, "fz_gcorr: null pointer");
  if (B < 0 || G < 1 || D < 1 || H < 1 || W < 1) return fail(FZ_E_SHAPE, "fz_gcorr: bad sizes");
  if (!fz_gcorr_supported(Ci, Co, kd, kh, kw))
    return fail(FZ_E_UNSUPPORTED, "fz_gcorr: needs odd kernel extents <= 7");
  if (B == 0) return FZ_OK;
  if (G > 65535 || B > 65535) return fail(FZ_E_UNSUPPORTED, "fz_gcorr: more than 65535 groups / samples");
  GcArgs a{in, w, out, mul_a, mul_b, B, G, Ci, Co, D, H, W, w_batched ? 1 : 0, mul_a ? 1 : 0, add_eps,
           (H + kTH - 1) / kTH, (W + kTW - 1) / kTW};
  const int64_t tiles = (int64_t)((D + kTD - 1) / kTD) * a.tiles_h * a.tiles_w;
  if (tiles > 0x7fffffff) return fail(FZ_E_UNSUPPORTED, "fz_gcorr: grid too large");
  dim3 grid((unsigned)tiles, (unsigned)G, (unsigned)B), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (!gc_fast_shape(Co, kd, kh, kw)) {   // anisotropic kernels, more than 16 channels per group: the run-time form
    const int nob = (Co + kAnyCO - 1) / kAnyCO;
    if ((int64_t)G * nob > 65535) return fail(FZ_E_UNSUPPORTED, "fz_gcorr: more than 65535 (group, channel block) pairs");
    const int lds = (kTD + 2 * (kd / 2)) * (kTH + 2 * (kh / 2)) * (kTW + 2 * (kw / 2) + 1) * (int)sizeof(float);
    hipLaunchKernelGGL(gcorr_any_kernel, dim3((unsigned)tiles, (unsigned)(G * nob), (unsigned)B), block, lds, st, a, kd, kh, kw, nob);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }
#define FZ_GC(CO_, KD_, KH_, KW_) hipLaunchKernelGGL((gcorr_kernel<CO_, KD_, KH_, KW_>), grid, block, 0, st, a)
#define FZ_GC_CO(KD_, KH_, KW_)                  \
  do {                                           \
    if (Co <= 1) FZ_GC(1, KD_, KH_, KW_);        \
    else if (Co <= 2) FZ_GC(2, KD_, KH_, KW_);   \
    else if (Co <= 4) FZ_GC(4, KD_, KH_, KW_);   \
    else if (Co <= 8) FZ_GC(8, KD_, KH_, KW_);   \
    else FZ_GC(16, KD_, KH_, KW_);               \
  } while (0)
  if (kd == 1) {
    if (kw == 3) FZ_GC_CO(1, 3, 3); else if (kw == 5) FZ_GC_CO(1, 5, 5); else FZ_GC_CO(1, 7, 7);
  } else {
    if (kw == 3) FZ_GC_CO(3, 3, 3); else if (kw == 5) FZ_GC_CO(5, 5, 5); else FZ_GC_CO(7, 7, 7);
  }
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

static int gcw_chunks(int B, int G, int Ci, int ntiles) {
  // enough workgroups to fill the chip (≈ 4 per CU), never more than one per tile
  const int64_t per = (int64_t)B * G * Ci;
  int64_t nc = (1024 + per - 1) / per;
  if (nc < 1) nc = 1;
  if (nc > ntiles) nc = ntiles;
  return (int)nc;
}

extern "C" int64_t fz_gcorr_wgrad_workspace_bytes(int B, int G, int Ci, int Co, int D, int H, int W, int kd, int kh, int kw) {
  if (B < 1 || G < 1 || Ci < 1 || Co < 1 || D < 1 || H < 1 || W < 1) return 0;
  const int ntiles = ((D + kTD - 1) / kTD) * ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
  return (int64_t)B * G * gcw_chunks(B, G, Ci, ntiles) * Co * Ci * kd * kh * kw * (int64_t)sizeof(float);
}

// Filter gradient of fz_gcorr: gw (Bw, G, Co, Ci, kd, kh, kw) from in (B, G·Ci, D, H, W) and gout (B, G·Co, D, H, W);
// ws: fz_gcorr_wgrad_workspace_bytes(...) bytes.  fp32.
extern "C" int fz_gcorr_wgrad(const float* in, const float* gout, float* gw, void* ws, int B, int G, int Ci, int Co, int D,
                              int H, int W, int kd, int kh, int kw, int w_batched, fz_stream_t stream) {
  if (!in || !gout || !gw || !ws) return fail(FZ_E_ARG, "fz_gcorr_wgrad: null pointer");
  if (B < 1 || G < 1 || D < 1 || H < 1 || W < 1) return fail(FZ_E_SHAPE, "fz_gcorr_wgrad: bad sizes");
  if (!fz_gcorr_supported(Ci, Co, kd, kh, kw))
    return fail(FZ_E_UNSUPPORTED, "fz_gcorr_wgrad: needs odd kernel extents <= 7");
  if ((int64_t)G * Ci > 65535 || B > 65535) return fail(FZ_E_UNSUPPORTED, "fz_gcorr_wgrad: more than 65535 (group, channel) pairs / samples");
  GcwArgs a{in, gout, (float*)ws, B, G, Ci, Co, D, H, W, (D + kTD - 1) / kTD, (H + kTH - 1) / kTH, (W + kTW - 1) / kTW, 0, 0};
  const int ntiles = a.tiles_d * a.tiles_h * a.tiles_w;
  a.nchunks = gcw_chunks(B, G, Ci, ntiles);
  a.tiles_per_chunk = (ntiles + a.nchunks - 1) / a.nchunks;
  dim3 grid((unsigned)a.nchunks, (unsigned)(G * Ci), (unsigned)B), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (!gc_fast_shape(Co, kd, kh, kw)) {
    const int nob = (Co + kAnyCO - 1) / kAnyCO;
    if ((int64_t)G * Ci * nob > 65535) return fail(FZ_E_UNSUPPORTED, "fz_gcorr_wgrad: more than 65535 (group, channel, channel block) triples");
    const int P = kAnyCO * kd * kh, S = P >= 256 ? 1 : 256 / P;
    const int lds = (kAnyCO * kGoStride + (kTD + 2 * (kd / 2)) * (kTH + 2 * (kh / 2)) * (kTW + 2 * (kw / 2) + 1) + (S + 1) * P * kw) *
                    (int)sizeof(float);
    if (lds > 160 * 1024) return fail(FZ_E_UNSUPPORTED, "fz_gcorr_wgrad: tile exceeds LDS");
    if (lds > 65536)
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(gcorr_wgrad_any_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(gcorr_wgrad_any_kernel, dim3((unsigned)a.nchunks, (unsigned)(G * Ci * nob), (unsigned)B), block, lds, st, a, kd, kh,
                       kw, nob, S);
    FZ_LAUNCH_CHECK();
    const int E = Co * Ci * kd * kh * kw;
    const int64_t n = (int64_t)(w_batched ? B : 1) * G * E;
    hipLaunchKernelGGL(gcorr_wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)ws, gw, B, G,
                       a.nchunks, E, w_batched ? 1 : 0);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }
#define FZ_GCW(CO_, KD_, KH_, KW_)                                                                                        \
  do {                                                                                                                    \
    constexpr int P_ = CO_ * KD_ * KH_, S_ = P_ >= 256 ? 1 : 256 / P_;                                                    \
    constexpr int lds_ = (CO_ * kGoStride + (kTD + 2 * (KD_ / 2)) * (kTH + 2 * (KH_ / 2)) * (kTW + 2 * (KW_ / 2) + 1) +    \
                          S_ * P_ * KW_) * (int)sizeof(float);                                                            \
    static_assert(lds_ <= 160 * 1024, "gcorr_wgrad: tile exceeds LDS");                                                   \
    auto kern = gcorr_wgrad_kernel<CO_, KD_, KH_, KW_>;                                                                   \
    if (lds_ > 65536)                                                                                                     \
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_)); \
    hipLaunchKernelGGL(kern, grid, block, lds_, st, a);                                                                   \
  } while (0)
#define FZ_GCW_CO(KD_, KH_, KW_)                  \
  do {                                            \
    if (Co <= 1) FZ_GCW(1, KD_, KH_, KW_);        \
    else if (Co <= 2) FZ_GCW(2, KD_, KH_, KW_);   \
    else if (Co <= 4) FZ_GCW(4, KD_, KH_, KW_);   \
    else if (Co <= 8) FZ_GCW(8, KD_, KH_, KW_);   \
    else FZ_GCW(16, KD_, KH_, KW_);               \
  } while (0)
  if (kd == 1) {
    if (kw == 3) FZ_GCW_CO(1, 3, 3); else if (kw == 5) FZ_GCW_CO(1, 5, 5); else FZ_GCW_CO(1, 7, 7);
  } else {
    if (kw == 3) FZ_GCW_CO(3, 3, 3); else if (kw == 5) FZ_GCW_CO(5, 5, 5); else FZ_GCW_CO(7, 7, 7);
  }
  FZ_LAUNCH_CHECK();
  const int E = Co * Ci * kd * kh * kw;
  const int64_t n = (int64_t)(w_batched ? B : 1) * G * E;
  hipLaunchKernelGGL(gcorr_wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)ws, gw, B, G,
                     a.nchunks, E, w_batched ? 1 : 0);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}
