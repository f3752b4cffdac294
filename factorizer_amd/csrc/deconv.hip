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

}  // namespace fz

using namespace fz;

extern "C" int fz_gcorr_supported(int Ci, int Co, int kd, int kh, int kw) {
  if (Ci < 1 || Co < 1 || Co > 16) return 0;
  if (kd == kh && kh == kw && (kw == 3 || kw == 5 || kw == 7)) return 1;
  if (kd == 1 && kh == kw && (kw == 3 || kw == 5 || kw == 7)) return 1;
  return 0;
}

// out = corr(in, w) [+ eps]  or, with mul_a / mul_b, the fused multiplicative update  mul_a ∘ mul_b / (corr + eps).
// in (B, G·Ci, D, H, W); w (Bw, G, Co, Ci, kd, kh, kw) with Bw = B if w_batched else 1; out (B, G·Co, D, H, W). fp32.
extern "C" int fz_gcorr(const float* in, const float* w, float* out, const float* mul_a, const float* mul_b, int B, int G,
                        int Ci, int Co, int D, int H, int W, int kd, int kh, int kw, int w_batched, float add_eps,
                        fz_stream_t stream) {
  if (!in || !w || !out || ((mul_a == nullptr) != (mul_b == nullptr))) return fail(FZ_E_ARG, "fz_gcorr: null pointer");
  if (B < 0 || G < 1 || D < 1 || H < 1 || W < 1) return fail(FZ_E_SHAPE, "fz_gcorr: bad sizes");
  if (!fz_gcorr_supported(Ci, Co, kd, kh, kw))
    return fail(FZ_E_UNSUPPORTED, "fz_gcorr: needs <= 16 output channels per group and a 3/5/7 cubic (or depth-1 square) kernel");
  if (B == 0) return FZ_OK;
  if (G > 65535 || B > 65535) return fail(FZ_E_UNSUPPORTED, "fz_gcorr: more than 65535 groups / samples");
  GcArgs a{in, w, out, mul_a, mul_b, B, G, Ci, Co, D, H, W, w_batched ? 1 : 0, mul_a ? 1 : 0, add_eps,
           (H + kTH - 1) / kTH, (W + kTW - 1) / kTW};
  const int64_t tiles = (int64_t)((D + kTD - 1) / kTD) * a.tiles_h * a.tiles_w;
  if (tiles > 0x7fffffff) return fail(FZ_E_UNSUPPORTED, "fz_gcorr: grid too large");
  dim3 grid((unsigned)tiles, (unsigned)G, (unsigned)B), block(256);
  hipStream_t st = (hipStream_t)stream;
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
