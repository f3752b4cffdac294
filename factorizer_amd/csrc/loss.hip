// loss.hip — fused Dice + BCE-with-logits loss for the segmentation head (training step of
// BASELINE configs[3]).  Restates the bundle's DiceCELoss(sigmoid=True, squared_pred=True)
// (model_zoo/factorizer_brats23/configs/train.yaml:67-70; MONAI is not importable, so the exact
// reduction conventions are this build's: mean over (b,c) of the soft-Dice term + mean BCE):
//   p = sigmoid(z);  dice_bc = 1 - (2 Σ p t + s) / (Σ p² + Σ t² + s);  L = mean(dice_bc) + mean(bce)
// One pass produces, per (b, c) plane and chunk, the four sums {Σ p t, Σ p², Σ t², Σ bce};
// a second elementwise pass produces dL/dz from the per-plane scalars.
#include "fz_common.h"

namespace fz {

__global__ __launch_bounds__(256) void dice_bce_sums_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                            float* __restrict__ part, int64_t V, int nchunk) {
  __shared__ float red[4][4];
  const int plane = blockIdx.x, chunk = blockIdx.y;
  const int64_t per = ((V / 4 + nchunk - 1) / nchunk) * 4;
  const int64_t v0 = chunk * per, v1 = min(V, v0 + per);
  const float* zp = z + (int64_t)plane * V;
  const float* tp = t + (int64_t)plane * V;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t v = v0 + threadIdx.x * 4; v < v1; v += 1024) {
    const float4 zz = *reinterpret_cast<const float4*>(zp + v);
    const float4 tt = *reinterpret_cast<const float4*>(tp + v);
    const float zs[4] = {zz.x, zz.y, zz.z, zz.w}, ts[4] = {tt.x, tt.y, tt.z, tt.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float az = fabsf(zs[e]);
      const float ex = __expf(-az);                 // e^{-|z|}
      const float p = zs[e] >= 0.f ? 1.0f / (1.0f + ex) : ex / (1.0f + ex);
      s[0] += p * ts[e];
      s[1] += p * p;
      s[2] += ts[e] * ts[e];
      s[3] += fmaxf(zs[e], 0.f) - zs[e] * ts[e] + log1pf(ex);  // stable BCE with logits
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float r = wave_sum(s[e]);
    if (lane == 0) red[wave][e] = r;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int e = threadIdx.x;
    part[((int64_t)plane * nchunk + chunk) * 4 + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
  }
}

// dL/dz = g * [ cd_plane * (dDice/dp) * p(1-p) + cb * (p - t) ],
//   dDice/dp = -(2 t / den) + (2 inter + s) * 2 p / den²   (per plane: coef[plane] = {inter2s, den})
__global__ __launch_bounds__(256) void dice_bce_grad_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                            const float* __restrict__ coef, float* __restrict__ gz,
                                                            int64_t V, int64_t total, float cd, float cb,
                                                            const float* __restrict__ gscale) {
  const float g = gscale ? gscale[0] : 1.0f;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total;
       i += (int64_t)gridDim.x * blockDim.x * 4) {
    const int plane = (int)(i / V);
    const float num = coef[plane * 2], den = coef[plane * 2 + 1];
    const float4 zz = *reinterpret_cast<const float4*>(z + i);
    const float4 tt = *reinterpret_cast<const float4*>(t + i);
    const float zs[4] = {zz.x, zz.y, zz.z, zz.w}, ts[4] = {tt.x, tt.y, tt.z, tt.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ex = __expf(-fabsf(zs[e]));
      const float p = zs[e] >= 0.f ? 1.0f / (1.0f + ex) : ex / (1.0f + ex);
      const float ddp = -2.0f * ts[e] / den + num * 2.0f * p / (den * den);
      o[e] = g * (cd * ddp * p * (1.0f - p) + cb * (p - ts[e]));
    }
    *reinterpret_cast<float4*>(gz + i) = make_float4(o[0], o[1], o[2], o[3]);
  }
}


// ---- DiceCELoss(sigmoid=True, squared_pred=True) for C > 1 channels, as MONAI 1.4 evaluates it
// (the bundle pins monai==1.4.0, docs/requirements.txt:11; train.yaml:67-70): soft Dice on
// sigmoid(z) per (b, c) plane + nn.CrossEntropyLoss over the CHANNEL softmax with the float
// multi-label target as class "probabilities":  CE = mean_{b,v} Σ_c t_c · (logsumexp_c'(z) − z_c).
// One pass: per batch item and chunk the 3·C Dice sums {Σ p t, Σ p², Σ t²}_c and Σ CE.
template <int C>
__global__ __launch_bounds__(256) void dice_ce_sums_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                           float* __restrict__ part, int64_t V, int nchunk) {
  __shared__ float red[4][3 * C + 1];
  const int b = blockIdx.x, chunk = blockIdx.y;
  const int64_t per = ((V / 4 + nchunk - 1) / nchunk) * 4;
  const int64_t v0 = chunk * per, v1 = min(V, v0 + per);
  const float* zp = z + (int64_t)b * C * V;
  const float* tp = t + (int64_t)b * C * V;
  float s[3 * C + 1];
#pragma unroll
  for (int i = 0; i < 3 * C + 1; ++i) s[i] = 0.f;
  for (int64_t v = v0 + threadIdx.x * 4; v < v1; v += 1024) {
    float zs[C][4], ts[C][4];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float4 zz = *reinterpret_cast<const float4*>(zp + c * V + v);
      const float4 tt = *reinterpret_cast<const float4*>(tp + c * V + v);
      zs[c][0] = zz.x, zs[c][1] = zz.y, zs[c][2] = zz.z, zs[c][3] = zz.w;
      ts[c][0] = tt.x, ts[c][1] = tt.y, ts[c][2] = tt.z, ts[c][3] = tt.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float mx = zs[0][e];
#pragma unroll
      for (int c = 1; c < C; ++c) mx = fmaxf(mx, zs[c][e]);
      float se = 0.f, st = 0.f, stz = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float zc = zs[c][e], tc = ts[c][e];
        se += __expf(zc - mx);
        st += tc;
        stz += tc * zc;
        const float ex = __expf(-fabsf(zc));
        const float p = zc >= 0.f ? 1.0f / (1.0f + ex) : ex / (1.0f + ex);
        s[3 * c + 0] += p * tc;
        s[3 * c + 1] += p * p;
        s[3 * c + 2] += tc * tc;
      }
      s[3 * C] += st * (mx + __logf(se)) - stz;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 3 * C + 1; ++i) {
    const float r = wave_sum(s[i]);
    if (lane == 0) red[wave][i] = r;
  }
  __syncthreads();
  if (threadIdx.x < 3 * C + 1) {
    const int i = threadIdx.x;
    part[((int64_t)b * nchunk + chunk) * (3 * C + 1) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

// dL/dz_c = g·[ cd·dDice_c/dp·p(1−p) + cb·(softmax_c·Σ_c' t_c' − t_c) ]; coef (B·C, 2) as in dice_bce_grad.
template <int C>
__global__ __launch_bounds__(256) void dice_ce_grad_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                           const float* __restrict__ coef, float* __restrict__ gz,
                                                           int64_t V, int B, float cd, float cb,
                                                           const float* __restrict__ gscale) {
  const float g = gscale ? gscale[0] : 1.0f;
  const int64_t total = (int64_t)B * V;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total;
       i += (int64_t)gridDim.x * blockDim.x * 4) {
    const int b = (int)(i / V);
    const int64_t v = i - (int64_t)b * V;
    const int64_t base = (int64_t)b * C * V + v;
    float zs[C][4], ts[C][4];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float4 zz = *reinterpret_cast<const float4*>(z + base + c * V);
      const float4 tt = *reinterpret_cast<const float4*>(t + base + c * V);
      zs[c][0] = zz.x, zs[c][1] = zz.y, zs[c][2] = zz.z, zs[c][3] = zz.w;
      ts[c][0] = tt.x, ts[c][1] = tt.y, ts[c][2] = tt.z, ts[c][3] = tt.w;
    }
    float o[C][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float mx = zs[0][e];
#pragma unroll
      for (int c = 1; c < C; ++c) mx = fmaxf(mx, zs[c][e]);
      float ez[C], se = 0.f, st = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        ez[c] = __expf(zs[c][e] - mx);
        se += ez[c];
        st += ts[c][e];
      }
      const float inv = st / se;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float zc = zs[c][e], tc = ts[c][e];
        const float num = coef[(b * C + c) * 2], den = coef[(b * C + c) * 2 + 1];
        const float ex = __expf(-fabsf(zc));
        const float p = zc >= 0.f ? 1.0f / (1.0f + ex) : ex / (1.0f + ex);
        const float ddp = -2.0f * tc / den + num * 2.0f * p / (den * den);
        o[c][e] = g * (cd * ddp * p * (1.0f - p) + cb * (ez[c] * inv - tc));
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c)
      *reinterpret_cast<float4*>(gz + base + c * V) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
  }
}

}  // namespace fz

using namespace fz;

extern "C" int fz_dice_bce_chunks(int64_t V) {
  int64_t n = V / 32768;
  if (n < 1) n = 1;
  if (n > 64) n = 64;
  return (int)n;
}

// part: (planes, nchunk, 4) partial sums {Σ p t, Σ p², Σ t², Σ bce}
extern "C" int fz_dice_bce_sums(const float* z, const float* t, float* part, int planes, int64_t V,
                                fz_stream_t stream) {
  if (!z || !t || !part) return fail(FZ_E_ARG, "fz_dice_bce_sums: null pointer");
  if (planes < 1 || V < 4 || (V % 4)) return fail(FZ_E_SHAPE, "fz_dice_bce_sums: bad sizes");
  const int nchunk = fz_dice_bce_chunks(V);
  hipLaunchKernelGGL(dice_bce_sums_kernel, dim3(planes, nchunk), dim3(256), 0, (hipStream_t)stream, z, t, part, V,
                     nchunk);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

// coef: (planes, 2) = {2·inter + smooth, den + smooth};  cd = 1/planes, cb = 1/(planes·V);
// gscale: optional device scalar multiplying the gradient (upstream dL)
extern "C" int fz_dice_bce_grad(const float* z, const float* t, const float* coef, float* gz, int planes, int64_t V,
                                float cd, float cb, const float* gscale, fz_stream_t stream) {
  if (!z || !t || !coef || !gz) return fail(FZ_E_ARG, "fz_dice_bce_grad: null pointer");
  if (planes < 1 || V < 4 || (V % 4)) return fail(FZ_E_SHAPE, "fz_dice_bce_grad: bad sizes");
  const int64_t total = (int64_t)planes * V;
  int64_t blocks = (total / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dice_bce_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z, t, coef, gz, V,
                     total, cd, cb, gscale);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

// ---- multi-channel DiceCE (C in 2..8) ----
// part: (B, nchunk, 3C+1) = per channel {Σ p t, Σ p², Σ t²}, then Σ CE
#define FZ_CE_DISPATCH(C_, CALL) \
  switch (C_) {                  \
    case 2: { constexpr int CC = 2; CALL; } break; \
    case 3: { constexpr int CC = 3; CALL; } break; \
    case 4: { constexpr int CC = 4; CALL; } break; \
    case 5: { constexpr int CC = 5; CALL; } break; \
    case 6: { constexpr int CC = 6; CALL; } break; \
    case 7: { constexpr int CC = 7; CALL; } break; \
    case 8: { constexpr int CC = 8; CALL; } break; \
    default: return fail(FZ_E_UNSUPPORTED, "fz_dice_ce: 2 <= C <= 8"); \
  }

extern "C" int fz_dice_ce_sums(const float* z, const float* t, float* part, int B, int C, int64_t V,
                               fz_stream_t stream) {
  if (!z || !t || !part) return fail(FZ_E_ARG, "fz_dice_ce_sums: null pointer");
  if (B < 1 || V < 4 || (V % 4)) return fail(FZ_E_SHAPE, "fz_dice_ce_sums: bad sizes");
  const int nchunk = fz_dice_bce_chunks(V);
  FZ_CE_DISPATCH(C, hipLaunchKernelGGL(dice_ce_sums_kernel<CC>, dim3(B, nchunk), dim3(256), 0, (hipStream_t)stream,
                                       z, t, part, V, nchunk));
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

// One workgroup turns the partial sums into the loss and the Dice coefficients of the gradient pass: replaces a dozen
// launch-bound framework kernels (sum over chunks, 2·inter + smooth, den + smooth, 1 - num/den, means) between the two
// passes.  Chunks are added in a fixed order: lane-strided partial sums, then the wave butterfly of fz_common.h.
__global__ __launch_bounds__(256) void dice_ce_finish_kernel(const float* __restrict__ part, int nchunk, int B, int C, float inv_bv,
                                                             float smooth, float* __restrict__ loss, float* __restrict__ coef) {
  __shared__ float cols[8 * 25 + 8];   // (B <= 8) x (3C + 1 <= 25) column totals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ncol = 3 * C + 1;
  for (int col = wave; col < B * ncol; col += 4) {
    const int b = col / ncol, k = col % ncol;
    float t = 0.f;
    for (int ch = lane; ch < nchunk; ch += 64) t += part[((int64_t)b * nchunk + ch) * ncol + k];
    t = wave_sum(t);
    if (lane == 0) cols[col] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float dice = 0.f, ce = 0.f;
    for (int b = 0; b < B; ++b) {
      for (int c = 0; c < C; ++c) {
        const float num = 2.0f * cols[b * ncol + 3 * c] + smooth;
        const float den = cols[b * ncol + 3 * c + 1] + cols[b * ncol + 3 * c + 2] + smooth;
        coef[(b * C + c) * 2] = num;
        coef[(b * C + c) * 2 + 1] = den;
        dice += 1.0f - num / den;
      }
      ce += cols[b * ncol + 3 * C];
    }
    loss[0] = dice / (float)(B * C) + ce * inv_bv;
  }
}

extern "C" int fz_dice_ce_finish(const float* part, int B, int C, int64_t V, float smooth, float* loss, float* coef,
                                 fz_stream_t stream) {
  if (!part || !loss || !coef) return fail(FZ_E_ARG, "fz_dice_ce_finish: null pointer");
  if (B < 1 || B > 8 || C < 2 || C > 8) return fail(FZ_E_UNSUPPORTED, "fz_dice_ce_finish: 1 <= B <= 8, 2 <= C <= 8");
  hipLaunchKernelGGL(dice_ce_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, fz_dice_bce_chunks(V), B, C,
                     1.0f / ((float)B * (float)V), smooth, loss, coef);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

// coef: (B·C, 2) = {2·inter + smooth, den + smooth};  cd = 1/(B·C), cb = 1/(B·V)
extern "C" int fz_dice_ce_grad(const float* z, const float* t, const float* coef, float* gz, int B, int C, int64_t V,
                               float cd, float cb, const float* gscale, fz_stream_t stream) {
  if (!z || !t || !coef || !gz) return fail(FZ_E_ARG, "fz_dice_ce_grad: null pointer");
  if (B < 1 || V < 4 || (V % 4)) return fail(FZ_E_SHAPE, "fz_dice_ce_grad: bad sizes");
  int64_t blocks = ((int64_t)B * V / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  FZ_CE_DISPATCH(C, hipLaunchKernelGGL(dice_ce_grad_kernel<CC>, dim3((unsigned)blocks), dim3(256), 0,
                                       (hipStream_t)stream, z, t, coef, gz, V, B, cd, cb, gscale));
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}
