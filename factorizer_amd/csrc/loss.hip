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
