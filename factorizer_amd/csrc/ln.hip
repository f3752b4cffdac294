// ln.hip — channels-first LayerNorm (normalise over C for every voxel), standalone kernels.
// Replaces the reference's permute → nn.LayerNorm → permute (layers/norm.py:29-34).  Inside
// FactorizerBlock the forward LayerNorm is fused into the following GEMM (gemm.hip); these
// kernels serve ft.LayerNorm used on its own and the LayerNorm backward.
//
// One thread owns 4 consecutive voxels (16-byte accesses, fully coalesced per channel plane)
// and walks the channel dimension; HBM-bound: fwd reads C·V and writes C·V floats.
#include "fz_common.h"

namespace fz {

// y = (x - mean) * rstd * g + b ; stats (B,2,V) optional output
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                     const float* __restrict__ bta, float* __restrict__ y,
                                                     float* __restrict__ stats, int B, int C, int64_t V,
                                                     float eps) {
  const int64_t nvec = V / 4;
  const int64_t total = nvec * B;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / nvec);
    const int64_t v = (i % nvec) * 4;
    const float* xp = x + (int64_t)b * C * V + v;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < C; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(xp + (int64_t)c * V);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const float inv = 1.0f / (float)C;
    const float4 mu = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < C; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(xp + (int64_t)c * V);
      const float dx = t.x - mu.x, dy = t.y - mu.y, dz = t.z - mu.z, dw = t.w - mu.w;
      q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
    }
    const float4 rs = make_float4(1.0f / sqrtf(q.x * inv + eps), 1.0f / sqrtf(q.y * inv + eps),
                                  1.0f / sqrtf(q.z * inv + eps), 1.0f / sqrtf(q.w * inv + eps));
    float* yp = y + (int64_t)b * C * V + v;
    for (int c = 0; c < C; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(xp + (int64_t)c * V);
      const float gc = g[c], bc = bta[c];
      float4 o;
      o.x = (t.x - mu.x) * rs.x * gc + bc; o.y = (t.y - mu.y) * rs.y * gc + bc;
      o.z = (t.z - mu.z) * rs.z * gc + bc; o.w = (t.w - mu.w) * rs.w * gc + bc;
      *reinterpret_cast<float4*>(yp + (int64_t)c * V) = o;
    }
    if (stats != nullptr) {
      float* sp = stats + (int64_t)b * 2 * V + v;
      *reinterpret_cast<float4*>(sp) = mu;
      *reinterpret_cast<float4*>(sp + V) = rs;
    }
  }
}

// gx = rstd * (gl*g - mean_c(gl*g) - n * mean_c(gl*g*n)) [+ gadd],  n = (x - mean) * rstd
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ gl, const float* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ g,
                                                     const float* __restrict__ gadd, float* __restrict__ gx,
                                                     int B, int C, int64_t V) {
  const int64_t nvec = V / 4;
  const int64_t total = nvec * B;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / nvec);
    const int64_t v = (i % nvec) * 4;
    const int64_t base = (int64_t)b * C * V + v;
    const float* sp = stats + (int64_t)b * 2 * V + v;
    const float4 mu = *reinterpret_cast<const float4*>(sp);
    const float4 rs = *reinterpret_cast<const float4*>(sp + V);
    float4 m1 = make_float4(0.f, 0.f, 0.f, 0.f), m2 = m1;
    for (int c = 0; c < C; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(x + base + (int64_t)c * V);
      const float4 d = *reinterpret_cast<const float4*>(gl + base + (int64_t)c * V);
      const float gc = g[c];
      const float ax = d.x * gc, ay = d.y * gc, az = d.z * gc, aw = d.w * gc;
      m1.x += ax; m1.y += ay; m1.z += az; m1.w += aw;
      m2.x += ax * (t.x - mu.x) * rs.x; m2.y += ay * (t.y - mu.y) * rs.y;
      m2.z += az * (t.z - mu.z) * rs.z; m2.w += aw * (t.w - mu.w) * rs.w;
    }
    const float inv = 1.0f / (float)C;
    m1.x *= inv; m1.y *= inv; m1.z *= inv; m1.w *= inv;
    m2.x *= inv; m2.y *= inv; m2.z *= inv; m2.w *= inv;
    for (int c = 0; c < C; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(x + base + (int64_t)c * V);
      const float4 d = *reinterpret_cast<const float4*>(gl + base + (int64_t)c * V);
      const float gc = g[c];
      float4 o;
      o.x = rs.x * (d.x * gc - m1.x - (t.x - mu.x) * rs.x * m2.x);
      o.y = rs.y * (d.y * gc - m1.y - (t.y - mu.y) * rs.y * m2.y);
      o.z = rs.z * (d.z * gc - m1.z - (t.z - mu.z) * rs.z * m2.z);
      o.w = rs.w * (d.w * gc - m1.w - (t.w - mu.w) * rs.w * m2.w);
      if (gadd != nullptr) {
        const float4 r = *reinterpret_cast<const float4*>(gadd + base + (int64_t)c * V);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      *reinterpret_cast<float4*>(gx + base + (int64_t)c * V) = o;
    }
  }
}

}  // namespace fz

using namespace fz;

static unsigned ln_grid(int64_t total) {
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

extern "C" int fz_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, int B,
                         int C, int64_t V, float eps, fz_stream_t stream) {
  if (!x || !gamma || !beta || !y) return fail(FZ_E_ARG, "fz_ln_fwd: null pointer");
  if (B < 0 || C < 1 || V < 1) return fail(FZ_E_SHAPE, "fz_ln_fwd: bad sizes");
  if (V % 4) return fail(FZ_E_UNSUPPORTED, "fz_ln_fwd: voxel count must be a multiple of 4");
  if (B == 0) return FZ_OK;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(ln_grid(V / 4 * B)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                     stats, B, C, V, eps);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_ln_bwd(const float* gl, const float* x, const float* stats, const float* gamma,
                         const float* gadd, float* gx, int B, int C, int64_t V, fz_stream_t stream) {
  if (!gl || !x || !stats || !gamma || !gx) return fail(FZ_E_ARG, "fz_ln_bwd: null pointer");
  if (B < 0 || C < 1 || V < 1) return fail(FZ_E_SHAPE, "fz_ln_bwd: bad sizes");
  if (V % 4) return fail(FZ_E_UNSUPPORTED, "fz_ln_bwd: voxel count must be a multiple of 4");
  if (B == 0) return FZ_OK;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(ln_grid(V / 4 * B)), dim3(256), 0, (hipStream_t)stream, gl, x, stats,
                     gamma, gadd, gx, B, C, V);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}
