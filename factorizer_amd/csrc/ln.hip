// ln.hip — channels-first LayerNorm (normalise over C for every voxel), standalone kernels.
// Replaces the reference's permute → nn.LayerNorm → permute (layers/norm.py:29-34).  Inside
// FactorizerBlock the forward LayerNorm is fused into the following GEMM (gemm.hip); these
// kernels serve ft.LayerNorm used on its own and the LayerNorm backward.
//
// One thread owns 4 consecutive voxels (16-byte accesses, fully coalesced per channel plane)
// and walks the channel dimension; HBM-bound: fwd reads C·V and writes C·V floats.
#include <cstdlib>

#include "fz_common.h"
#include "finish.h"

namespace fz {

// y = (x - mean) * rstd * g + b ; stats (B,2,V) optional output
template <typename AT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const AT* __restrict__ x, const float* __restrict__ g,
                                                     const float* __restrict__ bta, AT* __restrict__ y,
                                                     float* __restrict__ stats, int B, int C, int64_t V,
                                                     float eps) {
  const int64_t nvec = V / 4;
  const int64_t total = nvec * B;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / nvec);
    const int64_t v = (i % nvec) * 4;
    const AT* xp = x + (int64_t)b * C * V + v;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < C; ++c) {
      const float4 t = ld4(xp + (int64_t)c * V);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const float inv = 1.0f / (float)C;
    const float4 mu = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < C; ++c) {
      const float4 t = ld4(xp + (int64_t)c * V);
      const float dx = t.x - mu.x, dy = t.y - mu.y, dz = t.z - mu.z, dw = t.w - mu.w;
      q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
    }
    const float4 rs = make_float4(1.0f / sqrtf(q.x * inv + eps), 1.0f / sqrtf(q.y * inv + eps),
                                  1.0f / sqrtf(q.z * inv + eps), 1.0f / sqrtf(q.w * inv + eps));
    AT* yp = y + (int64_t)b * C * V + v;
    for (int c = 0; c < C; ++c) {
      const float4 t = ld4(xp + (int64_t)c * V);
      const float gc = g[c], bc = bta[c];
      float4 o;
      o.x = (t.x - mu.x) * rs.x * gc + bc; o.y = (t.y - mu.y) * rs.y * gc + bc;
      o.z = (t.z - mu.z) * rs.z * gc + bc; o.w = (t.w - mu.w) * rs.w * gc + bc;
      st4(yp + (int64_t)c * V, o);
    }
    if (stats != nullptr) {
      float* sp = stats + (int64_t)b * 2 * V + v;
      *reinterpret_cast<float4*>(sp) = mu;
      *reinterpret_cast<float4*>(sp + V) = rs;
    }
  }
}

// gx = rstd * (gl*g - mean_c(gl*g) - n * mean_c(gl*g*n)) [+ gadd],  n = (x - mean) * rstd
// CMAX > 0: the kernel also accumulates the affine gradients  gγ_c = Σ gl_c·n_c ,  gβ_c = Σ gl_c
// in registers (C <= CMAX) and writes one partial row per workgroup: part[blk][2][C].
template <int CMAX, typename AT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const AT* __restrict__ gl, const AT* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ g,
                                                     const AT* __restrict__ gadd, AT* __restrict__ gx,
                                                     float* __restrict__ part, int B, int C, int64_t V) {
  constexpr int CA = CMAX > 0 ? CMAX : 1;
  float ag[CA], ab[CA];
#pragma unroll
  for (int c = 0; c < CA; ++c) ag[c] = ab[c] = 0.f;
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
    auto pass1 = [&](int c, int ci) {
      const float4 t = ld4(x + base + (int64_t)c * V);
      const float4 d = ld4(gl + base + (int64_t)c * V);
      const float gc = g[c];
      const float nx = (t.x - mu.x) * rs.x, ny = (t.y - mu.y) * rs.y, nz = (t.z - mu.z) * rs.z,
                  nw = (t.w - mu.w) * rs.w;
      const float ax = d.x * gc, ay = d.y * gc, az = d.z * gc, aw = d.w * gc;
      m1.x += ax; m1.y += ay; m1.z += az; m1.w += aw;
      m2.x += ax * nx; m2.y += ay * ny; m2.z += az * nz; m2.w += aw * nw;
      if (CMAX > 0) {
        ag[ci] += (d.x * nx + d.y * ny) + (d.z * nz + d.w * nw);
        ab[ci] += (d.x + d.y) + (d.z + d.w);
      }
    };
    if (CMAX > 0) {
#pragma unroll
      for (int c = 0; c < CA; ++c)
        if (c < C) pass1(c, c);
    } else {
      for (int c = 0; c < C; ++c) pass1(c, 0);
    }
    const float inv = 1.0f / (float)C;
    m1.x *= inv; m1.y *= inv; m1.z *= inv; m1.w *= inv;
    m2.x *= inv; m2.y *= inv; m2.z *= inv; m2.w *= inv;
    for (int c = 0; c < C; ++c) {
      const float4 t = ld4(x + base + (int64_t)c * V);
      const float4 d = ld4(gl + base + (int64_t)c * V);
      const float gc = g[c];
      float4 o;
      o.x = rs.x * (d.x * gc - m1.x - (t.x - mu.x) * rs.x * m2.x);
      o.y = rs.y * (d.y * gc - m1.y - (t.y - mu.y) * rs.y * m2.y);
      o.z = rs.z * (d.z * gc - m1.z - (t.z - mu.z) * rs.z * m2.z);
      o.w = rs.w * (d.w * gc - m1.w - (t.w - mu.w) * rs.w * m2.w);
      if (gadd != nullptr) {
        const float4 r = ld4(gadd + base + (int64_t)c * V);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      st4(gx + base + (int64_t)c * V, o);
    }
  }
  if (CMAX > 0) {
    __shared__ float red[4][2 * CA];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < CA; ++c) {
      const float sg = wave_sum(ag[c]);
      const float sb = wave_sum(ab[c]);
      if (lane == 0) {
        red[wave][c] = sg;
        red[wave][CA + c] = sb;
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += blockDim.x) {
      const int which = e / C, c = e % C;
      const int idx = which * CA + c;
      part[(int64_t)blockIdx.x * 2 * C + e] = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
    }
  }
}

// Backward for wide tensors (C > 64: the 8^3..32^3 stages).  Lanes = consecutive 4-voxel quads
// (16-byte accesses stay coalesced: 1 KiB per channel row per wave), the NW waves of a workgroup
// split the channels; the two per-voxel means are combined through LDS.
template <int NW, typename AT>
__global__ __launch_bounds__(64 * NW) void ln_bwd_split_kernel(const AT* __restrict__ gl, const AT* __restrict__ x,
                                                               const float* __restrict__ stats,
                                                               const float* __restrict__ g,
                                                               const AT* __restrict__ gadd, AT* __restrict__ gx,
                                                               float* __restrict__ part, int B, int C, int64_t V) {
  __shared__ float red[NW][8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nvec = V / 4;
  const int64_t total = nvec * B;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const bool ok = i < total;
  const int64_t ii = ok ? i : 0;
  const int b = (int)(ii / nvec);
  const int64_t v = (ii % nvec) * 4;
  const int64_t base = (int64_t)b * C * V + v;
  const float* sp = stats + (int64_t)b * 2 * V + v;
  const float4 mu = *reinterpret_cast<const float4*>(sp);
  const float4 rs = *reinterpret_cast<const float4*>(sp + V);
  const int cs = (C + NW - 1) / NW;
  const int c0 = wave * cs, c1 = min(C, c0 + cs);
  float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c = c0; c < c1; ++c) {
    const float4 t = ld4(x + base + (int64_t)c * V);
    const float4 d = ld4(gl + base + (int64_t)c * V);
    const float gc = g[c];
    const float nx = (t.x - mu.x) * rs.x, ny = (t.y - mu.y) * rs.y, nz = (t.z - mu.z) * rs.z, nw = (t.w - mu.w) * rs.w;
    const float ax = d.x * gc, ay = d.y * gc, az = d.z * gc, aw = d.w * gc;
    m[0] += ax; m[1] += ay; m[2] += az; m[3] += aw;
    m[4] += ax * nx; m[5] += ay * ny; m[6] += az * nz; m[7] += aw * nw;
    if (part != nullptr) {
      // affine gradients of channel c over this workgroup's 64 quads: part[blk][c | C + c]
      const float sg = wave_sum(ok ? (d.x * nx + d.y * ny) + (d.z * nz + d.w * nw) : 0.f);
      const float sb = wave_sum(ok ? (d.x + d.y) + (d.z + d.w) : 0.f);
      if (lane == 0) {
        part[(int64_t)blockIdx.x * 2 * C + c] = sg;
        part[(int64_t)blockIdx.x * 2 * C + C + c] = sb;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[wave][e][lane] = m[e];
  __syncthreads();
  const float inv = 1.0f / (float)C;
  float mm[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += red[w][e][lane];
    mm[e] = t * inv;
  }
  if (!ok) return;
  for (int c = c0; c < c1; ++c) {
    const float4 t = ld4(x + base + (int64_t)c * V);
    const float4 d = ld4(gl + base + (int64_t)c * V);
    const float gc = g[c];
    float4 o;
    o.x = rs.x * (d.x * gc - mm[0] - (t.x - mu.x) * rs.x * mm[4]);
    o.y = rs.y * (d.y * gc - mm[1] - (t.y - mu.y) * rs.y * mm[5]);
    o.z = rs.z * (d.z * gc - mm[2] - (t.z - mu.z) * rs.z * mm[6]);
    o.w = rs.w * (d.w * gc - mm[3] - (t.w - mu.w) * rs.w * mm[7]);
    if (gadd != nullptr) {
      const float4 r = ld4(gadd + base + (int64_t)c * V);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    st4(gx + base + (int64_t)c * V, o);
  }
}

// Channel-sliced backward with compile-time slices (C = 8·CS: the 64..512-channel stages).
// 8 waves per workgroup, wave w owns channels [w·CS, (w+1)·CS).  SUB = 1: lanes = 64 consecutive voxel quads.
// SUB = 4 (the deep stages: a few thousand voxels): lanes = 16 voxel quads x 4 channel sub-slices of CS / 4 channels, so
// the grid is 4x larger (stage 4 of the README model at B = 2: 16 workgroups instead of 4, 16 channels per lane instead
// of 64 — one batch of loads instead of sixteen dependent ones) and the slice fits the registers up to C = 512.
// All loads of a pass are issued before the first use (no per-channel dependent round trips), the
// optional added gradient and the affine partials are template flags (no branches in the loops),
// and for <= 16 channels per lane the slice stays in registers between the two passes (each tensor is read once).
template <int CS, bool GADD, bool PART, typename AT, int SUB = 1>
__global__ __launch_bounds__(512) void ln_bwd_slice_kernel(const AT* __restrict__ gl, const AT* __restrict__ x,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ g,
                                                           const AT* __restrict__ gadd, AT* __restrict__ gx,
                                                           float* __restrict__ part, int B, int64_t V) {
  constexpr int NW = 8, C = NW * CS;
  constexpr int QW = 64 / SUB;        // voxel quads per workgroup
  constexpr int CL = CS / SUB;        // channels per lane
  constexpr bool KEEP = CL <= 16;
  constexpr int U = KEEP ? CL : 8;    // channels per batch of loads
  __shared__ float red[NW][8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ql = lane & (QW - 1), sub = lane / QW;
  const int64_t nvec = V / 4;
  const int64_t total = nvec * B;
  const int64_t i = (int64_t)blockIdx.x * QW + ql;
  const bool ok = i < total;
  const int64_t ii = ok ? i : 0;
  const int b = (int)(ii / nvec);
  const int64_t v = (ii % nvec) * 4;
  const int c0 = wave * CS + sub * CL;
  const int64_t base = ((int64_t)b * C + c0) * V + v;
  const float* sp = stats + (int64_t)b * 2 * V + v;
  const float4 mu = *reinterpret_cast<const float4*>(sp);
  const float4 rs = *reinterpret_cast<const float4*>(sp + V);
  float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float4 t[U], d[U];
#pragma unroll 1
  for (int cb = 0; cb < CL; cb += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      t[u] = ld4(x + base + (int64_t)(cb + u) * V);
      d[u] = ld4(gl + base + (int64_t)(cb + u) * V);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float gc = g[c0 + cb + u];
      // normalised input replaces x; reused by the second pass when the slice stays in registers
      t[u].x = (t[u].x - mu.x) * rs.x; t[u].y = (t[u].y - mu.y) * rs.y;
      t[u].z = (t[u].z - mu.z) * rs.z; t[u].w = (t[u].w - mu.w) * rs.w;
      const float ax = d[u].x * gc, ay = d[u].y * gc, az = d[u].z * gc, aw = d[u].w * gc;
      m[0] += ax; m[1] += ay; m[2] += az; m[3] += aw;
      m[4] += ax * t[u].x; m[5] += ay * t[u].y; m[6] += az * t[u].z; m[7] += aw * t[u].w;
      if (PART) {
        float sg = ok ? (d[u].x * t[u].x + d[u].y * t[u].y) + (d[u].z * t[u].z + d[u].w * t[u].w) : 0.f;
        float sb = ok ? (d[u].x + d[u].y) + (d[u].z + d[u].w) : 0.f;
        if (SUB == 1) {
          sg = wave_sum(sg);
          sb = wave_sum(sb);
        } else {   // the QW lanes of this channel sub-slice
#pragma unroll
          for (int o = 1; o < QW; o <<= 1) {
            sg += __shfl_xor(sg, o, 64);
            sb += __shfl_xor(sb, o, 64);
          }
        }
        if (ql == 0) {
          part[(int64_t)blockIdx.x * 2 * C + c0 + cb + u] = sg;
          part[(int64_t)blockIdx.x * 2 * C + C + c0 + cb + u] = sb;
        }
      }
    }
  }
  if (SUB > 1) {   // the sub-slices of one voxel quad sit QW lanes apart
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int o = QW; o < 64; o <<= 1) m[e] += __shfl_xor(m[e], o, 64);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[wave][e][lane] = m[e];
  __syncthreads();
  const float inv = 1.0f / (float)C;
  float mm[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) acc += red[w][e][lane];
    mm[e] = acc * inv;
  }
  auto finish = [&](int cb, int u, const float4& rr) {
    const float gc = g[c0 + cb + u];
    float4 o;
    o.x = rs.x * (d[u].x * gc - mm[0] - t[u].x * mm[4]);
    o.y = rs.y * (d[u].y * gc - mm[1] - t[u].y * mm[5]);
    o.z = rs.z * (d[u].z * gc - mm[2] - t[u].z * mm[6]);
    o.w = rs.w * (d[u].w * gc - mm[3] - t[u].w * mm[7]);
    if (GADD) { o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w; }
    if (ok) st4(gx + base + (int64_t)(cb + u) * V, o);
  };
  if constexpr (KEEP) {
    // the slice is in registers (one batch: cb = 0); the added gradient arrives in batches of <= 8 channels (16 more
    // float4 next to the 32 of a 16-channel slice would not fit 256 registers)
    constexpr int U2 = (GADD && CL > 8) ? 8 : CL;
#pragma unroll
    for (int hb = 0; hb < CL; hb += U2) {
      float4 r[GADD ? U2 : 1];
      if (GADD) {
#pragma unroll
        for (int u = 0; u < U2; ++u) r[GADD ? u : 0] = ld4(gadd + base + (int64_t)(hb + u) * V);
      }
#pragma unroll
      for (int u = 0; u < U2; ++u) finish(0, hb + u, r[GADD ? u : 0]);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll 1
    for (int cb = 0; cb < CL; cb += U) {
      float4 r[GADD ? U : 1];
      if (GADD) {
#pragma unroll
        for (int u = 0; u < U; ++u) r[GADD ? u : 0] = ld4(gadd + base + (int64_t)(cb + u) * V);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        t[u] = ld4(x + base + (int64_t)(cb + u) * V);
        d[u] = ld4(gl + base + (int64_t)(cb + u) * V);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        t[u].x = (t[u].x - mu.x) * rs.x; t[u].y = (t[u].y - mu.y) * rs.y;
        t[u].z = (t[u].z - mu.z) * rs.z; t[u].w = (t[u].w - mu.w) * rs.w;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) finish(cb, u, r[GADD ? u : 0]);
    }
  }
}

// (the fixed-order row reductions behind these kernels are jobs of the finish kernel: finish.h, FK_ROWS)

// per-channel sums over batch and voxels (bias gradient of the transposed conv, unet.py:123):
// part[(b*nchunk + chunk)][c] = Σ_{v in chunk} x[b, c, v]
template <typename AT>
__global__ __launch_bounds__(256) void rowsum_kernel(const AT* __restrict__ x, float* __restrict__ part, int C,
                                                     int64_t V, int nchunk) {
  __shared__ float red[4];
  const int c = blockIdx.x, chunk = blockIdx.y, b = blockIdx.z;
  const int64_t per = ((V / 4 + nchunk - 1) / nchunk) * 4;
  const int64_t v0 = chunk * per, v1 = min(V, v0 + per);
  const AT* xp = x + ((int64_t)b * C + c) * V;
  float s = 0.f;
  for (int64_t v = v0 + threadIdx.x * 4; v < v1; v += 1024) {
    const float4 t = ld4(xp + v);
    s += (t.x + t.y) + (t.z + t.w);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[((int64_t)b * nchunk + chunk) * C + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace fz

using namespace fz;

extern "C" int fz_reduce_rows(const float* part, int64_t rows, int n, float* out, float* tmp, fz_stream_t stream);

static unsigned ln_grid(int64_t total) {
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

template <typename AT>
static int ln_fwd_launch(const void* x, const float* gamma, const float* beta, void* y, float* stats, int B, int C,
                         int64_t V, float eps, fz_stream_t stream) {
  hipLaunchKernelGGL(ln_fwd_kernel<AT>, dim3(ln_grid(V / 4 * B)), dim3(256), 0, (hipStream_t)stream, (const AT*)x,
                     gamma, beta, (AT*)y, stats, B, C, V, eps);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int B,
                         int C, int64_t V, float eps, int act_dtype, fz_stream_t stream) {
  if (!x || !gamma || !beta || !y) return fail(FZ_E_ARG, "fz_ln_fwd: null pointer");
  if (B < 0 || C < 1 || V < 1) return fail(FZ_E_SHAPE, "fz_ln_fwd: bad sizes");
  if (V % 4) return fail(FZ_E_UNSUPPORTED, "fz_ln_fwd: voxel count must be a multiple of 4");
  if (B == 0) return FZ_OK;
  if (act_dtype == FZ_STORE_F32) return ln_fwd_launch<float>(x, gamma, beta, y, stats, B, C, V, eps, stream);
  if (act_dtype == FZ_STORE_BF16) return ln_fwd_launch<bf16>(x, gamma, beta, y, stats, B, C, V, eps, stream);
  return fail(FZ_E_ARG, "fz_ln_fwd: bad act_dtype");
}

// If gparams != NULL and C <= 64 the kernel also produces the affine gradients:
//   gparams[0..C) = gγ, gparams[C..2C) = gβ ; workspace must hold fz_ln_bwd_workspace_bytes().
// workspace: per-workgroup partial rows (C <= 64: <= 1024 rows; wider: one row per 64 quads) + 64
// rows of scratch for the two-stage reduce
static bool ln_slice_sub4(int64_t quads) { return (quads + 63) / 64 < 512; }
extern "C" int64_t fz_ln_bwd_workspace_bytes2(int B, int C, int64_t V) {
  const int64_t quads = (V / 4) * B;
  const int64_t wide = ln_slice_sub4(quads) ? (quads + 15) / 16 : (quads + 63) / 64;
  const int64_t rows = C < 64 ? 1024 : (wide > 1024 ? wide : 1024);  // C >= 64: one row per 64 quads
  return (rows + 64) * 2 * C * 4;
}
extern "C" int64_t fz_ln_bwd_workspace_bytes(int C) { return C <= 64 ? (int64_t)(1024 + 64) * 2 * C * 4 : 0; }

template <typename AT>
static int ln_bwd_launch(const void* gl_, const void* x_, const float* stats, const float* gamma,
                         const void* gadd_, void* gx_, float* gparams, void* workspace, int B, int C, int64_t V,
                         fz_stream_t stream) {
  const AT* gl = (const AT*)gl_;
  const AT* x = (const AT*)x_;
  const AT* gadd = (const AT*)gadd_;
  AT* gx = (AT*)gx_;
  hipStream_t st = (hipStream_t)stream;
  unsigned grid = ln_grid(V / 4 * B);
  {
    const int slice = 1;
    if (slice && (C == 64 || C == 128 || C == 256 || C == 512)) {
      const int64_t quads = (V / 4) * B;
      const bool sub4 = ln_slice_sub4(quads);   // fewer than 512 workgroups of 64 quads: 16 quads x 4 channel sub-slices each
      const unsigned gq = (unsigned)(sub4 ? (quads + 15) / 16 : (quads + 63) / 64);
      float* part = gparams ? (float*)workspace : nullptr;
#define FZ_LNS(CS, GA, PA)                                                                                                       \
  do {                                                                                                                           \
    if (sub4) hipLaunchKernelGGL((ln_bwd_slice_kernel<CS, GA, PA, AT, 4>), dim3(gq), dim3(512), 0, st, gl, x, stats, gamma, gadd, gx, part, B, V); \
    else hipLaunchKernelGGL((ln_bwd_slice_kernel<CS, GA, PA, AT, 1>), dim3(gq), dim3(512), 0, st, gl, x, stats, gamma, gadd, gx, part, B, V);      \
  } while (0)
#define FZ_LNS_F(CS)                                                                       \
  do {                                                                                     \
    if (gadd) { if (part) FZ_LNS(CS, true, true); else FZ_LNS(CS, true, false); }          \
    else { if (part) FZ_LNS(CS, false, true); else FZ_LNS(CS, false, false); }             \
  } while (0)
      if (C == 64) FZ_LNS_F(8); else if (C == 128) FZ_LNS_F(16); else if (C == 256) FZ_LNS_F(32); else FZ_LNS_F(64);
      FZ_LAUNCH_CHECK();
      if (gparams) {
        float* tmp = part + (int64_t)gq * 2 * C;
        return fz_reduce_rows(part, gq, 2 * C, gparams, tmp, stream);
      }
      return FZ_OK;
    }
  }
  if (gparams != nullptr && C <= 64) {
    if (grid > 1024) grid = 1024;
    float* part = (float*)workspace;
    if (C <= 32)
      hipLaunchKernelGGL((ln_bwd_kernel<32, AT>), dim3(grid), dim3(256), 0, st, gl, x, stats, gamma, gadd, gx, part, B, C, V);
    else
      hipLaunchKernelGGL((ln_bwd_kernel<64, AT>), dim3(grid), dim3(256), 0, st, gl, x, stats, gamma, gadd, gx, part, B, C, V);
    FZ_LAUNCH_CHECK();
    return finish_rows(part, (int)grid, 2 * C, (int)grid, 1, gparams, 0, st);
  }
  if (C > 64) {
    const int64_t quads = (V / 4) * B;
    const unsigned gq = (unsigned)((quads + 63) / 64);
    float* part = gparams ? (float*)workspace : nullptr;
    if (C >= 256)
      hipLaunchKernelGGL((ln_bwd_split_kernel<16, AT>), dim3(gq), dim3(1024), 0, st, gl, x, stats, gamma, gadd, gx, part, B, C, V);
    else
      hipLaunchKernelGGL((ln_bwd_split_kernel<8, AT>), dim3(gq), dim3(512), 0, st, gl, x, stats, gamma, gadd, gx, part, B, C, V);
    FZ_LAUNCH_CHECK();
    if (gparams) {
      float* tmp = part + (int64_t)gq * 2 * C;
      return fz_reduce_rows(part, gq, 2 * C, gparams, tmp, stream);
    }
    return FZ_OK;
  } else {
    hipLaunchKernelGGL((ln_bwd_kernel<0, AT>), dim3(grid), dim3(256), 0, st, gl, x, stats, gamma, gadd, gx,
                       (float*)nullptr, B, C, V);
  }
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_ln_bwd(const void* gl, const void* x, const float* stats, const float* gamma,
                         const void* gadd, void* gx, float* gparams, void* workspace, int B, int C, int64_t V,
                         int act_dtype, fz_stream_t stream) {
  if (!gl || !x || !stats || !gamma || !gx) return fail(FZ_E_ARG, "fz_ln_bwd: null pointer");
  if (B < 0 || C < 1 || V < 1) return fail(FZ_E_SHAPE, "fz_ln_bwd: bad sizes");
  if (V % 4) return fail(FZ_E_UNSUPPORTED, "fz_ln_bwd: voxel count must be a multiple of 4");
  if (gparams != nullptr && workspace == nullptr)
    return fail(FZ_E_ARG, "fz_ln_bwd: fused affine gradients need a workspace (fz_ln_bwd_workspace_bytes2)");
  if (B == 0) return FZ_OK;
  if (act_dtype == FZ_STORE_F32) return ln_bwd_launch<float>(gl, x, stats, gamma, gadd, gx, gparams, workspace, B, C, V, stream);
  if (act_dtype == FZ_STORE_BF16) return ln_bwd_launch<bf16>(gl, x, stats, gamma, gadd, gx, gparams, workspace, B, C, V, stream);
  return fail(FZ_E_ARG, "fz_ln_bwd: bad act_dtype");
}

// Two-stage when there are many rows: `tmp` (64 x n floats, may be NULL for rows <= 512) holds the
// per-slice sums of stage 1.
extern "C" int fz_reduce_rows(const float* part, int64_t rows, int n, float* out, float* tmp, fz_stream_t stream) {
  if (!part || !out || rows < 1 || n < 1 || rows > 0x7fffffff) return fail(FZ_E_ARG, "fz_reduce_rows: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 512 || tmp == nullptr) return finish_rows(part, (int)rows, n, (int)rows, 1, out, 0, st);
  const int groups = 64;
  const int rpg = (int)((rows + groups - 1) / groups);
  FinishJob j[2] = {finish_job(FK_ROWS, ((n + 31) / 32) * groups, 0), finish_job(FK_ROWS, (n + 31) / 32, 1)};
  j[0].u.rows = FinRows{part, tmp, (int)rows, n, rpg, (n + 31) / 32};
  j[1].u.rows = FinRows{tmp, out, groups, n, groups, (n + 31) / 32};
  return finish_run(j, 2, st);
}

// out[c] = Σ_{b,v} x[b,c,v]; part: workspace of B*nchunk*C floats with nchunk = fz_rowsum_chunks(V)
extern "C" int fz_rowsum_chunks(int64_t V) {
  int64_t n = V / 16384;
  if (n < 1) n = 1;
  if (n > 64) n = 64;
  return (int)n;
}

extern "C" int fz_rowsum(const void* x, float* part, float* out, int B, int C, int64_t V, int act_dtype,
                         fz_stream_t stream) {
  if (!x || !part || !out) return fail(FZ_E_ARG, "fz_rowsum: null pointer");
  if (B < 1 || C < 1 || V < 4 || (V % 4)) return fail(FZ_E_SHAPE, "fz_rowsum: bad sizes");
  const int nchunk = fz_rowsum_chunks(V);
  hipStream_t st = (hipStream_t)stream;
  if (act_dtype == FZ_STORE_BF16)
    hipLaunchKernelGGL(rowsum_kernel<bf16>, dim3(C, nchunk, B), dim3(256), 0, st, (const bf16*)x, part, C, V, nchunk);
  else if (act_dtype == FZ_STORE_F32)
    hipLaunchKernelGGL(rowsum_kernel<float>, dim3(C, nchunk, B), dim3(256), 0, st, (const float*)x, part, C, V, nchunk);
  else
    return fail(FZ_E_ARG, "fz_rowsum: bad act_dtype");
  FZ_LAUNCH_CHECK();
  return finish_rows(part, B * nchunk, C, B * nchunk, 1, out, 0, st);
}
