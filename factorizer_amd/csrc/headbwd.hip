// headbwd.hip — backward of a Linear(32 -> M), M <= 4, in ONE pass: the network's head (UNet out conv, kernel 1:
// factorizer/unet.py:253 through layers/linear.py:53-58) under autograd.
//   gx[c][v] = Σ_m W[m][c]·gy[m][v]        gW[m][c] = Σ_{b,v} gy[m][v]·x[c][v]        gb[m] = Σ_{b,v} gy[m][v]
// The GEMM family pads M = 3 rows to a 32-row MFMA tile: its input-gradient launch writes gx and its weight-gradient launch
// reads gy and x again at 2.8 TB/s (29 of 32 tile rows are zeros).  Here everything is VALU work on the lane's four voxels
// (2·M·32 FMAs per voxel — 0.4 GFMA per launch at 128^3) and the pass is bandwidth-bound: gy and x in, gx out, once.
// Persistent workgroups (2 per CU); the M·33 sums stay in registers across a workgroup's voxel quads, are added over its
// lanes and waves in a fixed order and leave as one row of `part` per workgroup; fz_chunk_reduce adds the rows in index order
// (no float atomics).
#include "gemm_common.h"

namespace fz {

constexpr int kHeadRow = 4 * 33;   // floats of one partial row: gW[m][c] at m*32 + c (m < 4), gb[m] at 128 + m

// M = compile-time row count of the register tiles (2 or 4), Mr <= M the layer's rows: rows Mr .. M-1 run on zeros (the
// three-row instantiation of the fp32 kernel spilled 19 registers where the four-row one fits — allocator heuristics)
template <int M, typename AT>
__global__ __launch_bounds__(256, 2) void head_bwd_kernel(const AT* __restrict__ gy, const AT* __restrict__ x,
                                                          const float* __restrict__ w, AT* __restrict__ gx,
                                                          float* __restrict__ part, int B, int64_t V, int Mr) {
  __shared__ float red[4][kHeadRow];
  __shared__ float sW[M * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < M * 32; i += 256) sW[i] = i < Mr * 32 ? w[i] : 0.f;
  __syncthreads();
  float gw[M][32];
  float gb[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    gb[m] = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) gw[m][c] = 0.f;
  }
  const unsigned quads = (unsigned)(V / 4), total = quads * (unsigned)B;   // host: below 2^31
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    // compiler-only fence: without it the loop-invariant LDS reads of W (M·32 per lane) are hoisted into VGPRs next to
    // the M·32 accumulators and spill
    asm volatile("" ::: "memory");
    const int b = (int)(i / quads);          // 32-bit division (the 64-bit one was ~200 vector instructions per iteration;
    const int64_t v = (int64_t)(i - (unsigned)b * quads) * 4;   // a stepped walk without any division spills 99 registers here)
    float g4[M][4];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      vload<4>(gy + ((int64_t)b * Mr + (m < Mr ? m : Mr - 1)) * V + v, g4[m]);   // (rows past Mr: a harmless re-read, zeroed)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        g4[m][e] = m < Mr ? g4[m][e] : 0.f;
        gb[m] += g4[m][e];
      }
    }
    const AT* xb = x + (int64_t)b * 32 * V + v;
    AT* ob = gx + (int64_t)b * 32 * V + v;
    constexpr int CB = 4;                  // channel loads in flight per batch
#pragma unroll
    for (int c0 = 0; c0 < 32; c0 += CB) {
      float xv[CB][4];
#pragma unroll
      for (int u = 0; u < CB; ++u) vload<4>(xb + (int64_t)(c0 + u) * V, xv[u]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        const int c = c0 + u;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = sW[c] * g4[0][e];
#pragma unroll
          for (int m = 1; m < M; ++m) t += sW[m * 32 + c] * g4[m][e];
          o[e] = t;
        }
        vstore<4>(ob + (int64_t)c * V, o);
#pragma unroll
        for (int m = 0; m < M; ++m)
          gw[m][c] += (g4[m][0] * xv[u][0] + g4[m][1] * xv[u][1]) + (g4[m][2] * xv[u][2] + g4[m][3] * xv[u][3]);
      }
    }
  }
  // lanes (eight sums at a time), then waves, then one row per workgroup
#pragma unroll
  for (int m = 0; m < M; ++m) {
#pragma unroll
    for (int c0 = 0; c0 < 32; c0 += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = gw[m][c0 + u];
      wave_sum8(t, lane);
      if (lane == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) red[wave][m * 32 + c0 + u] = t[u];
      }
    }
    const float s = wave_sum(gb[m]);
    if (lane == 0) red[wave][128 + m] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kHeadRow; e += 256) {
    const bool used = (e < 128) ? (e >> 5) < Mr : (e - 128) < Mr;
    part[(int64_t)blockIdx.x * kHeadRow + e] = used ? (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]) : 0.f;
  }
}

// ---- forward of the same layer: y[m][v] = b[m] + Σ_c W[m][c]·x[c][v], M <= 4 ----
// The persistent 32 -> 32 MFMA kernel (gemm_p32) computes 32 rows to keep 3: 64 fp32 MFMAs per 128 voxels that also block the SIMD's
// vector issue, 0.17 ms per step at 3.4 TB/s.  Here: 32·M FMAs per voxel on the lane's four voxels, eight channel
// loads in flight, bandwidth-bound (x in once, M rows out).
template <int M, typename AT>
__global__ __launch_bounds__(256) void head_fwd_kernel(const AT* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, AT* __restrict__ y, int B, int64_t V, int Mr) {
  __shared__ float sW[M * 32];
  __shared__ float sB[M];
  for (int i = threadIdx.x; i < M * 32; i += 256) sW[i] = i < Mr * 32 ? w[i] : 0.f;
  if (threadIdx.x < M) sB[threadIdx.x] = (bias != nullptr && (int)threadIdx.x < Mr) ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  const unsigned quads = (unsigned)(V / 4), stride = gridDim.x * 256u;
  unsigned first = blockIdx.x * 256u + threadIdx.x;
  int b = (int)(first / quads);
  unsigned qd = first - (unsigned)b * quads;
  for (; b < B; qd += stride) {
    while (qd >= quads) { qd -= quads; ++b; }
    if (b >= B) break;
    asm volatile("" ::: "memory");   // (keeps the M·32 weights in LDS instead of hoisting them into registers)
    const int64_t v = (int64_t)qd * 4;
    const AT* xb = x + (int64_t)b * 32 * V + v;
    float acc[M][4];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[m][e] = sB[m];
    constexpr int CB = 8;
#pragma unroll
    for (int c0 = 0; c0 < 32; c0 += CB) {
      float xv[CB][4];
#pragma unroll
      for (int u = 0; u < CB; ++u) vload<4>(xb + (int64_t)(c0 + u) * V, xv[u]);
#pragma unroll
      for (int u = 0; u < CB; ++u)
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const float wv = sW[m * 32 + c0 + u];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[m][e] += wv * xv[u][e];
        }
    }
#pragma unroll
    for (int m = 0; m < M; ++m)
      if (m < Mr) vstore<4>(y + ((int64_t)b * Mr + m) * V + v, acc[m]);
  }
}

}  // namespace fz

using namespace fz;

extern "C" int fz_head_bwd_rows(void) { return 512; }
extern "C" int64_t fz_head_bwd_workspace_bytes(void) { return (int64_t)fz_head_bwd_rows() * kHeadRow * (int64_t)sizeof(float); }

template <typename AT>
static int head_bwd_launch(const void* gy, const void* x, const float* w, void* gx, float* part, int B, int M, int64_t V,
                           hipStream_t st) {
  const unsigned grid = (unsigned)fz_head_bwd_rows();
#define FZ_HEAD(MM) hipLaunchKernelGGL((head_bwd_kernel<MM, AT>), dim3(grid), dim3(256), 0, st, (const AT*)gy, (const AT*)x, w, (AT*)gx, part, B, V, M)
  if (M <= 2) FZ_HEAD(2); else FZ_HEAD(4);
#undef FZ_HEAD
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

// gx, and the weight / bias gradients as `part` rows (fz_head_bwd_rows() rows of 132 floats: gW[m][c] at m*32 + c, gb[m] at
// 128 + m) for fz_chunk_reduce(part, rows, 132, out132, ...)
extern "C" int fz_head_bwd(const void* gy, const void* x, const float* w, void* gx, float* part, int B, int M, int C, int64_t V,
                           int act_dtype, fz_stream_t stream) {
  if (!gy || !x || !w || !gx || !part) return fail(FZ_E_ARG, "fz_head_bwd: null pointer");
  if (C != 32 || M < 1 || M > 4 || V < 4 || V % 4 != 0) return fail(FZ_E_UNSUPPORTED, "fz_head_bwd: needs C == 32, 1 <= M <= 4, V % 4 == 0");
  if ((int64_t)B * (V / 4) >= ((int64_t)1 << 31)) return fail(FZ_E_UNSUPPORTED, "fz_head_bwd: 2^31 or more voxel quads");
  if (B <= 0) return B == 0 ? FZ_OK : fail(FZ_E_SHAPE, "fz_head_bwd: negative batch");
  if (act_dtype == FZ_STORE_F32) return head_bwd_launch<float>(gy, x, w, gx, part, B, M, V, (hipStream_t)stream);
  if (act_dtype == FZ_STORE_BF16) return head_bwd_launch<bf16>(gy, x, w, gx, part, B, M, V, (hipStream_t)stream);
  return fail(FZ_E_ARG, "fz_head_bwd: act_dtype must be FZ_STORE_F32 or FZ_STORE_BF16");
}

template <typename AT>
static int head_fwd_launch(const void* x, const float* w, const float* bias, void* y, int B, int M, int64_t V, hipStream_t st) {
  const unsigned grid = 512u;   // persistent workgroups: two per CU
#define FZ_HEADF(MM) hipLaunchKernelGGL((head_fwd_kernel<MM, AT>), dim3(grid), dim3(256), 0, st, (const AT*)x, w, bias, (AT*)y, B, V, M)
  if (M <= 2) FZ_HEADF(2); else FZ_HEADF(4);
#undef FZ_HEADF
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

// y = W x + b for a Linear(32 -> M), M <= 4, on (B, 32, V) activations (the network's head, factorizer/unet.py:253); w row-major
// [M][32], bias [M] or null.  Called by fz_gemm for that shape; exported for callers that bind the head directly.
extern "C" int fz_head_fwd(const void* x, const float* w, const float* bias, void* y, int B, int M, int C, int64_t V, int act_dtype,
                           fz_stream_t stream) {
  if (!x || !w || !y) return fail(FZ_E_ARG, "fz_head_fwd: null pointer");
  if (C != 32 || M < 1 || M > 4 || V < 4 || V % 4 != 0) return fail(FZ_E_UNSUPPORTED, "fz_head_fwd: needs C == 32, 1 <= M <= 4, V % 4 == 0");
  if (V / 4 >= ((int64_t)1 << 31) || (int64_t)B * (V / 4) >= ((int64_t)1 << 31)) return fail(FZ_E_UNSUPPORTED, "fz_head_fwd: 2^31 or more voxel quads");
  if (B <= 0) return B == 0 ? FZ_OK : fail(FZ_E_SHAPE, "fz_head_fwd: negative batch");
  if (act_dtype == FZ_STORE_F32) return head_fwd_launch<float>(x, w, bias, y, B, M, V, (hipStream_t)stream);
  if (act_dtype == FZ_STORE_BF16) return head_fwd_launch<bf16>(x, w, bias, y, B, M, V, (hipStream_t)stream);
  return fail(FZ_E_ARG, "fz_head_fwd: act_dtype must be FZ_STORE_F32 or FZ_STORE_BF16");
}
