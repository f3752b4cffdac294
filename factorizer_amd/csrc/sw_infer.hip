// sw_infer.hip — sliding-window inference stitching.
//
// The BraTS bundle runs the network through MONAI's SlidingWindowInfererAdapt(roi_size 128^3,
// sw_batch_size 2, overlap 0.5, mode "gaussian") (model_zoo/factorizer_brats23/configs/
// inference.yaml:96-102, train.yaml:206-212).  MONAI (pinned monai>=1.3 by the bundle metadata)
// is not under /root/reference; its published algorithm (monai/inferers/utils.py
// sliding_window_inference + compute_importance_map) is restated here:
//   windows  = dense grid of roi-sized patches, interval = int(roi * (1 - overlap)), last window
//              shifted back so that it ends at the image border;
//   weights  = separable Gaussian, sigma = 0.125 * roi, centred on the patch, clamped from below
//              at max(min nonzero, 1e-3);
//   output   = Σ_w weights · net(window_w)  /  Σ_w weights.
// Three data movements, all HBM-bound and one launch each per window:
//   gather      window ← volume[:, z0:z0+rd, y0:y0+rh, x0:x0+rw]
//   accumulate  out[:, window] += g · prob ; cnt[window] += g      (windows of one call overlap, so
//               they are accumulated by sequential launches: deterministic, no float atomics)
//   finalize    out /= cnt
#include "fz_common.h"

namespace fz {

struct SwGeom {
  int C;           // channels of the moved tensor
  int D, H, W;     // volume
  int rd, rh, rw;  // window
  int z0, y0, x0;  // window origin
  int aligned;     // W % 4 == 0 && x0 % 4 == 0: 16-byte vectors on the volume side
};

// volume-side access of 4 consecutive x: BraTS volumes are 240 x 240 x 155 and the last window of a
// row starts at 155 - 128 = 27, so the vector path cannot be assumed
__device__ __forceinline__ float4 vol_ld4(const float* p, bool aligned) {
  if (aligned) return *reinterpret_cast<const float4*>(p);
  return make_float4(p[0], p[1], p[2], p[3]);
}
__device__ __forceinline__ void vol_st4(float* p, float4 v, bool aligned) {
  if (aligned) { *reinterpret_cast<float4*>(p) = v; return; }
  p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
}

// one thread per 4 consecutive x of the window (rw % 4 == 0 and x0 % 4 == 0 → 16-byte vectors)
__global__ __launch_bounds__(256) void sw_gather_kernel(const float* __restrict__ vol, float* __restrict__ win, SwGeom g) {
  const int64_t qw = g.rw / 4;
  const int64_t total = (int64_t)g.C * g.rd * g.rh * qw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xq = (int)(i % qw);
    int64_t t = i / qw;
    const int y = (int)(t % g.rh); t /= g.rh;
    const int z = (int)(t % g.rd);
    const int c = (int)(t / g.rd);
    const int64_t src = (((int64_t)c * g.D + g.z0 + z) * g.H + g.y0 + y) * g.W + g.x0 + xq * 4;
    *reinterpret_cast<float4*>(win + i * 4) = vol_ld4(vol + src, g.aligned != 0);
  }
}

// gz, gy, gx: the three 1-D factors of the (already clamped-from-below per product) weight map
__global__ __launch_bounds__(256) void sw_accumulate_kernel(const float* __restrict__ prob, float* __restrict__ out,
                                                            float* __restrict__ cnt, const float* __restrict__ gz,
                                                            const float* __restrict__ gy, const float* __restrict__ gx,
                                                            float wmin, SwGeom g) {
  const int64_t qw = g.rw / 4;
  const int64_t plane = (int64_t)g.rd * g.rh * qw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (int64_t)gridDim.x * blockDim.x) {
    const int xq = (int)(i % qw);
    int64_t t = i / qw;
    const int y = (int)(t % g.rh);
    const int z = (int)(t / g.rh);
    const float wzy = gz[z] * gy[y];
    const float4 wx = *reinterpret_cast<const float4*>(gx + xq * 4);
    float4 w = make_float4(fmaxf(wzy * wx.x, wmin), fmaxf(wzy * wx.y, wmin), fmaxf(wzy * wx.z, wmin), fmaxf(wzy * wx.w, wmin));
    const int64_t dst = (((int64_t)g.z0 + z) * g.H + g.y0 + y) * g.W + g.x0 + xq * 4;
    const bool al = g.aligned != 0;
    float4 cv = vol_ld4(cnt + dst, al);
    cv.x += w.x; cv.y += w.y; cv.z += w.z; cv.w += w.w;
    vol_st4(cnt + dst, cv, al);
    const int64_t V = (int64_t)g.D * g.H * g.W;
    for (int c = 0; c < g.C; ++c) {
      const float4 p = *reinterpret_cast<const float4*>(prob + ((int64_t)c * plane + i) * 4);
      float4 o = vol_ld4(out + c * V + dst, al);
      o.x += w.x * p.x; o.y += w.y * p.y; o.z += w.z * p.z; o.w += w.w * p.w;
      vol_st4(out + c * V + dst, o, al);
    }
  }
}

__global__ __launch_bounds__(256) void sw_finalize_kernel(float* __restrict__ out, const float* __restrict__ cnt, int C,
                                                          int64_t V) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
    const float cv = cnt[i];
    for (int c = 0; c < C; ++c) out[(int64_t)c * V + i] /= cv;
  }
}

static int sw_check(const char* who, int C, int D, int H, int W, int rd, int rh, int rw, int z0, int y0, int x0) {
  if (C < 1 || D < 1 || H < 1 || W < 1 || rd < 1 || rh < 1 || rw < 1) return fail(FZ_E_SHAPE, "fz_sw: sizes must be positive");
  if (z0 < 0 || y0 < 0 || x0 < 0 || z0 + rd > D || y0 + rh > H || x0 + rw > W)
    return fail(FZ_E_SHAPE, "fz_sw: window outside the volume");
  if (rw % 4) return fail(FZ_E_UNSUPPORTED, "fz_sw: window width must be a multiple of 4");
  (void)who;
  return FZ_OK;
}

static unsigned sw_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace fz

using namespace fz;

extern "C" int fz_sw_gather(const float* vol, float* win, int C, int D, int H, int W, int rd, int rh, int rw, int z0,
                            int y0, int x0, fz_stream_t stream) {
  if (!vol || !win) return fail(FZ_E_ARG, "fz_sw_gather: null pointer");
  int rc = sw_check("gather", C, D, H, W, rd, rh, rw, z0, y0, x0);
  if (rc != FZ_OK) return rc;
  SwGeom g{C, D, H, W, rd, rh, rw, z0, y0, x0, ((W % 4) == 0 && (x0 % 4) == 0) ? 1 : 0};
  hipLaunchKernelGGL(sw_gather_kernel, dim3(sw_grid((int64_t)C * rd * rh * rw / 4)), dim3(256), 0, (hipStream_t)stream,
                     vol, win, g);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_sw_accumulate(const float* prob, float* out, float* cnt, const float* gz, const float* gy,
                                const float* gx, float wmin, int C, int D, int H, int W, int rd, int rh, int rw,
                                int z0, int y0, int x0, fz_stream_t stream) {
  if (!prob || !out || !cnt || !gz || !gy || !gx) return fail(FZ_E_ARG, "fz_sw_accumulate: null pointer");
  int rc = sw_check("accumulate", C, D, H, W, rd, rh, rw, z0, y0, x0);
  if (rc != FZ_OK) return rc;
  SwGeom g{C, D, H, W, rd, rh, rw, z0, y0, x0, ((W % 4) == 0 && (x0 % 4) == 0) ? 1 : 0};
  hipLaunchKernelGGL(sw_accumulate_kernel, dim3(sw_grid((int64_t)rd * rh * rw / 4)), dim3(256), 0, (hipStream_t)stream,
                     prob, out, cnt, gz, gy, gx, wmin, g);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_sw_finalize(float* out, const float* cnt, int C, int64_t V, fz_stream_t stream) {
  if (!out || !cnt) return fail(FZ_E_ARG, "fz_sw_finalize: null pointer");
  if (C < 1 || V < 1) return fail(FZ_E_SHAPE, "fz_sw_finalize: sizes must be positive");
  hipLaunchKernelGGL(sw_finalize_kernel, dim3(sw_grid(V)), dim3(256), 0, (hipStream_t)stream, out, cnt, C, V);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}
