// optim.hip — AdamW over ONE flat parameter buffer.
//
// The training recipe is torch.optim.AdamW(lr 1e-4, weight_decay 1e-5)
// (model_zoo/factorizer_brats23/configs/train.yaml:72-76).  With parameters, gradients and both
// moments each in one flat fp32 buffer (factorizer_amd/parallel.py, training.py) the whole update is
// a single elementwise pass — 7 floats of traffic per parameter, 164 MB for the 5.86 M-parameter
// README model — instead of a multi-tensor launch sequence over 182 tensors.
//   p ← p·(1 − lr·wd);  m ← β1 m + (1−β1) g;  v ← β2 v + (1−β2) g²;
//   p ← p − (lr / (1−β1^t)) · m / (sqrt(v)/sqrt(1−β2^t) + eps)          (torch.optim.AdamW, no amsgrad)
#include "fz_common.h"

namespace fz {

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float step_size,
                                                    float inv_sqrt_bc2, float gscale) {
  const int64_t nv = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = *reinterpret_cast<float4*>(p + i * 4);
    const float4 gg = *reinterpret_cast<const float4*>(g + i * 4);
    float4 mm = *reinterpret_cast<float4*>(m + i * 4);
    float4 vv = *reinterpret_cast<float4*>(v + i * 4);
    float ps[4] = {pp.x, pp.y, pp.z, pp.w}, gs[4] = {gg.x, gg.y, gg.z, gg.w}, ms[4] = {mm.x, mm.y, mm.z, mm.w},
          vs[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gs[e] * gscale;
      ps[e] *= 1.0f - lr * wd;
      ms[e] = beta1 * ms[e] + (1.0f - beta1) * ge;
      vs[e] = beta2 * vs[e] + (1.0f - beta2) * ge * ge;
      ps[e] -= step_size * (ms[e] / (sqrtf(vs[e]) * inv_sqrt_bc2 + eps));
    }
    *reinterpret_cast<float4*>(p + i * 4) = make_float4(ps[0], ps[1], ps[2], ps[3]);
    *reinterpret_cast<float4*>(m + i * 4) = make_float4(ms[0], ms[1], ms[2], ms[3]);
    *reinterpret_cast<float4*>(v + i * 4) = make_float4(vs[0], vs[1], vs[2], vs[3]);
  }
  // tail (n not a multiple of 4)
  const int64_t t = nv * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    const float ge = g[t] * gscale;
    float pe = p[t] * (1.0f - lr * wd);
    const float me = beta1 * m[t] + (1.0f - beta1) * ge;
    const float ve = beta2 * v[t] + (1.0f - beta2) * ge * ge;
    pe -= step_size * (me / (sqrtf(ve) * inv_sqrt_bc2 + eps));
    p[t] = pe; m[t] = me; v[t] = ve;
  }
}

}  // namespace fz

using namespace fz;

extern "C" int fz_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                             fz_stream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return fail(FZ_E_ARG, "fz_adamw_step: null pointer");
  if (n < 0 || step < 1) return fail(FZ_E_ARG, "fz_adamw_step: n >= 0 and step >= 1");
  if (n == 0) return FZ_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_size, inv_sqrt_bc2, grad_scale);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}
