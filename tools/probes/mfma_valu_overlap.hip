// Do VALU and MFMA instructions overlap on gfx950 — inside one wave, and between the two waves of a SIMD?
// The fused MLP kernels spend ~44 % of their time with the MFMA pipe busy and ~43 % with the VALU busy (profiles/r04_p2_pmc_sq.md)
// and the sum, not the maximum, is what the launch takes.  Five loops of the same work per iteration (NV independent fp32 FMAs on
// 16 registers per lane = NV x 4 issue cycles; NM MFMAs on four independent accumulators):
//   valu   only the FMAs                         mfma   only the MFMAs
//   mixed  one MFMA, then NV / NM FMAs, repeated (sched_group_barrier)   phased  all MFMAs, then all FMAs (sched_barrier between)
// each at one and at two waves per SIMD (grid = 256 resp. 512 workgroups of 256 threads).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bx8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int KIND>   // KIND 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_32x32x16_bf16
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float s) {
  constexpr int NV = 64, NM = 8;
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  f32x4 c4[4] = {};
  f32x16 c16[4] = {};
  const float ma = threadIdx.x * 0.5f, mb = s;
  bx8 ba, bb;
  for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(ma + i); bb[i] = (__bf16)(mb + i); }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 3) {           // FMAs only / phased
      if (MODE == 3) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          if (KIND == 0) c4[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, c4[m & 3], 0, 0, 0);
          else c16[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, c16[m & 3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) a[v & 15] = __builtin_fmaf(a[v & 15], s, 0.25f);
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 1) {                 // MFMAs only
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if (KIND == 0) c4[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, c4[m & 3], 0, 0, 0);
        else c16[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, c16[m & 3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {                                // mixed: 1 MFMA : NV / NM FMAs
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if (KIND == 0) c4[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, c4[m & 3], 0, 0, 0);
        else c16[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, c16[m & 3], 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) a[v & 15] = __builtin_fmaf(a[v & 15], s, 0.25f);
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, NV / NM, 0);  // NV / NM VALU
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += a[i];
  for (int i = 0; i < 4; ++i) { r += c4[i][0] + c4[i][3]; r += c16[i][0] + c16[i][15]; }
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE, int KIND>
static void run(const char* name, float* out, int wgs) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, KIND>), dim3(wgs), dim3(256), 0, 0, out, 100, 1.0001f);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, KIND>), dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const double cyc = ms * 1e-3 * pr.clockRate * 1e3 / iters;
  printf("{\"loop\": \"%s\", \"mfma\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"cycles_per_iteration_at_%dMHz\": %.0f}\n", name,
         KIND ? "32x32x16_bf16" : "16x16x4_f32", wgs / 256, ms, pr.clockRate / 1000, cyc);
}

int main() {
  float* out; CK(hipMalloc(&out, 512 * 256 * 4));
  for (int wgs : {256, 512}) {
    run<0, 0>("valu", out, wgs);
    run<1, 0>("mfma", out, wgs);  run<2, 0>("mixed", out, wgs);  run<3, 0>("phased", out, wgs);
    run<1, 1>("mfma", out, wgs);  run<2, 1>("mixed", out, wgs);  run<3, 1>("phased", out, wgs);
  }
  return 0;
}
