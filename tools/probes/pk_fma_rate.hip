// pk_fma_rate.hip — does v_pk_fma_f32 double the fp32 FMA rate of a wave on gfx950, or does a plain v_fma_f32 already issue at the
// full 64-flop/cycle/SIMD rate?  (Decides whether packing the two ranks of the rank-2 NMF wave program is worth a rewrite:
// profiles/r06_cfg5.md.)  Each thread runs N dependent-chain-free FMAs on 16 accumulators (scalar form) or 8 accumulator pairs
// (packed form); 256 CUs x 8 waves per SIMD resident.  Prints GFMA/s (lane-FMAs) for both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_scalar(float* out, int iters, float a, float b) {
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (float)threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_packed(float* out, int iters, float a, float b) {
  f2 acc[8];
  const f2 av = {a, a * 1.0001f}, bv = {b, b * 0.9999f};
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f2{(float)threadIdx.x + i, (float)threadIdx.x - i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  const int blocks = 256 * 8, iters = 20000;
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int form = 0; form < 2; ++form) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (form == 0) hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
      else hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double fmas = (double)blocks * 256 * iters * 16;   // lane-FMAs (both forms do 16 per thread and iteration)
      if (rep == 2) printf("{\"form\": \"%s\", \"ms\": %.3f, \"TFLOPs\": %.1f}\n", form == 0 ? "v_fma_f32" : "v_pk_fma_f32", ms, 2.0 * fmas / ms / 1e9);
    }
  }
  return 0;
}
