// Raw MFMA issue rates: fp32 32x32x2 vs bf16 32x32x16 (gfx950), register operands, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/debug/mfma_peak/mfma_peak2.hip -o /tmp/mfma_peak2 && /tmp/mfma_peak2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NI, bool BF>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc[NI];
  for (int i = 0; i < NI; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float av = a + threadIdx.x * 1e-6f, bv = b;
  bf16x8 ab, bb;
  for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)(av + e); bb[e] = (__bf16)(bv); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NI; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NI, bool BF>
void run(int wgs, const char* tag) {
  float* out; hipMalloc(&out, (size_t)wgs * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NI, BF>), dim3(wgs), dim3(256), 0, 0, out, 100, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NI, BF>), dim3(wgs), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)wgs * 4 * iters * NI * 2.0 * 32 * 32 * (BF ? 16 : 2);
  printf("%s %s NI=%d wgs=%d: %.2f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", BF ? "bf16 32x32x16" : "f32 32x32x2", tag, NI, wgs, ms,
         fl / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * NI * (wgs / 256.0)));
  hipFree(out);
}
int main() {
  run<8, false>(256, "1 wave/SIMD");
  run<8, false>(512, "2 waves/SIMD");
  run<8, true>(256, "1 wave/SIMD");
  run<8, true>(512, "2 waves/SIMD");
  run<4, true>(1024, "4 waves/SIMD");
  return 0;
}
