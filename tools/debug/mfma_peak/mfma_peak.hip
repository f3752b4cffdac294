// Raw fp32 MFMA issue rate of the chip: every wave runs ITER x NI independent
// v_mfma_f32_32x32x2_f32 on register operands, no memory traffic.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/debug/mfma_peak/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NI>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc[NI];
  for (int i = 0; i < NI; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float av = a + threadIdx.x * 1e-6f, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NI; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NI>
void run(int wgs, const char* tag) {
  float* out; hipMalloc(&out, (size_t)wgs * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NI>, dim3(wgs), dim3(256), 0, 0, out, 100, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NI>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)wgs * 4 * iters * NI * 2.0 * 32 * 32 * 2;
  printf("%s NI=%d wgs=%d: %.2f ms  %.1f TFLOP/s\n", tag, NI, wgs, ms, fl / ms / 1e9);
  hipFree(out);
}
int main() {
  run<8>(256, "1 wave/SIMD");
  run<8>(512, "2 waves/SIMD");
  run<4>(1024, "4 waves/SIMD");
  run<2>(512, "2 waves/SIMD dependent-ish");
  return 0;
}
