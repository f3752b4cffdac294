// Does the power-of-two distance between the channel planes of a channels-first activation cost HBM bandwidth?
// (B, C, 128^3) fp32: plane stride 8 MiB exactly — the 32 loads a lane has in flight (one per channel, same in-plane offset) differ
// only in address bits >= 23.  Kernel: persistent workgroups, lane = 4 voxels, sum over C = 32 planes, one plane written (32 : 1).
// Plane stride = 8 MiB + pad for several pads.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int CB>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, float* __restrict__ y, size_t plane_stride, size_t sample_stride,
                                         int B, unsigned quads) {
  const unsigned stride = gridDim.x * 256u;
  for (int b = 0; b < B; ++b)
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < quads; q += stride) {
      const float* p = x + b * sample_stride + (size_t)q * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c0 = 0; c0 < 32; c0 += CB) {
        float4 v[CB];
#pragma unroll
        for (int u = 0; u < CB; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(c0 + u) * plane_stride);
#pragma unroll
        for (int u = 0; u < CB; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      *reinterpret_cast<float4*>(y + (size_t)b * quads * 4 + (size_t)q * 4) = acc;
    }
}

int main() {
  const int B = 2, C = 32;
  const size_t V = 128ull * 128 * 128;
  const size_t pads[] = {0, 64, 256, 1024, 4096, 16384, 65536, 1048576 + 256};   // floats
  float* y; CK(hipMalloc(&y, B * V * 4));
  for (size_t pad : pads) {
    const size_t ps = V + pad, ss = ps * C;
    float* x; CK(hipMalloc(&x, ss * B * 4 + 1024));
    CK(hipMemset(x, 0, ss * B * 4));
    for (int wgs : {512, 1024}) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<8>, dim3(wgs), dim3(256), 0, 0, x, y, ps, ss, B, (unsigned)(V / 4));
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<8>, dim3(wgs), dim3(256), 0, 0, x, y, ps, ss, B, (unsigned)(V / 4));
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
      printf("{\"plane_stride_bytes\": %zu, \"pad_floats\": %zu, \"workgroups\": %d, \"ms\": %.4f, \"GBps\": %.1f}\n", ps * 4, pad, wgs, ms,
             (double)(B * V * 4) * (C + 1) / (ms * 1e-3) / 1e9);
    }
    CK(hipFree(x));
  }
  return 0;
}
