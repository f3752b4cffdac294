// lds_stomp.hip — a kernel that does nothing but own `lds_bytes` of dynamic LDS per 256-thread workgroup and keep
// rewriting all of it (no global traffic beyond one word per workgroup).  Run on a second stream beside a kernel with
// long-lived LDS state (the NMF backward's history) it answers one question: can a co-resident workgroup of ANOTHER
// kernel change the contents of your LDS allocation?
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probes/bin/liblds_stomp.so tools/probes/lds_stomp.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void lds_stomp_kernel(int n_floats, int iters, float* sink) {
  extern __shared__ float buf[];
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < n_floats; i += 256) buf[i] = (float)(it * 131 + i) * 1e30f;   // huge values: unmissable
    __syncthreads();
    for (int i = threadIdx.x; i < n_floats; i += 256) acc += buf[i] * 1e-38f;
    __syncthreads();
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
}

extern "C" int lds_stomp(int lds_bytes, int blocks, int iters, float* sink, void* stream) {
  if (lds_bytes > 65536) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lds_stomp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return -1;
  }
  hipLaunchKernelGGL(lds_stomp_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, lds_bytes / 4, iters, sink);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
