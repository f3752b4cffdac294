// pk_mfma_repro.hip — standalone reproducer attempt (no torch, no library) for the run-to-run differences of round 3
// (profiles/r03_two_stream_interaction.md): the eight-wave `upcat_bx_kernel` stopped replaying bit for bit exactly when its
// epilogue was compiled to `v_mov_b32` + `v_pk_add_f32 ... op_sel` register-pair shuffles (SLP vectorizer on; with
// -fno-slp-vectorize or -O1 the same program replays: profiles/r04_nondeterminism.md).  This kernel keeps only that shape:
// two waves per SIMD, each alternating a chain of v_mfma_f32_32x32x16_bf16 on two accumulators with an epilogue that pairs
// element r of both accumulators, adds a per-row constant and stores 8 bytes.  All operands are ones, so every output is
// known exactly: 16 * NMFMA + r.
// Build (two variants):  hipcc --offload-arch=gfx950 -O3 [-fno-slp-vectorize] -o pk_mfma_repro pk_mfma_repro.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bx8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef WAVES
#define WAVES 8
#endif
constexpr int NMFMA = 36;   // per accumulator and tile, as in the failing kernel (skip 12 + deep 24)

__global__ __launch_bounds__(WAVES * 64, 2) void repro_kernel(float* __restrict__ out, const float* __restrict__ bias,
                                                             const bx8* __restrict__ ops, int ntiles, int64_t plane) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, hk = lane >> 5;
  float badd[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) badd[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * hk];
  // operands from memory (all ones) so that nothing folds: 6 "weight" vectors, 6 "activation" vectors per lane
  bx8 a[6], b[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { a[i] = ops[i * 64 + lane]; b[i] = ops[(6 + i) * 64 + lane]; }
  for (int tile = blockIdx.x * WAVES + wave; tile < ntiles; tile += gridDim.x * WAVES) {
    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll
    for (int g = 0; g < NMFMA / 6; ++g)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + g + q) % 6], acc[q], 0, 0, 0);
    // epilogue of the failing kernel: rows (r & 3) + 8 (r >> 2) + 4 hk, the lane's two values as one 8-byte store
    float* yb = out + (int64_t)tile * 64 + 2 * j + (int64_t)(4 * hk) * plane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rb = (r & 3) + 8 * (r >> 2);
      const float add = badd[r];
      *reinterpret_cast<float2*>(yb + (int64_t)rb * plane) = make_float2(acc[0][r] + add, acc[1][r] + add);
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
  const int ntiles = 1 << 16;                 // 64 outputs per tile and row
  const int64_t plane = (int64_t)ntiles * 64;  // 32 rows
  float *out, *bias; bx8* ops;
  CK(hipMalloc(&out, 32 * plane * sizeof(float)));
  CK(hipMalloc(&bias, 32 * sizeof(float)));
  CK(hipMalloc(&ops, 12 * 64 * sizeof(bx8)));
  std::vector<float> hb(32);
  for (int i = 0; i < 32; ++i) hb[i] = (float)i;
  CK(hipMemcpy(bias, hb.data(), 32 * sizeof(float), hipMemcpyHostToDevice));
  std::vector<unsigned short> ho(12 * 64 * 8, 0x3f80);   // bf16 1.0
  CK(hipMemcpy(ops, ho.data(), ho.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> h((size_t)32 * plane);
  long total_bad = 0;
  for (int rep = 0; rep < 10; ++rep) {
    CK(hipMemset(out, 0xff, 32 * plane * sizeof(float)));
    hipLaunchKernelGGL(repro_kernel, dim3(256), dim3(WAVES * 64), 0, 0, out, bias, ops, ntiles, plane);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    long bad = 0; long first = -1;
    for (int row = 0; row < 32; ++row)
      for (int64_t n = 0; n < plane; ++n)
        if (h[(size_t)row * plane + n] != 16.f * 16 * NMFMA / 16 + (float)row) { if (first < 0) first = row * plane + n; ++bad; }
    printf("{\"waves\": %d, \"rep\": %d, \"bad\": %ld, \"first_row\": %ld, \"first_col_mod64\": %ld, \"got\": %g}\n", WAVES, rep, bad,
           first < 0 ? -1 : first / plane, first < 0 ? -1 : (first % plane) % 64, first < 0 ? 0.f : h[first]);
    total_bad += bad;
  }
  return total_bad ? 2 : 0;
}
