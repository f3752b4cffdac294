// bx6_accuracy.hip — accuracy of fp32 products emulated on the bf16 matrix pipe (gfx950).
//   x = x1 + x2 + x3 (three bf16, round-to-nearest at each level: exact for normal fp32)
//   terms 6: a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1)      -> dropped a2b3 + a3b2 + a3b3 <= 2^-23 |ab|
//   terms 3: a1b1 + a1b2 + a2b1 ; terms 9: all nine
// against v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) and an fp64 host reference.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/bx6_accuracy tools/probes/bx6_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ inline void split3(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    f32x2 v = {x[i], x[i + 1]};
    bf16x2 h2 = __builtin_convertvector(v, bf16x2);
    f32x2 r = v - __builtin_convertvector(h2, f32x2);
    bf16x2 m2 = __builtin_convertvector(r, bf16x2);
    f32x2 r2 = r - __builtin_convertvector(m2, f32x2);
    bf16x2 l2 = __builtin_convertvector(r2, bf16x2);
    h[i] = h2[0]; h[i + 1] = h2[1]; m[i] = m2[0]; m[i + 1] = m2[1]; l[i] = l2[0]; l[i + 1] = l2[1];
  }
}

// C[32][32] = A[32][K] * B[K][32];  A row-major, B row-major.  one wave.
template <int MODE>  // 0: fp32 mfma, 3/6/9: split terms, 1: plain bf16
__global__ void gemm32(const float* A, const float* B, float* C, int K) {
  const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (MODE == 0) {
    for (int k = 0; k < K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j * K + k + h], B[(k + h) * 32 + j], acc, 0, 0, 0);
  } else {
    for (int k = 0; k < K; k += 16) {
      float a[8], b[8];
      for (int e = 0; e < 8; ++e) { a[e] = A[j * K + k + 8 * h + e]; b[e] = B[(k + 8 * h + e) * 32 + j]; }
      bf16x8 ah, am, al, bh, bm, bl;
      split3(a, ah, am, al);
      split3(b, bh, bm, bl);
      // smallest terms first
      if (MODE >= 9) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bm, acc, 0, 0, 0);
      }
      if (MODE >= 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
      }
      if (MODE >= 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}

int main() {
  std::mt19937 rng(7);
  const int Ks[] = {64, 256, 1024, 4096};
  printf("{\n");
  bool first = true;
  for (int dist = 0; dist < 3; ++dist)
    for (int K : Ks) {
      std::vector<float> A(32 * K), B(K * 32);
      std::normal_distribution<float> nd(0.f, 1.f);
      std::uniform_real_distribution<float> ud(0.f, 1.f);
      for (auto& v : A) v = dist == 0 ? nd(rng) : (dist == 1 ? ud(rng) : nd(rng) * std::exp(3.f * nd(rng)));
      for (auto& v : B) v = dist == 0 ? nd(rng) : (dist == 1 ? ud(rng) : nd(rng) * std::exp(3.f * nd(rng)));
      std::vector<double> ref(1024), mag(1024);
      for (int i = 0; i < 32; ++i)
        for (int jj = 0; jj < 32; ++jj) {
          double s = 0, m = 0;
          for (int k = 0; k < K; ++k) { double p = (double)A[i * K + k] * (double)B[k * 32 + jj]; s += p; m += std::fabs(p); }
          ref[i * 32 + jj] = s; mag[i * 32 + jj] = m;
        }
      float *dA, *dB, *dC;
      hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      const int modes[] = {0, 1, 3, 6, 9};
      for (int mode : modes) {
        switch (mode) {
          case 0: hipLaunchKernelGGL(gemm32<0>, 1, 64, 0, 0, dA, dB, dC, K); break;
          case 1: hipLaunchKernelGGL(gemm32<1>, 1, 64, 0, 0, dA, dB, dC, K); break;
          case 3: hipLaunchKernelGGL(gemm32<3>, 1, 64, 0, 0, dA, dB, dC, K); break;
          case 6: hipLaunchKernelGGL(gemm32<6>, 1, 64, 0, 0, dA, dB, dC, K); break;
          default: hipLaunchKernelGGL(gemm32<9>, 1, 64, 0, 0, dA, dB, dC, K); break;
        }
        std::vector<float> C(1024);
        hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
        double mx = 0, rms = 0, mxn = 0;
        for (int i = 0; i < 1024; ++i) {
          const double e = std::fabs((double)C[i] - ref[i]) / mag[i];   // relative to sum |a||b|
          mx = std::max(mx, e); rms += e * e;
          mxn = std::max(mxn, std::fabs((double)C[i] - ref[i]) / (std::fabs(ref[i]) + 1e-30));
        }
        rms = std::sqrt(rms / 1024);
        printf("%s \"dist%d_K%d_mode%d\": {\"max_err_over_sum_abs\": %.3e, \"rms_err_over_sum_abs\": %.3e, \"max_rel_err\": %.3e}",
               first ? "" : ",\n", dist, K, mode, mx, rms, mxn);
        first = false;
      }
      hipFree(dA); hipFree(dB); hipFree(dC);
    }
  printf("\n}\n");
  return 0;
}
