// pk_opsel_repro.hip — minimal standalone test of the instruction the assembly bisect singled out
// (tools/probes/upcat_asm_variants.py, profiles/r04_nondeterminism.md):
//     v_pk_add_f32 v[d:d+1], v[a:a+1], v[b:b+1] op_sel:[0,1]        lo = a.lo + b.HI ; hi = a.hi + b.hi
// Replacing exactly these instructions by two v_add_f32 makes the eight-wave kernel replay bit for bit; s_nop padding around
// them does not.  Here: waves 0 .. P-1 of a workgroup execute the instruction on known operands and check the result in
// registers; the other waves run v_mfma_f32_32x32x16_bf16 chains (or nothing).  Output: mismatch counts per form.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/pk_opsel_repro tools/probes/pk_opsel_repro.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __bf16 bx8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// FORM 0: op_sel:[0,1]   1: op_sel_hi:[1,0]   2: no modifier
template <int FORM>
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  if (FORM == 0) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  else if (FORM == 1) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
  else asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

template <int FORM>
__global__ __launch_bounds__(512, 2) void k(unsigned* __restrict__ bad, int pk_waves, int iters, int with_mfma) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < pk_waves) {
    unsigned nbad_lo = 0, nbad_hi = 0;
    for (int it = 0; it < iters; ++it) {
      f32x2 a = {(float)(lane + it), (float)(2 * lane + 1)}, b = {(float)(3 * it + 7), (float)(5 * lane + it)};
      asm volatile("" : "+v"(a), "+v"(b));
      const f32x2 d = pk_add<FORM>(a, b);
      const float elo = FORM == 0 ? a[0] + b[1] : a[0] + b[0];
      const float ehi = FORM == 1 ? a[1] + b[0] : a[1] + b[1];
      nbad_lo += d[0] != elo;
      nbad_hi += d[1] != ehi;
    }
    if (nbad_lo) atomicAdd(bad + (lane >> 4), nbad_lo);        // by lane quarter
    if (nbad_hi) atomicAdd(bad + 4 + (lane >> 4), nbad_hi);
  } else if (with_mfma) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bx8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)1.0f; b[e] = (__bf16)(float)(lane & 3); }
    for (int it = 0; it < iters / 8; ++it)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (acc[0] == 12345.678f) bad[15] = 1;
  }
}

template <int FORM>
static void run(unsigned* bad, int waves, int pk_waves, int with_mfma) {
  CK(hipMemset(bad, 0, 64));
  hipLaunchKernelGGL((k<FORM>), dim3(256 * 2), dim3(waves * 64), 0, 0, bad, pk_waves, 1 << 16, with_mfma);
  CK(hipDeviceSynchronize());
  unsigned h[16];
  CK(hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost));
  printf("{\"form\": \"%s\", \"waves_per_workgroup\": %d, \"pk_waves\": %d, \"mfma_beside\": %d, \"wrong_lo_by_lane_quarter\": [%u, %u, %u, %u], "
         "\"wrong_hi_by_lane_quarter\": [%u, %u, %u, %u]}\n", FORM == 0 ? "op_sel:[0,1]" : FORM == 1 ? "op_sel_hi:[1,0]" : "plain", waves, pk_waves,
         with_mfma, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  fflush(stdout);
}

int main() {
  unsigned* bad;
  CK(hipMalloc(&bad, 64));
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(bad, 8, 4, 1); run<1>(bad, 8, 4, 1); run<2>(bad, 8, 4, 1);   // beside MFMA waves, two waves per SIMD
    run<0>(bad, 8, 8, 0);                                               // two pk waves per SIMD, no MFMA
    run<0>(bad, 4, 4, 0);                                               // one wave per SIMD
    run<0>(bad, 8, 1, 1); run<0>(bad, 8, 7, 1);
  }
  return 0;
}
