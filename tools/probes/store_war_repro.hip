// store_war_repro.hip — is the DATA of a 64-bit global store safe from a VALU write of its registers in the very next
// instruction(s) on gfx950?  (The ISA's documented hazard covers stores of MORE than 64 bits only, and hipcc schedules
// exactly this in the epilogue of the kernel that stopped replaying: profiles/r04_nondeterminism.md.)
//   each wave, per iteration:   v[10:11] <- (good, good) ; global_store_dwordx2 addr, v[10:11] ; NOPS ; v10 <- poison ; v11 <- poison
// Every stored dword must read back `good`.  Background load on the same SIMDs is selectable: other waves of the workgroup
// run v_mfma_f32_32x32x16_bf16 chains (MODE 1), stream loads (MODE 2) or both (MODE 3); WAVES waves per workgroup, one
// workgroup per CU.  Output: one JSON line per (mode, waves, nops) with the number of poisoned dwords and the lanes hit.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/store_war_repro tools/probes/store_war_repro.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bx8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr unsigned GOOD = 0x3f800000u, POISON = 0x7fc0dead;
constexpr int ITERS = 512;

// STORERS waves store; the remaining waves of the workgroup make background traffic.  NOPS: -1 = no instruction between the
// store and the overwrite, n >= 0 = s_nop n.
template <int MODE, int NOPS>
__global__ __launch_bounds__(512, 2) void war_kernel(unsigned* __restrict__ out, const float4* __restrict__ junk, float* sink,
                                                     int storers, size_t junk_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < storers) {
    unsigned long long addr = (unsigned long long)(out + ((size_t)(blockIdx.x * storers + wave) * ITERS * 64 + lane) * 2);
    for (int it = 0; it < ITERS; ++it) {
      if (NOPS < 0)
        asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %1\n\tglobal_store_dwordx2 %0, v[10:11], off\n\t"
                     "v_mov_b32 v10, %2\n\tv_mov_b32 v11, %2" ::"v"(addr), "v"(GOOD), "v"(POISON) : "v10", "v11", "memory");
      else if (NOPS == 0)
        asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %1\n\tglobal_store_dwordx2 %0, v[10:11], off\n\ts_nop 0\n\t"
                     "v_mov_b32 v10, %2\n\tv_mov_b32 v11, %2" ::"v"(addr), "v"(GOOD), "v"(POISON) : "v10", "v11", "memory");
      else
        asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %1\n\tglobal_store_dwordx2 %0, v[10:11], off\n\ts_nop 3\n\t"
                     "v_mov_b32 v10, %2\n\tv_mov_b32 v11, %2" ::"v"(addr), "v"(GOOD), "v"(POISON) : "v10", "v11", "memory");
      addr += 64 * 8;
    }
  } else {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bx8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)1.0f; b[e] = (__bf16)(float)(lane & 3); }
    float4 s = make_float4(0, 0, 0, 0);
    size_t i = ((size_t)blockIdx.x * 512 + threadIdx.x) % junk_n;
    for (int it = 0; it < ITERS * 2; ++it) {
      if (MODE & 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
      if (MODE & 2) {
        const float4 v = junk[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        i += 4096 * 61; if (i >= junk_n) i -= junk_n;
      }
    }
    if (acc[0] + s.x == 12345.678f) sink[0] = acc[1] + s.y;
  }
}

template <int MODE, int NOPS>
static void run(unsigned* out, size_t out_n, const float4* junk, float* sink, size_t junk_n, int waves, int storers) {
  CK(hipMemset(out, 0, out_n * 4));
  hipLaunchKernelGGL((war_kernel<MODE, NOPS>), dim3(256 * 2), dim3(waves * 64), 0, 0, out, junk, sink, storers, junk_n);
  CK(hipDeviceSynchronize());
  const size_t used = (size_t)512 * storers * ITERS * 64 * 2;
  std::vector<unsigned> h(used);
  CK(hipMemcpy(h.data(), out, used * 4, hipMemcpyDeviceToHost));
  long bad = 0, lane_hist[4] = {0, 0, 0, 0}, dword_hist[2] = {0, 0};
  for (size_t k = 0; k < used; ++k)
    if (h[k] != GOOD) { ++bad; ++lane_hist[((k / 2) % 64) / 16]; ++dword_hist[k & 1]; }
  printf("{\"mode\": %d, \"waves\": %d, \"storing_waves\": %d, \"wait_states\": %d, \"dwords\": %zu, \"bad\": %ld, "
         "\"bad_by_lane_quarter\": [%ld, %ld, %ld, %ld], \"bad_low_high_dword\": [%ld, %ld]}\n",
         MODE, waves, storers, NOPS < 0 ? 0 : (NOPS == 0 ? 1 : 4), used, bad, lane_hist[0], lane_hist[1], lane_hist[2], lane_hist[3],
         dword_hist[0], dword_hist[1]);
  fflush(stdout);
}

int main() {
  const size_t out_n = (size_t)512 * 8 * ITERS * 64 * 2, junk_n = (size_t)1 << 26;   // 1 GiB of float4 junk
  unsigned* out; float4* junk; float* sink;
  CK(hipMalloc(&out, out_n * 4));
  CK(hipMalloc(&junk, junk_n * 16));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(junk, 0, junk_n * 16));
  for (int rep = 0; rep < 2; ++rep) {
    run<0, -1>(out, out_n, junk, sink, junk_n, 8, 8);   // only storers, two waves per SIMD
    run<0, -1>(out, out_n, junk, sink, junk_n, 4, 4);   // one wave per SIMD
    run<1, -1>(out, out_n, junk, sink, junk_n, 8, 4);   // storers beside MFMA waves
    run<2, -1>(out, out_n, junk, sink, junk_n, 8, 4);   // storers beside streaming loads
    run<3, -1>(out, out_n, junk, sink, junk_n, 8, 4);
    run<3, -1>(out, out_n, junk, sink, junk_n, 8, 7);
    run<3, 0>(out, out_n, junk, sink, junk_n, 8, 4);
    run<3, 1>(out, out_n, junk, sink, junk_n, 8, 4);
  }
  return 0;
}
