// pk_opsel_repro2.hip — second standalone attempt: the failing instruction IN ITS REGISTER CONTEXT.
// In the eight-wave upcat kernel the four instructions whose replacement by scalar adds restores bitwise replay are
//     v_mov_b32 v16, v1 ; ... ; v_pk_add_f32 v[0:1],  v[16:17], v[32:33] op_sel:[0,1]
//     v_mov_b32 v20, v5 ; ... ; v_pk_add_f32 v[2:3],  v[20:21], v[36:37] op_sel:[0,1]      (and v[24:25], v[28:29])
// i.e. src0.lo freshly moved out of an MFMA accumulator register, src0 / src1 pairs in the SAME VGPR banks (16 | 32, 17 | 33),
// src1 the result of a ds_read_b128, every wave of the SIMD alternating MFMA chains on v[0:31] with this epilogue.
// Here every wave runs: MFMA chain into v[0:15] and v[16:31] (all-ones operands: every element = 16 * n), then the exact
// instruction pair on fixed registers, then compares in registers.  WAVES = 8 (two per SIMD) or 4.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/pk_opsel_repro2 tools/probes/pk_opsel_repro2.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// SEL 1: op_sel:[0,1] (the failing form); 0: two v_add_f32 (the repaired form)
template <int SEL>
__global__ __launch_bounds__(512, 2) void k(unsigned* __restrict__ bad, int iters, int nm) {
  __shared__ __attribute__((aligned(16))) float bias[64];
  if (threadIdx.x < 64) bias[threadIdx.x] = (float)(threadIdx.x + 1);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lds_addr = 16 * (lane & 3);   // (the only LDS object: offset 0) ds_read_b128 of bias[4k .. 4k+3], k = lane & 3
  unsigned wrong_lo = 0, wrong_hi = 0;
  for (int it = 0; it < iters + wave; ++it) {   // (different trip counts: the waves of a SIMD drift out of phase)
    unsigned lo, hi;
    asm volatile(
        // operands: all ones in bf16 (0x3f80 pairs) -> each MFMA adds 16 to every accumulator element
        "v_mov_b32 v40, 0x3f803f80\n\tv_mov_b32 v41, 0x3f803f80\n\tv_mov_b32 v42, 0x3f803f80\n\tv_mov_b32 v43, 0x3f803f80\n\t"
        "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
        "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t"
        "v_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t"
        "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\tv_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t"
        "s_mov_b32 s20, %[nm]\n"
        "1:\n\t"
        "v_mfma_f32_32x32x16_bf16 v[0:15], v[40:43], v[40:43], v[0:15]\n\t"
        "v_mfma_f32_32x32x16_bf16 v[16:31], v[40:43], v[40:43], v[16:31]\n\t"
        "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t"
        "ds_read_b128 v[32:35], %[lds]\n\t"
        "s_nop 15\n\ts_nop 7\n\t"                       // XDL write -> VALU read wait states (16 passes: 18+)
        "v_mov_b32 v44, v0\n\tv_mov_b32 v45, v16\n\t"    // pair r = 0 (as in the kernel)
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_add_f32 v[44:45], v[44:45], v[32:33] op_sel_hi:[1,0]\n\t"
        "v_mov_b32 v16, v1\n\t"
        ".if %[sel]\n\t"
        "v_pk_add_f32 v[0:1], v[16:17], v[32:33] op_sel:[0,1]\n\t"
        ".else\n\t"
        "v_add_f32 v0, v16, v33\n\tv_add_f32 v1, v17, v33\n\t"
        ".endif\n\t"
        "v_mov_b32 %[lo], v0\n\tv_mov_b32 %[hi], v1\n\t"
        : [lo] "=v"(lo), [hi] "=v"(hi)
        : [lds] "v"(lds_addr), [nm] "s"(nm), [sel] "n"(SEL)
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v40", "v41", "v42", "v43",
          "v44", "v45", "s20", "scc", "memory");
    const float expect = 16.f * nm + bias[4 * (lane & 3) + 1];
    wrong_lo += __uint_as_float(lo) != expect;
    wrong_hi += __uint_as_float(hi) != expect;
  }
  if (wrong_lo) atomicAdd(bad + (lane >> 4), wrong_lo);
  if (wrong_hi) atomicAdd(bad + 4 + (lane >> 4), wrong_hi);
}

template <int SEL>
static void run(unsigned* bad, int waves, int nm) {
  CK(hipMemset(bad, 0, 64));
  hipLaunchKernelGGL((k<SEL>), dim3(256), dim3(waves * 64), 0, 0, bad, 20000, nm);
  CK(hipDeviceSynchronize());
  unsigned h[8];
  CK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
  printf("{\"form\": \"%s\", \"waves_per_workgroup\": %d, \"mfma_per_chain\": %d, \"wrong_lo_by_lane_quarter\": [%u, %u, %u, %u], "
         "\"wrong_hi_by_lane_quarter\": [%u, %u, %u, %u]}\n", SEL ? "v_pk_add_f32 op_sel:[0,1]" : "two v_add_f32", waves, nm, h[0], h[1], h[2], h[3],
         h[4], h[5], h[6], h[7]);
  fflush(stdout);
}

int main() {
  unsigned* bad;
  CK(hipMalloc(&bad, 64));
  for (int rep = 0; rep < 2; ++rep)
    for (int nm : {1, 3, 6, 18}) {
      run<1>(bad, 8, nm);
      run<1>(bad, 4, nm);
      run<0>(bad, 8, nm);
    }
  return 0;
}
