// gemm_common.h — argument block, operand fetch and epilogue helpers shared by the channels-first GEMM
// kernels (gemm.hip: fp32 MFMA family; gemm_bx.hip: split-bf16 MFMA family).
#pragma once
#include <cstdlib>

#include "fz_common.h"

namespace fz {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { LOAD_PLAIN = 0, LOAD_S2D = 1, LOAD_K3 = 2 };
enum { EPI_PLAIN = 0, EPI_D2S = 1, EPI_LNBWD = 2 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

// AT = storage type of the activation tensors (float or bf16, fz_common.h); everything else is fp32
template <typename AT>
struct GemmArgsT {
  const AT* x[4];      // input sources
  int nsrc;            // number of sources
  int src_mode;        // 0: channel concat (x[0]: c0 channels, x[1]: Cin-c0)   1: average of sources
  int c0;
  int Cin;             // input channels
  int64_t Vin;         // voxels per sample of the input tensor(s)
  int Di, Hi, Wi;      // input D, H, W (LOAD_S2D: fine tensor; LOAD_K3: the grid)
  const float* w;      // weights
  int w_t, ldw;        // A[m][k] = w_t ? w[k*ldw + m] : w[m*ldw + k]
  int M, K;            // output rows, reduction length
  const float* bias;   // [M] (EPI_PLAIN) or [M/8] (EPI_D2S), may be null
  int ln;              // LayerNorm prologue over the Cin channels
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  float* stats_out;    // [B][2][Vin] (mean, rstd) or null
  int bact;            // activation applied to the B operand
  const AT* bmul;      // B operand *= act'(bmul) (same shape as the input) or null
  int bmul_kind;
  int eact;            // activation applied to the result
  const AT* res;       // residual added in the epilogue (same shape as y) or null
  const AT* emul;      // epilogue multiply by act'(emul) (same shape as y) or null
  int emul_kind;
  AT* y;
  int64_t Ncol;        // columns per sample (= Vin for LOAD_PLAIN, coarse voxels for LOAD_S2D)
  int Ho, Wo;          // coarse H, W (LOAD_S2D columns / EPI_D2S input grid)
  int B;
  int dbg;             // diagnostics (FZ_GEMM_DBG): 3 = skip the weight staging
  int tile_map;        // workgroup -> column-tile order: 0 linear, 1 XCD-contiguous, 2 scattered
  int tune;            // host only: fz_gemm_desc.tune (tile choice of the split-bf16 family for timing probes)
  int ygroups, xtiles; // streaming kernel: > 1 row-block groups -> 1-D XCD-aware grid of xtiles column tiles
  // EPI_LNBWD (M == 32): the result is gl = dL/d(LN output); the epilogue applies the LayerNorm
  // backward in registers: y = rstd*(gl*g - mean_c(gl*g) - n*mean_c(gl*g*n)) + lnb_gadd
  const AT* lnb_x;         // (B, 32, V) LayerNorm input
  const float* lnb_stats;  // (B, 2, V) mean, rstd
  const float* lnb_g;      // (32) gamma
  const AT* lnb_gadd;      // (B, 32, V) gradient added to the result, or null
  float* lnb_part;         // [gridDim.x][64] per-workgroup partial (gγ | gβ) sums
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// gelu of the lane's two voxels at once on packed fp32 (v_pk_mul_f32 / v_pk_fma_f32 — plain forms, no op_sel: the two values
// are one 64-bit register pair; the exponentials and reciprocals stay scalar): the same operations in the same order as
// gelu_f (results equal up to the compiler's contraction choices); ~11 instead of ~19 full-rate VALU instructions per value in the GELU phase of the
// chained MLP kernels, whose VALU time equals their MFMA time (3328-line tile body: 1 700 VALU, 128 MFMA instructions).
typedef float fz_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu2_f(const float (&x)[2], float (&g)[2]) {
  const fz_f32x2 xv = {x[0], x[1]};
  const fz_f32x2 xs = xv * 0.70710678118654752f;                       // fast_erf's argument
  const fz_f32x2 ax = {fabsf(xs[0]), fabsf(xs[1])};
  const fz_f32x2 den = ax * 0.3275911f + 1.0f;
  const fz_f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const fz_f32x2 poly = t * (t * (t * (t * (t * 1.061405429f + -1.453152027f) + 1.421413741f) + -0.284496736f) + 0.254829592f);
  const fz_f32x2 nax2 = -ax * ax;
  const fz_f32x2 E = {__expf(nax2[0]), __expf(nax2[1])};
  const fz_f32x2 r = 1.0f - poly * E;
  const fz_f32x2 rs = {__builtin_copysignf(r[0], xs[0]), __builtin_copysignf(r[1], xs[1])};   // one v_bfi_b32 each (r >= -1 ulp)
  const fz_f32x2 gv = (xv * 0.5f) * (rs + 1.0f);
  g[0] = gv[0]; g[1] = gv[1];
}

__device__ __forceinline__ float act_f(int kind, float v) {
  if (kind == ACT_RELU) return v > 0.f ? v : 0.f;
  if (kind == ACT_GELU) return gelu_f(v);
  return v;
}
__device__ __forceinline__ float act_grad_f(int kind, float v) {
  if (kind == ACT_RELU) return v > 0.f ? 1.f : 0.f;
  if (kind == ACT_GELU) return gelu_grad_f(v);
  return 1.f;
}

// k index of A-operand step `a` for lane half h
template <int LOADER>
__device__ __forceinline__ int a_k(int a, int h) {
  if (LOADER == LOAD_S2D) return (2 * (a >> 3) + h) * 8 + (a & 7);
  if (LOADER == LOAD_K3) return (2 * (a / 27) + h) * 27 + (a % 27);
  return 2 * a + h;
}

template <typename AT>
__device__ __forceinline__ float weight_at(const GemmArgsT<AT>& p, int m, int k) {
  return p.w_t ? p.w[(int64_t)k * p.ldw + m] : p.w[(int64_t)m * p.ldw + k];
}

// ---- NACC-wide vector access (NACC = 4, 2, 1 consecutive voxels per lane) -------------------
template <int NACC, typename T>
__device__ __forceinline__ void vload(const T* p, float (&v)[NACC]) { aload<NACC>(p, v); }
template <int NACC, typename T>
__device__ __forceinline__ void vstore(T* p, const float (&v)[NACC]) { astore<NACC>(p, v); }

// ---- plain-loader fetch of NACC voxels of channel c -------------------------------------------
// BRANCH-FREE on purpose: hipcc wraps a conditional load in s_cbranch + s_waitcnt vmcnt(0), which
// serialises every load of the prefetch ring.  Out-of-range lanes/channels read a clamped (valid)
// address and are zeroed with a select; the two-source concat picks its pointer with a select.
template <int NACC, bool BMUL, typename AT>
__device__ __forceinline__ void fetch_plain(const GemmArgsT<AT>& p, int b, int c, int64_t off, bool ok,
                                            float (&v)[NACC]) {
  const bool cok = ok && c < p.Cin;
  const int cc = c < p.Cin ? c : p.Cin - 1;
  const bool first = cc < p.c0;
  const AT* base = first ? p.x[0] : p.x[1];
  const int cs = first ? p.c0 : p.Cin - p.c0;
  const int ci = first ? cc : cc - p.c0;
  const int64_t o = ((int64_t)b * cs + ci) * p.Vin + (ok ? off : 0);
  vload<NACC>(base + o, v);
  if (BMUL) {  // ReLU gate of the consumer's forward output (bmul_kind == ACT_RELU)
    float e[NACC];
    vload<NACC>(p.bmul + ((int64_t)b * p.Cin + cc) * p.Vin + (ok ? off : 0), e);
#pragma unroll
    for (int i = 0; i < NACC; ++i) v[i] = e[i] > 0.f ? v[i] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < NACC; ++i) v[i] = cok ? v[i] : 0.f;
}

// Streaming-kernel variant: NO select on the loaded values (a select right behind the load makes the
// scheduler wait for the load it just issued).  The consumer masks / gates when the ring slot is
// used, PF steps later.  With BMUL the gate operand rides in the upper half of the slot.
template <int NL, bool BMUL, typename AT>
__device__ __forceinline__ void fetch_plain_raw(const GemmArgsT<AT>& p, int b, int c, int64_t off, bool ok,
                                                float (&v)[BMUL ? 2 * NL : NL]) {
  const int cc = c < p.Cin ? c : p.Cin - 1;
  const bool first = cc < p.c0;
  const AT* base = first ? p.x[0] : p.x[1];
  const int cs = first ? p.c0 : p.Cin - p.c0;
  const int ci = first ? cc : cc - p.c0;
  const int64_t oo = ok ? off : 0;
  float a[NL];
  vload<NL>(base + ((int64_t)b * cs + ci) * p.Vin + oo, a);
#pragma unroll
  for (int i = 0; i < NL; ++i) v[i] = a[i];
  if (BMUL) {
    float e[NL];
    vload<NL>(p.bmul + ((int64_t)b * p.Cin + cc) * p.Vin + oo, e);
#pragma unroll
    for (int i = 0; i < NL; ++i) v[NL + i] = e[i];
  }
}

// ---- epilogue for one 32-row block -------------------------------------------------------------
// acc[q][r]: row (r&3)+8(r>>2)+4h of the block, column group q.  tw = per-row additive constant
// (LayerNorm β·W term), or null.
template <int NACC, int EPI, bool S2DCOLS, typename AT>
__device__ __forceinline__ void store_block(const GemmArgsT<AT>& p, const f32x16 (&acc)[NACC], int b, int mrow0,
                                            int64_t ncol, int h, const float* tWblk) {
  if (EPI == EPI_PLAIN) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int m = mrow0 + rl;
      if (m >= p.M) continue;
      float v[NACC];
      float add = p.bias ? p.bias[m] : 0.f;
      if (tWblk != nullptr) add += tWblk[rl];
#pragma unroll
      for (int q = 0; q < NACC; ++q) v[q] = acc[q][r] + add;
      if (p.eact) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) v[q] = act_f(p.eact, v[q]);
      }
      const int64_t o = ((int64_t)b * p.M + m) * p.Ncol + ncol;
      if (p.emul) {
        float e[NACC];
        vload<NACC>(p.emul + o, e);
#pragma unroll
        for (int q = 0; q < NACC; ++q) v[q] *= act_grad_f(p.emul_kind, e[q]);
      }
      if (p.res) {
        float e[NACC];
        vload<NACC>(p.res + o, e);
#pragma unroll
        for (int q = 0; q < NACC; ++q) v[q] += e[q];
      }
      vstore<NACC>(p.y + o, v);
    }
  } else {
    // rows are (o, tap): m = o*8 + td*4 + th*2 + tw ; inside a 32-row block td = h,
    // th = (r>>1)&1, tw = r&1, o_local = r>>2.  Columns ncol..ncol+3 are coarse voxels; the
    // fine tensor gets 8 consecutive voxels (tw pairs) per (o, td, th).
    const int Wf = 2 * p.Wo, Hf = 2 * p.Ho;
    const int64_t Vf = 8 * p.Ncol;
    const int Mo = p.M >> 3;
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) {
      const int r0 = 2 * rp;
      const int o = (mrow0 >> 3) + (r0 >> 2);
      if (o >= Mo) continue;
      const int th = (r0 >> 1) & 1, td = h;
      const float bs = p.bias ? p.bias[o] : 0.f;
      const int64_t obase = ((int64_t)b * Mo + o) * Vf;
      AT* ybase = p.y + obase;
      // p.res: a fine-resolution tensor added to the scattered block (the skip-connection gradient
      // joining the down-convolution's input gradient, unet.py:95-99 / autograd's accumulation)
      if (NACC == 4 && (p.Wo & 3) == 0) {
        const int wo = (int)(ncol % p.Wo);
        const int64_t t2 = ncol / p.Wo;
        const int ho = (int)(t2 % p.Ho);
        const int dz = (int)(t2 / p.Ho);
        const int64_t fo = ((int64_t)(2 * dz + td) * Hf + (2 * ho + th)) * Wf + 2 * wo;
        float o8[8] = {acc[0][r0] + bs, acc[0][r0 + 1] + bs, acc[1 % NACC][r0] + bs, acc[1 % NACC][r0 + 1] + bs,
                       acc[2 % NACC][r0] + bs, acc[2 % NACC][r0 + 1] + bs, acc[3 % NACC][r0] + bs, acc[3 % NACC][r0 + 1] + bs};
        if (p.res) {
          float r8[8];
          vload<8>(p.res + obase + fo, r8);
#pragma unroll
          for (int i = 0; i < 8; ++i) o8[i] += r8[i];
        }
        vstore<8>(ybase + fo, o8);
      } else {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
          const int64_t nq = ncol + q;
          const int wo = (int)(nq % p.Wo);
          const int64_t t2 = nq / p.Wo;
          const int ho = (int)(t2 % p.Ho);
          const int dz = (int)(t2 / p.Ho);
          const int64_t fo = ((int64_t)(2 * dz + td) * Hf + (2 * ho + th)) * Wf + 2 * wo;
          float v2[2] = {acc[q][r0] + bs, acc[q][r0 + 1] + bs};
          if (p.res) {
            float r2[2];
            vload<2>(p.res + obase + fo, r2);
            v2[0] += r2[0]; v2[1] += r2[1];
          }
          vstore<2>(ybase + fo, v2);
        }
      }
    }
  }
}

// split-bf16 MFMA family (gemm_bx.hip); pro: 0 none, 1 LayerNorm, 2 GELU, 3 ReLU gate (bmul)
template <typename AT>
int gemm_bx_launch(const GemmArgsT<AT>& a, int loader, int epilogue, int pro, fz_stream_t stream);

}  // namespace fz
