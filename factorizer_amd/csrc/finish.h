// finish.h — the fixed-order "finish" reductions behind every partial-sum kernel of the library (weight gradients, bias sums,
// LayerNorm affine gradients), as JOBS of one table-driven kernel (finish.hip: finish_batch_kernel).
//
// Every weight-gradient kernel leaves per-workgroup partial rows and a tiny second launch adds them in index order (no float
// atomics: bitwise reproducible).  Those launches do a microsecond of work but each one is a serial slot of the stream — a
// README training step had 54 of them, 5-10 us each, 0.33 ms of a 16 ms step (profiles/r05_p2_kernel_stats.md), most of them
// between the 10-40 us launches of the deep stages.  Their outputs are PARAMETER gradients: nothing in the backward reads them.
// So a caller that owns the step (training.FlatAdamW / parallel.FlatGradSync) may ask the library to DEFER them
// (fz_finish_defer): the job descriptors queue up on the host and fz_finish_flush runs all of them as one or two grids (a job
// of phase 1 reads outputs of phase-0 jobs).  Without deferral a job runs at once as a one-entry table through the same kernel,
// so the arithmetic — every sum, in the same order — is the same either way.
#pragma once
#include "fz_common.h"

namespace fz {

constexpr int kWgRow = 2048 + 2048 + 32 + 64 + 64;   // floats of one wpart row of gemm_chain_bwd_wg: dW2 [32][64] | S1 [64][32] | db2 | db1 | dγ | dβ
constexpr int kDwRow = 1024 + 32 + 64;               // floats of one wpart row of gemm_dw: dW [32][32] | db [32] | dγ [32] | dβ [32]

// out[grp][e] = Σ_{row in group grp} part[row][e]; rows in `gy` contiguous groups, 32 outputs per block (gx blocks per group)
struct FinRows {
  const float* part;
  float* out;
  int nrows, n, rows_per_group, gx;
};
// out[e] (+)= Σ_chunks part[chunk·stride + e], e < n
struct FinChunk {
  const float* part;
  float* out;
  long long n, stride;
  int nchunk, accumulate;
};
// one weight gradient from its partial blocks (+ bias sums, + LayerNorm affine fold): wgrad.hip
struct WgradFinishOne {
  const float* part;
  const float* part_bias;
  float* gw;
  float* gbias;
  const float* ln_g;
  const float* ln_b;
  int nchunk, M, K, accumulate, nbw, wide;   // wide: 0 = 8 elements per block, 1 = 256 (one thread each), 2 = 64 x 4 slices, 3 = 1 024 (four per thread)
};
// gemm_dw rows -> gw (ld ldgw), gb, (dγ | dβ)
struct FinDw {
  const float* wpart;
  const float* ln_g;
  const float* ln_b;
  float* gw;
  float* gb;
  float* gln;
  int rows, ldgw;
};
// gemm_chain_bwd_wg rows -> gw1, gb1, gw2 (ld ldw2), gb2, (dγ | dβ)
struct FinChainWg {
  const float* wpart;
  const float* ln_g;
  const float* ln_b;
  float* gw1;
  float* gb1;
  float* gw2;
  float* gb2;
  float* gln;
  int rows, ldw2;
};
// the decoder node's (deep x g) correlation -> dW_t, dW_b, db_t: upcat.hip
struct FinUpcat {
  const float* gt;
  const float* w_t;
  const float* w_b;
  const float* gb_ad;
  const float* b_t;
  float* gw_t;
  float* gw_b;
  float* gb_t;
  int ldb, ldg, Cd, O, M;
};

enum { FK_ROWS = 0, FK_CHUNK = 1, FK_WGRAD = 2, FK_DW = 3, FK_CHAIN_WG = 4, FK_UPCAT = 5 };

struct FinishJob {
  int kind, nblocks, phase, pad;
  union {
    FinRows rows;
    FinChunk chunk;
    WgradFinishOne wg;
    FinDw dw;
    FinChainWg cw;
    FinUpcat up;
  } u;
};

// Host side (finish.hip).  finish_run: now (one-entry table) or, while the calling process defers (fz_finish_defer), queued.
// A job that ACCUMULATES into its output is never queued (and drains the queue first: an earlier job may write what it adds to).
int finish_run(const FinishJob* jobs, int n, hipStream_t st);
inline FinishJob finish_job(int kind, int nblocks, int phase = 0) {
  FinishJob j;
  j.kind = kind; j.nblocks = nblocks; j.phase = phase; j.pad = 0;
  return j;
}
inline int finish_rows(const float* part, int nrows, int n, int rows_per_group, int groups, float* out, int phase, hipStream_t st) {
  FinishJob j = finish_job(FK_ROWS, ((n + 31) / 32) * groups, phase);
  j.u.rows = FinRows{part, out, nrows, n, rows_per_group, (n + 31) / 32};
  return finish_run(&j, 1, st);
}

#if defined(__HIPCC__)
// ---- device bodies: `vb` = the job's block index; every sum in the order of the launches these replaced ----

// 32 outputs per block, 8 strided partial sums per output
__device__ __forceinline__ void fin_rows_body(const FinRows& a, int vb, float (*red)[33]) {
  const int bx = vb % a.gx, by = vb / a.gx;
  const int el = threadIdx.x & 31, gq = threadIdx.x >> 5;
  const int e = bx * 32 + el;
  const int r0 = by * a.rows_per_group;
  const int r1 = min(a.nrows, r0 + a.rows_per_group);
  float s = 0.f;
  if (e < a.n)
    for (int ch = r0 + gq; ch < r1; ch += 8) s += a.part[(int64_t)ch * a.n + e];
  red[gq][el] = s;
  __syncthreads();
  if (gq == 0 && e < a.n)
    a.out[(int64_t)by * a.n + e] = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) +
                                   ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
}

// 32 strided partial sums per element (8 elements per 256-thread block), combined by a fixed tree through LDS
__device__ __forceinline__ void fin_chunk_body(const FinChunk& a, int vb, float (*red)[9]) {
  const int el = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int64_t e = (int64_t)vb * 8 + el;
  const int64_t n = a.n, ld = a.stride;
  const int nchunk = a.nchunk;
  const float* __restrict__ part = a.part;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n) {
    int ch = g;
    for (; ch + 96 < nchunk; ch += 128) {  // 4 independent loads in flight
      s0 += part[(int64_t)ch * ld + e];
      s1 += part[(int64_t)(ch + 32) * ld + e];
      s2 += part[(int64_t)(ch + 64) * ld + e];
      s3 += part[(int64_t)(ch + 96) * ld + e];
    }
    for (; ch < nchunk; ch += 32) s0 += part[(int64_t)ch * ld + e];
  }
  red[g][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && e < n) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red[i][el];
    a.out[e] = a.accumulate ? a.out[e] + t : t;
  }
}

// One weight gradient: blocks [0, nbw) reduce the M·K partial blocks (8 elements each, 32 strided partial sums per element,
// fixed tree), optionally folding the LayerNorm affine  gw[m][k] = Σ_ch (γ_k·part[ch][m][k] + β_k·pb[ch][m]);
// blocks [nbw, ..) reduce the M bias sums.
__device__ __forceinline__ void wgrad_finish_body(const float* __restrict__ part, const float* __restrict__ part_bias,
                                                  int nchunk, int M, int K, float* __restrict__ gw,
                                                  float* __restrict__ gbias, const float* __restrict__ ln_g,
                                                  const float* __restrict__ ln_b, int accumulate, int nbw,
                                                  float (*red)[9], const int bid) {
  const int el = threadIdx.x & 7, g = threadIdx.x >> 3;
  const bool bias_blk = bid >= nbw;
  const int64_t n = bias_blk ? (int64_t)M : (int64_t)M * K;
  const int64_t e = (int64_t)(bias_blk ? bid - nbw : bid) * 8 + el;
  const float* src = bias_blk ? part_bias : part;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n) {
    const bool fold = !bias_blk && ln_g != nullptr;
    const int m = fold ? (int)(e / K) : 0, k = fold ? (int)(e % K) : 0;
    const float gk = fold ? ln_g[k] : 1.f, bk = fold ? ln_b[k] : 0.f;
    int ch = g;
    if (fold) {
      for (; ch + 96 < nchunk; ch += 128) {
        s0 += gk * src[(int64_t)ch * n + e] + bk * part_bias[(int64_t)ch * M + m];
        s1 += gk * src[(int64_t)(ch + 32) * n + e] + bk * part_bias[(int64_t)(ch + 32) * M + m];
        s2 += gk * src[(int64_t)(ch + 64) * n + e] + bk * part_bias[(int64_t)(ch + 64) * M + m];
        s3 += gk * src[(int64_t)(ch + 96) * n + e] + bk * part_bias[(int64_t)(ch + 96) * M + m];
      }
      for (; ch < nchunk; ch += 32) s0 += gk * src[(int64_t)ch * n + e] + bk * part_bias[(int64_t)ch * M + m];
    } else {
      // 8 independent loads in flight: pure dependent-load latency (1024 partial blocks = 32 per thread at stage 0)
      for (; ch + 224 < nchunk; ch += 256) {
        const float a0 = src[(int64_t)ch * n + e], a1 = src[(int64_t)(ch + 32) * n + e];
        const float a2 = src[(int64_t)(ch + 64) * n + e], a3 = src[(int64_t)(ch + 96) * n + e];
        const float a4 = src[(int64_t)(ch + 128) * n + e], a5 = src[(int64_t)(ch + 160) * n + e];
        const float a6 = src[(int64_t)(ch + 192) * n + e], a7 = src[(int64_t)(ch + 224) * n + e];
        s0 += a0; s1 += a1; s2 += a2; s3 += a3;
        s0 += a4; s1 += a5; s2 += a6; s3 += a7;
      }
      for (; ch + 96 < nchunk; ch += 128) {
        s0 += src[(int64_t)ch * n + e];
        s1 += src[(int64_t)(ch + 32) * n + e];
        s2 += src[(int64_t)(ch + 64) * n + e];
        s3 += src[(int64_t)(ch + 96) * n + e];
      }
      for (; ch < nchunk; ch += 32) s0 += src[(int64_t)ch * n + e];
    }
  }
  red[g][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && e < n) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red[i][el];
    float* out = bias_blk ? gbias : gw;
    out[e] = accumulate ? out[e] + t : t;
  }
}

// Same result layout for FEW partial blocks of a LARGE weight (the deep stages: 512x2048 weights x 4 chunks): one thread per
// element walks the chunks in a fixed order with 256-byte coalesced reads.
__device__ __forceinline__ void wgrad_finish_wide_body(const float* __restrict__ part, const float* __restrict__ part_bias,
                                                       int nchunk, int M, int K, float* __restrict__ gw,
                                                       float* __restrict__ gbias, const float* __restrict__ ln_g,
                                                       const float* __restrict__ ln_b, int accumulate, int nbw,
                                                       const int bid) {
  const bool bias_blk = bid >= nbw;
  const int64_t n = bias_blk ? (int64_t)M : (int64_t)M * K;
  const int64_t e = (int64_t)(bias_blk ? bid - nbw : bid) * 256 + threadIdx.x;
  if (e >= n) return;
  const float* src = bias_blk ? part_bias : part;
  const bool fold = !bias_blk && ln_g != nullptr;
  const int m = fold ? (int)(e / K) : 0, k = fold ? (int)(e % K) : 0;
  const float gk = fold ? ln_g[k] : 1.f, bk = fold ? ln_b[k] : 0.f;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int ch = 0;
  for (; ch + 3 < nchunk; ch += 4) {
    float a0 = src[(int64_t)ch * n + e], a1 = src[(int64_t)(ch + 1) * n + e];
    float a2 = src[(int64_t)(ch + 2) * n + e], a3 = src[(int64_t)(ch + 3) * n + e];
    if (fold) {
      a0 = gk * a0 + bk * part_bias[(int64_t)ch * M + m];
      a1 = gk * a1 + bk * part_bias[(int64_t)(ch + 1) * M + m];
      a2 = gk * a2 + bk * part_bias[(int64_t)(ch + 2) * M + m];
      a3 = gk * a3 + bk * part_bias[(int64_t)(ch + 3) * M + m];
    }
    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
  }
  for (; ch < nchunk; ++ch) {
    float a0 = src[(int64_t)ch * n + e];
    if (fold) a0 = gk * a0 + bk * part_bias[(int64_t)ch * M + m];
    s0 += a0;
  }
  const float t = (s0 + s1) + (s2 + s3);
  float* out = bias_blk ? gbias : gw;
  out[e] = accumulate ? out[e] + t : t;
}

// The one-thread-per-element form with FOUR consecutive elements per thread (16-byte loads, 1 024 elements per block): the same
// sums in the same order as wgrad_finish_wide_body, a quarter of the blocks — the deep stages' weights (0.4-1.5 M elements, 4-8
// partial blocks) made two grids of 10-17 thousand near-empty blocks out of a deferred flush.  Needs M·K % 4 == 0, M % 4 == 0,
// K % 4 == 0 (the four elements then share their row m).
__device__ __forceinline__ void wgrad_finish_wide4_body(const float* __restrict__ part, const float* __restrict__ part_bias,
                                                        int nchunk, int M, int K, float* __restrict__ gw,
                                                        float* __restrict__ gbias, const float* __restrict__ ln_g,
                                                        const float* __restrict__ ln_b, int accumulate, int nbw,
                                                        const int bid) {
  const bool bias_blk = bid >= nbw;
  const int64_t n = bias_blk ? (int64_t)M : (int64_t)M * K;
  const int64_t e = ((int64_t)(bias_blk ? bid - nbw : bid) * 256 + threadIdx.x) * 4;
  if (e >= n) return;
  const float* src = bias_blk ? part_bias : part;
  const bool fold = !bias_blk && ln_g != nullptr;
  const int m = fold ? (int)(e / K) : 0, k = fold ? (int)(e % K) : 0;
  float4 gk = make_float4(1.f, 1.f, 1.f, 1.f), bk = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fold) {
    gk = *reinterpret_cast<const float4*>(ln_g + k);
    bk = *reinterpret_cast<const float4*>(ln_b + k);
  }
  auto ld = [&](int ch) {
    float4 a = *reinterpret_cast<const float4*>(src + (int64_t)ch * n + e);
    if (fold) {
      const float pb = part_bias[(int64_t)ch * M + m];
      a.x = gk.x * a.x + bk.x * pb; a.y = gk.y * a.y + bk.y * pb; a.z = gk.z * a.z + bk.z * pb; a.w = gk.w * a.w + bk.w * pb;
    }
    return a;
  };
  auto add = [](float4& s, const float4& a) { s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; };
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  int ch = 0;
  for (; ch + 3 < nchunk; ch += 4) {
    const float4 a0 = ld(ch), a1 = ld(ch + 1), a2 = ld(ch + 2), a3 = ld(ch + 3);
    add(s0, a0); add(s1, a1); add(s2, a2); add(s3, a3);
  }
  for (; ch < nchunk; ++ch) add(s0, ld(ch));
  float4 t = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                         (s0.w + s1.w) + (s2.w + s3.w));
  float* out = (bias_blk ? gbias : gw) + e;
  if (accumulate) {
    const float4 o = *reinterpret_cast<const float4*>(out);
    t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
  }
  *reinterpret_cast<float4*>(out) = t;
}

// The form between the two (65-256 partial blocks of a weight of >= 4 096 elements: the C = 64 and C = 128 stages): 64
// consecutive elements per block (256-byte coalesced rows), the chunks in 4 interleaved slices, 4 loads in flight per thread,
// the slices added in a fixed tree.  The 8-element form spends 8x the blocks on these problems and each block touches a quarter
// of every 128-byte line it reads: two grids of 15-21 thousand blocks took 110 us of a deferred flush.
__device__ __forceinline__ void wgrad_finish_mid_body(const float* __restrict__ part, const float* __restrict__ part_bias,
                                                      int nchunk, int M, int K, float* __restrict__ gw,
                                                      float* __restrict__ gbias, const float* __restrict__ ln_g,
                                                      const float* __restrict__ ln_b, int accumulate, int nbw,
                                                      float* red /* [4][64] */, const int bid) {
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const bool bias_blk = bid >= nbw;
  const int64_t n = bias_blk ? (int64_t)M : (int64_t)M * K;
  const int64_t e = (int64_t)(bias_blk ? bid - nbw : bid) * 64 + el;
  const float* src = bias_blk ? part_bias : part;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n) {
    const bool fold = !bias_blk && ln_g != nullptr;
    const int m = fold ? (int)(e / K) : 0, k = fold ? (int)(e % K) : 0;
    const float gk = fold ? ln_g[k] : 1.f, bk = fold ? ln_b[k] : 0.f;
    int ch = sl;
    for (; ch + 12 < nchunk; ch += 16) {
      float a0 = src[(int64_t)ch * n + e], a1 = src[(int64_t)(ch + 4) * n + e];
      float a2 = src[(int64_t)(ch + 8) * n + e], a3 = src[(int64_t)(ch + 12) * n + e];
      if (fold) {
        a0 = gk * a0 + bk * part_bias[(int64_t)ch * M + m];
        a1 = gk * a1 + bk * part_bias[(int64_t)(ch + 4) * M + m];
        a2 = gk * a2 + bk * part_bias[(int64_t)(ch + 8) * M + m];
        a3 = gk * a3 + bk * part_bias[(int64_t)(ch + 12) * M + m];
      }
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    for (; ch < nchunk; ch += 4) {
      float a0 = src[(int64_t)ch * n + e];
      if (fold) a0 = gk * a0 + bk * part_bias[(int64_t)ch * M + m];
      s0 += a0;
    }
  }
  red[sl * 64 + el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && e < n) {
    const float t = (red[el] + red[64 + el]) + (red[128 + el] + red[192 + el]);
    float* out = bias_blk ? gbias : gw;
    out[e] = accumulate ? out[e] + t : t;
  }
}

// gw[m][k] = (ln ? γ[k]·S[m][k] + β[k]·sg[m] : S[m][k]),  gb[m] = sg[m] (when wanted); rows added in slice order.
// kDwRow = 70 x 16: 256 threads = 16 elements x 16 row slices.
__device__ __forceinline__ void fin_dw_body(const FinDw& a, int vb, float (*s)[17], float* sm) {
  const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const float* wpart = a.wpart;
  const int rows = a.rows;
  auto total = [&](int e) {
    float t = 0.f;
    // 8 row loads in flight per thread (the adds stay in row order: same result, a fraction of the round trips)
    int r = sl;
    for (; r + 7 * 16 < rows; r += 8 * 16) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = wpart[(int64_t)(r + 16 * i) * kDwRow + e];
#pragma unroll
      for (int i = 0; i < 8; ++i) t += v[i];
    }
    for (; r < rows; r += 16) t += wpart[(int64_t)r * kDwRow + e];
    return t;
  };
  const int e = vb * 16 + el;
  s[sl][el] = total(e);
  __syncthreads();
  float v = 0.f;
  if (sl == 0)
    for (int q = 0; q < 16; ++q) v += s[q][el];
  const bool is_w = e < 1024;              // uniform per block
  if (is_w && a.ln_g != nullptr) {
    __syncthreads();
    const int m = (vb * 16) / 32;
    if (el == 0) s[sl][0] = total(1024 + m);
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int q = 0; q < 16; ++q) t += s[q][0];
      *sm = t;
    }
    __syncthreads();
  }
  if (sl != 0) return;
  if (is_w) {
    const int k = e & 31;
    a.gw[(e >> 5) * a.ldgw + k] = a.ln_g != nullptr ? a.ln_g[k] * v + a.ln_b[k] * *sm : v;
  } else if (e < 1024 + 32) {
    if (a.gb != nullptr) a.gb[e - 1024] = v;
  } else if (a.gln != nullptr) {
    a.gln[e - 1024 - 32] = v;
  }
}

// gw2 = Σ rows dW2;  gb2, gb1 likewise;  gw1[c][k] = γ[k]·Σ S1[c][k] + β[k]·gb1[c]   (z1 = W1·(γ x̂ + β) + b1).
// 256 threads = 16 elements x 16 row slices; slices walk the rows with stride 16 and are added in slice order.
// (hidden 128: one job per half with gw2 advanced by 64·half columns and ldw2 = 128, gw1 / gb1 by 64·half rows;
// gb2 from the first half, gln from the second — null where not wanted)
__device__ __forceinline__ void fin_chain_wg_body(const FinChainWg& a, int vb, float (*s)[17], float* sb1) {
  const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const float* wpart = a.wpart;
  const int rows = a.rows;
  auto total = [&](int e) {
    float t = 0.f;
    int r = sl;
    for (; r + 7 * 16 < rows; r += 8 * 16) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = wpart[(int64_t)(r + 16 * i) * kWgRow + e];
#pragma unroll
      for (int i = 0; i < 8; ++i) t += v[i];
    }
    for (; r < rows; r += 16) t += wpart[(int64_t)r * kWgRow + e];
    return t;
  };
  // every block also needs db1 of the rows it scales: blocks over S1 recompute the 16-row slice sums of their db1 entry
  const int e = vb * 16 + el;
  s[sl][el] = e < kWgRow ? total(e) : 0.f;
  __syncthreads();
  float v = 0.f;
  if (sl == 0) {
    for (int q = 0; q < 16; ++q) v += s[q][el];
  }
  const bool is_s1 = e >= 2048 && e < 4096;
  if (is_s1) {   // uniform per block: 16 consecutive elements of one S1 row c
    __syncthreads();
    const int cidx = (vb * 16 - 2048) / 32;
    if (el == 0) s[sl][0] = total(4096 + 32 + cidx);
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int q = 0; q < 16; ++q) t += s[q][0];
      sb1[0] = t;
    }
    __syncthreads();
  }
  if (sl != 0 || e >= kWgRow) return;
  if (e < 2048) a.gw2[(e >> 6) * a.ldw2 + (e & 63)] = v;
  else if (e < 4096) { const int k = (e - 2048) & 31; a.gw1[e - 2048] = a.ln_g[k] * v + a.ln_b[k] * sb1[0]; }
  else if (e < 4096 + 32) { if (a.gb2 != nullptr) a.gb2[e - 4096] = v; }
  else if (e < 4096 + 96) a.gb1[e - 4096 - 32] = v;
  else if (a.gln != nullptr) a.gln[e - 4096 - 96] = v;          // dγ (32) | dβ (32)
}

// With ncb = ceil(O/32) column blocks:  gw_t[k][c][t] = Σ_m w_b[m][c]·gt[k][m][t]  (blocks (k, cb): 32 columns x 8 taps, one
// output per thread);  gw_b[m][c] = Σ_{k,t} gt[k][m][t]·w_t[k][c][t] + gb_ad[m]·b_t[c]  (blocks (m, cb): the k range in 8 interleaved
// slices whose sums meet in a fixed tree; written with row stride ldg into the adapter's weight gradient);
// gb_t[c] = Σ_m gb_ad[m]·w_b[m][c] (last block).  (One block per k resp. per m, as rounds 3-4 had it, left a thread with up to
// 2 048 resp. 4 096 dependent products at the 512 -> 256 level: 30-60 us for a few kiloflops.)
__host__ __device__ inline int fin_upcat_blocks(int Cd, int O, int M) { return (Cd + M) * ((O + 31) / 32) + 1; }
__device__ __forceinline__ void fin_upcat_body(const FinUpcat& a, int blk, float (*red)[33]) {
  const int Cd = a.Cd, O = a.O, M = a.M, ldb = a.ldb;
  const int ncb = (O + 31) / 32;
  const float* __restrict__ gt = a.gt;
  const float* __restrict__ w_t = a.w_t;
  const float* __restrict__ w_b = a.w_b;
  if (blk < Cd * ncb) {
    const int k = blk / ncb, cb = blk % ncb;
    const float* gk = gt + (int64_t)k * M * 8;
    const int c = cb * 32 + (threadIdx.x >> 3), t = threadIdx.x & 7;
    if (c < O) {
      float acc = 0.f;
      for (int m = 0; m < M; ++m) acc += w_b[(int64_t)m * ldb + c] * gk[m * 8 + t];
      a.gw_t[((int64_t)k * O + c) * 8 + t] = acc;
    }
  } else if (blk < (Cd + M) * ncb) {
    const int m = (blk - Cd * ncb) / ncb, cb = (blk - Cd * ncb) % ncb;
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = cb * 32 + el;
    float acc = 0.f;
    if (c < O) {
      for (int k = sl; k < Cd; k += 8) {
        const float* g8 = gt + ((int64_t)k * M + m) * 8;
        const float* w8 = w_t + ((int64_t)k * O + c) * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc += g8[t] * w8[t];
      }
    }
    red[sl][el] = acc;
    __syncthreads();
    if (sl == 0 && c < O) {
      float v = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
      if (a.b_t != nullptr) v += a.gb_ad[m] * a.b_t[c];
      a.gw_b[(int64_t)m * a.ldg + c] = v;
    }
  } else if (a.gb_t != nullptr) {
    for (int c = threadIdx.x; c < O; c += 256) {
      float acc = 0.f;
      for (int m = 0; m < M; ++m) acc += a.gb_ad[m] * w_b[(int64_t)m * ldb + c];
      a.gb_t[c] = acc;
    }
  }
}
#endif  // __HIPCC__

}  // namespace fz
