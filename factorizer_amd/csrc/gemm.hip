// gemm.hip — channels-first GEMM family on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   Out[m, n] = epilogue( Σ_k A[m, k] · prologue(In)[k, n] )        n = voxel (column)
//
// One kernel template serves every dense layer of the Factorizer U-shape and their input
// gradients (reference call sites, paths relative to the reference root):
//   * Linear / in_proj / out_proj / adapter / MLP / head : Conv1d(k=1) on flatten(2)
//       layers/linear.py:53-58, factorizer.py:38,53,116, layers/mlp.py:54-60, unet.py:253
//   * downsample Conv3d(k=2, stride=2)        unet.py:53   (space-to-depth loader)
//   * upsample  ConvTranspose3d(k=2, stride=2) unet.py:123 (depth-to-space epilogue)
//   * LayerNorm over channels fused as a prologue (layers/norm.py:29-34), ReLU/GELU, bias,
//     residual add (factorizer.py:75-76) and window averaging (operations.py:426-433) fused
//     as prologue/epilogue so each full-resolution tensor is read/written once per layer.
//
// MFMA mapping (wave64, 32x32x2 f32): lane l = (j = l&31, h = l>>5).  The B operand of K-step
// s is In[k = 2s+h][column j]; each lane loads ONE 16-byte vector = 4 consecutive voxels of
// channel 2s+h, and the 4 components feed 4 MFMAs (column groups q = 0..3, voxel 4j+q), so
// global loads are 1 KiB-per-wave coalesced and no LDS staging of activations is needed.
// Weights (the A operand) are staged once per workgroup in LDS in operand order.
// The accumulator tile has its row in (register, h) and its column in j, so the epilogue
// stores 16-byte vectors (4 voxels) per register: 512 B contiguous per output row.
#include "gemm_bx.h"   // split-bf16 operand helpers for the fused kernels (brings gemm_common.h)
#include "finish.h"    // kWgRow / kDwRow and the jobs that add the weight-gradient rows

namespace fz {

// ---- LayerNorm-backward epilogue (M == 32, one row block) ------------------------------------------
// acc[q][r] = gl[row (r,h)][voxel 4j+q].  Everything stays in registers: the channel means are
// sums over the 16 registers + the other lane half; the affine gradients are reduced over the
// 32 lanes of each half on the DPP network, then over the 4 waves through LDS.
__device__ __forceinline__ float half_sum32(float v) {
  v += dpp_take<0xB1, 0xf>(v);   // xor 1
  v += dpp_take<0x4E, 0xf>(v);   // xor 2
  v += dpp_take<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_take<0x140, 0xf>(v);  // row_mirror  -> 16-lane row totals in every lane
  v += dpp_take<0x142, 0xa>(v);  // row_bcast15: rows 1,3 += rows 0,2  -> lanes 16-31 / 48-63 hold the half totals
  return v;
}

template <int NACC, bool GADD, bool GADD_LDS = false, typename AT = float>
__device__ __forceinline__ void lnbwd_block(const GemmArgsT<AT>& p, const f32x16 (&acc)[NACC], int b, int64_t ncol,
                                            bool col_ok, int lane, int wave, float* red /* [4][64] */,
                                            int64_t part_row, const float* g_lds /* gamma[32] in LDS */,
                                            const float* gadd_lds = nullptr /* [32][32*NACC] tile of lnb_gadd */) {
  const int h = lane >> 5;
  const int64_t nc = col_ok ? ncol : 0;
  // row = rbase(r) + 4h: the row part of every address is wave-uniform (scalar base) and ONE
  // 32-bit per-lane offset serves the 16 rows (global_load saddr + voffset; the host bounds Ncol so
  // that 20*Ncol bytes fit) — per-row 64-bit lane addresses cost a VGPR pair per load in flight
  const unsigned lane_off = (unsigned)(4 * h) * (unsigned)p.Ncol + (unsigned)nc;
  const int64_t sample = (int64_t)b * 32 * p.Ncol;
  const float* sp = p.lnb_stats + (int64_t)b * 2 * p.Ncol;
  float mu[NACC], rs[NACC];
  vload<NACC>(sp + nc, mu);
  vload<NACC>(sp + p.Ncol + nc, rs);
  __builtin_amdgcn_sched_barrier(0);  // do not hoist the x loads above the MFMA loop (operand regs still live)
  float xs[16][NACC];  // LayerNorm input rows of this lane (the only big live array besides acc)
  float m1[NACC], m2[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q) m1[q] = m2[q] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rbase = (r & 3) + 8 * (r >> 2);
    vload<NACC>(p.lnb_x + sample + (int64_t)rbase * p.Ncol + lane_off, xs[r]);
  }
  // the added gradient is fetched here, unconditionally and all rows at once: a load behind a
  // runtime `if` inside the row loop compiles to load → s_waitcnt vmcnt(0) per row (16 exposed
  // round trips per tile)
  // (with GADD_LDS the tile is already in LDS: read per row below, no registers held)
  float ga[(GADD && !GADD_LDS) ? 16 : 1][NACC];
  if (GADD && !GADD_LDS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rbase = (r & 3) + 8 * (r >> 2);
      vload<NACC>(p.lnb_gadd + sample + (int64_t)rbase * p.Ncol + lane_off, ga[(GADD && !GADD_LDS) ? r : 0]);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    const float gc = g_lds[row];  // (a global load here is a dependent L2 round trip per row)
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
      const float av = acc[q][r] * gc;
      xs[r][q] = (xs[r][q] - mu[q]) * rs[q];  // normalised input, reused below
      m1[q] += av;
      m2[q] += av * xs[r][q];
    }
    __builtin_amdgcn_sched_barrier(0);  // keep rows from interleaving (register pressure)
  }
#pragma unroll
  for (int q = 0; q < NACC; ++q) {
    m1[q] = (m1[q] + __shfl_xor(m1[q], 32, 64)) * (1.0f / 32.0f);
    m2[q] = (m2[q] + __shfl_xor(m2[q], 32, 64)) * (1.0f / 32.0f);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rbase = (r & 3) + 8 * (r >> 2);
    const int row = rbase + 4 * h;
    const int64_t so = sample + (int64_t)rbase * p.Ncol;  // uniform
    const float gc = g_lds[row];  // (a global load here is a dependent L2 round trip per row)
    float v[NACC], nhr[NACC], gl[NACC];
    if (GADD && GADD_LDS) vload<NACC>(gadd_lds + row * (32 * NACC) + NACC * (lane & 31), gl);
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
      nhr[q] = xs[r][q];
      v[q] = rs[q] * (acc[q][r] * gc - m1[q] - nhr[q] * m2[q]);
    }
    if (GADD) {
#pragma unroll
      for (int q = 0; q < NACC; ++q) v[q] += GADD_LDS ? gl[q] : ga[(GADD && !GADD_LDS) ? r : 0][q];
    }
    if (col_ok) vstore<NACC>(p.y + so + lane_off, v);
    // affine-gradient partials of this row over the wave's 32*NACC voxels
    float sg = 0.f, sb = 0.f;
    if (col_ok) {
#pragma unroll
      for (int q = 0; q < NACC; ++q) {
        sg += acc[q][r] * nhr[q];
        sb += acc[q][r];
      }
    }
    sg = half_sum32(sg);
    sb = half_sum32(sb);
    if ((lane & 31) == 31) {
      red[wave * 64 + row] = sg;
      red[wave * 64 + 32 + row] = sb;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int e = threadIdx.x;
    p.lnb_part[part_row * 64 + e] = (red[e] + red[64 + e]) + (red[128 + e] + red[192 + e]);
  }
}

// =================================================================================================
// Kernel A — register-resident operand, K <= 2*NSTEP (the HBM-bound layers, C <= 64).
// Every lane issues ALL its operand loads up front (NSTEP x 16 B in flight per lane), applies the
// prologue in registers (exact two-pass LayerNorm), then walks the RB row blocks of the
// workgroup sequentially with one accumulator set: the input is read from HBM exactly once for
// all output rows and 10+ KiB per wave are in flight.
// =================================================================================================
// RESPF (one 32-row block, plain epilogue): the residual tile is requested with the operand, before the products — in the
// epilogue its loads were a second exposed round trip per workgroup (wait share 0.72 of the wave lifetime, round-3 profile 8)
template <int NSTEP, int EPI, bool BMUL, bool GADD = false, typename AT = float, int PF = 3, bool RESPF = false>
// (waves per SIMD chosen so that NO variant needs scratch: see the note on scratch and concurrent streams in nmf_pcf.hip)
__global__ __launch_bounds__(256, ((NSTEP <= 16 && EPI == EPI_PLAIN && !BMUL && !(PF & 2) && !RESPF) ? 3 : (NSTEP <= 16 ? 2 : 1))) void gemm_resident_kernel(GemmArgsT<AT> p, int RB) {
  // PF: prologue code compiled in — bit 0 input activation, bit 1 LayerNorm.  A runtime branch alone keeps a second copy
  // of the operand registers alive (normalised / activated next to raw): the K <= 32 form spilled because of it.
  constexpr bool ACTIN = (PF & 1) != 0, LNP = (PF & 2) != 0;
  extern __shared__ __attribute__((aligned(16))) float lds_a[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int nA = (p.K + 1) / 2;
  float* As = lds_a;                      // [nA][RB][64]
  float* tW = lds_a + nA * RB * 64;       // [32*RB]
  const int tiles_per_sample = (int)((p.Ncol + 511) / 512);
  int bid = blockIdx.x;
  if (p.tile_map == 1) {         // XCD-contiguous: each XCD walks its own eighth of the tiles
    const int nb = gridDim.x, q8 = nb / 8, r8 = nb % 8, xcd = bid % 8, i8 = bid / 8;
    if (nb >= 16) bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + i8;
  } else if (p.tile_map == 2) {  // scattered: odd-multiplier permutation of a power-of-two grid
    const int nb = gridDim.x;
    if ((nb & (nb - 1)) == 0) bid = (int)(((unsigned)bid * 40503u) & (unsigned)(nb - 1));
  }
  const int b = bid / tiles_per_sample;
  const int64_t n0 = ((int64_t)(bid % tiles_per_sample) * 4 + wave) * 128;
  const int m0 = blockIdx.y * 32 * RB;

  // the column operand (and the residual tile) first: the weight fill below runs under their latency
  const int64_t col_off = n0 + 4 * j;
  const bool col_ok = col_off < p.Ncol;
  // (K <= 32 forms; the K <= 64 ones keep the loads behind the fill: one of them needed scratch with the operand live across it)
  constexpr bool kEarly = NSTEP <= 16;
  float bv[NSTEP][4];
  float rv[RESPF ? 16 : 1][4];
  auto load_operands = [&]() {
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) fetch_plain<4, BMUL>(p, b, 2 * s + h, col_off, col_ok, bv[s]);
    if (RESPF) {   // host-checked: M == 32, p.res != null
#pragma unroll
      for (int r = 0; r < 16; ++r)
        vload<4>(p.res + ((int64_t)b * p.M + (r & 3) + 8 * (r >> 2) + 4 * h) * p.Ncol + (col_ok ? col_off : 0), rv[RESPF ? r : 0]);
    }
  };
  if (kEarly) load_operands();

  // batched fill (8 independent loads per thread before the LDS stores)
  constexpr int kFill = (NSTEP <= 16 && EPI == EPI_PLAIN && !BMUL) ? 4 : 8;   // independent loads per thread and round
  for (int base = threadIdx.x; base < nA * RB * 64; base += blockDim.x * kFill) {
    float tmp[kFill];
#pragma unroll
    for (int uu = 0; uu < kFill; ++uu) {
      const int idx = base + uu * blockDim.x;
      const bool in = idx < nA * RB * 64;
      const int ii = in ? idx : 0;
      const int l = ii & 63;
      const int rb = (ii >> 6) % RB;
      const int a = ii / (64 * RB);
      const int m = m0 + rb * 32 + (l & 31);
      const int k = 2 * a + (l >> 5);
      const bool ok = in && m < p.M && k < p.K;
      const int mc = m < p.M ? m : p.M - 1, kc = k < p.K ? k : p.K - 1;
      float wv = weight_at(p, mc, kc);
      if (LNP && p.ln) wv *= p.ln_g[kc];
      tmp[uu] = ok ? wv : 0.f;
    }
#pragma unroll
    for (int uu = 0; uu < kFill; ++uu) {
      const int idx = base + uu * blockDim.x;
      if (idx < nA * RB * 64) As[idx] = tmp[uu];
    }
  }
  if (LNP && p.ln) {
    for (int r = threadIdx.x; r < 32 * RB; r += blockDim.x) {
      const int m = m0 + r;
      float t = 0.f;
      if (m < p.M)
        for (int k = 0; k < p.K; ++k) t += weight_at(p, m, k) * p.ln_b[k];
      tW[r] = t;
    }
  }

  if (EPI == EPI_LNBWD) {  // gamma of the fused LayerNorm backward (tW is free: no LN prologue here)
    if (threadIdx.x < 32) tW[threadIdx.x] = p.lnb_g[threadIdx.x];
  }

  if (!kEarly) load_operands();

  if (LNP && p.ln) {
    // exact two-pass statistics over the Cin channels (this lane holds the parity-h half)
    float mu[4], rs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = 0.f;
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) t += bv[s][e];
      t += __shfl_xor(t, 32, 64);
      mu[e] = t / (float)p.Cin;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = 0.f;
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const float d = (2 * s + h < p.Cin) ? bv[s][e] - mu[e] : 0.f;
        t += d * d;
      }
      t += __shfl_xor(t, 32, 64);
      rs[e] = 1.0f / sqrtf(t / (float)p.Cin + p.ln_eps);
    }
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[s][e] = (2 * s + h < p.Cin) ? (bv[s][e] - mu[e]) * rs[e] : 0.f;
    if (p.stats_out != nullptr && blockIdx.y == 0 && h == 0 && col_ok) {
      float* so = p.stats_out + (int64_t)b * 2 * p.Vin;
      *reinterpret_cast<float4*>(so + col_off) = make_float4(mu[0], mu[1], mu[2], mu[3]);
      *reinterpret_cast<float4*>(so + p.Vin + col_off) = make_float4(rs[0], rs[1], rs[2], rs[3]);
    }
  }
  // (ACTIN = false: no input-activation code at all — the runtime branch alone kept a second copy of the 64 operand
  // registers alive and spilled the K <= 32 form)
  if (ACTIN && p.bact == ACT_GELU) {
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[s][e] = gelu_f(bv[s][e]);
  } else if (ACTIN && p.bact == ACT_RELU) {
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[s][e] = bv[s][e] > 0.f ? bv[s][e] : 0.f;
  }
  __syncthreads();

  // NOTE: every lane must stay active through the MFMAs (the A operand lives in all 64 lanes);
  // lanes whose columns fall outside the tensor only skip the stores.
  // EPI_LNBWD is host-checked to M == 32 and K == 2*NSTEP: one row block, no tail guards — this
  // keeps the register allocation of the (register-heavy) fused epilogue free of dead paths
  constexpr bool kExact = (EPI == EPI_LNBWD);
  const int RBn = kExact ? 1 : RB;
  for (int rb = 0; rb < RBn; ++rb) {
    if (!kExact && m0 + rb * 32 >= p.M) break;
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (kExact || s < nA) {
        const float av = As[(s * RBn + rb) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[s][q], acc[q], 0, 0, 0);
      }
      // (at most 4 weight operands requested ahead: all NSTEP of them in flight cost the registers that put the K <= 32
      // form one over the 168 of three waves per SIMD — it spilled 3 of them to scratch)
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if (EPI == EPI_LNBWD) {
      lnbwd_block<4, GADD>(p, acc, b, col_off, col_ok, lane, wave, tW + 32 * RB, blockIdx.x, tW);
    } else if (RESPF) {   // store_block's plain form with the residual already in registers (same order of additions)
      if (col_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
          float add = p.bias ? p.bias[rl] : 0.f;
          if (LNP && p.ln) add += tW[rl];
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = acc[q][r] + add;
          if (p.eact) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = act_f(p.eact, v[q]);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += rv[RESPF ? r : 0][q];
          vstore<4>(p.y + ((int64_t)b * p.M + rl) * p.Ncol + col_off, v);
        }
      }
    } else {
      if (col_ok) store_block<4, (EPI == EPI_LNBWD ? EPI_PLAIN : EPI), false>(p, acc, b, m0 + rb * 32, col_off, h, (LNP && p.ln) ? tW + rb * 32 : nullptr);
    }
  }
}

// =================================================================================================
// Kernel A' — the 32 -> 32 layers of stage 0 without a residual (LayerNorm + Linear in-projection, plain projections):
// PERSISTENT waves.  The weights are this lane's 16 A operands for the whole walk (registers, no LDS image, no barrier in
// the loop), every wave walks 128-column tiles with the operand of its next TWO tiles in flight — the one-tile-per-workgroup
// forms (Kernel A, the streaming ring) pay the load round trip of every tile in the open (wait share 0.4-0.7, profile 8).
// Arithmetic as Kernel A: exact two-pass LayerNorm, K-steps in ascending order on v_mfma_f32_32x32x2_f32.
// Host-checked: M <= 32, K == Cin == 32, plain loader and epilogue, no gate, no residual, Ncol % 4 == 0.
// =================================================================================================
// (The split-bf16 form of this kernel — 48 bf16 MFMAs + operand splits instead of 64 fp32 MFMAs, DESIGN §10.4a — needs ~20 registers
// more than the 240 of two operand tiles in flight + 64 accumulators leave at two waves per SIMD: 10-16 spilled; computing and storing the
// tile two column groups at a time — 32 accumulators, 8-byte stores, split weights in LDS — fits and is SLOWER: ln_linear_32->32 0.51 -> 0.84 ms per
// step fp32, 0.57 -> 0.58 bf16.  Not built in.)
template <int PF, typename AT>
__global__ __launch_bounds__(256, 2) void gemm_p32_kernel(GemmArgsT<AT> p, unsigned ntiles) {
  constexpr bool ACTIN = (PF & 1) != 0, LNP = (PF & 2) != 0;
  __shared__ float tW[32];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 31, h = lane >> 5;
  const unsigned tps = (unsigned)((p.Ncol + 127) / 128);

  // (M <= 32: rows beyond M have zero weights and are not stored — the 32 -> 3 head)
  if (threadIdx.x < 32) {
    float t = 0.f;
    if ((int)threadIdx.x < p.M) {
      t = p.bias ? p.bias[threadIdx.x] : 0.f;
      if (LNP)
        for (int k = 0; k < 32; ++k) t += weight_at(p, (int)threadIdx.x, k) * p.ln_b[k];
    }
    tW[threadIdx.x] = t;
  }
  float aw[16];   // A operand of K-step s: W[row j][channel 2s + h] (x gamma)
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    aw[s] = j < p.M ? weight_at(p, j, 2 * s + h) : 0.f;
    if (LNP) aw[s] *= p.ln_g[2 * s + h];
  }
  __syncthreads();
  float add[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) add[r] = tW[(r & 3) + 8 * (r >> 2) + 4 * h];

  typedef float BvT[16][4];
  auto fetch = [&](unsigned t, BvT& bv) {
    // every address = wave-uniform base (sample, channel pair: SGPRs) + ONE 32-bit lane offset (channel parity h, column)
    const unsigned b = t / tps;
    const int64_t col = (int64_t)(t - b * tps) * 128 + 4 * j;
    const unsigned xoff = (unsigned)(((int64_t)h * p.Vin + (col < p.Ncol ? col : 0)) * (int64_t)sizeof(AT));
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int c = 2 * s;   // (c0 is even: both channels of the pair come from the same source)
      const bool first = c < p.c0;
      const AT* base = first ? p.x[0] : p.x[1];
      const int cs = first ? p.c0 : 32 - p.c0;
      const int ci = first ? c : c - p.c0;
      uload<4>(base + ((int64_t)b * cs + ci) * p.Vin, xoff, bv[s]);
    }
  };
  const unsigned tstep = gridDim.x * 4;
  auto run = [&](unsigned t, BvT& bv) {
    const unsigned b = t / tps;
    const int64_t col_off = (int64_t)(t - b * tps) * 128 + 4 * j;
    const bool col_ok = col_off < p.Ncol;
    if (LNP) {
      float mu[4], rs[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) v += bv[s][e];
        v += __shfl_xor(v, 32, 64);
        mu[e] = v / 32.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const float d = bv[s][e] - mu[e];
          v += d * d;
        }
        v += __shfl_xor(v, 32, 64);
        rs[e] = 1.0f / sqrtf(v / 32.f + p.ln_eps);
      }
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[s][e] = (bv[s][e] - mu[e]) * rs[e];
      if (p.stats_out != nullptr && h == 0 && col_ok) {
        float* so = p.stats_out + (int64_t)b * 2 * p.Vin;
        *reinterpret_cast<float4*>(so + col_off) = make_float4(mu[0], mu[1], mu[2], mu[3]);
        *reinterpret_cast<float4*>(so + p.Vin + col_off) = make_float4(rs[0], rs[1], rs[2], rs[3]);
      }
    }
    if (ACTIN && p.bact == ACT_GELU) {
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[s][e] = gelu_f(bv[s][e]);
    } else if (ACTIN && p.bact == ACT_RELU) {
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[s][e] = bv[s][e] > 0.f ? bv[s][e] : 0.f;
    }
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[s], bv[s][q], acc[q], 0, 0, 0);
    if (t + 2 * tstep < ntiles) fetch(t + 2 * tstep, bv);   // the operand registers are free: the tile two steps ahead
    if (col_ok) {
      const unsigned yoff = (unsigned)(((int64_t)4 * h * p.Ncol + col_off) * (int64_t)sizeof(AT));
      AT* yb = p.y + (int64_t)b * p.M * p.Ncol;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if ((r & 3) + 8 * (r >> 2) + 4 * h >= p.M) continue;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[q][r] + add[r];
        if (p.eact) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = act_f(p.eact, v[q]);
        }
        vstore<4>(reinterpret_cast<AT*>(reinterpret_cast<char*>(yb + (int64_t)((r & 3) + 8 * (r >> 2)) * p.Ncol) + yoff), v);
      }
    }
  };

  BvT bvA, bvB;
  unsigned tile = blockIdx.x * 4 + (unsigned)wave;
  if (tile < ntiles) fetch(tile, bvA);
  if (tile + tstep < ntiles) fetch(tile + tstep, bvB);
  for (; tile < ntiles; tile += 2 * tstep) {
    run(tile, bvA);
    if (tile + tstep < ntiles) run(tile + tstep, bvB);
  }
}

// =================================================================================================
// Kernel C — two chained GEMMs for the C = 32 MLP (layers/mlp.py:54-63 behind the second pre-norm
// residual, factorizer.py:76): the 64-row hidden tensor is produced in the accumulators of GEMM 1,
// transformed in registers and consumed as the B operand of GEMM 2 WITHOUT leaving the wave.
//
// Why no data movement is needed: after GEMM 1 register r of lane (j, h) holds row
// (r&3) + 8(r>>2) + 4h of the row block at column j — and an MFMA K-step wants B[k = h-th of a
// pair][column j].  So accumulator register r IS the operand of K-step r if the A operand (the
// weights, staged in LDS) is laid out with k(step r, half h) = (r&3) + 8(r>>2) + 4h.  The order
// of a reduction is free.
//
//   forward  (BWD = false): z = W1·LN(x1) + b1 → side (kept for the backward);
//                           out = x1 + W2·gelu(z) + b2
//   backward (BWD = true):  gz = (W2ᵀ·g2) ∘ gelu'(z) → side (kept for the weight gradients);
//                           out = LayerNormBackward(W1ᵀ·gz; x1, stats, γ) + g2   (+ dγ, dβ partials)
// Saves one write + one read of the 64-channel tensor per direction against the unfused layers.
// =================================================================================================
template <typename AT>
struct ChainArgsT {
  const float* wB;     // GEMM 2 weights: A[m][k] = wB_t ? wB[k*ldwB + m] : wB[m*ldwB + k]   (m < 32, k < 64)
  int wB_t, ldwB;
  const float* biasB;  // forward: [32] or null
  AT* side;            // (B, 64, V)
  // PRE (forward, round 5): the block's out-projection in front of the chain — x1 = preW · preA + preB + preRes is formed on
  // the accumulators, written to preOut (the backward needs it) and normalised in place: x1 is never read back
  const AT* preA;      // (B, 32, V) the core's output a
  const float* preW;   // (32, 32) out_proj weight W[m][k]
  const float* preB;   // (32) or null
  const AT* preRes;    // (B, 32, V) the block input x (residual)
  AT* preOut;          // (B, 32, V) x1
  // POST (forward, with PRE): the network's head Linear(32 -> postM <= 4) on the chain's output while it is in registers
  const float* postW;  // (postM, 32)
  const float* postB;  // (postM) or null
  AT* postOut;         // (B, postM, V) or null
  int postM;
  int stagger;         // start delay of the workgroups beyond the first 256, in units of 8 192 cycles per 256 workgroups (timing only)
};

// Resident workgroups of one launch start together and walk tiles of equal length: the waves that share a SIMD then sit in
// the same phase of the tile (all in their MFMA chains, or all in the GELU / epilogue VALU phase), and the matrix pipe idles
// while the vector pipe is contended (MI355X_MICROARCH.md "two waves that run the SAME program ... try a stagger").  Workgroups
// 256 .. 511 (the second resident workgroup of every CU under round-robin placement: speed only) start `stagger` sleep
// quanta later, workgroups 512 .. twice that.  Results do not depend on it.
__device__ __forceinline__ void chain_stagger(int stagger) {
  const int n = stagger * (int)(blockIdx.x >> 8);
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
}

// HB = 32-row blocks of the hidden tensor: 2 (mlp_ratio 2, the README model) or 4 (mlp_ratio 4, the
// BraTS bundle, train.yaml:62)
// BX: both GEMMs as split-bf16 products (gemm_bx.h: three-level operands, six products of v_mfma_f32_32x32x16_bf16), weights
// pre-split in LDS as bf16x8 triples (12 KB per GEMM at HB = 2 instead of 8: two workgroups per CU instead of three).  Why: an fp32
// MFMA blocks the SIMD's vector issue for its whole duration — the 128 fp32 MFMAs of a tile were 48 % of this
// kernel's time with nothing running beside them — a bf16 MFMA for a quarter of its own.
// (Six-wave workgroups — 73 KB, two per CU, three waves per SIMD again — were tried and are slower than these four-wave ones at two
// waves per SIMD: 1.06 against 0.97 ms per step for the two launches, fp32 form 1.09; profiles/r04_chain_fwd_bx_ab.log.)
template <bool BWD, int NACC, int HB, typename AT = float, bool BX = false, bool PRE = false>
__global__ __launch_bounds__(256, (NACC == 2 && HB == 2 && !BWD && !BX) ? 3 : 2) void gemm_chain_kernel(GemmArgsT<AT> p, ChainArgsT<AT> c, int ntiles) {
  constexpr int NW = 4;
  constexpr int HID = 32 * HB, N1 = BX ? 1536 * HB : 16 * HB * 64;  // hidden rows; floats of each staged weight block
  static_assert(!BX || (!BWD && NACC == 2), "the split-bf16 form is the forward chain");
  static_assert(!PRE || (BX && HB == 2), "the out-projection in front of the chain: split-bf16 forward, hidden 64");
  __shared__ __attribute__((aligned(16))) float As1[N1];
  __shared__ __attribute__((aligned(16))) float As2[N1];
  __shared__ __attribute__((aligned(16))) float As0[PRE ? 1536 : 4];   // (PRE) out_proj weights, pre-split: [g (2)][level][lane] x 16 B
  __shared__ float tW[HID];
  __shared__ float tB[32];
  __shared__ float tB0[32];                                              // (PRE) out_proj bias
  __shared__ float tP[PRE ? 4 * 32 + 4 : 1];                             // (PRE + head) head weights, rows >= postM zero | bias
  __shared__ float red[256];
  // raw operand tile of each wave (32 channels x 32*NACC columns): the epilogue needs the SAME tensor
  // again in the accumulator layout (residual x1 / added gradient g2) — served from LDS instead of a
  // second global read (PMC: 1 of 5 resp. 8 plane-sets of traffic)
  __shared__ __attribute__((aligned(16))) float stash[NW][32][32 * NACC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int tiles_per_sample = (int)((p.Ncol + 32 * NW * NACC - 1) / (32 * NW * NACC));
  chain_stagger(c.stagger);

  if constexpr (BX) {
    // operand items of 8 steps each: As1x[g (2)][rb (HB)][level][lane], As2x[g (2 HB)][level][lane]
    for (int it = threadIdx.x; it < (PRE ? 4 * HB + 2 : 4 * HB) * 64; it += 64 * NW) {
      float wv[8];
      const int l = it & 63;
      __bf16* dst;
      if (PRE && it >= 4 * HB * 64) {   // GEMM 0: element e of lane half h = channel 2 (8g + e) + h of a (the operand tile's order)
        const int g = (it - 4 * HB * 64) >> 6;
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = c.preW[(l & 31) * 32 + 2 * (8 * g + e) + (l >> 5)];
        dst = reinterpret_cast<__bf16*>(As0) + (g * 3 * 64 + l) * 8;
      } else if (it < 2 * HB * 64) {
        const int rb = (it >> 6) % HB, g = it / (64 * HB);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          // (PRE: the column operand of GEMM 1 is x̂ in the ACCUMULATOR layout of GEMM 0 — register r = 8g + e of lane half h is
          //  channel (r & 3) + 8 (r >> 2) + 4h, the order GEMM 2 uses for the hidden tensor)
          const int r = 8 * g + e;
          const int kk = PRE ? (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) : 2 * r + (l >> 5);
          wv[e] = weight_at(p, rb * 32 + (l & 31), kk) * p.ln_g[kk];
        }
        dst = reinterpret_cast<__bf16*>(As1) + ((g * HB + rb) * 3 * 64 + l) * 8;
      } else {
        const int g = (it - 2 * HB * 64) >> 6;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int s2 = 8 * g + e, r = s2 & 15, rb = s2 >> 4;
          const int kk = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), m = l & 31;
          wv[e] = c.wB_t ? c.wB[(int64_t)kk * c.ldwB + m] : c.wB[(int64_t)m * c.ldwB + kk];
        }
        dst = reinterpret_cast<__bf16*>(As2) + (g * 3 * 64 + l) * 8;
      }
      bx8 t3[3];
      bx_split<3>(wv, t3);
#pragma unroll
      for (int i = 0; i < 3; ++i) *reinterpret_cast<bx8*>(dst + i * 64 * 8) = t3[i];
    }
  } else
  // weights in operand order (8 independent loads per thread before the LDS stores)
  for (int base = threadIdx.x; base < 2 * N1; base += 256 * 8) {
    float tmp[8];
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
      const int idx = base + uu * 256;
      float wv;
      if (idx < N1) {  // GEMM 1: step a, row block rb
        const int l = idx & 63, rb = (idx >> 6) % HB, a = idx / (64 * HB);
        const int m = rb * 32 + (l & 31), k = 2 * a + (l >> 5);
        wv = weight_at(p, m, k);
        if (!BWD) wv *= p.ln_g[k];
      } else {           // GEMM 2: step (rb, r) consumes accumulator register r of row block rb
        const int i2 = idx - N1;
        const int l = i2 & 63, s2 = i2 >> 6;
        const int r = s2 & 15, rb = s2 >> 4;
        const int k = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), m = l & 31;
        wv = c.wB_t ? c.wB[(int64_t)k * c.ldwB + m] : c.wB[(int64_t)m * c.ldwB + k];
      }
      tmp[uu] = wv;
    }
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
      const int idx = base + uu * 256;
      if (idx < N1) As1[idx] = tmp[uu]; else As2[idx - N1] = tmp[uu];
    }
  }
  if (BWD) {
    if (threadIdx.x < 32) tB[threadIdx.x] = p.lnb_g[threadIdx.x];
  } else {
    for (int r = threadIdx.x; r < HID; r += blockDim.x) {
      float t = 0.f;
      for (int k = 0; k < 32; ++k) t += weight_at(p, r, k) * p.ln_b[k];
      tW[r] = t + (p.bias ? p.bias[r] : 0.f);
      if (r < 32) tB[r] = c.biasB ? c.biasB[r] : 0.f;
      if (PRE && r < 32) tB0[r] = c.preB ? c.preB[r] : 0.f;
    }
    if constexpr (PRE) {
      if (c.postOut != nullptr && threadIdx.x < 4 * 32 + 4) {
        const int i = threadIdx.x;
        if (i < 128) tP[i] = (i >> 5) < c.postM ? c.postW[i] : 0.f;
        else tP[i] = ((i - 128) < c.postM && c.postB) ? c.postB[i - 128] : 0.f;
      }
    }
  }

  // persistent over column tiles: the operand of the NEXT tile is fetched as soon as GEMM 1 has
  // consumed the current one, so its latency hides behind the transform, GEMM 2 and the epilogue
  int tile = blockIdx.x;
  float bv[16][NACC];
  float xr[PRE ? 16 : 1][NACC];   // (PRE) residual rows of the NEXT / current tile
  // operand loads: channel 2s + h → uniform part (b*32 + 2s)*V in scalar registers + ONE lane offset
  auto fetch_tile = [&](int t) {
    const int bt = t / tiles_per_sample;
    const int64_t ct = ((int64_t)(t % tiles_per_sample) * NW + wave) * (32 * NACC) + NACC * j;
    const unsigned lo = (unsigned)h * (unsigned)p.Ncol + (unsigned)(ct < p.Ncol ? ct : 0);
    const AT* xb = (PRE ? c.preA : p.x[0]) + (int64_t)bt * 32 * p.Ncol;
#pragma unroll
    for (int s = 0; s < 16; ++s) vload<NACC>(xb + (int64_t)(2 * s) * p.Ncol + lo, bv[s]);
    if constexpr (PRE) {   // the residual rows of x in the accumulator layout (row (r & 3) + 8 (r >> 2) + 4h)
      const unsigned lr = (unsigned)(4 * h) * (unsigned)p.Ncol + (unsigned)(ct < p.Ncol ? ct : 0);
      const AT* rb0 = c.preRes + (int64_t)bt * 32 * p.Ncol;
#pragma unroll
      for (int r = 0; r < 16; ++r) vload<NACC>(rb0 + (int64_t)((r & 3) + 8 * (r >> 2)) * p.Ncol + lr, xr[PRE ? r : 0]);
    }
  };
  fetch_tile(tile);
  __syncthreads();

  for (; tile < ntiles; tile += gridDim.x) {
    // compiler-only fence: without it the loop-invariant LDS reads (row constants, 48 per lane) are
    // hoisted out of the tile loop and kept in VGPRs, which spills the accumulators
    asm volatile("" ::: "memory");
    const int b = tile / tiles_per_sample;
    const int64_t col_off = ((int64_t)(tile % tiles_per_sample) * NW + wave) * (32 * NACC) + NACC * j;
    const bool col_ok = col_off < p.Ncol;
    const int64_t nc = col_ok ? col_off : 0;
    const unsigned lane_row = (unsigned)(4 * h) * (unsigned)p.Ncol + (unsigned)nc;
    if constexpr (PRE) {
      // ---- GEMM 0: x1 = W_o a + b_o + x on the accumulators; x1 -> HBM (for the backward) and -> stash (the chain's residual);
      //      bv becomes x̂ in the ACCUMULATOR layout (register r = row (r & 3) + 8 (r >> 2) + 4h) — GEMM 1's weights are staged
      //      in that order ----
      f32x16 acc0[NACC];
#pragma unroll
      for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[q][r] = 0.f;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        bx8 aop[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
          aop[i] = *reinterpret_cast<const bx8*>(reinterpret_cast<const __bf16*>(As0) + ((g * 3 + i) * 64 + lane) * 8);
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
          float x8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x8[e] = bv[8 * g + e][q];
          bx8 bop[3];
          bx_split<3>(x8, bop);
          bx_mfma<3, 3>(acc0[q], aop, bop);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rbase = (r & 3) + 8 * (r >> 2);
        const int row = rbase + 4 * h;
        const float add = tB0[row];
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
          float v = acc0[q][r] + add + xr[PRE ? r : 0][q];
          // bf16 storage: everything downstream (LayerNorm, the chain's residual, the backward) sees the STORED x1, as in the
          // two-launch form where the chain reads it back
          if constexpr (sizeof(AT) == 2) v = (float)(AT)v;
          bv[r][q] = v;
        }
        vstore<NACC>(&stash[wave][row][NACC * j], bv[r]);
        if (col_ok) vstore<NACC>(c.preOut + ((int64_t)b * 32 + rbase) * p.Ncol + lane_row, bv[r]);
      }
    } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) vstore<NACC>(&stash[wave][2 * s + h][NACC * j], bv[s]);
    }

    if (!BWD) {
      // exact two-pass LayerNorm statistics (this lane holds the parity-h half of the channels; PRE: rows 4h + ...: also half)
      float mu[NACC], rs[NACC];
#pragma unroll
      for (int e = 0; e < NACC; ++e) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) t += bv[s][e];
        t += __shfl_xor(t, 32, 64);
        mu[e] = t / 32.0f;
      }
#pragma unroll
      for (int e = 0; e < NACC; ++e) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const float d = bv[s][e] - mu[e];
          t += d * d;
        }
        t += __shfl_xor(t, 32, 64);
        rs[e] = 1.0f / sqrtf(t / 32.0f + p.ln_eps);
      }
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int e = 0; e < NACC; ++e) bv[s][e] = (bv[s][e] - mu[e]) * rs[e];
      if (p.stats_out != nullptr && h == 0 && col_ok) {
        float* so = p.stats_out + (int64_t)b * 2 * p.Vin;
        vstore<NACC>(so + col_off, mu);
        vstore<NACC>(so + p.Vin + col_off, rs);
      }
    }

    // ---- GEMM 1: 64 rows x 128 columns per wave ----
    f32x16 acc1[HB][NACC];
#pragma unroll
    for (int rb = 0; rb < HB; ++rb)
#pragma unroll
      for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[rb][q][r] = 0.f;
    if constexpr (BX) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {   // element e of lane half h = K-step 8g + e of the fp32 form (channel 2 (8g + e) + h)
        bx8 bop[NACC][3];
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
          float x8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x8[e] = bv[8 * g + e][q];
          bx_split<3>(x8, bop[q]);
        }
#pragma unroll
        for (int rb = 0; rb < HB; ++rb) {
          bx8 aop[3];
#pragma unroll
          for (int i = 0; i < 3; ++i)
            aop[i] = *reinterpret_cast<const bx8*>(reinterpret_cast<const __bf16*>(As1) + (((g * HB + rb) * 3 + i) * 64 + lane) * 8);
#pragma unroll
          for (int q = 0; q < NACC; ++q) bx_mfma<3, 3>(acc1[rb][q], aop, bop[q]);
        }
      }
    } else {
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int rb = 0; rb < HB; ++rb) {
        const float av = As1[(s * HB + rb) * 64 + lane];
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc1[rb][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[s][q], acc1[rb][q], 0, 0, 0);
      }
    }

    // ---- prefetch the operand of the next tile (clamped re-read of this one on the last pass) ----
    fetch_tile(tile + gridDim.x < ntiles ? tile + gridDim.x : tile);

    // ---- hidden tensor: transform in registers, keep a copy in HBM for the other pass ----
    if (!BWD) {
#pragma unroll
      for (int rb = 0; rb < HB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rbase = rb * 32 + (r & 3) + 8 * (r >> 2);
          const int row = rbase + 4 * h;
          const int64_t ob = ((int64_t)b * HID + rbase) * p.Ncol;  // uniform row part; + one 32-bit lane offset
          float v[NACC];
          const float add = tW[row];
#pragma unroll
          for (int q = 0; q < NACC; ++q) v[q] = acc1[rb][q][r] + add;
          if (col_ok && c.side != nullptr) vstore<NACC>(c.side + ob + lane_row, v);   // (null: timing probe FZ_CHAIN_NOZ1)
          if constexpr (NACC == 2) {
            float gq[2];
            gelu2_f(v, gq);
            acc1[rb][0][r] = gq[0]; acc1[rb][1][r] = gq[1];
          } else {
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc1[rb][q][r] = gelu_f(v[q]);
          }
        }
    } else {
      // groups of 8 rows: 8 loads of the saved pre-activation in flight, then 8 transforms + stores
      // (bounded on purpose: the scheduler otherwise hoists all 32 loads and spills accumulators)
#pragma unroll
      for (int g8 = 0; g8 < 2 * HB; ++g8) {
        float e[8][NACC];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rr = g8 * 8 + i, rb = rr >> 4, r = rr & 15;
          const int rbase = rb * 32 + (r & 3) + 8 * (r >> 2);
          vload<NACC>(p.emul + ((int64_t)b * HID + rbase) * p.Ncol + lane_row, e[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rr = g8 * 8 + i, rb = rr >> 4, r = rr & 15;
          const int rbase = rb * 32 + (r & 3) + 8 * (r >> 2);
          float v[NACC];
#pragma unroll
          for (int q = 0; q < NACC; ++q) v[q] = acc1[rb][q][r] * gelu_grad_f(e[i][q]);
          if (col_ok) vstore<NACC>(c.side + ((int64_t)b * HID + rbase) * p.Ncol + lane_row, v);
#pragma unroll
          for (int q = 0; q < NACC; ++q) acc1[rb][q][r] = v[q];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- GEMM 2: 32 rows, K = 64 straight from the accumulators of GEMM 1 ----
    f32x16 acc2[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[q][r] = 0.f;
    if constexpr (BX) {
#pragma unroll
      for (int g = 0; g < 2 * HB; ++g) {   // steps (rb, r) = (g >> 1, 8 (g & 1) + e): accumulator registers as the column operand
        bx8 aop[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
          aop[i] = *reinterpret_cast<const bx8*>(reinterpret_cast<const __bf16*>(As2) + ((g * 3 + i) * 64 + lane) * 8);
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
          float x8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x8[e] = acc1[g >> 1][q][8 * (g & 1) + e];
          bx8 bop[3];
          bx_split<3>(x8, bop);
          bx_mfma<3, 3>(acc2[q], aop, bop);
        }
      }
    } else {
#pragma unroll
    for (int rb = 0; rb < HB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float av = As2[(rb * 16 + r) * 64 + lane];
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc2[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, acc1[rb][q][r], acc2[q], 0, 0, 0);
      }
    }

    if (BWD) {
      lnbwd_block<NACC, true, true>(p, acc2, b, col_off, col_ok, lane, wave, red, tile, tB, &stash[wave][0][0]);
      __syncthreads();  // red is reused by the next tile
    } else if (col_ok) {
      const bool post = PRE && c.postOut != nullptr;   // uniform
      float pl[4][NACC] = {};
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rbase = (r & 3) + 8 * (r >> 2);
        const int row = rbase + 4 * h;
        const int64_t ob = ((int64_t)b * 32 + rbase) * p.Ncol;
        const float add = tB[row];
        float e[NACC], v[NACC];
        vload<NACC>(&stash[wave][row][NACC * j], e);
#pragma unroll
        for (int q = 0; q < NACC; ++q) v[q] = acc2[q][r] + add + e[q];
        vstore<NACC>(p.y + ob + lane_row, v);
        if (PRE && post) {   // the head sees what a separate launch would read back: the STORED value (bf16 storage: rounded)
#pragma unroll
          for (int q = 0; q < NACC; ++q) {
            const float vs = sizeof(AT) == 2 ? (float)(AT)v[q] : v[q];
#pragma unroll
            for (int o = 0; o < 4; ++o) pl[o][q] += tP[o * 32 + row] * vs;
          }
        }
      }
      if (PRE && post) {   // rows 4h + ... of this lane + the other half's (same column: both lanes are active together)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          float v[NACC];
#pragma unroll
          for (int q = 0; q < NACC; ++q) v[q] = pl[o][q] + __shfl_xor(pl[o][q], 32, 64) + tP[128 + o];
          if (h == 0 && o < c.postM) vstore<NACC>(c.postOut + ((int64_t)b * c.postM + o) * p.Ncol + nc, v);
        }
      }
    }
  }
}

// Eight K-steps of two of an fp32-MFMA loop as ONE split-bf16 K-step (gemm_bx.hip) for NRB row blocks x NQ column blocks:
// load_a(rb, a8) = the lane's weights of the eight steps (fp32, from the LDS operand image of the fp32 form, split here),
// get_x(q, x8) = the column operands of the same steps.  Element e of lane half h of v_mfma_f32_32x32x16_bf16 = step e of the
// group: any assignment of reduction indices to (half, element) slots is valid as long as both operands use the same one.
// HOIST splits the column operands once for all row blocks (NQ x NTB x 4 more live registers); without it they are split
// per row block (the fp32 chain kernels sit at the 256-register limit).
template <bool HOIST, int NRB, int NQ, int NTA, int NTB, typename FA, typename FX>
__device__ __forceinline__ void bx_group(f32x16 (&acc)[NRB][NQ], FA load_a, FX get_x) {
  if constexpr (HOIST) {
    bx8 bop[NQ][NTB];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float x8[8];
      get_x(q, x8);
      bx_split<NTB>(x8, bop[q]);
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      float a8[8];
      load_a(rb, a8);
      bx8 aop[NTA];
      bx_split<NTA>(a8, aop);
#pragma unroll
      for (int q = 0; q < NQ; ++q) bx_mfma<NTA, NTB>(acc[rb][q], aop, bop[q]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      bx8 bop[NTB];
      {
        float x8[8];
        get_x(q, x8);
        bx_split<NTB>(x8, bop);
      }
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        float a8[8];
        load_a(rb, a8);
        bx8 aop[NTA];
        bx_split<NTA>(a8, aop);
        bx_mfma<NTA, NTB>(acc[rb][q], aop, bop);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// The same group with the row operands PRE-SPLIT in LDS (load_term(rb, t) = level t of the row operand, one ds_read_b128):
// no weight split on the VALU; the levels are fetched one at a time — a_0 (b_0 + b_1 + b_2), a_1 (b_0 + b_1), a_2 b_0, small
// products first within a level — so only four operand registers are live next to the split column operand.
template <int NRB, int NQ, int NTB, typename FA, typename FX>
__device__ __forceinline__ void bx_group_ps(f32x16 (&acc)[NRB][NQ], FA load_term, FX get_x) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    bx8 bop[NTB];
    {
      float x8[8];
      get_x(q, x8);
      bx_split<NTB>(x8, bop);
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
      for (int t = 2; t >= 0; --t) {
        const bx8 a = load_term(rb, t);
#pragma unroll
        for (int jj = NTB - 1; jj >= 0; --jj)
          if (t + jj <= 2) acc[rb][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bop[jj], acc[rb][q], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);   // (keeps the operand reads of the next row block from being hoisted: registers)
    }
  }
}

// =================================================================================================
// MLP chain for C = 64, hidden 128 (stage 1 of the README model): the same two chained GEMMs as gemm_chain_kernel,
// with the hidden tensor produced and consumed in TWO passes of 64 rows — 64 accumulator registers for the pass,
// 64 for the 64-row result that GEMM 2 accumulates over both passes, 64 for the operand tile — so the chain still
// fits 256 VGPRs at two workgroups per CU with 8-byte lane loads.  Weights of both GEMMs (2 x 32 KB) sit in LDS in
// operand order; the residual (forward) / added gradient and the LayerNorm input (backward) are re-read in the
// accumulator layout (L2 / MALL) instead of being stashed.
//   forward : z = W1·LN(x1) + b1 -> side ; out = x1 + W2·gelu(z) + b2          5 plane-sets against 7 unfused
//   backward: gz = (W2ᵀ g2) ∘ gelu'(z) -> side ; out = LNbwd(W1ᵀ gz) + g2      8 against 12 (+ dγ, dβ partial rows)
// =================================================================================================
// SINGLE (BWD only): ONE 64 -> 64 input-gradient GEMM (in_proj of a C = 64 block) in front of the same LayerNorm-backward
// epilogue — fz_gemm with EPI_LNBWD and M = K = 64: the pre-LayerNorm gradient never reaches HBM.
// BX: every GEMM of the chain on split-bf16 products — the weights stay fp32 in LDS (64 KB: a pre-split image would be 96 KB
// and halve the occupancy) and are split per use, the column operands once per group of eight steps.
// P512 (BX, not SINGLE; both storage types): the fp32-weight form above does not fit 256 registers once the operand splits are
// added (6 / 23 spilled), so the split-bf16 chain runs as ONE workgroup of 512 threads per CU — two independent 4-wave
// halves, each walking its own tiles — sharing a PRE-SPLIT weight image (bf16x8 triples in operand order: 2 x 48 KB):
// same two waves per SIMD, no weight splits on the VALU, 3 ds_read_b128 per row operand instead of 8 ds_read_b32.
// PRE (forward, P512) [r5]: the block's out-projection in front of the chain, as gemm_chain_kernel<.., PRE> does at C = 32 —
// the tile loaded is a, GEMM 0 forms x1 = W_o·a + b_o + x on 64 accumulator registers (rounded to the stored value under bf16
// storage), x1 goes to preOut, is normalised in place and feeds GEMM 1 as the column operand in the ACCUMULATOR layout (the W1
// image is staged in that k order: the order the W2 image always had); the residual of the epilogue re-reads the lane's own x1.
// A third pre-split image (W_o: 24 KB) joins the two: 121 KB of LDS.
template <bool BWD, typename AT, bool SINGLE = false, bool BX = false, bool P512 = false, bool PRE = false>
__global__ __launch_bounds__(P512 ? 512 : 256, 2) void gemm_chain64_kernel(GemmArgsT<AT> p, ChainArgsT<AT> c, int ntiles) {
  constexpr int NACC = 2, C = 64, HID = 128;
  constexpr int NTA = BxTerms<AT>::A, NTB = bx_terms_b<AT>(BXPRO_GELU);
  constexpr bool HOIST = SINGLE || sizeof(AT) == 2;
  static_assert(!P512 || (BX && !SINGLE), "P512: the split-bf16 chain around a pre-split weight image");
  static_assert(!PRE || (P512 && !BWD), "PRE: the forward chain around the pre-split images");
  constexpr int NA = P512 ? 12288 : 8192;   // floats of one weight image (P512: [16 (group, row block)][3 terms][64 lanes] x 16 B)
  constexpr int NA0 = PRE ? 6144 : 0;       // the W_o image: [8 (group, row block)][3 terms][64 lanes] x 16 B
  extern __shared__ __attribute__((aligned(16))) float fz_lds_c64[];
  float* As1 = fz_lds_c64;            // [32 steps][4 row blocks][64]
  float* As2 = As1 + NA;              // [4 x 16 (rb, r) steps][2 row blocks][64]
  float* tW = As2 + NA + NA0;         // [128]
  float* tB = tW + 128;               // [64]
  float* tB0 = tB + 64;               // [64] (PRE: the out-projection's bias)
  const int half = P512 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;   // P512: which 4-wave half of the workgroup (wave-uniform)
  float* red = tB + 64 + (PRE ? 64 : 0) + half * 512;  // [4][128] per half
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) & 3);
  const int j = lane & 31, h = lane >> 5;
  const int tiles_per_sample = (int)((p.Ncol + 128 * NACC - 1) / (128 * NACC));
  // P512 operand reads: ONE opaque per-lane base per image + compile-time element offsets (ds_read_b128 immediates; left
  // to itself the optimiser materialises a loop-invariant VGPR address per (slot, level) — 40 of them — and spills)
  // P512 operand reads: one OPAQUE per-lane float index per image + compile-time slot offsets (ds_read_b128 immediates are
  // 16 bits: the second image starts at 48 KB, and left to itself the optimiser keeps one loop-invariant VGPR address for
  // every slot beyond 64 KB — 32 of them — and spills)
  int lane4 = lane * 4, lane4b = lane * 4 + NA, lane4c = lane * 4 + 2 * NA;
  if constexpr (P512) {
    asm volatile("" : "+v"(lane4));
    asm volatile("" : "+v"(lane4b));
    if constexpr (PRE) asm volatile("" : "+v"(lane4c));
  }
  auto ld_a1 = [&](int slot3) { return *reinterpret_cast<const bx8*>(As1 + slot3 * 256 + lane4); };
  auto ld_a2 = [&](int slot3) { return *reinterpret_cast<const bx8*>(As1 + slot3 * 256 + lane4b); };
  auto ld_a0 = [&](int slot3) { return *reinterpret_cast<const bx8*>(As1 + slot3 * 256 + lane4c); };
  (void)ld_a0; (void)lane4c; (void)tB0;

  if constexpr (P512) {
    // item = (image, slot [16], lane): eight weights -> three bf16 levels -> three 16-byte stores
    for (int item = threadIdx.x; item < (PRE ? 2560 : 2048); item += 512) {
      const int l = item & 63, slot = (item >> 6) & 15, img = item >> 10;
      float a8[8];
      if (img == 0) {        // slot = g*4 + rb: A1[m = rb*32 + (l & 31)][k = 2 (8g + e) + (l >> 5)]
        const int g = slot >> 2, rb = slot & 3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          // PRE: the column operand of GEMM 1 is the accumulator tile of GEMM 0 — k = row (mb = g >> 1, r = 8 (g & 1) + e, lane half)
          const int rr = 8 * (g & 1) + e;
          const int k = PRE ? (g >> 1) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * (l >> 5) : 2 * (8 * g + e) + (l >> 5);
          a8[e] = weight_at(p, rb * 32 + (l & 31), k);
          if (!BWD) a8[e] *= p.ln_g[k];
        }
      } else if (PRE && img == 2) {   // slot = g*2 + mb: A0[m = mb*32 + (l & 31)][k = 2 (8g + e) + (l >> 5)] = W_o[m][k]
        const int g = slot >> 1, mb = slot & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) a8[e] = c.preW[(int64_t)(mb * 32 + (l & 31)) * 64 + 2 * (8 * g + e) + (l >> 5)];
      } else {               // slot = (rb4*2 + g8)*2 + mb: A2[m = mb*32 + (l & 31)][k = rb4*32 + row(r = 8 g8 + e) + 4 (l >> 5)]
        const int mb = slot & 1, g8 = (slot >> 1) & 1, rb4 = slot >> 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = 8 * g8 + e;
          const int k = rb4 * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), m = mb * 32 + (l & 31);
          a8[e] = c.wB_t ? c.wB[(int64_t)k * c.ldwB + m] : c.wB[(int64_t)m * c.ldwB + k];
        }
      }
      bx8 t3[3];
      bx_split<3>(a8, t3);
      bx8* dst = reinterpret_cast<bx8*>(img == 0 ? As1 : (img == 1 ? As2 : As2 + NA)) + (slot * 3) * 64 + l;
      dst[0] = t3[0]; dst[64] = t3[1]; dst[128] = t3[2];
    }
  }
  for (int base = threadIdx.x; !P512 && base < (SINGLE ? 4096 : 16384); base += 256 * 8) {
    float tmp[8];
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
      const int idx = base + uu * 256;
      float wv;
      if (SINGLE) {   // A[m = mb*32 + (l & 31)][k = 2a + (l >> 5)], [32 steps][2 row blocks][64]
        const int l = idx & 63, mb = (idx >> 6) & 1, a = idx >> 7;
        wv = weight_at(p, mb * 32 + (l & 31), 2 * a + (l >> 5));
      } else if (idx < 8192) {
        const int l = idx & 63, rb = (idx >> 6) & 3, a = idx >> 8;
        const int m = rb * 32 + (l & 31), k = 2 * a + (l >> 5);
        wv = weight_at(p, m, k);
        if (!BWD) wv *= p.ln_g[k];
      } else {
        const int i2 = idx - 8192;
        const int l = i2 & 63, mb = (i2 >> 6) & 1, s2 = i2 >> 7;
        const int r = s2 & 15, rb = s2 >> 4;
        const int k = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), m = mb * 32 + (l & 31);
        wv = c.wB_t ? c.wB[(int64_t)k * c.ldwB + m] : c.wB[(int64_t)m * c.ldwB + k];
      }
      tmp[uu] = wv;
    }
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) fz_lds_c64[base + uu * 256] = tmp[uu];
  }
  if (BWD) {
    if (threadIdx.x < C) tB[threadIdx.x] = p.lnb_g[threadIdx.x];
  } else {
    for (int r = threadIdx.x; r < HID; r += blockDim.x) {
      float t = 0.f;
      for (int k = 0; k < C; ++k) t += weight_at(p, r, k) * p.ln_b[k];
      tW[r] = t + (p.bias ? p.bias[r] : 0.f);
      if (r < C) tB[r] = c.biasB ? c.biasB[r] : 0.f;
      if (PRE && r < C) tB0[r] = c.preB ? c.preB[r] : 0.f;
    }
  }

  // P512: the halves take tiles 2 i and 2 i + 1 (ntiles is even — host-checked — so both run the same number of rounds
  // and meet at the same barriers)
  const int tstep = P512 ? 2 * (int)gridDim.x : (int)gridDim.x;
  int tile = P512 ? 2 * (int)blockIdx.x + half : (int)blockIdx.x;
  float bv[32][NACC];
  auto fetch_tile = [&](int t) {
    const int bt = t / tiles_per_sample;
    const int64_t ct = ((int64_t)(t % tiles_per_sample) * 4 + wave) * (32 * NACC) + NACC * j;
    const unsigned lo = (unsigned)h * (unsigned)p.Ncol + (unsigned)(ct < p.Ncol ? ct : 0);
    const AT* xb = (PRE ? c.preA : p.x[0]) + (int64_t)bt * C * p.Ncol;
#pragma unroll
    for (int s = 0; s < 32; ++s) vload<NACC>(xb + (int64_t)(2 * s) * p.Ncol + lo, bv[s]);
  };
  fetch_tile(tile);
  __syncthreads();

  for (; tile < ntiles; tile += tstep) {
    asm volatile("" ::: "memory");
    const int b = tile / tiles_per_sample;
    const int64_t col_off = ((int64_t)(tile % tiles_per_sample) * 4 + wave) * (32 * NACC) + NACC * j;
    const bool col_ok = col_off < p.Ncol;
    const int64_t nc = col_ok ? col_off : 0;
    const unsigned lane_row = (unsigned)(4 * h) * (unsigned)p.Ncol + (unsigned)nc;

    f32x16 acc0[PRE ? 2 : 1][NACC];   // PRE: x1, then LN(x1), rows (mb, r, lane half) x the lane's two voxels
    if constexpr (PRE) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc0[mb][q][r] = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bx_group_ps<2, NACC, NTB>(acc0,
            [&](int mb, int t) { return ld_a0((g * 2 + mb) * 3 + t); },
            [&](int q, float (&x8)[8]) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x8[e] = bv[8 * g + e][q];
            });
        __builtin_amdgcn_sched_barrier(0);
      }
      const int64_t smp = (int64_t)b * C * p.Ncol;
      float s1[NACC] = {0.f, 0.f};
#pragma unroll
      for (int g8 = 0; g8 < 4; ++g8) {
        float e[8][NACC];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rr = g8 * 8 + i, mb = rr >> 4, r = rr & 15;
          vload<NACC>(c.preRes + smp + (int64_t)(mb * 32 + (r & 3) + 8 * (r >> 2)) * p.Ncol + lane_row, e[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rr = g8 * 8 + i, mb = rr >> 4, r = rr & 15;
          const int rbase = mb * 32 + (r & 3) + 8 * (r >> 2);
          const float add = tB0[rbase + 4 * h];
          float v[NACC];
#pragma unroll
          for (int q = 0; q < NACC; ++q) v[q] = acc0[mb][q][r] + add + e[i][q];
          if (col_ok) vstore<NACC>(c.preOut + smp + (int64_t)rbase * p.Ncol + lane_row, v);
#pragma unroll
          for (int q = 0; q < NACC; ++q) {
            if constexpr (sizeof(AT) == 2) v[q] = (float)(AT)v[q];   // the MLP sees the STORED x1, as the two-launch form does
            acc0[mb][q][r] = v[q];
            s1[q] += v[q];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      float mu[NACC], rs[NACC];
#pragma unroll
      for (int q = 0; q < NACC; ++q) {
        s1[q] += __shfl_xor(s1[q], 32, 64);
        mu[q] = s1[q] / 64.0f;
        float t = 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float dd = acc0[mb][q][r] - mu[q];
            t += dd * dd;
          }
        t += __shfl_xor(t, 32, 64);
        rs[q] = 1.0f / sqrtf(t / 64.0f + p.ln_eps);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc0[mb][q][r] = (acc0[mb][q][r] - mu[q]) * rs[q];
      }
      if (p.stats_out != nullptr && h == 0 && col_ok) {
        float* so = p.stats_out + (int64_t)b * 2 * p.Vin;
        vstore<NACC>(so + col_off, mu);
        vstore<NACC>(so + p.Vin + col_off, rs);
      }
    }

    if (!BWD && !PRE) {
      float mu[NACC], rs[NACC];
#pragma unroll
      for (int e = 0; e < NACC; ++e) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) t += bv[s][e];
        t += __shfl_xor(t, 32, 64);
        mu[e] = t / 64.0f;
      }
#pragma unroll
      for (int e = 0; e < NACC; ++e) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
          const float d = bv[s][e] - mu[e];
          t += d * d;
        }
        t += __shfl_xor(t, 32, 64);
        rs[e] = 1.0f / sqrtf(t / 64.0f + p.ln_eps);
      }
#pragma unroll
      for (int s = 0; s < 32; ++s)
#pragma unroll
        for (int e = 0; e < NACC; ++e) bv[s][e] = (bv[s][e] - mu[e]) * rs[e];
      if (p.stats_out != nullptr && h == 0 && col_ok) {
        float* so = p.stats_out + (int64_t)b * 2 * p.Vin;
        vstore<NACC>(so + col_off, mu);
        vstore<NACC>(so + p.Vin + col_off, rs);
      }
    }

    f32x16 acc2[2][NACC];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[mb][q][r] = 0.f;

    if (SINGLE) {
      if constexpr (BX) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bx_group<HOIST, 2, NACC, NTA, NTB>(acc2,
              [&](int mb, float (&a8)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a8[e] = As1[((8 * g + e) * 2 + mb) * 64 + lane];
              },
              [&](int q, float (&x8)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x8[e] = bv[8 * g + e][q];
              });
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int s = 0; s < 32; ++s) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const float av = As1[(s * 2 + mb) * 64 + lane];
#pragma unroll
          for (int q = 0; q < NACC; ++q) acc2[mb][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[s][q], acc2[mb][q], 0, 0, 0);
        }
        if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      }
      fetch_tile(tile + tstep < ntiles ? tile + tstep : tile);
    }
#pragma unroll
    for (int p2 = 0; p2 < (SINGLE ? 0 : 2); ++p2) {
      // ---- GEMM 1, hidden rows 64·p2 .. 64·p2 + 63 ----
      f32x16 acc1[2][NACC];
#pragma unroll
      for (int rbl = 0; rbl < 2; ++rbl)
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc1[rbl][q][r] = 0.f;
      if constexpr (BX) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if constexpr (P512)
            bx_group_ps<2, NACC, NTB>(acc1,
                [&](int rbl, int t) { return ld_a1((g * 4 + 2 * p2 + rbl) * 3 + t); },
                [&](int q, float (&x8)[8]) {
#pragma unroll
                  for (int e = 0; e < 8; ++e) x8[e] = PRE ? acc0[PRE ? (g >> 1) : 0][q][8 * (g & 1) + e] : bv[8 * g + e][q];
                });
          else
          bx_group<HOIST, 2, NACC, NTA, NTB>(acc1,
              [&](int rbl, float (&a8)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a8[e] = As1[((8 * g + e) * 4 + 2 * p2 + rbl) * 64 + lane];
              },
              [&](int q, float (&x8)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x8[e] = bv[8 * g + e][q];
              });
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int s = 0; s < 32; ++s) {
#pragma unroll
        for (int rbl = 0; rbl < 2; ++rbl) {
          const float av = As1[(s * 4 + 2 * p2 + rbl) * 64 + lane];
#pragma unroll
          for (int q = 0; q < NACC; ++q) acc1[rbl][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[s][q], acc1[rbl][q], 0, 0, 0);
        }
        if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      }
      if (p2 == 1) fetch_tile(tile + tstep < ntiles ? tile + tstep : tile);   // the operand tile is consumed

      // ---- hidden rows: transform in registers, copy to HBM for the other pass ----
      if (!BWD) {
#pragma unroll
        for (int rbl = 0; rbl < 2; ++rbl)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rbase = (2 * p2 + rbl) * 32 + (r & 3) + 8 * (r >> 2);
            const float add = tW[rbase + 4 * h];
            float v[NACC];
#pragma unroll
            for (int q = 0; q < NACC; ++q) v[q] = acc1[rbl][q][r] + add;
            if (col_ok) vstore<NACC>(c.side + ((int64_t)b * HID + rbase) * p.Ncol + lane_row, v);
            if constexpr (NACC == 2) {
              float gq[2];
              gelu2_f(v, gq);
              acc1[rbl][0][r] = gq[0]; acc1[rbl][1][r] = gq[1];
            } else {
#pragma unroll
              for (int q = 0; q < NACC; ++q) acc1[rbl][q][r] = gelu_f(v[q]);
            }
          }
      } else {
#pragma unroll
        for (int g8 = 0; g8 < 4; ++g8) {
          float e[8][NACC];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int rr = g8 * 8 + i, rbl = rr >> 4, r = rr & 15;
            const int rbase = (2 * p2 + rbl) * 32 + (r & 3) + 8 * (r >> 2);
            vload<NACC>(p.emul + ((int64_t)b * HID + rbase) * p.Ncol + lane_row, e[i]);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int rr = g8 * 8 + i, rbl = rr >> 4, r = rr & 15;
            const int rbase = (2 * p2 + rbl) * 32 + (r & 3) + 8 * (r >> 2);
            float v[NACC];
#pragma unroll
            for (int q = 0; q < NACC; ++q) v[q] = acc1[rbl][q][r] * gelu_grad_f(e[i][q]);
            if (col_ok) vstore<NACC>(c.side + ((int64_t)b * HID + rbase) * p.Ncol + lane_row, v);
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc1[rbl][q][r] = v[q];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      // ---- GEMM 2 += (64 result rows) x (these 64 hidden rows), straight from the accumulators ----
      if constexpr (BX) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {   // steps (rbl, r) = (g >> 1, 8 (g & 1) + e): accumulator registers as the column operand
          if constexpr (P512)
            bx_group_ps<2, NACC, NTB>(acc2,
                [&](int mb, int t) { return ld_a2((((2 * p2 + (g >> 1)) * 2 + (g & 1)) * 2 + mb) * 3 + t); },
                [&](int q, float (&x8)[8]) {
#pragma unroll
                  for (int e = 0; e < 8; ++e) x8[e] = acc1[g >> 1][q][8 * (g & 1) + e];
                });
          else
          bx_group<HOIST, 2, NACC, NTA, NTB>(acc2,
              [&](int mb, float (&a8)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a8[e] = As2[(((2 * p2 + (g >> 1)) * 16 + 8 * (g & 1) + e) * 2 + mb) * 64 + lane];
              },
              [&](int q, float (&x8)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x8[e] = acc1[g >> 1][q][8 * (g & 1) + e];
              });
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int rbl = 0; rbl < 2; ++rbl)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const float av = As2[(((2 * p2 + rbl) * 16 + r) * 2 + mb) * 64 + lane];
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc2[mb][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, acc1[rbl][q][r], acc2[mb][q], 0, 0, 0);
          }
          if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    const int64_t sample = (int64_t)b * C * p.Ncol;
    if (!BWD) {
      // out = acc2 + b2 + x1 (residual re-read in the accumulator layout), 8 rows at a time
#pragma unroll
      for (int g8 = 0; g8 < 4; ++g8) {
        float e[8][NACC];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rr = g8 * 8 + i, mb = rr >> 4, r = rr & 15;
          vload<NACC>(p.res + sample + (int64_t)(mb * 32 + (r & 3) + 8 * (r >> 2)) * p.Ncol + lane_row, e[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rr = g8 * 8 + i, mb = rr >> 4, r = rr & 15;
          const int rbase = mb * 32 + (r & 3) + 8 * (r >> 2);
          const float add = tB[rbase + 4 * h];
          float v[NACC];
#pragma unroll
          for (int q = 0; q < NACC; ++q) v[q] = acc2[mb][q][r] + add + e[i][q];
          if (col_ok) vstore<NACC>(p.y + sample + (int64_t)rbase * p.Ncol + lane_row, v);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // LayerNorm backward over the 64 channels of this lane's voxels (rows (mb, r, h)) + added gradient
      const float* sp = p.lnb_stats + (int64_t)b * 2 * p.Ncol;
      float mu[NACC], rs[NACC];
      vload<NACC>(sp + nc, mu);
      vload<NACC>(sp + p.Ncol + nc, rs);
      float xs[32][NACC];
      float m1[NACC] = {0.f, 0.f}, m2[NACC] = {0.f, 0.f};
#pragma unroll
      for (int g8 = 0; g8 < 4; ++g8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rr = g8 * 8 + i, mb = rr >> 4, r = rr & 15;
          vload<NACC>(p.lnb_x + sample + (int64_t)(mb * 32 + (r & 3) + 8 * (r >> 2)) * p.Ncol + lane_row, xs[rr]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rr = g8 * 8 + i, mb = rr >> 4, r = rr & 15;
          const float gc = tB[mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
          for (int q = 0; q < NACC; ++q) {
            const float av = acc2[mb][q][r] * gc;
            xs[rr][q] = (xs[rr][q] - mu[q]) * rs[q];
            m1[q] += av;
            m2[q] += av * xs[rr][q];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < NACC; ++q) {
        m1[q] = (m1[q] + __shfl_xor(m1[q], 32, 64)) * (1.0f / 64.0f);
        m2[q] = (m2[q] + __shfl_xor(m2[q], 32, 64)) * (1.0f / 64.0f);
      }
#pragma unroll
      for (int g8 = 0; g8 < 4; ++g8) {
        float ga[8][NACC];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rr = g8 * 8 + i, mb = rr >> 4, r = rr & 15;
          vload<NACC>(p.lnb_gadd + sample + (int64_t)(mb * 32 + (r & 3) + 8 * (r >> 2)) * p.Ncol + lane_row, ga[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rr = g8 * 8 + i, mb = rr >> 4, r = rr & 15;
          const int rbase = mb * 32 + (r & 3) + 8 * (r >> 2);
          const int row = rbase + 4 * h;
          const float gc = tB[row];
          float v[NACC];
#pragma unroll
          for (int q = 0; q < NACC; ++q) v[q] = rs[q] * (acc2[mb][q][r] * gc - m1[q] - xs[rr][q] * m2[q]) + ga[i][q];
          if (col_ok) vstore<NACC>(p.y + sample + (int64_t)rbase * p.Ncol + lane_row, v);
          float sg = col_ok ? acc2[mb][0][r] * xs[rr][0] + acc2[mb][1][r] * xs[rr][1] : 0.f;
          float sb = col_ok ? acc2[mb][0][r] + acc2[mb][1][r] : 0.f;
          sg = half_sum32(sg);
          sb = half_sum32(sb);
          if ((lane & 31) == 31) {
            red[wave * 128 + row] = sg;
            red[wave * 128 + 64 + row] = sb;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      if ((threadIdx.x & 255) < 128) {
        const int e = threadIdx.x & 255;
        p.lnb_part[(int64_t)tile * 128 + e] = (red[e] + red[128 + e]) + (red[256 + e] + red[384 + e]);
      }
      __syncthreads();
    }
  }
}

// =================================================================================================
// MLP backward chain WITH the two weight gradients (C = 32, hidden 64): the unfused step reads g2 and z1
// again for dW2 = g2 ⊗ gelu(z1) and writes + re-reads gz1 for dW1 = gz1 ⊗ LN(x1) — 13 plane-sets of traffic per
// block (7 chain + 3 + 3) where 5 suffice (g2, z1 ×2, x1 in; gx1 out).  A weight gradient reduces over VOXELS, so
// its MFMA operands need the channel on the lane axis; everything in the chain has the voxel there.  Each wave
// turns its tile through wave-private LDS ([channel][voxel] rows, stride 66 ≡ 2 (mod 32): ds_read_b32 banks are
// (a/4) mod 32 per 32-lane half, and the (channel16, k4 ∈ {0,1} resp. {2,3}) lanes of a v_mfma_f32_16x16x4_f32 operand
// read then hit 32 different banks; the 8-byte tile writes / accumulator-layout reads are conflict-free at any stride):
//   Bf  32 x 64: g2 (operand of dW2) during the first pass, then LN-normalised x1 (operand of dW1, and the
//                LayerNorm backward reads it back in the accumulator layout);
//   T   16 x 64: one 16-channel block of gelu(z1) (pass A) resp. gz1 (pass B) at a time.
// The (dW2 | dW1 | db2 | db1) sums stay in registers across the tiles of the persistent workgroup, are added
// over its four waves through LDS at the end and leave as one row of `wpart` per workgroup;
// the FK_CHAIN_WG job of the finish kernel (finish.h) adds the rows in index order (no float atomics) and applies the LayerNorm affine to dW1.
// 13 KB of LDS per wave + 64 accumulator registers: two workgroups per CU (the plain chain runs three).
// =================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kTS = 66;                       // LDS row stride of the transposable tiles (floats): ≡ 2 (mod 32), even
// (kWgRow — floats of one wpart row: dW2 [32][64] | S1 [64][32] | db2 | db1 | dγ | dβ — lives in finish.h with the job that adds the rows)

// gelu(x) and gelu'(x) with ONE exponential: erf(x/√2) by Abramowitz-Stegun 7.1.26 (fast_erf, fz_common.h) needs
// exp(−x²/2), which is also the Gaussian density of gelu'
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float E = __expf(-0.5f * x * x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * E;                       // erf(|x|/√2)
  const float cdf = 0.5f * (1.0f + __builtin_copysignf(r, x));
  g = x * cdf;
  dg = cdf + x * (0.3989422804014327f * E);
}

// The same for the lane's two voxels at once on packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of fp32 per
// issue slot; the exponentials and reciprocals stay scalar) — same operations in the same order, so the results are
// those of gelu_both; ≈ 23 instead of 44 VALU instructions per voxel pair in the VALU-heaviest phase of the fused kernel.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool SEL = false>   // SEL: sign by compare + select (the two-launch hidden-128 form keeps its round-3 register allocation)
__device__ __forceinline__ void gelu_both2(const float (&x)[2], float (&g)[2], float (&dg)[2]) {
  const f32x2 xv = {x[0], x[1]};
  const f32x2 ax = f32x2{fabsf(x[0]), fabsf(x[1])} * 0.70710678118654752f;
  const f32x2 xx = (xv * -0.5f) * xv;
  const f32x2 E = {__expf(xx[0]), __expf(xx[1])};
  const f32x2 den = ax * 0.3275911f + 1.0f;
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const f32x2 poly = t * (t * (t * (t * (t * 1.061405429f + -1.453152027f) + 1.421413741f) + -0.284496736f) + 0.254829592f);
  const f32x2 r = 1.0f - poly * E;
  f32x2 rs;
  if constexpr (SEL) rs = f32x2{x[0] < 0.f ? -r[0] : r[0], x[1] < 0.f ? -r[1] : r[1]};
  else rs = f32x2{__builtin_copysignf(r[0], x[0]), __builtin_copysignf(r[1], x[1])};   // one v_bfi_b32 instead of compare + select (fast_erf, fz_common.h)
  const f32x2 cdf = (rs + 1.0f) * 0.5f;
  const f32x2 gv = xv * cdf;
  const f32x2 dv = cdf + xv * (E * 0.3989422804014327f);
  g[0] = gv[0]; g[1] = gv[1];
  dg[0] = dv[0]; dg[1] = dv[1];
}

// (HALVES / HALF are compile-time: with a run-time half the one-launch form lost its spill-free register allocation —
// 32 spilled VGPRs, 0.94 -> 1.03 ms per launch, 215 MB of scratch writes in the WRITE_SIZE counter.)
// BX: the two input-gradient GEMMs (half of the kernel's matrix work) run as split-bf16 products (gemm_bx.hip): the K-steps
// of two of the fp32 form are packed eight at a time — element e of lane half h of a 32x32x16 bf16 MFMA = step 8g + e —
// with the weights pre-split in LDS (As1 / As2 hold bf16x8 triples instead of floats: 12 KB each instead of 8).  The two
// weight-gradient passes keep their transposed fp32 operands (v_mfma_f32_16x16x4_f32).
// WGB [r6]: the two weight-gradient passes on the bf16 matrix pipe as well.  Each value that enters a voxel reduction (g2, gelu(z1),
// x̂, gz1) is split ONCE, by the lane that holds it, into two bf16 levels (hi = rne(x), lo = rne(x - hi): 16 significand bits) and
// parked in wave-private LDS as a hi plane and a lo plane of [channel][64 voxels] bf16 — 2 x 2 bytes per value, the fp32 footprint —
// with the 16-byte voxel chunks of a row XOR-swizzled by (row & 7): the pair stores of the voxel-owner lanes and the 16-byte operand
// reads of the channel-owner lanes (lane (l16, k4) = channel l16, voxels 32 ks + 8 k4 .. + 7: one ds_read_b128 per level) are both
// conflict-free.  a·b = a_lo·b_hi + a_hi·b_lo + a_hi·b_hi on v_mfma_f32_16x16x32_bf16: 96 MFMAs of 16 cycles per tile that overlap
// the other wave's vector work, in place of 256 exclusive v_mfma_f32_16x16x4_f32 of 32 cycles.  The accumulator layout of the
// 16x16 tile does not depend on K: dW2 / dW1 registers, the partial rows and the finish job are unchanged.  g2's two levels are the
// ones GEMM 1 splits anyway; the bias sums Σ_v g2, Σ_v gz1 come from the same operand registers through v_dot2c_f32_bf16.
// Error: each product carries 2^-16 relative (the dropped a_lo·b_lo and third levels), random in sign over the >= 10^5 voxels of a
// sum (tests/test_gpu_dense.py: against float64 next to the fp32-MFMA form).
typedef __bf16 wg2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wg_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  fx2 v = {x0, x1};
  const wg2 a = __builtin_convertvector(v, wg2);
  v = v - __builtin_convertvector(a, fx2);
  const wg2 b = __builtin_convertvector(v, wg2);
  hi = __builtin_bit_cast(unsigned, a);
  lo = __builtin_bit_cast(unsigned, b);
}
__device__ __forceinline__ float2 wg_join2(unsigned hi, unsigned lo) {   // the two values a (hi, lo) dword pair stands for
  return make_float2(__uint_as_float(hi << 16) + __uint_as_float(lo << 16),
                     __uint_as_float(hi & 0xffff0000u) + __uint_as_float(lo & 0xffff0000u));
}
__device__ __forceinline__ float wg_sum8(const bx8& hi, const bx8& lo, float acc) {   // acc + Σ of the 8 values of a level pair
  const wg2 one = {(__bf16)1.0f, (__bf16)1.0f};
  struct Q { wg2 p[4]; };
  const Q h = __builtin_bit_cast(Q, hi), l = __builtin_bit_cast(Q, lo);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(l.p[i], one, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(h.p[i], one, acc, false);
  }
  return acc;
}
__device__ __forceinline__ void wg_mfma3(f32x4& acc, const bx8& ah, const bx8& al, const bx8& bh, const bx8& bl) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}

template <typename AT, int HALVES = 1, int HALF = 0, bool BX = false, bool WGB = false>
__global__ __launch_bounds__(256, 2) void gemm_chain_bwd_wg_kernel(GemmArgsT<AT> p, ChainArgsT<AT> c, int ntiles, float* wpart,
                                                                   float* glp) {
  static_assert(!WGB || (BX && HALVES == 1), "WGB: the one-launch split-bf16 form");
  constexpr int NACC = 2, HB = 2, HID = 64 * HALVES, N1 = BX ? 3072 : 16 * HB * 64;   // floats of each staged weight block
  constexpr int NTA = BxTerms<AT>::A, NTB = bx_terms_b<AT>(BXPRO_GELU);
  // WGF: GEMM 2's K-groups run inside pass B and its operand split also feeds the planes (fp32 storage; the bf16-storage
  // instantiation has no 32 registers for acc2 across pass B — 2 spilled registers — and splits gz1 a second time instead)
  constexpr bool WGF = WGB && sizeof(AT) == 4;
  constexpr int half = HALF;
  constexpr int hoff = 64 * HALF;             // first hidden row of this launch
  constexpr bool last = HALF == HALVES - 1;   // this launch ends with the LayerNorm backward
  constexpr int kWave = WGB ? 3072 : 48 * kTS;   // floats of one wave's (Bf | T) region (WGB: hi | lo planes of 32 + 16 rows x 128 B)
  extern __shared__ __attribute__((aligned(16))) float fz_lds_cw[];
  float* As1 = fz_lds_cw;
  float* As2 = As1 + N1;
  float* tB = As2 + N1;                       // gamma[32]
  float* red = tB + 32;                       // [4][64]
  float* R = red + 256;                       // 4 wave regions
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int l16 = lane & 15, k4 = lane >> 4;
  float* Bf = R + wave * kWave;
  float* T = Bf + 32 * kTS;
  // WGB planes of this wave (bytes): GH [0, 4096) g2 / x̂ hi, GL [4096, 8192) lo, TH [8192, 10240) the 16-row group hi, TL lo.
  // element (row, voxel v) of a plane: row * 128 + (((v >> 3) ^ (row & 7)) << 4) + (v & 7) * 2
  char* const PL = reinterpret_cast<char*>(Bf);
  constexpr int kGL = 4096, kTH = 8192, kTL = 10240;
  // voxel-owner stores of the pair (2j, 2j+1): rows 2s + h (operand layout) at wop[s & 3] + s * 256; rows (i & 3) + 8 (i >> 2) + 4h
  // (accumulator layout) at wac[i & 3] + (i & 3) * 128 + (i >> 2) * 1024; channel-owner reads of block b, k-step ks at rd[ks] + b * 2048
  unsigned wop[4], wac[4], rd[2];
  if constexpr (WGB) {
    const unsigned ob = (unsigned)h * 128u + ((((unsigned)j >> 2) ^ (unsigned)h) << 4) + ((unsigned)j & 3u) * 4u;
    const unsigned ab = (unsigned)h * 512u + ((((unsigned)j >> 2) ^ (4u * (unsigned)h)) << 4) + ((unsigned)j & 3u) * 4u;
#pragma unroll
    for (int q = 0; q < 4; ++q) { wop[q] = ob ^ ((unsigned)q << 5); wac[q] = ab ^ ((unsigned)q << 4); }
    rd[0] = (unsigned)l16 * 128u + ((((unsigned)k4) ^ ((unsigned)l16 & 7u)) << 4);
    rd[1] = rd[0] ^ 64u;
  }
  const int tiles_per_sample = (int)((p.Ncol + 128 * NACC - 1) / (128 * NACC));
  chain_stagger(c.stagger);

  if constexpr (BX) {
    // 512 operand items of 8 steps each: As1x[g (2)][rb (2)][term][lane], As2x[g (4)][term][lane]
    for (int it = threadIdx.x; it < 512; it += 256) {
      float wv[8];
      const int l = it & 63;
      __bf16* dst;
      if (it < 256) {
        const int rb = (it >> 6) & 1, g = it >> 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = weight_at(p, hoff + rb * 32 + (l & 31), 2 * (8 * g + e) + (l >> 5));
        dst = reinterpret_cast<__bf16*>(As1) + ((g * HB + rb) * NTA * 64 + l) * 8;
      } else {
        const int g = (it - 256) >> 6;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int s2 = 8 * g + e, r = s2 & 15, rb = s2 >> 4;
          const int k = hoff + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), m = l & 31;
          wv[e] = c.wB_t ? c.wB[(int64_t)k * c.ldwB + m] : c.wB[(int64_t)m * c.ldwB + k];
        }
        dst = reinterpret_cast<__bf16*>(As2) + (g * NTA * 64 + l) * 8;
      }
      bx8 t3[NTA];
      bx_split<NTA>(wv, t3);
#pragma unroll
      for (int i = 0; i < NTA; ++i) *reinterpret_cast<bx8*>(dst + i * 64 * 8) = t3[i];
    }
  } else {
  for (int base = threadIdx.x; base < 2 * N1; base += 256 * 8) {
    float tmp[8];
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
      const int idx = base + uu * 256;
      float wv;
      if (idx < N1) {
        const int l = idx & 63, rb = (idx >> 6) % HB, a = idx / (64 * HB);
        wv = weight_at(p, hoff + rb * 32 + (l & 31), 2 * a + (l >> 5));
      } else {
        const int i2 = idx - N1;
        const int l = i2 & 63, s2 = i2 >> 6;
        const int r = s2 & 15, rb = s2 >> 4;
        const int k = hoff + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), m = l & 31;
        wv = c.wB_t ? c.wB[(int64_t)k * c.ldwB + m] : c.wB[(int64_t)m * c.ldwB + k];
      }
      tmp[uu] = wv;
    }
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
      const int idx = base + uu * 256;
      if (idx < N1) As1[idx] = tmp[uu]; else As2[idx - N1] = tmp[uu];
    }
  }
  }
  if (threadIdx.x < 32) tB[threadIdx.x] = p.lnb_g[threadIdx.x];

  f32x4 dW2[2][4], dW1[4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b4 = 0; b4 < 4; ++b4)
#pragma unroll
      for (int v = 0; v < 4; ++v) { dW2[a][b4][v] = 0.f; dW1[b4][a][v] = 0.f; }
  float db2[2] = {0.f, 0.f}, db1[4] = {0.f, 0.f, 0.f, 0.f};
  float gln = 0.f;   // threads 0..63: running (dγ | dβ) sum of this workgroup's tiles, tiles in walking order

  int tile = blockIdx.x;
  float bv[16][NACC];
  auto fetch_tile = [&](int t) {
    const int bt = t / tiles_per_sample;
    const int64_t ct = ((int64_t)(t % tiles_per_sample) * 4 + wave) * (32 * NACC) + NACC * j;
    const unsigned lo = (unsigned)h * (unsigned)p.Ncol + (unsigned)(ct < p.Ncol ? ct : 0);
    const AT* xb = p.x[0] + (int64_t)bt * 32 * p.Ncol;
#pragma unroll
    for (int s = 0; s < 16; ++s) vload<NACC>(xb + (int64_t)(2 * s) * p.Ncol + lo, bv[s]);
  };
  fetch_tile(tile);
  __syncthreads();

  for (; tile < ntiles; tile += gridDim.x) {
    asm volatile("" ::: "memory");   // keep loop-invariant LDS reads out of VGPRs (see gemm_chain_kernel)
    const int b = tile / tiles_per_sample;
    const int64_t col_off = ((int64_t)(tile % tiles_per_sample) * 4 + wave) * (32 * NACC) + NACC * j;
    const bool col_ok = col_off < p.Ncol;
    const int64_t nc = col_ok ? col_off : 0;
    const unsigned lane_row = (unsigned)(4 * h) * (unsigned)p.Ncol + (unsigned)nc;
    const unsigned lane_par = (unsigned)h * (unsigned)p.Ncol + (unsigned)nc;
    const int64_t sample = (int64_t)b * 32 * p.Ncol;

    // Lanes past the last column (ragged last tile only: their loads are clamped to column 0) must contribute zero to the
    // voxel sums.  Zeroing g2 HERE, once, does it for every sum of the tile: gh = W2ᵀ·0 = 0 makes gz1 = gh∘gelu' = 0 (db1, dW1,
    // and through GEMM 2 the LayerNorm sums), g2 = 0 itself covers dW2 and db2 — gelu(z1) and x̂ of such a lane stay finite and
    // meet a zero factor.  (Rounds 2-3 selected on every LDS store instead: 224 v_cndmask per tile, 10 % of the VALU stream.)
    // (hidden 128, two launches: the second half adds to a parked part read at a clamped address, and its register
    // allocation does not survive the change — that form keeps the per-store selects, ZSEL)
    constexpr bool ZSEL = HALVES == 2;
    auto zs = [&](float v) { return (ZSEL && !col_ok) ? 0.f : v; };
    if (!ZSEL && __builtin_amdgcn_ballot_w64(!col_ok) != 0) {
#pragma unroll
      for (int s = 0; s < 16; ++s) { bv[s][0] = col_ok ? bv[s][0] : 0.f; bv[s][1] = col_ok ? bv[s][1] : 0.f; }
    }
    // ---- Bf <- g2 (WGB: the hi / lo planes are written from GEMM 1's own operand split below) ----
    if constexpr (!WGB) {
#pragma unroll
    for (int s = 0; s < 16; ++s)
      *reinterpret_cast<float2*>(Bf + (2 * s + h) * kTS + 2 * j) = make_float2(zs(bv[s][0]), zs(bv[s][1]));
    }

    // Every global operand of the tile is requested one phase AHEAD of its use (two waves per SIMD cannot hide a
    // memory round trip per phase): z1 block g+1 during block g, x1 during the last z1 block, the residual rows before
    // GEMM 2.
    float e[2][8][NACC];
    auto fetch_z1 = [&](int g8, float (&dst)[8][NACC]) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = g8 * 8 + i, rb = rr >> 4, r = rr & 15;
        const int rbase = hoff + rb * 32 + (r & 3) + 8 * (r >> 2);
        vload<NACC>(p.emul + ((int64_t)b * HID + rbase) * p.Ncol + lane_row, dst[i]);
      }
    };
    float xv[2][8][NACC];
    auto fetch_x1 = [&](int hf, float (&dst)[8][NACC]) {
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) vload<NACC>(p.lnb_x + sample + (int64_t)(2 * (hf * 8 + s8)) * p.Ncol + lane_par, dst[s8]);
    };
    const float* sp = p.lnb_stats + (int64_t)b * 2 * p.Ncol;
    float mu[NACC], rs[NACC];
    vload<NACC>(sp + nc, mu);
    vload<NACC>(sp + p.Ncol + nc, rs);
    fetch_z1(0, e[0]);

    // ---- GEMM 1: gh = W2ᵀ g2 ----
    f32x16 acc1[HB][NACC];
#pragma unroll
    for (int rb = 0; rb < HB; ++rb)
#pragma unroll
      for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[rb][q][r] = 0.f;
    if constexpr (BX) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        bx8 bop[NACC][NTB];
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
          float x8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x8[e] = bv[8 * g + e][q];
          bx_split<NTB>(x8, bop[q]);
        }
        if constexpr (WGB) {   // levels 0 and 1 of g2, as (voxel 2j, voxel 2j+1) pairs of channel 2s + h, into the planes
          static_assert(!WGB || NTB >= 2, "two levels of the column operand");
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int s = 8 * g + e;
            const wg2 ph = {bop[0][0][e], bop[1][0][e]};
            const wg2 pl = {bop[0][NTB >= 2 ? 1 : 0][e], bop[1][NTB >= 2 ? 1 : 0][e]};
            *reinterpret_cast<wg2*>(PL + wop[s & 3] + s * 256) = ph;
            *reinterpret_cast<wg2*>(PL + kGL + wop[s & 3] + s * 256) = pl;
          }
        }
#pragma unroll
        for (int rb = 0; rb < HB; ++rb) {
          bx8 aop[NTA];
#pragma unroll
          for (int i = 0; i < NTA; ++i)
            aop[i] = *reinterpret_cast<const bx8*>(reinterpret_cast<const __bf16*>(As1) + (((g * HB + rb) * NTA + i) * 64 + lane) * 8);
#pragma unroll
          for (int q = 0; q < NACC; ++q) bx_mfma<NTA, NTB>(acc1[rb][q], aop, bop[q]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int rb = 0; rb < HB; ++rb) {
        const float av = As1[(s * HB + rb) * 64 + lane];
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc1[rb][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[s][q], acc1[rb][q], 0, 0, 0);
        if (rb == HB - 1 && (s & 1) == 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
    // (the next tile's operand is requested after GEMM 2: its 32 registers would otherwise be live next to the 64
    // gz1 accumulators and the 64 weight-gradient accumulators; the epilogue and the other resident waves cover
    // the round trip)

    // ---- pass A: 16 hidden channels at a time: gz1 = gh ∘ gelu'(z1) (kept in acc1); gelu(z1) -> T; dW2 += g2 ⊗ gelu(z1) ----
#pragma unroll
    for (int g8 = 0; g8 < 4; ++g8) {
      // (compiler-only fence: the g2 operand reads below are the same for every group — left alone they are read once
      // and kept in 32 registers across all four)
      asm volatile("" ::: "memory");
      if (g8 < 3) fetch_z1(g8 + 1, e[(g8 + 1) & 1]);
      else { fetch_x1(0, xv[0]); fetch_x1(1, xv[1]); }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = g8 * 8 + i, rb = rr >> 4, r = rr & 15;
        float gl[NACC], dg[NACC];
        gelu_both2<HALVES == 2>(e[g8 & 1][i], gl, dg);
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
          float gz = acc1[rb][q][r] * dg[q];
          // pin the product HERE: its only readers are pass B and GEMM 2, and the optimiser otherwise sinks the
          // gelu' evaluation (and with it the liveness of all 64 z1 values) down to them
          asm volatile("" : "+v"(gz));
          acc1[rb][q][r] = gz;
        }
        if constexpr (WGB) {
          unsigned ph, pl;
          wg_split2(gl[0], gl[1], ph, pl);
          *reinterpret_cast<unsigned*>(PL + kTH + wac[i & 3] + (i & 3) * 128 + (i >> 2) * 1024) = ph;
          *reinterpret_cast<unsigned*>(PL + kTL + wac[i & 3] + (i & 3) * 128 + (i >> 2) * 1024) = pl;
        } else {
        const int loc = (i & 3) + 8 * (i >> 2) + 4 * h;
        *reinterpret_cast<float2*>(T + loc * kTS + 2 * j) = make_float2(zs(gl[0]), zs(gl[1]));
        }
      }
      if constexpr (WGB) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {   // 32 voxels per MFMA: six 16-byte operands in flight, then 6 MFMAs
          const bx8 bh = *reinterpret_cast<const bx8*>(PL + kTH + rd[ks]);
          const bx8 bl = *reinterpret_cast<const bx8*>(PL + kTL + rd[ks]);
          const bx8 a0h = *reinterpret_cast<const bx8*>(PL + rd[ks]);
          const bx8 a0l = *reinterpret_cast<const bx8*>(PL + kGL + rd[ks]);
          const bx8 a1h = *reinterpret_cast<const bx8*>(PL + 2048 + rd[ks]);
          const bx8 a1l = *reinterpret_cast<const bx8*>(PL + kGL + 2048 + rd[ks]);
          wg_mfma3(dW2[0][g8], a0h, a0l, bh, bl);
          wg_mfma3(dW2[1][g8], a1h, a1l, bh, bl);
          if (g8 == 0) {   // db2 = Σ_v g2 from the operands of the first group
            db2[0] = wg_sum8(a0h, a0l, db2[0]);
            db2[1] = wg_sum8(a1h, a1l, db2[1]);
            asm volatile("" : "+v"(db2[0]), "+v"(db2[1]));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {   // 8 voxel quads at a time: 24 LDS operands in flight, then 16 MFMAs
        float bq[8], a0[8], a1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = tc * 8 + u;
          bq[u] = T[l16 * kTS + 4 * t + k4];
          a0[u] = Bf[l16 * kTS + 4 * t + k4];
          a1[u] = Bf[(16 + l16) * kTS + 4 * t + k4];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          dW2[0][g8] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], bq[u], dW2[0][g8], 0, 0, 0);
          dW2[1][g8] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], bq[u], dW2[1][g8], 0, 0, 0);
        }
        if (g8 == 0 && (HALVES == 1 || half == 0)) {   // db2 = Σ_v g2 from the operands of the first group (pinned: the optimiser otherwise postpones
                         // the sums — and keeps the operands alive — to the end of the tile)
          db2[0] += ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a0[4] + a0[5]) + (a0[6] + a0[7]));
          db2[1] += ((a1[0] + a1[1]) + (a1[2] + a1[3])) + ((a1[4] + a1[5]) + (a1[6] + a1[7]));
          asm volatile("" : "+v"(db2[0]), "+v"(db2[1]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // the residual rows g2 in the ACCUMULATOR layout, read back from Bf before x̂ replaces it there (rounds 2-3 re-read them
    // from global memory before GEMM 2: 0.54 GB per launch that did not hit the caches — PMC traffic 1.22x algorithmic)
    // (BX form only: the fp32-MFMA form of the kernel — fz_gemm_bx_enable(0), diagnostics — has no 32 registers to spare
    // across pass B and keeps the global re-read)
    float ga[2][8][NACC];
    if (last && BX) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float2 gv;
        if constexpr (WGB)
          gv = wg_join2(*reinterpret_cast<const unsigned*>(PL + wac[r & 3] + (r & 3) * 128 + (r >> 2) * 1024),
                        *reinterpret_cast<const unsigned*>(PL + kGL + wac[r & 3] + (r & 3) * 128 + (r >> 2) * 1024));
        else
          gv = *reinterpret_cast<const float2*>(Bf + ((r & 3) + 8 * (r >> 2) + 4 * h) * kTS + 2 * j);
        ga[r >> 3][r & 7][0] = gv.x; ga[r >> 3][r & 7][1] = gv.y;
      }
      asm volatile("" ::: "memory");   // (the reads must stay ahead of the x̂ stores below: same addresses)
    }

    // ---- Bf <- LN-normalised x1 (requested during the last z1 block) ----
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        if constexpr (WGB) {
          const int s = hf * 8 + s8;
          unsigned ph, pl;
          wg_split2((xv[hf][s8][0] - mu[0]) * rs[0], (xv[hf][s8][1] - mu[1]) * rs[1], ph, pl);
          *reinterpret_cast<unsigned*>(PL + wop[s & 3] + s * 256) = ph;
          *reinterpret_cast<unsigned*>(PL + kGL + wop[s & 3] + s * 256) = pl;
        } else
        *reinterpret_cast<float2*>(Bf + (2 * (hf * 8 + s8) + h) * kTS + 2 * j) =
            make_float2(zs((xv[hf][s8][0] - mu[0]) * rs[0]), zs((xv[hf][s8][1] - mu[1]) * rs[1]));
      }

    f32x16 acc2[NACC];
    if constexpr (WGF) {
#pragma unroll
      for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[q][r] = 0.f;
    }
    // ---- pass B: gz1 block -> T; S1 += gz1 ⊗ x̂, db1 += Σ gz1 (WGB: and GEMM 2's K-group of the same channels) ----
#pragma unroll
    for (int g8 = 0; g8 < 4; ++g8) {
      asm volatile("" ::: "memory");   // as in pass A: re-read the x̂ operands per group instead of holding 32 registers
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = g8 * 8 + i, rb = rr >> 4, r = rr & 15;
        if constexpr (WGF) {
          // (the two levels come from GEMM 2's own three-level split of this group, below)
        } else if constexpr (WGB) {
          unsigned ph, pl;
          wg_split2(acc1[rb][0][r], acc1[rb][1][r], ph, pl);
          *reinterpret_cast<unsigned*>(PL + kTH + wac[i & 3] + (i & 3) * 128 + (i >> 2) * 1024) = ph;
          *reinterpret_cast<unsigned*>(PL + kTL + wac[i & 3] + (i & 3) * 128 + (i >> 2) * 1024) = pl;
        } else {
        const int loc = (i & 3) + 8 * (i >> 2) + 4 * h;
        *reinterpret_cast<float2*>(T + loc * kTS + 2 * j) = make_float2(zs(acc1[rb][0][r]), zs(acc1[rb][1][r]));
        }
      }
      if constexpr (WGB) {
        // K-group g8 of GEMM 2 (gl += W1ᵀ gz1) IS this group of hidden channels: split it once, multiply, and park levels 0 / 1
        if constexpr (WGF) {
          bx8 aop[NTA];
#pragma unroll
          for (int i = 0; i < NTA; ++i)
            aop[i] = *reinterpret_cast<const bx8*>(reinterpret_cast<const __bf16*>(As2) + ((g8 * NTA + i) * 64 + lane) * 8);
          bx8 bop[NACC][NTB];
#pragma unroll
          for (int q = 0; q < NACC; ++q) {
            float x8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x8[e] = acc1[g8 >> 1][q][8 * (g8 & 1) + e];
            bx_split<NTB>(x8, bop[q]);
            bx_mfma<NTA, NTB>(acc2[q], aop, bop[q]);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const wg2 ph = {bop[0][0][i], bop[1][0][i]};
            const wg2 pl = {bop[0][1][i], bop[1][1][i]};
            *reinterpret_cast<wg2*>(PL + kTH + wac[i & 3] + (i & 3) * 128 + (i >> 2) * 1024) = ph;
            *reinterpret_cast<wg2*>(PL + kTL + wac[i & 3] + (i & 3) * 128 + (i >> 2) * 1024) = pl;
          }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bx8 ah = *reinterpret_cast<const bx8*>(PL + kTH + rd[ks]);
          const bx8 al = *reinterpret_cast<const bx8*>(PL + kTL + rd[ks]);
          const bx8 b0h = *reinterpret_cast<const bx8*>(PL + rd[ks]);
          const bx8 b0l = *reinterpret_cast<const bx8*>(PL + kGL + rd[ks]);
          const bx8 b1h = *reinterpret_cast<const bx8*>(PL + 2048 + rd[ks]);
          const bx8 b1l = *reinterpret_cast<const bx8*>(PL + kGL + 2048 + rd[ks]);
          wg_mfma3(dW1[g8][0], ah, al, b0h, b0l);
          wg_mfma3(dW1[g8][1], ah, al, b1h, b1l);
          db1[g8] = wg_sum8(ah, al, db1[g8]);
          asm volatile("" : "+v"(db1[g8]));
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {
        float aq[8], b0[8], b1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = tc * 8 + u;
          aq[u] = T[l16 * kTS + 4 * t + k4];
          b0[u] = Bf[l16 * kTS + 4 * t + k4];
          b1[u] = Bf[(16 + l16) * kTS + 4 * t + k4];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          dW1[g8][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[u], b0[u], dW1[g8][0], 0, 0, 0);
          dW1[g8][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[u], b1[u], dW1[g8][1], 0, 0, 0);
        }
        db1[g8] += ((aq[0] + aq[1]) + (aq[2] + aq[3])) + ((aq[4] + aq[5]) + (aq[6] + aq[7]));
        asm volatile("" : "+v"(db1[g8]));
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      __builtin_amdgcn_sched_barrier(0);
    }


    if (last && !BX) {   // residual rows (g2 again: L2 / MALL), requested before GEMM 2
#pragma unroll
      for (int r = 0; r < 16; ++r)
        vload<NACC>(p.lnb_gadd + sample + (int64_t)((r & 3) + 8 * (r >> 2)) * p.Ncol + lane_row, ga[r >> 3][r & 7]);
    }

    // ---- GEMM 2: gl = W1ᵀ gz1 straight from the accumulators (second half: on top of the first half's part) ----
    if constexpr (WGF) {
      // (done inside pass B)
    } else if (HALVES == 2 && half == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v[NACC];
        vload<NACC>(glp + sample + (int64_t)((r & 3) + 8 * (r >> 2)) * p.Ncol + lane_row, v);
        acc2[0][r] = v[0]; acc2[1][r] = v[1];
      }
    } else {
#pragma unroll
      for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[q][r] = 0.f;
    }
    if constexpr (WGF) {
    } else if constexpr (BX) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {   // steps (rb, r) = (g >> 1, 8 (g & 1) + e): accumulator registers as the column operand
        bx8 aop[NTA];
#pragma unroll
        for (int i = 0; i < NTA; ++i)
          aop[i] = *reinterpret_cast<const bx8*>(reinterpret_cast<const __bf16*>(As2) + ((g * NTA + i) * 64 + lane) * 8);
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
          float x8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x8[e] = acc1[g >> 1][q][8 * (g & 1) + e];
          bx8 bop[NTB];
          bx_split<NTB>(x8, bop);
          bx_mfma<NTA, NTB>(acc2[q], aop, bop);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int rb = 0; rb < HB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float av = As2[(rb * 16 + r) * 64 + lane];
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc2[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, acc1[rb][q][r], acc2[q], 0, 0, 0);
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    fetch_tile(tile + gridDim.x < ntiles ? tile + gridDim.x : tile);

    if (HALVES == 2 && !last) {   // first half: park the partial W1ᵀ·gz1 (fp32), no epilogue
      if (col_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v[NACC] = {acc2[0][r], acc2[1][r]};
          vstore<NACC>(glp + sample + (int64_t)((r & 3) + 8 * (r >> 2)) * p.Ncol + lane_row, v);
        }
      }
      continue;
    }
    // ---- LayerNorm backward + residual gradient (x̂ from Bf in the accumulator layout, g2 re-read: L2 / MALL) ----
    float m1[NACC] = {0.f, 0.f}, m2[NACC] = {0.f, 0.f};
    float xkeep[WGB ? 16 : 1][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      const float gc = tB[row];
      float2 xh;
      if constexpr (WGB) {   // (rebuilt from its two levels once and kept: the registers of gz1 are free by now)
        xh = wg_join2(*reinterpret_cast<const unsigned*>(PL + wac[r & 3] + (r & 3) * 128 + (r >> 2) * 1024),
                      *reinterpret_cast<const unsigned*>(PL + kGL + wac[r & 3] + (r & 3) * 128 + (r >> 2) * 1024));
        xkeep[r][0] = xh.x; xkeep[r][1] = xh.y;
      } else
        xh = *reinterpret_cast<const float2*>(Bf + row * kTS + 2 * j);
      const float a0 = acc2[0][r] * gc, a1 = acc2[1][r] * gc;
      m1[0] += a0; m1[1] += a1;
      m2[0] += a0 * xh.x; m2[1] += a1 * xh.y;
    }
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
      m1[q] = (m1[q] + __shfl_xor(m1[q], 32, 64)) * (1.0f / 32.0f);
      m2[q] = (m2[q] + __shfl_xor(m2[q], 32, 64)) * (1.0f / 32.0f);
    }
#pragma unroll
    for (int r8 = 0; r8 < 2; ++r8) {
      float sgv[8], sbv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = r8 * 8 + i;
        const int rbase = (r & 3) + 8 * (r >> 2);
        const int row = rbase + 4 * h;
        const float gc = tB[row];
        float2 xh;
        if constexpr (WGB)
          xh = make_float2(xkeep[r][0], xkeep[r][1]);
        else
          xh = *reinterpret_cast<const float2*>(Bf + row * kTS + 2 * j);
        float v[NACC];
        v[0] = rs[0] * (acc2[0][r] * gc - m1[0] - xh.x * m2[0]) + ga[r8][i][0];
        v[1] = rs[1] * (acc2[1][r] * gc - m1[1] - xh.y * m2[1]) + ga[r8][i][1];
        if (col_ok) vstore<NACC>(p.y + sample + (int64_t)rbase * p.Ncol + lane_row, v);
        float sg = acc2[0][r] * xh.x + acc2[1][r] * xh.y;   // (lanes past the last column: acc2 = 0, see the top of the tile)
        float sb = acc2[0][r] + acc2[1][r];
        sg = zs(sg);
        sb = zs(sb);
        if constexpr (WGF) {   // the eight rows of the block are reduced together below (multi-value butterfly: 19 operations for
          sgv[i] = sg;         // sixteen half-wave sums instead of 5 per sum; fp32 storage: the bf16 instantiation spills with it)
          sbv[i] = sb;
        } else {
        sg = half_sum32(sg);
        sb = half_sum32(sb);
        if ((lane & 31) == 31) {
          red[wave * 64 + row] = sg;
          red[wave * 64 + 32 + row] = sb;
        }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (WGF) {
        const float tg = half_sum8_dist(sgv, lane), tb = half_sum8_dist(sbv, lane);   // 4-lane group i of a half holds row i's total
        const int gi = (lane >> 2) & 7, rr = r8 * 8 + gi;
        if ((lane & 3) == 0) {
          const int rw = (rr & 3) + 8 * (rr >> 2) + 4 * h;
          red[wave * 64 + rw] = tg;
          red[wave * 64 + 32 + rw] = tb;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int e = threadIdx.x;
      gln += (red[e] + red[64 + e]) + (red[128 + e] + red[192 + e]);
    }
    __syncthreads();
  }

  // ---- the workgroup's (dW2 | S1 | db2 | db1 | dγ | dβ) row: add the four waves through LDS, waves in index order ----
  float* row = wpart + (int64_t)blockIdx.x * kWgRow;
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int v = 0; v < 4; ++v) R[(wave * 32 + (a * 4 + cb) * 4 + v) * 64 + lane] = dW2[a][cb][v];
  __syncthreads();
  for (int e = threadIdx.x; e < 2048; e += 256) {
    const int idx = e >> 6, l = e & 63;
    const int a = idx >> 4, cb = (idx >> 2) & 3, v = idx & 3;
    const float t = (R[e] + R[2048 + e]) + (R[4096 + e] + R[6144 + e]);
    row[(16 * a + 4 * (l >> 4) + v) * 64 + 16 * cb + (l & 15)] = t;
  }
  __syncthreads();
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int v = 0; v < 4; ++v) R[(wave * 32 + (cb * 2 + kh) * 4 + v) * 64 + lane] = dW1[cb][kh][v];
  __syncthreads();
  for (int e = threadIdx.x; e < 2048; e += 256) {
    const int idx = e >> 6, l = e & 63;
    const int cb = idx >> 3, kh = (idx >> 2) & 1, v = idx & 3;
    const float t = (R[e] + R[2048 + e]) + (R[4096 + e] + R[6144 + e]);
    row[2048 + (16 * cb + 4 * (l >> 4) + v) * 32 + 16 * kh + (l & 15)] = t;
  }
  __syncthreads();
  R[(wave * 6 + 0) * 64 + lane] = db2[0];
  R[(wave * 6 + 1) * 64 + lane] = db2[1];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) R[(wave * 6 + 2 + cb) * 64 + lane] = db1[cb];
  __syncthreads();
  if (threadIdx.x < 96) {
    const int e = threadIdx.x, slot = e >> 4, i16 = e & 15;   // slots 0,1: db2 halves; 2..5: db1 blocks
    float t = 0.f;
    for (int w = 0; w < 4; ++w)
      for (int kk = 0; kk < 4; ++kk) t += R[(w * 6 + slot) * 64 + kk * 16 + i16];
    row[4096 + e] = t;
  }
  if (threadIdx.x < 64) row[4096 + 96 + threadIdx.x] = gln;
}

// (the rows are added, and the LayerNorm affine applied to dW1, by the FK_CHAIN_WG job of the finish kernel: finish.h)

// =================================================================================================
// Input gradient AND weight gradient of a 32 -> 32 layer in one pass (C = 32: in_proj behind LayerNorm, out_proj):
//   y  = Wᵀ g                                   (LNB: then the LayerNorm backward on the accumulators, + gadd)
//   dW = Σ_v g[m][v] · q[k][v],  db = Σ_v g     (LNB: q = LN-normalised x; the affine is applied by the finish kernel)
// Unfused, the weight-gradient launch reads g and q a second time (2 of the 6 resp. 4 plane-sets of the pair).  Both
// MFMA operands of dW come straight from memory in [channel][voxel] order, so each wave parks its g tile and its q
// tile in LDS (stride kTS) and reads them back with the channel on the lane axis — no register transposes.
// Persistent workgroups (two per CU), sums carried in registers across tiles, one wpart row per workgroup.
// =================================================================================================
// (kDwRow — floats of one wpart row: dW [32][32] | db [32] | dγ [32] | dβ [32] (LNB) — lives in finish.h)

template <typename AT>
struct DwArgsT {
  const AT* g;        // (B, 32, V) gradient of the layer output
  const AT* q;        // (B, 32, V) layer input (LNB: the LayerNorm input)
  const float* w;     // (32, 32) forward weight W[m][k], row stride ldw
  int ldw;
  const float* stats; // LNB: (B, 2, V)
  const float* ln_g;  // LNB: gamma
  const AT* gadd;     // LNB: (B, 32, V) added to y, or null
  AT* y;              // (B, 32, V)
  float* wpart;       // [gridDim.x][kDwRow]
  int64_t V;
  int B;
};

template <bool LNB, typename AT>
__global__ __launch_bounds__(256, 2) void gemm_dw_kernel(DwArgsT<AT> p, int ntiles) {
  constexpr int NACC = 2;
  constexpr int kWave = 64 * kTS;             // floats of one wave's (Gb | Qb) region
  // BXB (bf16 storage): both GEMMs on the bf16 matrix pipe with fp32-accurate products — the activations are exact bf16
  // terms, the weights and the LayerNorm-normalised input three levels (gemm_bx.h): 12 + 4 (LNB: 12) MFMAs of 32 cycles per
  // tile instead of 32 + 64 fp32 MFMAs; with half the bytes this kernel is matrix-pipe bound otherwise.  fp32 storage: it
  // is HBM-bound on the fp32 MFMAs and keeps them.
  constexpr bool BXB = sizeof(AT) == 2;
  extern __shared__ __attribute__((aligned(16))) float fz_lds_dw[];
  float* As = fz_lds_dw;                      // [16][64] operand order: A[m][k] = W[k][m]  (BXB: [2 groups][3 levels][64] x 16 B)
  float* tB = As + 1536;                      // gamma[32]
  float* red = tB + 32;                       // [4][64]
  float* R = red + 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int l16 = lane & 15, k4 = lane >> 4;
  float* Gb = R + wave * kWave;
  float* Qb = Gb + 32 * kTS;
  const int tiles_per_sample = (int)((p.V + 128 * NACC - 1) / (128 * NACC));

  if constexpr (BXB) {
    if (threadIdx.x < 128) {   // item (group g, lane l): the eight steps 8g + e of the fp32 form, split in three levels
      const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
      float a8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) a8[e] = p.w[(int64_t)(2 * (8 * g + e) + (l >> 5)) * p.ldw + (l & 31)];
      bx8 t3[3];
      bx_split<3>(a8, t3);
      bx8* dst = reinterpret_cast<bx8*>(As) + (g * 3) * 64 + l;
      dst[0] = t3[0]; dst[64] = t3[1]; dst[128] = t3[2];
    }
  } else {
  for (int idx = threadIdx.x; idx < 1024; idx += 256) {
    const int l = idx & 63, a = idx >> 6;
    As[idx] = p.w[(int64_t)(2 * a + (l >> 5)) * p.ldw + (l & 31)];   // A[m = l&31][k = 2a + h] = W[k][m]
  }
  }
  if (LNB && threadIdx.x < 32) tB[threadIdx.x] = p.ln_g[threadIdx.x];

  f32x4 dW[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
      for (int v = 0; v < 4; ++v) dW[a][b2][v] = 0.f;
  f32x16 dWx;                                   // BXB: the 32 x 32 sum as ONE accumulator tile (row = g channel, column = q channel)
#pragma unroll
  for (int r = 0; r < 16; ++r) dWx[r] = 0.f;
  float db[2] = {0.f, 0.f};
  float gln = 0.f;   // threads 0..63 (LNB): running (dγ | dβ) sum of this workgroup's tiles

  int tile = blockIdx.x;
  float bv[16][NACC];
  auto fetch_tile = [&](int t) {
    const int bt = t / tiles_per_sample;
    const int64_t ct = ((int64_t)(t % tiles_per_sample) * 4 + wave) * (32 * NACC) + NACC * j;
    const unsigned lo = (unsigned)h * (unsigned)p.V + (unsigned)(ct < p.V ? ct : 0);
    const AT* xb = p.g + (int64_t)bt * 32 * p.V;
#pragma unroll
    for (int s = 0; s < 16; ++s) vload<NACC>(xb + (int64_t)(2 * s) * p.V + lo, bv[s]);
  };
  fetch_tile(tile);
  __syncthreads();

  for (; tile < ntiles; tile += gridDim.x) {
    asm volatile("" ::: "memory");
    const int b = tile / tiles_per_sample;
    const int64_t col_off = ((int64_t)(tile % tiles_per_sample) * 4 + wave) * (32 * NACC) + NACC * j;
    const bool col_ok = col_off < p.V;
    const int64_t nc = col_ok ? col_off : 0;
    const unsigned lane_row = (unsigned)(4 * h) * (unsigned)p.V + (unsigned)nc;
    const unsigned lane_par = (unsigned)h * (unsigned)p.V + (unsigned)nc;
    const int64_t sample = (int64_t)b * 32 * p.V;

    // ---- q tile -> Qb (LNB: normalised), in two halves of 8 loads ----
    float mu[NACC] = {0.f, 0.f}, rs[NACC] = {1.f, 1.f};
    if (LNB) {
      const float* sp = p.stats + (int64_t)b * 2 * p.V;
      vload<NACC>(sp + nc, mu);
      vload<NACC>(sp + p.V + nc, rs);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float xv[8][NACC];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) vload<NACC>(p.q + sample + (int64_t)(2 * (hf * 8 + s8)) * p.V + lane_par, xv[s8]);
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8)
        *reinterpret_cast<float2*>(Qb + (2 * (hf * 8 + s8) + h) * kTS + 2 * j) =
            make_float2(col_ok ? (xv[s8][0] - mu[0]) * rs[0] : 0.f, col_ok ? (xv[s8][1] - mu[1]) * rs[1] : 0.f);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- g tile -> Gb ----
#pragma unroll
    for (int s = 0; s < 16; ++s)
      *reinterpret_cast<float2*>(Gb + (2 * s + h) * kTS + 2 * j) = make_float2(col_ok ? bv[s][0] : 0.f, col_ok ? bv[s][1] : 0.f);

    // ---- y = Wᵀ g ----
    f32x16 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    if constexpr (BXB) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        bx8 bop[NACC][1];
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
          float x8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x8[e] = bv[8 * g + e][q];
          bx_split<1>(x8, bop[q]);               // bf16 storage: exact
        }
#pragma unroll
        for (int t = 2; t >= 0; --t) {
          const bx8 aw = reinterpret_cast<const bx8*>(As)[(g * 3 + t) * 64 + lane];
#pragma unroll
          for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw, bop[q][0], acc[q], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float av = As[s * 64 + lane];
#pragma unroll
      for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[s][q], acc[q], 0, 0, 0);
    }
    }
    fetch_tile(tile + gridDim.x < ntiles ? tile + gridDim.x : tile);

    // ---- dW += g ⊗ q, db += Σ g ----
    if constexpr (BXB) {
      // operand element e of lane half h4 = voxel 16 gk + 8 h4 + e of the wave tile, for both operands (any assignment of
      // reduction indices to slots is valid); rows are 8-byte aligned (stride 66 floats): four ds_read_b64 per operand
      const int rowo = (lane & 31) * kTS + 8 * (lane >> 5);
#pragma unroll
      for (int gk = 0; gk < 4; ++gk) {
        float g8[8], q8[8];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          const float2 gv = *reinterpret_cast<const float2*>(Gb + rowo + 16 * gk + 2 * e2);
          const float2 qv = *reinterpret_cast<const float2*>(Qb + rowo + 16 * gk + 2 * e2);
          g8[2 * e2] = gv.x; g8[2 * e2 + 1] = gv.y;
          q8[2 * e2] = qv.x; q8[2 * e2 + 1] = qv.y;
        }
        bx8 ga[1];
        bx_split<1>(g8, ga);                     // the stored gradient: exact
        constexpr int NTQ = LNB ? 3 : 1;         // LNB: the normalised input is a computed fp32 value
        bx8 qb[NTQ];
        bx_split<NTQ>(q8, qb);
#pragma unroll
        for (int t = NTQ - 1; t >= 0; --t) dWx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[0], qb[t], dWx, 0, 0, 0);
        db[0] += ((g8[0] + g8[1]) + (g8[2] + g8[3])) + ((g8[4] + g8[5]) + (g8[6] + g8[7]));
        asm volatile("" : "+v"(db[0]));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else
#pragma unroll
    for (int tc = 0; tc < 2; ++tc) {
      float a0[8], a1[8], b0[8], b1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = tc * 8 + u;
        a0[u] = Gb[l16 * kTS + 4 * t + k4];
        a1[u] = Gb[(16 + l16) * kTS + 4 * t + k4];
        b0[u] = Qb[l16 * kTS + 4 * t + k4];
        b1[u] = Qb[(16 + l16) * kTS + 4 * t + k4];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        dW[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b0[u], dW[0][0], 0, 0, 0);
        dW[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b1[u], dW[0][1], 0, 0, 0);
        dW[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], b0[u], dW[1][0], 0, 0, 0);
        dW[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], b1[u], dW[1][1], 0, 0, 0);
      }
      db[0] += ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a0[4] + a0[5]) + (a0[6] + a0[7]));
      db[1] += ((a1[0] + a1[1]) + (a1[2] + a1[3])) + ((a1[4] + a1[5]) + (a1[6] + a1[7]));
      asm volatile("" : "+v"(db[0]), "+v"(db[1]));
      __builtin_amdgcn_sched_barrier(0);
    }

    if (!LNB) {
      if (col_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rbase = (r & 3) + 8 * (r >> 2);
          float v[NACC] = {acc[0][r], acc[1][r]};
          vstore<NACC>(p.y + sample + (int64_t)rbase * p.V + lane_row, v);
        }
      }
    } else {
      float m1[NACC] = {0.f, 0.f}, m2[NACC] = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float gc = tB[row];
        const float2 xh = *reinterpret_cast<const float2*>(Qb + row * kTS + 2 * j);
        const float a0 = acc[0][r] * gc, a1 = acc[1][r] * gc;
        m1[0] += a0; m1[1] += a1;
        m2[0] += a0 * xh.x; m2[1] += a1 * xh.y;
      }
#pragma unroll
      for (int q = 0; q < NACC; ++q) {
        m1[q] = (m1[q] + __shfl_xor(m1[q], 32, 64)) * (1.0f / 32.0f);
        m2[q] = (m2[q] + __shfl_xor(m2[q], 32, 64)) * (1.0f / 32.0f);
      }
#pragma unroll
      for (int r8 = 0; r8 < 2; ++r8) {
        float ga[8][NACC];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = r8 * 8 + i;
          if (p.gadd != nullptr) vload<NACC>(p.gadd + sample + (int64_t)((r & 3) + 8 * (r >> 2)) * p.V + lane_row, ga[i]);
          else ga[i][0] = ga[i][1] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = r8 * 8 + i;
          const int rbase = (r & 3) + 8 * (r >> 2);
          const int row = rbase + 4 * h;
          const float gc = tB[row];
          const float2 xh = *reinterpret_cast<const float2*>(Qb + row * kTS + 2 * j);
          float v[NACC];
          v[0] = rs[0] * (acc[0][r] * gc - m1[0] - xh.x * m2[0]) + ga[i][0];
          v[1] = rs[1] * (acc[1][r] * gc - m1[1] - xh.y * m2[1]) + ga[i][1];
          if (col_ok) vstore<NACC>(p.y + sample + (int64_t)rbase * p.V + lane_row, v);
          float sg = col_ok ? acc[0][r] * xh.x + acc[1][r] * xh.y : 0.f;
          float sb = col_ok ? acc[0][r] + acc[1][r] : 0.f;
          sg = half_sum32(sg);
          sb = half_sum32(sb);
          if ((lane & 31) == 31) {
            red[wave * 64 + row] = sg;
            red[wave * 64 + 32 + row] = sb;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
      if (threadIdx.x < 64) {
        const int e = threadIdx.x;
        gln += (red[e] + red[64 + e]) + (red[128 + e] + red[192 + e]);
      }
      __syncthreads();
    }
  }

  // ---- workgroup row: (dW | db), waves added in index order ----
  float* row = p.wpart + (int64_t)blockIdx.x * kDwRow;
  __syncthreads();
  if constexpr (BXB) {
#pragma unroll
    for (int r = 0; r < 16; ++r) R[(wave * 16 + r) * 64 + lane] = dWx[r];
  } else {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
      for (int v = 0; v < 4; ++v) R[(wave * 16 + (a * 2 + b2) * 4 + v) * 64 + lane] = dW[a][b2][v];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int idx = e >> 6, l = e & 63;
    const float t = (R[e] + R[1024 + e]) + (R[2048 + e] + R[3072 + e]);
    if constexpr (BXB) {   // accumulator register idx of lane l: row (g channel) = (idx & 3) + 8 (idx >> 2) + 4 (l >> 5), column = l & 31
      row[((idx & 3) + 8 * (idx >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = t;
    } else {
      const int a = idx >> 3, b2 = (idx >> 2) & 1, v = idx & 3;
      row[(16 * a + 4 * (l >> 4) + v) * 32 + 16 * b2 + (l & 15)] = t;
    }
  }
  __syncthreads();
  R[(wave * 2 + 0) * 64 + lane] = db[0];
  R[(wave * 2 + 1) * 64 + lane] = db[1];
  __syncthreads();
  if (threadIdx.x < 32) {
    const int e = threadIdx.x, slot = e >> 4, i16 = e & 15;
    float t = 0.f;
    if constexpr (BXB) {   // db[0] of lane l = partial sum of g channel l & 31 over its half of the voxels
      for (int w = 0; w < 4; ++w) t += R[(w * 2) * 64 + e] + R[(w * 2) * 64 + 32 + e];
    } else {
      for (int w = 0; w < 4; ++w)
        for (int kk = 0; kk < 4; ++kk) t += R[(w * 2 + slot) * 64 + kk * 16 + i16];
    }
    row[1024 + e] = t;
  }
  if (threadIdx.x < 64) row[1024 + 32 + threadIdx.x] = gln;
}

// (rows added in slice order, LayerNorm affine applied, by the FK_DW job of the finish kernel: finish.h)

// =================================================================================================
// Kernel B — streaming operand with a PF-deep register prefetch ring, any K, all loaders.
// NACC = consecutive voxels per lane (4/2/1 → 128/64/32-column wave tiles): small tiles give the
// deep, narrow stages (8^3..32^3 voxels, C = 128..512) enough workgroups to fill 256 CUs.
// =================================================================================================
constexpr int kAChunk = 64;  // A-operand steps staged in LDS at a time

// PRO = compile-time prologue: a runtime branch inside the K loop splits every step into its own
// basic block (ds_read → wait → MFMA serialised), so the variants are separate instantiations.
enum { PRO_NONE = 0, PRO_LN = 1, PRO_GELU = 2, PRO_BMUL = 3 };

// KS = 4: the four waves of a workgroup share ONE column tile and split the K steps between them
// (groups of kPF steps, round-robin), then add their accumulators through LDS.  For the deep
// stages (8^3, 16^3 voxels; K = 256..2048) this gives 4x the workgroups and 4x shorter dependent
// MFMA chains: 512->512 at 8^3 is 128 workgroups x 256 serial steps without it.
// (2 workgroups per CU: 3 or 4 — narrower tiles under tighter launch bounds — measured no faster,
// an occupancy sweep: the operand traffic of these launches runs at 4.1-5.1 TB/s even with the
// MFMAs compiled out (round-1/2 probe `gemm_probe7`), the fp32 MFMA time comes largely on top of it.)
template <int MB, int NACC, int LOADER, int EPI, int PRO, int KS = 1, typename AT = float>
__global__ __launch_bounds__(256, ((MB == 2 && NACC == 4 && PRO == 3 /* gate operand in the ring */) ? 1 : 2)) void gemm_stream_kernel(GemmArgsT<AT> p) {
  constexpr int TN = 32 * NACC;
  constexpr int NL = (LOADER == LOAD_S2D) ? 4 : NACC;  // floats fetched per load step
  // operand prefetch depth (load steps): narrow tiles are latency-bound (L2 round trip ≈ 500-900
  // cycles vs 64·NACC MFMA cycles per step), so they keep more loads in flight
  constexpr int kPF = (NL == 4) ? 8 : 16;
  // Without the K-split the weight chunks are DOUBLE-BUFFERED: the next chunk's weights travel global →
  // registers while the MFMAs of the current chunk run, and are stored to the other LDS buffer
  // afterwards (one barrier per chunk).  Exposed fills were 15-20 % of the K >= 256 GEMMs and convs.
  constexpr bool DB = (KS == 1);
  constexpr int CH = DB ? kAChunk / 2 : kAChunk;                 // A steps per chunk
  constexpr int kBufFloats = CH * MB * 64;
  constexpr int NWR = kBufFloats / 256;                          // staged weights per thread and chunk
  constexpr int kRedFloats = (KS > 1) ? (KS - 1) * (MB * NACC * 16 + 2 * NACC) * 64 : 0;
  constexpr int kAsFloats = (DB ? 2 : 1) * kBufFloats > kRedFloats ? (DB ? 2 : 1) * kBufFloats : kRedFloats;
  __shared__ float As[kAsFloats];
  __shared__ float sW[32 * MB];
  __shared__ float tW[32 * MB];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  constexpr int WT = (KS > 1) ? 1 : 4;  // column tiles per workgroup
  const int tiles_per_sample = (int)((p.Ncol + TN * WT - 1) / (TN * WT));
  // Workgroup -> (column tile bx, row-block group by).  With several row-block groups (M > 32·MB) the
  // grid is 1-D and XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so the `ygroups`
  // groups of ONE column tile are given consecutive slots of the SAME XCD — they run concurrently and
  // share the operand tile in that XCD's L2 instead of each pulling it from HBM / Infinity Cache.
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.ygroups > 1) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    by = slot % p.ygroups;
    bx = (slot / p.ygroups) * 8 + xcd;
    if (bx >= p.xtiles) return;  // padding of the last round (before any barrier)
  }
  const int b = bx / tiles_per_sample;
  const int64_t n0 = ((int64_t)(bx % tiles_per_sample) * WT + (KS > 1 ? 0 : wave)) * TN;
  const int m0 = by * 32 * MB;
  const int nA = (p.K + 1) / 2;

  if (PRO == PRO_LN) {
    // s[m] = Σ_k W[m][k]·γ[k], t[m] = Σ_k W[m][k]·β[k]: 8 threads per row, k interleaved (a single
    // thread per row is K dependent-latency loads: 60-110 us at K = 512..1024)
    for (int r0 = 0; r0 < 32 * MB; r0 += 32) {
      const int r = r0 + (threadIdx.x >> 3), part = threadIdx.x & 7;
      const int m = m0 + r;
      float s = 0.f, t = 0.f;
      if (m < p.M)
        for (int k = part; k < p.K; k += 8) {
          const float wv = weight_at(p, m, k);
          s += wv * p.ln_g[k];
          t += wv * p.ln_b[k];
        }
      s += __shfl_xor(s, 1, 64); t += __shfl_xor(t, 1, 64);
      s += __shfl_xor(s, 2, 64); t += __shfl_xor(t, 2, 64);
      s += __shfl_xor(s, 4, 64); t += __shfl_xor(t, 4, 64);
      if (part == 0) { sW[r] = s; tW[r] = t; }
    }
  }

  // ---- per-lane input addressing ----
  int64_t col_off;
  bool col_ok;
  int kw0 = 0, kh0 = 0, kd0 = 0;
  if (LOADER == LOAD_S2D) {
    const int64_t n = n0 + 2 * j;  // coarse voxel pair (wo even)
    col_ok = n < p.Ncol;
    const int64_t nn = col_ok ? n : 0;
    const int wo = (int)(nn % p.Wo);
    const int64_t t2 = nn / p.Wo;
    const int ho = (int)(t2 % p.Ho);
    const int dz = (int)(t2 / p.Ho);
    col_off = ((int64_t)(2 * dz) * p.Hi + 2 * ho) * p.Wi + 2 * wo;
  } else {
    col_off = n0 + NACC * j;
    col_ok = col_off < p.Ncol;
    if (LOADER == LOAD_K3) {
      const int64_t nn = col_ok ? col_off : 0;
      kw0 = (int)(nn % p.Wi);
      kh0 = (int)((nn / p.Wi) % p.Hi);
      kd0 = (int)(nn / ((int64_t)p.Wi * p.Hi));
    }
  }

  constexpr int NR = (PRO == PRO_BMUL) ? 2 * NL : NL;  // ring slot: operand (+ gate operand)
  auto fetch = [&](int s, float (&v)[NR]) {
    if constexpr (LOADER == LOAD_PLAIN) {
      fetch_plain_raw<NL, PRO == PRO_BMUL>(p, b, 2 * s + h, col_off, col_ok, v);
    } else if constexpr (LOADER == LOAD_S2D) {
      const int c = 2 * (s >> 2) + h;
      const bool ok = col_ok && c < p.Cin;
      const int cc = c < p.Cin ? c : p.Cin - 1;
      const int64_t off = (col_ok ? col_off : 0) + (int64_t)((s >> 1) & 1) * p.Hi * p.Wi + (int64_t)(s & 1) * p.Wi;
      vload<NL>(p.x[0] + ((int64_t)b * p.Cin + cc) * p.Vin + off, v);
#pragma unroll
      for (int e = 0; e < NL; ++e) v[e] = ok ? v[e] : 0.f;
    } else {
      // LOAD_K3 (NL == 4): taps of the 3x3x3 stencil, zero padding — clamped addresses + selects
      const int c = 2 * (s / 27) + h;
      const int tap = s % 27;
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int zd = kd0 + kd - 1, zh = kh0 + kh - 1;
      const bool ok = col_ok && c < p.Cin && zd >= 0 && zd < p.Di && zh >= 0 && zh < p.Hi;
      const int cc = c < p.Cin ? c : p.Cin - 1;
      const int zdc = zd < 0 ? 0 : (zd >= p.Di ? p.Di - 1 : zd);
      const int zhc = zh < 0 ? 0 : (zh >= p.Hi ? p.Hi - 1 : zh);
      const AT* row = p.x[0] + ((int64_t)b * p.Cin + cc) * p.Vin + ((int64_t)zdc * p.Hi + zhc) * p.Wi;
      float t4[4];
      vload<4>(row + kw0, t4);
      const float4 t = make_float4(t4[0], t4[1], t4[2], t4[3]);
      const float lft = aget(row + (kw0 > 0 ? kw0 - 1 : 0));
      const float rgt = aget(row + (kw0 + 4 < p.Wi ? kw0 + 4 : kw0));
      const float l0 = kw0 > 0 ? lft : 0.f;
      const float r0 = kw0 + 4 < p.Wi ? rgt : 0.f;
      float o0, o1, o2, o3;
      if (kw == 1) { o0 = t.x; o1 = t.y; o2 = t.z; o3 = t.w; }
      else if (kw == 0) { o0 = l0; o1 = t.x; o2 = t.y; o3 = t.z; }
      else { o0 = t.y; o1 = t.z; o2 = t.w; o3 = r0; }
      v[0] = ok ? o0 : 0.f; v[1 % NL] = ok ? o1 : 0.f; v[2 % NL] = ok ? o2 : 0.f; v[3 % NL] = ok ? o3 : 0.f;
    }
  };

  f32x16 acc[MB][NACC];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][q][r] = 0.f;

  float s1[NACC], s2[NACC], shift[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e) s1[e] = s2[e] = shift[e] = 0.f;

  const int nload = (LOADER == LOAD_S2D) ? nA / 2 : nA;
  float ring[kPF][NR];
  // unconditional, clamped prefetch: a load inside a branch costs an s_waitcnt vmcnt(0)
#pragma unroll
  for (int i = 0; i < kPF; ++i) {
    const int si = (KS > 1 ? wave * kPF : 0) + i;
    fetch(si < nload ? si : nload - 1, ring[i]);
  }
  if (PRO == PRO_LN) {
    // pivot = channel-0 value (held by half 0 in ring[0]): well-conditioned single-pass variance
    if (KS > 1) {
      float pv[NR];
      fetch(0, pv);  // every wave needs the SAME pivot
#pragma unroll
      for (int e = 0; e < NACC; ++e) shift[e] = __shfl(pv[e % NL], j, 64);
    } else {
#pragma unroll
      for (int e = 0; e < NACC; ++e) shift[e] = __shfl(ring[0][e % NL], j, 64);
    }
  }

  constexpr int kGroup = kPF * ((LOADER == LOAD_S2D) ? 2 : 1);
  static_assert(CH % (KS * kGroup) == 0, "chunk must hold whole rounds of (K-split) prefetch groups");
  // steps are processed in groups of kPF*ASTEP with NO per-step guard (a guard turns every K-step into
  // its own basic block: ds_read → s_waitcnt lgkmcnt(0) → MFMA, fully serialised); the tail of the
  // last group gets zero weights instead
  float wreg[NWR];
  // weights of chunk [a0, a0+an) in operand order → registers; branch-free (clamped address + select)
  auto load_chunk = [&](int a0, int an) {
#pragma unroll
    for (int uu = 0; uu < NWR; ++uu) {
      const int idx = threadIdx.x + uu * 256;
      const int l = idx & 63;
      const int mb = (idx >> 6) % MB;
      const int a = a0 + idx / (64 * MB);
      const int m = m0 + mb * 32 + (l & 31);
      const int kk = a_k<LOADER>(a, l >> 5);
      const bool ok = idx < an * MB * 64 && m < p.M && kk < p.K;
      const int mc = m < p.M ? m : p.M - 1, kc = kk < p.K ? kk : p.K - 1;
      float wv = weight_at(p, mc, kc);
      if (PRO == PRO_LN) wv *= p.ln_g[kc];
      wreg[uu] = ok ? wv : 0.f;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int uu = 0; uu < NWR; ++uu) As[buf * kBufFloats + threadIdx.x + uu * 256] = wreg[uu];
  };

  __syncthreads();  // sW / tW (and the previous use of LDS) settled
  load_chunk(0, min(CH, nA));
  store_chunk(0);
  __syncthreads();
  int cbuf = 0;
  for (int a0 = 0; a0 < nA; a0 += CH) {
    const int an = min(CH, nA - a0);
    const int an_pad = ((an + KS * kGroup - 1) / (KS * kGroup)) * (KS * kGroup);
    const bool more = a0 + CH < nA;
    if (DB && more) load_chunk(a0 + CH, min(CH, nA - a0 - CH));  // in flight during the MFMAs below
    const float* Ab = As + cbuf * kBufFloats;
    constexpr int ASTEP = (LOADER == LOAD_S2D) ? 2 : 1;
    // kAChunk is a multiple of kPF*ASTEP, so the ring slot of a step is static after unrolling
    for (int al = (KS > 1 ? wave * kPF * ASTEP : 0); al < an_pad; al += KS * kPF * ASTEP) {
#pragma unroll
      for (int u = 0; u < kPF; ++u) {
        const int ali = al + u * ASTEP;
        const int s = (a0 + ali) / ASTEP;
        float cur[NR];
#pragma unroll
        for (int e = 0; e < NR; ++e) cur[e] = ring[u][e];
        {
          if (LOADER == LOAD_S2D) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
              const float av0 = Ab[(ali * MB + mb) * 64 + lane];        // tw = 0
              const float av1 = Ab[((ali + 1) * MB + mb) * 64 + lane];  // tw = 1
              acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, cur[0], acc[mb][0], 0, 0, 0);
              acc[mb][1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, cur[2 % NL], acc[mb][1 % NACC], 0, 0, 0);
              acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, cur[1 % NL], acc[mb][0], 0, 0, 0);
              acc[mb][1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, cur[3 % NL], acc[mb][1 % NACC], 0, 0, 0);
            }
          } else {
            float bvv[NACC];
            // deferred masking of the raw ring slot: lanes past the last column and the odd-K pad
            // channel contribute zero (their clamped re-reads are finite; weights of the pad are 0)
            const bool cok = col_ok && (2 * s + h) < p.Cin;
#pragma unroll
            for (int e = 0; e < NACC; ++e) {
              float t = cur[e % NL];
              if (LOADER == LOAD_PLAIN) {
                if (PRO == PRO_BMUL) t = cur[(NL + e) % NR] > 0.f ? t : 0.f;
                t = cok ? t : 0.f;
              }
              if (PRO == PRO_LN) {
                t = cok ? t - shift[e] : 0.f;  // padded steps / lanes must not enter the statistics
                s1[e] += t;
                s2[e] += t * t;
              }
              bvv[e] = t;
            }
            if (PRO == PRO_GELU) {
#pragma unroll
              for (int e = 0; e < NACC; ++e) bvv[e] = gelu_f(bvv[e]);
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
              const float av = Ab[(ali * MB + mb) * 64 + lane];
#pragma unroll
              for (int q = 0; q < NACC; ++q) {
#if defined(FZ_PROBE_MFMA_NONE)
                acc[mb][q][0] += av * bvv[q];  // diagnostics build: operand traffic only
#elif defined(FZ_PROBE_MFMA_HALF)
                if (q & 1) acc[mb][q][0] += av * bvv[q];
                else acc[mb][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bvv[q], acc[mb][q], 0, 0, 0);
#elif defined(FZ_PROBE_MFMA_AGPR)
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[mb][q]) : "v"(av), "v"(bvv[q]));
#else
                acc[mb][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bvv[q], acc[mb][q], 0, 0, 0);
#endif
              }
            }
          }
        }
        // Refill the slot right behind the MFMAs that consumed it, and pin it there: the slot registers
        // are the MFMA operands, so the refill cannot be issued earlier; left alone, the scheduler sinks
        // all kPF refills of a group to the bottom of the unrolled body and the first slot of the next
        // group is awaited right after it was requested — one exposed memory round trip per group.
        {
          const int sn = s + KS * kPF;
          fetch(sn < nload ? sn : nload - 1, ring[u]);  // tail: harmless re-read of the last step
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // next chunk: registers → the other buffer (double-buffered) or, with the K-split, a plain refill
    if (more) {
      if (DB) {
        store_chunk(cbuf ^ 1);
        cbuf ^= 1;
      } else {
        __syncthreads();  // everyone is done reading the single buffer
        load_chunk(a0 + CH, min(CH, nA - a0 - CH));
        store_chunk(0);
      }
      __syncthreads();
    }
  }

  if (KS > 1) {
    // add the K-slices: waves 1..KS-1 park their accumulators (and LN sums) in LDS, wave 0 finishes
    __syncthreads();  // everyone is done reading As
    constexpr int kPer = (MB * NACC * 16 + 2 * NACC) * 64;
    if (wave > 0) {
      float* dst = As + (wave - 1) * kPer + lane;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((mb * NACC + q) * 16 + r) * 64] = acc[mb][q][r];
#pragma unroll
      for (int e = 0; e < NACC; ++e) {
        dst[(MB * NACC * 16 + e) * 64] = s1[e];
        dst[(MB * NACC * 16 + NACC + e) * 64] = s2[e];
      }
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < KS - 1; ++w) {
      const float* src = As + w * kPer + lane;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mb][q][r] += src[((mb * NACC + q) * 16 + r) * 64];
#pragma unroll
      for (int e = 0; e < NACC; ++e) {
        s1[e] += src[(MB * NACC * 16 + e) * 64];
        s2[e] += src[(MB * NACC * 16 + NACC + e) * 64];
      }
    }
  }

  float mu_d[NACC], rstd[NACC];
  if (PRO == PRO_LN) {
#pragma unroll
    for (int e = 0; e < NACC; ++e) {
      const float t1 = s1[e] + __shfl_xor(s1[e], 32, 64);
      const float t2 = s2[e] + __shfl_xor(s2[e], 32, 64);
      const float inv = 1.0f / (float)p.Cin;
      const float md = t1 * inv;
      float var = t2 * inv - md * md;
      var = var > 0.f ? var : 0.f;
      mu_d[e] = md;
      rstd[e] = 1.0f / sqrtf(var + p.ln_eps);
    }
    if (p.stats_out != nullptr && by == 0 && h == 0 && col_ok) {
      float mean[NACC];
#pragma unroll
      for (int e = 0; e < NACC; ++e) mean[e] = shift[e] + mu_d[e];
      float* so = p.stats_out + (int64_t)b * 2 * p.Vin;
      vstore<NACC>(so + col_off, mean);
      vstore<NACC>(so + p.Vin + col_off, rstd);
    }
  }
  if (!col_ok) return;
  const int64_t ncol = (LOADER == LOAD_S2D) ? n0 + 2 * j : col_off;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    if (PRO == PRO_LN) {
      // y = rstd·(acc − μ_d·s[m]) + t[m]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float sw = sW[mb * 32 + rl];
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[mb][q][r] = rstd[q] * (acc[mb][q][r] - mu_d[q] * sw);
      }
    }
    store_block<NACC, EPI, LOADER == LOAD_S2D>(p, acc[mb], b, m0 + mb * 32, ncol, h,
                                                PRO == PRO_LN ? tW + mb * 32 : nullptr);
  }
}

}  // namespace fz

using namespace fz;

// process-wide switch of the split-bf16 MFMA family: FZ_GEMM_BX (read once) unless fz_gemm_bx_enable() set it
static std::atomic<int> g_bx_on{-1};
// Diagnostic environment knobs are read ONCE per process and validated (> 0): the row count that sizes a caller's
// workspace and the grid of the launch that fills it can then never disagree, and an empty / zero value cannot produce
// a zero-sized grid.
// (probe builds only: FZ_KNOB is a compile-time "unset" in the shipped library — fz_common.h)
static int knob_pos(const fz::EnvKnob& k, int dflt) { return k.set && k.val > 0 ? k.val : dflt; }
static int knob_mlp_wg_wgs() { return knob_pos(FZ_KNOB("FZ_MLP_WG_WGS"), 512); }
static int knob_gemm_dw_wgs() { return knob_pos(FZ_KNOB("FZ_GEMM_DW_WGS"), 512); }
static int knob_chain64_p512() { return knob_pos(FZ_KNOB("FZ_CHAIN64_P512"), 1) == 1; }   // 2 = off (A/B runs)
static int knob_res_prefetch() { return knob_pos(FZ_KNOB("FZ_RES_PREFETCH"), 1) == 1; }   // 2 = off
static int knob_p32() { return knob_pos(FZ_KNOB("FZ_GEMM_P32"), 1) == 1; }                // 2 = off
static int knob_p32_wgs() { return knob_pos(FZ_KNOB("FZ_GEMM_P32_WGS"), 512); }           // resident: 2 per CU
static int knob_head_fwd() { const auto& k = FZ_KNOB("FZ_HEAD_FWD"); return k.set ? k.val : 1; }   // 0: the head through gemm_p32
static int knob_chain_wgb() { const auto& k = FZ_KNOB("FZ_CHAIN_WGB"); return k.set ? k.val : 1; }   // 0: fp32-MFMA weight-gradient passes
static int knob_chain_fwd_bx() { const auto& k = FZ_KNOB("FZ_CHAIN_FWD_BX"); return k.set ? k.val : 1; }   // 0: the fp32-MFMA forward chain
static int knob_mlp_wgs(int dflt) { return knob_pos(FZ_KNOB("FZ_MLP_WGS"), dflt); }

static int gemm_bx_enabled() {
  int v = g_bx_on.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("FZ_GEMM_BX");
    v = e ? (atoi(e) != 0) : 1;
    g_bx_on.store(v, std::memory_order_relaxed);
  }
  return v;
}
extern "C" int fz_gemm_bx_enable(int on) {
  const int prev = gemm_bx_enabled();
  if (on >= 0) g_bx_on.store(on != 0, std::memory_order_relaxed);
  return prev;
}

// Flat C view of GemmArgsT for the ABI (see include/factorizer_hip.h: fz_gemm_desc).
template <typename AT>
static int gemm_launch(const fz_gemm_desc* d, fz_stream_t stream) {
  if (!d->x[0] || !d->w || !d->y) return fail(FZ_E_ARG, "fz_gemm: null pointer");
  if (d->loader < LOAD_PLAIN || d->loader > LOAD_K3) return fail(FZ_E_ARG, "fz_gemm: bad loader");
  if (d->epilogue < EPI_PLAIN || d->epilogue > EPI_LNBWD) return fail(FZ_E_ARG, "fz_gemm: bad epilogue");
  const bool lnb64 = d->epilogue == EPI_LNBWD && d->M == 64 && d->K == 64 && d->nsrc == 1 && !d->bmul;
  if (d->epilogue == EPI_LNBWD && ((!lnb64 && (d->M != 32 || (d->K != 32 && d->K != 64))) || d->loader != LOAD_PLAIN || !d->lnb_x || !d->lnb_stats ||
                                   !d->lnb_g || !d->lnb_part || d->bias || d->res || d->emul || d->eact || d->ln))
    return fail(FZ_E_UNSUPPORTED, "fz_gemm: LayerNorm-backward epilogue needs M == 32 (K = 32 or 64) or M == K == 64, plain loader");
  if (d->epilogue == EPI_LNBWD && d->Ncol > ((int64_t)1 << 27)) return fail(FZ_E_UNSUPPORTED, "fz_gemm: LayerNorm-backward epilogue: more than 2^27 voxels per sample");
  if (d->B < 0 || d->Cin < 1 || d->M < 1 || d->K < 1) return fail(FZ_E_SHAPE, "fz_gemm: sizes must be positive");
  if ((d->K & 1) && (d->loader != LOAD_PLAIN || d->ln))
    return fail(FZ_E_UNSUPPORTED, "fz_gemm: odd K only with the plain loader and no LayerNorm prologue");
  if (d->loader == LOAD_K3 && (d->epilogue != EPI_PLAIN || d->ln || d->src_mode != 0 || d->nsrc != 1 ||
                               (d->Wi & 3) || d->K != 27 * d->Cin || (d->Cin & 1)))
    return fail(FZ_E_UNSUPPORTED, "fz_gemm: k3 loader needs W % 4 == 0, even Cin, K = 27*Cin, plain epilogue");
  if (d->bmul && (d->loader != LOAD_PLAIN || d->src_mode != 0 || d->nsrc != 1))
    return fail(FZ_E_UNSUPPORTED, "fz_gemm: bmul needs the plain single-source loader");
  if (d->Ncol % 4 != 0 && d->loader != LOAD_S2D) return fail(FZ_E_UNSUPPORTED, "fz_gemm: voxel count must be a multiple of 4");
  if (d->loader == LOAD_S2D && ((d->Wo & 1) || d->epilogue != EPI_PLAIN || d->ln || d->src_mode != 0 || d->nsrc != 1))
    return fail(FZ_E_UNSUPPORTED, "fz_gemm: space-to-depth loader needs even coarse width, plain epilogue");
  if (d->epilogue == EPI_D2S && (d->M % 8 != 0)) return fail(FZ_E_SHAPE, "fz_gemm: depth-to-space rows must be 8*C");
  if (d->nsrc < 1 || d->nsrc > 2 || d->src_mode != 0)
    return fail(FZ_E_UNSUPPORTED, "fz_gemm: one source, or two sources concatenated along channels");
  if (d->bmul && d->bmul_kind != ACT_RELU) return fail(FZ_E_UNSUPPORTED, "fz_gemm: bmul supports the ReLU gate");
  if (d->B == 0) return FZ_OK;
  GemmArgsT<AT> a;
  for (int i = 0; i < 4; ++i) a.x[i] = (const AT*)d->x[i];
  a.nsrc = d->nsrc; a.src_mode = d->src_mode; a.c0 = d->c0 > 0 ? d->c0 : d->Cin; a.Cin = d->Cin;
  a.Vin = d->Vin; a.Di = d->Di; a.Hi = d->Hi; a.Wi = d->Wi; a.bmul = (const AT*)d->bmul; a.bmul_kind = d->bmul_kind;
  a.w = d->w; a.w_t = d->w_t; a.ldw = d->ldw; a.M = d->M; a.K = d->K;
  a.bias = d->bias; a.ln = d->ln; a.ln_g = d->ln_g; a.ln_b = d->ln_b; a.ln_eps = d->ln_eps;
  a.stats_out = d->stats_out; a.bact = d->bact; a.eact = d->eact; a.res = (const AT*)d->res; a.emul = (const AT*)d->emul;
  a.emul_kind = d->emul_kind; a.y = (AT*)d->y; a.Ncol = d->Ncol; a.Ho = d->Ho; a.Wo = d->Wo; a.B = d->B;
  a.dbg = 0; a.tile_map = 1;
  a.ygroups = 0; a.xtiles = 0; a.tune = d->tune;
  a.lnb_x = (const AT*)d->lnb_x; a.lnb_stats = d->lnb_stats; a.lnb_g = d->lnb_g; a.lnb_gadd = (const AT*)d->lnb_gadd; a.lnb_part = d->lnb_part;
  hipStream_t st = (hipStream_t)stream;
  const int mblocks = (d->M + 31) / 32;
  if (lnb64) {   // 64 -> 64 input gradient + LayerNorm backward over 64 channels: gemm_chain64_kernel, SINGLE form
    if (d->Ncol % 4 != 0) return fail(FZ_E_UNSUPPORTED, "fz_gemm: voxel count must be a multiple of 4");
    if (!d->lnb_gadd) return fail(FZ_E_UNSUPPORTED, "fz_gemm: the 64-channel LayerNorm-backward epilogue needs the added gradient");
    ChainArgsT<AT> c = {};
    const int ntiles = (int)fz_mlp_partials(d->B, d->Ncol);
    constexpr int lds64 = (8192 + 8192 + 128 + 64 + 512) * (int)sizeof(float);
    auto kern = products_split(d->products) ? gemm_chain64_kernel<true, AT, true, true> : gemm_chain64_kernel<true, AT, true, false>;
    FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds64));
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < 512 ? ntiles : 512)), dim3(256), lds64, st, a, c, ntiles);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }

  // ---- split-bf16 MFMA family (gemm_bx.hip): every layer with a reduction length >= 64 (stages 1-4 of the U-shape) ----
  // fp32 products as six exact bf16 products on the bf16 matrix pipe (6/16 of the fp32-MFMA time, error <= the fp32
  // MFMA's own: tools/probes/bx6_accuracy.hip); FZ_GEMM_BX=0 keeps every GEMM on v_mfma_f32_32x32x2_f32
  {
    const int bx_on = products_split(d->products);
    const int pro_bx = d->ln ? 1 : (d->bact == ACT_GELU ? 2 : (d->bmul ? 3 : 0));
    const bool one_pro = (d->ln != 0) + (d->bact != 0) + (d->bmul != nullptr) <= 1;
    // (bf16 storage, no prologue: from K = 32 — three bf16 products per fp32 product; the stage-0 layers are matrix-pipe
    // bound there once their bytes are halved.  With a LayerNorm / GELU prologue the K = 32 half chunk would cost as
    // much as the fp32 MFMAs it replaces; fp32 storage: those layers are HBM-bound on the fp32 MFMA)
    const int bx_kmin = (sizeof(AT) == 2 && !d->ln && d->bact == 0) ? 32 : 64;
    if (bx_on && one_pro && d->bact != ACT_RELU && d->K >= bx_kmin && d->M >= 32 && (d->loader == LOAD_PLAIN || d->loader == LOAD_S2D) &&
        (d->epilogue == EPI_PLAIN || d->epilogue == EPI_D2S) && !(d->loader == LOAD_S2D && pro_bx) && !(d->epilogue == EPI_D2S && pro_bx))
    {
      const int rc = gemm_bx_launch<AT>(a, d->loader, d->epilogue, pro_bx, stream);
      if (rc != FZ_E_UNSUPPORTED) return rc;   // shapes outside the family (K % 64, M % 32, alignment): the kernels below
    }
  }

  // ---- the head: Linear(32 -> M <= 4), nothing fused: 32·M FMAs per voxel on the VALU, bandwidth-bound (headbwd.hip) ----
  if (knob_head_fwd() && d->loader == LOAD_PLAIN && d->epilogue == EPI_PLAIN && d->M <= 4 && d->K == 32 && d->Cin == 32 && a.c0 == 32 &&
      !d->ln && !d->bact && !d->eact && !d->res && !d->bmul && !d->emul && !d->w_t && d->ldw == 32 && d->Ncol == d->Vin && d->Ncol % 4 == 0 &&
      d->Ncol / 4 < ((int64_t)1 << 31) && (int64_t)d->B * (d->Ncol / 4) < ((int64_t)1 << 31) && d->stats_out == nullptr)
    return fz_head_fwd(d->x[0], d->w, d->bias, d->y, d->B, d->M, 32, d->Ncol, d->act_dtype, stream);

  // ---- Kernel A': persistent 32 -> 32 without a residual (stage 0: LayerNorm + in-projection, plain projections) ----
  if (knob_p32() && d->loader == LOAD_PLAIN && d->epilogue == EPI_PLAIN && d->M <= 32 && d->K == 32 && d->Cin == 32 && (a.c0 & 1) == 0 && d->Vin < ((int64_t)1 << 28) &&
      (int64_t)5 * d->Ncol * (int64_t)sizeof(AT) < ((int64_t)1 << 32) /* 32-bit store offsets (4·h·Ncol + col)·es */ && !d->res && !d->bmul &&
      !d->emul && d->Ncol % 4 == 0 && d->Ncol == d->Vin && d->B * ((d->Ncol + 127) / 128) >= 4096 && d->B * ((d->Ncol + 127) / 128) < ((int64_t)1 << 30) &&
      !(d->bact && !d->ln) /* the activation-only form needs scratch at two waves per SIMD: Kernel A keeps it */) {
    const unsigned ntiles = (unsigned)(d->B * ((d->Ncol + 127) / 128));
    const unsigned cap = (unsigned)knob_p32_wgs();
    const unsigned wgs = (ntiles + 3) / 4 < cap ? (ntiles + 3) / 4 : cap;
    const int pf = (d->bact ? 1 : 0) | (d->ln ? 2 : 0);
    dim3 grid(wgs), block(256);
    if (pf == 0) hipLaunchKernelGGL((gemm_p32_kernel<0, AT>), grid, block, 0, st, a, ntiles);
    else if (pf == 2) hipLaunchKernelGGL((gemm_p32_kernel<2, AT>), grid, block, 0, st, a, ntiles);
    else hipLaunchKernelGGL((gemm_p32_kernel<3, AT>), grid, block, 0, st, a, ntiles);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }

  // ---- Kernel A: whole operand in registers (K <= 64, plain loader) ----
  // measured (round-1/2 probe `gemm_probe5`): the register-resident kernel wins for K <= 32, the
  // streaming ring for K = 64 (4.4 vs 3.3 TB/s at 64->32, 128^3)
  int res_maxk = 32;
  if (d->epilogue == EPI_LNBWD) res_maxk = 64;
  // ... except one 32-row block without a residual or gate: with the ring refills pinned the streaming
  // kernel overlaps its MFMAs with the loads still in flight, which the LayerNorm prologue of the
  // resident kernel cannot (it needs the whole column first): 32->32 at 128^3, LayerNorm + ReLU 345 -> 274 us,
  // plain 278 -> 248 us; with a residual (357 vs 362 us) or two row blocks (391 vs 439 us) the resident
  // kernel stays ahead (round-1/2 probe `gemm_probe9`)
  const bool stream_small = mblocks == 1 && d->K >= 16 && d->K <= 32 && !d->res && !d->bmul && !d->emul &&
                            d->epilogue == EPI_PLAIN && d->bact == 0;
  if (d->loader == LOAD_PLAIN && d->K <= res_maxk && !stream_small) {
    const int nA = (d->K + 1) / 2;
    int RB = mblocks < 8 ? mblocks : 8;
    while ((size_t)(nA * RB * 64 + 32 * RB) * sizeof(float) > 65536) --RB;
    const size_t lds = (size_t)(nA * RB * 64 + 32 * RB + (d->epilogue == EPI_LNBWD ? 256 : 0)) * sizeof(float);
    const int64_t tiles = (d->Ncol + 511) / 512;
    dim3 grid((unsigned)(tiles * d->B), (unsigned)((mblocks + RB - 1) / RB)), block(256);
#define FZ_RES_PF(NS, E, BM, PFv) hipLaunchKernelGGL((gemm_resident_kernel<NS, E, BM, false, AT, PFv>), grid, block, lds, st, a, RB)
#define FZ_RES(NS, E, BM)                                                  \
  do {                                                                    \
    const int pf = (d->bact ? 1 : 0) | (d->ln ? 2 : 0);                   \
    if (pf == 0) FZ_RES_PF(NS, E, BM, 0);                                 \
    else if (pf == 1) FZ_RES_PF(NS, E, BM, 1);                            \
    else if (pf == 2) FZ_RES_PF(NS, E, BM, 2);                            \
    else FZ_RES_PF(NS, E, BM, 3);                                         \
  } while (0)
    if (d->epilogue == EPI_LNBWD) {
      if (d->bmul) return fail(FZ_E_UNSUPPORTED, "fz_gemm: bmul with LayerNorm-backward epilogue");
#define FZ_RES_LNB(NS, GA) hipLaunchKernelGGL((gemm_resident_kernel<NS, EPI_LNBWD, false, GA>), grid, block, lds, st, a, RB)
      if (d->lnb_gadd) { if (nA <= 16) FZ_RES_LNB(16, true); else FZ_RES_LNB(32, true); }
      else { if (nA <= 16) FZ_RES_LNB(16, false); else FZ_RES_LNB(32, false); }
    } else if (d->epilogue == EPI_D2S) {
      if (d->bmul) return fail(FZ_E_UNSUPPORTED, "fz_gemm: bmul with depth-to-space epilogue");
      if (nA <= 16) FZ_RES(16, EPI_D2S, false); else FZ_RES(32, EPI_D2S, false);
    } else if (d->bmul) {
      if (nA <= 16) FZ_RES(16, EPI_PLAIN, true); else FZ_RES(32, EPI_PLAIN, true);
    } else if (d->res && !d->emul && d->M == 32 && nA <= 16 && !d->ln && knob_res_prefetch()) {
      if (d->bact) hipLaunchKernelGGL((gemm_resident_kernel<16, EPI_PLAIN, false, false, AT, 1, true>), grid, block, lds, st, a, RB);
      else hipLaunchKernelGGL((gemm_resident_kernel<16, EPI_PLAIN, false, false, AT, 0, true>), grid, block, lds, st, a, RB);
    } else {
      if (nA <= 16) FZ_RES(16, EPI_PLAIN, false); else FZ_RES(32, EPI_PLAIN, false);
    }
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }

  // (A persistent variant of the streaming kernel — whole weight block resident, the prefetch ring running
  // across tile boundaries — was measured for 32 < K <= 128 at stages 0/1 and was not faster:
  // those GEMMs sit at 56-71 TFLOP/s of fp32 MFMA with the operand loads as the stall reason, not the
  // per-tile prologue / epilogue.)
  // ---- Kernel B: streaming; pick the column-tile width so the grid fills the chip ----
  // tile choice from a sweep on MI355X (round-1/2 probe `gemm_probe3`, FZ_GEMM_CFG): take the widest
  // column tile that still gives >= 256 workgroups (one per CU); two row blocks per workgroup only
  // when that still leaves >= 512 workgroups
  int nacc = 4, MBsel = mblocks >= 2 ? 2 : 1;
  if (d->loader == LOAD_S2D) { nacc = 2; MBsel = 1; }
  else {
    const bool narrow_ok = d->loader == LOAD_PLAIN && d->epilogue == EPI_PLAIN;
    auto wgs = [&](int na, int mb) {
      const int64_t t = (d->Ncol + 32 * na * 4 - 1) / (32 * na * 4);
      return t * d->B * ((mblocks + mb - 1) / mb);
    };
    if (wgs(nacc, MBsel) < 512 && MBsel == 2) MBsel = 1;
    if (narrow_ok) {
      if (wgs(nacc, MBsel) < 256) nacc = 2;
      // (when even 32-voxel tiles cannot give one workgroup per CU — the 8^3 bottleneck — stay with 64-voxel tiles and
      // let the K-split below fill the chip: 8-byte lane loads; 512->1024 at 2 x 8^3: 33 against 43 us, round-1/2 probe `gemm_deep`)
      if (wgs(nacc, MBsel) < 256 && !(wgs(1, MBsel) < 256 && d->K >= 256)) nacc = 1;
      const char* e = FZ_KNOB("FZ_GEMM_CFG").str;  // probe builds: "<nacc><mb>", e.g. 42
      if (e && e[0] && e[1]) { nacc = e[0] - '0'; MBsel = e[1] - '0'; if (MBsel == 2 && (nacc != 4 || mblocks < 2)) MBsel = 1; }
    }
  }
  const int TN = 32 * nacc;
  // K-split across the waves of a workgroup when the plain decomposition leaves most CUs idle
  // and K is long enough to give every wave whole prefetch groups
  int ks = 1;
  {
    const int64_t wg1 = ((d->Ncol + TN * 4 - 1) / (TN * 4)) * d->B * ((mblocks + MBsel - 1) / MBsel);
    const bool shape_ok = MBsel == 1 && ((d->loader == LOAD_PLAIN && d->epilogue == EPI_PLAIN && nacc <= 2) ||
                                         d->loader == LOAD_S2D);
    if (shape_ok && wg1 < 256 && d->K >= 256) ks = 4;
  }
  const int WT = ks > 1 ? 1 : 4;
  const int64_t tiles = (d->Ncol + TN * WT - 1) / (TN * WT);
  const int ygr = (mblocks + MBsel - 1) / MBsel;
  int xcd_grid = 1;
  // (only where the COLUMN operand dominates the traffic: few row-block groups, many column tiles;
  // with e.g. 32 groups x 16 tiles — the deep transposed convs — the weights dominate and the
  // x-fastest order, which runs equal-weight workgroups together, is the better one: 61 vs 105 us)
  a.ygroups = (xcd_grid && ygr > 1 && ygr <= 8 && tiles * d->B >= 64) ? ygr : 0;
  a.xtiles = (int)(tiles * d->B);
  dim3 grid((unsigned)(tiles * d->B), (unsigned)ygr), block(256);
  if (a.ygroups > 1) grid = dim3((unsigned)(((tiles * d->B + 7) / 8) * 8 * ygr), 1);
#define FZ_STR(MB, NA, L, E, PR) hipLaunchKernelGGL((gemm_stream_kernel<MB, NA, L, E, PR>), grid, block, 0, st, a)
#define FZ_STRK(MB, NA, L, E, PR) hipLaunchKernelGGL((gemm_stream_kernel<MB, NA, L, E, PR, 4>), grid, block, 0, st, a)
  if (d->bact == ACT_RELU) return fail(FZ_E_UNSUPPORTED, "fz_gemm: ReLU input prologue is not compiled (streaming)");
  const int pro = d->ln ? PRO_LN : (d->bact == ACT_GELU ? PRO_GELU : (d->bmul ? PRO_BMUL : PRO_NONE));
  if ((d->ln != 0) + (d->bact != 0) + (d->bmul != nullptr) > 1)
    return fail(FZ_E_UNSUPPORTED, "fz_gemm: at most one input prologue (LayerNorm / GELU / gate)");
  if (d->loader == LOAD_S2D) { if (ks == 4) FZ_STRK(1, 2, LOAD_S2D, EPI_PLAIN, PRO_NONE); else FZ_STR(1, 2, LOAD_S2D, EPI_PLAIN, PRO_NONE); }
  else if (d->loader == LOAD_K3) { if (MBsel == 2) FZ_STR(2, 4, LOAD_K3, EPI_PLAIN, PRO_NONE); else FZ_STR(1, 4, LOAD_K3, EPI_PLAIN, PRO_NONE); }
  else if (d->epilogue == EPI_D2S) {
    if (pro != PRO_NONE) return fail(FZ_E_UNSUPPORTED, "fz_gemm: prologue with depth-to-space epilogue");
    if (MBsel == 2) FZ_STR(2, 4, LOAD_PLAIN, EPI_D2S, PRO_NONE); else FZ_STR(1, 4, LOAD_PLAIN, EPI_D2S, PRO_NONE);
  } else {
#define FZ_STR_SHAPES(PR)                                                                                   \
  do {                                                                                                      \
    if (nacc == 4) { if (MBsel == 2) FZ_STR(2, 4, LOAD_PLAIN, EPI_PLAIN, PR); else FZ_STR(1, 4, LOAD_PLAIN, EPI_PLAIN, PR); } \
    else if (nacc == 2) { if (ks == 4) FZ_STRK(1, 2, LOAD_PLAIN, EPI_PLAIN, PR); else FZ_STR(1, 2, LOAD_PLAIN, EPI_PLAIN, PR); } \
    else { if (ks == 4) FZ_STRK(1, 1, LOAD_PLAIN, EPI_PLAIN, PR); else FZ_STR(1, 1, LOAD_PLAIN, EPI_PLAIN, PR); } \
  } while (0)
    if (pro == PRO_LN) FZ_STR_SHAPES(PRO_LN);
    else if (pro == PRO_GELU) FZ_STR_SHAPES(PRO_GELU);
    else if (pro == PRO_BMUL) FZ_STR_SHAPES(PRO_BMUL);
    else FZ_STR_SHAPES(PRO_NONE);
  }
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_gemm(const fz_gemm_desc* d, fz_stream_t stream) {
  if (!d) return fail(FZ_E_ARG, "fz_gemm: null descriptor");
  if (d->act_dtype == FZ_STORE_F32) return gemm_launch<float>(d, stream);
  if (d->act_dtype == FZ_STORE_BF16) return gemm_launch<bf16>(d, stream);
  return fail(FZ_E_ARG, "fz_gemm: act_dtype must be FZ_STORE_F32 or FZ_STORE_BF16");
}

extern "C" int64_t fz_gemm_lnbwd_partials(const fz_gemm_desc* d) {
  if (!d) return -1;
  if (d->M == 64) return fz_mlp_partials(d->B, d->Ncol);   // 256-voxel tiles, rows of 128 floats
  return ((d->Ncol + 511) / 512) * d->B;  // one row per workgroup of the resident kernel (rows of 64 floats)
}

// MLP chain for C = 32, hidden 64 (see gemm_chain_kernel).  Replaces, per FactorizerBlock,
// Linear∘LayerNorm + Linear∘GELU + residual (layers/mlp.py:54-63, factorizer.py:76) in the
// forward and the two input-gradient GEMMs + LayerNorm backward in the backward.
// columns per lane = 2: 64-column wave tiles, <= 168 VGPRs → 3 waves/SIMD with the next-tile prefetch
// (4 columns per lane: 0.71 / 1.38 ms against 0.63 / 1.00 ms, round-1/2 probe `mlp_probe`)
static int mlp_nacc() { return 2; }

extern "C" int64_t fz_mlp_partials(int B, int64_t V) {
  const int64_t tw = 128 * mlp_nacc();
  return ((V + tw - 1) / tw) * (int64_t)B;
}

// rows of `wpart` (fz_mlp_desc mode 2): one per resident workgroup (two per CU), kWgRow floats each
extern "C" int fz_mlp_wgrad_rows(int B, int64_t V) {
  const int64_t nt = fz_mlp_partials(B, V);
  const int wgs = knob_mlp_wg_wgs();
  return (int)(nt < wgs ? nt : wgs);
}
extern "C" int64_t fz_mlp_wgrad_workspace_bytes(int B, int64_t V) {
  return 2 * (int64_t)fz_mlp_wgrad_rows(B, V) * kWgRow * (int64_t)sizeof(float);   // two row blocks (hidden 128 runs in two halves)
}

extern "C" int fz_mlp_pre_supported(int C, int H, int64_t V, int products) {
  if (!(V > 0 && (V % 4) == 0 && V <= ((int64_t)1 << 27) && products_split(products))) return 0;
  if (C == 32 && H == 64) return knob_chain_fwd_bx() ? 1 : 0;
  // C = 64: the 512-thread form around the pre-split images, which walks the tiles in pairs (an even number per sample)
  if (C == 64 && H == 128) return (((V + 255) / 256) % 2 == 0 && knob_chain64_p512()) ? 1 : 0;
  return 0;
}

extern "C" int fz_mlp_supported(int C, int H, int64_t V) {
  const bool shape = (C == 32 && (H == 64 || H == 128)) || (C == 64 && H == 128);
  return (shape && V > 0 && (V % 4) == 0 && V <= ((int64_t)1 << 27)) ? 1 : 0;
}

template <typename AT>
static int mlp_launch(const fz_mlp_desc* d, fz_stream_t stream) {
  if (!fz_mlp_supported(d->C, d->H, d->V)) return fail(FZ_E_UNSUPPORTED, "fz_mlp_chain: needs (C, H) in {(32, 64), (32, 128), (64, 128)}, V % 4 == 0");
  if (d->C == 64 && d->mode == 2) return fail(FZ_E_UNSUPPORTED, "fz_mlp_chain: the fused weight gradients need C == 32, H == 64");
  if (d->B < 0) return fail(FZ_E_SHAPE, "fz_mlp_chain: negative batch");
  const bool pre = d->pre_in != nullptr;   // the block's out-projection in front of the forward chain (x1 is then an OUTPUT)
  if (pre && (d->mode != 0 || !d->pre_w || !d->pre_res || !d->pre_out))
    return fail(FZ_E_ARG, "fz_mlp_chain: pre_in needs mode 0, pre_w, pre_res, pre_out (see fz_mlp_pre_supported)");
  if (pre && !fz_mlp_pre_supported(d->C, d->H, d->V, d->products))
    return fail(FZ_E_UNSUPPORTED, "fz_mlp_chain: the fused out-projection runs on split-bf16 products only (fz_mlp_pre_supported)");
  if (d->post_out && (!pre || d->C != 32 || !d->post_w || d->post_m < 1 || d->post_m > 4))
    return fail(FZ_E_ARG, "fz_mlp_chain: post_out needs pre_in, C == 32, post_w and 1 <= post_m <= 4");
  if ((!d->in && !pre) || !d->w1 || !d->w2 || !d->out || !d->z1 || !d->stats)
    return fail(FZ_E_ARG, "fz_mlp_chain: null pointer");
  if (d->mode == 0 && (!d->ln_g || !d->ln_b)) return fail(FZ_E_ARG, "fz_mlp_chain: forward needs the LayerNorm affine");
  if (d->mode == 1 && (!d->gz1 || !d->x1 || !d->ln_g || !d->part)) return fail(FZ_E_ARG, "fz_mlp_chain: backward needs gz1, x1, gamma, part");
  if (d->mode == 2 && (!d->x1 || !d->ln_g || !d->ln_b || !d->gln || !d->wpart || !d->gw1 || !d->gb1 || !d->gw2 || !d->gb2))
    return fail(FZ_E_ARG, "fz_mlp_chain: backward with weight gradients needs x1, gamma, beta, gln, wpart, gw1, gb1, gw2, gb2");
  if (d->mode == 2 && d->H == 128 && !d->glp) return fail(FZ_E_ARG, "fz_mlp_chain: the fused weight gradients at H == 128 need the glp buffer");
  if (d->mode < 0 || d->mode > 2) return fail(FZ_E_ARG, "fz_mlp_chain: bad mode");
  if (d->B == 0) {
    if (d->mode == 2) {   // no voxels: the sums are empty
      hipStream_t s0 = (hipStream_t)stream;
      FZ_HIP_OK(hipMemsetAsync(d->gw1, 0, sizeof(float) * d->H * 32, s0));
      FZ_HIP_OK(hipMemsetAsync(d->gw2, 0, sizeof(float) * d->H * 32, s0));
      FZ_HIP_OK(hipMemsetAsync(d->gb1, 0, sizeof(float) * d->H, s0));
      FZ_HIP_OK(hipMemsetAsync(d->gb2, 0, sizeof(float) * 32, s0));
      FZ_HIP_OK(hipMemsetAsync(d->gln, 0, sizeof(float) * 64, s0));
    }
    return FZ_OK;
  }
  GemmArgsT<AT> a = {};
  ChainArgsT<AT> c = {};
  a.x[0] = (const AT*)d->in; a.nsrc = 1; a.c0 = 32; a.Cin = 32; a.Vin = d->V; a.M = d->H; a.K = 32; a.Ncol = d->V; a.B = d->B;
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = (int)fz_mlp_partials(d->B, d->V);
  if (d->C == 64) {
    if (d->mode == 1 && !d->in) return fail(FZ_E_ARG, "fz_mlp_chain: null pointer");
    a.x[0] = (const AT*)d->in; a.nsrc = 1; a.c0 = 64; a.Cin = 64; a.Vin = d->V; a.M = 128; a.K = 64; a.Ncol = d->V; a.B = d->B;
    const int wgs64 = knob_mlp_wgs(512);
    dim3 grid64((unsigned)(ntiles < wgs64 ? ntiles : wgs64));
    constexpr int lds64 = (8192 + 8192 + 128 + 64 + 512) * (int)sizeof(float);
    // fp32 storage with split-bf16 products: one 512-thread workgroup per CU around a pre-split weight image
    constexpr int lds512 = (12288 + 12288 + 128 + 64 + 1024) * (int)sizeof(float);
    const bool p512 = products_split(d->products) && ntiles % 2 == 0 && knob_chain64_p512();
    dim3 grid512((unsigned)(ntiles / 2 < 256 ? ntiles / 2 : 256));
    if (d->mode == 0) {
      a.w = d->w1; a.w_t = 0; a.ldw = 64;
      a.bias = d->b1; a.ln = 1; a.ln_g = d->ln_g; a.ln_b = d->ln_b; a.ln_eps = d->ln_eps; a.stats_out = d->stats;
      a.res = (const AT*)d->in; a.y = (AT*)d->out;
      c.wB = d->w2; c.wB_t = 0; c.ldwB = 128; c.biasB = d->b2; c.side = (AT*)d->z1;
      // split-bf16 products only where the kernel stays inside 256 registers without scratch (fp32 storage: 6 / 23 spilled)
      if (pre) {   // (fz_mlp_pre_supported: split-bf16 products, an even tile count)
        if (!p512) return fail(FZ_E_UNSUPPORTED, "fz_mlp_chain: the fused out-projection at C == 64 needs an even number of tiles");
        c.preA = (const AT*)d->pre_in; c.preW = d->pre_w; c.preB = d->pre_b; c.preRes = (const AT*)d->pre_res; c.preOut = (AT*)d->pre_out;
        a.res = (const AT*)d->pre_out;   // the epilogue's residual: the lane's own x1, written a moment earlier
        constexpr int lds_pre = lds512 + (6144 + 64) * (int)sizeof(float);
        auto kern = gemm_chain64_kernel<false, AT, false, true, true, true>;
        FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_pre));
        hipLaunchKernelGGL(kern, grid512, dim3(512), lds_pre, st, a, c, ntiles);
        FZ_LAUNCH_CHECK();
        return FZ_OK;
      }
      if (p512) {
        auto kern = gemm_chain64_kernel<false, AT, false, true, true>;
        FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds512));
        hipLaunchKernelGGL(kern, grid512, dim3(512), lds512, st, a, c, ntiles);
        FZ_LAUNCH_CHECK();
        return FZ_OK;
      }
      auto kern = gemm_chain64_kernel<false, AT, false, false>;   // (odd tile count: the fp32-MFMA form)
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds64));
      hipLaunchKernelGGL(kern, grid64, dim3(256), lds64, st, a, c, ntiles);
    } else {
      a.w = d->w2; a.w_t = 1; a.ldw = 128;             // A1[m = hidden][k = c] = W2[c][hidden]
      a.emul = (const AT*)d->z1; a.y = (AT*)d->out;
      a.lnb_x = (const AT*)d->x1; a.lnb_stats = d->stats; a.lnb_g = d->ln_g; a.lnb_gadd = (const AT*)d->in; a.lnb_part = d->part;
      c.wB = d->w1; c.wB_t = 1; c.ldwB = 64;           // A2[m = c][k = hidden] = W1[hidden][c]
      c.side = (AT*)d->gz1;
      if (p512) {
        auto kern = gemm_chain64_kernel<true, AT, false, true, true>;
        FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds512));
        hipLaunchKernelGGL(kern, grid512, dim3(512), lds512, st, a, c, ntiles);
        FZ_LAUNCH_CHECK();
        return FZ_OK;
      }
      auto kern = gemm_chain64_kernel<true, AT, false, false>;   // (odd tile count: the fp32-MFMA form)
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds64));
      hipLaunchKernelGGL(kern, grid64, dim3(256), lds64, st, a, c, ntiles);
    }
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }
  c.stagger = 0;
  const int wgs = knob_mlp_wgs(d->H == 128 ? 512 : 768);  // resident workgroups (2 or 3 per CU), each walking tiles with a stride of the grid
  dim3 grid((unsigned)(ntiles < wgs ? ntiles : wgs)), block(256);
  if (d->mode == 0) {
    a.w = d->w1; a.w_t = 0; a.ldw = 32;              // A1[m][k] = W1[m][k]
    a.bias = d->b1; a.ln = 1; a.ln_g = d->ln_g; a.ln_b = d->ln_b; a.ln_eps = d->ln_eps; a.stats_out = d->stats;
    a.res = (const AT*)d->in; a.y = (AT*)d->out;
    c.wB = d->w2; c.wB_t = 0; c.ldwB = d->H;         // A2[m][k] = W2[m][k]
    c.biasB = d->b2; c.side = (AT*)d->z1;
    if (d->H == 128) hipLaunchKernelGGL((gemm_chain_kernel<false, 2, 4>), grid, block, 0, st, a, c, ntiles);
    else if (products_split(d->products) && knob_chain_fwd_bx()) {   // split-bf16 form: two workgroups per CU
      const int wgs2 = knob_mlp_wgs(512);
      if (pre) {
        c.preA = (const AT*)d->pre_in; c.preW = d->pre_w; c.preB = d->pre_b; c.preRes = (const AT*)d->pre_res; c.preOut = (AT*)d->pre_out;
        c.postW = d->post_w; c.postB = d->post_b; c.postOut = (AT*)d->post_out; c.postM = d->post_m;
        hipLaunchKernelGGL((gemm_chain_kernel<false, 2, 2, AT, true, true>), dim3((unsigned)(ntiles < wgs2 ? ntiles : wgs2)), block, 0, st, a, c, ntiles);
      } else
      hipLaunchKernelGGL((gemm_chain_kernel<false, 2, 2, AT, true>), dim3((unsigned)(ntiles < wgs2 ? ntiles : wgs2)), block, 0, st, a, c, ntiles);
    } else hipLaunchKernelGGL((gemm_chain_kernel<false, 2, 2>), grid, block, 0, st, a, c, ntiles);
  } else if (d->mode == 2) {
    a.w = d->w2; a.w_t = 1; a.ldw = d->H;
    a.emul = (const AT*)d->z1; a.y = (AT*)d->out;
    a.lnb_x = (const AT*)d->x1; a.lnb_stats = d->stats; a.lnb_g = d->ln_g; a.lnb_gadd = (const AT*)d->in; a.lnb_part = d->part;
    c.wB = d->w1; c.wB_t = 1; c.ldwB = 32;
    const int rows = fz_mlp_wgrad_rows(d->B, d->V);
    const bool bxon = products_split(d->products);
    const int lds = (2 * (bxon ? 3072 : 2048) + 32 + 256 + 4 * 48 * kTS) * (int)sizeof(float);
    if (d->H == 64) {
      // split products: the weight-gradient passes on the bf16 pipe too (WGB, two operand levels); FZ_CHAIN_WGB=0 in a probe
      // build keeps them on v_mfma_f32_16x16x4_f32 (same-box A/B)
      const bool wgb = bxon && knob_chain_wgb();
      auto kern = bxon ? (wgb ? gemm_chain_bwd_wg_kernel<AT, 1, 0, true, true> : gemm_chain_bwd_wg_kernel<AT, 1, 0, true>)
                       : gemm_chain_bwd_wg_kernel<AT, 1, 0, false>;
      const int lds64 = wgb ? (2 * 3072 + 32 + 256 + 4 * 3072) * (int)sizeof(float) : lds;
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds64));
      hipLaunchKernelGGL(kern, dim3((unsigned)rows), block, lds64, st, a, c, ntiles, (float*)d->wpart, (float*)nullptr);
      FZ_LAUNCH_CHECK();
      FinishJob fj = finish_job(FK_CHAIN_WG, kWgRow / 16);
      fj.u.cw = FinChainWg{(const float*)d->wpart, d->ln_g, d->ln_b, d->gw1, d->gb1, d->gw2, d->gb2, d->gln, rows, 64};
      return finish_run(&fj, 1, st);
    } else {   // hidden 128: one launch per 64-row half (wpart holds two row blocks)
      auto kern0 = bxon ? gemm_chain_bwd_wg_kernel<AT, 2, 0, true> : gemm_chain_bwd_wg_kernel<AT, 2, 0, false>;
      auto kern1 = bxon ? gemm_chain_bwd_wg_kernel<AT, 2, 1, true> : gemm_chain_bwd_wg_kernel<AT, 2, 1, false>;
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern0), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern1), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      for (int half = 0; half < 2; ++half) {
        float* wp = (float*)d->wpart + (int64_t)half * rows * kWgRow;
        if (half == 0) hipLaunchKernelGGL(kern0, dim3((unsigned)rows), block, lds, st, a, c, ntiles, wp, d->glp);
        else hipLaunchKernelGGL(kern1, dim3((unsigned)rows), block, lds, st, a, c, ntiles, wp, d->glp);
        FZ_LAUNCH_CHECK();
        FinishJob fj = finish_job(FK_CHAIN_WG, kWgRow / 16);
        fj.u.cw = FinChainWg{(const float*)wp, d->ln_g, d->ln_b, d->gw1 + half * 64 * 32, d->gb1 + half * 64, d->gw2 + half * 64,
                             half == 0 ? d->gb2 : (float*)nullptr, half == 1 ? d->gln : (float*)nullptr, rows, 128};
        const int frc = finish_run(&fj, 1, st);
        if (frc != FZ_OK) return frc;
      }
    }
    return FZ_OK;
  } else {
    a.w = d->w2; a.w_t = 1; a.ldw = d->H;            // A1[m = hidden][k = c] = W2[c][hidden]
    a.emul = (const AT*)d->z1; a.y = (AT*)d->out;
    a.lnb_x = (const AT*)d->x1; a.lnb_stats = d->stats; a.lnb_g = d->ln_g; a.lnb_gadd = (const AT*)d->in; a.lnb_part = d->part;
    c.wB = d->w1; c.wB_t = 1; c.ldwB = 32;           // A2[m = c][k = hidden] = W1[hidden][c]
    c.side = (AT*)d->gz1;
    if (d->H == 128) hipLaunchKernelGGL((gemm_chain_kernel<true, 2, 4>), grid, block, 0, st, a, c, ntiles);
    else hipLaunchKernelGGL((gemm_chain_kernel<true, 2, 2>), grid, block, 0, st, a, c, ntiles);
  }
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_gemm_dw_rows(int B, int64_t V);
template <typename AT>
static int gemm_dw_launch(const fz_gemm_dw_desc* d, fz_stream_t stream) {
  DwArgsT<AT> a;
  a.g = (const AT*)d->g; a.q = (const AT*)d->q; a.w = d->w; a.ldw = d->ldw > 0 ? d->ldw : 32; a.stats = d->stats; a.ln_g = d->ln_g; a.gadd = (const AT*)d->gadd;
  a.y = (AT*)d->y; a.wpart = (float*)d->wpart; a.V = d->V; a.B = d->B;
  const int ntiles = (int)fz_mlp_partials(d->B, d->V);
  const int rows = fz_gemm_dw_rows(d->B, d->V);
  constexpr int lds = (1536 + 32 + 256 + 4 * 64 * kTS) * (int)sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (d->ln) {
    auto kern = gemm_dw_kernel<true, AT>;
    FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(256), lds, st, a, ntiles);
  } else {
    auto kern = gemm_dw_kernel<false, AT>;
    FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(256), lds, st, a, ntiles);
  }
  FZ_LAUNCH_CHECK();
  FinishJob fj = finish_job(FK_DW, kDwRow / 16);
  fj.u.dw = FinDw{(const float*)d->wpart, d->ln ? d->ln_g : (const float*)nullptr, d->ln_b, d->gw, d->gb, d->ln ? d->gln : (float*)nullptr,
                  rows, d->ldgw > 0 ? d->ldgw : 32};
  return finish_run(&fj, 1, st);
}

extern "C" int fz_gemm_dw_rows(int B, int64_t V) {
  const int64_t nt = fz_mlp_partials(B, V);
  const int wgs = knob_gemm_dw_wgs();
  return (int)(nt < wgs ? nt : wgs);
}
extern "C" int64_t fz_gemm_dw_workspace_bytes(int B, int64_t V) {
  return (int64_t)fz_gemm_dw_rows(B, V) * kDwRow * (int64_t)sizeof(float);
}

extern "C" int fz_gemm_dw(const fz_gemm_dw_desc* d, fz_stream_t stream) {
  if (!d) return fail(FZ_E_ARG, "fz_gemm_dw: null descriptor");
  if (!d->g || !d->q || !d->w || !d->y || !d->wpart || !d->gw) return fail(FZ_E_ARG, "fz_gemm_dw: null pointer");
  if (d->ln && (!d->stats || !d->ln_g || !d->ln_b || !d->gln)) return fail(FZ_E_ARG, "fz_gemm_dw: the LayerNorm form needs stats, gamma, beta, gln");
  if (!d->ln && d->gadd) return fail(FZ_E_UNSUPPORTED, "fz_gemm_dw: gadd only with the LayerNorm backward");
  if (d->C != 32) return fail(FZ_E_UNSUPPORTED, "fz_gemm_dw: needs C == 32");
  if (d->ldgw != 0 && d->ldgw < 32) return fail(FZ_E_ARG, "fz_gemm_dw: ldgw must be 0 (= 32) or >= 32");
  if (d->ldw != 0 && d->ldw < 32) return fail(FZ_E_ARG, "fz_gemm_dw: ldw must be 0 (= 32) or >= 32");
  if (d->B < 1 || d->V < 1 || d->V % 4 != 0 || d->V > ((int64_t)1 << 27)) return fail(FZ_E_UNSUPPORTED, "fz_gemm_dw: needs B >= 1, V % 4 == 0, V <= 2^27");
  if (d->act_dtype == FZ_STORE_F32) return gemm_dw_launch<float>(d, stream);
  if (d->act_dtype == FZ_STORE_BF16) return gemm_dw_launch<bf16>(d, stream);
  return fail(FZ_E_ARG, "fz_gemm_dw: act_dtype must be FZ_STORE_F32 or FZ_STORE_BF16");
}

extern "C" int fz_mlp_chain(const fz_mlp_desc* d, fz_stream_t stream) {
  if (!d) return fail(FZ_E_ARG, "fz_mlp_chain: null descriptor");
  if (d->act_dtype == FZ_STORE_F32) return mlp_launch<float>(d, stream);
  if (d->act_dtype == FZ_STORE_BF16) return mlp_launch<bf16>(d, stream);
  return fail(FZ_E_ARG, "fz_mlp_chain: act_dtype must be FZ_STORE_F32 or FZ_STORE_BF16");
}
