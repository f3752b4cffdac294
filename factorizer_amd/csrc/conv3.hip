// conv3.hip — the stem: Conv3d(kernel 3, padding 1, stride 1, bias optional) forward and
// weight gradient as direct implicit-GEMM kernels on the fp32 matrix cores.
// Replaces nn.Conv3d(in, 32, k=3, p=1, bias=False) of the Factorizer stem
// (factorizer/factorizer.py:145-149 → unet.py:231,261): C_in = 4 is far too thin for an im2col
// GEMM (K = 108), and MIOpen's fallback was the single most expensive kernel of a training step
// (profiles/r01_p1).
//
// Forward: one wave = 32 output channels x 128 consecutive voxels (4 per lane, W % 4 == 0).
// The 27 taps are unrolled at compile time; for every (channel pair, kd, kh) the lane loads ONE
// aligned 16-byte vector plus its two neighbours and derives the kw = 0,1,2 operands from them in
// registers (3 K-steps per load group), all addresses clamped and masked branch-free.
//
// Weight gradient: one wave stages, per 32-voxel tile along W, the 32 gY rows and the 9 (kd,kh)
// halo rows of every input channel (34 floats each) in LDS; the 27·C_in shifted operands are then
// LDS reads at lane-dependent offsets, so the input is fetched 9x (L1/L2 hits) instead of 27x and
// gY exactly once.  Partial sums per workgroup, deterministic reduction (wgrad.hip's reducer).
#include "gemm_bx.h"   // split-bf16 operand helpers (bx_split, bx_mfma); brings fz_common.h and f32x16

namespace fz {

// AT = storage type of the activation tensors x / y / gy (float or bf16)
template <typename AT>
struct Conv3ArgsT {
  const AT* x;        // (B, Cin, D, H, W)
  const float* w;     // (M, Cin, 3, 3, 3)
  const float* bias;  // (M) or null
  AT* y;              // (B, M, D, H, W)
  int B, Cin, M, D, H, W;
  BlockPrologueArgs pro;   // (PRO, M == 32) the consuming FactorizerBlock's LayerNorm 1 + in_proj + ReLU on the output tile (gemm_bx.h)
};

// ---------------------------------------------------------------------------------------------
template <int MB, typename AT>
__global__ __launch_bounds__(256, 2) void conv3_fwd_kernel(Conv3ArgsT<AT> p) {
  extern __shared__ __attribute__((aligned(16))) float As[];  // [Cin/2 * 27][MB][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int64_t V = (int64_t)p.D * p.H * p.W;
  const int tiles_per_sample = (int)((V + 511) / 512);
  const int b = blockIdx.x / tiles_per_sample;
  const int64_t n0 = ((int64_t)(blockIdx.x % tiles_per_sample) * 4 + wave) * 128;
  const int m0 = blockIdx.y * 32 * MB;
  const int ncp = p.Cin / 2;
  const int nA = ncp * 27;

  for (int base = threadIdx.x; base < nA * MB * 64; base += blockDim.x * 8) {
    float tmp[8];
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
      const int idx = base + uu * blockDim.x;
      const bool in = idx < nA * MB * 64;
      const int ii = in ? idx : 0;
      const int l = ii & 63, mb = (ii >> 6) % MB, a = ii / (64 * MB);
      const int m = m0 + mb * 32 + (l & 31);
      const int c = 2 * (a / 27) + (l >> 5), tap = a % 27;
      const int mc = m < p.M ? m : p.M - 1;
      const float wv = p.w[((int64_t)mc * p.Cin + c) * 27 + tap];
      tmp[uu] = (in && m < p.M) ? wv : 0.f;
    }
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
      const int idx = base + uu * blockDim.x;
      if (idx < nA * MB * 64) As[idx] = tmp[uu];
    }
  }
  __syncthreads();

  const int64_t col = n0 + 4 * j;
  const bool col_ok = col < V;
  const int64_t cc = col_ok ? col : 0;
  const int w0 = (int)(cc % p.W);
  const int h0 = (int)((cc / p.W) % p.H);
  const int d0 = (int)(cc / ((int64_t)p.W * p.H));
  const bool lok = w0 > 0, rok = w0 + 4 < p.W;

  f32x16 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][q][r] = 0.f;

  for (int cp = 0; cp < ncp; ++cp) {
    const AT* plane = p.x + ((int64_t)b * p.Cin + 2 * cp + h) * V;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int zd = d0 + kd - 1, zh = h0 + kh - 1;
        const bool ok = col_ok && zd >= 0 && zd < p.D && zh >= 0 && zh < p.H;
        const int zdc = zd < 0 ? 0 : (zd >= p.D ? p.D - 1 : zd);
        const int zhc = zh < 0 ? 0 : (zh >= p.H ? p.H - 1 : zh);
        const AT* row = plane + ((int64_t)zdc * p.H + zhc) * p.W;
        const float4 t = ld4(row + w0);
        const float lf = aget(row + (lok ? w0 - 1 : w0));
        const float rt = aget(row + (rok ? w0 + 4 : w0));
        const float c0 = ok ? t.x : 0.f, c1 = ok ? t.y : 0.f, c2 = ok ? t.z : 0.f, c3 = ok ? t.w : 0.f;
        const float l0 = (ok && lok) ? lf : 0.f, r0 = (ok && rok) ? rt : 0.f;
        const float bv[3][4] = {{l0, c0, c1, c2}, {c0, c1, c2, c3}, {c1, c2, c3, r0}};
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int a = cp * 27 + (kd * 3 + kh) * 3 + kw;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const float av = As[(a * MB + mb) * 64 + lane];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc[mb][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[kw][q], acc[mb][q], 0, 0, 0);
          }
        }
      }
    }
  }
  if (!col_ok) return;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m >= p.M) continue;
      const float bs = p.bias ? p.bias[m] : 0.f;
      st4(p.y + ((int64_t)b * p.M + m) * V + col,
          make_float4(acc[mb][0][r] + bs, acc[mb][1][r] + bs, acc[mb][2][r] + bs, acc[mb][3][r] + bs));
    }
}

// ---------------------------------------------------------------------------------------------
// The same forward on the bf16 matrix pipe with split fp32 operands (gemm_bx.hip: six exact bf16 products per fp32
// product, fp32 accuracy at 6/16 of the fp32-MFMA time).  The fp32 kernel above walks K = 27·C_in as K-steps of two
// (channel 2cp + h, tap): eight consecutive steps are ONE v_mfma_f32_32x32x16_bf16 K-step whose element e of lane half h
// is step 8g + e — any assignment of reduction indices to (half, element) slots is valid as long as both operands use
// the same one, so the operand registers of the fp32 form are simply packed eight at a time.  CP = C_in / 2 is a
// compile-time constant (the whole tap loop is unrolled: groups straddle channel pairs).  Weights are split once per
// workgroup into LDS: As[group][row block][term][lane] x 16 B; the tail group is zero-padded.
template <int MB, int CP, typename AT, bool PRO = false>
__global__ __launch_bounds__(256, 2) void conv3_fwd_bx_kernel(Conv3ArgsT<AT> p) {
  static_assert(!PRO || MB == 1, "the block prologue needs all 32 output channels in one row block");
  constexpr int NTA = BxTerms<AT>::A, NTB = bx_terms_b<AT>(BXPRO_NONE);
  constexpr int NS = CP * 27;                 // K-steps of two
  constexpr int NG = (NS + 7) / 8;            // groups of eight steps
  __shared__ __attribute__((aligned(16))) __bf16 As[NG * MB * NTA * 64 * 8];
  __shared__ float sBias[32 * MB];
  __shared__ __attribute__((aligned(16))) __bf16 Apro[PRO ? 2 * 3 * 64 * 8 : 8];
  __shared__ float twp[PRO ? 32 : 1];
  if constexpr (PRO) ln_inproj_stage(Apro, twp, p.pro, (int)threadIdx.x, 256);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int64_t V = (int64_t)p.D * p.H * p.W;
  const int tiles_per_sample = (int)((V + 511) / 512);
  const int b = blockIdx.x / tiles_per_sample;
  const int64_t n0 = ((int64_t)(blockIdx.x % tiles_per_sample) * 4 + wave) * 128;
  const int m0 = blockIdx.y * 32 * MB;
  const int Cin = 2 * CP;

  for (int idx = threadIdx.x; idx < NG * MB * 64; idx += 256) {
    const int l = idx & 63, mb = (idx >> 6) % MB, g = idx / (64 * MB);
    const int m = m0 + mb * 32 + (l & 31);
    const int mc = m < p.M ? m : p.M - 1;
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int a = 8 * g + e;
      const int ac = a < NS ? a : NS - 1;
      const int c = 2 * (ac / 27) + (l >> 5), tap = ac % 27;
      const float t = p.w[((int64_t)mc * Cin + c) * 27 + tap];
      wv[e] = (a < NS && m < p.M) ? t : 0.f;
    }
    bx8 t3[NTA];
    bx_split<NTA>(wv, t3);
#pragma unroll
    for (int i = 0; i < NTA; ++i) *reinterpret_cast<bx8*>(&As[(((g * MB + mb) * NTA + i) * 64 + l) * 8]) = t3[i];
  }
  if (threadIdx.x < 32 * MB) {
    const int m = m0 + threadIdx.x;
    sBias[threadIdx.x] = (p.bias != nullptr && m < p.M) ? p.bias[m] : 0.f;
  }
  __syncthreads();

  const int64_t col = n0 + 4 * j;
  const bool col_ok = col < V;
  const int64_t cc = col_ok ? col : 0;
  const int w0 = (int)(cc % p.W);
  const int h0 = (int)((cc / p.W) % p.H);
  const int d0 = (int)(cc / ((int64_t)p.W * p.H));
  const bool lok = w0 > 0, rok = w0 + 4 < p.W;

  f32x16 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][q][r] = 0.f;

  float xg[4][8];   // the group being collected: [voxel q][step in group]
#pragma unroll
  for (int cp = 0; cp < CP; ++cp) {
    const AT* plane = p.x + ((int64_t)b * Cin + 2 * cp + h) * V;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int zd = d0 + kd - 1, zh = h0 + kh - 1;
        const bool ok = col_ok && zd >= 0 && zd < p.D && zh >= 0 && zh < p.H;
        const int zdc = zd < 0 ? 0 : (zd >= p.D ? p.D - 1 : zd);
        const int zhc = zh < 0 ? 0 : (zh >= p.H ? p.H - 1 : zh);
        const AT* row = plane + ((int64_t)zdc * p.H + zhc) * p.W;
        const float4 t = ld4(row + w0);
        const float lf = aget(row + (lok ? w0 - 1 : w0));
        const float rt = aget(row + (rok ? w0 + 4 : w0));
        const float c0 = ok ? t.x : 0.f, c1 = ok ? t.y : 0.f, c2 = ok ? t.z : 0.f, c3 = ok ? t.w : 0.f;
        const float l0 = (ok && lok) ? lf : 0.f, r0 = (ok && rok) ? rt : 0.f;
        const float bv[3][4] = {{l0, c0, c1, c2}, {c0, c1, c2, c3}, {c1, c2, c3, r0}};
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int a = cp * 27 + (kd * 3 + kh) * 3 + kw;
#pragma unroll
          for (int q = 0; q < 4; ++q) xg[q][a & 7] = bv[kw][q];
          if ((a & 7) == 7 || a == NS - 1) {
            const int g = a >> 3;
            if (a == NS - 1) {
#pragma unroll
              for (int e = (NS - 1) % 8 + 1; e < 8; ++e)
#pragma unroll
                for (int q = 0; q < 4; ++q) xg[q][e] = 0.f;   // zero-padded tail (its weights are zero too)
            }
            bx8 bop[4][NTB];
#pragma unroll
            for (int q = 0; q < 4; ++q) bx_split<NTB>(xg[q], bop[q]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
              bx8 aop[NTA];
#pragma unroll
              for (int i = 0; i < NTA; ++i)
                aop[i] = *reinterpret_cast<const bx8*>(&As[(((g * MB + mb) * NTA + i) * 64 + lane) * 8]);
#pragma unroll
              for (int q = 0; q < 4; ++q) bx_mfma<NTA, NTB>(acc[mb][q], aop, bop[q]);
            }
          }
        }
      }
    }
  }
  if (!col_ok) return;
  float yv[PRO ? 4 : 1][16];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int m = m0 + mb * 32 + rl;
      const float bs = sBias[mb * 32 + rl];
      const float4 o = make_float4(acc[mb][0][r] + bs, acc[mb][1][r] + bs, acc[mb][2][r] + bs, acc[mb][3][r] + bs);
      if (m < p.M) st4(p.y + ((int64_t)b * p.M + m) * V + col, o);
      if constexpr (PRO) {   // bf16 storage: the block's first layer sees the STORED values
        constexpr bool R = sizeof(AT) == 2;
        yv[0][r] = R ? (float)(AT)o.x : o.x; yv[1][r] = R ? (float)(AT)o.y : o.y;
        yv[2][r] = R ? (float)(AT)o.z : o.z; yv[3][r] = R ? (float)(AT)o.w : o.w;
      }
    }
  if constexpr (PRO) {
    float mu[4], rs[4];
    f32x16 tacc[4];
    ln_inproj_tile<4>(yv, Apro, p.pro.ln_eps, lane, mu, rs, tacc);
    AT* tb = reinterpret_cast<AT*>(p.pro.t) + (int64_t)b * 32 * V + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
      const float add = twp[rl];
      st4(tb + (int64_t)rl * V, make_float4(fmaxf(tacc[0][r] + add, 0.f), fmaxf(tacc[1][r] + add, 0.f), fmaxf(tacc[2][r] + add, 0.f),
                                            fmaxf(tacc[3][r] + add, 0.f)));
    }
    if (h == 0) {
      float* so = p.pro.stats + (int64_t)b * 2 * V + col;
      *reinterpret_cast<float4*>(so) = make_float4(mu[0], mu[1], mu[2], mu[3]);
      *reinterpret_cast<float4*>(so + V) = make_float4(rs[0], rs[1], rs[2], rs[3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: GW[m][(ci, kd, kh, kw)] = Σ_{b,v} gY[b,m,v] · x[b,ci,v + off(kd,kh,kw)]
template <typename AT>
struct Conv3WgradArgsT {
  const AT* gy;     // (B, M, D, H, W)
  const AT* x;      // (B, Cin, D, H, W)
  float* part;      // [nchunk][M][27*Cin]
  float* part_bias; // [nchunk][M]
  int B, Cin, M, D, H, W;
  int tiles_per_unit;
};

// halo row stride: [3] = w0-1, [4..35] = the 32 interior voxels (16-byte aligned), [36] = w0+32.  The operand reads are
// one float per lane, lane = column (ci, kd, kh, kw): bank = (row·kXs + kw + const) mod 32 with row = (ci, kd, kh).  At
// stride 40 the rows fall into 4 bank classes (8·row mod 32) — 32 lanes on 12 banks, SQ_LDS_BANK_CONFLICT 0.57 of the LDS
// cycles in round 2; at 44 (12·row mod 32: eight classes 4 apart, 3 taps each) the first 24 lanes are conflict-free and
// the last 8 of a 32-column block share a bank with one other lane.
constexpr int kXs = 44;

// BX: products on the bf16 matrix pipe from split fp32 operands (see conv3_fwd_bx_kernel): the 32 voxels of a tile are two
// K-steps of 16, element e of lane half h = voxel 16g + 8h + e — for gY eight contiguous floats of the lane's row.
template <int KB, typename AT, bool BX = false>  // KB = number of 32-column blocks covering 27*Cin (4 for Cin = 4)
__global__ __launch_bounds__(256) void conv3_wgrad_kernel(Conv3WgradArgsT<AT> a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nrow = a.Cin * 9;                      // halo rows per tile (<= 36: host-checked)
  const int per_wave = 32 * 36 + 36 * kXs;
  float* Pt = lds + wave * per_wave;               // [32][36] gY tile
  float* Xs = Pt + 32 * 36;                        // [Cin*9][40] halo rows
  const int K = 27 * a.Cin;
  const int m0 = blockIdx.y * 32;
  const int64_t V = (int64_t)a.D * a.H * a.W;
  const int64_t tiles_per_sample = V / 32;         // W % 32 == 0: a tile never crosses a row
  const int64_t total = tiles_per_sample * a.B;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
  const int64_t t_begin = unit * a.tiles_per_unit, t_end = min(t_begin + a.tiles_per_unit, total);
  const int c = lane & 31, h = lane >> 5;

  // per-lane operand offsets inside Xs for each column block: column k = jb*32 + c
  int xoff[KB];
#pragma unroll
  for (int jb = 0; jb < KB; ++jb) {
    const int k = jb * 32 + c;
    const int kc = k < K ? k : 0;
    const int ci = kc / 27, tap = kc % 27;
    xoff[jb] = (ci * 9 + tap / 3) * kXs + 3 + (tap % 3);  // row (ci,kd,kh), shift kw
  }
  f32x16 acc[KB];
#pragma unroll
  for (int jb = 0; jb < KB; ++jb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[jb][r] = 0.f;
  float psum = 0.f;

  // Static per-lane staging plan (tile-invariant): 5 interior chunks (row q>>3, 16-byte chunk q&7 of
  // the 32 aligned interior voxels; 36 rows x 8 = 288 chunks) and 2 edge voxels (w0-1 / w0+32 of
  // row id>>1; 72 of them).  The next tile's 4 + 5 + 2 loads are issued before the MFMA loop of
  // the current one (the old per-element halo gather — 20 dependent scalar loads with div/mod
  // index math per lane and tile, not prefetched — was 2/3 of the kernel's time).
  float4 pv[4], hv[5];
  float ev[2];
  // Tile coordinates (sample, plane, row, first voxel of the 32 along W) are WAVE-UNIFORM and advance by one tile per call:
  // carried as scalar state and stepped with compares.  Every load address is `scalar tile base + a lane offset that does
  // not depend on the tile` (the halo taps as signed element offsets, biased by one plane row + 1 so that they are unsigned),
  // and whether a tap falls outside the volume is a bit test of a per-lane tap mask against four scalar border flags.
  // (Rounds 1-3 re-derived everything from the tile number per lane and tile — five 64-bit divisions, clamps and 64-bit
  // multiplies for each of the 11 loads: several hundred vector instructions, more than the tile's arithmetic.)
  int tb, td, th, tw;
  {
    const unsigned tps32 = (unsigned)tiles_per_sample;                       // < 2^26 (V < 2^31, host-checked)
    const unsigned t0 = (unsigned)__builtin_amdgcn_readfirstlane((int)t_begin);   // < 2^31 (host-checked)
    const unsigned rem = t0 % tps32, row = rem / (unsigned)(a.W / 32);
    tb = (int)(t0 / tps32);
    tw = (int)(rem % (unsigned)(a.W / 32)) * 32;
    th = (int)(row % (unsigned)a.H);
    td = (int)(row / (unsigned)a.H);
  }
  const unsigned bias = (unsigned)((a.H + 1) * a.W + 1);                     // -(smallest tap offset)
  unsigned goff[4], xo[5], eo[2];     // element offsets of the lane's loads relative to the tile's scalar bases
  unsigned xsafe[5], esafe[2];        // the centre tap of the same channel: always inside the volume
  unsigned mask = 0;                  // bits 4i..4i+3: load i is tap (kd == 0, kd == 2, kh == 0, kh == 2)
  bool gvalid[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + (lane >> 3) + 8 * i;
    gvalid[i] = m < a.M;
    goff[i] = (unsigned)((m < a.M ? m : a.M - 1) * V + (lane & 7) * 4);      // < 2^30 elements (host-checked)
  }
  auto tap_bits = [&](int kd, int kh) { return (unsigned)((kd == 0) | ((kd == 2) << 1) | ((kh == 0) << 2) | ((kh == 2) << 3)); };
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int q = lane + 64 * i;
    const int rr = min(q >> 3, nrow - 1), cq = q & 7;
    const int ci = rr / 9, kd = (rr % 9) / 3, kh = rr % 3;
    xo[i] = (unsigned)(ci * V + ((kd - 1) * a.H + (kh - 1)) * a.W + cq * 4 + (int64_t)bias);
    xsafe[i] = (unsigned)(ci * V + cq * 4 + (int64_t)bias);
    mask |= tap_bits(kd, kh) << (4 * i);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = lane + 64 * i;
    const int rr = min(id >> 1, nrow - 1), side = id & 1;
    const int ci = rr / 9, kd = (rr % 9) / 3, kh = rr % 3;
    eo[i] = (unsigned)(ci * V + ((kd - 1) * a.H + (kh - 1)) * a.W + (side ? 32 : -1) + (int64_t)bias);
    esafe[i] = (unsigned)(ci * V + (int64_t)bias);
  }
  // the two edge voxels: their own tap bits + side bits (left edge outside when w0 == 0, right edge when w0 + 32 == W)
  unsigned emask = 0, enever = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = lane + 64 * i;
    const int rr = min(id >> 1, nrow - 1), side = id & 1;
    const int kd = (rr % 9) / 3, kh = rr % 3;
    emask |= (tap_bits(kd, kh) | (side ? 0x20u : 0x10u)) << (8 * i);
    enever |= ((id >> 1) < nrow ? 0u : 1u) << i;
  }
  unsigned xnever = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) xnever |= (((lane + 64 * i) >> 3) < nrow ? 0u : 1u) << i;
  auto issue = [&]() {
    const int b = tb, w0 = tw, h0 = th, d0 = td;
    // border flags of this tile, in tap-bit order (kd == 0, kd == 2, kh == 0, kh == 2), replicated per load
    const unsigned f4 = (unsigned)((d0 == 0) | ((d0 == a.D - 1) << 1) | ((h0 == 0) << 2) | ((h0 == a.H - 1) << 3));
    const unsigned f20 = f4 * 0x11111u;
    const unsigned fe = (f4 | ((w0 == 0) << 4) | ((w0 + 32 == a.W) << 5)) * 0x101u;
    const int64_t n0 = ((int64_t)d0 * a.H + h0) * a.W + w0;
    const AT* gbase = a.gy + (int64_t)b * a.M * V + n0;
    const AT* xbase = a.x + (int64_t)b * a.Cin * V + n0 - (int64_t)bias;
    tw += 32;                                                                // the NEXT call's tile
    if (tw == a.W) { tw = 0; if (++th == a.H) { th = 0; if (++td == a.D) { td = 0; ++tb; } } }
    const unsigned bad = mask & f20, ebad = emask & fe;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = ld4(gbase + goff[i]);
      pv[i] = gvalid[i] ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const bool ok = ((bad >> (4 * i)) & 0xfu) == 0 && ((xnever >> i) & 1u) == 0;
      const float4 v = ld4(xbase + (ok ? xo[i] : xsafe[i]));
      hv[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = ((ebad >> (8 * i)) & 0x3fu) == 0 && ((enever >> i) & 1u) == 0;
      const float v = aget(xbase + (ok ? eo[i] : esafe[i]));
      ev[i] = ok ? v : 0.f;
    }
  };

  if (t_begin < t_end) issue();
  for (int64_t t = t_begin; t < t_end; ++t) {
    // registers -> LDS
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(Pt + ((lane >> 3) + 8 * i) * 36 + (lane & 7) * 4) = pv[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int q = lane + 64 * i;
      if (q < 288) *reinterpret_cast<float4*>(Xs + (q >> 3) * kXs + 4 + (q & 7) * 4) = hv[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = lane + 64 * i;
      if (id < 72) Xs[(id >> 1) * kXs + ((id & 1) ? 36 : 3)] = ev[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (t + 1 < t_end) issue();  // in flight during the MFMA loop
    if constexpr (BX) {
      constexpr int NTA = BxTerms<AT>::A;   // both operands are activations; gY may carry rounding of a bf16 tensor only
      constexpr int NT = sizeof(AT) == 4 ? 3 : 1;
      (void)NTA;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float a8[8];
        const float4 a0 = *reinterpret_cast<const float4*>(Pt + c * 36 + 16 * g + 8 * h);
        const float4 a1 = *reinterpret_cast<const float4*>(Pt + c * 36 + 16 * g + 8 * h + 4);
        a8[0] = a0.x; a8[1] = a0.y; a8[2] = a0.z; a8[3] = a0.w; a8[4] = a1.x; a8[5] = a1.y; a8[6] = a1.z; a8[7] = a1.w;
        psum += ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
        bx8 aop[NT];
        bx_split<NT>(a8, aop);
#pragma unroll
        for (int jb = 0; jb < KB; ++jb) {
          float b8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) b8[e] = Xs[xoff[jb] + 16 * g + 8 * h + e];
          bx8 bop[NT];
          bx_split<NT>(b8, bop);
          bx_mfma<NT, NT>(acc[jb], aop, bop);
        }
      }
    } else {
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const float av = Pt[c * 36 + 2 * s + h];
      psum += av;
#pragma unroll
      for (int jb = 0; jb < KB; ++jb) {
        const float bv = Xs[xoff[jb] + 2 * s + h];
        acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[jb], 0, 0, 0);
      }
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }

  // reduce the 4 waves, write the workgroup's partial block
  __syncthreads();
  float* red = lds;
#pragma unroll
  for (int jb = 0; jb < KB; ++jb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      red[wave * 1024 + row * 32 + c] = acc[jb][r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += 256) {
      const float sres = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
      const int m = m0 + (e >> 5), k = jb * 32 + (e & 31);
      if (m < a.M && k < K) a.part[((int64_t)blockIdx.x * a.M + m) * K + k] = sres;
    }
    __syncthreads();
  }
  {
    const float sres = psum + __shfl_xor(psum, 32, 64);
    if (h == 0) red[wave * 32 + c] = sres;
    __syncthreads();
    if (threadIdx.x < 32) {
      const int m = m0 + threadIdx.x;
      if (m < a.M)
        a.part_bias[(int64_t)blockIdx.x * a.M + m] =
            (red[threadIdx.x] + red[32 + threadIdx.x]) + (red[64 + threadIdx.x] + red[96 + threadIdx.x]);
    }
  }
}

}  // namespace fz

using namespace fz;

template <typename AT>
static int conv3_fwd_launch(const void* x, const float* w, const float* bias, void* y, int B, int Cin, int M, int D,
                            int H, int W, int products, const fz_block_prologue* pro, fz_stream_t stream) {
  Conv3ArgsT<AT> p{(const AT*)x, w, bias, (AT*)y, B, Cin, M, D, H, W, BlockPrologueArgs{}};
  if (pro) {
    if (!(Cin == 4 && M == 32 && products_split(products)))
      return fail(FZ_E_UNSUPPORTED, "fz_conv3_fwd2: the block prologue needs C_in = 4, 32 output channels, split-bf16 products (fz_conv3_prologue_supported)");
    p.pro = BlockPrologueArgs{pro->ln_g, pro->ln_b, pro->ln_eps, pro->w, pro->t, pro->stats};
  }
  const int64_t V = (int64_t)D * H * W;
  const int mblocks = (M + 31) / 32;
  const int MB = mblocks >= 2 ? 2 : 1;
  const size_t lds = (size_t)(Cin / 2) * 27 * MB * 64 * sizeof(float);
  if (lds > 65536) return fail(FZ_E_UNSUPPORTED, "fz_conv3_fwd: C_in too large for the stem kernel");
  dim3 grid((unsigned)(((V + 511) / 512) * B), (unsigned)((mblocks + MB - 1) / MB)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 4 && products_split(products)) {   // the stem of the README model: split-bf16 form (FZ_PRODUCTS_FP32_MFMA: fp32 MFMAs)
    if (MB == 2) hipLaunchKernelGGL((conv3_fwd_bx_kernel<2, 2, AT>), grid, block, 0, st, p);
    else if (pro) hipLaunchKernelGGL((conv3_fwd_bx_kernel<1, 2, AT, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv3_fwd_bx_kernel<1, 2, AT>), grid, block, 0, st, p);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }
  if (MB == 2) hipLaunchKernelGGL((conv3_fwd_kernel<2, AT>), grid, block, lds, st, p);
  else hipLaunchKernelGGL((conv3_fwd_kernel<1, AT>), grid, block, lds, st, p);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_conv3_prologue_supported(int Cin, int M, int W, int products) {
  return (Cin == 4 && M == 32 && W >= 4 && (W & 3) == 0 && products_split(products)) ? 1 : 0;
}

extern "C" int fz_conv3_fwd2(const void* x, const float* w, const float* bias, void* y, int B, int Cin, int M, int D,
                             int H, int W, int act_dtype, int products, const fz_block_prologue* pro, fz_stream_t stream) {
  if (!x || !w || !y) return fail(FZ_E_ARG, "fz_conv3_fwd: null pointer");
  if (pro && (!pro->ln_g || !pro->ln_b || !pro->w || !pro->t || !pro->stats)) return fail(FZ_E_ARG, "fz_conv3_fwd2: incomplete block prologue");
  if (B < 0 || Cin < 2 || (Cin & 1) || M < 1 || D < 1 || H < 1 || W < 4 || (W & 3))
    return fail(FZ_E_UNSUPPORTED, "fz_conv3_fwd: needs even C_in and W % 4 == 0");
  if (B == 0) return FZ_OK;
  if (act_dtype == FZ_STORE_F32) return conv3_fwd_launch<float>(x, w, bias, y, B, Cin, M, D, H, W, products, pro, stream);
  if (act_dtype == FZ_STORE_BF16) return conv3_fwd_launch<bf16>(x, w, bias, y, B, Cin, M, D, H, W, products, pro, stream);
  return fail(FZ_E_ARG, "fz_conv3_fwd: bad act_dtype");
}

extern "C" int fz_conv3_fwd(const void* x, const float* w, const float* bias, void* y, int B, int Cin, int M, int D,
                            int H, int W, int act_dtype, int products, fz_stream_t stream) {
  return fz_conv3_fwd2(x, w, bias, y, B, Cin, M, D, H, W, act_dtype, products, nullptr, stream);
}

static int conv3_units(int64_t total_tiles, int* tiles_per_unit) {
  int64_t units_target = 4096;
  int64_t tpu = (total_tiles + units_target - 1) / units_target;
  if (tpu < 1) tpu = 1;
  *tiles_per_unit = (int)tpu;
  const int64_t units = (total_tiles + tpu - 1) / tpu;
  return (int)((units + 3) / 4);
}

// workspace floats: nchunk*(M*K + M);  nchunk = fz_conv3_wgrad_chunks(...)
extern "C" int fz_conv3_wgrad_chunks(int B, int D, int H, int W) {
  int tpu;
  return conv3_units(((int64_t)D * H * W / 32) * B, &tpu);
}

template <typename AT>
static int conv3_wgrad_launch(const void* gy, const void* x, float* part, float* part_bias, int B, int Cin,
                              int M, int D, int H, int W, int products, fz_stream_t stream) {
  Conv3WgradArgsT<AT> a{(const AT*)gy, (const AT*)x, part, part_bias, B, Cin, M, D, H, W, 1};
  const int nchunk = conv3_units(((int64_t)D * H * W / 32) * B, &a.tiles_per_unit);
  const size_t lds = (size_t)4 * (32 * 36 + 36 * kXs) * sizeof(float);
  const size_t lds_red = 4096 * sizeof(float);
  dim3 grid(nchunk, (M + 31) / 32), block(256);
  if (products_split(products))
    hipLaunchKernelGGL((conv3_wgrad_kernel<4, AT, true>), grid, block, lds > lds_red ? lds : lds_red, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((conv3_wgrad_kernel<4, AT, false>), grid, block, lds > lds_red ? lds : lds_red, (hipStream_t)stream, a);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_conv3_wgrad_partials(const void* gy, const void* x, float* part, float* part_bias, int B, int Cin,
                                       int M, int D, int H, int W, int act_dtype, int products, fz_stream_t stream) {
  if (!gy || !x || !part || !part_bias) return fail(FZ_E_ARG, "fz_conv3_wgrad: null pointer");
  if (B < 1 || Cin < 1 || M < 1 || (W % 32) || 27 * Cin > 128 || Cin * 9 > 36)
    return fail(FZ_E_UNSUPPORTED, "fz_conv3_wgrad: needs W % 32 == 0 and C_in <= 4");
  {
    const int64_t V = (int64_t)D * H * W;   // 32-bit tile arithmetic and 32-bit lane offsets (elements) in the kernel
    if ((V / 32) * B >= ((int64_t)1 << 31) || (int64_t)(M < 32 ? 32 : M) * V >= ((int64_t)1 << 32) ||
        (int64_t)Cin * V + (int64_t)(H + 1) * W + 64 >= ((int64_t)1 << 32))
      return fail(FZ_E_UNSUPPORTED, "fz_conv3_wgrad: volume too large for the kernel's 32-bit offsets");
  }
  if (act_dtype == FZ_STORE_F32) return conv3_wgrad_launch<float>(gy, x, part, part_bias, B, Cin, M, D, H, W, products, stream);
  if (act_dtype == FZ_STORE_BF16) return conv3_wgrad_launch<bf16>(gy, x, part, part_bias, B, Cin, M, D, H, W, products, stream);
  return fail(FZ_E_ARG, "fz_conv3_wgrad: bad act_dtype");
}
