// nmf_cf.hip — FactMixer core on channels-first tensors: shifted-window matricize → NMF →
// inverse matricize in ONE kernel per window (hot shape: head_dim 8, patch 8x8x8).
//
// Replaces the chain SWMatricize.forward → NMF.forward → SWMatricize.inverse_forward
// (factorizer/factorizer.py:41-50; operations.py:417-434; matrix_factorization.py:514-546):
// the (W·B·h, G, 8, 512) matricized tensors are never materialised.  A wave gathers its 8x512
// matrix straight from t (B, C, D, H, W) with the window's cyclic shift, runs the per-wave NMF
// program of nmf_core.h with X in registers, and scatters u vᵀ back to the same voxels of the
// averaged output a:  window 0 stores (0 + z_0), window w>0 adds z_w, the last window divides by
// the number of windows — the reference's ((0.0 + z_0) + z_1 + …) / W order (operations.py:426-433).
// Windows are separate launches on one stream, so the accumulation is deterministic.
//
// Lane map: lane l = (p0 & 3 = l>>4, p1 = (l>>1)&7, half = l&1); for every channel dd and
// p0-group jp the lane moves one 16-byte vector = voxels p2 = 4·half..4·half+3 of patch row
// (p0 = 4·jp + (l>>4), p1).  Column index of local element (jp, e): n = (p0·8 + p1)·8 + 4·half + e.
#include "nmf_cf.h"

namespace fz {

// ---- in-launch hand-off between the two shift windows (two-window kernels below) -----------------------------------
// Window 0 writes its part of the running average WRITE-THROUGH (sc1) and signals a per-(slice, patch-plane) counter;
// window 1 polls the counters of the planes its voxels lie in and then reads the running average with sc1 loads
// (cdna_hip_programming.md Guideline 16, recipe R1 in its counter form; MI355X_MICROARCH.md § visibility, table row
// "agent-scope atomic adds by one lane of each storing workgroup / sc1 poll / barrier / sc1 dwordx4 stores, whole lines /
// sc1 dwordx4 loads").  Placement-independent: nothing assumes a dispatch order or a workgroup -> XCD map.
typedef int cf_v4i __attribute__((__vector_size__(16)));
typedef int cf_v2i __attribute__((__vector_size__(8)));
enum { CF_PLAIN = 0, CF_PRODUCE = 1, CF_CONSUME = 2 };  // role of a tile in the two-window launch

struct CfSync {
  __amdgpu_buffer_rsrc_t rsrc;   // the tensor that is handed over (out / gt), as a buffer (32-bit byte offsets)
  unsigned* done;                // [slice][patch-plane] completed window-0 tiles
  unsigned* timeout;             // set to a code when a bounded spin gives up (never, unless the protocol is broken)
  int need;                      // tiles per (slice, plane)
};
// (the counters a tile signals / waits for travel as plain ints beside the struct: a struct with a buffer resource in it is
// not split into registers by the optimiser, and fields written per tile would live in scratch)

__device__ __forceinline__ float4 cf_ld4_sc1(const CfSync& y, const float*, int64_t eoff) {
  const cf_v4i r = __builtin_amdgcn_raw_buffer_load_b128(y.rsrc, (int)(eoff * 4), 0, 16);
  return make_float4(__int_as_float(r[0]), __int_as_float(r[1]), __int_as_float(r[2]), __int_as_float(r[3]));
}
__device__ __forceinline__ float4 cf_ld4_sc1(const CfSync& y, const bf16*, int64_t eoff) {
  const cf_v2i r = __builtin_amdgcn_raw_buffer_load_b64(y.rsrc, (int)(eoff * 2), 0, 16);
  return make_float4(__uint_as_float((unsigned)r[0] << 16), __uint_as_float((unsigned)r[0] & 0xffff0000u),
                     __uint_as_float((unsigned)r[1] << 16), __uint_as_float((unsigned)r[1] & 0xffff0000u));
}
__device__ __forceinline__ void cf_st4_sc1(const CfSync& y, const float*, int64_t eoff, float4 v) {
  const cf_v4i r = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(r, y.rsrc, (int)(eoff * 4), 0, 16);
}
__device__ __forceinline__ void cf_st4_sc1(const CfSync& y, const bf16*, int64_t eoff, float4 v) {
  const f32v4 f = {v.x, v.y, v.z, v.w};
  const bf16v4 h = __builtin_convertvector(f, bf16v4);
  __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const cf_v2i*>(&h), y.rsrc, (int)(eoff * 2), 0, 16);
}

// producer side: EVERY storing wave drains its stores, the workgroup meets, ONE lane signals
__device__ __forceinline__ void cf_publish(const CfSync& y, int idx, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(y.done + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// consumer side: ONE lane polls (relaxed, bounded), the workgroup meets, then every load of the handed-over bytes is sc1
__device__ __forceinline__ void cf_await(const CfSync& y, int ia, int ib, int tid) {
  if (tid == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(y.done + ia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)y.need ||
           __hip_atomic_load(y.done + ib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)y.need) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 24)) {   // ~ seconds: a producer with a lower ticket is resident or done, so this cannot happen
        __hip_atomic_store(y.timeout, 0xdead0001u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the sc1 loads below the poll
}

// One tile (WPB patches along W) of one window of the forward.  ROLE: CF_PLAIN — the one-window-per-launch kernel;
// CF_PRODUCE / CF_CONSUME — window 0 / window 1 of the two-window launch (block-uniform at run time there: `role`).
template <int R, int SOLVER, int WPB, bool HALF, typename AT, bool SYNC>
__device__ __forceinline__ void cf_fwd_tile_body(const AT* __restrict__ t, const float* __restrict__ u0,
                                                 const float* __restrict__ v0, AT* __restrict__ out, const CfGeom& q,
                                                 const CfTileId& id, int T, float eps, float* S, int role, const CfSync& y, int ia, int ib) {
  using TL = CfTile<WPB>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t base, V;
  unsigned off[2], off2[2];
  int lidx[2];
  cf_tile_decode<WPB>(q, id, tid, base, V, off, lidx, off2);
  const int own0 = cf_owner_lidx<WPB>(lane, wave, 0), own1 = cf_owner_lidx<WPB>(lane, wave, 1);

  float x[8][8];
#pragma unroll
  for (int dd = 0; dd < 8; ++dd)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 v = cf_ld4<HALF>(t + base + dd * V, off[k], off2[k]);
      x[dd][k * 4 + 0] = v.x; x[dd][k * 4 + 1] = v.y; x[dd][k * 4 + 2] = v.z; x[dd][k * 4 + 3] = v.w;
    }
  cf_to_owner<WPB>(S, lidx, own0, own1, x);

  CfWave w{lane};
  float u[8][R], v[8][R];
  nmf_forward_wave<8, 8, R, SOLVER>(w, u0, v0, x, u, v, 8, T, eps);

  // owner → coalesced, then the (read-modify-)write of the running window average; all loads of
  // the running sum are issued at once (one exposed round trip)
  const float dv = (float)q.divisor;
  const bool dv_pow2 = cf_pow2(dv);
  // (the plane base back in a scalar register pair: after the wave program the compiler otherwise carries it in vector
  // registers and every epilogue access pays a 64-bit vector address)
  asm volatile("" : "+s"(base));
  float4 old[8][2];
  if (q.accumulate) {
    if (SYNC && role == CF_CONSUME) {
      cf_await(y, ia, ib, tid);
#pragma unroll
      for (int dd = 0; dd < 8; ++dd)
#pragma unroll
        for (int k = 0; k < 2; ++k) old[dd][k] = cf_ld4_sc1(y, out, base + dd * V + off[k]);
    } else {
#pragma unroll
      for (int dd = 0; dd < 8; ++dd)
#pragma unroll
        for (int k = 0; k < 2; ++k) old[dd][k] = cf_ld4<HALF>(out + base + dd * V, off[k], off2[k]);
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      *reinterpret_cast<float4*>(S + c * 64 * TL::LW + own0) =
          make_float4(x[2 * s + c][0], x[2 * s + c][1], x[2 * s + c][2], x[2 * s + c][3]);
      *reinterpret_cast<float4*>(S + c * 64 * TL::LW + own1) =
          make_float4(x[2 * s + c][4], x[2 * s + c][5], x[2 * s + c][6], x[2 * s + c][7]);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float4 z = *reinterpret_cast<const float4*>(S + c * 64 * TL::LW + lidx[k]);
        float4 o;
        if (q.accumulate) {
          o = old[2 * s + c][k];
          o.x += z.x; o.y += z.y; o.z += z.z; o.w += z.w;
        } else {
          o = make_float4(0.0f + z.x, 0.0f + z.y, 0.0f + z.z, 0.0f + z.w);
        }
        if (q.divisor > 1) o = cf_divide4(o, dv, dv_pow2);
        if (SYNC && role == CF_PRODUCE) cf_st4_sc1(y, out, base + (2 * s + c) * V + off[k], o);
        else cf_st4<HALF>(out + base + (2 * s + c) * V, off[k], off2[k], o);
      }
    __syncthreads();
  }
  if (SYNC && role == CF_PRODUCE) cf_publish(y, ia, tid);
}

template <int R, int SOLVER, int WPB, bool HALF, typename AT>
// (second launch-bounds argument = minimum WAVES PER SIMD in HIP, not workgroups per CU: 8 capped the one-patch variant
// at 64 VGPRs — 170 spilled registers)
// (rank 2 needs more than the 128 registers of four waves per SIMD: two — NO variant may spill to scratch, see nmf_pcf.hip)
__global__ __launch_bounds__(WPB * 64, (WPB == 8 || R >= 2) ? 2 : 4) void nmf_cf_fwd_tile_kernel(const AT* __restrict__ t,
                                                                   const float* __restrict__ u0,
                                                                   const float* __restrict__ v0,
                                                                   AT* __restrict__ out, CfGeom q, int T, float eps,
                                                                   int xcd_remap) {
  extern __shared__ __attribute__((aligned(16))) float fz_lds_tile[];
  CfSync none{};
  cf_fwd_tile_body<R, SOLVER, WPB, HALF, AT, false>(t, u0, v0, out, q, cf_tile_id<WPB>(q, cf_logical_block(xcd_remap)), T, eps,
                                                    fz_lds_tile, CF_PLAIN, none, 0, 0);
}

// backward: gY = gather_w(ga) / W ; gt (+)= [t > 0] ∘ scatter_w(gX)
template <int R, int SOLVER, typename AT>
// (launched with 64 .. 256 threads; everything but the hot HALS rank-1 form runs one wave per SIMD rather than spill)
__global__ __launch_bounds__(256, (R == 1 && SOLVER == 1) ? 2 : 1) void nmf_cf_bwd_kernel(const AT* __restrict__ t, const float* __restrict__ u0,
                                                            const float* __restrict__ v0,
                                                            const AT* __restrict__ ga, AT* __restrict__ gt,
                                                            CfGeom q, int64_t nmat, int T, int G, float eps,
                                                            int relu_gate, int xcd_remap) {
  extern __shared__ __attribute__((aligned(16))) float fz_lds_cf[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t mat = cf_logical_block(xcd_remap) * (blockDim.x >> 6) + wave;
  if (mat >= nmat) return;
  CfWave w{lane};
  CfAddr a;
  cf_decode(q, mat, lane, a);
  Hist<8, 8, R> h;
  h.carve(fz_lds_cf + wave * Hist<8, 8, R>::floats(G), G);
  float x[8][8], g[8][8];
  cf_load(t, a, x);
  cf_load(ga, a, g);
  cf_divide(g, q.gscale_div);
  nmf_backward_wave<8, 8, R, SOLVER>(w, u0, v0, x, g, h, 8, T, G, eps, nullptr, nullptr);
#pragma unroll
  for (int dd = 0; dd < 8; ++dd)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      AT* p = gt + a.base + dd * a.V + a.off[jp];
      float r[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gv = g[dd][jp * 4 + e];
        r[e] = (!relu_gate || x[dd][jp * 4 + e] > 0.f) ? gv : 0.f;
      }
      if (q.accumulate) {
        const float4 o = ld4(p);
        r[0] += o.x; r[1] += o.y; r[2] += o.z; r[3] += o.w;
      }
      st4(p, make_float4(r[0], r[1], r[2], r[3]));
    }
}

// line-coalesced backward: same exchange for t and for the incoming gradient, ReLU gate applied on
// the owner side before the exchange back, read-modify-write of gt with the coalesced map
template <int R, int SOLVER, int WPB, bool HALF, typename AT, bool SYNC>
__device__ __forceinline__ void cf_bwd_tile_body(const AT* __restrict__ t, const float* __restrict__ u0,
                                                 const float* __restrict__ v0, const AT* __restrict__ ga,
                                                 AT* __restrict__ gt, const CfGeom& q, const CfTileId& id, int T, int G, float eps,
                                                 int relu_gate, float* S, int role, const CfSync& y, int ia, int ib) {
  using TL = CfTile<WPB>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t base, V;
  unsigned off[2], off2[2];
  int lidx[2];
  cf_tile_decode<WPB>(q, id, tid, base, V, off, lidx, off2);
  const int own0 = cf_owner_lidx<WPB>(lane, wave, 0), own1 = cf_owner_lidx<WPB>(lane, wave, 1);
  Hist<8, 8, R> h;
  h.carve(S + TL::STAGE_FLOATS + wave * Hist<8, 8, R>::floats(G), G);

  float x[8][8], g[8][8];
#pragma unroll
  for (int dd = 0; dd < 8; ++dd)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 v = cf_ld4<HALF>(t + base + dd * V, off[k], off2[k]);
      x[dd][k * 4 + 0] = v.x; x[dd][k * 4 + 1] = v.y; x[dd][k * 4 + 2] = v.z; x[dd][k * 4 + 3] = v.w;
    }
#pragma unroll
  for (int dd = 0; dd < 8; ++dd)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 v = cf_ld4<HALF>(ga + base + dd * V, off[k], off2[k]);
      g[dd][k * 4 + 0] = v.x; g[dd][k * 4 + 1] = v.y; g[dd][k * 4 + 2] = v.z; g[dd][k * 4 + 3] = v.w;
    }
  cf_to_owner<WPB>(S, lidx, own0, own1, x);
  cf_to_owner<WPB>(S, lidx, own0, own1, g);
  cf_divide(g, q.gscale_div);
  CfWave w{lane};
  nmf_backward_wave<8, 8, R, SOLVER>(w, u0, v0, x, g, h, 8, T, G, eps, nullptr, nullptr);

  // gate first (frees x), then ALL loads of the running sum at once: one exposed round trip, not four
  if (relu_gate) {
#pragma unroll
    for (int dd = 0; dd < 8; ++dd)
#pragma unroll
      for (int e = 0; e < 8; ++e) g[dd][e] = x[dd][e] > 0.f ? g[dd][e] : 0.f;
  }
  // (the plane base back in a scalar register pair: after the wave program the compiler otherwise carries it in vector
  // registers and every epilogue access pays a 64-bit vector address)
  asm volatile("" : "+s"(base));
  float4 old[8][2];
  if (q.accumulate) {
    if (SYNC && role == CF_CONSUME) {
      cf_await(y, ia, ib, tid);
#pragma unroll
      for (int dd = 0; dd < 8; ++dd)
#pragma unroll
        for (int k = 0; k < 2; ++k) old[dd][k] = cf_ld4_sc1(y, gt, base + dd * V + off[k]);
    } else {
#pragma unroll
      for (int dd = 0; dd < 8; ++dd)
#pragma unroll
        for (int k = 0; k < 2; ++k) old[dd][k] = cf_ld4<HALF>(gt + base + dd * V, off[k], off2[k]);
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      *reinterpret_cast<float4*>(S + c * 64 * TL::LW + own0) =
          make_float4(g[2 * s + c][0], g[2 * s + c][1], g[2 * s + c][2], g[2 * s + c][3]);
      *reinterpret_cast<float4*>(S + c * 64 * TL::LW + own1) =
          make_float4(g[2 * s + c][4], g[2 * s + c][5], g[2 * s + c][6], g[2 * s + c][7]);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float4 z = *reinterpret_cast<const float4*>(S + c * 64 * TL::LW + lidx[k]);
        if (q.accumulate) {
          const float4 o = old[2 * s + c][k];
          z.x += o.x; z.y += o.y; z.z += o.z; z.w += o.w;
        }
        if (SYNC && role == CF_PRODUCE) cf_st4_sc1(y, gt, base + (2 * s + c) * V + off[k], z);
        else cf_st4<HALF>(gt + base + (2 * s + c) * V, off[k], off2[k], z);
      }
    __syncthreads();
  }
  if (SYNC && role == CF_PRODUCE) cf_publish(y, ia, tid);
}

template <int R, int SOLVER, int WPB, bool HALF, typename AT>
__global__ __launch_bounds__(WPB * 64, (R == 1 && SOLVER == 1 && (WPB == 4 || WPB == 1)) ? 2 : 1) void nmf_cf_bwd_tile_kernel(
    const AT* __restrict__ t, const float* __restrict__ u0, const float* __restrict__ v0,
    const AT* __restrict__ ga, AT* __restrict__ gt, CfGeom q, int T, int G, float eps, int relu_gate,
    int xcd_remap) {
  extern __shared__ __attribute__((aligned(16))) float fz_lds_cf[];
  CfSync none{};
  cf_bwd_tile_body<R, SOLVER, WPB, HALF, AT, false>(t, u0, v0, ga, gt, q, cf_tile_id<WPB>(q, cf_logical_block(xcd_remap)), T, G, eps,
                                                    relu_gate, fz_lds_cf, CF_PLAIN, none, 0, 0);
}


// ---- BOTH shift windows in ONE launch, scheduled for the Infinity Cache ---------------------------------------------
// The reference treats the windows as independent passes over the whole tensor (operations.py:417-434) and so did rounds
// 1-3: window 1 re-read t and read-modify-wrote the running average from HBM (forward 2·U + 3·U, backward 3·U + 4·U).
// A (sample, head) slice never interacts with another slice and window 1's patch-plane q only needs window 0's planes
// q - 1 and q (a D-axis shift of 1 .. 7 voxels), so the launch walks the tensor SLAB-MAJOR: for a group of slices, window 0
// of plane s, then window 1 of plane s - 1 - lag, ... — a few tens of MB between the two uses of every byte of t and of the
// running average, which the 256 MiB Infinity Cache holds (profiles/r04_memory_ceilings.json: the bare access pattern moves
// its 5·U at 8.1 TB/s this way against 5.4 TB/s layer-major).
//
// Work items are handed out by a TICKET counter in the order above (persistent workgroups; the next ticket is requested
// before the current tile is processed), so every tile a window-1 item waits for has a LOWER ticket, i.e. is held by a
// workgroup that is running or done: no assumption on dispatch order, timing or workgroup -> XCD placement, no deadlock.
// The hand-off itself is CfSync above.  Ticket -> item: bundles of `group x tiles-per-plane` items; inside a bundle
// consecutive tickets cycle over 8 lanes, each a contiguous run of tiles (neighbours along W, then H) of one slice, so the
// workgroups of one XCD (round-robin placement: speed only) keep sharing lines in their L2.
struct Cf2Plan {
  int GQ;                  // tiles per patch row (G2 / WPB)
  int tpp;                 // tiles per (slice, plane) = G1 * GQ
  int group;               // slices per group
  int items_w;             // group * tpp
  int lag;                 // extra planes window 1 trails window 0 by
  unsigned total;          // 2 * nslice * G0 * tpp
  int s1d, s1h, s1w;       // window 1's shift (window 0 is unshifted)
  int divisor;             // forward: number of windows (2)
  int nowait;              // timing probes only (FZ_CF2_NOWAIT): window 1 does not wait — results invalid
};

struct Cf2Item { int win, slice; CfTileId id; };

__device__ __forceinline__ Cf2Item cf2_decode(const CfGeom& q, const Cf2Plan& p, unsigned k) {
  const int G0 = q.G0;
  const unsigned per_group = 2u * (unsigned)G0 * (unsigned)p.items_w;
  const unsigned g = k / per_group, rem = k % per_group;
  const int bi = (int)(rem / (unsigned)p.items_w), i = (int)(rem % (unsigned)p.items_w);
  Cf2Item it;
  // bundle order of a group: w0(0..lag), then pairs w0(s), w1(s - 1 - lag) for s = lag + 1 .. G0 - 1, then the rest of w1
  int plane, j = 0;   // j: window-1 order index, plane (j + 1) % G0 — plane 0 needs window 0's LAST plane, so it comes last
  if (bi <= p.lag) { it.win = 0; plane = bi; }
  else {
    const int r = bi - (p.lag + 1), npair = 2 * (G0 - 1 - p.lag);
    if (r < npair) { const int s = p.lag + 1 + (r >> 1); it.win = r & 1; plane = s; j = s - 1 - p.lag; }
    else { it.win = 1; plane = 0; j = (G0 + (r - npair)) - 1 - p.lag; }
  }
  if (it.win) { plane = j + 1; if (plane == G0) plane = 0; }
  int flat = i;
  if ((p.items_w & 7) == 0) flat = (i & 7) * (p.items_w >> 3) + (i >> 3);
  const int sl = flat / p.tpp, tile = flat % p.tpp;
  it.slice = (int)g * p.group + sl;
  it.id.b = it.slice / q.h; it.id.hh = it.slice % q.h;
  it.id.g0 = plane; it.id.g1 = tile / p.GQ; it.id.gq = tile % p.GQ;
  return it;
}

__device__ __forceinline__ void cf2_roles(const Cf2Plan& p, const Cf2Item& it, CfGeom& q, int& role, int& ia, int& ib) {
  const int plane = it.id.g0;
  if (it.win == 0) {
    q.s0 = q.s1 = q.s2 = 0; q.accumulate = 0; q.divisor = 1;
    role = CF_PRODUCE;
    ia = ib = it.slice * q.G0 + plane;
  } else {
    q.s0 = p.s1d; q.s1 = p.s1h; q.s2 = p.s1w; q.accumulate = 1; q.divisor = p.divisor;
    role = CF_CONSUME;
    // voxels 8·plane - s0 .. 8·plane - s0 + 7 (0 < s0 < 8): window 0's planes plane - 1 and plane
    ia = it.slice * q.G0 + (plane == 0 ? q.G0 - 1 : plane - 1);
    ib = it.slice * q.G0 + plane;
  }
}

// ws: [0] ticket, [1] timeout code, [2..15] unused, [16 ..] done[slice][plane]   (zeroed by the launch function)
enum { CF2_WS_HEAD = 16 };

template <int R, int SOLVER, typename AT>
__global__ __launch_bounds__(512, 2) void nmf_cf_fwd2_kernel(const AT* __restrict__ t, const float* __restrict__ u0,
                                                            const float* __restrict__ v0, AT* __restrict__ out, CfGeom q,
                                                            Cf2Plan p, int T, float eps, unsigned* __restrict__ ws,
                                                            unsigned out_bytes) {
  constexpr int WPB = 8;
  extern __shared__ __attribute__((aligned(16))) float fz_lds_tile[];
  float* S = fz_lds_tile;
  unsigned* slot = reinterpret_cast<unsigned*>(S + CfTile<WPB>::STAGE_FLOATS);
  const int tid = threadIdx.x;
  CfSync y;
  y.rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_bytes, 0x00020000);
  y.done = ws + CF2_WS_HEAD; y.timeout = ws + 1; y.need = p.nowait ? 0 : p.tpp;
  if (tid == 0) *slot = __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  unsigned k = __builtin_amdgcn_readfirstlane(*slot);
  while (k < p.total) {
    unsigned nxt = 0;
    if (tid == 0) nxt = __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // used after the tile
    const Cf2Item it = cf2_decode(q, p, k);
    int role, ia, ib;
    cf2_roles(p, it, q, role, ia, ib);
    cf_fwd_tile_body<R, SOLVER, WPB, false, AT, true>(t, u0, v0, out, q, it.id, T, eps, S, role, y, ia, ib);
    if (tid == 0) *slot = nxt;
    __syncthreads();
    k = __builtin_amdgcn_readfirstlane(*slot);
  }
}

template <int R, int SOLVER, typename AT>
__global__ __launch_bounds__(256, 2) void nmf_cf_bwd2_kernel(const AT* __restrict__ t, const float* __restrict__ u0,
                                                            const float* __restrict__ v0, const AT* __restrict__ ga,
                                                            AT* __restrict__ gt, CfGeom q, Cf2Plan p, int T, int G, float eps,
                                                            int relu_gate, unsigned* __restrict__ ws, unsigned gt_bytes) {
  constexpr int WPB = 4;
  extern __shared__ __attribute__((aligned(16))) float fz_lds_cf[];
  float* S = fz_lds_cf;
  unsigned* slot = reinterpret_cast<unsigned*>(S + CfTile<WPB>::STAGE_FLOATS + WPB * Hist<8, 8, R>::floats(G));
  const int tid = threadIdx.x;
  CfSync y;
  y.rsrc = __builtin_amdgcn_make_buffer_rsrc(gt, 0, (int)gt_bytes, 0x00020000);
  y.done = ws + CF2_WS_HEAD; y.timeout = ws + 1; y.need = p.nowait ? 0 : p.tpp;
  if (tid == 0) *slot = __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  unsigned k = __builtin_amdgcn_readfirstlane(*slot);
  while (k < p.total) {
    unsigned nxt = 0;
    if (tid == 0) nxt = __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const Cf2Item it = cf2_decode(q, p, k);
    int role, ia, ib;
    cf2_roles(p, it, q, role, ia, ib);
    cf_bwd_tile_body<R, SOLVER, WPB, false, AT, true>(t, u0, v0, ga, gt, q, it.id, T, G, eps, relu_gate, S, role, y, ia, ib);
    if (tid == 0) *slot = nxt;
    __syncthreads();
    k = __builtin_amdgcn_readfirstlane(*slot);
  }
}

static int cf_geom(CfGeom& q, int B, int C, int D, int H, int W, const int* shift, int accumulate, int divisor) {
  if (B < 0 || C < 8 || (C % 8) || D < 8 || H < 8 || W < 8 || (D % 8) || (H % 8) || (W % 8))
    return fail(FZ_E_SHAPE, "fz_nmf_cf: needs C % 8 == 0 and spatial dims multiples of 8");
  if (!shift) return fail(FZ_E_ARG, "fz_nmf_cf: shift is null");
  if ((int64_t)D * H * W >= ((int64_t)1 << 30)) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf: 2^30 or more voxels per channel plane");   // 32-bit lane offsets
  q.B = B; q.C = C; q.D = D; q.H = H; q.W = W; q.h = C / 8; q.G0 = D / 8; q.G1 = H / 8; q.G2 = W / 8;
  int s[3];
  const int S[3] = {D, H, W};
  for (int i = 0; i < 3; ++i) { s[i] = shift[i] % S[i]; if (s[i] < 0) s[i] += S[i]; }
  if (s[2] % 2) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf: W-axis shift must be even");
  q.s0 = s[0]; q.s1 = s[1]; q.s2 = s[2];
  q.accumulate = accumulate; q.divisor = divisor; q.gscale_div = 1.0f;
  q.plane = (int64_t)D * H * W;
#ifdef FZ_PROBE_PLANE_PAD   // timing probe (tools/probes/gram_floor.sh): channel planes FZ_PROBE_PLANE_PAD elements further apart
  q.plane += FZ_PROBE_PLANE_PAD;
#endif
  return FZ_OK;
}

}  // namespace fz

using namespace fz;

extern "C" int fz_nmf_cf_supported(int C, int D, int H, int W, int d, int pd, int ph, int pw, int R, int T, int Tgrad) {
  if (d != 8 || pd != 8 || ph != 8 || pw != 8 || (C % 8) || (D % 8) || (H % 8) || (W % 8)) return 0;
  if (R < 1 || R > 2 || T < 0 || (int64_t)D * H * W >= ((int64_t)1 << 30)) return 0;
  const int G = Tgrad < 0 ? 0 : (Tgrad > T ? T : Tgrad);
  const int per_wave = ((G + 1) * R * 8 * 64 + (G + 1) * 8 * R + G * (8 * R + R * R)) * 4;
  return per_wave <= 160 * 1024 ? 1 : 0;
}

template <typename AT>
static int cf_fwd_launch(const AT* t, const float* u0, const float* v0, AT* out, int B, int C, int D,
                         int H, int W, const int* shift, int accumulate, int divisor, int R, int T,
                         int solver, float eps, fz_stream_t stream) {
  CfGeom q;
  int rc = cf_geom(q, B, C, D, H, W, shift, accumulate, divisor);
  if (rc != FZ_OK) return rc;
  if (!t || !u0 || !v0 || !out) return fail(FZ_E_ARG, "fz_nmf_cf_fwd: null pointer");
  if (R < 1 || R > 2) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_fwd: rank 1..2");
  if (solver != FZ_SOLVER_MU && solver != FZ_SOLVER_HALS) return fail(FZ_E_ARG, "fz_nmf_cf_fwd: bad solver");
  if (B == 0) return FZ_OK;
  const int64_t nmat = (int64_t)B * q.h * q.G0 * q.G1 * q.G2;
  if (nmat >= (int64_t)1 << 31) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf: more than 2^31 matrices");
  // measured on MI355X (round-1/2 probe `cf_probe`): a workgroup = one full row of patches along W
  // (up to 16 waves) consumes whole 128-B lines inside one CU: 0.856 -> 0.725 ms at stage 0
  int wpb = 16;
  { const auto& k = FZ_ENV_KNOB("FZ_CF_WPB"); if (k.set) wpb = k.val; }
  if (wpb > q.G2) wpb = q.G2;
  if (wpb < 1) wpb = 1;
  int xr = 1;
  { const auto& k = FZ_ENV_KNOB("FZ_CF_XCD"); if (k.set) xr = k.val; }
  hipStream_t st = (hipStream_t)stream;
  int tile = 1;
  { const auto& k = FZ_ENV_KNOB("FZ_CF_TILE"); if (k.set) tile = k.val; }
  const bool half = (q.s2 % 4) != 0;  // W-axis shift ≡ 2 (mod 4): only the line-coalesced kernels handle it
  if (half || (tile && (q.G2 % 8) == 0)) {
    // line-coalesced kernel: WPB patches along W per workgroup.  8 patches per workgroup, two
    // workgroups per CU out of phase: 0.524 ms vs 0.554 (16) vs 0.747 (direct gather) at stage 0
    const int twpb = (q.G2 % 8) == 0 ? 8 : ((q.G2 % 4) == 0 ? 4 : 1);
    const unsigned nblk = (unsigned)(nmat / twpb);
#define FZ_CF_TILE(RR, SS, WW, HH)                                                                          \
  do {                                                                                                      \
    auto kern = nmf_cf_fwd_tile_kernel<RR, SS, WW, HH, AT>;                                                   \
    const int lds = CfTile<WW>::STAGE_FLOATS * (int)sizeof(float);                                          \
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * WW), lds, st, t, u0, v0, out, q, T, eps, xr);            \
  } while (0)
#define FZ_CF_TILE_W(RR, SS)                                                                                \
  do {                                                                                                      \
    if (half) { if (twpb == 8) FZ_CF_TILE(RR, SS, 8, true); else if (twpb == 4) FZ_CF_TILE(RR, SS, 4, true); else FZ_CF_TILE(RR, SS, 1, true); } \
    else FZ_CF_TILE(RR, SS, 8, false);                                                                      \
  } while (0)
    if (R == 1) { if (solver == FZ_SOLVER_MU) FZ_CF_TILE_W(1, SOLVER_MU); else FZ_CF_TILE_W(1, SOLVER_HALS); }
    else { if (solver == FZ_SOLVER_MU) FZ_CF_TILE_W(2, SOLVER_MU); else FZ_CF_TILE_W(2, SOLVER_HALS); }
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }
  dim3 grid((unsigned)((nmat + wpb - 1) / wpb)), block(64 * wpb);
#define FZ_CF_FWD(RR, SS) hipLaunchKernelGGL((nmf_cf_fwd_kernel<RR, SS, AT>), grid, block, 0, st, t, u0, v0, out, q, nmat, T, eps, xr)
  if (R == 1) { if (solver == FZ_SOLVER_MU) FZ_CF_FWD(1, SOLVER_MU); else FZ_CF_FWD(1, SOLVER_HALS); }
  else { if (solver == FZ_SOLVER_MU) FZ_CF_FWD(2, SOLVER_MU); else FZ_CF_FWD(2, SOLVER_HALS); }
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_nmf_cf_fwd(const void* t, const float* u0, const float* v0, void* out, int B, int C, int D,
                             int H, int W, const int* shift, int accumulate, int divisor, int R, int T,
                             int solver, float eps, int act_dtype, fz_stream_t stream) {
  if (act_dtype == FZ_STORE_F32)
    return cf_fwd_launch<float>((const float*)t, u0, v0, (float*)out, B, C, D, H, W, shift, accumulate, divisor, R, T,
                                solver, eps, stream);
  if (act_dtype == FZ_STORE_BF16)
    return cf_fwd_launch<bf16>((const bf16*)t, u0, v0, (bf16*)out, B, C, D, H, W, shift, accumulate, divisor, R, T,
                               solver, eps, stream);
  return fail(FZ_E_ARG, "fz_nmf_cf_fwd: bad act_dtype");
}

template <typename AT>
static int cf_bwd_launch(const AT* t, const float* u0, const float* v0, const AT* ga, AT* gt, int B,
                         int C, int D, int H, int W, const int* shift, int accumulate, int nshift,
                         int relu_gate, int R, int T, int Tgrad, int solver, float eps, fz_stream_t stream) {
  CfGeom q;
  int rc = cf_geom(q, B, C, D, H, W, shift, accumulate, 1);
  if (rc != FZ_OK) return rc;
  if (!t || !u0 || !v0 || !ga || !gt) return fail(FZ_E_ARG, "fz_nmf_cf_bwd: null pointer");
  if (R < 1 || R > 2) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_bwd: rank 1..2");
  if (solver != FZ_SOLVER_MU && solver != FZ_SOLVER_HALS) return fail(FZ_E_ARG, "fz_nmf_cf_bwd: bad solver");
  if (B == 0) return FZ_OK;
  q.gscale_div = (float)(nshift > 1 ? nshift : 1);
  const int G = Tgrad < 0 ? 0 : (Tgrad > T ? T : Tgrad);
  const int64_t nmat = (int64_t)B * q.h * q.G0 * q.G1 * q.G2;
  if (nmat >= (int64_t)1 << 31) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf: more than 2^31 matrices");
  const int per_wave = (R == 1 ? Hist<8, 8, 1>::floats(G) : Hist<8, 8, 2>::floats(G)) * (int)sizeof(float);
  if (per_wave > 160 * 1024) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_bwd: history exceeds LDS");
  int wpb = 65536 / per_wave;
  if (wpb > 4) wpb = 4;
  { const auto& k = FZ_ENV_KNOB("FZ_CF_WPB_BWD"); if (k.set) { wpb = k.val; if (wpb * per_wave > 160 * 1024) wpb = 160 * 1024 / per_wave; } }
  if (wpb > 8) wpb = 8;
  if (wpb < 1) wpb = 1;
  // patch neighbours on the same XCD share its L2 (shifted windows straddle lines): 1.31 -> 1.17 ms
  int xr = 1;
  { const auto& k = FZ_ENV_KNOB("FZ_CF_XCD"); if (k.set) xr = k.val; }
  hipStream_t st = (hipStream_t)stream;
  int tile = 1;
  { const auto& k = FZ_ENV_KNOB("FZ_CF_TILE_BWD"); if (k.set) tile = k.val; }
  const bool half = (q.s2 % 4) != 0;
  if (half || (tile && (q.G2 % 4) == 0)) {
    const int twpb = (q.G2 % 4) == 0 ? 4 : 1;
    // HALS rank 1 behind a ReLU (t >= 0 by the relu_gate contract): the row-space reverse mode, no per-column history
    static const bool gram_on = !(FZ_ENV_KNOB("FZ_CF_GRAM").set && FZ_ENV_KNOB("FZ_CF_GRAM").val == 0);
    if (gram_on && R == 1 && solver == FZ_SOLVER_HALS && relu_gate && G >= 1) {
      rc = cf_bwd_gram_launch<AT>(t, v0, ga, gt, q, nmat, T, G, eps, xr, st);
      if (rc != FZ_E_UNSUPPORTED) return rc;
    }
    const int tlds = (twpb == 4 ? CfTile<4>::STAGE_FLOATS : CfTile<1>::STAGE_FLOATS) * (int)sizeof(float) + per_wave * twpb;
    if (tlds > 160 * 1024) {
      if (half) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_bwd: history exceeds LDS for a W-axis shift of 2 (mod 4)");
    } else {
      const unsigned nblk = (unsigned)(nmat / twpb);
#define FZ_CF_BWD_TILE(RR, SS, WW, HH)                                                                      \
  do {                                                                                                      \
    auto kern = nmf_cf_bwd_tile_kernel<RR, SS, WW, HH, AT>;                                                    \
    if (tlds > 65536)                                                                                       \
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, tlds));                     \
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * WW), tlds, st, t, u0, v0, ga, gt, q, T, G, eps, relu_gate, xr); \
  } while (0)
#define FZ_CF_BWD_TILE_W(RR, SS)                                                                            \
  do {                                                                                                      \
    if (half) { if (twpb == 4) FZ_CF_BWD_TILE(RR, SS, 4, true); else FZ_CF_BWD_TILE(RR, SS, 1, true); }     \
    else FZ_CF_BWD_TILE(RR, SS, 4, false);                                                                  \
  } while (0)
      if (R == 1) { if (solver == FZ_SOLVER_MU) FZ_CF_BWD_TILE_W(1, SOLVER_MU); else FZ_CF_BWD_TILE_W(1, SOLVER_HALS); }
      else { if (solver == FZ_SOLVER_MU) FZ_CF_BWD_TILE_W(2, SOLVER_MU); else FZ_CF_BWD_TILE_W(2, SOLVER_HALS); }
      FZ_LAUNCH_CHECK();
      return FZ_OK;
    }
  }
  const int lds = per_wave * wpb;
  dim3 grid((unsigned)((nmat + wpb - 1) / wpb)), block(64 * wpb);
#define FZ_CF_BWD(RR, SS)                                                                                 \
  do {                                                                                                    \
    auto kern = nmf_cf_bwd_kernel<RR, SS, AT>;                                                          \
    if (lds > 65536)                                                                                      \
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));                    \
    hipLaunchKernelGGL(kern, grid, block, lds, st, t, u0, v0, ga, gt, q, nmat, T, G, eps, relu_gate, xr);     \
  } while (0)
  if (R == 1) { if (solver == FZ_SOLVER_MU) FZ_CF_BWD(1, SOLVER_MU); else FZ_CF_BWD(1, SOLVER_HALS); }
  else { if (solver == FZ_SOLVER_MU) FZ_CF_BWD(2, SOLVER_MU); else FZ_CF_BWD(2, SOLVER_HALS); }
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_nmf_cf_bwd(const void* t, const float* u0, const float* v0, const void* ga, void* gt, int B,
                             int C, int D, int H, int W, const int* shift, int accumulate, int nshift,
                             int relu_gate, int R, int T, int Tgrad, int solver, float eps, int act_dtype,
                             fz_stream_t stream) {
  if (act_dtype == FZ_STORE_F32)
    return cf_bwd_launch<float>((const float*)t, u0, v0, (const float*)ga, (float*)gt, B, C, D, H, W, shift, accumulate,
                                nshift, relu_gate, R, T, Tgrad, solver, eps, stream);
  if (act_dtype == FZ_STORE_BF16)
    return cf_bwd_launch<bf16>((const bf16*)t, u0, v0, (const bf16*)ga, (bf16*)gt, B, C, D, H, W, shift, accumulate,
                               nshift, relu_gate, R, T, Tgrad, solver, eps, stream);
  return fail(FZ_E_ARG, "fz_nmf_cf_bwd: bad act_dtype");
}

// ---- two windows in one launch: host side ------------------------------------------------------------------------------
namespace fz {
static int knob_cf2_group() { return FZ_ENV_KNOB("FZ_CF2_GROUP").val; }    // slices per group (0 = auto)
static int knob_cf2_lag() { return FZ_ENV_KNOB("FZ_CF2_LAG").val; }
static int knob_cf2_nowait() { return FZ_ENV_KNOB("FZ_CF2_NOWAIT").val; }  // TIMING PROBES ONLY: results invalid
static int knob_cf2_wgs() { return FZ_ENV_KNOB("FZ_CF2_WGS").val; }        // persistent workgroups (0 = resident)

// WPB: patches per tile (8 forward, 4 backward); passes: tensor passes one (plane, both windows) step moves (5 forward, 7 backward)
static int cf2_plan(Cf2Plan& p, int B, int C, int D, int H, int W, const int* shifts, int WPB, int es, int passes,
                    const int* tune = nullptr) {
  if (C % 8 || D % 8 || H % 8 || W % 8) return 0;
  const int G0 = D / 8, G1 = H / 8, G2 = W / 8;
  if (G2 % WPB) return 0;
  // window 0 unshifted (its stores are whole 128-B lines: the measured hand-off form), window 1 a D-axis shift of 1 .. 7
  // voxels (so that it needs exactly window 0's planes q - 1 and q) and a W-axis shift that keeps 16-byte chunks whole
  if (shifts[0] || shifts[1] || shifts[2]) return 0;
  int s[3];
  const int Sz[3] = {D, H, W};
  for (int i = 0; i < 3; ++i) { s[i] = shifts[3 + i] % Sz[i]; if (s[i] < 0) s[i] += Sz[i]; }
  if (s[0] < 1 || s[0] > 7 || (s[2] % 4) || G0 < 2) return 0;
  const int64_t nslice = (int64_t)B * (C / 8);
  const int64_t tensor_bytes = nslice * 8 * (int64_t)D * H * W * es;
  if (tensor_bytes >= ((int64_t)1 << 31)) return 0;   // 32-bit buffer offsets
  p.GQ = G2 / WPB; p.tpp = G1 * p.GQ;
  // slices per group.  The design intent was a group small enough for the Infinity Cache (two steps of `passes` plane-sets
  // of the group within ~96 MB); measured (profiles/r04_cf2_sweep.json) small groups only make window-1 workgroups wait on a
  // resident slot — the 4 096 matrices in flight are 164 MB by themselves — so the library's own choice is the whole tensor
  // as ONE group with window 1 a plane further behind (the least slow setting: within 6 % of the one-window launches).
  (void)passes;
  int group = (int)nslice;
  if (tune && tune[0] > 0) group = tune[0];
  else if (knob_cf2_group() > 0) group = knob_cf2_group();
  if (group < 1 || nslice % group) return 0;
  p.group = group; p.items_w = group * p.tpp;
  int lag = (tune && tune[1] >= 0) ? tune[1] : (FZ_ENV_KNOB("FZ_CF2_LAG").set ? knob_cf2_lag() : 1);
  if (lag < 0) lag = 0;
  if (lag > G0 - 1) lag = G0 - 1;
  p.lag = lag;
  const int64_t total = 2 * nslice * G0 * p.tpp;
  if (total >= ((int64_t)1 << 31)) return 0;
  p.total = (unsigned)total;
  p.s1d = s[0]; p.s1h = s[1]; p.s1w = s[2];
  p.divisor = 2;
  p.nowait = knob_cf2_nowait();
  return 1;
}
}  // namespace fz

extern "C" int64_t fz_nmf_cf2_workspace_bytes(int B, int C, int D) {
  if (B < 0 || C < 8 || D < 8) return 0;
  const int64_t n = CF2_WS_HEAD + (int64_t)B * (C / 8) * (D / 8);
  return ((n * 4 + 15) / 16) * 16;
}

extern "C" int fz_nmf_cf2_supported(int B, int C, int D, int H, int W, const int* shifts, int nshift, int R, int T, int Tgrad,
                                    int act_dtype) {
  // rank 1 only: the rank-2 wave programs sit at the 256-register limit and would spill beside the ticket loop
  if (nshift != 2 || !shifts || B < 1 || R != 1) return 0;
  if (!fz_nmf_cf_supported(C, D, H, W, 8, 8, 8, 8, R, T, Tgrad)) return 0;
  const int es = act_dtype == FZ_STORE_BF16 ? 2 : 4;
  Cf2Plan p;
  if (!cf2_plan(p, B, C, D, H, W, shifts, 8, es, 5)) return 0;
  if (!cf2_plan(p, B, C, D, H, W, shifts, 4, es, 7)) return 0;
  const int G = Tgrad < 0 ? 0 : (Tgrad > T ? T : Tgrad);
  const int per_wave = Hist<8, 8, 1>::floats(G) * (int)sizeof(float);
  return CfTile<4>::STAGE_FLOATS * (int)sizeof(float) + 4 * per_wave + 16 <= 160 * 1024;
}

template <typename AT>
static int cf_fwd2_launch(const AT* t, const float* u0, const float* v0, AT* out, int B, int C, int D, int H, int W,
                          const int* shifts, int R, int T, int solver, float eps, void* workspace, const int* tune,
                          fz_stream_t stream) {
  if (!t || !u0 || !v0 || !out || !shifts || !workspace) return fail(FZ_E_ARG, "fz_nmf_cf_fwd2: null pointer");
  if (R != 1) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_fwd2: rank 1 (see fz_nmf_cf2_supported)");
  if (solver != FZ_SOLVER_MU && solver != FZ_SOLVER_HALS) return fail(FZ_E_ARG, "fz_nmf_cf_fwd2: bad solver");
  CfGeom q;
  const int zero[3] = {0, 0, 0};
  int rc = cf_geom(q, B, C, D, H, W, zero, 0, 1);
  if (rc != FZ_OK) return rc;
  Cf2Plan p;
  if (B < 1 || !cf2_plan(p, B, C, D, H, W, shifts, 8, (int)sizeof(AT), 5, tune))
    return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_fwd2: geometry outside the two-window launch (see fz_nmf_cf2_supported)");
  hipStream_t st = (hipStream_t)stream;
  FZ_HIP_OK(hipMemsetAsync(workspace, 0, (size_t)fz_nmf_cf2_workspace_bytes(B, C, D), st));
  const unsigned bytes = (unsigned)((int64_t)B * C * D * H * W * (int64_t)sizeof(AT));
  const int lds = CfTile<8>::STAGE_FLOATS * (int)sizeof(float) + 16;
#define FZ_CF_FWD2(RR, SS)                                                                                                  \
  do {                                                                                                                      \
    auto kern = nmf_cf_fwd2_kernel<RR, SS, AT>;                                                                             \
    static int resident = 0;                                                                                                \
    if (!resident) {                                                                                                        \
      int per_cu = 0, dev = 0;                                                                                              \
      hipDeviceProp_t prop;                                                                                                 \
      FZ_HIP_OK(hipGetDevice(&dev));                                                                                        \
      FZ_HIP_OK(hipGetDeviceProperties(&prop, dev));                                                                        \
      FZ_HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 512, lds));      \
      resident = (per_cu > 0 ? per_cu : 1) * prop.multiProcessorCount;                                                      \
    }                                                                                                                       \
    unsigned grid = (tune && tune[2] > 0) ? (unsigned)tune[2] : knob_cf2_wgs() > 0 ? (unsigned)knob_cf2_wgs() : (unsigned)resident;                                     \
    if (grid > p.total) grid = p.total;                                                                                     \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, t, u0, v0, out, q, p, T, eps, (unsigned*)workspace, bytes);    \
  } while (0)
  if (solver == FZ_SOLVER_MU) FZ_CF_FWD2(1, SOLVER_MU); else FZ_CF_FWD2(1, SOLVER_HALS);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_nmf_cf_fwd2(const void* t, const float* u0, const float* v0, void* out, int B, int C, int D, int H, int W,
                              const int* shifts, int R, int T, int solver, float eps, int act_dtype, void* workspace,
                              const int* tune, fz_stream_t stream) {
  if (act_dtype == FZ_STORE_F32)
    return cf_fwd2_launch<float>((const float*)t, u0, v0, (float*)out, B, C, D, H, W, shifts, R, T, solver, eps, workspace, tune, stream);
  if (act_dtype == FZ_STORE_BF16)
    return cf_fwd2_launch<bf16>((const bf16*)t, u0, v0, (bf16*)out, B, C, D, H, W, shifts, R, T, solver, eps, workspace, tune, stream);
  return fail(FZ_E_ARG, "fz_nmf_cf_fwd2: bad act_dtype");
}

template <typename AT>
static int cf_bwd2_launch(const AT* t, const float* u0, const float* v0, const AT* ga, AT* gt, int B, int C, int D, int H,
                          int W, const int* shifts, int relu_gate, int R, int T, int Tgrad, int solver, float eps,
                          void* workspace, const int* tune, fz_stream_t stream) {
  if (!t || !u0 || !v0 || !ga || !gt || !shifts || !workspace) return fail(FZ_E_ARG, "fz_nmf_cf_bwd2: null pointer");
  if (R != 1) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_bwd2: rank 1 (see fz_nmf_cf2_supported)");
  if (solver != FZ_SOLVER_MU && solver != FZ_SOLVER_HALS) return fail(FZ_E_ARG, "fz_nmf_cf_bwd2: bad solver");
  CfGeom q;
  const int zero[3] = {0, 0, 0};
  int rc = cf_geom(q, B, C, D, H, W, zero, 0, 1);
  if (rc != FZ_OK) return rc;
  q.gscale_div = 2.0f;
  Cf2Plan p;
  if (B < 1 || !cf2_plan(p, B, C, D, H, W, shifts, 4, (int)sizeof(AT), 7, tune))
    return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_bwd2: geometry outside the two-window launch (see fz_nmf_cf2_supported)");
  p.divisor = 1;
  const int G = Tgrad < 0 ? 0 : (Tgrad > T ? T : Tgrad);
  const int per_wave = Hist<8, 8, 1>::floats(G) * (int)sizeof(float);
  const int lds = CfTile<4>::STAGE_FLOATS * (int)sizeof(float) + 4 * per_wave + 16;
  if (lds > 160 * 1024) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_bwd2: history exceeds LDS");
  hipStream_t st = (hipStream_t)stream;
  FZ_HIP_OK(hipMemsetAsync(workspace, 0, (size_t)fz_nmf_cf2_workspace_bytes(B, C, D), st));
  const unsigned bytes = (unsigned)((int64_t)B * C * D * H * W * (int64_t)sizeof(AT));
#define FZ_CF_BWD2(RR, SS)                                                                                                  \
  do {                                                                                                                      \
    auto kern = nmf_cf_bwd2_kernel<RR, SS, AT>;                                                                             \
    static int resident = 0, resident_lds = -1;                                                                             \
    if (lds > 65536)                                                                                                        \
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    if (!resident || resident_lds != lds) {                                                                                 \
      int per_cu = 0, dev = 0;                                                                                              \
      hipDeviceProp_t prop;                                                                                                 \
      FZ_HIP_OK(hipGetDevice(&dev));                                                                                        \
      FZ_HIP_OK(hipGetDeviceProperties(&prop, dev));                                                                        \
      FZ_HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, lds));      \
      resident = (per_cu > 0 ? per_cu : 1) * prop.multiProcessorCount;                                                      \
      resident_lds = lds;                                                                                                   \
    }                                                                                                                       \
    unsigned grid = (tune && tune[2] > 0) ? (unsigned)tune[2] : knob_cf2_wgs() > 0 ? (unsigned)knob_cf2_wgs() : (unsigned)resident;                                     \
    if (grid > p.total) grid = p.total;                                                                                     \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, t, u0, v0, ga, gt, q, p, T, G, eps, relu_gate,                  \
                       (unsigned*)workspace, bytes);                                                                        \
  } while (0)
  if (solver == FZ_SOLVER_MU) FZ_CF_BWD2(1, SOLVER_MU); else FZ_CF_BWD2(1, SOLVER_HALS);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_nmf_cf_bwd2(const void* t, const float* u0, const float* v0, const void* ga, void* gt, int B, int C, int D,
                              int H, int W, const int* shifts, int relu_gate, int R, int T, int Tgrad, int solver, float eps,
                              int act_dtype, void* workspace, const int* tune, fz_stream_t stream) {
  if (act_dtype == FZ_STORE_F32)
    return cf_bwd2_launch<float>((const float*)t, u0, v0, (const float*)ga, (float*)gt, B, C, D, H, W, shifts, relu_gate, R, T,
                                 Tgrad, solver, eps, workspace, tune, stream);
  if (act_dtype == FZ_STORE_BF16)
    return cf_bwd2_launch<bf16>((const bf16*)t, u0, v0, (const bf16*)ga, (bf16*)gt, B, C, D, H, W, shifts, relu_gate, R, T, Tgrad,
                                solver, eps, workspace, tune, stream);
  return fail(FZ_E_ARG, "fz_nmf_cf_bwd2: bad act_dtype");
}
