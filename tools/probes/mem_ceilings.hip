// mem_ceilings.hip — what the MI355X memory system gives a hand-written streaming kernel (no torch, no library):
// 16 B per lane global_load_dwordx4 / global_store_dwordx4, persistent grid-stride workgroups.
//   (1) read-only, write-only and R:1 read:write mixes on a 1 GiB footprint (HBM) and a 64 MiB footprint
//       (Infinity Cache resident when replayed), for 4 / 8 / 16 waves per CU and 2 / 4 / 8 vectors in flight per lane;
//   (2) the access pattern of the two shift windows of the fused FactMixer core
//         pass A: read t, write a            pass B: read t, read a, write a
//       over 537 MB tensors (stage 0 of the README model, B = 2) run layer-major (A over everything, then B over
//       everything: what rounds 1-3 shipped) against chunk-major (A then B per chunk of 4 .. 134 MB), as separate
//       launches per chunk and as ONE launch whose block order interleaves the two passes.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/mem_ceilings tools/probes/mem_ceilings.hip
// Output: one JSON object on stdout (profiles/r04_memory_ceilings.json).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// NR source streams, NW (0/1) destination stream, U vectors in flight per lane and stream.
// Stream k of a launch lives at base + k * stride (in float4 units); n = float4 per stream.
template <int NR, int NW, int U>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                     size_t n, size_t stride, float4* __restrict__ sink) {
  const size_t step = (size_t)gridDim.x * 256 * U;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i0 = (size_t)blockIdx.x * 256 * U + threadIdx.x; i0 < n; i0 += step) {
    float4 v[NR > 0 ? NR : 1][U];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t i = i0 + (size_t)u * 256;
        v[r][u] = i < n ? src[r * stride + i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    float4 o[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      o[u] = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
      for (int r = 0; r < NR; ++r) { o[u].x += v[r][u].x; o[u].y += v[r][u].y; o[u].z += v[r][u].z; o[u].w += v[r][u].w; }
    }
    if (NW) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t i = i0 + (size_t)u * 256;
        if (i < n) dst[i] = o[u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) { acc.x += o[u].x; acc.y += o[u].y; acc.z += o[u].z; acc.w += o[u].w; }
    }
  }
  if (!NW && acc.x == 123.456f) sink[0] = acc;  // never true: keeps the loads alive
}

// The two window passes over chunk [c0, c0 + cn) of the float4 index space, one launch:
// blocks [0, nbA) run pass A, blocks [nbA, nbA + nbB) pass B — and with `interleave` the launch walks
// chunk by chunk: logical order A(chunk 0), A(chunk 1), B(chunk 0), A(chunk 2), B(chunk 1) ... (B lags A by one chunk).
// No inter-block dependency is enforced: this probe measures bandwidth, not values.
template <int U>
__device__ __forceinline__ void pass_a(const float4* __restrict__ t, float4* __restrict__ a, size_t lo, size_t hi,
                                       int blk, int nblk) {
  const size_t step = (size_t)nblk * 256 * U;
  for (size_t i0 = lo + (size_t)blk * 256 * U + threadIdx.x; i0 < hi; i0 += step) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = i0 + (size_t)u * 256; v[u] = i < hi ? t[i] : make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = i0 + (size_t)u * 256; if (i < hi) a[i] = make_float4(v[u].x + 0.f, v[u].y * 2.f, v[u].z, v[u].w); }
  }
}
template <int U>
__device__ __forceinline__ void pass_b(const float4* __restrict__ t, float4* __restrict__ a, size_t lo, size_t hi,
                                       int blk, int nblk) {
  const size_t step = (size_t)nblk * 256 * U;
  for (size_t i0 = lo + (size_t)blk * 256 * U + threadIdx.x; i0 < hi; i0 += step) {
    float4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = i0 + (size_t)u * 256; v[u] = i < hi ? t[i] : make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = i0 + (size_t)u * 256; w[u] = i < hi ? a[i] : make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + (size_t)u * 256;
      if (i < hi) a[i] = make_float4((v[u].x + w[u].x) * .5f, (v[u].y + w[u].y) * .5f, (v[u].z + w[u].z) * .5f, (v[u].w + w[u].w) * .5f);
    }
  }
}

// the same two passes with the hand-off forms of an in-launch producer/consumer protocol (cdna_hip_programming.md Guideline 16 R1):
// pass A stores `a` write-through (sc1), pass B loads `a` with sc1 (bypassing this CU's L1); t stays on plain loads
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int v4i;
template <int U>
__device__ __forceinline__ void pass_a_sc1(const float4* __restrict__ t, __amdgpu_buffer_rsrc_t ra, size_t lo, size_t hi) {
  const size_t i0 = lo + threadIdx.x;
  float4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const size_t i = i0 + (size_t)u * 256; v[u] = i < hi ? t[i] : make_float4(0, 0, 0, 0); }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t i = i0 + (size_t)u * 256;
    const float4 o = make_float4(v[u].x + 0.f, v[u].y * 2.f, v[u].z, v[u].w);
    if (i < hi) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4i*>(&o), ra, (int)(i * 16), 0, 16);
  }
}
template <int U>
__device__ __forceinline__ void pass_b_sc1(const float4* __restrict__ t, float4* __restrict__ a, __amdgpu_buffer_rsrc_t ra, size_t lo,
                                           size_t hi) {
  const size_t i0 = lo + threadIdx.x;
  float4 v[U];
  v4i w[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const size_t i = i0 + (size_t)u * 256; v[u] = i < hi ? t[i] : make_float4(0, 0, 0, 0); }
#pragma unroll
  for (int u = 0; u < U; ++u) { const size_t i = i0 + (size_t)u * 256; w[u] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(i * 16), 0, 16); }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t i = i0 + (size_t)u * 256;
    const float4 ww = *reinterpret_cast<const float4*>(&w[u]);
    if (i < hi) a[i] = make_float4((v[u].x + ww.x) * .5f, (v[u].y + ww.y) * .5f, (v[u].z + ww.z) * .5f, (v[u].w + ww.w) * .5f);
  }
}

template <int U, int PASS>
__global__ __launch_bounds__(256) void window_pass_kernel(const float4* __restrict__ t, float4* __restrict__ a, size_t lo,
                                                          size_t hi) {
  if (PASS == 0) pass_a<U>(t, a, lo, hi, blockIdx.x, gridDim.x);
  else pass_b<U>(t, a, lo, hi, blockIdx.x, gridDim.x);
}

// ONE launch, work items = (phase, tile): phase p runs A on chunk p (p < nchunk) and B on chunk p - lag (p >= lag);
// a workgroup takes items blockIdx.x, blockIdx.x + gridDim.x, ... in that logical order (persistent, in-order walk).
template <int U, bool SC1 = false>
__global__ __launch_bounds__(256) void window_interleaved_kernel(const float4* __restrict__ t, float4* __restrict__ a, size_t n,
                                                                 size_t chunk, int nchunk, int lag, int tiles_per_chunk) {
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(a, 0, (int)(n * 16), 0x00020000);
  const int nphase = nchunk + lag;
  const long items = (long)nphase * 2 * tiles_per_chunk;
  const size_t tile = (size_t)256 * U;
  for (long it = blockIdx.x; it < items; it += gridDim.x) {
    const int phase = (int)(it / (2 * tiles_per_chunk));
    const int r = (int)(it % (2 * tiles_per_chunk));
    const int which = r / tiles_per_chunk, tl = r % tiles_per_chunk;
    const int c = which == 0 ? phase : phase - lag;
    if (c < 0 || c >= nchunk) continue;
    const size_t lo = (size_t)c * chunk + (size_t)tl * tile;
    size_t hi = lo + tile; if (hi > n) hi = n;
    if (lo >= hi) continue;
    if (SC1) {
      if (which == 0) pass_a_sc1<U>(t, ra, lo, hi);
      else pass_b_sc1<U>(t, a, ra, lo, hi);
    } else {
      if (which == 0) pass_a<U>(t, a, lo, hi, 0, 1);
      else pass_b<U>(t, a, lo, hi, 0, 1);
    }
  }
}

static int g_cus = 256;
static hipEvent_t ev0, ev1;

template <typename F>
static double time_ms(F&& f, int iters, int warm = 3) {
  for (int i = 0; i < warm; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(ev0, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(ev1, 0));
  CK(hipEventSynchronize(ev1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, ev0, ev1));
  return ms / iters;
}

template <int NR, int NW, int U>
static double run_stream(const float4* src, float4* dst, size_t n, size_t stride, float4* sink, int wg_per_cu, int iters) {
  const int grid = g_cus * wg_per_cu;
  return time_ms([&] { hipLaunchKernelGGL((stream_kernel<NR, NW, U>), dim3(grid), dim3(256), 0, 0, src, dst, n, stride, sink); }, iters);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  CK(hipEventCreate(&ev0));
  CK(hipEventCreate(&ev1));
  const size_t GiB = (size_t)1 << 30;
  // five 1 GiB source streams + one destination (6 GiB) + sink
  const size_t n1g = GiB / 16;
  float4 *src, *dst, *sink;
  CK(hipMalloc(&src, 5 * GiB));
  CK(hipMalloc(&dst, GiB));
  CK(hipMalloc(&sink, 256));
  CK(hipMemset(src, 0, 5 * GiB));
  CK(hipMemset(dst, 0, GiB));

  std::string out = "{";
  char buf[512];
  snprintf(buf, sizeof buf, "\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"l2_bytes\": %d,\n", prop.gcnArchName, g_cus,
           prop.clockRate / 1000, prop.l2CacheSize);
  out += buf;

  // ---- (1) streaming mixes -----------------------------------------------------------------------------------
  out += "\"streams\": [\n";
  bool first = true;
  auto emit = [&](const char* mix, size_t bytes_per_stream, int nr, int nw, int wpc, int u, double ms) {
    const double gb = (double)(nr + nw) * bytes_per_stream / 1e9;
    snprintf(buf, sizeof buf,
             "%s{\"mix\": \"%s\", \"footprint_MiB_per_stream\": %zu, \"reads\": %d, \"writes\": %d, \"waves_per_cu\": %d, "
             "\"vec_in_flight\": %d, \"ms\": %.4f, \"GBps\": %.1f}",
             first ? "" : ",\n", mix, bytes_per_stream >> 20, nr, nw, wpc * 4, u, ms, gb / ms * 1e3);
    first = false;
    out += buf;
  };
  const int wpcs[3] = {1, 2, 4};  // workgroups of 4 waves per CU -> 4 / 8 / 16 waves per CU
  for (int fp = 0; fp < 2; ++fp) {
    const size_t bytes = fp == 0 ? GiB : (size_t)64 << 20;
    const size_t n = bytes / 16;
    // small footprint: streams sit back to back so that the whole working set is (NR + NW) x 64 MiB
    const size_t stride = fp == 0 ? n1g : n;
    float4* d = fp == 0 ? dst : src + 5 * n;  // keep the small footprint contiguous
    const int iters = fp == 0 ? 10 : 60;
    for (int w = 0; w < 3; ++w) {
      const int wpc = wpcs[w];
#define RUN(MIX, NR, NW, U) emit(MIX, bytes, NR, NW, wpc, U, run_stream<NR, NW, U>(src, d, n, stride, sink, wpc, iters))
      RUN("read_only", 1, 0, 2); RUN("read_only", 1, 0, 4); RUN("read_only", 1, 0, 8);
      RUN("write_only", 0, 1, 2); RUN("write_only", 0, 1, 4); RUN("write_only", 0, 1, 8);
      RUN("1:1", 1, 1, 2); RUN("1:1", 1, 1, 4); RUN("1:1", 1, 1, 8);
      RUN("2:1", 2, 1, 2); RUN("2:1", 2, 1, 4); RUN("2:1", 2, 1, 8);
      RUN("3:1", 3, 1, 2); RUN("3:1", 3, 1, 4);
      RUN("5:1", 5, 1, 2); RUN("5:1", 5, 1, 4);
#undef RUN
    }
  }
  out += "\n],\n";

  // ---- (2) the two window passes: layer-major vs chunk-major ------------------------------------------------
  // tensors of 512 MiB (stage 0, B = 2, C = 32, 128^3 fp32 = 537 MB); t = src, a = src + 1 GiB
  const size_t tbytes = (size_t)512 << 20, tn = tbytes / 16;
  const float4* t = src;
  float4* a = src + n1g;
  out += "\"window_passes\": {\"tensor_MiB\": 512, \"algorithmic_bytes\": \"5 x tensor (A: read t, write a; B: read t, read a, write a)\",\n";
  constexpr int U = 4;
  const int grid = g_cus * 4;
  {
    const double ms = time_ms([&] {
      hipLaunchKernelGGL((window_pass_kernel<U, 0>), dim3(grid), dim3(256), 0, 0, t, a, (size_t)0, tn);
      hipLaunchKernelGGL((window_pass_kernel<U, 1>), dim3(grid), dim3(256), 0, 0, t, a, (size_t)0, tn);
    }, 10);
    snprintf(buf, sizeof buf, "\"layer_major\": {\"ms\": %.4f, \"GBps_algorithmic\": %.1f},\n", ms, 5.0 * tbytes / 1e9 / ms * 1e3);
    out += buf;
  }
  out += "\"chunk_major_separate_launches\": [\n";
  const int chunk_mib[] = {4, 8, 16, 32, 64, 128, 256};
  for (int ci = 0; ci < 7; ++ci) {
    const size_t cb = (size_t)chunk_mib[ci] << 20, cn = cb / 16;
    const int nchunk = (int)(tn / cn);
    const double ms = time_ms([&] {
      for (int c = 0; c < nchunk; ++c) {
        hipLaunchKernelGGL((window_pass_kernel<U, 0>), dim3(grid), dim3(256), 0, 0, t, a, c * cn, (c + 1) * cn);
        hipLaunchKernelGGL((window_pass_kernel<U, 1>), dim3(grid), dim3(256), 0, 0, t, a, c * cn, (c + 1) * cn);
      }
    }, 10);
    snprintf(buf, sizeof buf, "%s{\"chunk_MiB\": %d, \"launches\": %d, \"ms\": %.4f, \"GBps_algorithmic\": %.1f}", ci ? ",\n" : "",
             chunk_mib[ci], 2 * nchunk, ms, 5.0 * tbytes / 1e9 / ms * 1e3);
    out += buf;
  }
  out += "\n],\n\"chunk_major_one_launch_interleaved\": [\n";
  bool f2 = true;
  for (int ci = 0; ci < 6; ++ci) {
    for (int lag = 1; lag <= 2; ++lag) {
      const size_t cb = (size_t)chunk_mib[ci] << 20, cn = cb / 16;
      const int nchunk = (int)(tn / cn);
      const int tiles = (int)(cn / (256 * U));
      const double ms = time_ms([&] {
        hipLaunchKernelGGL((window_interleaved_kernel<U>), dim3(grid), dim3(256), 0, 0, t, a, tn, cn, nchunk, lag, tiles);
      }, 10);
      snprintf(buf, sizeof buf, "%s{\"chunk_MiB\": %d, \"lag_chunks\": %d, \"ms\": %.4f, \"GBps_algorithmic\": %.1f}", f2 ? "" : ",\n",
               chunk_mib[ci], lag, ms, 5.0 * tbytes / 1e9 / ms * 1e3);
      f2 = false;
      out += buf;
    }
  }
  out += "\n],\n\"chunk_major_one_launch_interleaved_sc1_handoff\": [\n";
  f2 = true;
  for (int ci = 0; ci < 5; ++ci) {
    for (int lag = 1; lag <= 2; ++lag) {
      const size_t cb = (size_t)chunk_mib[ci] << 20, cn = cb / 16;
      const int nchunk = (int)(tn / cn);
      const int tiles = (int)(cn / (256 * U));
      const double ms = time_ms([&] {
        hipLaunchKernelGGL((window_interleaved_kernel<U, true>), dim3(grid), dim3(256), 0, 0, t, a, tn, cn, nchunk, lag, tiles);
      }, 10);
      snprintf(buf, sizeof buf, "%s{\"chunk_MiB\": %d, \"lag_chunks\": %d, \"ms\": %.4f, \"GBps_algorithmic\": %.1f}", f2 ? "" : ",\n",
               chunk_mib[ci], lag, ms, 5.0 * tbytes / 1e9 / ms * 1e3);
      f2 = false;
      out += buf;
    }
  }
  out += "\n]}\n}";
  puts(out.c_str());
  return 0;
}
