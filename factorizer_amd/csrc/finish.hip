// finish.hip — the table-driven finish kernel and the deferred-finish queue (finish.h has the job types, the device bodies and
// the reason).  C ABI: fz_finish_defer / fz_finish_pending / fz_finish_flush (include/factorizer_hip.h).
#include "finish.h"

#include <mutex>
#include <vector>

namespace fz {

template <int N>
struct FinishTable {
  FinishJob j[N];
  int start[N + 1];
  int n;
};
constexpr int kFinishBatch = 32;   // 32 x 104 B of jobs + 33 starts: inside the 4 KB of kernel arguments

template <int N>
__global__ __launch_bounds__(256) void finish_batch_kernel(FinishTable<N> t) {
  __shared__ float red9[32][9];
  __shared__ float red33[8][33];
  __shared__ float s17[16][17];
  __shared__ float s1[2];
  int i = 0;
  for (int k = 1; k < t.n; ++k)
    if ((int)blockIdx.x >= t.start[k]) i = k;
  const FinishJob& J = t.j[i];
  const int vb = (int)blockIdx.x - t.start[i];
  switch (J.kind) {
    case FK_ROWS: fin_rows_body(J.u.rows, vb, red33); break;
    case FK_CHUNK: fin_chunk_body(J.u.chunk, vb, red9); break;
    case FK_WGRAD: {
      const WgradFinishOne& f = J.u.wg;
      if (f.wide == 3) wgrad_finish_wide4_body(f.part, f.part_bias, f.nchunk, f.M, f.K, f.gw, f.gbias, f.ln_g, f.ln_b, f.accumulate, f.nbw, vb);
      else if (f.wide == 2) wgrad_finish_mid_body(f.part, f.part_bias, f.nchunk, f.M, f.K, f.gw, f.gbias, f.ln_g, f.ln_b, f.accumulate, f.nbw, &red33[0][0], vb);
      else if (f.wide) wgrad_finish_wide_body(f.part, f.part_bias, f.nchunk, f.M, f.K, f.gw, f.gbias, f.ln_g, f.ln_b, f.accumulate, f.nbw, vb);
      else wgrad_finish_body(f.part, f.part_bias, f.nchunk, f.M, f.K, f.gw, f.gbias, f.ln_g, f.ln_b, f.accumulate, f.nbw, red9, vb);
      break;
    }
    case FK_DW: fin_dw_body(J.u.dw, vb, s17, &s1[0]); break;
    case FK_CHAIN_WG: fin_chain_wg_body(J.u.cw, vb, s17, &s1[0]); break;
    default: fin_upcat_body(J.u.up, vb, red33); break;
  }
}

namespace {
std::mutex g_mu;
std::vector<FinishJob> g_queue;
hipStream_t g_queue_stream = nullptr;
std::atomic<int> g_defer{0};

template <int N>
int launch_table(const FinishJob* jobs, int n, hipStream_t st) {
  FinishTable<N> t;
  int total = 0;
  for (int i = 0; i < N; ++i) {
    t.j[i] = jobs[i < n ? i : n - 1];
    t.start[i] = total;
    if (i < n) total += jobs[i].nblocks;
  }
  t.start[N] = total;
  t.n = n;
  if (total < 1) return FZ_OK;
  hipLaunchKernelGGL(finish_batch_kernel<N>, dim3((unsigned)total), dim3(256), 0, st, t);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

int launch_jobs(const FinishJob* jobs, int n, hipStream_t st) {
  for (int i = 0; i < n; i += kFinishBatch) {
    const int m = n - i < kFinishBatch ? n - i : kFinishBatch;
    int rc;
    if (m == 1) rc = launch_table<1>(jobs + i, 1, st);
    else if (m <= 4) rc = launch_table<4>(jobs + i, m, st);
    else rc = launch_table<kFinishBatch>(jobs + i, m, st);
    if (rc != FZ_OK) return rc;
  }
  return FZ_OK;
}

bool accumulates(const FinishJob& j) {
  return (j.kind == FK_WGRAD && j.u.wg.accumulate) || (j.kind == FK_CHUNK && j.u.chunk.accumulate);
}

// everything queued, phase by phase, in queue order within a phase (caller holds g_mu)
int flush_locked(hipStream_t st) {
  if (g_queue.empty()) return 0;
  const int n = (int)g_queue.size();
  int maxph = 0;
  for (const auto& j : g_queue) maxph = j.phase > maxph ? j.phase : maxph;
  std::vector<FinishJob> sel;
  sel.reserve(g_queue.size());
  int rc = FZ_OK;
  for (int ph = 0; ph <= maxph && rc == FZ_OK; ++ph) {
    sel.clear();
    for (const auto& j : g_queue)
      if (j.phase == ph) sel.push_back(j);
    if (!sel.empty()) rc = launch_jobs(sel.data(), (int)sel.size(), st);
  }
  g_queue.clear();
  g_queue_stream = nullptr;
  return rc == FZ_OK ? n : rc;
}
}  // namespace

int finish_run(const FinishJob* jobs, int n, hipStream_t st) {
  if (n < 1) return FZ_OK;
  bool acc = false;
  for (int i = 0; i < n; ++i) acc = acc || accumulates(jobs[i]);
  if (g_defer.load(std::memory_order_relaxed) > 0 || acc) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_queue.empty() && (acc || g_queue_stream != st)) {   // (a queue that belongs to another stream, or a job that adds to earlier results)
      const int rc = flush_locked(g_queue_stream);
      if (rc < 0) return rc;
    }
    if (!acc && g_defer.load(std::memory_order_relaxed) > 0) {
      g_queue.insert(g_queue.end(), jobs, jobs + n);
      g_queue_stream = st;
      return FZ_OK;
    }
  }
  // now: jobs of one call are independent of each other except through their phases
  int maxph = 0;
  for (int i = 0; i < n; ++i) maxph = jobs[i].phase > maxph ? jobs[i].phase : maxph;
  if (maxph == 0) return launch_jobs(jobs, n, st);
  std::vector<FinishJob> sel;
  for (int ph = 0; ph <= maxph; ++ph) {
    sel.clear();
    for (int i = 0; i < n; ++i)
      if (jobs[i].phase == ph) sel.push_back(jobs[i]);
    if (!sel.empty()) {
      const int rc = launch_jobs(sel.data(), (int)sel.size(), st);
      if (rc != FZ_OK) return rc;
    }
  }
  return FZ_OK;
}

}  // namespace fz

using namespace fz;

extern "C" int fz_finish_defer(int on) {
  if (on < 0) return g_defer.load();
  return g_defer.exchange(on ? 1 : 0);
}

extern "C" int fz_finish_pending(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return (int)g_queue.size();
}

extern "C" int fz_finish_flush(fz_stream_t stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_queue.empty()) return 0;
  // the jobs read what kernels on THEIR stream produced: a flush on another stream would race with them
  if (g_queue_stream != (hipStream_t)stream) return fail(FZ_E_ARG, "fz_finish_flush: the queued finishes belong to another stream");
  return flush_locked((hipStream_t)stream);
}
