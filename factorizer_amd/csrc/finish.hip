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
// One queue per stream (a stream belongs to one device): the jobs of a queue read what kernels on THAT stream produced.  The
// table is process-wide and mutex-guarded; WHETHER a call queues is a per-thread flag (the thread that sets it — an autograd
// device thread running one backward — is the thread that issues the weight-gradient calls it covers; another thread's calls
// in the same process stay immediate).
struct FinishQueue {
  hipStream_t stream;
  int device;   // the device that was current when the queue was opened: the stream's device
  std::vector<FinishJob> jobs;
};
std::mutex g_mu;
std::vector<FinishQueue> g_queues;
thread_local int t_defer = 0;

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

// jobs phase by phase, in the given order within a phase
int launch_phases(const FinishJob* jobs, int n, hipStream_t st) {
  int maxph = 0;
  for (int i = 0; i < n; ++i) maxph = jobs[i].phase > maxph ? jobs[i].phase : maxph;
  if (maxph == 0) return launch_jobs(jobs, n, st);
  std::vector<FinishJob> sel;
  sel.reserve((size_t)n);
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

FinishQueue* find_queue(hipStream_t st) {
  for (auto& q : g_queues)
    if (q.stream == st) return &q;
  return nullptr;
}

// everything queued for one stream, on that stream (caller holds g_mu); returns the number of jobs or an error code
int flush_queue_locked(FinishQueue& q) {
  const int n = (int)q.jobs.size();
  if (n == 0) return 0;
  int cur = q.device;
  (void)hipGetDevice(&cur);
  if (cur != q.device) FZ_HIP_OK(hipSetDevice(q.device));
  const int rc = launch_phases(q.jobs.data(), n, q.stream);
  if (cur != q.device) (void)hipSetDevice(cur);
  q.jobs.clear();
  return rc == FZ_OK ? n : rc;
}
}  // namespace

int finish_run(const FinishJob* jobs, int n, hipStream_t st) {
  if (n < 1) return FZ_OK;
  bool acc = false;
  for (int i = 0; i < n; ++i) acc = acc || accumulates(jobs[i]);
  const bool defer = t_defer > 0 && !acc;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    FinishQueue* q = find_queue(st);
    if (acc && q != nullptr) {   // a job that ADDS to earlier results: those results must exist first
      const int rc = flush_queue_locked(*q);
      if (rc < 0) return rc;
    }
    if (defer) {
      if (q == nullptr) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        g_queues.push_back(FinishQueue{st, dev, {}});
        q = &g_queues.back();
      }
      q->jobs.insert(q->jobs.end(), jobs, jobs + n);
      return FZ_OK;
    }
  }
  return launch_phases(jobs, n, st);
}

}  // namespace fz

using namespace fz;

extern "C" int fz_finish_defer(int on) {
  const int was = t_defer;
  if (on >= 0) t_defer = on ? 1 : 0;
  return was;
}

extern "C" int fz_finish_pending(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  size_t n = 0;
  for (const auto& q : g_queues) n += q.jobs.size();
  return (int)n;
}

extern "C" int fz_finish_flush(fz_stream_t stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  FinishQueue* q = find_queue((hipStream_t)stream);
  return q == nullptr ? 0 : flush_queue_locked(*q);
}

extern "C" int fz_finish_flush_all(fz_stream_t waiter) {
  std::lock_guard<std::mutex> lk(g_mu);
  int total = 0;
  for (auto& q : g_queues) {
    if (q.jobs.empty()) continue;
    const int rc = flush_queue_locked(q);
    if (rc < 0) return rc;
    total += rc;
    if (q.stream != (hipStream_t)waiter) {   // whoever reads the gradients on `waiter` is ordered behind the queue's own stream
      hipEvent_t ev;
      FZ_HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      hipError_t e = hipEventRecord(ev, q.stream);
      if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
      hipEventDestroy(ev);
      if (e != hipSuccess) return fail(FZ_E_HIP, hipGetErrorString(e));
    }
  }
  return total;
}
