// api.hip -- C ABI (include/gpx.h) and the host-side drivers of the HIP path:
// the bordered right-looking blocked Cholesky with two-stream look-ahead, the
// solves against a cached factor, and the measurement hooks.

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "gpx_internal.h"

static std::string g_create_error;

static std::mutex g_err_mutex;

void HostWorker::start(int device) {
  if (started) return;
  started = true;
  th = std::thread([this, device]() {
    (void)hipSetDevice(device);
    for (;;) {
      int spins = 0;
      while (state.load(std::memory_order_acquire) == 0) {   // poll ~0.3 ms for the next job, then sleep
        if (++spins > 20000) {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [this] { return state.load(std::memory_order_acquire) != 0; });
          break;
        }
      }
      if (state.load(std::memory_order_acquire) == 3) return;
      state.store(2, std::memory_order_release);
      job();
      state.store(0, std::memory_order_release);
    }
  });
}
void HostWorker::post(std::function<void()> f) {
  job = std::move(f);
  { std::lock_guard<std::mutex> lk(m); state.store(1, std::memory_order_release); }
  cv.notify_one();
}
void HostWorker::wait() {
  while (state.load(std::memory_order_acquire) != 0) { }
}
void HostWorker::stop() {
  if (!started) return;
  wait();
  { std::lock_guard<std::mutex> lk(m); state.store(3, std::memory_order_release); }
  cv.notify_one();
  th.join();
  started = false;
}

void gpx_set_error(gpx_handle* h, const std::string& msg) {
  std::lock_guard<std::mutex> lk(g_err_mutex);
  if (h) h->err = msg; else g_create_error = msg;
}

// --------------------------------------------------------------------------- memory

template <typename T>
static int ensure(gpx_handle* h, T** p, int64_t* cap, int64_t count) {
  if (count <= *cap && *p) return GPX_OK;
  if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
  if (count <= 0) count = 1;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), (size_t)count * sizeof(T));
  if (e != hipSuccess) {
    gpx_set_error(h, std::string("hipMalloc failed: ") + hipGetErrorString(e));
    *p = nullptr;
    return GPX_ERR_NOMEM;
  }
  *cap = count;
  return GPX_OK;
}

// ------------------------------------------------------------------------ profiling

static hipEvent_t prof_event(gpx_handle* h) {
  if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence); return e;   // timing on; device-scope release only, like the hand-over events
}

// mode 1: every launch is bracketed; mode 2 (coarse): only the whole step, the gram build and the epilogue --
// three event pairs per step, so the brackets do not stretch what they measure; mode 3: only the launches of the
// two kernels that carry a roofline (bulk update on its own stream, gram build): nothing on the chain stream
static bool prof_stage_on(const gpx_handle* h, int stage) {
  if (h->prof_mode == 1) return true;
  if (h->prof_mode == 2) return stage == GPX_STAGE_FACTOR_TOTAL || stage == GPX_STAGE_GRAM || stage == GPX_STAGE_EPILOGUE;
  if (h->prof_mode == 3) return stage == GPX_STAGE_TRAILING || stage == GPX_STAGE_GRAM;   // the two roofline kernels only
  return false;
}

void prof_begin(gpx_handle* h, hipStream_t s, int stage, double flops, double bytes) {
  if (!h->prof_mode) return;
  h->prof_open = prof_stage_on(h, stage);
  if (!h->prof_open) return;
  gpx_handle::PendingEv p;
  p.stage = stage; p.a = prof_event(h); p.b = prof_event(h);
  (void)hipEventRecord(p.a, s);
  h->pending.push_back(p);
  h->prof[stage].flops += flops;
  h->prof[stage].bytes += bytes;
  h->prof[stage].launches += 1;
}

bool prof_pair(gpx_handle* h, int stage, double flops, double bytes, hipEvent_t* a, hipEvent_t* b) {
  *a = *b = nullptr;
  if (!h->prof_mode || !prof_stage_on(h, stage)) return false;
  gpx_handle::PendingEv p;
  p.stage = stage; p.a = prof_event(h); p.b = prof_event(h);
  h->pending.push_back(p);
  h->prof[stage].flops += flops;
  h->prof[stage].bytes += bytes;
  h->prof[stage].launches += 1;
  *a = p.a; *b = p.b;
  return true;
}

void prof_end(gpx_handle* h, hipStream_t s) {
  if (!h->prof_mode || !h->prof_open) return;
  (void)hipEventRecord(h->pending.back().b, s);
}

static int prof_collect(gpx_handle* h) {
  if (h->pending.empty()) return GPX_OK;
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_main));
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_trail));
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_far));
  for (auto& p : h->pending) {
    float ms = 0.f;
    GPX_HIP_CHECK(h, hipEventElapsedTime(&ms, p.a, p.b));
    h->prof[p.stage].ms += ms;
    h->ev_pool.push_back(p.a);
    h->ev_pool.push_back(p.b);
  }
  h->pending.clear();
  return GPX_OK;
}

// -------------------------------------------------------------------- create/destroy

// ---- the result block and its one-copy fetch
#define GPX_OUT_HEAD 10   // doubles in front of mean: scal[8], then 4 ints of info

// (Re)allocate the result block for `count` doubles of mean + var; the head (scal, info) survives a growth.
static int ensure_out(gpx_handle* h, int64_t count) {
  if (h->d_out && h->out_cap >= count) return GPX_OK;
  double* nw = nullptr;
  GPX_HIP_CHECK(h, hipMalloc(reinterpret_cast<void**>(&nw), (size_t)(GPX_OUT_HEAD + count) * sizeof(double)));
  if (h->d_out) {
    GPX_HIP_CHECK(h, hipDeviceSynchronize());
    GPX_HIP_CHECK(h, hipMemcpy(nw, h->d_out, GPX_OUT_HEAD * sizeof(double), hipMemcpyDeviceToDevice));
    (void)hipFree(h->d_out);
  } else {
    GPX_HIP_CHECK(h, hipMemset(nw, 0, GPX_OUT_HEAD * sizeof(double)));
  }
  h->d_out = nw; h->out_cap = count;
  h->d_scal = nw; h->d_info = reinterpret_cast<int*>(nw + 8); h->d_mean = nw + GPX_OUT_HEAD; h->d_var = h->d_mean + 1;
  return GPX_OK;
}

// One device-to-host copy (into pinned memory) of [scal | info | first `tail` doubles after the head], then a
// stream sync.  Returns a pointer to the pinned copy: scal at [0], info at ((int*)(p + 8))[0], mean at p + 10.
static int fetch_block(gpx_handle* h, int64_t tail, const double** out) {
  const int64_t cnt = GPX_OUT_HEAD + tail;
  if (h->pinned_cap < cnt) {
    if (h->h_pinned) (void)hipHostFree(h->h_pinned);
    h->h_pinned = nullptr; h->pinned_cap = 0;
    GPX_HIP_CHECK(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_pinned), (size_t)cnt * sizeof(double), hipHostMallocDefault));
    h->pinned_cap = cnt;
  }
  GPX_HIP_CHECK(h, hipMemcpyAsync(h->h_pinned, h->d_out, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, h->s_main));
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_main));
  *out = h->h_pinned;
  return GPX_OK;
}

// Host -> device copy of a few doubles through a ring of pinned slots: truly asynchronous, and the source may
// be reused by the caller immediately.  A slot is recycled only after its previous copy has completed.
static int upload_small(gpx_handle* h, hipStream_t s, double* dst, const double* src, int64_t count) {
  if (h->up_slot < count) {
    GPX_HIP_CHECK(h, hipDeviceSynchronize());
    if (h->h_up) (void)hipHostFree(h->h_up);
    h->h_up = nullptr;
    const int64_t slot = std::max<int64_t>(count, 64);
    GPX_HIP_CHECK(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_up), (size_t)(8 * slot) * sizeof(double), hipHostMallocDefault));
    h->up_slot = slot;
    for (auto& u : h->up_used) u = false;
  }
  const int i = h->up_next;
  h->up_next = (i + 1) % 8;
  if (!h->up_ev[i]) GPX_HIP_CHECK(h, hipEventCreateWithFlags(&h->up_ev[i], hipEventDisableTiming));
  if (h->up_used[i]) GPX_HIP_CHECK(h, hipEventSynchronize(h->up_ev[i]));
  double* stage = h->h_up + (int64_t)i * h->up_slot;
  memcpy(stage, src, (size_t)count * sizeof(double));
  GPX_HIP_CHECK(h, hipMemcpyAsync(dst, stage, (size_t)count * sizeof(double), hipMemcpyHostToDevice, s));
  GPX_HIP_CHECK(h, hipEventRecord(h->up_ev[i], s));
  h->up_used[i] = true;
  return GPX_OK;
}

// The same ring without the copy: the data is staged in a pinned slot that a kernel reads in place (pinned host memory is
// device-addressable); the caller records the slot's event behind that kernel with stage_small_done.
static int stage_small(gpx_handle* h, const double* src, int64_t count, const double** staged, int* slot) {
  if (h->up_slot < count) {
    GPX_HIP_CHECK(h, hipDeviceSynchronize());
    if (h->h_up) (void)hipHostFree(h->h_up);
    h->h_up = nullptr;
    const int64_t sl = std::max<int64_t>(count, 64);
    GPX_HIP_CHECK(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_up), (size_t)(8 * sl) * sizeof(double), hipHostMallocDefault));
    h->up_slot = sl;
    for (auto& u : h->up_used) u = false;
  }
  const int i = h->up_next;
  h->up_next = (i + 1) % 8;
  if (!h->up_ev[i]) GPX_HIP_CHECK(h, hipEventCreateWithFlags(&h->up_ev[i], hipEventDisableTiming));
  if (h->up_used[i]) GPX_HIP_CHECK(h, hipEventSynchronize(h->up_ev[i]));
  double* stage = h->h_up + (int64_t)i * h->up_slot;
  memcpy(stage, src, (size_t)count * sizeof(double));
  *staged = stage; *slot = i;
  return GPX_OK;
}
static int stage_small_done(gpx_handle* h, hipStream_t s, int slot) {
  GPX_HIP_CHECK(h, hipEventRecord(h->up_ev[slot], s));
  h->up_used[slot] = true;
  return GPX_OK;
}

#ifdef GPX_TIMELINE
void tl_set_gemm(unsigned long long* p, unsigned long long* ends);
void tl_set_potrf(unsigned long long* p);
static unsigned long long* g_tl_dev = nullptr;
static unsigned long long* g_tl_ends = nullptr;
extern "C" int gpx_timeline_start(void) {
  if (!g_tl_dev && hipMalloc(reinterpret_cast<void**>(&g_tl_dev), (4 + 4 * 16000) * 8) != hipSuccess) return -1;
  if (!g_tl_ends && hipMalloc(reinterpret_cast<void**>(&g_tl_ends), 8192 * 8) != hipSuccess) return -1;
  (void)hipMemset(g_tl_dev, 0, (4 + 4 * 16000) * 8);
  (void)hipMemset(g_tl_ends, 0, 8192 * 8);
  tl_set_gemm(g_tl_dev, g_tl_ends); tl_set_potrf(g_tl_dev);
  return 0;
}
extern "C" long long gpx_timeline_read_ends(unsigned long long* out, long long max_records) {
  (void)hipDeviceSynchronize();
  const long long n = max_records < 8192 ? max_records : 8192;
  (void)hipMemcpy(out, g_tl_ends, (size_t)n * 8, hipMemcpyDeviceToHost);
  return n;
}
extern "C" long long gpx_timeline_read(unsigned long long* out, long long max_records) {
  unsigned long long n = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&n, g_tl_dev, 8, hipMemcpyDeviceToHost);
  if (n > 16000) n = 16000;
  if ((long long)n > max_records) n = (unsigned long long)max_records;
  (void)hipMemcpy(out, g_tl_dev + 4, n * 32, hipMemcpyDeviceToHost);
  return (long long)n;
}
#endif

extern "C" int gpx_abi_version(void) { return GPX_ABI_VERSION; }

// Three streams.  s_main carries the critical path (diagonal block, panel solve, near look-ahead) at
// the highest priority on all CUs.  s_far (far look-ahead) and s_trail (bulk trailing update) may be
// put on a CU mask that leaves GPX_RESERVED_CUS CUs untouched (default 0 = no mask: measured no gain
// once the bulk kernel is released only after the near look-ahead has been queued, see factor_loop).
static hipError_t create_streams(gpx_handle* h) {
  int lo = 0, hi = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = least, hi = greatest priority
  if (e != hipSuccess) return e;
  if ((e = hipStreamCreateWithPriority(&h->s_main, hipStreamNonBlocking, hi)) != hipSuccess) return e;
  if ((e = hipStreamCreateWithPriority(&h->s_main2, hipStreamNonBlocking, hi)) != hipSuccess) return e;
  const char* env = getenv("GPX_RESERVED_CUS");
  const int ncu = h->prop.multiProcessorCount;
  int drop = env ? atoi(env) : 0;   // measured: no gain once the bulk kernel is released after the near look-ahead
  if (drop > 0 && drop < ncu) {
    // Mask bit i is CU i/8 of XCD i%8 (measured: excluding every 16th bit halves ONE XCD and the bulk kernel, whose
    // workgroups are dealt round-robin over the XCDs, slows to that XCD's pace).  The first `drop` bits (a multiple
    // of 8) therefore take drop/8 CUs from every XCD.  GPX_RESERVED_SPREAD=1: the old every-(ncu/drop)-th rule.
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    const bool spread = getenv("GPX_RESERVED_SPREAD") && atoi(getenv("GPX_RESERVED_SPREAD")) != 0;
    if (!spread) drop = (drop + 7) / 8 * 8;
    for (int i = 0; i < ncu; ++i) {
      const bool excluded = spread ? ((int64_t)i * drop / ncu) != ((int64_t)(i + 1) * drop / ncu) : i < drop;
      if (!excluded) mask[(size_t)i / 32] |= 1u << (i % 32);
    }
    // only the bulk stream T is kept off the reserved CUs (GPX_RESERVED_Q=1: the panel stream Q too)
    const bool mask_q = getenv("GPX_RESERVED_Q") && atoi(getenv("GPX_RESERVED_Q")) != 0;
    hipError_t e1 = hipExtStreamCreateWithCUMask(&h->s_trail, (uint32_t)mask.size(), mask.data());
    hipError_t e2 = mask_q ? hipExtStreamCreateWithCUMask(&h->s_far, (uint32_t)mask.size(), mask.data())
                           : hipStreamCreateWithPriority(&h->s_far, hipStreamNonBlocking, hi);
    if (e1 == hipSuccess && e2 == hipSuccess) return hipSuccess;
    if (e1 == hipSuccess) { (void)hipStreamDestroy(h->s_trail); h->s_trail = nullptr; }
    if (e2 == hipSuccess) { (void)hipStreamDestroy(h->s_far); h->s_far = nullptr; }
  }
  if ((e = hipStreamCreateWithPriority(&h->s_far, hipStreamNonBlocking, hi)) != hipSuccess) return e;
  return hipStreamCreateWithPriority(&h->s_trail, hipStreamNonBlocking, lo);
}

extern "C" int gpx_create(int device_id, gpx_handle** out) {
  if (!out) return GPX_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0");
    return GPX_ERR_HIP;
  }
  if (device_id < 0 || device_id >= ndev) { g_create_error = "device_id out of range"; return GPX_ERR_ARG; }
  gpx_handle* h = new gpx_handle();
  h->device = device_id;
  if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&h->prop, device_id)) != hipSuccess ||
      (e = create_streams(h)) != hipSuccess) {
    g_create_error = std::string("gpx_create: ") + hipGetErrorString(e);
    delete h;
    return GPX_ERR_HIP;
  }
  if (ensure_out(h, 2) != GPX_OK) {
    g_create_error = std::string("gpx_create: ") + h->err;
    delete h;
    return GPX_ERR_HIP;
  }
  if (strncmp(h->prop.gcnArchName, "gfx950", 6) != 0) {
    // The code object is gfx950-only; say so instead of failing at the first launch.
    g_create_error = std::string("gpx_create: device is ") + h->prop.gcnArchName + ", this library is built for gfx950";
    delete h;
    return GPX_ERR_UNSUPPORTED;
  }
  const char* env;
  if ((env = getenv("GPX_NB"))) { h->nb = atoi(env) == 64 ? 64 : 128; h->nb_auto = false; }
  if ((env = getenv("GPX_LOOKAHEAD"))) h->lookahead = atoi(env) != 0;
  if ((env = getenv("GPX_GRAPH"))) h->use_graph = atoi(env) != 0;
  if ((env = getenv("GPX_DEPTH"))) h->depth = std::min(std::max(atoi(env), 1), 3);
  if ((env = getenv("GPX_PAIR_MIN"))) h->pair_min = std::max(atoi(env), 0);
  if ((env = getenv("GPX_QUAD_MIN"))) h->quad_min = std::max(atoi(env), 0);
  if ((env = getenv("GPX_LA_ROWS"))) h->la_rows = std::max(atoi(env), 0);
  if ((env = getenv("GPX_COL_SPLIT"))) h->col_split = std::max(atoi(env), 0);
  if ((env = getenv("GPX_NEAR_BLOCK"))) h->near_block = atoi(env) != 0;
  if ((env = getenv("GPX_FUSE_BAND"))) h->fuse_band = atoi(env) != 0;
  if ((env = getenv("GPX_BORDER_SIDE"))) h->border_side = atoi(env) != 0;
  if ((env = getenv("GPX_DIAG_BAND"))) h->diag_band = atoi(env) != 0;
  if ((env = getenv("GPX_LP_FLAG"))) h->lp_flag = atoi(env) != 0;
  if ((env = getenv("GPX_EPI_SPLIT"))) h->epi_split = atoi(env) != 0;
  if ((env = getenv("GPX_POTRF_PERSIST"))) h->potrf_persist = atoi(env) != 0;
  if ((env = getenv("GPX_BUILD_SMALL"))) h->build_small = atoi(env) != 0;
  if ((env = getenv("GPX_SPLIT_MIN_BLOCKS"))) h->split_min_blocks = std::max(atoi(env), 2);
  if ((env = getenv("GPX_HOST_THREADS"))) h->host_threads = atoi(env) >= 2 ? 2 : 1;
  // in-kernel hand-overs need the chain's kernels to run CONCURRENTLY; hardware-counter collection
  // (rocprofv3 --pmc) serialises every dispatch, so fall back to kernel-boundary hand-overs under it
  if (getenv("ROCPROF_COUNTER_COLLECTION") || getenv("ROCPROF_COUNTERS")) h->resident = 0;
  if ((env = getenv("GPX_RESIDENT"))) h->resident = std::min(std::max(atoi(env), 0), 2);
  *out = h;
  return GPX_OK;
}

extern "C" int gpx_destroy(gpx_handle* h) {
  if (!h) return GPX_OK;
  (void)hipSetDevice(h->device);
  h->worker.stop();
  (void)hipDeviceSynchronize();
  gpx_dist_free(h);
  void* bufs[] = {h->d_x, h->d_xnew, h->d_r, h->d_mnew, h->d_hyp, h->d_noise, h->d_A, h->d_L, h->d_W,
                  h->d_S, h->d_cov, h->d_z, h->d_out, h->d_g1, h->d_g2,
                  h->d_gout, h->d_vec, h->d_A2, h->d_L2, h->d_W2, h->d_kinv, h->d_gpart, h->d_mini};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (h->h_pinned) (void)hipHostFree(h->h_pinned);
  if (h->h_up) (void)hipHostFree(h->h_up);
  for (auto e : h->up_ev) if (e) (void)hipEventDestroy(e);
  for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_join2) (void)hipEventDestroy(h->ev_join2);
  for (auto e : h->ev_far) (void)hipEventDestroy(e);
  for (auto e : h->ev_w) (void)hipEventDestroy(e);
  for (auto e : h->ev_panel) (void)hipEventDestroy(e);
  for (auto e : h->ev_trail) (void)hipEventDestroy(e);
  for (auto e : h->ev_pool) (void)hipEventDestroy(e);
  for (auto& p : h->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  if (h->s_main) (void)hipStreamDestroy(h->s_main);
  if (h->s_main2) (void)hipStreamDestroy(h->s_main2);
  if (h->ev_chain2) (void)hipEventDestroy(h->ev_chain2);
  if (h->ev_build) (void)hipEventDestroy(h->ev_build);
  if (h->d_flags) (void)hipFree(h->d_flags);
  if (h->d_epart) (void)hipFree(h->d_epart);
  if (h->s_trail) (void)hipStreamDestroy(h->s_trail);
  if (h->s_far) (void)hipStreamDestroy(h->s_far);
  delete h;
  return GPX_OK;
}

extern "C" const char* gpx_last_error(const gpx_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int gpx_device_name(gpx_handle* h, char* buf, int len) {
  if (!h || !buf || len <= 0) return GPX_ERR_ARG;
  snprintf(buf, (size_t)len, "%s (%s, %d CUs)", h->prop.name, h->prop.gcnArchName, h->prop.multiProcessorCount);
  return GPX_OK;
}

extern "C" int gpx_synchronize(gpx_handle* h) {
  if (!h) return GPX_ERR_ARG;
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_main));
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_trail));
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_far));
  return GPX_OK;
}

extern "C" int gpx_set_option(gpx_handle* h, const char* name, int64_t value) {
  if (!h || !name) return GPX_ERR_ARG;
  if (!strcmp(name, "nb")) {
    if (value != 0 && value != 64 && value != 128) { gpx_set_error(h, "nb must be 64, 128 or 0 (automatic)"); return GPX_ERR_ARG; }
    const int nv = value ? (int)value : ((h->n > 0 && h->n <= 64) ? 64 : 128);
    h->nb_auto = value == 0;
    if (h->nb != nv) { h->nb = nv; h->have_factor = false; }
    return GPX_OK;
  }
  if (!strcmp(name, "lookahead")) { h->lookahead = value != 0; return GPX_OK; }
  if (!strcmp(name, "graph")) { h->use_graph = value != 0; return GPX_OK; }
  if (!strcmp(name, "depth")) { h->depth = (int)std::min<int64_t>(std::max<int64_t>(value, 1), 3); return GPX_OK; }
  if (!strcmp(name, "pair_min")) { h->pair_min = (int)std::max<int64_t>(value, 0); return GPX_OK; }
  if (!strcmp(name, "quad_min")) { h->quad_min = (int)std::max<int64_t>(value, 0); return GPX_OK; }
  if (!strcmp(name, "la_rows")) { h->la_rows = (int)std::max<int64_t>(value, 0); return GPX_OK; }
  if (!strcmp(name, "col_split")) { h->col_split = (int)std::max<int64_t>(value, 0); return GPX_OK; }
  if (!strcmp(name, "near_block")) { h->near_block = value != 0; return GPX_OK; }
  if (!strcmp(name, "fuse_band")) { h->fuse_band = value != 0; return GPX_OK; }
  if (!strcmp(name, "border_side")) { h->border_side = value != 0; return GPX_OK; }
  if (!strcmp(name, "diag_band")) { h->diag_band = value != 0; return GPX_OK; }
  if (!strcmp(name, "lp_flag")) { h->lp_flag = value != 0; return GPX_OK; }
  if (!strcmp(name, "epi_split")) { h->epi_split = value != 0; return GPX_OK; }
  if (!strcmp(name, "potrf_persist")) { h->potrf_persist = value != 0; return GPX_OK; }
  if (!strcmp(name, "build_small")) { h->build_small = value != 0; return GPX_OK; }
  if (!strcmp(name, "split_min_blocks")) { h->split_min_blocks = (int)std::max<int64_t>(value, 2); return GPX_OK; }
  if (!strcmp(name, "host_threads")) { h->host_threads = value >= 2 ? 2 : 1; return GPX_OK; }
  if (!strcmp(name, "resident")) { h->resident = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 2); return GPX_OK; }
  gpx_set_error(h, std::string("unknown option ") + name);
  return GPX_ERR_ARG;
}

extern "C" int gpx_get_option(gpx_handle* h, const char* name, int64_t* value) {
  if (!h || !name || !value) return GPX_ERR_ARG;
  if (!strcmp(name, "nb")) { *value = h->nb; return GPX_OK; }
  if (!strcmp(name, "lookahead")) { *value = h->lookahead; return GPX_OK; }
  if (!strcmp(name, "graph")) { *value = h->use_graph; return GPX_OK; }
  if (!strcmp(name, "depth")) { *value = h->depth; return GPX_OK; }
  if (!strcmp(name, "pair_min")) { *value = h->pair_min; return GPX_OK; }
  if (!strcmp(name, "quad_min")) { *value = h->quad_min; return GPX_OK; }
  if (!strcmp(name, "la_rows")) { *value = h->la_rows; return GPX_OK; }
  if (!strcmp(name, "col_split")) { *value = h->col_split; return GPX_OK; }
  if (!strcmp(name, "near_block")) { *value = h->near_block; return GPX_OK; }
  if (!strcmp(name, "fuse_band")) { *value = h->fuse_band; return GPX_OK; }
  if (!strcmp(name, "border_side")) { *value = h->border_side; return GPX_OK; }
  if (!strcmp(name, "diag_band")) { *value = h->diag_band; return GPX_OK; }
  if (!strcmp(name, "lp_flag")) { *value = h->lp_flag; return GPX_OK; }
  if (!strcmp(name, "epi_split")) { *value = h->epi_split; return GPX_OK; }
  if (!strcmp(name, "potrf_persist")) { *value = h->potrf_persist; return GPX_OK; }
  if (!strcmp(name, "build_small")) { *value = h->build_small; return GPX_OK; }
  if (!strcmp(name, "split_min_blocks")) { *value = h->split_min_blocks; return GPX_OK; }
  if (!strcmp(name, "host_threads")) { *value = h->host_threads; return GPX_OK; }
  if (!strcmp(name, "resident")) { *value = h->resident; return GPX_OK; }
  gpx_set_error(h, std::string("unknown option ") + name);
  return GPX_ERR_ARG;
}

// ------------------------------------------------------------------ hyper-parameters

// Validate the keyword block against (n rows, d dims) and upload length[d], period[d]
// (and the per-row noise vector when n_noise == n) to the device.
static double kernel_diag_value(int kind, double k_scale);

// Device layout of d_hyp: length[d], period[d], then {k_scale, noise_scalar, k_jitter, prior_diag}
// (prior_diag = k(x*,x*) + new_noise + jitter, the diagonal of k_new, gpax/gp.py:287).
static int upload_params(gpx_handle* h, hipStream_t s, const gpx_kernel_params* p, int64_t n, int d,
                         const double** noise_dev, double* noise_scalar, double new_noise = 0.0) {
  if (!p) { gpx_set_error(h, "params is NULL"); return GPX_ERR_ARG; }
  if (p->kind != GPX_RBF && p->kind != GPX_MATERN52 && p->kind != GPX_PERIODIC) {
    gpx_set_error(h, "unknown kernel kind"); return GPX_ERR_ARG;
  }
  if (!p->k_length || (p->n_length != 1 && p->n_length != d)) {
    gpx_set_error(h, "k_length must have 1 or d entries"); return GPX_ERR_ARG;
  }
  if (p->kind == GPX_PERIODIC && (!p->k_period || (p->n_period != 1 && p->n_period != d))) {
    // the reference class defines no k_period Parameter (gpax/kernels.py:150-168): it must be passed
    gpx_set_error(h, "periodic kernel needs k_period with 1 or d entries"); return GPX_ERR_ARG;
  }
  if (p->n_noise != 0 && p->n_noise != 1 && p->n_noise != n) {
    gpx_set_error(h, "k_noise must have 0, 1 or n entries"); return GPX_ERR_ARG;
  }
  if (p->n_noise > 0 && !p->k_noise) { gpx_set_error(h, "k_noise is NULL"); return GPX_ERR_ARG; }
  std::vector<double> hyp(2 * (size_t)d + 4);
  for (int k = 0; k < d; ++k) {
    hyp[k] = p->k_length[p->n_length == 1 ? 0 : k];
    hyp[d + k] = (p->kind == GPX_PERIODIC) ? p->k_period[p->n_period == 1 ? 0 : k] : 1.0;
  }
  *noise_dev = nullptr;
  *noise_scalar = (p->n_noise == 1) ? p->k_noise[0] : 0.0;
  hyp[2 * d + 0] = p->k_scale;
  hyp[2 * d + 1] = *noise_scalar;
  hyp[2 * d + 2] = p->k_jitter;
  hyp[2 * d + 3] = kernel_diag_value(p->kind, p->k_scale) + (new_noise + p->k_jitter);
  GPX_RET(ensure(h, &h->d_hyp, &h->hyp_cap, 2 * (int64_t)d + 4));
  GPX_RET(upload_small(h, s, h->d_hyp, hyp.data(), (int64_t)hyp.size()));
  if (p->n_noise > 1) {
    GPX_RET(ensure(h, &h->d_noise, &h->noise_cap, n));
    GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_noise, p->k_noise, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    *noise_dev = h->d_noise;
  }
  return GPX_OK;
}

int gpx_upload_params(gpx_handle* h, hipStream_t s, const gpx_kernel_params* p, int64_t n, int d,
                      const double** noise_dev, double* noise_scalar, double new_noise) {
  return upload_params(h, s, p, n, d, noise_dev, noise_scalar, new_noise);
}

// k(x, x) without the diagonal rule: the kernel formulae at r2 = 0 (Matern keeps the 1e-12).
static double kernel_diag_value(int kind, double k_scale) {
  if (kind == GPX_MATERN52) {
    const double a = 2.2360679774997896964 * sqrt(1e-12);
    return k_scale * (1.0 + a) * exp(-a);
  }
  return k_scale;
}

static double gram_flops_per_entry(int kind, int d) {
  // direct differences: 2 flop (sub, fma) per dimension (+ sin ~ 40 for periodic), exp ~ 40, Matern + sqrt
  double f = (kind == GPX_PERIODIC ? 44.0 : 2.0) * d + 42.0;
  if (kind == GPX_MATERN52) f += 30.0;
  return f;
}

// ---------------------------------------------------------------------------- gram

extern "C" int gpx_gram(gpx_handle* h, const gpx_kernel_params* p, const double* x1, int64_t n1,
                        const double* x2, int64_t n2, int32_t d, double* out) {
  if (!h) return GPX_ERR_ARG;
  if (!p || !x1 || !x2 || !out || n1 < 0 || n2 < 0 || d <= 0) { gpx_set_error(h, "gpx_gram: bad argument"); return GPX_ERR_ARG; }
  if (n1 == 0 || n2 == 0) return GPX_OK;  // empty output
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  hipStream_t s = h->s_main;
  const bool diag = p->apply_noise && n1 == n2;  // gpax/kernels.py:124
  const double* noise_dev; double noise_scalar;
  gpx_kernel_params q = *p;
  if (!diag) q.n_noise = 0;  // the noise vector is only meaningful under the diagonal rule
  GPX_RET(upload_params(h, s, &q, n1, d, &noise_dev, &noise_scalar));
  GPX_RET(ensure(h, &h->d_g1, &h->g1_cap, n1 * d));
  GPX_RET(ensure(h, &h->d_g2, &h->g2_cap, n2 * d));
  GPX_RET(ensure(h, &h->d_gout, &h->gout_cap, n1 * n2));
  GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_g1, x1, (size_t)(n1 * d) * sizeof(double), hipMemcpyHostToDevice, s));
  GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_g2, x2, (size_t)(n2 * d) * sizeof(double), hipMemcpyHostToDevice, s));
  GramLaunch g;
  g.kind = p->kind; g.d = d; g.k_scale = p->k_scale; g.hyp = h->d_hyp;
  g.x1 = h->d_g1; g.x2 = h->d_g2; g.n1 = n1; g.n2 = n2; g.out = h->d_gout; g.ld = n2;
  g.lower_only = 0; g.diag_rule = diag ? 1 : 0; g.noise = noise_dev; g.noise_scalar = noise_scalar;
  g.jitter = p->k_jitter;
  prof_begin(h, s, GPX_STAGE_GRAM, gram_flops_per_entry(p->kind, d) * n1 * n2, 8.0 * n1 * n2 + 8.0 * (n1 + n2) * d);
  GPX_RET(launch_gram(h, s, g));
  prof_end(h, s);
  GPX_HIP_CHECK(h, hipMemcpyAsync(out, h->d_gout, (size_t)(n1 * n2) * sizeof(double), hipMemcpyDeviceToHost, s));
  GPX_HIP_CHECK(h, hipStreamSynchronize(s));
  return GPX_OK;
}

// ------------------------------------------------------------------- resident inputs

extern "C" int gpx_set_train(gpx_handle* h, const double* x, int64_t n, int32_t d) {
  if (!h) return GPX_ERR_ARG;
  if (!x || n <= 0 || d <= 0) { gpx_set_error(h, "gpx_set_train: need x (n>0, d>0)"); return GPX_ERR_ARG; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  GPX_RET(ensure(h, &h->d_x, &h->x_cap, n * d));
  GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_x, x, (size_t)(n * d) * sizeof(double), hipMemcpyHostToDevice, h->s_main));
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_main));
  if (h->d != d) { h->m = 0; }
  h->n = n; h->d = d;
  if (h->nb_auto) h->nb = n <= 64 ? 64 : 128;
  h->have_factor = false; h->have_z = false; h->have_r = false;
  return GPX_OK;
}

extern "C" int gpx_set_test(gpx_handle* h, const double* x_new, int64_t m) {
  if (!h) return GPX_ERR_ARG;
  if (h->d <= 0) { gpx_set_error(h, "gpx_set_test: call gpx_set_train first (d unknown)"); return GPX_ERR_STATE; }
  if (m < 0 || (m > 0 && !x_new)) { gpx_set_error(h, "gpx_set_test: bad argument"); return GPX_ERR_ARG; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  if (m > 0) {
    GPX_RET(ensure(h, &h->d_xnew, &h->xnew_cap, m * h->d));
    GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_xnew, x_new, (size_t)(m * h->d) * sizeof(double), hipMemcpyHostToDevice, h->s_main));
    GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_main));
  }
  h->m = m;
  h->have_mnew = false;
  return GPX_OK;
}

extern "C" int gpx_set_targets(gpx_handle* h, const double* y_minus_m, const double* m_new) {
  if (!h) return GPX_ERR_ARG;
  if (h->n <= 0) { gpx_set_error(h, "gpx_set_targets: call gpx_set_train first"); return GPX_ERR_STATE; }
  if (!y_minus_m) { gpx_set_error(h, "gpx_set_targets: y_minus_m is NULL"); return GPX_ERR_ARG; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  // through the pinned ring (asynchronous; the caller's arrays are free again on return) unless they are large
  auto up = [&](double* dst, const double* src, int64_t cnt) -> int {
    if (cnt <= 65536) return upload_small(h, h->s_main, dst, src, cnt);
    GPX_HIP_CHECK(h, hipMemcpyAsync(dst, src, (size_t)cnt * sizeof(double), hipMemcpyHostToDevice, h->s_main));
    GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_main));
    return GPX_OK;
  };
  GPX_RET(ensure(h, &h->d_r, &h->r_cap, h->n));
  GPX_RET(up(h->d_r, y_minus_m, h->n));
  h->have_r = true;
  h->have_mnew = false;
  if (m_new && h->m > 0) {
    GPX_RET(ensure(h, &h->d_mnew, &h->mnew_cap, h->m));
    GPX_RET(up(h->d_mnew, m_new, h->m));
    h->have_mnew = true;
  }
  return GPX_OK;
}

// ------------------------------------------------------------- bordered factorisation

// Events that only order work between the handle's own streams on one device: a device-scope
// release is all they need; the default system-scope fence of an event record costs the NEXT kernel of the
// recording stream ~8 us (measured: 2.24 -> 2.12 ms per N=4096 step with hipEventDisableSystemFence).  Results
// reach the host through stream-ordered copies, which carry their own fences.
static unsigned sync_event_flags() {
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("GPX_EVENT_MODE"); mode = e ? atoi(e) : 2; }
  unsigned f = hipEventDisableTiming;
  if (mode == 1) f |= hipEventReleaseToDevice;
  if (mode == 2) f |= hipEventDisableSystemFence;
  return f;
}

static int ensure_events(gpx_handle* h, size_t nblk) {
  const unsigned fl = sync_event_flags();
  if (!h->ev_join) GPX_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_join, fl));
  if (!h->ev_join2) GPX_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_join2, fl));
  if (!h->ev_chain2) GPX_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_chain2, fl));
  if (!h->ev_build) GPX_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_build, fl));
  GPX_RET(ensure(h, &h->d_flags, &h->flags_cap, 3 * (int64_t)std::max<size_t>(nblk, 64)));   // flagsU, flagsW, flagsS
  GPX_RET(ensure(h, &h->d_mini, &h->mini_cap, 2 * (int64_t)h->nb * h->nb));
  while (h->ev_far.size() < nblk) {
    hipEvent_t a, b;
    GPX_HIP_CHECK(h, hipEventCreateWithFlags(&a, fl));
    GPX_HIP_CHECK(h, hipEventCreateWithFlags(&b, fl));
    h->ev_far.push_back(a);
    h->ev_w.push_back(b);
  }
  while (h->ev_panel.size() < nblk) {
    hipEvent_t a, b;
    GPX_HIP_CHECK(h, hipEventCreateWithFlags(&a, fl));
    GPX_HIP_CHECK(h, hipEventCreateWithFlags(&b, fl));
    h->ev_panel.push_back(a);
    h->ev_trail.push_back(b);
  }
  return GPX_OK;
}

static int factor_loop_split(gpx_handle* h, int64_t rows_total, int64_t tri_base);

// Right-looking blocked Cholesky of the (npad x npad) matrix in d_A with `rows - npad`
// border rows riding along (they are only ever "rows below the panel").  Results in d_L.
//   step k:  potrf_diag  A_kk -> L_kk, W_k = inv(L_kk)            (one workgroup, LDS resident)   stream P
//            panel       L_ik = A_ik W_k^T      for all rows below  (MFMA GEMM)                    stream P
//            near look-ahead  column block k+1 -= panel k           (MFMA GEMM)                    stream P
//            far look-ahead   column blocks k+2 .. k+D              (MFMA GEMM)                    stream Q
//            bulk update      column blocks >= k+D+1                (MFMA GEMM)                    stream T
// P carries the critical path potrf -> panel -> near look-ahead and never waits for the bulk update of
// the SAME step: with depth D = 2 column k+2 is brought up to date by Q one step early, so the only
// thing P waits for is Q's short kernel of the previous step, which in turn had a whole step of slack
// behind the bulk update it depends on.  Every column block has exactly one writer stream at a time
// (T up to panel c-D-1, then Q, then P), each hand-over ordered by an event: no concurrent
// read-modify-write of a tile.  D = 1 is the classic single look-ahead (P waits for the previous bulk
// update); lookahead = 0 runs everything on P.
// tri_base >= 0: the border rows from `tri_base` on are an identity (gpx_lml_grad: they become
// U = L^-T); border block j is still all-zero in the panels k < j, so step k only carries rows
// below tri_base + (k+1)*nb -- the triangular inverse costs N^3/3 flops instead of N^3.
static int factor_loop(gpx_handle* h, int64_t rows_total, int64_t tri_base = -1) {
  // one block column (n <= nb): potrf_diag and the border rows' panel solve, nothing to overlap -- one stream,
  // no events (the multi-stream machinery costs ~20 us of hand-overs on a ~60 us step)
  // two block columns: one stream as well (measured: 0.136 -> 0.115 ms per N = 256 step; from three on the split schedule wins)
  if (h->lookahead != 0 && h->depth >= 3 && h->npad / h->nb >= h->split_min_blocks) return factor_loop_split(h, rows_total, tri_base);
  const int64_t nb = h->nb, npad = h->npad, ld = h->ld;
  const int64_t nblk = npad / nb;
  const bool la = h->lookahead != 0 && nblk >= (h->depth >= 3 ? h->split_min_blocks : 2);
  const int D = la ? (h->depth >= 2 ? 2 : 1) : 1;
  hipStream_t P = h->s_main, T = la ? h->s_trail : h->s_main, Q = (la && D >= 2) ? h->s_far : h->s_main;
  double* A = h->d_A; double* L = h->d_L;
  std::vector<char> trail_pending((size_t)nblk, 0), far_pending((size_t)nblk, 0);

  auto rows_at = [&](int64_t k) -> int64_t {
    return tri_base >= 0 ? std::min<int64_t>(rows_total, tri_base + (k + 1) * nb) : rows_total;
  };
  auto update = [&](hipStream_t st, int64_t k, int64_t cfirst, int64_t ncolblk, bool small) -> int {
    // columns [cfirst, cfirst + ncolblk) blocks, rows from block row cfirst down, -= panel k
    const int64_t rows = rows_at(k);
    const int64_t c0 = k * nb, rr = cfirst * nb, mr = rows - rr, nc = ncolblk * nb;
    GemmLaunch t;
    t.A = L + rr * ld + c0; t.lda = ld;
    t.B = L + rr * ld + c0; t.ldb = ld;
    t.C = A + rr * ld + rr; t.ldc = ld;
    t.M = mr; t.N = nc; t.K = nb; t.alpha = -1.0; t.beta = 1.0;
    t.lower = 1; t.row_off = rr; t.col_off = rr; t.b_lower_tri = 0; t.small_tile = small ? 1 : 0;
    const double entries = (double)nc * (nc + 1) / 2.0 + (double)(mr - nc) * nc;   // lower trapezoid, 2*nb flop each
    prof_begin(h, st, small ? GPX_STAGE_LOOKAHEAD : GPX_STAGE_TRAILING, 2.0 * entries * nb,
               8.0 * (2.0 * entries + (double)(mr + nc) * nb));
    int rc = launch_gemm_nt(h, st, t);
    prof_end(h, st);
    return rc;
  };

  for (int64_t k = 0; k < nblk; ++k) {
    const int64_t c0 = k * nb, r0 = c0 + nb, rows = rows_at(k);
    prof_begin(h, P, GPX_STAGE_POTRF_DIAG, (double)nb * nb * nb / 3.0 * 2.0, 3.0 * 8.0 * nb * nb);
    GPX_RET(launch_potrf_diag(h, P, (int)nb, A + c0 * ld + c0, ld, L + c0 * ld + c0, ld, h->d_W + k * nb * nb,
                              h->d_info, (int)c0));
    prof_end(h, P);
    const int64_t mrows = rows - r0;
    if (mrows <= 0) continue;
    GemmLaunch g;
    g.A = A + r0 * ld + c0; g.lda = ld;
    g.B = h->d_W + k * nb * nb; g.ldb = nb;
    g.C = L + r0 * ld + c0; g.ldc = ld;
    g.M = mrows; g.N = nb; g.K = nb; g.alpha = 1.0; g.beta = 0.0;
    g.lower = 0; g.row_off = 0; g.col_off = 0; g.b_lower_tri = 1; g.small_tile = 1;
    prof_begin(h, P, GPX_STAGE_TRSM_PANEL, (double)mrows * nb * nb, 8.0 * (2.0 * mrows * nb + nb * nb));
    GPX_RET(launch_gemm_nt(h, P, g));
    prof_end(h, P);
    if (k + 1 >= nblk) continue;
    const int64_t nfar = std::min<int64_t>(D - 1, nblk - (k + 2));       // column blocks k+2 .. k+D
    const int64_t nbulk = nblk - (k + 1 + D);                            // column blocks >= k+D+1
    // near look-ahead (P): column k+1.  Its previous writer was Q (depth 2) or T (depth 1) at step k-1.
    if (la && k >= 1) {
      if (D >= 2 && far_pending[(size_t)(k - 1)]) GPX_HIP_CHECK(h, hipStreamWaitEvent(P, h->ev_far[(size_t)(k - 1)], 0));
      if (D == 1 && trail_pending[(size_t)(k - 1)]) GPX_HIP_CHECK(h, hipStreamWaitEvent(P, h->ev_trail[(size_t)(k - 1)], 0));
    }
    GPX_RET(update(P, k, k + 1, 1, true));
    // The other streams are released only now, after the near look-ahead is queued: the next
    // diagonal-block kernel is ONE workgroup that needs a whole CU's LDS, and if the bulk kernel were
    // already resident everywhere it would sit in the dispatcher until that kernel drains (measured:
    // 37 -> 80 us per panel when the bulk update is released right after the panel solve).
    if (la && (nfar > 0 || nbulk > 0)) GPX_HIP_CHECK(h, hipEventRecord(h->ev_panel[(size_t)k], P));

    // far look-ahead (Q): columns k+2 .. k+D; column k+D was last written by the bulk update of step k-1
    if (nfar > 0) {
      if (la) {
        GPX_HIP_CHECK(h, hipStreamWaitEvent(Q, h->ev_panel[(size_t)k], 0));
        if (k >= 1 && trail_pending[(size_t)(k - 1)]) GPX_HIP_CHECK(h, hipStreamWaitEvent(Q, h->ev_trail[(size_t)(k - 1)], 0));
      }
      GPX_RET(update(Q, k, k + 2, nfar, true));
      if (la) { GPX_HIP_CHECK(h, hipEventRecord(h->ev_far[(size_t)k], Q)); far_pending[(size_t)k] = 1; }
    }
    // bulk update (T): everything right of the look-ahead columns
    if (nbulk > 0) {
      if (la) GPX_HIP_CHECK(h, hipStreamWaitEvent(T, h->ev_panel[(size_t)k], 0));
      GPX_RET(update(T, k, k + 1 + D, nbulk, false));
      if (la) { GPX_HIP_CHECK(h, hipEventRecord(h->ev_trail[(size_t)k], T)); trail_pending[(size_t)k] = 1; }
    }
  }
  if (la) {   // join: everything on T and Q is done before anything later on P (epilogue, copies)
    GPX_HIP_CHECK(h, hipEventRecord(h->ev_join, T));
    GPX_HIP_CHECK(h, hipStreamWaitEvent(P, h->ev_join, 0));
    if (Q != P) {
      GPX_HIP_CHECK(h, hipEventRecord(h->ev_join2, Q));
      GPX_HIP_CHECK(h, hipStreamWaitEvent(P, h->ev_join2, 0));
    }
  }
  return GPX_OK;
}

// Split-panel schedule (depth 3).  The chain stream P carries ONLY what the next diagonal block needs:
//   P:  potrf(k) -> S = A_{k+1,k} W_k^T (one block row of the panel, 16 workgroups, into scratch)
//                -> A_{k+1,k+1} -= S S^T (10 workgroups) -> potrf(k+1) ...
//   Q:  the full panel k (all rows, incl. block row k+1 again: the official L) -> update of columns k+1
//       (below its diagonal block) and k+2 by panel k
//   T:  bulk update of the columns >= k+3
// so the big panel GEMM (which runs 2-3x slower while the bulk kernel owns the CUs) and one
// cross-stream hand-over leave the critical path: per block column it is potrf + two ~6 us kernels.
// Hand-overs (each an event): W_k P->Q; panel k Q->T; columns k+1,k+2 Q->P (needed by S of step k+1:
// a whole diagonal block later); column k+2 T->Q (bulk of step k-1).  One writer per block at a time:
// block (k+1,k+1) belongs to P at step k (Q's column-k+1 update starts below it).
static int factor_loop_split(gpx_handle* h, int64_t rows_total, int64_t tri_base) {
  const int64_t nb = h->nb, npad = h->npad, ld = h->ld, nblk = npad / nb;
  hipStream_t P = h->s_main, Q = h->s_far, T = h->s_trail;
  double* A = h->d_A; double* L = h->d_L;
  auto rows_at = [&](int64_t k) -> int64_t {
    return tri_base >= 0 ? std::min<int64_t>(rows_total, tri_base + (k + 1) * nb) : rows_total;
  };
  // Grouped bulk updates while the factorisation is bulk-bound: T applies the g most recent panels in ONE
  // K = g nb pass every g-th step, cutting the read-modify-write passes over the trailing matrix g-fold (the bulk
  // kernel spends about as long on a tile's C traffic as on 128 columns of MFMA work; N = 16384: 42.0 ms with
  // g = 1, 36.4 with g = 2).  g is chosen per group from the block columns still to go: 4 above `quad_min`,
  // 2 above `pair_min`, else 1 -- in the chain-bound tail a g-fold kernel would sit one step before its
  // consumer (N = 4096 with g = 2 throughout: 1.99 -> 2.11 ms).  Invariant after T's step j: every column
  // >= j+3 is current through panel j; Q brings column k+2 up to date with the panels since then, so a change
  // of g at a group boundary needs no fix-up.
  // Q and T fork from P once up front (inside a stream capture a stream must join from the origin stream
  // first; joining T through an event recorded on Q crashed hipStreamEndCapture on ROCm 7.2)
  // Resident start of potrf_diag: consecutive diagonal blocks alternate between two chain streams, so
  // potrf_diag(k+1) is dispatched while block k is still being worked on, owns its CU early and sleeps on the
  // tile counter that the A -= S S^T kernel of step k publishes (potrf.hip).  Behind a saturating bulk update a
  // freshly launched potrf_diag waits 20-30 us for a CU to drain (it needs all of its LDS): this removes that
  // wait from the first ~11 block columns and the launch gap from all of them.
  const bool resident = h->resident != 0 && !h->in_capture && h->prof_mode != 1 && nblk >= 3;
  hipStream_t P2 = resident ? h->s_main2 : P;
  const unsigned n_signal = (unsigned)((nb / 32) * (nb / 32 + 1) / 2);   // lower 32x32 tiles of a diagonal block
  if (resident && !h->flags_clean) GPX_HIP_CHECK(h, hipMemsetAsync(h->d_flags, 0, (size_t)(3 * nblk) * sizeof(unsigned), P));
  h->flags_clean = false;
  // resident == 2: every hand-over of the chain is in-kernel.  The diagonal blocks run on one stream, the two band
  // kernels on the other; S = A W_k^T is launched while potrf_diag(k) still runs and sleeps on the flag that kernel
  // raises after its last (write-through) store of W_k -- the ~11 us from potrf_diag's last instruction to the
  // start of the next kernel of its stream (store drain, kernel-end release, completion signal, launch) shrink to
  // a flag round trip.
  const bool chain2 = resident && h->resident >= 2;
  unsigned* flagsU = h->d_flags;            // [k]: tiles of A_{k+1,k+1} -= S S^T published (potrf_diag(k+1) waits)
  unsigned* flagsW = h->d_flags + nblk;     // [k]: W_k published (S = A W_k^T of step k waits)
  unsigned* flagsS = h->d_flags + 2 * nblk; // [k]: tiles of S published (fused band kernel: its second phase waits)
  const bool fuse_band = chain2 && h->fuse_band != 0;
  const bool persist = chain2 && h->potrf_persist != 0 && h->kinv_U == nullptr && tri_base < 0;   // ONE potrf_diag launch for all blocks
  GPX_HIP_CHECK(h, hipEventRecord(h->ev_join, P));
  GPX_HIP_CHECK(h, hipStreamWaitEvent(Q, h->ev_join, 0));
  GPX_HIP_CHECK(h, hipStreamWaitEvent(T, h->ev_join, 0));
  if (resident) GPX_HIP_CHECK(h, hipStreamWaitEvent(P2, h->ev_join, 0));
  // gradient path: K^-1 is accumulated on T while the chain runs (see kinv_update below); T is idle until the
  // first panel is complete, so clearing the accumulator there costs nothing
  const bool kinv = h->kinv_U != nullptr;
  if (kinv) GPX_HIP_CHECK(h, hipMemsetAsync(h->d_kinv, 0, (size_t)(h->kinv_rows * npad) * sizeof(double), T));

  // Host-side hand-shake between the two enqueueing threads: an event may only be waited for after the call
  // that records it has been ISSUED (a wait on a not-yet-recorded event is a no-op).
  const bool threaded = h->host_threads >= 2 && !h->prof_mode && !h->in_capture && nblk >= 4;
  std::unique_ptr<std::atomic<int>[]> w_rec(new std::atomic<int>[(size_t)nblk]), far_rec(new std::atomic<int>[(size_t)nblk]);
  for (int64_t k = 0; k < nblk; ++k) { w_rec[(size_t)k].store(0); far_rec[(size_t)k].store(0); }
  std::atomic<int> abort_flag{0};
  auto await = [&](std::atomic<int>& f) -> bool {   // false: the other side failed
    if (!threaded) return true;
    while (f.load(std::memory_order_acquire) == 0)
      if (abort_flag.load(std::memory_order_acquire)) return false;
    return true;
  };

  // completion events ride on the kernels' own dispatch packets (hipExtLaunchKernelGGL stopEvent) except inside a
  // stream capture and under per-launch profiling, where they are recorded behind the kernel as usual
  const bool attach = !h->in_capture && h->prof_mode != 1;

  // ---- "diagonal block in the band kernel" (fused band kernel, ungrouped steps).  The column update of step j used to
  // cover column j+2 from its diagonal block down, behind the wait for the bulk launch that last wrote that column, and
  // the band kernel of step j+1 waited for all of it -- on alternate block columns of the chain-bound middle of the
  // factorisation that update (slowed 2x by the bulk launch it shares the CUs with) finished 10-15 us after
  // potrf_diag(j+1) (profiles/r01_step_timeline_mid.txt).  Now the panel stream's update of column j+2 starts BELOW
  // the diagonal block, the hand-over event is recorded behind the update of column j+1 alone, and the band kernel
  // of step j+1 applies panel j to block (j+2,j+2) itself, in a phase 0 that runs while it waits, resident, for
  // potrf_diag(j+1) -- same tile kernel, same order of the two subtractions, bit-identical result.
  // dib[k]: the band kernel of step k has that phase 0 (for panel k-1 on block (k+1,k+1)); band_t[k]: the bulk launch
  // (its step) that last wrote that block and must have finished, or -1.  Both follow from k alone (the grouping is a
  // function of k), so the chain and the side streams agree without talking.
  const bool use_dib = fuse_band && h->diag_band != 0 && !kinv;
  std::vector<char> dib((size_t)nblk + 2, 0);
  std::vector<int64_t> band_t((size_t)nblk + 2, -1);
  // lp_flag: the last border panel (chain stream) is ordered behind the panel stream's last column update by a device
  // counter that update's tiles raise (flagsS[nblk-1], otherwise unused) instead of a stream-level event wait in front
  // of the last potrf_diag, which kept that kernel from starting resident (+17 us at the tail of every step)
  bool lp_flag = false;
  if (chain2 && !kinv) {
    const int64_t pm = h->pair_min, qm = h->quad_min;
    auto gsz = [&](int64_t left) -> int64_t { return (qm > 0 && left > qm) ? 4 : ((pm > 0 && left > pm) ? 2 : 1); };
    int64_t lt = -1, ge = -1, ti = -1, tf = nblk;
    for (int64_t j = 0; j < nblk; ++j) {
      if (j > ge) ge = j + gsz(nblk - j) - 1;
      const int64_t rows = rows_at(j);
      if (rows - (j + 1) * nb <= 0 || j + 1 >= nblk) continue;
      int64_t near = 0;
      if (j == ge && h->near_block) {
        const int64_t gn = gsz(nblk - (j + 1));
        if (gn >= 2) near = std::max<int64_t>(0, std::min<int64_t>(gn, nblk - (j + 3)));
      }
      const int64_t mr = rows - (j + 2) * nb, ncb = std::min<int64_t>(2, nblk - (j + 1));
      if (h->lp_flag && j == nblk - 2 && j - lt == 1 && mr > 0 && (h->la_rows <= 0 || mr < h->la_rows) && (nb == 128 || nb == 64)) lp_flag = true;
      if (use_dib && mr > 0 && ncb > 1 && j - lt == 1) {
        dib[(size_t)(j + 1)] = 1;
        band_t[(size_t)(j + 1)] = (ti >= 0 && j + 2 >= tf) ? ti : -1;
      }
      if (j == ge) {
        if (nblk - (j + 3 + near) > 0) { ti = j; tf = j + 3 + near; }
        lt = j;
      }
    }
  }
  std::vector<hipEvent_t> tev((size_t)nblk, nullptr);   // hand-over event of the bulk launch of step k (set when it is launched)
  std::unique_ptr<std::atomic<int>[]> trail_rec(new std::atomic<int>[(size_t)nblk]);
  for (int64_t k = 0; k < nblk; ++k) trail_rec[(size_t)k].store(0);

  // ---- the chain (stream P) at step k: potrf, then the diagonal band
  auto chain = [&](int64_t k) -> int {
    // resident == 1: consecutive steps alternate between the two chain streams; resident == 2: potrf_diag on
    // s_main, the band kernels on P2 (these shadow the outer P)
    hipStream_t P = chain2 ? h->s_main : ((resident && (k & 1)) ? P2 : h->s_main);
    hipStream_t PB = chain2 ? P2 : P;   // stream of the two band kernels
    double* S = h->d_mini + (k & 1) * nb * nb;
    const int64_t c0 = k * nb, r0 = c0 + nb, rows = rows_at(k);
    // ev_w[k] = completion of potrf_diag(k): attached to the kernel's own dispatch when possible (no barrier
    // packet behind it), else recorded right after it -- NOT after the two band kernels (measured 1.91 -> 2.12 ms
    // per N = 4096 step when the panel stream learns of W_k 12 us later)
    const bool need_w = rows - r0 > 0;
    if (need_w && k + 1 >= nblk && !kinv && k >= 1 && !lp_flag) {
      // The last panel (below) reads the border rows of column k on THIS stream; their last writer is Q's column update
      // of step k-1.  Nothing else orders the two (potrf_diag(k) only waits for the band kernel of step k-1, which
      // waited for Q's step k-2): wait for it here, in front of potrf_diag, where it costs nothing -- that update is
      // normally long finished.  (Without it the race was real at two block columns once the border was built on
      // the panel stream: one wrong posterior in ~200 runs of the test suite.)
#ifndef GPX_NO_LASTPANEL_WAIT   // (diagnostic switch: tools/soak.py reproduces the race without the wait)
      if (!await(far_rec[(size_t)(k - 1)])) return GPX_ERR_HIP;
      GPX_HIP_CHECK(h, hipStreamWaitEvent(P, h->ev_far[(size_t)(k - 1)], 0));
#endif
    }
    if (persist) {
      // every diagonal block in ONE launch, issued with the first step: the workgroup keeps its CU (potrf.hip)
      if (k == 0) GPX_RET(launch_potrf_chain(h, P, (int)nb, A, ld, L, ld, h->d_W, h->d_info, (int)nblk, flagsU, n_signal, flagsW));
    } else {
      prof_begin(h, P, GPX_STAGE_POTRF_DIAG, (double)nb * nb * nb / 3.0 * 2.0, 3.0 * 8.0 * nb * nb);
      GPX_RET(launch_potrf_diag(h, P, (int)nb, A + c0 * ld + c0, ld, L + c0 * ld + c0, ld, h->d_W + k * nb * nb,
                                h->d_info, (int)c0, (need_w && attach) ? h->ev_w[(size_t)k] : nullptr,
                                (resident && k >= 1) ? flagsU + (k - 1) : nullptr, n_signal,
                                (chain2 && need_w && k + 1 < nblk) ? flagsW + k : nullptr));
      prof_end(h, P);
    }
    if (!need_w) return GPX_OK;
    if (k + 1 >= nblk && !kinv) {
      // the last panel is the border rows only and nothing follows it but the epilogue: it runs on the chain stream,
      // behind potrf_diag, instead of hopping to the panel stream and back (two ~8 us event hand-overs)
      GemmLaunch g;
      g.A = A + r0 * ld + c0; g.lda = ld;
      g.B = h->d_W + k * nb * nb; g.ldb = nb;
      g.C = L + r0 * ld + c0; g.ldc = ld;
      g.M = rows - r0; g.N = nb; g.K = nb; g.alpha = 1.0; g.beta = 0.0;
      g.lower = 0; g.row_off = 0; g.col_off = 0; g.b_lower_tri = 1; g.small_tile = 1;
      if (lp_flag && k >= 1) {   // its input rows were last written by the panel stream's column update of step k-1: wait for that update's tiles
        g.wait_flag = flagsS + k; g.wait_count = (unsigned)((nb / 32) * ((rows - r0) / 32)); g.wait_info = h->d_info;
      }
      prof_begin(h, P, GPX_STAGE_TRSM_PANEL, (double)g.M * nb * nb, 8.0 * (2.0 * g.M * nb + nb * nb));
      GPX_RET(launch_gemm_nt(h, P, g));
      prof_end(h, P);
      return GPX_OK;   // (the gradient path keeps its last panel on Q: the K^-1 update behind it belongs to T's sequence)
    }
    if (!attach && !persist) GPX_HIP_CHECK(h, hipEventRecord(h->ev_w[(size_t)k], P));
    w_rec[(size_t)k].store(1, std::memory_order_release);
    if (k + 1 >= nblk) return GPX_OK;
    if (k >= 1) {   // columns k, k+1 were brought up to date by Q at step k-1
      if (!await(far_rec[(size_t)(k - 1)])) return GPX_ERR_HIP;
      GPX_HIP_CHECK(h, hipStreamWaitEvent(PB, h->ev_far[(size_t)(k - 1)], 0));
    }
    GemmLaunch g;
    g.A = A + r0 * ld + c0; g.lda = ld;
    g.B = h->d_W + k * nb * nb; g.ldb = nb;
    g.C = S; g.ldc = nb;
    g.M = nb; g.N = nb; g.K = nb; g.alpha = 1.0; g.beta = 0.0;
    g.lower = 0; g.row_off = 0; g.col_off = 0; g.b_lower_tri = 1; g.small_tile = 1;
    if (chain2) { g.wait_flag = flagsW + k; g.wait_count = 1; g.wait_info = h->d_info; }   // W_k arrives in-kernel from potrf_diag(k)
    GemmLaunch u;
    u.A = S; u.lda = nb; u.B = S; u.ldb = nb;
    u.C = A + r0 * ld + r0; u.ldc = ld;
    u.M = nb; u.N = nb; u.K = nb; u.alpha = -1.0; u.beta = 1.0;
    u.lower = 1; u.row_off = 0; u.col_off = 0; u.b_lower_tri = 0; u.small_tile = 1;
    if (resident) u.signal = flagsU + k;   // potrf_diag(k+1) is already resident and polls this counter
    if (fuse_band) {   // both band kernels in one launch: S's tiles are handed over through flagsS[k]
      g.signal = flagsS + k;
      u.wait_flag = flagsS + k; u.wait_count = (unsigned)((nb / 32) * (nb / 32)); u.wait_info = h->d_info;
      GemmLaunch z;   // phase 0: block (k+1,k+1) -= L_{k+1,k-1} L_{k+1,k-1}^T (panel k-1), see dib[] above
      const bool phase0 = dib[(size_t)k] != 0;
      if (phase0) {
        if (band_t[(size_t)k] >= 0) {   // the bulk launch that last wrote the block
          if (!await(trail_rec[(size_t)band_t[(size_t)k]])) return GPX_ERR_HIP;
          GPX_HIP_CHECK(h, hipStreamWaitEvent(PB, tev[(size_t)band_t[(size_t)k]], 0));
        }
        z.A = L + r0 * ld + (k - 1) * nb; z.lda = ld;
        z.B = z.A; z.ldb = ld;
        z.C = A + r0 * ld + r0; z.ldc = ld;
        z.M = nb; z.N = nb; z.K = nb; z.alpha = -1.0; z.beta = 1.0;
        z.lower = 1; z.row_off = 0; z.col_off = 0; z.b_lower_tri = 0; z.small_tile = 1;
      }
      prof_begin(h, PB, GPX_STAGE_LOOKAHEAD, (phase0 ? 3.0 : 2.0) * (double)nb * nb * nb, 8.0 * 6.0 * nb * nb);
      GPX_RET(launch_chain_band(h, PB, phase0 ? &z : nullptr, g, u));
      prof_end(h, PB);
      return GPX_OK;
    }
    prof_begin(h, PB, GPX_STAGE_LOOKAHEAD, (double)nb * nb * nb, 8.0 * 3.0 * nb * nb);
    GPX_RET(launch_gemm_nt(h, PB, g));
    prof_end(h, PB);
    prof_begin(h, PB, GPX_STAGE_LOOKAHEAD, (double)nb * nb * nb, 8.0 * 3.0 * nb * nb);
    GPX_RET(launch_gemm_nt(h, PB, u));
    prof_end(h, PB);
    return GPX_OK;
  };

  // ---- the side streams at step k: Q (whole panel, columns k+1 and k+2), T (bulk)
  // the gradient path doubles the bulk stream's work (identity border rows, K^-1 accumulation): it is bound by
  // that stream almost to the end, so its bulk updates are grouped down to 8 remaining block columns
  const int64_t pair_min = kinv ? std::min<int64_t>(h->pair_min, 8) : h->pair_min, quad_min = h->quad_min;
  int64_t last_t = -1;        // last group end (panels <= last_t are in every column >= last_t+3: by Q's near block and T's bulk launch)
  int64_t group_end = -1;     // the step at which T acts next
  int64_t q_waited_t = -1;    // the T launch (its step) Q has already waited for
  int64_t t_idx = -1;         // step of the most recent bulk launch on T (its hand-over event is ev_trail[t_idx])
  int64_t t_from = nblk;      // first block column that launch wrote: Q waits for it only before touching a column >= t_from
  auto group_size = [&](int64_t left) -> int64_t {
    return (quad_min > 0 && left > quad_min) ? 4 : ((pair_min > 0 && left > pair_min) ? 2 : 1);
  };
  int64_t kinv_last = -1;     // last panel folded into K^-1 (gradient path)
  static int kinv_group_env = -1;
  if (kinv_group_env < 0) { const char* e = getenv("GPX_KINV_GROUP"); kinv_group_env = e ? atoi(e) : 2; }
  const int64_t kinv_group = std::max(kinv_group_env, 1);
  std::vector<char> t_pending((size_t)nblk, 0);
  auto side = [&](int64_t k) -> int {
    if (k > group_end) {
      group_end = k + group_size(nblk - k) - 1;
    }
    const int64_t c0 = k * nb, r0 = c0 + nb, rows = rows_at(k);
    const int64_t mrows = rows - r0;
    if (mrows <= 0) return GPX_OK;
    if (k + 1 >= nblk && !kinv) return GPX_OK;   // the last panel (border rows only) is launched by the chain itself
    // Grouped phases: the bulk launch at a group end leaves the NEXT group's columns (the "near block", as many block
    // columns as that group has panels) to Q, which applies the same panels to them in one K = g nb launch of its own.
    // The chain of the next group then depends on that short launch only and runs WHILE the bulk launch does; with
    // the bulk launch covering everything from column k+3 the chain stalled at the first step of every group until
    // the bulk launch had finished and the bulk stream then idled for the 2-4 chain steps that produce its next
    // panels (N = 16384: ~270 us of every ~2.3 ms).
    int64_t near_cols = 0;
    if (k == group_end && h->near_block && !kinv) {
      const int64_t g_next = group_size(nblk - (k + 1));
      if (g_next >= 2) near_cols = std::max<int64_t>(0, std::min<int64_t>(g_next, nblk - (k + 3)));
    }
    if (!await(w_rec[(size_t)k])) return GPX_ERR_HIP;
    // (Starting the panel kernel resident too -- it would sleep on potrf_diag's flag instead of waiting ~8 us for the
    // cross-stream event -- crowds the bulk update with sleeping workgroups: 1.76 -> 1.81 ms, not used.)
    if (persist) GPX_RET(launch_flag_relay(h, Q, flagsW + k, 1u, h->d_info));   // W_k: the one potrf_diag launch raises flagsW[k]
    else GPX_HIP_CHECK(h, hipStreamWaitEvent(Q, h->ev_w[(size_t)k], 0));
    {
      GemmLaunch g;
      g.A = A + r0 * ld + c0; g.lda = ld;
      g.B = h->d_W + k * nb * nb; g.ldb = nb;
      g.C = L + r0 * ld + c0; g.ldc = ld;
      g.M = mrows; g.N = nb; g.K = nb; g.alpha = 1.0; g.beta = 0.0;
      g.lower = 0; g.row_off = 0; g.col_off = 0; g.b_lower_tri = 1; g.small_tile = 1;
      // T gets its own event on the panel: the bulk update must not wait for the column kernels below (with one
      // shared event after them the N = 4096 step went from 1.90 to 2.17 ms: every bulk update starts late)
      const bool t_needs = (k + 1 < nblk && k == group_end && nblk - (k + 3 + near_cols) > 0) || (kinv && (k == group_end || k + 1 >= nblk));
      if (t_needs && attach) g.stop_event = h->ev_panel[(size_t)k];
      prof_begin(h, Q, GPX_STAGE_TRSM_PANEL, (double)mrows * nb * nb, 8.0 * (2.0 * mrows * nb + nb * nb));
      GPX_RET(launch_gemm_nt(h, Q, g));
      prof_end(h, Q);
      if (t_needs && !attach) GPX_HIP_CHECK(h, hipEventRecord(h->ev_panel[(size_t)k], Q));
      // every block column but the last is final now (V^T, z and the diagonal of L): their share of the epilogue's
      // reductions runs here, on the panel stream, beside the last potrf_diag instead of behind the last panel
      if (h->epi && k == nblk - 2 && nblk >= 3 && h->d_epart) {
        GPX_RET(launch_epilogue_partial(h, Q, *h->epi, (nblk - 1) * nb, h->d_epart));
        h->epi_split_col = (nblk - 1) * nb;
      }
    }
    // K^-1 += U_{:,j} U_{:,j}^T for the panels j = last_t+1 .. k that are complete now (gradient path): rank-(npan nb)
    // update of the leading (k+1) nb square, lower tiles, on T.  Small while the bulk updates are big and growing as
    // they shrink, it fills the chain-bound second half of the factorisation with the N^3/3 flops that used to
    // follow it as one U U^T product (2.9 -> 2.2 ms per gradient call at N = 4096).
    auto kinv_update = [&]() -> int {
      const int64_t kfirst = kinv_last + 1, npan = k - kinv_last;
      kinv_last = k;
      const int64_t msq = std::min<int64_t>(h->kinv_rows, (k + 1) * nb);
      GemmLaunch t;
      t.A = h->kinv_U + kfirst * nb; t.lda = ld;
      t.B = h->kinv_U + kfirst * nb; t.ldb = ld;
      t.C = h->d_kinv; t.ldc = npad;
      t.M = msq; t.N = msq; t.K = npan * nb; t.alpha = 1.0; t.beta = 1.0;
      t.lower = 1; t.row_off = 0; t.col_off = 0; t.b_lower_tri = 0; t.small_tile = 0;
      const double entries = (double)msq * (msq + 1) / 2.0;
      prof_begin(h, T, GPX_STAGE_TRAILING, 2.0 * entries * t.K, 8.0 * (2.0 * entries + 2.0 * (double)msq * t.K));
      int rc = launch_gemm_nt(h, T, t);
      prof_end(h, T);
      return rc;
    };
    if (k + 1 >= nblk) {   // border rows only: nothing left to update
      if (kinv) {
        GPX_HIP_CHECK(h, hipStreamWaitEvent(T, h->ev_panel[(size_t)k], 0));
        GPX_RET(kinv_update());
      }
      return GPX_OK;
    }

    // Q keeps two columns current ahead of the chain: column k+1 (below its diagonal block) gets panel k, column
    // k+2 every panel since T's last step; rows from block row k+2 down.  (A third window column, which takes T
    // off everything P waits for, was tried with and without grouped updates: 1.94-2.02 vs 1.88 ms at N = 4096.)
    const int64_t rr = (k + 2) * nb, mr = rows - rr;
    bool far_attached = false;
    auto col_update = [&](int64_t cfirst, int64_t ncol, int64_t kfirst, int64_t npan, bool last, int64_t skip_blocks = 0) -> int {
      GemmLaunch t;   // columns [cfirst, cfirst+ncol) blocks, rows >= rr (+ skip_blocks block rows), -= panels [kfirst, kfirst+npan)
      const int64_t rr = (k + 2 + skip_blocks) * nb, mr = rows - rr;
      if (mr <= 0) return GPX_OK;
      if (last && attach) { t.stop_event = h->ev_far[(size_t)k]; far_attached = true; }
      t.A = L + rr * ld + kfirst * nb; t.lda = ld;
      t.B = L + cfirst * nb * ld + kfirst * nb; t.ldb = ld;
      t.C = A + rr * ld + cfirst * nb; t.ldc = ld;
      t.M = mr; t.N = ncol * nb; t.K = npan * nb; t.alpha = -1.0; t.beta = 1.0;
      t.lower = 1; t.row_off = rr; t.col_off = cfirst * nb; t.b_lower_tri = 0;
      if (lp_flag && k == nblk - 2 && cfirst == nblk - 1) t.signal = flagsS + (nblk - 1);   // the last border panel waits for these tiles (chain stream)
      // tall columns (the bulk-bound phase of a large factorisation: no hurry, the machine is full) on 64x64 tiles,
      // which move half the operand bytes per flop; the one-pass 32x32 tiles are for latency
      t.small_tile = (h->la_rows > 0 && mr >= h->la_rows) ? 0 : 1;
      t.role = 1;
      double entries = 0.0;
      for (int64_t c = cfirst; c < cfirst + ncol; ++c)
        entries += (c * nb >= rr) ? (double)nb * (nb + 1) / 2.0 + (double)(mr - nb) * nb : (double)mr * nb;
      prof_begin(h, Q, GPX_STAGE_LOOKAHEAD, 2.0 * entries * t.K, 8.0 * (2.0 * entries + (double)(mr + ncol * nb) * t.K));
      int rc = launch_gemm_nt(h, Q, t);
      prof_end(h, Q);
      return rc;
    };
    const int64_t ncb = std::min<int64_t>(2, nblk - (k + 1));
    if (mr > 0) {
      auto wait_for_t = [&]() -> int {   // column k+2 was last written by the bulk launch of step t_idx -- if it lies in its range
        if (t_idx >= 0 && k + 2 >= t_from && t_idx != q_waited_t && t_pending[(size_t)t_idx]) {
          GPX_HIP_CHECK(h, hipStreamWaitEvent(Q, tev[(size_t)t_idx], 0));
          q_waited_t = t_idx;
        }
        return GPX_OK;
      };
      const int64_t npan2 = k - last_t;   // panels column k+2 still lacks: last_t+1 .. k
      // While the factorisation is bulk-bound the two columns go separately: column k+1 (whose last bulk update
      // Q waited for a step ago) needs no bulk update to finish, only column k+2 does.  As one two-column kernel
      // behind the wait, the update -- slowed 3-4x by the bulk kernel it shares the CUs with -- and the next panel
      // behind it took longer than one bulk update, and the bulk stream idled ~18 us per step at N = 4096.
      const bool split = ncb > 1 && (npan2 > 1 || (h->col_split > 0 && nblk - (k + 3) >= h->col_split));
      if (dib[(size_t)(k + 1)]) {
        // column k+1 alone carries the hand-over to the chain; column k+2 from BELOW its diagonal block, behind the
        // wait for the bulk launch (the band kernel of step k+1 applies panel k to that block itself)
        GPX_RET(col_update(k + 1, 1, k, 1, true));
        if (!far_attached) GPX_HIP_CHECK(h, hipEventRecord(h->ev_far[(size_t)k], Q));
        far_attached = true;   // recorded (or attached) already: not again behind the second column
        far_rec[(size_t)k].store(1, std::memory_order_release);
        GPX_RET(wait_for_t());
        GPX_RET(col_update(k + 2, 1, k, 1, false, 1));
      } else if (split) {
        GPX_RET(col_update(k + 1, 1, k, 1, false));
        GPX_RET(wait_for_t());
        GPX_RET(col_update(k + 2, 1, last_t + 1, npan2, true));
      } else {
        GPX_RET(wait_for_t());
        GPX_RET(col_update(k + 1, ncb, k, 1, true));
      }
    }
    if (!far_attached) GPX_HIP_CHECK(h, hipEventRecord(h->ev_far[(size_t)k], Q));
    far_rec[(size_t)k].store(1, std::memory_order_release);

    // Q: the near block (columns k+3 .. k+2+near_cols, panels last_t+1 .. k), behind the event the chain waits for
    if (near_cols > 0) {
      if (t_idx >= 0 && t_idx != q_waited_t && t_pending[(size_t)t_idx]) {   // those columns were last written by the previous bulk launch
        GPX_HIP_CHECK(h, hipStreamWaitEvent(Q, tev[(size_t)t_idx], 0));
        q_waited_t = t_idx;
      }
      const int64_t kfirst = last_t + 1, npan = k - last_t;
      const int64_t br = (k + 3) * nb, bm = rows - br, nc = near_cols * nb;
      GemmLaunch t;
      t.A = L + br * ld + kfirst * nb; t.lda = ld;
      t.B = L + br * ld + kfirst * nb; t.ldb = ld;
      t.C = A + br * ld + br; t.ldc = ld;
      t.M = bm; t.N = nc; t.K = npan * nb; t.alpha = -1.0; t.beta = 1.0;
      t.lower = 1; t.row_off = br; t.col_off = br; t.b_lower_tri = 0; t.small_tile = 0; t.role = 1;
      const double entries = (double)nc * (nc + 1) / 2.0 + (double)(bm - nc) * nc;
      prof_begin(h, Q, GPX_STAGE_LOOKAHEAD, 2.0 * entries * t.K, 8.0 * (2.0 * entries + (double)(bm + nc) * t.K));
      GPX_RET(launch_gemm_nt(h, Q, t));
      prof_end(h, Q);
    }
    // T: bulk, columns >= k+3 (+ the near block), panels last_t+1 .. k in one pass
    const int64_t nbulk = nblk - (k + 3 + near_cols);
    if (k == group_end) {
      if (nbulk > 0) {
        GPX_HIP_CHECK(h, hipStreamWaitEvent(T, h->ev_panel[(size_t)k], 0));
        const int64_t kfirst = last_t + 1, npan = k - last_t;
        const int64_t cf = k + 3 + near_cols, br = cf * nb, bm = rows - br, nc = nbulk * nb;
        GemmLaunch t;
        t.A = L + br * ld + kfirst * nb; t.lda = ld;
        t.B = L + br * ld + kfirst * nb; t.ldb = ld;
        t.C = A + br * ld + br; t.ldc = ld;
        t.M = bm; t.N = nc; t.K = npan * nb; t.alpha = -1.0; t.beta = 1.0;
        t.lower = 1; t.row_off = br; t.col_off = br; t.b_lower_tri = 0; t.small_tile = 0;
        const double entries = (double)nc * (nc + 1) / 2.0 + (double)(bm - nc) * nc;
        // profiled: the kernel's own dispatch timestamps (what rocprofv3 reports), the hand-over event recorded behind it
        const bool timed = prof_pair(h, GPX_STAGE_TRAILING, 2.0 * entries * t.K, 8.0 * (2.0 * entries + (double)(bm + nc) * t.K),
                                     &t.start_event, &t.stop_event);
        // timed (profile mode 3): the timing stop event, attached to the kernel's own dispatch, IS the hand-over event -- a
        // second record behind the launch delayed every consumer of the bulk update and stretched the measured step by 10 %
        if (attach && !timed) t.stop_event = h->ev_trail[(size_t)k];
        tev[(size_t)k] = (attach && timed) ? t.stop_event : h->ev_trail[(size_t)k];
        GPX_RET(launch_gemm_nt(h, T, t));
        if (!attach) GPX_HIP_CHECK(h, hipEventRecord(h->ev_trail[(size_t)k], T));
        t_pending[(size_t)k] = 1;
        trail_rec[(size_t)k].store(1, std::memory_order_release);
        t_idx = k; t_from = cf;
      }
      // behind the bulk update (Q's next column update waits for that one, not for this), two panels at a time:
      // nothing waits for K^-1 before the end and K = 2 nb passes run at 48 instead of 40 TFLOP/s (groups of 4 or 8
      // leave a long last pass exposed behind the chain: 2.58 / 2.68 / 2.85 ms per call at N = 4096)
      if (kinv && (k - kinv_last >= kinv_group || k + 2 >= nblk)) {
        if (nbulk <= 0) GPX_HIP_CHECK(h, hipStreamWaitEvent(T, h->ev_panel[(size_t)k], 0));
        GPX_RET(kinv_update());
      }
      last_t = k;
    }
    return GPX_OK;
  };

  int rc = GPX_OK;
  if (threaded) {
    int side_rc = GPX_OK;
    h->worker.start(h->device);
    h->worker.post([&]() {
      for (int64_t k = 0; k < nblk && side_rc == GPX_OK; ++k) side_rc = side(k);
      if (side_rc != GPX_OK) abort_flag.store(1, std::memory_order_release);
    });
    for (int64_t k = 0; k < nblk && rc == GPX_OK; ++k) rc = chain(k);
    if (rc != GPX_OK) abort_flag.store(1, std::memory_order_release);
    h->worker.wait();
    if (rc == GPX_OK) rc = side_rc;
  } else {
    for (int64_t k = 0; k < nblk && rc == GPX_OK; ++k) {
      rc = chain(k);
      if (rc == GPX_OK) rc = side(k);
    }
  }
  GPX_RET(rc);
  if (resident) {
    GPX_HIP_CHECK(h, hipEventRecord(h->ev_chain2, P2));
    GPX_HIP_CHECK(h, hipStreamWaitEvent(P, h->ev_chain2, 0));
  }
  GPX_HIP_CHECK(h, hipEventRecord(h->ev_join, T));
  GPX_HIP_CHECK(h, hipStreamWaitEvent(P, h->ev_join, 0));
  GPX_HIP_CHECK(h, hipEventRecord(h->ev_join2, Q));
  GPX_HIP_CHECK(h, hipStreamWaitEvent(P, h->ev_join2, 0));
  return GPX_OK;
}

// ---- the bordered factorisation, in three parts so that the enqueue part can be captured
// into a hipGraph: (1) sizes + allocations, (2) hyper-parameter upload, (3) pure enqueue.
struct FactorPlan {
  int kind; int apply_noise;
  int64_t m_rows; bool with_r;
  const double* noise_dev;   // per-row noise vector on the device, or nullptr
  bool want_k;               // full (not lower-only) K build for a K_out copy
};

static int factor_prepare(gpx_handle* h, int64_t m_rows, bool with_r) {
  if (h->n <= 0 || !h->d_x) { gpx_set_error(h, "no training inputs: call gpx_set_train first"); return GPX_ERR_STATE; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  const int64_t n = h->n, nb = h->nb;
  const int64_t npad = round_up(n, nb);
  const int64_t E = m_rows + (with_r ? 1 : 0);
  const int64_t epad = E > 0 ? round_up(E, 64) : 0;
  const int64_t rows = npad + epad, ld = npad;
  GPX_RET(ensure(h, &h->d_A, &h->a_cap, rows * ld));
  GPX_RET(ensure(h, &h->d_L, &h->l_cap, rows * ld));
  GPX_RET(ensure(h, &h->d_W, &h->w_cap, npad * nb));
  GPX_RET(ensure(h, &h->d_z, &h->z_cap, npad));
  GPX_RET(ensure_out(h, 2 * std::max<int64_t>(m_rows, 1)));
  GPX_RET(ensure_events(h, (size_t)(npad / nb)));
  h->have_factor = false; h->have_z = false;
  h->npad = npad; h->epad = epad; h->ld = ld;
  return GPX_OK;
}

// Build [K ; k_cross ; r] in d_A, factorise, leave [L ; V^T ; z^T] in d_L.  Enqueue only
// (kernel launches, memsets, device-to-device copies, stream events): capturable.
static int factor_enqueue(gpx_handle* h, const FactorPlan& fp, double* K_out) {
  const int64_t n = h->n, npad = h->npad, ld = h->ld; const int d = h->d;
  const int64_t m_rows = fp.m_rows;
  const int64_t E = m_rows + (fp.with_r ? 1 : 0);
  const int64_t rows = npad + h->epad;
  hipStream_t s = h->s_main;
  const double* dev_scalars = h->d_hyp + 2 * d;

  // K (diagonal rule on for k_train: gpax/gp.py:293 calls f(x, x, k_noise, k_jitter))
  GramLaunch g;
  g.kind = fp.kind; g.d = d; g.k_scale = 0.0; g.hyp = h->d_hyp; g.dev_scalars = dev_scalars;
  g.x1 = h->d_x; g.x2 = h->d_x; g.n1 = n; g.n2 = n; g.out = h->d_A; g.ld = ld;
  g.lower_only = fp.want_k ? 0 : 1; g.diag_rule = fp.apply_noise ? 1 : 0;
  g.noise = fp.noise_dev; g.noise_scalar = 0.0; g.jitter = 0.0;
  auto gram_cost = [&](int64_t c0, int64_t c1, double* fl, double* by) {   // lower tiles of the column slice [c0, c1)
    double entries = 0.0;
    for (int64_t c = c0 / 64; c * 64 < c1; ++c) entries += (double)((n + 63) / 64 - c) * 4096.0;
    if (fp.want_k) entries = (double)n * (double)(c1 - c0);
    *fl = gram_flops_per_entry(fp.kind, d) * entries; *by = 8.0 * entries + 8.0 * (double)(n + c1 - c0) * d;
  };
  auto cross_gram = [&](hipStream_t st) -> int {
    if (m_rows <= 0) return GPX_OK;
    GramLaunch c = g;
    c.x1 = h->d_xnew; c.n1 = m_rows; c.out = h->d_A + npad * ld; c.lower_only = 0; c.diag_rule = 0;
    c.noise = nullptr; c.zero_info = nullptr; c.zero_flags = nullptr; c.col0 = 0; c.x2 = h->d_x; c.n2 = n;
    c.start_event = c.stop_event = nullptr;
    prof_begin(h, st, GPX_STAGE_GRAM_CROSS, gram_flops_per_entry(fp.kind, d) * m_rows * n,
               8.0 * m_rows * n + 8.0 * (m_rows + n) * d);
    int rc = launch_gram(h, st, c);
    prof_end(h, st);
    return rc;
  };
  double fl, by;
  if (h->build_small && npad <= 256 && npad / h->nb < h->split_min_blocks && !fp.want_k && !K_out) {
    // one block column: the whole bordered matrix in ONE launch (gram.hip) -- at these sizes the step is launch latency
    gram_cost(0, n, &fl, &by);
    prof_begin(h, s, GPX_STAGE_GRAM, fl, by);
    GPX_RET(launch_build_small(h, s, fp.kind, h->d_x, h->d_xnew, h->d_hyp, dev_scalars, fp.noise_dev, h->d_r, h->d_A, ld, n, npad,
                               m_rows, fp.with_r ? npad + m_rows : -1, rows, d, fp.apply_noise ? 1 : 0, h->d_info, nullptr, 0));
    prof_end(h, s);
    const int lrc = factor_loop(h, rows);
    h->flags_clean = false;
    GPX_RET(lrc);
    return GPX_OK;
  }
  g.zero_info = h->d_info;
  if (h->d_flags) {   // the hand-over flags of the split-panel schedule are cleared by the same launch
    g.zero_flags = h->d_flags; g.n_zero_flags = (int)(3 * (npad / h->nb));
    h->flags_clean = true;
  }
  gram_cost(0, n, &fl, &by);
  (void)prof_pair(h, GPX_STAGE_GRAM, fl, by, &g.start_event, &g.stop_event);
  GPX_RET(launch_gram(h, s, g));
  if (K_out) {
    GPX_HIP_CHECK(h, hipMemcpy2DAsync(K_out, (size_t)n * sizeof(double), h->d_A, (size_t)ld * sizeof(double),
                                      (size_t)n * sizeof(double), (size_t)n, hipMemcpyDeviceToHost, s));
  }
  GPX_RET(launch_fill_identity_pad(h, s, h->d_A, ld, n, npad, rows));
  // The border (k_cross rows, the r row, the zero rows) is not read by the chain before the very last panel: with the
  // split-panel schedule it is built on the panel stream, whose first kernel (the panel of block column 0) is its
  // first reader, while the chain stream goes straight from K to potrf_diag(0) -- ~30 us off the head of the step.
  // (Building most of K itself on the bulk stream while the chain already factorises was tried: no gain at N = 4096,
  // 1.907 vs 1.905 ms -- the first steps are bound by that stream.)
  const bool border_on_q = h->border_side && h->lookahead != 0 && h->depth >= 3 && npad / h->nb >= h->split_min_blocks && !h->in_capture && h->prof_mode != 1;
  hipStream_t sb = s;
  if (border_on_q) {
    sb = h->s_far;
    GPX_HIP_CHECK(h, hipEventRecord(h->ev_build, s));
    GPX_HIP_CHECK(h, hipStreamWaitEvent(sb, h->ev_build, 0));
  }
  GPX_RET(cross_gram(sb));
  // the r row and the unused (zero) border rows in one launch
  GPX_RET(launch_border_rows(h, sb, h->d_A, ld, npad, n, fp.with_r ? npad + m_rows : -1, h->d_r, npad + E, rows, nullptr));

  const int lrc = factor_loop(h, rows);
  h->flags_clean = false;   // consumed by the split-panel loop, or unused by the others: never carried into another call
  GPX_RET(lrc);

  return GPX_OK;   // z = L^-1 r (row npad + m_rows of d_L) is copied to d_z by the fused epilogue of the caller
}

static void factor_commit(gpx_handle* h, const gpx_kernel_params* p, int64_t m_rows, bool with_r) {
  h->have_factor = true;
  h->m_in_factor = m_rows; h->r_in_factor = with_r;
  h->f_kind = p->kind; h->f_scale = p->k_scale; h->f_jitter = p->k_jitter;
  h->f_length.assign(p->k_length, p->k_length + p->n_length);
  h->f_period.clear();
  if (p->kind == GPX_PERIODIC) h->f_period.assign(p->k_period, p->k_period + p->n_period);
  h->have_z = with_r;
}

static int factor_common(gpx_handle* h, const gpx_kernel_params* p, int64_t m_rows, bool with_r, double* K_out,
                         double new_noise = 0.0) {
  GPX_RET(factor_prepare(h, m_rows, with_r));
  FactorPlan fp;
  double noise_scalar;
  GPX_RET(upload_params(h, h->s_main, p, h->n, h->d, &fp.noise_dev, &noise_scalar, new_noise));
  fp.kind = p->kind; fp.apply_noise = p->apply_noise; fp.m_rows = m_rows; fp.with_r = with_r; fp.want_k = K_out != nullptr;
  GPX_RET(factor_enqueue(h, fp, K_out));
  factor_commit(h, p, m_rows, with_r);
  return GPX_OK;
}

// info == -7: a resident kernel gave up waiting for its producer (potrf.hip / gemm.hip: bounded spin).  That only
// happens when something serialises the dispatches (a profiler collecting counters, a debugger): switch the handle
// to kernel-boundary hand-overs and tell the caller to repeat the call.
static int check_handover(gpx_handle* h, int info) {
  if (info != -7) return GPX_OK;
  h->resident = 0;
  h->have_factor = false; h->have_z = false;
  gpx_set_error(h, "in-kernel hand-over timed out (are kernel dispatches being serialised, e.g. by counter collection?); "
                   "switched to kernel-boundary hand-overs (option \"resident\" = 0): repeat the call");
  return GPX_ERR_STATE;
}

static int read_info(gpx_handle* h, int* info) {
  const double* blk;
  GPX_RET(fetch_block(h, 0, &blk));
  *info = reinterpret_cast<const int*>(blk + 8)[0];
  return check_handover(h, *info);
}

extern "C" int gpx_fit_factor(gpx_handle* h, const gpx_kernel_params* p, double* K_out, double* L_out) {
  if (!h) return GPX_ERR_ARG;
  GPX_RET(factor_common(h, p, 0, false, K_out));
  const int64_t n = h->n;
  if (L_out) {
    GPX_HIP_CHECK(h, hipMemcpy2DAsync(L_out, (size_t)n * sizeof(double), h->d_L, (size_t)h->ld * sizeof(double),
                                      (size_t)n * sizeof(double), (size_t)n, hipMemcpyDeviceToHost, h->s_main));
  }
  int info = 0;
  GPX_RET(read_info(h, &info));
  if (L_out) {  // only the lower triangle of the device buffer is defined
    for (int64_t i = 0; i < n; ++i)
      for (int64_t j = i + 1; j < n; ++j) L_out[i * n + j] = 0.0;
  }
  return info;
}

// ------------------------------------------------------- solves against the cached factor

// X L^T = S for `srows` right-hand-side ROWS (S, X: srows x npad, row-major): right-looking,
//   X_k = S_k W_k^T ;  S_j -= X_k L_jk^T  (j > k)   -- GEMMs only.
static int solve_rows(gpx_handle* h, double* S, double* X, int64_t srows) {
  const int64_t nb = h->nb, npad = h->npad, ld = h->ld, nblk = npad / nb;
  hipStream_t s = h->s_main;
  for (int64_t k = 0; k < nblk; ++k) {
    const int64_t c0 = k * nb, r0 = c0 + nb;
    GemmLaunch g;
    g.A = S + c0; g.lda = npad;
    g.B = h->d_W + k * nb * nb; g.ldb = nb;
    g.C = X + c0; g.ldc = npad;
    g.M = srows; g.N = nb; g.K = nb; g.alpha = 1.0; g.beta = 0.0;
    g.lower = 0; g.row_off = 0; g.col_off = 0; g.b_lower_tri = 1; g.small_tile = 1;
    GPX_RET(launch_gemm_nt(h, s, g));
    if (r0 >= npad) break;
    GemmLaunch u;
    u.A = X + c0; u.lda = npad;
    u.B = h->d_L + r0 * ld + c0; u.ldb = ld;
    u.C = S + r0; u.ldc = npad;
    u.M = srows; u.N = npad - r0; u.K = nb; u.alpha = -1.0; u.beta = 1.0;
    u.lower = 0; u.row_off = 0; u.col_off = 0; u.b_lower_tri = 0; u.small_tile = srows <= 64 ? 1 : 0;
    GPX_RET(launch_gemm_nt(h, s, u));
  }
  return GPX_OK;
}

extern "C" int gpx_log_marginal(gpx_handle* h, const double* y_minus_m, double* logp, double* alpha_out) {
  if (!h) return GPX_ERR_ARG;
  if (!h->have_factor) { gpx_set_error(h, "gpx_log_marginal: no cached factor (call gpx_fit_factor)"); return GPX_ERR_STATE; }
  if (!y_minus_m) { gpx_set_error(h, "gpx_log_marginal: y_minus_m is NULL"); return GPX_ERR_ARG; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  const int64_t n = h->n, npad = h->npad, nb = h->nb;
  hipStream_t s = h->s_main;
  GPX_RET(ensure(h, &h->d_r, &h->r_cap, n));
  GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_r, y_minus_m, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
  h->have_r = true;
  // one right-hand-side row, padded to a 64-row GEMM tile
  const int64_t srows = 64;
  GPX_RET(ensure(h, &h->d_S, &h->s_cap, 2 * srows * npad));
  double* S = h->d_S; double* X = h->d_S + srows * npad;
  GPX_HIP_CHECK(h, hipMemsetAsync(S, 0, (size_t)(srows * npad) * sizeof(double), s));
  GPX_RET(launch_set_row(h, s, S, h->d_r, n, npad));
  GPX_RET(solve_rows(h, S, X, srows));
  GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_z, X, (size_t)npad * sizeof(double), hipMemcpyDeviceToDevice, s));
  h->have_z = true;
  EpilogueLaunch e;
  e.L = h->d_L; e.ld = h->ld; e.n = n; e.z = h->d_z; e.vt = nullptr; e.vt_ld = 0; e.m = 0; e.m_new = nullptr;
  e.prior_diag = 0.0; e.scal = h->d_scal; e.mean = nullptr; e.var = nullptr;
  GPX_RET(launch_epilogue(h, s, e));
  if (alpha_out) {
    // alpha = L^-T z by blocked back-substitution (row form):
    //   alpha_i^T = (z_i^T - sum_{j>i} alpha_j^T L_ji) W_i
    GPX_RET(ensure(h, &h->d_vec, &h->vec_cap, 2 * npad));
    double* t = h->d_vec; double* al = h->d_vec + npad;
    GPX_HIP_CHECK(h, hipMemcpyAsync(t, h->d_z, (size_t)npad * sizeof(double), hipMemcpyDeviceToDevice, s));
    for (int64_t k = npad / nb - 1; k >= 0; --k) {
      const int64_t c0 = k * nb;
      GPX_RET(launch_gemv_t(h, s, h->d_W + k * nb * nb, nb, (int)nb, nb, t + c0, al + c0, 1.0, 0.0));
      GPX_RET(launch_gemv_t(h, s, h->d_L + c0 * h->ld, h->ld, (int)nb, c0, al + c0, t, -1.0, 1.0));
    }
    GPX_HIP_CHECK(h, hipMemcpyAsync(alpha_out, al, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  double scal[2]; int info = 0;
  GPX_HIP_CHECK(h, hipMemcpyAsync(scal, h->d_scal, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  GPX_RET(read_info(h, &info));
  if (logp) *logp = -0.5 * scal[1] - scal[0] - 0.5 * (double)n * log(2.0 * M_PI);
  return info;
}

// ----------------------------------------------- log-marginal likelihood with its gradient

// One bordered factorisation of [K ; r ; I] -> [L ; z^T ; U = L^-T] (the identity rows are activated
// block by block, factor_loop's tri_base), then alpha = U z, K^-1 = U U^T (lower tiles, dead-K skipped)
// and ONE pass over K^-1 that recomputes dK/dtheta from x (grad.hip).  ~N^3 flops in total against
// N^3/3 for the factorisation alone -- versus 2*p+1 factorisations for central differences.
extern "C" int gpx_lml_grad(gpx_handle* h, const gpx_kernel_params* p, const double* y_minus_m, double* logp,
                            double* grad_out, double* alpha_out) {
  if (!h) return GPX_ERR_ARG;
  if (!p || !y_minus_m) { gpx_set_error(h, "gpx_lml_grad: params / y_minus_m is NULL"); return GPX_ERR_ARG; }
  if (h->n <= 0 || !h->d_x) { gpx_set_error(h, "no training inputs: call gpx_set_train first"); return GPX_ERR_STATE; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  const int64_t n = h->n, nb = h->nb; const int d = h->d;
  const int64_t npad = round_up(n, nb), ld = npad;
  const int64_t RZ = 64;                              // the r row, padded to one GEMM tile
  const int64_t epad = RZ + round_up(n, 64);          // then the identity rows
  const int64_t rows = npad + epad, n64 = round_up(n, 64);
  hipStream_t s = h->s_main;

  h->have_factor = false; h->have_z = false;
  GPX_RET(ensure(h, &h->d_A, &h->a_cap, rows * ld));
  GPX_RET(ensure(h, &h->d_L, &h->l_cap, rows * ld));
  GPX_RET(ensure(h, &h->d_W, &h->w_cap, npad * nb));
  GPX_RET(ensure(h, &h->d_z, &h->z_cap, npad));
  GPX_RET(ensure(h, &h->d_r, &h->r_cap, n));
  GPX_RET(ensure(h, &h->d_vec, &h->vec_cap, 2 * npad));
  GPX_RET(ensure(h, &h->d_kinv, &h->kinv_cap, n64 * npad));
  // the small results travel in ONE pinned copy: [scal | info | alpha (npad) | the reduction's sums]
  const int per_dim = (p->n_length > 1 || (p->kind == GPX_PERIODIC && p->n_period > 1)) ? 1 : 0;
  const int DCH = grad_chunk_dims(), NPVv = grad_partials_per_chunk();
  const int nchunk = per_dim ? (d + DCH - 1) / DCH : 1;
  GPX_RET(ensure_out(h, npad + (int64_t)nchunk * NPVv));
  GPX_RET(ensure_events(h, (size_t)(npad / nb)));
  h->npad = npad; h->epad = epad; h->ld = ld;

  const double* noise_dev; double noise_scalar;
  GPX_RET(upload_params(h, s, p, n, d, &noise_dev, &noise_scalar));
  // small problems: the one-launch build below reads r straight from a pinned slot (and writes d_r on the way): no upload in the stream
  const bool r_in_place = h->build_small && npad <= 256 && npad / nb < h->split_min_blocks;
  const double* r_src = h->d_r; int r_slot = -1;
  if (r_in_place) GPX_RET(stage_small(h, y_minus_m, n, &r_src, &r_slot));
  else if (n <= 65536) GPX_RET(upload_small(h, s, h->d_r, y_minus_m, n));   // through the pinned ring: truly asynchronous
  else GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_r, y_minus_m, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
  h->have_r = true;

  // ---- [K ; r ; I] in d_A
  const bool one_launch = h->build_small && npad <= 256 && npad / nb < h->split_min_blocks;   // one stream, small: one launch builds everything below (gram.hip)
  if (one_launch) {
    const double T = (double)((n + 63) / 64);
    const double entries = T * (T + 1) / 2.0 * 4096.0;
    prof_begin(h, s, GPX_STAGE_GRAM, gram_flops_per_entry(p->kind, d) * entries, 8.0 * entries + 16.0 * n * d);
    GPX_RET(launch_build_small(h, s, p->kind, h->d_x, nullptr, h->d_hyp, h->d_hyp + 2 * d, noise_dev, r_src, h->d_A, ld, n, npad, 0,
                               npad, rows, d, p->apply_noise ? 1 : 0, h->d_info, nullptr, 0, npad + RZ, h->d_L + npad * ld,
                               r_slot >= 0 ? h->d_r : nullptr));
    prof_end(h, s);
    if (r_slot >= 0) GPX_RET(stage_small_done(h, s, r_slot));
  }
  GramLaunch g;
  g.kind = p->kind; g.d = d; g.k_scale = 0.0; g.hyp = h->d_hyp; g.dev_scalars = h->d_hyp + 2 * d;
  g.x1 = h->d_x; g.x2 = h->d_x; g.n1 = n; g.n2 = n; g.out = h->d_A; g.ld = ld;
  g.lower_only = 1; g.diag_rule = p->apply_noise ? 1 : 0;
  g.noise = noise_dev; g.noise_scalar = 0.0; g.jitter = 0.0;
  if (!one_launch) {
  {
    const double T = (double)((n + 63) / 64);
    const double entries = T * (T + 1) / 2.0 * 4096.0;
    prof_begin(h, s, GPX_STAGE_GRAM, gram_flops_per_entry(p->kind, d) * entries, 8.0 * entries + 16.0 * n * d);
    GPX_RET(launch_gram(h, s, g));
    prof_end(h, s);
  }
  GPX_RET(launch_fill_identity_pad(h, s, h->d_A, ld, n, npad, rows));
  double* Ab = h->d_A + npad * ld;                       // border
  GPX_RET(launch_set_row(h, s, Ab, h->d_r, n, npad));
  GPX_RET(launch_fill_rect(h, s, Ab, ld, 1, RZ, 0, n, 0.0));
  GPX_RET(launch_fill_rect(h, s, Ab + RZ * ld, ld, 0, n, 0, n, 1.0));          // identity rows
  GPX_RET(launch_fill_rect(h, s, Ab + RZ * ld, ld, n, epad - RZ, 0, n, 0.0));
  // border rows of L that no panel writes (the dead lower-left of U) must read as zeros
  GPX_HIP_CHECK(h, hipMemsetAsync(h->d_L + npad * ld, 0, (size_t)(epad * ld) * sizeof(double), s));
  GPX_HIP_CHECK(h, hipMemsetAsync(h->d_info, 0, sizeof(int), s));
  }

  const double* zrow = h->d_L + npad * ld;
  const double* U = h->d_L + (npad + RZ) * ld;
  // K^-1 = U U^T (lower 64x64 tiles) is accumulated panel by panel inside the split-panel schedule; the classic
  // schedules (options "depth" 1|2, "lookahead" 0) form it in one product after the factorisation
  const bool fused_kinv = h->lookahead != 0 && h->depth >= 3 && npad / nb >= h->split_min_blocks;   // same test as factor_loop's dispatch
  if (fused_kinv) { h->kinv_U = U; h->kinv_rows = n64; }
  int frc = factor_loop(h, rows, npad + RZ);
  h->kinv_U = nullptr; h->kinv_rows = 0;
  GPX_RET(frc);

  EpilogueLaunch e;
  e.L = h->d_L; e.ld = ld; e.n = n; e.z = zrow; e.vt = nullptr; e.vt_ld = 0; e.m = 0; e.m_new = nullptr;
  e.prior_diag = 0.0; e.scal = h->d_scal; e.mean = nullptr; e.var = nullptr;
  e.z_copy = h->d_z; e.npad = npad;
  // ---- alpha = L^-T z = U z rides on the epilogue's launch (extra blocks; it reads z from the border row itself)
  double* alpha = h->d_mean;   // inside the result block
  e.gemv_U = U; e.gemv_alpha = alpha; e.gemv_ncols = npad;
  GPX_RET(launch_epilogue(h, s, e));

  // ---- K^-1 = U U^T, lower 64x64 tiles; U_ik = 0 for k < i, so each tile starts at its first row
  if (!fused_kinv) {
  GemmLaunch u;
  u.A = U; u.lda = ld; u.B = U; u.ldb = ld; u.C = h->d_kinv; u.ldc = npad;
  u.M = n64; u.N = n64; u.K = npad; u.alpha = 1.0; u.beta = 0.0;
  u.lower = 1; u.row_off = 0; u.col_off = 0; u.b_lower_tri = 0; u.small_tile = 0; u.a_upper_tri = 1;
  {
    const double T = (double)(n64 / 64);
    prof_begin(h, s, GPX_STAGE_TRAILING, 2.0 * 4096.0 * 64.0 * T * (T + 1) * (T + 2) / 6.0, 8.0 * (double)n64 * npad);
    GPX_RET(launch_gemm_nt(h, s, u));
    prof_end(h, s);
  }
  }

  // ---- the reduction
  const int64_t T = (n + 63) / 64, ntiles = T * (T + 1) / 2;
  const int nwg = ntiles <= 3 ? 1 : (int)std::min<int64_t>(ntiles, 1024);   // up to three tiles (n <= 128): one workgroup, no final-sum launch
  GPX_RET(ensure(h, &h->d_gpart, &h->gpart_cap, (int64_t)nchunk * nwg * NPVv));
  double* sums_dev = h->d_mean + npad;
  GPX_RET(launch_grad_reduce(h, s, p->kind, h->d_x, h->d_hyp, h->d_kinv, npad, alpha, n, d, per_dim, nwg, h->d_gpart,
                             sums_dev));

  const double* blk;
  GPX_RET(fetch_block(h, npad + (int64_t)nchunk * NPVv, &blk));
  const double scal[2] = {blk[0], blk[1]};
  const int info = reinterpret_cast<const int*>(blk + 8)[0];
  GPX_RET(check_handover(h, info));
  const double* sums = blk + GPX_OUT_HEAD + npad;
  if (alpha_out) memcpy(alpha_out, blk + GPX_OUT_HEAD, (size_t)n * sizeof(double));
  factor_commit(h, p, 0, true);
  if (logp) *logp = -0.5 * scal[1] - scal[0] - 0.5 * (double)n * log(2.0 * M_PI);

  if (grad_out) {
    // layout: [k_scale | k_length (n_length) | k_period (n_period, periodic only) | k_noise]
    const double sc = p->k_scale;
    int o = 0;
    grad_out[o++] = 0.5 * sums[0];
    auto dim_sum = [&](int k, int which) { return sums[(size_t)(k / DCH) * NPVv + 4 + which * DCH + k % DCH]; };
    if (p->n_length == 1) {
      double acc = 0.0;
      if (per_dim) for (int k = 0; k < d; ++k) acc += dim_sum(k, 0); else acc = sums[2];
      grad_out[o++] = 0.5 * sc * acc / p->k_length[0];
    } else {
      for (int k = 0; k < d; ++k) grad_out[o++] = 0.5 * sc * dim_sum(k, 0) / p->k_length[k];
    }
    if (p->kind == GPX_PERIODIC) {
      if (p->n_period == 1) {
        double acc = 0.0;   // the sums carry 1/l_d^2 already
        if (per_dim) for (int k = 0; k < d; ++k) acc += dim_sum(k, 1); else acc = sums[3];
        grad_out[o++] = 0.5 * sc * acc / (2.0 * p->k_period[0]);
      } else {
        for (int k = 0; k < d; ++k) grad_out[o++] = 0.5 * sc * dim_sum(k, 1) / (2.0 * p->k_period[k]);
      }
    }
    grad_out[o++] = (p->apply_noise && p->n_noise == 1) ? 0.5 * sums[1] : 0.0;
    if (info != 0) for (int i = 0; i < o; ++i) grad_out[i] = NAN;
  }
  return info;
}

static void cached_params(gpx_handle* h, gpx_kernel_params* q) {
  memset(q, 0, sizeof(*q));
  q->kind = h->f_kind; q->k_scale = h->f_scale; q->k_jitter = h->f_jitter;
  q->k_length = h->f_length.data(); q->n_length = (int)h->f_length.size();
  q->k_period = h->f_period.empty() ? nullptr : h->f_period.data(); q->n_period = (int)h->f_period.size();
  q->k_noise = nullptr; q->n_noise = 0; q->apply_noise = 0;
}

extern "C" int gpx_posterior(gpx_handle* h, const double* m_new, double new_noise, double* mean_out,
                             double* var_out, double* cov_out) {
  if (!h) return GPX_ERR_ARG;
  if (!h->have_factor || !h->have_z) {
    gpx_set_error(h, "gpx_posterior: needs gpx_fit_factor + gpx_log_marginal (or gpx_fit_predict) first");
    return GPX_ERR_STATE;
  }
  const int64_t m = h->m;
  if (m <= 0) return GPX_OK;  // empty x_new: nothing to write
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  const int64_t n = h->n, npad = h->npad; const int d = h->d;
  hipStream_t s = h->s_main;
  const int64_t mpad = round_up(m, 64);
  GPX_RET(ensure(h, &h->d_S, &h->s_cap, 2 * mpad * npad));
  double* S = h->d_S; double* X = h->d_S + mpad * npad;
  gpx_kernel_params q; cached_params(h, &q);
  const double* noise_dev; double noise_scalar;
  GPX_RET(upload_params(h, s, &q, m, d, &noise_dev, &noise_scalar));
  if (mpad != m || npad != n) GPX_HIP_CHECK(h, hipMemsetAsync(S, 0, (size_t)(mpad * npad) * sizeof(double), s));
  GramLaunch g;
  g.kind = q.kind; g.d = d; g.k_scale = q.k_scale; g.hyp = h->d_hyp;
  g.x1 = h->d_xnew; g.x2 = h->d_x; g.n1 = m; g.n2 = n; g.out = S; g.ld = npad;
  g.lower_only = 0; g.diag_rule = 0; g.noise = nullptr; g.noise_scalar = 0.0; g.jitter = 0.0;
  GPX_RET(launch_gram(h, s, g));  // k_cross = f(x_new, x, apply_noise=False)  gpax/gp.py:282
  GPX_RET(solve_rows(h, S, X, mpad));
  GPX_RET(ensure_out(h, 2 * m));
  h->d_var = h->d_mean + m;
  const double* mnew_dev = nullptr;
  if (m_new) {
    GPX_RET(ensure(h, &h->d_mnew, &h->mnew_cap, m));
    GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_mnew, m_new, (size_t)m * sizeof(double), hipMemcpyHostToDevice, s));
    mnew_dev = h->d_mnew;
  }
  EpilogueLaunch e;
  e.L = h->d_L; e.ld = h->ld; e.n = n; e.z = h->d_z; e.vt = X; e.vt_ld = npad; e.m = m; e.m_new = mnew_dev;
  e.prior_diag = kernel_diag_value(q.kind, q.k_scale) + (new_noise + q.k_jitter);  // diagonal of k_new, gp.py:287
  e.scal = h->d_scal + 2; e.mean = h->d_mean; e.var = h->d_var;
  GPX_RET(launch_epilogue(h, s, e));
  if (cov_out) {
    // cov = k_new - V^T V  (gpax/gp.py:298-299 with A = K^-1 k_cross^T = L^-T V)
    GPX_RET(ensure(h, &h->d_cov, &h->cov_cap, mpad * mpad));
    GramLaunch c = g;
    c.x2 = h->d_xnew; c.n2 = m; c.out = h->d_cov; c.ld = mpad; c.diag_rule = 1; c.noise_scalar = new_noise;
    c.jitter = q.k_jitter;
    GPX_RET(launch_gram(h, s, c));
    GemmLaunch u;
    u.A = X; u.lda = npad; u.B = X; u.ldb = npad; u.C = h->d_cov; u.ldc = mpad;
    u.M = mpad; u.N = mpad; u.K = npad; u.alpha = -1.0; u.beta = 1.0;
    u.lower = 0; u.row_off = 0; u.col_off = 0; u.b_lower_tri = 0; u.small_tile = mpad <= 256 ? 1 : 0;
    GPX_RET(launch_gemm_nt(h, s, u));
    GPX_HIP_CHECK(h, hipMemcpy2DAsync(cov_out, (size_t)m * sizeof(double), h->d_cov, (size_t)mpad * sizeof(double),
                                      (size_t)m * sizeof(double), (size_t)m, hipMemcpyDeviceToHost, s));
  }
  const double* blk;   // [scal | info | mean(m) | var(m)] in one copy
  GPX_RET(fetch_block(h, 2 * m, &blk));
  GPX_RET(check_handover(h, reinterpret_cast<const int*>(blk + 8)[0]));
  if (mean_out) memcpy(mean_out, blk + GPX_OUT_HEAD, (size_t)m * sizeof(double));
  if (var_out) memcpy(var_out, blk + GPX_OUT_HEAD + m, (size_t)m * sizeof(double));
  return reinterpret_cast<const int*>(blk + 8)[0];
}

extern "C" int gpx_prior_cov(gpx_handle* h, const gpx_kernel_params* p, double new_noise, double* cov_out) {
  if (!h) return GPX_ERR_ARG;
  if (!cov_out || !p) { gpx_set_error(h, "gpx_prior_cov: bad argument"); return GPX_ERR_ARG; }
  const int64_t m = h->m;
  if (m <= 0 || !h->d_xnew) { gpx_set_error(h, "gpx_prior_cov: call gpx_set_test first"); return GPX_ERR_STATE; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  hipStream_t s = h->s_main;
  gpx_kernel_params q = *p; q.n_noise = 0; q.k_noise = nullptr;
  const double* noise_dev; double noise_scalar;
  GPX_RET(upload_params(h, s, &q, m, h->d, &noise_dev, &noise_scalar));
  GPX_RET(ensure(h, &h->d_cov, &h->cov_cap, m * m));
  GramLaunch g;
  g.kind = q.kind; g.d = h->d; g.k_scale = q.k_scale; g.hyp = h->d_hyp;
  g.x1 = h->d_xnew; g.x2 = h->d_xnew; g.n1 = m; g.n2 = m; g.out = h->d_cov; g.ld = m;
  g.lower_only = 0; g.diag_rule = 1; g.noise = nullptr; g.noise_scalar = new_noise; g.jitter = q.k_jitter;
  GPX_RET(launch_gram(h, s, g));
  GPX_HIP_CHECK(h, hipMemcpyAsync(cov_out, h->d_cov, (size_t)(m * m) * sizeof(double), hipMemcpyDeviceToHost, s));
  GPX_HIP_CHECK(h, hipStreamSynchronize(s));
  return GPX_OK;
}

// ------------------------------------------------------------------------ MVN sampling

// dist.MultivariateNormal(mean, cov).sample(key, (n,)) of gpax/gp.py:335-337:  out = mean + eps L_c^T,
// L_c = chol(cov).  The (m x m) Cholesky runs through the same blocked factorisation (its own
// buffers: the cached GP factor stays valid); the draws eps ~ N(0, I) come from the caller.
extern "C" int gpx_mvn_sample(gpx_handle* h, const double* mean, const double* cov, int64_t m, const double* eps,
                              int64_t nsamp, double* out) {
  if (!h) return GPX_ERR_ARG;
  if (!cov || !eps || !out || m <= 0 || nsamp <= 0) { gpx_set_error(h, "gpx_mvn_sample: bad argument"); return GPX_ERR_ARG; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  hipStream_t s = h->s_main;
  const int sv_nb = h->nb;
  const int64_t nb = h->nb_auto ? (m <= 64 ? 64 : 128) : h->nb;   // the sample covariance is its own problem size
  const int64_t npad = round_up(m, nb), spad = round_up(nsamp, 64);
  GPX_RET(ensure(h, &h->d_A2, &h->a2_cap, npad * npad));
  GPX_RET(ensure(h, &h->d_L2, &h->l2_cap, npad * npad));
  GPX_RET(ensure(h, &h->d_W2, &h->w2_cap, npad * nb));
  GPX_RET(ensure(h, &h->d_S, &h->s_cap, 2 * spad * npad));
  h->nb = (int)nb;
  int erc = ensure_events(h, (size_t)(npad / nb));   // sizes its scratch by h->nb
  h->nb = sv_nb;
  GPX_RET(erc);
  GPX_HIP_CHECK(h, hipMemcpy2DAsync(h->d_A2, (size_t)npad * sizeof(double), cov, (size_t)m * sizeof(double),
                                    (size_t)m * sizeof(double), (size_t)m, hipMemcpyHostToDevice, s));
  GPX_RET(launch_fill_identity_pad(h, s, h->d_A2, npad, m, npad, npad));
  GPX_HIP_CHECK(h, hipMemsetAsync(h->d_info, 0, sizeof(int), s));
  // run the factor loop on the secondary buffers
  std::swap(h->d_A, h->d_A2); std::swap(h->d_L, h->d_L2); std::swap(h->d_W, h->d_W2);
  const int64_t sv_npad = h->npad, sv_ld = h->ld, sv_epad = h->epad;
  h->npad = npad; h->ld = npad; h->epad = 0; h->nb = (int)nb;
  int rc = factor_loop(h, npad);
  double* Lc = h->d_L;
  h->npad = sv_npad; h->ld = sv_ld; h->epad = sv_epad; h->nb = sv_nb;
  std::swap(h->d_A, h->d_A2); std::swap(h->d_L, h->d_L2); std::swap(h->d_W, h->d_W2);
  GPX_RET(rc);
  double* E = h->d_S; double* Y = h->d_S + spad * npad;
  GPX_HIP_CHECK(h, hipMemsetAsync(E, 0, (size_t)(spad * npad) * sizeof(double), s));
  GPX_HIP_CHECK(h, hipMemcpy2DAsync(E, (size_t)npad * sizeof(double), eps, (size_t)m * sizeof(double),
                                    (size_t)m * sizeof(double), (size_t)nsamp, hipMemcpyHostToDevice, s));
  GemmLaunch g;
  g.A = E; g.lda = npad; g.B = Lc; g.ldb = npad; g.C = Y; g.ldc = npad;
  g.M = spad; g.N = npad; g.K = npad; g.alpha = 1.0; g.beta = 0.0;
  g.lower = 0; g.row_off = 0; g.col_off = 0; g.b_lower_tri = 1; g.small_tile = 0;
  GPX_RET(launch_gemm_nt(h, s, g));
  GPX_HIP_CHECK(h, hipMemcpy2DAsync(out, (size_t)m * sizeof(double), Y, (size_t)npad * sizeof(double),
                                    (size_t)m * sizeof(double), (size_t)nsamp, hipMemcpyDeviceToHost, s));
  int info = 0;
  GPX_RET(read_info(h, &info));
  if (mean)
    for (int64_t r = 0; r < nsamp; ++r)
      for (int64_t c = 0; c < m; ++c) out[r * m + c] += mean[c];
  return info;
}

// ------------------------------------------------------------------ fused fit+predict

static int fit_predict_enqueue(gpx_handle* h, const FactorPlan& fp) {
  const int64_t m = fp.m_rows;
  hipStream_t s = h->s_main;
  EpilogueLaunch e;
  e.L = h->d_L; e.ld = h->ld; e.n = h->n; e.z = h->d_L + (h->npad + m) * h->ld;
  e.vt = h->d_L + h->npad * h->ld; e.vt_ld = h->ld; e.m = m;
  e.m_new = h->have_mnew ? h->d_mnew : nullptr;
  e.prior_diag = 0.0; e.dev_prior_diag = h->d_hyp + 2 * h->d + 3;
  e.scal = h->d_scal; e.mean = h->d_mean; e.var = h->d_var;
  e.z_copy = h->d_z; e.npad = h->npad;   // keep z for later posterior calls against the cached factor
  // split epilogue (split-panel schedule, three block columns or more): see side() in factor_loop_split
  h->epi = nullptr; h->epi_split_col = 0;
  if (h->epi_split && !h->in_capture && h->prof_mode != 1 && h->lookahead != 0 && h->depth >= 3 && h->npad / h->nb >= std::max(3, h->split_min_blocks)) {
    GPX_RET(ensure(h, &h->d_epart, &h->epart_cap, 2 + 2 * std::max<int64_t>(m, 1)));
    h->epi = &e;
  }
  const int frc = factor_enqueue(h, fp, nullptr);
  h->epi = nullptr;
  GPX_RET(frc);
  prof_begin(h, s, GPX_STAGE_EPILOGUE, 4.0 * (m + 1.0) * h->n, 8.0 * (m + 2.0) * h->n);
  if (h->epi_split_col > 0) GPX_RET(launch_epilogue_final(h, s, e, h->epi_split_col, h->d_epart));
  else GPX_RET(launch_epilogue(h, s, e));
  prof_end(h, s);
  h->epi_split_col = 0;
  return GPX_OK;
}

extern "C" int gpx_fit_predict_async(gpx_handle* h, const gpx_kernel_params* p, double new_noise) {
  if (!h) return GPX_ERR_ARG;
  if (!h->have_r) { gpx_set_error(h, "gpx_fit_predict_async: call gpx_set_targets first"); return GPX_ERR_STATE; }
  const int64_t m = h->m;
  hipStream_t s = h->s_main;
  GPX_RET(factor_prepare(h, m, true));
  h->d_var = h->d_mean + std::max<int64_t>(m, 1);
  FactorPlan fp;
  double noise_scalar;
  GPX_RET(upload_params(h, s, p, h->n, h->d, &fp.noise_dev, &noise_scalar, new_noise));
  fp.kind = p->kind; fp.apply_noise = p->apply_noise; fp.m_rows = m; fp.with_r = true; fp.want_k = false;

  if (h->use_graph && !h->prof_mode) {
    // One captured graph per (shape, kernel kind, buffer addresses); hyper-parameter VALUES live in
    // d_hyp / d_noise / d_r / d_mnew, so the same graph is replayed for every hyper-parameter draw.
    char key[512];
    snprintf(key, sizeof(key), "%d|%d|%lld|%lld|%d|%d|%d|%p|%p|%p|%p|%p|%p|%p|%p|%p|%p|%p", fp.kind, fp.apply_noise,
             (long long)h->n, (long long)m, h->d, h->nb, h->lookahead * 4 + h->depth + 16 * h->pair_min + 4096 * h->quad_min, (void*)h->d_A, (void*)h->d_L, (void*)h->d_W,
             (void*)h->d_x, (void*)h->d_xnew, (void*)h->d_r, (void*)(h->have_mnew ? h->d_mnew : nullptr),
             (void*)h->d_hyp, (void*)fp.noise_dev, (void*)h->d_mean, (void*)h->d_z);
    auto it = h->graphs.find(key);
    if (it == h->graphs.end() && !h->graph_warm.count(key)) {
      // first call of a shape runs eagerly: lazy one-time initialisation (code-object loading, first use of the
      // side streams) inside a stream capture crashed the ROCm 7.2 runtime; the capture happens on the second call
      h->graph_warm.insert(key);
      GPX_RET(fit_predict_enqueue(h, fp));
      factor_commit(h, p, m, true);
      return GPX_OK;
    }
    if (it == h->graphs.end()) {
      hipGraph_t graph = nullptr;
      const bool trace = getenv("GPX_TRACE") != nullptr;
      if (trace) fprintf(stderr, "[gpx] begin capture\n");
      GPX_HIP_CHECK(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      h->in_capture = true;
      int rc = fit_predict_enqueue(h, fp);
      h->in_capture = false;
      if (trace) fprintf(stderr, "[gpx] enqueue rc=%d, end capture\n", rc);
      hipError_t ce = hipStreamEndCapture(s, &graph);
      if (trace) fprintf(stderr, "[gpx] end capture: %s\n", hipGetErrorString(ce));
      if (rc != GPX_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
      if (ce != hipSuccess) { gpx_set_error(h, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce)); return GPX_ERR_HIP; }
      hipGraphExec_t exec = nullptr;
      ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      if (trace) fprintf(stderr, "[gpx] instantiate: %s\n", hipGetErrorString(ce));
      (void)hipGraphDestroy(graph);
      if (ce != hipSuccess) { gpx_set_error(h, std::string("hipGraphInstantiate: ") + hipGetErrorString(ce)); return GPX_ERR_HIP; }
      if (h->graphs.size() >= 16) {  // bound the cache
        for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
        h->graphs.clear();
      }
      it = h->graphs.emplace(key, exec).first;
    }
    GPX_HIP_CHECK(h, hipGraphLaunch(it->second, s));
  } else {
    prof_begin(h, s, GPX_STAGE_FACTOR_TOTAL, 0.0, 0.0);
    GPX_RET(fit_predict_enqueue(h, fp));
    if (h->prof_mode) {  // close the FACTOR_TOTAL bracket (first pending of this call)
      for (size_t i = h->pending.size(); i-- > 0;)
        if (h->pending[i].stage == GPX_STAGE_FACTOR_TOTAL) { (void)hipEventRecord(h->pending[i].b, s); break; }
    }
  }
  factor_commit(h, p, m, true);
  return GPX_OK;
}

extern "C" int gpx_fetch_results(gpx_handle* h, double* logp, double* mean_out, double* var_out) {
  if (!h) return GPX_ERR_ARG;
  if (!h->have_factor || !h->r_in_factor) { gpx_set_error(h, "gpx_fetch_results: no fit_predict results"); return GPX_ERR_STATE; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  hipStream_t s = h->s_main;
  const int64_t m = h->m_in_factor;
  (void)s;
  // one copy: [scal | info | mean(m) | var(m)] (d_var = d_mean + max(m, 1), set by gpx_fit_predict_async)
  const double* blk;
  GPX_RET(fetch_block(h, m > 0 ? 2 * m : 0, &blk));
  const double scal[2] = {blk[0], blk[1]};
  const int info = reinterpret_cast<const int*>(blk + 8)[0];
  GPX_RET(check_handover(h, info));
  if (mean_out && m > 0) memcpy(mean_out, blk + GPX_OUT_HEAD, (size_t)m * sizeof(double));
  if (var_out && m > 0) memcpy(var_out, blk + GPX_OUT_HEAD + m, (size_t)m * sizeof(double));
  if (logp) *logp = -0.5 * scal[1] - scal[0] - 0.5 * (double)h->n * log(2.0 * M_PI);
  return info;
}

extern "C" int gpx_fit_predict(gpx_handle* h, const gpx_kernel_params* p, const double* y_minus_m,
                               const double* m_new, double new_noise, double* logp, double* mean_out,
                               double* var_out) {
  if (!h) return GPX_ERR_ARG;
  GPX_RET(gpx_set_targets(h, y_minus_m, m_new));
  GPX_RET(gpx_fit_predict_async(h, p, new_noise));
  return gpx_fetch_results(h, logp, mean_out, var_out);
}

// ------------------------------------------------------------------- measurement API

extern "C" int gpx_profile_mode(gpx_handle* h, int mode) {
  if (!h) return GPX_ERR_ARG;
  GPX_RET(prof_collect(h));
  h->prof_mode = mode;
  return GPX_OK;
}

// Diagnostic: write (stage, stream, start_us, end_us) of every bracketed launch still pending
// (i.e. since the last collect) to a CSV; times are relative to the first event.
extern "C" int gpx_profile_dump(gpx_handle* h, const char* path) {
  if (!h || !path) return GPX_ERR_ARG;
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_main));
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_trail));
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_far));
  FILE* f = fopen(path, "w");
  if (!f) { gpx_set_error(h, "gpx_profile_dump: cannot open file"); return GPX_ERR_ARG; }
  fprintf(f, "stage,start_us,end_us\n");
  if (!h->pending.empty()) {
    hipEvent_t e0 = h->pending.front().a;
    for (auto& p : h->pending) {
      float a = 0.f, b = 0.f;
      (void)hipEventElapsedTime(&a, e0, p.a);
      (void)hipEventElapsedTime(&b, e0, p.b);
      fprintf(f, "%d,%.2f,%.2f\n", p.stage, a * 1e3, b * 1e3);
    }
  }
  fclose(f);
  return GPX_OK;
}

extern "C" int gpx_profile_reset(gpx_handle* h) {
  if (!h) return GPX_ERR_ARG;
  GPX_RET(prof_collect(h));
  for (auto& p : h->prof) p = StageProf();
  return GPX_OK;
}

extern "C" int gpx_profile_get(gpx_handle* h, int stage, double* total_ms, int64_t* launches, double* flops,
                               double* bytes) {
  if (!h || stage < 0 || stage >= GPX_N_STAGES) return GPX_ERR_ARG;
  GPX_RET(prof_collect(h));
  if (total_ms) *total_ms = h->prof[stage].ms;
  if (launches) *launches = h->prof[stage].launches;
  if (flops) *flops = h->prof[stage].flops;
  if (bytes) *bytes = h->prof[stage].bytes;
  return GPX_OK;
}
