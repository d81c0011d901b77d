// api.hip -- C ABI (include/gpx.h) and the host-side drivers of the HIP path:
// the bordered right-looking blocked Cholesky with two-stream look-ahead, the
// solves against a cached factor, and the measurement hooks.

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "gpx_internal.h"

static std::string g_create_error;

static std::mutex g_err_mutex;

void gpx_set_error(gpx_handle* h, const std::string& msg) {
  std::lock_guard<std::mutex> lk(g_err_mutex);
  if (h) h->err = msg; else g_create_error = msg;
}

// --------------------------------------------------------------------------- memory

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

// (Re)allocate the result block for `batch` draws with `tail` doubles (mean + var, or alpha + sums) behind each head; the
// head of draw 0 (scal, info) survives a growth.  Sets the per-draw stride s_out.
static int ensure_out(gpx_handle* h, int64_t tail, int batch = 1) {
  h->mirror_valid = false;   // the result block is about to be rewritten
  const int64_t stride = round_up(GPX_OUT_HEAD + std::max<int64_t>(tail, 2), 2), need = stride * batch;
  h->s_out = stride;
  if (h->d_out && h->out_cap >= need) return GPX_OK;
  double* nw = nullptr;
  GPX_HIP_CHECK(h, hipMalloc(reinterpret_cast<void**>(&nw), (size_t)need * sizeof(double)));
  GPX_HIP_CHECK(h, hipDeviceSynchronize());
  GPX_HIP_CHECK(h, hipMemset(nw, 0, (size_t)need * sizeof(double)));
  if (h->d_out) {
    GPX_HIP_CHECK(h, hipMemcpy(nw, h->d_out, GPX_OUT_HEAD * sizeof(double), hipMemcpyDeviceToDevice));
    (void)hipFree(h->d_out);
  }
  h->d_out = nw; h->out_cap = need;
  h->d_scal = nw; h->d_info = reinterpret_cast<int*>(nw + 8); h->d_mean = nw + GPX_OUT_HEAD; h->d_var = h->d_mean + 1;
  return GPX_OK;
}

// One device-to-host copy (into pinned memory) of [scal | info | first `tail` doubles after the head] (of every draw of a
// batch: `batch` blocks of s_out doubles), then a stream sync.  Returns a pointer to the pinned copy: scal at [0], info at
// ((int*)(p + 8))[0], mean at p + 10; draw b at p + b * s_out.
// Pinned host mirror of the result block: the final kernels of a single-draw step write their results there themselves
// (EpilogueLaunch::mirror, launch_grad_reduce), so that fetching them is a stream synchronisation, not a copy behind the last kernel.
static int ensure_mirror(gpx_handle* h, int64_t cnt) {
  if (h->mirror_cap >= cnt) return GPX_OK;
  if (h->h_mirror) { GPX_HIP_CHECK(h, hipDeviceSynchronize()); (void)hipHostFree(h->h_mirror); }
  h->h_mirror = nullptr; h->mirror_cap = 0;
  GPX_HIP_CHECK(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_mirror), (size_t)cnt * sizeof(double), hipHostMallocDefault));
  h->mirror_cap = cnt;
  return GPX_OK;
}

static int fetch_block(gpx_handle* h, int64_t tail, const double** out, int batch = 1, bool mirror_ok = false) {
  const int64_t cnt = batch > 1 ? h->s_out * batch : GPX_OUT_HEAD + tail;
  if (mirror_ok && h->mirror_valid && batch == 1 && cnt <= h->mirror_cap) {   // the last kernel wrote the block to pinned memory itself
    GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_main));
    *out = h->h_mirror;
    return GPX_OK;
  }
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
  if (!g_tl_ends && hipMalloc(reinterpret_cast<void**>(&g_tl_ends), 16384 * 8) != hipSuccess) return -1;   // ends, then publish times
  (void)hipMemset(g_tl_dev, 0, (4 + 4 * 16000) * 8);
  (void)hipMemset(g_tl_ends, 0, 16384 * 8);
  tl_set_gemm(g_tl_dev, g_tl_ends); tl_set_potrf(g_tl_dev);
  return 0;
}
extern "C" long long gpx_timeline_read_ends(unsigned long long* out, long long max_records) {
  (void)hipDeviceSynchronize();
  const long long n = max_records < 16384 ? max_records : 16384;
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

// Four streams: P (s_main) and P2 (s_main2) carry the chain at the highest priority, Q (s_far) the panels and
// look-ahead columns, T (s_trail) the bulk updates at the lowest.  (CU masks that keep T off a few CUs for the
// chain were measured in round 1 and cost more than they return: N = 32768 228.5 -> 237 ms; removed.)
static hipError_t create_streams(gpx_handle* h) {
  int lo = 0, hi = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = least, hi = greatest priority
  if (e != hipSuccess) return e;
  if ((e = hipStreamCreateWithPriority(&h->s_main, hipStreamNonBlocking, hi)) != hipSuccess) return e;
  if ((e = hipStreamCreateWithPriority(&h->s_main2, hipStreamNonBlocking, hi)) != hipSuccess) return e;
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
  // in-kernel hand-overs need the chain's kernels to run CONCURRENTLY; hardware-counter collection
  // (rocprofv3 --pmc) serialises every dispatch, so fall back to kernel-boundary hand-overs under it
  if (getenv("ROCPROF_COUNTER_COLLECTION") || getenv("ROCPROF_COUNTERS")) h->resident = 0;
  const char* env;
  if ((env = getenv("GPX_RESIDENT"))) h->resident = std::min(std::max(atoi(env), 0), 3);
  *out = h;
  return GPX_OK;
}

extern "C" int gpx_destroy(gpx_handle* h) {
  if (!h) return GPX_OK;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  gpx_dist_free(h);
  void* bufs[] = {h->d_x, h->d_xnew, h->d_r, h->d_mnew, h->d_hyp, h->d_noise, h->d_A, h->d_L, h->d_W,
                  h->d_S, h->d_cov, h->d_z, h->d_out, h->d_g1, h->d_g2,
                  h->d_gout, h->d_vec, h->d_A2, h->d_L2, h->d_W2, h->d_kinv, h->d_gpart, h->d_mini, h->d_eps, h->d_smp};
  for (void* b : bufs) if (b) (void)hipFree(b);
  for (double* b : h->d_aug) if (b) (void)hipFree(b);
  if (h->h_pinned) (void)hipHostFree(h->h_pinned);
  if (h->h_up) (void)hipHostFree(h->h_up);
  if (h->h_mirror) (void)hipHostFree(h->h_mirror);
  for (auto e : h->up_ev) if (e) (void)hipEventDestroy(e);
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

static int* option_slot(gpx_handle* h, const char* name) {
  if (!strcmp(name, "lookahead")) return &h->lookahead;
  if (!strcmp(name, "resident")) return &h->resident;
  if (!strcmp(name, "pair_min")) return &h->pair_min;
  if (!strcmp(name, "quad_min")) return &h->quad_min;
  if (!strcmp(name, "split_min_blocks")) return &h->split_min_blocks;
  if (!strcmp(name, "la_rows")) return &h->la_rows;
  if (!strcmp(name, "batch_group")) return &h->batch_group;
  if (!strcmp(name, "bulk_first_col")) return &h->bulk_first_col;
  if (!strcmp(name, "debug_force_timeout")) return &h->debug_force_timeout;
  return nullptr;
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
  int* slot = option_slot(h, name);
  if (!slot) { gpx_set_error(h, std::string("unknown option ") + name); return GPX_ERR_ARG; }
  int64_t v = std::max<int64_t>(value, 0);
  if (!strcmp(name, "lookahead") || !strcmp(name, "debug_force_timeout")) v = v != 0;
  if (!strcmp(name, "resident")) v = std::min<int64_t>(v, 3);
  if (!strcmp(name, "split_min_blocks")) v = std::max<int64_t>(v, 2);
  *slot = (int)v;
  return GPX_OK;
}

extern "C" int gpx_get_option(gpx_handle* h, const char* name, int64_t* value) {
  if (!h || !name || !value) return GPX_ERR_ARG;
  if (!strcmp(name, "nb")) { *value = h->nb; return GPX_OK; }
  const int* slot = option_slot(h, name);
  if (!slot) { gpx_set_error(h, std::string("unknown option ") + name); return GPX_ERR_ARG; }
  *value = *slot;
  return GPX_OK;
}

// ------------------------------------------------------------------ hyper-parameters

// Validate the keyword blocks of `batch` draws against (n rows, d dims) and upload, per draw, length[d], period[d] and
// the scalars (and the per-row noise vector when n_noise == n) to the device.
static double kernel_diag_value(int kind, double k_scale);

// Device layout of d_hyp, per draw (stride s_hyp = 2 d + 4): length[d], period[d], then {k_scale, noise_scalar,
// k_jitter, prior_diag}  (prior_diag = k(x*,x*) + new_noise + jitter, the diagonal of k_new, gpax/gp.py:287).
// All draws of a batch share the kernel kind, apply_noise and the form of the noise (scalar or per-row vector).
static int upload_params_batch(gpx_handle* h, hipStream_t s, const gpx_kernel_params* ps, int batch, int64_t n, int d,
                               const double** noise_dev, const double* new_noise) {
  if (!ps || batch < 1) { gpx_set_error(h, "params is NULL"); return GPX_ERR_ARG; }
  const int64_t sh = 2 * (int64_t)d + 4;
  std::vector<double> hyp((size_t)(sh * batch));
  *noise_dev = nullptr;
  for (int b = 0; b < batch; ++b) {
    const gpx_kernel_params* p = ps + b;
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
    if (p->kind != ps->kind || p->apply_noise != ps->apply_noise || (p->n_noise > 1) != (ps->n_noise > 1)) {
      gpx_set_error(h, "the draws of a batch must share kind, apply_noise and the form of k_noise"); return GPX_ERR_ARG;
    }
    double* hp = hyp.data() + sh * b;
    for (int k = 0; k < d; ++k) {
      hp[k] = p->k_length[p->n_length == 1 ? 0 : k];
      hp[d + k] = (p->kind == GPX_PERIODIC) ? p->k_period[p->n_period == 1 ? 0 : k] : 1.0;
    }
    hp[2 * d + 0] = p->k_scale;
    hp[2 * d + 1] = (p->n_noise == 1) ? p->k_noise[0] : 0.0;
    hp[2 * d + 2] = p->k_jitter;
    hp[2 * d + 3] = kernel_diag_value(p->kind, p->k_scale) + ((new_noise ? new_noise[b] : 0.0) + p->k_jitter);
  }
  h->s_hyp = sh;
  GPX_RET(ensure(h, &h->d_hyp, &h->hyp_cap, sh * batch));
  // single draw, and the caller's first launch can carry the values in its kernel arguments (and write d_hyp for the rest): no
  // host-to-device copy + event in front of that launch (~12 us at the head of every step)
  // (up to 1024 points: reading them from the kernel-argument segment costs the gram kernel ~1 us at its start, which the large
  // builds -- whose HBM rate is what is judged -- do not get back from a step hundreds of times longer than the copy)
  const bool defer = h->hyp_defer && batch == 1 && sh <= GPX_HYP_INLINE && n <= 1024;
  h->hyp_defer = false;
  h->hyp_host_n = 0;
  if (defer) { memcpy(h->hyp_host, hyp.data(), (size_t)sh * sizeof(double)); h->hyp_host_n = (int)sh; }
  else GPX_RET(upload_small(h, s, h->d_hyp, hyp.data(), (int64_t)hyp.size()));
  h->s_noise = 0;
  if (ps->n_noise > 1) {
    GPX_RET(ensure(h, &h->d_noise, &h->noise_cap, n * batch));
    for (int b = 0; b < batch; ++b)
      GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_noise + n * b, ps[b].k_noise, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    h->s_noise = n;
    *noise_dev = h->d_noise;
  }
  return GPX_OK;
}

static int upload_params(gpx_handle* h, hipStream_t s, const gpx_kernel_params* p, int64_t n, int d,
                         const double** noise_dev, double* noise_scalar, double new_noise = 0.0) {
  if (!p) { gpx_set_error(h, "params is NULL"); return GPX_ERR_ARG; }
  GPX_RET(upload_params_batch(h, s, p, 1, n, d, noise_dev, &new_noise));
  *noise_scalar = (p->n_noise == 1) ? p->k_noise[0] : 0.0;
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
  h->have_factor = false; h->have_z = false; h->have_r = false; h->m_resident = 0;
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
  h->m_resident = 0;   // whatever (mean, var) the result block holds belongs to the previous test inputs
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

// ------------------------------------------------------------- bordered factorisation (drivers: schedule.hip)

// ---- the bordered factorisation in three parts: (1) sizes + allocations, (2) hyper-parameter upload, (3) pure enqueue.
struct FactorPlan {
  int kind; int apply_noise;
  int64_t m_rows; bool with_r;
  const double* noise_dev;   // per-row noise vector(s) on the device, or nullptr
  bool want_k;               // full (not lower-only) K build for a K_out copy
};

static int factor_prepare(gpx_handle* h, int64_t m_rows, bool with_r, int batch = 1) {
  if (h->n <= 0 || !h->d_x) { gpx_set_error(h, "no training inputs: call gpx_set_train first"); return GPX_ERR_STATE; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  const int64_t n = h->n, nb = h->nb;
  const int64_t npad = round_up(n, nb);
  const int64_t E = m_rows + (with_r ? 1 : 0);
  const int64_t epad = E > 0 ? round_up(E, 64) : 0;
  const int64_t rows = npad + epad, ld = npad;
  h->batch = batch;
  h->s_mat = rows * ld; h->s_w = npad * nb; h->s_z = npad;
  GPX_RET(ensure(h, &h->d_A, &h->a_cap, h->s_mat * batch));
  GPX_RET(ensure(h, &h->d_L, &h->l_cap, h->s_mat * batch));
  GPX_RET(ensure(h, &h->d_W, &h->w_cap, h->s_w * batch));
  GPX_RET(ensure(h, &h->d_z, &h->z_cap, h->s_z * batch));
  GPX_RET(ensure_out(h, 2 * std::max<int64_t>(m_rows, 1), batch));
  GPX_RET(ensure_events(h, (size_t)(npad / nb), (int)nb, batch));
  h->have_factor = false; h->have_z = false;
  h->m_resident = 0;   // the result block is about to be rewritten (gpx_posterior / gpx_fit_predict_async set it again)
  h->npad = npad; h->epad = epad; h->ld = ld;
  return GPX_OK;
}

// Build [K ; k_cross ; r] in d_A (for every draw of the batch), factorise, leave [L ; V^T ; z^T] in d_L.  Enqueue only
// (kernel launches, memsets, device-to-device copies, stream events).
static int factor_enqueue(gpx_handle* h, const FactorPlan& fp, double* K_out) {
  const int64_t n = h->n, npad = h->npad, ld = h->ld; const int d = h->d;
  const int64_t m_rows = fp.m_rows;
  const int64_t E = m_rows + (fp.with_r ? 1 : 0);
  const int64_t rows = npad + h->epad;
  const int B = h->batch;
  hipStream_t s = h->s_main;
  const double* dev_scalars = h->d_hyp + 2 * d;
  const int64_t s_info = 2 * h->s_out;

  // K (diagonal rule on for k_train: gpax/gp.py:293 calls f(x, x, k_noise, k_jitter))
  GramLaunch g;
  g.kind = fp.kind; g.d = d; g.k_scale = 0.0; g.hyp = h->d_hyp; g.dev_scalars = dev_scalars;
  g.x1 = h->d_x; g.x2 = h->d_x; g.n1 = n; g.n2 = n; g.out = h->d_A; g.ld = ld;
  g.lower_only = fp.want_k ? 0 : 1; g.diag_rule = fp.apply_noise ? 1 : 0;
  g.noise = fp.noise_dev; g.noise_scalar = 0.0; g.jitter = 0.0;
  g.batch = B; g.s_out = h->s_mat; g.s_hyp = h->s_hyp; g.s_noise = h->s_noise; g.s_info = s_info; g.s_flags = h->s_flags;
  auto gram_cost = [&](int64_t c0, int64_t c1, double* fl, double* by) {   // lower tiles of the column slice [c0, c1)
    double entries = 0.0;
    for (int64_t c = c0 / 64; c * 64 < c1; ++c) entries += (double)((n + 63) / 64 - c) * 4096.0;
    if (fp.want_k) entries = (double)n * (double)(c1 - c0);
    *fl = B * gram_flops_per_entry(fp.kind, d) * entries; *by = B * (8.0 * entries + 8.0 * (double)(n + c1 - c0) * d);
  };
  auto cross_gram = [&](hipStream_t st) -> int {
    if (m_rows <= 0) return GPX_OK;
    GramLaunch c = g;
    c.x1 = h->d_xnew; c.n1 = m_rows; c.out = h->d_A + npad * ld; c.lower_only = 0; c.diag_rule = 0;
    c.noise = nullptr; c.zero_info = nullptr; c.zero_flags = nullptr; c.col0 = 0; c.x2 = h->d_x; c.n2 = n;
    c.start_event = c.stop_event = nullptr;
    prof_begin(h, st, GPX_STAGE_GRAM_CROSS, B * gram_flops_per_entry(fp.kind, d) * m_rows * n,
               B * (8.0 * m_rows * n + 8.0 * (m_rows + n) * d));
    int rc = launch_gram(h, st, c);
    prof_end(h, st);
    return rc;
  };
  const bool split = h->lookahead != 0 && npad / h->nb >= h->split_min_blocks;   // factor_loop's dispatch
  double fl, by;
  if (npad <= 256 && !split && !fp.want_k && !K_out) {
    // one or two block columns on one stream: the whole bordered matrix in ONE launch (gram.hip) -- at these sizes the
    // step is launch latency
    gram_cost(0, n, &fl, &by);
    prof_begin(h, s, GPX_STAGE_GRAM, fl, by);
    const int hin = B == 1 ? h->hyp_host_n : 0;
    h->hyp_host_n = 0;
    GPX_RET(launch_build_small(h, s, fp.kind, h->d_x, h->d_xnew, h->d_hyp, dev_scalars, fp.noise_dev, h->d_r, h->d_A, ld, n, npad,
                               m_rows, fp.with_r ? npad + m_rows : -1, rows, d, fp.apply_noise ? 1 : 0, h->d_info, nullptr, 0,
                               -1, nullptr, nullptr, B, h->s_mat, h->s_hyp, h->s_noise, h->s_r, s_info, hin ? h->hyp_host : nullptr, hin));
    prof_end(h, s);
    const int lrc = factor_loop(h, primary_mats(h, rows));
    h->flags_clean = false;
    GPX_RET(lrc);
    return GPX_OK;
  }
  g.zero_info = h->d_info;
  if (h->d_flags) {   // the hand-over flags of the split-panel schedule are cleared by the same launch
    g.zero_flags = h->d_flags; g.n_zero_flags = (int)(GPX_N_FLAG_ARRAYS * (npad / h->nb));
    h->flags_clean = true;
  }
  gram_cost(0, n, &fl, &by);
  (void)prof_pair(h, GPX_STAGE_GRAM, fl, by, &g.start_event, &g.stop_event);
  if (B == 1 && h->hyp_host_n > 0) { g.hyp_inline = h->hyp_host; g.hyp_inline_n = h->hyp_host_n; }   // the step's first launch (see upload_params_batch)
  h->hyp_host_n = 0;
  GPX_RET(launch_gram(h, s, g));
  g.hyp_inline = nullptr; g.hyp_inline_n = 0;   // (g is the template of the launches below: they read d_hyp)
  if (K_out) {
    GPX_HIP_CHECK(h, hipMemcpy2DAsync(K_out, (size_t)n * sizeof(double), h->d_A, (size_t)ld * sizeof(double),
                                      (size_t)n * sizeof(double), (size_t)n, hipMemcpyDeviceToHost, s));
  }
  GPX_RET(launch_fill_identity_pad(h, s, h->d_A, ld, n, npad, rows, B, h->s_mat));
  // The border (k_cross rows, the r row, the zero rows) is not read by the chain before the very last panel: with the
  // split-panel schedule it is built on the panel stream, whose first kernel (the panel of block column 0) is its
  // first reader, while the chain stream goes straight from K to potrf_diag(0) -- ~30 us off the head of the step.
  // (Building most of K itself on the bulk stream while the chain already factorises was tried: no gain at N = 4096,
  // 1.907 vs 1.905 ms -- the first steps are bound by that stream.)
  hipStream_t sb = s;
  if (split && h->prof_mode != 1) {
    sb = h->s_far;
    GPX_HIP_CHECK(h, hipEventRecord(h->ev_build, s));
    GPX_HIP_CHECK(h, hipStreamWaitEvent(sb, h->ev_build, 0));
  }
  GPX_RET(cross_gram(sb));
  // the r row and the unused (zero) border rows in one launch
  GPX_RET(launch_border_rows(h, sb, h->d_A, ld, npad, n, fp.with_r ? npad + m_rows : -1, h->d_r, npad + E, rows, nullptr, B,
                             h->s_mat, h->s_r));

  const int lrc = factor_loop(h, primary_mats(h, rows));
  h->flags_clean = false;   // consumed by the split-panel loop, or unused by the single-stream one: never carried into another call
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
  h->hyp_defer = true;   // factor_enqueue's first launch carries the hyper-parameters
  GPX_RET(upload_params(h, h->s_main, p, h->n, h->d, &fp.noise_dev, &noise_scalar, new_noise));
  fp.kind = p->kind; fp.apply_noise = p->apply_noise; fp.m_rows = m_rows; fp.with_r = with_r; fp.want_k = K_out != nullptr;
  GPX_RET(factor_enqueue(h, fp, K_out));
  factor_commit(h, p, m_rows, with_r);
  return GPX_OK;
}

// info == -7: a resident kernel gave up waiting for its producer (potrf.hip / gemm.hip: bounded spin).  That only
// happens when something keeps the chain's kernels from running concurrently (a profiler collecting counters, a
// debugger, other work holding the CUs): the handle switches to kernel-boundary hand-overs for good and the call is
// repeated -- by the synchronous entry points themselves (with_retry), by the caller of the asynchronous pair
// (gpx_fetch_results returns GPX_ERR_RETRY; gpax_amd._lib.Core re-issues the step).
static int check_handover(gpx_handle* h, int info) {
  if (info == -8) {
    h->have_factor = false; h->have_z = false;
    gpx_set_error(h, "potrf_diag: a wait between its waves ran out (internal error); the results are undefined");
    return GPX_ERR_INTERNAL;
  }
  if (info != -7) return GPX_OK;
  h->resident = 0;
  h->have_factor = false; h->have_z = false;
  gpx_set_error(h, "in-kernel hand-over timed out (are kernel dispatches being serialised, e.g. by counter collection?); "
                   "switched to kernel-boundary hand-overs (option \"resident\" = 0): repeat the call");
  return GPX_ERR_RETRY;
}

template <typename F>
static int with_retry(F&& call) {
  int rc = call();
  if (rc == GPX_ERR_RETRY) rc = call();   // "resident" is 0 now: no in-kernel wait is left to time out
  return rc;
}

static int read_info(gpx_handle* h, int* info) {
  const double* blk;
  GPX_RET(fetch_block(h, 0, &blk));
  *info = reinterpret_cast<const int*>(blk + 8)[0];
  return check_handover(h, *info);
}

static int fit_factor_once(gpx_handle* h, const gpx_kernel_params* p, double* K_out, double* L_out) {
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

extern "C" int gpx_fit_factor(gpx_handle* h, const gpx_kernel_params* p, double* K_out, double* L_out) {
  if (!h) return GPX_ERR_ARG;
  return with_retry([&] { return fit_factor_once(h, p, K_out, L_out); });
}

// ------------------------------------------------------- solves against the cached factor

// X L^T = S for `srows` right-hand-side ROWS (S, X: srows x npad, row-major): right-looking,
//   X_k = S_k W_k^T ;  S_j -= X_k L_jk^T  (j > k)   -- GEMMs only.
// X = S L^-T for `srows` right-hand-side rows against the cached factor (the cho_solve of gpax/gp.py:297-299 as GEMMs against
// W_k = L_kk^-1): right-looking over block columns, the columns behind a GROUP of four or eight updated in one K = G nb pass (inside
// the group every solved column updates the group's remaining ones at K = nb) -- the bulk kernel runs K >= 512 at 55-65 instead
// of 44 TFLOP/s and the right-hand sides are read and written once per G panels (round 3; M = 10 000 rows against
// N = 8192: profiles/r03_posterior_cached_factor.txt).
static int solve_rows(gpx_handle* h, double* S, double* X, int64_t srows) {
  const int64_t nb = h->nb, npad = h->npad, ld = h->ld, nblk = npad / nb;
  static const int64_t g_env = getenv("GPX_SOLVE_GROUP") ? std::max(1, atoi(getenv("GPX_SOLVE_GROUP"))) : 0;   // diagnostic (1: round 2's column-by-column form)
  const int64_t G = g_env ? g_env : (nblk >= 32 ? 8 : 4);   // eight panels per pass from N = 4096 on (measured: -2..3 % against four; +2 % at N = 2048)
  hipStream_t s = h->s_main;
  auto update = [&](int64_t kfirst, int64_t npan, int64_t cfirst, int64_t ncolblk) -> int {   // S[:, cols) -= X[:, panels) L[cols, panels)^T
    GemmLaunch u;
    u.A = X + kfirst * nb; u.lda = npad;
    u.B = h->d_L + cfirst * nb * ld + kfirst * nb; u.ldb = ld;
    u.C = S + cfirst * nb; u.ldc = npad;
    u.M = srows; u.N = ncolblk * nb; u.K = npan * nb; u.alpha = -1.0; u.beta = 1.0;
    u.lower = 0; u.row_off = 0; u.col_off = 0; u.b_lower_tri = 0; u.small_tile = srows <= 64 ? 1 : 0;
    return launch_gemm_nt(h, s, u);
  };
  for (int64_t g0 = 0; g0 < nblk; g0 += G) {
    const int64_t gend = std::min(nblk, g0 + G);
    for (int64_t k = g0; k < gend; ++k) {
      GemmLaunch g;
      g.A = S + k * nb; g.lda = npad;
      g.B = h->d_W + k * nb * nb; g.ldb = nb;
      g.C = X + k * nb; g.ldc = npad;
      g.M = srows; g.N = nb; g.K = nb; g.alpha = 1.0; g.beta = 0.0;
      g.lower = 0; g.row_off = 0; g.col_off = 0; g.b_lower_tri = 1; g.small_tile = 1;
      GPX_RET(launch_gemm_nt(h, s, g));
      if (k + 1 < gend) GPX_RET(update(k, 1, k + 1, gend - (k + 1)));
    }
    if (gend < nblk) GPX_RET(update(g0, gend - g0, gend, nblk - gend));
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

// grad_out layout: [k_scale | k_length (n_length) | k_period (n_period, periodic only) | k_noise], from the reduction's sums
static int grads_from_sums(const gpx_kernel_params* p, const double* sums, int d, int per_dim, int info, double* grad_out) {
  const int DCH = grad_chunk_dims(), NPVv = grad_partials_per_chunk();
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
  return o;
}

// One bordered factorisation of [K ; r ; I] -> [L ; z^T ; U = L^-T] (the identity rows are activated
// block by block, factor_loop's tri_base), then alpha = U z, K^-1 = U U^T (lower tiles, dead-K skipped)
// and ONE pass over K^-1 that recomputes dK/dtheta from x (grad.hip).  ~N^3 flops in total against
// N^3/3 for the factorisation alone -- versus 2*p+1 factorisations for central differences.
static int lml_grad_once(gpx_handle* h, const gpx_kernel_params* p, const double* y_minus_m, double* logp,
                         double* grad_out, double* alpha_out) {
  if (!p || !y_minus_m) { gpx_set_error(h, "gpx_lml_grad: params / y_minus_m is NULL"); return GPX_ERR_ARG; }
  if (h->n <= 0 || !h->d_x) { gpx_set_error(h, "no training inputs: call gpx_set_train first"); return GPX_ERR_STATE; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  const int64_t n = h->n, nb = h->nb; const int d = h->d;
  const int64_t npad = round_up(n, nb), ld = npad;
  const int64_t RZ = 64;                              // the r row, padded to one GEMM tile
  const int64_t epad = RZ + round_up(n, 64);          // then the identity rows
  const int64_t rows = npad + epad, n64 = round_up(n, 64);
  hipStream_t s = h->s_main;

  h->have_factor = false; h->have_z = false; h->m_resident = 0;   // the result block is reused for alpha and the sums
  h->batch = 1; h->s_mat = rows * ld; h->s_w = npad * nb; h->s_z = npad; h->s_r = 0; h->s_mnew = 0;
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
  GPX_RET(ensure_mirror(h, GPX_OUT_HEAD + npad + (int64_t)nchunk * NPVv));
  GPX_RET(ensure_events(h, (size_t)(npad / nb), (int)nb, 1));
  h->npad = npad; h->epad = epad; h->ld = ld;

  const double* noise_dev; double noise_scalar;
  h->hyp_defer = true;   // the first launch below (one-launch build or gram) carries the hyper-parameters
  GPX_RET(upload_params(h, s, p, n, d, &noise_dev, &noise_scalar));
  const int hin = h->hyp_host_n;
  h->hyp_host_n = 0;
  // small problems: the one-launch build below reads r straight from a pinned slot (and writes d_r on the way): no upload in the stream
  const bool split = h->lookahead != 0 && npad / nb >= h->split_min_blocks;   // factor_loop's dispatch
  const bool r_in_place = npad <= 256 && !split;
  const double* r_src = h->d_r; int r_slot = -1;
  if (r_in_place) GPX_RET(stage_small(h, y_minus_m, n, &r_src, &r_slot));
  else if (n <= 65536) GPX_RET(upload_small(h, s, h->d_r, y_minus_m, n));   // through the pinned ring: truly asynchronous
  else GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_r, y_minus_m, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
  h->have_r = true;

  // ---- [K ; r ; I] in d_A
  const bool one_launch = r_in_place;   // one stream, small: one launch builds everything below (gram.hip)
  if (one_launch) {
    const double T = (double)((n + 63) / 64);
    const double entries = T * (T + 1) / 2.0 * 4096.0;
    prof_begin(h, s, GPX_STAGE_GRAM, gram_flops_per_entry(p->kind, d) * entries, 8.0 * entries + 16.0 * n * d);
    GPX_RET(launch_build_small(h, s, p->kind, h->d_x, nullptr, h->d_hyp, h->d_hyp + 2 * d, noise_dev, r_src, h->d_A, ld, n, npad, 0,
                               npad, rows, d, p->apply_noise ? 1 : 0, h->d_info, nullptr, 0, npad + RZ, h->d_L + npad * ld,
                               r_slot >= 0 ? h->d_r : nullptr, 1, 0, 0, 0, 0, 0, hin ? h->hyp_host : nullptr, hin));
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
    if (hin) { g.hyp_inline = h->hyp_host; g.hyp_inline_n = hin; }
    GPX_RET(launch_gram(h, s, g));
    g.hyp_inline = nullptr; g.hyp_inline_n = 0;
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
  // K^-1 = U U^T (lower 64x64 tiles) is accumulated panel by panel inside the split-panel schedule; the
  // single-stream schedule forms it in one product after the factorisation
  const bool fused_kinv = split;
  if (fused_kinv) { h->kinv_U = U; h->kinv_rows = n64; }
  int frc = factor_loop(h, primary_mats(h, rows), npad + RZ);
  h->kinv_U = nullptr; h->kinv_rows = 0;
  GPX_RET(frc);

  EpilogueLaunch e;
  e.L = h->d_L; e.ld = ld; e.n = n; e.z = zrow; e.vt = nullptr; e.vt_ld = 0; e.m = 0; e.m_new = nullptr;
  e.prior_diag = 0.0; e.scal = h->d_scal; e.mean = nullptr; e.var = nullptr;
  e.z_copy = h->d_z; e.npad = npad;
  // ---- alpha = L^-T z = U z rides on the epilogue's launch (extra blocks; it reads z from the border row itself)
  double* alpha = h->d_mean;   // inside the result block
  e.gemv_U = U; e.gemv_alpha = alpha; e.gemv_ncols = npad;
  e.mirror = h->h_mirror; e.info_src = h->d_info;   // scal, info and alpha also go to the pinned mirror (nothing behind this launch writes info)
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
                             sums_dev, 1, 0, 0, 0, h->h_mirror + GPX_OUT_HEAD + npad));
  h->mirror_valid = true;

  const double* blk;
  GPX_RET(fetch_block(h, npad + (int64_t)nchunk * NPVv, &blk, 1, true));
  const double scal[2] = {blk[0], blk[1]};
  const int info = reinterpret_cast<const int*>(blk + 8)[0];
  GPX_RET(check_handover(h, info));
  const double* sums = blk + GPX_OUT_HEAD + npad;
  if (alpha_out) memcpy(alpha_out, blk + GPX_OUT_HEAD, (size_t)n * sizeof(double));
  factor_commit(h, p, 0, true);
  if (logp) *logp = -0.5 * scal[1] - scal[0] - 0.5 * (double)n * log(2.0 * M_PI);

  if (grad_out) grads_from_sums(p, sums, d, per_dim, info, grad_out);
  return info;
}

extern "C" int gpx_lml_grad(gpx_handle* h, const gpx_kernel_params* p, const double* y_minus_m, double* logp,
                            double* grad_out, double* alpha_out) {
  if (!h) return GPX_ERR_ARG;
  return with_retry([&] { return lml_grad_once(h, p, y_minus_m, logp, grad_out, alpha_out); });
}

static void batch_invalidate(gpx_handle* h);

// B gradient evaluations in ONE schedule: what jax.vmap / chain_method="vectorized" make of the NUTS potential in the reference
// (gpax/gp.py:507-509, 517-525).  Every kernel of lml_grad_once is launched once with the draw index in its grid: per draw the
// bordered matrix [K ; r ; I], the factorisation with K^-1 accumulated on the bulk stream, the epilogue (+ alpha) and the
// reduction.  Same arithmetic per draw as the single-draw call with the same grouping thresholds.
static int lml_grad_batch_once(gpx_handle* h, int batch, const gpx_kernel_params* params, const double* y_minus_m, double* logp,
                               double* grad_out, double* alpha_out, int32_t* info_out) {
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  const gpx_kernel_params* p = params;
  const int64_t n = h->n, nb = h->nb; const int d = h->d;
  const int64_t npad = round_up(n, nb), ld = npad;
  const int64_t RZ = 64;
  const int64_t epad = RZ + round_up(n, 64);
  const int64_t rows = npad + epad, n64 = round_up(n, 64);
  hipStream_t s = h->s_main;
  for (int b = 1; b < batch; ++b)
    if (params[b].n_length != p->n_length || params[b].n_period != p->n_period) {
      gpx_set_error(h, "gpx_lml_grad_batch: the draws must share the form of k_length / k_period"); return GPX_ERR_ARG;
    }

  h->have_factor = false; h->have_z = false; h->m_resident = 0;
  h->batch = batch; h->s_mat = rows * ld; h->s_w = npad * nb; h->s_z = npad; h->s_r = n; h->s_mnew = 0; h->s_kinv = n64 * npad;
  GPX_RET(ensure(h, &h->d_A, &h->a_cap, h->s_mat * batch));
  GPX_RET(ensure(h, &h->d_L, &h->l_cap, h->s_mat * batch));
  GPX_RET(ensure(h, &h->d_W, &h->w_cap, h->s_w * batch));
  GPX_RET(ensure(h, &h->d_z, &h->z_cap, h->s_z * batch));
  GPX_RET(ensure(h, &h->d_r, &h->r_cap, n * batch));
  GPX_RET(ensure(h, &h->d_kinv, &h->kinv_cap, h->s_kinv * batch));
  const int per_dim = (p->n_length > 1 || (p->kind == GPX_PERIODIC && p->n_period > 1)) ? 1 : 0;
  const int DCH = grad_chunk_dims(), NPVv = grad_partials_per_chunk();
  const int nchunk = per_dim ? (d + DCH - 1) / DCH : 1;
  const int64_t tail = npad + (int64_t)nchunk * NPVv;
  GPX_RET(ensure_out(h, tail, batch));
  GPX_RET(ensure_events(h, (size_t)(npad / nb), (int)nb, batch));
  h->npad = npad; h->epad = epad; h->ld = ld;
  const int64_t s_info = 2 * h->s_out;

  const double* noise_dev;
  GPX_RET(upload_params_batch(h, s, params, batch, n, d, &noise_dev, nullptr));
  GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_r, y_minus_m, (size_t)(n * batch) * sizeof(double), hipMemcpyHostToDevice, s));

  // ---- [K ; r ; I] per draw in d_A
  GramLaunch g;
  g.kind = p->kind; g.d = d; g.k_scale = 0.0; g.hyp = h->d_hyp; g.dev_scalars = h->d_hyp + 2 * d;
  g.x1 = h->d_x; g.x2 = h->d_x; g.n1 = n; g.n2 = n; g.out = h->d_A; g.ld = ld;
  g.lower_only = 1; g.diag_rule = p->apply_noise ? 1 : 0;
  g.noise = noise_dev; g.noise_scalar = 0.0; g.jitter = 0.0;
  g.zero_info = h->d_info;
  g.batch = batch; g.s_out = h->s_mat; g.s_hyp = h->s_hyp; g.s_noise = h->s_noise; g.s_info = s_info; g.s_flags = h->s_flags;
  {
    const double T = (double)((n + 63) / 64);
    const double entries = T * (T + 1) / 2.0 * 4096.0;
    prof_begin(h, s, GPX_STAGE_GRAM, batch * gram_flops_per_entry(p->kind, d) * entries, batch * (8.0 * entries + 16.0 * n * d));
    GPX_RET(launch_gram(h, s, g));
    prof_end(h, s);
  }
  GPX_RET(launch_fill_identity_pad(h, s, h->d_A, ld, n, npad, rows, batch, h->s_mat));
  double* Ab = h->d_A + npad * ld;                       // border of draw 0
  GPX_RET(launch_border_rows(h, s, h->d_A, ld, npad, n, npad, h->d_r, npad + 1, npad + RZ, nullptr, batch, h->s_mat, n));
  GPX_RET(launch_fill_rect(h, s, Ab + RZ * ld, ld, 0, n, 0, n, 1.0, batch, h->s_mat));          // identity rows
  GPX_RET(launch_fill_rect(h, s, Ab + RZ * ld, ld, n, epad - RZ, 0, n, 0.0, batch, h->s_mat));
  GPX_RET(launch_fill_rect(h, s, h->d_L + npad * ld, ld, 0, epad, 0, npad, 0.0, batch, h->s_mat));   // the dead lower-left of U reads as zeros

  const double* zrow = h->d_L + npad * ld;
  const double* U = h->d_L + (npad + RZ) * ld;
  const bool split = h->lookahead != 0 && npad / nb >= h->split_min_blocks;
  if (split) { h->kinv_U = U; h->kinv_rows = n64; }
  int frc = factor_loop(h, primary_mats(h, rows), npad + RZ);
  h->kinv_U = nullptr; h->kinv_rows = 0;
  if (frc != GPX_OK) { batch_invalidate(h); return frc; }

  EpilogueLaunch e;
  e.L = h->d_L; e.ld = ld; e.n = n; e.z = zrow; e.vt = nullptr; e.vt_ld = 0; e.m = 0; e.m_new = nullptr;
  e.prior_diag = 0.0; e.scal = h->d_scal; e.mean = nullptr; e.var = nullptr;
  e.z_copy = h->d_z; e.npad = npad;
  double* alpha = h->d_mean;
  e.gemv_U = U; e.gemv_alpha = alpha; e.gemv_ncols = npad;
  e.batch = batch; e.s_mat = h->s_mat; e.s_out = h->s_out; e.s_zcopy = h->s_z;
  GPX_RET(launch_epilogue(h, s, e));

  if (!split) {
    GemmLaunch u;
    u.A = U; u.lda = ld; u.B = U; u.ldb = ld; u.C = h->d_kinv; u.ldc = npad;
    u.M = n64; u.N = n64; u.K = npad; u.alpha = 1.0; u.beta = 0.0;
    u.lower = 1; u.row_off = 0; u.col_off = 0; u.b_lower_tri = 0; u.small_tile = 0; u.a_upper_tri = 1;
    u.batch = batch; u.sA = h->s_mat; u.sB = h->s_mat; u.sC = h->s_kinv;
    GPX_RET(launch_gemm_nt(h, s, u));
  }

  const int64_t T = (n + 63) / 64, ntiles = T * (T + 1) / 2;
  const int nwg = ntiles <= 3 ? 1 : (int)std::min<int64_t>(ntiles, 1024);
  GPX_RET(ensure(h, &h->d_gpart, &h->gpart_cap, (int64_t)nchunk * nwg * NPVv * batch));
  double* sums_dev = h->d_mean + npad;
  GPX_RET(launch_grad_reduce(h, s, p->kind, h->d_x, h->d_hyp, h->d_kinv, npad, alpha, n, d, per_dim, nwg, h->d_gpart,
                             sums_dev, batch, h->s_hyp, h->s_kinv, h->s_out));

  const double* blk = nullptr;
  const int rc = fetch_block(h, tail, &blk, batch);
  const int64_t so = h->s_out;
  batch_invalidate(h);
  GPX_RET(rc);
  const int ng = 2 + p->n_length + (p->kind == GPX_PERIODIC ? p->n_period : 0);
  int worst = 0;
  for (int b = 0; b < batch; ++b) {
    const double* q = blk + so * b;
    const int info = reinterpret_cast<const int*>(q + 8)[0];
    GPX_RET(check_handover(h, info));
    if (info_out) info_out[b] = info;
    if (info > worst) worst = info;
    if (logp) logp[b] = -0.5 * q[1] - q[0] - 0.5 * (double)n * log(2.0 * M_PI);
    if (alpha_out) memcpy(alpha_out + n * b, q + GPX_OUT_HEAD, (size_t)n * sizeof(double));
    if (grad_out) grads_from_sums(params + b, q + GPX_OUT_HEAD + npad, d, per_dim, info, grad_out + (int64_t)ng * b);
  }
  return worst;
}

extern "C" int gpx_lml_grad_batch(gpx_handle* h, int32_t batch, const gpx_kernel_params* params, const double* y_minus_m,
                                  double* logp, double* grad_out, double* alpha_out, int32_t* info_out) {
  if (!h) return GPX_ERR_ARG;
  if (batch < 1 || batch > 4096 || !params || !y_minus_m) { gpx_set_error(h, "gpx_lml_grad_batch: bad argument"); return GPX_ERR_ARG; }
  if (h->n <= 0 || !h->d_x) { gpx_set_error(h, "gpx_lml_grad_batch: call gpx_set_train first"); return GPX_ERR_STATE; }
  return with_retry([&] { return lml_grad_batch_once(h, batch, params, y_minus_m, logp, grad_out, alpha_out, info_out); });
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
  h->m_resident = m;
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

// Cholesky of `batch` symmetric (m x m) matrices sitting in d_A2 (lower triangles, ld = npad, identity padding in place)
// through the same blocked factorisation on the secondary buffers (the cached GP factor stays valid), then
// Y = E L_c^T for `spad` rows of standard-normal draws per matrix (E in d_eps, Y in d_smp, both (spad, npad) per draw).
static int sample_factor_and_draw(gpx_handle* h, int64_t m, int64_t nb, int64_t npad, int64_t spad, int batch) {
  hipStream_t s = h->s_main;
  Mats M2;
  M2.A = h->d_A2; M2.L = h->d_L2; M2.W = h->d_W2; M2.npad = npad; M2.ld = npad; M2.rows_total = npad;
  M2.nb = (int)nb; M2.batch = batch; M2.sA = npad * npad; M2.sW = npad * nb;
  GPX_RET(factor_loop(h, M2));
  GemmLaunch g;
  g.A = h->d_eps; g.lda = npad; g.B = h->d_L2; g.ldb = npad; g.C = h->d_smp; g.ldc = npad;
  g.M = spad; g.N = npad; g.K = npad; g.alpha = 1.0; g.beta = 0.0;
  g.lower = 0; g.row_off = 0; g.col_off = 0; g.b_lower_tri = 1; g.small_tile = 0;
  g.batch = batch; g.sA = spad * npad; g.sB = npad * npad; g.sC = spad * npad;
  (void)m;
  return launch_gemm_nt(h, s, g);
}

static int sample_buffers(gpx_handle* h, int64_t nb, int64_t npad, int64_t spad, int batch) {
  GPX_RET(ensure(h, &h->d_A2, &h->a2_cap, npad * npad * batch));
  GPX_RET(ensure(h, &h->d_L2, &h->l2_cap, npad * npad * batch));
  GPX_RET(ensure(h, &h->d_W2, &h->w2_cap, npad * nb * batch));
  GPX_RET(ensure(h, &h->d_eps, &h->eps_cap, spad * npad * batch));
  GPX_RET(ensure(h, &h->d_smp, &h->smp_cap, spad * npad * batch));
  return ensure_events(h, (size_t)(npad / nb), (int)nb, batch);
}

// dist.MultivariateNormal(mean, cov).sample(key, (n,)) of gpax/gp.py:335-337:  out = mean + eps L_c^T,
// L_c = chol(cov); the draws eps ~ N(0, I) come from the caller.
static int mvn_sample_once(gpx_handle* h, const double* mean, const double* cov, int64_t m, const double* eps,
                           int64_t nsamp, double* out) {
  if (!cov || !eps || !out || m <= 0 || nsamp <= 0) { gpx_set_error(h, "gpx_mvn_sample: bad argument"); return GPX_ERR_ARG; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  hipStream_t s = h->s_main;
  const int64_t nb = h->nb_auto ? (m <= 64 ? 64 : 128) : h->nb;   // the sample covariance is its own problem size
  const int64_t npad = round_up(m, nb), spad = round_up(nsamp, 64);
  const int64_t sv_flags = h->s_flags, sv_mini = h->s_mini;
  GPX_RET(sample_buffers(h, nb, npad, spad, 1));
  GPX_HIP_CHECK(h, hipMemcpy2DAsync(h->d_A2, (size_t)npad * sizeof(double), cov, (size_t)m * sizeof(double),
                                    (size_t)m * sizeof(double), (size_t)m, hipMemcpyHostToDevice, s));
  GPX_RET(launch_fill_identity_pad(h, s, h->d_A2, npad, m, npad, npad));
  GPX_HIP_CHECK(h, hipMemsetAsync(h->d_info, 0, sizeof(int), s));
  GPX_HIP_CHECK(h, hipMemsetAsync(h->d_eps, 0, (size_t)(spad * npad) * sizeof(double), s));
  GPX_HIP_CHECK(h, hipMemcpy2DAsync(h->d_eps, (size_t)npad * sizeof(double), eps, (size_t)m * sizeof(double),
                                    (size_t)m * sizeof(double), (size_t)nsamp, hipMemcpyHostToDevice, s));
  const int rc = sample_factor_and_draw(h, m, nb, npad, spad, 1);
  h->s_flags = sv_flags; h->s_mini = sv_mini;
  GPX_RET(rc);
  GPX_HIP_CHECK(h, hipMemcpy2DAsync(out, (size_t)m * sizeof(double), h->d_smp, (size_t)npad * sizeof(double),
                                    (size_t)m * sizeof(double), (size_t)nsamp, hipMemcpyDeviceToHost, s));
  int info = 0;
  GPX_RET(read_info(h, &info));
  if (mean)
    for (int64_t r = 0; r < nsamp; ++r)
      for (int64_t c = 0; c < m; ++c) out[r * m + c] += mean[c];
  return info;
}

extern "C" int gpx_mvn_sample(gpx_handle* h, const double* mean, const double* cov, int64_t m, const double* eps,
                              int64_t nsamp, double* out) {
  if (!h) return GPX_ERR_ARG;
  return with_retry([&] { return mvn_sample_once(h, mean, cov, m, eps, nsamp, out); });
}

// ------------------------------------------------------------------ fused fit+predict

static EpilogueLaunch fit_predict_epilogue(gpx_handle* h, int64_t m) {
  EpilogueLaunch e;
  e.L = h->d_L; e.ld = h->ld; e.n = h->n; e.z = h->d_L + (h->npad + m) * h->ld;
  e.vt = h->d_L + h->npad * h->ld; e.vt_ld = h->ld; e.m = m;
  e.m_new = h->have_mnew ? h->d_mnew : nullptr;
  e.prior_diag = 0.0; e.dev_prior_diag = h->d_hyp + 2 * h->d + 3;
  e.scal = h->d_scal; e.mean = h->d_mean; e.var = h->d_var;
  e.z_copy = h->d_z; e.npad = h->npad;   // keep z for later posterior calls against the cached factor
  e.batch = h->batch; e.s_mat = h->s_mat; e.s_out = h->s_out; e.s_zcopy = h->s_z; e.s_mnew = h->s_mnew; e.s_hyp = h->s_hyp;
  e.s_part = h->s_epart;
  if (h->batch == 1 && h->h_mirror && h->mirror_cap >= GPX_OUT_HEAD + 2 * std::max<int64_t>(m, 1)) {
    e.mirror = h->h_mirror; e.info_src = h->d_info; e.mirror_var_off = h->d_var - h->d_mean;
  }
  return e;
}

static int fit_predict_enqueue(gpx_handle* h, const FactorPlan& fp) {
  const int64_t m = fp.m_rows;
  hipStream_t s = h->s_main;
  const double B = (double)h->batch;
  h->s_epart = 2 + 2 * std::max<int64_t>(m, 1);
  EpilogueLaunch e = fit_predict_epilogue(h, m);
  // split epilogue (split-panel schedule, three block columns or more): see SplitSchedule::panel (schedule.hip)
  h->epi = nullptr; h->epi_split_col = 0;
  if (h->prof_mode != 1 && h->lookahead != 0 && h->npad / h->nb >= std::max(3, h->split_min_blocks)) {
    GPX_RET(ensure(h, &h->d_epart, &h->epart_cap, h->s_epart * h->batch));
    h->epi = &e;
  }
  const int frc = factor_enqueue(h, fp, nullptr);
  h->epi = nullptr;
  GPX_RET(frc);
  prof_begin(h, s, GPX_STAGE_EPILOGUE, B * 4.0 * (m + 1.0) * h->n, B * 8.0 * (m + 2.0) * h->n);
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
  h->s_r = 0; h->s_mnew = 0;
  GPX_RET(ensure_mirror(h, GPX_OUT_HEAD + 2 * std::max<int64_t>(m, 1)));   // written by the final epilogue launch
  FactorPlan fp;
  double noise_scalar;
  h->hyp_defer = true;   // factor_enqueue's first launch carries the hyper-parameters
  GPX_RET(upload_params(h, s, p, h->n, h->d, &fp.noise_dev, &noise_scalar, new_noise));
  fp.kind = p->kind; fp.apply_noise = p->apply_noise; fp.m_rows = m; fp.with_r = true; fp.want_k = false;
  prof_begin(h, s, GPX_STAGE_FACTOR_TOTAL, 0.0, 0.0);
  GPX_RET(fit_predict_enqueue(h, fp));
  if (h->prof_mode) {  // close the FACTOR_TOTAL bracket (first pending of this call)
    for (size_t i = h->pending.size(); i-- > 0;)
      if (h->pending[i].stage == GPX_STAGE_FACTOR_TOTAL) { (void)hipEventRecord(h->pending[i].b, s); break; }
  }
  factor_commit(h, p, m, true);
  h->m_resident = m;
  h->mirror_valid = h->batch == 1;
  return GPX_OK;
}

extern "C" int gpx_fetch_results(gpx_handle* h, double* logp, double* mean_out, double* var_out) {
  if (!h) return GPX_ERR_ARG;
  if (!h->have_factor || !h->r_in_factor) { gpx_set_error(h, "gpx_fetch_results: no fit_predict results"); return GPX_ERR_STATE; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  const int64_t m = h->m_in_factor;
  // one copy: [scal | info | mean(m) | var(m)] (d_var = d_mean + max(m, 1), set by gpx_fit_predict_async)
  const double* blk;
  GPX_RET(fetch_block(h, m > 0 ? 2 * m : 0, &blk, 1, true));
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
  return with_retry([&]() -> int {
    GPX_RET(gpx_set_targets(h, y_minus_m, m_new));
    GPX_RET(gpx_fit_predict_async(h, p, new_noise));
    return gpx_fetch_results(h, logp, mean_out, var_out);
  });
}

// ---------------------------------------------------------------- batched hyper-parameter draws

// Upload the per-draw targets: r (batch, n) and, optionally, m_new (batch, m).
static int set_targets_batch(gpx_handle* h, int batch, const double* y_minus_m, const double* m_new) {
  const int64_t n = h->n, m = h->m;
  hipStream_t s = h->s_main;
  GPX_RET(ensure(h, &h->d_r, &h->r_cap, n * batch));
  GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_r, y_minus_m, (size_t)(n * batch) * sizeof(double), hipMemcpyHostToDevice, s));
  h->have_r = true; h->s_r = n;
  h->have_mnew = false; h->s_mnew = 0;
  if (m_new && m > 0) {
    GPX_RET(ensure(h, &h->d_mnew, &h->mnew_cap, m * batch));
    GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_mnew, m_new, (size_t)(m * batch) * sizeof(double), hipMemcpyHostToDevice, s));
    h->have_mnew = true; h->s_mnew = m;
  }
  return GPX_OK;
}

// After a batched call the handle's single-draw caches (factor, z, targets) describe draw 0 only in part: drop them.
static void batch_invalidate(gpx_handle* h) {
  h->have_factor = false; h->have_z = false; h->have_r = false; h->have_mnew = false;
  h->batch = 1; h->s_r = 0; h->s_mnew = 0; h->s_noise = 0; h->m_resident = 0;
}

static int fit_predict_batch_once(gpx_handle* h, int batch, const gpx_kernel_params* params, const double* y_minus_m,
                                  const double* m_new, const double* new_noise, double* logp, double* mean_out,
                                  double* var_out, int32_t* info_out) {
  const int64_t m = h->m, n = h->n;
  hipStream_t s = h->s_main;
  GPX_RET(factor_prepare(h, m, true, batch));
  h->d_var = h->d_mean + std::max<int64_t>(m, 1);
  GPX_RET(set_targets_batch(h, batch, y_minus_m, m_new));
  FactorPlan fp;
  GPX_RET(upload_params_batch(h, s, params, batch, n, h->d, &fp.noise_dev, new_noise));
  fp.kind = params->kind; fp.apply_noise = params->apply_noise; fp.m_rows = m; fp.with_r = true; fp.want_k = false;
  const int erc = fit_predict_enqueue(h, fp);
  const double* blk = nullptr;
  int frc = erc == GPX_OK ? fetch_block(h, 2 * std::max<int64_t>(m, 1), &blk, batch) : erc;
  const int64_t so = h->s_out;
  batch_invalidate(h);
  GPX_RET(frc);
  int worst = 0;
  for (int b = 0; b < batch; ++b) {
    const double* q = blk + so * b;
    const int info = reinterpret_cast<const int*>(q + 8)[0];
    GPX_RET(check_handover(h, info));
    if (info_out) info_out[b] = info;
    if (info > worst) worst = info;
    if (logp) logp[b] = -0.5 * q[1] - q[0] - 0.5 * (double)n * log(2.0 * M_PI);
    if (mean_out && m > 0) memcpy(mean_out + m * b, q + GPX_OUT_HEAD, (size_t)m * sizeof(double));
    if (var_out && m > 0) memcpy(var_out + m * b, q + GPX_OUT_HEAD + m, (size_t)m * sizeof(double));
  }
  return worst;
}

extern "C" int gpx_fit_predict_batch(gpx_handle* h, int32_t batch, const gpx_kernel_params* params, const double* y_minus_m,
                                     const double* m_new, const double* new_noise, double* logp, double* mean_out,
                                     double* var_out, int32_t* info_out) {
  if (!h) return GPX_ERR_ARG;
  if (batch < 1 || batch > 4096 || !params || !y_minus_m) { gpx_set_error(h, "gpx_fit_predict_batch: bad argument"); return GPX_ERR_ARG; }
  if (h->n <= 0) { gpx_set_error(h, "gpx_fit_predict_batch: call gpx_set_train first"); return GPX_ERR_STATE; }
  return with_retry([&] { return fit_predict_batch_once(h, batch, params, y_minus_m, m_new, new_noise, logp, mean_out, var_out, info_out); });
}

// cov = k_new - V^T V (draw b: V^T = the k_cross border rows of its factor), lower triangle, into d_A2 with identity padding
static int posterior_cov_to_A2(gpx_handle* h, int kind, int64_t m, int64_t npad2, int batch, bool conditioned) {
  hipStream_t s = h->s_main;
  const int d = h->d;
  GramLaunch c;
  c.kind = kind; c.d = d; c.k_scale = 0.0; c.hyp = h->d_hyp; c.dev_scalars = h->d_hyp + 2 * d;
  c.x1 = h->d_xnew; c.x2 = h->d_xnew; c.n1 = m; c.n2 = m; c.out = h->d_A2; c.ld = npad2;
  c.lower_only = 1; c.diag_rule = 1; c.noise = nullptr; c.noise_scalar = 0.0; c.jitter = 0.0;
  c.batch = batch; c.s_out = npad2 * npad2; c.s_hyp = h->s_hyp;
  GPX_RET(launch_gram(h, s, c));
  if (conditioned) {
    // rows [m, m64) of the border are not V^T (the z row, zero rows): the update writes rubbish into the padding of the
    // tiles it touches, and the identity padding is laid over it afterwards
    const int64_t m64 = round_up(m, 64);
    GemmLaunch u;
    u.A = h->d_L + h->npad * h->ld; u.lda = h->ld; u.B = u.A; u.ldb = h->ld; u.C = h->d_A2; u.ldc = npad2;
    u.M = m64; u.N = m64; u.K = h->npad; u.alpha = -1.0; u.beta = 1.0;
    u.lower = 1; u.row_off = 0; u.col_off = 0; u.b_lower_tri = 0; u.small_tile = 0;
    u.batch = batch; u.sA = h->s_mat; u.sB = h->s_mat; u.sC = npad2 * npad2;
    GPX_RET(launch_gemm_nt(h, s, u));
  }
  return launch_fill_identity_pad(h, s, h->d_A2, npad2, m, npad2, npad2, batch, npad2 * npad2);
}

__global__ void sample_finish_kernel(const double* smp, int64_t ld_s, int64_t s_smp, const double* mean, int64_t s_mean,
                                     int64_t m, int64_t nsamp, double* out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  if (c >= m) return;
  const double mu = mean ? mean[b * s_mean + c] : 0.0;
  out[(b * nsamp + r) * m + c] = mu + smp[b * s_smp + r * ld_s + c];
}

// _sample_gp_given_single_hp (gpax/gp.py:314-338) for `batch` hyper-parameter draws in one schedule -- what jax.vmap does
// with it in _sample_prior / _sample_posterior (gpax/gp.py:352-365, 382-391): per draw, posterior (or prior) mean and
// covariance at the resident x_new, sd = sqrt(diag cov), and nsamp MVN draws mean + eps chol(cov)^T.  The covariance never
// leaves HBM.
static int sample_batch_once(gpx_handle* h, int batch, const gpx_kernel_params* params, const double* y_minus_m,
                             const double* m_new, const double* new_noise, const double* eps, int64_t nsamp, int conditioned,
                             double* mean_out, double* sd_out, double* samples_out, int32_t* info_out) {
  const int64_t m = h->m, n = h->n;
  hipStream_t s = h->s_main;
  const int64_t nb2 = h->nb_auto ? (m <= 64 ? 64 : 128) : h->nb;
  const int64_t npad2 = round_up(m, nb2), spad = round_up(nsamp, 64);
  FactorPlan fp;
  if (conditioned) {
    GPX_RET(factor_prepare(h, m, true, batch));
    h->d_var = h->d_mean + std::max<int64_t>(m, 1);
    GPX_RET(set_targets_batch(h, batch, y_minus_m, m_new));
    GPX_RET(upload_params_batch(h, s, params, batch, n, h->d, &fp.noise_dev, new_noise));
    fp.kind = params->kind; fp.apply_noise = params->apply_noise; fp.m_rows = m; fp.with_r = true; fp.want_k = false;
    GPX_RET(fit_predict_enqueue(h, fp));
  } else {
    GPX_HIP_CHECK(h, hipSetDevice(h->device));
    h->batch = batch;
    GPX_RET(ensure_out(h, 2 * std::max<int64_t>(m, 1), batch));
    GPX_HIP_CHECK(h, hipMemsetAsync(h->d_out, 0, (size_t)(h->s_out * batch) * sizeof(double), s));
    const double* nd;
    gpx_kernel_params* q = const_cast<gpx_kernel_params*>(params);   // the prior covariance carries new_noise only: the per-row noise is ignored
    std::vector<gpx_kernel_params> tmp(q, q + batch);
    for (auto& t : tmp) { t.n_noise = 0; t.k_noise = nullptr; }
    GPX_RET(upload_params_batch(h, s, tmp.data(), batch, m, h->d, &nd, new_noise));
    h->have_mnew = false; h->s_mnew = 0;
    if (m_new) {
      GPX_RET(ensure(h, &h->d_mnew, &h->mnew_cap, m * batch));
      GPX_HIP_CHECK(h, hipMemcpyAsync(h->d_mnew, m_new, (size_t)(m * batch) * sizeof(double), hipMemcpyHostToDevice, s));
      h->have_mnew = true; h->s_mnew = m;
    }
  }
  // the hyper-parameter block carries prior_diag = k(x*,x*) + new_noise + jitter; the covariance build needs the noise and
  // jitter themselves on the diagonal: scalars {k_scale, new_noise, jitter} per draw in a second, transient block
  {
    const int64_t sh = h->s_hyp; const int d = h->d;
    std::vector<double> hyp2((size_t)(sh * batch));
    for (int b = 0; b < batch; ++b) {
      const gpx_kernel_params* p = params + b;
      double* hp = hyp2.data() + sh * b;
      for (int k = 0; k < d; ++k) {
        hp[k] = p->k_length[p->n_length == 1 ? 0 : k];
        hp[d + k] = (p->kind == GPX_PERIODIC) ? p->k_period[p->n_period == 1 ? 0 : k] : 1.0;
      }
      hp[2 * d] = p->k_scale; hp[2 * d + 1] = new_noise ? new_noise[b] : 0.0; hp[2 * d + 2] = p->k_jitter; hp[2 * d + 3] = 0.0;
    }
    GPX_RET(ensure(h, &h->d_vec, &h->vec_cap, std::max<int64_t>(sh * batch, 2 * h->npad)));
    GPX_RET(upload_small(h, s, h->d_vec, hyp2.data(), (int64_t)hyp2.size()));
  }
  const int64_t sv_flags = h->s_flags, sv_mini = h->s_mini;
  GPX_RET(sample_buffers(h, nb2, npad2, spad, batch));
  {
    double* sv_hyp = h->d_hyp; h->d_hyp = h->d_vec;
    const int crc = posterior_cov_to_A2(h, params->kind, m, npad2, batch, conditioned != 0);
    h->d_hyp = sv_hyp;
    GPX_RET(crc);
  }
  GPX_HIP_CHECK(h, hipMemsetAsync(h->d_eps, 0, (size_t)(spad * npad2 * batch) * sizeof(double), s));
  for (int b = 0; b < batch; ++b)
    GPX_HIP_CHECK(h, hipMemcpy2DAsync(h->d_eps + spad * npad2 * b, (size_t)npad2 * sizeof(double), eps + nsamp * m * b, (size_t)m * sizeof(double),
                                      (size_t)m * sizeof(double), (size_t)nsamp, hipMemcpyHostToDevice, s));
  const int src = sample_factor_and_draw(h, m, nb2, npad2, spad, batch);
  h->s_flags = sv_flags; h->s_mini = sv_mini;
  GPX_RET(src);
  // samples = mean + Y: one launch into a dense (batch, nsamp, m) block, one copy back
  GPX_RET(ensure(h, &h->d_S, &h->s_cap, nsamp * m * batch));
  {
    const double* meanp = conditioned ? h->d_mean : (h->have_mnew ? h->d_mnew : nullptr);
    const int64_t s_mean = conditioned ? h->s_out : h->s_mnew;
    hipLaunchKernelGGL(sample_finish_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)nsamp, (unsigned)batch), dim3(256), 0, s,
                       h->d_smp, npad2, spad * npad2, meanp, s_mean, m, nsamp, h->d_S);
    GPX_HIP_CHECK(h, hipGetLastError());
    GPX_HIP_CHECK(h, hipMemcpyAsync(samples_out, h->d_S, (size_t)(nsamp * m * batch) * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  const double* blk = nullptr;
  const int frc = fetch_block(h, 2 * std::max<int64_t>(m, 1), &blk, batch);
  const int64_t so = h->s_out;
  const bool had_mnew = h->have_mnew;
  batch_invalidate(h);
  GPX_RET(frc);
  int worst = 0;
  for (int b = 0; b < batch; ++b) {
    const double* q = blk + so * b;
    const int info = reinterpret_cast<const int*>(q + 8)[0];
    GPX_RET(check_handover(h, info));
    if (info_out) info_out[b] = info;
    if (info > worst) worst = info;
    if (mean_out) {
      if (conditioned) memcpy(mean_out + m * b, q + GPX_OUT_HEAD, (size_t)m * sizeof(double));
      else for (int64_t c = 0; c < m; ++c) mean_out[m * b + c] = (had_mnew && m_new) ? m_new[m * b + c] : 0.0;
    }
    if (sd_out) {   // sd = sqrt(diag cov) (gpax/gp.py:338): diag cov = k(x*,x*) + new_noise + jitter - |V_c|^2, the epilogue's variance
      const gpx_kernel_params* p = params + b;
      const double pd = kernel_diag_value(p->kind, p->k_scale) + ((new_noise ? new_noise[b] : 0.0) + p->k_jitter);
      for (int64_t c = 0; c < m; ++c) sd_out[m * b + c] = sqrt(conditioned ? q[GPX_OUT_HEAD + m + c] : pd);
    }
  }
  return worst;
}

extern "C" int gpx_sample_batch(gpx_handle* h, int32_t batch, const gpx_kernel_params* params, const double* y_minus_m,
                                const double* m_new, const double* new_noise, const double* eps, int64_t nsamp,
                                int32_t conditioned, double* mean_out, double* sd_out, double* samples_out, int32_t* info_out) {
  if (!h) return GPX_ERR_ARG;
  if (batch < 1 || batch > 4096 || !params || !eps || !samples_out || nsamp <= 0 || (conditioned && !y_minus_m)) {
    gpx_set_error(h, "gpx_sample_batch: bad argument"); return GPX_ERR_ARG;
  }
  if (h->m <= 0 || !h->d_xnew) { gpx_set_error(h, "gpx_sample_batch: call gpx_set_test first"); return GPX_ERR_STATE; }
  if (conditioned && h->n <= 0) { gpx_set_error(h, "gpx_sample_batch: call gpx_set_train first"); return GPX_ERR_STATE; }
  return with_retry([&] {
    return sample_batch_once(h, batch, params, y_minus_m, m_new, new_noise, eps, nsamp, conditioned, mean_out, sd_out, samples_out, info_out);
  });
}

// ------------------------------------------------------------------------------- acquisition

// Analytic acquisition values over the posterior mean / variance that the last gpx_posterior / gpx_fit_predict[_async] left in
// HBM (gpax/acquisition.py:260-266 UCB, :291-298 EI, evaluated by the reference on the host after model.predict): one
// launch, the values and their arg-max come back in one copy.  The reference's Halton optimiser pushes 10 000 candidates
// through predict() in 100 batches of 100 (gpax/acquisition.py:203-210), each re-factorising k_train; here all candidates are
// the border rows of ONE factorisation and this epilogue.
extern "C" int gpx_acquisition(gpx_handle* h, int32_t kind, double param, double mu_scale, double mu_shift, double sd_scale,
                               double* acq_out, int64_t* argmax_out, double* best_out) {
  if (!h) return GPX_ERR_ARG;
  if (kind != GPX_ACQ_EI && kind != GPX_ACQ_UCB) { gpx_set_error(h, "gpx_acquisition: kind must be GPX_ACQ_EI or GPX_ACQ_UCB"); return GPX_ERR_ARG; }
  const int64_t m = h->m_resident;
  if (m <= 0 || !h->d_out) { gpx_set_error(h, "gpx_acquisition: no posterior mean / variance resident (call gpx_posterior or gpx_fit_predict first)"); return GPX_ERR_STATE; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  hipStream_t s = h->s_main;
  GPX_RET(ensure(h, &h->d_vec, &h->vec_cap, m + 2));
  GPX_RET(launch_acquisition(h, s, h->d_mean, h->d_mean + m, m, kind, param, mu_scale, mu_shift, sd_scale, h->d_vec, h->d_vec + m));
  std::vector<double> tail(2);
  int info = 0;   // the factorisation's info word travels with the tail: after gpx_fit_predict_async without gpx_fetch_results
                  // nobody has looked at it yet (non-PD factor: positive; a hand-over that timed out: GPX_ERR_RETRY)
  if (acq_out) GPX_HIP_CHECK(h, hipMemcpyAsync(acq_out, h->d_vec, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, s));
  GPX_HIP_CHECK(h, hipMemcpyAsync(tail.data(), h->d_vec + m, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  GPX_HIP_CHECK(h, hipMemcpyAsync(&info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  GPX_HIP_CHECK(h, hipStreamSynchronize(s));
  GPX_RET(check_handover(h, info));
  if (best_out) *best_out = tail[0];
  if (argmax_out) *argmax_out = (int64_t)tail[1];
  return info > 0 ? info : GPX_OK;
}

// ------------------------------------------------------------------------- runtime identity

// Which HIP runtime this library is actually bound to.  A process that imports PyTorch first maps torch's bundled
// libamdhip64.so (SONAME libamdhip64.so.7, ROCm 7.0) and this library's NEEDED libamdhip64.so.7 then resolves to THAT
// copy instead of the /opt/rocm one it was compiled against; imported the other way round, two HIP and two HSA runtimes
// are mapped.  Round 1's hipStreamEndCapture crash lived in exactly that configuration.  Format:
// "built=<HIP_VERSION> runtime=<hipRuntimeGetVersion> path=<file the runtime was loaded from>".
extern "C" int gpx_runtime_info(char* buf, int len) {
  if (!buf || len <= 0) return GPX_ERR_ARG;
  int rt = 0;
  (void)hipRuntimeGetVersion(&rt);
  Dl_info di;
  const char* path = "?";
  if (dladdr(reinterpret_cast<void*>(&hipRuntimeGetVersion), &di) && di.dli_fname) path = di.dli_fname;
  snprintf(buf, (size_t)len, "built=%d runtime=%d path=%s", HIP_VERSION, rt, path);
  return (rt / 100000 == HIP_VERSION / 100000) ? GPX_OK : GPX_ERR_UNSUPPORTED;   // same major.minor (version = major*1e7 + minor*1e5 + patch)
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
