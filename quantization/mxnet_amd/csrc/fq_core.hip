// libfakequant — errors, event timing, device info, streaming policy
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_common.h"

namespace fqi {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- optional per-kernel event timing (fq_profile_*) -----------------------------------------------------------
bool g_prof_on = false;
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;

void prof_push(const ProfRec& r) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(r);
}

// Compute units of the device the calling thread has current (the caller selects the device of its tensors before it
// calls in; the grid caps follow that device).
int num_cu() {
  constexpr int kMaxDev = 64;
  static int cache[kMaxDev] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
  if (cache[dev] == 0) {
    hipDeviceProp_t prop;
    int cu = 0;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cu = prop.multiProcessorCount;
    cache[dev] = cu > 0 ? cu : 256;
  }
  return cache[dev];
}

// LDS a workgroup of the CURRENT device may ask for (160 KB on gfx950); the fused forms that size their tiles by it ask here
// instead of assuming the figure.  Without a device (the build check on a CPU box): gfx950's.
int max_lds_bytes() {
  constexpr int kMaxDev = 64;
  static int cache[kMaxDev] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 160 * 1024;
  if (cache[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 160 * 1024;
    cache[dev] = v;
  }
  return cache[dev];
}

// Streaming policy, by kernel and tensor size.  Measured on MI355X (profiles/r1_policy_sweep.txt, r1_kbench.txt):
// tensors that (with their output) fit the 256 MiB Infinity Cache want PLAIN loads/stores — the producer just left x
// there and the consumer (the convolution) will find y there; larger tensors want nontemporal loads+stores in the apply
// pass, and the online apply pass walks backwards to start on what the statistic pass read last.
// FQ_POLICY_STAT / FQ_POLICY_ONLINE / FQ_POLICY_OFFLINE (integer OR of the kPol* bits) override for tuning runs;
// FQ_POLICY_BIG_BYTES moves the size threshold.
int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

constexpr int kPolNtLoad_ = 1, kPolNtStore_ = 2, kPolReverse_ = 4;   // = kPol* of fq_common.h
int stream_policy(int kernel_id, int64_t numel) {
  static const int ov_stat = env_int("FQ_POLICY_STAT", -1);
  static const int ov_on = env_int("FQ_POLICY_ONLINE", -1);
  static const int ov_off = env_int("FQ_POLICY_OFFLINE", -1);
  static const int64_t big = (int64_t)env_int("FQ_POLICY_BIG_MB", 160) * 1000000;
  const bool is_big = numel * (int64_t)sizeof(float) >= big;
  switch (kernel_id) {
    case FQ_KERNEL_STAT:
      return ov_stat >= 0 ? ov_stat : 0;
    case FQ_KERNEL_APPLY_ONLINE:
      if (ov_on >= 0) return ov_on;
      return is_big ? (kPolReverse_ | kPolNtLoad_ | kPolNtStore_) : kPolReverse_;
    default:
      if (ov_off >= 0) return ov_off;
      return is_big ? (kPolNtLoad_ | kPolNtStore_) : 0;
  }
}

}  // namespace fqi

using namespace fqi;

extern "C" {

const char* fq_last_error(void) { return g_err; }
int fq_version(void) { return 101; }
// sha1 of the sources this library was built from (csrc/build.py: source_id); "FQ_BUILD_ID=<40 hex>" is also what
// build.py looks for in the file's bytes to decide whether a built library belongs to the tree it sits in
#ifndef FQ_BUILD_ID
#define FQ_BUILD_ID "0000000000000000000000000000000000000000"
#endif
const char* fq_build_id(void) {
  static const char id[] = "FQ_BUILD_ID=" FQ_BUILD_ID;
  return id + 12;
}

// 1 when this library was built with the named optional part ("pipe": the shelved pipe form of fq_pwconv_i8, csrc/build.py --dev)
int fq_build_has(const char* feature) {
  if (feature == nullptr) return 0;
#ifdef FQ_DEV_FORMS
  if (strcmp(feature, "pipe") == 0) return 1;
#endif
  return 0;
}

int fq_device_info(char* arch, int arch_len, int* compute_units, int* wavefront) {
  int dev = 0;
  FQ_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  FQ_HIP(hipGetDeviceProperties(&prop, dev));
  if (arch != nullptr && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (wavefront) *wavefront = prop.warpSize;
  return FQ_OK;
}

int fq_profile_enable(int on) {
  g_prof_on = on != 0;
  return FQ_OK;
}

int fq_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
  return FQ_OK;
}

int fq_profile_read_moved(int kernel_id, double* total_moved_bytes) {
  FQ_REQUIRE(kernel_id >= 0 && kernel_id < FQ_KERNEL_COUNT, "fq_profile_read_moved: bad kernel id %d", kernel_id);
  FQ_REQUIRE(total_moved_bytes, "fq_profile_read_moved: null pointer");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double moved = 0.0;
  for (auto& r : g_prof)
    if (r.kid == kernel_id) moved += r.moved;
  *total_moved_bytes = moved;
  return FQ_OK;
}

int fq_profile_read(int kernel_id, double* total_ms, int64_t* launches, double* total_bytes) {
  FQ_REQUIRE(kernel_id >= 0 && kernel_id < FQ_KERNEL_COUNT, "fq_profile_read: bad kernel id %d", kernel_id);
  FQ_REQUIRE(total_ms && launches && total_bytes, "fq_profile_read: null pointer");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double ms = 0.0, bytes = 0.0;
  int64_t cnt = 0;
  for (auto& r : g_prof) {
    if (r.kid != kernel_id) continue;
    FQ_HIP(hipEventSynchronize(r.b));
    float t = 0.f;
    FQ_HIP(hipEventElapsedTime(&t, r.a, r.b));
    ms += t;
    bytes += r.bytes;
    ++cnt;
  }
  *total_ms = ms;
  *launches = cnt;
  *total_bytes = bytes;
  return FQ_OK;
}

}  // extern "C"

namespace {
__global__ void spin_kernel(unsigned long long* __restrict__ t, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  unsigned long long t1 = t0;
  while (t1 - t0 < ticks) t1 = wall_clock64();
  if (threadIdx.x == 0) {
    t[0] = t0;
    t[1] = t1;
  }
}
}  // namespace

extern "C" {

int fq_profile_calibrate(void* scratch, int repeats, double* pair_ms, double* null_kernel_ms, fqStream_t stream) {
  FQ_REQUIRE(scratch && pair_ms && null_kernel_ms && repeats > 0 && repeats <= 4096, "fq_profile_calibrate: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  // (a) every bracketed launch enqueued back to back (a busy queue, like the timed region), synchronised once at the end
  std::vector<hipEvent_t> ev((size_t)repeats * 2 + 2);
  for (auto& e : ev) FQ_HIP(hipEventCreate(&e));
  for (int i = 0; i < repeats; ++i) {
    FQ_HIP(hipEventRecord(ev[2 * i], st));
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, st, (float*)scratch, (int64_t)1, 0.0f);
    FQ_HIP(hipEventRecord(ev[2 * i + 1], st));
  }
  // (b) the same launches with ONE pair around all of them
  FQ_HIP(hipEventRecord(ev[2 * repeats], st));
  for (int i = 0; i < repeats; ++i)
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, st, (float*)scratch, (int64_t)1, 0.0f);
  FQ_HIP(hipEventRecord(ev[2 * repeats + 1], st));
  FQ_HIP(hipEventSynchronize(ev.back()));
  std::vector<float> ts;
  for (int i = 0; i < repeats; ++i) {
    float t = 0.f;
    FQ_HIP(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]));
    ts.push_back(t);
  }
  float all = 0.f;
  FQ_HIP(hipEventElapsedTime(&all, ev[2 * repeats], ev[2 * repeats + 1]));
  for (auto& e : ev) (void)hipEventDestroy(e);
  std::sort(ts.begin(), ts.end());
  *pair_ms = ts[ts.size() / 2];
  *null_kernel_ms = (double)all / repeats;
  return FQ_OK;
}

// What bracketing a launch with an event pair adds to what the launch costs inside a stream of back-to-back launches -
// measured on a kernel LONG enough that dispatch cannot hide behind it the way it does behind a one-element kernel
// (fq_profile_calibrate's pair - null figure, ~4.6 us, over-corrects 30 us kernels by ~2 us: profiles/r4_events_vs_rocprof.txt).
// A one-wavefront kernel spins `spin_us` on the constant-rate wall clock; (a) `repeats` launches, each bracketed, enqueued back
// to back: median pair time P; (b) the same launches inside ONE pair: B / repeats;  overhead_ms <- P - B / repeats.
// (raw event time - overhead per launch) is held against rocprofv3's kernel table by tools/check_events_vs_rocprof.py.
// spin_ms <- the kernel's own median first-to-last clock distance.  Synchronises.  scratch: repeats * 16 bytes.
int fq_profile_launch_overhead(void* scratch, int repeats, double spin_us, double* overhead_ms, double* spin_ms,
                               fqStream_t stream) {
  FQ_REQUIRE(scratch && overhead_ms && spin_ms && repeats > 0 && repeats <= 4096 && spin_us > 0 && spin_us < 1e5,
             "fq_profile_launch_overhead: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int dev = 0, khz = 0;
  FQ_HIP(hipGetDevice(&dev));
  FQ_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
  FQ_REQUIRE(khz > 0, "fq_profile_launch_overhead: the device reports no wall clock rate");
  const unsigned long long ticks = (unsigned long long)(spin_us * 1e-3 * khz);
  std::vector<hipEvent_t> ev((size_t)repeats * 2 + 2);
  for (auto& e : ev) FQ_HIP(hipEventCreate(&e));
  unsigned long long* t = (unsigned long long*)scratch;
  for (int i = 0; i < repeats; ++i) {
    FQ_HIP(hipEventRecord(ev[2 * i], st));
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, t + 2 * i, ticks);
    FQ_HIP(hipEventRecord(ev[2 * i + 1], st));
  }
  FQ_HIP(hipEventRecord(ev[2 * repeats], st));
  for (int i = 0; i < repeats; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, t + 2 * i, ticks);
  FQ_HIP(hipEventRecord(ev[2 * repeats + 1], st));
  FQ_LAUNCH_CHECK();
  FQ_HIP(hipStreamSynchronize(st));
  std::vector<unsigned long long> host((size_t)repeats * 2);
  FQ_HIP(hipMemcpy(host.data(), t, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> pair, own;
  for (int i = 0; i < repeats; ++i) {
    float el = 0.f;
    FQ_HIP(hipEventElapsedTime(&el, ev[2 * i], ev[2 * i + 1]));
    pair.push_back((double)el);
    own.push_back((double)(host[2 * i + 1] - host[2 * i]) / (double)khz);
  }
  float all = 0.f;
  FQ_HIP(hipEventElapsedTime(&all, ev[2 * repeats], ev[2 * repeats + 1]));
  for (auto& e : ev) (void)hipEventDestroy(e);
  std::sort(pair.begin(), pair.end());
  std::sort(own.begin(), own.end());
  *overhead_ms = pair[pair.size() / 2] - (double)all / repeats;
  *spin_ms = own[own.size() / 2];
  return FQ_OK;
}

}  // extern "C"
