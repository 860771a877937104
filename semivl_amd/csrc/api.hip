// Error plumbing, version, and the per-(device, stream) helper-stream contexts of libsemivl_hip.so.
#include "svl_common.h"
#include <stdarg.h>

#include <map>
#include <mutex>
#include <utility>

static thread_local char g_err[512] = {0};

void svl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int svl_version(void) { return 600; }

extern "C" int svl_last_error(char* buf, size_t len) {
  const size_t n = strlen(g_err);
  if (buf && len > 0) {
    const size_t c = n < len - 1 ? n : len - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return (int)n;
}

// ---- helper streams (svl_common.h) ------------------------------------------------------------------------------
// A few entry points (svl_gemm_f32 on ragged token counts, svl_attention_*) run an independent, disjoint-output thin
// launch next to their main grid.  Each (device, caller stream) pair owns its OWN helper stream and fork/join events,
// created on first use on that stream's device and kept until svl_stream_release / svl_shutdown: calls on different
// caller streams (or devices, or host threads) never share one.  Host-side objects only -- no device memory.
namespace {
struct StreamCtx {
  hipStream_t aux = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
std::mutex g_mu;
std::map<std::pair<int, hipStream_t>, StreamCtx> g_ctx;

int stream_device(hipStream_t st, int* dev) {
  if (hipGetDevice(dev) != hipSuccess) {   // the caller's current device is where the kernels of this call are launched
    svl_set_error("hipGetDevice failed");
    return SVL_ERR_LAUNCH;
  }
  (void)st;
  return SVL_OK;
}

int ctx_for(hipStream_t st, StreamCtx* out) {
  int dev = 0;
  int rc = stream_device(st, &dev);
  if (rc != SVL_OK) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  auto key = std::make_pair(dev, st);
  auto it = g_ctx.find(key);
  if (it == g_ctx.end()) {
    StreamCtx c;
    SVL_HIP_CHECK(hipStreamCreateWithFlags(&c.aux, hipStreamNonBlocking));
    SVL_HIP_CHECK(hipEventCreateWithFlags(&c.fork, hipEventDisableTiming));
    SVL_HIP_CHECK(hipEventCreateWithFlags(&c.join, hipEventDisableTiming));
    it = g_ctx.emplace(key, c).first;
  }
  *out = it->second;
  return SVL_OK;
}

void destroy(StreamCtx& c) {
  if (c.aux) (void)hipStreamDestroy(c.aux);
  if (c.fork) (void)hipEventDestroy(c.fork);
  if (c.join) (void)hipEventDestroy(c.join);
}
}  // namespace

int svl_fork(hipStream_t st, hipStream_t* aux) {
  StreamCtx c;
  int rc = ctx_for(st, &c);
  if (rc != SVL_OK) return rc;
  SVL_HIP_CHECK(hipEventRecord(c.fork, st));
  SVL_HIP_CHECK(hipStreamWaitEvent(c.aux, c.fork, 0));
  *aux = c.aux;
  return SVL_OK;
}

int svl_join(hipStream_t st) {
  StreamCtx c;
  int rc = ctx_for(st, &c);
  if (rc != SVL_OK) return rc;
  SVL_HIP_CHECK(hipEventRecord(c.join, c.aux));
  SVL_HIP_CHECK(hipStreamWaitEvent(st, c.join, 0));
  return SVL_OK;
}

extern "C" int svl_stream_prepare(svl_stream_t stream) {
  StreamCtx c;
  return ctx_for((hipStream_t)stream, &c);
}

// The helper stream of `stream` (created if needed): measurement aid for callers that probe which hardware queue a stream of
// their own would share (semivl_amd/train.py: the communication stream is picked against the step's streams AND their helpers).
extern "C" int svl_stream_helper(svl_stream_t stream, void** helper) {
  SVL_CHECK_ARG(helper, "svl_stream_helper: null out pointer");
  StreamCtx c;
  const int rc = ctx_for((hipStream_t)stream, &c);
  if (rc != SVL_OK) return rc;
  *helper = (void*)c.aux;
  return SVL_OK;
}

extern "C" int svl_stream_release(svl_stream_t stream) {
  int dev = 0;
  int rc = stream_device((hipStream_t)stream, &dev);
  if (rc != SVL_OK) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_ctx.find(std::make_pair(dev, (hipStream_t)stream));
  if (it != g_ctx.end()) {
    destroy(it->second);
    g_ctx.erase(it);
  }
  return SVL_OK;
}

extern "C" int svl_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& kv : g_ctx) destroy(kv.second);
  g_ctx.clear();
  return SVL_OK;
}

extern "C" int svl_num_stream_contexts(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return (int)g_ctx.size();
}

// ------------------------------------------------------------------------------------------------ measurement aid
// n_waves single-wave workgroups, each comparing the shader-clock counter (s_memtime) with the constant 100 MHz counter
// (s_memrealtime) over `ticks_100mhz`: out[2 i] = shader-clock cycles, out[2 i + 1] = 100 MHz ticks.  Launched on a
// stream of its own NEXT TO a kernel under measurement it reports the clock the chip sustains under that kernel's
// instruction mix (the peaks of MI355X_MICROARCH.md are quoted at 2.4 GHz; dense-MFMA kernels are given 1.4 ... 1.7 GHz).
namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, unsigned long long ticks) {
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  unsigned long long w = w0, c = c0;
  while (w - w0 < ticks) {
    __builtin_amdgcn_s_sleep(100);
    c = clock64();
    w = wall_clock64();
  }
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c - c0;
    out[2 * blockIdx.x + 1] = w - w0;
  }
}
}  // namespace

extern "C" int svl_clock_probe(unsigned long long* out, int n_waves, unsigned long long ticks_100mhz, svl_stream_t stream) {
  SVL_CHECK_ARG(out && n_waves > 0 && n_waves <= 1024 && ticks_100mhz <= 1000000000ull, "svl_clock_probe: bad args");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(n_waves), dim3(64), 0, (hipStream_t)stream, out, ticks_100mhz);
  SVL_LAUNCH_CHECK("svl_clock_probe");
  return SVL_OK;
}
