// capi.hip -- extern "C" surface of libtorchfx_hip.so (see include/torchfx_hip.h) plus the
// small shared services: thread-local error text, per-kernel HIP-event timing, device scratch,
// (the elementwise kernels live in effects.hip).
#include "common.h"
#include "epilogue.h"
#include "../../include/torchfx_hip.h"

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace tfx {

// implemented in sos.hip / fir.hip / fftconv.hip
void sos_forward(const void *x, int x_dtype, void *y, int y_dtype, int64_t C, int64_t T,
                 const double *sos_host, int64_t K, const double *sx_in, const double *sy_in,
                 double *sx_out, double *sy_out, void *y_sections, int precision, hipStream_t stream, int64_t NB = 1,
                 bool sum_bands = false, const Epilogue *ep = nullptr);
void sos_plan_info(const double *sos_host, int64_t K, int *precision, int64_t *warmup, double *err_bound);
void sos_clear_plans();
void fir_direct_forward(const void *x, void *y, int dtype, int64_t C, int64_t T,
                        const void *kernel_host, int64_t K, hipStream_t stream, const void *hist = nullptr, int64_t H = 0);
void quantile_abs_forward(const float *x, int64_t n, double q, double *out_dev, hipStream_t stream);
bool chunk_supported(int64_t C, int64_t T, int64_t K, int64_t Kf);
void chunk_forward(const float *x, int64_t x_pitch, float *y, int64_t C, int64_t T, const double *sos_host, int64_t K,
                   const double *sx_in, const double *sy_in, double *sx_out, double *sy_out,
                   const float *taps_host, int64_t Kf, const float *hist_in, float *hist_out,
                   double gain, int scale, int clamp, int precision, hipStream_t stream);
void fir_hist_update(const void *x, const void *hist_in, void *hist_out, int dtype, int64_t C, int64_t T, int64_t H,
                     hipStream_t stream);
void fir_clear();
void fft_conv_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, const void *kernel_host,
                      int64_t K, int64_t pad_left, int64_t pad_right, hipStream_t stream, const void *hist = nullptr,
                      int64_t H = 0, const Epilogue *ep = nullptr);
bool sos_fft_conv_supported(int64_t T, const double *sos_host, int64_t Ksos, int64_t K, int64_t pad_left, int64_t pad_right, int force);
int64_t sos_fft_conv_warmup(const double *sos_host, int64_t Ksos);
bool sos_fft_conv_plan(int64_t T, const double *sos_host, int64_t Ksos, int64_t K, int64_t pad_left, int64_t pad_right, int force,
                       int64_t *N_out, int64_t *S_out, int64_t *F_out, int64_t *warm_out);
void sos_fft_conv_forward(const float *x, float *y, int64_t C, int64_t T, const double *sos_host, int64_t Ksos,
                          const float *kernel_host, int64_t K, int64_t pad_left, int64_t pad_right, double *sections, int force,
                          const Epilogue *ep, hipStream_t stream);
void normalize_apply_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, int mode, int per_row, double peak,
                             const double *stat, hipStream_t stream);
void fftconv_clear();
void olsnative_clear();
bool olsnative_supported(int64_t K, int64_t L, int64_t *N_out);
void olsnative_geometry(int64_t K, int64_t Tn, int64_t pl, int64_t pr, int64_t N, int64_t *S_out, int64_t *F_out);
void olsnative64_clear();
bool olsnative64_supported(int64_t K, int64_t L, bool has_hist);
void olsnative64_geometry(int64_t K, int64_t Tn, int64_t pl, int64_t pr, int64_t *S_out, int64_t *F_out);
void olslds_clear();
void olsnative_prewarm();
bool olslds_supported(int64_t K, int dtype, int64_t L, int64_t *N_out);
void olslds_geometry(int64_t K, int64_t Tn, int64_t pl, int64_t pr, int elem_bytes, int64_t N, int64_t *lead_out, int64_t *S_out);
// effects.hip
void gain_forward(const void *x, void *y, int dtype, int64_t n, double gain, int clamp, hipStream_t stream);
void stat_forward(const void *x, int dtype, int64_t C, int64_t T, int mode, int per_row, double *out_dev, hipStream_t stream);
void normalize_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, int mode, int per_row, double peak,
                       hipStream_t stream);
void sum_forward(const void *const *xs_host, int n, void *y, int dtype, int64_t numel, hipStream_t stream);
void delay_line_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, int64_t delay, double coeff,
                        hipStream_t stream);
// layout.hip
void deinterleave_forward(const void *in, int in_kind, float *out, int64_t F, int64_t C, int64_t ld_out, int64_t f_base,
                          double scale, hipStream_t stream);
void interleave_forward(const float *in, float *out, int64_t F, int64_t C, int64_t ld_in, int64_t f_base,
                        hipStream_t stream);
int64_t fftconv_block_size(int64_t K, int64_t L);

// ---- environment knobs, read once --------------------------------------------------------------
namespace {
struct EnvTable {
    std::mutex mu;
    std::map<std::string, std::pair<bool, int64_t>> by_name;      // keyed by the knob's TEXT: any pointer may ask
    std::atomic<bool> dynamic{false};                             // read by every lookup, written by env_reload
    std::atomic<uint64_t> generation{1};                          // bumped by env_reload: the per-thread shortcuts below expire
    EnvTable()
    {
        const char *e = getenv("TFX_ENV_DYNAMIC");
        dynamic.store(e && *e && *e != '0', std::memory_order_relaxed);
    }
};
EnvTable &env_table()
{
    static EnvTable t;
    return t;
}
}  // namespace

int64_t env_i64(const char *name, int64_t dflt)
{
    EnvTable &t = env_table();
    if (t.dynamic.load(std::memory_order_relaxed)) {
        const char *e = getenv(name);
        return (e && *e) ? atoll(e) : dflt;
    }
    // A per-thread shortcut keyed by the pointer sits in front of the shared table (a dispatch asks for about ten knobs; no lock
    // in steady state).  An entry is trusted only while the text behind the pointer is still the knob it was stored for, so a
    // reused buffer cannot alias another knob, and only within the generation it was read in.
    struct Local {
        uint64_t gen = 0;
        std::map<const void *, std::pair<std::string, std::pair<bool, int64_t>>> by_ptr;
    };
    thread_local Local loc;
    const uint64_t gen = t.generation.load(std::memory_order_acquire);
    if (loc.gen != gen) { loc.by_ptr.clear(); loc.gen = gen; }
    auto it = loc.by_ptr.find((const void *)name);
    if (it != loc.by_ptr.end() && it->second.first == name)
        return it->second.second.first ? it->second.second.second : dflt;
    std::pair<bool, int64_t> v;
    {
        std::lock_guard<std::mutex> lk(t.mu);
        auto in = t.by_name.find(name);
        if (in == t.by_name.end()) {
            const char *e = getenv(name);
            in = t.by_name.emplace(name, std::make_pair(e && *e, (e && *e) ? (int64_t)atoll(e) : (int64_t)0)).first;
        }
        v = in->second;
    }
    loc.by_ptr[(const void *)name] = std::make_pair(std::string(name), v);
    return v.first ? v.second : dflt;
}

void env_reload()
{
    EnvTable &t = env_table();
    std::lock_guard<std::mutex> lk(t.mu);
    t.by_name.clear();
    const char *e = getenv("TFX_ENV_DYNAMIC");
    t.dynamic.store(e && *e && *e != '0', std::memory_order_relaxed);
    t.generation.fetch_add(1, std::memory_order_release);
}

// ---- errors ------------------------------------------------------------------------------------
static thread_local std::string t_last_error;
void set_last_error(const std::string &msg) { t_last_error = msg; }

// ---- profiling -----------------------------------------------------------------------------------
// Low-overhead per-kernel timing: events come from a pool (no create/destroy per launch) and
// consecutive launches inside one API call share the boundary event (end of A == begin of B),
// so a launch costs one hipEventRecord.
struct ProfRec {
    const char *name;      // string literals only
    hipEvent_t a, b;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_pool;         // free events
static std::vector<hipEvent_t> g_used;         // to be recycled at collect
static const char *g_open_name = nullptr;
static hipEvent_t g_open_a = nullptr;
static hipEvent_t g_last_end = nullptr;        // reusable as the next begin while g_chain is true
static bool g_chain = false;
static hipStream_t g_last_stream = nullptr;

static hipEvent_t pool_get()
{
    hipEvent_t e = nullptr;
    if (!g_pool.empty()) { e = g_pool.back(); g_pool.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_used.push_back(e);
    return e;
}
bool prof_on() { return g_prof_on; }
void prof_break_chain() { g_chain = false; }
void prof_begin(const char *name, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_open_name = name;
    if (g_chain && g_last_end && g_last_stream == s) { g_open_a = g_last_end; return; }
    g_open_a = pool_get();
    if (g_open_a) (void)hipEventRecord(g_open_a, s);
}
void prof_end(hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_open_name || !g_open_a) return;
    hipEvent_t b = pool_get();
    if (!b) return;
    (void)hipEventRecord(b, s);
    g_prof.push_back(ProfRec{g_open_name, g_open_a, b});
    g_last_end = b;
    g_last_stream = s;
    g_chain = true;
    g_open_name = nullptr;
}

// ---- scratch -------------------------------------------------------------------------------------
// Buffers come from hipMalloc unless the host installed its own allocator (tfx_set_workspace_allocator: the torch module routes
// them through PyTorch's caching allocator, so torch.cuda.memory_allocated, its out-of-memory retry and its error type see them).
struct Scratch {
    void *p = nullptr;
    size_t bytes = 0;
    bool hooked = false;                 // allocated by the installed hook (freed through it, even if the hook was replaced since)
    tfx_free_fn free_fn = nullptr;
    void *ctx = nullptr;
    int dev = 0;
};
static std::mutex g_scr_mu;
static std::map<std::string, Scratch> g_scr;
static tfx_alloc_fn g_alloc_fn = nullptr;
static tfx_free_fn g_free_fn = nullptr;
static void *g_alloc_ctx = nullptr;

static void scratch_release(Scratch &s)
{
    if (!s.p) return;
    (void)hipDeviceSynchronize();        // the old buffer may still be in use by queued work: drain before freeing
    if (s.hooked) s.free_fn(s.p, s.dev, s.ctx);
    else (void)hipFree(s.p);
    s.p = nullptr;
    s.bytes = 0;
}

void *scratch_try(const char *tag, size_t bytes, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(g_scr_mu);
    char key[96];
    const int dev = current_device();
    snprintf(key, sizeof(key), "%s@%p#%d", tag, (void *)stream, dev);
    Scratch &s = g_scr[key];
    if (s.bytes >= bytes) return s.p;
    {
        // a workspace that has to grow would be allocated (and the old one freed behind a device synchronise) in the middle of a
        // stream capture: under PyTorch's allocator the new block would belong to the graph's private pool and die with it
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        TFX_CHECK(!(hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone),
                  "workspace '%s' (%zu bytes) would be allocated inside a stream capture -- run the op once on this stream with these "
                  "shapes before capturing", tag, bytes);
    }
    scratch_release(s);
    if (g_alloc_fn) {
        void *q = g_alloc_fn(bytes, dev, (void *)stream, g_alloc_ctx);
        if (!q) return nullptr;
        s.p = q; s.hooked = true; s.free_fn = g_free_fn; s.ctx = g_alloc_ctx;
    } else {
        void *q = nullptr;
        if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        s.p = q; s.hooked = false;
    }
    s.bytes = bytes; s.dev = dev;
    return s.p;
}
void *scratch(const char *tag, size_t bytes, hipStream_t stream)
{
    void *q = scratch_try(tag, bytes, stream);
    TFX_CHECK(q || bytes == 0, "out of device memory: a %zu-byte workspace (%s) could not be allocated%s", bytes, tag,
              g_alloc_fn ? " by the installed workspace allocator" : " by hipMalloc");
    return q;
}
size_t scratch_bytes()
{
    std::lock_guard<std::mutex> lk(g_scr_mu);
    size_t n = 0;
    for (auto &kv : g_scr) n += kv.second.bytes;
    return n;
}
void scratch_clear()
{
    std::lock_guard<std::mutex> lk(g_scr_mu);
    for (auto &kv : g_scr) scratch_release(kv.second);
    g_scr.clear();
}
void scratch_set_allocator(tfx_alloc_fn a, tfx_free_fn f, void *ctx)
{
    std::lock_guard<std::mutex> lk(g_scr_mu);
    g_alloc_fn = a; g_free_fn = f; g_alloc_ctx = ctx;
}

// ---- elementwise kernels ----------------------------------------------------------------------------
}  // namespace tfx

using namespace tfx;

// every compute entry point enqueues under one lock: the internal fork/join events, lane streams and
// plan caches are process-wide, and ctypes releases the GIL, so two Python threads may be in here
static std::recursive_mutex g_api_mu;
#define TFX_API_BEGIN try { std::lock_guard<std::recursive_mutex> _api_lock(g_api_mu); prof_break_chain();
#define TFX_API_END                                                                           \
    return 0;                                                                                 \
    }                                                                                         \
    catch (const std::exception &e) { set_last_error(e.what()); return 1; }                   \
    catch (...) { set_last_error("unknown error"); return 2; }

extern "C" {

int tfx_version(void) { return 100; }
const char *tfx_last_error(void) { return t_last_error.c_str(); }

int tfx_device_info(char *name, int len, int *cus)
{
    TFX_API_BEGIN
    int dev = 0;
    TFX_HIP(hipGetDevice(&dev));
    hipDeviceProp_t pr;
    TFX_HIP(hipGetDeviceProperties(&pr, dev));
    if (name && len > 0) snprintf(name, (size_t)len, "%s", pr.gcnArchName);
    if (cus) *cus = pr.multiProcessorCount;
    TFX_API_END
}

int tfx_sos_forward(const void *x, int x_dtype, void *y, int y_dtype, int64_t C, int64_t T,
                    const double *sos_host, int64_t K, const double *state_x_in, const double *state_y_in,
                    double *state_x_out, double *state_y_out, void *y_sections, int precision,
                    tfx_stream_t stream)
{
    TFX_API_BEGIN
    sos_forward(x, x_dtype, y, y_dtype, C, T, sos_host, K, state_x_in, state_y_in, state_x_out, state_y_out,
                y_sections, precision, (hipStream_t)stream);
    TFX_API_END
}

static Epilogue to_epilogue(const tfx_epilogue *e)
{
    Epilogue ep;
    if (!e) return ep;
    TFX_CHECK(e->stat_mode >= -1 && e->stat_mode <= 1, "epilogue: bad statistic mode %d", e->stat_mode);
    TFX_CHECK(e->gain == e->gain, "epilogue: NaN gain");
    ep.gain = e->gain; ep.scale = e->gain != 1.0; ep.clamp = e->clamp != 0;
    ep.stat_mode = e->stat_mode; ep.per_row = e->stat_per_row != 0; ep.stat_out = e->stat_out;
    return ep;
}

int tfx_sos_forward_ep(const void *x, int x_dtype, void *y, int y_dtype, int64_t C, int64_t T,
                       const double *sos_host, int64_t K, const double *state_x_in, const double *state_y_in,
                       double *state_x_out, double *state_y_out, int precision, const tfx_epilogue *epilogue,
                       tfx_stream_t stream)
{
    TFX_API_BEGIN
    const Epilogue ep = to_epilogue(epilogue);
    sos_forward(x, x_dtype, y, y_dtype, C, T, sos_host, K, state_x_in, state_y_in, state_x_out, state_y_out,
                nullptr, precision, (hipStream_t)stream, 1, false, &ep);
    TFX_API_END
}

int tfx_fft_conv_forward_ep(const void *x, void *y, int dtype, int64_t C, int64_t T, const void *kernel_host,
                            int64_t K, int64_t pad_left, int64_t pad_right, const tfx_epilogue *epilogue,
                            tfx_stream_t stream)
{
    TFX_API_BEGIN
    const Epilogue ep = to_epilogue(epilogue);
    fft_conv_forward(x, y, dtype, C, T, kernel_host, K, pad_left, pad_right, (hipStream_t)stream, nullptr, 0, &ep);
    TFX_API_END
}

int tfx_sos_fft_conv_supported(int64_t T, const double *sos_host, int64_t K, int64_t taps, int64_t pad_left, int64_t pad_right,
                               int force_block)
{
    try {
        return (sos_host && sos_fft_conv_supported(T, sos_host, K, taps, pad_left, pad_right, force_block)) ? 1 : 0;
    } catch (...) {
        return 0;
    }
}

int tfx_sos_fft_conv_plan_info(int64_t T, const double *sos_host, int64_t K, int64_t taps, int64_t pad_left, int64_t pad_right,
                               int force_block, int64_t *N, int64_t *S, int64_t *F, int64_t *warmup)
{
    try {
        return (sos_host && sos_fft_conv_plan(T, sos_host, K, taps, pad_left, pad_right, force_block, N, S, F, warmup)) ? 1 : 0;
    } catch (...) {
        return 0;
    }
}

int64_t tfx_sos_fft_conv_warmup(const double *sos_host, int64_t K)
{
    try {
        return sos_host ? sos_fft_conv_warmup(sos_host, K) : -1;
    } catch (...) {
        return -1;
    }
}

int tfx_sos_fft_conv_forward(const float *x, float *y, int64_t C, int64_t T, const double *sos_host, int64_t K,
                             const float *kernel_host, int64_t taps, int64_t pad_left, int64_t pad_right,
                             double *y_sections, int force_block, const tfx_epilogue *epilogue, tfx_stream_t stream)
{
    TFX_API_BEGIN
    const Epilogue ep = to_epilogue(epilogue);
    sos_fft_conv_forward(x, y, C, T, sos_host, K, kernel_host, taps, pad_left, pad_right, y_sections, force_block,
                         epilogue ? &ep : nullptr, (hipStream_t)stream);
    TFX_API_END
}

int tfx_normalize_apply(const void *x, void *y, int dtype, int64_t C, int64_t T, int mode, int per_row, double peak,
                        const double *stat_dev, tfx_stream_t stream)
{
    TFX_API_BEGIN
    normalize_apply_forward(x, y, dtype, C, T, mode, per_row, peak, stat_dev, (hipStream_t)stream);
    TFX_API_END
}

int tfx_sos_bank_forward(const void *x, int x_dtype, void *y, int y_dtype, int64_t C, int64_t T,
                         const double *sos_host, int64_t n_bands, int64_t K, const double *state_x_in,
                         const double *state_y_in, double *state_x_out, double *state_y_out, int precision,
                         tfx_stream_t stream)
{
    TFX_API_BEGIN
    sos_forward(x, x_dtype, y, y_dtype, C, T, sos_host, K, state_x_in, state_y_in, state_x_out, state_y_out,
                nullptr, precision, (hipStream_t)stream, n_bands);
    TFX_API_END
}

int tfx_sos_bank_sum_forward(const void *x, int x_dtype, void *y, int y_dtype, int64_t C, int64_t T,
                             const double *sos_host, int64_t n_bands, int64_t K, const double *state_x_in,
                             const double *state_y_in, double *state_x_out, double *state_y_out, int precision,
                             tfx_stream_t stream)
{
    TFX_API_BEGIN
    sos_forward(x, x_dtype, y, y_dtype, C, T, sos_host, K, state_x_in, state_y_in, state_x_out, state_y_out,
                nullptr, precision, (hipStream_t)stream, n_bands, true);
    TFX_API_END
}

int tfx_sos_plan_info(const double *sos_host, int64_t K, int *precision, int64_t *warmup, double *err_bound)
{
    TFX_API_BEGIN
    TFX_CHECK(sos_host && K >= 1 && K <= 512, "sos_plan_info: null coefficients or bad section count %lld", (long long)K);
    sos_plan_info(sos_host, K, precision, warmup, err_bound);
    TFX_API_END
}

int tfx_biquad_forward(const void *x, int x_dtype, void *y, int y_dtype, int64_t C, int64_t T,
                       const double *b_host, double a1, double a2, const double *state_x_in,
                       const double *state_y_in, double *state_x_out, double *state_y_out, int precision,
                       tfx_stream_t stream)
{
    TFX_API_BEGIN
    // a biquad is the K = 1 cascade; [C,2] states are the [1,C,2] layout
    TFX_CHECK(b_host, "biquad_forward: null numerator pointer");
    const double sos[6] = {b_host[0], b_host[1], b_host[2], 1.0, a1, a2};
    sos_forward(x, x_dtype, y, y_dtype, C, T, sos, 1, state_x_in, state_y_in, state_x_out, state_y_out, nullptr,
                precision, (hipStream_t)stream);
    TFX_API_END
}

int tfx_fir_direct_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, const void *kernel_host,
                           int64_t K, tfx_stream_t stream)
{
    TFX_API_BEGIN
    fir_direct_forward(x, y, dtype, C, T, kernel_host, K, (hipStream_t)stream);
    TFX_API_END
}

int tfx_fft_conv_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, const void *kernel_host,
                         int64_t K, int64_t pad_left, int64_t pad_right, tfx_stream_t stream)
{
    TFX_API_BEGIN
    fft_conv_forward(x, y, dtype, C, T, kernel_host, K, pad_left, pad_right, (hipStream_t)stream);
    TFX_API_END
}

int tfx_fir_stream_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, const void *kernel_host, int64_t K,
                           int direct, const void *hist_in, void *hist_out, tfx_stream_t stream)
{
    TFX_API_BEGIN
    TFX_CHECK(K >= 1, "fir_stream_forward: empty kernel");
    const int64_t H = K - 1;
    if (C > 0 && T > 0) {
        if (direct) fir_direct_forward(x, y, dtype, C, T, kernel_host, K, (hipStream_t)stream, H ? hist_in : nullptr, hist_in ? H : 0);
        else fft_conv_forward(x, y, dtype, C, T, kernel_host, K, H, 0, (hipStream_t)stream, H ? hist_in : nullptr, hist_in ? H : 0);
    }
    if (hist_out && C > 0) {
        TFX_CHECK(T == 0 || x, "fir_stream_forward: null signal");
        fir_hist_update(x, hist_in, hist_out, dtype, C, T, H, (hipStream_t)stream);
    }
    TFX_API_END
}

int tfx_quantile_abs(const float *x, int64_t n, double q, double *out_dev, tfx_stream_t stream)
{
    TFX_API_BEGIN
    quantile_abs_forward(x, n, q, out_dev, (hipStream_t)stream);
    TFX_API_END
}

int tfx_chunk_supported(int64_t C, int64_t T, int64_t K, int64_t Kf) { return chunk_supported(C, T, K, Kf) ? 1 : 0; }

int tfx_chunk_forward(const float *x, int64_t x_pitch, float *y, int64_t C, int64_t T, const double *sos_host, int64_t K,
                      const double *state_x_in, const double *state_y_in, double *state_x_out, double *state_y_out,
                      const float *taps_host, int64_t Kf, const float *hist_in, float *hist_out,
                      double gain, int scale, int clamp, int precision, tfx_stream_t stream)
{
    TFX_API_BEGIN
    chunk_forward(x, x_pitch, y, C, T, sos_host, K, state_x_in, state_y_in, state_x_out, state_y_out, taps_host, Kf, hist_in, hist_out,
                  gain, scale, clamp, precision, (hipStream_t)stream);
    TFX_API_END
}

static void ols_plan(int64_t K, int64_t T, int64_t pad_left, int64_t pad_right, int dtype, int64_t *N, int64_t *S,
                     int64_t *F, int *path)
{
    const int64_t L = T + pad_left + pad_right;
    TFX_CHECK(K >= 1 && L >= K, "ols_plan_info: kernel size %lld larger than the padded signal %lld", (long long)K, (long long)L);
    TFX_CHECK(dtype == TFX_F32 || dtype == TFX_F64, "ols_plan_info: bad dtype %d", dtype);
    int64_t n = 0, hop = 0, frames = 0;
    int p = 0;
    if (olslds_supported(K, dtype, L, &n)) {
        int64_t lead = 0;
        olslds_geometry(K, T, pad_left, pad_right, dtype == TFX_F32 ? 4 : 8, n, &lead, &hop);
        p = 2;
    } else if (dtype == TFX_F32 && olsnative_supported(K, L, &n)) {
        olsnative_geometry(K, T, pad_left, pad_right, n, &hop, &frames);
        p = 1;
    } else if (dtype == TFX_F64 && olsnative64_supported(K, L, false)) {
        n = (int64_t)1 << 20;
        olsnative64_geometry(K, T, pad_left, pad_right, &hop, &frames);
        p = 1;
    } else {
        n = fftconv_block_size(K, L);
        hop = n - K + 1;
    }
    if (N) *N = n;
    if (S) *S = hop;
    if (F) *F = frames > 0 ? frames : ceil_div(L - K + 1, hop);
    if (path) *path = p;
}

int tfx_ols_plan_info(int64_t K, int64_t T, int64_t pad_left, int64_t pad_right, int64_t *N, int64_t *S,
                      int64_t *F, int *native)
{
    TFX_API_BEGIN
    int path = 0;
    ols_plan(K, T, pad_left, pad_right, TFX_F32, N, S, F, &path);
    if (native) *native = path != 0;
    TFX_API_END
}

int tfx_ols_plan_info2(int64_t K, int64_t T, int64_t pad_left, int64_t pad_right, int dtype, int64_t *N, int64_t *S,
                       int64_t *F, int *path)
{
    TFX_API_BEGIN
    ols_plan(K, T, pad_left, pad_right, dtype, N, S, F, path);
    TFX_API_END
}

int tfx_env_reload(void)
{
    TFX_API_BEGIN
    env_reload();
    TFX_API_END
}

int tfx_prewarm(void)
{
    TFX_API_BEGIN
    olsnative_prewarm();
    TFX_API_END
}

int tfx_delay_line_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, int64_t delay, double decay,
                           double mix, tfx_stream_t stream)
{
    TFX_API_BEGIN
    TFX_CHECK(dtype == TFX_F32 || dtype == TFX_F64, "delay_line_forward: bad dtype");
    TFX_CHECK(delay >= 0, "delay_line_forward: negative delay");
    delay_line_forward(x, y, dtype, C, T, delay, mix * decay, (hipStream_t)stream);
    TFX_API_END
}

int tfx_sum_forward(const void *const *xs_host, int n, void *y, int dtype, int64_t numel, tfx_stream_t stream)
{
    TFX_API_BEGIN
    sum_forward(xs_host, n, y, dtype, numel, (hipStream_t)stream);
    TFX_API_END
}

int tfx_gain_forward(const void *x, void *y, int dtype, int64_t numel, double gain, int clamp, tfx_stream_t stream)
{
    TFX_API_BEGIN
    gain_forward(x, y, dtype, numel, gain, clamp, (hipStream_t)stream);
    TFX_API_END
}

int tfx_stat_forward(const void *x, int dtype, int64_t C, int64_t T, int mode, int per_row, double *out_dev,
                     tfx_stream_t stream)
{
    TFX_API_BEGIN
    stat_forward(x, dtype, C, T, mode, per_row, out_dev, (hipStream_t)stream);
    TFX_API_END
}

int tfx_normalize_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, int mode, int per_row,
                          double peak, tfx_stream_t stream)
{
    TFX_API_BEGIN
    normalize_forward(x, y, dtype, C, T, mode, per_row, peak, (hipStream_t)stream);
    TFX_API_END
}

int tfx_deinterleave_forward(const void *in, int in_kind, void *out, int64_t F, int64_t C, int64_t ld_out,
                             int64_t f_base, double scale, tfx_stream_t stream)
{
    TFX_API_BEGIN
    deinterleave_forward(in, in_kind, (float *)out, F, C, ld_out, f_base, scale, (hipStream_t)stream);
    TFX_API_END
}

int tfx_interleave_forward(const void *in, void *out, int64_t F, int64_t C, int64_t ld_in, int64_t f_base,
                           tfx_stream_t stream)
{
    TFX_API_BEGIN
    interleave_forward((const float *)in, (float *)out, F, C, ld_in, f_base, (hipStream_t)stream);
    TFX_API_END
}

int tfx_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return 0;
}

const char *tfx_prof_collect(void)
{
    static std::string out;
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::map<std::string, std::pair<int, double>> agg;
    std::vector<std::string> order;
    for (auto &r : g_prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            if (!agg.count(r.name)) order.push_back(r.name);
            agg[r.name].first += 1;
            agg[r.name].second += ms;
        }
    }
    g_prof.clear();
    for (hipEvent_t e : g_used) g_pool.push_back(e);
    g_used.clear();
    g_last_end = nullptr;
    g_chain = false;
    out = "{";
    bool first = true;
    for (auto &n : order) {
        char b[256];
        snprintf(b, sizeof(b), "%s\"%s\": {\"calls\": %d, \"total_ms\": %.6f}", first ? "" : ", ", n.c_str(),
                 agg[n].first, agg[n].second);
        out += b;
        first = false;
    }
    out += "}";
    return out.c_str();
}

int tfx_set_workspace_allocator(tfx_alloc_fn alloc_fn, tfx_free_fn free_fn, void *ctx)
{
    TFX_API_BEGIN
    TFX_CHECK((alloc_fn == nullptr) == (free_fn == nullptr), "tfx_set_workspace_allocator: give both functions or neither");
    scratch_set_allocator(alloc_fn, free_fn, ctx);       // buffers already held keep the allocator they came from
    TFX_API_END
}

int64_t tfx_workspace_bytes(void) { return (int64_t)scratch_bytes(); }

int tfx_clear_caches(void)
{
    TFX_API_BEGIN
    sos_clear_plans();
    fir_clear();
    fftconv_clear();
    olsnative_clear();
    olsnative64_clear();
    olslds_clear();
    scratch_clear();
    TFX_API_END
}

}  // extern "C"
