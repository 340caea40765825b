// common.h -- shared host-side plumbing for libtorchfx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace tfx {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

void set_last_error(const std::string &msg);

#define TFX_CHECK(cond, ...)                                                        \
    do {                                                                            \
        if (!(cond)) {                                                              \
            char _b[512];                                                           \
            snprintf(_b, sizeof(_b), __VA_ARGS__);                                  \
            throw ::tfx::Error(_b);                                                 \
        }                                                                           \
    } while (0)

#define TFX_HIP(expr)                                                               \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            char _b[512];                                                           \
            snprintf(_b, sizeof(_b), "HIP error %s at %s:%d (%s)",                  \
                     hipGetErrorString(_e), __FILE__, __LINE__, #expr);             \
            throw ::tfx::Error(_b);                                                 \
        }                                                                           \
    } while (0)

// ---- per-kernel timing (bench.py's roofline leg) --------------------------------
// When enabled, prof_begin/prof_end bracket a launch with hipEvents on the launch
// stream; collect() synchronises and aggregates by name.
bool prof_on();
void prof_begin(const char *name, hipStream_t s);
void prof_end(hipStream_t s);
void prof_break_chain();   // called at every API entry: events are only shared inside one call

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(const char *name, hipStream_t st) : s(st), on(prof_on()) {
        if (on) prof_begin(name, s);
    }
    ~ProfScope() {
        if (on) prof_end(s);
    }
};

// ---- stream-ordered device scratch ------------------------------------------------
// A tiny cache of device buffers keyed by (tag, stream): grown on demand and reused by later calls
// on the SAME stream (stream order makes the reuse safe); calls on different streams never share a
// buffer.  Host-side enqueueing is serialised by the API lock (capi.hip), so two threads can drive
// two streams concurrently.
void *scratch(const char *tag, size_t bytes, hipStream_t stream);        // throws when the allocation fails
void *scratch_try(const char *tag, size_t bytes, hipStream_t stream);    // null when it fails (callers that can do with less)
size_t scratch_bytes();                                                  // bytes held right now, all tags / streams / devices
void scratch_clear();

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- tuning knobs from the environment (TFX_*) --------------------------------------------------
// Read ONCE per process and name (capi.hip): a dispatch asks for about ten of them and a short-row call lasts 8-30 us.
// The table is keyed by the knob's text (a per-thread, pointer-keyed shortcut in front of it is checked against the text, so
// any pointer may ask).  A knob changed after its first read is NOT seen -- development and tests that flip knobs inside one
// process set TFX_ENV_DYNAMIC=1 (read when the library is first used: every lookup becomes a fresh getenv) or call
// tfx_env_reload() / torchfx_ext.env_reload(), which drops the table.
int64_t env_i64(const char *name, int64_t dflt);
void env_reload();

// Streaming (nontemporal) 16-byte global accesses for data a pass touches exactly once: a linear float32 sweep runs at
// 6.80 instead of 6.29 TB/s with them on MI355X (tools/ubench/stream_copy2.hip, profiles/r03_experiments.txt).
typedef unsigned tfx_u32x4 __attribute__((ext_vector_type(4)));
template <typename V> __device__ __forceinline__ V ldg16_stream(const void *p)
{
    static_assert(sizeof(V) == 16, "16-byte vector expected");
    const tfx_u32x4 r = __builtin_nontemporal_load((const tfx_u32x4 *)p);
    return __builtin_bit_cast(V, r);
}
template <typename V> __device__ __forceinline__ void stg16_stream(void *p, const V &v)
{
    static_assert(sizeof(V) == 16, "16-byte vector expected");
    __builtin_nontemporal_store(__builtin_bit_cast(tfx_u32x4, v), (tfx_u32x4 *)p);
}

// olsnative.hip: a planner-built (zero-state) SOS cascade that runs inside the forward column pass of the three-pass
// overlap-save pipeline (ols_col_fwd16_sos_kernel)
struct SosFuseHost {
    const double *sos;      // host [K, 6] rows b0 b1 b2 a0 a1 a2 (a0 ignored, iir_cpu.cpp:86)
    int64_t K;
    int64_t warm;           // samples after which the cascade has forgotten its past to float64 round-off (sos_plan_info)
    double *sections;       // optional DEVICE [K, C, T] float64: every section's output, or null
};

// fir.hip: device copy of a host tap vector, cached by content and device (uploaded, blocking, the first time a filter is seen)
const void *cached_taps(const void *host, size_t bytes, size_t padded);

// Every device-side cache (plans, taps, spectra, scratch, internal streams, occupancy answers) is
// keyed by the ordinal of the device that is current at the call: one process may drive several GPUs
// (the Python layer wraps each op in torch.cuda.device(x.device)).
constexpr int TFX_MAX_DEVICES = 64;
inline int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;      // host-only planning calls work without a device
    TFX_CHECK(dev >= 0 && dev < TFX_MAX_DEVICES, "device ordinal %d out of range", dev);
    return dev;
}

}  // namespace tfx
