// layout.hip -- the data-format edge either side of the filter path.
//
// Audio files decode to INTERLEAVED frames [F, C] (soundfile / libsndfile; the reference then does
// `data_np.T.copy()` on the host, src/torchfx/wave.py:448-452, and `.numpy().T` again before writing,
// :566-573); every kernel here works on PLANAR rows [C, F].  Doing the transposition on the device
// lets the host hand over the decoder's buffer as it is (pinned, chunked, asynchronous H2D) and, for
// 16-bit PCM, move half the bytes over PCIe and convert on the GPU (x / 32768, libsndfile's
// normalisation).  Both directions are pure HBM-bound transposes through a padded LDS tile.
#include "common.h"
#include "../../include/torchfx_hip.h"

namespace tfx {

constexpr int LAY_THREADS = 256;
constexpr int LAY_TILE_ELEMS = 8192;          // elements per workgroup tile (32 KB of float in LDS)

template <typename TIn> __device__ __forceinline__ float lay_cvt(TIn v, float scale);
template <> __device__ __forceinline__ float lay_cvt<float>(float v, float) { return v; }
template <> __device__ __forceinline__ float lay_cvt<short>(short v, float scale) { return (float)v * scale; }

// in: [F, C] interleaved (TIn), out: [C, ld_out] planar float, frames [f_base, f_base + F) of out's rows
template <typename TIn>
__global__ void __launch_bounds__(LAY_THREADS)
deinterleave_kernel(const TIn *__restrict__ in, float *__restrict__ out, int64_t F, int C, int64_t ld_out,
                    int64_t f_base, int ft, float scale)
{
    extern __shared__ float lay_lds[];
    const int stride = C | 1;                 // odd row pitch: the strided reads below are conflict-free
    const int64_t f0 = (int64_t)blockIdx.x * ft;
    const int nf = (int)((F - f0 < ft) ? (F - f0) : ft);
    const int64_t e0 = f0 * C;
    const int ne = nf * C;
    // coalesced read of the tile as it lies in memory
    for (int e = threadIdx.x; e < ne; e += LAY_THREADS) {
        const int f = e / C, c = e - f * C;
        lay_lds[f * stride + c] = lay_cvt<TIn>(in[e0 + e], scale);
    }
    __syncthreads();
    // one channel row at a time, consecutive lanes = consecutive frames: coalesced writes
    for (int i = threadIdx.x; i < C * nf; i += LAY_THREADS) {
        const int c = i / nf, f = i - c * nf;
        out[(int64_t)c * ld_out + f_base + f0 + f] = lay_lds[f * stride + c];
    }
}

// in: [C, ld_in] planar float (frames [f_base, f_base + F)), out: [F, C] interleaved float
__global__ void __launch_bounds__(LAY_THREADS)
interleave_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t F, int C, int64_t ld_in,
                  int64_t f_base, int ft)
{
    extern __shared__ float lay_lds[];
    const int stride = C | 1;
    const int64_t f0 = (int64_t)blockIdx.x * ft;
    const int nf = (int)((F - f0 < ft) ? (F - f0) : ft);
    for (int i = threadIdx.x; i < C * nf; i += LAY_THREADS) {
        const int c = i / nf, f = i - c * nf;
        lay_lds[f * stride + c] = in[(int64_t)c * ld_in + f_base + f0 + f];
    }
    __syncthreads();
    const int64_t e0 = f0 * C;
    const int ne = nf * C;
    for (int e = threadIdx.x; e < ne; e += LAY_THREADS) {
        const int f = e / C, c = e - f * C;
        out[e0 + e] = lay_lds[f * stride + c];
    }
}

static int frames_per_tile(int64_t C)
{
    int ft = (int)(LAY_TILE_ELEMS / C);
    return ft < 1 ? 1 : ft;
}

void deinterleave_forward(const void *in, int in_kind, float *out, int64_t F, int64_t C, int64_t ld_out, int64_t f_base,
                          double scale, hipStream_t stream)
{
    TFX_CHECK(in_kind == 0 || in_kind == 1, "deinterleave_forward: input kind must be 0 (float32) or 1 (int16)");
    TFX_CHECK(C >= 1 && C <= 4096, "deinterleave_forward: 1..4096 channels, got %lld", (long long)C);
    TFX_CHECK(F >= 0 && ld_out >= f_base + F && f_base >= 0, "deinterleave_forward: output rows too short");
    if (F == 0) return;
    TFX_CHECK(in && out, "deinterleave_forward: null pointer");
    const int ft = frames_per_tile(C);
    const int64_t grid = ceil_div(F, (int64_t)ft);
    TFX_CHECK(grid < (1ll << 31), "deinterleave_forward: grid too large");
    const size_t shm = (size_t)ft * (size_t)((int)C | 1) * sizeof(float);
    ProfScope ps("deinterleave_kernel", stream);
    if (in_kind == 0)
        hipLaunchKernelGGL(deinterleave_kernel<float>, dim3((unsigned)grid), dim3(LAY_THREADS), shm, stream,
                           (const float *)in, out, F, (int)C, ld_out, f_base, ft, 1.0f);
    else
        hipLaunchKernelGGL(deinterleave_kernel<short>, dim3((unsigned)grid), dim3(LAY_THREADS), shm, stream,
                           (const short *)in, out, F, (int)C, ld_out, f_base, ft, (float)scale);
    TFX_HIP(hipGetLastError());
}

void interleave_forward(const float *in, float *out, int64_t F, int64_t C, int64_t ld_in, int64_t f_base,
                        hipStream_t stream)
{
    TFX_CHECK(C >= 1 && C <= 4096, "interleave_forward: 1..4096 channels, got %lld", (long long)C);
    TFX_CHECK(F >= 0 && ld_in >= f_base + F && f_base >= 0, "interleave_forward: input rows too short");
    if (F == 0) return;
    TFX_CHECK(in && out, "interleave_forward: null pointer");
    const int ft = frames_per_tile(C);
    const int64_t grid = ceil_div(F, (int64_t)ft);
    TFX_CHECK(grid < (1ll << 31), "interleave_forward: grid too large");
    const size_t shm = (size_t)ft * (size_t)((int)C | 1) * sizeof(float);
    ProfScope ps("interleave_kernel", stream);
    hipLaunchKernelGGL(interleave_kernel, dim3((unsigned)grid), dim3(LAY_THREADS), shm, stream, in, out, F, (int)C, ld_in,
                       f_base, ft);
    TFX_HIP(hipGetLastError());
}

}  // namespace tfx
