// select.hip -- the percentile of |x| on the device (PercentileNormalizationStrategy, src/torchfx/effect.py:723-755:
// `torch.quantile(torch.abs(waveform), p / 100, interpolation="linear")`, then x / threshold * peak).
//
// The reference's torch.quantile sorts (and refuses inputs above 16 M elements); a percentile is a SELECTION, so here it is
// a three-level radix select on the bit patterns of |x| -- non-negative floats order like unsigned integers -- : histogram of
// bits 30..19 over the whole signal, pick the bin that holds the wanted rank, histogram of bits 18..7 inside that bin, then of
// bits 6..0: three streaming passes of 4 B/sample, no sort, no size limit, no host synchronisation (the threshold stays on
// the device for the apply pass).  Both order statistics the linear interpolation needs (floor and ceil of q (n - 1)) are
// selected in the same passes (two histograms per level).  Rank arithmetic and the interpolation are done in float32 exactly
// as ATen does them (quantile_compute: ranks = q * (n - 1) in the input dtype, lerp(below, above, ranks - floor(ranks))), so
// the result equals torch.quantile's wherever torch.quantile runs.  NaN anywhere -> NaN, like torch.
#include <cstddef>
#include "common.h"
#include "../../include/torchfx_hip.h"

namespace tfx {

constexpr int SEL_B1 = 4096, SEL_B2 = 4096, SEL_B3 = 128;       // bins per level: bits 30..19, 18..7, 6..0

struct SelState {                 // device-side state of one selection
    unsigned long long h1[SEL_B1];
    unsigned long long h2[2][SEL_B2];
    unsigned long long h3[2][SEL_B3];
    unsigned long long nan_count;
    unsigned long long rank[2];   // remaining rank inside the current prefix, for the two order statistics
    unsigned prefix[2];           // key bits fixed so far (right-aligned)
};

__device__ __forceinline__ unsigned abs_key(float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; }

// LEVEL 1: all elements, one histogram (bits 30..19) + NaN count.  LEVEL 2 / 3: two histograms, elements whose higher bits equal
// prefix[w].  Workgroup-private LDS histograms, merged into the global one with 64-bit atomics.
template <int LEVEL>
__global__ void __launch_bounds__(256) select_hist_kernel(const float *__restrict__ x, int64_t n, SelState *st)
{
    constexpr int NB = LEVEL == 1 ? SEL_B1 : (LEVEL == 2 ? SEL_B2 : SEL_B3);
    constexpr int NH = LEVEL == 1 ? 1 : 2;
    __shared__ unsigned lh[NH][NB];
    __shared__ unsigned lnan;
    const int tid = threadIdx.x;
    for (int i = tid; i < NH * NB; i += 256) (&lh[0][0])[i] = 0;
    if (tid == 0) lnan = 0;
    unsigned p0 = 0, p1 = 0;
    if (LEVEL > 1) { p0 = st->prefix[0]; p1 = st->prefix[1]; }
    __syncthreads();
    const int64_t n4 = n / 4;
    const bool al = (((uintptr_t)x) & 15) == 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < (al ? n4 : 0); i += (int64_t)gridDim.x * 256) {
        const float4 v = ldg16_stream<float4>(x + 4 * i);
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned u = abs_key(e[k]);
            if (LEVEL == 1) {
                atomicAdd(&lh[0][u >> 19], 1u);
                if (u > 0x7f800000u) atomicAdd(&lnan, 1u);
            } else if (LEVEL == 2) {
                if ((u >> 19) == p0) atomicAdd(&lh[0][(u >> 7) & 4095u], 1u);
                if ((u >> 19) == p1) atomicAdd(&lh[1][(u >> 7) & 4095u], 1u);
            } else {
                if ((u >> 7) == p0) atomicAdd(&lh[0][u & 127u], 1u);
                if ((u >> 7) == p1) atomicAdd(&lh[1][u & 127u], 1u);
            }
        }
    }
    // tail (and unaligned inputs): scalar
    for (int64_t i = (al ? 4 * n4 : 0) + (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned u = abs_key(x[i]);
        if (LEVEL == 1) {
            atomicAdd(&lh[0][u >> 19], 1u);
            if (u > 0x7f800000u) atomicAdd(&lnan, 1u);
        } else if (LEVEL == 2) {
            if ((u >> 19) == p0) atomicAdd(&lh[0][(u >> 7) & 4095u], 1u);
            if ((u >> 19) == p1) atomicAdd(&lh[1][(u >> 7) & 4095u], 1u);
        } else {
            if ((u >> 7) == p0) atomicAdd(&lh[0][u & 127u], 1u);
            if ((u >> 7) == p1) atomicAdd(&lh[1][u & 127u], 1u);
        }
    }
    __syncthreads();
    unsigned long long *g0 = LEVEL == 1 ? st->h1 : (LEVEL == 2 ? st->h2[0] : st->h3[0]);
    unsigned long long *g1 = LEVEL == 2 ? st->h2[1] : st->h3[1];
    for (int i = tid; i < NB; i += 256) {
        if (lh[0][i]) atomicAdd(&g0[i], (unsigned long long)lh[0][i]);
        if (NH == 2 && lh[NH - 1][i]) atomicAdd(&g1[i], (unsigned long long)lh[NH - 1][i]);
    }
    if (LEVEL == 1 && tid == 0 && lnan) atomicAdd(&st->nan_count, (unsigned long long)lnan);
}

// Clears the histograms and sets the two wanted ranks: a launch instead of memset + a host-to-device copy of a stack variable,
// so that the whole selection is stream-ordered, never blocks the host and can be captured into a HIP graph
__global__ void __launch_bounds__(256) select_init_kernel(SelState *st, unsigned long long rank0, unsigned long long rank1)
{
    unsigned long long *w = (unsigned long long *)st;
    constexpr int NW = (int)(offsetof(SelState, nan_count) / sizeof(unsigned long long));
    for (int i = blockIdx.x * 256 + threadIdx.x; i < NW; i += gridDim.x * 256) w[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->nan_count = 0;
        st->rank[0] = rank0; st->rank[1] = rank1;
        st->prefix[0] = st->prefix[1] = 0;
    }
}

// One workgroup: for each of the two order statistics find the bin of this level that holds its remaining rank, append the
// bin to its prefix and reduce the rank.  After level 3 the prefixes are the full 31-bit keys: interpolate and write the result.
template <int LEVEL>
__global__ void __launch_bounds__(1024) select_pick_kernel(SelState *st, float weight, double *out)
{
    constexpr int NB = LEVEL == 1 ? SEL_B1 : (LEVEL == 2 ? SEL_B2 : SEL_B3);
    constexpr int BITS = LEVEL == 3 ? 7 : 12;
    constexpr int PER = (NB + 1023) / 1024;
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    for (int w = 0; w < 2; ++w) {
        const unsigned long long *h = LEVEL == 1 ? st->h1 : (LEVEL == 2 ? st->h2[w] : st->h3[w]);
        unsigned long long loc[PER], sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) { const int b = tid * PER + i; loc[i] = b < NB ? h[b] : 0; sum += loc[i]; }
        part[tid] = sum;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {                  // inclusive scan over the threads' sums
            const unsigned long long v = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        const unsigned long long before = part[tid] - sum, k = st->rank[w];
        __syncthreads();
        if (k >= before && k < before + sum) {                       // exactly one thread (the histogram holds > k elements)
            unsigned long long acc = before;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                if (k < acc + loc[i]) {
                    st->prefix[w] = (LEVEL == 1 ? 0u : (st->prefix[w] << BITS)) | (unsigned)(tid * PER + i);
                    st->rank[w] = k - acc;
                    break;
                }
                acc += loc[i];
            }
        }
        __syncthreads();
    }
    if (LEVEL == 3 && tid == 0) {
        __threadfence_block();
        const float below = __builtin_bit_cast(float, st->prefix[0]), above = __builtin_bit_cast(float, st->prefix[1]);
        // at::lerp for float: a + w (b - a) for w < 0.5, b - (b - a) (1 - w) otherwise
        const float diff = above - below;
        const float r = weight < 0.5f ? below + weight * diff : above - diff * (1.0f - weight);
        out[0] = st->nan_count ? (double)__builtin_nanf("") : (double)r;
    }
}

// out_dev[0] (float64) = quantile_q(|x|) over all n elements with linear interpolation, computed like torch.quantile on float32
void quantile_abs_forward(const float *x, int64_t n, double q, double *out_dev, hipStream_t stream)
{
    TFX_CHECK(n >= 1 && x && out_dev, "quantile_abs: empty input or null pointer");
    TFX_CHECK(q >= 0.0 && q <= 1.0, "quantile_abs: q = %g outside [0, 1]", q);
    SelState *st = (SelState *)scratch("select_state", sizeof(SelState), stream);
    // ATen (quantile_compute): ranks = q * (n - 1) in the input dtype
    const float ranks = (float)q * (float)(n - 1);
    float below = floorf(ranks), above = ceilf(ranks);
    const float weight = ranks - below;
    auto clampi = [&](float v) { int64_t r = (int64_t)v; return r < 0 ? (int64_t)0 : (r > n - 1 ? n - 1 : r); };
    hipLaunchKernelGGL(select_init_kernel, dim3(16), dim3(256), 0, stream, st, (unsigned long long)clampi(below), (unsigned long long)clampi(above));
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(n, 256 * 16), 256 * 8);
    ProfScope ps("select_hist_kernel", stream);
    hipLaunchKernelGGL(select_hist_kernel<1>, dim3(grid), dim3(256), 0, stream, x, n, st);
    hipLaunchKernelGGL(select_pick_kernel<1>, dim3(1), dim3(1024), 0, stream, st, weight, out_dev);
    hipLaunchKernelGGL(select_hist_kernel<2>, dim3(grid), dim3(256), 0, stream, x, n, st);
    hipLaunchKernelGGL(select_pick_kernel<2>, dim3(1), dim3(1024), 0, stream, st, weight, out_dev);
    hipLaunchKernelGGL(select_hist_kernel<3>, dim3(grid), dim3(256), 0, stream, x, n, st);
    hipLaunchKernelGGL(select_pick_kernel<3>, dim3(1), dim3(1024), 0, stream, st, weight, out_dev);
    TFX_HIP(hipGetLastError());
}

}  // namespace tfx
