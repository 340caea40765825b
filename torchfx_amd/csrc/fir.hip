// fir.hip -- direct-form causal depthwise FIR for gfx950.
//
// Replaces the conv_mode="direct" branch of FIR.forward (src/torchfx/filter/fir.py:556-568:
// F.pad(x,(K-1,0)) + F.conv1d(groups=C) with one shared flipped kernel):
//     y[c,n] = sum_{t<K} kf[t] * xp[c, n+t],   xp[m] = x[m-(K-1)]  (zero for m < K-1)
//
// float32: 2K flop/sample (2048 at K=1024) makes this FP32-rate bound, not HBM bound.  gfx950 has
// an exact-f32 matrix instruction, v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain, same peak as
// the vector ALU but one instruction per 4096 flop), so the convolution is phrased as a
// Toeplitz product on the matrix pipe -- NOT to "reach MFMA" with reduced precision, the
// arithmetic is plain f32 FMA:
//     out(i,j) = y[nb + 32 i + j] = sum_s  A[i][s] * B[s][j]
//     A[i][s] = xw[32 i + s]          (signal window, LDS, rows padded 32->33 floats)
//     B[s][j] = kpad[s - j + 31]      (taps, Toeplitz, LDS, 31 zeros in front)
// Each wave owns NJ = 4 output tiles of 32x32 = 1024 consecutive samples and re-uses every B
// fragment across them; a workgroup (4 waves) covers 16384 outputs of one channel per pass and
// walks the taps in chunks of <= 1024 so the LDS window stays bounded for any K.
//
// float64 signals (the reference computes conv1d in the input dtype) use a plain LDS-tiled
// vector kernel: rare path, correctness first.
#include "common.h"
#include "../../include/torchfx_hip.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace tfx {

// device copies of tap vectors, keyed by content: uploaded (blocking) the first time a filter
// is seen, then reused -- no host sync on the steady-state path.
static std::mutex g_taps_mu;
static std::map<std::vector<char>, void *> g_taps;
static const std::vector<char> *g_taps_last_key[TFX_MAX_DEVICES] = {};   // per device: the entry used last
static const void *g_taps_last[TFX_MAX_DEVICES] = {};
const void *cached_taps(const void *host, size_t bytes, size_t padded)
{
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(g_taps_mu);
    // steady state (the same filter call after call): one memcmp, no key construction -- a long tap vector would
    // otherwise cost a heap allocation of its size per call
    if (const std::vector<char> *lk_ = g_taps_last_key[dev]) {
        if (lk_->size() == bytes + 2 && memcmp(lk_->data(), host, bytes) == 0 && (*lk_)[bytes] == (char)(padded & 0xff))
            return g_taps_last[dev];
    }
    std::vector<char> key((const char *)host, (const char *)host + bytes);
    key.push_back((char)(padded & 0xff));
    key.push_back((char)dev);
    auto it = g_taps.find(key);
    if (it != g_taps.end()) {
        g_taps_last_key[dev] = &it->first;
        g_taps_last[dev] = it->second;
        return it->second;
    }
    if (g_taps.size() > 128) {
        (void)hipDeviceSynchronize();
        for (auto &kv : g_taps) (void)hipFree(kv.second);
        g_taps.clear();
        for (int d2 = 0; d2 < TFX_MAX_DEVICES; ++d2) { g_taps_last_key[d2] = nullptr; g_taps_last[d2] = nullptr; }
    }
    std::vector<char> h(padded, 0);
    memcpy(h.data(), host, bytes);
    void *d = nullptr;
    TFX_HIP(hipMalloc(&d, padded));
    TFX_HIP(hipMemcpy(d, h.data(), padded, hipMemcpyHostToDevice));
    auto ins = g_taps.emplace(std::move(key), d).first;
    g_taps_last_key[dev] = &ins->first;                 // std::map nodes are stable
    g_taps_last[dev] = d;
    return d;
}
void fir_clear()
{
    std::lock_guard<std::mutex> lk(g_taps_mu);
    (void)hipDeviceSynchronize();
    for (auto &kv : g_taps) (void)hipFree(kv.second);
    g_taps.clear();
    for (int d = 0; d < TFX_MAX_DEVICES; ++d) { g_taps_last_key[d] = nullptr; g_taps_last[d] = nullptr; }
}

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int FIR_NJ = 4;                       // 32x32 tiles per wave
constexpr int FIR_WOUT = FIR_NJ * 1024;         // outputs per wave
constexpr int FIR_NOUT = 4 * FIR_WOUT;          // outputs per workgroup
constexpr int FIR_KC_MAX = 1024;                // largest tap chunk (taps are zero-padded to it)

__device__ __forceinline__ int xpad33(int m) { return m + (m >> 5); }

// kf_dev: [Kpad] flipped taps on device, zero-padded to a multiple of FIR_KC_MAX
// FIR_KC: taps per chunk (128 / 512 / 1024 -- short filters do not pay for 1024-tap chunks).
// DBG (tools/ubench/fir_probe.hip only): bit 0 drops the output stores, bit 1 the global loads.
// NJ = 32x32 output tiles per wave (TFX_FIR_NJ): 4 = 16384 outputs and a 76 KB window per workgroup, two workgroups per
// CU (rounds 1-2); 2 = 8192 outputs, 43 KB, three per CU; 1 = 4096 outputs, 26 KB, five per CU (default).  With two
// workgroups per CU, both fill their window, multiply and store in lockstep, so the matrix pipe idles during every fill
// and store; more, smaller workgroups interleave those phases: cfg 3 2.86 / 2.77 / 2.73 ms for NJ = 4 / 2 / 1 on one box
// (profiles/r03_fir.txt; NJ = 1 at six waves per SIMD spills 36 B and is back to 2.78).
template <int FIR_KC, int DBG = 0, int NJ = FIR_NJ>
__global__ void __launch_bounds__(256, NJ == 4 ? 2 : (NJ == 2 ? 3 : 5))
fir_direct_mfma_kernel(const float *__restrict__ x, float *__restrict__ y,
                       const float *__restrict__ kf_dev, int64_t C, int64_t T, int K, int nchunks,
                       int64_t tiles_per_row, const float *__restrict__ hist, int H)
{
    // hist (streaming, StatefulFIR): [C, H] samples that precede the row, x[-H .. -1]; null = zeros
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WOUT = NJ * 1024, NOUT = 4 * WOUT;
    constexpr int XW = NOUT + FIR_KC + 32;              // window length (floats)
    constexpr int XW_PAD = XW + (XW >> 5) + 1;
    float *xw = (float *)smem;                              // [XW_PAD]
    float *kp = xw + ((XW_PAD + 3) & ~3);                   // [31 + FIR_KC + 33]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Tile -> workgroup: workgroups are dealt to the eight XCDs round robin (observed; only speed and traffic depend on it), so
    // XCD b % 8 takes the tiles [xcd * per_xcd, (xcd + 1) * per_xcd) in order: neighbouring tiles of a row, whose windows share
    // K - 1 + 32 samples, are worked on by ONE XCD and the overlap is an L2 hit instead of a second fetch (round 4: PMC read
    // traffic of cfg 3 1.2 x -> see profiles/r04_traffic.json).
    const int64_t ntiles = C * tiles_per_row, per_xcd = (ntiles + 7) / 8;
    const int64_t bid = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || bid >= ntiles) return;
    // (64-bit divisions run on the vector unit: the wave-uniform quotient goes back to scalar registers, or row bases and tile
    // offsets sit in VGPR pairs for the whole tile -- the kernel spilled one register at its 96-VGPR budget: 46 MB of scratch
    // writes per cfg-3 call in the PMC write traffic)
    const auto uni64 = [](int64_t v) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)v);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    const int64_t c = uni64(bid / tiles_per_row);
    const int64_t n0 = (bid - c * tiles_per_row) * (int64_t)NOUT;
    const float *xrow = x + c * T;
    float *yrow = y + c * T;

    floatx16 acc[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int li = lane & 31, kk = lane >> 5;

    // Per-lane LDS bases.  With s = 32*blk + 2*q + kk the padded window index of A[i=li][s] for
    // tile t is  (xb + xb/32 + 33*li + kk) + 1056*t + 33*blk + 2*q  and the Toeplitz tap index is
    // (31 - li + kk) + 32*blk + 2*q : everything but `blk` is an immediate ds_read offset.
    const int xb = wave * WOUT;
    const float *pa0 = xw + xb + (xb >> 5) + 33 * li + kk;
    const float *pb0 = kp + 31 - li + kk;
    constexpr int XV = (XW + 255) / 256;                    // window loads per thread
    constexpr int KV = (31 + FIR_KC + 33 + 255) / 256;      // tap loads per thread

    for (int ch = 0; ch < nchunks; ++ch) {
        const int t0 = ch * FIR_KC;
        // window: xw[m] = xp[n0 + t0 + m] = x[n0 + t0 + m - (K-1)].  All loads are issued before
        // the first LDS write (branch-free: clamped address + select), so the fill costs one
        // memory latency instead of XV of them.
        const int64_t base = n0 + t0 - (int64_t)(K - 1);
        float xv[XV], kv[KV];
        // Buffer descriptor over the part of the window that lies inside the row: reads past
        // its end return 0 in hardware; elements before the row start are clamped to offset 0
        // here and zeroed at the LDS write.  32-bit offsets, no per-load address pairs.
        const int64_t gstart = base < 0 ? 0 : (base > T ? T : base);
        const int64_t gend0 = base + (int64_t)XV * 256;
        const int64_t gend = gend0 < gstart ? gstart : (gend0 > T ? T : gend0);
        const int shift = (int)(base - gstart);             // <= 0; 0 for every interior tile
        // (the 64-bit divide that produced `c` ran on the vector ALU: tell the compiler the
        // descriptor words are wave-uniform, or every load becomes a waterfall loop)
        const uint64_t pw = (uint64_t)(xrow + gstart);
        const uint64_t pu = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(pw >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pw);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)pu, 0, __builtin_amdgcn_readfirstlane((int)((gend - gstart) * 4)), 0x00020000);
        // Interior tiles (window entirely inside the row: all but the first and last tile or two of a row): ONE
        // per-lane offset and a scalar offset per load -- no 68 address registers (the per-load clamped offsets of
        // the general form made this kernel spill 292 VGPRs = 4.6 x the output bytes of scratch traffic).  The scalar
        // offset is not part of the hardware range check, so tiles that touch either end of the row take the
        // partially unrolled loop below, where the range check and the clamp do their work.
        const bool interior = (shift == 0) && (gend0 <= T);  // workgroup-uniform
        if (interior) {
#pragma unroll
            for (int r = 0; r < XV; ++r)
                xv[r] = (DBG & 2) ? 1.0f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, tid * 4, r * 1024, 0));
        }
#pragma unroll
        for (int r = 0; r < KV; ++r) {
            const int v = tid + 256 * r - 31;               // kf_dev is zero-padded to a multiple of KC
            const int vc = v < 0 ? 0 : (v >= FIR_KC ? FIR_KC - 1 : v);
            kv[r] = kf_dev[t0 + vc];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep every load above, every use below
        __syncthreads();    // previous chunk's readers are done
        if (interior) {
#pragma unroll
            for (int r = 0; r < XV; ++r) {
                const int m = tid + 256 * r;
                if (m < XW) xw[xpad33(m)] = xv[r];
            }
        } else {
#pragma unroll 2
            for (int r = 0; r < XV; ++r) {
                const int m = tid + 256 * r;
                const int e = shift + m;
                float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (e < 0 ? 0 : e) * 4, 0, 0));
                if (e < 0) {                                 // before the row start (first tiles only)
                    const int64_t g = base + m;              // < 0
                    v = (hist && g >= -(int64_t)H) ? hist[c * H + H + g] : 0.0f;
                }
                if (m < XW) xw[xpad33(m)] = v;
            }
        }
#pragma unroll
        for (int r = 0; r < KV; ++r) {
            const int u = tid + 256 * r;
            if (u < 31 + FIR_KC + 33) kp[u] = (u >= 31 && u < 31 + FIR_KC) ? kv[r] : 0.0f;
        }
        __syncthreads();

        // contraction over s in [0, KC+32) in blocks of 32 (16 k-steps of 2)
        for (int blk = 0; blk < (FIR_KC + 32) / 32; ++blk) {
            const float *pa = pa0 + 33 * blk;
            const float *pb = pb0 + 32 * blk;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float b = pb[2 * q];
#pragma unroll
                for (int t = 0; t < NJ; ++t) {
                    const float a = pa[1056 * t + 2 * q];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // D layout: lane holds column j = lane&31, rows i = 8*(r/4) + 4*(lane>>5) + (r&3).  The lane id is taken afresh (mbcnt):
    // keeping `li` / `kk` alive through the contraction cost the 96-VGPR kernel its one spilled register.
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int li_e = lane_e & 31, kk_e = lane_e >> 5;
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
        const int64_t nb = n0 + wave * WOUT + t * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 8 * (r >> 2) + 4 * kk_e + (r & 3);
            const int64_t n = nb + 32 * i + li_e;
            if ((DBG & 1) ? (acc[t][r] == 1234.5f) : (n < T)) yrow[n] = acc[t][r];
        }
    }
}

// generic (float64, or any dtype) LDS-tiled vector kernel: 256 threads x 4 outputs
template <typename T>
__global__ void __launch_bounds__(256)
fir_direct_simple_kernel(const T *__restrict__ x, T *__restrict__ y, const T *__restrict__ kf_dev,
                         int64_t C, int64_t Tn, int K, int64_t tiles_per_row, const T *__restrict__ hist, int H)
{
    constexpr int NO = 1024, KC = 512;
    __shared__ T xw[NO + KC];
    __shared__ T kp[KC];
    const int tid = threadIdx.x;
    const int64_t c = blockIdx.x / tiles_per_row;
    const int64_t n0 = (blockIdx.x % tiles_per_row) * (int64_t)NO;
    const T *xrow = x + c * Tn;
    T acc[4] = {0, 0, 0, 0};
    for (int t0 = 0; t0 < K; t0 += KC) {
        __syncthreads();
        const int64_t base = n0 + t0 - (int64_t)(K - 1);
        for (int m = tid; m < NO + KC; m += 256) {
            const int64_t g = base + m;
            xw[m] = (g >= 0 && g < Tn) ? xrow[g] : ((hist && g < 0 && g >= -(int64_t)H) ? hist[c * H + H + g] : (T)0);
        }
        for (int u = tid; u < KC; u += 256) kp[u] = (t0 + u < K) ? kf_dev[t0 + u] : (T)0;
        __syncthreads();
        const int kc = (K - t0 < KC) ? (K - t0) : KC;
        for (int u = 0; u < kc; ++u) {
            const T k = kp[u];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fma(k, xw[tid + 256 * r + u], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t n = n0 + tid + 256 * r;
        if (n < Tn) y[c * Tn + n] = acc[r];
    }
}

static int64_t envi_fir(const char *name, int64_t dflt) { return env_i64(name, dflt); }      // read once per process (common.h)

// The last H samples of the logical signal [hist | x] (what the next chunk needs as its history).
template <typename T>
__global__ void __launch_bounds__(256)
fir_hist_update_kernel(const T *__restrict__ x, const T *__restrict__ hist_in, T *__restrict__ hist_out,
                       int64_t C, int64_t Tn, int64_t H)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= C * H) return;
    const int64_t c = g / H, i = g - c * H;
    const int64_t j = Tn - H + i;                 // index into x; negative -> older history
    hist_out[g] = j >= 0 ? x[c * Tn + j] : (hist_in ? hist_in[c * H + H + j] : (T)0);
}

void fir_hist_update(const void *x, const void *hist_in, void *hist_out, int dtype, int64_t C, int64_t T, int64_t H,
                     hipStream_t stream)
{
    if (C * H == 0) return;
    TFX_CHECK(hist_out && hist_out != hist_in, "fir_stream_forward: the new history needs its own buffer");
    const unsigned grid = (unsigned)ceil_div(C * H, 256);
    if (dtype == TFX_F32)
        hipLaunchKernelGGL(fir_hist_update_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float *)x,
                           (const float *)hist_in, (float *)hist_out, C, T, H);
    else
        hipLaunchKernelGGL(fir_hist_update_kernel<double>, dim3(grid), dim3(256), 0, stream, (const double *)x,
                           (const double *)hist_in, (double *)hist_out, C, T, H);
    TFX_HIP(hipGetLastError());
}

void fir_direct_forward(const void *x, void *y, int dtype, int64_t C, int64_t T,
                        const void *kernel_host, int64_t K, hipStream_t stream, const void *hist, int64_t H)
{
    TFX_CHECK(dtype == TFX_F32 || dtype == TFX_F64, "fir_direct_forward: bad dtype %d", dtype);
    TFX_CHECK(K >= 1, "fir_direct_forward: empty kernel");
    TFX_CHECK(K < (1 << 30), "fir_direct_forward: kernel too long");
    TFX_CHECK(H >= 0 && H < (1 << 30) && (H == 0 || hist), "fir_direct_forward: bad history");
    if (C == 0 || T == 0) return;
    TFX_CHECK(x && y && kernel_host, "fir_direct_forward: null pointer");
    TFX_CHECK(C > 0 && T > 0, "fir_direct_forward: negative size");
    const size_t esz = dtype == TFX_F32 ? 4 : 8;
    const int64_t Kpad = ceil_div(K, FIR_KC_MAX) * FIR_KC_MAX;
    const void *kdev = cached_taps(kernel_host, (size_t)K * esz, (size_t)Kpad * esz);
    // rows much shorter than one 16384-sample MFMA tile (streaming chunks): the plain LDS-tiled kernel
    // has 1024-sample tiles and finishes in a few microseconds instead of a full tile's ~60
    // ... and so does any job whose 1024-sample tiles are all resident at once (8 workgroups per CU): one round of
    // the plain kernel costs ~40 clocks per tap, one MFMA workgroup walks its 16384-sample tile for ~128 clocks per
    // tap -- the MFMA kernel wins on throughput (2.5 x), not on latency (streaming chunks, 64 x 4096 and the like)
    const bool few_tiles = C * ceil_div(T, (int64_t)1024) <= envi_fir("TFX_FIR_ONE_ROUND_TILES", 2048);   // 0: MFMA whenever T allows (tests)
    const bool short_rows = dtype == TFX_F32 && (T < envi_fir("TFX_FIR_MFMA_MIN_T", FIR_NOUT / 4) || few_tiles);
    if (dtype == TFX_F32 && !short_rows) {
        const int njv = (int)envi_fir("TFX_FIR_NJ", 1);
        const int nj = (njv == 2 || njv == 4) ? njv : 1;
        const int nout = 4 * nj * 1024;
        const int64_t tiles = ceil_div(T, nout);
        TFX_CHECK(C * tiles < (1ll << 31), "fir_direct_forward: grid too large");
        // chunk size: cost per chunk ~ (KC+32)/32 contraction blocks + ~4 blocks' worth of refill
        int kc = 1024;
        int64_t best = -1;
        for (int cand : {128, 512, 1024}) {
            const int64_t cost = ceil_div(K, cand) * ((cand + 32) / 32 + 4);
            if (best < 0 || cost < best) best = cost, kc = cand;
        }
        {
            const int v = (int)envi_fir("TFX_FIR_KC", 0);
            if (v == 128 || v == 512 || v == 1024) kc = v;
        }
        const int nchunks = (int)ceil_div(K, kc);
        auto launch = [&](auto kern, int KC) {
            const int XW = nout + KC + 32;
            const int XW_PAD = XW + (XW >> 5) + 1;
            const size_t shmem = (((XW_PAD + 3) & ~3) + 31 + KC + 33) * sizeof(float);
            static bool attr_done[TFX_MAX_DEVICES][9] = {};
            bool &done = attr_done[current_device()][(KC == 128 ? 0 : (KC == 512 ? 1 : 2)) + (nj == 2 ? 3 : (nj == 1 ? 6 : 0))];
            if (!done) {
                TFX_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
                done = true;
            }
            ProfScope ps("fir_direct_mfma_kernel", stream);
            hipLaunchKernelGGL(kern, dim3((unsigned)(ceil_div(C * tiles, 8) * 8)), dim3(256), shmem, stream, (const float *)x,
                               (float *)y, (const float *)kdev, C, T, (int)K, nchunks, tiles, (const float *)hist, (int)H);
            TFX_HIP(hipGetLastError());
        };
        if (nj == 1) {
            if (kc == 128) launch(fir_direct_mfma_kernel<128, 0, 1>, 128);
            else if (kc == 512) launch(fir_direct_mfma_kernel<512, 0, 1>, 512);
            else launch(fir_direct_mfma_kernel<1024, 0, 1>, 1024);
        } else if (nj == 2) {
            if (kc == 128) launch(fir_direct_mfma_kernel<128, 0, 2>, 128);
            else if (kc == 512) launch(fir_direct_mfma_kernel<512, 0, 2>, 512);
            else launch(fir_direct_mfma_kernel<1024, 0, 2>, 1024);
        } else if (kc == 128) launch(fir_direct_mfma_kernel<128, 0>, 128);
        else if (kc == 512) launch(fir_direct_mfma_kernel<512, 0>, 512);
        else launch(fir_direct_mfma_kernel<1024, 0>, 1024);
    } else {
        const int64_t tiles = ceil_div(T, 1024);
        TFX_CHECK(C * tiles < (1ll << 31), "fir_direct_forward: grid too large");
        if (dtype == TFX_F32) {
            ProfScope ps("fir_direct_simple_kernel<f32>", stream);
            hipLaunchKernelGGL(fir_direct_simple_kernel<float>, dim3((unsigned)(C * tiles)), dim3(256), 0, stream,
                               (const float *)x, (float *)y, (const float *)kdev, C, T, (int)K, tiles, (const float *)hist, (int)H);
        } else {
            ProfScope ps("fir_direct_simple_kernel<f64>", stream);
            hipLaunchKernelGGL(fir_direct_simple_kernel<double>, dim3((unsigned)(C * tiles)), dim3(256), 0, stream,
                               (const double *)x, (double *)y, (const double *)kdev, C, T, (int)K, tiles, (const double *)hist, (int)H);
        }
        TFX_HIP(hipGetLastError());
    }
}

}  // namespace tfx
