// sos.hip -- fused K-section DF1 SOS cascade for gfx950 (MI355X).
//
// Replaces the reference's CPU loop (src/torchfx/_csrc/cpu/iir_cpu.cpp:64-159) and its
// CUDA path (cuda/biquad_forward.cu:49-92 + cuda/parallel_scan.cu: one forcing kernel and a
// 3-phase Blelloch scan PER SECTION, ~40 B/sample/section of HBM traffic) with ONE launch that
// reads x once and writes y once (8 B/sample for f32 I/O).  Not a port of either.
//
// Decomposition (DESIGN.md "IIR kernel"):
//   * a STREAM = (channel, time segment) is owned by one 64-lane wavefront; a workgroup is four
//     independent streams (no __syncthreads anywhere).  Segments other than the first start
//     `warm` samples early from zero state (the halo is part of the stream's tile grid: every
//     stream walks the same number of full tiles); `warm` is chosen on the host so that the cascade's
//     zero-input transition matrix A^warm is below 2^-60, i.e. the halo reproduces the true
//     state to float64 round-off.  Filters whose memory is too long get nseg = 1 (exact,
//     sequential tiles per channel).
//   * a TILE = 64 lanes x LC consecutive samples.  Coalesced 16-byte global loads are
//     transposed through a padded (bank-conflict-free) LDS stage so that lane j owns samples
//     [j*LC, (j+1)*LC) in registers.  The next tile's loads are issued before computing.
//   * per section, entirely in registers (7 flop per sample):  (1a) the feed-forward part
//     f[n] = b0 v[n] + b1 v[n-1] + b2 v[n-2] overwrites v in place, walking n downwards;  (1b) the
//     recursion runs over f from zero state (lane 0: from the carried true state) only to get the
//     lane's 2-vector end state;  (2) the end states are combined across lanes by a DPP-only scan
//     (row_shr Kogge-Stone inside 16-lane rows, row_bcast:15 / row_bcast:31 across rows, wave_shr:1)
//     with constant 2x2 matrices C^(LC*m) computed on the host in long double;  (3) the recursion
//     runs once more over f from the true start state and leaves the section output in place.
//     Section s+1 then consumes the exact output of s.
//   * arithmetic type TC is float64 by default -- the reference computes in float64
//     (_ops.py:149) -- or float32 (TFX_PREC_F32).
#include "common.h"
#include "epilogue.h"
#include "../../include/torchfx_hip.h"

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace tfx {

// ------------------------------------------------------------------------------------------
// Table layout per section (TC elements):
//   [0..4]  b0 b1 b2 -a1 -a2      [5..7] pad
//   [8 + 4k + {0..3}]             Cmp^(LC * 2^k) row-major, k < 6   (Cmp = [[-a1,-a2],[1,0]])
//   [32 + 4p + {0..3}]            Cmp^(LC * (p+1)), p < 32: per-lane matrices of the scan's two
//                                 cross-row steps (p = lane % 16 and p = lane % 32)
//   [160..163]                    1/Gin, 1/Gout, Gin, Gout: the scale of the section's input / output in the unit-b0 form (else 1)
// Unit-b0 form (UNIT kernels, float64, K >= 2): every b0 is pulled out of its section, section s computes
//   y' = v' + (b1/b0) v'[n-1] + (b2/b0) v'[n-2] - a1 y'[n-1] - a2 y'[n-2]
// on v' = v / Gin, y' = y / Gout (Gin = product of the b0 before it, Gout = Gin b0): step (1a) loses its multiply, the product
// of all b0 comes back in ONE multiply where the output is rounded and stored -- K - 1 vector instructions per sample less.
// The carried states enter and leave through the same factors.  Same recursion, same poles; the numerator arithmetic
// differs from the reference's by float64 round-off (parity at 2e-11 of the output scale like the plain form).
// Cascades with a zero / tiny / huge b0 keep the plain form (unit_form_ok).
// ------------------------------------------------------------------------------------------
__host__ __device__ constexpr int tab_stride(int) { return 32 + 128 + 8; }

struct SosParams {
    const void *x;
    void *y;
    void *taps;          // optional [K,C,T] of TOut
    const void *tab;     // device table, TC
    const double *sx_in, *sy_in;
    double *sx_out, *sy_out;
    int64_t C, T;        // C = output rows (= bands x input rows in filter-bank mode)
    int64_t x_pitch;     // elements between consecutive input rows (= T for a contiguous [C_in, T] signal)
    int64_t C_in;        // input rows: output row c reads input row c % C_in with the tables of band c / C_in
    int64_t seg_len;     // distance between the starts of consecutive streams of a row = seg_tiles * TILE - warm
    int64_t warm;        // halo of the streams g > 0; multiple of 32 samples (streams start on 128-byte lines)
    int K, nseg, nsteps;
    int nt;              // nontemporal global loads / stores of the signal (aligned 16-byte path)
    int nsum;            // > 0: sum mode -- every stream runs `nsum` bands over its input row and accumulates them
    // epilogue on the stored samples (epilogue.h): y *= gain, clip, partial of max|y| / sum y^2 per stream
    double ep_gain;
    int ep_scale, ep_clamp, ep_stat;
    double *ep_partial;  // [C * nseg], one per stream (row-major over (row, segment)), pre-zeroed
    int *nf_flag;        // [C * nseg]: stream ended with a non-finite carried state (nseg > 1 only; every stream writes its slot)
    const void *ep_host; // host side only: the Epilogue this launch serves
    int ep_fused;        // host side only: the kernel applies it (else separate passes follow the launch)
    int fair;            // > 0: waves that share a SIMD alternate their issue priority every 2^fair clocks
    int fair_nw;         // waves per SIMD of this launch (the priority levels that rotate): 2 ... 4
    int unit;            // host side only: `tab` holds the unit-b0 form (UNIT kernels)
};

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP shift of a value across lanes; lanes with no source lane receive 0 (bound_ctrl).
template <int CTRL> __device__ __forceinline__ float dpp_shift(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_shift(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// DPP row broadcast (row_bcast:15 = 0x142, row_bcast:31 = 0x143) into the rows of ROWMASK; all
// other lanes receive 0.
template <int CTRL, int ROWMASK> __device__ __forceinline__ float dpp_bcast(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false));
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dpp_bcast(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// streaming (nontemporal) 16-byte global accesses: the signal is read once and written once, and a plain streaming copy
// gains 3-8 % from them on this part (tools/ubench/stream_copy2.hip: linear sweep 6.29 -> 6.80 TB/s, persistent streams
// 5.06 -> 5.20 / 5.86 -> 6.12)
typedef unsigned uintx4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16(const uint4 *p, bool nt)
{
    if (nt) { const uintx4_t v = __builtin_nontemporal_load((const uintx4_t *)p); return make_uint4(v.x, v.y, v.z, v.w); }
    return *p;
}
__device__ __forceinline__ void st16(uint4 *p, uint4 v, bool nt)
{
    if (nt) __builtin_nontemporal_store((uintx4_t){v.x, v.y, v.z, v.w}, (uintx4_t *)p);
    else *p = v;
}
template <typename T> struct U16 {               // 16 bytes of T
    static constexpr int N = 16 / sizeof(T);
    union { uint4 u; T e[N]; };
};

// LC   samples per lane per tile          VEC  16-byte global accesses (aligned rows)
// TAPS also store every section's output  PF   keep the next tile's loads in flight in registers
// MINW waves per SIMD the register allocator must leave room for
// SUMB sum mode (`+` of IIR branches, __base.py:1019-1026): the stream applies p.nsum independent
//      cascades ("bands") to its input tile, which stays in the LDS stage, and accumulates their
//      outputs -- rounded to TOut per band and added in branch order, exactly like the reference's
//      zeros_like + in-place adds -- so N branches cost 8 B/sample instead of N x 8 + (N + 1) x 4
// EPI  epilogue on the stored samples (epilogue.h); a separate instantiation so that the plain kernel keeps
//      its register budget (the statistic accumulator and the extra selects cost ~30 VGPRs = one wave per SIMD)
// The stream body: one wavefront walks stream `sid` = (row, segment) with `stage` as its private LDS (transposition
// stage + carry).  Shared by the cascade kernel below and by the fused per-chunk kernel (chunk_iir_fir_kernel), whose
// output pointer is an LDS buffer.
template <typename TIn, typename TOut, typename TC, int LC, bool VEC, bool TAPS, bool PF, bool SUMB, bool EPI, bool UNIT = false>
__device__ __forceinline__ void sos_stream_body(const SosParams &p, const int64_t sid, char *const stage, const int lane)
{
    static_assert(!(TAPS && SUMB), "section taps are not available in sum mode");
    static_assert(!(UNIT && (TAPS || SUMB)), "the unit-b0 form serves the plain cascade only");
    constexpr int IOB = sizeof(TIn) > sizeof(TOut) ? sizeof(TIn) : sizeof(TOut);
    constexpr int CHUNK_B = LC * IOB + 16;   // per-lane chunk, padded: conflict-free b128 access
    constexpr int STAGE_B = 64 * CHUNK_B;
    constexpr int TILE = 64 * LC;
    constexpr int TS = tab_stride(LC);
#ifndef TFX_SB
#define TFX_SB 4
#endif
    constexpr int SB = TFX_SB;          // scheduling-barrier period (samples)
    constexpr int EI = 16 / sizeof(TIn), NUI = LC / EI;   // elems per 16 B, units per lane (in)
    constexpr int EO = 16 / sizeof(TOut), NUO = LC / EO;  // (out)

    const int K = p.K;
    if (sid >= p.C * p.nseg) return;                       // wave-uniform
    unsigned wave_slot = 0;
    if (p.fair) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        wave_slot = hw & 15u;                              // wave slot within the SIMD
    }
    const int64_t c = sid / p.nseg;
    const int g = (int)(sid - c * p.nseg);
    const int nbl = SUMB ? p.nsum : 1;                     // bands handled inside this stream
    TC *carry = (TC *)(stage + STAGE_B);                   // [nbl][K][4] = vin1 vin2 y1 y2
    TC *cap = carry + nbl * K * 4;                         // [2][LC] final-state capture scratch

    const int64_t T = p.T;
    const int64_t band = SUMB ? 0 : c / p.C_in;
    const int64_t st_rows = SUMB ? p.C_in * nbl : p.C;     // rows of the [K, rows, 2] state tensors
    const TIn *__restrict__ xrow = (const TIn *)p.x + (c - band * p.C_in) * p.x_pitch;
    TOut *__restrict__ yrow = (TOut *)p.y + c * T;
    // Coefficient tables live in the CONSTANT address space: wave-uniform indices then lower to
    // s_load (scalar cache, SGPR operands) instead of per-lane vector loads.
    typedef const TC __attribute__((address_space(4))) *ctab_t;
    const ctab_t tab = (ctab_t)(uintptr_t)p.tab + band * (int64_t)K * TS;

    // stream g reads [g * seg_len, (g + 1) * seg_len + warm): exactly seg_tiles full tiles for every stream, the
    // first `warm` samples of a stream g > 0 being its halo (plan_segments)
    const int64_t start = (int64_t)g * p.seg_len;
    const int64_t out_begin = g ? start + p.warm : 0;
    if (out_begin >= T) {
        if (p.nf_flag && lane == 0) p.nf_flag[sid] = 0;
        return;
    }
    int64_t out_end = start + p.seg_len + p.warm;
    if (out_end > T || g == p.nseg - 1) out_end = T;
    const bool last_seg = (out_end == T);

    // ---- initial carry: the caller's state for the stream that starts at n = 0, zeros for a
    //      warm-up start.  Layout of state tensors: [K, C, 2] (iir_cpu.cpp:125-130).
    for (int i = lane; i < nbl * K * 4; i += 64) {
        const int b = i / (K * 4), rem = i - b * K * 4;
        const int s = rem >> 2, f = rem & 3;
        TC v = (TC)0;
        if (start == 0) {
            const double *src = (f < 2) ? p.sx_in : p.sy_in;
            if (src) v = (TC)src[((int64_t)s * st_rows + (SUMB ? b * p.C_in + c : c)) * 2 + (f & 1)];
            if (UNIT) v *= ((const TC *)p.tab)[(band * K + s) * TS + (f < 2 ? 160 : 161)];      // into the section's scale
        }
        carry[i] = v;
    }
    wave_sync();

    // ---- global -> registers (coalesced).  Per-lane pointer + immediate offsets; the common
    //      full-tile case is branch-free, a partial tile (signal tail) is predicated.
    uint4 rawv[VEC ? NUI : 1];
    TIn raws[VEC ? 1 : LC];
    auto load_tile = [&](int64_t ts) {
        const int64_t left = T - ts;
        if constexpr (VEC) {
            const uint4 *__restrict__ xl = (const uint4 *)(xrow + ts) + lane;
            if (left >= TILE) {
#pragma unroll
                for (int i = 0; i < NUI; ++i) rawv[i] = ld16(xl + i * 64, p.nt);
            } else {
                const int nv = (int)(left / EI);   // valid 16-byte units
#pragma unroll
                for (int i = 0; i < NUI; ++i)
                    rawv[i] = (i * 64 + lane < nv) ? xl[i * 64] : make_uint4(0, 0, 0, 0);
            }
        } else {
            const TIn *__restrict__ xl = xrow + ts + lane;
            if (left >= TILE) {
#pragma unroll
                for (int i = 0; i < LC; ++i) raws[i] = xl[i * 64];
            } else {
                const int nv = (int)left;
#pragma unroll
                for (int i = 0; i < LC; ++i) raws[i] = (i * 64 + lane < nv) ? xl[i * 64] : (TIn)0;
            }
        }
    };
    // LDS stage addresses: unit q = i*64 + lane lives at chunk q/NU, slot q%NU; 64 % NU == 0
    char *const st_in_v = stage + (lane / NUI) * CHUNK_B + (lane % NUI) * 16;
    char *const st_out_v = stage + (lane / NUO) * CHUNK_B + (lane % NUO) * 16;
    char *const st_in_s = stage + (lane / LC) * CHUNK_B + (lane % LC) * (int)sizeof(TIn);     // LC | 64
    char *const st_out_s = stage + (lane / LC) * CHUNK_B + (lane % LC) * (int)sizeof(TOut);
    char *const st_own = stage + lane * CHUNK_B;

    if constexpr (PF) load_tile(start);
    double ep_acc = 0.0;                   // this lane's share of the stream's statistic (0 is neutral for both modes)

    for (int64_t ts = start; ts < out_end; ts += TILE) {
        TC d[LC];
        if constexpr (!PF) load_tile(ts);
        // ---- stage: coalesced order -> LDS -> blocked (lane j owns [j*LC,(j+1)*LC))
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < NUI; ++i) *(uint4 *)(st_in_v + i * (64 / NUI) * CHUNK_B) = rawv[i];
        } else {
#pragma unroll
            for (int i = 0; i < LC; ++i) *(TIn *)(st_in_s + i * (64 / LC) * CHUNK_B) = raws[i];
        }
        wave_sync();
        auto read_own = [&]() {
#pragma unroll
            for (int i = 0; i < NUI; ++i) {
                U16<TIn> v;
                v.u = *(const uint4 *)(st_own + i * 16);
#pragma unroll
                for (int e = 0; e < EI; ++e) d[i * EI + e] = (TC)v.e[e];
            }
        };
        if constexpr (!SUMB) {
            read_own();
            wave_sync();
        }

        // ---- prefetch the next tile while this one is computed
        if constexpr (PF) { if (ts + TILE < out_end) load_tile(ts + TILE); }

        const bool final_tile = last_seg && (ts + TILE >= T);
        const int r = (int)(T - ts);   // valid samples in the final tile (1..TILE)

        TOut acc[SUMB ? LC : 1];
        for (int bnd = 0; bnd < nbl; ++bnd) {
        if constexpr (SUMB) read_own();          // the input tile is still in this lane's LDS chunk
        const int64_t tband = SUMB ? bnd : band;
        TC *const carry_s = carry + bnd * K * 4;
        for (int s = 0; s < K; ++s) {
            if (p.fair) {
                // Fair share of the SIMD.  The waves of two workgroups share each SIMD for the whole launch, and the issue arbiter
                // serves them by priority, then AGE: the older wave runs almost unimpeded, the younger one on the leftover slots
                // (per-stream time stamps, round 4: the first wave of every SIMD finished after 251 us, the second after 347 us,
                // the last 96 us with one wave per SIMD and half the issue rate).  Both waves read the same clock, and each takes
                // the high priority in alternate epochs according to its slot, so they advance together and finish together.
                // (more than two waves per SIMD: the priority levels 0 ... nw - 1 rotate over the slots, s_setprio has four)
                const unsigned e = (unsigned)(__builtin_readcyclecounter() >> p.fair);
                const unsigned lvl = (e + wave_slot) % (unsigned)p.fair_nw;
                if (lvl == 0) __builtin_amdgcn_s_setprio(0);
                else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
                else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(3);
            }
            const ctab_t tb = tab + (SUMB ? (int64_t)bnd * K * TS : 0) + s * TS;
            const TC b0 = tb[0], b1 = tb[1], b2 = tb[2], na1 = tb[3], na2 = tb[4];
            TC mqa[4], mqb[4];                  // P^(lane%16 + 1), P^(lane%32 + 1)
            {
                const TC *mp = (const TC *)p.tab + (tband * K + s) * TS + 32;
#pragma unroll
                for (int i = 0; i < 4; ++i) mqa[i] = mp[4 * (lane & 15) + i];
#pragma unroll
                for (int i = 0; i < 4; ++i) mqb[i] = mp[4 * (lane & 31) + i];
            }
            const TC cv1 = carry_s[s * 4 + 0], cv2 = carry_s[s * 4 + 1];
            const TC cy1 = carry_s[s * 4 + 2], cy2 = carry_s[s * 4 + 3];

            // final tile only: read the section's sequence at tile-local index idx (-1/-2 = the
            // carried history) through a 2 x LC scratch in LDS
            auto capture2 = [&](int i1, int i2, TC h1, TC h2, TC &o1, TC &o2) {
                const int w1 = i1 >= 0 ? i1 / LC : -1, w2 = i2 >= 0 ? i2 / LC : -1;
                if (lane == w1) {
#pragma unroll
                    for (int n = 0; n < LC; ++n) cap[n] = d[n];
                }
                if (lane == w2) {
#pragma unroll
                    for (int n = 0; n < LC; ++n) cap[LC + n] = d[n];
                }
                wave_sync();
                o1 = i1 >= 0 ? cap[i1 - w1 * LC] : (i1 == -1 ? h1 : h2);
                o2 = i2 >= 0 ? cap[LC + i2 - w2 * LC] : (i2 == -1 ? h1 : h2);
                wave_sync();
            };

            // input history of this lane's chunk: previous lane's last two inputs
            TC pv1 = dpp_shift<0x138>(d[LC - 1]);     // wave_shr:1
            TC pv2 = dpp_shift<0x138>(d[LC - 2]);
            if (lane == 0) { pv1 = cv1; pv2 = cv2; }
            TC sx1 = (TC)0, sx2 = (TC)0;
            if (final_tile) capture2(r - 1, r - 2, cv1, cv2, sx1, sx2);
            if (lane == 63) { carry_s[s * 4 + 0] = d[LC - 1]; carry_s[s * 4 + 1] = d[LC - 2]; }

            // (1a) feed-forward part f[n] = b0 v[n] + b1 v[n-1] + b2 v[n-2], written IN PLACE by
            //      walking n downwards (f[n] never needs v[m] for m > n): 3 flop/sample, no second
            //      register set and no copies
#pragma unroll
            for (int n = LC - 1; n >= 0; --n) {
                const TC x1 = n >= 1 ? d[n - 1] : pv1;
                const TC x2 = n >= 2 ? d[n - 2] : (n == 1 ? pv1 : pv2);
                d[n] = UNIT ? fma(b2, x2, fma(b1, x1, d[n])) : fma(b2, x2, fma(b1, x1, b0 * d[n]));
            }
            // (1b) the recursion over f from zero start state (lane 0: from the carried true
            //      state), only to get this chunk's END STATE -- 2 flop/sample, nothing stored
            TC u1 = (lane == 0) ? cy1 : (TC)0;
            TC u2 = (lane == 0) ? cy2 : (TC)0;
#pragma unroll
            for (int n = 0; n < LC; ++n) {
                const TC u = fma(na1, u1, fma(na2, u2, d[n]));
                u2 = u1;   u1 = u;
            }

            // (2) inclusive scan of chunk end-states over lanes:  S_j = sum_i P^(j-i) z_i, then the
            //     exclusive shift (h1,h2) = S_{j-1}.  All cross-lane traffic is DPP -- a few cycles of
            //     latency per step instead of an LDS round trip per ds_bpermute:
            //       a. intra-row Kogge-Stone with P^1, P^2, P^4, P^8 (row_shr; lanes without a
            //          source add 0)
            //       b. row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3, each applied with
            //          the lane's own matrix P^(lane%16+1) / P^(lane%32+1) from the table
            //       c. wave_shr:1 turns the inclusive scan into every lane's start state
            const ctab_t pm = tb + 8;
            TC s0 = u1, s1 = u2;
#define TFX_KS_STEP(K_, CTRL_)                                                            \
            {                                                                             \
                const TC t0 = dpp_shift<CTRL_>(s0), t1 = dpp_shift<CTRL_>(s1);            \
                s0 += fma(pm[4 * K_ + 0], t0, pm[4 * K_ + 1] * t1);                       \
                s1 += fma(pm[4 * K_ + 2], t0, pm[4 * K_ + 3] * t1);                       \
            }
            TFX_KS_STEP(0, 0x111)   // row_shr:1
            TFX_KS_STEP(1, 0x112)   // row_shr:2
            TFX_KS_STEP(2, 0x114)   // row_shr:4
            TFX_KS_STEP(3, 0x118)   // row_shr:8
#undef TFX_KS_STEP
            {
                // b. rows 1 and 3 take in the aggregate of the row before them (row_bcast:15),
                //    then rows 2 and 3 the inclusive value of lane 31 (row_bcast:31): two DPP
                //    steps with per-lane matrices instead of readlanes + a select tree
                const TC a0 = dpp_bcast<0x142, 0xA>(s0), a1 = dpp_bcast<0x142, 0xA>(s1);
                s0 += fma(mqa[0], a0, mqa[1] * a1);
                s1 += fma(mqa[2], a0, mqa[3] * a1);
                const TC c0 = dpp_bcast<0x143, 0xC>(s0), c1 = dpp_bcast<0x143, 0xC>(s1);
                s0 += fma(mqb[0], c0, mqb[1] * c1);
                s1 += fma(mqb[2], c0, mqb[3] * c1);
            }
            TC h1 = dpp_shift<0x138>(s0), h2 = dpp_shift<0x138>(s1);     // wave_shr:1: state at chunk start
            if (lane == 0) { h1 = cy1; h2 = cy2; }                       // lane 0: the carried true state

            // (3) the recursion proper from the TRUE start state over the stored f[n]: 2 flop/sample,
            //     writes the section output in place.  (Cheaper than correcting the zero-state output
            //     with its homogeneous response, 3 flop/sample, and no table.)
#pragma unroll
            for (int n = 0; n < LC; ++n) {
                const TC t = fma(na2, h2, d[n]);
                const TC yv = fma(na1, h1, t);
                d[n] = yv;
                h2 = h1; h1 = yv;
                if ((n & (SB - 1)) == SB - 1) __builtin_amdgcn_sched_barrier(0);
            }

            if (lane == 63) { carry_s[s * 4 + 2] = d[LC - 1]; carry_s[s * 4 + 3] = d[LC - 2]; }

            if (final_tile) {
                TC sy1, sy2;
                capture2(r - 1, r - 2, cy1, cy2, sy1, sy2);
                if (lane == 0) {
                    const int64_t o = ((int64_t)s * st_rows + (SUMB ? bnd * p.C_in + c : c)) * 2;
                    const TC g_in = UNIT ? tb[162] : (TC)1, g_out = UNIT ? tb[163] : (TC)1;      // back to the true scale
                    if (p.sx_out) { p.sx_out[o] = (double)(sx1 * g_in); p.sx_out[o + 1] = (double)(sx2 * g_in); }
                    if (p.sy_out) { p.sy_out[o] = (double)(sy1 * g_out); p.sy_out[o + 1] = (double)(sy2 * g_out); }
                }
            }
            if constexpr (TAPS) {   // debug / section-by-section parity only
                TOut *trow = (TOut *)p.taps + ((int64_t)s * p.C + c) * T + ts;
                const int64_t lo64 = out_begin - ts, hi64 = out_end - ts;
                const int lo = lo64 > 0 ? (int)lo64 : 0, hi = hi64 < TILE ? (int)hi64 : TILE;
#pragma unroll
                for (int n = 0; n < LC; ++n) {
                    const int rel = lane * LC + n;
                    if (rel >= lo && rel < hi) trow[rel] = (TOut)d[n];
                }
            }
            wave_sync();   // carry[] written by lane 63 is read by all lanes next tile/section
        }
        if constexpr (SUMB) {
#pragma unroll
            for (int n = 0; n < LC; ++n) acc[n] = (bnd == 0 ? (TOut)0 : acc[n]) + (TOut)d[n];
        }
        }   // bands
        if constexpr (SUMB) {
#pragma unroll
            for (int n = 0; n < LC; ++n) d[n] = (TC)acc[n];
        }


        // ---- store (skipped entirely while still inside the warm-up halo)
        if (ts + TILE > out_begin) {
            const int64_t lo64 = out_begin - ts, hi64 = out_end - ts;
            const bool full = (lo64 <= 0) && (hi64 >= TILE);
            if constexpr (UNIT) {                       // the product of all b0, once, on the way out
                const TC gt = tab[(K - 1) * TS + 163];
#pragma unroll
                for (int n = 0; n < LC; ++n) d[n] *= gt;
            }
            if constexpr (!EPI) {
#pragma unroll
                for (int i = 0; i < NUO; ++i) {
                    U16<TOut> v;
#pragma unroll
                    for (int e = 0; e < EO; ++e) v.e[e] = (TOut)d[i * EO + e];
                    *(uint4 *)(st_own + i * 16) = v.u;
                }
            } else {
                // Gain / clamp on the rounded output value, in the output dtype (what a standalone Gain pass
                // would compute from the stored sample), and this lane's share of the statistic over the
                // samples that are really stored
                const TOut g_ = (TOut)p.ep_gain;
                const int lo_s = lo64 > 0 ? (int)lo64 : 0, hi_s = hi64 < TILE ? (int)hi64 : TILE;
#pragma unroll
                for (int i = 0; i < NUO; ++i) {
                    U16<TOut> v;
#pragma unroll
                    for (int e = 0; e < EO; ++e) {
                        TOut o = (TOut)d[i * EO + e];
                        if (p.ep_scale) o = o * g_;
                        if (p.ep_clamp) o = clamp_unit(o);
                        v.e[e] = o;
                        if (p.ep_stat >= 0) {
                            const int rel = lane * LC + i * EO + e;
                            if (full || (rel >= lo_s && rel < hi_s)) ep_acc = red_comb_rt(p.ep_stat, ep_acc, red_elem_rt(p.ep_stat, (double)o));
                        }
                    }
                    *(uint4 *)(st_own + i * 16) = v.u;
                }
            }
            wave_sync();
            if constexpr (VEC) {
                uint4 *__restrict__ yl = (uint4 *)(yrow + ts) + lane;
                if (full) {
#pragma unroll
                    for (int i = 0; i < NUO; ++i)
                        st16(yl + i * 64, *(const uint4 *)(st_out_v + i * (64 / NUO) * CHUNK_B), p.nt);
                } else {
                    const int lo = lo64 > 0 ? (int)(lo64 / EO) : 0;
                    const int hi = hi64 < TILE ? (int)(hi64 / EO) : TILE / EO;
#pragma unroll
                    for (int i = 0; i < NUO; ++i) {
                        const int q = i * 64 + lane;
                        const uint4 v = *(const uint4 *)(st_out_v + i * (64 / NUO) * CHUNK_B);
                        if (q >= lo && q < hi) yl[i * 64] = v;
                    }
                }
            } else {
                TOut *__restrict__ yl = yrow + ts + lane;
                const int lo = lo64 > 0 ? (int)lo64 : 0;
                const int hi = hi64 < TILE ? (int)hi64 : TILE;
#pragma unroll
                for (int i = 0; i < LC; ++i) {
                    const int e = i * 64 + lane;
                    const TOut v = *(const TOut *)(st_out_s + i * (64 / LC) * CHUNK_B);
                    if (e >= lo && e < hi) yl[i * 64] = v;
                }
            }
            wave_sync();
        }
    }
    if (p.nf_flag) {
        // the carried state at the end of the stream (every band, every section: last two inputs and outputs);
        // non-finite there = non-finite from some sample of this stream to the end of the row (iir_cpu.cpp:132-147),
        // whatever an epilogue (clamp) made of the stored samples
        int badc = 0;
        for (int i = lane; i < nbl * K * 4; i += 64) badc |= !(fabs((double)carry[i]) <= 1.79e308);
        const int anyb = __any(badc) ? 1 : 0;
        if (lane == 0) p.nf_flag[sid] = anyb;
    }
    if (EPI && p.ep_stat >= 0) {           // one partial per stream, lanes combined in a fixed order
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ep_acc = red_comb_rt(p.ep_stat, ep_acc, __shfl_xor(ep_acc, off));
        if (lane == 0) p.ep_partial[sid] = ep_acc;
    }
}

// bytes of LDS one stream needs: the transposition stage plus the carry (4 values per band and section, 2 x LC capture scratch)
template <typename TIn, typename TOut, typename TC, int LC>
__host__ __device__ constexpr int sos_stage_bytes()
{
    return 64 * (LC * (int)(sizeof(TIn) > sizeof(TOut) ? sizeof(TIn) : sizeof(TOut)) + 16);
}
template <typename TC, int LC> __host__ __device__ inline int sos_carry_bytes(int nbl, int K)
{
    return (((nbl * K * 4 + 2 * LC) * (int)sizeof(TC)) + 15) & ~15;
}

template <typename TIn, typename TOut, typename TC, int LC, bool VEC, bool TAPS, bool PF, int MINW, bool SUMB = false, bool EPI = false, bool UNIT = false>
__global__ void __launch_bounds__(256, MINW) sos_stream_kernel(const SosParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform -> SGPR addressing
    const int per_wave = sos_stage_bytes<TIn, TOut, TC, LC>() + sos_carry_bytes<TC, LC>(SUMB ? p.nsum : 1, p.K);
    sos_stream_body<TIn, TOut, TC, LC, VEC, TAPS, PF, SUMB, EPI, UNIT>(p, (int64_t)blockIdx.x * 4 + wave, smem + wave * per_wave, lane);
}

// Non-finite samples and time segmentation.  In the sequential recursion a NaN / Inf never leaves: once the
// state is non-finite every later output of the row is (iir_cpu.cpp:132-147).  A segment that starts from its
// warm-up halo does not see what happened before the halo, so every stream leaves a flag "my carried state ended
// non-finite" (every slot is written: no memset in front of the launch) and a second, tiny launch -- one workgroup
// per row; the kernel boundary is what makes the flags and samples of the eight XCDs' L2s visible -- finds the first
// flagged segment g0 of its row and overwrites the segments after it with NaN: their samples (and section taps),
// their statistic partials (so a following Normalize sees NaN like the standalone reduction would) and the returned
// states: "non-finite from the first bad sample to the end of the row", independent of how many segments the launch
// used and of what an epilogue (clamp) did to the stored samples.  Finite signals: C workgroups read nseg flags each
// and exit.  (Folding this into the main launch -- last workgroup to finish, device-scope fences -- was measured:
// the per-workgroup release fence writes back the XCD's L2 and the cfg-2 kernel went from 0.29-0.33 to 0.46-0.52 ms.)
//   y rows = C (output rows); state rows = st_rows with row c owning {b * C_in + c} for b < nbl in sum mode.
template <typename TOut>
__global__ void __launch_bounds__(256) sos_nonfinite_fix_kernel(const SosParams p, int nbl, int64_t st_rows)
{
    __shared__ int sh_first;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int64_t T = p.T, row = blockIdx.x;
    if (tid == 0) sh_first = p.nseg;
    __syncthreads();
    for (int q = tid; q < p.nseg; q += nthr)
        if (p.nf_flag[row * p.nseg + q]) atomicMin(&sh_first, q);
    __syncthreads();
    const int g0 = sh_first;
    if (g0 >= p.nseg - 1) return;                                // clean row, or only the last segment is bad (it poisons itself)
    const int64_t begin = (int64_t)(g0 + 1) * p.seg_len + p.warm;   // first sample segment g0 + 1 stores
    if (begin >= T) return;
    TOut *y = (TOut *)p.y, *taps = (TOut *)p.taps;
    const TOut nanv = (TOut)__builtin_nan("");
    for (int64_t n = begin + tid; n < T; n += nthr) y[row * T + n] = nanv;
    if (taps)
        for (int sct = 0; sct < p.K; ++sct)
            for (int64_t n = begin + tid; n < T; n += nthr) taps[((int64_t)sct * p.C + row) * T + n] = nanv;
    if (p.ep_stat >= 0 && p.ep_partial)
        for (int g = g0 + 1 + tid; g < p.nseg; g += nthr) p.ep_partial[row * p.nseg + g] = __builtin_nan("");
    for (int i = tid; i < nbl * p.K * 2; i += nthr) {
        const int b = i / (p.K * 2), sct = (i >> 1) % p.K, f = i & 1;
        const int64_t o = ((int64_t)sct * st_rows + (nbl > 1 ? b * p.C_in + row : row)) * 2 + f;
        if (p.sx_out && sct > 0) p.sx_out[o] = __builtin_nan("");   // section 0's input history is the signal itself: already exact
        if (p.sy_out) p.sy_out[o] = __builtin_nan("");
    }
}

// ------------------------------------------------------------------------------------------
// Host side: tables and plan cache
// ------------------------------------------------------------------------------------------
typedef long double ld;

struct SosPlan {
    int K = 0;          // sections per band
    int NB = 1;         // bands (filter-bank mode: independent SOS sets sharing the input)
    int nsteps32 = 6, nsteps16 = 6, nsteps64 = 6;
    int64_t warm = -1;            // samples; -1 = too long / not decaying
    double err_bound_f32 = -1.0;  // worst-case |err| of f32 arithmetic for |x| <= 1 (lazy)
    void *tab_f64_lc32 = nullptr; // device
    void *tab_f32_lc32 = nullptr;
    void *tab_f64_lc16 = nullptr;
    void *tab_f32_lc16 = nullptr;
    void *tab_f64_lc64 = nullptr;
    void *tab_f32_lc64 = nullptr;
    void *tab_f64_lc64_unit = nullptr;   // unit-b0 form (fill_tables)
    int unit_ok = -1;                    // the cascade has a unit-b0 form (lazy)
    std::vector<double> sos;
};

static std::mutex g_plan_mu;
static std::map<std::vector<double>, SosPlan *> g_plans;

// one zero-input step of the whole cascade on state w = [x1,x2, y1_1,y1_2, ..., yK_1,yK_2]
static void cascade_step(const std::vector<double> &sos, int K, std::vector<ld> &w)
{
    std::vector<ld> nw(w.size());
    ld v = 0.0L;                     // current input sample = 0
    ld p1 = w[0], p2 = w[1];         // input history of section 0
    nw[0] = 0.0L; nw[1] = w[0];
    for (int s = 0; s < K; ++s) {
        const double *co = &sos[s * 6];
        const ld y1 = w[2 + 2 * s], y2 = w[3 + 2 * s];
        const ld yn = (ld)co[0] * v + (ld)co[1] * p1 + (ld)co[2] * p2 - (ld)co[4] * y1 - (ld)co[5] * y2;
        nw[2 + 2 * s] = yn; nw[3 + 2 * s] = y1;
        p1 = y1; p2 = y2;            // next section's input history = this section's old outputs
        v = yn;
    }
    w.swap(nw);
}

static std::vector<ld> matmul(const std::vector<ld> &a, const std::vector<ld> &b, int D)
{
    std::vector<ld> c((size_t)D * D, 0.0L);
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < D; ++k) {
            const ld aik = a[(size_t)i * D + k];
            if (aik == 0.0L) continue;
            for (int j = 0; j < D; ++j) c[(size_t)i * D + j] += aik * b[(size_t)k * D + j];
        }
    return c;
}
static ld maxabs(const std::vector<ld> &a)
{
    ld m = 0;
    for (ld v : a) { ld t = fabsl(v); if (!(t <= m)) m = t; }   // NaN-propagating
    return m;
}

// smallest W (with margin) such that max|A^W| < tol; -1 if not reached within 2^26 samples
static int64_t warmup_length(const std::vector<double> &sos, int K, int bits = 60)
{
    // O(K^3) long-double matrix squarings: beyond ~100 sections the analysis would cost seconds of host
    // time, so such cascades run as one sequential segment per row (exact, just less parallel)
    if (K > 96) return -1;
    const int D = 2 * K + 2;
    std::vector<ld> A((size_t)D * D, 0.0L);
    for (int j = 0; j < D; ++j) {
        std::vector<ld> w(D, 0.0L);
        w[j] = 1.0L;
        cascade_step(sos, K, w);
        for (int i = 0; i < D; ++i) A[(size_t)i * D + j] = w[i];
    }
    const ld tol = ldexpl(1.0L, -bits);
    std::vector<std::vector<ld>> pw;   // A^(2^i)
    pw.push_back(A);
    int m = 0;
    for (; m < 26; ++m) {
        ld nrm = maxabs(pw.back());
        if (!(nrm == nrm) || nrm > 1e300L) return -1;
        if (nrm < tol) break;
        pw.push_back(matmul(pw.back(), pw.back(), D));
    }
    if (m == 26) return -1;
    // greedy: largest n with max|A^n| >= tol
    std::vector<ld> cur;
    int64_t n = 0;
    for (int i = m - 1; i >= 0; --i) {
        std::vector<ld> cand = cur.empty() ? pw[i] : matmul(cur, pw[i], D);
        if (maxabs(cand) >= tol) { cur.swap(cand); n += (int64_t)1 << i; }
    }
    int64_t W = n + 1;
    W += W / 8 + 8;                    // margin: the norm is not strictly monotone
    return W;
}

// Whether the cascade can run in the unit-b0 form: every b0 non-zero, not dwarfed by its section's other numerator
// coefficients (b1 / b0 must stay a sane number) and every partial product of them far inside the float64 range.
static bool unit_form_ok(const std::vector<double> &sos, int K)
{
    if (K < 2) return false;
    ld g = 1.0L;
    for (int s = 0; s < K; ++s) {
        const ld b0 = sos[s * 6], b1 = sos[s * 6 + 1], b2 = sos[s * 6 + 2];
        g *= b0;
        if (!(fabsl(b0) > 0) || !(fabsl(b0) * 1e9L >= fmaxl(fabsl(b1), fabsl(b2))) || !(fabsl(g) > 1e-100L && fabsl(g) < 1e100L)) return false;
    }
    return true;
}

template <typename TC>
static void fill_tables(const std::vector<double> &sos, int K, int LC, std::vector<TC> &out, int &nsteps, bool unit = false)
{
    const int TS = tab_stride(LC);
    out.assign((size_t)K * TS, (TC)0);
    nsteps = 0;
    ld gin = 1.0L;
    for (int s = 0; s < K; ++s) {
        const double *co = &sos[s * 6];
        TC *tb = &out[(size_t)s * TS];
        if (unit) {
            tb[0] = (TC)1; tb[1] = (TC)((ld)co[1] / (ld)co[0]); tb[2] = (TC)((ld)co[2] / (ld)co[0]);
            const ld gout = gin * (ld)co[0];
            tb[160] = (TC)(1.0L / gin); tb[161] = (TC)(1.0L / gout); tb[162] = (TC)gin; tb[163] = (TC)gout;
            gin = gout;
        } else {
            tb[0] = (TC)co[0]; tb[1] = (TC)co[1]; tb[2] = (TC)co[2];
            tb[160] = tb[161] = tb[162] = tb[163] = (TC)1;
        }
        tb[3] = (TC)(-co[4]); tb[4] = (TC)(-co[5]);
        const ld a1 = co[4], a2 = co[5];
        // alpha: response to state (1,0); beta: to (0,1)
        ld al1 = 1, al2 = 0, be1 = 0, be2 = 1;
        ld P[4] = {1, 0, 0, 1};
        for (int n = 0; n < LC; ++n) {
            const ld al = -a1 * al1 - a2 * al2, be = -a1 * be1 - a2 * be2;
            al2 = al1; al1 = al; be2 = be1; be1 = be;
        }
        P[0] = al1; P[1] = be1; P[2] = al2; P[3] = be2;          // Cmp^LC
        {
            ld Q[4] = {P[0], P[1], P[2], P[3]};                   // Cmp^(LC (p+1))
            for (int pp = 0; pp < 32; ++pp) {
                for (int i = 0; i < 4; ++i) tb[32 + 4 * pp + i] = (TC)Q[i];
                const ld q0 = Q[0] * P[0] + Q[1] * P[2], q1 = Q[0] * P[1] + Q[1] * P[3];
                const ld q2 = Q[2] * P[0] + Q[3] * P[2], q3 = Q[2] * P[1] + Q[3] * P[3];
                Q[0] = q0; Q[1] = q1; Q[2] = q2; Q[3] = q3;
            }
        }
        int need = 0;
        for (int k = 0; k < 6; ++k) {
            for (int i = 0; i < 4; ++i) tb[8 + 4 * k + i] = (TC)P[i];
            ld m = 0;
            for (int i = 0; i < 4; ++i) m = fmaxl(m, fabsl(P[i]));
            if (!(m < 1e-22L)) need = k + 1;    // this step still contributes (or is NaN/inf)
            const ld q0 = P[0] * P[0] + P[1] * P[2], q1 = P[0] * P[1] + P[1] * P[3];
            const ld q2 = P[2] * P[0] + P[3] * P[2], q3 = P[2] * P[1] + P[3] * P[3];
            P[0] = q0; P[1] = q1; P[2] = q2; P[3] = q3;
        }
        if (need > nsteps) nsteps = need;
    }
}

// Estimate of the largest error of the float32 recursion against the float64 one for |x| <= 1 (what
// precision = "auto" decides on).  A rounding-noise model (noise variance x energy gain) is off by orders of
// magnitude exactly where it matters: with poles near z = 1 the chunked formulation adds zero-state
// responses that are 10^3..10^5 times larger than the output, which float64 shrugs off and float32 does not
// (HiButterworth-2 @ 20 Hz: error 3.0 on a signal of amplitude 1; profiles/r02_iir_f32_calibration.txt).  So
// the estimate is MEASURED: the kernel's float32 arithmetic -- same tile / lane structure (LC = 32), same
// operation order, same float32 tables and scan tree -- is replayed on the host for 2^16 pseudo-random
// samples per cascade (a millisecond, once per plan) against the sequential float64 recursion, and the
// largest difference is scaled by 2.5 (device runs of 4.6e7 samples per cascade measure 0.54 .. 1.02 of twice the
// replayed error over five orders of magnitude of it, tools/iir_f32_calibrate.py).
template <typename TC> static void fill_tables(const std::vector<double> &sos, int K, int LC, std::vector<TC> &out, int &nsteps, bool unit);

static double f32_error_bound(const std::vector<double> &sos, int K)
{
    constexpr int LC = 32, TILE = 64 * LC;
    const int TS = tab_stride(LC);
    const int64_t N = 1 << 16;
    std::vector<float> tab;
    int nsteps = 0;
    fill_tables<float>(sos, K, LC, tab, nsteps, false);
    for (float v : tab) if (!std::isfinite(v)) return INFINITY;
    // input: uniform in [-1, 1], fixed LCG
    std::vector<float> x((size_t)N);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    for (auto &v : x) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = (float)((double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
    // float64 reference, sequential DF1
    std::vector<double> ref(x.begin(), x.end());
    for (int s = 0; s < K; ++s) {
        const double *co = &sos[s * 6];
        double x1 = 0, x2 = 0, y1 = 0, y2 = 0;
        for (int64_t n = 0; n < N; ++n) {
            const double v = ref[(size_t)n];
            const double y = co[0] * v + co[1] * x1 + co[2] * x2 - co[4] * y1 - co[5] * y2;
            x2 = x1; x1 = v; y2 = y1; y1 = y;
            ref[(size_t)n] = y;
        }
    }
    // float32 replay of sos_stream_kernel (one stream, no time segmentation)
    std::vector<float> cur(x);
    std::vector<float> carry((size_t)K * 4, 0.0f);        // vin1 vin2 y1 y2 per section
    double worst = 0.0, scale = 1.0;
    for (int64_t ts = 0; ts < N; ts += TILE) {
        float *d = &cur[(size_t)ts];                        // d[lane * LC + n]
        for (int s = 0; s < K; ++s) {
            const float *tb = &tab[(size_t)s * TS];
            const float b0 = tb[0], b1 = tb[1], b2 = tb[2], na1 = tb[3], na2 = tb[4];
            float *cs = &carry[(size_t)s * 4];
            float pv1[64], pv2[64];
            for (int l = 0; l < 64; ++l) {
                pv1[l] = l ? d[(l - 1) * LC + LC - 1] : cs[0];
                pv2[l] = l ? d[(l - 1) * LC + LC - 2] : cs[1];
            }
            cs[0] = d[63 * LC + LC - 1]; cs[1] = d[63 * LC + LC - 2];
            float s0[64], s1[64];
            for (int l = 0; l < 64; ++l) {
                float *c = d + l * LC;
                for (int n = LC - 1; n >= 0; --n) {          // (1a) in place, descending
                    const float x1 = n >= 1 ? c[n - 1] : pv1[l];
                    const float x2 = n >= 2 ? c[n - 2] : (n == 1 ? pv1[l] : pv2[l]);
                    c[n] = fmaf(b2, x2, fmaf(b1, x1, b0 * c[n]));
                }
                float u1 = l ? 0.0f : cs[2], u2 = l ? 0.0f : cs[3];   // (1b) end state from zero (lane 0: carried) state
                for (int n = 0; n < LC; ++n) { const float u = fmaf(na1, u1, fmaf(na2, u2, c[n])); u2 = u1; u1 = u; }
                s0[l] = u1; s1[l] = u2;
            }
            const float *pm = tb + 8;                        // (2) the scan tree of the kernel
            for (int k = 0; k < 4; ++k) {
                float t0[64], t1[64];
                for (int l = 0; l < 64; ++l) { const bool src = (l & 15) >= (1 << k); t0[l] = src ? s0[l - (1 << k)] : 0.0f; t1[l] = src ? s1[l - (1 << k)] : 0.0f; }
                for (int l = 0; l < 64; ++l) {
                    s0[l] += fmaf(pm[4 * k + 0], t0[l], pm[4 * k + 1] * t1[l]);
                    s1[l] += fmaf(pm[4 * k + 2], t0[l], pm[4 * k + 3] * t1[l]);
                }
            }
            {
                const float *mp = tb + 32;
                float a0[64], a1[64];
                for (int l = 0; l < 64; ++l) { const bool on = (l >> 4) & 1; a0[l] = on ? s0[(l & ~15) - 1] : 0.0f; a1[l] = on ? s1[(l & ~15) - 1] : 0.0f; }
                for (int l = 0; l < 64; ++l) {
                    const float *m = mp + 4 * (l & 15);
                    s0[l] += fmaf(m[0], a0[l], m[1] * a1[l]);
                    s1[l] += fmaf(m[2], a0[l], m[3] * a1[l]);
                }
                const float c0 = s0[31], c1 = s1[31];
                for (int l = 32; l < 64; ++l) {
                    const float *m = mp + 4 * (l & 31);
                    const float n0 = s0[l] + fmaf(m[0], c0, m[1] * c1), n1 = s1[l] + fmaf(m[2], c0, m[3] * c1);
                    s0[l] = n0; s1[l] = n1;
                }
            }
            const float cy1 = cs[2], cy2 = cs[3];
            for (int l = 0; l < 64; ++l) {                   // (3) the recursion from the true start state
                float h1 = l ? s0[l - 1] : cy1, h2 = l ? s1[l - 1] : cy2;
                float *c = d + l * LC;
                for (int n = 0; n < LC; ++n) {
                    const float yv = fmaf(na1, h1, fmaf(na2, h2, c[n]));
                    c[n] = yv; h2 = h1; h1 = yv;
                }
            }
            cs[2] = d[63 * LC + LC - 1]; cs[3] = d[63 * LC + LC - 2];
        }
        for (int i = 0; i < TILE; ++i) {
            const double e = fabs((double)d[i] - ref[(size_t)ts + i]);
            if (!(e <= worst)) worst = e;                     // NaN-propagating
            scale = fmax(scale, fabs(ref[(size_t)ts + i]));
        }
    }
    return 2.5 * worst / scale;
}

static void free_plan(SosPlan *pl)
{
    void *t[7] = {pl->tab_f64_lc32, pl->tab_f32_lc32, pl->tab_f64_lc16, pl->tab_f32_lc16, pl->tab_f64_lc64, pl->tab_f32_lc64, pl->tab_f64_lc64_unit};
    for (void *q : t) if (q) (void)hipFree(q);
    delete pl;
}
static double plan_err_bound(SosPlan *pl)
{
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (pl->err_bound_f32 < 0) {
        double worst = 0.0;
        for (int b = 0; b < pl->NB; ++b) {
            std::vector<double> one(pl->sos.begin() + (size_t)b * pl->K * 6, pl->sos.begin() + (size_t)(b + 1) * pl->K * 6);
            worst = fmax(worst, f32_error_bound(one, pl->K));
        }
        pl->err_bound_f32 = worst;
    }
    return pl->err_bound_f32;
}
static double auto_bound()
{
    const char *e = getenv("TFX_AUTO_F32_BOUND");
    // precision = "auto": float32 when the estimated largest error (relative to max(1, max|y|)) stays below this --
    // a fifth of the tolerance of the reference's own CUDA path (1e-4, tests/test_cuda_kernels.py:23-24)
    return (e && *e) ? atof(e) : 2e-5;
}

static SosPlan *get_plan(const double *sos_host, int64_t K, hipStream_t stream, int64_t NB = 1)
{
    std::vector<double> key(sos_host, sos_host + NB * K * 6);
    key.push_back((double)NB);             // a bank of NB x K sections is not a cascade of NB*K
    key.push_back((double)current_device());   // the tables live on one device
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    if (g_plans.size() > 256) {   // bound the cache
        for (auto &kv : g_plans) free_plan(kv.second);
        g_plans.clear();
    }
    SosPlan *pl = new SosPlan();
    pl->K = (int)K;
    pl->NB = (int)NB;
    pl->sos.assign(sos_host, sos_host + NB * K * 6);
    pl->warm = 0;
    for (int64_t b = 0; b < NB; ++b) {     // every band must have forgotten its start state
        std::vector<double> one(sos_host + b * K * 6, sos_host + (b + 1) * K * 6);
        const int64_t w = warmup_length(one, (int)K);
        if (w < 0) { pl->warm = -1; break; }
        if (w > pl->warm) pl->warm = w;
    }
    g_plans[key] = pl;
    return pl;
}

template <typename TC>
static void *ensure_table(SosPlan *pl, void **slot, int LC, int *nsteps, hipStream_t stream, bool unit = false)
{
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (!*slot) {
        std::vector<TC> h;
        for (int b = 0; b < pl->NB; ++b) {
            std::vector<double> one(pl->sos.begin() + (size_t)b * pl->K * 6, pl->sos.begin() + (size_t)(b + 1) * pl->K * 6);
            std::vector<TC> hb;
            fill_tables<TC>(one, pl->K, LC, hb, *nsteps, unit);
            h.insert(h.end(), hb.begin(), hb.end());
        }
        void *d = nullptr;
        TFX_HIP(hipMalloc(&d, h.size() * sizeof(TC)));
        // synchronous copy from a temporary: happens once per distinct filter
        TFX_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(TC), hipMemcpyHostToDevice));
        *slot = d;
    }
    return *slot;
}

void sos_clear_plans()
{
    std::lock_guard<std::mutex> lk(g_plan_mu);
    for (auto &kv : g_plans) {
        free_plan(kv.second);
    }
    g_plans.clear();
}

static int env_int(const char *name, int dflt) { return (int)env_i64(name, dflt); }      // read once per process (common.h)

static int g_cus[TFX_MAX_DEVICES] = {0};
static int device_cus()
{
    const int dev = current_device();
    if (!g_cus[dev]) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) == hipSuccess) g_cus[dev] = pr.multiProcessorCount;
        if (g_cus[dev] <= 0) g_cus[dev] = 256;
    }
    return g_cus[dev];
}

// Time segmentation.  Streams are persistent (one wavefront walks its whole segment), so the
// launch should be exactly ONE resident round: nstreams <= CUs x resident waves, otherwise the
// second round doubles the time.  `resident_waves_per_cu` comes from the occupancy query of the
// kernel actually launched.  Segments shorter than TFX_SOS_MIN_SEG_OVER_WARM x warm-up are not
// worth their halo.
static void plan_segments(SosParams &p, int64_t plan_warm, int TILE, int resident_waves_per_cu)
{
    const int64_t tiles_total = ceil_div(p.T, TILE);
    int64_t nseg = 1, seg_len = tiles_total * TILE, warm = 0;
    const int force_nseg = env_int("TFX_SOS_NSEG", 0);
    if (plan_warm >= 0) {
        {
            // halo rounded to 1 KB of float32: streams start on cache-line boundaries; 256 measured 1.5-4 % faster
            // than 32 on the float64 kernel (64 x 2.88 M / 10 M / 28.8 M), no difference beyond
            const int64_t q = env_int("TFX_SOS_WARM_ROUND", 256);         // samples; power of two >= 32
            warm = (plan_warm + q - 1) & ~(q - 1);
        }
        const int wpc = env_int("TFX_SOS_WAVES_PER_CU", 0);
        const int64_t capacity = (int64_t)device_cus() * (wpc > 0 ? wpc : resident_waves_per_cu);
        int64_t nseg_target = force_nseg > 0 ? force_nseg : capacity / p.C;      // floor: one round
        if (nseg_target < 1) nseg_target = 1;
        if (nseg_target > 1 && p.T > warm) {
            // Every stream walks seg_tiles FULL tiles: stream g starts at g * stride, stride = seg_tiles * TILE - warm,
            // and its first `warm` samples (g > 0) are the halo -- the halo is absorbed into the tile grid instead of
            // costing every stream an extra, mostly discarded tile (+4 % at 64 x 2.88 M with 4096-sample tiles).
            int64_t seg_tiles = ceil_div(ceil_div(p.T - warm, nseg_target) + warm, TILE);
            if (force_nseg <= 0) {
                const int64_t min_tiles = ceil_div(env_int("TFX_SOS_MIN_SEG_OVER_WARM", 8) * warm, TILE);
                if (seg_tiles < min_tiles) seg_tiles = min_tiles;
            }
            if (seg_tiles < 1) seg_tiles = 1;
            const int64_t stride = seg_tiles * TILE - warm;
            if (stride > 0) {
                nseg = ceil_div(p.T - warm, stride);
                seg_len = stride;
            }
        }
        if (nseg <= 1) { nseg = 1; warm = 0; seg_len = tiles_total * TILE; }
    }
    p.nseg = (int)nseg; p.seg_len = seg_len; p.warm = warm;
}

template <typename TIn, typename TOut, typename TC, int LC, bool VEC, bool TAPS, bool PF, int MINW, bool SUMB = false, bool EPI = false, bool UNIT = false>
static void launch_one(SosParams p, int64_t plan_warm, hipStream_t stream)
{
    constexpr int IOB = sizeof(TIn) > sizeof(TOut) ? sizeof(TIn) : sizeof(TOut);
    constexpr int STAGE_B = 64 * (LC * IOB + 16);
    const int nbl = SUMB ? p.nsum : 1;
    const int carry_b = (((nbl * p.K * 4 + 2 * LC) * (int)sizeof(TC)) + 15) & ~15;
    const size_t shmem = 4 * (size_t)(STAGE_B + carry_b);
    TFX_CHECK(shmem <= 160 * 1024, "sos_forward: %d band(s) x K=%d need %zu B of LDS (max 163840)", nbl, p.K, shmem);
    auto kern = sos_stream_kernel<TIn, TOut, TC, LC, VEC, TAPS, PF, MINW, SUMB, EPI, UNIT>;
    if (!EPI) p.ep_stat = -1;                      // the plain instantiation has no epilogue code
    if (shmem > 64 * 1024)
        TFX_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    static int blocks_per_cu_tab[TFX_MAX_DEVICES] = {0};     // per template instance and device
    static size_t blocks_shmem_tab[TFX_MAX_DEVICES] = {0};
    const int dev = current_device();
    int &blocks_per_cu = blocks_per_cu_tab[dev];
    size_t &blocks_shmem = blocks_shmem_tab[dev];
    if (!blocks_per_cu || blocks_shmem != shmem) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, 256, shmem) != hipSuccess || nb < 1) nb = 1;
        blocks_per_cu = nb;
        blocks_shmem = shmem;
    }
    plan_segments(p, plan_warm, 64 * LC, blocks_per_cu * 4);
    p.fair_nw = blocks_per_cu < 2 ? 1 : (blocks_per_cu > 4 ? 4 : blocks_per_cu);      // one wave of every resident workgroup per SIMD
    if (p.fair_nw < 2) p.fair = 0;
    const int64_t nstreams = p.C * p.nseg;
    const unsigned grid = (unsigned)ceil_div(nstreams, 4);
    if (p.ep_stat >= 0) {                  // streams that have nothing to store leave their (zeroed) slot alone
        p.ep_partial = (double *)scratch("sos_ep_partial", (size_t)nstreams * sizeof(double), stream);
        TFX_HIP(hipMemsetAsync(p.ep_partial, 0, (size_t)nstreams * sizeof(double), stream));
    }
    p.nf_flag = nullptr;
    if (p.nseg > 1)                        // see sos_nonfinite_fix_kernel: every stream writes its slot, no memset needed
        p.nf_flag = (int *)scratch("sos_nf_flag", (size_t)nstreams * sizeof(int), stream);
    {
        ProfScope ps(sizeof(TC) == 8 ? "sos_stream_kernel<f64>" : "sos_stream_kernel<f32>", stream);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, stream, p);
        TFX_HIP(hipGetLastError());
    }
    if (p.nseg > 1) {                      // ~2 us on cfg 2 (measured by leaving it out)
        hipLaunchKernelGGL(sos_nonfinite_fix_kernel<TOut>, dim3((unsigned)p.C), dim3(256), 0, stream, p, nbl, SUMB ? p.C_in * nbl : p.C);
        TFX_HIP(hipGetLastError());
    }
    if (p.ep_stat >= 0) {
        const Epilogue *ep = (const Epilogue *)p.ep_host;
        stat_finish(p.ep_partial, ep->per_row ? p.C : 1, ep->per_row ? p.nseg : nstreams, p.ep_stat, ep->stat_out, stream);
    }
}

// Variants (TFX_SOS_VARIANT): 0 = LC32, 1 = LC16, 2 = LC32 + register prefetch, 3 = LC16 + prefetch.
// Register budgets (waves/SIMD) were chosen from -Rpass-analysis so that nothing spills.
template <typename TIn, typename TOut, typename TC>
static void launch_main(const SosParams &p, bool vec, int variant, int64_t nstreams, hipStream_t stream)   // nstreams = plan warm-up
{
    constexpr bool F32 = sizeof(TC) == 4;
    if (p.taps && vec && variant >= 4) {   // the shipping LC = 64 geometry with every section's output tapped (parity tests)
        launch_one<TIn, TOut, TC, 64, true, true, false, F32 ? 3 : 2>(p, nstreams, stream);
        return;
    }
    if (p.taps || !vec) {     // debug taps / unaligned rows: plain dword path
        if (variant & 1) {
            if (p.taps) launch_one<TIn, TOut, TC, 16, false, true, false, F32 ? 5 : 3>(p, nstreams, stream);
            else launch_one<TIn, TOut, TC, 16, false, false, false, F32 ? 5 : 3>(p, nstreams, stream);
        } else {
            if (p.taps) launch_one<TIn, TOut, TC, 32, false, true, false, F32 ? 3 : 2>(p, nstreams, stream);
            else launch_one<TIn, TOut, TC, 32, false, false, false, F32 ? 3 : 2>(p, nstreams, stream);
        }
        return;
    }
    if constexpr (!F32) {
        if (p.unit) {          // unit-b0 form of the shipping geometry, plain and with the epilogue (same cascade arithmetic in both)
            if (p.ep_fused) launch_one<TIn, TOut, TC, 64, true, false, false, 2, false, true, true>(p, nstreams, stream);
            else launch_one<TIn, TOut, TC, 64, true, false, false, 2, false, false, true>(p, nstreams, stream);
            return;
        }
    }
    if (p.ep_fused) {          // epilogue instantiation: same tile geometry as the plain kernel (bit-identical cascade output)
        if (variant >= 4) launch_one<TIn, TOut, TC, 64, true, false, false, F32 ? 3 : 2, false, true>(p, nstreams, stream);
        else launch_one<TIn, TOut, TC, 32, true, false, true, 2, false, true>(p, nstreams, stream);
        return;
    }
    switch (variant) {
    case 1: launch_one<TIn, TOut, TC, 16, true, false, false, F32 ? 8 : 5>(p, nstreams, stream); break;
    case 2: launch_one<TIn, TOut, TC, 32, true, false, true, F32 ? 4 : 2>(p, nstreams, stream); break;
    case 3: launch_one<TIn, TOut, TC, 16, true, false, true, F32 ? 6 : 4>(p, nstreams, stream); break;
    case 4: launch_one<TIn, TOut, TC, 64, true, false, false, F32 ? 3 : 2>(p, nstreams, stream); break;   // 64 samples per lane: half the scan per sample
    case 5: launch_one<TIn, TOut, TC, 64, true, false, true, F32 ? 3 : 1>(p, nstreams, stream); break;
    default: launch_one<TIn, TOut, TC, 32, true, false, false, F32 ? 5 : 3>(p, nstreams, stream); break;
    }
}
// rarely used dtype mixes: one configuration only
template <typename TIn, typename TOut, typename TC>
static void launch_rare(const SosParams &p, bool vec, int64_t nstreams, hipStream_t stream)   // nstreams = plan warm-up
{
    if (p.taps) launch_one<TIn, TOut, TC, 16, false, true, false, 3>(p, nstreams, stream);
    else if (vec) launch_one<TIn, TOut, TC, 16, true, false, false, 4>(p, nstreams, stream);
    else launch_one<TIn, TOut, TC, 16, false, false, false, 3>(p, nstreams, stream);
}

// sum mode: one configuration per dtype mix (LC = 16: the accumulator costs registers)
template <typename TIn, typename TOut, typename TC>
static void launch_sum(const SosParams &p, bool vec, int64_t plan_warm, hipStream_t stream)
{
    if (vec) launch_one<TIn, TOut, TC, 16, true, false, false, 3, true>(p, plan_warm, stream);
    else launch_one<TIn, TOut, TC, 16, false, false, false, 3, true>(p, plan_warm, stream);
}

// NB > 1 = filter-bank mode: NB independent K-section cascades applied to the same C_in input
// rows; output rows (and state rows) are band-major: row = band * C_in + c.  C is the number of
// INPUT rows.  sum_bands: the bands' outputs are accumulated into C_in output rows (`+`); the state
// tensors keep the band-major [K, NB * C_in, 2] layout.
void sos_forward(const void *x, int x_dtype, void *y, int y_dtype, int64_t C_in, int64_t T,
                 const double *sos_host, int64_t K,
                 const double *sx_in, const double *sy_in, double *sx_out, double *sy_out,
                 void *y_sections, int precision, hipStream_t stream, int64_t NB, bool sum_bands, const Epilogue *ep)
{
    Epilogue none;
    if (!ep) ep = &none;
    TFX_CHECK(!(ep->any() && y_sections), "sos_forward: no epilogue together with section taps");
    TFX_CHECK(ep->stat_mode < 0 || ep->stat_out, "sos_forward: statistic requested without an output buffer");
    TFX_CHECK(NB >= 1, "sos_forward: need at least one band");
    TFX_CHECK(!(sum_bands && y_sections), "sos_forward: no section taps in sum mode");
    const int64_t C = sum_bands ? C_in : C_in * NB;
    TFX_CHECK(C >= 0 && T >= 0 && K >= 0, "sos_forward: negative size");
    // the per-wave carry (4 values per section) lives in LDS next to the transposition stage
    TFX_CHECK(K <= 512, "sos_forward: at most 512 sections per cascade (got %lld); split the cascade", (long long)K);
    TFX_CHECK(x_dtype == TFX_F32 || x_dtype == TFX_F64, "sos_forward: bad x dtype %d", x_dtype);
    TFX_CHECK(y_dtype == TFX_F32 || y_dtype == TFX_F64, "sos_forward: bad y dtype %d", y_dtype);
    if (C == 0) return;
    TFX_CHECK((T == 0 || (x && y)) && (K == 0 || sos_host), "sos_forward: null signal or coefficient pointer");   // an empty tensor has no storage
    const size_t st_bytes = (size_t)K * C_in * NB * 2 * sizeof(double);   // [K, NB * C_in, 2] in every mode
    TFX_CHECK(!(sum_bands && K == 0), "sos_forward: sum mode needs at least one section");
    if (T == 0 || K == 0) {
        // no samples: state passes through (iir_cpu.cpp writes back what it loaded);
        // no sections: y = x
        if (sx_out && K) { if (sx_in) TFX_HIP(hipMemcpyAsync(sx_out, sx_in, st_bytes, hipMemcpyDeviceToDevice, stream));
                           else TFX_HIP(hipMemsetAsync(sx_out, 0, st_bytes, stream)); }
        if (sy_out && K) { if (sy_in) TFX_HIP(hipMemcpyAsync(sy_out, sy_in, st_bytes, hipMemcpyDeviceToDevice, stream));
                           else TFX_HIP(hipMemsetAsync(sy_out, 0, st_bytes, stream)); }
        if (K == 0 && T > 0) {
            TFX_CHECK(x_dtype == y_dtype && NB == 1, "sos_forward: K=0 needs equal dtypes and a single band");
            TFX_HIP(hipMemcpyAsync(y, x, (size_t)C * T * (x_dtype == TFX_F32 ? 4 : 8), hipMemcpyDeviceToDevice, stream));
        }
        if (ep->any()) epilogue_as_passes(y, y_dtype, C, K == 0 ? T : 0, *ep, stream);
        return;
    }
    for (int64_t i = 0; i < NB * K * 6; ++i)
        TFX_CHECK(std::isfinite(sos_host[i]), "sos_forward: non-finite SOS coefficient");

    SosPlan *pl = get_plan(sos_host, K, stream, NB);
    int prec = precision;
    if (prec == TFX_PREC_AUTO) prec = (plan_err_bound(pl) <= auto_bound()) ? TFX_PREC_F32 : TFX_PREC_F64;
    if (x_dtype == TFX_F64 || y_dtype == TFX_F64) prec = TFX_PREC_F64;   // f64 signals: always f64 math
    const bool rare = !(x_dtype == TFX_F32 && y_dtype == TFX_F32);

    // float32 arithmetic: LC = 32 + register prefetch (4 waves per SIMD); float64: LC = 64 -- the kernel is bound by VALU
    // issue (1440 instructions per 2048-sample tile, ~100 % busy at 3 waves per SIMD), and 64 samples per lane halve the
    // scan's share per sample (0.352 vs 0.38-0.42 ms at cfg 2 on the same box)
    int variant = (rare || sum_bands) ? 1 : env_int("TFX_SOS_VARIANT", -1);
    if (variant < 0) variant = (prec == TFX_PREC_F32) ? 2 : 4;
    {
        const int xs_ = x_dtype == TFX_F32 ? 4 : 8, ys_ = y_dtype == TFX_F32 ? 4 : 8;
        const bool vec_ = (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 15) == 0) && ((T * xs_) % 16 == 0) && ((T * ys_) % 16 == 0);
        if (variant >= 4 && !vec_) variant = 2;                               // LC = 64 exists for the aligned (16-byte) path only
        if (variant >= 4 && ep->any()) variant = 4;                           // (its epilogue instantiation: no register prefetch)
    }
    // fused into the kernel on the main path (float32 I/O, aligned rows, no taps, single cascade); everything
    // else runs the plain kernel and the same arithmetic as separate passes over y
    const bool ep_fused = ep->any() && x_dtype == TFX_F32 && y_dtype == TFX_F32 && !sum_bands && !y_sections &&
                          (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 15) == 0) && ((T * 4) % 16 == 0);
    if (ep_fused && variant < 4) variant = 2;      // the epilogue kernel exists for LC = 32 and LC = 64: table and kernel must agree
    const int LC = variant >= 4 ? 64 : ((variant & 1) ? 16 : 32);
    SosParams p{};
    p.x = x; p.y = y; p.taps = y_sections;
    p.sx_in = sx_in; p.sy_in = sy_in; p.sx_out = sx_out; p.sy_out = sy_out;
    p.C = C; p.C_in = C_in; p.T = T; p.K = (int)K; p.x_pitch = T;
    p.nt = env_int("TFX_SOS_NT", 1);
    p.fair = env_int("TFX_SOS_FAIR", 15);
    p.nsum = sum_bands ? (int)NB : 0;
    p.ep_gain = ep->gain; p.ep_scale = ep->scale; p.ep_clamp = ep->clamp; p.ep_stat = ep->stat_mode;
    p.ep_partial = nullptr; p.ep_host = ep;
    p.ep_fused = ep_fused ? 1 : 0;
    if (!ep_fused) { p.ep_scale = p.ep_clamp = 0; p.ep_stat = -1; }

    const int64_t nstreams = pl->warm;     // segmentation is decided per kernel instance (launch_one)

    const int xsz = x_dtype == TFX_F32 ? 4 : 8, ysz = y_dtype == TFX_F32 ? 4 : 8;
    const bool vec = (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 15) == 0) &&
                     ((T * xsz) % 16 == 0) && ((T * ysz) % 16 == 0);

    if (prec == TFX_PREC_F32) {
        p.tab = ensure_table<float>(pl, LC == 64 ? &pl->tab_f32_lc64 : (LC == 32 ? &pl->tab_f32_lc32 : &pl->tab_f32_lc16), LC,
                                    LC == 64 ? &pl->nsteps64 : (LC == 32 ? &pl->nsteps32 : &pl->nsteps16), stream);
        p.nsteps = LC == 64 ? pl->nsteps64 : (LC == 32 ? pl->nsteps32 : pl->nsteps16);
        if (sum_bands) launch_sum<float, float, float>(p, vec, nstreams, stream);
        else launch_main<float, float, float>(p, vec, variant, nstreams, stream);
    } else {
        // the shipping float32-in / float32-out geometry (LC = 64, aligned rows, no section taps) runs the unit-b0 form when the
        // cascade has one (TFX_SOS_UNIT_B0=0: plain form)
        if (LC == 64 && variant == 4 && vec && !p.taps && !sum_bands && x_dtype == TFX_F32 && y_dtype == TFX_F32 && env_int("TFX_SOS_UNIT_B0", 1) != 0) {
            if (pl->unit_ok < 0) {
                bool ok = true;
                for (int b = 0; b < pl->NB && ok; ++b)
                    ok = unit_form_ok(std::vector<double>(pl->sos.begin() + (size_t)b * pl->K * 6, pl->sos.begin() + (size_t)(b + 1) * pl->K * 6), pl->K);
                pl->unit_ok = ok ? 1 : 0;
            }
            p.unit = pl->unit_ok;
        }
        if (p.unit)
            p.tab = ensure_table<double>(pl, &pl->tab_f64_lc64_unit, 64, &pl->nsteps64, stream, true);
        else
        p.tab = ensure_table<double>(pl, LC == 64 ? &pl->tab_f64_lc64 : (LC == 32 ? &pl->tab_f64_lc32 : &pl->tab_f64_lc16), LC,
                                     LC == 64 ? &pl->nsteps64 : (LC == 32 ? &pl->nsteps32 : &pl->nsteps16), stream);
        p.nsteps = LC == 64 ? pl->nsteps64 : (LC == 32 ? pl->nsteps32 : pl->nsteps16);
        if (sum_bands) {
            if (x_dtype == TFX_F32 && y_dtype == TFX_F32) launch_sum<float, float, double>(p, vec, nstreams, stream);
            else if (x_dtype == TFX_F64 && y_dtype == TFX_F64) launch_sum<double, double, double>(p, vec, nstreams, stream);
            else TFX_CHECK(false, "sos_forward: sum mode needs equal input and output dtypes");
        } else if (x_dtype == TFX_F32 && y_dtype == TFX_F32) launch_main<float, float, double>(p, vec, variant, nstreams, stream);
        else if (x_dtype == TFX_F32) launch_rare<float, double, double>(p, vec, nstreams, stream);
        else if (y_dtype == TFX_F32) launch_rare<double, float, double>(p, vec, nstreams, stream);
        else launch_rare<double, double, double>(p, vec, nstreams, stream);
    }
    if (ep->any() && !ep_fused) epilogue_as_passes(y, y_dtype, C, T, *ep, stream);
}

// ------------------------------------------------------------------------------------------
// One launch per small streaming chunk:  SOS cascade -> direct FIR with carried history -> gain / clip
// (the reference's small-block caller is RealtimeProcessor._audio_callback, realtime/processor.py:253-292, over
// StreamProcessor's chunk loop, realtime/stream.py:234-273: a 2 x 512 block is launch-bound -- four to five kernels
// for a few microseconds of arithmetic).  One workgroup per channel: wave 0 runs the cascade stream body on the chunk
// (carried DF1 state in / out, float64 or float32 arithmetic, output rounded to float32 exactly like the IIR module's)
// straight into an LDS buffer that sits behind the channel's FIR history, then all four waves convolve [history |
// cascade output] with the taps from LDS, apply the gain / clip to what they store and write the next history.
// Same arithmetic as the staged passes (same cascade code with LC = 16, same float32 FMA order as fir_direct_simple_kernel).
// ------------------------------------------------------------------------------------------
struct ChunkParams {
    SosParams sos;           // x = the chunk [C, T] with row pitch sos.x_pitch; y is replaced by the LDS buffer inside the kernel
    const float *taps;       // device: [Hpad - H zeros | flipped taps (Kf) | zeros up to Hpad + 4]   (cached_taps)
    const float *hist_in;    // [C, Kf - 1] or null (= silence)
    float *hist_out;         // [C, Kf - 1] or null
    float *y;                // [C, T]
    int Kf;
    int Tpad;                // T rounded up to a multiple of 4
    float gain;
    int scale, clamp;
};

template <typename TC, bool VEC>
__global__ void __launch_bounds__(1024) chunk_iir_fir_kernel(const ChunkParams q)
{
    constexpr int LC = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t c = blockIdx.x;
    const int T = (int)q.sos.T, H = q.Kf - 1;
    const int Hpad = (H + 3) & ~3;                         // the cascade output starts on a 16-byte boundary
    const int Kp = Hpad + 4;                               // padded tap count: leading zeros align the history, trailing ones fill the float4
    // ubuf = [Hpad - H zeros | history (H) | cascade output (T) | zeros]: every float4 the FIR phase touches is aligned
    float *ubuf = (float *)smem;                           // [Hpad + Tpad + 8]
    float *kp = ubuf + Hpad + q.Tpad + 8;                  // [Kp]
    char *stage = (char *)(kp + Kp);                       // cascade stage + carry (wave 0)
    float *u = ubuf + Hpad;
    for (int i = tid; i < Hpad; i += nthr) {
        const int j = i - (Hpad - H);
        ubuf[i] = (j >= 0 && q.hist_in) ? q.hist_in[c * H + j] : 0.0f;
    }
    for (int i = T + tid; i < q.Tpad + 8; i += nthr) u[i] = 0.0f;
    for (int i = tid; i < Kp; i += nthr) kp[i] = q.taps[i];
    if (q.sos.K == 0) {
        const float *xr = (const float *)q.sos.x + c * q.sos.x_pitch;
        for (int i = tid; i < T; i += nthr) u[i] = xr[i];
    } else if (wave == 0) {
        SosParams p = q.sos;
        p.y = (void *)(u - c * (int64_t)T);                // the body stores row c at y + c * T
        p.C_in = p.C;
        sos_stream_body<float, float, TC, LC, VEC, false, false, false, false>(p, c, stage, lane);
    }
    __syncthreads();
    // direct form, float32 FMA in tap order (as fir_direct_simple_kernel): y[n] = sum_k kp[k] * ubuf[n + k].  A thread owns
    // four consecutive outputs and slides an 8-sample register window over the taps: one ds_read_b128 of samples and one
    // (broadcast) of taps per 16 FMAs.
    const float4 *ub4 = (const float4 *)ubuf, *kp4 = (const float4 *)kp;
    for (int n0 = 4 * tid; n0 < T; n0 += 4 * nthr) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float4 a = ub4[n0 >> 2];
        for (int k = 0; k < Kp; k += 4) {
            const float4 b = ub4[((n0 + k) >> 2) + 1];
            const float4 h = kp4[k >> 2];
            const float win[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[r] = fmaf(h.x, win[r], acc[r]);
                acc[r] = fmaf(h.y, win[r + 1], acc[r]);
                acc[r] = fmaf(h.z, win[r + 2], acc[r]);
                acc[r] = fmaf(h.w, win[r + 3], acc[r]);
            }
            a = b;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + r;
            if (n < T) {
                float o = acc[r];
                if (q.scale) o *= q.gain;
                if (q.clamp) o = clamp_unit(o);
                q.y[c * T + n] = o;
            }
        }
    }
    if (q.hist_out)
        for (int i = tid; i < H; i += nthr) q.hist_out[c * H + i] = ubuf[(Hpad - H) + T + i];      // the last H samples of [hist | u]
}

bool chunk_supported(int64_t C, int64_t T, int64_t K, int64_t Kf)
{
    // one workgroup per channel, everything in LDS; the FIR phase costs T * Kf / 1024 FMAs per thread
    return C >= 1 && T >= 1 && T <= 4096 && K >= 0 && K <= 64 && Kf >= 1 && Kf <= 4096 && T * Kf <= ((int64_t)1 << 22);
}

void chunk_forward(const float *x, int64_t x_pitch, float *y, int64_t C, int64_t T, const double *sos_host, int64_t K,
                   const double *sx_in, const double *sy_in, double *sx_out, double *sy_out,
                   const float *taps_host, int64_t Kf, const float *hist_in, float *hist_out,
                   double gain, int scale, int clamp, int precision, hipStream_t stream)
{
    TFX_CHECK(chunk_supported(C, T, K, Kf), "chunk_forward: unsupported geometry C=%lld T=%lld K=%lld taps=%lld "
              "(T <= 4096, K <= 64, T * taps <= 2^22)", (long long)C, (long long)T, (long long)K, (long long)Kf);
    TFX_CHECK(x && y && taps_host && (K == 0 || sos_host), "chunk_forward: null pointer");
    TFX_CHECK(hist_out == nullptr || hist_out != hist_in, "chunk_forward: the new history needs its own buffer");   // NULL = silence in / no history out
    ChunkParams q{};
    SosParams &p = q.sos;
    p.x = x; p.y = nullptr; p.taps = nullptr;
    p.sx_in = sx_in; p.sy_in = sy_in; p.sx_out = sx_out; p.sy_out = sy_out;
    if (x_pitch <= 0) x_pitch = T;
    TFX_CHECK(x_pitch >= T, "chunk_forward: row pitch %lld smaller than the row length %lld", (long long)x_pitch, (long long)T);
    p.C = C; p.C_in = C; p.T = T; p.K = (int)K; p.x_pitch = x_pitch;
    p.nseg = 1; p.warm = 0; p.seg_len = ceil_div(T, 1024) * 1024; p.nsum = 0;
    p.ep_stat = -1; p.nf_flag = nullptr;
    int prec = precision;
    if (K > 0) {
        for (int64_t i = 0; i < K * 6; ++i) TFX_CHECK(std::isfinite(sos_host[i]), "chunk_forward: non-finite SOS coefficient");
        SosPlan *pl = get_plan(sos_host, K, stream, 1);
        if (prec == TFX_PREC_AUTO) prec = (plan_err_bound(pl) <= auto_bound()) ? TFX_PREC_F32 : TFX_PREC_F64;
        if (prec == TFX_PREC_F32) {
            p.tab = ensure_table<float>(pl, &pl->tab_f32_lc16, 16, &pl->nsteps16, stream);
        } else {
            p.tab = ensure_table<double>(pl, &pl->tab_f64_lc16, 16, &pl->nsteps16, stream);
        }
        p.nsteps = pl->nsteps16;
    }
    const int H = (int)Kf - 1, Hpad = (H + 3) & ~3, Kp = Hpad + 4;
    {   // device taps in the kernel's layout: Hpad - H leading zeros (they meet the zero-filled front of the LDS buffer),
        // the flipped taps, zeros up to Kp; cached by content like every other tap vector
        std::vector<float> lay((size_t)Kp, 0.0f);
        memcpy(lay.data() + (Hpad - H), taps_host, (size_t)Kf * 4);
        q.taps = (const float *)cached_taps(lay.data(), (size_t)Kp * 4, (size_t)Kp * 4);
    }
    q.hist_in = Kf > 1 ? hist_in : nullptr; q.hist_out = Kf > 1 ? hist_out : nullptr; q.y = y; q.Kf = (int)Kf;
    q.Tpad = (int)((T + 3) & ~3);
    q.gain = (float)gain; q.scale = scale; q.clamp = clamp;
    const bool vec = (((uintptr_t)x & 15) == 0) && (T % 4 == 0) && (x_pitch % 4 == 0);
    const bool f32 = prec == TFX_PREC_F32;
    const size_t stage_b = K > 0 ? (size_t)(sos_stage_bytes<float, float, double, 16>() + (f32 ? sos_carry_bytes<float, 16>(1, (int)K) : sos_carry_bytes<double, 16>(1, (int)K))) : 0;
    const size_t shmem = (size_t)(Hpad + q.Tpad + 8 + Kp) * 4 + stage_b;
    TFX_CHECK(shmem <= 64 * 1024, "chunk_forward: needs %zu B of LDS", shmem);
    // the cascade is one wavefront's work; the FIR phase scales with the threads: 4 outputs each
    const int threads = T * Kf <= (1 << 14) ? 256 : 1024;
    ProfScope ps("chunk_iir_fir_kernel", stream);
    if (f32) {
        if (vec) hipLaunchKernelGGL((chunk_iir_fir_kernel<float, true>), dim3((unsigned)C), dim3(threads), shmem, stream, q);
        else hipLaunchKernelGGL((chunk_iir_fir_kernel<float, false>), dim3((unsigned)C), dim3(threads), shmem, stream, q);
    } else {
        if (vec) hipLaunchKernelGGL((chunk_iir_fir_kernel<double, true>), dim3((unsigned)C), dim3(threads), shmem, stream, q);
        else hipLaunchKernelGGL((chunk_iir_fir_kernel<double, false>), dim3((unsigned)C), dim3(threads), shmem, stream, q);
    }
    TFX_HIP(hipGetLastError());
}

// olsnative.hip (cascade inside the forward column pass): the warm-up for max|A^W| < 2^-bits and the unit-b0 form of a
// cascade -- rows [G_s = b0_0 ... b0_s, b1 / b0, b2 / b0, -a1, -a2] (quotients in long double) -- or false when the cascade
// has no such form (unit_form_ok)
int64_t sos_warmup_bits(const double *sos_host, int64_t K, int bits)
{
    const std::vector<double> sos(sos_host, sos_host + 6 * K);
    return warmup_length(sos, (int)K, bits);
}
bool sos_unit_rows(const double *sos_host, int64_t K, double (*rows)[5])
{
    const std::vector<double> sos(sos_host, sos_host + 6 * K);
    if (!unit_form_ok(sos, (int)K)) return false;
    ld g = 1.0L;
    for (int64_t s = 0; s < K; ++s) {
        const ld b0 = sos[s * 6];
        g *= b0;
        rows[s][0] = (double)g;
        rows[s][1] = (double)((ld)sos[s * 6 + 1] / b0);
        rows[s][2] = (double)((ld)sos[s * 6 + 2] / b0);
        rows[s][3] = -sos[s * 6 + 4];
        rows[s][4] = -sos[s * 6 + 5];
    }
    return true;
}

void sos_plan_info(const double *sos_host, int64_t K, int *precision, int64_t *warmup, double *err_bound)
{
    SosPlan *pl = get_plan(sos_host, K, nullptr, 1);
    const double eb = plan_err_bound(pl);
    if (precision) *precision = (eb <= auto_bound()) ? TFX_PREC_F32 : TFX_PREC_F64;
    if (warmup) *warmup = pl->warm;
    if (err_bound) *err_bound = eb;
}

}  // namespace tfx
