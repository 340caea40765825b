// epilogue.h -- what a producing kernel (the SOS cascade, the inverse column pass of the overlap-save
// convolution) can do to every sample it stores, so that a `filter | Gain | Normalize` pipeline
// (src/torchfx/effect.py:361-383, 696-698, 719-721, 775-786) does not pay a streaming pass per effect:
//   * y *= gain and clip to [-1, 1], evaluated in the output dtype on the value the standalone Gain
//     pass would have read (bit-identical to running `Gain` afterwards);
//   * a partial of max|y| or sum y^2 over the samples the workgroup stores, reduced per row or globally by
//     reduce_finish (effects.hip) into the statistic `Normalize` divides by -- Normalize then costs one
//     apply pass instead of a reduction pass plus an apply pass.
#pragma once
#include "common.h"

namespace tfx {

struct Epilogue {
    double gain = 1.0;
    int scale = 0;          // multiply by gain
    int clamp = 0;          // clip to [-1, 1] (after the gain)
    int stat_mode = -1;     // -1 none, 0 max|y|, 1 sum of y^2
    int per_row = 0;        // statistic per output row, else one value for the whole tensor
    double *stat_out = nullptr;   // device [rows or 1] float64, raw statistic
    bool any() const { return scale || clamp || stat_mode >= 0; }
};

template <typename T> __device__ __forceinline__ T clamp_unit(T v)
{
    return v < (T)-1 ? (T)-1 : (v > (T)1 ? (T)1 : v);      // NaN stays NaN, like torch.clamp
}

// MODE 0  max|x| -- carried as the BIT PATTERN of a non-negative double: such patterns order like unsigned
//                   integers and NaN sorts above inf, so a NaN anywhere wins, like torch.max
// MODE 1  sum x^2 -- float64 accumulation
template <int MODE> __device__ __forceinline__ double red_init() { return 0.0; }
template <int MODE> __device__ __forceinline__ double red_elem(double v)
{
    if (MODE == 0) return __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, v) & 0x7fffffffffffffffull);
    return v * v;
}
template <int MODE> __device__ __forceinline__ double red_comb(double a, double b)
{
    if (MODE == 0) {
        const unsigned long long x = __builtin_bit_cast(unsigned long long, a), y = __builtin_bit_cast(unsigned long long, b);
        return __builtin_bit_cast(double, x > y ? x : y);
    }
    return a + b;
}
__device__ __forceinline__ double red_elem_rt(int mode, double v) { return mode == 0 ? red_elem<0>(v) : red_elem<1>(v); }
__device__ __forceinline__ double red_comb_rt(int mode, double a, double b) { return mode == 0 ? red_comb<0>(a, b) : red_comb<1>(a, b); }

// effects.hip: second stage of the statistics (one workgroup per row, fixed order), and the passes a
// producer without a fused epilogue falls back to
void stat_finish(const double *partial, int64_t rows, int64_t groups, int mode, double *stat_dev, hipStream_t stream);
void epilogue_as_passes(void *y, int dtype, int64_t C, int64_t T, const Epilogue &ep, hipStream_t stream);

}  // namespace tfx
