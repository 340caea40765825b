/* A non-torch host of the C ABI (include/torchfx_hip.h), plain C99: what a cgo / JNI / FFI binding would call.
 *
 *   gcc -std=c99 -I include examples/c_host.c -L torchfx_amd -ltorchfx_hip -Wl,-rpath,$PWD/torchfx_amd -o c_host
 *
 * Without arguments it only uses the host-side entry points (plan queries; runs on a machine without a GPU --
 * tests/test_capi_exports.py builds and runs it that way).  With `--gpu` it also filters a buffer on device 0
 * through tfx_sos_forward using the HIP runtime for memory (link with -lamdhip64 and define WITH_HIP).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "torchfx_hip.h"

#ifdef WITH_HIP
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#endif

/* A host that runs its own device allocator hands it to the library (tfx_set_workspace_allocator): every overlap-save workspace then
 * comes from -- and goes back to -- these two functions.  Here: the HIP runtime's own calls behind a byte counter. */
static long long ws_live = 0, ws_calls = 0;
static void *ws_alloc(size_t bytes, int device, void *stream, void *ctx)
{
    void *p = NULL;
    (void)device; (void)stream; (void)ctx;
    ++ws_calls;
#ifdef WITH_HIP
    if (hipMalloc(&p, bytes) != hipSuccess) return NULL;       /* NULL = "cannot": the pipeline asks again for a smaller slab */
    ws_live += (long long)bytes;
#else
    (void)bytes;
#endif
    return p;
}
static void ws_free(void *ptr, int device, void *ctx)
{
    (void)device; (void)ctx;
#ifdef WITH_HIP
    (void)hipFree(ptr);
#else
    (void)ptr;
#endif
}

/* 2nd-order Butterworth low-pass at fc/fs = 1/12 and a peaking section, [K, 6] rows b0 b1 b2 1 a1 a2 */
static const double SOS[2][6] = {
    {0.0495329964, 0.0990659928, 0.0495329964, 1.0, -1.2796324250, 0.4777644106},
    {1.0089, -1.9636, 0.9695, 1.0, -1.9636, 0.9784},
};

int main(int argc, char **argv)
{
    int precision = -1, native = -1;
    int64_t warm = 0, N = 0, S = 0, F = 0;
    double bound = 0.0;
    printf("tfx_version %d\n", tfx_version());
    if (tfx_sos_plan_info(&SOS[0][0], 2, &precision, &warm, &bound) != 0) {
        fprintf(stderr, "tfx_sos_plan_info: %s\n", tfx_last_error());
        return 1;
    }
    printf("cascade plan: auto precision %s, warm-up halo %lld samples, float32 error estimate %.3g\n",
           precision == TFX_PREC_F32 ? "f32" : "f64", (long long)warm, bound);
    if (tfx_ols_plan_info(65536, 28800000, 65535, 0, &N, &S, &F, &native) != 0) {
        fprintf(stderr, "tfx_ols_plan_info: %s\n", tfx_last_error());
        return 1;
    }
    printf("overlap-save plan: N %lld hop %lld blocks/row %lld native %d\n", (long long)N, (long long)S, (long long)F, native);
    {   /* the fused `cascade | FIR` step: host-only planning queries */
        int64_t fn = 0, fs = 0, ff = 0, fw = 0;
        const int ok = tfx_sos_fft_conv_plan_info(28800000, &SOS[0][0], 2, 66559, 66558, 0, 0, &fn, &fs, &ff, &fw);
        printf("fused cascade|FIR plan: served %d block %lld hop %lld frames/row %lld warm-up %lld; 28800001-sample rows served %d\n", ok,
               (long long)fn, (long long)fs, (long long)ff, (long long)fw,
               tfx_sos_fft_conv_supported(28800001, &SOS[0][0], 2, 66559, 66558, 0, 0));
    }
    /* workspaces under the host's control: install the pair above (host-only call; nothing is allocated before a kernel needs it) */
    if (tfx_set_workspace_allocator(ws_alloc, ws_free, NULL) != 0 || tfx_set_workspace_allocator(ws_alloc, NULL, NULL) == 0) {
        fprintf(stderr, "tfx_set_workspace_allocator: %s\n", tfx_last_error());
        return 1;
    }
    printf("workspace allocator installed; held now %lld bytes (%lld calls, %lld bytes through the host's allocator so far)\n",
           (long long)tfx_workspace_bytes(), ws_calls, ws_live);
    /* error path: a null coefficient pointer is an error code, not a crash */
    if (tfx_sos_plan_info(NULL, 2, &precision, &warm, &bound) == 0) {
        fprintf(stderr, "expected an error for a null SOS pointer\n");
        return 1;
    }
    printf("null pointer -> \"%s\"\n", tfx_last_error());
#ifdef WITH_HIP
    if (argc > 1 && strcmp(argv[1], "--gpu") == 0) {
        const int64_t C = 2, T = 48000;
        float *hx = (float *)malloc(sizeof(float) * C * T), *hy = (float *)malloc(sizeof(float) * C * T);
        float *dx = NULL, *dy = NULL;
        for (int64_t i = 0; i < C * T; ++i) hx[i] = (i % T == 0) ? 1.0f : 0.0f;      /* unit impulses */
        if (hipMalloc((void **)&dx, sizeof(float) * C * T) != hipSuccess || hipMalloc((void **)&dy, sizeof(float) * C * T) != hipSuccess) return 2;
        hipMemcpy(dx, hx, sizeof(float) * C * T, hipMemcpyHostToDevice);
        if (tfx_sos_forward(dx, TFX_F32, dy, TFX_F32, C, T, &SOS[0][0], 2, NULL, NULL, NULL, NULL, NULL, TFX_PREC_F64, NULL) != 0) {
            fprintf(stderr, "tfx_sos_forward: %s\n", tfx_last_error());
            return 1;
        }
        hipDeviceSynchronize();
        hipMemcpy(hy, dy, sizeof(float) * C * T, hipMemcpyDeviceToHost);
        printf("impulse response head: %.6f %.6f %.6f %.6f\n", hy[0], hy[1], hy[2], hy[3]);
        {   /* the same cascade inside the overlap-save pipeline (tfx_sos_fft_conv_forward) with a 9001-tap FIR whose only non-zero
             * tap is the LAST of the flipped kernel, i.e. h[0] = 1: the output must equal the cascade's */
            const int64_t Kf = 9001;
            float *taps = (float *)calloc((size_t)Kf, sizeof(float)), *hz = (float *)malloc(sizeof(float) * C * T);
            float *dz = NULL;
            double worst = 0.0;
            taps[Kf - 1] = 1.0f;
            if (hipMalloc((void **)&dz, sizeof(float) * C * T) != hipSuccess) return 2;
            if (!tfx_sos_fft_conv_supported(T, &SOS[0][0], 2, Kf, Kf - 1, 0, 1)) { fprintf(stderr, "fused step not served\n"); return 1; }
            if (tfx_sos_fft_conv_forward(dx, dz, C, T, &SOS[0][0], 2, taps, Kf, Kf - 1, 0, NULL, 1, NULL, NULL) != 0) {
                fprintf(stderr, "tfx_sos_fft_conv_forward: %s\n", tfx_last_error());
                return 1;
            }
            hipDeviceSynchronize();
            hipMemcpy(hz, dz, sizeof(float) * C * T, hipMemcpyDeviceToHost);
            for (int64_t i = 0; i < C * T; ++i) { const double d = fabs((double)hz[i] - (double)hy[i]); if (d > worst) worst = d; }
            printf("fused cascade|identity FIR vs cascade: max difference %.3g\n", worst);
            hipFree(dz); free(taps); free(hz);
        }
        /* the pipeline's workspaces came through ws_alloc; tfx_clear_caches hands them back through ws_free */
        printf("workspaces through the host's allocator: %lld calls, %lld bytes; held %lld\n", ws_calls, ws_live, (long long)tfx_workspace_bytes());
        tfx_clear_caches();
        printf("after tfx_clear_caches: held %lld\n", (long long)tfx_workspace_bytes());
        hipFree(dx); hipFree(dy); free(hx); free(hy);
    }
#else
    (void)argc; (void)argv;
#endif
    return 0;
}
