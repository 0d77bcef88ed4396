/* Plain-C host of libsed_hip.so: no Python, no torch, no C++ -- only include/sed_hip.h and the HIP runtime's C API.
 * Runs one ConvBlock convolution (nn.Conv2d 3x3, stride 1, pad 1, no bias: reference pytorch/models.py:77-85) through the
 * 2-D Winograd MFMA kernel and through the direct implicit-GEMM kernel and checks both against a naive CPU loop.
 *
 *   gcc -O2 examples/capi_conv.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       -Lsound_event_detection_dcase2017_task4_amd -lsed_hip -L/opt/rocm/lib -lamdhip64 -lm -o capi_conv
 *   (sound_event_detection_dcase2017_task4_amd/build.py builds it as build/capi_conv)
 * ... and one dense layer (y = x w^T + b, dw = gy^T x) through the split-f16 GEMMs of the GRU / MultiHead heads.
 * Exit code 0 = every kernel within tolerance of the CPU result. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "sed_hip.h"

#define CHECK_HIP(e)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (e);                                                                 \
        if (e__ != hipSuccess) { fprintf(stderr, "HIP error %d at line %d\n", (int)e__, __LINE__); return 2; } \
    } while (0)
#define CHECK_SED(e)                                                                          \
    do {                                                                                      \
        int e__ = (e);                                                                        \
        if (e__ != 0) { fprintf(stderr, "sed error %d at line %d\n", e__, __LINE__); return 3; } \
    } while (0)

static float frand(unsigned* s) {
    *s = *s * 1664525u + 1013904223u;
    return ((float)(*s >> 8) / 8388608.0f) - 1.0f;       /* [-1, 1) */
}

int main(void) {
    const int B = 2, H = 9, W = 16, Cin = 32, Cout = 64;  /* NHWC activations, OIHW weights */
    const long M = (long)B * H * W;
    unsigned seed = 1234u;
    float* x = (float*)malloc(sizeof(float) * M * Cin);
    float* w = (float*)malloc(sizeof(float) * Cout * Cin * 9);
    float* ref = (float*)malloc(sizeof(float) * M * Cout);
    float* got = (float*)malloc(sizeof(float) * M * Cout);
    for (long i = 0; i < M * Cin; ++i) x[i] = frand(&seed);
    for (long i = 0; i < (long)Cout * Cin * 9; ++i) w[i] = 0.1f * frand(&seed);
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int wi = 0; wi < W; ++wi)
                for (int co = 0; co < Cout; ++co) {
                    double acc = 0.0;
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const int hh = h + ky - 1, ww = wi + kx - 1;
                            if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
                            const float* xp = x + (((long)b * H + hh) * W + ww) * Cin;
                            for (int ci = 0; ci < Cin; ++ci) acc += (double)xp[ci] * w[((co * Cin + ci) * 3 + ky) * 3 + kx];
                        }
                    ref[(((long)b * H + h) * W + wi) * Cout + co] = (float)acc;
                }

    printf("%s\n", sed_version());
    float *dx, *dw, *dy, *dwf, *duf, *dwscale, *dxamax;
    void* dwp;
    CHECK_HIP(hipMalloc((void**)&dx, sizeof(float) * M * Cin));
    CHECK_HIP(hipMalloc((void**)&dw, sizeof(float) * Cout * Cin * 9));
    CHECK_HIP(hipMalloc((void**)&dy, sizeof(float) * M * Cout));
    CHECK_HIP(hipMalloc((void**)&dwf, sizeof(float) * 9 * Cout * Cin));
    CHECK_HIP(hipMalloc((void**)&duf, sizeof(float) * 16 * Cout * Cin));
    CHECK_HIP(hipMalloc(&dwp, 2 * (size_t)sed_conv_sf16_pack_halfs(Cin, Cout)));   /* f16 (hi, lo) planes */
    CHECK_HIP(hipMalloc((void**)&dwscale, sizeof(float) * (sed_amax_slots() + 1)));   /* amax slots + scale */
    CHECK_HIP(hipMalloc((void**)&dxamax, sizeof(float) * sed_amax_slots()));
    CHECK_HIP(hipMemcpy(dx, x, sizeof(float) * M * Cin, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dw, w, sizeof(float) * Cout * Cin * 9, hipMemcpyHostToDevice));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));

    int rc = 0;
    for (int pass = 0; pass < 3; ++pass) {
        CHECK_HIP(hipMemsetAsync(dy, 0, sizeof(float) * M * Cout, stream));
        if (pass == 0) {
            if (!sed_conv3x3_wino2_supported(H, W, Cin, Cout)) { fprintf(stderr, "wino2 not supported?\n"); return 4; }
            CHECK_SED(sed_pack_conv_weights_wino2(dw, Cout, Cin, duf, NULL, (sed_stream_t)stream));
            CHECK_SED(sed_conv3x3_wino2(dx, duf, dy, B, H, W, Cin, Cout, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL,
                                        (sed_stream_t)stream));
        } else if (pass == 2) {              /* the default path of the models: split-f16 operands on the f16 MFMA pipe */
            if (!sed_conv3x3_sf16_supported(H, W, Cin, Cout)) { fprintf(stderr, "sf16 not supported?\n"); return 4; }
            CHECK_SED(sed_pack_conv_weights_sf16(dw, Cout, Cin, 0, dwscale, dwp, (sed_stream_t)stream));
            CHECK_SED(sed_amax(dx, M * Cin, dxamax, (sed_stream_t)stream));      /* operand scale: taken on the device */
            CHECK_SED(sed_conv3x3_sf16(dx, dwp, dwscale, dy, B, H, W, Cin, Cout, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL,
                                       dxamax, NULL, NULL, NULL, 0, NULL, (sed_stream_t)stream));
        } else {
            CHECK_SED(sed_pack_conv_weights(dw, Cout, Cin, dwf, NULL, (sed_stream_t)stream));
            CHECK_SED(sed_conv3x3_igemm(dx, dwf, dy, B, H, W, Cin, Cout, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL,
                                        (sed_stream_t)stream));
        }
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_HIP(hipMemcpy(got, dy, sizeof(float) * M * Cout, hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (long i = 0; i < M * Cout; ++i) {
            const double e = fabs((double)got[i] - (double)ref[i]);
            if (e > worst) worst = e;
        }
        printf("%s: max |gpu - cpu| = %.3g\n", pass == 0 ? "sed_conv3x3_wino2" : (pass == 1 ? "sed_conv3x3_igemm" : "sed_conv3x3_sf16"), worst);
        if (!(worst < 1e-4)) rc = 1;
    }
    /* ---- the dense layers of the GRU / MultiHead heads (nn.GRU input projection, reference models.py:529-530): y = x w^T + b and
     * dw = gy^T x on the split-f16 GEMMs, from the same plain-C host */
    {
        const int GM = 300, GN = 128, GK = 128;          /* ragged M on purpose */
        float* gx = (float*)malloc(sizeof(float) * GM * GK);
        float* gw = (float*)malloc(sizeof(float) * GN * GK);
        float* gb = (float*)malloc(sizeof(float) * GN);
        float* gyh = (float*)malloc(sizeof(float) * GM * GN);
        float* gdw = (float*)malloc(sizeof(float) * GN * GK);
        for (int i = 0; i < GM * GK; ++i) gx[i] = frand(&seed);
        for (int i = 0; i < GN * GK; ++i) gw[i] = 0.1f * frand(&seed);
        for (int i = 0; i < GN; ++i) gb[i] = frand(&seed);
        float *dgx, *dgw, *dgb, *dgy, *dgdw, *dgws, *dgxa, *dgya, *dgpart;
        void* dgwp;
        CHECK_HIP(hipMalloc((void**)&dgx, sizeof(float) * GM * GK));
        CHECK_HIP(hipMalloc((void**)&dgw, sizeof(float) * GN * GK));
        CHECK_HIP(hipMalloc((void**)&dgb, sizeof(float) * GN));
        CHECK_HIP(hipMalloc((void**)&dgy, sizeof(float) * GM * GN));
        CHECK_HIP(hipMalloc((void**)&dgdw, sizeof(float) * GN * GK));
        CHECK_HIP(hipMalloc(&dgwp, 2 * (size_t)sed_gemm_pack_sf16_halfs(GN, GK)));
        CHECK_HIP(hipMalloc((void**)&dgws, sizeof(float) * (sed_amax_slots() + 1)));
        CHECK_HIP(hipMalloc((void**)&dgxa, sizeof(float) * sed_amax_slots()));
        CHECK_HIP(hipMalloc((void**)&dgya, sizeof(float) * sed_amax_slots()));
        CHECK_HIP(hipMalloc((void**)&dgpart, sizeof(float) * (size_t)sed_gemm_tn_sf16_partial_floats(GM, GN, GK)));
        CHECK_HIP(hipMemcpy(dgx, gx, sizeof(float) * GM * GK, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(dgw, gw, sizeof(float) * GN * GK, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(dgb, gb, sizeof(float) * GN, hipMemcpyHostToDevice));
        if (!sed_gemm_nt_sf16_supported(GM, GN, GK) || !sed_gemm_tn_sf16_supported(GM, GN, GK)) { fprintf(stderr, "sf16 GEMM not supported?\n"); return 4; }
        CHECK_SED(sed_gemm_pack_sf16(dgw, GN, GK, dgws, dgwp, (sed_stream_t)stream));
        CHECK_SED(sed_amax(dgx, (long)GM * GK, dgxa, (sed_stream_t)stream));
        CHECK_SED(sed_gemm_nt_sf16(dgx, dgwp, dgws, dgb, dgy, GM, GN, GK, dgxa, NULL, NULL, dgya /* amax of y, for the next call */,
                                   (sed_stream_t)stream));
        CHECK_SED(sed_gemm_tn_sf16(dgx, dgy, dgdw, dgpart, GM, GN, GK, dgxa, dgya, NULL, NULL, (sed_stream_t)stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_HIP(hipMemcpy(gyh, dgy, sizeof(float) * GM * GN, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(gdw, dgdw, sizeof(float) * GN * GK, hipMemcpyDeviceToHost));
        double worst = 0.0, worst_dw = 0.0;
        for (int m = 0; m < GM; ++m)
            for (int n = 0; n < GN; ++n) {
                double acc = gb[n];
                for (int k = 0; k < GK; ++k) acc += (double)gx[m * GK + k] * gw[n * GK + k];
                const double e = fabs(acc - (double)gyh[m * GN + n]);
                if (e > worst) worst = e;
            }
        for (int n = 0; n < GN; ++n)
            for (int k = 0; k < GK; ++k) {
                double acc = 0.0;
                for (int m = 0; m < GM; ++m) acc += (double)gyh[m * GN + n] * gx[m * GK + k];
                const double e = fabs(acc - (double)gdw[n * GK + k]);
                if (e > worst_dw) worst_dw = e;
            }
        printf("sed_gemm_nt_sf16: max |gpu - cpu| = %.3g\nsed_gemm_tn_sf16: max |gpu - cpu| = %.3g\n", worst, worst_dw);
        if (!(worst < 1e-4) || !(worst_dw < 1e-3)) rc = 1;
        hipFree(dgx); hipFree(dgw); hipFree(dgb); hipFree(dgy); hipFree(dgdw); hipFree(dgwp); hipFree(dgws); hipFree(dgxa);
        hipFree(dgya); hipFree(dgpart);
        free(gx); free(gw); free(gb); free(gyh); free(gdw);
    }
    /* bad arguments are reported, not executed */
    if (sed_conv3x3_wino2(dx, duf, dy, B, H, 7, Cin, Cout, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL,
                          (sed_stream_t)stream) != -22) { fprintf(stderr, "expected -22 for W=7\n"); rc = 1; }
    hipFree(dx); hipFree(dw); hipFree(dy); hipFree(dwf); hipFree(duf); hipFree(dwp); hipFree(dwscale);
    hipStreamDestroy(stream);
    free(x); free(w); free(ref); free(got);
    printf(rc == 0 ? "capi_conv ok\n" : "capi_conv FAILED\n");
    return rc;
}
