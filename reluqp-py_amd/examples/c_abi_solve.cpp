// c_abi_solve.cpp -- the C ABI (include/rqp_abi.h) from a host program with no Python and no torch: raw device pointers
// and sizes only.  Solves a batch of three 2-variable QPs that share their constraints,
//
//     minimize  1/2 x' H x + g_b' x     subject to   x1 + x2 = 1,   0 <= x1 <= 0.7,   0 <= x2
//
// with H = diag(2, 2) and g_0 = (-2, -5), g_1 = (-5, -2), g_2 = (-3, -3); the minimisers are (0, 1), (0.7, 0.3) and
// (0.5, 0.5) with objectives -4, -3.52 and -2.5 (substitute x2 = 1 - x1 and clip).  Prints one line per instance and
// exits non-zero on any ABI error or a wrong answer.
//
//     hipcc --offload-arch=gfx950 -I include reluqp-py_amd/examples/c_abi_solve.cpp \
//           -L reluqp-py_amd/reluqp/lib -lrqp_hip -Wl,-rpath,$PWD/reluqp-py_amd/reluqp/lib -o c_abi_solve
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rqp_abi.h"

#define HIP_OK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define RQP_OK_(h, call)                                                                      \
    do {                                                                                      \
        int rc_ = (call);                                                                     \
        if (rc_ != RQP_OK) { fprintf(stderr, "%s: %s (%s)\n", #call, rqp_strerror(rc_), (h) ? rqp_last_error(h) : ""); return 3; } \
    } while (0)

int main(void) {
    enum { B = 3, N = 2, M = 3 };
    const double H[N * N] = {2, 0, 0, 2};                                   // shared by the batch
    const double A[M * N] = {1, 1, 1, 0, 0, 1};                            // rows: x1 + x2, x1, x2
    const double g[B * N] = {-2, -5, -5, -2, -3, -3};
    double l[B * M], u[B * M];
    for (int b = 0; b < B; ++b) {
        l[b * M + 0] = 1.0; u[b * M + 0] = 1.0;                            // equality row (u - l <= eq_tol)
        l[b * M + 1] = 0.0; u[b * M + 1] = 0.7;
        l[b * M + 2] = 0.0; u[b * M + 2] = INFINITY;
    }
    const double want_x[B][N] = {{0.0, 1.0}, {0.7, 0.3}, {0.5, 0.5}}, want_obj[B] = {-4.0, -3.52, -2.5};

    double *dH, *dA, *dg, *dl, *du, *dx, *dz, *dy, *dobj, *dpri, *ddua;
    int32_t *diter, *dstatus;
    HIP_OK(hipMalloc((void**)&dH, sizeof(H)));   HIP_OK(hipMemcpy(dH, H, sizeof(H), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&dA, sizeof(A)));   HIP_OK(hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&dg, sizeof(g)));   HIP_OK(hipMemcpy(dg, g, sizeof(g), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&dl, sizeof(l)));   HIP_OK(hipMemcpy(dl, l, sizeof(l), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&du, sizeof(u)));   HIP_OK(hipMemcpy(du, u, sizeof(u), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&dx, B * N * sizeof(double)));
    HIP_OK(hipMalloc((void**)&dz, B * M * sizeof(double)));
    HIP_OK(hipMalloc((void**)&dy, B * M * sizeof(double)));
    HIP_OK(hipMalloc((void**)&dobj, B * sizeof(double)));
    HIP_OK(hipMalloc((void**)&dpri, B * sizeof(double)));
    HIP_OK(hipMalloc((void**)&ddua, B * sizeof(double)));
    HIP_OK(hipMalloc((void**)&diter, B * sizeof(int32_t)));
    HIP_OK(hipMalloc((void**)&dstatus, B * sizeof(int32_t)));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    rqp_dims dims = {N, M, B, /*shared_mats=*/1, RQP_F64, RQP_KERNEL_AUTO, RQP_TILE_SAME, /*flags=*/0};
    rqp_settings st;
    RQP_OK_(NULL, rqp_default_settings(&st));                              // the reference's defaults (classes.py:32-65)
    st.eps_abs = 1e-6;
    rqp_handle* h = NULL;
    RQP_OK_(NULL, rqp_create(&h, &dims, &st, /*device=*/0));
    RQP_OK_(h, rqp_setup(h, dH, dg, dA, dl, du, stream));
    rqp_info info = {0};
    info.iter = diter; info.status = dstatus; info.obj_val = dobj; info.pri_res = dpri; info.dua_res = ddua;
    RQP_OK_(h, rqp_solve(h, dx, dz, dy, &info, stream));
    HIP_OK(hipStreamSynchronize(stream));

    double x[B * N], obj[B], pri[B], dua[B];
    int32_t iter[B], status[B];
    HIP_OK(hipMemcpy(x, dx, sizeof(x), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(obj, dobj, sizeof(obj), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(pri, dpri, sizeof(pri), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(dua, ddua, sizeof(dua), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(iter, diter, sizeof(iter), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(status, dstatus, sizeof(status), hipMemcpyDeviceToHost));
    printf("%s kernel=%s\n", rqp_version(), rqp_kernel_name(h));
    int bad = 0;
    for (int b = 0; b < B; ++b) {
        printf("instance %d: status=%d iter=%d x=(%.6f, %.6f) obj=%.6f pri=%.2e dua=%.2e\n", b, status[b], iter[b], x[b * N],
               x[b * N + 1], obj[b], pri[b], dua[b]);
        if (status[b] != RQP_STATUS_SOLVED || fabs(x[b * N] - want_x[b][0]) > 1e-4 || fabs(x[b * N + 1] - want_x[b][1]) > 1e-4 ||
            fabs(obj[b] - want_obj[b]) > 1e-4)
            bad = 1;
    }
    RQP_OK_(h, rqp_destroy(h));
    void* bufs[] = {dH, dA, dg, dl, du, dx, dz, dy, dobj, dpri, ddua, diter, dstatus};
    for (size_t i = 0; i < sizeof(bufs) / sizeof(bufs[0]); ++i) (void)hipFree(bufs[i]);
    (void)hipStreamDestroy(stream);
    if (bad) { fprintf(stderr, "wrong answer\n"); return 1; }
    printf("ok\n");
    return 0;
}
