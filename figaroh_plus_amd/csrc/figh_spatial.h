// 3-vector / rotation helpers of the regressor kernels (fp64, 6-vectors are (linear, angular)).
#pragma once

#include <hip/hip_runtime.h>

namespace figh {

__device__ __forceinline__ void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void rot(const double *R, const double *x, double *y) {  // y = R x
    y[0] = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
    y[1] = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
    y[2] = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
}
__device__ __forceinline__ void rotT(const double *R, const double *x, double *y) {  // y = R^T x
    y[0] = R[0] * x[0] + R[3] * x[1] + R[6] * x[2];
    y[1] = R[1] * x[0] + R[4] * x[1] + R[7] * x[2];
    y[2] = R[2] * x[0] + R[5] * x[1] + R[8] * x[2];
}
__device__ __forceinline__ void rodrigues(const double *a, double c, double s, double *R) {
    const double t = 1.0 - c;
    R[0] = 1.0 - t * (a[2] * a[2] + a[1] * a[1]);
    R[1] = t * a[0] * a[1] - s * a[2];
    R[2] = t * a[0] * a[2] + s * a[1];
    R[3] = t * a[0] * a[1] + s * a[2];
    R[4] = 1.0 - t * (a[2] * a[2] + a[0] * a[0]);
    R[5] = t * a[1] * a[2] - s * a[0];
    R[6] = t * a[0] * a[2] - s * a[1];
    R[7] = t * a[1] * a[2] + s * a[0];
    R[8] = 1.0 - t * (a[1] * a[1] + a[0] * a[0]);
}
__device__ __forceinline__ void matmul3(const double *A, const double *B, double *C) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
__device__ __forceinline__ double sgn(double x) { return (double)((x > 0.0) - (x < 0.0)); }

// J^T B for the body regressor B = bodyRegressor(v, a) of one link and a motion axis J = (Jl, Ja) expressed in the
// link frame, in closed form -- the ten entries of one row of pin.computeJointTorqueRegressor for that link, in
// FIGAROH's column order [Ixx Ixy Ixz Iyy Iyz Izz mx my mz m] (regressor.py:73-82).  acc = a_lin + w x v_lin,
// dw = a_ang, w = v_ang.  Equal to propagating the 6 x 10 body regressor up the chain and projecting it on the joint
// axis (validated to 1.6e-16 relative), with 5x fewer flops and no per-thread arrays.
__device__ __forceinline__ void axis_times_body_regressor(const double *Jl, const double *Ja, const double *acc,
                                                          const double *dw, const double *w, double *o) {
    o[9] = Jl[0] * acc[0] + Jl[1] * acc[1] + Jl[2] * acc[2];  // m
    double h1[3], h2[3], h3[3], h4[3];  // mx my mz: Jl x dw + w x (w x Jl) + acc x Ja
    cross3(Jl, dw, h1);
    cross3(w, Jl, h2);
    cross3(w, h2, h3);
    cross3(acc, Ja, h4);
    o[6] = h1[0] + h3[0] + h4[0];
    o[7] = h1[1] + h3[1] + h4[1];
    o[8] = h1[2] + h3[2] + h4[2];
    // inertia: L(dw)^T Ja - L(w)^T (w x Ja),  L(x)^T y = [x0y0, x1y0+x0y1, x1y1, x2y0+x0y2, x2y1+x1y2, x2y2]
    double u[3];
    cross3(w, Ja, u);
    o[0] = dw[0] * Ja[0] - w[0] * u[0];                                              // Ixx
    o[1] = dw[1] * Ja[0] + dw[0] * Ja[1] - (w[1] * u[0] + w[0] * u[1]);                // Ixy
    o[3] = dw[1] * Ja[1] - w[1] * u[1];                                              // Iyy
    o[2] = dw[2] * Ja[0] + dw[0] * Ja[2] - (w[2] * u[0] + w[0] * u[2]);                // Ixz
    o[4] = dw[2] * Ja[1] + dw[1] * Ja[2] - (w[2] * u[1] + w[1] * u[2]);                // Iyz
    o[5] = dw[2] * Ja[2] - w[2] * u[2];                                              // Izz
}

// ... for a pure translation axis (Ja = 0: the three force rows of the external-wrench regressor): the six rotational-inertia
// entries are exact zeros and mx my mz lose their acc x Ja term -- 40 instead of 110 flops.  o[0 .. 5] are not written.
__device__ __forceinline__ void axis_times_body_regressor_lin(const double *Jl, const double *acc, const double *dw,
                                                              const double *w, double *o) {
    o[9] = Jl[0] * acc[0] + Jl[1] * acc[1] + Jl[2] * acc[2];  // m
    double h1[3], h2[3], h3[3];                                 // mx my mz: Jl x dw + w x (w x Jl)
    cross3(Jl, dw, h1);
    cross3(w, Jl, h2);
    cross3(w, h2, h3);
    o[6] = h1[0] + h3[0];
    o[7] = h1[1] + h3[1];
    o[8] = h1[2] + h3[2];
}

}  // namespace figh
