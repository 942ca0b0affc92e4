// dmath.h -- fp64 device math of the gfx950 DSAC engine: Rodrigues (both ways + derivative), pinhole
// projection, minimal-set P3P (Gao's quartic + triad alignment), pose-convention flips.
//
// Everything here runs on one lane in registers: all loops have compile-time bounds and every array
// index is static after unrolling, so nothing lands in scratch because of dynamic indexing.
// Semantics follow the routines the reference calls through OpenCV 2.4 (cv::Rodrigues, cv::projectPoints,
// cv::solvePnP(CV_P3P)) at core/cnn_softam.h:66,351,507-508,1042-1046 and core/types.h:186-214.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dm {

#define DM_INLINE __device__ __forceinline__

struct Cam {
    double fx, fy, cx, cy;
};

struct Pose6 {
    double v[6];  // rvec | tvec (mm)
};

DM_INLINE bool isnan_d(double x) { return x != x; }

// ------------------------------------------------------------------------------------------------
// Rodrigues: vector -> matrix (row-major R[9]); optional 3x9 derivative J[i*9+k] = dR_k / dr_i
// ------------------------------------------------------------------------------------------------
// Split in two so that a caller that already holds R(r) can add the derivative later without redoing sqrt / sincos / the division:
// rodrigues_R fills R and the intermediates, rodrigues_J turns the intermediates into the derivative.
// Both are branch-free (selects on the |r| < DBL_EPSILON case, which gives R = I and dR/dr = -[e_i]x exactly as OpenCV's special case).
struct RodAux { double s, c, c1, it, k4, ax, ay, az; };

DM_INLINE void rodrigues_R(const double r[3], double R[9], RodAux& a) {
    const double rx = r[0], ry = r[1], rz = r[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    const bool small = theta < 2.220446049250313e-16;
    double s, c;
    sincos(small ? 1.0 : theta, &s, &c);
    const double it = small ? 0.0 : 1.0 / theta;
    s = small ? 0.0 : s;
    c = small ? 1.0 : c;
    a.s = s; a.c = c; a.c1 = 1.0 - c; a.it = it;
    a.k4 = small ? 1.0 : s * it;  // sin(theta)/theta -> 1
    const double ax = rx * it, ay = ry * it, az = rz * it;
    a.ax = ax; a.ay = ay; a.az = az;
    const double aat[9] = {ax * ax, ax * ay, ax * az, ax * ay, ay * ay, ay * az, ax * az, ay * az, az * az};
    const double skew[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + a.c1 * aat[k] + s * skew[k];
}

DM_INLINE void rodrigues_J(const RodAux& q, double* J) {
    const double s = q.s, c = q.c, c1 = q.c1, it = q.it, ax = q.ax, ay = q.ay, az = q.az;
    const double aat[9] = {ax * ax, ax * ay, ax * az, ax * ay, ay * ay, ay * az, ax * az, ay * az, az * az};
    const double skew[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
    const double a[3] = {ax, ay, az};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double ai = a[i];
        const double k0 = -s * ai, k1 = (s - 2 * c1 * it) * ai, k2 = c1 * it, k3 = (c - s * it) * ai, k4 = q.k4;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int row = k / 3, col = k % 3;
            // d(a a^T)/d a_i  at (row, col):  delta(row,i) a_col + delta(col,i) a_row
            const double daat = ((row == i) ? a[col] : 0.0) + ((col == i) ? a[row] : 0.0);
            // d[a]x/d a_i at (row, col): -eps(row, col, i)
            double dsk = 0.0;
            if (row != col && row != i && col != i) dsk = (((col - row + 3) % 3) == 1) ? -1.0 : 1.0;
            J[i * 9 + k] = k0 * ((k % 4 == 0) ? 1.0 : 0.0) + k1 * aat[k] + k2 * daat + k3 * skew[k] + k4 * dsk;
        }
    }
}

template <bool WITH_J>
DM_INLINE void rodrigues_v2m(const double r[3], double R[9], double* J) {
    RodAux a;
    rodrigues_R(r, R, a);
    if (WITH_J) rodrigues_J(a, J);
}

// Rodrigues: matrix -> vector.  Inputs on this path are rotation matrices to rounding (outputs of P3P,
// of rodrigues_v2m, or their sign flips), so OpenCV's SVD re-orthonormalisation is the identity to
// ~1e-16 and is skipped.
DM_INLINE void rodrigues_m2v(const double R[9], double r[3]) {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 9; k++) bad |= !(R[k] > -100.0 && R[k] < 100.0);
    if (bad) { r[0] = r[1] = r[2] = 0; return; }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { rx = ry = rz = 0; }
        else {
            double t;
            t = (R[0] + 1) * 0.5; rx = sqrt(fmax(t, 0.));
            t = (R[4] + 1) * 0.5; ry = sqrt(fmax(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; rz = sqrt(fmax(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        const double vth = theta / (2 * s);
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
}

DM_INLINE double det3(const double A[9]) {
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

// jp::cv2our (core/types.h:186-214): R = Rodrigues(rvec) with rows 1,2 negated, t.y,t.z negated;
// if det < 0 negate everything; NaN translation -> 0.
DM_INLINE void cv2our(const double cv6[6], double R[9], double t[3]) {
    rodrigues_v2m<false>(cv6, R, nullptr);
    t[0] = cv6[3]; t[1] = -cv6[4]; t[2] = -cv6[5];
#pragma unroll
    for (int j = 3; j < 9; j++) R[j] = -R[j];
    if (det3(R) < 0) {
#pragma unroll
        for (int j = 0; j < 9; j++) R[j] = -R[j];
        t[0] = -t[0]; t[1] = -t[1]; t[2] = -t[2];
    }
    if (isnan_d(t[0]) || isnan_d(t[1]) || isnan_d(t[2])) { t[0] = t[1] = t[2] = 0; }
}

// getRodVecAndTrans(Hypothesis(cv2our(cv pose)))  (core/cnn_softam.h:121-122, Hypothesis.cpp:274-289)
DM_INLINE void cv_to_jp6(const double cv6[6], double jp6[6]) {
    double R[9], t[3];
    cv2our(cv6, R, t);
    rodrigues_m2v(R, jp6);
    jp6[3] = t[0]; jp6[4] = t[1]; jp6[5] = t[2];
}

// projectPoints for one point, double inside, float out (Point2f).
DM_INLINE void project_f(const double R[9], const double t[3], const Cam& K, float X, float Y, float Z, float& u, float& v) {
    const double Mx = X, My = Y, Mz = Z;
    const double Xc = R[0] * Mx + R[1] * My + R[2] * Mz + t[0];
    const double Yc = R[3] * Mx + R[4] * My + R[5] * Mz + t[1];
    const double Zc = R[6] * Mx + R[7] * My + R[8] * Mz + t[2];
    const double z = (Zc != 0.0) ? 1. / Zc : 1.;
    u = (float)(Xc * z * K.fx + K.cx);
    v = (float)(Yc * z * K.fy + K.cy);
}

// getDiffMap's residual for one cell (core/cnn_softam.h:351-358): float difference, double norm, clamp.
DM_INLINE float residual_f(const double R[9], const double t[3], const Cam& K, float X, float Y, float Z, float pu, float pv, double clampv) {
    float u, v;
    project_f(R, t, K, X, Y, Z, u, v);
    const float dx = pu - u, dy = pv - v;
    return (float)fmin(sqrt((double)dx * dx + (double)dy * dy), clampv);
}

// ------------------------------------------------------------------------------------------------
// real roots of polynomials of degree <= 4 (closed forms)
// The whole P3P solve (these, p3p_setup, p3p_eval_root) is compiled WITHOUT fused multiply-adds, in every file that includes it: it is a chain of
// cancellations that OpenCV and the oracle evaluate operation by operation, and a contracted build is a differently rounded solver (Makefile, K1_FLAGS).
// ------------------------------------------------------------------------------------------------
DM_INLINE int roots2(double a, double b, double c, double& x0, double& x1) {
#pragma clang fp contract(off)
    const double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    const double inv_2a = 0.5 / a;
    if (delta == 0) { x0 = x1 = -b * inv_2a; return 1; }
    const double sq = sqrt(delta);
    x0 = (-b + sq) * inv_2a;
    x1 = (-b - sq) * inv_2a;
    return 2;
}

DM_INLINE int roots3(double a, double b, double c, double d, double& x0, double& x1, double& x2) {
#pragma clang fp contract(off)
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            x0 = -d / c;
            return 1;
        }
        x2 = 0;
        return roots2(b, c, d, x0, x1);
    }
    const double inv_a = 1. / a;
    const double b_a = inv_a * b, b_a2 = b_a * b_a, c_a = inv_a * c, d_a = inv_a * d;
    const double Q = (3 * c_a - b_a2) / 9;
    const double Rr = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    const double Q3 = Q * Q * Q;
    const double D = Q3 + Rr * Rr;
    const double b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (Rr == 0) { x0 = x1 = x2 = -b_a_3; return 3; }
        x0 = pow(2 * Rr, 1 / 3.0) - b_a_3;
        return 1;
    }
    if (D <= 0) {
        const double theta = acos(Rr / sqrt(-Q3));
        const double sqrt_Q = sqrt(-Q);
        x0 = 2 * sqrt_Q * cos(theta / 3.0) - b_a_3;
        x1 = 2 * sqrt_Q * cos((theta + 2 * 3.14159265358979323846) / 3.0) - b_a_3;
        x2 = 2 * sqrt_Q * cos((theta + 4 * 3.14159265358979323846) / 3.0) - b_a_3;
        return 3;
    }
#ifdef DSAC_P3P_CBRT  // A/B only (profiles/r05_k1_cost.txt): OpenCV writes the cube root as pow(x, 1/3), and ocml's pow is closer to glibc's than its cbrt
    const double AD = cbrt(fabs(Rr) + sqrt(D)) * (Rr > 0 ? 1 : (Rr < 0 ? -1 : 0));
#else
    const double AD = pow(fabs(Rr) + sqrt(D), 1.0 / 3.0) * (Rr > 0 ? 1 : (Rr < 0 ? -1 : 0));
#endif
    const double BD = (AD == 0) ? 0 : -Q / AD;
    x0 = AD + BD - b_a_3;
    return 1;
}

DM_INLINE int roots4(double a, double b, double c, double d, double e, double x[4]) {
#pragma clang fp contract(off)
    if (a == 0) { x[3] = 0; return roots3(b, c, d, e, x[0], x[1], x[2]); }
    const double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    const double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0, r1, r2;
    const int n = roots3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r0, r1, r2);
    if (n == 0) return 0;
    const double R2 = 0.25 * b2 - c + r0;
    if (R2 < 0) return 0;
    const double Rr = sqrt(R2), inv_R = 1. / Rr;
    int nb = 0;
    double D2, E2;
    if (Rr < 10E-12) {
        const double temp = r0 * r0 - 4 * e;
        if (temp < 0) D2 = E2 = -1;
        else {
            const double st = sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * st;
            E2 = D2 - 4 * st;
        }
    } else {
        const double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    const double b_4 = 0.25 * b, R_2 = 0.5 * Rr;
    if (D2 >= 0) {
        const double D = sqrt(D2);
        nb = 2;
        x[0] = R_2 + 0.5 * D - b_4;
        x[1] = x[0] - D;
    }
    if (E2 >= 0) {
        const double E = sqrt(E2);
        if (nb == 0) {
            x[0] = -R_2 + 0.5 * E - b_4;
            x[1] = x[0] - E;
            nb = 2;
        } else {
            x[2] = -R_2 + 0.5 * E - b_4;
            x[3] = x[2] - E;
            nb = 4;
        }
    }
    return nb;
}

// ------------------------------------------------------------------------------------------------
// Absolute orientation of the P3P triangle: M[k] (camera-frame points) = R * X[k] + T, k = 0..2.
// OpenCV solves this with Horn's unit-quaternion least squares (a 4x4 Jacobi eigen-solve per root, the long
// pole of P3P on one lane).  The camera-frame triangle is congruent to the object triangle by construction
// (its side lengths come out of the P3P length solve), so the rotation is determined exactly by the two
// orthonormal triads of the triangles and the least-squares machinery is not needed: R = Tc * Tw^T with
// T* = [e1 e2 e3], e1 = (P1-P0)/|.|, e3 = e1 x (P2-P0)/|.|, e2 = e3 x e1 (Horn 1987, section 2.A).  It agrees
// with the quaternion solution to rounding (the residual incongruence of the two triangles is ~1e-13).
// ------------------------------------------------------------------------------------------------
DM_INLINE void triad(const double P[3][3], double E[9]) {  // E columns = e1, e2, e3 (row-major 3x3)
    double a[3] = {P[1][0] - P[0][0], P[1][1] - P[0][1], P[1][2] - P[0][2]};
    const double b[3] = {P[2][0] - P[0][0], P[2][1] - P[0][1], P[2][2] - P[0][2]};
    const double ia = 1.0 / sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    a[0] *= ia; a[1] *= ia; a[2] *= ia;
    double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    const double in = 1.0 / sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    n[0] *= in; n[1] *= in; n[2] *= in;
    const double m[3] = {n[1] * a[2] - n[2] * a[1], n[2] * a[0] - n[0] * a[2], n[0] * a[1] - n[1] * a[0]};
    E[0] = a[0]; E[1] = m[0]; E[2] = n[0];
    E[3] = a[1]; E[4] = m[1]; E[5] = n[1];
    E[6] = a[2]; E[7] = m[2]; E[8] = n[2];
}

// Ew = triad of the object triangle (hoisted: it is the same for all quartic roots)
DM_INLINE void align3(const double M[3][3], const double X[3][3], const double Ew[9], double R[9], double T[3]) {
    double Ec[9];
    triad(M, Ec);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R[i * 3 + j] = Ec[i * 3 + 0] * Ew[j * 3 + 0] + Ec[i * 3 + 1] * Ew[j * 3 + 1] + Ec[i * 3 + 2] * Ew[j * 3 + 2];
    const double third = 1.0 / 3.0;
    double Cs[3], Ce[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        Ce[j] = (M[0][j] + M[1][j] + M[2][j]) * third;
        Cs[j] = (X[0][j] + X[1][j] + X[2][j]) * third;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) T[i] = Ce[i] - (R[i * 3] * Cs[0] + R[i * 3 + 1] * Cs[1] + R[i * 3 + 2] * Cs[2]);
}

// ------------------------------------------------------------------------------------------------
// Horn's absolute orientation for 3 correspondences, the way OpenCV's P3P does it: largest eigenvector of the 4x4
// N matrix by cyclic Jacobi sweeps (threshold schedule of the classic routine: 0.2*sum/16 for the first three
// sweeps).  Used where parity with OpenCV's rounding matters more than speed: K5 (dPNP) takes central differences
// of this pose with a 0.1 mm step, and on near-degenerate minimal sets the two alignment methods distribute the
// residual incongruence of the triangles differently, which the 1/(2 eps) then amplifies.
// ------------------------------------------------------------------------------------------------
DM_INLINE void jrot(double& g_, double& h_, double s, double tau) {
    const double g = g_, h = h_;
    g_ = g - s * (h + g * tau);
    h_ = h + s * (g - h * tau);
}

DM_INLINE void jacobi4(double A[16], double D[4], double U[16]) {
    double B[4], Z[4];
#pragma unroll
    for (int i = 0; i < 16; i++) U[i] = (i % 5 == 0) ? 1.0 : 0.0;
    B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
#pragma unroll
    for (int i = 0; i < 4; i++) { D[i] = B[i]; Z[i] = 0; }
    for (int iter = 0; iter < 50; iter++) {
        const double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
        if (sum == 0.0) return;
        const double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = i + 1; j < 4; j++) {
                const double Aij = A[4 * i + j];
                const double eps_machine = 100.0 * fabs(Aij);
                if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j])) {
                    A[4 * i + j] = 0.0;
                } else if (fabs(Aij) > tresh) {
                    double hh = D[j] - D[i], t;
                    if (fabs(hh) + eps_machine == fabs(hh)) t = Aij / hh;
                    else {
                        const double theta = 0.5 * hh / Aij;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    hh = t * Aij;
                    Z[i] -= hh; Z[j] += hh; D[i] -= hh; D[j] += hh;
                    A[4 * i + j] = 0.0;
                    const double c = 1.0 / sqrt(1 + t * t), s = t * c, tau = s / (1.0 + c);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (k < i) jrot(A[k * 4 + i], A[k * 4 + j], s, tau);
                        else if (k > i && k < j) jrot(A[i * 4 + k], A[k * 4 + j], s, tau);
                        else if (k > j) jrot(A[i * 4 + k], A[j * 4 + k], s, tau);
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) jrot(U[k * 4 + i], U[k * 4 + j], s, tau);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { B[i] += Z[i]; D[i] = B[i]; Z[i] = 0; }
    }
}

// M[k] (camera-frame points) ~= R * X[k] + T  for k = 0..2
DM_INLINE void align3_horn(const double M[3][3], const double X[3][3], double R[9], double T[3]) {
    double Cs[3], Ce[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        Ce[j] = (M[0][j] + M[1][j] + M[2][j]) / 3;
        Cs[j] = (X[0][j] + X[1][j] + X[2][j]) / 3;
    }
    double s[9];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int j = 0; j < 3; j++) s[a * 3 + j] = (X[0][a] * M[0][j] + X[1][a] * M[1][j] + X[2][a] * M[2][j]) / 3 - Ce[j] * Cs[a];
    double Q[16], ev[4], U[16];
    Q[0] = s[0] + s[4] + s[8];
    Q[5] = s[0] - s[4] - s[8];
    Q[10] = s[4] - s[8] - s[0];
    Q[15] = s[8] - s[0] - s[4];
    Q[4] = Q[1] = s[5] - s[7];
    Q[8] = Q[2] = s[6] - s[2];
    Q[12] = Q[3] = s[1] - s[3];
    Q[9] = Q[6] = s[3] + s[1];
    Q[13] = Q[7] = s[6] + s[2];
    Q[14] = Q[11] = s[7] + s[5];
    jacobi4(Q, ev, U);
    // eigenvector of the largest eigenvalue, first maximum wins
    double evm = ev[0];
    double q0 = U[0], q1 = U[4], q2 = U[8], q3 = U[12];
#pragma unroll
    for (int i = 1; i < 4; i++) {
        if (ev[i] > evm) { evm = ev[i]; q0 = U[i]; q1 = U[4 + i]; q2 = U[8 + i]; q3 = U[12 + i]; }
    }
    const double q02 = q0 * q0, q12 = q1 * q1, q22 = q2 * q2, q32 = q3 * q3;
    const double q0_1 = q0 * q1, q0_2 = q0 * q2, q0_3 = q0 * q3, q1_2 = q1 * q2, q1_3 = q1 * q3, q2_3 = q2 * q3;
    R[0] = q02 + q12 - q22 - q32; R[1] = 2. * (q1_2 - q0_3); R[2] = 2. * (q1_3 + q0_2);
    R[3] = 2. * (q1_2 + q0_3); R[4] = q02 + q22 - q12 - q32; R[5] = 2. * (q2_3 - q0_1);
    R[6] = 2. * (q1_3 - q0_2); R[7] = 2. * (q2_3 + q0_1); R[8] = q02 + q32 - q12 - q22;
#pragma unroll
    for (int i = 0; i < 3; i++) T[i] = Ce[i] - (R[i * 3] * Cs[0] + R[i * 3 + 1] * Cs[1] + R[i * 3 + 2] * Cs[2]);
}

// ------------------------------------------------------------------------------------------------
// Horn's least-squares alignment of three correspondences in CLOSED FORM: the optimum OpenCV's Jacobi sweeps converge to, without the
// sweeps.  The centred cross-covariance S of three points has rank 2 (three centred vectors are linearly dependent), so the spectrum of
// Horn's 4x4 matrix N is {+-(s1 + s2), +-(s1 - s2)} with s1 >= s2 the non-zero singular values of S, and its largest eigenvalue is
//     lambda = sqrt(|S|_F^2 + 2 |cof S|_F)          (s1^2 + s2^2 = |S|_F^2,  s1 s2 = |cof S|_F for a rank-2 matrix)
// -- sums of squares only, no cancellation.  The eigenvector is a column of adj(N - lambda I) (= c q q^T: every column is a multiple of
// q; the one with the largest diagonal entry is the best conditioned).  Against numpy's eigh on 20 000 random triangles, congruent or
// not, thin ones included: rotation / translation within 5e-12 (scripts/micro/horn_closed_form.py).  This is what K1 aligns with since
// round 5: when the P3P lengths of a root are inconsistent (near a double root of the quartic) the triad and the least-squares solution
// are different poses -- by pixels on ill-conditioned minimal sets -- and a different pose accepts a different minimal set
// (profiles/r05_k1_alignment.txt).
// ------------------------------------------------------------------------------------------------
DM_INLINE void align3_lsq(const double M[3][3], const double X[3][3], double R[9], double T[3]) {
    // fused multiply-adds are welcome here, whatever the file is built with: this is not OpenCV's rounding sequence, it is the optimum itself
#pragma clang fp contract(fast)
    const double third = 1.0 / 3.0;
    double Cs[3], Ce[3], x[3][3], m[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        Ce[j] = (M[0][j] + M[1][j] + M[2][j]) * third;
        Cs[j] = (X[0][j] + X[1][j] + X[2][j]) * third;
#pragma unroll
        for (int k = 0; k < 3; k++) { x[k][j] = X[k][j] - Cs[j]; m[k][j] = M[k][j] - Ce[j]; }
    }
    double s[9];  // 3 x OpenCV's s (the scale drops out of the eigenvector)
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int j = 0; j < 3; j++) s[a * 3 + j] = x[0][a] * m[0][j] + x[1][a] * m[1][j] + x[2][a] * m[2][j];
    double f2 = 0, c2 = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) f2 += s[i] * s[i];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            const double k = s[i1 * 3 + j1] * s[i2 * 3 + j2] - s[i1 * 3 + j2] * s[i2 * 3 + j1];
            c2 += k * k;
        }
    const double lam = sqrt(f2 + 2 * sqrt(c2));
    const double a00 = s[0] + s[4] + s[8] - lam, a11 = s[0] - s[4] - s[8] - lam, a22 = s[4] - s[8] - s[0] - lam, a33 = s[8] - s[0] - s[4] - lam;
    const double a01 = s[5] - s[7], a02 = s[6] - s[2], a03 = s[1] - s[3], a12 = s[3] + s[1], a13 = s[6] + s[2], a23 = s[7] + s[5];
    // adjugate of the symmetric 4x4 through its 2x2 minors (rows 0-1: m*, rows 2-3: n*)
    const double m0 = a00 * a11 - a01 * a01, m1 = a00 * a12 - a01 * a02, m2 = a00 * a13 - a01 * a03;
    const double m3 = a01 * a12 - a11 * a02, m4 = a01 * a13 - a11 * a03, m5 = a02 * a13 - a12 * a03;
    const double n5 = a22 * a33 - a23 * a23, n4 = a12 * a33 - a13 * a23, n3 = a12 * a23 - a13 * a22;
    const double n2 = a02 * a33 - a03 * a23, n1 = a02 * a23 - a03 * a22;
    const double b00 = a11 * n5 - a12 * n4 + a13 * n3, b11 = a00 * n5 - a02 * n2 + a03 * n1;
    const double b22 = a03 * m4 - a13 * m2 + a33 * m0, b33 = a02 * m3 - a12 * m1 + a22 * m0;
    const double b01 = -a01 * n5 + a02 * n4 - a03 * n3, b02 = a13 * m5 - a23 * m4 + a33 * m3, b03 = -a12 * m5 + a22 * m4 - a23 * m3;
    const double b12 = -a03 * m5 + a23 * m2 - a33 * m1, b13 = a02 * m5 - a22 * m2 + a23 * m1, b23 = -a02 * m4 + a12 * m2 - a23 * m0;
    double q0 = b00, q1 = b01, q2 = b02, q3 = b03, best = fabs(b00);
    if (fabs(b11) > best) { best = fabs(b11); q0 = b01; q1 = b11; q2 = b12; q3 = b13; }
    if (fabs(b22) > best) { best = fabs(b22); q0 = b02; q1 = b12; q2 = b22; q3 = b23; }
    if (fabs(b33) > best) { q0 = b03; q1 = b13; q2 = b23; q3 = b33; }
    const double inv = 1.0 / sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    q0 *= inv; q1 *= inv; q2 *= inv; q3 *= inv;
    const double q02 = q0 * q0, q12 = q1 * q1, q22 = q2 * q2, q32 = q3 * q3;
    const double q0_1 = q0 * q1, q0_2 = q0 * q2, q0_3 = q0 * q3, q1_2 = q1 * q2, q1_3 = q1 * q3, q2_3 = q2 * q3;
    R[0] = q02 + q12 - q22 - q32; R[1] = 2. * (q1_2 - q0_3); R[2] = 2. * (q1_3 + q0_2);
    R[3] = 2. * (q1_2 + q0_3); R[4] = q02 + q22 - q12 - q32; R[5] = 2. * (q2_3 - q0_1);
    R[6] = 2. * (q1_3 - q0_2); R[7] = 2. * (q2_3 + q0_1); R[8] = q02 + q32 - q12 - q22;
#pragma unroll
    for (int i = 0; i < 3; i++) T[i] = Ce[i] - (R[i * 3] * Cs[0] + R[i * 3 + 1] * Cs[1] + R[i * 3 + 2] * Cs[2]);
}

// ------------------------------------------------------------------------------------------------
// solvePnP(CV_P3P): 4 correspondences -> cv pose.  X: 4 object points (float, mm), uv: 4 pixel positions.
// Split in two so that the (up to four) quartic roots can be evaluated either in sequence by one lane
// (p3p) or by four neighbouring lanes in parallel (p3p_setup + p3p_eval_root, used by K1).
// ------------------------------------------------------------------------------------------------
// The setup keeps only what the quartic and the length recovery need (20 doubles).  The object points stay in their float
// registers, the 4th image point and the object triangle's triad are re-derived per root: K1 evaluates the roots on four lanes in
// parallel, so hoisting them bought nothing there and cost 46 live registers across the quartic solve (288 -> 2 waves per SIMD).
struct P3PSetup {
    double f[3][3];   // unit bearing vectors of points 0..2
    double a, b, p, q, r, d2, inv_b0;
    double roots[4];
    int n;
};

// undistortPoints (zero distortion) rounds the normalised coordinates to float; the solver then maps them back to pixels.
DM_INLINE void p3p_image_point(const float uv[2], const Cam& K, double& mu, double& mv) {
#pragma clang fp contract(off)
    const float xn = (float)(((double)uv[0] - K.cx) * (1. / K.fx));
    const float yn = (float)(((double)uv[1] - K.cy) * (1. / K.fy));
    mu = (double)xn * K.fx + K.cx;
    mv = (double)yn * K.fy + K.cy;
}

DM_INLINE bool p3p_setup(const float X[4][3], const float uv[4][2], const Cam& K, P3PSetup& S) {
#pragma clang fp contract(off)
    const double inv_fx = 1. / K.fx, inv_fy = 1. / K.fy, cx_fx = K.cx / K.fx, cy_fy = K.cy / K.fy;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double mu, mv;
        p3p_image_point(uv[i], K, mu, mv);
        const double u = inv_fx * mu - cx_fx, v = inv_fy * mv - cy_fy;
        const double k = 1. / sqrt(u * u + v * v + 1);
        S.f[i][0] = u * k; S.f[i][1] = v * k; S.f[i][2] = k;
    }

    auto dist = [&](int a, int b) {
        const double dx = (double)X[a][0] - (double)X[b][0], dy = (double)X[a][1] - (double)X[b][1], dz = (double)X[a][2] - (double)X[b][2];
        return sqrt(dx * dx + dy * dy + dz * dz);
    };
    auto dot = [&](int a, int b) { return S.f[a][0] * S.f[b][0] + S.f[a][1] * S.f[b][1] + S.f[a][2] * S.f[b][2]; };
    const double d0 = dist(1, 2), d1 = dist(0, 2), d2 = dist(0, 1);
    const double p = dot(1, 2) * 2, q = dot(0, 2) * 2, r = dot(0, 1) * 2;
    const double inv_d22 = 1. / (d2 * d2);
    const double a = inv_d22 * (d0 * d0), b = inv_d22 * (d1 * d1);
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    const double pr = p * r, pqr = q * pr;
    S.n = 0;
    if (p2 + q2 + r2 - pqr - 1 == 0) return false;
    const double ab = a * b, a_2 = 2 * a;
    const double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return false;
    const double a_4 = 4 * a;
    const double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    const double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    const double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    const double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    const double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    const double b0 = b * temp * temp;
    if (b0 == 0) return false;
    S.n = roots4(A, B, C, D, E, S.roots);
    if (S.n == 0) return false;
    S.a = a; S.b = b; S.p = p; S.q = q; S.r = r; S.d2 = d2; S.inv_b0 = 1. / b0;
    return true;
}

// One quartic root x -> (R, T, squared reprojection error of the 4th point).  false: root rejected.
template <bool HORN = false>
DM_INLINE bool p3p_eval_root(const P3PSetup& S, const float X[4][3], const float uv[4][2], const Cam& K, double x, double Rc[9], double Tc[3],
                             double& reproj) {
#pragma clang fp contract(off)
    if (!(x > 0)) return false;
    const double a = S.a, b = S.b, p = S.p, q = S.q, r = S.r;
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r, ab = a * b, a_2 = 2 * a, a_4 = 4 * a;
    const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    const double x2 = x * x;
    const double b1 = ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
                      (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
                        (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
                       (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
                        pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
                       2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
                       p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
    if (!(b1 > 0)) return false;
    const double y = S.inv_b0 * b1;
    const double v = x2 + y * y - x * y * r;
    if (!(v > 0)) return false;
    const double Zl = S.d2 / sqrt(v);
    const double L[3] = {x * Zl, y * Zl, Zl};
    double M[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int j = 0; j < 3; j++) M[k][j] = L[k] * S.f[k][j];
    double Xw[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Xw[i][j] = X[i][j];
    if (HORN) align3_horn(M, Xw, Rc, Tc);   // OpenCV's own iteration (Jacobi sweeps), rounding and all
    else {
#ifdef DSAC_ALIGN_TRIAD                     // rounds 1-4: exact only for congruent triangles (kept for the A/B of profiles/r05_k1_alignment.txt)
        double Ew[9];
        triad(Xw, Ew);
        align3(M, Xw, Ew, Rc, Tc);
#else
        align3_lsq(M, Xw, Rc, Tc);          // the same least-squares optimum in closed form
#endif
    }
    const double X30 = X[3][0], X31 = X[3][1], X32 = X[3][2];
    const double X3p = Rc[0] * X30 + Rc[1] * X31 + Rc[2] * X32 + Tc[0];
    const double Y3p = Rc[3] * X30 + Rc[4] * X31 + Rc[5] * X32 + Tc[1];
    const double Z3p = Rc[6] * X30 + Rc[7] * X31 + Rc[8] * X32 + Tc[2];
    const double mu3p = K.cx + K.fx * X3p / Z3p, mv3p = K.cy + K.fy * Y3p / Z3p;
    double mu3, mv3;
    p3p_image_point(uv[3], K, mu3, mv3);
    reproj = (mu3p - mu3) * (mu3p - mu3) + (mv3p - mv3) * (mv3p - mv3);
    return true;
}

// All roots in sequence on one lane; the root whose pose re-projects the 4th point best wins (first on ties).
template <bool HORN = false>
DM_INLINE bool p3p(const float X[4][3], const float uv[4][2], const Cam& K, double cv6[6]) {
    P3PSetup S;
    if (!p3p_setup(X, uv, K, S)) return false;
    bool have = false;
    double best = 0, Rb[9], Tb[3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i >= S.n) continue;
        double Rc[9], Tc[3], reproj;
        if (!p3p_eval_root<HORN>(S, X, uv, K, S.roots[i], Rc, Tc, reproj)) continue;
        if (!have || best > reproj) {
            have = true;
            best = reproj;
#pragma unroll
            for (int k = 0; k < 9; k++) Rb[k] = Rc[k];
            Tb[0] = Tc[0]; Tb[1] = Tc[1]; Tb[2] = Tc[2];
        }
    }
    if (!have) return false;
    rodrigues_m2v(Rb, cv6);
    cv6[3] = Tb[0]; cv6[4] = Tb[1]; cv6[5] = Tb[2];
    return true;
}

// ---- the four root lanes of a P3P solve (K1's four-lane form, K5, the DSAC variant's replica start poses) --------------------------
// value of lane (lane & ~3) | I: the four root lanes of an attempt are one DPP quad
template <int I>
DM_INLINE int quad_bcast_i(int v) { return __builtin_amdgcn_update_dpp(0, v, I * 0x55, 0xf, 0xf, true); }
template <int I>
DM_INLINE double quad_bcast_d(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)quad_bcast_i<I>((int)(unsigned)u), hi = (unsigned)quad_bcast_i<I>((int)(unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// winner among the 4 roots of an attempt: smallest re-projection error of the 4th point, first on ties (-1: no candidate)
DM_INLINE int best_root_of_quad(bool cand, double reproj) {
    int win = -1;
    double best = 0;
    const int ci[4] = {quad_bcast_i<0>((int)cand), quad_bcast_i<1>((int)cand), quad_bcast_i<2>((int)cand), quad_bcast_i<3>((int)cand)};
    const double ri[4] = {quad_bcast_d<0>(reproj), quad_bcast_d<1>(reproj), quad_bcast_d<2>(reproj), quad_bcast_d<3>(reproj)};
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (ci[i] && (win < 0 || best > ri[i])) { win = i; best = ri[i]; }
    return win;
}

// ---- sampling RNG (include/dsac_hip.h) ------------------------------------------------------------
DM_INLINE uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// candidate cell k of attempt `attempt`: ONE 64-bit draw, x from its high half and y from its low half (multiply-shift into [0, W) x [0, H))
DM_INLINE int draw_cell(uint64_t key, uint32_t attempt, uint32_t k, uint32_t W, uint32_t H) {
    const uint64_t v = mix64(key + (((uint64_t)attempt << 16) | k));
    const uint32_t x = (uint32_t)(((v >> 32) * (uint64_t)W) >> 32);
    const uint32_t y = (uint32_t)(((v & 0xffffffffull) * (uint64_t)H) >> 32);
    return (int)(y * W + x);
}
DM_INLINE uint64_t hyp_key(uint64_t seed, uint32_t hyp) { return mix64(seed ^ mix64((uint64_t)hyp)); }

}  // namespace dm
