// oracle/cvlike.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// CPU stand-ins (double precision, plain C++17, no dependencies) for the *un-vendored*
// OpenCV 2.4 calib3d/core routines that the reference's hot path calls:
//
//   cv::Rodrigues      <- core/cnn_softam.h:507-508, core/maxloss.h:95-96, core/types.h:143,168,190,
//                         core/Hypothesis.cpp:94,270
//   cv::projectPoints  <- core/cnn_softam.h:351,1046
//   cv::solvePnP(P3P)  <- core/cnn_softam.h:66 via :120,129,597,1042
//   cv::solvePnP(ITERATIVE, useExtrinsicGuess=true) <- core/cnn_softam.h:66 via :708,1144
//
// OpenCV is a third-party dependency that is ABSENT from /root/reference (core/CMakeLists.txt:19 only says
// find_package(OpenCV REQUIRED); documentation.pdf p.1 names "OpenCV 2.4").  What follows restates the
// PUBLISHED algorithms those entry points implement:
//   * Rodrigues' rotation formula and its analytic derivative,
//   * the pinhole projection  (X,Y,Z) = R*M + t ; z = Z ? 1/Z : 1 ; u = fx*X*z + cx,
//   * Gao, Hou, Tang, Cheng, "Complete Solution Classification for the Perspective-Three-Point
//     Problem", PAMI 25(8) 2003 (main branch) + the closed-form quartic (Ferrari / MathWorld
//     "Quartic Equation") + Horn, "Closed-form solution of absolute orientation using unit
//     quaternions", JOSA-A 1987 with a cyclic Jacobi 4x4 eigen-solver; the root whose pose
//     re-projects the 4th point best is returned,
//   * Levenberg-Marquardt on the 6 pose parameters with Marquardt diagonal scaling (1+lambda),
//     lambda0 = 1e-3, x10 / /10 updates, <= 20 iterations, eps = FLT_EPSILON on the relative
//     parameter change (the state machine of OpenCV's CvLevMarq).
//
// PARITY UNPINNED: OpenCV cannot be built or imported in this container and the reference holds no
// golden vectors, so these stand-ins are pinned only by closed-form / SciPy / torch-autograd checks
// (tests/test_oracle_*.py, tests/golden/).  See DESIGN.md "Oracle".
#pragma once
#include <cmath>
#include <cfloat>
#include <cstring>
#include <algorithm>
#include <vector>

namespace cvl {

struct Cam { double fx, fy, cx, cy; };

static inline bool has_nan(const double* v, int n) {
    for (int i = 0; i < n; i++) if (v[i] != v[i]) return true;
    return false;
}

// ---------------------------------------------------------------------------------------------
// 3x3 helpers (row-major)
// ---------------------------------------------------------------------------------------------
static inline void mat3_mul(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            T[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
    std::memcpy(C, T, sizeof(T));
}
static inline void mat3_t(const double* A, double* At) {
    double T[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[j * 3 + i] = A[i * 3 + j];
    std::memcpy(At, T, sizeof(T));
}
static inline double mat3_det(const double* A) {
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

// General n x n inverse by Gauss-Jordan with partial pivoting (stand-in for cv::Mat::inv(), default
// DECOMP_LU).  Returns false when singular (OpenCV then returns a zero matrix).
static inline bool mat_inv(const double* A, double* Ainv, int n) {
    std::vector<double> M(n * 2 * n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) { M[i * 2 * n + j] = A[i * n + j]; M[i * 2 * n + n + j] = (i == j); }
    }
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++) if (std::fabs(M[r * 2 * n + c]) > std::fabs(M[piv * 2 * n + c])) piv = r;
        if (std::fabs(M[piv * 2 * n + c]) < DBL_EPSILON * 1e-3 || M[piv * 2 * n + c] != M[piv * 2 * n + c]) {
            std::fill(Ainv, Ainv + n * n, 0.0);
            return false;
        }
        if (piv != c) for (int j = 0; j < 2 * n; j++) std::swap(M[c * 2 * n + j], M[piv * 2 * n + j]);
        double d = 1.0 / M[c * 2 * n + c];
        for (int j = 0; j < 2 * n; j++) M[c * 2 * n + j] *= d;
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            double f = M[r * 2 * n + c];
            if (f == 0) continue;
            for (int j = 0; j < 2 * n; j++) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Ainv[i * n + j] = M[i * 2 * n + n + j];
    return true;
}

// Cyclic Jacobi eigen-decomposition of a symmetric 3x3 (used for the polar factor below).
static inline void sym3_eig(const double* S, double* evals, double* V) {
    double A[9];
    std::memcpy(A, S, sizeof(A));
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = std::fabs(A[1]) + std::fabs(A[2]) + std::fabs(A[5]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double apq = A[p * 3 + q];
                if (std::fabs(apq) < 1e-300) continue;
                double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 3; k++) {  // A <- A*G
                    double akp = A[k * 3 + p], akq = A[k * 3 + q];
                    A[k * 3 + p] = c * akp - s * akq;
                    A[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {  // A <- G^T*A
                    double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                    A[p * 3 + k] = c * apk - s * aqk;
                    A[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - s * vkq;
                    V[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    evals[0] = A[0]; evals[1] = A[4]; evals[2] = A[8];
}

// Orthogonal polar factor of R: U*V^T of the SVD R = U*W*V^T  (= R * (R^T R)^(-1/2)).
// OpenCV's matrix->vector Rodrigues starts with exactly this re-orthonormalisation.
static inline void polar_orthonormalise(const double* R, double* Q) {
    double Rt[9], S[9], ev[3], V[9];
    mat3_t(R, Rt);
    mat3_mul(Rt, R, S);
    sym3_eig(S, ev, V);
    // (R^T R)^(-1/2) = V diag(1/sqrt(ev)) V^T
    double D[9] = {0};
    for (int i = 0; i < 3; i++) D[i * 4] = ev[i] > 1e-300 ? 1.0 / std::sqrt(ev[i]) : 0.0;
    double Vt[9], T[9], Sm[9];
    mat3_t(V, Vt);
    mat3_mul(V, D, T);
    mat3_mul(T, Vt, Sm);
    mat3_mul(R, Sm, Q);
}

// ---------------------------------------------------------------------------------------------
// Rodrigues, vector -> matrix.  J (optional) is 3x9: row i = d(R row-major)/d r_i  (OpenCV layout;
// the reference transposes it to 9x3 at core/cnn_softam.h:509 and core/maxloss.h:160,168).
// ---------------------------------------------------------------------------------------------
static inline void rodrigues_vec2mat(const double* r, double* R, double* J /*27 or null*/) {
    double rx = r[0], ry = r[1], rz = r[2];
    double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0);
        if (J) {
            std::fill(J, J + 27, 0.0);
            J[5] = J[15] = J[19] = -1;   // d[r]x / dr at r = 0
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double c = std::cos(theta), s = std::sin(theta), c1 = 1.0 - c, it = 1.0 / theta;
    double ax = rx * it, ay = ry * it, az = rz * it;
    double aat[9] = {ax * ax, ax * ay, ax * az, ax * ay, ay * ay, ay * az, ax * az, ay * az, az * az};
    double ax_[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * aat[k] + s * ax_[k];
    if (!J) return;
    // d(aat)/d a_i and d([a]x)/d a_i, then chain through a = r/theta, theta = |r|
    const double daat[27] = {2 * ax, ay, az, ay, 0, 0, az, 0, 0,
                             0, ax, 0, ax, 2 * ay, az, 0, az, 0,
                             0, 0, ax, 0, 0, ay, ax, ay, 2 * az};
    const double dax[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
                            0, 0, 1, 0, 0, 0, -1, 0, 0,
                            0, -1, 0, 1, 0, 0, 0, 0, 0};
    const double a[3] = {ax, ay, az};
    for (int i = 0; i < 3; i++) {
        double ai = a[i];
        double k0 = -s * ai;                      // d cos / d r_i
        double k1 = (s - 2 * c1 * it) * ai;       // (d(1-cos) - 2(1-cos)/theta) a_i   on aat
        double k2 = c1 * it;                      // (1-cos)/theta                      on daat_i
        double k3 = (c - s * it) * ai;            // (cos - sin/theta) a_i              on [a]x
        double k4 = s * it;                       // sin/theta                          on d[a]x_i
        for (int k = 0; k < 9; k++)
            J[i * 9 + k] = k0 * I[k] + k1 * aat[k] + k2 * daat[i * 9 + k] + k3 * ax_[k] + k4 * dax[i * 9 + k];
    }
}

// Rodrigues, matrix -> vector (no Jacobian needed on this path).
static inline void rodrigues_mat2vec(const double* Rin, double* r) {
    for (int i = 0; i < 9; i++)
        if (!(Rin[i] > -100 && Rin[i] < 100)) { r[0] = r[1] = r[2] = 0; return; }  // range/NaN guard
    double R[9];
    polar_orthonormalise(Rin, R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { rx = ry = rz = 0; }
        else {
            double t;
            t = (R[0] + 1) * 0.5; rx = std::sqrt(std::max(t, 0.));
            t = (R[4] + 1) * 0.5; ry = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; rz = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = theta / (2 * s);
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
}

// ---------------------------------------------------------------------------------------------
// projectPoints (zero distortion): promoted to double, result rounded to float (Point2f output).
// Optional Jacobians d(u,v)/d(rvec) [2x3] and d(u,v)/d(tvec) [2x3] per point (double).
// ---------------------------------------------------------------------------------------------
static inline void project_points(int n, const float* X, const double* rvec, const double* tvec, const Cam& K,
                                  float* uv, double* uv_d = nullptr, double* dpdr = nullptr, double* dpdt = nullptr) {
    double R[9], dRdr[27];
    rodrigues_vec2mat(rvec, R, dpdr ? dRdr : nullptr);
    for (int i = 0; i < n; i++) {
        double Mx = X[i * 3 + 0], My = X[i * 3 + 1], Mz = X[i * 3 + 2];
        double Xc = R[0] * Mx + R[1] * My + R[2] * Mz + tvec[0];
        double Yc = R[3] * Mx + R[4] * My + R[5] * Mz + tvec[1];
        double Zc = R[6] * Mx + R[7] * My + R[8] * Mz + tvec[2];
        double z = Zc ? 1. / Zc : 1.;
        double x = Xc * z, y = Yc * z;
        double u = x * K.fx + K.cx, v = y * K.fy + K.cy;
        if (uv) { uv[i * 2 + 0] = (float)u; uv[i * 2 + 1] = (float)v; }
        if (uv_d) { uv_d[i * 2 + 0] = u; uv_d[i * 2 + 1] = v; }
        if (dpdt) {
            double* d = dpdt + i * 6;
            d[0] = K.fx * z; d[1] = 0; d[2] = -K.fx * x * z;
            d[3] = 0; d[4] = K.fy * z; d[5] = -K.fy * y * z;
        }
        if (dpdr) {
            double* d = dpdr + i * 6;
            for (int j = 0; j < 3; j++) {
                const double* dR = dRdr + j * 9;
                double dX = dR[0] * Mx + dR[1] * My + dR[2] * Mz;
                double dY = dR[3] * Mx + dR[4] * My + dR[5] * Mz;
                double dZ = dR[6] * Mx + dR[7] * My + dR[8] * Mz;
                d[j] = K.fx * z * (dX - x * dZ);
                d[3 + j] = K.fy * z * (dY - y * dZ);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Closed-form polynomial roots (real roots only).
// ---------------------------------------------------------------------------------------------
static inline int roots_deg2(double a, double b, double c, double* x) {
    double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    double inv_2a = 0.5 / a;
    if (delta == 0) { x[0] = x[1] = -b * inv_2a; return 1; }
    double sq = std::sqrt(delta);
    x[0] = (-b + sq) * inv_2a;
    x[1] = (-b - sq) * inv_2a;
    return 2;
}

static inline int roots_deg3(double a, double b, double c, double d, double* x) {
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            x[0] = -d / c;
            return 1;
        }
        x[2] = 0;
        return roots_deg2(b, c, d, x);
    }
    double inv_a = 1. / a;
    double b_a = inv_a * b, b_a2 = b_a * b_a, c_a = inv_a * c, d_a = inv_a * d;
    double Q = (3 * c_a - b_a2) / 9;
    double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    double Q3 = Q * Q * Q;
    double D = Q3 + R * R;
    double b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (R == 0) { x[0] = x[1] = x[2] = -b_a_3; return 3; }
        x[0] = std::pow(2 * R, 1 / 3.0) - b_a_3;
        return 1;
    }
    if (D <= 0) {  // three real roots (trigonometric form)
        double theta = std::acos(R / std::sqrt(-Q3));
        double sqrt_Q = std::sqrt(-Q);
        x[0] = 2 * sqrt_Q * std::cos(theta / 3.0) - b_a_3;
        x[1] = 2 * sqrt_Q * std::cos((theta + 2 * M_PI) / 3.0) - b_a_3;
        x[2] = 2 * sqrt_Q * std::cos((theta + 4 * M_PI) / 3.0) - b_a_3;
        return 3;
    }
    double AD = std::pow(std::fabs(R) + std::sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    double BD = (AD == 0) ? 0 : -Q / AD;
    x[0] = AD + BD - b_a_3;
    return 1;
}

static inline int roots_deg4(double a, double b, double c, double d, double e, double* x) {
    if (a == 0) { x[3] = 0; return roots_deg3(b, c, d, e, x); }
    double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r[3];
    int n = roots_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r);  // resolvent cubic
    if (n == 0) return 0;
    double R2 = 0.25 * b2 - c + r[0];
    if (R2 < 0) return 0;
    double R = std::sqrt(R2), inv_R = 1. / R;
    int nb = 0;
    double D2, E2;
    if (R < 10E-12) {
        double temp = r[0] * r[0] - 4 * e;
        if (temp < 0) D2 = E2 = -1;
        else {
            double st = std::sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * st;
            E2 = D2 - 4 * st;
        }
    } else {
        double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    double b_4 = 0.25 * b, R_2 = 0.5 * R;
    if (D2 >= 0) {
        double D = std::sqrt(D2);
        nb = 2;
        x[0] = R_2 + 0.5 * D - b_4;
        x[1] = x[0] - D;
    }
    if (E2 >= 0) {
        double E = std::sqrt(E2);
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

// ---------------------------------------------------------------------------------------------
// P3P (Gao et al. 2003, main branch) as used by solvePnP(..., CV_P3P)
// ---------------------------------------------------------------------------------------------
// Cyclic Jacobi eigen-solver for a symmetric 4x4 (Numerical-Recipes style thresholds).
static inline bool jacobi_4x4(double* A, double* D, double* U) {
    double B[4], Z[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; i++) U[i] = (i % 5 == 0);
    B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
    std::memcpy(D, B, sizeof(B));
    for (int iter = 0; iter < 50; iter++) {
        double sum = std::fabs(A[1]) + std::fabs(A[2]) + std::fabs(A[3]) + std::fabs(A[6]) + std::fabs(A[7]) + std::fabs(A[11]);
        if (sum == 0.0) return true;
        double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
        for (int i = 0; i < 3; i++) {
            for (int j = i + 1; j < 4; j++) {
                double* pAij = A + 4 * i + j;
                double Aij = *pAij;
                double eps_machine = 100.0 * std::fabs(Aij);
                if (iter > 3 && std::fabs(D[i]) + eps_machine == std::fabs(D[i]) && std::fabs(D[j]) + eps_machine == std::fabs(D[j]))
                    *pAij = 0.0;
                else if (std::fabs(Aij) > tresh) {
                    double hh = D[j] - D[i], t;
                    if (std::fabs(hh) + eps_machine == std::fabs(hh)) t = Aij / hh;
                    else {
                        double theta = 0.5 * hh / Aij;
                        t = 1.0 / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    hh = t * Aij;
                    Z[i] -= hh; Z[j] += hh; D[i] -= hh; D[j] += hh;
                    *pAij = 0.0;
                    double c = 1.0 / std::sqrt(1 + t * t), s = t * c, tau = s / (1.0 + c);
                    auto rot = [&](double& g_, double& h_) {
                        double g = g_, h = h_;
                        g_ = g - s * (h + g * tau);
                        h_ = h + s * (g - h * tau);
                    };
                    for (int k = 0; k <= i - 1; k++) rot(A[k * 4 + i], A[k * 4 + j]);
                    for (int k = i + 1; k <= j - 1; k++) rot(A[i * 4 + k], A[k * 4 + j]);
                    for (int k = j + 1; k < 4; k++) rot(A[i * 4 + k], A[j * 4 + k]);
                    for (int k = 0; k < 4; k++) rot(U[k * 4 + i], U[k * 4 + j]);
                }
            }
        }
        for (int i = 0; i < 4; i++) B[i] += Z[i];
        std::memcpy(D, B, sizeof(B));
        std::memset(Z, 0, sizeof(Z));
    }
    return false;
}

// Horn's absolute orientation: R,T with M_end[i] ~= R*X_i + T for the three correspondences.
static inline bool p3p_align(const double M_end[3][3], const double Xw[3][3], double R[3][3], double T[3]) {
    double C_start[3], C_end[3];
    for (int j = 0; j < 3; j++) {
        C_end[j] = (M_end[0][j] + M_end[1][j] + M_end[2][j]) / 3;
        C_start[j] = (Xw[0][j] + Xw[1][j] + Xw[2][j]) / 3;
    }
    double s[9];
    for (int j = 0; j < 3; j++)
        for (int a = 0; a < 3; a++)
            s[a * 3 + j] = (Xw[0][a] * M_end[0][j] + Xw[1][a] * M_end[1][j] + Xw[2][a] * M_end[2][j]) / 3 - C_end[j] * C_start[a];
    double Qs[16], evs[4], U[16];
    Qs[0 * 4 + 0] = s[0] + s[4] + s[8];
    Qs[1 * 4 + 1] = s[0] - s[4] - s[8];
    Qs[2 * 4 + 2] = s[4] - s[8] - s[0];
    Qs[3 * 4 + 3] = s[8] - s[0] - s[4];
    Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
    Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
    Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
    Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
    Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
    Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];
    jacobi_4x4(Qs, evs, U);
    int i_ev = 0;
    double ev_max = evs[0];
    for (int i = 1; i < 4; i++) if (evs[i] > ev_max) ev_max = evs[i_ev = i];
    double q[4];
    for (int i = 0; i < 4; i++) q[i] = U[i * 4 + i_ev];
    double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
    double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3];
    double q1_2 = q[1] * q[2], q1_3 = q[1] * q[3], q2_3 = q[2] * q[3];
    R[0][0] = q02 + q12 - q22 - q32; R[0][1] = 2. * (q1_2 - q0_3); R[0][2] = 2. * (q1_3 + q0_2);
    R[1][0] = 2. * (q1_2 + q0_3); R[1][1] = q02 + q22 - q12 - q32; R[1][2] = 2. * (q2_3 - q0_1);
    R[2][0] = 2. * (q1_3 - q0_2); R[2][1] = 2. * (q2_3 + q0_1); R[2][2] = q02 + q32 - q12 - q22;
    for (int i = 0; i < 3; i++) T[i] = C_end[i] - (R[i][0] * C_start[0] + R[i][1] * C_start[1] + R[i][2] * C_start[2]);
    return true;
}

// distances = |BC|,|AC|,|AB| ; cosines = cos BPC, cos APC, cos APB ; returns |PA|,|PB|,|PC| per solution.
static inline int p3p_lengths(double lengths[4][3], const double distances[3], const double cosines[3]) {
    double p = cosines[0] * 2, q = cosines[1] * 2, r = cosines[2] * 2;
    double inv_d22 = 1. / (distances[2] * distances[2]);
    double a = inv_d22 * (distances[0] * distances[0]);
    double b = inv_d22 * (distances[1] * distances[1]);
    double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0;  // coplanar with the projection centre
    double ab = a * b, a_2 = 2 * a;
    double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return 0;
    double a_4 = 4 * a;
    double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    double b0 = b * temp * temp;
    if (b0 == 0) return 0;
    double real_roots[4];
    int n = roots_deg4(A, B, C, D, E, real_roots);
    if (n == 0) return 0;
    int nb = 0;
    double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q, inv_b0 = 1. / b0;
    for (int i = 0; i < n; i++) {
        double x = real_roots[i];
        if (x <= 0) continue;
        double x2 = x * x;
        double b1 = ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
                    (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
                      (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
                     (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
                      pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
                     2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
                     p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
        if (b1 <= 0) continue;
        double y = inv_b0 * b1;
        double v = x2 + y * y - x * y * r;
        if (v <= 0) continue;
        double Z = distances[2] / std::sqrt(v);
        lengths[nb][0] = x * Z;
        lengths[nb][1] = y * Z;
        lengths[nb][2] = Z;
        nb++;
    }
    return nb;
}

// solvePnP(objPts(4 x Point3f), imgPts(4 x Point2f), K, no distortion, CV_P3P).  Returns false if no
// real solution; on success rvec = Rodrigues(R).  Image points pass through undistortPoints (-> float
// normalised coordinates) and are mapped back to pixels before the solver normalises them again, so the
// float rounding of the normalised coordinates is part of the published behaviour.
static inline bool solve_p3p(const float* X4, const float* uv4, const Cam& K, double* rvec, double* tvec, double* Rout = nullptr) {
    double mu[4], mv[4];
    for (int i = 0; i < 4; i++) {
        float xn = (float)(((double)uv4[i * 2 + 0] - K.cx) * (1. / K.fx));
        float yn = (float)(((double)uv4[i * 2 + 1] - K.cy) * (1. / K.fy));
        mu[i] = (double)xn * K.fx + K.cx;
        mv[i] = (double)yn * K.fy + K.cy;
    }
    const double inv_fx = 1. / K.fx, inv_fy = 1. / K.fy, cx_fx = K.cx / K.fx, cy_fy = K.cy / K.fy;
    double f[3][3];  // unit bearing vectors of the first three points
    for (int i = 0; i < 3; i++) {
        double u = inv_fx * mu[i] - cx_fx, v = inv_fy * mv[i] - cy_fy;
        double k = 1. / std::sqrt(u * u + v * v + 1);
        f[i][0] = u * k; f[i][1] = v * k; f[i][2] = k;
    }
    double Xw[4][3];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) Xw[i][j] = X4[i * 3 + j];
    auto dist = [&](int a, int b) {
        double dx = Xw[a][0] - Xw[b][0], dy = Xw[a][1] - Xw[b][1], dz = Xw[a][2] - Xw[b][2];
        return std::sqrt(dx * dx + dy * dy + dz * dz);
    };
    auto dot = [&](int a, int b) { return f[a][0] * f[b][0] + f[a][1] * f[b][1] + f[a][2] * f[b][2]; };
    double distances[3] = {dist(1, 2), dist(0, 2), dist(0, 1)};
    double cosines[3] = {dot(1, 2), dot(0, 2), dot(0, 1)};
    double lengths[4][3];
    int n = p3p_lengths(lengths, distances, cosines);
    double Rs[4][3][3], ts[4][3];
    int nb = 0;
    for (int i = 0; i < n; i++) {
        double M[3][3];
        for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) M[k][j] = lengths[i][k] * f[k][j];
        double X3[3][3];
        for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) X3[k][j] = Xw[k][j];
        if (!p3p_align(M, X3, Rs[nb], ts[nb])) continue;
        nb++;
    }
    if (nb == 0) return false;
    int ns = 0;
    double min_reproj = 0;
    for (int i = 0; i < nb; i++) {
        double X3p = Rs[i][0][0] * Xw[3][0] + Rs[i][0][1] * Xw[3][1] + Rs[i][0][2] * Xw[3][2] + ts[i][0];
        double Y3p = Rs[i][1][0] * Xw[3][0] + Rs[i][1][1] * Xw[3][1] + Rs[i][1][2] * Xw[3][2] + ts[i][1];
        double Z3p = Rs[i][2][0] * Xw[3][0] + Rs[i][2][1] * Xw[3][1] + Rs[i][2][2] * Xw[3][2] + ts[i][2];
        double mu3p = K.cx + K.fx * X3p / Z3p, mv3p = K.cy + K.fy * Y3p / Z3p;
        double reproj = (mu3p - mu[3]) * (mu3p - mu[3]) + (mv3p - mv[3]) * (mv3p - mv[3]);
        if (i == 0 || min_reproj > reproj) { ns = i; min_reproj = reproj; }
    }
    double R[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = Rs[ns][i][j];
    if (Rout) std::memcpy(Rout, R, sizeof(R));
    for (int i = 0; i < 3; i++) tvec[i] = ts[ns][i];
    rodrigues_mat2vec(R, rvec);
    return true;
}

// ---------------------------------------------------------------------------------------------
// solvePnP(..., useExtrinsicGuess = true, CV_ITERATIVE): Levenberg-Marquardt from the given pose.
// ---------------------------------------------------------------------------------------------
// Solve the (symmetric) n x n system A x = b by Gaussian elimination with partial pivoting.  OpenCV
// solves the damped normal equations with an SVD pseudo-inverse; for the full-rank 6x6 systems met
// here both give the same x to rounding.
static inline bool solve_linear(const double* A, const double* b, double* x, int n) {
    double M[6 * 7];
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) M[i * (n + 1) + j] = A[i * n + j]; M[i * (n + 1) + n] = b[i]; }
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++) if (std::fabs(M[r * (n + 1) + c]) > std::fabs(M[piv * (n + 1) + c])) piv = r;
        double pv = M[piv * (n + 1) + c];
        if (!(std::fabs(pv) > 0)) { for (int i = 0; i < n; i++) x[i] = 0; return false; }
        if (piv != c) for (int j = 0; j <= n; j++) std::swap(M[c * (n + 1) + j], M[piv * (n + 1) + j]);
        for (int r = c + 1; r < n; r++) {
            double f = M[r * (n + 1) + c] / M[c * (n + 1) + c];
            for (int j = c; j <= n; j++) M[r * (n + 1) + j] -= f * M[c * (n + 1) + j];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = M[i * (n + 1) + n];
        for (int j = i + 1; j < n; j++) s -= M[i * (n + 1) + j] * x[j];
        x[i] = s / M[i * (n + 1) + i];
    }
    return true;
}

struct LMStats { int iters; int evals; double err0; double err; };

static inline bool solve_pnp_iterative_guess(int n, const float* X, const float* uv, const Cam& K, double* rvec, double* tvec,
                                             LMStats* stats = nullptr, int max_iter = 20, double eps = FLT_EPSILON) {
    double param[6] = {rvec[0], rvec[1], rvec[2], tvec[0], tvec[1], tvec[2]};
    double prev[6];
    std::vector<double> proj(2 * n), dpdr(6 * n), dpdt(6 * n), err(2 * n);
    double JtJ[36], JtErr[6];
    int lambdaLg10 = -3, iters = 0, evals = 0;
    double prevErrNorm = DBL_MAX, errNorm = 0;
    auto residual = [&](const double* p, bool withJ) {
        project_points(n, X, p, p + 3, K, nullptr, proj.data(), withJ ? dpdr.data() : nullptr, withJ ? dpdt.data() : nullptr);
        double s = 0;
        for (int i = 0; i < n; i++) {
            err[2 * i] = proj[2 * i] - (double)uv[2 * i];
            err[2 * i + 1] = proj[2 * i + 1] - (double)uv[2 * i + 1];
            s += err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1];
        }
        evals++;
        return std::sqrt(s);
    };
    auto step = [&]() {
        double lambda = std::exp(lambdaLg10 * std::log(10.));
        double A[36], dx[6];
        std::memcpy(A, JtJ, sizeof(A));
        for (int i = 0; i < 6; i++) A[i * 7] *= 1. + lambda;
        solve_linear(A, JtErr, dx, 6);
        for (int i = 0; i < 6; i++) param[i] = prev[i] - dx[i];
    };
    bool done = false;
    // STARTED -> CALC_J
    double e0 = residual(param, true);
    if (stats) stats->err0 = e0;
    while (!done) {
        // CALC_J: normal equations at `param`, first damped step
        std::fill(JtJ, JtJ + 36, 0.0);
        std::fill(JtErr, JtErr + 6, 0.0);
        for (int i = 0; i < n; i++)
            for (int rr = 0; rr < 2; rr++) {
                double Jrow[6];
                for (int j = 0; j < 3; j++) { Jrow[j] = dpdr[i * 6 + rr * 3 + j]; Jrow[3 + j] = dpdt[i * 6 + rr * 3 + j]; }
                double e = err[2 * i + rr];
                for (int a = 0; a < 6; a++) {
                    JtErr[a] += Jrow[a] * e;
                    for (int b2 = 0; b2 < 6; b2++) JtJ[a * 6 + b2] += Jrow[a] * Jrow[b2];
                }
            }
        std::memcpy(prev, param, sizeof(prev));
        if (iters == 0) {
            double s = 0;
            for (int i = 0; i < 2 * n; i++) s += err[i] * err[i];
            prevErrNorm = std::sqrt(s);
        }
        step();
        // CHECK_ERR loop
        for (;;) {
            errNorm = residual(param, false);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) { step(); continue; }
            }
            lambdaLg10 = std::max(lambdaLg10 - 1, -16);
            double num = 0, den = 0;
            for (int i = 0; i < 6; i++) { num += (param[i] - prev[i]) * (param[i] - prev[i]); den += prev[i] * prev[i]; }
            double change = std::sqrt(num) / (std::sqrt(den) + DBL_EPSILON);
            if (++iters >= max_iter || change < eps) { done = true; break; }
            prevErrNorm = errNorm;
            residual(param, true);  // CALC_J at the accepted point
            break;
        }
    }
    for (int i = 0; i < 3; i++) { rvec[i] = param[i]; tvec[i] = param[3 + i]; }
    if (stats) { stats->iters = iters; stats->evals = evals; stats->err = errNorm; }
    return true;
}

}  // namespace cvl
