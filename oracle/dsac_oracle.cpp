// oracle/dsac_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (double precision, OpenMP over hypotheses like the reference) of the DSAC soft-argmax
// hot path of cvlab-dresden/DSAC, function for function, in the reference's conventions.  It is the
// checker for the HIP engine and the timed CPU baseline ("port") of bench.py.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing under dsac_amd/ does.
//
// PARITY: the reference's own build needs OpenCV 2.4, Lua/Torch7 and png++ (none present) and it ships no tests or
// golden vectors, but its hot-path sources compile where they lie against stand-ins for those two libraries
// (oracle/refbuild/ -> oracle/_ref/libdsac_ref.so).  This file -- the restatement of the reference's OWN code,
// citations are /root/reference/core/<file>:<line> -- is PINNED against that build function by function and end to
// end (tests/test_reference_pinning.py; tests/golden/ref_frame_v1.npz carries one frame to machines without the
// reference).  The arithmetic the reference delegates to OpenCV (cvlike.h: Rodrigues, projectPoints, solvePnP) is
// restated from the published algorithms and stays PARITY UNPINNED: both this file and the compiled reference sit
// on it; it is pinned only by closed-form, SciPy and torch-autograd known-answer tests (tests/test_oracle_*.py).
//
// Generalisations w.r.t. the reference (all switchable back):
//   * the scene-coordinate map is H x W (reference: 40 x 40, core/lua_calls.h:33) stored as float32 mm
//     (reference: int16 mm, core/types.h:43-44; pass integer-valued floats to reproduce it);
//   * pixel positions ("sampling") are float32 (u,v) per cell (reference: Point2i, cast to Point2f
//     wherever it is used, core/cnn_softam.h:337,1035);
//   * random minimal sets come from a counter-based generator shared bit-exactly with the HIP engine
//     (reference: per-OpenMP-thread mt19937, core/thread_rand.cpp:40-57, not reproducible);
//   * the hypothesis score may be a soft-inlier count instead of the score CNN (north_star).
#include "cvlike.h"
#include <cstdint>
#include <random>
#include <cstdio>
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif

using cvl::Cam;

#define ORC_EPS 0.00000001     // core/types.h:32
#define ORC_MAXINPUT 100.0     // core/lua_calls.h:36  CNN_OBJ_MAXINPUT
#define ORC_MAXLOSS 10000000.0 // core/maxloss.h:30

namespace {

struct Frame {
    const float* xyz;  // P x 3, mm
    const float* uv;   // P x 2, pixel position of each cell in the original frame
    int H, W;
    Cam K;
    int P() const { return H * W; }
};

// ---- core/types.h:186-214  jp::cv2our ------------------------------------------------------------
void cv2our(const double* cv6, double* R, double* t) {
    cvl::rodrigues_vec2mat(cv6, R, nullptr);
    t[0] = cv6[3]; t[1] = cv6[4]; t[2] = cv6[5];
    for (int j = 0; j < 3; j++) { R[3 + j] = -R[3 + j]; R[6 + j] = -R[6 + j]; }
    t[1] = -t[1]; t[2] = -t[2];
    if (cvl::mat3_det(R) < 0) {
        for (int j = 0; j < 9; j++) R[j] = -R[j];
        for (int j = 0; j < 3; j++) t[j] = -t[j];
    }
    if (cvl::has_nan(t, 3)) t[0] = t[1] = t[2] = 0;
}

// ---- core/types.h:137-151  jp::our2cv ------------------------------------------------------------
void our2cv(const double* R, const double* t, double* cv6) {
    double Rm[9];
    std::memcpy(Rm, R, sizeof(Rm));
    for (int j = 0; j < 3; j++) { Rm[3 + j] = -Rm[3 + j]; Rm[6 + j] = -Rm[6 + j]; }
    cvl::rodrigues_mat2vec(Rm, cv6);
    cv6[3] = t[0]; cv6[4] = -t[1]; cv6[5] = -t[2];
}

// ---- core/Hypothesis.cpp:274-289  Hypothesis::getRodVecAndTrans -----------------------------------
void rodvec_and_trans(const double* R, const double* t, double* out6) {
    cvl::rodrigues_mat2vec(R, out6);
    out6[3] = t[0]; out6[4] = t[1]; out6[5] = t[2];
}

// jp 6-vector of a cv pose: getRodVecAndTrans(Hypothesis(cv2our(.)))  (core/cnn_softam.h:121-122,721-722)
void cv_to_jp6(const double* cv6, double* jp6) {
    double R[9], t[3];
    cv2our(cv6, R, t);
    rodvec_and_trans(R, t, jp6);
}

// ---- core/cnn_softam.h:56-73  safeSolvePnP (CV_P3P) ------------------------------------------------
bool safe_p3p(const float* X4, const float* uv4, const Cam& K, double* cv6) {
    if (!cvl::solve_p3p(X4, uv4, K, cv6, cv6 + 3)) {
        for (int i = 0; i < 6; i++) cv6[i] = 0;
        return false;
    }
    return true;
}

// ---- core/cnn_softam.h:319-362  getDiffMap ---------------------------------------------------------
void get_diff_map(const double* cv6, const Frame& F, float* out) {
    const int P = F.P();
    std::vector<float> proj(2 * (size_t)P);
    cvl::project_points(P, F.xyz, cv6, cv6 + 3, F.K, proj.data());
    for (int p = 0; p < P; p++) {
        float dx = F.uv[2 * p] - proj[2 * p];        // Point2f - Point2f
        float dy = F.uv[2 * p + 1] - proj[2 * p + 1];
        double nrm = std::sqrt((double)dx * dx + (double)dy * dy);  // cv::norm(Point2f) is double
        out[p] = (float)std::min(nrm, ORC_MAXINPUT);
    }
}

// Same residual for one cell (used by refine's lazy walk; identical arithmetic).
float diff_at(const double* R, const double* t, const Frame& F, const float* xyz_p, int p) {
    double Mx = xyz_p[0], My = xyz_p[1], Mz = xyz_p[2];
    double Xc = R[0] * Mx + R[1] * My + R[2] * Mz + t[0];
    double Yc = R[3] * Mx + R[4] * My + R[5] * Mz + t[1];
    double Zc = R[6] * Mx + R[7] * My + R[8] * Mz + t[2];
    double z = Zc ? 1. / Zc : 1.;
    float u = (float)(Xc * z * F.K.fx + F.K.cx), v = (float)(Yc * z * F.K.fy + F.K.cy);
    float dx = F.uv[2 * p] - u, dy = F.uv[2 * p + 1] - v;
    return (float)std::min(std::sqrt((double)dx * dx + (double)dy * dy), ORC_MAXINPUT);
}

// ---- core/cnn_softam.h:373-393  project (jp convention) -------------------------------------------
float project_jp(const float* pt, const float* obj, const double* R, const double* t, const Cam& K) {
    double f = K.fx, ppx = K.cx, ppy = K.cy;
    double ex = R[0] * obj[0] + R[1] * obj[1] + R[2] * obj[2] + t[0];
    double ey = R[3] * obj[0] + R[4] * obj[1] + R[5] * obj[2] + t[1];
    double ez = R[6] * obj[0] + R[7] * obj[1] + R[8] * obj[2] + t[2];
    double px = -f * ex / ez + ppx;
    double py = f * ey / ez + ppy;
    return (float)std::min(std::sqrt((pt[0] - px) * (pt[0] - px) + (pt[1] - py) * (pt[1] - py)), ORC_MAXINPUT);
}

// ---- core/cnn_softam.h:404-453  dProjectdObj -------------------------------------------------------
void d_project_d_obj(const float* pt, const float* obj, const double* R, const double* t, const Cam& K, double* J3) {
    double f = K.fx, ppx = K.cx, ppy = K.cy;
    J3[0] = J3[1] = J3[2] = 0;
    double ex = R[0] * obj[0] + R[1] * obj[1] + R[2] * obj[2] + t[0];
    double ey = R[3] * obj[0] + R[4] * obj[1] + R[5] * obj[2] + t[1];
    double ez = R[6] * obj[0] + R[7] * obj[1] + R[8] * obj[2] + t[2];
    if (std::fabs(ez) < ORC_EPS) return;
    double px = -f * ex / ez + ppx;
    double py = f * ey / ez + ppy;
    double err = std::sqrt((pt[0] - px) * (pt[0] - px) + (pt[1] - py) * (pt[1] - py));
    if (err > ORC_MAXINPUT) return;
    err += ORC_EPS;
    for (int c = 0; c < 3; c++) {
        double pxd = -f * R[0 * 3 + c] / ez + f * ex / ez / ez * R[2 * 3 + c];
        double pyd = f * R[1 * 3 + c] / ez - f * ey / ez / ez * R[2 * 3 + c];
        J3[c] = 0.5 / err * (2 * (pt[0] - px) * -pxd + 2 * (pt[1] - py) * -pyd);
    }
}

// ---- core/cnn_softam.h:464-528  dProjectdHyp -------------------------------------------------------
// NB the reference re-orthonormalises `rot` in place through Rodrigues(rot)->rod->rot (quirk 7); the
// Jacobian dRdH is that of the re-derived Rodrigues vector.
void d_project_d_hyp(const float* pt, const float* obj, const double* R, const double* t, const Cam& K, double* J6, double* R_writeback = nullptr) {
    double f = K.fx, ppx = K.cx, ppy = K.cy;
    for (int i = 0; i < 6; i++) J6[i] = 0;
    double ox = obj[0], oy = obj[1], oz = obj[2];
    double ex = R[0] * ox + R[1] * oy + R[2] * oz + t[0];
    double ey = R[3] * ox + R[4] * oy + R[5] * oz + t[1];
    double ez = R[6] * ox + R[7] * oy + R[8] * oz + t[2];
    if (std::fabs(ez) < ORC_EPS) return;
    double px = -f * ex / ez + ppx;
    double py = f * ey / ez + ppy;
    double err = std::sqrt((pt[0] - px) * (pt[0] - px) + (pt[1] - py) * (pt[1] - py));
    if (err > ORC_MAXINPUT) return;
    err += ORC_EPS;
    double dNdP[2] = {-1 / err * (pt[0] - px), -1 / err * (pt[1] - py)};
    double dPdR[2][9] = {{0}};
    const double o[3] = {ox, oy, oz};
    for (int k = 0; k < 3; k++) {
        dPdR[0][k] = -f * o[k] / ez;
        dPdR[1][3 + k] = f * o[k] / ez;
        dPdR[0][6 + k] = f * ex / ez / ez * o[k];
        dPdR[1][6 + k] = -f * ey / ez / ez * o[k];
    }
    double rod[3], Rre[9], dRdH[27];
    cvl::rodrigues_mat2vec(R, rod);
    cvl::rodrigues_vec2mat(rod, Rre, dRdH);  // 3x9, used transposed (9x3)
    if (R_writeback) std::memcpy(R_writeback, Rre, sizeof(Rre));  // quirk 7: cnn_softam.h:508 writes through `const cv::Mat& rot`
    double dNdR[9];
    for (int k = 0; k < 9; k++) dNdR[k] = dNdP[0] * dPdR[0][k] + dNdP[1] * dPdR[1][k];
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 9; k++) s += dNdR[k] * dRdH[i * 9 + k];
        J6[i] = s;
    }
    double dPdT[2][3] = {{-f / ez, 0, f * ex / ez / ez}, {0, f / ez, -f * ey / ez / ez}};
    for (int k = 0; k < 3; k++) J6[3 + k] = dNdP[0] * dPdT[0][k] + dNdP[1] * dPdT[1][k];
}

// ---- core/cnn_softam.h:535-553  softMax ; :80-88 entropy -------------------------------------------
void soft_max(int n, const double* s, double* w) {
    double m = 0;
    for (int i = 0; i < n; i++) if (i == 0 || s[i] > m) m = s[i];
    double sum = 0;
    for (int i = 0; i < n; i++) { w[i] = std::exp(s[i] - m); sum += w[i]; }
    for (int i = 0; i < n; i++) w[i] /= sum;
}
double entropy(int n, const double* w) {
    double e = 0;
    for (int i = 0; i < n; i++) if (w[i] > 0) e -= w[i] * std::log2(w[i]);
    return e;
}

// ---- core/maxloss.h:39-61 getInvHyp ; :69-79 maxLoss ; core/Hypothesis.cpp:137-143 -----------------
void inv_hyp(const double* R, const double* t, double* Ri, double* ti) {
    double T[16] = {R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2], 0, 0, 0, 1}, Ti[16];
    cvl::mat_inv(T, Ti, 4);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Ri[i * 3 + j] = Ti[i * 4 + j]; ti[i] = Ti[i * 4 + 3]; }
}
double angular_distance(const double* Ra, const double* Rb) {  // this->R * h.invR
    double Rbi[9], D[9];
    cvl::mat_inv(Rb, Rbi, 3);
    cvl::mat3_mul(Ra, Rbi, D);
    double tr = D[0] + D[4] + D[8];
    tr = std::min(3.0, std::max(-1.0, tr));
    return 180 * std::acos((tr - 1.0) / 2.0) / M_PI;
}
void pose_errors(const double* R1, const double* t1, const double* R2, const double* t2, double* rotErr, double* tErr) {
    double Ri1[9], ti1[3], Ri2[9], ti2[3];
    inv_hyp(R1, t1, Ri1, ti1);
    inv_hyp(R2, t2, Ri2, ti2);
    *rotErr = angular_distance(Ri1, Ri2);
    double d[3] = {ti1[0] - ti2[0], ti1[1] - ti2[1], ti1[2] - ti2[2]};
    *tErr = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}
double max_loss(const double* R1, const double* t1, const double* R2, const double* t2) {
    double rotErr, tErr;
    pose_errors(R1, t1, R2, t2, &rotErr, &tErr);
    return std::min(std::max(rotErr, tErr / 10), ORC_MAXLOSS);
}

// ---- core/maxloss.h:87-198  dLossMax ---------------------------------------------------------------
void d_loss_max(const double* est, const double* gt, double* J6) {
    for (int i = 0; i < 6; i++) J6[i] = 0;
    double rot1[9], rot2[9], dRod[27];
    cvl::rodrigues_vec2mat(est, rot1, dRod);
    cvl::rodrigues_vec2mat(gt, rot2, nullptr);
    double invRot1[9], invRot2[9], diffRot[9];
    cvl::mat3_t(rot1, invRot1);
    cvl::mat3_t(rot2, invRot2);
    cvl::mat3_mul(rot1, invRot2, diffRot);
    double trace = diffRot[0] + diffRot[4] + diffRot[8];
    trace = std::min(3.0, std::max(-1.0, trace));
    double rotErr = 180 * std::acos((trace - 1.0) / 2.0) / M_PI;
    double a1[3] = {-est[3] / 10, -est[4] / 10, -est[5] / 10}, a2[3] = {-gt[3] / 10, -gt[4] / 10, -gt[5] / 10};
    double invT1[3], invT2[3];
    for (int i = 0; i < 3; i++) {
        invT1[i] = invRot1[i * 3] * a1[0] + invRot1[i * 3 + 1] * a1[1] + invRot1[i * 3 + 2] * a1[2];
        invT2[i] = invRot2[i * 3] * a2[0] + invRot2[i * 3 + 1] * a2[1] + invRot2[i * 3 + 2] * a2[2];
    }
    double dT[3] = {invT1[0] - invT2[0], invT1[1] - invT2[1], invT1[2] - invT2[2]};
    double tErr = std::sqrt(dT[0] * dT[0] + dT[1] * dT[1] + dT[2] * dT[2]);
    if (std::max(rotErr, tErr) > ORC_MAXLOSS) return;
    if ((tErr + rotErr) < ORC_EPS) return;
    if (tErr > rotErr) {
        double dDist[3] = {dT[0] / tErr, dT[1] / tErr, dT[2] / tErr};
        // dInvT1/dEstT = -invRot1  ->  J[3:6] = dDist * (-invRot1)
        for (int c = 0; c < 3; c++) J6[3 + c] = -(dDist[0] * invRot1[0 * 3 + c] + dDist[1] * invRot1[1 * 3 + c] + dDist[2] * invRot1[2 * 3 + c]);
        // dInvT1/dInvRot1 (3x9): D(r, r + 3c) = -est[3+c]/10     (core/maxloss.h:147-158)
        double D[3][9] = {{0}};
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) D[r][r + 3 * c] = -est[3 + c] / 10;
        double v9[9];
        for (int k = 0; k < 9; k++) v9[k] = dDist[0] * D[0][k] + dDist[1] * D[1][k] + dDist[2] * D[2][k];
        for (int i = 0; i < 3; i++) {  // * dRod^T (9x3)
            double s = 0;
            for (int k = 0; k < 9; k++) s += v9[k] * dRod[i * 9 + k];
            J6[i] = s;
        }
    } else {
        // dRotDiff (9x9): block-diagonal with invRot2 in each 3x3 block, then transposed (core/maxloss.h:170-183)
        double M[81] = {0};
        for (int b = 0; b < 3; b++)
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) M[(b * 3 + r) * 9 + (b * 3 + c)] = invRot2[r * 3 + c];
        double Mt[81];
        for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) Mt[j * 9 + i] = M[i * 9 + j];
        double dTrace[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, v9[9];
        for (int k = 0; k < 9; k++) {
            double s = 0;
            for (int j = 0; j < 9; j++) s += dTrace[j] * Mt[j * 9 + k];
            v9[k] = s;
        }
        double scale = 180 / M_PI * -1 / std::sqrt(3 - trace * trace + 2 * trace);
        for (int i = 0; i < 3; i++) {
            double s = 0;
            for (int k = 0; k < 9; k++) s += v9[k] * dRod[i * 9 + k];
            J6[i] = scale * s;
        }
    }
    if (cvl::has_nan(J6, 6)) for (int i = 0; i < 6; i++) J6[i] = 0;
}

// ---- core/cnn_softam.h:101-146  dPNP (4 points, CV_P3P) --------------------------------------------
// objPts are floats and are perturbed *in float*, sequentially (+eps, -2eps, +eps), as in the reference.
__attribute__((noinline)) void d_pnp(const float* uv4, const float* X4_in, float eps, const Cam& K, double* J /*6x12*/) {
    float X4[12];
    std::memcpy(X4, X4_in, sizeof(X4));
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) {
            double cv6[6], f6[6], b6[6];
            X4[i * 3 + j] += eps;
            safe_p3p(X4, uv4, K, cv6);
            cv_to_jp6(cv6, f6);
            X4[i * 3 + j] -= 2 * eps;
            safe_p3p(X4, uv4, K, cv6);
            cv_to_jp6(cv6, b6);
            X4[i * 3 + j] += eps;
            bool nan = false;
            for (int k = 0; k < 6; k++) {
                double v = (f6[k] - b6[k]) / (2 * eps);  // 2*eps is float, promoted
                J[k * 12 + i * 3 + j] = v;
                if (v != v) nan = true;
            }
            if (nan) { for (int k = 0; k < 72; k++) J[k] = 0; return; }
        }
}

// ---- counter-based generator shared with the HIP engine (include/dsac_hip.h "Sampling RNG") --------
inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// candidate cell k of an attempt: one 64-bit draw, x from its high half, y from its low half
inline int draw_cell(uint64_t key, uint32_t attempt, uint32_t k, uint32_t W, uint32_t H) {
    uint64_t v = mix64(key + (((uint64_t)attempt << 16) | k));
    uint32_t x = (uint32_t)(((v >> 32) * (uint64_t)W) >> 32);
    uint32_t y = (uint32_t)(((v & 0xffffffffull) * (uint64_t)H) >> 32);
    return (int)(y * W + x);
}
inline uint64_t hyp_key(uint64_t seed, uint32_t hyp) { return mix64(seed ^ mix64((uint64_t)hyp)); }

// One attempt of the sampling loop body, core/cnn_softam.h:1014-1059.  Returns true if accepted.
bool sample_attempt(const Frame& F, uint64_t key, uint32_t attempt, int thr_int, int32_t* set4, double* cv6) {
    uint32_t k = 0;
    int cnt = 0;
    while (cnt < 4) {
        if (k >= 32) return false;  // > 32 candidate cells: give up on this attempt (degenerate tiny maps)
        int idx = draw_cell(key, attempt, k++, (uint32_t)F.W, (uint32_t)F.H);   // (x, y) of a candidate: core/cnn_softam.h:1024-1025
        bool dup = false;
        for (int j = 0; j < cnt; j++) if (set4[j] == idx) dup = true;
        if (dup) continue;
        set4[cnt++] = idx;
    }
    float X4[12], uv4[8];
    for (int j = 0; j < 4; j++) {
        for (int c = 0; c < 3; c++) X4[j * 3 + c] = F.xyz[(size_t)set4[j] * 3 + c];
        uv4[j * 2] = F.uv[(size_t)set4[j] * 2];
        uv4[j * 2 + 1] = F.uv[(size_t)set4[j] * 2 + 1];
    }
    if (!safe_p3p(X4, uv4, F.K, cv6)) return false;
    float proj[8];
    cvl::project_points(4, X4, cv6, cv6 + 3, F.K, proj);
    for (int j = 0; j < 4; j++) {
        float dx = uv4[j * 2] - proj[j * 2], dy = uv4[j * 2 + 1] - proj[j * 2 + 1];
        if (!(std::sqrt((double)dx * dx + (double)dy * dy) < thr_int)) return false;
    }
    return true;
}

// Evaluate a given minimal set (no RNG): P3P + 4-point check.
bool eval_set(const Frame& F, const int32_t* set4, int thr_int, double* cv6) {
    float X4[12], uv4[8];
    for (int j = 0; j < 4; j++) {
        for (int c = 0; c < 3; c++) X4[j * 3 + c] = F.xyz[(size_t)set4[j] * 3 + c];
        uv4[j * 2] = F.uv[(size_t)set4[j] * 2];
        uv4[j * 2 + 1] = F.uv[(size_t)set4[j] * 2 + 1];
    }
    if (!safe_p3p(X4, uv4, F.K, cv6)) return false;
    float proj[8];
    cvl::project_points(4, X4, cv6, cv6 + 3, F.K, proj);
    for (int j = 0; j < 4; j++) {
        float dx = uv4[j * 2] - proj[j * 2], dy = uv4[j * 2 + 1] - proj[j * 2 + 1];
        if (!(std::sqrt((double)dx * dx + (double)dy * dy) < thr_int)) return false;
    }
    return true;
}

// ---- refinement: core/cnn_softam.h:1099-1154 (forward, fills inlierMap) and :663-723 (replay) -------
// `perm` = refSteps x P pixel indices (the reference's pixelIdxs).  `pert_px >= 0` replaces channel
// pert_c of that pixel by pert_value -- that is dRefineObj's localEstObj (:887,901).
// Returns the cv pose in out_cv6; optional inlier_map (P ints, += 1 per selection; forward only).
void refine_cv(const Frame& F, const int32_t* perm, int refSteps, int inlierCount, int minInliers, float thr,
               const double* init_cv6, int pert_px, int pert_c, float pert_value, double* out_cv6, int32_t* inlier_map,
               int* steps_done = nullptr) {
    const int P = F.P();
    double hyp[6];
    std::memcpy(hyp, init_cv6, sizeof(hyp));
    std::vector<float> X, UV;
    auto xyz_of = [&](int p, float* o) {
        o[0] = F.xyz[(size_t)p * 3]; o[1] = F.xyz[(size_t)p * 3 + 1]; o[2] = F.xyz[(size_t)p * 3 + 2];
        if (p == pert_px) o[pert_c] = pert_value;
    };
    int done = 0;
    for (int rStep = 0; rStep < refSteps; rStep++) {
        double R[9];
        cvl::rodrigues_vec2mat(hyp, R, nullptr);
        X.clear(); UV.clear();
        const int32_t* pidx = perm + (size_t)rStep * P;
        for (int idx = 0; idx < P; idx++) {
            int p = pidx[idx];
            float o[3];
            xyz_of(p, o);
            if (diff_at(R, hyp + 3, F, o, p) < thr) {
                UV.push_back(F.uv[2 * (size_t)p]); UV.push_back(F.uv[2 * (size_t)p + 1]);
                X.push_back(o[0]); X.push_back(o[1]); X.push_back(o[2]);
                if (inlier_map) inlier_map[p] += 1;
            }
            if ((int)(UV.size() / 2) >= inlierCount) break;
        }
        int n = (int)(UV.size() / 2);
        if (n < minInliers) break;
        double upd[6];
        std::memcpy(upd, hyp, sizeof(upd));
        cvl::solve_pnp_iterative_guess(n, X.data(), UV.data(), F.K, upd, upd + 3);
        if (cvl::has_nan(upd, 6)) break;
        std::memcpy(hyp, upd, sizeof(hyp));
        done++;
    }
    std::memcpy(out_cv6, hyp, sizeof(hyp));
    if (steps_done) *steps_done = done;
}

}  // namespace

// =================================================================================================
// C exports (ctypes).  All poses are 6 doubles (rvec, tvec[mm]); cam = {fx, fy, cx, cy} doubles.
// =================================================================================================
extern "C" {

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

void orc_rodrigues_vec2mat(const double* r, double* R, double* J27_or_null) { cvl::rodrigues_vec2mat(r, R, J27_or_null); }
void orc_rodrigues_mat2vec(const double* R, double* r) { cvl::rodrigues_mat2vec(R, r); }
void orc_project_points(int n, const float* X, const double* pose6, const double* cam, float* uv) {
    Cam K{cam[0], cam[1], cam[2], cam[3]};
    cvl::project_points(n, X, pose6, pose6 + 3, K, uv);
}
void orc_project_points_jac(int n, const float* X, const double* pose6, const double* cam, double* uv_d, double* dpdr, double* dpdt) {
    Cam K{cam[0], cam[1], cam[2], cam[3]};
    cvl::project_points(n, X, pose6, pose6 + 3, K, nullptr, uv_d, dpdr, dpdt);
}
int orc_roots_deg4(double a, double b, double c, double d, double e, double* x) { return cvl::roots_deg4(a, b, c, d, e, x); }
int orc_p3p_lengths(const double* distances, const double* cosines, double* lengths12) {
    double L[4][3];
    int n = cvl::p3p_lengths(L, distances, cosines);
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) lengths12[i * 3 + j] = L[i][j];
    return n;
}
int orc_solve_p3p(const float* X4, const float* uv4, const double* cam, double* pose6) {
    Cam K{cam[0], cam[1], cam[2], cam[3]};
    return safe_p3p(X4, uv4, K, pose6) ? 1 : 0;
}
int orc_solve_pnp_iterative(int n, const float* X, const float* uv, const double* cam, double* pose6, int* iters, double* err) {
    Cam K{cam[0], cam[1], cam[2], cam[3]};
    cvl::LMStats st{};
    cvl::solve_pnp_iterative_guess(n, X, uv, K, pose6, pose6 + 3, &st);
    if (iters) *iters = st.iters;
    if (err) { err[0] = st.err0; err[1] = st.err; }
    return 1;
}

void orc_cv2our(const double* cv6, double* R, double* t) { cv2our(cv6, R, t); }
void orc_our2cv(const double* R, const double* t, double* cv6) { our2cv(R, t, cv6); }
void orc_rodvec_and_trans(const double* R, const double* t, double* out6) { rodvec_and_trans(R, t, out6); }
void orc_cv_to_jp6(const double* cv6, double* jp6) { cv_to_jp6(cv6, jp6); }

// N error images, OpenMP over hypotheses (core/cnn_softam.h:1067-1069)
void orc_get_diff_maps(int N, const double* poses, const float* xyz, const float* uv, int H, int W, const double* cam, float* out) {
    Frame F{xyz, uv, H, W, Cam{cam[0], cam[1], cam[2], cam[3]}};
    const size_t P = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < N; h++) get_diff_map(poses + 6 * h, F, out + h * P);
}

// soft-inlier score per hypothesis: sum_p sigmoid(beta*(tau - e_p)) with e from getDiffMap (float),
// accumulated in double.  (north_star's DSAC++-style score; not in the reference.)
void orc_soft_inlier(int N, const float* err, int P, float tau, float beta, double* score) {
#pragma omp parallel for
    for (int h = 0; h < N; h++) {
        double s = 0;
        for (int p = 0; p < P; p++) s += 1.0 / (1.0 + std::exp(-(double)beta * ((double)tau - (double)err[(size_t)h * P + p])));
        score[h] = s;
    }
}

float orc_project(const float* pt, const float* obj, const double* R, const double* t, const double* cam) {
    return project_jp(pt, obj, R, t, Cam{cam[0], cam[1], cam[2], cam[3]});
}
void orc_dProjectdObj(const float* pt, const float* obj, const double* R, const double* t, const double* cam, double* J3) {
    d_project_d_obj(pt, obj, R, t, Cam{cam[0], cam[1], cam[2], cam[3]}, J3);
}
void orc_dProjectdHyp(const float* pt, const float* obj, const double* R, const double* t, const double* cam, double* J6) {
    d_project_d_hyp(pt, obj, R, t, Cam{cam[0], cam[1], cam[2], cam[3]}, J6);
}
void orc_softMax(int n, const double* s, double* w) { soft_max(n, s, w); }
double orc_entropy(int n, const double* w) { return entropy(n, w); }

// soft-argmax average, core/cnn_softam.h:1082-1094
void orc_avg_pose(int n, const double* w, const double* poses, double* avg6) {
    for (int k = 0; k < 6; k++) avg6[k] = 0;
    for (int h = 0; h < n; h++) for (int k = 0; k < 6; k++) avg6[k] += w[h] * poses[6 * h + k];
}

double orc_maxLoss(const double* R1, const double* t1, const double* R2, const double* t2) { return max_loss(R1, t1, R2, t2); }
void orc_pose_errors(const double* R1, const double* t1, const double* R2, const double* t2, double* rotErr, double* tErr) {
    pose_errors(R1, t1, R2, t2, rotErr, tErr);
}
void orc_dLossMax(const double* est6, const double* gt6, double* J6) { d_loss_max(est6, gt6, J6); }

void orc_dPNP(const float* uv4, const float* X4, float eps, const double* cam, double* J72) {
    d_pnp(uv4, X4, eps, Cam{cam[0], cam[1], cam[2], cam[3]}, J72);
}

// Sampling loop, core/cnn_softam.h:1010-1060.  sets_in == NULL: counter RNG, first accepted attempt
// < max_tries wins.  sets_in != NULL: evaluate the given sets once (ok = accepted).
void orc_sample(int N, uint64_t seed, const int32_t* sets_in, const float* xyz, const float* uv, int H, int W, const double* cam,
                float thr, int max_tries, double* poses, int32_t* sets_out, uint8_t* ok, int32_t* tries) {
    Frame F{xyz, uv, H, W, Cam{cam[0], cam[1], cam[2], cam[3]}};
    const int thr_int = (int)thr;  // quirk 5: threshold truncated to int
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < N; h++) {
        double cv6[6] = {0, 0, 0, 0, 0, 0};
        int32_t set4[4] = {0, 0, 0, 0};
        bool good = false;
        int a = 0;
        if (sets_in) {
            std::memcpy(set4, sets_in + 4 * h, sizeof(set4));
            good = eval_set(F, set4, thr_int, cv6);
            a = 1;
        } else {
            uint64_t key = hyp_key(seed, (uint32_t)h);
            for (a = 0; a < max_tries; a++) {
                if (sample_attempt(F, key, (uint32_t)a, thr_int, set4, cv6)) { good = true; a++; break; }
            }
        }
        if (!good) for (int k = 0; k < 6; k++) cv6[k] = 0;
        std::memcpy(poses + 6 * h, cv6, sizeof(cv6));
        std::memcpy(sets_out + 4 * h, set4, sizeof(set4));
        ok[h] = good ? 1 : 0;
        if (tries) tries[h] = a;
    }
}

}  // extern "C"
struct CountingMt {  // std::mt19937 that counts its outputs (same result_type, min and max: the distribution takes the same path through it)
    typedef std::mt19937::result_type result_type;
    std::mt19937 g;
    uint64_t n = 0;
    static constexpr result_type min() { return std::mt19937::min(); }
    static constexpr result_type max() { return std::mt19937::max(); }
    result_type operator()() { n++; return g(); }
};
extern "C" {
// The same loop in the REFERENCE'S OWN random stream (round 6): ThreadRand (core/thread_rand.cpp:40-69) keeps one std::mt19937(seed + t) per OpenMP thread and
// irand(0, n) is std::uniform_int_distribution<int>(0, n - 1) on it (:59-69, :95-98); the sampling loop (core/cnn_softam.h:1010-1060) draws x before y,
// re-draws a cell that is already in the set, and starts over after a failed P3P or re-projection check -- with no cap.  `#pragma omp parallel for` with
// the default static schedule hands thread t the hypotheses [t q + min(t, r), ...) (q = N / T, r = N % T, the first r threads one more), which it serves
// one after the other from its stream.  This restatement uses the standard library's generator and distribution themselves (whatever libstdc++ this
// checker is built with -- the same one oracle/_ref is built with), so it pins the product's own restatement of them (dsac_amd/csrc/refstream.h).
// skip32[t]: 32-bit outputs generator t has already produced (e.g. the drand calls of stochasticSubSample on thread 0: two each); consumed32[t] (out):
// outputs this call took from it.  max_attempts caps the attempts per stream (the reference has no cap); hypotheses it leaves unserved get ok = 0.
void orc_sample_refstream(int N, uint32_t seed, int T, const uint64_t* skip32, const float* xyz, const float* uv, int H, int W, const double* cam, float thr,
                          long long max_attempts, double* poses, int32_t* sets_out, uint8_t* ok, uint64_t* consumed32, int64_t* attempts_out) {
    Frame F{xyz, uv, H, W, Cam{cam[0], cam[1], cam[2], cam[3]}};
    const int thr_int = (int)thr;
    const int q = N / T, r = N % T;
    for (int t = 0; t < T; t++) {
        CountingMt gen;
        gen.g.seed(seed + (uint32_t)t);
        if (skip32 && skip32[t]) gen.g.discard(skip32[t]);
        const int h0 = t * q + std::min(t, r), nh = q + (t < r ? 1 : 0);
        long long attempts = 0;
        for (int h = h0; h < h0 + nh; h++) {
            double cv6[6] = {0, 0, 0, 0, 0, 0};
            int32_t set4[4] = {0, 0, 0, 0};
            bool good = false;
            while (!good && attempts < max_attempts) {
                attempts++;
                int cnt = 0;
                while (cnt < 4) {
                    const int x = std::uniform_int_distribution<int>(0, W - 1)(gen);
                    const int y = std::uniform_int_distribution<int>(0, H - 1)(gen);
                    const int idx = y * W + x;
                    bool dup = false;
                    for (int k = 0; k < cnt; k++) dup = dup || set4[k] == idx;
                    if (dup) continue;
                    set4[cnt++] = idx;
                }
                good = eval_set(F, set4, thr_int, cv6);
            }
            if (!good) for (int k = 0; k < 6; k++) cv6[k] = 0;
            std::memcpy(poses + 6 * h, cv6, sizeof(cv6));
            std::memcpy(sets_out + 4 * h, set4, sizeof(set4));
            ok[h] = good ? 1 : 0;
        }
        if (consumed32) consumed32[t] = gen.n;
        if (attempts_out) attempts_out[t] = attempts;
    }
}

// B independent refinements (forward form fills inlier_map of problem 0 only when B == 1).
// pert_px_c: B x 2 ints (pixel or -1, channel); pert_value: B floats (replacement value); may be NULL.
void orc_refine(int B, const double* init_poses, const int32_t* perm, int refSteps, int inlierCount, int minInliers, float thr,
                const float* xyz, const float* uv, int H, int W, const double* cam, const int32_t* pert_px_c, const float* pert_value,
                double* out_poses, int32_t* inlier_map_or_null, int32_t* steps_done_or_null) {
    Frame F{xyz, uv, H, W, Cam{cam[0], cam[1], cam[2], cam[3]}};
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; b++) {
        int px = pert_px_c ? pert_px_c[2 * b] : -1, c = pert_px_c ? pert_px_c[2 * b + 1] : 0;
        float d = pert_value ? pert_value[b] : 0.f;
        int sd = 0;
        refine_cv(F, perm, refSteps, inlierCount, minInliers, thr, init_poses + 6 * b, px, c, d, out_poses + 6 * b,
                  (B == 1) ? inlier_map_or_null : nullptr, &sd);
        if (steps_done_or_null) steps_done_or_null[b] = sd;
    }
}

// ---- core/cnn_softam.h:738-836  dRefineHyp (6x6) ---------------------------------------------------
void orc_dRefineHyp(const double* init_cv6, const int32_t* perm, int refSteps, int inlierCount, int minInliers, float thr,
                    const float* xyz, const float* uv, int H, int W, const double* cam, float eps, double* J36) {
    Frame F{xyz, uv, H, W, Cam{cam[0], cam[1], cam[2], cam[3]}};
    for (int i = 0; i < 6; i++) {
        double loc[6], f6[6], b6[6], o[6];
        std::memcpy(loc, init_cv6, sizeof(loc));
        double step = (i < 3) ? (double)eps : (double)(eps * 1000);  // float eps, float product (core/cnn_softam.h:758,798)
        loc[i] += step;
        refine_cv(F, perm, refSteps, inlierCount, minInliers, thr, loc, -1, 0, 0.f, o, nullptr);
        cv_to_jp6(o, f6);
        loc[i] -= 2 * step;
        refine_cv(F, perm, refSteps, inlierCount, minInliers, thr, loc, -1, 0, 0.f, o, nullptr);
        cv_to_jp6(o, b6);
        for (int k = 0; k < 3; k++) J36[k * 6 + i] = (f6[k] - b6[k]) / (2 * eps);
        for (int k = 3; k < 6; k++) J36[k * 6 + i] = (f6[k] - b6[k]) / (2 * eps * 1000);
    }
}

// ---- core/cnn_softam.h:853-923  dRefineObj (6 x 3P) -------------------------------------------------
// Column index y*W*3 + x*3 + c (reference: y*CNN_OBJ_PATCHSIZE*3 + x*3 + c with W = 40).
void orc_dRefineObj(const double* init_cv6, const int32_t* perm, int refSteps, int inlierCount, int minInliers, float thr,
                    float subSampleFactor, const int32_t* inlier_map, const float* xyz, const float* uv, int H, int W, const double* cam,
                    float eps, double* J /*6 x 3P, zero-filled here*/) {
    Frame F{xyz, uv, H, W, Cam{cam[0], cam[1], cam[2], cam[3]}};
    const size_t P = (size_t)H * W;
    std::fill(J, J + 6 * 3 * P, 0.0);
    int skip = (int)(1 / subSampleFactor);
    std::vector<int> todo;  // pixel indices processed, in the reference's x-outer / y-inner order
    int inCount = 0;
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
            if (inlier_map[y * W + x] == 0) continue;
            inCount++;
            if (inCount % skip != 0) continue;
            todo.push_back(y * W + x);
        }
#pragma omp parallel for schedule(dynamic, 1)
    for (int ti = 0; ti < (int)todo.size() * 3; ti++) {
        int p = todo[ti / 3], c = ti % 3;
        double o[6], f6[6], b6[6];
        // localEstObj(y,x)[c] += eps ; ... -= 2*eps : sequential float adds on the stored value
        float v0 = F.xyz[(size_t)p * 3 + c];
        float vf = v0 + eps;
        float vb = vf - 2 * eps;
        refine_cv(F, perm, refSteps, inlierCount, minInliers, thr, init_cv6, p, c, vf, o, nullptr);
        cv_to_jp6(o, f6);
        refine_cv(F, perm, refSteps, inlierCount, minInliers, thr, init_cv6, p, c, vb, o, nullptr);
        cv_to_jp6(o, b6);
        for (int k = 0; k < 6; k++) J[k * 3 * P + (size_t)p * 3 + c] = (f6[k] - b6[k]) / (2 * eps) * skip;
    }
}

// ---- core/cnn.h:786-852 refine (DSAC variant) and :854-990 dRefine --------------------------------------------------
// The DSAC variant's refine() restarts from P3P of the minimal set (cnn.h:797-800) instead of taking a pose, so its
// finite differences also perturb the first three set points -- on the map (localEstObj) and on the P3P input (objPts)
// alike (:875-880) -- before the inlier cells of dRefineObj's scheme (:930-986).  Column index y*W*3 + x*3 + c.
static void refine_from_set(const Frame& F, const int32_t* set4, const int32_t* perm, int refSteps, int inlierCount, int minInliers, float thr,
                            int pert_px, int pert_c, float pert_value, double* out_cv6) {
    float X4[12], uv4[8];
    for (int j = 0; j < 4; j++) {
        const int p = set4[j];
        for (int c = 0; c < 3; c++) X4[j * 3 + c] = (p == pert_px && c == pert_c) ? pert_value : F.xyz[(size_t)p * 3 + c];
        uv4[2 * j] = F.uv[2 * (size_t)p]; uv4[2 * j + 1] = F.uv[2 * (size_t)p + 1];
    }
    double init[6];
    safe_p3p(X4, uv4, F.K, init);
    refine_cv(F, perm, refSteps, inlierCount, minInliers, thr, init, pert_px, pert_c, pert_value, out_cv6, nullptr);
}

void orc_refine_from_set(const int32_t* set4, const int32_t* perm, int refSteps, int inlierCount, int minInliers, float thr, const float* xyz,
                         const float* uv, int H, int W, const double* cam, double* out_cv6) {
    Frame F{xyz, uv, H, W, Cam{cam[0], cam[1], cam[2], cam[3]}};
    refine_from_set(F, set4, perm, refSteps, inlierCount, minInliers, thr, -1, 0, 0.f, out_cv6);
}

void orc_dRefineDSAC(const int32_t* set4, const int32_t* perm, int refSteps, int inlierCount, int minInliers, float thr, float subSampleFactor,
                     const int32_t* inlier_map, const float* xyz, const float* uv, int H, int W, const double* cam, float eps,
                     double* J /*6 x 3P, zero-filled here*/) {
    Frame F{xyz, uv, H, W, Cam{cam[0], cam[1], cam[2], cam[3]}};
    const size_t P = (size_t)H * W;
    std::fill(J, J + 6 * 3 * P, 0.0);
    struct Job { int p, c; double scale; };
    std::vector<Job> jobs;
    for (int pt = 0; pt < 3; pt++)  // "skip last point, because gradient is anyway zero" (cnn.h:872)
        for (int c = 0; c < 3; c++) jobs.push_back({set4[pt], c, 1.0});
    const int skip = (int)(1 / subSampleFactor);
    int inCount = 0;
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
            if (inlier_map[y * W + x] == 0) continue;
            inCount++;
            if (inCount % skip != 0) continue;
            for (int c = 0; c < 3; c++) jobs.push_back({y * W + x, c, (double)skip});
        }
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < (int)jobs.size(); i++) {
        const Job& jb = jobs[i];
        const float v0 = F.xyz[(size_t)jb.p * 3 + jb.c];
        const float vf = v0 + eps;
        const float vb = vf - 2 * eps;
        double o[6], f6[6], b6[6];
        refine_from_set(F, set4, perm, refSteps, inlierCount, minInliers, thr, jb.p, jb.c, vf, o);
        cv_to_jp6(o, f6);
        refine_from_set(F, set4, perm, refSteps, inlierCount, minInliers, thr, jb.p, jb.c, vb, o);
        cv_to_jp6(o, b6);
        // a set point that is also a selected inlier cell would be written twice in the reference (second write wins, with
        // the skip factor); processImage clears the set's cells from the inlier map, so that does not happen
        for (int k = 0; k < 6; k++) J[k * 3 * P + (size_t)jb.p * 3 + jb.c] = (f6[k] - b6[k]) / (2 * eps) * jb.scale;
    }
}

// ---- core/cnn_softam.h:609-645  dScore, part (iii) --------------------------------------------------
// Given the gradient of the loss w.r.t. each error image (dDiff, N x P doubles, (y,x) row-major) this
// accumulates sum_h J_h into grad (P x 3, row-major y*W+x unless quirk_transpose).
// hyps are re-derived from the minimal sets exactly as dScore does (:583-602).  Also returns the
// per-hypothesis 1x6 pose gradient G6 (N x 6) and support gradient S (N x 12) for inspection.
void orc_dScore(int N, const int32_t* sets, const double* dDiff, const float* xyz, const float* uv, int H, int W, const double* cam,
                int quirks /*bit0: transposed columns (quirk 1); bit1: rotation written back per pixel (quirk 7)*/, double* grad /*P x 3, += */,
                double* G6_out, double* S_out) {
    const bool quirk_transpose = quirks & 1, quirk_writeback = quirks & 2;
    Frame F{xyz, uv, H, W, Cam{cam[0], cam[1], cam[2], cam[3]}};
    const int P = H * W;
    std::vector<std::vector<double>> jac(N);
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < N; h++) {
        std::vector<double>& J = jac[h];
        J.assign((size_t)P * 3, 0.0);
        float X4[12], uv4[8];
        for (int j = 0; j < 4; j++) {
            int p = sets[4 * h + j];
            for (int c = 0; c < 3; c++) X4[j * 3 + c] = xyz[(size_t)p * 3 + c];
            uv4[2 * j] = uv[2 * (size_t)p]; uv4[2 * j + 1] = uv[2 * (size_t)p + 1];
        }
        double cv6[6], R[9], t[3];
        safe_p3p(X4, uv4, F.K, cv6);
        cv2our(cv6, R, t);
        double dHdO[72];
        d_pnp(uv4, X4, 0.1f, F.K, dHdO);
        double G6[6] = {0, 0, 0, 0, 0, 0};
        for (int x = 0; x < W; x++)
            for (int y = 0; y < H; y++) {
                int p = y * W + x;
                double w = dDiff[(size_t)h * P + p];
                double dPdO[3], dPdH[6];
                d_project_d_obj(uv + 2 * (size_t)p, xyz + 3 * (size_t)p, R, t, F.K, dPdO);
                d_project_d_hyp(uv + 2 * (size_t)p, xyz + 3 * (size_t)p, R, t, F.K, dPdH, quirk_writeback ? R : nullptr);
                int col = quirk_transpose ? (x * W * 3 + y * 3) : (p * 3);
                for (int c = 0; c < 3; c++) J[col + c] = w * dPdO[c];  // copyTo (overwrite)
                for (int k = 0; k < 6; k++) G6[k] += w * dPdH[k];
            }
        double S[12];
        for (int j = 0; j < 12; j++) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += G6[k] * dHdO[k * 12 + j];
            S[j] = s;
        }
        for (int i = 0; i < 4; i++) {
            int p = sets[4 * h + i], x = p % W, y = p / W;
            int col = quirk_transpose ? (x * W * 3 + y * 3) : (p * 3);
            for (int c = 0; c < 3; c++) J[col + c] += S[i * 3 + c];
        }
        if (G6_out) std::memcpy(G6_out + 6 * h, G6, sizeof(G6));
        if (S_out) std::memcpy(S_out + 12 * h, S, sizeof(S));
    }
    for (int h = 0; h < N; h++)  // core/train_ransac_softam.cpp:382-383
        for (size_t i = 0; i < (size_t)P * 3; i++) grad[i] += jac[h][i];
}

// ---- core/train_ransac_softam.cpp:344-353 path I, second term, and :361-376 softmax backward ---------
// v6 = dLoss/dRef * dRef/dAvg (1x6).  Adds v6 * sum_h w_h scatter(dPNP_h) into grad (index idx*3, the
// reference's imgIdx = y*W + x) and writes the score gradients g[N].
void orc_path1_pnp_and_softmax_bwd(int N, const double* v6, const double* w, const double* poses, const int32_t* sets, const float* xyz,
                                   const float* uv, int H, int W, const double* cam, double* grad /*P x 3, +=*/, double* g /*N*/) {
    Cam K{cam[0], cam[1], cam[2], cam[3]};
    std::vector<double> Jall((size_t)N * 72);
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < N; h++) {
        float X4[12], uv4[8];
        for (int j = 0; j < 4; j++) {
            int p = sets[4 * h + j];
            for (int c = 0; c < 3; c++) X4[j * 3 + c] = xyz[(size_t)p * 3 + c];
            uv4[2 * j] = uv[2 * (size_t)p]; uv4[2 * j + 1] = uv[2 * (size_t)p + 1];
        }
        d_pnp(uv4, X4, 0.1f, K, Jall.data() + (size_t)h * 72);
    }
    for (int h = 0; h < N; h++)
        for (int i = 0; i < 4; i++) {
            int p = sets[4 * h + i];
            for (int c = 0; c < 3; c++) {
                double s = 0;
                for (int k = 0; k < 6; k++) s += v6[k] * w[h] * Jall[(size_t)h * 72 + k * 12 + i * 3 + c];
                grad[(size_t)p * 3 + c] += s;
            }
        }
    // softmax backward, written as the reference's O(N^2) double loop
    for (int j = 0; j < N; j++) g[j] = 0;
    for (int h = 0; h < N; h++) {
        double F = 0;
        for (int k = 0; k < 3; k++) F += v6[k] * poses[6 * h + k];
        for (int k = 3; k < 6; k++) F += v6[k] * (poses[6 * h + k] / 1000);
        g[h] += w[h] * F;
        for (int j = 0; j < N; j++) g[j] -= w[h] * w[j] * F;
    }
}

// ---- timing helper for bench.py's cpu_baseline: sample + N error images + soft-inlier + softmax ------
double orc_time_forward(int N, uint64_t seed, const float* xyz, const float* uv, int H, int W, const double* cam, float thr, int max_tries,
                        float tau, float beta, double alpha, int reps, double* weights_out) {
    const size_t P = (size_t)H * W;
    std::vector<double> poses(6 * (size_t)N), score(N), w(N);
    std::vector<int32_t> sets(4 * (size_t)N);
    std::vector<uint8_t> ok(N);
    std::vector<float> err((size_t)N * P);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; r++) {
        orc_sample(N, seed + r, nullptr, xyz, uv, H, W, cam, thr, max_tries, poses.data(), sets.data(), ok.data(), nullptr);
        orc_get_diff_maps(N, poses.data(), xyz, uv, H, W, cam, err.data());
        orc_soft_inlier(N, err.data(), (int)P, tau, beta, score.data());
        for (int h = 0; h < N; h++) score[h] *= alpha;
        soft_max(N, score.data(), w.data());
    }
    auto t1 = std::chrono::high_resolution_clock::now();
    if (weights_out) std::memcpy(weights_out, w.data(), sizeof(double) * N);
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
