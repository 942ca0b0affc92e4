// loss_math.h -- maxLoss (core/maxloss.h:69-79 with getInvHyp :39-61 and Hypothesis::calcAngularDistance, core/Hypothesis.cpp:137-143) as a device
// function: K7 (k_loss.hip) evaluates it per estimate, K6 (k_refine.hip) at the end of a refinement whose loss is asked for in the same launch.
#pragma once
#include "dmath.h"

namespace dk {

DM_INLINE void inv3(const double A[9], double Ai[9]) {
    const double d = dm::det3(A);
    const double id = 1.0 / d;
    Ai[0] = (A[4] * A[8] - A[5] * A[7]) * id; Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id; Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Ai[3] = (A[5] * A[6] - A[3] * A[8]) * id; Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id; Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Ai[6] = (A[3] * A[7] - A[4] * A[6]) * id; Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id; Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}
DM_INLINE void mul3(const double A[9], const double B[9], double C[9]) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// est_cv6: the estimate (cv convention, rvec | tvec mm); gt: the ground truth's jp 6-vector.  R1 / t1: the estimate in the jp convention (dLossMax
// continues from them).  out4 = loss, rotErr [deg], tErr [mm], correct (5 deg / 5 cm, core/cnn_softam.h:1172-1173).
DM_INLINE void max_loss_forward(const double* est_cv6, const double* gt_jp6, double R1[9], double t1[3], double gt[6], double* out4) {
    const double PI = 3.14159265358979323846;
    double cv6[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { cv6[k] = est_cv6[k]; gt[k] = gt_jp6[k]; }
    // estimate in the jp convention (cv2our), ground truth from its jp 6-vector (Hypothesis(std::vector<double>), Hypothesis.cpp:81-99)
    double R2[9];
    dm::cv2our(cv6, R1, t1);
    const double glen = sqrt(gt[0] * gt[0] + gt[1] * gt[1] + gt[2] * gt[2]);
    if (glen > 1e-5) dm::rodrigues_v2m<false>(gt, R2, nullptr);
    else { R2[0] = 1; R2[1] = 0; R2[2] = 0; R2[3] = 0; R2[4] = 1; R2[5] = 0; R2[6] = 0; R2[7] = 0; R2[8] = 1; }
    const double t2[3] = {gt[3], gt[4], gt[5]};
    // getInvHyp: inverse of [R t; 0 1] = [R^-1, -R^-1 t]
    double Ri1[9], Ri2[9], ti1[3], ti2[3];
    inv3(R1, Ri1);
    inv3(R2, Ri2);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        ti1[i] = -(Ri1[i * 3] * t1[0] + Ri1[i * 3 + 1] * t1[1] + Ri1[i * 3 + 2] * t1[2]);
        ti2[i] = -(Ri2[i * 3] * t2[0] + Ri2[i * 3 + 1] * t2[1] + Ri2[i * 3 + 2] * t2[2]);
    }
    // invH1.calcAngularDistance(invH2): trace(invR1 * inv(invR2))
    double Rii2[9], D[9];
    inv3(Ri2, Rii2);
    mul3(Ri1, Rii2, D);
    double tr = D[0] + D[4] + D[8];
    tr = fmin(3.0, fmax(-1.0, tr));
    const double rotErr = 180 * acos((tr - 1.0) / 2.0) / PI;
    const double dx = ti1[0] - ti2[0], dy = ti1[1] - ti2[1], dz = ti1[2] - ti2[2];
    const double tErr = sqrt(dx * dx + dy * dy + dz * dz);
    if (out4) {
        out4[0] = fmin(fmax(rotErr, tErr / 10), 10000000.0);
        out4[1] = rotErr;
        out4[2] = tErr;
        out4[3] = (rotErr < 5 && tErr < 50) ? 1.0 : 0.0;  // core/cnn_softam.h:1172-1173
    }
}

}  // namespace dk
