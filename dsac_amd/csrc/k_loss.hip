// k_loss.hip -- K7: pose loss max(rotation error [deg], translation error [cm]) and its analytic gradient.
//
// Replaces maxLoss (core/maxloss.h:69-79, with getInvHyp :39-61 and Hypothesis::calcAngularDistance,
// core/Hypothesis.cpp:137-143) and dLossMax (core/maxloss.h:87-198).  One lane; kept on the device so that a
// training step can chain loss -> refinement Jacobians -> score backward without a host round trip.
#include "kernels.h"
#include "dmath.h"
#include "loss_math.h"

namespace dk {

// One lane per estimate (B = 1: maxLoss / dLossMax of the soft-argmax pipeline; B = N: the per-hypothesis losses of
// expectedMaxLoss, core/cnn.h:137-150, and the per-hypothesis dLossMax of core/train_ransac.cpp:345-349).
__global__ __launch_bounds__(64) void k_pose_loss(int B, const double* __restrict__ est_all, const double* __restrict__ gt_jp6,
                                                  double* __restrict__ out4_all, double* __restrict__ J6_all, int gt_stride, int gt_group) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    gt_jp6 += (size_t)(b / gt_group) * gt_stride;  // stride 0: one ground truth for all estimates; 6: one per group of gt_group estimates (a frame batch)
    const double* est_cv6 = est_all + (size_t)b * 6;
    double* out4 = out4_all ? out4_all + (size_t)b * 4 : nullptr;
    double* J6 = J6_all ? J6_all + (size_t)b * 6 : nullptr;
    const double PI = 3.14159265358979323846;
    double R1[9], t1[3], gt[6];
    max_loss_forward(est_cv6, gt_jp6, R1, t1, gt, out4);
    if (!J6) return;

    // ---- dLossMax(est jp 6-vector, gt jp 6-vector) ----
    double est[6];
    dm::rodrigues_m2v(R1, est);
    est[3] = t1[0]; est[4] = t1[1]; est[5] = t1[2];
    double rot1[9], rot2[9], dRod[27];
    dm::rodrigues_v2m<true>(est, rot1, dRod);
    dm::rodrigues_v2m<false>(gt, rot2, nullptr);
    double J[6] = {0, 0, 0, 0, 0, 0};
    // diffRot = rot1 * rot2^T
    double trace = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) trace += rot1[i * 3 + k] * rot2[i * 3 + k];
    trace = fmin(3.0, fmax(-1.0, trace));
    const double rErr = 180 * acos((trace - 1.0) / 2.0) / PI;
    double invT1[3], invT2[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {  // rot^T * (-t/10)
        invT1[i] = rot1[0 * 3 + i] * (-est[3] / 10) + rot1[1 * 3 + i] * (-est[4] / 10) + rot1[2 * 3 + i] * (-est[5] / 10);
        invT2[i] = rot2[0 * 3 + i] * (-gt[3] / 10) + rot2[1 * 3 + i] * (-gt[4] / 10) + rot2[2 * 3 + i] * (-gt[5] / 10);
    }
    const double d0 = invT1[0] - invT2[0], d1 = invT1[1] - invT2[1], d2 = invT1[2] - invT2[2];
    const double tE = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    const bool zero = (fmax(rErr, tE) > 10000000.0) || ((tE + rErr) < 0.00000001);
    if (!zero) {
        if (tE > rErr) {
            const double dd[3] = {d0 / tE, d1 / tE, d2 / tE};
            // J[3:6] = dd * (-invRot1),  invRot1 = rot1^T
#pragma unroll
            for (int c = 0; c < 3; c++) J[3 + c] = -(dd[0] * rot1[c * 3 + 0] + dd[1] * rot1[c * 3 + 1] + dd[2] * rot1[c * 3 + 2]);
            // v9[r + 3c] = dd[r] * (-est[3+c]/10)    (core/maxloss.h:147-158)
            double v9[9];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) v9[r + 3 * c] = dd[r] * (-est[3 + c] / 10);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) s += v9[k] * dRod[i * 9 + k];
                J[i] = s;
            }
        } else {
            // v9 = dTrace * dRotDiff^T: v9[3b + r] = invRot2(r, b) = rot2(b, r)   (core/maxloss.h:170-188)
            double v9[9];
#pragma unroll
            for (int b = 0; b < 3; b++)
#pragma unroll
                for (int r = 0; r < 3; r++) v9[3 * b + r] = rot2[b * 3 + r];
            const double scale = 180 / PI * -1 / sqrt(3 - trace * trace + 2 * trace);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) s += v9[k] * dRod[i * 9 + k];
                J[i] = scale * s;
            }
        }
        bool nan = false;
#pragma unroll
        for (int i = 0; i < 6; i++) nan = nan || (J[i] != J[i]);
        if (nan) {
#pragma unroll
            for (int i = 0; i < 6; i++) J[i] = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) J6[i] = J[i];
}

hipError_t pose_loss(hipStream_t st, int B, const double* est_cv6, const double* gt_jp6, double* out4, double* J6, int gt_stride, int gt_group) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pose_loss, dim3((B + 63) / 64), dim3(64), 0, st, B, est_cv6, gt_jp6, out4, J6, gt_stride, gt_group < 1 ? 1 : gt_group);
    return hipGetLastError();
}


// --------------------------------------------------------------------------------------------------
// DSAC variant (core/cnn.h): probabilistic selection, expected loss and the softmax-expectation gradient on the device.
//   draw            core/cnn.h:102-127   entries below EPS skipped; cumulative sums in index order are the keys of a std::map (a later entry
//                                        that leaves the sum unchanged overwrites the earlier one), upper_bound(u * sum); argmax when u < 0
//   expectedMaxLoss core/cnn.h:137-150   sum_i probs[i] * losses[i] in index order
//   dSMScore        core/cnn.h:737-742   g[i] = w_i L_i - sum_j w_i w_j L_j, subtracted term by term in index order
// One workgroup; the two scalar scans run on one lane (N sequential fp64 additions, the reference's own order), the N x N subtraction on all lanes.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dsac_select(int N, const double* __restrict__ w, const double* __restrict__ losses, int loss_stride, double u,
                                                     double eps, int32_t* __restrict__ hyp_idx, double* __restrict__ expected, double* __restrict__ g,
                                                     const double* __restrict__ u_frames) {
    const int tid = threadIdx.x;
    if (u_frames) {  // image blockIdx.x of a frame batch: its N probabilities / losses / gradients, its own draw
        const size_t f = blockIdx.x;
        u = u_frames[f];
        w += f * N; losses += f * (size_t)N * loss_stride;
        if (hyp_idx) hyp_idx += f;
        if (expected) expected += f;
        if (g) g += f * N;
    }
    if (tid == 0) {
        double sum = 0;
        for (int i = 0; i < N; i++)
            if (!(w[i] < eps)) sum += w[i];
        int pick = -1;
        if (u < 0) {  // randomDraw == false: the first entry with the largest probability
            double best = -1;
            for (int i = 0; i < N; i++) {
                if (w[i] < eps) continue;
                if (best < 0 || w[i] > best) { best = w[i]; pick = i; }
            }
            if (pick < 0) pick = 0;  // maxIdx is initialised to 0
        } else {
            const double r = u * sum;  // drand(0, probSum)
            double s = 0, key = 0;
            bool found = false;
            int last = 0;
            for (int i = 0; i < N; i++) {
                if (w[i] < eps) continue;
                s += w[i];
                last = i;
                if (!found) {
                    if (s > r) { found = true; key = s; pick = i; }
                } else if (s == key) {
                    pick = i;  // same map key: the later index replaces the earlier one
                } else {
                    break;
                }
            }
            if (!found) pick = last;  // u * sum >= the last key: upper_bound would be end(); the last entry stands in
        }
        if (hyp_idx) *hyp_idx = pick;
        if (expected) {
            double e = 0;
            for (int i = 0; i < N; i++) e = __dadd_rn(e, __dmul_rn(w[i], losses[(size_t)i * loss_stride]));  // product rounded, then added: no fused multiply-add
            *expected = e;
        }
    }
    if (g) {
        for (int i = tid; i < N; i += blockDim.x) {
            const double wi = w[i];
            double v = __dmul_rn(wi, losses[(size_t)i * loss_stride]);
            for (int j = 0; j < N; j++) v = __dsub_rn(v, __dmul_rn(__dmul_rn(wi, w[j]), losses[(size_t)j * loss_stride]));  // (w_i w_j) L_j as the reference groups it
            g[i] = v;
        }
    }
}

hipError_t dsac_select(hipStream_t st, int N, const double* w, const double* losses, int loss_stride, double u, double eps, int32_t* hyp_idx, double* expected,
                       double* g) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_dsac_select, dim3(1), dim3(256), 0, st, N, w, losses, loss_stride, u, eps, hyp_idx, expected, g, (const double*)nullptr);
    return hipGetLastError();
}

hipError_t dsac_select_frames(hipStream_t st, int frames, int N, const double* w, const double* losses, int loss_stride, const double* u, double eps,
                              int32_t* hyp_idx, double* expected, double* g) {
    if (N <= 0 || frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_dsac_select, dim3(frames), dim3(256), 0, st, N, w, losses, loss_stride, 0.0, eps, hyp_idx, expected, g, u);
    return hipGetLastError();
}

}  // namespace dk
