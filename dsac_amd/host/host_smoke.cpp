// host_smoke.cpp -- C++ caller of the host shim (runs on the GPU box): builds a synthetic frame, runs
// processImage + the score backward through the reference-shaped API and prints the result.
#include <cstdio>
#include <random>

#include "cnn.h"

int main() {
    using namespace dsac;
    const int H = 48, W = 64;
    const Camera cam;
    std::mt19937 rng(1305);
    std::uniform_real_distribution<double> U(0, 1);
    std::normal_distribution<double> G(0, 1);
    const cv_trans_t gt = {{0.15, -0.1, 0.05}, {120.0, -80.0, 2300.0}};
    const Mat3 R = rodrigues(gt.rvec);
    const Mat3 Ri = inverse(R);
    std::vector<float> xyz((size_t)H * W * 3), uv((size_t)H * W * 2);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int p = y * W + x;
            const double u = 10.0 * x + 5, v = 10.0 * y + 5, d = 800 + 2700 * U(rng);
            const Vec3 Xc = {(u - cam.cx) / cam.fx * d - gt.tvec[0], (v - cam.cy) / cam.fy * d - gt.tvec[1], d - gt.tvec[2]};
            const bool outlier = U(rng) < 0.3;
            for (int c = 0; c < 3; c++) {
                double X = Ri[c * 3] * Xc[0] + Ri[c * 3 + 1] * Xc[1] + Ri[c * 3 + 2] * Xc[2];
                X = outlier ? 4000 * (U(rng) - 0.5) : X + 20 * G(rng);
                xyz[p * 3 + c] = (float)X;
            }
            uv[p * 2] = (float)u;
            uv[p * 2 + 1] = (float)v;
        }
    try {
        Frame frame(xyz.data(), uv.data(), H, W, cam);
        const std::vector<int32_t> perms = refinePermutations(H * W, 8);
        const Hypothesis poseGT(cv2our(gt));
        ProcessImageResult r = frame.processImage(poseGT, 256, 1305, 10, 100, 8, perms);
        std::printf("processImage: entropy %.3f bits, refine steps %d, loss %.4f (rot %.4f deg, trans %.3f mm), correct %d\n", r.sfEntropy,
                    r.refStepsDone, r.loss, r.rotErr, r.tErr, (int)r.correct);
        std::vector<float> dDiff((size_t)256 * H * W, 1e-3f);
        std::vector<double> jac;
        frame.dScore(r.hyps, r.imgIdx, dDiff, jac);
        double n = 0;
        for (double v : jac) n += v * v;
        std::printf("dScore: |grad| = %.6g\n", std::sqrt(n));
        const std::array<double, 6> dL = frame.dLossMax(r.refAvgHyp, poseGT);
        std::printf("dLossMax: %.4g %.4g %.4g %.4g %.4g %.4g\n", dL[0], dL[1], dL[2], dL[3], dL[4], dL[5]);
        // the DSAC (probabilistic selection) variant on the same frame: all hypotheses refined, expected loss, dRefine of the winner
        DsacFrame dframe(xyz.data(), uv.data(), H, W, cam);
        ProcessImageDsacResult d = dframe.processImage(poseGT, 64, 1305, 10, 100, 8, perms);
        std::array<double, 54> Jset{};
        std::vector<int32_t> px;
        std::vector<double> Jobj;
        dframe.dRefine(100, 8, 0.05f, 10.f, perms, d.imgIdx[d.hypIdx], d.inlierMaps.data() + (size_t)d.hypIdx * H * W, Jset, px, Jobj);
        double ns = 0;
        for (double v : Jset) ns += v * v;
        std::printf("DSAC variant: hypIdx %d (p = %.3f), expected loss %.4f, winner rot %.4f deg / trans %.3f mm, |dRefine_set| = %.4g, %zu inlier cells\n",
                    d.hypIdx, d.sfScores[d.hypIdx], d.expectedLoss, d.rotErr, d.tErr, std::sqrt(ns), px.size());
        return (r.correct && r.refStepsDone == 8 && n > 0 && d.correct && ns > 0 && d.expectedLoss > 0) ? 0 : 2;
    } catch (const Error& e) {
        std::printf("dsac error %d: %s\n", e.code, e.what());
        return 1;
    }
}
