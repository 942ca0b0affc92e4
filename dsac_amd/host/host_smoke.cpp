// host_smoke.cpp -- C++ caller of the host shim (runs on the GPU box): builds a synthetic frame, runs
// processImage + the score backward through the reference-shaped API and prints the result.
#include <cstdio>
#include <random>

#include "cnn.h"

// argv[1] (optional): a file that receives the frame and every result of processImage as raw little-endian arrays, in the order written below --
// tests/test_gpu_host_shim.py recomputes them with the CPU oracle.
template <typename T>
static void put(std::FILE* f, const T* p, size_t n) { std::fwrite(p, sizeof(T), n, f); }

int main(int argc, char** argv) {
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
        // the batched path on the same context: the frame twice, seeds 1305 and 1306 -- image 0 must equal the single-image call bit for bit
        FrameBatchOptions bo;
        bo.inlierMaps = true;
        // (the explicit sampling grid of this frame is not the implicit full-resolution one a FrameBatch assumes: compare on a grid frame)
        std::vector<float> gxyz = xyz;
        Frame gframe(gxyz.data(), nullptr, H, W, cam);
        const ProcessImageResult g1 = gframe.processImage(poseGT, 256, 1305, 10, 100, 8, perms), g2 = gframe.processImage(poseGT, 256, 1306, 10, 100, 8, perms);
        FrameBatch fb(gframe.engine(), 2, H, W, cam, 256, 8, perms, 2, bo);
        fb.setFrame(0, gxyz.data(), poseGT);
        fb.setFrame(1, gxyz.data(), poseGT);
        fb.processAll(1305, 10, 100);
        const std::vector<ProcessImageResult> br = fb.results();
        auto same = [](const ProcessImageResult& a, const ProcessImageResult& b) {
            bool ok = a.imgIdx == b.imgIdx && a.sfScores == b.sfScores && a.sfEntropy == b.sfEntropy && a.refStepsDone == b.refStepsDone && a.loss == b.loss &&
                      a.rotErr == b.rotErr && a.tErr == b.tErr && a.inlierMap == b.inlierMap;
            for (int k = 0; k < 3; k++) ok = ok && a.refAvgHyp.rvec[k] == b.refAvgHyp.rvec[k] && a.refAvgHyp.tvec[k] == b.refAvgHyp.tvec[k] && a.avgHyp.rvec[k] == b.avgHyp.rvec[k];
            for (size_t h = 0; h < a.hyps.size() && ok; h++)
                for (int k = 0; k < 3; k++) ok = ok && a.hyps[h].rvec[k] == b.hyps[h].rvec[k] && a.hyps[h].tvec[k] == b.hyps[h].tvec[k];
            return ok;
        };
        const bool batchEqual = br.size() == 2 && same(br[0], g1) && same(br[1], g2);
        std::printf("FrameBatch: 2 images in one launch chain %s the per-image calls (losses %.6f / %.6f)\n", batchEqual ? "equal" : "DIFFER FROM", br[0].loss, br[1].loss);
        // sub-sampled frames bring their own table of image positions (FrameBatchOptions::sampling): the batch on (xyz, uv) equals frame.processImage,
        // its backward pass equals Frame::backward, and a step batch filled device-to-device (gatherFramesFrom / copyFrameFrom) gives the same again
        bool samplingEqual = false;
        {
            FrameBatchOptions so = bo;
            so.sampling = true;
            FrameBatch sb(frame.engine(), 2, H, W, cam, 256, 8, perms, 2, so);
            sb.setFrame(0, xyz.data(), poseGT, uv.data());
            sb.setFrame(1, xyz.data(), poseGT, uv.data());
            sb.processAll(1305, 10, 100);
            const std::vector<ProcessImageResult> sr = sb.results();
            sb.backward(0, 2, 10, 100, 0.05f);
            const std::vector<double> gb = sb.gradients(0);
            const std::vector<double> gl = frame.backward(r, poseGT, 10, 100, 8, 0.05f, perms);
            double dmax = 0, gmax = 0;
            for (size_t i = 0; i < gl.size(); i++) { dmax = std::max(dmax, std::fabs(gl[i] - gb[i])); gmax = std::max(gmax, std::fabs(gl[i])); }
            FrameBatch step(frame.engine(), 2, H, W, cam, 256, 8, perms, 2, so);
            step.gatherFramesFrom(sb, std::vector<int32_t>{1, 0});
            step.processAll(1305, 10, 100);
            const std::vector<ProcessImageResult> s1 = step.results();
            step.copyFrameFrom(sb, 0, 1);
            step.copyFrameFrom(sb, 1, 0);
            step.processAll(1305, 10, 100);
            const std::vector<ProcessImageResult> s2 = step.results();
            samplingEqual = sr.size() == 2 && same(sr[0], r) && same(s1[0], sr[0]) && same(s1[1], sr[1]) && same(s2[0], sr[0]) && same(s2[1], sr[1]) && gmax > 0 &&
                            dmax <= 1e-4 * gmax;  // K4's fp32 partial sums are grouped by the launch's workgroup count (two frames per launch here, one in Frame::backward)
            std::printf("FrameBatch with sampling tables: forward %s Frame::processImage, backward |d| / |g| = %.3g, gathered step batch %s\n",
                        (sr.size() == 2 && same(sr[0], r)) ? "equals" : "DIFFERS FROM", gmax > 0 ? dmax / gmax : -1.0,
                        (same(s1[0], sr[0]) && same(s2[1], sr[1])) ? "equal" : "DIFFERS");
        }
        if (argc > 1) {
            std::FILE* f = std::fopen(argv[1], "wb");
            if (!f) { std::printf("cannot write %s\n", argv[1]); return 3; }
            const int32_t hdr[4] = {H, W, 256, 8};
            put(f, hdr, 4);
            put(f, xyz.data(), xyz.size());
            put(f, uv.data(), uv.size());
            put(f, perms.data(), perms.size());
            const std::vector<double> gtv = poseGT.getRodVecAndTrans();
            put(f, gtv.data(), 6);
            for (const auto& s4 : r.imgIdx) put(f, s4.data(), 4);
            for (const auto& h : r.hyps) { const Pose6 p6 = pack(h); put(f, p6.data(), 6); }
            put(f, r.sfScores.data(), r.sfScores.size());
            const Pose6 a6 = pack(r.avgHyp), r6 = pack(r.refAvgHyp);
            put(f, a6.data(), 6);
            put(f, r6.data(), 6);
            const double tail[5] = {r.sfEntropy, r.loss, r.rotErr, r.tErr, (double)r.refStepsDone};
            put(f, tail, 5);
            put(f, r.inlierMap.data(), r.inlierMap.size());
            std::fclose(f);
        }
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
        return (r.correct && r.refStepsDone == 8 && n > 0 && d.correct && ns > 0 && d.expectedLoss > 0 && batchEqual && samplingEqual) ? 0 : 2;
    } catch (const Error& e) {
        std::printf("dsac error %d: %s\n", e.code, e.what());
        return 1;
    }
}
