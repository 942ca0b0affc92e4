// oracle/refbuild/ref_glue_dsac.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// C entry points (refd_*) around the DSAC (probabilistic selection) variant of the reference: core/cnn.h compiled where it
// lies, with the same OpenCV / Lua stand-ins as ref_glue.cpp (which holds the soft-argmax twin, core/cnn_softam.h; the two
// headers define the same function names, hence two shared libraries).  The reference's own code on this path: draw,
// expectedMaxLoss, refine (restart from P3P of the minimal set), dRefine, dSMScore, processImage (core/cnn.h).
// refd_processImage additionally restates the call sequence of the trainer's backward section
// (core/train_ransac.cpp:303-373; main() itself needs the 7-Scenes files) on those real functions.
#include <iostream>
#include <fstream>
#include <cstdint>
#include <unistd.h>
#include <omp.h>

// same include order as core/train_ransac.cpp:32-39
#include "properties.h"
#include "thread_rand.h"
#include "util.h"
#include "stop_watch.h"
#include "dataset.h"
#include "generic_io.h"
#include "lua_calls.h"
#include "cnn.h"

#include "ref_common.h"

namespace {
void set_from(const int32_t* set4, const float* xyz, const int32_t* uv, int W, std::vector<cv::Point2f>& ip, std::vector<cv::Point3f>& op,
              std::vector<cv::Point2i>& sp) {
    for (int i = 0; i < 4; i++) {
        const int p = set4[i];
        ip.push_back(cv::Point2f(uv[2 * p], uv[2 * p + 1]));
        op.push_back(cv::Point3f(xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]));
        sp.push_back(cv::Point2i(p % W, p / W));
    }
}
}  // namespace

extern "C" {

int refd_init(const char* scratch_dir, float f, int imageWidth, int imageHeight, float xShift, float yShift, int randomDraw) {
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) return -1;
    if (chdir(scratch_dir) != 0) return -2;
    {
        std::ofstream o("./sensorTrans.dat", std::ios::binary);
        cv::Mat_<double> eye = cv::Mat_<double>::eye(4, 4);
        jp::write(o, eye);
    }
    std::cout.setstate(std::ios_base::failbit);
    GlobalProperties* gp = GlobalProperties::getInstance();
    if (chdir(cwd) != 0) return -3;
    gp->dP.focalLength = f; gp->dP.imageWidth = imageWidth; gp->dP.imageHeight = imageHeight; gp->dP.xShift = xShift; gp->dP.yShift = yShift;
    gp->pP.randomDraw = randomDraw != 0;
    return 0;
}
void refd_set_score_model(double tau, double beta, double alpha) { g_score.tau = tau; g_score.beta = beta; g_score.alpha = alpha; }

// draw (cnn.h:102-127) with the reference's own generator, re-seeded first
int refd_draw(unsigned seed, int n, const double* probs) {
    ThreadRand::forceInit(seed);
    return draw(std::vector<double>(probs, probs + n));
}

// refine (cnn.h:786-852): restart from P3P of the minimal set; returns the jp 6-vector
void refd_refine_from_set(int inlierCount, int refSteps, float thr, const int32_t* perm, const float* xyz, const int32_t* uv, int H, int W,
                          const int32_t* set4, double* out_jp6) {
    cv::Mat_<float> camMat = GlobalProperties::getInstance()->getCamMat();
    std::vector<cv::Point2f> ip; std::vector<cv::Point3f> op; std::vector<cv::Point2i> sp;
    set_from(set4, xyz, uv, W, ip, op, sp);
    std::vector<double> r = refine(inlierCount, refSteps, thr, make_perm(perm, refSteps, H * W), make_obj(xyz, H, W), make_sampling(uv, H, W), camMat, ip, op);
    for (int i = 0; i < 6; i++) out_jp6[i] = r[i];
}

// dRefine (cnn.h:854-990): J is 6 x 3*H*W, columns y*CNN_OBJ_PATCHSIZE*3 + x*3 + c (dense for W == 40)
void refd_dRefine(int inlierCount, int refSteps, float subSample, float thr, const int32_t* perm, const float* xyz, const int32_t* uv, int H, int W,
                  const int32_t* set4, const int32_t* inlierMap, double* J) {
    cv::Mat_<float> camMat = GlobalProperties::getInstance()->getCamMat();
    std::vector<cv::Point2f> ip; std::vector<cv::Point3f> op; std::vector<cv::Point2i> sp;
    set_from(set4, xyz, uv, W, ip, op, sp);
    cv::Mat_<int> im(H, W);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) im(y, x) = inlierMap[(size_t)y * W + x];
    put_mat(dRefine(inlierCount, refSteps, subSample, thr, make_perm(perm, refSteps, H * W), make_obj(xyz, H, W), make_sampling(uv, H, W), camMat, ip, op, sp, im), J);
}

struct RefdFrameOut {
    double expectedLoss, sfEntropy, tErr, rotErr;
    int correct, hypIdx;
};
// processImage (cnn.h:1000-1240) + backward section of train_ransac.cpp:303-373, single-threaded for a reproducible RNG stream
int refd_processImage(unsigned seed, int objHyps, int inlierThreshold2D, int inlierCount, int refSteps, const float* pred_mm, const double* gt_jp6,
                      RefdFrameOut* out, double* hyps_cv6, double* refHyps_cv6, int32_t* sampledPoints, double* sfScores, double* losses,
                      int32_t* sampling_uv, float* estObj_mm, int32_t* inlierMaps /*N x 1600*/, int32_t* pixelIdxs /*refSteps x 1600 (of hypothesis 0)*/,
                      double* dLoss_dObj /*1600 x 3 or NULL*/, float refSubSample) {
    const int S = CNN_OBJ_PATCHSIZE, P = S * S;
    const int saved_threads = omp_get_max_threads();
    omp_set_num_threads(1);
    ThreadRand::forceInit(seed);
    GlobalProperties* gp = GlobalProperties::getInstance();
    cv::Mat_<float> camMat = gp->getCamMat();
    g_pred_m.resize((size_t)P * 3);
    for (int i = 0; i < P * 3; i++) g_pred_m[i] = pred_mm[i] / 1000.f;
    jp::img_bgr_t img = jp::img_bgr_t::zeros(gp->dP.imageHeight, gp->dP.imageWidth);
    Hypothesis poseGT(std::vector<double>(gt_jp6, gt_jp6 + 6));

    double expectedLoss, sfEntropy, tErr, rotErr; bool correct; int hypIdx;
    std::vector<jp::cv_trans_t> hyps, refHyps;
    std::vector<std::vector<cv::Point2f>> imgPts; std::vector<std::vector<cv::Point3f>> objPts; std::vector<std::vector<int>> imgIdx;
    std::vector<cv::Mat_<cv::Vec3f>> patches; std::vector<double> sf, ls; jp::img_coord_t estObj; cv::Mat_<cv::Point2i> sampling;
    std::vector<std::vector<cv::Point2i>> sampled; std::vector<cv::Mat_<int>> inl; std::vector<std::vector<std::vector<int>>> pix;
    processImage(img, poseGT, coord_state(), score_state(), objHyps, 4, camMat, inlierThreshold2D, inlierCount, refSteps, expectedLoss, sfEntropy, correct,
                 hyps, refHyps, imgPts, objPts, imgIdx, patches, sf, estObj, sampling, sampled, ls, inl, pix, tErr, rotErr, hypIdx);

    out->expectedLoss = expectedLoss; out->sfEntropy = sfEntropy; out->tErr = tErr; out->rotErr = rotErr; out->correct = correct; out->hypIdx = hypIdx;
    for (int h = 0; h < objHyps; h++) {
        put_cv(hyps[h], hyps_cv6 + 6 * h); put_cv(refHyps[h], refHyps_cv6 + 6 * h);
        for (int i = 0; i < 4; i++) { sampledPoints[(h * 4 + i) * 2] = sampled[h][i].x; sampledPoints[(h * 4 + i) * 2 + 1] = sampled[h][i].y; }
        sfScores[h] = sf[h]; losses[h] = ls[h];
        for (int y = 0; y < S; y++) for (int x = 0; x < S; x++) inlierMaps[(size_t)h * P + y * S + x] = inl[h](y, x);
    }
    for (int y = 0; y < S; y++) for (int x = 0; x < S; x++) {
        const int p = y * S + x;
        sampling_uv[2 * p] = sampling(y, x).x; sampling_uv[2 * p + 1] = sampling(y, x).y;
        for (int c = 0; c < 3; c++) estObj_mm[3 * p + c] = estObj(y, x)[c];
    }
    for (int s = 0; s < refSteps; s++) for (int i = 0; i < P; i++) pixelIdxs[(size_t)s * P + i] = (i < (int)pix[0][s].size()) ? pix[0][s][i] : -1;
    // every hypothesis re-seeds the same default generator (cnn.h:1169): all permutation lists are equal
    for (int h = 1; h < objHyps; h++) for (int s = 0; s < refSteps; s++) if (pix[h][s].size() && pix[h][s] != pix[0][s]) return -7;

    if (dLoss_dObj) {
        // core/train_ransac.cpp:310-373: the call sequence of the backward section on the real functions
        std::vector<cv::Mat_<double>> dHyp_dObjs(refHyps.size());
        for (unsigned h = 0; h < refHyps.size(); h++) {
            if (sf[h] > 0.0001) dHyp_dObjs[h] = dRefine(inlierCount, refSteps, refSubSample, inlierThreshold2D, pix[h], estObj, sampling, camMat, imgPts[h], objPts[h], sampled[h], inl[h]);
            else dHyp_dObjs[h] = cv::Mat_<double>::zeros(6, P * 3);
        }
        cv::Mat_<double> acc = cv::Mat_<double>::zeros(P, 3);
        for (unsigned h = 0; h < refHyps.size(); h++) {
            jp::jp_trans_t jpTrans = jp::cv2our(refHyps[h]);
            cv::Mat_<double> dLoss_dHyp = dLossMax(Hypothesis(jpTrans.first, jpTrans.second).getRodVecAndTrans(), poseGT.getRodVecAndTrans());
            cv::Mat_<double> gradient = dLoss_dHyp * dHyp_dObjs[h];
            for (int idx = 0; idx < P; idx++) for (int c = 0; c < 3; c++) acc(idx, c) += sf[h] * gradient(0, idx * 3 + c);
        }
        std::vector<cv::Mat_<double>> dS = dSMScore(estObj, sampling, sampled, ls, sf, score_state());
        for (unsigned h = 0; h < hyps.size(); h++) acc += dS[h];
        for (int idx = 0; idx < P; idx++) for (int c = 0; c < 3; c++) dLoss_dObj[idx * 3 + c] = acc(idx, c);
    }
    omp_set_num_threads(saved_threads);
    return 0;
}

}  // extern "C"
