// oracle/refbuild/ref_glue.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// C entry points (ref_*) around the REAL reference sources, which are #included / compiled where they lie under
// /root/reference/core (see Makefile in this directory; nothing is copied into the repository).  Third-party
// dependencies the reference does not vendor are replaced by stand-ins:
//   OpenCV 2.4 -> include/opencv2/opencv.hpp (Mat semantics) + minicv_calib.cpp (Rodrigues / projectPoints / solvePnP
//                 forwarding to oracle/cvlike.h),
//   Lua/Torch  -> include/lua.hpp (host callbacks): the scene-coordinate CNN returns a stored prediction, the score
//                 CNN is score = alpha * sum_px sigmoid(beta * (tau - err)) with its analytic gradient, or an explicit
//                 gradient image supplied by the caller.
// Everything between those two seams is the reference's own code: getDiffMap, project, dProjectdObj, dProjectdHyp,
// softMax, entropy, dPNP, dScore, refine, dRefineHyp, dRefineObj, processImage (cnn_softam.h), maxLoss, dLossMax,
// getInvHyp (maxloss.h), cv2our / our2cv (types.h), Hypothesis (Hypothesis.cpp), ThreadRand (thread_rand.cpp),
// GlobalProperties (properties.cpp).  ref_train_backward restates the call sequence of train_ransac_softam.cpp:288-394
// (main() itself needs the 7-Scenes data set and png++), calling those real functions.
#include <iostream>
#include <fstream>
#include <cstdint>
#include <unistd.h>
#include <omp.h>

// same include order as core/train_ransac_softam.cpp:32-39
#include "properties.h"
#include "thread_rand.h"
#include "util.h"
#include "stop_watch.h"
#include "dataset.h"
#include "generic_io.h"
#include "lua_calls.h"
#include "cnn_softam.h"

#include "ref_common.h"

extern "C" {

// Constructs the GlobalProperties singleton from a scratch directory (its constructor reads ./sensorTrans.dat,
// properties.cpp:75-86) and fixes the camera: focal length f, image size, principal-point shift.
int ref_init(const char* scratch_dir, float f, int imageWidth, int imageHeight, float xShift, float yShift) {
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) return -1;
    if (chdir(scratch_dir) != 0) return -2;
    {
        std::ofstream o("./sensorTrans.dat", std::ios::binary);
        cv::Mat_<double> eye = cv::Mat_<double>::eye(4, 4);
        jp::write(o, eye);
    }
    std::cout.setstate(std::ios_base::failbit);  // the reference prints progress to stdout; keep test logs clean
    GlobalProperties* gp = GlobalProperties::getInstance();
    if (chdir(cwd) != 0) return -3;
    gp->dP.focalLength = f; gp->dP.imageWidth = imageWidth; gp->dP.imageHeight = imageHeight; gp->dP.xShift = xShift; gp->dP.yShift = yShift;
    return 0;
}
void ref_cam(double* cam4) {  // fx, fy, cx, cy as the reference derives them (properties.cpp:308-323)
    cv::Mat_<float> K = GlobalProperties::getInstance()->getCamMat();
    cam4[0] = K(0, 0); cam4[1] = K(1, 1); cam4[2] = K(0, 2); cam4[3] = K(1, 2);
}
void ref_set_score_model(double tau, double beta, double alpha) { g_score.tau = tau; g_score.beta = beta; g_score.alpha = alpha; }
void ref_seed(unsigned seed) { ThreadRand::forceInit(seed); }

// ---- types.h / Hypothesis.cpp ------------------------------------------------------------------------------------
void ref_cv2our(const double* cv6, double* R, double* t) {
    jp::jp_trans_t j = jp::cv2our(make_cv(cv6));
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) R[i * 3 + k] = j.first.at<double>(i, k);
    t[0] = j.second.x; t[1] = j.second.y; t[2] = j.second.z;
}
void ref_our2cv(const double* R, const double* t, double* cv6) { put_cv(jp::our2cv(jp::jp_trans_t(make_R(R), cv::Point3d(t[0], t[1], t[2]))), cv6); }
void ref_rodvec_and_trans(const double* R, const double* t, double* out6) {
    std::vector<double> v = Hypothesis(make_R(R), cv::Point3d(t[0], t[1], t[2])).getRodVecAndTrans();
    for (int i = 0; i < 6; i++) out6[i] = v[i];
}
void ref_cv_to_jp6(const double* cv6, double* jp6) {  // the conversion idiom of cnn_softam.h:121-122, 721-722
    jp::jp_trans_t j = jp::cv2our(make_cv(cv6));
    std::vector<double> v = Hypothesis(j.first, j.second).getRodVecAndTrans();
    for (int i = 0; i < 6; i++) jp6[i] = v[i];
}

// ---- cnn_softam.h ------------------------------------------------------------------------------------------------
void ref_getDiffMap(const double* cv6, const float* xyz, const int32_t* uv, int H, int W, float* out) {
    cv::Mat_<float> camMat = GlobalProperties::getInstance()->getCamMat();
    cv::Mat_<float> d = getDiffMap(make_cv(cv6), make_obj(xyz, H, W), make_sampling(uv, H, W), camMat);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) out[(size_t)y * W + x] = d(y, x);
}
float ref_project(const float* pt, const float* obj, const double* R, const double* t) {
    cv::Mat_<float> camMat = GlobalProperties::getInstance()->getCamMat();
    return project(cv::Point2f(pt[0], pt[1]), cv::Point3f(obj[0], obj[1], obj[2]), make_R(R), cv::Point3d(t[0], t[1], t[2]), camMat);
}
void ref_dProjectdObj(const float* pt, const float* obj, const double* R, const double* t, double* J3) {
    cv::Mat_<float> camMat = GlobalProperties::getInstance()->getCamMat();
    put_mat(dProjectdObj(cv::Point2f(pt[0], pt[1]), cv::Point3f(obj[0], obj[1], obj[2]), make_R(R), cv::Point3d(t[0], t[1], t[2]), camMat), J3);
}
void ref_dProjectdHyp(const float* pt, const float* obj, const double* R, const double* t, double* J6) {
    cv::Mat_<float> camMat = GlobalProperties::getInstance()->getCamMat();
    put_mat(dProjectdHyp(cv::Point2f(pt[0], pt[1]), cv::Point3f(obj[0], obj[1], obj[2]), make_R(R), cv::Point3d(t[0], t[1], t[2]), camMat), J6);
}
void ref_softMax(int n, const double* s, double* w) {
    std::vector<double> r = softMax(std::vector<double>(s, s + n));
    for (int i = 0; i < n; i++) w[i] = r[i];
}
double ref_entropy(int n, const double* w) { return entropy(std::vector<double>(w, w + n)); }
void ref_dPNP(const float* uv4, const float* X4, float eps, double* J72) {
    std::vector<cv::Point2f> ip; std::vector<cv::Point3f> op;
    for (int i = 0; i < 4; i++) { ip.push_back(cv::Point2f(uv4[2 * i], uv4[2 * i + 1])); op.push_back(cv::Point3f(X4[3 * i], X4[3 * i + 1], X4[3 * i + 2])); }
    put_mat(dPNP(ip, op, eps), J72);
}
int ref_safeSolveP3P(const float* uv4, const float* X4, double* cv6) {
    std::vector<cv::Point2f> ip; std::vector<cv::Point3f> op;
    for (int i = 0; i < 4; i++) { ip.push_back(cv::Point2f(uv4[2 * i], uv4[2 * i + 1])); op.push_back(cv::Point3f(X4[3 * i], X4[3 * i + 1], X4[3 * i + 2])); }
    cv::Mat_<float> camMat = GlobalProperties::getInstance()->getCamMat();
    jp::cv_trans_t h;
    const bool ok = safeSolvePnP(op, ip, camMat, cv::Mat(), h.first, h.second, false, CV_P3P);
    put_cv(h, cv6);
    return ok ? 1 : 0;
}

// refine (cnn_softam.h:663-723): perm is refSteps x (H*W); output is the jp 6-vector the reference returns
void ref_refine(int inlierCount, int refSteps, float thr, const int32_t* perm, const float* xyz, const int32_t* uv, int H, int W,
                const double* init_cv6, double* out_jp6) {
    cv::Mat_<float> camMat = GlobalProperties::getInstance()->getCamMat();
    std::vector<double> r = refine(inlierCount, refSteps, thr, make_perm(perm, refSteps, H * W), make_obj(xyz, H, W), make_sampling(uv, H, W), camMat, make_cv(init_cv6));
    for (int i = 0; i < 6; i++) out_jp6[i] = r[i];
}
void ref_dRefineHyp(int inlierCount, int refSteps, float thr, const int32_t* perm, const float* xyz, const int32_t* uv, int H, int W,
                    const double* init_cv6, double* J36) {
    cv::Mat_<float> camMat = GlobalProperties::getInstance()->getCamMat();
    put_mat(dRefineHyp(inlierCount, refSteps, thr, make_perm(perm, refSteps, H * W), make_obj(xyz, H, W), make_sampling(uv, H, W), camMat, make_cv(init_cv6)), J36);
}
// J is 6 x (3*H*W) in the reference's own column order y*CNN_OBJ_PATCHSIZE*3 + x*3 + c (needs W == 40 to be dense)
void ref_dRefineObj(int inlierCount, int refSteps, float subSample, float thr, const int32_t* perm, const float* xyz, const int32_t* uv, int H, int W,
                    const double* init_cv6, const int32_t* inlierMap, double* J) {
    cv::Mat_<float> camMat = GlobalProperties::getInstance()->getCamMat();
    cv::Mat_<int> im(H, W);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) im(y, x) = inlierMap[(size_t)y * W + x];
    put_mat(dRefineObj(inlierCount, refSteps, subSample, thr, make_perm(perm, refSteps, H * W), make_obj(xyz, H, W), make_sampling(uv, H, W), camMat, make_cv(init_cv6), im), J);
}

// dScore (cnn_softam.h:564-646) on a 40x40 map.  points: N x 4 x (x, y).  Either ddiff (N x 1600, what the score
// script returns, flattened [hyp][row][col]) or, if NULL, g (N) through the analytic soft-inlier backward.
// jac: N x 4800, the reference's 1 x 3n layout per hypothesis.
void ref_dScore(int N, const int32_t* points, const double* ddiff_or_null, const double* g, const float* xyz, const int32_t* uv, double* jac) {
    const int S = CNN_OBJ_PATCHSIZE;
    std::vector<std::vector<cv::Point2i>> pts(N);
    for (int h = 0; h < N; h++) for (int i = 0; i < 4; i++) pts[h].push_back(cv::Point2i(points[(h * 4 + i) * 2], points[(h * 4 + i) * 2 + 1]));
    std::vector<cv::Mat_<double>> J;
    g_score.explicit_ddiff = ddiff_or_null;
    std::vector<double> grads(N, 0.0);
    if (g) grads.assign(g, g + N);
    dScore(make_obj(xyz, S, S), make_sampling(uv, S, S), pts, score_state(), J, grads);
    g_score.explicit_ddiff = nullptr;
    for (int h = 0; h < N; h++) put_mat(J[h], jac + (size_t)h * S * S * 3);
}

// ---- producer side: stochasticSubSample (cnn_softam.h:283-309) and the patch assembly of getCoordImg (:224-254) ------------
// sampling_xy: 1600 x (x, y); patches: n x 3 x 42 x 42 floats in the order pushMaps would hand them to Lua (lua_calls.h:63-80).
// Returns the number of patches (border positions are skipped by the reference).
int ref_subsample_and_patches(unsigned seed, const unsigned char* bgr, int32_t* sampling_xy, float* patches) {
    const int S = CNN_OBJ_PATCHSIZE, PS = CNN_RGB_PATCHSIZE;
    GlobalProperties* gp = GlobalProperties::getInstance();
    const int saved_threads = omp_get_max_threads();
    omp_set_num_threads(1);
    ThreadRand::forceInit(seed);
    jp::img_bgr_t img(gp->dP.imageHeight, gp->dP.imageWidth);
    for (int y = 0; y < img.rows; y++) for (int x = 0; x < img.cols; x++)
        img(y, x) = jp::bgr_t(bgr[((size_t)y * img.cols + x) * 3], bgr[((size_t)y * img.cols + x) * 3 + 1], bgr[((size_t)y * img.cols + x) * 3 + 2]);
    cv::Mat_<cv::Point2i> sampling = stochasticSubSample(img, S, PS);
    for (int y = 0; y < S; y++) for (int x = 0; x < S; x++) { sampling_xy[2 * (y * S + x)] = sampling(y, x).x; sampling_xy[2 * (y * S + x) + 1] = sampling(y, x).y; }
    g_pred_m.assign((size_t)S * S * 3, 0.f);
    std::vector<cv::Mat_<cv::Vec3f>> pv;
    getCoordImg(img, sampling, PS, pv, coord_state());
    for (size_t n = 0; n < pv.size(); n++)
        for (int c = 0; c < 3; c++) for (int y = 0; y < PS; y++) for (int x = 0; x < PS; x++)
            patches[((n * 3 + c) * PS + y) * PS + x] = pv[n](y, x)[c];
    omp_set_num_threads(saved_threads);
    return (int)pv.size();
}

double ref_maxLoss(const double* R1, const double* t1, const double* R2, const double* t2) {
    return maxLoss(Hypothesis(make_R(R1), cv::Point3d(t1[0], t1[1], t1[2])), Hypothesis(make_R(R2), cv::Point3d(t2[0], t2[1], t2[2])));
}
void ref_dLossMax(const double* est6, const double* gt6, double* J6) {
    put_mat(dLossMax(std::vector<double>(est6, est6 + 6), std::vector<double>(gt6, gt6 + 6)), J6);
}

// ---- processImage (cnn_softam.h:960-1180) + the backward pass of train_ransac_softam.cpp:288-394 -------------------
// pred_mm: 1600 x 3 scene coordinates (mm) the stand-in coordinate CNN returns for the 40x40 stochastic sub-sampling.
// gt_jp6: ground-truth pose (rodrigues vector, translation in mm) as poseGT.getRodVecAndTrans() would give it.
// Runs single-threaded so that ThreadRand's per-thread generators give one reproducible stream.
struct RefFrameOut {
    double loss, sfEntropy, tErr, rotErr;
    int correct, n_hyps, ref_steps;
};
// OpenMP threads of the next ref_processImage calls (default 1: ThreadRand's per-thread generators then give ONE reproducible stream; T > 1: the reference's
// own `#pragma omp parallel for` over the hypotheses with its static schedule, generator t = mt19937(seed + t) -- deterministic as well, and what
// dsac_sample_refstream reproduces for T threads)
static int g_ref_threads = 1;
void ref_set_omp_threads(int n) { g_ref_threads = n < 1 ? 1 : n; }
int ref_processImage(unsigned seed, int objHyps, int inlierThreshold2D, int inlierCount, int refSteps, const float* pred_mm, const double* gt_jp6,
                     RefFrameOut* out, double* hyps_cv6 /*N x 6*/, int32_t* sampledPoints /*N x 4 x 2*/, double* sfScores /*N*/, double* avg_cv6, double* ref_cv6,
                     int32_t* sampling_uv /*1600 x 2*/, float* estObj_mm /*1600 x 3*/, int32_t* inlierMap /*1600*/, int32_t* pixelIdxs /*refSteps x 1600*/,
                     double* dLoss_dObj /*1600 x 3 or NULL: run the backward pass too*/, float refSubSample) {
    const int S = CNN_OBJ_PATCHSIZE, P = S * S;
    const int saved_threads = omp_get_max_threads();
    omp_set_num_threads(g_ref_threads);
    ThreadRand::forceInit(seed);
    GlobalProperties* gp = GlobalProperties::getInstance();
    cv::Mat_<float> camMat = gp->getCamMat();
    g_pred_m.resize((size_t)P * 3);
    for (int i = 0; i < P * 3; i++) g_pred_m[i] = pred_mm[i] / 1000.f;
    jp::img_bgr_t img = jp::img_bgr_t::zeros(gp->dP.imageHeight, gp->dP.imageWidth);
    Hypothesis poseGT(std::vector<double>(gt_jp6, gt_jp6 + 6));

    double loss, sfEntropy, tErr, rotErr; bool correct;
    std::vector<jp::cv_trans_t> hyps; jp::cv_trans_t refAvgHyp, avgHyp;
    std::vector<std::vector<cv::Point2f>> imgPts; std::vector<std::vector<cv::Point3f>> objPts; std::vector<std::vector<int>> imgIdx;
    std::vector<cv::Mat_<cv::Vec3f>> patches; std::vector<double> sf; jp::img_coord_t estObj; cv::Mat_<cv::Point2i> sampling;
    std::vector<std::vector<cv::Point2i>> sampled; cv::Mat_<int> inl; std::vector<std::vector<int>> pix;
    processImage(img, poseGT, coord_state(), score_state(), objHyps, 4, camMat, inlierThreshold2D, inlierCount, refSteps, loss, sfEntropy, correct, hyps,
                 refAvgHyp, avgHyp, imgPts, objPts, imgIdx, patches, sf, estObj, sampling, sampled, inl, pix, tErr, rotErr);

    out->loss = loss; out->sfEntropy = sfEntropy; out->tErr = tErr; out->rotErr = rotErr; out->correct = correct; out->n_hyps = (int)hyps.size();
    out->ref_steps = (int)pix.size();
    for (int h = 0; h < objHyps; h++) {
        put_cv(hyps[h], hyps_cv6 + 6 * h);
        for (int i = 0; i < 4; i++) { sampledPoints[(h * 4 + i) * 2] = sampled[h][i].x; sampledPoints[(h * 4 + i) * 2 + 1] = sampled[h][i].y; }
        sfScores[h] = sf[h];
    }
    put_cv(avgHyp, avg_cv6); put_cv(refAvgHyp, ref_cv6);
    for (int y = 0; y < S; y++) for (int x = 0; x < S; x++) {
        const int p = y * S + x;
        sampling_uv[2 * p] = sampling(y, x).x; sampling_uv[2 * p + 1] = sampling(y, x).y;
        for (int c = 0; c < 3; c++) estObj_mm[3 * p + c] = estObj(y, x)[c];
        inlierMap[p] = inl(y, x);
    }
    for (int s = 0; s < refSteps; s++) for (int i = 0; i < P; i++) pixelIdxs[(size_t)s * P + i] = (i < (int)pix[s].size()) ? pix[s][i] : -1;

    if (dLoss_dObj) {
        // train_ransac_softam.cpp:294-394, same statements on the same variables
        cv::Mat_<double> dLoss_dObj_1row = cv::Mat_<double>::zeros(1, patches.size() * 3);
        jp::jp_trans_t refAvgHypJP = jp::cv2our(refAvgHyp);
        cv::Mat_<double> dLoss_dRAvgHyp = dLossMax(Hypothesis(refAvgHypJP.first, refAvgHypJP.second).getRodVecAndTrans(), poseGT.getRodVecAndTrans());
        cv::Mat_<double> dRAvgHyp_dObj = dRefineObj(inlierCount, refSteps, refSubSample, inlierThreshold2D, pix, estObj, sampling, camMat, avgHyp, inl);
        dLoss_dObj_1row += dLoss_dRAvgHyp * dRAvgHyp_dObj;
        cv::Mat_<double> dRAvgHyp_dHyp = dRefineHyp(inlierCount, refSteps, inlierThreshold2D, pix, estObj, sampling, camMat, avgHyp);
        cv::Mat_<double> dHyp_dObj_complete = cv::Mat_<double>::zeros(6, patches.size() * 3);
        for (unsigned h = 0; h < hyps.size(); h++) {
            cv::Mat_<double> dHyp_dObj = sf[h] * dPNP(imgPts[h], objPts[h]);
            for (unsigned i = 0; i < imgIdx[h].size(); i++)
                dHyp_dObj_complete.colRange(imgIdx[h][i] * 3, imgIdx[h][i] * 3 + 3) += dHyp_dObj.colRange(i * 3, i * 3 + 3);
        }
        dLoss_dObj_1row += dLoss_dRAvgHyp * dRAvgHyp_dHyp * dHyp_dObj_complete;
        std::vector<double> scoreOutputGradients(hyps.size(), 0);
        for (unsigned h = 0; h < hyps.size(); h++) {
            cv::Mat_<double> hypMat(6, 1);
            hyps[h].first.copyTo(hypMat.rowRange(0, 3));
            hyps[h].second.copyTo(hypMat.rowRange(3, 6));
            for (int k = 3; k < 6; k++) hypMat(k, 0) /= 1000;  // hypMat.rowRange(3, 6) /= 1000
            hypMat = dLoss_dRAvgHyp * dRAvgHyp_dHyp * hypMat;
            double hypFactor = hypMat.at<double>(0, 0);
            scoreOutputGradients[h] += sf[h] * hypFactor;
            for (unsigned j = 0; j < hyps.size(); j++) scoreOutputGradients[j] -= sf[h] * sf[j] * hypFactor;
        }
        std::vector<cv::Mat_<double>> dLoss_dScores;
        dScore(estObj, sampling, sampled, score_state(), dLoss_dScores, scoreOutputGradients);
        for (unsigned h = 0; h < hyps.size(); h++) dLoss_dObj_1row += dLoss_dScores[h];
        for (int idx = 0; idx < P; idx++) for (int c = 0; c < 3; c++) dLoss_dObj[idx * 3 + c] = dLoss_dObj_1row(0, idx * 3 + c);
    }
    omp_set_num_threads(saved_threads);
    return 0;
}

}  // extern "C"
