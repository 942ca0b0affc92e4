// oracle/refbuild/minicv_calib.cpp -- TEST INFRASTRUCTURE ONLY.
// cv::Rodrigues / cv::projectPoints / cv::solvePnP of the mini OpenCV stand-in, forwarding to the restated
// algorithms in oracle/cvlike.h (the same ones the oracle restatement uses; PARITY UNPINNED, see cvlike.h).
#include <opencv2/opencv.hpp>
#include "../cvlike.h"

namespace cv {

static cvl::Cam cam_of(const Mat& K) {
    // the reference builds a 3x3 float matrix (properties.cpp:308-323); accept double too
    cvl::Cam c;
    c.fx = K.getd(0, 0); c.fy = K.getd(1, 1); c.cx = K.getd(0, 2); c.cy = K.getd(1, 2);
    return c;
}

static void vec3_of(const Mat& m, double* v) {
    if (m.total() != 3 || m.channels() != 1) throw std::runtime_error("mini-cv: expected a 3-vector");
    for (int i = 0; i < 3; i++) v[i] = m.rows == 3 ? m.getd(i, 0) : m.getd(0, i);
}

void Rodrigues(const Mat& src, OutputArray dst, OutputArray jacobian) {
    if (!dst.needed()) throw std::runtime_error("mini-cv: Rodrigues needs a destination");
    const int depth = src.depth() == CV_32F ? CV_32F : CV_64F;
    if (src.total() == 3) {
        double r[3], R[9], J[27];
        vec3_of(src, r);
        cvl::rodrigues_vec2mat(r, R, jacobian.needed() ? J : nullptr);
        dst.m->create(3, 3, depth);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) dst.m->setd(i, j, R[i * 3 + j]);
        if (jacobian.needed()) {
            jacobian.m->create(3, 9, depth);
            for (int i = 0; i < 3; i++) for (int j = 0; j < 9; j++) jacobian.m->setd(i, j, J[i * 9 + j]);
        }
    } else if (src.rows == 3 && src.cols == 3) {
        if (jacobian.needed()) throw std::runtime_error("mini-cv: Rodrigues(matrix) Jacobian is not used by the reference path");
        double R[9], r[3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = src.getd(i, j);
        cvl::rodrigues_mat2vec(R, r);
        dst.m->create(3, 1, depth);
        for (int i = 0; i < 3; i++) dst.m->setd(i, 0, r[i]);
    } else {
        throw std::runtime_error("mini-cv: Rodrigues input must be 3x1, 1x3 or 3x3");
    }
}

void projectPoints(const std::vector<Point3f>& objectPoints, const Mat& rvec, const Mat& tvec, const Mat& cameraMatrix, const Mat& distCoeffs,
                   std::vector<Point2f>& imagePoints) {
    if (!distCoeffs.empty()) throw std::runtime_error("mini-cv: distortion is not supported");
    double r[3], t[3];
    vec3_of(rvec, r); vec3_of(tvec, t);
    const int n = (int)objectPoints.size();
    static_assert(sizeof(Point3f) == 3 * sizeof(float) && sizeof(Point2f) == 2 * sizeof(float), "packed points");
    imagePoints.resize(n);
    if (n) cvl::project_points(n, &objectPoints[0].x, r, t, cam_of(cameraMatrix), &imagePoints[0].x);
}

bool solvePnP(const std::vector<Point3f>& objectPoints, const std::vector<Point2f>& imagePoints, const Mat& cameraMatrix, const Mat& distCoeffs,
              Mat& rvec, Mat& tvec, bool useExtrinsicGuess, int flags) {
    if (!distCoeffs.empty()) throw std::runtime_error("mini-cv: distortion is not supported");
    const int n = (int)objectPoints.size();
    if (n != (int)imagePoints.size() || n < 4) throw std::runtime_error("mini-cv: solvePnP needs >= 4 matched points");
    const cvl::Cam K = cam_of(cameraMatrix);
    double r[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    bool ok;
    if (flags == CV_P3P) {
        if (n != 4) throw std::runtime_error("mini-cv: P3P needs exactly 4 points");
        ok = cvl::solve_p3p(&objectPoints[0].x, &imagePoints[0].x, K, r, t);
        if (!ok) return false;  // OpenCV leaves rvec/tvec untouched; cnn_softam.h:68-69 then zeroes them
    } else if (flags == CV_ITERATIVE) {
        if (!useExtrinsicGuess) throw std::runtime_error("mini-cv: ITERATIVE without an extrinsic guess is not on the reference path");
        vec3_of(rvec, r); vec3_of(tvec, t);
        ok = cvl::solve_pnp_iterative_guess(n, &objectPoints[0].x, &imagePoints[0].x, K, r, t);
        (void)ok;  // cv::solvePnP(ITERATIVE) always reports success
    } else {
        throw std::runtime_error("mini-cv: unsupported solvePnP method");
    }
    rvec.create(3, 1, CV_64F); tvec.create(3, 1, CV_64F);
    for (int i = 0; i < 3; i++) { rvec.setd(i, 0, r[i]); tvec.setd(i, 0, t[i]); }
    return true;
}

}  // namespace cv
