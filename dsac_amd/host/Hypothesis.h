// Hypothesis.h -- pose container of the C++ host shim.  Keeps the public shape of the reference's class
// (core/Hypothesis.h / core/Hypothesis.cpp:31-143,219-289) with plain 3x3 / 3-vector PODs instead of cv::Mat, and
// the pose-convention helpers of core/types.h:137-214.  Host-side double arithmetic only (no GPU, no OpenCV).
#pragma once
#include <array>
#include <cmath>
#include <vector>

namespace dsac {

using Mat3 = std::array<double, 9>;  // row-major
using Vec3 = std::array<double, 3>;
using Pose6 = std::array<double, 6>;  // rvec | tvec (mm)

Mat3 rodrigues(const Vec3& r);                 // cv::Rodrigues, vector -> matrix
Vec3 rodrigues(const Mat3& R);                 // cv::Rodrigues, matrix -> vector (re-orthonormalises first)
Mat3 inverse(const Mat3& A);
Mat3 multiply(const Mat3& A, const Mat3& B);
double determinant(const Mat3& A);

struct cv_trans_t { Vec3 rvec; Vec3 tvec; };    // jp::cv_trans_t, core/types.h:91 (OpenCV convention, mm)
struct jp_trans_t { Mat3 R; Vec3 t; };          // jp::jp_trans_t, core/types.h:92

jp_trans_t cv2our(const cv_trans_t& trans);     // core/types.h:186-214
cv_trans_t our2cv(const jp_trans_t& trans);     // core/types.h:137-151
inline Pose6 pack(const cv_trans_t& p) { return {p.rvec[0], p.rvec[1], p.rvec[2], p.tvec[0], p.tvec[1], p.tvec[2]}; }
inline cv_trans_t unpack(const Pose6& v) { return {{v[0], v[1], v[2]}, {v[3], v[4], v[5]}}; }

class Hypothesis {
public:
    Hypothesis();
    Hypothesis(const Mat3& rot, const Vec3& trans);              // Hypothesis.cpp:38-43
    explicit Hypothesis(const std::vector<double>& rodVecAndTrans);  // Hypothesis.cpp:81-99
    explicit Hypothesis(const jp_trans_t& t) : Hypothesis(t.R, t.t) {}

    void setRotation(const Mat3& rot);
    void setTranslation(const Vec3& trans) { translation = trans; }
    const Mat3& getRotation() const { return rotation; }
    const Mat3& getInvRotation() const { return invRotation; }
    const Vec3& getTranslation() const { return translation; }

    Vec3 transform(const Vec3& p, bool isNormal = false) const;  // Hypothesis.cpp:112-123
    Vec3 invTransform(const Vec3& p) const;                      // Hypothesis.cpp:125-135
    double calcAngularDistance(const Hypothesis& h) const;       // Hypothesis.cpp:137-143 (degrees)
    Hypothesis getInv() const;                                   // Hypothesis.cpp:249-253
    Hypothesis operator*(const Hypothesis& other) const;         // Hypothesis.cpp:255-259
    Vec3 getRodriguesVector() const { return rodrigues(rotation); }
    std::vector<double> getRodVecAndTrans() const;               // Hypothesis.cpp:274-289

private:
    Mat3 rotation, invRotation;
    Vec3 translation;
};

}  // namespace dsac
