// Hypothesis.cpp -- see Hypothesis.h.  Host-only math of the shim.
#include "Hypothesis.h"

#include <algorithm>
#include <cfloat>

namespace dsac {

double determinant(const Mat3& A) {
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

Mat3 inverse(const Mat3& A) {
    const double id = 1.0 / determinant(A);
    return {(A[4] * A[8] - A[5] * A[7]) * id, (A[2] * A[7] - A[1] * A[8]) * id, (A[1] * A[5] - A[2] * A[4]) * id,
            (A[5] * A[6] - A[3] * A[8]) * id, (A[0] * A[8] - A[2] * A[6]) * id, (A[2] * A[3] - A[0] * A[5]) * id,
            (A[3] * A[7] - A[4] * A[6]) * id, (A[1] * A[6] - A[0] * A[7]) * id, (A[0] * A[4] - A[1] * A[3]) * id};
}

Mat3 multiply(const Mat3& A, const Mat3& B) {
    Mat3 C{};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    return C;
}

Mat3 rodrigues(const Vec3& r) {
    const double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) return {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double c = std::cos(theta), s = std::sin(theta), c1 = 1 - c;
    const double x = r[0] / theta, y = r[1] / theta, z = r[2] / theta;
    return {c + c1 * x * x, c1 * x * y - s * z, c1 * x * z + s * y,
            c1 * x * y + s * z, c + c1 * y * y, c1 * y * z - s * x,
            c1 * x * z - s * y, c1 * y * z + s * x, c + c1 * z * z};
}

// polar factor by Newton iteration X <- (X + X^-T)/2: the nearest rotation, what OpenCV gets from U*V^T
static Mat3 orthonormalise(const Mat3& A) {
    Mat3 X = A;
    for (int it = 0; it < 30; it++) {
        const Mat3 Xi = inverse(X);
        Mat3 N{};
        double diff = 0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                N[i * 3 + j] = 0.5 * (X[i * 3 + j] + Xi[j * 3 + i]);
                diff = std::max(diff, std::fabs(N[i * 3 + j] - X[i * 3 + j]));
            }
        X = N;
        if (diff < 1e-15) break;
    }
    return X;
}

Vec3 rodrigues(const Mat3& Rin) {
    for (double v : Rin)
        if (!(v > -100 && v < 100)) return {0, 0, 0};
    const Mat3 R = orthonormalise(Rin);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) return {0, 0, 0};
        double t = (R[0] + 1) * 0.5;
        rx = std::sqrt(std::max(t, 0.));
        t = (R[4] + 1) * 0.5;
        ry = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5;
        rz = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
        if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
        return {rx * theta, ry * theta, rz * theta};
    }
    const double vth = theta / (2 * s);
    return {rx * vth, ry * vth, rz * vth};
}

jp_trans_t cv2our(const cv_trans_t& trans) {
    Mat3 R = rodrigues(trans.rvec);
    Vec3 t = {trans.tvec[0], -trans.tvec[1], -trans.tvec[2]};
    for (int j = 3; j < 9; j++) R[j] = -R[j];
    if (determinant(R) < 0) {  // result may be reconstructed behind the camera
        for (double& v : R) v = -v;
        for (double& v : t) v = -v;
    }
    if (t[0] != t[0] || t[1] != t[1] || t[2] != t[2]) t = {0, 0, 0};
    return {R, t};
}

cv_trans_t our2cv(const jp_trans_t& trans) {
    Mat3 R = trans.R;
    for (int j = 3; j < 9; j++) R[j] = -R[j];
    return {rodrigues(R), {trans.t[0], -trans.t[1], -trans.t[2]}};
}

Hypothesis::Hypothesis() : rotation{1, 0, 0, 0, 1, 0, 0, 0, 1}, invRotation{1, 0, 0, 0, 1, 0, 0, 0, 1}, translation{0, 0, 0} {}

Hypothesis::Hypothesis(const Mat3& rot, const Vec3& trans) : rotation(rot), invRotation(inverse(rot)), translation(trans) {}

Hypothesis::Hypothesis(const std::vector<double>& v) {
    translation = {v.at(3), v.at(4), v.at(5)};
    const double length = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    rotation = (length > 1e-5) ? rodrigues(Vec3{v[0], v[1], v[2]}) : Mat3{1, 0, 0, 0, 1, 0, 0, 0, 1};
    invRotation = inverse(rotation);
}

void Hypothesis::setRotation(const Mat3& rot) {
    rotation = rot;
    invRotation = inverse(rot);
}

Vec3 Hypothesis::transform(const Vec3& p, bool isNormal) const {
    Vec3 o;
    for (int i = 0; i < 3; i++) o[i] = rotation[i * 3] * p[0] + rotation[i * 3 + 1] * p[1] + rotation[i * 3 + 2] * p[2] + (isNormal ? 0.0 : translation[i]);
    return o;
}

Vec3 Hypothesis::invTransform(const Vec3& p) const {
    const Vec3 q = {p[0] - translation[0], p[1] - translation[1], p[2] - translation[2]};
    Vec3 o;
    for (int i = 0; i < 3; i++) o[i] = invRotation[i * 3] * q[0] + invRotation[i * 3 + 1] * q[1] + invRotation[i * 3 + 2] * q[2];
    return o;
}

double Hypothesis::calcAngularDistance(const Hypothesis& h) const {
    const Mat3 D = multiply(rotation, h.invRotation);
    double trace = D[0] + D[4] + D[8];
    trace = std::min(3.0, std::max(-1.0, trace));
    return 180 * std::acos((trace - 1.0) / 2.0) / 3.14159265358979323846;
}

Hypothesis Hypothesis::getInv() const {
    const Vec3 ti = {-(invRotation[0] * translation[0] + invRotation[1] * translation[1] + invRotation[2] * translation[2]),
                     -(invRotation[3] * translation[0] + invRotation[4] * translation[1] + invRotation[5] * translation[2]),
                     -(invRotation[6] * translation[0] + invRotation[7] * translation[1] + invRotation[8] * translation[2])};
    return Hypothesis(invRotation, ti);
}

Hypothesis Hypothesis::operator*(const Hypothesis& o) const { return Hypothesis(multiply(rotation, o.rotation), transform(o.translation)); }

std::vector<double> Hypothesis::getRodVecAndTrans() const {
    const Vec3 rv = rodrigues(rotation);
    return {rv[0], rv[1], rv[2], translation[0], translation[1], translation[2]};
}

}  // namespace dsac
