// oracle/refbuild/ref_common.h -- TEST INFRASTRUCTURE ONLY.
// Helpers shared by ref_glue.cpp (core/cnn_softam.h) and ref_glue_dsac.cpp (core/cnn.h): the host callbacks that play the
// Torch scripts behind include/lua.hpp, and conversions between flat arrays and the reference's types.  Included AFTER the
// reference headers of the translation unit (it uses jp::, cv:: and CNN_OBJ_PATCHSIZE).
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace {

struct ScoreModel { double tau = 10, beta = 0.5, alpha = 1; const double* explicit_ddiff = nullptr; } g_score;
std::vector<float> g_pred_m;  // stored scene-coordinate "CNN output" in metres, 3 per patch

double sigmoid(double x) { return 1.0 / (1.0 + std::exp(-x)); }

// ---- the "Torch scripts" ------------------------------------------------------------------------------------------
std::vector<LuaValue> score_forward(std::vector<LuaValue>& a) {  // (n, maps[n][y][x]) -> n scores
    const int n = (int)a.at(0).num;
    const std::vector<double>& maps = *a.at(1).tab;
    const size_t P = maps.size() / (size_t)std::max(1, n);
    std::vector<LuaValue> out;
    for (int h = 0; h < n; h++) {
        double s = 0;
        for (size_t p = 0; p < P; p++) s += sigmoid(g_score.beta * (g_score.tau - maps[h * P + p]));
        out.push_back(LuaValue::number(g_score.alpha * s));
    }
    return out;
}
std::vector<LuaValue> score_backward(std::vector<LuaValue>& a) {  // (n, maps, g[n]) -> table [c][row][col]
    const int n = (int)a.at(0).num;
    const std::vector<double>& maps = *a.at(1).tab;
    const std::vector<double>& g = *a.at(2).tab;
    LuaValue t = LuaValue::table(maps.size());
    t.tab->resize(maps.size());
    const size_t P = maps.size() / (size_t)std::max(1, n);
    for (int h = 0; h < n; h++)
        for (size_t p = 0; p < P; p++) {
            if (g_score.explicit_ddiff) { (*t.tab)[h * P + p] = g_score.explicit_ddiff[h * P + p]; continue; }
            const double s = sigmoid(g_score.beta * (g_score.tau - maps[h * P + p]));
            (*t.tab)[h * P + p] = g[h] * g_score.alpha * (-g_score.beta) * s * (1 - s);
        }
    return {t};
}
std::vector<LuaValue> coord_forward(std::vector<LuaValue>& a) {  // (n, patches) -> table of 3n numbers (metres)
    const int n = (int)a.at(0).num;
    if ((size_t)n * 3 > g_pred_m.size()) throw std::runtime_error("ref_glue: more patches than stored predictions");
    LuaValue t = LuaValue::table((size_t)n * 3);
    for (int i = 0; i < n * 3; i++) t.tab->push_back(g_pred_m[i]);
    return {t};
}
lua_State* score_state() {
    static lua_State s;
    s.globals["forward"] = score_forward;
    s.globals["backward"] = score_backward;
    return &s;
}
lua_State* coord_state() {
    static lua_State s;
    s.globals["forward"] = coord_forward;
    return &s;
}

// ---- array <-> reference types ------------------------------------------------------------------------------------
jp::img_coord_t make_obj(const float* xyz, int H, int W) {
    jp::img_coord_t m(H, W);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        const float* p = xyz + ((size_t)y * W + x) * 3;
        m(y, x) = cv::Vec3f(p[0], p[1], p[2]);  // saturating float -> coord1_t conversion (types.h:40-41)
    }
    return m;
}
cv::Mat_<cv::Point2i> make_sampling(const int32_t* uv, int H, int W) {
    cv::Mat_<cv::Point2i> s(H, W);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) s(y, x) = cv::Point2i(uv[((size_t)y * W + x) * 2], uv[((size_t)y * W + x) * 2 + 1]);
    return s;
}
jp::cv_trans_t make_cv(const double* p) {
    cv::Mat r(3, 1, CV_64F), t(3, 1, CV_64F);
    for (int i = 0; i < 3; i++) { r.at<double>(i, 0) = p[i]; t.at<double>(i, 0) = p[3 + i]; }
    return jp::cv_trans_t(r, t);
}
void put_cv(const jp::cv_trans_t& c, double* p) {
    // 3x1 from solvePnP, 1x3 zeros after a failed safeSolvePnP (cnn_softam.h:68-69)
    for (int i = 0; i < 3; i++) { p[i] = c.first.at<double>(i); p[3 + i] = c.second.at<double>(i); }
}
cv::Mat make_R(const double* R) {
    cv::Mat m(3, 3, CV_64F);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m.at<double>(i, j) = R[i * 3 + j];
    return m;
}
std::vector<std::vector<int>> make_perm(const int32_t* perm, int steps, int P) {
    std::vector<std::vector<int>> v(steps);
    for (int s = 0; s < steps; s++) v[s].assign(perm + (size_t)s * P, perm + (size_t)(s + 1) * P);
    return v;
}
void put_mat(const cv::Mat_<double>& m, double* out) {
    for (int i = 0; i < m.rows; i++) for (int j = 0; j < m.cols; j++) out[(size_t)i * m.cols + j] = m(i, j);
}

}  // namespace
