// frame_io.cpp -- see frame_io.h.
#include "frame_io.h"

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>

#include "properties.h"

namespace dsac {

namespace {

// 4 x 4 float inverse by Gauss-Jordan with partial pivoting (cv::Mat_<float>::inv() is an LU solve in float)
bool invert4f(const float a[16], float out[16]) {
    float m[4][8];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) { m[r][c] = a[r * 4 + c]; m[r][4 + c] = (r == c) ? 1.f : 0.f; }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (std::fabs(m[r][col]) > std::fabs(m[piv][col])) piv = r;
        if (m[piv][col] == 0.f) return false;
        if (piv != col)
            for (int c = 0; c < 8; c++) std::swap(m[piv][c], m[col][c]);
        const float d = 1.f / m[col][col];
        for (int c = 0; c < 8; c++) m[col][c] *= d;
        for (int r = 0; r < 4; r++) {
            if (r == col) continue;
            const float f = m[r][col];
            for (int c = 0; c < 8; c++) m[r][c] -= f * m[col][c];
        }
    }
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out[r * 4 + c] = m[r][4 + c];
    return true;
}

bool readTranslationTxt(double t[3]) {
    std::ifstream transFile("translation.txt");
    if (!transFile.is_open()) return false;
    std::string line;
    std::getline(transFile, line);
    const std::vector<std::string> tokens = split(line);
    if (tokens.size() < 3) return false;
    for (int i = 0; i < 3; i++) t[i] = std::atof(tokens[i].c_str());
    return true;
}

std::vector<std::string> listFiles(const std::string& dir, const std::string& ext) {
    std::vector<std::string> out;
    DIR* d = opendir(dir.c_str());
    if (!d) return out;
    while (dirent* e = readdir(d)) {
        const std::string n = e->d_name;
        if (n.size() > ext.size() && n.compare(n.size() - ext.size(), ext.size(), ext) == 0) out.push_back(dir + n);
    }
    closedir(d);
    std::sort(out.begin(), out.end());
    return out;
}

std::string stemOf(const std::string& path) {
    const size_t s = path.find_last_of('/');
    const std::string n = s == std::string::npos ? path : path.substr(s + 1);
    const size_t d = n.find('.');
    return d == std::string::npos ? n : n.substr(0, d);
}

}  // namespace

bool readPose7Scenes(const std::string& infoFile, Hypothesis& out) {
    out = Hypothesis();
    std::ifstream file(infoFile);
    if (!file.is_open()) return false;
    float trans[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::string line;
    for (unsigned i = 0; i < 3; i++) {
        std::getline(file, line);
        const std::vector<std::string> tokens = split(line);
        if (tokens.size() < 4) return false;
        for (int c = 0; c < 4; c++) trans[i * 4 + c] = (float)std::atof(tokens[c].c_str());
    }
    double t[3];
    if (readTranslationTxt(t)) {
        for (int i = 0; i < 3; i++) trans[i * 4 + 3] -= (float)t[i];
    } else {
        std::cout << "WARNING! Cannot open translation.txt" << std::endl;
    }
    // correction for 7-scene poses (different coordinate frame definition): trans = trans * diag(1, -1, -1, 1), then invert
    for (int r = 0; r < 4; r++) { trans[r * 4 + 1] = -trans[r * 4 + 1]; trans[r * 4 + 2] = -trans[r * 4 + 2]; }
    float inv[16];
    if (!invert4f(trans, inv)) return false;
    Mat3 R;
    for (int y = 0; y < 3; y++)
        for (int x = 0; x < 3; x++) R[y * 3 + x] = (double)inv[y * 4 + x];
    const Vec3 tr = {(double)inv[3] * 1e3, (double)inv[7] * 1e3, (double)inv[11] * 1e3};  // metres -> mm, Hypothesis.cpp:53
    out = Hypothesis(R, tr);
    return true;
}

std::array<double, 16> poseTo7ScenesMatrix(const Hypothesis& h) {
    // readPose7Scenes computes M = (T * C)^-1 with M = [R | t / 1000]; hence T = M^-1 * C  (C = diag(1,-1,-1,1) is its own inverse)
    const Mat3& Ri = h.getInvRotation();
    const Vec3& t = h.getTranslation();
    std::array<double, 16> T{};
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[r * 4 + c] = Ri[r * 3 + c] * (c == 0 ? 1.0 : -1.0);
        T[r * 4 + 3] = -(Ri[r * 3] * t[0] + Ri[r * 3 + 1] * t[1] + Ri[r * 3 + 2] * t[2]) / 1000.0;
    }
    T[15] = 1.0;
    return T;
}

std::vector<double> exportPose7Scenes(const cv_trans_t& refAvgHyp) {
    const jp_trans_t jp = cv2our(refAvgHyp);
    const Hypothesis hyp(jp.R, jp.t);
    // hypTrans = [R | t]^-1 * diag(1, -1, -1, 1): rotation R^-1 with columns 1, 2 negated, translation -R^-1 t
    const Mat3& Ri = hyp.getInvRotation();
    const Vec3& t = hyp.getTranslation();
    Mat3 R2;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R2[r * 3 + c] = Ri[r * 3 + c] * (c == 0 ? 1.0 : -1.0);
    const Vec3 t2 = {-(Ri[0] * t[0] + Ri[1] * t[1] + Ri[2] * t[2]), -(Ri[3] * t[0] + Ri[4] * t[1] + Ri[5] * t[2]),
                     -(Ri[6] * t[0] + Ri[7] * t[1] + Ri[8] * t[2])};
    const Hypothesis out(R2, t2);
    std::vector<double> hypV = out.getRodVecAndTrans();
    hypV[3] /= 1000; hypV[4] /= 1000; hypV[5] /= 1000;  // translation in m
    double tt[3];
    if (readTranslationTxt(tt)) { hypV[3] += tt[0]; hypV[4] += tt[1]; hypV[5] += tt[2]; }
    return hypV;
}

std::vector<std::string> getSubPaths(const std::string& basePath) {
    std::vector<std::string> out;
    DIR* d = opendir(basePath.c_str());
    if (!d) return out;
    while (dirent* e = readdir(d)) {
        const std::string n = e->d_name;
        if (n == "." || n == "..") continue;
        const std::string p = basePath + n + "/";
        struct stat st;
        if (stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) out.push_back(p);
    }
    closedir(d);
    std::sort(out.begin(), out.end());
    return out;
}

static bool readCoordsFile(const std::string& path, DriverFrame& f) {
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) return false;
    char magic[8];
    int32_t hdr[3];
    in.read(magic, 8);
    in.read(reinterpret_cast<char*>(hdr), sizeof(hdr));
    if (!in || std::memcmp(magic, "DSACCRD1", 8) != 0 || hdr[0] <= 0 || hdr[1] <= 0) return false;
    f.H = hdr[0]; f.W = hdr[1];
    const size_t P = (size_t)f.H * f.W;
    f.estObj.resize(P * 3);
    in.read(reinterpret_cast<char*>(f.estObj.data()), (std::streamsize)(P * 3 * sizeof(float)));
    if (hdr[2]) {
        f.sampling.resize(P * 2);
        in.read(reinterpret_cast<char*>(f.sampling.data()), (std::streamsize)(P * 2 * sizeof(float)));
    }
    return (bool)in;
}

bool writeCoordsFile(const std::string& path, const DriverFrame& f) {
    std::ofstream out(path, std::ios::binary);
    if (!out.is_open()) return false;
    const int32_t hdr[3] = {f.H, f.W, f.sampling.empty() ? 0 : 1};
    out.write("DSACCRD1", 8);
    out.write(reinterpret_cast<const char*>(hdr), sizeof(hdr));
    out.write(reinterpret_cast<const char*>(f.estObj.data()), (std::streamsize)(f.estObj.size() * sizeof(float)));
    if (!f.sampling.empty()) out.write(reinterpret_cast<const char*>(f.sampling.data()), (std::streamsize)(f.sampling.size() * sizeof(float)));
    return (bool)out;
}

std::vector<DriverFrame> loadFrames(const std::string& splitDir) {
    const std::vector<std::string> scenes = getSubPaths(splitDir);
    if (scenes.empty()) throw Error(DSAC_ERR_INVALID, "no scene directory below " + splitDir + " (expected <scene>/coords/*.coords and <scene>/poses/*.txt; or pass -synth K)");
    const std::string scene = scenes[0];  // the reference uses the first scene only: jp::Dataset(testSets[0], 1)
    const std::vector<std::string> coords = listFiles(scene + "coords/", ".coords");
    const std::vector<std::string> poses = listFiles(scene + "poses/", ".txt");
    if (coords.empty()) throw Error(DSAC_ERR_INVALID, "no .coords files in " + scene + "coords/");
    std::vector<DriverFrame> out;
    for (size_t i = 0; i < coords.size(); i++) {
        DriverFrame f;
        f.name = stemOf(coords[i]);
        if (!readCoordsFile(coords[i], f)) throw Error(DSAC_ERR_INVALID, "cannot read " + coords[i]);
        // the pose belongs to the frame by NAME (7-Scenes: frame-000123.coords <-> frame-000123.pose.txt; the reference's dataset couples them by
        // file stem, core/dataset.h): a missing or extra pose file must not shift every later ground truth onto the wrong frame
        f.havePose = false;
        for (const std::string& pf : poses) {
            std::string ps = stemOf(pf);
            const size_t dot = ps.find(".pose");
            if (dot != std::string::npos) ps.erase(dot);
            if (ps == f.name) { f.havePose = readPose7Scenes(pf, f.poseGT); break; }
        }
        std::ifstream sets(scene + "replay/" + f.name + ".sets");
        if (sets.is_open()) {
            std::array<int32_t, 4> s;
            while (sets >> s[0] >> s[1] >> s[2] >> s[3]) f.sets.push_back(s);
        }
        std::ifstream perm(scene + "replay/" + f.name + ".perm", std::ios::binary);
        if (perm.is_open()) {
            int32_t hdr[2];
            perm.read(reinterpret_cast<char*>(hdr), sizeof(hdr));
            if (perm && hdr[0] > 0 && hdr[1] == f.H * f.W) {
                f.permSteps = hdr[0];
                f.pixelIdxs.resize((size_t)hdr[0] * hdr[1]);
                perm.read(reinterpret_cast<char*>(f.pixelIdxs.data()), (std::streamsize)(f.pixelIdxs.size() * sizeof(int32_t)));
                if (!perm) { f.pixelIdxs.clear(); f.permSteps = 0; }
            }
        }
        out.push_back(std::move(f));
    }
    return out;
}

DriverFrame synthFrame(int H, int W, const Camera& cam, unsigned long long seed) {
    DriverFrame f;
    f.name = "synth-" + intToString((int)(seed % 1000000), 6);
    f.H = H; f.W = W;
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0, 1);
    std::normal_distribution<double> G(0, 1);
    // ground truth: rotation <= 30 deg about a random axis, t in U[-1,1]^3 m + (0,0,2.5) m  (SURVEY.md 8(d), config 1 stand-in)
    Vec3 axis = {G(rng), G(rng), G(rng)};
    const double an = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    const double ang = U(rng) * 30.0 * M_PI / 180.0;
    const cv_trans_t gt = {{axis[0] / an * ang, axis[1] / an * ang, axis[2] / an * ang},
                           {(2 * U(rng) - 1) * 1000, (2 * U(rng) - 1) * 1000, (2 * U(rng) - 1) * 1000 + 2500}};
    const Mat3 Ri = inverse(rodrigues(gt.rvec));
    const size_t P = (size_t)H * W;
    f.estObj.resize(P * 3);
    const bool full = (H == 480 && W == 640);
    if (!full) f.sampling.resize(P * 2);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            // stratified positions in the manner of stochasticSubSample (core/cnn_softam.h:283-309) for sub-sampled maps
            const double u = full ? x : std::floor(21 + (x + U(rng)) * (640.0 - 42) / W), v = full ? y : std::floor(21 + (y + U(rng)) * (480.0 - 42) / H);
            const double d = 800 + 2700 * U(rng);
            const Vec3 Xc = {(u - cam.cx) / cam.fx * d - gt.tvec[0], (v - cam.cy) / cam.fy * d - gt.tvec[1], d - gt.tvec[2]};
            const bool outlier = U(rng) < 0.3;
            for (int c = 0; c < 3; c++) {
                double X = Ri[c * 3] * Xc[0] + Ri[c * 3 + 1] * Xc[1] + Ri[c * 3 + 2] * Xc[2];
                X = outlier ? 4000 * (U(rng) - 0.5) : X + 20 * G(rng);
                f.estObj[p * 3 + c] = (float)X;
            }
            if (!full) { f.sampling[p * 2] = (float)u; f.sampling[p * 2 + 1] = (float)v; }
        }
    f.poseGT = Hypothesis(cv2our(gt));
    f.havePose = true;
    return f;
}

void meanStdDev(const std::vector<double>& v, double& mean, double& stddev) {
    mean = stddev = 0;
    if (v.empty()) return;
    for (double x : v) mean += x;
    mean /= (double)v.size();
    for (double x : v) stddev += (x - mean) * (x - mean);
    stddev = std::sqrt(stddev / (double)v.size());
}

double medianOf(std::vector<double> v) {
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

}  // namespace dsac
