// properties.cpp -- see properties.h.
#include "properties.h"

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

namespace dsac {

GlobalProperties* GlobalProperties::getInstance() {
    static GlobalProperties instance;
    return &instance;
}

std::vector<std::string> split(const std::string& s) {
    std::istringstream iss(s);
    std::vector<std::string> out;
    std::string tok;
    while (iss >> tok) out.push_back(tok);
    return out;
}

std::string intToString(int number, int minLength) {
    std::string s = std::to_string(number);
    while ((int)s.size() < minLength) s = "0" + s;
    return s;
}

bool GlobalProperties::readArguments(std::vector<std::string> argv) {
    const int argc = (int)argv.size();
    for (int i = 0; i < argc; i++) {
        const std::string s = argv[i];
        auto next = [&]() -> std::string { i++; return i < argc ? argv[i] : std::string(); };
        // the reference's keys, in its order (core/properties.cpp:101-262), with its messages
        if (s == "-iw") { dP.imageWidth = std::atoi(next().c_str()); std::cout << "image width: " << dP.imageWidth << "\n"; continue; }
        if (s == "-ih") { dP.imageHeight = std::atoi(next().c_str()); std::cout << "image height: " << dP.imageHeight << "\n"; continue; }
        if (s == "-fl") { dP.focalLength = (float)std::atof(next().c_str()); std::cout << "focal length: " << dP.focalLength << "\n"; continue; }
        if (s == "-xs") { dP.xShift = (float)std::atof(next().c_str()); std::cout << "x shift: " << dP.xShift << "\n"; continue; }
        if (s == "-ys") { dP.yShift = (float)std::atof(next().c_str()); std::cout << "y shift: " << dP.yShift << "\n"; continue; }
        if (s == "-rd") { dP.rawData = std::atoi(next().c_str()); std::cout << "raw data (rescale rgb): " << dP.rawData << "\n"; continue; }
        if (s == "-sfl") { dP.secondaryFocalLength = (float)std::atof(next().c_str()); std::cout << "secondary focal length: " << dP.secondaryFocalLength << "\n"; continue; }
        if (s == "-rxs") { dP.rawXShift = (float)std::atof(next().c_str()); std::cout << "raw x shift: " << dP.rawXShift << "\n"; continue; }
        if (s == "-rys") { dP.rawYShift = (float)std::atof(next().c_str()); std::cout << "raw y shift: " << dP.rawYShift << "\n"; continue; }
        if (s == "-rdraw") { pP.randomDraw = std::atoi(next().c_str()); std::cout << "random draw: " << pP.randomDraw << "\n"; continue; }
        if (s == "-oscript") { dP.objScript = next(); std::cout << "object script: " << dP.objScript << "\n"; continue; }
        if (s == "-sscript") { dP.scoreScript = next(); std::cout << "score script: " << dP.scoreScript << "\n"; continue; }
        if (s == "-omodel") { dP.objModel = next(); std::cout << "object model: " << dP.objModel << "\n"; continue; }
        if (s == "-smodel") { dP.scoreModel = next(); std::cout << "score model: " << dP.scoreModel << "\n"; continue; }
        if (s == "-rT2D") { pP.ransacInlierThreshold2D = (float)std::atof(next().c_str()); std::cout << "ransac inlier threshold: " << pP.ransacInlierThreshold2D << "\n"; continue; }
        if (s == "-rT3D") { pP.ransacInlierThreshold3D = (float)std::atof(next().c_str()); std::cout << "ransac inlier threshold: " << pP.ransacInlierThreshold3D << "\n"; continue; }
        if (s == "-rRI") { pP.ransacRefinementIterations = std::atoi(next().c_str()); std::cout << "ransac iterations (refinement): " << pP.ransacRefinementIterations << "\n"; continue; }
        if (s == "-rI") { pP.ransacIterations = std::atoi(next().c_str()); std::cout << "ransac iterations: " << pP.ransacIterations << "\n"; continue; }
        if (s == "-rB") { pP.ransacBatchSize = std::atoi(next().c_str()); std::cout << "ransac batch size: " << pP.ransacBatchSize << "\n"; continue; }
        if (s == "-rSS") { pP.ransacSubSample = (float)std::atof(next().c_str()); std::cout << "ransac refinement gradient sub sampling: " << pP.ransacSubSample << "\n"; continue; }
        // keys of these drivers only (EngineParameters)
        if (s == "-synth") { eP.synthFrames = std::atoi(next().c_str()); std::cout << "synthetic frames: " << eP.synthFrames << "\n"; continue; }
        if (s == "-mw") { eP.mapWidth = std::atoi(next().c_str()); std::cout << "coordinate map width: " << eP.mapWidth << "\n"; continue; }
        if (s == "-mh") { eP.mapHeight = std::atoi(next().c_str()); std::cout << "coordinate map height: " << eP.mapHeight << "\n"; continue; }
        if (s == "-seed") { eP.seed = std::strtoull(next().c_str(), nullptr, 10); std::cout << "sampling seed: " << eP.seed << "\n"; continue; }
        if (s == "-tau") { eP.tau = (float)std::atof(next().c_str()); std::cout << "soft inlier threshold: " << eP.tau << "\n"; continue; }
        if (s == "-beta") { eP.beta = (float)std::atof(next().c_str()); std::cout << "soft inlier softness: " << eP.beta << "\n"; continue; }
        if (s == "-alpha") { eP.alpha = std::atof(next().c_str()); std::cout << "score scale: " << eP.alpha << "\n"; continue; }
        if (s == "-rounds") { eP.rounds = std::atoi(next().c_str()); std::cout << "training rounds: " << eP.rounds << "\n"; continue; }
        if (s == "-dev") { eP.device = std::atoi(next().c_str()); std::cout << "device: " << eP.device << "\n"; continue; }
        if (s == "-quirk") { eP.indexQuirk = std::atoi(next().c_str()) != 0; std::cout << "path II index quirk: " << eP.indexQuirk << "\n"; continue; }
        if (s == "-batch") { eP.batchGiven = true; eP.batch = std::atoi(next().c_str()); std::cout << "images per launch chain: " << eP.batch << "\n"; continue; }
        if (s == "-refstream") { eP.refstream = std::atoi(next().c_str()); std::cout << "reference random streams (threads): " << eP.refstream << "\n"; continue; }
        if (s == "-refsub") { eP.refsub = std::atoi(next().c_str()); std::cout << "skip the sub-sampler's draws: " << eP.refsub << "\n"; continue; }
        if (s == "-passes") { eP.passes = std::atoi(next().c_str()); std::cout << "passes over the data set: " << eP.passes << "\n"; continue; }
        if (s == "-warmup") { eP.warmupMs = std::atoi(next().c_str()); std::cout << "warm-up: " << eP.warmupMs << " ms\n"; continue; }
        if (s == "-gradstats") { eP.gradStats = std::atoi(next().c_str()); std::cout << "gradient statistics every: " << eP.gradStats << "\n"; continue; }
        if (s == "-seam") { eP.seam = std::atoi(next().c_str()) != 0; std::cout << "external score model through the seam: " << eP.seam << "\n"; continue; }
        if (s == "-defer") { eP.defer = std::atoi(next().c_str()); std::cout << "deferred tails: " << eP.defer << "\n"; continue; }
        if (s == "-errimg") { eP.errorImages = std::atoi(next().c_str()) != 0; std::cout << "error images: " << eP.errorImages << "\n"; continue; }
        std::cout << "unkown argument: " << argv[i] << "\n";  // (sic) core/properties.cpp:264
        return false;
    }
    return true;
}

void GlobalProperties::parseCmdLine(int argc, const char* argv[]) {
    std::vector<std::string> argVec;
    for (int i = 1; i < argc; i++) argVec.push_back(argv[i]);
    readArguments(argVec);
}

void GlobalProperties::parseConfig() {
    const std::string configFile = dP.config + ".config";
    std::cout << "Parsing config file: " << configFile << std::endl;
    std::ifstream file(configFile);
    if (!file.is_open()) return;
    std::vector<std::string> argVec;
    std::string line;
    while (true) {
        if (file.eof()) break;
        std::getline(file, line);
        if (line.length() == 0) continue;  // empty line
        if (line.at(0) == '#') continue;   // comment
        const std::vector<std::string> tokens = split(line);
        if (tokens.empty()) continue;
        argVec.push_back("-" + tokens[0]);
        argVec.push_back(tokens.size() > 1 ? tokens[1] : std::string());
    }
    readArguments(argVec);
}

Camera GlobalProperties::getCamMat() const {
    Camera c;
    c.fx = c.fy = dP.focalLength;
    c.cx = (float)(dP.imageWidth / 2) + dP.xShift;   // integer halves, core/properties.cpp:310-311
    c.cy = (float)(dP.imageHeight / 2) + dP.yShift;
    return c;
}

}  // namespace dsac
