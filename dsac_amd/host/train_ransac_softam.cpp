// train_ransac_softam.cpp -- the end-to-end training program of the soft-argmax pipeline on the HIP engine, in the shape of
// core/train_ransac_softam.cpp: parameters (:44-66), one random training frame per round (:227-235), processImage forward (:258-286),
// the backward section (:288-394), gradient statistics (:396-409), training log "round loss sfEntropy" (:116, :416-420).
//
// The reference then hands dLoss_dObj to the scene-coordinate CNN (backward(), :412) and the score gradients to the score CNN;
// both CNNs are outside this repository's scope (dsac_amd/e2e.py has PyTorch stand-ins), so the round ends with
// the gradient the CNN would receive: its statistics go to the console as in the reference and, additionally, to
// ransac_training_grad_<objScript>.txt (round, max, avg, median of the row norms, number of zero rows).
#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <random>

#include "frame_io.h"
#include "properties.h"

int main(int argc, const char* argv[]) {
    using namespace dsac;
    GlobalProperties* gp = GlobalProperties::getInstance();
    gp->parseConfig();
    gp->parseCmdLine(argc, argv);

    const int trainingRounds = gp->eP.rounds > 0 ? gp->eP.rounds : 5000;  // total number of parameter updates, train_ransac_softam.cpp:49
    const int refInlierCount = gp->pP.ransacBatchSize;
    const int refSteps = gp->pP.ransacRefinementIterations;
    const float refSubSample = gp->pP.ransacSubSample;
    const int objHyps = gp->pP.ransacIterations;
    const int inlierThreshold2D = (int)gp->pP.ransacInlierThreshold2D;
    const Camera camMat = gp->getCamMat();

    try {
        std::vector<DriverFrame> trainingDataset;
        if (gp->eP.synthFrames > 0) {
            const int mh = gp->eP.mapHeight > 0 ? gp->eP.mapHeight : 40, mw = gp->eP.mapWidth > 0 ? gp->eP.mapWidth : 40;
            for (int i = 0; i < gp->eP.synthFrames; i++) trainingDataset.push_back(synthFrame(mh, mw, camMat, gp->eP.seed + 7919ull * i));
        } else {
            std::cout << std::endl << "Loading training set ..." << std::endl;
            trainingDataset = loadFrames("./training/");
        }

        std::ofstream trainFile, gradFile;
        trainFile.open("ransac_training_loss_" + gp->dP.objScript + ".txt");  // contains statistics of the training process
        gradFile.open("ransac_training_grad_" + gp->dP.objScript + ".txt");
        trainFile.precision(10);
        gradFile.precision(10);
        std::mt19937 frameRng(1305);  // irand(0, size) of the reference's ThreadRand, thread_rand.h:100 (seed 1305 + thread id 0)

        for (int round = 0; round <= trainingRounds; round++) {
            std::cout << "Round " << round << " of " << trainingRounds << "." << std::endl;
            // load random training frame
            const int imgID = (int)(std::uniform_int_distribution<int>(0, (int)trainingDataset.size() - 1)(frameRng));
            const DriverFrame& fr = trainingDataset[imgID];
            Frame frame(fr.estObj.data(), fr.sampling.empty() ? nullptr : fr.sampling.data(), fr.H, fr.W, camMat, gp->eP.device);
            const std::vector<int32_t> pixelIdxs = (!fr.pixelIdxs.empty() && fr.permSteps >= refSteps) ? fr.pixelIdxs : refinePermutations(fr.H * fr.W, refSteps);
            // forward pass (also calculates many things needed for backward pass)
            const ProcessImageResult r = frame.processImage(fr.poseGT, objHyps, gp->eP.seed + 7919ull * round, inlierThreshold2D, refInlierCount, refSteps,
                                                            pixelIdxs, gp->eP.tau, gp->eP.beta, gp->eP.alpha, fr.sets.empty() ? nullptr : &fr.sets);
            // === doing the backward pass ===
            const std::vector<double> dLoss_dObj = frame.backward(r, fr.poseGT, inlierThreshold2D, refInlierCount, refSteps, refSubSample, pixelIdxs,
                                                                  gp->eP.tau, gp->eP.beta, gp->eP.alpha, gp->eP.indexQuirk);
            // gradient statistics (train_ransac_softam.cpp:396-409: getMax / getAvg / getMed over the row norms, zero rows below EPS)
            const size_t P = (size_t)fr.H * fr.W;
            std::vector<double> norms(P);
            int zeroGrads = 0;
            double avgN = 0, maxN = 0;
            for (size_t p = 0; p < P; p++) {
                const double n = std::sqrt(dLoss_dObj[p * 3] * dLoss_dObj[p * 3] + dLoss_dObj[p * 3 + 1] * dLoss_dObj[p * 3 + 1] + dLoss_dObj[p * 3 + 2] * dLoss_dObj[p * 3 + 2]);
                norms[p] = n;
                if (n < 1e-8) zeroGrads++;
                avgN += n;
                maxN = std::max(maxN, n);
            }
            avgN /= (double)P;
            const double medN = medianOf(norms);
            std::cout << "Combined statistics:" << std::endl;
            std::cout << "Max gradient: " << maxN << std::endl;
            std::cout << "Avg gradient: " << avgN << std::endl;
            std::cout << "Med gradient: " << medN << std::endl;
            std::cout << "Zero gradients: " << zeroGrads << std::endl;

            trainFile << round << " "      // 0 - training round (or number of parameter updates)
                      << r.loss << " "     // 1 - loss of the average hypothesis in this training round
                      << r.sfEntropy       // 2 - entropy of the score distribution (averaging weights)
                      << std::endl;
            gradFile << round << " " << maxN << " " << avgN << " " << medN << " " << zeroGrads << std::endl;
            std::cout << std::endl;
        }
        trainFile.close();
        gradFile.close();
    } catch (const Error& e) {
        std::cout << "dsac error " << e.code << ": " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
