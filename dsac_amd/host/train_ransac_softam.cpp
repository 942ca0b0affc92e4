// train_ransac_softam.cpp -- the end-to-end training program of the soft-argmax pipeline on the HIP engine, in the shape of
// core/train_ransac_softam.cpp: parameters (:44-66), one random training frame per round (:227-235), processImage forward (:258-286),
// the backward section (:288-394), gradient statistics (:396-409), training log "round loss sfEntropy" (:116, :416-420).
//
// The reference then hands dLoss_dObj to the scene-coordinate CNN (backward(), :412) and the score gradients to the score CNN;
// both CNNs are outside this repository's scope (dsac_amd/e2e.py has PyTorch stand-ins), so the round ends with
// the gradient the CNN would receive: its statistics go to the console as in the reference and, additionally, to
// ransac_training_grad_<objScript>.txt (round, max, avg, median of the row norms, number of zero rows).
#include <algorithm>
#include <chrono>
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

        // -batch F (F >= 1): the device-resident form of the loop below.  The training set stays in HBM, a round draws F random frames (the reference
        // draws one, :227-235; F frames per round is what a data-parallel step puts on one GPU), copies them device-to-device into the step's batch and
        // runs forward and backward of all F as ONE launch chain each (FrameBatch::processImages / backward); the gradients stay in HBM for a CNN that
        // runs there.  Frame k of a round samples from seed + 7919 round + k, so -batch 1 equals the per-image loop round by round.
        const int framesPerRound = gp->eP.batchGiven ? gp->eP.batch : 0;
        bool batchable = framesPerRound >= 1 && !trainingDataset.empty();
        for (const DriverFrame& fr : trainingDataset)
            batchable = batchable && fr.H == trainingDataset[0].H && fr.W == trainingDataset[0].W && fr.sets.empty() && fr.sampling.empty() == trainingDataset[0].sampling.empty() &&
                        (fr.pixelIdxs.empty() || fr.permSteps < refSteps);
        if (framesPerRound >= 1 && !batchable) std::cout << "-batch: the training set is not uniform (sizes, given sets or sampling tables); one image per round." << std::endl;
        if (batchable) {
            typedef std::chrono::steady_clock clk;
            const int H = trainingDataset[0].H, W = trainingDataset[0].W, F = framesPerRound;
            const size_t P = (size_t)H * W;
            Context& engine = Context::shared(gp->eP.device);
            const std::vector<int32_t> perms = refinePermutations(H * W, refSteps);
            FrameBatchOptions dataOpt;
            dataOpt.errorImages = false;
            dataOpt.sampling = !trainingDataset[0].sampling.empty();
            FrameBatch data(engine, (int)trainingDataset.size(), H, W, camMat, objHyps, refSteps, perms, 1, dataOpt);  // storage only: the resident training set
            for (size_t i = 0; i < trainingDataset.size(); i++) data.setFrame((int)i, trainingDataset[i].estObj.data(), trainingDataset[i].poseGT, dataOpt.sampling ? trainingDataset[i].sampling.data() : nullptr);
            FrameBatchOptions stepOpt;
            stepOpt.errorImages = gp->eP.errorImages || gp->eP.seam;  // the seam is the error images
            stepOpt.inlierMaps = true;
            stepOpt.sampling = dataOpt.sampling;
            stepOpt.deferTail = false;  // the backward pass follows the forward pass of the same frames: nothing to run the tail under
            FrameBatch step(engine, F, H, W, camMat, objHyps, refSteps, perms, F, stepOpt);
            engine.synchronize();
            // -seam 1: the score of a round comes from outside the library -- error images out, scores in (FrameBatch::processImages with a ScoreModel:
            // cnn_softam.h:1066-1078), score gradients out, gradient images in (FrameBatch::backward with it: train_ransac_softam.cpp:378-383).  A trainer with a
            // score CNN on the device plugs its forward / backward in here; this program's model is the soft-inlier score dressed as an external one.
            const ScoreModel scoreModel = step.softInlierModel(gp->eP.tau, gp->eP.beta, gp->eP.alpha);
            const bool seam = gp->eP.seam;
            auto forwardBackward = [&](unsigned long long seed) {
                if (seam) {
                    step.processImages(0, F, seed, inlierThreshold2D, refInlierCount, scoreModel, gp->eP.tau, gp->eP.beta);
                    step.backward(0, F, inlierThreshold2D, refInlierCount, refSubSample, scoreModel, gp->eP.indexQuirk);
                } else {
                    step.processImages(0, F, seed, inlierThreshold2D, refInlierCount, gp->eP.tau, gp->eP.beta, gp->eP.alpha);
                    step.backward(0, F, inlierThreshold2D, refInlierCount, refSubSample, gp->eP.tau, gp->eP.beta, gp->eP.alpha);
                }
            };
            const int statsEvery = gp->eP.gradStats;
            {   // -warmup: untimed, unlogged rounds on frames drawn from a generator of their own (the training sequence below is not disturbed)
                std::mt19937 warmRng(7);
                for (const clk::time_point tw = clk::now(); std::chrono::duration<double, std::milli>(clk::now() - tw).count() < gp->eP.warmupMs;) {
                    std::vector<int32_t> ids(F);
                    for (int k = 0; k < F; k++) ids[k] = (int32_t)(std::uniform_int_distribution<int>(0, (int)trainingDataset.size() - 1)(warmRng));
                    step.gatherFramesFrom(data, ids);
                    forwardBackward(gp->eP.seed);
                    engine.synchronize();
                }
            }
            clk::time_point t0 = clk::now();
            int timedRounds = 0;
            for (int round = 0; round <= trainingRounds; round++) {
                if (round == 1) { engine.synchronize(); t0 = clk::now(); }  // round 0 warms the device up
                std::vector<int32_t> ids(F);
                for (int k = 0; k < F; k++) ids[k] = (int32_t)(std::uniform_int_distribution<int>(0, (int)trainingDataset.size() - 1)(frameRng));
                step.gatherFramesFrom(data, ids);  // device-to-device, one launch per array
                forwardBackward(gp->eP.seed + 7919ull * round);
                if (round >= 1) timedRounds++;
                const bool stats = statsEvery > 0 && round % statsEvery == 0;
                if (!stats && round != trainingRounds) continue;  // nothing of this round is needed on the host: the next one is enqueued behind it
                const std::vector<ProcessImageResult> res = step.results(/*perHypothesis=*/false);
                double loss = 0, ent = 0;
                for (const ProcessImageResult& r : res) { loss += r.loss; ent += r.sfEntropy; }
                std::cout << "Round " << round << " of " << trainingRounds << "." << std::endl;
                if (stats) {
                    double maxN = 0, avgN = 0;
                    int zeroGrads = 0;
                    std::vector<double> norms;
                    norms.reserve(P * (size_t)F);
                    for (int k = 0; k < F; k++) {
                        const std::vector<double> d = step.gradients(k);
                        for (size_t p = 0; p < P; p++) {
                            const double n = std::sqrt(d[p * 3] * d[p * 3] + d[p * 3 + 1] * d[p * 3 + 1] + d[p * 3 + 2] * d[p * 3 + 2]);
                            norms.push_back(n);
                            if (n < 1e-8) zeroGrads++;
                            avgN += n;
                            maxN = std::max(maxN, n);
                        }
                    }
                    avgN /= (double)norms.size();
                    const double medN = medianOf(norms);
                    std::cout << "Combined statistics:" << std::endl;
                    std::cout << "Max gradient: " << maxN << std::endl;
                    std::cout << "Avg gradient: " << avgN << std::endl;
                    std::cout << "Med gradient: " << medN << std::endl;
                    std::cout << "Zero gradients: " << zeroGrads << std::endl;
                    gradFile << round << " " << maxN << " " << avgN << " " << medN << " " << zeroGrads << std::endl;
                }
                trainFile << round << " " << loss / F << " " << ent / F << std::endl;  // mean over the round's frames (F = 1: the reference's line)
                std::cout << std::endl;
            }
            engine.synchronize();
            const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            if (timedRounds > 0)
                std::cout << "Timing: " << timedRounds << " rounds x " << F << " frames x " << objHyps << " hypotheses, " << W << "x" << H
                          << " coordinate maps, forward + backward device-resident" << (seam ? " (score through the external seam)" : "") << ": " << ms * 1e3 / timedRounds << " us per round = "
                          << ms * 1e3 / timedRounds / F << " us per frame (gradient statistics every " << statsEvery << " rounds)" << std::endl;
            trainFile.close();
            gradFile.close();
            return 0;
        }

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
