// test_ransac_softam.cpp -- the evaluation program of the soft-argmax pipeline on the HIP engine: the shape of
// core/test_ransac_softam.cpp (parameters :44-58, per-image loop :97-230, pose export :161-210, output files :84-95,212-263).
//
// What differs from the reference, and why: its two Torch CNNs and its RGB-D reader are not part of this repository's scope, so a
// frame arrives as the scene-coordinate prediction itself (frame_io.h) and the hypothesis score is alpha x soft-inlier count
// (-tau / -beta / -alpha).  Everything else -- parameter names and defaults, default.config, the ./test/<scene>/ layout, translation.txt,
// the two output files with their names and column order, the console summary -- is the reference's.
#include <fstream>
#include <iostream>

#include "frame_io.h"
#include "properties.h"

int main(int argc, const char* argv[]) {
    using namespace dsac;
    // read in parameters
    GlobalProperties* gp = GlobalProperties::getInstance();
    gp->parseConfig();
    gp->parseCmdLine(argc, argv);

    const int objHyps = gp->pP.ransacIterations;
    const int inlierThreshold2D = (int)gp->pP.ransacInlierThreshold2D;  // truncated, test_ransac_softam.cpp:51
    const int refInlierCount = gp->pP.ransacBatchSize;
    const int refSteps = gp->pP.ransacRefinementIterations;
    const std::string modelFileRGB = gp->dP.objModel;
    const Camera camMat = gp->getCamMat();

    try {
        // load test data
        std::vector<DriverFrame> testDataset;
        if (gp->eP.synthFrames > 0) {
            const int mh = gp->eP.mapHeight > 0 ? gp->eP.mapHeight : 40, mw = gp->eP.mapWidth > 0 ? gp->eP.mapWidth : 40;
            for (int i = 0; i < gp->eP.synthFrames; i++) testDataset.push_back(synthFrame(mh, mw, camMat, gp->eP.seed + 7919ull * i));
        } else {
            std::cout << std::endl << "Loading test set ..." << std::endl;
            testDataset = loadFrames("./test/");
        }

        // The two result files keep the reference's names and column order (core/test_ransac_softam.cpp:161-263) so that runs can be diffed
        // against anyone's reference run; the columns are data here: {file, position, quantity}.
        //   per image:  loss | score-distribution entropy | translation error mm | rotation error deg | exported pose: 3 x Rodrigues, 3 x metres
        //   per run:    fraction correct | loss mean, stddev | entropy mean, stddev | median rotation error deg | median translation error mm
        const std::string suffix = modelFileRGB + "_rdraw" + intToString(gp->pP.randomDraw) + "_softam.txt";
        std::ofstream perRun("ransac_test_loss_" + suffix), perImage("ransac_test_errors_" + suffix);
        perRun.precision(10);
        perImage.precision(10);
        auto writeRow = [](std::ofstream& f, const std::vector<double>& cols, bool trailingBlank) {
            for (size_t k = 0; k < cols.size(); k++) f << cols[k] << ((k + 1 < cols.size() || trailingBlank) ? " " : "");
            f << std::endl;
        };

        double avgCorrect = 0;
        std::vector<double> losses, sfEntropies, rotErrs, tErrs;

        for (unsigned i = 0; i < testDataset.size(); i++) {
            std::cout << "Processing test image " << i << " of " << testDataset.size() << "." << std::endl;
            const DriverFrame& fr = testDataset[i];
            Frame frame(fr.estObj.data(), fr.sampling.empty() ? nullptr : fr.sampling.data(), fr.H, fr.W, camMat, gp->eP.device);
            const std::vector<int32_t> pixelIdxs = (!fr.pixelIdxs.empty() && fr.permSteps >= refSteps) ? fr.pixelIdxs : refinePermutations(fr.H * fr.W, refSteps);
            // process frame (same function used in training)
            const ProcessImageResult r = frame.processImage(fr.poseGT, objHyps, gp->eP.seed + i, inlierThreshold2D, refInlierCount, refSteps, pixelIdxs,
                                                            gp->eP.tau, gp->eP.beta, gp->eP.alpha, fr.sets.empty() ? nullptr : &fr.sets);
            avgCorrect += r.correct;

            // convert back to 7-Scenes norm, Rodriguez vector + translation in m, optional translation.txt
            const std::vector<double> hypV = exportPose7Scenes(r.refAvgHyp);

            writeRow(perImage, {r.loss, r.sfEntropy, r.tErr, r.rotErr, hypV[0], hypV[1], hypV[2], hypV[3], hypV[4], hypV[5]}, true);

            losses.push_back(r.loss);
            sfEntropies.push_back(r.sfEntropy);
            tErrs.push_back(r.tErr);
            rotErrs.push_back(r.rotErr);
        }

        double lossMean, lossStdDev, entropyMean, entropyStdDev;
        meanStdDev(losses, lossMean, lossStdDev);
        meanStdDev(sfEntropies, entropyMean, entropyStdDev);
        avgCorrect /= (double)testDataset.size();
        const double medianRotErr = medianOf(rotErrs), medianTErr = medianOf(tErrs);

        std::cout << "-----------------------------------------------------------" << std::endl;
        std::cout << "Avg. test loss: " << lossMean << ", accuracy: " << avgCorrect * 100 << "%" << std::endl;
        std::cout << "Median Rot. Error: " << medianRotErr << "deg, Median T. Error: " << medianTErr / 10 << "cm." << std::endl;

        writeRow(perRun, {avgCorrect, lossMean, lossStdDev, entropyMean, entropyStdDev, medianRotErr, medianTErr}, false);
    } catch (const Error& e) {
        std::cout << "dsac error " << e.code << ": " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
