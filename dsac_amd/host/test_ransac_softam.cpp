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

        std::ofstream testFile;  // contains evaluation information for the whole test sequence
        testFile.open("ransac_test_loss_" + modelFileRGB + "_rdraw" + intToString(gp->pP.randomDraw) + "_softam.txt");
        std::ofstream testErrFile;  // contains evaluation information for each test image
        testErrFile.open("ransac_test_errors_" + modelFileRGB + "_rdraw" + intToString(gp->pP.randomDraw) + "_softam.txt");
        testFile.precision(10);
        testErrFile.precision(10);

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

            testErrFile << r.loss << " "       // 0 - loss of the average hypothesis
                        << r.sfEntropy << " "  // 1 - entropy of the hypothesis score distribution
                        << r.tErr << " "       // 2 - translational error in mm
                        << r.rotErr << " "     // 3 - rotational error in deg
                        << hypV[0] << " "      // 4 - selected pose, rotation (1st component of Rodriguez vector)
                        << hypV[1] << " "      // 5 - selected pose, rotation (2nd component of Rodriguez vector)
                        << hypV[2] << " "      // 6 - selected pose, rotation (3th component of Rodriguez vector)
                        << hypV[3] << " "      // 7 - selected pose, translation in m (x)
                        << hypV[4] << " "      // 8 - selected pose, translation in m (y)
                        << hypV[5] << " "      // 9 - selected pose, translation in m (z)
                        << std::endl;

            // store statistics for calculation of mean, median, stddev
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

        testFile << avgCorrect << " "     // 0 - percentage of correct poses
                 << lossMean << " "       // 1 - mean loss of average hypotheses
                 << lossStdDev << " "     // 2 - standard deviation of losses of average hypotheses
                 << entropyMean << " "    // 3 - mean of the score distribution entropy
                 << entropyStdDev << " "  // 4 - standard deviation of the score distribution entropy
                 << medianRotErr << " "   // 5 - median rotational error of selected hypotheses
                 << medianTErr            // 6 - median translational error (in mm) of selected hypotheses
                 << std::endl;
        testFile.close();
        testErrFile.close();
    } catch (const Error& e) {
        std::cout << "dsac error " << e.code << ": " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
