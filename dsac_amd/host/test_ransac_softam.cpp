// test_ransac_softam.cpp -- the evaluation program of the soft-argmax pipeline on the HIP engine: the shape of
// core/test_ransac_softam.cpp (parameters :44-58, per-image loop :97-230, pose export :161-210, output files :84-95,212-263).
//
// What differs from the reference, and why: its two Torch CNNs and its RGB-D reader are not part of this repository's scope, so a
// frame arrives as the scene-coordinate prediction itself (frame_io.h) and the hypothesis score is alpha x soft-inlier count
// (-tau / -beta / -alpha).  Everything else -- parameter names and defaults, default.config, the ./test/<scene>/ layout, translation.txt,
// the two output files with their names and column order, the console summary -- is the reference's.
#include <algorithm>
#include <chrono>
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
        auto record = [&](const ProcessImageResult& r) {
            avgCorrect += r.correct;
            // convert back to 7-Scenes norm, Rodriguez vector + translation in m, optional translation.txt
            const std::vector<double> hypV = exportPose7Scenes(r.refAvgHyp);
            writeRow(perImage, {r.loss, r.sfEntropy, r.tErr, r.rotErr, hypV[0], hypV[1], hypV[2], hypV[3], hypV[4], hypV[5]}, true);
            losses.push_back(r.loss);
            sfEntropies.push_back(r.sfEntropy);
            tErrs.push_back(r.tErr);
            rotErrs.push_back(r.rotErr);
        };
        using clk = std::chrono::high_resolution_clock;  // as core/stop_watch.h:50-69
        auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };

        // ONE engine context for the whole run (dsac_create: device query, stream, scratch -- once, not per image)
        Context& engine = Context::shared(gp->eP.device);
        const size_t nImg = testDataset.size();
        // Images of one geometry that draw their own minimal sets go through the device in batches (FrameBatch: the data set resident in HBM, one
        // launch chain per 16 images, the refinement of a batch under sampling / scoring of the next); recorded replays (golden frames) and odd
        // hypothesis counts take the per-image path.  Both give the same numbers image by image (tests/test_gpu_drivers.py).
        bool batchable = gp->eP.batch > 0 && nImg > 0 && (gp->eP.batch == 1 || objHyps % 128 == 0);
        for (const DriverFrame& fr : testDataset)
            batchable = batchable && gp->eP.refstream <= 0 && fr.H == testDataset[0].H && fr.W == testDataset[0].W && fr.sets.empty() && fr.sampling.empty() == testDataset[0].sampling.empty() &&
                        (fr.pixelIdxs.empty() || fr.permSteps < refSteps);
        const int passes = std::max(1, gp->eP.passes);
        if (batchable) {
            const int H = testDataset[0].H, W = testDataset[0].W;
            FrameBatchOptions opt;
            opt.errorImages = gp->eP.errorImages || gp->eP.seam;  // the seam is the error images
            opt.sampling = !testDataset[0].sampling.empty();  // sub-sampled maps: every image has its own table of image positions
            opt.deferTail = gp->eP.defer >= 1;
            opt.deferScoreTail = gp->eP.defer >= 2;
            const clk::time_point tUp = clk::now();
            FrameBatch batch(engine, (int)nImg, H, W, camMat, objHyps, refSteps, refinePermutations(H * W, refSteps), gp->eP.batch, opt);
            for (size_t i = 0; i < nImg; i++) batch.setFrame((int)i, testDataset[i].estObj.data(), testDataset[i].poseGT, opt.sampling ? testDataset[i].sampling.data() : nullptr);
            engine.synchronize();
            const double upMs = ms_since(tUp);
            // -seam 1: every batch's score comes from OUTSIDE the library (FrameBatch::processImages with a ScoreModel: error images out, scores in) -- here the
            // soft-inlier score dressed as an external model, so the run reproduces -seam 0; with -defer 2 the score tail of a batch reads its scores while
            // the next batch is already being scored
            const ScoreModel scoreModel = batch.softInlierModel(gp->eP.tau, gp->eP.beta, gp->eP.alpha);
            auto runAll = [&]() {
                if (gp->eP.seam) batch.processAll(gp->eP.seed, inlierThreshold2D, refInlierCount, scoreModel, gp->eP.tau, gp->eP.beta);
                else batch.processAll(gp->eP.seed, inlierThreshold2D, refInlierCount, gp->eP.tau, gp->eP.beta, gp->eP.alpha);
            };
            if (gp->eP.seam) std::cout << "score through the external seam" << std::endl;
            double firstMs = 0, restMs = 0;
            // the first pass (device warm-up) is timed on its own; the timed passes are enqueued back to back and waited for ONCE -- a loop over a
            // longer data set has no host synchronisation between its batches either, and the deferred tail of a pass's last batch then runs
            // under the next pass's first batch like any other
            {
                const clk::time_point t0 = clk::now();
                runAll();
                batch.synchronize();
                firstMs = ms_since(t0);
            }
            for (const clk::time_point tw = clk::now(); ms_since(tw) < gp->eP.warmupMs;) {  // -warmup: let the clock settle before the timed passes
                runAll();
                batch.synchronize();
            }
            if (passes > 1) {
                const clk::time_point t0 = clk::now();
                for (int pass = 1; pass < passes; pass++) runAll();
                batch.synchronize();
                restMs = ms_since(t0);
            }
            const std::vector<ProcessImageResult> res = batch.results(/*perHypothesis=*/false);
            for (size_t i = 0; i < nImg; i++) {
                std::cout << "Processing test image " << i << " of " << nImg << "." << std::endl;
                record(res[i]);
            }
            const double perPass = passes > 1 ? restMs / (passes - 1) : firstMs;
            std::cout << "Timing: " << nImg << " images x " << objHyps << " hypotheses, " << W << "x" << H << " coordinate maps resident in HBM, batches of "
                      << std::min<int>(gp->eP.batch, (int)nImg) << ": " << perPass * 1e3 / (double)nImg << " us per image (" << perPass << " ms per pass over "
                      << (passes > 1 ? passes - 1 : 1) << " timed pass(es); first pass " << firstMs << " ms; upload + set-up " << upMs << " ms)" << std::endl;
        } else {
            double firstMs = 0, restMs = 0;
            // the reference re-seeds a default mt19937 per image (cnn_softam.h:1104): images of one size share their permutations -- made once, kept in HBM
            int permH = 0, permW = 0;
            std::vector<int32_t> hostPerm;  // the replay path's permutations when the frame brings none
            DeviceArray<int32_t> permDev;
            for (int pass = 0; pass < passes; pass++) {
                const clk::time_point t0 = clk::now();
                for (unsigned i = 0; i < nImg; i++) {
                    if (pass + 1 == passes) std::cout << "Processing test image " << i << " of " << nImg << "." << std::endl;
                    const DriverFrame& fr = testDataset[i];
                    Frame frame(engine, fr.estObj.data(), fr.sampling.empty() ? nullptr : fr.sampling.data(), fr.H, fr.W, camMat);
                    const bool replayPerm = !fr.pixelIdxs.empty() && fr.permSteps >= refSteps;
                    ProcessImageResult r;
                    // -refstream T: the minimal sets from the reference's own generators (they run on from image to image, as ThreadRand's do); the rest of the
                    // image is the replay path on those sets
                    std::vector<std::array<int32_t, 4>> drawn;
                    if (gp->eP.refstream > 0 && fr.sets.empty()) {
                        if (i == 0) engine.forceInitRand((unsigned)gp->eP.seed, gp->eP.refstream);
                        std::vector<cv_trans_t> hypsDrawn;
                        const unsigned long long sub = (gp->eP.refsub && !fr.sampling.empty()) ? 4ull * fr.H * fr.W : 0ull;
                        frame.sampleHypothesesRefStream(objHyps, inlierThreshold2D, hypsDrawn, drawn, sub);
                    }
                    const std::vector<std::array<int32_t, 4>>* useSets = !fr.sets.empty() ? &fr.sets : (!drawn.empty() ? &drawn : nullptr);
                    // process frame (same function used in training)
                    if (replayPerm || useSets) {
                        if (!replayPerm && (int)hostPerm.size() != fr.H * fr.W * refSteps) hostPerm = refinePermutations(fr.H * fr.W, refSteps);  // once per map size: 15 ms of host shuffling at 640 x 480
                        const std::vector<int32_t>& pixelIdxs = replayPerm ? fr.pixelIdxs : hostPerm;
                        r = frame.processImage(fr.poseGT, objHyps, gp->eP.seed + i, inlierThreshold2D, refInlierCount, refSteps, pixelIdxs, gp->eP.tau, gp->eP.beta,
                                               gp->eP.alpha, useSets);
                    } else {
                        if (permH != fr.H || permW != fr.W) {
                            const std::vector<int32_t> pixelIdxs = refinePermutations(fr.H * fr.W, refSteps);
                            permDev.resize(engine, pixelIdxs.size());
                            permDev.upload(pixelIdxs.data(), pixelIdxs.size());
                            engine.synchronize();
                            permH = fr.H; permW = fr.W;
                        }
                        r = frame.processImage(fr.poseGT, objHyps, gp->eP.seed + i, inlierThreshold2D, refInlierCount, refSteps, permDev, gp->eP.tau, gp->eP.beta,
                                               gp->eP.alpha, /*wantInlierMap=*/false);
                    }
                    if (pass + 1 == passes) record(r);
                }
                (pass == 0 ? firstMs : restMs) += ms_since(t0);
            }
            const double perPass = passes > 1 ? restMs / (passes - 1) : firstMs;
            std::cout << "Timing: " << nImg << " images x " << objHyps << " hypotheses, one image per call (coordinate-map upload + processImage + copy-back): "
                      << perPass * 1e3 / (double)std::max<size_t>(1, nImg) << " us per image (" << perPass << " ms per pass; first pass " << firstMs << " ms)" << std::endl;
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
