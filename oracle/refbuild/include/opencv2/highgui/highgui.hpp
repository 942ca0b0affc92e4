// TEST INFRASTRUCTURE ONLY.  Hypothesis.h:33 includes this header; nothing from highgui is used on the hot path.
#pragma once
#include <opencv2/opencv.hpp>
