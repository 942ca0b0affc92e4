// k_patches.hip -- the producer side of the path: RGB patch gather for the scene-coordinate CNN.
//
// Replaces the patch assembly of getCoordImg (core/cnn_softam.h:224-254) together with the table layout pushMaps builds
// for the CNN (core/lua_calls.h:63-80: patch n, channel c, row y, column x): for every sampled pixel the patchSize x
// patchSize window [orig - patchSize/2, orig + patchSize/2) of the BGR image, as float.  Byte work, store-bound: the image
// (0.9 MB) stays in cache, the output (1600 x 3 x 42 x 42 floats = 34 MB) is written once with 16-byte stores where the
// row length allows.  Patches whose window leaves the image are skipped by the reference (:235-239) and the remaining
// ones close ranks; the stratified sampler never produces such a position, so here they are written as zeros instead and
// `skipped` counts them (the host mirrors the reference's behaviour if it is ever non-zero).
#include "kernels.h"

namespace dk {

__global__ __launch_bounds__(256) void k_gather_patches(const uint8_t* __restrict__ bgr, int H, int W, const int32_t* __restrict__ sampling_xy, int n,
                                                        int patch, float* __restrict__ out, int32_t* __restrict__ skipped) {
    const int rows_total = n * 3 * patch;            // one workgroup row = one (patch, channel, y) line of `patch` floats
    const int row = blockIdx.x * (256 / 64) + (threadIdx.x >> 6);
    if (row >= rows_total) return;
    const int lane = threadIdx.x & 63;
    const int y = row % patch, c = (row / patch) % 3, i = row / (3 * patch);
    const int ox = sampling_xy[2 * i], oy = sampling_xy[2 * i + 1];
    const int half = patch / 2;
    const bool skip = (ox < half) || (oy < half) || (ox > W - half) || (oy > H - half);
    if (skip && lane == 0 && y == 0 && c == 0 && skipped) atomicAdd(skipped, 1);
    const int sy = oy - half + y;
    float* dst = out + (size_t)row * patch;
    for (int x = lane; x < patch; x += 64) {
        const int sx = ox - half + x;
        float v = 0.f;
        if (!skip && sy >= 0 && sy < H && sx >= 0 && sx < W) v = (float)bgr[((size_t)sy * W + sx) * 3 + c];
        __builtin_nontemporal_store(v, dst + x);
    }
}

hipError_t gather_patches(hipStream_t st, const uint8_t* bgr, int H, int W, const int32_t* sampling_xy, int n, int patch, float* out, int32_t* skipped) {
    if (n <= 0) return hipSuccess;
    const int rows_total = n * 3 * patch;
    hipLaunchKernelGGL(k_gather_patches, dim3((rows_total + 3) / 4), dim3(256), 0, st, bgr, H, W, sampling_xy, n, patch, out, skipped);
    return hipGetLastError();
}

}  // namespace dk
