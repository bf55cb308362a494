// On-device evaluation metrics: per-plane MSE (-> PSNR, reference test.py:249-251, train.py:175-176) and the
// pixel-averaged SSIM of reference metrics/ssim.py:81-110 (11x11 gaussian window, "valid" convolution, per channel) as used
// by test.py:36-57.  One workgroup per (frame, channel) plane: both planes staged in LDS, the five window moments filtered
// separably (the reference's 2-D window is the normalised outer product of the 1-D gaussian with itself), SSIM formed per
// window position and reduced.  HBM-bound: every input value is read once.
#include "common.h"
#include "../../include/srvp_hip.h"
#include <math.h>

namespace {
constexpr int MAXD = 64, MAXF = 15;

struct MetricsK {
    const float* x; const float* y;
    float* mse; float* ssim;
    int H, W, F;
    float c1, c2;
    float g[MAXF];
};

__device__ inline float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}

__global__ __launch_bounds__(256) void frame_metrics_kernel(const MetricsK a) {
    __shared__ float px[MAXD * MAXD], py[MAXD * MAXD];
    __shared__ float hm[5][MAXD * (MAXD - 2)];          // horizontally filtered x, y, xx, yy, xy: [row][out col]
    __shared__ float red[4];
    const int H = a.H, W = a.W, F = a.F, OW = W - F + 1, OH = H - F + 1;
    const size_t base = (size_t)blockIdx.x * H * W;
    float se = 0.f;
    for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
        const float xv = a.x[base + i], yv = a.y[base + i];
        px[i] = xv; py[i] = yv;
        se += (xv - yv) * (xv - yv);
    }
    se = block_sum(se, red);                             // (also the barrier after the staging loop)
    if (threadIdx.x == 0 && a.mse) a.mse[blockIdx.x] = se / (float)(H * W);
    if (!a.ssim) return;
    if (OW <= 0 || OH <= 0) { if (threadIdx.x == 0) a.ssim[blockIdx.x] = nanf(""); return; }
    for (int i = threadIdx.x; i < H * OW; i += blockDim.x) {
        const int r = i / OW, c = i - r * OW;
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
        for (int k = 0; k < F; ++k) {
            const float g = a.g[k], xv = px[r * W + c + k], yv = py[r * W + c + k];
            sx += g * xv; sy += g * yv; sxx += g * xv * xv; syy += g * yv * yv; sxy += g * xv * yv;
        }
        hm[0][i] = sx; hm[1][i] = sy; hm[2][i] = sxx; hm[3][i] = syy; hm[4][i] = sxy;
    }
    __syncthreads();
    float acc = 0.f;
    for (int i = threadIdx.x; i < OH * OW; i += blockDim.x) {
        const int r = i / OW, c = i - r * OW;
        float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < F; ++k) {
            const float g = a.g[k];
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] += g * hm[q][(r + k) * OW + c];
        }
        const float mu1 = m[0], mu2 = m[1];
        const float s1 = m[2] - mu1 * mu1, s2 = m[3] - mu2 * mu2, s12 = m[4] - mu1 * mu2;
        const float v1 = 2.f * s12 + a.c2, v2 = s1 + s2 + a.c2;
        acc += ((2.f * mu1 * mu2 + a.c1) * v1) / ((mu1 * mu1 + mu2 * mu2 + a.c1) * v2);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) a.ssim[blockIdx.x] = acc / (float)(OH * OW);
}
}  // namespace

extern "C" int srvp_frame_metrics(const float* x, const float* y, int64_t planes, int H, int W, float max_val, int filter_size,
                                  float sigma, float k1, float k2, float* mse, float* ssim, void* stream) {
    SRVP_REQUIRE(x && y && (mse || ssim), "srvp_frame_metrics: null pointer");
    SRVP_REQUIRE(H > 0 && W > 0 && H <= MAXD && W <= MAXD, "srvp_frame_metrics: planes up to 64x64 (got %dx%d)", H, W);
    SRVP_REQUIRE(filter_size >= 3 && filter_size <= MAXF && (filter_size & 1) && sigma > 0.f,
                 "srvp_frame_metrics: odd filter_size in [3, %d] and sigma > 0 required", MAXF);
    SRVP_REQUIRE(planes >= 0 && planes < (1ll << 31), "srvp_frame_metrics: bad plane count");
    if (planes == 0) return SRVP_OK;
    MetricsK k;
    k.x = x; k.y = y; k.mse = mse; k.ssim = ssim; k.H = H; k.W = W; k.F = filter_size;
    k.c1 = (k1 * max_val) * (k1 * max_val); k.c2 = (k2 * max_val) * (k2 * max_val);
    double g[MAXF], s = 0.;
    for (int i = 0; i < filter_size; ++i) {
        const double c = i - (filter_size - 1) / 2.;
        g[i] = exp(-c * c / (2. * (double)sigma * sigma)); s += g[i];
    }
    for (int i = 0; i < MAXF; ++i) k.g[i] = i < filter_size ? (float)(g[i] / s) : 0.f;
    hipLaunchKernelGGL(frame_metrics_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, k);
    SRVP_CHECK_LAUNCH("srvp_frame_metrics");
    return SRVP_OK;
}
