// Small-channel boundary layers of the encoder / decoder (image side: Cin or Cout in {1,3}).
// These layers have arithmetic intensity ~15-26 FLOP/B: they are HBM-bound, so they are direct
// fp32 VALU kernels that read / write the reference's own frame layout -- x and x_ are fp32 (T*B, C, 64, 64),
// i.e. the (T, B, C, H, W) tensor of module/srvp.py:421 flattened -- and the bf16 NHWC tensors of the MFMA layers.
//
// Replaces: first encoder conv (module/conv.py:174 / :200) and its weight gradient; the sigmoid backward of the last
// decoder layer (conv.py:273-274) -- the last ConvTranspose2d itself runs on the MFMA kernel with Cout padded to 32.
#include "common.h"
#include "../../include/srvp_hip.h"

namespace {

#define MAXC 4      // max image channels
#define MAXK 4      // max kernel size

// ---------------------------------------------------------------------------------------------------------
// first layer forward: raw[n][oy][ox][co] = sum_{ci,kh,kw} x[n][ci][oy*s-p+kh][ox*s-p+kw] * w[co][ci][kh][kw]
// thread = (pixel, group of 8 output channels); per-channel sum / sumsq for BatchNorm in the epilogue.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_in_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          bf16_t* __restrict__ raw, double* stats, int N, int Cin, int H,
                                                          int W, int Cout, int Cout_real, int k, int s, int p, int OH,
                                                          int OW) {
    extern __shared__ float wsh[];                 // [Cin*k*k][Cout] (transposed for conflict-free reads)
    const int KK = Cin * k * k;
    for (int i = threadIdx.x; i < KK * Cout; i += blockDim.x) {
        int co = i % Cout, q = i / Cout;
        wsh[i] = co < Cout_real ? w[(size_t)co * KK + q] : 0.f;
    }
    __syncthreads();
    const int CG = Cout / 8, PPB = blockDim.x / CG;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    const long long P = (long long)N * OH * OW;
    if (pl < PPB)
        for (long long pix = (long long)blockIdx.x * PPB + pl; pix < P; pix += (long long)gridDim.x * PPB) {
            int ox = (int)(pix % OW); long long q = pix / OW;
            int oy = (int)(q % OH); int n = (int)(q / OH);
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int ci = 0; ci < Cin; ++ci)
                for (int kh = 0; kh < k; ++kh) {
                    int iy = oy * s - p + kh;
                    if (iy < 0 || iy >= H) continue;
                    for (int kw = 0; kw < k; ++kw) {
                        int ix = ox * s - p + kw;
                        if (ix < 0 || ix >= W) continue;
                        float xv = x[(((size_t)n * Cin + ci) * H + iy) * W + ix];
                        const float* wr = wsh + ((ci * k + kh) * k + kw) * Cout + cg * 8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] += xv * wr[e];
                    }
                }
            *reinterpret_cast<u32x4_t*>(raw + (size_t)pix * Cout + cg * 8) = pack8(acc);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += acc[e]; s2[e] += acc[e] * acc[e]; }
        }
    if (!stats) return;
    __shared__ float sred[256][17];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sred[threadIdx.x][e] = s1[e]; sred[threadIdx.x][8 + e] = s2[e]; }
    __syncthreads();
    for (int t = threadIdx.x; t < CG * 16; t += blockDim.x) {
        int c = t / 16, kk = t % 16;
        double sum = 0.;
        for (int l = 0; l < PPB; ++l) sum += sred[l * CG + c][kk];
        atomicAdd(stats + (kk >> 3) * Cout + c * 8 + (kk & 7), sum);
    }
}

// first layer weight gradient: dw[co][ci][kh][kw] += sum_pix draw[pix][co] * x[n][ci][oy*s-p+kh][ox*s-p+kw]
// Register-tiled outer product: a thread owns (8 output channels) x (one input channel's k*k taps) = 8*KK fp32
// accumulators and walks a contiguous run of pixels; a "stream" of CG*Cin such threads covers the whole dw for its
// pixels.  Per pixel and thread: one 16-byte gradient load + k*k frame loads (L1 hits) feed 8*k*k FMAs.  Streams of a
// workgroup are summed through LDS, then one atomic per weight and workgroup.
template <int K>
__global__ __launch_bounds__(256) void conv_in_wgrad_kernel(const float* __restrict__ x, const bf16_t* __restrict__ draw,
                                                           float* dw, int N, int Cin, int H, int W, int Cout, int Cout_real,
                                                           int s, int p, int OH, int OW, long long pix_per_stream) {
    constexpr int KK = K * K;
    extern __shared__ float part[];                 // [streams-1][G][8*KK]
    const int CG = Cout / 8;
    const int G = CG * Cin;                         // threads per stream
    const int SPB = blockDim.x / G;                 // streams per workgroup
    const int sid = threadIdx.x / G, tg = threadIdx.x % G;
    const int cg = tg % CG, ci = tg / CG;
    float acc[8][KK];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int t = 0; t < KK; ++t) acc[e][t] = 0.f;
    const long long P = (long long)N * OH * OW;
    if (sid < SPB) {
        const long long beg = ((long long)blockIdx.x * SPB + sid) * pix_per_stream;
        long long end = beg + pix_per_stream; if (end > P) end = P;
        if (beg < end) {
            int ox = (int)(beg % OW); long long q = beg / OW;
            int oy = (int)(q % OH); int n = (int)(q / OH);
            for (long long pix = beg; pix < end; ++pix) {
                float g[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(draw + (((size_t)n * (OH + 2) + oy + 1) * (OW + 2) + ox + 1) * Cout + cg * 8), g);
                const float* xp = x + ((size_t)n * Cin + ci) * H * W;
                float xv[KK];
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    int iy = oy * s - p + kh;
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
                        int ix = ox * s - p + kw;
                        bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                        xv[kh * K + kw] = ok ? xp[iy * W + ix] : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int t = 0; t < KK; ++t) acc[e][t] += g[e] * xv[t];
                if (++ox == OW) { ox = 0; if (++oy == OH) { oy = 0; ++n; } }
            }
        }
    }
    // reduce the streams of this workgroup
    if (sid > 0 && sid < SPB) {
        float* dst = part + ((size_t)(sid - 1) * G + tg) * (8 * KK);
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int t = 0; t < KK; ++t) dst[e * KK + t] = acc[e][t];
    }
    __syncthreads();
    if (sid == 0) {
        for (int o = 1; o < SPB; ++o) {
            const float* src = part + ((size_t)(o - 1) * G + tg) * (8 * KK);
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int t = 0; t < KK; ++t) acc[e][t] += src[e * KK + t];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int co = cg * 8 + e;
            if (co < Cout_real)
#pragma unroll
                for (int t = 0; t < KK; ++t) atomicAdd(dw + ((size_t)co * Cin + ci) * KK + t, acc[e][t]);
        }
    }
}

}  // namespace

extern "C" int srvp_conv_in_fwd(const float* x, const float* w, void* raw, double* stats, int N, int Cin, int H, int W,
                                int Cout, int Cout_real, int k, int s, int p, void* stream) {
    SRVP_REQUIRE(x && w && raw, "srvp_conv_in_fwd: null pointer");
    SRVP_REQUIRE(Cin >= 1 && Cin <= MAXC && k <= MAXK && Cout % 8 == 0 && Cout / 8 <= 256, "srvp_conv_in_fwd: unsupported shape");
    int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
    size_t sh = (size_t)Cin * k * k * Cout * sizeof(float);
    long long P = (long long)N * OH * OW;
    int PPB = 256 / (Cout / 8);
    long long blocks = (P + (long long)PPB * 16 - 1) / ((long long)PPB * 16);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(conv_in_fwd_kernel, dim3((unsigned)blocks), dim3(256), sh, (hipStream_t)stream, x, w, (bf16_t*)raw, stats,
                       N, Cin, H, W, Cout, Cout_real, k, s, p, OH, OW);
    SRVP_CHECK_LAUNCH("srvp_conv_in_fwd");
    return SRVP_OK;
}

extern "C" int srvp_conv_in_wgrad(const float* x, const void* draw, float* dw, int N, int Cin, int H, int W, int Cout,
                                  int Cout_real, int k, int s, int p, void* stream) {
    SRVP_REQUIRE(x && draw && dw, "srvp_conv_in_wgrad: null pointer");
    SRVP_REQUIRE(Cin >= 1 && Cin <= MAXC && k <= MAXK && Cout % 8 == 0, "srvp_conv_in_wgrad: unsupported shape");
    int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
    SRVP_REQUIRE(k == 3 || k == 4, "srvp_conv_in_wgrad: k=%d unsupported", k);
    const int G = (Cout / 8) * Cin;
    SRVP_REQUIRE(G <= 256, "srvp_conv_in_wgrad: Cout*Cin too large");
    const int SPB = 256 / G;
    const long long P = (long long)N * OH * OW;
    long long blocks = 1024;
    long long pps = (P + blocks * SPB - 1) / (blocks * SPB);
    if (pps < 16) { pps = 16; blocks = (P + pps * SPB - 1) / (pps * SPB); }
    const size_t sh = (size_t)(SPB > 1 ? SPB - 1 : 1) * G * 8 * k * k * sizeof(float);
    SRVP_REQUIRE(sh <= 160 * 1024, "srvp_conv_in_wgrad: LDS budget");
    if (k == 3)
        hipLaunchKernelGGL(conv_in_wgrad_kernel<3>, dim3((unsigned)blocks), dim3(256), sh, (hipStream_t)stream, x,
                           (const bf16_t*)draw, dw, N, Cin, H, W, Cout, Cout_real, s, p, OH, OW, pps);
    else
        hipLaunchKernelGGL(conv_in_wgrad_kernel<4>, dim3((unsigned)blocks), dim3(256), sh, (hipStream_t)stream, x,
                           (const bf16_t*)draw, dw, N, Cin, H, W, Cout, Cout_real, s, p, OH, OW, pps);
    SRVP_CHECK_LAUNCH("srvp_conv_in_wgrad");
    return SRVP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Gradient hand-over of the image-side output layer: dpre = dx_ * x_ * (1 - x_) (sigmoid backward, conv.py:273-274)
// from the fp32 (N, nc, H, W) frame tensors into the bf16 NHWC tensor [N][H+2][W+2][C] (1-pixel zero border,
// channels >= nc zero) that the MFMA data-/weight-gradient kernels consume.
// ---------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void out_dpre_kernel(const float* __restrict__ xo, const float* __restrict__ dxo,
                                                       bf16_t* __restrict__ draw, int N, int nc, int H, int W, int C,
                                                       int sigmoid) {
    const int CG = C / 8;
    const long long total = (long long)N * H * W * CG;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
        int cg = (int)(it % CG); long long p = it / CG;
        int x = (int)(p % W); p /= W;
        int y = (int)(p % H); int n = (int)(p / H);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int c = cg * 8 + e;
            float v = 0.f;
            if (c < nc) {
                size_t o = (((size_t)n * nc + c) * H + y) * W + x;
                v = dxo[o];
                if (sigmoid) { float s = xo[o]; v *= s * (1.f - s); }
            }
            f[e] = v;
        }
        size_t off = (((size_t)n * (H + 2) + y + 1) * (W + 2) + x + 1) * C + cg * 8;
        *reinterpret_cast<u32x4_t*>(draw + off) = pack8(f);
    }
}
}  // namespace

extern "C" int srvp_out_dpre(const float* x_out, const float* dx_out, void* draw, int N, int nc, int H, int W, int C,
                             int apply_sigmoid, void* stream) {
    SRVP_REQUIRE(x_out && dx_out && draw && C % 8 == 0 && nc <= C, "srvp_out_dpre: bad args");
    long long total = (long long)N * H * W * (C / 8);
    long long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(out_dpre_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_out, dx_out, (bf16_t*)draw,
                       N, nc, H, W, C, apply_sigmoid);
    SRVP_CHECK_LAUNCH("srvp_out_dpre");
    return SRVP_OK;
}
