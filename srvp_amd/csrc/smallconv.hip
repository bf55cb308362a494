// Small-channel boundary layers of the encoder / decoder (image side: Cin or Cout in {1,3}).
// These layers have arithmetic intensity ~15-26 FLOP/B: they are HBM-bound, so they are direct
// fp32 VALU kernels that read / write the reference's own frame layout -- x and x_ are fp32 (T*B, C, 64, 64),
// i.e. the (T, B, C, H, W) tensor of module/srvp.py:421 flattened -- and the bf16 NHWC tensors of the MFMA layers.
//
// Replaces: first encoder conv (module/conv.py:174 / :200), last decoder ConvTranspose2d + sigmoid
// (conv.py:304 / :353, :273-274) and their autograd backward.
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

// first layer weight gradient: dw[co][ci][kh][kw] += sum_pix draw[pix][co] * x[..]
// thread = (group of 8 output channels, one (ci,kh,kw) tap); workgroup loops over a slice of pixels.
__global__ __launch_bounds__(1024) void conv_in_wgrad_kernel(const float* __restrict__ x, const bf16_t* __restrict__ draw,
                                                            float* dw, int N, int Cin, int H, int W, int Cout,
                                                            int Cout_real, int k, int s, int p, int OH, int OW) {
    const int KK = Cin * k * k;
    const int CG = Cout / 8;
    const int KKP = blockDim.x / CG;                // taps handled in parallel (>= KK required)
    const int cg = threadIdx.x % CG, tq = threadIdx.x / CG;
    const bool active = tq < KK;
    int ci = 0, kh = 0, kw = 0;
    if (active) { ci = tq / (k * k); kh = (tq / k) % k; kw = tq % k; }
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const long long P = (long long)N * OH * OW;
    const long long per = (P + gridDim.x - 1) / gridDim.x;
    const long long beg = (long long)blockIdx.x * per;
    long long end = beg + per; if (end > P) end = P;
    if (active)
        for (long long pix = beg; pix < end; ++pix) {
            int ox = (int)(pix % OW); long long q = pix / OW;
            int oy = (int)(q % OH); int n = (int)(q / OH);
            int iy = oy * s - p + kh, ix = ox * s - p + kw;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            float xv = x[(((size_t)n * Cin + ci) * H + iy) * W + ix];
            size_t off = (((size_t)n * (OH + 2) + oy + 1) * (OW + 2) + ox + 1) * Cout + cg * 8;
            float g[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(draw + off), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += g[e] * xv;
        }
    if (active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int co = cg * 8 + e;
            if (co < Cout_real) atomicAdd(dw + (size_t)co * KK + tq, acc[e]);
        }
    }
    (void)KKP;
}

// ---------------------------------------------------------------------------------------------------------
// last layer forward: x_[n][co][oy][ox] = sigmoid( sum_{ci,kh,kw} in[n][iy][ix][ci] * w[ci][co][kh][kw] ),
// oy = iy*s - p + kh (transposed convolution).  in = one or two bf16 NHWC tensors with a 1-pixel zero border.
// thread = output pixel.
// ---------------------------------------------------------------------------------------------------------
struct OutK {
    const bf16_t* src0; const bf16_t* src1; const int* map1;
    int C0, C1, C0_real, C1_real;
    int N, H, W, Cout, k, s, p, OH, OW, sigmoid;
};

__global__ __launch_bounds__(256) void convT_out_fwd_kernel(const OutK a, const float* __restrict__ w,
                                                            float* __restrict__ xo) {
    extern __shared__ float wsh[];                 // [C0+C1][k*k][Cout]  (zero rows for padded channels)
    const int Ct = a.C0 + a.C1, kk2 = a.k * a.k;
    for (int i = threadIdx.x; i < Ct * kk2 * a.Cout; i += blockDim.x) {
        int co = i % a.Cout; int q = i / a.Cout; int tap = q % kk2; int c = q / kk2;
        // map the padded channel index to the real input-channel index of the IOHW weight
        int cr = -1;
        if (c < a.C0) { if (c < a.C0_real) cr = c; }
        else { int c1 = c - a.C0; if (c1 < a.C1_real) cr = a.C0_real + c1; }
        wsh[i] = cr >= 0 ? w[((size_t)cr * a.Cout + co) * kk2 + tap] : 0.f;
    }
    __syncthreads();
    const long long P = (long long)a.N * a.OH * a.OW;
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < P; pix += (long long)gridDim.x * blockDim.x) {
        int ox = (int)(pix % a.OW); long long q = pix / a.OW;
        int oy = (int)(q % a.OH); int n = (int)(q / a.OH);
        float acc[MAXC] = {0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < a.k; ++kh) {
            int ty = oy + a.p - kh;
            if (ty < 0 || ty % a.s) continue;
            int iy = ty / a.s;
            if (iy >= a.H) continue;
            for (int kw = 0; kw < a.k; ++kw) {
                int tx = ox + a.p - kw;
                if (tx < 0 || tx % a.s) continue;
                int ix = tx / a.s;
                if (ix >= a.W) continue;
                const int tap = kh * a.k + kw;
                for (int srcI = 0; srcI < 2; ++srcI) {
                    const bf16_t* src = srcI ? a.src1 : a.src0;
                    const int C = srcI ? a.C1 : a.C0;
                    if (C == 0) continue;
                    int nn = (srcI && a.map1) ? a.map1[n] : n;
                    const bf16_t* px = src + (((size_t)nn * (a.H + 2) + iy + 1) * (a.W + 2) + ix + 1) * C;
                    const float* wr = wsh + ((size_t)(srcI ? a.C0 : 0) * kk2 + tap) * a.Cout;
                    for (int c8 = 0; c8 < C; c8 += 8) {
                        float f[8];
                        unpack8(*reinterpret_cast<const u32x4_t*>(px + c8), f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float* wp = wr + (size_t)(c8 + e) * kk2 * a.Cout;
                            for (int co = 0; co < a.Cout; ++co) acc[co] += f[e] * wp[co];
                        }
                    }
                }
            }
        }
        for (int co = 0; co < a.Cout; ++co) {
            float v = acc[co];
            if (a.sigmoid) v = 1.f / (1.f + __expf(-v));
            xo[(((size_t)n * a.Cout + co) * a.OH + oy) * a.OW + ox] = v;
        }
    }
}

// last layer backward, data gradient: dact[n][iy][ix][c] = sum_{co,kh,kw} dpre[n][co][oy][ox] * w[c][co][kh][kw]
// thread = (input pixel, group of 8 channels).  dpre = dx_ * x_ * (1 - x_) when the sigmoid is applied.
__global__ __launch_bounds__(256) void convT_out_dact_kernel(const OutK a, const float* __restrict__ w,
                                                             const float* __restrict__ xo, const float* __restrict__ dxo,
                                                             bf16_t* __restrict__ dact) {
    extern __shared__ float wsh[];                 // [k*k][Cout][C0+C1]
    const int Ct = a.C0 + a.C1, kk2 = a.k * a.k;
    for (int i = threadIdx.x; i < Ct * kk2 * a.Cout; i += blockDim.x) {
        int c = i % Ct; int q = i / Ct; int co = q % a.Cout; int tap = q / a.Cout;
        int cr = -1;
        if (c < a.C0) { if (c < a.C0_real) cr = c; }
        else { int c1 = c - a.C0; if (c1 < a.C1_real) cr = a.C0_real + c1; }
        wsh[i] = cr >= 0 ? w[((size_t)cr * a.Cout + co) * kk2 + tap] : 0.f;
    }
    __syncthreads();
    const int CG = Ct / 8;
    const long long total = (long long)a.N * a.H * a.W * CG;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
        int cg = (int)(it % CG); long long pq = it / CG;
        int ix = (int)(pq % a.W); pq /= a.W;
        int iy = (int)(pq % a.H); int n = (int)(pq / a.H);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int kh = 0; kh < a.k; ++kh) {
            int oy = iy * a.s - a.p + kh;
            if (oy < 0 || oy >= a.OH) continue;
            for (int kw = 0; kw < a.k; ++kw) {
                int ox = ix * a.s - a.p + kw;
                if (ox < 0 || ox >= a.OW) continue;
                for (int co = 0; co < a.Cout; ++co) {
                    size_t o = (((size_t)n * a.Cout + co) * a.OH + oy) * a.OW + ox;
                    float d = dxo[o];
                    if (a.sigmoid) { float xv = xo[o]; d *= xv * (1.f - xv); }
                    const float* wr = wsh + ((size_t)(kh * a.k + kw) * a.Cout + co) * Ct + cg * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += d * wr[e];
                }
            }
        }
        size_t off = (((size_t)n * a.H + iy) * a.W + ix) * Ct + cg * 8;
        *reinterpret_cast<u32x4_t*>(dact + off) = pack8(acc);
    }
}

// last layer backward, weight gradient: dw[c][co][kh][kw] += sum_pix in[n][iy][ix][c] * dpre[n][co][oy][ox]
// thread = (group of 8 input channels, one (co,kh,kw)); workgroup loops over a slice of input pixels.
__global__ __launch_bounds__(256) void convT_out_wgrad_kernel(const OutK a, const float* __restrict__ xo,
                                                              const float* __restrict__ dxo, float* dw, int cg_base) {
    const int Ct = a.C0 + a.C1, kk2 = a.k * a.k;
    const int TQ = a.Cout * kk2;                    // (co, tap) pairs
    const int tq = threadIdx.x % 64, cgl = threadIdx.x / 64;   // 64 lanes over (co,tap), 4 channel groups per block
    const int cg = cg_base + blockIdx.y * 4 + cgl;
    const bool active = tq < TQ && cg * 8 < Ct;
    int co = 0, kh = 0, kw = 0;
    if (active) { co = tq / kk2; kh = (tq % kk2) / a.k; kw = tq % a.k; }
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const long long P = (long long)a.N * a.H * a.W;
    const long long per = (P + gridDim.x - 1) / gridDim.x;
    const long long beg = (long long)blockIdx.x * per;
    long long end = beg + per; if (end > P) end = P;
    if (active) {
        const int c = cg * 8;
        const bool second = c >= a.C0;
        const bf16_t* src = second ? a.src1 : a.src0;
        const int C = second ? a.C1 : a.C0, cl = second ? c - a.C0 : c;
        for (long long pix = beg; pix < end; ++pix) {
            int ix = (int)(pix % a.W); long long q = pix / a.W;
            int iy = (int)(q % a.H); int n = (int)(q / a.H);
            int oy = iy * a.s - a.p + kh, ox = ix * a.s - a.p + kw;
            if (oy < 0 || oy >= a.OH || ox < 0 || ox >= a.OW) continue;
            size_t o = (((size_t)n * a.Cout + co) * a.OH + oy) * a.OW + ox;
            float d = dxo[o];
            if (a.sigmoid) { float xv = xo[o]; d *= xv * (1.f - xv); }
            int nn = (second && a.map1) ? a.map1[n] : n;
            float f[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(src + (((size_t)nn * (a.H + 2) + iy + 1) * (a.W + 2) + ix + 1) * C + cl), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f[e] * d;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int cc = c + e;
            int cr = -1;
            if (cc < a.C0) { if (cc < a.C0_real) cr = cc; }
            else { int c1 = cc - a.C0; if (c1 < a.C1_real) cr = a.C0_real + c1; }
            if (cr >= 0) atomicAdd(dw + ((size_t)cr * a.Cout + co) * kk2 + kh * a.k + kw, acc[e]);
        }
    }
}

int fill_out(const srvp_convout_desc* d, OutK& k) {
    SRVP_REQUIRE(d && d->src0, "srvp_convT_out: null pointer");
    SRVP_REQUIRE(d->Cout >= 1 && d->Cout <= MAXC && d->k <= MAXK, "srvp_convT_out: Cout=%d k=%d unsupported", d->Cout, d->k);
    SRVP_REQUIRE(d->C0 % 8 == 0 && d->C1 % 8 == 0, "srvp_convT_out: channels must be padded to 8");
    k.src0 = (const bf16_t*)d->src0; k.src1 = (const bf16_t*)d->src1; k.map1 = d->map1;
    k.C0 = d->C0; k.C1 = d->C1; k.C0_real = d->C0_real; k.C1_real = d->C1_real;
    k.N = d->N; k.H = d->H; k.W = d->W; k.Cout = d->Cout; k.k = d->k; k.s = d->s; k.p = d->p;
    k.OH = (d->H - 1) * d->s - 2 * d->p + d->k; k.OW = (d->W - 1) * d->s - 2 * d->p + d->k;
    k.sigmoid = d->apply_sigmoid;
    return SRVP_OK;
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
    int CG = Cout / 8, KK = Cin * k * k;
    int threads = CG * KK;
    threads = (threads + 63) / 64 * 64;
    SRVP_REQUIRE(threads <= 1024, "srvp_conv_in_wgrad: Cout*Cin*k*k too large");
    long long P = (long long)N * OH * OW;
    long long blocks = P / 512; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(conv_in_wgrad_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, x,
                       (const bf16_t*)draw, dw, N, Cin, H, W, Cout, Cout_real, k, s, p, OH, OW);
    SRVP_CHECK_LAUNCH("srvp_conv_in_wgrad");
    return SRVP_OK;
}

extern "C" int srvp_convT_out_fwd(const srvp_convout_desc* d, const float* w, float* x_out, void* stream) {
    OutK k;
    int rc = fill_out(d, k);
    if (rc) return rc;
    SRVP_REQUIRE(w && x_out, "srvp_convT_out_fwd: null pointer");
    size_t sh = (size_t)(k.C0 + k.C1) * k.k * k.k * k.Cout * sizeof(float);
    long long P = (long long)k.N * k.OH * k.OW;
    long long blocks = (P + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(convT_out_fwd_kernel, dim3((unsigned)blocks), dim3(256), sh, (hipStream_t)stream, k, w, x_out);
    SRVP_CHECK_LAUNCH("srvp_convT_out_fwd");
    return SRVP_OK;
}

extern "C" int srvp_convT_out_bwd(const srvp_convout_desc* d, const float* w, const float* x_out, const float* dx_out,
                                  void* dact, float* dw, void* stream) {
    OutK k;
    int rc = fill_out(d, k);
    if (rc) return rc;
    SRVP_REQUIRE(w && x_out && dx_out, "srvp_convT_out_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int Ct = k.C0 + k.C1;
    if (dact) {
        size_t sh = (size_t)Ct * k.k * k.k * k.Cout * sizeof(float);
        long long total = (long long)k.N * k.H * k.W * (Ct / 8);
        long long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(convT_out_dact_kernel, dim3((unsigned)blocks), dim3(256), sh, st, k, w, x_out, dx_out, (bf16_t*)dact);
        SRVP_CHECK_LAUNCH("srvp_convT_out_bwd(dact)");
    }
    if (dw) {
        SRVP_REQUIRE(k.Cout * k.k * k.k <= 64, "srvp_convT_out_bwd: Cout*k*k > 64");
        long long P = (long long)k.N * k.H * k.W;
        long long bx = P / 256; if (bx > 1024) bx = 1024; if (bx < 1) bx = 1;
        int by = (Ct / 8 + 3) / 4;
        hipLaunchKernelGGL(convT_out_wgrad_kernel, dim3((unsigned)bx, by), dim3(256), 0, st, k, x_out, dx_out, dw, 0);
        SRVP_CHECK_LAUNCH("srvp_convT_out_bwd(dw)");
    }
    return SRVP_OK;
}
