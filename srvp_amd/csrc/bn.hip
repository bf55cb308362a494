// BatchNorm2d (training / eval) + activation around the MFMA convolutions, forward and backward.
// HBM-bound elementwise / reduction kernels: 16-byte (8 x bf16) accesses, one channel group per thread so that the
// per-channel coefficients live in registers, fp64 atomics for the grid-wide per-channel reductions.
//
// Replaces nn.BatchNorm2d + LeakyReLU/Tanh (+ MaxPool2d) as placed by reference module/conv.py:81-107,204-222 and
// their autograd backward; running-statistics semantics as torch (momentum 0.1, unbiased running_var, eps 1e-5).
#include "common.h"
#include "../../include/srvp_hip.h"

namespace {

__global__ void bn_finalize_kernel(const double* stats, double count, const float* gamma, const float* beta,
                                   float* rmean, float* rvar, long long* nbt, float* scale, float* shift, float* mean,
                                   float* invstd, int C, int C_real, float eps, float momentum) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    if (c >= C_real) { scale[c] = 0.f; shift[c] = 0.f; mean[c] = 0.f; invstd[c] = 0.f; return; }
    double m = stats[c] / count;
    double var = stats[C + c] / count - m * m;
    if (var < 0.) var = 0.;
    double is = 1.0 / sqrt(var + (double)eps);
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean[c] = (float)m; invstd[c] = (float)is;
    scale[c] = (float)(g * is);
    shift[c] = (float)(b - m * g * is);
    if (rmean) {
        double unb = count > 1. ? var * count / (count - 1.) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                      float* scale, float* shift, int C, int C_real, float eps) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (c >= C_real) { scale[c] = 0.f; shift[c] = 0.f; return; }
    if (!gamma) { scale[c] = 1.f; shift[c] = 0.f; return; }     // block without BatchNorm
    float is = 1.f / sqrtf(rvar[c] + eps);
    scale[c] = gamma[c] * is;
    shift[c] = beta[c] - rmean[c] * gamma[c] * is;
}

// ---------------------------------------------------------------------------------------------------------
// forward: act = f(scale*raw + shift)
// ---------------------------------------------------------------------------------------------------------
// Pixel walker: pixel index p -> (n, y, x) without divisions inside the loop: decode once, then advance by the
// (constant) grid stride with carries.
struct PixWalk {
    int n, y, x, dn, dy, dx, H, W;
    __device__ __forceinline__ void init(unsigned p0, unsigned stride, int H_, int W_) {
        H = H_; W = W_;
        x = (int)(p0 % (unsigned)W); unsigned q = p0 / (unsigned)W; y = (int)(q % (unsigned)H); n = (int)(q / (unsigned)H);
        dx = (int)(stride % (unsigned)W); q = stride / (unsigned)W; dy = (int)(q % (unsigned)H); dn = (int)(q / (unsigned)H);
    }
    __device__ __forceinline__ void next() {
        x += dx; if (x >= W) { x -= W; ++y; }
        y += dy; if (y >= H) { y -= H; ++n; }
        n += dn;
    }
};

// Finalize folded into the consumer (srvp_bn_finalize_act / srvp_bn_bwd_finalize_apply): every workgroup derives the per-channel
// coefficients ONCE into LDS (thread t: channels t, t + 256, ... -- the same fp64 expressions as the stand-alone finalize kernels,
// so the values are bit-identical), workgroup 0 also stores them / updates the running statistics / adds the parameter
// gradients.  42 dependent ~6 us launches per step less.  (Every THREAD deriving the coefficients of its own eight channels --
// 8 fp64 divisions + square roots x 256 threads x 4096 workgroups -- was measured slower than the separate launches, round 1.)
struct BnFin {
    const double* stats; double count; const float* gamma; const float* beta; float* rmean; float* rvar; long long* nbt;
    float* scale; float* shift; float* mean; float* invstd; int C_real; float eps, momentum;
};
struct BnBwdFin {
    const double* red; double count; float* dgamma; float* dbeta; float* coef; int C_real; float pscale;
};
constexpr int BN_MAX_C = 2048;       // C / 8 <= 256 channel groups

// ACT >= 0: the activation is a compile-time constant (LeakyReLU, all but two layers of each network); ACT < 0: run-time
// `act` -- a per-element switch over five activations with exp / division bodies that the compiler cannot hoist out of the
// pixel loops, which made these HBM-streaming kernels instruction-bound.
template <class E, bool POOL, int ACT>
__global__ __launch_bounds__(256) void bn_act_kernel(const E* __restrict__ raw, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int act, int N, int H, int W, int C,
                                                     E* __restrict__ dst, int db, E* __restrict__ dpool, int pb,
                                                     float* __restrict__ dst_f32, const int* __restrict__ keep, const BnFin fin,
                                                     const int dst_s2d, E* __restrict__ rawpool) {
    if (ACT >= 0) act = ACT;
    const int CG = C / 8;
    const int PPB = blockDim.x / CG;             // (pooled) pixels handled in parallel by one workgroup
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    float sc[8], sh[8];
    if (fin.stats) {
        __shared__ float s_coef[2 * BN_MAX_C];
        const bool lead = blockIdx.x == 0;
        if (lead && threadIdx.x == 0 && fin.nbt) *fin.nbt += 1;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float scv = 0.f, shv = 0.f, mv = 0.f, isv = 0.f;
            if (c < fin.C_real) {
                const double m = fin.stats[c] / fin.count;
                double var = fin.stats[C + c] / fin.count - m * m;
                if (var < 0.) var = 0.;
                const double is = 1.0 / sqrt(var + (double)fin.eps);
                const float g = fin.gamma ? fin.gamma[c] : 1.f, b = fin.beta ? fin.beta[c] : 0.f;
                mv = (float)m; isv = (float)is; scv = (float)(g * is); shv = (float)(b - m * g * is);
                if (lead && fin.rmean) {
                    const double unb = fin.count > 1. ? var * fin.count / (fin.count - 1.) : var;
                    fin.rmean[c] = (1.f - fin.momentum) * fin.rmean[c] + fin.momentum * (float)m;
                    fin.rvar[c] = (1.f - fin.momentum) * fin.rvar[c] + fin.momentum * (float)unb;
                }
            }
            s_coef[c] = scv; s_coef[C + c] = shv;
            if (lead) { fin.scale[c] = scv; fin.shift[c] = shv; fin.mean[c] = mv; fin.invstd[c] = isv; }
        }
        __syncthreads();
        if (pl >= PPB) return;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = s_coef[cg * 8 + e]; sh[e] = s_coef[C + cg * 8 + e]; }
    } else {
        if (pl >= PPB) return;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = scale[cg * 8 + e]; sh[e] = shift[cg * 8 + e]; }
    }
    const int OH = POOL ? H / 2 : H, OW = POOL ? W / 2 : W;
    const unsigned P = (unsigned)N * OH * OW, stride = gridDim.x * PPB;
    PixWalk w;
    w.init(blockIdx.x * PPB + pl, stride, OH, OW);
    constexpr int R = POOL ? 2 : 1;
    for (unsigned p = blockIdx.x * PPB + pl; p < P; p += stride, w.next()) {
        const int n = w.n, y = w.y, x = w.x;
        float v[R * R][8];
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < R; ++j)
                El<E>::ld8_nt(raw + (((size_t)n * H + y * R + i) * W + x * R + j) * C + cg * 8, v[i * R + j]);
        float mx[8], rsel[8];
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int yy = y * R + i, xx = x * R + j;
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = act_fwd(__builtin_fmaf(v[i * R + j][e], sc[e], sh[e]), act);   // (explicit fma: bn_bwd_g_window recomputes it)
                // keep (pooled layers): the full-resolution activation is only ever read for the frames that feed a skip
                // connection (one per sample); the pooled tensor carries everything else forward
                if (dst && (!keep || keep[n])) {
                    // dst_s2d: the activated tensor is stored SPACE-TO-DEPTH for a 4x4 stride-2 consumer, [N][H/2+2][W/2+2][4C] with a 1-pixel
                    // border: pixel (y, x) -> position (y/2, x/2), channel group (y&1)*2 + (x&1) (srvp_conv_desc.tap_phase_chunks)
                    size_t doff = dst_s2d ? ((((size_t)n * (H / 2 + 2) + (yy >> 1) + 1) * (W / 2 + 2) + (xx >> 1) + 1) * 4 + ((yy & 1) * 2 + (xx & 1))) * C + cg * 8
                                          : (((size_t)n * (H + 2 * db) + yy + db) * (W + 2 * db) + xx + db) * C + cg * 8;
                    El<E>::st8_nt(dst + doff, f);
                }
                if (dst_f32) {
                    const size_t off = (((size_t)n * H + yy) * W + xx) * C + cg * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) dst_f32[off + e] = f[e];
                }
                if (POOL) {
                    // pooled values are taken from the activations as stored (bf16-rounded: what the consumer of `dst` sees)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        // (strictly greater replaces: torch's first-max tie rule, the one the backward routes the gradient by; rsel = the RAW
                        // value at that position -- srvp_bn_finalize_act's raw_pool, what the fused backward reduction of a pooled layer reads)
                        const float fr = El<E>::rnd(f[e]);
                        const bool take = (i == 0 && j == 0) || fr > mx[e];
                        mx[e] = take ? fr : mx[e];
                        rsel[e] = take ? v[i * R + j][e] : rsel[e];
                    }
                }
            }
        if (POOL) {
            size_t poff = (((size_t)n * (OH + 2 * pb) + y + pb) * (OW + 2 * pb) + x + pb) * C + cg * 8;
            El<E>::st8(dpool + poff, mx);
            if (rawpool) El<E>::st8(rawpool + (((size_t)n * OH + y) * OW + x) * C + cg * 8, rsel);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------
struct BnBwdK {
    const void* raw; const void* act; int act_border;
    const float* scale; const float* shift; const float* mean; const float* invstd; int act_kind;
    const void* da; int da_mode, da_cstride, da_coff, da_border, da_is_f32;
    const void* da2; const int* da2_idx;
    int N, H, W, C;
    void* tsum; int tsum_T;        // apply: also write the sum over the T time steps of each sample (frames are t*B + b)
    int s2d;                         // apply: draw is written space-to-depth [N][H/2+2][W/2+2][4C] (border 1)
    int det;                         // reduce: `red` is the slab [workgroup][2][C] of the deterministic two-launch form
};

// element offset of pixel (n, y, x), channel group cg in the gradient tensor `draw`
__device__ __forceinline__ size_t draw_off(const BnBwdK& a, int n, int y, int x, int cg, int db) {
    if (a.s2d)
        return ((((size_t)n * (a.H / 2 + 2) + (y >> 1) + 1) * (a.W / 2 + 2) + (x >> 1) + 1) * 4 + ((y & 1) * 2 + (x & 1))) * a.C + cg * 8;
    return (((size_t)n * (a.H + 2 * db) + y + db) * (a.W + 2 * db) + x + db) * a.C + cg * 8;
}

// g[8] = dA * f'(pre) for pixel (n,y,x), channel group cg; also returns raw values
template <class E, int MODE, int ACT>
__device__ __forceinline__ void bn_bwd_g(const BnBwdK& a, int n, int y, int x, int cg, const float* sc, const float* sh,
                                         float* g, float* rawf) {
    const int C = a.C, H = a.H, W = a.W;
    size_t roff = (((size_t)n * H + y) * W + x) * C + cg * 8;
    El<E>::ld8_nt((const E*)a.raw + roff, rawf);
    float d[8];
    const int db = a.da_border, cs = a.da_cstride, co = a.da_coff + cg * 8;
    if (a.da_is_f32) {
        const float* da = (const float*)a.da;
        size_t off = (((size_t)n * (H + 2 * db) + y + db) * (W + 2 * db) + x + db) * cs + co;
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = da[off + e];
    } else {
        const E* da = (const E*)a.da;
        if constexpr (MODE == 0) {
            size_t off = (((size_t)n * (H + 2 * db) + y + db) * (W + 2 * db) + x + db) * cs + co;
            El<E>::ld8_nt(da + off, d);
        } else if constexpr (MODE == 1) {
            const int H2 = 2 * H, W2 = 2 * W;
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    size_t off = (((size_t)n * (H2 + 2 * db) + 2 * y + i + db) * (W2 + 2 * db) + 2 * x + j + db) * cs + co;
                    float t[8];
                    El<E>::ld8(da + off, t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[e] += t[e];
                }
        } else {
            const int Hh = H / 2, Wh = W / 2;
            size_t off = (((size_t)n * (Hh + 2 * db) + (y >> 1) + db) * (Wh + 2 * db) + (x >> 1) + db) * cs + co;
            float t[8];
            El<E>::ld8(da + off, t);
            // arg-max routing with torch's first-max tie rule (scan order (0,0),(0,1),(1,0),(1,1) of the window)
            const int ab = a.act_border;
            const int y0 = y & ~1, x0 = x & ~1;
            const int me = (y & 1) * 2 + (x & 1);
            float w4[4][8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    size_t aoff = (((size_t)n * (H + 2 * ab) + y0 + i + ab) * (W + 2 * ab) + x0 + j + ab) * C + cg * 8;
                    El<E>::ld8((const E*)a.act + aoff, w4[i * 2 + j]);
                }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int best = 0; float bv = w4[0][e];
#pragma unroll
                for (int q = 1; q < 4; ++q) if (w4[q][e] > bv) { bv = w4[q][e]; best = q; }
                d[e] = (best == me) ? t[e] : 0.f;
            }
        }
    }
    if (a.da2) {
        int idx = a.da2_idx ? a.da2_idx[n] : n;
        if (idx >= 0) {
            size_t off = (((size_t)idx * H + y) * W + x) * C + cg * 8;
            float t[8];
            El<E>::ld8((const E*)a.da2 + off, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] += t[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = d[e] * act_bwd(rawf[e] * sc[e] + sh[e], ACT >= 0 ? ACT : a.act_kind);
}

// Pooled consumer (da_mode 2), whole 2x2 window (py, px) of image n at once: g[q][8] / raw[q][8] for the window pixels
// q = 2*i + j.  Arg-max routing with torch's first-max tie rule (scan order (0,0),(0,1),(1,0),(1,1)).
template <class E, int ACT>
__device__ __forceinline__ void bn_bwd_window_load(const BnBwdK& a, int n, int py, int px, int cg, float* t, float (*rawf)[8]) {
    // the five loads of a window (pooled gradient + the four raw pixels), kept apart from the arithmetic so that a caller can put
    // the loads of several windows in flight before the first use
    const int C = a.C, H = a.H, W = a.W, db = a.da_border;
    const int Hh = H / 2, Wh = W / 2;
    El<E>::ld8_nt((const E*)a.da + (((size_t)n * (Hh + 2 * db) + py + db) * (Wh + 2 * db) + px + db) * a.da_cstride + a.da_coff + cg * 8, t);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int y = 2 * py + (q >> 1), x = 2 * px + (q & 1);
        El<E>::ld8_nt((const E*)a.raw + (((size_t)n * H + y) * W + x) * C + cg * 8, rawf[q]);
    }
}

template <class E, int ACT>
__device__ __forceinline__ void bn_bwd_window_compute(const BnBwdK& a, int n, int py, int px, int cg, const float* sc, const float* sh,
                                                      const float* t, const float (*rawf)[8], float (*g)[8]) {
    const int C = a.C, H = a.H, W = a.W;
    float w4[4][8];
    // The activations the forward max-pooled over are RECOMPUTED from raw (same fma, same activation, same bf16 rounding as
    // bn_act_kernel stored them) instead of being read back: one full-size tensor read less in both BN-backward passes of the
    // four pooled layers.
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            w4[q][e] = El<E>::rnd(act_fwd(__builtin_fmaf(rawf[q][e], sc[e], sh[e]), ACT >= 0 ? ACT : a.act_kind));
    float d[4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int best = 0; float bv = w4[0][e];
#pragma unroll
        for (int q = 1; q < 4; ++q) if (w4[q][e] > bv) { bv = w4[q][e]; best = q; }
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q][e] = (best == q) ? t[e] : 0.f;
    }
    if (a.da2) {
        const int idx = a.da2_idx ? a.da2_idx[n] : n;
        if (idx >= 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int y = 2 * py + (q >> 1), x = 2 * px + (q & 1);
                float u[8];
                El<E>::ld8((const E*)a.da2 + (((size_t)idx * H + y) * W + x) * C + cg * 8, u);
#pragma unroll
                for (int e = 0; e < 8; ++e) d[q][e] += u[e];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) g[q][e] = d[q][e] * act_bwd(rawf[q][e] * sc[e] + sh[e], ACT >= 0 ? ACT : a.act_kind);
}

// one window, loads and arithmetic together
template <class E, int ACT>
__device__ __forceinline__ void bn_bwd_g_window(const BnBwdK& a, int n, int py, int px, int cg, const float* sc, const float* sh,
                                                float (*g)[8], float (*rawf)[8]) {
    float t[8];
    bn_bwd_window_load<E, ACT>(a, n, py, px, cg, t, rawf);
    bn_bwd_window_compute<E, ACT>(a, n, py, px, cg, sc, sh, t, rawf, g);
}

template <class E, int MODE, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdK a, double* red) {
    const int CG = a.C / 8;
    const int PPB = blockDim.x / CG;             // pixels handled in parallel by one workgroup
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    __shared__ float sred[256][17];
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (pl < PPB) {
        float sc[8], sh[8], mu[8], is[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = a.scale[cg * 8 + e]; sh[e] = a.shift[cg * 8 + e];
            mu[e] = a.mean ? a.mean[cg * 8 + e] : 0.f; is[e] = a.invstd ? a.invstd[cg * 8 + e] : 0.f;
        }
        if constexpr (MODE == 3) {
            // pooled consumer whose own gradient is reduced ELSEWHERE (da_mode 3: the consumer's data-gradient launch carries raw_pool and
            // accumulates the arg-max terms, srvp_conv_desc.bnr_*): only the frames that also receive a skip-connection gradient are walked
            // here -- a.N of them, row j of da2 belongs to frame da2_idx[j] -- every pixel with the da2 term alone (both sums are linear in
            // the gradient).  Four pixels per thread and iteration, all eight loads up front.
            const unsigned P = (unsigned)a.N * a.H * a.W, stride = gridDim.x * PPB;
            const unsigned hw = (unsigned)a.H * a.W;
            constexpr int U = 4;
            for (unsigned p = blockIdx.x * (PPB * U) + pl; p < P; p += U * stride) {
                float rv[U][8], uv[U][8];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const unsigned pu = p + u * PPB;
                    ok[u] = pu < P;
                    const unsigned pc = ok[u] ? pu : 0u;
                    const unsigned j = pc / hw, px = pc - j * hw;
                    const int n = a.da2_idx[j];
                    El<E>::ld8_nt((const E*)a.raw + ((size_t)n * hw + px) * a.C + cg * 8, rv[u]);
                    El<E>::ld8_nt((const E*)a.da2 + (size_t)pc * a.C + cg * 8, uv[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float m = ok[u] ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float g = m * uv[u][e] * act_bwd(rv[u][e] * sc[e] + sh[e], ACT >= 0 ? ACT : a.act_kind);
                        s1[e] += g; s2[e] += g * (rv[u][e] - mu[e]) * is[e];
                    }
                }
            }
        } else if constexpr (MODE == 2) {
            // pooled: one thread per 2x2 window (window activations and the pooled gradient are read once)
            // (one window per iteration: this loop is VALU-bound -- the four activations of a window are recomputed, rounded and
            // arg-max-routed, ~800 instructions per 80 bytes -- and two windows in flight only cost occupancy: 477 -> 613 us)
            const unsigned P = (unsigned)a.N * (a.H / 2) * (a.W / 2), stride = gridDim.x * PPB;
            PixWalk w;
            w.init(blockIdx.x * PPB + pl, stride, a.H / 2, a.W / 2);
            for (unsigned p = blockIdx.x * PPB + pl; p < P; p += stride, w.next()) {
                float g[4][8], rawf[4][8];
                const int idx = a.da2 ? (a.da2_idx ? a.da2_idx[w.n] : w.n) : -1;
                if (idx >= 0) {
                    // frames that also receive a skip-connection gradient (one per sample): every window pixel contributes
                    bn_bwd_g_window<E, ACT>(a, w.n, w.y, w.x, cg, sc, sh, g, rawf);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s1[e] += g[q][e]; s2[e] += g[q][e] * (rawf[q][e] - mu[e]) * is[e]; }
                    continue;
                }
                // otherwise only the arg-max pixel of a window carries gradient: the other three terms of both sums are exactly
                // zero, so they are not formed (a third less arithmetic in this VALU-bound loop; same sums)
                float t[8];
                bn_bwd_window_load<E, ACT>(a, w.n, w.y, w.x, cg, t, rawf);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float bv = El<E>::rnd(act_fwd(__builtin_fmaf(rawf[0][e], sc[e], sh[e]), ACT >= 0 ? ACT : a.act_kind)), rb = rawf[0][e];
#pragma unroll
                    for (int q = 1; q < 4; ++q) {
                        const float v = El<E>::rnd(act_fwd(__builtin_fmaf(rawf[q][e], sc[e], sh[e]), ACT >= 0 ? ACT : a.act_kind));
                        if (v > bv) { bv = v; rb = rawf[q][e]; }             // torch's first-max tie rule: strictly greater replaces
                    }
                    const float gb = t[e] * act_bwd(rb * sc[e] + sh[e], ACT >= 0 ? ACT : a.act_kind);
                    s1[e] += gb; s2[e] += gb * (rb - mu[e]) * is[e];
                }
            }
        } else {
            const unsigned P = (unsigned)a.N * a.H * a.W, stride = gridDim.x * PPB;
            if (MODE == 0 && !a.da_is_f32 && !a.da2 && a.da_border == 0) {
                // plain consumer, unbordered gradient: both tensors are linear in the pixel index -- four pixels per
                // iteration with all eight 16-byte loads issued up front (memory-level parallelism; the generic loop
                // below has one pixel in flight per thread because its loads sit behind the loop-exit test)
                // (the U groups of a workgroup are CONSECUTIVE pixels -- U * PPB * C contiguous elements per stream and iteration:
                // a two-stream read microbenchmark sustains 7.1 TB/s with this pattern against 5.6-6.8 with the groups one grid
                // apart)
                constexpr int U = 4;
                const E* da = (const E*)a.da;
                const unsigned p0 = blockIdx.x * (PPB * U) + pl;
                for (unsigned p = p0; p < P; p += U * stride) {
                    float rv[U][8], dv[U][8];
                    bool ok[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const unsigned pu = p + u * PPB;
                        ok[u] = pu < P;
                        const size_t pc = ok[u] ? pu : p0;
                        El<E>::ld8_nt((const E*)a.raw + pc * a.C + cg * 8, rv[u]);
                        El<E>::ld8_nt(da + pc * a.da_cstride + a.da_coff + cg * 8, dv[u]);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float* rawf = rv[u];
                        const float* d = dv[u];
                        const float m = ok[u] ? 1.f : 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float g = m * d[e] * act_bwd(rawf[e] * sc[e] + sh[e], ACT >= 0 ? ACT : a.act_kind);
                            s1[e] += g; s2[e] += g * (rawf[e] - mu[e]) * is[e];
                        }
                    }
                }
            } else {
                PixWalk w;
                w.init(blockIdx.x * PPB + pl, stride, a.H, a.W);
#pragma unroll 2
                for (unsigned p = blockIdx.x * PPB + pl; p < P; p += stride, w.next()) {
                    float g[8], rawf[8];
                    bn_bwd_g<E, MODE, ACT>(a, w.n, w.y, w.x, cg, sc, sh, g, rawf);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s1[e] += g[e]; s2[e] += g[e] * (rawf[e] - mu[e]) * is[e]; }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sred[threadIdx.x][e] = s1[e]; sred[threadIdx.x][8 + e] = s2[e]; }
    __syncthreads();
    // thread t < CG*16 sums column (cg = t / 16, k = t % 16) over the PPB pixel lanes
    for (int t = threadIdx.x; t < CG * 16; t += blockDim.x) {
        int c = t / 16, k = t % 16;
        double s = 0.;
        for (int l = 0; l < PPB; ++l) s += sred[l * CG + c][k];
        int ch = c * 8 + (k & 7);
        if (a.det) red[(size_t)blockIdx.x * 2 * a.C + (k >> 3) * a.C + ch] = s;       // (every workgroup writes all 2 C values)
        else atomicAdd(red + (k >> 3) * a.C + ch, s);
    }
}

__global__ void bn_bwd_finalize_kernel(const double* red, double count, const float* scale, const float* mean,
                                       const float* invstd, float* dgamma, float* dbeta, float* coef, int C, int C_real,
                                       int has_bn, float pscale) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (c >= C_real) { coef[c] = 0.f; coef[C + c] = 0.f; coef[2 * C + c] = 0.f; return; }
    if (!has_bn) { coef[c] = 1.f; coef[C + c] = 0.f; coef[2 * C + c] = 0.f; return; }
    double sg = red[c], sgx = red[C + c];
    if (dgamma) dgamma[c] += (float)(sgx * pscale);
    if (dbeta) dbeta[c] += (float)(sg * pscale);
    double mg = sg / count, mgx = sgx / count;
    double k1 = scale[c];
    double k3 = -k1 * mgx * invstd[c];
    double k2 = -k1 * mg - k3 * mean[c];
    coef[c] = (float)k1; coef[C + c] = (float)k2; coef[2 * C + c] = (float)k3;
}

template <class E, int MODE, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdK a, const float* __restrict__ coef,
                                                           E* __restrict__ draw, int db, const BnBwdFin fin) {
    const int CG = a.C / 8;
    const int PPB = blockDim.x / CG;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    float sc[8], sh[8], k1[8], k2[8], k3[8];
    if (fin.red) {
        __shared__ float s_k[3 * BN_MAX_C];
        const bool lead = blockIdx.x == 0;
        const int C = a.C;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float c1 = 0.f, c2 = 0.f, c3 = 0.f;
            if (c < fin.C_real) {
                const double sg = fin.red[c], sgx = fin.red[C + c];
                if (lead) {
                    if (fin.dgamma) fin.dgamma[c] += (float)(sgx * fin.pscale);
                    if (fin.dbeta) fin.dbeta[c] += (float)(sg * fin.pscale);
                }
                const double mg = sg / fin.count, mgx = sgx / fin.count;
                const double k1d = a.scale[c];
                const double k3d = -k1d * mgx * a.invstd[c];
                const double k2d = -k1d * mg - k3d * a.mean[c];
                c1 = (float)k1d; c2 = (float)k2d; c3 = (float)k3d;
            }
            s_k[c] = c1; s_k[C + c] = c2; s_k[2 * C + c] = c3;
            if (lead && fin.coef) { fin.coef[c] = c1; fin.coef[C + c] = c2; fin.coef[2 * C + c] = c3; }
        }
        __syncthreads();
        if (pl >= PPB) return;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = a.scale[cg * 8 + e]; sh[e] = a.shift[cg * 8 + e];
            k1[e] = s_k[cg * 8 + e]; k2[e] = s_k[C + cg * 8 + e]; k3[e] = s_k[2 * C + cg * 8 + e];
        }
    } else {
        if (pl >= PPB) return;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = a.scale[cg * 8 + e]; sh[e] = a.shift[cg * 8 + e];
            k1[e] = coef[cg * 8 + e]; k2[e] = coef[a.C + cg * 8 + e]; k3[e] = coef[2 * a.C + cg * 8 + e];
        }
    }
    if constexpr (MODE == 2) {
        const unsigned P = (unsigned)a.N * (a.H / 2) * (a.W / 2), stride = gridDim.x * PPB;
        constexpr int U = 2;                         // two consecutive window groups per iteration, loads up front (see the reduction)
        const unsigned p0 = blockIdx.x * (PPB * U) + pl;
        PixWalk wk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) wk[u].init(p0 + u * PPB, U * stride, a.H / 2, a.W / 2);
        for (unsigned p = p0; p < P; p += U * stride) {
            float g[4][8], rawf[U][4][8], t[U][8];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = p + u * PPB < P;
                if (!ok[u]) { wk[u].n = 0; wk[u].y = 0; wk[u].x = 0; }
                bn_bwd_window_load<E, ACT>(a, wk[u].n, wk[u].y, wk[u].x, cg, t[u], rawf[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                bn_bwd_window_compute<E, ACT>(a, wk[u].n, wk[u].y, wk[u].x, cg, sc, sh, t[u], rawf[u], g);
                if (ok[u]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = k1[e] * g[q][e] + k2[e] + k3[e] * rawf[u][q][e];
                        size_t off = (((size_t)wk[u].n * (a.H + 2 * db) + 2 * wk[u].y + (q >> 1) + db) * (a.W + 2 * db) + 2 * wk[u].x + (q & 1) + db) * a.C + cg * 8;
                        El<E>::st8(draw + off, o);
                    }
                }
                wk[u].next();
            }
        }
        return;
    }
    if (a.tsum) {
        // hoisted-skip blocks also need sum_t draw (weight / data gradient of the per-sample skip half): walk the pixels of
        // the B samples and loop over time inside, so the sum costs one extra store instead of re-reading draw
        const int Bs = a.N / a.tsum_T;
        const unsigned P = (unsigned)Bs * a.H * a.W, stride = gridDim.x * PPB;
        if (MODE == 0 && !a.da_is_f32 && !a.da2 && a.da_border == 0) {
            // U consecutive pixel groups per thread and time step, their 2 U loads issued up front; running sums over t in registers
            // (same values and the same summation order over t as the one-pixel loop below)
            constexpr int U = 4;
            const E* da = (const E*)a.da;
            const size_t hw = (size_t)a.H * a.W;
            const unsigned p0 = blockIdx.x * (PPB * U) + pl;
            PixWalk wk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) wk[u].init(p0 + u * PPB, U * stride, a.H, a.W);
            for (unsigned p = p0; p < P; p += U * stride) {
                float acc[U][8];
                bool ok[U];
                size_t pix[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    ok[u] = p + u * PPB < P;
                    if (!ok[u]) { wk[u].n = 0; wk[u].y = 0; wk[u].x = 0; }      // (past the end: reads pixel 0, stores nothing; last iteration)
                    pix[u] = (size_t)wk[u].n * hw + (size_t)wk[u].y * a.W + wk[u].x;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[u][e] = 0.f;
                }
                for (int t = 0; t < a.tsum_T; ++t) {
                    float rv[U][8], dv[U][8];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const size_t pc = (size_t)t * Bs * hw + pix[u];
                        El<E>::ld8_nt((const E*)a.raw + pc * a.C + cg * 8, rv[u]);
                        El<E>::ld8_nt(da + pc * a.da_cstride + a.da_coff + cg * 8, dv[u]);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float g = dv[u][e] * act_bwd(rv[u][e] * sc[e] + sh[e], ACT >= 0 ? ACT : a.act_kind);
                            o[e] = k1[e] * g + k2[e] + k3[e] * rv[u][e]; acc[u][e] += o[e];
                        }
                        if (ok[u]) El<E>::st8(draw + draw_off(a, t * Bs + wk[u].n, wk[u].y, wk[u].x, cg, db), o);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (ok[u]) {
                        const size_t soff = (((size_t)wk[u].n * (a.H + 2 * db) + wk[u].y + db) * (a.W + 2 * db) + wk[u].x + db) * a.C + cg * 8;
                        El<E>::st8((E*)a.tsum + soff, acc[u]);
                    }
                    wk[u].next();
                }
            }
            return;
        }
        PixWalk w;
        w.init(blockIdx.x * PPB + pl, stride, a.H, a.W);
        for (unsigned p = blockIdx.x * PPB + pl; p < P; p += stride, w.next()) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int t = 0; t < a.tsum_T; ++t) {
                const int n = t * Bs + w.n;
                float g[8], rawf[8], o[8];
                bn_bwd_g<E, MODE, ACT>(a, n, w.y, w.x, cg, sc, sh, g, rawf);
#pragma unroll
                for (int e = 0; e < 8; ++e) { o[e] = k1[e] * g[e] + k2[e] + k3[e] * rawf[e]; acc[e] += o[e]; }
                El<E>::st8(draw + draw_off(a, n, w.y, w.x, cg, db), o);
            }
            size_t soff = (((size_t)w.n * (a.H + 2 * db) + w.y + db) * (a.W + 2 * db) + w.x + db) * a.C + cg * 8;
            El<E>::st8((E*)a.tsum + soff, acc);
        }
        return;
    }
    const unsigned P = (unsigned)a.N * a.H * a.W, stride = gridDim.x * PPB;
    if (MODE == 0 && !a.da_is_f32 && !a.da2 && a.da_border == 0) {
        // plain consumer, unbordered gradient (both inputs linear in the pixel index): U consecutive pixel groups per iteration, all
        // 2 U 16-byte loads issued before the first use (see bn_bwd_reduce_kernel; a read-read-write microbenchmark sustains
        // 6.5 TB/s with this pattern, the one-pixel loop below ran at 5.1-6.0)
        constexpr int U = 4;
        const E* da = (const E*)a.da;
        const unsigned p0 = blockIdx.x * (PPB * U) + pl;
        PixWalk wk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) wk[u].init(p0 + u * PPB, U * stride, a.H, a.W);
        for (unsigned p = p0; p < P; p += U * stride) {
            float rv[U][8], dv[U][8];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned pu = p + u * PPB;
                ok[u] = pu < P;
                const size_t pc = ok[u] ? pu : p0;
                El<E>::ld8_nt((const E*)a.raw + pc * a.C + cg * 8, rv[u]);
                El<E>::ld8_nt(da + pc * a.da_cstride + a.da_coff + cg * 8, dv[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float g = dv[u][e] * act_bwd(rv[u][e] * sc[e] + sh[e], ACT >= 0 ? ACT : a.act_kind);
                    o[e] = k1[e] * g + k2[e] + k3[e] * rv[u][e];
                }
                if (ok[u]) El<E>::st8(draw + draw_off(a, wk[u].n, wk[u].y, wk[u].x, cg, db), o);
                wk[u].next();
            }
        }
        return;
    }
    PixWalk w;
    w.init(blockIdx.x * PPB + pl, stride, a.H, a.W);
#pragma unroll 2
    for (unsigned p = blockIdx.x * PPB + pl; p < P; p += stride, w.next()) {
        const int n = w.n, y = w.y, x = w.x;
        float g[8], rawf[8], o[8];
        bn_bwd_g<E, MODE, ACT>(a, n, y, x, cg, sc, sh, g, rawf);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = k1[e] * g[e] + k2[e] + k3[e] * rawf[e];
        El<E>::st8(draw + draw_off(a, n, y, x, cg, db), o);
    }
}

int fill_k(const srvp_bnbwd_desc* d, BnBwdK& k) {
    SRVP_REQUIRE(d && d->raw && d->scale && d->shift && d->da, "srvp_bn_bwd: null pointer");
    SRVP_REQUIRE(d->C % 8 == 0 && d->C / 8 <= 256, "srvp_bn_bwd: C=%d unsupported", d->C);
    SRVP_REQUIRE(d->da_mode != 2 || d->act, "srvp_bn_bwd: pooled mode needs the activated tensor");
    SRVP_REQUIRE(d->da_mode != 2 || (d->H % 2 == 0 && d->W % 2 == 0), "srvp_bn_bwd: pooled mode needs even H, W");
    k.raw = d->raw; k.act = d->act; k.act_border = d->act_border;
    k.scale = d->scale; k.shift = d->shift; k.mean = d->mean; k.invstd = d->invstd; k.act_kind = d->act_kind;
    k.da = d->da; k.da_mode = d->da_mode; k.da_cstride = d->da_cstride; k.da_coff = d->da_coff;
    k.da_border = d->da_border; k.da_is_f32 = d->da_is_f32; k.da2 = d->da2; k.da2_idx = d->da2_idx;
    k.N = d->N; k.H = d->H; k.W = d->W; k.C = d->C;
    k.tsum = d->tsum; k.tsum_T = d->tsum_T;
    k.s2d = d->draw_s2d;
    SRVP_REQUIRE(!d->draw_s2d || (d->da_mode != 2 && d->H % 2 == 0 && d->W % 2 == 0), "srvp_bn_bwd: draw_s2d needs even H, W and a non-pooled consumer");
    SRVP_REQUIRE(!d->tsum || (d->tsum_T > 0 && d->N % d->tsum_T == 0 && d->da_mode != 2), "srvp_bn_bwd: tsum needs N %% T == 0 and a non-pooled consumer");
    return SRVP_OK;
}

inline unsigned grid_for(long long work_items, int per_block) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int srvp_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* nbt, float* scale, float* shift,
                                float* mean, float* invstd, int C, int C_real, float eps, float momentum, void* stream) {
    SRVP_REQUIRE(stats && scale && shift && mean && invstd && C > 0, "srvp_bn_finalize: bad args");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, stats, count, gamma,
                       beta, running_mean, running_var, (long long*)nbt, scale, shift, mean, invstd, C, C_real, eps, momentum);
    SRVP_CHECK_LAUNCH("srvp_bn_finalize");
    return SRVP_OK;
}

extern "C" int srvp_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float* scale, float* shift, int C, int C_real, float eps,
                                   void* stream) {
    SRVP_REQUIRE(scale && shift && C > 0, "srvp_bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, scale, shift, C, C_real, eps);
    SRVP_CHECK_LAUNCH("srvp_bn_eval_coeffs");
    return SRVP_OK;
}

namespace {
template <class E>
int bn_act_launch(const void* raw, const float* scale, const float* shift, int act, int N, int H, int W, int C, void* dst,
                  int dst_border, void* dst_pool, int pool_border, float* dst_f32, const int32_t* keep, void* stream,
                  const BnFin fin = BnFin{}, int dst_s2d = 0, void* raw_pool = nullptr) {
    SRVP_REQUIRE(!dst_s2d || (dst && !dst_pool && H % 2 == 0 && W % 2 == 0), "srvp_bn_act: a space-to-depth destination needs even H, W and no pooling");
    SRVP_REQUIRE(raw && scale && shift && C % 8 == 0, "srvp_bn_act: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (dst_pool) {
        SRVP_REQUIRE(H % 2 == 0 && W % 2 == 0, "srvp_bn_act: pooling needs even H, W");
        long long total = (long long)N * (H / 2) * (W / 2);
        SRVP_REQUIRE(C / 8 <= 256 && (long long)N * H * W < (1ll << 31), "srvp_bn_act: C=%d / size unsupported", C);
        auto kern = act == ACT_LRELU ? bn_act_kernel<E, true, ACT_LRELU> : bn_act_kernel<E, true, -1>;
        hipLaunchKernelGGL(kern, dim3(grid_for(total, (256 / (C / 8)) * 2)), dim3(256), 0, st, (const E*)raw, scale, shift,
                           act, N, H, W, C, (E*)dst, dst_border, (E*)dst_pool, pool_border, dst_f32, (const int*)keep, fin, dst_s2d, (E*)raw_pool);
    } else {
        long long total = (long long)N * H * W;
        SRVP_REQUIRE(C / 8 <= 256 && total < (1ll << 31), "srvp_bn_act: C=%d / size unsupported", C);
        auto kern = act == ACT_LRELU ? bn_act_kernel<E, false, ACT_LRELU> : bn_act_kernel<E, false, -1>;
        hipLaunchKernelGGL(kern, dim3(grid_for(total, (256 / (C / 8)) * 4)), dim3(256), 0, st, (const E*)raw, scale, shift,
                           act, N, H, W, C, (E*)dst, dst_border, (E*)nullptr, 0, dst_f32, (const int*)nullptr, fin, dst_s2d, (E*)nullptr);
    }
    SRVP_CHECK_LAUNCH("srvp_bn_act");
    return SRVP_OK;
}
}  // namespace

extern "C" int srvp_bn_act_keep(const void* raw, const float* scale, const float* shift, int act, int N, int H, int W, int C,
                                void* dst, int dst_border, void* dst_pool, int pool_border, float* dst_f32, const int32_t* keep,
                                void* stream) {
    return bn_act_launch<bf16_t>(raw, scale, shift, act, N, H, W, C, dst, dst_border, dst_pool, pool_border, dst_f32, keep, stream);
}
extern "C" int srvp_bn_act_keep_f32(const void* raw, const float* scale, const float* shift, int act, int N, int H, int W, int C,
                                    void* dst, int dst_border, void* dst_pool, int pool_border, float* dst_f32, const int32_t* keep,
                                    void* stream) {
    return bn_act_launch<float>(raw, scale, shift, act, N, H, W, C, dst, dst_border, dst_pool, pool_border, dst_f32, keep, stream);
}

extern "C" int srvp_bn_finalize_act(const void* raw, const double* stats, double count, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, int64_t* nbt, float* scale, float* shift, float* mean,
                                    float* invstd, int C_real, float eps, float momentum, int act, int N, int H, int W, int C, void* dst,
                                    int dst_border, void* dst_pool, int pool_border, float* dst_f32, const int32_t* keep, void* raw_pool,
                                    int elem_f32, int dst_s2d, void* stream) {
    SRVP_REQUIRE(stats && scale && shift && mean && invstd && C > 0 && C <= BN_MAX_C && count > 0, "srvp_bn_finalize_act: bad args");
    SRVP_REQUIRE(!raw_pool || dst_pool, "srvp_bn_finalize_act: raw_pool needs dst_pool");
    BnFin fin{stats, count, gamma, beta, running_mean, running_var, (long long*)nbt, scale, shift, mean, invstd, C_real, eps, momentum};
    if (elem_f32) return bn_act_launch<float>(raw, scale, shift, act, N, H, W, C, dst, dst_border, dst_pool, pool_border, dst_f32, keep, stream, fin, dst_s2d, raw_pool);
    return bn_act_launch<bf16_t>(raw, scale, shift, act, N, H, W, C, dst, dst_border, dst_pool, pool_border, dst_f32, keep, stream, fin, dst_s2d, raw_pool);
}
// srvp_bn_act_keep with a space-to-depth destination (eval mode / separate-finalize path of a block whose consumer is a 4x4 stride-2 conv)
extern "C" int srvp_bn_act_s2d(const void* raw, const float* scale, const float* shift, int act, int N, int H, int W, int C, void* dst,
                               void* stream) {
    return bn_act_launch<bf16_t>(raw, scale, shift, act, N, H, W, C, dst, 1, nullptr, 0, nullptr, nullptr, stream, BnFin{}, 1);
}

extern "C" int srvp_bn_act(const void* raw, const float* scale, const float* shift, int act, int N, int H, int W, int C,
                           void* dst, int dst_border, void* dst_pool, int pool_border, float* dst_f32, void* stream) {
    return srvp_bn_act_keep(raw, scale, shift, act, N, H, W, C, dst, dst_border, dst_pool, pool_border, dst_f32, nullptr, stream);
}

extern "C" int srvp_bn_bwd_reduce(const srvp_bnbwd_desc* d, double* red, void* stream) {
    BnBwdK k;
    int rc = fill_k(d, k);
    if (rc) return rc;
    SRVP_REQUIRE(red, "srvp_bn_bwd_reduce: null red");
    const int CG = k.C / 8, PPB = 256 / CG;
    long long P = (long long)k.N * k.H * k.W;
    SRVP_REQUIRE(P < (1ll << 31), "srvp_bn_bwd_reduce: too many pixels");
    SRVP_REQUIRE(k.da_mode != 3 || (k.da2 && k.da2_idx && !d->elem_f32), "srvp_bn_bwd_reduce: da_mode 3 (skip-gradient term only) needs da2, da2_idx (row -> frame), bf16 tensors");
    if (k.da_mode == 2) P /= 4;
    // >= 64 pixel rows per thread slot: the 2C fp64 atomics per workgroup must stay small beside its streaming work -- unless that
    // leaves fewer than 64 workgroups (the 1x1 encoder output: 3 workgroups walked 2304 rows in 100 us of dependent loads)
    int rows = k.da_mode == 2 ? 16 : 64;
    while (rows > 4 && P / ((long long)PPB * rows) < 64) rows /= 2;
    dim3 g(grid_for(P, PPB * rows));
    const bool lr = k.act_kind == ACT_LRELU;
    double* red_out = red;
    k.det = 0;
    if (g_srvp_det && d->elem_f32) {
        // deterministic mode: at most 256 workgroups (the kernel strides over the pixels), partial sums into the workspace, added up in
        // workgroup order by a second launch
        SRVP_REQUIRE(g_srvp_det_ws && 256ll * 2 * k.C * 8 <= g_srvp_det_ws_bytes, "srvp_bn_bwd_reduce: deterministic workspace too small");
        if (g.x > 256) g.x = 256;
        k.det = 1;
        red = (double*)g_srvp_det_ws;
    }
    if (d->elem_f32) {
        auto kern = k.da_mode == 0 ? (lr ? bn_bwd_reduce_kernel<float, 0, ACT_LRELU> : bn_bwd_reduce_kernel<float, 0, -1>)
                  : k.da_mode == 1 ? (lr ? bn_bwd_reduce_kernel<float, 1, ACT_LRELU> : bn_bwd_reduce_kernel<float, 1, -1>)
                                   : (lr ? bn_bwd_reduce_kernel<float, 2, ACT_LRELU> : bn_bwd_reduce_kernel<float, 2, -1>);
        hipLaunchKernelGGL(kern, g, dim3(256), 0, (hipStream_t)stream, k, red);
        if (k.det) hipLaunchKernelGGL(det_sum_kernel<double>, dim3((2 * k.C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double*)red, (int)g.x, 2 * k.C, red_out);
        SRVP_CHECK_LAUNCH("srvp_bn_bwd_reduce(f32)");
        return SRVP_OK;
    }
    auto kern = k.da_mode == 0 ? (lr ? bn_bwd_reduce_kernel<bf16_t, 0, ACT_LRELU> : bn_bwd_reduce_kernel<bf16_t, 0, -1>)
              : k.da_mode == 1 ? (lr ? bn_bwd_reduce_kernel<bf16_t, 1, ACT_LRELU> : bn_bwd_reduce_kernel<bf16_t, 1, -1>)
              : k.da_mode == 3 ? (lr ? bn_bwd_reduce_kernel<bf16_t, 3, ACT_LRELU> : bn_bwd_reduce_kernel<bf16_t, 3, -1>)
                               : (lr ? bn_bwd_reduce_kernel<bf16_t, 2, ACT_LRELU> : bn_bwd_reduce_kernel<bf16_t, 2, -1>);
    hipLaunchKernelGGL(kern, g, dim3(256), 0, (hipStream_t)stream, k, red);
    SRVP_CHECK_LAUNCH("srvp_bn_bwd_reduce");
    return SRVP_OK;
}

extern "C" int srvp_bn_bwd_finalize(const double* red, double count, const float* scale, const float* mean,
                                    const float* invstd, float* dgamma, float* dbeta, float* coef, int C, int C_real,
                                    int has_bn, float param_grad_scale, void* stream) {
    SRVP_REQUIRE(coef && C > 0 && (!has_bn || (red && scale && mean && invstd)), "srvp_bn_bwd_finalize: bad args");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, red, count, scale,
                       mean, invstd, dgamma, dbeta, coef, C, C_real, has_bn, param_grad_scale);
    SRVP_CHECK_LAUNCH("srvp_bn_bwd_finalize");
    return SRVP_OK;
}

namespace {
int bn_bwd_apply_launch(const srvp_bnbwd_desc* d, const float* coef, void* draw, int dst_border, void* stream, const BnBwdFin fin);
}
extern "C" int srvp_bn_bwd_apply(const srvp_bnbwd_desc* d, const float* coef, void* draw, int dst_border, void* stream) {
    SRVP_REQUIRE(coef, "srvp_bn_bwd_apply: null pointer");
    return bn_bwd_apply_launch(d, coef, draw, dst_border, stream, BnBwdFin{});
}
extern "C" int srvp_bn_bwd_finalize_apply(const srvp_bnbwd_desc* d, const double* red, double count, float* dgamma, float* dbeta,
                                          float* coef, int C_real, float param_grad_scale, void* draw, int dst_border, void* stream) {
    SRVP_REQUIRE(d && red && count > 0 && d->mean && d->invstd && d->C <= BN_MAX_C, "srvp_bn_bwd_finalize_apply: bad args");
    return bn_bwd_apply_launch(d, coef, draw, dst_border, stream, BnBwdFin{red, count, dgamma, dbeta, coef, C_real, param_grad_scale});
}
namespace {
int bn_bwd_apply_launch(const srvp_bnbwd_desc* d, const float* coef, void* draw, int dst_border, void* stream, const BnBwdFin fin) {
    BnBwdK k;
    int rc = fill_k(d, k);
    if (rc) return rc;
    SRVP_REQUIRE(draw, "srvp_bn_bwd_apply: null pointer");
    const int CG = k.C / 8, PPB = 256 / CG;
    long long P = (long long)k.N * k.H * k.W;
    SRVP_REQUIRE(P < (1ll << 31), "srvp_bn_bwd_apply: too many pixels");
    if (k.da_mode == 2) P /= 4;
    if (k.tsum) P /= k.tsum_T;
    const dim3 g(grid_for(P, PPB * ((k.da_mode == 2 || k.tsum) ? 1 : 4)));
    const bool lr = k.act_kind == ACT_LRELU;
    if (d->elem_f32) {
        auto kern = k.da_mode == 0 ? (lr ? bn_bwd_apply_kernel<float, 0, ACT_LRELU> : bn_bwd_apply_kernel<float, 0, -1>)
                  : k.da_mode == 1 ? (lr ? bn_bwd_apply_kernel<float, 1, ACT_LRELU> : bn_bwd_apply_kernel<float, 1, -1>)
                                   : (lr ? bn_bwd_apply_kernel<float, 2, ACT_LRELU> : bn_bwd_apply_kernel<float, 2, -1>);
        hipLaunchKernelGGL(kern, g, dim3(256), 0, (hipStream_t)stream, k, coef, (float*)draw, dst_border, fin);
        SRVP_CHECK_LAUNCH("srvp_bn_bwd_apply(f32)");
        return SRVP_OK;
    }
    auto kern = k.da_mode == 0 ? (lr ? bn_bwd_apply_kernel<bf16_t, 0, ACT_LRELU> : bn_bwd_apply_kernel<bf16_t, 0, -1>)
              : k.da_mode == 1 ? (lr ? bn_bwd_apply_kernel<bf16_t, 1, ACT_LRELU> : bn_bwd_apply_kernel<bf16_t, 1, -1>)
                               : (lr ? bn_bwd_apply_kernel<bf16_t, 2, ACT_LRELU> : bn_bwd_apply_kernel<bf16_t, 2, -1>);
    hipLaunchKernelGGL(kern, g, dim3(256), 0, (hipStream_t)stream, k, coef, (bf16_t*)draw, dst_border, fin);
    SRVP_CHECK_LAUNCH("srvp_bn_bwd_apply");
    return SRVP_OK;
}
}  // namespace
