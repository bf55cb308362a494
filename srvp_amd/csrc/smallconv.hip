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
template <class E>
__global__ __launch_bounds__(256) void conv_in_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          E* __restrict__ raw, double* stats, int N, int Cin, int H,
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
            El<E>::st8(raw + (size_t)pix * Cout + cg * 8, acc);
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
template <class E, int K>
__global__ __launch_bounds__(256) void conv_in_wgrad_kernel(const float* __restrict__ x, const E* __restrict__ draw,
                                                           float* dw, int N, int Cin, int H, int W, int Cout, int Cout_real,
                                                           int s, int p, int OH, int OW, long long pix_per_stream, float* det_slab) {
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
                El<E>::ld8(draw + (((size_t)n * (OH + 2) + oy + 1) * (OW + 2) + ox + 1) * Cout + cg * 8, g);
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
                for (int t = 0; t < KK; ++t) {
                    if (det_slab) det_slab[(size_t)blockIdx.x * Cout_real * Cin * KK + ((size_t)co * Cin + ci) * KK + t] = acc[e][t];    // (deterministic mode)
                    else atomicAdd(dw + ((size_t)co * Cin + ci) * KK + t, acc[e][t]);
                }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// First-layer forward / weight gradient on the matrix cores in EXACT fp32 (v_mfma_f32_32x32x2_f32 runs at the fp32
// vector rate and is bit-equivalent to an fmaf chain, so the "image-side layers are fp32" statement stands): the
// VALU kernels above sit at ~10 % of the HBM roofline of this layer (1.2 GB of bf16 activations per pass) because
// they spend their time on LDS operand traffic and bounds predicates.  Frames are 64x64 (module/conv.py:129-154
// hard-codes the 64x64 encoders); kernel 3x3 s1 p1 (VGG) or 4x4 s2 p1 (DCGAN); Cin in {1, 3}; Cout <= 64.
// A workgroup walks `tiles_per_wg` consecutive 128-pixel output tiles (whole output rows of one image): the input
// patch incl. halo is staged zero-padded in LDS, K = Cin*k*k (padded to even) is walked two taps per MFMA.
// ---------------------------------------------------------------------------------------------------------
typedef float f32x16v __attribute__((ext_vector_type(16)));

template <int KS, int S, int CIN> struct InGeom {
    static constexpr int H = 64, W = 64;
    static constexpr int OH = H / S, OW = W / S;
    static constexpr int TR = 128 / OW;                  // output rows per 128-pixel tile
    static constexpr int TPI = OH / TR;                  // tiles per image
    static constexpr int PH = (TR - 1) * S + KS;         // patch rows
    static constexpr int PWp = W + 2;                    // patch columns (x = -1 .. 64)
    static constexpr int K = CIN * KS * KS;
    static constexpr int STEPS = (K + 1) / 2;
    __host__ __device__ static constexpr int off(int kidx) {           // patch offset of K index (ci, kh, kw)
        return kidx >= K ? 0 : ((kidx / (KS * KS)) * PH + (kidx / KS) % KS) * PWp + kidx % KS;
    }
};

// the same patch in two halves, so that the global loads of tile i+1 are in flight while tile i is computed: fetch into registers
// (element e of this thread is patch element threadIdx.x + 256 e) ...
template <int KS, int S, int CIN, int NP>
__device__ __forceinline__ void in_fetch_patch(const float* __restrict__ x, float (&pv)[NP], int n, int tile_in_img) {
    typedef InGeom<KS, S, CIN> Gm;
    const int iy0 = tile_in_img * Gm::TR * S - 1;
#pragma unroll
    for (int e = 0; e < NP; ++e) {
        const int i = threadIdx.x + e * 256;
        const int c = i % Gm::PWp, r = (i / Gm::PWp) % Gm::PH, ci = i / (Gm::PWp * Gm::PH);
        const int iy = iy0 + r, ix = c - 1;
        const bool ok = i < CIN * Gm::PH * Gm::PWp && iy >= 0 && iy < Gm::H && ix >= 0 && ix < Gm::W;
        pv[e] = ok ? x[(((size_t)n * CIN + ci) * Gm::H + iy) * Gm::W + ix] : 0.f;
    }
}
// ... and store into the LDS patch
template <int KS, int S, int CIN, int NP>
__device__ __forceinline__ void in_commit_patch(const float (&pv)[NP], float* xs) {
    typedef InGeom<KS, S, CIN> Gm;
#pragma unroll
    for (int e = 0; e < NP; ++e) {
        const int i = threadIdx.x + e * 256;
        if (i < CIN * Gm::PH * Gm::PWp) xs[i] = pv[e];
    }
}

template <int KS, int S, int CIN>
__device__ __forceinline__ void in_load_patch(const float* __restrict__ x, float* xs, int n, int tile_in_img) {
    typedef InGeom<KS, S, CIN> Gm;
    const int iy0 = tile_in_img * Gm::TR * S - 1;
    for (int i = threadIdx.x; i < CIN * Gm::PH * Gm::PWp; i += 256) {
        const int c = i % Gm::PWp, r = (i / Gm::PWp) % Gm::PH, ci = i / (Gm::PWp * Gm::PH);
        const int iy = iy0 + r, ix = c - 1;
        const bool ok = iy >= 0 && iy < Gm::H && ix >= 0 && ix < Gm::W;
        xs[i] = ok ? x[(((size_t)n * CIN + ci) * Gm::H + iy) * Gm::W + ix] : 0.f;
    }
}

// (plain forward: capped at 128 registers = 4 waves per SIMD -- 94 VGPRs, no spills -- 0.60 -> 0.55 ms at 2304 frames; the variant
// with the fused BatchNorm-backward sums and the weight-gradient kernels spill under that cap and measured slower: left alone)
template <int KS, int S, int CIN, int NTL, bool BNR = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BNR ? 2 : 4, BNR ? 8 : 4))) void conv_in_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               bf16_t* __restrict__ raw, double* stats, int N, int Cout,
                                                               int Cout_real, int tiles_per_wg, const bf16_t* __restrict__ bnr_raw,
                                                               const float* __restrict__ bnr_coef, double* bnr_red) {
    typedef InGeom<KS, S, CIN> Gm;
    constexpr int LDC = 64 + 8;
    __shared__ float wsh[Gm::STEPS * 2][64];             // B operand [k][cout], zero for padded k / cout
    __shared__ float xs[CIN * Gm::PH * Gm::PWp];
    __shared__ __attribute__((aligned(16))) bf16_t Cs[128 * LDC];
    __shared__ float red[4][64][2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lcol = lane & 31, lhalf = lane >> 5;
    for (int i = tid; i < Gm::STEPS * 2 * 64; i += 256) {
        const int co = i & 63, k = i >> 6;
        wsh[k][co] = (k < Gm::K && co < Cout_real) ? w[(size_t)co * Gm::K + k] : 0.f;
    }
    constexpr int ntl = NTL;                              // column tiles (compile time: a runtime test around the MFMAs makes
                                                          // hipcc shuttle the accumulators through AGPR moves every step)
    const int m = wid * 32 + lcol;                        // this lane's pixel (A operand row) inside the tile
    const int abase = ((m / Gm::OW) * S) * Gm::PWp + (m % Gm::OW) * S;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const long long ntiles = (long long)N * Gm::TPI;
    const long long t0 = (long long)blockIdx.x * tiles_per_wg;
    constexpr int NP = (CIN * Gm::PH * Gm::PWp + 255) / 256;
    float pv[NP];
    // (BNR is a template parameter: as a run-time branch its 48 registers cost the forward instantiation a wave of occupancy)
    float bsc[8], bsh[8], bmu[8], bis[8], b1[8], b2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bsc[e] = bsh[e] = bmu[e] = bis[e] = b1[e] = b2[e] = 0.f; }
    if constexpr (BNR) {
        const int c0 = (tid % (Cout / 8)) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bsc[e] = bnr_coef[c0 + e]; bsh[e] = bnr_coef[Cout + c0 + e]; bmu[e] = bnr_coef[2 * Cout + c0 + e]; bis[e] = bnr_coef[3 * Cout + c0 + e];
        }
    }
    if (t0 < ntiles) in_fetch_patch<KS, S, CIN, NP>(x, pv, (int)(t0 / Gm::TPI), (int)(t0 % Gm::TPI));
    for (int it = 0; it < tiles_per_wg; ++it) {
        const long long tile = t0 + it;
        if (tile >= ntiles) break;
        __syncthreads();                                  // previous tile's patch / staging are no longer read
        in_commit_patch<KS, S, CIN, NP>(pv, xs);
        __syncthreads();
        // next tile's patch: global loads in flight under this tile's MFMAs and epilogue
        if (it + 1 < tiles_per_wg && tile + 1 < ntiles)
            in_fetch_patch<KS, S, CIN, NP>(x, pv, (int)((tile + 1) / Gm::TPI), (int)((tile + 1) % Gm::TPI));
        f32x16v acc[NTL];
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // all operands of the tile first (3 * STEPS LDS reads in flight), then the MFMA chain without waits in between
        float av[Gm::STEPS], bv[NTL][Gm::STEPS];
#pragma unroll
        for (int st = 0; st < Gm::STEPS; ++st) {
            av[st] = xs[abase + (lhalf ? Gm::off(2 * st + 1) : Gm::off(2 * st))];
#pragma unroll
            for (int j = 0; j < NTL; ++j) bv[j][st] = wsh[2 * st + lhalf][j * 32 + lcol];
        }
#pragma unroll
        for (int st = 0; st < Gm::STEPS; ++st)
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], bv[j][st], acc[j], 0, 0, 0);
        // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[j][r];
                s1[j] += v; s2[j] += v * v;
                Cs[(wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * LDC + j * 32 + lcol] = f2bf(v);
            }
        }
        __syncthreads();
        // the 128 pixels of a tile are contiguous in the NHWC output: 128 * Cout bf16
        bf16_t* dst = raw + (size_t)tile * 128 * Cout;
        const int cch = Cout / 8;
        if constexpr (BNR) {
            // this launch is the DATA GRADIENT of the image-side output layer (the gradient frames are its "image"): its output is dA of
            // the producer block, whose BatchNorm-backward sums are accumulated here (as srvp_conv_desc.bnr_* does in the MFMA
            // convolutions) -- 256 % cch == 0, so a thread keeps its channel chunk over the tiles and its sums stay in registers
            const bf16_t* rsrc = bnr_raw + (size_t)tile * 128 * Cout;
            for (int q = tid; q < 128 * cch; q += 256) {
                const int row = q / cch, ch = q % cch;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(Cs + row * LDC + ch * 8);
                const u32x4_t rw = *reinterpret_cast<const u32x4_t*>(rsrc + (size_t)row * Cout + ch * 8);
                *reinterpret_cast<u32x4_t*>(dst + (size_t)row * Cout + ch * 8) = v;
                float da[8], rv[8];
                unpack8(v, da);
                unpack8(rw, rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gg = da[e] * ((rv[e] * bsc[e] + bsh[e]) > 0.f ? 1.f : LRELU_SLOPE);
                    b1[e] += gg; b2[e] += gg * (rv[e] - bmu[e]) * bis[e];
                }
            }
            continue;
        }
        for (int q = tid; q < 128 * cch; q += 256) {
            const int row = q / cch, ch = q % cch;
            *reinterpret_cast<u32x4_t*>(dst + (size_t)row * Cout + ch * 8) = *reinterpret_cast<const u32x4_t*>(Cs + row * LDC + ch * 8);
        }
    }
    if constexpr (BNR) {
        __syncthreads();
        float* Ps = reinterpret_cast<float*>(Cs);                 // [256 / cch][Cout][2] partial sums (<= 128 * LDC * 2 bytes)
        const int cch = Cout / 8, ch = tid % cch, grp = tid / cch;
#pragma unroll
        for (int e = 0; e < 8; ++e) { Ps[(grp * Cout + ch * 8 + e) * 2] = b1[e]; Ps[(grp * Cout + ch * 8 + e) * 2 + 1] = b2[e]; }
        __syncthreads();
        if (tid < Cout) {
            double t1 = 0., t2 = 0.;
            for (int r = 0; r < 256 / cch; ++r) { t1 += Ps[(r * Cout + tid) * 2]; t2 += Ps[(r * Cout + tid) * 2 + 1]; }
            atomicAdd(bnr_red + tid, t1);
            atomicAdd(bnr_red + Cout + tid, t2);
        }
        return;
    }
    if (!stats) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float a = s1[j] + __shfl_xor(s1[j], 32), b = s2[j] + __shfl_xor(s2[j], 32);
        if (lhalf == 0) { red[wid][j * 32 + lcol][0] = a; red[wid][j * 32 + lcol][1] = b; }
    }
    __syncthreads();
    if (tid < Cout) {
        double a = 0., b = 0.;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) { a += red[wv][tid][0]; b += red[wv][tid][1]; }
        atomicAdd(stats + tid, a);
        atomicAdd(stats + Cout + tid, b);
    }
}

// dw[co][k] += sum_pix draw[pix][co] * patch[pix][k]:  A = draw^T (32 couts x 2 pixels), B = patch (2 pixels x 32 K
// indices; K <= 32 per column tile, 48 needs two), accumulated over all tiles of the workgroup, then reduced over
// the four waves (which split the 128 pixels of a tile) through LDS and added to dw with one atomic per weight.
template <int KS, int S, int CIN, int NTL>
__global__ __launch_bounds__(256) void conv_in_wgrad_mfma_kernel(const float* __restrict__ x, const bf16_t* __restrict__ draw,
                                                                 float* dw, int N, int Cout, int Cout_real, int tiles_per_wg) {
    typedef InGeom<KS, S, CIN> Gm;
    constexpr int KT = (Gm::K + 31) / 32;                 // column tiles over the K indices (1 or 2)
    constexpr int XS_BYTES = ((CIN * Gm::PH * Gm::PWp * 4 + 15) / 16) * 16, GS_BYTES = 128 * 64 * 2;
    constexpr int PART_BYTES = 4 * 64 * (KT * 32 + 1) * 4;
    constexpr int SM = (XS_BYTES + GS_BYTES) > PART_BYTES ? (XS_BYTES + GS_BYTES) : PART_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char sm[SM];
    float* xs = reinterpret_cast<float*>(sm);                                        // input patch
    bf16_t* gs = reinterpret_cast<bf16_t*>(sm + XS_BYTES);                           // draw tile [128 px][Cout]
    float (*part)[64][KT * 32 + 1] = reinterpret_cast<float (*)[64][KT * 32 + 1]>(sm);   // wave partials (after the loop)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lcol = lane & 31, lhalf = lane >> 5;
    constexpr int ntl = NTL;
    // B operand offsets of this lane's K indices
    int boff[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int kidx = kt * 32 + lcol;
        boff[kt] = kidx >= Gm::K ? -1 : ((kidx / (KS * KS)) * Gm::PH + (kidx / KS) % KS) * Gm::PWp + kidx % KS;
    }
    f32x16v acc[NTL][KT];
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][kt][r] = 0.f;
    const long long ntiles = (long long)N * Gm::TPI;
    const long long t0 = (long long)blockIdx.x * tiles_per_wg;
    const int cch = Cout / 8;
    for (int it = 0; it < tiles_per_wg; ++it) {
        const long long tile = t0 + it;
        if (tile >= ntiles) break;
        const int n = (int)(tile / Gm::TPI), tin = (int)(tile % Gm::TPI);
        __syncthreads();
        in_load_patch<KS, S, CIN>(x, xs, n, tin);
        // draw tile: bordered tensor [N][OH+2][OW+2][Cout], rows tin*TR .. +TR-1 of image n
        for (int q = tid; q < 128 * cch; q += 256) {
            const int row = q / cch, ch = q % cch;
            const int oy = tin * Gm::TR + row / Gm::OW, ox = row % Gm::OW;
            *reinterpret_cast<u32x4_t*>(gs + row * Cout + ch * 8) =
                *reinterpret_cast<const u32x4_t*>(draw + (((size_t)n * (Gm::OH + 2) + oy + 1) * (Gm::OW + 2) + ox + 1) * Cout + ch * 8);
        }
        __syncthreads();
#pragma unroll 4
        for (int st = 0; st < 16; ++st) {                 // this wave's 32 pixels, two per MFMA
            const int mpix = wid * 32 + st * 2 + lhalf;
            const int pb = ((mpix / Gm::OW) * S) * Gm::PWp + (mpix % Gm::OW) * S;
            float bv[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) bv[kt] = boff[kt] >= 0 ? xs[pb + boff[kt]] : 0.f;
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const float av = bf2f(gs[mpix * Cout + j * 32 + lcol]);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) acc[j][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[kt], acc[j][kt], 0, 0, 0);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                part[wid][j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf][kt * 32 + lcol] = acc[j][kt][r];
    __syncthreads();
    for (int q = tid; q < Cout_real * Gm::K; q += 256) {
        const int co = q / Gm::K, k = q % Gm::K;
        atomicAdd(dw + (size_t)co * Gm::K + k, part[0][co][k] + part[1][co][k] + part[2][co][k] + part[3][co][k]);
    }
}

// The same weight gradient with the fp32 operand SPLIT INTO THREE bf16 TERMS (x = x1 + x2 + x3, 3 x 8 significant bits = the
// fp32 mantissa) against the bf16 output gradient on the bf16 matrix cores: every product bf16 x bf16 is exact in fp32 and the
// accumulation is fp32, so the result has the accuracy of the fp32 kernel above, but 16 pixels per MFMA at twice the issue rate
// instead of 2 (v_mfma_f32_32x32x2_f32 made this HBM-sized layer MFMA-bound: 0.60 ms against 1.2 GB of traffic).
// A = draw^T: the draw tile is staged TRANSPOSED in LDS, [cout][pixel], 16-byte pixel chunks XOR-swizzled by
// (cout & 7) ^ (cout >> 3 & 7) -- conflict-free for the transposing 2-byte stores and for the ds_read_b128 fragment reads.
// FUSE (srvp_conv_in_wgrad_bn): the output gradient of the first block is consumed by NOTHING but this weight gradient (there is no data
// gradient wrt the frames), so it is never written: the tile is formed on the way into LDS from the block's dA and raw output --
// draw = k1 g + k2 + k3 raw, g = dA lrelu'(scale raw + shift), rounded to bf16 as srvp_bn_bwd_apply stores it -- with the coefficients
// derived per workgroup from the BatchNorm-backward sums exactly as srvp_bn_bwd_finalize does (workgroup 0 adds dgamma / dbeta).  One
// read of (dA, raw) instead of apply's read of both + write of draw + this kernel's read of draw: 4.8 -> 2.4 GB at 2304 frames, at the
// very end of the step where nothing else is left to overlap it.
struct InWgradBn {
    const bf16_t* da; const bf16_t* raw;                   // [N][OH][OW][Cout], unbordered
    const double* red; double count;                       // BatchNorm-backward sums [2][Cout] (all-reduced), element count
    const float* coef4;                                    // [4][Cout] scale, shift, mean, inverse std
    float* dgamma; float* dbeta; float* bcoef;             // parameter gradients (+=), [3][Cout] k1 k2 k3 (may be null)
    int C_real; float pscale;
};
template <int KS, int S, int CIN, int NTL, bool FUSE = false>
__global__ __launch_bounds__(256) void conv_in_wgrad_mfma3_kernel(const float* __restrict__ x, const bf16_t* __restrict__ draw,
                                                                  float* dw, int N, int Cout, int Cout_real, int tiles_per_wg,
                                                                  const InWgradBn fb = InWgradBn{}) {
    typedef InGeom<KS, S, CIN> Gm;
    static_assert(Gm::OW % 8 == 0, "eight consecutive pixels of a K group lie in one output row");
    constexpr int KT = (Gm::K + 31) / 32;
    constexpr int XS_BYTES = ((CIN * Gm::PH * Gm::PWp * 4 + 15) / 16) * 16, GS_BYTES = 64 * 128 * 2;
    constexpr int PART_BYTES = 4 * 64 * (KT * 32 + 1) * 4;
    constexpr int SM = (XS_BYTES + GS_BYTES) > PART_BYTES ? (XS_BYTES + GS_BYTES) : PART_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char sm[SM];
    float* xs = reinterpret_cast<float*>(sm);
    bf16_t* gsT = reinterpret_cast<bf16_t*>(sm + XS_BYTES);                          // draw tile transposed [cout][128 px]
    float (*part)[64][KT * 32 + 1] = reinterpret_cast<float (*)[64][KT * 32 + 1]>(sm);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lcol = lane & 31, lhalf = lane >> 5;
    int boff[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int kidx = kt * 32 + lcol;
        boff[kt] = kidx >= Gm::K ? -1 : ((kidx / (KS * KS)) * Gm::PH + (kidx / KS) % KS) * Gm::PWp + kidx % KS;
    }
    f32x16v acc[NTL][KT];
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][kt][r] = 0.f;
    const long long ntiles = (long long)N * Gm::TPI;
    const long long t0 = (long long)blockIdx.x * tiles_per_wg;
    const int cch = Cout / 8;
    // both operands of tile i+1 are fetched into registers while tile i is computed
    constexpr int NP = (CIN * Gm::PH * Gm::PWp + 255) / 256, ND = 128 * 8 / 256;      // ND: 16-byte draw pieces per thread (Cout <= 64)
    float pv[NP];
    u32x4_t dv[ND], rv[FUSE ? ND : 1];
    // FUSE: this thread's eight channels (256 % cch == 0: the chunk index is the same for all its pieces) and their coefficients
    float fsc[8], fsh[8], fk1[8], fk2[8], fk3[8];
    if constexpr (FUSE) {
        const int C = Cout, c0 = (tid % cch) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            fsc[e] = fb.coef4[c]; fsh[e] = fb.coef4[C + c];
            fk1[e] = fk2[e] = fk3[e] = 0.f;
            if (c < fb.C_real) {
                // (the fp64 expressions of bn_bwd_finalize_kernel)
                const double sg = fb.red[c], sgx = fb.red[C + c];
                const double mg = sg / fb.count, mgx = sgx / fb.count;
                const double k1 = fb.coef4[c];
                const double k3 = -k1 * mgx * fb.coef4[3 * C + c];
                const double k2 = -k1 * mg - k3 * fb.coef4[2 * C + c];
                fk1[e] = (float)k1; fk2[e] = (float)k2; fk3[e] = (float)k3;
                if (blockIdx.x == 0 && tid < cch) {
                    if (fb.dgamma) fb.dgamma[c] += (float)(sgx * fb.pscale);
                    if (fb.dbeta) fb.dbeta[c] += (float)(sg * fb.pscale);
                }
            }
            if (blockIdx.x == 0 && tid < cch && fb.bcoef) { fb.bcoef[c] = fk1[e]; fb.bcoef[C + c] = fk2[e]; fb.bcoef[2 * C + c] = fk3[e]; }
        }
    }
    auto fetch_draw = [&](long long tile) {
        const int n = (int)(tile / Gm::TPI), tin = (int)(tile % Gm::TPI);
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int q = tid + u * 256;
            if (q >= 128 * cch) continue;
            const int row = q / cch, ch = q % cch;
            const int oy = tin * Gm::TR + row / Gm::OW, ox = row % Gm::OW;
            if constexpr (FUSE) {
                const size_t o = (((size_t)n * Gm::OH + oy) * Gm::OW + ox) * Cout + ch * 8;
                dv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(fb.da + o));
                rv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(fb.raw + o));
            } else {
                dv[u] = *reinterpret_cast<const u32x4_t*>(draw + (((size_t)n * (Gm::OH + 2) + oy + 1) * (Gm::OW + 2) + ox + 1) * Cout + ch * 8);
            }
        }
    };
    if (t0 < ntiles) { in_fetch_patch<KS, S, CIN, NP>(x, pv, (int)(t0 / Gm::TPI), (int)(t0 % Gm::TPI)); fetch_draw(t0); }
    for (int it = 0; it < tiles_per_wg; ++it) {
        const long long tile = t0 + it;
        if (tile >= ntiles) break;
        __syncthreads();
        in_commit_patch<KS, S, CIN, NP>(pv, xs);
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int q = tid + u * 256;
            if (q >= 128 * cch) continue;
            const int row = q / cch, ch = q % cch;
            if constexpr (FUSE) {
                float d8[8], r8[8];
                unpack8(dv[u], d8);
                unpack8(rv[u], r8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float g = d8[e] * (r8[e] * fsc[e] + fsh[e] > 0.f ? 1.f : LRELU_SLOPE);
                    d8[e] = fk1[e] * g + fk2[e] + fk3[e] * r8[e];
                }
                dv[u].x = pack2bf(d8[0], d8[1]); dv[u].y = pack2bf(d8[2], d8[3]); dv[u].z = pack2bf(d8[4], d8[5]); dv[u].w = pack2bf(d8[6], d8[7]);
            }
            const unsigned short* h = reinterpret_cast<const unsigned short*>(&dv[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int co = ch * 8 + e;
                gsT[co * 128 + ((((row >> 3) ^ ((co & 7) ^ ((co >> 3) & 7))) & 15) << 3) + (row & 7)] = h[e];
            }
        }
        __syncthreads();
        if (it + 1 < tiles_per_wg && tile + 1 < ntiles) {
            in_fetch_patch<KS, S, CIN, NP>(x, pv, (int)((tile + 1) / Gm::TPI), (int)((tile + 1) % Gm::TPI));
            fetch_draw(tile + 1);
        }
#pragma unroll
        for (int g16 = 0; g16 < 2; ++g16) {               // this wave's 32 pixels, sixteen per MFMA (this half-wave: eight of them)
            const int p0 = wid * 32 + g16 * 16 + lhalf * 8;
            const int pb = ((p0 / Gm::OW) * S) * Gm::PWp + (p0 % Gm::OW) * S;
            bf16x8_t bt[3][KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = boff[kt] >= 0 ? xs[pb + boff[kt] + i * S] : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const __bf16 h1 = (__bf16)v[i];
                    const float r1 = v[i] - (float)h1;
                    const __bf16 h2 = (__bf16)r1;
                    const __bf16 h3 = (__bf16)(r1 - (float)h2);
                    bt[0][kt][i] = h1; bt[1][kt][i] = h2; bt[2][kt][i] = h3;
                }
            }
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const int co = j * 32 + lcol;
                const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(gsT + co * 128 + ((((p0 >> 3) ^ ((co & 7) ^ ((co >> 3) & 7))) & 15) << 3));
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[j][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bt[t][kt], acc[j][kt], 0, 0, 0);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                part[wid][j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf][kt * 32 + lcol] = acc[j][kt][r];
    __syncthreads();
    for (int q = tid; q < Cout_real * Gm::K; q += 256) {
        const int co = q / Gm::K, k = q % Gm::K;
        atomicAdd(dw + (size_t)co * Gm::K + k, part[0][co][k] + part[1][co][k] + part[2][co][k] + part[3][co][k]);
    }
}

template <int KS, int S, int CIN>
static void launch_in_fwd(const float* x, const float* w, bf16_t* raw, double* stats, int N, int Cout, int Cout_real, hipStream_t st,
                          const bf16_t* bnr_raw = nullptr, const float* bnr_coef = nullptr, double* bnr_red = nullptr) {
    const long long ntiles = (long long)N * InGeom<KS, S, CIN>::TPI;
    static int tpw_env = -1;
    if (tpw_env < 0) { const char* e = getenv("SRVP_IN_TPW_F"); tpw_env = e ? atoi(e) : 0; }
    // tiles per (persistent) workgroup: ~2048 workgroups per launch, at most 8 tiles each (round 5: the step function 2 / 8 at 32768 tiles left
    // the 15360-tile layers of config 2 and the 9216-tile layer of a 24-sequence step with 2 tiles and one statistics epilogue per 2 tiles)
    const int tpw = tpw_env > 0 ? tpw_env : (int)(ntiles / 2048 < 1 ? 1 : (ntiles / 2048 > 8 ? 8 : ntiles / 2048));
    const dim3 grid((unsigned)((ntiles + tpw - 1) / tpw));
    if (bnr_red) {
        if (Cout == 64) hipLaunchKernelGGL((conv_in_fwd_mfma_kernel<KS, S, CIN, 2, true>), grid, dim3(256), 0, st, x, w, raw, stats, N, Cout, Cout_real, tpw, bnr_raw, bnr_coef, bnr_red);
        else hipLaunchKernelGGL((conv_in_fwd_mfma_kernel<KS, S, CIN, 1, true>), grid, dim3(256), 0, st, x, w, raw, stats, N, Cout, Cout_real, tpw, bnr_raw, bnr_coef, bnr_red);
    } else if (Cout == 64)
        hipLaunchKernelGGL((conv_in_fwd_mfma_kernel<KS, S, CIN, 2>), grid, dim3(256), 0, st, x, w, raw, stats, N, Cout, Cout_real, tpw, bnr_raw, bnr_coef, bnr_red);
    else
        hipLaunchKernelGGL((conv_in_fwd_mfma_kernel<KS, S, CIN, 1>), grid, dim3(256), 0, st, x, w, raw, stats, N, Cout, Cout_real, tpw, bnr_raw, bnr_coef, bnr_red);
}
template <int KS, int S, int CIN>
static void launch_in_wgrad(const float* x, const bf16_t* draw, float* dw, int N, int Cout, int Cout_real, hipStream_t st) {
    const long long ntiles = (long long)N * InGeom<KS, S, CIN>::TPI;
    static int tpw_env = -1;
    if (tpw_env < 0) { const char* e = getenv("SRVP_IN_TPW_W"); tpw_env = e ? atoi(e) : 0; }
    // tiles per workgroup: every workgroup ends with an LDS reduction + Cout x K fp32 atomics, so few, long workgroups: ~1024 per launch, at
    // most 16 tiles each (round 5: was 2 below 32768 tiles -- 7680 epilogues for the 15360 tiles of config 2's image-side layers)
    const int tpw = tpw_env > 0 ? tpw_env : (int)(ntiles / 1024 < 1 ? 1 : (ntiles / 1024 > 16 ? 16 : ntiles / 1024));
    static int split3 = -1;     // A/B switch: 1 = three-term bf16 split on the bf16 matrix cores, 0 = fp32 MFMA
    if (split3 < 0) { const char* e = getenv("SRVP_IN_WGRAD_SPLIT3"); split3 = e ? atoi(e) : 1; }
    const dim3 grid((unsigned)((ntiles + tpw - 1) / tpw));
    if (split3) {
        if (Cout == 64) hipLaunchKernelGGL((conv_in_wgrad_mfma3_kernel<KS, S, CIN, 2>), grid, dim3(256), 0, st, x, draw, dw, N, Cout, Cout_real, tpw);
        else hipLaunchKernelGGL((conv_in_wgrad_mfma3_kernel<KS, S, CIN, 1>), grid, dim3(256), 0, st, x, draw, dw, N, Cout, Cout_real, tpw);
    } else if (Cout == 64)
        hipLaunchKernelGGL((conv_in_wgrad_mfma_kernel<KS, S, CIN, 2>), grid, dim3(256), 0, st, x, draw, dw, N, Cout, Cout_real, tpw);
    else
        hipLaunchKernelGGL((conv_in_wgrad_mfma_kernel<KS, S, CIN, 1>), grid, dim3(256), 0, st, x, draw, dw, N, Cout, Cout_real, tpw);
}
template <int CIN>
static void launch_in_wgrad_bn(const float* x, float* dw, int N, int Cout, int Cout_real, const InWgradBn& fb, hipStream_t st) {
    const long long ntiles = (long long)N * InGeom<3, 1, CIN>::TPI;
    const int tpw = (int)(ntiles / 1024 < 1 ? 1 : (ntiles / 1024 > 16 ? 16 : ntiles / 1024));      // (as launch_in_wgrad)
    const dim3 grid((unsigned)((ntiles + tpw - 1) / tpw));
    hipLaunchKernelGGL((conv_in_wgrad_mfma3_kernel<3, 1, CIN, 2, true>), grid, dim3(256), 0, st, x, (const bf16_t*)nullptr, dw, N, Cout, Cout_real, tpw, fb);
}
static bool in_mfma_ok(int Cin, int H, int W, int Cout, int k, int s, int p) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SRVP_CONV_IN_MFMA"); on = e ? atoi(e) : 1; }
    return on && H == 64 && W == 64 && (Cin == 1 || Cin == 3) && (Cout == 32 || Cout == 64) && p == 1 && ((k == 3 && s == 1) || (k == 4 && s == 2));
}

}  // namespace

// conv_in_stream.hip (3x3 stride 1, 64 output channels: streaming kernel)
int srvp_conv_in_stream_launch(const float* x, const float* w, bf16_t* raw, double* stats, int N, int Cin, int Cout, int Cout_real,
                               const bf16_t* bnr_raw, const float* bnr_coef, double* bnr_red, hipStream_t st, int* taken);

namespace {
template <class E>
int conv_in_fwd_valu(const float* x, const float* w, void* raw, double* stats, int N, int Cin, int H, int W, int Cout, int Cout_real,
                     int k, int s, int p, void* stream) {
    int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
    size_t sh = (size_t)Cin * k * k * Cout * sizeof(float);
    long long P = (long long)N * OH * OW;
    int PPB = 256 / (Cout / 8);
    long long blocks = (P + (long long)PPB * 16 - 1) / ((long long)PPB * 16);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(conv_in_fwd_kernel<E>, dim3((unsigned)blocks), dim3(256), sh, (hipStream_t)stream, x, w, (E*)raw, stats,
                       N, Cin, H, W, Cout, Cout_real, k, s, p, OH, OW);
    SRVP_CHECK_LAUNCH("srvp_conv_in_fwd");
    return SRVP_OK;
}
template <class E>
int conv_in_wgrad_valu(const float* x, const void* draw, float* dw, int N, int Cin, int H, int W, int Cout, int Cout_real, int k, int s,
                       int p, void* stream) {
    int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
    const int G = (Cout / 8) * Cin;
    SRVP_REQUIRE(G <= 256, "srvp_conv_in_wgrad: Cout*Cin too large");
    const int SPB = 256 / G;
    const long long P = (long long)N * OH * OW;
    long long blocks = 1024;
    long long pps = (P + blocks * SPB - 1) / (blocks * SPB);
    if (pps < 16) { pps = 16; blocks = (P + pps * SPB - 1) / (pps * SPB); }
    const size_t sh = (size_t)(SPB > 1 ? SPB - 1 : 1) * G * 8 * k * k * sizeof(float);
    SRVP_REQUIRE(sh <= 160 * 1024, "srvp_conv_in_wgrad: LDS budget");
    float* slab = nullptr;
    const int n_dw = Cout_real * Cin * k * k;
    if (g_srvp_det && El<E>::is_f32) {
        // deterministic mode: at most 256 workgroups, their partial gradients into the workspace, summed in workgroup order afterwards
        SRVP_REQUIRE(g_srvp_det_ws && 256ll * n_dw * 4 <= g_srvp_det_ws_bytes, "srvp_conv_in_wgrad: deterministic workspace too small");
        if (blocks > 256) { blocks = 256; pps = (P + blocks * SPB - 1) / (blocks * SPB); }
        slab = (float*)g_srvp_det_ws;
    }
    if (k == 3)
        hipLaunchKernelGGL((conv_in_wgrad_kernel<E, 3>), dim3((unsigned)blocks), dim3(256), sh, (hipStream_t)stream, x,
                           (const E*)draw, dw, N, Cin, H, W, Cout, Cout_real, s, p, OH, OW, pps, slab);
    else
        hipLaunchKernelGGL((conv_in_wgrad_kernel<E, 4>), dim3((unsigned)blocks), dim3(256), sh, (hipStream_t)stream, x,
                           (const E*)draw, dw, N, Cin, H, W, Cout, Cout_real, s, p, OH, OW, pps, slab);
    if (slab) hipLaunchKernelGGL(det_sum_kernel<float>, dim3((n_dw + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)slab, (int)blocks, n_dw, dw);
    SRVP_CHECK_LAUNCH("srvp_conv_in_wgrad");
    return SRVP_OK;
}
}  // namespace

// precision = 'fp32' parity mode: raw / draw are fp32 NHWC tensors; the direct kernels (an fmaf chain in (ci, kh, kw) order)
extern "C" int srvp_conv_in_fwd_f32(const float* x, const float* w, void* raw, double* stats, int N, int Cin, int H, int W,
                                    int Cout, int Cout_real, int k, int s, int p, void* stream) {
    SRVP_REQUIRE(x && w && raw, "srvp_conv_in_fwd_f32: null pointer");
    SRVP_REQUIRE(Cin >= 1 && Cin <= MAXC && k <= MAXK && Cout % 8 == 0 && Cout / 8 <= 256, "srvp_conv_in_fwd_f32: unsupported shape");
    return conv_in_fwd_valu<float>(x, w, raw, stats, N, Cin, H, W, Cout, Cout_real, k, s, p, stream);
}
extern "C" int srvp_conv_in_wgrad_f32(const float* x, const void* draw, float* dw, int N, int Cin, int H, int W, int Cout,
                                      int Cout_real, int k, int s, int p, void* stream) {
    SRVP_REQUIRE(x && draw && dw, "srvp_conv_in_wgrad_f32: null pointer");
    SRVP_REQUIRE(Cin >= 1 && Cin <= MAXC && (k == 3 || k == 4) && Cout % 8 == 0, "srvp_conv_in_wgrad_f32: unsupported shape");
    return conv_in_wgrad_valu<float>(x, draw, dw, N, Cin, H, W, Cout, Cout_real, k, s, p, stream);
}

extern "C" int srvp_conv_in_fwd(const float* x, const float* w, void* raw, double* stats, int N, int Cin, int H, int W,
                                int Cout, int Cout_real, int k, int s, int p, void* stream) {
    SRVP_REQUIRE(x && w && raw, "srvp_conv_in_fwd: null pointer");
    SRVP_REQUIRE(Cin >= 1 && Cin <= MAXC && k <= MAXK && Cout % 8 == 0 && Cout / 8 <= 256, "srvp_conv_in_fwd: unsupported shape");
    if (in_mfma_ok(Cin, H, W, Cout, k, s, p)) {
        hipStream_t st = (hipStream_t)stream;
        if (k == 3 && s == 1) {
            int taken = 0;
            if (int rc = srvp_conv_in_stream_launch(x, w, (bf16_t*)raw, stats, N, Cin, Cout, Cout_real, nullptr, nullptr, nullptr, st, &taken)) return rc;
            if (taken) { SRVP_CHECK_LAUNCH("srvp_conv_in_fwd(stream)"); return SRVP_OK; }
        }
        if (k == 3 && Cin == 3) launch_in_fwd<3, 1, 3>(x, w, (bf16_t*)raw, stats, N, Cout, Cout_real, st);
        else if (k == 3) launch_in_fwd<3, 1, 1>(x, w, (bf16_t*)raw, stats, N, Cout, Cout_real, st);
        else if (Cin == 3) launch_in_fwd<4, 2, 3>(x, w, (bf16_t*)raw, stats, N, Cout, Cout_real, st);
        else launch_in_fwd<4, 2, 1>(x, w, (bf16_t*)raw, stats, N, Cout, Cout_real, st);
        SRVP_CHECK_LAUNCH("srvp_conv_in_fwd(mfma)");
        return SRVP_OK;
    }
    return conv_in_fwd_valu<bf16_t>(x, w, raw, stats, N, Cin, H, W, Cout, Cout_real, k, s, p, stream);
}

// srvp_conv_in_fwd as the data gradient of the image-side OUTPUT layer (x = gradient frames, w = the ConvTranspose weight read as
// (O, I, k, k), raw = dA of the producer block) with that block's BatchNorm-backward sums fused in (see srvp_conv_desc.bnr_*)
// 1 if srvp_conv_in_fwd_bnr serves this shape (the MFMA image-side kernel: SRVP_CONV_IN_MFMA switch included) -- the host asks before it
// drops the producer's separate srvp_bn_bwd_reduce launch
extern "C" int srvp_conv_in_fwd_bnr_ok(int Cin, int H, int W, int Cout, int k, int s, int p) { return in_mfma_ok(Cin, H, W, Cout, k, s, p) && s == 1 && k == 3 ? 1 : 0; }

extern "C" int srvp_conv_in_fwd_bnr(const float* x, const float* w, void* raw, int N, int Cin, int H, int W, int Cout, int Cout_real, int k, int s,
                                    int p, const void* bnr_raw, const float* bnr_coef, double* bnr_red, void* stream) {
    SRVP_REQUIRE(x && w && raw && bnr_raw && bnr_coef && bnr_red, "srvp_conv_in_fwd_bnr: null pointer");
    SRVP_REQUIRE(in_mfma_ok(Cin, H, W, Cout, k, s, p) && s == 1 && k == 3, "srvp_conv_in_fwd_bnr: shape not served by the MFMA image-side kernel (3x3 stride 1, 64x64, Cout 32 / 64)");
    hipStream_t st = (hipStream_t)stream;
    {
        int taken = 0;
        if (int rc = srvp_conv_in_stream_launch(x, w, (bf16_t*)raw, nullptr, N, Cin, Cout, Cout_real, (const bf16_t*)bnr_raw, bnr_coef, bnr_red, st, &taken)) return rc;
        if (taken) { SRVP_CHECK_LAUNCH("srvp_conv_in_fwd_bnr(stream)"); return SRVP_OK; }
    }
    if (Cin == 3) launch_in_fwd<3, 1, 3>(x, w, (bf16_t*)raw, nullptr, N, Cout, Cout_real, st, (const bf16_t*)bnr_raw, bnr_coef, bnr_red);
    else launch_in_fwd<3, 1, 1>(x, w, (bf16_t*)raw, nullptr, N, Cout, Cout_real, st, (const bf16_t*)bnr_raw, bnr_coef, bnr_red);
    SRVP_CHECK_LAUNCH("srvp_conv_in_fwd_bnr");
    return SRVP_OK;
}

// srvp_bn_bwd_finalize_apply + srvp_conv_in_wgrad of the FIRST block in one launch (see InWgradBn): d describes the block's BatchNorm
// backward as for srvp_bn_bwd_finalize_apply (da_mode 0, bf16, LeakyReLU, unbordered same-size dA, no da2 / tsum); the gradient wrt the
// block's pre-BatchNorm output is never stored.  srvp_conv_in_wgrad_bn_ok: 1 if this shape / descriptor is served.
extern "C" int srvp_conv_in_wgrad_bn_ok(const srvp_bnbwd_desc* d, int Cin, int H, int W, int Cout, int k, int s, int p) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SRVP_IN_WGRAD_BN"); on = e ? atoi(e) : 1; }
    return on && d && in_mfma_ok(Cin, H, W, Cout, k, s, p) && k == 3 && s == 1 && Cout == 64 && d->C == 64 && d->H == 64 && d->W == 64 &&
           d->da_mode == 0 && !d->da_is_f32 && !d->elem_f32 && !d->da2 && !d->tsum && !d->draw_s2d && d->da_border == 0 && d->da_coff == 0 &&
           d->da_cstride == 64 && d->act_kind == ACT_LRELU && d->mean && d->invstd &&
           (((uintptr_t)d->da | (uintptr_t)d->raw) & 15) == 0 ? 1 : 0;
}
extern "C" int srvp_conv_in_wgrad_bn(const float* x, const srvp_bnbwd_desc* d, const double* red, double count, float* dgamma, float* dbeta,
                                     float* coef, int C_real, float param_grad_scale, float* dw, int N, int Cin, int Cout_real, void* stream) {
    SRVP_REQUIRE(x && d && red && dw && count > 0, "srvp_conv_in_wgrad_bn: null pointer / count");
    SRVP_REQUIRE(srvp_conv_in_wgrad_bn_ok(d, Cin, 64, 64, 64, 3, 1, 1) && d->N == N, "srvp_conv_in_wgrad_bn: descriptor / shape not served (ask srvp_conv_in_wgrad_bn_ok)");
    // scale, shift, mean, invstd must be the four rows of ONE [4][64] tensor (they are: srvp_bn_finalize writes them so)
    SRVP_REQUIRE(d->shift == d->scale + 64 && d->mean == d->scale + 128 && d->invstd == d->scale + 192, "srvp_conv_in_wgrad_bn: coefficients must be one [4][64] tensor");
    InWgradBn fb;
    fb.da = (const bf16_t*)d->da; fb.raw = (const bf16_t*)d->raw; fb.red = red; fb.count = count; fb.coef4 = d->scale;
    fb.dgamma = dgamma; fb.dbeta = dbeta; fb.bcoef = coef; fb.C_real = C_real; fb.pscale = param_grad_scale;
    if (Cin == 3) launch_in_wgrad_bn<3>(x, dw, N, 64, Cout_real, fb, (hipStream_t)stream);
    else launch_in_wgrad_bn<1>(x, dw, N, 64, Cout_real, fb, (hipStream_t)stream);
    SRVP_CHECK_LAUNCH("srvp_conv_in_wgrad_bn");
    return SRVP_OK;
}

extern "C" int srvp_conv_in_wgrad(const float* x, const void* draw, float* dw, int N, int Cin, int H, int W, int Cout,
                                  int Cout_real, int k, int s, int p, void* stream) {
    SRVP_REQUIRE(x && draw && dw, "srvp_conv_in_wgrad: null pointer");
    SRVP_REQUIRE(Cin >= 1 && Cin <= MAXC && k <= MAXK && Cout % 8 == 0, "srvp_conv_in_wgrad: unsupported shape");
    int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
    SRVP_REQUIRE(k == 3 || k == 4, "srvp_conv_in_wgrad: k=%d unsupported", k);
    if (in_mfma_ok(Cin, H, W, Cout, k, s, p)) {
        hipStream_t st = (hipStream_t)stream;
        if (k == 3 && Cin == 3) launch_in_wgrad<3, 1, 3>(x, (const bf16_t*)draw, dw, N, Cout, Cout_real, st);
        else if (k == 3) launch_in_wgrad<3, 1, 1>(x, (const bf16_t*)draw, dw, N, Cout, Cout_real, st);
        else if (Cin == 3) launch_in_wgrad<4, 2, 3>(x, (const bf16_t*)draw, dw, N, Cout, Cout_real, st);
        else launch_in_wgrad<4, 2, 1>(x, (const bf16_t*)draw, dw, N, Cout, Cout_real, st);
        SRVP_CHECK_LAUNCH("srvp_conv_in_wgrad(mfma)");
        return SRVP_OK;
    }
    return conv_in_wgrad_valu<bf16_t>(x, draw, dw, N, Cin, H, W, Cout, Cout_real, k, s, p, stream);
}

// ---------------------------------------------------------------------------------------------------------
// Gradient hand-over of the image-side output layer: dpre = dx_ * x_ * (1 - x_) (sigmoid backward, conv.py:273-274)
// from the fp32 (N, nc, H, W) frame tensors into the bf16 NHWC tensor [N][H+2][W+2][C] (1-pixel zero border,
// channels >= nc zero) that the MFMA data-/weight-gradient kernels consume.
// ---------------------------------------------------------------------------------------------------------
namespace {
template <class E>
__global__ __launch_bounds__(256) void out_dpre_kernel(const float* __restrict__ xo, const float* __restrict__ dxo,
                                                       E* __restrict__ draw, float* __restrict__ dpre_f32, int N, int nc,
                                                       int H, int W, int C, int sigmoid) {
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
                if (dpre_f32) dpre_f32[o] = v;        // fp32 frame-layout copy: "image" of the fp32-MFMA data-gradient
            }
            f[e] = v;
        }
        size_t off = (((size_t)n * (H + 2) + y + 1) * (W + 2) + x + 1) * C + cg * 8;
        El<E>::st8(draw + off, f);
    }
}
}  // namespace

namespace {
// fp32 frames only (the image-side layer's gradients then run on the exact-fp32 MFMA first-layer kernels: no padded bf16 copy)
__global__ __launch_bounds__(256) void out_dpre_f32_kernel(const float* __restrict__ xo, const float* __restrict__ dxo,
                                                           float* __restrict__ dpre, long long n, int sigmoid) {
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= n) {
            f32x4_t d = *reinterpret_cast<const f32x4_t*>(dxo + i);
            if (sigmoid) { const f32x4_t s = *reinterpret_cast<const f32x4_t*>(xo + i); d = d * s * (1.f - s); }
            *reinterpret_cast<f32x4_t*>(dpre + i) = d;
        } else {
            for (long long j = i; j < n; ++j) { float v = dxo[j]; if (sigmoid) { const float s = xo[j]; v *= s * (1.f - s); } dpre[j] = v; }
        }
    }
}
}  // namespace

namespace {
template <class E>
int out_dpre_launch(const float* x_out, const float* dx_out, void* draw, float* dpre_f32, int N, int nc, int H, int W, int C,
                    int apply_sigmoid, void* stream) {
    long long total = (long long)N * H * W * (C / 8);
    long long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(out_dpre_kernel<E>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_out, dx_out, (E*)draw,
                       dpre_f32, N, nc, H, W, C, apply_sigmoid);
    SRVP_CHECK_LAUNCH("srvp_out_dpre");
    return SRVP_OK;
}
}  // namespace

extern "C" int srvp_out_dpre_f32(const float* x_out, const float* dx_out, void* draw, float* dpre_f32, int N, int nc, int H, int W,
                                 int C, int apply_sigmoid, void* stream) {
    SRVP_REQUIRE(x_out && dx_out && draw && C % 8 == 0 && nc <= C, "srvp_out_dpre_f32: bad args");
    return out_dpre_launch<float>(x_out, dx_out, draw, dpre_f32, N, nc, H, W, C, apply_sigmoid, stream);
}

extern "C" int srvp_out_dpre(const float* x_out, const float* dx_out, void* draw, float* dpre_f32, int N, int nc, int H, int W,
                             int C, int apply_sigmoid, void* stream) {
    SRVP_REQUIRE(x_out && dx_out && (draw || dpre_f32) && C % 8 == 0 && nc <= C, "srvp_out_dpre: bad args");
    if (!draw) {
        const long long n = (long long)N * nc * H * W;
        long long blocks = (n / 4 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(out_dpre_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_out, dx_out, dpre_f32, n,
                           apply_sigmoid);
        SRVP_CHECK_LAUNCH("srvp_out_dpre");
        return SRVP_OK;
    }
    return out_dpre_launch<bf16_t>(x_out, dx_out, draw, dpre_f32, N, nc, H, W, C, apply_sigmoid, stream);
}
