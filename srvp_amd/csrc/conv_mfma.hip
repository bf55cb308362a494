// Tap-table implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16).
//
//   out[m][j] = sum_t sum_c A_t[m][c] * W[t][j][c]        m = (n, oy, ox),  j = output channel
//
// One workgroup computes a BM x BN output tile; the K loop walks (tap, BK-channel chunk).  A rows are gathered
// straight from the NHWC bf16 activation tensor(s) (16-byte = 8-channel pieces, zero border physically present in
// memory so no bounds predicates), B rows from the tap-major packed weights.  Both tiles are register-staged into
// padded LDS rows (conflict-free ds_read_b128 fragment reads), double-buffered with the global loads of step s+1
// issued before the MFMAs of step s (one barrier per K step).  The fp32 accumulators feed (a) fp64-atomic
// per-channel sum / sum-of-squares for BatchNorm batch statistics and (b) a bf16 LDS-staged, 16-byte-wide store.
//
// Replaces: nn.Conv2d / nn.ConvTranspose2d forward + data-gradient (reference module/conv.py:174-179, 200-223,
// 299-304, 330-353), torch.cat of the skip connection (conv.py:270), nn.Upsample (conv.py:331-349).
#include "common.h"
#include "../../include/srvp_hip.h"

namespace {

struct RowInfo { int n, oy, ox; };

// Compact kernel-argument block (no arrays: tap offsets are packed 2 bits each so nothing is dynamically indexed
// in private memory).
struct ConvK {
    const bf16_t* src0; const bf16_t* src1; const int* map1;
    int C0, C1, H0p, W0p, H1p, W1p, ups0, ups1, si, ntaps;
    unsigned long long dy_bits, dx_bits;
    const bf16_t* wt; int Cout, N, OH, OW;
    bf16_t* dst; int DHp, DWp, so, ooy, oox, Cdst, cdst_off;
    double* stats; int stat_mod;
};

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void conv_mfma_kernel(const ConvK a) {
    constexpr int NT = WM * WN * 64;
    constexpr int CPR = BK / 8;                 // 16-byte chunks per tile row
    constexpr int LDR = BK + 8;                 // padded LDS row (elements): (row*LDR*2/16) % 16 is a bijection
    constexpr int A_LD = (BM * CPR + NT - 1) / NT;
    constexpr int B_LD = (BN * CPR + NT - 1) / NT;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDC = BN + 8;
    static_assert((BM * CPR) % NT == 0, "A tile must divide evenly over the workgroup");
    constexpr int AB_BYTES = 2 * (BM + BN) * LDR * 2;
    constexpr int C_BYTES = BM * LDC * 2;
    constexpr int SMEM = AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM + WM * BN * 8];
    bf16_t* As = reinterpret_cast<bf16_t*>(smem);                         // [2][BM][LDR]
    bf16_t* Bs = As + 2 * BM * LDR;                                       // [2][BN][LDR]
    float* red = reinterpret_cast<float*>(smem + SMEM);                   // [WM][BN][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int Ctot = a.C0 + a.C1;
    const long long M = (long long)a.N * a.OH * a.OW;
    // blockIdx.x -> (m tile, n tile): n fastest so that neighbouring workgroups share the same A rows in L2
    const int n_tiles = a.Cout / BN;
    const long long m0 = (long long)(blockIdx.x / n_tiles) * BM;
    const int n0 = (blockIdx.x % n_tiles) * BN;

    // ---- per-thread gather rows (fixed over the K loop) ----
    RowInfo ri[A_LD];
    int a_ch[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        int q = tid + i * NT;
        int row = q / CPR;
        a_ch[i] = q % CPR;
        long long m = m0 + row;
        if (m >= M) m = M - 1;                       // clamp: duplicates are masked at the store
        int hw = a.OH * a.OW;
        int n = (int)(m / hw);
        int r = (int)(m - (long long)n * hw);
        ri[i].n = n; ri[i].oy = r / a.OW; ri[i].ox = r - (r / a.OW) * a.OW;
    }
    const int kpt = Ctot / BK;                       // K steps per tap
    const int S = a.ntaps * kpt;

    u32x4_t ra[A_LD], rb[B_LD];
    auto load_step = [&](int s) {
        int t = s / kpt;
        int c = (s - t * kpt) * BK;
        const bf16_t* src; int C, Hp, Wp, ups; bool second = c >= a.C0;
        if (!second) { src = a.src0; C = a.C0; Hp = a.H0p; Wp = a.W0p; ups = a.ups0; }
        else { src = a.src1; C = a.C1; Hp = a.H1p; Wp = a.W1p; ups = a.ups1; c -= a.C0; }
        const int dy = (int)((a.dy_bits >> (4 * t)) & 15), dx = (int)((a.dx_bits >> (4 * t)) & 15);
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            int vy = ri[i].oy * a.si + dy, vx = ri[i].ox * a.si + dx;
            if (ups) { vy = (vy + 1) >> 1; vx = (vx + 1) >> 1; }
            int n = ri[i].n;
            if (second && a.map1) n = a.map1[n];
            size_t off = (((size_t)n * Hp + vy) * Wp + vx) * C + c + a_ch[i] * 8;
            ra[i] = *reinterpret_cast<const u32x4_t*>(src + off);
        }
        const bf16_t* w = a.wt + ((size_t)t * a.Cout + n0) * Ctot + (s - t * kpt) * BK;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            int q = tid + i * NT;
            int row = q / CPR, ch = q % CPR;
            if (q < BN * CPR) rb[i] = *reinterpret_cast<const u32x4_t*>(w + (size_t)row * Ctot + ch * 8);
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            int q = tid + i * NT;
            int row = q / CPR, ch = q % CPR;
            *reinterpret_cast<u32x4_t*>(As + ((size_t)buf * BM + row) * LDR + ch * 8) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            int q = tid + i * NT;
            int row = q / CPR, ch = q % CPR;
            if (q < BN * CPR) *reinterpret_cast<u32x4_t*>(Bs + ((size_t)buf * BN + row) * LDR + ch * 8) = rb[i];
        }
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_step(0);
    store_step(0);
    __syncthreads();
    const int lrow = lane & 31, lk = (lane >> 5) * 8;
    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        if (s + 1 < S) load_step(s + 1);
        const bf16_t* Ab = As + ((size_t)buf * BM + wm * (TM * 32) + lrow) * LDR + lk;
        const bf16_t* Bb = Bs + ((size_t)buf * BN + wn * (TN * 32) + lrow) * LDR + lk;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8_t af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(Ab + i * 32 * LDR + kk * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bb + j * 32 * LDR + kk * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < S) store_step(buf ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int lcol = lane & 31, lhalf = lane >> 5;
    if (a.stats) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    long long m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                    float v = (m < M) ? acc[i][j][r] : 0.f;
                    s1 += v; s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lhalf == 0) {
                int col = wn * (TN * 32) + j * 32 + lcol;
                red[(wm * BN + col) * 2 + 0] = s1;
                red[(wm * BN + col) * 2 + 1] = s2;
            }
        }
    }
    bf16_t* Cs = reinterpret_cast<bf16_t*>(smem);     // [BM][LDC], reuses the A/B buffers (all waves are past the loop)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                int col = wn * (TN * 32) + j * 32 + lcol;
                Cs[row * LDC + col] = f2bf(acc[i][j][r]);
            }
    __syncthreads();
    if (a.stats && tid < BN) {
        double s1 = 0., s2 = 0.;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s1 += red[(w * BN + tid) * 2]; s2 += red[(w * BN + tid) * 2 + 1]; }
        int ch = (n0 + tid) % a.stat_mod;
        atomicAdd(a.stats + ch, s1);
        atomicAdd(a.stats + a.stat_mod + ch, s2);
    }
    constexpr int CCH = BN / 8;
    bf16_t* dst = a.dst;
    for (int q = tid; q < BM * CCH; q += NT) {
        int row = q / CCH, ch = q % CCH;
        long long m = m0 + row;
        if (m >= M) continue;
        int hw = a.OH * a.OW;
        int n = (int)(m / hw);
        int r = (int)(m - (long long)n * hw);
        int oy = r / a.OW, ox = r - oy * a.OW;
        size_t off = (((size_t)n * a.DHp + oy * a.so + a.ooy) * a.DWp + ox * a.so + a.oox) * a.Cdst + a.cdst_off + n0 + ch * 8;
        *reinterpret_cast<u32x4_t*>(dst + off) = *reinterpret_cast<const u32x4_t*>(Cs + row * LDC + ch * 8);
    }
}

template <int BM, int BN, int BK, int WM, int WN>
int launch(const srvp_conv_desc* d, hipStream_t st) {
    long long M = (long long)d->N * d->OH * d->OW;
    long long mt = (M + BM - 1) / BM;
    long long blocks = mt * (d->Cout / BN);
    SRVP_REQUIRE(blocks > 0 && blocks < (1ll << 31), "srvp_conv_mfma: bad grid %lld", blocks);
    ConvK k;
    k.src0 = (const bf16_t*)d->src0; k.src1 = (const bf16_t*)d->src1; k.map1 = d->map1;
    k.C0 = d->C0; k.C1 = d->C1; k.H0p = d->H0p; k.W0p = d->W0p; k.H1p = d->H1p; k.W1p = d->W1p;
    k.ups0 = d->ups0; k.ups1 = d->ups1; k.si = d->si; k.ntaps = d->ntaps;
    k.dy_bits = 0; k.dx_bits = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        SRVP_REQUIRE(d->dy[t] >= 0 && d->dy[t] < 16 && d->dx[t] >= 0 && d->dx[t] < 16, "srvp_conv_mfma: tap offset out of [0,15]");
        k.dy_bits |= (unsigned long long)d->dy[t] << (4 * t);
        k.dx_bits |= (unsigned long long)d->dx[t] << (4 * t);
    }
    k.wt = (const bf16_t*)d->wt; k.Cout = d->Cout; k.N = d->N; k.OH = d->OH; k.OW = d->OW;
    k.dst = (bf16_t*)d->dst; k.DHp = d->DHp; k.DWp = d->DWp; k.so = d->so; k.ooy = d->ooy; k.oox = d->oox;
    k.Cdst = d->Cdst; k.cdst_off = d->cdst_off; k.stats = d->stats; k.stat_mod = d->stat_mod;
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, BK, WM, WN>), dim3((unsigned)blocks), dim3(WM * WN * 64), 0, st, k);
    SRVP_CHECK_LAUNCH("srvp_conv_mfma");
    return SRVP_OK;
}

}  // namespace

extern "C" int srvp_conv_mfma(const srvp_conv_desc* d, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SRVP_REQUIRE(d && d->src0 && d->wt && d->dst, "srvp_conv_mfma: null pointer");
    SRVP_REQUIRE(d->C0 % 32 == 0 && d->C1 % 32 == 0 && d->Cout % 32 == 0 && d->C0 > 0,
                 "srvp_conv_mfma: channel counts must be padded to 32 (C0=%d C1=%d Cout=%d)", d->C0, d->C1, d->Cout);
    SRVP_REQUIRE(d->C1 == 0 || d->src1, "srvp_conv_mfma: C1>0 needs src1");
    SRVP_REQUIRE(d->ntaps >= 1 && d->ntaps <= SRVP_MAX_TAPS, "srvp_conv_mfma: ntaps=%d", d->ntaps);
    SRVP_REQUIRE(d->Cdst % 8 == 0 && d->cdst_off % 8 == 0, "srvp_conv_mfma: dst channel slice must be 16-byte aligned");
    SRVP_REQUIRE(d->stats == nullptr || d->stat_mod > 0, "srvp_conv_mfma: stat_mod");
    const bool k64 = (d->C0 % 64 == 0) && (d->C1 % 64 == 0);
    if (d->Cout % 128 == 0) {
        return k64 ? launch<128, 128, 64, 2, 2>(d, st) : launch<128, 128, 32, 2, 2>(d, st);
    } else if (d->Cout % 64 == 0) {
        return k64 ? launch<128, 64, 64, 2, 2>(d, st) : launch<128, 64, 32, 2, 2>(d, st);
    } else {
        return launch<128, 32, 32, 4, 1>(d, st);
    }
}
