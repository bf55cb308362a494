// Latent path of SRVP in fp32: Linear / MLP building block (generic strided GEMM with fused bias, activation and
// ReLU-mask epilogues), LSTM recurrence, the residual Euler rollout with its prior network, and their backward
// passes.  <0.1 % of the model FLOPs but a long serial chain: the C entry points sequence the whole
// chain on the stream with no host synchronisation.
//
// Replaces: module/mlp.py:21-90, nn.LSTM (module/srvp.py:132,366), q_z / p_z / dynamics and the Euler loop
// (srvp.py:280-323, 370-405), utils.rsample_normal (module/utils.py:115-134), and their autograd backward.
#include "common.h"
#include "../../include/srvp_hip.h"

int srvp_rollout_fused_fwd(const srvp_rollout_desc* d, hipStream_t st, bool counters_cleared);          // rollout_fused.hip
int srvp_rollout_fused_bwd(const srvp_rollout_bwd_desc* d, hipStream_t st);
int64_t srvp_rollout_fused_cnt_words(const srvp_rollout_desc* d);                 // 4-byte words of the two counter blocks at the start of fused_ws
int srvp_rollout_gen_fwd(const srvp_rollout_desc* d, hipStream_t st);            // rollout_fused.hip: the inference chain, p_z inside

namespace {

inline bool use_fused(const srvp_rollout_desc& f) {
    if (!f.fused_ws) return false;
    const int64_t need = srvp_rollout_fused_ws_bytes(&f);
    return need > 0 && f.fused_ws_bytes >= need;
}

// ---------------------------------------------------------------------------------------------------------
// C[M][N] (+)= epi( A[M][K] * B[K][N] + bias ),  generic element strides.
// epi: activation, then optional multiply by (mask > 0) (ReLU derivative taken from the saved post-ReLU activation).
// ---------------------------------------------------------------------------------------------------------
struct GemmArgs {
    const float* A; long long a_rs, a_cs;
    const float* B; long long b_rs, b_cs;
    const float* bias; float* C; long long c_rs;
    const float* mask; long long mask_rs;
    int M, N, K, act, accumulate;
    float alpha;
};

// Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain).  One workgroup = one 32x32 tile of C;
// its 4 waves split K and are summed through LDS (the latent GEMMs are small and launch/latency bound, so the
// lever is parallelism over K and over 32x32 tiles, not tile size).  Operands are read straight into MFMA fragments
// with generic element strides: lane l feeds A[m0 + (l&31)][k + (l>>5)] and B[k + (l>>5)][n0 + (l&31)].
template <bool AVEC, bool BVEC>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const GemmArgs g) {
    __shared__ float red[3][32][33];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int r = lane & 31, kh = lane >> 5;
    const bool mv = (m0 + r) < g.M, nv = (n0 + r) < g.N;
    const float* Ap = g.A + (long long)(mv ? m0 + r : 0) * g.a_rs;
    const float* Bp = g.B + (long long)(nv ? n0 + r : 0) * g.b_cs;
    // K split over the 4 waves in multiples of 8 (the VEC path walks K in 8-wide groups: 4 per half-wave)
    const int kper = (((g.K + 3) / 4) + 7) & ~7;
    const int kb = wid * kper;
    int ke = kb + kper; if (ke > g.K) ke = g.K;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // These GEMMs are tiny (M = batch rows) and latency bound: ALL operand loads of a 128-wide K chunk are issued
    // before the first MFMA (one memory round trip per chunk instead of one per 16 k).  The MFMA K order is free as
    // long as A and B agree: half-wave kh takes the k indices 8j + 4kh + {0..3}, one float4 load for an operand that is
    // K-contiguous and 16-byte aligned (AVEC / BVEC), four lane-coalesced 4-byte loads otherwise.
    constexpr int CH = 128;
    for (int kc = kb; kc < ke; kc += CH) {
        f32x4_t a4[CH / 8], b4[CH / 8];
#pragma unroll
        for (int j = 0; j < CH / 8; ++j) {
            const int kk = kc + 8 * j + 4 * kh;
            if constexpr (AVEC) {
                // K % 4 == 0 and kper % 8 == 0: a float4 is all-in or all-out
                a4[j] = (kk < ke && mv) ? *reinterpret_cast<const f32x4_t*>(Ap + kk) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) a4[j][i] = (kk + i < ke && mv) ? Ap[(long long)(kk + i) * g.a_cs] : 0.f;
            }
            if constexpr (BVEC) {
                b4[j] = (kk < ke && nv) ? *reinterpret_cast<const f32x4_t*>(Bp + kk) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) b4[j][i] = (kk + i < ke && nv) ? Bp[(long long)(kk + i) * g.b_rs] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < CH / 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j][i], b4[j][i], acc, 0, 0, 0);
    }
    // C/D layout: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
    if (wid > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wid - 1][(i & 3) + 8 * (i >> 2) + 4 * kh][r] = acc[i];
    }
    __syncthreads();
    if (wid > 0) return;
    const int n = n0 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
        const int m = m0 + row;
        if (m >= g.M || n >= g.N) continue;
        float v = (acc[i] + red[0][row][r] + red[1][row][r] + red[2][row][r]) * g.alpha;
        if (g.bias) v += g.bias[n];
        v = act_fwd(v, g.act);
        if (g.mask) v = g.mask[(long long)m * g.mask_rs + n] > 0.f ? v : 0.f;
        float* c = g.C + (long long)m * g.c_rs + n;
        *c = g.accumulate ? *c + v : v;
    }
}

// The same GEMM for the shapes with thousands of rows (the batched inference MLPs and their data gradients: M = frames x batch, N and K a few
// hundred; round 6).  The 32x32 kernel above reads both operands straight from L2 with no reuse -- 4 MAC per byte: [2112 x 512] x [512 x 512]
// took 35 us (31 TFLOP/s of the 157 the fp32 matrix pipe has) and doubling these launches cost 0.30 ms of a 38 ms step, 0.20 of SM-MNIST's
// 5.6.  Here: a 64 x 64 tile per workgroup (four waves, 32 x 32 each), 32-wide K stages of both operands staged in LDS through registers
// (double-buffered), fragments read back as 16-byte pieces -- 16 MAC per byte from L2.  Same exact-fp32 arithmetic (v_mfma_f32_32x32x2_f32); the K
// order inside a stage is 8 j + 4 h + {0..3} for half-wave h on BOTH operands.  A: row-major, K contiguous.  BT: B given as [N][K] (K contiguous:
// y = x W^T); !BT: B row-major [K][N] (N contiguous: dx = dy W).
template <bool BT>
__global__ __launch_bounds__(256) void gemm_f32_tiled_kernel(const GemmArgs g) {
    // 32-wide K stages: the loads of stage s + 1 are issued before the arithmetic of stage s and stored behind it.  (Measured, round 6: two
    // register sets / two stages ahead -- behind the loop's branches hipcc waits with vmcnt(0) for the prefetch it has just issued, so the
    // distance collapses to one stage anyway -- and 64-wide stages: the same 27 us for [2112 x 512] x [512 x 512], slower at K = 128.)
    constexpr int BM = 64, BN = 64, BK = 32, LDA = BK + 4, LDBT = BK + 4, LDBN = BN + 4;
    constexpr int KQ = BK / 4, NP = BM * KQ / 256;           // float4 pieces per row, pieces per thread and operand
    __shared__ __attribute__((aligned(16))) float As[2][BM][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BT ? BN * LDBT : BK * LDBN];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int r = lane & 31, h = lane >> 5;
    // staging: loads are UNCONDITIONAL at clamped addresses, the validity is applied when the piece is stored (a guarded load is a branch)
    f32x4_t ra[NP], rb[NP];
    bool va[NP], vb[NP];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = tid + i * 256;
            const int row = q / KQ, kq = (q % KQ) * 4;
            const int m = m0 + row, k = k0 + kq;
            va[i] = m < g.M && k < g.K;
            const int mc = m < g.M ? m : g.M - 1, kc = k < g.K ? k : g.K - 4;
            ra[i] = *reinterpret_cast<const f32x4_t*>(g.A + (long long)mc * g.a_rs + kc);
            if constexpr (BT) {
                const int n = n0 + row;
                vb[i] = n < g.N && k < g.K;
                rb[i] = *reinterpret_cast<const f32x4_t*>(g.B + (long long)(n < g.N ? n : g.N - 1) * g.b_cs + kc);
            } else {
                const int kr = q >> 4, nq = (q & 15) * 4;
                const int kk = k0 + kr, n = n0 + nq;
                vb[i] = kk < g.K && n < g.N;
                rb[i] = *reinterpret_cast<const f32x4_t*>(g.B + (long long)(kk < g.K ? kk : g.K - 1) * g.b_rs + (n < g.N ? n : g.N - 4));
            }
        }
    };
    auto store = [&](int buf) {
        const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = tid + i * 256;
            *reinterpret_cast<f32x4_t*>(&As[buf][q / KQ][(q % KQ) * 4]) = va[i] ? ra[i] : zero;
            if constexpr (BT) *reinterpret_cast<f32x4_t*>(&Bs[buf][(q / KQ) * LDBT + (q % KQ) * 4]) = vb[i] ? rb[i] : zero;
            else *reinterpret_cast<f32x4_t*>(&Bs[buf][(q >> 4) * LDBN + (q & 15) * 4]) = vb[i] ? rb[i] : zero;
        }
    };
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    auto compute = [&](int buf) {
        const float* ar = &As[buf][wm * 32 + r][4 * h];
#pragma unroll
        for (int jj = 0; jj < BK / 32; ++jj) {               // fragments of 32 k requested, then their 16 MFMAs
            f32x4_t a4[4], b4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kb = 32 * jj + 8 * j;
                a4[j] = *reinterpret_cast<const f32x4_t*>(ar + kb);
                if constexpr (BT) b4[j] = *reinterpret_cast<const f32x4_t*>(&Bs[buf][(wn * 32 + r) * LDBT + kb + 4 * h]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b4[j][e] = Bs[buf][(kb + 4 * h + e) * LDBN + wn * 32 + r];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j][e], b4[j][e], acc, 0, 0, 0);
        }
    };
    const int nst = (g.K + BK - 1) / BK;
    load(0);
    store(0);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        if (s + 1 < nst) load((s + 1) * BK);                 // the next stage's global loads fly under this stage's MFMAs
        compute(s & 1);
        if (s + 1 < nst) store((s & 1) ^ 1);
        __syncthreads();
    }
    // C/D layout: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
    const int n = n0 + wn * 32 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (m >= g.M || n >= g.N) continue;
        float v = acc[i] * g.alpha;
        if (g.bias) v += g.bias[n];
        v = act_fwd(v, g.act);
        if (g.mask) v = g.mask[(long long)m * g.mask_rs + n] > 0.f ? v : 0.f;
        float* c = g.C + (long long)m * g.c_rs + n;
        *c = g.accumulate ? *c + v : v;
    }
}

__global__ void colsum_kernel(const float* A, long long a_rs, float* out, int M, int N, int rows_per_block);

// Weight-gradient shape: C[M][N] += A^T B with BOTH operands row-major over the reduction index (A = delta [K][M], B =
// activations [K][N], K = (steps x batch) rows in the thousands, M, N = layer widths).  The 32x32 kernel above reads such
// operands 4 bytes per lane per MFMA with nothing staged (4 MAC per L2 byte: 0.3-0.6 ms per GEMM at K = 4224, 1.6 ms per
// training step on the serial latent-backward path).  Here a workgroup owns a 64x64 tile, stages 32-row chunks of both
// operands in LDS with 16-byte loads (double-buffered, every LDS value feeds two MFMAs' worth of lanes), and K is split over
// gridDim.z with fp32 atomics into the (accumulated anyway) gradient.  Exact fp32 MFMA as everywhere in the latent path.
template <bool VEC>
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb,
                                                          float* C, long long ldc, int M, int N, int K, int kper, float* colsum) {
    constexpr int LD = 96;                                 // row stride: the two k rows of an MFMA land on disjoint bank halves
    __shared__ __attribute__((aligned(16))) float As[2][32 * LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][32 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, r = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int kb = blockIdx.z * kper;
    int ke = kb + kper; if (ke > K) ke = K;
    if (kb >= ke) return;
    f32x16_t acc, accs;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; accs[i] = 0.f; }
    // bias gradient for free: the column sums of A (= sum over rows of delta) are A^T x ones -- one extra MFMA per k pair in the
    // waves of the first column tile (the separate column-sum launch per layer was 16 launches per step)
    const bool do_sum = colsum != nullptr && blockIdx.x == 0 && wn == 0;
    // staging: 32 rows x 16 four-column pieces per operand = 512 pieces, two per thread
    f32x4_t ra[2], rb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = tid + 256 * u, row = p >> 4, c4 = (p & 15) * 4;
            const int k = k0 + row;
            f32x4_t va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (k < ke) {
                const float* ap = A + (long long)k * lda + m0 + c4;
                const float* bp = B + (long long)k * ldb + n0 + c4;
                if (VEC) {
                    if (m0 + c4 < M) va = *reinterpret_cast<const f32x4_t*>(ap);      // (M, N multiples of 4: a piece is all-in or all-out)
                    if (n0 + c4 < N) vb = *reinterpret_cast<const f32x4_t*>(bp);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { if (m0 + c4 + e < M) va[e] = ap[e]; if (n0 + c4 + e < N) vb[e] = bp[e]; }
                }
            }
            ra[u] = va; rb[u] = vb;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = tid + 256 * u, row = p >> 4, c4 = (p & 15) * 4;
            *reinterpret_cast<f32x4_t*>(&As[buf][row * LD + c4]) = ra[u];
            *reinterpret_cast<f32x4_t*>(&Bs[buf][row * LD + c4]) = rb[u];
        }
    };
    fetch(kb);
    int buf = 0;
    for (int k0 = kb; k0 < ke; k0 += 32, buf ^= 1) {
        commit(buf);
        __syncthreads();
        if (k0 + 32 < ke) fetch(k0 + 32);                 // next chunk's global loads fly under this chunk's MFMAs
        const float* ap = &As[buf][kh * LD + wm * 32 + r];
        const float* bp = &Bs[buf][kh * LD + wn * 32 + r];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk * LD], bp[2 * kk * LD], acc, 0, 0, 0);
        if (do_sum) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) accs = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk * LD], 1.f, accs, 0, 0, 0);
        }
        // (the other buffer is rewritten only after the next barrier: one barrier per chunk suffices with two buffers)
    }
    if (do_sum && r == 0) {                               // every column of accs holds the sums: lane column 0 of each half writes them
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
            if (m < M) atomicAdd(colsum + m, accs[i]);
        }
    }
    const int n = n0 + wn * 32 + r;
    if (n >= N) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
        if (m < M) atomicAdd(C + (long long)m * ldc + n, acc[i]);
    }
}

int gemm(hipStream_t st, const float* A, long long a_rs, long long a_cs, const float* B, long long b_rs, long long b_cs,
         const float* bias, float* C, long long c_rs, int M, int N, int K, int act, int accumulate,
         const float* mask = nullptr, long long mask_rs = 0, float alpha = 1.f, float* colsum = nullptr) {
    if (M <= 0 || N <= 0) return SRVP_OK;
    static int tn_on = -1;
    if (tn_on < 0) { const char* e = getenv("SRVP_GEMM_TN"); tn_on = e ? atoi(e) : 1; }
    if (tn_on && a_rs == 1 && b_cs == 1 && accumulate && !bias && !mask && act == ACT_NONE && alpha == 1.f && K >= 128) {
        // A^T B, both operands row-major over K (weight gradients)
        const long long tiles = (long long)((M + 63) / 64) * ((N + 63) / 64);
        int splits = (int)((768 + tiles - 1) / tiles);
        const int maxs = (K + 63) / 64;
        if (splits > maxs) splits = maxs;
        if (splits < 1 || g_srvp_det) splits = 1;           // deterministic mode: one split = one atomic per element, onto what stream order left there
        int kper = ((K + splits - 1) / splits + 31) / 32 * 32;
        splits = (K + kper - 1) / kper;
        const dim3 grid((N + 63) / 64, (M + 63) / 64, splits);
        const bool vec = M % 4 == 0 && N % 4 == 0 && a_cs % 4 == 0 && b_rs % 4 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0;
        if (vec) hipLaunchKernelGGL(gemm_tn_f32_kernel<true>, grid, dim3(256), 0, st, A, a_cs, B, b_rs, C, c_rs, M, N, K, kper, colsum);
        else hipLaunchKernelGGL(gemm_tn_f32_kernel<false>, grid, dim3(256), 0, st, A, a_cs, B, b_rs, C, c_rs, M, N, K, kper, colsum);
        SRVP_CHECK_LAUNCH("srvp_gemm_f32(tn)");
        return SRVP_OK;
    }
    if (colsum) {                                          // (small K: the column sums as their own launch)
        const int rpb_c = g_srvp_det ? K : 128;           // (deterministic mode: one row slice per column block)
        hipLaunchKernelGGL(colsum_kernel, dim3((M + 63) / 64, (K + rpb_c - 1) / rpb_c), dim3(256), 0, st, A, a_cs, colsum, K, M, rpb_c);
    }
    GemmArgs g{A, a_rs, a_cs, B, b_rs, b_cs, bias, C, c_rs, mask, mask_rs, M, N, K, act, accumulate, alpha};
    dim3 grid((N + 31) / 32, (M + 31) / 32);
    const bool av = a_cs == 1 && K % 4 == 0 && a_rs % 4 == 0 && ((uintptr_t)A % 16) == 0;
    const bool bv = b_rs == 1 && K % 4 == 0 && b_cs % 4 == 0 && ((uintptr_t)B % 16) == 0;
    // many rows, a wide enough output: the LDS-tiled kernel (SRVP_GEMM_TILED=0: the 32x32 kernel for every shape, A/B switch)
    static int tiled_on = -1;
    if (tiled_on < 0) { const char* e = getenv("SRVP_GEMM_TILED"); tiled_on = e ? atoi(e) : 1; }
    const bool bn = b_cs == 1 && N % 4 == 0 && b_rs % 4 == 0 && ((uintptr_t)B % 16) == 0;
    if (tiled_on && av && (bv || bn) && M >= 512 && N >= 96 && K >= 32) {
        dim3 tg((N + 63) / 64, (M + 63) / 64);
        if (bv) hipLaunchKernelGGL(gemm_f32_tiled_kernel<true>, tg, dim3(256), 0, st, g);
        else hipLaunchKernelGGL(gemm_f32_tiled_kernel<false>, tg, dim3(256), 0, st, g);
        SRVP_CHECK_LAUNCH("srvp_gemm_f32(tiled)");
        return SRVP_OK;
    }
    if (av && bv) hipLaunchKernelGGL((gemm_f32_mfma_kernel<true, true>), grid, dim3(256), 0, st, g);
    else if (av) hipLaunchKernelGGL((gemm_f32_mfma_kernel<true, false>), grid, dim3(256), 0, st, g);
    else if (bv) hipLaunchKernelGGL((gemm_f32_mfma_kernel<false, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_mfma_kernel<false, false>), grid, dim3(256), 0, st, g);
    SRVP_CHECK_LAUNCH("srvp_gemm_f32");
    return SRVP_OK;
}

__global__ void colsum_kernel(const float* A, long long a_rs, float* out, int M, int N, int rows_per_block) {
    // workgroup = 64 columns x one slice of rows; 4 waves stride over the slice, one atomic per column and workgroup
    __shared__ float part[4][64];
    int n = blockIdx.x * 64 + (threadIdx.x & 63);
    int w = threadIdx.x >> 6;
    int r0 = blockIdx.y * rows_per_block, r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float s = 0.f;
    if (n < N)
        for (int m = r0 + w; m < r1; m += 4) s += A[m * a_rs + n];
    part[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && n < N) atomicAdd(out + n, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

__global__ void act_bwd_kernel(const float* v, const float* dy, float* dx, long long n, int act, int from_output) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d;
    if (from_output) {
        float o = v[i];
        switch (act) {
            case ACT_RELU: d = o > 0.f ? 1.f : 0.f; break;
            case ACT_TANH: d = 1.f - o * o; break;
            case ACT_SIGMOID: d = o * (1.f - o); break;
            case ACT_LRELU: d = o > 0.f ? 1.f : LRELU_SLOPE; break;
            default: d = 1.f;
        }
    } else {
        d = act_bwd(v[i], act);
    }
    dx[i] = dy[i] * d;
}

// ---------------------------------------------------------------------------------------------------------
// rsample (utils.py:108-112,132-133): out = loc + eps * (softplus(raw) + 1e-8), softplus threshold 20
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(__expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ void rsample_fwd_kernel(const float* params, const float* eps, float* out, long long rows, int d) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * d) return;
    long long r = i / d; int c = (int)(i - r * d);
    const float* p = params + r * 2 * d;
    out[i] = p[c] + eps[i] * (softplus_f(p[d + c]) + 1e-8f);
}
// dparams[r][c] (+)= dout ; dparams[r][d+c] (+)= dout * eps * softplus'(raw)
__global__ void rsample_bwd_kernel(const float* params, const float* eps, const float* dout, float* dparams, long long rows,
                                   int d, int accumulate, long long dp_rs) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * d) return;
    long long r = i / d; int c = (int)(i - r * d);
    float raw = params[r * 2 * d + d + c];
    float ds = raw > 20.f ? 1.f : sigmoid_f(raw);
    float g = dout[i];
    float* o = dparams + r * dp_rs;
    if (accumulate) { o[c] += g; o[d + c] += g * eps[i] * ds; }
    else { o[c] = g; o[d + c] = g * eps[i] * ds; }
}

// ---------------------------------------------------------------------------------------------------------
// small row helpers for the rollout
// ---------------------------------------------------------------------------------------------------------
// inp[b] = [y[b] (ny), z[b] (nz)]
__global__ void concat_yz_kernel(const float* y, const float* z, float* inp, int B, int ny, int nz) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int w = ny + nz;
    if (i >= B * w) return;
    int b = i / w, c = i - b * w;
    inp[i] = c < ny ? y[b * ny + c] : z[b * nz + c - ny];
}
// res = dt*out ; y_next = y + res
__global__ void euler_update_kernel(const float* y, const float* out, float dt, float* res, float* y_next, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = dt * out[i];
    res[i] = r;
    y_next[i] = y[i] + r;
}
// backward seed of one Euler step: dy = d_y_all[i+1] + carry ; dout = dt * (d_res + dy)
__global__ void euler_bwd_seed_kernel(const float* d_y_next, const float* carry, const float* d_res, float dt, float* dy,
                                      float* dout, int B, int ny, int dout_rs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * ny) return;
    int b = i / ny, c = i - b * ny;
    float v = (d_y_next ? d_y_next[i] : 0.f) + carry[i];
    dy[i] = v;
    dout[(size_t)b * dout_rs + c] = dt * ((d_res ? d_res[i] : 0.f) + v);
}
// dst[b][c] (row stride rs) = src[b][c] (compact, width w) or 0
__global__ void rows_copy_kernel(float* dst, int rs, const float* src, int B, int w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * w) return;
    int b = i / w, c = i - b * w;
    dst[(size_t)b * rs + c] = src ? src[i] : 0.f;
}
// carry = dy + dinp[:, :ny] ; dz_acc (+)= dinp[:, ny:]
__global__ void euler_bwd_split_kernel(const float* dy, const float* dinp, float* carry, float* dz_acc, int B, int ny, int nz,
                                       int first_of_frame_in_reverse) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int w = ny + nz;
    if (i >= B * w) return;
    int b = i / w, c = i - b * w;
    if (c < ny) carry[b * ny + c] = dy[b * ny + c] + dinp[i];
    else {
        float* o = dz_acc + b * nz + c - ny;
        *o = first_of_frame_in_reverse ? dinp[i] : *o + dinp[i];
    }
}
// ---- posterior-only (pz_external) chains: everything that is not serial leaves the per-step sequence
// inp_all[i][b][ny + c] = z[i / ne][b][c] for every step; y part of step 0 = y0
__global__ void fill_inp_kernel(const float* z, const float* y0, float* inp_all, int S, int ne, int B, int ny, int nz) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = ny + nz;
    if (i >= (long long)S * B * w) return;
    const int c = (int)(i % w); long long q = i / w;
    const int b = (int)(q % B), st = (int)(q / B);
    if (c >= ny) inp_all[i] = z[((size_t)(st / ne) * B + b) * nz + c - ny];
    else if (st == 0) inp_all[i] = y0[b * ny + c];
}
// Everything in front of the persistent chain as ONE launch (round 5; it was a copy, two kernels and a memset on the step's serial path):
// y_all[0] = y0; z = rsample(q_z, eps) for every frame; the z half of every step's MLP input and the y half of step 0; the cluster counters
// of the persistent kernels (forward AND backward block) cleared.
__global__ void rollout_prep_kernel(const float* q_z, const float* eps, const float* y0, float* z, float* y_all, float* inp_all, unsigned* cnt,
                                    int cnt_words, int F, int S, int ne, int B, int ny, int nz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = ny + nz;
    if (i < cnt_words) cnt[i] = 0u;
    if (i < (long long)B * ny) {
        const int b = (int)(i / ny), c = (int)(i - (long long)b * ny);
        const float v = y0[i];
        y_all[i] = v;
        inp_all[(size_t)b * w + c] = v;
    }
    if (i >= (long long)F * B * nz) return;
    const long long r = i / nz; const int c = (int)(i - r * nz);
    const int f = (int)(r / B), b = (int)(r - (long long)f * B);
    const float* p = q_z + r * 2 * nz;
    const float v = p[c] + eps[i] * (softplus_f(p[nz + c]) + 1e-8f);
    z[i] = v;
    for (int st = f * ne; st < (f + 1) * ne && st < S; ++st) inp_all[((size_t)st * B + b) * w + ny + c] = v;
}
// res = dt*out ; y_next = y + res, also written as the y part of the next step's MLP input
__global__ void euler_update2_kernel(const float* y, const float* out, float dt, float* res, float* y_next, float* inp_next, int B,
                                     int ny, int nin) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * ny) return;
    const float r = dt * out[i];
    res[i] = r;
    const float v = y[i] + r;
    y_next[i] = v;
    if (inp_next) { const int b = i / ny, c = i - b * ny; inp_next[(size_t)b * nin + c] = v; }
}
// backward of step i finished (dinp = gradient wrt [y_i, z]): carry = dy + dinp[:, :ny]; and the seed of step i-1:
// dy = d_y_all[i] + carry ; dout(i-1) = dt * (d_res[i-1] + dy)
__global__ void euler_bwd_chain_kernel(float* dy, const float* dinp, float* carry, const float* d_y_i, const float* d_res_prev, float dt,
                                       float* dout_prev, int B, int ny, int nin, int dout_rs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * ny) return;
    const int b = i / ny, c = i - b * ny;
    const float cr = dy[i] + dinp[(size_t)b * nin + c];
    carry[i] = cr;
    if (dout_prev) {
        const float v = (d_y_i ? d_y_i[i] : 0.f) + cr;
        dy[i] = v;
        dout_prev[(size_t)b * dout_rs + c] = dt * ((d_res_prev ? d_res_prev[i] : 0.f) + v);
    }
}
// gradient wrt z of every frame = sum over its sub-steps of dinp[:, ny:] (+ d_z), then the posterior sample's backward
__global__ void dz_finalize_kernel(const float* dinp_all, const float* d_z, const float* q_params, const float* eps, float* d_qz, int F,
                                   int S, int ne, int B, int ny, int nz) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)F * B * nz) return;
    const int c = (int)(i % nz); long long q = i / nz;
    const int b = (int)(q % B), f = (int)(q / B);
    const int nin = ny + nz;
    float g = d_z ? d_z[i] : 0.f;
    for (int s = f * ne; s < (f + 1) * ne && s < S; ++s) g += dinp_all[((size_t)s * B + b) * nin + ny + c];
    const float raw = q_params[((size_t)f * B + b) * 2 * nz + nz + c];
    const float ds = raw > 20.f ? 1.f : sigmoid_f(raw);
    float* o = d_qz + ((size_t)f * B + b) * 2 * nz;
    o[c] = g; o[nz + c] = g * eps[i] * ds;
}
__global__ void add_inplace_kernel(float* a, const float* b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += b[i];
}
__global__ void add3_kernel(float* out, const float* a, const float* b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (a ? a[i] : 0.f) + (b ? b[i] : 0.f);
}

// ---------------------------------------------------------------------------------------------------------
// LSTM pointwise
// ---------------------------------------------------------------------------------------------------------
// gates: [B][4nh] pre-activation (i,f,g,o) -> activations in place; c = f*c_prev + i*g ; h = o*tanh(c)
__global__ void lstm_cell_fwd_kernel(float* gates, const float* c_prev, float* c, float* h, int B, int nh) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nh) return;
    int b = i / nh, j = i - b * nh;
    float* g = gates + (size_t)b * 4 * nh;
    float ig = sigmoid_f(g[j]), fg = sigmoid_f(g[nh + j]), gg = tanhf(g[2 * nh + j]), og = sigmoid_f(g[3 * nh + j]);
    float cp = c_prev ? c_prev[i] : 0.f;
    float cc = fg * cp + ig * gg;
    g[j] = ig; g[nh + j] = fg; g[2 * nh + j] = gg; g[3 * nh + j] = og;
    c[i] = cc;
    h[i] = og * tanhf(cc);
}
// dh = dh_out + dh_carry ; produces dgates (pre-activation grads) and dc_carry (in place)
__global__ void lstm_cell_bwd_kernel(const float* dh_out, const float* dh_carry, float* dc_carry, const float* gates_act,
                                     const float* c, const float* c_prev, float* dgates, int B, int nh, int has_carry) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nh) return;
    int b = i / nh, j = i - b * nh;
    const float* g = gates_act + (size_t)b * 4 * nh;
    float ig = g[j], fg = g[nh + j], gg = g[2 * nh + j], og = g[3 * nh + j];
    float dh = (dh_out ? dh_out[i] : 0.f) + (has_carry ? dh_carry[i] : 0.f);
    float tc = tanhf(c[i]);
    float dc = (has_carry ? dc_carry[i] : 0.f) + dh * og * (1.f - tc * tc);
    float cp = c_prev ? c_prev[i] : 0.f;
    float* d = dgates + (size_t)b * 4 * nh;
    d[j] = dc * gg * ig * (1.f - ig);
    d[nh + j] = dc * cp * fg * (1.f - fg);
    d[2 * nh + j] = dc * ig * (1.f - gg * gg);
    d[3 * nh + j] = dh * tc * og * (1.f - og);
    dc_carry[i] = dc * fg;
}

__global__ void axpby_kernel(float* out, float a, const float* x, float b, const float* y, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

inline dim3 g1(long long n) { return dim3((unsigned)((n + 255) / 256)); }

// y = MLP(x): saves post-ReLU hidden activations hid + l*hid_ls (l = 0..nl-2), each [B][nh]
int mlp_fwd(hipStream_t st, const float* const* W, const float* const* b, int nl, int nin, int nh, int nout, const float* x,
            int B, float* hid, size_t hid_ls, float* out) {
    const float* cur = x;
    int cin = nin;
    for (int l = 0; l < nl; ++l) {
        const bool last = l == nl - 1;
        int cout = last ? nout : nh;
        float* dst = last ? out : hid + (size_t)l * hid_ls;
        int rc = gemm(st, cur, cin, 1, W[l], 1, cin, b[l], dst, cout, B, cout, cin, last ? ACT_NONE : ACT_RELU, 0);
        if (rc) return rc;
        cur = dst; cin = cout;
    }
    return SRVP_OK;
}
// deltas + l*del_ls (l = 0..nl-1): gradient wrt the pre-activation output of layer l, rows [B][dw]; slot nl-1 must
// already hold the output gradient.  dx (optional) = gradient wrt the MLP input ([B][nin] compact).
int mlp_bwd(hipStream_t st, const float* const* W, int nl, int nin, int nh, int nout, int B, const float* hid, size_t hid_ls,
            float* deltas, size_t del_ls, int dw, float* dx) {
    for (int l = nl - 1; l >= 1; --l) {
        int cout = l == nl - 1 ? nout : nh;
        // delta_{l-1} = (delta_l W_l) * relu'(h_{l-1})
        int rc = gemm(st, deltas + (size_t)l * del_ls, dw, 1, W[l], nh, 1, nullptr, deltas + (size_t)(l - 1) * del_ls, dw, B, nh,
                      cout, ACT_NONE, 0, hid + (size_t)(l - 1) * hid_ls, nh);
        if (rc) return rc;
    }
    if (dx) {
        int cout0 = nl == 1 ? nout : nh;
        return gemm(st, deltas, dw, 1, W[0], nin, 1, nullptr, dx, nin, B, nin, cout0, ACT_NONE, 0);
    }
    return SRVP_OK;
}

}  // namespace

extern "C" int srvp_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs,
                             const float* bias, float* C, int64_t c_rs, int M, int N, int K, int act, int accumulate,
                             void* stream) {
    SRVP_REQUIRE(A && B && C, "srvp_gemm_f32: null pointer");
    return gemm((hipStream_t)stream, A, a_rs, a_cs, B, b_rs, b_cs, bias, C, c_rs, M, N, K, act, accumulate);
}

// gw[M][N] += delta[:, :M]^T act[:, :N] ; gb[M] += column sums of delta  (K rows): the weight / bias gradient of one Linear layer
extern "C" int srvp_linear_wgrad_f32(const float* delta, int64_t ld_delta, const float* act, int64_t ld_act, float* gw, int64_t ld_gw,
                                     float* gb, int M, int N, int K, void* stream) {
    SRVP_REQUIRE(delta && act && gw, "srvp_linear_wgrad_f32: null pointer");
    return gemm((hipStream_t)stream, delta, 1, ld_delta, act, ld_act, 1, nullptr, gw, ld_gw, M, N, K, ACT_NONE, 1, nullptr, 0, 1.f, gb);
}

namespace {
// dst[blk * dst_stride + i] += src[blk * n + i]
__global__ void add_blocks_kernel(float* dst, long long dst_stride, const float* src, int nblk, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nblk * n) return;
    const long long b = i / n, e = i - b * n;
    dst[b * dst_stride + e] += src[i];
}
}  // namespace
extern "C" int srvp_add_blocks_f32(float* dst, int64_t dst_stride, const float* src, int nblk, int64_t n, void* stream) {
    SRVP_REQUIRE(dst && src, "srvp_add_blocks_f32: null pointer");
    if ((long long)nblk * n <= 0) return SRVP_OK;
    hipLaunchKernelGGL(add_blocks_kernel, g1((long long)nblk * n), dim3(256), 0, (hipStream_t)stream, dst, (long long)dst_stride, src, nblk, (long long)n);
    SRVP_CHECK_LAUNCH("srvp_add_blocks_f32");
    return SRVP_OK;
}

// ---- glue between the latent path and the conv stacks (was torch.cat / .sum(0) / index_add_: model arithmetic belongs here)
namespace {
// decoder input rows (srvp.py:216-221): dst[t*B + b][c] = c < nh ? w[b][c] : c < nh + ny ? y[t][b][c - nh] : 0 (channel padding)
template <class E>
__global__ void latent_to_z_kernel(const float* __restrict__ w, const float* __restrict__ y, long long y_tstride, E* __restrict__ dst,
                                   int nt, int B, int nh, int ny, int Cz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nt * B * Cz) return;
    const int c = (int)(i % Cz); const long long r = i / Cz;
    const int b = (int)(r % B), t = (int)(r / B);
    float v = 0.f;
    if (c < nh) v = w[(size_t)b * nh + c];
    else if (c < nh + ny) v = y[(size_t)t * y_tstride + (size_t)b * ny + (c - nh)];
    El<E>::st(dst + i, v);
}
// gradient of the decoder input -> d_w[b][c] = sum_t dz[t*B+b][c] (+ d_w_add), d_y[t][b][c] = dz[t*B+b][nh+c] (+ d_y_add) (the
// backward of the time-expansion of w and of the concatenation, srvp.py:216-221)
template <class E>
__global__ void dz_split_kernel(const E* __restrict__ dz, int Cz, int nt, int B, int nh, int ny, const float* __restrict__ d_w_add,
                                const float* __restrict__ d_y_add, float* __restrict__ d_w, float* __restrict__ d_y, long long dy_tstride) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nw = (long long)B * nh, nyy = (long long)nt * B * ny;
    if (i < nw) {
        const int c = (int)(i % nh), b = (int)(i / nh);
        float s = 0.f;
        for (int t = 0; t < nt; ++t) s += El<E>::ld(dz + ((size_t)t * B + b) * Cz + c);         // fixed order: deterministic
        d_w[i] = s + (d_w_add ? d_w_add[i] : 0.f);
    } else if (i < nw + nyy) {
        const long long j = i - nw;
        const int c = (int)(j % ny); const long long r = j / ny;
        const int b = (int)(r % B), t = (int)(r / B);
        d_y[(size_t)t * dy_tstride + (size_t)b * ny + c] = El<E>::ld(dz + (size_t)r * Cz + nh + c) + (d_y_add ? d_y_add[j] : 0.f);
    }
}
// dst[idx[r]][c] += src[r][c]   (rows of idx are distinct in every use here; atomics keep it correct if they are not)
__global__ void rows_scatter_add_kernel(float* __restrict__ dst, const long long* __restrict__ idx64, const int* __restrict__ idx32,
                                        const float* __restrict__ src, long long rows, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const long long r = i / C; const int c = (int)(i - r * C);
    const long long d = idx64 ? idx64[r] : (long long)idx32[r];
    atomicAdd(dst + d * C + c, src[i]);
}
}  // namespace
extern "C" int srvp_latent_to_z(const float* w, const float* y, int64_t y_tstride, void* dst, int nt, int B, int nh, int ny, int Cz,
                                int dst_f32, void* stream) {
    SRVP_REQUIRE(w && y && dst && nt > 0 && B > 0 && nh + ny <= Cz, "srvp_latent_to_z: bad args");
    const long long n = (long long)nt * B * Cz;
    if (dst_f32) hipLaunchKernelGGL(latent_to_z_kernel<float>, g1(n), dim3(256), 0, (hipStream_t)stream, w, y, (long long)y_tstride, (float*)dst, nt, B, nh, ny, Cz);
    else hipLaunchKernelGGL(latent_to_z_kernel<bf16_t>, g1(n), dim3(256), 0, (hipStream_t)stream, w, y, (long long)y_tstride, (bf16_t*)dst, nt, B, nh, ny, Cz);
    SRVP_CHECK_LAUNCH("srvp_latent_to_z");
    return SRVP_OK;
}
extern "C" int srvp_dz_split(const void* dz, int Cz, int elem_f32, int nt, int B, int nh, int ny, const float* d_w_add, const float* d_y_add,
                             float* d_w, float* d_y, int64_t dy_tstride, void* stream) {
    SRVP_REQUIRE(dz && d_w && d_y && nt > 0 && B > 0 && nh + ny <= Cz, "srvp_dz_split: bad args");
    const long long n = (long long)B * nh + (long long)nt * B * ny;
    if (elem_f32) hipLaunchKernelGGL(dz_split_kernel<float>, g1(n), dim3(256), 0, (hipStream_t)stream, (const float*)dz, Cz, nt, B, nh, ny, d_w_add, d_y_add, d_w, d_y, (long long)dy_tstride);
    else hipLaunchKernelGGL(dz_split_kernel<bf16_t>, g1(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dz, Cz, nt, B, nh, ny, d_w_add, d_y_add, d_w, d_y, (long long)dy_tstride);
    SRVP_CHECK_LAUNCH("srvp_dz_split");
    return SRVP_OK;
}
extern "C" int srvp_rows_scatter_add_f32(float* dst, const void* idx, int idx_is_i64, const float* src, int64_t rows, int C, void* stream) {
    SRVP_REQUIRE(dst && idx && src && C > 0, "srvp_rows_scatter_add_f32: bad args");
    if (rows <= 0) return SRVP_OK;
    hipLaunchKernelGGL(rows_scatter_add_kernel, g1((long long)rows * C), dim3(256), 0, (hipStream_t)stream, dst,
                       idx_is_i64 ? (const long long*)idx : nullptr, idx_is_i64 ? nullptr : (const int*)idx, src, (long long)rows, C);
    SRVP_CHECK_LAUNCH("srvp_rows_scatter_add_f32");
    return SRVP_OK;
}

extern "C" int srvp_axpby_f32(float* out, float a, const float* x, float b, const float* y, int64_t n, void* stream) {
    SRVP_REQUIRE(out && x, "srvp_axpby_f32: null pointer");
    if (n <= 0) return SRVP_OK;
    hipLaunchKernelGGL(axpby_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, out, a, x, b, y, (long long)n);
    SRVP_CHECK_LAUNCH("srvp_axpby_f32");
    return SRVP_OK;
}

extern "C" int srvp_colsum_f32(const float* A, int64_t a_rs, float* out, int M, int N, int accumulate, void* stream) {
    SRVP_REQUIRE(A && out, "srvp_colsum_f32: null pointer");
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * N, (hipStream_t)stream);
        SRVP_REQUIRE(e == hipSuccess, "srvp_colsum_f32: memset failed");
    }
    const int rpb = g_srvp_det ? (M > 0 ? M : 1) : 128;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, A, (long long)a_rs,
                       out, M, N, rpb);
    SRVP_CHECK_LAUNCH("srvp_colsum_f32");
    return SRVP_OK;
}

extern "C" int srvp_act_bwd_f32(const float* v, const float* dy, float* dx, int64_t n, int act, int from_output, void* stream) {
    if (n <= 0) return SRVP_OK;
    hipLaunchKernelGGL(act_bwd_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, v, dy, dx, (long long)n, act, from_output);
    SRVP_CHECK_LAUNCH("srvp_act_bwd_f32");
    return SRVP_OK;
}

extern "C" int srvp_rsample_fwd(const float* params, const float* eps, float* out, int64_t rows, int d, void* stream) {
    if (rows * d <= 0) return SRVP_OK;
    hipLaunchKernelGGL(rsample_fwd_kernel, g1(rows * d), dim3(256), 0, (hipStream_t)stream, params, eps, out, (long long)rows, d);
    SRVP_CHECK_LAUNCH("srvp_rsample_fwd");
    return SRVP_OK;
}
extern "C" int srvp_rsample_bwd(const float* params, const float* eps, const float* dout, float* dparams, int64_t rows, int d,
                                int accumulate, void* stream) {
    if (rows * d <= 0) return SRVP_OK;
    hipLaunchKernelGGL(rsample_bwd_kernel, g1(rows * d), dim3(256), 0, (hipStream_t)stream, params, eps, dout, dparams,
                       (long long)rows, d, accumulate, (long long)2 * d);
    SRVP_CHECK_LAUNCH("srvp_rsample_bwd");
    return SRVP_OK;
}

extern "C" int srvp_lstm_fwd(const float* gates_x, const float* w_hh, float* h_out, float* c_out, float* gates_act, int T, int B,
                             int nh, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SRVP_REQUIRE(gates_x && w_hh && h_out && c_out && gates_act, "srvp_lstm_fwd: null pointer");
    const size_t gs = (size_t)B * 4 * nh, hs = (size_t)B * nh;
    hipError_t e = hipMemcpyAsync(gates_act, gates_x, sizeof(float) * gs * T, hipMemcpyDeviceToDevice, st);
    SRVP_REQUIRE(e == hipSuccess, "srvp_lstm_fwd: copy failed: %s", hipGetErrorString(e));
    for (int t = 0; t < T; ++t) {
        float* g = gates_act + gs * t;
        if (t > 0) {
            int rc = gemm(st, h_out + hs * (t - 1), nh, 1, w_hh, 1, nh, nullptr, g, 4 * nh, B, 4 * nh, nh, ACT_NONE, 1);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(lstm_cell_fwd_kernel, g1((long long)hs), dim3(256), 0, st, g, t > 0 ? c_out + hs * (t - 1) : nullptr,
                           c_out + hs * t, h_out + hs * t, B, nh);
    }
    SRVP_CHECK_LAUNCH("srvp_lstm_fwd");
    return SRVP_OK;
}

// dgates [T][B][4nh]; scratch (2*B*nh floats: dh_carry, dc_carry) is taken from the tail of dgates' allocation by
// the caller: pass `scratch` explicitly.
extern "C" int srvp_lstm_bwd(const float* dh_out, const float* w_hh, const float* c_out, const float* gates_act, float* dgates,
                             float* scratch, int T, int B, int nh, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SRVP_REQUIRE(dh_out && w_hh && c_out && gates_act && dgates && scratch, "srvp_lstm_bwd: null pointer");
    const size_t gs = (size_t)B * 4 * nh, hs = (size_t)B * nh;
    float* dh_carry = scratch;
    float* dc_carry = scratch + hs;
    for (int t = T - 1; t >= 0; --t) {
        const int has_carry = t < T - 1;
        hipLaunchKernelGGL(lstm_cell_bwd_kernel, g1((long long)hs), dim3(256), 0, st, dh_out + hs * t, dh_carry, dc_carry,
                           gates_act + gs * t, c_out + hs * t, t > 0 ? c_out + hs * (t - 1) : nullptr, dgates + gs * t, B, nh,
                           has_carry);
        if (t > 0) {
            // dh_carry = dgates[t] * W_hh      ([B][4nh] x [4nh][nh])
            int rc = gemm(st, dgates + gs * t, 4 * nh, 1, w_hh, nh, 1, nullptr, dh_carry, nh, B, nh, 4 * nh, ACT_NONE, 0);
            if (rc) return rc;
        }
    }
    SRVP_CHECK_LAUNCH("srvp_lstm_bwd");
    return SRVP_OK;
}

extern "C" int srvp_rollout_fwd(const srvp_rollout_desc* d, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SRVP_REQUIRE(d && d->y0 && d->y_all && d->res && d->z && d->p_z_params && d->eps_z && d->inp_all && d->scratch_out,
                 "srvp_rollout_fwd: null pointer");
    SRVP_REQUIRE(d->nl >= 2 && d->nl <= 8 && d->n_euler >= 1, "srvp_rollout_fwd: nl=%d n_euler=%d", d->nl, d->n_euler);
    SRVP_REQUIRE((d->hid_dyn && d->hid_pz) || d->scratch_hid, "srvp_rollout_fwd: no hidden-activation storage");
    SRVP_REQUIRE(!d->pz_external || (d->q_z_params && d->n_data_frames > (d->nsteps + d->n_euler - 1) / d->n_euler),
                 "srvp_rollout_fwd: pz_external needs posterior parameters for every frame");
    const int B = d->B, ny = d->ny, nz = d->nz, nh = d->nh, nl = d->nl;
    const size_t ys = (size_t)B * ny, zs = (size_t)B * nz, hl = (size_t)B * nh;
    const int nin = ny + nz;
    const int F = (d->nsteps + d->n_euler - 1) / d->n_euler;
    if (!d->pz_external && d->fused_ws) {
        // generation chain (posterior while data lasts, prior afterwards) as persistent launches: csrc/rollout_fused.hip rollout_gen_kernel
        const int64_t need = srvp_rollout_gen_ws_bytes(d);
        if (need > 0 && d->fused_ws_bytes >= need) return srvp_rollout_gen_fwd(d, st);
    }
    if (d->pz_external && d->hid_dyn && use_fused(*d)) {
        // the whole posterior-only chain: one preparation launch + one persistent kernel
        const int64_t cw = srvp_rollout_fused_cnt_words(d);
        long long n = (long long)F * zs;
        if (n < (long long)ys) n = (long long)ys;
        if (n < cw) n = cw;
        hipLaunchKernelGGL(rollout_prep_kernel, g1(n), dim3(256), 0, st, d->q_z_params, d->eps_z, d->y0, d->z, d->y_all, d->inp_all,
                           (unsigned*)d->fused_ws, (int)cw, F, d->nsteps, d->n_euler, B, ny, nz);
        return srvp_rollout_fused_fwd(d, st, true);
    }
    hipError_t e = hipMemcpyAsync(d->y_all, d->y0, sizeof(float) * ys, hipMemcpyDeviceToDevice, st);
    SRVP_REQUIRE(e == hipSuccess, "srvp_rollout_fwd: copy failed");
    if (d->pz_external && d->hid_dyn) {
        // posterior-only chain: all samples and the z halves of every MLP input up front, then nl GEMMs + one update per step
        hipLaunchKernelGGL(rsample_fwd_kernel, g1((long long)F * zs), dim3(256), 0, st, d->q_z_params, d->eps_z, d->z, (long long)F * B, nz);
        hipLaunchKernelGGL(fill_inp_kernel, g1((long long)d->nsteps * B * nin), dim3(256), 0, st, d->z, d->y0, d->inp_all, d->nsteps,
                           d->n_euler, B, ny, nz);
        for (int i = 0; i < d->nsteps; ++i) {
            float* inp = d->inp_all + (size_t)i * B * nin;
            int rc = mlp_fwd(st, d->dyn_w, d->dyn_b, nl, nin, nh, ny, inp, B, d->hid_dyn + (size_t)i * hl, (size_t)d->nsteps * hl, d->scratch_out);
            if (rc) return rc;
            hipLaunchKernelGGL(euler_update2_kernel, g1((long long)ys), dim3(256), 0, st, d->y_all + ys * i, d->scratch_out, d->dt,
                               d->res + ys * i, d->y_all + ys * (i + 1), i + 1 < d->nsteps ? inp + (size_t)B * nin : (float*)nullptr, B, ny, nin);
        }
        SRVP_CHECK_LAUNCH("srvp_rollout_fwd");
        return SRVP_OK;
    }
    for (int i = 0; i < d->nsteps; ++i) {
        const int f = i / d->n_euler;             // 0-based frame slot (frame index f+1)
        const float* y_prev = d->y_all + ys * i;
        if (i % d->n_euler == 0) {
            float* pz = d->p_z_params + (size_t)f * B * 2 * nz;
            float* hid = d->hid_pz ? d->hid_pz + (size_t)f * hl : d->scratch_hid;
            size_t ls = d->hid_pz ? (size_t)F * hl : hl;
            if (!d->pz_external) {
                int rc = mlp_fwd(st, d->pz_w, d->pz_b, nl, ny, nh, 2 * nz, y_prev, B, hid, ls, pz);
                if (rc) return rc;
            }
            const bool posterior = (f + 1) < d->n_data_frames;
            const float* params = posterior ? d->q_z_params + (size_t)f * B * 2 * nz : pz;
            hipLaunchKernelGGL(rsample_fwd_kernel, g1((long long)zs), dim3(256), 0, st, params, d->eps_z + zs * f, d->z + zs * f,
                               (long long)B, nz);
        }
        float* inp = d->inp_all + (size_t)i * B * nin;
        hipLaunchKernelGGL(concat_yz_kernel, g1((long long)B * nin), dim3(256), 0, st, y_prev, d->z + zs * f, inp, B, ny, nz);
        float* hid = d->hid_dyn ? d->hid_dyn + (size_t)i * hl : d->scratch_hid;
        size_t ls = d->hid_dyn ? (size_t)d->nsteps * hl : hl;
        int rc = mlp_fwd(st, d->dyn_w, d->dyn_b, nl, nin, nh, ny, inp, B, hid, ls, d->scratch_out);
        if (rc) return rc;
        hipLaunchKernelGGL(euler_update_kernel, g1((long long)ys), dim3(256), 0, st, y_prev, d->scratch_out, d->dt, d->res + ys * i,
                           d->y_all + ys * (i + 1), (int)ys);
    }
    SRVP_CHECK_LAUNCH("srvp_rollout_fwd");
    return SRVP_OK;
}

extern "C" int srvp_rollout_bwd(const srvp_rollout_bwd_desc* d, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SRVP_REQUIRE(d, "srvp_rollout_bwd: null descriptor");
    const srvp_rollout_desc& f = d->f;
    SRVP_REQUIRE(d->d_y_all && d->d_y0 && d->dhid_dyn && d->dhid_pz && d->work && f.hid_dyn && f.hid_pz && d->d_qz,
                 "srvp_rollout_bwd: null pointer");
    const int B = f.B, ny = f.ny, nz = f.nz, nh = f.nh, nl = f.nl, nin = ny + nz;
    const size_t ys = (size_t)B * ny, zs = (size_t)B * nz, hl = (size_t)B * nh;
    const int F = (f.nsteps + f.n_euler - 1) / f.n_euler;
    const int dwd = nh > ny ? nh : ny;             // delta row width (dynamics)
    const int dwp = nh > 2 * nz ? nh : 2 * nz;     // delta row width (p_z)
    const size_t dls_d = (size_t)f.nsteps * B * dwd, dls_p = (size_t)F * B * dwp;
    // work: carry[ys] | dy[ys] | dinp[B*nin] | dz_acc[zs] | dypz[ys]
    float* carry = d->work;
    float* dy = carry + ys;
    float* dinp = dy + ys;
    float* dz_acc = dinp + (size_t)B * nin;
    float* dypz = dz_acc + zs;
    if (f.pz_external && d->dinp_all && use_fused(f)) {
        if (int rc = srvp_rollout_fused_bwd(d, st)) return rc;        // writes every delta, dinp_all and d_y0 (the carry lives in its LDS)
        hipLaunchKernelGGL(dz_finalize_kernel, g1((long long)F * zs), dim3(256), 0, st, d->dinp_all, d->d_z, f.q_z_params, f.eps_z, d->d_qz,
                           F, f.nsteps, f.n_euler, B, ny, nz);
        SRVP_CHECK_LAUNCH("srvp_rollout_bwd(fused)");
        return SRVP_OK;
    }
    hipError_t e = hipMemsetAsync(carry, 0, sizeof(float) * ys, st);
    SRVP_REQUIRE(e == hipSuccess, "srvp_rollout_bwd: memset failed");
    if (f.pz_external && d->dinp_all) {
        // posterior-only chain: seed of the last step, then per step nl GEMMs + ONE kernel (close step i, seed step i-1); the
        // gradients wrt z and the posterior samples' backward are finished for all frames at once afterwards
        const int S = f.nsteps;
        hipLaunchKernelGGL(euler_bwd_seed_kernel, g1((long long)ys), dim3(256), 0, st, d->d_y_all + ys * S, carry,
                           d->d_res ? d->d_res + ys * (S - 1) : nullptr, f.dt, dy, d->dhid_dyn + (size_t)(S - 1) * B * dwd + (size_t)(nl - 1) * dls_d,
                           B, ny, dwd);
        for (int i = S - 1; i >= 0; --i) {
            float* deltas = d->dhid_dyn + (size_t)i * B * dwd;
            float* dinp_i = d->dinp_all + (size_t)i * B * nin;
            int rc = mlp_bwd(st, f.dyn_w, nl, nin, nh, ny, B, f.hid_dyn + (size_t)i * hl, (size_t)S * hl, deltas, dls_d, dwd, dinp_i);
            if (rc) return rc;
            hipLaunchKernelGGL(euler_bwd_chain_kernel, g1((long long)ys), dim3(256), 0, st, dy, dinp_i, carry, d->d_y_all + ys * i,
                               (i > 0 && d->d_res) ? d->d_res + ys * (i - 1) : (const float*)nullptr, f.dt,
                               i > 0 ? d->dhid_dyn + (size_t)(i - 1) * B * dwd + (size_t)(nl - 1) * dls_d : (float*)nullptr, B, ny, nin, dwd);
        }
        hipLaunchKernelGGL(dz_finalize_kernel, g1((long long)F * zs), dim3(256), 0, st, d->dinp_all, d->d_z, f.q_z_params, f.eps_z, d->d_qz,
                           F, S, f.n_euler, B, ny, nz);
        hipLaunchKernelGGL(add3_kernel, g1((long long)ys), dim3(256), 0, st, d->d_y0, d->d_y_all, carry, (int)ys);
        SRVP_CHECK_LAUNCH("srvp_rollout_bwd");
        return SRVP_OK;
    }
    for (int i = f.nsteps - 1; i >= 0; --i) {
        const int fr = i / f.n_euler;
        const bool last_sub = (i % f.n_euler) == f.n_euler - 1 || i == f.nsteps - 1;   // first visited sub-step of the frame
        const bool first_sub = (i % f.n_euler) == 0;
        float* deltas = d->dhid_dyn + (size_t)i * B * dwd;             // layer l at + l*dls_d
        hipLaunchKernelGGL(euler_bwd_seed_kernel, g1((long long)ys), dim3(256), 0, st, d->d_y_all + ys * (i + 1), carry,
                           d->d_res ? d->d_res + ys * i : nullptr, f.dt, dy, deltas + (size_t)(nl - 1) * dls_d, B, ny, dwd);
        int rc = mlp_bwd(st, f.dyn_w, nl, nin, nh, ny, B, f.hid_dyn + (size_t)i * hl, (size_t)f.nsteps * hl, deltas, dls_d, dwd, dinp);
        if (rc) return rc;
        hipLaunchKernelGGL(euler_bwd_split_kernel, g1((long long)B * nin), dim3(256), 0, st, dy, dinp, carry, dz_acc, B, ny, nz,
                           last_sub ? 1 : 0);
        if (first_sub) {
            // gradient wrt z of this frame is complete
            if (d->d_z) hipLaunchKernelGGL(add_inplace_kernel, g1((long long)zs), dim3(256), 0, st, dz_acc, d->d_z + zs * fr, (int)zs);
            const bool posterior = (fr + 1) < f.n_data_frames;
            if (f.pz_external) {
                // p_z backward is batched by the caller (its input gradient is already in d_y_all): only the sample's
                // gradient wrt the posterior parameters is part of the chain
                hipLaunchKernelGGL(rsample_bwd_kernel, g1((long long)zs), dim3(256), 0, st, f.q_z_params + (size_t)fr * B * 2 * nz,
                                   f.eps_z + zs * fr, dz_acc, d->d_qz + (size_t)fr * B * 2 * nz, (long long)B, nz, 0, (long long)2 * nz);
                continue;
            }
            float* pdel = d->dhid_pz + (size_t)fr * B * dwp;
            float* pout = pdel + (size_t)(nl - 1) * dls_p;
            hipLaunchKernelGGL(rows_copy_kernel, g1((long long)B * 2 * nz), dim3(256), 0, st, pout, dwp,
                               d->d_pz ? d->d_pz + (size_t)fr * B * 2 * nz : (const float*)nullptr, B, 2 * nz);
            if (posterior)
                hipLaunchKernelGGL(rsample_bwd_kernel, g1((long long)zs), dim3(256), 0, st, f.q_z_params + (size_t)fr * B * 2 * nz,
                                   f.eps_z + zs * fr, dz_acc, d->d_qz + (size_t)fr * B * 2 * nz, (long long)B, nz, 0, (long long)2 * nz);
            else
                hipLaunchKernelGGL(rsample_bwd_kernel, g1((long long)zs), dim3(256), 0, st, f.p_z_params + (size_t)fr * B * 2 * nz,
                                   f.eps_z + zs * fr, dz_acc, pout, (long long)B, nz, 1, (long long)dwp);
            rc = mlp_bwd(st, f.pz_w, nl, ny, nh, 2 * nz, B, f.hid_pz + (size_t)fr * hl, (size_t)F * hl, pdel, dls_p, dwp, dypz);
            if (rc) return rc;
            hipLaunchKernelGGL(add_inplace_kernel, g1((long long)ys), dim3(256), 0, st, carry, dypz, (int)ys);
        }
    }
    hipLaunchKernelGGL(add3_kernel, g1((long long)ys), dim3(256), 0, st, d->d_y0, d->d_y_all, carry, (int)ys);
    SRVP_CHECK_LAUNCH("srvp_rollout_bwd");
    return SRVP_OK;
}
