// Tap-table implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16).
//
//   out[m][j] = sum_t sum_c A_t[m][c] * W[t][j][c]        m = (n, oy, ox),  j = output channel
//
// One workgroup computes a BM x BN output tile; the K loop walks (tap, BK-channel chunk).  A rows are gathered
// straight from the NHWC bf16 activation tensor(s) (16-byte = 8-channel pieces, zero border physically present in
// memory so no bounds predicates), B rows from the tap-major packed weights.  Both tiles go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR staging, no ds_write pass) into XOR-swizzled rows (conflict-free ds_read_b128
// fragment reads), double-buffered with the DMA of step s+1 issued before the MFMAs of step s (one barrier per K step).  The fp32 accumulators feed (a) fp64-atomic
// per-channel sum / sum-of-squares for BatchNorm batch statistics and (b) a bf16 LDS-staged, 16-byte-wide store.
//
// Replaces: nn.Conv2d / nn.ConvTranspose2d forward + data-gradient (reference module/conv.py:174-179, 200-223,
// 299-304, 330-353), torch.cat of the skip connection (conv.py:270), nn.Upsample (conv.py:331-349).
#include "common.h"
#include "../../include/srvp_hip.h"

int srvp_conv_f32_launch(const srvp_conv_desc* d, hipStream_t st);     // conv_f32.hip (precision = 'fp32' parity mode)
int srvp_conv_stream64_launch(const srvp_conv_desc* d, hipStream_t st, int* taken);
int srvp_conv_stream_sub64_launch(const srvp_conv_desc* d, int n, hipStream_t st, int* taken);   // (four sub-pixel phases at 64 channels)     // conv_stream.hip (64 -> 64 channels at 64x64: streaming kernel)

namespace {

struct RowInfo { int n, oy, ox; };

// Compact kernel-argument block (no arrays: tap offsets are packed 4 bits each so nothing is dynamically indexed
// in private memory).
struct ConvK {
    const bf16_t* src0; const bf16_t* src1; const int* map1;
    int C0, C1, H0p, W0p, H1p, W1p, ups0, ups1, si, ntaps;
    unsigned long long dy_bits, dx_bits;
    const bf16_t* wt; int Cout, N, OH, OW;
    bf16_t* dst; int DHp, DWp, so, ooy, oox, Cdst, cdst_off;
    double* stats; int stat_mod;
    float* out_f32; int out_nc, out_sigmoid;
    const int* map0; int dst_is_f32; const float* add_f32; int add_mod;
    int phase_chunks;          // > 0: space-to-depth source, taps of chunk cc are entries [(cc / phase_chunks) * ntaps + t]
    int f32_quad;              // > 0: the fp32 tensor (dst of a dst_is_f32 launch / add_f32) is pixel-quad-major for consumer stride f32_quad
    int splitk;                // > 1 (generic kernel, fp32 destination): blockIdx.y walks its share of the K steps into slab blockIdx.y
    long long slab;            // elements per split-K slab
    const bf16_t* bnr_raw; const float* bnr_coef; double* bnr_red;     // fused BatchNorm-backward reduction of the producer (srvp_hip.h)
    const float* ep_coef; int ep_act, ep_border;                       // eval-mode BatchNorm + activation in the epilogue (srvp_hip.h)
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Accumulator start values: zero, or (hoisted skip half) conv_s(skip) of the frame's sample -- the fp32 S tensor of a
// dst_is_f32 launch.  Loading it HERE puts the latency of these reads under the first tile DMA instead of exposing it
// in the epilogue.
template <int WM, int WN, int TM, int TN, class RowMap, class SampleOf>
__device__ __forceinline__ void conv_acc_init(f32x16_t (&acc)[TM][TN], const ConvK& a, int n0, const RowMap& rowmap,
                                              const SampleOf& sample_of) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int lcol = lane & 31, lhalf = lane >> 5;
    const int hw = a.OH * a.OW;
    if (!a.add_f32) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        return;
    }
    if (a.f32_quad > 0) {
        // quad-major S (see srvp_conv_desc.f32_quad): the four accumulator registers 4g .. 4g+3 of a lane are four consecutive output
        // columns of one row -> ONE 16-byte load instead of four 4-byte loads (the S tile is the largest stream of a K = 4 C0
        // sub-pixel launch: 128 KB per workgroup against 83 KB of patch and 64 KB of output)
        const int wq = a.DWp / a.f32_quad / 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int n = 0, oy = 0, ox = 0;
                const bool ok = rowmap(wm * (TM * 32) + i * 32 + 8 * g + 4 * lhalf, n, oy, ox);
                const float* sp = a.add_f32 + (((((size_t)sample_of(n) * a.DHp + oy * a.so + a.ooy) * a.f32_quad + a.oox) * wq + (ox >> 2)) * a.Cout +
                                               n0 + wn * (TN * 32) + lcol) * 4;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
                    if (ok) v = *reinterpret_cast<const f32x4_t*>(sp + j * 128);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = v[e];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int n = 0, oy = 0, ox = 0;
            const bool ok = rowmap(wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf, n, oy, ox);
            // S has the layout of the (unbordered) destination: [sample][DHp][DWp][Cout] at the strided output position
            const float* sp = a.add_f32 + (((size_t)sample_of(n) * a.DHp + oy * a.so + a.ooy) * a.DWp + ox * a.so + a.oox) * a.Cout + n0 + wn * (TN * 32) + lcol;
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j][r] = ok ? sp[j * 32] : 0.f;
        }
}

// Shared epilogue of the MFMA convolution kernels.  `rowmap(row, n, oy, ox)` decodes tile row -> output pixel and
// returns false for rows beyond the problem (masked).  smem is reused for the bf16 staging tile [BM][BN + 8]; the
// caller has synchronised the workgroup after its last read of smem.
// C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int BM, int BN, int WM, int WN, int TM, int TN, class RowMap>
__device__ __forceinline__ void conv_epilogue(f32x16_t (&acc)[TM][TN], const ConvK& a, unsigned char* smem, float* red, int n0,
                                              const RowMap& rowmap, bool all_valid = false) {
    constexpr int NT = WM * WN * 64;
    constexpr int LDC = BN + 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int lcol = lane & 31, lhalf = lane >> 5;
    const int hw = a.OH * a.OW;
    if (a.dst_is_f32) {
        float* dstf = reinterpret_cast<float*>(a.dst) + (a.splitk > 1 ? (size_t)blockIdx.y * a.slab : (size_t)0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int n, oy, ox;
                if (!rowmap(wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf, n, oy, ox)) continue;
                if (a.f32_quad > 0) {
                    // element (n, Y, X, c) at ((((n DHp + Y) q + X % q) (DWp / q / 4) + (X / q) / 4) C + c) 4 + (X / q) % 4
                    const int Y = oy * a.so + a.ooy, X = ox * a.so + a.oox, xl = X / a.f32_quad, ph = X - xl * a.f32_quad;
                    float* dq = dstf + (((((size_t)n * a.DHp + Y) * a.f32_quad + ph) * (a.DWp / a.f32_quad / 4) + (xl >> 2)) * a.Cdst + a.cdst_off + n0 +
                                        wn * (TN * 32) + lcol) * 4 + (xl & 3);
#pragma unroll
                    for (int j = 0; j < TN; ++j) dq[j * 128] = acc[i][j][r];
                    continue;
                }
                float* dp = dstf + (((size_t)n * a.DHp + oy * a.so + a.ooy) * a.DWp + ox * a.so + a.oox) * a.Cdst + a.cdst_off + n0 +
                            wn * (TN * 32) + lcol;
#pragma unroll
                for (int j = 0; j < TN; ++j) dp[j * 32] = acc[i][j][r];
            }
        return;
    }
    if (a.stats && all_valid) {
        // every row of the tile is a real output pixel (workgroup-uniform): no per-row masks
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s1 += v; s2 += v * v; }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lhalf == 0) {
                int col = wn * (TN * 32) + j * 32 + lcol;
                red[(wm * BN + col) * 2 + 0] = s1;
                red[(wm * BN + col) * 2 + 1] = s2;
            }
        }
    } else if (a.stats) {
        bool ok[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int n, oy, ox;
                ok[i][r] = rowmap(wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf, n, oy, ox);
            }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = ok[i][r] ? acc[i][j][r] : 0.f;
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
    if (a.out_f32) {
        // image-side output layer: the first out_nc columns are the frame channels; sigmoid (reference
        // conv.py:273-274) and a store into the fp32 (N, C, H, W) frame tensor of module/srvp.py:226.
        // Only out_nc (<= 4) of the 32 columns are real: the few lanes that hold them park their values in LDS [row][4], then
        // EVERY thread finishes whole pixels (one row decode, out_nc sigmoids and stores, consecutive threads on consecutive
        // pixels) -- instead of three lanes per half-wave walking 16 * TM rows each with the rest of the wave idle.
        float* Os = reinterpret_cast<float*>(smem);
        if (lcol < 4 && wn == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Os[(wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * 4 + lcol] = acc[i][0][r];
        }
        __syncthreads();
        for (int row = tid; row < BM; row += NT) {
            int n, oy, ox;
            if (!rowmap(row, n, oy, ox)) continue;
            const f32x4_t o4 = *reinterpret_cast<const f32x4_t*>(Os + row * 4);
            float* op = a.out_f32 + ((size_t)n * a.out_nc * a.DHp + oy * a.so + a.ooy) * a.DWp + ox * a.so + a.oox;
            const size_t cs = (size_t)a.DHp * a.DWp;
            for (int c = 0; c < a.out_nc; ++c) {
                float v = o4[c];
                if (a.out_sigmoid) v = 1.f / (1.f + __expf(-v));
                op[c * cs] = v;
            }
        }
        return;
    }
    if (a.ep_coef) {
        // inference: y = act(scale * acc + shift) straight from the fp32 accumulators (srvp_conv_desc.ep_*)
        float sc[TN], sh[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = n0 + wn * (TN * 32) + j * 32 + lcol;
            sc[j] = a.ep_coef[c]; sh[j] = a.ep_coef[a.Cout + c];
        }
        if (a.ep_act == ACT_LRELU) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float v = __builtin_fmaf(acc[i][j][r], sc[j], sh[j]); acc[i][j][r] = v > 0.f ? v : LRELU_SLOPE * v; }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = act_fwd(__builtin_fmaf(acc[i][j][r], sc[j], sh[j]), a.ep_act);
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
    // copy-out: thread -> (16-byte channel chunk ch, rows row0 + k * RSTEP).  Unrolled in groups of four with the LDS reads
    // of a group issued before its stores and the descriptor fields hoisted (as a rolled loop hipcc re-loaded them from the
    // kernel arguments and serialised LDS read -> store every iteration: ~500 cycles x 16 iterations per tile, more than
    // the MFMA time of a K = 576 tile)
    constexpr int CCH = BN / 8;
    static_assert(NT % CCH == 0 && (BM * CCH) % NT == 0, "copy-out tiling");
    constexpr int RSTEP = NT / CCH, ITER = BM * CCH / NT, G = ITER % 4 == 0 ? 4 : (ITER % 2 == 0 ? 2 : 1);
    const int ch = tid % CCH, row0 = tid / CCH;
    // (ep_coef launches: dst is the consumer's activation tensor with an ep_border-pixel zero border around the DHp x DWp interior)
    const int eb = a.ep_coef ? a.ep_border : 0;
    const int so = a.so, ooy = a.ooy + eb, oox = a.oox + eb, DWp = a.DWp + 2 * eb, Cdst = a.Cdst;
    const size_t img = (size_t)(a.DHp + 2 * eb) * DWp * Cdst;
    bf16_t* dbase = a.dst + a.cdst_off + n0 + ch * 8;
    const bf16_t* cbase = Cs + row0 * LDC + ch * 8;
    constexpr bool BNR_FITS = RSTEP * BN * 2 * 4 <= BM * LDC * 2;     // (all 256-pixel variants; the launcher sends nothing else here)
    if constexpr (BNR_FITS) if (a.bnr_red) {
        // ---- fused BatchNorm-backward reduction of the PRODUCER layer (srvp_conv_desc.bnr_*): this launch's output IS dA of that
        // layer.  The copy-out loop already walks the tile as (pixel row, 16-byte channel chunk): beside the bf16 dA piece it reads
        // from the staging area, a thread loads the matching piece of the producer's raw output straight from global memory (same
        // offset as the store: both tensors are [N][H][W][C], one coalesced 256-byte run per pixel), forms g = dA * lrelu'(scale raw
        // + shift) on the values AS STORED (what srvp_bn_bwd_apply reads back) and keeps sum g, sum g * xhat of its eight channels in
        // registers; the RSTEP row groups are then summed through LDS and added with one fp64 atomic pair per channel and workgroup.
        const bf16_t* rbase = a.bnr_raw + n0 + ch * 8;
        float csc[8], csh[8], cmu[8], cis[8], s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = n0 + ch * 8 + e;
            csc[e] = a.bnr_coef[c]; csh[e] = a.bnr_coef[Cdst + c]; cmu[e] = a.bnr_coef[2 * Cdst + c]; cis[e] = a.bnr_coef[3 * Cdst + c];
            s1[e] = 0.f; s2[e] = 0.f;
        }
        // (the raw pieces of group g + 1 are requested before group g is worked on: issued inside their own group they cost every group a
        // full global-load latency, four times per 128-column tile)
        auto piece = [&](int gi, size_t& o, bool& k) {
            const int row = row0 + gi * RSTEP;
            int n, oy, ox;
            k = rowmap(row, n, oy, ox);
            if (!k) { n = 0; oy = 0; ox = 0; }
            o = (size_t)n * img + (size_t)(unsigned)(((oy * so + ooy) * DWp + ox * so + oox) * Cdst);
        };
        u32x4_t rwn[G];
        size_t offn[G];
        bool okn[G];
#pragma unroll
        for (int u = 0; u < G; ++u) { piece(u, offn[u], okn[u]); rwn[u] = *reinterpret_cast<const u32x4_t*>(rbase + offn[u]); }
#pragma unroll
        for (int g = 0; g < ITER; g += G) {
            u32x4_t v[G], rw[G];
            size_t off[G];
            bool ok[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                off[u] = offn[u]; ok[u] = okn[u]; rw[u] = rwn[u];
                v[u] = *reinterpret_cast<const u32x4_t*>(cbase + (g + u) * RSTEP * LDC);
            }
            if (g + G < ITER) {
#pragma unroll
                for (int u = 0; u < G; ++u) { piece(g + G + u, offn[u], okn[u]); rwn[u] = *reinterpret_cast<const u32x4_t*>(rbase + offn[u]); }
            }
#pragma unroll
            for (int u = 0; u < G; ++u) {
                if (ok[u]) *reinterpret_cast<u32x4_t*>(dbase + off[u]) = v[u];
                float da[8], rv[8];
                unpack8(v[u], da);
                unpack8(rw[u], rv);
                const float m = ok[u] ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gg = m * da[e] * ((rv[e] * csc[e] + csh[e]) > 0.f ? 1.f : LRELU_SLOPE);
                    s1[e] += gg; s2[e] += gg * (rv[e] - cmu[e]) * cis[e];
                }
            }
        }
        __syncthreads();                                   // every thread is done with the staging tile: it now holds the partial sums
        float* Ps = reinterpret_cast<float*>(smem);        // [RSTEP][BN][2]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            Ps[(row0 * BN + ch * 8 + e) * 2 + 0] = s1[e];
            Ps[(row0 * BN + ch * 8 + e) * 2 + 1] = s2[e];
        }
        __syncthreads();
        if (tid < BN) {
            double t1 = 0., t2 = 0.;
#pragma unroll
            for (int r = 0; r < RSTEP; ++r) { t1 += Ps[(r * BN + tid) * 2]; t2 += Ps[(r * BN + tid) * 2 + 1]; }
            atomicAdd(a.bnr_red + n0 + tid, t1);
            atomicAdd(a.bnr_red + Cdst + n0 + tid, t2);
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < ITER; g += G) {
        u32x4_t v[G];
        size_t off[G];
        bool ok[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int row = row0 + (g + u) * RSTEP;
            int n, oy, ox;
            ok[u] = rowmap(row, n, oy, ox);
            off[u] = (size_t)n * img + (size_t)(unsigned)(((oy * so + ooy) * DWp + ox * so + oox) * Cdst);
            v[u] = *reinterpret_cast<const u32x4_t*>(cbase + (g + u) * RSTEP * LDC);
        }
#pragma unroll
        for (int u = 0; u < G; ++u)
            if (ok[u]) *reinterpret_cast<u32x4_t*>(dbase + off[u]) = v[u];
    }
}


// BDIRECT (single-buffer BK = 64 variant only): the weights are packed MFMA-fragment-major and go from L2 straight into
// registers, like in the halo kernel below -- only the gathered A rows use the (scarce) LDS-DMA path and the LDS.
template <int BM, int BN, int BK, int WM, int WN, int NBUF, bool BDIRECT = false>
__global__ __launch_bounds__(WM * WN * 64) void conv_mfma_kernel(const ConvK a) {
    constexpr int NT = WM * WN * 64;
    constexpr int CPR = BK / 8;                 // 16-byte chunks per tile row
    constexpr int RPB = 16 / CPR;               // tile rows per 256-byte LDS bank row
    constexpr int A_LD = (BM * CPR + NT - 1) / NT;
    constexpr int B_LD = (BN * CPR + NT - 1) / NT;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDC = BN + 8;
    static_assert((BM * CPR) % NT == 0 && (BN * CPR) % 64 == 0, "tiles must split into whole-wave 1 KiB LDS-DMA pieces");
    static_assert(!BDIRECT || (NBUF == 1 && BK == 64), "BDIRECT: single-buffer BK = 64 variant");
    constexpr int G = A_LD + B_LD;              // LDS-DMA instructions per K step and wave (vmcnt is per wave)
    constexpr int AB_BYTES = NBUF * (BM + (BDIRECT ? 0 : BN)) * BK * 2;
    constexpr int C_BYTES = BM * LDC * 2;
    constexpr int SMEM = AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES;
    // ONE shared object (a second one makes hipcc drain the LDS-DMA queue before every ds_read)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM + WM * BN * 8];
    bf16_t* As = reinterpret_cast<bf16_t*>(smem);                         // [NBUF][BM][BK]  (XOR-swizzled 16-B chunks)
    bf16_t* Bs = As + NBUF * BM * BK;                                     // [NBUF][BN][BK]
    float* red = reinterpret_cast<float*>(smem + SMEM);                   // [WM][BN][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int Ctot = a.C0 + a.C1;
    const long long M = (long long)a.N * a.OH * a.OW;
    // logical workgroup index -> (m tile, n tile): n fastest and XCD-contiguous, so the Cout/BN workgroups that gather
    // the same A rows (and their spatial neighbours, which share the 3x3 halo) run on the same XCD / L2
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int n_tiles = a.Cout / BN;
    const long long m0 = (long long)(lb / n_tiles) * BM;
    const int n0 = (lb % n_tiles) * BN;

    // ---- per-thread gather rows (fixed over the K loop).  Tiles go global -> LDS by LDS-DMA (global_load_lds,
    // 16 B per lane, 1 KiB per wave instruction, destination lane-linear), so the bank-conflict swizzle is applied to
    // the SOURCE chunk index: LDS position (row, p) holds data chunk p ^ f(row), f(row) = (row / RPB) % CPR.
    RowInfo ri[A_LD];
    int a_ch[A_LD], n1[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        int q = tid + i * NT;
        int row = q / CPR;
        a_ch[i] = (q % CPR) ^ ((row / RPB) % CPR);
        long long m = m0 + row;
        if (m >= M) m = M - 1;                       // clamp: duplicates are masked at the store
        int hw = a.OH * a.OW;
        int n = (int)(m / hw);
        int r = (int)(m - (long long)n * hw);
        ri[i].n = n; ri[i].oy = r / a.OW; ri[i].ox = r - (r / a.OW) * a.OW;
        // image index inside the second source (skip connection), resolved ONCE: an ordinary VGPR load inside the K
        // loop would make hipcc drain the whole LDS-DMA queue (s_waitcnt vmcnt(0)) at every use
        n1[i] = (a.C1 > 0 && a.map1) ? a.map1[n] : n;
        if (a.map0) ri[i].n = a.map0[n];             // src0 indirection (hoisted skip half: images = samples)
    }
    // K order: channel chunk OUTER, tap INNER -- consecutive K steps re-read the same channel chunk at the 9 (16)
    // shifted pixel positions, i.e. mostly the same cache lines (reuse distance BM*BK*2 B = 16 KiB per workgroup),
    // instead of coming back to a pixel one whole tap (BM*Ctot*2 B) later (30 % L2 misses measured, profiles/).
    const int kpt = Ctot / BK;                       // channel chunks
    const int S_all = a.ntaps * kpt;
    // split-K (tiny-M, long-K launches: the 4x4 -> 1x1 layer and the data-gradient of its mirror image are 128 dependent K steps
    // on a handful of workgroups otherwise): blockIdx.y owns K steps [s_lo, S) and its own fp32 slab, summed in a fixed order by
    // srvp_splitk_finish (deterministic)
    const int s_lo = a.splitk > 1 ? (int)((long long)S_all * blockIdx.y / a.splitk) : 0;
    const int S = a.splitk > 1 ? (int)((long long)S_all * (blockIdx.y + 1) / a.splitk) : S_all;

    auto stage = [&](int s, int buf) {
        const int cc = s / a.ntaps;
        const int t = s - cc * a.ntaps;
        int c = cc * BK;
        const bf16_t* src; int C, Hp, Wp, ups; bool second = c >= a.C0;
        if (!second) { src = a.src0; C = a.C0; Hp = a.H0p; Wp = a.W0p; ups = a.ups0 ? 1 : 0; }
        else { src = a.src1; C = a.C1; Hp = a.H1p; Wp = a.W1p; ups = a.ups1 ? 1 : 0; c -= a.C0; }
        const int dy = (int)((a.dy_bits >> (4 * t)) & 15), dx = (int)((a.dx_bits >> (4 * t)) & 15);
        bf16_t* Ad = As + (size_t)buf * BM * BK + (size_t)wid * 64 * 8;       // wave-uniform LDS base of this wave's piece
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            int vy = ri[i].oy * a.si + dy, vx = ri[i].ox * a.si + dx;
            vy = (vy + ups) >> ups; vx = (vx + ups) >> ups;                   // nearest x2 upsample: (u + 1) >> 1
            const int n = second ? n1[i] : ri[i].n;
            // 32-bit element offsets (the launcher checks the tensors have fewer than 2^32 elements)
            unsigned off = (((unsigned)n * Hp + vy) * Wp + vx) * C + c + a_ch[i] * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + off), (lptr_t)(Ad + (size_t)i * NT * 8), 16, 0, 0);
        }
        if constexpr (BDIRECT) return;
        const bf16_t* w = a.wt + ((size_t)t * a.Cout + n0) * Ctot + cc * BK;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            // narrow weight tiles are re-loaded by the upper waves (same bytes, same LDS address) so that every wave
            // issues exactly G DMAs per step
            int q = (tid + i * NT) % (BN * CPR);
            int row = q / CPR, ch = (q % CPR) ^ ((row / RPB) % CPR);
            const int qb = ((wid * 64 + i * NT) % (BN * CPR)) * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)(w + (size_t)row * Ctot + ch * 8), (lptr_t)(Bs + (size_t)buf * BN * BK + qb), 16, 0, 0);
        }
    };

    const int hw_ = a.OH * a.OW;
    auto rowmap = [&](int row, int& n, int& oy, int& ox) -> bool {
        long long m = m0 + row;
        if (m >= M) return false;
        n = (int)(m / hw_);
        int r = (int)(m - (long long)n * hw_);
        oy = r / a.OW; ox = r - oy * a.OW;
        return true;
    };
    f32x16_t acc[TM][TN];
    conv_acc_init<WM, WN, TM, TN>(acc, a, n0, rowmap, [&](int n) { return n % a.add_mod; });

    const int lrow = lane & 31, lkc = lane >> 5;
    // fragment rows of this lane and their swizzle terms
    int a_off[TM], a_sw[TM], b_off[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { int r = wm * (TM * 32) + i * 32 + lrow; a_off[i] = r * BK; a_sw[i] = (r / RPB) % CPR; }
#pragma unroll
    for (int j = 0; j < TN; ++j) { int r = wn * (TN * 32) + j * 32 + lrow; b_off[j] = r * BK; b_sw[j] = (r / RPB) % CPR; }
    u32x4_t bq[4][TN];
    const int JT = a.Cout >> 5;
    const bf16_t* wlane = a.wt + (size_t)((n0 >> 5) + wn * TN) * 512 + lane * 8;
    // NBUF-deep LDS ring: the DMA runs NBUF-1 K steps ahead of the MFMAs behind COUNTED s_waitcnt vmcnt and ONE raw
    // s_barrier per step (a __syncthreads() would drain the whole DMA queue: vmcnt(0)).
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (s_lo + i < S) stage(s_lo + i, (s_lo + i) % NBUF);
    for (int s = s_lo; s < S; ++s) {
        const int buf = s % NBUF;
        if constexpr (NBUF == 1) {
            // single buffer, latency hidden by the other workgroups of the CU (3-4 resident at 34 KB of LDS each)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (BDIRECT) {
                // all B fragments of this step (4 k slices x TN column tiles, 1 KiB each), in flight with the A DMA
                const int cc = s / a.ntaps, t = s - cc * a.ntaps;
                const bf16_t* wb = wlane + (size_t)((t * kpt + cc) * 4) * JT * 512;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < TN; ++j) bq[kk][j] = *reinterpret_cast<const u32x4_t*>(wb + ((size_t)kk * JT + j) * 512);
            }
            stage(s, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            if (s + NBUF - 2 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (NBUF > 1 ? NBUF - 2 : 0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + NBUF - 1 < S) stage(s + NBUF - 1, (s + NBUF - 1) % NBUF);
        }
        const bf16_t* Ab = As + (size_t)buf * BM * BK;
        const bf16_t* Bb = Bs + (size_t)buf * BN * BK;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8_t af[TM], bfr[TN];
            const int kc = kk * 2 + lkc;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(Ab + a_off[i] + ((kc ^ a_sw[i]) * 8));
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (BDIRECT) bfr[j] = __builtin_bit_cast(bf16x8_t, bq[kk][j]);
                else bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bb + b_off[j] + ((kc ^ b_sw[j]) * 8));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();                                 // every wave is done reading the ring: the epilogue reuses it

    // ---------------- epilogue ----------------
    conv_epilogue<BM, BN, WM, WN, TM, TN>(acc, a, smem, red, n0, rowmap);
}

static int fill_convk(const srvp_conv_desc* d, ConvK& k) {
    k.src0 = (const bf16_t*)d->src0; k.src1 = (const bf16_t*)d->src1; k.map1 = d->map1;
    k.C0 = d->C0; k.C1 = d->C1; k.H0p = d->H0p; k.W0p = d->W0p; k.H1p = d->H1p; k.W1p = d->W1p;
    k.ups0 = d->ups0; k.ups1 = d->ups1; k.si = d->si; k.ntaps = d->ntaps;
    k.dy_bits = 0; k.dx_bits = 0;
    const int nent = d->tap_phase_chunks > 0 ? 4 * d->ntaps : d->ntaps;
    SRVP_REQUIRE(nent <= SRVP_MAX_TAPS, "srvp_conv_mfma: 4 x ntaps tap entries exceed %d", SRVP_MAX_TAPS);
    for (int t = 0; t < nent; ++t) {
        SRVP_REQUIRE(d->dy[t] >= 0 && d->dy[t] < 16 && d->dx[t] >= 0 && d->dx[t] < 16, "srvp_conv_mfma: tap offset out of [0,15]");
        k.dy_bits |= (unsigned long long)d->dy[t] << (4 * t);
        k.dx_bits |= (unsigned long long)d->dx[t] << (4 * t);
    }
    k.phase_chunks = d->tap_phase_chunks;
    k.splitk = 1; k.slab = 0;
    k.f32_quad = d->f32_quad;
    k.wt = (const bf16_t*)d->wt; k.Cout = d->Cout; k.N = d->N; k.OH = d->OH; k.OW = d->OW;
    k.dst = (bf16_t*)d->dst; k.DHp = d->DHp; k.DWp = d->DWp; k.so = d->so; k.ooy = d->ooy; k.oox = d->oox;
    k.Cdst = d->Cdst; k.cdst_off = d->cdst_off; k.stats = d->stats; k.stat_mod = d->stat_mod;
    k.out_f32 = d->out_f32; k.out_nc = d->out_nc; k.out_sigmoid = d->out_sigmoid;
    k.map0 = d->map0; k.dst_is_f32 = d->dst_is_f32; k.add_f32 = d->add_f32; k.add_mod = d->add_mod;
    k.bnr_raw = (const bf16_t*)d->bnr_raw; k.bnr_coef = d->bnr_coef; k.bnr_red = d->bnr_red;
    k.ep_coef = d->ep_coef; k.ep_act = d->ep_act; k.ep_border = d->ep_border;
    return SRVP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Halo-tiled 3x3 stride-1 convolution ("same" convs, their data-gradients, and the nearest-x2-upsampled variants):
// the workgroup owns a 256-pixel SPATIAL tile (16x16 pixels of one image, or whole 8x8 / 4x4 images) and stages the
// input patch including its 1-pixel halo in LDS ONCE per 64-channel chunk; the 9 taps are then LDS address offsets,
// so the activation traffic from L2 drops ~6.5x versus re-gathering the rows per tap (the generic kernel above), and
// the weights never touch LDS: B fragments are loaded from L2 straight into registers (fragment-major packing).
//
// Patch layout in LDS: [pixel][64 ch] (128-byte rows, 16-byte chunks XOR-swizzled by a function g(py, px) of the patch
// coordinates chosen per geometry so that the ds_read_b128 fragment reads of all 9 taps are bank-conflict free).  Two modes:
//   halo    (tile smaller than the image): pixel = py * PW + px over the (fh+2) x (fw+2) source window;
//   compact (tile = IMG whole images): rows of PW = fw+1 pixels, fh+1 rows per image: the right border pixel of a
//           row IS the left border pixel of the next row and the bottom border row of an image IS the top border row
//           of the next one (all zeros, physically present in the source), which keeps four 8x8 images at 334 pixels.
struct HaloK {
    ConvK a;
    int TH, TW, lgTW, lgTHW, IMG, PW, IS, Ppix, NA, tiles_x, tiles_y;
    unsigned rIS, rPW;                // ceil(2^32 / IS), ceil(2^32 / PW)
    int Tn;                           // > 1: time steps per sample for the S-sharing workgroup order
    int sw_sh, sw_c1, sw_c2;          // chunk swizzle g(py, px) = ((px >> sw_sh) + sw_c1 * py + sw_c2 * (py >> 1)) & 7
};

// Up to four launches that differ only in their descriptors (the four output phases of a sub-pixel upsample conv) run as
// ONE grid: blockIdx.y selects the descriptor.
struct HaloKN { HaloK k[4]; int n; };

// BN = 256 (wave tile 128 x 128, 256 accumulator registers -> AGPRs, one workgroup per CU): every A fragment read from LDS
// and every B fragment read from L2 feeds four MFMAs instead of two / four -- half the LDS and L1 bytes per flop.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 256 ? 1 : 2))) void conv_halo_kernel(const HaloKN pk) {
    // descriptor (output phase) = fastest-varying part of the logical workgroup index: the phases of one spatial tile read the SAME
    // input patch and run back to back on one XCD, so phases 2-4 find it in L2 (as grid.y they ran a whole grid apart and the
    // 0.3 GB input of the 64x64 stage entry was read from HBM four times)
    const unsigned lb_all = xcd_remap(blockIdx.x, gridDim.x);
    const HaloK& p = pk.k[pk.n > 1 ? lb_all % (unsigned)pk.n : 0u];
    constexpr int BK = 64, NT = 256, CPR = 8;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    // patch pieces per thread: 256-pixel tiles 11 * 256 16-byte pieces = 352 pixels >= 334 (13 = 416 pixels for the
    // 128-column variant, whose epilogue staging is larger than the patch anyway: sixteen 4x4 images + borders = 406);
    // 128-pixel tiles: 192 >= 180
    constexpr int NA_MAX = BM == 256 ? (BN >= 128 ? 13 : 11) : 6;
    constexpr int A_BYTES = NA_MAX * NT * 16;
    constexpr int LDC = BN + 8;
    constexpr int C_BYTES = BM * LDC * 2;
    constexpr int SMEM = A_BYTES > C_BYTES ? A_BYTES : C_BYTES;
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM + WM * BN * 8];
    unsigned char* Ab = smem;                                             // [<=352 px][64] bf16
    float* red = reinterpret_cast<float*>(smem + SMEM);

    const ConvK& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const unsigned lb = pk.n > 1 ? lb_all / (unsigned)pk.n : lb_all;
    const int n_tiles = a.Cout / BN;
    const int n0 = (lb % n_tiles) * BN;
    unsigned sp = lb / n_tiles;
    // hoisted-skip launches (+ S[n % add_mod]): the frames t*B + b of one sample b are made CONSECUTIVE workgroups, so
    // the fp32 S tile of (b, spatial tile) is read from HBM once and then from L2 for the other T - 1 time steps
    int tslot = 0, Gb = 0;
    if (p.Tn > 1) { tslot = sp % p.Tn; sp /= p.Tn; Gb = a.add_mod / p.IMG; }
    const int tx = sp % p.tiles_x; sp /= p.tiles_x;
    const int ty = sp % p.tiles_y;
    const int nb0 = (tslot * Gb + sp / p.tiles_y) * p.IMG;
    const int y0 = ty * p.TH, x0 = tx * p.TW;
    const int ups = a.ups0 ? 1 : 0;
    const int sy0 = y0 >> ups, sx0 = x0 >> ups;
    const int C = a.C0;

    // ---- patch DMA sources (fixed over the K loop except for the channel-chunk offset)
    unsigned aoff[NA_MAX];
#pragma unroll
    for (int i = 0; i < NA_MAX; ++i) {
        if (i >= p.NA) { aoff[i] = 0; continue; }
        const int q = tid + i * NT;
        const int pix = q >> 3, pos = q & 7;
        // exact small-integer divisions by the launch constants IS, PW through host-computed reciprocals ceil(2^32 / d)
        int img = (int)__umulhi((unsigned)pix, p.rIS);
        const int rem = pix - img * p.IS;
        int py = (int)__umulhi((unsigned)rem, p.rPW), px = rem - py * p.PW;
        int Y = sy0 + py, X = sx0 + px;
        if (pix >= p.Ppix || img >= p.IMG) { img = 0; Y = 0; X = 0; }      // padding / trailing pieces: a zero border pixel
        int n = nb0 + img;
        if (n >= a.N) n = a.N - 1;
        if (a.map0) n = a.map0[n];
        const int g = ((px >> p.sw_sh) + p.sw_c1 * py + p.sw_c2 * (py >> 1)) & 7;
        aoff[i] = (((unsigned)n * a.H0p + Y) * a.W0p + X) * C + ((pos ^ g) * 8);
    }
    // ---- fragment rows of this lane
    const int lrow = lane & 31, lkc = lane >> 5;
    int abase[TM], aoy[TM], aox[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * (TM * 32) + i * 32 + lrow;
        abase[i] = (r >> p.lgTHW) * p.IS;
        aoy[i] = (r >> p.lgTW) & (p.TH - 1);
        aox[i] = r & (p.TW - 1);
    }

    const int ntaps = a.ntaps;
    const int S = ntaps * (C / BK);
    auto stage_a = [&](int cc) {
#pragma unroll
        for (int i = 0; i < NA_MAX; ++i)
            if (i < p.NA)
                __builtin_amdgcn_global_load_lds((gptr_t)(a.src0 + aoff[i] + cc * BK), (lptr_t)(Ab + ((size_t)i * NT + wid * 64) * 16), 16, 0, 0);
    };
    // ---- weights: MFMA B fragments straight from L2/L1 into registers (pack layout 1: one contiguous 1 KiB load per
    // fragment), never through LDS.  Measured on the LDS-staged versions: writing the 16 KiB weight tile per tap into LDS
    // (LDS-DMA or ds_write_b128 alike) cost 20-30 % of the kernel -- the LDS was the contended unit (patch DMA writes +
    // 24 fragment reads per tap and wave + the tile writes), and the per-tap barrier came with it.  Now the LDS holds the
    // patch only and the waves synchronise once per 64-channel chunk.
    // Group g = (step s = (chunk, tap), 16-wide k slice kk): address of fragment j of group g
    const int JT = a.Cout >> 5;                          // 32-wide column tiles of the packed tensor
    const int wn_u = __builtin_amdgcn_readfirstlane(wn);  // wave-uniform (kept in SGPRs)
    const char* wbase = reinterpret_cast<const char*>(a.wt + (size_t)((n0 >> 5) + wn_u * TN) * 512);
    const unsigned lane_b = (unsigned)lane * 16u;        // this lane's 16 bytes inside a 1 KiB fragment
    // the fragment stream is walked with SCALAR byte offsets (wave-uniform): slice kk of step (chunk cc, tap t) lives at
    // ((t * NC + cc) * 4 + kk) * JT KiB; only the per-lane part sits in VGPRs
    const unsigned slice_b = (unsigned)JT * 1024u;        // bytes per 16-wide k slice (all column tiles)
    const unsigned tap_b = (unsigned)(C >> 6) * 4u * slice_b;
    auto step_off = [&](int s) -> unsigned {              // byte offset of step s, slice 0 (clamped: run-off prefetches)
        if (s >= S) s = S - 1;
        const int cc = s / ntaps, t = s - cc * ntaps;
        return (unsigned)t * tap_b + (unsigned)cc * 4u * slice_b;
    };

    auto rowmap = [&](int row, int& n, int& oy, int& ox) -> bool {
        n = nb0 + (row >> p.lgTHW);
        oy = y0 + ((row >> p.lgTW) & (p.TH - 1));
        ox = x0 + (row & (p.TW - 1));
        return n < a.N;
    };
    f32x16_t acc[TM][TN];
    // Tn > 1: frames are t * add_mod + sample, so the sample index needs no division
    const int nsub = p.Tn > 1 ? tslot * a.add_mod : 0;
    conv_acc_init<WM, WN, TM, TN>(acc, a, n0, rowmap, [&](int n) { return p.Tn > 1 ? n - nsub : n % a.add_mod; });

    // B fragment ring: slot kk holds slice kk of the current step; loads run two slices ahead (vmcnt counted by hand:
    // the loads are inline asm so that hipcc neither reorders them nor drains them at the LDS-DMA instructions)
    u32x4_t bq[4][TN];
    auto issue_b = [&](unsigned soff, int kk, u32x4_t (&dst)[TN]) {
        const char* sb = wbase + (soff + (unsigned)kk * slice_b);     // scalar base of slice kk; column tile j at +j KiB
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst[0]) : "v"(lane_b), "s"(sb) : "memory");
        if constexpr (TN >= 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=&v"(dst[1]) : "v"(lane_b), "s"(sb) : "memory");
        if constexpr (TN == 4) {
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=&v"(dst[2]) : "v"(lane_b), "s"(sb) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=&v"(dst[3]) : "v"(lane_b), "s"(sb) : "memory");
        }
    };
    static_assert(TN == 1 || TN == 2 || TN == 4, "B ring: 1, 2 or 4 column tiles per wave");
    const int NC = C / BK;
    unsigned soff = step_off(0);
    issue_b(soff, 0, bq[0]);
    issue_b(soff, 1, bq[1]);
    stage_a(0);
    int s = 0;
    for (int cc = 0; cc < NC; ++cc) {
        // new chunk: its patch DMA (and everything older) has landed, visible to all waves.  The wait is the BUILTIN so
        // that hipcc's own waitcnt pass knows the LDS-DMA is retired -- with an opaque asm wait it re-inserts vmcnt(0)
        // before the fragment reads of every tap, which drains the weight prefetch ring.
        __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int tph = a.phase_chunks > 0 ? (cc / a.phase_chunks) * ntaps : 0;      // first tap entry of this chunk's phase
        for (int t = 0; t < ntaps; ++t, ++s) {
            const unsigned soff_n = step_off(s + 1);
            const int dy = (int)((a.dy_bits >> (4 * (tph + t))) & 15), dx = (int)((a.dx_bits >> (4 * (tph + t))) & 15);
            int apix[TM], asw[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int vy = (aoy[i] + dy + ups) >> ups, vx = (aox[i] + dx + ups) >> ups;
                apix[i] = abase[i] + vy * p.PW + vx;
                asw[i] = ((vx >> p.sw_sh) + p.sw_c1 * vy + p.sw_c2 * (vy >> 1)) & 7;
            }
            bf16x8_t af[2][TM];
            const unsigned char* arow[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) arow[i] = Ab + (size_t)apix[i] * 128;
            auto load_a = [&](int kk, bf16x8_t (&fa)[TM]) {
                const int kc = kk * 2 + lkc;
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(arow[i] + ((kc ^ asw[i]) * 16));
            };
            load_a(0, af[0]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                // weights two slices ahead (slot (kk+2)&3 was consumed two slices ago)
                if (kk < 2) issue_b(soff, kk + 2, bq[kk + 2]); else issue_b(soff_n, kk - 2, bq[kk - 2]);
                if (kk + 1 < 4) load_a(kk + 1, af[(kk + 1) & 1]);
                // slice kk has landed when only the 2 younger slices (2*TN loads) are outstanding
                if constexpr (TN == 4) asm volatile("s_waitcnt vmcnt(8)" : "+v"(bq[kk][0]), "+v"(bq[kk][1]), "+v"(bq[kk][2]), "+v"(bq[kk][3])::"memory");
                else if constexpr (TN == 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(bq[kk][0]), "+v"(bq[kk][1])::"memory");
                else asm volatile("s_waitcnt vmcnt(2)" : "+v"(bq[kk][0])::"memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][i], __builtin_bit_cast(bf16x8_t, bq[kk][j]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            soff = soff_n;
        }
        if (cc + 1 < NC) {
            __builtin_amdgcn_s_barrier();                 // every wave is done with this chunk's patch
            asm volatile("" ::: "memory");
            stage_a(cc + 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // run-off prefetches
    __syncthreads();

    conv_epilogue<BM, BN, WM, WN, TM, TN>(acc, a, smem, red, n0, rowmap, nb0 + p.IMG <= a.N);
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static int g_halo = -1;      // -1: read SRVP_CONV_HALO on first use; 0 = generic kernel only; 1 = halo kernel where eligible

// Tile geometry of the halo kernel for this descriptor; false if the descriptor is not a 3x3 stride-1 single-source
// convolution on a 1-pixel-bordered tensor (or its patch does not fit) -- the generic kernel takes it then.
static bool halo_geometry(const srvp_conv_desc* d, HaloK& h, int BM) {
    if (g_halo < 0) { const char* e = getenv("SRVP_CONV_HALO"); g_halo = e ? atoi(e) : 1; }
    if (!g_halo || d->elem_f32 || d->splitk > 1) return false;
    if (d->ntaps > 9 || d->si != 1 || d->C1 != 0 || d->C0 % 64 != 0) return false;
    const int nent = d->tap_phase_chunks > 0 ? 4 * d->ntaps : d->ntaps;
    if (nent > SRVP_MAX_TAPS || (d->tap_phase_chunks > 0 && (d->C0 / 64) != 4 * d->tap_phase_chunks)) return false;
    for (int t = 0; t < nent; ++t) if (d->dy[t] < 0 || d->dy[t] > 2 || d->dx[t] < 0 || d->dx[t] > 2) return false;
    const int OH = d->OH, OW = d->OW, ups = d->ups0 ? 1 : 0;
    if (OH < 2 || OW < 2 || (OH & (OH - 1)) || (OW & (OW - 1))) return false;
    if (d->H0p != (OH >> ups) + 2 || d->W0p != (OW >> ups) + 2) return false;
    const bool compact = OH * OW <= BM;
    if (!compact && (OH % 16 || OW % 16)) return false;
    h.TH = compact ? OH : BM / 16; h.TW = compact ? OW : 16;
    h.IMG = BM / (h.TH * h.TW);
    h.lgTW = ilog2(h.TW); h.lgTHW = ilog2(h.TH * h.TW);
    const int fh = h.TH >> ups, fw = h.TW >> ups;
    if (compact) { h.PW = fw + 1; h.IS = (fh + 1) * h.PW; h.Ppix = h.IMG * h.IS + h.PW + 1; }
    else { h.PW = fw + 2; h.IS = (fh + 2) * h.PW; h.Ppix = h.IS; }
    h.NA = (h.Ppix * 8 + 255) / 256;
    if (h.NA > (BM == 256 ? (d->Cout % 128 == 0 ? 13 : 11) : 6)) return false;
    h.rIS = (unsigned)(((1ull << 32) + h.IS - 1) / h.IS); h.rPW = (unsigned)(((1ull << 32) + h.PW - 1) / h.PW);
    // conflict-free ds_read_b128 fragment reads for every tap (exhaustive search over this family per geometry)
    if (ups) { if (fw >= 8) { h.sw_sh = 1; h.sw_c1 = 4; h.sw_c2 = 0; } else { h.sw_sh = 0; h.sw_c1 = 0; h.sw_c2 = 4; } }
    else if (compact) { h.sw_sh = 0; h.sw_c1 = 0; h.sw_c2 = 0; }
    else { h.sw_sh = 1; h.sw_c1 = 0; h.sw_c2 = 0; }
    h.tiles_x = OW / h.TW; h.tiles_y = OH / h.TH;
    h.Tn = (d->add_f32 && d->add_mod > 0 && d->N % d->add_mod == 0 && d->add_mod % h.IMG == 0) ? d->N / d->add_mod : 1;
    return true;
}

template <int BM, int BN, int WM, int WN>
int launch_halo_n(const srvp_conv_desc* d, int n, int bm, hipStream_t st) {
    HaloKN hn;
    long long blocks = 0;
    for (int i = 0; i < n; ++i) {
        HaloK& h = hn.k[i];
        SRVP_REQUIRE(halo_geometry(d + i, h, bm), "srvp_conv_mfma(halo): descriptor %d is not eligible", i);
        if (int rc = fill_convk(d + i, h.a)) return rc;
        SRVP_REQUIRE(d[i].wt_fragmajor == 1, "srvp_conv_mfma: this launch runs on the halo kernel and needs fragment-major weights (srvp_conv_wants_fragmajor)");
        const long long b = (long long)((d[i].N + h.IMG - 1) / h.IMG) * h.tiles_x * h.tiles_y * (d[i].Cout / BN);
        SRVP_REQUIRE(i == 0 || b == blocks, "srvp_conv_mfma(halo): descriptors of one launch must have the same grid");
        blocks = b;
    }
    for (int i = n; i < 4; ++i) hn.k[i] = hn.k[0];
    SRVP_REQUIRE(blocks > 0 && blocks < (1ll << 31), "srvp_conv_mfma(halo): bad grid %lld", blocks);
    hn.n = n;
    SRVP_REQUIRE(blocks * n < (1ll << 31), "srvp_conv_mfma(halo): bad grid %lld x %d", blocks, n);
    hipLaunchKernelGGL((conv_halo_kernel<BM, BN, WM, WN>), dim3((unsigned)(blocks * n)), dim3(256), 0, st, hn);
    SRVP_CHECK_LAUNCH("srvp_conv_mfma(halo)");
    return SRVP_OK;
}

static int generic_mode() {
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("SRVP_CONV_MODE"); mode = e ? atoi(e) : 5; }
    return mode;
}
// generic kernel, default single-buffer BK = 64 variant: weights fragment-major, read from L2 into registers
static bool generic_wants_fragmajor(const srvp_conv_desc* d) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SRVP_CONV_BDIRECT"); on = e ? atoi(e) : 1; }
    return on && !d->elem_f32 && generic_mode() == 5 && d->C0 % 64 == 0 && d->C1 % 64 == 0 && d->Cout % 64 == 0;
}

// 0: not a halo launch; else the tile size (128 / 256) the dispatcher picks for this descriptor
static int halo_variant(const srvp_conv_desc* d) {
    HaloK h;
    static int bm128 = -1;      // A/B switch: 128-pixel tiles for the narrow (Cout < 128) layers
    if (bm128 < 0) { const char* e = getenv("SRVP_HALO_BM128"); bm128 = e ? atoi(e) : 0; }
    if (bm128 && d->Cout % 128 != 0 && halo_geometry(d, h, 128)) return (bm128 & 2) ? 129 : 128;
    static int bn256 = -1;      // A/B switch: 256 x 256 tiles (one workgroup per CU) for the wide layers
    if (bn256 < 0) { const char* e = getenv("SRVP_HALO_BN256"); bn256 = e ? atoi(e) : 0; }
    if (!halo_geometry(d, h, 256)) return 0;
    return (bn256 && d->Cout % 256 == 0) ? 257 : 256;
}

static int launch_halo_any(const srvp_conv_desc* d, int n, int variant, hipStream_t st) {
    if (variant == 257) return launch_halo_n<256, 256, 2, 2>(d, n, 256, st);
    if (variant == 256) {
        // small grids (few images): 64-column tiles double the number of workgroups of a launch that would not fill the chip with
        // 128-column ones
        static int below = -1;
        if (below < 0) { const char* e = getenv("SRVP_HALO_BN64_BELOW"); below = e ? atoi(e) : 600; }
        if (d->Cout % 128 == 0 && below > 0) {
            HaloK h;
            if (halo_geometry(d, h, 256)) {
                const long long b = (long long)((d->N + h.IMG - 1) / h.IMG) * h.tiles_x * h.tiles_y * (d->Cout / 128) * n;
                if (b < below && h.NA <= 11) return launch_halo_n<256, 64, 4, 1>(d, n, 256, st);     // (11 = patch pieces of the 64-column kernel)
            }
        }
        if (d->Cout % 128 == 0) return launch_halo_n<256, 128, 2, 2>(d, n, 256, st);
        if (d->Cout % 64 == 0) return launch_halo_n<256, 64, 4, 1>(d, n, 256, st);
        return launch_halo_n<256, 32, 4, 1>(d, n, 256, st);
    }
    if (d->Cout % 64 == 0) return variant == 129 ? launch_halo_n<128, 64, 4, 1>(d, n, 128, st) : launch_halo_n<128, 64, 2, 2>(d, n, 128, st);
    return launch_halo_n<128, 32, 4, 1>(d, n, 128, st);
}

template <int BM, int BN, int BK, int WM, int WN, int NBUF, bool BDIRECT = false>
int launch(const srvp_conv_desc* d, hipStream_t st) {
    long long M = (long long)d->N * d->OH * d->OW;
    long long mt = (M + BM - 1) / BM;
    long long blocks = mt * (d->Cout / BN);
    SRVP_REQUIRE(blocks > 0 && blocks < (1ll << 31), "srvp_conv_mfma: bad grid %lld", blocks);
    ConvK k;
    if (int rc = fill_convk(d, k)) return rc;
    if (d->splitk > 1) {
        const int steps = d->ntaps * ((d->C0 + d->C1) / BK);
        SRVP_REQUIRE(d->dst_is_f32 && !d->stats && !d->add_f32 && !d->out_f32 && d->splitk <= steps && d->splitk <= 64,
                     "srvp_conv_mfma: splitk = %d needs an fp32 slab destination without statistics / add_f32 and at most %d (64) splits", d->splitk, steps);
        k.splitk = d->splitk;
        k.slab = (long long)d->N * d->DHp * d->DWp * d->Cdst;
    }
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, BK, WM, WN, NBUF, BDIRECT>), dim3((unsigned)blocks, (unsigned)k.splitk), dim3(WM * WN * 64), 0, st, k);
    SRVP_CHECK_LAUNCH("srvp_conv_mfma");
    return SRVP_OK;
}

}  // namespace

extern "C" int srvp_conv_set_halo(int on) { g_halo = on; return SRVP_OK; }

extern "C" int srvp_conv_runs_on_halo(const srvp_conv_desc* d) { return d ? halo_variant(d) : 0; }

extern "C" int srvp_conv_wants_fragmajor(const srvp_conv_desc* d) { return d && (halo_variant(d) || generic_wants_fragmajor(d)) ? 1 : 0; }

extern "C" int srvp_conv_mfma(const srvp_conv_desc* d, void* stream);

// n <= 4 launches as one grid when they all run on the same halo kernel variant with the same grid (the output phases of
// a sub-pixel upsample conv); otherwise simply n launches.
// ep_coef (eval-mode BatchNorm + activation in the epilogue): the preconditions every kernel that serves it relies on, checked for every
// descriptor at BOTH entry points (the multi-launch forms took such descriptors unchecked)
static int check_ep_coef(const srvp_conv_desc* d) {
    SRVP_REQUIRE(!d->ep_coef || (!d->elem_f32 && !d->stats && !d->dst_is_f32 && !d->out_f32 && d->splitk <= 1 && !d->bnr_red && d->ep_act >= 0 && d->ep_act <= 4 && d->ep_border >= 0 && d->ep_border <= 1),
                 "srvp_conv_mfma: ep_coef (eval-mode BatchNorm epilogue) needs a plain bf16 forward launch without statistics");
    return SRVP_OK;
}

extern "C" int srvp_conv_mfma_multi(const srvp_conv_desc* d, int n, void* stream) {
    SRVP_REQUIRE(d && n >= 1, "srvp_conv_mfma_multi: bad args");
    for (int i = 0; i < n; ++i)
        if (int rc = check_ep_coef(d + i)) return rc;
    {
        int taken = 0;
        if (int rc = srvp_conv_stream_sub64_launch(d, n, (hipStream_t)stream, &taken)) return rc;
        if (taken) return SRVP_OK;
    }
    bool same = n <= 4;
    const int v = halo_variant(d);
    for (int i = 1; i < n && same; ++i)
        same = halo_variant(d + i) == v && d[i].Cout == d[0].Cout && d[i].N == d[0].N && d[i].OH == d[0].OH && d[i].OW == d[0].OW &&
               d[i].C0 == d[0].C0 && d[i].ntaps == d[0].ntaps && (d[i].add_f32 != nullptr) == (d[0].add_f32 != nullptr) && d[i].add_mod == d[0].add_mod;
    if (v && same && n > 1 && d->src0 && d->wt) return launch_halo_any(d, n, v, (hipStream_t)stream);
    for (int i = 0; i < n; ++i)
        if (int rc = srvp_conv_mfma(d + i, stream)) return rc;
    return SRVP_OK;
}

extern "C" int srvp_conv_mfma(const srvp_conv_desc* d, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SRVP_REQUIRE(d && d->src0 && d->wt && (d->dst || d->out_f32), "srvp_conv_mfma: null pointer");
    SRVP_REQUIRE(!d->out_f32 || (d->Cout == 32 && d->out_nc >= 1 && d->out_nc <= 32), "srvp_conv_mfma: fp32 frame output needs Cout == 32");
    SRVP_REQUIRE(d->C0 % 32 == 0 && d->C1 % 32 == 0 && d->Cout % 32 == 0 && d->C0 > 0,
                 "srvp_conv_mfma: channel counts must be padded to 32 (C0=%d C1=%d Cout=%d)", d->C0, d->C1, d->Cout);
    SRVP_REQUIRE(d->C1 == 0 || d->src1, "srvp_conv_mfma: C1>0 needs src1");
    SRVP_REQUIRE(d->ntaps >= 1 && d->ntaps <= SRVP_MAX_TAPS, "srvp_conv_mfma: ntaps=%d", d->ntaps);
    SRVP_REQUIRE(d->Cdst % 8 == 0 && d->cdst_off % 8 == 0, "srvp_conv_mfma: dst channel slice must be 16-byte aligned");
    SRVP_REQUIRE(d->stats == nullptr || d->stat_mod > 0, "srvp_conv_mfma: stat_mod");
    SRVP_REQUIRE(d->f32_quad == 0 || (!d->elem_f32 && d->splitk <= 1 && (d->dst_is_f32 || d->add_f32) && d->DWp % (4 * d->f32_quad) == 0 &&
                                      (d->add_f32 == nullptr || (d->so == d->f32_quad && d->OW % 4 == 0)) && d->cdst_off == 0),
                 "srvp_conv_mfma: f32_quad = %d needs an fp32 S tensor whose width is a multiple of 4 x the consumer stride (and so == f32_quad, OW %% 4 == 0 on the consumer)", d->f32_quad);
    SRVP_REQUIRE(!(d->elem_f32 && d->splitk > 1), "srvp_conv_mfma: splitk is not available in fp32 parity mode");
    SRVP_REQUIRE(!d->bnr_red || (d->bnr_raw && d->bnr_coef && !d->elem_f32 && !d->stats && !d->dst_is_f32 && !d->out_f32 && d->so == 1 && d->ooy == 0 &&
                                 d->oox == 0 && d->cdst_off == 0 && d->Cdst == d->Cout && d->DHp == d->OH && d->DWp == d->OW && halo_variant(d) >= 256 &&
                                 (long long)d->N * d->OH * d->OW * d->Cout < (1ll << 32)),
                 "srvp_conv_mfma: bnr_red (fused BatchNorm-backward reduction) needs a plain bf16 data-gradient launch on the halo kernel");
    if (int rc = check_ep_coef(d)) return rc;
    if (d->elem_f32) return srvp_conv_f32_launch(d, st);
    SRVP_REQUIRE((long long)d->N * d->H0p * d->W0p * d->C0 < (1ll << 32) && (d->C1 == 0 || d->map1 || (long long)d->N * d->H1p * d->W1p * d->C1 < (1ll << 32)),
                 "srvp_conv_mfma: source tensors must have fewer than 2^32 elements");
    {
        int taken = 0;
        if (int rc = srvp_conv_stream64_launch(d, st, &taken)) return rc;
        if (taken) return SRVP_OK;
    }
    if (const int v = halo_variant(d)) return launch_halo_any(d, 1, v, st);
    SRVP_REQUIRE(d->tap_phase_chunks == 0, "srvp_conv_mfma: tap_phase_chunks launches run on the halo kernel only, and this descriptor is not eligible for it");
    const bool k64 = (d->C0 % 64 == 0) && (d->C1 % 64 == 0);
    // LDS ring depth / K step (A/B switch SRVP_CONV_MODE): 5 = BK64 single buffer (default: 3 workgroups per CU hide the
    // DMA latency better than a deeper ring at 1-2 workgroups per CU: 39.0 vs 40.9 (x2) / 43 (BK32 x3) / 46 (BK32 x4) /
    // 56 ms (BK64 x3) per step), 0 = BK64 x2, 1 = BK32 x4, 2 = BK32 x3, 3 = BK64 x3
    const int mode = generic_mode();
    // (the caller may keep tap-major weights on a launch that could take fragment-major ones: both variants exist)
    SRVP_REQUIRE(d->wt_fragmajor == 0 || generic_wants_fragmajor(d), "srvp_conv_mfma: fragment-major weights on a launch whose kernel reads them tap-major (srvp_conv_wants_fragmajor)");
    if (d->wt_fragmajor) {
        if (d->Cout % 128 == 0) return launch<128, 128, 64, 2, 2, 1, true>(d, st);
        return launch<128, 64, 64, 2, 2, 1, true>(d, st);
    }
    const int m = k64 ? mode : (mode == 2 ? 2 : 1);   // 5 = BK64, single buffer
    if (d->Cout % 128 == 0) {
        switch (m) {
            case 0: return launch<128, 128, 64, 2, 2, 2>(d, st);
            case 1: return launch<128, 128, 32, 2, 2, 4>(d, st);
            case 2: return launch<128, 128, 32, 2, 2, 3>(d, st);
            case 5: return launch<128, 128, 64, 2, 2, 1>(d, st);
            default: return launch<128, 128, 64, 2, 2, 3>(d, st);
        }
    } else if (d->Cout % 64 == 0) {
        switch (m) {
            case 0: return launch<128, 64, 64, 2, 2, 2>(d, st);
            case 1: return launch<128, 64, 32, 2, 2, 4>(d, st);
            case 2: return launch<128, 64, 32, 2, 2, 3>(d, st);
            case 5: return launch<128, 64, 64, 2, 2, 1>(d, st);
            default: return launch<128, 64, 64, 2, 2, 3>(d, st);
        }
    } else {
        return launch<128, 32, 32, 4, 1, 4>(d, st);
    }
}
