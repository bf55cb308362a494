// Image-side INPUT layer as a streaming kernel: Conv2d(nc -> 64, 3x3, stride 1, pad 1) on 64x64 fp32 frames (reference
// module/conv.py:200, first vgg_layer), bf16 NHWC raw output + BatchNorm statistics -- and, with the same arithmetic, the DATA GRADIENT
// of the image-side output layer (ConvTranspose2d(64 -> nc, 3, 1, 1), conv.py:353: the gradient frames are its "image") with the
// producer block's BatchNorm-backward sums accumulated from the values it has just computed (srvp_conv_in_fwd_bnr).
//
// Why: the layer WRITES 1.2 GB of bf16 at 2304 frames (and reads 0.1 GB of frames): HBM work.  The tile kernel of smallconv.hip
// (conv_in_fwd_mfma_kernel) spends 0.22 ms of it inside v_mfma_f32_32x32x2_f32 alone (the fp32 matrix instruction runs at the vector
// rate), stages every 128-pixel tile's patch and output through LDS behind barriers, and sits at 0.53 ms (0.78 ms with the fused
// sums) -- twice the time the bytes take.
//
// Here: ONE persistent workgroup (8 waves) per CU walks whole frames.  The fp32 frame arrives by LDS-DMA into a staging area under the
// previous frame's arithmetic and is re-laid as one 16-byte RECORD per padded pixel: the nc values as bf16 HIGH parts, then the nc bf16
// LOW parts (x = hi + lo + O(2^-17 x)), zero-padded to 8 slots.  A record IS the 8-value k group a lane of v_mfma_f32_32x32x16_bf16
// holds, so one ds_read_b128 per lane is the operand for two taps of 32 pixels (lanes 0-31: tap 2s, lanes 32-63: tap 2s + 1), five
// reads per 32-pixel tile.  The WEIGHTS are the stationary M operand, split the same way and held in registers for the whole launch:
// set 1 = (w_hi | w_hi | 0) and set 2 = (w_lo | w_lo | 0) meet (x_hi | x_lo): w x = (w_hi + w_lo)(x_hi + x_lo), relative
// error 2^-17 per product (what two bf16 terms leave of each operand), accumulated in fp32 -- two orders below the bf16 rounding of the stored output (the precision = 'fp32'
// parity mode keeps the exact direct kernels).  20 MFMAs of 8 passes per 32 pixels x 64 channels instead of 28 of 16 passes.
// nc = 1 (KTH, Moving MNIST): the record holds (x1, x2, x3, x1, x2, x1) -- three bf16 terms = all 24 bits -- against the weights' (w1, w1, w1,
// w2, w2, w3): every product term down to 2^-24 in one weight set, i.e. fp32-exact products in 10 MFMAs.
// Output channels are PERMUTED over the MFMA rows so that a lane ends up with 16 consecutive channels of one pixel: it rounds them and
// stores 32 contiguous bytes straight from registers (no LDS staging, no barrier in the pixel loop), and the BatchNorm sums are plain
// per-lane register accumulators over the whole launch (one double atomic per channel and workgroup at the end).
#include "common.h"
#include "../../include/srvp_hip.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int IW = 64, PWI = 66;
constexpr int REC_BYTES = PWI * PWI * 16;                   // 69 696: [66][66] records of 8 bf16 (1-pixel zero border)

__device__ __forceinline__ unsigned short bf_hi(float v) { return f2bf(v); }
__device__ __forceinline__ unsigned short bf_lo(float v) { return f2bf(v - bf2f(f2bf(v))); }
// three bf16 terms = all 24 significant bits of an fp32 value (the subtractions are exact)
__device__ __forceinline__ void bf_split3(float v, unsigned short (&t)[3]) {
    t[0] = f2bf(v); const float r1 = v - bf2f(t[0]);
    t[1] = f2bf(r1); t[2] = f2bf(r1 - bf2f(t[1]));
}

// Sum of each of 16 per-lane values over the 32 lanes of a lane group (lanes 0-31 / 32-63), as a reduce-scatter butterfly: 15 + 1 lane
// exchanges instead of 16 x 5; afterwards lane lcol holds the total of element rs16_elem(lcol) (both lanes of a pair hold the same one).
// The array is left zeroed: the per-lane fp32 accumulators of the streaming kernels live for ONE work item (<= 32 values per lane), the
// totals are widened to double per item (the fp32 accumulation over a whole launch lost precision with N / ncu).
__device__ __forceinline__ float rs16_flush(float (&a)[16], int lcol) {
#pragma unroll
    for (int m = 16, h = 8; h >= 1; m >>= 1, h >>= 1) {
        const bool up = (lcol & m) != 0;
#pragma unroll
        for (int i = 0; i < h; ++i) {
            const float send = up ? a[i] : a[i + h], keep = up ? a[i + h] : a[i];
            a[i] = keep + __shfl_xor(send, m);
        }
    }
    const float r = a[0] + __shfl_xor(a[0], 1);
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = 0.f;
    return r;
}
__device__ __forceinline__ int rs16_elem(int lcol) { return (lcol >> 1) & 15; }
// The same result by 16 x 5 plain xor exchanges: inside conv_in_bnr_ring_kernel the butterfly's selects cost hipcc 100 registers (spills)
__device__ __forceinline__ float rs16_flush_plain(float (&a)[16], int lcol) {
    float r = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float u = a[e];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) u += __shfl_xor(u, m);
        if (rs16_elem(lcol) == e) r = u;
        a[e] = 0.f;
    }
    return r;
}

struct InStreamK {
    const float* x;          // (N, CIN, 64, 64) fp32 frames
    const float* w;          // (Cout_real, CIN, 3, 3) fp32
    bf16_t* raw;             // [N][64*64][64] bf16
    double* stats;           // [2][64] or null
    const bf16_t* bnr_raw;   // BNR: the producer block's raw output, same geometry as `raw`
    const float* bnr_coef;   // BNR: [4][64] scale, shift, mean, inverse std
    double* bnr_red;         // BNR: [2][64]
    int N, Cout_real, P;     // P: items per frame (1, 2, 4: bands of 64 / P rows)
};

template <int CIN, bool BNR>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_in_stream_kernel(const InStreamK a) {
    __shared__ __attribute__((aligned(16))) unsigned char rec[REC_BYTES];
    __shared__ __attribute__((aligned(16))) float stg[CIN * IW * IW];          // the frame (rows of this item) as it lies in HBM, lane-linear
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lcol = lane & 31, kg = lane >> 5;
    const int jw = wid & 1, g = wid >> 1;                   // this wave: output channels [32 jw, 32 jw + 32), rows g, g + 4, ... of the item
    // ---- records: zero once (the border stays zero; the interior rows an item needs are rewritten per item)
    for (int i = tid; i < REC_BYTES / 16; i += 512) reinterpret_cast<u32x4_t*>(rec)[i] = u32x4_t{0u, 0u, 0u, 0u};
    // ---- A fragments (weights): MFMA row R of this wave's tile <-> channel 32 jw + 16 ((R >> 2) & 1) + (R & 3) + 4 (R >> 3), so that lane
    // (pixel, half h) of the result holds channels 32 jw + 16 h + 0..15.  step s, k group kg = tap 2 s + kg; slot i of the group: see top.
    bf16x8_t wf[5][2];
    {
        const int R = lcol, co = 32 * jw + 16 * ((R >> 2) & 1) + (R & 3) + 4 * (R >> 3);
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int tap = 2 * s + kg;
            unsigned short v1[8], v2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ci = i % CIN;
                const bool ok = tap < 9 && co < a.Cout_real;
                const float wv = ok ? a.w[((size_t)co * CIN + ci) * 9 + (tap < 9 ? tap : 0)] : 0.f;
                if constexpr (CIN == 1) {
                    // one channel: the record has room for (x1, x2, x3, x1, x2, x1) against (w1, w1, w1, w2, w2, w3) -- every product term down to
                    // 2^-24 in ONE weight set: fp32-exact products, fp32 accumulation
                    unsigned short t3[3];
                    bf_split3(wv, t3);
                    v1[i] = i < 3 ? t3[0] : (i < 5 ? t3[1] : (i == 5 ? t3[2] : (unsigned short)0));
                    v2[i] = 0;
                } else {
                    v1[i] = i < 2 * CIN ? bf_hi(wv) : (unsigned short)0;
                    v2[i] = i < 2 * CIN ? bf_lo(wv) : (unsigned short)0;       // (w_lo x_lo rides along for free)
                }
            }
            u32x4_t p1, p2;
            p1.x = v1[0] | ((unsigned)v1[1] << 16); p1.y = v1[2] | ((unsigned)v1[3] << 16); p1.z = v1[4] | ((unsigned)v1[5] << 16); p1.w = v1[6] | ((unsigned)v1[7] << 16);
            p2.x = v2[0] | ((unsigned)v2[1] << 16); p2.y = v2[2] | ((unsigned)v2[3] << 16); p2.z = v2[4] | ((unsigned)v2[5] << 16); p2.w = v2[6] | ((unsigned)v2[7] << 16);
            wf[s][0] = __builtin_bit_cast(bf16x8_t, p1);
            wf[s][1] = __builtin_bit_cast(bf16x8_t, p2);
        }
#pragma unroll
        for (int s = 0; s < 5; ++s) { asm volatile("" : "+v"(wf[s][0])); asm volatile("" : "+v"(wf[s][1])); }
    }
    // byte offset of this lane's tap inside the records, per step (tap 9 does not exist: its weights are zero, it re-reads tap 8)
    unsigned tapoff[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + kg < 9 ? 2 * s + kg : 8;
        tapoff[s] = (unsigned)(((tap / 3) * PWI + tap % 3) * 16);
    }
    const int c0 = 32 * jw + 16 * kg;                        // first of this lane's 16 output channels
    // per-lane sums over ONE work item: plain forward: sum / sum of squares of the fp32 results; BNR: c1 = sum g, c2 = sum g * raw.
    // Widened to double per item (rs16_flush): the centring is * (c2 - mean c1) is done in double on the totals.
    float s1[16], s2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) s1[e] = s2[e] = 0.f;
    double d1 = 0., d2 = 0.;
    // BNR: the activation gate act'(scale raw + shift) = [fmaf(raw, scale, shift) > 0] as a THRESHOLD on the bf16 value raw -- the predicate
    // is monotone in raw, so there is a bf16 value T with predicate <=> (raw > T) for scale > 0 resp. (raw <= T) for scale < 0; T is found
    // exactly by evaluating the predicate itself on the bf16 neighbours of -shift / scale.  One compare per element, 16 registers less.
    float thr[16];
    bool neg[16];
    if constexpr (BNR) {
        auto ord = [](float f) { int o = __builtin_bit_cast(int, f); return o < 0 ? (int)(0x80000000u - (unsigned)o) : o; };       // monotone int image
        auto unord = [](int o) { return __builtin_bit_cast(float, o < 0 ? (int)(0x80000000u - (unsigned)o) : o); };
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float sc = a.bnr_coef[c0 + e], sh = a.bnr_coef[64 + c0 + e];
            neg[e] = sc < 0.f;
            if (!(sc > 0.f) && !(sc < 0.f)) { thr[e] = sh > 0.f ? -__builtin_huge_valf() : __builtin_huge_valf(); continue; }   // constant predicate
            // with u = sign(scale) raw: predicate <=> fmaf(u, |scale|, shift) > 0, non-decreasing in u.  T_u = the largest bf16 u where it is false
            const float as = fabsf(sc);
            int o = ord(bf2f(f2bf(-sh / as))) & ~0xFFFF;
            const int omax = ord(bf2f((unsigned short)0x7F7F)), omin = -omax;
            o = o > omax ? omax : (o < omin ? omin : o);
            for (int it = 0; it < 6 && o > omin && fmaf(unord(o), as, sh) > 0.f; ++it) o -= 0x10000;
            for (int it = 0; it < 6 && o < omax && !(fmaf(unord(o + 0x10000), as, sh) > 0.f); ++it) o += 0x10000;
            // scale > 0: raw > T_u.   scale < 0: -raw > T_u <=> raw < -T_u <=> !(raw > prev(-T_u)) on the bf16 grid
            thr[e] = neg[e] ? unord(ord(-unord(o)) - 0x10000) : unord(o);
        }
    }
    const int NR = IW / a.P;
    const int nitems = a.N * a.P;
    const unsigned rec_base = (unsigned)(uintptr_t)rec;
    // the frame rows [lo, hi] an item needs (its band + one halo row each side, inside the frame), every plane: DMA chunk c of 16 bytes
    // -> staging offset 16 c (the LDS side of a DMA is lane-linear)
    auto dma = [&](int item) {
        const int n = item / a.P, R0 = (item - n * a.P) * NR;
        const int lo = R0 > 0 ? R0 - 1 : 0, hi = R0 + NR < IW ? R0 + NR : IW - 1;
        const int per_plane = (hi - lo + 1) * 16, total = per_plane * CIN;
        const float* src = a.x + ((size_t)n * CIN * IW + lo) * IW;
        for (int c = wid * 64; c < total; c += 512) {          // wave-uniform trip count; lanes past the end are masked
            const int cc = c + lane;
            if (cc < total) {
                const int pl = cc / per_plane, wi = cc - pl * per_plane;
                __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)pl * IW * IW + wi * 4), (lptr_t)(stg + (size_t)c * 4), 16, 0, 0);
            }
        }
    };
    u32x4_t rwA[2][2], rwB[2][2];
    bool first = true;
#pragma unroll
    for (int t = 0; t < 2; ++t) { rwA[t][0] = rwA[t][1] = rwB[t][0] = rwB[t][1] = u32x4_t{0u, 0u, 0u, 0u}; }
    int item = blockIdx.x;
    if (item < nitems) dma(item);
    for (; item < nitems; item += gridDim.x) {
        const int n = item / a.P, R0 = (item - n * a.P) * NR;
        const int lo = R0 > 0 ? R0 - 1 : 0, hi = R0 + NR < IW ? R0 + NR : IW - 1;
        const int nrow = hi - lo + 1;
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this wave's DMA pieces have landed
        __syncthreads();                                     // ... everybody's; and every wave is done reading the previous item's records
        // ---- staging -> records: one pixel per thread and round (three conflict-free 4-byte reads, one 16-byte write)
        for (int q = tid; q < nrow * IW; q += 512) {
            const int r = q >> 6, xx = q & 63;
            unsigned short sl[8];
            if constexpr (CIN == 1) {
                unsigned short t3[3];
                bf_split3(stg[r * IW + xx], t3);
                sl[0] = t3[0]; sl[1] = t3[1]; sl[2] = t3[2]; sl[3] = t3[0]; sl[4] = t3[1]; sl[5] = t3[0]; sl[6] = 0; sl[7] = 0;
            } else {
                unsigned short hv[CIN], lv[CIN];
#pragma unroll
                for (int c = 0; c < CIN; ++c) {
                    const float v = stg[(c * nrow + r) * IW + xx];
                    hv[c] = bf_hi(v); lv[c] = bf_lo(v);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) sl[i] = i < CIN ? hv[i % CIN] : (i < 2 * CIN ? lv[i % CIN] : (unsigned short)0);
            }
            u32x4_t pr;
            pr.x = sl[0] | ((unsigned)sl[1] << 16); pr.y = sl[2] | ((unsigned)sl[3] << 16); pr.z = sl[4] | ((unsigned)sl[5] << 16); pr.w = sl[6] | ((unsigned)sl[7] << 16);
            *reinterpret_cast<u32x4_t*>(rec + ((size_t)(lo + r + 1) * PWI + xx + 1) * 16) = pr;
        }
        // (a band's halo row outside its own rows may hold the PREVIOUS item's data only if it lies inside the frame -- then it was just
        // rewritten; outside the frame it is record row 0 / 65, zero for ever)
        __syncthreads();
        if (item + (int)gridDim.x < nitems) dma(item + gridDim.x);             // the next frame arrives under this one's arithmetic
        bf16_t* obase = a.raw + (size_t)n * IW * IW * 64 + c0;
        // one output row of this wave: two 32-pixel tiles
        auto row = [&](int y, const u32x4_t (&rw)[2][2]) {
            // operands: five 16-byte reads per tile (inline asm: behind a pending LDS-DMA -- the prefetch of the next frame -- hipcc would
            // put an s_waitcnt vmcnt(0) in front of a C++ LDS load); tile 1's are issued behind tile 0's MFMAs, under tile 0's epilogue
            const unsigned rb = rec_base + (unsigned)((y * PWI + lcol) * 16);
            u32x4_t fb[2][5];
#pragma unroll
            for (int s = 0; s < 5; ++s) asm volatile("ds_read_b128 %0, %1" : "=&v"(fb[0][s]) : "v"(rb + tapoff[s]) : "memory");
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16_t acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    // LDS returns in order: 4 - s reads may still be outstanding when this one is needed
                    switch (4 - s) {
                        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fb[t][s])::"memory"); break;
                        case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fb[t][s])::"memory"); break;
                        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[t][s])::"memory"); break;
                        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fb[t][s])::"memory"); break;
                        default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[t][s])::"memory"); break;
                    }
                    const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, fb[t][s]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][0], bf, acc, 0, 0, 0);
                    if constexpr (CIN != 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][1], bf, acc, 0, 0, 0);
                }
                if (t == 0) {
#pragma unroll
                    for (int s = 0; s < 5; ++s) asm volatile("ds_read_b128 %0, %1" : "=&v"(fb[1][s]) : "v"(rb + 512u + tapoff[s]) : "memory");
                }
                // result layout: column (= pixel) lane & 31, rows (r & 3) + 8 (r >> 2) + 4 kg = channels c0 + r by the permutation above
                u32x4_t o0, o1;
                o0.x = pack2bf(acc[0], acc[1]); o0.y = pack2bf(acc[2], acc[3]); o0.z = pack2bf(acc[4], acc[5]); o0.w = pack2bf(acc[6], acc[7]);
                o1.x = pack2bf(acc[8], acc[9]); o1.y = pack2bf(acc[10], acc[11]); o1.z = pack2bf(acc[12], acc[13]); o1.w = pack2bf(acc[14], acc[15]);
                u32x4_t* op = reinterpret_cast<u32x4_t*>(obase + ((size_t)y * IW + 32 * t + lcol) * 64);
                op[0] = o0; op[1] = o1;
                if constexpr (BNR) {
                    // the BatchNorm-backward sums of the producer (srvp_conv_desc.bnr_*): g = dA * act'(scale raw + shift) on the value AS STORED
                    float da[16], rv[16];
                    unpack8(o0, da); unpack8(o1, da + 8);
                    unpack8(rw[t][0], rv); unpack8(rw[t][1], rv + 8);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float gg = da[e] * (((rv[e] > thr[e]) != neg[e]) ? 1.f : LRELU_SLOPE);
                        s1[e] += gg; s2[e] = fmaf(gg, rv[e], s2[e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) { const float v = acc[e]; s1[e] += v; s2[e] = fmaf(v, v, s2[e]); }
                }
            }
        };
        if constexpr (BNR) {
            // the producer's raw values of a row (this lane: its pixel's 16 channels of both tiles) are loaded ONE ROW AHEAD, across the
            // frame boundary too -- two register sets in turn (a copy between sets would wait for the load it copies)
            auto fetch = [&](int n_, int y_, u32x4_t (&dst)[2][2]) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const u32x4_t* rp = reinterpret_cast<const u32x4_t*>(a.bnr_raw + (((size_t)n_ * IW + y_) * IW + 32 * t + lcol) * 64 + c0);
                    dst[t][0] = __builtin_nontemporal_load(rp); dst[t][1] = __builtin_nontemporal_load(rp + 1);
                }
            };
            // (plain C++ loads: hipcc waits for them with vmcnt(0) at the first use -- a hand-counted wait on inline-asm loads is not an option
            // here, the register sets are live across the branches and back-edges of this loop and hipcc may copy them before the data is there;
            // measured, the launch is not bound by this wait: with or without it 0.80 ms at 2304 frames)
            if (first) { fetch(n, R0 + g, rwA); first = false; }
            for (int y = R0 + g; y < R0 + NR; y += 8) {               // (NR / 4 rows per wave: 16, 8 or 4)
                fetch(n, y + 4, rwB);
                row(y, rwA);
                if (y + 8 < R0 + NR) fetch(n, y + 8, rwA);
                else if (item + (int)gridDim.x < nitems) {
                    const int it2 = item + gridDim.x, n2 = it2 / a.P;
                    fetch(n2, (it2 - n2 * a.P) * NR + g, rwA);
                }
                row(y + 4, rwB);
            }
        } else {
            for (int y = R0 + g; y < R0 + NR; y += 4) row(y, rwA);
        }
        // this item's sums over the 32 pixels of the lane group, widened: lane lcol carries channel c0 + rs16_elem(lcol)
        d1 += (double)rs16_flush(s1, lcol); d2 += (double)rs16_flush(s2, lcol);
    }
    // ---- sums over the waves of a channel tile through LDS
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    double* red = reinterpret_cast<double*>(stg);            // [4 row groups][64 channels][2]
    if (!(lcol & 1)) { red[(g * 64 + c0 + rs16_elem(lcol)) * 2] = d1; red[(g * 64 + c0 + rs16_elem(lcol)) * 2 + 1] = d2; }
    __syncthreads();
    if (tid < 64) {
        double t1 = 0., t2 = 0.;
#pragma unroll
        for (int r = 0; r < 4; ++r) { t1 += red[(r * 64 + tid) * 2]; t2 += red[(r * 64 + tid) * 2 + 1]; }
        if constexpr (BNR) {
            // sum g (raw - mean) inv_std = inv_std (sum g raw - mean sum g), in double from the per-item sums
            const double mu = a.bnr_coef[128 + tid], is = a.bnr_coef[192 + tid];
            atomicAdd(a.bnr_red + tid, t1);
            atomicAdd(a.bnr_red + 64 + tid, is * (t2 - mu * t1));
        } else if (a.stats) {
            atomicAdd(a.stats + tid, t1);
            atomicAdd(a.stats + 64 + tid, t2);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// The fused-sum variant with the producer's raw rows in an LDS RING (round 4, second form).  Measured on the kernel above: arithmetic
// alone 0.29 ms, + the raw loads 0.43, + the stores instead 0.42, both 0.78 -- a persistent 8-wave workgroup with its raw values in
// registers one row ahead has <= 32 KB of reads in flight per CU, half of what 4.5 TB/s x the loaded latency ask for.  Here an item is
// HALF a frame (quarter for few frames), which shrinks the records to 34 rows (36 KB) and the staging area to 26 KB and leaves room for
// three 32 KB ring slots: the four raw rows of iteration j + 2 are requested by LDS-DMA (no registers) while iteration j computes, 64 KB
// in flight per CU.  A slot is lane-linear with the 16-byte channel chunk XOR-swizzled by the pixel on the SOURCE side, so that the
// epilogue's reads (one pixel per lane, two chunks) are conflict-free.  vmcnt is counted by hand per iteration: it retires in order and
// counts the four stores of every row.
constexpr int RG_ROWS = 34;                                   // record rows of an item: 32 + halo
constexpr int RG_REC = RG_ROWS * PWI * 16;                    // 35 904
constexpr int RG_SLOT = 4 * IW * 128;                         // four raw rows: 32 768 bytes
constexpr int RG_NSLOT = 3;

template <int CIN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_in_bnr_ring_kernel(const InStreamK a) {
    __shared__ __attribute__((aligned(16))) unsigned char rec[RG_REC];
    __shared__ __attribute__((aligned(16))) float stg[CIN * RG_ROWS * IW];
    __shared__ __attribute__((aligned(1024))) unsigned char ring[RG_NSLOT * RG_SLOT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lcol = lane & 31, kg = lane >> 5;
    const int jw = wid & 1, g = wid >> 1;
    for (int i = tid; i < RG_REC / 16; i += 512) reinterpret_cast<u32x4_t*>(rec)[i] = u32x4_t{0u, 0u, 0u, 0u};
    // ---- weights (as in conv_in_stream_kernel)
    bf16x8_t wf[5][2];
    {
        const int R = lcol, co = 32 * jw + 16 * ((R >> 2) & 1) + (R & 3) + 4 * (R >> 3);
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int tap = 2 * s + kg;
            unsigned short v1[8], v2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ci = i % CIN;
                const bool ok = tap < 9 && co < a.Cout_real;
                const float wv = ok ? a.w[((size_t)co * CIN + ci) * 9 + (tap < 9 ? tap : 0)] : 0.f;
                if constexpr (CIN == 1) {
                    unsigned short t3[3];
                    bf_split3(wv, t3);
                    v1[i] = i < 3 ? t3[0] : (i < 5 ? t3[1] : (i == 5 ? t3[2] : (unsigned short)0));
                    v2[i] = 0;
                } else {
                    v1[i] = i < 2 * CIN ? bf_hi(wv) : (unsigned short)0;
                    v2[i] = i < 2 * CIN ? bf_lo(wv) : (unsigned short)0;
                }
            }
            u32x4_t p1, p2;
            p1.x = v1[0] | ((unsigned)v1[1] << 16); p1.y = v1[2] | ((unsigned)v1[3] << 16); p1.z = v1[4] | ((unsigned)v1[5] << 16); p1.w = v1[6] | ((unsigned)v1[7] << 16);
            p2.x = v2[0] | ((unsigned)v2[1] << 16); p2.y = v2[2] | ((unsigned)v2[3] << 16); p2.z = v2[4] | ((unsigned)v2[5] << 16); p2.w = v2[6] | ((unsigned)v2[7] << 16);
            wf[s][0] = __builtin_bit_cast(bf16x8_t, p1);
            wf[s][1] = __builtin_bit_cast(bf16x8_t, p2);
        }
#pragma unroll
        for (int s = 0; s < 5; ++s) { asm volatile("" : "+v"(wf[s][0])); asm volatile("" : "+v"(wf[s][1])); }
    }
    unsigned tapoff[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + kg < 9 ? 2 * s + kg : 8;
        tapoff[s] = (unsigned)(((tap / 3) * PWI + tap % 3) * 16);
    }
    const int c0 = 32 * jw + 16 * kg;
    float s1[16], s2[16];                                    // per item, widened per item (as in conv_in_stream_kernel)
#pragma unroll
    for (int e = 0; e < 16; ++e) s1[e] = s2[e] = 0.f;
    double d1 = 0., d2 = 0.;
    float thr[16];
    bool neg[16];
    {
        auto ord = [](float f) { int o = __builtin_bit_cast(int, f); return o < 0 ? (int)(0x80000000u - (unsigned)o) : o; };
        auto unord = [](int o) { return __builtin_bit_cast(float, o < 0 ? (int)(0x80000000u - (unsigned)o) : o); };
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float sc = a.bnr_coef[c0 + e], sh = a.bnr_coef[64 + c0 + e];
            neg[e] = sc < 0.f;
            if (!(sc > 0.f) && !(sc < 0.f)) { thr[e] = sh > 0.f ? -__builtin_huge_valf() : __builtin_huge_valf(); continue; }
            const float as = fabsf(sc);
            int o = ord(bf2f(f2bf(-sh / as))) & ~0xFFFF;
            const int omax = ord(bf2f((unsigned short)0x7F7F)), omin = -omax;
            o = o > omax ? omax : (o < omin ? omin : o);
            for (int it = 0; it < 6 && o > omin && fmaf(unord(o), as, sh) > 0.f; ++it) o -= 0x10000;
            for (int it = 0; it < 6 && o < omax && !(fmaf(unord(o + 0x10000), as, sh) > 0.f); ++it) o += 0x10000;
            thr[e] = neg[e] ? unord(ord(-unord(o)) - 0x10000) : unord(o);
        }
    }
    const int NR = IW / a.P, n_it = NR / 4;                  // P in {2, 4}: 8 or 4 iterations of four rows (one per row group g)
    const int nitems = a.N * a.P;
    const unsigned rec_base = (unsigned)(uintptr_t)rec, ring_base = (unsigned)(uintptr_t)ring;
    auto dma = [&](int item) {
        const int n = item / a.P, R0 = (item - n * a.P) * NR;
        const int lo = R0 > 0 ? R0 - 1 : 0, hi = R0 + NR < IW ? R0 + NR : IW - 1;
        const int per_plane = (hi - lo + 1) * 16, total = per_plane * CIN;
        const float* src = a.x + ((size_t)n * CIN * IW + lo) * IW;
        for (int c = wid * 64; c < total; c += 512) {
            const int cc = c + lane;
            if (cc < total) {
                const int pl = cc / per_plane, wi = cc - pl * per_plane;
                __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)pl * IW * IW + wi * 4), (lptr_t)(stg + (size_t)c * 4), 16, 0, 0);
            }
        }
    };
    // ring piece of this thread inside a row: pixel tid >> 3, LDS chunk position tid & 7 holds source chunk (tid & 7) ^ (pixel & 7)
    const unsigned poff = (unsigned)((tid >> 3) * 64 + (((tid & 7) ^ ((tid >> 3) & 7)) * 8));
    auto ringD = [&](const bf16_t* frame, int R0, int j) {       // rows R0 + 4 j .. + 3 of the producer's raw tensor into slot j % 3
        unsigned char* dst = ring + (j % RG_NSLOT) * RG_SLOT;
        const bf16_t* src = frame + (size_t)(R0 + 4 * j) * IW * 64 + poff;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)k * IW * 64), (lptr_t)(dst + ((size_t)k * 512 + wid * 64) * 16), 16, 0, 0);
    };
    int item = blockIdx.x;
    if (item < nitems) dma(item);
    for (; item < nitems; item += gridDim.x) {
        const int n = item / a.P, R0 = (item - n * a.P) * NR;
        const int lo = R0 > 0 ? R0 - 1 : 0, hi = R0 + NR < IW ? R0 + NR : IW - 1;
        const int nrow = hi - lo + 1;
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): staging pieces landed, every store / ring piece of the previous item retired
        __syncthreads();
        // ---- staging -> records; record row of frame row Y: Y - R0 + 1 (halo above = 0, below = NR + 1; outside the frame: zero)
        for (int q = tid; q < nrow * IW; q += 512) {
            const int r = q >> 6, xx = q & 63;
            unsigned short sl[8];
            if constexpr (CIN == 1) {
                unsigned short t3[3];
                bf_split3(stg[r * IW + xx], t3);
                sl[0] = t3[0]; sl[1] = t3[1]; sl[2] = t3[2]; sl[3] = t3[0]; sl[4] = t3[1]; sl[5] = t3[0]; sl[6] = 0; sl[7] = 0;
            } else {
                unsigned short hv[CIN], lv[CIN];
#pragma unroll
                for (int c = 0; c < CIN; ++c) {
                    const float v = stg[(c * nrow + r) * IW + xx];
                    hv[c] = bf_hi(v); lv[c] = bf_lo(v);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) sl[i] = i < CIN ? hv[i % CIN] : (i < 2 * CIN ? lv[i % CIN] : (unsigned short)0);
            }
            u32x4_t pr;
            pr.x = sl[0] | ((unsigned)sl[1] << 16); pr.y = sl[2] | ((unsigned)sl[3] << 16); pr.z = sl[4] | ((unsigned)sl[5] << 16); pr.w = sl[6] | ((unsigned)sl[7] << 16);
            *reinterpret_cast<u32x4_t*>(rec + ((size_t)(lo + r - R0 + 1) * PWI + xx + 1) * 16) = pr;
        }
        if (tid < IW) {
            const u32x4_t z = u32x4_t{0u, 0u, 0u, 0u};
            if (R0 == 0) *reinterpret_cast<u32x4_t*>(rec + ((size_t)tid + 1) * 16) = z;
            if (R0 + NR == IW) *reinterpret_cast<u32x4_t*>(rec + ((size_t)(NR + 1) * PWI + tid + 1) * 16) = z;
        }
        __syncthreads();
        if (item + (int)gridDim.x < nitems) dma(item + gridDim.x);
        const bf16_t* frame = a.bnr_raw + (size_t)n * IW * IW * 64;
        ringD(frame, R0, 0);
        if (n_it > 1) ringD(frame, R0, 1);
        bf16_t* obase = a.raw + (size_t)n * IW * IW * 64 + c0;
        for (int j = 0; j < n_it; ++j) {
            // ring rows of iteration j: issued two iterations ago; younger operations of this wave: the stores of rows j - 2 and j - 1 (four
            // each) and the ring pieces of iteration j + 1 (four)
            const int younger = (j >= 2 ? 4 : 0) + (j >= 1 ? 4 : 0) + (j + 1 < n_it ? 4 : 0);
            if (younger == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // everybody's pieces; and every wave is past row j - 1 (slot (j + 2) % 3 is free)
            asm volatile("" ::: "memory");
            if (j + 2 < n_it) ringD(frame, R0, j + 2);
            const int y = R0 + 4 * j + g;
            const unsigned rb = rec_base + (unsigned)((((y - R0) * PWI) + lcol) * 16);
            const unsigned sb = ring_base + (unsigned)((j % RG_NSLOT) * RG_SLOT + g * 512 * 16);
            u32x4_t fb[2][5], rw[2][2];
#pragma unroll
            for (int s = 0; s < 5; ++s) asm volatile("ds_read_b128 %0, %1" : "=&v"(fb[0][s]) : "v"(rb + tapoff[s]) : "memory");
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned px = (unsigned)(32 * t + lcol), ck = (unsigned)(c0 / 8 + h);
                    asm volatile("ds_read_b128 %0, %1" : "=&v"(rw[t][h]) : "v"(sb + ((px * 8 + (ck ^ (px & 7))) << 4)) : "memory");
                }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16_t acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    // LDS returns in order.  tile 0: the four ring reads and 4 - s operand reads are younger; tile 1: 4 - s operand reads
                    if (t == 0) {
                        switch (8 - s) {
                            case 8: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fb[t][s])::"memory"); break;
                            case 7: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(fb[t][s])::"memory"); break;
                            case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fb[t][s])::"memory"); break;
                            case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fb[t][s])::"memory"); break;
                            default: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fb[t][s])::"memory"); break;
                        }
                    } else {
                        switch (4 - s) {
                            case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fb[t][s])::"memory"); break;
                            case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fb[t][s])::"memory"); break;
                            case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[t][s])::"memory"); break;
                            case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fb[t][s])::"memory"); break;
                            default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[t][s])::"memory"); break;
                        }
                    }
                    const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, fb[t][s]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][0], bf, acc, 0, 0, 0);
                    if constexpr (CIN != 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][1], bf, acc, 0, 0, 0);
                }
                if (t == 0) {
#pragma unroll
                    for (int s = 0; s < 5; ++s) asm volatile("ds_read_b128 %0, %1" : "=&v"(fb[1][s]) : "v"(rb + 512u + tapoff[s]) : "memory");
                    // the ring reads are older than these five
                    asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(rw[0][0]), "+v"(rw[0][1]), "+v"(rw[1][0]), "+v"(rw[1][1])::"memory");
                }
                u32x4_t o0, o1;
                o0.x = pack2bf(acc[0], acc[1]); o0.y = pack2bf(acc[2], acc[3]); o0.z = pack2bf(acc[4], acc[5]); o0.w = pack2bf(acc[6], acc[7]);
                o1.x = pack2bf(acc[8], acc[9]); o1.y = pack2bf(acc[10], acc[11]); o1.z = pack2bf(acc[12], acc[13]); o1.w = pack2bf(acc[14], acc[15]);
                u32x4_t* op = reinterpret_cast<u32x4_t*>(obase + ((size_t)y * IW + 32 * t + lcol) * 64);
                op[0] = o0; op[1] = o1;
                float da[16], rv[16];
                unpack8(o0, da); unpack8(o1, da + 8);
                unpack8(rw[t][0], rv); unpack8(rw[t][1], rv + 8);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float gg = da[e] * (((rv[e] > thr[e]) != neg[e]) ? 1.f : LRELU_SLOPE);
                    s1[e] += gg; s2[e] = fmaf(gg, rv[e], s2[e]);
                }
            }
        }
        d1 += (double)rs16_flush_plain(s1, lcol); d2 += (double)rs16_flush_plain(s2, lcol);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    double* red = reinterpret_cast<double*>(stg);
    if (!(lcol & 1)) { red[(g * 64 + c0 + rs16_elem(lcol)) * 2] = d1; red[(g * 64 + c0 + rs16_elem(lcol)) * 2 + 1] = d2; }
    __syncthreads();
    if (tid < 64) {
        double t1 = 0., t2 = 0.;
#pragma unroll
        for (int r = 0; r < 4; ++r) { t1 += red[(r * 64 + tid) * 2]; t2 += red[(r * 64 + tid) * 2 + 1]; }
        const double mu = a.bnr_coef[128 + tid], is = a.bnr_coef[192 + tid];
        atomicAdd(a.bnr_red + tid, t1);
        atomicAdd(a.bnr_red + 64 + tid, is * (t2 - mu * t1));
    }
}

int g_in_stream = -1;

}  // namespace

extern "C" int srvp_conv_set_in_stream(int on) { g_in_stream = on ? 1 : 0; return SRVP_OK; }

// 3x3 stride-1 pad-1 image-side layer with 64 output channels on 64x64 frames -> the streaming kernel (SRVP_CONV_IN_STREAM=0: tile kernel)
int srvp_conv_in_stream_launch(const float* x, const float* w, bf16_t* raw, double* stats, int N, int Cin, int Cout, int Cout_real,
                               const bf16_t* bnr_raw, const float* bnr_coef, double* bnr_red, hipStream_t st, int* taken) {
    *taken = 0;
    if (g_in_stream < 0) { const char* e = getenv("SRVP_CONV_IN_STREAM"); g_in_stream = e ? atoi(e) : 1; }
    if (!g_in_stream || Cout != 64 || (Cin != 1 && Cin != 3) || N < 1) return SRVP_OK;
    // 16-byte LDS-DMA pieces of the frames, 16-byte stores / loads of the NHWC tensors: anything less aligned stays on the tile kernel
    if ((((uintptr_t)x) | ((uintptr_t)raw) | ((uintptr_t)bnr_raw)) & 15) return SRVP_OK;
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount; if (ncu <= 0) ncu = 256; }
    InStreamK a;
    a.x = x; a.w = w; a.raw = raw; a.stats = stats; a.bnr_raw = bnr_raw; a.bnr_coef = bnr_coef; a.bnr_red = bnr_red;
    a.N = N; a.Cout_real = Cout_real;
    a.P = N >= 4 * ncu ? 1 : (N >= 2 * ncu ? 2 : 4);
    const long long items = (long long)N * a.P;
    const dim3 grid((unsigned)(items < ncu ? items : ncu)), blk(512);
    static int ring_on = -1;
    if (ring_on < 0) { const char* e = getenv("SRVP_CONV_IN_BNR_RING"); ring_on = e ? atoi(e) : 1; }
    if (bnr_red && ring_on) {
        // the raw rows of the producer in an LDS ring: items are half frames (quarter frames when there are few)
        a.P = N >= 2 * ncu ? 2 : 4;
        const long long it2 = (long long)N * a.P;
        const dim3 grid2((unsigned)(it2 < ncu ? it2 : ncu));
        if (Cin == 3) hipLaunchKernelGGL((conv_in_bnr_ring_kernel<3>), grid2, blk, 0, st, a);
        else hipLaunchKernelGGL((conv_in_bnr_ring_kernel<1>), grid2, blk, 0, st, a);
    } else if (bnr_red) {
        if (Cin == 3) hipLaunchKernelGGL((conv_in_stream_kernel<3, true>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((conv_in_stream_kernel<1, true>), grid, blk, 0, st, a);
    } else {
        if (Cin == 3) hipLaunchKernelGGL((conv_in_stream_kernel<3, false>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((conv_in_stream_kernel<1, false>), grid, blk, 0, st, a);
    }
    *taken = 1;
    return SRVP_OK;
}
