// Streaming 3x3 convolution for the 64 -> 64 channel layers at 64x64 resolution (VGG encoder.conv.0.1, reference module/conv.py:200-203:
// forward with BatchNorm statistics, and its data gradient with the producer's BatchNorm-backward sums fused in).
//
// These layers sit at the ridge of the roofline (288 FLOP per byte): 2.4 GB of activations per pass at 2304 frames against 0.7 TFLOP.
// On the 256-pixel tile kernel (conv_mfma.hip) they ran at 640-750 TFLOP/s, 0.92 / 1.09 ms per pass: a tile's K loop is only nine taps of
// one 64-channel chunk -- 1.9 us of matrix work inside ~8 us of serial prologue / patch DMA / epilogue latencies, 36864 times, with
// 27 % of the input re-read as halo.  Here (the dataflow of conv_out.hip, on 32x32x16 tiles): ONE persistent workgroup per CU walks
// whole images top to bottom with a ROLLING window of input rows in LDS -- three groups of four 66-pixel rows, filled by LDS-DMA one
// group ahead -- so every input byte crosses HBM -> LDS exactly once and the DMA runs under the previous band's arithmetic.
//   * The WEIGHTS are stationary: wave w owns output channels [32 (w & 1), +32) and columns [32 (w >> 1), +32) of the band's four rows; its
//     36 B fragments (9 taps x 4 sixteen-channel slices, fragment-major packing: one 16-byte load per lane and fragment) live in 144
//     registers for every image the workgroup processes (one wave per SIMD: 512 registers per lane).
//   * An A fragment (32 pixels of input row R at column offset dx, 16 channels: one ds_read_b128 per lane, chunk index XOR-swizzled by
//     the pixel) is read once and feeds the up to three output rows R - dy it contributes to: 72 reads and 144 MFMAs per band and wave.
//   * Epilogue per band (four full output rows = one contiguous 32 KB run of the NHWC tensor): accumulators -> bf16 -> LDS staging ->
//     16-byte coalesced stores; forward: per-channel sum / sum of squares from the fp32 accumulators, kept in registers (fp64) for the
//     whole item and added with ONE atomic pair per channel, wave and item (16x fewer atomics than one per 256-pixel tile);
//     data gradient: the producer's BatchNorm-backward sums (srvp_conv_desc.bnr_*) formed in the copy-out loop exactly as in
//     conv_mfma.hip's epilogue, per-thread partial sums carried over the item.
// Same arithmetic as the tile kernels (bf16 operands, fp32 accumulation, one bf16 rounding of the output); the K order differs
// (input row outer, tap inner), i.e. results agree to fp32 summation order.
#include "common.h"
#include "../../include/srvp_hip.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int SW = 64, SPW = 66, SROWB = SPW * 128;           // padded row: 66 pixels x 64 channels x 2 bytes
constexpr int SGROUP_ROWS = 4, SNRING = 3, SGROUP_B = SGROUP_ROWS * SROWB;
constexpr int SSLOTS = SGROUP_ROWS * SPW * 8;                 // 16-byte pieces per group = 2112 = 8.25 x 256
constexpr int SLDC = 64 + 8;                                  // staging row pitch (bf16 elements)
constexpr int SBAND_PX = SGROUP_ROWS * SW;                    // 256 output pixels per band

struct StreamK {
    const bf16_t* src; const bf16_t* wt; bf16_t* dst;
    double* stats;
    const bf16_t* bnr_raw; const float* bnr_coef; double* bnr_red;
    int N, HS;
    unsigned long long tapmap64;   // 9 x 4 bits: packed-weight tap index of the padded offset (dy, dx), entry dy * 3 + dx
};

template <bool BNR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_stream64_kernel(const StreamK a) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[SNRING * SGROUP_B];
    __shared__ __attribute__((aligned(16))) bf16_t cs[SBAND_PX * SLDC];          // [256 px][72]; after an item: partial sums of the fused reduction
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nt = wid & 1, ph = wid >> 1;
    const int lcol = lane & 31, lhalf = lane >> 5;
    // ---- DMA pieces of this thread inside a group (a group = four padded rows; group g lives in ring slot g % 3): slot q = i * 256 + tid
    // -> (row r, pixel px, LDS chunk position s) holds the source chunk s ^ (px & 7) of that pixel
    unsigned soff[9];
    bool svalid[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int q = i * 256 + tid;
        svalid[i] = q < SSLOTS;
        const int qq = svalid[i] ? q : 0;
        const int r = qq / (SPW * 8), rem = qq - r * (SPW * 8), px = rem >> 3, s = rem & 7;
        soff[i] = (unsigned)(((r * SPW + px) * 64) + ((s ^ (px & 7)) * 8));
    }
    const bf16_t* img = a.src;
    auto stage = [&](int g) {
        const int row0 = g * SGROUP_ROWS;
        unsigned char* dst = ring + (g % SNRING) * SGROUP_B;
        const bf16_t* src = img + (size_t)row0 * SPW * 64;
        const int valid_rows = SPW - row0 < SGROUP_ROWS ? SPW - row0 : SGROUP_ROWS;
        const unsigned lim = (unsigned)valid_rows * SPW * 64;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (i == 8 && wid != 0) continue;                                          // the ninth round is a quarter round (wave 0 only)
            const unsigned o = (svalid[i] && soff[i] < lim) ? soff[i] : 0u;            // rows that do not exist: re-read the first pixel (never used)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + o), (lptr_t)(dst + ((size_t)i * 256 + wid * 64) * 16), 16, 0, 0);
        }
    };
    // ---- B fragments of this wave's 32 output channels, by padded offset o = dy * 3 + dx and 16-channel slice kk
    bf16x8_t wf[9][4];
    {
        const bf16_t* wl = a.wt + (size_t)nt * 512 + lane * 8;
#pragma unroll
        for (int o = 0; o < 9; ++o) {
            const int t = (int)((a.tapmap64 >> (4 * o)) & 15);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wf[o][kk] = *reinterpret_cast<const bf16x8_t*>(wl + (size_t)((t * 4 + kk) * 2) * 512);
        }
#pragma unroll
        for (int o = 0; o < 9; ++o)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(wf[o][kk]));
    }
    const unsigned ring_base = (unsigned)(uintptr_t)ring;
    // lane-constant part of the fragment addresses: column offset dx and 16-channel slice kk (pixel byte offset + swizzled chunk)
    unsigned swz[3][4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int px = ph * 32 + dx + lcol;
            swz[dx][kk] = (unsigned)(px * 128) + ((((unsigned)(kk * 2 + lhalf)) ^ (unsigned)(px & 7)) << 4);
        }
    const int nb = (SW / SGROUP_ROWS) / a.HS;                    // bands per item
    const int cch = tid & 7;                                     // this thread's 16-byte channel chunk in the copy-out loop
    float csc[8], csh[8], cmu[8], cis[8];
    if constexpr (BNR) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cch * 8 + e;
            csc[e] = a.bnr_coef[c]; csh[e] = a.bnr_coef[64 + c]; cmu[e] = a.bnr_coef[128 + c]; cis[e] = a.bnr_coef[192 + c];
        }
    }
    for (int item = blockIdx.x; item < a.N * a.HS; item += gridDim.x) {
        const int n = item / a.HS, b0 = (item - n * a.HS) * nb, b1 = b0 + nb;
        img = a.src + (size_t)n * SPW * SPW * 64;
        bf16_t* obase = a.dst + (size_t)n * SW * SW * 64;
        const bf16_t* rbase = BNR ? a.bnr_raw + (size_t)n * SW * SW * 64 : nullptr;
        double d1 = 0., d2 = 0.;                               // forward statistics of this lane's column over the item
        float g1[8], g2[8];                                    // fused reduction: this thread's eight channels over the item
#pragma unroll
        for (int e = 0; e < 8; ++e) { g1[e] = 0.f; g2[e] = 0.f; }
        __syncthreads();                                       // previous item: every wave is past its last ring / staging access
        stage(b0);
        stage(b0 + 1);
        for (int b = b0; b < b1; ++b) {
            __builtin_amdgcn_s_barrier();                      // every wave is done with band b - 1 (ring slot of group b - 1, staging tile)
            asm volatile("" ::: "memory");
            if (b + 2 <= b1) {
                stage(b + 2);
                // groups <= b + 1 have landed when only what was issued after group b + 1 is outstanding: this group's 9 (8) pieces and,
                // from the second band on, the 8 stores of the previous band's copy-out (vmcnt counts stores too and retires in order:
                // a count that excluded them made every band wait for the store acknowledgements)
                // (BNR: + the previous band's 8 loads of the producer's raw values, issued behind group b + 1 as well)
                if (b > b0) {
                    if constexpr (BNR) { if (wid == 0) __builtin_amdgcn_s_waitcnt(0x4F70 | 9); else __builtin_amdgcn_s_waitcnt(0x4F70 | 8); }   // vmcnt(25) / (24)
                    else { if (wid == 0) __builtin_amdgcn_s_waitcnt(0x4F70 | 1); else __builtin_amdgcn_s_waitcnt(0x4F70); }                  // vmcnt(17) / (16)
                }
                else if (wid == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 9); else __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int y0 = b * SGROUP_ROWS;
            // BNR: the band's eight pieces of the producer's raw output (consumed in the copy-out loop) are requested NOW, so that they
            // travel under the band's MFMA phase instead of in two exposed rounds of four behind it
            u32x4_t rwb[BNR ? 8 : 1];
            if constexpr (BNR) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    rwb[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(rbase + (size_t)y0 * SW * 64 + ((size_t)i * 256 + tid) * 8));
            }
            f32x16_t acc[SGROUP_ROWS];
#pragma unroll
            for (int r = 0; r < SGROUP_ROWS; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
            // fragment reads as inline asm (a C++ LDS load behind a pending LDS-DMA gets an s_waitcnt vmcnt(0) from hipcc: the band would
            // wait for the group it has just started to prefetch); PF steps in flight ahead of the MFMAs (LDS returns in order)
            constexpr int NSTEP = (SGROUP_ROWS + 2) * 3 * 4;
            constexpr int PF = 3;                                                      // fragment reads in flight ahead of the MFMAs (an LDS read
            u32x4_t fa[PF + 1];                                                        // returns after ~150+ cycles, a step issues 1-3 MFMAs of 32)
            unsigned rowb[SGROUP_ROWS + 2];                                            // LDS byte address of padded input row y0 + R (wave-uniform)
#pragma unroll
            for (int R = 0; R < SGROUP_ROWS + 2; ++R) rowb[R] = ring_base + (unsigned)(((b + (R >> 2)) % SNRING) * SGROUP_B + (R & 3) * SROWB);
            auto issue = [&](int k, u32x4_t& v) {
                const int R = k / 12, dx = (k - R * 12) >> 2, kk = k & 3;
                const unsigned a0 = rowb[R] + swz[dx][kk];
                asm volatile("ds_read_b128 %0, %1" : "=&v"(v) : "v"(a0) : "memory");
            };
            auto mac = [&](int k, const u32x4_t& v) {
                const int R = k / 12, dx = (k - R * 12) >> 2, kk = k & 3;
                const bf16x8_t af = __builtin_bit_cast(bf16x8_t, v);
#pragma unroll
                for (int r = 0; r < SGROUP_ROWS; ++r) {
                    const int dy = R - r;
                    if (dy < 0 || dy > 2) continue;
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf[dy * 3 + dx][kk], acc[r], 0, 0, 0);
                }
            };
#pragma unroll
            for (int k = 0; k < PF; ++k) issue(k, fa[k]);
#pragma unroll
            for (int k = 0; k < NSTEP; ++k) {
                u32x4_t& cur = fa[k % (PF + 1)];
                if (k + PF < NSTEP) issue(k + PF, fa[(k + PF) % (PF + 1)]);
                // the reads of steps k + 1 .. min(k + PF, NSTEP - 1) may still be in flight (LDS returns in order)
                const int younger = (k + PF < NSTEP ? k + PF : NSTEP - 1) - k;
                if (younger >= 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(cur)::"memory");
                else if (younger == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(cur)::"memory");
                else if (younger == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur)::"memory");
                else if (younger == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(cur)::"memory");
                else if (younger == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(cur)::"memory");
                else if (younger == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cur)::"memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur)::"memory");
                mac(k, cur);
            }
            // ---- epilogue.  C layout of the 32x32 MFMA: column (output channel) lane & 31, rows (pixels) (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
            if (a.stats) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < SGROUP_ROWS; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const float v = acc[r][i]; s1 += v; s2 += v * v; }
                d1 += (double)s1; d2 += (double)s2;
            }
#pragma unroll
            for (int r = 0; r < SGROUP_ROWS; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int pix = r * SW + ph * 32 + (i & 3) + 8 * (i >> 2) + 4 * lhalf;
                    cs[pix * SLDC + nt * 32 + lcol] = f2bf(acc[r][i]);
                }
            __syncthreads();
            // copy-out: the band is one contiguous run of 256 pixels x 128 bytes; piece q = i * 256 + tid = (pixel q >> 3, chunk tid & 7)
            bf16_t* ob = obase + (size_t)y0 * SW * 64;
#pragma unroll
            for (int h = 0; h < 2; ++h) {                      // (two rounds of four pieces: loads of a round in flight together)
                u32x4_t v[4], rw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int q = (h * 4 + i) * 256 + tid;
                    v[i] = *reinterpret_cast<const u32x4_t*>(cs + (q >> 3) * SLDC + cch * 8);
                    if constexpr (BNR) rw[i] = rwb[h * 4 + i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int q = (h * 4 + i) * 256 + tid;
                    *reinterpret_cast<u32x4_t*>(ob + (size_t)q * 8) = v[i];
                    if constexpr (BNR) {
                        float da[8], rv[8];
                        unpack8(v[i], da);
                        unpack8(rw[i], rv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float gg = da[e] * ((rv[e] * csc[e] + csh[e]) > 0.f ? 1.f : LRELU_SLOPE);
                            g1[e] += gg; g2[e] += gg * (rv[e] - cmu[e]) * cis[e];
                        }
                    }
                }
            }
        }
        // ---- item totals
        if (a.stats) {
            d1 += __shfl_xor(d1, 32);
            d2 += __shfl_xor(d2, 32);
            if (lhalf == 0) {
                atomicAdd(a.stats + nt * 32 + lcol, d1);
                atomicAdd(a.stats + 64 + nt * 32 + lcol, d2);
            }
        }
        if constexpr (BNR) {
            __syncthreads();                                   // every thread is done with the staging tile: it now holds the partial sums
            float* Ps = reinterpret_cast<float*>(cs);          // [32 pixel groups][64 channels][2]
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                Ps[((tid >> 3) * 64 + cch * 8 + e) * 2 + 0] = g1[e];
                Ps[((tid >> 3) * 64 + cch * 8 + e) * 2 + 1] = g2[e];
            }
            __syncthreads();
            if (tid < 64) {
                double t1 = 0., t2 = 0.;
#pragma unroll
                for (int r = 0; r < 32; ++r) { t1 += Ps[(r * 64 + tid) * 2]; t2 += Ps[(r * 64 + tid) * 2 + 1]; }
                atomicAdd(a.bnr_red + tid, t1);
                atomicAdd(a.bnr_red + 64 + tid, t2);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Sub-pixel stage entry at 64 channels (decoder.conv.3.0, reference module/conv.py:331-349 + srvp.py:222-223): nearest x2 upsample + 3x3
// conv of the 32x32 main input, evaluated as four output phases with 2x2 folded taps (DESIGN "Sub-pixel form"), plus the hoisted skip
// half S[sample] (fp32) in the accumulators, BatchNorm statistics, bf16 raw output at 64x64.  K = 4 x 64 per output pixel: on the tile
// kernel the launch was all prologue / epilogue (0.82 ms at 2304 frames; HBM floor 0.33 ms), and its largest stream was the S tile:
// 64 KB of fp32 per 256 output pixels, re-read for every frame.
// Here a work item is (sample b, band of four output rows): the band's S values are loaded ONCE into 64 registers per lane and
// seed the accumulators of all T frames t * B + b of that sample (2.4 GB of L2 reads -> 0.2 GB); per frame the item needs four
// low-resolution input rows (17 KB by LDS-DMA, double-buffered: the rows of frame t + 1 arrive under the arithmetic of frame t) and
// writes one contiguous 32 KB run of the raw tensor.  Wave w = (32 output channels nt, column phase b): its 2 row phases x 4 taps x 4
// sixteen-channel slices = 32 weight fragments stay in 128 registers; an input fragment (32 low-resolution pixels of row R at column
// offset b + v) is read once and feeds every (row phase a, tap row u, output row) combination with y + a + u = R: 32 reads, 64 MFMAs
// per frame and wave.  Same products as the four phase launches of the tile kernel; fp32 summation order differs.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int UW = 32, UPW = 34, UROWB = UPW * 128;           // low-resolution padded row: 34 pixels x 64 channels x 2 bytes
constexpr int UGROUP_B = 4 * UROWB;                           // the four input rows of a band
constexpr int USLOTS = 4 * UPW * 8;                           // 1088 sixteen-byte pieces = 4.25 x 256

struct SubK {
    const bf16_t* src; const bf16_t* wt; bf16_t* dst; const float* S; double* stats;
    int N, B;                  // frames t * B + b, samples
    int TS;                    // a (sample, band)'s T frames are shared out over TS work items (few samples: inference)
    const float* ep_coef; int ep_border;       // inference: LeakyReLU(scale * acc + shift) into a tensor with an ep_border-pixel border (srvp_conv_desc.ep_*)
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_stream_sub64_kernel(const SubK a) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[2 * UGROUP_B];
    __shared__ __attribute__((aligned(16))) bf16_t cs[SBAND_PX * SLDC];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nt = wid & 1, bph = wid >> 1;                   // 32 output channels, column phase of this wave
    const int lcol = lane & 31, lhalf = lane >> 5;
    unsigned soff[5];
    bool svalid[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int q = i * 256 + tid;
        svalid[i] = q < USLOTS;
        const int qq = svalid[i] ? q : 0;
        const int r = qq / (UPW * 8), rem = qq - r * (UPW * 8), px = rem >> 3, s = rem & 7;
        soff[i] = (unsigned)(((r * UPW + px) * 64) + ((s ^ (px & 7)) * 8));
    }
    // weight fragments: [row phase a][tap u * 2 + v][slice kk]; packed tap index (a * 2 + bph) * 4 + u * 2 + v, fragment-major
    bf16x8_t wf[2][4][4];
    {
        const bf16_t* wl = a.wt + (size_t)nt * 512 + lane * 8;
#pragma unroll
        for (int ap = 0; ap < 2; ++ap)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    wf[ap][tp][kk] = *reinterpret_cast<const bf16x8_t*>(wl + (size_t)((((ap * 2 + bph) * 4 + tp) * 4 + kk) * 2) * 512);
#pragma unroll
        for (int ap = 0; ap < 2; ++ap)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(wf[ap][tp][kk]));
    }
    const unsigned ring_base = (unsigned)(uintptr_t)ring;
    unsigned swz[2][4];                                       // column offset bph + v, slice kk: pixel byte offset + swizzled chunk
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int px = lcol + bph + v;
            swz[v][kk] = (unsigned)(px * 128) + ((((unsigned)(kk * 2 + lhalf)) ^ (unsigned)(px & 7)) << 4);
        }
    const int cch = tid & 7;
    const int T = a.N / a.B;
    double d1 = 0., d2 = 0.;
    const float ep_sc = a.ep_coef ? a.ep_coef[nt * 32 + lcol] : 1.f, ep_sh = a.ep_coef ? a.ep_coef[64 + nt * 32 + lcol] : 0.f;
    const int Tper = (T + a.TS - 1) / a.TS;
    for (int item0 = blockIdx.x; item0 < a.B * 16 * a.TS; item0 += gridDim.x) {
        const int ts = item0 % a.TS, item = item0 / a.TS;
        const int b = item >> 4, band = item & 15;            // output rows 4 band .. + 3 = low-resolution rows 2 band, 2 band + 1
        const int t_lo = ts * Tper, t_hi = t_lo + Tper < T ? t_lo + Tper : T;
        if (t_lo >= t_hi) continue;
        // ---- the hoisted skip half of this band: tile m = yl * 2 + ap (output row 4 band + 2 yl + ap), this lane's 16 pixels x its channel
        f32x16_t sreg[4];
        {
            const float* sp = a.S + ((size_t)b * 64 + 4 * band) * 64 * 64 + nt * 32 + lcol;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int xl = (i & 3) + 8 * (i >> 2) + 4 * lhalf;
                    sreg[m][i] = sp[((size_t)((m >> 1) * 2 + (m & 1)) * 64 + 2 * xl + bph) * 64];
                }
        }
        auto stage = [&](int t, int buf) {
            const bf16_t* src = a.src + ((size_t)(t * a.B + b) * UPW + 2 * band) * UPW * 64;
            unsigned char* dst = ring + buf * UGROUP_B;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const unsigned o = svalid[i] ? soff[i] : 0u;
                if (i != 4 || wid == 0)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + o), (lptr_t)(dst + ((size_t)i * 256 + wid * 64) * 16), 16, 0, 0);
            }
        };
        __syncthreads();                                       // previous item: every wave is past its last ring / staging access
        stage(t_lo, t_lo & 1);
        for (int t = t_lo; t < t_hi; ++t) {
            __builtin_amdgcn_s_barrier();                      // every wave is done with frame t - 1 (the other ring buffer, the staging tile)
            asm volatile("" ::: "memory");
            if (t + 1 < t_hi) {
                stage(t + 1, (t + 1) & 1);
                if (wid == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 5); else __builtin_amdgcn_s_waitcnt(0x0F70 | 4);      // frame t's rows landed
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            f32x16_t acc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = sreg[m];
            constexpr int NSTEP = 4 * 2 * 4, PF = 3;           // (input row R, column offset v, slice kk)
            u32x4_t fa[PF + 1];
            const unsigned bufb = ring_base + (unsigned)((t & 1) * UGROUP_B);
            auto issue = [&](int k, u32x4_t& v) {
                const int R = k >> 3, cv = (k >> 2) & 1, kk = k & 3;
                const unsigned a0 = bufb + (unsigned)(R * UROWB) + swz[cv][kk];
                asm volatile("ds_read_b128 %0, %1" : "=&v"(v) : "v"(a0) : "memory");
            };
            auto mac = [&](int k, const u32x4_t& v) {
                const int R = k >> 3, cv = (k >> 2) & 1, kk = k & 3;
                const bf16x8_t af = __builtin_bit_cast(bf16x8_t, v);
#pragma unroll
                for (int yl = 0; yl < 2; ++yl)
#pragma unroll
                    for (int ap = 0; ap < 2; ++ap) {
                        const int u = R - yl - ap;             // padded input row (2 band + yl) + ap + u = 2 band + R
                        if (u < 0 || u > 1) continue;
                        acc[yl * 2 + ap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf[ap][u * 2 + cv][kk], acc[yl * 2 + ap], 0, 0, 0);
                    }
            };
#pragma unroll
            for (int k = 0; k < PF; ++k) issue(k, fa[k]);
#pragma unroll
            for (int k = 0; k < NSTEP; ++k) {
                u32x4_t& cur = fa[k % (PF + 1)];
                if (k + PF < NSTEP) issue(k + PF, fa[(k + PF) % (PF + 1)]);
                const int younger = (k + PF < NSTEP ? k + PF : NSTEP - 1) - k;
                if (younger >= 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(cur)::"memory");
                else if (younger == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(cur)::"memory");
                else if (younger == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cur)::"memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur)::"memory");
                mac(k, cur);
            }
            if (a.stats) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const float v = acc[m][i]; s1 += v; s2 += v * v; }
                d1 += (double)s1; d2 += (double)s2;
            }
            if (a.ep_coef) {                                   // inference: eval-mode BatchNorm + LeakyReLU from the fp32 accumulators
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const float v = __builtin_fmaf(acc[m][i], ep_sc, ep_sh); acc[m][i] = v > 0.f ? v : LRELU_SLOPE * v; }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int xl = (i & 3) + 8 * (i >> 2) + 4 * lhalf;
                    const int pix = m * SW + 2 * xl + bph;     // (tile m = output row 4 band + m)
                    cs[pix * SLDC + nt * 32 + lcol] = f2bf(acc[m][i]);
                }
            __syncthreads();
            // copy-out: piece i * 256 + tid = (output row i >> 1 of the band, column (i & 1) * 32 + tid / 8, chunk tid & 7); without a
            // border the band is one contiguous 32 KB run
            const int eb = a.ep_coef ? a.ep_border : 0, PW2 = SW + 2 * eb;
            bf16_t* ob = a.dst + (((size_t)(t * a.B + b) * PW2 + 4 * band + eb) * PW2 + eb) * 64 + cch * 8;
            u32x4_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const u32x4_t*>(cs + (i * 32 + (tid >> 3)) * SLDC + cch * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4_t*>(ob + ((size_t)(i >> 1) * PW2 + (i & 1) * 32 + (tid >> 3)) * 64) = v[i];
        }
    }
    if (a.stats) {                                             // one atomic pair per channel, wave and WORKGROUP (all its items)
        d1 += __shfl_xor(d1, 32);
        d2 += __shfl_xor(d2, 32);
        if (lhalf == 0) {
            atomicAdd(a.stats + nt * 32 + lcol, d1);
            atomicAdd(a.stats + 64 + nt * 32 + lcol, d2);
        }
    }
}

}  // namespace

static int g_stream64 = -1;      // -1: SRVP_CONV_STREAM64 env (default 1)
// A/B switch (tests): 1 = 64 -> 64 channel 64x64 launches on the streaming kernel (default), 0 = on the tile kernels.  Same results up to
// fp32 summation order.
extern "C" int srvp_conv_set_stream64(int on) { g_stream64 = on ? 1 : 0; return SRVP_OK; }
// launches taken by the streaming kernels since the library was loaded (host counters; tests assert that a case really went through them):
// which = 0: conv_stream64_kernel<false> (forward / plain data gradient), 1: conv_stream64_kernel<true> (data gradient + fused BatchNorm-backward
// sums), 2: conv_stream_sub64_kernel
static long long g_stream_count[3] = {0, 0, 0};
extern "C" long long srvp_conv_stream_count(int which) { return which >= 0 && which < 3 ? g_stream_count[which] : -1; }

// Called by srvp_conv_mfma before the tile kernels: *taken = 1 if this launch was handled here.
int srvp_conv_stream64_launch(const srvp_conv_desc* d, hipStream_t st, int* taken) {
    *taken = 0;
    static int minN = -1;
    if (g_stream64 < 0) { const char* e = getenv("SRVP_CONV_STREAM64"); g_stream64 = e ? atoi(e) : 1; }
    if (minN < 0) { const char* m = getenv("SRVP_CONV_STREAM64_MIN_N"); minN = m ? atoi(m) : 96; }
    if (!g_stream64 || d->elem_f32 || d->splitk > 1 || d->C1 != 0 || d->C0 != 64 || d->Cout != 64 || d->ntaps != 9 || d->si != 1 || d->so != 1 || d->ooy != 0 ||
        d->oox != 0 || d->ups0 || d->map0 || d->add_f32 || d->dst_is_f32 || d->out_f32 || d->tap_phase_chunks || d->ep_coef || d->wt_fragmajor != 1 ||
        d->OH != 64 || d->OW != 64 || d->H0p != 66 || d->W0p != 66 || d->DHp != 64 || d->DWp != 64 || d->Cdst != 64 || d->cdst_off != 0 || d->f32_quad ||
        (d->stats && d->stat_mod != 64) || d->N < minN || !d->dst)
        return SRVP_OK;
    if (d->bnr_red && (!d->bnr_raw || !d->bnr_coef || d->stats)) return SRVP_OK;
    unsigned long long map = 0, seen = 0;
    for (int t = 0; t < 9; ++t) {
        if (d->dy[t] < 0 || d->dy[t] > 2 || d->dx[t] < 0 || d->dx[t] > 2) return SRVP_OK;
        const int o = d->dy[t] * 3 + d->dx[t];
        if ((seen >> o) & 1) return SRVP_OK;
        seen |= 1ull << o;
        map |= (unsigned long long)t << (4 * o);
    }
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount; if (ncu <= 0) ncu = 256; }
    StreamK k{};
    k.src = (const bf16_t*)d->src0; k.wt = (const bf16_t*)d->wt; k.dst = (bf16_t*)d->dst; k.stats = d->stats;
    k.bnr_raw = (const bf16_t*)d->bnr_raw; k.bnr_coef = d->bnr_coef; k.bnr_red = d->bnr_red;
    k.N = d->N; k.HS = d->N >= 4 * ncu ? 1 : (d->N >= 2 * ncu ? 2 : 4);
    k.tapmap64 = map;
    const long long items = (long long)k.N * k.HS;
    const dim3 g((unsigned)(items < ncu ? items : ncu)), b(256);
    if (d->bnr_red) hipLaunchKernelGGL(conv_stream64_kernel<true>, g, b, 0, st, k);
    else hipLaunchKernelGGL(conv_stream64_kernel<false>, g, b, 0, st, k);
    SRVP_CHECK_LAUNCH("srvp_conv_mfma(stream64)");
    ++g_stream_count[d->bnr_red ? 1 : 0];
    *taken = 1;
    return SRVP_OK;
}

// Called by srvp_conv_mfma_multi: the four phase launches of a 64-channel sub-pixel stage entry with a hoisted skip half as ONE streaming
// launch (*taken = 1), if they are exactly that.
int srvp_conv_stream_sub64_launch(const srvp_conv_desc* d, int n, hipStream_t st, int* taken) {
    *taken = 0;
    static int on = -1, minN = -1;
    if (on < 0) { const char* e = getenv("SRVP_CONV_STREAM_SUB64"); on = e ? atoi(e) : 1; }
    if (minN < 0) { const char* m = getenv("SRVP_CONV_STREAM64_MIN_N"); minN = m ? atoi(m) : 96; }
    if (!on || g_stream64 == 0 || n != 4) return SRVP_OK;
    const srvp_conv_desc& d0 = d[0];
    if (d0.N < minN || !d0.add_f32 || d0.add_mod <= 0 || d0.N % d0.add_mod != 0 || !d0.dst) return SRVP_OK;
    for (int ph = 0; ph < 4; ++ph) {
        const srvp_conv_desc& p = d[ph];
        const int ap = ph >> 1, bp = ph & 1;
        if (p.elem_f32 || p.splitk > 1 || p.C1 != 0 || p.C0 != 64 || p.Cout != 64 || p.ntaps != 4 || p.si != 1 || p.so != 2 || p.ooy != ap || p.oox != bp ||
            p.ups0 || p.map0 || p.dst_is_f32 || p.out_f32 || p.tap_phase_chunks || p.bnr_red || p.wt_fragmajor != 1 || p.f32_quad ||
            p.ep_coef != d0.ep_coef || (p.ep_coef && (p.ep_act != ACT_LRELU || p.ep_border != d0.ep_border || p.ep_border < 0 || p.ep_border > 1 || p.stats)) ||
            p.OH != 32 || p.OW != 32 || p.H0p != 34 || p.W0p != 34 || p.DHp != 64 || p.DWp != 64 || p.Cdst != 64 || p.cdst_off != 0 ||
            p.N != d0.N || p.src0 != d0.src0 || p.dst != d0.dst || p.add_f32 != d0.add_f32 || p.add_mod != d0.add_mod || p.stats != d0.stats ||
            (p.stats && p.stat_mod != 64) || (const char*)p.wt != (const char*)d0.wt + (size_t)ph * 4 * 64 * 64 * 2)
            return SRVP_OK;
        for (int u = 0; u < 2; ++u)
            for (int v = 0; v < 2; ++v)
                if (p.dy[u * 2 + v] != ap + u || p.dx[u * 2 + v] != bp + v) return SRVP_OK;
    }
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount; if (ncu <= 0) ncu = 256; }
    SubK k{};
    k.src = (const bf16_t*)d0.src0; k.wt = (const bf16_t*)d0.wt; k.dst = (bf16_t*)d0.dst; k.S = d0.add_f32; k.stats = d0.stats;
    k.N = d0.N; k.B = d0.add_mod; k.ep_coef = d0.ep_coef; k.ep_border = d0.ep_border;
    const int T = k.N / k.B;
    k.TS = 1;
    while ((long long)k.B * 16 * k.TS < 2ll * ncu && T / (2 * k.TS) >= 4) k.TS *= 2;      // few samples: share a (sample, band)'s frames out
    const long long items = (long long)k.B * 16 * k.TS;
    hipLaunchKernelGGL(conv_stream_sub64_kernel, dim3((unsigned)(items < ncu ? items : ncu)), dim3(256), 0, st, k);
    SRVP_CHECK_LAUNCH("srvp_conv_mfma_multi(stream sub64)");
    ++g_stream_count[2];
    *taken = 1;
    return SRVP_OK;
}
