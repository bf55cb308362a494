// Weight gradient of the tap-table convolution on the CDNA4 matrix cores.
//
//   dW[t][j][c] += sum_p dout_t[p][j] * in_t[p][c]          p = (n, oy, ox)
//
// A "TN" GEMM: the reduction index (pixels) is the slow index of both NHWC operands, so each K step stages a
// [BP pixels][channels] tile of both operands in LDS and the MFMA fragments (8 consecutive pixels of one channel)
// are read through the gfx950 LDS transpose-read ds_read_b64_tr_b16 (USE_TR) or, as a conservative fallback,
// eight 16-bit LDS reads.  Split-K over pixel chunks, fp32 atomic accumulation into the tap-major gradient.
//
// Replaces the implicit weight-gradient kernels behind autograd for nn.Conv2d / nn.ConvTranspose2d
// (reference module/conv.py:174-179, 200-223, 299-304, 330-353; train.py:109-119 backward).
#include "common.h"
#include "../../include/srvp_hip.h"

int srvp_wgrad_f32_launch(const srvp_wgrad_desc* d, hipStream_t st);   // conv_f32.hip (precision = 'fp32' parity mode)

namespace {

struct WgradK {
    const bf16_t* src0; const bf16_t* src1; const int* map1;
    int C0, C1, H0p, W0p, H1p, W1p, ups0, ups1, si, ntaps;
    unsigned long long dy_bits, dx_bits, ooy_bits, oox_bits;
    const bf16_t* dout; int DHp, DWp, so, Cout;
    int N, OH, OW;
    float* dw; int splitk;
    const int* map0;
    int lg_hw, lg_ow;          // log2 of OH*OW and OW when both are powers of two, else -1
    int dcs, dco;              // dout pixel stride in channels (= Cout unless dout is a channel slice) and first channel
    int ptaps;                 // > 0: tap t reads the slice of phase t / ptaps (first channel dco + (t / ptaps) * Cout)
};

typedef short s16x4_t __attribute__((ext_vector_type(4)));

// ds_read_b64_tr_b16: every lane supplies the address of 4 contiguous bf16; within each 16-lane group the 16x4
// element block is transposed so that lane s receives element (s&3) of lanes 4j+(s>>2), j = 0..3.
__device__ __forceinline__ s16x4_t lds_tr_read(const bf16_t* p) {
    typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p);
}

template <int BJ, int BC, int WJ, int WC, bool USE_TR, bool POW2>
__global__ __launch_bounds__(WJ * WC * 64) void wgrad_regstage_kernel(const WgradK a) {
    constexpr int NT = WJ * WC * 64;
    constexpr int BP = 32;                       // pixels per K step
    constexpr int LDX = BJ + 32;                 // row stride (elements): +64 B keeps the 4 rows of a tr-read on distinct banks
    constexpr int LDY = BC + 32;
    constexpr int XCH = BJ / 8, YCH = BC / 8;    // 16-byte chunks per pixel row
    constexpr int X_LD = (BP * XCH + NT - 1) / NT, Y_LD = (BP * YCH + NT - 1) / NT;
    constexpr int TJ = BJ / WJ / 32, TC = BC / WC / 32;
    __shared__ __attribute__((aligned(16))) bf16_t Xs[2][BP][LDX];
    __shared__ __attribute__((aligned(16))) bf16_t Ys[2][BP][LDY];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wj = wid / WC, wc = wid % WC;
    const int Ctot = a.C0 + a.C1;
    const int tj_n = a.Cout / BJ, tc_n = Ctot / BC;
    // logical index: tap fastest, XCD-contiguous -- the ntaps workgroups of one (tile, K range) read the same gradient
    // tile and overlapping (shifted) input tiles, so they should run back to back on ONE XCD / L2
    int b = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int t = b % a.ntaps; b /= a.ntaps;
    const int tc = b % tc_n; b /= tc_n;
    const int tj = b % tj_n; b /= tj_n;
    const int split = b;
    const int j0 = tj * BJ, c0 = tc * BC;
    const long long M = (long long)a.N * a.OH * a.OW;
    const long long nchunks = (M + BP - 1) / BP;
    const long long per = (nchunks + a.splitk - 1) / a.splitk;
    const long long ch_beg = (long long)split * per;
    long long ch_end = ch_beg + per; if (ch_end > nchunks) ch_end = nchunks;
    if (ch_beg >= ch_end) return;

    // source selection for this channel tile (a tile never straddles the two sources)
    const bool second = c0 >= a.C0;
    const bf16_t* src = second ? a.src1 : a.src0;
    const int C = second ? a.C1 : a.C0, Hp = second ? a.H1p : a.H0p, Wp = second ? a.W1p : a.W0p;
    const int ups = (second ? a.ups1 : a.ups0) ? 1 : 0;
    const int cs = second ? c0 - a.C0 : c0;
    const int dy = (int)((a.dy_bits >> (4 * t)) & 15), dx = (int)((a.dx_bits >> (4 * t)) & 15);
    const int ooy = (int)((a.ooy_bits >> (4 * t)) & 15), oox = (int)((a.oox_bits >> (4 * t)) & 15);
    const int hw = a.OH * a.OW;

    u32x4_t rx[X_LD], ry[Y_LD];
    // pixel index -> (image, row, column): shifts when the grid dims are powers of two (always, for 64x64 models)
    auto decode = [&](int m, int& n, int& oy, int& ox) {
        if constexpr (POW2) {
            n = m >> a.lg_hw; int r = m & (hw - 1);
            oy = r >> a.lg_ow; ox = r & (a.OW - 1);
        } else {
            n = m / hw; int r = m - n * hw;
            oy = r / a.OW; ox = r - oy * a.OW;
        }
    };
    // Loads are unconditional (clamped pixel index, 32-bit element offsets: the launcher checks that the tensors have
    // fewer than 2^32 elements); rows past the end contribute nothing because their gradient row is zeroed by a select.
    const int Mi = (int)M;
    auto load_step = [&](long long chunk) {
#pragma unroll
        for (int i = 0; i < X_LD; ++i) {
            int q = tid + i * NT;
            int row = q / XCH, ch = q % XCH;
            int m = (int)chunk * BP + row;
            const bool valid = m < Mi;
            int n, oy, ox;
            decode(valid ? m : Mi - 1, n, oy, ox);
            unsigned off = (((unsigned)n * a.DHp + oy * a.so + ooy) * a.DWp + ox * a.so + oox) * a.dcs + a.dco + (a.ptaps ? (t / a.ptaps) * a.Cout : 0) + j0 + ch * 8;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (q < BP * XCH) v = *reinterpret_cast<const u32x4_t*>(a.dout + off);
            const unsigned keep = valid ? 0xffffffffu : 0u;
            v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
            rx[i] = v;
        }
#pragma unroll
        for (int i = 0; i < Y_LD; ++i) {
            int q = tid + i * NT;
            int row = q / YCH, ch = q % YCH;
            int m = (int)chunk * BP + row;
            int n, oy, ox;
            decode(m < Mi ? m : Mi - 1, n, oy, ox);
            int vy = oy * a.si + dy, vx = ox * a.si + dx;
            vy = (vy + ups) >> ups; vx = (vx + ups) >> ups;
            if (second ? a.map1 != nullptr : a.map0 != nullptr) n = (second ? a.map1 : a.map0)[n];
            unsigned off = (((unsigned)n * Hp + vy) * Wp + vx) * C + cs + ch * 8;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (q < BP * YCH) v = *reinterpret_cast<const u32x4_t*>(src + off);
            ry[i] = v;
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < X_LD; ++i) {
            int q = tid + i * NT;
            if (q < BP * XCH) *reinterpret_cast<u32x4_t*>(&Xs[buf][q / XCH][(q % XCH) * 8]) = rx[i];
        }
#pragma unroll
        for (int i = 0; i < Y_LD; ++i) {
            int q = tid + i * NT;
            if (q < BP * YCH) *reinterpret_cast<u32x4_t*>(&Ys[buf][q / YCH][(q % YCH) * 8]) = ry[i];
        }
    };

    f32x16_t acc[TJ][TC];
#pragma unroll
    for (int i = 0; i < TJ; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_step(ch_beg);
    store_step(0);
    __syncthreads();
    const int lrow = lane & 31, lk = (lane >> 5) * 8;
    // tr-read addressing: 16-lane group g reads the [4 pixels][16 channels] block; lane s supplies the address of
    // pixel (s>>2), channels (s&3)*4..+3 and receives channel s, pixels 0..3.
    const int g = lane >> 4, sl = lane & 15;
    const int tr_col = (g & 1) * 16 + (sl & 3) * 4, tr_row = (g >> 1) * 8 + (sl >> 2);
    int it = 0;
    for (long long chunk = ch_beg; chunk < ch_end; ++chunk, ++it) {
        const int buf = it & 1;
        if (chunk + 1 < ch_end) load_step(chunk + 1);
#pragma unroll
        for (int kk = 0; kk < BP / 16; ++kk) {
            bf16x8_t xf[TJ], yf[TC];
            if constexpr (USE_TR) {
#pragma unroll
                for (int i = 0; i < TJ; ++i) {
                    const bf16_t* p = &Xs[buf][kk * 16 + tr_row][wj * (TJ * 32) + i * 32 + tr_col];
                    s16x4_t lo = lds_tr_read(p), hi = lds_tr_read(p + 4 * LDX);
                    typedef short s16x8_t __attribute__((ext_vector_type(8)));
                    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    xf[i] = __builtin_bit_cast(bf16x8_t, v);
                }
#pragma unroll
                for (int j = 0; j < TC; ++j) {
                    const bf16_t* p = &Ys[buf][kk * 16 + tr_row][wc * (TC * 32) + j * 32 + tr_col];
                    s16x4_t lo = lds_tr_read(p), hi = lds_tr_read(p + 4 * LDY);
                    typedef short s16x8_t __attribute__((ext_vector_type(8)));
                    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    yf[j] = __builtin_bit_cast(bf16x8_t, v);
                }
            } else {
                typedef short s16x8_t __attribute__((ext_vector_type(8)));
#pragma unroll
                for (int i = 0; i < TJ; ++i) {
                    s16x8_t v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (short)Xs[buf][kk * 16 + lk + e][wj * (TJ * 32) + i * 32 + lrow];
                    xf[i] = __builtin_bit_cast(bf16x8_t, v);
                }
#pragma unroll
                for (int j = 0; j < TC; ++j) {
                    s16x8_t v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (short)Ys[buf][kk * 16 + lk + e][wc * (TC * 32) + j * 32 + lrow];
                    yf[j] = __builtin_bit_cast(bf16x8_t, v);
                }
            }
#pragma unroll
            for (int i = 0; i < TJ; ++i)
#pragma unroll
                for (int j = 0; j < TC; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[i], yf[j], acc[i][j], 0, 0, 0);
        }
        if (chunk + 1 < ch_end) store_step(buf ^ 1);
        __syncthreads();
    }

    const int lcol = lane & 31, lhalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TJ; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int jj = j0 + wj * (TJ * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                int cc = c0 + wc * (TC * 32) + j * 32 + lcol;
                atomicAdd(a.dw + ((size_t)t * a.Cout + jj) * Ctot + cc, acc[i][j][r]);
            }
}

// ---------------------------------------------------------------------------------------------------------------
// Main kernel: LDS-DMA (global_load_lds) staging into an NBUF-deep LDS ring, DMA running NBUF-1 K steps ahead of the
// MFMAs behind COUNTED s_waitcnt vmcnt + one raw s_barrier per K step.  (Measured on the register-staged version:
// waves parked in s_waitcnt 60-75 % of the time -- bytes in flight per CU, not bandwidth, was the limit.)
// LDS rows are unpadded (DMA writes lane-linear); the transpose-read bank spread comes from XOR-ing the 16-byte chunk
// index with f(row) on the SOURCE side and on the read side (same involution).
// Requirements (else the register-staged kernel above is used): pixel count % 32 == 0, power-of-two grid dims.
// ---------------------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Four transpose reads (k step 0 lo/hi, k step 1 lo/hi) of one 32-channel MFMA tile from one base address, issued
// from inline asm WITH their s_waitcnt: hipcc gives the ds_read_tr builtin no memory operand, so beside in-flight LDS-DMA
// it would insert s_waitcnt vmcnt(0) in front of every read and serialise the DMA ring; asm reads are invisible to
// that pass (and are counted here by hand).
template <int OFF_HI, int OFF_KK>
__device__ __forceinline__ void tr_read_tile(unsigned addr, s16x4_t& k0lo, s16x4_t& k0hi, s16x4_t& k1lo, s16x4_t& k1hi) {
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                 "ds_read_b64_tr_b16 %1, %4 offset:%5\n\t"
                 "ds_read_b64_tr_b16 %2, %4 offset:%6\n\t"
                 "ds_read_b64_tr_b16 %3, %4 offset:%7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(k0lo), "=&v"(k0hi), "=&v"(k1lo), "=&v"(k1hi)
                 : "v"(addr), "n"(OFF_HI), "n"(OFF_KK), "n"(OFF_KK + OFF_HI)
                 : "memory");
}

template <int CH> __device__ __forceinline__ int tr_swz(int row) {
    // CH = 16-byte chunks per LDS row.  A transpose read touches 4 consecutive rows x 64 B: move them to 4 distinct
    // 64-byte bank groups of the 256-byte LDS bank row.
    if constexpr (CH >= 16) return (row & 3) << 2;
    else if constexpr (CH == 8) return ((row >> 1) & 1) << 2;
    else return 0;
}

template <int BJ, int BC, int WJ, int WC, int NBUF>
__global__ __launch_bounds__(WJ * WC * 64) void wgrad_mfma_kernel(const WgradK a) {
    constexpr int NT = WJ * WC * 64;
    constexpr int BP = 32;
    constexpr int XCH = BJ / 8, YCH = BC / 8;
    constexpr int X_LD = (BP * XCH + NT - 1) / NT, Y_LD = (BP * YCH + NT - 1) / NT;
    constexpr int G = X_LD + Y_LD;                      // LDS-DMA instructions per K step and wave
    constexpr int TJ = BJ / WJ / 32, TC = BC / WC / 32;
    constexpr int XB = BP * BJ, YB = BP * BC;           // elements per buffer
    constexpr int MAXMAP = 4096;                        // skip-connection image map staged in LDS (ds_read, not vmcnt)
    __shared__ __attribute__((aligned(1024))) bf16_t lds[NBUF * (XB + YB) + MAXMAP * 2];
    int* maps = reinterpret_cast<int*>(lds + NBUF * (XB + YB));

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wj = wid / WC, wc = wid % WC;
    const int Ctot = a.C0 + a.C1;
    const int tj_n = a.Cout / BJ, tc_n = Ctot / BC;
    int b = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int t = b % a.ntaps; b /= a.ntaps;
    const int tc = b % tc_n; b /= tc_n;
    const int tj = b % tj_n; b /= tj_n;
    const int split = b;
    const int j0 = tj * BJ, c0 = tc * BC;
    const int M = a.N * a.OH * a.OW;
    const int nchunks = M / BP;
    const int per = (nchunks + a.splitk - 1) / a.splitk;
    const int ch_beg = split * per;
    int ch_end = ch_beg + per; if (ch_end > nchunks) ch_end = nchunks;
    if (ch_beg >= ch_end) return;
    const int nsteps = ch_end - ch_beg;

    const bool second = c0 >= a.C0;
    const bf16_t* src = second ? a.src1 : a.src0;
    const int C = second ? a.C1 : a.C0, Hp = second ? a.H1p : a.H0p, Wp = second ? a.W1p : a.W0p;
    const int ups = (second ? a.ups1 : a.ups0) ? 1 : 0;
    const int cs = second ? c0 - a.C0 : c0;
    const int dy = (int)((a.dy_bits >> (4 * t)) & 15), dx = (int)((a.dx_bits >> (4 * t)) & 15);
    const int ooy = (int)((a.ooy_bits >> (4 * t)) & 15), oox = (int)((a.oox_bits >> (4 * t)) & 15);
    const int hw = a.OH * a.OW;
    const int* mp = second ? a.map1 : a.map0;        // image indirection of this block's source
    const bool mapped = mp != nullptr;

    // fixed (row, source chunk) of this thread's DMA pieces; narrow tiles are re-loaded by the upper waves (same
    // bytes to the same LDS address) so that every wave issues exactly G DMAs per step -- vmcnt is per wave.
    int xrow[X_LD], xch[X_LD], yrow[Y_LD], ych[Y_LD];
#pragma unroll
    for (int i = 0; i < X_LD; ++i) { int q = (tid + i * NT) % (BP * XCH); xrow[i] = q / XCH; xch[i] = (q % XCH) ^ tr_swz<XCH>(q / XCH); }
#pragma unroll
    for (int i = 0; i < Y_LD; ++i) { int q = (tid + i * NT) % (BP * YCH); yrow[i] = q / YCH; ych[i] = (q % YCH) ^ tr_swz<YCH>(q / YCH); }

    auto stage = [&](int step, int buf) {
        const int mbase = (ch_beg + step) * BP;
        bf16_t* Xd = lds + (size_t)buf * (XB + YB);
        bf16_t* Yd = Xd + XB;
#pragma unroll
        for (int i = 0; i < X_LD; ++i) {
            const int m = mbase + xrow[i];
            const int n = m >> a.lg_hw, r = m & (hw - 1);
            const int oy = r >> a.lg_ow, ox = r & (a.OW - 1);
            unsigned off = (((unsigned)n * a.DHp + oy * a.so + ooy) * a.DWp + ox * a.so + oox) * a.dcs + a.dco + (a.ptaps ? (t / a.ptaps) * a.Cout : 0) + j0 + xch[i] * 8;
            const int qb = ((wid * 64 + i * NT) % (BP * XCH)) * 8;        // wave-uniform LDS element offset of this piece
            __builtin_amdgcn_global_load_lds((gptr_t)(a.dout + off), (lptr_t)(Xd + qb), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < Y_LD; ++i) {
            const int m = mbase + yrow[i];
            int n = m >> a.lg_hw; const int r = m & (hw - 1);
            const int oy = r >> a.lg_ow, ox = r & (a.OW - 1);
            int vy = oy * a.si + dy, vx = ox * a.si + dx;
            vy = (vy + ups) >> ups; vx = (vx + ups) >> ups;
            if (mapped) n = maps[n];
            unsigned off = (((unsigned)n * Hp + vy) * Wp + vx) * C + cs + ych[i] * 8;
            const int qb = ((wid * 64 + i * NT) % (BP * YCH)) * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + off), (lptr_t)(Yd + qb), 16, 0, 0);
        }
    };

    if (mapped) {
        for (int i = tid; i < a.N; i += NT) maps[i] = mp[i];
        __syncthreads();
    }
    f32x16_t acc[TJ][TC];
#pragma unroll
    for (int i = 0; i < TJ; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transpose-read addressing (see lds_tr_read): lane s of a 16-lane group addresses pixel (s>>2), channels (s&3)*4..
    // The swizzle term only depends on (s>>2)&3 (the other row terms are multiples of 4), so it is a per-lane constant.
    const int g = lane >> 4, sl = lane & 15;
    const int tr_c = (g & 1) * 16 + (sl & 3) * 4;       // channel within the 32-wide MFMA tile
    const int tr_r = (g >> 1) * 8 + (sl >> 2);          // pixel within the 16-pixel k step (+4 for the second read)
    unsigned xoff[TJ], yoff[TC];                        // byte offsets inside a buffer (k step 0, first read)
#pragma unroll
    for (int i = 0; i < TJ; ++i) {
        const int c = wj * (TJ * 32) + i * 32 + tr_c;
        xoff[i] = 2u * (unsigned)(tr_r * BJ + (((c >> 3) ^ tr_swz<XCH>(tr_r)) << 3) + (c & 7));
    }
#pragma unroll
    for (int j = 0; j < TC; ++j) {
        const int c = wc * (TC * 32) + j * 32 + tr_c;
        yoff[j] = 2u * (unsigned)(XB + tr_r * BC + (((c >> 3) ^ tr_swz<YCH>(tr_r)) << 3) + (c & 7));
    }
    const unsigned lds_base = (unsigned)(uintptr_t)lds;     // LDS byte address of the ring

#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (i < nsteps) stage(i, i);
    for (int s = 0; s < nsteps; ++s) {
        // tile s has landed when at most the NBUF-2 younger tiles' DMAs are still outstanding
        if (s + NBUF - 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (NBUF - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + NBUF - 1 < nsteps) stage(s + NBUF - 1, (s + NBUF - 1) % NBUF);
        const unsigned bb = lds_base + (unsigned)(s % NBUF) * (unsigned)((XB + YB) * 2);
        typedef short s16x8_t __attribute__((ext_vector_type(8)));
        bf16x8_t xf[2][TJ], yf[2][TC];
#pragma unroll
        for (int i = 0; i < TJ; ++i) {
            s16x4_t a0, a1, b0, b1;
            tr_read_tile<4 * BJ * 2, 16 * BJ * 2>(bb + xoff[i], a0, a1, b0, b1);
            s16x8_t v0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            s16x8_t v1 = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            xf[0][i] = __builtin_bit_cast(bf16x8_t, v0); xf[1][i] = __builtin_bit_cast(bf16x8_t, v1);
        }
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            s16x4_t a0, a1, b0, b1;
            tr_read_tile<4 * BC * 2, 16 * BC * 2>(bb + yoff[j], a0, a1, b0, b1);
            s16x8_t v0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            s16x8_t v1 = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            yf[0][j] = __builtin_bit_cast(bf16x8_t, v0); yf[1][j] = __builtin_bit_cast(bf16x8_t, v1);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < TJ; ++i)
#pragma unroll
                for (int j = 0; j < TC; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[kk][i], yf[kk][j], acc[i][j], 0, 0, 0);
    }

    const int lcol = lane & 31, lhalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TJ; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int jj = j0 + wj * (TJ * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                int cc = c0 + wc * (TC * 32) + j * 32 + lcol;
                atomicAdd(a.dw + ((size_t)t * a.Cout + jj) * Ctot + cc, acc[i][j][r]);
            }
}

// ---------------------------------------------------------------------------------------------------------------
// Halo-tiled weight gradient of the 3x3 stride-1 convolutions: ONE workgroup accumulates all 9 taps of a 64 x 64
// (Cout x Cin) tile.  Per K step it stages 32 output-gradient pixels (a 2x16 or 4x8 spatial tile) and the input patch
// INCLUDING its halo once ([<=72 px][64 ch]); the 9 taps are LDS address offsets of the transpose reads, so the input
// operand is fetched once instead of 9 times and the output gradient once instead of 9 times (L2->LDS bytes per FLOP
// down ~3x versus the per-tap kernel above, whose waves sat in s_waitcnt 60-75 % of the time).  Each wave owns one
// 32x32 tile for the 9 taps (144 accumulator registers).  Same 4-deep LDS-DMA ring / counted vmcnt / raw barrier
// pipeline and the same ds_read_b64_tr_b16 fragment reads; split-K over spatial tiles, fp32 atomics.
// ---------------------------------------------------------------------------------------------------------------
struct WgradHaloK {
    WgradK a;
    int lgTW, RH, PW, Ppix, lg_nxb, lg_nyb, ntiles, oo;   // oo: border offset of dout (ooy = oox)
};

template <bool UPS, int BJ, int NTAPS = 9, int NX = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wgrad_halo_kernel(const WgradHaloK p) {
    // BJ = 64: wave = (cout tile, cin tile), all 9 taps.  BJ = 32 (image-side layer, Cout padded to 32): wave = (cin tile,
    // tap group 0-4 / 5-8); the second group computes one duplicate tap that is not written back.
    // NTAPS = 4 (BJ = 64): one phase of a sub-pixel upsample convolution -- dout is a channel slice of the space-to-depth
    // gradient, the four taps are that phase's offsets inside the same 3x3 window.
    // NTAPS = 8, NX = 2 (round 4): TWO phases of such a block per workgroup -- both phases' 64-channel slices of the space-to-depth
    // gradient are staged beside ONE input patch (the 16 (phase, folded tap) products all read the same 3x3 window of it), taps 0-3 use
    // slice 0 and taps 4-7 slice 1: 16 MFMAs per wave and 20 KB stage (the 9-tap kernel: 18 per 16 KB; one phase: 8 per 16 KB, which
    // was LDS-DMA bound).  blockIdx carries the phase pair; 3-deep ring so that two workgroups still share a CU.
    constexpr int NT = 256, NBUF = NX == 2 ? 3 : 4, BC = 64;
    constexpr int NTW = BJ == 64 ? NTAPS : 5;            // taps per wave
    constexpr int NTOT = NX == 2 ? 16 : NTAPS;           // taps of the launch (dy / dx entries, dw slabs)
    static_assert(NX == 1 || (BJ == 64 && NTAPS == 8 && !UPS), "two-slice variant: 8 taps, 64 x 64 tiles");
    constexpr int XROW = BJ * 2;                         // bytes per dout pixel row
    constexpr int XTILE = 32 * XROW;                     // [32 px][BJ ch]
    constexpr int XBYTES = NX * XTILE;
    constexpr int YPIECES = 3 * NT;                      // 96 px * 8 chunks
    constexpr int YBYTES = YPIECES * 16;                 // 12 KiB
    constexpr int STAGE = XBYTES + YBYTES;
    constexpr int G = 4;                                 // LDS-DMA instructions per stage and wave
    constexpr int MAXMAP = 2048;                         // image map of a mapped source, staged in LDS (ds_read: lgkmcnt, not vmcnt)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NBUF * STAGE + MAXMAP * 4];
    int* maps = reinterpret_cast<int*>(lds + NBUF * STAGE);
    const WgradK& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int jt = BJ == 64 ? (wid >> 1) : 0, ct = wid & 1;
    const int tc_n = a.C0 / BC, tj_n = a.Cout / BJ;
    int b = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int tc = b % tc_n; b /= tc_n;
    const int tj = b % tj_n; b /= tj_n;
    int pp = 0;                                          // phase pair (NX = 2)
    if constexpr (NX == 2) { pp = b & 1; b >>= 1; }
    const int tap0 = NX == 2 ? pp * 8 : (BJ == 64 ? 0 : (wid >> 1) * 5);
    const int split = b;
    const int j0 = tj * BJ, c0 = tc * BC;
    const int per = (p.ntiles + a.splitk - 1) / a.splitk;
    const int t_beg = split * per;
    int t_end = t_beg + per; if (t_end > p.ntiles) t_end = p.ntiles;
    if (t_beg >= t_end) return;
    const int nsteps = t_end - t_beg;
    const int TW = 1 << p.lgTW, PW = p.PW;
    const int ups = UPS ? 1 : 0;

    // ---- DMA pieces of this thread: lane-constant parts of the source offsets
    unsigned xlane, ylane[3];
    {
        // BJ = 32: 128 pieces of 64-byte rows (no swizzle needed: 4 rows = one 256-byte bank row); the upper two waves
        // re-load the same pieces so that every wave issues the same number of DMAs per stage
        const int xq = BJ == 64 ? tid : (tid & 127);
        const int pk = BJ == 64 ? (xq >> 3) : (xq >> 2), pos = BJ == 64 ? (xq & 7) : (xq & 3);
        const int ty = pk >> p.lgTW, tx = pk & (TW - 1);
        xlane = (unsigned)((ty * a.DWp + tx) * a.dcs + ((BJ == 64 ? (pos ^ (((pk >> 1) & 1) << 2)) : pos) * 8));
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = tid + i * NT;
            int pix = q >> 3;
            const int posy = q & 7;
            const int sw = ((pix >> 1) & 1) << 2;
            if (pix >= p.Ppix) pix = 0;                  // filler pieces: re-load pixel 0 (never read)
            const int py = pix / PW, px = pix - py * PW;
            ylane[i] = (unsigned)((py * a.W0p + px) * a.C0 + ((posy ^ sw) * 8));
        }
    }
    // LDS-DMA is the scarce path (~1 KiB per ~100 cycles per CU): a wave only issues the patch pieces that exist
    // (the 2x16 tile's patch is 576 pieces = 2.25 rounds of 256, the 4x8 tile's 480 = 1.9), so the DMA count per
    // stage -- and with it the counted vmcnt -- is per wave: 1 (gradient tile) + ny
    const int ny = __builtin_amdgcn_readfirstlane((p.Ppix * 8 - wid * 64 + NT - 1) / NT);      // rounds with a real piece for this wave
    auto stage = [&](int step, int buf) {
        const int tau = t_beg + step;
        const int xb = tau & ((1 << p.lg_nxb) - 1);
        const int yb = (tau >> p.lg_nxb) & ((1 << p.lg_nyb) - 1);
        const int n = tau >> (p.lg_nxb + p.lg_nyb);
        const int y0 = yb * p.RH, x0 = xb << p.lgTW;
        const int ns = a.map0 ? maps[n] : n;             // LDS copy: a global load here would sit in the DMA's vmcnt queue
        const unsigned xbase = (((unsigned)n * a.DHp + y0 + p.oo) * a.DWp + x0 + p.oo) * a.dcs + a.dco + (NX == 2 ? 2 * pp * a.Cout : 0) + j0;
        const unsigned ybase = (((unsigned)ns * a.H0p + (y0 >> ups)) * a.W0p + (x0 >> ups)) * a.C0 + c0;
        unsigned char* sb = lds + (size_t)buf * STAGE;
        __builtin_amdgcn_global_load_lds((gptr_t)(a.dout + xbase + xlane), (lptr_t)(sb + (BJ == 64 ? wid : (wid & 1)) * 1024), 16, 0, 0);
        if constexpr (NX == 2)
            __builtin_amdgcn_global_load_lds((gptr_t)(a.dout + xbase + a.Cout + xlane), (lptr_t)(sb + XTILE + wid * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < ny)
                __builtin_amdgcn_global_load_lds((gptr_t)(a.src0 + ybase + ylane[i]), (lptr_t)(sb + XBYTES + (i * NT + wid * 64) * 16), 16, 0, 0);
    };

    if (a.map0) {
        for (int i = tid; i < a.N; i += NT) maps[i] = a.map0[i];
        __syncthreads();
    }
    f32x16_t acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- transpose-read addressing (see lds_tr_read): lane s of a 16-lane group addresses pixel (s>>2), channels (s&3)*4..
    const int g = lane >> 4, sl = lane & 15;
    const int tr_c = (g & 1) * 16 + (sl & 3) * 4;
    const int tr_r = (g >> 1) * 8 + (sl >> 2);           // pixel within a 16-pixel K slice (+4: second read, +16: second slice)
    // dout tile (linear [32 px][64 ch]): rows tr_r, +4, +16, +20 share the swizzle term
    const int cx = jt * 32 + tr_c;
    const unsigned xoff = BJ == 64 ? (unsigned)(tr_r * 128 + ((((cx >> 3) ^ (((tr_r >> 1) & 1) << 2))) << 4) + (cx & 7) * 2)
                                   : (unsigned)(tr_r * 64 + (cx >> 3) * 16 + (cx & 7) * 2);
    // input patch: address(P, c) = P*128 + ((c>>3) ^ 4*((P>>1)&1))*16 + (c&7)*2 = ybase_c + P*128 + bit8(P*128) * ydelta
    const int cy = ct * 32 + tr_c;
    const unsigned ybase_c = (unsigned)(XBYTES + (cy >> 3) * 16 + (cy & 7) * 2);
    const int ydelta = ((cy >> 3) & 4) ? -64 : 64;
    const int ty0 = tr_r >> p.lgTW, tx0 = tr_r & (TW - 1);
    const int ty1 = ty0 + (16 >> p.lgTW);                // pixel row of the second 16-pixel slice
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    typedef short s16x8_t __attribute__((ext_vector_type(8)));

    // per-tap transpose-read offsets inside a stage (K-step invariant): the inner loop only adds the ring base
    constexpr int NAD = UPS ? 4 : 2;
    unsigned toff[NTW][NAD];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int tg = (tap0 + t) > NTOT - 1 ? NTOT - 1 : tap0 + t;
        const int dy = (int)((a.dy_bits >> (4 * tg)) & 15), dx = (int)((a.dx_bits >> (4 * tg)) & 15);
        if constexpr (!UPS) {
            const unsigned P0 = (unsigned)((ty0 + dy) * PW + tx0 + dx) << 7, P1 = (unsigned)((ty1 + dy) * PW + tx0 + dx) << 7;
            toff[t][0] = ybase_c + P0 + ((P0 >> 8) & 1) * ydelta;
            toff[t][1] = ybase_c + P1 + ((P1 >> 8) & 1) * ydelta;
        } else {
            const int sy0 = (ty0 + dy + 1) >> 1, sy1 = (ty1 + dy + 1) >> 1, sx0 = (tx0 + dx + 1) >> 1;
            const unsigned P00 = (unsigned)(sy0 * PW + sx0) << 7, P10 = (unsigned)(sy1 * PW + sx0) << 7;
            const unsigned P01 = P00 + 2 * 128, P11 = P10 + 2 * 128;      // tx + 4 -> source x + 2
            toff[t][0] = ybase_c + P00 + ((P00 >> 8) & 1) * ydelta;
            toff[t][1] = ybase_c + P01 + ((P01 >> 8) & 1) * ydelta;
            toff[t][2] = ybase_c + P10 + ((P10 >> 8) & 1) * ydelta;
            toff[t][3] = ybase_c + P11 + ((P11 >> 8) & 1) * ydelta;
        }
    }
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (i < nsteps) stage(i, i);
    for (int s = 0; s < nsteps; ++s) {
        if (s + NBUF - 2 < nsteps) {
            // tile s has landed when only the NBUF-2 younger stages ((1 + ny) DMAs each) are outstanding
            if (ny == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NX + 3) * (NBUF - 2)) : "memory");
            else if (ny == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NX + 2) * (NBUF - 2)) : "memory");
            else if (ny == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NX + 1) * (NBUF - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NX * (NBUF - 2)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + NBUF - 1 < nsteps) stage(s + NBUF - 1, (s + NBUF - 1) % NBUF);
        const unsigned bb = lds_base + (unsigned)(s % NBUF) * (unsigned)STAGE;
        bf16x8_t xf[NX][2];
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            s16x4_t a0, a1, b0, b1;
            tr_read_tile<4 * XROW, 16 * XROW>(bb + xoff + q * XTILE, a0, a1, b0, b1);
            s16x8_t v0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            s16x8_t v1 = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            xf[q][0] = __builtin_bit_cast(bf16x8_t, v0); xf[q][1] = __builtin_bit_cast(bf16x8_t, v1);
        }
        // software pipeline over the taps: the four transpose reads of tap t+1 are issued before the MFMAs of tap t (LDS
        // returns in order, so lgkmcnt(4) = "tap t has landed"); the waits carry the fragment registers as operands so
        // that the compiler keeps issue -> wait -> MFMA in this order.
        s16x4_t y0[2], y1[2], y2[2], y3[2];
        auto issue = [&](int t, s16x4_t& r0, s16x4_t& r1, s16x4_t& r2, s16x4_t& r3) {
            if constexpr (!UPS) {
                // second / fourth read = +4 pixels (same (P>>1)&1 swizzle term): immediate offsets
                const unsigned a0 = bb + toff[t][0], a1 = bb + toff[t][1];
                asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                             "ds_read_b64_tr_b16 %1, %4 offset:512\n\t"
                             "ds_read_b64_tr_b16 %2, %5\n\t"
                             "ds_read_b64_tr_b16 %3, %5 offset:512"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                             : "v"(a0), "v"(a1)
                             : "memory");
            } else {
                unsigned ad[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) ad[h] = bb + toff[t][h];
                asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                             "ds_read_b64_tr_b16 %1, %5\n\t"
                             "ds_read_b64_tr_b16 %2, %6\n\t"
                             "ds_read_b64_tr_b16 %3, %7"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                             : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3])
                             : "memory");
            }
        };
        issue(0, y0[0], y1[0], y2[0], y3[0]);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int c = t & 1, n = c ^ 1;
            if (t < NTW - 1) {
                issue(t + 1, y0[n], y1[n], y2[n], y3[n]);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(y0[c]), "+v"(y1[c]), "+v"(y2[c]), "+v"(y3[c])::"memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y0[c]), "+v"(y1[c]), "+v"(y2[c]), "+v"(y3[c])::"memory");
            }
            s16x8_t v0 = {y0[c][0], y0[c][1], y0[c][2], y0[c][3], y1[c][0], y1[c][1], y1[c][2], y1[c][3]};
            s16x8_t v1 = {y2[c][0], y2[c][1], y2[c][2], y2[c][3], y3[c][0], y3[c][1], y3[c][2], y3[c][3]};
            constexpr int XS = NX == 2 ? 4 : 1 << 30;     // taps per dout slice
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[t / XS][0], __builtin_bit_cast(bf16x8_t, v0), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[t / XS][1], __builtin_bit_cast(bf16x8_t, v1), acc[t], 0, 0, 0);
        }
    }

    const int lcol = lane & 31, lhalf = lane >> 5;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        if (tap0 + t > NTOT - 1) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = j0 + jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
            const int cc = c0 + ct * 32 + lcol;
            atomicAdd(a.dw + ((size_t)(tap0 + t) * a.Cout + jj) * a.C0 + cc, acc[t][r]);
        }
    }
}

int g_wgrad_halo = -1;       // -1: SRVP_WGRAD_HALO env (default 1)

// Launch the halo kernel if the descriptor is a 3x3 stride-1 single-source weight gradient it covers.
static int try_launch_halo(const srvp_wgrad_desc* d, const WgradK& k, hipStream_t st, bool& done) {
    done = false;
    if (g_wgrad_halo < 0) { const char* e = getenv("SRVP_WGRAD_HALO"); g_wgrad_halo = e ? atoi(e) : 1; }
    if (!g_wgrad_halo) return SRVP_OK;
    // one phase of a sub-pixel block (space-to-depth dout): with 4 taps per staged tile the halo kernel is LDS-DMA bound (16 KB staged
    // per 8 MFMAs per wave); it beats the per-tap kernel only on the 64-channel 64x64 stage (0.65 vs 0.90 ms), wider layers
    // measured 0.55-0.60 vs 0.50-0.52 ms and stay on the per-tap kernel (which reads the channel slice through dcs / dco as well)
    static int four_max = -1;
    if (four_max < 0) { const char* e = getenv("SRVP_WGRAD_HALO4_MAXC"); four_max = e ? atoi(e) : 64; }
    const bool four = d->ntaps == 4 && d->Cout % 64 == 0 && d->Cout <= four_max && !d->ups0;
    // all 16 (phase, folded tap) gradients of a sub-pixel block in one launch, two phases per workgroup (SRVP_WGRAD_HALO2, default on)
    static int halo2 = -1;
    if (halo2 < 0) { const char* e = getenv("SRVP_WGRAD_HALO2"); halo2 = e ? atoi(e) : 1; }
    const bool two = halo2 && d->ntaps == 16 && d->dout_phase_taps == 4 && d->Cout % 64 == 0 && !d->ups0 && d->dout_cstride == 4 * d->Cout;
    if ((d->ntaps != 9 && !four && !two) || d->si != 1 || d->so != 1 || d->C1 != 0 || d->C0 % 64 || (d->Cout % 64 && d->Cout != 32)) return SRVP_OK;
    if ((!four && !two && (d->dout_cstride || d->dout_coff)) || (d->dout_phase_taps && !two)) return SRVP_OK;
    for (int t = 0; t < d->ntaps; ++t)
        if (d->dy[t] < 0 || d->dy[t] > 2 || d->dx[t] < 0 || d->dx[t] > 2 || d->ooy[t] != d->ooy[0] || d->oox[t] != d->ooy[0]) return SRVP_OK;
    const int OH = d->OH, OW = d->OW, ups = d->ups0 ? 1 : 0;
    if (OW < 8 || (OW & (OW - 1)) || (OH & (OH - 1))) return SRVP_OK;
    if (d->H0p != (OH >> ups) + 2 || d->W0p != (OW >> ups) + 2) return SRVP_OK;
    if (d->DHp != OH + 2 * d->ooy[0] || d->DWp != OW + 2 * d->ooy[0]) return SRVP_OK;
    WgradHaloK h;
    h.a = k;
    const int TW = OW >= 16 ? 16 : 8;
    h.lgTW = TW == 16 ? 4 : 3;
    h.RH = 32 / TW;
    if (OH % h.RH) return SRVP_OK;
    h.PW = (TW >> ups) + 2;
    h.Ppix = ((h.RH >> ups) + 2) * h.PW;
    if (h.Ppix > 96 || (d->map0 && d->N > 2048)) return SRVP_OK;
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    h.lg_nxb = lg(OW / TW); h.lg_nyb = lg(OH / h.RH);
    h.ntiles = d->N * (OH / h.RH) * (OW / TW);
    h.oo = d->ooy[0];
    const int bj = d->Cout == 32 ? 32 : 64;
    const int pairs = (d->Cout / bj) * (d->C0 / 64) * (two ? 2 : 1);      // (two: x phase pairs)
    // workgroups per launch the split-K aims at: every split pays 9 x 64 x 64 fp32 atomics, amortised over its K steps -- 512 is
    // the measured best at 2304 frames (43.35 vs 44.1 / 44.4 ms per step for 384 / 256), 320 at 288 frames (10.70 vs 10.98 ms) in round 3;
    // round 5 at 288 frames, 192 / 256 / 320 / 448 / 640: 7.18 / 7.11 / 7.15 / 7.36 / 7.46 ms per step (two same-box sweeps): 256
    static int target_env = -2;
    if (target_env == -2) { const char* e = getenv("SRVP_WGRAD_HALO_WGS"); target_env = e ? atoi(e) : -1; }
    const int target = target_env > 0 ? target_env : (d->N < 1024 ? 256 : 512);
    int splitk = (target + pairs - 1) / pairs;
    if (splitk > h.ntiles / 8) splitk = h.ntiles / 8 > 0 ? h.ntiles / 8 : 1;
    h.a.splitk = splitk;
    const long long blocks = (long long)pairs * splitk;
    if (two) {
        hipLaunchKernelGGL((wgrad_halo_kernel<false, 64, 8, 2>), dim3((unsigned)blocks), dim3(256), 0, st, h);
        SRVP_CHECK_LAUNCH("srvp_wgrad_mfma(halo, 2 x 8 taps)");
        done = true;
        return SRVP_OK;
    }
    if (four) {
        hipLaunchKernelGGL((wgrad_halo_kernel<false, 64, 4>), dim3((unsigned)blocks), dim3(256), 0, st, h);
        SRVP_CHECK_LAUNCH("srvp_wgrad_mfma(halo, 4 taps)");
        done = true;
        return SRVP_OK;
    }
    if (bj == 32) {
        if (ups) hipLaunchKernelGGL((wgrad_halo_kernel<true, 32>), dim3((unsigned)blocks), dim3(256), 0, st, h);
        else hipLaunchKernelGGL((wgrad_halo_kernel<false, 32>), dim3((unsigned)blocks), dim3(256), 0, st, h);
    } else if (ups) hipLaunchKernelGGL((wgrad_halo_kernel<true, 64>), dim3((unsigned)blocks), dim3(256), 0, st, h);
    else hipLaunchKernelGGL((wgrad_halo_kernel<false, 64>), dim3((unsigned)blocks), dim3(256), 0, st, h);
    SRVP_CHECK_LAUNCH("srvp_wgrad_mfma(halo)");
    done = true;
    return SRVP_OK;
}

template <int BJ, int BC, int WJ, int WC>
int launch(const WgradK& k, hipStream_t st, bool use_tr, bool dma_off) {
    long long blocks = (long long)(k.Cout / BJ) * ((k.C0 + k.C1) / BC) * k.ntaps * k.splitk;
    SRVP_REQUIRE(blocks > 0 && blocks < (1ll << 31), "srvp_wgrad_mfma: bad grid");
    const bool p2 = k.lg_ow >= 0;
    const dim3 g((unsigned)blocks), b(WJ * WC * 64);
    const long long Mtot = (long long)k.N * k.OH * k.OW;
    if (use_tr && p2 && Mtot % 32 == 0 && !dma_off && ((k.map1 == nullptr && k.map0 == nullptr) || k.N <= 4096)) {
        hipLaunchKernelGGL((wgrad_mfma_kernel<BJ, BC, WJ, WC, 4>), g, b, 0, st, k);
        SRVP_CHECK_LAUNCH("srvp_wgrad_mfma");
        return SRVP_OK;
    }
    if (use_tr && p2) hipLaunchKernelGGL((wgrad_regstage_kernel<BJ, BC, WJ, WC, true, true>), g, b, 0, st, k);
    else if (use_tr) hipLaunchKernelGGL((wgrad_regstage_kernel<BJ, BC, WJ, WC, true, false>), g, b, 0, st, k);
    else hipLaunchKernelGGL((wgrad_regstage_kernel<BJ, BC, WJ, WC, false, false>), g, b, 0, st, k);
    SRVP_CHECK_LAUNCH("srvp_wgrad_mfma");
    return SRVP_OK;
}

int g_use_tr = -1;

}  // namespace

extern "C" int srvp_wgrad_set_tr(int on) { g_use_tr = on; return SRVP_OK; }
extern "C" int srvp_wgrad_set_halo(int on) { g_wgrad_halo = on; return SRVP_OK; }

extern "C" int srvp_wgrad_mfma(const srvp_wgrad_desc* d, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SRVP_REQUIRE(d && d->src0 && d->dout && d->dw, "srvp_wgrad_mfma: null pointer");
    SRVP_REQUIRE(d->C0 % 32 == 0 && d->C1 % 32 == 0 && d->Cout % 32 == 0 && d->C0 > 0,
                 "srvp_wgrad_mfma: channel counts must be padded to 32");
    SRVP_REQUIRE(d->ntaps >= 1 && d->ntaps <= SRVP_MAX_TAPS && d->splitk >= 1, "srvp_wgrad_mfma: ntaps/splitk");
    if (d->elem_f32) return srvp_wgrad_f32_launch(d, st);
    if (g_use_tr < 0) {
        const char* e = getenv("SRVP_WGRAD_TR");
        g_use_tr = e ? atoi(e) : 1;
    }
    WgradK k;
    k.src0 = (const bf16_t*)d->src0; k.src1 = (const bf16_t*)d->src1; k.map1 = d->map1; k.map0 = d->map0;
    k.C0 = d->C0; k.C1 = d->C1; k.H0p = d->H0p; k.W0p = d->W0p; k.H1p = d->H1p; k.W1p = d->W1p;
    k.ups0 = d->ups0; k.ups1 = d->ups1; k.si = d->si; k.ntaps = d->ntaps;
    k.dy_bits = k.dx_bits = k.ooy_bits = k.oox_bits = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        SRVP_REQUIRE(d->dy[t] >= 0 && d->dy[t] < 16 && d->dx[t] >= 0 && d->dx[t] < 16 && d->ooy[t] >= 0 && d->ooy[t] < 16 &&
                         d->oox[t] >= 0 && d->oox[t] < 16, "srvp_wgrad_mfma: tap offset out of [0,15]");
        k.dy_bits |= (unsigned long long)d->dy[t] << (4 * t); k.dx_bits |= (unsigned long long)d->dx[t] << (4 * t);
        k.ooy_bits |= (unsigned long long)d->ooy[t] << (4 * t); k.oox_bits |= (unsigned long long)d->oox[t] << (4 * t);
    }
    k.dout = (const bf16_t*)d->dout; k.DHp = d->DHp; k.DWp = d->DWp; k.so = d->so; k.Cout = d->Cout;
    k.dcs = d->dout_cstride ? d->dout_cstride : d->Cout; k.dco = d->dout_coff; k.ptaps = d->dout_phase_taps;
    k.N = d->N; k.OH = d->OH; k.OW = d->OW; k.dw = d->dw; k.splitk = d->splitk;
    SRVP_REQUIRE((long long)d->N * d->OH * d->OW < (1ll << 31), "srvp_wgrad_mfma: too many pixels");
    SRVP_REQUIRE((long long)d->N * d->DHp * d->DWp * k.dcs < (1ll << 32) && (long long)d->N * d->H0p * d->W0p * d->C0 < (1ll << 32),
                 "srvp_wgrad_mfma: operand tensors must have fewer than 2^32 elements");
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    k.lg_ow = lg(d->OW); k.lg_hw = lg(d->OH * d->OW);
    if (k.lg_hw < 0) k.lg_ow = -1;
    const bool tr = (g_use_tr & 1) != 0;
    // (a channel-sliced 4-tap phase launch exists on the halo kernel only: it is taken whatever the A/B switches say)
    if (tr && !(g_use_tr & 4)) {
        bool done = false;
        if (int rc = try_launch_halo(d, k, st, done)) return rc;
        if (done) return SRVP_OK;
    }
    const bool dma_off = (g_use_tr & 4) != 0;     // srvp_wgrad_set_tr(5): transpose reads, register-staged kernel (A/B switch)
    // channel tile of the input operand must not straddle the two sources
    auto divides = [&](int bc) { return d->C0 % bc == 0 && (d->C1 == 0 || d->C1 % bc == 0); };
    const int bj = d->Cout % 128 == 0 ? 128 : (d->Cout % 64 == 0 ? 64 : 32);
    const int bc = divides(128) ? 128 : (divides(64) ? 64 : 32);
    if (bj == 128 && bc == 128) return launch<128, 128, 2, 2>(k, st, tr, dma_off);
    if (bj == 128 && bc == 64) return launch<128, 64, 2, 2>(k, st, tr, dma_off);
    if (bj == 128 && bc == 32) return launch<128, 32, 4, 1>(k, st, tr, dma_off);
    if (bj == 64 && bc == 128) return launch<64, 128, 2, 2>(k, st, tr, dma_off);
    if (bj == 64 && bc == 64) return launch<64, 64, 2, 2>(k, st, tr, dma_off);
    if (bj == 64 && bc == 32) return launch<64, 32, 2, 1>(k, st, tr, dma_off);
    if (bj == 32 && bc == 128) return launch<32, 128, 1, 4>(k, st, tr, dma_off);
    if (bj == 32 && bc == 64) return launch<32, 64, 1, 2>(k, st, tr, dma_off);
    return launch<32, 32, 1, 1>(k, st, tr, dma_off);
}
