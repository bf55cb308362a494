// Image-side OUTPUT layer of the VGG decoder as a streaming kernel: ConvTranspose2d(64 -> nc, 3x3, stride 1, pad 1) + sigmoid on 64x64
// frames (reference module/conv.py:353 + 273-274), bf16 activations in, fp32 (N, nc, 64, 64) frames out.
//
// Why not the MFMA tile kernel (where this layer ran through round 3): nc <= 3 real output channels were padded to a 32-column tile
// (10x the useful MACs) and, worse, the layer is HBM-bound -- 1.2 GB of activations in, 0.1 GB of frames out, 33 GFLOP at 2304
// frames -- while a 256-pixel tile kernel pays its fixed per-tile chain (patch DMA incl. 27 % halo re-reads, barriers, LDS-staged
// epilogue) 36864 times: 0.54 ms = 2.3 TB/s, against 0.2 ms at HBM speed.
//
// Dataflow here: ONE workgroup walks a whole image top to bottom with a ROLLING window of input rows in LDS -- four groups of four
// 66-pixel rows (135 KB), filled by LDS-DMA (global_load_lds, 16 bytes per lane, no VGPR staging) two groups ahead of the compute --
// so every activation byte crosses HBM -> LDS exactly once, there is no halo re-read, no per-tile prologue / epilogue, and the DMA of
// rows 4b+12 .. 4b+15 runs under the arithmetic of output rows 4b .. 4b+7.
// Work split inside the workgroup: the WEIGHTS are the stationary operand.  The arithmetic is v_mfma_f32_16x16x32_bf16 with the PIXELS
// as the M dimension (16 consecutive columns per tile) and the nc <= 3 output channels padded to N = 16 -- the narrowest tile the matrix
// cores have; all 9 taps x 2 channel halves = 18 B fragments (72 registers) are loaded once per workgroup and stay in registers.  Wave w
// owns output columns [16 w, 16 w + 16) of all four rows of the band: an A fragment (16 pixels x 32 channels of input row R at column
// offset dx: ONE ds_read_b128 per lane, chunk index XOR-swizzled by the pixel) is read from LDS once and feeds the up to three output
// rows R - dy it contributes to: 36 fragment reads and 72 MFMAs per band and wave (~1200 cycles on each of the four SIMDs, against the
// ~2900 cycles the band's 34 KB take to arrive from HBM).  (A first version on v_dot2c_f32_bf16 was VALU-bound: that instruction issues
// at a quarter of the fp32 FMA rate, 0.46 ms at 2304 frames.)  Epilogue: the three lanes of each 16-lane group that hold real channels
// apply the sigmoid to their four consecutive pixels and store them as one 16-byte piece.
#include "common.h"
#include "../../include/srvp_hip.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

constexpr int OW_ = 64, PW_ = 66, ROWB = PW_ * 128;          // padded row: 66 pixels x 64 channels x 2 bytes
constexpr int GROUP_ROWS = 4, NGROUP_RING = 4, GROUP_B = GROUP_ROWS * ROWB;
constexpr int SLOTS = GROUP_ROWS * PW_ * 8;                  // 16-byte pieces per group = 2112 = 8.25 x 256

template <int NCO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_out_stream_kernel(
    const bf16_t* __restrict__ act, const uint32_t* __restrict__ wt, float* __restrict__ out, int N, int do_sigmoid, int HS) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[NGROUP_RING * GROUP_B];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bf16_t* img = act;                                  // (set per image below)
    // ---- DMA pieces of this thread inside a group (a group = four padded rows; group g lives in ring slot g % 4): slot q = i * 256 + tid -> (row r, pixel px, LDS chunk position s) holds the
    // source chunk s ^ (px & 7) of that pixel (swizzle applied on the source side: the LDS destination of a DMA is lane-linear)
    unsigned soff[9];
    bool svalid[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int q = i * 256 + tid;
        svalid[i] = q < SLOTS;
        const int qq = svalid[i] ? q : 0;
        const int r = qq / (PW_ * 8), rem = qq - r * (PW_ * 8), px = rem >> 3, s = rem & 7;
        soff[i] = (unsigned)(((r * PW_ + px) * 64) + ((s ^ (px & 7)) * 8));          // element offset inside the group's four rows
    }
    auto stage = [&](int g) {          // rows 4g .. 4g+3 of the padded image into ring slot g % 4 (rows past 65 do not exist: the group is cut)
        const int row0 = g * GROUP_ROWS;
        unsigned char* dst = ring + (g % NGROUP_RING) * GROUP_B;
        const bf16_t* src = img + (size_t)row0 * PW_ * 64;
        const int valid_rows = PW_ - row0 < GROUP_ROWS ? PW_ - row0 : GROUP_ROWS;      // 2 for the last group (rows 64, 65)
        const unsigned lim = (unsigned)valid_rows * PW_ * 64;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (i == 8 && wid != 0) continue;                                          // the ninth round is a quarter round (wave 0 only)
            // pieces of rows that do not exist re-read the group's first pixel (those LDS rows are never used)
            const unsigned o = (svalid[i] && soff[i] < lim) ? soff[i] : 0u;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + o), (lptr_t)(dst + ((size_t)i * 256 + wid * 64) * 16), 16, 0, 0);
        }
    };
    constexpr int NG = (PW_ + GROUP_ROWS - 1) / GROUP_ROWS;   // 17 groups of input rows
    const size_t plane = (size_t)OW_ * OW_;
    // ---- B fragments: tap t, channel half ks: lane l holds weights of output channel l % 16 (zero rows beyond nc: the packed tensor pads
    // Cout to 32 with zeros), input channels ks * 32 + (l / 16) * 8 .. + 7 -- pinned in vector registers for every item of this workgroup
    bf16x8_t wf[9][2];
    {
        const bf16_t* wl = reinterpret_cast<const bf16_t*>(wt) + (size_t)(lane & 15) * 64 + (lane >> 4) * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf[t][ks] = *reinterpret_cast<const bf16x8_t*>(wl + (size_t)t * 32 * 64 + ks * 32);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(wf[t][ks]));
    }
    const unsigned ring_base = (unsigned)(uintptr_t)ring;
  // work item = (image, 1 / HS of its rows): whole images at large N, halves / quarters when there are fewer images than ~2 per CU
  const int nb = (OW_ / GROUP_ROWS) / HS;                      // bands per item
  for (int item = blockIdx.x; item < N * HS; item += gridDim.x) {           // persistent: one workgroup per CU walks its share of the items
    const int n = item / HS, b0 = (item - n * HS) * nb, b1 = b0 + nb;        // bands [b0, b1) need the row groups b0 .. b1
    img = act + (size_t)n * PW_ * PW_ * 64;
    float* obase = out + (size_t)n * NCO * plane;
    __syncthreads();                                          // (previous item: every wave is past its last ring access)
    stage(b0);
    stage(b0 + 1);
    if (b0 + 2 <= b1) stage(b0 + 2);
    for (int b = b0; b < b1; ++b) {
        // every wave is done with band b - 1 (whose first group's ring slot the next DMA overwrites)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the DMA runs TWO groups ahead (issue -> landed is ~2 us under load, a band's arithmetic ~1 us): groups <= b + 1 have landed
        // when only the younger groups' pieces are outstanding (9 per group for wave 0, 8 for the others)
        if (b + 3 <= b1) stage(b + 3);
        const int younger = (b + 3 <= b1 ? b + 3 : b1) - (b + 1);
        // (+ 4 from the second band on: the previous band's four frame stores were issued after group b + 1 -- vmcnt counts stores too
        // and retires in order, so a count without them makes every band wait for the store acknowledgements)
        if (b > b0) {
            if (younger >= 2) { if (wid == 0) __builtin_amdgcn_s_waitcnt(0x4F70 | 6); else __builtin_amdgcn_s_waitcnt(0x4F70 | 4); }   // vmcnt(22) / (20)
            else if (younger == 1) { if (wid == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 13); else __builtin_amdgcn_s_waitcnt(0x0F70 | 12); }
            else __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
        } else if (younger >= 2) { if (wid == 0) __builtin_amdgcn_s_waitcnt(0x4F70 | 2); else __builtin_amdgcn_s_waitcnt(0x4F70); }     // vmcnt(18) / (16)
        else if (younger == 1) { if (wid == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 9); else __builtin_amdgcn_s_waitcnt(0x0F70 | 8); }
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int y0 = b * GROUP_ROWS;
        f32x4_t acc[GROUP_ROWS];
#pragma unroll
        for (int r = 0; r < GROUP_ROWS; ++r) acc[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // The fragment reads are inline asm: hipcc gives a C++ LDS load behind a pending LDS-DMA an s_waitcnt vmcnt(0), which would
        // make every band wait for the groups it has just started to prefetch.  Software pipeline over the 36 (input row, dx, channel
        // half) steps: the reads of steps k + 1 and k + 2 are in flight while the MFMAs of step k issue (LDS returns in order).
        constexpr int NSTEP = (GROUP_ROWS + 2) * 3 * 2;
        u32x4_t fa[3];
        auto issue = [&](int k, u32x4_t& v) {
            const int R = k / 6, dx = (k - R * 6) >> 1, ks = k & 1;
            const int row = y0 + R, px = 16 * wid + dx + (lane & 15);                  // padded input row / pixel of this lane's fragment row
            const unsigned pb = ring_base + (unsigned)(((row >> 2) % NGROUP_RING) * GROUP_B + (row & 3) * ROWB + px * 128);
            const unsigned a0 = pb + ((((unsigned)(ks * 4 + (lane >> 4))) ^ (unsigned)(px & 7)) << 4);
            asm volatile("ds_read_b128 %0, %1" : "=&v"(v) : "v"(a0) : "memory");
        };
        auto mac = [&](int k, const u32x4_t& v) {
            const int R = k / 6, dx = (k - R * 6) >> 1, ks = k & 1;
            const bf16x8_t af = __builtin_bit_cast(bf16x8_t, v);
#pragma unroll
            for (int r = 0; r < GROUP_ROWS; ++r) {
                const int dy = R - r;                                                  // output row y0 + r reads padded input rows y0 + r + {0, 1, 2}
                if (dy < 0 || dy > 2) continue;
                const int t = (2 - dy) * 3 + (2 - dx);                                  // transposed conv: padded offset (dy, dx) <-> kernel tap (2 - dy, 2 - dx)
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[t][ks], acc[r], 0, 0, 0);
            }
        };
        issue(0, fa[0]);
        issue(1, fa[1]);
#pragma unroll
        for (int k = 0; k < NSTEP; ++k) {
            u32x4_t& cur = fa[k % 3];
            if (k + 2 < NSTEP) { issue(k + 2, fa[(k + 2) % 3]); asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(cur)::"memory"); }
            else if (k + 1 < NSTEP) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cur)::"memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur)::"memory");
            mac(k, cur);
        }
        // C layout of the 16x16 MFMA: column (= output channel) lane % 16, rows (= pixels) 4 * (lane / 16) + 0..3
        if ((lane & 15) < NCO) {
#pragma unroll
            for (int r = 0; r < GROUP_ROWS; ++r) {
                f32x4_t v = acc[r];
                if (do_sigmoid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
                }
                *reinterpret_cast<f32x4_t*>(obase + (size_t)(lane & 15) * plane + (size_t)(y0 + r) * OW_ + 16 * wid + 4 * (lane >> 4)) = v;
            }
        }
    }
  }
}

}  // namespace

extern "C" int srvp_conv_out_eligible(int C0, int H, int W, int Cout_real, int k, int s, int p) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SRVP_CONV_OUT_STREAM"); on = e ? atoi(e) : 1; }
    return on && C0 == 64 && H == 64 && W == 64 && Cout_real >= 1 && Cout_real <= 3 && k == 3 && s == 1 && p == 1;
}

extern "C" int srvp_conv_out_fwd(const void* act, const void* wt_tapmajor, float* out, int N, int Cout_real, int sigmoid, void* stream) {
    SRVP_REQUIRE(act && wt_tapmajor && out && N > 0 && Cout_real >= 1 && Cout_real <= 3, "srvp_conv_out_fwd: bad args (1..3 output channels)");
    SRVP_REQUIRE((long long)N * PW_ * PW_ * 64 < (1ll << 40), "srvp_conv_out_fwd: too many frames");
    // one workgroup per CU (101 KB of LDS, 512 registers per lane), each walking N / grid images
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount; if (ncu <= 0) ncu = 256; }
    const int HS = N >= 4 * ncu ? 1 : (N >= 2 * ncu ? 2 : 4);
    const long long items = (long long)N * HS;
    const dim3 g((unsigned)(items < ncu ? items : ncu)), b(256);
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* a = (const bf16_t*)act;
    const uint32_t* w = (const uint32_t*)wt_tapmajor;
    if (Cout_real == 3) hipLaunchKernelGGL(conv_out_stream_kernel<3>, g, b, 0, st, a, w, out, N, sigmoid, HS);
    else if (Cout_real == 2) hipLaunchKernelGGL(conv_out_stream_kernel<2>, g, b, 0, st, a, w, out, N, sigmoid, HS);
    else hipLaunchKernelGGL(conv_out_stream_kernel<1>, g, b, 0, st, a, w, out, N, sigmoid, HS);
    SRVP_CHECK_LAUNCH("srvp_conv_out_fwd");
    return SRVP_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Image-side OUTPUT layer of the DCGAN decoder, the same way: ConvTranspose2d(64 -> nc, 4x4, stride 2, pad 1) + sigmoid from 32x32 to 64x64
// (reference module/conv.py:304-305), bf16 activations [N][34][34][64] (1-pixel zero border) in, fp32 (N, nc, 64, 64) frames out.  Through
// round 5 this layer ran on the MFMA tile kernel as four phase convolutions with nc = 1 padded to a 32-column tile: 0.31 ms at 1920 frames
// (config 2) for 283 MB in / 31 MB out -- 55 us at HBM speed.
//
// Sub-pixel form: the four outputs (2i + a, 2j + b) of low-resolution position (i, j) read the 3x3 window of padded input pixels
// (i + dy, j + dx), dy, dx in 0..2; phase a uses kernel row kh = KH[a][dy] (a = 0: dy 0 -> 3, 1 -> 1; a = 1: dy 1 -> 2, 2 -> 0; the third dy
// does not reach that phase), likewise b / dx.  So the (phase, channel) pairs -- 4 nc <= 12 of them -- are the N dimension of
// v_mfma_f32_16x16x32_bf16, column (a nc + c) 2 + b, with a zero weight column where a window tap does not reach a phase; M = 16 consecutive
// low-resolution columns, K = 64 channels x 9 window taps = 18 MFMAs per 16 positions (64 output pixels).  Same machinery as the kernel above:
// one persistent workgroup per CU, a rolling LDS window of four groups of four padded input rows filled by LDS-DMA two groups ahead, the 18 B
// fragments in registers for the whole launch -- here formed by the kernel itself from the fp32 master weights (Cin, nc, 4, 4), rounded to
// bf16 once (RNE, what srvp_pack_weight produces), so the layer needs no packed copy.  Wave w = (column tile w & 1, row pair w >> 1) of a
// 4-row band.  Epilogue: lanes l and l ^ 1 hold the b = 0 / 1 outputs of the same four positions; after one exchange each stores 16
// contiguous bytes of an output row.
namespace {

constexpr int UPW = 34, UROWB = UPW * 128;                   // padded low-resolution row: 34 pixels x 64 channels x 2 bytes
constexpr int UGROUP_B = GROUP_ROWS * UROWB;                 // 17408 bytes
constexpr int USLOTS = GROUP_ROWS * UPW * 8;                 // 16-byte pieces per group = 1088 = 4.25 x 256

template <int NCO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_up_out_stream_kernel(
    const bf16_t* __restrict__ act, const float* __restrict__ w32, float* __restrict__ out, int N, int do_sigmoid, int HS) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[NGROUP_RING * UGROUP_B];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ct = wid & 1, rh = wid >> 1;                    // this wave: low-resolution columns [16 ct, 16 ct + 16), rows 2 rh, 2 rh + 1 of the band
    const bf16_t* img = act;
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
    auto stage = [&](int g) {          // padded rows 4g .. 4g+3 into ring slot g % 4 (rows past 33 do not exist: the last group holds two)
        const int row0 = g * GROUP_ROWS;
        unsigned char* dst = ring + (g % NGROUP_RING) * UGROUP_B;
        const bf16_t* src = img + (size_t)row0 * UPW * 64;
        const int valid_rows = UPW - row0 < GROUP_ROWS ? UPW - row0 : GROUP_ROWS;
        const unsigned lim = (unsigned)valid_rows * UPW * 64;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i == 4 && wid != 0) continue;                                          // the fifth round is a quarter round (wave 0 only)
            const unsigned o = (svalid[i] && soff[i] < lim) ? soff[i] : 0u;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + o), (lptr_t)(dst + ((size_t)i * 256 + wid * 64) * 16), 16, 0, 0);
        }
    };
    // ---- B fragments from the fp32 master weights W[ci][c][kh][kw]: lane l = column l % 16 = (a NCO + c) 2 + b, input channels
    // ks * 32 + (l / 16) * 8 .. + 7; window tap t = dy * 3 + dx
    bf16x8_t wf[9][2];
    {
        const int col = lane & 15, bq = col & 1, ac = col >> 1, aq = ac / NCO, cq = ac - aq * NCO;
        const bool real = col < 4 * NCO;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3, dx = t - dy * 3;
            const int kh = aq == 0 ? (dy == 0 ? 3 : (dy == 1 ? 1 : -1)) : (dy == 1 ? 2 : (dy == 2 ? 0 : -1));
            const int kw = bq == 0 ? (dx == 0 ? 3 : (dx == 1 ? 1 : -1)) : (dx == 1 ? 2 : (dx == 2 ? 0 : -1));
            const bool on = real && kh >= 0 && kw >= 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ci = ks * 32 + (lane >> 4) * 8 + e;
                    // (unconditional load at a clamped index, select afterwards: a guarded load is a branch + s_waitcnt per element)
                    const float raw = w32[(((size_t)ci * NCO + (real ? cq : 0)) * 4 + (kh >= 0 ? kh : 0)) * 4 + (kw >= 0 ? kw : 0)];
                    f[e] = (__bf16)(on ? raw : 0.f);          // (round to nearest even: v_cvt_pk_bf16_f32)
                }
                wf[t][ks] = f;
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(wf[t][ks]));
    }
    const unsigned ring_base = (unsigned)(uintptr_t)ring;
    const size_t plane = (size_t)OW_ * OW_;
    constexpr int NBAND = 32 / GROUP_ROWS;                    // 8 bands of 4 low-resolution rows per image
    const int nb = NBAND / HS;
    for (int item = blockIdx.x; item < N * HS; item += gridDim.x) {
        const int n = item / HS, b0 = (item - n * HS) * nb, b1 = b0 + nb;       // bands [b0, b1) need the row groups b0 .. b1
        img = act + (size_t)n * UPW * UPW * 64;
        float* obase = out + (size_t)n * NCO * plane;
        __syncthreads();
        stage(b0);
        stage(b0 + 1);
        if (b0 + 2 <= b1) stage(b0 + 2);
        for (int b = b0; b < b1; ++b) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (b + 3 <= b1) stage(b + 3);
            const int younger = (b + 3 <= b1 ? b + 3 : b1) - (b + 1);
            // outstanding operations allowed behind group b + 1: the younger groups' pieces (5 per group for wave 0, 4 for the others) and, from
            // the second band on, the previous band's two frame stores (vmcnt counts stores and retires in order)
            if (b > b0) {
                if (younger >= 2) { if (wid == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 12); else __builtin_amdgcn_s_waitcnt(0x0F70 | 10); }
                else if (younger == 1) { if (wid == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 7); else __builtin_amdgcn_s_waitcnt(0x0F70 | 6); }
                else __builtin_amdgcn_s_waitcnt(0x0F70 | 2);
            } else if (younger >= 2) { if (wid == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 10); else __builtin_amdgcn_s_waitcnt(0x0F70 | 8); }
            else if (younger == 1) { if (wid == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 5); else __builtin_amdgcn_s_waitcnt(0x0F70 | 4); }
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int y0 = b * GROUP_ROWS + 2 * rh;           // first low-resolution row of this wave (= first padded input row it reads)
            f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
            constexpr int NSTEP = 4 * 3 * 2;                  // (padded input row R = 0..3, dx, channel half)
            u32x4_t fa[3];
            auto issue = [&](int k, u32x4_t& v) {
                const int R = k / 6, dx = (k - R * 6) >> 1, ks = k & 1;
                const int row = y0 + R, px = 16 * ct + dx + (lane & 15);
                const unsigned pb = ring_base + (unsigned)(((row >> 2) % NGROUP_RING) * UGROUP_B + (row & 3) * UROWB + px * 128);
                const unsigned a0 = pb + ((((unsigned)(ks * 4 + (lane >> 4))) ^ (unsigned)(px & 7)) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=&v"(v) : "v"(a0) : "memory");
            };
            auto mac = [&](int k, const u32x4_t& v) {
                const int R = k / 6, dx = (k - R * 6) >> 1, ks = k & 1;
                const bf16x8_t af = __builtin_bit_cast(bf16x8_t, v);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int dy = R - r;                     // output low-resolution row y0 + r reads padded rows y0 + r + {0, 1, 2}
                    if (dy < 0 || dy > 2) continue;
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[dy * 3 + dx][ks], acc[r], 0, 0, 0);
                }
            };
            issue(0, fa[0]);
            issue(1, fa[1]);
#pragma unroll
            for (int k = 0; k < NSTEP; ++k) {
                u32x4_t& cur = fa[k % 3];
                if (k + 2 < NSTEP) { issue(k + 2, fa[(k + 2) % 3]); asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(cur)::"memory"); }
                else if (k + 1 < NSTEP) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cur)::"memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur)::"memory");
                mac(k, cur);
            }
            // C layout: column (a, c, b) = lane % 16, rows = positions 16 ct + 4 (lane / 16) + 0..3.  Lanes l / l ^ 1 (b = 0 / 1) exchange halves:
            // the b = 0 lane ends up with output pixels 2 j0 .. 2 j0 + 3, the b = 1 lane with 2 j0 + 4 .. 2 j0 + 7 of output row 2 (y0 + r) + a
            const int col = lane & 15, bq = col & 1, ac = col >> 1, aq = ac / NCO, cq = ac - aq * NCO;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                f32x4_t v = acc[r];
                if (do_sigmoid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
                }
                // send what the partner needs (b = 0 sends its positions 2, 3; b = 1 sends its positions 0, 1), receive likewise
                const float s0 = bq ? v[0] : v[2], s1 = bq ? v[1] : v[3];
                // (DPP quad_perm [1, 0, 3, 2]: lane ^ 1 inside the VALU -- no LDS operation, which hipcc would fence behind the pending LDS-DMA)
                const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, false));
                const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, false));
                f32x4_t o;
                if (bq == 0) o = f32x4_t{v[0], r0, v[1], r1};          // pixels 2 j0 + {0, 1, 2, 3} = (b0 e0, b1 e0, b0 e1, b1 e1)
                else o = f32x4_t{r0, v[2], r1, v[3]};                  // pixels 2 j0 + {4, 5, 6, 7} = (b0 e2, b1 e2, b0 e3, b1 e3)
                if (col < 4 * NCO)
                    *reinterpret_cast<f32x4_t*>(obase + (size_t)cq * plane + (size_t)(2 * (y0 + r) + aq) * OW_ + 2 * (16 * ct + 4 * (lane >> 4)) + 4 * bq) = o;
            }
        }
    }
}

}  // namespace

extern "C" int srvp_conv_up_out_eligible(int C0, int Hin, int Win, int Cout_real, int k, int s, int p) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SRVP_CONV_OUT_STREAM"); on = e ? atoi(e) : 1; }
    return on && C0 == 64 && Hin == 32 && Win == 32 && Cout_real >= 1 && Cout_real <= 3 && k == 4 && s == 2 && p == 1;
}

extern "C" int srvp_conv_up_out_fwd(const void* act, const float* w_f32, float* out, int N, int Cout_real, int sigmoid, void* stream) {
    SRVP_REQUIRE(act && w_f32 && out && N > 0 && Cout_real >= 1 && Cout_real <= 3, "srvp_conv_up_out_fwd: bad args (1..3 output channels)");
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount; if (ncu <= 0) ncu = 256; }
    const int HS = N >= 4 * ncu ? 1 : (N >= 2 * ncu ? 2 : 4);
    const long long items = (long long)N * HS;
    const dim3 g((unsigned)(items < ncu ? items : ncu)), b(256);
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* a = (const bf16_t*)act;
    if (Cout_real == 3) hipLaunchKernelGGL(conv_up_out_stream_kernel<3>, g, b, 0, st, a, w_f32, out, N, sigmoid, HS);
    else if (Cout_real == 2) hipLaunchKernelGGL(conv_up_out_stream_kernel<2>, g, b, 0, st, a, w_f32, out, N, sigmoid, HS);
    else hipLaunchKernelGGL(conv_up_out_stream_kernel<1>, g, b, 0, st, a, w_f32, out, N, sigmoid, HS);
    SRVP_CHECK_LAUNCH("srvp_conv_up_out_fwd");
    return SRVP_OK;
}
