// precision = 'fp32' parity mode: the tap-table convolution (forward / data-gradient) and its weight gradient on FP32
// tensors in EXACT fp32 on the matrix cores -- v_mfma_f32_32x32x2_f32 multiplies and accumulates in IEEE fp32, bitwise a
// k-ordered fmaf chain -- behind the same descriptors as the bf16 kernels (srvp_conv_desc / srvp_wgrad_desc with
// elem_f32 = 1: two sources, image maps, nearest-x2 upsample, strides, BatchNorm statistics, hoisted-skip S tensor,
// fp32 frame output with sigmoid).  This mode exists to compare the WHOLE HIP pipeline with the reference's fp32
// arithmetic at tight tolerance (north_star: "ELBO within 1e-4 relative"; tests/test_gpu_fp32_mode.py asserts 1e-5 and
// per-parameter gradients to 2e-3 against the reference-generated fixtures); it is not the throughput path: plain
// LDS-staged 128x32 / 32x32 tiles, ~1/16 of the bf16 MFMA rate by construction.
//
// Replaces (in that mode): nn.Conv2d / nn.ConvTranspose2d forward, data- and weight-gradient of reference
// module/conv.py:174-179, 200-223, 299-304, 330-353 incl. torch.cat (conv.py:270) and nn.Upsample (conv.py:331-349).
#include "common.h"
#include "../../include/srvp_hip.h"

namespace {

struct ConvF {
    const float* src0; const float* src1; const int* map1; const int* map0;
    int C0, C1, H0p, W0p, H1p, W1p, ups0, ups1, si, ntaps;
    unsigned long long dy_bits, dx_bits;
    const float* wt; int Cout, N, OH, OW;
    float* dst; int DHp, DWp, so, ooy, oox, Cdst, cdst_off;
    double* stats; int stat_mod;
    float* out_f32; int out_nc, out_sigmoid;
    const float* add_f32; int add_mod;
};

constexpr int LD = 33;          // LDS row stride in floats: lanes 0..31 (rows) x the two k of an MFMA hit 64 distinct banks

// workgroup = 4 waves stacked along the pixel axis: tile 128 pixels x 32 output channels, K step = 32 channels of one tap
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvF a) {
    __shared__ float As[128 * LD];
    __shared__ float Bs[32 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lcol = lane & 31, lhalf = lane >> 5;
    const int Ctot = a.C0 + a.C1;
    const long long M = (long long)a.N * a.OH * a.OW;
    const int n_tiles = a.Cout / 32;
    const long long m0 = (long long)(blockIdx.x / n_tiles) * 128;
    const int n0 = (blockIdx.x % n_tiles) * 32;
    const int hw = a.OH * a.OW;

    // gather rows of this thread: 4 float4 pieces of the A tile (row = q / 8, 4-channel piece = q % 8)
    int rn[4], rn1[4], roy[4], rox[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid + i * 256) >> 3;
        long long m = m0 + row;
        if (m >= M) m = M - 1;                        // clamped duplicates are masked at the store
        const int n = (int)(m / hw), r = (int)(m - (long long)n * hw);
        roy[i] = r / a.OW; rox[i] = r - roy[i] * a.OW;
        rn[i] = a.map0 ? a.map0[n] : n;
        rn1[i] = (a.C1 > 0 && a.map1) ? a.map1[n] : n;
    }
    f32x16_t acc;
    {
        // accumulator start: zero or the hoisted skip half S[sample] (fp32, layout of the unbordered destination)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = 0.f;
            if (a.add_f32) {
                const long long m = m0 + wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                if (m < M) {
                    const int n = (int)(m / hw), q = (int)(m - (long long)n * hw);
                    const int oy = q / a.OW, ox = q - oy * a.OW;
                    v = a.add_f32[(((size_t)(n % a.add_mod) * a.DHp + oy * a.so + a.ooy) * a.DWp + ox * a.so + a.oox) * a.Cout + n0 + lcol];
                }
            }
            acc[r] = v;
        }
    }
    for (int t = 0; t < a.ntaps; ++t) {
        const int dy = (int)((a.dy_bits >> (4 * t)) & 15), dx = (int)((a.dx_bits >> (4 * t)) & 15);
        for (int c = 0; c < Ctot; c += 32) {
            const bool second = c >= a.C0;
            const float* src = second ? a.src1 : a.src0;
            const int C = second ? a.C1 : a.C0, Hp = second ? a.H1p : a.H0p, Wp = second ? a.W1p : a.W0p;
            const int ups = (second ? a.ups1 : a.ups0) ? 1 : 0;
            const int cs = second ? c - a.C0 : c;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = tid + i * 256, row = q >> 3, pc = q & 7;
                int vy = roy[i] * a.si + dy, vx = rox[i] * a.si + dx;
                vy = (vy + ups) >> ups; vx = (vx + ups) >> ups;
                const int n = second ? rn1[i] : rn[i];
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + (((size_t)n * Hp + vy) * Wp + vx) * C + cs + pc * 4);
                float* d = As + row * LD + pc * 4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
            {
                const int row = tid >> 3, pc = tid & 7;     // weight tile [32 output channels][32 k]
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(a.wt + ((size_t)t * a.Cout + n0 + row) * Ctot + c + pc * 4);
                float* d = Bs + row * LD + pc * 4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
            __syncthreads();
            const float* ar = As + (wid * 32 + lcol) * LD + lhalf;
            const float* br = Bs + lcol * LD + lhalf;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[2 * kk], br[2 * kk], acc, 0, 0, 0);
        }
    }
    // ---- epilogue.  C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int col = n0 + lcol;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long m = m0 + wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        if (m >= M) continue;
        const float v = acc[r];
        s1 += v; s2 += v * v;
        const int n = (int)(m / hw), q = (int)(m - (long long)n * hw);
        const int oy = q / a.OW, ox = q - oy * a.OW;
        const int Y = oy * a.so + a.ooy, X = ox * a.so + a.oox;
        if (a.out_f32) {
            if (col < a.out_nc) {
                const float o = a.out_sigmoid ? 1.f / (1.f + __expf(-v)) : v;
                a.out_f32[(((size_t)n * a.out_nc + col) * a.DHp + Y) * a.DWp + X] = o;
            }
        } else {
            a.dst[(((size_t)n * a.DHp + Y) * a.DWp + X) * a.Cdst + a.cdst_off + col] = v;
        }
    }
    if (a.stats) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lhalf == 0) {
            const int ch = col % a.stat_mod;
            atomicAdd(a.stats + ch, (double)s1);
            atomicAdd(a.stats + a.stat_mod + ch, (double)s2);
        }
    }
}

struct WgradF {
    const float* src0; const float* src1; const int* map1; const int* map0;
    int C0, C1, H0p, W0p, H1p, W1p, ups0, ups1, si, ntaps;
    unsigned long long dy_bits, dx_bits, ooy_bits, oox_bits;
    const float* dout; int DHp, DWp, so, Cout;
    int N, OH, OW;
    float* dw; int splitk;
};

// workgroup = one 32 (output channels) x 32 (input channels) tile of one tap over a range of pixel chunks; per K step 128
// pixels are staged and each of the 4 waves contracts its own 32 of them (A = dout^T, B = gathered input rows)
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const WgradF a) {
    __shared__ float Xs[128 * LD];     // dout tile  [pixel][j]
    __shared__ float Ys[128 * LD];     // input tile [pixel][c]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lcol = lane & 31, lhalf = lane >> 5;
    const int Ctot = a.C0 + a.C1;
    const int tj_n = a.Cout / 32, tc_n = Ctot / 32;
    int b = blockIdx.x;
    const int t = b % a.ntaps; b /= a.ntaps;
    const int tc = b % tc_n; b /= tc_n;
    const int tj = b % tj_n; b /= tj_n;
    const int split = b;
    const int j0 = tj * 32, c0 = tc * 32;
    const long long M = (long long)a.N * a.OH * a.OW;
    const long long nchunks = (M + 127) / 128;
    const long long per = (nchunks + a.splitk - 1) / a.splitk;
    const long long ch_beg = (long long)split * per;
    long long ch_end = ch_beg + per; if (ch_end > nchunks) ch_end = nchunks;
    if (ch_beg >= ch_end) return;
    const bool second = c0 >= a.C0;
    const float* src = second ? a.src1 : a.src0;
    const int C = second ? a.C1 : a.C0, Hp = second ? a.H1p : a.H0p, Wp = second ? a.W1p : a.W0p;
    const int ups = (second ? a.ups1 : a.ups0) ? 1 : 0;
    const int cs = second ? c0 - a.C0 : c0;
    const int* map = second ? a.map1 : a.map0;
    const int dy = (int)((a.dy_bits >> (4 * t)) & 15), dx = (int)((a.dx_bits >> (4 * t)) & 15);
    const int ooy = (int)((a.ooy_bits >> (4 * t)) & 15), oox = (int)((a.oox_bits >> (4 * t)) & 15);
    const int hw = a.OH * a.OW;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (long long ch = ch_beg; ch < ch_end; ++ch) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * 256, row = q >> 3, pc = q & 7;
            const long long m = ch * 128 + row;
            f32x4_t vx4 = {0.f, 0.f, 0.f, 0.f}, vy4 = {0.f, 0.f, 0.f, 0.f};
            if (m < M) {
                int n = (int)(m / hw); const int r = (int)(m - (long long)n * hw);
                const int oy = r / a.OW, ox = r - oy * a.OW;
                vx4 = *reinterpret_cast<const f32x4_t*>(a.dout + (((size_t)n * a.DHp + oy * a.so + ooy) * a.DWp + ox * a.so + oox) * a.Cout + j0 + pc * 4);
                int vy = oy * a.si + dy, vx = ox * a.si + dx;
                vy = (vy + ups) >> ups; vx = (vx + ups) >> ups;
                if (map) n = map[n];
                vy4 = *reinterpret_cast<const f32x4_t*>(src + (((size_t)n * Hp + vy) * Wp + vx) * C + cs + pc * 4);
            }
            float* dxp = Xs + row * LD + pc * 4;
            float* dyp = Ys + row * LD + pc * 4;
            dxp[0] = vx4[0]; dxp[1] = vx4[1]; dxp[2] = vx4[2]; dxp[3] = vx4[3];
            dyp[0] = vy4[0]; dyp[1] = vy4[1]; dyp[2] = vy4[2]; dyp[3] = vy4[3];
        }
        __syncthreads();
        // A[row = j][k = pixel] = X[pixel][j], B[k = pixel][col = c] = Y[pixel][c]
        const float* xr = Xs + (wid * 32 + lhalf) * LD + lcol;
        const float* yr = Ys + (wid * 32 + lhalf) * LD + lcol;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[2 * kk * LD], yr[2 * kk * LD], acc, 0, 0, 0);
    }
    __syncthreads();
    float* red = Xs;                    // [3][32][LD]
    if (wid > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wid - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * LD + lcol] = acc[r];
    }
    __syncthreads();
    if (wid > 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        const float v = acc[r] + red[row * LD + lcol] + red[(32 + row) * LD + lcol] + red[(64 + row) * LD + lcol];
        atomicAdd(a.dw + ((size_t)t * a.Cout + j0 + row) * Ctot + c0 + lcol, v);
    }
}

}  // namespace

int srvp_conv_f32_launch(const srvp_conv_desc* d, hipStream_t st) {
    SRVP_REQUIRE(d->wt_fragmajor == 0 && d->tap_phase_chunks == 0, "srvp_conv_mfma(fp32): weights must be tap-major, no space-to-depth sources");
    ConvF k;
    k.src0 = (const float*)d->src0; k.src1 = (const float*)d->src1; k.map1 = d->map1; k.map0 = d->map0;
    k.C0 = d->C0; k.C1 = d->C1; k.H0p = d->H0p; k.W0p = d->W0p; k.H1p = d->H1p; k.W1p = d->W1p;
    k.ups0 = d->ups0; k.ups1 = d->ups1; k.si = d->si; k.ntaps = d->ntaps;
    k.dy_bits = 0; k.dx_bits = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        SRVP_REQUIRE(d->dy[t] >= 0 && d->dy[t] < 16 && d->dx[t] >= 0 && d->dx[t] < 16, "srvp_conv_mfma(fp32): tap offset out of [0,15]");
        k.dy_bits |= (unsigned long long)d->dy[t] << (4 * t);
        k.dx_bits |= (unsigned long long)d->dx[t] << (4 * t);
    }
    k.wt = (const float*)d->wt; k.Cout = d->Cout; k.N = d->N; k.OH = d->OH; k.OW = d->OW;
    k.dst = (float*)d->dst; k.DHp = d->DHp; k.DWp = d->DWp; k.so = d->so; k.ooy = d->ooy; k.oox = d->oox;
    k.Cdst = d->Cdst; k.cdst_off = d->cdst_off; k.stats = d->stats; k.stat_mod = d->stat_mod;
    k.out_f32 = d->out_f32; k.out_nc = d->out_nc; k.out_sigmoid = d->out_sigmoid;
    k.add_f32 = d->add_f32; k.add_mod = d->add_mod;
    const long long M = (long long)d->N * d->OH * d->OW;
    const long long blocks = ((M + 127) / 128) * (d->Cout / 32);
    SRVP_REQUIRE(blocks > 0 && blocks < (1ll << 31), "srvp_conv_mfma(fp32): bad grid %lld", blocks);
    hipLaunchKernelGGL(conv_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, k);
    SRVP_CHECK_LAUNCH("srvp_conv_mfma(fp32)");
    return SRVP_OK;
}

int srvp_wgrad_f32_launch(const srvp_wgrad_desc* d, hipStream_t st) {
    SRVP_REQUIRE(d->dout_cstride == 0 && d->dout_coff == 0 && d->dout_phase_taps == 0, "srvp_wgrad_mfma(fp32): channel-sliced dout is not supported");
    WgradF k;
    k.src0 = (const float*)d->src0; k.src1 = (const float*)d->src1; k.map1 = d->map1; k.map0 = d->map0;
    k.C0 = d->C0; k.C1 = d->C1; k.H0p = d->H0p; k.W0p = d->W0p; k.H1p = d->H1p; k.W1p = d->W1p;
    k.ups0 = d->ups0; k.ups1 = d->ups1; k.si = d->si; k.ntaps = d->ntaps;
    k.dy_bits = k.dx_bits = k.ooy_bits = k.oox_bits = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        SRVP_REQUIRE(d->dy[t] >= 0 && d->dy[t] < 16 && d->dx[t] >= 0 && d->dx[t] < 16 && d->ooy[t] >= 0 && d->ooy[t] < 16 &&
                         d->oox[t] >= 0 && d->oox[t] < 16, "srvp_wgrad_mfma(fp32): tap offset out of [0,15]");
        k.dy_bits |= (unsigned long long)d->dy[t] << (4 * t); k.dx_bits |= (unsigned long long)d->dx[t] << (4 * t);
        k.ooy_bits |= (unsigned long long)d->ooy[t] << (4 * t); k.oox_bits |= (unsigned long long)d->oox[t] << (4 * t);
    }
    k.dout = (const float*)d->dout; k.DHp = d->DHp; k.DWp = d->DWp; k.so = d->so; k.Cout = d->Cout;
    k.N = d->N; k.OH = d->OH; k.OW = d->OW; k.dw = d->dw;
    const long long M = (long long)d->N * d->OH * d->OW;
    const long long nchunks = (M + 127) / 128;
    const long long tiles = (long long)d->ntaps * (d->Cout / 32) * ((d->C0 + d->C1) / 32);
    long long sk = (2048 + tiles - 1) / tiles;          // enough workgroups to fill the chip
    if (sk > nchunks) sk = nchunks;
    if (sk < 1 || g_srvp_det) sk = 1;                   // deterministic mode: one workgroup per weight tile = one atomic per element
    k.splitk = (int)sk;
    const long long blocks = tiles * sk;
    SRVP_REQUIRE(blocks > 0 && blocks < (1ll << 31), "srvp_wgrad_mfma(fp32): bad grid %lld", blocks);
    hipLaunchKernelGGL(wgrad_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, k);
    SRVP_CHECK_LAUNCH("srvp_wgrad_mfma(fp32)");
    return SRVP_OK;
}
