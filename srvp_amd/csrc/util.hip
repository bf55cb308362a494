// Error plumbing + small utility kernels of libsrvp_hip.so.
#include "common.h"
#include "../../include/srvp_hip.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";
int g_srvp_det = 0;
void* g_srvp_det_ws = nullptr;
long long g_srvp_det_ws_bytes = 0;
void srvp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* srvp_last_error(void) { return g_err; }
extern "C" int srvp_version(void) { return 1; }

// A non-blocking stream of the LOWEST priority the device offers: the product's second stream (weight gradients, packing) -- work that
// feeds nothing but the optimizer -- so that whenever both streams have workgroups ready the dispatcher serves the critical path first.
extern "C" int srvp_stream_create_low_priority(void** stream_out, int* priority_out) {
    SRVP_REQUIRE(stream_out, "srvp_stream_create_low_priority: null pointer");
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    SRVP_REQUIRE(e == hipSuccess, "hipDeviceGetStreamPriorityRange: %s", hipGetErrorString(e));
    hipStream_t st = nullptr;
    e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least);
    SRVP_REQUIRE(e == hipSuccess, "hipStreamCreateWithPriority(%d): %s", least, hipGetErrorString(e));
    *stream_out = (void*)st;
    if (priority_out) *priority_out = least;
    return SRVP_OK;
}

namespace {
__global__ void fill_f64_kernel(double* p, long long n, double v) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
template <class E>
__global__ void cast_f32_kernel(const float* src, E* dst, long long rows, int cols, int dst_cols) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * dst_cols) return;
    long long r = i / dst_cols;
    int c = (int)(i - r * dst_cols);
    El<E>::st(dst + i, c < cols ? src[r * cols + c] : 0.f);
}
}  // namespace

extern "C" int srvp_fill_f64(double* p, int64_t n, double v, void* stream) {
    if (n <= 0) return SRVP_OK;
    hipLaunchKernelGGL(fill_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, (long long)n, v);
    SRVP_CHECK_LAUNCH("srvp_fill_f64");
    return SRVP_OK;
}
// uint8 videos [B][T][H][W][C] (the per-video arrays of reference data/base.py:71-84 stacked) -> float32 frames (T, B, C, H, W) / 255
namespace {
__global__ void frames_u8_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, int B, int T, int H, int W, int C) {
    const long long n = (long long)T * B * C * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W); long long q = i / W;
        const int y = (int)(q % H); q /= H;
        const int c = (int)(q % C); q /= C;
        const int b = (int)(q % B); const int t = (int)(q / B);
        out[i] = (float)in[((((size_t)b * T + t) * H + y) * W + x) * C + c] / 255.f;
    }
}
}  // namespace
extern "C" int srvp_frames_u8_to_f32(const void* in, float* out, int B, int T, int H, int W, int C, void* stream) {
    SRVP_REQUIRE(in && out && B > 0 && T > 0 && H > 0 && W > 0 && C > 0, "srvp_frames_u8_to_f32: bad args");
    const long long n = (long long)T * B * C * H * W;
    long long blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(frames_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)in, out, B, T, H, W, C);
    SRVP_CHECK_LAUNCH("srvp_frames_u8_to_f32");
    return SRVP_OK;
}

// Stochastic Moving-MNIST rasteriser (the frame assembly of reference data/mmnist.py:116-124 + the collate of
// data/base.py:71-84): every video b of the batch is the clamped sum of its digits stamped at the trajectory positions,
// out (T, B, 1, nx, nx) float32 = min(255, sum) / 255.  pos[b][n][t] = (row offset, column offset) of digit n at frame t.
namespace {
__global__ void mmnist_render_kernel(const unsigned char* __restrict__ digits, const int* __restrict__ idx,
                                     const int* __restrict__ pos, float* __restrict__ out, unsigned char* __restrict__ out_u8,
                                     int B, int T, int nd, int dh, int dw, int nx) {
    const long long n = (long long)T * B * nx * nx;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % nx); long long q = i / nx;
        const int r = (int)(q % nx); q /= nx;
        const int b = (int)(q % B); const int t = (int)(q / B);
        int s = 0;
        for (int d = 0; d < nd; ++d) {
            const int* pp = pos + (((size_t)b * nd + d) * T + t) * 2;
            const int rr = r - pp[0], cc = c - pp[1];
            if (rr >= 0 && rr < dh && cc >= 0 && cc < dw) s += digits[((size_t)idx[b * nd + d] * dh + rr) * dw + cc];
        }
        s = s > 255 ? 255 : s;
        if (out) out[i] = (float)s / 255.f;
        if (out_u8) out_u8[((size_t)b * T + t) * nx * nx + (size_t)r * nx + c] = (unsigned char)s;
    }
}
}  // namespace
extern "C" int srvp_mmnist_render(const void* digits_u8, int n_digits, int dh, int dw, const int* idx, const int* pos, int B, int T,
                                  int num_digits, int nx, float* out, void* out_u8, void* stream) {
    SRVP_REQUIRE(digits_u8 && idx && pos && (out || out_u8), "srvp_mmnist_render: null pointer");
    SRVP_REQUIRE(n_digits > 0 && dh > 0 && dw > 0 && B > 0 && T > 0 && num_digits > 0 && nx >= dh && nx >= dw,
                 "srvp_mmnist_render: bad sizes");
    const long long n = (long long)T * B * nx * nx;
    long long blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mmnist_render_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)digits_u8, idx, pos, out, (unsigned char*)out_u8, B, T, num_digits, dh, dw, nx);
    SRVP_CHECK_LAUNCH("srvp_mmnist_render");
    return SRVP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Stochastic Moving-MNIST TRAJECTORIES on the device (the process of reference data/mmnist.py:113-237: per object a
// random digit, a uniform start position and integer speed, straight motion inside the box [0, nx - dh] x [0, nx - dw], and at
// every wall contact a fresh uniform speed pointed back inside (stochastic variant) or a plain reflection (deterministic
// variant), the rest of the time step spent with the new speed).  One thread per object; the bounce loop lives in registers.
//
// Formulation (not the reference's "step outside, then find the wall that was crossed" but the equivalent ray / box walk): with
// time budget tau = 1 per frame, the time to the first wall along the motion is th = min over the axes of (wall - s) / v; the
// object moves min(th, tau), and at a contact (th < tau) the speed is redrawn / reflected and the walk continues with the
// remaining budget.  An exact corner contact (both axes within 1e-12) turns both components, as the reference does.
//
// Randomness: counter-based Philox4x32-10, key = seed, counter = (block of four draws, object, batch counter): any batch of any
// run is reproducible from (seed, batch index) alone, on any number of ranks -- and NOT the reference's global np.random stream
// (a sequential Mersenne twister with a data-dependent number of draws per object cannot be reproduced in parallel); equality with
// the reference is distributional (tests/test_gpu_metrics.py), equality with the CPU restatement oracle/mmnist_ref.py (philox_trajectories) is exact.
namespace {
struct Philox {
    unsigned k0, k1, c0, c1, c2, c3, out[4];
    int have;
    __device__ void init(unsigned long long seed, unsigned obj, unsigned long long batch) {
        k0 = (unsigned)seed; k1 = (unsigned)(seed >> 32); c0 = 0; c1 = obj; c2 = (unsigned)batch; c3 = (unsigned)(batch >> 32); have = 0;
    }
    __device__ void block() {
        unsigned a0 = c0, a1 = c1, a2 = c2, a3 = c3, x0 = k0, x1 = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * a0, p1 = (unsigned long long)0xCD9E8D57u * a2;
            const unsigned n0 = (unsigned)(p1 >> 32) ^ a1 ^ x0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ a3 ^ x1, n3 = (unsigned)p0;
            a0 = n0; a1 = n1; a2 = n2; a3 = n3;
            x0 += 0x9E3779B9u; x1 += 0xBB67AE85u;
        }
        out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3;
        ++c0; have = 4;
    }
    __device__ unsigned next() {
        if (!have) block();
        const unsigned v = have == 4 ? out[0] : have == 3 ? out[1] : have == 2 ? out[2] : out[3];
        --have;
        return v;
    }
    // uniform integer in [0, n): Lemire's multiply-shift with rejection (unbiased)
    __device__ unsigned below(unsigned n) {
        unsigned long long m = (unsigned long long)next() * n;
        unsigned l = (unsigned)m;
        if (l < n) {
            const unsigned t = (0u - n) % n;
            while (l < t) { m = (unsigned long long)next() * n; l = (unsigned)m; }
        }
        return (unsigned)(m >> 32);
    }
};

__global__ void mmnist_traj_kernel(unsigned long long seed, unsigned long long batch, int B, int nd, int T, int x_max, int y_max,
                                   int max_speed, int deterministic, int n_digits, int* __restrict__ idx, int* __restrict__ pos,
                                   int* __restrict__ contacts) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= B * nd) return;
    Philox g;
    g.init(seed, (unsigned)o, batch);
    idx[o] = (int)g.below((unsigned)n_digits);
    double sx = (double)g.below((unsigned)x_max + 1u), sy = (double)g.below((unsigned)y_max + 1u);
    const unsigned span = 2u * (unsigned)max_speed + 1u;
    int vx = (int)g.below(span) - max_speed, vy = (int)g.below(span) - max_speed;
    int nc = 0;
    for (int t = 0; t < T; ++t) {
        pos[((size_t)o * T + t) * 2 + 0] = (int)rint(sx);
        pos[((size_t)o * T + t) * 2 + 1] = (int)rint(sy);
        double tau = 1.0;
        for (int it = 0; it < 64 && tau > 0.0; ++it) {
            const double inf = 1e300;
            const double tx = vx > 0 ? ((double)x_max - sx) / vx : (vx < 0 ? (0.0 - sx) / vx : inf);
            const double ty = vy > 0 ? ((double)y_max - sy) / vy : (vy < 0 ? (0.0 - sy) / vy : inf);
            const double th = tx < ty ? tx : ty;
            if (th >= tau) { sx += vx * tau; sy += vy * tau; break; }
            const bool hx = tx <= th + 1e-12, hy = ty <= th + 1e-12;
            sx += vx * th; sy += vy * th; tau -= th;
            const int wx = hx ? (vx > 0 ? 1 : -1) : 0, wy = hy ? (vy > 0 ? 1 : -1) : 0;     // which wall: +1 = far, -1 = near
            if (hx) sx = vx > 0 ? (double)x_max : 0.0;
            if (hy) sy = vy > 0 ? (double)y_max : 0.0;
            if (!deterministic) { vx = (int)g.below(span) - max_speed; vy = (int)g.below(span) - max_speed; }
            if (wx) vx = wx > 0 ? -abs(vx) : abs(vx);
            if (wy) vy = wy > 0 ? -abs(vy) : abs(vy);
            ++nc;
        }
        // (guard against accumulated rounding: the object never leaves the box)
        sx = sx < 0.0 ? 0.0 : (sx > x_max ? (double)x_max : sx);
        sy = sy < 0.0 ? 0.0 : (sy > y_max ? (double)y_max : sy);
    }
    if (contacts) contacts[o] = nc;
}
}  // namespace
extern "C" int srvp_mmnist_trajectories(uint64_t seed, uint64_t batch_counter, int B, int num_digits, int T, int nx, int dh, int dw,
                                        int max_speed, int deterministic, int n_digits, int* idx, int* pos, int* contacts, void* stream) {
    SRVP_REQUIRE(idx && pos && B > 0 && num_digits > 0 && T > 0 && nx >= dh && nx >= dw && dh > 0 && dw > 0 && max_speed >= 0 && n_digits > 0,
                 "srvp_mmnist_trajectories: bad args");
    const int n = B * num_digits;
    hipLaunchKernelGGL(mmnist_traj_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, (unsigned long long)seed,
                       (unsigned long long)batch_counter, B, num_digits, T, nx - dh, nx - dw, max_speed, deterministic, n_digits, idx, pos, contacts);
    SRVP_CHECK_LAUNCH("srvp_mmnist_trajectories");
    return SRVP_OK;
}

extern "C" int srvp_cast_f32_bf16(const float* src, void* dst, int64_t rows, int cols, int dst_cols, void* stream) {
    long long n = (long long)rows * dst_cols;
    if (n <= 0) return SRVP_OK;
    hipLaunchKernelGGL(cast_f32_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst, (long long)rows, cols, dst_cols);
    SRVP_CHECK_LAUNCH("srvp_cast_f32_bf16");
    return SRVP_OK;
}
// bf16 -> fp32 with a scale: the way back of the opt-in bf16 gradient payload (SRVP_GRAD_BF16=1: a gradient slice is cast to bf16,
// all-reduced at half the bytes over xGMI, and widened back into the flat fp32 gradient buffer)
namespace {
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long long n, float scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = bf2f(src[i]) * scale;
}
}  // namespace
extern "C" int srvp_cast_bf16_f32(const void* src, float* dst, int64_t n, float scale, void* stream) {
    SRVP_REQUIRE(src && dst, "srvp_cast_bf16_f32: null pointer");
    if (n <= 0) return SRVP_OK;
    long long b = (n + 255) / 256;
    if (b > 16384) b = 16384;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, (long long)n, scale);
    SRVP_CHECK_LAUNCH("srvp_cast_bf16_f32");
    return SRVP_OK;
}
// fp32 parity mode: the same zero-padding copy into an fp32 tensor
extern "C" int srvp_pad_f32(const float* src, float* dst, int64_t rows, int cols, int dst_cols, void* stream) {
    long long n = (long long)rows * dst_cols;
    if (n <= 0) return SRVP_OK;
    hipLaunchKernelGGL(cast_f32_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       (long long)rows, cols, dst_cols);
    SRVP_CHECK_LAUNCH("srvp_pad_f32");
    return SRVP_OK;
}

// Sum of the split-K slabs of srvp_conv_mfma(splitk > 1) in a fixed order -> bf16 rows (+ the BatchNorm batch statistics the
// convolution epilogue would have formed from its fp32 accumulators: per-column sum / sum of squares, fp64 atomics).
namespace {
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ parts, int splitk, long long slab, long long M, int C,
                                                            bf16_t* __restrict__ dst, double* __restrict__ stats, int stat_mod, int rows_per_block) {
    __shared__ float red[4][64][8];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    const int cg = C >> 2;
    for (int c4 = tx; c4 < ((cg + 63) / 64) * 64; c4 += 64) {
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        if (c4 < cg)
            for (long long r = r0 + ty; r < r1; r += 4) {
                const float* p = parts + r * C + c4 * 4;
                // all slab loads of a group of 8 in flight before the first add (the sum order stays z ascending)
                f32x4_t v = *reinterpret_cast<const f32x4_t*>(p);
                for (int z0 = 1; z0 < splitk; z0 += 8) {
                    f32x4_t u[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) u[i] = z0 + i < splitk ? *reinterpret_cast<const f32x4_t*>(p + (size_t)(z0 + i) * slab) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 8; ++i) if (z0 + i < splitk) { v[0] += u[i][0]; v[1] += u[i][1]; v[2] += u[i][2]; v[3] += u[i][3]; }
                }
                bf16_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = f2bf(v[e]); s1[e] += v[e]; s2[e] += v[e] * v[e]; }
                *reinterpret_cast<unsigned long long*>(dst + r * C + c4 * 4) = *reinterpret_cast<const unsigned long long*>(o);
            }
        if (stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[ty][tx][e] = s1[e]; red[ty][tx][4 + e] = s2[e]; }
            __syncthreads();
            if (ty == 0 && c4 < cg) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    double a = 0., b = 0.;
                    for (int w = 0; w < 4; ++w) { a += red[w][tx][e]; b += red[w][tx][4 + e]; }
                    const int ch = (c4 * 4 + e) % stat_mod;
                    atomicAdd(stats + ch, a);
                    atomicAdd(stats + stat_mod + ch, b);
                }
            }
            __syncthreads();
        }
    }
}
}  // namespace
extern "C" int srvp_splitk_finish(const float* parts, int splitk, int64_t slab_elems, int64_t M, int C, void* dst, double* stats,
                                  int stat_mod, void* stream) {
    SRVP_REQUIRE(parts && dst && splitk >= 1 && M > 0 && C > 0 && C % 4 == 0 && slab_elems >= M * C, "srvp_splitk_finish: bad args");
    SRVP_REQUIRE(!stats || stat_mod > 0, "srvp_splitk_finish: stat_mod");
    // latency-bound: one row per thread -- unless statistics are wanted: 2 C fp64 atomics per workgroup on 2 C / 16 cache lines made
    // 576 workgroups take 55 us; ~64 workgroups then
    int rpb = 4;
    if (stats && (M + 63) / 64 > rpb) rpb = (int)((M + 63) / 64);
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((M + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, parts, splitk,
                       (long long)slab_elems, (long long)M, C, (bf16_t*)dst, stats, stat_mod, rpb);
    SRVP_CHECK_LAUNCH("srvp_splitk_finish");
    return SRVP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Deterministic mode (fp32 parity mode only).  The workspace (>= 8 MiB of device memory, caller-owned) holds the
// per-workgroup partial sums of the two-launch reductions; launches must come from ONE stream while the mode is on.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int srvp_set_deterministic(int on, void* workspace, int64_t workspace_bytes) {
    SRVP_REQUIRE(!on || (workspace && workspace_bytes >= (8ll << 20)), "srvp_set_deterministic: needs a device workspace of at least 8 MiB");
    g_srvp_det = on ? 1 : 0;
    g_srvp_det_ws = on ? workspace : nullptr;
    g_srvp_det_ws_bytes = on ? workspace_bytes : 0;
    return SRVP_OK;
}
extern "C" int srvp_get_deterministic(void) { return g_srvp_det; }

namespace {
// BatchNorm batch statistics of an fp32 tensor [rows][C] in a fixed order: workgroup w walks rows [w * chunk, ...) sequentially, thread =
// (row lane, channel); the row lanes are added in lane order, the workgroups by det_sum_kernel.  slab: [nwg][2][C] doubles.
__global__ __launch_bounds__(256) void bn_stats_det_kernel(const float* __restrict__ raw, long long rows, int C, long long chunk, double* __restrict__ slab) {
    __shared__ double part[256][2];
    const int Cb = C < 256 ? C : 256, L = 256 / Cb;            // channels per pass, row lanes
    const int cl = threadIdx.x % Cb, lane = threadIdx.x / Cb;
    const long long r0 = (long long)blockIdx.x * chunk;
    long long r1 = r0 + chunk; if (r1 > rows) r1 = rows;
    for (int c0 = 0; c0 < C; c0 += Cb) {
        const int c = c0 + cl;
        double s1 = 0., s2 = 0.;
        if (lane < L && c < C)
            for (long long r = r0 + lane; r < r1; r += L) { const double v = (double)raw[r * C + c]; s1 += v; s2 += v * v; }
        part[threadIdx.x][0] = s1; part[threadIdx.x][1] = s2;
        __syncthreads();
        if (lane == 0 && c < C) {
            double t1 = 0., t2 = 0.;
            for (int l = 0; l < L; ++l) { t1 += part[l * Cb + cl][0]; t2 += part[l * Cb + cl][1]; }
            slab[((size_t)blockIdx.x * 2 + 0) * C + c] = t1;
            slab[((size_t)blockIdx.x * 2 + 1) * C + c] = t2;
        }
        __syncthreads();
    }
}
}  // namespace

// stats[0][c] += sum over rows of raw[r][c], stats[1][c] += sum of squares (fp64, fixed summation order): what the convolution
// epilogues add with atomics, recomputed from the stored fp32 raw output (= the accumulators, unrounded, in fp32 mode)
extern "C" int srvp_bn_stats_f32_det(const float* raw, int64_t rows, int C, double* stats, void* stream) {
    SRVP_REQUIRE(raw && stats && rows > 0 && C > 0 && C <= 4096, "srvp_bn_stats_f32_det: bad args");
    SRVP_REQUIRE(g_srvp_det && g_srvp_det_ws, "srvp_bn_stats_f32_det: deterministic mode is off (srvp_set_deterministic)");
    int nwg = 256;
    if (rows < 4 * nwg) nwg = (int)((rows + 3) / 4);
    while ((long long)nwg * 2 * C * 8 > g_srvp_det_ws_bytes && nwg > 1) nwg /= 2;
    const long long chunk = (rows + nwg - 1) / nwg;
    nwg = (int)((rows + chunk - 1) / chunk);
    double* slab = (double*)g_srvp_det_ws;
    hipLaunchKernelGGL(bn_stats_det_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, raw, (long long)rows, C, chunk, slab);
    hipLaunchKernelGGL(det_sum_kernel<double>, dim3((2 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double*)slab, nwg, 2 * C, stats);
    SRVP_CHECK_LAUNCH("srvp_bn_stats_f32_det");
    return SRVP_OK;
}
