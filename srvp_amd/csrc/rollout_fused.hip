// Persistent fused residual rollout (training / posterior chain): ONE launch runs all n_euler * (T - 1) Euler steps of
//   y_{i+1} = y_i + dt * MLP_dynamics([y_i, z_{frame(i)}])                 (reference module/srvp.py:300-323, 377-405)
// forward, and ONE launch runs the whole backward-through-time chain, instead of (nl GEMMs + 1 update) launches per step
// (236 dependent micro-launches per training step at the headline config, ~6 ms that did not shrink with the batch).
//
// Decomposition: samples are independent, hidden units are not.  The batch is cut into 16-row tiles (32 rows for the generation chain); a
// CLUSTER of G = nh / 32 workgroups (LSTM: nh / 16) owns one row tile, workgroup g of the cluster owns a slice of the hidden columns of every
// layer for the whole kernel, so its weight slices are loaded ONCE into LDS / registers and stay there for all steps -- no weight traffic
// inside the chain.  Per layer the cluster exchanges the activation tile through global memory (the hid_dyn / dhid_dyn tensors the
// weight-gradient GEMMs need anyway) and synchronises with a monotonic counter; the first-in-cluster layer of a step needs no exchange
// (every workgroup carries the y / dy state itself) and the last layer is split-K over the cluster with fixed-order partial sums.
// Inside a workgroup the K loop of a layer (not the output tile) is dealt to the four waves; partial tiles are added in a fixed order through
// LDS.  Arithmetic: exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32 = a k-ordered fmaf chain).
//
// All workgroups of a launch must be co-resident (they spin on each other): the launcher keeps the grid <= the number of CUs and walks
// larger batches in several launches on the same stream (tile0).  Round 6: the 32-row / output-tiled kernels of rounds 2-4 that these forms
// replaced in round 5 (and that stayed as a fallback for B > 256, nh != 512) are gone -- every shape takes the kernels below, the generation
// chain of widths other than 512 takes the per-layer launch sequence of csrc/latent.hip.
#include <atomic>
#include <type_traits>
#include "common.h"
#include "../../include/srvp_hip.h"

namespace {

constexpr int RT = 32;           // rows per cluster tile
constexpr int CW = 32;           // hidden columns per workgroup
constexpr int KP0_MAX = 128;     // padded width of the first-layer input (ny + nz <= 128)
constexpr int NYP_MAX = 64;      // padded ny (<= 64)
constexpr int MAX_NL = 8;

struct RollF {
    int B, ny, nz, nh, nl, S, ne, G, nin, kp0, nyp, ntiles, tile0, cl_per_xcd;
    float dt;
    const float* W[MAX_NL]; const float* b[MAX_NL];
    // forward
    const float* y0; float* y_all; float* res; float* inp_all; float* hid;
    // backward
    const float* d_y_all; const float* d_res; float* dhid; float* dinp_all; float* d_y0; int dwd;
    float* part; unsigned* cnt; int xcd_local;
    unsigned long long* dbg;                              // SRVP_RF_DEBUG=1: per-phase time of cluster 0 / member 0 (10 ns ticks), else null
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

// B-operand slice [K][32] in LDS: element (k, c) at [k / 4][c][k % 4] -- the four consecutive k values one lane feeds to four MFMAs
// are ONE 16-byte read (ds_read_b128; the 16 lanes of a k group read 256 contiguous bytes: conflict-free).  Round 5 (SRVP_RF_DEBUG
// timestamps): with one 4-byte LDS read in front of every v_mfma_f32_16x16x4_f32 and one wave per SIMD the K = 512 GEMM of a hidden
// layer took 5.1 us of a 20 us Euler step -- 96 cycles per MFMA, the LDS latency un-overlapped -- against 1.7 us of MFMA issue.
__device__ __forceinline__ int sw(int k, int c) { return (k >> 2) * 128 + c * 4 + (k & 3); }
constexpr int IPAD = 4;          // padding of the LDS staging tiles' rows (floats): rows stay 16-byte aligned for ds_read_b128 and sixteen
                                 // consecutive rows at one column fall on all 64 banks (row strides 116 / 68 floats: 52 i mod 64 and 4 i mod 64 are distinct multiples of 4)

// Everything the workgroups of a cluster exchange inside the kernel is written and read with AGENT-SCOPE accesses (sc1:
// write-through / miss-always, the code the compiler emits for relaxed agent-scope atomics), so the barrier needs no cache
// maintenance: a whole-L2 write-back + invalidate per barrier (what an agent-scope release / acquire FENCE costs on this
// multi-XCD part while other kernels keep the L2 full of dirty lines) measured 27 us per layer.
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// XCD-LOCAL EXCHANGE (round 5).  The launchers deal the workgroups of a cluster to ONE XCD (locate()); HIP promises nothing about placement, so
// each launch VERIFIES it: every member ORs the bit of the XCC it really runs on (HW_REG_XCC_ID) into word 2 of the cluster's counter block before
// the first barrier and reads the mask back after it.  One bit set = all members share one L2: the exchanged tiles are then written with PLAIN
// stores, which stay in that L2 (write-back), and read by the other members with the same L1-bypassing `sc1` loads as before, now served from
// the L2 they were written to instead of from memory (sc1 stores are written through AND dropped from the L2: MI355X_MICROARCH.md, "stores of
// each flavour").  Any other mask: the agent-scope (sc1 write-through) stores of rounds 2-4.  Never a correctness assumption: a decision per
// launch and per cluster from what the hardware reports.  SRVP_CLUSTER_XCD_LOCAL=0 keeps the write-through stores everywhere (A/B switch).
__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}
__device__ __forceinline__ void st_x(float* p, float v, bool local) {
    if (local) *p = v; else st_agent(p, v);
}

// Timeouts of the bounded spin below, over the lifetime of the library (the per-launch counter blocks are wiped by the next launch):
// the host reads it through srvp_cluster_timeouts_read once in a while and refuses to go on if it is non-zero.
__device__ unsigned g_cluster_timeouts = 0;
// clusters (per launch) that found all their members on one XCC and exchanged through its L2 (srvp_cluster_stats_read)
__device__ unsigned g_cluster_xcd_local = 0;

__device__ __forceinline__ void cluster_barrier(unsigned* cnt, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's agent-scope stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // bounded spin (~seconds): if the cluster's workgroups are NOT co-resident (CU mask / partition / a device shared with
        // another process: the launcher's eligibility test cannot see those) the kernel ends with garbage and a non-zero word
        // word 1 of the tile's 256-byte counter block at the start of the workspace instead of hanging the device
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 25)) {
                __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&g_cluster_timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// see "XCD-LOCAL EXCHANGE" above: announce() before the kernel's first cluster barrier, then agreed() after it (one extra barrier per launch
// where the kernel has none before its first exchange)
__device__ __forceinline__ void xcd_announce(unsigned* cnt) {
    if (threadIdx.x == 0) __hip_atomic_fetch_or(cnt + 2, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool xcd_agreed(const unsigned* cnt, int enabled, bool count_it) {
    const unsigned m = __hip_atomic_load(cnt + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = enabled && m != 0u && (m & (m - 1u)) == 0u;
    if (ok && count_it && threadIdx.x == 0) __hip_atomic_fetch_add(&g_cluster_xcd_local, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ok;
}

// workgroup -> (cluster, member): members of a cluster sit on ONE XCD when workgroups are dealt round-robin to the 8 XCDs
// (speed only: the exchange then stays inside that XCD's L2 -- verified per launch, xcd_announce / xcd_agreed)
__device__ __forceinline__ bool locate(const RollF& a, int& cl, int& g) {
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
    cl = x * a.cl_per_xcd + k / a.G;
    g = k % a.G;
    return cl < a.ntiles;
}

// Sum of the G split-K slabs for two (row, 4-column) items per thread: all 2 GP 16-byte agent-scope loads in flight at once
// (a serial per-slab loop of 4-byte loads cost 24-48 us per Euler step: half of the whole kernel), fixed summation order.
// Unconditional loads (slab index clamped): no in-flight register lives across a branch (hipcc places phi copies BEFORE a hand-written wait).
#define WAIT8(v, o) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[o]), "+v"(v[o + 1]), "+v"(v[o + 2]), "+v"(v[o + 3]), "+v"(v[o + 4]), "+v"(v[o + 5]), "+v"(v[o + 6]), "+v"(v[o + 7]) :: "memory")
template <int GP>
__device__ __forceinline__ void sum_slabs2(f32x4v& sa, f32x4v& sb, const float* pa, const float* pb, int G, size_t slab_stride) {
    f32x4v va[GP], vb[GP];
#pragma unroll
    for (int gg = 0; gg < GP; ++gg) {
        const size_t o = (size_t)(gg < G ? gg : G - 1) * slab_stride;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(va[gg]) : "v"(pa + o) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(vb[gg]) : "v"(pb + o) : "memory");
    }
#pragma unroll
    for (int o = 0; o < GP; o += 8) { WAIT8(va, o); WAIT8(vb, o); }
#pragma unroll
    for (int gg = 0; gg < GP; ++gg)
        if (gg < G) { sa += va[gg]; sb += vb[gg]; }
}

// ------------------------------------------------------------------------------------------------------------ 16-row tiles, K split over the waves
// The two training kernels.  What a hidden-layer GEMM costs per Euler step (SRVP_RF_DEBUG timestamps) is moving the activation tile into
// every workgroup plus the MFMA issue behind it; both scale with the ROWS a cluster owns, and the chip has CUs to spare (96 of 256 at 192
// sequences, 16 at 24): 16-row tiles, and the K loop dealt to the four waves: wave w takes the w-th 16-wide k block of every 64 (one
// 16-byte load per lane and block, every byte fetched once per workgroup), feeds it to BOTH 16-column tiles of the slice, and the four
// partial 16 x 32 tiles are added in a fixed order (((w0 + w1) + w2) + w3) through 8 KB of LDS, where the epilogue (bias / ReLU or ReLU
// mask, store, hand-off) then runs on all 256 threads, two outputs each, along rows.
constexpr int RT16 = 16;
#define WAITV1(r, n) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(r) :: "memory")

// acc[ct] (16 x 16, column tiles ct of the 32-column slice) += A(16 rows from global, this wave's k blocks) x B(LDS slice); blocks past K
// contribute exact zeros (A fragment cleared), no branch between a load and its wait
template <bool CACHED>
__device__ __forceinline__ void gemm_ks(f32x4v (&acc)[2], const float* arow, const float* Bs, int K, int w, int q, int c16) {
    const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 512) {
        f32x4v av[8], b0[8], b1[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int kb = k0 + 64 * jj + 16 * w;
            const float* p = arow + (kb < K ? kb : 0) + 4 * q;
            if constexpr (CACHED) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(av[jj]) : "v"(p) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(av[jj]) : "v"(p) : "memory");
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {                  // the B fragments arrive from LDS while the global loads are in flight
            const int kb = k0 + 64 * jj + 16 * w;
            const float* bp = Bs + (((kb < K ? kb : 0) >> 2) + q) * 128 + c16 * 4;
            b0[jj] = *reinterpret_cast<const f32x4v*>(bp);
            b1[jj] = *reinterpret_cast<const f32x4v*>(bp + 64);
        }
#define KS_STEP(jj, n)                                                                                               \
        {                                                                                                            \
            WAITV1(av[jj], n);                                                                                       \
            const f32x4v x = (k0 + 64 * (jj) + 16 * w < K) ? av[jj] : zero;                                          \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                          \
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], b0[jj][e], acc[0], 0, 0, 0);                     \
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], b1[jj][e], acc[1], 0, 0, 0);                     \
            }                                                                                                        \
        }
        KS_STEP(0, 7) KS_STEP(1, 6) KS_STEP(2, 5) KS_STEP(3, 4) KS_STEP(4, 3) KS_STEP(5, 2) KS_STEP(6, 1) KS_STEP(7, 0)
#undef KS_STEP
    }
}

// the four waves' partial tiles -> LDS (MFMA lane layout, 16 bytes per lane and tile); read back per output element by red_sum
__device__ __forceinline__ void red_store(float* Red, const f32x4v (&acc)[2], int w, int lane) {
    *reinterpret_cast<f32x4v*>(Red + (w * 2 + 0) * 256 + lane * 4) = acc[0];
    *reinterpret_cast<f32x4v*>(Red + (w * 2 + 1) * 256 + lane * 4) = acc[1];
}
// element (r, c) of the 16 x 32 tile: rows 4 q + e of lane (q, c & 15) of column tile c >> 4
__device__ __forceinline__ float red_sum(const float* Red, int r, int c) {
    const int off = (c >> 4) * 256 + ((r >> 2) * 16 + (c & 15)) * 4 + (r & 3);
    return ((Red[off] + Red[512 + off]) + Red[1024 + off]) + Red[1536 + off];
}

// the same sum (same order) with half of the loads in flight at a time: for kernels whose registers are spoken for
template <int GP>
__device__ __forceinline__ void sum_slabs2h(f32x4v& sa, f32x4v& sb, const float* pa, const float* pb, int G, size_t slab_stride) {
    constexpr int H = GP / 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4v va[H], vb[H];
#pragma unroll
        for (int gg = 0; gg < H; ++gg) {
            const int gi = h * H + gg;
            const size_t o = (size_t)(gi < G ? gi : G - 1) * slab_stride;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(va[gg]) : "v"(pa + o) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(vb[gg]) : "v"(pb + o) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int gg = 0; gg < H; ++gg) { asm volatile("" : "+v"(va[gg])); asm volatile("" : "+v"(vb[gg])); }
#pragma unroll
        for (int gg = 0; gg < H; ++gg)
            if (h * H + gg < G) { sa += va[gg]; sb += vb[gg]; }
    }
}

template <int GP>
__device__ __forceinline__ void sum_slabs1(f32x4v& sa, const float* pa, int G, size_t slab_stride) {
    f32x4v va[GP];
#pragma unroll
    for (int gg = 0; gg < GP; ++gg) {
        const size_t o = (size_t)(gg < G ? gg : G - 1) * slab_stride;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(va[gg]) : "v"(pa + o) : "memory");
    }
#pragma unroll
    for (int o = 0; o < GP; o += 8) { WAIT8(va, o); }
#pragma unroll
    for (int gg = 0; gg < GP; ++gg)
        if (gg < G) sa += va[gg];
}

template <int GP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rollout_ks_fwd_kernel(const RollF a) {
    extern __shared__ float lds[];
    int cl, g;
    if (!locate(a, cl, g)) return;
    constexpr int RT = RT16;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, c16 = lane & 15;
    const int ec = tid & 31, er = tid >> 5;               // epilogue: outputs (er, ec) and (er + 8, ec) of the 16 x 32 tile
    const int row0 = (a.tile0 + cl) * RT;
    const int colbase = g * CW;
    const int nl = a.nl, nh = a.nh, ny = a.ny, nin = a.nin, B = a.B, kp0 = a.kp0;
    const int nfull = nl - 2;
    // LDS: [nfull][nh][32] slices | Is [16][kp0 + IPAD] | Ys [16][ny] | Hs [16][33] | Bl [64] | Red [4][2][256]
    float* Wl = lds;
    float* Is = Wl + (size_t)nfull * nh * CW;
    float* Ys = Is + RT * (kp0 + IPAD);
    float* Hs = Ys + RT * ny;
    float* Bl = Hs + RT * 33;
    float* Red = Bl + NYP_MAX;
    unsigned* cnt = a.cnt + (size_t)(a.tile0 + cl) * 64;
    float* part = a.part + (size_t)(a.tile0 + cl) * a.S * a.G * RT * KP0_MAX;       // a fresh slab per Euler step
    unsigned target = 0;
    xcd_announce(cnt);

    for (int l = 1; l <= nfull; ++l) {
        const float* W = a.W[l];
        float* dst = Wl + (size_t)(l - 1) * nh * CW;
        for (int idx = tid; idx < nh * CW; idx += 256) {
            const int c = idx / nh, k = idx - c * nh;
            dst[sw(k, c)] = W[(size_t)(colbase + c) * nh + k];
        }
    }
    // first layer: this wave's k blocks j = w + 4 jj of the [kp0] input, both column tiles: B0[k = 16 j + 4 q + e][16 ct + c16]
    float w0[2][2][4];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 16 * (w + 4 * jj) + 4 * q + e;
                w0[jj][ct][e] = k < nin ? a.W[0][(size_t)(colbase + 16 * ct + c16) * nin + k] : 0.f;
            }
    // last layer (split-K over the cluster): wave w owns output column tile w: BL[k = 4 jj + q][col] = W_last[col][colbase + k]
    float wl[8];
    const bool has_lt = 16 * w < a.nyp;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int col = 16 * w + c16;
        wl[jj] = (has_lt && col < ny) ? a.W[nl - 1][(size_t)col * nh + colbase + 4 * jj + q] : 0.f;
    }
    for (int idx = tid; idx < RT * ny; idx += 256) {
        const int r = idx / ny, c = idx - r * ny;
        const int row = row0 + r < B ? row0 + r : B - 1;
        Ys[idx] = a.y0[(size_t)row * ny + c];
    }
    if (tid < ny) Bl[tid] = a.b[nl - 1][tid];
    float zr[KP0_MAX * RT / 256];
    auto fetch_z = [&](int step) {
#pragma unroll
        for (int u = 0; u < KP0_MAX * RT / 256; ++u) {
            const int idx = tid + 256 * u, r = idx / kp0, k = idx - r * kp0;
            const int row = row0 + r < B ? row0 + r : B - 1;
            zr[u] = (idx < RT * kp0 && k >= ny && k < nin) ? a.inp_all[((size_t)step * B + row) * nin + k] : 0.f;
        }
    };
    fetch_z(0);
    cluster_barrier(cnt, target += a.G);
    const bool xl = xcd_agreed(cnt, a.xcd_local, g == 0);
    const int grow = row0 + c16 < B ? row0 + c16 : B - 1;
    const size_t hs = (size_t)a.S * B * nh;

    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool dbg = a.dbg != nullptr && cl == 0 && g == 0;
    auto tick = [&](int k) {
        if (dbg) { const unsigned long long t = __builtin_amdgcn_s_memrealtime(); tacc[k] += t - tprev; tprev = t; }
    };
    unsigned long long clk0 = 0, rt0 = 0;
    if (dbg) { tprev = rt0 = __builtin_amdgcn_s_memrealtime(); clk0 = __builtin_readcyclecounter(); }
    for (int i = 0; i < a.S; ++i) {
#pragma unroll
        for (int u = 0; u < KP0_MAX * RT / 256; ++u) {
            const int idx = tid + 256 * u, r = idx / kp0, k = idx - r * kp0;
            if (idx < RT * kp0) Is[r * (kp0 + IPAD) + k] = k < ny ? Ys[r * ny + k] : zr[u];
        }
        if (i + 1 < a.S) fetch_z(i + 1);
        __syncthreads();
        // ---- layer 0: A from LDS, B from registers, this wave's k blocks
        f32x4v acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float bl = a.b[0][colbase + ec];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int kb = 16 * (w + 4 * jj);
            const bool ok = kb < kp0;
            f32x4v x = *reinterpret_cast<const f32x4v*>(Is + c16 * (kp0 + IPAD) + (ok ? kb : 0) + 4 * q);
            if (!ok) x = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], w0[jj][0][e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], w0[jj][1][e], acc[1], 0, 0, 0);
            }
        }
        tick(0);
        for (int l = 0; l <= nfull; ++l) {
            red_store(Red, acc, w, lane);
            __syncthreads();
            float* hdst = a.hid + (size_t)l * hs + (size_t)i * B * nh;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = er + 8 * u;
                float v = red_sum(Red, r, ec) + bl;
                v = v > 0.f ? v : 0.f;
                if (row0 + r < B) st_x(hdst + (size_t)(row0 + r) * nh + colbase + ec, v, xl);
                if (l == nfull) Hs[r * 33 + ec] = v;
            }
            if (l == nfull) break;
            bl = a.b[l + 1][colbase + ec];
            cluster_barrier(cnt, target += a.G);
            tick(l == 0 ? 1 : 3);
            acc[0] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (xl) gemm_ks<true>(acc, hdst + (size_t)grow * nh, Wl + (size_t)l * nh * CW, nh, w, q, c16);
            else gemm_ks<false>(acc, hdst + (size_t)grow * nh, Wl + (size_t)l * nh * CW, nh, w, q, c16);
            tick(l == 0 ? 2 : 4);
        }
        __syncthreads();                                  // Hs complete
        float* pdst = part + ((size_t)i * a.G + g) * RT * KP0_MAX;
        if (has_lt) {
            f32x4v o = {0.f, 0.f, 0.f, 0.f};
            const float* hr = Hs + c16 * 33 + q;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) o = __builtin_amdgcn_mfma_f32_16x16x4f32(hr[4 * jj], wl[jj], o, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) st_x(pdst + (4 * q + e) * NYP_MAX + 16 * w + c16, o[e], xl);
        }
        tick(5);
        cluster_barrier(cnt, target += a.G);
        tick(6);
        const float* psrc = part + (size_t)i * a.G * RT * KP0_MAX;
        const int nq = a.nyp / 4;                          // 16 x nq <= 256 items: one per thread
        {
            const int it = tid < RT * nq ? tid : RT * nq - 1;
            const int r = it / nq, c0 = 4 * (it % nq);
            f32x4v sv = {0.f, 0.f, 0.f, 0.f};
            sum_slabs1<GP>(sv, psrc + r * NYP_MAX + c0, a.G, (size_t)RT * KP0_MAX);
            if (tid < RT * nq) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = c0 + e;
                    if (c >= ny) break;
                    const float rs = a.dt * (sv[e] + Bl[c]);
                    const float yn = Ys[r * ny + c] + rs;
                    Ys[r * ny + c] = yn;
                    if (g == 0 && row0 + r < B) {
                        const size_t o = (size_t)(row0 + r) * ny + c;
                        a.res[(size_t)i * B * ny + o] = rs;
                        a.y_all[(size_t)(i + 1) * B * ny + o] = yn;
                        if (i + 1 < a.S) a.inp_all[((size_t)(i + 1) * B + row0 + r) * nin + c] = yn;
                    }
                }
            }
        }
        __syncthreads();
        tick(7);
    }
    if (dbg && tid == 0) {
        for (int k = 0; k < 8; ++k) a.dbg[k] = tacc[k];
        a.dbg[8] = __builtin_readcyclecounter() - clk0;
        a.dbg[9] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

template <int GP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rollout_ks_bwd_kernel(const RollF a) {
    extern __shared__ float lds[];
    int cl, g;
    if (!locate(a, cl, g)) return;
    constexpr int RT = RT16;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, c16 = lane & 15;
    const int ec = tid & 31, er = tid >> 5;
    const int row0 = (a.tile0 + cl) * RT;
    const int colbase = g * CW;
    const int nl = a.nl, nh = a.nh, ny = a.ny, nin = a.nin, B = a.B, dwd = a.dwd, nyp = a.nyp, kp0 = a.kp0;
    const int nfull = nl - 2;
    // LDS: [nfull][nh][32] slices (B[k][c] = W_l[k][colbase + c]) | Do [16][nyp + IPAD] | Dy [16][ny] | Hs [16][33] | Red [4][2][256]
    float* Wl = lds;
    float* Do = Wl + (size_t)nfull * nh * CW;
    float* Dy = Do + RT * (nyp + IPAD);
    float* Hs = Dy + RT * ny;
    float* Red = Hs + RT * 33;
    unsigned* cnt = a.cnt + (size_t)(a.tile0 + cl) * 64;
    float* part = a.part + (size_t)(a.tile0 + cl) * a.S * a.G * RT * KP0_MAX;
    unsigned target = 0;
    xcd_announce(cnt);
    for (int l = 1; l <= nfull; ++l) {
        const float* W = a.W[l];
        float* dst = Wl + (size_t)(l - 1) * nh * CW;
        for (int idx = tid; idx < nh * CW; idx += 256) {
            const int k = idx / CW, c = idx - k * CW;
            dst[sw(k, c)] = W[(size_t)k * nh + colbase + c];
        }
    }
    // last layer backward (K = nyp <= 64: k block w of this wave): B[k = 16 w + 4 q + e][16 ct + c16] = W_{nl-1}[k][colbase + 16 ct + c16]
    float wlb[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 16 * w + 4 * q + e;
            wlb[ct][e] = k < ny ? a.W[nl - 1][(size_t)k * nh + colbase + 16 * ct + c16] : 0.f;
        }
    // input gradient (split-K over the cluster): this wave's column tiles ct = w + 4 u of the [kp0] outputs: B[k = 4 jj + q][col] = W_0[colbase + k][col]
    const int nct = kp0 / 16;
    float w0b[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ct = w + 4 * u, col = 16 * ct + c16;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
            w0b[u][jj] = (ct < nct && col < nin) ? a.W[0][(size_t)(colbase + 4 * jj + q) * nin + col] : 0.f;
    }
    for (int idx = tid; idx < RT * ny; idx += 256) {
        const int r = idx / ny, c = idx - r * ny;
        const int row = row0 + r < B ? row0 + r : B - 1;
        Dy[idx] = a.d_y_all[((size_t)a.S * B + row) * ny + c];
    }
    cluster_barrier(cnt, target += a.G);
    const bool xl = xcd_agreed(cnt, a.xcd_local, g == 0);
    const int grow = row0 + c16 < B ? row0 + c16 : B - 1;
    const size_t hs = (size_t)a.S * B * nh;
    const size_t ds = (size_t)a.S * B * dwd;

    float dres[NYP_MAX * RT / 256];
    auto fetch_dres = [&](int step) {
#pragma unroll
        for (int u = 0; u < NYP_MAX * RT / 256; ++u) {
            const int idx = tid + 256 * u, r = idx / nyp, c = idx - r * nyp;
            const int row = row0 + r < B ? row0 + r : B - 1;
            dres[u] = (a.d_res && idx < RT * nyp && c < ny) ? a.d_res[((size_t)step * B + row) * ny + c] : 0.f;
        }
    };
    float hm[2];                                          // saved activations (ReLU masks) of this thread's two outputs
    auto fetch_mask = [&](int l, int step) {
        const float* hsrc = a.hid + (size_t)l * hs + (size_t)step * B * nh;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = er + 8 * u;
            const int row = row0 + r < B ? row0 + r : B - 1;
            hm[u] = hsrc[(size_t)row * nh + colbase + ec];
        }
    };
    fetch_dres(a.S - 1);
    for (int i = a.S - 1; i >= 0; --i) {
        fetch_mask(nl - 2, i);
#pragma unroll
        for (int u = 0; u < NYP_MAX * RT / 256; ++u) {
            const int idx = tid + 256 * u, r = idx / nyp, c = idx - r * nyp;
            if (idx >= RT * nyp) break;
            float v = 0.f;
            if (c < ny) {
                v = a.dt * (dres[u] + Dy[r * ny + c]);
                if (g == 0 && row0 + r < B) a.dhid[(size_t)(nl - 1) * ds + ((size_t)i * B + row0 + r) * dwd + c] = v;
            }
            Do[r * (nyp + IPAD) + c] = v;
        }
        if (i > 0) fetch_dres(i - 1);
        __syncthreads();
        // ---- delta_{nl-2} slice: A = dout (LDS), B = registers, k block w
        f32x4v acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        {
            const bool ok = 16 * w < nyp;
            f32x4v x = *reinterpret_cast<const f32x4v*>(Do + c16 * (nyp + IPAD) + (ok ? 16 * w : 0) + 4 * q);
            if (!ok) x = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], wlb[0][e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], wlb[1][e], acc[1], 0, 0, 0);
            }
        }
        for (int l = nl - 2; l >= 0; --l) {
            red_store(Red, acc, w, lane);
            __syncthreads();
            float* ddst = a.dhid + (size_t)l * ds + (size_t)i * B * dwd;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = er + 8 * u;
                const float s = red_sum(Red, r, ec);
                const float v = hm[u] > 0.f ? s : 0.f;
                if (row0 + r < B) st_x(ddst + (size_t)(row0 + r) * dwd + colbase + ec, v, xl);
                if (l == 0) Hs[r * 33 + ec] = v;
            }
            if (l == 0) break;
            fetch_mask(l - 1, i);
            cluster_barrier(cnt, target += a.G);
            acc[0] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (xl) gemm_ks<true>(acc, ddst + (size_t)grow * dwd, Wl + (size_t)(l - 1) * nh * CW, nh, w, q, c16);
            else gemm_ks<false>(acc, ddst + (size_t)grow * dwd, Wl + (size_t)(l - 1) * nh * CW, nh, w, q, c16);
        }
        __syncthreads();
        // ---- dinp partial (split-K over the cluster)
        float* pdst = part + ((size_t)i * a.G + g) * RT * KP0_MAX;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ct = w + 4 * u;
            if (ct >= nct) break;
            f32x4v o = {0.f, 0.f, 0.f, 0.f};
            const float* hr = Hs + c16 * 33 + q;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) o = __builtin_amdgcn_mfma_f32_16x16x4f32(hr[4 * jj], w0b[u][jj], o, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) st_x(pdst + (4 * q + e) * KP0_MAX + 16 * ct + c16, o[e], xl);
        }
        const int nq = kp0 / 4;                            // 16 x nq <= 512 items: two per thread
        const int n_it = RT * nq;
        const int it = tid < n_it ? tid : n_it - 1, itb = tid + 256 < n_it ? tid + 256 : it;
        float dyv[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int item = h ? itb : it, r = item / nq, c0 = 4 * (item % nq);
            const int row = row0 + r < B ? row0 + r : B - 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) dyv[h][e] = c0 + e < ny ? a.d_y_all[((size_t)i * B + row) * ny + c0 + e] : 0.f;
        }
        cluster_barrier(cnt, target += a.G);
        const float* psrc = part + (size_t)i * a.G * RT * KP0_MAX;
        {
            f32x4v sv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            sum_slabs2<GP>(sv[0], sv[1], psrc + (it / nq) * KP0_MAX + 4 * (it % nq), psrc + (itb / nq) * KP0_MAX + 4 * (itb % nq), a.G,
                           (size_t)RT * KP0_MAX);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h ? (tid + 256 >= n_it) : (tid >= n_it)) break;
                const int item = h ? itb : it;
                const int r = item / nq, c0 = 4 * (item % nq);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = c0 + e;
                    if (c >= nin) break;
                    const float sum = sv[h][e];
                    if (g == 0 && row0 + r < B) a.dinp_all[((size_t)i * B + row0 + r) * nin + c] = sum;
                    if (c < ny) Dy[r * ny + c] = dyv[h][e] + Dy[r * ny + c] + sum;
                }
            }
        }
        __syncthreads();
    }
    if (g == 0)
        for (int idx = tid; idx < RT * ny; idx += 256) {
            const int r = idx / ny, c = idx - r * ny;
            if (row0 + r < B) a.d_y0[(size_t)(row0 + r) * ny + c] = Dy[idx];
        }
}

// ------------------------------------------------------------------------------------------------------------ generation chain
// The INFERENCE rollout (reference module/srvp.py:377-405 with nt > the number of observed frames; test.py:237-246, train.evaluate) as ONE
// persistent launch per group of row tiles: per frame the prior MLP p_z(y) (srvp.py:383), the sample z ~ posterior while data lasts / prior
// afterwards (srvp.py:385-391), then the n_euler residual steps with that z (srvp.py:394-400) -- as launches: (nl + 2) + n_euler (nl + 2)
// dependent micro-kernels per frame (579 GEMM launches for the 53-frame horizon of config 5).  Same decomposition as the training kernels
// (32-row tiles, a cluster of nh / 32 workgroups per tile, workgroup g owns hidden columns [32 g, 32 g + 32) of every layer); the weight
// placement is described at rollout_gen_ks_kernel.  Nothing is saved for a backward pass: the exchanged activation tiles live
// in one small per-tile buffer per layer that is rewritten every step (a cluster barrier separates any write from the previous reads), which
// is only safe when every member of the cluster shares one L2 -- the XCD-local exchange, VERIFIED per launch; a launch that finds its
// cluster spread over several XCCs writes nothing and counts a cluster failure (srvp_cluster_stats_read word 0; the host refuses to go on).
// The launcher's eligibility test probes the placement once per process, so such a launch is not expected to happen.
struct GenF {
    int B, ny, nz, nh, nl, S, ne, G, nin, kp0, nyp, nzp2, F, n_data, ntiles, tile0, cl_per_xcd, xcd_local;
    float dt;
    const float* W[MAX_NL]; const float* b[MAX_NL];       // dynamics
    const float* PW[MAX_NL]; const float* Pb[MAX_NL];     // p_z
    const float* y0; const float* qz; const float* eps;
    float* y_all; float* res; float* z; float* pz;
    float* hbuf; float* part; unsigned* cnt;
};

__device__ __forceinline__ float softplus_g(float x) { return x > 20.f ? x : log1pf(__expf(x)); }

// K-split GEMM of the generation chain's second form: acc[mt][ct] += A(16 MT rows of the exchange buffer, this wave's eight 16-wide k blocks:
// L1-bypassing loads) x B, B either register-resident (GLOB = false: bw) or fetched with the A fragments from two weight rows in global memory
// (GLOB = true: b0 / b1).  DEPTH k blocks in flight: the fragments of block jj + DEPTH are requested into the registers block jj has just been
// consumed from.  nh == 512: every k block exists and block jj is an IMMEDIATE offset on MT (+ 2) base pointers -- as computed addresses hipcc
// hoisted one 64-bit pointer per unrolled load out of the step loop, 170 registers.
template <int JJ, int MT, bool GLOB>
__device__ __forceinline__ void gk_load(f32x4v (&av)[8][MT], f32x4v (&bv)[8][2], const float* const (&pm)[MT], const float* b0, const float* b1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=&v"(av[JJ][mt]) : "v"(pm[mt]), "n"(256 * JJ) : "memory");
    if constexpr (GLOB) {
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(bv[JJ][0]) : "v"(b0), "n"(256 * JJ) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(bv[JJ][1]) : "v"(b1), "n"(256 * JJ) : "memory");
    }
}
template <int JJ, int MT, bool GLOB, int DEPTH>
__device__ __forceinline__ void gk_step(f32x4v (&acc)[MT][2], f32x4v (&av)[8][MT], f32x4v (&bv)[8][2], const float* const (&pm)[MT], const float* b0,
                                        const float* b1, const float (&bw)[8][2][4]) {
    constexpr int LPB = MT + (GLOB ? 2 : 0);
    constexpr int left = (7 - JJ < DEPTH - 1 ? 7 - JJ : DEPTH - 1) * LPB;
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(left) : "memory");
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(av[JJ][mt]));
    if constexpr (GLOB) { asm volatile("" : "+v"(bv[JJ][0])); asm volatile("" : "+v"(bv[JJ][1])); }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[JJ][mt][e], GLOB ? bv[JJ][0][e] : bw[JJ][0][e], acc[mt][0], 0, 0, 0);
            acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[JJ][mt][e], GLOB ? bv[JJ][1][e] : bw[JJ][1][e], acc[mt][1], 0, 0, 0);
        }
    if constexpr (JJ + DEPTH < 8) gk_load<JJ + DEPTH, MT, GLOB>(av, bv, pm, b0, b1);
    if constexpr (JJ < 7) gk_step<JJ + 1, MT, GLOB, DEPTH>(acc, av, bv, pm, b0, b1, bw);
}
template <int MT, bool GLOB>
__device__ __forceinline__ void gk_gemm(f32x4v (&acc)[MT][2], const float* const (&pm)[MT], const float* b0, const float* b1, const float (&bw)[8][2][4]) {
    constexpr int DEPTH = GLOB ? (MT <= 2 ? 4 : 2) : (MT <= 2 ? 4 : 3);
    f32x4v av[8][MT], bv[8][2];
    gk_load<0, MT, GLOB>(av, bv, pm, b0, b1);
    gk_load<1, MT, GLOB>(av, bv, pm, b0, b1);
    if constexpr (DEPTH > 2) gk_load<2, MT, GLOB>(av, bv, pm, b0, b1);
    if constexpr (DEPTH > 3) gk_load<3, MT, GLOB>(av, bv, pm, b0, b1);
    gk_step<0, MT, GLOB, DEPTH>(acc, av, bv, pm, b0, b1, bw);
}

// ---- the generation kernel.  The K loop of every hidden layer is dealt to the
// four waves as in rollout_ks_* (wave w owns the w-th 16-wide k block of every 64), which has a consequence the output-tiled form cannot have:
// a wave only ever touches ITS quarter of each weight slice, 64 values per lane and layer -- so the slices of the dynamics' two hidden layers
// (used n_euler times per frame) live in REGISTERS for the whole launch, 128 VGPRs per lane, and the LDS holds only the staging tiles.  (All four
// hidden layers in registers -- 256 VGPRs -- was built first: hipcc spilled 300 of them and the chain got slower, 91.1 vs 90.2 ms.)  The prior's
// two hidden layers (once per frame) take their B fragments from global memory (L2-resident).  32-row tiles (800 rows = 25 tiles must fit two co-resident launches): every A fragment feeds four MFMA
// tiles (2 row tiles x 2 column tiles), 16 loads per lane and layer instead of 32 + 32.
template <int GP, int MT>       // MT: 16-row tiles per cluster tile (2: 32 rows, the one instantiated; 4 was measured: see the launcher)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rollout_gen_ks_kernel(const GenF a) {
    extern __shared__ float lds[];
    constexpr int RTG = 16 * MT;
    const int xq = blockIdx.x & 7, kblk = blockIdx.x >> 3;
    const int cl = xq * a.cl_per_xcd + kblk / a.G, g = kblk % a.G;
    if (cl >= a.ntiles) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane >> 4, c16 = lane & 15;
    const int ec = tid & 31, er = tid >> 5;               // epilogue: outputs (er + 8 u, ec), u < 2 MT, of the (16 MT) x 32 tile
    const int row0 = (a.tile0 + cl) * RTG;
    const int colbase = g * CW;
    const int nl = a.nl, nh = a.nh, ny = a.ny, nz = a.nz, nin = a.nin, B = a.B, kp0 = a.kp0;
    const int nfull = nl - 2;                             // 1 or 2 (launcher)
    const int ils = kp0 + IPAD;
    // LDS: Is [32][kp0 + IPAD] | Ys [32][ny] | Zs [32][nz] | Hs [32][33] | Bl [ny] | Bp [2 nz] (padded to 16 bytes) | Red [4][MT][2][256]
    float* Is = lds;
    float* Ys = Is + RTG * ils;
    float* Zs = Ys + RTG * ny;
    float* Hs = Zs + RTG * nz;
    float* Bl = Hs + RTG * 33;
    float* Bp = Bl + NYP_MAX;
    float* Red = Bp + 2 * NYP_MAX;
    unsigned* cnt = a.cnt + (size_t)(a.tile0 + cl) * 64;
    float* hbuf = a.hbuf + (size_t)(a.tile0 + cl) * nfull * RTG * nh;
    float* part = a.part + (size_t)(a.tile0 + cl) * a.G * RTG * KP0_MAX;
    unsigned target = 0;
    xcd_announce(cnt);
    // hidden layers: B[k = 64 jj + 16 w + 4 q + e][16 ct + c16] = W_l[colbase + 16 ct + c16][k], this wave's k blocks only
    float dw[2][8][2][4];
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 64 * jj + 16 * w + 4 * q + e;
                    const bool ok = l < nfull && k < nh;
                    dw[l][jj][ct][e] = ok ? a.W[l + 1][(size_t)(colbase + 16 * ct + c16) * nh + k] : 0.f;
                }
    // first layers: dynamics K = kp0 (k blocks w and w + 4), prior K = nyp (k block w)
    float w0[2][2][4], pw0[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int k = 16 * (w + 4 * jj) + 4 * q + e;
                w0[jj][ct][e] = k < nin ? a.W[0][(size_t)(colbase + 16 * ct + c16) * nin + k] : 0.f;
            }
            const int k = 16 * w + 4 * q + e;
            pw0[ct][e] = k < ny ? a.PW[0][(size_t)(colbase + 16 * ct + c16) * ny + k] : 0.f;
        }
    // last layers (split-K over the cluster): dynamics: output column tile w; prior: column tiles w and w + 4; both row tiles each
    float wl[8], pwl[2][8];
    const int nctl = a.nyp / 16, nctp = a.nzp2 / 16;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int col = 16 * w + c16;
        wl[jj] = (w < nctl && col < ny) ? a.W[nl - 1][(size_t)col * nh + colbase + 4 * jj + q] : 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ct = w + 4 * u, colp = 16 * ct + c16;
            pwl[u][jj] = (ct < nctp && colp < 2 * nz) ? a.PW[nl - 1][(size_t)colp * nh + colbase + 4 * jj + q] : 0.f;
        }
    }
    for (int idx = tid; idx < RTG * ny; idx += 256) {
        const int r = idx / ny, c = idx - r * ny;
        const int row = row0 + r < B ? row0 + r : B - 1;
        Ys[idx] = a.y0[(size_t)row * ny + c];
    }
    for (int idx = tid; idx < RTG * ils; idx += 256) Is[idx] = 0.f;
    if (tid < ny) Bl[tid] = a.b[nl - 1][tid];
    if (tid < 2 * nz) Bp[tid] = a.Pb[nl - 1][tid];
    cluster_barrier(cnt, target += a.G);
    if (!xcd_agreed(cnt, a.xcd_local, g == 0)) {
        if (g == 0 && tid == 0) __hip_atomic_fetch_add(&g_cluster_timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const size_t hls = (size_t)RTG * nh;
    const int nq = a.nyp / 4, nq2 = a.nzp2 / 4;
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};

    auto gemm_rr = [&](f32x4v (&acc)[MT][2], const float* abuf, const float (&bw)[8][2][4]) {
        const float* pm[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) pm[mt] = abuf + (size_t)(16 * mt + c16) * nh + 16 * w + 4 * q;
        gk_gemm<MT, false>(acc, pm, nullptr, nullptr, bw);
    };
    auto gemm_rg = [&](f32x4v (&acc)[MT][2], const float* abuf, const float* W) {
        const float* pm[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) pm[mt] = abuf + (size_t)(16 * mt + c16) * nh + 16 * w + 4 * q;
        const float* b0 = W + (size_t)(colbase + c16) * nh + 16 * w + 4 * q;
        gk_gemm<MT, true>(acc, pm, b0, b0 + (size_t)16 * nh, dw[0]);
    };
    auto red_store4 = [&](const f32x4v (&acc)[MT][2]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) *reinterpret_cast<f32x4v*>(Red + ((w * MT + mt) * 2 + ct) * 256 + lane * 4) = acc[mt][ct];
    };
    auto red_sum4 = [&](int r, int c) {
        constexpr int WS = MT * 512;                      // floats per wave
        const int off = (((r >> 4) * 2) + (c >> 4)) * 256 + (((r & 15) >> 2) * 16 + (c & 15)) * 4 + (r & 3);
        return ((Red[off] + Red[WS + off]) + Red[2 * WS + off]) + Red[3 * WS + off];
    };
    // one MLP trunk up to the last hidden activation in Hs: first-layer accumulators in, hidden layers from registers, exchange through hbuf
    auto trunk = [&](f32x4v (&acc)[MT][2], const float* const* bias, const float* const* Wg) {     // Wg: the prior's weights (global) or null: dynamics (registers)
        float bl = bias[0][colbase + ec];
#pragma unroll
        for (int l = 0; l < 3; ++l) {                       // (unrolled: the register-resident slices need compile-time indices)
            red_store4(acc);
            __syncthreads();
            float* hdst = hbuf + (size_t)l * hls;
#pragma unroll
            for (int u = 0; u < 2 * MT; ++u) {
                const int r = er + 8 * u;
                float v = red_sum4(r, ec) + bl;
                v = v > 0.f ? v : 0.f;
                if (l < nfull) hdst[(size_t)r * nh + colbase + ec] = v;
                else Hs[r * 33 + ec] = v;
            }
            if (l == nfull) break;
            bl = bias[l + 1][colbase + ec];
            cluster_barrier(cnt, target += a.G);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = zero4; acc[mt][1] = zero4; }
            if (Wg) gemm_rg(acc, hdst, Wg[l + 1]);
            else gemm_rr(acc, hdst, dw[l < 2 ? l : 1]);
        }
        __syncthreads();                                  // Hs complete
    };

    for (int i = 0; i < a.S; ++i) {
        const int f = i / a.ne;
        if (i % a.ne == 0) {
            // =============================== frame start: p_z(y), then z
            for (int idx = tid; idx < RTG * ny; idx += 256) {
                const int r = idx / ny, c = idx - r * ny;
                Is[r * ils + c] = Ys[idx];
            }
            __syncthreads();
            f32x4v acc[MT][2];
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = zero4; acc[mt][1] = zero4; }
            {
                const bool ok = 16 * w < a.nyp;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4v x = *reinterpret_cast<const f32x4v*>(Is + (16 * mt + c16) * ils + (ok ? 16 * w : 0) + 4 * q);
                    if (!ok) x = zero4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], pw0[0][e], acc[mt][0], 0, 0, 0);
                        acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], pw0[1][e], acc[mt][1], 0, 0, 0);
                    }
                }
            }
            trunk(acc, a.Pb, a.PW);
            float* pdst = part + (size_t)g * RTG * KP0_MAX;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ct = w + 4 * u;
                if (ct >= nctp) break;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4v o = zero4;
                    const float* hr = Hs + (16 * mt + c16) * 33 + q;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) o = __builtin_amdgcn_mfma_f32_16x16x4f32(hr[4 * jj], pwl[u][jj], o, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) pdst[(16 * mt + 4 * q + e) * KP0_MAX + 16 * ct + c16] = o[e];
                }
            }
            const bool posterior = f + 1 < a.n_data;
            // (64-row tiles: no registers left for it -- the noise is fetched where it is used, one more latency per frame)
            constexpr bool PRE = MT <= 2;
            constexpr int NPRE = PRE ? NYP_MAX * RTG / 256 : 1;
            float epsr[NPRE], qlr[NPRE], qsr[NPRE];
            if constexpr (PRE) {
#pragma unroll
                for (int u = 0; u < NPRE; ++u) {
                    const int idx = tid + 256 * u, r = idx / nz, c = idx - r * nz;
                    const int row = row0 + r < B ? row0 + r : B - 1;
                    const bool ok = idx < RTG * nz;
                    epsr[u] = ok ? a.eps[((size_t)f * B + row) * nz + c] : 0.f;
                    qlr[u] = (ok && posterior) ? a.qz[((size_t)f * B + row) * 2 * nz + c] : 0.f;
                    qsr[u] = (ok && posterior) ? a.qz[((size_t)f * B + row) * 2 * nz + nz + c] : 0.f;
                }
            }
            cluster_barrier(cnt, target += a.G);
            for (int it = tid; it < RTG * nq2; it += 512) {
                const int itb = it + 256 < RTG * nq2 ? it + 256 : it;
                f32x4v sv[2] = {zero4, zero4};
                sum_slabs2h<GP>(sv[0], sv[1], part + (it / nq2) * KP0_MAX + 4 * (it % nq2), part + (itb / nq2) * KP0_MAX + 4 * (itb % nq2), a.G,
                               (size_t)RTG * KP0_MAX);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int item = h ? itb : it;
                    if (h && itb == it) break;
                    const int r = item / nq2, c0 = 4 * (item % nq2);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c0 + e < 2 * nz) Is[r * ils + c0 + e] = sv[h][e] + Bp[c0 + e];
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NYP_MAX * RTG / 256; ++u) {
                const int idx = tid + 256 * u, r = idx / nz, c = idx - r * nz;
                if (idx >= RTG * nz) break;
                const float pl = Is[r * ils + c], ps = Is[r * ils + nz + c];
                float ev, ql, qs;
                if constexpr (PRE) { ev = epsr[u]; ql = qlr[u]; qs = qsr[u]; }
                else {
                    const int row = row0 + r < B ? row0 + r : B - 1;
                    ev = a.eps[((size_t)f * B + row) * nz + c];
                    ql = posterior ? a.qz[((size_t)f * B + row) * 2 * nz + c] : 0.f;
                    qs = posterior ? a.qz[((size_t)f * B + row) * 2 * nz + nz + c] : 0.f;
                }
                const float loc = posterior ? ql : pl, raw = posterior ? qs : ps;
                const float zz = loc + ev * (softplus_g(raw) + 1e-8f);
                Zs[idx] = zz;
                if (g == 0 && row0 + r < B) {
                    const size_t o = (size_t)f * B + row0 + r;
                    a.z[o * nz + c] = zz;
                    a.pz[o * 2 * nz + c] = pl;
                    a.pz[o * 2 * nz + nz + c] = ps;
                }
            }
            __syncthreads();
            for (int idx = tid; idx < RTG * kp0; idx += 256) {
                const int r = idx / kp0, k = idx - r * kp0;
                Is[r * ils + k] = k < ny ? Ys[r * ny + k] : (k < nin ? Zs[r * nz + k - ny] : 0.f);
            }
        } else {
            for (int idx = tid; idx < RTG * ny; idx += 256) {
                const int r = idx / ny, c = idx - r * ny;
                Is[r * ils + c] = Ys[idx];
            }
        }
        __syncthreads();
        // =============================== one residual step
        f32x4v acc[MT][2];
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = zero4; acc[mt][1] = zero4; }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int kb = 16 * (w + 4 * jj);
            const bool ok = kb < kp0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4v x = *reinterpret_cast<const f32x4v*>(Is + (16 * mt + c16) * ils + (ok ? kb : 0) + 4 * q);
                if (!ok) x = zero4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], w0[jj][0][e], acc[mt][0], 0, 0, 0);
                    acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], w0[jj][1][e], acc[mt][1], 0, 0, 0);
                }
            }
        }
        trunk(acc, a.b, nullptr);
        float* pdst = part + (size_t)g * RTG * KP0_MAX;
        if (w < nctl) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4v o = zero4;
                const float* hr = Hs + (16 * mt + c16) * 33 + q;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) o = __builtin_amdgcn_mfma_f32_16x16x4f32(hr[4 * jj], wl[jj], o, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) pdst[(16 * mt + 4 * q + e) * KP0_MAX + 16 * w + c16] = o[e];
            }
        }
        cluster_barrier(cnt, target += a.G);
        for (int it = tid; it < RTG * nq; it += 512) {
            const int itb = it + 256 < RTG * nq ? it + 256 : it;
            f32x4v sv[2] = {zero4, zero4};
            sum_slabs2h<GP>(sv[0], sv[1], part + (it / nq) * KP0_MAX + 4 * (it % nq), part + (itb / nq) * KP0_MAX + 4 * (itb % nq), a.G,
                           (size_t)RTG * KP0_MAX);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int item = h ? itb : it;
                if (h && itb == it) break;
                const int r = item / nq, c0 = 4 * (item % nq);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = c0 + e;
                    if (c >= ny) break;
                    const float rs = a.dt * (sv[h][e] + Bl[c]);
                    const float yn = Ys[r * ny + c] + rs;
                    Ys[r * ny + c] = yn;
                    if (g == 0 && row0 + r < B) {
                        const size_t o = (size_t)(row0 + r) * ny + c;
                        a.res[(size_t)i * B * ny + o] = rs;
                        a.y_all[(size_t)(i + 1) * B * ny + o] = yn;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------ LSTM
// The posterior LSTM (nn.LSTM(nhx, nh, 1), reference module/srvp.py:132,366) as ONE persistent launch over its T steps (as launches: a GEMM
// + a cell kernel per step, 20 us of dependent latency each), forward and backward recurrence.
struct LstmF {
    int B, nh, T, G, ntiles, tile0, cl_per_xcd;
    const float* gx; const float* whh; float* h; float* c; float* ga; unsigned* cnt; int xcd_local;
};

__device__ __forceinline__ float sigmoid_l(float x) { return 1.f / (1.f + __expf(-x)); }

// 16-row tiles and 16 hidden units per workgroup (a cluster of nh / 16 workgroups per tile), v_mfma_f32_16x16x4_f32, the A tile read from LDS
// as 16-byte fragments:
//   forward : wave q = gate q of the workgroup's 16 units; h_{t-1} tile [16][nh] gathered into LDS; 64 MFMAs per wave and step.
//   backward: the contraction dh_carry[b][u] = sum_k dgates[b][k] W_hh[k][u] is split over the cluster along k INSTEAD of u: a workgroup
//     contracts the 64 gate gradients it has just formed itself (they never leave the CU) with its [64][nh] slice of W_hh (64 VGPRs per lane,
//     wave w = output units [w nh/4, (w+1) nh/4)) into a partial [16][nh] slab, and after ONE barrier gathers the G partial values of its own
//     16 x 16 units (16 KB written + 16 KB read per step instead of a 128 KB dgates tile), summed in a fixed order.
struct LstmB2 {
    int B, nh, T, G, ntiles, tile0, cl_per_xcd;
    const float* dh_out; const float* whh; const float* c; const float* ga; float* dgates; unsigned* cnt; int xcd_local; float* part;
};

template <int KST, int NCT>     // nh = 16 KST, 16 NCT hidden units per workgroup
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_ks_fwd_kernel(const LstmF a) {
    extern __shared__ float lds[];
    constexpr int NH = 16 * KST, HLD = NH + 4, CWL = 16 * NCT, GLD = CWL + 1, RTL = 16;
    float* Hs = lds;                          // [16][NH + 4]   h_{t-1} of this batch tile (rows 16-byte aligned)
    float* Gs = lds + RTL * HLD;              // [4][16][CWL + 1]  activated gates of this workgroup's units
    const int x = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int cl = x * a.cl_per_xcd + kk / a.G, g = kk % a.G;
    if (cl >= a.ntiles) return;
    const int tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = lane >> 4, c16 = lane & 15;
    const int row0 = (a.tile0 + cl) * RTL;
    float wreg[NCT][KST][4];                  // B[k = 16 j + 4 quad + e][16 ct + c16] = W_hh[gate row of that unit][k]
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int j = 0; j < KST; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                wreg[ct][j][e] = a.whh[(size_t)(q * NH + g * CWL + 16 * ct + c16) * NH + 16 * j + 4 * quad + e];
    float cst[NCT];
#pragma unroll
    for (int e = 0; e < NCT; ++e) cst[e] = 0.f;
    unsigned* cnt = a.cnt + (size_t)(a.tile0 + cl) * 64;
    const size_t gs = (size_t)a.B * 4 * NH, hs = (size_t)a.B * NH;
    xcd_announce(cnt);
    cluster_barrier(cnt, (unsigned)a.G);
    const bool xl = xcd_agreed(cnt, a.xcd_local, g == 0);
    for (int t = 0; t < a.T; ++t) {
        f32x4v acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = row0 + 4 * quad + e;
                acc[ct][e] = row < a.B ? a.gx[gs * t + (size_t)row * 4 * NH + q * NH + g * CWL + 16 * ct + c16] : 0.f;
            }
        if (t > 0) {
            cluster_barrier(cnt, (unsigned)((t + 1) * a.G));   // every member has stored its slice of h_{t-1}
            const float* hp = a.h + hs * (t - 1);
            constexpr int NP = RTL * NH / 4;                    // 16-byte pieces of the tile
            constexpr int NLD = (NP + 255) / 256;
            f32x4v hv[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int idx0 = tid + 256 * i, idx = idx0 < NP ? idx0 : NP - 1, row = idx / (NH / 4), c4 = idx % (NH / 4);
                const int gr = row0 + row < a.B ? row0 + row : a.B - 1;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(hv[i]) : "v"(hp + (size_t)gr * NH + c4 * 4) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NLD; ++i) asm volatile("" : "+v"(hv[i]));
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int idx = tid + 256 * i, row = idx / (NH / 4), c4 = idx % (NH / 4);
                if (idx < NP) *reinterpret_cast<f32x4v*>(Hs + row * HLD + c4 * 4) = hv[i];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < KST; ++j) {
                const f32x4v av = *reinterpret_cast<const f32x4v*>(Hs + c16 * HLD + 16 * j + 4 * quad);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], wreg[ct][j][e], acc[ct], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rl = 4 * quad + e;
                const float v = q == 2 ? tanhf(acc[ct][e]) : sigmoid_l(acc[ct][e]);
                if (row0 + rl < a.B) a.ga[gs * t + (size_t)(row0 + rl) * 4 * NH + q * NH + g * CWL + 16 * ct + c16] = v;
                Gs[(q * RTL + rl) * GLD + 16 * ct + c16] = v;
            }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NCT; ++e) {
            const int idx = tid + 256 * e, rl = idx / CWL, u = idx % CWL;
            const float ig = Gs[(0 * RTL + rl) * GLD + u], fg = Gs[(1 * RTL + rl) * GLD + u], gg = Gs[(2 * RTL + rl) * GLD + u], og = Gs[(3 * RTL + rl) * GLD + u];
            cst[e] = fg * cst[e] + ig * gg;
            if (row0 + rl < a.B) {
                const size_t o = hs * t + (size_t)(row0 + rl) * NH + g * CWL + u;
                a.c[o] = cst[e];
                st_x(a.h + o, og * tanhf(cst[e]), xl);
            }
        }
        __syncthreads();                                   // Gs is rewritten by the next step's gates
    }
}

template <int KST, int NCT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_ks_bwd_kernel(const LstmB2 a) {
    extern __shared__ float lds[];
    constexpr int NH = 16 * KST, CWL = 16 * NCT, KL = 4 * CWL, DLD = KL + 4, RTL = 16, G = NH / CWL, NCW = KST / 4;
    static_assert(KST % 4 == 0, "nh must be a multiple of 64");
    float* Dl = lds;                          // [16][4 CWL + 4]  the gate gradients this workgroup forms at step t (k_local = gate CWL + u)
    const int x = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int cl = x * a.cl_per_xcd + kk / G, g = kk % G;
    if (cl >= a.ntiles) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = lane >> 4, c16 = lane & 15;
    const int row0 = (a.tile0 + cl) * RTL;
    float wreg[NCW][KL / 16][4];              // B[k_local = 16 jj + 4 quad + e][col] = W_hh[gate NH + g CWL + u][col], col = w NH / 4 + 16 ct + c16
#pragma unroll
    for (int ct = 0; ct < NCW; ++ct)
#pragma unroll
        for (int jj = 0; jj < KL / 16; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kl = 16 * jj + 4 * quad + e, gate = kl / CWL, u = kl % CWL;
                wreg[ct][jj][e] = a.whh[(size_t)(gate * NH + g * CWL + u) * NH + w * (NH / 4) + 16 * ct + c16];
            }
    float dcs[NCT], dhc[NCT];
#pragma unroll
    for (int e = 0; e < NCT; ++e) { dcs[e] = 0.f; dhc[e] = 0.f; }
    unsigned* cnt = a.cnt + (size_t)(a.tile0 + cl) * 64;
    float* part = a.part + (size_t)(a.tile0 + cl) * a.T * G * RTL * NH;     // a fresh partial slab per step and member
    const size_t gs = (size_t)a.B * 4 * NH, hs = (size_t)a.B * NH;
    unsigned target = 0;
    xcd_announce(cnt);
    cluster_barrier(cnt, target += (unsigned)G);
    const bool xl = xcd_agreed(cnt, a.xcd_local, g == 0);
    // operands of the cell backward, fetched one step ahead (their latency otherwise sits on the chain)
    float p_ig[NCT], p_fg[NCT], p_gg[NCT], p_og[NCT], p_dh[NCT], p_c[NCT], p_cp[NCT];
    auto fetch = [&](int t, bool first) {
#pragma unroll
        for (int e = 0; e < NCT; ++e) {
            const int idx = tid + 256 * e, rl = idx / CWL, u = idx % CWL;
            const int row = row0 + rl < a.B ? row0 + rl : a.B - 1;
            const size_t ho = hs * t + (size_t)row * NH + g * CWL + u;
            const size_t go = gs * t + (size_t)row * 4 * NH + g * CWL + u;
            p_ig[e] = a.ga[go]; p_fg[e] = a.ga[go + NH]; p_gg[e] = a.ga[go + 2 * NH]; p_og[e] = a.ga[go + 3 * NH];
            p_dh[e] = a.dh_out[ho];
            p_c[e] = first ? a.c[ho] : p_cp[e];
            p_cp[e] = t > 0 ? a.c[ho - hs] : 0.f;
        }
    };
    fetch(a.T - 1, true);
    for (int t = a.T - 1; t >= 0; --t) {
#pragma unroll
        for (int e = 0; e < NCT; ++e) {
            const int idx = tid + 256 * e, rl = idx / CWL, u = idx % CWL;
            const bool valid = row0 + rl < a.B;
            const float ig = p_ig[e], fg = p_fg[e], gg = p_gg[e], og = p_og[e];
            const float dh = p_dh[e] + dhc[e];
            const float tc = tanhf(p_c[e]);
            const float dc = dcs[e] + dh * og * (1.f - tc * tc);
            const float d0 = dc * gg * ig * (1.f - ig), d1 = dc * p_cp[e] * fg * (1.f - fg), d2 = dc * ig * (1.f - gg * gg), d3 = dh * tc * og * (1.f - og);
            if (valid) {
                const size_t go = gs * t + (size_t)(row0 + rl) * 4 * NH + g * CWL + u;
                a.dgates[go] = d0; a.dgates[go + NH] = d1; a.dgates[go + 2 * NH] = d2; a.dgates[go + 3 * NH] = d3;
            }
            float* dl = Dl + rl * DLD + u;
            dl[0] = valid ? d0 : 0.f; dl[CWL] = valid ? d1 : 0.f; dl[2 * CWL] = valid ? d2 : 0.f; dl[3 * CWL] = valid ? d3 : 0.f;
            dcs[e] = dc * fg;
        }
        if (t == 0) break;
        fetch(t - 1, false);
        __syncthreads();
        f32x4v acc[NCW];
#pragma unroll
        for (int ct = 0; ct < NCW; ++ct) acc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < KL / 16; ++jj) {
            const f32x4v av = *reinterpret_cast<const f32x4v*>(Dl + c16 * DLD + 16 * jj + 4 * quad);
#pragma unroll
            for (int ct = 0; ct < NCW; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], wreg[ct][jj][e], acc[ct], 0, 0, 0);
        }
        float* slab = part + ((size_t)t * G + g) * RTL * NH;
#pragma unroll
        for (int ct = 0; ct < NCW; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) st_x(slab + (4 * quad + e) * NH + w * (NH / 4) + 16 * ct + c16, acc[ct][e], xl);
        cluster_barrier(cnt, target += (unsigned)G);          // every member's partial slab of step t is complete (and Dl has been read)
        const float* psrc = part + (size_t)t * G * RTL * NH;
#pragma unroll
        for (int e = 0; e < NCT; ++e) {
            const int idx = tid + 256 * e, rl = idx / CWL, u = idx % CWL;
            const float* pp = psrc + rl * NH + g * CWL + u;
            float v[G];
#pragma unroll
            for (int gg2 = 0; gg2 < G; ++gg2)
                asm volatile("global_load_dword %0, %1, off sc1" : "=&v"(v[gg2]) : "v"(pp + (size_t)gg2 * RTL * NH) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int gg2 = 0; gg2 < G; ++gg2) asm volatile("" : "+v"(v[gg2]));
            float s = v[0];
#pragma unroll
            for (int gg2 = 1; gg2 < G; ++gg2) s += v[gg2];
            dhc[e] = s;
        }
    }
}

// Placement probe (once per process): does block b of a plain launch run on XCC b % 8?  The generation chain needs it for correctness of its
// buffer reuse (and verifies it again inside every launch); the other persistent kernels only run faster when it holds.
__global__ void xcc_probe_kernel(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }
int g_placement = -1;        // -1 unknown, 0 no, 1 yes
bool placement_round_robin() {
    if (g_placement >= 0) return g_placement == 1;
    g_placement = 0;
    const int nb = 256;
    unsigned* d = nullptr;
    unsigned h[nb];
    if (hipMalloc(&d, nb * sizeof(unsigned)) != hipSuccess) return false;
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(nb), dim3(64), 0, 0, d);
    const bool ok = hipMemcpy(h, d, nb * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return false;
    unsigned seen = 0;
    for (int b = 0; b < 8; ++b) seen |= 1u << (h[b] & 15u);
    if (__builtin_popcount(seen) != 8) return false;      // (fewer than 8 XCCs visible: another partition mode)
    for (int b = 0; b < nb; ++b)
        if (h[b] != h[b & 7]) return false;
    g_placement = 1;
    return true;
}

int g_fused = -1, g_ncu = 0, g_xcd_local = -1;
int xcd_local_on() {
    if (g_xcd_local < 0) { const char* e = getenv("SRVP_CLUSTER_XCD_LOCAL"); g_xcd_local = e ? atoi(e) : 1; }
    return g_xcd_local;
}

int device_cus() {
    if (!g_ncu) {
        hipDeviceProp_t p; int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) p.multiProcessorCount = 256;
        g_ncu = p.multiProcessorCount;
    }
    return g_ncu;
}
// The persistent kernels spin on inter-workgroup counters: every workgroup of a launch must be co-resident.  The smallest launch
// is one cluster per XCD = 8 * G workgroups at one workgroup per CU (LDS), so a device (partition, CU mask) with fewer CUs than
// that -- or a process that shares the device with others running the same kernels -- must keep the per-layer launch sequence:
// SRVP_ROLLOUT_FUSED=0 / SRVP_LSTM_FUSED=0, or automatically when 8 * G exceeds the CU count.
bool clusters_fit(int G) { return 8 * G <= device_cus(); }

}  // namespace

// 0 if this chain cannot run fused (the caller keeps the per-layer launch sequence), else the workspace size in bytes
extern "C" int64_t srvp_rollout_fused_ws_bytes(const srvp_rollout_desc* d) {
    if (g_fused < 0) { const char* e = getenv("SRVP_ROLLOUT_FUSED"); g_fused = e ? atoi(e) : 1; }
    if (!g_fused || !d) return 0;
    const int nin = d->ny + d->nz;
    if (d->nl < 2 || d->nl > MAX_NL || d->nh % CW != 0 || d->nh / CW > 32 || nin > KP0_MAX || d->ny > NYP_MAX || d->ny > d->nh ||
        d->nsteps < 1) return 0;
    if (!d->pz_external || !d->hid_dyn) return 0;
    if (!clusters_fit(d->nh / CW)) return 0;
    const int kp0 = (nin + 15) / 16 * 16, nyp = (d->ny + 15) / 16 * 16;
    const size_t lds_f = ((size_t)(d->nl - 2) * d->nh * CW + RT16 * (kp0 + IPAD) + RT16 * d->ny + RT16 * 33 + NYP_MAX + 2048) * 4;
    const size_t lds_b = ((size_t)(d->nl - 2) * d->nh * CW + RT16 * (nyp + IPAD) + RT16 * d->ny + RT16 * 33 + 2048) * 4;
    if (lds_f > 160 * 1024 || lds_b > 160 * 1024) return 0;
    const int64_t tiles16 = (d->B + RT16 - 1) / RT16;
    // two counter blocks (forward / backward launch: the forward's preparation kernel clears both), then the split-K slabs
    return 2 * tiles16 * 256 + tiles16 * (int64_t)d->nsteps * (d->nh / CW) * RT16 * KP0_MAX * 4;
}

static int fused_common(const srvp_rollout_desc& f, RollF& k, void* ws, int rt) {
    device_cus();
    k.B = f.B; k.ny = f.ny; k.nz = f.nz; k.nh = f.nh; k.nl = f.nl; k.S = f.nsteps; k.ne = f.n_euler; k.G = f.nh / CW;
    k.nin = f.ny + f.nz; k.kp0 = (k.nin + 15) / 16 * 16; k.nyp = (f.ny + 15) / 16 * 16; k.dt = f.dt;
    for (int l = 0; l < MAX_NL; ++l) { k.W[l] = l < f.nl ? f.dyn_w[l] : nullptr; k.b[l] = l < f.nl ? f.dyn_b[l] : nullptr; }
    const int tiles = (f.B + rt - 1) / rt;
    k.cnt = (unsigned*)ws; k.xcd_local = xcd_local_on();                // (the backward launch takes the second block: cnt + tiles * 64)
    k.part = (float*)((char*)ws + (size_t)2 * tiles * 256);
    return tiles;
}

// the counter blocks of this chain's persistent launches (forward + backward) in 4-byte words, from the start of fused_ws
int64_t srvp_rollout_fused_cnt_words(const srvp_rollout_desc* d) {
    return (int64_t)2 * ((d->B + RT16 - 1) / RT16) * 64;
}
// the workspace whose backward counter block the last forward launch left cleared (host order; one entry: a training step is forward, backward)
static std::atomic<const void*> g_bwd_cnt_clean{nullptr};        // (host threads: a lost update only costs the memset)

// clusters per launch: all workgroups co-resident (grid <= CUs), whole clusters per XCD
static int clusters_per_launch(int G, int tiles, int& cl_per_xcd) {
    int per_xcd = (g_ncu / 8) / G;
    if (per_xcd < 1) per_xcd = 1;
    int need = (tiles + 7) / 8;
    cl_per_xcd = need < per_xcd ? need : per_xcd;
    return cl_per_xcd * 8;
}

int srvp_rollout_fused_fwd(const srvp_rollout_desc* d, hipStream_t st, bool counters_cleared) {
    RollF k{};
    const int tiles = fused_common(*d, k, d->fused_ws, RT16);
    k.y0 = d->y0; k.y_all = d->y_all; k.res = d->res; k.inp_all = d->inp_all; k.hid = d->hid_dyn;
    const size_t lds = ((size_t)(k.nl - 2) * k.nh * CW + RT16 * (k.kp0 + IPAD) + RT16 * k.ny + RT16 * 33 + NYP_MAX + 2048) * 4;
    auto kern = k.G <= 8 ? rollout_ks_fwd_kernel<8> : (k.G <= 16 ? rollout_ks_fwd_kernel<16> : rollout_ks_fwd_kernel<32>);
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SRVP_REQUIRE(e == hipSuccess, "srvp_rollout_fwd(fused): cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    if (!counters_cleared) {
        e = hipMemsetAsync(k.cnt, 0, (size_t)tiles * 256, st);
        SRVP_REQUIRE(e == hipSuccess, "srvp_rollout_fwd(fused): memset failed");
    }
    g_bwd_cnt_clean.store(counters_cleared ? d->fused_ws : nullptr);
    int cpx;
    const int per = clusters_per_launch(k.G, tiles, cpx);
    static int dbg_on = -1;
    static unsigned long long* dbg_buf = nullptr;
    if (dbg_on < 0) { const char* e2 = getenv("SRVP_RF_DEBUG"); dbg_on = e2 ? atoi(e2) : 0; }
    if (dbg_on && !dbg_buf) (void)hipMalloc(&dbg_buf, 10 * sizeof(unsigned long long));
    k.dbg = dbg_on ? dbg_buf : nullptr;
    for (int t0 = 0; t0 < tiles; t0 += per) {
        k.tile0 = t0; k.ntiles = tiles - t0 < per ? tiles - t0 : per; k.cl_per_xcd = cpx;
        hipLaunchKernelGGL(kern, dim3(8 * cpx * k.G), dim3(256), lds, st, k);
    }
    SRVP_CHECK_LAUNCH("srvp_rollout_fwd(fused)");
    if (dbg_on && dbg_buf) {
        // diagnostic only (synchronises): per-phase microseconds per Euler step of cluster 0 / member 0
        unsigned long long h[10];
        if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[SRVP_RF_DEBUG] rollout_fused_fwd B=%d S=%d us/step:", k.B, k.S);
            static const char* nm[8] = {"stage+L0", "epi+bar1", "gemm1", "epi+bar2", "gemm2", "lastL", "bar3", "sum+upd"};
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.2f", nm[i], h[i] * 0.01 / k.S);
            fprintf(stderr, "  | s_memtime / s_memrealtime = %.2f MHz", h[9] ? 100.0 * (double)h[8] / (double)h[9] : 0.0);
            fprintf(stderr, "\n");
        }
    }
    return SRVP_OK;
}

int srvp_rollout_fused_bwd(const srvp_rollout_bwd_desc* d, hipStream_t st) {
    RollF k{};
    const srvp_rollout_desc& f = d->f;
    const int tiles = fused_common(f, k, f.fused_ws, RT16);
    k.hid = f.hid_dyn; k.d_y_all = d->d_y_all; k.d_res = d->d_res; k.dhid = d->dhid_dyn; k.dinp_all = d->dinp_all; k.d_y0 = d->d_y0;
    k.dwd = f.nh > f.ny ? f.nh : f.ny;
    const size_t lds = ((size_t)(k.nl - 2) * k.nh * CW + RT16 * (k.nyp + IPAD) + RT16 * k.ny + RT16 * 33 + 2048) * 4;
    auto kern = k.G <= 8 ? rollout_ks_bwd_kernel<8> : (k.G <= 16 ? rollout_ks_bwd_kernel<16> : rollout_ks_bwd_kernel<32>);
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SRVP_REQUIRE(e == hipSuccess, "srvp_rollout_bwd(fused): cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    k.cnt += (size_t)tiles * 64;                            // the backward's own counter block
    if (g_bwd_cnt_clean.exchange(nullptr) != f.fused_ws) {
        e = hipMemsetAsync(k.cnt, 0, (size_t)tiles * 256, st);
        SRVP_REQUIRE(e == hipSuccess, "srvp_rollout_bwd(fused): memset failed");
    }
    int cpx;
    const int per = clusters_per_launch(k.G, tiles, cpx);
    for (int t0 = 0; t0 < tiles; t0 += per) {
        k.tile0 = t0; k.ntiles = tiles - t0 < per ? tiles - t0 : per; k.cl_per_xcd = cpx;
        hipLaunchKernelGGL(kern, dim3(8 * cpx * k.G), dim3(256), lds, st, k);
    }
    SRVP_CHECK_LAUNCH("srvp_rollout_bwd(fused)");
    return SRVP_OK;
}

// ---- generation chain (inference: p_z inside the loop).  0 = not eligible (the caller keeps the per-layer launch sequence), else the
// workspace size in bytes
extern "C" int64_t srvp_rollout_gen_ws_bytes(const srvp_rollout_desc* d) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SRVP_ROLLOUT_GEN_FUSED"); on = e ? atoi(e) : 1; }
    if (g_fused < 0) { const char* e = getenv("SRVP_ROLLOUT_FUSED"); g_fused = e ? atoi(e) : 1; }
    if (!on || !g_fused || !d || !xcd_local_on()) return 0;
    const int nin = d->ny + d->nz;
    if (d->nl < 3 || d->nl > MAX_NL || d->nh % CW != 0 || d->nh / CW > 32 || d->nh < 32 || nin > KP0_MAX || d->ny > NYP_MAX || d->nz > NYP_MAX || d->ny > d->nh ||
        2 * d->nz > KP0_MAX || d->nsteps < 1 || d->n_euler < 1 || d->pz_external) return 0;
    // the persistent form keeps the dynamics' hidden slices in registers: <= two hidden layers of 512 units per network (every recipe of the
    // reference: nh_res 512, nlayers_res 4).  Other widths take the per-layer launch sequence (round 6: the 32-row LDS-resident kernel of
    // rounds 3-4 that used to catch them is gone).
    if (d->nl - 2 > 2 || d->nh != 512) return 0;
    const int kp0 = (nin + 15) / 16 * 16, nzp2 = (2 * d->nz + 15) / 16 * 16;
    if (nzp2 > kp0) return 0;                             // the summed prior parameters are staged in the input tile
    if (!clusters_fit(d->nh / CW)) return 0;
    if (!placement_round_robin()) return 0;
    // (rows rounded up to 64: either tile height of the second form, rollout_gen_ks_kernel<., 2 | 4>)
    const int64_t rows = (d->B + 63) / 64 * 64, tiles = rows / RT;
    return tiles * 256 + rows * (int64_t)(d->nl - 2) * d->nh * 4 + rows * (int64_t)(d->nh / CW) * KP0_MAX * 4;
}

int srvp_rollout_gen_fwd(const srvp_rollout_desc* d, hipStream_t st) {
    device_cus();
    GenF k{};
    k.B = d->B; k.ny = d->ny; k.nz = d->nz; k.nh = d->nh; k.nl = d->nl; k.S = d->nsteps; k.ne = d->n_euler; k.G = d->nh / CW;
    k.nin = d->ny + d->nz; k.kp0 = (k.nin + 15) / 16 * 16; k.nyp = (d->ny + 15) / 16 * 16; k.nzp2 = (2 * d->nz + 15) / 16 * 16; k.dt = d->dt;
    k.F = (d->nsteps + d->n_euler - 1) / d->n_euler; k.n_data = d->n_data_frames; k.xcd_local = xcd_local_on();
    for (int l = 0; l < MAX_NL; ++l) {
        k.W[l] = l < d->nl ? d->dyn_w[l] : nullptr; k.b[l] = l < d->nl ? d->dyn_b[l] : nullptr;
        k.PW[l] = l < d->nl ? d->pz_w[l] : nullptr; k.Pb[l] = l < d->nl ? d->pz_b[l] : nullptr;
    }
    k.y0 = d->y0; k.qz = d->q_z_params; k.eps = d->eps_z; k.y_all = d->y_all; k.res = d->res; k.z = d->z; k.pz = d->p_z_params;
    SRVP_REQUIRE(k.n_data <= 1 || k.qz, "srvp_rollout_fwd(gen): posterior frames need q_z_params");
    // tile height: 32 rows.  (64-row tiles, rollout_gen_ks_kernel<., 4>, turn the two co-resident launches of the config-5 protocol's 800 rows
    // into one -- a launch lasts as long as its chain of steps whatever its tile count -- and were built and measured in round 5: hipcc spills
    // 226 registers in that instantiation and the call gets slower, 89.8 vs 89.2 ms; not instantiated)
    int cpx;
    const int rt = RT;
    const int tiles = (d->B + rt - 1) / rt;
    const int64_t rows64 = (d->B + 63) / 64 * 64;
    k.cnt = (unsigned*)d->fused_ws;
    k.hbuf = (float*)((char*)d->fused_ws + (size_t)(rows64 / RT) * 256);
    k.part = k.hbuf + (size_t)rows64 * (d->nl - 2) * d->nh;
    SRVP_REQUIRE(k.nl - 2 <= 2 && k.nh == 512, "srvp_rollout_fwd(gen): shape not eligible (srvp_rollout_gen_ws_bytes returned 0)");
    const size_t lds = ((size_t)rt * (k.kp0 + IPAD) + rt * k.ny + rt * k.nz + rt * 33 + 3 * NYP_MAX + 4 * (rt / 16) * 512) * 4;
    void (*kern)(const GenF) = rollout_gen_ks_kernel<16, 2>;        // (nh == 512: G = 16)
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SRVP_REQUIRE(e == hipSuccess, "srvp_rollout_fwd(gen): cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    e = hipMemsetAsync(k.cnt, 0, (size_t)tiles * 256, st);
    SRVP_REQUIRE(e == hipSuccess, "srvp_rollout_fwd(gen): memset failed");
    e = hipMemcpyAsync(d->y_all, d->y0, sizeof(float) * (size_t)d->B * d->ny, hipMemcpyDeviceToDevice, st);
    SRVP_REQUIRE(e == hipSuccess, "srvp_rollout_fwd(gen): copy failed");
    // (more tiles than one co-resident launch holds: the groups run one after the other on this stream.  Putting the second group on another
    // stream under the decoding of the first group's rows was measured in round 5 and is slower: its spinning workgroups hold 144 CUs' worth
    // of LDS while they become resident one by one, and the decoder's convolutions lose more than the 3.6 ms the overlap hides)
    const int per = clusters_per_launch(k.G, tiles, cpx);
    for (int t0 = 0; t0 < tiles; t0 += per) {
        k.tile0 = t0; k.ntiles = tiles - t0 < per ? tiles - t0 : per; k.cl_per_xcd = cpx;
        hipLaunchKernelGGL(kern, dim3(8 * cpx * k.G), dim3(256), lds, st, k);
    }
    SRVP_CHECK_LAUNCH("srvp_rollout_fwd(gen)");
    return SRVP_OK;
}

// ---- persistent LSTM forward: 0 = not eligible (the caller keeps srvp_lstm_fwd), else the workspace size in bytes
extern "C" int64_t srvp_lstm_fused_ws_bytes(int T, int B, int nh) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SRVP_LSTM_FUSED"); on = e ? atoi(e) : 1; }
    if (!on || T < 1 || B < 1 || !(nh == 64 || nh == 128 || nh == 256)) return 0;
    if (!clusters_fit(nh / CW)) return 0;
    const int64_t t16 = (B + 15) / 16;                    // (counter blocks + the backward's partial slabs)
    return t16 * 256 + t16 * (int64_t)T * (nh / 16) * 16 * nh * 4;
}
extern "C" int srvp_lstm_fwd_fused(const float* gates_x, const float* w_hh, float* h_out, float* c_out, float* gates_act, int T, int B,
                                   int nh, void* ws, int64_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SRVP_REQUIRE(gates_x && w_hh && h_out && c_out && gates_act && ws, "srvp_lstm_fwd_fused: null pointer");
    const int64_t need = srvp_lstm_fused_ws_bytes(T, B, nh);
    SRVP_REQUIRE(need > 0 && ws_bytes >= need, "srvp_lstm_fwd_fused: shape not eligible or workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)need);
    device_cus();
    LstmF k{};
    k.B = B; k.nh = nh; k.T = T; k.G = nh / 16; k.gx = gates_x; k.whh = w_hh; k.h = h_out; k.c = c_out; k.ga = gates_act; k.cnt = (unsigned*)ws; k.xcd_local = xcd_local_on();
    const int tiles = (B + 15) / 16;
    const size_t lds = ((size_t)16 * (nh + 4) + 4 * 16 * 17) * 4;
    auto kern = nh == 256 ? lstm_ks_fwd_kernel<16, 1> : (nh == 128 ? lstm_ks_fwd_kernel<8, 1> : lstm_ks_fwd_kernel<4, 1>);
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SRVP_REQUIRE(e == hipSuccess, "srvp_lstm_fwd_fused: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    e = hipMemsetAsync(k.cnt, 0, (size_t)tiles * 256, st);
    SRVP_REQUIRE(e == hipSuccess, "srvp_lstm_fwd_fused: memset failed");
    int cpx;
    const int per = clusters_per_launch(k.G, tiles, cpx);
    for (int t0 = 0; t0 < tiles; t0 += per) {
        k.tile0 = t0; k.ntiles = tiles - t0 < per ? tiles - t0 : per; k.cl_per_xcd = cpx;
        hipLaunchKernelGGL(kern, dim3(8 * cpx * k.G), dim3(256), lds, st, k);
    }
    SRVP_CHECK_LAUNCH("srvp_lstm_fwd_fused");
    return SRVP_OK;
}


// ---- persistent LSTM backward recurrence: same eligibility / workspace as the forward (srvp_lstm_fused_ws_bytes)
extern "C" int srvp_lstm_bwd_fused(const float* dh_out, const float* w_hh, const float* c_out, const float* gates_act, float* dgates, int T,
                                   int B, int nh, void* ws, int64_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SRVP_REQUIRE(dh_out && w_hh && c_out && gates_act && dgates && ws, "srvp_lstm_bwd_fused: null pointer");
    const int64_t need = srvp_lstm_fused_ws_bytes(T, B, nh);
    SRVP_REQUIRE(need > 0 && ws_bytes >= need, "srvp_lstm_bwd_fused: shape not eligible or workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)need);
    device_cus();
    LstmB2 k{};
    const int tiles = (B + 15) / 16;
    k.B = B; k.nh = nh; k.T = T; k.G = nh / 16; k.dh_out = dh_out; k.whh = w_hh; k.c = c_out; k.ga = gates_act; k.dgates = dgates;
    k.cnt = (unsigned*)ws; k.xcd_local = xcd_local_on(); k.part = (float*)((char*)ws + (size_t)tiles * 256);
    const size_t lds = (size_t)16 * (4 * 16 + 4) * 4;
    auto kern = nh == 256 ? lstm_ks_bwd_kernel<16, 1> : (nh == 128 ? lstm_ks_bwd_kernel<8, 1> : lstm_ks_bwd_kernel<4, 1>);
    hipError_t e = hipMemsetAsync(k.cnt, 0, (size_t)tiles * 256, st);
    SRVP_REQUIRE(e == hipSuccess, "srvp_lstm_bwd_fused: memset failed");
    int cpx;
    const int per = clusters_per_launch(k.G, tiles, cpx);
    // (more 16-row tiles than one co-resident launch holds -- B > 256 at nh = 256: the groups run one after the other on this stream)
    for (int t0 = 0; t0 < tiles; t0 += per) {
        k.tile0 = t0; k.ntiles = tiles - t0 < per ? tiles - t0 : per; k.cl_per_xcd = cpx;
        hipLaunchKernelGGL(kern, dim3(8 * cpx * k.G), dim3(256), lds, st, k);
    }
    SRVP_CHECK_LAUNCH("srvp_lstm_bwd_fused");
    return SRVP_OK;
}

// A/B switch of the XCD-local exchange of the persistent latent kernels (default 1 / env SRVP_CLUSTER_XCD_LOCAL): 0 = agent-scope write-through
// stores whatever the placement.  Same results bit for bit (the arithmetic and its order do not change).
extern "C" int srvp_cluster_set_xcd_local(int on) { g_xcd_local = on ? 1 : 0; return SRVP_OK; }
// host_words[0] = cluster-barrier timeouts, host_words[1] = clusters (summed over launches) that verified all their members on one XCC and
// exchanged through its L2, both since the library was loaded; copied to pinned host memory in stream order.
extern "C" int srvp_cluster_stats_read(unsigned* host_words2, void* stream) {
    SRVP_REQUIRE(host_words2, "srvp_cluster_stats_read: null pointer");
    hipError_t e = hipMemcpyFromSymbolAsync(host_words2, HIP_SYMBOL(g_cluster_timeouts), sizeof(unsigned), 0, hipMemcpyDeviceToHost, (hipStream_t)stream);
    SRVP_REQUIRE(e == hipSuccess, "srvp_cluster_stats_read: %s", hipGetErrorString(e));
    e = hipMemcpyFromSymbolAsync(host_words2 + 1, HIP_SYMBOL(g_cluster_xcd_local), sizeof(unsigned), 0, hipMemcpyDeviceToHost, (hipStream_t)stream);
    SRVP_REQUIRE(e == hipSuccess, "srvp_cluster_stats_read: %s", hipGetErrorString(e));
    return SRVP_OK;
}

// Number of cluster-barrier timeouts since the library was loaded (0 = every persistent latent kernel -- fused rollout forward / backward,
// LSTM forward / backward -- ran with its workgroups co-resident), copied to `host_word` (pinned host memory) in stream order.
extern "C" int srvp_cluster_timeouts_read(unsigned* host_word, void* stream) {
    SRVP_REQUIRE(host_word, "srvp_cluster_timeouts_read: null pointer");
    hipError_t e = hipMemcpyFromSymbolAsync(host_word, HIP_SYMBOL(g_cluster_timeouts), sizeof(unsigned), 0, hipMemcpyDeviceToHost, (hipStream_t)stream);
    SRVP_REQUIRE(e == hipSuccess, "srvp_cluster_timeouts_read: %s", hipGetErrorString(e));
    return SRVP_OK;
}
