// Weight packing (state-dict fp32 OIHW / IOHW <-> tap-major padded bf16), skip-gradient reduction, the ELBO terms
// and the fused Adam update.
//
// Replaces: the reference's torch.distributions / reductions in train.py:90-106 (NLL, KL(q(y0)||N(0,1)), KL(q(z)||p(z)),
// sum of residual norms) with their gradients, and torch.optim.Adam (train.py:289) as one elementwise launch over a
// flat parameter buffer.
#include "common.h"
#include "../../include/srvp_hip.h"

namespace {

struct PackArgs {
    int ntaps; int tap_off[SRVP_MAX_TAPS];
    int J, K;                 // padded sizes of the packed [t][J][K] tensor
    int J0, J0r, J1r;         // J axis = segment 0 (padded J0, real J0r) followed by segment 1 (real J1r)
    int K0, K0r, K1r;
    long long sj, sk;         // element strides of the real j / k index in the fp32 tensor
    int tap_set[SRVP_MAX_TAPS];   // != 0: packed tap = sum of the source taps in the bit set
    int layout;                   // 0 tap-major, 1 MFMA-fragment-major (see srvp_hip.h)
    int dst_f32;                  // packed tensor is fp32 (precision = 'fp32' parity mode; tap-major only)
    int kc_total, kc_off;         // layout 1: K chunks per tap of the destination (0 = K/64) and first chunk of this job
};

__host__ __device__ __forceinline__ int real_index(int i, int seg0_pad, int seg0_real, int seg1_real) {
    if (i < seg0_pad) return i < seg0_real ? i : -1;
    int i1 = i - seg0_pad;
    return i1 < seg1_real ? seg0_real + i1 : -1;
}

__device__ __forceinline__ void pack_one(const float* __restrict__ src, bf16_t* __restrict__ dst, const PackArgs& a, long long i) {
    int k = (int)(i % a.K); long long q = i / a.K;
    int j = (int)(q % a.J); int t = (int)(q / a.J);
    int jr = real_index(j, a.J0, a.J0r, a.J1r), kr = real_index(k, a.K0, a.K0r, a.K1r);
    float v = 0.f;
    if (jr >= 0 && kr >= 0) {
        int off = 0, set = 0;
#pragma unroll
        for (int u = 0; u < SRVP_MAX_TAPS; ++u) if (u == t) { off = a.tap_off[u]; set = a.tap_set[u]; }
        if (set == 0) v = src[jr * a.sj + kr * a.sk + off];
        else
            for (int sidx = 0; sidx < 16; ++sidx) if ((set >> sidx) & 1) v += src[jr * a.sj + kr * a.sk + sidx];   // fp32 sum, one rounding
    }
    long long o = i;
    if (a.layout == 1) {
        const int cc = k >> 6, kk = (k >> 4) & 3, kh = (k >> 3) & 1;
        const int KC = a.kc_total ? a.kc_total : (a.K >> 6);
        o = ((((long long)(t * KC + cc + a.kc_off) * 4 + kk) * (a.J >> 5) + (j >> 5)) * 64 + kh * 32 + (j & 31)) * 8 + (k & 7);
    }
    if (a.dst_f32) reinterpret_cast<float*>(dst)[o] = v;
    else dst[o] = f2bf(v);
}

__device__ __forceinline__ void unpack_one(const float* __restrict__ src, float* __restrict__ dst, const PackArgs& a, long long i) {
    int k = (int)(i % a.K); long long q = i / a.K;
    int j = (int)(q % a.J); int t = (int)(q / a.J);
    int jr = real_index(j, a.J0, a.J0r, a.J1r), kr = real_index(k, a.K0, a.K0r, a.K1r);
    if (jr < 0 || kr < 0) return;
    int off = 0, set = 0;
#pragma unroll
    for (int u = 0; u < SRVP_MAX_TAPS; ++u) if (u == t) { off = a.tap_off[u]; set = a.tap_set[u]; }
    if (set == 0) { dst[jr * a.sj + kr * a.sk + off] += src[i]; return; }   // every (t,j,k) maps to a distinct element: no atomics
    for (int sidx = 0; sidx < 16; ++sidx)
        if ((set >> sidx) & 1) atomicAdd(dst + jr * a.sj + kr * a.sk + sidx, src[i]);     // several packed taps share a source tap
}

__global__ void pack_weight_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, const PackArgs a) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long long)a.ntaps * a.J * a.K) pack_one(src, dst, a, i);
}
__global__ void unpack_wgrad_kernel(const float* __restrict__ src, float* __restrict__ dst, const PackArgs a) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long long)a.ntaps * a.J * a.K) unpack_one(src, dst, a, i);
}

// All layers in ONE launch: blockIdx.y = job (a device-resident table of srvp_pack_job), grid-stride over its elements.
// (40 + 30 tiny launches per step were ~1.2 ms of dispatch latency.)
__device__ __forceinline__ void job_args(const srvp_pack_job& j, PackArgs& a) {
    a.ntaps = j.d.ntaps;
#pragma unroll
    for (int t = 0; t < SRVP_MAX_TAPS; ++t) { a.tap_off[t] = j.d.tap_off[t]; a.tap_set[t] = j.d.tap_set[t]; }
    a.J = j.d.J; a.K = j.d.K; a.J0 = j.d.J0; a.J0r = j.d.J0r; a.J1r = j.d.J1r; a.K0 = j.d.K0; a.K0r = j.d.K0r; a.K1r = j.d.K1r;
    a.sj = j.d.sj; a.sk = j.d.sk; a.layout = j.d.layout; a.dst_f32 = j.d.dst_f32; a.kc_total = j.d.kc_total; a.kc_off = j.d.kc_off;
}
// Vector path (K % 8 == 0, source taps within 16 elements of the (j, k) base -- every conv / convT weight): one work item
// = (j, eight consecutive k), lanes along j.  The fp32 tensor keeps its taps innermost, so an item's reads / read-modify-writes
// are runs of `source taps` consecutive floats (re-touched over the tap loop: L1/L2 hits) instead of 4-byte accesses 36
// bytes apart, and the packed side moves as whole 16-byte (pack) / 32-byte (unpack) vectors.
// (dst_f32 describes the PACKED WEIGHT tensor of a pack job; an unpack job reads the fp32 gradient whatever it says: ignore_f32)
__host__ __device__ __forceinline__ bool vec_ok(const srvp_pack_desc& d, bool ignore_f32 = false) {
    if (d.K % 8 != 0 || (d.dst_f32 && !ignore_f32)) return false;
    for (int t = 0; t < d.ntaps; ++t)
        if (d.tap_set[t] == 0 ? (d.tap_off[t] < 0 || d.tap_off[t] >= 16) : (d.tap_set[t] >> 16) != 0) return false;
    return true;
}
// Workgroups of a job: one per 512 vector items, at most 256 -- the jobs of a network span three orders of magnitude in size,
// and a uniform 2-D grid (every job x the largest job's workgroups) spent most of the launch on workgroups with nothing to do.
__host__ __device__ inline int pack_job_wgs(long long total) {
    long long w = (total / 8 + 511) / 512;
    return (int)(w < 1 ? 1 : (w > 256 ? 256 : w));
}
// blockIdx.x -> (job, workgroup within the job, workgroups of the job)
// (the job sizes are read by all threads in ONE parallel round and scanned from LDS: a serial walk of the table in global memory
// was ~40 dependent ~1 us loads at the head of every workgroup -- most of the 0.2-0.3 ms these launches took, whatever the copy loop)
__device__ __forceinline__ const srvp_pack_job* locate_job(const srvp_pack_job* jobs, int njobs, unsigned& wg, unsigned& nwg) {
    __shared__ unsigned s_n[256];
    __shared__ int s_job;
    __shared__ unsigned s_wg;
    if (njobs <= 256) {
        for (int i = threadIdx.x; i < njobs; i += blockDim.x)
            s_n[i] = (unsigned)pack_job_wgs((long long)jobs[i].d.ntaps * jobs[i].d.J * jobs[i].d.K);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned w = blockIdx.x;
            int found = -1;
            for (int i = 0; i < njobs; ++i) {
                if (w < s_n[i]) { found = i; break; }
                w -= s_n[i];
            }
            s_job = found; s_wg = w;
        }
        __syncthreads();
        if (s_job < 0) return nullptr;
        wg = s_wg; nwg = s_n[s_job];
        return jobs + s_job;
    }
    wg = blockIdx.x;
    for (int i = 0; i < njobs; ++i) {
        nwg = (unsigned)pack_job_wgs((long long)jobs[i].d.ntaps * jobs[i].d.J * jobs[i].d.K);
        if (wg < nwg) return jobs + i;
        wg -= nwg;
    }
    return nullptr;
}

// ---- LDS-tiled path (conv / convT weights: the taps are the innermost axis of the fp32 tensor and one of the two channel axes
// comes next with stride TS = min(sj, sk) <= 16): a workgroup moves a tile of 8 "outer" x 64 "inner" channel indices x TS taps.  On
// the fp32 side the tile is 8 contiguous runs of 64 * TS floats (coalesced; the item-per-thread path above touches 64 different
// lines per wave instruction there and sustains 0.3-0.8 TB/s); the packed side moves as 16- / 32-byte pieces; the transposition
// between the two happens in LDS.  Same values and the same summation order as the paths above.
constexpr int PT_OUT = 8, PT_INN = 64;
__host__ __device__ __forceinline__ bool tile_ok(const srvp_pack_desc& d, int& TS, bool& inner_k, bool ignore_f32 = false) {
    if (!vec_ok(d, ignore_f32)) return false;
    inner_k = d.sk < d.sj;
    const long long ts = inner_k ? d.sk : d.sj;
    if (ts < 1 || ts > 16) return false;
    TS = (int)ts;
    for (int t = 0; t < d.ntaps; ++t)
        if (d.tap_set[t] == 0 ? d.tap_off[t] >= TS : (d.tap_set[t] >> TS) != 0) return false;
    return (inner_k ? d.K : d.J) % 8 == 0;
}
// padded (outer, inner) -> element offset of tap 0 in the fp32 tensor, or -1 (channel padding)
__host__ __device__ __forceinline__ long long tile_src_base(const srvp_pack_desc& d, bool inner_k, int po, int pi) {
    const int pj = inner_k ? po : pi, pk = inner_k ? pi : po;
    if (pj >= d.J || pk >= d.K) return -1;
    const int jr = real_index(pj, d.J0, d.J0r, d.J1r), kr = real_index(pk, d.K0, d.K0r, d.K1r);
    return (jr >= 0 && kr >= 0) ? (long long)jr * d.sj + (long long)kr * d.sk : -1;
}
__device__ __forceinline__ long long packed_off(const srvp_pack_desc& d, int t, int jj, int k) {
    if (d.layout == 1) {
        const int cc = k >> 6, kk = (k >> 4) & 3, kh = (k >> 3) & 1;
        const int KC = d.kc_total ? d.kc_total : (d.K >> 6);
        return ((((long long)(t * KC + cc + d.kc_off) * 4 + kk) * (d.J >> 5) + (jj >> 5)) * 64 + kh * 32 + (jj & 31)) * 8;
    }
    return ((long long)t * d.J + jj) * d.K + k;
}

__device__ void pack_job_tiled(const srvp_pack_job& j, unsigned wg, unsigned nwg, int TS, bool inner_k, float* lds) {
    const srvp_pack_desc& d = j.d;
    const float* __restrict__ src = (const float*)j.src;
    bf16_t* __restrict__ dst = (bf16_t*)j.dst;
    const int OUT = inner_k ? d.J : d.K, INN = inner_k ? d.K : d.J;
    const int to = (OUT + PT_OUT - 1) / PT_OUT, ti = (INN + PT_INN - 1) / PT_INN;
    const int pitch = PT_INN * TS + 1;
    const int ntaps = d.ntaps;
    for (int tile = (int)wg; tile < to * ti; tile += (int)nwg) {
        const int o0 = (tile / ti) * PT_OUT, i0 = (tile % ti) * PT_INN;
        __syncthreads();
        // (loads in batches of eight per thread, all issued before the first LDS store: one at a time, each iteration waited for its
        // own HBM round trip -- 18-32 of them in a row per tile)
        constexpr int LU = 8;
        for (int e0 = threadIdx.x; e0 < PT_OUT * PT_INN * TS; e0 += 256 * LU) {
            float v[LU];
            int la[LU];
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const int e = e0 + u * 256;
                la[u] = -1; v[u] = 0.f;
                if (e < PT_OUT * PT_INN * TS) {
                    const int o = e / (PT_INN * TS), rem = e - o * (PT_INN * TS), i = rem / TS, tp = rem - i * TS;
                    const long long b = tile_src_base(d, inner_k, o0 + o, i0 + i);
                    la[u] = o * pitch + rem;
                    if (b >= 0) v[u] = src[b + tp];
                }
            }
#pragma unroll
            for (int u = 0; u < LU; ++u) if (la[u] >= 0) lds[la[u]] = v[u];
        }
        __syncthreads();
        // packed items of the tile: (tap, j, eight consecutive k)
        const int nj = inner_k ? PT_OUT : PT_INN, nk8 = (inner_k ? PT_INN : PT_OUT) / 8;
        for (int it = threadIdx.x; it < ntaps * nj * nk8; it += 256) {
            // inner index fastest: consecutive threads write neighbouring pieces
            int t, jl, k8;
            if (inner_k) { k8 = it % nk8; jl = (it / nk8) % nj; t = it / (nk8 * nj); }
            else { jl = it % nj; k8 = (it / nj) % nk8; t = it / (nj * nk8); }
            const int jj = (inner_k ? o0 : i0) + jl, k = (inner_k ? i0 : o0) + k8 * 8;
            if (jj >= d.J || k >= d.K) continue;
            const int off = d.tap_off[t], set = d.tap_set[t];
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float* cell = inner_k ? lds + jl * pitch + (k8 * 8 + e) * TS : lds + (k8 * 8 + e) * pitch + jl * TS;
                if (set == 0) v[e] = cell[off];
                else {
                    float a = 0.f;
                    for (int sidx = 0; sidx < 16; ++sidx) if ((set >> sidx) & 1) a += cell[sidx];     // fp32 sum, one rounding, fixed order
                    v[e] = a;
                }
            }
            *reinterpret_cast<u32x4_t*>(dst + packed_off(d, t, jj, k)) = pack8(v);
        }
    }
}

__device__ void unpack_job_tiled(const srvp_pack_job& j, unsigned wg, unsigned nwg, int TS, bool inner_k, float* lds) {
    const srvp_pack_desc& d = j.d;
    const float* __restrict__ src = (const float*)j.src;       // packed fp32 gradient [t][J][K] (tap-major)
    float* __restrict__ dst = (float*)j.dst;
    const int OUT = inner_k ? d.J : d.K, INN = inner_k ? d.K : d.J;
    const int to = (OUT + PT_OUT - 1) / PT_OUT, ti = (INN + PT_INN - 1) / PT_INN;
    const int pitch = PT_INN * TS + 1;
    const int ntaps = d.ntaps;
    unsigned need = 0;
    for (int t = 0; t < ntaps; ++t) need |= d.tap_set[t] ? (unsigned)d.tap_set[t] : 1u << d.tap_off[t];
    for (int tile = (int)wg; tile < to * ti; tile += (int)nwg) {
        const int o0 = (tile / ti) * PT_OUT, i0 = (tile % ti) * PT_INN;
        __syncthreads();
        for (int e = threadIdx.x; e < PT_OUT * pitch; e += 256) lds[e] = 0.f;
        __syncthreads();
        // one thread per (j, eight consecutive k) walks the packed taps in order (the sums over packed taps that share a source tap
        // are formed in a fixed order) -- 64 items per tile
        const int nj = inner_k ? PT_OUT : PT_INN, nk8 = (inner_k ? PT_INN : PT_OUT) / 8;
        for (int it = threadIdx.x; it < nj * nk8; it += 256) {
            int jl, k8;
            if (inner_k) { k8 = it % nk8; jl = it / nk8; } else { jl = it % nj; k8 = it / nj; }
            const int jj = (inner_k ? o0 : i0) + jl, k = (inner_k ? i0 : o0) + k8 * 8;
            if (jj >= d.J || k >= d.K) continue;
            f32x4_t plo[SRVP_MAX_TAPS], phi[SRVP_MAX_TAPS];
#pragma unroll
            for (int t = 0; t < SRVP_MAX_TAPS; ++t)
                if (t < ntaps) {
                    const float* sp = src + ((long long)t * d.J + jj) * d.K + k;
                    plo[t] = *reinterpret_cast<const f32x4_t*>(sp); phi[t] = *reinterpret_cast<const f32x4_t*>(sp + 4);
                }
#pragma unroll
            for (int t = 0; t < SRVP_MAX_TAPS; ++t) {
                if (t >= ntaps) break;
                const unsigned m = d.tap_set[t] ? (unsigned)d.tap_set[t] : 1u << d.tap_off[t];
                const f32x4_t lo = plo[t], hi = phi[t];
                for (int sidx = 0; sidx < TS; ++sidx)
                    if ((m >> sidx) & 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float* cell = inner_k ? lds + jl * pitch + (k8 * 8 + e) * TS : lds + (k8 * 8 + e) * pitch + jl * TS;
                            cell[sidx] += e < 4 ? lo[e] : hi[e - 4];
                        }
                    }
            }
        }
        __syncthreads();
        constexpr int LU = 8;
        for (int e0 = threadIdx.x; e0 < PT_OUT * PT_INN * TS; e0 += 256 * LU) {
            float v[LU];
            long long ga[LU];
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const int e = e0 + u * 256;
                ga[u] = -1; v[u] = 0.f;
                if (e < PT_OUT * PT_INN * TS) {
                    const int o = e / (PT_INN * TS), rem = e - o * (PT_INN * TS), i = rem / TS, tp = rem - i * TS;
                    if ((need >> tp) & 1) {
                        const long long b = tile_src_base(d, inner_k, o0 + o, i0 + i);
                        if (b >= 0) { ga[u] = b + tp; v[u] = dst[b + tp] + lds[o * pitch + rem]; }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < LU; ++u) if (ga[u] >= 0) dst[ga[u]] = v[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Lean tile kernels (round 4).  Measured per job in isolation (tools/pack_time.py, PER_JOB=1): on the kernels above EVERY job, from
// 0.04 M to 4 M elements, took the same 21 us (pack) / 39 us (unpack) -- the time of ONE workgroup's dependent chain (job lookup,
// descriptor fields from global memory, two or three rounds of loads, tap tables read inside the loops), at 2-3 workgroups per CU
// (164 VGPRs / 32 KB of LDS), so a whole network cost (workgroups / resident workgroups) x that chain: 95 + 236 us forward,
// 130 + 190 us backward, whatever the access pattern.  Here: one workgroup per SMALL tile (18.5 KB of LDS, <= 64 VGPRs: 8 workgroups
// per CU), the job descriptor copied into LDS once, every global load of the tile in flight at once (16-byte loads wherever the
// fp32 run is contiguous), tap tables read from LDS.
//   pack   tile = JT j x 16 k x all taps (JT = 32, or 16 for 4x4 kernels): per packed tap one (half) MFMA fragment of 64 (32) lanes
//          x 16 bytes in the fragment-major layout; fp32 side: JT (16) contiguous runs of 16 TS (JT TS) floats
//   unpack tile = JT j x 32 k x all taps (JT = 16 / 8): packed fp32 gradient rows of 128 bytes; fp32 side: runs of 32 TS (JT TS)
// Same values and summation order as pack_one / unpack_one (tests/test_gpu_blocks.py: byte-equal on whole networks).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int LT_FLOATS = 4640;
struct LeanGeo { int TS, JT, KT; bool inner_k; };
__host__ __device__ inline bool lean_geo(const srvp_pack_desc& d, bool unpack, LeanGeo& g) {
    if (!tile_ok(d, g.TS, g.inner_k, unpack)) return false;
    if (!unpack) { g.JT = g.TS <= 9 ? 32 : 16; g.KT = 16; }
    else { g.JT = g.TS <= 9 ? 16 : 8; g.KT = 32; }
    if (d.J % g.JT != 0 || d.K % g.KT != 0 || (d.dst_f32 && !unpack)) return false;
    const int OUTN = g.inner_k ? g.JT : g.KT, INN = g.inner_k ? g.KT : g.JT;
    if ((INN * g.TS) % 4 != 0 || OUTN * (INN * g.TS + 1) > LT_FLOATS) return false;
    return (long long)(d.J / g.JT) * (d.K / g.KT) < (1ll << 24);
}
__host__ __device__ inline int lean_tiles(const srvp_pack_desc& d, bool unpack) {
    LeanGeo g;
    return lean_geo(d, unpack, g) ? (d.J / g.JT) * (d.K / g.KT) : 0;
}

template <bool UNPACK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void pack_lean_kernel(const srvp_pack_job* __restrict__ jobs, int njobs) {
    __shared__ float lds[LT_FLOATS];
    __shared__ unsigned s_n[256];
    __shared__ srvp_pack_job sj;
    __shared__ int s_job;
    __shared__ unsigned s_tile;
    // ---- blockIdx.x -> (job, tile): all job sizes in one parallel round, scanned from LDS
    for (int i = threadIdx.x; i < njobs; i += 256) s_n[i] = (unsigned)lean_tiles(jobs[i].d, UNPACK);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned w = blockIdx.x;
        int found = -1;
        for (int i = 0; i < njobs; ++i) {
            if (w < s_n[i]) { found = i; break; }
            w -= s_n[i];
        }
        s_job = found; s_tile = w;
    }
    __syncthreads();
    if (s_job < 0) return;
    {
        const unsigned* gp = reinterpret_cast<const unsigned*>(jobs + s_job);
        unsigned* lp = reinterpret_cast<unsigned*>(&sj);
        if (threadIdx.x < sizeof(srvp_pack_job) / 4) lp[threadIdx.x] = gp[threadIdx.x];
    }
    __syncthreads();
    const srvp_pack_desc& d = sj.d;
    LeanGeo g;
    lean_geo(d, UNPACK, g);
    const int TS = g.TS, JT = g.JT, KT = g.KT;
    const bool inner_k = g.inner_k;
    const int tj = d.J / JT;
    const int tile = (int)s_tile;
    const int j0 = (tile % tj) * JT, k0 = (tile / tj) * KT;
    const int o0 = inner_k ? j0 : k0, i0 = inner_k ? k0 : j0;
    const int OUTN = inner_k ? JT : KT, INN = inner_k ? KT : JT;
    const int run = INN * TS, pitch = run + 1, total = OUTN * run;
    const unsigned r_run = (unsigned)(((1ull << 32) + run - 1) / run), r_ts = (unsigned)(((1ull << 32) + TS - 1) / TS);
    const int ntaps = d.ntaps;
    static_assert(sizeof(srvp_pack_job) / 4 <= 256 && sizeof(srvp_pack_job) % 4 == 0, "descriptor copy");
    constexpr int LU = 5;                                  // 16-byte units per thread: total / 4 <= 1160 <= 5 * 256
    if constexpr (!UNPACK) {
        const float* __restrict__ src = (const float*)sj.src;
        bf16_t* __restrict__ dst = (bf16_t*)sj.dst;
        f32x4_t v[LU];
        int la[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int e = (threadIdx.x + u * 256) * 4;
            la[u] = -1; v[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (e < total) {
                const int o = (int)__umulhi((unsigned)e, r_run), rem = e - o * run;
                la[u] = o * pitch + rem;
                const int ia = (int)__umulhi((unsigned)rem, r_ts), tpa = rem - ia * TS;
                const int ib = (int)__umulhi((unsigned)(rem + 3), r_ts), tpb = rem + 3 - ib * TS;
                const long long ba = tile_src_base(d, inner_k, o0 + o, i0 + ia), bb = tile_src_base(d, inner_k, o0 + o, i0 + ib);
                if (ba >= 0 && bb >= 0 && bb + tpb == ba + tpa + 3 && ((ba + tpa) & 3) == 0) {
                    v[u] = *reinterpret_cast<const f32x4_t*>(src + ba + tpa);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int i = (int)__umulhi((unsigned)(rem + c), r_ts), tp = rem + c - i * TS;
                        const long long b = tile_src_base(d, inner_k, o0 + o, i0 + i);
                        if (b >= 0) v[u][c] = src[b + tp];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < LU; ++u)
            if (la[u] >= 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) lds[la[u] + c] = v[u][c];
            }
        __syncthreads();
        // packed items (tap, k half, j): consecutive threads = consecutive j = consecutive 16-byte pieces of a fragment
        for (int it = threadIdx.x; it < ntaps * 2 * JT; it += 256) {
            const int t = it / (2 * JT), r = it - t * (2 * JT), kh = r / JT, jl = r - kh * JT;
            const int off = d.tap_off[t], set = d.tap_set[t];
            float o8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float* cell = inner_k ? lds + jl * pitch + (kh * 8 + e) * TS : lds + (kh * 8 + e) * pitch + jl * TS;
                if (set == 0) o8[e] = cell[off];
                else {
                    // every tap of the cell in flight, then the fp32 sum in ascending tap order (a + 0.f == a exactly)
                    float w[16];
#pragma unroll
                    for (int sidx = 0; sidx < 16; ++sidx) w[sidx] = sidx < TS ? cell[sidx] : 0.f;
                    float a = 0.f;
#pragma unroll
                    for (int sidx = 0; sidx < 16; ++sidx) a += ((set >> sidx) & 1) ? w[sidx] : 0.f;
                    o8[e] = a;
                }
            }
            *reinterpret_cast<u32x4_t*>(dst + packed_off(d, t, j0 + jl, k0 + kh * 8)) = pack8(o8);
        }
    } else {
        const float* __restrict__ src = (const float*)sj.src;       // packed fp32 gradient [t][J][K] (tap-major)
        float* __restrict__ dst = (float*)sj.dst;
        unsigned need = 0;
        for (int t = 0; t < ntaps; ++t) need |= d.tap_set[t] ? (unsigned)d.tap_set[t] : 1u << d.tap_off[t];
        const unsigned full = (1u << TS) - 1u;
        // the flat side's current values: in flight while the packed taps are summed
        f32x4_t v[LU];
        long long ga[LU];          // >= 0: one 16-byte read-modify-write at ga; -2: element by element; -1: nothing
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int e = (threadIdx.x + u * 256) * 4;
            ga[u] = -1; v[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (e < total) {
                const int o = (int)__umulhi((unsigned)e, r_run), rem = e - o * run;
                const int ia = (int)__umulhi((unsigned)rem, r_ts), tpa = rem - ia * TS;
                const int ib = (int)__umulhi((unsigned)(rem + 3), r_ts), tpb = rem + 3 - ib * TS;
                const long long ba = tile_src_base(d, inner_k, o0 + o, i0 + ia), bb = tile_src_base(d, inner_k, o0 + o, i0 + ib);
                if (need == full && ba >= 0 && bb >= 0 && bb + tpb == ba + tpa + 3 && ((ba + tpa) & 3) == 0) {
                    ga[u] = ba + tpa;
                    v[u] = *reinterpret_cast<const f32x4_t*>(dst + ga[u]);
                } else ga[u] = -2;
            }
        }
        for (int e = threadIdx.x; e < OUTN * pitch; e += 256) lds[e] = 0.f;
        __syncthreads();
        // item = (j, four consecutive k): owns its LDS cells; packed taps in ascending order (= unpack_one's summation order)
        const int nk4 = KT / 4;
        for (int it = threadIdx.x; it < JT * nk4; it += 256) {
            const int k4 = it % nk4, jl = it / nk4;
            const float* sp = src + ((long long)(j0 + jl)) * d.K + k0 + k4 * 4;
            const long long tstride = (long long)d.J * d.K;
            for (int t0 = 0; t0 < ntaps; t0 += 4) {
                f32x4_t pv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    pv[q] = t0 + q < ntaps ? *reinterpret_cast<const f32x4_t*>(sp + (t0 + q) * tstride) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (t0 + q >= ntaps) break;
                    const unsigned m = d.tap_set[t0 + q] ? (unsigned)d.tap_set[t0 + q] : 1u << d.tap_off[t0 + q];
                    for (int sidx = 0; sidx < TS; ++sidx)
                        if ((m >> sidx) & 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float* cell = inner_k ? lds + jl * pitch + (k4 * 4 + e) * TS : lds + (k4 * 4 + e) * pitch + jl * TS;
                                cell[sidx] += pv[q][e];
                            }
                        }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int e = (threadIdx.x + u * 256) * 4;
            if (ga[u] >= 0) {
                const int o = (int)__umulhi((unsigned)e, r_run), rem = e - o * run;
                f32x4_t w = v[u];
#pragma unroll
                for (int c = 0; c < 4; ++c) w[c] += lds[o * pitch + rem + c];
                *reinterpret_cast<f32x4_t*>(dst + ga[u]) = w;
            } else if (ga[u] == -2) {
                const int o = (int)__umulhi((unsigned)e, r_run), rem = e - o * run;
                for (int c = 0; c < 4; ++c) {
                    const int i = (int)__umulhi((unsigned)(rem + c), r_ts), tp = rem + c - i * TS;
                    if (!((need >> tp) & 1)) continue;
                    const long long b = tile_src_base(d, inner_k, o0 + o, i0 + i);
                    if (b >= 0) dst[b + tp] += lds[o * pitch + rem + c];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void pack_multi_kernel(const srvp_pack_job* __restrict__ jobs, int njobs, int g_pack_tiled) {
    unsigned wg, nwg;
    const srvp_pack_job* jp = locate_job(jobs, njobs, wg, nwg);
    if (!jp) return;
    const srvp_pack_job& j = *jp;
    const srvp_pack_desc& d = j.d;
    __shared__ float tile_lds[PT_OUT * (PT_INN * 16 + 1)];
    int TS; bool inner_k;
    if ((g_pack_tiled & 1) && tile_ok(d, TS, inner_k)) { pack_job_tiled(j, wg, nwg, TS, inner_k, tile_lds); return; }
    if (!vec_ok(d)) {
        PackArgs a;
        job_args(j, a);
        const long long total = (long long)a.ntaps * a.J * a.K;
        for (long long i = (long long)wg * blockDim.x + threadIdx.x; i < total; i += (long long)nwg * blockDim.x)
            pack_one((const float*)j.src, (bf16_t*)j.dst, a, i);
        return;
    }
    const float* __restrict__ src = (const float*)j.src;
    bf16_t* __restrict__ dst = (bf16_t*)j.dst;
    const int J = d.J, K = d.K, K8 = K >> 3, ntaps = d.ntaps;
    unsigned need = 0;                                   // source taps any packed tap reads
    for (int t = 0; t < ntaps; ++t) need |= d.tap_set[t] ? (unsigned)d.tap_set[t] : 1u << d.tap_off[t];
    const long long items = (long long)J * K8;
    for (long long q = (long long)wg * blockDim.x + threadIdx.x; q < items; q += (long long)nwg * blockDim.x) {
        const int jj = (int)(q % J), k8 = (int)(q / J);
        const int jr = real_index(jj, d.J0, d.J0r, d.J1r);
        long long base[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kr = real_index(k8 * 8 + e, d.K0, d.K0r, d.K1r);
            base[e] = (jr >= 0 && kr >= 0) ? (long long)jr * d.sj + (long long)kr * d.sk : -1;
        }
        // every source element this item needs (the taps lie within 16 elements of each base) is loaded up front -- up to 128
        // independent loads in flight instead of one dependent (tap-table load -> element loads -> store) chain per tap
        float w[8][16];
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx) w[e][sidx] = (base[e] >= 0 && ((need >> sidx) & 1)) ? src[base[e] + sidx] : 0.f;
        for (int t = 0; t < ntaps; ++t) {
            const int off = d.tap_off[t], set = d.tap_set[t];
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = 0.f;
                if (set == 0) {
#pragma unroll
                    for (int sidx = 0; sidx < 16; ++sidx) if (sidx == off) v[e] = w[e][sidx];
                } else {
#pragma unroll
                    for (int sidx = 0; sidx < 16; ++sidx) if ((set >> sidx) & 1) v[e] += w[e][sidx];   // fp32 sum, one rounding, fixed order
                }
            }
            long long o;
            const int k = k8 * 8;
            if (d.layout == 1) {
                const int cc = k >> 6, kk = (k >> 4) & 3, kh = (k >> 3) & 1;
                const int KC = d.kc_total ? d.kc_total : (K >> 6);
                o = ((((long long)(t * KC + cc + d.kc_off) * 4 + kk) * (J >> 5) + (jj >> 5)) * 64 + kh * 32 + (jj & 31)) * 8;
            } else {
                o = ((long long)t * J + jj) * K + k;
            }
            *reinterpret_cast<u32x4_t*>(dst + o) = pack8(v);
        }
    }
}
__global__ __launch_bounds__(256) void unpack_multi_kernel(const srvp_pack_job* __restrict__ jobs, int njobs, int g_pack_tiled) {
    unsigned wg, nwg;
    const srvp_pack_job* jp = locate_job(jobs, njobs, wg, nwg);
    if (!jp) return;
    const srvp_pack_job& j = *jp;
    const srvp_pack_desc& d = j.d;
    __shared__ float tile_lds[PT_OUT * (PT_INN * 16 + 1)];
    int TS; bool inner_k;
    if ((g_pack_tiled & 2) && tile_ok(d, TS, inner_k, true)) { unpack_job_tiled(j, wg, nwg, TS, inner_k, tile_lds); return; }
    if (!vec_ok(d, true)) {
        PackArgs a;
        job_args(j, a);
        const long long total = (long long)a.ntaps * a.J * a.K;
        for (long long i = (long long)wg * blockDim.x + threadIdx.x; i < total; i += (long long)nwg * blockDim.x)
            unpack_one((const float*)j.src, (float*)j.dst, a, i);
        return;
    }
    const float* __restrict__ src = (const float*)j.src;
    float* __restrict__ dst = (float*)j.dst;
    const int J = d.J, K = d.K, K8 = K >> 3, ntaps = d.ntaps;
    unsigned need = 0, shared = 0;                      // source taps written / written by more than one packed tap
    for (int t = 0; t < ntaps; ++t) {
        const unsigned m = d.tap_set[t] ? (unsigned)d.tap_set[t] : 1u << d.tap_off[t];
        shared |= need & m; need |= m;
    }
    const long long items = (long long)J * K8;
    for (long long q = (long long)wg * blockDim.x + threadIdx.x; q < items; q += (long long)nwg * blockDim.x) {
        const int jj = (int)(q % J), k8 = (int)(q / J);
        const int jr = real_index(jj, d.J0, d.J0r, d.J1r);
        if (jr < 0) continue;
        float acc[8][16];
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx) acc[e][sidx] = 0.f;
        for (int t = 0; t < ntaps; ++t) {
            const unsigned m = d.tap_set[t] ? (unsigned)d.tap_set[t] : 1u << d.tap_off[t];
            const float* sp = src + ((long long)t * J + jj) * K + k8 * 8;
            const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(sp), hi = *reinterpret_cast<const f32x4_t*>(sp + 4);
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx)
                if ((m >> sidx) & 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[e][sidx] += lo[e]; acc[4 + e][sidx] += hi[e]; }
                }
        }
        // the item owns every element (jr, kr, *) of this job: one plain read-modify-write per element (the sums over the packed
        // taps that share a source tap were formed above, in a fixed order)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kr = real_index(k8 * 8 + e, d.K0, d.K0r, d.K1r);
            if (kr < 0) continue;
            float* dp = dst + (long long)jr * d.sj + (long long)kr * d.sk;
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx)
                if ((need >> sidx) & 1) dp[sidx] += acc[e][sidx];
        }
    }
}

int fill_pack(const srvp_pack_desc* d, PackArgs& a) {
    SRVP_REQUIRE(d && d->ntaps >= 1 && d->ntaps <= SRVP_MAX_TAPS, "srvp_pack: ntaps");
    a.ntaps = d->ntaps;
    for (int t = 0; t < SRVP_MAX_TAPS; ++t) { a.tap_off[t] = t < d->ntaps ? d->tap_off[t] : 0; a.tap_set[t] = t < d->ntaps ? d->tap_set[t] : 0; }
    a.J = d->J; a.K = d->K; a.J0 = d->J0; a.J0r = d->J0r; a.J1r = d->J1r; a.K0 = d->K0; a.K0r = d->K0r; a.K1r = d->K1r;
    a.sj = d->sj; a.sk = d->sk;
    a.layout = d->layout;
    a.dst_f32 = d->dst_f32;
    a.kc_total = d->kc_total; a.kc_off = d->kc_off;
    SRVP_REQUIRE(!(d->dst_f32 && d->layout), "srvp_pack: fp32 packed weights are tap-major only");
    SRVP_REQUIRE(d->layout == 0 || (d->layout == 1 && d->J % 32 == 0 && d->K % 64 == 0), "srvp_pack: layout %d needs J %% 32 == 0, K %% 64 == 0", d->layout);
    return SRVP_OK;
}

// dsel[b][hw][c] = sum_t dcat[(t*B+b)][hw][coff + c]
template <class E>
__global__ void skip_grad_reduce_kernel(const E* __restrict__ dcat, int cstride, int coff, int C, int HW, int T, int B,
                                        E* __restrict__ dsel) {
    const int CG = C / 8;
    long long total = (long long)B * HW * CG;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cg = (int)(i % CG); long long q = i / CG;
        int hw = (int)(q % HW); int b = (int)(q / HW);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int t = 0; t < T; ++t) {
            size_t off = (((size_t)(t * B + b)) * HW + hw) * cstride + coff + cg * 8;
            float f[8];
            El<E>::ld8(dcat + off, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
        El<E>::st8(dsel + ((size_t)b * HW + hw) * C + cg * 8, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// ELBO terms
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_sum_to(double v, double* out) {
    __shared__ double part[4];
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += part[w];
        atomicAdd(out, s);
    }
}

__global__ __launch_bounds__(256) void nll_kernel(const float* __restrict__ xr, const float* __restrict__ x, float* __restrict__ dxr,
                                                  long long n, float inv2s2, float logc, float gcoef, double* out) {
    double acc = 0.;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= n) {
            f32x4_t a = *reinterpret_cast<const f32x4_t*>(xr + i), b = *reinterpret_cast<const f32x4_t*>(x + i);
            f32x4_t d = a - b;
            acc += (double)((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * inv2s2 + 4.0 * logc;
            if (dxr) *reinterpret_cast<f32x4_t*>(dxr + i) = d * gcoef;
        } else {
            for (long long j = i; j < n; ++j) {
                float d = xr[j] - x[j];
                acc += (double)(d * d) * inv2s2 + logc;
                if (dxr) dxr[j] = d * gcoef;
            }
        }
    }
    block_sum_to(acc, out);
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(__expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

// KL(N(ql,qs) || N(pl,ps)) = 0.5*(r + ((ql-pl)/ps)^2 - 1 - log r), r = (qs/ps)^2 ; s = softplus(raw)+1e-8
__global__ __launch_bounds__(256) void kl_kernel(const float* __restrict__ q, const float* __restrict__ p, float* dq, float* dp,
                                                 long long rows, int d, float gscale, double* out) {
    double acc = 0.;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * d; i += (long long)gridDim.x * blockDim.x) {
        long long r = i / d; int c = (int)(i - r * d);
        float ql = q[r * 2 * d + c], qraw = q[r * 2 * d + d + c];
        float qs = softplus_f(qraw) + 1e-8f;
        float pl = 0.f, ps = 1.f, praw = 0.f;
        if (p) { pl = p[r * 2 * d + c]; praw = p[r * 2 * d + d + c]; ps = softplus_f(praw) + 1e-8f; }
        float ratio = qs / ps;
        float r2 = ratio * ratio;
        float t = (ql - pl) / ps;
        acc += 0.5 * ((double)r2 + (double)t * t - 1.0 - (double)logf(r2));
        if (dq) {
            float dql = t / ps;
            float dqs = qs / (ps * ps) - 1.f / qs;
            float dsq = qraw > 20.f ? 1.f : sigmoid_f(qraw);
            dq[r * 2 * d + c] = gscale * dql;
            dq[r * 2 * d + d + c] = gscale * dqs * dsq;
        }
        if (dp && p) {
            float dpl = -t / ps;
            float dps = -(qs * qs) / (ps * ps * ps) - t * t / ps + 1.f / ps;
            float dsp = praw > 20.f ? 1.f : sigmoid_f(praw);
            dp[r * 2 * d + c] = gscale * dpl;
            dp[r * 2 * d + d + c] = gscale * dps * dsp;
        }
    }
    block_sum_to(acc, out);
}

// one wave per row: ||row||_2
__global__ __launch_bounds__(256) void l2rows_kernel(const float* __restrict__ res, float* dres, long long rows, int d, float gscale,
                                                     double* out) {
    const int lane = threadIdx.x & 63;
    double acc = 0.;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane; c < d; c += 64) { float v = res[r * d + c]; s += v * v; }
        s = wave_sum(s);
        float nrm = sqrtf(s);
        if (lane == 0) acc += nrm;
        if (dres) {
            float inv = nrm > 0.f ? gscale / nrm : 0.f;
            for (int c = lane; c < d; c += 64) dres[r * d + c] = res[r * d + c] * inv;
        }
    }
    block_sum_to(acc, out);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float step_size, float beta1, float beta2,
                                                   float eps, float inv_bc2_sqrt, float grad_scale) {
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= n) {
            f32x4_t gv = *reinterpret_cast<const f32x4_t*>(g + i) * grad_scale;
            f32x4_t mv = *reinterpret_cast<f32x4_t*>(m + i), vv = *reinterpret_cast<f32x4_t*>(v + i);
            f32x4_t pv = *reinterpret_cast<f32x4_t*>(p + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mv[e] = beta1 * mv[e] + (1.f - beta1) * gv[e];
                vv[e] = beta2 * vv[e] + (1.f - beta2) * gv[e] * gv[e];
                float denom = sqrtf(vv[e]) * inv_bc2_sqrt + eps;
                pv[e] -= step_size * (mv[e] / denom);
            }
            *reinterpret_cast<f32x4_t*>(m + i) = mv; *reinterpret_cast<f32x4_t*>(v + i) = vv;
            *reinterpret_cast<f32x4_t*>(p + i) = pv;
        } else {
            for (long long j = i; j < n; ++j) {
                float gj = g[j] * grad_scale;
                float mj = beta1 * m[j] + (1.f - beta1) * gj;
                float vj = beta2 * v[j] + (1.f - beta2) * gj * gj;
                m[j] = mj; v[j] = vj;
                p[j] -= step_size * (mj / (sqrtf(vj) * inv_bc2_sqrt + eps));
            }
        }
    }
}

inline unsigned capped_grid(long long items, int per_block, int cap) {
    long long b = (items + per_block - 1) / per_block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}
// the ELBO reductions (one fp64 atomic per workgroup onto the accumulator): a single workgroup in deterministic mode
inline unsigned elbo_grid(long long items, int per_block, int cap) { return g_srvp_det ? 1u : capped_grid(items, per_block, cap); }

}  // namespace

extern "C" int srvp_pack_weight(const float* src, void* dst, const srvp_pack_desc* d, void* stream) {
    PackArgs a;
    int rc = fill_pack(d, a);
    if (rc) return rc;
    SRVP_REQUIRE(src && dst, "srvp_pack_weight: null pointer");
    long long total = (long long)a.ntaps * a.J * a.K;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst, a);
    SRVP_CHECK_LAUNCH("srvp_pack_weight");
    return SRVP_OK;
}

extern "C" int srvp_unpack_wgrad(const float* src, float* dst, const srvp_pack_desc* d, void* stream) {
    PackArgs a;
    int rc = fill_pack(d, a);
    if (rc) return rc;
    SRVP_REQUIRE(src && dst, "srvp_unpack_wgrad: null pointer");
    long long total = (long long)a.ntaps * a.J * a.K;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, a);
    SRVP_CHECK_LAUNCH("srvp_unpack_wgrad");
    return SRVP_OK;
}

// A/B switch SRVP_PACK_TILED: bit 0 = pack, bit 1 = unpack on the LDS-tiled path.  Default 2: the tiled UNPACK (a read-modify-write of the
// fp32 gradient whose item-per-thread form touched 64 lines per wave instruction) went 0.63 -> 0.29 ms in isolation, 0.89 -> 0.36 ms
// inside the 192-sequence step; the tiled PACK measured no faster than the item-per-thread form (0.44 vs 0.40 ms: its writes, 16-byte
// pieces of a fragment-major tensor, are the scattered side either way) and stays off.
static int pack_tiled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SRVP_PACK_TILED"); on = e ? atoi(e) : 2; }
    return on;
}
extern "C" int srvp_pack_weight_multi(const srvp_pack_job* jobs_dev, int njobs, int64_t total_wgs, void* stream) {
    SRVP_REQUIRE(jobs_dev && njobs > 0 && total_wgs > 0 && total_wgs < (1ll << 31), "srvp_pack_weight_multi: bad args");
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)total_wgs), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs, pack_tiled());
    SRVP_CHECK_LAUNCH("srvp_pack_weight_multi");
    return SRVP_OK;
}

extern "C" int srvp_unpack_wgrad_multi(const srvp_pack_job* jobs_dev, int njobs, int64_t total_wgs, void* stream) {
    SRVP_REQUIRE(jobs_dev && njobs > 0 && total_wgs > 0 && total_wgs < (1ll << 31), "srvp_unpack_wgrad_multi: bad args");
    hipLaunchKernelGGL(unpack_multi_kernel, dim3((unsigned)total_wgs), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs, pack_tiled());
    SRVP_CHECK_LAUNCH("srvp_unpack_wgrad_multi");
    return SRVP_OK;
}

extern "C" int srvp_pack_job_wgs(int64_t total) { return pack_job_wgs(total); }

// Lean tile kernels: `jobs_dev` holds ONLY jobs with srvp_pack_job_tiles(...) > 0 (the caller sends the others through the multi
// launches above); total_tiles = the sum of their tile counts, njobs <= 256.
extern "C" int srvp_pack_job_tiles(const srvp_pack_desc* d, int unpack) { return d ? lean_tiles(*d, unpack != 0) : 0; }
extern "C" int srvp_pack_weight_tiles(const srvp_pack_job* jobs_dev, int njobs, int64_t total_tiles, void* stream) {
    SRVP_REQUIRE(jobs_dev && njobs > 0 && njobs <= 256 && total_tiles > 0 && total_tiles < (1ll << 31), "srvp_pack_weight_tiles: bad args (at most 256 jobs)");
    hipLaunchKernelGGL(pack_lean_kernel<false>, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    SRVP_CHECK_LAUNCH("srvp_pack_weight_tiles");
    return SRVP_OK;
}
extern "C" int srvp_unpack_wgrad_tiles(const srvp_pack_job* jobs_dev, int njobs, int64_t total_tiles, void* stream) {
    SRVP_REQUIRE(jobs_dev && njobs > 0 && njobs <= 256 && total_tiles > 0 && total_tiles < (1ll << 31), "srvp_unpack_wgrad_tiles: bad args (at most 256 jobs)");
    hipLaunchKernelGGL(pack_lean_kernel<true>, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    SRVP_CHECK_LAUNCH("srvp_unpack_wgrad_tiles");
    return SRVP_OK;
}

extern "C" int srvp_skip_grad_reduce(const void* dcat, int cstride, int coff, int C, int HW, int T, int B, void* dsel,
                                     void* stream) {
    SRVP_REQUIRE(dcat && dsel && C % 8 == 0 && coff % 8 == 0 && cstride % 8 == 0, "srvp_skip_grad_reduce: bad args");
    long long total = (long long)B * HW * (C / 8);
    hipLaunchKernelGGL(skip_grad_reduce_kernel<bf16_t>, dim3(capped_grid(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dcat, cstride, coff, C, HW, T, B, (bf16_t*)dsel);
    SRVP_CHECK_LAUNCH("srvp_skip_grad_reduce");
    return SRVP_OK;
}
extern "C" int srvp_skip_grad_reduce_f32(const void* dcat, int cstride, int coff, int C, int HW, int T, int B, void* dsel,
                                         void* stream) {
    SRVP_REQUIRE(dcat && dsel && C % 8 == 0 && coff % 8 == 0 && cstride % 8 == 0, "srvp_skip_grad_reduce_f32: bad args");
    long long total = (long long)B * HW * (C / 8);
    hipLaunchKernelGGL(skip_grad_reduce_kernel<float>, dim3(capped_grid(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)dcat, cstride, coff, C, HW, T, B, (float*)dsel);
    SRVP_CHECK_LAUNCH("srvp_skip_grad_reduce_f32");
    return SRVP_OK;
}

extern "C" int srvp_nll(const float* x_, const float* x, float* d_x_, int64_t n, float scale, float gscale, double* out,
                        void* stream) {
    SRVP_REQUIRE(x_ && x && out && scale > 0.f, "srvp_nll: bad args");
    const float inv2s2 = 1.f / (2.f * scale * scale);
    const float logc = logf(scale) + 0.91893853320467274f;
    hipLaunchKernelGGL(nll_kernel, dim3(elbo_grid(n, 1024, 2048)), dim3(256), 0, (hipStream_t)stream, x_, x, d_x_, (long long)n,
                       inv2s2, logc, gscale / (scale * scale), out);
    SRVP_CHECK_LAUNCH("srvp_nll");
    return SRVP_OK;
}

extern "C" int srvp_kl(const float* q, const float* p, float* dq, float* dp, int64_t rows, int d, float gscale, double* out,
                       void* stream) {
    SRVP_REQUIRE(q && out, "srvp_kl: bad args");
    if (rows * d <= 0) return SRVP_OK;
    hipLaunchKernelGGL(kl_kernel, dim3(elbo_grid(rows * d, 256, 1024)), dim3(256), 0, (hipStream_t)stream, q, p, dq, dp,
                       (long long)rows, d, gscale, out);
    SRVP_CHECK_LAUNCH("srvp_kl");
    return SRVP_OK;
}

extern "C" int srvp_l2rows(const float* res, float* d_res, int64_t rows, int d, float gscale, double* out, void* stream) {
    SRVP_REQUIRE(res && out, "srvp_l2rows: bad args");
    if (rows <= 0) return SRVP_OK;
    hipLaunchKernelGGL(l2rows_kernel, dim3(elbo_grid(rows, 4, 1024)), dim3(256), 0, (hipStream_t)stream, res, d_res,
                       (long long)rows, d, gscale, out);
    SRVP_CHECK_LAUNCH("srvp_l2rows");
    return SRVP_OK;
}

extern "C" int srvp_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                         int step, float grad_scale, void* stream) {
    SRVP_REQUIRE(p && g && m && v && step >= 1, "srvp_adam: bad args");
    if (n <= 0) return SRVP_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(capped_grid(n, 1024, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n,
                       (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
    SRVP_CHECK_LAUNCH("srvp_adam");
    return SRVP_OK;
}
