// Batch normalisation over NHWC-flattened [M][C] activations, HBM-bound streaming kernels.
//
// forward (train):  pass 1  bn_stats      per row-slice partial (count, mean, M2) per channel,
//                                          16-byte vector loads along C (coalesced), fp32 maths
//                   pass 1b bn_finalize   merges slices with Chan's parallel-variance update,
//                                          writes mean / invstd / running stats / scale+shift
//                   pass 2  bn_apply      y = act(x*scale + shift [+ residual])      (fused ReLU / add)
// backward:         pass 1  bn_bwd_reduce per-slice sums of dy' and dy'*xhat (dy' = ReLU-masked dy)
//                   pass 1b bn_bwd_final  dgamma / dbeta (+= when accumulating) and apply coeffs
//                   pass 2  bn_bwd_apply  dx = a*(dy' - b1 - xhat*b2); optional dresidual = dy'
// Algorithmic bytes: forward 3*E*s (x twice, y once), backward reads dy,x(,y) twice + writes dx.
#include "common.h"
#include <atomic>
#include <mutex>
#include <type_traits>

namespace {

template <typename T, int VEC> struct VLoad {
    float f[VEC];
    __device__ __forceinline__ void load(const T* p) {
        if constexpr (VEC == 1) {
            f[0] = to_f32(p[0]);
        } else {
            Vec16<T> v;
            v.v = *(const decltype(v.v)*)p;
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] = v.get(i);
        }
    }
    __device__ __forceinline__ void store(T* p) const {
        if constexpr (VEC == 1) {
            p[0] = from_f32<T>(f[0]);
        } else {
            Vec16<T> v;
#pragma unroll
            for (int i = 0; i < VEC; ++i) v.set(i, f[i]);
            *(decltype(v.v)*)p = v.v;
        }
    }
};

struct SliceGeom {
    int cols;        // vector columns (C / VEC)
    int cpb;         // vector columns per block (<= 256)
    int rpb;         // rows per block iteration (256 / cpb)
    int slice_rows;  // rows per slice
    int slices;      // grid.x
    int colgroups;   // grid.y
};

SliceGeom make_geom(int64_t M, int C, int vec) {
    SliceGeom g;
    g.cols = C / vec;
    // <= 32 vector columns (512 B of a row) per block: wide-C tensors then have 8 row-threads per
    // column and spread their column groups over grid.y, so small late-stage tensors still launch
    // ~2048 blocks without shrinking the rows per slice (each slice costs C*12 B of partials)
    g.cpb = g.cols < 32 ? g.cols : 32;
    g.rpb = 256 / g.cpb;
    g.colgroups = ceil_div(g.cols, g.cpb);
    constexpr int target = 2048, rpt_max = 32;       // swept 768..4096 blocks / 32..128 rows: flat within 0.4 %
    int64_t want_slices = target / g.colgroups;
    if (want_slices < 1) want_slices = 1;
    int64_t sr = ceil_div64(M, want_slices);
    int64_t rpt = ceil_div64(sr, g.rpb);
    rpt = rpt < 8 ? 8 : (rpt > rpt_max ? rpt_max : rpt);
    sr = (int64_t)g.rpb * rpt;
    if (sr < 64) sr = ceil_div64(64, g.rpb) * g.rpb;          // partial traffic 6/sr of the tensor: < 10 %
    if (ceil_div64(M, sr) > 8192) sr = ceil_div64(ceil_div64(M, 8192), g.rpb) * g.rpb;
    g.slice_rows = (int)sr;
    g.slices = (int)ceil_div64(M, sr);
    return g;
}

// workspace layout (floats): [0,C) scale  [C,2C) shift  [2C,3C) coefB1  [3C,4C) coefB2
//                            then partials: slices * C * 3
__host__ __device__ inline size_t ws_partials_off(int C) { return (size_t)4 * C; }

template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x,
                                                       float* __restrict__ partial, int64_t M,
                                                       int C, SliceGeom g) {
    __shared__ float sh[3][256 * (VEC > 1 ? VEC : 1)];
    const int tid = threadIdx.x, tcol = tid % g.cpb, trow = tid / g.cpb;
    const int vcol = blockIdx.y * g.cpb + tcol;
    const bool active = trow < g.rpb && vcol < g.cols;
    const int64_t m0 = (int64_t)blockIdx.x * g.slice_rows;
    const int64_t m1 = m0 + g.slice_rows < M ? m0 + g.slice_rows : M;
    float s[VEC], ss[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = ss[i] = 0.f;
    int n = 0;
    if (active) {
        for (int64_t m = m0 + trow; m < m1; m += g.rpb) {
            VLoad<T, VEC> v;
            v.load(x + m * C + (int64_t)vcol * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                s[i] += v.f[i];
                ss[i] += v.f[i] * v.f[i];
            }
            ++n;
        }
    }
    // per-thread (n, mean, M2), then Chan merge over the block's row-threads
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        float mean = n ? s[i] / n : 0.f;
        float m2 = n ? ss[i] - s[i] * mean : 0.f;
        sh[0][tid * VEC + i] = (float)n;
        sh[1][tid * VEC + i] = mean;
        sh[2][tid * VEC + i] = m2 > 0.f ? m2 : 0.f;
    }
    __syncthreads();
    int span = 1;
    while (span < g.rpb) span <<= 1;
    for (int st = span >> 1; st >= 1; st >>= 1) {
        if (trow < st && trow + st < g.rpb && vcol < g.cols) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                int a = tid * VEC + i, b = (tid + st * g.cpb) * VEC + i;
                float na = sh[0][a], ma = sh[1][a], qa = sh[2][a];
                float nb = sh[0][b], mb = sh[1][b], qb = sh[2][b];
                if (nb > 0.f) {
                    float nt = na + nb, d = mb - ma, f = nb / nt;
                    ma += d * f;
                    qa += qb + d * d * (na * f);
                    na = nt;
                }
                sh[0][a] = na; sh[1][a] = ma; sh[2][a] = qa;
            }
        }
        __syncthreads();
    }
    if (trow == 0 && vcol < g.cols) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            int c = vcol * VEC + i;
            float* o = partial + ((size_t)blockIdx.x * C + c) * 3;
            o[0] = sh[0][tid * VEC + i]; o[1] = sh[1][tid * VEC + i]; o[2] = sh[2][tid * VEC + i];
        }
    }
}

// Chan merge of (n, mean, M2) pairs
__device__ __forceinline__ void chan_merge(float& na, float& ma, float& qa, float nb, float mb, float qb) {
    if (nb > 0.f) {
        float nt = na + nb, d = mb - ma, f = nb / nt;
        ma += d * f;
        qa += qb + d * d * (na * f);
        na = nt;
    }
}

// Pre-merge for long partial lists (a conv-fused layer has one slice per 128-row tile: 6272 at 56x56,
// bs 256).  Lane = channel, so a wave reads NV*64 consecutive floats per slice (coalesced); the four
// waves of a block take interleaved slices of the block's share; grid = (channel groups, shares).
// NV = 3: Chan merge of (n, mean, M2); NV = 2: plain sums (backward).  Output [shares][C][NV].
template <int NV>
__global__ __launch_bounds__(256) void bn_premerge_kernel(const float* __restrict__ partial, int slices, int C,
                                                          float* __restrict__ out) {
    __shared__ float sh[NV][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int per = (slices + gridDim.y - 1) / gridDim.y;
    const int s0 = blockIdx.y * per, s1 = min(slices, s0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (c < C) {
        int s = s0 + wave;
        for (; s + 12 < s1; s += 16) {                 // four independent loads in flight per lane
            float v[4][NV];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < NV; ++k) v[u][k] = partial[((size_t)(s + 4 * u) * C + c) * NV + k];
            if constexpr (NV == 3) {
                chan_merge(v[0][0], v[0][1], v[0][2], v[1][0], v[1][1], v[1][2]);
                chan_merge(v[2][0], v[2][1], v[2][2], v[3][0], v[3][1], v[3][2]);
                chan_merge(v[0][0], v[0][1], v[0][2], v[2][0], v[2][1], v[2][2]);
                chan_merge(a0, a1, a2, v[0][0], v[0][1], v[0][2]);
            } else {
                a0 += (v[0][0] + v[1][0]) + (v[2][0] + v[3][0]);
                a1 += (v[0][1] + v[1][1]) + (v[2][1] + v[3][1]);
            }
        }
        for (; s < s1; s += 4) {
            const float* o = partial + ((size_t)s * C + c) * NV;
            if constexpr (NV == 3) chan_merge(a0, a1, a2, o[0], o[1], o[2]);
            else { a0 += o[0]; a1 += o[1]; }
        }
    }
    sh[0][wave][lane] = a0;
    sh[1][wave][lane] = a1;
    if constexpr (NV == 3) sh[2][wave][lane] = a2;
    __syncthreads();
    if (wave == 0 && c < C) {
        float* o = out + ((size_t)blockIdx.y * C + c) * NV;
        if constexpr (NV == 3) {
            for (int w = 1; w < 4; ++w) chan_merge(a0, a1, a2, sh[0][w][lane], sh[1][w][lane], sh[2][w][lane]);
            o[0] = a0; o[1] = a1; o[2] = a2;
        } else {
            o[0] = (sh[0][0][lane] + sh[0][1][lane]) + (sh[0][2][lane] + sh[0][3][lane]);
            o[1] = (sh[1][0][lane] + sh[1][1][lane]) + (sh[1][2][lane] + sh[1][3][lane]);
        }
    }
}

// One 256-thread block per channel: threads stride over the slices (a conv-fused layer has one
// slice per 128-row tile, thousands of them), butterfly merge per wave, then across the 4 waves.
__global__ __launch_bounds__(256) void bn_finalize_kernel(
    const float* __restrict__ partial, int slices, int C, int64_t M, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, float* save_mean,
    float* save_invstd, float* scale, float* shift, float eps, float momentum) {
    __shared__ float sh[3][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = blockIdx.x;
    // the channel's parameters, requested by the finalising thread together with its partials: fetched at the end they
    // were four dependent round trips (gamma, beta, running_mean, running_var) of a ~6 us latency-bound launch
    float pg = 0.f, pb = 0.f, prm = 0.f, prv = 0.f;
    if (tid == 0) {
        pg = gamma[c]; pb = beta[c];
        prm = (running_mean ? running_mean : gamma)[c];
        prv = (running_var ? running_var : gamma)[c];
    }
    float na = 0.f, ma = 0.f, qa = 0.f;
    int s = tid;
    for (; s + 768 < slices; s += 1024) {       // four independent loads in flight per thread
        const float* o0 = partial + ((size_t)s * C + c) * 3;
        const float* o1 = o0 + (size_t)256 * C * 3;
        const float* o2 = o1 + (size_t)256 * C * 3;
        const float* o3 = o2 + (size_t)256 * C * 3;
        float n0 = o0[0], m0 = o0[1], q0 = o0[2], n1 = o1[0], m1 = o1[1], q1 = o1[2];
        float n2 = o2[0], m2 = o2[1], q2 = o2[2], n3 = o3[0], m3 = o3[1], q3 = o3[2];
        chan_merge(n0, m0, q0, n1, m1, q1);
        chan_merge(n2, m2, q2, n3, m3, q3);
        chan_merge(n0, m0, q0, n2, m2, q2);
        chan_merge(na, ma, qa, n0, m0, q0);
    }
    for (; s < slices; s += 256) {
        const float* o = partial + ((size_t)s * C + c) * 3;
        chan_merge(na, ma, qa, o[0], o[1], o[2]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float nb = __shfl_xor(na, o, 64), mb = __shfl_xor(ma, o, 64), qb = __shfl_xor(qa, o, 64);
        chan_merge(na, ma, qa, nb, mb, qb);
    }
    if (lane == 0) { sh[0][wave] = na; sh[1][wave] = ma; sh[2][wave] = qa; }
    __syncthreads();
    if (tid != 0) return;
    for (int w = 1; w < 4; ++w) chan_merge(na, ma, qa, sh[0][w], sh[1][w], sh[2][w]);
    float var = qa / (float)M;
    float invstd = rsqrtf(var + eps);
    // one Newton step: rsqrtf is ~1 ulp, this keeps fp32 parity tight
    invstd = invstd * (1.5f - 0.5f * (var + eps) * invstd * invstd);
    save_mean[c] = ma;
    save_invstd[c] = invstd;
    if (running_mean) {
        float unb = M > 1 ? qa / (float)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * prm + momentum * ma;
        running_var[c] = (1.f - momentum) * prv + momentum * unb;
    }
    float sc = pg * invstd;
    scale[c] = sc;
    shift[c] = pb - ma * sc;
}

// ---- merge + finalisation of a long partial list in ONE launch (round 3) -----------------------------------------------
// bn_premerge + bn_finalize (or bn_bwd_final) were two dependent ~6 us launches in front of every BN streaming pass whose
// partials come from a conv epilogue with >= 1024 row tiles.  Here the merge blocks publish their share with write-through
// (sc1) stores, drain them, and draw a ticket; the block that draws the last ticket of its 64-channel group merges the
// shares in share order (fixed summation order: the result does not depend on which block is last) and finalises.
// No release fence anywhere (MI355X_MICROARCH.md: an agent-scope release is an L2 write-back -- round 2's attempt with
// __threadfence() in every block of the STREAMING pass lost 1.4 ms per step); the last block of a group runs one acquire
// (L1 invalidate of its own CU) and reads the shares with sc1 loads.  cdna_hip_programming.md 5, "Projection GEMM" item 2.
// Tickets live in a __device__ array (zero at module load; the last block resets its counter for the slot's next user).  The
// 128 slots are 32 BANKS of 4: a bank belongs to ONE stream (host table, first come first served), so two launches that can be
// in flight together -- different streams -- never share a counter, and launches of one stream, which execute in order, walk
// their bank round-robin.  Once every bank is claimed, a new stream gets the least recently used bank of an idle stream
// (bnf_slot_for), else the two-kernel path.  (the slots used to be handed out from one
// process-wide counter, so launches 64 issues apart on two streams could have met on one counter.)  What is NOT covered: two
// captured graphs holding launches of the same bank replayed CONCURRENTLY on two streams -- include/cnk.h says so.
constexpr int BNF_BANKS = 32, BNF_PER_BANK = 4, BNF_SLOTS = BNF_BANKS * BNF_PER_BANK, BNF_GROUPS = 64;
__device__ unsigned int bn_tickets[BNF_SLOTS * BNF_GROUPS];

struct BnFinalFwd {
    static constexpr int NV = 3;
    const float* gamma; const float* beta;
    float* running_mean; float* running_var; float* save_mean; float* save_invstd; float* scale; float* shift;
    float eps, momentum;
    struct Pre { float g, b, rm, rv; };
    // the channel's parameters, requested together with the shares (not one dependent round trip each at the end)
    __device__ __forceinline__ Pre pre(int c) const {
        Pre q;
        q.g = gamma[c]; q.b = beta[c];
        const float* rmp = running_mean ? running_mean : gamma;      // unconditional loads (always-valid address)
        const float* rvp = running_var ? running_var : gamma;
        q.rm = rmp[c]; q.rv = rvp[c];
        return q;
    }
    __device__ __forceinline__ void operator()(int c, int64_t M, const Pre& q, float /*n*/, float ma, float qa) const {
        float var = qa / (float)M;
        float invstd = rsqrtf(var + eps);
        invstd = invstd * (1.5f - 0.5f * (var + eps) * invstd * invstd);      // as bn_finalize_kernel
        save_mean[c] = ma;
        save_invstd[c] = invstd;
        if (running_mean) {
            float unb = M > 1 ? qa / (float)(M - 1) : var;
            running_mean[c] = (1.f - momentum) * q.rm + momentum * ma;
            running_var[c] = (1.f - momentum) * q.rv + momentum * unb;
        }
        float sc = q.g * invstd;
        scale[c] = sc;
        shift[c] = q.b - ma * sc;
    }
};
struct BnFinalBwd {
    static constexpr int NV = 2;
    const float* gamma; const float* invstd;
    float* dgamma; float* dbeta; float* coefA; float* coefB1; float* coefB2;
    int accumulate;
    struct Pre { float g, is, dg, db; };
    __device__ __forceinline__ Pre pre(int c) const {
        Pre q;
        q.g = gamma[c]; q.is = invstd[c]; q.dg = dgamma[c]; q.db = dbeta[c];      // dgamma / dbeta are valid buffers either way
        return q;
    }
    __device__ __forceinline__ void operator()(int c, int64_t M, const Pre& q, float a, float b, float) const {
        dbeta[c] = accumulate ? q.db + a : a;
        dgamma[c] = accumulate ? q.dg + b : b;
        coefA[c] = q.g * q.is;
        coefB1[c] = a / (float)M;
        coefB2[c] = b / (float)M;
    }
};

template <class FIN>
__global__ __launch_bounds__(256) void bn_merge_final_kernel(const float* __restrict__ partial, int slices, int C, int64_t M,
                                                             float* __restrict__ out, int slot, const FIN fin) {
    constexpr int NV = FIN::NV;
    __shared__ float sh[NV][4][64];
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int shares = gridDim.y;
    const int per = (slices + shares - 1) / shares;
    const int s0 = blockIdx.y * per, s1 = min(slices, s0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (c < C) {
        int s = s0 + wave;
        for (; s + 28 < s1; s += 32) {                 // eight independent records in flight per lane
            // one 8- / 12-byte load per record (records are 4-byte aligned: the vector type says so)
            typedef float RecA __attribute__((ext_vector_type(NV == 3 ? 3 : 2), aligned(4)));
            float v[8][NV];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const RecA r = *(const RecA*)(partial + ((size_t)(s + 4 * u) * C + c) * NV);
#pragma unroll
                for (int k = 0; k < NV; ++k) v[u][k] = r[k];
            }
            if constexpr (NV == 3) {
#pragma unroll
                for (int u = 0; u < 8; u += 2) chan_merge(v[u][0], v[u][1], v[u][2], v[u + 1][0], v[u + 1][1], v[u + 1][2]);
                chan_merge(v[0][0], v[0][1], v[0][2], v[2][0], v[2][1], v[2][2]);
                chan_merge(v[4][0], v[4][1], v[4][2], v[6][0], v[6][1], v[6][2]);
                chan_merge(v[0][0], v[0][1], v[0][2], v[4][0], v[4][1], v[4][2]);
                chan_merge(a0, a1, a2, v[0][0], v[0][1], v[0][2]);
            } else {
                a0 += ((v[0][0] + v[1][0]) + (v[2][0] + v[3][0])) + ((v[4][0] + v[5][0]) + (v[6][0] + v[7][0]));
                a1 += ((v[0][1] + v[1][1]) + (v[2][1] + v[3][1])) + ((v[4][1] + v[5][1]) + (v[6][1] + v[7][1]));
            }
        }
        for (; s < s1; s += 4) {
            const float* o = partial + ((size_t)s * C + c) * NV;
            if constexpr (NV == 3) chan_merge(a0, a1, a2, o[0], o[1], o[2]);
            else { a0 += o[0]; a1 += o[1]; }
        }
    }
    sh[0][wave][lane] = a0;
    sh[1][wave][lane] = a1;
    if constexpr (NV == 3) sh[2][wave][lane] = a2;
    __syncthreads();
    if (wave == 0 && c < C) {
        if constexpr (NV == 3) {
            for (int w = 1; w < 4; ++w) chan_merge(a0, a1, a2, sh[0][w][lane], sh[1][w][lane], sh[2][w][lane]);
        } else {
            a0 = (sh[0][0][lane] + sh[0][1][lane]) + (sh[0][2][lane] + sh[0][3][lane]);
            a1 = (sh[1][0][lane] + sh[1][1][lane]) + (sh[1][2][lane] + sh[1][3][lane]);
        }
        // publish: write-through stores (global_store_dword sc1), visible device-wide once drained
        float* o = out + ((size_t)blockIdx.y * C + c) * NV;
        __hip_atomic_store(o + 0, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 1, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (NV == 3) __hip_atomic_store(o + 2, a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores ...
    __syncthreads();                                          // ... before the one lane that signals for all of them
    unsigned int* ticket = bn_tickets + slot * BNF_GROUPS + blockIdx.x;
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = t == (unsigned)shares - 1u;
        if (is_last) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the slot's next launch
            // (L1 invalidate of this CU; it completes asynchronously -- the loads below bypass L1 anyway (sc1) and the
            // vmcnt(0) that retires them also covers it: nobody waits for it alone)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    if (!is_last) return;
    // the last block of this channel group: merge the shares in share order (wave w: shares w, w+4, ...).  One write-through-
    // coherent (sc1) 8- / 12-byte load per share, all of a wave's <= 16 shares in flight together, retired by one vmcnt(0).
    a0 = a1 = a2 = 0.f;
    typename FIN::Pre q = {};
    if (wave == 0 && c < C) q = fin.pre(c);
    {
        typedef __attribute__((ext_vector_type(3))) float f32x3;
        typedef typename std::conditional<NV == 3, f32x3, f32x2>::type Rec;
        Rec v[16];
        const int cc = c < C ? c : C - 1;                                      // lanes past C read a valid record, use none
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int ss = wave + 4 * u, sc = ss < shares ? ss : shares - 1;       // unconditional loads (clamped address)
            const float* src = out + ((size_t)sc * C + cc) * NV;
            if constexpr (NV == 3) asm volatile("global_load_dwordx3 %0, %1, off sc1" : "=v"(v[u]) : "v"(src) : "memory");
            else asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v[u]) : "v"(src) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                       "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                     :: "memory");
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const bool on = wave + 4 * u < shares;
            if constexpr (NV == 3) chan_merge(a0, a1, a2, on ? v[u][0] : 0.f, v[u][1], v[u][2]);      // n == 0: no-op
            else { a0 += on ? v[u][0] : 0.f; a1 += on ? v[u][1] : 0.f; }
        }
    }
    sh[0][wave][lane] = a0;
    sh[1][wave][lane] = a1;
    if constexpr (NV == 3) sh[2][wave][lane] = a2;
    __syncthreads();
    if (wave != 0 || c >= C) return;
    if constexpr (NV == 3) {
        for (int w = 1; w < 4; ++w) chan_merge(a0, a1, a2, sh[0][w][lane], sh[1][w][lane], sh[2][w][lane]);
    } else {
        a0 = (sh[0][0][lane] + sh[0][1][lane]) + (sh[0][2][lane] + sh[0][3][lane]);
        a1 = (sh[1][0][lane] + sh[1][1][lane]) + (sh[1][2][lane] + sh[1][3][lane]);
    }
    fin(c, M, q, a0, a1, a2);
}

__global__ void bn_eval_coef_kernel(const float* gamma, const float* beta, const float* rmean,
                                    const float* rvar, float* scale, float* shift, int C, float eps) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sc = gamma[c] / sqrtf(rvar[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

// ---- streaming passes ---------------------------------------------------------------------------
// All three streaming kernels share one shape: a thread owns VEC channels (one 16-byte vector) for
// all its rows, so per-channel coefficients live in registers and the loop body is payload loads,
// arithmetic and payload stores only.  Every mode (residual operand, ReLU, mask source, optional
// outputs) is a template parameter: the main loop has no branches, issues the loads of ROWS_U rows
// back to back and only then consumes them, which keeps ROWS_U x streams x 16 B in flight per lane.
constexpr int ROWS_U = 4;

template <int VEC> __device__ __forceinline__ void load_coef(float (&d)[VEC], const float* __restrict__ p) {
    if constexpr (VEC % 4 == 0) {
        if (((uintptr_t)p & 15) == 0) {              // wave-uniform: the arrays of one launch share alignment
#pragma unroll
            for (int i = 0; i < VEC; i += 4) {
                f32x4 v = *(const f32x4*)(p + i);
                d[i] = v[0]; d[i + 1] = v[1]; d[i + 2] = v[2]; d[i + 3] = v[3];
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) d[i] = p[i];
}

template <typename T, int VEC> struct Raw {          // one thread's 16-byte payload (or one element)
    using V = typename std::conditional<VEC == 1, T, decltype(Vec16<T>::v)>::type;
    V v;
    __device__ __forceinline__ void load(const T* p) { v = *(const V*)p; }
    __device__ __forceinline__ void store(T* p) const { *(V*)p = v; }
    __device__ __forceinline__ float get(int i) const {
        if constexpr (VEC == 1) return to_f32(v); else return to_f32(v[i]);
    }
    __device__ __forceinline__ void set(int i, float f) {
        if constexpr (VEC == 1) v = from_f32<T>(f); else v[i] = from_f32<T>(f);
    }
};

// y = act(x*scale + shift [+ res]); optional packed ReLU mask (one byte per vector: bit k = y_k > 0)
// RAFF: the residual operand is the INPUT of another training-mode BN layer without activation (a projection shortcut,
// resnet.py:90-91) whose apply pass was deferred into this one: its value is round_T(r * rsc + rsh), exactly what that layer's
// own apply pass would have stored, so the result is bit-identical and one full read + write pass of the block is gone.
template <typename T, int VEC, bool HAS_RES, bool RELU, bool WMASK, bool RAFF = false>
__device__ __forceinline__ void bn_apply_row(const Raw<T, VEC>& xv, const Raw<T, VEC>& rv, const float (&sc)[VEC],
                                             const float (&sh)[VEC], T* __restrict__ yp,
                                             unsigned char* __restrict__ mp, const float (&rsc)[VEC], const float (&rsh)[VEC]) {
    Raw<T, VEC> o;
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        float a = xv.get(k) * sc[k] + sh[k];
        if constexpr (HAS_RES && RAFF) a += to_f32(from_f32<T>(rv.get(k) * rsc[k] + rsh[k]));
        else if constexpr (HAS_RES) a += rv.get(k);
        if constexpr (RELU) a = a > 0.f ? a : 0.f;
        if constexpr (WMASK) bits |= (a > 0.f ? 1u : 0u) << k;
        o.set(k, a);
    }
    o.store(yp);
    if constexpr (WMASK) *mp = (unsigned char)bits;
}

template <typename T, int VEC, bool HAS_RES, bool RELU, bool WMASK, bool RAFF = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x,
                                                       const T* __restrict__ res, T* __restrict__ y,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       unsigned char* __restrict__ mask, int64_t M,
                                                       int C, SliceGeom g, const float* __restrict__ raff) {
    const int tid = threadIdx.x, tcol = tid % g.cpb, trow = tid / g.cpb;
    const int vcol = blockIdx.y * g.cpb + tcol;
    if (trow >= g.rpb || vcol >= g.cols) return;
    const int64_t m0 = (int64_t)blockIdx.x * g.slice_rows;
    const int64_t m1 = m0 + g.slice_rows < M ? m0 + g.slice_rows : M;
    const int64_t rstep = (int64_t)g.rpb * C;
    int64_t m = m0 + trow;
    int64_t off = m * C + (int64_t)vcol * VEC;
    float sc[VEC], sh[VEC], rsc[VEC], rsh[VEC];
    load_coef<VEC>(sc, scale + vcol * VEC);
    load_coef<VEC>(sh, shift + vcol * VEC);
    if constexpr (RAFF) {
        load_coef<VEC>(rsc, raff + vcol * VEC);
        load_coef<VEC>(rsh, raff + C + vcol * VEC);
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) rsc[k] = rsh[k] = 0.f;
    }
    for (; m + (ROWS_U - 1) * g.rpb < m1; m += ROWS_U * g.rpb, off += ROWS_U * rstep) {
        Raw<T, VEC> xv[ROWS_U], rv[ROWS_U];
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) {
            xv[u].load(x + off + u * rstep);
            if constexpr (HAS_RES) rv[u].load(res + off + u * rstep);
        }
        // keep the ROWS_U rows' loads together IN FRONT of the arithmetic: hipcc's occupancy-driven scheduler otherwise sinks
        // each load to its use to save registers (67-76 VGPRs, one or two loads in flight per lane; round 3, from the .s)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u)
            bn_apply_row<T, VEC, HAS_RES, RELU, WMASK, RAFF>(xv[u], rv[u], sc, sh, y + off + u * rstep,
                                                             mask + (m + (int64_t)u * g.rpb) * g.cols + vcol, rsc, rsh);
    }
    for (; m < m1; m += g.rpb, off += rstep) {
        Raw<T, VEC> xv, rv;
        xv.load(x + off);
        if constexpr (HAS_RES) rv.load(res + off);
        bn_apply_row<T, VEC, HAS_RES, RELU, WMASK, RAFF>(xv, rv, sc, sh, y + off, mask + m * g.cols + vcol, rsc, rsh);
    }
}

// eval without workspace: coefficients recomputed per element from the running statistics
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_eval_kernel(const T* __restrict__ x,
                                                      const T* __restrict__ res, T* __restrict__ y,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      const float* __restrict__ rmean,
                                                      const float* __restrict__ rvar, int64_t nvec,
                                                      int C, float eps, int relu) {
    const int cols = C / VEC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (int64_t)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cols) * VEC;
        VLoad<T, VEC> v, r;
        v.load(x + i * VEC);
        if (res) r.load(res + i * VEC);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            int c = c0 + k;
            float sc = gamma[c] / sqrtf(rvar[c] + eps);
            float o = (v.f[k] - rmean[c]) * sc + beta[c];
            if (res) o += r.f[k];
            if (relu) o = o > 0.f ? o : 0.f;
            v.f[k] = o;
        }
        v.store(y + i * VEC);
    }
}

// Where the backward takes the ReLU mask from: nothing (no ReLU), re-derived from x*scale+shift
// (no residual operand), the packed bit mask the forward saved, or the forward output y.
enum { ACT_NONE = 0, ACT_REMASK = 1, ACT_BITS = 2, ACT_Y = 3 };

// masked gradient d' of one vector; `aux` is the mask byte (ACT_BITS) -- yv the forward output (ACT_Y)
template <typename T, int VEC, int ACT>
__device__ __forceinline__ void bn_masked(const Raw<T, VEC>& dv, const Raw<T, VEC>& xv, const Raw<T, VEC>& yv,
                                          unsigned aux, const float (&sc)[VEC], const float (&sf)[VEC],
                                          float (&d)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        float v = dv.get(i);
        if constexpr (ACT == ACT_REMASK) { if (!(xv.get(i) * sc[i] + sf[i] > 0.f)) v = 0.f; }
        if constexpr (ACT == ACT_BITS) { if (!((aux >> i) & 1u)) v = 0.f; }
        if constexpr (ACT == ACT_Y) { if (!(yv.get(i) > 0.f)) v = 0.f; }
        d[i] = v;
    }
}

template <typename T, int VEC, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
    const unsigned char* __restrict__ mask, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, float* __restrict__ partial, int64_t M, int C, SliceGeom g) {
    __shared__ float sh[2][256 * (VEC > 1 ? VEC : 1)];
    const int tid = threadIdx.x, tcol = tid % g.cpb, trow = tid / g.cpb;
    const int vcol = blockIdx.y * g.cpb + tcol;
    const bool active = trow < g.rpb && vcol < g.cols;
    const int64_t m0 = (int64_t)blockIdx.x * g.slice_rows;
    const int64_t m1 = m0 + g.slice_rows < M ? m0 + g.slice_rows : M;
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;
    if (active) {
        float mu[VEC], sc[VEC], sf[VEC];
        load_coef<VEC>(mu, mean + vcol * VEC);
        if constexpr (ACT == ACT_REMASK) {
            float is[VEC];
            load_coef<VEC>(is, invstd + vcol * VEC);
            load_coef<VEC>(sc, gamma + vcol * VEC);
            load_coef<VEC>(sf, beta + vcol * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) { sc[i] *= is[i]; sf[i] -= mu[i] * sc[i]; }
        }
        const int64_t rstep = (int64_t)g.rpb * C;
        int64_t m = m0 + trow;
        int64_t off = m * C + (int64_t)vcol * VEC;
        // s2 accumulates d'*(x - mu); the common factor invstd is applied once at the end
        for (; m + (ROWS_U - 1) * g.rpb < m1; m += ROWS_U * g.rpb, off += ROWS_U * rstep) {
            Raw<T, VEC> dv[ROWS_U], xv[ROWS_U], yv[ROWS_U];
            unsigned aux[ROWS_U];
#pragma unroll
            for (int u = 0; u < ROWS_U; ++u) {
                dv[u].load(dy + off + u * rstep);
                xv[u].load(x + off + u * rstep);
                if constexpr (ACT == ACT_BITS) aux[u] = mask[(m + (int64_t)u * g.rpb) * g.cols + vcol];
                if constexpr (ACT == ACT_Y) yv[u].load(y + off + u * rstep);
            }
#pragma unroll
            for (int u = 0; u < ROWS_U; ++u) {
                float d[VEC];
                bn_masked<T, VEC, ACT>(dv[u], xv[u], yv[u], aux[u], sc, sf, d);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    s1[i] += d[i];
                    s2[i] += d[i] * (xv[u].get(i) - mu[i]);
                }
            }
        }
        for (; m < m1; m += g.rpb, off += rstep) {
            Raw<T, VEC> dv, xv, yv;
            unsigned aux = 0;
            dv.load(dy + off);
            xv.load(x + off);
            if constexpr (ACT == ACT_BITS) aux = mask[m * g.cols + vcol];
            if constexpr (ACT == ACT_Y) yv.load(y + off);
            float d[VEC];
            bn_masked<T, VEC, ACT>(dv, xv, yv, aux, sc, sf, d);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                s1[i] += d[i];
                s2[i] += d[i] * (xv.get(i) - mu[i]);
            }
        }
        float is[VEC];
        load_coef<VEC>(is, invstd + vcol * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s2[i] *= is[i];
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        sh[0][tid * VEC + i] = s1[i];
        sh[1][tid * VEC + i] = s2[i];
    }
    __syncthreads();
    int span = 1;
    while (span < g.rpb) span <<= 1;
    for (int st = span >> 1; st >= 1; st >>= 1) {
        if (trow < st && trow + st < g.rpb && vcol < g.cols) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                int a = tid * VEC + i, b = (tid + st * g.cpb) * VEC + i;
                sh[0][a] += sh[0][b];
                sh[1][a] += sh[1][b];
            }
        }
        __syncthreads();
    }
    if (trow == 0 && vcol < g.cols) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            int c = vcol * VEC + i;
            float* o = partial + ((size_t)blockIdx.x * C + c) * 2;
            o[0] = sh[0][tid * VEC + i]; o[1] = sh[1][tid * VEC + i];
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_final_kernel(
    const float* __restrict__ partial, int slices, int C, int64_t M, const float* __restrict__ gamma,
    const float* __restrict__ invstd, float* dgamma, float* dbeta, float* coefA, float* coefB1,
    float* coefB2, int accumulate) {
    // one 256-thread block per channel, fixed summation order (deterministic)
    __shared__ float sh[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = blockIdx.x;
    float pg = 0.f, pis = 0.f, pdb = 0.f, pdg = 0.f;          // prefetched by the finalising thread (see bn_finalize_kernel)
    if (tid == 0) { pg = gamma[c]; pis = invstd[c]; pdb = dbeta[c]; pdg = dgamma[c]; }
    float a = 0.f, b = 0.f;
    int s = tid;
    for (; s + 768 < slices; s += 1024) {
        const float* o0 = partial + ((size_t)s * C + c) * 2;
        const float* o1 = o0 + (size_t)256 * C * 2;
        const float* o2 = o1 + (size_t)256 * C * 2;
        const float* o3 = o2 + (size_t)256 * C * 2;
        float a0 = o0[0], b0 = o0[1], a1 = o1[0], b1 = o1[1], a2 = o2[0], b2 = o2[1], a3 = o3[0], b3 = o3[1];
        a += (a0 + a1) + (a2 + a3);
        b += (b0 + b1) + (b2 + b3);
    }
    for (; s < slices; s += 256) {
        const float* o = partial + ((size_t)s * C + c) * 2;
        a += o[0];
        b += o[1];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) { sh[0][wave] = a; sh[1][wave] = b; }
    __syncthreads();
    if (tid != 0) return;
    a = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    b = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    dbeta[c] = accumulate ? pdb + a : a;
    dgamma[c] = accumulate ? pdg + b : b;
    coefA[c] = pg * pis;
    coefB1[c] = a / (float)M;
    coefB2[c] = b / (float)M;
}

// dx = a*(d' - b1 - (x-mu)*(is*b2)); the centred form keeps fp32 accuracy when |mu| >> sigma
template <typename T, int VEC, int ACT, bool WRES>
__device__ __forceinline__ void bn_bwd_row(const Raw<T, VEC>& dv, const Raw<T, VEC>& xv, const Raw<T, VEC>& yv,
                                           unsigned aux, const float (&ka)[VEC], const float (&kb1)[VEC],
                                           const float (&kmu)[VEC], const float (&kis)[VEC],
                                           const float (&ksf)[VEC], T* __restrict__ dxp, T* __restrict__ drp) {
    float d[VEC];
    bn_masked<T, VEC, ACT>(dv, xv, yv, aux, ka, ksf, d);
    Raw<T, VEC> o, r;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        o.set(k, ka[k] * (d[k] - kb1[k] - (xv.get(k) - kmu[k]) * kis[k]));
        if constexpr (WRES) r.set(k, d[k]);
    }
    o.store(dxp);
    if constexpr (WRES) r.store(drp);
}

template <typename T, int VEC, int ACT, bool WRES>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ dx,
    T* __restrict__ dres, const unsigned char* __restrict__ mask, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ coefA, const float* __restrict__ coefB1,
    const float* __restrict__ coefB2, int64_t M, int C, SliceGeom g) {
    const int tid = threadIdx.x, tcol = tid % g.cpb, trow = tid / g.cpb;
    const int vcol = blockIdx.y * g.cpb + tcol;
    if (trow >= g.rpb || vcol >= g.cols) return;
    const int64_t m0 = (int64_t)blockIdx.x * g.slice_rows;
    const int64_t m1 = m0 + g.slice_rows < M ? m0 + g.slice_rows : M;
    const int64_t rstep = (int64_t)g.rpb * C;
    int64_t m = m0 + trow;
    int64_t off = m * C + (int64_t)vcol * VEC;
    float ka[VEC], kb1[VEC], kmu[VEC], kis[VEC], ksf[VEC];
    load_coef<VEC>(ka, coefA + vcol * VEC);             // gamma*invstd == the forward's scale
    load_coef<VEC>(kb1, coefB1 + vcol * VEC);
    load_coef<VEC>(kmu, mean + vcol * VEC);
    load_coef<VEC>(kis, invstd + vcol * VEC);
    {
        float b2[VEC];
        load_coef<VEC>(b2, coefB2 + vcol * VEC);
#pragma unroll
        for (int k = 0; k < VEC; ++k) kis[k] *= b2[k];
    }
    if constexpr (ACT == ACT_REMASK) {
        load_coef<VEC>(ksf, beta + vcol * VEC);
#pragma unroll
        for (int k = 0; k < VEC; ++k) ksf[k] -= kmu[k] * ka[k];
    }
    for (; m + (ROWS_U - 1) * g.rpb < m1; m += ROWS_U * g.rpb, off += ROWS_U * rstep) {
        Raw<T, VEC> dv[ROWS_U], xv[ROWS_U], yv[ROWS_U];
        unsigned aux[ROWS_U];
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) {
            dv[u].load(dy + off + u * rstep);
            xv[u].load(x + off + u * rstep);
            if constexpr (ACT == ACT_BITS) aux[u] = mask[(m + (int64_t)u * g.rpb) * g.cols + vcol];
            if constexpr (ACT == ACT_Y) yv[u].load(y + off + u * rstep);
        }
        __builtin_amdgcn_sched_barrier(0);          // loads of all ROWS_U rows stay in front of the arithmetic (see bn_apply_kernel)
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u)
            bn_bwd_row<T, VEC, ACT, WRES>(dv[u], xv[u], yv[u], aux[u], ka, kb1, kmu, kis, ksf,
                                          dx + off + u * rstep, dres + off + u * rstep);
    }
    for (; m < m1; m += g.rpb, off += rstep) {
        Raw<T, VEC> dv, xv, yv;
        unsigned aux = 0;
        dv.load(dy + off);
        xv.load(x + off);
        if constexpr (ACT == ACT_BITS) aux = mask[m * g.cols + vcol];
        if constexpr (ACT == ACT_Y) yv.load(y + off);
        bn_bwd_row<T, VEC, ACT, WRES>(dv, xv, yv, aux, ka, kb1, kmu, kis, ksf, dx + off, dres + off);
    }
}

// ---- BN backward fed by a max-pool backward (the stems: conv -> BN -> ReLU -> MaxPool2d) -----------------------------
// The gradient of the BN output is never materialised: both passes gather it from the POOLED gradient and the saved
// arg-max positions (<= 2 x 2 windows cover an element when ceil(k/s) <= 2), summing in the window order
// maxpool_bwd_kernel uses.  Against cnk_maxpool2d_bwd + cnk_bn_bwd that removes one full-size write and two full-size
// reads (the stem of ResNet-50 at batch 256: 411 MB each) for two gathers of the 4x smaller pooled tensors.
struct PoolGeom {
    int H, W, Ho, Wo, kh, kw, sh, sw, ph, pw;
    FastDiv fd_w, fd_h;
};

template <typename T, int VEC> struct PoolGather {
    using IB = typename std::conditional<VEC == 8, u32x2, typename std::conditional<VEC == 4, unsigned int, unsigned char>::type>::type;
    Raw<T, VEC> g[4];
    IB b[4];
    int pos[4];          // position of this element inside window k (row-major), -1: no such window
    __device__ __forceinline__ void issue(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                          const PoolGeom& p, uint32_t m, int vcol, int C) {
        const uint32_t t = fd_div(m, p.fd_w), w = m - t * p.W;
        const uint32_t n = fd_div(t, p.fd_h), h = t - n * p.H;
        int ho_lo = ((int)h + p.ph - p.kh + p.sh) / p.sh;          // ceil((h+p-k+1)/s) when non-negative
        if ((int)h + p.ph - p.kh + 1 <= 0) ho_lo = 0;
        int ho_hi = ((int)h + p.ph) / p.sh;
        if (ho_hi > p.Ho - 1) ho_hi = p.Ho - 1;
        int wo_lo = ((int)w + p.pw - p.kw + p.sw) / p.sw;
        if ((int)w + p.pw - p.kw + 1 <= 0) wo_lo = 0;
        int wo_hi = ((int)w + p.pw) / p.sw;
        if (wo_hi > p.Wo - 1) wo_hi = p.Wo - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ho = ho_lo + (k >> 1), wo = wo_lo + (k & 1);
            const bool ok = ho <= ho_hi && wo <= wo_hi;
            pos[k] = ok ? ((int)h - (ho * p.sh - p.ph)) * p.kw + ((int)w - (wo * p.sw - p.pw)) : -1;
            b[k] = IB{};
            if (ok) {
                const size_t o = ((size_t)(n * p.Ho + ho) * p.Wo + wo) * C + (size_t)vcol * VEC;
                g[k].load(dy + o);
                b[k] = *(const IB*)(idx + o);
            }
        }
    }
    __device__ __forceinline__ void sum(float (&d)[VEC]) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) d[i] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned char by[VEC];
            __builtin_memcpy(by, &b[k], VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if ((int)by[i] == pos[k]) d[i] += g[k].get(i);
        }
    }
};

constexpr int POOL_ROWS_U = 2;

// slices of one launch are walked XCD by XCD (blocks are dealt to the 8 XCDs round-robin): neighbouring slices share
// the pooled rows at their seam, which then hit in that XCD's L2
__device__ __forceinline__ int pool_slice_of_block(int bx, int slices) {
    const int s8 = slices & ~7;
    return bx < s8 ? (bx & 7) * (s8 >> 3) + (bx >> 3) : bx;
}

template <typename T, int VEC, bool RELU>
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(
    const T* __restrict__ dy, const unsigned char* __restrict__ idx, const T* __restrict__ x,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, float* __restrict__ partial, int64_t M, int C, SliceGeom g, PoolGeom pg) {
    __shared__ float sh[2][256 * (VEC > 1 ? VEC : 1)];
    const int tid = threadIdx.x, tcol = tid % g.cpb, trow = tid / g.cpb;
    const int vcol = blockIdx.y * g.cpb + tcol;
    const bool active = trow < g.rpb && vcol < g.cols;
    const int slice = pool_slice_of_block(blockIdx.x, g.slices);
    const int64_t m0 = (int64_t)slice * g.slice_rows;
    const int64_t m1 = m0 + g.slice_rows < M ? m0 + g.slice_rows : M;
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;
    if (active) {
        float mu[VEC], sc[VEC], sf[VEC], is[VEC];
        load_coef<VEC>(mu, mean + vcol * VEC);
        load_coef<VEC>(is, invstd + vcol * VEC);
        if constexpr (RELU) {
            load_coef<VEC>(sc, gamma + vcol * VEC);
            load_coef<VEC>(sf, beta + vcol * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) { sc[i] *= is[i]; sf[i] -= mu[i] * sc[i]; }
        }
        const int64_t rstep = (int64_t)g.rpb * C;
        int64_t m = m0 + trow;
        int64_t off = m * C + (int64_t)vcol * VEC;
        auto consume = [&](const PoolGather<T, VEC>& pv, const Raw<T, VEC>& xv) {
            float d[VEC];
            pv.sum(d);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float xi = xv.get(i);
                if constexpr (RELU) { if (!(xi * sc[i] + sf[i] > 0.f)) d[i] = 0.f; }
                s1[i] += d[i];
                s2[i] += d[i] * (xi - mu[i]);
            }
        };
        for (; m + (POOL_ROWS_U - 1) * g.rpb < m1; m += POOL_ROWS_U * g.rpb, off += POOL_ROWS_U * rstep) {
            PoolGather<T, VEC> pv[POOL_ROWS_U];
            Raw<T, VEC> xv[POOL_ROWS_U];
#pragma unroll
            for (int u = 0; u < POOL_ROWS_U; ++u) {
                xv[u].load(x + off + u * rstep);
                pv[u].issue(dy, idx, pg, (uint32_t)(m + (int64_t)u * g.rpb), vcol, C);
            }
#pragma unroll
            for (int u = 0; u < POOL_ROWS_U; ++u) consume(pv[u], xv[u]);
        }
        for (; m < m1; m += g.rpb, off += rstep) {
            PoolGather<T, VEC> pv;
            Raw<T, VEC> xv;
            xv.load(x + off);
            pv.issue(dy, idx, pg, (uint32_t)m, vcol, C);
            consume(pv, xv);
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) s2[i] *= is[i];
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        sh[0][tid * VEC + i] = s1[i];
        sh[1][tid * VEC + i] = s2[i];
    }
    __syncthreads();
    int span = 1;
    while (span < g.rpb) span <<= 1;
    for (int st = span >> 1; st >= 1; st >>= 1) {
        if (trow < st && trow + st < g.rpb && vcol < g.cols) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                int a = tid * VEC + i, b = (tid + st * g.cpb) * VEC + i;
                sh[0][a] += sh[0][b];
                sh[1][a] += sh[1][b];
            }
        }
        __syncthreads();
    }
    if (trow == 0 && vcol < g.cols) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            int c = vcol * VEC + i;
            float* o = partial + ((size_t)slice * C + c) * 2;
            o[0] = sh[0][tid * VEC + i]; o[1] = sh[1][tid * VEC + i];
        }
    }
}

template <typename T, int VEC, bool RELU>
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(
    const T* __restrict__ dy, const unsigned char* __restrict__ idx, const T* __restrict__ x, T* __restrict__ dx,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ coefA, const float* __restrict__ coefB1, const float* __restrict__ coefB2, int64_t M,
    int C, SliceGeom g, PoolGeom pg) {
    const int tid = threadIdx.x, tcol = tid % g.cpb, trow = tid / g.cpb;
    const int vcol = blockIdx.y * g.cpb + tcol;
    if (trow >= g.rpb || vcol >= g.cols) return;
    const int slice = pool_slice_of_block(blockIdx.x, g.slices);
    const int64_t m0 = (int64_t)slice * g.slice_rows;
    const int64_t m1 = m0 + g.slice_rows < M ? m0 + g.slice_rows : M;
    const int64_t rstep = (int64_t)g.rpb * C;
    int64_t m = m0 + trow;
    int64_t off = m * C + (int64_t)vcol * VEC;
    float ka[VEC], kb1[VEC], kmu[VEC], kis[VEC], ksf[VEC];
    load_coef<VEC>(ka, coefA + vcol * VEC);             // gamma*invstd == the forward's scale
    load_coef<VEC>(kb1, coefB1 + vcol * VEC);
    load_coef<VEC>(kmu, mean + vcol * VEC);
    load_coef<VEC>(kis, invstd + vcol * VEC);
    {
        float b2[VEC];
        load_coef<VEC>(b2, coefB2 + vcol * VEC);
#pragma unroll
        for (int k = 0; k < VEC; ++k) kis[k] *= b2[k];
    }
    if constexpr (RELU) {
        load_coef<VEC>(ksf, beta + vcol * VEC);
#pragma unroll
        for (int k = 0; k < VEC; ++k) ksf[k] -= kmu[k] * ka[k];
    }
    auto emit = [&](const PoolGather<T, VEC>& pv, const Raw<T, VEC>& xv, T* __restrict__ dxp) {
        float d[VEC];
        pv.sum(d);
        Raw<T, VEC> o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float xk = xv.get(k);
            if constexpr (RELU) { if (!(xk * ka[k] + ksf[k] > 0.f)) d[k] = 0.f; }
            o.set(k, ka[k] * (d[k] - kb1[k] - (xk - kmu[k]) * kis[k]));
        }
        o.store(dxp);
    };
    for (; m + (POOL_ROWS_U - 1) * g.rpb < m1; m += POOL_ROWS_U * g.rpb, off += POOL_ROWS_U * rstep) {
        PoolGather<T, VEC> pv[POOL_ROWS_U];
        Raw<T, VEC> xv[POOL_ROWS_U];
#pragma unroll
        for (int u = 0; u < POOL_ROWS_U; ++u) {
            xv[u].load(x + off + u * rstep);
            pv[u].issue(dy, idx, pg, (uint32_t)(m + (int64_t)u * g.rpb), vcol, C);
        }
#pragma unroll
        for (int u = 0; u < POOL_ROWS_U; ++u) emit(pv[u], xv[u], dx + off + u * rstep);
    }
    for (; m < m1; m += g.rpb, off += rstep) {
        PoolGather<T, VEC> pv;
        Raw<T, VEC> xv;
        xv.load(x + off);
        pv.issue(dy, idx, pg, (uint32_t)m, vcol, C);
        emit(pv, xv, dx + off);
    }
}

inline int grid_for(int64_t nvec) {
    int64_t b = (nvec + 255) / 256;
    return (int)(b < 2048 ? (b > 0 ? b : 1) : 2048);
}

template <typename T> constexpr int vec_of() { return ElemTraits<T>::PER16; }

}  // namespace

size_t bn_workspace_bytes(int64_t M, int C) {
    // worst case geometry: VEC = 1
    SliceGeom g1 = make_geom(M, C, 1);
    size_t n = ws_partials_off(C) + (size_t)g1.slices * C * 3;
    SliceGeom g4 = make_geom(M, C, C % 4 == 0 ? 4 : 1), g8 = make_geom(M, C, C % 8 == 0 ? 8 : 1);
    size_t n4 = ws_partials_off(C) + (size_t)g4.slices * C * 3;
    size_t n8 = ws_partials_off(C) + (size_t)g8.slices * C * 3;
    if (n4 > n) n = n4;
    if (n8 > n) n = n8;
    return n * sizeof(float);
}

// Long external partial lists are first merged down to <= 128 entries per channel (into the
// workspace's own-partials area, unused when the partials are external).
template <int NV>
static int bn_premerge(const float*& partial, int& slices, int C, float* scratch, int64_t scratch_floats,
                       hipStream_t s) {
    if (slices < 1024) return 0;
    int shares = slices >= 2048 ? 128 : 64;
    if ((int64_t)shares * C * NV > scratch_floats) return 0;
    hipLaunchKernelGGL((bn_premerge_kernel<NV>), dim3(ceil_div(C, 64), shares), dim3(256), 0, s, partial, slices,
                       C, scratch);
    CNK_CHECK_LAUNCH("bn_premerge");
    partial = scratch;
    slices = shares;
    return 0;
}

// One launch for merge + finalisation (bn_merge_final_kernel) when the list is long enough to need two levels.
// Returns 1 when taken (done), 0 when the caller should run the two-kernel path, < 0 on error.
static std::mutex g_bnf_mu;
static hipStream_t g_bnf_stream[BNF_BANKS];
static unsigned g_bnf_next[BNF_BANKS];
static uint64_t g_bnf_used[BNF_BANKS];                 // clock of the bank's last launch (least recently used goes first)
static bool g_bnf_pinned[BNF_BANKS];                   // a launch of this bank was CAPTURED: a graph may replay it at any time
static hipEvent_t g_bnf_done[BNF_BANKS];               // recorded behind the bank's last (eager) launch
static uint64_t g_bnf_clock = 0;
static int g_bnf_banks = 0;
// A bank belongs to one stream.  When all are taken (a long-lived process that keeps creating streams: every GraphedTrainStep,
// every test stream), the least recently used bank whose last launch has COMPLETED moves to the new stream (banks
// used to be claimed for good, and the stream after the last bank silently took the two-kernel path for the rest of the
// process).  Completion is read off an event the library records behind every eager launch of the bank -- never off the old
// stream's handle, which may have been destroyed (touching it crashed the first version of this).  Never recycled: a bank with
// captured launches (no event can vouch for a graph's replays), and nothing is recycled on behalf of a capturing stream.  The
// fallback is counted (CNK_CNT_BN_FINAL_NOBANK).
static int bnf_slot_for(hipStream_t s, int* bank_out, bool* capturing_out) {               // -1: no bank for this stream right now
    // (g_bnf_mu is held by the caller until the launch's event is recorded: a bank is never seen idle between the two)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    int b = 0;
    while (b < g_bnf_banks && g_bnf_stream[b] != s) ++b;
    if (b == g_bnf_banks) {
        if (g_bnf_banks < BNF_BANKS) {
            if (hipEventCreateWithFlags(&g_bnf_done[b], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return -1; }
            ++g_bnf_banks;
        } else {
            if (capturing) return -1;
            int victim = -1;
            for (int k = 0; k < BNF_BANKS; ++k) {
                if (g_bnf_pinned[k] || (victim >= 0 && g_bnf_used[k] >= g_bnf_used[victim])) continue;
                if (hipEventQuery(g_bnf_done[k]) != hipSuccess) { (void)hipGetLastError(); continue; }      // still running (or unknown)
                victim = k;
            }
            if (victim < 0) return -1;
            b = victim;
        }
        g_bnf_stream[b] = s;
        g_bnf_next[b] = 0;
        g_bnf_pinned[b] = false;
    }
    g_bnf_used[b] = ++g_bnf_clock;
    if (capturing) g_bnf_pinned[b] = true;
    *bank_out = b;
    *capturing_out = capturing;
    return b * BNF_PER_BANK + (int)(g_bnf_next[b]++ % BNF_PER_BANK);
}
template <class FIN>
static int bn_merge_final(const float* partial, int slices, int C, int64_t M, float* scratch, int64_t scratch_floats,
                          const FIN& fin, hipStream_t s) {
    if (!cnk_sw().bn_fused_final || slices < 1024 || ceil_div(C, 64) > BNF_GROUPS) return 0;
    const int shares = slices >= 2048 ? 64 : 32;          // <= 64: bn_merge_final_kernel reads up to 16 shares per wave, all at once
    if ((int64_t)shares * C * FIN::NV > scratch_floats) return 0;
    int bank = 0;
    bool capturing = false;
    std::lock_guard<std::mutex> lk(g_bnf_mu);
    const int slot = bnf_slot_for(s, &bank, &capturing);
    if (slot < 0) {
        cnk_count(CNK_CNT_BN_FINAL_NOBANK);
        return 0;
    }
    hipLaunchKernelGGL((bn_merge_final_kernel<FIN>), dim3(ceil_div(C, 64), shares), dim3(256), 0, s, partial, slices, C, M,
                       scratch, slot, fin);
    if (!capturing) (void)hipEventRecord(g_bnf_done[bank], s);          // (what lets the bank move on once this stream is gone)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cnk_set_error("bn_merge_final: launch failed: %s", hipGetErrorString(e));
        return -2;
    }
    cnk_count(CNK_CNT_BN_FUSED_FINAL);
    return 1;
}

template <typename T, int VEC>
static int bn_fwd_train_t(const void* x, const float* gamma, const float* beta, const void* res,
                          void* y, unsigned char* mask, float* rm, float* rv, float* sm, float* si,
                          const float* ext_partial, int64_t ext_slices, int64_t M, int C, float eps, float mom, int relu,
                          float* ws, hipStream_t s, const float* raff = nullptr) {
    SliceGeom g = make_geom(M, C, VEC);
    float* scale = ws;
    float* shift = ws + C;
    const float* partial = ext_partial;
    int slices = (int)ext_slices;
    if (!ext_partial) {                  // statistics not supplied by the producing conv: own pass
        float* own = ws + ws_partials_off(C);
        hipLaunchKernelGGL((bn_stats_kernel<T, VEC>), dim3(g.slices, g.colgroups), dim3(256), 0, s,
                           (const T*)x, own, M, C, g);
        CNK_CHECK_LAUNCH("bn_stats");
        partial = own;
        slices = g.slices;
    }
    int fused_final = 0;
    if (ext_partial) {
        const BnFinalFwd fin = {gamma, beta, rm, rv, sm, si, scale, shift, eps, mom};
        fused_final = bn_merge_final(partial, slices, C, M, ws + ws_partials_off(C), (int64_t)g.slices * C * 3, fin, s);
        if (fused_final < 0) return -fused_final;
        if (!fused_final)
            if (int rc = bn_premerge<3>(partial, slices, C, ws + ws_partials_off(C), (int64_t)g.slices * C * 3, s)) return rc;
    }
    if (!fused_final) {
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, s, partial, slices, C,
                           M, gamma, beta, rm, rv, sm, si, scale, shift, eps, mom);
        CNK_CHECK_LAUNCH("bn_finalize");
        cnk_count(CNK_CNT_BN_FINAL_LAUNCH);
    }
    if (!y) return 0;                        // statistics only: scale | shift stay at ws[0..C) | ws[C..2C) for the caller
    const dim3 grid(g.slices, g.colgroups);
#define CNK_BN_APPLY(R, A, W)                                                                              \
    hipLaunchKernelGGL((bn_apply_kernel<T, VEC, R, A, W>), grid, dim3(256), 0, s, (const T*)x, (const T*)res, \
                       (T*)y, scale, shift, mask, M, C, g, nullptr)
#define CNK_BN_APPLY_RAFF(A, W)                                                                                        \
    hipLaunchKernelGGL((bn_apply_kernel<T, VEC, true, A, W, true>), grid, dim3(256), 0, s, (const T*)x, (const T*)res, \
                       (T*)y, scale, shift, mask, M, C, g, raff)
    const int key = (res ? 4 : 0) | (relu ? 2 : 0) | (mask ? 1 : 0);
    ProfScope kscope(CNK_KF_BN_APPLY, s, 0.0, (double)M * C * sizeof(T) * (res ? 3 : 2) + (mask ? (double)M * C / 8 : 0.0));
    if (raff && res) {
        switch (key & 3) {
            case 0: CNK_BN_APPLY_RAFF(false, false); break;
            case 1: CNK_BN_APPLY_RAFF(false, true); break;
            case 2: CNK_BN_APPLY_RAFF(true, false); break;
            default: CNK_BN_APPLY_RAFF(true, true); break;
        }
    } else
    switch (key) {
        case 0: CNK_BN_APPLY(false, false, false); break;
        case 1: CNK_BN_APPLY(false, false, true); break;
        case 2: CNK_BN_APPLY(false, true, false); break;
        case 3: CNK_BN_APPLY(false, true, true); break;
        case 4: CNK_BN_APPLY(true, false, false); break;
        case 5: CNK_BN_APPLY(true, false, true); break;
        case 6: CNK_BN_APPLY(true, true, false); break;
        default: CNK_BN_APPLY(true, true, true); break;
    }
#undef CNK_BN_APPLY
#undef CNK_BN_APPLY_RAFF
    CNK_CHECK_LAUNCH("bn_apply");
    return 0;
}

int bn_fwd_train(const void* x, const float* gamma, const float* beta, const void* res, void* y,
                 unsigned char* mask, float* rm, float* rv, float* sm, float* si, const float* ep, int64_t es, int64_t M, int C,
                 float eps, float mom, int relu, int dtype, float* ws, hipStream_t s, const float* raff) {
    if (dtype == CNK_BF16) {
        if (C % 8 == 0) return bn_fwd_train_t<bf16, 8>(x, gamma, beta, res, y, mask, rm, rv, sm, si, ep, es, M, C, eps, mom, relu, ws, s, raff);
        return bn_fwd_train_t<bf16, 1>(x, gamma, beta, res, y, mask, rm, rv, sm, si, ep, es, M, C, eps, mom, relu, ws, s, raff);
    }
    if (C % 4 == 0) return bn_fwd_train_t<float, 4>(x, gamma, beta, res, y, mask, rm, rv, sm, si, ep, es, M, C, eps, mom, relu, ws, s, raff);
    return bn_fwd_train_t<float, 1>(x, gamma, beta, res, y, mask, rm, rv, sm, si, ep, es, M, C, eps, mom, relu, ws, s, raff);
}

template <typename T, int VEC>
static int bn_eval_t(const void* x, const float* gamma, const float* beta, const float* rm,
                     const float* rv, const void* res, void* y, int64_t M, int C, float eps, int relu,
                     hipStream_t s) {
    int64_t nvec = M * C / VEC;
    hipLaunchKernelGGL((bn_eval_kernel<T, VEC>), dim3(grid_for(nvec)), dim3(256), 0, s, (const T*)x,
                       (const T*)res, (T*)y, gamma, beta, rm, rv, nvec, C, eps, relu);
    CNK_CHECK_LAUNCH("bn_eval");
    return 0;
}

int bn_fwd_eval(const void* x, const float* gamma, const float* beta, const float* rm,
                const float* rv, const void* res, void* y, int64_t M, int C, float eps, int relu,
                int dtype, hipStream_t s) {
    if (dtype == CNK_BF16) {
        if (C % 8 == 0) return bn_eval_t<bf16, 8>(x, gamma, beta, rm, rv, res, y, M, C, eps, relu, s);
        return bn_eval_t<bf16, 1>(x, gamma, beta, rm, rv, res, y, M, C, eps, relu, s);
    }
    if (C % 4 == 0) return bn_eval_t<float, 4>(x, gamma, beta, rm, rv, res, y, M, C, eps, relu, s);
    return bn_eval_t<float, 1>(x, gamma, beta, rm, rv, res, y, M, C, eps, relu, s);
}

template <typename T, int VEC>
static int bn_bwd_t(const void* dy, const void* x, const void* y, const unsigned char* mask, const float* gamma,
                    const float* beta, const float* sm, const float* si, void* dx, void* dres, float* dgamma,
                    float* dbeta, int accumulate, const float* ext_partial, int64_t ext_slices, int64_t M,
                    int C, int relu, float* ws, hipStream_t s) {
    SliceGeom g = make_geom(M, C, VEC);
    float* coefA = ws;
    float* coefB1 = ws + 2 * C;
    float* coefB2 = ws + 3 * C;
    float* partial = ws + ws_partials_off(C);
    // ReLU mask source: saved bit mask > forward output y > re-derived from x*scale+shift (no residual);
    // with partials from a fused conv dgrad, dy arrives masked and the reduction pass is skipped
    const int act = (!relu || ext_partial) ? ACT_NONE : mask ? ACT_BITS : y ? ACT_Y : ACT_REMASK;
    const dim3 grid(g.slices, g.colgroups);
    int nslices = g.slices;
    int fused_final = 0;
    if (ext_partial) {
        const float* ep = ext_partial;
        nslices = (int)ext_slices;
        const BnFinalBwd fin = {gamma, si, dgamma, dbeta, coefA, coefB1, coefB2, accumulate};
        fused_final = bn_merge_final(ep, nslices, C, M, partial, (int64_t)g.slices * C * 3, fin, s);
        if (fused_final < 0) return -fused_final;
        if (!fused_final)
            if (int rc = bn_premerge<2>(ep, nslices, C, partial, (int64_t)g.slices * C * 3, s)) return rc;
        partial = const_cast<float*>(ep);
    } else {
#define CNK_BN_RED(A)                                                                                       \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, VEC, A>), grid, dim3(256), 0, s, (const T*)dy, (const T*)x, \
                       (const T*)y, mask, gamma, beta, sm, si, partial, M, C, g)
    switch (act) {
        case ACT_NONE: CNK_BN_RED(ACT_NONE); break;
        case ACT_REMASK: CNK_BN_RED(ACT_REMASK); break;
        case ACT_BITS: CNK_BN_RED(ACT_BITS); break;
        default: CNK_BN_RED(ACT_Y); break;
    }
#undef CNK_BN_RED
    CNK_CHECK_LAUNCH("bn_bwd_reduce");
    }
    if (!fused_final) {
        hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(256), 0, s, partial, nslices,
                           C, M, gamma, si, dgamma, dbeta, coefA, coefB1, coefB2, accumulate);
        CNK_CHECK_LAUNCH("bn_bwd_final");
        cnk_count(CNK_CNT_BN_FINAL_LAUNCH);
    }
#define CNK_BN_BAPP(A, W)                                                                                     \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, VEC, A, W>), grid, dim3(256), 0, s, (const T*)dy, (const T*)x, \
                       (const T*)y, (T*)dx, (T*)dres, mask, beta, sm, si, coefA, coefB1, coefB2, M, C, g)
    ProfScope kscope(CNK_KF_BN_BWD_APPLY, s, 0.0, (double)M * C * sizeof(T) * (dres ? 4 : 3));
    switch (act * 2 + (dres ? 1 : 0)) {
        case ACT_NONE * 2: CNK_BN_BAPP(ACT_NONE, false); break;
        case ACT_NONE * 2 + 1: CNK_BN_BAPP(ACT_NONE, true); break;
        case ACT_REMASK * 2: CNK_BN_BAPP(ACT_REMASK, false); break;
        case ACT_REMASK * 2 + 1: CNK_BN_BAPP(ACT_REMASK, true); break;
        case ACT_BITS * 2: CNK_BN_BAPP(ACT_BITS, false); break;
        case ACT_BITS * 2 + 1: CNK_BN_BAPP(ACT_BITS, true); break;
        case ACT_Y * 2: CNK_BN_BAPP(ACT_Y, false); break;
        default: CNK_BN_BAPP(ACT_Y, true); break;
    }
#undef CNK_BN_BAPP
    CNK_CHECK_LAUNCH("bn_bwd_apply");
    return 0;
}

int bn_bwd(const void* dy, const void* x, const void* y, const unsigned char* mask, const float* gamma, const float* beta, const float* sm,
           const float* si, void* dx, void* dres, float* dgamma, float* dbeta, int accumulate,
           const float* ep, int64_t es, int64_t M, int C, int relu, int dtype, float* ws, hipStream_t s) {
    if (dtype == CNK_BF16) {
        if (C % 8 == 0) return bn_bwd_t<bf16, 8>(dy, x, y, mask, gamma, beta, sm, si, dx, dres, dgamma, dbeta, accumulate, ep, es, M, C, relu, ws, s);
        return bn_bwd_t<bf16, 1>(dy, x, y, mask, gamma, beta, sm, si, dx, dres, dgamma, dbeta, accumulate, ep, es, M, C, relu, ws, s);
    }
    if (C % 4 == 0) return bn_bwd_t<float, 4>(dy, x, y, mask, gamma, beta, sm, si, dx, dres, dgamma, dbeta, accumulate, ep, es, M, C, relu, ws, s);
    return bn_bwd_t<float, 1>(dy, x, y, mask, gamma, beta, sm, si, dx, dres, dgamma, dbeta, accumulate, ep, es, M, C, relu, ws, s);
}

// The finalisation half of the backward alone (cnk_bn_bwd_coef): dgamma / dbeta and the three coefficient vectors of
//   dx = A * (dy' - B1 - xhat * B2),  A = gamma * invstd, B1 = mean(dy'), B2 = mean(dy' * xhat)
// from the partial sums a consumer conv's data-gradient epilogue left (cnk_bn_bwd_fuse) -- or, for a layer WITHOUT activation
// whose gradient arrives as a plain tensor (a projection shortcut's BN), from this call's own reduction pass over (dy, x).
// The apply pass is not run: the caller folds it into the backward of the 1x1 conv in front of this layer
// (cnk_conv2d_dgrad_bnfold / _wgrad_bnfold).
template <typename T, int VEC>
static int bn_bwd_coef_t(const void* dy, const void* x, const float* ext_partial, int64_t ext_slices, const float* gamma,
                         const float* sm, const float* si, float* dgamma, float* dbeta, int accumulate, float* coef, int64_t M,
                         int C, float* ws, hipStream_t s) {
    const SliceGeom g = make_geom(M, C, VEC);
    float* coefA = coef;
    float* coefB1 = coef + C;
    float* coefB2 = coef + 2 * (size_t)C;
    float* scratch = ws + ws_partials_off(C);
    const float* ep = ext_partial;
    int nslices = (int)ext_slices;
    int fused_final = 0;
    if (ext_partial) {
        const BnFinalBwd fin = {gamma, si, dgamma, dbeta, coefA, coefB1, coefB2, accumulate};
        fused_final = bn_merge_final(ep, nslices, C, M, scratch, (int64_t)g.slices * C * 3, fin, s);
        if (fused_final < 0) return -fused_final;
        if (!fused_final)
            if (int rc = bn_premerge<2>(ep, nslices, C, scratch, (int64_t)g.slices * C * 3, s)) return rc;
    } else {
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, VEC, ACT_NONE>), dim3(g.slices, g.colgroups), dim3(256), 0, s, (const T*)dy,
                           (const T*)x, (const T*)nullptr, (const unsigned char*)nullptr, gamma, (const float*)nullptr, sm, si, scratch,
                           M, C, g);
        CNK_CHECK_LAUNCH("bn_bwd_reduce");
        ep = scratch;
        nslices = g.slices;
    }
    if (!fused_final) {
        hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(256), 0, s, ep, nslices, C, M, gamma, si, dgamma, dbeta, coefA,
                           coefB1, coefB2, accumulate);
        CNK_CHECK_LAUNCH("bn_bwd_final");
        cnk_count(CNK_CNT_BN_FINAL_LAUNCH);
    }
    return 0;
}

int bn_bwd_coef(const void* dy, const void* x, const float* ext_partial, int64_t ext_slices, const float* gamma, const float* sm,
                const float* si, float* dgamma, float* dbeta, int accumulate, float* coef, int64_t M, int C, int dtype, float* ws,
                hipStream_t s) {
    if (dtype == CNK_BF16) {
        if (C % 8 == 0) return bn_bwd_coef_t<bf16, 8>(dy, x, ext_partial, ext_slices, gamma, sm, si, dgamma, dbeta, accumulate, coef, M, C, ws, s);
        return bn_bwd_coef_t<bf16, 1>(dy, x, ext_partial, ext_slices, gamma, sm, si, dgamma, dbeta, accumulate, coef, M, C, ws, s);
    }
    if (C % 4 == 0) return bn_bwd_coef_t<float, 4>(dy, x, ext_partial, ext_slices, gamma, sm, si, dgamma, dbeta, accumulate, coef, M, C, ws, s);
    return bn_bwd_coef_t<float, 1>(dy, x, ext_partial, ext_slices, gamma, sm, si, dgamma, dbeta, accumulate, coef, M, C, ws, s);
}

namespace {

// Stride-2 pools (every pool of the reference's models): one thread owns a 2 x 2 cell of input elements.  The <= 2 x 2
// windows that touch the cell are loaded once and serve its four elements, and which element sits where inside which
// window does not depend on the cell -- pos[dy][dx][a][b] below is a launch constant (-1: not covered), so the
// per-element work is byte compares against scalars.
struct PoolCell {
    int H, W, Ho, Wo, cells_h, cells_w;
    int lo_h, lo_w;              // first window of cell (cj, ci): (cj + lo_h, ci + lo_w)
    int nwin_h, nwin_w;          // windows per cell and dimension (1 or 2)
    int pos[2][2][2][2];         // [dy][dx][a][b]
    FastDiv fd_cw, fd_ch;
};

template <typename T, int VEC> struct CellGather {
    using IB = typename PoolGather<T, VEC>::IB;
    Raw<T, VEC> g[4];
    IB b[4];
    Raw<T, VEC> xv[4];
    int64_t xoff;                // element offset of the cell's first element
    bool in_h1, in_w1;           // second row / column of the cell inside the image
    __device__ __forceinline__ void issue(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                          const T* __restrict__ x, const PoolCell& p, uint32_t cell, int vcol, int C) {
        const uint32_t t = fd_div(cell, p.fd_cw), ci = cell - t * p.cells_w;
        const uint32_t n = fd_div(t, p.fd_ch), cj = t - n * p.cells_h;
        const int h0 = 2 * (int)cj, w0 = 2 * (int)ci;
        in_h1 = h0 + 1 < p.H;
        in_w1 = w0 + 1 < p.W;
        xoff = ((int64_t)(n * p.H + h0) * p.W + w0) * C + (int64_t)vcol * VEC;
        // All twelve loads of a cell are UNCONDITIONAL (absent neighbours / windows re-read the cell's first element or
        // window: addresses always valid, results unused): bounds branches around the loads made them one dependent round
        // trip each (round 3, from the .s); the caller keeps them together with a sched_barrier.
        const int64_t dw = in_w1 ? C : 0, dh = in_h1 ? (int64_t)p.W * C : 0;
        xv[0].load(x + xoff);
        xv[1].load(x + xoff + dw);
        xv[2].load(x + xoff + dh);
        xv[3].load(x + xoff + dh + dw);
        IB none;
        if constexpr (VEC == 8) none = IB{0xffffffffu, 0xffffffffu};      // 255 is no window position (kh*kw <= 255)
        else if constexpr (VEC == 4) none = 0xffffffffu;
        else none = 0xff;
        bool okk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int a = k >> 1, c = k & 1;
            const int ho = (int)cj + p.lo_h + a, wo = (int)ci + p.lo_w + c;
            okk[k] = a < p.nwin_h && c < p.nwin_w && (unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo;
            const int hoc = okk[k] ? ho : 0, woc = okk[k] ? wo : 0;
            const size_t o = ((size_t)(n * p.Ho + hoc) * p.Wo + woc) * C + (size_t)vcol * VEC;
            g[k].load(dy + o);
            b[k] = *(const IB*)(idx + o);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (!okk[k]) b[k] = none;
    }
    // gradient reaching element e = dy*2+dx of the cell, windows summed in (ho, wo) order
    template <int E> __device__ __forceinline__ void sum(const PoolCell& p, float (&d)[VEC]) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) d[i] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ps = p.pos[E >> 1][E & 1][k >> 1][k & 1];
            if (ps < 0) continue;                       // launch constant: a scalar branch
            unsigned char by[VEC];
            __builtin_memcpy(by, &b[k], VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if ((int)by[i] == ps) d[i] += g[k].get(i);
        }
    }
    template <int E> __device__ __forceinline__ bool inside() const {
        return (E & 1 ? in_w1 : true) && (E >> 1 ? in_h1 : true);
    }
    template <int E> __device__ __forceinline__ int64_t offset(const PoolCell& p, int C) const {
        return xoff + (E & 1 ? C : 0) + (E >> 1 ? (int64_t)p.W * C : 0);
    }
};

template <typename T, int VEC, bool RELU>
__global__ __launch_bounds__(256) void bn_cell_bwd_reduce_kernel(
    const T* __restrict__ dy, const unsigned char* __restrict__ idx, const T* __restrict__ x,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, float* __restrict__ partial, int64_t cells, int C, SliceGeom g, PoolCell pc) {
    __shared__ float sh[2][256 * (VEC > 1 ? VEC : 1)];
    const int tid = threadIdx.x, tcol = tid % g.cpb, trow = tid / g.cpb;
    const int vcol = blockIdx.y * g.cpb + tcol;
    const bool active = trow < g.rpb && vcol < g.cols;
    const int slice = pool_slice_of_block(blockIdx.x, g.slices);
    const int64_t m0 = (int64_t)slice * g.slice_rows;
    const int64_t m1 = m0 + g.slice_rows < cells ? m0 + g.slice_rows : cells;
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;
    if (active) {
        float mu[VEC], sc[VEC], sf[VEC], is[VEC];
        load_coef<VEC>(mu, mean + vcol * VEC);
        load_coef<VEC>(is, invstd + vcol * VEC);
        if constexpr (RELU) {
            load_coef<VEC>(sc, gamma + vcol * VEC);
            load_coef<VEC>(sf, beta + vcol * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) { sc[i] *= is[i]; sf[i] -= mu[i] * sc[i]; }
        }
        for (int64_t m = m0 + trow; m < m1; m += g.rpb) {
            CellGather<T, VEC> cg;
            cg.issue(dy, idx, x, pc, (uint32_t)m, vcol, C);
            __builtin_amdgcn_sched_barrier(0);
            auto element = [&](auto tag) {
                constexpr int E = decltype(tag)::value;
                if (!cg.template inside<E>()) return;
                float d[VEC];
                cg.template sum<E>(pc, d);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float xi = cg.xv[E].get(i);
                    if constexpr (RELU) { if (!(xi * sc[i] + sf[i] > 0.f)) d[i] = 0.f; }
                    s1[i] += d[i];
                    s2[i] += d[i] * (xi - mu[i]);
                }
            };
            element(std::integral_constant<int, 0>{});
            element(std::integral_constant<int, 1>{});
            element(std::integral_constant<int, 2>{});
            element(std::integral_constant<int, 3>{});
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) s2[i] *= is[i];
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        sh[0][tid * VEC + i] = s1[i];
        sh[1][tid * VEC + i] = s2[i];
    }
    __syncthreads();
    int span = 1;
    while (span < g.rpb) span <<= 1;
    for (int st = span >> 1; st >= 1; st >>= 1) {
        if (trow < st && trow + st < g.rpb && vcol < g.cols) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                int a = tid * VEC + i, b = (tid + st * g.cpb) * VEC + i;
                sh[0][a] += sh[0][b];
                sh[1][a] += sh[1][b];
            }
        }
        __syncthreads();
    }
    if (trow == 0 && vcol < g.cols) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            int c = vcol * VEC + i;
            float* o = partial + ((size_t)slice * C + c) * 2;
            o[0] = sh[0][tid * VEC + i]; o[1] = sh[1][tid * VEC + i];
        }
    }
}

template <typename T, int VEC, bool RELU>
__global__ __launch_bounds__(256) void bn_cell_bwd_apply_kernel(
    const T* __restrict__ dy, const unsigned char* __restrict__ idx, const T* __restrict__ x, T* __restrict__ dx,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ coefA, const float* __restrict__ coefB1, const float* __restrict__ coefB2, int64_t cells,
    int C, SliceGeom g, PoolCell pc) {
    const int tid = threadIdx.x, tcol = tid % g.cpb, trow = tid / g.cpb;
    const int vcol = blockIdx.y * g.cpb + tcol;
    if (trow >= g.rpb || vcol >= g.cols) return;
    const int slice = pool_slice_of_block(blockIdx.x, g.slices);
    const int64_t m0 = (int64_t)slice * g.slice_rows;
    const int64_t m1 = m0 + g.slice_rows < cells ? m0 + g.slice_rows : cells;
    float ka[VEC], kb1[VEC], kmu[VEC], kis[VEC], ksf[VEC];
    load_coef<VEC>(ka, coefA + vcol * VEC);             // gamma*invstd == the forward's scale
    load_coef<VEC>(kb1, coefB1 + vcol * VEC);
    load_coef<VEC>(kmu, mean + vcol * VEC);
    load_coef<VEC>(kis, invstd + vcol * VEC);
    {
        float b2[VEC];
        load_coef<VEC>(b2, coefB2 + vcol * VEC);
#pragma unroll
        for (int k = 0; k < VEC; ++k) kis[k] *= b2[k];
    }
    if constexpr (RELU) {
        load_coef<VEC>(ksf, beta + vcol * VEC);
#pragma unroll
        for (int k = 0; k < VEC; ++k) ksf[k] -= kmu[k] * ka[k];
    }
    for (int64_t m = m0 + trow; m < m1; m += g.rpb) {
        CellGather<T, VEC> cg;
        cg.issue(dy, idx, x, pc, (uint32_t)m, vcol, C);
        __builtin_amdgcn_sched_barrier(0);
        auto element = [&](auto tag) {
            constexpr int E = decltype(tag)::value;
            if (!cg.template inside<E>()) return;
            float d[VEC];
            cg.template sum<E>(pc, d);
            Raw<T, VEC> o;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float xk = cg.xv[E].get(k);
                if constexpr (RELU) { if (!(xk * ka[k] + ksf[k] > 0.f)) d[k] = 0.f; }
                o.set(k, ka[k] * (d[k] - kb1[k] - (xk - kmu[k]) * kis[k]));
            }
            o.store(dx + cg.template offset<E>(pc, C));
        };
        element(std::integral_constant<int, 0>{});
        element(std::integral_constant<int, 1>{});
        element(std::integral_constant<int, 2>{});
        element(std::integral_constant<int, 3>{});
    }
}

// stride-2 pools whose 2 x 2 cells meet at most 2 windows per dimension: fill the cell description
static bool make_pool_cell(PoolCell& pc, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw) {
    if (sh != 2 || sw != 2) return false;
    auto cdiv = [](int a, int b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); };     // ceil(a / b), b > 0
    auto fdiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };       // floor(a / b)
    pc.H = H; pc.W = W;
    pc.Ho = (H + 2 * ph - kh) / sh + 1;
    pc.Wo = (W + 2 * pw - kw) / sw + 1;
    pc.cells_h = (H + 1) / 2; pc.cells_w = (W + 1) / 2;
    pc.lo_h = cdiv(ph - kh + 1, 2); pc.lo_w = cdiv(pw - kw + 1, 2);
    pc.nwin_h = fdiv(1 + ph, 2) - pc.lo_h + 1; pc.nwin_w = fdiv(1 + pw, 2) - pc.lo_w + 1;
    if (pc.nwin_h < 1 || pc.nwin_h > 2 || pc.nwin_w < 1 || pc.nwin_w > 2) return false;
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx)
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    const int r = dy - (pc.lo_h + a) * 2 + ph, q = dx - (pc.lo_w + b) * 2 + pw;
                    const bool in = a < pc.nwin_h && b < pc.nwin_w && r >= 0 && r < kh && q >= 0 && q < kw;
                    pc.pos[dy][dx][a][b] = in ? r * kw + q : -1;
                }
    pc.fd_cw = make_fastdiv(pc.cells_w); pc.fd_ch = make_fastdiv(pc.cells_h);
    return true;
}

}  // namespace

template <typename T, int VEC>
static int bn_bwd_pool_t(const void* dy, const unsigned char* idx, const void* x, const void* xmax, const float* gamma,
                         const float* beta, const float* sm, const float* si, void* dx, float* dgamma, float* dbeta,
                         int accumulate, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, int relu,
                         float* ws, hipStream_t s) {
    const int64_t M = (int64_t)N * H * W;
    SliceGeom g = make_geom(M, C, VEC);
    PoolGeom pg;
    pg.H = H; pg.W = W; pg.kh = kh; pg.kw = kw; pg.sh = sh; pg.sw = sw; pg.ph = ph; pg.pw = pw;
    pg.Ho = (H + 2 * ph - kh) / sh + 1;
    pg.Wo = (W + 2 * pw - kw) / sw + 1;
    pg.fd_w = make_fastdiv(W); pg.fd_h = make_fastdiv(H);
    float* coefA = ws;
    float* coefB1 = ws + 2 * C;
    float* coefB2 = ws + 3 * C;
    float* partial = ws + ws_partials_off(C);
    PoolCell pc;
    const bool cell = make_pool_cell(pc, H, W, kh, kw, sh, sw, ph, pw) && !cnk_sw().bn_pool_elementwise;
    if (cell) {
        const int64_t cells = (int64_t)N * pc.cells_h * pc.cells_w;
        SliceGeom gc = make_geom(cells, C, VEC);
        const dim3 cgrid(gc.slices, gc.colgroups);
        int red_slices = gc.slices;
        if (xmax) {
            // The pooled gradient only reaches arg-max elements, and the forward kept their INPUT values (x_at_max): both sums
            // run over the POOLED tensors -- sum d' and sum d' (x - mean) over windows equal the sums over elements, an element
            // that wins two windows contributes twice -- i.e. the regular reduce pass on [N Ho Wo][C] rows, a quarter of the
            // bytes of the gather over x (ResNet-50 stem: 206 MB instead of 565).
            const int64_t Mp = (int64_t)N * pc.Ho * pc.Wo;
            SliceGeom gp = make_geom(Mp, C, VEC);
            red_slices = gp.slices;
            const dim3 pgrid(gp.slices, gp.colgroups);
            if (relu)
                hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, VEC, ACT_REMASK>), pgrid, dim3(256), 0, s, (const T*)dy, (const T*)xmax,
                                   (const T*)nullptr, (const unsigned char*)nullptr, gamma, beta, sm, si, partial, Mp, C, gp);
            else
                hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, VEC, ACT_NONE>), pgrid, dim3(256), 0, s, (const T*)dy, (const T*)xmax,
                                   (const T*)nullptr, (const unsigned char*)nullptr, gamma, beta, sm, si, partial, Mp, C, gp);
            CNK_CHECK_LAUNCH("bn_bwd_reduce(pooled)");
        } else {
            if (relu)
                hipLaunchKernelGGL((bn_cell_bwd_reduce_kernel<T, VEC, true>), cgrid, dim3(256), 0, s, (const T*)dy, idx,
                                   (const T*)x, gamma, beta, sm, si, partial, cells, C, gc, pc);
            else
                hipLaunchKernelGGL((bn_cell_bwd_reduce_kernel<T, VEC, false>), cgrid, dim3(256), 0, s, (const T*)dy, idx,
                                   (const T*)x, gamma, beta, sm, si, partial, cells, C, gc, pc);
            CNK_CHECK_LAUNCH("bn_cell_bwd_reduce");
        }
        hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(256), 0, s, partial, red_slices, C, M, gamma, si, dgamma,
                           dbeta, coefA, coefB1, coefB2, accumulate);
        CNK_CHECK_LAUNCH("bn_bwd_final");
        const double pooled = (double)N * pc.Ho * pc.Wo * C * (sizeof(T) + 1);
        ProfScope kscope(CNK_KF_BN_BWD_APPLY, s, 0.0, (double)M * C * sizeof(T) * 2 + pooled);
        if (relu)
            hipLaunchKernelGGL((bn_cell_bwd_apply_kernel<T, VEC, true>), cgrid, dim3(256), 0, s, (const T*)dy, idx,
                               (const T*)x, (T*)dx, beta, sm, si, coefA, coefB1, coefB2, cells, C, gc, pc);
        else
            hipLaunchKernelGGL((bn_cell_bwd_apply_kernel<T, VEC, false>), cgrid, dim3(256), 0, s, (const T*)dy, idx,
                               (const T*)x, (T*)dx, beta, sm, si, coefA, coefB1, coefB2, cells, C, gc, pc);
        CNK_CHECK_LAUNCH("bn_cell_bwd_apply");
        return 0;
    }
    const dim3 grid(g.slices, g.colgroups);
    if (relu)
        hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<T, VEC, true>), grid, dim3(256), 0, s, (const T*)dy, idx, (const T*)x,
                           gamma, beta, sm, si, partial, M, C, g, pg);
    else
        hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<T, VEC, false>), grid, dim3(256), 0, s, (const T*)dy, idx,
                           (const T*)x, gamma, beta, sm, si, partial, M, C, g, pg);
    CNK_CHECK_LAUNCH("bn_pool_bwd_reduce");
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(256), 0, s, partial, g.slices, C, M, gamma, si, dgamma, dbeta,
                       coefA, coefB1, coefB2, accumulate);
    CNK_CHECK_LAUNCH("bn_bwd_final");
    const double pooled = (double)N * pg.Ho * pg.Wo * C * (sizeof(T) + 1);
    ProfScope kscope(CNK_KF_BN_BWD_APPLY, s, 0.0, (double)M * C * sizeof(T) * 2 + pooled);
    if (relu)
        hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<T, VEC, true>), grid, dim3(256), 0, s, (const T*)dy, idx, (const T*)x,
                           (T*)dx, beta, sm, si, coefA, coefB1, coefB2, M, C, g, pg);
    else
        hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<T, VEC, false>), grid, dim3(256), 0, s, (const T*)dy, idx,
                           (const T*)x, (T*)dx, beta, sm, si, coefA, coefB1, coefB2, M, C, g, pg);
    CNK_CHECK_LAUNCH("bn_pool_bwd_apply");
    return 0;
}

int bn_bwd_pool(const void* dy, const unsigned char* idx, const void* x, const void* xmax, const float* gamma, const float* beta,
                const float* sm, const float* si, void* dx, float* dgamma, float* dbeta, int accumulate, int N, int H,
                int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, int relu, int dtype, float* ws,
                hipStream_t s) {
#define CNK_BN_POOL(T, V) bn_bwd_pool_t<T, V>(dy, idx, x, xmax, gamma, beta, sm, si, dx, dgamma, dbeta, accumulate, N, H, W, C, kh, kw, sh, sw, ph, pw, relu, ws, s)
    if (dtype == CNK_BF16) return C % 8 == 0 ? CNK_BN_POOL(bf16, 8) : CNK_BN_POOL(bf16, 1);
    return C % 4 == 0 ? CNK_BN_POOL(float, 4) : CNK_BN_POOL(float, 1);
#undef CNK_BN_POOL
}

// bytes of the packed ReLU mask: one byte per 16-byte channel vector the kernels use for (C, dtype)
size_t bn_mask_bytes(int64_t M, int C, int dtype) {
    int vec = dtype == CNK_BF16 ? (C % 8 == 0 ? 8 : 1) : (C % 4 == 0 ? 4 : 1);
    return (size_t)M * (C / vec);
}
