// Selective-kernel unit (src/sknet.py:172-240 == src/sk_resnet.py:171-239) on NHWC feature maps.
//
//   fuse   : pooled[n][c] = mean_hw sum_i U_i[n][hw][c]                      (sknet.py:222-228)
//   select : a = softmax_i(logit_i[n][c]);  out = sum_i U_i * a_i            (sknet.py:231-237)
//
// All of it is HBM-bound streaming work: the P feature maps are read once per pass with 16-byte lanes along C,
// per-(n, c) reductions go through fixed-order partials in a workspace (deterministic, no atomics), and the
// backward of both reads of U_i is produced by ONE pass (dU_i = dout * a_i + dpooled / HW).
#include "common.h"

namespace {

constexpr int SK_MAX_PATHS = 4;

struct SkMaps {
    const void* u[SK_MAX_PATHS];
};
struct SkMapsOut {
    void* u[SK_MAX_PATHS];
};
struct SkVecs {
    const float* v[SK_MAX_PATHS];
};
struct SkVecsOut {
    float* v[SK_MAX_PATHS];
};

template <typename T> struct SkVec;
template <> struct SkVec<bf16> {
    static constexpr int N = 8;
    using Raw = uint4;
    static __device__ __forceinline__ void unpack(const Raw& r, float* f) {
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ Raw pack(const float* f) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16 lo = from_f32<bf16>(f[2 * i]), hi = from_f32<bf16>(f[2 * i + 1]);
            w[i] = (unsigned)(*reinterpret_cast<unsigned short*>(&lo)) | ((unsigned)(*reinterpret_cast<unsigned short*>(&hi)) << 16);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct SkVec<float> {
    static constexpr int N = 4;
    using Raw = float4;
    static __device__ __forceinline__ void unpack(const Raw& r, float* f) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
    static __device__ __forceinline__ Raw pack(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
};

// Column/row split of one 256-thread block over an [rows][C] slab: cpp vector columns per pass (<= 256), the
// remaining lanes walk rows.  blockIdx.z picks the 256-column block when C has more than 256 vectors.
struct SkLane {
    int col, row0, rstep;
    bool on;
};
__device__ __forceinline__ SkLane sk_lane(int CV) {
    const int cpp = CV < 256 ? CV : 256;
    const int rpp = 256 / cpp;
    SkLane l;
    l.col = blockIdx.z * 256 + (int)threadIdx.x % cpp;
    l.row0 = (int)threadIdx.x / cpp;
    l.rstep = rpp;
    l.on = l.row0 < rpp && l.col < CV;
    return l;
}

// partial[n][chunk][q][c] = sum over the chunk's rows of  (DOT ? dout * U_q : sum_i U_i)   (Q = DOT ? P : 1)
// WDX (SE unit backward, P = 1): the same pass also writes dx = dout * gate[n][c] (channel-scale gradient).
template <typename T, int P, bool DOT, bool WDX = false>
__global__ __launch_bounds__(256) void sk_reduce_kernel(SkMaps maps, const T* __restrict__ dout,
                                                        float* __restrict__ partial, int HW, int C, int chunks,
                                                        const float* __restrict__ gate = nullptr, T* __restrict__ dx = nullptr) {
    constexpr int V = SkVec<T>::N;
    constexpr int Q = DOT ? P : 1;
    using Raw = typename SkVec<T>::Raw;
    __shared__ float red[256 * V];
    const int CV = C / V;
    const SkLane l = sk_lane(CV);
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int rows = (HW + chunks - 1) / chunks;
    const int r_lo = chunk * rows, r_hi = min(HW, r_lo + rows);
    float acc[Q][V];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[q][j] = 0.f;
    if (l.on) {
        const size_t img = (size_t)n * HW * CV;
        float gv[V];
        if (WDX) {
#pragma unroll
            for (int j = 0; j < V; ++j) gv[j] = gate[(size_t)n * C + (size_t)l.col * V + j];
        }
        for (int r = r_lo + l.row0; r < r_hi; r += l.rstep) {
            const size_t at = img + (size_t)r * CV + l.col;
            float d[V];
            if (DOT) SkVec<T>::unpack(reinterpret_cast<const Raw*>(dout)[at], d);
            if (WDX) {
                float o[V];
#pragma unroll
                for (int j = 0; j < V; ++j) o[j] = d[j] * gv[j];
                reinterpret_cast<Raw*>(dx)[at] = SkVec<T>::pack(o);
            }
#pragma unroll
            for (int i = 0; i < P; ++i) {
                float u[V];
                SkVec<T>::unpack(reinterpret_cast<const Raw*>(maps.u[i])[at], u);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    if (DOT) acc[i][j] += d[j] * u[j];
                    else acc[0][j] += u[j];
                }
            }
        }
    }
    // fold the row lanes of each column (fixed order), one Q at a time through the same LDS slab
    const int cpp = CV < 256 ? CV : 256;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < V; ++j) red[threadIdx.x * V + j] = acc[q][j];
        __syncthreads();
        if (l.row0 == 0 && l.col < CV) {
            float s[V];
#pragma unroll
            for (int j = 0; j < V; ++j) s[j] = 0.f;
            for (int rl = 0; rl < l.rstep; ++rl)
#pragma unroll
                for (int j = 0; j < V; ++j) s[j] += red[(rl * cpp + (int)threadIdx.x) * V + j];
            float* o = partial + (((size_t)n * chunks + chunk) * Q + q) * C + (size_t)l.col * V;
#pragma unroll
            for (int j = 0; j < V; ++j) o[j] = s[j];
        }
    }
}

// pooled[n][c] = (1 / HW) * sum_chunk partial[n][chunk][0][c]
__global__ __launch_bounds__(256) void sk_pool_finish_kernel(const float* __restrict__ partial, float* __restrict__ pooled,
                                                             int N, int C, int chunks, float inv_hw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += partial[((size_t)n * chunks + k) * C + c];
    pooled[i] = s * inv_hw;
}

// attn[i][n][c] = softmax over the P paths of logit_i[n][c]     (nn.Softmax(dim=1) on [N, P, C, 1, 1])
template <int P>
__global__ __launch_bounds__(256) void sk_attention_fwd_kernel(SkVecs logits, float* __restrict__ attn, int NC) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    float l[P], m = -INFINITY, s = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) { l[p] = logits.v[p][i]; m = fmaxf(m, l[p]); }
#pragma unroll
    for (int p = 0; p < P; ++p) { l[p] = expf(l[p] - m); s += l[p]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int p = 0; p < P; ++p) attn[(size_t)p * NC + i] = l[p] * inv;
}

// da_p = sum_chunk partial[n][chunk][p][c];  dlogit_p = a_p * (da_p - sum_q a_q * da_q)
template <int P>
__global__ __launch_bounds__(256) void sk_attention_bwd_kernel(const float* __restrict__ attn, const float* __restrict__ partial,
                                                               SkVecsOut dlogits, int N, int C, int chunks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int NC = N * C;
    if (i >= NC) return;
    const int n = i / C, c = i - n * C;
    float a[P], da[P], dot = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        float s = 0.f;
        for (int k = 0; k < chunks; ++k) s += partial[(((size_t)n * chunks + k) * P + p) * C + c];
        da[p] = s;
        a[p] = attn[(size_t)p * NC + i];
        dot += a[p] * s;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) dlogits.v[p][i] = a[p] * (da[p] - dot);
}

// out = sum_i U_i * a_i[n][c]
template <typename T, int P>
__global__ __launch_bounds__(256) void sk_select_fwd_kernel(SkMaps maps, const float* __restrict__ attn, T* __restrict__ out,
                                                            int64_t total_v, int HW, int C, int NC) {
    constexpr int V = SkVec<T>::N;
    using Raw = typename SkVec<T>::Raw;
    const int CV = C / V;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total_v; v += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(v % CV);
        const int64_t n = v / ((int64_t)HW * CV);
        const float* a0 = attn + n * C + (int64_t)cv * V;
        float o[V];
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = 0.f;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            float u[V];
            SkVec<T>::unpack(reinterpret_cast<const Raw*>(maps.u[i])[v], u);
            const float* a = a0 + (size_t)i * NC;
#pragma unroll
            for (int j = 0; j < V; ++j) o[j] += u[j] * a[j];
        }
        reinterpret_cast<Raw*>(out)[v] = SkVec<T>::pack(o);
    }
}

// dU_i = dout * a_i[n][c] + dpooled[n][c] / HW      (either operand may be absent)
template <typename T, int P>
__global__ __launch_bounds__(256) void sk_maps_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ attn,
                                                          const float* __restrict__ dpooled, SkMapsOut du,
                                                          int64_t total_v, int HW, int C, int NC, float inv_hw) {
    constexpr int V = SkVec<T>::N;
    using Raw = typename SkVec<T>::Raw;
    const int CV = C / V;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total_v; v += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(v % CV);
        const int64_t n = v / ((int64_t)HW * CV);
        const int64_t nc = n * C + (int64_t)cv * V;
        float d[V], g[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { d[j] = 0.f; g[j] = 0.f; }
        if (dout) SkVec<T>::unpack(reinterpret_cast<const Raw*>(dout)[v], d);
        if (dpooled)
#pragma unroll
            for (int j = 0; j < V; ++j) g[j] = dpooled[nc + j] * inv_hw;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            float o[V];
#pragma unroll
            for (int j = 0; j < V; ++j) o[j] = g[j];
            if (dout) {
                const float* a = attn + (size_t)i * NC + nc;
#pragma unroll
                for (int j = 0; j < V; ++j) o[j] += d[j] * a[j];
            }
            reinterpret_cast<Raw*>(du.u[i])[v] = SkVec<T>::pack(o);
        }
    }
}

int sk_vec(int dtype) { return dtype == CNK_BF16 ? 8 : 4; }

}  // namespace

int sk_chunks(int N, int HW) {
    int want = (2048 + N - 1) / N;                 // ~2048 blocks on the chip
    int cap = (HW + 15) / 16;                      // at least 16 rows per chunk
    int c = want < cap ? want : cap;
    return c < 1 ? 1 : c;
}
bool sk_supported(int C, int P, int dtype) { return P >= 1 && P <= SK_MAX_PATHS && C > 0 && C % sk_vec(dtype) == 0; }
size_t sk_workspace(int N, int HW, int C, int P) { return (size_t)N * sk_chunks(N, HW) * P * C * sizeof(float); }

static int sk_grid1d(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096);
}

template <typename T, int P, bool DOT>
static void launch_reduce(const SkMaps& m, const void* dout, float* partial, int N, int HW, int C, hipStream_t s) {
    const int CV = C / SkVec<T>::N, chunks = sk_chunks(N, HW);
    dim3 grid(chunks, N, (CV + 255) / 256);
    hipLaunchKernelGGL((sk_reduce_kernel<T, P, DOT>), grid, dim3(256), 0, s, m, (const T*)dout, partial, HW, C, chunks);
}
template <int P, bool DOT>
static void launch_reduce_t(const SkMaps& m, const void* dout, float* partial, int N, int HW, int C, int dtype, hipStream_t s) {
    if (dtype == CNK_BF16) launch_reduce<bf16, P, DOT>(m, dout, partial, N, HW, C, s);
    else launch_reduce<float, P, DOT>(m, dout, partial, N, HW, C, s);
}

#define SK_SWITCH_P(P, ...)                  \
    switch (P) {                             \
        case 1: { constexpr int PP = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int PP = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int PP = 3; __VA_ARGS__; } break; \
        default: { constexpr int PP = 4; __VA_ARGS__; } break; \
    }

int sk_pool_fwd(const void* const* u, int P, float* pooled, int N, int HW, int C, int dtype, float* ws, hipStream_t s) {
    SkMaps m = {};
    for (int i = 0; i < P; ++i) m.u[i] = u[i];
    SK_SWITCH_P(P, launch_reduce_t<PP, false>(m, nullptr, ws, N, HW, C, dtype, s));
    CNK_CHECK_LAUNCH("sk_pool_reduce");
    hipLaunchKernelGGL(sk_pool_finish_kernel, dim3(ceil_div(N * C, 256)), dim3(256), 0, s, ws, pooled, N, C, sk_chunks(N, HW), 1.f / (float)HW);
    CNK_CHECK_LAUNCH("sk_pool_finish");
    return 0;
}

int sk_select_fwd(const void* const* u, const float* const* logits, int P, float* attn, void* out, int N, int HW, int C,
                  int dtype, hipStream_t s) {
    SkMaps m = {};
    SkVecs l = {};
    for (int i = 0; i < P; ++i) { m.u[i] = u[i]; l.v[i] = logits[i]; }
    const int NC = N * C;
    SK_SWITCH_P(P, hipLaunchKernelGGL(sk_attention_fwd_kernel<PP>, dim3(ceil_div(NC, 256)), dim3(256), 0, s, l, attn, NC));
    CNK_CHECK_LAUNCH("sk_attention_fwd");
    const int64_t total_v = (int64_t)N * HW * (C / sk_vec(dtype));
    if (dtype == CNK_BF16) {
        SK_SWITCH_P(P, hipLaunchKernelGGL((sk_select_fwd_kernel<bf16, PP>), dim3(sk_grid1d(total_v)), dim3(256), 0, s, m, attn, (bf16*)out, total_v, HW, C, NC));
    } else {
        SK_SWITCH_P(P, hipLaunchKernelGGL((sk_select_fwd_kernel<float, PP>), dim3(sk_grid1d(total_v)), dim3(256), 0, s, m, attn, (float*)out, total_v, HW, C, NC));
    }
    CNK_CHECK_LAUNCH("sk_select_fwd");
    return 0;
}

int sk_select_bwd_logits(const void* const* u, const float* attn, const void* dout, float* const* dlogits, int P, int N, int HW,
                         int C, int dtype, float* ws, hipStream_t s) {
    SkMaps m = {};
    SkVecsOut dl = {};
    for (int i = 0; i < P; ++i) { m.u[i] = u[i]; dl.v[i] = dlogits[i]; }
    SK_SWITCH_P(P, launch_reduce_t<PP, true>(m, dout, ws, N, HW, C, dtype, s));
    CNK_CHECK_LAUNCH("sk_select_reduce");
    SK_SWITCH_P(P, hipLaunchKernelGGL(sk_attention_bwd_kernel<PP>, dim3(ceil_div(N * C, 256)), dim3(256), 0, s, attn, ws, dl, N, C, sk_chunks(N, HW)));
    CNK_CHECK_LAUNCH("sk_attention_bwd");
    return 0;
}

int sk_maps_bwd(const void* dout, const float* attn, const float* dpooled, void* const* du, int P, int N, int HW, int C,
                int dtype, hipStream_t s) {
    SkMapsOut o = {};
    for (int i = 0; i < P; ++i) o.u[i] = du[i];
    const int NC = N * C;
    const int64_t total_v = (int64_t)N * HW * (C / sk_vec(dtype));
    const float inv_hw = 1.f / (float)HW;
    if (dtype == CNK_BF16) {
        SK_SWITCH_P(P, hipLaunchKernelGGL((sk_maps_bwd_kernel<bf16, PP>), dim3(sk_grid1d(total_v)), dim3(256), 0, s, (const bf16*)dout, attn, dpooled, o, total_v, HW, C, NC, inv_hw));
    } else {
        SK_SWITCH_P(P, hipLaunchKernelGGL((sk_maps_bwd_kernel<float, PP>), dim3(sk_grid1d(total_v)), dim3(256), 0, s, (const float*)dout, attn, dpooled, o, total_v, HW, C, NC, inv_hw));
    }
    CNK_CHECK_LAUNCH("sk_maps_bwd");
    return 0;
}

// ---- squeeze-and-excitation unit on the same passes (P = 1) -------------------------------------------------
// y = x * gate[n][c]: the select pass with the gate as the (already normalised) attention
int sk_scale_fwd(const void* x, const float* gate, void* y, int N, int HW, int C, int dtype, hipStream_t s) {
    SkMaps m = {};
    m.u[0] = x;
    const int64_t total_v = (int64_t)N * HW * (C / sk_vec(dtype));
    if (dtype == CNK_BF16)
        hipLaunchKernelGGL((sk_select_fwd_kernel<bf16, 1>), dim3(sk_grid1d(total_v)), dim3(256), 0, s, m, gate, (bf16*)y, total_v, HW, C, N * C);
    else
        hipLaunchKernelGGL((sk_select_fwd_kernel<float, 1>), dim3(sk_grid1d(total_v)), dim3(256), 0, s, m, gate, (float*)y, total_v, HW, C, N * C);
    CNK_CHECK_LAUNCH("sk_scale_fwd");
    return 0;
}
// dx = dy * gate, dgate[n][c] = sum_hw dy * x: ONE pass over dy and x (chunked fixed-order partials) + the chunk sum
int sk_scale_bwd(const void* dy, const void* x, const float* gate, void* dx, float* dgate, int N, int HW, int C, int dtype,
                 float* ws, hipStream_t s) {
    SkMaps m = {};
    m.u[0] = x;
    const int chunks = sk_chunks(N, HW);
    // dx == nullptr: only the reduction (the caller writes dx later, together with the squeeze branch's gradient)
    if (dtype == CNK_BF16) {
        dim3 grid(chunks, N, (C / 8 + 255) / 256);
        if (dx) hipLaunchKernelGGL((sk_reduce_kernel<bf16, 1, true, true>), grid, dim3(256), 0, s, m, (const bf16*)dy, ws, HW, C, chunks, gate, (bf16*)dx);
        else hipLaunchKernelGGL((sk_reduce_kernel<bf16, 1, true, false>), grid, dim3(256), 0, s, m, (const bf16*)dy, ws, HW, C, chunks, nullptr, nullptr);
    } else {
        dim3 grid(chunks, N, (C / 4 + 255) / 256);
        if (dx) hipLaunchKernelGGL((sk_reduce_kernel<float, 1, true, true>), grid, dim3(256), 0, s, m, (const float*)dy, ws, HW, C, chunks, gate, (float*)dx);
        else hipLaunchKernelGGL((sk_reduce_kernel<float, 1, true, false>), grid, dim3(256), 0, s, m, (const float*)dy, ws, HW, C, chunks, nullptr, nullptr);
    }
    CNK_CHECK_LAUNCH("sk_scale_bwd_reduce");
    hipLaunchKernelGGL(sk_pool_finish_kernel, dim3(ceil_div(N * C, 256)), dim3(256), 0, s, ws, dgate, N, C, chunks, 1.f);
    CNK_CHECK_LAUNCH("sk_scale_bwd_finish");
    return 0;
}

