// Classifier head on HIP: nn.Linear forward / input gradient / weight gradient,
// softmax + CrossEntropyLoss(reduction='sum') (src/basemodel.py:46) and nn.Dropout
// (src/resnet.py:105, src/vggnet.py:81-90).  The head computes in fp32 (exact-fp32 MFMA).
#include "common.h"

namespace {

// ---- strided fp32 GEMM -----------------------------------------------------------------------------
//   C[m][n] (+)= sum_k A(m,k) * B(n,k) [+ bias[n]],   A(m,k) = a[m*sam + k*sak],  B(n,k) = b[n*sbn + k*sbk]
// One of the two strides of each operand is 1; the loader walks that one with consecutive lanes.
// 64x64 output tile per 256-thread workgroup, 4 waves in 2x2, one 32x32 block per wave computed by
// v_mfma_f32_32x32x2f32 (A: lane -> row lane%32, k lane/32); K step 32; LDS tiles [k][row], pitch 68
// floats (rows of one k on consecutive banks: conflict-free fragment reads).
//   Linear forward   y[B][N]  = x[B][K] W[N][K]^T + b : A = x (k contiguous), B = W (k contiguous)
//   input gradient   dx[B][K] = dy[B][N] W[N][K]      : A = dy (k contiguous), B(n=k', k=n) = W (row contiguous)
//   weight gradient  dW[N][K] = dy[B][N]^T x[B][K]    : A(m=n, k=b) = dy, B(n=k', k=b) = x (both row contiguous)
struct GemmParams {
    const float* a;
    const float* b;
    const float* bias;
    float* c;
    int M, N, K, ldc;
    int64_t sam, sak, sbn, sbk;
    int accumulate;
    int ksplit;             // K range per blockIdx.z (multiple of GK); partial results go to slabs
    float* slab;            // [gridDim.z][M][N] when gridDim.z > 1
};

constexpr int GT = 64, GK = 32, GP = 68;

// VEC4: 16-byte loads along the contiguous index (host checks stride and base alignment)
template <bool KCONTIG, bool VEC4>
__device__ __forceinline__ void load_tile(float* __restrict__ T, const float* __restrict__ src, int row0, int nrows,
                                          int k0, int k1, int64_t srow, int64_t sk, int tid) {
    if constexpr (VEC4) {
#pragma unroll
        for (int i = 0; i < GT * GK / 1024; ++i) {
            const int idx = tid + 256 * i;                      // one float4 each
            if constexpr (KCONTIG) {
                const int r = idx / (GK / 4), k = (idx % (GK / 4)) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row0 + r < nrows) {
                    const float* p = src + (int64_t)(row0 + r) * srow + (k0 + k);
                    if (k0 + k + 4 <= k1) v = *(const f32x4*)p;
                    else
                        for (int e = 0; e < 4; ++e) if (k0 + k + e < k1) v[e] = p[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) T[(k + e) * GP + r] = v[e];
            } else {
                const int k = idx / (GT / 4), r = (idx % (GT / 4)) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k0 + k < k1) {
                    const float* p = src + (int64_t)(k0 + k) * sk + (row0 + r);
                    if (row0 + r + 4 <= nrows) v = *(const f32x4*)p;
                    else
                        for (int e = 0; e < 4; ++e) if (row0 + r + e < nrows) v[e] = p[e];
                }
                *(f32x4*)(T + k * GP + r) = v;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < GT * GK / 256; ++i) {
            const int idx = tid + 256 * i;
            const int r = KCONTIG ? idx / GK : idx % GT;
            const int k = KCONTIG ? idx % GK : idx / GT;
            float v = 0.f;
            if (row0 + r < nrows && k0 + k < k1) v = src[(int64_t)(row0 + r) * srow + (int64_t)(k0 + k) * sk];
            T[k * GP + r] = v;
        }
    }
}

template <bool AK, bool BK, bool VEC4>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float As[GK * GP];
    __shared__ __attribute__((aligned(16))) float Bs[GK * GP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int kb = blockIdx.z * p.ksplit, ke = min(p.K, kb + p.ksplit);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = kb; k0 < ke; k0 += GK) {
        load_tile<AK, VEC4>(As, p.a, m0, p.M, k0, ke, p.sam, p.sak, tid);
        load_tile<BK, VEC4>(Bs, p.b, n0, p.N, k0, ke, p.sbn, p.sbk, tid);
        __syncthreads();
#pragma unroll
        for (int ss = 0; ss < GK / 2; ++ss) {
            const float fa = As[(2 * ss + lh) * GP + wm * 32 + l31];
            const float fb = Bs[(2 * ss + lh) * GP + wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C layout of the 32x32 block: column = lane%32, rows (r&3) + 8*(r>>2) + 4*(lane/32)
    const int n = n0 + wn * 32 + l31;
    if (n >= p.N) return;
    if (gridDim.z > 1) {                                  // split K: plain stores into this split's slab
        float* sl = p.slab + (size_t)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < p.M) sl[(int64_t)m * p.N + n] = acc[r];
        }
        return;
    }
    const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.M) {
            float* o = p.c + (int64_t)m * p.ldc + n;
            const float v = acc[r] + bias;
            *o = p.accumulate ? *o + v : v;
        }
    }
}

// C (+)= sum of the split-K slabs in fixed order (+ bias): deterministic, unlike float atomics
__global__ __launch_bounds__(256) void gemm_slab_sum_kernel(const float* __restrict__ slab, int splits, float* __restrict__ c,
                                                            const float* __restrict__ bias, int M, int N, int ldc,
                                                            int accumulate) {
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
        float a = 0.f;
        for (int sp = 0; sp < splits; ++sp) a += slab[(size_t)sp * total + i];
        if (bias) a += bias[n];
        float* o = c + (int64_t)m * ldc + n;
        *o = accumulate ? *o + a : a;
    }
}

// The accumulator's column index runs with the lane, so C rows are written with consecutive lanes on
// consecutive n: coalesced for y / dx / dW alike.

// ---- softmax + cross entropy, reduction='sum' --------------------------------------------------------
// One block per sample: row max, sum of exponentials, loss_i = log(sum) + max - logit[target] and, in
// the same pass, d(loss)/d(logits) = softmax - onehot (the backward only scales it by the upstream
// scalar).  argmax (first maximum, as torch.argmax) feeds score() without another pass.
__device__ __forceinline__ float block_max_256(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return t;
}
__device__ __forceinline__ float block_add_256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits,
                                                         const int64_t* __restrict__ target,
                                                         float* __restrict__ row_loss, float* __restrict__ dlogits,
                                                         int32_t* __restrict__ argmax_out, int N) {
    __shared__ float sh[4];
    __shared__ int shi[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (int64_t)b * N;
    float mx = -INFINITY;
    int mi = 0x7fffffff;
    for (int i = tid; i < N; i += 256) {
        const float v = x[i];
        if (v > mx) { mx = v; mi = i; }            // strided scan keeps the smallest index per thread
    }
    const float gmax = block_max_256(mx, sh);
    int cand = (mx == gmax) ? mi : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    if ((tid & 63) == 0) shi[tid >> 6] = cand;
    float se = 0.f;
    for (int i = tid; i < N; i += 256) se += __expf(x[i] - gmax);
    const float sum = block_add_256(se, sh);          // (barriers inside also publish shi)
    const int64_t t = target[b];
    if (tid == 0) {
        row_loss[b] = (t >= 0 && t < N) ? logf(sum) + gmax - x[t] : 0.f;
        if (argmax_out) argmax_out[b] = min(min(shi[0], shi[1]), min(shi[2], shi[3]));
    }
    if (dlogits) {
        const float inv = 1.f / sum;
        float* d = dlogits + (int64_t)b * N;
        for (int i = tid; i < N; i += 256) d[i] = __expf(x[i] - gmax) * inv - (i == t ? 1.f : 0.f);
    }
}

// loss = sum_b row_loss[b], fixed order (one block)
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ float sh[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += v[i];
    const float t = block_add_256(a, sh);
    if (threadIdx.x == 0) *out = t;
}

// eight consecutive elements as floats (one 16-byte vector of bf16, two of fp32)
template <typename T> struct VL8;
template <> struct VL8<float> {
    float f[8];
    __device__ __forceinline__ void load(const float* p) {
        const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
        *(float4*)(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
};
template <> struct VL8<bf16> {
    float f[8];
    __device__ __forceinline__ void load(const bf16* p) {
        const bf16x8 v = *(const bf16x8*)p;
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = (float)v[k];
    }
    __device__ __forceinline__ void store(bf16* p) const {
        bf16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (bf16)f[k];
        *(bf16x8*)p = v;
    }
};

// ---- dropout ------------------------------------------------------------------------------------------
// Counter-based: element i of call `offset` under `seed` keeps iff hash(seed, offset, i) >= p*2^32; kept
// values are scaled by 1/(1-p) (nn.Dropout training semantics).  The keep mask is saved bit-packed, (n+7)/8 bytes.
// fp32 (the classifier head) or bf16 (feature-map dropout: DenseNet's dense layers, SqueezeNet's head), in place of
// memory order -- the tensor layout does not matter.
__device__ __forceinline__ uint32_t mix32(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 16);
}

// One thread handles 8 consecutive elements (memory order) and one byte of the bit-packed keep mask.
template <typename T>
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          unsigned char* __restrict__ mask, int64_t n,
                                                          uint32_t thresh, float scale, uint64_t seed, uint64_t offset,
                                                          const uint64_t* __restrict__ offset_dev) {
    if (offset_dev) offset += *offset_dev;        // hipGraph-captured step: the call counter of THIS replay
    const uint64_t key = seed * 0xD1342543DE82EF95ull + offset * 0x2545F4914F6CDD1Dull;
    const int64_t groups = (n + 7) >> 3;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t base = g << 3;
        unsigned bits = 0;
        if (base + 8 <= n) {
            VL8<T> v;
            v.load(x + base);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool keep = mix32(key + (uint64_t)(base + k)) >= thresh;
                bits |= (unsigned)keep << k;
                v.f[k] = keep ? v.f[k] * scale : 0.f;
            }
            v.store(y + base);
        } else {
            for (int k = 0; base + k < n; ++k) {
                const bool keep = mix32(key + (uint64_t)(base + k)) >= thresh;
                bits |= (unsigned)keep << k;
                y[base + k] = from_f32<T>(keep ? to_f32(x[base + k]) * scale : 0.f);
            }
        }
        mask[g] = (unsigned char)bits;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ mask,
                                                          T* __restrict__ dx, int64_t n, float scale) {
    const int64_t groups = (n + 7) >> 3;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t base = g << 3;
        const unsigned bits = mask[g];
        if (base + 8 <= n) {
            VL8<T> v;
            v.load(dy + base);
#pragma unroll
            for (int k = 0; k < 8; ++k) v.f[k] = (bits >> k) & 1u ? v.f[k] * scale : 0.f;
            v.store(dx + base);
        } else {
            for (int k = 0; base + k < n; ++k) dx[base + k] = from_f32<T>((bits >> k) & 1u ? to_f32(dy[base + k]) * scale : 0.f);
        }
    }
}

// ---- squeeze-and-excitation unit (src/senet.py:215-254, src/se_resnet.py:216-255) -----------------------
// excitation gate: nn.Sigmoid on the [N][C] fp32 descriptor
__global__ __launch_bounds__(256) void sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = 1.f / (1.f + __expf(-x[i]));
}
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          float* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * y[i] * (1.f - y[i]);
}

// scale: y[n][p][c] = x[n][p][c] * g[n][c]   (x * output.expand_as(x)); lanes along C
template <typename T>
__global__ __launch_bounds__(256) void channel_scale_fwd_kernel(const T* __restrict__ x, const float* __restrict__ g,
                                                                T* __restrict__ y, int64_t total, int HW, int C) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t n = i / ((int64_t)HW * C);
        y[i] = from_f32<T>(to_f32(x[i]) * g[n * C + c]);
    }
}
// dx = dy * g;  dg[n][c] = sum_p dy*x : one thread per (n, c) walks the HW pixels (coalesced across c)
template <typename T>
__global__ __launch_bounds__(256) void channel_scale_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                const float* __restrict__ g, T* __restrict__ dx,
                                                                float* __restrict__ dg, int N, int HW, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    const size_t base = (size_t)n * HW * C + c;
    const float gv = g[i];
    float acc = 0.f;
    for (int k = 0; k < HW; ++k) {
        const float d = to_f32(dy[base + (size_t)k * C]);
        acc += d * to_f32(x[base + (size_t)k * C]);
        dx[base + (size_t)k * C] = from_f32<T>(d * gv);
    }
    dg[i] = acc;
}

inline int grid1d(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 2048 ? (b > 0 ? b : 1) : 2048);
}

// Split K so that skinny problems (batch rows x huge weight matrix: the VGG / AlexNet heads) still put
// ~512 workgroups on the chip; partial tiles go to slabs in the workspace and are summed in fixed order.
int gemm_splits(int M, int N, int K) {
    int tiles = ceil_div(M, GT) * ceil_div(N, GT);
    if (tiles >= 256) return 1;
    int want = 512 / tiles;
    int maxk = K / (4 * GK);                               // at least four K steps per split
    int sp = want < maxk ? want : maxk;
    return sp < 1 ? 1 : sp;
}

size_t gemm_workspace(int M, int N, int K) {
    int sp = gemm_splits(M, N, K);
    return sp > 1 ? (size_t)sp * M * N * sizeof(float) : 0;
}

template <bool AK, bool BK> int launch_gemm_v(GemmParams& p, bool vec4, dim3 grid, hipStream_t s) {
    if (vec4) hipLaunchKernelGGL((gemm_f32_kernel<AK, BK, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<AK, BK, false>), grid, dim3(256), 0, s, p);
    return 0;
}

int launch_gemm(GemmParams p, bool ak, bool bk, float* ws, size_t ws_bytes, hipStream_t s) {
    int sp = gemm_splits(p.M, p.N, p.K);
    if (sp > 1 && (!ws || ws_bytes < (size_t)sp * p.M * p.N * sizeof(float))) sp = 1;
    p.ksplit = ceil_div(ceil_div(p.K, sp), GK) * GK;
    sp = ceil_div(p.K, p.ksplit);
    p.slab = ws;
    // 16-byte loads need the contiguous index to start 16-byte aligned in every row / column
    auto ok = [](const float* base, int64_t other_stride) {
        return ((uintptr_t)base % 16 == 0) && (other_stride % 4 == 0);
    };
    const bool vec4 = ok(p.a, ak ? p.sam : p.sak) && ok(p.b, bk ? p.sbn : p.sbk);
    dim3 grid(ceil_div(p.N, GT), ceil_div(p.M, GT), sp);
    if (ak && bk) launch_gemm_v<true, true>(p, vec4, grid, s);
    else if (ak && !bk) launch_gemm_v<true, false>(p, vec4, grid, s);
    else if (!ak && bk) launch_gemm_v<false, true>(p, vec4, grid, s);
    else launch_gemm_v<false, false>(p, vec4, grid, s);
    CNK_CHECK_LAUNCH("gemm_f32");
    if (sp > 1) {
        int64_t total = (int64_t)p.M * p.N;
        hipLaunchKernelGGL(gemm_slab_sum_kernel, dim3(grid1d(total)), dim3(256), 0, s, ws, sp, p.c, p.bias, p.M, p.N,
                           p.ldc, p.accumulate);
        CNK_CHECK_LAUNCH("gemm_slab_sum");
    }
    return 0;
}

}  // namespace

size_t linear_workspace(int pass, int B, int K, int N) {
    if (pass == 0) return gemm_workspace(B, N, K);
    if (pass == 1) return gemm_workspace(B, K, N);
    return gemm_workspace(N, K, B);
}

int linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, float* ws,
               size_t ws_bytes, hipStream_t s) {
    GemmParams p = {x, w, bias, y, B, N, K, N, K, 1, K, 1, 0, 0, nullptr};
    return launch_gemm(p, true, true, ws, ws_bytes, s);
}

int linear_dgrad(const float* dy, const float* w, float* dx, int B, int K, int N, float* ws, size_t ws_bytes,
                 hipStream_t s) {
    // dx[b][k'] = sum_n dy[b][n] * w[n][k']
    GemmParams p = {dy, w, nullptr, dx, B, K, N, K, N, 1, 1, K, 0, 0, nullptr};
    return launch_gemm(p, true, false, ws, ws_bytes, s);
}

int linear_wgrad(const float* dy, const float* x, float* dw, int B, int K, int N, int accumulate, float* ws,
                 size_t ws_bytes, hipStream_t s) {
    // dw[n][k'] = sum_b dy[b][n] * x[b][k']
    GemmParams p = {dy, x, nullptr, dw, N, K, B, K, 1, N, 1, K, accumulate, 0, nullptr};
    return launch_gemm(p, false, false, ws, ws_bytes, s);
}

int softmax_ce_sum(const float* logits, const int64_t* target, float* loss, float* dlogits, int32_t* argmax,
                   float* row_loss_ws, int B, int N, hipStream_t s) {
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(B), dim3(256), 0, s, logits, target, row_loss_ws, dlogits, argmax, N);
    CNK_CHECK_LAUNCH("softmax_ce");
    hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, s, row_loss_ws, B, loss);
    CNK_CHECK_LAUNCH("sum_rows");
    return 0;
}

int dropout_fwd(const void* x, void* y, unsigned char* mask, int64_t n, float p, uint64_t seed, uint64_t offset,
                const uint64_t* offset_dev, int dtype, hipStream_t s) {
    double t = (double)p * 4294967296.0;
    uint32_t thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    const int grid = grid1d((n + 7) / 8);
    if (dtype == CNK_BF16)
        hipLaunchKernelGGL(dropout_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, mask, n, thresh,
                           1.f / (1.f - p), seed, offset, offset_dev);
    else
        hipLaunchKernelGGL(dropout_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, mask, n, thresh,
                           1.f / (1.f - p), seed, offset, offset_dev);
    CNK_CHECK_LAUNCH("dropout_fwd");
    return 0;
}

int dropout_bwd(const void* dy, const unsigned char* mask, void* dx, int64_t n, float p, int dtype, hipStream_t s) {
    const int grid = grid1d((n + 7) / 8);
    if (dtype == CNK_BF16)
        hipLaunchKernelGGL(dropout_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)dy, mask, (bf16*)dx, n, 1.f / (1.f - p));
    else
        hipLaunchKernelGGL(dropout_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, mask, (float*)dx, n, 1.f / (1.f - p));
    CNK_CHECK_LAUNCH("dropout_bwd");
    return 0;
}

int sigmoid_fwd(const float* x, float* y, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(grid1d(n)), dim3(256), 0, s, x, y, n);
    CNK_CHECK_LAUNCH("sigmoid_fwd");
    return 0;
}
int sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(grid1d(n)), dim3(256), 0, s, dy, y, dx, n);
    CNK_CHECK_LAUNCH("sigmoid_bwd");
    return 0;
}
int channel_scale_fwd(const void* x, const float* g, void* y, int N, int HW, int C, int dtype, hipStream_t s) {
    const int64_t total = (int64_t)N * HW * C;
    if (dtype == CNK_BF16) hipLaunchKernelGGL(channel_scale_fwd_kernel<bf16>, dim3(grid1d(total)), dim3(256), 0, s, (const bf16*)x, g, (bf16*)y, total, HW, C);
    else hipLaunchKernelGGL(channel_scale_fwd_kernel<float>, dim3(grid1d(total)), dim3(256), 0, s, (const float*)x, g, (float*)y, total, HW, C);
    CNK_CHECK_LAUNCH("channel_scale_fwd");
    return 0;
}
int channel_scale_bwd(const void* dy, const void* x, const float* g, void* dx, float* dg, int N, int HW, int C, int dtype,
                      hipStream_t s) {
    const int grid = ceil_div(N * C, 256);
    if (dtype == CNK_BF16) hipLaunchKernelGGL(channel_scale_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, g, (bf16*)dx, dg, N, HW, C);
    else hipLaunchKernelGGL(channel_scale_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, (const float*)x, g, (float*)dx, dg, N, HW, C);
    CNK_CHECK_LAUNCH("channel_scale_bwd");
    return 0;
}
