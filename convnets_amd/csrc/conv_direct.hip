// Generic NHWC convolution kernels for shapes the MFMA implicit-GEMM tiles do not cover
// (Cin of 1/3/6/32..., Cout below 64: the LeNet-5 / ConvNet layers and the first layer of
// AlexNet / VGG / ResNet), plus the weight re-layout kernels.  fp32 accumulation throughout.
//
// Direct-path weight image (fwd and dgrad share it): w_t[kh][kw][Cin][Cout] fp32.
#include "common.h"

namespace {

struct DirectParams {
    const void* a;      // fwd: x      dgrad: dy     wgrad: dy
    const void* b;      // fwd: w_t    dgrad: w_t    wgrad: x
    void* out;          // fwd: y      dgrad: dx     wgrad: slab
    const float* bias;
    const void* addend; // dgrad: optional tensor added to dx
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, sh, sw, ph, pw;
    int M;              // fwd/wgrad: N*Ho*Wo   dgrad: N*H*W
    int chunk;          // wgrad: pixels per grid.y slice
    FastDiv fd_a, fd_b; // fwd/wgrad: (Ho*Wo, Wo)   dgrad: (H*W, W)
};

// y[m][co] = bias[co] + sum_{t,ci} x[src(m,t)][ci] * w_t[t][ci][co]; threads: x = co, y = pixel.
template <typename T> __global__ void direct_fwd_kernel(const DirectParams p) {
    int co = blockIdx.y * 64 + threadIdx.x;
    int m = blockIdx.x * blockDim.y + threadIdx.y;
    if (m >= p.M || co >= p.Cout) return;
    uint32_t img = fd_div(m, p.fd_a), rem = m - img * p.fd_a.d;
    uint32_t ho = fd_div(rem, p.fd_b), wo = rem - ho * p.fd_b.d;
    const T* x = (const T*)p.a;
    const float* w = (const float*)p.b;
    float acc = p.bias ? p.bias[co] : 0.f;
    for (int r = 0; r < p.KH; ++r) {
        int ih = (int)ho * p.sh - p.ph + r;
        if ((unsigned)ih >= (unsigned)p.H) continue;
        for (int q = 0; q < p.KW; ++q) {
            int iw = (int)wo * p.sw - p.pw + q;
            if ((unsigned)iw >= (unsigned)p.W) continue;
            const T* xp = x + ((size_t)(img * p.H + ih) * p.W + iw) * p.Cin;
            const float* wp = w + (size_t)(r * p.KW + q) * p.Cin * p.Cout + co;
            for (int ci = 0; ci < p.Cin; ++ci) acc += to_f32(xp[ci]) * wp[(size_t)ci * p.Cout];
        }
    }
    ((T*)p.out)[(size_t)m * p.Cout + co] = from_f32<T>(acc);
}

// dx[pix][ci] = sum over taps with (h+p-kh)%s==0 of dy[src][co] * w_t[t][ci][co]; threads: x = ci.
template <typename T> __global__ void direct_dgrad_kernel(const DirectParams p) {
    int ci = blockIdx.y * 64 + threadIdx.x;
    int m = blockIdx.x * blockDim.y + threadIdx.y;
    if (m >= p.M || ci >= p.Cin) return;
    uint32_t img = fd_div(m, p.fd_a), rem = m - img * p.fd_a.d;
    uint32_t h = fd_div(rem, p.fd_b), w0 = rem - h * p.fd_b.d;
    const T* dy = (const T*)p.a;
    const float* w = (const float*)p.b;
    float acc = 0.f;
    for (int r = 0; r < p.KH; ++r) {
        int th = (int)h + p.ph - r;
        if (th < 0 || th % p.sh) continue;
        int ho = th / p.sh;
        if (ho >= p.Ho) continue;
        for (int q = 0; q < p.KW; ++q) {
            int tw = (int)w0 + p.pw - q;
            if (tw < 0 || tw % p.sw) continue;
            int wo = tw / p.sw;
            if (wo >= p.Wo) continue;
            const T* dp = dy + ((size_t)(img * p.Ho + ho) * p.Wo + wo) * p.Cout;
            const float* wp = w + ((size_t)(r * p.KW + q) * p.Cin + ci) * p.Cout;
            for (int co = 0; co < p.Cout; ++co) acc += to_f32(dp[co]) * wp[co];
        }
    }
    T o = from_f32<T>(acc);
    if (p.addend) o = from_f32<T>(to_f32(o) + to_f32(((const T*)p.addend)[(size_t)m * p.Cin + ci]));
    ((T*)p.out)[(size_t)m * p.Cin + ci] = o;
}

// slab[slice][t][ci][co] = sum over the slice's pixels of dy[m][co] * x[src(m,t)][ci]
template <typename T> __global__ void direct_wgrad_kernel(const DirectParams p) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int total = p.KH * p.KW * p.Cin * p.Cout;
    if (idx >= total) return;
    int co = idx % p.Cout, r1 = idx / p.Cout;
    int ci = r1 % p.Cin, t = r1 / p.Cin;
    int r = t / p.KW, q = t - r * p.KW;
    const T* dy = (const T*)p.a;
    const T* x = (const T*)p.b;
    int m0 = blockIdx.y * p.chunk, m1 = min(p.M, m0 + p.chunk);
    float acc = 0.f;
    for (int m = m0; m < m1; ++m) {
        uint32_t img = fd_div(m, p.fd_a), rem = m - img * p.fd_a.d;
        uint32_t ho = fd_div(rem, p.fd_b), wo = rem - ho * p.fd_b.d;
        int ih = (int)ho * p.sh - p.ph + r, iw = (int)wo * p.sw - p.pw + q;
        if ((unsigned)ih >= (unsigned)p.H || (unsigned)iw >= (unsigned)p.W) continue;
        acc += to_f32(dy[(size_t)m * p.Cout + co]) *
               to_f32(x[((size_t)(img * p.H + ih) * p.W + iw) * p.Cin + ci]);
    }
    ((float*)p.out)[(size_t)blockIdx.y * total + idx] = acc;
}

// partial[slice][c] = sum over the slice's rows of a[m][c]
template <typename T>
__global__ void colsum_partial_kernel(const T* __restrict__ a, float* __restrict__ partial, int M,
                                      int C, int chunk) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    int m0 = blockIdx.y * chunk, m1 = min(M, m0 + chunk);
    float acc = 0.f;
    for (int m = m0; m < m1; ++m) acc += to_f32(a[(size_t)m * C + c]);
    partial[(size_t)blockIdx.y * C + c] = acc;
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                    int slices, int C, int accumulate) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float acc = 0.f;
    for (int s = 0; s < slices; ++s) acc += partial[(size_t)s * C + c];
    out[c] = accumulate ? out[c] + acc : acc;
}

// ---- weight re-layout -----------------------------------------------------------------------------
// out[(r*ntaps + t)*Cred + c] = w[co][ci][kh[t]][kw[t]] with (r,c) = (co,ci) [fwd] or (ci,co) [dgrad]
struct PackParams {
    const float* w;
    void* out;
    int Cout, Cin, KH, KW, ntaps, transpose, groups;
    int Rpad, Cpad;          // image [Rpad][ntaps][Cpad], zero outside [R][..][Cred] (off-grid channel counts)
    signed char kh[32], kw[32];
};
// master weight value for dense image position (co, ci, tap): with groups > 1 the master tensor is
// [Cout][Cin/groups][KH][KW] and the image is block-diagonal (zero outside co's group)
__device__ __forceinline__ float master_w(const float* __restrict__ w, int co, int ci, int Cout, int Cin, int KH, int KW,
                                          int kh, int kw, int groups) {
    if (groups <= 1) return w[(((size_t)co * Cin + ci) * KH + kh) * KW + kw];
    const int gw = Cin / groups, g = co / (Cout / groups), l = ci - g * gw;
    if ((unsigned)l >= (unsigned)gw) return 0.f;
    return w[(((size_t)co * gw + l) * KH + kh) * KW + kw];
}
template <typename T> __global__ void pack_kernel(const PackParams p) {
    int R = p.transpose ? p.Cin : p.Cout, Cred = p.transpose ? p.Cout : p.Cin;
    size_t total = (size_t)p.Rpad * p.ntaps * p.Cpad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(i % p.Cpad);
        size_t q = i / p.Cpad;
        int t = (int)(q % p.ntaps), r = (int)(q / p.ntaps);
        float v = 0.f;
        if (r < R && c < Cred) {
            int co = p.transpose ? c : r, ci = p.transpose ? r : c;
            v = master_w(p.w, co, ci, p.Cout, p.Cin, p.KH, p.KW, p.kh[t], p.kw[t], p.groups);
        }
        ((T*)p.out)[i] = from_f32<T>(v);
    }
}
// w_t[t][ci][co] = w[co][ci][t]
__global__ void pack_direct_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout,
                                   int Cin, int T_) {
    size_t total = (size_t)Cout * Cin * T_;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        int co = (int)(i % Cout);
        size_t q = i / Cout;
        int ci = (int)(q % Cin), t = (int)(q / Cin);
        out[i] = w[((size_t)co * Cin + ci) * T_ + t];
    }
}

// Multi-tensor form: one launch re-lays every conv weight of the model.  jobs[] and the
// block -> (job, first block of the job) table live in device memory (built once by the host).
// Block decomposition (cnk_pack_job::nblocks, pack_job_blocks below):
//  * 1x1 forward images ([co][ci], the master's own order) and grouped / large-kernel jobs: 1024 consecutive
//    outputs per block, element-wise;
//  * k x k forward images ([co][t][ci]): one co row x 256 ci per block, the contiguous 256*k*k-float segment of
//    the master goes through LDS and comes out tap-major;
//  * data-gradient images ([ci][t][co]) swap the two channel axes: an LDS tile of 64 co x 16 ci (64 for 1x1)
//    x all KH*KW taps -- each co row of the tile is ONE contiguous segment of the master -- written with co
//    contiguous.  (The element-wise form read a different cache line per lane: 1.2 GB of HBM
//    fetches per launch for 0.1 GB of weights, rocprofv3 FETCH_SIZE, profiles/pmc_traffic.json r01e.)
constexpr int PACK_CO = 64, PACK_KK = 9, PACK_ROW = 144;       // LDS tile: 64 co rows of <= 144 floats (+1 pad)
constexpr int PACK_FWD_CI = 256;                                // forward tile: 1 co x 256 ci x KH*KW taps

// data-gradient image, dense conv, KH*KW <= 9: tile of 64 co x (64 ci for 1x1, 16 ci otherwise)
__host__ __device__ inline bool pack_job_tiled(const cnk_pack_job& j) {
    return j.transpose && j.groups <= 1 && j.KH * j.KW <= PACK_KK;
}
__host__ __device__ inline int pack_tile_ci(const cnk_pack_job& j) { return j.KH * j.KW == 1 ? 64 : 16; }
// forward image of a dense k x k conv (k > 1): one co row, 256 ci per block, taps re-ordered through LDS
__host__ __device__ inline bool pack_job_fwd_rows(const cnk_pack_job& j) {
    return !j.transpose && j.groups <= 1 && j.KH * j.KW > 1 && j.KH * j.KW <= PACK_KK;
}

__global__ __launch_bounds__(256) void pack_batch_kernel(const cnk_pack_job* __restrict__ jobs,
                                                         const int2* __restrict__ block_map) {
    __shared__ float tile[PACK_CO][PACK_ROW + 1];
    const int2 bm = block_map[blockIdx.x];
    // the job record in LDS: read through the global pointer, every field (and tap) would be re-loaded after each
    // store to `out`, which the compiler must assume may alias it
    __shared__ cnk_pack_job sj;
    {
        static_assert(sizeof(cnk_pack_job) % 4 == 0, "cnk_pack_job is copied in dwords");
        const unsigned int* src = reinterpret_cast<const unsigned int*>(jobs + bm.x);
        unsigned int* dst = reinterpret_cast<unsigned int*>(&sj);
        if (threadIdx.x < sizeof(cnk_pack_job) / 4) dst[threadIdx.x] = src[threadIdx.x];
    }
    __syncthreads();
    const cnk_pack_job& j = sj;
    const int lb = blockIdx.x - bm.y;
    const int KK = j.KH * j.KW;
    if (pack_job_tiled(j)) {
        const int TCI = pack_tile_ci(j), tiles_co = (j.Cout + PACK_CO - 1) / PACK_CO;
        const int co0 = (lb % tiles_co) * PACK_CO, ci0 = (lb / tiles_co) * TCI;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int nci = min(TCI, j.Cin - ci0), seg = nci * KK;               // contiguous floats per co row
        for (int r = wave; r < PACK_CO; r += 4) {
            const int co = co0 + r;
            if (co < j.Cout) {
                const float* src = j.w + ((size_t)co * j.Cin + ci0) * KK;
                for (int k = lane; k < seg; k += 64) tile[r][k] = src[k];
            }
        }
        __syncthreads();
        const int co = co0 + lane;
        const int pairs = nci * j.ntaps;                                      // (ci, tap) rows of the image
        for (int q = wave; q < pairs; q += 4) {
            const int ci = q / j.ntaps, t = q - ci * j.ntaps;
            if (co < j.Cout) {
                const float v = tile[lane][ci * KK + j.kh[t] * j.KW + j.kw[t]];
                const size_t o = ((size_t)(ci0 + ci) * j.ntaps + t) * j.Cout + co;
                if (j.dtype == CNK_BF16) ((bf16*)j.out)[o] = (bf16)v;
                else ((float*)j.out)[o] = v;
            }
        }
        return;
    }
    if (pack_job_fwd_rows(j)) {
        float* row = &tile[0][0];                                             // 256 * 9 floats fit the tile
        const int chunks = (j.Cin + PACK_FWD_CI - 1) / PACK_FWD_CI;
        const int co = lb / chunks, ci0 = (lb - co * chunks) * PACK_FWD_CI;
        const int nci = min(PACK_FWD_CI, j.Cin - ci0), seg = nci * KK;
        const float* src = j.w + ((size_t)co * j.Cin + ci0) * KK;
        for (int k = threadIdx.x; k < seg; k += 256) row[k] = src[k];
        __syncthreads();
        const int ci = threadIdx.x;
        if (ci < nci)
            for (int t = 0; t < j.ntaps; ++t) {
                const float v = row[ci * KK + j.kh[t] * j.KW + j.kw[t]];
                const size_t o = ((size_t)co * j.ntaps + t) * j.Cin + ci0 + ci;
                if (j.dtype == CNK_BF16) ((bf16*)j.out)[o] = (bf16)v;
                else ((float*)j.out)[o] = v;
            }
        return;
    }
    const int R = j.transpose ? j.Cin : j.Cout, Cred = j.transpose ? j.Cout : j.Cin;
    // (one conv image has < 2^31 elements: 32-bit index arithmetic)
    const uint32_t total = (uint32_t)R * (uint32_t)j.ntaps * (uint32_t)Cred;
    const uint32_t base = (uint32_t)lb * 1024u;
    if (!j.transpose && j.groups <= 1 && KK == 1 && j.ntaps == 1 && j.dtype == CNK_BF16 && (total & 7u) == 0 &&
        (((uintptr_t)j.w | (uintptr_t)j.out) & 15) == 0) {
        // 1x1 forward image = the master's own order: a cast.  Eight elements per thread (two 16-byte loads, one 16-byte
        // store) instead of an index decomposition and a 2-byte store per element.
        const uint32_t i = base + threadIdx.x * 8;
        if (threadIdx.x < 128 && i < total) {
            const f32x4 a = *(const f32x4*)(j.w + i), b = *(const f32x4*)(j.w + i + 4);
            bf16x8 o = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
            *(bf16x8*)((bf16*)j.out + i) = o;
        }
        return;
    }
    const uint32_t ucred = (uint32_t)Cred, utaps = (uint32_t)j.ntaps;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t i = base + u * 256 + threadIdx.x;
        if (i < total) {
            const uint32_t q = i / ucred;
            const int c = (int)(i - q * ucred);
            const uint32_t rr = q / utaps;
            const int t = (int)(q - rr * utaps), r = (int)rr;
            int co = j.transpose ? c : r, ci = j.transpose ? r : c;
            float v = master_w(j.w, co, ci, j.Cout, j.Cin, j.KH, j.KW, j.kh[t], j.kw[t], j.groups);
            if (j.dtype == CNK_BF16) ((bf16*)j.out)[i] = (bf16)v;
            else ((float*)j.out)[i] = v;
        }
    }
}

DirectParams make_params(const cnk_conv_desc* d) {
    DirectParams p = {};
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = d->Cout;
    p.KH = d->kh; p.KW = d->kw; p.sh = d->stride_h; p.sw = d->stride_w; p.ph = d->pad_h; p.pw = d->pad_w;
    return p;
}

}  // namespace

int pack_job_blocks(const cnk_pack_job& j) {
    if (pack_job_tiled(j)) return ceil_div(j.Cout, PACK_CO) * ceil_div(j.Cin, pack_tile_ci(j));
    if (pack_job_fwd_rows(j)) return j.Cout * ceil_div(j.Cin, PACK_FWD_CI);
    return (int)(((size_t)j.Cout * j.Cin * j.ntaps + 1023) / 1024);
}

int launch_wgrad_reduce(const float* slab, float* dw, int splits, size_t slab_stride, int Cout,
                        int Cin, int ntaps, size_t s_co, size_t s_t, size_t s_ci, int inner_is_ci,
                        int accumulate, hipStream_t s);

int direct_fwd(const cnk_conv_desc* d, const void* x, const void* w_t, const float* bias, void* y,
               hipStream_t s) {
    DirectParams p = make_params(d);
    p.a = x; p.b = w_t; p.out = y; p.bias = bias;
    p.M = d->N * p.Ho * p.Wo;
    p.fd_a = make_fastdiv(p.Ho * p.Wo); p.fd_b = make_fastdiv(p.Wo);
    dim3 block(64, 4), grid(ceil_div(p.M, 4), ceil_div(d->Cout, 64));
    if (d->dtype == CNK_BF16) hipLaunchKernelGGL(direct_fwd_kernel<bf16>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(direct_fwd_kernel<float>, grid, block, 0, s, p);
    CNK_CHECK_LAUNCH("direct_fwd");
    return 0;
}

int direct_dgrad(const cnk_conv_desc* d, const void* dy, const void* w_t, void* dx, const void* addend,
                 hipStream_t s) {
    DirectParams p = make_params(d);
    p.a = dy; p.b = w_t; p.out = dx; p.addend = addend;
    p.M = d->N * d->H * d->W;
    p.fd_a = make_fastdiv(d->H * d->W); p.fd_b = make_fastdiv(d->W);
    dim3 block(64, 4), grid(ceil_div(p.M, 4), ceil_div(d->Cin, 64));
    if (d->dtype == CNK_BF16) hipLaunchKernelGGL(direct_dgrad_kernel<bf16>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(direct_dgrad_kernel<float>, grid, block, 0, s, p);
    CNK_CHECK_LAUNCH("direct_dgrad");
    return 0;
}

static int direct_wgrad_slices(const cnk_conv_desc* d, int* chunk) {
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    int M = d->N * Ho * Wo;
    int total = d->kh * d->kw * d->Cin * d->Cout;
    int blocks = ceil_div(total, 256);
    int want = ceil_div(2048, blocks);
    int c = ceil_div(M, want);
    if (c < 64) c = 64;
    *chunk = c;
    return ceil_div(M, c);
}
size_t direct_wgrad_workspace(const cnk_conv_desc* d) {
    int chunk;
    int slices = direct_wgrad_slices(d, &chunk);
    return (size_t)slices * d->kh * d->kw * d->Cin * d->Cout * sizeof(float);
}
int direct_wgrad(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate,
                 void* ws, size_t ws_bytes, hipStream_t s) {
    DirectParams p = make_params(d);
    int slices = direct_wgrad_slices(d, &p.chunk);
    int total = d->kh * d->kw * d->Cin * d->Cout;
    size_t need = (size_t)slices * total * sizeof(float);
    CNK_CHECK_ARG(ws && ws_bytes >= need, "conv2d_wgrad(direct): workspace too small (%zu < %zu)", ws_bytes, need);
    p.a = dy; p.b = x; p.out = ws;
    p.M = d->N * p.Ho * p.Wo;
    p.fd_a = make_fastdiv(p.Ho * p.Wo); p.fd_b = make_fastdiv(p.Wo);
    dim3 block(256), grid(ceil_div(total, 256), slices);
    if (d->dtype == CNK_BF16) hipLaunchKernelGGL(direct_wgrad_kernel<bf16>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(direct_wgrad_kernel<float>, grid, block, 0, s, p);
    CNK_CHECK_LAUNCH("direct_wgrad");
    int T_ = d->kh * d->kw;
    // slab order [t][ci][co]
    return launch_wgrad_reduce((const float*)ws, dw, slices, (size_t)total, d->Cout, d->Cin, T_,
                               (size_t)1, (size_t)d->Cin * d->Cout, (size_t)d->Cout, 0, accumulate, s);
}

// dbias[c] (+)= sum_m dy[m][c]; workspace: colsum_workspace(M, C)
static int colsum_slices(int64_t M, int* chunk) {
    int64_t c = ceil_div64(M, 512);
    if (c < 64) c = 64;
    *chunk = (int)c;
    return (int)ceil_div64(M, c);
}
size_t colsum_workspace(int64_t M, int C) {
    int chunk;
    return (size_t)colsum_slices(M, &chunk) * C * sizeof(float);
}
int colsum(const void* a, float* out, int64_t M, int C, int dtype, int accumulate, void* ws,
           size_t ws_bytes, hipStream_t s) {
    int chunk;
    int slices = colsum_slices(M, &chunk);
    CNK_CHECK_ARG(ws && ws_bytes >= (size_t)slices * C * sizeof(float), "colsum: workspace too small");
    dim3 block(64), grid(ceil_div(C, 64), slices);
    if (dtype == CNK_BF16)
        hipLaunchKernelGGL(colsum_partial_kernel<bf16>, grid, block, 0, s, (const bf16*)a, (float*)ws, (int)M, C, chunk);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, block, 0, s, (const float*)a, (float*)ws, (int)M, C, chunk);
    CNK_CHECK_LAUNCH("colsum_partial");
    hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, s, (const float*)ws, out, slices, C, accumulate);
    CNK_CHECK_LAUNCH("colsum_final");
    return 0;
}

int pack_direct(const cnk_conv_desc* d, const float* w, void* out, hipStream_t s) {
    size_t total = (size_t)d->Cout * d->Cin * d->kh * d->kw;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(pack_direct_kernel, dim3(grid), dim3(256), 0, s, w, (float*)out, d->Cout, d->Cin, d->kh * d->kw);
    CNK_CHECK_LAUNCH("pack_direct");
    return 0;
}

// taps: ntaps pairs (kh,kw); transpose=0 -> [Cout][t][Cin], 1 -> [Cin][t][Cout]
int pack_taps(const cnk_conv_desc* d, const float* w, void* out, int ntaps, const signed char* kh,
              const signed char* kw, int transpose, hipStream_t s) {
    if (ntaps == 0) return 0;
    PackParams p = {};
    p.w = w; p.out = out; p.Cout = d->Cout; p.Cin = d->Cin; p.KH = d->kh; p.KW = d->kw;
    p.ntaps = ntaps; p.transpose = transpose; p.groups = conv_groups(d);
    for (int t = 0; t < ntaps; ++t) { p.kh[t] = kh[t]; p.kw[t] = kw[t]; }
    const int kstep = d->dtype == CNK_BF16 ? 64 : 32;
    const int R = transpose ? d->Cin : d->Cout, Cred = transpose ? d->Cout : d->Cin;
    p.Rpad = ceil_div(R, 64) * 64;                       // == R, Cred on the tile grid
    p.Cpad = ceil_div(Cred, kstep) * kstep;
    size_t total = (size_t)p.Rpad * p.Cpad * ntaps;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (d->dtype == CNK_BF16) hipLaunchKernelGGL(pack_kernel<bf16>, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(pack_kernel<float>, dim3(grid), dim3(256), 0, s, p);
    CNK_CHECK_LAUNCH("pack_taps");
    return 0;
}

int pack_batch(const cnk_pack_job* jobs_dev, const void* block_map_dev, int nblocks, hipStream_t s) {
    if (nblocks <= 0) return 0;
    hipLaunchKernelGGL(pack_batch_kernel, dim3(nblocks), dim3(256), 0, s, jobs_dev, (const int2*)block_map_dev);
    CNK_CHECK_LAUNCH("pack_batch");
    return 0;
}
