// Depthwise convolution (groups == Cin == Cout, channel multiplier 1) on NHWC activations:
// conv2d_depthwise of the reference's layer factory (src/basemodel.py:981-997), the spatial half of
// MobileNet-v1's Conv2dBlockDW (src/mobilenet_v1.py:92-125).  SURVEY.md 8f n4.
//
// There is no reduction over channels, so this is HBM-bound vector work, not a GEMM: a thread owns
// one 16-byte channel vector (8 bf16 / 4 fp32 channels) with its taps' weights in registers and
// walks a strip of pixels; lanes run along C, so every load and store is a coalesced row segment.
//   forward   y[m][c]   = sum_t x[pix(m,t)][c] * w[c][t]  (+ bias[c])
//   dgrad     dx[p][c]  = sum_t dy[opix(p,t)][c] * w[c][t]      (positions that divide by the stride)
//   wgrad     dw[c][t]  = sum_m dy[m][c] * x[pix(m,t)][c]       per-slice partials, fixed-order sum
// Weights are read from the fp32 master tensor [C][1][kh][kw] directly (no packed image).
#include "common.h"

namespace {

constexpr int DW_MAX_TAPS = 25;

struct DwParams {
    const void* a;        // fwd: x      dgrad: dy     wgrad: dy
    const void* b;        // fwd: -      dgrad: -      wgrad: x
    const float* w;       // [C][taps]
    const float* bias;
    void* out;            // fwd: y      dgrad: dx
    float* partial;       // wgrad: [slices][taps][C]
    int N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw;
    int cols, cpb, rpb;   // vector columns, columns / rows per block
    int64_t M;            // pixels walked (fwd/wgrad: N*Ho*Wo, dgrad: N*H*W)
    int64_t slice;        // pixels per block
    FastDiv fd_hw, fd_w;  // decomposition of the walked pixel index
};

// VEC = elements per 16-byte vector, or 1 (scalar path for channel counts off the vector width)
template <typename T, int VEC> struct RawV {
    using V = decltype(Vec16<T>::v);
    V v;
    __device__ __forceinline__ void load(const T* p) { v = *(const V*)p; }
    __device__ __forceinline__ void store(T* p) const { *(V*)p = v; }
    __device__ __forceinline__ float get(int i) const { return to_f32(v[i]); }
    __device__ __forceinline__ void set(int i, float f) { v[i] = from_f32<T>(f); }
};
template <typename T> struct RawV<T, 1> {
    T v;
    __device__ __forceinline__ void load(const T* p) { v = *p; }
    __device__ __forceinline__ void store(T* p) const { *p = v; }
    __device__ __forceinline__ float get(int) const { return to_f32(v); }
    __device__ __forceinline__ void set(int, float f) { v = from_f32<T>(f); }
};

// TAPS > 0: compile-time tap count with the weights in registers (3x3: 9); TAPS == 0: run-time
template <typename T, int VEC, int TAPS, bool DGRAD>
__global__ __launch_bounds__(256) void dwconv_kernel(const DwParams p) {
    const int tid = threadIdx.x, tcol = tid % p.cpb, trow = tid / p.cpb;
    const int vcol = blockIdx.y * p.cpb + tcol;
    if (trow >= p.rpb || vcol >= p.cols) return;
    const int c0 = vcol * VEC;
    const int ntaps = TAPS > 0 ? TAPS : p.kh * p.kw;
    float wr[TAPS > 0 ? TAPS : 1][VEC];
    if constexpr (TAPS > 0) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int e = 0; e < VEC; ++e) wr[t][e] = p.w[(size_t)(c0 + e) * TAPS + t];
    }
    float bs[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) bs[e] = (!DGRAD && p.bias) ? p.bias[c0 + e] : 0.f;
    const T* src = (const T*)p.a;
    T* dst = (T*)p.out;
    const int64_t m0 = (int64_t)blockIdx.x * p.slice;
    const int64_t m1 = m0 + p.slice < p.M ? m0 + p.slice : p.M;
    const int SH = DGRAD ? p.Ho : p.H, SW = DGRAD ? p.Wo : p.W;     // source grid
    for (int64_t m = m0 + trow; m < m1; m += p.rpb) {
        const uint32_t img = fd_div((uint32_t)m, p.fd_hw), rem = (uint32_t)m - img * p.fd_hw.d;
        const int oh = (int)fd_div(rem, p.fd_w), ow = (int)(rem - (uint32_t)oh * p.fd_w.d);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = bs[e];
        auto tap = [&](int t, int r, int q) {
            int ih, iw;
            bool ok;
            if constexpr (DGRAD) {
                const int th = oh + p.ph - r, tw = ow + p.pw - q;
                ok = th >= 0 && tw >= 0 && th % p.sh == 0 && tw % p.sw == 0;
                ih = th / p.sh; iw = tw / p.sw;
            } else {
                ih = oh * p.sh - p.ph + r; iw = ow * p.sw - p.pw + q;
                ok = true;
            }
            if (ok && (unsigned)ih < (unsigned)SH && (unsigned)iw < (unsigned)SW) {
                RawV<T, VEC> v;
                v.load(src + ((size_t)(img * SH + ih) * SW + iw) * p.C + c0);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float wv = TAPS > 0 ? wr[TAPS > 0 ? t : 0][e] : p.w[(size_t)(c0 + e) * ntaps + t];
                    acc[e] += v.get(e) * wv;
                }
            }
        };
        if constexpr (TAPS == 9) {
            // all nine loads are issued before the first FMA: taps that fall outside read the (valid) centre-safe
            // pixel 0 of the image and are multiplied away, so no load sits behind a branch
            RawV<T, VEC> tv[9];
            bool tok[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int r = t / 3, q = t % 3;
                int ih, iw;
                bool ok;
                if constexpr (DGRAD) {
                    const int th = oh + p.ph - r, tw = ow + p.pw - q;
                    ok = th >= 0 && tw >= 0 && th % p.sh == 0 && tw % p.sw == 0;
                    ih = th / p.sh; iw = tw / p.sw;
                } else {
                    ih = oh * p.sh - p.ph + r; iw = ow * p.sw - p.pw + q;
                    ok = true;
                }
                ok = ok && (unsigned)ih < (unsigned)SH && (unsigned)iw < (unsigned)SW;
                tok[t] = ok;
                const size_t pix = ok ? (size_t)(img * SH + ih) * SW + iw : (size_t)img * SH * SW;
                tv[t].load(src + pix * p.C + c0);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += (tok[t] ? tv[t].get(e) : 0.f) * wr[t][e];
        } else {
            for (int r = 0, t = 0; r < p.kh; ++r)
                for (int q = 0; q < p.kw; ++q, ++t) tap(t, r, q);
        }
        RawV<T, VEC> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
        o.store(dst + (size_t)m * p.C + c0);
    }
}

// wgrad partials: thread (column, row-thread) accumulates dy*x for its pixels and taps, rows of a
// block are summed through LDS in fixed order -> partial[blockIdx.x][t][c]
template <typename T, int VEC, int TAPS>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const DwParams p) {
    extern __shared__ float sh[];                       // [256][VEC]
    const int tid = threadIdx.x, tcol = tid % p.cpb, trow = tid / p.cpb;
    const int vcol = blockIdx.y * p.cpb + tcol;
    const bool active = trow < p.rpb && vcol < p.cols;
    const int c0 = vcol * VEC;
    float acc[TAPS][VEC];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
    const T* dy = (const T*)p.a;
    const T* x = (const T*)p.b;
    const int64_t m0 = (int64_t)blockIdx.x * p.slice;
    const int64_t m1 = m0 + p.slice < p.M ? m0 + p.slice : p.M;
    if (active) {
        for (int64_t m = m0 + trow; m < m1; m += p.rpb) {
            const uint32_t img = fd_div((uint32_t)m, p.fd_hw), rem = (uint32_t)m - img * p.fd_hw.d;
            const int oh = (int)fd_div(rem, p.fd_w), ow = (int)(rem - (uint32_t)oh * p.fd_w.d);
            RawV<T, VEC> d;
            d.load(dy + (size_t)m * p.C + c0);
            // every tap's load is issued before the first FMA (out-of-image taps read pixel 0 of the image and
            // are selected away): no load behind a branch
            RawV<T, VEC> tv[TAPS];
            bool tok[TAPS];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int r = t / p.kw, q = t - r * p.kw;
                const int ih = oh * p.sh - p.ph + r, iw = ow * p.sw - p.pw + q;
                tok[t] = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const size_t pix = tok[t] ? (size_t)(img * p.H + ih) * p.W + iw : (size_t)img * p.H * p.W;
                tv[t].load(x + pix * p.C + c0);
            }
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[t][e] += d.get(e) * (tok[t] ? tv[t].get(e) : 0.f);
        }
    }
    int span = 1;
    while (span < p.rpb) span <<= 1;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) sh[tid * VEC + e] = acc[t][e];
        __syncthreads();
        for (int st = span >> 1; st >= 1; st >>= 1) {
            if (trow < st && trow + st < p.rpb && vcol < p.cols) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) sh[tid * VEC + e] += sh[(tid + st * p.cpb) * VEC + e];
            }
            __syncthreads();
        }
        if (trow == 0 && vcol < p.cols) {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                p.partial[((size_t)blockIdx.x * TAPS + t) * p.C + c0 + e] = sh[tid * VEC + e];
        }
        __syncthreads();
    }
}

// generic tap count: one launch per tap with TAPS == 1 semantics is avoided by looping taps here
template <typename T, int VEC>
__global__ __launch_bounds__(256) void dwconv_wgrad_generic_kernel(const DwParams p) {
    extern __shared__ float sh[];
    const int tid = threadIdx.x, tcol = tid % p.cpb, trow = tid / p.cpb;
    const int vcol = blockIdx.y * p.cpb + tcol;
    const bool active = trow < p.rpb && vcol < p.cols;
    const int c0 = vcol * VEC, ntaps = p.kh * p.kw;
    const T* dy = (const T*)p.a;
    const T* x = (const T*)p.b;
    const int64_t m0 = (int64_t)blockIdx.x * p.slice;
    const int64_t m1 = m0 + p.slice < p.M ? m0 + p.slice : p.M;
    int span = 1;
    while (span < p.rpb) span <<= 1;
    for (int t = 0; t < ntaps; ++t) {
        const int r = t / p.kw, q = t - r * p.kw;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        if (active) {
            for (int64_t m = m0 + trow; m < m1; m += p.rpb) {
                const uint32_t img = fd_div((uint32_t)m, p.fd_hw), rem = (uint32_t)m - img * p.fd_hw.d;
                const int oh = (int)fd_div(rem, p.fd_w), ow = (int)(rem - (uint32_t)oh * p.fd_w.d);
                const int ih = oh * p.sh - p.ph + r, iw = ow * p.sw - p.pw + q;
                if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
                    RawV<T, VEC> d, v;
                    d.load(dy + (size_t)m * p.C + c0);
                    v.load(x + ((size_t)(img * p.H + ih) * p.W + iw) * p.C + c0);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += d.get(e) * v.get(e);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) sh[tid * VEC + e] = acc[e];
        __syncthreads();
        for (int st = span >> 1; st >= 1; st >>= 1) {
            if (trow < st && trow + st < p.rpb && vcol < p.cols) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) sh[tid * VEC + e] += sh[(tid + st * p.cpb) * VEC + e];
            }
            __syncthreads();
        }
        if (trow == 0 && vcol < p.cols) {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                p.partial[((size_t)blockIdx.x * ntaps + t) * p.C + c0 + e] = sh[tid * VEC + e];
        }
        __syncthreads();
    }
}

// dw[c][t] (+)= sum_slices partial[s][t][c], fixed order
__global__ __launch_bounds__(256) void dwconv_wgrad_sum_kernel(const float* __restrict__ partial, int slices, int taps, int C,
                                                               float* __restrict__ dw, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // over taps*C, c fastest (coalesced reads)
    if (i >= taps * C) return;
    const int t = i / C, c = i - t * C;
    float a0 = 0.f, a1 = 0.f;
    int s = 0;
    for (; s + 1 < slices; s += 2) {
        a0 += partial[((size_t)s * taps + t) * C + c];
        a1 += partial[((size_t)(s + 1) * taps + t) * C + c];
    }
    if (s < slices) a0 += partial[((size_t)s * taps + t) * C + c];
    float* o = dw + (size_t)c * taps + t;
    *o = accumulate ? *o + (a0 + a1) : (a0 + a1);
}

int dw_vec(int dtype, int C) {
    const int v = dtype == CNK_BF16 ? 8 : 4;
    return C % v == 0 ? v : 1;
}

void dw_geom(DwParams& p, int64_t M, int C, int vec, int target_blocks) {
    p.cols = C / vec;
    p.cpb = p.cols < 32 ? p.cols : 32;
    p.rpb = 256 / p.cpb;
    int colgroups = ceil_div(p.cols, p.cpb);
    int64_t want = target_blocks / colgroups;
    if (want < 1) want = 1;
    int64_t sl = ceil_div64(M, want);
    int64_t per = ceil_div64(sl, p.rpb);
    if (per < 4) per = 4;
    p.slice = per * p.rpb;
    p.M = M;
}

}  // namespace

bool dwconv_supported(const cnk_conv_desc* d) { return d->Cin == d->Cout && d->kh * d->kw <= DW_MAX_TAPS; }

static DwParams dw_params(const cnk_conv_desc* d) {
    DwParams p = {};
    cnk_conv2d_out_hw(d, &p.Ho, &p.Wo);
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->Cin; p.kh = d->kh; p.kw = d->kw;
    p.sh = d->stride_h; p.sw = d->stride_w; p.ph = d->pad_h; p.pw = d->pad_w;
    return p;
}

template <bool DGRAD> static void dw_launch(const cnk_conv_desc* d, const DwParams& p, dim3 grid, hipStream_t s) {
    const bool k3 = d->kh == 3 && d->kw == 3, vec = dw_vec(d->dtype, d->Cin) > 1;
#define CNK_DW(T, V)                                                                                     \
    do {                                                                                                 \
        if (k3) hipLaunchKernelGGL((dwconv_kernel<T, V, 9, DGRAD>), grid, dim3(256), 0, s, p);          \
        else hipLaunchKernelGGL((dwconv_kernel<T, V, 0, DGRAD>), grid, dim3(256), 0, s, p);             \
    } while (0)
    if (d->dtype == CNK_BF16) { if (vec) CNK_DW(bf16, 8); else CNK_DW(bf16, 1); }
    else { if (vec) CNK_DW(float, 4); else CNK_DW(float, 1); }
#undef CNK_DW
}

int dwconv_fwd(const cnk_conv_desc* d, const void* x, const float* w, const float* bias, void* y, hipStream_t s) {
    DwParams p = dw_params(d);
    p.a = x; p.w = w; p.bias = bias; p.out = y;
    dw_geom(p, (int64_t)d->N * p.Ho * p.Wo, d->Cin, dw_vec(d->dtype, d->Cin), 4096);
    if (p.M == 0) return 0;
    p.fd_hw = make_fastdiv(p.Ho * p.Wo);
    p.fd_w = make_fastdiv(p.Wo);
    dim3 grid((unsigned)ceil_div64(p.M, p.slice), ceil_div(p.cols, p.cpb));
    dw_launch<false>(d, p, grid, s);
    CNK_CHECK_LAUNCH("dwconv_fwd");
    return 0;
}

int dwconv_dgrad(const cnk_conv_desc* d, const void* dy, const float* w, void* dx, hipStream_t s) {
    DwParams p = dw_params(d);
    p.a = dy; p.w = w; p.out = dx;
    dw_geom(p, (int64_t)d->N * d->H * d->W, d->Cin, dw_vec(d->dtype, d->Cin), 4096);
    if (p.M == 0) return 0;
    p.fd_hw = make_fastdiv(d->H * d->W);
    p.fd_w = make_fastdiv(d->W);
    dim3 grid((unsigned)ceil_div64(p.M, p.slice), ceil_div(p.cols, p.cpb));
    dw_launch<true>(d, p, grid, s);
    CNK_CHECK_LAUNCH("dwconv_dgrad");
    return 0;
}

static int dw_wgrad_blocks() { return 512; }      // slices to sum afterwards: keep them few

size_t dwconv_wgrad_workspace(const cnk_conv_desc* d) {
    DwParams p = dw_params(d);
    dw_geom(p, (int64_t)d->N * p.Ho * p.Wo, d->Cin, dw_vec(d->dtype, d->Cin), dw_wgrad_blocks());
    size_t slices = (size_t)ceil_div64(p.M > 0 ? p.M : 1, p.slice);
    return slices * d->kh * d->kw * d->Cin * sizeof(float);
}

int dwconv_wgrad(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate, float* ws,
                 hipStream_t s) {
    DwParams p = dw_params(d);
    p.a = dy; p.b = x; p.partial = ws;
    const int vec = dw_vec(d->dtype, d->Cin), taps = d->kh * d->kw;
    dw_geom(p, (int64_t)d->N * p.Ho * p.Wo, d->Cin, vec, dw_wgrad_blocks());
    if (p.M == 0) return 0;
    p.fd_hw = make_fastdiv(p.Ho * p.Wo);
    p.fd_w = make_fastdiv(p.Wo);
    const int slices = (int)ceil_div64(p.M, p.slice);
    dim3 grid(slices, ceil_div(p.cols, p.cpb));
    const size_t lds = 256 * vec * sizeof(float);
    const bool k3 = d->kh == 3 && d->kw == 3;
#define CNK_DWW(T, V)                                                                                    \
    do {                                                                                                 \
        if (k3) hipLaunchKernelGGL((dwconv_wgrad_kernel<T, V, 9>), grid, dim3(256), lds, s, p);         \
        else hipLaunchKernelGGL((dwconv_wgrad_generic_kernel<T, V>), grid, dim3(256), lds, s, p);       \
    } while (0)
    if (d->dtype == CNK_BF16) { if (vec > 1) CNK_DWW(bf16, 8); else CNK_DWW(bf16, 1); }
    else { if (vec > 1) CNK_DWW(float, 4); else CNK_DWW(float, 1); }
#undef CNK_DWW
    CNK_CHECK_LAUNCH("dwconv_wgrad");
    hipLaunchKernelGGL(dwconv_wgrad_sum_kernel, dim3(ceil_div(taps * d->Cin, 256)), dim3(256), 0, s, ws, slices, taps,
                       d->Cin, dw, accumulate);
    CNK_CHECK_LAUNCH("dwconv_wgrad_sum");
    return 0;
}

// ---- general grouped conv through the dense kernels, and ShuffleNet's channel shuffle ------------------------
// A grouped conv whose shape the chunked implicit-GEMM path cannot take (Cin != Cout or a group width that does
// not divide 64: ShuffleNet's grouped 1x1 convs, shufflenet_v1.py:117-141) runs as the DENSE conv of its
// block-diagonal weight: expand [Cout][Cin/g][T] -> [Cout][Cin][T] (zeros off the diagonal) before the conv, extract
// the diagonal blocks of the dense weight gradient after it.
namespace {
__global__ __launch_bounds__(256) void grouped_expand_kernel(const float* __restrict__ w, float* __restrict__ wd, int Cout,
                                                             int Cin, int groups, int T) {
    const size_t total = (size_t)Cout * Cin * T;
    const int gw = Cin / groups, go = Cout / groups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const size_t r = i / T;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        const int l = ci - (co / go) * gw;
        wd[i] = (unsigned)l < (unsigned)gw ? w[((size_t)co * gw + l) * T + t] : 0.f;
    }
}
__global__ __launch_bounds__(256) void grouped_extract_kernel(const float* __restrict__ dwd, float* __restrict__ dw, int Cout,
                                                              int Cin, int groups, int T) {
    const int gw = Cin / groups, go = Cout / groups;
    const size_t total = (size_t)Cout * gw * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const size_t r = i / T;
        const int l = (int)(r % gw), co = (int)(r / gw);
        dw[i] = dwd[((size_t)co * Cin + (co / go) * gw + l) * T + t];
    }
}
// y[p][j*g + i] = x[p][i*(C/g) + j]  (view(N, g, C/g, H, W).transpose(1, 2), shufflenet_v1.py:173-186) on NHWC rows
template <typename T>
__global__ __launch_bounds__(256) void channel_shuffle_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t total,
                                                              int C, int groups) {
    const int cg = C / groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % C);
        const int64_t p = i / C;
        const int j = co / groups, g = co - j * groups;
        y[i] = x[p * C + (int64_t)g * cg + j];
    }
}
inline int grid_g(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096);
}
}  // namespace

int grouped_expand(const float* w, float* wd, int Cout, int Cin, int groups, int T, hipStream_t s) {
    hipLaunchKernelGGL(grouped_expand_kernel, dim3(grid_g((int64_t)Cout * Cin * T)), dim3(256), 0, s, w, wd, Cout, Cin, groups, T);
    CNK_CHECK_LAUNCH("grouped_expand");
    return 0;
}
int grouped_extract(const float* dwd, float* dw, int Cout, int Cin, int groups, int T, hipStream_t s) {
    hipLaunchKernelGGL(grouped_extract_kernel, dim3(grid_g((int64_t)Cout * (Cin / groups) * T)), dim3(256), 0, s, dwd, dw, Cout, Cin, groups, T);
    CNK_CHECK_LAUNCH("grouped_extract");
    return 0;
}
int channel_shuffle(const void* x, void* y, int64_t pixels, int C, int groups, int dtype, hipStream_t s) {
    const int64_t total = pixels * C;
    if (total == 0) return 0;
    if (dtype == CNK_BF16) hipLaunchKernelGGL(channel_shuffle_kernel<bf16>, dim3(grid_g(total)), dim3(256), 0, s, (const bf16*)x, (bf16*)y, total, C, groups);
    else hipLaunchKernelGGL(channel_shuffle_kernel<float>, dim3(grid_g(total)), dim3(256), 0, s, (const float*)x, (float*)y, total, C, groups);
    CNK_CHECK_LAUNCH("channel_shuffle");
    return 0;
}
