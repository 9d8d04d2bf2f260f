// Streaming (HBM-bound) kernels of the path: max-pool, ReLU, add+ReLU, global average pool,
// NCHW->NHWC hand-off at the model boundary, flat-arena Adam.  16-byte vector access along C.
#include "common.h"

namespace {

template <typename T, int VEC> struct VL {
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

// one thread's raw 16-byte payload (no conversion at load time)
template <typename T, int VEC> struct Raw16 {
    decltype(Vec16<T>::v) v;
    __device__ __forceinline__ void load(const T* p) { v = *(const decltype(v)*)p; }
    __device__ __forceinline__ void store(T* p) const { *(decltype(v)*)p = v; }
    __device__ __forceinline__ float get(int i) const { return to_f32(v[i]); }
    __device__ __forceinline__ void set(int i, float f) { v[i] = from_f32<T>(f); }
};

struct PoolParams {
    int N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw;
    FastDiv fd_cols, fd_wo, fd_ho, fd_w, fd_h;
};

// AFFINE: the pooled tensor is act(x * scale[c] + shift[c]) -- a training-mode BatchNorm(+ReLU) applied on the fly (the
// stems: conv -> BN -> ReLU -> MaxPool2d), each value rounded to T as the stand-alone BN pass would have stored it.
template <typename T, int VEC, bool AFFINE = false>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          unsigned char* __restrict__ idx,
                                                          PoolParams p, int64_t total,
                                                          const float* __restrict__ scale = nullptr,
                                                          const float* __restrict__ shift = nullptr, int relu = 0,
                                                          T* __restrict__ xmax = nullptr) {
    const int cols = p.C / VEC;
    // blocks are dealt to the 8 XCDs round-robin: walk the tensor so that each XCD gets a CONTIGUOUS run of every grid
    // sweep -- neighbouring output rows share input rows, which then hit in that XCD's L2 (r02c FETCH_SIZE: 626 MB per
    // launch for a 411 MB input with the interleaved walk).  The stem pool itself is VALU-bound (9 taps x 8 channels of
    // affine + ReLU + rounding + arg-max per thread): its time did not move.
    const int g8 = gridDim.x & ~7;
    const int lb = (int)blockIdx.x < g8 ? ((int)blockIdx.x & 7) * (g8 >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    for (int64_t i = (int64_t)lb * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t pix = fd_div((uint32_t)i, p.fd_cols), vc = (uint32_t)i - pix * cols;
        uint32_t t = fd_div(pix, p.fd_wo), wo = pix - t * p.Wo;
        uint32_t n = fd_div(t, p.fd_ho), ho = t - n * p.Ho;
        float best[VEC], bx[VEC];
        unsigned char bi[VEC];
        bool any = false;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { best[k] = -INFINITY; bi[k] = 0; bx[k] = 0.f; }
        for (int r = 0; r < p.kh; ++r) {
            int h = (int)ho * p.sh - p.ph + r;
            if ((unsigned)h >= (unsigned)p.H) continue;
            for (int q = 0; q < p.kw; ++q) {
                int w = (int)wo * p.sw - p.pw + q;
                if ((unsigned)w >= (unsigned)p.W) continue;
                VL<T, VEC> v;
                v.load(x + ((size_t)(n * p.H + h) * p.W + w) * p.C + vc * VEC);
                float raw[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) raw[k] = v.f[k];
                if constexpr (AFFINE) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float a = v.f[k] * scale[vc * VEC + k] + shift[vc * VEC + k];
                        if (relu) a = a > 0.f ? a : 0.f;
                        v.f[k] = to_f32(from_f32<T>(a));
                    }
                }
                unsigned char pos = (unsigned char)(r * p.kw + q);
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    if (!any || v.f[k] > best[k] || v.f[k] != v.f[k]) { best[k] = v.f[k]; bi[k] = pos; bx[k] = raw[k]; }
                any = true;
            }
        }
        VL<T, VEC> o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.f[k] = best[k];
        o.store(y + (size_t)i * VEC);
        if (idx) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) idx[(size_t)i * VEC + k] = bi[k];
        }
        if constexpr (AFFINE) {
            if (xmax) {                                  // the INPUT value at the arg-max (cnk_maxpool2d_fwd_bn_keep)
                VL<T, VEC> ox;
#pragma unroll
                for (int k = 0; k < VEC; ++k) ox.f[k] = bx[k];
                ox.store(xmax + (size_t)i * VEC);
            }
        }
    }
}

// The stem pool of the ResNet family (3x3 / stride 2 over conv -> BN -> ReLU, src/resnet.py:67), round 3.  Same results as
// maxpool_fwd_kernel<T, VEC, true>, bit for bit (same affine, rounding, scan order and tie rule), but built as a streaming
// pass: all nine taps of an output vector are requested UNCONDITIONALLY (clamped addresses) before any arithmetic, the
// channel coefficients sit in registers for the thread's whole grid sweep (the sweep stride is a multiple of the vector
// columns, so a thread keeps its channels), and the eight arg-max bytes go out as one store.  The generic kernel ran the
// taps one dependent load at a time behind bounds branches and re-read scale / shift per tap: 270 us for a 411 MB input.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_k3s2_affine_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                  unsigned char* __restrict__ idx, PoolParams p, int64_t total,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  int relu, T* __restrict__ xmax) {
    const int cols = p.C / VEC;
    const int g8 = gridDim.x & ~7;
    const int lb = (int)blockIdx.x < g8 ? ((int)blockIdx.x & 7) * (g8 >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int64_t i0 = (int64_t)lb * blockDim.x + threadIdx.x;
    if (i0 >= total) return;
    const int vc = (int)(i0 % cols);                     // constant over the sweep: (gridDim.x * 256) % cols == 0 (host)
    float sc[VEC], sf[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { sc[k] = scale[vc * VEC + k]; sf[k] = shift[vc * VEC + k]; }
    for (int64_t i = i0; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t pix = fd_div((uint32_t)i, p.fd_cols);
        const uint32_t t = fd_div(pix, p.fd_wo), wo = pix - t * p.Wo;
        const uint32_t n = fd_div(t, p.fd_ho), ho = t - n * p.Ho;
        const int h0 = (int)ho * 2 - 1, w0 = (int)wo * 2 - 1;
        Raw16<T, VEC> v[9];
        unsigned okm = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int h = h0 + r, w = w0 + q;
                const bool ok = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                const int hc = ok ? h : (int)ho * 2, wc = ok ? w : (int)wo * 2;       // (2ho, 2wo) is always inside
                v[r * 3 + q].load(x + ((size_t)(n * p.H + hc) * p.W + wc) * p.C + vc * VEC);
                okm |= (ok ? 1u : 0u) << (r * 3 + q);
            }
        __builtin_amdgcn_sched_barrier(0);                // the nine loads stay in front of the arithmetic
        if constexpr (sizeof(T) == 2 && VEC == 8) {
            if (relu) {
                // bf16 after ReLU is non-negative, so its bit pattern orders like the value: one unsigned max per tap and channel
                // over the key (bits << 16) | (8 - tap) takes the largest ROUNDED value and, among equal ones, the first tap in
                // scan order -- the rule of the compare-and-select chain below (no NaN survives the ReLU) at half its VALU work
                // (the kernel is VALU-bound: 9 taps x 8 channels)
                unsigned key[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) key[k] = 0u;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const bool on = (okm >> tp) & 1u;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float a = v[tp].get(k) * sc[k] + sf[k];
                        a = a > 0.f ? a : 0.f;
                        const T r = from_f32<T>(a);
                        const unsigned kk = ((unsigned)__builtin_bit_cast(unsigned short, r) << 16) | (unsigned)(8 - tp);
                        const unsigned cand = on ? kk : 0u;
                        key[k] = cand > key[k] ? cand : key[k];
                    }
                }
                Raw16<T, VEC> o;
                u32x2 pk = {0u, 0u};
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    o.v[k] = __builtin_bit_cast(T, (unsigned short)(key[k] >> 16));
                    pk[k >> 2] |= (8u - (key[k] & 0xFFu)) << (8 * (k & 3));
                }
                o.store(y + (size_t)i * VEC);
                if (idx) *(u32x2*)(idx + (size_t)i * VEC) = pk;
                if (xmax) {
                    // the INPUT value at the arg-max (cnk_maxpool2d_fwd_bn_keep): one select per tap and channel by the winner's tap
                    Raw16<T, VEC> ox;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        const unsigned win = 8u - (key[k] & 0xFFu);
                        T xr = v[0].v[k];
#pragma unroll
                        for (int tp = 1; tp < 9; ++tp) xr = win == (unsigned)tp ? v[tp].v[k] : xr;
                        ox.v[k] = xr;
                    }
                    ox.store(xmax + (size_t)i * VEC);
                }
                continue;
            }
        }
        float best[VEC], bx[VEC];
        unsigned char bi[VEC];
        bool any = false;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { best[k] = -INFINITY; bi[k] = 0; bx[k] = 0.f; }
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            if (!((okm >> tp) & 1u)) continue;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float raw = v[tp].get(k);
                float a = raw * sc[k] + sf[k];
                if (relu) a = a > 0.f ? a : 0.f;
                const float f = to_f32(from_f32<T>(a));
                if (!any || f > best[k] || f != f) { best[k] = f; bi[k] = (unsigned char)tp; bx[k] = raw; }
            }
            any = true;
        }
        Raw16<T, VEC> o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.set(k, best[k]);
        o.store(y + (size_t)i * VEC);
        if (xmax) {
            Raw16<T, VEC> ox;
#pragma unroll
            for (int k = 0; k < VEC; ++k) ox.set(k, bx[k]);
            ox.store(xmax + (size_t)i * VEC);
        }
        if (idx) {
            if constexpr (VEC == 8) {
                u32x2 pk;
                pk[0] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
                pk[1] = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24);
                *(u32x2*)(idx + (size_t)i * VEC) = pk;
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) idx[(size_t)i * VEC + k] = bi[k];
            }
        }
    }
}

// nn.AvgPool2d (count_include_pad=True, floor mode): y = sum over the window's in-image pixels / (kh*kw)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, PoolParams p,
                                                          int64_t total) {
    const int cols = p.C / VEC;
    const float inv = 1.f / (float)(p.kh * p.kw);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t pix = fd_div((uint32_t)i, p.fd_cols), vc = (uint32_t)i - pix * cols;
        uint32_t t = fd_div(pix, p.fd_wo), wo = pix - t * p.Wo;
        uint32_t n = fd_div(t, p.fd_ho), ho = t - n * p.Ho;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        for (int r = 0; r < p.kh; ++r) {
            int h = (int)ho * p.sh - p.ph + r;
            if ((unsigned)h >= (unsigned)p.H) continue;
            for (int q = 0; q < p.kw; ++q) {
                int w = (int)wo * p.sw - p.pw + q;
                if ((unsigned)w >= (unsigned)p.W) continue;
                VL<T, VEC> v;
                v.load(x + ((size_t)(n * p.H + h) * p.W + w) * p.C + vc * VEC);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += v.f[k];
            }
        }
        VL<T, VEC> o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.f[k] = acc[k] * inv;
        o.store(y + (size_t)i * VEC);
    }
}

// dx[pixel] = sum over the windows that contain it of dy / (kh*kw)   (gather form, no atomics)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, PoolParams p,
                                                          int64_t total) {
    const int cols = p.C / VEC;
    const float inv = 1.f / (float)(p.kh * p.kw);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t pix = fd_div((uint32_t)i, p.fd_cols), vc = (uint32_t)i - pix * cols;
        uint32_t t = pix / p.W, w = pix - t * p.W;
        uint32_t n = t / p.H, h = t - n * p.H;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        for (int r = 0; r < p.kh; ++r) {
            int th = (int)h + p.ph - r;
            if (th < 0 || th % p.sh) continue;
            int ho = th / p.sh;
            if (ho >= p.Ho) continue;
            for (int q = 0; q < p.kw; ++q) {
                int tw = (int)w + p.pw - q;
                if (tw < 0 || tw % p.sw) continue;
                int wo = tw / p.sw;
                if (wo >= p.Wo) continue;
                VL<T, VEC> v;
                v.load(dy + ((size_t)(n * p.Ho + ho) * p.Wo + wo) * p.C + vc * VEC);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += v.f[k];
            }
        }
        VL<T, VEC> o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.f[k] = acc[k] * inv;
        o.store(dx + (size_t)i * VEC);
    }
}

// Gather form of the backward from the saved window positions (first maximum in row-major window
// order, torch's rule): every input element sums dy over the <= ceil(k/s)^2 windows that cover it
// and whose saved arg-max is this element.  No atomics, deterministic.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy,
                                                          const unsigned char* __restrict__ idx,
                                                          T* __restrict__ dx, PoolParams p,
                                                          int64_t total) {
    const int cols = p.C / VEC;
    const int g8 = gridDim.x & ~7;                               // XCD-contiguous walk, as in maxpool_fwd_kernel
    const int lb = (int)blockIdx.x < g8 ? ((int)blockIdx.x & 7) * (g8 >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    for (int64_t i = (int64_t)lb * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t pix = fd_div((uint32_t)i, p.fd_cols), vc = (uint32_t)i - pix * cols;
        uint32_t t = fd_div(pix, p.fd_w), w = pix - t * p.W;
        uint32_t n = fd_div(t, p.fd_h), h = t - n * p.H;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        int ho_lo = ((int)h + p.ph - p.kh + p.sh) / p.sh;     // ceil((h+p-k+1)/s) for non-negative
        if ((int)h + p.ph - p.kh + 1 <= 0) ho_lo = 0;
        int ho_hi = ((int)h + p.ph) / p.sh;
        if (ho_hi > p.Ho - 1) ho_hi = p.Ho - 1;
        int wo_lo = ((int)w + p.pw - p.kw + p.sw) / p.sw;
        if ((int)w + p.pw - p.kw + 1 <= 0) wo_lo = 0;
        int wo_hi = ((int)w + p.pw) / p.sw;
        if (wo_hi > p.Wo - 1) wo_hi = p.Wo - 1;
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const int pos = ((int)h - (ho * p.sh - p.ph)) * p.kw + ((int)w - (wo * p.sw - p.pw));
                const size_t o = ((size_t)(n * p.Ho + ho) * p.Wo + wo) * p.C + vc * VEC;
                VL<T, VEC> g;
                g.load(dy + o);
                unsigned char b[VEC];
                if constexpr (VEC == 8) {
                    u32x2 raw = *(const u32x2*)(idx + o);
                    __builtin_memcpy(b, &raw, 8);
                } else if constexpr (VEC == 4) {
                    unsigned int raw = *(const unsigned int*)(idx + o);
                    __builtin_memcpy(b, &raw, 4);
                } else {
                    b[0] = idx[o];
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    if ((int)b[k] == pos) acc[k] += g.f[k];
            }
        VL<T, VEC> o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.f[k] = acc[k];
        o.store(dx + (size_t)i * VEC);
    }
}

template <typename T, int VEC, int OP>   // OP 0: relu(a)  1: relu-bwd(dy=a, y=b)  2: relu(a+b)
__global__ __launch_bounds__(256) void eltwise_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                      T* __restrict__ o, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (int64_t)gridDim.x * blockDim.x) {
        VL<T, VEC> va, vb;
        va.load(a + i * VEC);
        if (OP != 0) vb.load(b + i * VEC);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if (OP == 0) va.f[k] = va.f[k] > 0.f ? va.f[k] : 0.f;
            if (OP == 1) va.f[k] = vb.f[k] > 0.f ? va.f[k] : 0.f;
            if (OP == 2) { float s = va.f[k] + vb.f[k]; va.f[k] = s > 0.f ? s : 0.f; }
        }
        va.store(o + i * VEC);
    }
}

template <typename T>
__global__ void gap_fwd_kernel(const T* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    int n = i / C, c = i - n * C;
    const T* p = x + (size_t)n * HW * C + c;
    float acc = 0.f;
    for (int k = 0; k < HW; ++k) acc += to_f32(p[(size_t)k * C]);
    y[i] = acc / (float)HW;
}
template <typename T>
__global__ void gap_bwd_kernel(const float* __restrict__ dy, T* __restrict__ dx, int N, int HW, int C) {
    size_t total = (size_t)N * HW * C;
    float inv = 1.f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        int n = (int)(i / ((size_t)HW * C));
        dx[i] = from_f32<T>(dy[(size_t)n * C + c] * inv);
    }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C,
                                    int H, int W) {
    size_t total = (size_t)N * C * H * W, hw = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        size_t pix = i / C;
        size_t n = pix / hw, s = pix - n * hw;
        y[i] = from_f32<T>(x[(n * C + c) * hw + s]);
    }
}

// Few input planes (the image batch at the model boundary: C <= 4) and H*W a multiple of 4: a thread takes FOUR consecutive
// pixels -- one 16-byte load per plane (a wave reads 1 KiB runs of each plane) and 4*C consecutive output elements
// (the element-per-thread form read 84-byte pieces of three planes per wave and wrote 2 bytes per lane: 2.6 TB/s).
template <typename T, int C>
__global__ __launch_bounds__(256) void nchw_to_nhwc_planes_kernel(const float* __restrict__ x, T* __restrict__ y, int64_t quads,
                                                                  int64_t hw4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / hw4, q = i - n * hw4;            // image, pixel quad inside it
        const float* src = x + (n * C * hw4 + q) * 4;
        f32x4 v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = *(const f32x4*)(src + (int64_t)c * hw4 * 4);
        T o[4 * C];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < C; ++c) o[p * C + c] = from_f32<T>(v[c][p]);
        T* dst = y + i * 4 * C;
        constexpr int BYTES = 4 * C * (int)sizeof(T);          // 8 (bf16, C = 1) ... 64 (fp32, C = 4): a multiple of 8
        static_assert(BYTES % 8 == 0, "stored as 8-byte pieces");
#pragma unroll
        for (int b = 0; b < BYTES / 8; ++b) {
            u32x2 w;
            __builtin_memcpy(&w, (const unsigned char*)o + b * 8, 8);
            *(u32x2*)((unsigned char*)dst + b * 8) = w;
        }
    }
}

// On-device input pipeline (src/mngrdata.py:139-165): RandomCrop(size, padding) + RandomHorizontalFlip + ToTensor
// + Normalize(mean, std) + the NCHW->NHWC hand-off, in one pass over uint8 HWC images.  The random draws stay with
// the caller: per sample a crop origin (oy, ox) in the zero-padded image and a flip flag.
//   y[n][h][w][c] = (u8(x[n][h+oy-pad][wsrc+ox-pad][c]) / 255 - mean[c]) / std[c],  wsrc = flip ? Wo-1-w : w,
//   pixels that fall into the padding are the value 0 before normalisation (torchvision pads the image, fill 0).
template <typename T>
__global__ __launch_bounds__(256) void input_pipeline_kernel(const unsigned char* __restrict__ x, T* __restrict__ y,
                                                             const int* __restrict__ crop_yx, const unsigned char* __restrict__ flip,
                                                             const int* __restrict__ affine_fix,
                                                             const float* __restrict__ mean, const float* __restrict__ stdv,
                                                             int N, int H, int W, int C, int Ho, int Wo, int pad) {
    const int64_t total = (int64_t)N * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t q = i / C;
        const int w = (int)(q % Wo);
        q /= Wo;
        const int h = (int)(q % Ho), n = (int)(q / Ho);
        // RandomAffine, the LAST geometric transform of the reference's list, so the first one seen from the output:
        // PIL's nearest-neighbour affine in 16.16 fixed point (Geometry.c affine_fixed): source pixel of output (w, h)
        // is (xx >> 16, yy >> 16) with xx = a2 + w*a0 + h*a1, yy = a5 + w*a3 + h*a4; outside the image -> fill 0
        int hh = h, ww = w;
        bool inside = true;
        if (affine_fix) {
            const int* a = affine_fix + 6 * n;
            ww = (a[2] + w * a[0] + h * a[1]) >> 16;
            hh = (a[5] + w * a[3] + h * a[4]) >> 16;
            inside = (unsigned)ww < (unsigned)Wo && (unsigned)hh < (unsigned)Ho;
        }
        const int oy = crop_yx ? crop_yx[2 * n] : pad, ox = crop_yx ? crop_yx[2 * n + 1] : pad;
        const int ws = (flip && flip[n]) ? Wo - 1 - ww : ww;
        const int sy = hh + oy - pad, sx = ws + ox - pad;
        float v = 0.f;
        if (inside && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) v = (float)x[(((size_t)n * H + sy) * W + sx) * C + c] * (1.f / 255.f);
        y[i] = from_f32<T>((v - mean[c]) / stdv[c]);
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n4, int64_t n, float lr, float b1, float b2,
                                                   float eps, float wd, float bc1, float bc2_sqrt,
                                                   const float* __restrict__ bias_dev) {
    // step_size = lr / (1 - b1^t); denom = sqrt(v) / sqrt(1 - b2^t) + eps   (torch.optim.Adam)
    // bias_dev: the two bias corrections in device memory (a hipGraph-captured step replays this launch with
    // the values of ITS step, cnk_adam_step_dev)
    if (bias_dev) { bc1 = bias_dev[0]; bc2_sqrt = bias_dev[1]; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t base = i * 4;
        if (base + 4 <= n) {
            f32x4 pp = *(f32x4*)(p + base), gg = *(const f32x4*)(g + base);
            f32x4 mm = *(f32x4*)(m + base), vv = *(f32x4*)(v + base);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float gk = gg[k] + wd * pp[k];
                mm[k] = b1 * mm[k] + (1.f - b1) * gk;
                vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
                float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
                pp[k] -= (lr / bc1) * (mm[k] / denom);
            }
            *(f32x4*)(p + base) = pp;
            *(f32x4*)(m + base) = mm;
            *(f32x4*)(v + base) = vv;
        } else {
            for (int64_t j = base; j < n; ++j) {
                float gk = g[j] + wd * p[j];
                m[j] = b1 * m[j] + (1.f - b1) * gk;
                v[j] = b2 * v[j] + (1.f - b2) * gk * gk;
                float denom = sqrtf(v[j]) / bc2_sqrt + eps;
                p[j] -= (lr / bc1) * (m[j] / denom);
            }
        }
    }
}

// ---- gradient clipping over the flat gradient arena (basemodel.py:131-178) ----------------------
// clip_grad_norm_: total = ||g||_2 over ALL gradients, g *= min(1, max_norm / (total + 1e-6)).
// Pass 1: per-block sums of squares (fixed order).  Pass 2: every block re-derives the same total
// from the partials (<= 1024 floats, fixed order => identical in every block, reproducible), then
// scales its share; block 0 also stores the norm for the caller.
constexpr int CLIP_BLOCKS = 1024;

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void grad_sqsum_kernel(const float* __restrict__ g, int64_t n4, int64_t n,
                                                         float* __restrict__ partial) {
    __shared__ float sh[4];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t base = i * 4;
        if (base + 4 <= n) {
            f32x4 v = *(const f32x4*)(g + base);
            a0 += v[0] * v[0]; a1 += v[1] * v[1]; a2 += v[2] * v[2]; a3 += v[3] * v[3];
        } else {
            for (int64_t j = base; j < n; ++j) a0 += g[j] * g[j];
        }
    }
    float t = block_sum_256((a0 + a1) + (a2 + a3), sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void grad_clip_scale_kernel(float* __restrict__ g, int64_t n4, int64_t n,
                                                              const float* __restrict__ partial, int nparts,
                                                              float max_norm, float* __restrict__ norm_out) {
    __shared__ float sh[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) a += partial[i];
    const float total = sqrtf(block_sum_256(a, sh));
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = total;
    float coef = max_norm / (total + 1e-6f);
    if (!(coef < 1.f)) return;                       // clamp(coef, max=1): nothing to scale
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t base = i * 4;
        if (base + 4 <= n) {
            f32x4 v = *(f32x4*)(g + base);
            v[0] *= coef; v[1] *= coef; v[2] *= coef; v[3] *= coef;
            *(f32x4*)(g + base) = v;
        } else {
            for (int64_t j = base; j < n; ++j) g[j] *= coef;
        }
    }
}

// clip_grad_value_: g = clamp(g, -c, +c)
__global__ __launch_bounds__(256) void grad_clip_value_kernel(float* __restrict__ g, int64_t n4, int64_t n, float c) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t base = i * 4;
        if (base + 4 <= n) {
            f32x4 v = *(f32x4*)(g + base);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fminf(fmaxf(v[k], -c), c);
            *(f32x4*)(g + base) = v;
        } else {
            for (int64_t j = base; j < n; ++j) g[j] = fminf(fmaxf(g[j], -c), c);
        }
    }
}

inline int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096);
}

PoolParams make_pool(int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                     int vec) {
    PoolParams p;
    p.N = N; p.H = H; p.W = W; p.C = C; p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
    p.Ho = (H + 2 * ph - kh) / sh + 1;
    p.Wo = (W + 2 * pw - kw) / sw + 1;
    p.fd_cols = make_fastdiv(C / vec);
    p.fd_wo = make_fastdiv(p.Wo); p.fd_ho = make_fastdiv(p.Ho);
    p.fd_w = make_fastdiv(W); p.fd_h = make_fastdiv(H);
    return p;
}

}  // namespace

template <typename T, int VEC>
static int maxpool_t(bool bwd, const void* a, void* b, void* o, int N, int H, int W, int C,
                     int kh, int kw, int sh, int sw, int ph, int pw, hipStream_t s) {
    PoolParams p = make_pool(N, H, W, C, kh, kw, sh, sw, ph, pw, VEC);
    if (!bwd) {
        int64_t total = (int64_t)N * p.Ho * p.Wo * (C / VEC);
        CNK_CHECK_ARG(total < (1ll << 31), "maxpool: tensor too large");
        hipLaunchKernelGGL((maxpool_fwd_kernel<T, VEC>), dim3(grid_for(total)), dim3(256), 0, s,
                           (const T*)a, (T*)o, (unsigned char*)b, p, total);
    } else {
        int64_t total = (int64_t)N * H * W * (C / VEC);
        CNK_CHECK_ARG(total < (1ll << 31), "maxpool: tensor too large");
        hipLaunchKernelGGL((maxpool_bwd_kernel<T, VEC>), dim3(grid_for(total)), dim3(256), 0, s,
                           (const T*)a, (const unsigned char*)b, (T*)o, p, total);
    }
    CNK_CHECK_LAUNCH("maxpool");
    return 0;
}

template <typename T, int VEC>
static int maxpool_affine_t(const void* x, const float* scale, const float* shift, int relu, void* y, unsigned char* idx,
                            void* xmax, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, hipStream_t s) {
    PoolParams p = make_pool(N, H, W, C, kh, kw, sh, sw, ph, pw, VEC);
    int64_t total = (int64_t)N * p.Ho * p.Wo * (C / VEC);
    CNK_CHECK_ARG(total < (1ll << 31), "maxpool: tensor too large");
    int grid = grid_for(total);
    const int cols = C / VEC;
    if constexpr (VEC > 1) {
        // 3x3 / 2 / pad 1 (the ResNet stems): the streaming kernel; a thread keeps its channel vector when the sweep stride
        // (grid * 256) is a multiple of the vector columns
        if (kh == 3 && kw == 3 && sh == 2 && sw == 2 && ph == 1 && pw == 1 && H >= 2 && W >= 2 && ((int64_t)grid * 256) % cols == 0) {
            hipLaunchKernelGGL((maxpool_k3s2_affine_kernel<T, VEC>), dim3(grid), dim3(256), 0, s, (const T*)x, (T*)y, idx, p,
                               total, scale, shift, relu, (T*)xmax);
            CNK_CHECK_LAUNCH("maxpool_k3s2_affine");
            return 0;
        }
    }
    hipLaunchKernelGGL((maxpool_fwd_kernel<T, VEC, true>), dim3(grid), dim3(256), 0, s, (const T*)x, (T*)y, idx, p,
                       total, scale, shift, relu, (T*)xmax);
    CNK_CHECK_LAUNCH("maxpool_affine");
    return 0;
}

int maxpool_affine_dispatch(const void* x, const float* scale, const float* shift, int relu, void* y, unsigned char* idx,
                            void* xmax, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dtype,
                            hipStream_t s) {
    if (dtype == CNK_BF16) {
        if (C % 8 == 0) return maxpool_affine_t<bf16, 8>(x, scale, shift, relu, y, idx, xmax, N, H, W, C, kh, kw, sh, sw, ph, pw, s);
        return maxpool_affine_t<bf16, 1>(x, scale, shift, relu, y, idx, xmax, N, H, W, C, kh, kw, sh, sw, ph, pw, s);
    }
    if (C % 4 == 0) return maxpool_affine_t<float, 4>(x, scale, shift, relu, y, idx, xmax, N, H, W, C, kh, kw, sh, sw, ph, pw, s);
    return maxpool_affine_t<float, 1>(x, scale, shift, relu, y, idx, xmax, N, H, W, C, kh, kw, sh, sw, ph, pw, s);
}

template <typename T, int VEC>
static int avgpool_t(bool bwd, const void* a, void* o, int N, int H, int W, int C, int kh, int kw, int sh, int sw,
                     int ph, int pw, hipStream_t s) {
    PoolParams p = make_pool(N, H, W, C, kh, kw, sh, sw, ph, pw, VEC);
    int64_t total = (int64_t)N * (bwd ? (int64_t)H * W : (int64_t)p.Ho * p.Wo) * (C / VEC);
    CNK_CHECK_ARG(total < (1ll << 31), "avgpool: tensor too large");
    if (total == 0) return 0;
    if (!bwd) hipLaunchKernelGGL((avgpool_fwd_kernel<T, VEC>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)a, (T*)o, p, total);
    else hipLaunchKernelGGL((avgpool_bwd_kernel<T, VEC>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)a, (T*)o, p, total);
    CNK_CHECK_LAUNCH("avgpool");
    return 0;
}

int avgpool_dispatch(bool bwd, const void* a, void* o, int N, int H, int W, int C, int kh, int kw, int sh, int sw,
                     int ph, int pw, int dtype, hipStream_t s) {
    if (dtype == CNK_BF16) {
        if (C % 8 == 0) return avgpool_t<bf16, 8>(bwd, a, o, N, H, W, C, kh, kw, sh, sw, ph, pw, s);
        return avgpool_t<bf16, 1>(bwd, a, o, N, H, W, C, kh, kw, sh, sw, ph, pw, s);
    }
    if (C % 4 == 0) return avgpool_t<float, 4>(bwd, a, o, N, H, W, C, kh, kw, sh, sw, ph, pw, s);
    return avgpool_t<float, 1>(bwd, a, o, N, H, W, C, kh, kw, sh, sw, ph, pw, s);
}

int maxpool_dispatch(bool bwd, const void* a, void* b, void* o, int N, int H, int W, int C,
                     int kh, int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s) {
    if (dtype == CNK_BF16) {
        if (C % 8 == 0) return maxpool_t<bf16, 8>(bwd, a, b, o, N, H, W, C, kh, kw, sh, sw, ph, pw, s);
        return maxpool_t<bf16, 1>(bwd, a, b, o, N, H, W, C, kh, kw, sh, sw, ph, pw, s);
    }
    if (C % 4 == 0) return maxpool_t<float, 4>(bwd, a, b, o, N, H, W, C, kh, kw, sh, sw, ph, pw, s);
    return maxpool_t<float, 1>(bwd, a, b, o, N, H, W, C, kh, kw, sh, sw, ph, pw, s);
}

template <int OP>
int eltwise_dispatch(const void* a, const void* b, void* o, int64_t n, int dtype, hipStream_t s) {
    if (n == 0) return 0;
    if (dtype == CNK_BF16) {
        if (n % 8 == 0)
            hipLaunchKernelGGL((eltwise_kernel<bf16, 8, OP>), dim3(grid_for(n / 8)), dim3(256), 0, s,
                               (const bf16*)a, (const bf16*)b, (bf16*)o, n / 8);
        else
            hipLaunchKernelGGL((eltwise_kernel<bf16, 1, OP>), dim3(grid_for(n)), dim3(256), 0, s,
                               (const bf16*)a, (const bf16*)b, (bf16*)o, n);
    } else {
        if (n % 4 == 0)
            hipLaunchKernelGGL((eltwise_kernel<float, 4, OP>), dim3(grid_for(n / 4)), dim3(256), 0, s,
                               (const float*)a, (const float*)b, (float*)o, n / 4);
        else
            hipLaunchKernelGGL((eltwise_kernel<float, 1, OP>), dim3(grid_for(n)), dim3(256), 0, s,
                               (const float*)a, (const float*)b, (float*)o, n);
    }
    CNK_CHECK_LAUNCH("eltwise");
    return 0;
}
template int eltwise_dispatch<0>(const void*, const void*, void*, int64_t, int, hipStream_t);
template int eltwise_dispatch<1>(const void*, const void*, void*, int64_t, int, hipStream_t);
template int eltwise_dispatch<2>(const void*, const void*, void*, int64_t, int, hipStream_t);

int gap_dispatch(bool bwd, const void* a, void* o, int N, int HW, int C, int dtype, hipStream_t s) {
    if (!bwd) {
        int total = N * C;
        if (dtype == CNK_BF16)
            hipLaunchKernelGGL(gap_fwd_kernel<bf16>, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const bf16*)a, (float*)o, N, HW, C);
        else
            hipLaunchKernelGGL(gap_fwd_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const float*)a, (float*)o, N, HW, C);
    } else {
        int64_t total = (int64_t)N * HW * C;
        if (dtype == CNK_BF16)
            hipLaunchKernelGGL(gap_bwd_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)a, (bf16*)o, N, HW, C);
        else
            hipLaunchKernelGGL(gap_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)a, (float*)o, N, HW, C);
    }
    CNK_CHECK_LAUNCH("gap");
    return 0;
}

int nchw_to_nhwc_dispatch(const float* x, void* y, int N, int C, int H, int W, int dtype,
                          hipStream_t s) {
    int64_t total = (int64_t)N * C * H * W;
    const int64_t hw = (int64_t)H * W;
    if (C <= 4 && hw % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0) {
        const int64_t quads = (int64_t)N * (hw / 4);
#define CNK_PLANES(TT, CC) hipLaunchKernelGGL((nchw_to_nhwc_planes_kernel<TT, CC>), dim3(grid_for(quads)), dim3(256), 0, s, x, (TT*)y, quads, hw / 4)
        if (dtype == CNK_BF16) {
            if (C == 1) CNK_PLANES(bf16, 1); else if (C == 2) CNK_PLANES(bf16, 2); else if (C == 3) CNK_PLANES(bf16, 3); else CNK_PLANES(bf16, 4);
        } else {
            if (C == 1) CNK_PLANES(float, 1); else if (C == 2) CNK_PLANES(float, 2); else if (C == 3) CNK_PLANES(float, 3); else CNK_PLANES(float, 4);
        }
#undef CNK_PLANES
        CNK_CHECK_LAUNCH("nchw_to_nhwc");
        return 0;
    }
    if (dtype == CNK_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, x, (bf16*)y, N, C, H, W);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, x, (float*)y, N, C, H, W);
    CNK_CHECK_LAUNCH("nchw_to_nhwc");
    return 0;
}

void adam_bias_corrections(float b1, float b2, int step, float* out) {
    out[0] = 1.f - powf(b1, (float)step);
    out[1] = sqrtf(1.f - powf(b2, (float)step));
}

int adam_dispatch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                  float b2, float eps, float wd, int step, const float* bias_dev, hipStream_t s) {
    float bc[2] = {1.f, 1.f};
    if (!bias_dev) adam_bias_corrections(b1, b2, step, bc);
    int64_t n4 = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n4)), dim3(256), 0, s, p, g, m, v, n4, n, lr, b1, b2,
                       eps, wd, bc[0], bc[1], bias_dev);
    CNK_CHECK_LAUNCH("adam");
    return 0;
}

size_t grad_clip_workspace_bytes() { return CLIP_BLOCKS * sizeof(float); }

int grad_clip_norm(float* g, int64_t n, float max_norm, float* norm_out, float* ws, hipStream_t s) {
    int64_t n4 = (n + 3) / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > CLIP_BLOCKS) blocks = CLIP_BLOCKS;
    hipLaunchKernelGGL(grad_sqsum_kernel, dim3(blocks), dim3(256), 0, s, g, n4, n, ws);
    CNK_CHECK_LAUNCH("grad_sqsum");
    hipLaunchKernelGGL(grad_clip_scale_kernel, dim3(blocks), dim3(256), 0, s, g, n4, n, ws, blocks, max_norm, norm_out);
    CNK_CHECK_LAUNCH("grad_clip_scale");
    return 0;
}

int grad_clip_value(float* g, int64_t n, float c, hipStream_t s) {
    int64_t n4 = (n + 3) / 4;
    hipLaunchKernelGGL(grad_clip_value_kernel, dim3(grid_for(n4)), dim3(256), 0, s, g, n4, n, c);
    CNK_CHECK_LAUNCH("grad_clip_value");
    return 0;
}

// torchvision.transforms.functional._get_inverse_affine_matrix (the matrix RandomAffine hands to PIL's
// Image.transform(AFFINE); torchvision is a dependency of the reference, mngrdata.py:148, not vendored in it) followed
// by PIL's conversion to 16.16 fixed point (Geometry.c: FIX(v) = floor(v * 65536 + 0.5), half-pixel offset folded
// into the constant terms).  Centre of rotation = (W/2, H/2), angles in degrees.
void affine_fixed_coeffs(double angle, double tx, double ty, double scale, double shear_x, double shear_y, int W, int H,
                         int* out6) {
    const double rad = 3.14159265358979323846 / 180.0;
    const double rot = angle * rad, sx = shear_x * rad, sy = shear_y * rad;
    const double cx = W * 0.5, cy = H * 0.5;
    const double a = cos(rot - sy) / cos(sy);
    const double b = -cos(rot - sy) * tan(sx) / cos(sy) - sin(rot);
    const double c = sin(rot - sy) / cos(sy);
    const double d = -sin(rot - sy) * tan(sx) / cos(sy) + cos(rot);
    double m[6] = {d / scale, -b / scale, 0.0, -c / scale, a / scale, 0.0};
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty);
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty);
    m[2] += cx;
    m[5] += cy;
    auto fix = [](double v) { return (int)floor(v * 65536.0 + 0.5); };
    out6[0] = fix(m[0]); out6[1] = fix(m[1]); out6[3] = fix(m[3]); out6[4] = fix(m[4]);
    out6[2] = fix(m[2] + m[0] * 0.5 + m[1] * 0.5);
    out6[5] = fix(m[5] + m[3] * 0.5 + m[4] * 0.5);
}

int input_pipeline(const unsigned char* x, void* y, const int* crop_yx, const unsigned char* flip, const int* affine_fix,
                   const float* mean, const float* stdv, int N, int H, int W, int C, int Ho, int Wo, int pad, int dtype,
                   hipStream_t s) {
    const int64_t total = (int64_t)N * Ho * Wo * C;
    if (total == 0) return 0;
    if (dtype == CNK_BF16)
        hipLaunchKernelGGL(input_pipeline_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, x, (bf16*)y, crop_yx, flip, affine_fix, mean, stdv, N, H, W, C, Ho, Wo, pad);
    else
        hipLaunchKernelGGL(input_pipeline_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, x, (float*)y, crop_yx, flip, affine_fix, mean, stdv, N, H, W, C, Ho, Wo, pad);
    CNK_CHECK_LAUNCH("input_pipeline");
    return 0;
}
