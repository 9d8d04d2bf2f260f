// Weight-gradient convolution on MFMA for gfx950.
//
//   dw[co][t][ci] = sum_m dy[m][co] * xg[m,t][ci],   m = (img, ho, wo),  xg = x at (ho*s+kh-p, wo*s+kw-p)
//
// GEMM view: rows = co, cols = ci (one tap per workgroup), reduction over pixels m.  Both operands
// are stored pixel-major in HBM (NHWC), i.e. with the REDUCTION index strided, so the LDS images
// are [pixel][channel] and the MFMA fragments are fetched with the gfx950 transposing LDS read
// (ds_read_b64_tr_b16) for bf16; for fp32 the 32x32x2 MFMA takes one value per lane, which the
// [pixel][channel] image serves with conflict-free ds_read_b32.  The pixel range is split over
// workgroups (split-K); each writes an fp32 partial slab [split][co][t][ci] with plain stores and a
// second kernel sums the slabs in fixed order (deterministic, unlike float atomics), applies the
// .grad accumulation and re-lays into the master layout [Cout][Cin][kh][kw].
#include <cstdlib>

#include "common.h"
#include "wgrad_chain.h"

namespace {

struct WgradParams {
    const void* dy;
    const void* x;
    float* slab;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, sh, sw, ph, pw, dil;
    int M;
    int ksteps, ksteps_per_split, splits;
    int tiles_co, tiles_ci, ntaps;
    int diag;               // grouped conv: only the diagonal 64x64 (co chunk, same ci chunk) tiles exist
    FastDiv fd_howo, fd_wo;
    // FOLD kernels (cnk_conv2d_wgrad_bnfold): `dy` is the gradient w.r.t. the OUTPUT of the BN layer behind this conv (already
    // ReLU-masked) and the gradient w.r.t. the conv output is formed while the tile is staged, exactly as bn_bwd_apply_kernel
    // would have written it: dy_conv = A * (dy - B1 - (y - mean) * (invstd * B2)), rounded to bf16.
    const void* fy;         // the conv output / BN input [M][Cout]
    const float* fcoef;     // [3][Cout]: A = gamma * invstd, B1 = mean(dy), B2 = mean(dy * xhat)  (cnk_bn_bwd_coef)
    const float* fmean;
    const float* finvstd;
    WgradPending prev;      // the slab sum of the PREVIOUS weight gradient of this stream, done in this launch's prologue (wgrad_chain.h)
};

template <typename T> struct WgCfg;
template <> struct WgCfg<bf16> {
    static constexpr int KP = 64;      // pixels per K step
    static constexpr int PAD = 64;     // row padding (bytes): stride == 64 mod 256 -> tr reads conflict-free
};
template <> struct WgCfg<float> {
    static constexpr int KP = 32;
    static constexpr int PAD = 0;
};

// WM x WN waves (2 x 2: 256 threads, two workgroups per CU; 4 x 2 / 2 x 4: 512 threads, ONE 256x128 / 128x256 tile per
// CU -- the kernel is bound by the global -> LDS path (rocprofv3 MFMA-busy 21 % with 64 B/clk/CU demanded at full MFMA
// rate for 128x128 against ~28 B/clk delivered), and a 256x128 tile moves 25 % fewer bytes per flop).
template <typename T, int BMW, int BNW, int NBUF, int WM = 2, int WN = 2, bool FOLD = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 && NBUF != 2) ? 3 : 2) void wgrad_mfma_kernel(const WgradParams p) {
    constexpr int NT = 64 * WM * WN;
    constexpr int PER16 = ElemTraits<T>::PER16;
    constexpr int KP = WgCfg<T>::KP;
    constexpr int ROWA = BMW * (int)sizeof(T) + WgCfg<T>::PAD;
    constexpr int ROWB = BNW * (int)sizeof(T) + WgCfg<T>::PAD;
    constexpr int CPRA = BMW / PER16, CPRB = BNW / PER16;       // 16-B chunks per row
    constexpr int PASSA = KP * CPRA / NT, PASSB = KP * CPRB / NT;
    constexpr int TM = BMW / (32 * WM), TN = BNW / (32 * WN);   // 32x32 blocks per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                      // [NBUF][KP][ROWA]
    unsigned char* Bs = smem + NBUF * KP * ROWA;   // [NBUF][KP][ROWB]
    wgrad_pending_sum<NT>(p.prev);                 // (nothing pending: one uniform branch)

    // XCD-aware order: blocks b, b+8, ... share an XCD (and its L2).  Give each XCD a contiguous run
    // of logical ids so the workgroups of ONE pixel split -- which re-read the same dy / x rows for
    // every (co tile, tap, ci tile) -- run on one XCD close in time and hit its L2.
    int bid = blockIdx.x;
    {
        int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tiles_nc = p.ntaps * p.tiles_ci;
    const int per_split = p.tiles_co * tiles_nc;
    const int split = bid / per_split;
    bid -= split * per_split;
    const int tile_co = bid / tiles_nc;
    const int rest = bid - tile_co * tiles_nc;
    const int tap = rest / p.tiles_ci;
    // grouped conv: the x columns are those of the output chunk; the slab keeps 64 local ci per co
    const int tile_ci = p.diag ? tile_co : rest - tap * p.tiles_ci;
    const int slab_ci0 = p.diag ? 0 : tile_ci * BNW, slab_row = p.diag ? BNW : p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const T* dy = (const T*)p.dy;
    const T* x = (const T*)p.x;
    const int ks0 = split * p.ksteps_per_split;
    const int ks1 = min(p.ksteps, ks0 + p.ksteps_per_split);

    u32x4 ra[PASSA], rb[PASSB];
    // FOLD: this thread always stages the same 16-byte channel group of the dy tile (NT is a multiple of CPRA): its BN-backward
    // coefficients stay in registers
    static_assert(!FOLD || (NT % CPRA == 0 && sizeof(T) == 2), "FOLD: bf16, fixed channel group per thread");
    u32x4 ry[FOLD ? PASSA : 1];
    unsigned fok = 0;
    float ka[FOLD ? PER16 : 1], kb1[FOLD ? PER16 : 1], kmu[FOLD ? PER16 : 1], kis[FOLD ? PER16 : 1];
    if constexpr (FOLD) {
        const int c0 = tile_co * BMW + (tid % CPRA) * PER16;
        if (c0 < p.Cout) {
#pragma unroll
            for (int e = 0; e < PER16; ++e) {
                ka[e] = p.fcoef[c0 + e];
                kb1[e] = p.fcoef[p.Cout + c0 + e];
                kmu[e] = p.fmean[c0 + e];
                kis[e] = p.finvstd[c0 + e];
            }
#pragma unroll
            for (int e = 0; e < PER16; ++e) kis[e] *= p.fcoef[2 * p.Cout + c0 + e];
        } else {
#pragma unroll
            for (int e = 0; e < PER16; ++e) ka[e] = kb1[e] = kmu[e] = kis[e] = 0.f;
        }
    }
    auto load_step = [&](int ks) {
        const int m0 = ks * KP;
        if constexpr (FOLD) fok = 0;
#pragma unroll
        for (int i = 0; i < PASSA; ++i) {
            int idx = tid + NT * i, r = idx / CPRA, c = idx - r * CPRA;
            int m = m0 + r;
            u32x4 v = {0, 0, 0, 0};
            // (channel bound: Cout / Cin may be any multiple of one vector; columns past them are zeros)
            if constexpr (FOLD) {
                // unconditional loads from a clamped address, the zero selected when the tile is staged: with the two loads and
                // the flag update inside the bounds branch hipcc waited vmcnt(0) in front of every pair -- four dependent round
                // trips per K step (cdna_hip_programming.md 5, trap (c)), which doubled this kernel's time
                const bool ok = m < p.M && tile_co * BMW + c * PER16 < p.Cout;
                const size_t off = ok ? (size_t)m * p.Cout + tile_co * BMW + c * PER16 : 0;
                v = *(const u32x4*)(dy + off);
                ry[i] = *(const u32x4*)((const T*)p.fy + off);
                fok |= (ok ? 1u : 0u) << i;
            } else if (m < p.M && tile_co * BMW + c * PER16 < p.Cout) {
                v = *(const u32x4*)(dy + (size_t)m * p.Cout + tile_co * BMW + c * PER16);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < PASSB; ++i) {
            int idx = tid + NT * i, r = idx / CPRB, c = idx - r * CPRB;
            int m = m0 + r;
            u32x4 v = {0, 0, 0, 0};
            if (m < p.M && tile_ci * BNW + c * PER16 < p.Cin) {
                uint32_t img = fd_div(m, p.fd_howo);
                uint32_t rem = m - img * p.fd_howo.d;
                uint32_t ho = fd_div(rem, p.fd_wo);
                uint32_t wo = rem - ho * p.fd_wo.d;
                int ih = (int)ho * p.sh - p.ph + kh * p.dil, iw = (int)wo * p.sw - p.pw + kw * p.dil;
                if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                    v = *(const u32x4*)(x + ((size_t)(img * p.H + ih) * p.W + iw) * p.Cin +
                                        tile_ci * BNW + c * PER16);
            }
            rb[i] = v;
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PASSA; ++i) {
            int idx = tid + NT * i, r = idx / CPRA, c = idx - r * CPRA;
            if constexpr (FOLD) {
                // same expression, same rounding as bn_bwd_row (bn.hip): the staged tile IS the tensor that kernel would write
                const bf16x8 dv = __builtin_bit_cast(bf16x8, ra[i]), yv = __builtin_bit_cast(bf16x8, ry[i]);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < PER16; ++e)
                    o[e] = (bf16)(ka[e] * ((float)dv[e] - kb1[e] - ((float)yv[e] - kmu[e]) * kis[e]));
                const u32x4 ov = __builtin_bit_cast(u32x4, o);
                const bool ok = (fok >> i) & 1u;
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[i][e] = ok ? ov[e] : 0u;
            }
            *(u32x4*)(As + buf * (KP * ROWA) + r * ROWA + c * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < PASSB; ++i) {
            int idx = tid + NT * i, r = idx / CPRB, c = idx - r * CPRB;
            *(u32x4*)(Bs + buf * (KP * ROWB) + r * ROWB + c * 16) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (ks0 < ks1) {
        load_step(ks0);
        store_step(0);
    }
    __syncthreads();
    for (int ks = ks0; ks < ks1; ++ks) {
        const int buf = NBUF == 2 ? ((ks - ks0) & 1) : 0;
        if (ks + 1 < ks1) load_step(ks + 1);
        const unsigned char* Ab = As + buf * (KP * ROWA);
        const unsigned char* Bb = Bs + buf * (KP * ROWB);
        if constexpr (sizeof(T) == 2) {
            // lane l: group g = l>>4 (h = g>>1 selects pixels 8h..8h+7, g&1 the 16-channel half),
            // q = (l&15)>>2 the pixel row of the 4x16 block whose address this lane supplies,
            // pp = l&3 the 4-channel column group.  Two reads give k = 8h + {0..3}, {4..7}.
            const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, h = g >> 1;
#pragma unroll
            for (int ss = 0; ss < KP / 16; ++ss) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    int col = wm * (BMW / WM) + i * 32 + 16 * (g & 1) + 4 * pp;
                    const unsigned char* a0 = Ab + (ss * 16 + 8 * h + q) * ROWA + col * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(a0));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(a0 + 4 * ROWA));
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    s16x8 w8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    fa[i] = __builtin_bit_cast(bf16x8, w8);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    int col = wn * (BNW / WN) + j * 32 + 16 * (g & 1) + 4 * pp;
                    const unsigned char* b0 = Bb + (ss * 16 + 8 * h + q) * ROWB + col * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(b0));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(b0 + 4 * ROWB));
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    s16x8 w8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    fb[j] = __builtin_bit_cast(bf16x8, w8);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j],
                                                                            0, 0, 0);
            }
        } else {
            const int l31 = lane & 31, lk = lane >> 5;
#pragma unroll 4
            for (int ss = 0; ss < KP / 2; ++ss) {
                float fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = *(const float*)(Ab + (ss * 2 + lk) * ROWA +
                                            (wm * (BMW / WM) + i * 32 + l31) * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = *(const float*)(Bb + (ss * 2 + lk) * ROWB +
                                            (wn * (BNW / WN) + j * 32 + l31) * 4);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0,
                                                                         0, 0);
            }
        }
        if (NBUF == 1) __syncthreads();          // all waves finished reading the single buffer
        if (ks + 1 < ks1) store_step(NBUF == 2 ? (buf ^ 1) : 0);
        __syncthreads();
    }

    const int l31 = lane & 31, lh = lane >> 5;
    float* slab = p.slab + (size_t)split * p.Cout * p.ntaps * slab_row;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = tile_co * BMW + wm * (BMW / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                int ci = slab_ci0 + wn * (BNW / WN) + j * 32 + l31;
                if (co < p.Cout && ci < slab_row) slab[((size_t)co * p.ntaps + tap) * slab_row + ci] = acc[i][j][r];
            }
}

// dw[co][ci][t] (+)= sum_s slab[s*slab_stride + co*s_co + t*s_t + ci*s_ci]
// Four lanes per output element, lane q summing slabs q, q+4, ... (independent loads in flight),
// combined with two xor-shuffles: fixed summation order => bitwise reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(
    const float* __restrict__ slab, float* __restrict__ dw, int splits, size_t slab_stride, int Cout,
    int Cin, int ntaps, size_t s_co, size_t s_t, size_t s_ci, int inner_is_ci, int accumulate) {
    size_t total = (size_t)Cout * Cin * ntaps;
    const int q = threadIdx.x & 3;
    const size_t total_r = (total + 63) / 64 * 64;        // whole waves stay converged for the shuffles
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; i < total_r;
         i += ((size_t)gridDim.x * blockDim.x) >> 2) {
        const bool live = i < total;
        size_t ii = live ? i : 0;
        int co, t, ci;
        if (inner_is_ci) {            // slab order [co][t][ci]
            ci = (int)(ii % Cin);
            size_t r = ii / Cin;
            t = (int)(r % ntaps);
            co = (int)(r / ntaps);
        } else {                      // slab order [t][ci][co]
            co = (int)(ii % Cout);
            size_t r = ii / Cout;
            ci = (int)(r % Cin);
            t = (int)(r / Cin);
        }
        const float* sp = slab + co * s_co + t * s_t + ci * s_ci;
        float a0 = 0.f, a1 = 0.f;
        int s = q;
        for (; s + 4 < splits; s += 8) {
            a0 += sp[(size_t)s * slab_stride];
            a1 += sp[(size_t)(s + 4) * slab_stride];
        }
        if (s < splits) a0 += sp[(size_t)s * slab_stride];
        float acc = a0 + a1;
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (q == 0 && live) {
            size_t o = ((size_t)co * Cin + ci) * ntaps + t;
            dw[o] = accumulate ? dw[o] + acc : acc;
        }
    }
}

// The same sum for the slab order [co][t][ci] with Cin % 4 == 0 (every MFMA weight-gradient kernel): 16-byte loads of FOUR
// consecutive input channels (a wave instruction reads whole 16-byte vectors instead of single floats of four slabs), four
// slabs in flight per lane, fixed summation order -- bitwise reproducible.
// G lanes share one 4-channel group when the gradient is small and the slabs many (64 x 64 x 9 weights from 100+ slabs: one
// thread per group would be 36 workgroups walking 100 dependent steps): lane q sums slabs q, q+G, ... and the G partial sums
// are combined by a fixed xor-shuffle tree.
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits,
                                                            size_t slab_stride, int Cout, int Cin, int ntaps, int accumulate) {
    const int c4 = Cin >> 2;
    const size_t total = (size_t)Cout * ntaps * c4;
    const size_t total_r = (total + 63) / 64 * 64;                        // whole waves stay converged for the shuffles
    const int q = threadIdx.x % G;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / G; i < total_r; i += ((size_t)gridDim.x * blockDim.x) / G) {
        const bool live = i < total;
        const size_t ii = live ? i : 0;
        const int cq = (int)(ii % c4);
        const size_t r = ii / c4;
        const int t = (int)(r % ntaps), co = (int)(r / ntaps);
        const float* sp = slab + ii * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        int s = q;
        for (; s + 3 * G < splits; s += 4 * G) {
            const f32x4 v0 = *(const f32x4*)(sp + (size_t)s * slab_stride);
            const f32x4 v1 = *(const f32x4*)(sp + (size_t)(s + G) * slab_stride);
            const f32x4 v2 = *(const f32x4*)(sp + (size_t)(s + 2 * G) * slab_stride);
            const f32x4 v3 = *(const f32x4*)(sp + (size_t)(s + 3 * G) * slab_stride);
            a = a + v0;
            a = a + v1;
            a = a + v2;
            a = a + v3;
        }
        for (; s < splits; s += G) a = a + *(const f32x4*)(sp + (size_t)s * slab_stride);
#pragma unroll
        for (int m = 1; m < G; m <<= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += __shfl_xor(a[j], m, 64);
        if (q == 0 && live) {
            float* o = dw + ((size_t)co * Cin + cq * 4) * ntaps + t;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[(size_t)j * ntaps] = accumulate ? o[(size_t)j * ntaps] + a[j] : a[j];
        }
    }
}

template <typename T, int BMW, int BNW, int WM = 2, int WN = 2, bool FOLD = false> int launch_wgrad(const WgradParams& p, hipStream_t s) {
    constexpr int KP = WgCfg<T>::KP;
    size_t lds = 2 * (KP * (BMW * sizeof(T) + WgCfg<T>::PAD) + KP * (BNW * sizeof(T) + WgCfg<T>::PAD));   // double-buffered
    int grid = p.splits * p.tiles_co * p.ntaps * p.tiles_ci;
    // algorithmic bytes: dy and x read once, dw written once (the split-K slabs are implementation traffic: the PMC
    // passes show them, the roofline does not credit them -- ADVICE r02)
    ProfScope kscope(CNK_KF_WGRAD, s, 2.0 * p.M * p.Cout * (double)(p.diag ? BNW : p.Cin) * p.ntaps,
                     ((double)p.M * p.Cout * (FOLD ? 2 : 1) + (double)p.N * p.H * p.W * p.Cin) * sizeof(T) +
                         (double)p.Cout * p.ntaps * (p.diag ? BNW : p.Cin) * sizeof(float));
    if (lds > 64 * 1024) CNK_LDS_OPTIN((wgrad_mfma_kernel<T, BMW, BNW, 2, WM, WN, FOLD>), 160 * 1024, "wgrad_mfma");
    hipLaunchKernelGGL((wgrad_mfma_kernel<T, BMW, BNW, 2, WM, WN, FOLD>), dim3(grid), dim3(64 * WM * WN), lds, s, p);
    CNK_CHECK_LAUNCH("wgrad_mfma");
    return 0;
}

}  // namespace

int launch_wgrad_reduce(const float* slab, float* dw, int splits, size_t slab_stride, int Cout,
                        int Cin, int ntaps, size_t s_co, size_t s_t, size_t s_ci, int inner_is_ci,
                        int accumulate, hipStream_t s) {
    size_t total = (size_t)Cout * Cin * ntaps;
    if (inner_is_ci && Cin % 4 == 0 && s_ci == 1 && s_t == (size_t)Cin && s_co == (size_t)ntaps * Cin && slab_stride % 4 == 0) {
        // lanes per 4-channel group: enough threads to fill the chip (>= ~256 k), never more lanes than slabs
        const size_t groups = total / 4;
        int G = 1;
        while (G < 16 && groups * G < 262144 && 2 * G <= splits) G <<= 1;
        int grid4 = (int)((groups * G + 255) / 256);
        if (grid4 > 4096) grid4 = 4096;
        if (G == 1) hipLaunchKernelGGL(wgrad_reduce4_kernel<1>, dim3(grid4), dim3(256), 0, s, slab, dw, splits, slab_stride, Cout, Cin, ntaps, accumulate);
        else if (G == 2) hipLaunchKernelGGL(wgrad_reduce4_kernel<2>, dim3(grid4), dim3(256), 0, s, slab, dw, splits, slab_stride, Cout, Cin, ntaps, accumulate);
        else if (G == 4) hipLaunchKernelGGL(wgrad_reduce4_kernel<4>, dim3(grid4), dim3(256), 0, s, slab, dw, splits, slab_stride, Cout, Cin, ntaps, accumulate);
        else if (G == 8) hipLaunchKernelGGL(wgrad_reduce4_kernel<8>, dim3(grid4), dim3(256), 0, s, slab, dw, splits, slab_stride, Cout, Cin, ntaps, accumulate);
        else hipLaunchKernelGGL(wgrad_reduce4_kernel<16>, dim3(grid4), dim3(256), 0, s, slab, dw, splits, slab_stride, Cout, Cin, ntaps, accumulate);
        CNK_CHECK_LAUNCH("wgrad_reduce4");
        return 0;
    }
    int grid = (int)((total * 4 + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, s, slab, dw, splits,
                       slab_stride, Cout, Cin, ntaps, s_co, s_t, s_ci, inner_is_ci, accumulate);
    CNK_CHECK_LAUNCH("wgrad_reduce");
    return 0;
}

// the last sum of a chain (nothing launches behind it): the same device function on a launch of its own
namespace {
__global__ __launch_bounds__(256) void wgrad_pending_kernel(const WgradPending q) { wgrad_pending_sum<256>(q); }
}  // namespace
int wgrad_pending_reduce(const WgradPending& q, hipStream_t s) {
    if (!q.slab) return 0;
    const size_t lanes = (size_t)q.Cout * q.ntaps * (q.Cin / 4) * q.G;
    int grid = (int)((lanes + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(wgrad_pending_kernel, dim3(grid), dim3(256), 0, s, q);
    CNK_CHECK_LAUNCH("wgrad_pending");
    return 0;
}

bool igemm_grouped_ok(const cnk_conv_desc* d);
bool wgrad_mfma_supported(const cnk_conv_desc* d) {
    if (conv_groups(d) > 1 && !igemm_grouped_ok(d)) return false;
    if (d->Cin % 64 == 0 && d->Cout % 64 == 0) return true;
    // off the tile grid: any multiple of one 16-byte vector (zero-filled columns), not for the tiny layers
    const int vec = d->dtype == CNK_BF16 ? 8 : 4;
    return conv_groups(d) == 1 && d->Cin % vec == 0 && d->Cout % vec == 0 && d->Cin >= 16 && d->Cout >= 16;
}

// Tile shape and split count shared by the workspace query and the launch.
// grouped conv: dw[co][l][t] (+)= sum_s slab[s][co][t][local ci], local ci = ((co % 64) / gw) * gw + l
__global__ __launch_bounds__(256) void wgrad_reduce_grouped_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                   int splits, int Cout, int gw, int ntaps, int accumulate) {
    const size_t total = (size_t)Cout * gw * ntaps, per = (size_t)Cout * ntaps * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % ntaps);
        const size_t r = i / ntaps;
        const int l = (int)(r % gw), co = (int)(r / gw);
        const float* sp = slab + ((size_t)co * ntaps + t) * 64 + ((co & 63) / gw) * gw + l;
        float a0 = 0.f, a1 = 0.f;
        int s = 0;
        for (; s + 1 < splits; s += 2) {
            a0 += sp[(size_t)s * per];
            a1 += sp[(size_t)(s + 1) * per];
        }
        if (s < splits) a0 += sp[(size_t)s * per];
        dw[i] = accumulate ? dw[i] + (a0 + a1) : (a0 + a1);
    }
}

// Share of the chip's workgroup slots the split-K of a weight gradient fills, as the CALLER bounds it: cnk_conv2d_wgrad takes
// any workspace of at least one slab, and a workspace smaller than cnk_conv2d_workspace_bytes() makes the kernel split the pixel
// range less deep (max_splits = workspace / slab).  Why a caller would: bf16 weight gradients run on a second HIP stream BESIDE
// the data gradients and BN backward passes of the training step (convnets_amd/ops.py); they only have to be done when the
// backward pass ends, and every CU that holds one of their 112-KiB-LDS workgroups admits a single implicit-GEMM workgroup of the
// main stream instead of four.  Measured (round 3, A/B in one call, hipGraph replays): ResNet-50 bs 256 with all slots
// 20.54-20.61 ms, 75 % 20.35-20.41, 50 % 20.21-20.22, 35 % 21.3 (the weight gradients become the tail); ResNet-152 bs 128
// 26.40 -> 25.80 ms at 50 %; but VGG-16 (every layer a heavy 3x3: the weight gradients ARE the critical path) 10.25 -> 11.3 ms
// and ResNeXt-50 16.99 -> 18.24: the right share is a property of the MODEL, so the host side measures it once
// (ops.calibrate_wgrad_share).  Fewer splits also mean proportionally less slab traffic and a shorter slab sum.
// CNK_WGRAD_SLOTS_PCT forces a share inside the library (A/B runs).
static int wgrad_cap_splits(int splits, int ksteps, size_t slab_bytes, size_t ws_bytes, int* per_split) {
    const size_t fit = ws_bytes / slab_bytes;
    if (fit >= (size_t)splits || fit == 0) return splits;       // enough room (or none: the caller's size check reports it)
    *per_split = ceil_div(ksteps, (int)fit);
    return ceil_div(ksteps, *per_split);
}

static void wgrad_plan(const cnk_conv_desc* d, int* bmw, int* bnw, int* splits, int* ksteps,
                       int* per_split) {
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    int M = d->N * Ho * Wo;
    int kp = d->dtype == CNK_BF16 ? 64 : 32;
    const bool grouped = conv_groups(d) > 1;
    *bmw = (!grouped && d->Cout % 128 == 0) ? 128 : 64;
    *bnw = (!grouped && d->Cin % 128 == 0) ? 128 : 64;
    // bf16, both channel counts on the 128 grid and one of them on the 256 grid: one 256x128 / 128x256 tile per CU
    // (CNK_WGRAD_BIG=0: the 128x128 tiles, for in-process A/B runs)
    int slots = 512;                                         // resident workgroups: 2 per CU (80 KiB LDS each) ...
    {
        const bool big = cnk_sw().wgrad_big && d->dtype == CNK_BF16 && *bmw == 128 && *bnw == 128;
        if (big && d->Cout % 256 == 0) { *bmw = 256; slots = 256; }            // ... or 1 (112 KiB)
        else if (big && d->Cin % 256 == 0) { *bnw = 256; slots = 256; }
    }
    int tiles = grouped ? (d->Cout / 64) * d->kh * d->kw
                        : ceil_div(d->Cout, *bmw) * ceil_div(d->Cin, *bnw) * d->kh * d->kw;
    if (cnk_sw().wgrad_slots_pct > 0) slots = slots * cnk_sw().wgrad_slots_pct / 100;
    if (slots < 1) slots = 1;
    *ksteps = ceil_div(M, kp);
    // The kernel is resident at 2 workgroups per CU (80 KiB LDS each): 512 slots on 256 CUs.  Fill
    // them exactly once -- tiles*splits just ABOVE a multiple of 512 costs a whole extra round.
    int want = tiles >= slots ? 1 : slots / tiles;
    int max_by_k = *ksteps / 4 > 0 ? *ksteps / 4 : 1;       // at least 4 K steps per split
    int sp = want < max_by_k ? want : max_by_k;
    if (sp < 1) sp = 1;
    *per_split = ceil_div(*ksteps, sp);
    *splits = ceil_div(*ksteps, *per_split);
}

size_t wgrad_mfma_workspace(const cnk_conv_desc* d) {
    int bmw, bnw, splits, ksteps, per;
    wgrad_plan(d, &bmw, &bnw, &splits, &ksteps, &per);
    return (size_t)splits * d->Cout * (conv_groups(d) > 1 ? 64 : d->Cin) * d->kh * d->kw * sizeof(float);
}

int wgrad_mfma(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate,
               void* ws, size_t ws_bytes, hipStream_t s, const WgradFold* fold, const WgradPending* prev, WgradPending* out) {
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    WgradParams p = {};
    if (prev) p.prev = *prev;
    CNK_CHECK_ARG(!out || conv_groups(d) == 1, "wgrad_mfma: grouped convs sum their slabs themselves (no chained sum)");
    int bmw, bnw;
    wgrad_plan(d, &bmw, &bnw, &p.splits, &p.ksteps, &p.ksteps_per_split);
    const bool grouped = conv_groups(d) > 1;
    const size_t slab_bytes = (size_t)d->Cout * (grouped ? 64 : d->Cin) * d->kh * d->kw * sizeof(float);
    p.splits = wgrad_cap_splits(p.splits, p.ksteps, slab_bytes, ws ? ws_bytes : 0, &p.ksteps_per_split);
    size_t need = (size_t)p.splits * slab_bytes;
    CNK_CHECK_ARG(ws && ws_bytes >= need, "conv2d_wgrad: workspace too small (%zu < %zu; at least one slab of %zu bytes)", ws_bytes, need, slab_bytes);
    p.dy = dy; p.x = x; p.slab = (float*)ws;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = d->Cout;
    p.KH = d->kh; p.KW = d->kw; p.sh = d->stride_h; p.sw = d->stride_w; p.ph = d->pad_h; p.pw = d->pad_w;
    p.dil = conv_dil(d);
    p.M = d->N * Ho * Wo;
    p.tiles_co = ceil_div(d->Cout, bmw); p.tiles_ci = grouped ? 1 : ceil_div(d->Cin, bnw); p.ntaps = d->kh * d->kw;
    p.diag = grouped;
    p.fd_howo = make_fastdiv(Ho * Wo);
    p.fd_wo = make_fastdiv(Wo);
    int rc;
    bool b = d->dtype == CNK_BF16;
    if (fold) {
        CNK_CHECK_ARG(b && !grouped && d->Cout % 64 == 0 && d->Cin % 64 == 0, "wgrad_mfma: the folded BN backward needs a dense bf16 conv on the 64-channel grid");
        p.fy = fold->y; p.fcoef = fold->coef; p.fmean = fold->mean; p.finvstd = fold->invstd;
        // (Round 5, measured and dropped: ONE staging buffer for the 512-thread tiles -- 56 instead of 112 KiB of LDS, so that a CU
        // holding such a workgroup could admit two implicit-GEMM workgroups of the compute stream -- 19.10 / 19.26 ms against
        // 19.36 / 19.26: nothing.  It is the registers that decide who shares a CU: two waves per SIMD of this kernel hold 2 x 168
        // (FOLD: 2 x 208) of the 512 VGPRs, which leaves room for ONE 120-register implicit-GEMM wave per SIMD (FOLD: none).)
        if (bmw == 256) rc = launch_wgrad<bf16, 256, 128, 4, 2, true>(p, s);
        else if (bnw == 256) rc = launch_wgrad<bf16, 128, 256, 2, 4, true>(p, s);
        else if (bmw == 128 && bnw == 128) rc = launch_wgrad<bf16, 128, 128, 2, 2, true>(p, s);
        else if (bmw == 128) rc = launch_wgrad<bf16, 128, 64, 2, 2, true>(p, s);
        else if (bnw == 128) rc = launch_wgrad<bf16, 64, 128, 2, 2, true>(p, s);
        else rc = launch_wgrad<bf16, 64, 64, 2, 2, true>(p, s);
    } else
    if (bmw == 256) rc = launch_wgrad<bf16, 256, 128, 4, 2>(p, s);
    else if (bnw == 256) rc = launch_wgrad<bf16, 128, 256, 2, 4>(p, s);
    else if (bmw == 128 && bnw == 128) rc = b ? launch_wgrad<bf16, 128, 128>(p, s) : launch_wgrad<float, 128, 128>(p, s);
    else if (bmw == 128) rc = b ? launch_wgrad<bf16, 128, 64>(p, s) : launch_wgrad<float, 128, 64>(p, s);
    else if (bnw == 128) rc = b ? launch_wgrad<bf16, 64, 128>(p, s) : launch_wgrad<float, 64, 128>(p, s);
    else rc = b ? launch_wgrad<bf16, 64, 64>(p, s) : launch_wgrad<float, 64, 64>(p, s);
    if (rc) return rc;
    if (grouped) {
        const int gw = d->Cin / conv_groups(d);
        size_t total = (size_t)d->Cout * gw * p.ntaps;
        int grid = (int)((total + 255) / 256);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(wgrad_reduce_grouped_kernel, dim3(grid), dim3(256), 0, s, p.slab, dw, p.splits, d->Cout, gw,
                           p.ntaps, accumulate);
        CNK_CHECK_LAUNCH("wgrad_reduce_grouped");
        return 0;
    }
    size_t per = (size_t)d->Cout * p.ntaps * d->Cin;
    if (out) {                                   // chained: the sum is left to the next launch of the stream (wgrad_chain.h)
        CNK_CHECK_ARG(d->Cin % 4 == 0, "wgrad_mfma: a chained slab sum needs Cin % 4 == 0");
        wgrad_pending_fill(*out, p.slab, dw, p.splits, per, d->Cout, d->Cin, p.ntaps, accumulate);
        return 0;
    }
    return launch_wgrad_reduce(p.slab, dw, p.splits, per, d->Cout, d->Cin, p.ntaps,
                               (size_t)p.ntaps * d->Cin, (size_t)d->Cin, 1, 1, accumulate, s);
}
