// Implicit-GEMM "gather" convolution on MFMA for gfx950: forward and data-gradient.
//
//   C[m][n] = sum_{t,c} A[m][(t,c)] * B[n][(t,c)]
//     m -> (img, oh, ow) over an output grid; A row m, tap t = src[img][oh*sm+dh[t]][ow*sm+dw[t]][c]
//     (zero outside the source), B = packed weights [n][t][c] (K contiguous for both operands).
//   forward : src = x,  sm = stride, dst = y (every pixel),            B = w_fwd [Cout][kh][kw][Cin]
//   dgrad   : src = dy, sm = 1,      dst = dx pixels of one stride-parity class (oh*s+ph, ow*s+pw),
//             B = w_dgrad[class] [Cin][taps(class)][Cout]; classes without taps just store zeros.
//
// Tile: 128 (m) x BN (n) x 128 bytes of K per step (64 bf16 / 32 fp32 channels of one tap),
// 256 threads = 4 waves (2 x 2), wave tile 64 x BN/2 from 32x32 MFMA blocks
// (v_mfma_f32_32x32x16_bf16, or v_mfma_f32_32x32x2_f32 for exact fp32).  Global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write), ONE LDS buffer and two barriers per K
// step (35 KiB, 116 VGPRs: four workgroups per CU overlap each other's loads and MFMAs); LDS rows are
// 128 B with the 16-byte chunk index XORed by (row>>1)&7 -- applied to the DMA SOURCE address, the
// LDS image is lane-linear -- so every ds_read_b128 lane group hits 16 distinct bank slots.
// Stride-1 3x3 shapes: igemm_halo3x3*_kernel below (pixel patch loaded once per chunk), and for bf16
// the ping-pong core of conv_pp.hip where launch_gather_bn's measured dispatch picks it.
#include <cstdlib>

#include "common.h"

#ifndef CNK_ABLATE
#define CNK_ABLATE 0      // debug builds only (tools/ablate.sh): 1 no MFMA phase, 2 no global loads, 4 no LDS reads, 16 no tap masks
#endif
namespace {

constexpr int BM = 128;
constexpr int MAX_TAPS = 32;
// LDS header of the implicit-GEMM kernels: row_dst[BM], then the (dh, dw) tap table of the gather kernel -- p.dh[t] with a
// run-time t is a VECTOR load from the kernarg segment whose wait (vmcnt, in order) sat in front of every K step's DMAs
constexpr int IG_HDR = (BM + MAX_TAPS) * (int)sizeof(int);

struct GatherParams {
    const void* src;
    const void* wpk;
    void* dst;
    const float* bias;
    // optional second K segment of a dense 1x1 launch (CAT kernels): C2 more channels per pixel read from src2 [M][C2]; the
    // weight image rows are then [ntaps * Cpad + C2] long.  Used by the data gradient that carries the backward of the BN
    // layer behind the conv (cnk_conv2d_dgrad_bnfold): K = [dy' | conv input].
    const void* src2;
    int C2;
    const void* addend;      // optional tensor shaped like dst: dst = conv + addend (fused grad sum)
    const unsigned char* addend_mask;   // optional packed ReLU mask (1 byte per 16-B vector) gating addend
    float* stats;            // optional [tiles_m][Nout][3] BN partials (n, mean, M2) of the OUTPUT
    // optional fused backward of the BN(+ReLU) layer that produced this dgrad's input tensor
    const void* bnf_x;       // that layer's input, shaped like dst (nullptr: off)
    const float* bnf_mean;
    const float* bnf_invstd;
    const float* bnf_gamma;
    const float* bnf_beta;
    const unsigned char* bnf_mask;
    float* bnf_partial;      // [slices][Nout][2]
    int bnf_relu, bnf_tile0; // tile0: first slice of this launch (stride-parity classes)
    int relu;                // inference epilogue: dst = max(conv + bias [+ addend], 0)
    int is_dgrad;            // data-gradient launch
    int chunked;             // grouped conv: n-tile t (64 wide) reduces over channel chunk [64t, 64t+64) only
    int Cpad;                // K length of one tap in the weight image (C rounded up to the K step; == C on the grid)
    int N, IH, IW, C;
    int OH, OW;
    int FH, FW, Nout;
    int sm_h, sm_w;
    int dm_h, dm_w, d_h0, d_w0;
    int ntaps;
    int M;
    int tiles_n;
    int tiles_total;
    FastDiv fd_ohw, fd_ow;
    signed char dh[MAX_TAPS], dw[MAX_TAPS];
};

template <typename T> struct Mma32;
template <> struct Mma32<bf16> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                      __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct Mma32<float> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
    }
};

__device__ __forceinline__ int swz(int row, int chunk) { return (chunk ^ ((row >> 1) & 7)) << 4; }

// 16 zero bytes in global memory: the source of LDS-DMA loads for padded / out-of-range A rows
__device__ __attribute__((aligned(16))) unsigned int cnk_zero16[4] = {0, 0, 0, 0};
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef const __attribute__((address_space(1))) void* gptr_t;

// Shared epilogue of the implicit-GEMM kernels (see the comment inside): bias, LDS-staged
// coalesced row stores, optional gradient addend (+ReLU mask), optional BN statistics partials.
// EDGE: Nout is any multiple of one 16-byte vector (not of BN): the last n-tile has ncols < BN valid columns.
template <typename T, int BN, bool EDGE = false, bool BIAS_DONE = false>
__device__ __forceinline__ void igemm_epilogue(const GatherParams& p, unsigned char* smem,
                                               const int* row_dst, f32x16 (&acc)[2][BN / 64],
                                               int tile_m, int tile_n, int tid, int wm, int wn,
                                               int l31, int lh) {
    constexpr int PER16 = ElemTraits<T>::PER16;
    constexpr int NJ = BN / 64;
    // Epilogue.  The MFMA computed the TRANSPOSED block (weights as the row operand), so lane
    // (l31, lh) holds pixel row l31 and, in registers 4g..4g+3, the four consecutive output
    // channels 8g + 4*lh + {0..3}: one 8-byte (bf16) / 16-byte (fp32) LDS store per group.  The
    // tile is then written to HBM as whole rows, 16 bytes per lane, coalesced.
    constexpr int ES = (int)sizeof(T);
    constexpr int CS = BN * ES + 16;                 // staged row stride
    constexpr int CPR = BN * ES / 16;                // 16-B chunks per row
    constexpr int RPP = 256 / CPR;                   // rows per pass
    constexpr int NP = BM / RPP;                     // passes
    constexpr int GP = NP > 4 ? 4 : NP;              // passes whose side loads are in flight together
    unsigned char* Cs = smem + IG_HDR;
    unsigned char* dst = (unsigned char*)p.dst;
    const int crow = tid / CPR, cc = tid - crow * CPR;
    const int ncols = EDGE ? p.Nout - tile_n * BN : BN;          // valid columns of this n-tile
    const bool colok = !EDGE || cc * PER16 < ncols;              // this thread's 16-byte column chunk exists
    float s8[PER16], q8[PER16];
#pragma unroll
    for (int e = 0; e < PER16; ++e) s8[e] = q8[e] = 0.f;
    // stage the accumulators first: their registers are free before the side operands come in
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int row = wm * 64 + i * 32 + l31;
                const int col = wn * (BN / 2) + j * 32 + 8 * gq + 4 * lh;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * gq + e];
                if (!BIAS_DONE && p.bias && (!EDGE || col < ncols)) {          // (Nout % 4 == 0: a group of four is all in or all out)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += p.bias[tile_n * BN + col + e];
                }
                if (p.relu && !p.addend) {           // with an addend the ReLU follows the add (below)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                if constexpr (ES == 2) {
                    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    *(bf16x4*)(Cs + row * CS + col * 2) = o;
                } else {
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    *(f32x4*)(Cs + row * CS + col * 4) = o;
                }
            }
    // fused BN backward (dgrad only): per-thread coefficients of this thread's PER16 channels
    float bmu[PER16], bsc[PER16], bsf[PER16], is_fin = 0.f;
    const bool bnf = p.bnf_x != nullptr;
    const bool bnf_remask = bnf && p.bnf_relu && !p.bnf_mask;
    if (bnf && tid < BN && (!EDGE || tid < ncols)) is_fin = p.bnf_invstd[tile_n * BN + tid];      // for the partial store at the end
    if (bnf && colok) {
        // all four coefficient vectors requested back to back and unconditionally (an absent gamma / beta reads the mean
        // array instead): a per-element "load or zero" select made hipcc branch around every load and wait vmcnt(0) after
        // each -- 8 dependent L2 round trips per tile (cdna_hip_programming.md 5, trap (c))
        const int c0 = tile_n * BN + cc * PER16;
        const float* gp = bnf_remask ? p.bnf_gamma : p.bnf_mean;
        const float* bp = bnf_remask ? p.bnf_beta : p.bnf_mean;
        float mu[PER16], gm[PER16], is[PER16], bt[PER16];
#pragma unroll
        for (int e = 0; e < PER16; ++e) { mu[e] = p.bnf_mean[c0 + e]; gm[e] = gp[c0 + e]; is[e] = p.bnf_invstd[c0 + e]; bt[e] = bp[c0 + e]; }
#pragma unroll
        for (int e = 0; e < PER16; ++e) {
            const float sc = gm[e] * is[e];
            bmu[e] = mu[e];
            bsc[e] = bnf_remask ? sc : 0.f;
            bsf[e] = bnf_remask ? bt[e] - mu[e] * sc : 0.f;
        }
    }
    if (!p.addend && !bnf) {
        // plain store loop (forward, and data gradients without side operands)
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int row = ps * RPP + crow;
            const int pix = row_dst[row];
            if (pix >= 0 && colok) {
                Vec16<T> v;
                v.v = *(const decltype(v.v)*)(Cs + row * CS + cc * 16);
                const size_t goff = ((size_t)pix * p.Nout + tile_n * BN) * ES + cc * 16;
                *(decltype(v.v)*)(dst + goff) = v.v;
                if (p.stats) {
#pragma unroll
                    for (int e = 0; e < PER16; ++e) {
                        float f = v.get(e);
                        s8[e] += f;
                        q8[e] += f * f;
                    }
                }
            }
        }
    } else
#pragma unroll
    for (int g0 = 0; g0 < NP; g0 += GP) {
        // Side operands of GP row passes (gradient addend, BN input, their masks) are requested
        // together, ahead of the barrier and of the stores: the stores may alias them as far as the
        // compiler knows, so loads left inside the store loop would run one latency per pass.
        Vec16<T> av[GP], xv[GP];
        unsigned am[GP], km[GP];
        int pixs[GP];
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const int pix = colok ? row_dst[(g0 + u) * RPP + crow] : -1;
            pixs[u] = pix;
            am[u] = km[u] = 0xFFu;
            if (pix >= 0) {
                const size_t goff = ((size_t)pix * p.Nout + tile_n * BN) * ES + cc * 16;
                if (p.addend) {
                    av[u].v = *(const decltype(av[u].v)*)((const unsigned char*)p.addend + goff);
                    if (p.addend_mask) am[u] = p.addend_mask[goff >> 4];
                }
                if (bnf) {
                    xv[u].v = *(const decltype(xv[u].v)*)((const unsigned char*)p.bnf_x + goff);
                    if (p.bnf_relu && p.bnf_mask) km[u] = p.bnf_mask[goff >> 4];
                }
            }
        }
        if (g0 == 0) __syncthreads();               // staged tile complete (side loads already in flight)
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const int row = (g0 + u) * RPP + crow;
            const int pix = pixs[u];
            if (pix >= 0) {
                Vec16<T> v;
                v.v = *(const decltype(v.v)*)(Cs + row * CS + cc * 16);
                const size_t goff = ((size_t)pix * p.Nout + tile_n * BN) * ES + cc * 16;
                if (p.addend) {          // same rounding as a separate add kernel: round(round(conv) + addend)
                    // with a mask the addend is dy of the block output and the mask its ReLU pattern:
                    // addend*mask is the residual-branch gradient, formed here instead of being written
                    // out by the BN backward and read back
#pragma unroll
                    for (int e = 0; e < PER16; ++e) v.set(e, v.get(e) + (((am[u] >> e) & 1u) ? av[u].get(e) : 0.f));
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < PER16; ++e) v.set(e, v.get(e) > 0.f ? v.get(e) : 0.f);
                    }
                }
                if (bnf) {
                    // dx' = dx * ReLU-mask of the producing BN layer; s8 += dx', q8 += dx'*(x - mean)
                    unsigned keep = km[u];
                    if (bnf_remask) {
                        keep = 0;
#pragma unroll
                        for (int e = 0; e < PER16; ++e) keep |= (xv[u].get(e) * bsc[e] + bsf[e] > 0.f ? 1u : 0u) << e;
                    }
#pragma unroll
                    for (int e = 0; e < PER16; ++e) {
                        const float dv = ((keep >> e) & 1u) ? v.get(e) : 0.f;
                        v.set(e, dv);
                        s8[e] += dv;
                        q8[e] += dv * (xv[u].get(e) - bmu[e]);
                    }
                }
                *(decltype(v.v)*)(dst + goff) = v.v;
                if (p.stats) {
#pragma unroll
                    for (int e = 0; e < PER16; ++e) {
                        float f = v.get(e);
                        s8[e] += f;
                        q8[e] += f * f;
                    }
                }
            }
        }
    }
    // Fused BN statistics of the values just stored (i.e. of the ROUNDED output, exactly what the
    // BN kernels would read back): per-channel sum / sum-of-squares over this tile's <=128 rows ->
    // (n, mean, M2) partial, merged across tiles by bn_finalize (Chan).  Saves one full read pass.
    // With bnf (data gradient) the two sums are those of the BN backward instead.
    if (p.stats || bnf) {
        // the reduction scratch reuses the staged tile (everyone has finished reading it), which
        // keeps the epilogue LDS at 35 KiB so four workgroups fit a CU
        __syncthreads();
        float* red = (float*)Cs;                      // [2][RPP][BN]
#pragma unroll
        for (int e = 0; e < PER16; ++e) {
            red[crow * BN + cc * PER16 + e] = s8[e];
            red[(RPP + crow) * BN + cc * PER16 + e] = q8[e];
        }
        __syncthreads();
        if (tid < BN && (!EDGE || tid < ncols)) {
            float S = 0.f, Q = 0.f;
#pragma unroll 4
            for (int r = 0; r < RPP; ++r) {
                S += red[r * BN + tid];
                Q += red[(RPP + r) * BN + tid];
            }
            if (bnf) {
                const int c = tile_n * BN + tid;
                float* o = p.bnf_partial + ((size_t)(p.bnf_tile0 + tile_m) * p.Nout + c) * 2;
                o[0] = S;
                o[1] = Q * is_fin;
            } else {
                const int nvalid = min(BM, p.M - tile_m * BM);
                const float mean = S / (float)nvalid;
                float m2 = Q - S * mean;
                float* o = p.stats + ((size_t)tile_m * p.Nout + tile_n * BN + tid) * 3;
                o[0] = (float)nvalid;
                o[1] = mean;
                o[2] = m2 > 0.f ? m2 : 0.f;
            }
        }
    }
}

// EDGE: channel counts that are multiples of one 16-byte vector but not of the K step / BN: chunks past C fetch
// zeros, the weight image is zero-padded to [tiles_n*BN][taps][Cpad], the epilogue skips the missing columns.
template <typename T, int BN, bool EDGE = false, bool CAT = false>
__global__ __launch_bounds__(256, 4) void igemm_gather_kernel(const GatherParams p) {
    constexpr int PER16 = ElemTraits<T>::PER16;      // elements per 16-B chunk
    constexpr int KSTEP = 8 * PER16;                 // channels per K step (128 B)
    constexpr int NJ = BN / 64;                      // 32-wide MFMA blocks per wave along n
    constexpr int B_LOADS = BN / 32;                 // 16-B chunks per thread for the B tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* row_dst = (int*)smem;                       // [128] dst pixel per tile row (-1: none)
    int* tap_tab = row_dst + BM;                     // [MAX_TAPS] (dh << 16) | (dw & 0xffff)
    unsigned char* As = smem + IG_HDR;               // [128][128 B]
    unsigned char* Bs = As + BM * 128;               // [BN][128 B]

    // XCD-aware remap: blocks b, b+8, ... share an XCD (and its L2); give each XCD a contiguous
    // run of tiles so neighbours (same A rows, other n tile; adjacent pixels) share L2 lines.
    int bid = blockIdx.x;
    {
        int nwg = p.tiles_total, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int chunk = tid & 7, lrow = tid >> 3;      // loader role: 8 threads cover one 128-B row

    // per-thread A rows: lrow + 32*i
    int a_base[4], a_ih[4], a_iw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = tile_m * BM + lrow + 32 * i;
        if (m < p.M) {
            uint32_t img = fd_div(m, p.fd_ohw);
            uint32_t rem = m - img * p.fd_ohw.d;
            uint32_t oh = fd_div(rem, p.fd_ow);
            uint32_t ow = rem - oh * p.fd_ow.d;
            a_base[i] = img * p.IH;
            a_ih[i] = oh * p.sm_h;
            a_iw[i] = ow * p.sm_w;
        } else {
            a_base[i] = 0;
            a_ih[i] = -100000;       // never in range
            a_iw[i] = 0;
        }
    }
    if (tid < BM) {
        int m = tile_m * BM + tid, v = -1;
        if (m < p.M) {
            uint32_t img = fd_div(m, p.fd_ohw);
            uint32_t rem = m - img * p.fd_ohw.d;
            uint32_t oh = fd_div(rem, p.fd_ow);
            uint32_t ow = rem - oh * p.fd_ow.d;
            v = (img * p.FH + oh * p.dm_h + p.d_h0) * p.FW + ow * p.dm_w + p.d_w0;
        }
        row_dst[tid] = v;
    }
    if (tid >= BM && tid < BM + MAX_TAPS) {
        const int t = tid - BM;
        tap_tab[t] = t < p.ntaps ? (((int)p.dh[t]) << 16) | (((int)p.dw[t]) & 0xffff) : 0;
    }

    const T* src = (const T*)p.src;
    const T* wpk = (const T*)p.wpk;
    // grouped conv (block-diagonal image): this 64-wide n-tile only sees its own 64 input channels
    const int csteps = p.chunked ? 64 / KSTEP : p.Cpad / KSTEP;
    const int cs0 = p.chunked ? tile_n * (64 / KSTEP) : 0;
    const int nk1 = p.ntaps * csteps;
    const int nk = CAT ? nk1 + p.C2 / KSTEP : nk1;
    const size_t wrow = (size_t)p.ntaps * p.Cpad + (CAT ? p.C2 : 0);

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    if constexpr (CAT) {
        // the bias goes in as the accumulators' start value: its loads are issued here, under the first K step's DMAs, instead of
        // as dependent round trips inside the epilogue (where they cost the 56x56 launches +60 %); CAT launches always carry one
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 b4 = *(const f32x4*)(p.bias + tile_n * BN + wn * (BN / 2) + j * 32 + 8 * gq + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][j][4 * gq + e] = acc[1][j][4 * gq + e] = b4[e];
            }
    }
    auto compute = [&]() {
        const unsigned char* Ab = As;
        const unsigned char* Bb = Bs;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 fa[2], fb[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int r = wm * 64 + i * 32 + l31;
                fa[i] = *(const u32x4*)(Ab + r * 128 + swz(r, 2 * s + lh));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                int r = wn * (BN / 2) + j * 32 + l31;
                fb[j] = *(const u32x4*)(Bb + r * 128 + swz(r, 2 * s + lh));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) Mma32<T>::run(acc[i][j], fb[j], fa[i]);   // C^T block
        }
    };

    // LDS-DMA main loop: global_load_lds_dwordx4 writes 64 lanes x 16 B = 8 rows x 128 B straight
    // into LDS (no staging VGPRs, no ds_write).  The LDS image is lane-linear, so the XOR swizzle
    // goes on the SOURCE: lane (row r, physical chunk pc) fetches logical chunk pc ^ ((r>>1)&7).
    // Padded / out-of-range rows fetch a 16-byte zero block instead.  ONE buffer, two barriers per
    // K step: 35 KiB of LDS and 116 VGPRs put four workgroups on a CU, and on this latency-bound
    // loop that beat every double-buffered / register-staged variant tried (2 or 3 workgroups/CU).
    lds_u8* As3 = (lds_u8*)As;
    lds_u8* Bs3 = (lds_u8*)Bs;
    int ld_t = 0, ld_cs = 0, ld_k = 0;
    auto issue = [&]() {
        // (CAT: after the nk1 steps of the first segment the same pixel rows are read from src2 -- dense 1x1, no taps, no
        // padding -- and the weight rows continue behind the first segment; everything that differs is wave-uniform)
        const bool seg2 = CAT && ld_k >= nk1;
        const int t = ld_t, cb = seg2 ? (ld_k - nk1) * KSTEP : (cs0 + ld_cs) * KSTEP;
        if constexpr (CAT) ++ld_k;
        if (!seg2 && ++ld_cs == csteps) { ld_cs = 0; ++ld_t; }
        const int tp = seg2 ? 0 : tap_tab[t];        // LDS (uniform address: broadcast), lgkmcnt only
        const int dh = tp >> 16, dw = (int)(short)(tp & 0xffff);
        const size_t wo = seg2 ? (size_t)p.ntaps * p.Cpad + cb : (size_t)t * p.Cpad + cb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lrow + 32 * i;
            const int c = chunk ^ ((r >> 1) & 7);
            const int ih = a_ih[i] + dh, iw = a_iw[i] + dw;
            const bool ok = (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW &&
                            (!EDGE || cb + c * PER16 < p.C);
            const T* gp = ok ? src + ((size_t)(a_base[i] + ih) * p.IW + iw) * p.C + cb + c * PER16
                             : (const T*)cnk_zero16;
            if constexpr (CAT) {
                // second segment: the row of src2 is the DESTINATION pixel of this tile row (the conv input at the pixel whose
                // gradient the row computes)
                if (seg2) {
                    const int pix = row_dst[r];
                    gp = pix >= 0 ? (const T*)p.src2 + (size_t)pix * p.C2 + cb + c * PER16 : (const T*)cnk_zero16;
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)gp, (__attribute__((address_space(3))) void*)(As3 + (wave * 8 + 32 * i) * 128), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int r = lrow + 32 * i;
            const int c = chunk ^ ((r >> 1) & 7);
            const T* gp = wpk + (size_t)(tile_n * BN + r) * wrow + wo + c * PER16;
            __builtin_amdgcn_global_load_lds((gptr_t)gp, (__attribute__((address_space(3))) void*)(Bs3 + (wave * 8 + 32 * i) * 128), 16, 0, 0);
        }
    };
    __syncthreads();                                 // tap table (and row_dst) written
    for (int ks = 0; ks < nk; ++ks) {
#if !(CNK_ABLATE & 2)
        issue();
#endif
        __syncthreads();                             // vmcnt(0) + barrier: the tile has landed
#if !(CNK_ABLATE & 1)
        compute();
#endif
        __syncthreads();                             // every wave is done reading the only buffer
    }
    if (nk == 0) __syncthreads();

    igemm_epilogue<T, BN, EDGE, CAT>(p, smem, row_dst, acc, tile_m, tile_n, tid, wm, wn, l31, lh);
}

// 3x3 / stride 1 / "same" convolutions (forward, and the data gradient of such a conv): halo reuse.
// All nine taps of an output pixel read input pixels within +-1 row / +-1 column, i.e. within
// +-(W+1) positions of the flattened NHW pixel index.  So instead of gathering a fresh 128-row A
// tile for every tap (9x the input bytes through L2), the workgroup loads the pixel range
// [m0-W-1, m0+128+W+1) of one 64(32)-channel chunk into LDS ONCE and the nine taps read SHIFTED
// windows of that patch; only the 16 KiB weight tile is streamed per tap.  L2 bytes per K step drop
// from 32 KiB to ~19 KiB (BN=128) / 24 -> 11 KiB (BN=64).  Image-border and cross-image positions
// are real (wrong) pixels in the patch, so each lane keeps a 9-bit validity mask per MFMA row and
// zeroes the fragment when the tap falls outside the image.
template <typename T, int BN>
__global__ __launch_bounds__(256, 3) void igemm_halo3x3_kernel(const GatherParams p) {
    constexpr int PER16 = ElemTraits<T>::PER16;
    constexpr int KSTEP = 8 * PER16;
    constexpr int NJ = BN / 64;
    constexpr int B_LOADS = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* row_dst = (int*)smem;
    const int W = p.IW, NP = BM + 2 * W + 2, NPG = (NP + 7) >> 3;
    unsigned char* Ps = smem + IG_HDR;                   // [NPG*8][128 B] pixel patch
    unsigned char* Bs = Ps + NPG * 8 * 128;              // [BN][128 B]
    int bid = blockIdx.x;
    {
        int nwg = p.tiles_total, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    const int chunk = tid & 7, lrow = tid >> 3;
    if (tid < BM) {
        int m = tile_m * BM + tid, v = -1;
        if (m < p.M) {
            uint32_t img = fd_div(m, p.fd_ohw), rem = m - img * p.fd_ohw.d;
            uint32_t oh = fd_div(rem, p.fd_ow), ow = rem - oh * p.fd_ow.d;
            v = (img * p.FH + oh * p.dm_h + p.d_h0) * p.FW + ow * p.dm_w + p.d_w0;
        }
        row_dst[tid] = v;
    }
    unsigned vm[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = tile_m * BM + wm * 64 + i * 32 + l31;
        unsigned bits = 0;
        if (m < p.M) {
            uint32_t img = fd_div(m, p.fd_ohw), rem = m - img * p.fd_ohw.d;
            uint32_t oh = fd_div(rem, p.fd_ow), ow = rem - oh * p.fd_ow.d;
            for (int t = 0; t < 9; ++t)
                if ((unsigned)((int)oh + p.dh[t]) < (unsigned)p.IH && (unsigned)((int)ow + p.dw[t]) < (unsigned)p.IW)
                    bits |= 1u << t;
        }
        vm[i] = bits;
    }
    const T* src = (const T*)p.src;
    const T* wpk = (const T*)p.wpk;
    const int csteps = p.C / KSTEP;
    const size_t wrow = (size_t)9 * p.C;
    const long long P0 = (long long)tile_m * BM - W - 1, Mtot = (long long)p.N * p.IH * p.IW;
    lds_u8* Ps3 = (lds_u8*)Ps;
    lds_u8* Bs3 = (lds_u8*)Bs;
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int cs = 0; cs < csteps; ++cs) {
        // pixel patch of this channel chunk: row group g = 8 consecutive pixels, one wave instruction
#if !(CNK_ABLATE & 2)
        for (int gq = wave; gq < NPG; gq += 4) {
            const int j = gq * 8 + (lane >> 3);
            const long long q = P0 + j;
            const int c = chunk ^ ((j >> 1) & 7);
            const T* gp = (q >= 0 && q < Mtot && j < NP) ? src + (size_t)q * p.C + cs * KSTEP + c * PER16
                                                         : (const T*)cnk_zero16;
            __builtin_amdgcn_global_load_lds((gptr_t)gp, (__attribute__((address_space(3))) void*)(Ps3 + gq * 8 * 128), 16, 0, 0);
        }
#endif
        for (int t = 0; t < 9; ++t) {
#if !(CNK_ABLATE & 2)
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                const int r = lrow + 32 * i;
                const int c = chunk ^ ((r >> 1) & 7);
                const T* gp = wpk + (size_t)(tile_n * BN + r) * wrow + (size_t)t * p.C + cs * KSTEP + c * PER16;
                __builtin_amdgcn_global_load_lds((gptr_t)gp, (__attribute__((address_space(3))) void*)(Bs3 + (wave * 8 + 32 * i) * 128), 16, 0, 0);
            }
#endif
            __syncthreads();                         // vmcnt(0) + barrier: patch (t == 0) and weights landed
            const int off = W + 1 + p.dh[t] * W + p.dw[t];
#if !(CNK_ABLATE & 1)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u32x4 fa[2], fb[NJ];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int j = wm * 64 + i * 32 + l31 + off;
#if (CNK_ABLATE & 4)
                    u32x4 v = u32x4{(unsigned)j, (unsigned)s, (unsigned)t, (unsigned)cs};
#else
                    u32x4 v = *(const u32x4*)(Ps + j * 128 + swz(j, 2 * s + lh));
#endif
#if (CNK_ABLATE & 16)
                    fa[i] = v;
#else
                    const bool ok = (vm[i] >> t) & 1u;
                    fa[i] = ok ? v : u32x4{0, 0, 0, 0};
#endif
                }
#pragma unroll
                for (int jn = 0; jn < NJ; ++jn) {
                    const int r = wn * (BN / 2) + jn * 32 + l31;
#if (CNK_ABLATE & 4)
                    fb[jn] = u32x4{(unsigned)r, (unsigned)s, (unsigned)t, (unsigned)cs};
#else
                    fb[jn] = *(const u32x4*)(Bs + r * 128 + swz(r, 2 * s + lh));
#endif
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < NJ; ++jn) Mma32<T>::run(acc[i][jn], fb[jn], fa[i]);
            }
#endif
            __syncthreads();                         // weights (and at t == 8 the patch) free again
        }
    }
    igemm_epilogue<T, BN>(p, smem, row_dst, acc, tile_m, tile_n, tid, wm, wn, l31, lh);
}

// Same algorithm with the weight stream software-pipelined: three weight buffers, the tiles of steps
// g+1 and g+2 are in flight (LDS-DMA) while step g is computed.  An LDS-DMA is a pending write on
// the vector-memory counter, so the pipeline uses counted `s_waitcnt vmcnt(N)` and raw `s_barrier`
// (a __syncthreads() would drain everything): a tile is read one barrier AFTER the wait that
// retires it, and a buffer is re-issued only after the barrier that follows its last read.
// 32 KiB patch + 3 x 16 KiB weights = 80 KiB -> two workgroups per CU.
#define CNK_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
template <typename T, int BN>
__global__ __launch_bounds__(256, 2) void igemm_halo3x3_pf_kernel(const GatherParams p) {
    constexpr int PER16 = ElemTraits<T>::PER16;
    constexpr int KSTEP = 8 * PER16;
    constexpr int NJ = BN / 64;
    constexpr int B_LOADS = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* row_dst = (int*)smem;
    const int W = p.IW, NP = BM + 2 * W + 2, NPG = (NP + 7) >> 3;
    unsigned char* Ps = smem + IG_HDR;                   // [NPG*8][128 B]
    unsigned char* Bs = Ps + NPG * 8 * 128;              // [3][BN][128 B]
    int bid = blockIdx.x;
    {
        int nwg = p.tiles_total, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    const int chunk = tid & 7, lrow = tid >> 3;
    if (tid < BM) {
        int m = tile_m * BM + tid, v = -1;
        if (m < p.M) {
            uint32_t img = fd_div(m, p.fd_ohw), rem = m - img * p.fd_ohw.d;
            uint32_t oh = fd_div(rem, p.fd_ow), ow = rem - oh * p.fd_ow.d;
            v = (img * p.FH + oh * p.dm_h + p.d_h0) * p.FW + ow * p.dm_w + p.d_w0;
        }
        row_dst[tid] = v;
    }
    unsigned vm[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = tile_m * BM + wm * 64 + i * 32 + l31;
        unsigned bits = 0;
        if (m < p.M) {
            uint32_t img = fd_div(m, p.fd_ohw), rem = m - img * p.fd_ohw.d;
            uint32_t oh = fd_div(rem, p.fd_ow), ow = rem - oh * p.fd_ow.d;
            for (int t = 0; t < 9; ++t)
                if ((unsigned)((int)oh + p.dh[t]) < (unsigned)p.IH && (unsigned)((int)ow + p.dw[t]) < (unsigned)p.IW)
                    bits |= 1u << t;
        }
        vm[i] = bits;
    }
    const T* src = (const T*)p.src;
    const T* wpk = (const T*)p.wpk;
    const int csteps = p.C / KSTEP, G = csteps * 9;
    const size_t wrow = (size_t)9 * p.C;
    const long long P0 = (long long)tile_m * BM - W - 1, Mtot = (long long)p.N * p.IH * p.IW;
    lds_u8* Ps3 = (lds_u8*)Ps;
    lds_u8* Bs3 = (lds_u8*)Bs;
    auto issue_patch = [&](int cs) {
        for (int gq = wave; gq < NPG; gq += 4) {
            const int j = gq * 8 + (lane >> 3);
            const long long q = P0 + j;
            const int c = chunk ^ ((j >> 1) & 7);
            const T* gp = (q >= 0 && q < Mtot && j < NP) ? src + (size_t)q * p.C + cs * KSTEP + c * PER16
                                                         : (const T*)cnk_zero16;
            __builtin_amdgcn_global_load_lds((gptr_t)gp, (__attribute__((address_space(3))) void*)(Ps3 + gq * 8 * 128), 16, 0, 0);
        }
    };
    auto issue_b = [&](int gstep) {
        const int cs = gstep / 9, t = gstep - cs * 9, buf = gstep % 3;
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int r = lrow + 32 * i;
            const int c = chunk ^ ((r >> 1) & 7);
            const T* gp = wpk + (size_t)(tile_n * BN + r) * wrow + (size_t)t * p.C + cs * KSTEP + c * PER16;
            __builtin_amdgcn_global_load_lds((gptr_t)gp, (__attribute__((address_space(3))) void*)(Bs3 + buf * (BN * 128) + (wave * 8 + 32 * i) * 128), 16, 0, 0);
        }
    };
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const uint32_t ps_lds = (uint32_t)(uintptr_t)Ps3, bs_lds = (uint32_t)(uintptr_t)Bs3;
    // tap -> patch row offset, fetched BEFORE any LDS-DMA is in flight: p.dh[t] with a run-time t is
    // a vector load from the kernarg segment, and hipcc drains vmcnt(0) for any ordinary load result
    // while DMAs are pending.
    // (kept packed, 8 bits each, in three scalars: no memory access and no 9x unroll needed to index them)
    uint32_t offp[3] = {0, 0, 0};
#pragma unroll
    for (int t = 0; t < 9; ++t) offp[t >> 2] |= (uint32_t)(W + 1 + p.dh[t] * W + p.dw[t]) << ((t & 3) * 8);
    issue_patch(0);
    issue_b(0);
    if (G > 1) issue_b(1);
    int gs = 0;
    for (int cs = 0; cs < csteps; ++cs) {
        for (int t = 0; t < 9; ++t, ++gs) {
            const int off_t = (int)(((t < 4 ? offp[0] : (t < 8 ? offp[1] : offp[2])) >> ((t & 3) * 8)) & 0xFFu);
            // retire the weights of step gs (and, at the first tap of a chunk, the patch issued after them)
            if (t == 0 || gs + 1 >= G) { CNK_WAIT_VM(0); }
            else if (B_LOADS == 4) { CNK_WAIT_VM(4); }
            else { CNK_WAIT_VM(2); }
            __builtin_amdgcn_s_barrier();            // every wave's share has landed; step gs-1 fully read
            if (gs + 2 < G) issue_b(gs + 2);         // buffer (gs+2)%3 held step gs-1
            // Fragment reads are inline asm: hipcc cannot prove that a ds_read does not alias an
            // LDS-DMA still in flight and would put `s_waitcnt vmcnt(0)` in front of them.  Reads of
            // sub-step s+1 are issued before the MFMAs of sub-step s; a counted lgkmcnt retires
            // exactly the older group.
            const uint32_t Bb = bs_lds + (gs % 3) * (BN * 128);
            uint32_t a_addr[2], b_addr[NJ];
            int a_row[2], b_row[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) { a_row[i] = wm * 64 + i * 32 + l31 + off_t; a_addr[i] = ps_lds + a_row[i] * 128; }
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) { b_row[jn] = wn * (BN / 2) + jn * 32 + l31; b_addr[jn] = Bb + b_row[jn] * 128; }
            const bool ok0 = (vm[0] >> t) & 1u, ok1 = (vm[1] >> t) & 1u;
            u32x4 fa[2][2], fb[2][NJ];
            auto issue_reads = [&](int s, int set) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(fa[set][i]) : "v"(a_addr[i] + swz(a_row[i], 2 * s + lh)));
#pragma unroll
                for (int jn = 0; jn < NJ; ++jn)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(fb[set][jn]) : "v"(b_addr[jn] + swz(b_row[jn], 2 * s + lh)));
            };
            issue_reads(0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int cur = s & 1;
                if (s < 3) {
                    issue_reads(s + 1, cur ^ 1);
                    if (NJ == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                u32x4 za = ok0 ? fa[cur][0] : u32x4{0, 0, 0, 0};
                u32x4 zb = ok1 ? fa[cur][1] : u32x4{0, 0, 0, 0};
#pragma unroll
                for (int jn = 0; jn < NJ; ++jn) {
                    Mma32<T>::run(acc[0][jn], fb[cur][jn], za);
                    Mma32<T>::run(acc[1][jn], fb[cur][jn], zb);
                }
            }
        }
        if (cs + 1 < csteps) {
            // the single patch buffer: every wave must be done with this chunk before the next
            // chunk's pixels are DMA'd over it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue_patch(cs + 1);
        }
    }
    __syncthreads();                                 // LDS is reused by the epilogue
    igemm_epilogue<T, BN>(p, smem, row_dst, acc, tile_m, tile_n, tid, wm, wn, l31, lh);
}

// algorithmic cost of ONE launch: the source tensor and the weight image read once, the output written once,
// plus the fused side operands (gradient addend, the BN layer's input for its backward reduction)
static double gather_flops(const GatherParams& p) { return 2.0 * p.M * p.Nout * ((double)p.C * p.ntaps + (p.src2 ? p.C2 : 0)); }
static double gather_bytes(const GatherParams& p, size_t esz) {
    double out = (double)p.M * p.Nout;
    double e = (double)p.N * p.IH * p.IW * p.C + out + (double)p.Nout * p.ntaps * p.C;
    if (p.src2) e += (double)p.M * p.C2 + (double)p.Nout * p.C2;
    if (p.addend) e += out;
    if (p.bnf_x) e += out;
    return e * esz;
}

template <typename T, int BN, bool EDGE = false, bool CAT = false> int launch_gather(const GatherParams& p, hipStream_t s) {
    size_t pipe = BM * 128 + BN * 128;
    size_t epi = (size_t)BM * (BN * sizeof(T) + 16);
    size_t red = 2 * (256 / (BN * sizeof(T) / 16)) * BN * sizeof(float);     // aliases the staged tile
    if (red > epi) epi = red;
    size_t lds = IG_HDR + (pipe > epi ? pipe : epi);
    ProfScope kscope(CNK_KF_GATHER, s, gather_flops(p), gather_bytes(p, sizeof(T)));
    cnk_count(CNK_CNT_GATHER);
    hipLaunchKernelGGL((igemm_gather_kernel<T, BN, EDGE, CAT>), dim3(p.tiles_total), dim3(256), lds, s, p);
    CNK_CHECK_LAUNCH("igemm_gather");
    return 0;
}

static bool pick_bn128(int M, int Nout) {
    // BN = 128 when it divides Nout and the grid still fills the chip, else 64
    int tiles_m = ceil_div(M, BM);
    return (Nout % 128 == 0) && ((int64_t)tiles_m * (Nout / 128) >= 512);
}

static int igemm_halo_enabled() { return cnk_sw().igemm_halo != 0; }

// Which halo kernel: the plain one runs three workgroups per CU, the weight-prefetching one two
// (80 KiB of LDS each) but each of them ~1.16x faster than a third of a CU.  With T tiles on 256 CUs
// the launch takes ceil(T/768) rounds of 3 or ceil(T/512) rounds of 2; prefetching pays when it
// saves a round (measured: 256->256 @14x14, 784 tiles: 776 vs 687 TFLOP/s; every other ResNet
// shape is faster on the plain kernel).  CNK_IGEMM_HALO_PF=0/1 forces one of them.
static bool igemm_halo_pf(int tiles, int bn) {
    const int forced = cnk_sw().igemm_halo_pf;
    if (forced >= 0) return forced != 0;
    if (bn != 128 || tiles <= 768) return false;
    return ceil_div(tiles, 512) * 2 * 1.16 < ceil_div(tiles, 768) * 3;
}

template <typename T, int BN> size_t halo_lds(const GatherParams& p) {
    size_t pipe = (size_t)((BM + 2 * p.IW + 2 + 7) / 8) * 8 * 128 + (size_t)BN * 128 * (igemm_halo_pf(p.tiles_total, BN) ? 3 : 1);
    size_t epi = (size_t)BM * (BN * sizeof(T) + 16);
    size_t red = 2 * (256 / (BN * sizeof(T) / 16)) * BN * sizeof(float);
    if (red > epi) epi = red;
    return IG_HDR + (pipe > epi ? pipe : epi);
}

// 3x3 taps within +-1, stride 1, output grid == source grid ("same" conv or its data gradient)
static bool halo_shape(const GatherParams& p) {
    if (p.ntaps != 9 || p.sm_h != 1 || p.sm_w != 1 || p.dm_h != 1 || p.dm_w != 1) return false;
    if (p.IH != p.OH || p.IW != p.OW || p.FH != p.OH || p.FW != p.OW || p.d_h0 || p.d_w0) return false;
    for (int t = 0; t < 9; ++t)
        if (p.dh[t] < -1 || p.dh[t] > 1 || p.dw[t] < -1 || p.dw[t] > 1) return false;
    return (long long)p.N * p.IH * p.IW == p.M;
}

template <typename T, int BN> int launch_halo(const GatherParams& p, hipStream_t s) {
    const size_t lds = halo_lds<T, BN>(p);
    ProfScope kscope(CNK_KF_HALO, s, gather_flops(p), gather_bytes(p, sizeof(T)));
    cnk_count(igemm_halo_pf(p.tiles_total, BN) ? CNK_CNT_HALO_PF : CNK_CNT_HALO);
    if (igemm_halo_pf(p.tiles_total, BN))
        hipLaunchKernelGGL((igemm_halo3x3_pf_kernel<T, BN>), dim3(p.tiles_total), dim3(256), lds, s, p);
    else
        hipLaunchKernelGGL((igemm_halo3x3_kernel<T, BN>), dim3(p.tiles_total), dim3(256), lds, s, p);
    CNK_CHECK_LAUNCH("igemm_halo3x3");
    return 0;
}

// a 1x1 / stride-1 conv (or its data gradient) whose operand is the plain [M][C] matrix and whose output pixel is the row index
static bool dense_1x1(const GatherParams& p) {
    return p.ntaps == 1 && p.dh[0] == 0 && p.dw[0] == 0 && p.sm_h == 1 && p.sm_w == 1 && p.dm_h == 1 && p.dm_w == 1 &&
           p.d_h0 == 0 && p.d_w0 == 0 && p.IH == p.OH && p.IW == p.OW && p.FH == p.OH && p.FW == p.OW &&
           (p.Cpad == 0 || p.Cpad == p.C);
}

template <typename T> int launch_gather_bn(GatherParams& p, hipStream_t s) {
    int tiles_m = ceil_div(p.M, BM);
    constexpr int KS = 8 * ElemTraits<T>::PER16;
    if (p.Cpad == 0) p.Cpad = ceil_div(p.C, KS) * KS;
    if (p.C % KS || p.Nout % 64) {                   // off the tile grid: 64-wide tiles with edge handling
        p.tiles_n = ceil_div(p.Nout, 64);
        p.tiles_total = tiles_m * p.tiles_n;
        if (p.tiles_total == 0) return 0;
        return launch_gather<T, 64, true>(p, s);
    }
    if (sizeof(T) == 2 && !p.chunked && halo_shape(p) && pp3x3_ok(p.N, p.IH, p.IW, p.C, p.Nout, p.is_dgrad, p.bnf_x != nullptr)) {
        // stride-1 3x3 on the ping-pong core (conv_pp.hip); partials are per 64-row slice there
        PPArgs a = {};
        a.src = p.src; a.wpk = p.wpk; a.dst = p.dst; a.bias = p.bias; a.addend = p.addend; a.addend_mask = p.addend_mask;
        a.stats = p.stats; a.bnf_x = p.bnf_x; a.bnf_mean = p.bnf_mean; a.bnf_invstd = p.bnf_invstd; a.bnf_gamma = p.bnf_gamma;
        a.bnf_beta = p.bnf_beta; a.bnf_mask = p.bnf_mask; a.bnf_partial = p.bnf_partial; a.bnf_relu = p.bnf_relu;
        a.bnf_tile0 = p.bnf_tile0; a.relu = p.relu;
        a.N = p.N; a.H = p.IH; a.W = p.IW; a.C = p.C; a.Nout = p.Nout; a.dgrad = p.is_dgrad;
        for (int t = 0; t < 9; ++t) { a.dh[t] = p.dh[t]; a.dw[t] = p.dw[t]; }
        a.algo_bytes = gather_bytes(p, sizeof(T));
        return pp3x3_launch(a, s);
    }
    if (p.src2) CNK_CHECK_ARG(sizeof(T) == 2 && !p.chunked && p.ntaps == 1 && p.dh[0] == 0 && p.dw[0] == 0 && p.sm_h == 1 && p.sm_w == 1 &&
                                  (p.Cpad == 0 || p.Cpad == p.C) && p.C2 % KS == 0 && !p.relu && !p.stats && p.bias,
                              "igemm: a second K segment needs a bf16 one-tap launch with C2 on the K-step grid and a bias");
    if (sizeof(T) == 2 && !p.chunked && dense_1x1(p) && !p.relu && !(p.stats && (p.addend || p.bnf_x)))
        if (const int nj = gemm1x1_pick(p.M, p.C + (p.src2 ? p.C2 : 0), p.Nout, p.addend || p.bnf_x)) {
            G1Args a = {};
            a.a = p.src; a.b = p.wpk; a.c = p.dst; a.bias = p.bias; a.stats = p.stats;
            a.a2 = p.src2; a.K2 = p.src2 ? p.C2 : 0;
            a.addend = p.addend; a.addend_mask = p.addend_mask;
            a.bnf_x = p.bnf_x; a.bnf_mean = p.bnf_mean; a.bnf_invstd = p.bnf_invstd; a.bnf_gamma = p.bnf_gamma; a.bnf_beta = p.bnf_beta;
            a.bnf_mask = p.bnf_mask; a.bnf_partial = p.bnf_partial; a.bnf_relu = p.bnf_relu; a.bnf_tile0 = p.bnf_tile0;
            a.M = p.M; a.K = p.C; a.Nout = p.Nout;
            a.algo_flops = gather_flops(p); a.algo_bytes = gather_bytes(p, sizeof(T));
            return gemm1x1_launch(nj, a, s);
        }
    bool use128 = !p.chunked && pick_bn128(p.M, p.Nout);
    int bn = use128 ? 128 : 64;
    p.tiles_n = p.Nout / bn;
    p.tiles_total = tiles_m * p.tiles_n;
    if (p.tiles_total == 0) return 0;
    if (igemm_halo_enabled() && !p.chunked && halo_shape(p)) {
        // three workgroups per CU need <= 53 KiB each; wider images fall back to the gather kernel
        size_t lds = use128 ? halo_lds<T, 128>(p) : halo_lds<T, 64>(p);
        if (lds <= (size_t)(igemm_halo_pf(p.tiles_total, bn) ? 80 : 53) * 1024) return use128 ? launch_halo<T, 128>(p, s) : launch_halo<T, 64>(p, s);
    }
    if constexpr (sizeof(T) == 2)
        if (p.src2) return use128 ? launch_gather<T, 128, false, true>(p, s) : launch_gather<T, 64, false, true>(p, s);
    return use128 ? launch_gather<T, 128>(p, s) : launch_gather<T, 64>(p, s);
}

}  // namespace

// ---- host entry points used by api.hip --------------------------------------------------------
// grouped conv the chunked kernels can run: see cnk_conv_desc
bool igemm_grouped_ok(const cnk_conv_desc* d) {
    const int g = conv_groups(d);
    if (g == 1) return true;
    return d->Cin == d->Cout && d->Cin % 64 == 0 && d->Cin % g == 0 && 64 % (d->Cin / g) == 0 && g != d->Cin;
}

bool igemm_supported(const cnk_conv_desc* d, int pass) {
    if (conv_groups(d) > 1 && !igemm_grouped_ok(d)) return false;
    int kstep = d->dtype == CNK_BF16 ? 64 : 32, vec = kstep / 8;
    int cred = pass == CNK_PASS_FWD ? d->Cin : d->Cout;
    int nout = pass == CNK_PASS_FWD ? d->Cout : d->Cin;
    if (d->kh * d->kw > MAX_TAPS) return false;
    if (cred % kstep == 0 && nout % 64 == 0) return true;          // on the tile grid
    // off the grid (Inception / SqueezeNet / MobileNet widths): any multiple of one 16-byte vector, as long as
    // the zero padding of K stays below 2x; narrower layers keep the small-channel kernels
    if (conv_groups(d) > 1) return false;
    if (cred % vec || nout % vec || nout < 16) return false;
    // a data gradient with a narrow reduction but a wide output (Inception's 1x1 "reduce" convs: Cout 16/24, Cin
    // 192..512) writes N*H*W*Cin values per 16..24-deep dot product: HBM-bound whatever the MFMA does with the
    // zero-padded K, and 6x faster here than on the direct kernel
    if (pass == CNK_PASS_DGRAD && nout >= 64) return true;
    return cred >= kstep / 2;
}

// elements of the (zero-padded) weight image of one pass: [ceil(nout/64)*64][taps][ceil(cred/kstep)*kstep]
size_t igemm_image_elems(const cnk_conv_desc* d, int pass, int ntaps) {
    int kstep = d->dtype == CNK_BF16 ? 64 : 32;
    int cred = pass == CNK_PASS_FWD ? d->Cin : d->Cout;
    int nout = pass == CNK_PASS_FWD ? d->Cout : d->Cin;
    return (size_t)(ceil_div(nout, 64) * 64) * ntaps * (ceil_div(cred, kstep) * kstep);
}

bool igemm_on_grid(const cnk_conv_desc* d, int pass) {
    int kstep = d->dtype == CNK_BF16 ? 64 : 32;
    int cred = pass == CNK_PASS_FWD ? d->Cin : d->Cout;
    int nout = pass == CNK_PASS_FWD ? d->Cout : d->Cin;
    return cred % kstep == 0 && nout % 64 == 0;
}

// stride-1 "same" 3x3 conv (or its data gradient) that launch_gather_bn hands to the ping-pong core
// (for the data gradient: with the BN backward fused, the case the partial slices are asked for)
static bool pp_desc(const cnk_conv_desc* d, int pass) {
    if (d->dtype != CNK_BF16 || conv_groups(d) > 1 || conv_dil(d) > 1) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h != 1 || d->pad_w != 1) return false;
    const int cred = pass == CNK_PASS_FWD ? d->Cin : d->Cout, nout = pass == CNK_PASS_FWD ? d->Cout : d->Cin;
    return pp3x3_ok(d->N, d->H, d->W, cred, nout, pass == CNK_PASS_DGRAD, 1);
}

int64_t igemm_fwd_stat_slices(const cnk_conv_desc* d) {
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    if (pp_desc(d, CNK_PASS_FWD)) return pp3x3_slices(d->N, d->H, d->W, d->Cin, d->Cout, 0);
    if (d->dtype == CNK_BF16 && conv_groups(d) == 1 && d->kh == 1 && d->kw == 1 && d->stride_h == 1 && d->stride_w == 1 &&
        d->pad_h == 0 && d->pad_w == 0 && d->Cin % 64 == 0 && d->Cout % 64 == 0 && gemm1x1_pick((int64_t)d->N * Ho * Wo, d->Cin, d->Cout, 0))
        return gemm1x1_slices((int64_t)d->N * Ho * Wo);
    return ceil_div(d->N * Ho * Wo, BM);
}

int igemm_fwd(const cnk_conv_desc* d, const void* x, const void* w_fwd, const float* bias, void* y,
              float* stats, const void* residual, int relu, hipStream_t s) {
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    GatherParams p = {};
    p.src = x; p.wpk = w_fwd; p.dst = y; p.bias = bias; p.stats = stats;
    p.addend = residual; p.relu = relu; p.chunked = conv_groups(d) > 1;
    p.N = d->N; p.IH = d->H; p.IW = d->W; p.C = d->Cin;
    p.OH = Ho; p.OW = Wo; p.FH = Ho; p.FW = Wo; p.Nout = d->Cout;
    p.sm_h = d->stride_h; p.sm_w = d->stride_w;
    p.dm_h = 1; p.dm_w = 1; p.d_h0 = 0; p.d_w0 = 0;
    p.ntaps = d->kh * d->kw;
    for (int r = 0; r < d->kh; ++r)
        for (int c = 0; c < d->kw; ++c) {
            p.dh[r * d->kw + c] = (signed char)(r * conv_dil(d) - d->pad_h);
            p.dw[r * d->kw + c] = (signed char)(c * conv_dil(d) - d->pad_w);
        }
    p.M = d->N * Ho * Wo;
    p.fd_ohw = make_fastdiv(Ho * Wo);
    p.fd_ow = make_fastdiv(Wo);
    return d->dtype == CNK_BF16 ? launch_gather_bn<bf16>(p, s) : launch_gather_bn<float>(p, s);
}

// Enumerates the stride-parity classes of the data gradient in the order the packed dgrad
// weights use.  Returns the number of classes; taps[c] lists (kh,kw) pairs of class c.
int dgrad_classes(const cnk_conv_desc* d, DgradClass* out /* [stride_h*stride_w] */) {
    int n = 0;
    for (int ph = 0; ph < d->stride_h; ++ph)
        for (int pw = 0; pw < d->stride_w; ++pw) {
            DgradClass& c = out[n++];
            c.ph = ph; c.pw = pw;
            c.OH = ph < d->H ? (d->H - ph + d->stride_h - 1) / d->stride_h : 0;
            c.OW = pw < d->W ? (d->W - pw + d->stride_w - 1) / d->stride_w : 0;
            c.ntaps = 0;
            for (int r = 0; r < d->kh; ++r) {
                int th = ph + d->pad_h - r * conv_dil(d);
                if (th % d->stride_h) continue;
                for (int q = 0; q < d->kw; ++q) {
                    int tw = pw + d->pad_w - q * conv_dil(d);
                    if (tw % d->stride_w) continue;
                    if (c.ntaps < MAX_TAPS) {
                        c.kh[c.ntaps] = (signed char)r; c.kw[c.ntaps] = (signed char)q;
                        c.dh[c.ntaps] = (signed char)(th / d->stride_h);
                        c.dw[c.ntaps] = (signed char)(tw / d->stride_w);
                    }
                    c.ntaps++;
                }
            }
        }
    return n;
}

// slices of the fused BN-backward partials: one per 128-row tile of every stride-parity class
int64_t igemm_dgrad_bn_slices(const cnk_conv_desc* d) {
    if (d->stride_h * d->stride_w > 16 || d->Cin % (d->dtype == CNK_BF16 ? 8 : 4)) return 0;
    if (pp_desc(d, CNK_PASS_DGRAD)) return pp3x3_slices(d->N, d->H, d->W, d->Cout, d->Cin, 1);
    if (d->dtype == CNK_BF16 && conv_groups(d) == 1 && d->kh == 1 && d->kw == 1 && d->stride_h == 1 && d->stride_w == 1 &&
        d->pad_h == 0 && d->pad_w == 0 && d->Cin % 64 == 0 && d->Cout % 64 == 0 &&
        gemm1x1_pick((int64_t)d->N * d->H * d->W, d->Cout, d->Cin, 1))
        return gemm1x1_slices((int64_t)d->N * d->H * d->W);          // the ring GEMM (conv_gemm1x1.hip): one record per 256 rows
    DgradClass cls[16];
    int nc = dgrad_classes(d, cls);
    int64_t n = 0;
    for (int c = 0; c < nc; ++c) n += ceil_div64((int64_t)d->N * cls[c].OH * cls[c].OW, BM);
    return n;
}

// Data gradient of a 1x1 / stride-1 conv (no padding) with the backward of the BN layer behind it folded in (csrc/bnfold.hip):
// K = [g (Cout channels) | x (Cin channels)], weight image w_fold [Cin][Cout + Cin], bias c (+ addend, + fused BN backward of the
// layer in front).
int igemm_dgrad_bnfold(const cnk_conv_desc* d, const void* g, const void* x, const void* w_fold, const float* bias_fold,
                       void* dx, const void* addend, const unsigned char* addend_mask, const cnk_bn_bwd_fuse* bn, hipStream_t s) {
    CNK_CHECK_ARG(d->stride_h == 1 && d->stride_w == 1 && d->kh == 1 && d->kw == 1, "igemm_dgrad_bnfold: a 1x1 / stride-1 conv");
    GatherParams p = {};
    p.src = g; p.wpk = w_fold; p.dst = dx; p.is_dgrad = 1;
    p.addend = addend; p.addend_mask = addend_mask;
    p.N = d->N; p.IH = d->H; p.IW = d->W; p.C = d->Cout;
    p.OH = d->H; p.OW = d->W; p.FH = d->H; p.FW = d->W; p.Nout = d->Cin;
    p.sm_h = 1; p.sm_w = 1; p.dm_h = 1; p.dm_w = 1;
    p.ntaps = 1; p.dh[0] = 0; p.dw[0] = 0;
    p.src2 = x; p.C2 = d->Cin; p.bias = bias_fold;
    p.M = d->N * d->H * d->W;
    if (p.M == 0) return 0;
    if (bn) {
        p.bnf_x = bn->x; p.bnf_mean = bn->mean; p.bnf_invstd = bn->invstd; p.bnf_gamma = bn->gamma;
        p.bnf_beta = bn->beta; p.bnf_mask = bn->relu_mask; p.bnf_partial = bn->partial;
        p.bnf_relu = bn->relu; p.bnf_tile0 = 0;
    }
    p.fd_ohw = make_fastdiv(d->H * d->W);
    p.fd_ow = make_fastdiv(d->W);
    return launch_gather_bn<bf16>(p, s);
}

int igemm_dgrad(const cnk_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                const void* addend, const unsigned char* addend_mask, const cnk_bn_bwd_fuse* bn,
                hipStream_t s) {
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    DgradClass cls[16];
    CNK_CHECK_ARG(d->stride_h * d->stride_w <= 16, "igemm_dgrad: stride too large");
    int nc = dgrad_classes(d, cls);
    size_t esz = d->dtype == CNK_BF16 ? 2 : 4;
    size_t woff = 0;
    int tile0 = 0;
    for (int c = 0; c < nc; ++c) {
        const DgradClass& k = cls[c];
        GatherParams p = {};
        p.src = dy; p.wpk = (const char*)w_dgrad + woff * esz; p.dst = dx; p.bias = nullptr;
        p.addend = addend; p.addend_mask = addend_mask; p.chunked = conv_groups(d) > 1; p.is_dgrad = 1;
        p.N = d->N; p.IH = Ho; p.IW = Wo; p.C = d->Cout;
        p.OH = k.OH; p.OW = k.OW; p.FH = d->H; p.FW = d->W; p.Nout = d->Cin;
        p.sm_h = 1; p.sm_w = 1;
        p.dm_h = d->stride_h; p.dm_w = d->stride_w; p.d_h0 = k.ph; p.d_w0 = k.pw;
        p.ntaps = k.ntaps;
        for (int t = 0; t < k.ntaps; ++t) { p.dh[t] = k.dh[t]; p.dw[t] = k.dw[t]; }
        p.M = d->N * k.OH * k.OW;
        woff += igemm_image_elems(d, CNK_PASS_DGRAD, k.ntaps);
        if (p.M == 0) continue;
        if (bn) {
            p.bnf_x = bn->x; p.bnf_mean = bn->mean; p.bnf_invstd = bn->invstd; p.bnf_gamma = bn->gamma;
            p.bnf_beta = bn->beta; p.bnf_mask = bn->relu_mask; p.bnf_partial = bn->partial;
            p.bnf_relu = bn->relu; p.bnf_tile0 = tile0;
            tile0 += ceil_div(p.M, BM);
        }
        p.fd_ohw = make_fastdiv(k.OH * k.OW);
        p.fd_ow = make_fastdiv(k.OW);
        int rc = d->dtype == CNK_BF16 ? launch_gather_bn<bf16>(p, s) : launch_gather_bn<float>(p, s);
        if (rc) return rc;
    }
    return 0;
}
