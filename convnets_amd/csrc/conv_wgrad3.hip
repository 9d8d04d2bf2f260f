// Weight gradient of the 3x3 / stride-1 / pad-1 convolutions, bf16: ALL NINE TAPS per workgroup (autograd site
// /root/reference/src/basemodel.py:278; layers src/resnet.py:171, src/vggnet.py:69).
//
//   dw[co][t][ci] = sum_m dy[m][co] * x[m + tap t][ci]            m = (img, h, w) flat, zero outside the image
//
// conv_wgrad.hip runs one tap per workgroup: every tap re-reads the dy tile and its own x tile, and rocprofv3 shows the
// result -- MFMA busy 21-23 %, traffic == algorithmic: the kernel is bound by the global -> LDS path (64 B/clk/CU asked at
// full MFMA rate, ~28 delivered).  Here a workgroup owns a (TCO output channels) x (64 input channels) tile for all nine taps
// over a range of pixels (split-K): per K step it stages ONE dy tile and only the NEW rows of a sliding x window, and each
// wave keeps nine 32x32 accumulator blocks (one per tap) -- 3.6x fewer global -> LDS bytes per flop, the dy fragment of a
// k-step is read from LDS once for nine MFMAs.
//
// x lives in LDS in PADDED coordinates (row pitch P = W+1 with one shared zero cell between image rows, image pitch
// (H+1)*P with one shared zero row between images, as in conv_pp.hip): a tap is the uniform row offset dh*P+dw and the
// borders are zero cells, no masks.  The window is a ring of 512 padded rows; per step only the rows that entered the
// window are fetched (LDS-DMA, a fixed number of 1-KiB pieces so the counted vmcnt stays static; a piece that would run
// too far ahead goes to a dump area).  dy stays in compact pixel order (three stages); a 64-entry table per step maps
// the step's pixels to their padded rows (lanes supply one row address each to ds_read_b64_tr_b16, so the rows of a
// fragment need not be equidistant).  Both operands are stored as [row][32 channels] images (64-byte rows, one image per
// 32-channel block): four consecutive rows then cover all 64 banks and the transposing reads are conflict-free without any
// swizzle, which keeps the per-read address arithmetic at two VALU instructions (add the tap offset, wrap into the ring).
//
// Output: fp32 slabs [split][co][t][ci], summed in fixed order by wgrad_reduce_kernel (conv_wgrad.hip) -- deterministic.
#include <cstdlib>

#include "common.h"
#include "wgrad_chain.h"

#ifndef W3_ABL
#define W3_ABL 0      // debug builds only (tools/ablate_w3.sh): 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no MFMA, 8 no slab store
#endif

namespace {

constexpr int W3_RING = 512;                       // padded rows per x ring (two rings: input channels 0-31, 32-63)
constexpr int W3_RINGB = W3_RING * 64;             // 32 KiB

__device__ __attribute__((aligned(16))) unsigned int w3_zero16[4] = {0, 0, 0, 0};
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct W3Params {
    const void* dy;
    const void* x;
    float* slab;
    int N, H, W, Cin, Cout, M;
    int P, IP;
    int ksteps, ksteps_per_split, splits;
    int tiles_co, tiles_ci;
    int toff[9];                                   // (kh-1)*P + (kw-1)
    FastDiv fd_hw, fd_w, fd_ip, fd_p;
    WgradPending prev;       // the slab sum of the PREVIOUS weight gradient of this stream, done in this launch's prologue (wgrad_chain.h)
};

template <int TCO_, int KP_> struct W3Cfg {
    static constexpr int TCO = TCO_, KP = KP_;
    static constexpr int NWM = TCO / 32;                     // waves along co (32 channels each), 2 along ci
    static constexpr int NW = NWM * 2, NT = NW * 64;
    static constexpr int SS = KP / 16;                       // MFMA k-steps per K step
    static constexpr int DY_IMG = KP * 64;                   // bytes of one 32-channel dy image of a stage
    static constexpr int DY_STAGE = NWM * DY_IMG;
    static constexpr int DYPW = DY_STAGE / 1024 / NW;        // dy DMA pieces per wave per step
    static constexpr int XPW = KP == 64 ? 2 : 3;             // x DMA pieces per wave per step
    static constexpr int XROWS = NW * XPW / 2 * 16;          // padded rows fetched per step (per ring)
    static constexpr int NPW = DYPW + XPW;
    static constexpr int LDS_DY = 2 * W3_RINGB, LDS_MAP = LDS_DY + 3 * DY_STAGE, LDS_DUMP = LDS_MAP + 2 * KP * 4;
    static constexpr int LDS_BYTES = LDS_DUMP + 1024;
};

__device__ __forceinline__ void w3_dma16(const void* g, uint32_t lds_addr) {
    unsigned keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
#define W3_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory")
// transposing LDS read of one half fragment (4 x bf16 per lane) and the wait that retires it (form (ii) of
// cdna_hip_programming.md 5.7 item 1: the wait names the registers "+v", so no consumer is scheduled above it)
__device__ __forceinline__ void w3_tr_read(u32x2& d, uint32_t addr) {
#if (W3_ABL & 2)
    asm volatile("" : "+v"(d) : "v"(addr));
#else
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(d) : "v"(addr));
#endif
}
template <int N> __device__ __forceinline__ void w3_wait_lgkm(u32x2& a, u32x2& b, u32x2& c, u32x2& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N));
}
template <int N> __device__ __forceinline__ void w3_wait_lgkm2(u32x2& c, u32x2& d) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(c), "+v"(d) : "i"(N));
}

__device__ __forceinline__ int w3_pp(const W3Params& p, int m) {
    uint32_t img = fd_div(m, p.fd_hw), rem = m - img * p.fd_hw.d;
    uint32_t h = fd_div(rem, p.fd_w), w = rem - h * p.fd_w.d;
    return (int)(img * p.IP + (h + 1) * p.P + w + 1);
}

template <class CFG>
__global__ __launch_bounds__(CFG::NT, CFG::NW == 8 ? 2 : 2) void wgrad3x3_kernel(const W3Params p) {
    constexpr int KP = CFG::KP, SS = CFG::SS, NW = CFG::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the ring addressing ORs the ring base into a wrapped offset: the dynamic LDS segment must start 32-KiB aligned
    // (it starts at 0: the kernel has no static LDS)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    if (lds0 & (W3_RINGB - 1)) __builtin_trap();
    // the previous layer's slab sum (compiler-visible loads / stores: retired here, before the hand-counted DMA pipeline starts)
    if (p.prev.slab) {
        wgrad_pending_sum<CFG::NT>(p.prev);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // tile / split of this workgroup (XCD-aware: the splits of one tile share an L2)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tiles = p.tiles_co * p.tiles_ci;
    const int split = bid / tiles;
    const int tile = bid - split * tiles;
    const int tile_co = tile / p.tiles_ci, tile_ci = tile - tile_co * p.tiles_ci;
    const int ks0 = split * p.ksteps_per_split;
    const int ks1 = min(p.ksteps, ks0 + p.ksteps_per_split);

    const unsigned char* dyb = (const unsigned char*)p.dy;
    const unsigned char* xb = (const unsigned char*)p.x;
    const unsigned char* zero = (const unsigned char*)w3_zero16;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- loaders ---------------------------------------------------------------------------------------------
    // dy of K step ks into stage st: piece dp = image (dp / PPI) rows (dp % PPI)*16.., lane -> row lane/4, 16-byte unit lane&3
    auto issue_dy = [&](int ks, int st, bool live) {
        constexpr int PPI = CFG::DY_IMG / 1024;
#pragma unroll
        for (int u = 0; u < CFG::DYPW; ++u) {
            const int dp = wave * CFG::DYPW + u;
            const int img = dp / PPI, row = (dp % PPI) * 16 + (lane >> 2);
            const int m = ks * KP + row;
            const bool ok = live && m < p.M;
            const uint32_t off = ((uint32_t)m * p.Cout + tile_co * CFG::TCO + img * 32 + (lane & 3) * 8) * 2;
            w3_dma16(ok ? dyb + off : zero, lds0 + CFG::LDS_DY + st * CFG::DY_STAGE + dp * 1024);
        }
    };
    // x rows [xl, xl + XROWS) of both rings (or nothing, to the dump area, when `live` is false)
    auto issue_x = [&](int xl, bool live) {
#pragma unroll
        for (int u = 0; u < CFG::XPW; ++u) {
            const int xp = wave * CFG::XPW + u;
            const int ring = xp & 1, r = xl + (xp >> 1) * 16 + (lane >> 2);
            const uint32_t img = fd_div((uint32_t)r, p.fd_ip), rem = (uint32_t)r - img * p.fd_ip.d;
            const uint32_t rr = fd_div(rem, p.fd_p), cc = rem - rr * p.fd_p.d;
            const bool ok = live && rr >= 1 && cc >= 1 && img < (uint32_t)p.N;
            const uint32_t pix = (img * p.H + (rr - 1)) * p.W + (cc - 1);
            const uint32_t off = (pix * p.Cin + tile_ci * 64 + ring * 32 + (lane & 3) * 8) * 2;
            const uint32_t dst = live ? lds0 + ring * W3_RINGB + (((uint32_t)(xl + (xp >> 1) * 16) * 64u) & (W3_RINGB - 1))
                                      : lds0 + CFG::LDS_DUMP;
            w3_dma16(ok ? xb + off : zero, dst);
        }
    };
    // first padded row NOT needed by K step ks (its last pixel's row + the halo below it)
    auto need = [&](int ks) {
        int m = (ks + 1) * KP - 1;
        m = m < p.M - 1 ? m : p.M - 1;
        return w3_pp(p, m) + p.P + 2;
    };
    int* ppmap = (int*)(smem + CFG::LDS_MAP);
    auto fill_map = [&](int ks) {
        if (tid < KP) {
            int m = ks * KP + tid;
            m = m < p.M ? m : p.M - 1;                       // rows past M multiply zero dy rows
            ppmap[(ks & 1) * KP + tid] = w3_pp(p, m);
        }
    };

    if (ks0 < ks1) {
        // ---- prologue: window of steps ks0 and ks0+1, dy of both, the first pixel map -----------------------------
        int xl = (w3_pp(p, ks0 * KP) - p.P - 1) & ~15;
        const int want = need(ks0 + 1 < ks1 ? ks0 + 1 : ks0);
        while (xl < want) {
            issue_x(xl, true);
            xl += CFG::XROWS;
        }
        issue_dy(ks0, ks0 % 3, true);
        issue_dy(ks0 + 1, (ks0 + 1) % 3, ks0 + 1 < ks1);
        fill_map(ks0);
        W3_WAIT_VM(0);
        __syncthreads();

        // static per-lane pieces of the fragment addresses: lane l supplies pixel row 8h+q of a 16-pixel k-step and the
        // 4-channel column group; two reads (rows +0 and +4) make one 8-element MFMA fragment
        const int g = lane >> 4, q = (lane & 15) >> 2, pp4 = lane & 3, h = g >> 1;
        const int within = 32 * (g & 1) + 8 * pp4;                          // byte offset inside the 64-byte row
        const uint32_t a_lane = lds0 + CFG::LDS_DY + wm * CFG::DY_IMG + (8 * h + q) * 64 + within;
        uint32_t ring_hi = lds0 + wn * W3_RINGB;
        asm volatile("" : "+v"(ring_hi));             // a VGPR operand lets the wrap-and-base be ONE v_and_or_b32
        int offs[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) offs[t] = p.toff[t] * 64;

        for (int ks = ks0; ks < ks1; ++ks) {
            if (ks > ks0) {
                W3_WAIT_VM((W3_ABL & 1) ? 0 : CFG::NPW);                        // everything but the previous step's own pieces has landed
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // ... and this wave's pixel-map writes
                __builtin_amdgcn_s_barrier();
            }
            // pixel map of step ks+1 (the window and dy of step ks+2 are issued further down, behind the first k-step's
            // MFMAs: their address arithmetic then hides under the matrix pipe instead of delaying the step's first reads)
            if (ks + 1 < ks1) fill_map(ks + 1);
            // this step's padded rows: 2*SS table reads per lane
            uint32_t arow[SS][2];
#pragma unroll
            for (int ss = 0; ss < SS; ++ss)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    arow[ss][u] = (uint32_t)ppmap[(ks & 1) * KP + ss * 16 + 8 * h + q + 4 * u] * 64u + within;
            // (materialised BEFORE the asm reads below: a table read hipcc slipped in between them would bring its own
            // lgkmcnt wait, which -- blind to the asm reads -- drains them all)
#pragma unroll
            for (int ss = 0; ss < SS; ++ss) asm volatile("" : "+v"(arow[ss][0]), "+v"(arow[ss][1]));
            const uint32_t a_st = a_lane + (ks % 3) * CFG::DY_STAGE;
            // Fragment reads as inline asm (hipcc otherwise sinks every read next to its MFMA into one reused register
            // pair and pays the LDS round trip nine times: measured 2 reads, s_waitcnt lgkmcnt(0), 1 MFMA, ...), ROLLING one
            // k-step ahead through two register sets: the reads of tap t of k-step ss+1 are issued right before the MFMA of
            // tap t of k-step ss.  LDS returns in order and 20 reads are younger than the pair an MFMA consumes, so
            // `lgkmcnt(15)` (the counter's maximum) retires it; the last k-step drains with counted waits.
            u32x2 fa[2][2], fb[2][9][2];
#if (W3_ABL & 2)
            for (int a_ = 0; a_ < 2; ++a_) { fa[a_][0] = fa[a_][1] = u32x2{0x3f803f80u, 0x3f803f80u}; for (int t_ = 0; t_ < 9; ++t_) fb[a_][t_][0] = fb[a_][t_][1] = u32x2{0x3f803f80u, 0x3f803f80u}; }
#endif
            auto reads_a = [&](int ss, int set) {
                w3_tr_read(fa[set][0], a_st + ss * 16 * 64);
                w3_tr_read(fa[set][1], a_st + ss * 16 * 64 + 4 * 64);
            };
            auto reads_b = [&](int ss, int set, int t) {
                w3_tr_read(fb[set][t][0], ((arow[ss][0] + offs[t]) & (W3_RINGB - 1)) | ring_hi);
                w3_tr_read(fb[set][t][1], ((arow[ss][1] + offs[t]) & (W3_RINGB - 1)) | ring_hi);
            };
            reads_a(0, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) reads_b(0, 0, t);
#define W3_MMA(SET, T, NLEFT)                                                                                          \
    {                                                                                                                  \
        if ((T) == 0) w3_wait_lgkm<NLEFT>(fa[SET][0], fa[SET][1], fb[SET][T][0], fb[SET][T][1]);                       \
        else w3_wait_lgkm2<NLEFT>(fb[SET][T][0], fb[SET][T][1]);                                                       \
        const u32x4 a4 = {fa[SET][0][0], fa[SET][0][1], fa[SET][1][0], fa[SET][1][1]};                                 \
        const u32x4 b4 = {fb[SET][T][0][0], fb[SET][T][0][1], fb[SET][T][1][0], fb[SET][T][1][1]};                     \
        if (W3_ABL & 4) asm volatile("" ::"v"(a4), "v"(b4));                                                           \
        else acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, b4), acc[T], 0, 0, 0); \
    }
#define W3_PIPE(SS, SET)                                                                                               \
    reads_a((SS) + 1, (SET) ^ 1);                                                                                      \
    reads_b((SS) + 1, (SET) ^ 1, 0); W3_MMA(SET, 0, 15) reads_b((SS) + 1, (SET) ^ 1, 1); W3_MMA(SET, 1, 15)           \
    reads_b((SS) + 1, (SET) ^ 1, 2); W3_MMA(SET, 2, 15) reads_b((SS) + 1, (SET) ^ 1, 3); W3_MMA(SET, 3, 15)           \
    reads_b((SS) + 1, (SET) ^ 1, 4); W3_MMA(SET, 4, 15) reads_b((SS) + 1, (SET) ^ 1, 5); W3_MMA(SET, 5, 15)           \
    reads_b((SS) + 1, (SET) ^ 1, 6); W3_MMA(SET, 6, 15) reads_b((SS) + 1, (SET) ^ 1, 7); W3_MMA(SET, 7, 15)           \
    reads_b((SS) + 1, (SET) ^ 1, 8); W3_MMA(SET, 8, 15)
#define W3_LAST(SET)                                                                                                   \
    W3_MMA(SET, 0, 15) W3_MMA(SET, 1, 14) W3_MMA(SET, 2, 12) W3_MMA(SET, 3, 10) W3_MMA(SET, 4, 8)                      \
    W3_MMA(SET, 5, 6) W3_MMA(SET, 6, 4) W3_MMA(SET, 7, 2) W3_MMA(SET, 8, 0)
#define W3_ISSUE                                                                                                       \
    {                                                                                                                  \
        const bool more = ks + 2 < ks1;                                                                                \
        const bool xlive = more && xl < need(ks + 2);                                                                  \
        if (!(W3_ABL & 1)) {                                                                                           \
            issue_x(xl, xlive);                                                                                        \
            issue_dy(ks + 2, (ks + 2) % 3, more);                                                                      \
        }                                                                                                              \
        if (xlive) xl += CFG::XROWS;                                                                                   \
    }
            if constexpr (SS == 4) {
                W3_PIPE(0, 0)
                W3_ISSUE
                W3_PIPE(1, 1)
                W3_PIPE(2, 0)
                W3_LAST(1)
            } else {
                W3_PIPE(0, 0)
                W3_ISSUE
                W3_LAST(1)
            }
#undef W3_ISSUE
#undef W3_PIPE
#undef W3_LAST
#undef W3_MMA
        }
        W3_WAIT_VM(0);                                       // the trailing (dummy) pieces must not outlive the workgroup's LDS
    }

    // ---- slab: [split][co][t][ci] ----------------------------------------------------------------------------
    const int l31 = lane & 31, lh = lane >> 5;
    float* slab = p.slab + (size_t)split * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = tile_co * CFG::TCO + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ci = tile_ci * 64 + wn * 32 + l31;
            if (!(W3_ABL & 8) || acc[t][r] == 12345.f) slab[((size_t)co * 9 + t) * p.Cin + ci] = acc[t][r];
        }
}

typedef W3Cfg<128, 64> W3Big;       // 512 threads, 128 co x 64 ci x 9 taps, one workgroup per CU (112 KiB)
typedef W3Cfg<64, 32> W3Small;      // 256 threads,  64 co x 64 ci x 9 taps, two per CU (76 KiB)

}  // namespace

int launch_wgrad_reduce(const float* slab, float* dw, int splits, size_t slab_stride, int Cout, int Cin, int ntaps, size_t s_co,
                        size_t s_t, size_t s_ci, int inner_is_ci, int accumulate, hipStream_t s);

static int w3_max_new_rows(int H, int W, int kp) {
    // padded rows entering the window per K step: kp pixels + their row wraps + image wraps
    return kp + (kp - 1) / W + 1 + ((kp - 1) / (H * W) + 1) * (W + 1);
}

// CNK_WGRAD3=0 turns the kernel off (A/B runs, tools/bench_wgrad.py)
static bool w3_enabled() { return cnk_sw().wgrad3 != 0; }

static void w3_plan(const cnk_conv_desc* d, bool* big, int* splits, int* ksteps, int* per_split) {
    *big = d->Cout % 128 == 0;
    const int kp = *big ? 64 : 32;
    const int M = d->N * d->H * d->W;
    const int tiles = (d->Cout / (*big ? 128 : 64)) * (d->Cin / 64);
    const int pct3 = cnk_sw().wgrad3_slots_pct > 0 ? cnk_sw().wgrad3_slots_pct : (cnk_sw().wgrad_slots_pct > 0 ? cnk_sw().wgrad_slots_pct : 100);
    int slots = (*big ? 256 : 512) * pct3 / 100;          // A/B switches; the product bounds the splits by the workspace (below)
    if (slots < 1) slots = 1;
    *ksteps = ceil_div(M, kp);
    int want = tiles >= slots ? 1 : slots / tiles;
    int max_by_k = *ksteps / 4 > 0 ? *ksteps / 4 : 1;
    int sp = want < max_by_k ? want : max_by_k;
    if (sp < 1) sp = 1;
    *per_split = ceil_div(*ksteps, sp);
    *splits = ceil_div(*ksteps, *per_split);
}

bool wgrad3x3_supported(const cnk_conv_desc* d) {
    if (!w3_enabled()) return false;
    if (d->dtype != CNK_BF16 || conv_groups(d) > 1 || conv_dil(d) > 1) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h != 1 || d->pad_w != 1) return false;
    if (d->Cin % 64 || d->Cout % 64) return false;
    const bool big = d->Cout % 128 == 0;
    const int kp = big ? 64 : 32, xrows = big ? W3Big::XROWS : W3Small::XROWS;
    if (w3_max_new_rows(d->H, d->W, kp) > xrows) return false;
    // ring capacity: halo + the windows of three steps + one fetch ahead
    if (2 * (d->W + 1) + 2 + 3 * w3_max_new_rows(d->H, d->W, kp) + xrows > W3_RING) return false;
    const long long M = (long long)d->N * d->H * d->W;
    const long long padded = (long long)d->N * (d->H + 1) * (d->W + 1) + 4 * (d->W + 2);
    if (M * (d->Cin > d->Cout ? d->Cin : d->Cout) * 2 >= (1ll << 31) || padded >= (1ll << 30)) return false;
    return M >= 4096;
}

size_t wgrad3x3_workspace(const cnk_conv_desc* d) {
    bool big;
    int splits, ksteps, per;
    w3_plan(d, &big, &splits, &ksteps, &per);
    return (size_t)splits * d->Cout * 9 * d->Cin * sizeof(float);
}

int wgrad3x3(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t s,
             const WgradPending* prev, WgradPending* out) {
    W3Params p = {};
    if (prev) p.prev = *prev;
    bool big;
    w3_plan(d, &big, &p.splits, &p.ksteps, &p.ksteps_per_split);
    const size_t slab_bytes = (size_t)d->Cout * 9 * d->Cin * sizeof(float);
    // a workspace below cnk_conv2d_workspace_bytes() bounds the split-K depth (conv_wgrad.hip: wgrad_cap_splits)
    if (ws && ws_bytes / slab_bytes >= 1 && ws_bytes / slab_bytes < (size_t)p.splits) {
        p.ksteps_per_split = ceil_div(p.ksteps, (int)(ws_bytes / slab_bytes));
        p.splits = ceil_div(p.ksteps, p.ksteps_per_split);
    }
    const size_t need = (size_t)p.splits * slab_bytes;
    CNK_CHECK_ARG(ws && ws_bytes >= need, "conv2d_wgrad: workspace too small (%zu < %zu; at least one slab of %zu bytes)", ws_bytes, need, slab_bytes);
    p.dy = dy; p.x = x; p.slab = (float*)ws;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.M = d->N * d->H * d->W;
    p.P = d->W + 1; p.IP = (d->H + 1) * p.P;
    p.tiles_co = d->Cout / (big ? 128 : 64);
    p.tiles_ci = d->Cin / 64;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) p.toff[kh * 3 + kw] = (kh - 1) * p.P + (kw - 1);
    p.fd_hw = make_fastdiv(d->H * d->W);
    p.fd_w = make_fastdiv(d->W);
    p.fd_ip = make_fastdiv(p.IP);
    p.fd_p = make_fastdiv(p.P);
    const int grid = p.splits * p.tiles_co * p.tiles_ci;
    {
        ProfScope kscope(CNK_KF_WGRAD, s, 2.0 * p.M * d->Cout * (double)d->Cin * 9,
                         ((double)p.M * d->Cout + (double)p.M * d->Cin) * 2 + (double)d->Cout * 9 * d->Cin * sizeof(float));      // slabs not credited
        if (big) {
            CNK_LDS_OPTIN((wgrad3x3_kernel<W3Big>), 160 * 1024, "wgrad3x3");
            hipLaunchKernelGGL((wgrad3x3_kernel<W3Big>), dim3(grid), dim3(W3Big::NT), W3Big::LDS_BYTES, s, p);
        } else {
            CNK_LDS_OPTIN((wgrad3x3_kernel<W3Small>), 160 * 1024, "wgrad3x3");
            hipLaunchKernelGGL((wgrad3x3_kernel<W3Small>), dim3(grid), dim3(W3Small::NT), W3Small::LDS_BYTES, s, p);
        }
        CNK_CHECK_LAUNCH("wgrad3x3");
    }
    const size_t per = (size_t)d->Cout * 9 * d->Cin;
    if (out) {                                   // chained: the sum is left to the next launch of the stream (wgrad_chain.h)
        wgrad_pending_fill(*out, p.slab, dw, p.splits, per, d->Cout, d->Cin, 9, accumulate);
        return 0;
    }
    return launch_wgrad_reduce(p.slab, dw, p.splits, per, d->Cout, d->Cin, 9, (size_t)9 * d->Cin, (size_t)d->Cin, 1, 1, accumulate, s);
}
