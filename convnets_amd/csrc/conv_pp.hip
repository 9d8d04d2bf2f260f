// Ping-pong implicit-GEMM core for the 3x3 / stride-1 / "same" convolutions (forward and data gradient), bf16.
// Sites served: the bottleneck 3x3 (/root/reference/src/resnet.py:171) and the VGG 3x3 stack (src/vggnet.py:69).
//
//   C[m][n] = sum_{c,t} A[m + tap t][c] * B[n][t][c]        m = (img, oh, ow) flat, n = output channel
//
// One 512-thread workgroup per CU, PERSISTENT over a contiguous run of 256(m) x BN(n) tiles.  The eight waves form
// two groups of four (one wave of each group per SIMD) that run the SAME instruction stream one phase apart
// (group 1 executes one extra s_barrier up front):
//
//   phase L(j): 16 ds_read_b128 (all fragments of K step j = one tap of one 64-channel chunk), the LDS-DMAs of
//               step j+2's weight tile, counted vmcnt, barrier
//   phase C(j): 16 v_mfma_f32_32x32x16_bf16 from registers only, with the LDS-DMA of one 1-KiB piece of the NEXT
//               chunk's pixel patch (and its address arithmetic) issued underneath, barrier
//
// so on every SIMD one wave always owns the matrix pipe while its partner moves data: loads, LDS reads and MFMAs
// overlap inside the workgroup by construction (the r01 kernels relied on 3-4 independent workgroups per CU).
// Everything that has to stay in flight across barriers is an LDS-DMA retired by counted `s_waitcnt vmcnt(N)`;
// barriers are raw `s_barrier`; fragment reads are inline asm (hipcc would drain the DMAs in front of them).
//
// A operand = pixel patch in PADDED coordinates: row pitch P = W+1 (one shared zero cell between image rows),
// image pitch (H+1)*P (one shared zero row between images), so a tap is the uniform row offset dh*P+dw and the
// image borders are real zero cells in LDS -- no per-lane validity masks in the loop.  The patch of a 64-channel
// chunk (256 + 2P + 2 + wraps rows x 128 B) is double-buffered and reused by the nine taps; the next chunk's patch
// (possibly the next tile's) streams in during taps 0..6.  B operand = packed weights [n][t][c], a ring of three
// 128-B-row tiles.  LDS rows are 128 B with the 16-byte chunk index XORed by (row>>1)&7 (applied to the DMA SOURCE
// address, the LDS image is lane-linear) => conflict-free ds_read_b128.
//
// The epilogue runs per wave on a private 2.5-KiB staging block (no workgroup barrier): 32x32 blocks are staged
// as bf16 rows and written with 16-byte row stores; bias, the gradient addend (+ReLU bit mask), the fused BN-backward
// reduction and the BN statistics partials (one (n, mean, M2) record per 64-row wave slice) as in igemm_epilogue.
#include <cstdlib>

#include "common.h"

#ifndef PP_ABL
#define PP_ABL 0      // debug builds only (tools/ablate_pp.sh): 1 no patch DMA, 2 no weight DMA, 4 no LDS reads, 8 no MFMA
#endif

namespace {

constexpr int PP_BM = 256;
constexpr int PP_EPI = 2560;                 // per-wave staging: 32 rows x 80 B

__device__ __attribute__((aligned(16))) unsigned int pp_zero16[4] = {0, 0, 0, 0};
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef const __attribute__((address_space(1))) void* gptr_t;

struct PPParams {
    const void* src;
    const void* wpk;
    void* dst;
    const float* bias;
    const void* addend;
    const unsigned char* addend_mask;
    float* stats;
    const void* bnf_x;
    const float* bnf_mean;
    const float* bnf_invstd;
    const float* bnf_gamma;
    const float* bnf_beta;
    const unsigned char* bnf_mask;
    float* bnf_partial;
    int bnf_relu, bnf_tile0;
    int relu;
    int N, H, W, C, Nout, M;
    int P, IP;
    int csteps;
    int tiles_n, tiles_total, nwg;
    int npg;
    int toff[9];
    FastDiv fd_hw, fd_w, fd_ip, fd_p;
};

#define PP_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory")
#define PP_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int OFF> __device__ __forceinline__ void ds_read128(u32x4& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}

// LDS-DMA as inline asm (cdna_hip_programming.md 5.7): hipcc then tracks NO outstanding vector-memory operation of
// this kernel, so it never drains the pipeline in front of an LDS access or a side-operand load; every completion is
// counted by hand (PP_WAIT_VM).  M0 (the LDS destination) is saved and restored inside the statement.
__device__ __forceinline__ void dma16(const void* g, uint32_t lds_addr) {
    unsigned keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

template <int BN> struct PPState {
    static constexpr int NJ = BN / 64;            // 32-wide column blocks per wave
    static constexpr int NB = BN / 64;            // weight DMA pieces per wave per K step
    f32x16 acc[2][NJ];
    u32x4 fa[2][4], fb[NJ][4];
    int prow[2];                                   // patch row (relative) of this lane's pixel, per row block
    uint32_t bB[NJ];                               // LDS byte address of this lane's weight rows (stage 0, s = 0)
    int brow[NB];                                  // weight image element offset of this lane's DMA rows
    // uniform
    const unsigned char* src;
    const bf16* wpk;
    uint32_t lds_patch, lds_b, lds_dump, npgb;
    int wave, lane, l31, lh, wm, wn;
    int k, c, pb;                                  // tile, chunk, patch buffer of the current chunk
    int kn, cn, has_next;                          // next chunk
    int tile_m, tile_n, tile_n_next;
    int pbase_n;                                   // padded index of patch row 0 of the next chunk's tile
    int wuni, wuni_next;                           // weight element offset of (tile_n, chunk) for current / next chunk
    int post_epi, epi_ops;
};

__device__ __forceinline__ int pp_index(const PPParams& p, int m) {
    uint32_t img = fd_div(m, p.fd_hw), rem = m - img * p.fd_hw.d;
    uint32_t h = fd_div(rem, p.fd_w), w = rem - h * p.fd_w.d;
    return (int)(img * p.IP + (h + 1) * p.P + w + 1);
}

// Source pointer and LDS destination of one 1-KiB piece (8 patch rows) of chunk cn of the tile whose patch starts at
// padded index pbase.  Branch-free 32-bit arithmetic (the tensor is < 2 GiB, checked on the host) so that the caller
// can let it issue between MFMAs.
template <int BN> __device__ __forceinline__ void pp_patch_piece_addr(const PPParams& p, const PPState<BN>& st, int piece, int pbase, int cn, int pbuf, bool live,
                                                                     const unsigned char*& g, uint32_t& dst) {
    const int row = piece * 8 + (st.lane >> 3);
    const uint32_t q = (uint32_t)(pbase + row);
    const uint32_t img = fd_div(q, p.fd_ip), rem = q - img * p.fd_ip.d;
    const uint32_t r = fd_div(rem, p.fd_p), cc = rem - r * p.fd_p.d;
    const int ch = (st.lane & 7) ^ ((row >> 1) & 7);
    const bool use = live && piece < p.npg;
    const bool ok = use && r >= 1 && cc >= 1 && img < (uint32_t)p.N;
    const uint32_t pix = (img * p.H + (r - 1)) * p.W + (cc - 1);
    const uint32_t off = (pix * p.C + cn * 64 + ch * 8) * 2;
    const unsigned char* gz = (const unsigned char*)pp_zero16;
    const unsigned char* gp = st.src + off;
    g = ok ? gp : gz;
    dst = use ? st.lds_patch + pbuf * st.npgb + piece * 1024 : st.lds_dump;
}
template <int BN> __device__ __forceinline__ void pp_issue_patch_piece(const PPParams& p, const PPState<BN>& st, int piece, int pbase, int cn, int pbuf, bool live) {
    const unsigned char* g;
    uint32_t dst;
    pp_patch_piece_addr<BN>(p, st, piece, pbase, cn, pbuf, live, g, dst);
    dma16(g, dst);
}

template <int BN> __device__ __forceinline__ void pp_issue_b(const PPParams& p, const PPState<BN>& st, int wuni, int tap, int stage, bool live) {
#pragma unroll
    for (int u = 0; u < PPState<BN>::NB; ++u) {
        const bf16* g = live ? st.wpk + (size_t)(uint32_t)(st.brow[u] + wuni + tap * p.C) : (const bf16*)pp_zero16;
        dma16(g, st.lds_b + stage * (BN * 128) + (st.wave * PPState<BN>::NB + u) * 1024);
    }
}

// Phase L + phase C of K step (chunk st.c, tap T).
template <int BN, int T> __device__ __forceinline__ void pp_tap(const PPParams& p, PPState<BN>& st) {
    constexpr int NJ = PPState<BN>::NJ;
    constexpr int NB = PPState<BN>::NB;
    constexpr int STAGE = T % 3;
    // ---- phase L ---------------------------------------------------------------------------------------
    {
        const uint32_t pbuf = st.lds_patch + st.pb * st.npgb;
        const int toff = p.toff[T];
#if !(PP_ABL & 4)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t r = (uint32_t)(st.prow[i] + toff);
            const uint32_t a0 = (pbuf + ((r << 7) | ((r << 3) & 0x70u))) ^ ((uint32_t)st.lh << 4);
            ds_read128<0>(st.fa[i][0], a0);
            ds_read128<0>(st.fa[i][1], a0 ^ 0x20u);
            ds_read128<0>(st.fa[i][2], a0 ^ 0x40u);
            ds_read128<0>(st.fa[i][3], a0 ^ 0x60u);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            ds_read128<STAGE * BN * 128>(st.fb[j][0], st.bB[j]);
            ds_read128<STAGE * BN * 128>(st.fb[j][1], st.bB[j] ^ 0x20u);
            ds_read128<STAGE * BN * 128>(st.fb[j][2], st.bB[j] ^ 0x40u);
            ds_read128<STAGE * BN * 128>(st.fb[j][3], st.bB[j] ^ 0x60u);
        }
#else
        (void)toff; (void)pbuf;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(st.fa[i][s]));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(st.fb[j][s]));
#endif
        // weights of step j+2 (tap T+2 of this chunk, or tap T-7 of the next chunk) into stage (T+2)%3
#if !(PP_ABL & 2)
        if (T + 2 < 9) pp_issue_b<BN>(p, st, st.wuni, T + 2, (T + 2) % 3, true);
        else pp_issue_b<BN>(p, st, st.wuni_next, T + 2 - 9, (T + 2) % 3, st.has_next != 0);
#endif
        // retire everything issued before this phase except the patch piece of the previous phase C
        constexpr int KEEP = ((PP_ABL & 2) ? 0 : NB) + ((PP_ABL & 1) ? 0 : ((T >= 1 && T <= 7) ? 1 : 0));
        if (T == 0) {
            if (st.post_epi == 1) {
                if (st.epi_ops == NJ * 5) PP_WAIT_VM(KEEP + NJ * 5);
                else PP_WAIT_VM(KEEP + NJ * 4);
            } else PP_WAIT_VM(KEEP);
        } else PP_WAIT_VM(KEEP);
        PP_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- phase C ---------------------------------------------------------------------------------------
#if (PP_ABL & 8)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(st.fa[i][s]));
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(st.fb[j][s]));
    if (T == 0 && st.c == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) st.acc[i][j][r] = 0.f;
    }
#else
    // One piece of the next chunk's patch (taps 0..6: 56 slots).  Its address arithmetic is interleaved with the first
    // twelve MFMAs (sched_group_barrier: one MFMA, three VALU, ...) and the DMA itself goes out before the last four.
    const unsigned char* pg = nullptr;
    uint32_t pdst = 0;
    constexpr bool PATCH = T <= 6 && !(PP_ABL & 1);
    __builtin_amdgcn_s_setprio(1);
    auto mma = [&](int s) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                st.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, st.fb[j][s]), __builtin_bit_cast(bf16x8, st.fa[i][s]), st.acc[i][j], 0, 0, 0);
    };
    if (T == 0 && st.c == 0) {
        // first K step of a tile: start the accumulators from zero
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                st.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, st.fb[j][0]), __builtin_bit_cast(bf16x8, st.fa[i][0]), z, 0, 0, 0);
    } else {
        mma(0);
    }
    if (PATCH) pp_patch_piece_addr<BN>(p, st, T * 8 + st.wave, st.pbase_n, st.cn, st.pb ^ 1, st.has_next != 0, pg, pdst);
    mma(1);
    mma(2);
    if (PATCH) {
#pragma unroll
        for (int g = 0; g < 2 * NJ * 3 - 1; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x6, 4, 0);
        }
        dma16(pg, pdst);
    }
    mma(3);
    __builtin_amdgcn_s_setprio(0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// Stores as inline asm: hipcc does not count them, the K loop's vmcnt immediates do (st.epi_ops).  The trailing
// s_nop keeps the data registers intact until the store has read them (cdna_hip_programming.md 5.7 item 1).
__device__ __forceinline__ void pp_store16(void* g, const bf16x8& v) {
    const u32x4 d = __builtin_bit_cast(u32x4, v);
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(g), "v"(d) : "memory");
}
__device__ __forceinline__ void pp_store12(void* g, float a, float b, float c) {
    typedef __attribute__((ext_vector_type(3))) float f32x3;
    f32x3 v = {a, b, c};
    asm volatile("global_store_dwordx3 %0, %1, off\n\ts_nop 1" ::"v"(g), "v"(v) : "memory");
}
__device__ __forceinline__ void pp_store8(void* g, float a, float b) {
    f32x2 v = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(g), "v"(v) : "memory");
}

// Side-operand loads of the full epilogue, as inline asm like every other vector-memory operation of this kernel
// (exact vmcnt bookkeeping; cdna_hip_programming.md 5.7 item 1 form (ii): the destination is named "+v" by the wait).
__device__ __forceinline__ void pp_load16(u32x4& d, const void* g) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(g) : "memory");
}
__device__ __forceinline__ void pp_load1(unsigned& d, const void* g) {
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(d) : "v"(g) : "memory");
}

enum { PP_EPI_PLAIN = 0, PP_EPI_STATS = 1, PP_EPI_FULL = 2 };

// Per-wave epilogue of tile (tile_m, tile_n): see the header comment.  MODE: PLAIN = rounded rows only; STATS = + BN
// statistics partials; FULL = bias / ReLU / gradient addend (+mask) / fused BN backward / statistics, all run-time.
// Every vector-memory operation is inline asm and their number per wave is static: NJ*4 row stores, + NJ partial
// stores in STATS and FULL (when sums are written), so the K loop's counted vmcnt stays exact (st.epi_ops).
// FULL issues the 16 side-operand loads of a column block (4 row passes x {addend, its mask, BN input, its mask};
// an absent operand reads the zero block) BEFORE the block's first store: vmcnt retires in order, so a load waited
// for behind a store would wait for that store's round trip too.
template <int BN, int MODE> __device__ __forceinline__ void pp_epilogue(const PPParams& p, PPState<BN>& st, unsigned char* smem_epi) {
    constexpr int NJ = PPState<BN>::NJ;
    unsigned char* stg = smem_epi + st.wave * PP_EPI;
    const int lane = st.lane, l31 = st.l31, lh = st.lh;
    const int crow = lane >> 2, cc = lane & 3;
    const int R0 = st.tile_m * PP_BM + st.wm * 64;
    const int C0 = st.tile_n * BN + st.wn * (BN / 2);
    unsigned char* dst = (unsigned char*)p.dst;
    const bool bnf = MODE == PP_EPI_FULL && p.bnf_x != nullptr;
    const bool bnf_remask = bnf && p.bnf_relu && !p.bnf_mask;
    const bool has_add = MODE == PP_EPI_FULL && p.addend != nullptr;
    const bool want_sums = MODE == PP_EPI_STATS || (MODE == PP_EPI_FULL && (p.stats || bnf));
    const bool full_tile = (st.tile_m + 1) * PP_BM <= p.M;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cb = C0 + j * 32;                                  // first channel of this column block
        float s8[8], q8[8], bmu[8], bsc[8], bsf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s8[e] = q8[e] = bmu[e] = bsc[e] = bsf[e] = 0.f;
        u32x4 av[4], xv[4];
        unsigned am[4], km[4];
        int ms[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = R0 + (k >> 1) * 32 + (k & 1) * 16 + crow;
            ms[k] = m < p.M ? m : -1;
        }
        if (MODE == PP_EPI_FULL) {
            if (bnf) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ch = cb + cc * 8 + e;
                    bmu[e] = p.bnf_mean[ch];
                    bsc[e] = bnf_remask ? p.bnf_gamma[ch] * p.bnf_invstd[ch] : 0.f;
                    bsf[e] = bnf_remask ? p.bnf_beta[ch] - bmu[e] * bsc[e] : 0.f;
                }
            }
            // the coefficient loads above are hipcc's own (waited for by hipcc, ahead of every asm operation below)
            asm volatile("" : "+v"(bmu[0]), "+v"(bsc[0]), "+v"(bsf[0]));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t goff = ((size_t)(ms[k] < 0 ? 0 : ms[k]) * p.Nout + cb) * 2 + cc * 16;
                const unsigned char* z = (const unsigned char*)pp_zero16;
                const bool on = ms[k] >= 0;
                pp_load16(av[k], on && has_add ? (const unsigned char*)p.addend + goff : z);
                pp_load1(am[k], on && has_add && p.addend_mask ? p.addend_mask + (goff >> 4) : z);
                pp_load16(xv[k], on && bnf ? (const unsigned char*)p.bnf_x + goff : z);
                pp_load1(km[k], on && bnf && p.bnf_relu && p.bnf_mask ? p.bnf_mask + (goff >> 4) : z);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // stage block (i, j): lane (l31, lh) owns pixel row l31 and channels 8g + 4lh + {0..3}
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = st.acc[i][j][4 * g + e];
                if (MODE == PP_EPI_FULL) {
                    if (p.bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += p.bias[cb + 8 * g + 4 * lh + e];
                    }
                    if (p.relu && !has_add) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                }
                bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                *(bf16x4*)(stg + l31 * 80 + (8 * g + 4 * lh) * 2) = o;
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int k = i * 2 + h2;
                bf16x8 v = *(const bf16x8*)(stg + (h2 * 16 + crow) * 80 + cc * 16);
                const int m = ms[k];
                const size_t goff = ((size_t)(m < 0 ? 0 : m) * p.Nout + cb) * 2 + cc * 16;
                if (MODE == PP_EPI_FULL) {
                    // retire this pass's four loads: younger are the 4*(3-k) loads of the later passes and the k
                    // stores of the earlier ones (a partial last tile may have skipped stores: count none there)
#define PP_EPI_WAIT(K, NFULL, NPART)                                                                                    \
    if (full_tile) asm volatile("s_waitcnt vmcnt(" #NFULL ")" : "+v"(av[K]), "+v"(xv[K]), "+v"(am[K]), "+v"(km[K])::"memory"); \
    else asm volatile("s_waitcnt vmcnt(" #NPART ")" : "+v"(av[K]), "+v"(xv[K]), "+v"(am[K]), "+v"(km[K])::"memory")
                    if (k == 0) { PP_EPI_WAIT(0, 12, 12); }
                    else if (k == 1) { PP_EPI_WAIT(1, 9, 8); }
                    else if (k == 2) { PP_EPI_WAIT(2, 6, 4); }
                    else { PP_EPI_WAIT(3, 3, 0); }
#undef PP_EPI_WAIT
                }
                if (m >= 0) {
                    if (MODE == PP_EPI_FULL) {
                        const bf16x8 a8 = __builtin_bit_cast(bf16x8, av[k]);
                        const bf16x8 x8 = __builtin_bit_cast(bf16x8, xv[k]);
                        if (has_add) {
                            const unsigned amk = p.addend_mask ? am[k] : 0xFFu;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (((amk >> e) & 1u) ? (float)a8[e] : 0.f));
                            if (p.relu) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = (float)v[e] > 0.f ? v[e] : (bf16)0.f;
                            }
                        }
                        if (bnf) {
                            unsigned keep = (p.bnf_relu && p.bnf_mask) ? km[k] : 0xFFu;
                            if (bnf_remask) {
                                keep = 0;
#pragma unroll
                                for (int e = 0; e < 8; ++e) keep |= ((float)x8[e] * bsc[e] + bsf[e] > 0.f ? 1u : 0u) << e;
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float dv = ((keep >> e) & 1u) ? (float)v[e] : 0.f;
                                v[e] = (bf16)dv;
                                s8[e] += dv;
                                q8[e] += dv * ((float)x8[e] - bmu[e]);
                            }
                        }
                    }
                    if (MODE == PP_EPI_STATS || (MODE == PP_EPI_FULL && p.stats)) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = (float)v[e];
                            s8[e] += f;
                            q8[e] += f * f;
                        }
                    }
                    // one store instruction per pass as long as ANY lane has a row (a partial last tile may skip
                    // whole passes: the caller then stops trusting the static store count, see post_epi)
                    pp_store16(dst + goff, v);
                }
            }
        }
        if (want_sums) {
            // per-channel sums over this wave's 64 rows: lanes with equal (lane & 3) hold the same 8 channels
            float* red = (float*)stg;                                   // [16][32]
            float S = 0.f, Q = 0.f;
            const int chn = lane & 31, half = lane >> 5;
#pragma unroll
            for (int e = 0; e < 8; ++e) red[crow * 32 + cc * 8 + e] = s8[e];
#pragma unroll
            for (int r = 0; r < 8; ++r) S += red[(half * 8 + r) * 32 + chn];
#pragma unroll
            for (int e = 0; e < 8; ++e) red[crow * 32 + cc * 8 + e] = q8[e];
#pragma unroll
            for (int r = 0; r < 8; ++r) Q += red[(half * 8 + r) * 32 + chn];
            S += __shfl_xor(S, 32, 64);
            Q += __shfl_xor(Q, 32, 64);
            const int ch = cb + chn;
            const int nvalid = min(64, p.M - R0);
            const int slice = st.tile_m * 4 + st.wm;
            if (lane < 32 && nvalid > 0) {
                if (bnf) {
                    pp_store8(p.bnf_partial + ((size_t)(p.bnf_tile0 + slice) * p.Nout + ch) * 2, S, Q * p.bnf_invstd[ch]);
                } else {
                    const float mean = S / (float)nvalid;
                    float m2 = Q - S * mean;
                    pp_store12(p.stats + ((size_t)slice * p.Nout + ch) * 3, (float)nvalid, mean, m2 > 0.f ? m2 : 0.f);
                }
            }
        }
    }
}

template <int BN, int MODE>
__global__ __launch_bounds__(512, 2) void pp3x3_kernel(const PPParams p) {
    constexpr int NJ = PPState<BN>::NJ;
    constexpr int NB = PPState<BN>::NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* smem = smem_raw;
    PPState<BN> st;
    const int tid = threadIdx.x;
    st.lane = tid & 63;
    st.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    st.l31 = st.lane & 31;
    st.lh = st.lane >> 5;
    const int grp = st.wave >> 2, q4 = st.wave & 3;
    st.wm = grp * 2 + (q4 >> 1);
    st.wn = q4 & 1;
    st.src = (const unsigned char*)p.src;
    st.wpk = (const bf16*)p.wpk;
    st.npgb = (uint32_t)p.npg * 1024u;
    // 128-byte aligned patch rows: the swizzle XORs are applied to whole LDS addresses
    const uint32_t lds_raw = (uint32_t)(uintptr_t)(lds_u8*)smem;
    st.lds_patch = (lds_raw + 127u) & ~127u;
    smem += st.lds_patch - lds_raw;
    st.lds_b = st.lds_patch + 2 * st.npgb;
    const uint32_t lds_epi = st.lds_b + 3 * BN * 128;
    st.lds_dump = lds_epi + 8 * PP_EPI;
    unsigned char* smem_epi = smem + 2 * st.npgb + 3 * BN * 128;

    // this workgroup's run of tiles (XCD-aware: blocks b, b+8, ... share an XCD and get adjacent runs)
    int L;
    {
        const int q = p.nwg >> 3, r = p.nwg & 7, xcd = blockIdx.x & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int k0 = (int)((long long)L * p.tiles_total / p.nwg), k1 = (int)((long long)(L + 1) * p.tiles_total / p.nwg);
    if (k0 >= k1) return;

    // static per-lane addresses of the weight tile: fragment rows and DMA rows
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const uint32_t row = st.wn * (BN / 2) + j * 32 + st.l31;
        st.bB[j] = (st.lds_b + ((row << 7) | ((row << 3) & 0x70u))) ^ ((uint32_t)st.lh << 4);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int row = (st.wave * NB + u) * 8 + (st.lane >> 3);
        const int ch = (st.lane & 7) ^ ((row >> 1) & 7);
        st.brow[u] = row * 9 * p.C + ch * 8;
    }

    auto set_tile = [&](int k) {
        st.tile_m = k / p.tiles_n;
        st.tile_n = k - st.tile_m * p.tiles_n;
        const int m0 = st.tile_m * PP_BM;
        const int pbase = pp_index(p, m0) - p.P - 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = m0 + st.wm * 64 + i * 32 + st.l31;
            m = m < p.M ? m : p.M - 1;
            st.prow[i] = pp_index(p, m) - pbase;
        }
        return pbase;
    };
    auto set_next = [&]() {
        if (st.c + 1 < p.csteps) { st.kn = st.k; st.cn = st.c + 1; st.has_next = 1; }
        else { st.kn = st.k + 1; st.cn = 0; st.has_next = st.kn < k1; }
        const int kk = st.has_next ? st.kn : st.k;
        const int tm = kk / p.tiles_n, tn = kk - tm * p.tiles_n;
        st.pbase_n = pp_index(p, tm * PP_BM) - p.P - 1;
        st.wuni_next = tn * BN * 9 * p.C + st.cn * 64;
    };

#if (PP_ABL & 4)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        st.fa[0][s] = st.fa[1][s] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
        for (int j = 0; j < NJ; ++j) st.fb[j][s] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    }
#endif
    st.k = k0; st.c = 0; st.pb = 0; st.post_epi = 0;
    st.epi_ops = (MODE == PP_EPI_STATS || (MODE == PP_EPI_FULL && (p.stats || p.bnf_x))) ? NJ * 5 : NJ * 4;
    const int pbase0 = set_tile(k0);
    st.wuni = st.tile_n * BN * 9 * p.C;
    set_next();
    // prologue: the first chunk's whole patch, weights of steps 0 and 1
    for (int piece = st.wave; piece < p.npg; piece += 8) pp_issue_patch_piece<BN>(p, st, piece, pbase0, 0, 0, true);
    pp_issue_b<BN>(p, st, st.wuni, 0, 0, true);
    pp_issue_b<BN>(p, st, st.wuni, 1, 1, true);
    PP_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // stagger: group 1 runs one phase behind group 0

    while (true) {
        pp_tap<BN, 0>(p, st);
        st.post_epi = 0;
        pp_tap<BN, 1>(p, st);
        pp_tap<BN, 2>(p, st);
        pp_tap<BN, 3>(p, st);
        pp_tap<BN, 4>(p, st);
        pp_tap<BN, 5>(p, st);
        pp_tap<BN, 6>(p, st);
        pp_tap<BN, 7>(p, st);
        pp_tap<BN, 8>(p, st);
        if (st.c == p.csteps - 1) {
            pp_epilogue<BN, MODE>(p, st, smem_epi);
            // a full tile issued exactly epi_ops stores; after a partial one the next wait keeps no allowance
            st.post_epi = (st.tile_m + 1) * PP_BM <= p.M ? 1 : 2;
        }
        if (!st.has_next) break;
        st.k = st.kn; st.c = st.cn; st.pb ^= 1;
        st.wuni = st.wuni_next;
        if (st.c == 0) set_tile(st.k);
        set_next();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
}

}  // namespace

// ---- host side -------------------------------------------------------------------------------------------
static int pp_patch_rows(int H, int W) {
    // worst-case padded-index span of 256 consecutive pixels, plus the halo on both sides
    const int P = W + 1;
    const int row_wraps = (PP_BM - 1) / W + 1;
    const int img_wraps = (PP_BM - 1) / (H * W) + 1;
    return PP_BM + row_wraps + img_wraps * P + 2 * P + 3;
}

static size_t pp_lds_bytes(int H, int W, int bn) {
    const int npg = (pp_patch_rows(H, W) + 7) / 8;
    return (size_t)2 * npg * 1024 + (size_t)3 * bn * 128 + 8 * PP_EPI + 1024 + 128;
}

static int pp_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CNK_PP3X3");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v;
}

// Shapes the ping-pong core takes: bf16, 3x3 taps within +-1 on an output grid equal to the source grid, channel
// counts on the 64 grid, a patch that fits (W <= 56) and at least 8 tiles.
bool pp3x3_ok(int N, int H, int W, int C, int Nout) {
    if (!pp_enabled()) return false;
    if (C % 64 || Nout % 64) return false;
    const int npg = (pp_patch_rows(H, W) + 7) / 8;
    if (npg > 56) return false;
    const int bn = Nout % 128 == 0 ? 128 : 64;
    if (pp_lds_bytes(H, W, bn) > 160 * 1024) return false;
    const long long M = (long long)N * H * W;
    if (M >= (1ll << 30) || M * (C > Nout ? C : Nout) * 2 >= (1ll << 31)) return false;
    const long long tiles = (M + PP_BM - 1) / PP_BM * (Nout / bn);
    return tiles >= 8;
}

int64_t pp3x3_slices(int N, int H, int W) { return ((int64_t)N * H * W + 63) / 64; }

int pp3x3_launch(const PPArgs& a, hipStream_t s) {
    PPParams p = {};
    p.src = a.src; p.wpk = a.wpk; p.dst = a.dst; p.bias = a.bias; p.addend = a.addend; p.addend_mask = a.addend_mask;
    p.stats = a.stats; p.bnf_x = a.bnf_x; p.bnf_mean = a.bnf_mean; p.bnf_invstd = a.bnf_invstd; p.bnf_gamma = a.bnf_gamma;
    p.bnf_beta = a.bnf_beta; p.bnf_mask = a.bnf_mask; p.bnf_partial = a.bnf_partial; p.bnf_relu = a.bnf_relu;
    p.bnf_tile0 = a.bnf_tile0; p.relu = a.relu;
    p.N = a.N; p.H = a.H; p.W = a.W; p.C = a.C; p.Nout = a.Nout;
    p.M = a.N * a.H * a.W;
    p.P = a.W + 1; p.IP = (a.H + 1) * p.P;
    p.csteps = a.C / 64;
    const int bn = a.Nout % 128 == 0 ? 128 : 64;
    p.tiles_n = a.Nout / bn;
    p.tiles_total = (p.M + PP_BM - 1) / PP_BM * p.tiles_n;
    const int nwg = p.tiles_total < 256 ? p.tiles_total : 256;
    CNK_CHECK_ARG(nwg >= 8, "pp3x3: fewer than 8 tiles");
    p.nwg = nwg;
    p.npg = (pp_patch_rows(a.H, a.W) + 7) / 8;
    for (int t = 0; t < 9; ++t) p.toff[t] = a.dh[t] * p.P + a.dw[t];
    p.fd_hw = make_fastdiv(a.H * a.W);
    p.fd_w = make_fastdiv(a.W);
    p.fd_ip = make_fastdiv(p.IP);
    p.fd_p = make_fastdiv(p.P);
    const size_t lds = pp_lds_bytes(a.H, a.W, bn);
    ProfScope kscope(CNK_KF_HALO, s, 2.0 * p.M * (double)a.Nout * a.C * 9, a.algo_bytes);
    const int mode = (a.bias || a.addend || a.bnf_x || a.relu) ? PP_EPI_FULL : (a.stats ? PP_EPI_STATS : PP_EPI_PLAIN);
#define PP_LAUNCH(BNV, MODEV)                                                                                              \
    do {                                                                                                                   \
        static bool attr = false;                                                                                          \
        if (!attr) {                                                                                                       \
            (void)hipFuncSetAttribute((const void*)pp3x3_kernel<BNV, MODEV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr = true;                                                                                                   \
        }                                                                                                                  \
        hipLaunchKernelGGL((pp3x3_kernel<BNV, MODEV>), dim3(nwg), dim3(512), lds, s, p);                                   \
    } while (0)
    if (bn == 128) {
        if (mode == PP_EPI_FULL) PP_LAUNCH(128, PP_EPI_FULL);
        else if (mode == PP_EPI_STATS) PP_LAUNCH(128, PP_EPI_STATS);
        else PP_LAUNCH(128, PP_EPI_PLAIN);
    } else {
        if (mode == PP_EPI_FULL) PP_LAUNCH(64, PP_EPI_FULL);
        else if (mode == PP_EPI_STATS) PP_LAUNCH(64, PP_EPI_STATS);
        else PP_LAUNCH(64, PP_EPI_PLAIN);
    }
#undef PP_LAUNCH
    CNK_CHECK_LAUNCH("pp3x3");
    return 0;
}
