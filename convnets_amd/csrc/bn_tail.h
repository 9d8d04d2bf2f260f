// Finalisation of a BN partial list INSIDE the kernel that writes it (round 5).
//
// A conv epilogue that emits per-tile BN partials -- forward: (n, mean, M2) of the stored outputs; data gradient: the two
// sums of the producing BN layer's backward (igemm_epilogue, gemm1x1_kernel) -- used to be followed by a dependent ~6-14 us
// launch (bn_finalize / bn_merge_final / bn_bwd_final) whose only job is to merge that list: 120 such launches per ResNet-50
// step, each with its ~4.4 us dependent-launch gap, all on the critical path.  Here the workgroups of the
// conv kernel finish the job themselves, two levels of "the last one to arrive merges":
//   level 1: the BNT_FAN (64) consecutive row tiles of a column tile form a group; every workgroup publishes its record with
//            write-through (sc1) stores, drains them, and draws a ticket of its group; the workgroup that draws the group's
//            last ticket merges the group's records IN SLICE ORDER and publishes one level-1 record;
//   level 2: ... and draws a ticket of the column tile; the last of those merges the level-1 records in group order and
//            finalises its channels (mean / invstd / running statistics / scale | shift, or A | B1 | B2 | dbeta | dgamma).
// The summation order is a function of the shapes only (which workgroup happens to be last does not matter): results are
// bit-reproducible run to run.  They are NOT bit-identical to the separate finalise kernels (another merge tree).
// Memory ordering as bn_merge_final_kernel (bn.hip): no release fence anywhere -- sc1 stores + s_waitcnt vmcnt(0) before the
// ticket, one acquire in the workgroup that goes on, sc1 loads of what other workgroups wrote (MI355X_MICROARCH.md: an
// agent-scope release is an L2 write-back; cdna_hip_programming.md 5, "Projection GEMM" item 2).
// Tickets live in caller memory (zero before the first use, every counter is reset by its last arriver): one buffer per
// stream on the host side, so launches that may be in flight together never share a counter.
#pragma once
#include "common.h"

__device__ __forceinline__ float bnt_ld(const float* p) {
    return __builtin_bit_cast(float, __hip_atomic_load((const unsigned int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void bnt_st(float* p, float v) {
    __hip_atomic_store((unsigned int*)p, __builtin_bit_cast(unsigned int, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void bnt_chan(float& na, float& ma, float& qa, float nb, float mb, float qb) {
    if (nb > 0.f) {
        float nt = na + nb, d = mb - ma, f = nb / nt;
        ma += d * f;
        qa += qb + d * d * (na * f);
        na = nt;
    }
}

// Merges records first .. first + count - 1 of `src` ([.][C][NV]) for the channels c0 .. c0 + ncols - 1: thread (c, r) of the
// NT / BN threads per channel takes records r, r + tpc, ...; the r == 0 threads then fold the others in, in order.  The
// result is valid in threads with tid < BN (r == 0).
template <int NT, int BN, int NV>
__device__ __forceinline__ void bnt_merge(const float* __restrict__ src, int first, int count, int C, int c0, int ncols, float* lds, int tid,
                                          float& a0, float& a1, float& a2) {
    constexpr int TPC = NT / BN;
    const int c = tid % BN, r = tid / BN;
    a0 = a1 = a2 = 0.f;
    const int cc = c < ncols ? c : ncols - 1;                      // lanes past the edge read a valid record and use none
    const float* base = src + ((size_t)first * C + c0 + cc) * NV;
    for (int s0 = r; s0 < count; s0 += 8 * TPC) {
        float v[8][NV];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + u * TPC, sc = s < count ? s : count - 1;
            const float* q = base + (size_t)sc * C * NV;
#pragma unroll
            for (int k = 0; k < NV; ++k) v[u][k] = bnt_ld(q + k);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool on = s0 + u * TPC < count;
            if constexpr (NV == 3) bnt_chan(a0, a1, a2, on ? v[u][0] : 0.f, v[u][1], v[u][2]);
            else { a0 += on ? v[u][0] : 0.f; a1 += on ? v[u][1] : 0.f; }
        }
    }
    if constexpr (TPC > 1) {
        __syncthreads();                                           // (the scratch may still hold the previous level)
        if (r > 0) {
            lds[((r - 1) * BN + c) * NV + 0] = a0;
            lds[((r - 1) * BN + c) * NV + 1] = a1;
            if constexpr (NV == 3) lds[((r - 1) * BN + c) * NV + 2] = a2;
        }
        __syncthreads();
        if (r == 0) {
#pragma unroll
            for (int q = 1; q < TPC; ++q) {
                const float* o = lds + ((q - 1) * BN + c) * NV;
                if constexpr (NV == 3) bnt_chan(a0, a1, a2, o[0], o[1], o[2]);
                else { a0 += o[0]; a1 += o[1]; }
            }
        }
    }
}

// One ticket for the whole workgroup: every wave has drained its stores; thread 0 draws; true in ALL threads of the workgroup
// that drew the last of `expect` tickets (it has reset the counter and run the acquire).  `flag`: one int of LDS.
__device__ __forceinline__ bool bnt_last(unsigned int* counter, int expect, int* flag, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (unsigned int)expect - 1u;
        if (last) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // for the buffer's next user
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *flag = last;
    }
    __syncthreads();
    return *flag != 0;
}

// Called by ALL NT threads of a workgroup after its threads tid < ncols have stored (bnt_st) record `slice` of `list` for the
// channels c0 .. c0 + ncols - 1 (column tile cg of ncg).  lds: 16 + NT * NV * 4 bytes that nobody else uses any more.
template <int NT, int BN, int NV>
__device__ __forceinline__ void bn_tail_run(const BnTailParams& t, const float* __restrict__ list, int slice, int c0, int ncols, int cg,
                                            int ncg, float* lds, int tid) {
    int* flag = (int*)lds;
    float* scr = lds + 4;
    const int ngroups = (t.slices + BNT_FAN - 1) / BNT_FAN;
    const int g = slice / BNT_FAN;
    const int gsize = min(BNT_FAN, t.slices - g * BNT_FAN);
    if (!bnt_last(t.tickets + (size_t)g * ncg + cg, gsize, flag, tid)) return;
    float a0, a1, a2;
    bnt_merge<NT, BN, NV>(list, g * BNT_FAN, gsize, t.C, c0, ncols, scr, tid, a0, a1, a2);
    const int ch = c0 + tid;
    const bool mine = tid < BN && tid < ncols;
    if (ngroups > 1) {
        if (mine) {
            float* o = t.l2 + ((size_t)g * t.C + ch) * NV;
            bnt_st(o, a0);
            bnt_st(o + 1, a1);
            if constexpr (NV == 3) bnt_st(o + 2, a2);
        }
        if (!bnt_last(t.tickets + (size_t)ngroups * ncg + cg, ngroups, flag, tid)) return;
        bnt_merge<NT, BN, NV>(t.l2, 0, ngroups, t.C, c0, ncols, scr, tid, a0, a1, a2);
    }
    if (!mine) return;
    if constexpr (NV == 3) {
        // as bn_finalize_kernel (bn.hip)
        const float var = a2 / (float)t.M;
        float invstd = rsqrtf(var + t.eps);
        invstd = invstd * (1.5f - 0.5f * (var + t.eps) * invstd * invstd);
        t.save_mean[ch] = a1;
        t.save_invstd[ch] = invstd;
        if (t.running_mean) {
            const float unb = t.M > 1 ? a2 / (float)(t.M - 1) : var;
            t.running_mean[ch] = (1.f - t.momentum) * t.running_mean[ch] + t.momentum * a1;
            t.running_var[ch] = (1.f - t.momentum) * t.running_var[ch] + t.momentum * unb;
        }
        const float sc = t.gamma[ch] * invstd;
        t.out[ch] = sc;
        t.out[t.C + ch] = t.beta[ch] - a1 * sc;
    } else {
        // as bn_bwd_final_kernel; the raw sums ride along: whoever confirms the fused path adds them to dbeta / dgamma
        t.out[ch] = t.gamma[ch] * t.save_invstd[ch];
        t.out[t.C + ch] = a0 / (float)t.M;
        t.out[2 * t.C + ch] = a1 / (float)t.M;
        t.out[3 * t.C + ch] = a0;
        t.out[4 * t.C + ch] = a1;
    }
}

template <int NT, int BN>
__device__ __forceinline__ void bn_tail(const BnTailParams& t, const float* __restrict__ list, int slice, int c0, int ncols, int cg, int ncg,
                                        float* lds, int tid) {
    if (t.kind == 1) bn_tail_run<NT, BN, 3>(t, list, slice, c0, ncols, cg, ncg, lds, tid);
    else bn_tail_run<NT, BN, 2>(t, list, slice, c0, ncols, cg, ncg, lds, tid);
}
