// The slab sum of one weight gradient, done by the workgroups of the NEXT weight-gradient launch (round 5).
//
// Every split-K weight-gradient kernel leaves fp32 slabs [split][Cout][taps][Cin] that a second, tiny launch summed into
// dw (wgrad_reduce4_kernel): 52 such launches per ResNet-50 step on the weight-gradient stream, 22-37 us each there.  The
// weight-gradient stream runs one such kernel after the other, so the sum of layer k rides in the prologue of layer k + 1's
// kernel: every workgroup of that launch takes a grid-stride share of it before it starts on its own tile
// (cdna_hip_programming.md 5, "combine in the NEXT kernel's prologue (the launch-boundary reduce)"); only a chain's last sum
// needs the stand-alone kernel.  Same element-wise summation order whoever runs it (lane q of a G-lane group takes slabs
// q, q + G, ... with eight loads in flight, xor-shuffle tree): bit-reproducible.
// Measured (profiles/r05_ab_wgrad_chain.txt): 0.05 ms of the step.  A timing probe with the sums skipped altogether had
// promised 0.4-0.55 ms (profiles/r05_probe_skip_wgrad_reduce.txt): the launches were the small part of that, the slabs' own
// traffic (2.2 GB per step at full split depth, written and read back) is the rest and stays.
#pragma once
#include "common.h"

// == cnk_wgrad_pending (include/cnk.h)
struct WgradPending {
    const float* slab;         // [splits][Cout][ntaps][Cin], slab_stride floats apart; nullptr: nothing pending
    float* dw;                 // master layout [Cout][Cin][ntaps]
    uint64_t slab_stride;
    int32_t splits, Cout, Cin, ntaps, accumulate;
    int32_t G;                 // lanes per 4-channel group (1, 2, 4, 8, 16): chosen once, by wgrad_pending_fill
};

static_assert(sizeof(WgradPending) == sizeof(cnk_wgrad_pending) && offsetof(WgradPending, G) == offsetof(cnk_wgrad_pending, lanes),
              "WgradPending must mirror cnk_wgrad_pending");

static inline void wgrad_pending_fill(WgradPending& q, const float* slab, float* dw, int splits, size_t slab_stride, int Cout, int Cin,
                                      int ntaps, int accumulate) {
    q.slab = slab; q.dw = dw; q.slab_stride = slab_stride; q.splits = splits; q.Cout = Cout; q.Cin = Cin; q.ntaps = ntaps;
    q.accumulate = accumulate;
    // lanes per group: enough threads for ~64 k lanes of work, never more lanes than slabs (as launch_wgrad_reduce)
    const size_t groups = (size_t)Cout * ntaps * (Cin / 4);
    int G = 1;
    while (G < 16 && groups * G < 65536 && 2 * G <= splits) G <<= 1;
    q.G = G;
}

// Called by ALL threads of ALL workgroups of a launch (NT threads each): a grid-stride share of the pending sum.  Cin % 4 == 0.
template <int NT>
__device__ __forceinline__ void wgrad_pending_sum(const WgradPending& q) {
    if (!q.slab) return;
    const int G = q.G;
    const int c4 = q.Cin >> 2;
    const size_t total = (size_t)q.Cout * q.ntaps * c4;
    const size_t total_r = (total + 63) / 64 * 64;                      // whole waves stay converged for the shuffles
    const int lane_q = threadIdx.x % G;
    const size_t nthreads = (size_t)gridDim.x * NT;
    for (size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) / G; i < total_r; i += nthreads / G) {
        const bool live = i < total;
        const size_t ii = live ? i : 0;
        const int cq = (int)(ii % c4);
        const size_t r = ii / c4;
        const int t = (int)(r % q.ntaps), co = (int)(r / q.ntaps);
        const float* sp = q.slab + ii * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        int s = lane_q;
        for (; s + 7 * G < q.splits; s += 8 * G) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(sp + (size_t)(s + u * G) * q.slab_stride);
            a = a + (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
        }
        for (; s < q.splits; s += G) a = a + *(const f32x4*)(sp + (size_t)s * q.slab_stride);
        for (int m = 1; m < G; m <<= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += __shfl_xor(a[j], m, 64);
        if (lane_q == 0 && live) {
            float* o = q.dw + ((size_t)co * q.Cin + cq * 4) * q.ntaps + t;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[(size_t)j * q.ntaps] = q.accumulate ? o[(size_t)j * q.ntaps] + a[j] : a[j];
        }
    }
}
