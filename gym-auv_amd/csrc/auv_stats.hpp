// Episode statistics of a batch, reduced on the device in two launches.
//
// k_stats_partial   one workgroup per 16 to kStatsEnvs envs (stats_envs_per_block): per env the nearest LiDAR
//                   return (min over its rays) and the largest obstacle risk, and
//                   the sums of the flags and the reward; one record of
//                   kStatsFields doubles per workgroup into a workspace.
// k_stats_final     one workgroup folds the records, always in the same order
//                   (thread t takes records t, t + 256, ...; then a fixed tree), so
//                   the result does not depend on which workspace was used or on
//                   how the first launch was scheduled.
// The launch boundary between the two is what makes the records visible to the
// second kernel; nothing is exchanged between workgroups inside a launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace auv {

constexpr int kStatsFields = 8;   // env_steps, done, collisions, goals, truncations, reward, min range, max risk
constexpr int kStatsEnvs = 64;    // envs per workgroup of k_stats_partial, at most
constexpr int kStatsThreads = 256;

struct StatsArgs {
    int V, R, O;
    int envs_per_block;  // stats_envs_per_block(V)
    const uint8_t* done;
    const uint8_t* collision;
    const uint8_t* reached_goal;
    const uint8_t* truncated;
    const double* reward;   // [V]
    const double* ranges;   // [V,R]
    const double* risk;     // [V,O]
    double* records;        // [blocks, kStatsFields]
};

// Minimum / maximum that a NaN wins, as torch's amin / amax and numpy's min / max have it: a NaN in the ranges
// or the risks shows in the sums instead of being dropped by the comparison.
__device__ __forceinline__ double nan_min(double v, double o) { return (o < v || o != o) ? o : v; }
__device__ __forceinline__ double nan_max(double v, double o) { return (o > v || o != o) ? o : v; }
__device__ __forceinline__ double wave_min(double v) {
    for (int d = 32; d >= 1; d >>= 1) v = nan_min(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    for (int d = 32; d >= 1; d >>= 1) v = nan_max(v, __shfl_xor(v, d, 64));
    return v;
}

__global__ __launch_bounds__(kStatsThreads) void k_stats_partial(StatsArgs a) {
    __shared__ double part[kStatsThreads / 64][kStatsFields];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e0 = blockIdx.x * a.envs_per_block;
    double acc[kStatsFields];
#pragma unroll
    for (int f = 0; f < kStatsFields; ++f) acc[f] = 0.0;
    // Wave w takes the envs e0 + 4 w .. e0 + 4 w + 3, then 16 further on, up to the workgroup's share.  Four envs
    // at a time: four independent loads per lane are in flight (the kernel only reads; what it waits for is
    // memory).  Past the batch's end the last env is read again, without a branch, and left out of the sums.
    constexpr int kGroup = 4;
    for (int k = wave * kGroup; k < a.envs_per_block; k += (kStatsThreads / 64) * kGroup) {
        const int e = e0 + k;
        if (e >= a.V) break;  // wave-uniform
        const double* rg[kGroup];
        const double* rk[kGroup];
        double lo[kGroup], hi[kGroup];
#pragma unroll
        for (int g = 0; g < kGroup; ++g) {
            const int eg = e + g < a.V ? e + g : a.V - 1;
            rg[g] = a.ranges + (size_t)eg * a.R;
            rk[g] = a.risk + (size_t)eg * a.O;
            lo[g] = __builtin_inf();
            hi[g] = 0.0;  // risks are never negative
        }
        for (int r = lane; r < a.R; r += 64) {
            double t[kGroup];
#pragma unroll
            for (int g = 0; g < kGroup; ++g) t[g] = rg[g][r];
#pragma unroll
            for (int g = 0; g < kGroup; ++g) lo[g] = nan_min(lo[g], t[g]);
        }
        for (int o = lane; o < a.O; o += 64) {
            double t[kGroup];
#pragma unroll
            for (int g = 0; g < kGroup; ++g) t[g] = rk[g][o];
#pragma unroll
            for (int g = 0; g < kGroup; ++g) hi[g] = nan_max(hi[g], t[g]);
        }
#pragma unroll
        for (int g = 0; g < kGroup; ++g) {
            if (e + g >= a.V) break;  // wave-uniform
            acc[0] = acc[0] + 1.0;
            acc[1] = acc[1] + (double)a.done[e + g];
            acc[2] = acc[2] + (double)a.collision[e + g];
            acc[3] = acc[3] + (double)a.reached_goal[e + g];
            acc[4] = acc[4] + (double)a.truncated[e + g];
            acc[5] = acc[5] + a.reward[e + g];
            acc[6] = acc[6] + wave_min(lo[g]);
            acc[7] = acc[7] + wave_max(hi[g]);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int f = 0; f < kStatsFields; ++f) part[wave][f] = acc[f];
    }
    __syncthreads();
    if (threadIdx.x < kStatsFields) {
        double s = part[0][threadIdx.x];
        for (int w = 1; w < kStatsThreads / 64; ++w) s = s + part[w][threadIdx.x];
        a.records[(size_t)blockIdx.x * kStatsFields + threadIdx.x] = s;
    }
}

// out[f] = (accumulate ? out[f] : 0) + sum over records, in a fixed order
__global__ __launch_bounds__(kStatsThreads) void k_stats_final(const double* records, int n_records, double* out,
                                                              int accumulate) {
    __shared__ double fold[kStatsThreads][kStatsFields];
    double acc[kStatsFields];
#pragma unroll
    for (int f = 0; f < kStatsFields; ++f) acc[f] = 0.0;
    for (int i = threadIdx.x; i < n_records; i += kStatsThreads) {
#pragma unroll
        for (int f = 0; f < kStatsFields; ++f) acc[f] = acc[f] + records[(size_t)i * kStatsFields + f];
    }
#pragma unroll
    for (int f = 0; f < kStatsFields; ++f) fold[threadIdx.x][f] = acc[f];
    __syncthreads();
    for (int half = kStatsThreads / 2; half >= 1; half >>= 1) {
        if ((int)threadIdx.x < half) {
#pragma unroll
            for (int f = 0; f < kStatsFields; ++f)
                fold[threadIdx.x][f] = fold[threadIdx.x][f] + fold[threadIdx.x + half][f];
        }
        __syncthreads();
    }
    if (threadIdx.x < kStatsFields) {
        const double base = accumulate ? out[threadIdx.x] : 0.0;
        out[threadIdx.x] = base + fold[0][threadIdx.x];
    }
}

// Envs per workgroup: 16 (four waves x four envs in flight) while that keeps the grid under 4,096 workgroups, then
// 32, then 64: small batches still spread over the device, and a batch of 1,048,576 envs has 16,384 records.
inline int stats_envs_per_block(int V) { return V <= 16 * 4096 ? 16 : V <= 32 * 4096 ? 32 : kStatsEnvs; }
inline int stats_blocks(int V) { return (V + stats_envs_per_block(V) - 1) / stats_envs_per_block(V); }

}  // namespace auv
