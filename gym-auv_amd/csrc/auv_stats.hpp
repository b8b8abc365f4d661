// Episode statistics of a batch, reduced on the device in two launches.
//
// k_stats_partial   one workgroup per kStatsEnvs envs: per env the nearest LiDAR
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
constexpr int kStatsEnvs = 64;    // envs per workgroup of k_stats_partial
constexpr int kStatsThreads = 256;

struct StatsArgs {
    int V, R, O;
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
    const int e0 = blockIdx.x * kStatsEnvs;
    double acc[kStatsFields];
#pragma unroll
    for (int f = 0; f < kStatsFields; ++f) acc[f] = 0.0;
    // wave w takes envs e0 + w, e0 + w + 4, ...: every lane runs the same trip count
    for (int k = wave; k < kStatsEnvs; k += kStatsThreads / 64) {
        const int e = e0 + k;
        if (e >= a.V) break;  // wave-uniform
        double lo = __builtin_inf(), hi = 0.0;  // risks are never negative
        const double* rg = a.ranges + (size_t)e * a.R;
        for (int r = lane; r < a.R; r += 64) {
            lo = nan_min(lo, rg[r]);
        }
        const double* rk = a.risk + (size_t)e * a.O;
        for (int o = lane; o < a.O; o += 64) {
            hi = nan_max(hi, rk[o]);
        }
        lo = wave_min(lo);
        hi = wave_max(hi);
        acc[0] = acc[0] + 1.0;
        acc[1] = acc[1] + (double)a.done[e];
        acc[2] = acc[2] + (double)a.collision[e];
        acc[3] = acc[3] + (double)a.reached_goal[e];
        acc[4] = acc[4] + (double)a.truncated[e];
        acc[5] = acc[5] + a.reward[e];
        acc[6] = acc[6] + lo;
        acc[7] = acc[7] + hi;
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

inline int stats_blocks(int V) { return (V + kStatsEnvs - 1) / kStatsEnvs; }

}  // namespace auv
