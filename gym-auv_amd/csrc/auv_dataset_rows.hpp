// The dataset rows of a running batch, assembled on the device: which envs
// are logged at this step, which obstacles each row lists, and the rows
// themselves (LiDAR ranges; risk per listed obstacle; speed_OS, heading_OS and
// five metadata columns per listed obstacle: utils/explore_dataset.py:8,40-53,
// utils/dataloader.py:55,59,108), compacted and ready for the text formatter
// (auv_format_device.hpp).  Optionally every row is also kept as the training
// sample the reference's loader would make of it (utils/dataloader.py:44-120:
// the scan as closeness, the label over at most 8 of the row's risks).
//
// Listing rule (what the reference's shipped rows show): an env's obstacle list
// is rebuilt at every multiple of `refresh_every` steps of its episode from the
// obstacles whose logged distance is below the sensor range, and in between
// only loses obstacles whose slot went away.  An env is logged when its episode
// is at a multiple of `every` steps and its list is not empty.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace auv {

enum RowsLabel { ROWS_LABEL_NONE = 0, ROWS_LABEL_MAX = 1, ROWS_LABEL_SUM = 2, ROWS_LABEL_MEAN = 3 };

struct RowsArgs {
    int V, O, R;
    int every, refresh_every;
    int first;  // 1: no lists yet, build them now whatever the step
    double sensor_range;
    const int32_t* t_step;    // [V]
    const double* meta;       // [V,O,5]; column 1 is the logged distance
    const double* obst_dist;  // [V,O]; +inf for an absent slot
    const double* risk;       // [V,O]
    const double* own;        // [V,2]
    const double* ranges;     // [V,R]
    const float* perception;  // [V,R]
    uint8_t* tracked;         // [V,O] in/out: the lists
    long long* row_of;        // [V+1]: 1 / 0 per env, then (after the scan) its row; the number of rows last
    int32_t* due_count;       // [1]: envs whose episode is at a multiple of `every`
    double* lidar_rows;       // [V,R] out
    double* risk_rows;        // [V,O] out, the listed obstacles first, id order kept
    double* meta_rows;        // [V,2+5*O] out
    int32_t* risk_len;        // [V] out: listed obstacles of the row
    int32_t* meta_len;        // [V] out: 2 + 5 * risk_len
    float* x_rows;            // [V,R] out or nullptr: the scan as the networks see it
    float* label;             // [V] out or nullptr
    int label_mode;           // RowsLabel
};

// how many envs are due at this step (the lists are only touched on steps where some env is)
__global__ __launch_bounds__(256) void k_rows_due(RowsArgs a) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool due = e < a.V && (a.t_step[e] % a.every) == 0;
    const unsigned long long m = __ballot(due);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(a.due_count, __popcll(m));
}

// a wave per env: its list after this step, and whether it gives a row
__global__ __launch_bounds__(256) void k_rows_select(RowsArgs a) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= a.V) return;
    const int t = a.t_step[e];
    const bool due = (t % a.every) == 0;
    const bool rebuild = a.first || (t % a.refresh_every) == 0;
    int listed = 0;
    for (int o = lane; o < a.O; o += 64) {
        const size_t eo = (size_t)e * a.O + o;
        const double d = a.obst_dist[eo];
        const bool present = d - d == 0.0;  // finite
        const bool fresh = present && a.meta[eo * 5 + 1] < a.sensor_range;
        const bool trk = rebuild ? fresh : (a.tracked[eo] != 0 && present);
        a.tracked[eo] = trk ? 1 : 0;
        listed += trk ? 1 : 0;
    }
    for (int d = 32; d >= 1; d >>= 1) listed += __shfl_xor(listed, d, 64);
    if (lane == 0) a.row_of[e] = (due && listed > 0) ? 1 : 0;
}

// a wave per env that gives a row: the row's three records, its listed obstacles compacted to the front
__global__ __launch_bounds__(256) void k_rows_write(RowsArgs a) {
    __shared__ double first8[4][8];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + w;
    if (e >= a.V) return;
    const long long r = a.row_of[e];
    if (a.row_of[e + 1] == r) return;  // no row (the whole wave leaves: nothing below is a workgroup barrier)
    const size_t MS = 2 + 5 * (size_t)a.O;
    for (int j = lane; j < a.R; j += 64) {
        a.lidar_rows[(size_t)r * a.R + j] = a.ranges[(size_t)e * a.R + j];
        if (a.x_rows) a.x_rows[(size_t)r * a.R + j] = a.perception[(size_t)e * a.R + j];
    }
    if (lane < 2) a.meta_rows[(size_t)r * MS + lane] = a.own[2 * (size_t)e + lane];
    int n = 0;  // listed obstacles so far (the same in every lane)
    for (int o0 = 0; o0 < a.O; o0 += 64) {
        const int o = o0 + lane;
        const bool trk = o < a.O && a.tracked[(size_t)e * a.O + o] != 0;
        const unsigned long long m = __ballot(trk);
        if (trk) {
            const int k = n + __popcll(m & ((1ull << lane) - 1ull));
            const size_t eo = (size_t)e * a.O + o;
            const double rk = a.risk[eo];
            a.risk_rows[(size_t)r * a.O + k] = rk;
            if (k < 8) first8[w][k] = rk;
            double* mr = a.meta_rows + (size_t)r * MS + 2 + 5 * (size_t)k;
            const double* ms = a.meta + eo * 5;
            mr[0] = ms[0];
            mr[1] = ms[1];
            mr[2] = ms[2];
            mr[3] = ms[3];
            mr[4] = ms[4];
        }
        n += __popcll(m);
    }
    if (lane == 0) {
        a.risk_len[r] = n;
        a.meta_len[r] = 2 + 5 * n;
    }
    if (a.label && a.label_mode != ROWS_LABEL_NONE) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            // the label of utils/dataloader.py:104-120 over the first 8 risks of the row.  Its sum is pandas'
            // over 8 columns with zeros for the absent ones, which numpy adds as pairs of pairs
            const int k = n < 8 ? n : 8;
            const double* f = first8[w];
            double v;
            if (a.label_mode == ROWS_LABEL_MAX) {
                v = f[0];
                for (int i = 1; i < k; ++i) v = (f[i] > v || f[i] != f[i]) ? f[i] : v;  // a NaN wins, as in numpy's max
            } else {
                double g[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] = i < k ? f[i] : 0.0;
                v = ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
                if (a.label_mode == ROWS_LABEL_MEAN) v = v / (double)k;
            }
            a.label[r] = (float)v;
        }
    }
}

}  // namespace auv
