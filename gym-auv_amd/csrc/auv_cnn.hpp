// Forward passes of the LiDAR-CNN consumers of the scan, fp32, for gfx950.
//
// The three reference networks (models/lidar_cnn_shallow.py:23-52,
// models/lidar_cnn_deep.py:19-96 and :110-157; their RL-side prefixes in
// radarCNN.py:10-182) are tiny (59 / 178 / 3,723 parameters) and read 720 B
// per scan, so the pass is HBM-bound: one workgroup takes a tile of scans
// straight from the simulation's perception buffer, keeps every intermediate
// activation in LDS, and writes only the 1 / 8 / 12 outputs per scan.
// 16 lanes cooperate on one scan; weights are read through the scalar/L1 path.
// Products and sums are fused (fmaf) although the translation unit is built
// without contraction: the networks are compared with the reference's outputs
// by tolerance, not bit for bit, and a fused step only tightens them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace auv {

enum CnnKind { CNN_SHALLOW = 0, CNN_2DEEP = 1, CNN_DEEP = 2 };
enum CnnHead { HEAD_FEATURES = 0, HEAD_RISK = 1 };

constexpr int kCnnLen = 180;         // scan width the shipped weights were trained on
constexpr int kCnnLanes = 16;        // lanes per scan
constexpr int kCnnScans = 16;        // scans per 256-thread workgroup
constexpr int kCnnBuf = 184;         // floats per activation buffer (>= 180, 16-B multiple)

__host__ __device__ constexpr int cnn_n_weights(int kind) {
    return kind == CNN_SHALLOW ? 59 : kind == CNN_2DEEP ? 178 : 3723;
}
__host__ __device__ inline int cnn_out_dim(int kind, int head) {
    if (head == HEAD_RISK) return 1;
    return kind == CNN_DEEP ? 8 : 12;
}

struct CnnArgs {
    const float* scans;  // [B,1,180]
    const float* w;      // packed state-dict order
    float* out;          // [B,out_dim]
    long long B;
    int kind, head;
};

// A layer works on `ns` scans at once: scan s reads in + s*in_stride and writes
// out + s*out_stride; the ns*outputs-per-scan output elements are dealt to
// `n_workers` threads (this one is `worker`).  Every element is accumulated by
// one thread in a fixed order, so results do not depend on the dealing.
//
// out[co][lo] = act(b[co] + sum_ci sum_k w[co][ci][k] * in[ci][(pos*stride + k - pad) mod L]),
// optionally followed by MaxPool1d(2, 2, ceil_mode=True) fused into the store
template <bool RELU, bool POOL>
__device__ __forceinline__ int conv_layer(const float* in, int in_stride, float* out, int out_stride, int ns,
                                          const float* w, const float* b, int Cin, int Cout, int L, int K, int stride,
                                          int pad, int worker, int n_workers) {
    const int Lc = (L + 2 * pad - K) / stride + 1;
    const int Lo = POOL ? (Lc + 1) / 2 : Lc;
    const int per_scan = Cout * Lo;
    for (int idx = worker; idx < ns * per_scan; idx += n_workers) {
        const int s = idx / per_scan;
        const int el = idx - s * per_scan;
        const int co = el / Lo;
        const int lo = el - co * Lo;
        float best = 0.f;
        const int n_pos = (POOL && 2 * lo + 1 < Lc) ? 2 : 1;
        for (int j = 0; j < n_pos; ++j) {
            const int pos = POOL ? 2 * lo + j : lo;
            float acc = b[co];
            for (int ci = 0; ci < Cin; ++ci) {
                const float* wr = w + ((size_t)co * Cin + ci) * K;
                const float* xr = in + (size_t)s * in_stride + ci * L;
                int p = pos * stride - pad;
                p = p < 0 ? p + L : p;
#pragma unroll 9
                for (int k = 0; k < K; ++k) {
                    acc = __builtin_fmaf(wr[k], xr[p], acc);
                    p = (p + 1 == L) ? 0 : p + 1;
                }
            }
            if (RELU) acc = acc > 0.f ? acc : 0.f;
            best = (j == 0 || acc > best) ? acc : best;
        }
        out[(size_t)s * out_stride + el] = best;
    }
    return Lo;
}

// Between layers: the workers of one scan group either sit in one wave (a
// wave-level LDS fence is enough) or are the whole workgroup.
template <bool BLOCK>
__device__ __forceinline__ void layer_sync() {
    if (BLOCK) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// The first layer of the two small networks: Conv1d(1 -> Cout, kernel 45, stride
// 15, circular padding 15) over the 180-ray scan, i.e. 12 outputs per channel,
// each over three consecutive 15-ray sectors.  Summation order (the same in
// every kernel that evaluates it): one partial sum per sector, taps in order
// from 0, then ((bias + s0) + s1) + s2.  Sector sums are independent work
// items (stage A, into `tmp`: Cout*36 floats per scan), combined after a sync
// (stage B, into `out`); a sector never wraps, 180 being 12 sectors.  With
// `rows` the layer is the network's last: its Cout*12 values per scan also go
// to rows + s*Cout*12 (global), from the thread that formed them.
template <bool RELU, bool BLOCK>
__device__ __forceinline__ void conv_sectors(const float* x, int x_stride, float* out, float* tmp, int a_stride, int ns,
                                             const float* w, const float* b, int Cout, int worker, int n_workers,
                                             float* rows = nullptr) {
    if (!BLOCK) {
        // few lanes per scan: each keeps the three sector sums of its outputs in
        // registers (same order of operations, no staging)
        for (int idx = worker; idx < ns * Cout * 12; idx += n_workers) {
            const int s = idx / (Cout * 12);
            const int col = idx - s * (Cout * 12);
            const int co = col / 12;
            const int lo = col - co * 12;
            float part[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                int sec = lo - 1 + j;
                sec = sec < 0 ? sec + 12 : (sec >= 12 ? sec - 12 : sec);
                const float* wr = w + co * 45 + j * 15;
                const float* xr = x + (size_t)s * x_stride + sec * 15;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 15; ++k) acc = __builtin_fmaf(wr[k], xr[k], acc);
                part[j] = acc;
            }
            float acc = ((b[co] + part[0]) + part[1]) + part[2];
            if (RELU) acc = acc > 0.f ? acc : 0.f;
            out[(size_t)s * a_stride + col] = acc;
            if (rows) rows[(size_t)s * (Cout * 12) + col] = acc;
        }
        return;
    }
    const int per_scan = Cout * 36;
    for (int idx = worker; idx < ns * per_scan; idx += n_workers) {
        const int s = idx / per_scan;
        const int el = idx - s * per_scan;  // (co * 12 + lo) * 3 + part
        const int col = el / 3;
        const int part = el - col * 3;
        const int co = col / 12;
        const int lo = col - co * 12;
        int sec = lo - 1 + part;  // sector of the scan under this third of the kernel
        sec = sec < 0 ? sec + 12 : (sec >= 12 ? sec - 12 : sec);
        const float* wr = w + co * 45 + part * 15;
        const float* xr = x + (size_t)s * x_stride + sec * 15;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 15; ++k) acc = __builtin_fmaf(wr[k], xr[k], acc);
        tmp[(size_t)s * a_stride + el] = acc;
    }
    layer_sync<BLOCK>();
    for (int idx = worker; idx < ns * Cout * 12; idx += n_workers) {
        const int s = idx / (Cout * 12);
        const int col = idx - s * (Cout * 12);
        const float* t = tmp + (size_t)s * a_stride + col * 3;
        float acc = ((b[col / 12] + t[0]) + t[1]) + t[2];
        if (RELU) acc = acc > 0.f ? acc : 0.f;
        out[(size_t)s * a_stride + col] = acc;
        if (rows) rows[(size_t)s * (Cout * 12) + col] = acc;
    }
}

template <bool RELU>
__device__ __forceinline__ void linear_layer(const float* in, int in_stride, float* out, int out_stride, int ns,
                                             const float* w, const float* b, int n_in, int n_out, int worker,
                                             int n_workers) {
    for (int idx = worker; idx < ns * n_out; idx += n_workers) {
        const int s = idx / n_out;
        const int o = idx - s * n_out;
        const float* x = in + (size_t)s * in_stride;
        float acc = 0.f;
#pragma unroll 4
        for (int i = 0; i < n_in; ++i) acc = __builtin_fmaf(x[i], w[(size_t)o * n_in + i], acc);
        acc += b[o];
        if (RELU) acc = acc > 0.f ? acc : 0.f;
        out[(size_t)s * out_stride + o] = acc;
    }
}

// LDS floats of the two activation buffers p and q of one scan (q directly behind p)
__host__ __device__ inline int cnn_p_floats(int kind) { return kind == CNN_DEEP ? kCnnBuf : 40; }
__host__ __device__ inline int cnn_q_floats(int kind) { return kind == CNN_DEEP ? kCnnBuf : 112; }  // 3 * 36 sector sums
__host__ __device__ inline int cnn_act_floats(int kind) { return cnn_p_floats(kind) + cnn_q_floats(kind); }
// the two small networks can run inside the sensing kernel, weights staged in LDS
__host__ __device__ inline bool cnn_fits_sense_kernel(int kind) { return kind == CNN_SHALLOW || kind == CNN_2DEEP; }

// Forward pass of `ns` scans by `n_workers` cooperating threads.  Scan s has
// its 180 closeness values at x + s*x_stride and two activation buffers of
// cnn_p_floats / cnn_q_floats(kind) floats at p + s*a_stride and q + s*a_stride (all LDS);
// its row of the result goes to out + s*out_dim (global).  BLOCK: the workers
// are the whole workgroup (all of its threads must call), else lanes of one
// wave.  WITH_DEEP = false leaves the largest network out of the instantiation
// (kind must not be CNN_DEEP then).
template <bool WITH_DEEP, bool BLOCK>
__device__ __forceinline__ void cnn_forward(const float* x, int x_stride, float* p, float* q, int a_stride, int ns,
                                            const float* w, float* out, int kind, int head, int worker,
                                            int n_workers) {
    const float* res = p;
    int n_res = 0;
#define AUV_CONV(RELU, POOL, IN, IN_STRIDE, OUT, ...) \
    conv_layer<RELU, POOL>(IN, IN_STRIDE, OUT, a_stride, ns, __VA_ARGS__, worker, n_workers)
#define AUV_LIN(RELU, IN, OUT, ...) linear_layer<RELU>(IN, a_stride, OUT, a_stride, ns, __VA_ARGS__, worker, n_workers)
    if (kind == CNN_SHALLOW) {
        // w: conv.weight[45] conv.bias[1] linear.weight[12] linear.bias[1]
        if (head == HEAD_FEATURES) {
            // the one layer is the result: stored as it is formed, no copy behind another sync
            conv_sectors<false, BLOCK>(x, x_stride, p, q, a_stride, ns, w, w + 45, 1, worker, n_workers, out);
            return;
        } else {
            conv_sectors<true, BLOCK>(x, x_stride, p, q, a_stride, ns, w, w + 45, 1, worker, n_workers);
            layer_sync<BLOCK>();
            AUV_LIN(true, p, q, w + 46, w + 58, 12, 1);
            res = q;
            n_res = 1;
        }
    } else if (!WITH_DEEP || kind == CNN_2DEEP) {
        // w: c1.w[3*45] c1.b[3] c2.w[2*3*3] c2.b[2] c3.w[1*2*3] c3.b[1] lin.w[12] lin.b[1]
        conv_sectors<true, BLOCK>(x, x_stride, p, q, a_stride, ns, w, w + 135, 3, worker, n_workers);
        layer_sync<BLOCK>();
        AUV_CONV(true, false, p, a_stride, q, w + 138, w + 156, 3, 2, 12, 3, 1, 1);
        layer_sync<BLOCK>();
        AUV_CONV(false, false, q, a_stride, p, w + 158, w + 164, 2, 1, 12, 3, 1, 1);
        res = p;
        n_res = 12;
        if (head == HEAD_RISK) {
            layer_sync<BLOCK>();
            AUV_LIN(true, p, q, w + 165, w + 177, 12, 1);
            res = q;
            n_res = 1;
        }
    } else if constexpr (WITH_DEEP) {
        // w: c1.w[2*1*9] b[2] c2.w[4*2*9] b[4] c3.w[4*4*9] b[4] c4.w[6*4*9] b[6]
        //    l1.w[40*72] b[40] l2.w[8*40] b[8] l3.w[8] b[1]
        int L = AUV_CONV(true, true, x, x_stride, p, w, w + 18, 1, 2, kCnnLen, 9, 1, 4);  // 2 x 90
        layer_sync<BLOCK>();
        L = AUV_CONV(true, true, p, a_stride, q, w + 20, w + 92, 2, 4, L, 9, 1, 4);       // 4 x 45
        layer_sync<BLOCK>();
        L = AUV_CONV(true, true, q, a_stride, p, w + 96, w + 240, 4, 4, L, 9, 1, 4);      // 4 x 23
        layer_sync<BLOCK>();
        L = AUV_CONV(true, true, p, a_stride, q, w + 244, w + 460, 4, 6, L, 9, 1, 4);     // 6 x 12
        layer_sync<BLOCK>();
        AUV_LIN(true, q, p, w + 466, w + 3346, 72, 40);
        layer_sync<BLOCK>();
        AUV_LIN(false, p, q, w + 3386, w + 3706, 40, 8);
        res = q;
        n_res = 8;
        if (head == HEAD_RISK) {
            layer_sync<BLOCK>();
            // ReLU on the 8 features, then Linear(8 -> 1) + ReLU
            for (int i = worker; i < ns * 8; i += n_workers) {
                const int s = i >> 3, o = i & 7;
                const float v = q[(size_t)s * a_stride + o];
                p[(size_t)s * a_stride + o] = v > 0.f ? v : 0.f;
            }
            layer_sync<BLOCK>();
            AUV_LIN(true, p, q + 8, w + 3714, w + 3722, 8, 1);
            res = q + 8;
            n_res = 1;
        }
    }
#undef AUV_CONV
#undef AUV_LIN
    layer_sync<BLOCK>();
    for (int i = worker; i < ns * n_res; i += n_workers) {
        const int s = i / n_res;
        const int o = i - s * n_res;
        out[(size_t)s * n_res + o] = res[(size_t)s * a_stride + o];
    }
}

// dynamic LDS of one k_cnn workgroup: the scans, then two activation buffers per scan
__host__ __device__ inline size_t cnn_lds_bytes(int kind) {
    return (size_t)kCnnScans * (kCnnBuf + 2 * cnn_p_floats(kind)) * sizeof(float);  // no sector staging here
}

__global__ __launch_bounds__(256) void k_cnn(CnnArgs a) {
    extern __shared__ __align__(16) float cnn_lds[];
    const int na = 2 * cnn_p_floats(a.kind);
    float* xs = cnn_lds;                       // [kCnnScans][kCnnBuf], rows 16-B aligned
    float* act = cnn_lds + kCnnScans * kCnnBuf;  // [kCnnScans][na]: p then q
    const int tid = threadIdx.x;
    const long long s0 = (long long)blockIdx.x * kCnnScans;
    const int ns = (int)((a.B - s0) < kCnnScans ? (a.B - s0) : kCnnScans);
    // coalesced 16-B loads of the tile's scans (180 floats = 45 float4 per scan; the
    // tile starts at a multiple of 16 * 720 B)
    const float4* src = reinterpret_cast<const float4*>(a.scans + s0 * kCnnLen);
    for (int i = tid; i < ns * (kCnnLen / 4); i += blockDim.x) {
        const int s = i / (kCnnLen / 4);
        const int c = i - s * (kCnnLen / 4);
        *reinterpret_cast<float4*>(xs + s * kCnnBuf + 4 * c) = src[i];
    }
    __syncthreads();
    // 16 lanes per scan, one scan per lane group
    const int s = tid / kCnnLanes;
    const int lane = tid - s * kCnnLanes;
    if (s >= ns) return;
    cnn_forward<true, false>(xs + s * kCnnBuf, 0, act + (size_t)s * na, act + (size_t)s * na + cnn_p_floats(a.kind), 0, 1,
                             a.w,
                             a.out + (s0 + s) * cnn_out_dim(a.kind, a.head), a.kind, a.head, lane, kCnnLanes);
}

// ---- the deep network, its own kernel -------------------------------------
// models/lidar_cnn_deep.py:19-96 with channels [2,4,4,6], kernel 9, circular
// padding 4, ReLU + MaxPool1d(2, ceil) after every convolution, then
// Linear(72,40)+ReLU, Linear(40,8) [features], ReLU + Linear(8,1)+ReLU [risk].
// 16 lanes per scan as in k_cnn, but: every shape is a compile-time constant;
// activation rows carry their circular halo (4 left, 5 right), so a tap is one
// LDS read at a fixed offset; a lane owns pooled position lo, reads its 10-wide
// input window once and runs all output channels over it with the weights in
// scalar registers (the channel loop is wave-uniform); the two linear layers
// read transposed weights (lane = output: consecutive floats per wave) made once
// when the network is created, through the vector cache.
// Accumulation order is that of conv_layer / linear_layer; products and sums
// are fused (fmaf), which only tightens the result (tolerance-compared).
constexpr int kDeepHaloL = 4, kDeepHaloR = 5;
constexpr int kDeepRowA = 216;  // floats per scan of buffer A: max(180+9, 4*(45+9), 72)
constexpr int kDeepRowB = 200;  // buffer B: max(2*(90+9), 4*(23+9), 40)
constexpr int kDeepL1T = 72 * 40, kDeepL2T = 40 * 8;

__host__ __device__ inline size_t cnn_deep_lds_bytes() { return (size_t)kCnnScans * (kDeepRowA + kDeepRowB) * sizeof(float); }
// floats of the deep network's device copy: the packed parameters, then linear_1.0.weight and linear_1.2.weight
// transposed ([72][40] and [40][8]: lane = output, so a wave reads consecutive floats), made once at creation
constexpr int kDeepWeightsDev = 3723 + kDeepL1T + kDeepL2T;

// store v as logical element lo of a row of length Lo that carries halos
__device__ __forceinline__ void put_with_halo(float* row, int Lo, int lo, float v) {
    row[kDeepHaloL + lo] = v;
    if (lo >= Lo - kDeepHaloL) row[lo - (Lo - kDeepHaloL)] = v;
    if (lo < kDeepHaloR) row[kDeepHaloL + Lo + lo] = v;
}

typedef float cnn_float2 __attribute__((ext_vector_type(2)));

// The two conv positions under a pooled cell share every weight: they are the
// two halves of one packed fused multiply-add.  Tap k needs (x[k], x[k+1]):
// even taps take the pairs (x0,x1) (x2,x3) ..., odd taps the pairs (x1,x2) ...
template <int CIN, int COUT, int L, bool LAST>
__device__ __forceinline__ void deep_conv(const float* in, float* out, const float* w, const float* b, int lane) {
    constexpr int Lo = (L + 1) / 2;
    constexpr int LsI = L + kDeepHaloL + kDeepHaloR;
    constexpr int LsO = LAST ? Lo : Lo + kDeepHaloL + kDeepHaloR;
    for (int lo = lane; lo < Lo; lo += kCnnLanes) {
        // inputs 2lo-4 .. 2lo+5 of every channel, as even and odd pairs; the odd pairs are formed from the
        // registers of the even ones (the CU's one LDS pipe is the scarcer unit here)
        cnn_float2 xe[CIN][5], xo[CIN][4];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const float* r = in + ci * LsI + 2 * lo;
#pragma unroll
            for (int j = 0; j < 5; ++j) xe[ci][j] = cnn_float2{r[2 * j], r[2 * j + 1]};
#pragma unroll
            for (int j = 0; j < 4; ++j) xo[ci][j] = cnn_float2{xe[ci][j].y, xe[ci][j + 1].x};
        }
        const bool two = 2 * lo + 1 < L;  // the last pooled cell of an odd row covers one position
#pragma nounroll
        for (int co = 0; co < COUT; ++co) {  // wave-uniform: weights come through the scalar unit
            cnn_float2 acc = cnn_float2{b[co], b[co]};
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float wv = w[(co * CIN + ci) * 9 + k];
                    acc = __builtin_elementwise_fma(cnn_float2{wv, wv}, (k & 1) ? xo[ci][k >> 1] : xe[ci][k >> 1], acc);
                }
            // ReLU then the larger of the pair = the largest of the two sums and zero
            const float a1 = ((L & 1) && !two) ? acc.x : acc.y;
            const float v = __builtin_fmaxf(__builtin_fmaxf(acc.x, a1), 0.f);
            out[co * LsO + (LAST ? 0 : kDeepHaloL) + lo] = v;
        }
    }
    if (!LAST) {
        // the rows' circular halos, copied once the rows are complete: 4 cells before, 5 behind, per channel
        layer_sync<false>();
        for (int i = lane; i < COUT * (kDeepHaloL + kDeepHaloR); i += kCnnLanes) {
            const int co = i / (kDeepHaloL + kDeepHaloR), h = i - co * (kDeepHaloL + kDeepHaloR);
            float* row = out + co * LsO;
            if (h < kDeepHaloL) row[h] = row[Lo + h];                       // cell Lo - 4 + h
            else row[Lo + h] = row[h];                                      // cell h - 4 behind the end
        }
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) void k_cnn_deep(CnnArgs a) {
    extern __shared__ __align__(16) float cnn_lds[];
    float* bufA = cnn_lds;                                   // [kCnnScans][kDeepRowA]
    float* bufB = bufA + kCnnScans * kDeepRowA;               // [kCnnScans][kDeepRowB]
    const int tid = threadIdx.x;
    const long long s0 = (long long)blockIdx.x * kCnnScans;
    const int ns = (int)((a.B - s0) < kCnnScans ? (a.B - s0) : kCnnScans);
    const float* w = a.w;
    // packed order: c1.w[18] b[2] c2.w[72] b[4] c3.w[144] b[4] c4.w[216] b[6] l1.w[2880] b[40] l2.w[320] b[8] l3.w[8] b[1],
    // then the two transposed linear weights (kDeepWeightsDev)
    const float* w1t = w + 3723;            // [72][40]
    const float* w2t = w1t + kDeepL1T;      // [40][8]
    // the tile's scans, each into a row with its circular halo, in 16-byte pieces (45 per scan; rows and the
    // tile's first scan start at multiples of 16 B): the last piece is also the left halo, the first one and the
    // first float of the second the right halo
    static_assert(kDeepHaloL == 4 && kDeepHaloR == 5 && kDeepRowA % 4 == 0 && kCnnLen % 4 == 0, "halo pieces");
    const float4* src = reinterpret_cast<const float4*>(a.scans + s0 * kCnnLen);
    for (int i = tid; i < ns * (kCnnLen / 4); i += 256) {
        const int sc = i / (kCnnLen / 4), c = i - sc * (kCnnLen / 4);
        const float4 v = src[i];
        float* row = bufA + sc * kDeepRowA;
        *reinterpret_cast<float4*>(row + kDeepHaloL + 4 * c) = v;
        if (c == kCnnLen / 4 - 1) *reinterpret_cast<float4*>(row) = v;
        if (c == 0) *reinterpret_cast<float4*>(row + kDeepHaloL + kCnnLen) = v;
        if (c == 1) row[kDeepHaloL + kCnnLen + 4] = v.x;
    }
    __syncthreads();
    const int s = tid / kCnnLanes;
    const int lane = tid - s * kCnnLanes;
    if (s >= ns) return;
    float* A = bufA + s * kDeepRowA;
    float* B = bufB + s * kDeepRowB;
    deep_conv<1, 2, 180, false>(A, B, w, w + 18, lane);         // 2 x 90
    layer_sync<false>();
    deep_conv<2, 4, 90, false>(B, A, w + 20, w + 92, lane);     // 4 x 45
    layer_sync<false>();
    deep_conv<4, 4, 45, false>(A, B, w + 96, w + 240, lane);    // 4 x 23
    layer_sync<false>();
    deep_conv<4, 6, 23, true>(B, A, w + 244, w + 460, lane);    // 6 x 12, plain rows = the 72 inputs of linear 1
    layer_sync<false>();
    {
        // Linear(72 -> 40) + ReLU: a lane owns outputs lane, lane + 16, lane + 32; the 72 inputs come from LDS
        // four at a time and serve the three; each output is summed over k in order
        const float4* A4 = reinterpret_cast<const float4*>(A);  // rows start at multiples of 16 B
        const int o2 = lane + 32 < 40 ? lane + 32 : 39;         // lanes 8..15 have no third output: a harmless repeat
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll 3
        for (int k4 = 0; k4 < 18; ++k4) {
            const float4 v = A4[k4];
            const float* wr = w1t + k4 * 160;
            c0 = __builtin_fmaf(v.x, wr[lane], c0);
            c1 = __builtin_fmaf(v.x, wr[lane + 16], c1);
            c2 = __builtin_fmaf(v.x, wr[o2], c2);
            c0 = __builtin_fmaf(v.y, wr[40 + lane], c0);
            c1 = __builtin_fmaf(v.y, wr[40 + lane + 16], c1);
            c2 = __builtin_fmaf(v.y, wr[40 + o2], c2);
            c0 = __builtin_fmaf(v.z, wr[80 + lane], c0);
            c1 = __builtin_fmaf(v.z, wr[80 + lane + 16], c1);
            c2 = __builtin_fmaf(v.z, wr[80 + o2], c2);
            c0 = __builtin_fmaf(v.w, wr[120 + lane], c0);
            c1 = __builtin_fmaf(v.w, wr[120 + lane + 16], c1);
            c2 = __builtin_fmaf(v.w, wr[120 + o2], c2);
        }
        c0 += w[3346 + lane];
        c1 += w[3346 + lane + 16];
        c2 += w[3346 + o2];
        B[lane] = c0 > 0.f ? c0 : 0.f;
        B[lane + 16] = c1 > 0.f ? c1 : 0.f;
        if (lane + 32 < 40) B[lane + 32] = c2 > 0.f ? c2 : 0.f;
    }
    layer_sync<false>();
    float feat = 0.f;
    {
        // Linear(40 -> 8): lanes 0..7 own an output each (the others repeat one and drop it)
        const float4* B4 = reinterpret_cast<const float4*>(B);
        const int o = lane & 7;
        float acc = 0.f;
#pragma unroll 5
        for (int k4 = 0; k4 < 10; ++k4) {
            const float4 v = B4[k4];
            const float* wr = w2t + k4 * 32 + o;
            acc = __builtin_fmaf(v.x, wr[0], acc);
            acc = __builtin_fmaf(v.y, wr[8], acc);
            acc = __builtin_fmaf(v.z, wr[16], acc);
            acc = __builtin_fmaf(v.w, wr[24], acc);
        }
        feat = acc + w[3706 + o];
    }
    float* out = a.out + (s0 + s) * cnn_out_dim(CNN_DEEP, a.head);
    if (a.head == HEAD_FEATURES) {
        if (lane < 8) out[lane] = feat;
        return;
    }
    if (lane < 8) A[lane] = feat > 0.f ? feat : 0.f;
    layer_sync<false>();
    if (lane == 0) {
        float acc = 0.f;
        for (int k = 0; k < 8; ++k) acc = __builtin_fmaf(A[k], w[3714 + k], acc);
        acc += w[3722];
        out[0] = acc > 0.f ? acc : 0.f;
    }
}

}  // namespace auv
