// The reference's training step (trainer.py:44-53: forward, MSE loss, backward,
// optimizer step, zero_grad) for its one-convolution network
// (models/lidar_cnn_shallow.py:23-52: Conv1d(1->1, k45, s15, p15, circular) +
// ReLU, Linear(12->1) + ReLU; 59 parameters), fused: ONE launch runs every
// batch of an epoch (trainer.py:80-83), the parameters and the optimizer state
// staying on chip from the first batch to the last.
//
// One workgroup of 256 threads: 16 lanes per sample of a batch of up to 16
// (trainer.py:120-125 uses 16).  Batches are inherently sequential (each step
// reads the parameters the previous one wrote), so there is nothing for a
// second workgroup to do; an epoch of the reference's 4,937 training rows is
// 308 batches.  fp32 throughout, as torch computes it; AdamW and SGD with
// momentum follow torch.optim's update formulas, with the bias corrections
// formed in fp64 like torch's Python scalars.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "auv_cnn.hpp"

namespace auv {

constexpr int kTrainLen = 180, kTrainK = 45, kTrainStride = 15, kTrainPad = 15, kTrainOut = 12;
constexpr int kTrainParams = kTrainK + 1 + kTrainOut + 1;  // conv weight, conv bias, linear weight, linear bias
constexpr int kTrainBatchMax = 16;

struct TrainArgs {
    const float* X;       // [N,180] scans as the network sees them (closeness)
    const float* y;       // [N] labels
    const int32_t* perm;  // [n_batches * batch] sample index of every slot of every batch
    long long n_batches;
    int batch;            // 1..16
    int optimizer;        // 0: AdamW, 1: SGD with momentum
    long long step0;      // optimizer steps taken before this launch
    float* w;             // [59] in/out, state-dict order
    float* opt;           // [2*59] in/out: AdamW exp_avg then exp_avg_sq; SGD: momentum buffer, then unused
    float* batch_loss;    // [n_batches] out: the loss train_batch returns
    double lr, beta1, beta2, eps, weight_decay, momentum;
};

__device__ __forceinline__ float group16_sum(float v) {
    for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_train_shallow(TrainArgs a) {
    __shared__ float xs[kTrainBatchMax][kTrainLen + 4];
    __shared__ float dpre[kTrainBatchMax][kTrainOut];
    __shared__ float gs[kTrainBatchMax][kTrainParams + 1];
    __shared__ float w[kTrainParams + 1];
    __shared__ float err2[kTrainBatchMax];
    const int tid = threadIdx.x, g = tid >> 4, l = tid & 15;
    const int B = a.batch;
    // optimizer state of parameter `tid` lives in this thread's registers for the whole epoch
    float m = 0.f, v = 0.f;
    if (tid < kTrainParams) {
        w[tid] = a.w[tid];
        m = a.opt[tid];
        v = a.opt[kTrainParams + tid];
    }
    double pow1 = 1.0, pow2 = 1.0;  // beta^step, as running products
    for (long long s = 0; s < a.step0; ++s) {
        pow1 *= a.beta1;
        pow2 *= a.beta2;
    }
    __syncthreads();
    for (long long b = 0; b < a.n_batches; ++b) {
        const bool live = g < B;
        float yy = 0.f;
        if (live) {
            const long long idx = a.perm[b * B + g];
            const float* x = a.X + idx * kTrainLen;
            for (int j = l; j < kTrainLen; j += 16) xs[g][j] = x[j];
            yy = a.y[idx];
        }
        __syncthreads();
        // forward
        float pre = 0.f, h = 0.f;
        if (live && l < kTrainOut) {
            float acc = w[kTrainK];
            int j = kTrainStride * l - kTrainPad;
            j = j < 0 ? j + kTrainLen : j;
            for (int k = 0; k < kTrainK; ++k) {
                acc += w[k] * xs[g][j];
                j = j + 1 == kTrainLen ? 0 : j + 1;
            }
            pre = acc;
            h = pre > 0.f ? pre : 0.f;
        }
        const float wl = l < kTrainOut ? w[kTrainK + 1 + l] : 0.f;
        const float z = group16_sum(wl * h) + w[kTrainParams - 1];
        const float yhat = z > 0.f ? z : 0.f;
        const float err = yhat - yy;
        // backward: loss = mean over the batch of err^2
        const float dz = (live && z > 0.f) ? 2.f * err / (float)B : 0.f;
        const float dp = (l < kTrainOut && pre > 0.f) ? dz * wl : 0.f;
        const float gb = group16_sum(dp);
        if (live) {
            if (l < kTrainOut) {
                gs[g][kTrainK + 1 + l] = dz * h;
                dpre[g][l] = dp;
            }
            if (l == 0) {
                gs[g][kTrainParams - 1] = dz;
                gs[g][kTrainK] = gb;
                err2[g] = err * err;
            }
        }
        __syncthreads();
        if (live) {
            for (int k = l; k < kTrainK; k += 16) {
                float acc = 0.f;
                int j = k - kTrainPad;
                j = j < 0 ? j + kTrainLen : j;
                for (int o = 0; o < kTrainOut; ++o) {
                    acc += dpre[g][o] * xs[g][j];
                    j += kTrainStride;
                    j = j >= kTrainLen ? j - kTrainLen : j;
                }
                gs[g][k] = acc;
            }
        }
        __syncthreads();
        // optimizer step: one thread per parameter
        pow1 *= a.beta1;
        pow2 *= a.beta2;
        if (tid < kTrainParams) {
            float grad = 0.f;
            for (int q = 0; q < B; ++q) grad += gs[q][tid];
            float p = w[tid];
            if (a.optimizer == 0) {
                p = p * (float)(1.0 - a.lr * a.weight_decay);
                m = m + (grad - m) * (float)(1.0 - a.beta1);
                v = v * (float)a.beta2 + (float)(1.0 - a.beta2) * (grad * grad);
                const double step_size = a.lr / (1.0 - pow1);
                const float bc2_sqrt = (float)__builtin_sqrt(1.0 - pow2);
                const float denom = __builtin_sqrtf(v) / bc2_sqrt + (float)a.eps;
                p = p - (float)step_size * (m / denom);
            } else {
                m = (a.step0 + b == 0) ? grad : (float)a.momentum * m + grad;
                p = p - (float)a.lr * m;
            }
            w[tid] = p;
        } else if (tid == 64) {
            float s = 0.f;
            for (int q = 0; q < B; ++q) s += err2[q];
            a.batch_loss[b] = s / (float)B;
        }
        __syncthreads();
    }
    if (tid < kTrainParams) {
        a.w[tid] = w[tid];
        a.opt[tid] = m;
        a.opt[kTrainParams + tid] = v;
    }
}


// ---- the same step for any of the three networks --------------------------
// models/lidar_cnn_deep.py:110-146 (three convolutions, 178 parameters) and
// :19-82 with channels [2,4,4,6] and kernel 9 (trainer.py:144; four
// convolution + max-pool stages and three linear layers, 3,723 parameters).
//
// Every layer of the three networks is one shape: a circular Conv1d, then an
// optional ReLU, then an optional MaxPool1d(2, 2, ceil_mode=True); a Linear
// layer is the convolution with length 1 and kernel 1 over n_in channels
// (Flatten is channel-major, which is how the activations are stored).
struct TrainLayer {
    int Cin, Cout, K, stride, pad, L, relu, pool;
};

__host__ __device__ constexpr int train_n_layers(int kind) { return kind == CNN_SHALLOW ? 2 : kind == CNN_2DEEP ? 4 : 7; }

__host__ __device__ constexpr TrainLayer train_layer(int kind, int l) {
    if (kind == CNN_SHALLOW) return l == 0 ? TrainLayer{1, 1, 45, 15, 15, 180, 1, 0} : TrainLayer{12, 1, 1, 1, 0, 1, 1, 0};
    if (kind == CNN_2DEEP)
        return l == 0   ? TrainLayer{1, 3, 45, 15, 15, 180, 1, 0}
               : l == 1 ? TrainLayer{3, 2, 3, 1, 1, 12, 1, 0}
               : l == 2 ? TrainLayer{2, 1, 3, 1, 1, 12, 0, 0}
                        : TrainLayer{12, 1, 1, 1, 0, 1, 1, 0};
    return l == 0   ? TrainLayer{1, 2, 9, 1, 4, 180, 1, 1}
           : l == 1 ? TrainLayer{2, 4, 9, 1, 4, 90, 1, 1}
           : l == 2 ? TrainLayer{4, 4, 9, 1, 4, 45, 1, 1}
           : l == 3 ? TrainLayer{4, 6, 9, 1, 4, 23, 1, 1}
           : l == 4 ? TrainLayer{72, 40, 1, 1, 0, 1, 1, 0}
           : l == 5 ? TrainLayer{40, 8, 1, 1, 0, 1, 1, 0}   // linear_1's last layer, then the ReLU that opens linear_2
                    : TrainLayer{8, 1, 1, 1, 0, 1, 1, 0};
}
__host__ __device__ constexpr int tl_conv_len(TrainLayer d) { return (d.L + 2 * d.pad - d.K) / d.stride + 1; }
__host__ __device__ constexpr int tl_out_len(TrainLayer d) { return d.pool ? (tl_conv_len(d) + 1) / 2 : tl_conv_len(d); }
__host__ __device__ constexpr int tl_n_out(TrainLayer d) { return d.Cout * tl_out_len(d); }
__host__ __device__ constexpr int tl_n_in(TrainLayer d) { return d.Cin * d.L; }
__host__ __device__ constexpr int tl_n_weights(TrainLayer d) { return d.Cout * d.Cin * d.K; }
// packed parameters (state-dict order): weight then bias of every layer in turn
__host__ __device__ constexpr int train_w_off(int kind, int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += tl_n_weights(train_layer(kind, i)) + train_layer(kind, i).Cout;
    return o;
}
// activation record of one sample: the scan, then every layer's output
__host__ __device__ constexpr int train_act_off(int kind, int l) {  // of layer l's output
    int o = kCnnLen;
    for (int i = 0; i < l; ++i) o += tl_n_out(train_layer(kind, i));
    return o;
}
__host__ __device__ constexpr int train_act_floats(int kind) { return train_act_off(kind, train_n_layers(kind)) | 1; }  // odd stride
// which element of its pair a pooled output came from, one byte per pooled output
__host__ __device__ constexpr int train_mask_off(int kind, int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += train_layer(kind, i).pool ? tl_n_out(train_layer(kind, i)) : 0;
    return o;
}
__host__ __device__ constexpr int train_mask_bytes(int kind) { return (train_mask_off(kind, train_n_layers(kind)) + 4) & ~3; }
// gradient buffers per sample: with respect to a convolution's output (before ReLU and pooling), and to a layer's input
__host__ __device__ constexpr int train_dpre_floats(int kind) {
    int m = 1;
    for (int i = 0; i < train_n_layers(kind); ++i) {
        const int n = train_layer(kind, i).Cout * tl_conv_len(train_layer(kind, i));
        m = n > m ? n : m;
    }
    return m | 1;
}
__host__ __device__ constexpr int train_dout_floats(int kind) {
    int m = 1;
    for (int i = 1; i < train_n_layers(kind); ++i) m = tl_n_in(train_layer(kind, i)) > m ? tl_n_in(train_layer(kind, i)) : m;
    return m | 1;
}
constexpr int kTrainNetThreads = 1024;
__host__ __device__ constexpr int train_params_padded(int kind) { return (cnn_n_weights(kind) + 3) & ~3; }
__host__ __device__ constexpr size_t train_net_lds_bytes(int kind) {
    return ((size_t)2 * train_params_padded(kind) +
            (size_t)kTrainBatchMax * (train_act_floats(kind) + train_dpre_floats(kind) + train_dout_floats(kind) + 1)) * 4 +
           (size_t)kTrainBatchMax * train_mask_bytes(kind);
}
// lanes that share one parameter's gradient sum: a power of two up to 64, as many as keep one pass over the
// workgroup's nt threads
__host__ __device__ constexpr int train_group(int n_items, int nt) {
    int g = 1;
    while (g < 64 && n_items * g * 2 <= nt) g *= 2;
    return g;
}

struct TrainLds {
    float* w;             // parameters
    float* grad;          // their gradients of the current batch
    float* act;           // [16][act_floats]
    float* dpre;          // [16][dpre_floats]
    float* dout;          // [16][dout_floats]
    float* err2;          // [16]
    unsigned char* mask;  // [16][mask_bytes]
};

// forward of layer l for the B samples of the batch; one thread per output element
template <int KIND, int l, int NT>
__device__ __forceinline__ void train_forward_layer(const TrainLds& m, int B, int tid) {
    constexpr TrainLayer D = train_layer(KIND, l);
    constexpr int Lc = tl_conv_len(D), Lo = tl_out_len(D), n_out = tl_n_out(D), A = train_act_floats(KIND);
    constexpr int in_off = l == 0 ? 0 : train_act_off(KIND, l - 1), out_off = train_act_off(KIND, l);
    constexpr int w_off = train_w_off(KIND, l), b_off = w_off + tl_n_weights(D);
    static_assert(D.pad <= D.L && (Lc - 1) * D.stride + D.K - 1 - D.pad < 2 * D.L, "one wrap at most");
    for (int idx = tid; idx < B * n_out; idx += NT) {
        const int s = idx / n_out, el = idx - s * n_out;
        const int co = el / Lo, lo = el - co * Lo;
        const float* in = m.act + s * A + in_off;
        float best = 0.f;
        int arg = 0;
        const int n_pos = (D.pool && 2 * lo + 1 < Lc) ? 2 : 1;
        for (int j = 0; j < n_pos; ++j) {
            const int pos = D.pool ? 2 * lo + j : lo;
            float acc = m.w[b_off + co];
            for (int ci = 0; ci < D.Cin; ++ci) {
                const float* wr = m.w + w_off + (co * D.Cin + ci) * D.K;
                const float* xr = in + ci * D.L;
                int p = pos * D.stride - D.pad;
                p = p < 0 ? p + D.L : p;
#pragma unroll
                for (int k = 0; k < D.K; ++k) {
                    acc += wr[k] * xr[p];
                    p = (p + 1 == D.L) ? 0 : p + 1;
                }
            }
            if (D.relu) acc = acc > 0.f ? acc : 0.f;
            if (j == 0 || acc > best) {
                best = acc;
                arg = j;
            }
        }
        m.act[s * A + out_off + el] = best;
        if (D.pool) m.mask[s * train_mask_bytes(KIND) + train_mask_off(KIND, l) + el] = (unsigned char)arg;
    }
}

template <int KIND, int l, int NT>
__device__ __forceinline__ void train_forward_from(const TrainLds& m, int B, int tid) {
    train_forward_layer<KIND, l, NT>(m, B, tid);
    __syncthreads();
    if constexpr (l + 1 < train_n_layers(KIND)) train_forward_from<KIND, l + 1, NT>(m, B, tid);
}

// backward of layer l: dout holds the gradient with respect to the layer's
// output on entry, with respect to its input on return (layer 0: untouched);
// the gradients of the layer's parameters go to grad[]
template <int KIND, int l, int NT>
__device__ __forceinline__ void train_backward_layer(const TrainLds& m, int B, int tid) {
    constexpr TrainLayer D = train_layer(KIND, l);
    constexpr int Lc = tl_conv_len(D), Lo = tl_out_len(D), n_in = tl_n_in(D), A = train_act_floats(KIND);
    constexpr int in_off = l == 0 ? 0 : train_act_off(KIND, l - 1), out_off = train_act_off(KIND, l);
    constexpr int w_off = train_w_off(KIND, l), b_off = w_off + tl_n_weights(D);
    constexpr int DP = train_dpre_floats(KIND), DO = train_dout_floats(KIND);
    // through the pooling and the ReLU: only the element a pooled output came from, and only if it was positive
    for (int idx = tid; idx < B * D.Cout * Lc; idx += NT) {
        const int s = idx / (D.Cout * Lc), r = idx - s * (D.Cout * Lc);
        const int co = r / Lc, pos = r - co * Lc;
        const int o = co * Lo + (D.pool ? pos >> 1 : pos);
        float g = m.dout[s * DO + o];
        if (D.relu) g = m.act[s * A + out_off + o] > 0.f ? g : 0.f;
        if (D.pool) g = m.mask[s * train_mask_bytes(KIND) + train_mask_off(KIND, l) + o] == (pos & 1) ? g : 0.f;
        m.dpre[s * DP + r] = g;
    }
    __syncthreads();
    // parameters: one item per weight, and per bias; G lanes share an item's sum over samples and positions
    constexpr int per_co = D.Cin * D.K + 1, n_items = D.Cout * per_co, G = train_group(n_items, NT);
    for (int base = 0; base < n_items * G; base += NT) {  // the same trip count for every thread
        const int idx = base + tid, item = idx / G, g = idx - item * G;
        const bool on = item < n_items;
        const int co = on ? item / per_co : 0, q = on ? item - co * per_co : 0;
        float acc = 0.f;
        if (on && q < D.Cin * D.K) {
            const int ci = q / D.K, k = q - ci * D.K;
            for (int t = g; t < B * Lc; t += G) {
                const int s = t / Lc, pos = t - s * Lc;
                int p = pos * D.stride + k - D.pad;
                p = p < 0 ? p + D.L : (p >= D.L ? p - D.L : p);
                acc += m.dpre[s * DP + co * Lc + pos] * m.act[s * A + in_off + ci * D.L + p];
            }
        } else if (on) {
            for (int t = g; t < B * Lc; t += G) {
                const int s = t / Lc, pos = t - s * Lc;
                acc += m.dpre[s * DP + co * Lc + pos];
            }
        }
#pragma unroll
        for (int d = G >> 1; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if (on && g == 0) m.grad[q < D.Cin * D.K ? w_off + co * D.Cin * D.K + q : b_off + co] = acc;
    }
    // input of the layer = output of the one before
    if constexpr (l > 0) {
        static_assert(D.stride == 1 && Lc == D.L, "the inner layers keep their length");
        for (int idx = tid; idx < B * n_in; idx += NT) {
            const int s = idx / n_in, r = idx - s * n_in;
            const int ci = r / D.L, p = r - ci * D.L;
            float acc = 0.f;
            for (int co = 0; co < D.Cout; ++co) {
                const float* wr = m.w + w_off + (co * D.Cin + ci) * D.K;
                const float* gr = m.dpre + s * DP + co * Lc;
                int pos = p + D.pad;
                pos = pos >= D.L ? pos - D.L : pos;
#pragma unroll
                for (int k = 0; k < D.K; ++k) {
                    acc += gr[pos] * wr[k];
                    pos = pos == 0 ? D.L - 1 : pos - 1;
                }
            }
            m.dout[s * DO + r] = acc;
        }
    }
    __syncthreads();
}

template <int KIND, int l, int NT>
__device__ __forceinline__ void train_backward_from(const TrainLds& m, int B, int tid) {
    train_backward_layer<KIND, l, NT>(m, B, tid);
    if constexpr (l > 0) train_backward_from<KIND, l - 1, NT>(m, B, tid);
}

// One workgroup of 1,024 threads; parameters, the batch's activations and
// gradients in LDS (121 KiB for the largest network), optimizer state in the
// registers of the thread that owns the parameter.  TrainArgs as above, with
// w [n] and opt [2n] for the network's n parameters.
template <int KIND>
__global__ __launch_bounds__(kTrainNetThreads) void k_train_net(TrainArgs a) {
    extern __shared__ __align__(16) unsigned char train_lds[];
    constexpr int NP = cnn_n_weights(KIND), NPP = train_params_padded(KIND), A = train_act_floats(KIND);
    constexpr int NL = train_n_layers(KIND), PER = (NP + kTrainNetThreads - 1) / kTrainNetThreads;
    static_assert(train_w_off(KIND, NL) == NP, "layer table and parameter count disagree");
    static_assert(tl_n_out(train_layer(KIND, NL - 1)) == 1, "one output per sample");
    TrainLds m;
    m.w = reinterpret_cast<float*>(train_lds);
    m.grad = m.w + NPP;
    m.act = m.grad + NPP;
    m.dpre = m.act + kTrainBatchMax * A;
    m.dout = m.dpre + kTrainBatchMax * train_dpre_floats(KIND);
    m.err2 = m.dout + kTrainBatchMax * train_dout_floats(KIND);
    m.mask = reinterpret_cast<unsigned char*>(m.err2 + kTrainBatchMax);
    const int tid = threadIdx.x;
    const int B = a.batch;
    float mom[PER], var[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int p = tid + j * kTrainNetThreads;
        mom[j] = var[j] = 0.f;
        if (p < NP) {
            m.w[p] = a.w[p];
            mom[j] = a.opt[p];
            var[j] = a.opt[NP + p];
        }
    }
    double pow1 = 1.0, pow2 = 1.0;
    for (long long s = 0; s < a.step0; ++s) {
        pow1 *= a.beta1;
        pow2 *= a.beta2;
    }
    __syncthreads();
    for (long long b = 0; b < a.n_batches; ++b) {
        for (int idx = tid; idx < B * kCnnLen; idx += kTrainNetThreads) {
            const int s = idx / kCnnLen, j = idx - s * kCnnLen;
            m.act[s * A + j] = a.X[(long long)a.perm[b * B + s] * kCnnLen + j];
        }
        __syncthreads();
        train_forward_from<KIND, 0, kTrainNetThreads>(m, B, tid);
        // loss = mean over the batch of (yhat - y)^2
        if (tid < B) {
            const float err = m.act[tid * A + train_act_off(KIND, NL - 1)] - a.y[a.perm[b * B + tid]];
            m.err2[tid] = err * err;
            m.dout[tid * train_dout_floats(KIND)] = 2.f * err / (float)B;
        }
        __syncthreads();
        train_backward_from<KIND, NL - 1, kTrainNetThreads>(m, B, tid);
        pow1 *= a.beta1;
        pow2 *= a.beta2;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int p = tid + j * kTrainNetThreads;
            if (p < NP) {
                const float grad = m.grad[p];
                float w = m.w[p];
                if (a.optimizer == 0) {
                    w = w * (float)(1.0 - a.lr * a.weight_decay);
                    mom[j] = mom[j] + (grad - mom[j]) * (float)(1.0 - a.beta1);
                    var[j] = var[j] * (float)a.beta2 + (float)(1.0 - a.beta2) * (grad * grad);
                    const double step_size = a.lr / (1.0 - pow1);
                    const float bc2_sqrt = (float)__builtin_sqrt(1.0 - pow2);
                    const float denom = __builtin_sqrtf(var[j]) / bc2_sqrt + (float)a.eps;
                    w = w - (float)step_size * (mom[j] / denom);
                } else {
                    mom[j] = (a.step0 + b == 0) ? grad : (float)a.momentum * mom[j] + grad;
                    w = w - (float)a.lr * mom[j];
                }
                m.w[p] = w;
            }
        }
        if (tid == kTrainNetThreads - 1) {
            float s = 0.f;
            for (int q = 0; q < B; ++q) s += m.err2[q];
            a.batch_loss[b] = s / (float)B;
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int p = tid + j * kTrainNetThreads;
        if (p < NP) {
            a.w[p] = m.w[p];
            a.opt[p] = mom[j];
            a.opt[NP + p] = var[j];
        }
    }
}

// ---- the same step with a workgroup per sample ------------------------------
// For the largest network one CU is the limit of k_train_net (its batch is
// ~1.2 M multiply-adds on 64 lanes per cycle).  Here a batch is two launches:
// k_train_sample, one workgroup per sample of the batch (forward, loss,
// backward; the sample's parameter gradients go to `gsample`), then
// k_train_update (gradient = the samples' in order, optimizer step; parameters
// and optimizer state stay in global memory between launches).  Batches are
// still sequential: the host enqueues the 2 x n_batches launches of an epoch.
struct TrainSplitArgs {
    TrainArgs t;
    long long batch_index;  // which batch of perm / batch_loss
    float* gsample;         // [16][params padded] per-sample gradients
    float* err2;            // [16] squared errors of the batch
    double pow1, pow2;      // beta1^step, beta2^step of THIS step (step counted from 1)
};

constexpr int kTrainSampleThreads = 1024;
__host__ __device__ constexpr size_t train_sample_lds_bytes(int kind) {
    return ((size_t)2 * train_params_padded(kind) + train_act_floats(kind) + train_dpre_floats(kind) +
            train_dout_floats(kind) + 1) * 4 + train_mask_bytes(kind);
}

template <int KIND>
__global__ __launch_bounds__(kTrainSampleThreads) void k_train_sample(TrainSplitArgs a) {
    extern __shared__ __align__(16) unsigned char train_lds[];
    constexpr int NP = cnn_n_weights(KIND), NPP = train_params_padded(KIND), NL = train_n_layers(KIND);
    constexpr int NT = kTrainSampleThreads;
    TrainLds m;
    m.w = reinterpret_cast<float*>(train_lds);
    m.grad = m.w + NPP;
    m.act = m.grad + NPP;
    m.dpre = m.act + train_act_floats(KIND);
    m.dout = m.dpre + train_dpre_floats(KIND);
    m.err2 = m.dout + train_dout_floats(KIND);
    m.mask = reinterpret_cast<unsigned char*>(m.err2 + 1);
    const int tid = threadIdx.x, s = blockIdx.x, B = a.t.batch;
    const long long row = a.t.perm[a.batch_index * B + s];
    for (int p = tid; p < NP; p += NT) m.w[p] = a.t.w[p];
    for (int j = tid; j < kCnnLen; j += NT) m.act[j] = a.t.X[row * kCnnLen + j];
    __syncthreads();
    train_forward_from<KIND, 0, NT>(m, 1, tid);
    if (tid == 0) {
        const float err = m.act[train_act_off(KIND, NL - 1)] - a.t.y[row];
        a.err2[s] = err * err;
        m.dout[0] = 2.f * err / (float)B;
    }
    __syncthreads();
    train_backward_from<KIND, NL - 1, NT>(m, 1, tid);
    for (int p = tid; p < NP; p += NT) a.gsample[(size_t)s * NPP + p] = m.grad[p];
}

template <int KIND>
__global__ __launch_bounds__(256) void k_train_update(TrainSplitArgs a) {
    constexpr int NP = cnn_n_weights(KIND), NPP = train_params_padded(KIND);
    const int p = blockIdx.x * 256 + threadIdx.x, B = a.t.batch;
    if (p < NP) {
        float grad = 0.f;
        for (int q = 0; q < B; ++q) grad += a.gsample[(size_t)q * NPP + p];
        float w = a.t.w[p], mom = a.t.opt[p], var = a.t.opt[NP + p];
        if (a.t.optimizer == 0) {
            w = w * (float)(1.0 - a.t.lr * a.t.weight_decay);
            mom = mom + (grad - mom) * (float)(1.0 - a.t.beta1);
            var = var * (float)a.t.beta2 + (float)(1.0 - a.t.beta2) * (grad * grad);
            const double step_size = a.t.lr / (1.0 - a.pow1);
            const float bc2_sqrt = (float)__builtin_sqrt(1.0 - a.pow2);
            const float denom = __builtin_sqrtf(var) / bc2_sqrt + (float)a.t.eps;
            w = w - (float)step_size * (mom / denom);
        } else {
            mom = (a.t.step0 + a.batch_index == 0) ? grad : (float)a.t.momentum * mom + grad;
            w = w - (float)a.t.lr * mom;
        }
        a.t.w[p] = w;
        a.t.opt[p] = mom;
        a.t.opt[NP + p] = var;
    }
    if (p == 0) {
        float sum = 0.f;
        for (int q = 0; q < B; ++q) sum += a.err2[q];
        a.t.batch_loss[a.batch_index] = sum / (float)B;
    }
}

}  // namespace auv
