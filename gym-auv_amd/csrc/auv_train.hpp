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

}  // namespace auv
