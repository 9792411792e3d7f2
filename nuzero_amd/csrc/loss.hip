// Batched policy/value loss with its gradients in one launch (C ABI nz_loss_*).
//
// Replaces the per-sample Python loop of AlphaZero.calculate_loss (Training/AlphaZero.py:891-921): for every
// sample i of the batch it calls policy_loss_function(flatten(policy_logits[i]), target_policy[i]) and
// value_loss_function(value[i], target_value[i]), sums them, divides the policy sum by log(batch) when
// `normalize_cel` is on (AlphaZero.py:912-915: `target_size = len(targets)` is the batch size), then both by the
// batch size; combined = policy + value.  Policy losses (AlphaZero.py:325-333):
//   CE   nn.CrossEntropyLoss(label_smoothing = 0.02) with probability targets:
//        t' = t (1 - eps) + eps / A ;  loss = - sum_a t'_a log_softmax(x)_a
//   KLD  Utils/Functions/loss_functions.py:7-11: nn.KLDivLoss() (reduction 'mean' = over the A elements) on
//        log_softmax(x): (1 / A) sum_a [t_a > 0] t_a (log t_a - log_softmax(x)_a)
//   MSE  loss_functions.py:13-26: over the actions with a non-zero target, (t_a - softmax(x)_a)^2, mean over them
// Value losses (loss_functions.py:28-33): SE (t - v)^2, AE |t - v|.
//
// One workgroup per sample: log-sum-exp of the logits (wave shuffles + one LDS exchange), the sample's loss terms and
// d(combined) / d(logits), d(combined) / d(value) in the same pass over the A actions; per-sample losses are summed
// on the host side of the ABI by a second tiny kernel in a fixed order (so the result does not depend on scheduling).
// float32 throughout, like the reference (which accumulates float32 tensors); HBM-bound: 12 bytes per action.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/nuzero_amd.h"

namespace {
thread_local std::string g_err;

struct LossArgs {
  const float* logits;          // [B][A]
  const float* values;          // [B]
  const float* target_policies; // [B][A]
  const float* target_values;   // [B]
  float* dlogits;               // [B][A] or nullptr
  float* dvalues;               // [B] or nullptr
  float* per_sample;            // [B][2]: policy term, value term (before the batch division)
  int32_t batch, actions, policy_loss, value_loss;
  float policy_scale, value_scale;   // 1 / (batch [* log batch]), 1 / batch
  float smoothing;
};

__device__ __forceinline__ float block_reduce(float v, float* scratch, bool is_max) {
  for (int o = 32; o; o >>= 1) {
    const float w = __shfl_xor(v, o, 64);
    v = is_max ? fmaxf(v, w) : v + w;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float r = scratch[0];
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, scratch[i]) : r + scratch[i];
  return r;
}

__global__ __launch_bounds__(256) void loss_kernel(LossArgs a) {
  __shared__ float scratch[4];
  const int b = blockIdx.x, tid = threadIdx.x, A = a.actions;
  const float* x = a.logits + (size_t)b * A;
  const float* t = a.target_policies + (size_t)b * A;
  float mx = -INFINITY;
  for (int i = tid; i < A; i += 256) mx = fmaxf(mx, x[i]);
  mx = block_reduce(mx, scratch, true);
  float se = 0.f;
  for (int i = tid; i < A; i += 256) se += expf(x[i] - mx);
  se = block_reduce(se, scratch, false);
  const float lse = mx + logf(se);

  float loss_part = 0.f, aux = 0.f, cnt = 0.f;
  if (a.policy_loss == NZ_LOSS_CE) {
    const float eps = a.smoothing, uni = eps / (float)A;
    for (int i = tid; i < A; i += 256) {
      const float tp = t[i] * (1.0f - eps) + uni;
      loss_part -= tp * (x[i] - lse);
    }
  } else if (a.policy_loss == NZ_LOSS_KLD) {
    for (int i = tid; i < A; i += 256) {
      const float ti = t[i];
      if (ti > 0.f) loss_part += ti * (logf(ti) - (x[i] - lse));
      aux += ti;                                   // sum of the targets (gradient of the log-softmax term)
    }
  } else {                                         // masked MSE on softmax(x)
    for (int i = tid; i < A; i += 256) {
      const float ti = t[i];
      if (ti != 0.f) {
        const float p = expf(x[i] - lse), d = ti - p;
        loss_part += d * d;
        aux += d * p;                              // sum_j [t_j != 0] (t_j - p_j) p_j
        cnt += 1.f;
      }
    }
  }
  loss_part = block_reduce(loss_part, scratch, false);
  if (a.policy_loss != NZ_LOSS_CE) aux = block_reduce(aux, scratch, false);
  if (a.policy_loss == NZ_LOSS_MSE) cnt = block_reduce(cnt, scratch, false);
  float policy_term = loss_part;
  if (a.policy_loss == NZ_LOSS_KLD) policy_term = loss_part / (float)A;
  if (a.policy_loss == NZ_LOSS_MSE) policy_term = loss_part / cnt;

  if (a.dlogits) {
    float* g = a.dlogits + (size_t)b * A;
    const float s = a.policy_scale;
    if (a.policy_loss == NZ_LOSS_CE) {
      const float eps = a.smoothing, uni = eps / (float)A;
      float st = 0.f;                              // d/dx_i of -sum_a t'_a log_softmax(x)_a = softmax(x)_i sum(t') - t'_i
      for (int i = tid; i < A; i += 256) st += t[i] * (1.0f - eps) + uni;
      st = block_reduce(st, scratch, false);
      for (int i = tid; i < A; i += 256) {
        const float tp = t[i] * (1.0f - eps) + uni;
        g[i] = s * (expf(x[i] - lse) * st - tp);
      }
    } else if (a.policy_loss == NZ_LOSS_KLD) {
      const float k = s / (float)A;
      for (int i = tid; i < A; i += 256) g[i] = k * (expf(x[i] - lse) * aux - t[i]);
    } else {
      // (no target entry at all: the loss is 0 / 0 -- the reference raises ZeroDivisionError, loss_functions.py:7-26; the
      // sample's policy term is NaN, so the batch's loss says so, but its gradient row is zeros: one bad sample does
      // not poison the other samples' gradients)
      const float k = cnt > 0.f ? 2.0f * s / cnt : 0.f;
      for (int i = tid; i < A; i += 256) {         // d/dx_i sum_j m_j (t_j - p_j)^2 = 2 p_i (sum_j m_j (t_j - p_j) p_j - m_i (t_i - p_i))
        const float p = expf(x[i] - lse);
        const float mi = t[i] != 0.f ? (t[i] - p) : 0.f;
        g[i] = k * p * (aux - mi);
      }
    }
  }
  if (tid == 0) {
    const float v = a.values[b], tv = a.target_values[b], d = tv - v;
    float value_term, dv;
    if (a.value_loss == NZ_LOSS_SE) { value_term = d * d; dv = -2.0f * d; }
    else { value_term = fabsf(d); dv = d > 0.f ? -1.0f : (d < 0.f ? 1.0f : 0.0f); }
    if (a.dvalues) a.dvalues[b] = a.value_scale * dv;
    a.per_sample[2 * b] = policy_term;
    a.per_sample[2 * b + 1] = value_term;
  }
}

// sums of the per-sample terms in sample order (what the reference's `+=` loop does), one thread: B is a batch size
__global__ void loss_sum_kernel(const float* __restrict__ per_sample, int batch, float policy_scale, float value_scale,
                                float* __restrict__ out3) {
  float p = 0.f, v = 0.f;
  for (int i = 0; i < batch; ++i) {
    p += per_sample[2 * i];
    v += per_sample[2 * i + 1];
  }
  const float vl = v * value_scale, pl = p * policy_scale;
  out3[0] = vl;
  out3[1] = pl;
  out3[2] = pl + vl;
}
}  // namespace

extern "C" {

const char* nz_loss_last_error(void) { return g_err.c_str(); }

nz_status nz_loss_forward_backward(const float* logits_dev, const float* values_dev, const float* target_policies_dev,
                                   const float* target_values_dev, int32_t batch, int32_t actions, int32_t policy_loss,
                                   int32_t value_loss, int32_t normalize_policy, float* losses3_dev, float* dlogits_dev,
                                   float* dvalues_dev, float* workspace_dev, void* stream) {
  if (!logits_dev || !values_dev || !target_policies_dev || !target_values_dev || !losses3_dev || !workspace_dev) {
    g_err = "null argument";
    return NZ_ERR_ARG;
  }
  if (batch <= 0 || actions <= 0 || policy_loss < NZ_LOSS_CE || policy_loss > NZ_LOSS_MSE ||
      (value_loss != NZ_LOSS_SE && value_loss != NZ_LOSS_AE)) {
    g_err = "bad batch / actions / loss selector";
    return NZ_ERR_ARG;
  }
  if (normalize_policy && batch == 1) {
    g_err = "normalize_cel with a batch of one divides by log(1) = 0 (AlphaZero.py:912-915)";
    return NZ_ERR_ARG;
  }
  const float inv_b = 1.0f / (float)batch;
  LossArgs a{logits_dev, values_dev, target_policies_dev, target_values_dev, dlogits_dev, dvalues_dev, workspace_dev,
             batch, actions, policy_loss, value_loss,
             normalize_policy ? (1.0f / logf((float)batch)) * inv_b : inv_b, inv_b, 0.02f};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_kernel, dim3(batch), dim3(256), 0, s, a);
  hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(1), 0, s, workspace_dev, batch, a.policy_scale, a.value_scale,
                     losses3_dev);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_err = std::string("launch failed: ") + hipGetErrorString(e);
    return NZ_ERR_HIP;
  }
  return NZ_OK;
}

}  // extern "C"
