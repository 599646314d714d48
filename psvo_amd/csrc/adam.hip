// Fused Adam step on one flat fp32 parameter vector (every trainable variable of the model lives
// in a single contiguous buffer so that the data-parallel gradient all-reduce and the optimizer
// are one collective and one launch).  Restates tf.train.AdamOptimizer(lr) as used by the
// reference (src/trainer.py:115-118; TF 1.12 "epsilon-hat" form):
//     lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
//     theta -= lr_t * m / (sqrt(v) + eps)
// The reference minimises -log_ZSMC; `grad_scale` folds that sign and the 1/world_size of the
// all-reduced (summed) gradient.
#include "common.h"

namespace psvo {
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr_t, float b1, float b2, float eps,
                            float grad_scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * grad_scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
}
}  // namespace psvo

extern "C" int psvo_adam_step(float* params, const float* grads, float* m, float* v, long long n, float lr,
                              float beta1, float beta2, float eps, long long step, float grad_scale,
                              void* stream) {
    using namespace psvo;
    if (!params || !grads || !m || !v || n <= 0 || step <= 0) return PSVO_ERR_INVALID;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    clear_hip_error();
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       params, grads, m, v, n, (float)lr_t, beta1, beta2, eps, grad_scale);
    return launch_status();
}
