// backward.hip -- HBM-bound fp32 kernels of the backward pass (everything that is not a convolution
// dgrad / wgrad): 2x2 sum-pool (backward of the nearest x2 upsample), per-channel sums (bias grads),
// GroupNorm+FiLM+SiLU backward, channel-LayerNorm backward, resnet-block output backward, loss grad.
#include "common.h"
#include "ops.h"
#include "../../include/tedm_hip.h"

namespace {

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP)

// backward of nn.Upsample(scale_factor=2, mode='nearest') (unet_model.py:41): out[y][x] = sum of the 2x2 block
__global__ void pool2x2_sum_kernel(const float* __restrict__ g, float* __restrict__ out, int B, int h, int w, int C,
                                   long long total4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (C / 4)) * 4;
        long long r = i / (C / 4);
        const int x = (int)(r % w); r /= w;
        const int y = (int)(r % h);
        const int n = (int)(r / h);
        const float* p = g + (((size_t)n * 2 * h + 2 * y) * 2 * w + 2 * x) * C + c4;
        const float4 a = *(const float4*)p, b = *(const float4*)(p + C);
        const float4 c = *(const float4*)(p + (size_t)2 * w * C), d = *(const float4*)(p + (size_t)2 * w * C + C);
        *(float4*)(out + (((size_t)n * h + y) * w + x) * C + c4) =
            make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
    }
}

}  // namespace

int op_pool2x2_sum(const float* g, float* out, int B, int h, int w, int C, hipStream_t st) {
    if (C % 4) return TEDM_EINVAL;
    const long long total4 = (long long)B * h * w * (C / 4);
    long long blocks = (total4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pool2x2_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, out, B, h, w, C, total4);
    return LAUNCH_OK();
}
