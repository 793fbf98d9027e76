// ovfl_probe.hip -- diagnostic: does v_cvt_pk_f16_f32 honour MODE.FP16_OVFL (bit 23) on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/ovfl_probe.hip -o tools/build/ovfl_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
__global__ void k(const float* in, unsigned* out, int mode) {
    if (mode) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);     // hwreg(HW_REG_MODE, 23, 1) = 1
    const float a = in[2 * threadIdx.x], b = in[2 * threadIdx.x + 1];
    unsigned hp;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(a), "v"(b));
    float r0, r1;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hp), "v"(a));
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hp), "v"(b));
    out[3 * threadIdx.x] = hp; out[3 * threadIdx.x + 1] = __float_as_uint(r0); out[3 * threadIdx.x + 2] = __float_as_uint(r1);
}
int main() {
    const float h[8] = {1.0f, 65504.f, 65520.f, 70000.f, -1e6f, 3e38f, 65519.9f, -65536.f};
    float* d; unsigned* o; unsigned ho[12];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o, mode);
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d:", mode);
        for (int i = 0; i < 4; ++i) { float r0, r1; memcpy(&r0, &ho[3*i+1], 4); memcpy(&r1, &ho[3*i+2], 4); printf("  (%g,%g)->%08x res %g %g", h[2*i], h[2*i+1], ho[3*i], r0, r1); }
        printf("\n");
    }
    return 0;
}
