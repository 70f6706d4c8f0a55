// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs?  (decides whether a two-piece fp16 split is usable for small values)
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_f16_denorm.hip -o gpurun_out/f16_denorm ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float a, float b, float *out) {
    h8 A, Bv;
    for (int i = 0; i < 8; i++) { A[i] = (_Float16)a; Bv[i] = (_Float16)b; }
    f16v acc; for (int i = 0; i < 16; i++) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, Bv, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float *d; hipMalloc(&d, 4);
    const float as[] = {1.0f, 0x1p-14f, 0x1p-15f, 0x1p-20f, 0x1p-24f, 0x1p-20f};
    const float bs[] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 0x1p-20f};
    for (int i = 0; i < 6; i++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, as[i], bs[i], d);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g  mfma sum=%g  expected=%g\n", as[i], bs[i], h, 16.0 * as[i] * bs[i]);
    }
    return 0;
}
