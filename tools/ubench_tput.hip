// throughput of VALU instruction classes with every SIMD saturated (8 waves per SIMD, 8 independent chains per wave)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 4096
#define BODY8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double seed, float fseed) {
    double d[8]; float f[8]; int sg[8] = {0,0,0,0,0,0,0,0};
    const unsigned long long msk = __ballot(threadIdx.x & 1);
    for (int i = 0; i < 8; i++) { d[i] = seed + i + threadIdx.x; f[i] = fseed + i + threadIdx.x; }
    for (int it = 0; it < N_IT; it++) {
#define S_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(seed));
#define S_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(seed));
#define S_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(seed));
#define S_CVT6432(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
#define S_CVT3264(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
#define S_ADD32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fseed));
#define S_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(fseed));
#define S_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]));
#define S_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fseed));
#define S_CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fseed), "s"(msk));
#define S_CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(fseed) : "vcc");
#define S_RDL(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg[i]) : "v"(f[i]));
#define S_CMP(i) asm volatile("v_cmp_eq_f32 vcc, %0, %1" :: "v"(f[i]), "v"(fseed) : "vcc");
        if (OP == 0) { BODY8(S_ADD64) } if (OP == 1) { BODY8(S_FMA64) } if (OP == 2) { BODY8(S_MUL64) }
        if (OP == 3) { BODY8(S_CVT6432) } if (OP == 4) { BODY8(S_CVT3264) } if (OP == 5) { BODY8(S_ADD32) }
        if (OP == 6) { BODY8(S_CND) } if (OP == 7) { BODY8(S_DPP) } if (OP == 8) { BODY8(S_MAX3) } if (OP == 9) { BODY8(S_CMP) } if (OP == 10) { BODY8(S_CND64) } if (OP == 11) { BODY8(S_CMPCND) } if (OP == 12) { BODY8(S_RDL) }
    }
    double s = 0; for (int i = 0; i < 8; i++) s += d[i] + f[i] + sg[i];
    if (s == 12345.678) out[0] = s;
}
template <int OP> void run(const char *name) {
    double *o; hipMalloc(&o, 8);
    const int blocks = 256 * 8;       // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(o, 1.5, 2.5f); hipDeviceSynchronize();
    hipEventRecord(a); k<OP><<<blocks, 256>>>(o, 1.5, 2.5f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double insts_per_simd = 8.0 /*waves*/ * N_IT * 8.0;
    printf("%-16s %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / insts_per_simd);
    hipFree(o);
}
int main() {
    run<0>("v_add_f64"); run<1>("v_fma_f64"); run<2>("v_mul_f64"); run<3>("v_cvt_f32_f64"); run<4>("v_cvt_f64_f32"); run<5>("v_add_f32");
    run<6>("v_cndmask_b32"); run<7>("v_mov_dpp ror"); run<8>("v_max3_f32"); run<9>("v_cmp_eq_f32"); run<10>("v_cndmask e64 sgpr"); run<11>("v_cmp+v_cndmask vcc (2 instr)"); run<12>("v_readlane");
    return 0;
}
