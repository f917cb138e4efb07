// Microbenchmark (tuning aid): issue cost of the VALU instructions the EqThree kernels are made of, gfx950.
// 4 waves per SIMD, 8 independent streams per lane: cycles per wave-instruction per SIMD at saturation.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_cost.cpp -o /tmp/valu_cost && /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void k_op(double* out, int iters) {
    double d[8]; float f[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { d[i] = 1.0 + threadIdx.x * 1e-3 + i; f[i] = 1.0f + i; u[i] = threadIdx.x + i; }
    double s1 = 0.999, s2 = 1e-9; unsigned long long m64 = 0x5555555555555555ull;
    asm volatile("" : "+s"(m64));
    asm volatile("" : "+s"(s1), "+s"(s2));
    for (int it = 0; it < iters; ++it) {
#define ADD(i) if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "s"(s2));
#define MUL(i) if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "s"(s1));
#define FMA(i) if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "s"(s1), "v"(d[(i + 1) & 7]));
#define C64(i) if (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
#define C32(i) if (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
#define CU(i)  if (OP == 5) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
#define CMP(i) if (OP == 6) asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(d[i]), "s"(s1) : "vcc");
#define CND(i) if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "vcc");
#define MN3(i) if (OP == 8) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
#define AD2(i) if (OP == 9) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define F32(i) if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
#define MOV(i) if (OP == 11) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7]));
#define CN6(i) if (OP == 12) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "s"(m64));
#define CM6(i) if (OP == 13) asm volatile("v_cmp_gt_f64_e64 %0, %1, %2" : "=s"(m64) : "v"(d[i]), "s"(s1));
#define MN2(i) if (OP == 14) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define ADU(i) if (OP == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define F32V(i) if (OP == 16) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
#define CNV(i) if (OP == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 4) & 7]));
        REP8(ADD) REP8(MUL) REP8(FMA) REP8(C64) REP8(C32) REP8(CU) REP8(CMP) REP8(CND) REP8(MN3) REP8(AD2) REP8(F32) REP8(MOV) REP8(CN6) REP8(CM6) REP8(MN2) REP8(ADU) REP8(F32V) REP8(CNV)
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += d[i] + f[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> void run(const char* name) {
    const int waves = 4, blocks = 256 * 4 * waves, iters = 20000;
    double* out; (void)hipMalloc(&out, (size_t)blocks * 64 * sizeof(double));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_op<OP><<<blocks, 64>>>(out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k_op<OP><<<blocks, 64>>>(out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = (double)iters * 8 * waves;
    printf("%-16s %.3f ms  => %.2f ns per wave-instruction per SIMD (x 2.1 GHz = %.2f cycles)\n", name, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.1);
    (void)hipFree(out);
}

int main() {
    run<0>("v_add_f64 (s)"); run<9>("v_add_f64 (v)"); run<1>("v_mul_f64"); run<2>("v_fma_f64"); run<3>("v_cvt_f64_f32"); run<4>("v_cvt_f32_f64");
    run<5>("v_cvt_f64_u32"); run<6>("v_cmp_gt_f64"); run<7>("v_cndmask_b32"); run<8>("v_min3_u32"); run<10>("v_fma_f32"); run<11>("v_mov_b64");
    run<12>("v_cndmask_e64 sgpr"); run<13>("v_cmp_f64_e64->sgpr"); run<14>("v_min_u32"); run<15>("v_add_u32"); run<16>("v_fma_f32 b"); run<17>("v_cndmask vcc b");
    return 0;
}
