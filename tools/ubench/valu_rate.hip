// Microbenchmark: sustained issue rate of v_fma_f32, v_pk_fma_f32, v_min_f32, v_cndmask on gfx950
// as a function of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 1.0000001f, c = 1e-7f;
    const f2 mm = {m, m}, cc = {c, c};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            } else if (KIND == 1) {
                p0 = __builtin_elementwise_fma(p0, mm, cc); p1 = __builtin_elementwise_fma(p1, mm, cc);
                p2 = __builtin_elementwise_fma(p2, mm, cc); p3 = __builtin_elementwise_fma(p3, mm, cc);
                p4 = __builtin_elementwise_fma(p4, mm, cc); p5 = __builtin_elementwise_fma(p5, mm, cc);
                p6 = __builtin_elementwise_fma(p6, mm, cc); p7 = __builtin_elementwise_fma(p7, mm, cc);
            } else {
                a0 = __builtin_fminf(a0, a1 + c); a1 = __builtin_fmaxf(a1, a2 - c); a2 = __builtin_fminf(a2, a3 + c); a3 = __builtin_fmaxf(a3, a4 - c);
                a4 = __builtin_fminf(a4, a5 + c); a5 = __builtin_fmaxf(a5, a6 - c); a6 = __builtin_fminf(a6, a7 + c); a7 = __builtin_fmaxf(a7, a0 - c);
            }
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if (r == 12345.678f) out[0] = r;
}

template <int KIND> double run(int waves_per_simd, int iters, float *d) {
    int cus = 256;
    // blocks of 256 threads = 4 waves = 1 per SIMD; waves_per_simd blocks per CU
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *d; hipMalloc(&d, 64);
    const int iters = 20000;
    const char *names[3] = {"v_fma_f32", "v_pk_fma_f32", "min/max+add (2 instr)"};
    for (int kind = 0; kind < 3; kind++)
        for (int w : {1, 2, 4, 8}) {
            double ms = kind == 0 ? run<0>(w, iters, d) : kind == 1 ? run<1>(w, iters, d) : run<2>(w, iters, d);
            double instr_per_wave = (double)iters * 64 * (kind == 2 ? 2 : 1);
            double cyc = ms * 1e-3 * 2.4e9;   // assumes 2.4 GHz
            printf("%-22s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instr per SIMD (at 2.4 GHz)\n", names[kind], w, ms,
                   cyc / (instr_per_wave * w));
        }
    return 0;
}
