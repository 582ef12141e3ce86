// Microbenchmark: sustained issue rate of v_fma_f32, v_pk_fma_f32, v_min_f32, v_cndmask (plain and DPP),
// v_mov_b32_dpp, v_permlane32_swap_b32 and ds_bpermute_b32 on gfx950
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
            } else if (KIND == 3) {          /* v_cndmask_b32_dpp with a quad permutation (the in-quad exchange of the record fetch) */
                asm volatile("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %1, %2, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %3, %4, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %5, %6, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %6, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %7, %0, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (KIND == 4) {          /* v_mov_b32_dpp */
                asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 5) {          /* v_permlane32_swap_b32: upper half of the first register <-> lower half of the second */
                asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
                             "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                             "v_permlane32_swap_b32 %1, %2\n\tv_permlane32_swap_b32 %3, %4\n\t"
                             "v_permlane32_swap_b32 %5, %6\n\tv_permlane32_swap_b32 %7, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 6) {          /* plain v_cndmask_b32 */
                asm volatile("v_cndmask_b32 %0, %1, %0, vcc\n\tv_cndmask_b32 %1, %2, %1, vcc\n\t"
                             "v_cndmask_b32 %2, %3, %2, vcc\n\tv_cndmask_b32 %3, %4, %3, vcc\n\t"
                             "v_cndmask_b32 %4, %5, %4, vcc\n\tv_cndmask_b32 %5, %6, %5, vcc\n\t"
                             "v_cndmask_b32 %6, %7, %6, vcc\n\tv_cndmask_b32 %7, %0, %7, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (KIND == 9) {          /* plain v_mov_b32 through inline asm (checks the asm harness itself) */
                asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %4\n\t"
                             "v_mov_b32 %4, %5\n\tv_mov_b32 %5, %6\n\tv_mov_b32 %6, %7\n\tv_mov_b32 %7, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 10) {         /* v_cndmask_b32 with the mask in an SGPR pair (VOP3) */
                const unsigned long long msk = 0x5555555555555555ull;
                asm volatile("v_cndmask_b32_e64 %0, %1, %0, %8\n\tv_cndmask_b32_e64 %1, %2, %1, %8\n\t"
                             "v_cndmask_b32_e64 %2, %3, %2, %8\n\tv_cndmask_b32_e64 %3, %4, %3, %8\n\t"
                             "v_cndmask_b32_e64 %4, %5, %4, %8\n\tv_cndmask_b32_e64 %5, %6, %5, %8\n\t"
                             "v_cndmask_b32_e64 %6, %7, %6, %8\n\tv_cndmask_b32_e64 %7, %0, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(msk));
            } else if (KIND == 11) {         /* integer a = lane-select written in C (what the compiler makes of it) */
                const bool odd = threadIdx.x & 1;
                float t0 = odd ? a1 : a0, t1 = odd ? a2 : a1, t2 = odd ? a3 : a2, t3 = odd ? a4 : a3;
                float t4 = odd ? a5 : a4, t5 = odd ? a6 : a5, t6 = odd ? a7 : a6, t7 = odd ? a0 : a7;
                a0 = t0; a1 = t1; a2 = t2; a3 = t3; a4 = t4; a5 = t5; a6 = t6; a7 = t7;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 12) {         /* VOP3 encoding, mask in VCC */
                asm volatile("v_cndmask_b32_e64 %0, %1, %0, vcc\n\tv_cndmask_b32_e64 %1, %2, %1, vcc\n\t"
                             "v_cndmask_b32_e64 %2, %3, %2, vcc\n\tv_cndmask_b32_e64 %3, %4, %3, vcc\n\t"
                             "v_cndmask_b32_e64 %4, %5, %4, vcc\n\tv_cndmask_b32_e64 %5, %6, %5, vcc\n\t"
                             "v_cndmask_b32_e64 %6, %7, %6, vcc\n\tv_cndmask_b32_e64 %7, %0, %7, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (KIND == 13) {         /* VOP2 encoding, VCC written by the scalar unit just before */
                asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\t"
                             "v_cndmask_b32 %0, %1, %0, vcc\n\tv_cndmask_b32 %1, %2, %1, vcc\n\t"
                             "v_cndmask_b32 %2, %3, %2, vcc\n\tv_cndmask_b32 %3, %4, %3, vcc\n\t"
                             "v_cndmask_b32 %4, %5, %4, vcc\n\tv_cndmask_b32 %5, %6, %5, vcc\n\t"
                             "v_cndmask_b32 %6, %7, %6, vcc\n\tv_cndmask_b32 %7, %0, %7, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (KIND == 14) {         /* v_cndmask_b32_dpp, VCC written by the scalar unit just before */
                asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\t"
                             "v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %1, %2, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %3, %4, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %5, %6, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %6, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_cndmask_b32_dpp %7, %0, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (KIND == 15) {         /* integer values instead of floats that may be denormal: v_cndmask VOP2 */
                asm volatile("v_cndmask_b32 %0, %1, %0, vcc\n\tv_cndmask_b32 %1, %2, %1, vcc\n\t"
                             "v_cndmask_b32 %2, %3, %2, vcc\n\tv_cndmask_b32 %3, %4, %3, vcc\n\t"
                             "v_cndmask_b32 %4, %5, %4, vcc\n\tv_cndmask_b32 %5, %6, %5, vcc\n\t"
                             "v_cndmask_b32 %6, %7, %6, vcc\n\tv_cndmask_b32 %7, %0, %7, vcc\n\t"
                             "v_add_f32 %0, %0, %1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (KIND == 16 || KIND == 17) {         /* eight selects into registers that nothing in the block reads */
                float d0, d1, d2, d3, d4, d5, d6, d7;
                if (KIND == 16)
                    asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %1, %9, %10, vcc\n\t"
                                 "v_cndmask_b32 %2, %10, %11, vcc\n\tv_cndmask_b32 %3, %11, %12, vcc\n\t"
                                 "v_cndmask_b32 %4, %12, %13, vcc\n\tv_cndmask_b32 %5, %13, %14, vcc\n\t"
                                 "v_cndmask_b32 %6, %14, %15, vcc\n\tv_cndmask_b32 %7, %15, %8, vcc"
                                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
                else
                    asm volatile("v_cndmask_b32_e64 %0, %8, %9, vcc\n\tv_cndmask_b32_e64 %1, %9, %10, vcc\n\t"
                                 "v_cndmask_b32_e64 %2, %10, %11, vcc\n\tv_cndmask_b32_e64 %3, %11, %12, vcc\n\t"
                                 "v_cndmask_b32_e64 %4, %12, %13, vcc\n\tv_cndmask_b32_e64 %5, %13, %14, vcc\n\t"
                                 "v_cndmask_b32_e64 %6, %14, %15, vcc\n\tv_cndmask_b32_e64 %7, %15, %8, vcc"
                                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
                a0 = d0; a1 = d1; a2 = d2; a3 = d3; a4 = d4; a5 = d5; a6 = d6; a7 = d7;
            } else if (KIND == 18) {         /* the chain pattern of the select tests with v_add_f32 (VOP2) */
                asm volatile("v_add_f32 %0, %1, %0\n\tv_add_f32 %1, %2, %1\n\tv_add_f32 %2, %3, %2\n\tv_add_f32 %3, %4, %3\n\t"
                             "v_add_f32 %4, %5, %4\n\tv_add_f32 %5, %6, %5\n\tv_add_f32 %6, %7, %6\n\tv_add_f32 %7, %0, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 19) {         /* v_cndmask VOP2 on integer-looking values (no denormal / NaN bit patterns) */
                asm volatile("v_cndmask_b32 %0, %1, %0, vcc\n\tv_cndmask_b32 %1, %2, %1, vcc\n\t"
                             "v_cndmask_b32 %2, %3, %2, vcc\n\tv_cndmask_b32 %3, %4, %3, vcc\n\t"
                             "v_cndmask_b32 %4, %5, %4, vcc\n\tv_cndmask_b32 %5, %6, %5, vcc\n\t"
                             "v_cndmask_b32 %6, %7, %6, vcc\n\tv_cndmask_b32 %7, %0, %7, vcc"
                             : "+v"(p0.x), "+v"(p0.y), "+v"(p1.x), "+v"(p1.y), "+v"(p2.x), "+v"(p2.y), "+v"(p3.x), "+v"(p3.y) : : "vcc");
            } else if (KIND == 7) {          /* ds_bpermute_b32 */
                const int addr = ((threadIdx.x ^ 1) & 63) * 4;
                a0 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a0))); a1 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a1)));
                a2 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a2))); a3 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a3)));
                a4 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a4))); a5 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a5)));
                a6 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a6))); a7 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a7)));
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
    const char *names[19] = {"v_fma_f32", "v_pk_fma_f32", "min/max+add (2 instr)", "v_cndmask_b32_dpp", "v_mov_b32_dpp", "v_permlane32_swap_b32", "v_cndmask_b32", "ds_bpermute_b32", "v_mov_b32 (asm)", "v_cndmask_b32_e64 sgpr", "select in C", "v_cndmask_b32_e64 vcc", "s_mov vcc + 8 v_cndmask_b32", "s_mov vcc + 8 v_cndmask_b32_dpp", "8 v_cndmask_b32 + v_add_f32", "v_cndmask_b32 fresh dst", "v_cndmask_b32_e64 fresh dst", "v_add_f32 chain (asm)", "v_cndmask_b32 other regs"};
    for (int kind = 0; kind < 19; kind++)
        for (int w : {1, 2, 4, 8}) {
            if (kind >= 3 && w != 1 && w != 8) continue;
            double ms = kind == 0 ? run<0>(w, iters, d) : kind == 1 ? run<1>(w, iters, d) : kind == 2 ? run<8>(w, iters, d) : kind == 3 ? run<3>(w, iters, d)
                      : kind == 4 ? run<4>(w, iters, d) : kind == 5 ? run<5>(w, iters, d) : kind == 6 ? run<6>(w, iters, d) : kind == 7 ? run<7>(w, iters, d) : kind == 8 ? run<9>(w, iters, d) : kind == 9 ? run<10>(w, iters, d) : kind == 10 ? run<11>(w, iters, d) : kind == 11 ? run<12>(w, iters, d) : kind == 12 ? run<13>(w, iters, d) : kind == 13 ? run<14>(w, iters, d) : kind == 14 ? run<15>(w, iters, d) : kind == 15 ? run<16>(w, iters, d) : kind == 16 ? run<17>(w, iters, d) : kind == 17 ? run<18>(w, iters, d) : run<19>(w, iters, d);
            double instr_per_wave = (double)iters * 64 * (kind == 2 ? 2 : 1);
            double cyc = ms * 1e-3 * 2.4e9;   // assumes 2.4 GHz
            printf("%-22s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instr per SIMD (at 2.4 GHz)\n", names[kind], w, ms,
                   cyc / (instr_per_wave * w));
        }
    return 0;
}
