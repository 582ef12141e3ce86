// Microbenchmark: issue cost of single VALU instructions on gfx950 (cycles per wave instruction per SIMD at
// 8 waves per SIMD), each measured as 64 copies per loop pass on eight registers through inline assembly.
// Written to find out which encodings of compare and select are cheap: v_cndmask_b32 in its VOP2 encoding
// (mask in VCC) turned out to cost about ten plain instructions, its VOP3 encoding two.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_ops valu_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0,1) X(1,2) X(2,3) X(3,4) X(4,5) X(5,6) X(6,7) X(7,0)
#define OPS(X) asm volatile(REP8(X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(msk), "v"(lanebits) : "vcc")

#define I_MOV(d, s)        "v_mov_b32 %" #d ", %" #s "\n\t"
#define I_CND_E32(d, s)    "v_cndmask_b32_e32 %" #d ", %" #s ", %" #d ", vcc\n\t"
#define I_CND_E64(d, s)    "v_cndmask_b32_e64 %" #d ", %" #s ", %" #d ", %8\n\t"
#define I_CND_E64V(d, s)   "v_cndmask_b32_e64 %" #d ", %" #s ", %" #d ", vcc\n\t"
#define I_CMP_E32(d, s)    "v_cmp_lt_f32_e32 vcc, %" #d ", %" #s "\n\t"
#define I_CMP_E64(d, s)    "v_cmp_lt_f32_e64 s[20:21], %" #d ", %" #s "\n\t"
#define I_CMP_E64V(d, s)   "v_cmp_lt_f32_e64 vcc, %" #d ", %" #s "\n\t"
#define I_CMPCND64(d, s)   "v_cmp_lt_f32_e64 s[20:21], %" #d ", %" #s "\n\tv_cndmask_b32_e64 %" #d ", %" #s ", %" #d ", s[20:21]\n\t"
#define I_CMPCND32(d, s)   "v_cmp_lt_f32_e32 vcc, %" #d ", %" #s "\n\tv_cndmask_b32_e32 %" #d ", %" #s ", %" #d ", vcc\n\t"
#define I_MIN(d, s)        "v_min_f32 %" #d ", %" #s ", %" #d "\n\t"
#define I_MAX3(d, s)       "v_max3_f32 %" #d ", %" #s ", %" #d ", %" #s "\n\t"
#define I_BFI(d, s)        "v_bfi_b32 %" #d ", %9, %" #s ", %" #d "\n\t"
#define I_XOR3(d, s)       "v_xad_u32 %" #d ", %" #s ", %" #d ", %9\n\t"
#define I_ANDOR(d, s)      "v_and_or_b32 %" #d ", %" #s ", %9, %" #d "\n\t"
#define I_ADDU(d, s)       "v_add_u32 %" #d ", %" #s ", %" #d "\n\t"
#define I_LSHLADD(d, s)    "v_lshl_add_u32 %" #d ", %" #s ", 6, %" #d "\n\t"
#define I_AND_DPP(d, s)    "v_and_b32_dpp %" #d ", %" #s ", %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_MOV_DPP(d, s)    "v_mov_b32_dpp %" #d ", %" #s " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_ADD_DPP(d, s)    "v_add_u32_dpp %" #d ", %" #s ", %" #d " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_ADDC(d, s)       "v_addc_co_u32_e32 %" #d ", vcc, %" #s ", %" #d ", vcc\n\t"
#define I_FMA(d, s)        "v_fma_f32 %" #d ", %" #s ", %" #d ", %" #s "\n\t"
#define I_PKFMA(d, s)      "v_pk_mul_f32 v[40:41], v[42:43], v[44:45]\n\t"
#define I_MUL_E32(d, s)    "v_mul_f32_e32 %" #d ", %" #s ", %" #d "\n\t"
#define I_CMPX(d, s)       "v_cmp_class_f32_e64 s[20:21], %" #d ", %9\n\t"
#define I_SWAP(d, s)       "v_swap_b32 %" #d ", %" #s "\n\t"

template <int KIND>
__global__ void k(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned long long msk = 0x5555555555555555ull;
    const unsigned lanebits = (threadIdx.x & 1) ? 0xffffffffu : 0u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) OPS(I_MOV);
            if (KIND == 1) OPS(I_CND_E32);
            if (KIND == 2) OPS(I_CND_E64);
            if (KIND == 3) OPS(I_CND_E64V);
            if (KIND == 4) OPS(I_CMP_E32);
            if (KIND == 5) OPS(I_CMP_E64);
            if (KIND == 6) OPS(I_CMP_E64V);
            if (KIND == 7) OPS(I_CMPCND64);
            if (KIND == 8) OPS(I_CMPCND32);
            if (KIND == 9) OPS(I_MIN);
            if (KIND == 10) OPS(I_MAX3);
            if (KIND == 11) OPS(I_BFI);
            if (KIND == 12) OPS(I_XOR3);
            if (KIND == 13) OPS(I_ANDOR);
            if (KIND == 14) OPS(I_ADDU);
            if (KIND == 15) OPS(I_LSHLADD);
            if (KIND == 16) OPS(I_AND_DPP);
            if (KIND == 17) OPS(I_MOV_DPP);
            if (KIND == 18) OPS(I_ADD_DPP);
            if (KIND == 19) OPS(I_ADDC);
            if (KIND == 20) OPS(I_FMA);
            if (KIND == 21) OPS(I_MUL_E32);
            if (KIND == 22) OPS(I_CMPX);
            if (KIND == 23) OPS(I_SWAP);
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r == 12345.678f) out[0] = r;
}

template <int KIND> static double run(int waves_per_simd, int iters, float *d) {
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, 10, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int KIND> static void report(const char *name, int per_slot, float *d) {
    const int iters = 5000, w = 8;
    double ms = run<KIND>(w, iters, d);
    double slots = (double)iters * 64 * w;
    printf("%-44s %7.2f cycles per instruction per SIMD (8 waves per SIMD, 2.4 GHz assumed)\n", name, ms * 1e-3 * 2.4e9 / (slots * per_slot));
}

int main() {
    float *d; (void)hipMalloc(&d, 64);
    report<0>("v_mov_b32", 1, d);
    report<20>("v_fma_f32", 1, d);
    report<21>("v_mul_f32_e32", 1, d);
    report<9>("v_min_f32", 1, d);
    report<10>("v_max3_f32", 1, d);
    report<14>("v_add_u32", 1, d);
    report<15>("v_lshl_add_u32", 1, d);
    report<11>("v_bfi_b32", 1, d);
    report<12>("v_xad_u32", 1, d);
    report<13>("v_and_or_b32", 1, d);
    report<23>("v_swap_b32", 1, d);
    report<1>("v_cndmask_b32_e32 (VOP2, vcc)", 1, d);
    report<2>("v_cndmask_b32_e64 (VOP3, sgpr pair)", 1, d);
    report<3>("v_cndmask_b32_e64 (VOP3, vcc)", 1, d);
    report<4>("v_cmp_lt_f32_e32 (VOPC, writes vcc)", 1, d);
    report<5>("v_cmp_lt_f32_e64 (VOP3, writes sgpr pair)", 1, d);
    report<6>("v_cmp_lt_f32_e64 (VOP3, writes vcc)", 1, d);
    report<22>("v_cmp_class_f32_e64", 1, d);
    report<7>("v_cmp_e64 + v_cndmask_e64 (pair)", 2, d);
    report<8>("v_cmp_e32 + v_cndmask_e32 (pair)", 2, d);
    report<19>("v_addc_co_u32_e32 (VOP2, reads+writes vcc)", 1, d);
    report<17>("v_mov_b32_dpp quad_perm", 1, d);
    report<16>("v_and_b32_dpp quad_perm", 1, d);
    report<18>("v_add_u32_dpp quad_perm", 1, d);
    return 0;
}
