// Microbenchmark: how fast does one CU of gfx950 take 64-lane gather instructions of the shape the
// traversal issues?  Every lane follows its own chain through a table of 64-byte records: it loads
// the record (three 16-byte and one 8-byte buffer load, or fewer), takes the next index from the
// loaded words and goes on, like a BVH walk without the arithmetic.  Reported: wave-visits per
// microsecond per CU, cycles per visit per CU and vector-memory instructions per cycle per CU, for
// several table sizes (vector L1 / L2 / Infinity Cache resident), loads per visit, active lanes per
// wave, waves per SIMD, and lanes that share one chain (coherent rays: the same address in several lanes).
// With an argument "pmc" only five launches run (for a counter pass under rocprofv3): L2-resident table,
// four loads, 6 waves per SIMD, 64 lanes sharing chains in groups of 1, 2, 4, 16 and 64.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <cstdint>

typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// LOADS: 1 = b128 only, 2 = 2 x b128, 4 = 3 x b128 + b64 (the traversal's record fetch)
template <int LOADS>
__global__ __launch_bounds__(256) void walk(const void *table, uint32_t bytes, uint32_t n_records, int iters,
                                            unsigned long long lane_mask, uint32_t share, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, (int)bytes, 0x00020000);
    uint32_t idx = ((gid / share) * 2654435761u) % n_records;     /* `share` neighbouring lanes walk the same chain */
    uint32_t acc = 0;
    if ((lane_mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
            const uint32_t off = idx * 64u;
            u4_t w0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
            uint32_t next = w0.x;
            acc += w0.y;
            if (LOADS >= 2) { u4_t w1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u, 0, 0); acc += w1.y; next ^= w1.x; }
            if (LOADS >= 4) {
                u4_t w2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32u, 0, 0);
                u2_t w3 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 48u, 0, 0);
                acc += w2.y + w3.y; next ^= w2.x ^ w3.x;
            }
            idx = next;
        }
    }
    if (acc == 0x12345678u && idx == 0xffffffffu) out[gid] = acc;
}

// record r: word 0 = next index (a random permutation cycle), words 4, 8, 12 = 0 so that the xor of the
// four "next" words is word 0; everything else small noise
static void fill(std::vector<uint32_t> &t, uint32_t n) {
    std::vector<uint32_t> perm(n);
    for (uint32_t i = 0; i < n; i++) perm[i] = i;
    uint64_t s = 88172645463325252ull;
    for (uint32_t i = n - 1; i > 0; i--) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t j = (uint32_t)(s % (i + 1));
        uint32_t tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
    }
    t.assign((size_t)n * 16, 0);
    for (uint32_t i = 0; i < n; i++) t[(size_t)perm[i] * 16] = perm[(i + 1) % n];   // one big cycle
}

template <int LOADS>
static double run(const void *d, uint32_t n, int blocks_per_cu, int iters, unsigned long long mask, uint32_t *out, uint32_t share = 1) {
    dim3 grid(256 * blocks_per_cu), block(256);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(walk<LOADS>, grid, block, 0, 0, d, n * 64u, n, 64, mask, share, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(walk<LOADS>, grid, block, 0, 0, d, n * 64u, n, iters, mask, share, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return ms;
}

// "ceiling": the rate at which the chip takes the traversal's gathers (four per visit, L2-resident table, 6 waves per
// SIMD) as a function of the live lanes per wave and of how many of them share a record, one JSON object per line.
// lines_per_instruction = distinct records a wave's instruction touches (what the vector L1 counts as tag lookups).
static int ceiling(void) {
    uint32_t *out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    const uint32_t n = 8192;
    std::vector<uint32_t> t;
    fill(t, n);
    void *d; CHECK(hipMalloc(&d, (size_t)n * 64));
    CHECK(hipMemcpy(d, t.data(), (size_t)n * 64, hipMemcpyHostToDevice));
    const int w = 6, iters = 6000;
    for (int lanes : {16, 24, 32, 36, 40, 44, 48, 56, 64})
        for (uint32_t share : {1u, 2u, 4u, 8u}) {
            unsigned long long mask = 0;
            for (int l = 0; l < lanes; l++) mask |= 1ull << ((l * 64) / lanes);
            int lines = 0;                       /* distinct chains among the active lanes of a wave */
            for (uint32_t g = 0; g < 64; g += share) {
                bool any = false;
                for (uint32_t l = g; l < g + share && l < 64; l++) any = any || ((mask >> l) & 1ull);
                lines += any;
            }
            double best = 1e30;
            for (int rep = 0; rep < 3; rep++) { double ms = run<4>(d, n, w, iters, mask, out, share); if (ms < best) best = ms; }
            const double instr = 256.0 * 4 * w * (double)iters * 4;      /* CUs x SIMDs x waves x visits x gathers per visit */
            printf("{\"lanes\": %d, \"share\": %u, \"lines_per_instruction\": %d, \"ms\": %.4f, \"gather_instructions\": %.0f, "
                   "\"ginstr_per_s\": %.3f, \"cycles_per_instruction_per_cu_at_2400mhz\": %.2f}\n",
                   lanes, share, lines, best, instr, instr / (best * 1e6), best * 1e3 * 2400.0 * 256.0 / instr);
        }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "ceiling") return ceiling();
    if (argc > 1) {      /* counter pass: few launches, told apart by their order */
        uint32_t *out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
        const uint32_t n = 8192;
        std::vector<uint32_t> t;
        fill(t, n);
        void *d; CHECK(hipMalloc(&d, (size_t)n * 64));
        CHECK(hipMemcpy(d, t.data(), (size_t)n * 64, hipMemcpyHostToDevice));
        for (uint32_t share : {1u, 2u, 4u, 16u, 64u}) {
            double ms = run<4>(d, n, 6, 6000, ~0ull, out, share);
            printf("share %2u: %.3f ms, %.1f cycles per wave-visit per CU at 2.4 GHz (timed launch: 6000 visits x 24 waves per CU; warm-up launch: 64 visits)\n",
                   share, ms, ms * 1e3 * 2400.0 / (6000.0 * 24));
        }
        return 0;
    }
    const uint32_t sizes[] = {256, 8192, 32768, 772000, 4000000};   // 16 KiB (L1), 512 KiB, 2 MiB (L2), 49 MB (Infinity Cache), 256 MB
    uint32_t *out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    printf("%10s %5s %5s %5s | %9s %12s %12s %14s\n", "records", "loads", "lanes", "w/SIMD", "ms", "visit/us/CU", "cyc/visit/CU", "vmem instr/cyc/CU");
    for (uint32_t n : sizes) {
        std::vector<uint32_t> t;
        fill(t, n);
        void *d; CHECK(hipMalloc(&d, (size_t)n * 64));
        CHECK(hipMemcpy(d, t.data(), (size_t)n * 64, hipMemcpyHostToDevice));
        for (int loads : {4, 2, 1})
            for (int lanes : {64, 40, 16})
                for (int w : {6, 8, 2}) {
                    if ((loads != 4 && (lanes != 64 || w != 6))) continue;      // the full grid only for the traversal's shape
                    unsigned long long mask = 0;
                    for (int l = 0; l < lanes; l++) mask |= 1ull << ((l * 64) / lanes);     // spread over the wave
                    const int iters = n >= 772000 ? 2000 : 6000;
                    double ms = loads == 4 ? run<4>(d, n, w, iters, mask, out) : loads == 2 ? run<2>(d, n, w, iters, mask, out) : run<1>(d, n, w, iters, mask, out);
                    double visits_per_cu = (double)iters * 4 * w;       // wave-visits per CU
                    double us = ms * 1e3;
                    printf("%10u %5d %5d %5d | %9.3f %12.2f %12.1f %14.4f\n", n, loads, lanes, w, ms, visits_per_cu / us,
                           us * 2400.0 / visits_per_cu, visits_per_cu * loads / (us * 2400.0));
                }
        if (n == 8192)
            for (uint32_t share : {2u, 4u, 16u, 64u}) {
                double ms = run<4>(d, n, 6, 6000, ~0ull, out, share);
                double visits_per_cu = 6000.0 * 4 * 6, us = ms * 1e3;
                printf("%10u %5d %5s %5d | %9.3f %12.2f %12.1f %14.4f   (%u lanes share a chain)\n", n, 4, "64", 6, ms, visits_per_cu / us,
                       us * 2400.0 / visits_per_cu, visits_per_cu * 4 / (us * 2400.0), share);
            }
        CHECK(hipFree(d));
    }
    return 0;
}
