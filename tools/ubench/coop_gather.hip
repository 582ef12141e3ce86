// Microbenchmark: what does a 64-byte record fetch per lane cost one CU of gfx950 when the four lanes of
// a quad fetch each record together?  A lane's own fetch is four 16-byte gathers whose 64 lanes touch 64
// different cache lines.  Fetched together, gather k of a quad reads the record of the quad's lane k, one
// 16-byte piece per lane (four lanes, one line), and a 4x4 transposition inside the quad (two rounds of
// v_cndmask_b32_dpp) hands every lane the four pieces of its own record.
// Every lane follows its own chain through a table of 64-byte records like a BVH walk without arithmetic.
//   mode 0  own record, four gathers per lane (what the traversal does today)
//   mode 1  quad fetch, no transposition: only the next index travels (one DPP move per gather)
//   mode 2  quad fetch + transposition, all 16 words consumed by their owner
//   mode 4, 5  as modes 1, 2 with the gathers of idle lanes' records switched off by EXEC (whole quads)
//   mode 6, 7  quad fetch straight into LDS (buffer_load_dwordx4 ... lds), every lane reads its record back from there
//   mode 3  as mode 1, but the four lanes that share a record sit in the four rows of 16 lanes
//           (lane, lane+16, lane+32, lane+48): does the address unit merge lines across rows?
// Lanes outside `lane_mask` are idle: their own fetches are switched off (mode 0) or their record index
// is out of range for the buffer descriptor, which returns zeros without touching memory (modes 1-3).
// Modes 0 and 2 must print the same checksum.
// Build: hipcc --offload-arch=gfx950 -O3 -o coop_gather coop_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define DPP_QUAD(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

__device__ __forceinline__ uint32_t quad_bcast(uint32_t v, int k) {
    switch (k) {
    case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, DPP_QUAD(0, 0, 0, 0), 0xf, 0xf, true);
    case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, DPP_QUAD(1, 1, 1, 1), 0xf, 0xf, true);
    case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, DPP_QUAD(2, 2, 2, 2), 0xf, 0xf, true);
    default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, DPP_QUAD(3, 3, 3, 3), 0xf, 0xf, true);
    }
}

// One exchange round of the transposition on one pair of registers: lanes whose `bit` of the lane number is
// clear give away `hi` and receive the partner's `lo` into it; the others give away `lo` and receive the
// partner's `hi`.  v_cndmask_b32_dpp: D = VCC ? S1 : dpp(S0).
#define XCHG(lo, hi, PERM, MASK_KEEP_LO, MASK_KEEP_HI)                                                          \
    do {                                                                                                        \
        uint32_t nlo_;                                                                                          \
        asm volatile("s_mov_b32 vcc_lo, " MASK_KEEP_LO "\n\ts_mov_b32 vcc_hi, " MASK_KEEP_LO "\n\t"             \
                     "v_cndmask_b32_dpp %0, %1, %2, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                \
                     "s_mov_b32 vcc_lo, " MASK_KEEP_HI "\n\ts_mov_b32 vcc_hi, " MASK_KEEP_HI "\n\t"             \
                     "v_cndmask_b32_dpp %1, %2, %1, vcc " PERM " row_mask:0xf bank_mask:0xf"                    \
                     : "=&v"(nlo_), "+v"(hi) : "v"(lo) : "vcc");                                                \
        lo = nlo_;                                                                                              \
    } while (0)

__device__ __forceinline__ void quad_transpose(u4_t &w0, u4_t &w1, u4_t &w2, u4_t &w3) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t a = w0[d], b = w1[d], c = w2[d], e = w3[d];
        XCHG(a, b, "quad_perm:[1,0,3,2]", "0x55555555", "0xaaaaaaaa");
        XCHG(c, e, "quad_perm:[1,0,3,2]", "0x55555555", "0xaaaaaaaa");
        XCHG(a, c, "quad_perm:[2,3,0,1]", "0x33333333", "0xcccccccc");
        XCHG(b, e, "quad_perm:[2,3,0,1]", "0x33333333", "0xcccccccc");
        w0[d] = a; w1[d] = b; w2[d] = c; w3[d] = e;
    }
}

__device__ __forceinline__ uint32_t mix(uint32_t acc, u4_t w, uint32_t pos) {
    acc = (w.x ^ (pos * 4u + 1u)) + acc; acc = (w.y ^ (pos * 4u + 2u)) + acc;
    acc = (w.z ^ (pos * 4u + 3u)) + acc; acc = (w.w ^ (pos * 4u + 4u)) + acc;
    return acc;
}

template <int MODE>
__global__ __launch_bounds__(256) void walk(const void *table, uint32_t bytes, uint32_t n_records, int iters,
                                            unsigned long long lane_mask, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, (int)bytes, 0x00020000);
    const bool active = (lane_mask >> lane) & 1ull;
    uint32_t idx = (gid * 2654435761u) % n_records;
    uint32_t acc = 0;
    if (MODE == 0) {
        if (active) {
            for (int i = 0; i < iters; i++) {
                const uint32_t off = idx * 64u;
                u4_t w0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
                u4_t w1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u, 0, 0);
                u4_t w2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32u, 0, 0);
                u4_t w3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 48u, 0, 0);
                acc = mix(mix(mix(mix(acc, w0, 0), w1, 1), w2, 2), w3, 3);
                idx = w0.x;
            }
        }
    } else if (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5) {
        const uint32_t piece = (lane & 3u) * 16u;
        for (int i = 0; i < iters; i++) {
            const uint32_t mine = active ? idx : 0x03ffffffu;       /* x 64 = beyond any table: zeros, no memory access */
            u4_t w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t rec = quad_bcast(mine, k);
                if (MODE >= 4) {         /* quads whose lane k is idle skip gather k (EXEC) instead of reading out of range */
                    w[k] = (u4_t){0u, 0u, 0u, 0u};
                    if (rec != 0x03ffffffu) w[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, rec * 64u + piece, 0, 0);
                } else
                w[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, rec * 64u + piece, 0, 0);
            }
            if (MODE == 2 || MODE == 5) {
                quad_transpose(w[0], w[1], w[2], w[3]);
                acc = mix(mix(mix(mix(acc, w[0], 0), w[1], 1), w[2], 2), w[3], 3);
                if (active) idx = w[0].x;
            } else {
                /* the next index of lane k's record is word 0 of the piece quad lane 0 fetched in gather k */
                const uint32_t n0 = quad_bcast(w[0].x, 0), n1 = quad_bcast(w[1].x, 0), n2 = quad_bcast(w[2].x, 0), n3 = quad_bcast(w[3].x, 0);
                const uint32_t q = lane & 3u;
                const uint32_t next = q == 0 ? n0 : q == 1 ? n1 : q == 2 ? n2 : n3;
                acc += w[0].y + w[1].y + w[2].y + w[3].y;
                if (active) idx = next;
            }
        }
    } else if (MODE == 6 || MODE == 7) {
        /* quad fetch straight into LDS (buffer_load_dwordx4 ... lds: lane l of gather k lands at base_k + 16 l, so the
         * record of quad g's lane k is the 64 bytes at base_k + 64 g), then every lane reads its own record back with
         * four ds_read_b128.  MODE 7 skews the four areas by 16 bytes each so that the four lanes of a quad read from
         * different banks. */
        __shared__ __attribute__((aligned(16))) uint32_t xchg[4][4 * (1024 + 64) / 4];
        const uint32_t wave = threadIdx.x >> 6;
        const uint32_t piece = (lane & 3u) * 16u;
        const uint32_t area = (MODE == 7 ? 1024u + 16u : 1024u);
        const uint32_t q = lane & 3u, g = lane >> 2;
        const u4_t *own = (const u4_t *)((const char *)xchg[wave] + q * area + g * 64u);
        for (int i = 0; i < iters; i++) {
            const uint32_t mine = active ? idx : 0x03ffffffu;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t rec = quad_bcast(mine, k);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)((char *)xchg[wave] + k * area), 16,
                                                     rec * 64u + piece, 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0070);      /* vmcnt(0): the four gathers have landed in LDS */
            u4_t w0 = own[0], w1 = own[1], w2 = own[2], w3 = own[3];
            acc = mix(mix(mix(mix(acc, w0, 0), w1, 1), w2, 2), w3, 3);
            if (active) idx = w0.x;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    } else {
        /* rows: gather k serves the lanes of row k; lane l of every row fetches piece (own row) of the record of lane 16k + l */
        const uint32_t piece = (lane >> 4) * 16u;
        for (int i = 0; i < iters; i++) {
            const uint32_t mine = active ? idx : 0x03ffffffu;
            u4_t w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t rec = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane & 15u) + 16u * k) * 4u), (int)mine);
                w[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, rec * 64u + piece, 0, 0);
            }
            /* next index of lane 16k + l: word 0 of what lane l (row 0) fetched in gather k */
            uint32_t next = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane & 15u) * 4u), (int)w[k].x);
                if ((lane >> 4) == (uint32_t)k) next = v;
            }
            acc += w[0].y + w[1].y + w[2].y + w[3].y;
            if (active) idx = next;
        }
    }
    out[gid] = acc + idx;
}

// record r: word 0 = next index (one big random cycle), the other 15 words a hash of (r, word)
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
    for (uint32_t r = 0; r < n; r++)
        for (uint32_t w = 1; w < 16; w++) t[(size_t)r * 16 + w] = (r * 16u + w) * 2246822519u;
    for (uint32_t i = 0; i < n; i++) t[(size_t)perm[i] * 16] = perm[(i + 1) % n];
}

template <int MODE>
static double run(const void *d, uint32_t n, int blocks_per_cu, int iters, unsigned long long mask, uint32_t *out, uint32_t *checksum) {
    dim3 grid(256 * blocks_per_cu), block(256);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(walk<MODE>, grid, block, 0, 0, d, n * 64u, n, 64, mask, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(walk<MODE>, grid, block, 0, 0, d, n * 64u, n, iters, mask, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    std::vector<uint32_t> h(4096);
    CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    uint32_t c = 0;
    for (size_t i = 0; i < h.size(); i++) if ((mask >> (i & 63)) & 1ull) c = c * 16777619u + h[i];
    *checksum = c;
    return ms;
}

int main() {
    const uint32_t sizes[] = {256, 8192};      // 16 KiB (vector L1), 512 KiB and 4 MiB (L2)
    uint32_t *out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    printf("%8s %5s %5s %6s | %9s %12s  %s\n", "records", "mode", "lanes", "w/SIMD", "ms", "cyc/visit/CU", "checksum");
    for (uint32_t n : sizes) {
        std::vector<uint32_t> t;
        fill(t, n);
        void *d; CHECK(hipMalloc(&d, (size_t)n * 64));
        CHECK(hipMemcpy(d, t.data(), (size_t)n * 64, hipMemcpyHostToDevice));
        for (int lanes : {64, 40, 16, -16})         /* -16: the first 16 lanes (four whole quads) instead of every fourth lane */
            for (int mode : {0, 1, 2, 4, 5, 6, 7, 3}) {
                unsigned long long mask = 0;
                for (int l = 0; l < abs(lanes); l++) mask |= 1ull << (lanes < 0 ? l : (l * 64) / lanes);
                const int iters = 4000, w = 6;
                uint32_t c = 0;
                double ms = mode == 0 ? run<0>(d, n, w, iters, mask, out, &c) : mode == 1 ? run<1>(d, n, w, iters, mask, out, &c)
                          : mode == 2 ? run<2>(d, n, w, iters, mask, out, &c) : mode == 3 ? run<3>(d, n, w, iters, mask, out, &c)
                          : mode == 4 ? run<4>(d, n, w, iters, mask, out, &c) : mode == 5 ? run<5>(d, n, w, iters, mask, out, &c)
                          : mode == 6 ? run<6>(d, n, w, iters, mask, out, &c) : run<7>(d, n, w, iters, mask, out, &c);
                double visits_per_cu = (double)iters * 4 * w;
                printf("%8u %5d %5d %6d | %9.3f %12.1f  %08x\n", n, mode, lanes, w, ms, ms * 1e3 * 2400.0 / visits_per_cu, c);
            }
        if (n == 256) {         /* latency of one visit: the same with fewer waves per SIMD */
            unsigned long long mask = 0;
            for (int l = 0; l < 40; l++) mask |= 1ull << ((l * 64) / 40);
            for (int w : {1, 2, 3, 4, 6})
                for (int mode : {0, 1, 7}) {
                    uint32_t c = 0;
                    const int iters = 4000;
                    double ms = mode == 0 ? run<0>(d, n, w, iters, mask, out, &c) : mode == 1 ? run<1>(d, n, w, iters, mask, out, &c) : run<7>(d, n, w, iters, mask, out, &c);
                    printf("%8u %5d %5d %6d | %9.3f %12.1f  latency of a visit %.0f cycles\n", n, mode, 40, w, ms, ms * 1e3 * 2400.0 / (iters * 4.0 * w),
                           ms * 1e3 * 2400.0 / iters);
                }
        }
        CHECK(hipFree(d));
    }
    return 0;
}
