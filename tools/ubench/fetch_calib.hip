// Calibration of the rocprofv3 FETCH_SIZE counter on the access patterns of the path kernel:
// every lane reads whole records at unrelated addresses of a table far larger than the caches,
// so each record read must come from memory and the true byte count is known.
//   mode 0: coalesced 16 B per lane stream (control; the counter is known to report half)
//   mode 1: 64-B records (four 16-B loads per lane), random record per lane
//   mode 2: 48-B records (three 16-B loads per lane), random record per lane
//   mode 3: 16-B records, random record per lane
// usage: fetch_calib <mode> [table MiB] [reads per lane]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void stream_kernel(const float4 *__restrict__ t, uint64_t n16, float *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (; i < n16; i += stride) { float4 v = t[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[0] = s;
}

template <int Q>   // Q 16-B pieces per record
__global__ void record_kernel(const float4 *__restrict__ t, uint32_t nrec, int reads, float *out) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    for (int k = 0; k < reads; k++) {
        uint32_t r = mix(tid * 0x9e3779b9u + (uint32_t)k * 0x85ebca6bu + 1u) % nrec;
        const float4 *p = t + (uint64_t)r * Q;
#pragma unroll
        for (int q = 0; q < Q; q++) { float4 v = p[q]; s += v.x + v.y + v.z + v.w; }
    }
    if (s == 123.456f) out[0] = s;
}

int main(int argc, char **argv) {
    int mode = argc > 1 ? atoi(argv[1]) : 0;
    uint64_t mib = argc > 2 ? strtoull(argv[2], nullptr, 10) : 4096;
    int reads = argc > 3 ? atoi(argv[3]) : 16;
    uint64_t bytes = mib << 20;
    float4 *t; float *out;
    CK(hipMalloc(&t, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(t, 0, bytes)); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int block = 256, grid = 256 * 32;
    uint64_t lanes = (uint64_t)block * grid;
    double algo = 0;
    CK(hipEventRecord(e0));
    if (mode == 0) { stream_kernel<<<grid, block>>>(t, bytes / 16, out); algo = (double)bytes; }
    else if (mode == 1) { record_kernel<4><<<grid, block>>>(t, (uint32_t)(bytes / 64), reads, out); algo = (double)lanes * reads * 64; }
    else if (mode == 2) { record_kernel<3><<<grid, block>>>(t, (uint32_t)(bytes / 48), reads, out); algo = (double)lanes * reads * 48; }
    else { record_kernel<1><<<grid, block>>>(t, (uint32_t)(bytes / 16), reads, out); algo = (double)lanes * reads * 16; }
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"mode\": %d, \"table_mib\": %llu, \"algorithmic_bytes\": %.0f, \"ms\": %.3f, \"GBps\": %.1f}\n", mode,
           (unsigned long long)mib, algo, ms, algo / ms * 1e-6);
    return 0;
}
