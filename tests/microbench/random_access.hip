// tests/microbench/random_access.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS path's access pattern:
// dependent-free random 8-byte reads (the BT4 descent's pair reads) and random 4-byte stores (its link stores) in a 4 GiB
// buffer.  MEASUREMENT TOOL ONLY (not part of the library).  Run under rocprofv3 --pmc FETCH_SIZE and again --pmc WRITE_SIZE:
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/random_access tests/microbench/random_access.hip
//     rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- /tmp/random_access
// It prints how many accesses each kernel made; profiles/rNN_pmc_calibration.json relates the counters to N x 64 B
// (one 64-byte line per access) and to N x 8 / N x 4 (the bytes asked for).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void random_reads8(const unsigned long long *buf, unsigned long long mask, unsigned long long per_thread, unsigned long long *sink)
{
    unsigned long long x = (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
    unsigned long long acc = 0;
    for (unsigned long long k = 0; k < per_thread; k++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        acc += buf[(x >> 20) & mask];
    }
    if (acc == 0x1234567) sink[0] = acc;            // (keeps the loads)
}
__global__ void random_writes4(uint32_t *buf, unsigned long long mask, unsigned long long per_thread)
{
    unsigned long long x = (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 7;
    for (unsigned long long k = 0; k < per_thread; k++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        buf[(x >> 20) & mask] = (uint32_t)x;
    }
}
int main()
{
    const unsigned long long bytes = 4ull << 30;
    void *buf = nullptr; unsigned long long *sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **)&sink, 8) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes);
    const unsigned blocks = 1024, threads = 256;
    const unsigned long long per_thread = 256;
    hipLaunchKernelGGL(random_reads8, dim3(blocks), dim3(threads), 0, 0, (const unsigned long long *)buf, (bytes / 8) - 1, per_thread, sink);
    hipLaunchKernelGGL(random_writes4, dim3(blocks), dim3(threads), 0, 0, (uint32_t *)buf, (bytes / 4) - 1, per_thread);
    (void)hipDeviceSynchronize();
    printf("{\"accesses_per_kernel\": %llu, \"buffer_bytes\": %llu}\n", (unsigned long long)blocks * threads * per_thread, bytes);
    return 0;
}
