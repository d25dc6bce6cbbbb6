// Stream-concurrency probe: which pairs of HIP streams dispatch kernels concurrently?
// For every ordered pair (i, j): a long many-round kernel on stream i, then a one-workgroup
// kernel on stream j; the time until the small one completes tells whether j's packet was
// processed while i's kernel was still dispatching workgroups (concurrent) or only after it.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/qprobe tools/qprobe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void k_big(int *sink, int spin)
{
    __shared__ int pad[18 * 1024];  // 72 KiB: two workgroups per CU, like the trailing update
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    int acc = 0;
    for (int i = 0; i < spin; ++i) {
        __builtin_amdgcn_s_sleep(100);
        acc += pad[(threadIdx.x + i) & 255];
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}
__global__ void k_small(int *sink)
{
    if (sink[1] == 0x7fffffff) sink[2] = 1;
}

int main(int argc, char **argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 8;
    const int kind = argc > 2 ? atoi(argv[2]) : 0;  // 0 plain non-blocking, 1 CU-mask API (full mask)
    int *sink;
    hipMalloc(&sink, 64);
    hipMemset(sink, 0, 64);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    std::vector<hipStream_t> st(S);
    for (int i = 0; i < S; ++i) {
        if (kind == 1) {
            std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xFFFFFFFFu);
            hipExtStreamCreateWithCUMask(&st[i], (uint32_t)mask.size(), mask.data());
        } else {
            hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        }
    }
    for (int i = 0; i < S; ++i) {  // first use (queues are bound lazily)
        hipLaunchKernelGGL(k_small, dim3(1), 64, 0, st[i], sink);
        hipStreamSynchronize(st[i]);
    }
    const int spin = 40;
    printf("rows: stream of the long kernel; columns: stream of the small kernel; us until the small one is done\n");
    for (int i = 0; i < S; ++i) {
        for (int j = 0; j < S; ++j) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_big, dim3(512 * 8), 256, 0, st[i], sink, spin);
            hipLaunchKernelGGL(k_small, dim3(1), 64, 0, st[j], sink);
            hipStreamSynchronize(st[j]);
            auto t1 = std::chrono::steady_clock::now();
            hipDeviceSynchronize();
            auto t2 = std::chrono::steady_clock::now();
            printf("%6.0f/%-6.0f", std::chrono::duration<double, std::micro>(t1 - t0).count(),
                   std::chrono::duration<double, std::micro>(t2 - t0).count());
        }
        printf("\n");
    }
    return 0;
}
