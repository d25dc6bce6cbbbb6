// Stand-alone micro-benchmark of factor16 (gp_amd/csrc/factor16.h): one wave, REPS factorisations
// of the same SPD tile; prints cycles per call (s_memtime) and checks L L^T.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#include "../gp_amd/csrc/factor16.h"

__global__ __launch_bounds__(512) void k(const double *in, double *out, unsigned long long *cyc, int reps)
{
    __shared__ double s_inv[256];
    __shared__ double s_d16[16][17];
    const int lane = threadIdx.x;
    int bad = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < 4; ++i) s_d16[lane & 15][(lane >> 4) + 4 * i] = in[(lane & 15) * 16 + (lane >> 4) + 4 * i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bad += factor16(s_d16, s_inv, lane);
        __builtin_amdgcn_wave_barrier();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 4; ++i) out[(lane & 15) * 16 + (lane >> 4) + 4 * i] = s_d16[lane & 15][(lane >> 4) + 4 * i];
    for (int i = 0; i < 4; ++i) out[256 + i * 64 + lane] = s_inv[i * 64 + lane];
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = bad; }
}

int main()
{
    std::vector<double> A(256), out(512);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) A[i * 16 + j] = std::exp(-0.05 * (i - j) * (i - j)) + (i == j ? 0.1 : 0.0);
    double *din, *dout; unsigned long long *dc, hc[2];
    hipMalloc(&din, 256 * 8); hipMalloc(&dout, 512 * 8); hipMalloc(&dc, 16);
    hipMemcpy(din, A.data(), 256 * 8, hipMemcpyHostToDevice);
    const int reps = 1000;
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dc, reps);
        hipDeviceSynchronize();
    }
    hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    hipMemcpy(out.data(), dout, 512 * 8, hipMemcpyDeviceToHost);
    double err = 0, ierr = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0, t = 0;
            for (int k = 0; k < 16; ++k) s += out[i * 16 + k] * out[j * 16 + k];
            err = std::fmax(err, std::fabs(s - A[i * 16 + j]));
            // Linv in A-operand order: s_inv[kg*64 + l] = Linv[l&15][(l>>4)+4kg]
            for (int k = 0; k < 16; ++k) t += out[256 + (k >> 2) * 64 + i + 16 * (k & 3)] * out[k * 16 + j];
            ierr = std::fmax(ierr, std::fabs(t - (i == j ? 1.0 : 0.0)));
        }
    printf("factor16: %.0f cycles/call (s_memtime ticks), bad=%llu, |LL^T-A|=%.2e |Linv L - I|=%.2e\n",
           (double)hc[0] / reps, hc[1], err, ierr);
    return 0;
}
