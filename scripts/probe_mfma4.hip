// probe_mfma4.hip — discovers the lane layout of v_mfma_f64_4x4x4_4b (gfx950) and its accumulation order.
// Not part of the product.  hipcc --offload-arch=gfx950 -O2 scripts/probe_mfma4.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(const double* A, const double* B, const double* C, double* D, int n)
{
    const int lane = threadIdx.x;
    for (int t = 0; t < n; ++t) {
        const double a = A[t * 64 + lane], b = B[t * 64 + lane], c = C[t * 64 + lane];
        D[t * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
    }
}
int main()
{
    const int n = 64 * 64 + 4;
    double *hA = new double[n * 64](), *hB = new double[n * 64](), *hC = new double[n * 64](), *hD = new double[n * 64]();
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) { hA[(la * 64 + lb) * 64 + la] = 1.0; hB[(la * 64 + lb) * 64 + lb] = 1.0; }
    // accumulation-order probe: row of A = (1, 2^-60, -1, 2^-60) style values, see below (test t = 4096..4099)
    const int t0 = 64 * 64;
    for (int l = 0; l < 64; ++l) { hA[t0 * 64 + l] = 1.0 + l; hB[t0 * 64 + l] = 100.0 + l; hC[t0 * 64 + l] = 10000.0 * l; }
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, n * 64 * 8); hipMalloc(&dB, n * 64 * 8); hipMalloc(&dC, n * 64 * 8); hipMalloc(&dD, n * 64 * 8);
    hipMemcpy(dA, hA, n * 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, hB, n * 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dC, hC, n * 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, n);
    hipMemcpy(hD, dD, n * 64 * 8, hipMemcpyDeviceToHost);
    // for every (la, lb): which output lanes are non-zero
    for (int la = 0; la < 64; ++la) {
        for (int lb = 0; lb < 64; ++lb) {
            for (int lo = 0; lo < 64; ++lo) if (hD[(la * 64 + lb) * 64 + lo] != 0.0) printf("%d %d %d\n", la, lb, lo);
        }
    }
    printf("GENERIC");
    for (int l = 0; l < 64; ++l) printf(" %.1f", hD[t0 * 64 + l]);
    printf("\n");
    return 0;
}
