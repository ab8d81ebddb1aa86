// ubench.hip — per-primitive VALU cost on gfx950 (issue cycles per wave-instruction sequence).
// Not part of the product: a measuring tool.  Build+run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../klara.jl_amd/csrc/klara_kernels.h"

#define ITERS 2000
template <int OP>
__global__ __launch_bounds__(256) void k_op(double* out, double seedv, int G)
{
    kd_tables_to_lds();
    const int lane = threadIdx.x & 63;
    double a = seedv + 1e-3 * lane, b = 0.5 + 1e-4 * lane, acc = 0.0;
    uint32_t c0 = threadIdx.x, c1 = blockIdx.x;
    for (int i = 0; i < ITERS; ++i) {
        if (OP == 0) { kd_u32x4 r = kd_philox4x32_10(c0, c1, 7u, i, 1u, 2u); c0 = r.x; c1 = r.y ^ r.z ^ r.w; }
        else if (OP == 1) { a = kd_log(a) + 3.0; }
        else if (OP == 2) { a = kd_exp(a * 1e-3) + 0.1; }
        else if (OP == 3) { double s, c; kd_sincos2pi(b, &s, &c); b = 0.5 + 0.25 * s * c; }
        else if (OP == 4) { a = __builtin_sqrt(a) + 2.0; }
        else if (OP == 5) { a = 3.0 / a + 1.0; }
        else if (OP == 6) { kd_u32x4 r = kd_philox4x32_10(c0, c1, 7u, i, 1u, 2u); double z0, z1, u1_, lg_; kd_normal_pair_w(r.x, r.y, &z0, &z1, &u1_, &lg_); acc += z0 * z1; c0 = r.x; c1 = r.y; }
        else if (OP == 7) { double v[3] = { a, b, acc }; group_allreduce<3>(v, G, lane); a = v[0] * 1e-2 + 1.0; b = v[1] * 1e-3 + 0.5; acc = v[2] * 1e-3; }
        else if (OP == 8) { a = kd_fma(a, 0.999, 0.001); }
        else if (OP == 9) { c0 = c0 * 0x9E3779B9u + c1; }   // v_mul_lo_u32 chain
        else if (OP == 10) { uint64_t p = (uint64_t)c0 * 0xD2511F53u; c0 = (uint32_t)(p >> 32) ^ (uint32_t)p ^ c1; }  // mad_u64_u32
        else if (OP == 11) { double u = kd_u52(c0, c1); a += u; c0 += 77; }
        else if (OP == 12) { a = kd_log_u01(a) + 3.0; }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a + b + acc + c0 + c1;
}

// issue cost of single instructions: 8 independent chains per wavefront (throughput, not latency), 4 wavefronts per SIMD
template <int OP>
__global__ __launch_bounds__(256) void k_tp(double* out, double seedv)
{
    const int lane = threadIdx.x & 63;
    double a[8]; uint32_t u[8]; uint64_t w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = seedv + 1e-3 * lane + k; u[k] = threadIdx.x * 7u + k; w[k] = (uint64_t)threadIdx.x * 0x9E3779B97F4A7C15ull + k; }
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (OP == 0) a[k] = kd_fma(a[k], 0.999, 0.001);
            else if (OP == 1) a[k] = a[k] + 1.25;
            else if (OP == 2) a[k] = a[k] * 1.0000001;
            else if (OP == 3) { const uint64_t p = (uint64_t)u[k] * 0xD2511F53u + w[k]; w[k] = p; u[k] = (uint32_t)(p >> 32); }     // v_mad_u64_u32
            else if (OP == 4) u[k] = u[k] * 0x9E3779B9u;                                                                        // v_mul_lo_u32
            else if (OP == 5) u[k] = __umulhi(u[k], 0xD2511F53u) + 1u;                                                          // v_mul_hi_u32 (+ add)
            else if (OP == 6) u[k] = KD_XOR3(u[k], 0x12345u, (uint32_t)i);                                                      // v_bitop3
            else if (OP == 7) a[k] = __builtin_amdgcn_rsq(a[k]) + 2.0;                                                          // v_rsq_f64 (+ add)
            else if (OP == 8) a[k] = __builtin_amdgcn_rcp(a[k]) + 2.0;                                                          // v_rcp_f64 (+ add)
            else if (OP == 9) a[k] = (double)(int)(a[k]) + 1.5;                                                                 // cvt_i32_f64 + cvt_f64_i32 (+ add)
            else if (OP == 10) a[k] = dpp_mov<0xB1>(a[k]) + 1.0;                                                                // 2 v_mov_dpp (+ add)
            else if (OP == 11) a[k] = a[k] > 3.0 ? a[k] - 1.0 : a[k] + 0.5;                                                      // cmp + 2 cndmask + 2 add
        }
    }
    double s = 0.0; uint32_t t = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += a[k]; t += u[k] + (uint32_t)w[k]; }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s + t;
}
template <int OP>
static void run_tp(const char* name)
{
    double* d; hipMalloc(&d, sizeof(double) * 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;
    hipLaunchKernelGGL(k_tp<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_tp<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %6.2f cycles per wave-instruction group at 2.4 GHz (8 independent chains x 4 waves/SIMD, %.3f ms)\n", name,
           ms * 1e-3 * 2.4e9 / (4.0 * ITERS * 8.0), ms);
    hipFree(d);
}

template <int OP>
static void run(const char* name, int G = 64)
{
    double* d; hipMalloc(&d, sizeof(double) * 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;   // 4 blocks of 4 waves per CU -> 4 waves per SIMD
    hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5, G);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5, G);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // each SIMD ran 4 waves x ITERS ops back to back; assume 2.4 GHz
    const double cyc = ms * 1e-3 * 2.4e9 / (4.0 * ITERS);
    printf("%-28s %8.1f cycles per wave-op (4 waves/SIMD, %.3f ms)\n", name, cyc, ms);
    hipFree(d);
}

typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_mfma_peak(double* out, int iters)
{
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-3, b = 0.5 - lane * 1e-3;
    d4 acc[7];
    for (int t = 0; t < 7; ++t) acc[t] = (d4){ 0.0, 0.0, 0.0, 0.0 };
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 7; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0.0;
    for (int t = 0; t < 7; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static void mfma_peak()
{
    double* d; hipMalloc(&d, sizeof(double) * 512 * 512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k_mfma_peak, dim3(256), dim3(512), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma_peak, dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 8 * iters * 7 * 2048.0;
    printf("v_mfma_f64_16x16x4_f64 sustained: %.1f TFLOP/s (2 waves/SIMD, 7 accumulators, %.2f ms) -> %.1f cycles/MFMA at 2.4 GHz\n",
           flops / (ms * 1e-3) / 1e12, ms, ms * 1e-3 * 2.4e9 / (2.0 * iters * 7));
    hipFree(d);
}

// Do vector instructions of one wavefront issue while another wavefront of the same SIMD keeps the FP64 matrix pipe busy?  One workgroup
// of 8 wavefronts per CU (2 per SIMD): wavefronts 0-3 run a chain-free MFMA loop, wavefronts 4-7 a vector loop (KIND 0: v_fma_f64,
// 1: v_mad_u64_u32 + v_bitop3, 2: Philox + Box-Muller pairs); each half alone, then both together.
template <int KIND>
__global__ __launch_bounds__(512) void k_overlap(double* out, int mfma_iters, int valu_iters)
{
    kd_tables_to_lds();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0;
    if (wave < 4) {
        double a = 1.0 + lane * 1e-3, b = 0.5 - lane * 1e-3;
        d4 acc[7];
        for (int t = 0; t < 7; ++t) acc[t] = (d4){ 0.0, 0.0, 0.0, 0.0 };
        for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
            for (int t = 0; t < 7; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 7; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    } else {
        double a[8]; uint32_t u[8]; uint64_t w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] = 1.0 + 1e-3 * lane + k; u[k] = threadIdx.x * 7u + k; w[k] = (uint64_t)threadIdx.x * 0x9E3779B97F4A7C15ull + k; }
        for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (KIND == 0) a[k] = kd_fma(a[k], 0.999, 0.001);
                else if (KIND == 1) { const uint64_t p = (uint64_t)u[k] * 0xD2511F53u + w[k]; w[k] = p; u[k] = KD_XOR3((uint32_t)(p >> 32), (uint32_t)p, (uint32_t)i); }
                else { double z0, z1, u1_, lg_; const kd_u32x4 b_ = kd_stream_block(7u, (uint64_t)u[k], (uint64_t)i, (uint32_t)k); kd_normal_pair_w(b_.x, b_.y, &z0, &z1, &u1_, &lg_); a[k] += z0 * z1; }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += a[k] + (double)u[k] + (double)w[k];
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
static void overlap(const char* name, int valu_iters)
{
    double* d; hipMalloc(&d, sizeof(double) * 512 * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int mi = 20000;
    float t[3];
    for (int c = 0; c < 3; ++c) {
        const int m = c == 1 ? 0 : mi, v = c == 0 ? 0 : valu_iters;
        hipLaunchKernelGGL(k_overlap<KIND>, dim3(256), dim3(512), 0, 0, d, m / 20, v / 20);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_overlap<KIND>, dim3(256), dim3(512), 0, 0, d, m, v);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&t[c], e0, e1);
    }
    printf("overlap MFMA f64 | %-28s: MFMA alone %.2f ms, vector alone %.2f ms, together %.2f ms (sum %.2f, max %.2f)\n", name, t[0], t[1], t[2], t[0] + t[1],
           t[0] > t[1] ? t[0] : t[1]);
    hipFree(d);
}

// Shader clock under load: s_memtime counts shader-engine cycles, s_memrealtime a constant 100 MHz reference.  KIND 0: Philox + Box-Muller pairs
// (the proposal normals of every kernel), 1: v_fma_f64 only, 2: integer only (v_mad_u64_u32 + v_bitop3); WAVES wavefronts per SIMD on every CU.
template <int KIND>
__global__ __launch_bounds__(256) void k_clock(unsigned long long* out, int iters)
{
    kd_tables_to_lds();
    const int lane = threadIdx.x & 63;
    double a[8]; uint32_t u[8]; uint64_t w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = 1.0 + 1e-3 * lane + k; u[k] = threadIdx.x * 7u + k; w[k] = (uint64_t)threadIdx.x * 0x9E3779B97F4A7C15ull + k; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (KIND == 1) a[k] = kd_fma(a[k], 0.999, 0.001);
            else if (KIND == 2) { const uint64_t p = (uint64_t)u[k] * 0xD2511F53u + w[k]; w[k] = p; u[k] = KD_XOR3((uint32_t)(p >> 32), (uint32_t)p, (uint32_t)i); }
            else { double z0, z1, u1_, lg_; const kd_u32x4 b_ = kd_stream_block(7u, (uint64_t)u[k], (uint64_t)i, (uint32_t)k); kd_normal_pair_w(b_.x, b_.y, &z0, &z1, &u1_, &lg_); a[k] += z0 * z1; }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] + (double)u[k] + (double)w[k];
    if (s == 12345.678) out[3] = 1;
    if (blockIdx.x == 17 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
template <int KIND>
static void clock_probe(const char* name, int waves_per_simd, int iters)
{
    unsigned long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipLaunchKernelGGL(k_clock<KIND>, dim3(256 * waves_per_simd), dim3(256), 0, 0, d, iters / 10);
    hipLaunchKernelGGL(k_clock<KIND>, dim3(256 * waves_per_simd), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("shader clock under %-34s (%d wavefronts per SIMD, %.1f ms): %.0f MHz\n", name, waves_per_simd, h[1] / 1e5, 100.0 * (double)h[0] / (double)h[1]);
    hipFree(d);
}

// Issue cadence in real shader cycles (s_memtime inside the kernel): NCH independent chains per wavefront, W wavefronts per SIMD.
template <int OP, int NCH>
__global__ __launch_bounds__(256) void k_issue(unsigned long long* out, int iters)
{
    const int lane = threadIdx.x & 63;
    double a[NCH]; uint32_t u[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) { a[k] = 1.0 + 1e-3 * lane + k; u[k] = threadIdx.x * 7u + k; }
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if (OP == 0) a[k] = kd_fma(a[k], 0.999, 0.001);
                else if (OP == 1) u[k] = KD_XOR3(u[k], 0x12345u, (uint32_t)i);
                else a[k] = a[k] * a[k];
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NCH; ++k) s += a[k] + (double)u[k];
    if (s == 12345.678) out[3] = 1;
    if (blockIdx.x == 5 && threadIdx.x == 0) out[0] = c1 - c0;
}
template <int OP, int NCH>
static void issue(const char* name)
{
    unsigned long long* d; hipMalloc(&d, 64);
    const int iters = 4000;
    printf("%-14s %d chains:", name, NCH);
    for (int w : { 1, 2, 3, 4, 8 }) {
        hipMemset(d, 0, 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k_issue<OP, NCH>), dim3(256 * w), dim3(256), 0, 0, d, iters / 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_issue<OP, NCH>), dim3(256 * w), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("  W=%d: wave %.2f, kernel %.2f", w, (double)h / ((double)iters * 8 * NCH), ms * 1e-3 * 2.3e9 / ((double)iters * 8 * NCH * w));
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("   (wave: shader cycles per instruction of one wavefront; kernel: elapsed x 2.3 GHz per wave-instruction of a SIMD)\n");
    hipFree(d);
}

int main(int argc, char** argv)
{
    if (argc > 1 && argv[1][0] == 'i') {
        issue<0, 1>("v_fma_f64"); issue<0, 2>("v_fma_f64"); issue<0, 4>("v_fma_f64"); issue<0, 8>("v_fma_f64");
        issue<2, 1>("v_mul_f64"); issue<2, 4>("v_mul_f64");
        issue<1, 1>("v_bitop3_b32"); issue<1, 2>("v_bitop3_b32"); issue<1, 8>("v_bitop3_b32");
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'c') {
        for (int rep = 0; rep < 2; ++rep) {
            clock_probe<0>("Philox + Box-Muller pairs", 3, 4000);
            clock_probe<1>("v_fma_f64", 3, 60000);
            clock_probe<2>("v_mad_u64_u32 + v_bitop3", 3, 40000);
            clock_probe<0>("Philox + Box-Muller pairs", 1, 4000);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'o') {
        overlap<0>("v_fma_f64", 400000);
        overlap<1>("v_mad_u64_u32 + v_bitop3", 160000);
        overlap<2>("Philox + Box-Muller pair", 7000);
        return 0;
    }
    run_tp<0>("v_fma_f64");
    run_tp<1>("v_add_f64");
    run_tp<2>("v_mul_f64");
    run_tp<3>("v_mad_u64_u32");
    run_tp<4>("v_mul_lo_u32");
    run_tp<5>("v_mul_hi_u32 + v_add_u32");
    run_tp<6>("v_bitop3_b32");
    run_tp<7>("v_rsq_f64 + v_add_f64");
    run_tp<8>("v_rcp_f64 + v_add_f64");
    run_tp<9>("v_cvt_i32_f64 + v_cvt_f64_i32 + v_add_f64");
    run_tp<10>("2 v_mov_dpp + v_add_f64");
    run_tp<11>("v_cmp_f64 + 2 v_cndmask + 2 v_add_f64");
    mfma_peak();
    run<8>("fma_f64 (dependent)");
    run<9>("mul_lo_u32 chain");
    run<10>("mad_u64_u32+2xor");
    run<0>("philox4x32_10");
    run<11>("u52");
    run<1>("kd_log");
    run<12>("kd_log_u01 (table)");
    run<2>("kd_exp");
    run<3>("kd_sincos2pi (table)");
    run<4>("sqrt_f64");
    run<5>("div_f64");
    run<6>("philox+normal_pair");
    run<7>("allreduce<3> G=64", 64);
    run<7>("allreduce<3> G=32", 32);
    run<7>("allreduce<3> G=16", 16);
    run<7>("allreduce<3> G=8", 8);
    return 0;
}
