// How fast does the matrix core run under the socket power cap?  One wave per SIMD, four per CU, all CUs: a dependent
// chain of v_mfma_f32_32x32x16_f16 on RANDOM fp16 operands (zeros draw far less power), nothing else, for a few seconds.
// Prints the sustained cycles/MFMA and the implied clock; run tools/power_trace.py-style sampling next to it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 1) void k(float* out, const h8* src, int iters, unsigned long long* cyc) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(lane + 64 * i) & 1023]; b[i] = src[(lane + 64 * (i + 4)) & 1023]; }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < iters; ++t) {
#pragma unroll
        for (int q = 0; q < 48; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q & 3], b[(q >> 2) & 3], acc, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    const bool zeros = argc > 1 && atoi(argv[1]) == 0;
    const int keep_a = argc > 2 ? atoi(argv[2]) : 10, keep_b = argc > 3 ? atoi(argv[3]) : 10;   // mantissa bits kept in the A / B operands
    const int nblk = 256, iters = 20000;   // 48 * 20000 MFMAs per launch = ~15 ms
    float* out; h8* src; unsigned long long* cyc;
    (void)hipMalloc(&out, nblk * 256 * 4); (void)hipMalloc(&src, 1024 * 16); (void)hipMalloc(&cyc, nblk * 8);
    std::vector<_Float16> h(1024 * 8);
    srand(1);
    for (auto& v : h) v = zeros ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.01f);
    {   // operand values 0..511 feed A (a[i] = src[lane + 64 i], i < 4 -> h8 index < 256... A uses h8 0..255, B uses 256..511)
        unsigned short* u = reinterpret_cast<unsigned short*>(h.data());
        for (size_t i = 0; i < h.size(); ++i) {
            const int keep = (i / 8) % 512 < 256 ? keep_a : keep_b;
            u[i] &= (unsigned short)(0xFFFFu << (10 - keep));
        }
    }
    (void)hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 12; ++rep) {   // ~0.5 s each
        (void)hipEventRecord(e0);
        for (int l = 0; l < 30; ++l) k<<<nblk, 256, 140 * 1024>>>(out, src, iters, cyc);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n = 48.0 * iters;
        printf("%s operands (mantissa bits A %d, B %d): %.2f cycles/MFMA, %.2f ns/MFMA -> %.0f MHz, %.0f TFLOP/s\n", zeros ? "zero" : "random", keep_a, keep_b, c / n, ms * 1e6 / 30 / n,
               (c / n) / (ms * 1e6 / 30 / n) * 1e3, 1024.0 * 32768.0 / (ms * 1e6 / 30 / n) / 1e3);
        fflush(stdout);
    }
    return 0;
}
