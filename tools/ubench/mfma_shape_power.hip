// Which fp16 MFMA shape does more work under the socket power cap?  Same harness as mfma_power.hip (one wave per SIMD, all
// CUs, a dependent chain on random operands), for v_mfma_f32_32x32x16_f16 (16 accumulator registers, 0.31 register-file
// bytes per FLOP) and v_mfma_f32_16x16x32_f16 (4 accumulator registers, 0.24 B/FLOP), with the B operand optionally half
// zeros (ReLU activations).  argv: shape (32|16)  zero_share_B_percent
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void k(float* out, const h8* src, int iters, unsigned long long* cyc) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(lane + 64 * i) & 1023]; b[i] = src[(lane + 64 * (i + 4)) & 1023]; }
    float s = 0.f;
    unsigned long long t0, t1;
    if (SHAPE == 32) {
        f16v acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int t = 0; t < iters; ++t) {
#pragma unroll
            for (int q = 0; q < 48; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q & 3], b[(q >> 2) & 3], acc, 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < 16; ++r) s += acc[r];
    } else {
        f4v acc = {0.f, 0.f, 0.f, 0.f};
        t0 = __builtin_amdgcn_s_memtime();
        for (int t = 0; t < iters; ++t) {
#pragma unroll
            for (int q = 0; q < 96; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q & 3], b[(q >> 2) & 3], acc, 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < 4; ++r) s += acc[r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 32, zero_b = argc > 2 ? atoi(argv[2]) : 0;
    const int nblk = 256, iters = 20000;
    float* out; h8* src; unsigned long long* cyc;
    (void)hipMalloc(&out, nblk * 256 * 4); (void)hipMalloc(&src, 1024 * 16); (void)hipMalloc(&cyc, nblk * 8);
    std::vector<_Float16> h(1024 * 8);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) {
        const bool is_b = (i / 8) % 512 >= 256;
        const float v = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
        h[i] = (is_b && (rand() % 100) < zero_b) ? (_Float16)0.f : (_Float16)v;
    }
    (void)hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 10; ++rep) {
        (void)hipEventRecord(e0);
        for (int l = 0; l < 30; ++l) {
            if (shape == 32) k<32><<<nblk, 256, 140 * 1024>>>(out, src, iters, cyc);
            else k<16><<<nblk, 256, 140 * 1024>>>(out, src, iters, cyc);
        }
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double flop = 1024.0 * 48.0 * iters * 32768.0 * 30;   // both shapes: 48 x 32768 FLOP per inner round per wave
        printf("%dx%d, B %d%% zeros: %.2f cycles per 32768 FLOP, %.0f TFLOP/s\n", shape, shape, zero_b, c / (48.0 * iters), flop / (ms * 1e-3) / 1e12);
        fflush(stdout);
    }
    return 0;
}
