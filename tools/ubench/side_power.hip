// What the render kernel's SIDE traffic draws at the socket, one stream at a time ("price the 7.4 TB/s
// L2->LDS stream in watts").  One wave per SIMD, four per CU, all CUs, a few seconds per mode; tools/ubench/side_power.py samples
// rocm-smi next to it.  Rates are the kernel's (profiles/r02_pmc_summary.txt, per wave): one LDS-DMA piece of 1 KiB per ~256
// cycles (528 per 135 k-cycle sample iteration = 7.4 TB/s over the chip at 1.8 GHz), one ds_read_b128 per ~64 cycles, ~3 VALU
// per MFMA.  Modes (argv[1]):
//   0 spin (s_nop only)                       1 LDS-DMA stream at the kernel's rate      2 the same at twice the rate
//   3 ds_read_b128 at the kernel's rate        4 VALU (v_fma_f32) at the kernel's rate    5 DMA + reads + VALU together
//   6 MFMA chain on random operands alone      7 MFMA chain + DMA + reads + VALU (the kernel's mix)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int LDS_BYTES = 144 * 1024, STREAM_TILES = 2048;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const h8* src, int iters, unsigned long long* cyc) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) reinterpret_cast<h8*>(lds)[i] = src[i & 1023];
    __syncthreads();
    constexpr bool DMA = MODE == 1 || MODE == 2 || MODE == 5 || MODE == 7;
    constexpr bool RD = MODE == 3 || MODE == 5 || MODE == 7;
    constexpr bool VALU = MODE == 4 || MODE == 5 || MODE == 7;
    constexpr bool MFMA = MODE == 6 || MODE == 7;
    h8 a[2], b[2], f[2];
    for (int i = 0; i < 2; ++i) { a[i] = src[lane + 64 * i]; b[i] = src[1024 + lane + 64 * i]; f[i] = a[i]; }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float va[4] = {1.f + lane, 2.f, 3.f, 4.f};
    const unsigned lane_off = lane * 16;
    const char* base = lds + lane * 16;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(80u * 1024 + wave * 4096));
    unsigned pos = blockIdx.x * 37u + wave * 8u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < iters; ++t) {
        // one iteration = 8 MFMA slots of 32 cycles = 256 cycles: 1 piece (2 in mode 2), 4 reads, 24 VALU
        pos = (pos + 32) & (STREAM_TILES - 1);
        const h8* s = src + (size_t)pos * 64;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(q & 1 ? f[0] : a[q >> 2], b[q & 1], acc, 0, 0, 0);
            else asm volatile("s_nop 7");       // the slot's 32 cycles
            if (DMA && (q == 1 || (MODE == 2 && q == 5))) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(s + 64 * (q >> 2)) : "memory");
            if (RD && (q & 1) == 0) f[(q >> 1) & 1] = *reinterpret_cast<const h8*>(base + ((t & 7) * 8 + q) * 1024);
            if (VALU) {
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(va[0]) : "v"(va[1]));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(va[2]) : "v"(va[3]));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(va[1]) : "v"(va[2]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DMA && (t & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float sum = va[0] + va[1] + va[2] + va[3];
    for (int r = 0; r < 16; ++r) sum += acc[r];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) sum += (float)f[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (lane == 0) { cyc[(blockIdx.x * 4 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

template <int MODE>
void run(float* out, h8* src, unsigned long long* cyc, double seconds) {
    auto fn = k<MODE>;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int iters = 40000, nblk = 256;   // 40000 x 256 cycles = ~5 ms per launch
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0; double total = 0; int launches = 0;
    while (total < seconds * 1e3) {
        (void)hipEventRecord(e0);
        for (int l = 0; l < 20; ++l) fn<<<nblk, 256, LDS_BYTES>>>(out, src, iters, cyc);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        total += ms; launches += 20;
    }
    unsigned long long c[2]; (void)hipMemcpy(c, cyc + 2 * 4 * 100, 16, hipMemcpyDeviceToHost);
    const double per_iter = (double)c[0] / iters, ghz = (double)c[0] / ((double)c[1] * 10.0);
    const double it_s = 1024.0 * iters * 20 / (ms * 1e-3);     // wave-iterations per second over the chip (last batch)
    printf("mode %d: %.1f cycles per 8-slot iteration, clock %.3f GHz; chip-wide: LDS-DMA %.2f TB/s, LDS reads %.2f TB/s, MFMA %.0f TFLOP/s  %s\n", MODE, per_iter, ghz,
           (MODE == 1 || MODE == 5 || MODE == 7 ? 1 : (MODE == 2 ? 2 : 0)) * 1024.0 * it_s / 1e12, (MODE == 3 || MODE == 5 || MODE == 7 ? 4 : 0) * 1024.0 * it_s / 1e12,
           (MODE == 6 || MODE == 7 ? 8 : 0) * 32768.0 * it_s / 1e12, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    float* out; h8* src; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&src, (size_t)(STREAM_TILES + 64) * 1024); (void)hipMalloc(&cyc, 256 * 4 * 16);
    std::vector<_Float16> h((size_t)(STREAM_TILES + 64) * 512);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.01f);
    (void)hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    switch (mode) {
        case 0: run<0>(out, src, cyc, seconds); break; case 1: run<1>(out, src, cyc, seconds); break;
        case 2: run<2>(out, src, cyc, seconds); break; case 3: run<3>(out, src, cyc, seconds); break;
        case 4: run<4>(out, src, cyc, seconds); break; case 5: run<5>(out, src, cyc, seconds); break;
        case 6: run<6>(out, src, cyc, seconds); break; default: run<7>(out, src, cyc, seconds); break;
    }
    return 0;
}
