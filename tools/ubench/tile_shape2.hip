// Follow-up to tile_shape.hip: how well can the 16x16x32 tile loop be scheduled?  Per 32-k block: 12 MFMAs (16 cycles each), 4 fragment
// reads, 12 epilogue VALU ops, 1 LDS-DMA piece; PAT selects which gap (0..11) gets what.  Random operands, power-capped steady state.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int LDS_BYTES = 140 * 1024, BLOCKS_PER_TILE = 8;
#define VALU(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(va[(i) & 7]) : "v"(va[((i) + 1) & 7]))
#define GAP() do { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_barrier(0); } while (0)

// PAT 0: reads in gaps 0,1,6,7 | dma gap 3 | 2 valu in the other gaps until 12 are out
// PAT 1: reads in gaps 0,3,6,9 | dma gap 11 | 1 valu in every gap but the dma gap, 2 in gaps 1 (12 total)
// PAT 2: like 1 but the dma piece only in every second block (two pieces there: gaps 5 and 11)
// PAT 3: like 1, no dma at all (what the dma costs)
// PAT 4: like 1, dma yes, no barrier
template <int PAT>
__global__ __launch_bounds__(256, 1) void k(float* out, const h8* src, int tiles, unsigned long long* cyc) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) reinterpret_cast<h8*>(lds)[i] = src[i & 1023];
    __syncthreads();
    h8 xh[2], xl[2];
    for (int i = 0; i < 2; ++i) { xh[i] = src[1024 + lane + 64 * i]; xl[i] = src[1024 + lane + 128 + 64 * i]; }
    asm volatile("" : "+a"(xh[0]), "+a"(xl[0]), "+a"(xh[1]), "+a"(xl[1]));
    h8 fh[2][2], fl[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { fh[i][j] = src[lane + 64 * (2 * i + j)]; fl[i][j] = src[lane + 256 + 64 * (2 * i + j)]; }
    float va[8];
    for (int i = 0; i < 8; ++i) va[i] = (float)lane * 0.001f + i;
    const char* base = lds + lane * 16;
    const unsigned lane_off = lane * 16;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(72u * 1024 + wave * 8192));
    f4v a16[4];
    for (int i = 0; i < 4; ++i) a16[i] = f4v{0.f, 0.f, 0.f, 0.f};
    const bool barrier = PAT != 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        const char* cb = base + (t & 1) * 32768;
#pragma unroll
        for (int b = 0; b < BLOCKS_PER_TILE; ++b) {
            const int use = b & 1, nxt = use ^ 1;
            if (barrier && b == BLOCKS_PER_TILE - 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __syncthreads(); }
            int v = 0;
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                const int r = (m / 6) & 1, p = (m / 3) & 1, pass = m % 3;
                a16[2 * r + p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pass == 1 ? fl[use][r] : fh[use][r], pass == 2 ? xl[p] : xh[p], a16[2 * r + p], 0, 0, 0);
                const int r0 = PAT == 0 ? 0 : 0, r1 = PAT == 0 ? 1 : 3, r2 = 6, r3 = PAT == 0 ? 7 : 9;
                if (m == r0) fh[nxt][0] = *reinterpret_cast<const h8*>(cb + (4 * b) * 1024);
                if (m == r1) fl[nxt][0] = *reinterpret_cast<const h8*>(cb + (4 * b + 1) * 1024);
                if (m == r2) fh[nxt][1] = *reinterpret_cast<const h8*>(cb + (4 * b + 2) * 1024);
                if (m == r3) fl[nxt][1] = *reinterpret_cast<const h8*>(cb + (4 * b + 3) * 1024);
                bool dma = false;
                if (PAT == 0) dma = m == 3;
                if (PAT == 1 || PAT == 4) dma = m == 11;
                if (PAT == 2) dma = (b & 1) && (m == 5 || m == 11);
                if (dma) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(src + 64 * wave) : "memory");
                else {
                    const bool is_read = m == r0 || m == r1 || m == r2 || m == r3;
                    int nv = PAT == 0 ? (is_read ? 0 : 2) : 1;
                    if (PAT != 0 && m == 1) nv = 2;
                    for (int i = 0; i < nv && v < 12; ++i) VALU(v++);
                }
                GAP();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) s += a16[i][r];
    for (int i = 0; i < 8; ++i) s += va[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int PAT>
void run(float* out, h8* src, unsigned long long* cyc) {
    auto fn = k<PAT>;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int tiles = 4000, nblk = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0; double cy = 0;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        for (int l = 0; l < 10; ++l) fn<<<nblk, 256, LDS_BYTES>>>(out, src, tiles, cyc);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        cy = (double)c / (tiles * BLOCKS_PER_TILE);
    }
    const double flop = 1024.0 * 10 * tiles * BLOCKS_PER_TILE * 6.0 * 32768.0;
    printf("16x16x32 pattern %d: %6.1f cycles/block, %6.0f TFLOP/s executed  %s\n", PAT, cy, flop / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main() {
    float* out; h8* src; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&src, 2048 * 16); (void)hipMalloc(&cyc, 256 * 32);
    std::vector<_Float16> h(2048 * 8);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) {
        const float v = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
        h[i] = (i >= 1024 * 8 && (rand() & 1)) ? (_Float16)0.f : (_Float16)v;
    }
    (void)hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<0>(out, src, cyc); run<1>(out, src, cyc); run<2>(out, src, cyc); run<3>(out, src, cyc); run<4>(out, src, cyc);
    return 0;
}
