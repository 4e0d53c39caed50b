// Round-3 gate for the MFMA shape of the render kernel's tile loop: the tile loop in miniature with its
// real side traffic, at the socket power cap, on random operands, weights walked through an L2-resident 2 MB stream.
//
// A "block" = 32 output features x 32 k x 128 points per CU, three split products (hi.hi, lo.hi, hi.lo): 192 matrix-pipe
// cycles per SIMD in every variant.  Per block and CU the side traffic is the render kernel's: the A fragments of the block
// read from LDS by every wave (ds_read_b128), 4 LDS-DMA pieces of 1 KiB (the next chunk's share), the deferred epilogue of the
// previous tile (16 values per lane and 32-row tile at 32 points per wave: fma, max, cvt_pk, 2 fma_mix, cvt_pk = ~12 VALU per
// block and wave), one bias read per tile, one workgroup barrier per tile (8 blocks) with the waits in front of it.
//
// Variants:
//   S32/1  v_mfma_f32_32x32x16_f16, one wave per SIMD, 32 points per wave          (the round-2 kernel)
//   S16/1  v_mfma_f32_16x16x32_f16, one wave per SIMD, 2 x 16 points per wave: 12 MFMAs, 4 reads, 12 VALU, 1 piece per block
//   S16/2  v_mfma_f32_16x16x32_f16, TWO waves per SIMD (512 threads, <= 256 registers), 16 points per wave: 6 MFMAs, 4 reads,
//          6 VALU per block and wave, one piece every second block - twice the LDS read traffic per CU
// DMA placement: 0 none | 1 the render kernel's plan: a wave's 8 pieces of the next chunk go out one per half block right behind
//   the barrier (half blocks 6.5, 7, 7.5 of this tile, 0 .. 2 of the next), so the barrier's vmcnt(0) finds them landed
//   | 2 one piece at the END of every block (the last one lands late: what a careless plan costs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int LDS_BYTES = 144 * 1024, BLOCKS_PER_TILE = 8;
constexpr int STREAM_TILES = 2048;   // 1-KiB tiles of the weight stream (2 MB, L2-resident like one network)

#define VALU(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(va[(i) & 7]) : "v"(va[((i) + 1) & 7]))
#define GAP() do { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DMA_PIECE(SRC) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(SRC) : "memory")

// the render kernel's DMA plan at half-block granularity (q = half of block b): half blocks 13, 14, 15 and 0 .. 4
__host__ __device__ constexpr bool has_piece(int b, int q) { return 2 * b + q >= 13 || 2 * b + q <= 4; }
__host__ __device__ constexpr int piece_no(int b, int q) { return (2 * b + q) & 7; }

template <int SHAPE, int WPS, int NV, int DMA, bool BAR>
__global__ __launch_bounds__(256 * WPS) void k(float* out, const h8* src, int tiles, unsigned long long* cyc) {
    extern __shared__ char lds[];
    constexpr int NW = 4 * WPS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256 * WPS) reinterpret_cast<h8*>(lds)[i] = src[i & 1023];
    __syncthreads();
    h8 xh[2], xl[2];     // B operands: two ray groups (S16/1), two k-steps (S32), one ray group (S16/2: index 0 only); half zeros
    for (int i = 0; i < 2; ++i) { xh[i] = src[1024 + lane + 64 * i]; xl[i] = src[1024 + lane + 128 + 64 * i]; }
    asm volatile("" : "+v"(xh[0]), "+v"(xl[0]), "+v"(xh[1]), "+v"(xl[1]));
    h8 fh[2][2], fl[2][2];   // A fragment ring, two deep: [slot][sub-tile or k-step]
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { fh[i][j] = src[lane + 64 * (2 * i + j)]; fl[i][j] = src[lane + 256 + 64 * (2 * i + j)]; }
    float va[8];
    for (int i = 0; i < 8; ++i) va[i] = (float)lane * 0.001f + i;
    const char* base = lds + lane * 16;
    const unsigned lane_off = lane * 16;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(80u * 1024 + wave * 4096));
    f16v a32;
    f4v a16[4];
    for (int r = 0; r < 16; ++r) a32[r] = 0.f;
    for (int i = 0; i < 4; ++i) a16[i] = f4v{0.f, 0.f, 0.f, 0.f};
    float4 bias = {0.f, 0.f, 0.f, 0.f};
    unsigned tile_pos = blockIdx.x * 37u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < tiles; ++t) {
        const char* cb = base + (t & 1) * 32768;
        bias = *reinterpret_cast<const float4*>(lds + 70 * 1024 + ((t & 63) * 128 + (lane >> 5) * 16));   // the tile's bias row
        tile_pos = (tile_pos + 32) & (STREAM_TILES - 1);
        const h8* chunk = src + (size_t)(tile_pos + wave * (32 / NW)) * 64;   // this wave's share of the 32 pieces of the next chunk
#pragma unroll
        for (int b = 0; b < BLOCKS_PER_TILE; ++b) {
            const int use = b & 1, nxt = use ^ 1;
            if (BAR && b == BLOCKS_PER_TILE - 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __syncthreads(); }
            int v = 0;
            if (SHAPE == 32) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    a32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[use][q], xh[q], a32, 0, 0, 0);
                    fh[nxt][q] = *reinterpret_cast<const h8*>(cb + (4 * b + 2 * q) * 1024);
                    fl[nxt][q] = *reinterpret_cast<const h8*>(cb + (4 * b + 2 * q + 1) * 1024);
                    for (int i = 0; i < NV / 6; ++i) VALU(v++);
                    GAP();
                    a32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[use][q], xh[q], a32, 0, 0, 0);
                    const bool piece = DMA == 1 ? has_piece(b, q) : (DMA == 2 && q == 1);
                    if (piece) DMA_PIECE(chunk + 64 * piece_no(b, q));
                    else for (int i = 0; i < NV / 6; ++i) VALU(v++);
                    GAP();
                    a32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[use][q], xl[q], a32, 0, 0, 0);
                    for (int i = 0; i < NV / 6 + (piece ? NV / 6 : 0); ++i) VALU(v++);
                    GAP();
                }
            } else if (WPS == 1) {
                // 12 MFMAs: sub-tile r (A: fh/fl[use][r]) x ray group p (B: xh/xl[p]) x product
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    const int r = (m / 6) & 1, p = (m / 3) & 1, pass = m % 3;
                    a16[2 * r + p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pass == 1 ? fl[use][r] : fh[use][r], pass == 2 ? xl[p] : xh[p], a16[2 * r + p], 0, 0, 0);
                    if (m == 0) fh[nxt][0] = *reinterpret_cast<const h8*>(cb + (4 * b) * 1024);
                    if (m == 3) fl[nxt][0] = *reinterpret_cast<const h8*>(cb + (4 * b + 1) * 1024);
                    if (m == 6) fh[nxt][1] = *reinterpret_cast<const h8*>(cb + (4 * b + 2) * 1024);
                    if (m == 9) fl[nxt][1] = *reinterpret_cast<const h8*>(cb + (4 * b + 3) * 1024);
                    bool dma = false;
                    if (DMA == 1) dma = (m == 5 || m == 11) && has_piece(b, m / 6);
                    if (DMA == 2) dma = m == 11;
                    if (dma) DMA_PIECE(chunk + 64 * piece_no(b, m / 6));
                    else if (v < NV && m != 0 && m != 3 && m != 6 && m != 9) { VALU(v++); if (v < NV && (m == 1 || m == 4 || m == 7 || m == 10)) VALU(v++); }
                    GAP();
                }
                while (v < NV) VALU(v++);
            } else {
                // two waves per SIMD, 16 points per wave: 6 MFMAs per block = sub-tile r x product
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const int r = m / 3, pass = m % 3;
                    a16[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pass == 1 ? fl[use][r] : fh[use][r], pass == 2 ? xl[0] : xh[0], a16[r], 0, 0, 0);
                    if (m == 0) fh[nxt][0] = *reinterpret_cast<const h8*>(cb + (4 * b) * 1024);
                    if (m == 1) fl[nxt][0] = *reinterpret_cast<const h8*>(cb + (4 * b + 1) * 1024);
                    if (m == 3) fh[nxt][1] = *reinterpret_cast<const h8*>(cb + (4 * b + 2) * 1024);
                    if (m == 4) fl[nxt][1] = *reinterpret_cast<const h8*>(cb + (4 * b + 3) * 1024);
                    bool dma = false;
                    if (DMA == 1) dma = m == 5 && (b == 7 || b <= 2);    // 4 pieces per wave: blocks 7, 0, 1, 2
                    if (DMA == 2) dma = (b & 1) && m == 5;
                    if (dma) DMA_PIECE(chunk + 64 * (b & 3));
                    else if (v < NV / 2 && (m == 2 || m == 5 || m == 0 || m == 3)) { VALU(v++); if (v < NV / 2 && (m == 2 || m == 5)) VALU(v++); }
                    GAP();
                }
                while (v < NV / 2) VALU(v++);
            }
        }
        va[0] += bias.x;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a32[r];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) s += a16[i][r];
    for (int i = 0; i < 8; ++i) s += va[i];
    out[blockIdx.x * 256 * WPS + threadIdx.x] = s;
    if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

template <int SHAPE, int WPS, int NV, int DMA, bool BAR>
void run(float* out, h8* src, unsigned long long* cyc, const char* what) {
    auto fn = k<SHAPE, WPS, NV, DMA, BAR>;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int tiles = 4000, nblk = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 0; double cy = 0, ghz = 0;
    for (int rep = 0; rep < 6; ++rep) {   // the later repetitions are the power-capped steady state
        (void)hipEventRecord(e0);
        for (int l = 0; l < 10; ++l) fn<<<nblk, 256 * WPS, LDS_BYTES>>>(out, src, tiles, cyc);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[2]; (void)hipMemcpy(c, cyc + 2 * 8 * 100, 16, hipMemcpyDeviceToHost);   // workgroup 100, wave 0
        best = ms; cy = (double)c[0] / (tiles * BLOCKS_PER_TILE); ghz = (double)c[0] / ((double)c[1] * 10.0);
    }
    const double flop = 1024.0 * 10 * tiles * BLOCKS_PER_TILE * 6.0 * 32768.0;
    printf("%-7s valu/block %2d dma %d barrier %d: %6.1f cycles/block (192 = pipe rate), in-kernel clock %.3f GHz, %6.0f TFLOP/s executed  %s\n", what, NV, DMA,
           (int)BAR, cy, ghz, flop / (best * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main() {
    float* out; h8* src; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&src, (size_t)(STREAM_TILES + 64) * 1024); (void)hipMalloc(&cyc, 256 * 8 * 16);
    std::vector<_Float16> h((size_t)(STREAM_TILES + 64) * 512);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) {
        const float v = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
        h[i] = (i >= 1024 * 8 && i < 2048 * 8 && (rand() & 1)) ? (_Float16)0.f : (_Float16)v;    // the B operands: half zeros (ReLU outputs)
    }
    (void)hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<32, 1, 12, 0, true>(out, src, cyc, "S32/1");
    run<32, 1, 12, 1, true>(out, src, cyc, "S32/1");
    run<16, 1, 12, 0, true>(out, src, cyc, "S16/1");
    run<16, 1, 12, 1, true>(out, src, cyc, "S16/1");
    run<16, 1, 12, 2, true>(out, src, cyc, "S16/1");
    run<16, 1, 16, 1, true>(out, src, cyc, "S16/1");
    run<16, 1, 8, 1, true>(out, src, cyc, "S16/1");
    run<32, 1, 12, 2, true>(out, src, cyc, "S32/1");
    run<16, 2, 12, 0, true>(out, src, cyc, "S16/2");
    run<16, 2, 12, 1, true>(out, src, cyc, "S16/2");
    run<16, 2, 12, 2, true>(out, src, cyc, "S16/2");
    run<16, 2, 24, 1, true>(out, src, cyc, "S16/2");
    run<32, 1, 12, 1, true>(out, src, cyc, "S32/1");   // again: drift of the box over the run
    return 0;
}
