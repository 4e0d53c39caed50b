// What does one LDS-DMA piece cost a wave that paces the matrix pipe, and why?  The tile loop of tile_shape3.hip (one wave per
// SIMD, 12 epilogue VALU + 4 fragment reads per 32-k block, barrier per tile), one "extra" per block placed in the gap behind
// the block's last MFMA; X selects the extra:
//   0 nothing | 1 global_load_lds_dwordx4 (1 KiB) | 2 global_load_dwordx4 to registers | 3 global_load_lds_dword (256 B)
//   4 the piece with only lanes 0..31 active | 5 buffer_load_dwordx4 .. lds | 6..9 s_nop padding of 8/16/24/32 cycles
//   10 the piece, only wave 0 issues (x4) | 11 the piece, no barrier | 12 piece placed behind the FIRST MFMA of the block
//   13 two half pieces (lanes 0..31 / 32..63) in two different gaps
//   14/15/16 the piece (as 1) with the waves skewed behind the barrier by wave x 8 / 16 / 24 cycles of s_nop (a branch ladder)
//   18 the piece as 1, but nothing ever waits for it (no vmcnt wait in the loop): pure issue cost
//   19 one piece per block issued by wave 0 only | 20 two pieces per block in blocks 0..3 (middle and last gap), none in 4..7
//   21 as 20 and no vmcnt wait
//   17 rotating issuer: in block b wave (b & 3) issues four pieces back to back, the others none (uniform branch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));
constexpr int LDS_BYTES = 144 * 1024, BLOCKS_PER_TILE = 8, STREAM_TILES = 2048, NV = 12;
#define VALU(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(va[(i) & 7]) : "v"(va[((i) + 1) & 7]))
#define GAP() do { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int SHAPE, int X>
__global__ __launch_bounds__(256) void k(float* out, const h8* src, int tiles, unsigned long long* cyc) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) reinterpret_cast<h8*>(lds)[i] = src[i & 1023];
    __syncthreads();
    h8 xh[2], xl[2];
    for (int i = 0; i < 2; ++i) { xh[i] = src[1024 + lane + 64 * i]; xl[i] = src[1024 + lane + 128 + 64 * i]; }
    asm volatile("" : "+v"(xh[0]), "+v"(xl[0]), "+v"(xh[1]), "+v"(xl[1]));
    h8 fh[2][2], fl[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { fh[i][j] = src[lane + 64 * (2 * i + j)]; fl[i][j] = src[lane + 256 + 64 * (2 * i + j)]; }
    float va[8];
    for (int i = 0; i < 8; ++i) va[i] = (float)lane * 0.001f + i;
    const char* base = lds + lane * 16;
    const unsigned lane_off = lane * 16, lane_off4 = lane * 4;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(80u * 1024 + wave * 4096));
    f16v a32;
    f4v a16[4];
    for (int r = 0; r < 16; ++r) a32[r] = 0.f;
    for (int i = 0; i < 4; ++i) a16[i] = f4v{0.f, 0.f, 0.f, 0.f};
    unsigned tile_pos = blockIdx.x * 37u;
    i4v rsrc;   // buffer resource over the whole stream: base, stride 0, num_records, flags (raw dword access)
    rsrc[0] = (int)(uintptr_t)src; rsrc[1] = (int)((uintptr_t)src >> 32) & 0xffff; rsrc[2] = (STREAM_TILES + 64) * 1024; rsrc[3] = 0x00020000;
    const unsigned long long lo_half = 0xffffffffull, hi_half = 0xffffffff00000000ull;
    f4v sink = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < tiles; ++t) {
        const char* cb = base + (t & 1) * 32768;
        tile_pos = (tile_pos + 32) & (STREAM_TILES - 1);
        const h8* chunk = src + (size_t)(tile_pos + wave * 8) * 64;
        const unsigned chunk_off = (tile_pos + wave * 8) * 1024;
#pragma unroll
        for (int b = 0; b < BLOCKS_PER_TILE; ++b) {
            const int use = b & 1, nxt = use ^ 1;
            if (X != 11 && b == BLOCKS_PER_TILE - 2) {
                if (X == 18 || X == 21) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
#define SKEW(K) asm volatile("s_cmp_lt_u32 %0, 1\n\ts_cbranch_scc1 .Lskew%=\n\ts_nop " #K "\n\ts_cmp_lt_u32 %0, 2\n\ts_cbranch_scc1 .Lskew%=\n\ts_nop " #K \
                          "\n\ts_cmp_lt_u32 %0, 3\n\ts_cbranch_scc1 .Lskew%=\n\ts_nop " #K "\n.Lskew%=:" :: "s"(wave) : "scc", "memory")
                if (X == 14) SKEW(1);
                if (X == 15) SKEW(3);
                if (X == 16) SKEW(5);
            }
            if (X == 11 && b == BLOCKS_PER_TILE - 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
            int v = 0;
            auto extra = [&](int where) __attribute__((always_inline)) {   // where: 0 = behind the first MFMA, 1 = middle, 2 = behind the last
                const h8* s = chunk + 64 * b;
                if (X == 12) { if (where == 0) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(s) : "memory"); return; }
                if (X == 13) {
                    if (where == 1) asm volatile("s_mov_b64 exec, %2\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(lane_off), "s"(s), "s"(lo_half) : "memory");
                    if (where == 2) asm volatile("s_mov_b64 exec, %2\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(lane_off), "s"(s), "s"(hi_half) : "memory");
                    return;
                }
                if (X == 20 || X == 21) {
                    if (b < 4 && where == 1) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(s) : "memory");
                    if (b < 4 && where == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" :: "v"(lane_off), "s"(s) : "memory");
                    return;
                }
                if (where != 2) return;
                if (X == 19 && wave == 0) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(s) : "memory");
                if (X == 17) {
                    if (wave == (b & 3)) {
                        asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(s) : "memory");
                        asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(lane_off), "s"(s) : "memory");
                        asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" :: "v"(lane_off), "s"(s) : "memory");
                        asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(lane_off), "s"(s) : "memory");
                    }
                }
                if (X == 1 || X == 11 || X == 18 || (X >= 14 && X <= 16)) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(s) : "memory");
                if (X == 2) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(sink) : "v"(lane_off), "s"(s) : "memory");   // "+v": one fixed destination (the load lands later)
                if (X == 3) asm volatile("global_load_lds_dword %0, %1" :: "v"(lane_off4), "s"(s) : "memory");
                if (X == 4) asm volatile("s_mov_b64 exec, %2\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(lane_off), "s"(s), "s"(lo_half) : "memory");
                if (X == 5) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:0 lds" :: "v"(lane_off), "s"(rsrc), "s"(chunk_off + 1024 * b) : "memory");
                if (X == 6) asm volatile("s_nop 7");
                if (X == 7) asm volatile("s_nop 15");
                if (X == 8) asm volatile("s_nop 15\n\ts_nop 7");
                if (X == 9) asm volatile("s_nop 15\n\ts_nop 15");
                if (X == 10 && wave == 0) {
                    asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(s) : "memory");
                    asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(lane_off), "s"(s) : "memory");
                    asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" :: "v"(lane_off), "s"(s) : "memory");
                    asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(lane_off), "s"(s) : "memory");
                }
            };
            if (SHAPE == 32) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    a32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[use][q], xh[q], a32, 0, 0, 0);
                    fh[nxt][q] = *reinterpret_cast<const h8*>(cb + (4 * b + 2 * q) * 1024);
                    fl[nxt][q] = *reinterpret_cast<const h8*>(cb + (4 * b + 2 * q + 1) * 1024);
                    if (q == 0) extra(0);
                    VALU(v++); VALU(v++);
                    GAP();
                    a32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[use][q], xh[q], a32, 0, 0, 0);
                    if (q == 0) extra(1);
                    VALU(v++); VALU(v++);
                    GAP();
                    a32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[use][q], xl[q], a32, 0, 0, 0);
                    if (q == 1) extra(2); else { }
                    VALU(v++); VALU(v++);
                    GAP();
                }
            } else {
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    const int r = (m / 6) & 1, p = (m / 3) & 1, pass = m % 3;
                    a16[2 * r + p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pass == 1 ? fl[use][r] : fh[use][r], pass == 2 ? xl[p] : xh[p], a16[2 * r + p], 0, 0, 0);
                    if (m == 0) fh[nxt][0] = *reinterpret_cast<const h8*>(cb + (4 * b) * 1024);
                    if (m == 3) fl[nxt][0] = *reinterpret_cast<const h8*>(cb + (4 * b + 1) * 1024);
                    if (m == 6) fh[nxt][1] = *reinterpret_cast<const h8*>(cb + (4 * b + 2) * 1024);
                    if (m == 9) fl[nxt][1] = *reinterpret_cast<const h8*>(cb + (4 * b + 3) * 1024);
                    if (m == 1) extra(0);
                    if (m == 5) extra(1);
                    if (m == 11) extra(2);
                    // 12 VALU: one in gaps 1, 2, 4, 5, 7, 8, 10 and two in none -> gaps 1,2,4,5,7,8,10 = 7, plus second in 2, 4, 7, 8, 10
                    if (m == 1 || m == 2 || m == 4 || m == 5 || m == 7 || m == 8 || m == 10) VALU(v++);
                    if (m == 2 || m == 4 || m == 7 || m == 8 || m == 10) VALU(v++);
                    GAP();
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a32[r];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) s += a16[i][r] + sink[r];
    for (int i = 0; i < 8; ++i) s += va[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) { cyc[(blockIdx.x * 4 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

template <int SHAPE, int X>
void run(float* out, h8* src, unsigned long long* cyc) {
    auto fn = k<SHAPE, X>;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int tiles = 4000, nblk = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 0; double cy[4] = {0, 0, 0, 0}, ghz = 0;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        for (int l = 0; l < 10; ++l) fn<<<nblk, 256, LDS_BYTES>>>(out, src, tiles, cyc);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[8]; (void)hipMemcpy(c, cyc + 2 * 4 * 100, 64, hipMemcpyDeviceToHost);
        best = ms; for (int w = 0; w < 4; ++w) cy[w] = (double)c[2 * w] / (tiles * BLOCKS_PER_TILE); ghz = (double)c[0] / ((double)c[1] * 10.0);
    }
    const double flop = 1024.0 * 10 * tiles * BLOCKS_PER_TILE * 6.0 * 32768.0;
    printf("S%d extra %2d: %6.1f cycles/block (waves 1-3: %.1f %.1f %.1f), clock %.3f GHz, %6.0f TFLOP/s  %s\n", SHAPE, X, cy[0], cy[1], cy[2], cy[3], ghz,
           flop / (best * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

template <int SHAPE>
void all(float* out, h8* src, unsigned long long* cyc) {
    run<SHAPE, 0>(out, src, cyc); run<SHAPE, 1>(out, src, cyc); run<SHAPE, 18>(out, src, cyc); run<SHAPE, 19>(out, src, cyc); run<SHAPE, 20>(out, src, cyc); run<SHAPE, 21>(out, src, cyc);

}

int main() {
    float* out; h8* src; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&src, (size_t)(STREAM_TILES + 64) * 1024); (void)hipMalloc(&cyc, 256 * 8 * 16);
    std::vector<_Float16> h((size_t)(STREAM_TILES + 64) * 512);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) {
        const float v = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
        h[i] = (i >= 1024 * 8 && i < 2048 * 8 && (rand() & 1)) ? (_Float16)0.f : (_Float16)v;
    }
    (void)hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    all<16>(out, src, cyc);
    all<32>(out, src, cyc);
    return 0;
}
