// Micro-benchmark 2: one wave per SIMD, four waves per workgroup, a "tile" of 16 k-steps x 3 dependent MFMAs with the
// render kernel's side traffic placed in chosen MFMA gaps: fragment reads (2 ds_read_b128 / k-step, consumed 3 k-steps
// later), LDS-DMA (8 global_load_lds_dwordx4 per tile and wave: k-steps 0-4 and 13-15), one workgroup barrier per tile,
// independent VALU filler.  Prints shader cycles per tile (1536 = matrix pipe saturated).
//   RD: 0 both reads behind MFMA 0 | 1 hi behind MFMA 0, lo behind MFMA 1 | 2 hi behind 0, lo behind 2
//   DG: gap of the DMA piece (-1 none); a gap with a piece takes no VALU
//   V0/V1/V2: VALU ops behind MFMA 0/1/2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int KSTEPS = 16, TILES = 1000, LDS_BYTES = 140 * 1024;

template <int RD, int DG, bool BARRIER, int V0, int V1, int V2, int FK = 0, bool BAGPR = false>
__global__ __launch_bounds__(256, 1) void k(float* out, const h8* src, unsigned long long* cyc) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) reinterpret_cast<h8*>(lds)[i] = src[i & 1023];
    __syncthreads();
    h8 xh = src[lane], xl = src[lane + 64];
    if (BAGPR) asm volatile("" : "+a"(xh), "+a"(xl));
    h8 fh[4], fl[4];
    for (int i = 0; i < 4; ++i) { fh[i] = src[lane + 128 + 64 * i]; fl[i] = src[lane + 384 + 64 * i]; }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float va[8];
    for (int i = 0; i < 8; ++i) va[i] = (float)lane * 0.001f + i;
    const char* base = lds + lane * 16;
    const unsigned lane_off = lane * 16;
    const unsigned dma_dst = 72 * 1024 + wave * 8192;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dma_dst));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < TILES; ++t) {
        const char* cb = base + (t & 1) * 32768;
#pragma unroll
        for (int q = 0; q < KSTEPS; ++q) {
            const int use = q & 3, slot = (q + 3) & 3;
            if (BARRIER && q == KSTEPS - 3) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
            }
            const bool dma_k = DG >= 0 && (q < 5 || q >= KSTEPS - 3);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(m == 1 ? fl[use] : fh[use], m == 2 ? xl : xh, acc, 0, 0, 0);
                if (dma_k && m == DG) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(src + 64 * wave) : "memory");
                if (m == 0) fh[slot] = *reinterpret_cast<const h8*>(cb + (2 * q) * 1024);
                if (m == (RD == 0 ? 0 : RD)) fl[slot] = *reinterpret_cast<const h8*>(cb + (2 * q + 1) * 1024);
                const int nv = (dma_k && m == DG) ? 0 : (m == 0 ? V0 : (m == 1 ? V1 : V2));
#pragma unroll
                for (int i = 0; i < nv; ++i) {
                    if (FK == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(va[i & 7]) : "v"(va[(i + 1) & 7]));
                    if (FK == 1) asm volatile("v_accvgpr_read_b32 %0, a200" : "=v"(va[i & 7]));
                    if (FK == 2) asm volatile("v_accvgpr_write_b32 a201, %0" :: "v"(va[i & 7]));
                    if (FK == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(va[i & 7]) : "v"(va[(i + 1) & 7]), "v"(va[(i + 2) & 7]));
                    if (FK == 4) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(va[i & 7]) : "v"(va[(i + 1) & 7]), "v"(va[(i + 2) & 7]));
                    if (FK == 5) asm volatile("v_max_f32 %0, %1, %2" : "=v"(va[i & 7]) : "v"(va[(i + 1) & 7]), "v"(va[(i + 2) & 7]));
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += va[i];
    for (int i = 0; i < 4; ++i) s += (float)fh[i][0] + (float)fl[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int RD, int DG, bool BARRIER, int V0, int V1, int V2, int FK = 0, bool BAGPR = false>
void run(float* out, h8* src, unsigned long long* cyc, int nblk) {
    auto fn = k<RD, DG, BARRIER, V0, V1, V2, FK, BAGPR>;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    for (int rep = 0; rep < 2; ++rep) { fn<<<nblk, 256, LDS_BYTES>>>(out, src, cyc); (void)hipDeviceSynchronize(); }
    unsigned long long c[4];
    (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("reads %d  dma gap %2d  barrier %d  valu %d,%d,%d kind %d Bagpr %d: %7.1f cycles/tile  (x%.3f)  %s\n", RD, DG, (int)BARRIER, V0, V1, V2, FK, (int)BAGPR,
           (double)c[0] / TILES, (double)c[0] / TILES / 1536.0, hipGetErrorString(hipGetLastError()));
}

int main() {
    float* out; h8* src; unsigned long long* cyc;
    const int nblk = 256;
    (void)hipMalloc(&out, nblk * 256 * 4); (void)hipMalloc(&src, 1024 * 16); (void)hipMalloc(&cyc, nblk * 32);
    (void)hipMemset(src, 0, 1024 * 16);
#define R(...) run<__VA_ARGS__>(out, src, cyc, nblk)
    R(0, -1, false, 0, 0, 0);
    R(0, -1, false, 4, 4, 4, 0, false);
    R(0, -1, false, 4, 4, 4, 0, true);
    R(0, -1, false, 4, 4, 4, 1, false);
    R(0, -1, false, 4, 4, 4, 1, true);
    R(0, -1, false, 4, 4, 4, 2, false);
    R(0, -1, false, 4, 4, 4, 2, true);
    R(0, -1, false, 4, 4, 4, 3, false);
    R(0, -1, false, 4, 4, 4, 3, true);
    R(0, -1, false, 4, 4, 4, 4, true);
    R(0, -1, false, 4, 4, 4, 5, true);
    R(0, 1, true, 4, 4, 4, 0, true);
    R(0, 1, true, 4, 4, 4, 1, true);
    R(0, 1, true, 4, 4, 4, 3, true);
    return 0;
}
