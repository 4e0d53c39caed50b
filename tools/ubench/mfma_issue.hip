// Micro-benchmark: what does ONE wave per SIMD sustain around a chain of v_mfma_f32_32x32x16_f16?
// Each k-step = 3 dependent MFMAs (as the render kernel's hi.hi + lo.hi + hi.lo) with NV filler instructions of one KIND
// behind each MFMA; optional 2 ds_read_b128 per k-step feeding the A operands 3 k-steps later.
// Prints shader cycles per MFMA (32 = matrix pipe saturated).
// Build+run on the GPU box: hipcc -w -O3 --offload-arch=gfx950 mfma_issue.hip -o /tmp/mfma_issue && /tmp/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int KSTEPS = 16;
constexpr int TILES = 1000;
constexpr int LDS_BYTES = 140 * 1024;

enum Kind { FMA = 0, CVTPK = 1, ACCRD = 2, ACCWR = 3, SNOP = 4, SMOV = 5, FMA_DEP = 6, MAXF = 7, MIX = 8, PKFMA = 9, CVT1 = 10, FMA_DEP2 = 11,
            DSREAD = 12, CVTPK_DEP = 13, MOV = 14, PKMAXH = 15, SADD = 16, GLDS = 17, DSREAD_ASM = 18, GLDS_K2 = 19, GLDS_K2S = 20, GLDS_K2F = 21, GLDS_X0 = 22, GLDS_W0 = 23, GLDS_W0x4 = 24, GLDS_ROT = 25 };

template <int KIND>
__device__ __forceinline__ void filler(float& a, float& b, unsigned& u) {
    if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    if (KIND == FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    if (KIND == CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(a), "v"(b));
    if (KIND == ACCRD) asm volatile("v_accvgpr_read_b32 %0, a100" : "=v"(u));
    if (KIND == ACCWR) asm volatile("v_accvgpr_write_b32 a101, %0" :: "v"(u));
    if (KIND == SNOP) asm volatile("s_nop 0");
    if (KIND == SMOV) asm volatile("s_mov_b32 s90, s91" ::: "s90");
    if (KIND == SADD) asm volatile("s_add_u32 s90, s91, 4" ::: "s90", "scc");
    if (KIND == MAXF) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(u));
    if (KIND == MIX) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(a) : "v"(u), "v"(b));
    if (KIND == CVT1) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u) : "v"(a));
    if (KIND == CVTPK_DEP) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(u) : "v"(b));
    if (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(a));
    if (KIND == PKMAXH) asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(u) : "v"(a), "v"(b));
}
template <int KIND>
__device__ __forceinline__ void filler2(float2& a, float2& b, float& dep, float& other) {
    if (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %2" : "=v"(a) : "v"(b), "v"(a));
    if (KIND == FMA_DEP2) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(dep) : "v"(other)); }
}

template <int NV, int KIND, bool READS, bool MFMA>
__global__ __launch_bounds__(256, 1) void k(float* out, const h8* src, unsigned long long* cyc) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned mask0 = __builtin_amdgcn_readfirstlane(wave_id == 0 ? ~0u : 0u);
    unsigned maskq[4];
    for (int i = 0; i < 4; ++i) maskq[i] = __builtin_amdgcn_readfirstlane(wave_id == i ? ~0u : 0u);
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) reinterpret_cast<h8*>(lds)[i] = src[i & 1023];
    __syncthreads();
    h8 xh = src[lane], xl = src[lane + 64];
    h8 fh[4], fl[4];
    for (int i = 0; i < 4; ++i) { fh[i] = src[lane + 128 + 64 * i]; fl[i] = src[lane + 384 + 64 * i]; }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float va[8];
    unsigned vu[8];
    float2 vp[4];
    h8 dsink[4];
    asm volatile("s_mov_b32 m0, %0" :: "s"(100 * 1024));
    for (int i = 0; i < 4; ++i) vp[i] = make_float2(lane * 0.5f, i);
    for (int i = 0; i < 8; ++i) { va[i] = (float)lane * 0.001f + i; vu[i] = i; }
    const char* base = lds + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < TILES; ++t) {
        const char* cb = base + (t & 1) * 65536;
#pragma unroll
        for (int q = 0; q < KSTEPS; ++q) {
            const int use = q & 3, slot = (q + 3) & 3;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if ((KIND == GLDS || KIND == DSREAD_ASM) && q == KSTEPS - 1 && m == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(m == 1 ? fl[use] : fh[use], m == 2 ? xl : xh, acc, 0, 0, 0);
                if (READS && m == 0) {
                    fl[slot] = *reinterpret_cast<const h8*>(cb + (2 * q + 1) * 1024);
                    fh[slot] = *reinterpret_cast<const h8*>(cb + (2 * q) * 1024);
                }
                if (KIND == GLDS_K2 && (q & 1) == 0 && m == 0)
                    asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(lane * 16), "s"(src) : "memory");
                if (KIND == GLDS_K2F && (q & 1) == 0 && m == 0) {   // with 4 independent VALU in the same gap
                    asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(lane * 16), "s"(src) : "memory");
                }
                if (KIND == GLDS_K2S && m == (wave_id % 3) && (q & 1) == (wave_id / 3))
                    asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(lane * 16), "s"(src) : "memory");
                if (KIND == GLDS_X0 && (q & 1) == 0 && m == 0)      // every wave, EXEC = 0: what does a skipped DMA cost?
                    asm volatile("s_mov_b64 exec, 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_mov_b64 exec, -1" :: "v"(lane * 16), "s"(src) : "memory");
                if (KIND == GLDS_W0 && (q & 1) == 0 && m == 0)      // only wave 0 live
                    asm volatile("s_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, %2\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_mov_b64 exec, -1" :: "v"(lane * 16), "s"(src), "s"(__builtin_amdgcn_readfirstlane(mask0)) : "memory");
                if (KIND == GLDS_W0x4 && m == 0) {                  // only wave 0 live, 2 per k-step = the whole CU's rate from one wave
                    asm volatile("s_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, %2\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\ts_mov_b64 exec, -1" :: "v"(lane * 16), "s"(src), "s"(__builtin_amdgcn_readfirstlane(mask0)) : "memory");
                }
                if (KIND == GLDS_ROT && m == 0) {                   // every k-step a slot, live for wave (q/2 ... ) : rotating owner, 2 k-steps apart
                    asm volatile("s_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, %2\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_mov_b64 exec, -1" :: "v"(lane * 16), "s"(src), "s"(__builtin_amdgcn_readfirstlane(maskq[q & 3])) : "memory");
                }
                if ((KIND >= GLDS_X0) && q == KSTEPS - 1 && m == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((KIND == GLDS_K2 || KIND == GLDS_K2S || KIND == GLDS_K2F) && q == KSTEPS - 1 && m == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < (KIND >= GLDS_K2 ? (KIND == GLDS_K2F ? 4 : 0) : NV); ++i) {
                    if (KIND == PKFMA) filler2<KIND>(vp[i & 3], vp[(i + 1) & 3], va[0], va[1]);
                    else if (KIND == FMA_DEP2) filler2<KIND>(vp[0], vp[1], va[i & 1], va[2]);
                    else if (KIND == GLDS) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(lane * 16), "s"(src) : "memory");
                    else if (KIND == DSREAD_ASM) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(dsink[i & 3]) : "v"(lane * 16) : "memory");
                    else if (KIND == DSREAD) { h8 t = *reinterpret_cast<const h8*>(cb + ((i + 3 * m + 7 * q) & 63) * 1024); asm volatile("" :: "v"(t)); }
                    else filler<(KIND >= GLDS_K2 ? FMA : KIND)>(va[KIND == FMA_DEP ? 0 : (i & 7)], va[(i + 1) & 7], vu[i & 7]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += va[i] + (float)vu[i];
    for (int i = 0; i < 4; ++i) s += vp[i].x + vp[i].y;
    if (KIND == DSREAD_ASM) for (int i = 0; i < 4; ++i) s += (float)dsink[i][0];
    for (int i = 0; i < 4; ++i) s += (float)fh[i][0] + (float)fl[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int KIND, bool READS, bool MFMA>
void run(const char* name, float* out, h8* src, unsigned long long* cyc, int nblk) {
    auto fn = k<NV, KIND, READS, MFMA>;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    fn<<<nblk, 256, LDS_BYTES>>>(out, src, cyc);
    (void)hipDeviceSynchronize();
    fn<<<nblk, 256, LDS_BYTES>>>(out, src, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long c[4];
    (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double n = (double)TILES * KSTEPS * 3;
    printf("%-34s NV=%d reads=%d : %.2f cycles per MFMA slot (%s)\n", name, NV, (int)READS, (double)c[0] / n, hipGetErrorString(hipGetLastError()));
}

#define SWEEP(KIND, NAME)                                                                                \
    run<4, KIND, false, false>(NAME " alone", out, src, cyc, nblk); run<8, KIND, false, false>(NAME " alone", out, src, cyc, nblk); \
    run<2, KIND, false, true>(NAME " +mfma", out, src, cyc, nblk); run<4, KIND, false, true>(NAME " +mfma", out, src, cyc, nblk); \
    run<6, KIND, false, true>(NAME " +mfma", out, src, cyc, nblk);

int main() {
    float* out; h8* src; unsigned long long* cyc;
    const int nblk = 256;
    (void)hipMalloc(&out, nblk * 256 * 4); (void)hipMalloc(&src, 1024 * 16); (void)hipMalloc(&cyc, nblk * 8);
    (void)hipMemset(src, 0, 1024 * 16);
    run<0, FMA, true, true>("baseline mfma + reads", out, src, cyc, nblk);
    run<1, GLDS_K2, true, true>("glds /2 k-steps, all waves", out, src, cyc, nblk);
    run<1, GLDS_X0, true, true>("glds /2 k-steps, all EXEC=0", out, src, cyc, nblk);
    run<1, GLDS_W0, true, true>("glds /2 k-steps, only wave 0 live", out, src, cyc, nblk);
    run<1, GLDS_W0x4, true, true>("2 glds / k-step, only wave 0 live", out, src, cyc, nblk);
    run<1, GLDS_ROT, true, true>("1 slot / k-step, owner rotates", out, src, cyc, nblk);
    return 0;
}
