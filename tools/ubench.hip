// ubench.hip -- gfx950 VALU issue-rate microbenchmarks that decide the force kernel's shape.
// Development tool (not part of libmapn.so).  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
//
// For each instruction mix it reports wave-instructions per cycle per SIMD (from s_memtime) at
// 1, 2, 4 and 8 waves per SIMD, plus the clock the chip held (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

enum Mix { FMA = 0, PKFMA, RSQ, PAIR_SCALAR, PAIR_PK, FMA_SGPR, PKFMA_BCAST, PAIR_MFMA, MFMA4, PAIR_PK8, RSQ_PK_ALT, RSQ_PK_SEQ,
           MFMA16, MFMA32, PK16_MFMA16_1, PK16_MFMA16_2, PK16_MFMA32_1, PAIR_ACC_MFMA16, PAIR_ACC_MFMA32, PAIR_R2_MFMA16,
           DPP_WAVE_ROR, DPP_ROW_ROR, PAIR_SYM, PAIR_SYM_NOROT, PAIR_SYM8_BPERM, PAIR_SYM8_DPP, BPERM, PAIR_SYM4_LDS, PAIR_SYM8_LDS, SWIZZLE, PAIR_SYM8_SWZ, MIXES };
static const char *mix_name[MIXES] = {"v_fma_f32 x16", "v_pk_fma_f32 x16", "v_rsq_f32 x16", "pair scalar (12 ops)", "pair packed (2 bodies)", "v_fma_f32 sgpr-src x16", "v_pk_fma_f32 op_sel bcast x16", "pair packed, accumulate on mfma 4x4x1", "v_mfma_f32_4x4x1_16b x8", "8 pk + 2 rsq (no accumulate)", "8 x (v_rsq, v_pk_fma) alternating", "8 v_rsq then 8 v_pk_fma",
                                      "v_mfma_f32_16x16x4_f32 x8", "v_mfma_f32_32x32x2_f32 x4", "16 v_pk_fma + 1 mfma16x16x4", "16 v_pk_fma + 2 mfma16x16x4",
                                      "16 v_pk_fma + 1 mfma32x32x2", "pair packed, accumulate on mfma16x16x4", "pair packed, accumulate on mfma32x32x2",
                                      "pair: r^2 on mfma16x16x4, sum-form accumulate",
                                      "v_mov_b32_dpp wave_ror:1 x16", "v_mov_b32_dpp row_ror:1 x16",
                                      "pair SYMMETRIC (a_i += , b_j -= ; j-set rotates wave_ror:1)", "pair SYMMETRIC without the rotation",
                                      "pair SYMMETRIC, 8 bodies i per lane, rotation by ds_bpermute_b32", "pair SYMMETRIC, 8 bodies i per lane, rotation by dpp wave_ror", "ds_bpermute_b32 x16",
                                      "pair SYMMETRIC, 4 bodies i per lane, j and b_j in wave-private LDS", "pair SYMMETRIC, 8 bodies i per lane, j and b_j in wave-private LDS",
                                      "ds_swizzle_b32 rotate x16", "pair SYMMETRIC, 8 bodies i per lane, 6 regs moved by ds_swizzle_b32 rotate"};
static const int mix_insts[MIXES] = {16, 16, 16, 12 * 4, 13 * 2, 16, 16, 12 * 2, 8, 10 * 2, 16, 16, 8, 4, 17, 18, 17, 10 * 2 + 4, 10 * 2 + 4, 17, 16, 16, 16 * 2 + 9, 16 * 2, 16 * 4 + 9, 16 * 4 + 9, 16, 16 * 2 + 9, 16 * 4 + 9, 16, 16 * 4 + 6};      // wave-instructions per loop body
static const double mix_pairs[MIXES] = {0, 0, 0, 4, 4, 0, 0, 4, 0, 4, 0, 0, 0, 0, 0, 0, 0, 4, 4, 4, 0, 0, 8, 8, 16, 16, 0, 8, 16, 0, 16};                  // pairs per lane per loop body

template <int MIX>
__global__ __launch_bounds__(256) void ub(float *out, unsigned long long *cyc, unsigned long long *rt, int iters, float seed)
{
    typedef float f4v __attribute__((ext_vector_type(4)));
    __shared__ f4v tile[4][64];
    __shared__ float bacc_lds[4][3][64];
    __shared__ f4v bacc4[4][64];
    const int wv = threadIdx.x >> 6;
    int jslot = threadIdx.x & 63;
    tile[wv][jslot] = f4v{seed * 3.f + jslot, seed * 5.f - jslot, seed * 7.f + 0.5f * jslot, 0.f};
    bacc_lds[wv][0][jslot] = 0.f; bacc_lds[wv][1][jslot] = 0.f; bacc_lds[wv][2][jslot] = 0.f; bacc4[wv][jslot] = f4v{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = seed + i * 0.001f + threadIdx.x * 1e-6f;
    float2v p[16];
    for (int i = 0; i < 16; i++) p[i] = float2v{a[i], a[i] * 0.5f};
    float2v q[12], r[12];
    for (int i = 0; i < 12; i++) { q[i] = float2v{a[i] * 1.5f, a[i] * 0.25f}; r[i] = float2v{0.f, 0.f}; }
    const float m = 0.999f, c = 0.0001f;
    float sx = seed * 3.f, sy = seed * 5.f, sz = seed * 7.f, soft2 = 25.f;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sx) : "v"(seed * 3.f));
    f4v macc[4];
    for (int i = 0; i < 4; i++) macc[i] = f4v{0.f, 0.f, 0.f, 0.f};
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v bacc[2];
    for (int i = 0; i < 2; i++) for (int q = 0; q < 16; q++) bacc[i][q] = 0.f;
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MIX == FMA) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (MIX == FMA_SGPR) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(sx));
        } else if (MIX == PKFMA) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 15]), "v"(p[(i + 2) & 15]));
        } else if (MIX == PKFMA_BCAST) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(p[(i + 1) & 15]), "v"(p[(i + 2) & 15]));
        } else if (MIX == RSQ) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
        } else if (MIX == PAIR_SCALAR) {
            // 4 bodies i per lane, one j: 12 ops each, all-VGPR scalar f32
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float dx, dy, dz, d, inv, i3;
                asm volatile(
                    "v_sub_f32 %0, %9, %6\n v_sub_f32 %1, %10, %7\n v_sub_f32 %2, %11, %8\n"
                    "v_fma_f32 %3, %0, %0, %12\n v_fma_f32 %3, %1, %1, %3\n v_fma_f32 %3, %2, %2, %3\n"
                    "v_rsq_f32 %4, %3\n v_mul_f32 %5, %4, %4\n v_mul_f32 %5, %5, %4\n"
                    : "=&v"(dx), "=&v"(dy), "=&v"(dz), "=&v"(d), "=&v"(inv), "=&v"(i3)
                    : "v"(a[3 * k]), "v"(a[3 * k + 1]), "v"(a[3 * k + 2]), "v"(sx), "v"(sy), "v"(sz), "v"(soft2));
                asm volatile("v_fma_f32 %0, %3, %6, %0\n v_fma_f32 %1, %4, %6, %1\n v_fma_f32 %2, %5, %6, %2"
                             : "+v"(p[k].x), "+v"(p[k].y), "+v"(p[k + 4].x) : "v"(dx), "v"(dy), "v"(dz), "v"(i3));
            }
        } else if (MIX == PAIR_PK) {
            // 2 bodies per packed op, one j, x2 -> 4 pairs: 3 pk_add, 3 pk_fma, 2 rsq, 2 pk_mul, 3 pk_fma = 13
#pragma unroll
            for (int k = 0; k < 2; k++) {
                float2v dx, dy, dz, d, inv, i3;
                asm volatile(
                    "v_pk_add_f32 %0, %9, %6 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %10, %7 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %11, %8 neg_lo:[0,1] neg_hi:[0,1]\n"
                    "v_pk_fma_f32 %3, %0, %0, %12\n v_pk_fma_f32 %3, %1, %1, %3\n v_pk_fma_f32 %3, %2, %2, %3\n"
                    : "=&v"(dx), "=&v"(dy), "=&v"(dz), "=&v"(d), "=&v"(inv), "=&v"(i3)
                    : "v"(p[3 * k]), "v"(p[3 * k + 1]), "v"(p[3 * k + 2]), "v"(p[12]), "v"(p[13]), "v"(p[14]), "v"(p[15]));
                asm volatile("v_rsq_f32 %0, %2\n v_rsq_f32 %1, %3" : "=&v"(inv.x), "=&v"(inv.y) : "v"(d.x), "v"(d.y));
                asm volatile("v_pk_mul_f32 %0, %1, %1\n v_pk_mul_f32 %0, %0, %1" : "=&v"(i3) : "v"(inv));
                asm volatile("v_pk_fma_f32 %0, %3, %6, %0\n v_pk_fma_f32 %1, %4, %6, %1\n v_pk_fma_f32 %2, %5, %6, %2"
                             : "+v"(p[6 + k]), "+v"(p[8 + k]), "+v"(p[10 + k]) : "v"(dx), "v"(dy), "v"(dz), "v"(i3));
            }
        }
        if (MIX == PAIR_MFMA || MIX == PAIR_PK8) {
            // 2 bodies per packed op, one j, x2: 3 pk_add, 3 pk_fma, 2 rsq, 2 pk_mul, then the
            // accumulation as one v_mfma_f32_4x4x1_16b per body (A = s of the 64 lanes' bodies,
            // B = (x_j, y_j, z_j, 1) by lane % 4) instead of 3 pk_fma
#pragma unroll
            for (int k = 0; k < 2; k++) {
                float2v dx, dy, dz, d, inv, i3;
                asm volatile(
                    "v_pk_add_f32 %0, %9, %6 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %10, %7 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %11, %8 neg_lo:[0,1] neg_hi:[0,1]\n"
                    "v_pk_fma_f32 %3, %0, %0, %12\n v_pk_fma_f32 %3, %1, %1, %3\n v_pk_fma_f32 %3, %2, %2, %3\n"
                    : "=&v"(dx), "=&v"(dy), "=&v"(dz), "=&v"(d), "=&v"(inv), "=&v"(i3)
                    : "v"(p[3 * k]), "v"(p[3 * k + 1]), "v"(p[3 * k + 2]), "v"(p[12]), "v"(p[13]), "v"(p[14]), "v"(p[15]));
                asm volatile("v_rsq_f32 %0, %2\n v_rsq_f32 %1, %3" : "=&v"(inv.x), "=&v"(inv.y) : "v"(d.x), "v"(d.y));
                asm volatile("v_pk_mul_f32 %0, %1, %1\n v_pk_mul_f32 %0, %0, %1" : "=&v"(i3) : "v"(inv));
                if (MIX == PAIR_MFMA) {
                    macc[2 * k] = __builtin_amdgcn_mfma_f32_4x4x1f32(i3.x, a[15], macc[2 * k], 0, 0, 0);
                    macc[2 * k + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(i3.y, a[15], macc[2 * k + 1], 0, 0, 0);
                } else {
                    asm volatile("" :: "v"(i3));
                }
            }
        } else if (MIX == RSQ_PK_ALT) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[8 + (i & 3)]), "v"(p[12 + (i & 3)]));
            }
        } else if (MIX == RSQ_PK_SEQ) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[8 + (i & 3)]), "v"(p[12 + (i & 3)]));
        } else if (MIX == MFMA16) {
            // f32-input MFMA, 1024 MACs: 32 cycles/SIMD back to back (MI355X_MICROARCH.md cycle constants)
#pragma unroll
            for (int i = 0; i < 8; i++) macc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[15], macc[i & 3], 0, 0, 0);
        } else if (MIX == MFMA32) {
#pragma unroll
            for (int i = 0; i < 4; i++) bacc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], a[15], bacc[i & 1], 0, 0, 0);
        } else if (MIX == PK16_MFMA16_1 || MIX == PK16_MFMA16_2 || MIX == PK16_MFMA32_1) {
            // does the matrix pipe run BESIDE the packed VALU stream of the same wave / SIMD?
            // 16 independent v_pk_fma_f32 (74 cycles alone) with 1 or 2 independent MFMAs in the middle
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[8 + (i & 3)]), "v"(p[12 + (i & 3)]));
            if (MIX == PK16_MFMA32_1) bacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], a[15], bacc[0], 0, 0, 0);
            else macc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[15], macc[0], 0, 0, 0);
            if (MIX == PK16_MFMA16_2) macc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], a[15], macc[1], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[8 + (i & 3)]), "v"(p[12 + (i & 3)]));
        } else if (MIX == PAIR_ACC_MFMA16 || MIX == PAIR_ACC_MFMA32) {
            // the accumulation a_i = sum_j s_ij (x_j,y_j,z_j,1) recast on the LARGE f32 MFMA shapes.  A 4-column B
            // fills 4 of the 16 (32) N columns; the block-diagonal form (k = i-group, n = (group, component))
            // reaches 256 useful MACs of 1024 (2048): ONE j for the wave's 64 bodies per 16x16x4 (32x32x2).
            // Two bodies per lane -> 2 MFMAs per j.  VALU side as PAIR_PK8 (no v_pk_fma accumulate).
#pragma unroll
            for (int k = 0; k < 2; k++) {
                float2v dx, dy, dz, d, inv, i3;
                asm volatile(
                    "v_pk_add_f32 %0, %9, %6 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %10, %7 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %11, %8 neg_lo:[0,1] neg_hi:[0,1]\n"
                    "v_pk_fma_f32 %3, %0, %0, %12\n v_pk_fma_f32 %3, %1, %1, %3\n v_pk_fma_f32 %3, %2, %2, %3\n"
                    : "=&v"(dx), "=&v"(dy), "=&v"(dz), "=&v"(d), "=&v"(inv), "=&v"(i3)
                    : "v"(p[3 * k]), "v"(p[3 * k + 1]), "v"(p[3 * k + 2]), "v"(p[12]), "v"(p[13]), "v"(p[14]), "v"(p[15]));
                asm volatile("v_rsq_f32 %0, %2\n v_rsq_f32 %1, %3" : "=&v"(inv.x), "=&v"(inv.y) : "v"(d.x), "v"(d.y));
                asm volatile("v_pk_mul_f32 %0, %1, %1\n v_pk_mul_f32 %0, %0, %1" : "=&v"(i3) : "v"(inv));
                if (MIX == PAIR_ACC_MFMA16) {
                    macc[2 * k] = __builtin_amdgcn_mfma_f32_16x16x4f32(i3.x, a[15], macc[2 * k], 0, 0, 0);
                    macc[2 * k + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(i3.y, a[15], macc[2 * k + 1], 0, 0, 0);
                } else {
                    bacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(i3.x, a[15], bacc[0], 0, 0, 0);
                    bacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(i3.y, a[15], bacc[1], 0, 0, 0);
                }
            }
        } else if (MIX == PAIR_R2_MFMA16) {
            // r^2_ij = |x_i|^2 + soft^2 + |x_j|^2 - 2 x_i.x_j as ONE 16x16x4 MFMA per 16 i x 16 j tile (K = x,y,z,1;
            // the i-only term rides in the C operand): a lane then holds d for 4 bodies i against ITS j.  VALU:
            // 4 v_rsq, 4 v_pk_mul (inv^3), sum-form accumulate sum_j s x_j and sum_j s: 6 v_pk_fma + 2 v_pk_add.
            // (Numerically unusable: |x|^2 ~ 5e5 against soft^2 = 25 -- tools/mfma_recast_error.py.)
            f4v dt = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[15], macc[3], 0, 0, 0);
            float2v i0, i1, s0, s1;
            asm volatile("v_rsq_f32 %0, %4\n v_rsq_f32 %1, %5\n v_rsq_f32 %2, %6\n v_rsq_f32 %3, %7"
                         : "=&v"(i0.x), "=&v"(i0.y), "=&v"(i1.x), "=&v"(i1.y) : "v"(dt.x), "v"(dt.y), "v"(dt.z), "v"(dt.w));
            asm volatile("v_pk_mul_f32 %0, %2, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %3, %3\n v_pk_mul_f32 %1, %1, %3"
                         : "=&v"(s0), "=&v"(s1) : "v"(i0), "v"(i1));
            asm volatile("v_pk_fma_f32 %0, %6, %8, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %6, %8, %1 op_sel:[0,1,0]\n v_pk_fma_f32 %2, %6, %9, %2 op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %3, %7, %8, %3 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %4, %7, %8, %4 op_sel:[0,1,0]\n v_pk_fma_f32 %5, %7, %9, %5 op_sel_hi:[1,0,1]\n"
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]) : "v"(s0), "v"(s1), "v"(p[12]), "v"(p[13]));
            asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3" : "+v"(p[6]), "+v"(p[7]) : "v"(s0), "v"(s1));
        } else if (MIX == DPP_WAVE_ROR || MIX == DPP_ROW_ROR) {
#pragma unroll
            for (int i = 0; i < 16; i++)
                a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), MIX == DPP_WAVE_ROR ? 0x13C : 0x121, 0xf, 0xf, false));
        } else if (MIX == PAIR_SYM || MIX == PAIR_SYM_NOROT) {
            // Newton's third law inside a wave: a lane owns 4 bodies i (two packed pairs, accumulators p[6..11]) and
            // carries ONE body j (position a[0..2], its accumulators kept as packed lo/hi partial sums p[12..14]).
            // One step: 2 packed evaluations (14 v_pk + 2 v_rsq each: a_i += r s, b_j -= r s) = 8 interactions per
            // lane, then the j body and its accumulators rotate one lane (9 v_mov_b32_dpp wave_ror:1).
            const float2v sft = float2v{soft2, soft2};
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const float2v dx = a[0] - p[3 * k], dy = a[1] - p[3 * k + 1], dz = a[2] - p[3 * k + 2];
                float2v d = __builtin_elementwise_fma(dx, dx, sft);
                d = __builtin_elementwise_fma(dy, dy, d);
                d = __builtin_elementwise_fma(dz, dz, d);
                float2v inv;
                inv.x = __builtin_amdgcn_rsqf(d.x); inv.y = __builtin_amdgcn_rsqf(d.y);
                const float2v i3 = inv * inv * inv;
                p[6 + 3 * k] = __builtin_elementwise_fma(dx, i3, p[6 + 3 * k]);
                p[7 + 3 * k] = __builtin_elementwise_fma(dy, i3, p[7 + 3 * k]);
                p[8 + 3 * k] = __builtin_elementwise_fma(dz, i3, p[8 + 3 * k]);
                p[12] = __builtin_elementwise_fma(-dx, i3, p[12]);
                p[13] = __builtin_elementwise_fma(-dy, i3, p[13]);
                p[14] = __builtin_elementwise_fma(-dz, i3, p[14]);
            }
            if (MIX == PAIR_SYM) {
#define ROR1(v) v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, false))
                ROR1(a[0]); ROR1(a[1]); ROR1(a[2]);
                ROR1(p[12].x); ROR1(p[12].y); ROR1(p[13].x); ROR1(p[13].y); ROR1(p[14].x); ROR1(p[14].y);
#undef ROR1
            }
        } else if (MIX == BPERM) {
            const int addr = (int)(((threadIdx.x & 63u) + 1u) & 63u) * 4;
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, a[i])));
        } else if (MIX == PAIR_SYM8_BPERM || MIX == PAIR_SYM8_DPP) {
            // as PAIR_SYM with 8 bodies i per lane (4 packed pairs: positions q[0..11], accumulators r[0..11]): 4 packed
            // evaluations = 16 interactions per lane per rotation step, the 9 rotating registers moved either through
            // the LDS crossbar (ds_bpermute_b32: no VALU cycles) or by DPP
            const float2v sft = float2v{soft2, soft2};
            const int addr = (int)(((threadIdx.x & 63u) + 1u) & 63u) * 4;
            float nx, ny, nz;
            if (MIX == PAIR_SYM8_BPERM) {           // the j position does not change during the step: rotate it up front
                nx = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, a[0])));
                ny = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, a[1])));
                nz = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, a[2])));
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float2v dx = a[0] - q[3 * k], dy = a[1] - q[3 * k + 1], dz = a[2] - q[3 * k + 2];
                float2v d = __builtin_elementwise_fma(dx, dx, sft);
                d = __builtin_elementwise_fma(dy, dy, d);
                d = __builtin_elementwise_fma(dz, dz, d);
                float2v inv;
                inv.x = __builtin_amdgcn_rsqf(d.x); inv.y = __builtin_amdgcn_rsqf(d.y);
                const float2v i3 = inv * inv * inv;
                r[3 * k] = __builtin_elementwise_fma(dx, i3, r[3 * k]);
                r[3 * k + 1] = __builtin_elementwise_fma(dy, i3, r[3 * k + 1]);
                r[3 * k + 2] = __builtin_elementwise_fma(dz, i3, r[3 * k + 2]);
                p[12] = __builtin_elementwise_fma(-dx, i3, p[12]);
                p[13] = __builtin_elementwise_fma(-dy, i3, p[13]);
                p[14] = __builtin_elementwise_fma(-dz, i3, p[14]);
            }
            if (MIX == PAIR_SYM8_BPERM) {
#define BP(v) v = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)))
                a[0] = nx; a[1] = ny; a[2] = nz;
                BP(p[12].x); BP(p[12].y); BP(p[13].x); BP(p[13].y); BP(p[14].x); BP(p[14].y);
#undef BP
            } else {
#define ROR1(v) v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, false))
                ROR1(a[0]); ROR1(a[1]); ROR1(a[2]);
                ROR1(p[12].x); ROR1(p[12].y); ROR1(p[13].x); ROR1(p[13].y); ROR1(p[14].x); ROR1(p[14].y);
#undef ROR1
            }
        } else if (MIX == PAIR_SYM4_LDS || MIX == PAIR_SYM8_LDS) {
            // no rotation at all: the 64 bodies j of the block sit in a wave-private LDS tile; at step k lane l reads
            // body (l + k) % 64 (one ds_read_b128, distinct address per lane) and adds the reaction it accumulated over
            // its bodies i to that body's wave-private LDS accumulator (3 ds_add_f32, one lane per address per step ->
            // the order of the adds is the program order: deterministic)
            constexpr int NE = MIX == PAIR_SYM8_LDS ? 4 : 2;
            const float2v sft = float2v{soft2, soft2};
            const f4v pj = tile[wv][jslot];
            float2v bx, by, bz;
#pragma unroll
            for (int k = 0; k < NE; k++) {
                const float2v dx = pj.x - q[3 * k], dy = pj.y - q[3 * k + 1], dz = pj.z - q[3 * k + 2];
                float2v d = __builtin_elementwise_fma(dx, dx, sft);
                d = __builtin_elementwise_fma(dy, dy, d);
                d = __builtin_elementwise_fma(dz, dz, d);
                float2v inv;
                inv.x = __builtin_amdgcn_rsqf(d.x); inv.y = __builtin_amdgcn_rsqf(d.y);
                const float2v i3 = inv * inv * inv;
                r[3 * k] = __builtin_elementwise_fma(dx, i3, r[3 * k]);
                r[3 * k + 1] = __builtin_elementwise_fma(dy, i3, r[3 * k + 1]);
                r[3 * k + 2] = __builtin_elementwise_fma(dz, i3, r[3 * k + 2]);
                if (k == 0) { bx = dx * i3; by = dy * i3; bz = dz * i3; }
                else { bx = __builtin_elementwise_fma(dx, i3, bx); by = __builtin_elementwise_fma(dy, i3, by); bz = __builtin_elementwise_fma(dz, i3, bz); }
            }
            // read-modify-write by the VALU, no LDS atomic (ds_add_f32 measured ~195 cycles per wave-instruction): only
            // this lane touches body jslot's accumulator in this step and a wave's LDS operations execute in order
            f4v acc = bacc4[wv][jslot];
            acc.x -= bx.x + bx.y; acc.y -= by.x + by.y; acc.z -= bz.x + bz.y;
            bacc4[wv][jslot] = acc;
            jslot = (jslot + 1) & 63;
        } else if (MIX == SWIZZLE) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, a[i]), 0xC000 | (1 << 5)));
        } else if (MIX == PAIR_SYM8_SWZ) {
            // what mapn_sym.hip does per step, with the six travelling registers moved by ds_swizzle_b32 (rotate within
            // 32 lanes: one source register, no address) instead of ds_bpermute_b32
            const float2v sft = float2v{soft2, soft2};
#define SW(v) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0xC000 | (1 << 5)))
            const float nx = SW(a[0]), ny = SW(a[1]), nz = SW(a[2]);
            float2v rx = p[12], ry = p[13], rz = p[14];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float2v dx = a[0] - q[3 * k], dy = a[1] - q[3 * k + 1], dz = a[2] - q[3 * k + 2];
                float2v d = __builtin_elementwise_fma(dx, dx, sft);
                d = __builtin_elementwise_fma(dy, dy, d);
                d = __builtin_elementwise_fma(dz, dz, d);
                float2v inv;
                inv.x = __builtin_amdgcn_rsqf(d.x); inv.y = __builtin_amdgcn_rsqf(d.y);
                const float2v i3 = inv * inv * inv;
                r[3 * k] = __builtin_elementwise_fma(dx, i3, r[3 * k]);
                r[3 * k + 1] = __builtin_elementwise_fma(dy, i3, r[3 * k + 1]);
                r[3 * k + 2] = __builtin_elementwise_fma(dz, i3, r[3 * k + 2]);
                rx = __builtin_elementwise_fma(-dx, i3, rx);
                ry = __builtin_elementwise_fma(-dy, i3, ry);
                rz = __builtin_elementwise_fma(-dz, i3, rz);
            }
            a[0] = nx; a[1] = ny; a[2] = nz;
            p[12].x = SW(rx.x + rx.y); p[13].x = SW(ry.x + ry.y); p[14].x = SW(rz.x + rz.y);
#undef SW
        } else if (MIX == MFMA4) {
#pragma unroll
            for (int i = 0; i < 8; i++) macc[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[i], a[15], macc[i & 3], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
    for (int i = 0; i < 12; i++) s += q[i].x + r[i].x + r[i].y;
    s += bacc_lds[wv][0][jslot] + bacc4[wv][jslot].x + bacc4[wv][jslot].y + bacc4[wv][jslot].z;
    for (int i = 0; i < 4; i++) s += macc[i].x + macc[i].y + macc[i].z + macc[i].w;
    for (int i = 0; i < 2; i++) for (int q = 0; q < 16; q++) s += bacc[i][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        cyc[w] = t1 - t0;
        rt[w] = r1 - r0;
    }
}

static bool g_json = false;     // --ab: one JSON object per line instead of the table (tests/test_gpu_mfma_ab.py parses it)

template <int MIX>
static void run(int waves_per_simd, int iters)
{
    int dev; CHECK(hipGetDevice(&dev));
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, dev));
    const int cus = pr.multiProcessorCount;
    const int blocks = cus * waves_per_simd;      // 256 threads = 4 waves = one wave per SIMD
    const int waves = blocks * 4;
    float *out; unsigned long long *cyc, *rt;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
    CHECK(hipMalloc(&cyc, 8 * waves)); CHECK(hipMalloc(&rt, 8 * waves));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(ub<MIX>, dim3(blocks), dim3(256), 0, 0, out, cyc, rt, iters, 1.0f + rep);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hc(waves), hr(waves);
    CHECK(hipMemcpy(hc.data(), cyc, 8 * waves, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hr.data(), rt, 8 * waves, hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end()); std::sort(hr.begin(), hr.end());
    const double cmed = (double)hc[waves / 2], rmed = (double)hr[waves / 2];
    const double insts = (double)iters * mix_insts[MIX];
    // per SIMD: waves_per_simd waves, each issuing `insts` wave-instructions in cmed cycles
    const double ipc = insts * waves_per_simd / cmed;
    const double ghz = cmed / rmed * 0.1;
    if (g_json) {
        // cycles_per_body_per_simd from the launch WALL (a kernel that needs more registers than `waves_per_simd` waves leave
        // runs its workgroups in turns; the wall accounts for that, the in-kernel stamps of one wave do not)
        const double pairs = (double)iters * mix_pairs[MIX] * 64.0 * waves;
        printf("{\"mix\": \"%s\", \"id\": %d, \"waves_per_simd\": %d, \"iters\": %d, \"insts_per_body\": %d, \"wall_ms\": %.5f, \"clk_ghz\": %.4f, "
               "\"cycles_per_body_per_simd\": %.3f, \"pairs_per_s\": %.5e}\n", mix_name[MIX], (int)MIX, waves_per_simd, iters, mix_insts[MIX], ms, ghz,
               ms * 1e-3 * ghz * 1e9 / ((double)iters * waves_per_simd), pairs / (ms * 1e-3));
        CHECK(hipFree(out)); CHECK(hipFree(cyc)); CHECK(hipFree(rt));
        return;
    }
    printf("%-32s waves/SIMD=%d  cyc/inst/SIMD=%6.3f  inst/cyc/SIMD=%5.3f  clk=%.2f GHz  wall=%.3f ms", mix_name[MIX], waves_per_simd, 1.0 / ipc, ipc, ghz, ms);
    if (mix_pairs[MIX] > 0) {
        const double pairs = (double)iters * mix_pairs[MIX] * 64.0 * waves;
        printf("  pairs/s=%.3e (%.1f%% of %0.1f TF @20flop)", pairs / (ms * 1e-3), 100.0 * 20.0 * pairs / (ms * 1e-3) / (cus * 2.4e9 * 256), cus * 2.4e9 * 256 / 1e12);
    }
    printf("\n");
    CHECK(hipFree(out)); CHECK(hipFree(cyc)); CHECK(hipFree(rt));
}

template <int MIX>
static void sweep(int iters)
{
    for (int w : {1, 2, 4, 8}) run<MIX>(w, iters);
}

int main(int argc, char **argv)
{
    // ubench --ab [iters]: the MFMA-against-packed-VALU A/B of BASELINE configs[4] only, as JSON lines
    if (argc > 1 && std::string(argv[1]) == "--ab") {
        const int it = argc > 2 ? atoi(argv[2]) : 20000;
        g_json = true;
        for (int w : {2, 8}) {
            run<PKFMA>(w, it); run<MFMA4>(w, it); run<MFMA16>(w, it); run<MFMA32>(w, it);
            run<PK16_MFMA16_1>(w, it); run<PK16_MFMA16_2>(w, it); run<PK16_MFMA32_1>(w, it);
            run<PAIR_PK>(w, it); run<PAIR_PK8>(w, it); run<PAIR_MFMA>(w, it); run<PAIR_ACC_MFMA16>(w, it); run<PAIR_ACC_MFMA32>(w, it); run<PAIR_R2_MFMA16>(w, it);
        }
        return 0;
    }
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    printf("device: %s  arch=%s  CUs=%d  clock=%d kHz  wave=%d\n", pr.name, pr.gcnArchName, pr.multiProcessorCount, pr.clockRate, pr.warpSize);
    if (getenv("UBENCH_SYM_ONLY")) { sweep<PAIR_PK>(iters); sweep<DPP_WAVE_ROR>(iters); sweep<DPP_ROW_ROR>(iters); sweep<PAIR_SYM>(iters); sweep<PAIR_SYM_NOROT>(iters); sweep<PAIR_SYM8_BPERM>(iters); sweep<SWIZZLE>(iters); sweep<PAIR_SYM8_SWZ>(iters); return 0; }
    sweep<FMA>(iters); sweep<FMA_SGPR>(iters); sweep<PKFMA>(iters); sweep<PKFMA_BCAST>(iters); sweep<RSQ>(iters);
    sweep<PAIR_SCALAR>(iters); sweep<PAIR_PK>(iters); sweep<PAIR_PK8>(iters); sweep<PAIR_MFMA>(iters); sweep<MFMA4>(iters); sweep<RSQ_PK_ALT>(iters); sweep<RSQ_PK_SEQ>(iters);
    sweep<MFMA16>(iters); sweep<MFMA32>(iters); sweep<PK16_MFMA16_1>(iters); sweep<PK16_MFMA16_2>(iters); sweep<PK16_MFMA32_1>(iters);
    sweep<PAIR_ACC_MFMA16>(iters); sweep<PAIR_ACC_MFMA32>(iters); sweep<PAIR_R2_MFMA16>(iters);
    sweep<DPP_WAVE_ROR>(iters); sweep<DPP_ROW_ROR>(iters); sweep<PAIR_SYM>(iters); sweep<PAIR_SYM_NOROT>(iters); sweep<BPERM>(iters); sweep<PAIR_SYM8_BPERM>(iters); sweep<PAIR_SYM8_DPP>(iters); sweep<PAIR_SYM4_LDS>(iters); sweep<PAIR_SYM8_LDS>(iters); sweep<SWIZZLE>(iters); sweep<PAIR_SYM8_SWZ>(iters);
    return 0;
}
