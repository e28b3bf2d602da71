// Shared by the 16-bit 3x3 convolution kernels (conv_lp16.hip, conv_lp16_halo.hip): kernel parameters, tile constants, the
// MFMA 16x16x32 wrapper and the four-channel epilogue store.  Device helpers are static inline: one copy per translation unit.
#pragma once
#include <algorithm>
#include <type_traits>
#include "common.h"

namespace mmh { namespace lp16 {


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TBM = 256, TBN = 256, TBK = 64;
constexpr int ROWB = TBK * 2;                 // 128 bytes per LDS row
constexpr int STAGE = (TBM + TBN) * ROWB;     // 64 KiB

struct LpConvKP {
    const char* x;          // 16-bit activations, pixel stride cs elements
    const char* w;          // 16-bit weights [tap][N][K]
    const char* zeros;      // >= 128 zero bytes
    float* y;               // fp32 output [M][y_cs] (y16 == nullptr) ...
    char* y16;              // ... or 16-bit output [M][y_cs]
    const float* bias;
    int B, H, W, C, cs;     // input geometry (same spatial size out: stride 1, 'same' padding)
    int N, y_cs;
    int tap_sign;           // source pixel of tap (kh, kw) = output pixel + tap_sign * (kh - 1, kw - 1):
                            // +1 correlation (fprop), -1 flipped filter (dgrad)
    int reflect;            // mirror the source pixel into the image (else zero outside)
    int act, h16;
    int MT, NT;             // row / column tiles
    int tap_inner;          // k order: 1 = (channel chunk, tap), 0 = (tap, channel chunk)
    int dbg;                // timing-only ablation bits (mmh_set_option "lp16_dbg"): results wrong
    float* stats;           // conv_lp16h2_kernel fprop: per (image, half tile, channel) count / mean / M2 of the stored
                            // outputs, [B][chunks][3][N] (mmh_norm_stats_merge layout), or nullptr
    const float* addend;    // conv_lp16h2_kernel, fp32 output: y += addend (same [M][y_cs] layout) - the other gradient
                            // of a tensor with two consumers, added in the dgrad's epilogue instead of by a pass of its own
    // conv_lp16h2_kernel, 16-bit dgrad whose output is the gradient of a norm's output (the first norm of a two-conv block,
    // models/Generator.py:66-77): the epilogue also takes that norm's backward sums of the values it stores -
    // s1 = sum dz, s2 = sum dz * xhat per (group, channel), dz = keep ? g * dsc : 0 - as partials per (image, half tile),
    // nbr_part [B][chunks][2][N] (mmh_norm_bwd_reduce's partial layout), so the reduce pass over g and x is gone
    const char* nbr_x;          // the norm's input, 16-bit [M][N] contiguous
    const uint16_t* nbr_bits;   // its keep bits (16 per 8 elements, scale_shift_act's layout) or nullptr
    const float* nbr_mean;      // [groups][N]
    const float* nbr_invstd;
    float* nbr_part;            // nullptr: off
    float nbr_dsc;              // 1 / (1 - drop_p) where bits are given, else 1
    int nbr_groups;             // B (instance) or 1 (batch)
};

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == MMH_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MMH_ACT_TANH) return tanhf(v);
    return v;
}

typedef __attribute__((address_space(3))) void* lds_vp;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Four consecutive output channels of one pixel from one lane (the accumulator layout of an MFMA 16x16x32 whose FIRST
// operand is the weight fragment: row = channel 4 g4 + r, column = pixel l15): bias, activation, one 8- or 16-byte store.
template <bool H16>
__device__ __forceinline__ void store4(float* y, char* y16, size_t elem, f32x4 v, const float* bv, int act) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = v[r] + bv[r];
        v[r] = act == MMH_ACT_RELU ? (t > 0.f ? t : 0.f) : (act == MMH_ACT_TANH ? tanhf(t) : t);
    }
    if (y16) {
        if (H16) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *reinterpret_cast<h4*>(y16 + elem * 2) = o;
        } else {
            typedef __bf16 b4 __attribute__((ext_vector_type(4)));
            b4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<b4*>(y16 + elem * 2) = o;
        }
    } else {
        *reinterpret_cast<f32x4*>(y + elem) = v;
    }
}
template <bool H16>
__device__ __forceinline__ f32x4 mfma16s(bf16x8 a, bf16x8 b, f32x4 c) {
    if (H16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// halo kernel (conv_lp16_halo.hip): 16 x 16 output pixels per tile, its 18 x 18 halo at row pitch 20
constexpr int HT = 16;                      // tile edge (output pixels)
constexpr int HW_ = HT + 2;                 // halo edge
constexpr int HROWS = HW_ * HW_;            // 324 halo pixel rows
constexpr int HSTAGE_A = ((HROWS * ROWB + 1023) / 1024) * 1024;     // 41984
constexpr int HSTAGE_B = TBN * ROWB;                                 // 32768
constexpr int HROUNDS = (HROWS + 63) / 64;                           // 6 DMA instructions per wave and chunk
constexpr int HP2 = 20;                                     // halo pitch
constexpr int HROWS2 = HW_ * HP2;                           // 360 LDS rows per stage
constexpr int HSTAGE_A2 = HROWS2 * ROWB;                    // 46080 = 45 KiB
constexpr int HROUNDS2 = (HROWS2 + 63) / 64;                // 6

typedef const bf16x8 __attribute__((address_space(3))) * lds_frag_p;
__device__ __forceinline__ bf16x8 lds_frag(unsigned addr) { return *reinterpret_cast<lds_frag_p>(addr); }

// conv_lp16_halo.hip: the halo kernel on a filled parameter block (MT / NT as for the row-tile kernels; the launcher sets
// its own tile count).  mode 0 fprop | 1 zero-pad dgrad | 2 reflect-fold dgrad; solo: one wave per SIMD (A/B builds only)
int launch_conv_lp16_halo(const LpConvKP& p, const mmh_conv_desc* d, int mode, bool solo, hipStream_t st);
bool conv_lp16_halo_has_solo();
int clock_stamps(unsigned long long* host_pairs, int max_workgroups);      // A/B builds: per-workgroup (s_memtime, s_memrealtime) deltas

} }  // namespace mmh::lp16
