// 16-bit 3x3 / stride-1 convolution with the activation tile held in LDS once for all nine taps ("halo kernel"): fprop, zero-pad
// dgrad and the complete dgrad of a ReflectionPad2d(1) conv (models/Generator.py:40-113, models/Discriminator.py:18-55) on the
// 256 / 512-channel stack.  The M tile is a 16 x 16 pixel block of one image; per 64-channel chunk its 18 x 18 halo is brought in
// by LDS-DMA one whole chunk (nine k-steps) ahead and the nine taps read it at shifted rows - the A operand crosses L2 -> LDS
// once instead of nine times; only the weights stream per k-step (32 KiB, L2 resident).  LDS: 2 halo + 2 weight stages = 154 KiB.
//   k order: 64-channel chunk outer, tap inner.  Halo row of output pixel (py, px), tap offset (dh, dw) in {0,1,2}^2:
//   (py + dh) * 20 + px + dw; fprop (dh, dw) = (kh, kw), dgrad (2 - kh, 2 - kw).
// The fragment reads are pipelined into the MFMA stream: 4 MFMAs on activation fragment i, then the ds_read that refills
// fragment i for the NEXT 32-deep step (rolling reuse; the weight fragments alternate between two sets); one barrier per
// k-step, in its middle.  (Round 2-4 generations of this kernel - row tiles, per-k-step address arithmetic - are in the git
// history: conv_lp16s / conv_lp16h / conv_lp16_kernel, removed in round 5.)
#include "lp16_common.h"

namespace {
using namespace mmh::lp16;

// ---------------------------------------------------------------------------------------------
// conv_lp16h2_kernel: conv_lp16h_kernel with the fragment ADDRESS arithmetic taken out of the k-loop.
// conv_lp16h_kernel recomputes, per A fragment and k-step, the halo row of the tap and its swizzle key: its loop body
// holds 142 vector-ALU instructions beside 64 MFMAs per wave, and with two waves per SIMD that is MORE vector issue
// time (2 x 142 x 4 cycles) than the MFMAs leave free (8 of every 16 cycles): the MFMA stream waits on address
// arithmetic.  Here
//   * halo row (hy, hx) sits at LDS row hy * 20 + hx (pitch 20: even, so the row's bank parity is hx & 1) and its
//     16-byte chunks are XOR-ed with hx & 6 - a key that does not depend on hy, so the lane's address depends on the
//     tap only through dw: lane_base[dw] + (wr*8 + i + dh) * 2560 - the row part is an IMMEDIATE.  hx & 6 (not
//     (row >> 1) & 7 as in conv_lp16h_kernel): ds_read_b128 serves a wave in groups {lanes 0-3, 12-15, 20-27}, ...
//     (MI355X guide), i.e. 8 rows with chunk c and the 8 rows between them with chunk c ^ 1; with (row >> 1) & 7 those
//     16 slots are distinct only when the fragment starts on an even row (dw = 0, 2) and 4 of 16 lanes collide on an
//     odd start (the 25 M conflict cycles per launch the round-2 profile could not place); hx & 6 is conflict-free
//     for every start (checked by enumeration over the real lane groups);
//   * six lane-constant A addresses (3 dw x 2 halves of the k-step) and two for B; per k-step the stage offset and
//     dh rows are added as scalars (a handful of vector adds instead of ~100) - the taps stay a run-time loop (unrolled
//     nine-fold the compiler keeps every per-tap DMA address alive and spills);
//   * LDS is addressed through 32-bit local pointers (no 64-bit flat address arithmetic).  SIGN = +1 fprop, -1 dgrad.
// Same tile, weights path, pipelining (fragments of the next half k-step requested while the current half multiplies,
// one barrier per k-step in its middle) and epilogue as conv_lp16h_kernel; LDS 2 x 45 KiB + 2 x 32 KiB = 154 KiB.


// The MFMA as inline asm with the accumulator pinned to AGPRs and updated in place.  Why: with 256 accumulator registers per
// lane (the whole AGPR file) hipcc's allocator no longer ties an MFMA's destination to its C operand - every multiply came
// back as "copy four AGPRs through VGPRs, multiply into another quad, copy back" (v_accvgpr_read / _write around each of the
// 128 MFMAs of a k-step, and 160-500 registers in scratch).  volatile: the statements keep their order, and nothing the
// compiler schedules moves across them, so the ds_reads and DMA instructions written between two multiplies stay there.
// The compiler does not know these are MFMAs: the caller keeps the hazards itself (no accumulator is read by another
// instruction class until the nops behind the k-loop; operands come from ds_reads, which the waitcnt pass does see).
template <bool H16>
__device__ __forceinline__ void mfma16s_agpr(f32x4& c, const bf16x8& a, const bf16x8& b) {
    if (H16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <bool H16>
__device__ __forceinline__ void mfma16s_vgpr(f32x4& c, const bf16x8& a, const bf16x8& b) {     // accumulator in VGPRs
    if (H16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// LDS reads and their waits as inline asm (one wave per SIMD).  hipcc's waitcnt pass puts s_waitcnt lgkmcnt(0) in front of the
// first asm statement that uses a register an LDS read is still writing - a full drain at the head of every half k-step, where
// the reads issued last (64-128 cycles of LDS latency) have nobody to hide behind.  LDS operations return in order, so the
// kernel counts: the compiler does not see these reads at all and adds no wait for them; every wait is written out below.
// (A compiler-visible LDS operation in between only makes a counted wait more conservative; there must be no scalar memory
// load in the loop - SMEM shares the counter and returns out of order - checked in the ISA: tools/isa.sh.)
template <int OFF>
__device__ __forceinline__ void lds_frag_asm(bf16x8& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ void lds_u32_asm(unsigned& d, unsigned addr) {
    asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(addr));
}
template <int N>
__device__ __forceinline__ void lgk_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N)); }

// NWC = wave columns of the workgroup (its waves are a 2 x NWC grid over the 16 x 16-pixel x 256-channel tile):
//   NWC = 4: 512 threads, two waves per SIMD, wave tile 8 rows x 16 pixels x 64 channels (128 accumulator registers);
//   NWC = 2: 256 threads, ONE wave per SIMD with the SIMD's whole register file (256 accumulator registers in AGPRs),
//            wave tile 8 rows x 16 pixels x 128 channels: 8 + 8 fragment reads per 64 MFMAs where NWC = 4 reads 8 + 4 per
//            32 (a third less LDS traffic and half the vector-ALU bookkeeping per MFMA, four waves at the barrier instead
//            of eight).  No second wave hides anything on the SIMD, so the DMA of the next stages is issued one
//            instruction at a time between the MFMA groups of half 1 (a DMA instruction costs its wave's issue slot
//            tens of cycles: MI355X guide, cycle constants).  DESIGN.md section 4.3d: why there is no fifth "producer" wave -
//            a kernel has ONE register allocation, and a fifth wave does not fit beside four 512-register waves.
template <bool H16, int SIGN, bool FOLD, int NWC, bool NBR = false>
__device__ __forceinline__ void conv_lp16h2_body(const LpConvKP& p) {
    constexpr int NWAVES = 2 * NWC;
    constexpr int NJ = 16 / NWC;                    // 16-channel MFMA column tiles per wave
    constexpr int WCH = 16 * NJ;                    // channels per wave
    constexpr int HRD = (HROWS2 / 8 + NWAVES - 1) / NWAVES;     // halo DMA instructions per wave and chunk: 6 | 12
    constexpr int WRD = 32 / NWAVES;                // weight DMA instructions per wave and k-step: 4 | 8
    constexpr bool SOLO = NWC == 2;                 // one wave per SIMD
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const sAh = smem;                         // [2][HSTAGE_A2]
    char* const sBh = smem + 2 * HSTAGE_A2;         // [2][HSTAGE_B]
    const int tid = threadIdx.x;
    // wave index as a SCALAR: everything derived from it (DMA destinations, row roles) stays in SGPRs
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wr = wave / NWC, wc = wave % NWC;
    // Tile lists: XCD x owns the tiles [x per_xcd, (x + 1) per_xcd); its workgroup `slot` takes tile slot, slot + wpx, ...
    // (wpx = workgroups per XCD).  Launched with one workgroup per tile (wpx = per_xcd) this is the one-tile mapping;
    // launched PERSISTENT (wpx = CUs / 8: mmh_set_option("lp16_persist")) a workgroup walks several tiles, each with its
    // own prologue: 3-8 % faster from 512 tiles up (no second wave of workgroup launches behind the first, no ragged last
    // round).  Prefetching the next tile's first stages during the last k-steps was built as well and added nothing to that
    // (145 against 143 us at 256 -> 256), while its live state spilled the reflect-fold variant (148 -> 201 us): not kept.
    // (The reflect-fold variant walks tile lists as well since it has one fold accumulator per wave and its halo offsets in
    // LDS: 250 registers, nothing in scratch; mmh_set_option("lp16_persist", 2) = every variant but that one.)
    const int per_xcd = (p.MT * p.NT + 7) / 8;
    const int wpx = (int)(gridDim.x >> 3);
    const int tile_end = min(((int)(blockIdx.x & 7) + 1) * per_xcd, p.MT * p.NT);
    for (int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3); tile < tile_end; tile += wpx) {
    const int mt = tile / p.NT, nt = tile - mt * p.NT;
    const int n0 = nt * TBN;
    const int TX = (p.W + HT - 1) / HT, TY = (p.H + HT - 1) / HT;
    const int b = mt / (TX * TY);
    const int trem = mt - b * (TX * TY);
    const int ty = trem / TX, tx = trem - ty * TX;
    const int oh0 = ty * HT, ow0 = tx * HT;

    // halo DMA roles: instruction q = rd * NWAVES + wave (q < 45) covers LDS rows q*8 .. +7 (row = hy * 20 + hx), lane l
    // the 16-byte chunk l & 7 of row q*8 + l/8.
    // Source offset of LDS row r with the row's swizzle key in the free low bits (the offset is a multiple of 128 bytes):
    // a lane XORs its own chunk (lane & 7) * 16 into it.  0xfffffffe: zero page, 0xffffffff: no such row.
    auto halo_row_off = [&](int r) -> unsigned {
        const int hy = r / HP2, hx = r - hy * HP2;
        int ih = oh0 + hy - 1, iw = ow0 + hx - 1;
        const bool row = r < HROWS2 && hx < HW_;
        if (p.reflect) {
            ih = ih < 0 ? -ih : ih;
            iw = iw < 0 ? -iw : iw;
            ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
            iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
        }
        const bool ok = row && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        return ok ? (unsigned)((b * p.H + ih) * p.W + iw) * (unsigned)p.cs * 2u + (unsigned)(hx & 6) * 16u
                  : (row ? 0xfffffffeu : 0xffffffffu);
    };
    // The plain variants hold their offsets in registers.  The two-waves-per-SIMD reflect-fold variant is six registers short
    // of keeping its fold accumulators out of scratch (a scratch reload in the k-loop is followed by s_waitcnt vmcnt(0): a
    // drain of the whole LDS-DMA ring in every fold k-step), so it parks the 384 row offsets in LDS behind the stages (1.5 of
    // the 6 KiB the stages leave) and fetches them per chunk.
    constexpr bool OFF_LDS = FOLD || SOLO;     // (one wave per SIMD: the table lets a run-time round index pick the offset)
    unsigned a_off[OFF_LDS ? 1 : HRD];
    unsigned* const sOff = reinterpret_cast<unsigned*>(smem + 2 * HSTAGE_A2 + 2 * HSTAGE_B);
    if (OFF_LDS) {
        for (int r = tid; r < HRD * NWAVES * 8; r += NWAVES * 64) sOff[r] = halo_row_off(r);
        __syncthreads();
    } else {
#pragma unroll
        for (int rd = 0; rd < HRD; ++rd) {
            const unsigned ro = halo_row_off((rd * NWAVES + wave) * 8 + (lane >> 3));
            a_off[rd] = ro >= 0xfffffffeu ? ro : ro ^ ((unsigned)(lane & 7) * 16u);
        }
    }
    // weight DMA: wave w, round j moves rows (w * WRD + j) * 8 + lane / 8 of the [256][64] tile.  The swizzle key (r >> 1) & 7
    // = (4 j + lane / 16) & 7 splits into a lane part and bit 0 of j: ONE lane offset, ^ 64 (chunk ^ 4) for odd j; the row
    // advance of j is a scalar added to the base pointer
    unsigned b_off0;
    {
        const int r = wave * (WRD * 8) + (lane >> 3);
        b_off0 = (unsigned)(n0 + r) * (unsigned)p.C * 2u + (unsigned)((lane & 7) ^ ((r >> 1) & 7)) * 16u;
    }
    const int KC = p.C / TBK;
    auto issue_halo1 = [&](int kc, int rd) {        // halo DMA instruction rd of this wave for chunk kc
        char* sA = sAh + (kc & 1) * HSTAGE_A2;
        const char* xb = p.x + (size_t)kc * (TBK * 2);
        unsigned ao;
        if (OFF_LDS) {          // one offset at a time (the asm keeps the reads from being gathered in front)
            ao = sOff[(rd * NWAVES + wave) * 8 + (lane >> 3)];
            asm volatile("" : "+v"(ao) :: "memory");
            ao = ao >= 0xfffffffeu ? ao : ao ^ ((unsigned)(lane & 7) * 16u);
        } else {
            ao = a_off[OFF_LDS ? 0 : rd];
        }
        if (ao != 0xffffffffu) {
            const char* g = ao != 0xfffffffeu ? xb + ao : p.zeros + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds(g, (lds_vp)(sA + (rd * NWAVES + wave) * 1024), 16, 0, 0);
        }
    };
    auto issue_halo = [&](int kc) {
#pragma unroll
        for (int rd = 0; rd < HRD; ++rd) issue_halo1(kc, rd);
    };
    // one wave per SIMD: halo DMA instruction `rd` (a RUN-TIME round: the k-step picks it) without a per-lane branch - rows
    // the pitch-20 layout pads with (hx = 18, 19) load zeros; only the whole-instruction test q < 45 is a (scalar) branch
    auto issue_halo_ao = [&](int kc, int rd, unsigned ao, unsigned ln) {        // ao: the table entry of (rd, this lane's row)
        const int q = rd * NWAVES + wave;
        if (q < HROWS2 / 8) {
            const unsigned ch = (ln & 7u) * 16u;
            const char* g = ao < 0xfffffffeu ? p.x + (size_t)kc * (TBK * 2) + (ao ^ ch) : p.zeros + ch;
            __builtin_amdgcn_global_load_lds(g, (lds_vp)(sAh + (kc & 1) * HSTAGE_A2 + q * 1024), 16, 0, 0);
        }
    };
    auto issue_halo_rt = [&](int kc, int rd) {
        issue_halo_ao(kc, rd, sOff[(rd * NWAVES + wave) * 8 + (lane >> 3)], (unsigned)lane);
    };
    auto issue_w1 = [&](int kc, int t, int j) {     // weight DMA instruction j of this wave: tile of (chunk kc, tap t)
        char* sB = sBh + ((kc + t) & 1) * HSTAGE_B;     // 9 kc + t and kc + t have the same parity
        const char* wbase = p.w + ((size_t)t * p.N * p.C + (size_t)kc * TBK) * 2;
        __builtin_amdgcn_global_load_lds(wbase + (size_t)(j * 8) * p.C * 2 + ((j & 1) ? (b_off0 ^ 64u) : b_off0),
                                         (lds_vp)(sB + (wave * WRD + j) * 1024), 16, 0, 0);
    };
    auto issue_w = [&](int kc, int t) {             // weight tile of (chunk kc, tap t) -> stage (9 kc + t) & 1
#pragma unroll
        for (int j = 0; j < WRD; ++j) issue_w1(kc, t, j);
    };

    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // lane-constant LDS byte addresses (32-bit) of stage 0: A [dw][half], B [half]; the k-step adds the stage offset
    // and dh rows (scalars), the fragment index i an immediate
    const unsigned lds0 = mmh::lds_addr_of(smem);
    // (named scalars, not an array: a select over array elements comes back from the compiler as a run-time indexed
    // load from a SCRATCH copy of the array)
    auto a_lane = [&](int dw, int hf) -> unsigned {
        const unsigned hx = (unsigned)(dw + l15);
        return lds0 + (unsigned)(wr * 8 * HP2) * ROWB + hx * ROWB + ((((unsigned)(4 * hf + g4)) ^ (hx & 6u)) << 4);
    };
    const unsigned aA00 = a_lane(0, 0), aA10 = a_lane(1, 0), aA20 = a_lane(2, 0);
    unsigned aB[2];
    {
        const unsigned bkey = (unsigned)(l15 >> 1);     // weight rows wc*WCH + j*16 + l15: key (row >> 1) & 7 = l15 >> 1
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            aB[hf] = lds0 + 2 * HSTAGE_A2 + (unsigned)(wc * WCH + l15) * ROWB + ((((unsigned)(4 * hf + g4)) ^ bkey) << 4);
    }
    // The half-0 address of tap (dh, dw) in halo stage st: aA00 + [dw == 1] * d1 + [dw == 2] * d2 + st * HSTAGE_A2 +
    // dh * 2560 - two multiply-adds by 0 / 1 scalars instead of a three-way select (which the compiler turns into
    // scalar branches at the head of every k-step); the half-1 address of the same tap is that ^ 64 (every offset
    // added is a multiple of the 128-byte row).  Taps advance by counters, not by t / 3 and t % 3.
    const unsigned d1 = aA10 - aA00, d2 = aA20 - aA00;
    auto a_half0 = [&](int dh, int dw, int st) -> unsigned {
        const unsigned m1 = dw == 1 ? 1u : 0u, m2 = dw == 2 ? 1u : 0u;
        return aA00 + m1 * d1 + m2 * d2 + (unsigned)(st * HSTAGE_A2 + dh * (HP2 * ROWB));
    };
    const int nk = 9 * KC;

    // FOLD (dgrad of a ReflectionPad2d(1) conv; H, W multiples of 16, at least two tiles each way): the gradient of the pad
    // ring, folded back onto rows 1 / H-2 and columns 1 / W-2, is computed HERE with the weight fragments the k-step holds
    // anyway - instead of eight border GEMMs + an add kernel per conv (mmh_conv2d_dgrad_border: 46-70 us, 90 times per step).
    //   ring row -1 -> row 1 (top tiles, taps kh = 0): one extra A fragment (halo row of dy row 0), accumulated by the
    //       waves that own tile rows 0-7 and added to acc[1] after the loop; ring row H -> row H-2 (bottom tiles, kh = 2)
    //       likewise by the waves of rows 8-15 into acc[6];
    //   ring column -1 -> column 1 (left tiles, taps kw = 0): the ring values of the tile's 16 ROWS form one MFMA
    //       column block (lane <-> tile row, its fragment read down the halo column of dy column 0); ring column W ->
    //       column W-2 (right tiles, kw = 2) likewise; after the loop the block goes through LDS to the lanes that hold
    //       pixel column 1 / 14;
    //   the four ring corners (one tap each) are single-lane fragments of the row term.
    // A tile has at most one row term and one column term (two tiles each way), and a workgroup two wave rows: ONE fold
    // accumulator FA per wave - the row term on its own wave row, the column term on the other one (on wr = 0 / 1 for left /
    // right when the tile has no row term).  One accumulator, one fragment, one multiply site per half k-step: the earlier
    // build (row term into acc[1] | acc[6] in place, a second accumulator for the column term) made the register allocator
    // copy acc[6] through temporaries in every fold k-step and spill part of the column accumulator - whose reload was
    // followed by s_waitcnt vmcnt(0), a drain of the LDS-DMA ring - and left no room for the tile loop (534 -> 482 us at
    // 512 -> 512, plain dgrad 441: tools/bench_lp16_fold.py).  Cost: NJ MFMAs on 8 NJ in a third of the k-steps of edge tiles.
    const bool fold_on = FOLD && !(p.dbg & 4);
    const bool t_top = fold_on && !(p.dbg & 16) && ty == 0, t_bot = fold_on && !(p.dbg & 16) && ty == TY - 1;
    const bool f_left = fold_on && !(p.dbg & 8) && tx == 0, f_right = fold_on && !(p.dbg & 8) && tx == TX - 1;
    const bool my_row = (t_top && wr == 0) || (t_bot && wr == 1);
    const int col_wr = t_top ? 1 : (t_bot ? 0 : (f_left ? 0 : 1));
    const bool my_col = (f_left || f_right) && wr == col_wr;
    const int fold_kh = t_top ? 0 : 2, fold_kw = f_left ? 0 : 2;    // the taps of this wave's term
    unsigned fold_taps = 0, cnr_taps = 0;       // bit t = 3 kh + kw: this wave multiplies a fold term / the corner term at tap t
    if (FOLD) {
        for (int tt = 0; tt < 9; ++tt) {
            const int kh_ = tt / 3, kw_ = tt - 3 * kh_;
            const bool row = my_row && kh_ == fold_kh;
            if (row || (my_col && kw_ == fold_kw)) fold_taps |= 1u << tt;
            if (row && ((kw_ == 0 && f_left) || (kw_ == 2 && f_right))) cnr_taps |= 1u << tt;
        }
    }
    f32x4 FA[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) FA[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // column term: lane <-> tile row l15: halo row (l15 + dh) * 20 + hx, hx = 1 (left) or 16 (right): both have key hx & 6 == 0
    const unsigned baseT = lds0 + (unsigned)l15 * (HP2 * ROWB) + (unsigned)(f_left ? 1 : 16) * ROWB + ((unsigned)g4 << 4);
    // row term: the tap's halo row 1 (top) / 16 (bottom) at this lane's column: a_cur without its dh rows and wave rows
    const unsigned f_rsel = (unsigned)(((t_top ? 1 : 16) - wr * 8) * (HP2 * ROWB));

    bf16x8 af[8], b0[NJ], b1[NJ];
    issue_halo(0);
    issue_w(0, 0);
    issue_w(0, 1);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    int kc = 0, t = 0, kh = 0, kw = 0;                  // (chunk, tap = 3 kh + kw) of k-step ks
    unsigned a_cur = a_half0(SIGN > 0 ? 0 : 2, SIGN > 0 ? 0 : 2, 0);
    // One wave per SIMD: the 16 fragment reads of a half k-step go out in ONE order everywhere - group g: activation row g,
    // and behind the groups 0-3 two weight fragments each - so that every group of the next half finds its operands among
    // the twelve oldest reads (all eight weight fragments, rows 0-3) or behind one more counted wait (rows 4-7).
    auto solo_reads = [&](auto I, unsigned a_addr, unsigned b_addr, bf16x8* bdst) {
        constexpr int i = decltype(I)::value;
        if (p.dbg & 64) return;         // timing only: the loop without its fragment reads
        lds_frag_asm<i * (HP2 * ROWB)>(af[i], a_addr);
        if (i < NJ / 2) {
            lds_frag_asm<(2 * i) * (16 * ROWB)>(bdst[2 * i], b_addr);
            lds_frag_asm<(2 * i + 1) * (16 * ROWB)>(bdst[2 * i + 1], b_addr);
        }
    };
    auto unroll8 = [&](auto&& f) {
        f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{});
        f(std::integral_constant<int, 3>{}); f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 5>{});
        f(std::integral_constant<int, 6>{}); f(std::integral_constant<int, 7>{});
    };
    if (SOLO) {
        unroll8([&](auto I) { solo_reads(I, a_cur, aB[0], b0); });
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) b0[j] = lds_frag(aB[0] + j * (16 * ROWB));
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = lds_frag(a_cur + i * (HP2 * ROWB));
    }
    for (int ks = 0; ks < nk; ++ks) {
        const unsigned sb = (unsigned)(ks & 1) * HSTAGE_B;
        const unsigned a1 = a_cur ^ 64u;                // second half of this tap
        const unsigned bb1 = aB[1] + sb;
        int kc2 = kc, t2 = t + 1, kh2 = kh, kw2 = kw + 1;
        if (kw2 == 3) { kw2 = 0; ++kh2; }
        if (t2 == 9) { t2 = 0; kh2 = 0; ++kc2; }
        const unsigned a0n = a_half0(SIGN > 0 ? kh2 : 2 - kh2, SIGN > 0 ? kw2 : 2 - kw2, kc2 & 1);
        const unsigned bb0n = aB[0] + (HSTAGE_B - sb);
        // fold role of this k-step (wave-uniform): this wave's row term (its kh) or column term (its kw); corner = a row
        // k-step of a left / right tile with kw = 0 / 2.  The fragment is fetched BEFORE the MFMA block of each half so
        // that the LDS latency hides behind it: the tap's halo row 1 / 16 at this lane's column, or the halo column 1 / 16
        // at this lane's row; lane 1 / 14 of the dw = 0 / 2 variant for the corner.
        // (one bit test per k-step outside the fold taps: fold_taps / cnr_taps are this wave's nine-bit tap masks of the tile)
        const bool do_fold = FOLD && ((fold_taps >> t) & 1u);
        const bool do_cnr = FOLD && ((cnr_taps >> t) & 1u);
        const unsigned f_st = (unsigned)((kc & 1) * HSTAGE_A2);
        unsigned f_addr = 0;
        bf16x8 axf;
        if (FOLD && do_fold) {
            const unsigned f_dhb = (unsigned)((2 - kh) * (HP2 * ROWB));
            f_addr = my_row ? a_cur - f_dhb + f_rsel : baseT + f_st + f_dhb;
            axf = lds_frag(f_addr);
        }
        // ---- half 0: multiply (ks, 0) while the fragments of (ks, 1) stream in
        if (!SOLO) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) b1[j] = lds_frag(bb1 + j * (16 * ROWB));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16s<H16>(b0[j], af[i], acc[i][j]);
                af[i] = lds_frag(a1 + i * (HP2 * ROWB));
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, NJ, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        } else {
            lgk_wait<4>();          // all but the four youngest reads (activation rows 4-7 of the last half)
            unroll8([&](auto I) {
                constexpr int i = decltype(I)::value;
                if (i == 4) lgk_wait<12>();     // rows 4-7 of the last half (the twelve reads of groups 0-3 may fly)
#pragma unroll
                for (int j = 0; j < NJ; ++j) mfma16s_agpr<H16>(acc[i][j], b0[j], af[i]);
                solo_reads(I, a1, bb1, b1);
            });
        }
        if (FOLD && do_fold) {    // the fold term of this half (b0 = this tap's weights); fragment fetched above
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (SOLO) mfma16s_vgpr<H16>(FA[j], b0[j], axf);
                else FA[j] = mfma16s<H16>(b0[j], axf, FA[j]);
            }
            if (do_cnr) {         // once per chunk in the four corner tiles: not worth registers for a prefetch
                axf = lds_frag((kw == 0 ? aA00 : aA20) + f_st + f_rsel);
                if (l15 != (kw == 0 ? 1 : 14) || (p.dbg & 2048)) axf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (SOLO) asm volatile("s_nop 3" : "+v"(axf));      // a VALU write two wait states in front of an MFMA read
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (SOLO) mfma16s_vgpr<H16>(FA[j], b0[j], axf);
                    else FA[j] = mfma16s<H16>(b0[j], axf, FA[j]);
                }
            }
        }
        // ---- middle of the k-step: the weight stage ks is read; stage ks+1 (issued one k-step ago) must have
        // landed.  The halo of the next chunk, issued right after the weights at t == 0, may stay in flight across
        // the barrier of t == 1 (it is needed nine k-steps after its issue): the wait then leaves the newest
        // HRD - 1 loads outstanding (every wave issues HRD - 1 or HRD of them)
        if (SOLO) {
            // vmcnt(0) lgkmcnt(4): every DMA of the last half has landed; every LDS read but the four youngest (activation
            // rows 4-7, from the halo) is complete - the weight stage read in this half is the one the next DMA overwrites.
            // A bare s_barrier: __syncthreads() brings a fence, i.e. a full lgkmcnt(0) drain
            if (p.dbg & 512) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(4)" ::: "memory");       // timing only: no barrier
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(4)\n\ts_barrier" ::: "memory");
        } else {
            // (lgkmcnt(8) and a bare s_barrier here - only the four weight-fragment reads of half 1, the oldest of the twelve,
            // read the stage the next DMA overwrites - measured no different from the full drain: 458 against 456 us at
            // 512 -> 512, profiles/r05_lp16_barrier_ab.txt; the plain form stays)
            if (t == 1 && kc + 1 < KC) __builtin_amdgcn_s_waitcnt(0x0070 | (HRD - 1));
            else __builtin_amdgcn_s_waitcnt(0x0070);
            __syncthreads();
        }
        // The DMA of the next stages: a DMA instruction costs its wave 60-180 cycles of issue time.  Two waves per SIMD
        // (wr = 0 / 1) issue theirs at different points of half 1 - one wave's issue runs under the other's multiplies
        // (wino_wgrad_dma.hip: 1045 -> 906 us from the same change; mmh_set_option("lp16_dbg") bit 32 = everybody here).
        // One wave per SIMD: one instruction behind every MFMA group of half 1 (weights first, then the halo).
        const bool w_next = ks + 2 < nk && !(p.dbg & 1);     // dbg: timing-only ablations (mmh_set_option "lp16_dbg"; results wrong)
        const bool h_next = t == 0 && kc + 1 < KC && !(p.dbg & 2);
        int kc3 = kc, t3 = t + 2;
        if (t3 >= 9) { t3 -= 9; ++kc3; }
        auto issue_next = [&]() {
            if (w_next) issue_w(kc3, t3);
            if (h_next) issue_halo(kc + 1);
        };
        const bool early = wr == 0 || (p.dbg & 32);
        if (!SOLO && early) issue_next();
        if (SOLO && (p.dbg & 32)) {
            if (w_next) issue_w(kc3, t3);
            if (t < HRD / 2 && kc + 1 < KC && !(p.dbg & 2)) { issue_halo_rt(kc + 1, 2 * t); issue_halo_rt(kc + 1, 2 * t + 1); }
        }
        if (FOLD && do_fold) axf = lds_frag(f_addr ^ 64u);      // the fold fragment of half 1
        // ---- half 1: multiply (ks, 1) while the fragments of (ks+1, 0) stream in
        // (after the last k-step these reads fetch fragments nobody uses, from addresses inside the stages: cheaper
        // than a branch around each of them)
        if (!SOLO) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) b0[j] = lds_frag(bb0n + j * (16 * ROWB));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16s<H16>(b1[j], af[i], acc[i][j]);
                af[i] = lds_frag(a0n + i * (HP2 * ROWB));
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, NJ, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (!early) issue_next();
#pragma unroll
            for (int i = 4; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16s<H16>(b1[j], af[i], acc[i][j]);
                af[i] = lds_frag(a0n + i * (HP2 * ROWB));
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, NJ, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        } else {
            // one wave per SIMD: per MFMA group one activation refill, one weight fragment of the next k-step, one DMA
            // instruction of the weight stage two k-steps ahead; the halo of the next chunk two instructions per k-step
            // in the taps 0..5, behind the groups 6 and 7.  (Wave-uniform scalar branches: the volatile MFMAs pin the order.)
            const int kcn = kc + 1;
            const bool h_now = t < HRD / 2 && kcn < KC && !(p.dbg & 2);
            // the two halo source offsets of this k-step from the table, requested here (consumed behind groups 6 and 7: by
            // then they are the oldest LDS operations in flight); lane id from mbcnt - a lane-constant address kept across
            // the loop went to scratch, and its reload (s_waitcnt vmcnt(0)) drained the DMA in flight
            const int rd0 = min(2 * t, HRD - 2);
            const unsigned ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            const unsigned tab = lds0 + 2 * HSTAGE_A2 + 2 * HSTAGE_B + (unsigned)(((rd0 * NWAVES + wave) * 8) * 4) + (ln >> 3) * 4u;
            unsigned ao0, ao1;
            lgk_wait<4>();
            lds_u32_asm(ao0, tab);
            lds_u32_asm(ao1, tab + NWAVES * 8 * 4);
            unroll8([&](auto I) {
                constexpr int i = decltype(I)::value;
                if (i == 4) lgk_wait<12>();     // rows 4-7 of the last half and the two table entries
#pragma unroll
                for (int j = 0; j < NJ; ++j) mfma16s_agpr<H16>(acc[i][j], b1[j], af[i]);
                solo_reads(I, a0n, bb0n, b0);
                if (!(p.dbg & 32)) {
                    if (w_next && i < WRD) issue_w1(kc3, t3, i);
                    if (h_now && i >= 6) issue_halo_ao(kcn, rd0 + (i - 6), i == 6 ? ao0 : ao1, ln);
                }
            });
        }
        if (FOLD && do_fold) {    // the fold term of this half (b1 = this tap's weights); fragment fetched above
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (SOLO) mfma16s_vgpr<H16>(FA[j], b1[j], axf);
                else FA[j] = mfma16s<H16>(b1[j], axf, FA[j]);
            }
            if (do_cnr) {
                axf = lds_frag(((kw == 0 ? aA00 : aA20) + f_st + f_rsel) ^ 64u);
                if (l15 != (kw == 0 ? 1 : 14) || (p.dbg & 2048)) axf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (SOLO) asm volatile("s_nop 3" : "+v"(axf));
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (SOLO) mfma16s_vgpr<H16>(FA[j], b1[j], axf);
                    else FA[j] = mfma16s<H16>(b1[j], axf, FA[j]);
                }
            }
        }
        kc = kc2; t = t2; kh = kh2; kw = kw2; a_cur = a0n;
    }

    if (SOLO) {
        lgk_wait<0>();      // the fragment reads behind the last k-step (nobody uses them) still write registers
        if (p.dbg & 256) {  // timing only: no epilogue (one store keeps the accumulators alive)
            f32x4 sum = acc[0][0];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) asm volatile("" :: "a"(acc[i][j]));
            if (sum[0] == 12345.678f) p.y16[0] = 1;
            __syncthreads();
            continue;
        }
        // the compiler does not know the asm statements above were MFMAs: an accumulator needs up to 18 wait states between
        // the MFMA that writes it and a v_accvgpr_read (no hardware interlock); FA likewise
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" : "+a"(acc[i][j]));
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    }
    if (FOLD && my_row) {               // the row term: same layout as the accumulators of tile row 1 / 14
        if (t_top) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[1][j] += FA[j];
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[6][j] += FA[j];
        }
    }
    if (FOLD && (f_left || f_right)) {
        // the column term: FA [channel 4 g4 + r of block j][tile row l15] -> LDS X[256 channels][16 rows] -> the lanes
        // that hold pixel column 1 (left) / 14 (right) of each tile row
        float* X = reinterpret_cast<float*>(smem);
        __syncthreads();                    // every wave is done reading the last stages
        if (my_col) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[(wc * WCH + j * 16 + 4 * g4 + r) * 16 + l15] = FA[j][r];
        }
        __syncthreads();
        if (l15 == (f_left ? 1 : 14)) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += X[(wc * WCH + j * 16 + 4 * g4 + r) * 16 + wr * 8 + i];
        }
    }

    float bv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = p.bias ? p.bias[n0 + wc * WCH + j * 16 + 4 * g4 + r] : 0.f;
    const int ow = ow0 + l15;
    // The activation as a compile-time constant per branch (a run-time test per element put the tanh expansion behind every
    // one of the 128 values of a lane: 21 000 instructions of epilogue); the dgrad variants have none.
    auto with_act = [&](auto&& body) {
        if (SIGN > 0 && p.act == MMH_ACT_RELU) body(std::integral_constant<int, MMH_ACT_RELU>{});
        else if (SIGN > 0 && p.act == MMH_ACT_TANH) body(std::integral_constant<int, MMH_ACT_TANH>{});
        else body(std::integral_constant<int, MMH_ACT_NONE>{});
    };
    if (p.y16 && !(p.dbg & 128)) {
        // 16-bit output: 16-byte stores after the lane-pair trade (common.h: pair_swap8) - 16 store instructions per tile
        // instead of 32; 256 -> 256 fprop 150 -> 134 us, the 16-bit step 102.5 -> 100.8 ms (tools/ab_lp16_stores.py;
        // mmh_set_option("lp16_dbg", 128) = 8-byte stores)
        const bool odd = (g4 & 1) != 0;
        const int cb0 = (odd ? 16 : 0) + 4 * (g4 & 2);      // this lane's 8 channels within a 32-channel tile pair
        if (NBR) {
            // dx is the gradient g of a norm's output: the norm's backward sums of the values AS STORED, from here (full
            // tiles: host side).  Per lane 8 rows x 8 channels x NJ / 2 blocks: s1 += dz, sx += dz * x with
            // dz = keep ? g * dsc : 0; the 16 pixel lanes are summed by four swizzles, lane 0 of each channel group turns
            // sx into s2 = invstd * (sx - mean * s1) and writes the partial of this (image, half tile).  The norm's input
            // comes back from LDS (nbx), its keep bits from kq: both requested above.
            // This wave's slice of the norm's input (8 rows x 16 pixels x WCH channels) comes by LDS-DMA into the stage
            // buffers the k-loop is done with - every lane's 16 bytes land where the lane reads them back, 16 KiB per wave
            // behind the column fold's 16 KiB - and its keep bits into registers: all of it in flight at once, under the
            // stores of dx.  (Through registers one tile row ahead the epilogue waited a memory round trip per row, 10 us
            // per tile; four rows ahead the ring spilled.  The DMA is inline asm: issued through the builtin, the compiler
            // waits for it in front of the next LDS read of any kind.)
            // The keep bits of a pixel's WCH channels are 16 contiguous bytes: two more DMA instructions bring the wave's
            // 8 x 16 of them (lane (l15, g4) of instruction q: row 4 q + g4), read back as 16-bit words - in registers the
            // 16 words were what spilled.
            static_assert(!NBR || NJ == 4, "the keep-bit DMA assumes 64 channels per wave");
            char* const nbx = smem + wave * 18432;
            {
                const size_t e00 = (((size_t)b * p.H + (oh0 + wr * 8)) * p.W + ow) * p.N + (size_t)(n0 + wc * WCH + cb0);
                const size_t erow = (size_t)p.W * p.N;
                const unsigned nb0 = __builtin_amdgcn_readfirstlane(mmh::lds_addr_of(nbx));
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done reading LDS
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int jp = 0; jp < NJ / 2; ++jp)
                        mmh::lds_dma16(p.nbr_x + (e00 + i * erow + jp * 32) * 2, nb0 + (unsigned)(i * (NJ / 2) + jp) * 1024u);
                if (p.nbr_bits) {
                    const size_t k00 = (((size_t)b * p.H + (oh0 + wr * 8 + g4)) * p.W + ow) * p.N + (size_t)(n0 + wc * WCH);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        mmh::lds_dma16(p.nbr_bits + ((k00 + (size_t)(4 * q) * erow) >> 3), nb0 + 16384u + (unsigned)q * 1024u);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {               // the stores first: they do not depend on what is in flight
                const size_t m = ((size_t)b * p.H + oh0 + wr * 8 + i) * p.W + ow;
#pragma unroll
                for (int jp = 0; jp < NJ / 2; ++jp) {
                    float v[8];
                    mmh::pair_swap8(acc[i][2 * jp], acc[i][2 * jp + 1], v);
                    mmh::store8_lp16<H16>(p.y16 + (m * p.y_cs + (n0 + wc * WCH + jp * 32 + cb0)) * 2, v);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the DMA above has landed (this wave reads only its own)
            // one 32-channel block at a time, one tile row after the other (sched_barrier: unrolled freely the scheduler
            // gathers all the LDS reads in front and spills - and a spill in this epilogue is a memory round trip nothing hides)
#define MMH_SWZ(val, s) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, val), 0x1f | ((s) << 10)))
#pragma unroll
            for (int jp = 0; jp < NJ / 2; ++jp) {
                float ns1[8], nsx[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) ns1[e] = nsx[e] = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v[8];
                    mmh::pair_swap8(acc[i][2 * jp], acc[i][2 * jp + 1], v);
                    const uint4 xr = *reinterpret_cast<const uint4*>(nbx + (i * (NJ / 2) + jp) * 1024 + lane * 16);
                    const unsigned xw[4] = {xr.x, xr.y, xr.z, xr.w};
                    const unsigned kb = p.nbr_bits ? *reinterpret_cast<const unsigned short*>(
                                                         nbx + 16384 + (i >> 2) * 1024 + ((i & 3) * 16 + l15) * 16 + (jp * 4 + (cb0 >> 3)) * 2)
                                                   : 0xffffu;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned short xs = (unsigned short)((e & 1) ? (xw[e >> 1] >> 16) : (xw[e >> 1] & 0xffffu));
                        const float xf = H16 ? (float)__builtin_bit_cast(_Float16, xs) : __builtin_bit_cast(float, (unsigned)xs << 16);
                        const float gr = H16 ? (float)(_Float16)v[e] : (float)(__bf16)v[e];
                        const float gg = ((kb >> (e < 4 ? e : e + 4)) & 1u) ? gr * p.nbr_dsc : 0.f;
                        ns1[e] += gg;
                        nsx[e] = fmaf(gg, xf, nsx[e]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float a1 = ns1[e], ax = nsx[e];
                    a1 += MMH_SWZ(a1, 8); ax += MMH_SWZ(ax, 8);
                    a1 += MMH_SWZ(a1, 4); ax += MMH_SWZ(ax, 4);
                    a1 += MMH_SWZ(a1, 2); ax += MMH_SWZ(ax, 2);
                    a1 += MMH_SWZ(a1, 1); ax += MMH_SWZ(ax, 1);
                    ns1[e] = a1; nsx[e] = ax;
                }
                if (l15 == 0) {
                    const int chunks = TX * TY * 2;
                    const int grp = p.nbr_groups == 1 ? 0 : b;
                    const int c0 = n0 + wc * WCH + jp * 32 + cb0;
                    float* pp = p.nbr_part + ((size_t)(b * chunks + (ty * TX + tx) * 2 + wr) * 2) * p.N + c0;
                    float o2[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float mu = p.nbr_mean[(size_t)grp * p.N + c0 + e], is = p.nbr_invstd[(size_t)grp * p.N + c0 + e];
                        o2[e] = is * (nsx[e] - mu * ns1[e]);
                    }
                    *reinterpret_cast<f32x4*>(pp) = (f32x4){ns1[0], ns1[1], ns1[2], ns1[3]};
                    *reinterpret_cast<f32x4*>(pp + 4) = (f32x4){ns1[4], ns1[5], ns1[6], ns1[7]};
                    *reinterpret_cast<f32x4*>(pp + p.N) = (f32x4){o2[0], o2[1], o2[2], o2[3]};
                    *reinterpret_cast<f32x4*>(pp + p.N + 4) = (f32x4){o2[4], o2[5], o2[6], o2[7]};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef MMH_SWZ
        } else
        with_act([&](auto A) {
            constexpr int ACT = decltype(A)::value;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int oh = oh0 + wr * 8 + i;
                const size_t m = ((size_t)b * p.H + oh) * p.W + ow;
#pragma unroll
                for (int jp = 0; jp < NJ / 2; ++jp) {
                    float v[8];
                    mmh::pair_swap8(acc[i][2 * jp], acc[i][2 * jp + 1], v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float t = v[e] + (p.bias ? p.bias[n0 + wc * WCH + jp * 32 + cb0 + e] : 0.f);
                        v[e] = ACT == MMH_ACT_RELU ? (t > 0.f ? t : 0.f) : (ACT == MMH_ACT_TANH ? tanhf(t) : t);
                    }
                    if (oh < p.H && ow < p.W) {
                        mmh::store8_lp16<H16>(p.y16 + (m * p.y_cs + (n0 + wc * WCH + jp * 32 + cb0)) * 2, v);
                    }
                }
            }
        });
    } else if (SIGN < 0 && p.addend) {
        // dx = dgrad + addend, fp32 (mmh_conv3x3_lp16_dgrad_add: no bias, no activation).  The addend of tile row i + 1 is
        // requested while row i is added and stored: written as "load, add, store" per accumulator the compiler put an
        // s_waitcnt vmcnt(0) behind every load - 32 full memory round trips per tile, each also waiting for the store in
        // front of it (256 -> 256: 182 us against 142 for the same dgrad without addend).
        const size_t o00 = (((size_t)b * p.H + (oh0 + wr * 8)) * p.W + ow) * p.y_cs + (size_t)(n0 + wc * WCH + 4 * g4);
        const size_t rstep = (size_t)p.W * p.y_cs;             // one tile row down
        if (FOLD || (oh0 + HT <= p.H && ow0 + HT <= p.W)) {     // a full tile (always, with the fold): rows in flight
            constexpr int NB = SOLO ? 2 : 3;
            f32x4 ad[NB][NJ];
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) ad[i][j] = *reinterpret_cast<const f32x4*>(p.addend + o00 + i * rstep + j * 16);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] += ad[i % NB][j];
                    *reinterpret_cast<f32x4*>(p.y + o00 + i * rstep + j * 16) = acc[i][j];
                }
                if (i + NB < 8) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        ad[i % NB][j] = *reinterpret_cast<const f32x4*>(p.addend + o00 + (i + NB) * rstep + j * 16);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (oh0 + wr * 8 + i >= p.H || ow >= p.W) continue;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] += *reinterpret_cast<const f32x4*>(p.addend + o00 + i * rstep + j * 16);
                    *reinterpret_cast<f32x4*>(p.y + o00 + i * rstep + j * 16) = acc[i][j];
                }
            }
        }
    } else {
        with_act([&](auto A) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int oh = oh0 + wr * 8 + i;
                if (oh >= p.H || ow >= p.W) continue;
                const size_t m = ((size_t)b * p.H + oh) * p.W + ow;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const size_t elem = m * p.y_cs + (n0 + wc * WCH + j * 16 + 4 * g4);
                    store4<H16>(p.y, p.y16, elem, acc[i][j], bv[j], decltype(A)::value);
                }
            }
        });
    }
    if (SIGN > 0 && !FOLD && p.stats) {
        // The InstanceNorm behind this conv (models/Generator.py:66-77) wants mean and M2 per (image, channel): each wave
        // owns 8 rows x 16 pixels of WCH channels - count / mean / M2 of the values AS STORED (rounded to 16 bits) per
        // half tile (wave row) and channel, merged later (Chan) by mmh_norm_stats_merge[_finalize]: y is not read again
        // for statistics.  Lane: 8 values per channel (two passes in registers), then four equal-count Chan merges across
        // the 16 pixel lanes, 64 channels (four column tiles) at a time.  Host side guarantees H, W multiples of 16 (no
        // ragged tiles), no activation.
        const int chunks = TX * TY * 2;
#pragma unroll
        for (int jb = 0; jb < NJ; jb += 4) {
        float* sp = p.stats + ((size_t)(b * chunks + (ty * TX + tx) * 2 + wr) * 3) * p.N + n0 + wc * WCH + jb * 16 + 4 * g4;
        float mu[16], m2[16];               // channel c = 4 j + r of this lane's 16: mean and M2 of its 8 rows
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v[8], mean = 0.f, q = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float t = acc[i][jb + j][r] + bv[jb + j][r];
                    v[i] = H16 ? (float)(_Float16)t : (float)(__bf16)t;
                    mean += v[i];
                }
                mean *= 0.125f;
#pragma unroll
                for (int i = 0; i < 8; ++i) q = fmaf(v[i] - mean, v[i] - mean, q);
                mu[4 * j + r] = mean; m2[4 * j + r] = q;
            }
        // Four equal-count Chan merges across the 16 pixel lanes as a reduce-scatter: at the step with lane distance s the
        // lane keeps the half of its channels whose index bit matches its own lane bit and hands the other half to its
        // partner (ds_swizzle bit mode: and 0x1f, or 0, xor s) - 8 + 4 + 2 + 1 merges instead of 4 x 16, and lane l15 ends
        // up with channel l15 of the 16.
#define MMH_SWZ(val, s) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, val), 0x1f | ((s) << 10)))
#define MMH_RS_STEP(NOUT, S, BIT, W, MI, QI, MO, QO)                                            \
    _Pragma("unroll") for (int c = 0; c < NOUT; ++c) {                                         \
        const float km = BIT ? MI[NOUT + c] : MI[c], sm = BIT ? MI[c] : MI[NOUT + c];           \
        const float kq = BIT ? QI[NOUT + c] : QI[c], sq = BIT ? QI[c] : QI[NOUT + c];           \
        const float om = MMH_SWZ(sm, S), oq = MMH_SWZ(sq, S), dl = om - km;                     \
        QO[c] = kq + oq + dl * dl * W; MO[c] = 0.5f * (km + om);                                \
    }
        const bool b3 = (l15 & 8) != 0, b2 = (l15 & 4) != 0, b1_ = (l15 & 2) != 0, b0_ = (l15 & 1) != 0;
        float ma[8], qa[8], mb[4], qb[4], mc[2], qc[2], md[1], qd[1];
        MMH_RS_STEP(8, 8, b3, 4.f, mu, m2, ma, qa)
        MMH_RS_STEP(4, 4, b2, 8.f, ma, qa, mb, qb)
        MMH_RS_STEP(2, 2, b1_, 16.f, mb, qb, mc, qc)
        MMH_RS_STEP(1, 1, b0_, 32.f, mc, qc, md, qd)
#undef MMH_RS_STEP
#undef MMH_SWZ
        const int co = (l15 >> 2) * 16 + (l15 & 3);     // channel 4 j + r = l15 of the lane's 16 -> j * 16 + r of the 64
        sp[co] = 128.f;
        sp[p.N + co] = md[0];
        sp[2 * p.N + co] = qd[0];
        }
    }
    __syncthreads();        // the next tile's prologue refills the stages: every wave must be done with this tile's
    }   // tiles of this workgroup
}

#ifdef MMH_AB_KERNELS
// Diagnostic (A/B) builds only - the shipped kernel executes no stamp: with mmh_set_option("lp16_dbg", 4096) wave 0 of every
// workgroup brackets the WHOLE kernel body with the shader-cycle counter (s_memtime) and the constant 100 MHz counter
// (s_memrealtime); delta s_memtime / delta s_memrealtime x 100 MHz is the clock the chip holds under this kernel's load
// (MI355X_MICROARCH.md 'DVFS give-back' item 6).  The values go to a buffer of their own (mmh_lp16_clock_stamps reads it)
// and no output is computed from them.
constexpr int CLOCK_STAMP_WGS = 2048;
__device__ unsigned long long g_lp16_clock_stamps[2 * CLOCK_STAMP_WGS];
#endif

template <bool H16, int SIGN, bool FOLD>
__global__ void __launch_bounds__(512, 2) conv_lp16h2_kernel(const LpConvKP p) {
#ifdef MMH_AB_KERNELS
    if (p.dbg & 4096) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0) alone: the stamps are back before the body's counted LDS waits
        conv_lp16h2_body<H16, SIGN, FOLD, 4>(p);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0 && blockIdx.x < CLOCK_STAMP_WGS) {
            g_lp16_clock_stamps[2 * blockIdx.x] = t1 - t0;
            g_lp16_clock_stamps[2 * blockIdx.x + 1] = r1 - r0;
        }
        return;
    }
#endif
    conv_lp16h2_body<H16, SIGN, FOLD, 4>(p);
}

// the 16-bit dgrad whose epilogue also takes the norm-backward sums of what it stores (LpConvKP::nbr_*): its own
// instantiations, so that the register budget of the plain variants is untouched
template <bool H16, bool FOLD>
__global__ void __launch_bounds__(512, 2) conv_lp16h2_nbr_kernel(const LpConvKP p) {
    conv_lp16h2_body<H16, -1, FOLD, 4, true>(p);
}

// One wave per SIMD, 512 registers per lane (256 of them accumulators): the NWC = 2 form of the body above.  It LOSES to the
// two-waves-per-SIMD form by 30 % (profiles/r05_lp16q_ablation.txt) and is compiled into A/B builds
// only (make AB=1: -DMMH_AB_KERNELS; mmh_set_option("lp16_shape", 20), tools/bench_lp16q.py).
#ifdef MMH_AB_KERNELS
template <bool H16, int SIGN, bool FOLD>
__global__ void __launch_bounds__(256, 1) conv_lp16q_kernel(const LpConvKP p) {
    conv_lp16h2_body<H16, SIGN, FOLD, 2>(p);
}
#endif

}  // namespace

namespace mmh { namespace lp16 {

int clock_stamps(unsigned long long* host_pairs, int max_workgroups) {
#ifdef MMH_AB_KERNELS
    const int n = max_workgroups < CLOCK_STAMP_WGS ? max_workgroups : CLOCK_STAMP_WGS;
    if (n <= 0 || !host_pairs) return 0;
    if (hipMemcpyFromSymbol(host_pairs, HIP_SYMBOL(g_lp16_clock_stamps), (size_t)n * 2 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    return n;
#else
    (void)host_pairs; (void)max_workgroups;
    return 0;
#endif
}

bool conv_lp16_halo_has_solo() {
#ifdef MMH_AB_KERNELS
    return true;
#else
    return false;
#endif
}

int launch_conv_lp16_halo(const LpConvKP& p, const mmh_conv_desc* d, int mode, bool solo, hipStream_t st) {
    constexpr int lds2 = 2 * HSTAGE_A2 + 2 * HSTAGE_B + HROUNDS2 * 64 * 4;    // + the fold variant's row-offset table
    MMH_REQUIRE(!solo || conv_lp16_halo_has_solo(), "lp16_shape 20 (one wave per SIMD) is compiled into A/B builds only: make AB=1");
    static int ready = -1;
    if (ready != 0) {
        hipError_t e = hipSuccess;
        const void* fs[] = {reinterpret_cast<const void*>(conv_lp16h2_kernel<false, 1, false>),
                            reinterpret_cast<const void*>(conv_lp16h2_kernel<false, -1, false>),
                            reinterpret_cast<const void*>(conv_lp16h2_kernel<true, 1, false>),
                            reinterpret_cast<const void*>(conv_lp16h2_kernel<true, -1, false>),
                            reinterpret_cast<const void*>(conv_lp16h2_kernel<false, -1, true>),
                            reinterpret_cast<const void*>(conv_lp16h2_kernel<true, -1, true>),
                            reinterpret_cast<const void*>(conv_lp16h2_nbr_kernel<false, false>),
                            reinterpret_cast<const void*>(conv_lp16h2_nbr_kernel<true, false>),
                            reinterpret_cast<const void*>(conv_lp16h2_nbr_kernel<false, true>),
                            reinterpret_cast<const void*>(conv_lp16h2_nbr_kernel<true, true>),
#ifdef MMH_AB_KERNELS
                            reinterpret_cast<const void*>(conv_lp16q_kernel<false, 1, false>),
                            reinterpret_cast<const void*>(conv_lp16q_kernel<false, -1, false>),
                            reinterpret_cast<const void*>(conv_lp16q_kernel<true, 1, false>),
                            reinterpret_cast<const void*>(conv_lp16q_kernel<true, -1, false>),
                            reinterpret_cast<const void*>(conv_lp16q_kernel<false, -1, true>),
                            reinterpret_cast<const void*>(conv_lp16q_kernel<true, -1, true>),
#endif
        };
        for (const void* f : fs)
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
        ready = e == hipSuccess ? 0 : mmh::fail("conv_lp16h2_kernel: %s", hipGetErrorString(e));
    }
    if (ready != 0) return ready;
    LpConvKP ph = p;
    ph.MT = d->B * ((d->H + HT - 1) / HT) * ((d->W + HT - 1) / HT);
    // one workgroup per tile, or - with more tiles than CUs - one PERSISTENT workgroup per CU
    // that walks its XCD's tiles (mmh_set_option("lp16_persist", 0): off)
    int wpx = (ph.MT * ph.NT + 7) / 8;
    if (mmh::g_lp16_persist && (mode != 2 || mmh::g_lp16_persist == 1)) {     // option value 2: not the reflect-fold variant
        static int cus = 0;
        if (!cus) {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
                n = 256;
            cus = n;
        }
        wpx = std::min(wpx, cus / 8);
    }
    const dim3 grid(8 * wpx);
#define MMH_LAUNCH_HALO(KERNEL, THREADS)                                                                                   \
    do {                                                                                                                   \
        if (mode == 2) {                                                                                                   \
            if (p.h16) hipLaunchKernelGGL((KERNEL<true, -1, true>), grid, dim3(THREADS), lds2, st, ph);                    \
            else hipLaunchKernelGGL((KERNEL<false, -1, true>), grid, dim3(THREADS), lds2, st, ph);                         \
        } else if (p.h16 && mode == 0) hipLaunchKernelGGL((KERNEL<true, 1, false>), grid, dim3(THREADS), lds2, st, ph);    \
        else if (p.h16) hipLaunchKernelGGL((KERNEL<true, -1, false>), grid, dim3(THREADS), lds2, st, ph);                  \
        else if (mode == 0) hipLaunchKernelGGL((KERNEL<false, 1, false>), grid, dim3(THREADS), lds2, st, ph);              \
        else hipLaunchKernelGGL((KERNEL<false, -1, false>), grid, dim3(THREADS), lds2, st, ph);                            \
    } while (0)
    if (p.nbr_part) {
        MMH_REQUIRE(!solo && mode != 0 && p.y16 && !(ph.dbg & 128) && p.nbr_x && p.nbr_mean && p.nbr_invstd,
                    "conv_lp16h2_nbr_kernel: a 16-bit dgrad on the two-waves-per-SIMD kernel with 16-byte stores");
        if (mode == 2) {
            if (p.h16) hipLaunchKernelGGL((conv_lp16h2_nbr_kernel<true, true>), grid, dim3(512), lds2, st, ph);
            else hipLaunchKernelGGL((conv_lp16h2_nbr_kernel<false, true>), grid, dim3(512), lds2, st, ph);
        } else {
            if (p.h16) hipLaunchKernelGGL((conv_lp16h2_nbr_kernel<true, false>), grid, dim3(512), lds2, st, ph);
            else hipLaunchKernelGGL((conv_lp16h2_nbr_kernel<false, false>), grid, dim3(512), lds2, st, ph);
        }
        return mmh::check_launch("conv_lp16h2_nbr_kernel");
    }
#ifdef MMH_AB_KERNELS
    if (solo) { MMH_LAUNCH_HALO(conv_lp16q_kernel, 256); return mmh::check_launch("conv_lp16q_kernel"); }
#endif
    MMH_LAUNCH_HALO(conv_lp16h2_kernel, 512);
#undef MMH_LAUNCH_HALO
    return mmh::check_launch("conv_lp16h2_kernel");
}

} }  // namespace mmh::lp16
