/*
 * mmhand_hip.h — C-ABI of libmmhand_hip.so (MI355X / gfx950 only).
 *
 * The reference (VITA-Group/mm-hand) has no FFI of its own: its hot path,
 * MMHandModel.optimize_parameters() (models/MMHandModel.py:310-330), reaches
 * the GPU only through torch.nn modules.  This header therefore declares the
 * operator set those modules imply; every entry point
 * names the reference call site it replaces.
 *
 * Conventions
 *  - Activations are NHWC fp32 in HBM: element (b,h,w,c) of a tensor with
 *    channel stride `cs` sits at ((b*H + h)*W + w)*cs + c.  Channel counts are
 *    multiples of 4 (callers zero-pad 3/6/42-channel tensors to 4/8/44).
 *  - Conv weights are [kh][kw][Cin][Cout] ("RSCK", Cout contiguous) for Conv2d
 *    and [kh][kw][Cout_T][Cin_T] for ConvTranspose2d — the same physical
 *    object as the Conv2d whose dgrad the transposed conv is.
 *  - All device buffers are owned by the caller.  Nothing here allocates,
 *    frees or synchronises; every call enqueues on the caller's hipStream_t.
 *  - Return value: 0 on success, non-zero on error (mmh_last_error() gives the
 *    thread-local message).  No exceptions cross this boundary.
 */
#ifndef MMHAND_HIP_H
#define MMHAND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the entry points declared between this push and
 * the pop at the end of the header are its ONLY exported symbols (tests/test_host_cpu.py checks
 * `nm -D` against this header in both directions). */
#pragma GCC visibility push(default)

typedef void* mmh_stream_t; /* a hipStream_t */

enum { MMH_PAD_ZERO = 0, MMH_PAD_REFLECT = 1 };
enum { MMH_ACT_NONE = 0, MMH_ACT_RELU = 1, MMH_ACT_TANH = 2 };
enum { MMH_F32 = 0, MMH_BF16 = 1, MMH_FP16 = 2 };

/* Geometry of one Conv2d (forward orientation).  For ConvTranspose2d fill it
 * in as the Conv2d whose input-gradient the transposed conv computes. */
typedef struct mmh_conv_desc {
    int32_t B, H, W;      /* input  (x)  batch and spatial size            */
    int32_t Cin, Cout;    /* channels of w ([kh][kw][Cin][Cout]); %4 == 0   */
    int32_t kh, kw;       /* 3x3 or 7x7 in the reference                    */
    int32_t stride;       /* 1 or 2                                         */
    int32_t pad;          /* symmetric padding                              */
    int32_t pad_mode;     /* MMH_PAD_ZERO | MMH_PAD_REFLECT (stride 1 only) */
    int32_t Ho, Wo;       /* output (y) spatial size                        */
    int32_t x_cs, y_cs;   /* channel strides of x and y buffers (>= C)      */
    int32_t dtype;        /* MMH_F32, or MMH_BF16 / MMH_FP16 = 16-bit MFMA  */
                          /* operands (bf16, or IEEE fp16 as apex O1 uses), */
                          /* fp32 accumulation.  What is 16-bit IN HBM is   */
                          /* said per entry point: the first-generation     */
                          /* kernels (mmh_conv2d_*, mmh_convT2d_*) read     */
                          /* fp32 x / dy and round while staging unless     */
                          /* their io16 / lp16 argument says a tensor is    */
                          /* 16-bit; the conv_lp16 family (mmh_conv3x3_lp16,*/
                          /* mmh_conv_lp16, mmh_conv_lp16_flat, mmh_wgrad*  */
                          /* _lp16*) takes 16-bit x / dy and writes fp32 or */
                          /* 16-bit outputs (y_is16).  `w` is then a tensor */
                          /* made by mmh_prep_weights_bf16 / _fp16 (fprop / */
                          /* convT dgrad: w_t, or w_flat when Cin % 64 != 0;*/
                          /* dgrad / convT fprop: w_plain, Cout % 64 == 0). */
} mmh_conv_desc;

const char* mmh_last_error(void);
int mmh_version(void);
/* Tuning knobs for A/B measurements inside one process: kernel variants ("lp16_shape" 19 = the halo
 * kernel (default) | 17 = row tiles | 20 = one wave per SIMD (make AB=1 builds only), "lp16_wgrad_ring" 2 | 3,
 * "lp16_wgrad_s2" 0 | 1 (the stride-2 3x3 weight gradients on the flat-row kernel | on the nine-tap halo kernel,
 * default), "conv_dbuf",
 * "wino_gemm_levels", ...) and work-list parameters ("wgrad_slots", "conv_xcd", ...): the results
 * stay within the kernels' documented tolerances.  The "*_dbg" keys are NOT such knobs: "conv_dbg",
 * "lp16_dbg" (bits 1-16), "dgrad_s2_dbg", "stem_f32_dbg" switch parts of a kernel OFF for timing
 * ablations (tools/ablate*.py, tools/bench_lp16_fold.py) and make its results wrong ("lp16_dbg" bit 32 only
 * moves the DMA issue of half the waves, results unchanged: tools/ab_lp16_stagger.py); they default to 0
 * and nothing in mmhand_amd/ sets them.  Unknown keys are an error.                       */
int mmh_set_option(const char* key, int value);

/* ---- convolutions: nn.Conv2d / nn.ReflectionPad2d / nn.ConvTranspose2d ----
 * replaces models/Generator.py:40-113,158-259, models/Discriminator.py:14-99,
 * losses/L1_plus_perceptualLoss.py:22-27 (cuDNN kernels behind torch.nn).   */

/* y = act(conv(pad(x), w) + bias).  bias may be NULL. */
int mmh_conv2d_fprop(const mmh_conv_desc* d, const void* x, const void* w,
                     const void* bias, void* y, int act, mmh_stream_t s);

/* mmh_conv2d_fprop (fp32, no activation) that also writes, per (128-row tile, wave row) of the output GEMM,
 * the count / mean / M2 of every output column: stats [chunks][3][Cout], chunks =
 * mmh_conv2d_fprop_stats_chunks(d) (0: not available - B*Ho*Wo % 128 != 0 or a 16-bit dtype).  The norm
 * behind the conv (models/Generator.py:158-175 stems and down convs) merges them with
 * mmh_norm_stats_merge[_finalize] instead of reading y again; consecutive chunks cover consecutive pixels, so
 * with Ho*Wo % 128 == 0 the chunks of an image are contiguous (InstanceNorm: groups = B).              */
int mmh_conv2d_fprop_stats_chunks(const mmh_conv_desc* d);
int mmh_conv2d_fprop_stats(const mmh_conv_desc* d, const void* x, const void* w, const void* bias,
                           void* y, void* stats, mmh_stream_t s);
/* 1 when mmh_conv2d_dgrad / mmh_conv2d_dgrad_folded / mmh_convT2d_fprop run conv `d` (fp32, 3x3, stride 2,
 * zero pad 1, 64 -> 128 channels, dense dy) on the halo-resident kernel of dgrad_s2.hip - the 9 x 17 dy halo of
 * an 8 x 16 block of dy positions in LDS for all nine taps and all four output parity classes - instead of the
 * four parity-class implicit GEMMs (mmh_set_option("dgrad_s2_halo", 0) forces those).  Replaces the data
 * gradient of nn.Conv2d(ngf, 2 ngf, 3, 2, 1), models/Generator.py:192-199, and the forward of
 * nn.ConvTranspose2d(2 ngf, ngf, 3, 2, 1, output_padding=1), models/Generator.py:212-219.                      */
int mmh_dgrad_s2_halo_supported(const mmh_conv_desc* d, int dx_cs);

/* Gradient w.r.t. x.  MMH_PAD_ZERO: dx is [B,H,W,Cin] (channel stride
 * dx_cs).  MMH_PAD_REFLECT: dx is the gradient on the padded domain,
 * [B,H+2p,W+2p,Cin]; fold it with mmh_reflect_fold.                        */
int mmh_conv2d_dgrad(const mmh_conv_desc* d, const void* dy, const void* w,
                     void* dx, int dx_cs, mmh_stream_t s);

/* Gradient w.r.t. x, already folded back onto [B,H,W,Cin] for either pad mode.  For the
 * 3x3 / pad 1 reflect convs (the PATBlock / ResnetBlock stack) no padded buffer exists: a
 * tile-aligned zero-pad dgrad on the real domain plus one multi-piece launch that adds the
 * eight border terms of the pad ring.  Larger reflect pads go through `ws`
 * (mmh_conv2d_dgrad_folded_ws_bytes) and mmh_reflect_fold.                       */
size_t mmh_conv2d_dgrad_folded_ws_bytes(const mmh_conv_desc* d);
int mmh_conv2d_dgrad_folded(const mmh_conv_desc* d, const void* dy, const void* w,
                            void* dx, void* ws, size_t ws_bytes, mmh_stream_t s);

/* ---- Winograd for the fp32 3x3 / stride 1 / pad 1 convs (the PATBlock / ResnetBlock stack) ----
 * y = A^T[(G g G^T) . (B^T d B)]A with tile = 2 (F(2x2,3x3): 16 planes, 2.25x fewer
 * multiplications than the direct implicit GEMM), tile = 4 (F(4x4,3x3): 36 planes, 4x fewer;
 * fp32 error 2.4e-6 relative at K = 512) or tile = 6 (F(6x6,3x3): 64 planes, 5.06x fewer, 5e-6;
 * ragged tiles: any H, W >= 4).  tiles = B*ceil(H/tile)*ceil(W/tile) (tile 2, 4: H, W % tile
 * == 0); planes P = (tile+2)^2.
 *   U  [P][K][N]     = mmh_wino_weights(w)  (flip_transpose=1: the dgrad filter [P][Cout][Cin])
 *   V  [P][tiles][C] = mmh_wino_input(x)    (reflect or zero padding folded into the gather)
 *   M  [P][tiles][N] = mmh_wino_gemm(V, U)  (P batched GEMMs, one launch)
 *   y                = mmh_wino_output(M)   (+bias, activation)
 * dgrad: the same three stages on dy with zero padding and the flipped filter give g on the
 *        real domain; mmh_conv2d_dgrad_border adds the eight reflect-border terms.
 * wgrad: Yh = mmh_wino_dy(dy) = A dY A^T; dU = mmh_wino_wgrad_gemm(V, Yh) (P split-K GEMMs over
 *        the tiles, deterministic slab reduction); dw = mmh_wino_dw(dU) = G^T dU G.
 * dtype = MMH_F32: every Winograd-domain tensor is fp32 (tile 2 or 4).
 * dtype = MMH_BF16 (tile 2 only; the --opt_level O1/O2 path): U, V, M and Yh are bf16 in HBM,
 *        transforms compute in fp32 and round once, the GEMMs run on the bf16 MFMA with fp32
 *        accumulation; x, y, dy, dU, dw, bias stay fp32.  bf16 U is [P][N][K] (contraction index
 *        contiguous).  Needs K % 64 == 0, N % 32 == 0 (wgrad: Cin, Cout % 128 == 0).        */
int mmh_wino_weights(const void* w, int Cin, int Cout, int flip_transpose, int tile, int dtype,
                     void* U, mmh_stream_t s);
/* reflect: 0 = zero padding 1, 1 = reflection padding 1, 2 (tile 6, fp32) = zero padding 2 with
 * tiles over the (H+2) x (W+2) outputs of the full correlation: the first stage of the
 * REFLECT-FOLD dgrad.  The gradient of a ReflectionPad2d(1) conv lives on the padded domain and
 * its transpose adds ring row -1 onto row 1 and ring row H onto row H-2; with the tile origin on
 * the ring, each ring pixel and its partner sit in one 6x6 tile, so mmh_wino_output(fold=1) adds
 * them in registers: dgrad costs the same 64 GEMMs as fprop and no border terms.  Needs
 * (H+1) % 6 >= 2 and (W+1) % 6 >= 2 (true for 16, 64, 128: the shapes of the path); tiles =
 * B*ceil((H+2)/6)*ceil((W+2)/6).                                                           */
int mmh_wino_input(const void* x, int B, int H, int W, int C, int reflect, int tile, int dtype,
                   void* V, mmh_stream_t s);
int mmh_wino_dy(const void* dy, int B, int H, int W, int C, int tile, int dtype, void* Yh,
                mmh_stream_t s);
/* Both backward transforms of dy in one pass (tile 6, fp32): V = mmh_wino_input(dy, zero pad) for
 * the dgrad GEMMs and Yh = mmh_wino_dy(dy) for the wgrad GEMMs; dy is read once.
 * fold != 0: V is taken on the padded domain as mmh_wino_input(reflect=2) does (needs, besides
 * its conditions, ceil((H+2)/6) == ceil(H/6) so that V and Yh share one tile grid).           */
int mmh_wino_input_dy(const void* dy, int B, int H, int W, int C, int tile, int dtype, void* V,
                      void* Yh, int fold, mmh_stream_t s);
int mmh_wino_gemm(const void* V, const void* U, void* M, int64_t tiles, int K, int N,
                  int nbatch, int dtype, mmh_stream_t s);
/* The fp32 GEMMs with the summation chosen per call: levels = 2 folds every 32-deep k-step's chain into the
 * totals (the default of mmh_wino_gemm for 64 planes: the F(6x6,3x3) output transform amplifies the rounding of
 * one k-ordered chain), levels = 1 is one chain (5-6 % faster).  The package runs the FORWARD GEMMs with 2 and
 * the dgrad GEMMs with 1: a forward difference of 7e-6 flips ReLU masks and moves the parameter gradients of
 * this network by 3e-3, a backward one cannot (profiles/r03_wino_grad_split.txt: Winograd dgrad alone 6e-6).  */
int mmh_wino_gemm_levels(const void* V, const void* U, void* M, int64_t tiles, int K, int N,
                         int nbatch, int levels, mmh_stream_t s);
/* The F(6x6,3x3) filter transform of MANY fp32 filters in one launch (after an optimizer step a network's 74
 * transforms are 8-25 us launches that cannot fill the chip).  table: n rows of six int64 in device memory -
 * {w pointer, U pointer, Cin, Cout, flip_transpose, first block} - entry e owning the 256-thread blocks
 * [first block of e, first block of e + 1), ceil(Cin Cout / 256) of them; total_blocks their sum.  Each entry
 * gets exactly what mmh_wino_weights(w, Cin, Cout, flip_transpose, 6, MMH_F32, U) writes.                  */
int mmh_wino_weights_multi(const void* table, int n, int64_t total_blocks, mmh_stream_t s);
/* stats (tile 6, fp32; may be NULL): [B][tiles per image][3][C] floats = per (image, tile, channel)
 * the count, mean and M2 of the tile's outputs - the partial-statistics layout that
 * mmh_norm_stats_merge reduces, so the InstanceNorm after the conv does not re-read y.
 * fold != 0 (tile 6, fp32, no bias / act / stats): M is on the padded domain (see mmh_wino_input,
 * reflect = 2); y receives the reflect-folded gradient on the real H x W domain.               */
int mmh_wino_output(const void* M, void* y, const void* bias, int B, int H, int W, int C,
                    int act, int tile, int dtype, void* stats, int fold, mmh_stream_t s);
size_t mmh_wino_wgrad_gemm_ws_bytes(int64_t tiles, int Cin, int Cout, int nbatch);
int mmh_wino_wgrad_gemm(const void* V, const void* Yh, int64_t tiles, int Cin, int Cout,
                        int nbatch, int dtype, void* ws, size_t ws_bytes, void* dU,
                        mmh_stream_t s);
int mmh_wino_dw(const void* dU, int Cin, int Cout, int tile, void* dw, int accumulate,
                mmh_stream_t s);
size_t mmh_conv2d_dgrad_border_ws_bytes(const mmh_conv_desc* d);
/* phase 1 = the eight border GEMMs (dy, w -> ws; independent of dx, so the caller may run
 * them on a second stream beside the Winograd transforms), 2 = add ws into dx, 3 = both.
 * io16: bit 0 = dy, bit 1 = dx is a 16-bit tensor of d->dtype (gradients of a 16-bit convolution). */
int mmh_conv2d_dgrad_border(const mmh_conv_desc* d, const void* dy, const void* w, void* dx,
                            void* ws, size_t ws_bytes, int phase, int io16, mmh_stream_t s);

/* dw[kh][kw][Cin][Cout] (+)= sum over pixels.  Split-K partial slabs go to
 * `ws` (mmh_conv2d_wgrad_ws_bytes); the fixed-order second stage makes the
 * result deterministic.  accumulate!=0 adds into dw.                       */
size_t mmh_conv2d_wgrad_ws_bytes(const mmh_conv_desc* d);
/* io16 (16-bit dtypes): 0 = x and dy are fp32 (converted on load), 3 = both are already 16-bit in HBM. */
int mmh_conv2d_wgrad(const mmh_conv_desc* d, const void* x, const void* dy,
                     void* dw, void* ws, size_t ws_bytes, int accumulate,
                     int io16, mmh_stream_t s);

/* ConvTranspose2d(k3,s2,p1,op1) (models/Generator.py:240-253).  `d` is the
 * stride-2 Conv2d of which this is the dgrad: d->{H,W,Cin} describe the
 * transposed conv's OUTPUT, d->{Ho,Wo,Cout} its INPUT.                     */
int mmh_convT2d_fprop(const mmh_conv_desc* d, const void* x, const void* w,
                      const void* bias, void* y, int y_cs, int act,
                      mmh_stream_t s);
int mmh_convT2d_dgrad(const mmh_conv_desc* d, const void* dy, const void* w,
                      void* dx, mmh_stream_t s);
int mmh_convT2d_wgrad(const mmh_conv_desc* d, const void* x, const void* dy,
                      void* dw, void* ws, size_t ws_bytes, int accumulate,
                     int io16, mmh_stream_t s);

/* fp32 weights [taps][Cin][Cout] -> bf16 copies for the MMH_BF16 conv path:
 * w_plain [taps][Cin][Cout] (contraction = Cout, used by dgrad) and
 * w_t [taps][Cout][Cin] (contraction = Cin, used by fprop).  Either may be NULL. */
int mmh_prep_weights_bf16(const void* w, int taps, int Cin, int Cout,
                          void* w_plain, void* w_t, mmh_stream_t s);
/* The same two copies as IEEE fp16 (MMH_FP16; values beyond +-65504 become inf).          */
int mmh_prep_weights_fp16(const void* w, int taps, int Cin, int Cout,
                          void* w_plain, void* w_t, mmh_stream_t s);
/* Both copies of MANY weights in one launch (after an optimizer step every 16-bit copy of a network is stale: 78
 * conversions of 5-15 us per 16-bit iteration).  table: n rows of eight int64 in device memory - {w pointer, w_plain
 * pointer, w_t pointer (either copy may be 0), taps, Cin, Cout, first block, 1 for fp16 | 0 for bf16} - entry e owning
 * the 256-thread blocks [first block of e, first block of e + 1), taps ceil(Cin / 64) ceil(Cout / 64) of them; total_blocks
 * their sum.  Each entry gets exactly what mmh_prep_weights_bf16 | _fp16 writes.  The optimizer the reference steps
 * (models/MMHandModel.py:317-330, apex O1 casting weights per forward) is where the copies go stale.           */
int mmh_prep_weights_lp16_multi(const void* table, int n, int64_t total_blocks, mmh_stream_t s);
int mmh_prep_weights_fp16_flat(const void* w, int taps, int Cin, int Cout,
                               void* w_flat, mmh_stream_t s);
/* For fprop of convs whose Cin is not a multiple of 64 (the 7x7 stems): w_flat is
 * [Cout][Kpad] bf16, Kpad = ceil(taps*Cin/64)*64, contraction index k = tap*Cin + ci. */
int mmh_prep_weights_bf16_flat(const void* w, int taps, int Cin, int Cout,
                               void* w_flat, mmh_stream_t s);

/* Diagnostic builds only (make AB=1; mmh_set_option("lp16_dbg", 4096)): after launches of the 16-bit halo kernel, copies
 * per workgroup the pair (delta s_memtime, delta s_memrealtime) that wave 0 stamped around the kernel body into
 * host_pairs_u64[2 * max_workgroups]; delta s_memtime / delta s_memrealtime x 100 MHz is the clock the chip held under the
 * kernel's load.  Synchronises the device.  Returns the number of workgroups copied: 0 in ordinary builds (no stamp is
 * compiled into the shipped kernel), -1 on a copy error.                                                                */
int mmh_lp16_clock_stamps(void* host_pairs_u64, int max_workgroups);

/* ---- 16-bit direct 3x3 convolution, both operands 16-bit in HBM (conv_lp16.hip) ----------------
 * The second-generation MFMA kernel of the --opt_level O1/O2 path for the 3x3 / stride 1 / pad 1
 * stack (channels % 64 == 0, output channels % 256 == 0): 256x256x64 block tile, both operands
 * global -> LDS by LDS-DMA with an XOR-swizzled image, two 64 KiB stages, one barrier per k-step.
 *   x16   16-bit NHWC activations (the twin of the fp32 tensor: mmh_cvt_lp16, or written by the
 *         producer), pixel stride d->x_cs (mode 0) / d->y_cs (mode 1) elements
 *   w16   mode 0 (fprop): w_t [tap][Cout][Cin]; mode 1 (dgrad): w_plain [tap][Cin][Cout]
 *         (mmh_prep_weights_bf16 / _fp16)
 *   y     fp32 (y_is16 = 0) or 16-bit (y_is16 = 1) NHWC output, +bias, activation
 *   zeros >= 128 zero bytes of device memory (what out-of-image taps read)
 * mode 1 computes the zero-padded correlation with the flipped filter; for MMH_PAD_REFLECT the
 * caller adds the border terms (mmh_conv2d_dgrad_border, phase 3) afterwards.
 * mmh_wgrad3x3_lp16: dw [3][3][Cin][Cout] fp32 (+)= wgrad from the 16-bit x and dy (Cin, Cout
 * % 256 == 0): [64 pixels][256 channels] tiles by LDS-DMA, both MFMA operands read transposed
 * (ds_read_b64_tr_b16), split-K over pixel ranges with fp32 slabs in ws, fixed-order reduction.  */
int mmh_cvt_lp16(const void* x, int64_t n, int dtype, void* out, mmh_stream_t s);
size_t mmh_wgrad3x3_lp16_ws_bytes(const mmh_conv_desc* d);
int mmh_wgrad3x3_lp16(const mmh_conv_desc* d, const void* x16, const void* dy16, void* dw,
                      void* ws, size_t ws_bytes, int accumulate, const void* zeros,
                      mmh_stream_t s);
int mmh_conv3x3_lp16_supported(const mmh_conv_desc* d);
/* mode 2 of mmh_conv3x3_lp16: the dgrad of a ReflectionPad2d(1) conv complete in one launch -
 * the gradient of the pad ring is folded onto rows 1 / H-2 and columns 1 / W-2 inside the halo
 * kernel (H, W multiples of 16, Cin % 256 == 0), so no mmh_conv2d_dgrad_border call follows.     */
int mmh_conv3x3_lp16_fold_supported(const mmh_conv_desc* d);
int mmh_conv3x3_lp16(const mmh_conv_desc* d, int mode, const void* x16, const void* w16,
                     const void* bias, void* y, int y_is16, int act, const void* zeros,
                     mmh_stream_t s);
/* fprop (16-bit y, no activation) that also writes the partial statistics of its output from the
 * epilogue: stats [B][chunks][3][Cout] floats = count / mean / M2 per (image, half pixel tile,
 * channel) of the values as stored - the mmh_norm_stats_merge[_finalize] layout, so the InstanceNorm
 * behind the conv (models/Generator.py:66-77) does not read y for statistics.  chunks =
 * mmh_conv3x3_lp16_stats_chunks(d) = 2 (H/16)(W/16); 0 = not available (H or W not a multiple of 16). */
/* dgrad (mode 1 | 2) of the 3x3 stack with a 16-bit dx that is the gradient of a norm's OUTPUT (conv -> norm -> ReLU ->
 * Dropout -> pad -> conv, models/Generator.py:66-77, models/Discriminator.py:35-48): the epilogue also takes that norm's
 * backward sums s1 = sum dz, s2 = sum dz * xhat per (group, channel) - dz = keep ? g * dsc : 0 of the values as stored -
 * as partials per (image, half tile) in ws, which mmh_norm_bwd_sums_final adds up: mmh_norm_bwd_reduce's pass over g, x
 * and the keep bits is gone.  xn: the norm's input, 16-bit [B,H,W,Cin] contiguous; bits: its keep bits (16 per 8
 * elements, mmh_scale_shift_act's) or NULL; mean / invstd fp32 [groups][Cin], groups = B (InstanceNorm) | 1
 * (BatchNorm); s1 / s2 fp32 [groups][Cin]; ws >= B * chunks * 2 * Cin floats, 16-byte aligned.
 * _chunks: partials per image, 0 = not available (ragged tiles, strided dx, another kernel selected).               */
int mmh_conv3x3_lp16_dgrad_nbr_chunks(const mmh_conv_desc* d, int mode);
int mmh_conv3x3_lp16_dgrad_nbr(const mmh_conv_desc* d, int mode, const void* dy16, const void* w16, void* dx16,
                               const void* xn, const void* bits, const void* mean, const void* invstd, int groups,
                               float drop_p, void* s1, void* s2, void* ws, size_t ws_bytes, const void* zeros,
                               mmh_stream_t s);
/* partials [groups][chunks][2][C] -> s1, s2 [groups][C]: the second half of mmh_norm_bwd_reduce alone */
int mmh_norm_bwd_sums_final(const void* part, int groups, int C, int chunks, void* s1, void* s2, mmh_stream_t s);

/* dgrad (mode 1 | 2) with an fp32 dx that also receives `addend` (fp32, same layout as dx) in the epilogue:
 * dx = dgrad(dy) + addend.  The conv's input has a second consumer - the residual stream of a PATBlock
 * (models/Generator.py:115-130) or a ResnetBlock (models/Discriminator.py:50) - whose gradient
 * autograd would otherwise add in a pass of its own (35 launches, 1.7 ms per 16-bit step).        */
int mmh_conv3x3_lp16_dgrad_add_supported(const mmh_conv_desc* d);
int mmh_conv3x3_lp16_dgrad_add(const mmh_conv_desc* d, int mode, const void* dy16, const void* w16,
                               const void* addend, void* dx, const void* zeros, mmh_stream_t s);
int mmh_conv3x3_lp16_stats_chunks(const mmh_conv_desc* d);
int mmh_conv3x3_lp16_fprop_stats(const mmh_conv_desc* d, const void* x16, const void* w16,
                                 const void* bias, void* y16, void* stats, const void* zeros,
                                 mmh_stream_t s);

/* The same machine for the other 3x3 / pad 1 convolutions of the step (stride 2 down-sampling,
 * ConvTranspose2d, 64 / 128 output channels): mode 0 fprop, mode 1 dgrad (stride 2: the four
 * output-parity classes in one launch; ConvTranspose2d(k3,s2,p1,op1).forward is mode 1 of the
 * stride-2 conv it is the adjoint of).  Needs Cin, Cout % 64 == 0, even H and W for stride 2.
 * Operands as for mmh_conv3x3_lp16.  models/Generator.py:165-223,240-253, Discriminator.py:86-99. */
int mmh_conv_lp16_supported(const mmh_conv_desc* d, int mode);
int mmh_conv_lp16(const mmh_conv_desc* d, int mode, const void* x16, const void* w16,
                  const void* bias, void* y, int y_is16, int act, const void* zeros,
                  mmh_stream_t s);
/* fprop (mode 0) with a 16-bit output AND the partial statistics of that output for the InstanceNorm behind the conv,
 * where the stride-2 kernel of conv_s2_lp16.hip takes the shape (3x3 / stride 2 / zero pad 1, Cin 64, Cout %% 128 == 0,
 * Ho %% 8 == 0, Wo %% 16 == 0): stats [B][chunks][3][Cout], chunks = mmh_conv_lp16_stats_chunks(d) (0: use
 * mmh_conv_lp16 and mmh_norm_stats).                                                                                  */
int mmh_conv_lp16_stats_chunks(const mmh_conv_desc* d);
int mmh_conv_lp16_fprop_stats(const mmh_conv_desc* d, const void* x16, const void* w16, const void* bias, void* y16,
                              void* stats, const void* zeros, mmh_stream_t s);

/* Flat-K 16-bit fprop for the 7x7 stems (Cin = 3..42: models/Generator.py:158-164,
 * models/Discriminator.py:79-84): contraction index k = tap * C8 + c over the channels padded to C8
 * (a multiple of 8, <= 64).  x16p = mmh_lp16_pad_cvt(x) [B][H][W][C8]; w_flat =
 * mmh_prep_weights_lp16_flat8(w) [Cout][roundup(kh*kw*C8, 64)]; stride 1, 'same' zero / reflect padding. */
int mmh_lp16_pad_cvt(const void* x, int64_t rows, int C, int C8, int dtype, void* out, mmh_stream_t s);
int mmh_prep_weights_lp16_flat8(const void* w, int taps, int Cin, int Cout, int C8, int dtype,
                                void* out, mmh_stream_t s);
int mmh_conv_lp16_flat_supported(const mmh_conv_desc* d, int C8);
int mmh_conv_lp16_flat(const mmh_conv_desc* d, const void* x16p, int C8, const void* w_flat,
                       const void* bias, void* y, int y_is16, int act, const void* zeros,
                       mmh_stream_t s);

/* 16-bit wgrad with flat (tap, channel) rows - the stems, the stride-2 convs and ConvTranspose2d under
 * apex O1 (models/Generator.py:158-223,240-253, models/Discriminator.py:79-99): dw [kh][kw][Cin][Cout]
 * fp32 (+)= from x16 [B][H][W][x_cs] (C8 channels per tap read: C8 = Cin, or the padded width of
 * mmh_lp16_pad_cvt) and dy16 [B][Ho][Wo][y_cs].  ConvTranspose2d: x16 := its output gradient, dy16 :=
 * its input, d = the stride-2 conv it is the adjoint of (as mmh_convT2d_wgrad).              */
int mmh_wgrad_lp16_flat_supported(const mmh_conv_desc* d, int C8);
size_t mmh_wgrad_lp16_flat_ws_bytes(const mmh_conv_desc* d, int C8);
int mmh_wgrad_lp16_flat(const mmh_conv_desc* d, const void* x16, int C8, int x_cs, const void* dy16,
                        void* dw, void* ws, size_t ws_bytes, int accumulate, const void* zeros,
                        mmh_stream_t s);

/* 16-bit fprop of the same stems with the 22 x 22 pixel input halo of a 16 x 16 pixel tile resident in LDS and
 * the filter's column taps flattened into the contraction (conv_stem16.hip): the pixel fragment is read
 * straight from the flat halo row at pixel pitch C8, no im2col tile is staged (mmh_conv_lp16_flat moves 40 KB
 * per 64-deep k-step).  x16p [B,H,W,C8] as above; w_stem16 from mmh_prep_weights_stem16 (the stem's fp32
 * weight [7][7][Cin][64] as 16-bit [7][64][32 ceil(7 C8 / 32) + 8], mmh_conv_stem16_weights_bytes(C8) bytes);
 * y fp32 or 16-bit (y_is16) [B,H,W,y_cs], +bias, activation.                                         */
int mmh_conv_stem16_supported(const mmh_conv_desc* d, int C8);
size_t mmh_conv_stem16_weights_bytes(int C8);
/* the same kernel with a 3x3 / pad 1 filter at C8 == 8 (ks = 3; ks = 7: the functions above): VGG19's conv1_1 (3 -> 64 at
 * full resolution, losses/L1_plus_perceptualLoss.py:22-27) - mmh_conv_stem16 takes it from the descriptor's kh           */
size_t mmh_conv_stem16_weights_bytes_k(int C8, int ks);
int mmh_prep_weights_stem16_k(const void* w, int Cin, int C8, int ks, int dtype, void* out, mmh_stream_t s);
int mmh_prep_weights_stem16(const void* w, int Cin, int C8, int dtype, void* out, mmh_stream_t s);
int mmh_conv_stem16(const mmh_conv_desc* d, const void* x16p, int C8, const void* w_stem16,
                    const void* bias, void* y, int y_is16, int act, const void* zeros,
                    mmh_stream_t s);
/* The same launch also leaving the partial statistics of its 16-bit output for the InstanceNorm behind the stem
 * (H, W multiples of 16; no activation): stats [B][chunks][3][64] (n, mean, M2 per wave tile and channel) in the layout
 * mmh_norm_stats_merge_finalize reduces - the norm does not read y for statistics.  chunks = mmh_conv_stem16_stats_chunks
 * (0: not available for this shape).                                                                                  */
int mmh_conv_stem16_stats_chunks(const mmh_conv_desc* d, int C8);
int mmh_conv_stem16_stats(const mmh_conv_desc* d, const void* x16p, int C8, const void* w_stem16, const void* bias,
                          void* y16, void* stats, const void* zeros, mmh_stream_t s);

/* 16-bit wgrad of the 7x7 / stride 1 / pad 3 stems with 64 output channels (models/Generator.py:158-164,
 * models/Discriminator.py:60-64): x16p [B,H,W,C8] = the stem's 16-bit input with its channels padded
 * to C8 (mmh_lp16_pad_cvt; C8 % 8 == 0, 8..48), dy16 [B,H,W,y_cs >= 64].  The filter's column taps are
 * flattened into the operand (a row of the NHWC input is one contiguous array: the window of pixel
 * ow starts at element ow * C8), the contraction runs over the pixels of a row, both MFMA operands
 * are read transposed from an LDS halo staged once per 4 x 16 pixel block for all 49 taps; split-K
 * slabs in ws, summed in a fixed order.  dw [7][7][Cin][64] fp32 (+)= ...                          */
int mmh_wgrad_stem_lp16_supported(const mmh_conv_desc* d, int C8);
size_t mmh_wgrad_stem_lp16_ws_bytes(const mmh_conv_desc* d, int C8);
int mmh_wgrad_stem_lp16(const mmh_conv_desc* d, const void* x16p, int C8, const void* dy16, void* dw,
                        void* ws, size_t ws_bytes, int accumulate, const void* zeros,
                        mmh_stream_t s);

/* Weight gradient of the Generator head (ReflectionPad2d(3) + Conv2d(64, 3, 7) + Tanh, models/Generator.py:254-259;
 * Cout = 3 padded to 4) in 16 bits: d describes the head conv (Cin = 64, Cout = 4, reflect pad 3, dtype BF16 | FP16),
 * x16 = its 16-bit input [B,H,W,x_cs >= 64], dy = the fp32 gradient of its output [B,H,W,y_cs >= 4] (after the Tanh
 * backward).  dw[kh][kw][ci][co] = sum over q of the padded domain of xpad[q][ci] * E[q + (3-kh, 3-kw)][co] with E = dy
 * embedded at offset (3,3): the stem wgrad above on the padded domain with the roles of the operands swapped (E is the
 * 8-channel "input", the 64 channels of x the "output gradient", read through reflected source addresses), mirrored and
 * transposed by the slab reduction.  dw [7][7][64][4] fp32 (+)= ...; ws 16-byte aligned.  Replaces the fp32 vector-ALU
 * kernel mmh_conv7_thin_wgrad in 16-bit mode.                                                                        */
int mmh_conv7_head_wgrad_lp16_supported(const mmh_conv_desc* d);
size_t mmh_conv7_head_wgrad_lp16_ws_bytes(const mmh_conv_desc* d);
int mmh_conv7_head_wgrad_lp16(const mmh_conv_desc* d, const void* x16, const void* dy, void* dw, void* ws,
                              size_t ws_bytes, int accumulate, const void* zeros, mmh_stream_t s);

/* Transpose of ReflectionPad2d(p): dx[b,h,w,c] = sum of dxp over the padded
 * positions that mirror onto (h,w).  dxp is [B,H+2p,W+2p,C].               */
int mmh_reflect_fold(const void* dxp, void* dx, int B, int H, int W, int C,
                     int p, mmh_stream_t s);

/* out[c] (+)= sum over rows of x[rows][C] (conv-bias gradient).  x_dtype: element type of x
 * (MMH_F32 | MMH_BF16 | MMH_FP16); sums are fp32.                          */
size_t mmh_colsum_ws_bytes(int64_t rows, int C);
int mmh_colsum(const void* x, int64_t rows, int C, int cs, void* out,
               void* ws, size_t ws_bytes, int accumulate, int x_dtype, mmh_stream_t s);

/* ---- BatchNorm2d / InstanceNorm2d + ReLU + Dropout ------------------------
 * replaces norm_layer / nn.ReLU / nn.Dropout sites, models/Generator.py:66-77,
 * models/Discriminator.py:29-34, models/network_utils.py:74-84.
 * The *_dtype arguments name the element type of one activation / gradient tensor in HBM
 * (MMH_F32 | MMH_BF16 | MMH_FP16).  Under --opt_level O1 the tensors that face a convolution
 * (its output, its input, both gradients) are 16-bit, as under apex; statistics, scale / shift,
 * sums and all arithmetic stay fp32.                                        */

/* Per-(group,channel) mean and M2 = sum (x-mean)^2.  groups = B for instance
 * norm (rows_per_group = H*W), 1 for batch norm (rows_per_group = B*H*W).   */
size_t mmh_norm_stats_ws_bytes(int groups, int64_t rows_per_group, int C);
int mmh_norm_stats(const void* x, int groups, int64_t rows_per_group, int C,
                   int cs, void* mean, void* m2, void* ws, size_t ws_bytes,
                   int x_dtype, mmh_stream_t s);

/* Chan merge of `chunks` partial (count, mean, M2) triples per (group, channel), laid out
 * [groups][chunks][3][C] (what mmh_wino_output(stats) writes): the second stage of mmh_norm_stats. */
int mmh_norm_stats_merge(const void* partials, int groups, int chunks, int C, void* mean,
                         void* m2, mmh_stream_t s);

/* The same merge for ONE group over many chunks in two levels (BatchNorm statistics from conv-epilogue
 * partials, models/network_utils.py:74-84 with --norm batch: B * chunks-per-image partials per channel):
 * `sub` blocks of chunks (chunks % sub == 0) are merged in parallel into [sub][3][C] floats in ws, which
 * a single-group merge finishes.                                                                      */
size_t mmh_norm_stats_merge2_ws_bytes(int sub, int C);
int mmh_norm_stats_merge2(const void* partials, int chunks, int C, int sub, void* ws,
                          size_t ws_bytes, void* mean, void* m2, mmh_stream_t s);

/* mmh_norm_stats_merge + mmh_norm_finalize of a norm without affine parameters and running statistics
 * (nn.InstanceNorm2d) in one launch: the same fp32 mean / M2 / scale / shift / invstd as the two calls.   */
int mmh_norm_stats_merge_finalize(const void* partials, int groups, int chunks, int C, double count,
                                  float eps, void* mean, void* m2, void* scale, void* shift,
                                  void* invstd, mmh_stream_t s);
/* SyncBN (apex convert_syncbn_model, models/MMHandModel.py:109-116): the (count, mean, M2) triples of one norm site as the
 * ranks' all-gather delivered them - rank r's triple at gathered + r * rank_stride floats, [3][C] - merged in rank order
 * (Chan) AND finalised (affine, running statistics with momentum and the unbiased variance) in one launch: the same values as
 * mmh_norm_stats_merge on the [ranks][3][C] block followed by mmh_norm_finalize, without the block copy and the two launches.
 * count = rows over all ranks.                                                                                          */
int mmh_syncbn_merge_finalize(const void* gathered, int ranks, int64_t rank_stride, int C, double count,
                              float eps, const void* gamma, const void* beta, void* mean, void* m2,
                              void* scale, void* shift, void* invstd, void* running_mean,
                              void* running_var, float momentum, mmh_stream_t s);
/* scale = gamma*rsqrt(m2/count+eps), shift = beta - mean*scale, invstd.
 * gamma/beta may be NULL (affine=False).  If running_mean != NULL (batch
 * norm, groups==1) they are updated with momentum and the unbiased variance
 * m2/(count-1), as nn.BatchNorm2d does in training mode.                   */
int mmh_norm_finalize(const void* mean, const void* m2, double count,
                      const void* gamma, const void* beta, float eps,
                      int groups, int C, void* scale, void* shift,
                      void* invstd, void* running_mean, void* running_var,
                      float momentum, mmh_stream_t s);

/* out = dropout(relu(x*scale[g][c] + shift[g][c])) (+ residual).  scale and
 * shift are [groups][C].  Dropout keeps an element when a counter-based hash
 * of (seed, element position) clears the drop threshold and scales by 1/(1-p);
 * if mask != NULL (uint8 per element, test hook) it is used instead.
 * keep_bits != NULL (uint8 per 4 channels): bit e = lane e survived ReLU / dropout -
 * all the backward needs of `out`, at 1/16 of its bytes.
 * out_dtype: MMH_F32, or MMH_BF16 / MMH_FP16 = `out` is written as that 16-bit type
 * (the input of a 16-bit convolution: no fp32 copy, no conversion pass).      */
int mmh_scale_shift_act(const void* x, const void* scale, const void* shift,
                        const void* residual, void* out, int groups,
                        int64_t rows_per_group, int C, int relu, float drop_p,
                        uint64_t seed, const void* mask, void* keep_bits,
                        int x_dtype, int out_dtype, mmh_stream_t s);
/* The same pass writing `out` AND the same values in 16 bits to `twin` (twin_dtype MMH_BF16 | MMH_FP16): a norm output with
 * two consumers - the residual stream of a ResnetBlock (models/Discriminator.py:50), a PATBlock's first stream-1 input -
 * stays fp32 for the residual add, and the 3x3 conv that reads it takes the twin instead of a conversion pass of its
 * own (mmh_cvt_lp16: 6 B per element against 2).  Needs C / 8 a power of two <= 256.                                      */
int mmh_scale_shift_act_twin(const void* x, const void* scale, const void* shift,
                             const void* residual, void* out, int groups,
                             int64_t rows_per_group, int C, int relu, float drop_p,
                             uint64_t seed, const void* mask, void* keep_bits,
                             int x_dtype, int out_dtype, void* twin, int twin_dtype, mmh_stream_t s);

/* Backward of norm+relu+dropout.  dz = g * (relu||drop ? (out>0)/(1-p) : 1).
 * masked: 0 = no ReLU / dropout (`out` unused), 1 = `out` is the fp32 forward output,
 * 2 = `out` is the keep_bits array written by mmh_scale_shift_act.
 * reduce: s1[g][c] = sum dz, s2[g][c] = sum dz*xhat, xhat = (x-mean)*invstd.
 * apply:  dx = gamma*invstd*(dz - s1/count - xhat*s2/count).               */
size_t mmh_norm_bwd_ws_bytes(int groups, int64_t rows_per_group, int C);
int mmh_norm_bwd_reduce(const void* g, const void* out, const void* x,
                        const void* mean, const void* invstd, int groups,
                        int64_t rows_per_group, int C, int masked,
                        float drop_p, void* s1, void* s2, void* ws,
                        size_t ws_bytes, int g_dtype, int x_dtype, mmh_stream_t s);
int mmh_norm_bwd_apply(const void* g, const void* out, const void* x,
                       const void* mean, const void* invstd,
                       const void* gamma, const void* s1, const void* s2,
                       double count, int groups, int64_t rows_per_group,
                       int C, int masked, float drop_p, void* dx,
                       int g_dtype, int x_dtype, int dx_dtype, mmh_stream_t s);

/* The same backward in ONE pass for small planes (InstanceNorm at 64x64 and below; 16-bit x): a workgroup holds the whole
 * (group, 8-16 channels) plane in registers - g, x and the keep bits are read once, s1 / s2 summed on chip (and written
 * out: the affine parameters' gradients need them), dx applied.  Same per-element arithmetic as the two entry points
 * above, different (fixed) order of the plane sums.  Not for SyncBN (its sums cross ranks between the two passes).
 * mmh_norm_bwd_fused_supported: 1 when (groups, rows, C, masked, types) fit - rows <= 8192 / 8-channel lane groups.  */
int mmh_norm_bwd_fused_supported(int groups, int64_t rows, int C, int masked, int g_dtype, int x_dtype);
int mmh_norm_bwd_fused(const void* g, const void* out, const void* x, const void* mean, const void* invstd,
                       const void* gamma, double count, int groups, int64_t rows, int C, int masked, float drop_p,
                       void* s1, void* s2, void* dx, int g_dtype, int x_dtype, int dx_dtype, mmh_stream_t s);

/* ---- norm-apply fused into the consuming convolution (fp32, Winograd F(6x6,3x3) stack) ----
 * The reference runs conv -> Norm -> ReLU -> Dropout -> ReflectionPad -> conv as separate modules
 * (models/Generator.py:66-77, models/Discriminator.py:29-34).  Here the apply pass of the norm between
 * two 3x3 convs runs inside the second conv's input transform (mmh_wino_input_normact), and the apply
 * pass of its backward inside the first conv's backward transform (mmh_wino_input_dy_normbwd): the
 * activation between norm and conv, and the gradient between conv and norm, never reach HBM.  No
 * keep bits are stored either: the backward decides again from fma(x, scale, shift) > 0 - the
 * expression the forward evaluated - and the dropout bit array of the site.
 *
 * mmh_dropout_bits: bit e of byte i of `bits` [n/8] = element 8 i + e is kept (hash of (seed,
 * position) against p, or mask != NULL: uint8 per element, test hook).
 * *_rc: mmh_norm_bwd_reduce / _apply (fp32 tensors) with the keep decision taken again instead of
 * read; dbits NULL iff drop_p == 0.  C / 8 must be a power of two <= 256.                      */
int mmh_dropout_bits(int64_t n, float drop_p, uint64_t seed, const void* mask, void* bits,
                     mmh_stream_t s);
/* the same decisions as row words for the two transforms below, which walk one channel along image rows:
 * rows (uint32) [image_rows = B*H][ceil(W/32)][C], bit k of word j = element (row, 32 j + k, c)      */
int mmh_dropout_bits_rows(const void* bits, int64_t image_rows, int W, int C, void* rows,
                          mmh_stream_t s);
/* both arrays of an [image_rows][W][C] tensor in one launch (same decisions as the two calls above) */
int mmh_dropout_bits_both(int64_t image_rows, int W, int C, float drop_p, uint64_t seed,
                          const void* mask, void* bits, void* rows, mmh_stream_t s);
int mmh_norm_bwd_reduce_rc(const void* g, const void* x, const void* mean, const void* invstd,
                           const void* scale, const void* shift, const void* dbits, int groups,
                           int64_t rows_per_group, int C, int relu, float drop_p, void* s1,
                           void* s2, void* ws, size_t ws_bytes, mmh_stream_t s);
int mmh_norm_bwd_apply_rc(const void* g, const void* x, const void* mean, const void* invstd,
                          const void* gamma, const void* s1, const void* s2, const void* scale,
                          const void* shift, const void* dbits, double count, int groups,
                          int64_t rows_per_group, int C, int relu, float drop_p, void* dx,
                          mmh_stream_t s);
/* V = mmh_wino_input(dropout(relu(x*scale + shift)))   (tile 6, fp32; reflect 0 | 1; groups 1 | B;
 * drows = mmh_dropout_bits_rows, NULL iff drop_p == 0)                                           */
int mmh_wino_input_normact(const void* x, int B, int H, int W, int C, int reflect, void* V,
                           const void* scale, const void* shift, int groups, int relu,
                           float drop_p, const void* drows, mmh_stream_t s);
/* (V, Yh) = mmh_wino_input_dy(dx), dx = the result of mmh_norm_bwd_apply_rc(g, x, ...) computed per
 * element inside the transform (x = the norm's input = the forward output of the conv whose
 * backward this is).                                                                            */
int mmh_wino_input_dy_normbwd(const void* g, const void* x, int B, int H, int W, int C, void* V,
                              void* Yh, int fold, const void* mean, const void* invstd,
                              const void* gamma, const void* s1, const void* s2, double count,
                              const void* scale, const void* shift, const void* drows, int groups,
                              int relu, float drop_p, mmh_stream_t s);

/* dx = g * act'(y): relu -> (y>0), tanh -> 1-y^2 (Generator.py:259 head).   */
int mmh_act_bwd(const void* g, const void* y, void* dx, int64_t n, int act,
                mmh_stream_t s);
/* the same product written in 16 bits only (dtype MMH_BF16 | MMH_FP16; n % 8 == 0): the activation
 * backward of a conv whose dgrad is a 16-bit kernel and whose weights are frozen (VGG19 conv1_1 /
 * conv1_2 + ReLU, losses/L1_plus_perceptualLoss.py:22-27) - one pass instead of mmh_act_bwd +
 * mmh_cvt_lp16.                                                                               */
int mmh_act_bwd_lp16(const void* g, const void* y, int64_t n, int act, int dtype, void* out16,
                     mmh_stream_t s);
/* ... with g and / or y already in 16 bits (g_is16, y_is16; the storage type is `dtype`): the 16-bit edge between VGG19's
 * conv1_1 and conv1_2 (ops.VggPairFn) - the gradient comes out of conv1_2's 16-bit dgrad, the mask from conv1_1's 16-bit
 * output.                                                                                                      */
int mmh_act_bwd_lp16_io(const void* g, int g_is16, const void* y, int y_is16, int64_t n, int act, int dtype,
                        void* out16, mmh_stream_t s);

/* ---- PATBlock gate + concat (models/Generator.py:115-130) -----------------
 * out = x1 + s1*sigmoid(s2)*sigmoid(s3);  x2n = cat(s3,out); x3n = cat(s2,out)
 * x1,s1,s2,s3,out: [rows][C];  x2n,x3n: [rows][2C] (NULL -> not produced).
 * cat_dtype: element type of x2n / x3n, s23_dtype: of s2 / s3 (conv outputs)
 * (MMH_F32 | MMH_BF16 | MMH_FP16); x1, s1 and `out` are always fp32.        */
int mmh_patblock_gate_fwd(const void* x1, const void* s1, const void* s2,
                          const void* s3, void* out, void* x2n, void* x3n,
                          int64_t rows, int C, int cat_dtype, int s23_dtype, mmh_stream_t s);
/* g_out/g_x2n/g_x3n may be NULL (treated as zero).  gcat_dtype: element type of g_x2n / g_x3n
 * (gradients from 16-bit convolutions), s23_dtype: of s2 / s3, gs23_dtype: of g_s2 / g_s3. */
int mmh_patblock_gate_bwd(const void* g_out, const void* g_x2n,
                          const void* g_x3n, const void* s1, const void* s2,
                          const void* s3, void* g_x1, void* g_s1, void* g_s2,
                          void* g_s3, int64_t rows, int C, int gcat_dtype, int s23_dtype,
                          int gs23_dtype, mmh_stream_t s);

/* The gate with the block's LAST InstanceNorm inside (16-bit mode; models/Generator.py:66-77 + 115-130): s1 is not
 * materialised - the gate reads the stream-1 conv output y2 (16-bit, [groups * rows][C]) and scale / shift [groups][C] and
 * evaluates s1 = fma(y2, scale, shift) itself.  Backward: besides the gate's four gradients (g_s1 = the gradient of the
 * norm's output, fp32) the same launch leaves the norm backward's sums sum1[g][c] = sum g_s1, sum2[g][c] = sum g_s1 * xhat
 * (xhat = (y2 - mean) * invstd), taken in mmh_norm_bwd_reduce's chunks and order: follow it with mmh_norm_bwd_apply
 * (masked 0).  Bit-identical to mmh_scale_shift_act + mmh_patblock_gate_fwd and to mmh_patblock_gate_bwd +
 * mmh_norm_bwd_reduce.  ws: mmh_norm_bwd_ws_bytes(groups, rows, C).  x2n / x3n as in mmh_patblock_gate_fwd.   */
int mmh_patblock_gate_norm_supported(int groups, int64_t rows_per_group, int C);
int mmh_patblock_gate_norm_fwd(const void* x1, const void* y2, const void* scale, const void* shift,
                               const void* s2, const void* s3, void* out, void* x2n, void* x3n, int groups,
                               int64_t rows_per_group, int C, int y_dtype, int cat_dtype, int s23_dtype,
                               mmh_stream_t s);
int mmh_patblock_gate_norm_bwd(const void* g_out, const void* g_x2n, const void* g_x3n, const void* y2,
                               const void* scale, const void* shift, const void* mean, const void* invstd,
                               const void* s2, const void* s3, void* g_x1, void* g_s1, void* g_s2, void* g_s3,
                               void* sum1, void* sum2, void* ws, size_t ws_bytes, int groups,
                               int64_t rows_per_group, int C, int y_dtype, int gcat_dtype, int s23_dtype,
                               mmh_stream_t s);

/* ---- losses ----------------------------------------------------------------
 * GANLoss = BCEWithLogits vs a constant target, mean (network_utils.py:129-163)
 * L1 mean (L1_plus_perceptualLoss.py:37,66-67).  `out` is one device float:
 * out = weight * mean(...) with the mean taken over `denom` elements.       */
size_t mmh_reduce_ws_bytes(int64_t n);
int mmh_bce_logits_fwd(const void* x, int64_t n, float target, float weight,
                       double denom, void* out, void* ws, size_t ws_bytes,
                       mmh_stream_t s);
/* dx = gscalar[0] * weight/denom * (sigmoid(x) - target)                    */
int mmh_bce_logits_bwd(const void* x, int64_t n, float target, float weight,
                       double denom, const void* gscalar, void* dx,
                       mmh_stream_t s);
int mmh_l1_fwd(const void* a, const void* b, int64_t n, float weight,
               double denom, void* out, void* ws, size_t ws_bytes,
               mmh_stream_t s);
/* da = gscalar[0] * weight/denom * sign(a-b)                                */
int mmh_l1_bwd(const void* a, const void* b, int64_t n, float weight,
               double denom, const void* gscalar, void* da, mmh_stream_t s);
/* The perceptual term on 16-bit VGG features (losses/L1_plus_perceptualLoss.py:60-66 under apex O1, where the VGG
 * convolutions return fp16 and F.l1_loss runs on them): a16, b16 are the bf16 / fp16 feature maps (`dtype`), the
 * difference and the sum are fp32.  n % 8 == 0; ws as for mmh_l1_fwd (mmh_reduce_ws_bytes(n)).
 * mmh_l1_relu_bwd_lp16 is the L1 gradient folded with the mask of the ReLU that produced a16 (features[3]), written in
 * `dtype` for the 16-bit dgrad of conv1_2:  out16 = [a16 > 0] * gscalar[0] * weight/denom * sign(a16 - b16)          */
int mmh_l1_fwd_lp16(const void* a16, const void* b16, int64_t n, float weight,
                    double denom, int dtype, void* out, void* ws, size_t ws_bytes,
                    mmh_stream_t s);
int mmh_l1_relu_bwd_lp16(const void* a16, const void* b16, int64_t n, float weight,
                         double denom, const void* gscalar, int dtype, void* out16,
                         mmh_stream_t s);

/* MSE mean (F.mse_loss, the --percep_is_l1 0 branch of losses/L1_plus_perceptualLoss.py:68-71):
 * out = weight * sum (a-b)^2 / denom;  da = gscalar[0] * weight/denom * 2 (a-b)            */
int mmh_mse_fwd(const void* a, const void* b, int64_t n, float weight,
                double denom, void* out, void* ws, size_t ws_bytes,
                mmh_stream_t s);
int mmh_mse_bwd(const void* a, const void* b, int64_t n, float weight,
                double denom, const void* gscalar, void* da, mmh_stream_t s);

/* MaxPool2d(2, 2) of vgg19.features (indices 4, 9, 18, 27, 36) for --perceptual_layers > 3
 * (losses/L1_plus_perceptualLoss.py:22-27 slices the feature stack at any index).  NHWC fp32, even H and W, C % 4 == 0.
 * bwd: dx [B,H,W,C] receives g [B,H/2,W/2,C] at the first maximum of each window (scan order), zeros elsewhere.  */
int mmh_maxpool2x2_fwd(const void* x, int B, int H, int W, int C, void* y, mmh_stream_t s);
int mmh_maxpool2x2_bwd(const void* x, const void* g, int B, int H, int W, int C, void* dx, mmh_stream_t s);

/* ---- thin 7x7 convolutions (at most 4 output channels) on the vector ALU -----
 * The Generator head ReflectionPad2d(3)+Conv2d(64,3,7)+Tanh (models/Generator.py:255-259)
 * and the dgrad of the Discriminator stems (models/Discriminator.py:79-84) with respect
 * to the generated image only, i.e. the first <= 4 of their input channels
 * (MMHandModel.backward_G, models/MMHandModel.py:236-261: P2 / H1 carry no gradient).
 * A 32-wide MFMA tile wastes 7/8 of the matrix core on 4 columns; see conv_thin.hip.
 * fprop: d->Cout == 4, d->Cin % 4 == 0, 7x7 / stride 1 / pad 3 (reflect or zero), fp32;
 *        w [7][7][Cin][4]; same result as mmh_conv2d_fprop.
 * dgrad: writes channels [0,4) of dx (pixel stride d->x_cs) and leaves the others alone;
 *        w is the conv's full weight [7][7][d->Cin][d->Cout]; ws from the _ws_bytes query;
 *        dy_dtype: element type of dy (MMH_F32 | MMH_BF16 | MMH_FP16), arithmetic fp32. */
int mmh_conv7_thin_fprop(const mmh_conv_desc* d, const void* x, const void* w,
                         const void* bias, void* y, int act, mmh_stream_t s);
size_t mmh_conv7_thin_dgrad_ws_bytes(const mmh_conv_desc* d);
int mmh_conv7_thin_dgrad(const mmh_conv_desc* d, const void* dy, const void* w,
                         void* dx, void* ws, size_t ws_bytes, int dy_dtype, mmh_stream_t s);
/* wgrad of the head (d->Cout == 4, d->Cin % 64 == 0): dw [7][7][Cin][4] (+)= sum over pixels;
 * per-workgroup partial sums in ws, added in a fixed order (deterministic).              */
size_t mmh_conv7_thin_wgrad_ws_bytes(const mmh_conv_desc* d);
int mmh_conv7_thin_wgrad(const mmh_conv_desc* d, const void* x, const void* dy, void* dw,
                         void* ws, size_t ws_bytes, int accumulate, mmh_stream_t s);

/* The input gradient of the Generator head (ReflectionPad2d(3) + Conv2d(64, 3, 7): models/Generator.py:254-259, reached
 * from the L1 / perceptual / GAN losses of models/MMHandModel.py:236-275) in 16-bit mode, as a stem-shaped convolution
 * (conv_stem16.hip): dy fp32 [B,H,W,y_cs] (channels 0..3: the head's Cout padded to 4) is embedded, in 16 bits and 8
 * channels, into the zero-padded (H+6) x (W+6) domain, convolved 'same' with the mirrored, transposed filter (4 -> 64
 * channels) and the pad ring folded back (the transpose of ReflectionPad2d(3)): dx fp32 or 16-bit [B,H,W,x_cs] (64
 * channels written).  d describes the head conv itself (Cin = 64, Cout = 4, 7x7, stride 1, MMH_PAD_REFLECT, 16-bit
 * dtype); w is its fp32 weight [7][7][64][4]; ws >= mmh_conv7_head_dgrad_lp16_ws_bytes(d), 256-byte aligned.          */
int mmh_conv7_head_dgrad_lp16_supported(const mmh_conv_desc* d);
size_t mmh_conv7_head_dgrad_lp16_ws_bytes(const mmh_conv_desc* d);
int mmh_conv7_head_dgrad_lp16(const mmh_conv_desc* d, const void* dy, const void* w, void* dx, int dx_is16, void* ws,
                              size_t ws_bytes, const void* zeros, mmh_stream_t s);

/* The same two convolutions from 16-bit tensors on the 16-column MFMA (conv7_n4.hip): an (8+6) x (16+6) pixel halo of
 * the 64-channel input and the whole [49][4][64] filter resident in LDS, 4 of the 16 weight rows meaningful.
 * mode 0: fprop of the Generator head (models/Generator.py:254-259): x16 [B,H,W,x_cs >= 64] 16-bit, w the head's fp32
 *         weight [7][7][64][Cout <= 4], y fp32 [B,H,W,y_cs] (channels 0..3 written), +bias, activation.
 * mode 1: gradient of a Discriminator stem towards its first four input channels (models/Discriminator.py:60-64 from
 *         models/MMHandModel.py:238-243): x16 = dy16 [B,H,W,y_cs >= 64], w the stem's fp32 weight [7][7][Cin][64],
 *         y = dx fp32 [B,H,W,x_cs]: channels [0,4) written, the others left alone (as mmh_conv7_thin_dgrad).  With
 *         MMH_PAD_REFLECT the kernel runs over the padded domain into ws and the pad ring is folded back.
 *         Also 3x3 / pad 1 / zero padding with Cin == 4: VGG19 conv1_1 seen from the perceptual loss
 *         (losses/L1_plus_perceptualLoss.py:22-27,60-67).
 * ws (mmh_conv7_n4_lp16_ws_bytes): the 16-bit filter twin, built by the call, + the padded-domain gradient.        */
int mmh_conv7_n4_lp16_supported(const mmh_conv_desc* d, int mode);
size_t mmh_conv7_n4_lp16_ws_bytes(const mmh_conv_desc* d, int mode);
int mmh_conv7_n4_lp16(const mmh_conv_desc* d, int mode, const void* x16, const void* w,
                      const void* bias, void* y, int act, void* ws, size_t ws_bytes,
                      const void* zeros, mmh_stream_t s);

/* wgrad of the 7x7 / stride 1 / ReflectionPad2d(3) stems (models/Generator.py:158-168,
 * models/Discriminator.py:60-64; fp32, Cin 8 | 44 (where it beats the generic kernel), Cout == 64 dense, H % 2 == 0,
 * W % 64 == 0): the input is reflect-padded once into ws, a workgroup stages the two input rows of a
 * filter row and a 2 x 64 pixel tile in LDS and both MFMA operands are plain LDS reads - every input
 * element is read 7 times instead of 49.  dw [7][7][Cin][64] (+)= ...; split-K slabs in ws, summed in a
 * fixed order.                                                                              */
int mmh_conv7_stem_wgrad_supported(const mmh_conv_desc* d);
size_t mmh_conv7_stem_wgrad_ws_bytes(const mmh_conv_desc* d);
int mmh_conv7_stem_wgrad(const mmh_conv_desc* d, const void* x, const void* dy, void* dw,
                         void* ws, size_t ws_bytes, int accumulate, mmh_stream_t s);

/* ---- Adam (torch.optim.Adam, MMHandModel.py:90-98) over a flat buffer ------
 * step is the 1-based step count; no weight decay, no amsgrad.
 * skip_flag (device int32, may be NULL): when *skip_flag != 0 the launch leaves
 * p, m and v untouched - the `if not self.overflow: optimizer.step()` of
 * MMHandModel.py:316-328 decided on the device, without a host round trip.
 * loss_scale (device float, may be NULL): the gradient is additionally divided by
 * *loss_scale - the unscale step of dynamic loss scaling, read on the device.  */
int mmh_adam_step(void* p, const void* g, void* m, void* v, int64_t n,
                  float lr, float beta1, float beta2, float eps, int step,
                  float grad_scale, const void* skip_flag, const void* loss_scale,
                  mmh_stream_t s);

/* The same update with the step count and the learning rate resident on the device - nothing in the launch depends on the
 * iteration, so a whole training step can sit in a captured hipGraph (MMHandModel.py:310-330 replayed).  step (device
 * int32): incremented first, unless *skip_flag != 0 (apex does not count a skipped step: no host correction afterwards);
 * lr (device float): what update_learning_rate (models/base_model.py:82-87) last wrote; coef (device float[2]): scratch
 * for lr / (1 - beta1^t) and 1 / sqrt(1 - beta2^t), computed in double as mmh_adam_step does on the host.             */
int mmh_adam_step_dev(void* p, const void* g, void* m, void* v, int64_t n, const void* lr, float beta1, float beta2,
                      float eps, void* step, float grad_scale, const void* skip_flag, const void* loss_scale, void* coef,
                      mmh_stream_t s);

/* Dropout in a captured step: every kernel that draws dropout decisions from a by-value seed (mmh_scale_shift_act[_twin],
 * mmh_dropout_bits[_both]) adds *salt (a device uint64; NULL = none, the default) to that seed when it RUNS.  The salt is
 * read from the pointer registered here at launch time; mmh_u64_add advances it on the stream (inside the graph), so each
 * replay of the same launches draws fresh masks (models/Generator.py:66-77 nn.Dropout(0.5) per forward).              */
int mmh_set_dropout_salt(const void* salt_u64);
int mmh_u64_add(void* value_u64, uint64_t inc, mmh_stream_t s);

/* ImagePool.query (util/image_pool.py:14-34) on a device-resident pool [slots][elems_per_image] fp32 with the host's
 * decisions as device int32 indices [B]: out[i] = src_idx[i] >= 0 ? pool[src_idx[i]] (content before this query)
 * : images[-1 - src_idx[i]]; then pool[dst_idx[i]] = images[i] where dst_idx[i] >= 0 (the host emits one writer per slot).
 * The launches are the same every iteration - only the index arrays change - so the query can sit in a captured step.  */
int mmh_pool_exchange(void* pool, const void* images, void* out, const void* src_idx, const void* dst_idx, int B,
                      int64_t elems_per_image, mmh_stream_t s);

/* ---- overflow detection (MMHandModel.loss_backward, MMHandModel.py:294-308) --
 * *flag_out = (flag_in ? *flag_in : 0) | any(!isfinite(g[0..n))).  flag_in carries
 * the sticky `self.overflow` of the steps already taken this iteration.  Run on
 * the flat gradient buffer AFTER the data-parallel all-reduce: a non-finite
 * value on any rank is non-finite in the sum on every rank, which is the
 * flag all-reduce of reduce_tensor (MMHandModel.py:381-384) for free.
 * own_out (device int32, may be NULL) = any(!isfinite(g)) of THIS gradient alone: what
 * apex's per-loss scaler sees when its scale_loss context exits.               */
int mmh_grad_nonfinite(const void* g, int64_t n, const void* flag_in,
                       void* flag_out, void* own_out, mmh_stream_t s);

/* ---- dynamic loss scaling (apex.amp.initialize(..., num_losses=3) + amp.scale_loss,
 * MMHandModel.py:99-108,294-299): state = {scale, clean steps} (2 device floats).  The
 * backward runs on loss * scale; mmh_adam_step(loss_scale = state) divides it out again;
 * then, as apex's LossScaler.update_scale: *overflow != 0 -> scale = max(scale * backoff,
 * min_scale), clean = 0; else ++clean, and clean == interval -> scale = min(scale * growth,
 * max_scale), clean = 0.  apex's dynamic defaults: start 2^16, growth 2, backoff 0.5,
 * interval 2000, max 2^24.                                                           */
int mmh_loss_scale_update(void* state, const void* overflow, float growth, float backoff,
                          int interval, float min_scale, float max_scale, mmh_stream_t s);

/* ---- layout: NCHW (any strides) <-> padded NHWC, with channel concat -------
 * replaces torch.cat at MMHandModel.py:216-220,238,242,278-289.            */
typedef struct mmh_plane_src {
    void* ptr;                       /* NULL -> skipped                     */
    int32_t C;                       /* channels taken from this source     */
    int64_t sb, sc, sh, sw;          /* element strides                     */
} mmh_plane_src;
/* Cd % 4 == 0, nhwc 16-byte aligned.
 * dir 0: gather srcs -> nhwc[B,H,W,Cd] (channels beyond sum(C) zeroed).
 * dir 1: scatter nhwc -> srcs (backward of dir 0 / NHWC->NCHW export).     */
int mmh_pack_nhwc(const mmh_plane_src* srcs, int nsrc, void* nhwc, int B,
                  int H, int W, int Cd, int dir, mmh_stream_t s);
/* The gather (dir 0) staged through LDS - 256 pixels per workgroup, read with the lanes along the pixels of a source
 * plane, written as whole 16-byte lanes in address order - with, in the same pass, the 16-bit copy out16 [B,H,W,C8]
 * (channels zero-padded to C8, C8 % 8 == 0, Cd <= C8 <= 56; dtype MMH_BF16 | MMH_FP16) that the 16-bit 7x7 stems read:
 * what mmh_lp16_pad_cvt would make of nhwc, bit for bit.  nhwc or out16 may be NULL (only the other one is written).
 * Cd % 4 == 0, Cd <= 56 (58 KB of LDS).  (MMHandModel.py:216-220,238,242,278-289: the concatenations in front of every stem.)      */
int mmh_pack_nhwc_lp16(const mmh_plane_src* srcs, int nsrc, void* nhwc, void* out16, int B, int H, int W,
                       int Cd, int C8, int dtype, mmh_stream_t s);

/* ---- pose maps (data/generic_dataset.py:191-217,239-242; util/util.py:94-114)
 * uv: [n_maps][2] float64 (x,y).  out: [n_maps][H][W] fp32 =
 * clamp/threshold(exp(-((gx-x)^2+(gy-y)^2)/(2 sigma^2))) computed in f64.  */
int mmh_pose_heatmaps(const void* uv, int n_maps, int H, int W, double sigma,
                      void* out, mmh_stream_t s);
/* cords[n_maps][2] int32 = (y,x) of the first (row-major) arg-max above
 * `threshold`, or (-1,-1).  maps: [n_maps][H][W] fp32.                      */
int mmh_map_to_cord(const void* maps, int n_maps, int H, int W,
                    float threshold, void* cords, mmh_stream_t s);

/* ---- on-device input pipeline (data/generic_dataset.py:133-180) ---------------
 * One kernel turns what the loader workers produce per sample on the CPU —
 * normalize(BGR->RGB image), 21 pose maps per hand, depth = 256*G+R -> /700 ->
 * (.-0.5)/0.5 replicated x3 — into the stems' NHWC buffers directly:
 *   img1,img2 : uint8 [B,H,W,3] BGR (as cv2.imread returns them)
 *   dep1,dep2 : uint8 [B,H,W,3] BGR depth PNGs
 *   uv1,uv2   : float64 [B,21,2] joint (x,y)
 *   x_h1,x_h2 : fp32 [B,H,W,4]  RGB in [-1,1], lane 3 = 0
 *   x_p       : fp32 [B,H,W,44] P1 in 0..20, P2 in 21..41, lanes 42,43 = 0
 *   x_d       : fp32 [B,H,W,8]  D1 x3, D2 x3, lanes 6,7 = 0
 * All arithmetic is float64 then cast, exactly as numpy does it in the reference. */
int mmh_decode_inputs(const void* img1, const void* img2, const void* dep1,
                      const void* dep2, const void* uv1, const void* uv2,
                      int B, int H, int W, double sigma, void* x_h1, void* x_h2,
                      void* x_p, void* x_d, mmh_stream_t s);

/* ---- data-parallel gradient all-reduce (apex DistributedDataParallel behind
 * models/MMHandModel.py:109-116; reduce_tensor :381-384) -------------------------
 * The training process already holds ONE RCCL communicator per GPU (torch.distributed's "nccl"
 * backend is RCCL on ROCm).  mmh_rccl_bind resolves ncclAllReduce from the RCCL image that
 * communicator lives in (the path of the librccl the process has loaded; no link-time
 * dependency, no second RCCL).  mmh_allreduce_bucket enqueues the in-place SUM all-reduce of
 * `count` elements of type `dtype` (MMH_F32 | MMH_BF16 | MMH_FP16) at `buf` - a contiguous
 * bucket of a network's flat gradient buffer - on communicator `comm` (an ncclComm_t) and
 * stream `s`; the 1/world factor is folded into mmh_adam_step.  Every rank must issue its
 * buckets in the same order (mmhand_amd/dp.py cuts them in reverse layer order).
 * mmh_rccl_comm_ranks: the communicator's rank count, -1 if unbound / invalid.              */
int mmh_rccl_bind(const char* librccl_path);
int mmh_rccl_comm_ranks(void* comm);
int mmh_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype, mmh_stream_t s);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* MMHAND_HIP_H */
