// Thin convolutions of the MM-HAND step: 7x7 / stride 1 with at most 4 output channels.
//
//   * the Generator head, ReflectionPad2d(3) + Conv2d(64, 3, 7) + Tanh (models/Generator.py:255-259);
//   * the dgrad of the two Discriminator stems w.r.t. the generated image, 3 of their 6 / 24 input
//     channels (models/Discriminator.py:79-84 reached from MMHandModel.backward_G, :236-261).
//
// As an implicit GEMM these have N = 4 columns: a 32-wide MFMA tile wastes 7/8 of the matrix core
// (measured 3.7-4.0 ms per launch at B=32, 256x256: 14 TFLOP/s of useful work).  Here they run on the
// vector ALU instead: one thread owns 4 vertically adjacent output pixels x 4 output channels
// (16 fp32 accumulators as 8 packed pairs), lanes run along the image row so LDS reads of the
// staged input are contiguous 16-byte lanes (conflict-free), a column of 10 staged inputs serves
// the 7 vertical taps of all 4 pixels, and the 16 weights of a (tap, 4 input channels) group are
// wave-uniform: they are read with scalar loads and fed to v_pk_fma_f32 from SGPRs, so LDS
// bandwidth (10 b128 reads per 448 FMAs per lane) is far from binding.
//
// fprop form: y[b,oh,ow,0..3] = act(bias + sum_{kh,kw,ci} x[b, oh+kh-pad, ow+kw-pad, ci] * w[kh][kw][ci][0..3])
// with reflect or zero padding and Ho = H + 2*pad - 6.  dgrad is the same form on dy with the
// flipped, transposed filter and pad 6 (full correlation) into the padded domain, then the
// reflect fold (mmh_reflect_fold's kernel) when the forward conv was reflect padded.
#include <algorithm>
#include "common.h"

namespace {

constexpr int TW = 64, TH = 16;         // output tile: 64 columns (lanes) x 16 rows (4 waves x 4 rows)
constexpr int HC = TW + 6, HR = TH + 6; // staged input tile incl. the 7x7 halo
constexpr unsigned OOBT = 0xFFFFFFF0u;

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct ThinKP {
    const float* x;
    unsigned x_bytes;
    int B, H, W, Cin, x_cs;     // input NHWC; channels [0, Cin) used, Cin % 4 == 0
    const float* w;             // [7][7][Cin][4]
    const float* bias;          // 4 floats or null
    float* y;
    int Ho, Wo, y_cs;           // output NHWC; channels [0, 4) written
    int pad, reflect, act;
    int tiles_x, tiles_y;
    int x_lp;                   // element type of x: 0 fp32, 1 bf16, 2 fp16 (x_bytes in that type)
};

__device__ __forceinline__ float thin_act(float v, int act) {
    if (act == MMH_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MMH_ACT_TANH) return tanhf(v);
    return v;
}

__global__ void __launch_bounds__(256) thin_conv7_kernel(const ThinKP p) {
    __shared__ float4 xs[HR * HC];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int oh0 = ty * TH, ow0 = tx * TW;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);

    // this thread's staged pixels: byte offsets of channel 0 (or OOBT outside a zero-padded image)
    constexpr int NLD = (HR * HC + 255) / 256;
    unsigned soff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / HC, c = idx - r * HC;
        int ih = oh0 + r - p.pad, iw = ow0 + c - p.pad;
        bool ok = idx < HR * HC;
        if (p.reflect) {
            ih = ih < 0 ? -ih : ih;
            iw = iw < 0 ? -iw : iw;
            ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
            iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
            ok = ok && ih >= 0 && iw >= 0 && ih < p.H && iw < p.W;   // tiles hanging over the image edge
        } else {
            ok = ok && ih >= 0 && iw >= 0 && ih < p.H && iw < p.W;
        }
        soff[i] = ok ? (unsigned)(((b * p.H + ih) * p.W + iw) * p.x_cs) * (p.x_lp ? 2u : 4u) : OOBT;
    }

    f2 acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[q][0] = (f2){0.f, 0.f}; acc[q][1] = (f2){0.f, 0.f}; }

    for (int c0 = 0; c0 < p.Cin; c0 += 4) {
        float4 st[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (p.x_lp) {       // 4 channels = 8 bytes of a 16-bit tensor
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, soff[i] == OOBT ? OOBT : soff[i] + c0 * 2u, 0, 0);
                if (p.x_lp == 1) {
                    st[i] = make_float4(__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u),
                                        __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u));
                } else {
                    // (element-wise: hipcc / ROCm 7.2 dropped the conversion of the second dword when both were taken
                    // through 2 x _Float16 vectors - channels 2, 3 came out as copies of 0, 1; tools/thin_probe.py)
                    auto h = [](unsigned bits) { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits); };
                    st[i] = make_float4(h(v[0] & 0xffffu), h(v[0] >> 16), h(v[1] & 0xffffu), h(v[1] >> 16));
                }
            } else {
                const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, soff[i] == OOBT ? OOBT : soff[i] + c0 * 4u, 0, 0);
                st[i] = __builtin_bit_cast(float4, v);
            }
        }
        __syncthreads();        // the previous chunk's readers are done
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (tid + 256 * i < HR * HC) xs[tid + 256 * i] = st[i];
        __syncthreads();
        const float* wc = p.w + c0 * 4;
#pragma unroll 1
        for (int kw = 0; kw < 7; ++kw) {
            float4 xv[10];
#pragma unroll
            for (int r = 0; r < 10; ++r) xv[r] = xs[(wave * 4 + r) * HC + lane + kw];
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
                // 16 wave-uniform weights [ci 4][co 4] of tap (kh, kw): scalar loads
                const float* wt = wc + (size_t)(kh * 7 + kw) * p.Cin * 4;
                f2 w01[4], w23[4];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    w01[ci] = (f2){wt[ci * 4 + 0], wt[ci * 4 + 1]};
                    w23[ci] = (f2){wt[ci * 4 + 2], wt[ci * 4 + 3]};
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 xx = xv[q + kh];
                    const float xc[4] = {xx.x, xx.y, xx.z, xx.w};
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) {
                        const f2 xb = (f2){xc[ci], xc[ci]};
                        acc[q][0] = __builtin_elementwise_fma(xb, w01[ci], acc[q][0]);
                        acc[q][1] = __builtin_elementwise_fma(xb, w23[ci], acc[q][1]);
                    }
                }
            }
        }
    }

    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) { bs[0] = p.bias[0]; bs[1] = p.bias[1]; bs[2] = p.bias[2]; bs[3] = p.bias[3]; }
    const int ow = ow0 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int oh = oh0 + wave * 4 + q;
        if (oh < p.Ho && ow < p.Wo) {
            float4 o;
            o.x = thin_act(acc[q][0].x + bs[0], p.act);
            o.y = thin_act(acc[q][0].y + bs[1], p.act);
            o.z = thin_act(acc[q][1].x + bs[2], p.act);
            o.w = thin_act(acc[q][1].y + bs[3], p.act);
            *reinterpret_cast<float4*>(p.y + ((size_t)(b * p.Ho + oh) * p.Wo + ow) * p.y_cs) = o;
        }
    }
}

// wq[kh'][kw'][co][j] = w[6-kh'][6-kw'][j][co], j < 4: the flipped, transposed filter restricted to
// the first 4 input channels (the dgrad's output channels).  w is [7][7][wCin][Cout].
__global__ void thin_flip_weights_kernel(const float* __restrict__ w, int wCin, int Cout, float* __restrict__ wq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 49 * Cout * 4) return;
    const int j = i & 3;
    const int co = (i >> 2) % Cout;
    const int tap = (i >> 2) / Cout;
    const int kh = tap / 7, kw = tap - kh * 7;
    wq[i] = j < wCin ? w[(((6 - kh) * 7 + (6 - kw)) * wCin + j) * Cout + co] : 0.f;
}

// dx[b,h,w,0..3] = sum over the padded positions that ReflectionPad2d(p) maps onto (h,w) of dxp
// (4 channels; p < H, W).  Same index algebra as reflect_fold_kernel in conv_igemm.hip.
__global__ void thin_reflect_fold_kernel(const float* __restrict__ dxp, float* __restrict__ dx, int B, int H, int W,
                                         int dx_cs, int p) {
    const long long total = (long long)B * H * W;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int HP = H + 2 * p, WP = W + 2 * p;
    for (; i < total; i += stride) {
        const int w = (int)(i % W);
        const long long t = i / W;
        const int h = (int)(t % H);
        const int b = (int)(t / H);
        // padded rows / columns that reflect onto h / w
        int hs[3], ws[3], nh = 0, nw = 0;
        hs[nh++] = h + p;
        if (h >= 1 && h <= p) hs[nh++] = p - h;
        if (h <= H - 2 && h >= H - 1 - p) hs[nh++] = 2 * (H - 1) - h + p;
        ws[nw++] = w + p;
        if (w >= 1 && w <= p) ws[nw++] = p - w;
        if (w <= W - 2 && w >= W - 1 - p) ws[nw++] = 2 * (W - 1) - w + p;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int a = 0; a < nh; ++a)
            for (int c = 0; c < nw; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(dxp + ((size_t)(b * HP + hs[a]) * WP + ws[c]) * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        *reinterpret_cast<float4*>(dx + (size_t)i * dx_cs) = s;
    }
}

// ---------------------------------------------------------------------------
// wgrad of the same thin conv (the Generator head): dw[kh][kw][ci][0..3] = sum over pixels of
// x[pix + tap][ci] * dy[pix][0..3].  As a GEMM it has N = 4 columns (3.9 ms on a 32-wide MFMA
// tile).  Here: one workgroup per (image, 32-column strip, 64-channel chunk) walks down the strip;
// 7 waves = the 7 tap rows kh, lane = input channel; each thread keeps its 7 kw x 4 co = 28 sums
// in registers.  Input rows live in an 8-slot LDS ring (one new row per step, loaded while the
// previous step computes); a thread reads the 38 staged inputs of its row once and slides the
// 7-wide window over them in registers: 38 + 32 LDS reads per 448 packed FMAs.
// Partial sums per workgroup go to `slab`, thin_wgrad_reduce_kernel adds them in a fixed order.
// A strip is cut into row segments (thin_wg_segs: a function of the shape only, so the order of the sums is too) when
// images x strips x chunks alone would not fill the chip: 4 images of 512 x 512 are 64 strips on 256 CUs (1251 us;
// 32 images of 256 x 256 took 676) - each segment pays the 6 halo rows of its prologue again.
// ---------------------------------------------------------------------------
constexpr int WGW = 32;                 // strip width (output columns per workgroup)
constexpr int WGC = WGW + 6;            // staged input columns

struct ThinWgKP {
    const float* x;
    unsigned x_bytes;
    const float* dy;
    unsigned dy_bytes;
    float* slab;                        // [nblk][49][64][4]
    int B, H, W, Cin, x_cs, dy_cs, reflect;
    int strips, chunks;
    int segs, seg_rows;                 // row segments per strip, rows per segment
};

__global__ void __launch_bounds__(448) thin_wgrad7_kernel(const ThinWgKP p) {
    __shared__ float ring[8][WGC][64];
    __shared__ float4 dyrow[2][WGW];
    const int tid = threadIdx.x;
    const int lane = tid & 63, kh = tid >> 6;
    int t = blockIdx.x;
    const int chunk = t % p.chunks; t /= p.chunks;
    const int strip = t % p.strips; t /= p.strips;
    const int seg = t % p.segs;
    const int b = t / p.segs;
    const int r0 = seg * p.seg_rows, r1 = min(p.H, r0 + p.seg_rows);
    const int w0 = strip * WGW;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);

    // staging: input row `ir` (may be outside [0,H): reflected or zero) -> ring slot (ir + 3) & 7
    constexpr int NF4 = WGC * 16;       // float4 per staged row (38 pixels x 16 groups of 4 channels)
    auto row_off = [&](int ir, int idx) -> unsigned {    // byte offset of staged float4 `idx` of input row ir
        const int px = idx >> 4, c4 = idx & 15;
        int iw = w0 + px - 3;
        bool ok = idx < NF4;
        if (p.reflect) {
            ir = ir < 0 ? -ir : ir; ir = ir >= p.H ? 2 * (p.H - 1) - ir : ir;
            iw = iw < 0 ? -iw : iw; iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
        }
        ok = ok && ir >= 0 && ir < p.H && iw >= 0 && iw < p.W;
        return ok ? (unsigned)(((b * p.H + ir) * p.W + iw) * p.x_cs + chunk * 64 + c4 * 4) * 4u : OOBT;
    };
    float4 st[2];
    float4 sd = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_row = [&](int ir, int orow) {     // input row ir and the dy row of output row orow -> registers
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsx, row_off(ir, tid + 448 * i), 0, 0);
            st[i] = __builtin_bit_cast(float4, v);
        }
        if (tid < WGW) {
            const int ow = w0 + tid;
            const unsigned off = (orow < p.H && ow < p.W) ? (unsigned)(((b * p.H + orow) * p.W + ow) * p.dy_cs) * 4u : OOBT;
            sd = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsd, off, 0, 0));
        }
    };
    auto store_row = [&](int ir, int orow) {
        float* slot = &ring[(ir + 3) & 7][0][0];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 448 * i;
            if (idx < NF4) *reinterpret_cast<float4*>(slot + (idx >> 4) * 64 + (idx & 15) * 4) = st[i];
        }
        if (tid < WGW) dyrow[orow & 1][tid] = sd;
    };

    f2 acc[7][2];
#pragma unroll
    for (int k = 0; k < 7; ++k) { acc[k][0] = (f2){0.f, 0.f}; acc[k][1] = (f2){0.f, 0.f}; }

    // prologue: input rows r0 - 3 .. r0 + 2; row r0 + 3 and dy row r0 complete the first step
    for (int ir = r0 - 3; ir <= r0 + 2; ++ir) { load_row(ir, p.H); store_row(ir, r0 + 1); }   // dy slot (r0 + 1) & 1 gets zeros (unused)
    load_row(r0 + 3, r0);
    store_row(r0 + 3, r0);
    __syncthreads();
    for (int r = r0; r < r1; ++r) {
        if (r + 1 < r1) load_row(r + 4, r + 1);         // next step's new row, in flight during the FMAs
        const float* xr = &ring[(r + kh) & 7][0][lane];  // input row r + kh - 3
        float xv[WGC];
#pragma unroll
        for (int c = 0; c < WGC; ++c) xv[c] = xr[c * 64];
#pragma unroll
        for (int c = 0; c < WGW; ++c) {
            const float4 d4 = dyrow[r & 1][c];
            const f2 d01 = (f2){d4.x, d4.y}, d23 = (f2){d4.z, d4.w};
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const f2 xb = (f2){xv[c + kw], xv[c + kw]};
                acc[kw][0] = __builtin_elementwise_fma(xb, d01, acc[kw][0]);
                acc[kw][1] = __builtin_elementwise_fma(xb, d23, acc[kw][1]);
            }
        }
        if (r + 1 < r1) store_row(r + 4, r + 1);        // slot (r + 7) & 7 is not read in step r
        __syncthreads();
    }
    float* out = p.slab + ((size_t)blockIdx.x * 49 + kh * 7) * 256 + lane * 4;
#pragma unroll
    for (int kw = 0; kw < 7; ++kw)
        *reinterpret_cast<float4*>(out + kw * 256) = make_float4(acc[kw][0].x, acc[kw][0].y, acc[kw][1].x, acc[kw][1].y);
}

// dw[tap][chunk*64 + ci][0..3] (+)= sum over (image, strip) of slab[(b, strip, chunk)][tap][ci][0..3]
__global__ void thin_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nbs, int chunks,
                                         int Cin, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // float4 index into [49][Cin]
    if (i >= 49 * Cin) return;
    const int tap = i / Cin, ci = i - tap * Cin;
    const int chunk = ci >> 6, cl = ci & 63;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < nbs; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(slab + (((size_t)(k * chunks + chunk) * 49 + tap) * 64 + cl) * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4* o = reinterpret_cast<float4*>(dw) + i;
    if (accumulate) { const float4 v = *o; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    *o = s;
}

int launch_thin(ThinKP& p, hipStream_t st) {
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    const long long blocks = (long long)p.B * p.tiles_x * p.tiles_y;
    MMH_REQUIRE(blocks > 0 && blocks < (1ll << 31), "thin conv: grid too large");
    hipLaunchKernelGGL(thin_conv7_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
    return mmh::check_launch("thin_conv7_kernel");
}

bool thin_shape_ok(const mmh_conv_desc* d) {
    return d && d->kh == 7 && d->kw == 7 && d->stride == 1 && d->pad == 3 && d->dtype == MMH_F32 && d->B > 0 &&
           d->H > 3 && d->W > 3 && d->Ho == d->H && d->Wo == d->W;
}

}  // namespace

extern "C" {

int mmh_conv7_thin_fprop(const mmh_conv_desc* d, const void* x, const void* w, const void* bias, void* y, int act,
                         mmh_stream_t s) {
    MMH_REQUIRE(thin_shape_ok(d), "mmh_conv7_thin_fprop: needs a 7x7 / stride 1 / pad 3 fp32 conv");
    MMH_REQUIRE(d->Cout == 4 && d->Cin % 4 == 0 && d->Cin > 0 && d->x_cs >= d->Cin && d->x_cs % 4 == 0 &&
                    d->y_cs >= 4 && d->y_cs % 4 == 0,
                "mmh_conv7_thin_fprop: needs Cout == 4 (3 padded to 4) and channel counts %% 4 == 0");
    MMH_REQUIRE(x && w && y, "mmh_conv7_thin_fprop: NULL buffer");
    const long long xb = (long long)d->B * d->H * d->W * d->x_cs * 4;
    MMH_REQUIRE(xb < (1ll << 32) - 64, "mmh_conv7_thin_fprop: input too large for 32-bit offsets");
    ThinKP p{};
    p.x = static_cast<const float*>(x); p.x_bytes = (unsigned)xb;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.x_cs = d->x_cs;
    p.w = static_cast<const float*>(w); p.bias = static_cast<const float*>(bias);
    p.y = static_cast<float*>(y); p.Ho = d->Ho; p.Wo = d->Wo; p.y_cs = d->y_cs;
    p.pad = 3; p.reflect = d->pad_mode == MMH_PAD_REFLECT; p.act = act;
    return launch_thin(p, mmh::as_stream(s));
}

size_t mmh_conv7_thin_dgrad_ws_bytes(const mmh_conv_desc* d) {
    if (!thin_shape_ok(d)) return 0;
    const size_t wq = (size_t)49 * d->Cout * 4 * sizeof(float);
    const size_t dxp = d->pad_mode == MMH_PAD_REFLECT ? (size_t)d->B * (d->H + 6) * (d->W + 6) * 4 * sizeof(float) : 0;
    return ((wq + 255) & ~(size_t)255) + dxp;
}

int mmh_conv7_thin_dgrad(const mmh_conv_desc* d, const void* dy, const void* w, void* dx, void* ws, size_t ws_bytes,
                         int dy_dtype, mmh_stream_t s) {
    MMH_REQUIRE(dy_dtype == MMH_F32 || dy_dtype == MMH_BF16 || dy_dtype == MMH_FP16, "mmh_conv7_thin_dgrad: bad dy_dtype");
    MMH_REQUIRE(thin_shape_ok(d), "mmh_conv7_thin_dgrad: needs a 7x7 / stride 1 / pad 3 fp32 conv");
    MMH_REQUIRE(d->Cout % 4 == 0 && d->Cout > 0 && d->Cin >= 1 && d->x_cs >= 4 && d->x_cs % 4 == 0 &&
                    d->y_cs >= d->Cout && d->y_cs % 4 == 0,
                "mmh_conv7_thin_dgrad: needs Cout %% 4 == 0 and an input pixel stride of at least 4 channels");
    MMH_REQUIRE(dy && w && dx && ws && ws_bytes >= mmh_conv7_thin_dgrad_ws_bytes(d), "mmh_conv7_thin_dgrad: bad buffers");
    hipStream_t st = mmh::as_stream(s);
    const long long yb = (long long)d->B * d->Ho * d->Wo * d->y_cs * (dy_dtype == MMH_F32 ? 4 : 2);
    MMH_REQUIRE(yb < (1ll << 32) - 64, "mmh_conv7_thin_dgrad: dy too large for 32-bit offsets");
    float* wq = static_cast<float*>(ws);
    const int nq = 49 * d->Cout * 4;
    hipLaunchKernelGGL(thin_flip_weights_kernel, dim3((nq + 255) / 256), dim3(256), 0, st,
                       static_cast<const float*>(w), d->Cin, d->Cout, wq);
    const bool refl = d->pad_mode == MMH_PAD_REFLECT;
    float* dxp = refl ? reinterpret_cast<float*>(static_cast<char*>(ws) + (((size_t)nq * 4 + 255) & ~(size_t)255))
                      : static_cast<float*>(dx);
    ThinKP p{};
    p.x = static_cast<const float*>(dy); p.x_bytes = (unsigned)yb;
    p.B = d->B; p.H = d->Ho; p.W = d->Wo; p.Cin = d->Cout; p.x_cs = d->y_cs;
    p.w = wq; p.bias = nullptr; p.act = MMH_ACT_NONE; p.reflect = 0;
    p.x_lp = dy_dtype;
    if (refl) {     // padded domain (H+6)x(W+6): full correlation, then the reflect fold
        p.pad = 6; p.Ho = d->H + 6; p.Wo = d->W + 6; p.y = dxp; p.y_cs = 4;
    } else {        // zero padding: dx is the 'same' correlation with the flipped filter
        p.pad = 3; p.Ho = d->H; p.Wo = d->W; p.y = dxp; p.y_cs = d->x_cs;
    }
    if (int rc = launch_thin(p, st)) return rc;
    if (refl) {
        const long long total = (long long)d->B * d->H * d->W;
        const int blocks = (int)std::min<long long>(mmh::cdiv(total, 256), 8192);
        hipLaunchKernelGGL(thin_reflect_fold_kernel, dim3(blocks), dim3(256), 0, st, dxp, static_cast<float*>(dx), d->B,
                           d->H, d->W, d->x_cs, 3);
        return mmh::check_launch("thin_reflect_fold_kernel");
    }
    return 0;
}

// row segments per strip: one workgroup per CU (two per CU share the vector ALUs this kernel is bound by: 962 us against
// 676 at B = 32, 256 x 256), at least 32 rows per segment
static int thin_wg_segs(const mmh_conv_desc* d) {
    const long long base = (long long)d->B * ((d->W + WGW - 1) / WGW) * (d->Cin / 64);
    const long long want = (256 + base - 1) / base;
    return (int)std::max<long long>(1, std::min<long long>(want, d->H / 32));
}

size_t mmh_conv7_thin_wgrad_ws_bytes(const mmh_conv_desc* d) {
    if (!thin_shape_ok(d) || d->Cin % 64) return 0;
    const size_t nblk = (size_t)d->B * ((d->W + WGW - 1) / WGW) * (d->Cin / 64) * thin_wg_segs(d);
    return nblk * 49 * 64 * 4 * sizeof(float);
}

int mmh_conv7_thin_wgrad(const mmh_conv_desc* d, const void* x, const void* dy, void* dw, void* ws, size_t ws_bytes,
                         int accumulate, mmh_stream_t s) {
    MMH_REQUIRE(thin_shape_ok(d), "mmh_conv7_thin_wgrad: needs a 7x7 / stride 1 / pad 3 fp32 conv");
    MMH_REQUIRE(d->Cout == 4 && d->Cin % 64 == 0 && d->Cin > 0 && d->x_cs >= d->Cin && d->x_cs % 4 == 0 && d->y_cs >= 4 &&
                    d->y_cs % 4 == 0,
                "mmh_conv7_thin_wgrad: needs Cout == 4 and Cin %% 64 == 0");
    MMH_REQUIRE(x && dy && dw && ws && ws_bytes >= mmh_conv7_thin_wgrad_ws_bytes(d), "mmh_conv7_thin_wgrad: bad buffers");
    const long long xb = (long long)d->B * d->H * d->W * d->x_cs * 4, yb = (long long)d->B * d->H * d->W * d->y_cs * 4;
    MMH_REQUIRE(xb < (1ll << 32) - 64 && yb < (1ll << 32) - 64, "mmh_conv7_thin_wgrad: tensors too large for 32-bit offsets");
    hipStream_t st = mmh::as_stream(s);
    ThinWgKP p{};
    p.x = static_cast<const float*>(x); p.x_bytes = (unsigned)xb;
    p.dy = static_cast<const float*>(dy); p.dy_bytes = (unsigned)yb;
    p.slab = static_cast<float*>(ws);
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.x_cs = d->x_cs; p.dy_cs = d->y_cs;
    p.reflect = d->pad_mode == MMH_PAD_REFLECT;
    p.strips = (d->W + WGW - 1) / WGW;
    p.chunks = d->Cin / 64;
    p.segs = thin_wg_segs(d);
    p.seg_rows = (d->H + p.segs - 1) / p.segs;
    const int nblk = d->B * p.segs * p.strips * p.chunks;
    hipLaunchKernelGGL(thin_wgrad7_kernel, dim3(nblk), dim3(448), 0, st, p);
    if (int rc = mmh::check_launch("thin_wgrad7_kernel")) return rc;
    const int n4 = 49 * d->Cin;
    hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((n4 + 255) / 256), dim3(256), 0, st, p.slab, static_cast<float*>(dw),
                       d->B * p.segs * p.strips, p.chunks, d->Cin, accumulate);
    return mmh::check_launch("thin_wgrad_reduce_kernel");
}

}  // extern "C"
