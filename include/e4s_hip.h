/* e4s_hip.h -- C-ABI of libe4s_hip.so, the MI355X (gfx950) native library behind the E4S
 * hot path (Net3 regional encoder + mask-guided StyleGAN2 generator).
 *
 * Drop-in boundary: the reference binds its two native ops through pybind11
 * torch extensions
 *     fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)
 *                                   src/models/stylegan2/op/fused_bias_act.cpp:11-21
 *     upfirdn2d_op.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
 *                                   src/models/stylegan2/op/upfirdn2d.cpp:12-22
 * and does everything else (modulation, demodulation, the grouped convs, region compose, noise,
 * ToRGB, the IR-SE encoder, regional pooling, the LocalMLPs) as chains of ATen launches
 * (src/models/stylegan2/model.py:242-448, src/models/encoders/helpers.py:122-144,
 * src/models/encoders/psp_encoders.py:264-309, src/models/networks.py:15-39).  This library
 * replaces both: e4s_fused_bias_act_f32 / e4s_upfirdn2d_f32 are 1:1 replacements of the two
 * native entry points; the rest are the fused kernels those ATen chains collapse into.
 *
 * Conventions: plain C, device pointers + sizes, no torch types.  Every entry point enqueues on
 * `stream` (a hipStream_t passed as void*; NULL = the null stream) on the CURRENT device, never
 * allocates, never synchronises, and returns 0 or a hipError_t value (launch errors ARE checked,
 * unlike the reference, .cu:80).  All tensors are fp32 and contiguous.  Activations inside the
 * generator/encoder are NHWC ("pixel-major"); the public tensors (images, masks, RGB skips) are
 * NCHW as in the reference.
 */
#ifndef E4S_HIP_H
#define E4S_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library info ------------------------------------------------------------------------ */
int e4s_abi_version(void);                 /* bumped on any signature change */
const char* e4s_build_arch(void);          /* "gfx950" */

/* ---- 1:1 replacements of the reference's native ops ---------------------------------------- */

/* y[i] = act(x[i] + b[(i / step_b) % size_b]) * scale      fused_bias_act_kernel.cu:19-49
 * act*10+grad: 10/11 linear, 30 lrelu fwd, 31 lrelu grad gated on sign of ref[i], 12/32 -> 0.
 * b may be NULL (no bias), ref may be NULL unless act*10+grad == 31. */
int e4s_fused_bias_act_f32(const float* x, const float* b, const float* ref, float* y, int64_t n,
                           int step_b, int size_b, int act, int grad, float alpha, float scale,
                           void* stream);

/* Up-FIR-down on x viewed as [major, in_h, in_w, minor]      upfirdn2d_kernel.cu:52-137,140-272
 * (true convolution: kernel flipped; negative pads crop).  y is [major, out_h, out_w, minor] with
 * out = (in*up + pad0 + pad1 - k) / down + 1.  Any up/down >= 1 and kernel size <= 16x16
 * (the reference leaves unmatched modes undefined, .cu:172-175,216). */
int e4s_upfirdn2d_f32(const float* x, const float* k, float* y, int major, int in_h, int in_w, int minor,
                      int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                      int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* grad_bias[c] = sum over all i with channel c of g[i]      op/fused_act.py:33-38 */
int e4s_channel_sum_f32(const float* g, float* out, int64_t n, int step_b, int size_b, void* stream);

/* ---- style prologue (model.py:276-281) ----------------------------------------------------- */

/* out[g, o] = post( sum_i pre(in[g*in_stride + i]) * M[o*K + i] )
 * mode 0 (modulation, EqualLinear model.py:159-162): post = acc*scale + bias[o]
 * mode 1 (demodulation, model.py:279-281): pre = x*x, post = scale * rsqrt(scale*scale*acc + 1e-8)
 *        (M = sum_k W^2 [Cout,Cin]; the conv scale 1/sqrt(9*Cin) is folded into the coefficient) */
int e4s_rowdot_f32(const float* in, int64_t in_stride, const float* M, const float* bias, float* out,
                   int G, int O, int K, int mode, float scale, void* stream);
/* e4s_rowdot_f32 for many (layer) jobs in ONE launch: out_base[job.out_off + g*O + o] from in_base[job.in_off + g*in_stride + k].
 * `jobs` is DEVICE memory (njobs records); M / bias inside a record are device pointers (bias may be NULL); K % 4 == 0;
 * max_O / max_G: the largest O / G over the jobs (grid extent). mode as e4s_rowdot_f32. */
typedef struct e4s_rowdot_job {
    int64_t in_off, in_stride, out_off;
    const float* M;
    const float* bias;
    int G, O, K;
    float scale;
} e4s_rowdot_job;
int e4s_rowdot_multi_f32(const e4s_rowdot_job* jobs, int njobs, const float* in_base, float* out_base, int max_O, int max_G,
                         int mode, void* stream);

/* wsq[co, ci] = sum_k w[co, ci, k]^2 (w is [Cout, Cin, taps]) */
int e4s_weight_sqsum_f32(const float* w, float* wsq, int cout, int cin, int taps, void* stream);

/* Weight re-packing for e4s_conv_mfma_f32 (done once per weight version, cached by the host):
 *   e4s_pack_taps_f32:         w [Cout,Cin,taps] -> out [taps][Cout][Cin]
 *   e4s_polyphase_weights_f32: up-sampling conv: w [Cout,Cin,3,3] and the 4x4 blur kernel k4
 *     (conv_transpose2d stride 2 + Blur pad (1,1), model.py:287-300) -> out [4 phases][9][Cout][Cin] */
int e4s_pack_taps_f32(const float* w, float* out, int cout, int cin, int taps, void* stream);
int e4s_polyphase_weights_f32(const float* w, const float* k4, float* out, int cout, int cin, void* stream);
/* the transpose of e4s_polyphase_weights_f32 (weight gradient of an up-sampling StyledConv, model.py:287-300 under autograd):
 * deff [4 phases * 9][Cout][Cin] (gradient of the polyphase kernels) -> dw [Cout][Cin][9] (gradient of the 3x3 weight) */
int e4s_polyphase_fold_f32(const float* deff, const float* k4, float* dw, int cout, int cin, void* stream);

/* ws[g, c, ci] = scale * w[c, ci] * s[g, ci]   (ToRGB: demodulate=False, model.py:417) */
int e4s_rgb_weights_f32(const float* w, const float* s, float* ws, int G, int cin, float scale, void* stream);

/* ---- mask plan (region-select; model.py:386-400 without the 12x redundancy) ---------------- */

/* labels[b, y, x] = argmax_r mask[b, r, y, x]; flags[0] |= 1 if some pixel is not one-hot. */
int e4s_mask_labels(const float* mask, uint8_t* labels, int* flags, int B, int R, int Hm, int Wm, void* stream);

/* Build the row plan for one layer geometry.  Anchor grid [B, Ha, Wa]; an anchor expands into
 * nphase (1 or 4) GEMM rows, one per output phase; the row's region is the label of its OUTPUT
 * pixel (oy = ay*os + py) looked up with legacy-nearest from the [Hm, Wm] label map
 * (F.interpolate(mode='nearest'), model.py:391).  Rows are grouped by (b, region, phase), each
 * group padded to a multiple of BM with -1.
 *   rows  [rows_cap]  anchor ids (b*Ha + ay)*Wa + ax, or -1
 *   tiles [tiles_cap*4] {row_start, group = b*R + region, phase, nvalid}
 *   meta  [4] {ntiles, nrows_padded, 0, 0};  work [3*B*R*nphase] ints of scratch
 * Caps: rows_cap >= B*Ha*Wa*nphase + B*R*nphase*BM, tiles_cap >= rows_cap / BM. */
int e4s_region_plan(const uint8_t* labels, int B, int R, int Hm, int Wm, int Ha, int Wa, int nphase,
                    int BM, int* rows, int* tiles, int* meta, int* work, int rows_cap, int tiles_cap,
                    void* stream);

/* ---- the hot kernel: implicit-GEMM 3x3 / 1x1 convolution on fp32 MFMA ---------------------- */

typedef struct {
    const float* x;          /* input  NHWC [B, Hi, Wi, Cin] */
    const float* w;          /* weights [ncls][ntaps][Cout][Cin]  (Cin contiguous) */
    float* y;                /* output NHWC [B, Ho, Wo, Cout] */
    const int* rows;         /* plan rows or NULL (natural order) */
    const int* tiles;        /* plan tiles or NULL */
    const int* meta;         /* plan meta (device) or NULL */
    int tiles_cap;           /* grid upper bound in plan mode */
    int B, Ha, Wa;           /* anchor grid */
    int Hi, Wi, Ho, Wo, Cin, Cout;
    int istride;             /* input coord = anchor*istride + tap - 1 (3x3) / anchor*istride (1x1) */
    int ostride;             /* output coord = anchor*ostride + phase */
    int ntaps;               /* 9 (3x3) or 1 (1x1); 25 (5x5, pad 2) in e4s_conv_mfma_f32's per-tap gather mode (spatial = 0) */
    int ncls;                /* 1, or 4 = polyphase up-conv (phase-specific weights) */
    const float* in_scale;   /* [G][Cin] style modulation s, or NULL */
    const float* out_scale;  /* [G][Cout] demodulation coefficient (x conv scale), or NULL */
    int groups_per_batch;    /* R: regions per sample (plan mode and labelled spatial mode) */
    const uint8_t* labels;   /* spatial mode: label map [B,Hm,Wm]; the row's group is b*R + label(output pixel),
                                in_scale/out_scale are [B*R][C].  NULL: group = b (tables [B][C]) */
    int Hm, Wm;
    const float* noise;      /* [Bn,1,Ho,Wo], or NHWC [Bn,Ho,Wo,Cout] when noise_per_channel; or NULL */
    const float* noise_w;    /* device scalar (NoiseInjection.weight) */
    int64_t noise_bstride;   /* Ho*Wo (pixels per sample), or 0 when the noise is shared by the batch */
    int noise_per_channel;   /* scripts/face_edit.py:49-52 passes [1,C,H,W] noise */
    const float* bias;       /* [Cout] or NULL */
    const float* slope;      /* [Cout] PReLU slopes (act == 2) */
    int act;                 /* 0 none, 1 leaky-relu(alpha)*gain, 2 PReLU */
    float alpha, gain;
    const float* in_stats;   /* e4s_conv_bf16x3_f32 only: [B][Cin][2] {mean, rstd} (e4s_instnorm_stats_f32): the input is
                                InstanceNorm-ed, (x - mean) * rstd, while it is staged (helpers.py:128-131 folded into the
                                unit's first conv); excludes in_scale.  NULL elsewhere */
    int y_cstride;           /* channels per pixel of the buffer y points into (0 = Cout): the conv may write the first Cout
                                channels of a wider NHWC tensor -- GPEN's StyledConv concatenates its noise (the encoder
                                feature map) behind the conv output (gpen_model.py:343-353).  Plain (unlabelled) kernels */
    float* splitk_ws;        /* scratch of e4s_conv_bf16x3_ws_floats(p) / e4s_conv_mfma_ws_floats(p, spatial) floats (0 -> may be
                                NULL): launches with too few tiles to fill the chip (batch-1 latency runs) split the input
                                channels over several blocks per tile; the raw partial sums land here and are added in a
                                fixed order by a second kernel that also applies the epilogue */
    double* stats_ws;        /* e4s_conv_bf16x3_f32, unlabelled, act == 0, no noise: when set, the epilogue also emits the
                                InstanceNorm partial sums of the OUTPUT it just computed -- per (sample, channel, 256-pixel
                                tile): {sum, sum of squares} in fp64 at stats_ws[((b*Cout + c)*stats_slots + tile)*2] -- so the
                                statistics pass over the conv output (helpers.py:138-139) costs no extra read;
                                e4s_instnorm_finalize_f32 adds the slots in order.  Ignored (must be re-done by the caller)
                                when the launch is split over K: check e4s_conv_bf16x3_ws_floats(p) == 0 */
    int stats_slots;         /* tiles per sample = slots per (b, c) */
    int tap_shift;           /* gather mode (istride 2 / ntaps 1 kernels): input coord = anchor*istride + tap - 1 + tap_shift;
                                1 = the padding-0 stride-2 conv behind a Blur (ConvLayer, model.py:683-700) */
    int split_hint;          /* ABI v14, e4s_conv_mfma_f32 only: 1 = the caller accepts a K split on PLAIN maps of more than 2 pixel tiles with
                                fewer than 256 blocks per sample (the frozen loss networks' 28^2 / 56^2 layers at batch 1-2: one block per CU and
                                every stage an exposed round trip otherwise).  Same products, another order of fp32 additions; 0 (default): such
                                maps are never split -- trained networks keep one summation order whatever the policy of the day */
} e4s_conv_params;

/* y = epilogue( sum_{tap,ci} x[anchor*istride + tap - 1, ci] * in_scale[g,ci] * w[cls,tap,co,ci] )
 * epilogue: v*out_scale[g,co] + noise_w*noise + bias[co], then act.
 * Replaces ModulatedConv2d.forward + region compose + NoiseInjection + FusedLeakyReLU
 * (model.py:276-320, 386-404) and the encoder's Conv2d+PReLU (helpers.py:128-137).
 * Requirements: Cin % 32 == 0, Cout % 32 == 0.  `spatial` selects the halo-tiled loader
 * (natural order, istride == 1, Ha % 8 == 0, Wa % 16 == 0). */
int e4s_conv_mfma_f32(const e4s_conv_params* p, int spatial, void* stream);
/* floats of p->splitk_ws this launch may use (0: none).  Maps of <= 2 pixel tiles per sample (<= 16x16: 14x14 / 7x7 layers,
 * the 4^2-16^2 generator layers) split their input-channel chunks up to 8 ways over blockIdx.y; partial sums are added in a
 * fixed order by the second stage that applies the epilogue.  The policy looks at one sample's geometry only, so results do
 * not depend on the batch.  With splitk_ws == NULL the launch never splits.  (ABI v14: p->split_hint widens the policy to plain maps
 * with < 256 blocks per sample.) */
int64_t e4s_conv_mfma_ws_floats(const e4s_conv_params* p, int spatial);

/* Exact up-sampling StyledConv: conv_transpose2d(stride 2) + 4x4 blur (model.py:287-300) with the transposed conv's
 * minimal 9*Cin*Cout MACs per input pixel on the matrix cores and the blur applied from an LDS-resident intermediate
 * (the polyphase ncls = 4 form of e4s_conv_mfma_f32 spends 36).  Same params struct: x NHWC [B,Hi,Wi,Cin], w = plain
 * tap-packed 3x3 weights [9][Cout][Cin] (e4s_pack_taps_f32), y NHWC [B,2Hi,2Wi,Cout], in_scale/out_scale/labels/noise/
 * bias/act as for the spatial mode; k4 = the module's 4x4 blur kernel.  Cin % 32 == 0, Cout % 32 == 0. */
int e4s_upconv_mfma_f32(const e4s_conv_params* p, const float* k4, void* stream);
int e4s_upconv_blocks_per_cu(void);    /* diagnostic: occupancy of that kernel as the runtime computes it */

/* Split-bf16 ("bf16x3") variant of the natural-order 3x3 stride-1 contraction: every fp32 operand v = hi + lo (two bf16),
 * product = a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with an fp32 accumulator (relative error per
 * product <= ~2^-16; 5.3x the fp32-MFMA rate).  Same params struct and epilogue as e4s_conv_mfma_f32(spatial = 1) with:
 * ntaps = 9, istride = 1, noise_per_channel = 0, Cin % 32 == 0, Cout % 32 == 0, and p->w pointing at the SPLIT weights
 * produced by e4s_split_bf16x2_f32 from the tap-packed fp32 weights ([9][Cout][Cin], or the polyphase [4][9][Cout][Cin]
 * with ncls = 4, ostride = 2).  Variants:
 *   labels == NULL  one style per sample at most (in_scale/out_scale [B][C]) or in_stats (fused InstanceNorm): the
 *                   encoder's Conv2d(+PReLU) (helpers.py:128-137) and the unmasked StyledConvs incl. the 512^2 / 1024^2
 *                   up-convs (model.py:655-657); column tile 128 / 64 / 32 by Cout
 *   labels != NULL  region-select (per-pixel style on the A fragment): masked StyledConvs (model.py:386-400), plain or
 *                   polyphase; Cout % 128 == 0, in_scale required, act != 2 */
int e4s_conv_bf16x3_f32(const e4s_conv_params* p, void* stream);
int64_t e4s_conv_bf16x3_ws_floats(const e4s_conv_params* p);
/* w fp32 [rows][cin] -> out [rows][cin/32][32 hi bf16 | 32 lo bf16] (same byte size), cin % 32 == 0 */
int e4s_split_bf16x2_f32(const float* w, void* out, int64_t rows, int cin, void* stream);

/* Masked StyledConv (model.py:386-400; plain or polyphase up-conv), "variant rows" kernel (csrc/conv_region.hip): the halo of a
 * 16x16-pixel tile is staged once per 16 input channels, scaled with the style of each halo pixel's OWN region and split to hi|lo
 * bf16; the (pixel, tap) pairs that cross a region boundary read extra LDS rows scaled with the reading pixel's style, so the
 * matrix-core loop does no per-fragment arithmetic (e4s_conv_bf16x3_f32's region-select kernel scales and splits every A
 * fragment).  Same results contract (split-bf16 products, fp32 accumulation, fixed summation order per output).
 * p as e4s_conv_bf16x3_f32 with labels != NULL (p->w = the e4s_split_bf16x2_f32 image: tiles that need more than 256 variant
 * rows, and launches of <= 128 tiles that split K, run on the region-select kernel); w16 = the e4s_split16_bf16x2_f32 image of
 * the same tap-packed weights; p->splitk_ws = e4s_conv_region_ws_floats(p) floats of scratch (tile flags or split-K slabs).
 * e4s_split16_bf16x2_f32: w fp32 [rows][Cout][Cin] (rows = ncls * 9) -> out [rows][Cin/16][Cout][16 hi bf16 | 16 lo bf16]
 * (the byte size of w), Cin % 16 == 0 -- and, ABI v12, for Cout % 32 == 0 behind it the same values fragment-major ([rows][Cin/16][Cout/32]
 * [hi: 64 lanes x 16 B | lo]), which the one-wave-per-SIMD kernel loads straight into its B fragments: `out` holds e4s_split16_bytes(rows,
 * Cout, Cin) bytes (twice the size of w then).
 * Which kernel a launch takes (host-side policy, no launch; ABI v12): e4s_conv_region_path(p) = 0 not covered, 1 the 8-wave kernel
 * (256 pixels x 128 channels per block, two waves per SIMD; with a K split for launches of <= 128 tiles), 2 the one-wave-per-SIMD
 * kernel (csrc/conv_region1w.hip: 256 x 256 tiles, 16 accumulator tiles per wave; Cout % 256 == 0 and no K split). */
int e4s_conv_region_bf16x3_f32(const e4s_conv_params* p, const void* w16, void* stream);
int64_t e4s_conv_region_ws_floats(const e4s_conv_params* p);
int e4s_conv_region_path(const e4s_conv_params* p);
int e4s_split16_bf16x2_f32(const float* w, void* out, int64_t rows, int cout, int cin, void* stream);
int64_t e4s_split16_bytes(int64_t rows, int cout, int cin);

/* ---- backward of the fused generator (configs 3 and 5) -------------------- */
typedef struct {
    const float* gz;         /* dL/d(out_pre), NHWC [B, Hy, Wy, Cy]  (Cy = forward Cout) */
    const float* wt;         /* e4s_pack_taps_bwd_f32 layout [ncls][9][Cx][Cy] */
    float* dx;               /* out: NHWC [B, Hx, Wx, Cx]  (Cx = forward Cin) */
    const float* x;          /* forward input, NHWC [B, Hx, Wx, Cx] (needed for ds) */
    float* ds;               /* out (overwritten): [G][Cx] grad w.r.t. the modulation s; or NULL */
    const float* s;          /* [G][Cx] forward modulation, or NULL (= 1) */
    const float* d;          /* [G][Cy] forward demodulation coefficient (x conv scale), or NULL (= 1) */
    const uint8_t* labels;   /* [B,Hm,Wm] label map; group = b*R + label(output pixel); NULL: group = b */
    int Hm, Wm, R;
    int B, Hx, Wx, Cx, Hy, Wy, Cy;
    int ncls;                /* 1: 3x3 stride-1 conv; 4: polyphase up-conv (Hy = 2*Hx) */
    float* ds_ws;            /* scratch of e4s_conv_bwd_ws_floats(p) floats (may be 0 -> NULL): ds partial sums, one slot per
                                (tile of the sample, tap-group split, group, channel), and -- at low resolutions, where the
                                tap groups are split over several blocks per tile -- the partial dx; both are added in a
                                fixed order after the MFMA pass: dx and ds are bit-reproducible */
} e4s_conv_bwd_params;

/* dx[q,ci] = sum_{tap} s[r(p),ci] * sum_co wt[tap,ci,co] * d[r(p),co] * gz[p,co]   (p = pixel fed by q through tap)
 * ds[g,ci] += sum_{q,tap: r(p)=g} x[q,ci] * (sum_co wt*d*gz)      -- one fp32-MFMA pass, Cx % 64 == 0, Cy % 32 == 0 */
int e4s_conv_bwd_mfma_f32(const e4s_conv_bwd_params* p, void* stream);
int64_t e4s_conv_bwd_ws_floats(const e4s_conv_bwd_params* p);
/* Weight gradient of the 3x3 / 1x1 convs on fp32 MFMA (config 5): the contraction over the PIXELS of the operands the
 * forward contracts over (Cin, taps):  dw[tap][co][ci] = sum_a (gz[o(a)][co] * d[g(a)][co]) * (x[a*istride + tap - 1][ci]
 * * s[g(a)][ci]),  o(a) = a*ostride + (py, px), g(a) = b*R + label(o(a)) (b when labels == NULL); 1x1: x[a*istride]. */
typedef struct {
    const float* gz;         /* dL/d(out_pre), NHWC [B, Ho, Wo, Cout] */
    const float* x;          /* forward input, NHWC [B, Hi, Wi, Cin] */
    float* dw;               /* out (overwritten): [ntaps][Cout][Cin] -- the tap-packed layout of e4s_pack_taps_f32 */
    float* ws;               /* scratch: e4s_conv_wgrad_ws_floats(p) floats (split-K slabs, added in a fixed order) */
    const float* s;          /* [G][Cin] forward modulation or NULL */
    const float* d;          /* [G][Cout] forward demodulation coefficient or NULL */
    const uint8_t* labels;   /* [B,Hm,Wm] or NULL */
    int Hm, Wm, R;
    int B, Hi, Wi, Cin, Ha, Wa, Ho, Wo, Cout;
    int istride;             /* 1 or 2 */
    int ostride, py, px;     /* 1,0,0; or 2 and the phase of a polyphase up-conv (one call per phase) */
    int ntaps;               /* 9 or 1 */
    int tap_shift;           /* 0; 1 = padding-0 3x3 conv (input coord = a*istride + tap), as e4s_conv_params.tap_shift */
} e4s_conv_wgrad_params;
int e4s_conv_wgrad_f32(const e4s_conv_wgrad_params* p, void* stream);
int64_t e4s_conv_wgrad_ws_floats(const e4s_conv_wgrad_params* p);
/* Which kernel e4s_conv_wgrad_f32 launches (host-side policy, no launch; ABI v13): 1 = the split-bf16 kernel (3x3; istride 1 with or without a
 * region map, istride 2 without: both operands as hi + lo bf16, three bf16 MFMAs per product, fp32 accumulate -- the forward kernels'
 * arithmetic, 2^-17-class rounding); 0 = the exact-fp32 MFMA kernel (1x1; everything with env E4S_WGRAD_BF16X3=0). */
int e4s_conv_wgrad_path(const e4s_conv_wgrad_params* p);
/* forward-packed weights [ncls][9][Cout][Cin] -> backward layout [ncls][9][Cin][Cout], taps flipped */
int e4s_pack_taps_bwd_f32(const float* w, float* wt, int ncls, int cout, int cin, void* stream);
/* FusedLeakyReLU backward and dL/dd of a StyledConv in one pass over dy and y (both NHWC [B,H,W,C]): gz = dy * lrelu'(y) * gain (what
 * e4s_fused_bias_act_f32(act 3, grad 1) writes) and dd as e4s_demod_grad_f32 computes it from that gz -- without re-reading gz and y.
 * C a multiple of 4 in [4, 1024] with 256 % (C/4) == 0 (else hipErrorInvalidValue); labels >= R are counted in region R - 1; ws: e4s_reduce_parts_ws_floats(e4s_act_bwd_demod_nsplit(B, H, W, C), B * R * C) floats */
int e4s_act_bwd_demod_nsplit(int B, int H, int W, int C);
int e4s_act_bwd_demod_f32(const float* dy, const float* y, float* gz, const float* noise, const float* noise_w, int64_t noise_bstride,
                          const float* bias, float alpha, float gain, const uint8_t* labels, int Hm, int Wm, int R, float* dd, float* ws,
                          int B, int H, int W, int C, void* stream);
/* dd[b,r,co] = sum_{p in r} gz[p,co] * (lrelu^-1(y[p,co])/gain - noise_w*noise[p] - bias[co])   (= d * dL/dd; the
 * caller divides by d); dd is overwritten */
int e4s_demod_grad_f32(const float* gz, const float* y, const float* noise, const float* noise_w,
                       int64_t noise_bstride, const float* bias, float alpha, float gain,
                       const uint8_t* labels, int Hm, int Wm, int R, float* dd, float* ws, int B, int H, int W, int C,
                       void* stream);
/* Input gradient of the MASKED StyledConvs in scatter form (csrc/dgrad_scatter.hip; autograd of model.py:386-400): the contraction
 * G[m, t, ci] = sum_co u[m, co] W[t][co][ci] is a plain 1x1 e4s_conv_bf16x3_f32 launch (rows = source pixels, 9 Cx columns); these two
 * passes hold everything that depends on the region map:
 *   e4s_region_scale_f32: u = gz * d[region of the output pixel]; gz NHWC [B, H*os, W*os, C] (os = 2 when ncls == 4), d [B*R, C];
 *     ncls == 1: u like gz; ncls == 4 (polyphase up-conv): u [4 phases][B, H, W, C], one contiguous map per output phase.
 *   e4s_col2im_region_f32: G [ncls][B, H, W, 9, C] (tap-major columns), x NHWC [B, H, W, C] (the layer's input), s [B*R, C] ->
 *     dx [B, H, W, C] = sum_ph sum_t s[region(m_t, ph)] * G_ph[m_t, t] (m_t = h - (t - 1)),
 *     ds [B*R, C]    = sum over rows (m, ph) of region rho of sum_t x[m + t - 1] * G_ph[m, t]   (ordered sums: bit-reproducible).
 *   C a multiple of 4 in [4, 1024] with 256 % (C/4) == 0, R <= 16; ws: e4s_reduce_parts_ws_floats(e4s_col2im_region_nsplit(B,H,W,C), B*R*C). */
int e4s_region_scale_f32(const float* gz, const float* d, const uint8_t* labels, int Hm, int Wm, int R, float* u, int B, int H, int W, int C,
                         int ncls, void* stream);
int e4s_col2im_region_nsplit(int B, int H, int W, int C);
int e4s_col2im_region_f32(const float* G, const float* x, const float* s, const uint8_t* labels, int Hm, int Wm, int R, float* dx, float* ds,
                          float* ws, int B, int H, int W, int C, int ncls, void* stream);

/* pixel splits of the two segmented reductions here: their `ws` scratch holds e4s_reduce_parts_ws_floats(nsplit, output
 * elements) floats, one slot per split, added in order (bit-reproducible; no floating-point atomics) */
int e4s_seg_reduce_nsplit(int B, int H, int W, int C);
/* ToRGB backward: dws[g,c,ci] = sum_{p in g} drgb[b,c,p]*x[p,ci] (overwritten); dx[p,ci] (+)= sum_c drgb*ws[g(p),c,ci] */
int e4s_torgb_bwd_w_f32(const float* drgb, const float* x, const uint8_t* labels, int Hm, int Wm, int R,
                        float* dws, float* ws, int B, int H, int W, int C, void* stream);
int e4s_torgb_bwd_x_f32(const float* drgb, const float* ws, const uint8_t* labels, int Hm, int Wm, int R,
                        float* dx, int B, int H, int W, int C, int accumulate, void* stream);

/* Weight-gradient operand builder (config 5): out[b,a,c] = tab[(b*R + label(out pixel a*os+(py,px)))*C + c] *
 * in[b, a*istride + (dy,dx), c], zero outside the image; anchors a on [Ha,Wa].  The [C1 x P] x [P x C2] contraction over
 * the pixels is a plain GEMM (BLAS).  tab/labels may be NULL (no scaling / group = b). */
int e4s_shift_scale_f32(const float* in, const float* tab, const uint8_t* labels, int Hm, int Wm, int R,
                        float* out, int B, int Ha, int Wa, int Hi, int Wi, int C, int istride, int dy, int dx,
                        int os, int py, int px, void* stream);

/* ---- ToRGB (model.py:422-448) -------------------------------------------------------------- */
/* out[b,c,y,x] = sum_ci x[b,y,x,ci]*ws[g,c,ci] + bias[c] + upfirdn2d(skip, k4, up=2, pad=(2,1))
 * x NHWC, out/skip NCHW; labels ([B,Hm,Wm] label map) NULL => group = b (unmasked). */
int e4s_torgb_f32(const float* x, const float* ws, const float* bias, const float* skip, const float* k4,
                  const uint8_t* labels, int Hm, int Wm, int R, float* out,
                  int B, int H, int W, int Cin, void* stream);

/* Soft-mask fallback (reference formulation, model.py:391-398): out (+)= y * nearest(mask)[:, r]; mask [B,R,Hm,Wm];
 * y/out NHWC [B,H,W,C] (channels_last = 1) or NCHW [B,C,H,W] (0). */
int e4s_mask_mul_add_f32(const float* y, const float* mask, float* out, int r, int B, int H, int W, int C,
                         int R, int Hm, int Wm, int channels_last, int accumulate, void* stream);

/* y = lrelu(x + noise_w*noise[b,p] + bias[c]) * gain on NHWC x [B,HW,C] (NoiseInjection + FusedLeakyReLU as a
 * separate pass; only the soft-mask fallback needs it, the fused path does it in the conv epilogue). */
int e4s_noise_bias_act_nhwc_f32(const float* x, const float* noise, const float* noise_w, int64_t noise_bstride,
                                const float* bias, float* y, int B, int HW, int C, float alpha, float gain,
                                void* stream);

/* ---- backward of the regional style encoder (config 5: joint train step, src/training/coach.py:340-356) ------------- */
/* InstanceNorm2d backward (+ the SE gate that multiplied its output): sums[b,c] = {A = sum_p dy, Bq = sum_p dy*xhat}
 * (Bq is also dL/d(gate)), dx (+)= rstd * gate * (dy - A/N - xhat*Bq/N); x NHWC [B,HW,C], stats from
 * e4s_instnorm_stats_f32, gate [B,C] or NULL; ws: e4s_instnorm_bwd_ws_doubles(B,HW,C) doubles.  Ordered reductions. */
int e4s_instnorm_bwd_f32(const float* dy, const float* x, const float* stats, const float* gate, float* sums, float* dx,
                         double* ws, int B, int HW, int C, int accumulate, void* stream);
int64_t e4s_instnorm_bwd_ws_doubles(int B, int HW, int C);
/* PReLU(C) on NHWC [npix, C]: y = u > 0 ? u : slope[c]*u;  backward: du = dy * (u > 0 ? 1 : slope[c]),
 * dslope[c] = sum_p dy*u*[u <= 0] (ordered; ws: e4s_prelu_bwd_ws_floats(npix, C) floats) */
int e4s_prelu_f32(const float* u, const float* slope, float* y, int64_t npix, int C, void* stream);
int e4s_prelu_bwd_f32(const float* dy, const float* u, const float* slope, float* du, float* dslope, float* ws,
                      int64_t npix, int C, void* stream);
int64_t e4s_prelu_bwd_ws_floats(int64_t npix, int C);
/* in NHWC [B,2H,2W,C] -> out NHWC [B,H,W,4C], out[b,a,c,(py*2+px)*C + ch] = in[b,2a+py,2c+px,ch] (pixel unshuffle): lays the four output
 * phases of an up-sampling conv side by side in the channel dimension of its input grid, so that the dgrad of the polyphase form
 * (model.py:287-300 backward) is ONE 3x3 convolution with 4*Cout input channels on e4s_conv_bf16x3_f32.  C % 4 == 0 */
int e4s_pixel_unshuffle2_f32(const float* in, float* out, int B, int H, int W, int C, void* stream);
/* out[b, y*s, x*s, :] (+)= in[b, y, x, :], in NHWC [B,H,W,C], out NHWC [B,H*s,W*s,C]; without accumulate the other
 * positions are zero-filled (zero insertion: the stride-2 conv's dgrad; with accumulate: MaxPool2d(1, s)'s backward) */
int e4s_strided_scatter_f32(const float* in, float* out, int B, int H, int W, int C, int s, int accumulate, void* stream);
/* out[b, y*s + oy, x*s + ox, :] = in[b, y, x, :] on an output grid NHWC [B,Ho,Wo,C], zero elsewhere: the dgrad operand of a
 * stride-s padding-0 conv (Discriminator ConvLayers, model.py:683-700) */
int e4s_strided_place_f32(const float* in, float* out, int B, int H, int W, int C, int s, int oy, int ox, int Ho, int Wo,
                          void* stream);
/* backward of e4s_region_mean_f32: dfeat[b,p,c] (+)= dcodes[b, label(p), off + c] / count[b, label(p)];
 * counts: int scratch [B*R] */
int e4s_region_mean_bwd_f32(const float* dcodes, const uint8_t* labels, int Hm, int Wm, int* counts, float* dfeat, int B,
                            int H, int W, int C, int R, int stride, int off, int accumulate, void* stream);

/* ---- GPEN FullGenerator / Discriminator support ---------------------------------- */
/* 1x1 conv with tiny Cin (<= 4; the 3 -> C stem ConvLayer, gpen_model.py:658, model.py:756): x NCHW [B,Cin,HW],
 * w [Cout,Cin], y NHWC (channel stride y_cstride or Cout): act(x.w*scale + bias), act 1 = leaky-relu(alpha)*gain */
int e4s_conv1x1_small_f32(const float* x, const float* w, const float* bias, float* y, int B, int HW, int Cin, int Cout,
                          int y_cstride, float scale, int act, float alpha, float gain, void* stream);
/* y[pix, coff + c] = lrelu(noise_w[0]*feat[pix, c] + bias[c], alpha)*gain: the noise half of GPEN's concatenating
 * StyledConv (gpen_model.py:343-353); feat NHWC [npix, C], y NHWC with y_cstride channels per pixel */
int e4s_noise_half_f32(const float* feat, const float* noise_w, const float* bias, float* y, int64_t npix, int C,
                       int y_cstride, int coff, float alpha, float gain, void* stream);
/* PixelNorm over dim 1 of x [B,D] */
int e4s_pixelnorm_f32(const float* x, float* y, int B, int D, void* stream);
/* out = (a + b) * scale over n floats (ResBlock combine, model.py:733-737) */
int e4s_add_scale_f32(const float* a, const float* b, float* out, float scale, int64_t n, void* stream);
/* minibatch-stddev feature (model.py:783-790, stddev_feat = 1): x NHWC [B,HW,C] -> y NHWC [B,HW,Cy]: x, then the group
 * statistic in channel C, zeros up to Cy (Cy > C; pads 513 -> 544 channels for the 32-channel K step) */
int e4s_minibatch_stddev_f32(const float* x, float* y, int B, int HW, int C, int Cy, int group, void* stream);

/* ---- optimisation / training step around the generator backward ------------------------- */
/* out[b,r,k] = base + mul * scale * gate * sum_o g[b,r,o] * w[r,o,k]   -- "x @ W": the transposed contraction of the
 * LocalMLP backward (networks.py:15-39) and of the style prologue's chain rule; w [R][O][K] streamed once.
 * base/mul/ref [B,R,K] optional; gate = ref > 0 ? 1 : alpha (leaky-ReLU derivative on the saved activation).
 * B <= 16; ws: e4s_grouped_linear_t_ws_floats(B,R,O,K) floats (split partial sums, added in order). */
int e4s_grouped_linear_t_f32(const float* g, const float* w, float* out, float* ws, int B, int R, int O, int K,
                             float scale, const float* base, const float* mul, const float* ref, float alpha,
                             void* stream);
int64_t e4s_grouped_linear_t_ws_floats(int B, int R, int O, int K);
/* out[i] = scale * sum_{p < nparts} parts[p*n + i] in a fixed order (second stage of every split reduction): 64-part
 * chunks first, then the chunk sums.  `parts` must have room for e4s_reduce_parts_ws_floats(nparts, n) floats (the parts
 * plus the chunk sums behind them). */
int e4s_reduce_parts_f32(float* parts, float* out, int nparts, int64_t n, float scale, void* stream);
int64_t e4s_reduce_parts_ws_floats(int nparts, int64_t n);
/* dw[r,o,k] = scale * sum_b g[b,r,o] * h[b,r,k] (LocalMLP weight gradients);  out[i] = sum_b x[b*n + i] (bias gradients) */
int e4s_grouped_outer_f32(const float* g, const float* h, float* dw, int B, int R, int O, int K, float scale, void* stream);
/* Style-gradient tail of the generator backward (autograd of model.py:242-320 / 422-448 w.r.t. the style, all layers in two launches;
 * replaces per layer: dd3 = dd_d * d^2, ds = ds_raw - s * (dd3 @ Wsq), dstyle = mod_scale * ds @ Wmod, dlat[:, :, slot] += dstyle).
 * StyledConv job: ds_raw [G,Cin] (dL/ds of the contraction), dd_d [G,Cout] (d * dL/dd), d [G,Cout], s [G,Cin], wsq [Cout,Cin], dws = NULL.
 * ToRGB job: dws [G,3,Cin] (dL/d(ws)), w3 [3,Cin], conv_scale (ws = conv_scale * w3 * s); the conv pointers unused.
 * Both: wmod [Cin,S] (the modulation EqualLinear's weight), mod_scale, ds_total [G,Cin] (OUT: dL/ds), slot (latent index),
 * masked (1: G = B*R, row g = b*R + r feeds dlat[b][r][slot]; 0: G = B, row b feeds dlat[b][0][slot]).  Jobs of one slot are added in
 * job order; every element of dlat [B][R][NL][S] is written.  Ordered sums only: bit-reproducible. */
#define E4S_STYLE_GRAD_MAX_JOBS 32    /* 32 jobs by value = 3.4 KB of kernel arguments (limit 4 KB); a 1024^2 generator has 26 */
typedef struct e4s_style_grad_job {
    const float* ds_raw; const float* dd_d; const float* d; const float* s; const float* wsq;
    const float* dws; const float* w3;
    const float* wmod; float* ds_total;
    float conv_scale, mod_scale;
    int G, Cin, Cout, slot, masked;
} e4s_style_grad_job;
int e4s_style_grad_multi_f32(const e4s_style_grad_job* jobs, int njobs, float* dlat, int B, int R, int NL, int S, void* stream);
int e4s_batch_sum_f32(const float* x, float* out, int B, int64_t n, void* stream);
/* out[c] = sum_r x[r*C + c] over a tall [rows][C] matrix (C % 4 == 0, C <= 1024): bias gradients of channels-last activations
 * (rows = B*H*W); two ordered levels, bit-reproducible; ws: e4s_colsum_ws_floats(rows, C) floats */
int e4s_colsum_f32(const float* x, float* out, float* ws, int64_t rows, int C, void* stream);
int64_t e4s_colsum_ws_floats(int64_t rows, int C);
/* Tail of an unmasked StyledConv's input gradient when the contraction ran on a forward kernel (model.py:276-320 backward):
 * u NHWC [B,hw,C] = conv(gz * d, flipped W^T) without the style; in one pass u <- u * s[b] (= dL/dx, in place) and
 * ds[b][c] = sum_p x[b,p,c] * u[b,p,c] (dL/ds through the modulated weight; ordered two-level sum, bit-reproducible).
 * C % 4 == 0, C <= 1024; ws: e4s_scale_dot_ws_floats(B, hw, C) floats */
int e4s_scale_dot_f32(float* u, const float* x, const float* s, float* ds, float* ws, int B, int64_t hw, int C, void* stream);
int64_t e4s_scale_dot_ws_floats(int B, int64_t hw, int C);
/* torch.optim.Adam's update (no amsgrad) fused into one pass over p/grad/m/v [n]; `step` >= 1 is the step being taken;
 * bias corrections are evaluated in double on the host as torch does */
int e4s_adam_step_f32(float* p, const float* grad, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                      double eps, double weight_decay, int step, void* stream);
/* the same update with the step count in DEVICE memory (int64, >= 1 = the step being taken): nothing step-dependent is computed on the
 * host, so the launch can be captured in a HIP graph and replayed.  lr_dev: optional DEVICE double that overrides `lr` (a learning-rate
 * schedule -- coach.py:377-381 -- then reaches captured launches).  The caller advances the counts first (e4s_advance_i64: one launch
 * for all the counts of a parameter group, kept in one flat tensor) */
int e4s_adam_step_dev_f32(float* p, const float* grad, float* m, float* v, int64_t n, double lr, const double* lr_dev, double beta1,
                          double beta2, double eps, double weight_decay, const int64_t* step, void* stream);
int e4s_advance_i64(int64_t* steps, int64_t n, void* stream);
/* multi-tensor forms: `count` tensors given as HOST arrays of device pointers / element counts.  The library hands them to the kernels
 * BY VALUE, 48 tensors per launch (no device-side table, no host-to-device copy: capturable): Net3's 344 parameter tensors are 8
 * launches instead of 344.  Arithmetic per element identical to the single-tensor entry points (bit for bit). */
int e4s_adam_multi_dev_f32(int count, float* const* p, const float* const* grad, float* const* m, float* const* v, const int64_t* n,
                           const int64_t* const* step, double lr, const double* lr_dev, double beta1, double beta2, double eps,
                           double weight_decay, void* stream);
int e4s_ema_multi_f32(int count, float* const* dst, const float* const* src, const int64_t* n, double decay, void* stream);
/* 3x3 stride-1 conv as Winograd F(2,3) along the image rows on the split-bf16 matrix-core path (csrc/conv_wino.hip): 1.5x fewer MFMAs than
 * e4s_conv_bf16x3_f32 for the encoder's Conv2d(3x3, stride 1) layers (src/models/encoders/helpers.py:128-137).
 * e4s_wino_weights_f32: w9 [9][Cout][Cin] (tap-packed, e4s_pack_taps_f32) -> the kernel's transformed, hi/lo-split operand
 * (e4s_wino_weights_bytes(Cout, Cin) bytes, opaque: the plane-major image [3 ky][Cin/16][4 positions][Cout][16 hi | 16 lo] and, for Cout % 32 == 0,
 * behind it the same values fragment-major, which the one-wave-per-SIMD kernel of csrc/conv_wino1w.hip loads straight into its B fragments).
 * e4s_conv_wino_bf16x3_f32: p->x / p->y NHWC, p->w = that operand; covered: istride = ostride = 1, ntaps 9, ncls 1, H % 16 == W % 16 == 0,
 * Cin % 16 == 0 (>= 32), Cout % 128 == 0, no styles / labels / noise / plan / y_cstride; launches of <= 128 tiles split the input channels over blocks (p->splitk_ws:
 * e4s_conv_wino_ws_floats floats; stats_ws is ignored then); honoured: in_stats (InstanceNorm folded
 * into the input transform), bias, act (0 none, 1 leaky * gain, 2 PReLU with p->slope), stats_ws / stats_slots (per-tile {sum, sum^2} of the
 * output as e4s_conv_bf16x3_f32 emits them).  e4s_conv_wino_covers: 1 if the launch would be accepted.  Everything else: hipErrorInvalidValue. */
int64_t e4s_wino_weights_bytes(int Cout, int Cin);
int e4s_wino_weights_f32(const float* w9, void* out, int Cout, int Cin, void* stream);
int e4s_conv_wino_covers(const e4s_conv_params* p);
int64_t e4s_conv_wino_ws_floats(const e4s_conv_params* p);   /* split-K slabs for p->splitk_ws (launches of <= 128 tiles); 0: none */
int e4s_conv_wino_bf16x3_f32(const e4s_conv_params* p, void* stream);
/* Exact up-sampling StyledConv on the split-bf16 matrix-core path (csrc/upconv_bf16x3.hip): conv_transpose2d(stride 2) +
 * Blur (src/models/stylegan2/model.py:287-300, 206-213) + NoiseInjection + FusedLeakyReLU (:396-404) in one kernel, one style
 * per sample (unmasked layers, model.py:655-657); Cin % 32 == 0, Cout % 32 == 0.
 * e4s_subpixel_weights_f32: w [Cout,Cin,3,3] -> the kernel's packed, hi/lo-split operand (9 * Cout * Cin * 4 bytes, opaque).
 * p: x / y NHWC, w = that buffer, in_scale [B,Cin] | NULL, out_scale [B,Cout] | NULL, noise / bias / act / alpha / gain as
 * e4s_conv_bf16x3_f32; k4: DEVICE pointer to the 4x4 blur taps. */
int e4s_subpixel_weights_f32(const float* w, void* out, int Cout, int Cin, void* stream);
int e4s_upconv_bf16x3_f32(const e4s_conv_params* p, const float* k4, void* stream);

/* 3x3 stride-1 conv with Cin == 32 on the split-bf16 path, weights resident in LDS, halos fetched two tiles ahead
 * (csrc/conv_c32.hip): the generator's 32 -> 32 StyledConv at 1024^2 (model.py:537-549).  p as e4s_conv_bf16x3_f32 (no
 * labels / stats / split-K); w = split image of the tap-packed weights.  rgb_ws [B][3][32] and rgb_partial [B][3][H][W], both
 * or neither (Cout == 32): the ToRGB 1x1 modulated conv (model.py:422-440) of the layer's output leaves the same pass;
 * e4s_torgb_finish_f32 then adds bias + the FIR-upsampled skip (model.py:441-446) -> out NCHW [B,3,H,W]. */
int e4s_conv_c32_bf16x3_f32(const e4s_conv_params* p, const float* rgb_ws, float* rgb_partial, void* stream);
int e4s_torgb_finish_f32(const float* partial, const float* bias, const float* skip, const float* k4, float* out, int B, int H,
                         int W, void* stream);

/* ---- stitching the swapped face back onto the target on the device (csrc/stitch.hip) ----
 * scripts/face_swap.py:81-97 smooth_face_boundry = e4s_erode_u8 (cv2.erode, flat (2r+1)^2, BORDER_CONSTANT) ->
 * e4s_gaussian_blur_u8 (cv2.GaussianBlur on CV_8U: 8.8 fixed-point taps summing to 256, exact integer passes, one rounding;
 * BORDER_REFLECT_101) -> e4s_alpha_composite_u8 (PIL Image.alpha_composite onto an opaque target, RGB of the result).
 * e4s_mask_to_u8: 255 * uint8(bilinear resize of a [B,Hm,Wm] fp32 mask) (face_swap.py:291-294).
 * src/utils/multi_band_blending.py:4-75: e4s_pyrdown_{u8,f32} / e4s_pyrup_f32 (cv2.pyrDown / pyrUp, HWC images),
 * e4s_lap_level_f32 (one pyramid level of the blend, optionally + the up-sampled reconstruction), e4s_clip_u8.
 * All images HWC; masks and alpha [B,H,W]. */
int e4s_mask_to_u8(const float* mask, uint8_t* out, int B, int H, int W, int Hm, int Wm, void* stream);
int e4s_erode_u8(const uint8_t* src, uint8_t* dst, int B, int H, int W, int radius, int border_value, void* stream);
int e4s_gaussian_blur_u8(const uint8_t* src, uint8_t* dst, int B, int H, int W, int ksize, const int* taps_fixed8, void* stream);
int e4s_alpha_composite_u8(const uint8_t* face, const uint8_t* target, const uint8_t* alpha, uint8_t* out, int B, int H, int W,
                           void* stream);
int e4s_pyrdown_u8(const uint8_t* src, uint8_t* dst, int B, int H, int W, int C, void* stream);
int e4s_pyrdown_f32(const float* src, float* dst, int B, int H, int W, int C, void* stream);
int e4s_pyrup_f32(const float* src, float* dst, int B, int h, int w, int C, void* stream);
int e4s_u8_to_f32(const uint8_t* src, float* dst, int64_t n, void* stream);
int e4s_lap_level_f32(const void* a, const float* ua, const void* b, const float* ub, const float* m, const float* acc,
                      float* out, int64_t n, int a_is_u8, void* stream);
int e4s_clip_u8(const float* src, uint8_t* dst, int64_t n, void* stream);

/* EMA of the weights, torch_utils.accumulate (src/utils/torch_utils.py:189-194; coach.py:396-398):
 * dst[i] = dst[i] * decay + src[i] * (1 - decay), one launch per tensor */
int e4s_ema_f32(float* dst, const float* src, int64_t n, double decay, void* stream);

/* ---- device pre/post-processing of the face-swap pipeline ------------------------------- */
/* labelMap2OneHot (src/utils/torch_utils.py:166-172): labels u8 [B,H,W] -> one-hot fp32 [B,R,H,W] */
int e4s_onehot_u8_f32(const uint8_t* labels, float* out, int B, int R, int H, int W, void* stream);
/* swap_head_mask_revisit_considerGlass(source, target, hair_first=True) (src/utils/swap_face_mask.py:33-82) on n label
 * pixels of the 12-class maps: out = swapped label, hole = 255 where a hole was filled with skin, else 0 */
int e4s_swap_head_mask_u8(const uint8_t* src, const uint8_t* tgt, uint8_t* out, uint8_t* hole, int64_t n, void* stream);
/* fg = not(label in {0, 11, 4}) or hole == 255   (scripts/face_swap.py:280-284) */
int e4s_foreground_mask_f32(const uint8_t* labels, const uint8_t* hole, float* fg, int64_t n, void* stream);
/* Flat (2*radius+1)^2 structuring element, geodesic border (src/utils/morphology.py:23-198 with kernel = ones):
 * dil = window max, ero = window min of x [N,H,W]; either output may be NULL; radius <= 16 */
int e4s_morph_f32(const float* x, float* dil, float* ero, int N, int H, int W, int radius, void* stream);
/* create_masks (scripts/face_swap.py:30-48): operation 0 'dilation', 1 'erosion', 2 'expansion'; mask/border/full
 * [N,H,W]; ws = 2*N*H*W floats of scratch */
int e4s_create_masks_f32(const float* mask, float* border, float* full, float* ws, int N, int H, int W, int radius,
                         int operation, void* stream);
/* tensor2im (src/utils/torch_utils.py:63-69): NCHW fp32 [B,3,H,W] in [-1,1] -> HWC uint8 [B,H,W,3] (4x fewer bytes to
 * all-gather than the fp32 image) */
int e4s_tensor2im_u8(const float* img, uint8_t* out, int B, int H, int W, void* stream);
/* copy `bytes` (multiple of 16) with exactly `blocks` persistent 512-thread workgroups: the measurement stand-in for the RCCL copy kernels
 * that land the peers' all-gather shards on a rank (bench.py `gather_contention`; ABI v12) */
int e4s_stream_copy_u8(const void* src, void* dst, int64_t bytes, int blocks, void* stream);
/* out = uint8(face * m + target * (1 - m)), m = bilinear(align_corners=False) resize of mask [B,Hm,Wm] to [H,W]
 * (scripts/face_swap.py:291-292,301-303); face/target/out HWC uint8 [B,H,W,3] */
int e4s_paste_u8(const uint8_t* face, const uint8_t* target, const float* mask, uint8_t* out, int B, int H, int W,
                 int Hm, int Wm, void* stream);

/* ---- layout helpers ------------------------------------------------------------------------ */
int e4s_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int H, int W, void* stream);
int e4s_nhwc_to_nchw_f32(const float* x, float* y, int B, int C, int H, int W, void* stream);
/* y[b, :, :, :] = x[0, :, :, :] (ConstantInput.repeat, model.py:345-349), NCHW src -> NHWC dst */
int e4s_const_input_f32(const float* x, float* y, int B, int C, int H, int W, void* stream);

/* ---- regional style encoder (psp_encoders.py:238-309, helpers.py:122-144) ------------------- */
/* bilinear (align_corners=False, no antialias) NCHW [B,C,Hi,Wi] -> NHWC [B,Ho,Wo,C]; networks.py:131 */
int e4s_resize_bilinear_f32(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo, void* stream);
/* direct 3x3 conv for tiny Cin (input layer 3->64): x NHWC [B,H,W,Cin], w [Cout,Cin,3,3] (reference layout) */
int e4s_conv3x3_small_f32(const float* x, const float* w, float* y, int B, int H, int W, int Cin, int Cout, void* stream);
/* the same stem conv on 16x16-pixel tiles for Cin = 3, Cout = 64, H % 16 == W % 16 == 0 (anything else: hipErrorInvalidValue): halo in
 * LDS, weights in registers; stats_ws != NULL: also the per-(sample, channel, tile) fp64 {sum, sum^2} slots of the OUTPUT
 * ((H/16)*(W/16) slots per (b, c): B * 64 * slots * 2 doubles) for e4s_instnorm_finalize_f32 */
int e4s_conv3x3_stem_f32(const float* x, const float* w, float* y, double* stats_ws, int B, int H, int W, int Cin, int Cout, void* stream);
/* InstanceNorm2d statistics (biased var, eps): stats[b, c] = {mean, rstd}; x NHWC [B,HW,C].
 * pooled (optional) [B,C] = spatial mean of the normalised tensor (what SEModule's avg-pool sees).
 * ws: scratch of e4s_instnorm_ws_doubles(B, HW, C) doubles: fp64 partial sums, one slot per (b, c, pixel split), added
 * in a fixed order -- the statistics are bit-reproducible (no floating-point atomics). */
int e4s_instnorm_stats_f32(const float* x, float* stats, float* pooled, double* ws, int B, int HW, int C,
                           float eps, void* stream);
/* second stage alone: ws[((b*C + c)*nslots + k)*2 + {0,1}] = k-th partial {sum, sum of squares} (from a conv epilogue's
 * stats_ws or from e4s_instnorm_apply_stats_f32) -> stats / pooled as above, slots added in order */
int e4s_instnorm_finalize_f32(const double* ws, float* stats, float* pooled, int B, int HW, int C, int nslots, float eps,
                              void* stream);
/* e4s_instnorm_apply_f32 that also emits the partial sums of its OUTPUT (the next unit's InstanceNorm statistics) into
 * ws (e4s_instnorm_ws_doubles(B, H*W, C) doubles; slots = that call's split count, returned through *nslots) */
int e4s_instnorm_apply_stats_f32(const float* x, const float* stats, const float* gate, const float* res,
                                 const float* res_stats, const float* slope, float* y, double* ws, int* nslots,
                                 int B, int H, int W, int C, int rs, void* stream);
int64_t e4s_instnorm_ws_doubles(int B, int HW, int C);
/* y = ((x - mean)*rstd) [* gate[b,c]] [+ res[b, (y*rs)*Wr + x*rs, c]] ; optional PReLU(slope[c]) last.
 * res is NHWC [B, H*rs, W*rs, C] sampled at stride rs (MaxPool2d(1, stride), helpers.py:125-126). */
int e4s_instnorm_apply_f32(const float* x, const float* stats, const float* gate, const float* res,
                           const float* res_stats, const float* slope, float* y,
                           int B, int H, int W, int C, int rs, void* stream);
/* SE gate (helpers.py:56-72): gate[b,c] = sigmoid(fc2 . relu(fc1 . pooled[b]))  */
int e4s_se_gate_f32(const float* pooled, const float* fc1, const float* fc2, float* gate,
                    int B, int C, int Cr, void* stream);
/* e4s_instnorm_finalize_f32 + e4s_se_gate_f32 in one launch (one block per sample): stats [B,C,2] of the tensor whose partial
 * sums are in ws, and the SE gate of its normalisation (bottleneck_IR_SE's IN -> SE tail, helpers.py:56-72,132-144) */
int e4s_instnorm_finalize_se_f32(const double* ws, float* stats, const float* fc1, const float* fc2, float* gate, int B, int HW,
                                 int C, int Cr, int nslots, float eps, void* stream);
/* regional average pooling (psp_encoders.py:264-283): out[b, r, out_off + c] = mean over pixels with
 * label r of feats[b, p, c]; exact 0 for empty regions.  feats NHWC [B,H,W,C]. */
int e4s_region_mean_f32(const float* feats, const uint8_t* labels, int Hm, int Wm, float* out,
                        int B, int H, int W, int C, int R, int out_stride, int out_off, void* stream);
/* Regional style swap (scripts/face_swap.py:117-146, per sample) in one launch: out[b][r] = (sel >> r) & 1 ? src[b][r] : tgt[b][r];
 * region `ear` of a source whose style vector sums to 0 (no ears): the mean of both; region `teeth` likewise empty: the target's;
 * `below` >= 0: the mean of both for that region (belowFace_interpolation).  tgt / src / out [B][R][C], R <= 32. */
int e4s_swap_styles_f32(const float* tgt, const float* src, float* out, int B, int R, int C, unsigned sel, int ear, int teeth, int below,
                        void* stream);
/* LocalMLP layer (networks.py:15-39): y[b, r, o] = act(sum_i x[b, r, i]*W[r][o, i]*scale + bias[r][o]) [+ add[o]]
 * W is [R][O][K] (stacked EqualLinear weights); act: 0 none, 1 leaky(alpha). */
int e4s_grouped_linear_f32(const float* x, const float* W, const float* bias, const float* add, float* y,
                           int B, int R, int K, int O, float scale, int act, float alpha, void* stream);

/* ---- loss networks of the optimisation loop (scripts/optimization.py:88-122) ---------------------- */
/* F.adaptive_avg_pool2d of a crop, with the per-channel affine that follows it: y NHWC [B,Ho,Wo,C] =
 * mean_bin(x[b, c, y0:y0+Hc, x0:x0+Wc]) * scale[c] + shift[c] (scale/shift NULL: none).  x is NCHW (in_nchw = 1, the
 * reference's image layout) or NHWC.  Replaces face_pool_1 / crop / face_pool_2 of src/criteria/id_loss.py:26-29 and
 * F.adaptive_avg_pool2d + BaseNet.z_score of optimization.py:103-106, src/criteria/lpips/networks.py:50-51. */
int e4s_adaptive_pool_f32(const float* x, float* y, int B, int C, int Hi, int Wi, int y0, int x0, int Hc, int Wc, int Ho,
                          int Wo, int in_nchw, const float* scale, const float* shift, void* stream);
/* its backward: dx (layout of x) (+)= ...; pixels outside the crop receive 0 */
int e4s_adaptive_pool_bwd_f32(const float* dy, float* dx, int B, int C, int Hi, int Wi, int y0, int x0, int Hc, int Wc,
                              int Ho, int Wo, int in_nchw, const float* scale, int accumulate, void* stream);
/* k x k conv of an image with Cin <= 4 channels (AlexNet features[0]: 3 -> 64, k 11, stride 4, pad 2): x NHWC, wp
 * [k*k*Cin][Cout] (tap-major), bias [Cout] or NULL, optional ReLU; Cout % 16 == 0.  *_bwd: the gradient w.r.t. x. */
int e4s_conv_smallcin_f32(const float* x, const float* wp, const float* bias, float* y, int B, int Hi, int Wi, int Cin,
                          int Ho, int Wo, int Cout, int k, int stride, int pad, int relu, void* stream);
int e4s_conv_smallcin_bwd_f32(const float* dy, const float* wp, float* dx, int B, int Hi, int Wi, int Cin, int Ho, int Wo,
                              int Cout, int k, int stride, int pad, void* stream);
/* second half of the GEMM form of that gradient for large kernels: z NHWC [B,Ho,Wo,ZC] with z[..., (ky*k+kx)*Cin+ci] = dy . w[(ky,kx,ci)]
 * (one 1x1 contraction on e4s_conv_mfma_f32 with the packed weights as its [ZC][Cout] matrix, ZC = k*k*Cin padded to a multiple of 32);
 * dx[b,y,x,ci] = the sum of the entries that land on the pixel, in (oy, ox) order */
int e4s_conv_smallcin_col2im_f32(const float* z, float* dx, int B, int Hi, int Wi, int Cin, int Ho, int Wo, int ZC, int k, int stride,
                                 int pad, void* stream);
/* nn.MaxPool2d(3, 2): x NHWC [B,Hi,Wi,C] -> y [B,(Hi-3)/2+1,(Wi-3)/2+1,C] and idx (uint8, same shape): position of the first
 * maximum in the window (ATen's tie rule); the backward routes dy through idx. */
int e4s_maxpool3s2_f32(const float* x, float* y, uint8_t* idx, int B, int Hi, int Wi, int C, void* stream);
int e4s_maxpool3s2_bwd_f32(const float* dy, const uint8_t* idx, float* dx, int B, int Hi, int Wi, int C, void* stream);
/* nn.MaxPool2d(2) (the parsing UNet's encoder, src/criteria/face_parsing/unet.py:27-36): y [B,Hi/2,Wi/2,C], idx as above */
int e4s_maxpool2_f32(const float* x, float* y, uint8_t* idx, int B, int Hi, int Wi, int C, void* stream);
int e4s_maxpool2_bwd_f32(const float* dy, const uint8_t* idx, float* dx, int B, int Hi, int Wi, int C, void* stream);
/* ReLU backward from the output: dx (+)= dy * [y > 0]; n % 4 == 0 */
int e4s_relu_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, int accumulate, void* stream);
/* LPIPS distance of one feature level (src/criteria/lpips/lpips.py:32-33 with utils.py:normalize_activation): out[b] =
 * mean_p sum_c w[c] (fx/(|fx|+1e-10) - fy/(|fy|+1e-10))^2, fx / fy NHWC [B,HW,C], C <= 512; ws:
 * e4s_lpips_layer_ws_doubles(B, HW) doubles (ordered reduction).  *_bwd: dfx (+)= gout[0] * gmul * d(out[b])/d(fx). */
int e4s_lpips_layer_f32(const float* fx, const float* fy, const float* w, float* out, double* ws, int B, int HW, int C,
                        void* stream);
int64_t e4s_lpips_layer_ws_doubles(int B, int HW);
int e4s_lpips_layer_bwd_f32(const float* fx, const float* fy, const float* w, const float* gout, float gmul, float* dfx,
                            int B, int HW, int C, int accumulate, void* stream);
/* normalisation with FROZEN statistics (BatchNorm2d in eval mode inside IDLoss, helpers.py:108-113): the two ordered sums
 * of e4s_instnorm_bwd_f32 alone (sums[b,c,1] = dL/dgate), and dx (+)= rstd[b,c] * (gate[b,c] * dy + extra[b,c]). */
int e4s_instnorm_bwd_sums_f32(const float* dy, const float* x, const float* stats, float* sums, double* ws, int B, int HW,
                              int C, void* stream);
int e4s_norm_bwd_frozen_f32(const float* dy, const float* stats, const float* gate, const float* extra, float* dx, int B,
                            int HW, int C, int accumulate, void* stream);
/* cosine similarity of rows (id_loss.py:41-48 on l2-normalised features): out[b] = {sim, alpha, beta} with
 * d(sim)/da = alpha*b + beta*a; ws: e4s_cosine_ws_doubles(B, D) doubles.  *_bwd: da (+)= gout[0]*gmul*(alpha*b + beta*a). */
int e4s_cosine_f32(const float* a, const float* b, float* out, double* ws, int B, int64_t D, void* stream);
int64_t e4s_cosine_ws_doubles(int B, int64_t D);
int e4s_cosine_bwd_f32(const float* a, const float* b, const float* coef, const float* gout, float gmul, float* da, int B,
                       int64_t D, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* E4S_HIP_H */
