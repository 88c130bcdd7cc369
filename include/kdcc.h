/*
 * kdcc.h -- C-ABI of libkdcc_hip.so: the MI355X (gfx950) kernels behind the
 * KD train step of lehduong/Knowledge-Distillation-by-Replacing-Cheap-Conv.
 *
 * The reference has no FFI / operator ABI: its hot path is Python classes
 * resolved by name (parse_config.py:80-93 `init_obj`; train.py:38-71) that
 * bottom out in torch.nn.functional calls.  Each entry point below replaces
 * the torch call(s) cited next to it; the Python host mirror
 * (knowledge-distillation-by-replacing-cheap-conv_amd/) keeps the reference's
 * class names and binds these symbols with ctypes.
 *
 * Conventions
 *  - plain pointers and sizes only; the caller owns every buffer (device memory);
 *    the library allocates nothing persistent and keeps no state;
 *  - all work is enqueued on the hipStream_t passed as `stream` (never the
 *    default stream implicitly, never a synchronisation);
 *  - return 0 on success, a negative KD_ERR_* otherwise; never throws;
 *    kd_last_error() returns a thread-local message for the last failure;
 *  - activations are NHWC ("channels last"): element (n,h,w,c) of a view with
 *    pixel stride `ld` lives at base[((n*H + h)*W + w)*ld + c]; ld >= C lets a
 *    kernel read / write a channel slice of a wider buffer (concat-free ASPP
 *    and decoder, deeplabv3.py:74,157);
 *  - dtype KD_BF16: bf16 storage, fp32 accumulate (the measured path);
 *    dtype KD_F32: fp32 storage, exact-fp32 MFMA (the parity path, 1e-3 gate).
 */
#ifndef KDCC_H
#define KDCC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *kd_stream_t; /* hipStream_t */

enum { KD_F32 = 0, KD_BF16 = 1 };
enum {
    KD_OK = 0,
    KD_ERR_INVALID = -1,     /* bad descriptor / null pointer / misalignment */
    KD_ERR_UNSUPPORTED = -2, /* shape outside what the kernels implement */
    KD_ERR_WORKSPACE = -3,   /* workspace too small */
    KD_ERR_HIP = -4          /* launch failed (hipGetLastError) */
};

int kd_version(void);
const char *kd_last_error(void);

/* ---------------------------------------------------------------- dense conv
 * Implicit-GEMM convolution on MFMA (im2col tile gathered straight into LDS).
 * Replaces nn.Conv2d forward at models/encoders/wider_resnet.py:124-167,
 * models/deeplabv3/deeplabv3.py:46-55,127-137 and the 1x1 `pointwise_conv` of
 * models/students/transform_blocks/depthwise_separable_conv.py:9, together
 * with the BN(eval)+ReLU (wider_resnet.py:43-48) and in-place residual add
 * (wider_resnet.py:181) that surround them, which are fused into the epilogue.
 * The same entry point computes the input gradient (autograd of those calls,
 * reached from loss.backward(), trainer/layerwise_trainer.py:235) when given
 * weights packed by kd_pack_conv_weight(..., KD_PACK_DGRAD) and the gradient as
 * `x`: for stride 1 the dgrad of a conv is a conv with flipped taps.
 *
 * Requirements: kh*kw in {1, 9}; Cin % 64 == 0 (bf16) or % 32 == 0 (f32);
 * x and w 16-byte aligned, ldx*sizeof(elem) % 16 == 0.
 * w: packed [Cout][kh][kw][Cin] in `dtype`.
 */
typedef struct kd_conv_desc {
    int32_t dtype;
    int32_t N, H, W, Cin; /* input view */
    int32_t Ho, Wo, Cout; /* output view */
    int32_t kh, kw, stride, pad, dil;
    int32_t ldx;          /* input pixel stride, elements */
} kd_conv_desc;

/* Epilogue, applied per output element (m = pixel, c = output channel):
 *   v = acc
 *   if res_pre   v += res_pre[m*ld_res_pre + c]
 *   if mask      v = mask[m*ld_mask + c] > 0 ? v * (mask_scale ? mask_scale[c] : 1) : 0
 *   if res_post  v += res_post[m*ld_res_post + c]
 *   if out_raw   out_raw[m*ld_raw + c] = v            (as float when raw_f32 != 0)
 *   if out_act   a = v * (act_scale ? act_scale[c] : 1) + (act_shift ? act_shift[c] : 0);
 *                out_act[m*ld_act + c] = act_relu ? max(a, 0) : a
 * forward:  res_pre = shortcut, out_raw = block output / hint, out_act = input of the next conv
 *           (next layer's eval-mode BN folded to scale/shift, + ReLU);
 * backward: mask = the saved activated tensor, mask_scale = that BN's scale
 *           (d relu(bn(x)) / dx), res_post = gradient arriving over the identity shortcut.
 */
typedef struct kd_conv_epilogue {
    const void *res_pre;  int32_t ld_res_pre;
    const void *mask;     int32_t ld_mask;     const float *mask_scale;
    const void *res_post; int32_t ld_res_post;
    void *out_raw;        int32_t ld_raw;      int32_t raw_f32;
    void *out_act;        int32_t ld_act;      const float *act_scale; const float *act_shift; int32_t act_relu;
    float *bn_sums;       /* optional, with `mask`: [M/128][2][Cout] fp32 partial sums over blocks of 128 output pixels of
                             g = (mask > 0 ? v * mask_scale : 0) and of g * mask -- the eval-mode BN parameter gradients'
                             reductions, taken where the input gradient is produced instead of by kd_channel_sums reading it
                             back.  Only the kernels kd_conv2d_bn_sums_rows() reports write it; finish with kd_bn_sums_finish.
                             WITHOUT `mask` (forward, round 6): part[r][0][c] = sum over the 128 pixels of row block r of the
                             values stored to out_raw, part[r][1][c] = 0 -- the global average pool of the ASPP image-pooling branch
                             (models/deeplabv3/deeplabv3.py:59-62) taken where its input is produced; kd_aspp_image_pool_sums
                             consumes the rows.  Only the ping-pong 1x1 kernel with no epilogue operand and out_raw alone. */
    /* Optional, round 6: a 1x1 classifier applied to the activation INSTEAD of storing it (out_raw and out_act NULL):
     *   cls_out[m][c] = sum_k bf16(act(scale[k] * v[m][k] + shift[k])) * cls_w[c][k],  c < ncls <= 32, fp32, pixel stride ld_cls floats;
     * cls_w is a [32][Cout] matrix in the conv's dtype, rows >= ncls zero (kd_pack_conv_weight of the 1x1 weight padded to 32 outputs).
     * The reference's `final` head (models/deeplabv3/deeplabv3.py:127-139: conv3x3 -> BN -> ReLU -> conv1x1 onto the classes) where nothing
     * else reads the 256-channel activation: it is neither written nor read back.  Same bf16 activation and bf16 weights as the
     * two-launch form, fp32 summation in another order.  Only conv_row_lw_kernel (bf16 3x3 / stride 1 / 'same', Cout == 256, no epilogue
     * operand, no bn_sums); kd_conv2d_cls_supported() says whether (d, ep) selects it, kd_conv2d_fwd refuses otherwise. */
    const void *cls_w;    float *cls_out;      int32_t ld_cls;      int32_t ncls;
} kd_conv_epilogue;
int32_t kd_conv2d_cls_supported(const kd_conv_desc *d, const kd_conv_epilogue *ep);

int kd_conv2d_fwd(const kd_conv_desc *d, const void *x, const void *w_packed,
                  const kd_conv_epilogue *ep, kd_stream_t stream);
/* Rows of `bn_sums` partials (one per 128 output pixels) the kernel kd_conv2d_fwd selects for (d, ep) writes, or 0 when that
 * kernel does not produce them -- the caller then leaves ep->bn_sums NULL (kd_conv2d_fwd refuses it otherwise) and takes the
 * sums with kd_channel_sums from the stored gradient.  The reference gets these reductions from autograd of
 * bn -> relu in IdentityResidualBlock (models/encoders/wider_resnet.py:124-167, reached from loss.backward(),
 * trainer/layerwise_trainer.py:235).
 * kd_bn_sums_finish: s1[c] = sum_r part[r][0][c], s2[c] = sum_r part[r][1][c] in a fixed order (fp64 accumulators; more than
 *   256 rows go through a first stage of 64- or 256-row chunks in `workspace`). */
int32_t kd_conv2d_bn_sums_rows(const kd_conv_desc *d, const kd_conv_epilogue *ep);
size_t kd_bn_sums_finish_workspace(int32_t rows, int32_t C);
int kd_bn_sums_finish(const float *part, int32_t rows, int32_t C, float *s1, float *s2, void *workspace, size_t workspace_bytes,
                      kd_stream_t stream);

/* K-concatenated 1x1 convolution: y = epilogue([x | x2] . w_cat^T) with ONE accumulator chain over K = d->Cin + Cin2 -- the two
 * 1x1 convs that land on the same tensor in the bottleneck blocks of WRN-38 (models/encoders/wider_resnet.py:143-182:
 * `out = self.convs(bn1); out.add_(shortcut)` with `shortcut = self.proj_conv(bn1)`: conv3 and proj_conv in mod6 / mod7), and in
 * their backward the input gradients of conv1 and proj_conv, which both land on bn1's output.  The shortcut tensor is never
 * written and read back and one wide epilogue disappears.  d describes the FIRST source (N,H,W,Cin,ldx; 1x1, stride 1, pad 0,
 * bf16); x2 is an (N,H,W,Cin2) view with pixel stride ldx2; w_cat is [Cout][Cin + Cin2] in `dtype` (the two packed weights
 * concatenated along K).  Runs on the persistent 256 x 256 ping-pong kernel only: kd_conv1x1_dual_supported() says whether (d,
 * Cin2, ep) selects it (M % 256 == 0, Cout % 256 == 0, >= 224 tiles, 16-B friendly epilogue); otherwise KD_ERR_UNSUPPORTED and
 * the caller runs the two convs (the second with res_pre).  Same values as those two launches up to the bf16 rounding of the
 * intermediate (the concatenated form rounds once). */
int32_t kd_conv1x1_dual_supported(const kd_conv_desc *d, int32_t Cin2, int32_t ldx2, const kd_conv_epilogue *ep);
int kd_conv1x1_dual_fwd(const kd_conv_desc *d, const void *x, const void *x2, int32_t Cin2, int32_t ldx2, const void *w_cat,
                        const kd_conv_epilogue *ep, kd_stream_t stream);

/* Workgroups of the persistent conv kernels' grids from now on: 0 = one per CU (default), else n rounded down to a multiple of 8
 * and clamped to [8, CU count] (0 < n < 8 gives 8, n above the CU count gives the full chip; the value is process-wide and atomic) -- leaves CUs to a kernel on another stream, i.e. the RCCL all-reduce of a gradient bucket that
 * parallel.GradReducer launches from inside backward (the reference's single-process trainer has no counterpart,
 * trainer/layerwise_trainer.py:235-239).  Overrides the KDCC_PERSIST_CUS environment default; results are bit-identical for any n. */
int kd_conv_set_persist_cus(int32_t n);

/* Weight packing (runs on device, on `stream`).  src: the reference's
 * nn.Conv2d.weight, fp32 (Cout, Cin, kh, kw) contiguous.
 *   KD_PACK_FWD   dst[co][ky][kx][ci]            = src[co][ci][ky][kx]
 *   KD_PACK_DGRAD dst[ci][kh-1-ky][kw-1-kx][co]  = src[co][ci][ky][kx]
 * cin_pad >= Cin (fwd) zero-fills channels [Cin, cin_pad) (decoder 304 -> 320). */
enum { KD_PACK_FWD = 0, KD_PACK_DGRAD = 1 };
int kd_pack_conv_weight(const float *src, void *dst, int32_t dtype, int32_t mode,
                        int32_t Cout, int32_t Cin, int32_t kh, int32_t kw, int32_t cin_pad,
                        kd_stream_t stream);

/* Weight gradient of a 1x1 convolution (the trainable `pointwise_conv`,
 * depthwise_separable_conv.py:9):  dw[co][ci] = sum_m dy[m][co] * a[m][ci],
 * fp32 (Cout, Cin, 1, 1) like the reference's .grad.  accumulate != 0 adds into dw
 * (gradient accumulation, layerwise_trainer.py:237-239).
 * workspace: kd_pw_wgrad_workspace() bytes. */
size_t kd_pw_wgrad_workspace(int32_t M, int32_t Cin, int32_t Cout);
int kd_pw_wgrad(int32_t dtype, int32_t M, int32_t Cin, int32_t Cout,
                const void *a, int32_t lda, const void *dy, int32_t ldy,
                float *dw, int32_t accumulate, void *workspace, size_t workspace_bytes,
                kd_stream_t stream);

/* Weight gradient of a general convolution (autograd of nn.Conv2d w.r.t. its weight, reached from loss.backward(),
 * trainer/layerwise_trainer.py:235 / trainer/classification_trainer.py:39, whenever a dense conv of the student is
 * trainable: `pruning.unfreeze` naming a dense block, models/students/depthwise_student.py:80-84, or the "identical
 * architecture" branch that unfreezes everything, layerwise_trainer.py:88-100):
 *   dw[co][ci][ky][kx] = sum_{n,ho,wo} dy[n,ho,wo,co] * x[n, ho*stride - pad + ky*dil, wo*stride - pad + kx*dil, ci]
 * fp32 (Cout, Cin, kh, kw) like the reference's .grad; d describes the forward conv (x view N,H,W,Cin with stride ldx;
 * dy view N,Ho,Wo,Cout with stride ld_dy).  One TN GEMM per tap over pixel splits, partial slabs in the workspace,
 * fixed-order reduction (deterministic).  Cin, Cout: any (multiples of 8 take the LDS-DMA path in bf16). */
size_t kd_conv2d_wgrad_workspace(const kd_conv_desc *d);
int kd_conv2d_wgrad(const kd_conv_desc *d, const void *x, const void *dy, int32_t ld_dy, float *dw,
                    int32_t accumulate, void *workspace, size_t workspace_bytes, kd_stream_t stream);

/* ------------------------------------------------------------ depthwise conv
 * `separable_conv` of depthwise_separable_conv.py:7-8: Conv2d(C, C, k,
 * padding, dilation, groups=C), stride 1 (9x9 / dilation 5 / padding 20 in
 * cfg/cityscapes/58M_deeplab_all.json:117-122; 3x3 in the CIFAR configs).
 * w_taps: [k*k][C] float (tap-major), from kd_pack_dw_weight.
 *   flip != 0 in the pack gives the dgrad operand (dx = dwconv(dy, flipped w)
 *   with pad' = dil*(k-1) - pad).
 * bias: (C) float or NULL (always NULL for the WRN-38 / ASPP targets). */
typedef struct kd_dw_desc {
    int32_t dtype;
    int32_t N, H, W, C;
    int32_t k, pad, dil;
    int32_t ldx, ldy;
} kd_dw_desc;
/* Optional epilogue (NULL = none), same meaning and order as kd_conv_epilogue's first three
 * steps; used when the kernel computes an input gradient that continues through the
 * BN(eval)+ReLU in front of the replaced conv, or accumulates into an existing gradient:
 *   v = acc; v += res_pre; v = mask > 0 ? v*mask_scale[c] : 0; v += res_post; y = v */
typedef struct kd_dw_epilogue {
    const void *res_pre;  int32_t ld_res_pre;
    const void *mask;     int32_t ld_mask;     const float *mask_scale;
    const void *res_post; int32_t ld_res_post;
} kd_dw_epilogue;
int kd_pack_dw_weight(const float *src /* (C,1,k,k) */, float *dst /* [k*k][C] */,
                      int32_t C, int32_t k, int32_t flip, kd_stream_t stream);
int kd_dwconv_fwd(const kd_dw_desc *d, const void *x, const float *w_taps, const float *bias,
                  const kd_dw_epilogue *ep, void *y, kd_stream_t stream);
/* y = sum_{i<n} dwconv(xs[i], w_taps[i]): the gradient of ONE tensor read by n depthwise convs of one geometry -- the
 * ASPP input under its replaced branches (models/deeplabv3/deeplabv3.py:64-75: every branch of `features` reads the same
 * x, so autograd sums their input gradients; with flipped tap tables and pad' as above each term is a dwconv).  All inputs
 * share d (shape, ldx); y must not alias an input.  n <= 3 bf16 9x9 inputs are summed in registers inside one launch. */
int kd_dwconv_fwd_sum(const kd_dw_desc *d, int32_t n, const void *const *xs, const float *const *w_taps, void *y,
                      kd_stream_t stream);
/* ys[i] = dwconv(x, w_taps[i]), i < n: the forward of the same fan-out (deeplabv3.py:71-75, `for f in self.features:
 * out = torch.cat((out, f(x)), 1)` with each f's 3x3 conv replaced by a DepthwiseSeparableBlock): one pass over x for up to
 * three bf16 9x9 outputs per launch.  Outputs share d->ldy and must not alias x or each other. */
int kd_dwconv_fwd_fanout(const kd_dw_desc *d, int32_t n, const void *x, const float *const *w_taps, void *const *ys,
                         kd_stream_t stream);
/* dw[c][ky][kx] = sum_{n,h,w} dy[n,h,w,c] * x[n,h-pad+ky*dil,w-pad+kx*dil,c]; fp32 (C,1,k,k). */
size_t kd_dwconv_wgrad_workspace(const kd_dw_desc *d);
int kd_dwconv_wgrad(const kd_dw_desc *d, const void *x, const void *dy, int32_t ld_dy,
                    float *dw, int32_t accumulate, void *workspace, size_t workspace_bytes,
                    kd_stream_t stream);
/* dws[i] = the weight gradient above for x and dys[i], i < n: n depthwise convs of one geometry that read ONE input -- the
 * replaced ASPP branches again (models/deeplabv3/deeplabv3.py:71-75; autograd of depthwise_separable_conv.py:7-8 per
 * branch).  For n = 2, 3 bf16 9x9 branches one launch stages each tile of x once and shares its operand windows between the
 * branches; other shapes run one kd_dwconv_wgrad per branch.  All dys share ld_dy. */
size_t kd_dwconv_wgrad_multi_workspace(const kd_dw_desc *d, int32_t n);
int kd_dwconv_wgrad_multi(const kd_dw_desc *d, int32_t n, const void *x, const void *const *dys, int32_t ld_dy,
                          float *const *dws, int32_t accumulate, void *workspace, size_t workspace_bytes,
                          kd_stream_t stream);

/* ------------------------------------------------ lattice-planar intermediates of the replaced ASPP branches
 * The depthwise outputs of the replaced branches and their gradients (models/students/transform_blocks/
 * depthwise_separable_conv.py:11-13, `x = self.separable_conv(x); x = self.pointwise_conv(x)`, under
 * models/deeplabv3/deeplabv3.py:64-75) are 4096-channel tensors that nothing but the depthwise kernels and the 1x1 convs next
 * to them ever reads.  In NHWC a depthwise workgroup (16 channels) moves them as 32-B pieces 40 KiB apart; in the
 * LATTICE-PLANAR layout of a dilation d
 *     [C/16 planes][rows][16 channels],   row(n, ry, rx, ly, lx) = ((n*d*d + ry*d + rx)*Ly + ly)*Lx + lx,
 *     Ly = ceil(H/d), Lx = ceil(W/d),  pixel (y, x) = (ry + d*ly, rx + d*lx),  rows = kd_lattice_rows() (a multiple of 256)
 * the pixels of one residue class of one image are consecutive rows, so the same tile is one contiguous run per lattice row.
 * Cells whose pixel is outside the image and the tail rows of a plane hold zeros (writers keep them zero).
 * EXPERIMENTAL / benchmark-only: no GEMM entry point of this library consumes the [C/16][rows][16] layout (the 1x1 convs next
 * to the depthwise kernels read NHWC), so nothing in the engine uses these entry points; they exist to measure what contiguous
 * depthwise intermediates are worth (1-5 %, profiles/r05_aspp_dw_lattice_ab.json) and are tested bit for bit against the NHWC
 * ones.  kd_lattice_rows_move converts between a dense [rows][C] matrix in lattice row order and image order. */
int64_t kd_lattice_rows(int32_t N, int32_t H, int32_t W, int32_t dil);
/* 1 when the fan-out / sum / multi-gradient launches of n (2 or 3) branches of geometry d can run on lattice-planar
 * intermediates (bf16, 9x9, C % 16 == 0, 32-bit plane offsets); otherwise the caller keeps NHWC intermediates. */
int32_t kd_dwconv_lattice_ok(const kd_dw_desc *d, int32_t n);
/* to_rows != 0: rows[r][c] = img[n, y, x, c] for the pixel of lattice row r (zero rows for padded cells and the tail);
 * to_rows == 0: img[n, y, x, c] = rows[r][c] for every pixel.  img: (N,H,W,C) view with pixel stride ld_img; rows:
 * kd_lattice_rows() x C with row stride ld_rows.  C and both strides multiples of 16 B. */
int kd_lattice_rows_move(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dil, void *img, int32_t ld_img,
                         void *rows, int32_t ld_rows, int32_t to_rows, kd_stream_t stream);
/* kd_dwconv_fwd_fanout / kd_dwconv_fwd_sum / kd_dwconv_wgrad_multi with the n-side tensors lattice-planar: the fan-out's
 * outputs ys[] (x stays NHWC), the sum's inputs xs[] (y stays NHWC, stride d->ldy), the gradients dys[] (x stays NHWC).
 * n = 2 or 3; KD_ERR_UNSUPPORTED unless kd_dwconv_lattice_ok(d, n).  Same values as the NHWC entry points. */
int kd_dwconv_fwd_fanout_lattice(const kd_dw_desc *d, int32_t n, const void *x, const float *const *w_taps, void *const *ys,
                                 kd_stream_t stream);
int kd_dwconv_fwd_sum_lattice(const kd_dw_desc *d, int32_t n, const void *const *xs, const float *const *w_taps, void *y,
                              kd_stream_t stream);
int kd_dwconv_wgrad_multi_lattice(const kd_dw_desc *d, int32_t n, const void *x, const void *const *dys, float *const *dws,
                                  int32_t accumulate, void *workspace, size_t workspace_bytes /* kd_dwconv_wgrad_multi_workspace */,
                                  kd_stream_t stream);

/* ----------------------------------------------------------- trunk plumbing
 * Stem conv mod1.conv1 (wider_resnet.py:307-309): 3x3, 3 -> 64, stride 1, pad 1,
 * reading the trainer's NCHW fp32 batch directly (layerwise_trainer.py:221).
 * w: fp32 (64,3,3,3) as in the reference.  y: NHWC `dtype`, 64 channels. */
int kd_stem_conv(int32_t dtype, const float *x_nchw, const float *w, void *y,
                 int32_t N, int32_t H, int32_t W, kd_stream_t stream);
/* The same conv followed by pool2 = MaxPool2d(3, stride=2, padding=1) (wider_resnet.py:307-309, 353-356: mod1's output
 * has no other reader in DeepWV3Plus) and, for y_act, the BN(eval)+ReLU of mod2.block1 (scale/shift as in
 * kd_maxpool3x3s2), without the 64-channel full-resolution tensor ever reaching memory: for a frozen stem whose
 * output nothing else needs.  bf16 outputs (N,Ho,Wo,64), Ho = (H-1)/2+1; bit-identical to
 * kd_stem_conv(KD_BF16) + kd_maxpool3x3s2.  y_raw or y_act may be NULL. */
int kd_stem_conv_pool(const float *x_nchw, const float *w, void *y_raw, void *y_act, const float *scale,
                      const float *shift, int32_t N, int32_t H, int32_t W, kd_stream_t stream);

/* MaxPool2d(3, stride=2, padding=1) (wider_resnet.py:353-356), optionally followed
 * by the next block's BN(eval)+ReLU: y_raw (may be NULL) = pool, y_act (may be NULL) =
 * relu(pool*scale+shift). */
int kd_maxpool3x3s2(int32_t dtype, const void *x, int32_t ldx, void *y_raw, int32_t ld_raw,
                    void *y_act, int32_t ld_act, const float *scale, const float *shift,
                    int32_t N, int32_t H, int32_t W, int32_t C, kd_stream_t stream);

/* F.interpolate(bilinear, align_corners=True) (deeplabv3.py:16-18,69,155,160).
 * Input and output element types are given separately (the final logits are
 * produced in fp32 from an fp32 decoder output whatever the compute dtype). */
int kd_upsample_bilinear_ac(const void *x, int32_t x_dtype, int32_t ldx, void *y, int32_t y_dtype, int32_t ldy,
                            int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                            kd_stream_t stream);

/* The same with the align_corners flag: GSCNN's last upsample (models/gscnn/gscnn.py:323) omits align_corners, i.e.
 * src = max((o + 0.5) * I/O - 0.5, 0). */
int kd_upsample_bilinear(const void *x, int32_t x_dtype, int32_t ldx, void *y, int32_t y_dtype, int32_t ldy,
                         int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, int32_t align_corners,
                         kd_stream_t stream);

/* ASPP image-pooling branch (deeplabv3.py:59-62,67-70): AdaptiveAvgPool2d(1) ->
 * 1x1 conv (w fp32 (Cout,Cin)) -> BN(eval) scale/shift -> ReLU -> broadcast
 * ("upsample" of a 1x1 map) into y[.., 0:Cout] of an NHWC view with stride ldy.
 * workspace: kd_aspp_image_pool_workspace() bytes (pixel-chunk partial sums, reduced in fixed order). */
size_t kd_aspp_image_pool_workspace(int32_t N, int32_t Cin, int32_t Cout);
int kd_aspp_image_pool(int32_t dtype, const void *x, int32_t ldx, const float *w, const float *scale,
                       const float *shift, void *y, int32_t ldy, int32_t N, int32_t H, int32_t W,
                       int32_t Cin, int32_t Cout, void *workspace, size_t workspace_bytes, kd_stream_t stream);
/* The same branch from per-128-pixel channel sums that the producing conv's epilogue already took (kd_conv_epilogue.bn_sums without
 * a mask: `part` = [N*H*W/128][2][Cin], rows of an image consecutive; H*W % 128 == 0): no pass over x.  Rows are added in order
 * (bit-reproducible).  Same workspace size as kd_aspp_image_pool. */
int kd_aspp_image_pool_sums(int32_t dtype, const float *part, const float *w, const float *scale, const float *shift, void *y,
                            int32_t ldy, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, void *workspace,
                            size_t workspace_bytes, kd_stream_t stream);

/* ------------------------------------------------- backward of the trunk plumbing
 * (autograd of the calls above; needed once gradients flow through the whole student: loss = kd + hint with every
 * parameter trainable, or hints taken behind a BN+ReLU such as the `aspp` module output).
 *
 * kd_stem_wgrad: weight gradient of mod1.conv1 (wider_resnet.py:307-309) from the trainer's NCHW fp32 batch:
 *   dw (64,3,3,3) fp32 = sum_pixels dy[n,h,w,:] (x) x[n,:,h-1+ky,w-1+kx].
 * kd_maxpool3x3s2_bwd: gx[n,h,w,c] = sum of gy over the (<= 4) windows whose first maximum is (h,w)
 *   (nn.MaxPool2d(3,2,1) backward, wider_resnet.py:353-356); x is the pool's input.  workspace (optional; one byte per window
 *   and channel, kd_maxpool3x3s2_bwd_workspace) selects the two-pass arg-max-index path; without it a slower gather runs.
 * kd_upsample_bilinear_ac_bwd: transpose of kd_upsample_bilinear_ac (deeplabv3.py:16-18,155,160): gy (N,Ho,Wo,C) ->
 *   gx (N,H,W,C); separable gather, workspace = one (N,Ho,W,C) float plane.
 * kd_zero_insert: y[n, h*stride, w*stride, :] = x[n,h,w,:], zeros elsewhere (Hy x Wy output): turns the input gradient
 *   of a stride-s conv (mod4.block1, wider_resnet.py:322-332) into a stride-1 kd_conv2d_fwd with KD_PACK_DGRAD weights.
 * kd_relu_bn_bwd: y = (mask > 0 ? g * scale[c] : 0) + res -- d relu(bn_eval(x))/dx applied to a gradient that arrives
 *   behind the activation (wider_resnet.py:43-48); res may be NULL.
 * kd_channel_sums: s1[b][c] = sum_m (g - sub)[m][c], s2[b][c] = sum_m (g - sub)[m][c] * a[m][c] over `groups` groups of
 *   rows_per_group consecutive rows (sub, a, s2 optional); float (groups, C); two fixed-order stages.
 * kd_bn_eval_param_grads: eval-mode BatchNorm2d weight / bias gradients from those sums of the gradient w.r.t. the BN
 *   input: dbeta = s1/scale, dgamma = (s2 - beta*s1)/(scale*gamma)   (scale = gamma/sqrt(var+eps), s2 taken against
 *   relu(bn(x)); masked elements carry zero gradient).
 * kd_broadcast_add: y[n,p,c] = (accumulate ? y : 0) + alpha * v[n,c] (backward of the ASPP image-pooling broadcast). */
size_t kd_stem_wgrad_workspace(int32_t N, int32_t H, int32_t W);
int kd_stem_wgrad(int32_t dtype, const float *x_nchw, const void *dy, int32_t ld_dy, float *dw, int32_t N, int32_t H,
                  int32_t W, int32_t accumulate, void *workspace, size_t workspace_bytes, kd_stream_t stream);
size_t kd_maxpool3x3s2_bwd_workspace(int32_t N, int32_t H, int32_t W, int32_t C);
int kd_maxpool3x3s2_bwd(int32_t dtype, const void *x, int32_t ldx, const void *gy, int32_t ldgy, void *gx, int32_t ldgx,
                        int32_t N, int32_t H, int32_t W, int32_t C, void *workspace, size_t workspace_bytes,
                        kd_stream_t stream);
size_t kd_upsample_bilinear_ac_bwd_workspace(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo);
int kd_upsample_bilinear_ac_bwd(const void *gy, int32_t gy_dtype, int32_t ldgy, void *gx, int32_t gx_dtype, int32_t ldgx,
                                int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                                void *workspace, size_t workspace_bytes, kd_stream_t stream);
/* The same transpose for either corner convention (align_corners = 0: the last upsample of GSCNN, models/gscnn/gscnn.py:323). */
int kd_upsample_bilinear_bwd(const void *gy, int32_t gy_dtype, int32_t ldgy, void *gx, int32_t gx_dtype, int32_t ldgx,
                             int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, int32_t align_corners,
                             void *workspace, size_t workspace_bytes, kd_stream_t stream);
int kd_zero_insert(int32_t dtype, const void *x, int32_t ldx, void *y, int32_t ldy, int32_t N, int32_t H, int32_t W,
                   int32_t C, int32_t stride, int32_t Hy, int32_t Wy, kd_stream_t stream);
int kd_relu_bn_bwd(int32_t dtype, const void *g, int32_t ldg, const void *mask, int32_t ldm, const float *scale,
                   const void *res, int32_t ldres, void *y, int32_t ldy, int64_t M, int32_t C, kd_stream_t stream);
size_t kd_channel_sums_workspace(int32_t groups, int64_t rows_per_group, int32_t C);
int kd_channel_sums(int32_t dtype, const void *g, int32_t ldg, const void *sub, int32_t ldsub, const void *a, int32_t lda,
                    int32_t groups, int64_t rows_per_group, int32_t C, float *s1, float *s2,
                    void *workspace, size_t workspace_bytes, kd_stream_t stream);
int kd_bn_eval_param_grads(const float *s1, const float *s2, const float *scale, const float *gamma, const float *beta,
                           float *dgamma, float *dbeta, int32_t C, int32_t accumulate, kd_stream_t stream);
int kd_broadcast_add(int32_t dtype, const float *v, void *y, int32_t ldy, int32_t N, int64_t HW, int32_t C, float alpha,
                     int32_t accumulate, kd_stream_t stream);

/* BN(eval) scale/shift from running statistics (nn.BatchNorm2d in eval mode):
 *   scale = gamma / sqrt(var + eps),  shift = beta - mean * scale. */
int kd_bn_fold(const float *gamma, const float *beta, const float *mean, const float *var, float eps,
               float *scale, float *shift, int32_t C, kd_stream_t stream);

/* generic strided copy / cast between NCHW-or-NHWC fp32/bf16 views:
 * dst(n,c,p) = src(n,c,p), element (n,c,p) at n*sN + c*sC + p*sP (elements). */
int kd_copy_cast(const void *src, int32_t src_dtype, int64_t s_sN, int64_t s_sC, int64_t s_sP,
                 void *dst, int32_t dst_dtype, int64_t d_sN, int64_t d_sC, int64_t d_sP,
                 int32_t N, int32_t C, int64_t P, kd_stream_t stream);

/* ------------------------------------------------- small-shape path (CIFAR plumbing config)
 * NCHW fp32, the reference's own layout, for the shapes the MFMA kernels do not take: CIFAR ResNet-20
 * (models/cifar_models/resnet.py: 3/16/32/64 channels, 32x32..8x8) and the 3x3 cheap-conv blocks of the CIFAR configs
 * (depthwise_separable_conv.py:7-9 with groups = C and optional bias).
 * kd_conv2d_direct_*: nn.Conv2d forward / input gradient / weight (+ bias) gradient for any channels, kernel, stride, padding,
 *   dilation and groups.  w: (K, C/groups, kh, kw); dw like w; dbias (K) or NULL.
 * kd_bn2d_fwd / kd_bn2d_bwd: nn.BatchNorm2d forward / backward, optional fused ReLU.  training != 0: batch statistics (biased
 *   variance), save_mean / save_invstd (C) returned for the backward, running statistics updated in place with `momentum`
 *   and the unbiased variance (trainer/classification_trainer.py:21: the student runs in train mode);
 *   training == 0: the running statistics.  bwd: y (the forward output) is needed only with relu != 0; dx / dgamma / dbeta
 *   may each be NULL. */
typedef struct kd_dconv_desc {
    int32_t N, C, H, W;   /* input  (N, C, H, W) */
    int32_t K;            /* output channels */
    int32_t kh, kw, stride, pad, dil, groups;
} kd_dconv_desc;
int kd_conv2d_direct_fwd(const kd_dconv_desc *d, const float *x, const float *w, const float *bias, float *y,
                         kd_stream_t stream);
int kd_conv2d_direct_dgrad(const kd_dconv_desc *d, const float *dy, const float *w, float *dx, kd_stream_t stream);
int kd_conv2d_direct_wgrad(const kd_dconv_desc *d, const float *x, const float *dy, float *dw, float *dbias,
                           int32_t accumulate, kd_stream_t stream);
int kd_bn2d_fwd(const float *x, const float *gamma, const float *beta, float *y, float *save_mean, float *save_invstd,
                float *running_mean, float *running_var, float momentum, float eps, int32_t training, int32_t relu,
                int32_t N, int32_t C, int32_t HW, kd_stream_t stream);
int kd_bn2d_bwd(const float *dy, const float *x, const float *y, const float *gamma, const float *save_mean,
                const float *save_invstd, float *dx, float *dgamma, float *dbeta, int32_t training, int32_t relu,
                int32_t accumulate, int32_t N, int32_t C, int32_t HW, kd_stream_t stream);

/* ------------------------------------------------- Gated-SCNN shape stream (BASELINE config 5)
 * The full-resolution pieces of models/gscnn/gscnn.py:183-325 that are not MFMA-sized convolutions.
 * kd_gated_conv: GatedSpatialConv2d.forward (models/gscnn/gate_spatial_conv.py:50-60) fused per pixel, C in {8,16,32}:
 *     u = [feat(C); gate(1)];  z = relu(W1 u + b1);  alpha = sigmoid(w2 . z + b2);  out = Wg (feat * (alpha + 1))
 *   params (fp32, contiguous): W1 [(C+1)*(C+1)], b1 [C+1], w2 [C+1], b2 [1], Wg [C*C] -- the module's two eval-mode
 *   BatchNorms folded in by the caller.  feat / out: NHWC views (ld), gate: one value per pixel (stride ldg).
 * kd_edge_attention: acts = sigmoid(cw0 * sigmoid(fuse . cs) + cw1 * canny) (gscnn.py:308-314); weights = fuse[8], cw[2].
 * kd_edge_aspp: edge branch of the ASPP module (gscnn.py:168-171): acts (N,H,W) float resampled bilinearly (align_corners)
 *   to Ho x Wo, then 1x1 conv 1 -> C (w), BN scale/shift, ReLU, written into a channel slice y (ld).
 * kd_canny: the edge map the reference gets from cv2.Canny(uint8(image), low, high) on the host (gscnn.py:284-288), on the
 *   device: Sobel 3x3 per colour channel (largest L1 magnitude wins), non-maximum suppression, hysteresis sweeps.  x: the
 *   trainer's NCHW float batch; out: (N,H,W) float 0 / 255.  `sweeps` hysteresis sweeps run; *changed (device int) ends as
 *   the number of blocks that still promoted a pixel in the last sweep: call kd_canny_continue until it reads 0.
 *   Parity of this operator is unpinned (the reference's arithmetic lives in opencv-python, not vendored). */
/* 3x3 / stride 1 / pad 1 convolution on C = 16, 32 or 64 channels, bf16 NHWC: the two convs of the shape stream's BasicBlocks res1 / res2 / res3
 * (models/encoders/Resnet.py:64-99, models/gscnn/gscnn.py:237-243) without padding them to the 64-channel GEMM granule.
 * w: bf16 [C][3][3][C] (kd_pack_conv_weight, KD_PACK_FWD; an eval-mode BatchNorm folded in by the caller), bias: fp32 (C) or NULL
 * (the folded BN shift), res: optional residual added before the ReLU (the block's identity shortcut), relu: 0 / 1.
 *   y = relu?(conv(x, w) + bias + res) */
int kd_conv3x3_small(const void *x, int32_t ldx, const void *w, const float *bias, const void *res, int32_t ldres, void *y,
                     int32_t ldy, int32_t N, int32_t H, int32_t W, int32_t C, int32_t relu, kd_stream_t stream);
/* 1x1 conv + bias on few channels, bf16 NHWC: the squeezes d1 / d2 / d3 of the shape stream (models/gscnn/gscnn.py:232-235,
 * Conv2d(64, 32, 1), Conv2d(32, 16, 1), Conv2d(16, 8, 1) at full resolution).  w: fp32 (Cout, Cin), bias: fp32 (Cout) or NULL. */
int kd_pointwise_small(const void *x, int32_t ldx, const float *w, const float *bias, void *y, int32_t ldy, int64_t npix,
                       int32_t Cin, int32_t Cout, kd_stream_t stream);
int kd_gated_conv(int32_t dtype, const void *feat, int32_t ldf, const void *gate, int32_t ldg, const float *params,
                  void *out, int32_t ldo, int64_t npix, int32_t C, kd_stream_t stream);
int kd_edge_attention(int32_t dtype, const void *cs, int32_t ldc, const float *canny, const float *weights, float *acts,
                      int64_t npix, kd_stream_t stream);
int kd_edge_aspp(int32_t dtype, const float *acts, int32_t H, int32_t W, const float *w, const float *scale,
                 const float *shift, void *y, int32_t ldy, int32_t N, int32_t Ho, int32_t Wo, int32_t C, kd_stream_t stream);
size_t kd_canny_workspace(int32_t N, int32_t H, int32_t W);
int kd_canny(const float *x, int32_t N, int32_t H, int32_t W, int32_t low, int32_t high, int32_t sweeps, float *out,
             int32_t *changed, void *workspace, size_t workspace_bytes, kd_stream_t stream);
int kd_canny_continue(int32_t N, int32_t H, int32_t W, int32_t sweeps, float *out, int32_t *changed, void *workspace,
                      size_t workspace_bytes, kd_stream_t stream);

/* -------------------------------------------------------------------- losses
 * Each writes the scalar loss (fp32, device) and, when grad != NULL, the
 * gradient w.r.t. `s` in one pass.  Views are (N, C, P) with element strides
 * (sN, sC, sP) so NCHW (reference layout) and NHWC (engine layout) are both
 * accepted; dtype per operand.  partial: workspace of kd_loss_workspace() bytes.
 *
 * kd_kldiv: KLDivergenceLoss.forward, losses/KLDiv.py:19-23
 *   loss = kl_div(log_softmax(s/T,1), softmax(t/T,1), 'mean') * T^2 * C
 *   grad = T/(N*P) * (softmax(s/T) - softmax(t/T))
 * kd_hint_mse: MSELoss.forward, losses/MSELoss.py:14-16
 *   loss = mean((s-t)^2) * num_classes ; grad = 2*num_classes*(s-t)/numel
 * kd_weighted_hint_mse: WeightedHintMSELoss.forward, losses/WeightedHintMSELoss.py:12-16
 *   w: (C) if w_per_sample == 0 else (N,C), fp32.
 * kd_ce2d: CrossEntropyLoss2d.forward, losses/CrossEntropy.py:10-14 (logged metric);
 *   target int64 (N,P), ignore_index excluded from the mean.
 */
typedef struct kd_view3 {
    const void *ptr;
    int32_t dtype;
    int64_t sN, sC, sP;
} kd_view3;
typedef struct kd_mview3 {
    void *ptr;
    int32_t dtype;
    int64_t sN, sC, sP;
} kd_mview3;

size_t kd_loss_workspace(int32_t N, int32_t C, int64_t P);
int kd_kldiv(const kd_view3 *s, const kd_view3 *t, float temperature, int32_t N, int32_t C, int64_t P,
             float *loss, const kd_mview3 *grad, float grad_scale, void *workspace, size_t workspace_bytes,
             kd_stream_t stream);
int kd_hint_mse(const kd_view3 *s, const kd_view3 *t, float num_classes, int32_t N, int32_t C, int64_t P,
                float *loss, const kd_mview3 *grad, float grad_scale, void *workspace, size_t workspace_bytes,
                kd_stream_t stream);
int kd_weighted_hint_mse(const kd_view3 *s, const kd_view3 *t, const float *w, int32_t w_per_sample,
                         int32_t N, int32_t C, int64_t P, float *loss, const kd_mview3 *grad, float grad_scale,
                         void *workspace, size_t workspace_bytes, kd_stream_t stream);
int kd_ce2d(const kd_view3 *x, const int64_t *target, int32_t ignore_index, int32_t N, int32_t C, int64_t P,
            float *loss, void *workspace, size_t workspace_bytes, kd_stream_t stream);

/* The two logged logit losses straight from the LOW-RESOLUTION logits: x_lo / s_lo / t_lo are the classifier's dense fp32
 * (N,h,w,C) NHWC outputs; the value equals kd_ce2d / kd_kldiv on F.interpolate(x, (H,W), mode='bilinear', align_corners)
 * (models/deeplabv3/deeplabv3.py:160-162 followed by losses/CrossEntropy.py:10-14 / losses/KLDiv.py:19-23), but the two
 * full-resolution fp32 tensors are never written or read: a pixel's C logits are interpolated in registers (same expression
 * tree as kd_upsample_bilinear).  Forward only (the logged metrics of trainer/layerwise_trainer.py:222-227); C <= 48 / C <= 24;
 * KD_ERR_UNSUPPORTED when the resampling ratio puts more than 160 source columns under 256 output pixels (materialise then).
 * target int64 (N,H,W); workspace as kd_loss_workspace(N, C, H*W). */
int kd_ce2d_up(const float *x_lo, const int64_t *target, int32_t ignore_index, int32_t N, int32_t h, int32_t w, int32_t C,
               int32_t H, int32_t W, int32_t align_corners, float *loss, void *workspace, size_t workspace_bytes,
               kd_stream_t stream);
int kd_kldiv_up(const float *s_lo, const float *t_lo, float temperature, int32_t N, int32_t h, int32_t w, int32_t C, int32_t H,
                int32_t W, int32_t align_corners, float *loss, void *workspace, size_t workspace_bytes, kd_stream_t stream);

/* gradient of kd_ce2d w.r.t. x (needed when the supervised loss is back-propagated: trainer/taylor_prune_trainer.py:204-206,
 * or any loss = supervised + kd + hint mix): grad[n,c,p] = grad_scale * (softmax_c(x[n,:,p]) - [c == target[n,p]]) / #valid,
 * zero for ignored pixels. */
int kd_ce2d_grad(const kd_view3 *x, const int64_t *target, int32_t ignore_index, int32_t N, int32_t C, int64_t P,
                 const kd_mview3 *grad, float grad_scale, void *workspace, size_t workspace_bytes, kd_stream_t stream);

/* CrossEntropyLoss2d(weight, size_average) (losses/CrossEntropy.py:5-14: nn.NLLLoss(weight, size_average, ignore_index) on
 * log_softmax): class_weight fp32 (C) on the device or NULL; the mean is weighted, sum_i w[y_i] nll_i / sum_i w[y_i]
 * (sum_reduction == 0, size_average=True) or the plain weighted sum (sum_reduction != 0, size_average=False); ignored
 * pixels contribute to neither.  kd_ce2d / kd_ce2d_grad are these with class_weight == NULL, sum_reduction == 0. */
int kd_ce2d_weighted(const kd_view3 *x, const int64_t *target, const float *class_weight, int32_t sum_reduction, int32_t ignore_index,
                     int32_t N, int32_t C, int64_t P, float *loss, void *workspace, size_t workspace_bytes, kd_stream_t stream);
int kd_ce2d_weighted_grad(const kd_view3 *x, const int64_t *target, const float *class_weight, int32_t sum_reduction, int32_t ignore_index,
                          int32_t N, int32_t C, int64_t P, const kd_mview3 *grad, float grad_scale, void *workspace,
                          size_t workspace_bytes, kd_stream_t stream);

/* CityscapesMetricTracker.update / confusion_for_batch (utils/util.py:108-128), the logged train mIoU, without the
 * reference's two full-logit D2H copies per step (trainer/layerwise_trainer.py:249-250):
 *   for every pixel with 0 <= target < C:  conf[target][argmax_c x(n,c,p)] += 1
 * (labels == ignore_index, which the reference rewrites to C before masking, fall outside the range; argmax = first
 * index of the maximum like torch.argmax).  conf: int64 (C, C) on the device, row = label, column = prediction;
 * accumulate == 0 zeroes it first.  Integer result, exact and order-independent.  C <= 64. */
int kd_confusion(const kd_view3 *x, const int64_t *target, int32_t N, int32_t C, int64_t P,
                 int64_t *conf, int32_t accumulate, kd_stream_t stream);

/* x *= *scale with the test `*scale == 1` made on the device (then nothing is read or written): the backward of the fused
 * loss Functions, whose stored gradient is multiplied by autograd's upstream d(total)/d(loss) -- exactly 1 for
 * `loss = hint_loss` (trainer/layerwise_trainer.py:229-235).  scale: one fp32 on the device. */
int kd_scale_by_device_scalar(void *x, int32_t dtype, int64_t n, const float *scale, kd_stream_t stream);

/* ----------------------------------------------------------------- optimizer
 * RAdam.step for one tensor, utils/optim/radam.py:30-98 (fp32 params/state).
 * `step` is the per-tensor step count after the increment (radam.py:62). */
int kd_radam_step(float *p, const float *g, float *exp_avg, float *exp_avg_sq, int64_t n, int32_t step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, kd_stream_t stream);
/* The same update for many tensors in one launch (mode B steps ~150 tensors): per-tensor hyper-parameters and
 * step counts, exactly the arithmetic of kd_radam_step.  `ts` is a HOST array; nothing is copied to the device besides the
 * kernel arguments. */
typedef struct kd_radam_tensor {
    float *p; const float *g; float *exp_avg; float *exp_avg_sq;
    int64_t n; int32_t step;
    float lr, beta1, beta2, eps, weight_decay;
} kd_radam_tensor;
int kd_radam_step_multi(const kd_radam_tensor *ts, int32_t count, kd_stream_t stream);

/* ------------------------------------------------- Gated-SCNN shape stream, backward (models/gscnn/gscnn.py:269-314 under autograd)
 * The forward kernels (kd_gated_conv, kd_pointwise_small, kd_edge_attention, kd_edge_aspp) fuse per-pixel algebra; loss.backward()
 * (trainer/layerwise_trainer.py:235) through them -- `aspp` hints, loss terms on the logits, trainable shape-stream parameters --
 * is built from these general pieces (fp32 arithmetic, either storage type, deterministic reductions):
 *
 * kd_small_linear: y[p][co] (+)= bias[co] + sum_ci w[co*Cin + ci] * x[p][ci], optional ReLU; 1 <= Cin, Cout <= 72.  The forward of any
 *   1x1 map of the stream (nn.Conv2d(C, C', 1): d1..d3, gscnn.py:232-236; the two 1x1 convs inside GatedSpatialConv2d._gate_conv,
 *   gate_spatial_conv.py:36-43) and, given the transposed matrix, its input gradient.
 * kd_small_wgrad: dw[cb*Ca + ca] = sum_p b[p][cb] * a[p][ca], db[cb] = sum_p b[p][cb] (db may be NULL): weight / bias gradient of the
 *   same maps with a = the map's input, b = the gradient of its output.
 * kd_gate_mix_bwd: GatedSpatialConv2d's tail `input_features * (alphas + 1)` (gate_spatial_conv.py:58-59) with alphas = sigmoid(a):
 *   v = feat * (sigmoid(a) + 1) (if v), gfeat = gv * (sigmoid(a) + 1) (if gfeat), ga = (sum_c gv * feat) * sigmoid'(a) (if ga).
 * kd_edge_attention_bwd: backward of kd_edge_attention (gscnn.py:308-314): g_t = dL/d(cw pre-activation), g_s = dL/d(fuse
 *   pre-activation), eo_canny[p] = (sigmoid(fuse . cs), canny) -- the cw conv's input, for its weight gradient.
 * kd_rank1_add: y[p][c] (+)= g[p] * w[c]: input gradient of a C -> 1 1x1 conv (the dsn3 / dsn4 / dsn7 side outputs, gscnn.py:272-277). */
int kd_small_linear(const void *x, int32_t x_dtype, int32_t ldx, int32_t Cin, const float *w, const float *bias, void *y,
                    int32_t y_dtype, int32_t ldy, int32_t Cout, int64_t npix, int32_t accumulate, int32_t relu,
                    const float *mask /* fp32 (npix, ldm) or NULL: y = mask > 0 ? y : 0 */, int32_t ldm, kd_stream_t stream);
size_t kd_small_wgrad_workspace(int32_t Ca, int32_t Cb, int64_t npix);
int kd_small_wgrad(const void *a, int32_t a_dtype, int32_t lda, int32_t Ca, const void *b, int32_t b_dtype, int32_t ldb, int32_t Cb,
                   int64_t npix, float *dw, float *db, int32_t accumulate, void *workspace, size_t workspace_bytes, kd_stream_t stream);
int kd_gate_mix_bwd(const void *feat, int32_t feat_dtype, int32_t ldf, const float *a, const float *gv, int32_t ldgv, float *gfeat,
                    int32_t ldgf, float *ga, float *v, int32_t ldv, int32_t C, int64_t npix, kd_stream_t stream);
int kd_edge_attention_bwd(int32_t dtype, const void *cs, int32_t ldc, const float *canny, const float *weights, const float *g_acts,
                          float *g_t, float *g_s, float *eo_canny, int64_t npix, kd_stream_t stream);
int kd_rank1_add(int32_t dtype, void *y, int32_t ldy, const float *g, const float *w, int32_t C, int64_t npix, int32_t accumulate,
                 kd_stream_t stream);

/* ----------------------------------------------------------------- diagnostics (host side only; no launch changes)
 * Kernel-selection log: the dispatchers behind kd_conv2d_fwd / kd_conv2d_wgrad / kd_pw_wgrad / kd_dwconv_* /
 * kd_stem_conv* pick one of several device kernels by shape, dtype and epilogue.  The reference has
 * nothing to mirror here -- its dispatch is cuDNN's, behind torch.nn.functional.conv2d (models/encoders/wider_resnet.py:
 * 124-167) -- but a parity test is only worth what it covers, so tests assert the kernel a case reached.
 *   kd_debug_kernel_log_enable(1) clears the counters and starts counting, (0) stops;
 *   kd_debug_kernel_log_read writes "name\tcount\n" lines (NUL-terminated, truncated to `bytes`) and returns the length
 *   needed; kd_debug_last_kernel is the name the calling thread's last dispatch picked ("" before any). */
int kd_debug_kernel_log_enable(int32_t on);
int64_t kd_debug_kernel_log_read(char *buf, size_t bytes);
const char *kd_debug_last_kernel(void);

#ifdef __cplusplus
}
#endif
#endif /* KDCC_H */
