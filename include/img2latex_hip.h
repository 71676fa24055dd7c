/*
 * img2latex_hip.h -- C ABI of libimg2latex_hip.so (MI355X / gfx950, HIP).
 *
 * The reference (Jeremy-Cleland/hmer-img2latex) has no FFI: its hot path is
 * the Python class surface of img2latex.model calling ATen through torch.nn.
 * Each entry point below replaces the ATen work of the cited reference lines;
 * hmer-img2latex_amd/img2latex_amd/model/ binds them with ctypes and
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions:
 *   - extern "C", plain pointers and sizes; every pointer is a DEVICE pointer
 *     into a caller-owned, contiguous, 16-byte-aligned fp32/int32 buffer unless
 *     the parameter is documented as a HOST array of device pointers.
 *   - explicit stream (a hipStream_t passed as void*; NULL = default stream);
 *     every call only ENQUEUES work, it never synchronises or allocates.
 *   - workspace is caller-provided; its size comes from *_workspace_bytes().
 *   - return value: I2L_OK (0) or a negative I2L_ERR_* code; never throws,
 *     never aborts; no global state (thread-compatible): the library reads no
 *     environment variable and keeps nothing between calls -- where an entry
 *     point has more than one kernel behind it, the choice is the explicit
 *     `flags` argument (I2L_FLAG_*, 0 = automatic); streams and events used
 *     beside `stream` live in caller-owned objects (i2l_lanes).  The only
 *     file-scope data are function-local "attribute already set on device d"
 *     bit masks (an idempotent cache of hipFuncSetAttribute).
 *   - tensors are row-major with the reference's (PyTorch) shapes.
 */
#ifndef IMG2LATEX_HIP_H
#define IMG2LATEX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I2L_OK 0
#define I2L_ERR_ARG (-1)          /* null pointer / non-positive dimension           */
#define I2L_ERR_UNSUPPORTED (-2)  /* dimension outside what the kernels are built for */
#define I2L_ERR_WORKSPACE (-3)    /* workspace missing or too small                   */
#define I2L_ERR_LAUNCH (-4)       /* HIP reported a launch error                      */

#define I2L_MAX_LSTM_LAYERS 4
#define I2L_MAX_BEAM 8

typedef void* i2l_stream_t;

/* Kernel-selection flags (bit-or; 0 = automatic = the fastest kernel built for the shape).  Every variant computes
 * the same function; tests run the variants against each other and callers use them to fall back. */
#define I2L_FLAG_EXACT_FP32 0x1        /* conv / linear / training GEMMs: exact fp32 products (fp32 MFMA / VALU fmaf
                                          chains) instead of the 3 x bf16 split on the bf16 matrix cores             */
#define I2L_FLAG_NO_GROUP 0x2          /* beam search, training recurrences: one workgroup per image / row instead of
                                          the grouped kernels (4 co-resident workgroups exchanging through L2)       */
#define I2L_FLAG_RESNET_NO_RING 0x4    /* i2l_conv_bn_act_bf16_fwd: single-buffered GEMM instead of the LDS ring      */
#define I2L_FLAG_RESNET_IM2COL_STEM 0x8 /* i2l_conv_bn_act_bf16_fwd: im2col image + GEMM instead of the fused stem    */
#define I2L_FLAG_WEIGHTS_PACKED 0x10    /* i2l_conv3x3_relu_pool2_fwd without argmax_out: the workspace still holds the packed
                                          (3 x bf16) filter image an EARLIER call wrote for the SAME weights -- the caller
                                          kept the buffer and the weights have not changed -- so the weight-only packing
                                          launch is skipped; ignored by the kernels that do not pack                 */
#define I2L_FLAG_AGENT_SCOPE_EXCHANGE 0x20 /* grouped kernels (greedy, beam, training recurrences): every exchange store at
                                          agent scope (sc1, write-through) -- the HSA-memory-model-conformant flavour --
                                          even when the group's members share an XCD and the faster L2-local
                                          (workgroup-scope) stores would be used; same results, ~0.8 us per step slower */
#define I2L_FLAG_TEST_SHORT_TIMEOUT 0x40 /* grouped kernels, TEST hook: every poll limit 2 ms instead of (10 ms + 50 us per
                                          step) for a workgroup's first poll and 3 s for the later ones               */
#define I2L_FLAG_TEST_DROP_MEMBER 0x80 /* grouped kernels, TEST hook: member 3 of every group exits at once, so its peers
                                          time out: exercises the failure path (ids -3 / len -3 / NaN, host fallback)  */
#define I2L_FLAG_DECODE_GROUP8 0x1000  /* i2l_greedy_decode_ex (ids only: no logits_out, no forced tokens): the grouped kernel
                                          with EIGHT members x EIGHT rows per group -- one wave per SIMD and ~80 KB of LDS
                                          per CU instead of two waves and 140 KB, so that a conv workgroup of the NEXT
                                          batch's encoder fits beside it; same ids                                   */
#define I2L_FLAG_DECODE_GROUP16 0x8000 /* i2l_greedy_decode_ex (ids only): the grouped kernel with SIXTEEN members x SIXTEEN rows per
                                          group and the per-step products (LSTM gates, logits) on the matrix cores as split-bf16
                                          v_mfma_f32_16x16x32_bf16 bursts (fp32-grade: 3 bf16 pieces per operand, 6 partial
                                          products) -- ~250 registers and 49 KB of LDS per CU, and a burst that conv waves on
                                          the same CU barely stretch; same ids up to fp32 near-ties.  Wins over _GROUP8 if both set */
#define I2L_FLAG_CONV_COL_READY 0x10000 /* i2l_conv_f32_bwd: the workspace is the one the matching i2l_conv_f32_fwd call used and
                                          nothing has written to it since, so its head still holds the column image of x: skip
                                          the im2col launch (the caller keeps one workspace per unit on its tape)               */
#define I2L_FLAG_RESNET_WIDE_TILES 0x40000 /* i2l_conv_bn_act_bf16_fwd: 128-column tiles even where 64-column tiles balance the CUs better (A/B) */
#define I2L_FLAG_RESNET_NO_PATCH 0x20000 /* i2l_conv_bn_act_bf16_fwd, 3x3 / stride 1 / pad 1: the implicit-GEMM ring kernel instead of
                                          the kernel that stages the input patch in LDS (A/B switch) */
#define I2L_FLAG_DECODE_REGION_CLEARED 0x2000000 /* i2l_greedy_decode_ex, grouped kernels: the caller has zeroed the workspace's group region
                                             (i2l_decoder_group_status_offset / _region_bytes) since its last use, ordered before this call: the
                                             launch skips its own memset (GreedyPipeline clears it on the encoder stream, off the decode queue) */
#define I2L_FLAG_CONV_NO_SPARSE_WGRAD 0x4000 /* i2l_conv3x3_relu_pool2_bwd with dx == NULL, Cin <= 3, Cout % 32 == 0: the
                                          implicit-im2col GEMM instead of the sparse first-block kernel (A/B, tests)   */
#define I2L_FLAG_RESNET_RING_DEPTH(n) (((n) & 0xF) << 8)   /* force the ring depth (2..4; 5 = four stages of 32-deep K tiles); 0 = automatic */
#define I2L_FLAG_RESNET_PATCH_SHAPE(n) (((n) & 0xF) << 20) /* 3x3 / stride 1 layers: force tile shape n (1..5: 160x128, 96x128,
                                                              128x128, 320x64, 256x64 pixels x channels); 0 = by balance */

int i2l_version(void);
const char* i2l_error_string(int code);
/* Side lanes of the training backward pass (caller-owned; the library itself keeps no state, see the header comment).
 * i2l_decoder_train_bwd, i2l_linear_bias_act_bwd and i2l_conv3x3_relu_pool2_bwd take an `i2l_lanes* lanes` argument.
 * NULL: everything is enqueued on `stream`.  Otherwise the WEIGHT gradients (and bias sums) -- which nothing in the
 * backward chain consumes -- are enqueued on the caller's side streams (lane 0: decoder + Linear, lane 1: conv blocks;
 * with one lane both share it), forked from `stream` by an event at the point where their inputs are complete, beside
 * the data-gradient chain that stays on `stream`.  The caller must call i2l_lanes_join(lanes, stream) before anything
 * reads those gradients and keep every buffer passed to the calls (workspace, activations, dy) alive until then.  Same
 * kernels, same sums: results identical to lanes == NULL.  The reference has no counterpart: its autograd engine orders
 * `loss.backward()` (trainer.py:337) before `optimizer.step()` (trainer.py:343) by itself.
 *   create:  `streams` = n (1..I2L_MAX_LANES) non-default streams of the CURRENT device, owned by the caller and alive as
 *            long as the object; the object adds the events it records on them
 *   destroy: releases the events (not the streams); NULL is accepted
 * One object serves one host thread at a time (like a stream's enqueue order, its fork / join events are not re-entrant);
 * two threads training on one device create one each. */
#define I2L_MAX_LANES 4
typedef struct i2l_lanes i2l_lanes;
int i2l_lanes_create(const i2l_stream_t* streams, int n, i2l_lanes** out);
int i2l_lanes_destroy(i2l_lanes* lanes);
int i2l_lanes_join(i2l_lanes* lanes, i2l_stream_t stream);
/* Enqueues a one-wave kernel on `stream` that returns once (int32)(*flag - value) >= 0 -- `flag` a caller-owned device word
 * -- or after timeout_us (<= 100000) microseconds, whichever comes first.  GreedyPipeline's dependency between decode(i)
 * and encoder(i + 1): i2l_greedy_decode_ex publishes `resident_value` to `resident_flag` once the grouped decode kernel's
 * workgroups are all resident, and the encoder stream waits for it here, so that the conv workgroups never get onto the
 * compute units before the decode's (no counterpart in the reference, which runs one batch at a time,
 * predictor.py:205-381).  The wait is bounded, so a signal that never comes costs time, never a hang. */
int i2l_stream_wait_value32(const uint32_t* flag, uint32_t value, float timeout_us, i2l_stream_t stream);

/* ------------------------------------------------------------------------
 * Encoder (reference img2latex/model/encoder.py)
 * ---------------------------------------------------------------------- */

/* One CNN block: y = maxpool2x2(relu(conv3x3_pad1(x, w) + bias)), floor pooling.
 * Replaces nn.Conv2d + nn.ReLU + nn.MaxPool2d, encoder.py:78-95 executed at :122.
 * x (B,Cin,H,W)  w (Cout,Cin,3,3)  bias (Cout)  y (B,Cout,H/2,W/2), NCHW fp32.
 * Arithmetic: with Cin <= 3 or Cin % 16 == 0, Cout % 32/64 == 0 the products
 * run on the bf16 matrix cores with every fp32 operand split exactly into three bf16 pieces (six partial
 * products, fp32 accumulation): fp32-grade results (~2^-24 relative per product), not bit-identical to an fmaf
 * chain.  flags & I2L_FLAG_EXACT_FP32 and other shapes use the exact fp32 kernels.  With argmax_out (training forward)
 * the split kernels also list every pooling window whose arg max or ReLU gate is decided by less than 2^-13 of its
 * magnitude, and a fix-up kernel re-evaluates those windows with fp32 FMAs, so the discrete decisions the backward
 * pass branches on are an fp32 computation's.  The same split applies to i2l_linear_bias_act_fwd for K >= 2048.
 * Limits of the split arithmetic (tests/test_hip_parity.py::test_bf16x3_*): the error class is that of an fp32 fmaf
 * chain, |err| <= ~2^-22 (sum |x||w| + |b|), for operands of magnitude 0 or >= 2^-110 (below that the low split
 * pieces fall into bf16's subnormal range and the class degrades towards bf16); inputs must be finite -- a
 * non-finite input affects only the outputs whose receptive field holds it, but their value is unspecified (NaN
 * where the fp32 product would be +-Inf; ReLU's max may drop a NaN). */
size_t i2l_conv_workspace_bytes(int Cin, int Cout);   /* packed-weight scratch; 0 when none is needed */
/* argmax_out: NULL, or (B,Cout,H/2,W/2) uint8 receiving the position 2*dy+dx of each pooling
 * window's maximum (first maximum wins, as ATen) -- what the backward pass needs. */
int i2l_conv3x3_relu_pool2_fwd(const float* x, const float* w, const float* bias, float* y,
                               uint8_t* argmax_out, int B, int Cin, int H, int W, int Cout, void* workspace,
                               size_t workspace_bytes, int flags, i2l_stream_t stream);

/* Backward of one CNN block (autograd of encoder.py:78-95 under loss.backward(), trainer.py:337):
 * given dy (B,Cout,H/2,W/2), the block's input x, output y and pooling argmax, computes
 * dw (Cout,Cin,3,3), db (Cout) and, unless dx is NULL, dx (B,Cin,H,W).  Gradients are overwritten. */
size_t i2l_conv_bwd_workspace_bytes(int B, int Cin, int H, int W, int Cout);
int i2l_conv3x3_relu_pool2_bwd(const float* x, const float* w, const float* y, const uint8_t* argmax,
                               const float* dy, float* dx, float* dw, float* db, int B, int Cin, int H,
                               int W, int Cout, void* workspace, size_t workspace_bytes, int flags,
                               i2l_lanes* lanes, i2l_stream_t stream);

/* y = act(x @ w^T + bias): nn.Flatten + nn.Linear + nn.ReLU, encoder.py:105-107,125-127
 * (also nn.Linear(Hd->V), decoder.py:90).  x (M,K)  w (N,K)  bias (N) or NULL  y (M,N).
 * relu != 0 applies ReLU.  Split-K partials live in the workspace. */
size_t i2l_linear_workspace_bytes(int M, int K, int N);
int i2l_linear_bias_act_fwd(const float* x, const float* w, const float* bias, float* y,
                            int M, int K, int N, int relu, void* workspace, size_t workspace_bytes,
                            int flags, i2l_stream_t stream);
/* Backward of the above: dy (M,N) -> dx (M,K) (or NULL), dw (N,K), db (N) (or NULL: a layer without bias, e.g. the
 * ResNet convolutions that reach this entry through an im2col image); with relu != 0 the gradient is first masked by
 * y > 0.  Gradients are overwritten. */
size_t i2l_linear_bwd_workspace_bytes(int M, int K, int N);
int i2l_linear_bias_act_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx,
                            float* dw, float* db, int M, int K, int N, int relu, void* workspace,
                            size_t workspace_bytes, int flags, i2l_lanes* lanes, i2l_stream_t stream);

/* ResNet encoder building blocks (reference encoder.py:132-249: torchvision ResNet trunk at :242),
 * inference, bf16 on the matrix cores with fp32 accumulation.  Activations are NHWC bf16 (void*).
 *   y = act( BatchNorm_eval(conv(x, w)) + residual )    w (Cout,Cin,kh,kw) fp32, bn_* (Cout) fp32
 * i2l_conv_bn_bf16_pack is the weight-only preparation (bf16 filter image, BatchNorm folded into
 * scale/bias); its result stays valid until a weight or BatchNorm statistic changes.
 * x is NHWC bf16 (B,H,W,Cin), or the NCHW fp32 image batch when x_is_nchw_f32 != 0 (the stem);
 * residual is NHWC bf16 (B,Ho,Wo,Cout) or NULL; y NHWC bf16 (B,Ho,Wo,Cout).
 * Kernel choice: K = kh*kw*Cin a multiple of 64 with NHWC bf16 input runs the ring-buffered direct-to-LDS GEMM
 * (I2L_FLAG_RESNET_NO_RING: the single-buffered kernel); the 7x7 / stride 2 / pad 3 / 3 -> 64 stem on fp32 images runs
 * the fused stem kernel and needs no workspace (I2L_FLAG_RESNET_IM2COL_STEM: im2col image + GEMM); every other
 * shape goes through an im2col image in the workspace.  All paths compute the same bf16 x bf16 -> fp32 sums. */
size_t i2l_conv_bf16_packed_bytes(int Cout, int Cin, int kh, int kw);
int i2l_conv_bn_bf16_pack(const float* w, const float* bn_weight, const float* bn_bias, const float* bn_mean,
                          const float* bn_var, float bn_eps, void* packed, size_t packed_bytes, int Cout,
                          int Cin, int kh, int kw, i2l_stream_t stream);
size_t i2l_conv_bf16_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                                     int flags);
int i2l_conv_bn_act_bf16_fwd(const void* x, int x_is_nchw_f32, const void* packed, const void* residual,
                             void* y, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                             int pad, int relu, void* workspace, size_t workspace_bytes, int flags,
                             i2l_stream_t stream);
/* nn.MaxPool2d(3, stride 2, padding 1) on NHWC bf16: (B,H,W,C) -> (B,(H-1)/2+1,(W-1)/2+1,C). */
/* Bottleneck tail + the next block's head in ONE launch (r04; encoder.py:185-249, torchvision Bottleneck.forward's
 * `out = relu(bn3(conv3(out)) + identity)` followed by the next block's `relu(bn1(conv1(x)))`):
 *   y[p][256] = relu(bn3(conv3(o2[p][64])) + identity[p][256]),  z[p][n2] = relu(bn1'(conv1'(y[p])))      NHWC bf16
 * The 128-position tile of y is completed in LDS, written out once (the next block's identity) and multiplied by conv1'
 * from LDS, so y is not read back: layer1 of the bottleneck ResNets is bound by exactly those bytes.  `packed3` /
 * `packed1_next` = i2l_conv_bn_bf16_pack of the two 1x1 convs.  Same roundings as the two launches it replaces (bit-identical
 * y and z).  Shapes: c_mid 64, c_out 256, n2 64 or 128; otherwise I2L_ERR_UNSUPPORTED (the caller makes two launches). */
int i2l_bottleneck_join_bf16_fwd(const void* o2, const void* packed3, const void* identity, void* y, const void* packed1_next,
                                 void* z, long positions, int c_mid, int c_out, int n2, i2l_stream_t stream);
int i2l_maxpool3x3s2_bf16_fwd(const void* x, void* y, int B, int H, int W, int C, i2l_stream_t stream);
/* nn.AdaptiveAvgPool2d(1) + Flatten: NHWC bf16 (B,H,W,C) -> fp32 (B,C). */
int i2l_global_avgpool_bf16_fwd(const void* x, float* y, int B, int H, int W, int C, i2l_stream_t stream);

/* ResNet encoder in TRAINING mode (encoder.py:185-249 under model.train(): every BatchNorm2d of the torchvision trunk
 * normalises with batch statistics and updates its running statistics -- the frozen ones too, freeze_backbone only
 * clears requires_grad, :201-210 -- and layer4 + the Linear, or with freeze_backbone=False every layer, get gradients).
 * fp32 GRADE throughout, as the reference's fp32 branch (trainer.py:334-343): activations, raw conv outputs z and
 * gradients are NHWC fp32 = row-major (M = B*H*W, C) matrices; the convolutions are GEMMs on the split-bf16
 * matrix-core kernel (3 bf16 pieces per operand, fp32 accumulation; I2L_FLAG_EXACT_FP32: fp32 MFMA).
 *
 * Convolution without bias (torchvision's convs have none): x_kind 1 = NHWC fp32 (B,H,W,Cin), 2 = NCHW fp32 (the image
 * batch of the stem); w (Cout,Cin,kh,kw) fp32 as stored by nn.Conv2d; z / dz (B,Ho,Wo,Cout) NHWC fp32.
 *   fwd: z = conv(x, w)
 *   bwd: dw (Cout,Cin,kh,kw) = d/dw, dx (B,H,W,Cin) NHWC fp32 = d/dx; either may be NULL (dx needs x_kind 1).
 * The workspace holds the fp32 column image (none for 1x1 / stride 1) at its head, for dx its gradient, and the GEMM slabs;
 * backward_dx != 0 sizes it for a call with dx (a workspace of that size also serves the forward call, and handing the
 * SAME one to the backward call with I2L_FLAG_CONV_COL_READY saves the second im2col). */
size_t i2l_conv_f32_workspace_bytes(int x_kind, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                                    int pad, int backward_dx);
int i2l_conv_f32_fwd(const float* x, int x_kind, const float* w, float* z, int B, int H, int W, int Cin, int Cout,
                     int kh, int kw, int stride, int pad, void* workspace, size_t workspace_bytes, int flags,
                     i2l_stream_t stream);
int i2l_conv_f32_bwd(const float* x, int x_kind, const float* w, const float* dz, float* dx, float* dw, int B, int H,
                     int W, int Cin, int Cout, int kh, int kw, int stride, int pad, void* workspace,
                     size_t workspace_bytes, int flags, i2l_stream_t stream);
/* nn.BatchNorm2d in training mode on a row-major (M, C) fp32 matrix z, C % 8 == 0 (+ residual add, + ReLU):
 *   fwd: mean and biased variance from ONE pass over z - c (c = row 0 of z: a shift of the order of the mean, so that no
 *        digits cancel; slab sums combined in double); running_* (may both be NULL) <- (1 - momentum) * running +
 *        momentum * (mean, UNBIASED variance);
 *        y = act(gamma * (z - mean) * invstd + beta + residual); save_mean / save_invstd (C)
 *   bwd: g = dy masked by y_relu > 0 (y_relu NULL: no ReLU);  dz = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat));
 *        dgamma = sum g * xhat, dbeta = sum g (either may be NULL); dres (may be NULL) = g (or += g): the gradient of
 *        the residual branch. */
size_t i2l_bn_train_workspace_bytes(int64_t M, int C);
int i2l_bn_train_fwd_f32(const float* z, const float* residual, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, int relu, float* y,
                         float* save_mean, float* save_invstd, int64_t M, int C, void* workspace,
                         size_t workspace_bytes, i2l_stream_t stream);
int i2l_bn_train_bwd_f32(const float* dy, const float* y_relu, const float* z, const float* gamma, const float* save_mean,
                         const float* save_invstd, float* dz, float* dgamma, float* dbeta, float* dres,
                         int dres_accumulate, int64_t M, int C, void* workspace, size_t workspace_bytes,
                         i2l_stream_t stream);
/* The two halves of the convolution entry points above, on their own: col (B*Ho*Wo, Cin*kh*kw) fp32 with the column
 * order (ci, ky, kx) of the weight tensor, and the deterministic gather of a column-image gradient back to
 * dx (B,H,W,Cin) NHWC fp32.  x_kind: 0 = NHWC bf16, 1 = NHWC fp32, 2 = NCHW fp32. */
int i2l_im2col_f32(const void* x, int x_kind, int B, int H, int W, int C, int kh, int kw, int stride, int pad,
                   float* col, i2l_stream_t stream);
int i2l_col2im_f32(const float* dcol, int B, int H, int W, int C, int kh, int kw, int stride, int pad, float* dx,
                   int accumulate, i2l_stream_t stream);
/* nn.MaxPool2d(3, 2, 1) on NHWC fp32 (C % 4 == 0) and its backward (x = the forward input; the first maximum of a
 * window takes its gradient, as ATen); nn.AdaptiveAvgPool2d(1) + Flatten (B,H,W,C) -> (B,C) and its backward. */
int i2l_maxpool3x3s2_f32_fwd(const float* x, float* y, int B, int H, int W, int C, i2l_stream_t stream);
int i2l_maxpool3x3s2_f32_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, i2l_stream_t stream);
int i2l_global_avgpool_f32_fwd(const float* x, float* y, int B, int H, int W, int C, i2l_stream_t stream);
int i2l_global_avgpool_bwd_f32(const float* dfeat, float* dx, int B, int H, int W, int C, i2l_stream_t stream);

/* ------------------------------------------------------------------------
 * Decoder (reference img2latex/model/decoder.py, seq2seq.py, predictor.py)
 * ---------------------------------------------------------------------- */

/* Weights of LSTMDecoder as the reference's state_dict holds them (decoder.py:69-90).
 * w_ih/w_hh/b_ih/b_hh are HOST arrays of `layers` device pointers:
 *   w_ih[0] (4H, 2E), w_ih[l>0] (4H, H), w_hh[l] (4H, H), b_ih[l], b_hh[l] (4H); gate order i,f,g,o. */
typedef struct i2l_decoder_weights {
    const float* embedding;      /* (V, E)  decoder.embedding.weight     */
    const float* const* w_ih;    /* decoder.lstm.weight_ih_l{l}          */
    const float* const* w_hh;    /* decoder.lstm.weight_hh_l{l}          */
    const float* const* b_ih;    /* decoder.lstm.bias_ih_l{l}            */
    const float* const* b_hh;    /* decoder.lstm.bias_hh_l{l}            */
    const float* w_out;          /* (V, H)  decoder.output_layer.weight  */
    const float* b_out;          /* (V)     decoder.output_layer.bias    */
    int vocab, embed, hidden, layers;
} i2l_decoder_weights;

/* Bytes of workspace i2l_decoder_prepare() fills for a batch of `rows` encoder rows. */
size_t i2l_decoder_workspace_bytes(int rows, int vocab, int embed, int hidden, int layers);

/* Re-lay the decoder weights for the step kernels (gate-interleaved transposes),
 * fold the token-side half of W_ih_l0 into a (V,4H) table, and compute the per-row
 * constant  enc @ W_ih_l0[:, E:]^T + b_ih_l0 + b_hh_l0  (the half of the layer-0 gate
 * GEMM that does not change over the decode loop; cat([emb, enc]) at decoder.py:228,274).
 * With the reference's Attention over a length-1 source the context equals enc
 * bit-for-bit (decoder.py:338-341, softmax over one element), so `enc` serves both the
 * attention and the no-attention decoder.  enc (rows, E).  `what` selects the part to
 * (re)build: the weight images must be rebuilt whenever a weight changes, the row part
 * whenever enc changes; everything it writes is in `workspace`. */
#define I2L_PREP_WEIGHTS 1   /* transposes + token table: depends on the weights only          */
#define I2L_PREP_ROWS 2      /* per-row constant: depends on enc (and W_ih_l0, biases)        */
#define I2L_PREP_ALL 3
int i2l_decoder_prepare(const i2l_decoder_weights* w, const float* enc, int rows, int what,
                        void* workspace, size_t workspace_bytes, i2l_stream_t stream);

/* Stop rules of the two greedy loops of the reference. */
#define I2L_STOP_NONE 0    /* run all `steps`; Seq2SeqModel._greedy_search stops only when ALL
                              rows emit END in one step (seq2seq.py:220) -- the host finds that
                              step in the returned ids                                          */
#define I2L_STOP_STICKY 1  /* Predictor.predict_batch: a row is finished after its first END
                              (predictor.py:343); a row's ids after its END are -1              */

/* Token selection. */
#define I2L_SELECT_LOGITS 0   /* argmax(logits / temperature)           seq2seq.py:213-215      */
#define I2L_SELECT_SOFTMAX 1  /* argmax(softmax(logits / temperature))  predictor.py:295-297,333: the row-per-workgroup
                                 kernel evaluates the fp32 probabilities literally; the grouped kernel takes the arg max of
                                 the scaled logits, which is the same token unless the two largest are adjacent floats
                                 AND smaller than 2 in magnitude (only then can rounding make their probabilities equal
                                 and hand the tie to the lower index) */
#define I2L_SELECT_SAMPLE 2   /* multinomial over the top-k / top-p masked softmax  predictor.py:299-331 (i2l_sample_decode) */

/* The decode loop: `steps` iterations of [embedding lookup, LSTM step (all layers),
 * output projection, token selection], one persistent launch, no host sync inside.
 * Replaces the loops at seq2seq.py:210-221 and predictor.py:283-347 and, with
 * steps == 1, LSTMDecoder.decode_step (decoder.py:197-284).
 *   workspace   filled by i2l_decoder_prepare() for these `rows`; its trailing scratch region (exchange
 *               granules of the grouped kernel) is WRITTEN by the call: one decode at a time per workspace
 *   tok0        (rows) int32 first input token of every row
 *   forced      (rows, steps) int32 or NULL; if given, the input token of step t is
 *               forced[r][t] (teacher forcing) instead of the previous selection
 *   h0, c0      (L, rows, H) or NULL (zeros, decoder.py:231-244)
 *   ids_out     (rows, steps) int32 or NULL   selected token per step
 *   logits_out  (rows, steps, V) or NULL      raw logits (before temperature)
 *   h_out,c_out (L, rows, H) or NULL          state after the last executed step
 * First index wins ties, as torch.argmax.
 * Kernel choice: L == 1, H == 256, V <= 512, select != I2L_SELECT_SAMPLE, no state in/out, steps >= 8 run the
 * grouped kernel (4 workgroups share 4 rows and keep the weights on chip, in-launch exchanges bounded by a 3 s
 * wall-clock limit); everything else the row-per-workgroup kernel.  If a bounded wait expires (GPU heavily
 * oversubscribed) every id of the affected rows is -3, every requested logit of those rows is NaN and no other
 * output is defined; i2l_greedy_decode_ex with rows_per_workgroup != 0 selects the row-per-workgroup kernel,
 * which needs no co-resident partner. */
int i2l_greedy_decode(const i2l_decoder_weights* w, const void* workspace, int rows, int steps,
                      const int32_t* tok0, const int32_t* forced, const float* h0, const float* c0,
                      float temperature, int select, int stop, int end_id,
                      int32_t* ids_out, float* logits_out, float* h_out, float* c_out,
                      i2l_stream_t stream);

/* Same as i2l_greedy_decode with an explicit number of batch rows per workgroup (0 = automatic, else 1, 2
 * or 4; a non-zero value also selects the row-per-workgroup kernel), flags (I2L_FLAG_AGENT_SCOPE_EXCHANGE,
 * I2L_FLAG_DECODE_GROUP8) and a RESIDENCY SIGNAL: with resident_flag != NULL (a caller-owned device word) the launch
 * stores resident_value there as soon as its workgroups own their compute units -- a grouped kernel when its last group
 * has completed the placement exchange, any other kernel at once -- for i2l_stream_wait_value32 on another stream
 * (GreedyPipeline: the encoder of batch i + 1 starts beside the decode of batch i, never before it).  If a group times
 * out the word is not written; the waiter's own bound then ends the wait.  Results do not depend on these parameters. */
int i2l_greedy_decode_ex(const i2l_decoder_weights* w, const void* workspace, int rows, int steps,
                         const int32_t* tok0, const int32_t* forced, const float* h0, const float* c0,
                         float temperature, int select, int stop, int end_id, int rows_per_workgroup,
                         int32_t* ids_out, float* logits_out, float* h_out, float* c_out, int flags,
                         uint32_t* resident_flag, uint32_t resident_value, i2l_stream_t stream);

/* The sampling branch of Predictor.predict_batch (predictor.py:295-331, taken when temperature > 0 and
 * (top_k > 0 or top_p > 0)): probs = softmax(logits/T); top-k keeps p >= k-th largest; top-p drops a
 * token once the probability mass sorted ahead of it exceeds top_p; renormalise; ONE multinomial draw per
 * row and step by inverse CDF with a counter-based uniform of (seed, row, step) -- deterministic for a
 * seed, not bit-comparable with torch.multinomial.  probs_out (rows, steps, V) or NULL receives the final
 * sampling distribution of every step.  vocab <= 2048. */
int i2l_sample_decode(const i2l_decoder_weights* w, const void* workspace, int rows, int steps,
                      const int32_t* tok0, const float* h0, const float* c0, float temperature, int top_k,
                      float top_p, uint64_t seed, int stop, int end_id, int32_t* ids_out, float* probs_out,
                      float* h_out, float* c_out, i2l_stream_t stream);

/* Beam search for `images` independent images, `beam` beams each (<= I2L_MAX_BEAM):
 * per image exactly Seq2SeqModel._beam_search at batch 1 (seq2seq.py:234-298) --
 * log_softmax in fp32, top-k sorted descending with lower index first on ties, scores
 * accumulated in fp64, stable descending selection, ended beams retire one iteration
 * later, no length normalisation.  workspace prepared for rows == images.
 *   seq_out  (images, steps+1) int32: best sequence, START stripped, cut at END, -1 padded
 *   len_out  (images) int32;  score_out (images) fp64 or NULL.
 * Kernel choice: hidden == 256, one layer, vocab <= 512, 2 <= beam <= 6 run the grouped kernel (four workgroups
 * share 12 beam slots = 12/beam images and keep the weights on chip; per step they exchange h and, per slot, each
 * member's local top-k and (max, sum exp), from which log_softmax is assembled); other shapes, and
 * flags & I2L_FLAG_NO_GROUP, run one workgroup per image.  Both give the same sequences; scores agree to fp32
 * log_softmax rounding.  If a poll of the grouped kernel times out (GPU shared or oversubscribed) the affected
 * images get len_out = -3 and no other output: call again with I2L_FLAG_NO_GROUP. */
size_t i2l_beam_workspace_bytes(int images, int beam, int hidden, int layers, int steps);
int i2l_beam_decode(const i2l_decoder_weights* w, const void* workspace, int images, int beam,
                    int steps, int start_id, int end_id, void* beam_workspace,
                    size_t beam_workspace_bytes, int32_t* seq_out, int32_t* len_out,
                    double* score_out, int flags, i2l_stream_t stream);

/* Additive attention, general source length S (Attention.forward, decoder.py:312-343):
 * context[b] = softmax_s(v . tanh(W [hidden[b] ; enc[b,s]] + b_a)) @ enc[b].
 * hidden (B,H)  enc (B,S,E)  w_attn (H, H+E)  b_attn (H)  v (H)  context (B,E). */
int i2l_attention_context_fwd(const float* hidden, const float* enc, const float* w_attn,
                              const float* b_attn, const float* v, float* context,
                              int B, int S, int H, int E, i2l_stream_t stream);

/* ------------------------------------------------------------------------
 * Training step (reference img2latex/training/trainer.py:303-343, fp32 branch)
 * ---------------------------------------------------------------------- */

/* Gradient buffers, same shapes / key names as i2l_decoder_weights (HOST arrays of device pointers). */
typedef struct i2l_decoder_grads {
    float* embedding;
    float* const* w_ih;
    float* const* w_hh;
    float* const* b_ih;
    float* const* b_hh;
    float* w_out;
    float* b_out;
} i2l_decoder_grads;

/* Teacher-forced LSTMDecoder.forward in training mode (decoder.py:100-195) for input tokens
 * (B,T) int32 [= formulas[:, :-1], seq2seq.py:115-120]; keeps what BPTT needs in `workspace`.
 * dropout_p: nn.Dropout p (masks from a counter-based hash of `seed`; 0 disables);
 * attention_path: 0 = decoder.py:121-143 (dropout on cat[emb,enc]), 1 = :144-193 (dropout on emb).
 * logits_out (B,T,V).  flags: I2L_FLAG_NO_GROUP (row-per-workgroup recurrences), I2L_FLAG_EXACT_FP32 (GEMMs);
 * pass the same flags to the backward call.  If a bounded wait of the grouped recurrence expires its outputs are
 * NaN (the loss and every gradient become NaN; i2l_grad_clip_adam_step then skips the update). */
size_t i2l_decoder_train_workspace_bytes(int B, int T, int vocab, int embed, int hidden, int layers);
int i2l_decoder_train_fwd(const i2l_decoder_weights* w, const float* enc, const int32_t* tokens, int B, int T,
                          float dropout_p, uint64_t seed, int attention_path, void* workspace,
                          size_t workspace_bytes, float* logits_out, int flags, i2l_stream_t stream);
/* Backward of the above for a given dlogits (B,T,V): fills every gradient of `grads` (overwrite)
 * and denc_out (B,E).  `workspace` must be the one the forward call filled. */
int i2l_decoder_train_bwd(const i2l_decoder_weights* w, const int32_t* tokens, int B, int T, float dropout_p,
                          uint64_t seed, int attention_path, void* workspace, size_t workspace_bytes,
                          const float* dlogits, const i2l_decoder_grads* grads, float* denc_out,
                          int flags, i2l_lanes* lanes, i2l_stream_t stream);

/* nn.CrossEntropyLoss(ignore_index=pad, label_smoothing=eps) over `rows` = B*T rows of logits (rows,V)
 * (trainer.py:111-115,335-336).  loss_sum_and_count_out[0] = SUM over non-pad rows of the per-row loss,
 * [1] = number of non-pad rows (loss = [0]/[1]); dlogits_out (rows,V) or NULL = gradient of the SUM
 * (divide by the GLOBAL count after the data-parallel all-reduce). */
size_t i2l_ce_workspace_bytes(int rows);
int i2l_ce_label_smooth_fwd_bwd(const float* logits, const int32_t* targets, int rows, int vocab, int pad_id,
                                float label_smoothing, void* workspace, size_t workspace_bytes,
                                float* dlogits_out, float* loss_sum_and_count_out, i2l_stream_t stream);

/* clip_grad_norm_(max_norm) + Adam(lr, betas, eps, coupled L2 weight_decay) over flat fp32 buffers of
 * n elements (trainer.py:91-93,338-342).  grads hold the gradient of the SUM loss; count_ptr (device,
 * may be NULL = 1) is the number of non-pad target tokens the sum runs over (the GLOBAL count after the
 * data-parallel all-reduce).  step >= 1 is the optimizer step number (bias correction).
 * stats_out (device, 4 floats): total gradient norm, clip coefficient, 1/count, skipped (1.0 when the total norm
 * was not finite -- Inf/NaN gradients, e.g. after a grouped kernel's timeout -- and the call therefore left
 * params, exp_avg and exp_avg_sq untouched, exactly what GradScaler's skipped step does; else 0.0).
 * max_norm <= 0: no clipping.  The workspace also carries the number of skipped calls so far (zero it once before
 * the first call; the bias correction uses step minus that number, so `step` may simply count calls). */
size_t i2l_optimizer_workspace_bytes(void);
int i2l_grad_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                            const float* count_ptr, float max_norm, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, void* workspace, size_t workspace_bytes,
                            float* stats_out, i2l_stream_t stream);

/* ------------------------------------------------------------------------
 * Evaluation metrics (reference img2latex/training/metrics.py), SURVEY section 8(f)-4
 * ---------------------------------------------------------------------- */

/* Integer statistics of `pairs` (prediction, target) id sequences, one workgroup per pair:
 *   lev_out[p]       the corner dist_tab[rows][cols] of levenshtein_distance's table, metrics.py:63-83
 *                    (equal tokens copy the diagonal neighbour, otherwise 1 + min(up, left, diagonal))
 *   match_out[p][g]  g = 0..3: sum over the DISTINCT (g+1)-grams of the prediction of
 *                    min(count in prediction, count in target) -- bleu_n_score's matching_grams_sum,
 *                    metrics.py:136-160; 0 for g >= max_n or when either sequence is shorter than g+1
 *   tla_out[p][0..1] token_list_accuracy's (correct, non-pad) counts over the first min(len) positions,
 *                    metrics.py:259-274; may be NULL
 * pred (pairs, pred_stride) / target (pairs, target_stride) int32 rows, lengths clamped to [0, max_len].
 * The float formulas on top (1 - d / max_len; precisions, geometric mean, brevity penalty) are host
 * arithmetic in float64, exactly as the reference writes them (img2latex_amd/training/metrics.py). */
int i2l_sequence_metrics(const int32_t* pred, const int32_t* pred_len, int pred_stride, const int32_t* target,
                         const int32_t* target_len, int target_stride, int pairs, int max_len, int max_n,
                         int pad_id, int32_t* lev_out, int32_t* match_out, int32_t* tla_out, i2l_stream_t stream);

/* Host helper (no GPU work): mean BLEU-n and mean Levenshtein similarity (calculate_metrics, metrics.py:184-223) from the
 * integer statistics of i2l_sequence_metrics, `pairs` rows of `stride` >= 9 int32 = [lev, match 1..4, 2 unused, gen_len,
 * true_len].  The reference's float64 formulas operation by operation (libm log / exp = CPython's math.log / math.exp):
 * bit-identical scores without 256 x 2 interpreter calls per batch. */
int i2l_scores_from_statistics(const int32_t* stats, int pairs, int stride, int n, double* bleu_mean_out,
                               double* lev_mean_out);

/* The id post-processing between the decode loop and the metrics, on the device (cli.py:466-481 with
 * predictor.py:350-358,384-391 and tokenizer.py:166-192): per row keep, in order, the ids BEFORE the first stop
 * position -- id == end_id, or id < 0 (the sticky stop rule's filler) -- that are not one of drop_ids[0..n_drop)
 * (DEVICE array of <= 8 ids: the special tokens LaTeXTokenizer.decode(skip_special_tokens=True) removes;
 * for target rows: the PAD id, with end_id = -1 so that nothing stops a row).  ids (rows, stride) int32, `width`
 * columns used; out_ids (rows, out_stride >= width); out_len (rows).  Bit-exact integer work. */
int i2l_compact_ids(const int32_t* ids, int rows, int width, int stride, int end_id, const int32_t* drop_ids,
                    int n_drop, int32_t* out_ids, int out_stride, int32_t* out_len, i2l_stream_t stream);

/* masked_accuracy, metrics.py:226-238 (trainer.py:391,526): over rows = B*T logits rows of `vocab` floats,
 * correct_total_out[0] = #(argmax == target and target != pad), [1] = #(target != pad); first index wins
 * ties.  The (B,T,V) logits never leave the device (the reference copies them to the host every step). */
int i2l_masked_accuracy(const float* logits, const int64_t* targets, int64_t rows, int vocab, int64_t pad_id,
                        uint64_t* correct_total_out, i2l_stream_t stream);

/* Where the grouped greedy kernel's status words live inside the decoder workspace (0: these dimensions have no
 * grouped path): uint32 [0] != 0 -> a poll timed out (the ids are -3), [1] groups that completed the placement
 * exchange in the last launch, [2] of those, the groups whose four workgroups measured themselves on ONE XCD and
 * therefore exchanged through that XCD's L2 (the fast flavour).  Diagnostics: read after synchronising the stream. */
size_t i2l_decoder_group_status_offset(int rows, int vocab, int embed, int hidden, int layers);
size_t i2l_decoder_group_region_bytes(int rows, int vocab, int embed, int hidden, int layers);   /* bytes of that region (status + exchange granules) */

/* ------------------------------------------------------------------------
 * Image preprocessing (reference img2latex/data/utils.py:18-90, data/transforms.py:26-56), SURVEY section 8(f)-3
 * ---------------------------------------------------------------------- */

/* Host helpers (no GPU work): Pillow's LANCZOS coefficient tables for resampling `in_size` source samples to
 * `out_size` (libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc over the full source range, double
 * precision, libm sin): bounds_out (out_size, 2) = first source sample and count, kk_out (out_size, ksize) = 22-bit
 * fixed-point weights, ksize = i2l_lanczos_ksize(in_size, out_size). */
int i2l_lanczos_ksize(int in_size, int out_size);
int i2l_lanczos_coeffs(int in_size, int out_size, int32_t* bounds_out, int32_t* kk_out);
/* The same tables for a named Pillow filter (values of PIL.Image.Resampling): LANCZOS is what load_image asks for
 * (transforms.py:20-24), BICUBIC what Image.resize() defaults to -- the PIL.Image branch of Predictor._prepare_image
 * (predictor.py:432-439) resizes straight to (800, 64) with it, aspect ratio NOT kept. */
#define I2L_FILTER_LANCZOS 1
#define I2L_FILTER_BICUBIC 3
int i2l_resample_ksize(int filter, int in_size, int out_size);
int i2l_resample_coeffs(int filter, int in_size, int out_size, int32_t* bounds_out, int32_t* kk_out);
/* n tables at once on up to `threads` host threads: entry i (in_sizes[i] -> out_sizes[i]) is written at out +
 * offsets[i] as bounds (out, 2) followed by weights (out, ksize).  No global state: the threads live for the call. */
int i2l_resample_coeffs_batch(int filter, int n, const int32_t* in_sizes, const int32_t* out_sizes,
                              const int64_t* offsets, int32_t* out, int threads);

/* The same tables built ON THE DEVICE for n resamplings at once (in_sizes / out_sizes / offsets are DEVICE arrays; entry i
 * is written at tables + offsets[i] as i2l_resample_coeffs_batch lays it out; max_out_size = the largest out_sizes[i]):
 * the host function's double-precision arithmetic repeated operation by operation, with the device's sin() in place of
 * libm's -- both faithfully rounded, so a 22-bit weight can differ by one unit where the normalised value sits within
 * ~1e-16 of a rounding boundary (about once in 1e9 weights).  i2l_resample_coeffs stays the Pillow-identical reference.
 * A batch of ragged pages then uploads nothing but its pixels, sizes and plans (the host-side tables cost
 * more than the device chain). */
int i2l_resample_coeffs_device(int filter, int n, const int32_t* in_sizes, const int32_t* out_sizes, const int64_t* offsets,
                               int32_t* tables, int max_out_size, i2l_stream_t stream);
/* Host helper (no GPU work): copies n host buffers srcs[i] (sizes[i] bytes) to dst + offsets[i] on up to `threads` host
 * threads that live for the call -- the separate page arrays PIL hands the reference's load_image, gathered into the
 * one pinned block that is uploaded. */
int i2l_pack_host(const void* const* srcs, const int64_t* sizes, const int64_t* offsets, int n, void* dst, int threads);

/* One image of a ragged batch.  Offsets index `pixels` (bytes), `tables` (int32 elements) and the workspace
 * (bytes).  Bounds are ABSOLUTE source indices as i2l_resample_coeffs writes them; when need_h the kernel itself
 * subtracts ybox_first from the vertical ones (Resample.c: "Shift bounds for vertical pass"), so one table serves
 * every image of the same (source size, target size). */
typedef struct i2l_resize_plan {
    int64_t src_offset;            /* first byte of the uint8 image: (src_h, src_w) or (src_h, src_w, 3) interleaved */
    int64_t tmp_offset;            /* intermediate image (tmp_rows, new_w, out_c) uint8 in the workspace             */
    int64_t bh_offset, kh_offset;  /* horizontal bounds / weights                                                    */
    int64_t bv_offset, kv_offset;  /* vertical bounds / weights                                                      */
    int32_t src_h, src_w, src_c;
    int32_t new_w;                 /* int(round(out_h * src_w / src_h)), transforms.py:33-36; = out_w for a plain resize */
    int32_t ybox_first, tmp_rows;  /* source rows the vertical pass needs                                            */
    int32_t need_h, need_v;        /* new_w != src_w, out_h != src_h                                                 */
    int32_t kh_ksize, kv_ksize;
} i2l_resize_plan;

/* load_image for n decoded images: convert("L"/"RGB") -> LANCZOS resize to (new_w, out_h) -> right-pad (white for
 * 1 channel; Pillow's integer colour 255 = (255,0,0) for 3 channels, as the reference executes) or centre-crop to
 * out_w -> /255 -> [-1,1] (1 channel) or ImageNet mean/std (3 channels) when `normalize`.  out (n, out_c, out_h,
 * out_w) fp32, bit-identical to the reference's tensor.  plans / tables live in device memory; max_tmp_px = the
 * largest tmp_rows*new_w over the batch (0 when no image needs a horizontal pass).  normalize: 0 = /255 only,
 * 1 = load_image's rule above, 2 = x * 2 - 1 on every channel (the PIL.Image branch of Predictor._prepare_image,
 * predictor.py:447-451).  The filter is whatever the plan's tables were built with. */
int i2l_preprocess_images(const uint8_t* pixels, const i2l_resize_plan* plans, const int32_t* tables, int n,
                          int max_tmp_px, int out_c, int out_h, int out_w, int normalize, void* workspace,
                          float* out, i2l_stream_t stream);

/* The tensor branch of Predictor._prepare_image (predictor.py:483-491): torch.nn.functional.interpolate(mode=
 * "bilinear", align_corners=False) of `planes` fp32 images (in_h, in_w) -> (out_h, out_w), contiguous; fp32 results
 * within 1 ulp-level rounding of ATen's (tested <= 1e-6). */
int i2l_resize_bilinear_f32(const float* in, float* out, int64_t planes, int in_h, int in_w, int out_h, int out_w,
                            i2l_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IMG2LATEX_HIP_H */
