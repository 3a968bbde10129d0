/*
 * n3d.h -- C ABI of libn3d.so: gfx950 (MI355X) HIP kernels for the nas_3d_unet hot path.
 *
 * The reference (woodywff/nas_3d_unet) has NO native layer and NO FFI: its hot path is a
 * Python module API (prim_ops.py / cell.py) whose arithmetic is delegated to torch.nn.
 * Each entry point below therefore cites the reference *call site* whose torch op it replaces
 * (file:line relative to the reference root).  The Python host (nas_3d_unet_amd/) binds these
 * with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (n3d_last_error() has the text);
 *     nothing throws, allocates, or synchronises; all work is enqueued on `stream`
 *     (a hipStream_t passed as void*), so calls are HIP-graph capturable.
 *   - activations are fp32, NDHWC ("channels-last-3d"): element (b,d,h,w,c) of a tensor with
 *     voxel pitch `ld` (floats, ld >= C) lives at ((((b*D+d)*H+h)*W+w)*ld + c).  A pitch larger
 *     than C addresses a channel slice of a wider buffer (zero-copy concat, cell.py:82).
 *   - weights keep torch's native layouts: Conv3d (Cout, Cin/g, k,k,k), ConvTranspose3d
 *     (Cin, Cout/g, k,k,k), Linear (out, in), so reference state_dicts load unchanged.
 *   - per-(sample,channel) reductions are deterministic two-stage sums: a producer kernel writes
 *     one partial row of doubles per workgroup, `double[B][rows][C][nv]`, and a tiny coefficient
 *     kernel adds the rows in a fixed order.  `rows` comes from n3d_stats_rows() /
 *     n3d_conv_stats_rows() so the caller can size the buffer; no atomics, no zeroing.
 */
#ifndef N3D_H_
#define N3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define N3D_OK 0
#define N3D_ERR_INVALID (-1)
#define N3D_ERR_UNSUPPORTED (-2)
#define N3D_ERR_HIP (-3)
#define N3D_ERR_WORKSPACE (-4)

/* flags */
#define N3D_RELU_IN 1     /* apply ReLU to the conv input on load ('act_weight_norm', cell.py:47-50) */
#define N3D_RELU 2        /* epilogue activation is ReLU (prim_ops.py:63,80) */
#define N3D_ACCUMULATE 4  /* destination += result instead of = result */
#define N3D_POOL_MAX 8    /* pooling type (prim_ops.py:160-163) */
#define N3D_NO_MFMA 16    /* force the generic VALU kernels (A/B testing) */
#define N3D_PREPACKED 32  /* `ws` already holds this conv's packed weights (written by n3d_pack_batch) */
/* storage types (bf16 configuration, BASELINE configs[4]): activation pointers are declared `float*` for the fp32 path; with one of
 * these flags the tensor holds bfloat16 elements instead (pitches then count bf16 elements, 4-channel groups must be 8-byte aligned)
 * and the kernel converts on load / store, computing in fp32.  Conv family: SRC = the first activation tensor of the call (x; dy of a
 * data gradient; x of a weight gradient), DST = the second (y; dx and relu_src of a data gradient; dy of a weight gradient).
 * Epilogue family (n3d_channel_stats / n3d_affine_act*): every activation tensor of one call shares ONE type, N3D_ACT_BF16 (or the
 * `dtype` field of the term structures).
 * READABLE SLACK: the 3x3x3 kernels of the bf16 path fill their LDS tiles 16 bytes per voxel (LDS-DMA), so a bf16 tensor with
 * 4 channels (an 8-byte voxel) is read up to 8 bytes past its last element.  Every bf16 activation tensor handed to the conv
 * family must therefore be followed by at least 16 readable bytes in the same allocation (their content is ignored); the
 * Python host allocates bf16 tensors with that slack and repacks foreign ones (kernels.as_view). */
#define N3D_SRC_BF16 64
#define N3D_DST_BF16 128
#define N3D_ACT_BF16 64
/* bf16 configuration, the C >= 16 levels (fp32 storage): the MFMA kernels of the conv family (forward, data and weight gradients with
 * channel counts that are multiples of 16) round BOTH operands of their matrix products to bfloat16 in registers and accumulate in
 * fp32 (v_mfma_f32_16x16x16_bf16 instead of four v_mfma_f32_16x16x4_f32).  Ignored by every other kernel.  Never set by the fp32
 * configuration, whose arithmetic is exact fp32. */
#define N3D_MM_BF16 256

/* storage type of an activation tensor (entry points that take a dtype argument; all others are fp32) */
#define N3D_F32 0
#define N3D_BF16 1        /* bfloat16 storage, fp32 arithmetic: BASELINE configs[4] (4x128^3 patches, HBM-bound levels) */
#define N3D_U8 2          /* bytes holding exactly {0, 1}: the targets of n3d_head_fwd / n3d_head_bwd (n3d_head.t_dtype) and of n3d_patch_batch */

/* Geometry of a (possibly strided / dilated) 3-D convolution, torch Conv3d semantics:
 * o = floor((i + 2*pad - dil*(k-1) - 1)/stride) + 1.  "i side" is what the window slides over.
 * For a transposed convolution the i side is its OUTPUT and the o side its INPUT. */
#define N3D_MAX_GROUP_TERMS 8   /* terms / jobs per batched launch (the N-term entry points below) */
#define N3D_MAX_REDUCE_TERMS 16 /* n3d_affine_act_bwd_reduceN / n3d_node_bwd_coeffs: every term of a supernet node level in one launch */

typedef struct n3d_conv_geom {
  int32_t B;
  int32_t Di, Hi, Wi, Ci;
  int32_t Do, Ho, Wo, Co;
  int32_t k, stride, dil, pad;
  int32_t depthwise; /* 1: groups == Ci == Co (prim_ops.py:95-97,105-106) */
} n3d_conv_geom;

const char* n3d_last_error(void);
int n3d_version(void);
/* 1 if the library was built for gfx950 and a HIP device is usable, 0 otherwise */
int n3d_device_ok(void);

int n3d_zero(void* p, size_t bytes, void* stream);

/* ---- convolution family: nn.Conv3d / nn.ConvTranspose3d (prim_ops.py:95-110,140-147) -------------
 * workspace: packed weights; query with n3d_conv_workspace_bytes().
 * in_gate : optional (B, Ci_of_the_conv_input) per-sample channel scale applied on load: the SE
 *           gate x*y (prim_ops.py:152) or a Dropout3d mask (prim_ops.py:72-73).
 * stats   : optional double[B][rows][Cout][2] partial (sum, sum of squares) of the raw output incl.
 *           bias: GroupNorm statistics produced in the conv epilogue (prim_ops.py:58,77). */
size_t n3d_conv_workspace_bytes(const n3d_conv_geom* g);
/* rows per sample that n3d_conv_fwd (transposed=0) / n3d_convT_fwd (transposed=1) write into `stats` */
int n3d_conv_stats_rows(const n3d_conv_geom* g, int transposed, int flags);
/* rows per sample written by n3d_channel_stats / n3d_affine_act_bwd_reduce for N voxels, C channels */
int n3d_stats_rows(int64_t N, int C);

/* Batched weight packing: the conv kernels read weights from a kernel-friendly packed copy.  By default each
 * conv call packs into its workspace (one tiny extra launch); a trainer instead packs ALL weights of the net
 * with one launch per step (n3d_pack_batch) and passes N3D_PREPACKED + the packed slot as `ws`.
 * n3d_conv_pack_info: layout id / padded channel count / float count of the packed form that the kernel
 * selected for (geometry, forward or data-gradient) expects. */
typedef struct n3d_pack_job {
  const float* w; float* dst;   /* layout 4 (bf16-storage 3x3x3 kernels): dst holds bfloat16 elements */
  int32_t Co, Ci, taps, data_grad, layout, cdp;
} n3d_pack_job;
int n3d_conv_pack_info(const n3d_conv_geom* g, int data_grad, int flags, int32_t* layout, int32_t* cdp, int64_t* floats);
int n3d_pack_batch(const n3d_pack_job* jobs /* host array */, int njobs, void* stream);

/* Deferred weight-gradient reduction: n3d_conv(T)_bwd_weight leave their partial slabs in `ws` and describe the
 * remaining fixed-order reduction in *deferred; n3d_wgrad_finalize_batch then finishes many convs in one launch. */
typedef struct n3d_final_job {
  const float* partial; const float* pbias; float* dw; float* dbias;
  int32_t nchunks, ntiles, tci, tco, ci_t, co_t, Co, Ci, taps, pad_;
} n3d_final_job;
int n3d_wgrad_finalize_batch(const n3d_final_job* jobs /* host array */, int njobs, void* stream);
/* Both batch calls compact their job table into the kernel arguments (pointers as 3-bit segment + 29-bit float offset against
 * up to eight 2 GB address segments per launch); jobs whose addresses do not fit one table go to further launches.
 * n3d_selftest_job_tables: host-only check of that encoding on scattered fake addresses (no device needed);
 * returns the number of launches its 300 jobs take, < 0 on a mismatch. */
int n3d_selftest_job_tables(void);

/* ---- "weight_norm" 1x1x1 conv WITHOUT its raw output (round 4; stem0: ConvOps(in, 3 c, kernel_size=1, ops_order='weight_norm'),
 * nas.py:28 / searched.py:69, prim_ops.py:68-83).  With 4 (8) input channels, raw = W x + bias costs 4 (8) FMAs per channel to
 * recompute, while storing it and reading it back in the GroupNorm epilogue and in both passes of its backward -- and writing d(raw)
 * for the weight gradient -- moves the 12-channel tensor five more times (0.2 ms of a 4x128^3 step).  The op then runs as
 *   forward : n3d_conv_k1_norm_fwd(y = NULL, stats)   statistics of raw only, nothing stored
 *             n3d_gn_coeffs                           (a, b) per (sample, channel)
 *             n3d_conv_k1_norm_fwd(y, oscale = a, oshift = b)   y = a * (W x + bias) + b in one pass over x
 *   backward: n3d_conv_k1_norm_bwd_reduce             rows [B][n3d_conv_k1_norm_rows][Co][3] as n3d_affine_act_bwd_reduce writes them
 *             n3d_gn_bwd_coeffs                       (A, Bc, Cc), d gamma, d beta, d bias
 *             n3d_conv_k1_norm_bwd_apply_wgrad        d(raw) = A g + (Cc raw + Bc) in registers -> dW slabs (deferred as
 *                                                     n3d_conv_bwd_weight); d(raw) is never written, so the op has NO input gradient
 * (Ci, Co) in {(4,4), (4,8), (4,12), (8,4)}, stride 1, >= 32768 voxels per sample (n3d_conv_k1_norm_ok); x fp32 or bf16
 * (N3D_SRC_BF16), y / dout fp32 or bf16 (N3D_DST_BF16); flags & N3D_RELU: the op has a ReLU behind its norm.  w: the native
 * (Co, Ci) weight for the backward calls; ws of the forward call as n3d_conv_fwd (packed weights, N3D_PREPACKED honoured). */
int n3d_conv_k1_norm_ok(const n3d_conv_geom* g);
int n3d_conv_k1_norm_rows(const n3d_conv_geom* g);
int n3d_conv_k1_norm_fwd(const n3d_conv_geom* g, const float* x, int64_t xld, const float* w, const float* bias, float* y, int64_t yld,
                         int flags, const float* oscale, const float* oshift, double* stats, void* ws, size_t ws_bytes, void* stream);
int n3d_conv_k1_norm_bwd_reduce(const n3d_conv_geom* g, const void* x, int64_t xld, const float* w, const float* bias, const void* dout,
                                int64_t dld, const float* a, const float* b, int flags, double* sums, void* stream);
int n3d_conv_k1_norm_bwd_apply_wgrad(const n3d_conv_geom* g, const void* x, int64_t xld, const float* w, const float* bias,
                                     const void* dout, int64_t dld, const float* a, const float* b, const float* A, const float* Bc,
                                     const float* Cc, int flags, float* dw, void* ws, size_t ws_bytes, n3d_final_job* deferred,
                                     void* stream);


/* ---- node-planar tensors (round 5).  A cell's output is torch.cat of its node outputs (cell.py:82, searched.py:51); as channel slices of
 * one (B, n c) buffer every node epilogue writes 16 / 32 bytes on a 48 / 96-byte pitch (1.3-1.7x the algorithmic HBM traffic).  Where the
 * only reader of the concatenation is a 1x1x1 preprocess conv, the nodes stay n DENSE (B, c, D, H, W) NDHWC tensors in one allocation --
 * storage (n, B, D, H, W, c) -- and the conv entry points take the whole thing as ONE tensor whose PITCH IS SMALLER THAN ITS CHANNEL COUNT:
 * (pointer to node 0, ld = c) with C = n c channels means node k at pointer + k * B * voxels * c elements.  Accepted by
 *   n3d_conv_fwd        x     (xld < Ci)                      }  1x1x1, stride 1, no gates, >= 32768 voxels per sample, channels % 4 == 0,
 *   n3d_conv_bwd_data   dx and relu_src (dxld = rld < Ci)     }  <= 24 channels on the planar side, <= 12 per node for the data gradient (one
 *   n3d_conv_bwd_weight x     (xld < Ci)                      }  node per blockIdx.z); fp32 or bf16 storage -- N3D_ERR_UNSUPPORTED otherwise.
 * The fused head reads the same layout through n3d_head.node_c. */

/* y[o side] = conv(x[i side]) + bias */
int n3d_conv_fwd(const n3d_conv_geom* g, const float* x, int64_t xld, const float* w, const float* bias,
                 float* y, int64_t yld, int flags, const float* in_gate, double* stats,
                 void* ws, size_t ws_bytes, void* stream);
/* dx[i side] (+)= conv^T(dy[o side]); if relu_src != NULL (N3D_RELU_IN) dx is masked by relu_src > 0;
 * out_gate multiplies dx per (b, ci) */
int n3d_conv_bwd_data(const n3d_conv_geom* g, const float* dy, int64_t dyld, const float* w,
                      float* dx, int64_t dxld, int flags, const float* relu_src, int64_t rld,
                      const float* out_gate, void* ws, size_t ws_bytes, void* stream);
/* dw (native layout) = sum x * dy, dbias = sum dy (either may be NULL) */
int n3d_conv_bwd_weight(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld,
                        float* dw, float* dbias, int flags, const float* in_gate,
                        void* ws, size_t ws_bytes, n3d_final_job* deferred /* NULL: finish now */, void* stream);

/* Data gradient AND weight gradient of one convolution: exactly n3d_conv_bwd_data(...) followed by
 * n3d_conv_bwd_weight(...) (same arguments, same results), but issued as ONE launch when both halves are small
 * MFMA problems (channel counts multiples of 16 on the 2^3..8^3 levels), where each half alone leaves most of the
 * chip idle.  Replaces the autograd backward of nn.Conv3d (prim_ops.py:109-110) on those levels. */
int n3d_conv_bwd_both(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, const float* w, float* dx,
                      int64_t dxld, int flags_data, const float* relu_src, int64_t rld, const float* out_gate, void* ws_data,
                      size_t ws_data_bytes, float* dw, float* dbias, int flags_weight, const float* in_gate, void* ws_weight,
                      size_t ws_weight_bytes, n3d_final_job* deferred, void* stream);

/* The same for nn.ConvTranspose3d (prim_ops.py:100-102): n3d_convT_bwd_data + n3d_convT_bwd_weight (weight gradient
 * only; the bias gradient of a transposed conv is a plain channel sum the hot path derives from the GroupNorm sums). */
int n3d_convT_bwd_both(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, const float* w, float* dx,
                       int64_t dxld, int flags_data, void* ws_data, size_t ws_data_bytes, float* dw, int flags_weight, void* ws_weight,
                       size_t ws_weight_bytes, n3d_final_job* deferred, void* stream);

/* ---- two independent convolutions in one launch: the two ops of a searched-cell node (searched.py:45-50) or a cell's
 * two preprocess convs (cell.py:47-50).  Exactly the two single calls described by the structures (same arguments,
 * same results, same fallbacks); ONE launch when both are small MFMA problems of the same K-split plan (channel
 * counts multiples of 16 on the 2^3..16^3 levels), where each alone leaves most of the chip idle. */
typedef struct n3d_conv_fwd_call {     /* arguments of n3d_conv_fwd / n3d_convT_fwd */
  const n3d_conv_geom* g; int32_t transposed; int32_t flags;
  const float* x; int64_t xld; const float* w; const float* bias; float* y; int64_t yld;
  const float* in_gate; double* stats; void* ws; size_t ws_bytes;
} n3d_conv_fwd_call;
int n3d_conv_fwd2(const n3d_conv_fwd_call* c0, const n3d_conv_fwd_call* c1, void* stream);
/* the same for up to four calls (`calls` = array): the plain-conv primitives of a supernet node (cell.py:76-81).  One launch when
 * all are small MFMA problems of one K-split plan, or all one-plane-tile 3x3x3 convs of one channel count in {4, 8} (stride 1 or 2,
 * distinct outputs); otherwise two at a time as n3d_conv_fwd2, a leftover alone.  Results and statistics rows are those of the
 * single calls bit for bit whatever the grouping.  Calls that are not N3D_PREPACKED should bring distinct workspaces: calls that share
 * one are never folded (a folded launch reads every call's packed weights at once) and run one after the other. */
int n3d_conv_fwdN(const n3d_conv_fwd_call* calls, int n, void* stream);

/* The depthwise 3x3x3 convs of up to N3D_MAX_GROUP_TERMS primitives of a supernet node (one depthwise-separable primitive per
 * edge, cell.py:76-81) in ONE launch.  A job is one gather pass: data_grad = 0: dst[o side] = conv(src[i side]) + bias
 * (n3d_conv_fwd, or the data gradient of a transposed conv); data_grad = 1: dst[i side] (+)= conv^T(src[o side]) (n3d_conv_bwd_data,
 * or n3d_convT_fwd).  All jobs share batch, channel count and destination shape; destinations must not alias. */
typedef struct n3d_dw_job {
  const n3d_conv_geom* g; int32_t data_grad; int32_t flags /* N3D_ACCUMULATE */;
  const float* src; int64_t sld; const float* w; const float* bias; float* dst; int64_t dld;
} n3d_dw_job;
int n3d_dwconv_batch(const n3d_dw_job* jobs, int n, void* stream);

typedef struct n3d_conv_bwd_call {     /* arguments of n3d_conv_bwd_both / n3d_convT_bwd_both */
  const n3d_conv_geom* g; int32_t transposed; int32_t flags_data; int32_t flags_weight; int32_t pad_;
  const float* x; int64_t xld; const float* dy; int64_t dyld; const float* w; float* dx; int64_t dxld;
  const float* relu_src; int64_t rld; const float* out_gate; void* ws_data; size_t ws_data_bytes;
  float* dw; float* dbias; const float* in_gate; void* ws_weight; size_t ws_weight_bytes; n3d_final_job* deferred;
} n3d_conv_bwd_call;
/* the two data gradients must not alias (dx of c0 != dx of c1) for the one-launch form; aliasing falls back to two launches */
int n3d_conv_bwd_both2(const n3d_conv_bwd_call* c0, const n3d_conv_bwd_call* c1, void* stream);
/* Data gradients only of two convs (the architecture pass of the search step computes no weight gradients, search.py:223-231;
 * the reference runs loss.backward() through every conv's input there): fields x, dw, dbias, in_gate, ws_weight, deferred and
 * flags_weight of the calls are ignored.  One launch where both fit the small-tensor MFMA kernel and dx of c0 != dx of c1. */
int n3d_conv_bwd_data2(const n3d_conv_bwd_call* c0, const n3d_conv_bwd_call* c1, void* stream);
/* transposed convolution y[i side] = convT(x[o side]) + bias; same kernels with the roles swapped */
int n3d_convT_fwd(const n3d_conv_geom* g, const float* x, int64_t xld, const float* w, const float* bias,
                  float* y, int64_t yld, int flags, const float* in_gate, double* stats,
                  void* ws, size_t ws_bytes, void* stream);
int n3d_convT_bwd_data(const n3d_conv_geom* g, const float* dy, int64_t dyld, const float* w,
                       float* dx, int64_t dxld, int flags, void* ws, size_t ws_bytes, void* stream);
int n3d_convT_bwd_weight(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld,
                         float* dw, float* dbias, int flags, void* ws, size_t ws_bytes,
                         n3d_final_job* deferred /* NULL: finish now */, void* stream);

/* ---- per-(sample,channel) statistics and the normalise / activate / weighted-sum epilogue ----------
 * n3d_channel_stats: stats[b][row][c] = partial (sum x, sum x^2) over the N voxels (GroupNorm of a tensor
 *   that no conv of ours produced: IdentityOp prim_ops.py:170-174; SE mean prim_ops.py:149). */
int n3d_channel_stats(const float* x, int64_t ld, int B, int64_t N, int C, double* stats, void* stream);
/* the same with a storage type (N3D_F32 / N3D_BF16) */
int n3d_channel_stats_t(const void* x, int64_t ld, int dtype, int B, int64_t N, int C, double* stats, void* stream);
/* the same for up to N3D_MAX_GROUP_TERMS tensors of one (B, N, C) shape in one launch: xs / lds / stats are host arrays of n entries */
int n3d_channel_statsN(const float* const* xs, const int64_t* lds, double* const* stats, int n, int B, int64_t N, int C, void* stream);
/* ---- padded channels (round 5).  The reference takes feature maps of any channel count (nas.py:13-26, searched.py:55-66: init_n_kernels = 2,
 * 6, ...); the kernels move channels four at a time.  Such a net runs with every feature map zero-padded to a multiple of 4 (zero conv
 * weights / gamma / beta on the padded channels: they hold exactly 0 forward and receive exactly 0 backward).  Sums over a padded tensor
 * are the real tensor's sums, so one thing changes: the ELEMENT COUNT of a GroupNorm group.  Every entry point below that takes a group
 * count G accepts G < 0 = "ONE group whose real channel count is -G, stored in C >= -G channels": statistics are divided by N * (-G)
 * instead of N * C.  (A real count that needs padding is never a multiple of 16, i.e. always one group: prim_ops.py:57.)  Not taken with
 * G < 0: the one-launch small backward (n3d_bwd_small2_ok returns 0), n3d_node_fwd_coeffs / n3d_node_bwd_coeffs; conv-bias gradients out
 * of the GroupNorm sums (dbias_conv) need the true count and must not be requested. */
/* GroupNorm(G, C) statistics -> per-(b,c) affine y = a*x + b; mean_rstd[b][g] = (mean, rstd) (prim_ops.py:56-58) */
int n3d_gn_coeffs(const double* stats, int rows, const float* gamma, const float* beta, int B, int C, int G,
                  int64_t N, float eps, float* a, float* b, float* mean_rstd, double* sumraw /* [B][C] or NULL */,
                  void* stream);
/* out (+)= w * act(a[b,c]*raw + b[b,c]);  a,b NULL -> identity affine; wptr NULL -> 1.
 * This is GroupNorm-apply + ReLU (prim_ops.py:75-80) fused with the MixedOp weighting and the node
 * sum (cell.py:29-32,81; searched.py:50). */
int n3d_affine_act(const float* raw, int64_t rld, const float* a, const float* b, const float* wptr,
                   float* out, int64_t old_, int B, int64_t N, int C, int flags, void* stream);
/* Fused form for small tensors (rows <= n3d_fused_max_rows()): the GroupNorm coefficients are computed from the
 * statistics rows in every workgroup's prologue (no separate n3d_gn_coeffs launch); a / b / mean_rstd are also
 * stored for the backward pass. */
int n3d_fused_max_rows(void);
int n3d_affine_act_gn(const float* raw, int64_t rld, const double* stats, int rows, const float* gamma,
                      const float* beta, int G, float eps, const float* wptr, float* out, int64_t old_, int B,
                      int64_t N, int C, int flags, float* a_out, float* b_out, float* mean_rstd_out,
                      double* sumraw /* [B][C] per-channel sum of raw, or NULL */, void* stream);
/* backward, pass 1: sums[b][row][c] = partial (S1 = sum g, S2 = sum g*raw, Sz = sum dout*z),
 * g = dout * act'(a*raw+b), z = act(a*raw+b) */
int n3d_affine_act_bwd_reduce(const float* dout, int64_t dld, const float* raw, int64_t rld, const float* a,
                              const float* b, int B, int64_t N, int C, int flags, double* sums, void* stream);
/* GroupNorm backward coefficients: dgamma, dbeta (summed over b), dalpha = sum Sz (if not NULL),
 * and the per-(b,c) affine draw = A*g + Bc + Cc*raw.  wptr: the MixedOp weight (NULL -> 1).
 * If dbias_conv != NULL (needs sumraw[b][c] = sum_v raw, saved by the forward coefficient kernels) it also
 * writes the gradient of the bias of the conv that produced raw:  sum_v draw = A*S1 + N*Bc + Cc*sum(raw). */
int n3d_gn_bwd_coeffs(const double* sums, int rows, const float* gamma, const float* mean_rstd,
                      const float* wptr, int B, int C, int G, int64_t N, float* dgamma, float* dbeta,
                      float* dalpha, float* A, float* Bc, float* Cc, const double* sumraw,
                      float* dbias_conv, void* stream);
/* n3d_gn_bwd_coeffs + n3d_affine_act_bwd_apply in one launch for small tensors (rows <= n3d_fused_max_rows()) */
int n3d_affine_act_bwd_apply_gn(const float* dout, int64_t dld, const float* raw, int64_t rld, const float* a,
                                const float* b, const double* sums, int rows, const float* gamma,
                                const float* mean_rstd, const float* wptr, const double* sumraw,
                                float* draw, int64_t drld, int B, int64_t N, int C, int G, int flags, float* dgamma,
                                float* dbeta, float* dalpha, float* dbias_conv, void* stream);
/* ---- node-level pair epilogues: a searched-cell node is op_a(x_i) + op_b(x_j) (searched.py:45-50); both ops end in
 * GroupNorm -> [ReLU] -> weighted sum on tensors of one shape and share the node gradient in backward.  One launch
 * does the work of two n3d_affine_act_gn / n3d_affine_act_bwd_reduce / n3d_affine_act_bwd_apply_gn calls (same
 * results; the node buffer is written once, the node gradient read once).  Requirements: rows <= n3d_fused_max_rows(),
 * C a power of two <= 64, C / G <= 16, B <= 4 for the backward apply; otherwise N3D_ERR_UNSUPPORTED (call the single
 * forms). */
typedef struct n3d_gn_fwd_term {
  const float* raw; int64_t rld;          /* conv output (pitched NDHWC) */
  const double* stats; int32_t rows;      /* [B][rows][C][2] partial (sum, sum of squares) */
  int32_t relu;                           /* 1: ReLU after the norm */
  const float* gamma; const float* beta;  /* GroupNorm affine */
  const float* wptr;                      /* optional scalar weight (MixedOp alpha), NULL = 1 */
  float* a_out; float* b_out; float* mean_rstd_out; double* sumraw;  /* saved for backward, as n3d_affine_act_gn */
  int32_t dtype; int32_t pad_;            /* storage of raw (N3D_F32 / N3D_BF16); entry points with a `flags` argument use N3D_ACT_BF16 */
} n3d_gn_fwd_term;
int n3d_affine_act_gn2(const n3d_gn_fwd_term* t0, const n3d_gn_fwd_term* t1, int G, float eps, float* out, int64_t old_,
                       float* out1 /* NULL: out = term0 + term1 (a node); else two independent outputs (the two preprocess
                       ops of a cell, cell.py:47-50): term0 -> out, term1 -> out1 */, int64_t old1, int B, int64_t N, int C,
                       int flags /* N3D_ACCUMULATE */, void* stream);

typedef struct n3d_gn_bwd_term {
  const float* raw; int64_t rld; const float* a; const float* b;   /* forward operands / coefficients */
  double* sums; int32_t rows;             /* [B][rows][C][3]: written by n3d_affine_act_bwd_reduce2, read by ..._apply_gn2 */
  int32_t relu;
  const float* gamma; const float* mean_rstd; const float* wptr; const double* sumraw;
  float* draw; int64_t drld;              /* d(raw), output of the apply pass */
  float* dgamma; float* dbeta; float* dalpha; float* dbias_conv;   /* parameter gradients (dalpha / dbias_conv may be NULL) */
  float* cA; float* cB; float* cC;        /* [B][C] draw = cA*g + cB + cC*raw: written by n3d_gn_bwd_coeffs2, read by n3d_affine_act_bwd_apply2 */
  int32_t dtype; int32_t pad_;            /* storage of raw / draw and of the output gradient(s) of the call (N3D_F32 / N3D_BF16) */
} n3d_gn_bwd_term;
/* dout1: NULL = both terms share the output gradient dout (a node); else the gradient of term1's own output */
int n3d_affine_act_bwd_reduce2(const float* dout, int64_t dld, const float* dout1, int64_t dld1, const n3d_gn_bwd_term* t0,
                               const n3d_gn_bwd_term* t1, int B, int64_t N, int C, void* stream);
int n3d_affine_act_bwd_apply_gn2(const float* dout, int64_t dld, const float* dout1, int64_t dld1, const n3d_gn_bwd_term* t0,
                                 const n3d_gn_bwd_term* t1, int B, int64_t N, int C, int G, void* stream);
/* Small levels (n3d_bwd_small2_ok: B <= 4, B * roundup(N * (C/G)/4, 64) <= 2048 channel quads per group): the
 * whole epilogue backward of a node whose terms carry no MixedOp weight gradient -- reduction, coefficients, parameter
 * gradients and both d(raw) -- in ONE launch, one workgroup per GroupNorm group, elements held in registers between the passes.
 * Same outputs as n3d_affine_act_bwd_reduce2 + n3d_affine_act_bwd_apply_gn2 (sums / dalpha of the terms are not used). */
int n3d_bwd_small2_ok(int B, int64_t N, int C, int G);
int n3d_affine_act_bwd_small2(const float* dout, int64_t dld, const float* dout1 /* as in ..._reduce2 */, int64_t dld1, const n3d_gn_bwd_term* t0,
                              const n3d_gn_bwd_term* t1, int B, int64_t N, int C, int G, void* stream);
/* The general form (round 3).  n3d_bwd_small_mode: 0 = shape not taken, 1 = one workgroup per group as above, 2 = one workgroup per
 * (group, sample) for B = 2..4 samples of up to 2048 channel quads per group EACH (the 8^3 level at 16 channels per group): d(raw)
 * is per sample; each workgroup stores its parameter-gradient contribution to `scratch` (n3d_bwd_small_scratch_bytes) and the one
 * that draws the last of the group's tickets adds them in sample order.  `tickets`: G zero-initialised words that no other
 * launch in flight uses; they are zero again when the launch is over (atomicInc wrapping at B - 1), so a replayed graph needs
 * no reset.  scratch / tickets may be NULL for mode 1.  t1 == NULL: a single epilogue (dout1 must be NULL then). */
int n3d_bwd_small_mode(int B, int64_t N, int C, int G);
size_t n3d_bwd_small_scratch_bytes(int B, int G);
int n3d_affine_act_bwd_small(const float* dout, int64_t dld, const float* dout1, int64_t dld1, const n3d_gn_bwd_term* t0,
                             const n3d_gn_bwd_term* t1, int B, int64_t N, int C, int G, void* scratch, size_t scratch_bytes,
                             uint32_t* tickets, void* stream);
/* the same pairing for tensors with more partial rows than the fused prologues accept (the 32^3 / 64^3 levels):
 * n3d_gn_coeffs2 = two n3d_gn_coeffs in one launch (fills a_out, b_out, mean_rstd_out, sumraw of both terms);
 * n3d_affine_act2 = two n3d_affine_act into one output (reads a_out / b_out as the coefficients; stats unused);
 * n3d_gn_bwd_coeffs2 = two n3d_gn_bwd_coeffs (fills cA / cB / cC and the parameter gradients);
 * n3d_affine_act_bwd_apply2 = two n3d_affine_act_bwd_apply reading the node gradient once. */
int n3d_gn_coeffs2(const n3d_gn_fwd_term* t0, const n3d_gn_fwd_term* t1, int B, int C, int G, int64_t N, float eps, void* stream);
int n3d_affine_act2(const n3d_gn_fwd_term* t0, const n3d_gn_fwd_term* t1, float* out, int64_t old_, float* out1, int64_t old1, int B,
                    int64_t N, int C, int flags, void* stream);
int n3d_gn_bwd_coeffs2(const n3d_gn_bwd_term* t0, const n3d_gn_bwd_term* t1, int B, int C, int G, int64_t N, void* stream);
int n3d_affine_act_bwd_apply2(const float* dout, int64_t dld, const float* dout1, int64_t dld1, const n3d_gn_bwd_term* t0,
                              const n3d_gn_bwd_term* t1, int B, int64_t N, int C, void* stream);
/* N-term forms for a supernet node (cell.py:76-81: node = sum over its edges of sum_k alpha[e][k] * op_k(x_e); 8-16 of the
 * 10-22 terms end in a GroupNorm).  `terms`: array of n <= N3D_MAX_GROUP_TERMS descriptors, C a power of two in 4..64.
 * n3d_gn_coeffsN: n x n3d_gn_coeffs in one launch.  n3d_affine_actN: out (+)= sum_k w_k * act_k(a_k * raw_k + b_k) in term
 * order, one pass over the node buffer; a_k = a_out, b_k = b_out of the term, NULL meaning 1 / 0, so the node's other
 * primitives (SE gate: a = gate; pooling, identity-with-norm) ride in the same pass.  Backward: n3d_affine_act_bwd_reduceN (fills sums of every term; all terms read the
 * same node gradient; a / b NULL = 1 / 0 as in n3d_affine_act_bwd_reduce, so the other primitives' reductions ride along; this entry and n3d_affine_act_bwd_applyN take up to N3D_MAX_REDUCE_TERMS terms), n3d_gn_bwd_coeffsN (cA / cB / cC and the parameter gradients), n3d_affine_act_bwd_applyN (every draw). */
int n3d_gn_coeffsN(const n3d_gn_fwd_term* terms, int n, int B, int C, int G, int64_t N, float eps, void* stream);
int n3d_affine_actN(const n3d_gn_fwd_term* terms, int n, float* out, int64_t old_, int B, int64_t N, int C, int flags, void* stream);
int n3d_affine_act_bwd_reduceN(const float* dout, int64_t dld, const n3d_gn_bwd_term* terms, int n, int B, int64_t N, int C, void* stream);
int n3d_gn_bwd_coeffsN(const n3d_gn_bwd_term* terms, int n, int B, int C, int G, int64_t N, void* stream);
int n3d_affine_act_bwd_applyN(const float* dout, int64_t dld, const n3d_gn_bwd_term* terms, int n, int B, int64_t N, int C, void* stream);
/* The apply passes of a node level's single primitives (SE gates, identity-with-norm: cell.py:29-32) in ONE launch: every term is
 * draw' = cA * g + cB + cC * raw (g = dout behind the term's ReLU mask a * raw + b > 0 when relu and a are set; NULL a / b / cA / cB / cC
 * = 1 / 0 / 1 / 0 / 0); CONSECUTIVE terms with the same `draw` (at most 4) are summed into it in term order, on top of its previous
 * content if bit 0 of the first term's `pad_` is set -- what n3d_affine_act_bwd_apply launches in that order would leave there.
 * n <= N3D_MAX_REDUCE_TERMS terms, at most 8 distinct targets. */
int n3d_affine_act_bwd_apply_sum(const float* dout, int64_t dld, const n3d_gn_bwd_term* terms, int n, int B, int64_t N, int C, void* stream);
/* plain (no norm) epilogue backward coefficients: A = w, Bc = Cc = 0, dalpha = sum Sz */
int n3d_plain_bwd_coeffs(const double* sums, int rows, const float* wptr, int B, int C, float* dalpha,
                         float* A, void* stream);
/* the same for up to N3D_MAX_GROUP_TERMS primitives of a node in one launch (A or dalpha of a term may be NULL) */
typedef struct n3d_plain_coef_term { const double* sums; int32_t rows; int32_t pad_; const float* wptr; float* dalpha; float* A; } n3d_plain_coef_term;
int n3d_plain_bwd_coeffsN(const n3d_plain_coef_term* terms, int n, int B, int C, void* stream);
/* backward, pass 2: draw (+)= A[b,c]*g + Bc[b,c] + Cc[b,c]*raw   (Bc, Cc may be NULL) */
int n3d_affine_act_bwd_apply(const float* dout, int64_t dld, const float* raw, int64_t rld, const float* a,
                             const float* b, const float* A, const float* Bc, const float* Cc, float* draw,
                             int64_t drld, int B, int64_t N, int C, int flags, void* stream);

/* ---- squeeze-excitation gate (prim_ops.py:133-139,148-152) ----------------------------------------
 * fwd: mean = stats.sum/N; hidden = relu(w1.mean + b1); gate = sigmoid(w2*hidden + b2) */
int n3d_se_gate_fwd(const double* stats, int rows, int64_t N, const float* w1, const float* b1,
                    const float* w2, const float* b2, int B, int C, float* mean, float* hidden, float* gate,
                    void* stream);
/* bwd: dgate[b,c] = w * sums[b][c].S2; returns fc grads and the input-side affine
 * dx = A*g + Bc with A = w*gate, Bc = dmean/N; dalpha = sum Sz if not NULL */
int n3d_se_gate_bwd(const double* sums, int rows, const float* wptr, const float* mean, const float* hidden,
                    const float* gate, const float* w1, const float* w2, int B, int C, int64_t N,
                    float* dw1, float* db1, float* dw2, float* db2, float* dalpha, float* A, float* Bc,
                    void* stream);
/* the gates of up to N3D_MAX_GROUP_TERMS SE primitives of a supernet node (one per stride-1 edge, cell.py:76-81) in one launch.
 * `sums`: forward = the [B][rows][C][2] channel statistics of the gate input, backward = the [B][rows][C][3] reduction rows
 * of n3d_affine_act_bwd_reduce(N); other fields as the arguments of n3d_se_gate_fwd / n3d_se_gate_bwd. */
typedef struct n3d_se_term {
  const double* sums; int32_t rows; int32_t pad_;
  const float* w1; const float* b1; const float* w2; const float* b2;
  float* mean; float* hidden; float* gate;      /* forward outputs, backward inputs */
  const float* wptr; float* dw1; float* db1; float* dw2; float* db2; float* dalpha; float* A; float* Bc;   /* backward only */
} n3d_se_term;
int n3d_se_gate_fwdN(const n3d_se_term* terms, int n, int64_t N, int B, int C, void* stream);
int n3d_se_gate_bwdN(const n3d_se_term* terms, int n, int64_t N, int B, int C, void* stream);
/* All coefficient computations of one node level of the supernet backward (cell.py:76-81: a node's MixedOps all consume the node's
 * gradient) -- n3d_gn_bwd_coeffsN of up to N3D_MAX_REDUCE_TERMS GroupNorm-type terms and n3d_se_gate_bwdN of up to
 * N3D_MAX_GROUP_TERMS SE gates -- in ONE launch at B = 2 with both kinds present (the same per-term work, same results), else as
 * those launches. */
/* forward counterpart: n3d_gn_coeffsN of 1..8 GroupNorm-type terms and n3d_se_gate_fwdN of 1..8 SE gates of one group in one launch */
int n3d_node_fwd_coeffs(const n3d_gn_fwd_term* gn, int n_gn, const n3d_se_term* se, int n_se, int B, int C, int G, int64_t N, float eps, void* stream);
int n3d_node_bwd_coeffs(const n3d_gn_bwd_term* gn, int n_gn, const n3d_se_term* se, int n_se, int B, int C, int G, int64_t N, void* stream);


/* ---- 2x2x2 pooling, stride 2 (prim_ops.py:160-163) ------------------------------------------------ */
int n3d_pool2_fwd(const float* x, int64_t xld, float* y, int64_t yld, int B, int Di, int Hi, int Wi, int C,
                  int flags, void* stream);
int n3d_pool2_bwd(const float* dy, int64_t dyld, const float* x, int64_t xld, float* dx, int64_t dxld, int B,
                  int Di, int Hi, int Wi, int C, int flags, void* stream);
/* dx (+)= w * pool^T(dy): the MixedOp weight of a pooling primitive (cell.py:29-32) folded into its backward (wptr NULL = 1) */
int n3d_pool2_bwd_scaled(const float* dy, int64_t dyld, const float* x, int64_t xld, float* dx, int64_t dxld, int B, int Di, int Hi,
                         int Wi, int C, int flags, const float* wptr, void* stream);
/* average AND max pooling of one tensor (both primitives sit on every stride-2 edge of a down cell, prim_ops.py:29-30; cell.py:16-22):
 * forward in one pass over x; backward dx (+)= w_avg * avgpool^T(dy) + w_max * maxpool^T(dy) in one pass over dx */
int n3d_pool2_fwd_both(const float* x, int64_t xld, float* y_avg, int64_t yald, float* y_max, int64_t ymld, int B, int Di, int Hi, int Wi,
                       int C, void* stream);
int n3d_pool2_bwd_both(const float* dy, int64_t dyld, const float* x, int64_t xld, float* dx, int64_t dxld, int B, int Di, int Hi, int Wi,
                       int C, int flags, const float* w_avg, const float* w_max, void* stream);

/* ---- sigmoid head + Dice loss (nas.py:52, searched.py:93, loss.py:12-14) --------------------------
 * element (b,c,v) of p / t / dp is at  ptr[b*sb + c*sc + v*sv]  (works for NCDHW and NDHWC).
 * partial: double[B][C][n3d_dice_rows(N)][3] scratch; sums: double[B][C][3] (sum p*t, sum p, sum t). */
int n3d_dice_rows(int64_t N);
int n3d_dice_fwd(const float* p, int64_t psb, int64_t psc, int64_t psv, const float* t, int64_t tsb, int64_t tsc,
                 int64_t tsv, int B, int C, int64_t N, float smooth, double* partial, double* sums,
                 float* loss, void* stream);
int n3d_dice_bwd(const float* p, int64_t psb, int64_t psc, int64_t psv, const float* t, int64_t tsb, int64_t tsc,
                 int64_t tsv, int B, int C, int64_t N, float smooth, const double* sums, const float* dloss,
                 float* dp, int64_t dsb, int64_t dsc, int64_t dsv, void* stream);

/* ---- fused head: Dropout3d -> Conv3d(k=1) -> Sigmoid (nas.py:50-52, searched.py:91-93) and, optionally in the same passes,
 * the Dice loss on its output (loss.py:12-14).  x: (B, Ci, N voxels) pitched NDHWC of dtype x_dtype; w: (Co, Ci) = the
 * Conv3d weight (Co, Ci, 1, 1, 1); gate: (B, Ci) Dropout3d gate (0 or 1/(1-p) per sample and channel; prim_ops.py:66,72-73)
 * or NULL.  Ci in {4, 8, 12, 16, 24, 32}, Co <= 4, otherwise N3D_ERR_UNSUPPORTED.  p / logits / t / dp are fp32 with element
 * (b, c, v) at ptr[b*sb + c*sc + v*sv] (NCDHW or NDHWC).
 * n3d_dropout3d_gate: draws the gate on the device: gate[i] = u(seed, counter, i) >= p ? 1/(1-p) : 0 with u =
 *   n3d_dropout3d_uniform (splitmix64, host-callable); state = device uint32[3] {seed_lo, seed_hi, counter}; the launch
 *   increments the counter, so a captured HIP graph draws a new mask on every replay.
 * n3d_head_fwd: p = sigmoid(conv(x * gate) + bias) (logits optionally stored too; p may be NULL when t is given: a training step
 *   needs the loss and the sums only, and the backward pass recomputes p); with t != NULL also
 *   sums[b][c] = (sum p*t, sum p, sum t) and *loss = 1 - mean_bc (2*sum pt + smooth) / (sum p + sum t + smooth);
 *   partial: double[B][Co][n3d_head_rows(N)][3] scratch.
 * n3d_head_bwd: one pass writes dx (+)= and the weight / bias gradient slabs.  Either dp (gradient w.r.t. p) is given, or
 *   (t, sums [, dloss]) and the Dice gradient is formed on the fly.  p is recomputed from x, nothing saved is read.
 *   ws: n3d_head_workspace_bytes(); deferred as in n3d_conv_bwd_weight. */
/* NODE-PLANAR input (round 4): the head's input is the concatenation of a cell's node outputs (cell.py:82, searched.py:51).
 * Written into one (B, Ci) buffer every node epilogue stores 16 bytes on a 48-byte voxel pitch (0.28 of the HBM roofline at
 * 128^3); with node_c > 0 the nodes stay DENSE tensors of their own -- channel c of voxel (b, v) is element
 * x[(c / node_c) * x_node_stride + (b * N + v) * xld + c % node_c], xld >= node_c -- and the head (its only consumer) gathers
 * the Ci / node_c pointers itself; dx of n3d_head_bwd is laid out the same way with dx_node_stride.  node_c == 0: the ordinary
 * pitched layout (the two strides are ignored). */
typedef struct n3d_head {
  const void* x; int64_t xld; int32_t x_dtype; int32_t B; int32_t Ci; int32_t Co; int64_t N;
  const float* w; const float* bias; const float* gate;
  int64_t x_node_stride; int64_t dx_node_stride; int32_t node_c;
  int32_t t_dtype;      /* storage of the targets t: N3D_F32, or N3D_U8 (Co = 3: generator.py:230-248 yields three boolean maps; the strides
                           count elements; same sums and gradients bit for bit, a quarter of the target bytes) */
} n3d_head;
float n3d_dropout3d_uniform(uint64_t seed, uint32_t counter, uint32_t index);
int n3d_dropout3d_gate(uint32_t* state, float p, int B, int C, float* gate, void* stream);
int n3d_head_rows(int64_t N);
size_t n3d_head_workspace_bytes(const n3d_head* h);
int n3d_head_fwd(const n3d_head* h, float* p, int64_t psb, int64_t psc, int64_t psv, float* logits, const void* t, int64_t tsb,
                 int64_t tsc, int64_t tsv, float smooth, double* partial, double* sums, float* loss, void* stream);
int n3d_head_bwd(const n3d_head* h, const float* dp, int64_t dsb, int64_t dsc, int64_t dsv, const void* t, int64_t tsb, int64_t tsc,
                 int64_t tsv, float smooth, const double* sums, const float* dloss, void* dx, int64_t dxld, int dx_dtype, int flags,
                 float* dw, float* dbias, void* ws, size_t ws_bytes, n3d_final_job* deferred, void* stream);

/* ---- layout: NCDHW <-> NDHWC (caller tensors arrive NCDHW: train.py:118-119) ---------------------- */
int n3d_ncdhw_to_ndhwc(const float* src, float* dst, int64_t dld, int B, int C, int64_t N, void* stream);
int n3d_ndhwc_to_ncdhw(const float* src, int64_t sld, float* dst, int B, int C, int64_t N, void* stream);

/* ---- on-device data step (the work of the reference's generator just ahead of the hot path, train.py:117-119):
 * patch crop with zero padding (patches.py:99-115,152-169) + one cube isometry per patch (augment.py:73-131; the same
 * for data and truth) + label expansion to three binary channels (generator.py:230-248, including its quirk that the
 * inclusive "WT" channel is labels {1,2}).  out[b][i] = vol[corner + src(i)], src_a(i) = i[perm[a]] or P-1-i[perm[a]]
 * (flip[a]); zero outside the volume.  vol: (Cv, X, Y, Z) contiguous fp32; truth: (X, Y, Z) uint8 labels {0,1,2,4} or
 * NULL; x_out: pitched NDHWC (B, Cv, P, P, P); t_out: (B, 3, P, P, P) contiguous fp32 -- uint8 with N3D_PATCH_T_U8 -- or NULL.
 * flags: N3D_PATCH_INCLUSIVE (the reference's `inclusive_label`) | N3D_PATCH_T_U8.  descs: HOST array. */
#define N3D_PATCH_MAX_BATCH 64
#define N3D_PATCH_INCLUSIVE 1
#define N3D_PATCH_T_U8 2       /* the three boolean maps as bytes (generator.py:230-248 yields booleans; train.py:118 casts them) */
typedef struct n3d_patch_desc { int32_t corner[3]; int32_t perm[3]; int32_t flip[3]; } n3d_patch_desc;
int n3d_patch_batch(const float* vol, int Cv, const uint8_t* truth, int X, int Y, int Z, const n3d_patch_desc* descs, int B, int P,
                    int flags, float* x_out, int64_t xld, void* t_out, void* stream);

/* ---- step after the hot path (prediction.py:120-170): stitch the per-patch predictions into the brain-wide volume with
 * mean blending (patches.py:172-207) and fuse the three sigmoid channels into one label volume.
 * n3d_stitch: patches element (b, c, voxel v = (lx*P+ly)*P+lz) at patches[b*sb + c*sc + v*sv] (any of the layouts the
 * net produces); corners: DEVICE int32 [B][3] on the brain-wide grid (may be negative / hang over the border);
 * out: float64 (C, FX, FY, FZ) full image, the (X,Y,Z) brain-wide box is written at offset (ox,oy,oz); voxels of the box
 * no patch covers become 0.  Sums run in list order in fp64 like the reference's.  1 <= C <= 4.
 * n3d_tumor_labels: pred float64 (3, N) -> uint8 labels {0,1,2,4} (inclusive: TC/WT/ET channels; else NCR-NET/ED/ET). */
int n3d_stitch(const float* patches, int64_t sb, int64_t sc, int64_t sv, int C, int P, const int32_t* corners, int B, int X, int Y, int Z,
               double* out, int FX, int FY, int FZ, int ox, int oy, int oz, void* stream);
int n3d_tumor_labels(const double* pred, int64_t N, double threshold, int inclusive, uint8_t* out, void* stream);

/* ---- RCCL exchange step of data-parallel training (no reference counterpart: config.yml:54 `multi_gpus` is never read;
 * SURVEY 8(e)).  One process per GPU; the hot path's only exchange is a SUM all-reduce of the flat fp32 gradient buffer, in
 * place, stream-ordered on `stream` (a side HIP stream lets it run under the backward kernels of the next bucket).
 * RCCL is bound with dlopen at first use (the copy already in the process -- torch's -- wins).
 * n3d_comm_unique_id: rank 0 fills 128 bytes (ncclUniqueId) that the host distributes out of band;
 * n3d_comm_init: ncclCommInitRank on the CURRENT HIP device (collective: every rank calls it). */
#define N3D_COMM_ID_BYTES 128
int n3d_comm_available(void);
int n3d_comm_unique_id(void* id_out /* N3D_COMM_ID_BYTES */);
int n3d_comm_init(const void* id, int world, int rank, void** comm_out);
int n3d_comm_allreduce_sum(void* comm, float* buf, int64_t n, void* stream);
/* in-place broadcast of n floats from rank `root` (the one-off weight broadcast when a trainer is built), stream-ordered */
int n3d_comm_broadcast(void* comm, float* buf, int64_t n, int root, void* stream);
int n3d_comm_destroy(void* comm);

/* ---- flat Adam (train.py:49,128; search.py:103-104,228,238): torch.optim.Adam defaults ------------
 * step_ptr: device int32 holding the number of steps already taken; inc_step == 1: a second tiny launch
 * increments it after the update (graph-replay safe); inc_step == 2: step_ptr points to int32[2] =
 * {steps, ticket counter (zero between launches)} and the last workgroup of the update launch itself counts
 * the step (no second launch).  grad_scale multiplies g (DP mean). */
int n3d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  const float* lr_ptr /* if not NULL the learning rate is read from this device float instead of
                  `lr`: a captured graph then follows the ReduceLROnPlateau schedule (train.py:50,77) */,
                  float beta1, float beta2, float eps, float weight_decay, float grad_scale, int32_t* step_ptr,
                  int inc_step, void* stream);

/* Guarded update (round 4): the stream hand-offs below are BOUNDED waits -- one that gives up lets its stream go on with operands
 * that are not there yet.  Such a step must never reach the weights (train.py:121-128: a step is forward, backward, update --
 * there is no "update from garbage" in the reference).  n3d_adam_step_guarded is n3d_adam_step with a test in front, uniform
 * over the grid: if *timeouts != *acked (uint32: time-outs counted by n3d_sync_wait vs. time-outs the host has acknowledged;
 * both NULL = no test) or *peer_flag != 0 (data parallel: the flags of all ranks, SUM-all-reduced together with the
 * gradients; NULL = no test) the launch updates NOTHING -- parameters, moments and the step counter stay as they are -- writes
 * NaN to *loss (if not NULL) and 1 to *host_word (if not NULL; a word from n3d_host_word_alloc, which the host polls without a
 * HIP call).  inc_step must be 0 or 2 when a test is requested.  n3d_guard_flag writes the local test (1.0 / 0.0) to *flag for
 * the all-reduce. */
int n3d_adam_step_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                          const float* lr_ptr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                          int32_t* step_ptr, int inc_step, const void* timeouts, const void* acked, const float* peer_flag,
                          float* loss, void* host_word, void* stream);
int n3d_guard_flag(const void* timeouts, const void* acked, float* flag, void* stream);
/* 64 bytes of pinned, device-mapped, coherent host memory (zeroed); the pointer is valid on the host and in kernels */
int n3d_host_word_alloc(void** host_ptr);
int n3d_host_word_free(void* host_ptr);

/* ---- stream hand-off (no reference counterpart: the reference runs one CUDA stream, train.py:117-128) ----------------
 * Device-side ordering between two HIP streams whose work was launched as SEPARATE graphs: the weight-gradient kernels of
 * a train step run on a side stream next to the backward chain (train.Trainer).  On this stack an event between two graph
 * launches costs ~190 us per hand-off and an intra-graph fork ~19 us per edge without overlap; a flag in device memory costs
 * a ~2 us kernel on each side (tools/handoff_cost.cpp).
 * Protocol: each stream owns a device uint32 STEP counter (starts at 1, bumped once per step by that stream's last sync
 * call, bump != 0).  n3d_sync_signal stores *step to *flag (release, agent scope) behind everything enqueued so far on
 * `stream`; n3d_sync_wait holds `stream` until *flag >= *step (its own stream's counter; wrap-safe compare), polling with
 * one lane.  The poll is bounded: after max_polls tries (~0.3 us each) the wait gives up, adds 1 to *timeouts and lets the
 * stream continue -- results are then wrong, the GPU is never hung; callers read *timeouts (train.Trainer.check_sync).
 * The two streams must map to different hardware queues (a wait at the head of the queue that also carries the signal
 * would time out); train.Trainer probes this once. */
int n3d_sync_signal(void* flag, void* step, int bump, void* stream);
/* diagnostic: stores the 100 MHz wall clock (s_memrealtime) to *out (device uint64) when the stream gets there -- rocprofv3's
 * kernel trace serialises the two streams, so the timeline of the side-stream schedule is taken with these (tools/side_timeline.py) */
int n3d_stamp(void* out, void* stream);
int n3d_sync_wait(const void* flag, void* step, void* timeouts, int bump, int64_t max_polls, void* stream);
/* the same for TWO flags in one launch (both must have reached *step; one time-out is counted if either has not) */
int n3d_sync_wait2(const void* flag0, const void* flag1, void* step, void* timeouts, int bump, int64_t max_polls, void* stream);
/* The side streams and their graphs, made through the HIP runtime libn3d is linked against (the one that launches the kernels):
 * a non-blocking stream of the lowest priority the device offers; thread-local capture of a stream into an instantiated
 * executable graph (the side streams are captured NEXT TO torch's capture of the main stream); launch / destroy. */
int n3d_stream_create_low_priority(void** stream_out);
int n3d_stream_capture_begin(void* stream);
int n3d_stream_capture_end(void* stream, void** graph_exec_out);
int n3d_graph_launch(void* graph_exec, void* stream);
int n3d_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* N3D_H_ */
