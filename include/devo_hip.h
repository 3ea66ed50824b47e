/* libdevo_hip.so — C ABI of the MI355X-native (gfx950) DEVO update + bundle-adjustment hot path.
 *
 * Every entry point replaces one function of the reference's three pybind11 extension modules
 * (cuda_corr, cuda_ba, lietorch_backends); the reference binding it stands in for is cited as
 * file:line relative to the tum-vision/DEVO tree.  Conventions:
 *   - all pointers are DEVICE pointers (HBM) unless marked host; nothing is allocated or freed
 *     inside a call; scratch space is passed in as `ws` (size from the matching *_workspace_bytes);
 *   - every call only ENQUEUES work on `stream` (a hipStream_t; NULL = the legacy default stream
 *     the reference launches on) and never synchronises with the host;
 *   - return value: DEVO_OK (0) or a DEVO_ERR_* code; devo_last_error() gives the message
 *     (the Python layer raises RuntimeError, as the reference's TORCH_CHECK / C++ exceptions do);
 *   - index arrays are int64 (torch.long), as in the reference;
 *   - `dtype`: DEVO_F32 / DEVO_F16 / DEVO_F64 select the element type of the floating tensors
 *     named "T*" below (void* in the signature).
 */
#ifndef DEVO_HIP_H
#define DEVO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* devo_stream_t; /* hipStream_t */

enum { DEVO_OK = 0, DEVO_ERR_ARG = 1, DEVO_ERR_LAUNCH = 2, DEVO_ERR_UNSUPPORTED = 3, DEVO_ERR_WORKSPACE = 4 };
enum { DEVO_F32 = 0, DEVO_F16 = 1, DEVO_F64 = 2 };

#define DEVO_ABI_VERSION 5 /* 2: fp32 split formats (devo_corr_pyramid_split, exponents), group plans (plan buffer tail); 3: per-slot conversions of a ring
                              (devo_corr_pyramid_split_frames, devo_corr_patch_transpose_range), devo_stream_capturing; 4: devo_ba_table_offsets, devo_upd_graph_tables; 5: devo_ba_forward_prepared_delta_plan, devo_ba_import_tables, devo_upd_rs_corr_f16_net32,
                              devo_upd_rs_gru_f16_out32, devo_instnorm_cl, devo_instnorm_bias_cl, devo_bias_act_cl;
                              callers compare with devo_abi_version() */
int devo_abi_version(void);
const char* devo_last_error(void); /* thread-local message of the last failing call */
/* 1 while `stream` is being captured into a HIP graph (a capture executes nothing: a binding must not cache device state a captured call
 * would have produced — the prepared BA tables, a locality plan), 0 when it is not, -1 if the runtime cannot tell. */
int devo_stream_capturing(devo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * altcorr  (reference module cuda_corr: devo/altcorr/correlation.cpp:57-63)
 * ---------------------------------------------------------------------------------------------- */

/* cuda_corr.forward  (correlation.cpp:58 -> correlation_kernel.cu:193-233, kernel :82-136).
 *   fmap1  T [B, Np, C, P, P] contiguous                (patch features, "gmap")
 *   fmap2  T [B, n2, C, H2, W2] with ELEMENT strides f2s[5] = (b, n, c, h, w); channels-last
 *            storage (c stride 1) selects the LDS-staged fast kernel, anything else the generic one
 *   coords f32 [B, E, 2, P, P] contiguous (x then y)
 *   ii, jj i64 [E]  (patch index into fmap1 dim 1, frame index into fmap2 dim 1)
 *   out    T: logical tensor [B, E, D-1 (x offset), D-1 (y offset), P, P], D = 2*radius+2 — i.e. the
 *            reference's permuted result — element (b,e,l) with l the row-major logical index is
 *            written at out[(b*E + e) * out_estride + l * out_lstride + out_offset].
 *            (out_estride = (D-1)^2*P*P, out_lstride = 1, out_offset = 0 gives a contiguous tensor;
 *             out_lstride = 2 and out_offset = level writes straight into the stacked
 *             [B, E, (D-1)^2*P*P, 2] buffer that devo/devo.py:217 / enet.py:216 build with torch.stack.)
 *   order  optional locality plan from devo_corr_order (NULL = process edges in list order).
 *   fmap1_t optional: fmap1 as devo_corr_patch_transpose lays it out.  With it, fp16 lookups into channels-last or channel-blocked
 *          storage (C = 128 / 256) and fp32 lookups into SPLIT-BLOCKED storage (cblock = DEVO_CBLOCK_SPLIT8, devo_corr_pyramid_split;
 *          C % 32 == 0, C <= 128) run as one dense product per edge on v_mfma_f32_16x16x32_f16 (csrc/corr_mm.h; fp32 values enter as
 *          fp16 hi + lo pairs of the value scaled by a power of two per patch / frame: 2^-22 relative per factor at any magnitude, fp32
 *          accumulation).  NULL, or raw fp32 storage: the 4x4 matrix-core kernel with exact fp32 products (csrc/corr_mfma.h) or, for
 *          other layouts, the staged / generic kernels.
 *   fmap2_exps: i32 [B * n2], the scale exponents devo_corr_pyramid_split wrote next to a split-blocked level (NULL otherwise). */
#define DEVO_CBLOCK_SPLIT8 (-8)
int devo_corr_forward(const void* fmap1, const void* fmap2, const float* coords, const int64_t* ii,
                      const int64_t* jj, void* out, int B, int E, int Np, int n2, int C, int P, int H2, int W2,
                      const int64_t* f2s /* host, 5 */, int cblock, int64_t out_estride, int64_t out_lstride,
                      int64_t out_offset, int radius, int dtype, const int* order /* plan buffer of devo_corr_order, i32 [2*B*E + 2], or NULL */,
                      float coord_div /* coords are divided by this in the kernel (correctly rounded IEEE division; pyramid level
                                         scale, 1 = as given; DEVO's scales 1 and 4 are exact either way) */,
                      const void* fmap1_t, const int* fmap2_exps, devo_stream_t stream);

/* Both levels of a 2-level pyramid lookup (devo/devo.py:215-217) in ONE launch: workgroups of the fine and the coarse
 * level alternate on every CU (the fine level waits on memory, the coarse one is LDS/VALU-bound), each writing its
 * slice of the edge's output record (out_offset[l], stride out_lstride).  Same results as two devo_corr_forward
 * calls.  Only for levels the staged kernel reads (channels-last or channel-blocked storage, fp32/fp16); otherwise
 * DEVO_ERR_UNSUPPORTED is returned and nothing is launched. */
int devo_corr_forward_pyramid2(const void* fmap1, const void* fmap2_l0, const void* fmap2_l1, const float* coords,
                               const int64_t* ii, const int64_t* jj, void* out, int B, int E, int Np, int n2, int C,
                               int P, const int* hw /* host: H0, W0, H1, W1 */, const int64_t* f2s /* host: 5 + 5 */,
                               const int* cblock /* host, 2 */, int64_t out_estride, int64_t out_lstride,
                               const int64_t* out_offset /* host, 2 */, int radius, int dtype, const int* order,
                               const float* coord_div /* host, 2 */,
                               const void* fmap1_t /* optional, as for devo_corr_forward: the dense-product kernel does both levels of an
                                                      edge in one wave */,
                               const int* fmap2_exps_l0, const int* fmap2_exps_l1 /* scale exponents of split-blocked levels, or NULL */,
                               int order_kind /* DEVO_PLAN_EDGES: `order` is an edge plan (or NULL); DEVO_PLAN_GROUPS: a group plan
                                                 (devo_corr_order with l1 = 4 on level 0's coordinates) — radius 3, C = 128, level 1 = the
                                                 quarter-resolution level in 8-channel blocks: level 1 is then read from LDS regions shared by
                                                 the edges of a group (csrc/corr_mm.h, the group form); same results */,
                               devo_stream_t stream);

/* fmap1 T [n_patches, C, 3, 3] -> fmap1_t, the patch operand of the dense-product lookup kernel, an opaque buffer of
 * devo_corr_patch_operand_bytes(n_patches, C, dtype) bytes (16-byte aligned) — fp16: the transposed features [n_patches, 9, C]; fp32:
 * [n_patches, 9, C / 8] split records of 32 bytes, fp16 (hi0..7 | lo0..7) with x 2^-e = hi + lo, followed by one scale exponent e (i32) per
 * patch (e puts the patch's largest magnitude into [2^13, 2^14): no magnitude overflows or underflows fp16).  C % 8 == 0.  The patch
 * features of DEVO change once per frame, not per update iteration: convert once, reuse.  DEVO_F32 / DEVO_F16. */
size_t devo_corr_patch_operand_bytes(int n_patches, int C, int dtype);
int devo_corr_patch_transpose(const void* fmap1, void* fmap1_t, int n_patches, int C, int dtype, devo_stream_t stream);
/* The same for patches [first, first + count) of an operand of n_patches (records and exponents of the others untouched): the reference
 * rewrites one frame's patches per frame (`self.gmap_[self.n % self.mem] = gmap`, devo/devo.py:524), so the operand of the ring's
 * patch features is maintained slot by slot. */
int devo_corr_patch_transpose_range(const void* fmap1, void* fmap1_t, int n_patches, int first, int count, int C, int dtype, devo_stream_t stream);

/* fp32 pyramid level -> the SPLIT-BLOCKED format the dense-product lookup kernel multiplies (no reference counterpart; the reference's
 * kernel reads fp32 NCHW, correlation_kernel.cu:82-136).  F frames fmap2 f32 [F, C, H, W] in ANY layout: element strides f2s[4] = (frame,
 * channel or channel block, row, column), cblock > 1 = channel-blocked with cblock contiguous channels per block, else plain strides.
 * dst f32-sized [F, C/8, H, W, 8] (frame stride dst_fstride elements, 16-byte aligned): the strides of an 8-channel blocked level, every
 * 32-byte pixel block holding fp16 (hi0..7 | lo0..7) with x 2^-e = hi + lo, e = exps[f] (one exponent per frame: its largest magnitude
 * lands in [2^13, 2^14)).  exps i32 [2 F]: the exponents, then F ints of scratch.  Two launches (max, convert); the pyramid of DEVO
 * changes once per frame, not per update iteration: convert once per version, pass cblock = DEVO_CBLOCK_SPLIT8 + exps to the lookups. */
int devo_corr_pyramid_split(const void* fmap2, const int64_t* f2s /* host, 4 */, int cblock, int F, int C, int H, int W, void* dst,
                            int64_t dst_fstride, int* exps, devo_stream_t stream);
/* devo_corr_pyramid_split with the F ints of scratch given separately (`exps` i32 [F] receives the exponents only): frames [k, k + F) of a
 * ring buffer are converted in place of their old records — exps = ring_exps + k, dst = ring_dst + k * dst_fstride — while the other
 * frames keep theirs (`self.fmap1_[:, self.n % self.mem] = ...`, devo/devo.py:526-527: one slot per frame). */
int devo_corr_pyramid_split_frames(const void* fmap2, const int64_t* f2s /* host, 4 */, int cblock, int F, int C, int H, int W, void* dst,
                                   int64_t dst_fstride, int* exps, int* scratch /* i32 [F] */, devo_stream_t stream);

/* Locality plan for devo_corr_forward (no reference counterpart: the reference walks edges in list order).
 * order i32 [2*B*E + 2] (the plan buffer; devo_corr_forward reads the first B*E + 1 entries and the last one (number of DEAD edges at
 * the end of the order, 0 for a single-level plan), the rest is scratch of this call): first the HEAVY edge slots (union box of the 9 windows clearly larger than a compact patch's at
 * this radius — more than 128 positions for radius <= 3, more than 256 for radius <= 5 — or larger than the staged
 * kernel's LDS tile: they run 2-4x longer and should start first), then the others sorted by (batch, target frame jj,
 * bin of the patch centre: 16-row bands x 8-px columns, numbered in blocks of 4 bands x ~64 px), so that the lookup
 * kernel's XCD-aware schedule streams every feature row through an L2 about once; order[B*E] = number of heavy
 * edges.  `coord_scale`
 * is the factor the caller divides coords by for the pyramid level whose height is H2 (1 for level 0); one
 * plan serves all levels of a pyramid.  The plan only changes WHICH edges run together, never any result. */
int devo_corr_order(const float* coords, const int64_t* jj, int* order, int B, int E, int n2, int P, int H2,
                    float coord_scale, int radius,
                    int W2 /* width of the plan's level; only read when l1 >= 2 */,
                    int l1 /* 0: edge plan (above).  >= 2 (DEVO: 4): GROUP plan for devo_corr_forward_pyramid2(order_kind = DEVO_PLAN_GROUPS) —
                              the lookup has a second level at 1 / l1 of this resolution, radius 3; `order` is then i32
                              [DEVO_CORR_PLAN_INTS(B*E)].  A group = the edges of one target frame whose patch centre lies in one tile of
                              6 x 6 level-1 cells; the plan sorts the edges by group (same prefix as an edge plan: any lookup accepts it) and
                              stores every group's first slot behind the 2*B*E + 2 ints.  Classes: DEAD edges (union box outside the frame at
                              both levels: every output is 0) go BEHIND all others, order[2*B*E + 1] = their number; HEAVY (in front) = more
                              than 128 level-0 box positions, or a level-1 box that leaves the group's 15 x 15 region (patch pixels more than
                              a level-1 cell from the centre).  DEVO_ERR_UNSUPPORTED when the frames have more than 4095 groups together
                              (or radius != 3): use an edge plan */,
                    devo_stream_t stream);
#define DEVO_CORR_PLAN_TAIL 4104
#define DEVO_CORR_PLAN_INTS(BE) (2 * (BE) + 2 + DEVO_CORR_PLAN_TAIL) /* ints of a plan buffer that can hold a group plan (an edge plan uses the first 2 BE + 2) */
#define DEVO_PLAN_EDGES 0
#define DEVO_PLAN_GROUPS 1

/* Pyramid build for the lookup (devo/devo.py:526-527: fmap1_[slot] = avg_pool2d(fmap, 1, 1), fmap2_[slot] =
 * avg_pool2d(fmap, 4, 4); devo/utils.py:70-79): F frames fmap T [F, C, H, W] (contiguous frames, frame stride
 * fmap_fstride elements) -> channel-blocked level 0  l0 T [F, C/8, H, W, 8] and level 1  l1 T [F, C/8, H/4, W/4, 8]
 * (4x4 mean; NULL = skip), frame strides l0_fstride / l1_fstride so that a frame can be written into a slot of a
 * ring buffer.  One pass over the input.  DEVO_F32 / DEVO_F16 (fp32 accumulation). */
int devo_pyramid_build(const void* fmap, void* l0, void* l1, int F, int C, int H, int W, int64_t fmap_fstride,
                       int64_t l0_fstride, int64_t l1_fstride, int dtype, devo_stream_t stream);

/* cuda_corr.backward  (correlation.cpp:59 -> correlation_kernel.cu:236-286, kernel :139-190).
 *   grad f32: the gradient of the logical [B,E,D-1,D-1,P,P] output, contiguous.
 *   fmap1_grad [B,Np,C,P,P] contiguous, fmap2_grad with the strides of fmap2: both are ZEROED and then
 *   accumulated here.  DEVO_F32 only (the reference's grad accessor is float, :146,280).
 *   ws / ws_bytes: scratch of devo_corr_backward_workspace_bytes(B, E, Np, n2, C, radius, channels_last) bytes, 16-byte aligned (the
 *   window gradients, the frames' window lists, fmap1 transposed: the product form for channels-last fmap2 with C % 128 == 0,
 *   DESIGN.md 3.2).  channels_last = 1 when the C channels of a pixel of fmap2 are contiguous and pixels / rows are C / W*C elements
 *   apart; every other layout takes the one-kernel atomic path and the query returns 0.  NULL or too small: the atomic path as well.
 *   The library never allocates.  devo_corr_backward_last_path(): which path the calling thread's last devo_corr_backward took. */
#define DEVO_CORR_BWD_ATOMIC   0
#define DEVO_CORR_BWD_SEGMENTS 1
#define DEVO_CORR_BWD_PRODUCT  2
size_t devo_corr_backward_workspace_bytes(int B, int E, int Np, int n2, int C, int radius, int channels_last);
int devo_corr_backward_last_path(void);
/* Which kernel the last forward lookup of the calling thread launched: 0 the dense-product kernel (corr_mm.h: the default for blocked fp16 /
 * split-blocked fp32 levels), 4 its group form (level 1 from LDS), 1 the exact-fp32 4x4 matrix-core kernel (raw fp32 blocked / channels-last
 * levels, DEVO_CORR_MM=0), 2 the staged tap-centric kernel (other channel counts, DEVO_CORR_MFMA=0; ~2x), 3 the generic kernel (fp64, raw NCHW,
 * unaligned strides; ~24x); -1 before the first call.  The first call of a process that takes kernel 2 or 3 with >= 2048 edges also says so on
 * stderr (DEVO_LOG_FALLBACK=0: silent). */
int devo_corr_forward_last_path(void);
/* What the last devo_ba_forward* of the calling thread ran: accumulate kind (0 register-resident, N <= 16; 1 general in LDS; 2 global memory, N > 32)
 * + 4 * solve kind (0 one-barrier chain, 6N <= 128; 1 general in LDS; 2 global memory); -1 before the first call.  The first call of a process
 * whose system leaves the LDS says so on stderr (DEVO_LOG_FALLBACK=0: silent). */
int devo_ba_last_path(void);
int devo_corr_backward(const void* fmap1, const void* fmap2, const float* coords, const int64_t* ii,
                       const int64_t* jj, const float* grad, void* fmap1_grad, void* fmap2_grad, int B, int E,
                       int Np, int n2, int C, int P, int H2, int W2, const int64_t* f2s /* host, 5 */,
                       int64_t f2_numel_span /* elements spanned by fmap2 storage */, int radius, int dtype,
                       void* ws, size_t ws_bytes, devo_stream_t stream);

/* cuda_corr.patchify_forward  (correlation.cpp:61 -> correlation_kernel.cu:288-307, kernel :16-47).
 *   net T [B, C, H, W] with element strides ns[4]; coords f32 [B, M, 2]; out T [B, M, C, D, D] contiguous
 *   (zero where the tap is out of bounds). */
int devo_patchify_forward(const void* net, const float* coords, void* out, int B, int M, int C, int H, int W,
                          const int64_t* ns /* host, 4 */, int radius, int dtype, devo_stream_t stream);

/* cuda_corr.patchify_backward  (correlation.cpp:62 -> correlation_kernel.cu:310-333, kernel :49-80).
 *   grad T [B, M, C, D, D] contiguous -> net_grad T [B, C, H, W] (zeroed here) with element strides gs[4] (host; NULL: contiguous; any
 *   dense permutation, e.g. channels-last — the layout the encoders' convolutions take their gradient in). F32/F64. */
int devo_patchify_backward(const float* coords, const void* grad, void* net_grad, int B, int M, int C, int H,
                           int W, const int64_t* gs, int radius, int dtype, devo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * fastba  (reference module cuda_ba: devo/fastba/ba.cpp:152-157)
 * ---------------------------------------------------------------------------------------------- */

/* 0 when the sizes are not supported: N = t1 - t0 <= 128 optimised poses per call (the reference has no limit, ba_cuda.cu:516-522; DEVO
 * uses <= 14).  Up to 32 the reduced system lives in one workgroup's LDS; beyond, in global memory (device-scope atomics like the
 * reference's, the blocked Cholesky in place on the global image). */
size_t devo_ba_workspace_bytes(int E, int Np /* patch slots = patches.shape[1] */, int N /* t1 - t0 */);

/* cuda_ba.forward  (ba.cpp:153 -> ba_cuda.cu:422-540).  IN-PLACE on poses and patches, returns nothing.
 *   poses f32 [Nbuf,7]; patches f32 [Np,3,P,P]; intrinsics f32 [>=1,4] (row 0 only, ba_cuda.cu:233-237);
 *   target, weight f32 [E,2]; lmbda f32 [1]; ii,jj,kk i64 [E]; frames < t0 are fixed; t1-t0 == 0 is the
 *   structure-only branch.  A Cholesky breakdown (the reference's cuSOLVER exception, swallowed by
 *   devo/devo.py:336-340) leaves poses/patches of that iteration untouched and sets *status_flag (i32,
 *   device, may be NULL) to the failing iteration + 1. */
int devo_ba_forward(float* poses, float* patches, const float* intrinsics, const float* target,
                    const float* weight, const float* lmbda, const int64_t* ii, const int64_t* jj,
                    const int64_t* kk, int E, int Nbuf, int Np, int P, int t0, int t1, int iterations, void* ws,
                    size_t ws_bytes, int* status_flag, devo_stream_t stream);

/* The same call split in two, for callers that know the graph before target/weight are ready (in DEVO: before the
 * correlation lookup and the update network, devo/devo.py:213-240): devo_ba_prepare does the index work of
 * ba_cuda.cu:435-437 (unique patches, edges grouped by patch; depends on kk only) — e.g. on a second stream —
 * and devo_ba_forward_prepared runs the Gauss-Newton iterations on the prepared workspace.  A prepared workspace
 * stays valid (any number of forward calls) until kk, E, Np or t1 - t0 change.  devo_ba_forward_prepared on a
 * workspace that was not prepared for this (E, t1 - t0) touches nothing and sets *status_flag to -1. */
int devo_ba_prepare(const int64_t* kk, int E, int Np, int N /* t1 - t0 */, void* ws, size_t ws_bytes,
                    devo_stream_t stream);
/* devo_ba_prepare + the ordering step of the lookup's locality plan (devo_corr_order with coords = NULL) in ONE launch:
 * both are single-workgroup, latency-bound kernels that do not depend on each other, so they run as two workgroups side
 * by side.  plan: i32 [2E + 2] (group plans: [DEVO_CORR_PLAN_INTS(E)]) whose bins devo_transform(..., plan, plan_frames, plan_height,
 * radius, plan_width, plan_l1) has written (batch 1); afterwards it is the finished plan for devo_corr_forward*.  plan_width, plan_l1:
 * the values given to devo_transform (l1 = 0: edge plan). */
int devo_ba_prepare_plan(const int64_t* kk, int E, int Np, int N /* t1 - t0 */, void* ws, size_t ws_bytes, int* plan,
                         int plan_frames, int plan_height, int plan_width, int plan_l1, devo_stream_t stream);
/* Inspection of a prepared workspace (tests / debugging): copies (device to device, any pointer may be NULL)
 * *n_seg = number of distinct patches with edges, kx i32 [min(E,Np)] = their ids ascending (the first output of
 * torch::_unique(kk), ba_cuda.cu:435-437), seg_start i32 [min(E,Np)+1] and perm i32 [E]: the edges of patch kx[s]
 * are perm[seg_start[s] .. seg_start[s+1]) in ascending edge order (the inverse map of _unique, grouped). */
int devo_ba_prepared_tables(const void* ws, size_t ws_bytes, int E, int Np, int N, int* n_seg, int* kx,
                            int* seg_start, int* perm, devo_stream_t stream);
/* The same tables IN PLACE: byte offsets into a prepared workspace of (E, Np, N) — offsets[0] the i32 n_seg, [1] kx, [2] seg_start, [3] perm —
 * and offsets[4] = min(E, Np), the element count of kx (seg_start has one more).  A caller that owns the workspace reads the tables where
 * they lie (devo_amd.update's group tables: eight device copies per frame of DEVO's steady state otherwise).  No device work. */
int devo_ba_table_offsets(int E, int Np, int N, size_t* offsets /* [5] */);
/* The index tables of a kk from a workspace prepared for other sizes of the SAME edge list (src_Np patch slots, src_N optimised poses: e.g.
 * devo_upd_graph_tables' patch-group workspace: its bound, 0) into `ws` (Np, N): one launch instead of devo_ba_prepare's nine — devo.py:311,337
 * hand the same kk to the Update operator and, right behind it, to the BA.  Afterwards `ws` is what devo_ba_prepare(kk, E, Np, N) leaves, provided
 * every id of kk is below Np; a source with ids in [Np, src_Np) leaves `ws` unprepared (the BA then reports status -1). */
int devo_ba_import_tables(const void* src_ws, size_t src_bytes, int src_Np, int src_N, void* ws, size_t ws_bytes, int E, int Np, int N,
                          devo_stream_t stream);
int devo_ba_forward_prepared(float* poses, float* patches, const float* intrinsics, const float* target,
                             const float* weight, const float* lmbda, const int64_t* ii, const int64_t* jj,
                             const int64_t* kk, int E, int Nbuf, int Np, int P, int t0, int t1, int iterations,
                             void* ws, size_t ws_bytes, int* status_flag, devo_stream_t stream);

/* devo_ba_forward_prepared with devo/devo.py:330 (`target = coords[..., P//2, P//2] + delta`) folded in: the target of
 * edge e, component c is coords[e * coords_edge_stride + c * coords_xy_stride + coords_centre] + delta[2e + c] — the same
 * single fp32 addition, so the result is bit-identical to forming `target` first; one elementwise launch less per
 * update iteration.  coords: the f32 buffer devo_transform wrote ([E,2,P,P]: strides 2PP, PP, centre (P/2)(P+1);
 * [E,P,P,2]: strides 2PP, 1, centre 2 (P/2)(P+1)); delta f32 [E,2] (the update operator's output). */
int devo_ba_forward_prepared_delta(float* poses, float* patches, const float* intrinsics, const float* coords,
                                   int coords_edge_stride, int coords_xy_stride, int coords_centre, const float* delta,
                                   const float* weight, const float* lmbda, const int64_t* ii, const int64_t* jj,
                                   const int64_t* kk, int E, int Nbuf, int Np, int P, int t0, int t1, int iterations,
                                   void* ws, size_t ws_bytes, int* status_flag, devo_stream_t stream);

/* devo_ba_forward_prepared_delta, and the ordering step of a locality plan (devo_corr_order / devo_ba_prepare_plan's second half) for the
 * buffer `plan` whose bins devo_transform(..., plan, ...) has written — carried by extra workgroups of the first Gauss-Newton iteration's
 * solver launch (no launch of its own, nothing on the critical path; a separate launch where this call has no such solver launch).  A plan
 * only decides which edges run together: the lookup of update iteration k + 1 (devo.py:308-344 runs them back to back on one patch graph)
 * takes the plan made from iteration k's coordinates, the BA of iteration k carries its ordering.  Same results as the two calls. */
int devo_ba_forward_prepared_delta_plan(float* poses, float* patches, const float* intrinsics, const float* coords,
                                        int coords_edge_stride, int coords_xy_stride, int coords_centre, const float* delta,
                                        const float* weight, const float* lmbda, const int64_t* ii, const int64_t* jj,
                                        const int64_t* kk, int E, int Nbuf, int Np, int P, int t0, int t1, int iterations,
                                        void* ws, size_t ws_bytes, int* status_flag, int* plan, int plan_frames, int plan_height,
                                        int plan_width, int plan_l1, devo_stream_t stream);

/* One differentiable Gauss-Newton step of devo/ba.py:108-170 (normal equations, Schur complement over the patches, damped
 * Cholesky solve) from GIVEN per-edge terms, for devo_amd.ba.BA (the reference builds it with ten matmuls + ten
 * torch_scatter.scatter_sum calls and solves with CholeskySolver, ba.py:12-37).
 * terms f32 [E,30]: r[2] w[2] Jz[2] Ji[2][6] Jj[2][6] with Ji = MINUS d coords / d xi_i (this library's sign convention);
 * frames < t0 or >= t0 + N are fixed; damping S_dd + ep + 1e-4 S_dd (ba.py:73); lmbda f32 [1] on the device.
 * Outputs: dX f32 [6 N] (zero when the factorisation breaks down, like ba.py:16-20; *status_flag = 1 then),
 * dZ f32 [Np] per patch SLOT (zero for patches without an edge).  N <= 32 here.  The workspace (devo_ba_workspace_bytes(E, Np, N)) keeps
 * what the adjoint needs: pass it unmodified to devo_ba_solve_terms_backward. */
int devo_ba_solve_terms(const float* terms, const float* lmbda, const int64_t* ii, const int64_t* jj, const int64_t* kk, int E,
                        int Np, int t0, int N, float ep, void* ws, size_t ws_bytes, float* dX, float* dZ, int* status_flag,
                        devo_stream_t stream);

/* Adjoint of devo_ba_solve_terms (what autograd derives for ba.py:108-170 + CholeskySolver.backward, ba.py:28-37): from the
 * gradients of dX [6 N] and dZ [Np] to the gradient of the 30 terms of every edge, g_terms f32 [E,30].  One more solve with
 * the saved matrix, then per-patch and per-edge kernels. */
int devo_ba_solve_terms_backward(const float* terms, const int64_t* ii, const int64_t* jj, const int64_t* kk, int E, int Np, int t0, int N, void* ws,
                                 size_t ws_bytes, const float* g_dX, const float* g_dZ, float* g_terms, devo_stream_t stream);

/* The update step behind the differentiable solve (devo/ba.py:172-182: `poses.retr(dX)` on the optimised window, inverse depth + dZ clamped to
 * [dmin, dmax] over every patch's P x P pixels, torch.stack) as ONE kernel per direction instead of ~8 / ~12 ATen + SE3 launches; fp32.
 * poses [N,7], patches [Np,3,P,P], dX [6 n_opt] (NULL when n_opt == 0), dZ [Np].  The backward takes the gradients of both outputs (either
 * may be NULL = zero; pose gradients in lietorch's embedding, tangent in the first six of seven) and returns all four input gradients. */
int devo_ba_apply_step(const float* poses, const float* patches, const float* dX, const float* dZ, int N, int Np, int P, int fixedp, int n_opt,
                       float dmin, float dmax, float* poses_out, float* patches_out, devo_stream_t s);
int devo_ba_apply_step_backward(const float* poses, const float* patches, const float* dX, const float* dZ, const float* g_poses_out,
                                const float* g_patches_out, int N, int Np, int P, int fixedp, int n_opt, float dmin, float dmax,
                                float* g_poses, float* g_patches, float* g_dX, float* g_dZ, devo_stream_t s);

/* devo/ba.py:95-106 for the differentiable BA (training): residuals, gate and the 30 per-edge numbers of devo_ba_solve_terms from
 * the outputs of devo_transform(jacobian): coords [E,P,P,2], valid [E], Ji / Jj [E,2,6], Jz [E,2], target / weight [E,2], bounds (host:
 * x0, y0, x1, y1) -> terms [E,30] = r | w | Jz | -Ji | Jj and the gate [E] (kept for the adjoint). */
int devo_ba_edge_terms(const float* coords, const float* valid, const float* Ji, const float* Jj, const float* Jz, const float* target,
                       const float* weight, const float* bounds /* host, 4 */, int E, int P, float* terms, float* gate, devo_stream_t stream);
/* its adjoint: g_terms [E,30] -> g_coords [E,P,P,2] (zeroed here; only the centre pixel carries gradient), g_target, g_weight [E,2],
 * g_Ji, g_Jj [E,2,6], g_Jz [E,2]. */
int devo_ba_edge_terms_backward(const float* g_terms, const float* gate, int E, int P, float* g_coords, float* g_target, float* g_weight,
                                float* g_Ji, float* g_Jj, float* g_Jz, devo_stream_t stream);

/* Adjoint of devo_transform (what autograd derives for devo/projective_ops.py:53-105, second-order terms of the Jacobians
 * included): cotangents of the coordinates g_coords f32 [E,P,P,2|3] ("pp2" layout; NULL = none) and of the centre pixel's
 * Jacobians g_Ji / g_Jj f32 [E,2,6], g_Jz f32 [E,2] (NULL = none) -> g_poses f32 [Nbuf,7] (6-vector of the left perturbation
 * G <- Exp(xi) G in the first six slots, lietorch's convention: lietorch_gpu.cu:174-256) and g_patches f32 [Np,3,P,P]; both are
 * zeroed and accumulated here.  flags as devo_transform (1 = depth channel, 2 = translation only). */
int devo_transform_vjp(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii, const int64_t* jj,
                       const int64_t* kk, const float* g_coords, const float* g_Ji, const float* g_Jj, const float* g_Jz, int E,
                       int Nbuf, int Np, int P, int flags, float* g_poses, float* g_patches, devo_stream_t stream);

size_t devo_neighbors_workspace_bytes(int E);

/* cuda_ba.neighbors  (ba.cpp:154 -> ba.cpp:104-149): for every edge the previous / next edge of the same
 * ii[e] ordered by jj (stable), -1 at the ends.  ix, jx i64 [E] on the device; bit-exact. */
int devo_ba_neighbors(const int64_t* ii, const int64_t* jj, int64_t* ix, int64_t* jx, int E, void* ws,
                      size_t ws_bytes, devo_stream_t stream);

/* cuda_ba.reproject  (ba.cpp:155 -> ba_cuda.cu:543-575, kernel :368-418): coords f32 [E,2,P,P], no Z clamp. */
int devo_ba_reproject(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                      const int64_t* jj, const int64_t* kk, float* coords, int E, int P, devo_stream_t stream);

/* Fused restatement of devo/projective_ops.py:53-105 `transform` (the reprojection DEVO.update really
 * calls, devo/devo.py:222): per-frame intrinsics [n,4], Z clamped at 0.1 in the projection.
 *   coords  f32 [E, P, P, 2 (+1 if depth)]  when coords_pp2 != NULL   (reference layout, :70)
 *   coords_2pp f32 [E, 2, P, P] when != NULL (the permute(0,1,4,2,3).contiguous() of devo.py:223)
 *   valid   f32 [E] or NULL (Z > 0.2 at the centre pixel, :100/:103)
 *   Ji, Jj  f32 [E,2,6], Jz f32 [E,2] or NULL  (:73-98; Ji already negated as in :96)
 *   flags   bit0 = depth, bit1 = tonly
 *   plan    optional locality-plan buffer of the lookup (i32 [2*E + 2], group plans [DEVO_CORR_PLAN_INTS(E)], see devo_corr_order): while the
 *           coordinates are still in registers the kernel writes every edge's plan bin (for a pyramid whose
 *           level 0 has plan_frames frames of plan_height rows, lookup radius plan_radius) into the buffer's
 *           scratch half; devo_corr_order(coords = NULL, ...) then only sorts.  P == 3.  NULL = no plan. */
int devo_transform(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                   const int64_t* jj, const int64_t* kk, float* coords_pp2, float* coords_2pp, float* valid,
                   float* Ji, float* Jj, float* Jz, int E, int P, int flags, int* plan, int plan_frames,
                   int plan_height, int plan_radius, int plan_width, int plan_l1 /* as W2, l1 of devo_corr_order */,
                   devo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * lietorch SE3 subset  (reference module lietorch_backends: devo/lietorch/src/lietorch.cpp:286-316,
 * group_id 3 only; kernels lietorch_gpu.cu:20-294).  X: T [n,7], tangent: T [n,6], gradients of group
 * elements are written to the first 6 of 7 slots (slot 7 = 0).  dtype F32 or F64.
 * ---------------------------------------------------------------------------------------------- */
int devo_se3_exp(const void* a, void* X, int64_t n, int dtype, devo_stream_t s);                    /* :287 expm */
int devo_se3_exp_backward(const void* grad, const void* a, void* da, int64_t n, int dtype, devo_stream_t s);
int devo_se3_log(const void* X, void* a, int64_t n, int dtype, devo_stream_t s);                    /* :290 logm */
int devo_se3_log_backward(const void* grad, const void* X, void* dX, int64_t n, int dtype, devo_stream_t s);
int devo_se3_inv(const void* X, void* Y, int64_t n, int dtype, devo_stream_t s);                    /* :293 inv */
int devo_se3_inv_backward(const void* grad, const void* X, void* dX, int64_t n, int dtype, devo_stream_t s);
int devo_se3_mul(const void* X, const void* Y, void* Z, int64_t n, int dtype, devo_stream_t s);     /* :296 mul */
int devo_se3_mul_backward(const void* grad, const void* X, const void* Y, void* dX, void* dY, int64_t n,
                          int dtype, devo_stream_t s);
int devo_se3_adj(const void* X, const void* a, void* b, int64_t n, int dtype, devo_stream_t s);     /* :299 adj */
int devo_se3_adj_backward(const void* grad, const void* X, const void* a, void* dX, void* da, int64_t n,
                          int dtype, devo_stream_t s);
int devo_se3_adjT(const void* X, const void* a, void* b, int64_t n, int dtype, devo_stream_t s);    /* :302 adjT */
int devo_se3_adjT_backward(const void* grad, const void* X, const void* a, void* dX, void* da, int64_t n,
                           int dtype, devo_stream_t s);
int devo_se3_act(const void* X, const void* p, void* q, int64_t n, int dtype, devo_stream_t s);     /* :305 act */
int devo_se3_act_backward(const void* grad, const void* X, const void* p, void* dX, void* dp, int64_t n,
                          int dtype, devo_stream_t s);
int devo_se3_act4(const void* X, const void* p, void* q, int64_t n, int dtype, devo_stream_t s);    /* :308 act4 */
int devo_se3_act4_backward(const void* grad, const void* X, const void* p, void* dX, void* dp, int64_t n,
                           int dtype, devo_stream_t s);
int devo_se3_as_matrix(const void* X, void* T44, int64_t n, int dtype, devo_stream_t s);            /* :313 as_matrix */
int devo_se3_jinv(const void* X, const void* a, void* b, int64_t n, int dtype, devo_stream_t s);    /* :314 Jinv */

/* ------------------------------------------------------------------------------------------------
 * Update operator pieces (devo/enet.py:32-99, devo/blocks.py:15-48) — the reductions and
 * element-wise steps between the GEMMs (the dense layers are plain library GEMMs issued by the host side).
 * T = DEVO_F32 / DEVO_F16 storage, fp32 arithmetic; rows are contiguous [rows, dim].
 * ---------------------------------------------------------------------------------------------- */

/* out = LayerNorm(x + add1 + add2 + hy[group_of[row]] + sigmoid(gate) * res) * gamma + beta — nn.LayerNorm(dim, eps)
 * (enet.py:47,53-55,65) with the sums in front of it fused in: the residual sums of enet.py:82-83 (add1, add2), the
 * SoftAgg expand of blocks.py:46 (hy + group_of) and the GatedResidual of blocks.py:28-29 (gate with row stride
 * ld_gate, res); every term may be NULL.  Optionally followed by ReLU (enet.py:65-66).  dim <= 1024. */
/* InstanceNorm2d without affine parameters or running statistics (the encoders' norm: devo/extractor.py:27-38) on a CHANNELS-LAST activation
 * x [N, H, W, C] (HW pixels per image), fp32 / fp16, 16-byte aligned, C a multiple of 4 / 8:  y = (x - mean_nc) * rstd_nc, biased variance,
 * rounded to the storage type, then ReLU'd when `relu`; with `res` (same layout; needs relu) y = relu(res + relu(norm(x))) — the tail of a
 * residual block (extractor.py:48-54).  Two launches (partial sums in a fixed order, apply): deterministic.  workspace: devo_instnorm_workspace_bytes. */
size_t devo_instnorm_workspace_bytes(int N, int C);
int devo_instnorm_cl(const void* x, const void* res, void* y, int N, int HW, int C, float eps, int relu, void* workspace, size_t ws_bytes, int dtype,
                     devo_stream_t stream);
/* ... with the convolution's bias [C] (same dtype, 16-byte aligned; or NULL) added in front as ATen adds it behind a bias-free MIOpen call:
 * the norm sees round(x + bias) — the caller leaves the bias out of the convolution (one elementwise launch per convolution less). */
int devo_instnorm_bias_cl(const void* x, const void* bias, const void* res, void* y, int N, int HW, int C, float eps, int relu, void* workspace,
                          size_t ws_bytes, int dtype, devo_stream_t stream);
/* y = act(round(x + bias)) — act = ReLU when `relu` — and, with `res` (needs relu), relu(round(res + y)), on a channels-last activation of
 * `pixels` x C elements (fp32 / fp16, C a multiple of 4 / 8, 16-byte aligned; bias may be NULL): the bias add, ReLU, sum and ReLU behind a
 * convolution of the context encoder (extractor.py:27-54 with norm_fn = 'none') as one launch. */
int devo_bias_act_cl(const void* x, const void* bias, const void* res, void* y, int64_t pixels, int C, int relu, int dtype, devo_stream_t stream);
int devo_upd_layernorm(const void* x, const void* add1, const void* add2, const void* hy, const int* group_of,
                       const void* gate, int64_t ld_gate, const void* res, const void* gamma, const void* beta,
                       void* out, int64_t rows, int dim, float eps, int relu, int dtype, devo_stream_t stream);

/* The adjoint of devo_upd_layernorm without the hy / gate terms (training; the reference differentiates nn.LayerNorm through
 * torch.autograd, enet.py:44,52-56,62): fp32, dim == 384.  y = LN(x + add1 + add2) [ReLU'd], dout = dL/dy  ->  dx (the gradient of x and of
 * either addend) and dgamma += sum_rows dout xhat, dbeta += sum_rows dout (ADDED into buffers the caller has zeroed or is accumulating
 * in).  Mean and variance are recomputed from the inputs: the forward saves nothing but its inputs.
 * partials: NULL — the workgroups add their column sums with float atomics (order, hence the last bits, varies from run to run); or f32
 * [512 * 2 * dim] of scratch — they store them and a second small kernel adds them in workgroup order: bit-reproducible (what
 * torch.use_deterministic_algorithms(True) selects in devo_amd.update). */
int devo_upd_layernorm_backward(const float* x, const float* add1, const float* add2, const float* gamma, const float* beta,
                                const float* dout, float* dx, float* dgamma, float* dbeta, int64_t rows, int dim, float eps, int relu,
                                float* partials, devo_stream_t stream);

/* The graph tables of the Update operator (devo/enet.py:86-95) for one edge list, in one call:
 *   ws_kk  <- devo_ba_prepare(kk, bound, 0): the edges grouped by patch (the scatter_softmax / scatter_sum index of agg_kk, blocks.py:42-43)
 *   ix, jx <- cuda_ba.neighbors(kk, jj) (ba.cpp:104-149) read off those groups (NULL, NULL: skipped); kk must lie in [0, bound)
 *   ws_ij  <- devo_ba_prepare(pair, bound, 0) with pair = (ii - min ii) * (max jj - min jj + 1) + (jj - min jj): the groups of
 *             ii * 12345 + jj (enet.py:94) with keys inside (frames in the window)^2, which must stay below bound
 * ws_kk, ws_ij: devo_ba_workspace_bytes(E, bound, 0) bytes each, read through devo_ba_table_offsets(E, bound, 0); pair_key i64 [E + 2]
 * scratch (holds the keys afterwards).  DEVO's inference hands the operator new index tensors every frame: this is per-frame work. */
int devo_upd_graph_tables(const int64_t* ii, const int64_t* jj, const int64_t* kk, int E, int bound, void* ws_kk,
                          size_t ws_kk_bytes, void* ws_ij, size_t ws_ij_bytes, int64_t* pair_key, int64_t* ix, int64_t* jx,
                          devo_stream_t stream);

/* out[e] = idx[e] >= 0 ? src[idx[e]] : 0   — `mask * net[:, ix]` of enet.py:87-91 (idx from devo_ba_neighbors). */
int devo_upd_masked_gather(const void* src, const int64_t* idx, void* out, int64_t E, int dim, int dtype,
                           devo_stream_t stream);

/* SoftAgg reduction (blocks.py:42-43: torch_scatter.scatter_softmax + scatter_sum over dim 1): for every group s
 * (edges perm[seg_start[s] .. seg_start[s+1]), tables from devo_ba_prepare / devo_ba_prepared_tables on the group key,
 * *n_seg groups) y[s] = sum_e f[e] * softmax_over_group(g)[e], channel-wise.  group_of i32 [E] (optional) receives
 * the group index of every edge for devo_upd_expand_add.  f and g may be column blocks of one wider matrix (the two
 * Linear layers share their input: one GEMM): rows of stride ld_fg.  dim even, operands 8-byte aligned. */
int devo_upd_softagg(const void* f, const void* g, int64_t ld_fg /* row stride of f and g (>= dim) */, const int* perm,
                     const int* seg_start, const int* n_seg, void* y, int* group_of, int64_t E, int dim, int dtype,
                     devo_stream_t stream);
/* ... with the caller's estimate of the rows per group (E / groups; 0 = unknown): groups of >= 48 rows (DEVO's frame pairs) run on 16 waves per
 * workgroup instead of 4 (a wave's rows are a chain of dependent row latencies). */
int devo_upd_softagg_hint(const void* f, const void* g, int64_t ld_fg, const int* perm, const int* seg_start, const int* n_seg, void* y, int* group_of,
                          int64_t E, int dim, int dtype, int rows_per_group, devo_stream_t stream);

/* Adjoint of devo_upd_softagg for training (the backward of blocks.py:42-43's scatter_softmax * f -> scatter_sum): dy [n_seg, dim] ->
 * df, dg (rows of stride ld_d; every edge of every group is written):  d f_e = w_e dy,  d g_e = w_e dy (f_e - y),  w = softmax of g
 * over the edge's group, channel-wise.  Same tables and alignment rules as devo_upd_softagg. */
int devo_upd_softagg_backward(const void* f, const void* g, int64_t ld_fg, const int* perm, const int* seg_start, const int* n_seg,
                              const void* dy, void* df, void* dg, int64_t ld_d, int64_t E, int dim, int dtype, devo_stream_t stream);

/* net[e] += hy[group_of[e]]   — `net + h(y)[:, jx]` of blocks.py:46 / enet.py:93-94, in place. */
int devo_upd_expand_add(void* net, const void* hy, const int* group_of, int64_t E, int dim, int dtype,
                        devo_stream_t stream);

/* out = x + sigmoid(gate) * res   — GatedResidual (blocks.py:28-29) after its three Linear layers. */
int devo_upd_gated_residual(const void* x, const void* gate, int64_t ld_gate /* row stride of gate */, const void* res,
                            void* out, int64_t rows, int dim, int dtype, devo_stream_t stream);

/* Adjoint of devo_upd_gated_residual with respect to the gate pre-activation and res (d x = d out), training:
 * d gate = d out * res * s (1 - s),  d res = d out * s,  s = sigmoid(gate).  dgate / dres contiguous [rows, dim]. */
int devo_upd_gated_residual_backward(const void* gate, int64_t ld_gate, const void* res, const void* dout, void* dgate, void* dres,
                                     int64_t rows, int dim, int dtype, devo_stream_t stream);

/* net[e] = x[e] + sigmoid(gate[e]) * res[e] -> net_out (the last GatedResidual; gate == NULL: net = x, nothing stored),
 * then delta[e] = Wd relu(net[e]) + bd;  weight[e] = sigmoid(Ww relu(net[e]) + bw)   (Wd, Ww [2, dim]; enet.py:68-78). */
int devo_upd_heads(const void* x, const void* gate, int64_t ld_gate, const void* res, void* net_out, const void* Wd,
                   const void* bd, const void* Ww, const void* bw, void* delta, void* weight, int64_t E, int dim,
                   int dtype, devo_stream_t stream);

/* The Update operator's dense layers in fp32 storage on the fp16 matrix cores (csrc/linear.hip): y[M, N] = act(x[M, K] W^T + bias) [+ residual]
 * with every fp32 value split into fp16 hi + lo (2^-22 relative per factor, fp32 accumulation) — the 18 000 / 21 600-row Linear layers of
 * the update operator (enet.py:41-78, blocks.py:15-48) at half the fp32 library GEMM's time.  Both operands are scaled by exact powers
 * of two before the split (weight columns once per version; activation rows by a running scale inside the kernel), so gradient-sized
 * rows (1e-7) keep their 22 bits.  Any N and K (the corr MLP's first layer has K = 882, its dX 882 outputs, the heads 2): the kernel works
 * on column blocks of 96 and K steps of 32, the weight image carries zeros behind N and K.
 *   devo_upd_split_weight: the weight, element (n, k) at W[n * s_n + k * s_k] (s_n = K, s_k = 1: a Linear's [N, K] weight for the forward;
 *     s_n = 1, s_k = N_in: the same storage read as its transpose for dX = dY W), -> wsplit (devo_upd_split_weight_bytes(N, K) =
 *     ceil96(N) ceil32(K) 4 + ceil96(N) 4 bytes — the operand image over whole column blocks of 96, then the inverse column scales —,
 *     16-byte aligned; ALLOCATE WITH THE QUERY, the call takes no size): once per version of the weight.
 *   devo_upd_linear_split: x fp32, rows ldx >= K elements apart (any alignment of 4 bytes); y fp32, rows ldy >= N apart (16-byte pieces when
 *     ldy is a multiple of 4 and y / residual are 16-byte aligned, single values otherwise); bias fp32 [N] or NULL; residual NULL or fp32 with y's row pitch, added after the activation (it may be y itself:
 *     x.add_(linear(t)) in one launch); columns >= relu_from get max(., 0) (0: all of them, >= N: none — a gate | res pair of a
 *     GatedResidual in one launch); gate NULL or fp32 with y's row pitch: the result is kept where gate > 0 and zeroed elsewhere, before
 *     the residual — dX = (dY W) masked by the ReLU output it passes back through (threshold_backward in the epilogue). */
/* The update operator's Linear layers in fp16 storage, second structure (csrc/gemm_rs.hip; enet.py:41-78): a workgroup's rows live in LDS once,
 * the weights go from the L2 into the registers of the wave that multiplies them.  y[M, N] = act(x W^T + bias) [+ residual]; the ReLU applies from
 * column relu_from on (N: none); residual may be y.  N a multiple of 384, K in (352, 384] (devo_upd_rs_supported); x rows 4-byte aligned, y /
 * residual rows 16-byte aligned.  Weight image:
 * devo_upd_rs_pack_weight_f16 (devo_upd_rs_weight_bytes(N, K) bytes, 16-byte aligned), once per weight version. */
size_t devo_upd_rs_weight_bytes(int N, int K);
int devo_upd_rs_supported(int N, int K);
int devo_upd_rs_pack_weight_f16(const void* W /* f16, element (n, k) at W[n * s_n + k * s_k] */, int64_t s_n, int64_t s_k, int N, int K, void* wimage, void* stream);
int devo_upd_rs_linear_f16(const void* x, int64_t ldx, const void* wimage, const void* bias, const void* residual, void* y, int64_t ldy, int M, int N, int K,
                           int relu_from, void* stream);
/* What follows the frame-pair aggregation in the update operator as ONE launch, fp16 storage (enet.py:52-57, 96-99; blocks.py:29-48) — the rows
 * stay in LDS between the layers:  net = LN0(x + hy[group_of]);  net = LN2(net + sigmoid(gate1(net)) res1(net));
 * net_out = net + sigmoid(gate3(net)) res3(net);  delta = d(relu(net_out)), weight = sigmoid(w(relu(net_out))).
 * x, net_out [E, 384] contiguous, hy [groups, 384], delta / weight [E, 2]; wgr*: the [gate[0] | res[0]] weights concatenated to [768, 384], wr2_*:
 * res[2] [384, 384], both as devo_upd_rs_pack_weight_f16 images; bgr* [768], br2_* [384]; Wd / Ww [2, 384]; everything fp16, 16-byte aligned.
 * Every layer output and LayerNorm output is rounded to fp16 where the layer-by-layer path stores it. */
/* fp32 storage in the row-resident structure (csrc/gemm_rs.hip): devo_upd_linear_split's contract and arithmetic (fp32 in and out, every value an
 * exact fp16 hi + lo pair, power-of-two scales per row — the row's own largest magnitude — and per weight column), N a multiple of 384, K a multiple
 * of 4 in (352, 384] (devo_upd_rs_split_supported); x, y, residual, gate rows 16-byte aligned.  Weight image: devo_upd_rs_split_weight
 * (devo_upd_rs_split_weight_bytes(N, K) bytes, 16-byte aligned; element (n, k) at W[n * s_n + k * s_k]: the weight or its transpose view). */
size_t devo_upd_rs_split_weight_bytes(int N, int K);
int devo_upd_rs_split_supported(int N, int K);
int devo_upd_rs_split_weight(const float* W, int64_t s_n, int64_t s_k, int N, int K, void* wimage, void* stream);
int devo_upd_rs_linear_split(const float* x, int64_t ldx, const void* wimage, const float* bias, const float* residual, const float* gate, float* y, int64_t ldy,
                             int M, int N, int K, int relu_from, void* stream);
/* l2(relu(l1(x[gather]))) [+ residual] as one launch with the rows resident in LDS, both layers 384 -> 384 (enet.py:46-50, 86-91: c1 / c2 with
 * their neighbour gather and the sum).  x rows 16-byte aligned (ldx a multiple of 8), residual / y [M, 384] contiguous; gather i64 [M] (negative or
 * >= x_rows: a zero row) or null; weight images of devo_upd_rs_pack_weight_f16; biases 8-byte aligned. */
int devo_upd_rs_mlp2_f16(const void* x, int64_t ldx, int x_rows, const int64_t* gather, const void* w1image, const void* b1, const void* w2image, const void* b2,
                         const void* residual, void* y, int M, void* stream);
/* ... and, on the result rows still in LDS, the 768-wide f | g layer of the SoftAgg that follows (blocks.py:36-40): fg [M, 768] = y [Wf | Wg]^T + [bf | bg]
 * (wfg: the image of the concatenated [768, 384] weight; wfg / bfg / fg all or none). */
int devo_upd_rs_mlp2_fg_f16(const void* x, int64_t ldx, int x_rows, const int64_t* gather, const void* w1image, const void* b1, const void* w2image, const void* b2,
                            const void* residual, void* y, int M, const void* wfg_image, const void* bfg, void* fg, void* stream);
/* x += hy[group_of] (the expand-add behind a SoftAgg, enet.py:93; in place, rounded to fp16 like devo_upd_expand_add) and the f | g layer of the
 * SoftAgg that follows on the same rows, one launch: x [M, 384] contiguous, hy [groups, 384], fg [M, 768]. */
int devo_upd_rs_expand_fg_f16(void* x, const void* hy, const int* group_of, const void* wfg_image, const void* bfg, void* fg, int M, void* stream);
/* The correlation branch and the first LayerNorm of the update operator as ONE launch, fp16 storage (enet.py:59-66, 82-83):
 *   c = l5(relu(LN3(l2(relu(l0(corr))))));  out = LN(net + inp + c).   corr [E, K0], 768 < K0 <= 896 (DEVO: 882), rows 4-byte aligned; net / inp / out
 * [E, 384] contiguous; weight images of devo_upd_rs_pack_weight_f16 ([384, K0], [384, 384], [384, 384]); vectors fp16, 16-byte aligned. */
int devo_upd_rs_corr_f16(const void* corr, int64_t ldc, int K0, const void* w0image, const void* b0, const void* w2image, const void* b2, const void* ln3_w,
                         const void* ln3_b, float eps3, const void* w5image, const void* b5, const void* net, const void* inp, const void* ln_w, const void* ln_b,
                         float eps, void* out, int E, void* stream);
/* ... with `net` as fp32 [E, 384] (the recurrent state as autocast keeps it): it enters net + inp + c unrounded, no conversion pass in front. */
int devo_upd_rs_corr_f16_net32(const void* corr, int64_t ldc, int K0, const void* w0image, const void* b0, const void* w2image, const void* b2, const void* ln3_w,
                               const void* ln3_b, float eps3, const void* w5image, const void* b5, const float* net, const void* inp, const void* ln_w, const void* ln_b,
                               float eps, void* out, int E, void* stream);
int devo_upd_rs_gru_f16(const void* x, const void* hy, const int* group_of, const void* ln0_w, const void* ln0_b, float eps0, const void* wgr1_img,
                        const void* bgr1, const void* wr2_1_img, const void* br2_1, const void* ln2_w, const void* ln2_b, float eps2, const void* wgr3_img,
                        const void* bgr3, const void* wr2_3_img, const void* br2_3, const void* Wd, const void* bd, const void* Ww, const void* bw,
                        void* net_out, void* delta, void* weight, int E, void* stream);
/* ... with net_out as fp32 [E, 384]: the values the fp16 form stores, widened — what devo.py:311's call under autocast returns (the recurrent
 * state stays fp32 there), without a conversion pass behind the launch. */
int devo_upd_rs_gru_f16_out32(const void* x, const void* hy, const int* group_of, const void* ln0_w, const void* ln0_b, float eps0, const void* wgr1_img,
                              const void* bgr1, const void* wr2_1_img, const void* br2_1, const void* ln2_w, const void* ln2_b, float eps2, const void* wgr3_img,
                              const void* bgr3, const void* wr2_3_img, const void* br2_3, const void* Wd, const void* bd, const void* Ww, const void* bw,
                              float* net_out, void* delta, void* weight, int E, void* stream);

/* Linear - ReLU - Linear of the update operator as ONE launch, fp16 storage / fp32 accumulation (csrc/mlp2.hip; enet.py:46-50 c1 / c2,
 * :59-61 the corr MLP's first two layers): y[r] = (residual[r] +) W2 relu(W1 x[src(r)] + b1) + b2 with both layers 384 wide, any K1 (W1 is
 * [384, K1]); src(r) = r, or gather[r] (i64 [M]; negative = a zero row: `mask * net[:, ix]`).  The 384-wide intermediate stays in LDS, rounded
 * to fp16 like the two-launch form's.  Weight images: devo_upd_mlp2_pack_weight (devo_upd_mlp2_weight_bytes(K) bytes, 16-byte aligned), once per
 * version of a weight.  x: rows ldx elements apart (even), x_rows of them; y / residual: rows ldy apart (multiple of 8), 16-byte aligned; y must
 * not alias x when gather is given. */
size_t devo_upd_mlp2_weight_bytes(int K);
int devo_upd_mlp2_pack_weight(const void* W /* f16, element (n, k) at W[n * s_n + k * s_k], n < 384 */, int64_t s_n, int64_t s_k, int K, void* wimage,
                              devo_stream_t stream);
int devo_upd_mlp2_f16(const void* x, int64_t ldx, int x_rows, const int64_t* gather, const void* w1image, const void* b1, int K1, const void* w2image,
                      const void* b2, const void* residual, void* y, int64_t ldy, int M, devo_stream_t stream);

size_t devo_upd_split_weight_bytes(int N, int K);
int devo_upd_split_weight(const float* W, int64_t s_n, int64_t s_k, int N, int K, void* wsplit, devo_stream_t stream);
int devo_upd_linear_split(const float* x, int64_t ldx, const void* wsplit, const float* bias, const float* residual, const float* gate, float* y,
                          int64_t ldy, int M, int N, int K, int relu_from, devo_stream_t stream);

/* The same workgroup shape for fp16 storage — the update operator's inference precision (devo.py:71-77: autocast): y[M, N] (fp16) =
 * act(x[M, K] W^T + bias) [+ residual], fp32 accumulation, one MFMA per block (no split, no scales), K steps of 64.
 *   devo_upd_pack_weight_f16: W fp16, element (n, k) at W[n * s_n + k * s_k] -> wimage (devo_upd_pack_weight_f16_bytes(N, K) = N ceil64(K) 2
 *     bytes, 16-byte aligned): the B-operand image, once per version of the weight.
 *   devo_upd_linear_f16: x fp16 rows ldx >= K elements apart (a multiple of 2, 4-byte aligned), y / residual fp16 rows ldy apart (a
 *     multiple of 4, 8-byte aligned), bias fp16 [N] or NULL; relu_from / residual as devo_upd_linear_split.  N % 96 == 0, any K. */
size_t devo_upd_pack_weight_f16_bytes(int N, int K);
int devo_upd_pack_weight_f16(const void* W, int64_t s_n, int64_t s_k, int N, int K, void* wimage, devo_stream_t stream);
int devo_upd_linear_f16(const void* x, int64_t ldx, const void* wimage, const void* bias, const void* residual, void* y, int64_t ldy, int M,
                        int N, int K, int relu_from, devo_stream_t stream);

/* The third product of a Linear layer's training step (csrc/linear_dw.hip): dW[No, Ni] = dY[R, No]^T X[R, Ni] and db[No] = the column sums of
 * dY (db may be NULL), fp32 in and out on the fp16 matrix cores like devo_upd_linear_split (exact hi + lo splits, per-column running
 * power-of-two scales) — what torch.autograd computes for the reference's nn.Linear layers (enet.py:41-78, blocks.py:15-48) as
 * `grad_output.t() @ input` and `grad_output.sum(0)`.  Any No, Ni and row pitches (the kernel works on 128 x 128 blocks: the corr MLP's 882
 * inputs are 7 block columns, the last one masked); tensors 4-byte aligned, the workspace (devo_upd_dw_workspace_bytes(R, No, Ni) bytes:
 * the row slices' partial blocks — no atomics, reproducible) 16-byte aligned. */
size_t devo_upd_dw_workspace_bytes(int R, int No, int Ni);
int devo_upd_dw_split(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, int R, int No, int Ni, void* workspace, float* dW,
                      int64_t ld_dw, float* db, devo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Event voxelisation — the step in front of the encoders.
 * ---------------------------------------------------------------------------------------------- */

/* to_voxel_grid (utils/event_utils.py:180-232): N events (x, y f32 pixel coordinates, t f64 ascending timestamps,
 * p i8 polarity with 0 meaning -1) vote trilinearly into grid f32 [bins, H, W] (zeroed here; corners outside are
 * dropped).  Weights in fp64 like the reference, accumulation with fp32 atomics (order differs: fp32 rounding). */
int devo_voxelize(const float* xs, const float* ys, const double* ts, const signed char* ps, int64_t N, int H, int W,
                  int bins, float* grid, devo_stream_t stream);

/* std (utils/voxel_utils.py:6-28, devo/devo.py:438-452), in place: vox f32 [nseg, len] — nseg = b (sequence-wise) or
 * b*n (frame-wise); every segment's NON-ZERO entries are standardised with the mean / stddev of those entries; nothing
 * is changed if any segment has no non-zero entry.  ws: devo_voxel_std_workspace_bytes(nseg). */
size_t devo_voxel_std_workspace_bytes(int nseg);
int devo_voxel_std(float* vox, int nseg, int64_t len, void* ws, size_t ws_bytes, devo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEVO_HIP_H */
