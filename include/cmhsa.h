/* cmhsa.h -- C ABI of the MI355X-native fp32 causal multi-head self-attention forward.
 *
 * This is the drop-in boundary under the reference's torch op
 *   causal_multihead_self_attention::causal_multihead_self_attention_torch(Tensor Q, Tensor K,
 *                                                    Tensor V, int num_heads) -> Tensor
 * (reference torch_extensions/causal_multihead_self_attention/causal_multihead_self_attention.cu
 *  :464-525, abbreviated CMHSA.cu below).  The library (libcmhsa_hip.so) holds hand-written
 * gfx950 kernels only; it has no torch types, allocates nothing, never synchronises and is
 * re-entrant.  The torch binding (llm-inference_amd/csrc/torch_binding.cpp) does validation,
 * torch::empty and the current-stream lookup exactly where CMHSA.cu:464-515 does them.
 *
 * All pointers are DEVICE pointers to fp32; strides are in ELEMENTS (floats).
 */
#ifndef CMHSA_H
#define CMHSA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes (0 = success; negative = argument error; positive = hipError_t of a failed
 * runtime call).  Replaces CMHSA.cu:24-31 (gpuAssert -> exit) and :432-449
 * (std::invalid_argument): nothing here throws or exits. */
#define CMHSA_OK                 0
#define CMHSA_ERR_NULL_POINTER  -1   /* a tensor pointer is NULL while its size is non-zero        */
#define CMHSA_ERR_BAD_SHAPE     -2   /* batch/seq_len/num_heads negative, num_heads == 0, negative stride */
#define CMHSA_ERR_HEAD_DIM      -3   /* head_dim != 64          (CMHSA.cu:432 "Head dimension must be 64") */
#define CMHSA_ERR_ALIGNMENT     -4   /* pointer not 16-B aligned or stride % 4 != 0 (CMHSA.cu:480-486)     */
#define CMHSA_ERR_TOO_LARGE     -5   /* grid or 32-bit tile index would overflow                   */
#define CMHSA_ERR_WORKSPACE     -6   /* workspace too small                                         */

/* Human-readable text for a status code returned by this library (static storage). */
const char *cmhsa_status_string(int status);

/* Library/ABI version: (major << 16) | minor. */
int cmhsa_version(void);

/* Causal multi-head self-attention forward, fp32 in / fp32 out, head_dim = 64.
 *
 * Replaces the launcher causal_multihead_self_attention() (CMHSA.cu:409-462) and the kernel it
 * launches (CMHSA.cu:64-405).  Per (batch b, head h):
 *     O[b, :, 64h:64h+64] = softmax( mask( Q_h K_h^T ) / sqrt(64) ) V_h ,   key j <= query i.
 *
 *   q, k, v    : (batch, seq_len, num_heads*64) views; element (b, t, c) is at
 *                ptr[b*batch_stride + t*row_stride + c]   (column stride 1, as CMHSA.cu:480-482).
 *                In the model they are column slices of one (T, 3*d_model) buffer with
 *                row_stride = 3*d_model (reference causal_self_attention.py:224-227).
 *   o          : same indexing; every element of (batch, seq_len, num_heads*64) is written,
 *                so the buffer may come from torch::empty (CMHSA.cu:498).
 *   batch      : the reference supports only batch 1 (causal_self_attention.py:103-106); here it is
 *                an extra, independent grid dimension.  batch*_stride is ignored when batch == 1.
 *   head_dim   : must be 64 (CMHSA.cu:432).  d_model = num_heads*64 is NOT restricted to the
 *                reference's {768,1024,1280,1600} (CMHSA.cu:435-449): any num_heads >= 1 works.
 *   stream     : a hipStream_t (NULL = the legacy default stream the reference launches on,
 *                CMHSA.cu:437).  The launch is asynchronous; no host synchronisation.
 *
 * seq_len == 0 or batch == 0 is a successful no-op.  Inputs are read-only and must not alias o.
 */
int cmhsa_fp32_fwd(const float *q, const float *k, const float *v, float *o,
                   int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                   int64_t q_batch_stride, int64_t k_batch_stride, int64_t v_batch_stride, int64_t o_batch_stride,
                   int batch, int seq_len, int num_heads, int head_dim,
                   void *stream);

/* The same forward with a WORKSPACE -- the fastest entry point on grids of about one round of the chip's CUs
 * (BASELINE config 2: B=1, H=12, T=1024).  The reference's grid is one block per (query tile, head) with no
 * inter-block communication (CMHSA.cu:425); with only device memory of the caller's this library can likewise only
 * give whole query blocks to workgroups, and the longest block then sets the kernel's time.  Given a workspace, the
 * planner may cut a query block's KEYS into two ranges held by different workgroups; each stores its un-normalised
 * (O, m, l) to the workspace and the workgroup that finishes last merges them (no workgroup waits for another, the
 * merge order is fixed: results are bitwise reproducible and within rounding of cmhsa_fp32_fwd's).
 *
 *   workspace        : device memory, 256-B aligned, hipMalloc'd (or from a pool of such memory, e.g. torch's caching
 *                      allocator).  CMHSA_WORKSPACE_BYTES always suffices; cmhsa_fp32_fwd_workspace_bytes() gives
 *                      the exact need of one shape (0 for shapes whose plan splits nothing).
 *   workspace_bytes  : its size.  NULL / 0 / too short is not an error: the call then uses the plan that needs none
 *                      (= cmhsa_fp32_fwd).
 * Before its FIRST use (and after a launch that did not run to completion) the workspace's counter block must be
 * zero: cmhsa_workspace_init().  Every completed launch leaves it zero again.  Calls that overlap in time on one GPU
 * must not share a workspace (one per stream). */
#define CMHSA_WORKSPACE_BYTES ((size_t)(16384 + 1024 * (2048 + 64) * 4))     /* 8 667 136 B */
int cmhsa_fp32_fwd_ws(const float *q, const float *k, const float *v, float *o,
                      int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                      int64_t q_batch_stride, int64_t k_batch_stride, int64_t v_batch_stride, int64_t o_batch_stride,
                      int batch, int seq_len, int num_heads, int head_dim,
                      void *workspace, size_t workspace_bytes, void *stream);

/* The same forward for callers that cannot allocate -- code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE (its launcher needs no scratch memory, CMHSA.cu:409-462).
 * It is cmhsa_fp32_fwd_ws with the workspace taken from a static array in device memory that the library's code
 * object carries: CMHSA_NOALLOC_SCRATCH_BYTES (1 MiB) on every GPU the library runs on, held until the process ends.
 * The call allocates nothing (no hipMalloc / hipFree, no allocating library routine); the array's address on the
 * current device is looked up once per device (hipGetSymbolAddress, at the first call on that device).  The first call
 * for a (device, shape) also builds the shape's plan and may raise a kernel's dynamic-LDS limit: host-side set-up that does
 * not belong in a capture -- make ONE un-captured call per (device, shape) before capturing it (tests/test_zz_no_alloc.py
 * captures and replays the call after such a warm-up).
 *   - A shape whose plan needs more scratch than the array holds is rejected with CMHSA_ERR_WORKSPACE (nothing is
 *     launched); cmhsa_fp32_fwd_workspace_bytes() tells the need in advance.  BASELINE config 2 needs 0.8 MiB.
 *   - Every call on a GPU shares that GPU's array: calls that OVERLAP in time on one GPU (different streams) race.
 *     The kernel itself leaves the array's arrival counters zero, so back-to-back calls need no re-initialisation.
 *   - Results equal cmhsa_fp32_fwd_ws's bit for bit (same plan, same arithmetic); the merging workgroup additionally
 *     runs an agent-scope acquire, which costs about a microsecond on shapes that split query blocks. */
#define CMHSA_NOALLOC_SCRATCH_BYTES ((size_t)1048576)
int cmhsa_fp32_fwd_noalloc(const float *q, const float *k, const float *v, float *o,
                           int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                           int64_t q_batch_stride, int64_t k_batch_stride, int64_t v_batch_stride, int64_t o_batch_stride,
                           int batch, int seq_len, int num_heads, int head_dim, void *stream);

/* CMHSA_WORKSPACE_BYTES as a function (for bindings that cannot read the macro). */
size_t cmhsa_workspace_bytes(void);

/* Zeroes the workspace's counter block (the first 16 KiB) on `stream`; call once after allocating it.  Also the one place
 * where the memory's kind is checked: anything but ordinary (coarse-grained, hipMalloc'd) device memory -- managed,
 * fine-grained, host-mapped or unregistered -- is refused with CMHSA_ERR_WORKSPACE, because the hand-off between the
 * workgroups that share a split query block relies on write-through stores reaching the memory side (see cmhsa_fp32_fwd_ws). */
int cmhsa_workspace_init(void *workspace, size_t workspace_bytes, void *stream);

/* Workspace bytes the plan of this shape uses on the current device (0: the plan splits no query block). */
size_t cmhsa_fp32_fwd_workspace_bytes(int batch, int seq_len, int num_heads);

/* Diagnostics: writes a one-line description of the plan cmhsa_fp32_fwd (with_workspace = 0) or cmhsa_fp32_fwd_ws
 * with a full-size workspace (1) uses for this shape on the current device into buf (NUL-terminated). */
int cmhsa_plan_describe(int batch, int seq_len, int num_heads, int with_workspace, char *buf, int buf_len);

/* Diagnostics / tests (no reference counterpart; touches no GPU when cus > 0): the whole plan of a shape as integers, so
 * that a host-side test can check that every 32-key unit of every 32-row query block is computed exactly once.
 * `cus` = compute units to plan for (0: the current device's).  Returns the number of ints the dump needs (> 0; nothing
 * is written when out is NULL or cap is smaller) or a negative CMHSA_ERR_*.  Layout:
 *   out[0..15]  waves per workgroup | segments per record (2 or 4) | workgroups per (b,h) | grid x | y | z |
 *               part slots per (b,h) | split blocks per (b,h) | SIMD cap | query blocks | table plan? |
 *               longest-first group size (0 = no) | chain cap | n_paired | n_alone | workspace bytes / 256
 *   then per workgroup j of a (b,h), per segment: n (units of the query block = block index + 1; 0 = absent) |
 *               first unit | unit count | wave mask | part word (0 = whole block, else first part slot | part index << 8 |
 *               parts << 12 | split id << 16).
 * The waves of a segment's mask take its units round-robin in ascending wave order; one-wave workgroups run their
 * segments one after the other, whole.
 * Two families have layouts of their own, told by out[10]:
 *   out[10] == 2, the 16-key-tile kernel (grids of one round): out[0] waves per workgroup (4 or 8) | out[1] job words per
 *               wave (32 / waves) | out[2] workgroups per (b,h) | out[3..5] grid | out[8] tiles on the busiest SIMD |
 *               out[9] query blocks | out[12] 16-row halves per block (1 or 2) | out[13] XCD-contiguous 1-D grid? | out[14]
 *               granule g (first tile and tiles of a job word count g tiles each: 1, or 2 beyond T = 4080); then per
 *               workgroup and per job list in run order its job words: block | first tile << 8 | tiles << 16 | rank of the
 *               piece << 24 | pieces of the block << 27 | (first word of a list) the workgroup merges << 31; tiles == 0 =
 *               no job.  Tile t of block b covers keys [16 t, 16 t + 16), block b the query rows [16 s b, 16 s (b + 1)),
 *               s = out[12].
 *   out[10] == 3, job lists (one-wave workgroups that run whole blocks of several (b,h)s): out[3..5] grid = (8, lists,
 *               ceil(periods / 8)) | out[8] units of the longest list | out[9] query blocks | out[12] (b,h)s per period |
 *               out[13] lists per period | out[14] jobs of the longest list | out[15] blocks of at most that many units are
 *               taken from the diagonal unit down; then 8 words per list = 16 half-word entries: units of the block |
 *               (b,h) within the period << 8; 0 = end.  Period P holds the (b,h)s [P out[12], (P + 1) out[12]) that exist. */
int cmhsa_plan_dump(int batch, int seq_len, int num_heads, int with_workspace, int cus, int *out, int cap);

/* Convenience form of the reference's exact contract: 2-D (N, d_model) tensors, batch 1
 * (CMHSA.cu:476-478, :488-489, :500-512).  Equivalent to cmhsa_fp32_fwd(..., batch = 1). */
int cmhsa_fp32_fwd_2d(const float *q, const float *k, const float *v, float *o,
                      int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                      int seq_len, int d_model, int num_heads,
                      void *stream);

/* The same forward over ONE packed QKV buffer, as the model produces it: row t of batch b holds
 * [ Q (d_model) | K (d_model) | V (d_model) ] -- the output of the fused QKV linear layer that the
 * reference slices into three views before calling the op (reference causal_self_attention.py:224-227).
 * Equivalent to cmhsa_fp32_fwd(qkv, qkv + d_model, qkv + 2*d_model, o, rs, rs, rs, o_row_stride, bs, bs,
 * bs, o_batch_stride, ...) with d_model = num_heads * head_dim; one pointer and one stride pair to
 * marshal instead of three.  Same status codes; a row stride below
 * 3*d_model is CMHSA_ERR_BAD_SHAPE.  workspace / workspace_bytes as in cmhsa_fp32_fwd_ws (NULL, 0 = none). */
int cmhsa_fp32_fwd_packed_qkv(const float *qkv, float *o,
                              int64_t qkv_row_stride, int64_t o_row_stride,
                              int64_t qkv_batch_stride, int64_t o_batch_stride,
                              int batch, int seq_len, int num_heads, int head_dim,
                              void *workspace, size_t workspace_bytes, void *stream);

/* One-token ("decode") step over a KV cache -- opt-in, SURVEY.md section 8(f) row 4.  The reference keeps no
 * KV cache (README.md:12-13; sampler.py:14-19 re-runs the full forward per token); this entry point computes
 * only the LAST row of the same formula (causal_self_attention.py:173-194) for a token whose q, k, v have just
 * come out of the fused QKV layer, and appends that k, v to the cache in the same launch.  Per (b, h):
 *     k_cache[b, pos, 64h:64h+64] = k_new ;  v_cache[b, pos, ...] = v_new
 *     o[b, 64h:64h+64] = softmax( q_new . K[b, 0..pos]^T / 8 ) V[b, 0..pos]
 *   qkv_new   : (batch, 3*d_model), row b = [ q | k | v ] of the new token (row stride >= 3*d_model)
 *   k_cache,
 *   v_cache   : (batch, capacity, d_model), element (b, t, c) at ptr[b*cache_batch_stride + t*cache_row_stride + c];
 *               rows 0..pos-1 hold the earlier tokens (e.g. written by a prefill through cmhsa_fp32_fwd)
 *   o         : (batch, d_model), row stride o_row_stride
 *   pos_dev   : DEVICE pointer to one int32 = index of the new token, 0 <= pos < capacity.  It is read by the
 *               kernel, not by the host, so a captured HIP graph of the step can be replayed for every token; a
 *               value outside [0, capacity) reads no key and leaves the caches alone; o is set to zero (so that a
 *               caller never sees uninitialised memory and needs no fill of its own in front of the launch).
 * HBM/L2-bound; no allocation, no synchronisation; same status codes as cmhsa_fp32_fwd. */
int cmhsa_fp32_decode_step(const float *qkv_new, int64_t qkv_row_stride,
                           float *k_cache, float *v_cache, int64_t cache_row_stride, int64_t cache_batch_stride,
                           float *o, int64_t o_row_stride, const int *pos_dev,
                           int batch, int capacity, int num_heads, int head_dim, void *stream);

/* Algorithmic HBM bytes of one decode step with num_keys = pos + 1 keys:
 * 4 * batch * d_model * (2*num_keys + 6)  (every K and V row once, q/k/v in, the appended rows and o out). */
double cmhsa_decode_algorithmic_bytes(int batch, int num_keys, int num_heads, int head_dim);

/* Measurement aid (no counterpart in the reference): runs fp32 MFMA chains on every CU for `iters` x 64 MFMAs per
 * wave and stores, from one wave, { s_memtime at start, at end, s_memrealtime at start, at end } into out4_dev
 * (4 x uint64, DEVICE memory supplied by the caller).  s_memtime ticks are shader-clock cycles, s_memrealtime runs
 * at 100 MHz, so (t1 - t0) / (r1 - r0) * 100 MHz is the shader clock the chip SUSTAINS under matrix load -- the
 * clock every "% of peak" figure silently assumes (bench.py prints it next to the 2.4 GHz of the peak constant). */
int cmhsa_probe_shader_clock(unsigned long long *out4_dev, int iters, void *stream);

/* Algorithmic work of one call, the figure every TFLOP/s number in this repo uses
 * (SURVEY.md section 8(d)):  F = 2 * batch * num_heads * head_dim * T * (T+1). */
double cmhsa_algorithmic_flops(int batch, int seq_len, int num_heads, int head_dim);

#ifdef __cplusplus
}
#endif
#endif /* CMHSA_H */
