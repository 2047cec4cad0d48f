// torch_binding.cpp -- thin PyTorch-ROCm binding over the C ABI of include/cmhsa.h.
//
// Mirrors the reference's torch-op boundary (reference torch_extensions/
// causal_multihead_self_attention/causal_multihead_self_attention.cu:464-525, "CMHSA.cu"):
//   * importable top-level module `causal_multihead_self_attention` whose import registers the op
//     (reference causal_self_attention.py:60-63, test_torch_extension.py:3; empty module body as
//     CMHSA.cu:39);
//   * schema `causal_multihead_self_attention::causal_multihead_self_attention_torch(Tensor Q,
//     Tensor K, Tensor V, int num_heads) -> Tensor` (CMHSA.cu:517-521) -- argument NAMES are part of
//     the ABI because the call site uses keywords (causal_self_attention.py:144);
//   * implementation registered for dispatch key CUDA (CMHSA.cu:523-525), which is the key HIP
//     tensors carry under PyTorch-ROCm.
// Differences, all supersets: (B, T, d_model) inputs are accepted next to the reference's 2-D
// (N, d_model); the launch goes on torch's CURRENT stream instead of the legacy default stream
// (CMHSA.cu:437); device errors raise instead of exit()ing the process (CMHSA.cu:24-31).
// All compute is in libcmhsa_hip.so; nothing here touches tensor data.

#include <Python.h>

#include <ATen/ATen.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <hip/hip_runtime_api.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "cmhsa.h"
#include "gpt2_gelu.h"
#include "gpt2_layernorm.h"
#include "gpt2_step.h"
#include "gpt2_linear.h"

namespace causal_multihead_self_attention {

static void check_one(const at::Tensor &t, const char *name, int64_t dim) {
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor");                       // CMHSA.cu:468-470
    TORCH_CHECK(t.dtype() == at::kFloat, name, " must be float32");                 // CMHSA.cu:472-474
    TORCH_CHECK(t.dim() == dim, name, " must be a ", dim, "D tensor");              // CMHSA.cu:476-478
    TORCH_CHECK(t.stride(dim - 1) == 1, name, "'s last-dim stride must be 1 for float4 loads to work");      // :480-482
    TORCH_CHECK(t.stride(dim - 2) % 4 == 0, name, "'s row stride must be a multiple of 4 for float4 loads to work");  // :484-486
    if (dim == 3) TORCH_CHECK(t.stride(0) % 4 == 0, name, "'s batch stride must be a multiple of 4 for float4 loads to work");
}

// Workspace of cmhsa_fp32_fwd_ws (query blocks split across workgroups), one per (device, stream): calls on one stream
// are ordered, calls on different streams must not share it.
//   * Sized by need: cmhsa_fp32_fwd_workspace_bytes(shape) rounded up to 256 KiB, grown when a larger shape arrives (most
//     shapes need none: no entry is made for them).  At most kMaxEntries (device, stream) pairs are kept; beyond that the
//     least recently used one that no graph capture has seen is dropped.
//   * Under graph capture nothing is allocated and nothing is memset: a capturing call uses its stream's workspace if
//     one of sufficient size exists ALREADY -- made by an earlier eager call on that stream, e.g. the warm-up step that
//     torch's capture recipe runs first, or by reserve_workspace() -- and otherwise the plan that needs none (the same
//     function, another summation order: results within rounding of each other, bitwise reproducible either way).
//   * A captured graph has the workspace's address baked in and owns it from then on: such an entry is never dropped or
//     replaced in place (a larger one is added beside it and the old buffer kept alive), and two graphs captured on one
//     stream must not be replayed concurrently on different streams -- they share that stream's workspace.
namespace {
struct WorkspaceEntry {
    at::Tensor buf;
    bool seen_by_capture = false;
    uint64_t last_use = 0;
};
constexpr size_t kMaxEntries = 16;
constexpr size_t kWorkspaceGranule = 256 * 1024;
std::mutex g_ws_mu;
std::map<std::pair<int, hipStream_t>, WorkspaceEntry> g_ws_cache;
std::vector<at::Tensor> g_ws_retired;            // outgrown buffers that a captured graph may still point at
uint64_t g_ws_tick = 0;
}  // namespace

static void *stream_workspace(int device, hipStream_t stream, size_t need, size_t *bytes) {
    *bytes = 0;
    if (need == 0) return nullptr;
    std::lock_guard<std::mutex> lock(g_ws_mu);
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &status) != hipSuccess || status != hipStreamCaptureStatusNone;
    auto it = g_ws_cache.find({device, stream});
    const bool have = it != g_ws_cache.end() && (size_t)it->second.buf.numel() >= need;
    if (!have) {
        if (capturing) return nullptr;               // no allocation, no memset node in a capture
        const size_t size = (need + kWorkspaceGranule - 1) / kWorkspaceGranule * kWorkspaceGranule;
        at::Tensor ws = at::empty({(int64_t)size}, at::TensorOptions().dtype(at::kByte).device(at::kCUDA, device));
        const int rc = cmhsa_workspace_init(ws.data_ptr(), (size_t)ws.numel(), stream);
        TORCH_CHECK(rc == CMHSA_OK, "cmhsa_workspace_init failed (", rc, "): ", cmhsa_status_string(rc));
        if (it != g_ws_cache.end()) {
            if (it->second.seen_by_capture) g_ws_retired.push_back(it->second.buf);
            it->second.buf = ws;
        } else {
            if (g_ws_cache.size() >= kMaxEntries) {
                auto victim = g_ws_cache.end();
                for (auto e = g_ws_cache.begin(); e != g_ws_cache.end(); ++e)
                    if (!e->second.seen_by_capture && (victim == g_ws_cache.end() || e->second.last_use < victim->second.last_use)) victim = e;
                if (victim != g_ws_cache.end()) g_ws_cache.erase(victim);      // (its stream's queued launches keep the memory: torch frees stream-ordered)
            }
            it = g_ws_cache.emplace(std::make_pair(device, stream), WorkspaceEntry{ws, false, 0}).first;
        }
    }
    it->second.seen_by_capture |= capturing;
    it->second.last_use = ++g_ws_tick;
    *bytes = (size_t)it->second.buf.numel();
    return it->second.buf.data_ptr();
}

// Makes sure that the CURRENT stream of `like`'s device has a workspace large enough for this shape (outside capture), so
// that a capture on that stream later runs the plans that split query blocks.  Returns the bytes the shape needs (0: none).
int64_t reserve_workspace(at::Tensor like, int64_t batch, int64_t seq_len, int64_t num_heads) {
    TORCH_CHECK(like.is_cuda(), "like must be a CUDA tensor (it names the device)");
    TORCH_CHECK(batch >= 0 && seq_len >= 0 && num_heads > 0 && batch <= INT32_MAX && seq_len <= INT32_MAX && num_heads <= INT32_MAX, "bad shape");
    const c10::hip::HIPGuard guard(like.device().index());
    const hipStream_t stream = c10::hip::getCurrentHIPStream(like.device().index()).stream();
    const size_t need = cmhsa_fp32_fwd_workspace_bytes((int)batch, (int)seq_len, (int)num_heads);
    size_t got = 0;
    stream_workspace(like.device().index(), stream, need, &got);
    TORCH_CHECK(got >= need, "reserve_workspace must be called outside graph capture");
    return (int64_t)need;
}

at::Tensor causal_multihead_self_attention_torch(at::Tensor Q, at::Tensor K, at::Tensor V, int64_t num_heads) {
    const int64_t dim = Q.dim();
    TORCH_CHECK(dim == 2 || dim == 3, "Q must be a 2D tensor (or 3D: batch, sequence, feature)");
    check_one(Q, "Q", dim);
    check_one(K, "K", dim);
    check_one(V, "V", dim);
    TORCH_CHECK(K.device() == Q.device() && V.device() == Q.device(), "Q, K and V must be on the same device");

    const int64_t B = dim == 3 ? Q.size(0) : 1;
    const int64_t N = Q.size(dim - 2);
    const int64_t d = Q.size(dim - 1);
    TORCH_CHECK(K.size(dim - 2) == N, "K must have the same sequence length as Q");      // CMHSA.cu:491-494
    TORCH_CHECK(V.size(dim - 2) == N, "V must have the same sequence length as Q");
    TORCH_CHECK(K.size(dim - 1) == d, "K must have the same feature dimension as Q");
    TORCH_CHECK(V.size(dim - 1) == d, "V must have the same feature dimension as Q");
    if (dim == 3) TORCH_CHECK(K.size(0) == B && V.size(0) == B, "K and V must have the same batch size as Q");
    TORCH_CHECK(num_heads > 0 && d % num_heads == 0,
                "Feature dimension of Q must be evenly divisible by the number of heads");   // CMHSA.cu:496
    TORCH_CHECK(d / num_heads == 64, "Head dimension must be 64.");                         // CMHSA.cu:432-434
    TORCH_CHECK(B <= INT32_MAX && N <= INT32_MAX && num_heads <= INT32_MAX, "shape exceeds 32-bit indexing");

    at::Tensor output = at::empty(Q.sizes(), Q.options().memory_format(c10::MemoryFormat::Contiguous));  // CMHSA.cu:498
    if (output.numel() == 0) return output;

    const c10::hip::HIPGuard guard(Q.device().index());
    const hipStream_t stream = c10::hip::getCurrentHIPStream(Q.device().index()).stream();
    auto bs = [&](const at::Tensor &t) -> int64_t { return dim == 3 ? t.stride(0) : 0; };
    size_t ws_bytes = 0;
    void *ws = stream_workspace(Q.device().index(), stream, cmhsa_fp32_fwd_workspace_bytes((int)B, (int)N, (int)num_heads), &ws_bytes);
    const int rc = cmhsa_fp32_fwd_ws(Q.data_ptr<float>(), K.data_ptr<float>(), V.data_ptr<float>(), output.data_ptr<float>(),
                                     Q.stride(dim - 2), K.stride(dim - 2), V.stride(dim - 2), output.stride(dim - 2),
                                     bs(Q), bs(K), bs(V), bs(output),
                                     (int)B, (int)N, (int)num_heads, (int)(d / num_heads), ws, ws_bytes, stream);
    TORCH_CHECK(rc == CMHSA_OK, "cmhsa_fp32_fwd_ws failed (", rc, "): ", cmhsa_status_string(rc));
    return output;
}

// The same op over the packed output of the fused QKV linear layer (reference causal_self_attention.py:224-227
// slices it into three views first): QKV is (N, 3*d_model) or (B, N, 3*d_model), row = [Q | K | V].
// SURVEY.md 8(f) row 2; no reference schema to mirror -- the reference signature above stays the compatibility entry.
at::Tensor causal_multihead_self_attention_packed_qkv(at::Tensor QKV, int64_t num_heads) {
    const int64_t dim = QKV.dim();
    TORCH_CHECK(dim == 2 || dim == 3, "QKV must be a 2D (N, 3*d_model) or 3D (B, N, 3*d_model) tensor");
    check_one(QKV, "QKV", dim);
    TORCH_CHECK(num_heads > 0 && QKV.size(dim - 1) % (3 * num_heads) == 0,
                "last dimension of QKV must be 3 * d_model with d_model divisible by the number of heads");
    const int64_t d = QKV.size(dim - 1) / 3, B = dim == 3 ? QKV.size(0) : 1, N = QKV.size(dim - 2);
    TORCH_CHECK(d / num_heads == 64, "Head dimension must be 64.");
    TORCH_CHECK(B <= INT32_MAX && N <= INT32_MAX && num_heads <= INT32_MAX, "shape exceeds 32-bit indexing");
    std::vector<int64_t> out_sizes(QKV.sizes().begin(), QKV.sizes().end());
    out_sizes.back() = d;
    at::Tensor output = at::empty(out_sizes, QKV.options().memory_format(c10::MemoryFormat::Contiguous));
    if (output.numel() == 0) return output;
    const c10::hip::HIPGuard guard(QKV.device().index());
    const hipStream_t stream = c10::hip::getCurrentHIPStream(QKV.device().index()).stream();
    size_t ws_bytes = 0;
    void *ws = stream_workspace(QKV.device().index(), stream, cmhsa_fp32_fwd_workspace_bytes((int)B, (int)N, (int)num_heads), &ws_bytes);
    const int rc = cmhsa_fp32_fwd_packed_qkv(QKV.data_ptr<float>(), output.data_ptr<float>(), QKV.stride(dim - 2), output.stride(dim - 2),
                                             dim == 3 ? QKV.stride(0) : 0, dim == 3 ? output.stride(0) : 0,
                                             (int)B, (int)N, (int)num_heads, 64, ws, ws_bytes, stream);
    TORCH_CHECK(rc == CMHSA_OK, "cmhsa_fp32_fwd_packed_qkv failed (", rc, "): ", cmhsa_status_string(rc));
    return output;
}

// One-token decode step over a KV cache (opt-in; the reference has no KV cache, so there is no
// reference schema): appends the new token's k, v to the caches at row `pos` and returns its context row.
//   qkv_new (B, 3*d_model) | k_cache, v_cache (B, capacity, d_model), updated in place | pos: int32 device tensor, 1 element
at::Tensor causal_multihead_self_attention_decode(at::Tensor qkv_new, at::Tensor k_cache, at::Tensor v_cache, at::Tensor pos,
                                                  int64_t num_heads) {
    for (const at::Tensor *t : {&qkv_new, &k_cache, &v_cache}) {
        TORCH_CHECK(t->is_cuda() && t->dtype() == at::kFloat, "qkv_new, k_cache and v_cache must be float32 CUDA tensors");
        TORCH_CHECK(t->stride(t->dim() - 1) == 1, "last-dim stride must be 1");
    }
    TORCH_CHECK(pos.is_cuda() && pos.dtype() == at::kInt && pos.numel() == 1, "pos must be an int32 CUDA tensor with one element");
    TORCH_CHECK(k_cache.device() == qkv_new.device() && v_cache.device() == qkv_new.device() && pos.device() == qkv_new.device(),
                "qkv_new, k_cache, v_cache and pos must be on the same device");
    TORCH_CHECK(qkv_new.dim() == 2 && k_cache.dim() == 3 && v_cache.dim() == 3, "qkv_new must be (B, 3*d_model), caches (B, capacity, d_model)");
    TORCH_CHECK(num_heads > 0 && qkv_new.size(1) % (3 * num_heads) == 0, "last dimension of qkv_new must be 3 * d_model");
    const int64_t B = qkv_new.size(0), d = qkv_new.size(1) / 3, cap = k_cache.size(1);
    TORCH_CHECK(d / num_heads == 64, "Head dimension must be 64.");
    TORCH_CHECK(k_cache.sizes() == v_cache.sizes() && k_cache.strides() == v_cache.strides(), "k_cache and v_cache must have the same shape and strides");
    TORCH_CHECK(k_cache.size(0) == B && k_cache.size(2) == d, "cache shape must be (B, capacity, d_model)");
    TORCH_CHECK(B <= INT32_MAX && cap <= INT32_MAX, "shape exceeds 32-bit indexing");
    // no fill in front of the launch: the kernel itself zeroes o when pos is outside [0, capacity) (include/cmhsa.h)
    at::Tensor output = at::empty({B, d}, qkv_new.options());
    if (output.numel() == 0) return output;
    const c10::hip::HIPGuard guard(qkv_new.device().index());
    const hipStream_t stream = c10::hip::getCurrentHIPStream(qkv_new.device().index()).stream();
    const int rc = cmhsa_fp32_decode_step(qkv_new.data_ptr<float>(), qkv_new.stride(0), k_cache.data_ptr<float>(), v_cache.data_ptr<float>(),
                                          k_cache.stride(1), k_cache.stride(0), output.data_ptr<float>(), output.stride(0),
                                          pos.data_ptr<int>(), (int)B, (int)cap, (int)num_heads, 64, stream);
    TORCH_CHECK(rc == CMHSA_OK, "cmhsa_fp32_decode_step failed (", rc, "): ", cmhsa_status_string(rc));
    return output;
}

// Fused tanh-GELU (reference gelu.py:4-12 is a plain nn.Module with no extension op, so
// this op has no reference schema to mirror): any-shape contiguous fp32 device tensor in, new tensor out.
at::Tensor gelu_tanh(at::Tensor x) {
    TORCH_CHECK(x.is_cuda(), "x must be a CUDA tensor");
    TORCH_CHECK(x.dtype() == at::kFloat, "x must be float32");
    const at::Tensor xc = x.contiguous();
    at::Tensor y = at::empty_like(xc);
    if (y.numel() == 0) return y;
    const c10::hip::HIPGuard guard(x.device().index());
    const hipStream_t stream = c10::hip::getCurrentHIPStream(x.device().index()).stream();
    const int rc = gpt2_gelu_tanh_f32_fwd(xc.data_ptr<float>(), y.data_ptr<float>(), xc.numel(), stream);
    TORCH_CHECK(rc == CMHSA_OK, "gpt2_gelu_tanh_f32_fwd failed (", rc, "): ", cmhsa_status_string(rc));
    return y;
}

// Fused residual add + LayerNorm (include/gpt2_layernorm.h; the reference has nn.LayerNorm and a torch add,
// transformer_block.py:26-39, no extension op to mirror): x (..., d) and optionally delta of the same shape, fp32 device
// tensors whose last dimension is contiguous and whose leading dimensions collapse to one row stride (else copied).
// Returns (x + delta, LayerNorm(x + delta)); without delta the first is x itself.
std::tuple<at::Tensor, at::Tensor> add_layer_norm(at::Tensor x, const c10::optional<at::Tensor> &delta, at::Tensor weight,
                                                  at::Tensor bias, double eps) {
    TORCH_CHECK(x.is_cuda() && weight.is_cuda() && bias.is_cuda(), "x, weight and bias must be CUDA tensors");
    TORCH_CHECK(x.dtype() == at::kFloat && weight.dtype() == at::kFloat && bias.dtype() == at::kFloat, "tensors must be float32");
    TORCH_CHECK(x.dim() >= 1, "x must have at least one dimension");
    const int64_t d = x.size(-1);
    TORCH_CHECK(weight.numel() == d && bias.numel() == d, "weight and bias must have x.size(-1) elements");
    TORCH_CHECK(weight.device() == x.device() && bias.device() == x.device(), "all tensors must be on the same device");
    const at::Tensor xc = x.contiguous(), wc = weight.contiguous(), bc = bias.contiguous();
    at::Tensor dc;
    if (delta.has_value()) {
        TORCH_CHECK(delta->is_cuda() && delta->dtype() == at::kFloat && delta->sizes() == x.sizes() && delta->device() == x.device(),
                    "delta must be a float32 CUDA tensor of x's shape on x's device");
        dc = delta->contiguous();
    }
    at::Tensor y = at::empty_like(xc);
    at::Tensor sum = delta.has_value() ? at::empty_like(xc) : xc;
    if (xc.numel() == 0) return {sum, y};
    const c10::hip::HIPGuard guard(x.device().index());
    const hipStream_t stream = c10::hip::getCurrentHIPStream(x.device().index()).stream();
    const int rc = gpt2_add_layernorm_f32_fwd(xc.data_ptr<float>(), delta.has_value() ? dc.data_ptr<float>() : nullptr, wc.data_ptr<float>(),
                                              bc.data_ptr<float>(), delta.has_value() ? sum.data_ptr<float>() : nullptr, y.data_ptr<float>(),
                                              xc.numel() / d, (int)d, d, d, (float)eps, stream);
    TORCH_CHECK(rc == CMHSA_OK, "gpt2_add_layernorm_f32_fwd failed (", rc, "): ", cmhsa_status_string(rc));
    return {sum, y};
}

// Few-row linear layer (include/gpt2_linear.h; torch.nn.functional.linear for x of 1..8 rows, optionally followed by the
// reference's tanh-GELU, gelu.py:8-12): x (..., k) with at most 8 rows in all, weight (n_out, k), bias (n_out)? -> (..., n_out).
at::Tensor linear_rows(at::Tensor x, at::Tensor weight, const c10::optional<at::Tensor> &bias, int64_t act) {
    TORCH_CHECK(x.is_cuda() && weight.is_cuda() && x.dtype() == at::kFloat && weight.dtype() == at::kFloat, "x and weight must be float32 CUDA tensors");
    TORCH_CHECK(x.dim() >= 1 && weight.dim() == 2 && x.size(-1) == weight.size(1), "x must be (..., k) and weight (n_out, k)");
    TORCH_CHECK(weight.device() == x.device(), "x and weight must be on the same device");
    const int64_t k = weight.size(1), n_out = weight.size(0);
    const at::Tensor xc = x.contiguous(), wc = weight.stride(1) == 1 ? weight : weight.contiguous();
    const int64_t rows = k ? xc.numel() / k : 0;
    TORCH_CHECK(rows <= GPT2_LINEAR_MAX_ROWS, "linear_rows is for at most ", GPT2_LINEAR_MAX_ROWS, " rows of x; use torch.nn.functional.linear beyond");
    at::Tensor bc;
    if (bias.has_value()) {
        TORCH_CHECK(bias->is_cuda() && bias->dtype() == at::kFloat && bias->numel() == n_out && bias->device() == x.device(),
                    "bias must be a float32 CUDA tensor with n_out elements on x's device");
        bc = bias->contiguous();
    }
    std::vector<int64_t> out_sizes(x.sizes().begin(), x.sizes().end());
    out_sizes.back() = n_out;
    at::Tensor y = at::empty(out_sizes, x.options().memory_format(c10::MemoryFormat::Contiguous));
    if (y.numel() == 0) return y;
    TORCH_CHECK(n_out <= INT32_MAX && k <= INT32_MAX, "shape exceeds 32-bit indexing");
    const c10::hip::HIPGuard guard(x.device().index());
    const hipStream_t stream = c10::hip::getCurrentHIPStream(x.device().index()).stream();
    const int rc = gpt2_linear_rows_f32_fwd(xc.data_ptr<float>(), k, wc.data_ptr<float>(), wc.stride(0), bias.has_value() ? bc.data_ptr<float>() : nullptr,
                                            y.data_ptr<float>(), n_out, (int)rows, (int)n_out, (int)k, (int)act, stream);
    TORCH_CHECK(rc == CMHSA_OK, "gpt2_linear_rows_f32_fwd failed (", rc, "): ", cmhsa_status_string(rc));
    return y;
}

// The same with the residual add and the LayerNorm of the input in front (include/gpt2_linear.h): returns
// (x + delta, act(LayerNorm(x + delta) W^T + b)); without delta the first is x itself.  At most 4 rows of x.
std::tuple<at::Tensor, at::Tensor> ln_linear_rows(at::Tensor x, const c10::optional<at::Tensor> &delta, at::Tensor ln_weight, at::Tensor ln_bias,
                                                  double eps, at::Tensor weight, const c10::optional<at::Tensor> &bias, int64_t act) {
    for (const at::Tensor *t : {&x, &ln_weight, &ln_bias, &weight})
        TORCH_CHECK(t->is_cuda() && t->dtype() == at::kFloat && t->device() == x.device(), "tensors must be float32 CUDA tensors on one device");
    TORCH_CHECK(x.dim() >= 1 && weight.dim() == 2 && x.size(-1) == weight.size(1), "x must be (..., k) and weight (n_out, k)");
    const int64_t k = weight.size(1), n_out = weight.size(0);
    TORCH_CHECK(ln_weight.numel() == k && ln_bias.numel() == k, "ln_weight and ln_bias must have k elements");
    const at::Tensor xc = x.contiguous(), wc = weight.stride(1) == 1 ? weight : weight.contiguous(), gc = ln_weight.contiguous(), bec = ln_bias.contiguous();
    const int64_t rows = k ? xc.numel() / k : 0;
    TORCH_CHECK(rows <= GPT2_LN_LINEAR_MAX_ROWS && k <= GPT2_LN_LINEAR_MAX_K, "ln_linear_rows is for at most ", GPT2_LN_LINEAR_MAX_ROWS, " rows of at most ", GPT2_LN_LINEAR_MAX_K, " floats");
    at::Tensor dc, bc;
    if (delta.has_value()) {
        TORCH_CHECK(delta->is_cuda() && delta->dtype() == at::kFloat && delta->sizes() == x.sizes() && delta->device() == x.device(),
                    "delta must be a float32 CUDA tensor of x's shape on x's device");
        dc = delta->contiguous();
    }
    if (bias.has_value()) {
        TORCH_CHECK(bias->is_cuda() && bias->dtype() == at::kFloat && bias->numel() == n_out && bias->device() == x.device(),
                    "bias must be a float32 CUDA tensor with n_out elements on x's device");
        bc = bias->contiguous();
    }
    std::vector<int64_t> out_sizes(x.sizes().begin(), x.sizes().end());
    out_sizes.back() = n_out;
    at::Tensor y = at::empty(out_sizes, x.options().memory_format(c10::MemoryFormat::Contiguous));
    at::Tensor sum = delta.has_value() ? at::empty_like(xc) : xc;
    if (y.numel() == 0 || xc.numel() == 0) return {sum, y};
    TORCH_CHECK(n_out <= INT32_MAX, "shape exceeds 32-bit indexing");
    const c10::hip::HIPGuard guard(x.device().index());
    const hipStream_t stream = c10::hip::getCurrentHIPStream(x.device().index()).stream();
    const int rc = gpt2_ln_linear_rows_f32_fwd(xc.data_ptr<float>(), k, delta.has_value() ? dc.data_ptr<float>() : nullptr, k, gc.data_ptr<float>(),
                                               bec.data_ptr<float>(), (float)eps, delta.has_value() ? sum.data_ptr<float>() : nullptr, wc.data_ptr<float>(),
                                               wc.stride(0), bias.has_value() ? bc.data_ptr<float>() : nullptr, y.data_ptr<float>(), n_out,
                                               (int)rows, (int)n_out, (int)k, (int)act, stream);
    TORCH_CHECK(rc == CMHSA_OK, "gpt2_ln_linear_rows_f32_fwd failed (", rc, "): ", cmhsa_status_string(rc));
    return {sum, y};
}

// The generation loop's input side (include/gpt2_step.h; reference gpt2.py:43-44): tokens (B, L) int64 (a view of the
// static token buffer: any row stride), wte (vocab, d), wpe (context, d) -> (B, L, d) = wte[tokens] + wpe[arange(L)].
at::Tensor embed_tokens(at::Tensor tokens, at::Tensor wte, at::Tensor wpe) {
    TORCH_CHECK(tokens.is_cuda() && wte.is_cuda() && wpe.is_cuda(), "tokens, wte and wpe must be CUDA tensors");
    TORCH_CHECK(tokens.dtype() == at::kLong && tokens.dim() == 2 && (tokens.size(1) <= 1 || tokens.stride(1) == 1), "tokens must be (B, L) int64 with contiguous rows");
    TORCH_CHECK(wte.dtype() == at::kFloat && wpe.dtype() == at::kFloat && wte.dim() == 2 && wpe.dim() == 2 && wte.is_contiguous() && wpe.is_contiguous(),
                "wte and wpe must be contiguous float32 matrices");
    TORCH_CHECK(wte.size(1) == wpe.size(1) && tokens.size(1) <= wpe.size(0), "embedding widths differ or the sequence is longer than the context");
    TORCH_CHECK(wte.device() == tokens.device() && wpe.device() == tokens.device(), "all tensors must be on the same device");
    const int64_t B = tokens.size(0), L = tokens.size(1), d = wte.size(1);
    at::Tensor x = at::empty({B, L, d}, wte.options());
    if (x.numel() == 0) return x;
    const c10::hip::HIPGuard guard(tokens.device().index());
    const hipStream_t stream = c10::hip::getCurrentHIPStream(tokens.device().index()).stream();
    const int rc = gpt2_embed_f32_fwd(tokens.data_ptr<int64_t>(), tokens.stride(0), wte.data_ptr<float>(), wpe.data_ptr<float>(), x.data_ptr<float>(),
                                      (int)B, (int)L, (int)d, (int)wte.size(0), stream);
    TORCH_CHECK(rc == CMHSA_OK, "gpt2_embed_f32_fwd failed (", rc, "): ", cmhsa_status_string(rc));
    return x;
}

// The sampler's side (include/gpt2_step.h; reference sampler.py:14-19): logits (B, vocab), done (B) bool, tokens (B, cap)
// int64, pos (1) int64.  Writes tokens[:, pos + 1] and done in place; pos itself is left for the caller to increment.
void greedy_step(at::Tensor logits, at::Tensor done, at::Tensor tokens, at::Tensor pos, int64_t eos) {
    TORCH_CHECK(logits.is_cuda() && done.is_cuda() && tokens.is_cuda() && pos.is_cuda(), "all tensors must be CUDA tensors");
    TORCH_CHECK(logits.dtype() == at::kFloat && logits.dim() == 2 && logits.stride(1) == 1, "logits must be (B, vocab) float32 with contiguous rows");
    TORCH_CHECK(done.dtype() == at::kBool && done.dim() == 1 && done.is_contiguous() && done.size(0) == logits.size(0), "done must be (B) bool");
    TORCH_CHECK(tokens.dtype() == at::kLong && tokens.dim() == 2 && tokens.stride(1) == 1 && tokens.size(0) == logits.size(0), "tokens must be (B, cap) int64");
    TORCH_CHECK(pos.dtype() == at::kLong && pos.numel() == 1, "pos must hold one int64");
    TORCH_CHECK(done.device() == logits.device() && tokens.device() == logits.device() && pos.device() == logits.device(), "all tensors must be on the same device");
    if (logits.size(0) == 0) return;
    const c10::hip::HIPGuard guard(logits.device().index());
    const hipStream_t stream = c10::hip::getCurrentHIPStream(logits.device().index()).stream();
    const int rc = gpt2_greedy_step(logits.data_ptr<float>(), logits.stride(0), reinterpret_cast<unsigned char *>(done.data_ptr<bool>()),
                                    tokens.data_ptr<int64_t>(), tokens.stride(0), pos.data_ptr<int64_t>(), (int)logits.size(0), (int)logits.size(1),
                                    (int)tokens.size(1), eos, stream);
    TORCH_CHECK(rc == CMHSA_OK, "gpt2_greedy_step failed (", rc, "): ", cmhsa_status_string(rc));
}

TORCH_LIBRARY(causal_multihead_self_attention, m) {
    m.def("causal_multihead_self_attention_torch(Tensor Q, Tensor K, Tensor V, int num_heads) -> Tensor");
    m.def("causal_multihead_self_attention_packed_qkv(Tensor QKV, int num_heads) -> Tensor");
    m.def("reserve_workspace(Tensor like, int batch, int seq_len, int num_heads) -> int");
    m.def("causal_multihead_self_attention_decode(Tensor qkv_new, Tensor(a!) k_cache, Tensor(b!) v_cache, Tensor pos, int num_heads) -> Tensor");
    m.def("gelu_tanh(Tensor x) -> Tensor");
    m.def("add_layer_norm(Tensor x, Tensor? delta, Tensor weight, Tensor bias, float eps) -> (Tensor, Tensor)");
    m.def("linear_rows(Tensor x, Tensor weight, Tensor? bias, int act) -> Tensor");
    m.def("ln_linear_rows(Tensor x, Tensor? delta, Tensor ln_weight, Tensor ln_bias, float eps, Tensor weight, Tensor? bias, int act) -> (Tensor, Tensor)");
    m.def("embed_tokens(Tensor tokens, Tensor wte, Tensor wpe) -> Tensor");
    m.def("greedy_step(Tensor logits, Tensor(a!) done, Tensor(b!) tokens, Tensor pos, int eos) -> ()");
}

TORCH_LIBRARY_IMPL(causal_multihead_self_attention, CUDA, m) {
    m.impl("causal_multihead_self_attention_torch", &causal_multihead_self_attention_torch);
    m.impl("causal_multihead_self_attention_packed_qkv", &causal_multihead_self_attention_packed_qkv);
    m.impl("reserve_workspace", &reserve_workspace);
    m.impl("causal_multihead_self_attention_decode", &causal_multihead_self_attention_decode);
    m.impl("gelu_tanh", &gelu_tanh);
    m.impl("add_layer_norm", &add_layer_norm);
    m.impl("linear_rows", &linear_rows);
    m.impl("ln_linear_rows", &ln_linear_rows);
    m.impl("embed_tokens", &embed_tokens);
    m.impl("greedy_step", &greedy_step);
}

}  // namespace causal_multihead_self_attention

// The module body is empty, as in the reference (CMHSA.cu:39): importing it is what registers the op.
static struct PyModuleDef cmhsa_module = {
    PyModuleDef_HEAD_INIT, "causal_multihead_self_attention",
    "MI355X-native fp32 causal multi-head self-attention (registers "
    "torch.ops.causal_multihead_self_attention.causal_multihead_self_attention_torch)",
    -1, nullptr, nullptr, nullptr, nullptr, nullptr};

PyMODINIT_FUNC PyInit_causal_multihead_self_attention(void) { return PyModule_Create(&cmhsa_module); }
