// The reference-side binding of INTEGRATION.md section B, compilable: a PyBind/torch extension module with the
// reference's op signatures (quest/ops/csrc/bsk_ops.h:23-117) and module definition (bsk_ops.cu:4-20) whose
// bodies only build a quest_paged_kv_t view and call the C ABI of libquest_hip.so -- this is what a maintainer
// puts in place of the bodies of quest/ops/csrc/{page,estimate,topk,approx_attn,rms_norm}.cu.
// prefill_with_paged_kv_cache (bsk_ops.h:78-86, batch_prefill.cu:27-117) is bound the same way since round 5's MFMA flash
// kernel (csrc/prefill.hip), so the module carries the reference's WHOLE surface over the C ABI.
// Built by __graft_entry__.build() / scripts/check_cpp_binding.py, checked against quest_amd._kernels on the GPU
// by tests/test_gpu_cpp_binding.py.
#include <ATen/hip/HIPContext.h>
#include <torch/extension.h>

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>

#include "quest_hip.h"

static quest_paged_kv_t view(const torch::Tensor& data, const torch::Tensor& indices, const torch::Tensor& indptr,
                             unsigned last_len, unsigned last_idx, unsigned layout, unsigned budget = 0) {
    const bool hnd = layout == QUEST_LAYOUT_HND;
    quest_paged_kv_t p{};
    p.data = data.data_ptr();
    p.indices = static_cast<const int32_t*>(indices.data_ptr());
    p.indptr = static_cast<const int32_t*>(indptr.data_ptr());
    p.num_heads = data.size(hnd ? 2 : 3);
    p.page_size = data.size(hnd ? 3 : 2);
    p.head_dim = data.size(4);
    p.page_budget = budget;
    p.last_page_len = last_len;
    p.last_page_idx = (int32_t)last_idx;
    p.layout = layout;
    return p;
}

static quest_stream_t stream() { return (quest_stream_t)at::hip::getCurrentHIPStream().stream(); }

// bsk_ops.h:23-27
void apply_rope_in_place(torch::Tensor q, torch::Tensor k, unsigned int past_kv_len, float rope_scale,
                         float rope_theta) {
    int rc = quest_apply_rope_in_place(q.data_ptr(), k.data_ptr(), q.size(0), past_kv_len, q.size(1), k.size(1),
                                       q.size(2), rope_scale, rope_theta, stream());
    TORCH_CHECK(rc == 0, "apply_rope_in_place failed with error code ", quest_error_string(rc));
}

// bsk_ops.h:29-32; input [1][rows][cols]
void rms_norm_forward(torch::Tensor input, torch::Tensor weight, torch::Tensor output, float epsilon) {
    const auto cols = input.size(-1);
    int rc = quest_rms_norm_forward(input.data_ptr(), weight.data_ptr(), output.data_ptr(), input.numel() / cols, cols,
                                    epsilon, stream());
    TORCH_CHECK(rc == 0, "rms_norm_forward failed with error code ", quest_error_string(rc));
}

// bsk_ops.h:34-39
void topk_filtering(torch::Tensor estimated_value, torch::Tensor estimated_indices, torch::Tensor d_out,
                    torch::Tensor indices_out, torch::Tensor buf, unsigned int page_budget) {
    int rc = quest_topk_filtering(estimated_value.data_ptr(), static_cast<const int32_t*>(estimated_indices.data_ptr()),
                                  d_out.data_ptr(), static_cast<int32_t*>(indices_out.data_ptr()), buf.data_ptr(),
                                  estimated_value.size(0), estimated_value.size(1), page_budget, stream());
    TORCH_CHECK(rc == 0, "topk_filtering failed with error code ", quest_error_string(rc));
}

// bsk_ops.h:41-48
void estimate_attn_score(torch::Tensor q, torch::Tensor o, torch::Tensor metadata_data, torch::Tensor metadata_indices,
                         torch::Tensor metadata_indptr, unsigned int metadata_last_page_len,
                         unsigned int metadata_last_page_idx, unsigned int layout) {
    int rc = quest_estimate_attn_score(q.data_ptr(), o.data_ptr(), q.size(1), o.size(1),
                                       view(metadata_data, metadata_indices, metadata_indptr, metadata_last_page_len,
                                            metadata_last_page_idx, layout),
                                       stream());
    TORCH_CHECK(rc == 0, "Estimate_attn_score failed with error code ", quest_error_string(rc));
}

// bsk_ops.h:50-62
void append_kv_cache_prefill(torch::Tensor k, torch::Tensor v, torch::Tensor kv_data, torch::Tensor kv_indices,
                             torch::Tensor kv_indptr, unsigned int kv_last_page_len, unsigned int kv_last_page_idx,
                             torch::Tensor metadata_data, torch::Tensor metadata_indices, torch::Tensor metadata_indptr,
                             unsigned int metadata_last_page_len, unsigned int metadata_last_page_idx,
                             unsigned int layout) {
    int rc = quest_append_kv_cache_prefill(
        k.data_ptr(), v.data_ptr(), k.size(0), kv_indices.size(0),
        view(kv_data, kv_indices, kv_indptr, kv_last_page_len, kv_last_page_idx, layout),
        view(metadata_data, metadata_indices, metadata_indptr, metadata_last_page_len, metadata_last_page_idx, layout),
        stream());
    TORCH_CHECK(rc == 0, "Append_kv_cache_prefill failed with error code ", quest_error_string(rc));
}

// bsk_ops.h:64-76
void append_kv_cache_decode(torch::Tensor k, torch::Tensor v, torch::Tensor kv_data, torch::Tensor kv_indices,
                            torch::Tensor kv_indptr, unsigned int kv_last_page_len, unsigned int kv_last_page_idx,
                            torch::Tensor metadata_data, torch::Tensor metadata_indices, torch::Tensor metadata_indptr,
                            unsigned int metadata_last_page_len, unsigned int metadata_last_page_idx,
                            unsigned int layout) {
    int rc = quest_append_kv_cache_decode(
        k.data_ptr(), v.data_ptr(), view(kv_data, kv_indices, kv_indptr, kv_last_page_len, kv_last_page_idx, layout),
        view(metadata_data, metadata_indices, metadata_indptr, metadata_last_page_len, metadata_last_page_idx, layout),
        stream());
    TORCH_CHECK(rc == 0, "Append_kv_cache_decode failed with error code ", quest_error_string(rc));
}

// bsk_ops.h:78-86 (batch_prefill.cu:27-117): q [n][Hq][D] against the sequence's pages, query i sees keys 0 .. kv_len - n + i
// when causal.  One launch of the MFMA flash kernel straight over the page table (quest_prefill_with_paged_kv_cache).
// rope_* unused: the reference prefills with RotaryMode::kNone on this path (QuestAttention.py rotates q / k beforehand).
torch::Tensor prefill_with_paged_kv_cache(torch::Tensor q, torch::Tensor kv_data, torch::Tensor kv_indices,
                                          unsigned int kv_last_page_len, bool causal, unsigned int layout,
                                          bool /*allow_fp16_qk_reduction*/, float /*rope_scale*/, float /*rope_theta*/) {
    TORCH_CHECK(q.is_cuda() && kv_data.is_cuda() && kv_indices.is_cuda(), "prefill_with_paged_kv_cache: tensors must be on the GPU");
    TORCH_CHECK(q.is_contiguous() && kv_data.is_contiguous() && kv_indices.is_contiguous(), "prefill_with_paged_kv_cache: tensors must be contiguous");
    TORCH_CHECK(q.dim() == 3 && kv_data.dim() == 5 && kv_indices.dim() == 1, "prefill_with_paged_kv_cache: q must be 3-D, kv_data 5-D, kv_indices 1-D");
    TORCH_CHECK(kv_indices.scalar_type() == torch::kInt32, "prefill_with_paged_kv_cache: kv_indices must be int32");
    TORCH_CHECK(q.size(2) == kv_data.size(4), "prefill_with_paged_kv_cache: head_dim of q and kv_data differ");
    TORCH_CHECK(q.scalar_type() == torch::kHalf, "BatchPrefillWithPagedKVCache failed to dispatch with dtype ", q.scalar_type());
    torch::Tensor o = torch::empty_like(q);
    if (q.size(0) == 0) return o;  // nothing to attend from
    quest_paged_kv_t kv = view(kv_data, kv_indices, kv_indices, kv_last_page_len, 0, layout);
    kv.indptr = nullptr;  // one sequence: the page count travels as a host integer (batch_prefill.cu:41 builds {0, n})
    const int rc = quest_prefill_with_paged_kv_cache(q.data_ptr(), o.data_ptr(), q.size(0), q.size(1), kv, kv_indices.size(0),
                                                     causal ? 1 : 0, stream());
    if (rc == QUEST_EINVAL) throw std::invalid_argument(std::string("BatchPrefillWithPagedKVCache: ") + quest_error_string(rc));
    TORCH_CHECK(rc == 0, "BatchPrefillWithPagedKVCache failed with error code ", quest_error_string(rc));
    return o;
}

// bsk_ops.h:88-117: the handler class over quest_decode_handler_t
class BatchDecodeWithPagedKVCachePyTorchWrapper {
public:
    static BatchDecodeWithPagedKVCachePyTorchWrapper Create(unsigned int layout) {
        return BatchDecodeWithPagedKVCachePyTorchWrapper(layout);
    }
    BatchDecodeWithPagedKVCachePyTorchWrapper(BatchDecodeWithPagedKVCachePyTorchWrapper&& o) noexcept
        : h_(o.h_), layout_(o.layout_) {
        o.h_ = nullptr;
    }
    BatchDecodeWithPagedKVCachePyTorchWrapper(const BatchDecodeWithPagedKVCachePyTorchWrapper&) = delete;
    ~BatchDecodeWithPagedKVCachePyTorchWrapper() { quest_decode_handler_destroy(h_); }

    void BeginForward(torch::Tensor indptr, unsigned int num_qo_heads, unsigned int num_kv_heads, unsigned int head_dim,
                      unsigned int page_size, torch::Tensor empty_data) {
        TORCH_CHECK(empty_data.scalar_type() == torch::kHalf, "BatchDecodeWithPagedKVCache failed to dispatch with dtype ",
                    empty_data.scalar_type());
        // the planner needs the page count on the host, as the reference's does (decode_attn.cuh:866-873)
        const torch::Tensor host = indptr.cpu();
        const int32_t* p = host.data_ptr<int32_t>();
        int rc = quest_decode_begin_forward(h_, (uint32_t)(p[host.numel() - 1] - p[0]), num_qo_heads, num_kv_heads, head_dim,
                                            page_size, stream());
        if (rc == QUEST_EINVAL) throw std::invalid_argument(quest_error_string(rc));  // decode_attn.cuh:1045-1050
        TORCH_CHECK(rc == 0, "BatchDecodeWithPagedKVCache failed with error code ", quest_error_string(rc));
    }

    void EndForward() { quest_decode_end_forward(h_); }

    void Forward(torch::Tensor q, torch::Tensor o, torch::Tensor paged_kv_data, torch::Tensor paged_kv_indices,
                 torch::Tensor paged_kv_indptr, unsigned int paged_kv_last_page_len, unsigned int paged_kv_last_page_idx,
                 float /*rope_scale*/, float /*rope_theta*/) {  // RotaryMode::kNone on this path (approx_attn.cu:131)
        int rc = quest_decode_forward(h_, q.data_ptr(), o.data_ptr(),
                                      view(paged_kv_data, paged_kv_indices, paged_kv_indptr, paged_kv_last_page_len,
                                           paged_kv_last_page_idx, layout_, paged_kv_indices.size(1)),
                                      q.size(1), nullptr, stream());
        TORCH_CHECK(rc == 0, "BatchDecodeWithPagedKVCache failed with error code ", quest_error_string(rc));
    }

private:
    explicit BatchDecodeWithPagedKVCachePyTorchWrapper(unsigned int layout) : layout_(layout) {
        int rc = quest_decode_handler_create(&h_, layout);
        TORCH_CHECK(rc == 0, "BatchDecodeWithPagedKVCache: ", quest_error_string(rc));
    }
    quest_decode_handler_t* h_ = nullptr;
    unsigned int layout_;
};

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {  // bsk_ops.cu:4-20 (the reference names the module _kernels)
    m.def("apply_rope_in_place", &apply_rope_in_place, "rotate q and k rows in place (HIP)");
    m.def("rms_norm_forward", &rms_norm_forward, "row-wise RMS normalisation (HIP)");
    m.def("topk_filtering", &topk_filtering, "deterministic per-head top-k page selection (HIP)");
    m.def("estimate_attn_score", &estimate_attn_score, "page criticality scores from (max, min) metadata (HIP)");
    m.def("append_kv_cache_prefill", &append_kv_cache_prefill, "append many tokens + fold page metadata (HIP)");
    m.def("append_kv_cache_decode", &append_kv_cache_decode, "append one token + fold page metadata (HIP)");
    m.def("prefill_with_paged_kv_cache", &prefill_with_paged_kv_cache, "prefill attention over the paged cache");
    py::class_<BatchDecodeWithPagedKVCachePyTorchWrapper>(m, "BatchDecodeWithPagedKVCachePyTorchWrapper")
        .def(py::init(&BatchDecodeWithPagedKVCachePyTorchWrapper::Create))
        .def("begin_forward", &BatchDecodeWithPagedKVCachePyTorchWrapper::BeginForward)
        .def("end_forward", &BatchDecodeWithPagedKVCachePyTorchWrapper::EndForward)
        .def("forward", &BatchDecodeWithPagedKVCachePyTorchWrapper::Forward);
}
