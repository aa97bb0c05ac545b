// The reference-side binding of INTEGRATION.md section B, compilable: a PyBind/torch extension module
// with the reference's op signatures (quest/ops/csrc/bsk_ops.h:38-52, :70-82) whose bodies only build a
// quest_paged_kv_t view and call the C ABI of libquest_hip.so.  Built and checked against quest_amd._kernels
// by scripts/check_cpp_binding.py.  (Three ops are enough to show the mapping; the rest follow the table in
// INTEGRATION.md.)
#include <ATen/hip/HIPContext.h>
#include <torch/extension.h>

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

void estimate_attn_score(torch::Tensor q, torch::Tensor o, torch::Tensor metadata_data, torch::Tensor metadata_indices,
                         torch::Tensor metadata_indptr, unsigned int metadata_last_page_len,
                         unsigned int metadata_last_page_idx, unsigned int layout) {
    int rc = quest_estimate_attn_score(q.data_ptr(), o.data_ptr(), q.size(1), o.size(1),
                                       view(metadata_data, metadata_indices, metadata_indptr, metadata_last_page_len,
                                            metadata_last_page_idx, layout),
                                       stream());
    TORCH_CHECK(rc == 0, "Estimate_attn_score failed with error code ", quest_error_string(rc));
}

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

void topk_filtering(torch::Tensor estimated_value, torch::Tensor estimated_indices, torch::Tensor d_out,
                    torch::Tensor indices_out, torch::Tensor buf, unsigned int page_budget) {
    int rc = quest_topk_filtering(estimated_value.data_ptr(), static_cast<const int32_t*>(estimated_indices.data_ptr()),
                                  d_out.data_ptr(), static_cast<int32_t*>(indices_out.data_ptr()), buf.data_ptr(),
                                  estimated_value.size(0), estimated_value.size(1), page_budget, stream());
    TORCH_CHECK(rc == 0, "topk_filtering failed with error code ", quest_error_string(rc));
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("estimate_attn_score", &estimate_attn_score);
    m.def("append_kv_cache_decode", &append_kv_cache_decode);
    m.def("topk_filtering", &topk_filtering);
}
