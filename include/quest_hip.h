/*
 * quest_hip.h -- C ABI of libquest_hip.so, the MI355X (gfx950) implementation of Quest's
 * query-aware sparse decode path.
 *
 * This is the drop-in boundary: every entry point takes plain device pointers, sizes and a
 * HIP stream -- no torch types -- and corresponds one-to-one to an operator the reference
 * exposes through its PyBind module `quest._kernels` (quest/ops/csrc/bsk_ops.cu:4-20,
 * declarations bsk_ops.h:23-117).  INTEGRATION.md shows the binding a maintainer adds on
 * the reference side.
 *
 * Conventions
 *   - All tensors are fp16 (the reference dispatches only Half: pytorch_extension_utils.h:25-36),
 *     contiguous, resident in device memory.  Index tensors are int32.
 *   - Return value: 0 on success; a positive value is a hipError_t from a launch; a negative
 *     value is one of the QUEST_E* argument errors below.  quest_error_string() names both.
 *     Nothing throws; nothing allocates except quest_decode_begin_forward (workspace, like
 *     BatchDecodeHandler::BeginForward, decode_handler.cuh:73-121).
 *   - Every launch is asynchronous on `stream` (the reference launches on the NULL stream,
 *     page.cu:89 / approx_attn.cu:141; here the caller passes torch's current stream).
 *   - layout: 0 = NHD pool [pages][2][page_size][heads][dim], 1 = HND [pages][2][heads][page_size][dim]
 *     (quest/utils/utils.py:1-5, kernels/include/decode/decode_page.cuh:196-239).
 */
#ifndef QUEST_HIP_H_
#define QUEST_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* quest_stream_t; /* hipStream_t */

#define QUEST_LAYOUT_NHD 0u
#define QUEST_LAYOUT_HND 1u
/*
 * EXTENSION (round 6): NHD with the heads of an entry ROTATED by the entry.  A pool layer has the NHD shape
 * [pages][2][page_size][num_heads][head_dim], but inside entry e's row of heads the K (metadata pool: max) vector of
 * head h sits in head slot   h ^ (e & rot),   rot   = min(num_heads & -num_heads, 4) - 1        (0, 1 or 3)
 * and its V (min) vector in   that slot ^ flip,  flip  = (min(num_heads & -num_heads, 32) - 1) & ~3  (0, 4, 12 or 28).
 * Why: on the plain NHD pool all 256-byte pieces of a head share address bits 8-12, and MI355X serves pieces whose
 * address bits 8-9 are 01 ~20 % slower than the others under mixed traffic (scripts/probe/addr_class_probe.hip), so a
 * launch with one workgroup per head ends with its "slow" heads while most CUs idle.  Rotated, every head's pieces
 * cycle through all values of those bits: every workgroup sees the mix's mean.  Same bytes, same arithmetic, same fold
 * order as NHD -> the same scores, selections and outputs bit for bit; only WHERE a vector lives differs
 * (quest_pool_slot, declared with the pool view below, is the whole definition).  Not a layout the reference has (quest/utils/utils.py:1-5 knows NHD and
 * HND; decode_page.cuh:196-239 are their offsets): a pool in this layout must be written and read through this library.
 */
#define QUEST_LAYOUT_NHD_ROT 2u

#define QUEST_EINVAL (-1)      /* malformed argument (null pointer, zero size, bad layout) */
#define QUEST_EUNSUPPORTED (-2) /* head_dim / page_size / group size outside the built set */
#define QUEST_ESTATE (-3)      /* forward without begin_forward (decode_handler.cuh:226-231) */
#define QUEST_ETOOLARGE (-4)   /* top-k row longer than QUEST_TOPK_MAX_ROW */

#define QUEST_TOPK_MAX_ROW 16384u

/*
 * View of one layer of a paged pool plus one sequence's page table.  Replaces
 * paged_kv_t<kIndices, layout, half, int32_t> (decode_page.cuh:79-110), which the reference
 * passes to its kernels by value; so is this.
 */
typedef struct quest_paged_kv {
    void* data;             /* fp16 pool base of this layer (kv_cache.py:111-113 buf_layer) */
    const int32_t* indices; /* page table; for quest_decode_forward: [num_qo_heads][page_budget] */
    const int32_t* indptr;  /* device int32[2] = {0, n_pages} */
    uint32_t num_heads;     /* heads stored in the pool (kv heads) */
    uint32_t page_size;
    uint32_t head_dim;
    uint32_t page_budget;   /* row stride of `indices` in quest_decode_forward, else 0 */
    uint32_t last_page_len; /* 1..page_size */
    int32_t last_page_idx;  /* physical id of the sequence's last page */
    uint32_t layout;
    uint32_t reserved;
} quest_paged_kv_t;

/* Where a vector lives inside an entry's row of heads (host function, no GPU): the head SLOT of head `head`'s K / max vector
 * (v_slot == 0) or V / min vector (v_slot != 0) of entry `entry` (its index inside the page) in a pool of `num_heads` heads.
 * NHD and HND: the head itself; NHD_ROT: head ^ (entry & rot) [^ flip for V], the definition in the comment of
 * QUEST_LAYOUT_NHD_ROT above.  The address of the vector is then, in halves from the layer's base,
 * page * 2*S*H*D + v * S*H*D + entry * H*D + slot * D (NHD family).  0xffffffff for a malformed layout / head. */
uint32_t quest_pool_slot(uint32_t layout, uint32_t num_heads, uint32_t head, uint32_t entry, int v_slot);

const char* quest_error_string(int code);

/* Build identification: "gfx950" and the kernel generation, for logs. */
const char* quest_build_info(void);

/*
 * append_kv_cache_decode (bsk_ops.h:64-76, page.cu:6-99 -> AppendPagedKVCacheDecode,
 * decode_page.cuh:577-597, kernel :398-449).
 * k, v: [1][num_heads][head_dim].  Writes the token into the last page and folds k into the
 * page's (max -> K slot, min -> V slot) metadata entry, re-initialising it to -/+65504 when the
 * token opens a new page.
 */
int quest_append_kv_cache_decode(const void* k, const void* v, quest_paged_kv_t kv,
                                 quest_paged_kv_t metadata, quest_stream_t stream);

/*
 * append_kv_cache_prefill (bsk_ops.h:50-62, page.cu:101-210 -> AppendPagedKVCachePrefill,
 * decode_page.cuh:613-642, kernel :471-562).  k, v: [append_len][num_heads][head_dim];
 * n_pages_host = number of pages of the sequence (the host knows it: len(kv_indices)).
 */
int quest_append_kv_cache_prefill(const void* k, const void* v, uint32_t append_len,
                                  uint32_t n_pages_host, quest_paged_kv_t kv,
                                  quest_paged_kv_t metadata, quest_stream_t stream);

/*
 * estimate_attn_score (bsk_ops.h:41-48, estimate.cu:6-84 -> MaxPossibleSampleWithPagedKVCache,
 * decode_attn.cuh:1092-1149, kernel :245-401).
 * q: [1][num_qo_heads][head_dim]; o: [num_qo_heads][n_out] fp16 with n_out = (number of KV
 * pages) - 1; o[h][p] = fp16( sum_d max(q*Kmax, q*Kmin) ).
 */
int quest_estimate_attn_score(const void* q, void* o, uint32_t num_qo_heads, uint32_t n_out,
                              quest_paged_kv_t metadata, quest_stream_t stream);

/*
 * Fused form of the two calls QuestAttention.forward makes back to back for a decode token
 * (QuestAttention.py:106 append_kv, :136 decode_estimate): ONE launch writes the new token (k, v) into
 * the cache + folds it into its page's metadata, and scores all pages but the current one.  The two
 * halves touch disjoint bytes (the estimate excludes the current page's entry), so the result is
 * identical to calling quest_append_kv_cache_decode then quest_estimate_attn_score.
 */
int quest_append_estimate(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                          uint32_t num_qo_heads, uint32_t n_out, quest_paged_kv_t metadata,
                          quest_stream_t stream);
/* The same with a row stride for `o` (elements between the rows of consecutive query heads; 0 = n_out). */
int quest_append_estimate_strided(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                                  uint32_t num_qo_heads, uint32_t n_out, uint32_t o_stride, quest_paged_kv_t metadata,
                                  quest_stream_t stream);

/*
 * topk_filtering (bsk_ops.h:34-39, topk.cu:7-46 -> decode_select_k, decode_select_k.cuh:25-62,
 * which calls RAFT's radix_topk_one_block_kernel; re-implemented here).
 * estimated_value/indices: [num_heads][num_pages]; d_out/indices_out: [num_heads][page_budget].
 * Deterministic: ties at the k-th value go to the lowest column; output in ascending column order.
 * `buf` is the reference's scratch argument; unused (may be NULL).
 */
int quest_topk_filtering(const void* estimated_value, const int32_t* estimated_indices, void* d_out,
                         int32_t* indices_out, void* buf, uint32_t num_heads, uint32_t num_pages,
                         uint32_t page_budget, quest_stream_t stream);

/* The same with a row stride for `estimated_value` (elements between consecutive heads' rows; 0 = num_pages): serves the
 * padded score rows quest_append_estimate_strided writes without a compacting copy. */
int quest_topk_filtering_strided(const void* estimated_value, uint32_t value_stride, const int32_t* estimated_indices,
                                 void* d_out, int32_t* indices_out, void* buf, uint32_t num_heads, uint32_t num_pages,
                                 uint32_t page_budget, quest_stream_t stream);

/*
 * BatchDecodeWithPagedKVCachePyTorchWrapper (bsk_ops.h:88-117, approx_attn.cu:27-150 ->
 * BatchDecodeHandler, decode_handler.cuh:39-244).
 */
typedef struct quest_decode_handler quest_decode_handler_t;

int quest_decode_handler_create(quest_decode_handler_t** out, uint32_t layout);
void quest_decode_handler_destroy(quest_decode_handler_t* h);

/* begin_forward: n_selected_pages = indptr[1] - indptr[0] = page budget - 1 (the caller has it on
 * the host; the reference copies it back from the device, decode_attn.cuh:866-873).  Plans the
 * split of each head's page list over workgroups and sizes the partial-state workspace. */
int quest_decode_begin_forward(quest_decode_handler_t* h, uint32_t n_selected_pages,
                               uint32_t num_qo_heads, uint32_t num_kv_heads, uint32_t head_dim,
                               uint32_t page_size, quest_stream_t stream);
int quest_decode_end_forward(quest_decode_handler_t* h);

/* forward: q, o: [1][num_qo_heads][head_dim]; paged_kv.indices = [num_qo_heads][page_budget]
 * selected physical pages (row stride paged_kv.page_budget), to which the sequence's last page
 * (last_page_idx, last_page_len tokens) is always added (decode_page.cuh:325-351).
 * lse: optional float[num_qo_heads] (natural log), may be NULL. */
int quest_decode_forward(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t paged_kv,
                         uint32_t num_qo_heads, float* lse, quest_stream_t stream);

/*
 * forward for the case where every query head attends the SAME page list (what the reference passes as
 * kv_indices_without_last.repeat(num_heads, 1) when the budget covers the cache, controller.py:106,
 * QuestAttention.py:125-132): paged_kv.indices = ONE row [n_selected_pages].  With GQA the K/V tiles of
 * a kv head are then fetched once for its whole query group.  page_size 16, head_dim 64/128.
 */
int quest_decode_forward_shared(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t paged_kv,
                                uint32_t num_qo_heads, float* lse, quest_stream_t stream);

/*
 * Fused form of decode_topk + decode_sparse_attn (QuestAttention.py:144-157): the top-k selection of
 * quest_topk_filtering runs at the head of the attention kernel (every workgroup of a head recomputes
 * it from the head's score row -- a 4 KiB L2-resident read -- so no second launch and no cross-workgroup
 * hand-off), then the selected pages are gathered exactly as in quest_decode_forward.
 *   scores      [num_qo_heads][n_scores] fp16, the estimate output (n_scores = pages - 1)
 *   page_table  paged_kv.indices = the sequence's page table [n_scores + 1] (kv_indices_with_last)
 *   topk_val_out / topk_idx_out  optional [num_qo_heads][n_selected_pages]: the selection, bit-identical
 *                to quest_topk_filtering's outputs (same routine), for callers that inspect it.
 */
int quest_decode_forward_fused_topk(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t paged_kv,
                                    uint32_t num_qo_heads, const void* scores, uint32_t n_scores,
                                    void* topk_val_out, int32_t* topk_idx_out, float* lse, quest_stream_t stream);
/* The same with a row stride for `scores` (elements; 0 = n_scores).  The reference's score tensor is a contiguous
 * [num_qo_heads][pages - 1] (quest/utils/__init__.py:171-205), whose rows are only 2-byte aligned; rows padded to a
 * multiple of 8 columns (16 bytes: quest_append_estimate_strided writes them) let the front end fetch a thread's
 * scores with one vector load.  Both layouts are served in ONE launch (since round 5 also the reference's layout beyond 4096
 * pages: every workgroup reads the 8-byte aligned stream below its row and skips the 0-3 leading columns of the previous
 * row; up to QUEST_TOPK_MAX_ROW - 3 pages -- QUEST_EUNSUPPORTED beyond: callers then issue quest_topk_filtering +
 * quest_decode_forward). */
int quest_decode_forward_fused_topk_strided(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t paged_kv,
                                            uint32_t num_qo_heads, const void* scores, uint32_t n_scores,
                                            uint32_t score_stride, void* topk_val_out, int32_t* topk_idx_out, float* lse,
                                            quest_stream_t stream);

/*
 * Device-resident step state (host planning rewrite).  The reference rebuilds page-table
 * tensors and re-plans on the host for every token (quest/utils/controller.py:80-129); with the sequence
 * state in device memory a decode step captured ONCE in a hipGraph can be replayed token after token while
 * the sequence grows: quest_step_state_advance is prepare_metadata(1) on the device, and the *_dyn entry
 * points read lengths / last-page ids from the state instead of from their by-value arguments.
 * The page tables must be materialised up to the pool capacity (page i of the sequence = kv_table[i]).
 */
typedef struct quest_step_state {
    int32_t seq_len;            /* tokens in the cache, including the one being decoded */
    int32_t n_pages;            /* KV pages in use */
    int32_t kv_last_page_len;   /* 1..page_size */
    int32_t kv_last_page_idx;   /* physical id */
    int32_t n_meta_pages;
    int32_t meta_last_page_len;
    int32_t meta_last_page_idx;
    int32_t reserved;           /* set to 1 by quest_step_state_advance when the pool is exhausted */
} quest_step_state_t;

/* Reserve room for one more token: the device-side prepare_metadata(1) (controller.py:72-76).  The tables
 * hold max_kv_pages / max_meta_pages entries (the pools' capacities); when the next token would not fit the
 * state is left unchanged and `reserved` is set, so a replayed graph can never index past a table. */
int quest_step_state_advance(quest_step_state_t* state, const int32_t* kv_table, const int32_t* meta_table,
                             uint32_t page_size, uint32_t max_kv_pages, uint32_t max_meta_pages,
                             quest_stream_t stream);

/* quest_append_estimate with lengths / last-page ids / n_out (= state->n_pages - 1) taken from `state`.
 * o is [num_qo_heads][o_stride] with o_stride >= max_n_out, the largest n_out the graph will ever see
 * (the grid is sized for it; surplus workgroups exit). */
int quest_append_estimate_dyn(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                              uint32_t num_qo_heads, uint32_t o_stride, uint32_t max_n_out,
                              quest_paged_kv_t metadata, const quest_step_state_t* state, quest_stream_t stream);

/* quest_decode_forward_fused_topk with the row length (state->n_pages - 1) and the current page taken
 * from `state`; scores is [num_qo_heads][score_stride].  The plan (selected-page count) is the one of
 * begin_forward and must not exceed state->n_pages - 1 for the life of the graph. */
int quest_decode_forward_fused_topk_dyn(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t paged_kv,
                                        uint32_t num_qo_heads, const void* scores, uint32_t score_stride,
                                        uint32_t max_n_scores, const quest_step_state_t* state, float* lse,
                                        quest_stream_t stream);

/*
 * The same pair for LONG score rows (round 5; cfg 4: 8191 pages), with TILE MAXIMA handed across the kernel boundary: the
 * estimate workgroup that holds the scores of 8 consecutive pages also stores the largest of them (as an order-preserving
 * 16-bit key) at o[h][tile_max_offset + page / 8], and the attention launch selects in two short passes -- the top-k TILES by
 * maximum (a sufficient candidate set under the declared tie rule: csrc/decode_device.cuh sparse_decode_tiles_body), then the
 * exact top-k over those tiles' 8 k scores -- instead of passes over the whole row in every workgroup of a head.  Same
 * selection, same outputs as quest_append_estimate_dyn + quest_decode_forward_fused_topk_dyn.
 * tile_max_offset must be max_n rounded up to 8 columns; rows 16-byte aligned with
 * stride >= tile_max_offset + ceil(max_n / 8) rounded up to 4.  Plans with at most 256 selected pages, page_size 16, rows
 * up to 16384 pages; QUEST_EUNSUPPORTED otherwise.  The estimate side needs its workgroup tile to be a multiple of 8 pages
 * wide (head_dim 256: the tile covers 4 kv heads instead of 8); a shape for which no such tile exists returns
 * QUEST_EUNSUPPORTED before anything is launched, and the caller takes quest_append_estimate_dyn instead.
 */
int quest_append_estimate_tiles_dyn(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                                    uint32_t num_qo_heads, uint32_t o_stride, uint32_t max_n_out, uint32_t tile_max_offset,
                                    quest_paged_kv_t metadata, const quest_step_state_t* state, quest_stream_t stream);
int quest_decode_forward_fused_topk_tiles_dyn(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t paged_kv,
                                              uint32_t num_qo_heads, const void* scores, uint32_t score_stride,
                                              uint32_t max_n_scores, uint32_t tile_max_offset,
                                              const quest_step_state_t* state, float* lse, quest_stream_t stream);

/* quest_append_kv_cache_decode with lengths / last-page ids from `state` (dense layers of a replayed step). */
int quest_append_kv_cache_decode_dyn(const void* k, const void* v, quest_paged_kv_t kv, quest_paged_kv_t metadata,
                                     const quest_step_state_t* state, quest_stream_t stream);

/* quest_decode_forward_shared over ALL pages of the sequence (state->n_pages - 1 listed pages + the current
 * one): paged_kv.indices = the full page table.  Plan with begin_forward(n_selected_pages = pool capacity - 1);
 * workgroups past the live length contribute empty partial states. */
int quest_decode_forward_shared_dyn(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t paged_kv,
                                    uint32_t num_qo_heads, const quest_step_state_t* state, float* lse,
                                    quest_stream_t stream);

/* quest_append_kv_cache_decode_dyn + quest_decode_forward_shared_dyn in ONE launch (round 4: a full-KV layer of a captured
 * step = attention + merge instead of append + attention + merge; the reference's pair of calls is QuestAttention.py:106
 * and :125-132 / page.cu:6-99).  k, v: the new token [1][kv heads][head_dim], not yet in the pool: the one workgroup per
 * kv head that attends the current page takes the row from k / v, writes it to the pool and folds k into the page's
 * (max, min) metadata entry -- same pool bytes as the separate append.  metadata: the layer's metadata pool (same
 * geometry as the KV pool; lengths from `state`).  QUEST_EUNSUPPORTED for shapes outside the group-shared kernel
 * (page_size 16, head_dim 64 / 128): issue the two launches then. */
int quest_decode_append_forward_shared_dyn(quest_decode_handler_t* h, const void* k, const void* v,
                                           quest_paged_kv_t metadata, const void* q, void* o, quest_paged_kv_t paged_kv,
                                           uint32_t num_qo_heads, const quest_step_state_t* state, float* lse,
                                           quest_stream_t stream);

/* quest_apply_rope_in_place for one decode token with past_kv_len = state->seq_len - 1. */
int quest_apply_rope_in_place_dyn(void* q, void* k, uint32_t num_qo_heads, uint32_t num_kv_heads, uint32_t head_dim,
                                  float rope_scale, float rope_theta, const quest_step_state_t* state,
                                  quest_stream_t stream);

/*
 * Batched state-driven step (SURVEY.md 8f-3 second half: the reference fixes `constexpr batch_size = 1`,
 * approx_attn.cu:113 / estimate.cu:14; BASELINE config 5 decodes 8 sequences per GPU).  One launch serves
 * n_seqs independent sequences (grid.z / grid.y = sequence), no data is shared between them:
 *   - ONE KV pool and ONE metadata pool hold every sequence's pages (`data` of the views below);
 *   - page tables are rows of one int32 matrix: sequence i's table = indices + i * *_table_stride,
 *     materialised up to its capacity like the single-sequence tables above;
 *   - `state` is an array of n_seqs step states; q / o / k / v are [n_seqs][heads][dim], scores
 *     [n_seqs][num_qo_heads][score_stride], lse [n_seqs][num_qo_heads].
 * Per sequence: pools, estimates and page selections are bit-identical to the single-sequence *_dyn entry
 * points; attention outputs too when the work split (pages per workgroup) is the same, otherwise they differ
 * by the fp32 rounding of a different partial-state merge order (tests: <= 2e-3).
 */
typedef struct quest_batch {
    uint32_t n_seqs;
    uint32_t kv_table_stride;   /* int32 entries between consecutive sequences' KV page tables */
    uint32_t meta_table_stride; /* ... metadata page tables */
    uint32_t reserved;
    /* Optional per-sequence page budgets, device int32[n_seqs]: pages sequence i attends INCLUDING its current page
     * (the reference's per-request page budget: InferenceController.set_page_budget, quest/utils/controller.py:66-68,
     * one controller per request).  NULL: every sequence takes the budget the handler was planned with.  With
     * budgets, plan the handler (begin_forward) with the LARGEST one; sequence i then selects
     * min(budget[i] - 1, its pages - 1) pages. */
    const int32_t* page_budgets;
} quest_batch_t;

/* quest_step_state_advance for every sequence of the batch (tables: [n_seqs][*_table_stride], each row
 * valid up to max_kv_pages / max_meta_pages entries). */
int quest_step_state_advance_batched(quest_step_state_t* state, const int32_t* kv_tables, const int32_t* meta_tables,
                                     uint32_t page_size, uint32_t max_kv_pages, uint32_t max_meta_pages,
                                     quest_batch_t batch, quest_stream_t stream);

/* The same reservation RIDING IN A STEP'S LAST LAUNCH (round 6): quest_step_state_advance at the head of a captured step is a
 * 1-thread launch of dependent loads -- 4.7 us per token.  Nothing after a layer's merge launch reads the step state, so the
 * reservation for the NEXT token can be made there: arm it on the handler right before the last layer's forward call of a
 * step; the merge launch of that call carries it (one thread per sequence of its first workgroup) and the handler disarms.
 * A forward whose plan has no merge launch issues it as its own launch behind the attention kernel (correct, nothing saved).
 * The caller reserves the FIRST token with quest_step_state_advance before the first step; from then on the device state
 * (and its host mirror, InferenceController.prepare_metadata(1) after every step) runs one reserved token ahead of the
 * tokens appended.  state == NULL disarms.  At most 64 sequences. */
int quest_decode_arm_step_advance(quest_decode_handler_t* h, quest_step_state_t* state, const int32_t* kv_tables,
                                  const int32_t* meta_tables, uint32_t page_size, uint32_t max_kv_pages,
                                  uint32_t max_meta_pages, quest_batch_t batch);
int quest_append_estimate_batched(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                                  uint32_t num_qo_heads, uint32_t o_stride, uint32_t max_n_out,
                                  quest_paged_kv_t metadata, const quest_step_state_t* state, quest_batch_t batch,
                                  quest_stream_t stream);
/* The handler must have been planned for the batch: quest_decode_set_batch(h, n_seqs) before begin_forward. */
int quest_decode_forward_fused_topk_batched(quest_decode_handler_t* h, const void* q, void* o,
                                            quest_paged_kv_t paged_kv, uint32_t num_qo_heads, const void* scores,
                                            uint32_t score_stride, uint32_t max_n_scores,
                                            const quest_step_state_t* state, quest_batch_t batch, float* lse,
                                            quest_stream_t stream);
int quest_append_kv_cache_decode_batched(const void* k, const void* v, quest_paged_kv_t kv, quest_paged_kv_t metadata,
                                         const quest_step_state_t* state, quest_batch_t batch, quest_stream_t stream);
int quest_decode_forward_shared_batched(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t paged_kv,
                                        uint32_t num_qo_heads, const quest_step_state_t* state, quest_batch_t batch,
                                        float* lse, quest_stream_t stream);
/* quest_decode_append_forward_shared_dyn for every sequence of the batch (k, v: [n_seqs][kv heads][head_dim]). */
int quest_decode_append_forward_shared_batched(quest_decode_handler_t* h, const void* k, const void* v,
                                               quest_paged_kv_t metadata, const void* q, void* o,
                                               quest_paged_kv_t paged_kv, uint32_t num_qo_heads,
                                               const quest_step_state_t* state, quest_batch_t batch, float* lse,
                                               quest_stream_t stream);
/*
 * One launch per layer of a batched decode step (round 5): quest_append_estimate_batched +
 * quest_decode_forward_fused_topk_batched as ONE grid of (sequence, query head) workgroups, each of which owns its head
 * end to end -- appends its kv head's new token (QuestAttention.py:106, page.cu:6-99), scores its head's pages into LDS
 * (:136, estimate.cu:6-84), selects from LDS (:144, topk.cu:7-46) and gathers the selected pages + the current one
 * (:147-157, approx_attn.cu:68-151).  No score scratch, no cross-workgroup hand-off, the same pool bytes, selections and
 * outputs as the two launches.  k, v: [n_seqs][kv heads][head_dim] (not yet in the pool); metadata / paged_kv: the
 * layer's pools (same geometry) with indices = the stacked page tables; max_n_scores = page capacity - 1.
 * scores_out: optional inspection copy of the page scores, [n_seqs][num_qo_heads][score_stride] fp16 (NULL: they stay
 * in LDS); the selection can be inspected with quest_decode_set_selection_out.
 * Serves plans with ONE workgroup per head (quest_decode_set_batch + begin_forward on a batch that fills the chip),
 * page_size 16, head_dim 64 / 128, at most 255 selected pages (token budget 4096), rows up to 4096 pages; QUEST_EUNSUPPORTED otherwise
 * (issue the two launches then).
 */
int quest_decode_layer_fused_batched(quest_decode_handler_t* h, const void* k, const void* v, quest_paged_kv_t metadata,
                                     const void* q, void* o, quest_paged_kv_t paged_kv, uint32_t num_qo_heads,
                                     uint32_t max_n_scores, const quest_step_state_t* state, quest_batch_t batch,
                                     void* scores_out, uint32_t score_stride, float* lse, quest_stream_t stream);
/*
 * The four operators of a decode step one by one for a whole batch (the state-driven counterparts of
 * append_kv_cache_decode / estimate_attn_score / topk_filtering / BatchDecodeWithPagedKVCache.forward, bsk_ops.h:34-117,
 * as quest/utils/__init__.py:141-276 calls them per request): quest_append_kv_cache_decode_batched above, and
 *   estimate : o[n_seqs][num_qo_heads][o_stride], row i scored over state[i].n_pages - 1 pages;
 *   top-k    : per (sequence, head) the k_i = min(budget_i - 1, n_pages_i - 1) largest scores of the row and the
 *              physical pages they belong to (kv_tables[i][column]), written to the first k_i entries of rows of
 *              out_stride entries; page_budget = the launch-wide budget (pages incl. the current one) used where
 *              batch.page_budgets is NULL;
 *   attention: per (sequence, head) over indices[i][h][0 .. k_i) + the current page; plan the handler with
 *              begin_forward(n_selected_pages = max_i budget_i - 1) after quest_decode_set_batch.
 * Same bits as the single-sequence operators on each sequence.
 */
int quest_estimate_attn_score_batched(const void* q, void* o, uint32_t num_qo_heads, uint32_t o_stride, uint32_t max_n_out,
                                      quest_paged_kv_t metadata, const quest_step_state_t* state, quest_batch_t batch,
                                      quest_stream_t stream);
int quest_topk_filtering_batched(const void* scores, uint32_t score_stride, uint32_t max_num_pages, const int32_t* kv_tables,
                                 void* d_out, int32_t* indices_out, uint32_t out_stride, uint32_t num_heads,
                                 uint32_t page_budget, const quest_step_state_t* state, quest_batch_t batch,
                                 quest_stream_t stream);
int quest_decode_forward_batched(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t paged_kv,
                                 uint32_t num_qo_heads, const int32_t* indices, uint32_t idx_stride,
                                 const quest_step_state_t* state, quest_batch_t batch, float* lse, quest_stream_t stream);
/* q: [n_seqs][num_qo_heads][dim], k: [n_seqs][num_kv_heads][dim]; row i sits at position state[i].seq_len - 1. */
int quest_apply_rope_in_place_batched(void* q, void* k, uint32_t num_qo_heads, uint32_t num_kv_heads,
                                      uint32_t head_dim, float rope_scale, float rope_theta,
                                      const quest_step_state_t* state, quest_batch_t batch, quest_stream_t stream);
/* Sequences per launch the next begin_forward plans (workspace, work split) for; default 1. */
int quest_decode_set_batch(quest_decode_handler_t* h, uint32_t n_seqs);

/* Introspection of the current plan (for benches/tests): pages per workgroup, workgroups per head. */
int quest_decode_plan_info(const quest_decode_handler_t* h, uint32_t* pages_per_chunk,
                           uint32_t* chunks_per_head);
/* Which kernel instantiation the handler's most recent per-head-list launch (quest_decode_forward*, fused or not) took:
 * info = {keys per thread of the fused top-k front end (0: index-tensor launch), waves per workgroup, front-end variant
 * (0 / 1 / 3: first generation with scalar / vector-fed staging arrays / keys straight into registers; 2: second
 * generation; 8: tiles -- the rows carry tile maxima; 7: the one-launch layer, quest_decode_layer_fused_batched), 1 if the
 * one-variant instantiation was launched (0: the generic kernel), workgroups per head, sequences}.  Tests and benches
 * assert with it that they run the kernel they mean to. */
int quest_decode_last_launch_info(const quest_decode_handler_t* h, uint32_t info[6]);
/* Developer aid: the handler's partial-state workspace ([sequences][heads][workgroups per head][record_floats] fp32:
 * acc[head_dim], m, d, spare).  Builds with -DQUEST_WALLSTAMPS leave per-workgroup wall-clock stamps in the spare words
 * (scripts/wallstamps.py). */
int quest_decode_debug_workspace(const quest_decode_handler_t* h, void** ptr, uint64_t* bytes, uint32_t* record_floats);
/* Measurement aid: with skip != 0, quest_decode_forward* launch only the attention kernel and leave the
 * per-workgroup partial states in the handler's workspace (the output tensor is NOT written when the plan has
 * more than one workgroup per head).  Lets a bench time the dominant kernel by itself. */
int quest_decode_set_skip_merge(quest_decode_handler_t* h, int skip);
/* Inspection aid for the state-driven / batched fused launches (quest_decode_forward_fused_topk_dyn/_batched), which
 * otherwise keep the selected pages inside the kernel: while set, every such launch also writes its selection --
 * values (fp16 scores) to val_out and physical page ids to idx_out, both [n_seqs][num_qo_heads][n_selected_pages of
 * the plan] (rows of a sequence still shorter than the budget are filled up to its live page count).  NULL
 * pointers turn it off.  Not an entry the reference has: its top-k output lives in topk_filtering's tensors. */
int quest_decode_set_selection_out(quest_decode_handler_t* h, void* val_out, int32_t* idx_out);
/* Which top-k front end the fused launches use: 0 = automatic (by row length and alignment), 1 = first generation
 * (csrc/topk_select.cuh; rows up to 4096 pages), 2 = second generation (csrc/topk_bitmap.cuh; needs 8-byte aligned
 * score rows) without its histogram pre-filter, 3 = second generation with the pre-filter.  All are slot ownership and
 * implement the same selection: bit-identical page lists and outputs.  Tuning / test aid.  (The tiles front end is chosen
 * by calling quest_decode_forward_fused_topk_tiles_dyn.  Round 4's column-range ownership and third generation measured
 * slower and were removed in round 5.) */
int quest_decode_set_front_end(quest_decode_handler_t* h, int generation);
/* Override the planner (0 = automatic).  Used by tuning sweeps. */
int quest_decode_set_pages_per_chunk(quest_decode_handler_t* h, uint32_t pages_per_chunk);

/*
 * apply_rope_in_place (bsk_ops.h:23-27, page.cu:212-252 -> QKApplyRotaryInPlace,
 * decode_page.cuh:695-728).  q: [n][num_qo_heads][dim], k: [n][num_kv_heads][dim], in place.
 */
int quest_apply_rope_in_place(void* q, void* k, uint32_t n, uint32_t past_kv_len,
                              uint32_t num_qo_heads, uint32_t num_kv_heads, uint32_t head_dim,
                              float rope_scale, float rope_theta, quest_stream_t stream);

/* rms_norm_forward (bsk_ops.h:29-32, rms_norm.cu:160-212). input/output: [rows][cols], weight [cols]. */
int quest_rms_norm_forward(const void* input, const void* weight, void* output, uint32_t rows,
                           uint32_t cols, float epsilon, quest_stream_t stream);

/*
 * prefill_with_paged_kv_cache (bsk_ops.h:78-86, batch_prefill.cu:27-117 -> BatchPrefillWithPagedKVCache,
 * kernels/include/prefill/prefill.cuh:1008-1119, kernel :688-882).  q, o: [n_q][num_qo_heads][head_dim] fp16; the
 * sequence's K/V (the n_q new tokens included: append_kv_cache_prefill ran before, utils/__init__.py:127-170) are the
 * n_pages_host pages kv.indices lists, the last one holding kv.last_page_len tokens.  Row i attends the keys
 * 0 .. kv_len - n_q + i when `causal`, every key otherwise; no rotary (RotaryMode::kNone, batch_prefill.cu:102),
 * softmax scale 1/sqrt(head_dim).  GQA: num_qo_heads a multiple of kv.num_heads, query head h reads kv head
 * h / (num_qo_heads / kv.num_heads).  head_dim 64 / 128 / 256 (the set of the reference's SWITCH_HEAD_DIM_PREFILL,
 * prefill.cuh:1073), any page_size, both layouts; QUEST_EUNSUPPORTED otherwise;
 * causal with n_q > kv_len is QUEST_EINVAL (the reference assumes kv_len >= qo_len, test_prefill_attention.py:50-51).
 * MFMA flash kernel (csrc/prefill.hip): 128 query rows per workgroup, 64-key tiles, nothing but o is written.
 */
int quest_prefill_with_paged_kv_cache(const void* q, void* o, uint32_t n_q, uint32_t num_qo_heads,
                                      quest_paged_kv_t kv, uint32_t n_pages_host, int causal,
                                      quest_stream_t stream);

/*
 * Decode-token projections of a Llama decoder layer around the attention path, fused (EXTENSION; csrc/decode_layer.hip).
 * The reference leaves them to cuBLAS + PyTorch kernels: RMSNorm (quest/ops/csrc/rms_norm.cu:82-213 via
 * quest/models/llama.py:72), q/k/v projections (QuestAttention.py:88-90) + RoPE (:99, decode_page.cuh:644-728), o_proj (QuestAttention.py:175),
 * residual adds and the SwiGLU MLP (llama.py LlamaMLP).  All vectors / matrices fp16, row-major weights [out][in]
 * (torch.nn.Linear.weight), fp32 accumulation, batch 1.
 *   quest_decode_norm_gemv     out[out_dim] = W . rmsnorm(x; gamma, eps)      (gamma NULL: out = W . x)
 *   quest_decode_gemv_residual h[out_dim]  += W . x                           (in place on the residual stream)
 *   quest_decode_mlp_gate_up   act[intermediate] = silu(Wg . n) * (Wu . n),   n = rmsnorm(h; gamma, eps)
 *   quest_decode_qkv_rope      q, k, v = Wq . n, Wk . n, Wv . n with rotate-half RoPE on q and k at position
 *                              state->seq_len - 1 (call after quest_step_state_advance), n = rmsnorm(h; gamma, eps)
 */
int quest_decode_norm_gemv(const void* x, const void* gamma, float eps, const void* w, void* out, uint32_t in_dim,
                           uint32_t out_dim, quest_stream_t stream);
int quest_decode_gemv_residual(const void* x, const void* w, void* h, uint32_t in_dim, uint32_t out_dim,
                               quest_stream_t stream);
int quest_decode_mlp_gate_up(const void* h, const void* gamma, float eps, const void* w_gate, const void* w_up, void* act,
                             uint32_t hidden, uint32_t intermediate, quest_stream_t stream);
int quest_decode_qkv_rope(const void* h, const void* gamma, float eps, const void* wq, const void* wk, const void* wv,
                          void* q, void* k, void* v, uint32_t hidden, uint32_t num_qo_heads, uint32_t num_kv_heads,
                          uint32_t head_dim, float rope_scale, float rope_theta, const quest_step_state_t* state,
                          quest_stream_t stream);
/*
 * The same four launches for n_tokens <= 16 decode tokens at once -- one per sequence of a batch (the reference asserts
 * batch size 1; BatchedInferenceController): x / h `[n_tokens][in]`, outputs `[n_tokens][out]` (q / k / v
 * `[n_tokens][heads][head_dim]`), all contiguous; `states` = n_tokens consecutive quest_step_state_t records (the batched
 * step state), token i rotated at states[i].seq_len - 1.  The weights are read once for the whole batch (MFMA row-dots:
 * weights = A operand, tokens = the 16 columns of B); head_dim % 16 == 0 for the q/k/v launch.
 */
int quest_decode_norm_gemv_batched(const void* x, const void* gamma, float eps, const void* w, void* out, uint32_t in_dim,
                                   uint32_t out_dim, uint32_t n_tokens, quest_stream_t stream);
int quest_decode_gemv_residual_batched(const void* x, const void* w, void* h, uint32_t in_dim, uint32_t out_dim,
                                       uint32_t n_tokens, quest_stream_t stream);
int quest_decode_mlp_gate_up_batched(const void* h, const void* gamma, float eps, const void* w_gate, const void* w_up,
                                     void* act, uint32_t hidden, uint32_t intermediate, uint32_t n_tokens,
                                     quest_stream_t stream);
int quest_decode_qkv_rope_batched(const void* h, const void* gamma, float eps, const void* wq, const void* wk,
                                  const void* wv, void* q, void* k, void* v, uint32_t hidden, uint32_t num_qo_heads,
                                  uint32_t num_kv_heads, uint32_t head_dim, float rope_scale, float rope_theta,
                                  const quest_step_state_t* states, uint32_t n_tokens, quest_stream_t stream);

/* Which kernel an n-token launch of this shape takes (tests / tuning; nothing is launched): 1 and info = {k slices per
 * quad of rows, rounds, dynamic LDS bytes, workgroups, 128-wide k steps per row, weight fragments per set} for the persistent
 * kernel (inputs resident in LDS), 0 for the inputs-from-L2 kernel, < 0 for argument errors.  virtual_rows = output rows
 * of the launch (q + k + v rows; 2 x intermediate for the gate/up launch). */
int quest_decode_batched_plan(uint32_t in_dim, uint32_t virtual_rows, uint32_t n_tokens, int rope, uint32_t info[6]);

#ifdef __cplusplus
}
#endif
#endif /* QUEST_HIP_H_ */
