/*
 * quest_oracle.c -- CPU restatement of the Quest sparse-decode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product path (quest_amd/) never links, imports or calls this file.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against fixtures under tests/golden/ that were produced by importing the
 * reference's own pure-torch oracles (quest/tests/test_estimate.py:17-75,
 * test_approx_attention.py:17-110, test_decode_attention.py:17-44) in the build
 * container with tests/golden/make_golden.py.
 *
 * Every function cites the reference file:line it restates.  Arithmetic is
 * scalar C; fp16 is handled in software (round-to-nearest-even) so the result
 * does not depend on the host's half support.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint16_t qo_half;

/* ------------------------------------------------------------------ fp16 */

float qo_h2f(qo_half h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                man <<= 1;
                ++e;
            } while ((man & 0x400u) == 0);
            man &= 0x3ffu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* float -> half, round to nearest even (what static_cast<half>(float) does on
 * both CUDA __float2half_rn and gfx950 v_cvt_f16_f32 in the default mode). */
qo_half qo_f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { /* inf / nan */
        if (ax > 0x7f800000u) return (qo_half)(sign | 0x7e00u | ((ax >> 13) & 0x3ffu));
        return (qo_half)(sign | 0x7c00u);
    }
    if (ax >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (qo_half)(sign | 0x7c00u);
    }
    if (ax < 0x33000001u) { /* < 2^-25 (or == 2^-25 tie to even 0) */
        return (qo_half)sign;
    }
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7fffffu) | 0x800000u;
    uint32_t shift;
    uint32_t hexp;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t q = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t half_ulp = 1u << (shift - 1);
    if (rem > half_ulp || (rem == half_ulp && (q & 1u))) ++q;
    uint32_t out;
    if (hexp == 0) {
        out = q; /* q may reach 0x400 -> smallest normal, encoding is contiguous */
    } else {
        out = ((hexp - 1) << 10) + q; /* q has the implicit bit at 0x400 */
    }
    return (qo_half)(sign | out);
}

static int qo_isnan_h(qo_half h) { return (h & 0x7c00u) == 0x7c00u && (h & 0x3ffu) != 0; }

/* __hmax / __hmin (decode_page.cuh:367-378): NaN-suppressing, +0 > -0. */
static qo_half qo_hmax(qo_half a, qo_half b) {
    if (qo_isnan_h(a)) return qo_isnan_h(b) ? (qo_half)0x7fffu : b;
    if (qo_isnan_h(b)) return a;
    float fa = qo_h2f(a), fb = qo_h2f(b);
    if (fa == fb) return (a & 0x8000u) ? b : a; /* +0 beats -0 */
    return fa > fb ? a : b;
}
static qo_half qo_hmin(qo_half a, qo_half b) {
    if (qo_isnan_h(a)) return qo_isnan_h(b) ? (qo_half)0x7fffu : b;
    if (qo_isnan_h(b)) return a;
    float fa = qo_h2f(a), fb = qo_h2f(b);
    if (fa == fb) return (a & 0x8000u) ? a : b; /* -0 beats +0 */
    return fa < fb ? a : b;
}

#define QO_HALF_MAX ((qo_half)0x7bffu)     /* +65504 = CUDART_MAX_NORMAL_FP16 */
#define QO_HALF_NEG_MAX ((qo_half)0xfbffu) /* -65504 */

/* ------------------------------------------------------- paged pool view */

/* The fields of paged_kv_t (decode_page.cuh:79-110) that the path reads, for
 * batch_size == 1 (page.cu:19, estimate.cu:14, approx_attn.cu:78). */
typedef struct {
    uint32_t num_heads;     /* heads stored in the pool (kv heads) */
    uint32_t page_size;     /* entries per page */
    uint32_t head_dim;
    uint32_t layout;        /* 0 = NHD, 1 = HND (quest/utils/utils.py:1-5); 2 = NHD_ROT, this build's extension (below) */
    qo_half* data;          /* [max_pages][2][...] */
    const int32_t* indices; /* [n_pages] page table */
    int32_t n_pages;        /* indptr[1] - indptr[0] */
    uint32_t last_page_len; /* valid entries in the last page, 1..page_size */
} qo_paged_t;

/* Layout 2 (NHD_ROT, include/quest_hip.h) is NOT a layout of the reference: the NHD shape with the heads of an entry
 * rotated by the entry -- the K / max vector of head h sits in head slot h ^ (e & rot), its V / min vector in that
 * slot ^ flip.  The oracle restates only WHERE the vectors live; every algorithm below is the reference's and reads the
 * pool through these two functions, so the same inputs give the same results in all three layouts. */
static uint32_t qo_rot(const qo_paged_t* p) {
    uint32_t low = p->num_heads & (0u - p->num_heads);
    return p->layout == 2 ? (low < 4u ? low : 4u) - 1u : 0u;
}
static uint32_t qo_flip(const qo_paged_t* p) {
    uint32_t low = p->num_heads & (0u - p->num_heads);
    return p->layout == 2 ? ((low < 32u ? low : 32u) - 1u) & ~3u : 0u;
}
/* decode_page.cuh:196-205 */
static size_t qo_k_off(const qo_paged_t* p, size_t page, size_t head, size_t entry, size_t feat) {
    size_t slot = head ^ (entry & qo_rot(p));
    return p->layout == 1
               ? ((page * 2 * p->num_heads + head) * p->page_size + entry) * p->head_dim + feat
               : ((page * 2 * p->page_size + entry) * p->num_heads + slot) * p->head_dim + feat;
}
/* decode_page.cuh:228-239 */
static size_t qo_v_off(const qo_paged_t* p, size_t page, size_t head, size_t entry, size_t feat) {
    size_t slot = (head ^ (entry & qo_rot(p))) ^ qo_flip(p);
    return p->layout == 1
               ? (((page * 2 + 1) * p->num_heads + head) * p->page_size + entry) * p->head_dim +
                     feat
               : (((page * 2 + 1) * p->page_size + entry) * p->num_heads + slot) * p->head_dim +
                     feat;
}

/* ---------------------------------------------------------------- append */

/* AppendPagedKVCacheDecodeKernel, decode_page.cuh:398-449.
 * key/value: [1][num_heads][head_dim].  The metadata pool has the same struct;
 * its "entries" are KV pages, K slot = running max, V slot = running min. */
void qo_append_decode(const qo_paged_t* kv, const qo_paged_t* meta, const qo_half* key,
                      const qo_half* value) {
    uint32_t S = kv->page_size, H = kv->num_heads, D = kv->head_dim;
    uint32_t seq_len = (uint32_t)(kv->n_pages - 1) * S + kv->last_page_len; /* :408-410 */
    uint32_t page_iter = (seq_len - 1) / S;                                 /* :413 */
    uint32_t entry = (seq_len - 1) % S;                                     /* :414 */
    size_t page = (size_t)kv->indices[page_iter];
    size_t mpage = (size_t)meta->indices[meta->n_pages - 1]; /* :419 */
    uint32_t mentry = meta->last_page_len - 1;               /* :420 */
    for (uint32_t h = 0; h < H; ++h) {
        for (uint32_t d = 0; d < D; ++d) {
            qo_half mx, mn;
            if (entry > 0) { /* :424-432 */
                mx = meta->data[qo_k_off(meta, mpage, h, mentry, d)];
                mn = meta->data[qo_v_off(meta, mpage, h, mentry, d)];
            } else {
                mx = QO_HALF_NEG_MAX;
                mn = QO_HALF_MAX;
            }
            qo_half k = key[(size_t)h * D + d];
            mx = qo_hmax(mx, k); /* :440-441 */
            mn = qo_hmin(mn, k);
            kv->data[qo_k_off(kv, page, h, entry, d)] = k;
            meta->data[qo_k_off(meta, mpage, h, mentry, d)] = mx; /* :443-446 */
            meta->data[qo_v_off(meta, mpage, h, mentry, d)] = mn;
            kv->data[qo_v_off(kv, page, h, entry, d)] = value[(size_t)h * D + d];
        }
    }
}

/* AppendPagedKVCachePrefillKernel, decode_page.cuh:471-562.
 * key/value: [append_len][num_heads][head_dim]; the pool already accounts for
 * the appended tokens (seq_len includes them). */
void qo_append_prefill(const qo_paged_t* kv, const qo_paged_t* meta, const qo_half* key,
                       const qo_half* value, int32_t append_len) {
    int32_t S = (int32_t)kv->page_size, H = (int32_t)kv->num_heads, D = (int32_t)kv->head_dim;
    int32_t MS = (int32_t)meta->page_size;
    int32_t page_nums = kv->n_pages;
    int32_t seq_len = (page_nums - 1) * S + (int32_t)kv->last_page_len; /* :485 */
    int32_t start_seq = seq_len - append_len;                           /* :487 */
    for (int32_t po = start_seq / S; po < page_nums; ++po) {            /* :489-492 */
        size_t mpage = (size_t)meta->indices[po / MS];                  /* :496 */
        int32_t mentry = po % MS;                                       /* :497 */
        size_t page = (size_t)kv->indices[po];
        int32_t e0 = start_seq - po * S;
        if (e0 < 0) e0 = 0; /* :500-501 */
        int32_t e1 = seq_len - po * S;
        if (e1 > S) e1 = S; /* :502-504 */
        for (int32_t h = 0; h < H; ++h) {
            for (int32_t d = 0; d < D; ++d) {
                qo_half mx, mn;
                if (e0 > 0) { /* :509-523 */
                    mx = meta->data[qo_k_off(meta, mpage, h, mentry, d)];
                    mn = meta->data[qo_v_off(meta, mpage, h, mentry, d)];
                } else {
                    mx = QO_HALF_NEG_MAX;
                    mn = QO_HALF_MAX;
                }
                for (int32_t e = e0; e < e1; ++e) { /* :525-550 */
                    size_t src = ((size_t)(po * S + e - start_seq) * H + h) * D + d;
                    qo_half k = key[src];
                    mx = qo_hmax(mx, k);
                    mn = qo_hmin(mn, k);
                    kv->data[qo_k_off(kv, page, h, e, d)] = k;
                    kv->data[qo_v_off(kv, page, h, e, d)] = value[src];
                }
                meta->data[qo_k_off(meta, mpage, h, mentry, d)] = mx; /* :551-560 */
                meta->data[qo_v_off(meta, mpage, h, mentry, d)] = mn;
            }
        }
    }
}

/* -------------------------------------------------------------- estimate */

/* MaxPossibleSampleWithPagedKVCacheKernel, decode_attn.cuh:245-401 with the
 * arithmetic of compute_max_possible, :137-168.
 *
 * out[qo_head][e] for e in [0, chunk_len), chunk_len = (n_meta_pages-1)*S +
 * meta.last_page_len - 1 (:266-272: the current KV page is excluded).
 *
 * Summation: vec_size = 8 consecutive features per lane accumulated left to
 * right in fp32 from 0.f (as the reference kernel, :152-156; bdx = head_dim / 8
 * lanes per row, decode_attn.cuh:1108-1110 with fp16).  The cross-lane step is
 * THIS BUILD's, chosen for CDNA4's DPP unit, and the HIP kernel implements the
 * same tree (quest_common.cuh row_allreduce_sum_fast), so HIP == oracle bit for
 * bit: within each 16-lane row, four rotation steps lane[i] += lane[(i-n) mod 16]
 * for n = 8, 4, 2, 1 (all lanes updated simultaneously); 32-lane rows (D=256) add
 * lane[i ^ 16] last; 8-lane rows (D=64) use xor 1, xor 2, then the mirror 7-i.
 * The result is lane 0's value, cast once to fp16 (RNE).  The reference kernel
 * uses an xor butterfly instead (:157-160); either order is far inside its own
 * tolerance (5e-3; measured <= 4 fp16 ulp against the reference's torch oracle).
 * fp16 x fp16 products are exact in fp32, so no FMA-contraction ambiguity. */
static void qo_row_reduce(float* lane, uint32_t w) {
    float nxt[32];
    if (w == 8) {
        for (uint32_t off = 1; off <= 2; off <<= 1) {
            for (uint32_t i = 0; i < 8; ++i) nxt[i] = lane[i] + lane[i ^ off];
            memcpy(lane, nxt, sizeof(float) * 8);
        }
        for (uint32_t i = 0; i < 8; ++i) nxt[i] = lane[i] + lane[7 - i];
        memcpy(lane, nxt, sizeof(float) * 8);
        return;
    }
    for (uint32_t n = 8; n >= 1; n >>= 1) {
        for (uint32_t i = 0; i < w; ++i) {
            uint32_t base = i & ~15u, l = i & 15u;
            nxt[i] = lane[i] + lane[base + ((l + 16u - n) & 15u)];
        }
        memcpy(lane, nxt, sizeof(float) * w);
    }
    if (w == 32) {
        for (uint32_t i = 0; i < 32; ++i) nxt[i] = lane[i] + lane[i ^ 16u];
        memcpy(lane, nxt, sizeof(float) * 32);
    }
}

void qo_estimate(const qo_half* q, const qo_paged_t* meta, uint32_t num_qo_heads, qo_half* out) {
    uint32_t S = meta->page_size, Hkv = meta->num_heads, D = meta->head_dim;
    uint32_t group = num_qo_heads / Hkv; /* qo_head = kv_head*bdy + ty, :256 */
    uint32_t chunk_len = meta->n_pages > 0
                             ? (uint32_t)(meta->n_pages - 1) * S + (meta->last_page_len - 1)
                             : 0;
    const uint32_t vec = 8;
    uint32_t bdx = D / vec;
    float lane[32];
    for (uint32_t hq = 0; hq < num_qo_heads; ++hq) {
        uint32_t hk = hq / group;
        const qo_half* qh = q + (size_t)hq * D;
        for (uint32_t e = 0; e < chunk_len; ++e) {
            size_t mpage = (size_t)meta->indices[e / S];
            uint32_t slot = e % S;
            for (uint32_t tx = 0; tx < bdx; ++tx) {
                float acc = 0.f;
                for (uint32_t i = 0; i < vec; ++i) {
                    uint32_t d = tx * vec + i;
                    float qf = qo_h2f(qh[d]);
                    float a = qf * qo_h2f(meta->data[qo_k_off(meta, mpage, hk, slot, d)]);
                    float b = qf * qo_h2f(meta->data[qo_v_off(meta, mpage, hk, slot, d)]);
                    acc += fmaxf(a, b);
                }
                lane[tx] = acc;
            }
            qo_row_reduce(lane, bdx);
            out[(size_t)hq * chunk_len + e] = qo_f2h(lane[0]);
        }
    }
}

/* ------------------------------------------------------------------ top-k */

/* Order-preserving 16-bit key of an fp16 bit pattern, as RAFT's radix select
 * twiddles it (raft branch-24.02, matrix/detail/select_radix.cuh twiddle_in,
 * select_min = false; call site kernels/include/topk/decode_select_k.cuh:38).
 * RAFT is an un-vendored dependency; its published algorithm is restated. */
static uint16_t qo_key(qo_half h) { return (h & 0x8000u) ? (uint16_t)~h : (uint16_t)(h | 0x8000u); }

/* decode_select_k, decode_select_k.cuh:25-62, one row.
 * Selects the k largest of n fp16 values and emits (value, in_idx[col]).
 *
 * The reference guarantees only the selected value multiset (test_topk.py:10-13,
 * 62-64); tie order and output order are unspecified (atomic arrival order).
 * This build's declared rule (SURVEY.md section 8a row T-tie), which the HIP
 * kernel must match bit for bit:
 *   - every value with key > threshold key is selected;
 *   - among values with key == threshold, the lowest columns are selected first;
 *   - output is in ascending column order. */
void qo_topk_row(const qo_half* val, const int32_t* in_idx, uint32_t n, uint32_t k,
                 qo_half* out_val, int32_t* out_idx) {
    if (k == 0) return;
    if (k > n) k = n;
    uint32_t* hist = (uint32_t*)calloc(65536, sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) hist[qo_key(val[i])]++;
    uint32_t above = 0;
    int32_t thr = 65535;
    for (; thr >= 0; --thr) {
        if (above + hist[thr] >= k) break;
        above += hist[thr];
    }
    free(hist);
    uint32_t need_eq = k - above;
    uint32_t w = 0;
    for (uint32_t i = 0; i < n && w < k; ++i) {
        uint16_t key = qo_key(val[i]);
        int take = 0;
        if ((int32_t)key > thr) {
            take = 1;
        } else if ((int32_t)key == thr && need_eq > 0) {
            take = 1;
            --need_eq;
        }
        if (take) {
            out_val[w] = val[i];
            out_idx[w] = in_idx[i];
            ++w;
        }
    }
}

/* topk_filtering, quest/ops/csrc/topk.cu:7-46: rows = heads. */
void qo_topk(const qo_half* val, const int32_t* in_idx, uint32_t rows, uint32_t n, uint32_t k,
             qo_half* out_val, int32_t* out_idx) {
    for (uint32_t r = 0; r < rows; ++r)
        qo_topk_row(val + (size_t)r * n, in_idx + (size_t)r * n, n, k, out_val + (size_t)r * k,
                    out_idx + (size_t)r * k);
}

/* ------------------------------------------------------- sparse attention */

/* BatchDecodeWithPagedKVCacheKernel, decode_attn.cuh:440-646, stated as the
 * scalar reference the repo's own C++ test uses (selected_single_mha,
 * kernels/src/include/cpu_reference.h:162-292): softmax(q.K^T / sqrt(D)).V over
 * the tokens of the n_sel indexed pages plus last_page_len tokens of
 * last_page_idx (decode_page.cuh:325-351: per-head index row with row stride
 * idx_stride; the last page is appended for every head).
 *
 * q: [Hq][D]; idx: [Hq][idx_stride] physical page ids; out: [Hq][D] fp16.
 * Accumulation is in double so the oracle is the mathematically tightest
 * statement; the reference kernels accumulate in fp32 and the repo's tolerance
 * for this stage is 5e-3 (test_approx_attention.py:10-15). */
void qo_sparse_attn(const qo_half* q, const qo_paged_t* kv, const int32_t* idx,
                    uint32_t idx_stride, uint32_t n_sel, int32_t last_page_idx,
                    uint32_t last_page_len, uint32_t num_qo_heads, qo_half* out, float* lse_out) {
    uint32_t S = kv->page_size, Hkv = kv->num_heads, D = kv->head_dim;
    uint32_t group = num_qo_heads / Hkv;
    double sm_scale = 1.0 / sqrt((double)D);
    size_t n_tok = (size_t)n_sel * S + last_page_len;
    double* att = (double*)malloc(sizeof(double) * n_tok);
    double* acc = (double*)malloc(sizeof(double) * D);
    for (uint32_t hq = 0; hq < num_qo_heads; ++hq) {
        uint32_t hk = hq / group;
        const qo_half* qh = q + (size_t)hq * D;
        double mx = -INFINITY;
        size_t t = 0;
        for (uint32_t s = 0; s <= n_sel; ++s) {
            size_t page = s < n_sel ? (size_t)idx[(size_t)hq * idx_stride + s] : (size_t)last_page_idx;
            uint32_t len = s < n_sel ? S : last_page_len;
            for (uint32_t e = 0; e < len; ++e, ++t) {
                double dot = 0.0;
                for (uint32_t d = 0; d < D; ++d)
                    dot += (double)qo_h2f(qh[d]) *
                           (double)qo_h2f(kv->data[qo_k_off(kv, page, hk, e, d)]);
                att[t] = dot * sm_scale;
                if (att[t] > mx) mx = att[t];
            }
        }
        double denom = 0.0;
        for (size_t i = 0; i < n_tok; ++i) {
            att[i] = exp(att[i] - mx);
            denom += att[i];
        }
        for (uint32_t d = 0; d < D; ++d) acc[d] = 0.0;
        t = 0;
        for (uint32_t s = 0; s <= n_sel; ++s) {
            size_t page = s < n_sel ? (size_t)idx[(size_t)hq * idx_stride + s] : (size_t)last_page_idx;
            uint32_t len = s < n_sel ? S : last_page_len;
            for (uint32_t e = 0; e < len; ++e, ++t) {
                double p = att[t];
                for (uint32_t d = 0; d < D; ++d)
                    acc[d] += p * (double)qo_h2f(kv->data[qo_v_off(kv, page, hk, e, d)]);
            }
        }
        for (uint32_t d = 0; d < D; ++d) out[(size_t)hq * D + d] = qo_f2h((float)(acc[d] / denom));
        if (lse_out) lse_out[hq] = (float)(mx + log(denom)); /* natural-log lse */
    }
    free(att);
    free(acc);
}

/* ------------------------------------------------------------ rope / norm */

/* QKApplyRotaryInPlaceKernel, decode_page.cuh:644-692 (+ flashinfer's
 * vec_apply_llama_rope, un-vendored: rotate-half form).  x: [N][H][D] in place,
 * position of row i is past_len + i.
 * freq_i = (1/rope_scale) * (1/rope_theta)^(2*(i mod D/2)/D)  (:658-660). */
void qo_rope(qo_half* x, uint32_t n, uint32_t H, uint32_t D, uint32_t past_len, float rope_scale,
             float rope_theta) {
    uint32_t half = D / 2;
    float* tmp = (float*)malloc(sizeof(float) * D);
    for (uint32_t i = 0; i < n; ++i) {
        for (uint32_t h = 0; h < H; ++h) {
            qo_half* row = x + ((size_t)i * H + h) * D;
            for (uint32_t d = 0; d < D; ++d) {
                double freq = (1.0 / (double)rope_scale) *
                              pow(1.0 / (double)rope_theta, (double)(2 * (d % half)) / (double)D);
                double ang = (double)(past_len + i) * freq;
                double c = cos(ang), s = sin(ang);
                double self = (double)qo_h2f(row[d]);
                double other = d < half ? -(double)qo_h2f(row[d + half]) : (double)qo_h2f(row[d - half]);
                tmp[d] = (float)(self * c + other * s);
            }
            for (uint32_t d = 0; d < D; ++d) row[d] = qo_f2h(tmp[d]);
        }
    }
    free(tmp);
}

/* rmsnorm_twoPassAlgo_e8, quest/ops/csrc/rms_norm.cu:82-158.
 * out = half( float(x) * rsqrt(mean(x^2) + eps) * float(w) ). */
void qo_rms_norm(const qo_half* x, const qo_half* w, uint32_t rows, uint32_t cols, float eps,
                 qo_half* out) {
    for (uint32_t r = 0; r < rows; ++r) {
        double ss = 0.0;
        for (uint32_t c = 0; c < cols; ++c) {
            double v = (double)qo_h2f(x[(size_t)r * cols + c]);
            ss += v * v;
        }
        float inv = (float)(1.0 / sqrt(ss / (double)cols + (double)eps));
        for (uint32_t c = 0; c < cols; ++c)
            out[(size_t)r * cols + c] =
                qo_f2h(qo_h2f(x[(size_t)r * cols + c]) * inv * qo_h2f(w[c]));
    }
}
