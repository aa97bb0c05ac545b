// Developer aids of the attention kernels, all in one place: two measurement builds leave time stamps behind; a product
// build compiles every macro below to nothing (tests/test_build_metadata.py guards the product kernels' resources).
//
//   -DQUEST_TIMELINE   (scripts/timeline.py)   ONE workgroup of a launch writes cycle stamps of its phases into the `lse`
//                      buffer instead of the log-sum-exp: QUEST_STAMP(i) in the kernel bodies, QUEST_SUBSTAMP(i) inside the
//                      selection routines (topk_select.cuh).  Warm, one workgroup: the shape of a phase, not of a launch.
//   -DQUEST_WALLSTAMPS (scripts/wallstamps.py) EVERY workgroup leaves wall-clock stamps (100 MHz, chip-wide: kernel entry,
//                      page list known, wave 0's pages folded, result written) and its page count in the spare floats of
//                      its partial-state record: the distribution over the workgroups of a launch in the cold regime the
//                      bench measures.
#pragma once

#ifdef QUEST_TIMELINE
#define QUEST_LSE_ENABLED false
#define QUEST_STAMP(i) \
    do { __builtin_amdgcn_s_waitcnt(0); tl[i] = clock64(); } while (0)
#define QUEST_SUBSTAMP(i) \
    do { if (sub) { __builtin_amdgcn_s_waitcnt(0); sub[i] = clock64(); } } while (0)
#define QUEST_TL_PARAM , long long* tl
#define QUEST_TL_ARG , tl
// at the top of a kernel body: the stamp arrays (tl = phases, sub_out = inside the selection) and the wall clock at entry
#define QUEST_TL_BEGIN                  \
    long long tl[10] = {};              \
    long long sub_out[9] = {};          \
    const long long wall0 = wall_clock64(); \
    QUEST_STAMP(0);
#define QUEST_TL_SUB sub_out            // what a body hands to a selection routine's `sub` parameter
#define QUEST_TL_SUB_ARG , sub_out      // ... as a trailing argument of routines that only take it in this build
#define QUEST_TL_SUB_PARAM , long long* sub
#ifdef QUEST_TL_FIRST_HEAD  // stamp a workgroup of head 0 (the FIRST workgroup dispatched to its CU) instead of the middle
#define QUEST_TL_HEAD(n) 0u  // head (a second one, which waits for the first one's issue slots)
#else
#define QUEST_TL_HEAD(n) ((n) / 2)
#endif
// at the end of a kernel body: the stamped workgroup (middle chunk of QUEST_TL_HEAD) reports through the lse buffer
#define QUEST_TL_REPORT(p, chunk, hq, seq, num_qo_heads)                                                                    \
    do {                                                                                                                    \
        QUEST_STAMP(9);                                                                                                     \
        if ((p).lse && (chunk) == (p).n_chunks / 2 && (hq) == QUEST_TL_HEAD(num_qo_heads) && (seq) == 0 && threadIdx.x == 0) { \
            for (int i_ = 0; i_ < 10; ++i_) (p).lse[i_] = (float)(tl[i_] - tl[0]);                                          \
            for (int i_ = 0; i_ < 9; ++i_) (p).lse[16 + i_] = (float)(sub_out[i_] - tl[0]);                                 \
            (p).lse[10] = (float)(wall_clock64() - wall0); /* 100 MHz ticks over the same span as tl[9] - tl[0] */          \
        }                                                                                                                   \
    } while (0)
#else
#define QUEST_LSE_ENABLED true
#define QUEST_STAMP(i) \
    do { } while (0)
#define QUEST_SUBSTAMP(i) \
    do { } while (0)
#define QUEST_TL_PARAM
#define QUEST_TL_ARG
#define QUEST_TL_BEGIN
#define QUEST_TL_SUB nullptr
#define QUEST_TL_SUB_ARG
#define QUEST_TL_SUB_PARAM
#define QUEST_TL_REPORT(p, chunk, hq, seq, num_qo_heads) \
    do { } while (0)
#endif

#ifdef QUEST_WALLSTAMPS
#define QUEST_WS_PARAM , const unsigned ws_entry, const unsigned ws_fe
#define QUEST_WS_ARG , ws_entry, (unsigned)wall_clock64()
#define QUEST_WS_ENTRY const unsigned ws_entry = (unsigned)wall_clock64();
#define QUEST_WS_NOW(name) const unsigned name = (unsigned)wall_clock64();
// the workgroup's stamps + page count into the spare words of a partial-state record that starts at `rec` (floats)
#define QUEST_WS_RECORD(rec, D_, ws_stride, n_pages)                                             \
    do {                                                                                         \
        if ((rec) && (ws_stride) >= (uint32_t)(D_) + 8u) {                                       \
            unsigned* u_ = reinterpret_cast<unsigned*>((rec) + (D_) + 2);                        \
            u_[0] = ws_entry, u_[1] = ws_fe, u_[2] = ws_gather, u_[3] = (unsigned)wall_clock64(); \
            u_[4] = (n_pages);                                                                   \
        }                                                                                        \
    } while (0)
#define QUEST_WS_RECORD_EXTRA(rec, D_, ws_stride, slot, value)                                   \
    do {                                                                                         \
        if ((rec) && (ws_stride) >= (uint32_t)(D_) + 8u) reinterpret_cast<unsigned*>((rec) + (D_) + 2)[slot] = (value); \
    } while (0)
#else
#define QUEST_WS_PARAM
#define QUEST_WS_ARG
#define QUEST_WS_ENTRY
#define QUEST_WS_NOW(name)
#define QUEST_WS_RECORD(rec, D_, ws_stride, n_pages) \
    do { } while (0)
#define QUEST_WS_RECORD_EXTRA(rec, D_, ws_stride, slot, value) \
    do { } while (0)
#endif
