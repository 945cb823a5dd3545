// ems_plan.h -- data shared between the host orchestration (ems_core.cpp) and the gfx950 kernels
// (ems_kernels.hip).  Plain structs only.
#pragma once
#include <stdint.h>

// One message entry as it lives in LDS during a check-node visit: the LLR and the GF symbol in the
// *binary-image* domain (GF addition == XOR there, which replaces the q*q ADDGF table of
// init.c:37-55).  bit 8 of `s` marks an invalid entry (the reference's symbol index -1 paired with the
// value 1e5, bubble_decoder.c:370-374); bit 9 is used transiently inside the bubble loop.
struct EmsEntry {
    float v;
    uint32_t s;
};

#define EMS_SENTINEL 1e5f
#define EMS_S_INVALID 0x100u
#define EMS_S_BOUND 0x200u
#define EMS_LH_NOAPP 0x10000   // t_lh: this edge row keeps its APP row and decision to itself (a later edge of the check node names the same variable node)
#define EMS_MAX_NM 64      // messages longer than this are not built (reference usage: 8..30; 40 until round 4); above EMS_FAST_NM the literal forms run
#define EMS_FAST_NM 32     // the merge form of the elementary step records a list position in 5 bits, the selection keeps 32 survivors
#define EMS_MAX_STEPS 14   // dc-2 forward/backward steps of the kernels that keep a stage table (dc <= 16)
#define EMS_MAX_DC 64      // check degrees 17..64 run k_decode_wide (edges in passes, tasks computed from the degree); LDS permitting

// Compact check-to-variable record kept in HBM per (frame, edge).  The reference stores a dense
// q-wide float row per edge (init.c:323-328) although a C->V message is nm (LLR, symbol) pairs plus
// one default value (bubble_decoder.c:262-278); the values are copies, never re-summed, so the compact
// form is bit-exact.  Layout (bytes): [0,4nm) float vals[nm]; [4nm,4nm+4) float dflt;
// [4nm+4,4nm+8) float hint (the selection threshold that worked for this edge in the previous sweep:
// a search start only, never part of the result); [4nm+8] uint8 cnt; [4nm+9, 5nm+9) uint8 sym[nm]
// (symbol INDEX after the inverse rotation); padded to a multiple of 16.
#ifndef EMS_REC_ALIGN
#define EMS_REC_ALIGN 16
#endif
#define EMS_Q16T_SCR 1152   /* 8 APP rows | 8 records | 8 decisions | 8 row numbers, see q16t_redo_trav */
static inline int ems_rec_bytes(int nm) { return ((5 * nm + 9) + EMS_REC_ALIGN - 1) & ~(EMS_REC_ALIGN - 1); }
#define EMS_REC_DFLT(nm) (4 * (nm))
#define EMS_REC_HINT(nm) (4 * (nm) + 4)
#define EMS_REC_CNT(nm) (4 * (nm) + 8)
#define EMS_REC_SYM(nm) (4 * (nm) + 9)

// v2 flavour: the two messages of an edge as 64-byte-aligned records next to each other (round 3; five arrays of 20 / 40 / 80-byte
// pieces per edge before: every piece cost whole 64-byte sectors, 1.25 x the algorithmic traffic).
//   V->C record: float vals[nm] | uint8 sym[nm] (symbol index, always valid) | float hint (the edge's last selection threshold: a
//                search start only) at EMS_V2_HINT
//   C->V record: float vals[nm] (-1000 behind the valid prefix) | int16 sym[nm], -1 = none (v2.c:1854-1870)
#if defined(__HIPCC__)
#define EMS_HD __host__ __device__
#else
#define EMS_HD
#endif
EMS_HD static inline int ems_v2_rec_bytes(int nm) { return ((6 * nm > 5 * nm + 8 ? 6 * nm : 5 * nm + 8) + 63) & ~63; }
#define EMS_V2_SYMS(nm) (4 * (nm))
#define EMS_V2_HINT(nm) ((5 * (nm) + 3) & ~3)

struct EmsPlan {
    // code
    int N, M, E, q, logq, dc;
    // decoder parameters
    int nm, nbOper, nsweeps, early_exit;
    float offset;
    // schedule: check nodes grouped in levels; no two in a level share a variable node
    int n_levels;
    const int *lvl_ptr;     // [n_levels+1]
    const int *lvl_cn;      // [M]
    const int *cn_vn;       // [M*dc] variable node of edge (cn*dc+i)      (code_t.mat)
    const uint8_t *cn_lh;   // [M*dc] log_alpha of the edge coefficient = index-1 (code_t.matValue)
    // the same check nodes as a task list in dependency (level) order, for the dataflow scheduler
    const int *t_cn;        // [M]    check node of task i (bits 0..23) | SIMD priority 0..3 (bits 24..25)
    const int *t_vn;        // [M*dc] its variable nodes
    const int *t_lh;        // [M*dc] log of its coefficients
    const int *t_pred;      // [M*dc] task index of the previous visitor of that variable node in this sweep, or -1
    // irregular check degrees (dc = the largest; tables above keep its stride): per task its own degree and its first edge; null = regular
    const int *t_dc;        // [M]
    const int *t_e0;        // [M]
    const int *cn_ptr;      // [M+1] first edge of every check node (cn_vn / cn_lh are flat [E])
    // the same check nodes in PAIRS for the two-visits-per-wave kernel (k_decode_dual; dc = 4 or 8): the members of a pair
    // share no variable node and every predecessor of either lies in an earlier pair
    int n_pairs, dual;      // dual: launch k_decode_dual (wave_lds_bytes / waves_per_wg then describe ITS workgroup)
    int pair16;             // launch k_decode_pair16 instead: the same pairs, row-wise phases one visit at a time on rows of 16 lanes, sixteen waves per CU (round 5)
    // GF(16) / dc = 8 / nm = 8 in lane-serial form (k_decode_q16t): the check nodes in GROUPS of up to 16 mutually independent ones, group i
    // depending only on groups < i
    int q16t, n_groups;     // q16t: launch k_decode_q16t
    int q16t_overlap;       // its opt-in barrier-free form of a fixed-work run (EMS_Q16T_OVERLAP=1): Syndrom only after the last sweep
    const int *q_cn;        // [n_groups*16]  check node of slot s, or -1
    const int *q_vn;        // [n_groups*128] variable node of edge (slot * 8 + t); 0 for an empty slot
    const int *q_lh;        // [n_groups*128] log of its coefficient (0..14)
    const int *q_pred;      // [n_groups*128] the previous visitor of that variable node: (group << 1) | 1 = an earlier group of this sweep,
                            //                (group << 1) | 0 = none in this sweep: the node's LAST visitor of the sweep before; -1 = empty slot
    // ... with the APP row TRAVELLING (round 5): one 128-byte line per edge in q_trav = [C->V record 64 B | APP row 64 B of the edge's variable node];
    // a visit reads its own line (record + row: one full line) and writes the new record there and the new APP row into the line of the node's
    // OTHER edge, which is the next to read it (variable degree 2).  The first visit of a node in sweep 0 reads the row from `app`.
    int q16t_trav;          // launch k_decode_q16t<.., true>
    const unsigned *q_vo;   // [n_groups*128] variable node (bits 0..13) | the node's other edge (14..28) | this edge is the node's first visit of a sweep (29)
    uint8_t *q_trav;        // [B][E][128]
    uint8_t *q_scr;         // [B][waves][EMS_Q16T_SCR] staging of the rare merge-form redo of one check node (process_cn on the plain layout)
    int wide;               // launch k_decode_wide (check degree above 16: the edges of a visit in passes; t_dc / t_e0 always set)
    // k_decode_wide keeps the 4 dc - 6 lists of a visit (inputs, prefixes, suffixes, merges: bubble_decoder.c:166-227) in FEWER buffers: a result
    // takes the buffer of a list whose last reader is an EARLIER stage (host: wide_buffer_plan; dc = 20: 42 buffers for 74 lists).
    const uint8_t *wide_buf;  // [EMS_MAX_DC + 1][256] buffer of list id, per check degree
    int wide_nbuf;            // buffers the wave's list area holds (the largest need among the code's degrees)
    int twin;               // launch k_decode_twin: one wave per PAIR OF FRAMES, check nodes in reference order (tiny codes)
    const int *p_cn;        // [n_pairs*2] check node (bits 0..23) | SIMD priority of the pair (24..25) | bit 30: no second member (entry repeats the first)
    const int *p_vn;        // [n_pairs*2dc] variable nodes, row = member * dc + edge
    const int *p_lh;        // [n_pairs*2dc] log of the coefficients
    const int *p_pred;      // [n_pairs*2dc] PAIR index of the previous visitor of that variable node in this sweep, or -1
    const unsigned *p_pack; // [n_pairs*8] (k_decode_pair16; dc = 4, N <= 8192, n_pairs <= 2046) p_vn | p_lh << 13 | (p_pred + 1) << 21; null otherwise
    int nsteps;             // dc-2 forward/backward steps
    int8_t out_list[16];    // list id holding the output for edge t
    int n_lists;
    // field tables
    const uint8_t *idx2bin;
    const uint8_t *bin2idx;
    // per-batch state in HBM
    float *app;             // [B][N][q]
    uint8_t *ctov;          // [B][E][rec_bytes]
    uint8_t *decide;        // [B][N]
    int *iters;             // [B]
    int *synd;              // [B]
    int *fail;              // one word per handle: set (never cleared by a kernel) when a dependency wait gave up, see ems_get_results
    int rec_bytes;
    int wave_lds_bytes;     // LDS per wave
    int waves_per_wg;
    int occ4;               // launch the 128-VGPR build (four 4-wave workgroups per CU), see ems_core.cpp waves_for_batch
    unsigned long long *stamps; // diagnostic builds only (EMS_STAMPS), else null
    // ---- v2 flavour (truncated messages both ways, max-is-best; EMS_HS_L-BubbleCheck_UBS_decoder_v2.c) ----
    const float *v2_like;   // [B][N][q]  likelihood of symbol g of node n (decoder.likelihood scattered by likelihood_index)
    uint8_t *v2_msg;        // [B][E][2][v2_rec] per edge (row order) its V->C record, then its C->V record (ems_v2_rec_bytes)
    int v2_rec;             // bytes of one record
    float *v2_app;          // [B][N][q]  APP of the last iteration run (ComputeAPPLog)
    const int *t_dst;       // [M*dc]     per (task, edge): row-order id of the OTHER edge of its variable node (dv = 2)
    const int *vn_edge;     // [N][2]     NtoB: the two edges of a variable node in row order
    const uint8_t *t_last;  // [M*dc]     per (task, edge): 1 when this check node is the LAST visitor of the edge's variable node in a sweep
    const int *p_dst;       // [n_pairs*2dc] t_dst by (pair, row) for the two-visit kernel
    const int *p_last;      // [n_pairs*2dc] t_last by (pair, row)
    // the statistics of the older generation's main() (v2.c:2174-2243): bit errors of the decisions in force when iterations 9, 19, 29,
    // 39, 49 begin, against the transmitted codewords -- null when not asked for
    const int *v2_cw;       // [B][N] transmitted codewords (symbol indices)
    int *v2_snap;           // [B][5]
    int K;                  // N - M: symbols the bit errors are counted over
};

// Device-side frame source (ems_source.hip): the reference's encoder matrices in CSR.
struct EmsSourcePlan {
    int N, M, K, q, logq;
    int Bs;                 // frame stride of sym / rhs (= the handle's max batch)
    const int *tptr, *tcol; // triangular part of matUT: row m -> entries with m < col < M
    const uint8_t *tcoef;
    const int *rptr, *rcol; // information part: entries with col >= M
    const uint8_t *rcoef;
    const uint8_t *diag;    // [M] matUT[m][m]
    const int *iperm;       // [N] inverse of code_t.Perm: codeword position -> matUT column
    const uint8_t *idx2bin, *bin2idx;
    uint8_t *sym;           // [N][Bs] symbols in matUT column order
    uint8_t *rhs;           // [M][Bs] binary image of the information part's contribution to row m
};
