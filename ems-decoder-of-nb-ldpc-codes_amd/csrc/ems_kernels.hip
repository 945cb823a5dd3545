// ems_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels of the EMS NB-LDPC hot path.
//
// Mapping (wave64 throughout, no MFMA: this is compare/select/merge work, HBM- and VALU-bound):
//   * one WORKGROUP per frame, persistent over the whole decode (all sweeps): frames are independent, so there is no
//     inter-workgroup communication at all; inside a sweep the check nodes are claimed from an LDS counter in a
//     topological order of the layered schedule's dependency DAG (NB_LDPC.c:320 order is preserved per variable node)
//     and a visit waits only for the visits it depends on (done flags in LDS); sweeps are separated by __syncthreads();
//   * k_decode: one WAVEFRONT per check-node visit; the 64 lanes are split into dc "edge rows" of LPE lanes (LPE = 16 for
//     dc <= 4, 8 for dc <= 8, 4 beyond); k_decode_dual (the throughput form, BASELINE message sizes): TWO visits per
//     wavefront, rows of 8 lanes (dc = 4) or 4 lanes (dc = 8).  Each row owns one edge: its q-wide APP row is read from
//     HBM with 16-byte coalesced loads and stays in registers (q/LPE values per lane) from the variable-node extract
//     (NB_LDPC.c:329-337) to the posterior update (NB_LDPC.c:444-450);
//   * the L-bubble elementary check node (bubble_decoder.c:316) runs on groups of 8 lanes = its 8 competitors: a serial
//     8-way merge (min-of-8 by DPP butterflies; the GF(q) addition is an XOR of binary images) records the picks, a
//     parallel post-pass applies the stop rules and the "already in output" test (one ds_max per pick on a q-entry LDS
//     table); up to four (k_decode) or eight (k_decode_dual) elementary steps run side by side in one wave;
//   * nm-entry sorted (LLR, symbol) lists, the two q-byte field tables and a q-wide staging row per edge live in LDS;
//     check-to-variable messages live in HBM in a compact nm*5+9 byte record.
//
// Bit-exactness: only f32 add/sub/compare on the decode path, compiled with -ffp-contract=off, all
// tie-breaks (lowest symbol / lowest competitor) and the 1e5 sentinel comparisons as in the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ems_kernels.h"

#ifndef EMS_Q16_RELANE
#define EMS_Q16_RELANE 0     // 1: the GF(16) two-visit kernel re-derives its lane geometry per phase too (see RELANE in process_cn); measured slower
#endif

namespace {

// Diagnostic build only (-DEMS_STAMPS): per-phase shader-cycle totals of the check-node visits,
// accumulated per wave and added to P.stamps at kernel end.  Never compiled into the product.
#ifdef EMS_STAMPS
#define STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_last; st_last = now_; } while (0)
#define ST_PARAMS , unsigned long long (&st_acc)[16], unsigned long long &st_last
#define ST_ARGS , st_acc, st_last
#define ST_DUMMY unsigned long long st_acc[16] = {}; unsigned long long st_last = 0; (void)st_acc; (void)st_last;
__device__ unsigned long long g_dbg[8];
#ifdef EMS_COUNTERS
#define DBG_ADD(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define DBG_ADD(i, v) do { } while (0)
#endif
#else
#define STAMP(i) do { } while (0)
#define DBG_ADD(i, v) do { } while (0)
#define ST_PARAMS
#define ST_ARGS
#define ST_DUMMY
#endif

// ------------------------------------------------------------------------------------------------
// cross-lane helpers (GFX9 DPP controls)
// ------------------------------------------------------------------------------------------------
constexpr int DPP_XOR1 = 0xB1;     // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;     // quad_perm [2,3,0,1]
constexpr int DPP_HMIRROR = 0x141; // row_half_mirror: lane i <-> 7-i inside each 8
constexpr int DPP_MIRROR = 0x140;  // row_mirror:      lane i <-> 15-i inside each 16

template <int CTRL> __device__ __forceinline__ int dpp_i(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_f(float x)
{
    return __int_as_float(dpp_i<CTRL>(__float_as_int(x)));
}

// all-reduce over aligned groups of 8 lanes (every lane of the group receives the result)
__device__ __forceinline__ float red8_minf(float x)
{
    x = fminf(x, dpp_f<DPP_XOR1>(x));
    x = fminf(x, dpp_f<DPP_XOR2>(x));
    x = fminf(x, dpp_f<DPP_HMIRROR>(x));
    return x;
}
__device__ __forceinline__ int red8_mini(int x)
{
    x = min(x, dpp_i<DPP_XOR1>(x));
    x = min(x, dpp_i<DPP_XOR2>(x));
    x = min(x, dpp_i<DPP_HMIRROR>(x));
    return x;
}
__device__ __forceinline__ int red16_mini(int x)
{
    x = red8_mini(x);
    x = min(x, dpp_i<DPP_MIRROR>(x));
    return x;
}
__device__ __forceinline__ unsigned red8_or(unsigned x)
{
    x |= (unsigned)dpp_i<DPP_XOR1>((int)x);
    x |= (unsigned)dpp_i<DPP_XOR2>((int)x);
    x |= (unsigned)dpp_i<DPP_HMIRROR>((int)x);
    return x;
}

// lexicographic (value, symbol) arg-min over a row of LPE lanes; ties -> lowest symbol
template <int CTRL> __device__ __forceinline__ void argmin_step(float &v, int &g)
{
    const float ov = dpp_f<CTRL>(v);
    const int og = dpp_i<CTRL>(g);
    const bool take = (ov < v) || (ov == v && og < g);
    v = take ? ov : v;
    g = take ? og : g;
}
template <int LPE> __device__ __forceinline__ void row_argmin(float &v, int &g)
{
    argmin_step<DPP_XOR1>(v, g);
    argmin_step<DPP_XOR2>(v, g);
    if (LPE >= 8) argmin_step<DPP_HMIRROR>(v, g);
    if (LPE == 16) argmin_step<DPP_MIRROR>(v, g);
}

// Lanes of one wave exchange data through LDS; DS operations of a wave execute in order, so only the
// compiler has to be kept from reordering.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// GF(q) in the exponent domain: index 0 = zero, index i>=1 = alpha^(i-1)  (init.c:65-130)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int gf_mul_log(int g, int lh, int qm1)   // MULGF[g][lh+1]
{
    int x = g - 1 + lh;
    x = x >= qm1 ? x - qm1 : x;
    return g == 0 ? 0 : x + 1;
}
__device__ __forceinline__ int gf_div_log(int g, int lh, int qm1)   // DIVGF[g][lh+1]
{
    int x = g - 1 - lh;
    x = x < 0 ? x + qm1 : x;
    return g == 0 ? 0 : x + 1;
}

// Every message list in LDS has nm entries followed by two (+inf, invalid) pads, so that a competitor
// walking off its list needs no bounds test: the pad makes its head +inf and it never wins again.
__host__ __device__ inline int list_stride(int nm) { return nm + 2; }

// The two-visits-per-wave kernel (k_decode_dual): its eight q-wide staging rows (8 lanes x 16 bytes per access) all
// start on the same LDS bank, and a 16-byte wave access is served in groups of 16 lanes = pieces of four rows, two of
// which then collide.  Rows 2, 3, 6, 7 therefore keep their 32-dword groups pairwise swapped (dword index ^ 32): the
// four pieces of each group fall on four different quarters of the 64 banks, at no cost in space.
#ifndef EMS_DUAL_NOSWZ
__device__ __forceinline__ int dual_row_swz(int row) { return (row & 2) << 4; }   // 0 or 32 (dwords)
#else
__device__ __forceinline__ int dual_row_swz(int) { return 0; }                    // (A/B build)
#endif

__device__ __forceinline__ unsigned sym_comb(unsigned a, unsigned b)
{
    return ((a ^ b) & 0xFFu) | ((a | b) & EMS_S_INVALID);
}

// ------------------------------------------------------------------------------------------------
// ElementaryStep (bubble_decoder.c:316-593) on one aligned group of 8 lanes.
// lists : the wave's LDS message lists, nm entries each.  a_id/b_id/o_id: Input1, Input2, Output list
// ids; o_id < 0 -> this group has no task in this step (its lanes idle along).
// Competitor p (= lane & 7): p < 4 sits on row p and walks right (bubble_decoder.c:552-555), p >= 4
// starts at (4, p-4) and walks down (:548-551).  Each lane keeps its own head (value + packed symbol)
// and prefetches its next cell, so the only per-step cross-lane traffic is three 8-lane butterflies.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void es_stage(EmsEntry *lists, int *lens, int lane, int a_id, int b_id, int o_id, int l_id,
                                         int nm, int nbOper)
{
    const int p = lane & 7;
    const bool has = o_id >= 0;
    const int nmp = list_stride(nm);
    const EmsEntry *in1 = lists + (has ? a_id : 0) * nmp;
    const EmsEntry *in2 = lists + (has ? b_id : 0) * nmp;
    EmsEntry *out = lists + (has ? o_id : 0) * nmp;

    if (has)
        for (int k = p; k < nmp; k += 8) out[k] = EmsEntry{k < nm ? EMS_SENTINEL : __builtin_inff(), EMS_S_INVALID}; // :370-374

    const bool rowc = p < 4;
    const EmsEntry *mv = rowc ? in2 : in1;
    int pos = rowc ? 0 : 4;
    const EmsEntry fx = rowc ? in1[p] : in2[p - 4];
    const EmsEntry h0 = mv[pos];
    EmsEntry nx = mv[min(pos + 1, nm - 1)];
    float hv = fx.v + h0.v;                                    // tab_aux[r][c] = In1[r] + In2[c]
    unsigned hs = sym_comb(fx.s, h0.s) | (pos >= nm - 1 ? EMS_S_BOUND : 0u);

    unsigned bits = 0; // this lane's 32-symbol slice of GFvalues_already_in_Out (:334)
    int s = 0;
    bool act = has;
    for (int ss = 0; ss < nbOper; ++ss) {                      // :472
        if (__ballot(act) == 0ull) break;
        // minimum() (:38-56): first strictly smallest head below 1e5, else competitor 0
        const float m = red8_minf(hv);
        const bool vmin = m < EMS_SENTINEL;
        int wp = red8_mini((hv == m) ? p : 8);
        wp = vmin ? wp : 0;
        const bool iw = (p == wp);
        const unsigned wh = red8_or(iw ? hs : 0u);
        const float wv = vmin ? m : hv;                        // read on lane p == 0 only
        const unsigned sym = wh & 0xFFu;
        const bool inv = (wh & EMS_S_INVALID) != 0;            // :478-484 source index -1 -> break
        const bool bnd = (wh & EMS_S_BOUND) != 0;              // :506 row/col >= nm-1 -> break after
        const bool mine = (sym >> 5) == (unsigned)p;
        const unsigned bit = 1u << (sym & 31u);
        const bool dup = red8_or((mine && (bits & bit)) ? 1u : 0u) != 0; // :490
        const bool ap = act && !inv && !dup;
        if (ap && mine) bits |= bit;
        if (ap && p == 0) out[s] = EmsEntry{wv, sym};          // :492-494
        s += ap ? 1 : 0;
        const bool stop = inv || (s == nm) || bnd;             // :484, :502, :506
        const bool adv = act && !stop && iw;
        act = act && !stop;
        if (adv) {                                             // :548-557
            pos += 1;
            hv = fx.v + nx.v;
            hs = sym_comb(fx.s, nx.s) | (pos >= nm - 1 ? EMS_S_BOUND : 0u);
            nx = mv[min(pos + 1, nm - 1)];
        }
    }
    if (has && p == 0) lens[l_id] = s;   // l_id: the output's slot in lens (= o_id except in the two-visit layout)
}


// ------------------------------------------------------------------------------------------------
// ElementaryStep, fast form.  The bubble loop of bubble_decoder.c:472-562 is a stable 8-way merge whose
// pick sequence does not depend on the output bookkeeping, so it is split in two:
//   1. the serial loop only merges: per step one 8-lane min, the winner records (competitor, position)
//      as one byte and advances -- 10 VALU + 2 LDS instructions (es_merge);
//   2. a parallel post-pass over the T recorded picks (16 lanes per task, es_post) recomputes value +
//      symbol of every pick, finds the first stop (invalid source index -> before the pick,
//      bubble_decoder.c:478; last row/column -> after it, :506), removes symbols already emitted (first
//      occurrence wins, :490; one ds_max per pick on a q-entry LDS table) and compacts the survivors in
//      order, cutting at nm (:502).
// The loop first runs nm+3 steps (the reference needs 20.3 on average at nm = 20); a task with neither nm
// outputs nor a stop by then gets 8 more steps, post-processed incrementally, and only then the full nbOper.
// Values inside the check node are non-negative floats (normalised inputs and their sums), so the
// IEEE order equals the unsigned order of the bit patterns; the loop compares bit patterns.
// ------------------------------------------------------------------------------------------------
// (bit pattern of 1e5f is 0x47C35000)
constexpr int ES_MAX_STEPS = 64;
constexpr int ES_PICK_BYTES = 8 * (ES_MAX_STEPS + 4) + 64 * 4 + ES_MAX_STEPS;   // 8 pick areas + one dummy dword per lane (+ step offset)

__device__ __forceinline__ unsigned red8_minu(unsigned x)
{
    x = min(x, (unsigned)dpp_i<DPP_XOR1>((int)x));
    x = min(x, (unsigned)dpp_i<DPP_XOR2>((int)x));
    x = min(x, (unsigned)dpp_i<DPP_HMIRROR>((int)x));
    return x;
}

// Register-light form of es_post for the rare continuation (T up to 64 picks): same algorithm, the
// per-pick value/symbol are recomputed in each of the three passes instead of being kept in registers.
// (p, gbase: lane within / first lane of the task's 16 lanes.)
// min over the TL (8 or 16) lanes of a task
template <int TL> __device__ __forceinline__ int redT_mini(int x)
{
    x = red8_mini(x);
    if (TL == 16) x = min(x, dpp_i<DPP_MIRROR>(x));
    return x;
}

// TL: lanes per task (16 in the one-visit-per-wave kernels, 8 where a wave carries two visits)
template <int FL = 0, int TL = 16>
__device__ __forceinline__ bool es_post_rt(const EmsEntry *in1, const EmsEntry *in2, EmsEntry *out, unsigned *tbl,
                                           const uint8_t *picks, int q, int p, int gbase, bool has, int T, int nm,
                                           int nbOper, int &S_out)
{
    constexpr unsigned TMASK = (1u << TL) - 1u;
    const int NJ = (T + TL - 1) / TL;
    auto gather = [&](int i, bool ok, float &v, unsigned &sy, int &k) {
        const unsigned pk = picks[ok ? i : 0];
        const int w = pk >> 5;
        k = pk & 31;
        constexpr int NFIX = FL ? 2 : 4;                 // row walkers (the others walk down a column)
        const int r = w < NFIX ? w : k, c = w < NFIX ? k : w - NFIX;
        const EmsEntry e1 = in1[r], e2 = in2[c];
        v = e1.v + e2.v;
        sy = sym_comb(e1.s, e2.s);
    };
    wave_lds_sync();
    int lim = 1 << 20;
    for (int j = 0; j < NJ; ++j) {
        const int i = p + TL * j;
        const bool ok = has && i < T;
        float v; unsigned sy; int k;
        gather(i, ok, v, sy, k);
        const int li = !ok ? (1 << 20) : ((sy & EMS_S_INVALID) ? i : (k >= nm - 1 ? i + 1 : (1 << 20)));
        lim = min(lim, li);
    }
    lim = redT_mini<TL>(lim);
    const bool stopped = lim <= T;
    lim = min(lim, T);
    wave_lds_sync();
    for (int j = 0; j < NJ; ++j) {
        const int i = p + TL * j;
        float v; unsigned sy; int k;
        gather(i, has && i < T, v, sy, k);
        if (has && i < lim) atomicMax(&tbl[sy & 0xFFu], (unsigned)(ES_MAX_STEPS - i));
    }
    wave_lds_sync();
    int base = 0;
    for (int j = 0; j < NJ; ++j) {
        const int i = p + TL * j;
        float v; unsigned sy; int k;
        gather(i, has && i < T, v, sy, k);
        const bool u = has && i < lim && tbl[sy & 0xFFu] == (unsigned)(ES_MAX_STEPS - i);
        const unsigned m8 = (unsigned)(__ballot(u) >> gbase) & TMASK;
        const int sidx = base + __popc(m8 & ((1u << p) - 1u));
        if (u && sidx < nm) out[sidx] = EmsEntry{v, sy & 0xFFu};
        base += __popc(m8);
    }
    wave_lds_sync();
    for (int j = 0; j < NJ; ++j) {
        const int i = p + TL * j;
        float v; unsigned sy; int k;
        gather(i, has && i < T, v, sy, k);
        if (has && i < lim) tbl[sy & 0xFFu] = 0u;
    }
    S_out = min(base, nm);
    return (base >= nm) || stopped || (T >= nbOper);
}

__device__ __forceinline__ int red8_sumi(int x)
{
    x += dpp_i<DPP_XOR1>(x);
    x += dpp_i<DPP_XOR2>(x);
    x += dpp_i<DPP_HMIRROR>(x);
    return x;
}

// Merge loops.  A single wave executes a dependent instruction stream at ~7 cycles per instruction, and
// the visit latency -- not throughput -- bounds the decoder (the layered schedule is a 441-deep chain per
// sweep), so the loop is pared down to the bone:
//   SPECULATIVE form (12 VALU/LDS instructions per step): the winner of a step is every lane whose head
//   equals the group minimum -- exactly one lane unless two heads tie.  A tie makes two lanes advance, so
//   it shows up as (sum of advances over the group) != (number of steps); the stage is then redone in the
//   EXACT form, where the winner is the lowest competitor among the equal heads (bubble_decoder.c:47).
// Both rely on the caller's guard: every valid value in the visit is in [0, 1e4), hence (a) float order ==
// unsigned order of the bit patterns and (b) a head >= 1e5 always involves an invalid (padded) entry, so
// whichever such head is taken the step is the "source index -1 -> break" stop of bubble_decoder.c:478.
// A pick is recorded as one byte, competitor p in bits 5..7 and its list position in bits 0..4; the loop
// carries y = 32 p + pos so that the byte is y itself and the list address is one shift-add away.
// FL = 1: the v2 flavour's comparator (v2.c:1705-1720 on 4 competitors = aligned groups of 4 lanes): first strict
// maximum above 0.0, competitor 0 when none.  For non-NaN floats "x > 0.0f" <=> "bits as signed int > 0" and positive
// floats order like their bit patterns, so the comparison runs on signed integers (any negative / zero value simply
// never wins); p = lane & 3, gbase = lane & ~3.
template <bool EXACT, int FL = 0>
__device__ __forceinline__ void es_merge(unsigned &hv, int &y, float fxv, const EmsEntry *mvy, uint8_t *pk,
                                         uint8_t *dummy, int p, int gbase, int from, int to)
{
    // The next-head operand mv[pos+1] is fetched two steps ahead so that the LDS latency (~110 cycles, more
    // than one step) stays off the loop-carried path: each step issues the pair (mv[pos+1], mv[pos+2]) for
    // its updated position; the step after next picks the element of that pair its predecessor's outcome
    // (advance or not) makes current.  Two pairs alternate, each reloaded by the step that consumed it.
    // mvy = mv - 32 p entries, i.e. mvy[y] == mv[pos].
    // The list address is recomputed from y with one shift-add per step (opaque to the optimiser, which would
    // otherwise carry y and the address as two separate induction variables at one more instruction per step).
    typedef __attribute__((address_space(3))) const float lds_cf32;
    const unsigned mvb = (unsigned)(uintptr_t)(lds_cf32 *)&mvy[0].v;
    auto at = [&](int yy) { unsigned a; asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(a) : "v"(yy), "v"(mvb)); return (lds_cf32 *)(uintptr_t)a; };
    lds_cf32 *q0 = at(y);
    float a0 = q0[2], a1 = q0[4];
    float b0 = a0, b1 = a1;
    bool wprev = false;
    auto step = [&](float &x0, float &x1, int ss) {
        bool iw;
        if (FL == 0) {
            const unsigned m = red8_minu(hv);
            iw = (hv == m);
            if (EXACT) iw = (p == __ffs((unsigned)(__ballot(iw) >> gbase)) - 1);
        } else {
            int m = (int)hv;
            m = max(m, dpp_i<DPP_XOR1>(m));
            m = max(m, dpp_i<DPP_XOR2>(m));
            iw = ((int)hv == m);
            if (EXACT) iw = (p == __ffs((unsigned)(__ballot(iw) >> gbase) & 0xFu) - 1);
            iw = (m > 0) ? iw : (p == 0);
        }
        const float nx = wprev ? x1 : x0;
        (iw ? pk : dummy)[ss] = (uint8_t)y;      // losers write to their own dummy dword (+ss: still one dword per lane)
        y += iw ? 1 : 0;
        asm("" : "+v"(y));                    // one add-with-carry per step (no re-association across steps)
        hv = iw ? (unsigned)__float_as_int(fxv + nx) : hv;
        lds_cf32 *qn = at(y);
        x0 = qn[2];
        x1 = qn[4];
        wprev = iw;
    };
    int ss = from;
#pragma unroll
    for (; ss + 1 < to; ss += 2) {
        step(a0, a1, ss);
        step(b0, b1, ss + 1);
    }
    if (ss < to) step(a0, a1, ss);
}

// Post-pass of the first T picks of a stage by the 16 lanes of its task (T <= 16 NJ; p, gbase: lane within / first
// lane of the 16; p8, gbase8: the same for the 8-lane merge group): stop rules, first-occurrence dedupe through the
// symbol table, ordered compaction into `out`.  When some group has neither nm outputs nor a stop yet (about
// one stage in ten at nm = 20: duplicates eat picks), the merge runs 8 more steps with the exact rule and the
// new picks go through the same table incrementally -- its first-pass claims are still in place, and a later
// pick never displaces an earlier one (ds_max of ES_MAX_STEPS - i).  tbl is all-zero on entry and on exit.
template <int NJ, int FL = 0, int TL = 16>
__device__ __forceinline__ bool es_post(const EmsEntry *in1, const EmsEntry *in2, EmsEntry *out, unsigned *tbl,
                                        uint8_t *picks, uint8_t *dummy, int q, int p, int gbase, int p8, int gbase8, bool has, int T, int nm,
                                        int nbOper, int &S_out, unsigned &hv, int &y, float fxv, const EmsEntry *mvy, int &T_done)
{
    wave_lds_sync();
    float val[NJ];
    unsigned sym[NJ];
    int lim = 1 << 20;
    auto gather = [&](int i, bool ok, float &v, unsigned &sy, int &li) {
        const unsigned pk = picks[ok ? i : 0];
        const int w = pk >> 5, k = pk & 31;
        constexpr int NFIX = FL ? 2 : 4;
        const int r = w < NFIX ? w : k, c = w < NFIX ? k : w - NFIX;
        const EmsEntry e1 = in1[r], e2 = in2[c];
        v = e1.v + e2.v;
        sy = sym_comb(e1.s, e2.s);
        li = !ok ? (1 << 20) : ((sy & EMS_S_INVALID) ? i : (k >= nm - 1 ? i + 1 : (1 << 20)));
        sy &= 0xFFu;
    };
    {   // the NJ picks of this lane, phase by phase so that each phase is ONE LDS round trip: pick bytes, then the
        // two source entries of every pick, then the arithmetic (same operations as gather() above)
        unsigned pkb[NJ];
        EmsEntry e1[NJ], e2[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) pkb[j] = picks[(has && p + TL * j < T) ? p + TL * j : 0];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int w = pkb[j] >> 5, k = pkb[j] & 31;
            constexpr int NFIX = FL ? 2 : 4;
            e1[j] = in1[w < NFIX ? w : k];
            e2[j] = in2[w < NFIX ? k : w - NFIX];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int i = p + TL * j, k = pkb[j] & 31;
            const bool ok = has && i < T;
            val[j] = e1[j].v + e2[j].v;
            const unsigned sy = sym_comb(e1[j].s, e2[j].s);
            const int li = !ok ? (1 << 20) : ((sy & EMS_S_INVALID) ? i : (k >= nm - 1 ? i + 1 : (1 << 20)));
            sym[j] = sy & 0xFFu;
            lim = min(lim, li);
        }
    }
    if (__ballot(lim < (1 << 20)) != 0ull) lim = redT_mini<TL>(lim);   // (no lane of the wave saw a stop: nothing to reduce)
    bool stopped = lim <= T;
    lim = min(lim, T);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = p + TL * j;
        if (has && i < lim) atomicMax(&tbl[sym[j]], (unsigned)(ES_MAX_STEPS - i));
    }
    wave_lds_sync();
    int base = 0;
    unsigned tv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) tv[j] = tbl[sym[j]];          // (all read-backs in flight together: one round trip)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = p + TL * j;
        const bool u = has && i < lim && tv[j] == (unsigned)(ES_MAX_STEPS - i);
        const unsigned m8 = (unsigned)(__ballot(u) >> gbase) & ((1u << TL) - 1u);
        const int sidx = base + __popc(m8 & ((1u << p) - 1u));
        if (u && sidx < nm) out[sidx] = EmsEntry{val[j], sym[j]};
        base += __popc(m8);
    }
    bool done = (base >= nm) || stopped || (T >= nbOper);
    T_done = T;
    bool clrx = false;
    unsigned symx = 0;
    if (__ballot(has && !done) != 0ull) {
        DBG_ADD(4, 1);
        const int T2 = min(T + 8, nbOper);
        es_merge<true, FL>(hv, y, fxv, mvy, picks, dummy, p8, gbase8, T, T2);
        T_done = T2;
        wave_lds_sync();
        const int i = T + p;
        const bool ok = has && !done && p < 8 && i < T2;
        float v;
        int li;
        gather(i, ok, v, symx, li);
        int lim2 = redT_mini<TL>(li);
        const bool stopped2 = lim2 <= T2;
        lim2 = min(lim2, T2);
        clrx = ok && i < lim2;
        if (clrx) atomicMax(&tbl[symx], (unsigned)(ES_MAX_STEPS - i));
        wave_lds_sync();
        const bool u = clrx && tbl[symx] == (unsigned)(ES_MAX_STEPS - i);
        const unsigned m8 = (unsigned)(__ballot(u) >> gbase) & ((1u << TL) - 1u);
        const int sidx = base + __popc(m8 & ((1u << p) - 1u));
        if (u && sidx < nm) out[sidx] = EmsEntry{v, symx};
        if (!done) {
            base += __popc(m8);
            done = (base >= nm) || stopped2 || (T2 >= nbOper);
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (has && p + TL * j < lim) tbl[sym[j]] = 0u;
    if (clrx) tbl[symx] = 0u;
    S_out = min(base, nm);
    return done;
}

template <int TL = 16>
__device__ __forceinline__ void es_stage_fast(EmsEntry *lists, int *lens, unsigned *tbl_base, uint8_t *picks_base, int q,
                                              int lane, uint2 task, int nm, int nbOper, int T1 ST_PARAMS)
{
    // A task owns TL lanes.  The merge needs 8 (its 8 competitors): with TL = 16 both halves run it identically (same
    // inputs, same picks written to the same bytes); the post-pass then spreads the picks over all TL lanes.
    const int p = lane & 7;                         // competitor
    const int gbase = lane & ~7;
    const int p16 = lane & (TL - 1);                // post-pass lane
    const int gbase16 = lane & ~(TL - 1);
    const int gsel = lane / TL;
    const bool has = (task.y >> 16) != 0u;          // see es_task(): byte offsets of the lists, output id + 1
    const int o_id = (int)(task.y >> 16) - 1;
    uint8_t *lb = reinterpret_cast<uint8_t *>(lists);
    const EmsEntry *in1 = reinterpret_cast<const EmsEntry *>(lb + (task.x & 0xFFFFu));
    const EmsEntry *in2 = reinterpret_cast<const EmsEntry *>(lb + (task.x >> 16));
    EmsEntry *out = reinterpret_cast<EmsEntry *>(lb + (task.y & 0xFFFFu));
    unsigned *tbl = tbl_base + gsel * q;
    uint8_t *picks = picks_base + gsel * (ES_MAX_STEPS + 4);            // pick areas 17 dwords apart (bank spread); lanes without a task scribble in theirs
    uint8_t *dummy = picks_base + 8 * (ES_MAX_STEPS + 4) + lane * 4;  // one dword per lane: no same-dword byte writes

    // (Output = (1e5, -1) everywhere, bubble_decoder.c:370-374: only the tail behind the S entries emitted can show through,
    //  and S = nm nearly always -- the tail is written at the end, when there is one, and so are the two pads.)

    const bool rowc = p < 4;
    const EmsEntry *mv = rowc ? in2 : in1;
    const EmsEntry *mvy = mv - 32 * p;
    const int pos0 = (rowc ? 0 : 4) + 32 * p;
    const float fxv = rowc ? in1[p].v : in2[p - 4].v;
    unsigned hv;
    int pos;
    auto init = [&]() {
        pos = pos0;
        hv = (unsigned)__float_as_int(fxv + mvy[pos0].v);        // tab_aux[r][c] = In1[r] + In2[c]
    };

    int S = 0;
    bool done;
    init();
    STAMP(8);
    es_merge<false>(hv, pos, fxv, mvy, picks, dummy, p, gbase, 0, T1);
    STAMP(9);
    DBG_ADD(3, 1);
#if defined(EMS_ABL) && EMS_ABL == 8      // ablation (results wrong where it matters): no tie redo, no long continuation, no selection fallback
    if (false) {
#else
    if (__ballot(has && red8_sumi(pos - pos0) != T1) != 0ull) {   // a tie somewhere: redo with the exact rule
#endif
        DBG_ADD(5, 1);
        init();
        es_merge<true>(hv, pos, fxv, mvy, picks, dummy, p, gbase, 0, T1);
    }
    STAMP(10);
    int Td = T1;   // picks merged so far
    if (T1 <= TL) done = es_post<1, 0, TL>(in1, in2, out, tbl, picks, dummy, q, p16, gbase16, p, gbase, has, T1, nm, nbOper, S, hv, pos, fxv, mvy, Td);
    else if (T1 <= 2 * TL) done = es_post<2, 0, TL>(in1, in2, out, tbl, picks, dummy, q, p16, gbase16, p, gbase, has, T1, nm, nbOper, S, hv, pos, fxv, mvy, Td);
    else if (TL == 8 && T1 <= 24) done = es_post<3, 0, TL>(in1, in2, out, tbl, picks, dummy, q, p16, gbase16, p, gbase, has, T1, nm, nbOper, S, hv, pos, fxv, mvy, Td);
    else if (TL == 8 && T1 <= 32) done = es_post<4, 0, TL>(in1, in2, out, tbl, picks, dummy, q, p16, gbase16, p, gbase, has, T1, nm, nbOper, S, hv, pos, fxv, mvy, Td);
    else done = es_post_rt<0, TL>(in1, in2, out, tbl, picks, q, p16, gbase16, has, T1, nm, nbOper, S);
#if defined(EMS_ABL) && EMS_ABL == 8
    if (false) {
#else
    if (__ballot(has && !done) != 0ull) {
#endif
        // rare: neither nm outputs nor a stop within the first nm+3 (+8) picks -> merge on up to nbOper
        DBG_ADD(6, 1);
        es_merge<true>(hv, pos, fxv, mvy, picks, dummy, p, gbase, Td, nbOper);
        es_post_rt<0, TL>(in1, in2, out, tbl, picks, q, p16, gbase16, has, nbOper, nm, nbOper, S);
    }
    if (__ballot(has && S < nm) != 0ull) {                        // a short output (rare): the initial fill behind it
        if (has)
            for (int k = S + p16; k < nm; k += TL) out[k] = EmsEntry{EMS_SENTINEL, EMS_S_INVALID};
    }
    if (has && p16 < 2) out[nm + p16] = EmsEntry{__builtin_inff(), EMS_S_INVALID};   // list pads (scratch of other phases may have been here)
    STAMP(11);
    if (has && p16 == 0) lens[o_id] = S;
}

// ------------------------------------------------------------------------------------------------
// ElementaryStep over GF(16) with nm = 8 (BASELINE configs[2]), "first occurrence per symbol" form.
//
// The bubble loop of bubble_decoder.c:472-562 walks the 8 nm - 16 = 48 cells of the L-shaped region in the order of
// (value, competitor, position) -- every competitor's list is non-decreasing, so the 8-way merge is that sort -- and its output is
// the first nm DISTINCT symbols met, each with the value of its first occurrence, cut by the stops.  With 16 symbols the
// scan can be replaced by one LDS ds_min per cell into a 16-entry table (bin s = smallest value of a cell with symbol s =
// the first occurrence of s), a rank of the 16 bins, and three closed-form tests for the stops:
//   * a source index of -1 (:478): such a cell's value is >= 1e5 under the visit's guard, it sorts behind every valid cell, so
//     cells >= 1e5 simply do not exist;
//   * last row / column (:506): the scan ends after the cheapest boundary cell (row 7 or column 7), value K; a symbol is
//     emitted iff its bin is <= K;
//   * nm outputs (:502): the first 8 bins in rank order;
//   * nbOper (:472): the scan up to the last emitted entry (or K) visits N = #{cells <= that value} cells; if N > nbOper the
//     reference stops on its nbOper-th cell (0.35 % of the tasks at NbOper = 16) -- what it has emitted by then is the prefix of
//     the rank order whose entries have at most nbOper cells up to and including their own (block at the end of the function);
// a task where the order of two equal values could matter is redone by the merge form: two bins with the same value (their ranks then do not
// add up), or a second cell with the value K when the scan gets that far.  Equal values with the same symbol, and ties among
// cells that are not first occurrences, change nothing.  Values are compared as floats by the sign of their difference
// (non-negative, no NaN among valid ones) or as bit patterns.
// TL = 8 lanes per task (lane = competitor: its fixed entry and the 8 cells of its row, or the 4 of its column) or 16 (three
// cells per lane, one bin per lane).  bins: 16 words per task, all +inf on entry and on exit.  Returns true on the lanes of a
// task that must be redone by es_stage_fast.
// ------------------------------------------------------------------------------------------------
constexpr unsigned Q16_INF = 0x7F800000u;
constexpr unsigned Q16_LIM = 0x47C34FFFu;                  // largest float below 1e5f (0x47C35000)

template <int TL> __device__ __forceinline__ int redT_sumi(int x)
{
    x = red8_sumi(x);
    if (TL == 16) x += dpp_i<DPP_MIRROR>(x);
    return x;
}
template <int TL> __device__ __forceinline__ unsigned redT_minu(unsigned x)
{
    x = red8_minu(x);
    if (TL == 16) x = min(x, (unsigned)dpp_i<DPP_MIRROR>((int)x));
    return x;
}
template <int TL> __device__ __forceinline__ unsigned redT_maxu(unsigned x)
{
    x = max(x, (unsigned)dpp_i<DPP_XOR1>((int)x));
    x = max(x, (unsigned)dpp_i<DPP_XOR2>((int)x));
    x = max(x, (unsigned)dpp_i<DPP_HMIRROR>((int)x));
    if (TL == 16) x = max(x, (unsigned)dpp_i<DPP_MIRROR>((int)x));
    return x;
}
// number of elements of x[0..N) that are > T (floats, T >= 0 finite, x may hold +inf): sign bits of T - x
template <int N> __device__ __forceinline__ int count_above(const float (&x)[N], float T)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f t2 = {T, T};
    unsigned a = 0u;
    int e = 0;
#pragma unroll
    for (; e + 1 < N; e += 2) {                            // two differences per v_pk_add_f32
        const v2f d = t2 - v2f{x[e], x[e + 1]};
        a = __builtin_amdgcn_alignbit(a, __float_as_uint(d.x), 31);
        a = __builtin_amdgcn_alignbit(a, __float_as_uint(d.y), 31);
    }
    if (e < N) a = __builtin_amdgcn_alignbit(a, __float_as_uint(T - x[e]), 31);
    return __popc(a);
}

// number of elements of x[0..N) that are < T (T finite; +inf elements never count): sign bits of x - T
template <int N> __device__ __forceinline__ int count_below(const float (&x)[N], float T)
{
    unsigned a = 0u;
#pragma unroll
    for (int e = 0; e < N; ++e) a = __builtin_amdgcn_alignbit(a, __float_as_uint(x[e] - T), 31);
    return __popc(a);
}

template <int TL>
__device__ __forceinline__ bool es_stage_q16(EmsEntry *lists, int *lens, unsigned *bins_base, int lane, uint2 task, int nbOper)
{
    constexpr int NS = TL == 8 ? 8 : 3;                    // cells per lane
    constexpr int NB = 16 / TL;                            // bins per lane
    const int p = lane & (TL - 1);
    const int gsel = lane / TL;
    const bool has = (task.y >> 16) != 0u;
    const int o_id = (int)(task.y >> 16) - 1;
    uint8_t *lb = reinterpret_cast<uint8_t *>(lists);
    const EmsEntry *in1 = reinterpret_cast<const EmsEntry *>(lb + (task.x & 0xFFFFu));
    const EmsEntry *in2 = reinterpret_cast<const EmsEntry *>(lb + (task.x >> 16));
    EmsEntry *out = reinterpret_cast<EmsEntry *>(lb + (task.y & 0xFFFFu));
    unsigned *bins = bins_base + gsel * 16;

    float v[NS];
    unsigned sy[NS];
    float bnd;                                             // this lane's boundary cell (row 7 / column 7), +inf when it has none
    if constexpr (TL == 8) {
        // competitor p < 4: row p, cells (p, 0..7); p >= 4: column p - 4, cells (4..7, p - 4) in slots 4..7 -- slot 7 is the boundary cell
        const bool rowc = p < 4;
        const EmsEntry fx = rowc ? in1[p] : in2[p - 4];
        const uint4 *mv4 = static_cast<const uint4 *>(__builtin_assume_aligned(rowc ? in2 : in1, 16));   // (lists are 80 bytes apart from a 16-byte-aligned base)
        uint4 e01 = uint4{Q16_INF, 0u, Q16_INF, 0u}, e23 = e01;
        if (rowc) { e01 = mv4[0]; e23 = mv4[1]; }          // (a column competitor needs rows 4..7 only: half the lanes sit two reads out, +1.8 %)
        const uint4 e45 = mv4[2], e67 = mv4[3];
        const unsigned ev[8] = {e01.x, e01.z, e23.x, e23.z, e45.x, e45.z, e67.x, e67.z};
        const unsigned es[8] = {e01.y, e01.w, e23.y, e23.w, e45.y, e45.w, e67.y, e67.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = fx.v + __uint_as_float(ev[j]);          // tab_aux[r][c] = In1[r] + In2[c] (:386-423)
            sy[j] = (fx.s ^ es[j]) & 15u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rowc ? v[j] : __builtin_inff();   // a column competitor starts on row 4
        bnd = v[7];
    } else {
        // cell i = p + 16 k: k = 0, 1 -> (row (p >> 3) + 2 k, column p & 7); k = 2 -> (row 4 + (p >> 2), column p & 3)
        const int r0 = p >> 3, c0 = p & 7, r2 = 4 + (p >> 2), c2 = p & 3;
        const EmsEntry a0 = in1[r0], a1 = in1[r0 + 2], b0 = in2[c0], a2 = in1[r2], b2 = in2[c2];
        v[0] = a0.v + b0.v; sy[0] = (a0.s ^ b0.s) & 15u;
        v[1] = a1.v + b0.v; sy[1] = (a1.s ^ b0.s) & 15u;
        v[2] = a2.v + b2.v; sy[2] = (a2.s ^ b2.s) & 15u;
        bnd = c0 == 7 ? fminf(v[0], v[1]) : __builtin_inff();
        bnd = r2 == 7 ? fminf(bnd, v[2]) : bnd;
    }
    // (cells >= 1e5 land above every valid value: harmless.  A column competitor has no cell in slots 0..3: its lanes sit those four
    //  atomics out -- the LDS pipe is as busy as the VALUs in this kernel, and a quarter fewer lane-atomics was worth 3.4 %)
    if constexpr (TL == 8) {
        const bool rowc_ = p < 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (rowc_) atomicMin(&bins[sy[j]], __float_as_uint(v[j]));
#pragma unroll
        for (int j = 4; j < 8; ++j) atomicMin(&bins[sy[j]], __float_as_uint(v[j]));
    } else {
#pragma unroll
        for (int j = 0; j < NS; ++j) atomicMin(&bins[sy[j]], __float_as_uint(v[j]));
    }
    const unsigned kstop = redT_minu<TL>(__float_as_uint(bnd));                   // :506, value of the cell that ends the scan
    const unsigned klim = min(kstop, Q16_LIM);
    wave_lds_sync();

    // ---- rank of the 16 first occurrences ----
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f *b4 = static_cast<const v4f *>(__builtin_assume_aligned(bins, 64));
    unsigned mine[NB];
    if constexpr (NB == 2) {
        const uint2 m2 = *reinterpret_cast<const uint2 *>(bins + 2 * p);
        mine[0] = m2.x; mine[1] = m2.y;
    } else {
        mine[0] = bins[p];
    }
    unsigned acc[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) acc[k] = 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {                          // four bins per LDS read, two differences per v_pk_add_f32
        const v4f q = b4[c];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const float me = __uint_as_float(mine[k]);
            const v2f m2 = {me, me};
            const v2f dl = v2f{q.x, q.y} - m2, dh = v2f{q.z, q.w} - m2;                 // sign: that bin < mine
            acc[k] = __builtin_amdgcn_alignbit(acc[k], __float_as_uint(dl.x), 31);
            acc[k] = __builtin_amdgcn_alignbit(acc[k], __float_as_uint(dl.y), 31);
            acc[k] = __builtin_amdgcn_alignbit(acc[k], __float_as_uint(dh.x), 31);
            acc[k] = __builtin_amdgcn_alignbit(acc[k], __float_as_uint(dh.y), 31);
        }
    }
    // Five small counts travel through ONE butterfly: valid bins (bits 0-4), the sum of their ranks (5-12), valid bins whose value is
    // the one that ends the scan (13-17), boundary cells with that value (23-27) and, of those, the ones that are the first
    // occurrence of their symbol (18-22).
    int rank[NB];
    bool valid[NB];
    int pack = 0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        rank[k] = __popc(acc[k]);
        valid[k] = mine[k] <= klim;
        pack += valid[k] ? 1 + (rank[k] << 5) : 0;
        if constexpr (TL == 8) pack += (valid[k] && mine[k] == kstop) ? (1 << 13) : 0;
    }
    if constexpr (TL == 8) {
        const bool is_b = __float_as_uint(bnd) == kstop && kstop <= Q16_LIM;
        pack += is_b ? (1 << 23) + (bins[sy[7]] == kstop ? (1 << 18) : 0) : 0;
    }
    pack = redT_sumi<TL>(pack);
    const int D = pack & 31, rsum = (pack >> 5) & 255;
    bool bad = 2 * rsum != D * (D - 1);                    // two first occurrences with the same value: their order is the competitors'
    int S = min(D, 8);
    unsigned x7 = 0u;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        if (has && valid[k] && rank[k] < 8) out[rank[k]] = EmsEntry{__uint_as_float(mine[k]), (unsigned)(NB * p + k)};   // :492-494
        x7 = (valid[k] && rank[k] == 7) ? mine[k] : x7;
    }
    // ---- the stops that depend on how far the scan got ----
    const unsigned xcut = D >= 8 ? redT_maxu<TL>(x7) : klim;                      // value of the last cell the scan visits
    int cnt = NS - count_above(v, __uint_as_float(xcut));                        // cells visited until then (bits 0-7)
    if constexpr (TL == 16) {                                                    // ... and cells with the value that ends the scan (8-15)
#pragma unroll
        for (int j = 0; j < NS; ++j) cnt += (__float_as_uint(v[j]) == kstop) ? 256 : 0;
    }
    cnt = redT_sumi<TL>(cnt);
    const bool ovf = (cnt & 255) > nbOper;                                       // :472 cuts the scan short (handled below)
    // Another cell with the value K that ends the scan, when the scan gets that far: the order of the two would matter.  Rows of 16
    // lanes count the cells with that value.  Rows of 8 read it off the bins: a tie matters only when the other cell is the first
    // occurrence of ANOTHER symbol (a repeated symbol changes nothing, the same symbol gives the same entry either way) -- then its
    // bin holds K; if the boundary cell is a first occurrence too, two bins hold K and the rank sum above has already failed, so what
    // is left is "some valid bin holds K and no boundary cell at K is a first occurrence", plus two boundary cells at K.
    if constexpr (TL == 16) bad = bad || (kstop <= xcut && (cnt >> 8) > 1);
    else bad = bad || (kstop <= xcut && (((pack >> 23) & 31) > 1 || (((pack >> 13) & 31) >= 1 && ((pack >> 18) & 31) == 0)));
    // ---- the scan cut by NbOper (:472; 0.35 % of the tasks, one wave-stage in fifty): it ends on its NbOper-th cell, before any other
    //      stop, so what it emits are the first occurrences with at most NbOper cells up to and including their own -- a prefix of the
    //      rank order.  Thresholds = the entries just stored; per threshold one count over the task's cells, eight counts through
    //      two butterflies.  A cell with the same value as the first entry left out could have come before or after the cut when
    //      fewer than NbOper cells are strictly below it: that task goes to the merge form. ----
    if (__builtin_expect(__ballot(has && ovf && !bad) != 0ull, 0)) {
        wave_lds_sync();
        const uint4 *o4 = static_cast<const uint4 *>(__builtin_assume_aligned(out, 16));
        const uint4 t01 = o4[0], t23 = o4[1], t45 = o4[2], t67 = o4[3];
        const unsigned thr[8] = {t01.x, t01.z, t23.x, t23.z, t45.x, t45.z, t67.x, t67.z};
        int lo = 0, hi = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            lo += (NS - count_above(v, __uint_as_float(thr[r]))) << (8 * r);
            hi += (NS - count_above(v, __uint_as_float(thr[4 + r]))) << (8 * r);
        }
        lo = redT_sumi<TL>(lo);
        hi = redT_sumi<TL>(hi);
        int S2 = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) S2 += (r < S && (((r < 4 ? lo : hi) >> (8 * (r & 3))) & 255) <= nbOper) ? 1 : 0;
        unsigned tS = thr[7];
#pragma unroll
        for (int r = 6; r >= 0; --r) tS = (S2 == r) ? thr[r] : tS;
        const int below = redT_sumi<TL>(count_below(v, __uint_as_float(tS)));
        if (ovf && !bad) {
            bad = S2 < S && below < nbOper;
            S = S2;
        }
    } else {
        bad = bad || ovf;
    }
    // ---- tail, pads, length; bins back to +inf ----
    if (has) {
        if (p < 8 && p >= S) out[p] = EmsEntry{EMS_SENTINEL, EMS_S_INVALID};     // :370-374 shows through behind the S entries
        if (p < 2) out[8 + p] = EmsEntry{__builtin_inff(), EMS_S_INVALID};
        if (p == 0) lens[o_id] = S;
    }
    if constexpr (NB == 2) *reinterpret_cast<uint2 *>(bins + 2 * p) = uint2{Q16_INF, Q16_INF};
    else bins[p] = Q16_INF;
#if defined(EMS_ABLQ) && EMS_ABLQ == 2     // (ablation: no detection either -- the tests above are dead code then)
    return false;
#endif
    return has && bad;
}

// Decoder parameters as the device functions see them.  In the specialised kernel instantiations
// (template DC, NM > 0) dc, nm and the first-pass length T1 are compile-time constants after inlining, so
// the small loops over list entries unroll and most of the scalar bookkeeping folds away.
struct Cfg {
    int dc, nm, nbOper, T1, q, nsteps, n_lists;
    float offset;
    bool onfly = false;   // the elementary-step tasks of a visit are computed from its degree (irregular codes) instead of read from the table
};

// list ids inside one check-node visit: inputs 0..dc-1, prefixes P(k), suffixes S(k), merges M(k)
__device__ __forceinline__ int list_P(int dc, int k) { return k == 0 ? 0 : dc + (k - 1); }
__device__ __forceinline__ int list_S(int dc, int k) { return k == dc - 1 ? dc - 1 : dc + (dc - 2) + (k - 1); }
__device__ __forceinline__ int list_M(int dc, int k) { return dc + 2 * (dc - 2) + k; }

// One elementary-step task of a stage as the 8-lane groups read it from LDS: x = byte offset of In1 | byte
// offset of In2 << 16 (relative to the wave's list area), y = byte offset of Out | (list id of Out + 1) << 16;
// all-zero = no task for this group.
__device__ __forceinline__ uint2 es_task(int a, int b, int o, int nmp)
{
    if (o < 0) return uint2{0u, 0u};
    return uint2{(unsigned)(a * nmp * 8) | ((unsigned)(b * nmp * 8) << 16), (unsigned)(o * nmp * 8) | ((unsigned)(o + 1) << 16)};
}
constexpr int CN_SCHED_BYTES = EMS_MAX_STEPS * 8 * 8;   // [stage][group] tasks of one check-node visit

// Task (stage t = 1..dc-2, group grp) of the forward/backward/merge schedule of bubble_decoder.c:166-227.
__device__ __forceinline__ uint2 cn_task(int dc, int nmp, int t, int grp)
{
    // step t: forward P(t) = ES(P(t-1), in(t)); backward S(dc-1-t) = ES(S(dc-t), in(dc-1-t)); and the
    // merges M(k) = ES(P(k), S(k+2)) that became ready after step t-1, lowest k first
    const int k1 = t - 1, k2 = dc - 2 - t;
    const bool v1 = k1 <= dc - 3 && 2 * k1 >= dc - 3;
    const bool v2 = k2 >= 0 && k2 <= t - 1 && k2 != k1;
    int ka = -1, kb = -1;
    if (v1 && v2) { ka = min(k1, k2); kb = max(k1, k2); }
    else if (v1) ka = k1;
    else if (v2) ka = k2;
    int a = -1, b = -1, o = -1;
    if (grp == 0) { a = list_P(dc, t - 1); b = t; o = list_P(dc, t); }
    else if (grp == 1) { a = list_S(dc, dc - t); b = dc - 1 - t; o = list_S(dc, dc - 1 - t); }
    else if (grp == 2 && ka >= 0) { a = list_P(dc, ka); b = list_S(dc, ka + 2); o = list_M(dc, ka); }
    else if (grp == 3 && kb >= 0) { a = list_P(dc, kb); b = list_S(dc, kb + 2); o = list_M(dc, kb); }
    return es_task(a, b, o, nmp);
}
// Two visits per wave (k_decode_dual, dc = 4): the wave's list area holds the four input lists of visit 0, the four of
// visit 1, then the six other lists (prefix, suffix, merges) of visit 0 and of visit 1 -- the selection's survivor
// scratch aliases everything behind the inputs, which is dead until the elementary steps start.  Offsets in entries.
// (dc = 4: rows of 8 lanes; dc = 8: rows of 4 lanes.)
__host__ __device__ constexpr int dual_nlists(int dc) { return dc + 3 * (dc - 2); }
__host__ __device__ inline int dual_list_off(int half, int id, int nmp, int dc)
{
    return id < dc ? (half * dc + id) * nmp : (2 * dc + half * (dual_nlists(dc) - dc) + (id - dc)) * nmp;
}
// The first stages of a visit hold its forward and backward step only (the merges M(k) start when the two chains have met):
// 2 tasks per visit, 4 per wave -- such a stage runs with SIXTEEN lanes per task (the merge is the same serial loop either way,
// the post-pass then has half the picks per lane), its tasks in entries 0..3 of the stage's row: visit g >> 1, task g & 1.
#ifndef EMS_NO_WIDE_STAGE
__host__ __device__ constexpr bool dual_stage_wide(int dc, int t) { return 2 * t + 1 < dc; }
#else
__host__ __device__ constexpr bool dual_stage_wide(int, int) { return false; }
#endif
// task of 8-lane group grp (visit grp >> 2, task grp & 3) in stage t
__device__ __forceinline__ uint2 cn_task_dual(int dc, int nmp, int t, int grp)
{
    const bool wide = dual_stage_wide(dc, t);
    if (wide && grp >= 4) return uint2{0u, 0u};
    const int half = wide ? grp >> 1 : grp >> 2;
    const uint2 tk = cn_task(dc, 1, t, wide ? (grp & 1) : (grp & 3));                 // nmp = 1: offsets / 8 = list ids
    if ((tk.y >> 16) == 0u) return uint2{0u, 0u};
    const int a = (int)(tk.x & 0xFFFFu) / 8, b = (int)(tk.x >> 16) / 8, o = (int)(tk.y & 0xFFFFu) / 8;
    return uint2{(unsigned)(dual_list_off(half, a, nmp, dc) * 8) | ((unsigned)(dual_list_off(half, b, nmp, dc) * 8) << 16),
                 (unsigned)(dual_list_off(half, o, nmp, dc) * 8) | ((unsigned)(half * dual_nlists(dc) + o + 1) << 16)};
}
__device__ __forceinline__ void cn_sched_build_dual(uint2 *sched, int dc, int nm, int tid, int nthreads)
{
    for (int i = tid; i < (dc - 2) * 8; i += nthreads) sched[i] = cn_task_dual(dc, list_stride(nm), (i >> 3) + 1, i & 7);
}

// filled once per kernel by the whole workgroup (caller synchronises)
__device__ __forceinline__ void cn_sched_build(uint2 *sched, int dc, int nm, int tid, int nthreads)
{
    for (int i = tid; i < (dc - 2) * 8; i += nthreads) sched[i] = cn_task(dc, list_stride(nm), (i >> 3) + 1, i & 7);
}

// The merge form of a GF(16) / nm = 8 stage of a two-visit wave, for the stages of a visit from the one whose first-occurrence form met
// a tie (see cn_core), as a REAL call: inlined, the merge loop and its post-pass pushed the whole kernel from 115 VGPRs to the 128 cap
// plus 7-11 spilled ones and cost the common path 15 % (33.9 against 40.1 ms per launch with the redo compiled out).  LDS pointers
// travel as address-space-3 pointers, so that the callee addresses LDS directly instead of through flat instructions.
#define EMS_LDS(T) __attribute__((address_space(3))) T *
__device__ __attribute__((noinline)) void es_redo_q16(EMS_LDS(EmsEntry) lists, EMS_LDS(int) lens, EMS_LDS(unsigned) tbl, EMS_LDS(uint8_t) picks,
                                                      int lane, uint2 task, int nbOper, int T1, bool wide)
{
    ST_DUMMY
    if (wide) es_stage_fast<16>((EmsEntry *)lists, (int *)lens, (unsigned *)tbl, (uint8_t *)picks, 16, lane, task, 8, nbOper, T1 ST_ARGS);
    else es_stage_fast<8>((EmsEntry *)lists, (int *)lens, (unsigned *)tbl, (uint8_t *)picks, 16, lane, task, 8, nbOper, T1 ST_ARGS);
}

// TL lanes per elementary-step task: 16 (one visit per wave; the literal form uses 8 of them twice over) or 8 (two visits)
// bins: the 16-word-per-task tables of es_stage_q16 (all +inf), or null: GF(16) / nm = 8 visits then take the first-occurrence form
template <int TL = 16>
__device__ __forceinline__ void cn_core(const Cfg &C, EmsEntry *lists, int *lens, unsigned *tbl, uint8_t *picks, const uint2 *sched,
                                        int lane, bool literal, unsigned *bins ST_PARAMS)
{
    const bool fast = !literal && C.nbOper <= ES_MAX_STEPS && C.nm <= EMS_FAST_NM;   // (a pick byte holds a list position in 5 bits)
    const bool q16 = fast && bins != nullptr && C.q == 16 && C.nm == 8;
    auto task_of = [&](int t) {
        const bool wide = TL == 8 && dual_stage_wide(C.dc, t);      // (two-visit waves: see dual_stage_wide)
        if (TL == 16 && C.onfly) return cn_task(C.dc, list_stride(C.nm), t, lane / 16);   // this visit's own degree
        return sched[(t - 1) * 8 + (wide ? lane / 16 : lane / TL)];
    };
    int t0 = 1;
    if (TL == 8 && q16) {
        // Two-visit waves, GF(16) / nm = 8: the stages in the first-occurrence form, in a loop of their own.  A stage that needs the merge
        // form (two first occurrences with the same value: 5 wave-stages in 100 000) leaves the loop, and the REST of the visit's stages
        // run in the merge form through a real call -- with the call inside the loop the common path paid 3 % for its presence.
        for (; t0 <= C.nsteps; ++t0) {
            const bool wide = dual_stage_wide(C.dc, t0);
            const uint2 task = task_of(t0);
            STAMP(8);
            const bool redo = wide ? es_stage_q16<16>(lists, lens, bins, lane, task, C.nbOper) : es_stage_q16<8>(lists, lens, bins, lane, task, C.nbOper);
            STAMP(9);
            DBG_ADD(3, 1);
#ifdef EMS_ABLQ      // ablation builds of the GF(16) stage (timing experiments; results wrong in the tasks that need the redo):
            if (redo) picks[0] = 1;                                 // 1 = detection kept alive, never acted upon (2: see es_stage_q16)
            wave_lds_sync();
#else
            const bool again = __ballot(redo) != 0ull;
            wave_lds_sync();
            if (__builtin_expect(again, 0)) break;
#endif
        }
        for (; t0 <= C.nsteps; ++t0) {                              // (rare) from the stage that asked for it to the end of the visit
            DBG_ADD(5, 1);
            es_redo_q16((EMS_LDS(EmsEntry))lists, (EMS_LDS(int))lens, (EMS_LDS(unsigned))tbl, (EMS_LDS(uint8_t))picks, lane, task_of(t0), C.nbOper, C.T1,
                        dual_stage_wide(C.dc, t0));
            wave_lds_sync();
        }
        return;
    }
#ifdef EMS_TASK_PREFETCH
    uint2 task_next = task_of(1);
#endif
    for (int t = 1; t <= C.nsteps; ++t) {
        const bool wide = TL == 8 && dual_stage_wide(C.dc, t);
#ifdef EMS_TASK_PREFETCH
        const uint2 task = task_next;
        if (t < C.nsteps) task_next = task_of(t + 1);               // the next stage's task travels under this stage
#else
        const uint2 task = task_of(t);
#endif
        if (q16) {
            STAMP(8);
            const bool redo = es_stage_q16<16>(lists, lens, bins, lane, task, C.nbOper);
            STAMP(9);
            DBG_ADD(3, 1);
            if (__ballot(redo) == 0ull) { wave_lds_sync(); continue; }
            DBG_ADD(5, 1);
            wave_lds_sync();                                        // (rare) some task of the stage needs the exact merge: the whole stage again
        }
        if (fast && wide) es_stage_fast<16>(lists, lens, tbl, picks, C.q, lane, task, C.nm, C.nbOper, C.T1 ST_ARGS);
        else if (fast) es_stage_fast<TL>(lists, lens, tbl, picks, C.q, lane, task, C.nm, C.nbOper, C.T1 ST_ARGS);
        else {
            // literal form: list ids from the byte offsets (es_stage takes entry pointers through ids of stride nmp)
            const int lb = list_stride(C.nm) * 8;
            const bool has = (task.y >> 16) != 0u;
            es_stage(lists, lens, lane, has ? (int)(task.x & 0xFFFFu) / lb : -1, has ? (int)(task.x >> 16) / lb : -1,
                     has ? (int)(task.y & 0xFFFFu) / lb : -1, has ? (int)(task.y >> 16) - 1 : -1, C.nm, C.nbOper);
        }
        wave_lds_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// geometry of one wave for field order Q and LPE lanes per edge
// ------------------------------------------------------------------------------------------------
template <int Q, int LPE> struct Geo {
    static constexpr int NROWS = 64 / LPE;
    static constexpr int EPL = Q / LPE;            // symbols per lane
    static constexpr int VW = EPL >= 4 ? 4 : EPL;  // floats per vector access
    static constexpr int NV = EPL / VW;
    // symbol index of element e of row-lane rl: consecutive lanes own consecutive VW-float vectors
    __device__ static __forceinline__ int sym_of(int e, int rl) { return (e / VW) * (VW * LPE) + rl * VW + (e % VW); }
};

template <int VW> struct VecT;
template <> struct VecT<4> { using T = float4; };
template <> struct VecT<2> { using T = float2; };
template <> struct VecT<1> { using T = float; };

template <int Q, int LPE, typename PT> __device__ __forceinline__ void load_row(PT base, int rl, float (&x)[Geo<Q, LPE>::EPL])
{
    using G = Geo<Q, LPE>;
    using V = typename VecT<G::VW>::T;
#pragma unroll
    for (int j = 0; j < G::NV; ++j) {
        const V v = *reinterpret_cast<const V *>(base + j * (G::VW * LPE) + rl * G::VW);
        const float *pv = reinterpret_cast<const float *>(&v);
#pragma unroll
        for (int i = 0; i < G::VW; ++i) x[j * G::VW + i] = pv[i];
    }
}
template <int Q, int LPE, typename PT> __device__ __forceinline__ void store_row(PT base, int rl, const float (&x)[Geo<Q, LPE>::EPL])
{
    using G = Geo<Q, LPE>;
    using V = typename VecT<G::VW>::T;
#pragma unroll
    for (int j = 0; j < G::NV; ++j) {
        V v;
        float *pv = reinterpret_cast<float *>(&v);
#pragma unroll
        for (int i = 0; i < G::VW; ++i) pv[i] = x[j * G::VW + i];
        *reinterpret_cast<V *>(base + j * (G::VW * LPE) + rl * G::VW) = v;
    }
}
// HBM forms for the APP rows of the two-visit kernel, which are streamed (read once and written once per visit, 15 GB
// per launch): non-temporal accesses, so that they do not displace what IS re-read (descriptors, records); +0.8 % measured,
// only with both sides non-temporal
template <int Q, int LPE> __device__ __forceinline__ void load_row_nt(const float *base, int rl, float (&x)[Geo<Q, LPE>::EPL])
{
    using G = Geo<Q, LPE>;
    typedef float v4f __attribute__((ext_vector_type(4)));
    static_assert(G::VW == 4, "float4 rows");
#pragma unroll
    for (int j = 0; j < G::NV; ++j) {
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(base + j * (4 * LPE) + rl * 4));
        x[j * 4 + 0] = v.x; x[j * 4 + 1] = v.y; x[j * 4 + 2] = v.z; x[j * 4 + 3] = v.w;
    }
}
template <int Q, int LPE> __device__ __forceinline__ void store_row_nt(float *base, int rl, const float (&x)[Geo<Q, LPE>::EPL])
{
    using G = Geo<Q, LPE>;
    typedef float v4f __attribute__((ext_vector_type(4)));
    static_assert(G::VW == 4, "float4 rows");
#pragma unroll
    for (int j = 0; j < G::NV; ++j)
        __builtin_nontemporal_store(v4f{x[j * 4 + 0], x[j * 4 + 1], x[j * 4 + 2], x[j * 4 + 3]}, reinterpret_cast<v4f *>(base + j * (4 * LPE) + rl * 4));
}
template <int Q, int LPE, typename PT> __device__ __forceinline__ void fill_row(PT base, int rl, float val)
{
    float x[Geo<Q, LPE>::EPL];
#pragma unroll
    for (int e = 0; e < Geo<Q, LPE>::EPL; ++e) x[e] = val;
    store_row<Q, LPE>(base, rl, x);
}

// LDS staging-row forms with the index swizzle sw (0 or 32 dwords; 16-byte vectors of 8-lane rows only): vector j of
// the row lives at dword (32 j) ^ sw, i.e. at +sw for even j and -sw for odd j -- two per-lane bases, immediate offsets.
template <int Q, int LPE> __device__ __forceinline__ void load_row_sw(const float *base, int rl, float (&x)[Geo<Q, LPE>::EPL], int sw)
{
    using G = Geo<Q, LPE>;
    static_assert(G::VW == 4 && LPE == 8 && G::NV % 2 == 0, "swizzled rows: 8 lanes x float4");
    const float *be = base + rl * 4 + sw, *bo = base + rl * 4 - sw;
#pragma unroll
    for (int j = 0; j < G::NV; ++j) {
        const float4 v = *reinterpret_cast<const float4 *>((j & 1 ? bo : be) + j * 32);
        x[j * 4 + 0] = v.x; x[j * 4 + 1] = v.y; x[j * 4 + 2] = v.z; x[j * 4 + 3] = v.w;
    }
}
template <int Q, int LPE> __device__ __forceinline__ void store_row_sw(float *base, int rl, const float (&x)[Geo<Q, LPE>::EPL], int sw)
{
    using G = Geo<Q, LPE>;
    static_assert(G::VW == 4 && LPE == 8 && G::NV % 2 == 0, "swizzled rows: 8 lanes x float4");
    float *be = base + rl * 4 + sw, *bo = base + rl * 4 - sw;
#pragma unroll
    for (int j = 0; j < G::NV; ++j)
        *reinterpret_cast<float4 *>((j & 1 ? bo : be) + j * 32) = float4{x[j * 4 + 0], x[j * 4 + 1], x[j * 4 + 2], x[j * 4 + 3]};
}

// ------------------------------------------------------------------------------------------------
// Variable-node top-nm selection (NB_LDPC.c:354-374), one edge per row, literal "nm rounds of first
// strict minimum below 1e5" semantics.  Writes raw (value, symbol index) pairs to list `t`.
// ------------------------------------------------------------------------------------------------
template <int Q, int LPE>
__device__ __forceinline__ void select_rounds(const float (&mvc)[Geo<Q, LPE>::EPL], EmsEntry *raw, int rl, int nm)
{
    using G = Geo<Q, LPE>;
    float w[G::EPL];
#pragma unroll
    for (int e = 0; e < G::EPL; ++e) w[e] = mvc[e];
    for (int k = 0; k < nm; ++k) {
        float lv = w[0];
        int le = 0;
#pragma unroll
        for (int e = 1; e < G::EPL; ++e) {
            const bool lt = w[e] < lv;
            lv = lt ? w[e] : lv;
            le = lt ? e : le;
        }
        const int myg = G::sym_of(le, rl);
        float mv = lv;
        int mg = myg;
        row_argmin<LPE>(mv, mg);
        const bool valid = mv < EMS_SENTINEL;                  // :362 strict '<' against 1e5
        if (rl == 0) raw[k] = EmsEntry{valid ? mv : EMS_SENTINEL, valid ? (unsigned)mg : 0u};
        const bool owner = valid && (myg == mg);               // :368 mark the winner as taken
#pragma unroll
        for (int e = 0; e < G::EPL; ++e) w[e] = (owner && e == le) ? __builtin_inff() : w[e];
    }
}


// ------------------------------------------------------------------------------------------------
// Fast top-nm selection (same result as select_rounds, ~8x fewer instructions).
//   A. find a threshold T with nm <= #{v <= T} <= CAP by counting (secant/bisection between the row
//      minimum and the largest lane-local minimum; every decision is taken on an exact count, the
//      float arithmetic only steers the search);
//   B. compact the survivors {v <= T} of each row into LDS as (value bits, symbol) pairs;
//   C. rank them exactly by the 64-bit key (order-preserving image of the float, symbol) -- ties
//      -> lowest symbol, -0 == +0 -- and let the survivors of rank < nm write the normalised
//      (NB_LDPC.c:372-373), rotated (bubble_decoder.c:133-152) message entry.
// Returns false (wave-uniform) when some row found no such T (huge tie groups, e.g. a noiseless
// channel): the caller then runs the literal select_rounds path.
// ------------------------------------------------------------------------------------------------
constexpr int SEL_CAP = 32;
// survivor entries per edge row: SEL_CAP survivors + one dummy slot per lane + a pad that spreads the rows
// over the LDS banks (row bases 8/20/10 dwords apart modulo 32 for 4/8/16 rows)
__host__ __device__ constexpr int surv_stride(int lpe) { return SEL_CAP + lpe + (lpe == 16 ? 4 : (lpe == 8 ? 2 : 1)); }

// IEEE min/max of two non-NaN floats in one instruction (fminf/fmaxf would add canonicalising v_max's; a
// compare + select costs three issue slots with the VCC hazard).  Not volatile: the compiler may schedule it.
__device__ __forceinline__ float vminf(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmaxf(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// one butterfly step min/max(x, x from the partner lane) as a single DPP instruction.  Inline asm is opaque to
// the hazard recogniser, so the two wait states a DPP read needs after the VALU write of its source are spelt out.
#define EMS_DPP_MM(op, ctl) asm("s_nop 1\n\t" op " %0, %1, %1 " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(x))
template <int CTRL, bool MAX> __device__ __forceinline__ float dpp_minmax_f(float x)
{
    float r;
    if (CTRL == DPP_XOR1) { if (MAX) EMS_DPP_MM("v_max_f32_dpp", "quad_perm:[1,0,3,2]"); else EMS_DPP_MM("v_min_f32_dpp", "quad_perm:[1,0,3,2]"); }
    else if (CTRL == DPP_XOR2) { if (MAX) EMS_DPP_MM("v_max_f32_dpp", "quad_perm:[2,3,0,1]"); else EMS_DPP_MM("v_min_f32_dpp", "quad_perm:[2,3,0,1]"); }
    else if (CTRL == DPP_HMIRROR) { if (MAX) EMS_DPP_MM("v_max_f32_dpp", "row_half_mirror"); else EMS_DPP_MM("v_min_f32_dpp", "row_half_mirror"); }
    else { if (MAX) EMS_DPP_MM("v_max_f32_dpp", "row_mirror"); else EMS_DPP_MM("v_min_f32_dpp", "row_mirror"); }
    return r;
}
__device__ __forceinline__ float vmin3f(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// minimum of a register array, two elements per instruction
template <int N> __device__ __forceinline__ float array_min_f(const float (&x)[N])
{
    float m = x[0];
    int e = 1;
#pragma unroll
    for (; e + 1 < N; e += 2) m = vmin3f(m, x[e], x[e + 1]);
    if (e < N) m = vminf(m, x[e]);
    return m;
}
template <int LPE> __device__ __forceinline__ unsigned row_min_u(unsigned x)
{
    x = min(x, (unsigned)dpp_i<DPP_XOR1>((int)x));
    x = min(x, (unsigned)dpp_i<DPP_XOR2>((int)x));
    if (LPE >= 8) x = min(x, (unsigned)dpp_i<DPP_HMIRROR>((int)x));
    if (LPE == 16) x = min(x, (unsigned)dpp_i<DPP_MIRROR>((int)x));
    return x;
}
template <int LPE, bool MAX> __device__ __forceinline__ float row_minmax_f(float x)
{
    x = dpp_minmax_f<DPP_XOR1, MAX>(x);
    x = dpp_minmax_f<DPP_XOR2, MAX>(x);
    if (LPE >= 8) x = dpp_minmax_f<DPP_HMIRROR, MAX>(x);
    if (LPE == 16) x = dpp_minmax_f<DPP_MIRROR, MAX>(x);
    return x;
}
template <int LPE> __device__ __forceinline__ int row_sum_i(int x)
{
    x += dpp_i<DPP_XOR1>(x);
    x += dpp_i<DPP_XOR2>(x);
    if (LPE >= 8) x += dpp_i<DPP_HMIRROR>(x);
    if (LPE == 16) x += dpp_i<DPP_MIRROR>(x);
    return x;
}
// exclusive prefix sum over the LPE lanes of a row (row_shr shifts; contributions from outside the row masked)
template <int LPE> __device__ __forceinline__ int row_excl_scan_i(int x, int rl)
{
    int inc = x, t;
    t = dpp_i<0x111>(inc); inc += (rl >= 1) ? t : 0;
    t = dpp_i<0x112>(inc); inc += (rl >= 2) ? t : 0;
    if (LPE >= 8) { t = dpp_i<0x114>(inc); inc += (rl >= 4) ? t : 0; }
    if (LPE == 16) { t = dpp_i<0x118>(inc); inc += (rl >= 8) ? t : 0; }
    return inc - x;
}

// bit e of the result is set iff x[e] > T, read off the sign of the packed difference T - x[e] (T must not be -0,
// no NaNs): two elements per subtraction, one v_alignbit per element, no VCC traffic.
template <int N> __device__ __forceinline__ unsigned above_mask(const float (&x)[N], float T)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    static_assert(N % 2 == 0, "pairs");
    const v2f t2 = {T, T};
    unsigned a = 0u;
#pragma unroll
    for (int e = N - 2; e >= 0; e -= 2) {          // descending: element e ends up on bit e
        const v2f d = t2 - v2f{x[e], x[e + 1]};
        a = __builtin_amdgcn_alignbit(a, __float_as_uint(d.y), 31);
        a = __builtin_amdgcn_alignbit(a, __float_as_uint(d.x), 31);
    }
    return a;
}

template <int Q, int LPE, bool RAW = false>
__device__ __forceinline__ bool select_threshold(const float (&mvc)[Geo<Q, LPE>::EPL], EmsEntry *surv, EmsEntry *inl, int rl,
                                                 int lane, int nm, int lh, const uint8_t *t_i2b, bool rowact, bool &big, float hint, float &T_out,
                                                 float *drow, int sw = 0)   // sw: index swizzle of the staging row (dual_row_swz), else 0
{
    using G = Geo<Q, LPE>;
    constexpr int CAPQ = Q < SEL_CAP ? Q : SEL_CAP;   // survivor slots that can hold something (a field of 16 never fills 32)
    constexpr int SPL = CAPQ / LPE;                    // survivors ranked per lane
    static_assert(SPL >= 1 && SPL * LPE == CAPQ, "rows of LPE lanes must tile the survivor slots");
    // wide rows: the survivors of a threshold are kept as a per-lane bit mask (above_mask) and compacted by walking
    // the set bits -- a lane owns EPL symbols but only one or two survive
    constexpr bool BYMASK = (G::EPL >= 4) && (G::EPL <= 32) && (Q > SEL_CAP);   // (the mask is one 32-bit word)
    unsigned am = 0u;
    if (nm > SEL_CAP) return false;                    // more entries wanted than survivor slots: the literal nm rounds (wave-uniform)
    // ---- A ----
    const float lmin = array_min_f(mvc);
    const float rmin = row_minmax_f<LPE, false>(lmin);
    float T = __builtin_inff();
    int cnt = Q;
    int lc = G::EPL;     // this lane's survivors under the final T
    bool done = true;
    if (Q > SEL_CAP) {
        const float rmaxl = row_minmax_f<LPE, true>(lmin);
        float lo = rmin, hi = rmaxl;
        int clo = 1, chi = 0;
        bool hiv = false;
        const bool hinted = hint > rmin && hint < 3.0e38f;             // previous sweep's threshold of this edge
        T = (hinted ? hint : rmaxl) + 0.0f;      // (+0.0f: never -0, see above_mask)
        done = false;
        for (int it = 0; it < 12; ++it) {
            int c = 0;
            if (BYMASK) {
                if constexpr (G::EPL % 2 == 0) am = above_mask(mvc, T);
                c = G::EPL - __popc(am);
            } else {
#pragma unroll
                for (int e = 0; e < G::EPL; ++e) c += (mvc[e] <= T) ? 1 : 0;
            }
            lc = c;                          // (a finished row keeps its T, so the recount is the same number)
            c = row_sum_i<LPE>(c);
            if (!done) {
                cnt = c;
                if (c >= nm && c <= SEL_CAP) {
                    done = true;
                } else {
                    if (c < nm) { lo = T; clo = c; } else { hi = T; chi = c; hiv = true; }
                    float nt;
                    if (!hiv) {
                        nt = (T < rmaxl) ? rmaxl : T + (T - rmin) + 1.0f;       // try the largest lane minimum, then gallop
                    } else {
                        float fr = (float)((nm + SEL_CAP) / 2 - clo) * __builtin_amdgcn_rcpf((float)(chi - clo));   // steering only
                        fr = fr < 0.125f ? 0.125f : (fr > 0.875f ? 0.875f : fr);
                        nt = lo + (hi - lo) * fr;
                        if (!(nt > lo && nt < hi)) nt = lo + (hi - lo) * 0.5f;
                        if (!(nt > lo && nt < hi)) nt = hi;                // no float in between: give up below
                    }
                    T = nt + 0.0f;
                }
            }
            DBG_ADD(1, 1);
            if (__ballot(!done) == 0ull) break;
        }
    }
    DBG_ADD(0, 1);
    if (__ballot(!done) != 0ull) { DBG_ADD(2, 1); return false; }
    T_out = T;

    // ---- B ----  survivors as two arrays (values | symbols) so that the ranking reads four values per LDS access
    // (the row base goes through an opaque register so that the 16 compaction stores address it with their small
    //  immediate offsets instead of re-adding the wave-relative constant each time)
    typedef __attribute__((address_space(3))) float lds_f32;
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    unsigned srow = (unsigned)(uintptr_t)(lds_f32 *)reinterpret_cast<float *>(surv);
    asm volatile("" : "+v"(srow));
    lds_f32 *svals = (lds_f32 *)(uintptr_t)srow;
    lds_u32 *ssyms = (lds_u32 *)(uintptr_t)srow + (SEL_CAP + LPE);
    int off = row_excl_scan_i<LPE>(lc, rl);
    if (BYMASK) {
        // park the row in its (now free) staging row so that a lane can fetch the element a mask bit names
        if constexpr (LPE == 8 && G::VW == 4 && G::NV % 2 == 0) store_row_sw<Q, LPE>(drow, rl, mvc, sw);
        else store_row<Q, LPE>(drow, rl, mvc);
        unsigned dbase = (unsigned)(uintptr_t)(lds_f32 *)drow;
        asm volatile("" : "+v"(dbase));
        const lds_f32 *dl = (const lds_f32 *)(uintptr_t)dbase;
        unsigned S = ~am & (G::EPL >= 32 ? 0xFFFFFFFFu : ((1u << (G::EPL & 31)) - 1u));
        wave_lds_sync();
        // two survivors per round: their values are requested together (one LDS round trip per round instead of one per
        // survivor; measured: 2 per round 388.2 ms per launch, 1 or 4 per round 390.0); a lane that has run out reads element 0
        // and parks it in its dummy slot
        while (__ballot(S != 0u) != 0ull) {
            constexpr int R = 2;
            unsigned symk[R];
            float vk[R];
            bool hk[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                hk[k] = S != 0u;
                const int e = hk[k] ? __ffs(S) - 1 : 0;
                S &= S - 1u;
                symk[k] = (unsigned)(((e & ~(G::VW - 1)) * LPE) + rl * G::VW + (e & (G::VW - 1)));   // Geo::sym_of
                vk[k] = dl[symk[k] ^ (unsigned)sw];
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int slot = hk[k] ? off : SEL_CAP + rl;
                svals[slot] = vk[k];
                ssyms[slot] = symk[k];
                off += hk[k] ? 1 : 0;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < G::EPL; ++e) {
            const bool sv = mvc[e] <= T;
            const int slot = sv ? off : SEL_CAP + rl;                   // non-survivors land in this lane's dummy slot
            svals[slot] = mvc[e];
            ssyms[slot] = (unsigned)G::sym_of(e, rl);
            off += sv ? 1 : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < SPL; ++k) {                                     // slots cnt..CAPQ-1 never count as smaller
        const int idx = rl + k * LPE;
        svals[idx >= cnt ? idx : SEL_CAP + rl] = __builtin_inff();
    }
    wave_lds_sync();

    // ---- C ----  rank = number of strictly smaller survivors, from the sign bits of (v_j - v_i).  Exact unless
    // two survivors compare equal (the reference then takes the lower symbol first) or a -0 is present: both are
    // detected (ranks of a row must add up to cnt(cnt-1)/2) and sent to the literal path.
    typedef float v2f __attribute__((ext_vector_type(2)));
    float myv[SPL];
    unsigned mys[SPL], acc[SPL];
    bool odd = false;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int idx = rl + k * LPE;
        myv[k] = svals[idx];
        mys[k] = ssyms[idx];
        acc[k] = 0u;
        odd = odd || (idx < cnt && __float_as_uint(myv[k]) == 0x80000000u);
    }
#pragma unroll
    for (int j0 = 0; j0 < (Q < SEL_CAP ? Q : SEL_CAP); j0 += 4) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f vj = *(const __attribute__((address_space(3))) v4f *)(svals + j0);
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            const v2f me = {myv[k], myv[k]};
            const v2f dl = v2f{vj.x, vj.y} - me, dh = v2f{vj.z, vj.w} - me;
            acc[k] = __builtin_amdgcn_alignbit(acc[k], __float_as_uint(dl.x), 31);     // acc = acc << 1 | sign
            acc[k] = __builtin_amdgcn_alignbit(acc[k], __float_as_uint(dl.y), 31);
            acc[k] = __builtin_amdgcn_alignbit(acc[k], __float_as_uint(dh.x), 31);
            acc[k] = __builtin_amdgcn_alignbit(acc[k], __float_as_uint(dh.y), 31);
        }
    }
    int rank[SPL], rsum = 0;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const bool valid = rl + k * LPE < cnt;
        rank[k] = valid ? __popc(acc[k]) : 0x7FFF;
        rsum += valid ? rank[k] : 0;
    }
    rsum = row_sum_i<LPE>(rsum);
    if (__ballot(odd || 2 * rsum != cnt * (cnt - 1)) != 0ull) { DBG_ADD(2, 1); return false; }
    const float v0 = rmin < EMS_SENTINEL ? rmin : EMS_SENTINEL;         // the first selected value
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        if (rowact && rank[k] < nm) {
            const bool ok = myv[k] < EMS_SENTINEL;                      // NB_LDPC.c:358-366
            const float v = ok ? myv[k] : EMS_SENTINEL;
            const int g = ok ? (int)mys[k] : 0;
            if (RAW) {                                                  // (value, symbol index) as selected
                inl[rank[k]] = EmsEntry{v, (unsigned)g};
            } else {
                const int gi = gf_mul_log(g, lh, Q - 1);
                const float nv = rank[k] == 0 ? 0.0f : v - v0;
                big = big || !(nv < 1e4f);
                inl[rank[k]] = EmsEntry{nv, (unsigned)t_i2b[gi]};
            }
        }
    }
    if (rowact && rl < 2) inl[nm + rl] = EmsEntry{__builtin_inff(), EMS_S_INVALID};   // list pads
    (void)lane;
    return true;
}

// ------------------------------------------------------------------------------------------------
// One check-node visit by one wave: NB_LDPC.c:329-450 around bubble_decoder.c:72-305.
// DUAL (dc = 4, LPE = 8): TWO visits by one wave -- edge rows 0..3 belong to the visit of check node `cn` as lanes
// 0..31 see it, rows 4..7 to the one lanes 32..63 see; every row-wise phase serves both, and the elementary steps of
// both (2 + 2 tasks in the first stage, 4 + 4 in the second, 8 lanes each) share one merge loop.  gact = false masks
// the global stores of a half that has no visit of its own (it shadows the other half's inputs).
// ------------------------------------------------------------------------------------------------
template <int Q, int LPE, bool DUAL = false>
__device__ __forceinline__ void process_cn(const EmsPlan &P, const Cfg &C, int f, int e0, int vn, int lh, bool first_sweep,
                                           const uint8_t *t_i2b, const uint8_t *t_b2i, float *dense, EmsEntry *surv,
                                           EmsEntry *lists, int *lens, uint8_t *picks, const uint2 *sched, int oid, int lane,
                                           bool gact ST_PARAMS)
{
    using G = Geo<Q, LPE>;
    static_assert(!DUAL || LPE == 8 || LPE == 4, "two visits per wave: 2 x 4 rows of 8 lanes or 2 x 8 rows of 4");
    constexpr int DCD = 32 / LPE;                  // check degree of the two-visit layout
    constexpr bool SWZ = DUAL && LPE == 8 && G::VW == 4 && G::NV % 2 == 0;   // swizzled staging rows (dual_row_swz)
    const int nm = C.nm;
    const int qm1 = Q - 1;
    // Everything a lane derives from its position.  RELANE (the GF(64) two-visit kernels): re-derived at each phase from a laundered
    // lane id -- the compiler otherwise keeps some eighty such values (LDS addresses, lane masks, literals) in VGPRs across the whole
    // kernel loop; re-deriving them costs ~2 % in instructions and brings these kernels under 128 VGPRs = 4 waves per SIMD.
#ifndef EMS_NO_RELANE
    constexpr bool RELANE = DUAL && (Q == 64 || (Q == 16 && EMS_Q16_RELANE));
#else
    constexpr bool RELANE = false;
#endif
    int rl, row, half, t, edge, sw;
    bool rowact, stact;
    float *app_row, *drow;
    uint8_t *rec;
    auto regeo = [&](bool first) {
        if (!first && !RELANE) return;
        if constexpr (RELANE) asm volatile("" : "+v"(lane));
        rl = lane & (LPE - 1);
        row = lane / LPE;
        half = DUAL ? row / DCD : 0;
        rowact = DUAL ? true : row < C.dc;  // the row computes (LDS side)
        stact = DUAL ? gact : rowact;       // ... and stores its results to HBM
        t = DUAL ? (row % DCD) : (rowact ? row : C.dc - 1);   // idle rows shadow the last edge, stores masked
        edge = e0 + t;                      // e0: first edge of the check node this lane works for
        // frame bases are wave-uniform (scalar registers); the per-lane part is a 32-bit offset
        app_row = P.app + (size_t)f * P.N * Q + (unsigned)vn * (unsigned)Q;
        rec = P.ctov + (size_t)f * P.E * P.rec_bytes + (unsigned)edge * (unsigned)P.rec_bytes;
        drow = dense + row * Q;
        sw = SWZ ? dual_row_swz(row) : 0;   // index swizzle of this staging row (see dual_row_swz)
    };
    regeo(true);

    // ---- Mvc = APP - CtoV (NB_LDPC.c:329-337); CtoV is all-zero in the first sweep (:273-279).
    //      Every HBM read of the visit (APP row, record header, the nm record entries) is issued
    //      before the first use, so the visit pays one memory round trip. ----
    constexpr int NREC = (EMS_MAX_NM + LPE - 1) / LPE;
    float mvc[G::EPL];
    float hint = __builtin_inff();
    if constexpr (DUAL && Q >= 256) load_row_nt<Q, LPE>(app_row, rl, mvc);      // (small fields: the whole APP block lives in the caches, n96 lost 3 % with it)
    else load_row<Q, LPE>(app_row, rl, mvc);
    if (!first_sweep) {
        const float dflt = *reinterpret_cast<const float *>(rec + EMS_REC_DFLT(nm));
        hint = *reinterpret_cast<const float *>(rec + EMS_REC_HINT(nm));
        const int cnt = rec[EMS_REC_CNT(nm)];
        float rv[NREC];
        int rg[NREC];
#pragma unroll
        for (int c = 0; c < NREC; ++c) {
            rv[c] = 0.0f; rg[c] = 0;
            if (c * LPE >= nm) continue;                 // (no entry of the record lives there)
            const int k = rl + c * LPE;
            const int kk = k < nm ? k : 0;
            rv[c] = reinterpret_cast<const float *>(rec)[kk];
            rg[c] = rec[EMS_REC_SYM(nm) + kk];
        }
        fill_row<Q, LPE>(drow, rl, dflt);
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < NREC; ++c)
            if (rl + c * LPE < cnt) drow[rg[c] ^ sw] = rv[c];
        wave_lds_sync();
        float c[G::EPL];
        if constexpr (SWZ) load_row_sw<Q, LPE>(drow, rl, c, sw);
        else load_row<Q, LPE>(drow, rl, c);
#pragma unroll
        for (int e = 0; e < G::EPL; ++e) mvc[e] = mvc[e] - c[e];
    }

    STAMP(0);
    regeo(false);
    // ---- nm smallest in ascending order, ties -> lowest symbol (NB_LDPC.c:354-369) ----
    const int nmp = list_stride(nm);
    EmsEntry *inl = lists + (DUAL ? dual_list_off(half, t, nmp, DCD) : t * nmp);
    bool big = false;   // some selected value outside [0, 1e4): the visit then uses the literal bubble loop
    float T_sel = __builtin_inff();
#if defined(EMS_ABL) && EMS_ABL == 5      // ... and no selection either: a fixed list (only with the elementary steps off -- they would choke on its ties)
    for (int k = rl; k < nm + 2; k += LPE) inl[k] = EmsEntry{k < nm ? (float)k * 0.5f + mvc[0] * 1e-9f : __builtin_inff(), k < nm ? (unsigned)((k * 7 + row * 3 + (lh & 15)) & (Q - 1)) : EMS_S_INVALID};
    if (false) {
#else
    if (!select_threshold<Q, LPE>(mvc, surv + row * surv_stride(LPE), inl, rl, lane, nm, lh, t_i2b, rowact, big, hint, T_sel, drow, sw)
#if defined(EMS_ABL) && EMS_ABL == 8
        && false
#endif
       ) {
#endif
        // literal path (huge tie groups): nm rounds of first-minimum, then normalise + rotate
        select_rounds<Q, LPE>(mvc, inl, rl, nm);
        wave_lds_sync();
        constexpr int NT = (EMS_MAX_NM + LPE - 1) / LPE;
        const float v0 = inl[0].v;
        EmsEntry tmp[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const int k = rl + c * LPE;
            const EmsEntry r = inl[k < nm ? k : 0];
            const int gi = gf_mul_log((int)r.s, lh, qm1);
            tmp[c] = EmsEntry{k == 0 ? 0.0f : r.v - v0, (unsigned)t_i2b[gi]};
            big = big || (k < nm && !(tmp[c].v < 1e4f));
        }
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const int k = rl + c * LPE;
            if (rowact && k < nm) inl[k] = tmp[c];
        }
        if (rowact && rl < 2) inl[nm + rl] = EmsEntry{__builtin_inff(), EMS_S_INVALID};
    }
    wave_lds_sync();

    STAMP(1);
    regeo(false);
    // ---- forward / backward / merge elementary steps (bubble_decoder.c:166-227) ----
    // the dense staging rows are dead until the expansion below: their first 4Q words serve as the four dedupe
    // tables of the elementary steps, zeroed here and kept all-zero from stage to stage
    unsigned *tbl = reinterpret_cast<unsigned *>(dense);
    for (int i = lane * 4; i < (DUAL ? 8 : 4) * Q; i += 256) *reinterpret_cast<uint4 *>(tbl + i) = uint4{0u, 0u, 0u, 0u};
    // GF(16): the staging area is twice the dedupe tables (2 dc rows of 16 words against 8 or 4 tables of 16): its second half holds
    // the first-occurrence tables of es_stage_q16, one per task, all +inf
    unsigned *bins = nullptr;
#ifndef EMS_NO_Q16
    if constexpr (Q == 16 && G::NROWS * Q >= 2 * (DUAL ? 8 : 4) * Q) {
        bins = tbl + (DUAL ? 8 : 4) * Q;
        for (int i = lane * 4; i < (DUAL ? 8 : 4) * 16; i += 256) *reinterpret_cast<uint4 *>(bins + i) = uint4{Q16_INF, Q16_INF, Q16_INF, Q16_INF};
    }
#endif
    wave_lds_sync();
#if defined(EMS_ABL) && (EMS_ABL == 1 || EMS_ABL == 5)      // ablation builds (make ablation; timing experiments, results wrong by construction): no elementary steps, the output list of an edge is its input list
    (void)tbl; (void)bins;
#else
#if defined(EMS_ABL) && EMS_ABL == 8
    cn_core<DUAL ? 8 : 16>(C, lists, lens, tbl, picks, sched, lane, false, bins ST_ARGS);
#else
    cn_core<DUAL ? 8 : 16>(C, lists, lens, tbl, picks, sched, lane, __ballot(big) != 0ull, bins ST_ARGS);
#endif
#endif

    STAMP(2);
    regeo(false);
    // ---- inverse rotation, q-wide expansion with default LLR[Stp-1]+offset (:231-278), store the
    //      compact C->V record (NB_LDPC.c:434-440) and APP = Mcv + Mvc (:444-450) ----
#if defined(EMS_ABL) && (EMS_ABL == 1 || EMS_ABL == 5)
    const EmsEntry *ol = inl;
    const int stp = nm;
#else
    const EmsEntry *ol = lists + (DUAL ? dual_list_off(half, oid, nmp, DCD) : oid * nmp);
    const int stp = (C.dc == 2) ? nm : lens[DUAL ? half * dual_nlists(DCD) + oid : oid];
#endif
    const float dflt = ol[stp - 1].v + C.offset;
#if defined(EMS_ABL) && EMS_ABL == 3      // no expansion, no stores, no Decision: one word per row keeps the work before alive
    if (stact && rl == 0) *reinterpret_cast<float *>(rec + EMS_REC_DFLT(nm)) = dflt + mvc[0] + ol[0].v;
    return;
#endif
    fill_row<Q, LPE>(drow, rl, dflt);
    wave_lds_sync();
    for (int k = rl; k < stp; k += LPE) {
        const EmsEntry e = ol[k];
        const int g = gf_div_log((int)t_b2i[e.s & 0xFFu], lh, qm1);
        drow[g ^ sw] = e.v;
        if (stact) {
            reinterpret_cast<float *>(rec)[k] = e.v;
            rec[EMS_REC_SYM(nm) + k] = (uint8_t)g;
        }
    }
    if (stact && rl == 0) {
        *reinterpret_cast<float *>(rec + EMS_REC_HINT(nm)) = T_sel;
        *reinterpret_cast<float *>(rec + EMS_REC_DFLT(nm)) = dflt;
        rec[EMS_REC_CNT(nm)] = (uint8_t)stp;
    }
    wave_lds_sync();
    float napp[G::EPL];
    if constexpr (SWZ) load_row_sw<Q, LPE>(drow, rl, napp, sw);
    else load_row<Q, LPE>(drow, rl, napp);
#pragma unroll
    for (int e = 0; e < G::EPL; ++e) napp[e] = napp[e] + mvc[e];
    if constexpr (DUAL && Q >= 256) { if (stact) store_row_nt<Q, LPE>(app_row, rl, napp); }
    else if (stact) store_row<Q, LPE>(app_row, rl, napp);

    // ---- Decision (tools.c:312-330) of this variable node; the later visit of the sweep wins ----
    //      first strict minimum below 1e5 = the row minimum, at the lowest symbol that attains it
    const float lv = row_minmax_f<LPE, false>(array_min_f(napp));
    unsigned lgu = 0x7FFFFFFFu;
#pragma unroll
    for (int e = G::EPL - 1; e >= 0; --e) lgu = (napp[e] == lv) ? (unsigned)G::sym_of(e, rl) : lgu;   // a lane's symbols ascend with e
    const int lg = (int)row_min_u<LPE>(lgu);
    if (stact && rl == 0) (P.decide + (size_t)f * P.N)[(unsigned)vn] = (uint8_t)(lv < EMS_SENTINEL ? lg : 0);
    wave_lds_sync();
    STAMP(3);
}

// LDS carve of a workgroup: [idx2bin Q][bin2idx Q][first-nonzero-syndrome word, padded to 16] then one
// region per wave: dense rows | lists | lens.
// LDS carve of a workgroup: [idx2bin Q][bin2idx Q][first-nonzero-syndrome word, task counter, pad to 16]
// [done flags, one byte per task, padded to 16] then one region per wave: dense rows | survivors | lists |
// lens | picks.
__host__ __device__ inline int wg_header_bytes(int q, int M) { return 2 * q + 16 + CN_SCHED_BYTES + ((M + 15) & ~15); }

// ------------------------------------------------------------------------------------------------
// The decode kernel: one workgroup per frame, persistent over all sweeps.
//
// Inside a sweep the check nodes form a DAG (two check nodes depend on each other iff they share a
// variable node; the earlier one in reference order goes first).  The host lists them in a topological
// order (level by level) together with, per edge, the task index of the previous visitor of that
// variable node in the same sweep.  Waves claim tasks in that order from an LDS counter, wait on the
// one-byte done flags of their (at most dc) predecessors, run the visit and publish their own flag:
// every APP row sees its visits in exactly the order of NB_LDPC.c:320, but no wave waits for a whole
// level.  The smallest unfinished task never waits on anything unfinished, so the scheme cannot
// deadlock; the spin is bounded anyway.  Sweeps are separated by __syncthreads() (Decision + Syndrom
// need the whole sweep, NB_LDPC.c:468-471).
// ------------------------------------------------------------------------------------------------
// OCC4: the build of a dc <= 4 kernel for four 4-wave workgroups per CU (128 VGPRs instead of ~143; the
// register allocator pays a few moves for it, so the 12-wave-per-CU shapes keep the roomier build).
template <int Q, int LPE, int DC, int NM, bool OCC4 = false>
// (the small GF(64)/dc=4 specialisation runs many 2-wave workgroups per CU: ask for 4 waves per SIMD there)
// (rows of 8 lanes over GF(256) keep 32 symbols per lane: 12 waves = 168 VGPRs instead of the 128 of 16 waves)
__global__ void __launch_bounds__(OCC4 ? 256 : (LPE == 8 ? (Q >= 256 ? 12 : 16) : (LPE == 4 ? 8 : 12)) * 64, (OCC4 || (Q == 64 && DC == 4)) ? 4 : 1) k_decode(const EmsPlan P)
{
    using G = Geo<Q, LPE>;
    Cfg C;
    C.dc = DC > 0 ? DC : P.dc;
    C.nm = NM > 0 ? NM : P.nm;
    C.nbOper = P.nbOper;
    C.T1 = NM > 0 ? NM + 3 : min(P.nbOper, P.nm + 3);          // the specialised kernels require nbOper >= nm+3
    C.q = Q;
    C.nsteps = C.dc - 2;
    C.n_lists = C.dc + 3 * (C.dc - 2);
    C.offset = P.offset;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *t_i2b = smem;
    uint8_t *t_b2i = smem + Q;
    unsigned *first_bad = reinterpret_cast<unsigned *>(smem + 2 * Q);
    unsigned *next_task = first_bad + 1;
    unsigned *spin_fail = first_bad + 2;
    uint2 *sched = reinterpret_cast<uint2 *>(smem + 2 * Q + 16);
    uint8_t *done = smem + 2 * Q + 16 + CN_SCHED_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int f = blockIdx.x;
    const int rl = lane & (LPE - 1);
    const int row = lane / LPE;
    const int trow = row < C.dc ? row : C.dc - 1;
    const int oid = P.out_list[trow];     // list holding this row's output (read once: a kernel-argument load)

    uint8_t *wbase = smem + wg_header_bytes(Q, P.M) + (size_t)wave * P.wave_lds_bytes;
    float *dense = reinterpret_cast<float *>(wbase);
    constexpr int SURV_BYTES = G::NROWS * surv_stride(LPE) * 8;      // survivor scratch of the selection
    EmsEntry *surv = reinterpret_cast<EmsEntry *>(wbase + G::NROWS * Q * 4);
    EmsEntry *lists = reinterpret_cast<EmsEntry *>(wbase + G::NROWS * Q * 4 + SURV_BYTES);
    int *lens = reinterpret_cast<int *>(wbase + G::NROWS * Q * 4 + SURV_BYTES + ((C.n_lists * list_stride(C.nm) * 8 + 15) & ~15));
    uint8_t *picks = reinterpret_cast<uint8_t *>(lens) + ((C.n_lists * 4 + 15) & ~15);   // 8 x 68 pick bytes + 64 dummy dwords

    for (int i = tid; i < Q; i += blockDim.x) {
        t_i2b[i] = P.idx2bin[i];
        t_b2i[i] = P.bin2idx[i];
    }
    for (int i = tid; i < P.M; i += blockDim.x) done[i] = 0;
    cn_sched_build(sched, C.dc, C.nm, tid, blockDim.x);
    if (tid == 0) { *first_bad = 0xFFFFFFFFu; *next_task = 0; *spin_fail = 0; }
    __syncthreads();

#ifdef EMS_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_readcyclecounter();
#endif
    int it = P.nsweeps + 1; // value added to sum_it when the frame never converges (NB_LDPC.c:474)
    unsigned synd = 0;
    for (int sweep = 0; sweep < P.nsweeps; ++sweep) {          // NB_LDPC.c:314
        const unsigned tag = (unsigned)(sweep + 1) & 0xFFu;
        auto claim = [&]() {
            unsigned i = 0;
            if (lane == 0) i = atomicAdd(next_task, 1u);
            return (int)__builtin_amdgcn_readfirstlane((int)i);
        };
        // Irregular check degrees (generic instantiations only): C.dc is the LARGEST degree -- it fixes the lane rows and the stride of the
        // task tables -- and every visit brings its own degree and first edge (t_dc / t_e0); a regular code has them implied.
        const bool irregular = DC == 0 && P.t_dc != nullptr;
        int cur = claim();
        int cn = 0, vn = 0, lh = 0, pred = -1, dcv = C.dc, e0v = 0;
        if (cur < P.M) {
            cn = P.t_cn[cur];                                  // check node | priority << 24
            vn = P.t_vn[cur * C.dc + trow];
            lh = P.t_lh[cur * C.dc + trow];
            pred = P.t_pred[cur * C.dc + trow];
            if (irregular) { dcv = P.t_dc[cur]; e0v = P.t_e0[cur]; }
        }
        while (cur < P.M) {                                    // NB_LDPC.c:320, in dependency order
            // claim the following task now and fetch its descriptor under the current visit
            const int nxt = claim();
            int ncn = 0, nvn = 0, nlh = 0, npred = -1, ndcv = C.dc, ne0v = 0;
            if (nxt < P.M) {
                ncn = P.t_cn[nxt];
                nvn = P.t_vn[nxt * C.dc + trow];
                nlh = P.t_lh[nxt * C.dc + trow];
                npred = P.t_pred[nxt * C.dc + trow];
                if (irregular) { ndcv = P.t_dc[nxt]; ne0v = P.t_e0[nxt]; }
            }
            // wait for the previous visitors of my variable nodes (same sweep)
            {
                const int pp = (rl == 0 && row < C.dc) ? pred : -1;
                unsigned spins = 0;
                bool waited = false;
                for (;;) {
                    const bool ok = pp < 0 || __hip_atomic_load(&done[pp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == tag;
                    if (__ballot(!ok) == 0ull) break;
                    waited = true;
                    if (++spins > (1u << 24)) {                // never expected; keeps a bug from hanging the GPU
                        if (lane == 0) atomicOr(spin_fail, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                // A visit that had to wait for a predecessor sits on the frontier of the dependency chain;
                // one whose inputs were ready long ago has slack.  Let the frontier win the SIMD arbitration.
#ifndef EMS_NO_PRIO
                (void)waited;
                switch (cn >> 24) {                            // static priority from the task's slack (host)
                case 3: __builtin_amdgcn_s_setprio(3); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                default: __builtin_amdgcn_s_setprio(0); break;
                }
#endif
            }
            STAMP(4);
            if (irregular) {
                // this visit's degree: its rows, its forward / backward / merge schedule (bubble_decoder.c:166-227 with dc = rowDegree[node]),
                // the list that holds each edge's output (:231-254); rows beyond it shadow its last edge with their stores masked
                Cfg Cv = C;
                Cv.dc = dcv; Cv.nsteps = dcv - 2; Cv.n_lists = dcv + 3 * (dcv - 2); Cv.onfly = true;
                const int tv = row < dcv ? row : dcv - 1;
                const int oidv = tv == 0 ? list_S(dcv, 1) : (tv == dcv - 1 ? list_P(dcv, dcv - 2) : list_M(dcv, tv - 1));
                process_cn<Q, LPE>(P, Cv, f, e0v, vn, lh, sweep == 0, t_i2b, t_b2i, dense, surv, lists, lens, picks, sched, oidv, lane, true ST_ARGS);
            } else {
                process_cn<Q, LPE>(P, C, f, (cn & 0xFFFFFF) * C.dc, vn, lh, sweep == 0, t_i2b, t_b2i, dense, surv, lists, lens, picks, sched, oid, lane, true ST_ARGS);
            }
            // publish: all of this wave's global stores first (the fence waits for the whole wave's vmcnt)
#if defined(EMS_ABL) && EMS_ABL == 7      // timing experiment (results wrong): the flag goes up without waiting for the stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#else
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#endif
            if (lane == 0) __hip_atomic_store(&done[cur], (uint8_t)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            cur = nxt; cn = ncn; vn = nvn; lh = nlh; pred = npred; dcv = ndcv; e0v = ne0v;
        }
        STAMP(6);
        __syncthreads();
        STAMP(4);
        if (tid == 0) *next_task = 0;
        // ---- Syndrom (tools.c:284-299): value of the first check (reference order) with a non-zero
        //      sum; 0 iff codeword.  One thread per check node.
        for (int c = tid; c < P.M; c += blockDim.x) {
            unsigned acc = 0;
            const int e_lo = DC > 0 ? c * DC : P.cn_ptr[c], e_hi = DC > 0 ? e_lo + DC : P.cn_ptr[c + 1];
            for (int ed = e_lo; ed < e_hi; ++ed) {
                const int x = P.decide[(size_t)f * P.N + P.cn_vn[ed]];
                acc ^= t_i2b[gf_mul_log(x, P.cn_lh[ed], Q - 1)];
            }
            if (acc) atomicMin(first_bad, ((unsigned)c << 8) | t_b2i[acc]);
        }
        __syncthreads();
        const unsigned fb = *first_bad;
        __syncthreads();
        if (tid == 0) *first_bad = 0xFFFFFFFFu;
        synd = (fb == 0xFFFFFFFFu) ? 0u : (fb & 0xFFu);
        STAMP(5);
        if (synd == 0 && P.early_exit) {                       // NB_LDPC.c:470-471
            it = sweep + 1;
            break;
        }
    }
#ifdef EMS_STAMPS
    if (lane == 0 && P.stamps)
        for (int i = 0; i < 8; ++i) atomicAdd(&P.stamps[(wave < 9 ? 0 : 8) + i], st_acc[i]);
#endif
    if (tid == 0) {
        P.iters[f] = *spin_fail ? -1 : it;
        P.synd[f] = (int)synd;
        if (*spin_fail && P.fail) atomicOr(P.fail, 1);           // sticky per handle: every waiting entry point reports it
    }
}


// ------------------------------------------------------------------------------------------------
// The decode kernel with TWO check-node visits per wave (dc = 4; see process_cn<.., DUAL>): the elementary steps are
// 45 % of a visit and their serial merge loop costs the same whether 2 or 8 of the wave's 8-lane groups carry a task,
// so a wave takes a PAIR of mutually independent check nodes and the merge loops run half as often per check node.
// The host pairs the check nodes (ems_core.cpp, build_pairs): pair i depends only on pairs < i, claimed in that order
// from the LDS counter like the tasks of k_decode; a pair with one member leaves rows 4..7 shadowing rows 0..3.
// One done BIT per pair (reset between sweeps) instead of a tagged byte per task, the schedule table holds the two
// stages only, the survivor scratch aliases the non-input lists: 12.4 KiB of LDS per wave, and 2-wave workgroups run six
// to a CU -- 12 waves, 24 visits in flight (k_decode: 16 waves, 16 visits).
// LDS carve: [idx2bin Q][bin2idx Q][first_bad, next_task, spin_fail, pad][schedule 2 x 8 tasks][done bits] then per
// wave: 8 staging rows | 8 input lists | 12 other lists | lens | picks  (survivors alias from the other lists on).
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int dual_sched_bytes(int dc) { return (dc - 2) * 8 * 8; }
__host__ __device__ inline int dual_header_bytes(int q, int dc, int n_pairs) { return 2 * q + 16 + dual_sched_bytes(dc) + (((n_pairs + 31) / 32 * 4 + 15) & ~15); }
__host__ __device__ inline int dual_dense_bytes(int q, int dc) { return (2 * dc * q > 8 * q ? 2 * dc * q : 8 * q) * 4; }   // 2 dc staging rows; at least the 8 dedupe tables
__host__ __device__ inline int dual_wave_bytes(int q, int dc, int nm)
{
    const int nmp = list_stride(nm), nl = dual_nlists(dc), lpe = 32 / dc;
    const int other = (2 * (nl - dc) * nmp * 8 + ((2 * nl * 4 + 15) & ~15) + ES_PICK_BYTES + 15) & ~15;   // other lists | lens | picks
    const int survb = 2 * dc * surv_stride(lpe) * 8;
    return dual_dense_bytes(q, dc) + 2 * dc * nmp * 8 + (other > survb ? other : ((survb + 15) & ~15));
}

#ifndef EMS_Q64_BLOCKS
#define EMS_Q64_BLOCKS 4
#endif
#ifndef EMS_DC8_WAVES
#define EMS_DC8_WAVES 16     // waves per workgroup of the dc = 8 build (16: 128 VGPRs, 12: 168)
#endif
// DC = 4 (GF(256) headline): at most four waves per workgroup, 168 VGPRs for three waves per SIMD (GF(64): 128 for four, see RELANE).
// DC = 8 (GF(16), 4 symbols per lane): up to 16 waves per workgroup.
template <int Q, int NM, int DC>
__global__ void __launch_bounds__(DC == 4 ? 256 : EMS_DC8_WAVES * 64, DC == 4 ? (Q == 64 ? EMS_Q64_BLOCKS : 3) : 1) k_decode_dual(const EmsPlan P)
{
    constexpr int LPE = 32 / DC;
    constexpr int DUAL_DC = DC, DUAL_NLISTS = dual_nlists(DC), DUAL_SCHED_BYTES = dual_sched_bytes(DC);
    Cfg C;
    C.dc = DUAL_DC;
    C.nm = NM > 0 ? NM : P.nm;
    C.nbOper = P.nbOper;
#ifndef EMS_T1X
#define EMS_T1X 5
#endif
#ifndef EMS_T1A
#define EMS_T1A 3
#endif
    // first-pass length of the merge: nm + 3 picks; GF(16) sees so many duplicate symbols (16 symbols, 8 wanted) that a third
    // of the stages needed the 8-step continuation -- a longer first pass is cheaper there
    C.T1 = NM > 0 ? min(P.nbOper, NM + (Q == 16 ? EMS_T1X : EMS_T1A)) : min(P.nbOper, P.nm + 3);   // (specialised kernels: nbOper >= nm+3)
    C.q = Q;
    C.nsteps = DUAL_DC - 2;
    C.n_lists = DUAL_NLISTS;
    C.offset = P.offset;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *t_i2b = smem;
    uint8_t *t_b2i = smem + Q;
    unsigned *first_bad = reinterpret_cast<unsigned *>(smem + 2 * Q);
    unsigned *next_task = first_bad + 1;
    unsigned *spin_fail = first_bad + 2;
    uint2 *sched = reinterpret_cast<uint2 *>(smem + 2 * Q + 16);
    unsigned *done = reinterpret_cast<unsigned *>(smem + 2 * Q + 16 + DUAL_SCHED_BYTES);
    const int NP = P.n_pairs, ndw = (NP + 31) / 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int f = blockIdx.x;
    const int rl = lane & (LPE - 1);
    const int row = lane / LPE;
    const int half = row / DC;
    const int oid = P.out_list[row % DC];

    const int nmp = list_stride(C.nm);
    uint8_t *wbase = smem + dual_header_bytes(Q, DC, NP) + (size_t)wave * P.wave_lds_bytes;
    float *dense = reinterpret_cast<float *>(wbase);
    EmsEntry *lists = reinterpret_cast<EmsEntry *>(wbase + dual_dense_bytes(Q, DC));
    uint8_t *other = wbase + dual_dense_bytes(Q, DC) + 2 * DUAL_DC * nmp * 8;
    EmsEntry *surv = reinterpret_cast<EmsEntry *>(other);                       // dead before the elementary steps start
    int *lens = reinterpret_cast<int *>(other + 2 * (DUAL_NLISTS - DUAL_DC) * nmp * 8);
    uint8_t *picks = reinterpret_cast<uint8_t *>(lens) + ((2 * DUAL_NLISTS * 4 + 15) & ~15);

    for (int i = tid; i < Q; i += blockDim.x) {
        t_i2b[i] = P.idx2bin[i];
        t_b2i[i] = P.bin2idx[i];
    }
    for (int i = tid; i < ndw; i += blockDim.x) done[i] = 0u;
    cn_sched_build_dual(sched, DC, C.nm, tid, blockDim.x);
    if (tid == 0) { *first_bad = 0xFFFFFFFFu; *next_task = 0; *spin_fail = 0; }
    __syncthreads();

#ifdef EMS_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_readcyclecounter();
#endif
    int it = P.nsweeps + 1; // value added to sum_it when the frame never converges (NB_LDPC.c:474)
    unsigned synd = 0;
    for (int sweep = 0; sweep < P.nsweeps; ++sweep) {          // NB_LDPC.c:314
        auto claim = [&]() {
            unsigned i = 0;
            if (lane == 0) i = atomicAdd(next_task, 1u);
            return (int)__builtin_amdgcn_readfirstlane((int)i);
        };
        int cur = claim();
        int cnw = 0, vn = 0, lh = 0, pred = -1;
        if (cur < NP) {
            cnw = P.p_cn[cur * 2 + half];                      // check node | priority << 24 | (no visit of its own) << 30
            vn = P.p_vn[cur * (2 * DC) + row];
            lh = P.p_lh[cur * (2 * DC) + row];
            pred = P.p_pred[cur * (2 * DC) + row];
        }
        while (cur < NP) {                                     // NB_LDPC.c:320, in dependency order
            const int nxt = claim();                           // the following pair's descriptor travels under this one
            int ncnw = 0, nvn = 0, nlh = 0, npred = -1;
            if (nxt < NP) {
                ncnw = P.p_cn[nxt * 2 + half];
                nvn = P.p_vn[nxt * (2 * DC) + row];
                nlh = P.p_lh[nxt * (2 * DC) + row];
                npred = P.p_pred[nxt * (2 * DC) + row];
            }
            const bool gact = ((cnw >> 30) & 1) == 0;
            {   // wait for the previous visitors of my variable nodes (same sweep)
                const int pp = (rl == 0 && gact) ? pred : -1;
                unsigned spins = 0;
                for (;;) {
                    const bool ok = pp < 0 || ((__hip_atomic_load(&done[pp >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> (pp & 31)) & 1u) != 0u;
                    if (__ballot(!ok) == 0ull) break;
                    if (++spins > (1u << 24)) {                // never expected; keeps a bug from hanging the GPU
                        if (lane == 0) atomicOr(spin_fail, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                switch ((__builtin_amdgcn_readfirstlane(cnw) >> 24) & 3) {   // static priority: the more urgent of the two (host)
                case 3: __builtin_amdgcn_s_setprio(3); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                default: __builtin_amdgcn_s_setprio(0); break;
                }
            }
            STAMP(4);
            process_cn<Q, LPE, true>(P, C, f, (cnw & 0xFFFFFF) * DC, vn, lh, sweep == 0, t_i2b, t_b2i, dense, surv, lists, lens, picks, sched, oid, lane, gact ST_ARGS);
            // publish: all of this wave's global stores first (the fence waits for the whole wave's vmcnt)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) atomicOr(&done[cur >> 5], 1u << (cur & 31));
            cur = nxt; cnw = ncnw; vn = nvn; lh = nlh; pred = npred;
        }
        STAMP(6);
        __syncthreads();
        STAMP(7);
        if (tid == 0) *next_task = 0;
        for (int i = tid; i < ndw; i += blockDim.x) done[i] = 0u;
        // ---- Syndrom (tools.c:284-299): value of the first check (reference order) with a non-zero sum ----
        for (int c = tid; c < P.M; c += blockDim.x) {
            unsigned acc = 0;
            for (int i = 0; i < DUAL_DC; ++i) {
                const int ed = c * DUAL_DC + i;
                const int x = P.decide[(size_t)f * P.N + P.cn_vn[ed]];
                acc ^= t_i2b[gf_mul_log(x, P.cn_lh[ed], Q - 1)];
            }
            if (acc) atomicMin(first_bad, ((unsigned)c << 8) | t_b2i[acc]);
        }
        __syncthreads();
        const unsigned fb = *first_bad;
        __syncthreads();
        if (tid == 0) *first_bad = 0xFFFFFFFFu;
        synd = (fb == 0xFFFFFFFFu) ? 0u : (fb & 0xFFu);
        STAMP(5);
        if (synd == 0 && P.early_exit) {                       // NB_LDPC.c:470-471
            it = sweep + 1;
            break;
        }
    }
#ifdef EMS_STAMPS
    if (lane == 0 && P.stamps)
        for (int i = 0; i < 16; ++i) atomicAdd(&P.stamps[i], st_acc[i]);
#endif
    if (tid == 0) {
        P.iters[f] = *spin_fail ? -1 : it;
        P.synd[f] = (int)synd;
        if (*spin_fail && P.fail) atomicOr(P.fail, 1);           // sticky per handle: every waiting entry point reports it
    }
}

// ------------------------------------------------------------------------------------------------
// TWO FRAMES per wave (k_decode_twin; tiny codes, dc = 4).  The N = 96 bits code has 8 check nodes in dependency levels of
// 1, 3, 3, 1: pairing check nodes of ONE frame needs six pair slots for them, four of which hold a single check node.  Here a
// wave owns frames 2w and 2w+1 and walks the check nodes in reference order (NB_LDPC.c:320): rows 0..3 are the edges of the
// check node in the first frame, rows 4..7 the same edges in the second -- always independent, every slot full, no schedule
// table, no dependency flags, no claims.  A frame that has converged (or does not exist: odd batch) keeps computing with
// its stores masked.  Everything else is process_cn<.., DUAL> with a per-lane frame index.
// LDS carve: [idx2bin Q][bin2idx Q][first_bad of frame 0, of frame 1, pad][schedule] then the wave's region as k_decode_dual.
// ------------------------------------------------------------------------------------------------
template <int Q, int NM, int DC>
__global__ void __launch_bounds__(64, Q == 64 ? 16 : 12) k_decode_twin(const EmsPlan P, int B)
{
    static_assert(DC == 4, "rows of 8 lanes");
    constexpr int LPE = 32 / DC;
    constexpr int DUAL_NLISTS = dual_nlists(DC);
    Cfg C;
    C.dc = DC;
    C.nm = NM;
    C.nbOper = P.nbOper;
    C.T1 = min(P.nbOper, NM + 3);
    C.q = Q;
    C.nsteps = DC - 2;
    C.n_lists = DUAL_NLISTS;
    C.offset = P.offset;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *t_i2b = smem;
    uint8_t *t_b2i = smem + Q;
    unsigned *first_bad = reinterpret_cast<unsigned *>(smem + 2 * Q);      // [2]
    uint2 *sched = reinterpret_cast<uint2 *>(smem + 2 * Q + 16);
    const int lane = threadIdx.x;
    const int rl = lane & (LPE - 1);
    const int row = lane / LPE;
    const int half = row / DC;
    const int t = row % DC;
    const int oid = P.out_list[t];
    const int f0 = 2 * (int)blockIdx.x;
    const bool two = f0 + 1 < B;
    const int fl = f0 + ((half != 0 && two) ? 1 : 0);          // this lane's frame (the second half shadows the first when there is none)

    const int nmp = list_stride(C.nm);
    uint8_t *wbase = smem + dual_header_bytes(Q, DC, 0);
    float *dense = reinterpret_cast<float *>(wbase);
    EmsEntry *lists = reinterpret_cast<EmsEntry *>(wbase + dual_dense_bytes(Q, DC));
    uint8_t *other = wbase + dual_dense_bytes(Q, DC) + 2 * DC * nmp * 8;
    EmsEntry *surv = reinterpret_cast<EmsEntry *>(other);
    int *lens = reinterpret_cast<int *>(other + 2 * (DUAL_NLISTS - DC) * nmp * 8);
    uint8_t *picks = reinterpret_cast<uint8_t *>(lens) + ((2 * DUAL_NLISTS * 4 + 15) & ~15);

    for (int i = lane; i < Q; i += 64) {
        t_i2b[i] = P.idx2bin[i];
        t_b2i[i] = P.bin2idx[i];
    }
    cn_sched_build_dual(sched, DC, C.nm, lane, 64);
    if (lane < 2) first_bad[lane] = 0xFFFFFFFFu;
    __syncthreads();

#ifdef EMS_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_readcyclecounter();
#endif
    bool run0 = true, run1 = two;                              // (wave-uniform) the frame is still being decoded
    int it0 = P.nsweeps + 1, it1 = P.nsweeps + 1;              // value added to sum_it when the frame never converges (NB_LDPC.c:474)
    unsigned synd0 = 0, synd1 = 0;
    for (int sweep = 0; sweep < P.nsweeps; ++sweep) {          // NB_LDPC.c:314
        const bool gact = half ? run1 : run0;
        for (int c = 0; c < P.M; ++c) {                        // NB_LDPC.c:320, reference order
            const int vn = P.cn_vn[c * DC + t];
            const int lh = P.cn_lh[c * DC + t];
            process_cn<Q, LPE, true>(P, C, fl, c * DC, vn, lh, sweep == 0, t_i2b, t_b2i, dense, surv, lists, lens, picks, sched, oid, lane, gact ST_ARGS);
        }
        // ---- Syndrom (tools.c:284-299) of both frames: lanes 0..31 take the check nodes of the first, 32..63 of the second ----
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        const int fs = f0 + ((lane >= 32 && two) ? 1 : 0);
        for (int c = lane & 31; c < P.M; c += 32) {
            unsigned acc = 0;
            for (int i = 0; i < DC; ++i) {
                const int ed = c * DC + i;
                const int x = P.decide[(size_t)fs * P.N + P.cn_vn[ed]];
                acc ^= t_i2b[gf_mul_log(x, P.cn_lh[ed], Q - 1)];
            }
            if (acc) atomicMin(&first_bad[lane >> 5], ((unsigned)c << 8) | t_b2i[acc]);
        }
        __syncthreads();
        const unsigned fb0 = first_bad[0], fb1 = first_bad[1];
        __syncthreads();
        if (lane < 2) first_bad[lane] = 0xFFFFFFFFu;
        if (run0) synd0 = (fb0 == 0xFFFFFFFFu) ? 0u : (fb0 & 0xFFu);
        if (run1) synd1 = (fb1 == 0xFFFFFFFFu) ? 0u : (fb1 & 0xFFu);
        if (P.early_exit) {                                    // NB_LDPC.c:470-471, frame by frame
            if (run0 && synd0 == 0) { it0 = sweep + 1; run0 = false; }
            if (run1 && synd1 == 0) { it1 = sweep + 1; run1 = false; }
            if (!run0 && !run1) break;
        }
    }
#ifdef EMS_STAMPS
    if (lane == 0 && P.stamps)
        for (int i = 0; i < 16; ++i) atomicAdd(&P.stamps[i], st_acc[i]);
#endif
    if (lane == 0) {
        P.iters[f0] = it0;
        P.synd[f0] = (int)synd0;
        if (two) {
            P.iters[f0 + 1] = it1;
            P.synd[f0 + 1] = (int)synd1;
        }
    }
}

// ================================================================================================
// v2 flavour: the decode core of EMS_HS_L-BubbleCheck_UBS_decoder_v2.c --
// max-is-best log-likelihoods, nm-long messages in both directions, 4-candidate bubble check, the same
// horizontal schedule.  Oracle: oracle/ems_v2_oracle.c (pinned against the real functions).
// One wave per check-node visit as in k_decode; first version: literal elementary steps (one lane per
// task, v2.c:1547-1701), the variable-node update (DataPassLog, v2.c:1513-1545) on the selection
// machinery of the main path (the nm largest of x = the nm smallest of -x, negation is exact).
// ================================================================================================
#define V2_INVALID_V (-1000.0f)

// v2.c:1547-1701 on one lane.  in1/in2/out: LDS lists of nm (value, symbol) entries, symbol = binary image, bit 8
// = "index -1".  seen: q words, all zero on entry (left dirty: the caller clears before the next stage).
__device__ __forceinline__ int es_v2_task(const EmsEntry *in1, const EmsEntry *in2, EmsEntry *out, unsigned *seen, int nm,
                                          int nbOper, bool has)
{
    float c0 = 0, c1 = 0, c2 = 0, c3 = 0;               // tab_comp[0][.]
    int r0 = 0, r1 = 1, r2 = 2, r3 = 2, k0 = 0, k1 = 0, k2 = 0, k3 = 1;   // rows / columns (:1620-1631)
    if (has) {
        for (int i = 0; i < nm; ++i) out[i] = EmsEntry{V2_INVALID_V, EMS_S_INVALID};   // :1590-1593
        const float b0 = in2[0].v;
        c0 = in1[0].v + b0; c1 = in1[1].v + b0; c2 = in1[2].v + b0; c3 = in1[2].v + in2[1].v;
    }
    int s = 0;
    bool act = has;
    for (int ss = 0; ss < nbOper; ++ss) {                // :1648
        if (__ballot(act) == 0ull) break;
        // maximum() (:1705-1720): first strict maximum above 0.0, else competitor 0
        float aux = 0.0f;
        int pos = 0;
        if (c0 > aux) { aux = c0; pos = 0; }
        if (c1 > aux) { aux = c1; pos = 1; }
        if (c2 > aux) { aux = c2; pos = 2; }
        if (c3 > aux) { aux = c3; pos = 3; }
        const int r = pos == 0 ? r0 : (pos == 1 ? r1 : (pos == 2 ? r2 : r3));
        const int c = pos == 0 ? k0 : (pos == 1 ? k1 : (pos == 2 ? k2 : k3));
        const float v = pos == 0 ? c0 : (pos == 1 ? c1 : (pos == 2 ? c2 : c3));
        EmsEntry e1 = EmsEntry{0.0f, EMS_S_INVALID}, e2 = e1;
        if (act) { e1 = in1[r]; e2 = in2[c]; }
        if (act && ((e1.s | e2.s) & EMS_S_INVALID)) act = false;           // :1653-1656 break
        if (act) {
            const unsigned sym = (e1.s ^ e2.s) & 0xFFu;                       // ADDGF = XOR of binary images
            if (seen[sym] == 0u) {                                            // :1662-1667
                out[s] = EmsEntry{v, sym};
                seen[sym] = 1u;
                ++s;
            }
            if (s == nm) act = false;                                         // :1669
        }
        if (act && (r >= nm - 1 || c >= nm - 1)) act = false;                 // :1674-1677
        if (act) {                                                            // :1682-1688
            const int u = pos >> 1;
            const int nr = r + u, nc = c + 1 - u;
            const float nv = in1[nr].v + in2[nc].v;
            if (pos == 0) { r0 = nr; k0 = nc; c0 = nv; }
            else if (pos == 1) { r1 = nr; k1 = nc; c1 = nv; }
            else if (pos == 2) { r2 = nr; k2 = nc; c2 = nv; }
            else { r3 = nr; k3 = nc; c3 = nv; }
        }
    }
    if (has && s > 0) {                                                       // :1692-1693 normalisation to the last valid entry
        const float last = out[s - 1].v;
        for (int i = 0; i < s; ++i) out[i].v = out[i].v - last;
    }
    return s;
}

__device__ __forceinline__ int red4_sumi(int x)
{
    x += dpp_i<DPP_XOR1>(x);
    x += dpp_i<DPP_XOR2>(x);
    return x;
}

// v2 elementary step in the merge + post-pass form of the main path (es_stage_fast): the pick sequence of the
// 4-candidate bubble (competitors 0,1 walk along rows 0,1; competitors 2,3 walk down columns 0,1 from row 2,
// v2.c:1620-1631 / :1682-1688) does not depend on the output bookkeeping, so a serial merge records the picks and
// a parallel post-pass applies the stops (:1653 invalid -> before the pick, :1669 nm outputs, :1674 last row/column
// -> after it) and the redundancy control (:1662).  A task owns 16 lanes; every aligned group of 4 runs the merge.
template <int TL = 16>
__device__ __forceinline__ void es_stage_fast_v2(EmsEntry *lists, int *lens, unsigned *tbl_base, uint8_t *picks_base, int q,
                                                 int lane, uint2 task, int nm, int nbOper, int T1)
{
    const int p = lane & 3;                         // competitor
    const int gbase = lane & ~3;
    const int p16 = lane & (TL - 1), gbase16 = lane & ~(TL - 1), gsel = lane / TL;   // TL lanes per task (8: two visits per wave)
    const bool has = (task.y >> 16) != 0u;
    const int o_id = (int)(task.y >> 16) - 1;
    const int nmp = list_stride(nm);
    uint8_t *lb = reinterpret_cast<uint8_t *>(lists);
    const EmsEntry *in1 = reinterpret_cast<const EmsEntry *>(lb + (task.x & 0xFFFFu));
    const EmsEntry *in2 = reinterpret_cast<const EmsEntry *>(lb + (task.x >> 16));
    EmsEntry *out = reinterpret_cast<EmsEntry *>(lb + (task.y & 0xFFFFu));
    unsigned *tbl = tbl_base + gsel * q;
    uint8_t *picks = picks_base + gsel * (ES_MAX_STEPS + 4);
    uint8_t *dummy = picks_base + 8 * (ES_MAX_STEPS + 4) + lane * 4;
    if (has)
        for (int k = p16; k < nmp; k += TL) out[k] = EmsEntry{k < nm ? V2_INVALID_V : -__builtin_inff(), EMS_S_INVALID};   // :1590-1593
    const bool rowc = p < 2;
    const EmsEntry *mv = rowc ? in2 : in1;
    const EmsEntry *mvy = mv - 32 * p;
    const int pos0 = (rowc ? 0 : 2) + 32 * p;
    const float fxv = rowc ? in1[p].v : in2[p - 2].v;
    unsigned hv;
    int pos;
    auto init = [&]() {
        pos = pos0;
        hv = (unsigned)__float_as_int(fxv + mvy[pos0].v);
    };
    int S = 0;
    bool done;
    init();
    es_merge<false, 1>(hv, pos, fxv, mvy, picks, dummy, p, gbase, 0, T1);
    if (__ballot(has && red4_sumi(pos - pos0) != T1) != 0ull) {   // two equal maxima somewhere: redo with the exact rule
        init();
        es_merge<true, 1>(hv, pos, fxv, mvy, picks, dummy, p, gbase, 0, T1);
    }
    int Td = T1;
    if (T1 <= TL) done = es_post<1, 1, TL>(in1, in2, out, tbl, picks, dummy, q, p16, gbase16, p, gbase, has, T1, nm, nbOper, S, hv, pos, fxv, mvy, Td);
    else if (T1 <= 2 * TL) done = es_post<2, 1, TL>(in1, in2, out, tbl, picks, dummy, q, p16, gbase16, p, gbase, has, T1, nm, nbOper, S, hv, pos, fxv, mvy, Td);
    else if (TL == 8 && T1 <= 24) done = es_post<3, 1, TL>(in1, in2, out, tbl, picks, dummy, q, p16, gbase16, p, gbase, has, T1, nm, nbOper, S, hv, pos, fxv, mvy, Td);
    else if (TL == 8 && T1 <= 32) done = es_post<4, 1, TL>(in1, in2, out, tbl, picks, dummy, q, p16, gbase16, p, gbase, has, T1, nm, nbOper, S, hv, pos, fxv, mvy, Td);
    else done = es_post_rt<1, TL>(in1, in2, out, tbl, picks, q, p16, gbase16, has, T1, nm, nbOper, S);
    if (__ballot(has && !done) != 0ull) {
        es_merge<true, 1>(hv, pos, fxv, mvy, picks, dummy, p, gbase, Td, nbOper);
        es_post_rt<1, TL>(in1, in2, out, tbl, picks, q, p16, gbase16, has, nbOper, nm, nbOper, S);
    }
    wave_lds_sync();
    const float last = (has && S > 0) ? out[S - 1].v : 0.0f;         // :1692-1693 normalisation to the last valid entry
    wave_lds_sync();
    if (has)
        for (int k = p16; k < S; k += TL) out[k].v = out[k].v - last;
    if (has && p16 == 0) lens[o_id] = S;
}

// DUAL (dc = 4, rows of 8 lanes): two visits per wave, as process_cn<.., DUAL> -- rows 0..3 / 4..7, 8-lane elementary-step
// tasks of both visits in one merge loop; dst / lastf are then this lane's entries of the pair tables (p_dst, p_last) and
// gact = false masks the HBM stores of a half without a visit of its own.
template <int Q, int LPE, bool DUAL = false>
__device__ __forceinline__ void process_cn_v2(const EmsPlan &P, const Cfg &C, int f, int cn, int task, int vn, int lh,
                                              const uint8_t *t_i2b, const uint8_t *t_b2i, float *dense, EmsEntry *surv,
                                              EmsEntry *lists, int *lens, uint8_t *picks, const uint2 *sched, int oid, int lane,
                                              int dst_in = 0, bool last_in = false, bool gact = true)
{
    using G = Geo<Q, LPE>;
    static_assert(!DUAL || LPE == 8, "two visits per wave: dc = 4, rows of 8 lanes");
    constexpr int TL = DUAL ? 8 : 16;
    const int rl = lane & (LPE - 1), row = lane / LPE;
    const int half = DUAL ? row >> 2 : 0;
    const bool rowact = DUAL ? gact : row < C.dc;              // gates the HBM stores (in DUAL every row computes)
    const int t = DUAL ? (row & 3) : (rowact ? row : C.dc - 1);
    const int nm = C.nm, nmp = list_stride(nm), qm1 = Q - 1;
    const int edge = cn * C.dc + t;
    // frame base is wave-uniform (scalar registers); per-lane parts are 32-bit offsets.  Messages: one block of two records per edge
    // (ems_plan.h): V->C at 2 e R, C->V at (2 e + 1) R
    const unsigned R = (unsigned)ems_v2_rec_bytes(nm);     // (== P.v2_rec: the v2 flavour runs at its create-time nm; a constant in the specialised kernels)
    uint8_t *const f_msg = P.v2_msg + (size_t)f * P.E * 2u * R;
    // every access is f_msg (scalar registers) + ONE 32-bit per-lane byte offset: per-lane 64-bit pointers cost the kernel 13 spilled VGPRs
    auto vrec = [&](int e) { return (unsigned)e * 2u * R; };           // byte offset of edge e's V->C record
    auto crec = [&](int e) { return (unsigned)e * 2u * R + R; };       // ... and of its C->V record
    auto ldf = [&](unsigned off) { return *reinterpret_cast<const float *>(f_msg + off); };
    auto stf = [&](unsigned off, float v) { *reinterpret_cast<float *>(f_msg + off) = v; };
    float *drow = dense + row * Q;

    // the node's likelihood row is needed only by the variable-node update at the end; it is requested first so that
    // its HBM round trip runs under the elementary steps
    float x[G::EPL];
    load_row<Q, LPE>(P.v2_like + (size_t)f * P.N * Q + (unsigned)vn * (unsigned)Q, rl, x);
    // ... and so is the OTHER C->V message of the node where this visit closes the node's sweep (APP below): it was
    // written by an earlier visitor, a predecessor of this task, so it is final now
    const int dst = DUAL ? dst_in : P.t_dst[task * C.dc + t];
    const bool lastv = rowact && (DUAL ? last_in : P.t_last[task * C.dc + t] != 0);
    const unsigned vin = vrec(edge);
    const float hint = ldf(vin + EMS_V2_HINT(nm));           // search start of the selection at the end of the visit
    constexpr int NPRE = (EMS_MAX_NM + LPE - 1) / LPE;
    float ov[NPRE];
    int og[NPRE];
    {
        const unsigned oc = crec(dst);
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            og[j] = -1; ov[j] = 0.0f;
            if (j * LPE >= nm) continue;
            const int k = rl + j * LPE;
            const bool need = lastv && k < nm;
            og[j] = need ? (int)*reinterpret_cast<const int16_t *>(f_msg + (oc + (unsigned)(EMS_V2_SYMS(nm) + 2 * k))) : -1;
            ov[j] = need ? ldf(oc + 4u * (unsigned)k) : 0.0f;
        }
    }

    // ---- inputs: VtoC of my edges (v2.c:2125-2131), rotated by the edge coefficient (:1772-1784) ----
    EmsEntry *inl = lists + (DUAL ? dual_list_off(half, t, nmp, 4) : t * nmp);
    for (int k = rl; k < nmp; k += LPE) {
        EmsEntry e = EmsEntry{__builtin_inff(), EMS_S_INVALID};
        if (k < nm) {
            const float v = ldf(vin + 4u * (unsigned)k);
            const int g = f_msg[vin + (unsigned)(EMS_V2_SYMS(nm) + k)];
            e = EmsEntry{v, (unsigned)t_i2b[gf_mul_log(g, lh, qm1)]};
        }
        if (DUAL || rowact) inl[k] = e;
    }
    wave_lds_sync();

    // ---- forward / backward / merge elementary steps (:1798-1850), one lane per task ----
    unsigned *seen = reinterpret_cast<unsigned *>(dense);
    const bool fast = C.nbOper <= ES_MAX_STEPS && nm <= EMS_FAST_NM;   // pick areas hold 64 steps and 5-bit positions; else the literal form
#if defined(EMS_ABLV) && EMS_ABLV == 1     // ablation builds of the v2 flavour (timing experiments, results wrong): no elementary steps
    if (false) {
#else
    if (fast) {
#endif
        for (int i = lane * 4; i < (DUAL ? 8 : 4) * Q; i += 256) *reinterpret_cast<uint4 *>(seen + i) = uint4{0u, 0u, 0u, 0u};
        wave_lds_sync();
        for (int st = 1; st <= C.nsteps; ++st) {             // TL lanes per task; the dedupe tables stay clean between stages
            const uint2 tk = sched[(st - 1) * 8 + lane / TL];   // (a forward / backward-only stage has its 4 tasks in entries 0..3, see dual_stage_wide;
            es_stage_fast_v2<TL>(lists, lens, seen, picks, Q, lane, tk, nm, C.nbOper, C.T1);   //  16 lanes per task measured 3 % slower here: 4-competitor merges)
            wave_lds_sync();
        }
    }
#if defined(EMS_ABLV) && EMS_ABLV == 1
    for (int st = 1; false && st <= C.nsteps; ++st) {
#else
    for (int st = 1; !fast && st <= C.nsteps; ++st) {
#endif
        for (int i = lane * 4; i < (DUAL ? 8 : 4) * Q; i += 256) *reinterpret_cast<uint4 *>(seen + i) = uint4{0u, 0u, 0u, 0u};
        wave_lds_sync();
        const uint2 tk = sched[(st - 1) * 8 + (lane & 7)];
        const bool has = lane < (DUAL ? 8 : 4) && (tk.y >> 16) != 0u;
        uint8_t *lb = reinterpret_cast<uint8_t *>(lists);
        const EmsEntry *in1 = reinterpret_cast<const EmsEntry *>(lb + (tk.x & 0xFFFFu));
        const EmsEntry *in2 = reinterpret_cast<const EmsEntry *>(lb + (tk.x >> 16));
        EmsEntry *out = reinterpret_cast<EmsEntry *>(lb + (tk.y & 0xFFFFu));
        const int s = es_v2_task(in1, in2, out, seen + (lane & (DUAL ? 7 : 3)) * Q, nm, C.nbOper, has);
        if (has) lens[(int)(tk.y >> 16) - 1] = s;
        wave_lds_sync();
    }

    // ---- outputs: inverse rotation over the valid prefix (:1854-1870), C->V record (:2135-2141) ----
#if defined(EMS_ABLV) && EMS_ABLV == 1
    const EmsEntry *ol = inl;
    const int stp = nm;
#else
    const EmsEntry *ol = lists + (DUAL ? dual_list_off(half, oid, nmp, 4) : oid * nmp);
    const int stp = (C.dc == 2) ? nm : lens[DUAL ? half * dual_nlists(4) + oid : oid];
#endif
    {
        const unsigned co = crec(edge);
        for (int k = rl; k < nm; k += LPE) {
            const EmsEntry e = ol[k];
            const bool valid = k < stp;
            if (rowact) {
                stf(co + 4u * (unsigned)k, e.v);
                *reinterpret_cast<int16_t *>(f_msg + (co + (unsigned)(EMS_V2_SYMS(nm) + 2 * k))) =
                    valid ? (int16_t)gf_div_log((int)t_b2i[e.s & 0xFFu], lh, qm1) : (int16_t)-1;
            }
        }
    }

    // ---- ComputeAPPLog + Decision (:1887-1921, :1961-1975) of my edge's variable node when this is its last visit of
    //      the sweep: both C->V messages of the node are final then (the other one was written by the earlier visitor,
    //      a predecessor of this task).  APP = (likelihood + TMP[first edge]) + TMP[this edge], TMP = -offset except at
    //      the listed symbols.  The APP rows themselves are stored once, by app_decide_v2 when the decode ends. ----
#if defined(EMS_ABLV) && EMS_ABLV == 2     // no in-visit ComputeAPPLog / Decision
    if (false) {
#else
    if (__ballot(lastv) != 0ull) {
#endif
        float a[G::EPL], c[G::EPL];
        fill_row<Q, LPE>(drow, rl, -C.offset);
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < NPRE; ++j)
            if (og[j] >= 0) drow[og[j]] = ov[j];             // (og = -1 where not needed or "index -1")
        wave_lds_sync();
        load_row<Q, LPE>(drow, rl, c);
#pragma unroll
        for (int e = 0; e < G::EPL; ++e) a[e] = x[e] + c[e];
        wave_lds_sync();
        fill_row<Q, LPE>(drow, rl, -C.offset);
        wave_lds_sync();
        for (int k = rl; k < stp; k += LPE) {
            const EmsEntry e = ol[k];
            drow[gf_div_log((int)t_b2i[e.s & 0xFFu], lh, qm1)] = e.v;
        }
        wave_lds_sync();
        load_row<Q, LPE>(drow, rl, c);
#pragma unroll
        for (int e = 0; e < G::EPL; ++e) a[e] = -(a[e] + c[e]);          // negated: first strict maximum = first strict minimum of -APP
        wave_lds_sync();
        const float lv = row_minmax_f<LPE, false>(array_min_f(a));
        unsigned lgu = 0x7FFFFFFFu;
#pragma unroll
        for (int e = G::EPL - 1; e >= 0; --e) lgu = (a[e] == lv) ? (unsigned)G::sym_of(e, rl) : lgu;
        const unsigned lg = row_min_u<LPE>(lgu);
        if (lastv && rl == 0) (P.decide + (size_t)f * P.N)[(unsigned)vn] = (uint8_t)lg;
    }

    // ---- DataPassLog (:1513-1545) for my edge's variable node; result goes to its OTHER edge (:2159-2162) ----
#pragma unroll
    for (int e = 0; e < G::EPL; ++e) x[e] = x[e] - C.offset;                 // :1524
    store_row<Q, LPE>(drow, rl, x);
    wave_lds_sync();
    for (int k = rl; k < stp; k += LPE) {                                     // :1526-1531 (stops at the first -1)
        const EmsEntry e = ol[k];
        const int g = gf_div_log((int)t_b2i[e.s & 0xFFu], lh, qm1);
        drow[g] = drow[g] + e.v + C.offset;
    }
    wave_lds_sync();
    load_row<Q, LPE>(drow, rl, x);
    float mvc[G::EPL];
#pragma unroll
    for (int e = 0; e < G::EPL; ++e) mvc[e] = -x[e];                          // the nm largest of x = the nm smallest of -x
    wave_lds_sync();
    bool big = false;
    float T_sel = __builtin_inff();
#if defined(EMS_ABLV) && EMS_ABLV == 3     // no selection: the old list with a trace of the new values
    for (int k = rl; k < nm; k += LPE) inl[k].v = inl[k].v + mvc[0] * 1e-12f;
#else
    if (!select_threshold<Q, LPE, true>(mvc, surv + row * surv_stride(LPE), inl, rl, lane, nm, lh, t_i2b, DUAL || rowact, big, hint, T_sel, drow)) {
        select_rounds<Q, LPE>(mvc, inl, rl, nm);
    }
#endif
    wave_lds_sync();
    // The threshold that worked goes into the V->C record written below, i.e. to the node's OTHER edge, whose next visit reads it with
    // its message: a search start only -- the two rows of a node differ by one C->V message and want nearly the same threshold -- and
    // the record of this visit's own edge is not written a second time (64 bytes of HBM write per edge and iteration)
    if (rowact && rl == 0) stf(vrec(dst) + EMS_V2_HINT(nm), T_sel);
    const float last = -inl[nm - 1].v;                                        // :1545 normalisation to the nm-th value
    {
        const unsigned vo = vrec(dst);
        for (int k = rl; k < nm; k += LPE) {
            const EmsEntry e = inl[k];
            if (rowact) {
                stf(vo + 4u * (unsigned)k, (-e.v) - last);
                f_msg[vo + (unsigned)(EMS_V2_SYMS(nm) + k)] = (uint8_t)e.s;
            }
        }
    }
    wave_lds_sync();
}

// ComputeAPPLog + Decision (v2.c:1887-1921, :1961-1975) for variable node n on one row of 16 lanes.
template <int Q>
__device__ __forceinline__ void app_decide_v2(const EmsPlan &P, const Cfg &C, int f, int n, bool active, float *drow, int rl, bool store)
{
    constexpr int LPE = 16;
    using G = Geo<Q, LPE>;
    const int nm = C.nm;
    const unsigned R = (unsigned)ems_v2_rec_bytes(nm);
    const uint8_t *const f_msg = P.v2_msg + (size_t)f * P.E * 2u * R;
    float a[G::EPL];
    load_row<Q, LPE>(P.v2_like + (size_t)f * P.N * Q + (unsigned)n * (unsigned)Q, rl, a);           // APP[n][index] = likelihood (:1907-1908)
    for (int tt = 0; tt < 2; ++tt) {                                          // its edges in interleaver order (:1910-1913)
        const int e = P.vn_edge[n * 2 + tt];
        fill_row<Q, LPE>(drow, rl, -C.offset);                                // TMP = -offset (:1899-1900)
        wave_lds_sync();
        const unsigned co = (unsigned)e * 2u * R + R;                         // the edge's C->V record
        for (int k = rl; k < nm; k += LPE) {
            const int g = *reinterpret_cast<const int16_t *>(f_msg + (co + (unsigned)(EMS_V2_SYMS(nm) + 2 * k)));
            if (g >= 0) drow[g] = *reinterpret_cast<const float *>(f_msg + (co + 4u * (unsigned)k));   // :1901-1903 (every valid entry, no break)
        }
        wave_lds_sync();
        float c[G::EPL];
        load_row<Q, LPE>(drow, rl, c);
#pragma unroll
        for (int i = 0; i < G::EPL; ++i) a[i] = a[i] + c[i];
        wave_lds_sync();
    }
    if (store && active) store_row<Q, LPE>(P.v2_app + (size_t)f * P.N * Q + (unsigned)n * (unsigned)Q, rl, a);
    // Decision: first strict maximum from symbol 0 = the row maximum at its lowest symbol
    float neg[G::EPL];
#pragma unroll
    for (int i = 0; i < G::EPL; ++i) neg[i] = -a[i];
    const float lv = row_minmax_f<LPE, false>(array_min_f(neg));
    unsigned lgu = 0x7FFFFFFFu;
#pragma unroll
    for (int i = G::EPL - 1; i >= 0; --i) lgu = (neg[i] == lv) ? (unsigned)G::sym_of(i, rl) : lgu;
    const unsigned lg = row_min_u<LPE>(lgu);
    if (active && rl == 0) (P.decide + (size_t)f * P.N)[n] = (uint8_t)lg;
}

// The older generation's main() samples the bit error count when iterations 9, 19, 29, 39 and 49 BEGIN (v2.c:2174-2243, before that
// iteration's ComputeAPPLog / Decision): the decisions it reads are those of iterations 8, 18, ...  Called by the whole workgroup after
// the Decision + Syndrom of iteration `sweep` when the frame goes on to iteration sweep + 1: bit errors over the first K symbols
// (NB_LDPC.c:479-485 counting) into snap[sweep / 10].  acc: one LDS word, zero on entry and on exit.
__device__ __forceinline__ void v2_snapshot(const EmsPlan &P, int f, int sweep, const uint8_t *t_i2b, unsigned *acc, int tid, int nthreads)
{
    if (!P.v2_snap || sweep % 10 != 8 || sweep > 48 || sweep + 1 >= P.nsweeps) return;   // (workgroup-uniform)
    int n = 0;
    for (int k = tid; k < P.K; k += nthreads)
        n += __popc((unsigned)(t_i2b[P.decide[(size_t)f * P.N + k]] ^ t_i2b[P.v2_cw[(size_t)f * P.N + k]]));
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    if ((tid & 63) == 0 && n) atomicAdd(acc, (unsigned)n);
    __syncthreads();
    if (tid == 0) { P.v2_snap[f * 5 + sweep / 10] = (int)*acc; *acc = 0u; }
    __syncthreads();
}

template <int Q, int LPE = 16, int NM = 0, bool OCC4 = false>
__global__ void __launch_bounds__(OCC4 ? 256 : 768, OCC4 ? 4 : 1) k_decode_v2(const EmsPlan P)
{
    using G = Geo<Q, LPE>;
    Cfg C;
    C.dc = P.dc; C.nm = NM > 0 ? NM : P.nm; C.nbOper = P.nbOper; C.T1 = NM > 0 ? NM + 3 : min(P.nbOper, P.nm + 3);   // (NM > 0: nbOper >= NM + 3)
    C.q = Q; C.nsteps = C.dc - 2;
    C.n_lists = C.dc + 3 * (C.dc - 2); C.offset = P.offset;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *t_i2b = smem, *t_b2i = smem + Q;
    unsigned *first_bad = reinterpret_cast<unsigned *>(smem + 2 * Q);
    unsigned *next_task = first_bad + 1, *spin_fail = first_bad + 2;
    uint2 *sched = reinterpret_cast<uint2 *>(smem + 2 * Q + 16);
    uint8_t *done = smem + 2 * Q + 16 + CN_SCHED_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, f = blockIdx.x;
    const int rl = lane & (LPE - 1), row = lane / LPE;
    const int trow = row < C.dc ? row : C.dc - 1;
    const int oid = P.out_list[trow];
    uint8_t *wbase = smem + wg_header_bytes(Q, P.M) + (size_t)wave * P.wave_lds_bytes;
    float *dense = reinterpret_cast<float *>(wbase);
    constexpr int SURV_BYTES = G::NROWS * surv_stride(LPE) * 8;
    EmsEntry *surv = reinterpret_cast<EmsEntry *>(wbase + G::NROWS * Q * 4);
    EmsEntry *lists = reinterpret_cast<EmsEntry *>(wbase + G::NROWS * Q * 4 + SURV_BYTES);
    int *lens = reinterpret_cast<int *>(wbase + G::NROWS * Q * 4 + SURV_BYTES + ((C.n_lists * list_stride(C.nm) * 8 + 15) & ~15));
    uint8_t *picks = reinterpret_cast<uint8_t *>(lens) + ((C.n_lists * 4 + 15) & ~15);
    for (int i = tid; i < Q; i += blockDim.x) { t_i2b[i] = P.idx2bin[i]; t_b2i[i] = P.bin2idx[i]; }
    for (int i = tid; i < P.M; i += blockDim.x) done[i] = 0;
    cn_sched_build(sched, C.dc, C.nm, tid, blockDim.x);
    if (tid == 0) { *first_bad = 0xFFFFFFFFu; *next_task = 0; *spin_fail = 0; first_bad[3] = 0u; }
    if (P.v2_snap && tid < 5) P.v2_snap[f * 5 + tid] = 0;
    __syncthreads();

    int it = 0;
    unsigned synd = 0;
    for (int sweep = 0; sweep < P.nsweeps; ++sweep) {                      // iter = 0..NbIterMax (v2.c:2119)
        const unsigned tag = (unsigned)(sweep + 1) & 0xFFu;
        auto claim = [&]() {
            unsigned i = 0;
            if (lane == 0) i = atomicAdd(next_task, 1u);
            return (int)__builtin_amdgcn_readfirstlane((int)i);
        };
        int cur = claim();
        int cnp = 0, vn = 0, lh = 0, pred = -1;
        if (cur < P.M) {
            cnp = P.t_cn[cur];
            vn = P.t_vn[cur * C.dc + trow]; lh = P.t_lh[cur * C.dc + trow]; pred = P.t_pred[cur * C.dc + trow];
        }
        while (cur < P.M) {                                                // check nodes in dependency order (:2122)
            const int nxt = claim();                                       // the next task's descriptor travels under this visit
            int ncnp = 0, nvn = 0, nlh = 0, npred = -1;
            if (nxt < P.M) {
                ncnp = P.t_cn[nxt];
                nvn = P.t_vn[nxt * C.dc + trow]; nlh = P.t_lh[nxt * C.dc + trow]; npred = P.t_pred[nxt * C.dc + trow];
            }
            const int cn = cnp & 0xFFFFFF;
            const int pp = (rl == 0 && row < C.dc) ? pred : -1;
            unsigned spins = 0;
            for (;;) {
                const bool ok = pp < 0 || __hip_atomic_load(&done[pp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == tag;
                if (__ballot(!ok) == 0ull) break;
                if (++spins > (1u << 24)) { if (lane == 0) atomicOr(spin_fail, 1u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            switch (cnp >> 24) {                               // static priority from the task's slack (host)
            case 3: __builtin_amdgcn_s_setprio(3); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
            }
            process_cn_v2<Q, LPE>(P, C, f, cn, cur, vn, lh, t_i2b, t_b2i, dense, surv, lists, lens, picks, sched, oid, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&done[cur], (uint8_t)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            cur = nxt; cnp = ncnp; vn = nvn; lh = nlh; pred = npred;
        }
        __syncthreads();
        if (tid == 0) *next_task = 0;
        const int nwaves = (int)(blockDim.x >> 6);
        // (ComputeAPPLog + Decision ran inside the visits, by the last visitor of every variable node)
        // ---- Syndrom (:1934-1949): value of the first check with a non-zero sum ----
        for (int c = tid; c < P.M; c += blockDim.x) {
            unsigned acc = 0;
            for (int i = 0; i < C.dc; ++i) {
                const int ed = c * C.dc + i;
                const int xx = P.decide[(size_t)f * P.N + P.cn_vn[ed]];
                acc ^= t_i2b[gf_mul_log(xx, P.cn_lh[ed], Q - 1)];
            }
            if (acc) atomicMin(first_bad, ((unsigned)c << 8) | t_b2i[acc]);
        }
        __syncthreads();
        const unsigned fb = *first_bad;
        __syncthreads();
        if (tid == 0) *first_bad = 0xFFFFFFFFu;
        synd = (fb == 0xFFFFFFFFu) ? 0u : (fb & 0xFFu);
        it = sweep + 1;
        if (!(synd == 0 && P.early_exit)) v2_snapshot(P, f, sweep, t_i2b, first_bad + 3, tid, (int)blockDim.x);   // (the frame goes on)
        if ((synd == 0 && P.early_exit) || sweep == P.nsweeps - 1) {      // leaving: store the APP rows of this iteration
            for (int n0 = wave * 4; n0 < P.N; n0 += nwaves * 4) {          // (rows of 16 lanes here, whatever LPE is)
                const int n = n0 + (lane >> 4);
                app_decide_v2<Q>(P, C, f, n < P.N ? n : P.N - 1, n < P.N, dense + (lane >> 4) * Q, lane & 15, true);
            }
        }
        if ((synd == 0 && P.early_exit)) break;
    }
    if (tid == 0) {
        P.iters[f] = *spin_fail ? -1 : it;
        P.synd[f] = (int)synd;
        if (*spin_fail && P.fail) atomicOr(P.fail, 1);           // sticky per handle: every waiting entry point reports it
    }
}

// The v2 decode kernel with TWO check-node visits per wave (dc = 4): k_decode_dual's organisation (pairs claimed in order, one
// done bit per pair, survivor scratch aliased behind the input lists, 2-wave workgroups six to a CU) around
// process_cn_v2<.., DUAL>.  The scratch alias holds here too: the variable-node update (DataPassLog) runs its selection
// AFTER the elementary steps, when the non-input lists, lens and picks are dead.
template <int Q, int NM>
__global__ void __launch_bounds__(256, 3) k_decode_v2_dual(const EmsPlan P)
{
    constexpr int LPE = 8, DC = 4;
    constexpr int DUAL_NLISTS = dual_nlists(DC), DUAL_SCHED_BYTES = dual_sched_bytes(DC);
    Cfg C;
#ifndef EMS_T1V
#define EMS_T1V 3
#endif
    C.dc = DC; C.nm = NM > 0 ? NM : P.nm; C.nbOper = P.nbOper; C.T1 = NM > 0 ? NM + EMS_T1V : min(P.nbOper, P.nm + 3);   // (NM > 0: nbOper >= NM + 3)
    C.q = Q; C.nsteps = DC - 2; C.n_lists = DUAL_NLISTS; C.offset = P.offset;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *t_i2b = smem, *t_b2i = smem + Q;
    unsigned *first_bad = reinterpret_cast<unsigned *>(smem + 2 * Q);
    unsigned *next_task = first_bad + 1, *spin_fail = first_bad + 2;
    uint2 *sched = reinterpret_cast<uint2 *>(smem + 2 * Q + 16);
    unsigned *done = reinterpret_cast<unsigned *>(smem + 2 * Q + 16 + DUAL_SCHED_BYTES);
    const int NP = P.n_pairs, ndw = (NP + 31) / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, f = blockIdx.x;
    const int rl = lane & (LPE - 1), row = lane / LPE, half = row / DC;
    const int oid = P.out_list[row % DC];
    const int nmp = list_stride(C.nm);
    uint8_t *wbase = smem + dual_header_bytes(Q, DC, NP) + (size_t)wave * P.wave_lds_bytes;
    float *dense = reinterpret_cast<float *>(wbase);
    EmsEntry *lists = reinterpret_cast<EmsEntry *>(wbase + dual_dense_bytes(Q, DC));
    uint8_t *other = wbase + dual_dense_bytes(Q, DC) + 2 * DC * nmp * 8;
    EmsEntry *surv = reinterpret_cast<EmsEntry *>(other);
    int *lens = reinterpret_cast<int *>(other + 2 * (DUAL_NLISTS - DC) * nmp * 8);
    uint8_t *picks = reinterpret_cast<uint8_t *>(lens) + ((2 * DUAL_NLISTS * 4 + 15) & ~15);
    for (int i = tid; i < Q; i += blockDim.x) { t_i2b[i] = P.idx2bin[i]; t_b2i[i] = P.bin2idx[i]; }
    for (int i = tid; i < ndw; i += blockDim.x) done[i] = 0u;
    cn_sched_build_dual(sched, DC, C.nm, tid, blockDim.x);
    if (tid == 0) { *first_bad = 0xFFFFFFFFu; *next_task = 0; *spin_fail = 0; first_bad[3] = 0u; }
    if (P.v2_snap && tid < 5) P.v2_snap[f * 5 + tid] = 0;
    __syncthreads();

    int it = 0;
    unsigned synd = 0;
    for (int sweep = 0; sweep < P.nsweeps; ++sweep) {                      // iter = 0..NbIterMax (v2.c:2119)
        auto claim = [&]() {
            unsigned i = 0;
            if (lane == 0) i = atomicAdd(next_task, 1u);
            return (int)__builtin_amdgcn_readfirstlane((int)i);
        };
        int cur = claim();
        int cnw = 0, vn = 0, lh = 0, pred = -1, dst = 0, lastf = 0;
        if (cur < NP) {
            cnw = P.p_cn[cur * 2 + half];
            vn = P.p_vn[cur * 8 + row]; lh = P.p_lh[cur * 8 + row]; pred = P.p_pred[cur * 8 + row];
            dst = P.p_dst[cur * 8 + row]; lastf = P.p_last[cur * 8 + row];
        }
        while (cur < NP) {                                                 // pairs of check nodes in dependency order (:2122)
            const int nxt = claim();
            int ncnw = 0, nvn = 0, nlh = 0, npred = -1, ndst = 0, nlastf = 0;
            if (nxt < NP) {
                ncnw = P.p_cn[nxt * 2 + half];
                nvn = P.p_vn[nxt * 8 + row]; nlh = P.p_lh[nxt * 8 + row]; npred = P.p_pred[nxt * 8 + row];
                ndst = P.p_dst[nxt * 8 + row]; nlastf = P.p_last[nxt * 8 + row];
            }
            const bool gact = ((cnw >> 30) & 1) == 0;
            const int pp = (rl == 0 && gact) ? pred : -1;
            unsigned spins = 0;
            for (;;) {
                const bool ok = pp < 0 || ((__hip_atomic_load(&done[pp >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> (pp & 31)) & 1u) != 0u;
                if (__ballot(!ok) == 0ull) break;
                if (++spins > (1u << 24)) { if (lane == 0) atomicOr(spin_fail, 1u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            switch ((__builtin_amdgcn_readfirstlane(cnw) >> 24) & 3) {
            case 3: __builtin_amdgcn_s_setprio(3); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
            }
            process_cn_v2<Q, LPE, true>(P, C, f, cnw & 0xFFFFFF, 0, vn, lh, t_i2b, t_b2i, dense, surv, lists, lens, picks, sched, oid, lane, dst, lastf != 0, gact);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) atomicOr(&done[cur >> 5], 1u << (cur & 31));
            cur = nxt; cnw = ncnw; vn = nvn; lh = nlh; pred = npred; dst = ndst; lastf = nlastf;
        }
        __syncthreads();
        if (tid == 0) *next_task = 0;
        for (int i = tid; i < ndw; i += blockDim.x) done[i] = 0u;
        const int nwaves = (int)(blockDim.x >> 6);
        // ---- Syndrom (:1934-1949): value of the first check with a non-zero sum ----
        for (int c = tid; c < P.M; c += blockDim.x) {
            unsigned acc = 0;
            for (int i = 0; i < DC; ++i) {
                const int ed = c * DC + i;
                const int xx = P.decide[(size_t)f * P.N + P.cn_vn[ed]];
                acc ^= t_i2b[gf_mul_log(xx, P.cn_lh[ed], Q - 1)];
            }
            if (acc) atomicMin(first_bad, ((unsigned)c << 8) | t_b2i[acc]);
        }
        __syncthreads();
        const unsigned fb = *first_bad;
        __syncthreads();
        if (tid == 0) *first_bad = 0xFFFFFFFFu;
        synd = (fb == 0xFFFFFFFFu) ? 0u : (fb & 0xFFu);
        it = sweep + 1;
        if (!(synd == 0 && P.early_exit)) v2_snapshot(P, f, sweep, t_i2b, first_bad + 3, tid, (int)blockDim.x);   // (the frame goes on)
        if ((synd == 0 && P.early_exit) || sweep == P.nsweeps - 1) {      // leaving: store the APP rows of this iteration
            for (int n0 = wave * 4; n0 < P.N; n0 += nwaves * 4) {          // (rows of 16 lanes here)
                const int n = n0 + (lane >> 4);
                app_decide_v2<Q>(P, C, f, n < P.N ? n : P.N - 1, n < P.N, dense + (lane >> 4) * Q, lane & 15, true);
            }
        }
        if ((synd == 0 && P.early_exit)) break;
    }
    if (tid == 0) {
        P.iters[f] = *spin_fail ? -1 : it;
        P.synd[f] = (int)synd;
        if (*spin_fail && P.fail) atomicOr(P.fail, 1);           // sticky per handle: every waiting entry point reports it
    }
}

// The older generation's initialisation from SYMBOL-INDEXED likelihood rows (what k_llr_const leaves, mode 2): the nm largest of
// a node in decreasing order, ties -> lowest symbol (= the head of decoder.likelihood / likelihood_index, v2.c:1483-1497) go
// to both edges of the node as their first V->C messages (v2.c:2101-2113).  One 64-thread block per (frame, node); rank by
// counting.  A selected value <= -1e5 (the reference would leave index -1 there) is reported through *bad.
__global__ void k_v2_init_rows(const float *__restrict__ likeS, int N, int q, int nm, int E, const int *__restrict__ vn_edge,
                               uint8_t *__restrict__ msg, int rec, int *__restrict__ bad)
{
    __shared__ float v[256];
    const size_t fn = blockIdx.x;
    const int f = (int)(fn / N), n = (int)(fn % N);
    for (int g = threadIdx.x; g < q; g += blockDim.x) v[g] = likeS[fn * q + g];
    __syncthreads();
    const int e0 = vn_edge[n * 2], e1 = vn_edge[n * 2 + 1];
    for (int g = threadIdx.x; g < q; g += blockDim.x) {
        const float vg = v[g];
        int rank = 0;
        for (int j = 0; j < q; ++j) rank += (v[j] > vg || (v[j] == vg && j < g)) ? 1 : 0;
        if (rank < nm) {
            if (!(vg > -EMS_SENTINEL)) atomicOr(bad, 1);
            for (int e : {e0, e1}) {                                    // the V->C record of both edges of the node
                uint8_t *v = msg + ((size_t)f * E + e) * 2 * rec;
                reinterpret_cast<float *>(v)[rank] = vg;
                v[EMS_V2_SYMS(nm) + rank] = (uint8_t)g;
            }
        }
    }
    if (threadIdx.x == 0)
        for (int e : {e0, e1}) *reinterpret_cast<float *>(msg + ((size_t)f * E + e) * 2 * rec + EMS_V2_HINT(nm)) = __builtin_inff();
}

// decoder.likelihood / likelihood_index of one frame (v2.c:1483-1497) from symbol-indexed rows: decreasing (value, symbol)
// order as a stable rank by counting; values <= -1e5 come out as (-1e5, -1)
__global__ void k_sort_like_desc(const float *__restrict__ likeS, int q, float *__restrict__ s_like, int *__restrict__ s_idx)
{
    __shared__ float v[256];
    const size_t base = (size_t)blockIdx.x * q;
    for (int g = threadIdx.x; g < q; g += blockDim.x) v[g] = likeS[base + g];
    __syncthreads();
    for (int g = threadIdx.x; g < q; g += blockDim.x) {
        const float vg = v[g];
        int rank = 0;
        for (int j = 0; j < q; ++j) rank += (v[j] > vg || (v[j] == vg && j < g)) ? 1 : 0;
        const bool ok = vg > -EMS_SENTINEL;
        s_like[base + rank] = ok ? vg : -EMS_SENTINEL;
        s_idx[base + rank] = ok ? g : -1;
    }
}

// builds the symbol-indexed likelihood rows and the initial V->C messages (v2.c:2101-2113) from the sorted layout
__global__ void k_v2_init(const float *__restrict__ like, const int *__restrict__ like_idx, int N, int q, int nm, int E,
                          const int *__restrict__ vn_edge, float *__restrict__ likeS, uint8_t *__restrict__ msg, int rec,
                          int *__restrict__ bad)
{
    const size_t fn = blockIdx.x;                         // (frame, node)
    const int f = (int)(fn / N), n = (int)(fn % N);
    for (int g = threadIdx.x; g < q; g += blockDim.x) {
        const int s = like_idx[fn * q + g];
        if (s < 0 || s >= q) { atomicOr(bad, 1); continue; }
        likeS[fn * q + s] = like[fn * q + g];
    }
    for (int i = threadIdx.x; i < 2 * nm; i += blockDim.x) {
        const int tt = i / nm, k = i - tt * nm;
        const int e = vn_edge[n * 2 + tt];
        const int s = like_idx[fn * q + k];
        uint8_t *v = msg + ((size_t)f * E + e) * 2 * rec;                   // the edge's V->C record
        reinterpret_cast<float *>(v)[k] = like[fn * q + k];
        v[EMS_V2_SYMS(nm) + k] = (uint8_t)(s & 0xFF);
        if (k == 0) *reinterpret_cast<float *>(v + EMS_V2_HINT(nm)) = __builtin_inff();
    }
}

// ------------------------------------------------------------------------------------------------
// Channel LLR (channel.c:66-76).  The reference evaluates, per symbol and bit,
//     TMP = TMP + SQR(y - BPSK(bit)) / (2.0 * SQR(sigma))        (float, float ops, double divide + add)
// The double quotient depends only on (variable node, bit position, bit value): the 2*logq quotients of a
// node are computed once (same operands, same IEEE divide => same bits) and every symbol then accumulates
// them in the reference's order, rounding the running sum to float after every bit as the reference does.
// ------------------------------------------------------------------------------------------------
// One wavefront per variable node (q >= 64; 64/q nodes per wavefront below that), no workgroup barrier: the
// lanes that own a (bit, value) pair publish the quotient in the wave's LDS slot.  The running sum of a symbol
// after bit l depends only on the low l+1 bits of its binary image, so for q >= 64 lane i runs the chain once
// for the six low bits (= i) and then extends it over the high bits for the q/64 images i + 64h -- 12 rounded
// additions instead of 32 for q = 256, each one the very operation the reference performs for that symbol.
// The results go through a per-wave LDS row (image -> symbol index) and leave as one 16-byte store per lane.
__global__ void __launch_bounds__(256) k_llr_init(const float *__restrict__ noisy, float sigma, int logq, int q, size_t nvn,
                                                  const uint8_t *__restrict__ idx2bin, const uint8_t *__restrict__ bin2idx,
                                                  float *__restrict__ app)
{
    __shared__ double quo[4][4][8][2];                         // [wave][node of the wave][bit][bit value]
    __shared__ __attribute__((aligned(16))) float row[4][256]; // [wave][symbol]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int vpw = q >= 64 ? 1 : 64 / q;                      // variable nodes per wavefront pass
    const int spl = q >= 64 ? q / 64 : 1;                      // symbols per lane (<= 4)
    const double denom = 2.0 * (double)(sigma * sigma);        // 2.0*SQR(sigma), SQR in float
    const int nquo = vpw * logq * 2;
    const int qv = lane / (logq * 2), qr = lane - qv * (logq * 2), ql = qr >> 1, qb = qr & 1;
    const size_t stride = (size_t)gridDim.x * 4 * vpw;
    const size_t first = ((size_t)blockIdx.x * 4 + wave) * vpw;
    auto publish = [&](size_t base) {
        if (lane < nquo && base + qv < nvn) {
            const float sg = qb ? -1.0f : 1.0f;                // BPSK(x) = 1-2x (channel.h:14)
            const float a = noisy[(base + qv) * logq + ql] - sg;
            const float sq = a * a;
            quo[wave][qv][ql][qb] = (double)sq / denom;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto retire = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    if (q >= 64) {
        int g[4];                                              // symbol index of image lane + 64h
#pragma unroll
        for (int h = 0; h < 4; ++h) g[h] = h < spl ? bin2idx[lane + 64 * h] : 0;
        for (size_t base = first; base < nvn; base += stride) {
            publish(base);
            float t = 0.0f;
#pragma unroll
            for (int l = 0; l < 6; ++l) t = (float)((double)t + quo[wave][0][l][(lane >> l) & 1]);
            if (spl == 1) {
                row[wave][g[0]] = t;
            } else if (spl == 2) {
                row[wave][g[0]] = (float)((double)t + quo[wave][0][6][0]);
                row[wave][g[1]] = (float)((double)t + quo[wave][0][6][1]);
            } else {
                const float t0 = (float)((double)t + quo[wave][0][6][0]), t1 = (float)((double)t + quo[wave][0][6][1]);
                const double q70 = quo[wave][0][7][0], q71 = quo[wave][0][7][1];
                row[wave][g[0]] = (float)((double)t0 + q70);   // image bits (7,6) = 00, 01, 10, 11
                row[wave][g[1]] = (float)((double)t1 + q70);
                row[wave][g[2]] = (float)((double)t0 + q71);
                row[wave][g[3]] = (float)((double)t1 + q71);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float *dst = app + base * q;
            if (spl == 4) *reinterpret_cast<float4 *>(dst + 4 * lane) = *reinterpret_cast<const float4 *>(&row[wave][4 * lane]);
            else if (spl == 2) *reinterpret_cast<float2 *>(dst + 2 * lane) = *reinterpret_cast<const float2 *>(&row[wave][2 * lane]);
            else dst[lane] = row[wave][lane];
            retire();
        }
        return;
    }
    const int v = lane / q, g0 = lane - v * q;
    const unsigned img = idx2bin[g0];
    for (size_t base = first; base < nvn; base += stride) {
        publish(base);
        if (base + v < nvn) {
            float acc = 0.0f;
            for (int l = 0; l < logq; ++l) acc = (float)((double)acc + quo[wave][v][l][(img >> l) & 1u]);
            app[(base + v) * q + g0] = acc;
        }
        retire();
    }
}

// ------------------------------------------------------------------------------------------------
// Channel LLR of the constellation front ends (channel.c:270-283 ModelChannel_AWGN_64, :676-704 ModelChannel,
// :861-888 ModelChannel_AWGN_256QAM_4D): one GF(q) symbol = one point of a D-dimensional
// constellation (D = 2 or 4), observed per axis with an attenuation (fading, 0 = erased axis).
//   mode 0 (SUM): TMP = SQR(y0 - t0*a)/(2.0*SQR(sigma)) + SQR(y1 - a*t1)/(2.0*SQR(sigma))   -- both quotients double, one rounding;
//                 0 for a node with attenuation <= 0 (channel.c:681-684); a = 1 when no attenuations are given (:277)
//   mode 1 (SSD): TMP = 0; per axis with attenuation > 0: TMP = TMP + SQR(y - a*t)/(2.0*SQR(sigma))   -- float running value
//   mode 2 (NEG): TMP = -(SQR(..)/(..)) - SQR(..)/(..), the older generation's max-is-best likelihood (v2.c:1479)
// with the reference's promotions: differences, products and squares in float, quotient and sum in double.
// One wavefront per (frame, variable node): the table [q][D] (indexed by the symbol's BINARY IMAGE, as the reference's
// `som`) sits in LDS, lane i evaluates the images i + 64 h, the results pass through a per-wave LDS row (image -> symbol
// index) and leave as one coalesced store.  q = 16 (ModelChannel's third field, channel.c:520), 64 or 256.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_llr_const(const float *__restrict__ obs, const float *__restrict__ att,
                                                   const float *__restrict__ table, int q, int D, int mode, float sigma, size_t nvn,
                                                   const uint8_t *__restrict__ bin2idx, float *__restrict__ app)
{
    __shared__ float tab[256 * 4];
    __shared__ uint8_t b2i[256];
    __shared__ __attribute__((aligned(16))) float row[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < q * D; i += blockDim.x) tab[i] = table[i];
    for (int i = threadIdx.x; i < q; i += blockDim.x) b2i[i] = bin2idx[i];
    __syncthreads();
    const double den = 2.0 * (double)(sigma * sigma);          // 2.0*SQR(sigma), SQR in float (channel.h:13)
    const int spl = q >= 64 ? q / 64 : 1;                      // images per lane (GF(16): lanes 16.. idle)
    for (size_t v = (size_t)blockIdx.x * 4 + wave; v < nvn; v += (size_t)gridDim.x * 4) {
        float y[4], a[4];
        for (int d = 0; d < D; ++d) {
            y[d] = obs[v * D + d];
            a[d] = att ? att[v * D + d] : 1.0f;
        }
        for (int h = 0; h < spl; ++h) {
            const int img = lane + 64 * h;
            if (img >= q) break;
            const float *t = tab + img * D;
            float tmp;
            if (mode == 0 || mode == 2) {                       // (a = 1 without attenuations: t * 1.0f is t)
                const float e0 = y[0] - t[0] * a[0], e1 = y[1] - a[1] * t[1];
                if (!(a[0] > 0)) tmp = 0.0f;                     // channel.c:681-684 / v2.c:1479-1481
                else if (mode == 0) tmp = (float)((double)(e0 * e0) / den + (double)(e1 * e1) / den);
                else tmp = (float)(-((double)(e0 * e0) / den) - (double)(e1 * e1) / den);
            } else {
                tmp = 0.0f;
                for (int d = 0; d < D; ++d)
                    if (a[d] > 0) {
                        const float e = y[d] - a[d] * t[d];
                        tmp = (float)((double)tmp + (double)(e * e) / den);
                    }
            }
            row[wave][b2i[img]] = tmp;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float *dst = app + v * q;
        if (spl == 4) *reinterpret_cast<float4 *>(dst + 4 * lane) = *reinterpret_cast<const float4 *>(&row[wave][4 * lane]);
        else if (lane < q) dst[lane] = row[wave][lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// NB_LDPC.c:479-485: bit errors between decide[k] and the codeword over the first K symbols
__global__ void k_count_errors(const uint8_t *__restrict__ decide, const int *__restrict__ cw, int N, int K,
                               const uint8_t *__restrict__ idx2bin, int *__restrict__ nberr)
{
    const int f = blockIdx.x;
    int acc = 0;
    for (int k = threadIdx.x; k < K; k += blockDim.x)
        acc += __popc((unsigned)(idx2bin[decide[(size_t)f * N + k]] ^ idx2bin[cw[(size_t)f * N + k]]));
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    __shared__ int part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += part[w];
        nberr[f] = s;
    }
}

// channel.c:78-91: ascending (value, symbol) order of one variable node's q LLRs, as a stable rank by
// counting (== q rounds of "first strict minimum below 1e5"); values >= 1e5 come out as (1e5, -1).
__global__ void k_sort_llr(const float *__restrict__ app, int q, float *__restrict__ s_llr, int *__restrict__ s_gf)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float *v = reinterpret_cast<float *>(smem);
    const size_t base = (size_t)blockIdx.x * q;
    for (int g = threadIdx.x; g < q; g += blockDim.x) v[g] = app[base + g];
    __syncthreads();
    for (int i = threadIdx.x; i < q; i += blockDim.x) {
        const float vi = v[i];
        int rank = 0;
        for (int j = 0; j < q; ++j) {
            const float vj = v[j];
            rank += (vj < vi || (vj == vi && j < i)) ? 1 : 0;
        }
        const bool ok = vi < EMS_SENTINEL;
        s_llr[base + rank] = ok ? vi : EMS_SENTINEL;
        s_gf[base + rank] = ok ? i : -1;
    }
}

// decoder_t.CtoV as the reference keeps it (dense [E][q]) from the compact records, for parity tests
__global__ void k_expand_ctov(const uint8_t *__restrict__ rec_base, int rec_bytes, int nm, int q, int E,
                              float *__restrict__ out)
{
    const int e = blockIdx.x;
    if (e >= E) return;
    const uint8_t *rec = rec_base + (size_t)e * rec_bytes;
    const float dflt = *reinterpret_cast<const float *>(rec + EMS_REC_DFLT(nm));
    const int cnt = rec[EMS_REC_CNT(nm)];
    for (int g = threadIdx.x; g < q; g += blockDim.x) out[(size_t)e * q + g] = dflt;
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += blockDim.x)
        out[(size_t)e * q + rec[EMS_REC_SYM(nm) + k]] = reinterpret_cast<const float *>(rec)[k];
}

// ------------------------------------------------------------------------------------------------
// Single-node kernels (one wave) behind the reference-shaped ElementaryStep / CheckPassLogEMS /
// Decision / Syndrom symbols: same device functions as the decode kernel.
// ------------------------------------------------------------------------------------------------
__global__ void k_es_unit(const float *in1, const float *in2, const int *idx1, const int *idx2, float *out,
                          int *idxout, int nm, int nbOper, int q, const uint8_t *idx2bin, const uint8_t *bin2idx)
{
    __shared__ EmsEntry lists[3 * (EMS_MAX_NM + 2)];
    __shared__ int lens[4];
    __shared__ __attribute__((aligned(16))) unsigned tbl[4 * 256];
    __shared__ uint8_t picks[ES_PICK_BYTES];
    const int lane = threadIdx.x;
    const int nmp = list_stride(nm);
    bool big = false;
    for (int i = lane; i < 4 * 256; i += 64) tbl[i] = 0u;
    for (int k = lane; k < nmp; k += 64) {
        const bool pad = k >= nm;
        const float v1 = pad ? __builtin_inff() : in1[k], v2 = pad ? __builtin_inff() : in2[k];
        const bool i1 = pad || idx1[k] < 0, i2 = pad || idx2[k] < 0;
        lists[k] = EmsEntry{v1, i1 ? EMS_S_INVALID : (unsigned)idx2bin[idx1[k]]};
        lists[nmp + k] = EmsEntry{v2, i2 ? EMS_S_INVALID : (unsigned)idx2bin[idx2[k]]};
        big = big || (!i1 && !(v1 >= 0.0f && v1 < 1e4f)) || (!i2 && !(v2 >= 0.0f && v2 < 1e4f));
    }
    wave_lds_sync();
    const bool g0 = lane < 16;
    // the fast form needs every valid value in [0, 1e4) (true inside the decoder); anything else -> literal loop
    if (nbOper <= ES_MAX_STEPS && nm <= EMS_FAST_NM && __ballot(big) == 0ull)
    {
        ST_DUMMY
        es_stage_fast(lists, lens, tbl, picks, q, lane, es_task(g0 ? 0 : -1, g0 ? 1 : -1, g0 ? 2 : -1, nmp), nm, nbOper, min(nbOper, nm + 3) ST_ARGS);
    }
    else es_stage(lists, lens, lane, g0 ? 0 : -1, g0 ? 1 : -1, g0 ? 2 : -1, g0 ? 2 : -1, nm, nbOper);
    wave_lds_sync();
    for (int k = lane; k < nm; k += 64) {
        const EmsEntry e = lists[2 * nmp + k];
        out[k] = e.v;
        idxout[k] = (e.s & EMS_S_INVALID) ? -1 : (int)bin2idx[e.s & 0xFFu];
    }
}

template <int Q>
__global__ void k_cn_unit(const EmsPlan P, const int *coef, const float *vtoc_llr, const int *vtoc_gf, float *ctov)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int nm = P.nm, dc = P.dc, nmp = list_stride(P.nm);
    EmsEntry *lists = reinterpret_cast<EmsEntry *>(smem);
    int *lens = reinterpret_cast<int *>(smem + ((P.n_lists * nmp * 8 + 15) & ~15));
    __shared__ __attribute__((aligned(16))) unsigned tbl[4 * 256];
    __shared__ uint8_t picks[ES_PICK_BYTES];
    __shared__ uint2 sched[CN_SCHED_BYTES / 8];
    __shared__ __attribute__((aligned(16))) unsigned bins[4 * 16];           // es_stage_q16's tables (GF(16), nm = 8)
    const int lane = threadIdx.x;
    bool big = false;
    cn_sched_build(sched, dc, nm, lane, 64);
    for (int i = lane; i < 4 * 256; i += 64) tbl[i] = 0u;
    bins[lane] = Q16_INF;
    for (int i = lane; i < dc * nmp; i += 64) {                // rotation, bubble_decoder.c:133-152
        const int t = i / nmp, k = i - t * nmp;
        if (k >= nm) {
            lists[i] = EmsEntry{__builtin_inff(), EMS_S_INVALID};
        } else {
            const int g = vtoc_gf[t * nm + k];
            const float v = vtoc_llr[t * nm + k];
            lists[i] = EmsEntry{v, g < 0 ? EMS_S_INVALID : (unsigned)P.idx2bin[gf_mul_log(g, coef[t] - 1, Q - 1)]};
            big = big || (g >= 0 && !(v >= 0.0f && v < 1e4f));
        }
    }
    wave_lds_sync();
    Cfg C;
    C.dc = dc; C.nm = nm; C.nbOper = P.nbOper; C.T1 = min(P.nbOper, nm + 3); C.q = Q; C.nsteps = dc - 2;
    C.n_lists = P.n_lists; C.offset = P.offset;
    ST_DUMMY
    cn_core(C, lists, lens, tbl, picks, sched, lane, __ballot(big) != 0ull, Q == 16 ? bins : nullptr ST_ARGS);
    for (int t = 0; t < dc; ++t) {                             // bubble_decoder.c:231-278
        const int oid = P.out_list[t];
        const EmsEntry *ol = lists + oid * nmp;
        const int stp = (dc == 2) ? nm : lens[oid];
        const float dflt = ol[stp - 1].v + P.offset;
        for (int g = lane; g < Q; g += 64) ctov[t * Q + g] = dflt;
        __threadfence_block();
        for (int k = lane; k < stp; k += 64)
            ctov[t * Q + gf_div_log((int)P.bin2idx[ol[k].s & 0xFFu], coef[t] - 1, Q - 1)] = ol[k].v;
    }
}

__global__ void k_decision_unit(const float *__restrict__ app, int N, int q, int *__restrict__ decide)
{
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    float lv = EMS_SENTINEL;
    int lg = 0x7FFFFFFF;
    for (int g = lane; g < q; g += 64) {
        const float v = app[(size_t)n * q + g];
        if (v < lv) { lv = v; lg = g; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(lv, o);
        const int og = __shfl_xor(lg, o);
        if (ov < lv || (ov == lv && og < lg)) { lv = ov; lg = og; }
    }
    if (lane == 0) decide[n] = lv < EMS_SENTINEL ? lg : 0;
}

// Decision (tools.c:312-330) for a list of variable nodes of every frame, into the decoder's u8 decision block: the
// variable nodes that no check node touches (their APP is the channel LLR for the whole decode).
__global__ void k_decide_rows(const float *__restrict__ app, const int *__restrict__ rows, int nrows, int N, int q,
                              uint8_t *__restrict__ decide)
{
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    if (r >= nrows) return;
    const int n = rows[r];
    const float *a = app + ((size_t)f * N + n) * q;
    float lv = EMS_SENTINEL;
    int lg = 0x7FFFFFFF;
    for (int g = lane; g < q; g += 64) {
        const float v = a[g];
        if (v < lv) { lv = v; lg = g; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(lv, o);
        const int og = __shfl_xor(lg, o);
        if (ov < lv || (ov == lv && og < lg)) { lv = ov; lg = og; }
    }
    if (lane == 0) decide[(size_t)f * N + n] = (uint8_t)(lv < EMS_SENTINEL ? lg : 0);
}

__global__ void k_syndrome_unit(const int *__restrict__ row_ptr, const int *__restrict__ col, const int *__restrict__ coef,
                                int M, int q, const int *__restrict__ decide, const uint8_t *__restrict__ idx2bin,
                                const uint8_t *__restrict__ bin2idx, unsigned *first_bad)
{
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < M; c += gridDim.x * blockDim.x) {
        unsigned acc = 0;
        for (int e = row_ptr[c]; e < row_ptr[c + 1]; ++e) acc ^= idx2bin[gf_mul_log(decide[col[e]], coef[e] - 1, q - 1)];
        if (acc) atomicMin(first_bad, ((unsigned)c << 8) | bin2idx[acc]);
    }
}

template <typename T> hipError_t set_max_lds(T kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

} // namespace

// ================================================================================================
// host-side launchers
// ================================================================================================
int ems_lpe_for(int dc) { return dc <= 4 ? 16 : (dc <= 8 ? 8 : (dc <= 16 ? 4 : 0)); }   // dc 9..16: GF(16)/GF(64) only
// waves per workgroup the decode kernel is compiled for: the dc <= 4 kernels keep q/16 values per lane in
// registers and need the 170-VGPR budget of 12 waves; the dc <= 8 kernels fit the 128 of 16 waves
int ems_max_waves_for(int dc, int q) { const int l = ems_lpe_for(dc); return l == 8 ? (q >= 256 ? 12 : 16) : (l == 4 ? 8 : 12); }

size_t ems_wave_lds_bytes(int q, int dc, int nm, int n_lists)
{
    const int lpe = ems_lpe_for(dc);
    if (!lpe) return 0;
    const size_t dense = (size_t)(64 / lpe) * q * 4 + (size_t)(64 / lpe) * surv_stride(lpe) * 8; // staging rows + survivor scratch
    const size_t lists = ((size_t)n_lists * list_stride(nm) * 8 + 15) & ~(size_t)15;
    const size_t lens = ((size_t)n_lists * 4 + 15) & ~(size_t)15;
    return dense + lists + lens + ((ES_PICK_BYTES + 15) & ~15); // + pick bytes of the elementary steps (their dedupe tables alias the staging rows)
}

size_t ems_wg_lds_bytes(int q, int M, size_t wave_bytes, int waves) { return (size_t)wg_header_bytes(q, M) + wave_bytes * waves; }

// name of a kernel instantiation as rocprofv3 prints it (without the argument list)
static const char *kname_fmt(const char *fmt, int a, int b, int c = 0, int d = 0, int e = 0)
{
    static thread_local char buf[96];
    snprintf(buf, sizeof buf, fmt, a, b, c, d, e);
    return buf;
}
static thread_local const char *g_kname = "";

template <int Q, int LPE, int DC = 0, int NM = 0, bool OCC4 = false> static hipError_t launch_decode_t(const EmsPlan &P, int B, hipStream_t st)
{
    g_kname = kname_fmt(OCC4 ? "k_decode<%d, %d, %d, %d, true>" : "k_decode<%d, %d, %d, %d, false>", Q, LPE, DC, NM);
    const size_t lds = ems_wg_lds_bytes(Q, P.M, P.wave_lds_bytes, P.waves_per_wg);
    hipError_t e = set_max_lds(k_decode<Q, LPE, DC, NM, OCC4>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode<Q, LPE, DC, NM, OCC4>), dim3(B), dim3(P.waves_per_wg * 64), lds, st, P);
    return hipGetLastError();
}
static bool spec_ok(const EmsPlan &P) { return P.nbOper >= P.nm + 3 && P.nbOper <= 64; }
// two-visits-per-wave build (k_decode_dual): the BASELINE headline configuration
bool ems_has_dual_kernel(const EmsPlan &P)
{
    return spec_ok(P) && P.n_pairs > 0 &&
           ((P.q == 256 && P.dc == 4 && P.nm == 20) || (P.q == 16 && P.dc == 8 && P.nm == 8) || (P.q == 64 && P.dc == 4 && (P.nm == 16 || P.nm == 24)));   // (nm = 24: the reference's own test description, NB_LDPC.c:48-60)
}
int ems_dual_max_waves(int dc) { return dc == 4 ? 4 : EMS_DC8_WAVES; }
size_t ems_dual_wave_lds_bytes(int q, int dc, int nm) { return (size_t)dual_wave_bytes(q, dc, nm); }
size_t ems_dual_wg_lds_bytes(int q, int dc, int nm, int n_pairs, int waves) { return (size_t)dual_header_bytes(q, dc, n_pairs) + (size_t)dual_wave_bytes(q, dc, nm) * waves; }
template <int Q, int NM, int DC> static hipError_t launch_decode_dual_t(const EmsPlan &P, int B, hipStream_t st)
{
    g_kname = kname_fmt("k_decode_dual<%d, %d, %d>", Q, NM, DC);
    const size_t lds = ems_dual_wg_lds_bytes(Q, DC, P.nm, P.n_pairs, P.waves_per_wg);
    hipError_t e = set_max_lds(k_decode_dual<Q, NM, DC>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode_dual<Q, NM, DC>), dim3(B), dim3(P.waves_per_wg * 64), lds, st, P);
    return hipGetLastError();
}
// two frames per wave (k_decode_twin): the tiny GF(64) dc = 4 codes at nm = 16
bool ems_has_twin_kernel(const EmsPlan &P) { return spec_ok(P) && P.q == 64 && P.dc == 4 && P.nm == 16 && P.M <= 64; }
size_t ems_twin_wg_lds_bytes(int q, int dc, int nm) { return (size_t)dual_header_bytes(q, dc, 0) + (size_t)dual_wave_bytes(q, dc, nm); }
template <int Q, int NM, int DC> static hipError_t launch_decode_twin_t(const EmsPlan &P, int B, hipStream_t st)
{
    g_kname = kname_fmt("k_decode_twin<%d, %d, %d>", Q, NM, DC);
    const size_t lds = ems_twin_wg_lds_bytes(Q, DC, NM);
    hipError_t e = set_max_lds(k_decode_twin<Q, NM, DC>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode_twin<Q, NM, DC>), dim3((B + 1) / 2), dim3(64), lds, st, P, B);
    return hipGetLastError();
}
// kernels that also exist in the 16-waves-per-CU build
bool ems_has_occ4_kernel(const EmsPlan &P) { return spec_ok(P) && P.q == 256 && P.dc == 4 && P.nm == 20; }

static hipError_t launch_decode_any(const EmsPlan &P, int B, hipStream_t st);
hipError_t ems_launch_decode(const EmsPlan &P, int B, hipStream_t st, const char **kname)
{
    g_kname = "";
    const hipError_t e = launch_decode_any(P, B, st);
    if (kname) *kname = g_kname;
    return e;
}
static hipError_t launch_decode_any(const EmsPlan &P, int B, hipStream_t st)
{
    const int lpe = ems_lpe_for(P.dc);
    // kernels specialised for the BASELINE configurations (dc, nm compile-time; need nm+3 <= nbOper <= 64)
    const bool spec = spec_ok(P) && P.t_dc == nullptr;       // (irregular check degrees: the generic kernels with per-visit degrees)
    if (P.t_dc != nullptr && (P.twin || P.dual || P.occ4)) return hipErrorInvalidValue;
    if (P.twin) {
        if (!ems_has_twin_kernel(P)) return hipErrorInvalidValue;
        return launch_decode_twin_t<64, 16, 4>(P, B, st);
    }
    if (P.dual) {
        if (!ems_has_dual_kernel(P) || P.waves_per_wg > ems_dual_max_waves(P.dc) || P.wave_lds_bytes != (int)ems_dual_wave_lds_bytes(P.q, P.dc, P.nm))
            return hipErrorInvalidValue;
        if (P.dc == 4 && P.q == 256) return launch_decode_dual_t<256, 20, 4>(P, B, st);
        if (P.dc == 4) return P.nm == 16 ? launch_decode_dual_t<64, 16, 4>(P, B, st) : launch_decode_dual_t<64, 24, 4>(P, B, st);
        return launch_decode_dual_t<16, 8, 8>(P, B, st);
    }
    if (P.occ4 && P.waves_per_wg <= 4 && ems_has_occ4_kernel(P)) return launch_decode_t<256, 16, 4, 20, true>(P, B, st);
    if (spec && P.q == 256 && P.dc == 4 && P.nm == 20) return launch_decode_t<256, 16, 4, 20>(P, B, st);
    if (spec && P.q == 64 && P.dc == 4 && P.nm == 16) return launch_decode_t<64, 16, 4, 16>(P, B, st);
    if (spec && P.q == 16 && P.dc == 8 && P.nm == 8) return launch_decode_t<16, 8, 8, 8>(P, B, st);
    if (spec && P.q == 256 && P.dc == 8 && P.nm == 20) return launch_decode_t<256, 8, 8, 20>(P, B, st);   // the rate-3/4 GF(256) matrices of the reference
    if (spec && P.q == 256 && P.dc == 12 && P.nm == 20) return launch_decode_t<256, 4, 12, 20>(P, B, st);  // ... and the rate-5/6 one
    if (lpe == 16) {
        if (P.q == 256) return launch_decode_t<256, 16>(P, B, st);
        if (P.q == 64) return launch_decode_t<64, 16>(P, B, st);
        if (P.q == 16) return launch_decode_t<16, 16>(P, B, st);
    } else if (lpe == 8) {
        if (P.q == 256) return launch_decode_t<256, 8>(P, B, st);
        if (P.q == 64) return launch_decode_t<64, 8>(P, B, st);
        if (P.q == 16) return launch_decode_t<16, 8>(P, B, st);
    } else if (lpe == 4) {
        if (P.q == 256) return launch_decode_t<256, 4>(P, B, st);   // 64 symbols per lane: correct, register-bound (spills)
        if (P.q == 64) return launch_decode_t<64, 4>(P, B, st);
        if (P.q == 16) return launch_decode_t<16, 4>(P, B, st);
    }
    return hipErrorInvalidValue;
}

hipError_t ems_launch_llr_init(const float *noisy_dev, float sigma, int logq, int q, size_t total,
                               const uint8_t *idx2bin_dev, const uint8_t *bin2idx_dev, float *app_dev, hipStream_t st)
{
    const size_t nvn = total / q;                              // (frame, variable node) pairs
    const int per = 4 * (q >= 64 ? 1 : 64 / q);                // variable nodes per block pass
    size_t blocks = (nvn + per - 1) / per;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_llr_init, dim3((unsigned)blocks), dim3(256), 0, st, noisy_dev, sigma, logq, q, nvn, idx2bin_dev, bin2idx_dev, app_dev);
    return hipGetLastError();
}

hipError_t ems_launch_llr_const(const float *obs_dev, const float *att_dev, const float *table_dev, int q, int D, int mode, float sigma,
                                size_t nvn, const uint8_t *bin2idx_dev, float *app_dev, hipStream_t st)
{
    size_t blocks = (nvn + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_llr_const, dim3((unsigned)blocks), dim3(256), 0, st, obs_dev, att_dev, table_dev, q, D, mode, sigma, nvn, bin2idx_dev, app_dev);
    return hipGetLastError();
}

hipError_t ems_launch_count_errors(const uint8_t *decide_dev, const int *cw_dev, int B, int N, int K,
                                   const uint8_t *idx2bin_dev, int *nberr_dev, hipStream_t st)
{
    hipLaunchKernelGGL(k_count_errors, dim3(B), dim3(256), 0, st, decide_dev, cw_dev, N, K, idx2bin_dev, nberr_dev);
    return hipGetLastError();
}

hipError_t ems_launch_expand_ctov(const uint8_t *rec_dev, int rec_bytes, int nm, int q, int E, float *out_dev,
                                  hipStream_t st)
{
    hipLaunchKernelGGL(k_expand_ctov, dim3(E), dim3(64), 0, st, rec_dev, rec_bytes, nm, q, E, out_dev);
    return hipGetLastError();
}

hipError_t ems_launch_es_unit(const float *in1, const float *in2, const int *idx1, const int *idx2, float *out,
                              int *idxout, int nm, int nbOper, int q, const uint8_t *idx2bin, const uint8_t *bin2idx,
                              hipStream_t st)
{
    hipLaunchKernelGGL(k_es_unit, dim3(1), dim3(64), 0, st, in1, in2, idx1, idx2, out, idxout, nm, nbOper, q, idx2bin,
                       bin2idx);
    return hipGetLastError();
}

hipError_t ems_launch_cn_unit(const EmsPlan &P, const int *coef, const float *vtoc_llr, const int *vtoc_gf, float *ctov,
                              hipStream_t st)
{
    const size_t lds = (((size_t)P.n_lists * list_stride(P.nm) * 8 + 15) & ~(size_t)15) + (((size_t)P.n_lists * 4 + 15) & ~(size_t)15);
    if (P.q == 256) hipLaunchKernelGGL((k_cn_unit<256>), dim3(1), dim3(64), lds, st, P, coef, vtoc_llr, vtoc_gf, ctov);
    else if (P.q == 64) hipLaunchKernelGGL((k_cn_unit<64>), dim3(1), dim3(64), lds, st, P, coef, vtoc_llr, vtoc_gf, ctov);
    else if (P.q == 16) hipLaunchKernelGGL((k_cn_unit<16>), dim3(1), dim3(64), lds, st, P, coef, vtoc_llr, vtoc_gf, ctov);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t ems_launch_decision_unit(const float *app, int N, int q, int *decide, hipStream_t st)
{
    hipLaunchKernelGGL(k_decision_unit, dim3((N + 3) / 4), dim3(256), 0, st, app, N, q, decide);
    return hipGetLastError();
}

hipError_t ems_launch_decide_rows(const float *app_dev, const int *rows_dev, int nrows, int B, int N, int q, uint8_t *decide_dev,
                                  hipStream_t st)
{
    hipLaunchKernelGGL(k_decide_rows, dim3((nrows + 3) / 4, B), dim3(256), 0, st, app_dev, rows_dev, nrows, N, q, decide_dev);
    return hipGetLastError();
}

hipError_t ems_launch_syndrome_unit(const int *row_ptr, const int *col, const int *coef, int M, int q, const int *decide,
                                    const uint8_t *idx2bin, const uint8_t *bin2idx, unsigned *first_bad, hipStream_t st)
{
    hipLaunchKernelGGL(k_syndrome_unit, dim3((M + 255) / 256), dim3(256), 0, st, row_ptr, col, coef, M, q, decide,
                       idx2bin, bin2idx, first_bad);
    return hipGetLastError();
}

hipError_t ems_launch_sort_llr(const float *app_dev, int N, int q, float *s_llr_dev, int *s_gf_dev, hipStream_t st)
{
    hipLaunchKernelGGL(k_sort_llr, dim3(N), dim3(64), (size_t)q * 4, st, app_dev, q, s_llr_dev, s_gf_dev);
    return hipGetLastError();
}

#ifdef EMS_STAMPS
hipError_t ems_debug_counters(unsigned long long *out8, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dbg), 8 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof z);
    }
    return e;
}
#else
hipError_t ems_debug_counters(unsigned long long *out8, int reset)
{
    (void)reset;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    return hipSuccess;
}
#endif

// ---- v2 flavour launchers ----
hipError_t ems_launch_v2_init(const float *like, const int *like_idx, int B, int N, int q, int nm, int E, const int *vn_edge,
                              float *likeS, uint8_t *msg, int rec, int *bad, hipStream_t st)
{
    hipLaunchKernelGGL(k_v2_init, dim3((unsigned)((size_t)B * N)), dim3(64), 0, st, like, like_idx, N, q, nm, E, vn_edge, likeS, msg, rec, bad);
    return hipGetLastError();
}
hipError_t ems_launch_v2_init_rows(const float *likeS, int B, int N, int q, int nm, int E, const int *vn_edge, uint8_t *msg, int rec,
                                   int *bad, hipStream_t st)
{
    hipLaunchKernelGGL(k_v2_init_rows, dim3((unsigned)((size_t)B * N)), dim3(64), 0, st, likeS, N, q, nm, E, vn_edge, msg, rec, bad);
    return hipGetLastError();
}
hipError_t ems_launch_sort_like_desc(const float *likeS, int N, int q, float *s_like, int *s_idx, hipStream_t st)
{
    hipLaunchKernelGGL(k_sort_like_desc, dim3(N), dim3(64), 0, st, likeS, q, s_like, s_idx);
    return hipGetLastError();
}
template <int Q, int LPE, int NM = 0, bool OCC4 = false> static hipError_t launch_decode_v2_t(const EmsPlan &P, int B, hipStream_t st)
{
    g_kname = kname_fmt(OCC4 ? "k_decode_v2<%d, %d, %d, true>" : "k_decode_v2<%d, %d, %d, false>", Q, LPE, NM);
    const size_t lds = ems_wg_lds_bytes(Q, P.M, P.wave_lds_bytes, P.waves_per_wg);
    hipError_t e = set_max_lds(k_decode_v2<Q, LPE, NM, OCC4>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode_v2<Q, LPE, NM, OCC4>), dim3(B), dim3(P.waves_per_wg * 64), lds, st, P);
    return hipGetLastError();
}
bool ems_has_v2_dual_kernel(const EmsPlan &P) { return P.q == 256 && P.dc == 4 && P.nm == 20 && P.nbOper >= 23 && P.nbOper <= 64 && P.n_pairs > 0 && P.p_dst; }
static hipError_t launch_decode_v2_any(const EmsPlan &P0, int B, hipStream_t st);
hipError_t ems_launch_decode_v2(const EmsPlan &P0, int B, hipStream_t st, const char **kname)
{
    g_kname = "";
    const hipError_t e = launch_decode_v2_any(P0, B, st);
    if (kname) *kname = g_kname;
    return e;
}
static hipError_t launch_decode_v2_any(const EmsPlan &P0, int B, hipStream_t st)
{
    if (P0.dual) {
        if (!ems_has_v2_dual_kernel(P0) || P0.waves_per_wg > 4 || P0.wave_lds_bytes != (int)ems_dual_wave_lds_bytes(P0.q, P0.dc, P0.nm)) return hipErrorInvalidValue;
        const size_t lds = ems_dual_wg_lds_bytes(P0.q, P0.dc, P0.nm, P0.n_pairs, P0.waves_per_wg);
        g_kname = "k_decode_v2_dual<256, 20>";
        hipError_t e = set_max_lds(k_decode_v2_dual<256, 20>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_decode_v2_dual<256, 20>), dim3(B), dim3(P0.waves_per_wg * 64), lds, st, P0);
        return hipGetLastError();
    }
    const int lpe = ems_lpe_for(P0.dc);
    if ((lpe != 16 && lpe != 8 && lpe != 4) || P0.dc < 2) return hipErrorInvalidValue;
    EmsPlan P = P0;
    if (P.waves_per_wg > 12) P.waves_per_wg = 12;              // the kernel's launch bound
    const bool base = P.q == 256 && P.nm == 20 && P.nbOper >= 23 && P.nbOper <= 64 && lpe == 16;   // the BASELINE message size, compile-time
    if (base && P.occ4 && P.waves_per_wg <= 4) return launch_decode_v2_t<256, 16, 20, true>(P, B, st);   // 4 frames per CU
    if (base) return launch_decode_v2_t<256, 16, 20>(P, B, st);
    if (P.q == 256 && lpe == 16) return launch_decode_v2_t<256, 16>(P, B, st);
    if (P.q == 256 && lpe == 8 && P.nm == 20 && P.nbOper >= 23 && P.nbOper <= 64) return launch_decode_v2_t<256, 8, 20>(P, B, st);   // rate-3/4 matrices, compile-time nm
    if (P.q == 256 && lpe == 8) return launch_decode_v2_t<256, 8>(P, B, st);
    if (P.q == 64 && lpe == 16) return launch_decode_v2_t<64, 16>(P, B, st);
    if (P.q == 64 && lpe == 8) return launch_decode_v2_t<64, 8>(P, B, st);
    if (P.waves_per_wg > 8) P.waves_per_wg = 8;
    if (P.q == 256 && lpe == 4) return launch_decode_v2_t<256, 4>(P, B, st);   // 64 symbols per lane: register-bound
    if (P.q == 64 && lpe == 4) return launch_decode_v2_t<64, 4>(P, B, st);
    return hipErrorInvalidValue;
}
