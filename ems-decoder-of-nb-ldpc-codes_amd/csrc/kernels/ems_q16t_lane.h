// kernels/ems_q16t_lane.h -- GF(16) / nm = 8 / dc = 8 (BASELINE configs[2]): the check-node visit in LANE-SERIAL form.
//
// k_decode_dual<16, 8, 8> spends ~35 wave-instructions per elementary step: 8 (or 16) lanes co-operate on one step's 48 cells and pay
// for it in butterflies, broadcasts and masked lanes.  Here ONE LANE runs one whole elementary step (bubble_decoder.c:316-593 in its
// "first occurrence per symbol" closed form, see ems_es_q16.inc) and one lane owns one EDGE of a check node for the row-wise work
// (NB_LDPC.c:329-374 and :415-450), as straight-line scalar code.  The only thing a lane cannot do in registers is index by a
// run-time value (symbol -> slot); that goes through a lane-private column of LDS, `scr[slot][lane]`, bank-conflict-free by
// construction.  64 elementary steps cost one pass of this code, ~7 wave-instructions each.
//
// This header is plain C++ on purpose: the same functions compile for the device (k_decode_q16t in ems_q16t.inc, Scr = a column of
// LDS) and for the host (tests/ctest/test_q16t.cpp, Scr = an array), where they are compared with the oracle's ElementaryStep /
// CheckPassLogEMS / decode on the CPU -- the lane code is checked without a GPU.
//
// Exactness.  Every value that reaches an output is produced by the reference's own float operations (In1[r] + In2[c], Mvc - min,
// LLR[Stp-1] + offset, Mcv + Mvc).  ORDER is decided on 32-bit keys (value bits with the low four bits replaced by the symbol): two
// candidates within 16 ulp of each other -- exact ties included -- cannot be told apart that way, and a lane that sees such a pair
// among the entries that matter, or any other situation the closed form does not decide (listed at each `bad |=`), reports `bad`:
// the kernel then redoes that check node with the merge-form code of k_decode (process_cn), which follows the reference's rules
// literally.  Wrong answers are impossible by construction as long as `bad` is conservative; how OFTEN it fires only costs time.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)          // (the other branch is the HOST compile of tests/ctest/test_q16t.cpp with g++)
#define Q16T_HD __host__ __device__ __forceinline__
#else
#define Q16T_HD inline
#endif

namespace q16t {

constexpr uint32_t LIM = 0x47C34FFFu;        // largest float below 1e5f (0x47C35000): valid values are <= LIM as bit patterns
constexpr uint32_t SENT = 0x47C35000u;       // 1e5f, the reference's "empty" value (bubble_decoder.c:370-374)
constexpr uint32_t INF = 0x7F800000u;

Q16T_HD uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
Q16T_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
Q16T_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
Q16T_HD uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// One message list of a check node: 8 values in ascending order (entries past the length hold 1e5) and the 8 symbols as nibbles
// (entry k in bits 4k..4k+3; inside the check node symbols are binary images, GF addition = XOR).
struct List {
    float v[8];
    uint32_t s;
};

// ---- sorting networks on 32-bit keys (verified exhaustively by the 0-1 principle in tests/ctest/test_q16t.cpp) ----
#define Q16T_CE(a, b) do { const uint32_t lo_ = umin(k[a], k[b]), hi_ = umax(k[a], k[b]); k[a] = lo_; k[b] = hi_; } while (0)
Q16T_HD void sort8(uint32_t *k)     // optimal 19-comparator network
{
    Q16T_CE(0, 1); Q16T_CE(2, 3); Q16T_CE(4, 5); Q16T_CE(6, 7);
    Q16T_CE(0, 2); Q16T_CE(1, 3); Q16T_CE(4, 6); Q16T_CE(5, 7);
    Q16T_CE(1, 2); Q16T_CE(5, 6); Q16T_CE(0, 4); Q16T_CE(3, 7);
    Q16T_CE(1, 5); Q16T_CE(2, 6);
    Q16T_CE(1, 4); Q16T_CE(3, 6);
    Q16T_CE(2, 4); Q16T_CE(3, 5);
    Q16T_CE(3, 4);
}
// k[0..15] -> k[0..7] = the 8 smallest in ascending order; returns the 9th smallest
Q16T_HD uint32_t top8of16(uint32_t *k)
{
    sort8(k);
    sort8(k + 8);
    uint32_t h = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 4; ++i) {              // the 8 smallest = min(A[i], B[7-i]) (a bitonic sequence), the rest = the max's
        const uint32_t a0 = k[i], b0 = k[15 - i], a1 = k[7 - i], b1 = k[8 + i];
        k[i] = umin(a0, b0);
        k[7 - i] = umin(a1, b1);
        h = umin(h, umin(umax(a0, b0), umax(a1, b1)));
    }
    Q16T_CE(0, 4); Q16T_CE(1, 5); Q16T_CE(2, 6); Q16T_CE(3, 7);      // bitonic merge of 8
    Q16T_CE(0, 2); Q16T_CE(1, 3); Q16T_CE(4, 6); Q16T_CE(5, 7);
    Q16T_CE(0, 1); Q16T_CE(2, 3); Q16T_CE(4, 5); Q16T_CE(6, 7);
    return h;
}
#undef Q16T_CE

// ---- branch-free helpers: on gfx950 a compare that feeds a select costs two instructions plus wait states (the result travels through
//      an SGPR pair), so validity, counts and selections are done with sign bits, masks and min / max instead ----
Q16T_HD uint32_t shl1_sign(uint32_t acc, uint32_t x)      // acc << 1 | sign bit of x
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(acc, x, 31);
#else
    return (acc << 1) | (x >> 31);
#endif
}
Q16T_HD uint32_t sign_mask(uint32_t x) { return (uint32_t)((int32_t)x >> 31); }             // all ones when bit 31 is set
Q16T_HD uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (mask & a) | (~mask & b); }   // mask ? a : b, bit by bit (v_bfi_b32)
Q16T_HD int popc(uint32_t x) { return __builtin_popcount(x); }
#if defined(__HIP_DEVICE_COMPILE__) && defined(Q16T_MARKS)
#define Q16T_MARK(n) asm volatile("; Q16T_MARK " #n)
#else
#define Q16T_MARK(n) do { } while (0)
#endif

// two of the sorted keys s[0..7], s8 have the same value bits above the symbol nibble (closer than 16 ulp, or equal): their order is
// not decided here.  Invalid keys (all ones) never match a valid one (valid value bits are <= LIM); an invalid s[j] is kept from
// matching the invalid key behind it by its own sign bit.  Returns the smallest difference (a tie iff < 16).
Q16T_HD uint32_t near_tie9(const uint32_t *s, uint32_t s8)
{
    uint32_t m = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 7; ++j) m = umin(m, (s[j] ^ s[j + 1]) | ((s[j] >> 27) & 16u));
    m = umin(m, (s[7] ^ s8) | ((s[7] >> 27) & 16u));
    return m;
}

// The L-shaped region of an elementary step: rows 0..3 x columns 0..7 and rows 4..7 x columns 0..3 (nm = 8: 48 cells).  With NbOper = 16
// a scan never reaches a cell with (r + 1)(c + 1) > 16 -- the cells it dominates come first -- unless exact ties reorder them (checked
// where it matters): SMALL restricts a loop to the 34 cells that can be reached.
template <bool SMALL> Q16T_HD constexpr bool cell_in(int r, int c) { return c < (r < 4 ? 8 : 4) && (!SMALL || (r + 1) * (c + 1) <= 16); }

// number of cells whose value In1[r] + In2[c] is <= T (STRICT: < T); T a valid value (0 <= T < 1e5); no NaN / inf among the sums
template <bool SMALL, bool STRICT> Q16T_HD int count_cells(const List &A, const List &B, float T)
{
    uint32_t acc0 = 0, acc1 = 0;
    int n = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (cell_in<SMALL>(r, c)) {
                const float v = A.v[r] + B.v[c];
                const uint32_t d = f2u(STRICT ? v - T : T - v);            // STRICT: negative iff v < T; else negative iff v > T
                if (n < 32) acc0 = shl1_sign(acc0, d); else acc1 = shl1_sign(acc1, d);
                ++n;
            }
    const int neg = popc(acc0) + popc(acc1);
    return STRICT ? neg : n - neg;
}

// ------------------------------------------------------------------------------------------------
// ElementaryStep (bubble_decoder.c:316-593) for GF(16), nm = 8, by ONE lane.
// bins: this lane's 16-slot scratch column, all slots INF on entry and on exit (Scr: ld(i), st(i, u), amin(i, u), sync()).
// Closed form (ems_es_q16.inc has the derivation): bin s = smallest value of a cell with symbol s = the first occurrence of s in the
// scan; the scan ends after the cheapest boundary cell (row 7 / column 7), value K; the output is the first 8 bins <= K in ascending
// order, cut by NbOper (the scan's NbOper-th cell).  Returns true when the closed form does not decide this task (see header).
// SMALL16: NbOper == 16 known at compile time -- only the 34 reachable cells are binned and counted.
// ------------------------------------------------------------------------------------------------
template <bool SMALL16, typename Scr> Q16T_HD bool es_lane(const List &A, const List &B, int nbOper, Scr &bins, List &out)
{
    // ---- the cells: value In1[r] + In2[c] (:386-423), symbol = XOR of the binary images; first occurrence per symbol ----
    Q16T_MARK(es_cells);
    uint32_t sa[8], sb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sa[i] = (A.s >> (4 * i)) & 15u; sb[i] = (B.s >> (4 * i)) & 15u; }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (cell_in<SMALL16>(r, c)) bins.amin(sa[r] ^ sb[c], f2u(A.v[r] + B.v[c]));
    // the cell that ends the scan (:506): the cheapest of row 7 and column 7 -- lists ascend and float addition is monotone, so it is
    // (7, 0) or (0, 7); `second` = the next candidates, for the tie test below
    const uint32_t kA = f2u(A.v[7] + B.v[0]), kB = f2u(A.v[0] + B.v[7]);
    const uint32_t kstop = umin(kA, kB);
    const uint32_t second = umin(umax(kA, kB), umin(f2u(A.v[7] + B.v[1]), f2u(A.v[1] + B.v[7])));
    const uint32_t klim = umin(kstop, LIM);                                   // (cells >= 1e5 involve an empty source entry, :478)
    bins.sync();
    Q16T_MARK(es_keys);
    uint32_t k[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) k[i] = (bins.ld(i) & ~15u) | (uint32_t)i;   // (a bin above K, or empty, sorts behind every valid one as it is)
    Q16T_MARK(es_sort);
    const uint32_t s8 = top8of16(k);
    Q16T_MARK(es_post);
    const uint32_t kmax = klim | 15u;                                         // a sorted key is valid (its bin <= K) iff it is <= kmax
    uint32_t vm[8], inv = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { vm[j] = sign_mask(kmax - k[j]); inv = shl1_sign(inv, kmax - k[j]); }   // vm: all ones = not valid
    int S = 8 - popc(inv);                                                    // valid among the eight smallest = min(D, 8)
    // two first occurrences whose order the keys do not decide: same value bits above the symbol nibble (an invalid key is kept from
    // matching the one behind it)
    uint32_t tie = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 7; ++j) tie = umin(tie, (k[j] ^ k[j + 1]) | (vm[j] & 16u));
    tie = umin(tie, (k[7] ^ s8) | (vm[7] & 16u));
    // exact values of the outputs (the keys lost four bits): the bins, indexed by symbol
    uint32_t ov[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ov[j] = bfi(vm[j], SENT, bins.ld((int)(k[j] & 15u)));
    bins.sync();
#pragma unroll
    for (int i = 0; i < 16; ++i) bins.st(i, INF);
    // ---- another cell with the value K that ends the scan: the order of the two would matter when the other one is the first occurrence
    //      of another symbol (its bin then holds K).  Conservative, on the keys: two boundary cells within 16 ulp of K, or a valid bin
    //      within 16 ulp of K that is not the boundary cell's own symbol.  (More than eight bins below K: the scan ends on its eighth
    //      output, strictly below the ninth bin unless `tie` says otherwise.)  No boundary cell below 1e5: no such stop. ----
    Q16T_MARK(es_ktest);
    tie = umin(tie, (second ^ kstop) | (((LIM - kstop) >> 27) & 16u));
    const uint32_t symb = bfi(sign_mask(kB - kA), (A.s ^ (B.s >> 28)) & 15u, ((A.s >> 28) ^ B.s) & 15u);   // kB < kA: (0, 7) ends the scan, else (7, 0)
    const uint32_t kk = (kstop & ~15u) | symb;
    uint32_t other = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 8; ++j) other = umin(other, (k[j] ^ kk) - 1u);             // 0 .. 14: same value bits as K, another symbol
    bool bad = tie < 16u || other < 15u;
    // ---- NbOper (:472): the scan visits N = #{cells <= the last emitted value (or K)} cells; beyond NbOper it stops on its NbOper-th
    //      cell, before any other stop: what it has emitted by then are the outputs with at most NbOper cells up to and including their
    //      own -- a prefix.  A cell with the value of the first entry left out could fall on either side of the cut when fewer than
    //      NbOper cells are strictly below it: undecided. ----
    Q16T_MARK(es_count);
    const uint32_t xcut = bfi(vm[7], klim, ov[7]);
    // SMALL16: vb = the cheapest cell that cannot be reached ((2,5), (3,4), (4,3), (5,2) dominate at least 17 cells each, every other
    // such cell is at least one of them): a threshold at or above it has more than 16 cells below it
    const uint32_t vb = SMALL16 ? umin(umin(f2u(A.v[2] + B.v[5]), f2u(A.v[3] + B.v[4])), umin(f2u(A.v[4] + B.v[3]), f2u(A.v[5] + B.v[2]))) : INF;
    auto cells_le = [&](uint32_t t) { return count_cells<SMALL16, false>(A, B, u2f(t)) + (SMALL16 && vb <= t ? 64 : 0); };
    if (cells_le(xcut) > nbOper) {
        Q16T_MARK(es_ovf);
        int lo = 0, hi = S;                                                   // S2 = #{r < S: count(ov[r]) <= NbOper}; counts ascend with r
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
            const int mid = (lo + hi) >> 1;
            uint32_t t = ov[0];
#pragma unroll
            for (int r = 1; r < 8; ++r) t = mid == r ? ov[r] : t;
            const bool fits = lo < hi && cells_le(t) <= nbOper;
            const bool over = lo < hi && !fits;
            lo = fits ? mid + 1 : lo;
            hi = over ? mid : hi;
        }
        const int S2 = lo;
        if (S2 < S) {
            uint32_t t = ov[0];
#pragma unroll
            for (int r = 1; r < 8; ++r) t = S2 == r ? ov[r] : t;
            bad = bad || count_cells<SMALL16, true>(A, B, u2f(t)) + (SMALL16 && vb < t ? 64 : 0) < nbOper;
        }
        // a cell that "cannot be reached" can be, when exact ties among the cells it dominates let it come early: fewer than NbOper
        // cells strictly below the cheapest of them, and the scan gets that far
        if (SMALL16 && vb <= xcut) bad = bad || count_cells<true, true>(A, B, u2f(vb)) < nbOper;
        S = S2;
    }
    Q16T_MARK(es_out);
    uint32_t os = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t keep = sign_mask((uint32_t)(j - S));                   // j < S
        out.v[j] = u2f(bfi(keep, ov[j], SENT));                               // :370-374 shows through behind the S entries
        os |= (keep & k[j] & 15u) << (4 * j);
    }
    out.s = os;
    return bad || S == 0;                                                     // (S = 0: the reference would index LLR[-1], :264; never seen)
}

// ------------------------------------------------------------------------------------------------
// Variable-node side of one edge, by ONE lane: top-8 of the 16 Mvc values in ascending order, ties -> lowest symbol
// (NB_LDPC.c:354-369), normalised to the smallest (:372-373) and rotated by the edge's coefficient (bubble_decoder.c:133-152).
// mv[g]: Mvc by symbol index.  rot: nibble table, symbol index -> binary image of (symbol x coefficient) (rot_table()).
// scr: this lane's 16-slot scratch column (contents irrelevant on entry, undefined on exit).
// Returns true (undecided here) when some value is not in [0, 1e5) as a bit pattern (negative, -0, NaN, >= 1e5: the literal rules of
// :358-366 apply), when the order of two of the nine smallest is not decided by the keys, or when a normalised value reaches 1e4 (the
// closed form of the elementary steps needs sums below 1e5).
// ------------------------------------------------------------------------------------------------
template <typename Scr> Q16T_HD bool select_lane(const float *mv, uint64_t rot, Scr &scr, List &out)
{
    uint32_t k[16], mx = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const uint32_t u = f2u(mv[g]);
        mx = umax(mx, u);
        k[g] = (u & ~15u) | (uint32_t)g;
        scr.st(g, u);
    }
    bool bad = mx > LIM;
    const uint32_t s8 = top8of16(k);
    bad = bad || near_tie9(k, s8) < 16u;
    scr.sync();
    uint32_t os = 0;
    float v0 = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t g = k[j] & 15u;
        const float v = u2f(scr.ld((int)g));
        if (j == 0) v0 = v;
        out.v[j] = j == 0 ? 0.0f : v - v0;
        os |= (uint32_t)((rot >> (4 * g)) & 15u) << (4 * j);
    }
    out.s = os;
    bad = bad || !(out.v[7] < 1e4f);
    scr.sync();
    return bad;
}

// ------------------------------------------------------------------------------------------------
// Check-node side of one edge, by ONE lane: the output list of the edge -> dense C->V row with the default LLR[Stp-1] + offset
// (bubble_decoder.c:231-278), APP = Mcv + Mvc (NB_LDPC.c:444-450), Decision (tools.c:312-330).
// O: the edge's output list (binary images).  irot: nibble table, binary image -> symbol index of (symbol / coefficient).
// Outputs: napp[16], the record's fields (values as stored, symbols as symbol indices packed 4 bits each, default, count) and the
// decision.  scr: scratch column.
// ------------------------------------------------------------------------------------------------
struct EdgeOut {
    float napp[16];
    float val[8];
    uint32_t syms;       // symbol index of entry k in bits 4k..4k+3
    float dflt;
    int cnt, decide;
};
template <typename Scr> Q16T_HD void output_lane(const List &O, const float *mvc, float offset, uint64_t irot, Scr &scr, EdgeOut &e)
{
    uint32_t inv = 0, last = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t u = f2u(O.v[j]), im = sign_mask(LIM - u);               // all ones behind the valid prefix (1e5)
        inv = shl1_sign(inv, LIM - u);
        last = umax(last, u & ~im);                                            // ascending, non-negative: the largest valid one is LLR[Stp-1]
    }
    const int S = 8 - popc(inv);
    e.cnt = S;
    e.dflt = u2f(last) + offset;
#pragma unroll
    for (int g = 0; g < 16; ++g) scr.st(g, f2u(e.dflt));
    scr.sync();
    uint32_t ps = 0;
#pragma unroll
    for (int j = 7; j >= 0; --j) {                                             // descending: an entry behind the prefix (symbol nibble 0) writes the
        const uint32_t g = (uint32_t)((irot >> (4 * ((O.s >> (4 * j)) & 15u))) & 15u);   // default first, the valid entries overwrite it
        const uint32_t u = f2u(O.v[j]);
        e.val[j] = O.v[j];
        ps |= g << (4 * j);
        scr.st((int)g, bfi(sign_mask(LIM - u), f2u(e.dflt), u));
    }
    e.syms = ps;
    scr.sync();
    float lv = 0.0f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        e.napp[g] = u2f(scr.ld(g)) + mvc[g];
        lv = g == 0 ? e.napp[0] : (e.napp[g] < lv ? e.napp[g] : lv);
    }
    int lg = 15;
#pragma unroll
    for (int g = 14; g >= 0; --g) lg = e.napp[g] == lv ? g : lg;                // first strict minimum = the lowest symbol that attains it
    e.decide = lv < 1e5f ? lg : 0;
    scr.sync();
}

// ---- GF(16) helper tables (symbol index 0 = zero, i >= 1 = alpha^(i-1); init.c:65-130) as 16 nibbles in a 64-bit word ----
// rot_table: symbol index g -> binary image of g x alpha^lh;  irot_table: binary image b -> symbol index of b / alpha^lh
inline uint64_t rot_table(const uint8_t *i2b, int lh)
{
    uint64_t t = 0;
    for (int g = 1; g < 16; ++g) t |= (uint64_t)i2b[(g - 1 + lh) % 15 + 1] << (4 * g);
    return t;
}
inline uint64_t irot_table(const uint8_t *b2i, int lh)
{
    uint64_t t = 0;
    for (int b = 1; b < 16; ++b) t |= (uint64_t)((b2i[b] - 1 - lh + 15) % 15 + 1) << (4 * b);
    return t;
}

// The forward / backward / merge schedule of a dc = 8 check node (bubble_decoder.c:166-227) on 14 list buffers.  A stage reads all its
// operands before it writes, so a result may take the buffer of a list whose last reader is that same stage:
//   inputs in(t) = 0..7; merge M(k) overwrites in(k + 1) (both chains have passed that input by the stage that makes M(k));
//   prefixes P(1) P(2) P(3) = 8 9 10, then P(4) -> 9 (P(2) is last read by M(2) in stage 4), P(5) -> 10, P(6) -> 8;
//   suffixes S(6) S(5) S(4) = 11 12 13, then S(3) -> 12 (S(5) is last read by M(3) in stage 4), S(2) -> 13, S(1) -> 11.
// Output of edge 0 = S(1), of edge 7 = P(6), of edge t = M(t-1).  (tests/ctest/test_q16t.cpp runs whole decodes through this table.)
constexpr int NBUF = 14;
Q16T_HD int buf_P(int k) { return k == 0 ? 0 : (k == 1 || k == 6 ? 8 : (k == 2 || k == 4 ? 9 : 10)); }
Q16T_HD int buf_S(int k) { return k == 7 ? 7 : (k == 6 || k == 1 ? 11 : (k == 5 || k == 3 ? 12 : 13)); }
Q16T_HD int buf_out(int t) { return t == 0 ? buf_S(1) : (t == 7 ? buf_P(6) : t); }
// task `j` (0: forward, 1: backward, 2 / 3: merges) of stage t = 1..6: operands a, b and destination; false = no such task
Q16T_HD bool stage_task(int t, int j, int &a, int &b, int &o)
{
    if (j == 0) { a = buf_P(t - 1); b = t; o = buf_P(t); return true; }
    if (j == 1) { a = buf_S(8 - t); b = 7 - t; o = buf_S(7 - t); return true; }
    if (t < 4) return false;
    const int kk = j == 2 ? 6 - t : t - 1;                                     // lowest k first, as cn_task: k2 = dc-2-t, k1 = t-1
    a = buf_P(kk); b = buf_S(kk + 2); o = kk + 1;
    return true;
}

} // namespace q16t
