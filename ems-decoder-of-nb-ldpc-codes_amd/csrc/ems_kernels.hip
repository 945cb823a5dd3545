// ems_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels of the EMS NB-LDPC hot path.
//
// Mapping (wave64 throughout, no MFMA: this is compare/select/merge work, HBM- and VALU-bound):
//   * one WORKGROUP per frame, persistent over the whole decode (all sweeps): frames are independent, so there is no
//     inter-workgroup communication at all; inside a sweep the check nodes are claimed from an LDS counter in a
//     topological order of the layered schedule's dependency DAG (NB_LDPC.c:320 order is preserved per variable node)
//     and a visit waits only for the visits it depends on (done flags in LDS); sweeps are separated by __syncthreads();
//   * k_decode: one WAVEFRONT per check-node visit; the 64 lanes are split into dc "edge rows" of LPE lanes (LPE = 16 for
//     dc <= 4, 8 for dc <= 8, 4 up to 16; above 16 k_decode_wide walks the edges in passes of 64 / LPE rows); k_decode_dual (the throughput form, BASELINE message sizes): TWO visits per
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
// Layout of this translation unit: the device code lives in kernels/*.inc, included below inside one anonymous namespace (lanes, es,
// es_q16, cn, decode, v2, channel, units -- each needs the ones before it); the host-side launchers follow at the end of this file.
//
// Bit-exactness: only f32 add/sub/compare on the decode path, compiled with -ffp-contract=off, all
// tie-breaks (lowest symbol / lowest competitor) and the 1e5 sentinel comparisons as in the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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
#elif defined(EMS_MARKS)     // listing aid (hipcc -S -DEMS_MARKS; scripts/asm_marks.py): the phase boundaries as comments in the ISA
#define STAMP(i) asm volatile("; Q16T_MARK phase" #i)
#define DBG_ADD(i, v) do { } while (0)
#define ST_PARAMS
#define ST_ARGS
#define ST_DUMMY
#else
#define STAMP(i) do { } while (0)
#define DBG_ADD(i, v) do { } while (0)
#define ST_PARAMS
#define ST_ARGS
#define ST_DUMMY
#endif

#ifdef EMS_MARKS
#define EMS_MARK(x) asm volatile("; Q16T_MARK " #x)
#else
#define EMS_MARK(x) do { } while (0)
#endif

#include "kernels/ems_lanes.inc"
#include "kernels/ems_es.inc"
#include "kernels/ems_es_q16.inc"
#include "kernels/ems_cn.inc"
#include "kernels/ems_decode.inc"
#include "kernels/ems_pair16.inc"
#include "kernels/ems_q16t_lane.h"
#include "kernels/ems_q16t.inc"
#include "kernels/ems_v2.inc"
#include "kernels/ems_channel.inc"
#include "kernels/ems_units.inc"

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
    const size_t dense = (size_t)(64 / lpe) * q * 4;                     // staging rows (the elementary steps' dedupe tables alias them)
    const size_t lists = (size_t)row_list_area_bytes(lpe, dc, nm, n_lists);   // the visit's lists, the selection's survivor scratch aliased behind the inputs
    const size_t lens = ((size_t)n_lists * 4 + 15) & ~(size_t)15;
    return dense + lists + lens + ((ES_PICK_BYTES + 15) & ~15); // + pick bytes of the elementary steps
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
    static const size_t pad = getenv("EMS_LDS_PAD") ? (size_t)atoi(getenv("EMS_LDS_PAD")) : 0;   // developer knob: unused LDS per workgroup = fewer workgroups per CU (occupancy A/B, profiles/NOTES.md round 5)
    const size_t lds = ems_dual_wg_lds_bytes(Q, DC, P.nm, P.n_pairs, P.waves_per_wg) + pad;
    hipError_t e = set_max_lds(k_decode_dual<Q, NM, DC>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode_dual<Q, NM, DC>), dim3(B), dim3(P.waves_per_wg * 64), lds, st, P);
    return hipGetLastError();
}
// the same pairs at sixteen waves per CU (k_decode_pair16): the headline configuration
bool ems_has_pair16_kernel(const EmsPlan &P) { return spec_ok(P) && P.n_pairs > 0 && P.p_pack != nullptr && P.q == 256 && P.dc == 4 && P.nm == 20 && P.t_dc == nullptr; }
size_t ems_pair16_wave_lds_bytes(int q, int nm) { return (size_t)pair16_wave_bytes(q, nm); }
size_t ems_pair16_wg_lds_bytes(int q, int nm, int n_pairs, int waves) { return (size_t)dual_header_bytes(q, 4, n_pairs) + (size_t)pair16_wave_bytes(q, nm) * waves; }
static hipError_t launch_decode_pair16(const EmsPlan &P, int B, hipStream_t st)
{
    g_kname = "k_decode_pair16<256, 20>";
    const size_t lds = ems_pair16_wg_lds_bytes(P.q, P.nm, P.n_pairs, P.waves_per_wg);
    hipError_t e = set_max_lds(k_decode_pair16<256, 20>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode_pair16<256, 20>), dim3(B), dim3(P.waves_per_wg * 64), lds, st, P);
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
// GF(16) / dc = 8 / nm = 8 in lane-serial form (k_decode_q16t)
bool ems_has_q16t_kernel(const EmsPlan &P) { return P.q == 16 && P.dc == 8 && P.nm == 8 && P.nbOper >= 1 && P.n_groups > 0 && P.q_cn != nullptr && P.t_dc == nullptr; }
int ems_q16t_max_waves() { return EMS_Q16T_WAVES; }
size_t ems_q16t_wg_lds_bytes(int n_groups, int waves) { return (size_t)q16t_header_bytes(n_groups) + (size_t)q16t_wave_bytes() * waves; }
template <bool SMALL16, bool TRAV> static hipError_t launch_decode_q16t_t(const EmsPlan &P, int B, hipStream_t st)
{
    g_kname = TRAV ? (SMALL16 ? "k_decode_q16t<true, true>" : "k_decode_q16t<false, true>") : (SMALL16 ? "k_decode_q16t<true, false>" : "k_decode_q16t<false, false>");   // (as the profiler prints them)
    const size_t lds = ems_q16t_wg_lds_bytes(P.n_groups, P.waves_per_wg);
    hipError_t e = set_max_lds(k_decode_q16t<SMALL16, TRAV>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode_q16t<SMALL16, TRAV>), dim3(B), dim3(P.waves_per_wg * 64), lds, st, P);
    return hipGetLastError();
}
static hipError_t launch_decode_q16t(const EmsPlan &P, int B, hipStream_t st)
{
    if (P.q16t_trav) {                              // the travelling layout: its buffers and the per-edge words must be there
        if (!P.q_vo || !P.q_trav || !P.q_scr || P.N > (1 << 14) || P.E > (1 << 15)) return hipErrorInvalidValue;
        return P.nbOper == 16 ? launch_decode_q16t_t<true, true>(P, B, st) : launch_decode_q16t_t<false, true>(P, B, st);
    }
    return P.nbOper == 16 ? launch_decode_q16t_t<true, false>(P, B, st) : launch_decode_q16t_t<false, false>(P, B, st);
}
// the state of a decode on the travelling layout back in `app` / `ctov` (k_q16t_export)
hipError_t ems_launch_q16t_export(const EmsPlan &P, int B, hipStream_t st)
{
    if (!P.q_vo || !P.q_trav || !P.q_cn || P.n_groups <= 0 || B <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_q16t_export, dim3((P.n_groups * 128 + 255) / 256, B), dim3(256), 0, st, P);
    return hipGetLastError();
}
// check degree above 16 (k_decode_wide)
int ems_wide_lpe(int q) { return wide_lpe(q); }
int ems_wide_max_waves(int q) { return wide_max_waves(q); }
size_t ems_wide_wave_lds_bytes(int q, int dc, int nm, int nbuf) { return (size_t)wide_wave_bytes(q, dc, nm, nbuf); }
template <int Q, int LPE, int NM = 0> static hipError_t launch_decode_wide_t(const EmsPlan &P, int B, hipStream_t st)
{
    g_kname = NM > 0 ? kname_fmt("k_decode_wide<%d, %d, %d>", Q, LPE, NM) : kname_fmt("k_decode_wide<%d, %d>", Q, LPE);
    const size_t lds = ems_wg_lds_bytes(Q, P.M, P.wave_lds_bytes, P.waves_per_wg);
    hipError_t e = set_max_lds(k_decode_wide<Q, LPE, NM>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode_wide<Q, LPE, NM>), dim3(B), dim3(P.waves_per_wg * 64), lds, st, P);
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
    if (P.q16t) {
        if (!ems_has_q16t_kernel(P) || P.twin || P.dual || P.occ4 || P.wide || P.waves_per_wg < 1 || P.waves_per_wg > EMS_Q16T_WAVES) return hipErrorInvalidValue;
        return launch_decode_q16t(P, B, st);
    }
    if (P.wide) {
        if (P.twin || P.dual || P.occ4 || !P.t_dc || !P.t_e0 || P.waves_per_wg > wide_max_waves(P.q) || !P.wide_buf || P.wide_nbuf < P.dc ||
            P.wave_lds_bytes != (int)ems_wide_wave_lds_bytes(P.q, P.dc, P.nm, P.wide_nbuf))
            return hipErrorInvalidValue;
        if (P.q == 256 && P.nm == 20 && spec_ok(P)) return launch_decode_wide_t<256, EMS_WIDE_LPE256, 20>(P, B, st);   // the reference's GF(256) message size, compile-time
        if (P.q == 256) return launch_decode_wide_t<256, EMS_WIDE_LPE256>(P, B, st);
        if (P.q == 64) return launch_decode_wide_t<64, 4>(P, B, st);
        if (P.q == 16) return launch_decode_wide_t<16, 4>(P, B, st);
        return hipErrorInvalidValue;
    }
    const int lpe = ems_lpe_for(P.dc);
    // kernels specialised for the BASELINE configurations (dc, nm compile-time; need nm+3 <= nbOper <= 64)
    const bool spec = spec_ok(P) && P.t_dc == nullptr;       // (irregular check degrees: the generic kernels with per-visit degrees)
    if (P.t_dc != nullptr && (P.twin || P.dual || P.occ4)) return hipErrorInvalidValue;
    if (P.twin) {
        if (!ems_has_twin_kernel(P)) return hipErrorInvalidValue;
        return launch_decode_twin_t<64, 16, 4>(P, B, st);
    }
    if (P.pair16) {
        if (!ems_has_pair16_kernel(P) || P.dual || P.twin || P.occ4 || P.waves_per_wg < 1 || P.waves_per_wg > 2 ||
            P.wave_lds_bytes != (int)ems_pair16_wave_lds_bytes(P.q, P.nm))
            return hipErrorInvalidValue;
        return launch_decode_pair16(P, B, st);
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
        if (P.q == 256) return P.nm > 40 ? launch_decode_t<256, 16, 0, -1>(P, B, st) : launch_decode_t<256, 16>(P, B, st);
        if (P.q == 64) return P.nm > 40 ? launch_decode_t<64, 16, 0, -1>(P, B, st) : launch_decode_t<64, 16>(P, B, st);
        if (P.q == 16) return P.nm > 40 ? launch_decode_t<16, 16, 0, -1>(P, B, st) : launch_decode_t<16, 16>(P, B, st);
    } else if (lpe == 8) {
        if (P.q == 256) return P.nm > 40 ? launch_decode_t<256, 8, 0, -1>(P, B, st) : launch_decode_t<256, 8>(P, B, st);
        if (P.q == 64) return P.nm > 40 ? launch_decode_t<64, 8, 0, -1>(P, B, st) : launch_decode_t<64, 8>(P, B, st);
        if (P.q == 16) return P.nm > 40 ? launch_decode_t<16, 8, 0, -1>(P, B, st) : launch_decode_t<16, 8>(P, B, st);
    } else if (lpe == 4) {
        if (P.q == 256) return P.nm > 40 ? launch_decode_t<256, 4, 0, -1>(P, B, st) : launch_decode_t<256, 4>(P, B, st);   // 64 symbols per lane: correct, register-bound (spills)
        if (P.q == 64) return P.nm > 40 ? launch_decode_t<64, 4, 0, -1>(P, B, st) : launch_decode_t<64, 4>(P, B, st);
        if (P.q == 16) return P.nm > 40 ? launch_decode_t<16, 4, 0, -1>(P, B, st) : launch_decode_t<16, 4>(P, B, st);
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
    hipError_t e = hipSuccess;                                  // (large degrees: the lists of one check node can pass the default 64 KiB)
    if (P.q == 256) { if ((e = set_max_lds(k_cn_unit<256>, lds)) == hipSuccess) hipLaunchKernelGGL((k_cn_unit<256>), dim3(1), dim3(64), lds, st, P, coef, vtoc_llr, vtoc_gf, ctov); }
    else if (P.q == 64) { if ((e = set_max_lds(k_cn_unit<64>, lds)) == hipSuccess) hipLaunchKernelGGL((k_cn_unit<64>), dim3(1), dim3(64), lds, st, P, coef, vtoc_llr, vtoc_gf, ctov); }
    else if (P.q == 16) { if ((e = set_max_lds(k_cn_unit<16>, lds)) == hipSuccess) hipLaunchKernelGGL((k_cn_unit<16>), dim3(1), dim3(64), lds, st, P, coef, vtoc_llr, vtoc_gf, ctov); }
    else return hipErrorInvalidValue;
    if (e != hipSuccess) return e;
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
size_t ems_v2wide_wave_lds_bytes(int q, int dc, int nm, int nbuf) { return (size_t)v2wide_wave_bytes(q, dc, nm, nbuf); }
template <int Q, int LPE, int NM = 0> static hipError_t launch_decode_v2_wide_t(const EmsPlan &P, int B, hipStream_t st)
{
    g_kname = NM > 0 ? kname_fmt("k_decode_v2_wide<%d, %d, %d>", Q, LPE, NM) : kname_fmt("k_decode_v2_wide<%d, %d>", Q, LPE);
    const size_t lds = ems_wg_lds_bytes(Q, P.M, P.wave_lds_bytes, P.waves_per_wg);
    hipError_t e = set_max_lds(k_decode_v2_wide<Q, LPE, NM>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode_v2_wide<Q, LPE, NM>), dim3(B), dim3(P.waves_per_wg * 64), lds, st, P);
    return hipGetLastError();
}
static hipError_t launch_decode_v2_any(const EmsPlan &P0, int B, hipStream_t st)
{
    if (P0.t_dc != nullptr) {              // per-visit degrees: irregular (2, dc) codes and check degrees above 16 (k_decode_v2_wide)
        if (P0.dual || !P0.t_e0 || !P0.cn_ptr || !P0.wide_buf || P0.wide_nbuf < P0.dc || P0.waves_per_wg < 1 || P0.waves_per_wg > wide_max_waves(P0.q) ||
            P0.dc < 2 || P0.dc > EMS_MAX_DC || P0.wave_lds_bytes != (int)ems_v2wide_wave_lds_bytes(P0.q, P0.dc, P0.nm, P0.wide_nbuf))
            return hipErrorInvalidValue;
        if (P0.q == 256 && P0.nm == 20 && P0.nbOper >= 23 && P0.nbOper <= 64) return launch_decode_v2_wide_t<256, EMS_WIDE_LPE256, 20>(P0, B, st);
        if (P0.q == 256) return launch_decode_v2_wide_t<256, EMS_WIDE_LPE256>(P0, B, st);
        if (P0.q == 64) return launch_decode_v2_wide_t<64, 4>(P0, B, st);
        return hipErrorInvalidValue;
    }
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
