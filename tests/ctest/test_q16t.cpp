// test_q16t.cpp -- CPU test of the lane-serial GF(16) check-node code (csrc/kernels/ems_q16t_lane.h): the SAME functions the
// gfx950 kernel k_decode_q16t runs per lane, compiled for the host with an array as the scratch column, against the oracle
// (oracle/ems_oracle.c, pinned against the reference): sorting networks exhaustively (0-1 principle), ElementaryStep on random /
// coarse / short lists, and whole decodes of the 64800-bit GF(16) code where every check node goes through select_lane -> 18 x
// es_lane -> output_lane (a check node that reports `bad` takes the oracle's CheckPassLogEMS instead, as the kernel falls back to the
// merge-form code).  Exit code 0 = everything bit-exact and the fallback rate is small.
//   usage: test_q16t <matrix> [frames] [EbN0] [sweeps] [NbOper]      (Q16T_GENERIC=1: the 48-cell form even at NbOper = 16)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "ems_q16t_lane.h"
extern "C" {
#include "ems_oracle.h"
}

using namespace q16t;

struct HostScr {
    uint32_t a[16];
    uint32_t ld(int i) const { return a[i]; }
    void st(int i, uint32_t u) { a[i] = u; }
    void amin(uint32_t i, uint32_t u) { a[i] = a[i] < u ? a[i] : u; }
    void sync() {}
};

static unsigned long long rng_state = 88172645463325252ull;
static unsigned rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 11); }
static float rndf() { return (float)(rnd() & 0xFFFFFF) / 16777216.0f; }

static int check_networks()
{
    for (unsigned v = 0; v < 65536; ++v) {
        uint32_t k[16], ref[16];
        for (int i = 0; i < 16; ++i) ref[i] = k[i] = (v >> i) & 1u;
        for (int i = 0; i < 16; ++i) for (int j = i + 1; j < 16; ++j) if (ref[j] < ref[i]) { uint32_t t = ref[i]; ref[i] = ref[j]; ref[j] = t; }
        const uint32_t h = top8of16(k);
        if (memcmp(k, ref, 32) != 0 || h != ref[8]) { printf("network fails on %04x\n", v); return 1; }
    }
    for (int it = 0; it < 20000; ++it) {
        uint32_t k[16], ref[16];
        for (int i = 0; i < 16; ++i) ref[i] = k[i] = rnd();
        for (int i = 0; i < 16; ++i) for (int j = i + 1; j < 16; ++j) if (ref[j] < ref[i]) { uint32_t t = ref[i]; ref[i] = ref[j]; ref[j] = t; }
        const uint32_t h = top8of16(k);
        if (memcmp(k, ref, 32) != 0 || h != ref[8]) { printf("network fails on random keys\n"); return 1; }
    }
    return 0;
}

// a list as the check node sees them: len valid entries ascending (first one 0 when `norm`), distinct symbols, (1e5, -1) behind
static void make_list(float *v, int *g, int len, bool norm, float step)
{
    float x = norm ? 0.0f : rndf() * 3.0f;
    int used[16] = {0};
    for (int k = 0; k < 8; ++k) {
        if (k < len) {
            v[k] = step > 0 ? floorf(x / step) * step : x;
            x += step > 0 ? (float)(rnd() % 3) * step : rndf() * 2.5f;
            int s;
            do s = (int)(rnd() & 15); while (used[s]);
            used[s] = 1;
            g[k] = s;
        } else { v[k] = 1e5f; g[k] = -1; }
    }
}
static List to_list(const float *v, const int *g, const orc_gf *gf)
{
    List L;
    L.s = 0;
    for (int k = 0; k < 8; ++k) { L.v[k] = v[k]; L.s |= (uint32_t)(g[k] < 0 ? 0 : gf->idx2bin[g[k]]) << (4 * k); }
    return L;
}

static int check_es(const orc_gf *gf, long trials, int *n_bad_cont)
{
    long bad_cont = 0, tot_cont = 0, bad_all = 0;
    for (long it = 0; it < trials; ++it) {
        const int kind = (int)(it % 8);                       // 0..4 continuous full, 5 short lists, 6 coarse grid, 7 coarse + short
        const float step = kind >= 6 ? 0.25f : 0.0f;
        const int la = (kind == 5 || kind == 7) ? 1 + (int)(rnd() % 8) : 8, lb = (kind == 5 || kind == 7) ? 1 + (int)(rnd() % 8) : 8;
        float va[8], vb[8], vo[8];
        int ga[8], gb[8], go[8];
        make_list(va, ga, la, (rnd() & 1) != 0, step);
        make_list(vb, gb, lb, (rnd() & 1) != 0, step);
        const int nop = (const int[]){16, 16, 16, 12, 24, 8, 20, 13}[rnd() & 7];
        orc_elementary_step(va, vb, ga, gb, vo, go, gf, 8, nop);
        HostScr bins;
        for (int i = 0; i < 16; ++i) bins.a[i] = INF;
        List out;
        const bool bad = (nop == 16 && (it & 16)) ? es_lane<true>(to_list(va, ga, gf), to_list(vb, gb, gf), nop, bins, out)
                                                   : es_lane<false>(to_list(va, ga, gf), to_list(vb, gb, gf), nop, bins, out);
        for (int i = 0; i < 16; ++i) if (bins.a[i] != INF) { printf("bins not restored\n"); return 1; }
        if (kind < 5) { ++tot_cont; bad_cont += bad; }
        bad_all += bad;
        if (bad) continue;
        for (int k = 0; k < 8; ++k) {
            const int sym = go[k];
            const bool ok = f2u(out.v[k]) == f2u(vo[k]) && (sym < 0 ? f2u(out.v[k]) == SENT : (int)gf->bin2idx[(out.s >> (4 * k)) & 15u] == sym);
            if (!ok) {
                printf("ES mismatch (trial %ld kind %d nbOper %d) at entry %d: lane (%g, %u) oracle (%g, %d)\n", it, kind, nop, k, out.v[k],
                       (unsigned)gf->bin2idx[(out.s >> (4 * k)) & 15u], vo[k], sym);
                return 1;
            }
        }
    }
    printf("ElementaryStep: %ld trials bit-exact where decided; undecided %ld in all, %ld of %ld continuous full-length ones\n", trials, bad_all, bad_cont, tot_cont);
    *n_bad_cont = (int)bad_cont;
    return bad_cont * 200 > tot_cont;                          // the closed form must decide > 99.5 % of the ordinary tasks
}

int main(int argc, char **argv)
{
    if (check_networks()) return 1;
    printf("sorting networks: exhaustive 0-1 check passed\n");
    orc_gf gf;
    if (orc_gf_init(&gf, 16)) return 2;
    int nb = 0;
    if (check_es(&gf, 400000, &nb)) { printf("too many undecided ordinary tasks\n"); return 1; }
    if (argc < 2) return 0;

    // ---- whole decodes ----
    orc_code *c = orc_load_code(argv[1], ORC_FMT_AUTO);
    if (!c || c->q != 16 || c->dc_max != 8) { printf("need a GF(16) dc = 8 code\n"); return 2; }
    const int frames = argc > 2 ? atoi(argv[2]) : 2;
    const float EbN = argc > 3 ? (float)atof(argv[3]) : 1.5f;
    const int itmax = (argc > 4 ? atoi(argv[4]) : 5) + 1;
    const int nm = 8, nop = argc > 5 ? atoi(argv[5]) : 16, q = 16, N = c->N, M = c->M, E = c->E;
    const bool small16 = nop == 16 && !getenv("Q16T_GENERIC");
    const float offset = 0.3f;
    orc_encoder *enc = orc_encoder_create(c, &gf);
    orc_rng rng = {0};
    const float sigma = orc_sigma(c->rate, EbN);
    std::vector<int> cw(N), dec_ref(N), dec(N);
    std::vector<float> noisy((size_t)N * c->logq), llr((size_t)N * q), app_ref((size_t)N * q), ctov_ref((size_t)E * q), app((size_t)N * q), ctov((size_t)E * q);
    long cn_total = 0, cn_bad = 0;
    for (int f = 0; f < frames; ++f) {
        orc_draw_frame(c, &gf, enc, &rng, sigma, cw.data(), noisy.data());
        orc_llr(c, &gf, noisy.data(), sigma, llr.data());
        int synd = 0;
        orc_decode_frame(c, &gf, llr.data(), nm, nop, offset, itmax, NULL, 0, dec_ref.data(), &synd, app_ref.data(), ctov_ref.data(), NULL, NULL, NULL);
        app = llr;
        std::fill(ctov.begin(), ctov.end(), 0.0f);
        HostScr scr;
        for (int i = 0; i < 16; ++i) scr.a[i] = INF;
        for (int sweep = 0; sweep < itmax - 1; ++sweep)
            for (int m = 0; m < M; ++m) {
                const int e0 = c->row_ptr[m];
                float mvc[8][16];
                List L[NBUF];
                bool bad = false;
                for (int t = 0; t < 8; ++t) {
                    const float *a = &app[(size_t)c->col[e0 + t] * q], *cv = &ctov[(size_t)(e0 + t) * q];
                    for (int g = 0; g < q; ++g) mvc[t][g] = a[g] - cv[g];
                    HostScr s2;
                    bad = select_lane(mvc[t], rot_table(gf.idx2bin, c->coef[e0 + t] - 1), s2, L[t]) || bad;
                }
                for (int st = 1; st <= 6 && !bad; ++st) {
                    List in[4][2], out[4];
                    int dst[4], nt = 0;
                    for (int j = 0; j < 4; ++j) {
                        int a, b, o;
                        if (!stage_task(st, j, a, b, o)) continue;
                        in[nt][0] = L[a]; in[nt][1] = L[b]; dst[nt] = o; ++nt;         // a stage reads all its operands before it writes
                    }
                    for (int j = 0; j < nt; ++j) bad = (small16 ? es_lane<true>(in[j][0], in[j][1], nop, scr, out[j]) : es_lane<false>(in[j][0], in[j][1], nop, scr, out[j])) || bad;
                    for (int j = 0; j < nt; ++j) L[dst[j]] = out[j];
                }
                ++cn_total;
                const bool audit = getenv("Q16T_AUDIT") != NULL;            // every decided check node against the oracle's, first mismatch printed
                if (bad || audit) {                          // the kernel's fallback: the literal check node (here: the oracle's)
                    cn_bad += bad;
                    float vl[64], mcv[128], w[16];
                    int vg[64];
                    for (int t = 0; t < 8; ++t) {
                        memcpy(w, mvc[t], sizeof w);
                        for (int k = 0; k < nm; ++k) {
                            float best = 1e5f; int bg = 0;
                            for (int g = 0; g < q; ++g) if (w[g] < best) { best = w[g]; bg = g; }
                            vl[t * nm + k] = best; vg[t * nm + k] = bg; w[bg] = 1e5f;
                        }
                        for (int k = 1; k < nm; ++k) vl[t * nm + k] -= vl[t * nm];
                        vl[t * nm] = 0.0f;
                    }
                    orc_check_node(8, nm, &gf, c->coef + e0, vl, vg, nop, offset, mcv);
                    if (!bad) {
                        for (int t = 0; t < 8; ++t) {
                            EdgeOut eo;
                            HostScr s2;
                            output_lane(L[buf_out(t)], mvc[t], offset, irot_table(gf.bin2idx, c->coef[e0 + t] - 1), s2, eo);
                            float cv[16];
                            for (int g = 0; g < q; ++g) cv[g] = eo.dflt;
                            for (int k = 0; k < eo.cnt; ++k) cv[(eo.syms >> (4 * k)) & 15u] = eo.val[k];
                            if (memcmp(cv, mcv + t * q, sizeof cv) != 0) {
                                printf("AUDIT: sweep %d check node %d edge %d differs\n  selected:", sweep, m, t);
                                for (int tt = 0; tt < 8; ++tt) { printf("\n   in%d:", tt); for (int k = 0; k < 8; ++k) printf(" (%.9g,%d)", vl[tt * nm + k], vg[tt * nm + k]); }
                                printf("\n  lane row:"); for (int g = 0; g < q; ++g) printf(" %.9g", cv[g]);
                                printf("\n  orcl row:"); for (int g = 0; g < q; ++g) printf(" %.9g", mcv[t * q + g]);
                                printf("\n");
                                return 1;
                            }
                        }
                    }
                    for (int t = 0; t < 8; ++t)
                        for (int g = 0; g < q; ++g) {
                            ctov[(size_t)(e0 + t) * q + g] = mcv[t * q + g];
                            app[(size_t)c->col[e0 + t] * q + g] = mcv[t * q + g] + mvc[t][g];
                        }
                    for (int t = 0; t < 8; ++t) {                      // Decision (tools.c:312-330) of the rows just written
                        const float *a = &app[(size_t)c->col[e0 + t] * q];
                        float best = 1e5f; int bg = 0;
                        for (int g = 0; g < q; ++g) if (a[g] < best) { best = a[g]; bg = g; }
                        dec[c->col[e0 + t]] = bg;
                    }
                    continue;
                }
                for (int t = 0; t < 8; ++t) {
                    EdgeOut eo;
                    HostScr s2;
                    output_lane(L[buf_out(t)], mvc[t], offset, irot_table(gf.bin2idx, c->coef[e0 + t] - 1), s2, eo);
                    float *cv = &ctov[(size_t)(e0 + t) * q];
                    for (int g = 0; g < q; ++g) cv[g] = eo.dflt;
                    for (int k = 0; k < eo.cnt; ++k) cv[(eo.syms >> (4 * k)) & 15u] = eo.val[k];
                    memcpy(&app[(size_t)c->col[e0 + t] * q], eo.napp, sizeof eo.napp);
                    dec[c->col[e0 + t]] = eo.decide;
                }
            }
        const bool ok = memcmp(app.data(), app_ref.data(), app.size() * 4) == 0 && memcmp(ctov.data(), ctov_ref.data(), ctov.size() * 4) == 0 &&
                        memcmp(dec.data(), dec_ref.data(), (size_t)N * 4) == 0;
        printf("frame %d: %d sweeps, APP / CtoV / decisions %s\n", f, itmax - 1, ok ? "bit-exact" : "DIFFER");
        if (!ok) return 1;
    }
    printf("check nodes: %ld visits, %ld undecided (%.3f %%)\n", cn_total, cn_bad, 100.0 * (double)cn_bad / (double)cn_total);
    return cn_bad * 50 > cn_total;                             // under 2 % of the visits may need the fallback
}
